#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output per kernel: mean HBM bytes per dispatch.

    python tools/pmc_summarize.py <dir-with-*_counter_collection.csv> [...]
    python tools/pmc_summarize.py --json <workload> <summary-file-path> <dir> [...]
        also writes profiles/pmc_traffic.json: fetch + write bytes per launch per kernel family, stamped with the digest of
        the kernel sources (bench.py reports `roofline.traffic` from it only while that digest matches the build it runs)

FETCH_SIZE / WRITE_SIZE count KB; on gfx950 FETCH_SIZE under-reports by 2x
(/opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section), so it is doubled here.
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

CORR = {'FETCH_SIZE': (2048.0, 'counter KB x1024 x2 gfx950 correction'), 'WRITE_SIZE': (1024.0, 'counter KB x1024')}


def short_name(name):
    m = re.search(r'(k_\w+(?:<[^>]*>)?)', name)
    return m.group(1) if m else name[:60]


argv = sys.argv[1:]
as_json = None
if argv and argv[0] == '--json':
    as_json = dict(workload=argv[1], file=argv[2])
    argv = argv[3:]
total = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))   # kernel family (template arguments dropped) -> counter -> [sum, n]

for d in argv:
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get('Kernel_Name', row.get('Kernel Name', '?'))
                ctr = row.get('Counter_Name', row.get('Counter Name'))
                val = float(row.get('Counter_Value', row.get('Counter Value', 0)))
                a = agg[short_name(name)][ctr]
                a[0] += val
                a[1] += 1
                fam = total[short_name(name).split('<')[0]][ctr]
                fam[0] += val
                fam[1] += 1
    for k in sorted(agg, key=lambda k: -sum(v[0] for v in agg[k].values())):
        for c, v in sorted(agg[k].items()):
            if c in CORR:
                scale, note = CORR[c]
                print('%-12s %-44s launches=%4d  bytes/launch=%.4g GB (%s)' % (c, k, v[1], v[0] / v[1] * scale / 1e9, note))
            else:
                print('%-12s %-44s launches=%4d  mean=%.4g' % (c, k, v[1], v[0] / v[1]))

if as_json:
    import json
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    from bench import source_digest
    kernels = {}
    for fam, ctrs in total.items():
        if 'FETCH_SIZE' in ctrs and 'WRITE_SIZE' in ctrs:
            kernels[fam] = sum(ctrs[c][0] / ctrs[c][1] * CORR[c][0] for c in ('FETCH_SIZE', 'WRITE_SIZE'))
    out = dict(source_digest=source_digest(), workload=as_json['workload'], file=as_json['file'], kernels=kernels)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles', 'pmc_traffic.json')
    json.dump(out, open(path, 'w'), indent=1, sort_keys=True)
