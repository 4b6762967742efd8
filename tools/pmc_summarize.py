#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output per kernel: mean HBM bytes per dispatch.

    python tools/pmc_summarize.py <dir-with-*_counter_collection.csv> [...]

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


for d in sys.argv[1:]:
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
    for k in sorted(agg, key=lambda k: -sum(v[0] for v in agg[k].values())):
        for c, v in sorted(agg[k].items()):
            if c in CORR:
                scale, note = CORR[c]
                print('%-12s %-44s launches=%4d  bytes/launch=%.4g GB (%s)' % (c, k, v[1], v[0] / v[1] * scale / 1e9, note))
            else:
                print('%-12s %-44s launches=%4d  mean=%.4g' % (c, k, v[1], v[0] / v[1]))
