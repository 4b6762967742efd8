#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output per kernel: mean counter value per dispatch.

    python tools/pmc_summarize.py <dir-with-*_counter_collection.csv> [...]
"""
import csv
import glob
import os
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get('Kernel_Name', row.get('Kernel Name', '?'))
                short = name.split('(')[0][-60:]
                ctr = row.get('Counter_Name', row.get('Counter Name'))
                val = float(row.get('Counter_Value', row.get('Counter Value', 0)))
                a = agg[short][ctr]
                a[0] += val
                a[1] += 1
    print('==', d)
    for k in sorted(agg, key=lambda k: -sum(v[0] for v in agg[k].values())):
        parts = ['%s mean=%.4g n=%d total=%.4g' % (c, v[0] / max(v[1], 1), v[1], v[0]) for c, v in sorted(agg[k].items())]
        print('%-62s %s' % (k, ' | '.join(parts)))
