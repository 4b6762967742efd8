#!/usr/bin/env python3
"""Kernel durations and inter-kernel gaps from a rocprofv3 --kernel-trace CSV (developer tool).
usage: tools/trace_gaps.py <kernel_trace.csv> [tail_kernels]
Takes the LAST `tail_kernels` dispatches (default 3000: graph replays of the timed steps), sorts them by start time and
prints: busy time, idle time between consecutive kernels, and the per-kernel table (count, mean duration, mean gap BEFORE it)."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:60]))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
rows = rows[-n:]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = defaultdict(list)
durs = defaultdict(list)
for (s0, e0, _), (s1, e1, k) in zip(rows, rows[1:]):
    gaps[k].append(max(0, s1 - e0))
    durs[k].append(e1 - s1)
print(f'dispatches {len(rows)}  span {span / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms  idle {(span - busy) / 1e6:.3f} ms '
      f'({100 * (span - busy) / span:.1f} %)')
print(f'{"kernel":60s} {"n":>6s} {"dur us":>8s} {"gap us":>8s} {"sum ms":>8s}')
for k in sorted(durs, key=lambda k: -sum(durs[k]) - sum(gaps[k])):
    print(f'{k:60s} {len(durs[k]):6d} {sum(durs[k]) / len(durs[k]) / 1e3:8.2f} {sum(gaps[k]) / len(gaps[k]) / 1e3:8.2f} '
          f'{(sum(durs[k]) + sum(gaps[k])) / 1e6:8.3f}')
