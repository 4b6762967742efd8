#!/usr/bin/env python3
"""Board power and clocks while ONE launch shape of tools/bench_gemm.py runs back to back (developer tool).

    python tools/power_probe.py <shape index> [seconds]

Starts bench_gemm.py --only <i> --reps N as a child, samples `rocm-smi --showpower --showclocks --json` twice a second
while it runs, and prints the samples: what the chip draws and which shader clock it holds under this kernel."""
import json
import subprocess
import sys
import time
import os

idx = sys.argv[1]
reps = sys.argv[2] if len(sys.argv) > 2 else '20000'
env = dict(os.environ)
child = subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bench_gemm.py'), '--only', idx, '--reps', reps],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
t0 = time.time()
rows = []
while child.poll() is None:
    try:
        r = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showtemp', '--json'], capture_output=True, text=True, timeout=10)
        j = json.loads(r.stdout)
        c = next(iter(j.values()))
        keep = {k: v for k, v in c.items() if any(s in k.lower() for s in ('power', 'sclk', 'mclk', 'fclk', 'junction', 'hotspot'))}
        rows.append((time.time() - t0, keep))
    except Exception as e:
        rows.append((time.time() - t0, {'error': str(e)[:200]}))
    time.sleep(0.5)
out = child.stdout.read()
for t, k in rows:
    print('%6.1f s  %s' % (t, json.dumps(k)))
print(out[-1500:])
