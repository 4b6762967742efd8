#!/usr/bin/env python3
"""Board power and clocks while a command keeps ONE kernel (or the whole step) busy (developer tool).

    python tools/power_probe.py -- python tools/bench_gemm.py --only 4 --reps 12000
    python tools/power_probe.py -- python tools/bench_conv.py --gen f4 --only 0 --reps 3000
    python tools/power_probe.py -- python bench.py --no-cpu-baseline --no-prof --no-full-trajectory --steps 300

Starts the command as a child, samples `rocm-smi --showpower --showclocks --showtemp --json` twice a second while it runs and
prints the samples (seconds, shader clock, package power) and the tail of the child's output: what the chip draws and which
shader clock it holds under that work.  The package limit is `rocm-smi --showmaxpower` (1400 W on the MI355X boxes)."""
import json
import re
import subprocess
import sys
import time

if '--' not in sys.argv:
    sys.exit(__doc__)
cmd = sys.argv[sys.argv.index('--') + 1:]
child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
t0 = time.time()
rows = []
while child.poll() is None:
    try:
        r = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showtemp', '--json'], capture_output=True, text=True, timeout=10)
        c = next(iter(json.loads(r.stdout).values()))
        sclk = int(re.sub('[^0-9]', '', c.get('sclk clock speed:', '0')) or 0)
        pw = float(c.get('Current Socket Graphics Package Power (W)', 'nan'))
        tj = c.get('Temperature (Sensor junction) (C)', '')
        rows.append((time.time() - t0, sclk, pw, tj))
    except Exception as e:   # a sample that fails is a gap, not an error of the probe
        rows.append((time.time() - t0, -1, float('nan'), str(e)[:60]))
    time.sleep(0.5)
out = child.stdout.read()
print('# %s' % ' '.join(cmd))
for t, s, p, tj in rows:
    print('%6.1f s  sclk %4d MHz  %6.1f W  Tj %s' % (t, s, p, tj))
busy = [r for r in rows if r[2] > 600]
if busy:
    last = busy[len(busy) // 2:]   # the second half of the busy samples: the steady state of the LAST phase of the command
    print('# steady state (second half of the %d samples above 600 W): sclk %.0f MHz, %.0f W' % (len(busy), sum(r[1] for r in last) / len(last), sum(r[2] for r in last) / len(last)))
print(out[-2500:])
sys.exit(child.returncode)
