#!/usr/bin/env python3
"""bench.py -- samples/sec of DLPM's reverse-sampling loop at T=1000 on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: started under torch.distributed.run (RANK / WORLD_SIZE set) the process is one rank; started
plainly, it spawns the N ranks itself (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 ...`, before this process has touched the GPU), waits, and exits with their code.

Workload (N = 1 and per GPU for N > 1, weak scaling): BASELINE.json configs[2] -- CIFAR-10 shaped
state [1024, 3, 32, 32], the reference's cifar10.yml UNet (39.6 M parameters, attention at 8x8 and
4x4), random init with the zero-initialised tensors re-drawn (dlpm_amd/weights.py), T = 1000,
alpha = 1.7, clamp_a = 10, clamp_eps = 50, fp32, Philox noise keyed by the global sample index.

A "step" is ONE reverse step (UNet forward + fused update) over the whole batch: W warm-up steps,
then exactly K timed steps between barrier + synchronize pairs, max over ranks -> `ms_per_step`.
`value` [samples/s at T=1000] is then MEASURED, not extrapolated: one whole trajectory -- init (A
draws, tables, x_T) + all T-1 = 999 reverse steps + the final RCCL all-gather (N > 1) -- is timed
between barriers in the same run (`full_trajectory_s`, max over ranks) and value = B_total / that.
`--no-full-trajectory` skips it; value is then init + 999 * ms_per_step + gather and says so
(`value_source`).  With --steps 999 --warmup 0 the K-step region is itself a whole trajectory.
"""
import os as _os
# (before torch is imported: the CPU-baseline leg's OpenMP barriers wait passively -- spinning threads lose whole steps to any
#  core a neighbour on the GPU box's shared host takes; the GPU path does not use OpenMP)
_os.environ.setdefault('OMP_WAIT_POLICY', 'PASSIVE')
_os.environ.setdefault('GOMP_SPINCOUNT', '10000')
import argparse
import ctypes as C
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (config, per-GPU batch, T, alpha)
    'cifar10_unet_b1024_T1000': ('cifar10', 1024, 1000, 1.7),
    'mnist_unet_b256_T1000': ('mnist', 256, 1000, 1.7),
    'celeba64_unet_b256_T1000': ('celeba64', 256, 1000, 1.8),   # per-GPU shard of BASELINE configs[4]
}
METRIC = {
    'cifar10': 'samples/sec at T=1000 (CIFAR-10 32x32, alpha=1.7)',
    'mnist': 'samples/sec at T=1000 (MNIST 32x32, alpha=1.7) [parity config, not the headline]',
    'celeba64': 'samples/sec at T=1000 (CelebA 64x64, alpha=1.8) [builder-defined net, not the headline]',
}
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 (the guide's figure; tools/mb/mfma_bf16.hip measures 2.3-2.5 PFLOP/s in a bare loop)
PEAK_HBM_GBS = 8000.0
# HBM traffic of the dominant kernel comes from rocprofv3 PMC passes (separate --pmc runs, FETCH_SIZE doubled per the
# gfx950 guide), which cannot run inside this process.  tools/pmc_summarize.py --json writes profiles/pmc_traffic.json
# with the digest of the kernel sources it measured; the figure is reported only while that digest matches this build.
PMC_TRAFFIC_JSON = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')


def source_digest():
    """sha256 over the kernel sources: identifies the build a PMC measurement belongs to."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'dlpm_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h', '.cpp')):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def pmc_traffic(kname, workload):
    try:
        rec = json.load(open(PMC_TRAFFIC_JSON))
    except Exception:
        return None, 'no PMC pass recorded (profiles/pmc_traffic.json absent)'
    if rec.get('source_digest') != source_digest():
        return None, 'the recorded PMC pass (%s) measured an older build of the kernels; not reported' % rec.get('file')
    if rec.get('workload') != workload:
        return None, 'the recorded PMC pass measured workload %s' % rec.get('workload')
    v = rec.get('kernels', {}).get(kname)
    if v is None:
        return None, 'kernel absent from %s' % rec.get('file')
    return v, 'HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, mean over the launches of a step) from the rocprofv3 PMC passes of this build: %s' % rec.get('file')


def parse_prof(txt):
    out = {}
    for line in txt.strip().splitlines():
        name, n, ms, fl, by = line.split()
        out[name] = dict(launches=int(n), ms=float(ms), flops=float(fl), bytes=float(by))
    return out


def physical_cores():
    """Distinct (package, core) pairs of the CPUs this process may run on, from /proc/cpuinfo; None if it cannot be read."""
    try:
        allowed = os.sched_getaffinity(0)
        seen, cur = set(), {}
        for line in open('/proc/cpuinfo').read().splitlines() + ['']:
            if ':' in line:
                k, v = line.split(':', 1)
                cur[k.strip()] = v.strip()
            elif cur:
                if int(cur.get('processor', -1)) in allowed:
                    seen.add((cur.get('physical id', '0'), cur.get('core id', cur.get('processor'))))
                cur = {}
        return len(seen) or None
    except Exception:
        return None


def cpu_baseline(cfg_name, T, alpha, B=32, steps=60, warm=2, threads=None):
    """The oracle (a torch-CPU port of the reference loop) on this box's host cores -- SURVEY.md 8d / BASELINE.md 4:
    B = 32, the real T-step schedule and tables, the thread count that runs it fastest, `warm` untimed reverse steps, then `steps` >= 20
    reverse steps timed ONE BY ONE;
    the figure is the MEDIAN OF THREE: the steps form three consecutive blocks of `steps` / 3 >= 20, each block gives B / (set-up +
    999 x its median step), and the middle one is reported (all three are printed).  Both data
    layouts are timed on the same steps: "reference-faithful" (full-size [T,B,C,H,W] A / Sigma tensors, the schedule
    re-broadcast by `repeat` twice per step: what the reference executes, and the reported `value`) and "scalar-table"
    ([T,B] tables); the network forward is common to both.  The table set-up (A expansion + Sigma recursion) is timed
    once per layout and enters the trajectory figure.  Plus the C++ host library's noise streams (1 core)."""
    import numpy as np
    from oracle import nets, sampler as osampler, process as P
    import dlpm_amd
    from dlpm_amd import _lib
    p = dlpm_amd.load_config(cfg_name)
    torch.manual_seed(1234)
    net = dlpm_amd.rerandomize_(dlpm_amd.init_model_by_parameter(p), 4321)
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    heads = p['model']['num_heads']
    shape = [B, p['data']['channels'], p['data']['image_size'], p['data']['image_size']]
    model = lambda x, t: nets.unet_forward(sd, x, t, heads)
    # one torch thread per PHYSICAL core (OpenMP's spinning barriers lose badly to SMT siblings and to any core the host does not
    # actually schedule: the figure moved by 15 % between boxes with every logical CPU in use)
    # The GPU boxes are slices of a shared two-socket host (256 logical CPUs, neighbours at load 30): the oracle's step at B = 32 takes
    # 0.90 s on 128 threads, 0.42 s on 64 and 0.28 s on 32 (profiles/r03/cpu_baseline_thread_sweep.txt) -- more threads than the net's
    # small per-layer work can feed only add barrier traffic across sockets.  The baseline is the BEST of a short sweep (two timed
    # forwards per candidate), stated in the JSON; OpenMP waits passively (set at the top of this file).
    logical = torch.get_num_threads()
    phys = physical_cores() or logical
    tried = {}
    if threads:
        cores = threads
    else:
        xs = torch.randn(shape, generator=torch.Generator().manual_seed(1))
        ts = torch.full((B,), 0.5)
        with torch.inference_mode():
            for cand in [c for c in (8, 16, 24, 32, 48, 64, 96, 128) if c <= phys] or [phys]:
                torch.set_num_threads(cand)
                model(xs, ts)
                best = 1e9
                for _ in range(3):          # the FASTEST of three forwards: a neighbour's burst must not pick the thread count
                    t0 = time.perf_counter()
                    model(xs, ts)
                    best = min(best, time.perf_counter() - t0)
                tried[cand] = round(best, 4)
        cores = min(tried, key=tried.get)
        # hysteresis: on the shared host the forwards of neighbouring candidates differ by less than their own run-to-run noise, and a
        # different count in consecutive runs moved the figure by more than the count itself is worth -- stay on 32 (the sweep's optimum
        # on this pool's hosts, profiles/r03/cpu_baseline_thread_sweep.txt) unless another count is more than 10 % faster
        prefer = 32 if 32 in tried else cores
        if tried[prefer] <= 1.10 * tried[cores]:
            cores = prefer
    torch.set_num_threads(cores)
    ev = p['eval']['dlpm']
    with torch.inference_mode():
        streams = osampler.Streams(0, 0)
        g, bg, s_, bs = P.schedule(T, alpha)
        t0 = time.perf_counter()
        A = torch.stack([streams.skewed_levy(alpha, B, ev['clamp_a']) for _ in range(T)])
        draw_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        Sig = P.sigma_table(A, g, s_)
        init_scalar = draw_s + time.perf_counter() - t0
        t0 = time.perf_counter()
        Af, Sigf = osampler.full_size_tables(A, shape, g, s_)
        init_full = draw_s + time.perf_counter() - t0
        del Af
        x = bs[-1] * torch.clamp(torch.sqrt(streams.skewed_levy(alpha, B, None)).view(-1, 1, 1, 1) * streams.randn(shape),
                                 -ev['clamp_eps'], ev['clamp_eps'])
        net_t, full_t, scal_t = [], [], []
        same_bits = True          # the two data layouts step to the same bits (recorded, not asserted: a NaN must not void the bench line)
        i = T - 1
        for k in range(warm + steps):
            t0 = time.perf_counter()
            eps = model(x, torch.full((B,), i, dtype=torch.int64).float() * (1.0 / T))
            z = streams.randn(shape)                          # th.randn_like in p_sample: part of the reference's step
            t1 = time.perf_counter()
            xf = osampler.full_size_step(x, eps, i, Sigf, g, bs, z)
            t2 = time.perf_counter()
            xs, _, _ = P.dlpm_step(x, eps, i, Sig, g, bs, z)
            t3 = time.perf_counter()
            same_bits = same_bits and bool(torch.equal(torch.nan_to_num(xf, nan=7.0), torch.nan_to_num(xs, nan=7.0)))
            x, i = xf, i - 1
            if k >= warm:
                net_t.append(t1 - t0); full_t.append(t2 - t1); scal_t.append(t3 - t2)
        net_t, full_t, scal_t = np.array(net_t), np.array(full_t), np.array(scal_t)
        # the C++ host library (libdlpm_amd's parity streams, one core): seconds of noise per B=32 trajectory
        L = _lib.lib()
        mt = _lib.MT19937()
        _lib.check(L.dlpm_mt19937_seed(C.byref(mt), 0))
        nz, na = 1 << 21, 1 << 17
        buf = np.empty(nz, np.float32)
        t0 = time.perf_counter()
        _lib.check(L.dlpm_randn_host_f32(C.byref(mt), nz, buf.ctypes.data))
        randn_rate = nz / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        _lib.check(L.dlpm_skewed_levy_host_f32(C.byref(mt), float(alpha), na, -1.0, buf.ctypes.data))
        levy_rate = na / (time.perf_counter() - t0)
    D = shape[1] * shape[2] * shape[3]
    nb = steps // 3
    blocks_full = [float(np.median((net_t + full_t)[j * nb:(j + 1) * nb])) for j in range(3)]     # three runs of >= 20 steps
    blocks_scal = [float(np.median((net_t + scal_t)[j * nb:(j + 1) * nb])) for j in range(3)]
    med_full, med_scal = sorted(blocks_full)[1], sorted(blocks_scal)[1]                           # ... and their median
    thirds = [round(v, 4) for v in blocks_full]
    traj_full, traj_scal = init_full + (T - 1) * med_full, init_scalar + (T - 1) * med_scal
    return dict(value=round(B / traj_full, 6), unit='samples/s at T=1000', cores=cores, threads=cores, physical_cores=phys,
                logical_cpus=logical, layouts_bit_identical=same_bits, kind='port',
                sample='oracle (torch-CPU port of the reference loop), same UNet and schedule, B=%d, %d untimed + %d timed reverse '
                       'steps (t = %d..%d) on %d torch threads (the fastest of a sweep over %s, 32 kept unless another count is > 10 %% faster, on a host of %d physical cores / %d default torch '
                       'threads); value = the median of three blocks of %d steps, each B / (table set-up + 999 x its median step), in the reference\'s '
                       'full-size [T,B,C,H,W] layout'
                       % (B, warm, steps, T - 1 - warm, T - warm - steps, cores, sorted(tried) if tried else [cores], phys, logical, steps // 3),
                threads_tried_forward_s=tried,
                median_step_s=round(med_full, 4), median_step_s_of_the_three_blocks=thirds,
                value_of_the_three_blocks=[round(B / (init_full + (T - 1) * v), 6) for v in blocks_full],
                step_s_min_max=[round(float((net_t + full_t).min()), 4), round(float((net_t + full_t).max()), 4)],
                network_share_of_step=round(float(np.median(net_t)) / med_full, 4),
                table_setup_s=round(init_full, 3), timed_cpu_seconds=round(float((net_t + full_t + scal_t).sum()), 1),
                scalar_table_variant=dict(value=round(B / traj_scal, 6), median_step_s=round(med_scal, 4), table_setup_s=round(init_scalar, 3),
                                          note='[T,B] tables instead of [T,B,C,H,W]: same arithmetic per element, same bits'),
                host_library=dict(randn_per_s=round(randn_rate), skewed_levy_per_s=round(levy_rate), cores=1,
                                  noise_seconds_per_trajectory=round(T * B * D / randn_rate + (T + 1) * B / levy_rate, 2),
                                  note='libdlpm_amd C++ parity streams (MT19937 -> torch-compatible randn, scipy-compatible CMS): the '
                                       'noise of one B=%d, T=%d trajectory' % (B, T)))


def _proc_table():
    """{pid: (ppid, starttime)} of every process visible in /proc."""
    tab = {}
    for d in os.listdir('/proc'):
        if d.isdigit():
            try:
                f = open('/proc/%s/stat' % d).read()
                rest = f[f.rindex(')') + 2:].split()      # fields after "(comm)": state ppid ... starttime is the 20th of them
                tab[int(d)] = (int(rest[1]), rest[19])
            except (OSError, ValueError, IndexError):
                pass
    return tab


def _descendants(root):
    tab = _proc_table()
    kids = {}
    for pid, (ppid, st) in tab.items():
        kids.setdefault(ppid, []).append(pid)
    out, todo = {}, [root]
    while todo:
        for c in kids.get(todo.pop(), []):
            if c not in out:
                out[c] = tab[c][1]
                todo.append(c)
    return out


def spawn_ranks(n, argv, timeout_s=None):
    """`bench.py --gpus N` started plainly: launch the N ranks as FRESH children.  This process has not initialised the GPU
    (importing torch does not), and it never execs: it waits and hands the children's exit code on.  A rank that dies
    mid-trajectory leaves its peers in a barrier: torch.distributed.run notices the dead worker, terminates the others
    and exits non-zero; should the ranks still be running at the deadline (--rank-timeout, default 2 h), the launcher gets
    SIGTERM and the parent exits 124.  torch.distributed.run puts every rank into a session of its own, so neither a process
    group nor the launcher's death reaches them (a rank orphaned in a gloo / RCCL barrier spins on its cores for the 30 minutes
    of the collective's timeout): the parent therefore keeps the set of the launcher's descendants it has SEEN -- exact PIDs with
    their start times, never a pattern -- and ends whichever of them outlive the launcher."""
    import signal
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    child = subprocess.Popen(cmd, env=env, start_new_session=True)
    seen = {}                       # pid -> start time of every descendant of the launcher observed so far
    deadline = None if timeout_s is None else time.monotonic() + timeout_s

    def sweep():
        tab = _proc_table()
        for pid, st in seen.items():
            if pid in tab and tab[pid][1] == st:      # still the process we saw (same start time), still alive
                try:
                    os.kill(pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass

    def on_term(signum, frame):     # the parent itself is being ended (a driver's timeout): take the ranks along
        raise KeyboardInterrupt
    signal.signal(signal.SIGTERM, on_term)
    rc = None
    try:
        while True:
            rc = child.poll()
            if rc is not None:
                break
            seen.update(_descendants(child.pid))
            if deadline is not None and time.monotonic() > deadline:
                sys.stderr.write('bench.py: ranks still running after %.0f s -- terminating the launcher (pid %d) and its %d descendants\n'
                                 % (timeout_s, child.pid, len(seen)))
                child.send_signal(signal.SIGTERM)       # the launcher's handler terminates its workers
                try:
                    child.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    child.kill()
                    child.wait()
                rc = 124
                break
            time.sleep(0.25)
    except KeyboardInterrupt:
        seen.update(_descendants(child.pid))
        child.send_signal(signal.SIGTERM)
        try:
            child.wait(timeout=20)
        except subprocess.TimeoutExpired:
            child.kill()
        rc = 130
    finally:
        time.sleep(0.2)
        sweep()
    return rc


class BoardSampler:
    """Board power and shader clock while the timed steps and the whole trajectory run: a thread of THIS
    process starts `rocm-smi --showpower --showmaxpower --showclocks --json` as a fresh child about twice a second (a subprocess of a process
    that holds the GPU is fine; nothing is ever exec'ed in place) and keeps (time, sclk MHz, package W).  `window(t0, t1)`
    summarises the samples taken between two perf_counter readings.  A box without rocm-smi, or one whose output does not
    parse, gives null fields -- never an error."""

    CMD = ['rocm-smi', '--showpower', '--showmaxpower', '--showclocks', '--json']

    def __init__(self, period_s=0.5):
        import threading
        self.period, self.rows, self.limit, self.err = period_s, [], None, None
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._loop, daemon=True)

    @staticmethod
    def parse(txt):
        """(sclk MHz, package W, package limit W) of the first card in rocm-smi's JSON; None where a field is absent."""
        import re as _re
        c = next(iter(json.loads(txt).values()))
        sclk = _re.sub('[^0-9]', '', str(c.get('sclk clock speed:', '')))
        pw = c.get('Current Socket Graphics Package Power (W)', c.get('Average Graphics Package Power (W)'))
        lim = c.get('Max Graphics Package Power (W)')
        return (int(sclk) if sclk else None, float(pw) if pw not in (None, 'N/A') else None,
                float(lim) if lim not in (None, 'N/A') else None)

    def _once(self):
        r = subprocess.run(self.CMD, capture_output=True, text=True, timeout=10)
        return self.parse(r.stdout[r.stdout.index('{'):])

    def _loop(self):
        while not self._stop.is_set():
            t = time.perf_counter()
            try:
                sclk, pw, lim = self._once()
                self.rows.append((0.5 * (t + time.perf_counter()), sclk, pw))
                if lim is not None:
                    self.limit = lim
            except Exception as e:      # a sample that fails is a gap; a box without rocm-smi yields nulls
                self.err = repr(e)[:120]
                if isinstance(e, FileNotFoundError):
                    return
            self._stop.wait(self.period)

    def start(self):
        self._th.start()
        return self

    def stop(self):
        self._stop.set()
        self._th.join(timeout=15)

    def window(self, t0, t1):
        rows = [r for r in self.rows if t0 <= r[0] <= t1]
        pw = [r[2] for r in rows if r[2] is not None]
        ck = [r[1] for r in rows if r[1] is not None]
        return dict(samples=len(rows),
                    board_power_w=dict(mean=round(sum(pw) / len(pw), 1), max=round(max(pw), 1)) if pw else None,
                    sclk_mhz=dict(mean=round(sum(ck) / len(ck)), min=min(ck)) if ck else None)

    def block(self, windows):
        """The JSON block: one summary per named (t0, t1) window + the union of them as the top-level fields."""
        out = {k: self.window(*w) for k, w in windows.items() if w is not None}
        lo, hi = min(w[0] for w in windows.values() if w), max(w[1] for w in windows.values() if w)
        allw = [r for k, w in windows.items() if w for r in self.rows if w[0] <= r[0] <= w[1]]
        pw = [r[2] for r in allw if r[2] is not None]
        ck = [r[1] for r in allw if r[1] is not None]
        return dict(board_power_w=dict(mean=round(sum(pw) / len(pw), 1), max=round(max(pw), 1)) if pw else None,
                    sclk_mhz=dict(mean=round(sum(ck) / len(ck)), min=min(ck)) if ck else None,
                    package_limit_w=self.limit, samples=len(allw), period_s=self.period, windows=out,
                    source='rocm-smi --showpower --showmaxpower --showclocks --json, a fresh child about every %.1f s from a thread of rank 0 while the '
                           'timed steps and the whole trajectory run (sclk = the PLL reading at the sample instant, power = socket '
                           'package power)' % self.period,
                    error=self.err if not allw else None)


class NativeRunner:
    """The hot path through the C ABI: dlpm_sampler_begin / _steps / _copy_state on this rank's GPU."""

    def __init__(self, args, cfg_name, B, T, alpha, rank, dev):
        import dlpm_amd
        from dlpm_amd import _lib
        self.lib, self.L, self.dev = _lib, _lib.lib(), dev
        p = dlpm_amd.load_config(cfg_name)
        torch.manual_seed(1234)
        self.net = dlpm_amd.rerandomize_(dlpm_amd.init_model_by_parameter(p), 4321)
        # the per-GPU batch of the BASELINE configuration is a declared property of the workload (the same on every rank and
        # for every chunk), so the dispatch policy may weigh grid occupancy for it (dlpm_unet_set_conv_policy)
        self.net.set_conv_policy(args.conv, args.dispatch_batch if args.dispatch_batch >= 0 else (args.batch or WORKLOADS[args.workload][1]))
        self.net.set_gemm_policy(args.gemm)
        self.shape = [B, p['data']['channels'], p['data']['image_size'], p['data']['image_size']]
        ev = p['eval']['dlpm']
        self.meth = dlpm_amd.GenerativeLevyProcess(alpha, str(dev), T, rescale_timesteps=True, seed=0, sample_offset=rank * B,
                                                   use_graph=not args.no_graph, isotropic=not args.non_iso, LIM=args.lim)
        self.st = _lib.stream_ptr()
        flags = (_lib.SMP_LIM if args.lim else 0) | (_lib.UPD_CLIP if args.clip else 0) | (_lib.UPD_DLIM if args.deterministic else 0)
        self.h = self.meth._native_sampler(self.net, self.shape, flags, float(ev.get('dlim_eta', 0.0)) if args.deterministic else 0.0,
                                           ev['clamp_a'], ev['clamp_eps'], 0)
        self.flops_per_sample = self.net.flops_per_sample(self.shape[2])

    def begin(self):
        self.lib.check(self.L.dlpm_sampler_begin(self.h, self.st))

    def steps(self, n):
        self.lib.check(self.L.dlpm_sampler_steps(self.h, n, self.st))

    def state(self):
        x = torch.empty(self.shape, device=self.dev)
        self.lib.check(self.L.dlpm_sampler_copy_state(self.h, x.data_ptr(), self.st))
        return x

    def sync(self):
        torch.cuda.synchronize()

    def profile(self, nprof):
        self.lib.check(self.L.dlpm_prof_enable(1))
        self.steps(nprof)
        buf = C.create_string_buffer(1 << 16)
        self.lib.check(self.L.dlpm_prof_report(buf, len(buf)))
        self.lib.check(self.L.dlpm_prof_enable(0))
        return parse_prof(buf.value.decode())


class DryRunner:
    """DLPM_BENCH_DRY_RUN=1 (tests only, never a measurement): the same launch / rendezvous / barrier / gather / JSON
    control flow on CPU tensors, with the sampler replaced by a stub, so the N-rank path is exercised without a GPU."""

    def __init__(self, args, cfg_name, B, T, alpha, rank, dev):
        self.shape, self.rank, self.flops_per_sample, self.t = [B, 3, 4, 4], rank, 1.0, 0

    def begin(self):
        self.t = 0

    def steps(self, n):
        # test hooks: a rank that dies / hangs in the middle of the trajectory
        die = os.environ.get('DLPM_BENCH_DIE_RANK')
        if die is not None and int(die) == self.rank and self.t + n > int(os.environ.get('DLPM_BENCH_DIE_AT', '0')):
            if os.environ.get('DLPM_BENCH_DIE_MODE') == 'hang':
                time.sleep(3600)
            os._exit(17)
        self.t += n
        time.sleep(1e-4 * n)

    def state(self):
        B = self.shape[0]
        idx = torch.arange(self.rank * B, (self.rank + 1) * B, dtype=torch.float32)
        return idx.view(-1, 1, 1, 1) * torch.ones(self.shape) + self.t

    def sync(self):
        pass

    def profile(self, nprof):
        return {}


def roofline_block(prof, nprof, cfg_name, B, workload):
    wino4, wino = prof.get('conv3x3_wino4'), prof.get('conv3x3_wino')
    c = wino4 or wino or prof.get('conv3x3_halo') or prof.get('conv3x3_igemm')
    if not c:
        return None
    alg = c['flops'] / (c['ms'] * 1e-3) / 1e12
    if wino4:
        kname, executed = 'k_conv3x3_wino4', 36.0 / 144.0
        kdesc = ('k_conv3x3_wino4 (3x3 stride-1 conv as Winograd F(4x4,3x3) on the fp32 MFMA 16x16x4: 36 instead of 144 '
                 'multiplies per 4x4 output tile and channel pair; fused GN+SiLU staging, in-register output transform, '
                 'bias/residual/GN-stats epilogue)')
    elif wino:
        kname, executed = 'k_conv3x3_wino_q', 16.0 / 36.0
        kdesc = ('k_conv3x3_wino_q (3x3 stride-1 conv as Winograd F(2x2,3x3) on the fp32 MFMA 32x32x2: 16 instead of 36 '
                 'multiplies per 2x2 output tile and channel pair; fused GN+SiLU staging, in-register output transform, '
                 'bias/residual/GN-stats epilogue)')
    else:
        kname, executed = 'k_conv3x3_halo_ws', 1.0
        kdesc = ('k_conv3x3_halo_ws<128,2,2,2,2,2> (3x3 stride-1 conv, fp32 MFMA 32x32x2, LDS halo tile + register-streamed '
                 'weights, fused GN+SiLU/bias/residual/GN-stats)')
    ach = alg * executed                      # FLOP/s actually executed on the matrix pipe
    traffic, tnote = pmc_traffic(kname, workload)
    mfma = [v for k, v in prof.items() if k.startswith('conv') and 'stem' not in k and 'direct' not in k and 'head' not in k]
    return dict(kernel=kdesc, bound='mfma', achieved=round(ach, 3), peak=PEAK_FP32_MFMA_TFLOPS, unit='TFLOP/s',
                frac=round(ach / PEAK_FP32_MFMA_TFLOPS, 4), traffic=traffic, traffic_note=tnote,
                algorithmic_bytes_per_launch=c['bytes'] / c['launches'],
                note=('achieved = FLOP/s EXECUTED on the fp32 matrix pipe = algorithmic (direct-convolution, 2*9*Cin*Cout per '
                      'output pixel) FLOP/s x %.4f, the share of those multiplies this kernel executes; frac = achieved / the '
                      'dense fp32 MFMA peak' % executed),
                algorithmic_tflops=round(alg, 3), executed_share_of_algorithmic_flops=round(executed, 4),
                launches_per_step=c['launches'] // nprof, avg_launch_ms=round(c['ms'] / c['launches'], 5),
                algorithmic_flops_per_launch_avg=c['flops'] / c['launches'], share_of_step_ms=round(c['ms'] / nprof, 3),
                all_mfma_conv_classes_algorithmic_tflops=round(sum(v['flops'] for v in mfma) / (sum(v['ms'] for v in mfma) * 1e-3) / 1e12, 3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--workload', default='cifar10_unet_b1024_T1000', choices=list(WORKLOADS))
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default: the workload\'s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-threads', type=int, default=0, help='torch threads of the cpu_baseline leg (default: the fastest of a sweep over 8..128 <= physical cores; 32 kept unless another count is > 10 %% faster)')
    ap.add_argument('--cpu-baseline-only', action='store_true', help='print only the cpu_baseline block (no GPU work)')
    ap.add_argument('--no-prof', action='store_true', help='skip the instrumented eager pass (roofline = null)')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-full-trajectory', action='store_true',
                    help='do not time a whole T-step trajectory; value = init + (T-1) * ms_per_step + gather')
    ap.add_argument('--conv', default='auto', choices=['auto', 'f4', 'f2', 'igemm'], help='convolution generation (A/B runs)')
    ap.add_argument('--gemm', default='auto', choices=['auto', 'f32', 'bf16x3'],
                    help='matrix pipe of the 1x1 and stride-2 convolutions (dlpm_unet_set_gemm_policy): bf16x3 = fp32 operands split '
                         'exactly into three bf16 planes, six partial products, fp32 accumulate (default where the shape admits it); '
                         'f32 = the fp32 MFMA everywhere (A/B runs)')
    ap.add_argument('--dispatch-batch', type=int, default=-1,
                    help='dlpm_unet_set_conv_policy dispatch batch (default: --batch if given, else the per-GPU batch of the workload\'s BASELINE config; 0: geometry only)')
    ap.add_argument('--non-iso', action='store_true',
                    help='non-isotropic noise variant (--non_iso of the reference): [T,B,D] tables; not the headline config')
    ap.add_argument('--rank-timeout', type=float, default=7200.0,
                    help='N > 1 started plainly: seconds after which the parent ends its ranks (their own process group) and exits 124')
    ap.add_argument('--clip', action='store_true',
                    help='clip_denoised variant (--clip of the reference): x0 predicted, clamped, eps recomputed; the update runs in '
                         'k_update_rows behind the head convolution instead of inside k_head_fused; not the headline config')
    ap.add_argument('--deterministic', action='store_true',
                    help='DLIM variant (--deterministic of the reference, the config\'s dlim_eta); update in k_update_rows; not the headline config')
    ap.add_argument('--no-board-sampler', action='store_true', help='do not sample rocm-smi during the run (board = null)')
    ap.add_argument('--lim', action='store_true',
                    help='LIM sampler variant (--method lim of the reference, SDE updates): T network evaluations; '
                         'not the headline config')
    args = ap.parse_args()

    if args.cpu_baseline_only:
        cfg_name, _, T, alpha = WORKLOADS[args.workload]
        print(json.dumps(cpu_baseline(cfg_name, T, alpha, threads=args.cpu_threads or None)), flush=True)
        return
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], args.rank_timeout))

    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert world == args.gpus, 'WORLD_SIZE=%d but --gpus %d' % (world, args.gpus)
    # test hooks (never used by the driver): DLPM_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0 (N-rank control flow on a
    # 1-GPU box, with DLPM_BENCH_BACKEND=gloo); DLPM_BENCH_DRY_RUN=1 replaces the sampler by a CPU stub (no measurement)
    dry = os.environ.get('DLPM_BENCH_DRY_RUN') == '1'
    if os.environ.get('DLPM_BENCH_SINGLE_DEVICE') == '1':
        local = 0
    backend = 'gloo' if dry else os.environ.get('DLPM_BENCH_BACKEND', 'nccl')
    if dry:
        dev = torch.device('cpu')
    else:
        torch.cuda.set_device(local)
        dev = torch.device('cuda', local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    from dlpm_amd.dist import all_gather_samples

    def device_identity():
        """What tells this rank's GPU from every other GPU of the node: UUID and PCI address from the driver."""
        if dev.type != 'cuda':
            return dict(device_index=None, device_uuid=None, pci_bus_id=None, device_name='cpu (dry run)')
        pr = torch.cuda.get_device_properties(dev)
        pci = None
        if hasattr(pr, 'pci_bus_id'):
            pci = '%04x:%02x:%02x' % (getattr(pr, 'pci_domain_id', 0), pr.pci_bus_id, getattr(pr, 'pci_device_id', 0))
        uuid = getattr(pr, 'uuid', None)
        return dict(device_index=dev.index, device_uuid=None if uuid is None else str(uuid), pci_bus_id=pci, device_name=pr.name)

    cfg_name, B, T, alpha = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    K, W = args.steps, args.warmup
    nsteps = T if args.lim else T - 1        # network evaluations of one sample() call
    assert 1 <= K and W + K <= nsteps, 'at most %d steps exist' % nsteps
    run = (DryRunner if dry else NativeRunner)(args, cfg_name, B, T, alpha, rank, dev)
    shape = run.shape

    def barrier():
        run.sync()
        if world > 1:
            dist.barrier()
        run.sync()

    def max_over_ranks(*vals):
        t = torch.tensor(vals, dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.tolist()

    board = None
    if rank == 0 and not args.no_board_sampler:
        board = BoardSampler().start()
    # ---- init (A, tables, x_T)
    barrier()
    t0 = time.perf_counter()
    run.begin()
    run.sync()
    init_s = time.perf_counter() - t0
    # ---- warm-up (the first step runs eagerly, the second is captured into the graph)
    run.steps(W)
    # ---- timed region: exactly K steps
    barrier()
    t0 = time.perf_counter()
    run.steps(K)
    barrier()
    dt = time.perf_counter() - t0
    win_steps, win_traj = (t0, t0 + dt), None
    own_ms_per_step = dt / K * 1e3          # this rank's clock (the line's ms_per_step is the max over ranks)
    dt, init_s = max_over_ranks(dt, init_s)
    ms_per_step = dt / K * 1e3
    # ---- final gather of the finished samples: ONE RCCL all-gather (nothing to gather on one GPU)
    x = run.state()
    gather_s = None
    if world > 1:
        barrier()
        t0 = time.perf_counter()
        full = all_gather_samples(x, B * world)
        barrier()
        gather_s = max_over_ranks(time.perf_counter() - t0)[0]
    else:
        full = x
    finite = bool(torch.isfinite(full).all().item())
    assert full.shape[0] == B * world
    # ---- who ran: one record per rank (rank, device index, UUID / PCI address, its own step time, a checksum of ITS shard and of
    # the same rows of the gathered batch), collected with ONE all_gather_object -- so that the line proves N distinct devices and
    # that the collective of the stated backend delivered every rank's shard
    me = dict(rank=rank, local_rank=local, pid=os.getpid(), ms_per_step=round(own_ms_per_step, 4), **device_identity())
    me['shard_checksum'] = float(x.double().sum().item())
    me['gathered_rows_checksum'] = float(full[rank * B:(rank + 1) * B].double().sum().item())
    if world > 1:
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, me)
        dist_info = dict(backend=dist.get_backend(), world_size=dist.get_world_size(),
                         collective='all_gather_into_tensor (dlpm_amd.dist.all_gather_samples)',
                         allgather_bytes=int(full.numel() * full.element_size()),
                         allgather_bytes_per_rank_sent=int(x.numel() * x.element_size()))
        # every rank holds the same gathered batch: each rank's shard checksum must reappear in EVERY rank's copy of those rows
        mine = [float(full[r * B:(r + 1) * B].double().sum().item()) for r in range(world)]
        shard_sums = [ri['shard_checksum'] for ri in ranks_info]
        # NaN-aware: a shard that holds a NaN (samples_finite = false is REPORTED, it must not void the line) has a NaN checksum on
        # both sides, and NaN != NaN -- such a pair counts as equal; only a finite mismatch is a broken gather
        same = [(a == b) or (math.isnan(a) and math.isnan(b)) for a, b in zip(mine, shard_sums)]
        dist_info['gather_verified'] = bool(all(same))
        # per pair, unconditionally: one NaN shard must not switch the check off for a finite mismatch elsewhere
        assert dist_info['gather_verified'], 'the gathered batch does not hold the shards the ranks produced: %s vs %s' % (mine, shard_sums)
        ids = [ri['device_uuid'] or ri['pci_bus_id'] for ri in ranks_info]
        dist_info['distinct_devices'] = len(set(ids)) if all(i is not None for i in ids) else None
        if backend == 'nccl' and os.environ.get('DLPM_BENCH_SINGLE_DEVICE') != '1' and dist_info['distinct_devices'] is not None:
            assert dist_info['distinct_devices'] == world, 'RCCL ranks share a device: %s' % ids
    else:
        ranks_info, dist_info = [me], dict(backend=None, world_size=1, collective=None, allgather_bytes=0,
                                           allgather_bytes_per_rank_sent=0, gather_verified=None, distinct_devices=1)

    # ---- one WHOLE trajectory, timed end to end: init + every reverse step + gather
    full_s = None
    if K == nsteps and W == 0:
        full_s = init_s + dt + (gather_s or 0.0)
    elif not args.no_full_trajectory:
        barrier()
        t0 = time.perf_counter()
        run.begin()
        run.steps(nsteps)
        x = run.state()
        full = all_gather_samples(x, B * world) if world > 1 else x
        barrier()
        win_traj = (t0, time.perf_counter())
        full_s = max_over_ranks(time.perf_counter() - t0)[0]
        finite = finite and bool(torch.isfinite(full).all().item())
    if full_s is not None:
        value, value_source = B * world / full_s, 'measured: one whole trajectory (init + %d reverse steps%s) timed between barriers' % (
            nsteps, ' + all-gather' if world > 1 else '')
    else:
        total_s = init_s + nsteps * ms_per_step / 1e3 + (gather_s or 0.0)
        value, value_source = B * world / total_s, 'extrapolated: init + %d x ms_per_step (%d timed steps)%s' % (
            nsteps, K, ' + all-gather' if world > 1 else '')

    board_block = None
    if board is not None:
        board.stop()
        board_block = board.block(dict(timed_steps=win_steps, whole_trajectory=win_traj))
    roofline, upd, breakdown, att, split = None, None, None, None, None
    upd_unfused = None
    if not args.no_prof and rank == 0 and not dry:
        # instrumented eager pass on the same stream: HIP events around every launch, by kernel class
        nprof = 3
        run.begin()
        run.steps(2)
        prof = run.profile(nprof)
        breakdown = {k: round(v['ms'] / nprof, 4) for k, v in prof.items()}
        roofline = roofline_block(prof, nprof, cfg_name, B, args.workload if not args.batch else '%s@B%d' % (args.workload, B))
        at = prof.get('attention')
        if at and at['ms'] > 0:
            # QKV attention reads qkv (3C) and writes out (C) per token: 16 T C bytes for 4 T^2 C flops per sample, i.e.
            # T/4 FLOP/B -- below the ridge (157.3 TFLOP/s / 8 TB/s = 19.7 FLOP/B) for T <= 64: HBM-bound there
            tf = at['flops'] / (at['ms'] * 1e-3) / 1e12
            gbs = at['bytes'] / (at['ms'] * 1e-3) / 1e9
            hbm_bound = at['flops'] / at['bytes'] < PEAK_FP32_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
            att = dict(kernel='k_attention<64,NT> (QK^T / softmax / AV on the fp32 MFMA 16x16x4, K and V of a head in LDS, one workgroup per (sample, head, 64 queries))',
                       bound='hbm' if hbm_bound else 'mfma',
                       achieved=round(gbs, 1) if hbm_bound else round(tf, 2),
                       peak=PEAK_HBM_GBS if hbm_bound else PEAK_FP32_MFMA_TFLOPS, unit='GB/s' if hbm_bound else 'TFLOP/s',
                       frac=round(gbs / PEAK_HBM_GBS, 4) if hbm_bound else round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
                       tflops=round(tf, 2), gbytes_per_s=round(gbs, 1), flop_per_byte=round(at['flops'] / at['bytes'], 2),
                       launches_per_step=at['launches'] // nprof, avg_launch_ms=round(at['ms'] / at['launches'], 5),
                       note='algorithmic bytes = qkv read + out write; 0.2 % of the step FLOPs')
        sp = [v for k, v in prof.items() if 'bf16x3' in k]
        if sp:
            fl, ms_ = sum(v['flops'] for v in sp), sum(v['ms'] for v in sp)
            eq = fl / (ms_ * 1e-3) / 1e12
            split = dict(kernel='k_conv_split (1x1 and stride-2 3x3 convolutions as an fp32 GEMM on the bf16 matrix pipe: 3 bf16 planes per '
                                'operand, 6 of the 9 partial products, v_mfma_f32_32x32x16_bf16, fp32 accumulate)', bound='mfma',
                         achieved=round(6 * eq, 1), peak=PEAK_BF16_MFMA_TFLOPS, unit='TFLOP/s', frac=round(6 * eq / PEAK_BF16_MFMA_TFLOPS, 4),
                         fp32_equivalent_tflops=round(eq, 2), launches_per_step=sum(v['launches'] for v in sp) // nprof,
                         share_of_step_ms=round(ms_ / nprof, 3),
                         note='achieved = bf16 FLOP/s EXECUTED (6 x the fp32-equivalent rate) over the dense bf16 MFMA peak; the same '
                              'launches on the fp32 MFMA (--gemm f32) run at 85-115 fp32 TFLOP/s')
        u, hu, hg, hf = prof.get('update'), prof.get('conv3x3_head+update'), prof.get('head_gather+update'), prof.get('head_fused+update')
        if hf:   # round 4: head convolution + reverse update as ONE pass over HBM (head_fused.hip)
            n_el = B * shape[1] * shape[2] * shape[3]
            moved = hf['bytes'] / hf['launches']                       # head input once + read x + write x
            alg = moved + 4.0 * n_el                                   # + eps (12 B per state element), which never reaches HBM here
            sec = hf['ms'] / hf['launches'] * 1e-3
            gbs = alg / sec / 1e9
            traffic, tnote = pmc_traffic('k_head_fused', args.workload if not args.batch else '%s@B%d' % (args.workload, B))
            upd = dict(kernel='k_head_fused: the head convolution (GroupNorm + SiLU + 3x3 conv to the image channels, as a GEMM onto the 9 Cout tap '
                              'channels -- bf16 matrix pipe, exact three-plane split, fp32 result -- whose result stays in an LDS ring) and the '
                              'x_{t-1} update (Philox noise) in ONE launch of persistent workgroups',
                       bound='hbm', achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit='GB/s', frac=round(gbs / PEAK_HBM_GBS, 4),
                       traffic=traffic, traffic_note=tnote, bytes_per_launch=alg, bytes_moved_per_launch=moved,
                       achieved_on_moved_bytes=round(moved / sec / 1e9, 1), avg_launch_ms=round(sec * 1e3, 5),
                       note='algorithmic bytes = the head\'s input once + 12 B per state element (read x, eps, write x: SURVEY 8d); eps never '
                            'reaches HBM in this kernel, so the bytes it moves are 4 B per element fewer (achieved_on_moved_bytes); '
                            'north-star target for the fused update: 0.60 of HBM')
        elif hg:   # round 3 path: head convolution = 1x1 GEMM onto 9 Cout tap channels (its own launch, class conv1x1_igemm) + this kernel
            gbs = hg['bytes'] / (hg['ms'] * 1e-3) / 1e9
            traffic, tnote = pmc_traffic('k_head_gather', args.workload if not args.batch else '%s@B%d' % (args.workload, B))
            upd = dict(kernel='k_head_gather with the x_{t-1} update (Philox noise): reads the 9 Cout partial products per pixel the head GEMM '
                              'left, sums the nine taps, applies the reverse step to the NCHW state in place -- eps never reaches HBM',
                       bound='hbm', achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit='GB/s', frac=round(gbs / PEAK_HBM_GBS, 4),
                       traffic=traffic, traffic_note=tnote, bytes_per_launch=hg['bytes'] / hg['launches'],
                       avg_launch_ms=round(hg['ms'] / hg['launches'], 5),
                       note='algorithmic bytes = 4 * 9 Cout per pixel of tap partial products + read x + write x (12 B/element of state); '
                            'north-star target for this kernel: 0.60 of HBM')
        elif hu:   # nets whose head the GEMM + gather form does not take: the VALU head kernel applies the update in its epilogue
            gbs = hu['bytes'] / (hu['ms'] * 1e-3) / 1e9
            traffic, tnote = pmc_traffic('k_conv3x3_head', args.workload if not args.batch else '%s@B%d' % (args.workload, B))
            upd = dict(kernel='k_conv3x3_head with the x_{t-1} update (Philox noise) fused into its epilogue: eps never reaches HBM, no '
                              'separate update launch', bound='hbm', achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit='GB/s',
                       frac=round(gbs / PEAK_HBM_GBS, 4), traffic=traffic, traffic_note=tnote, bytes_per_launch=hu['bytes'] / hu['launches'],
                       avg_launch_ms=round(hu['ms'] / hu['launches'], 5),
                       note='algorithmic bytes = the convolution input once + read x + write x; the launch is VALU-bound (36 FMAs and '
                            'one SiLU per input element), not bandwidth-bound: see DESIGN.md section 3')
        elif u:
            gbs = u['bytes'] / (u['ms'] * 1e-3) / 1e9
            upd = dict(kernel='k_update_rows (fused x_{t-1} update, Philox noise)', bound='hbm', achieved=round(gbs, 1),
                       peak=PEAK_HBM_GBS, unit='GB/s', frac=round(gbs / PEAK_HBM_GBS, 4), traffic=None,
                       bytes_per_launch=u['bytes'] / u['launches'], avg_launch_ms=round(u['ms'] / u['launches'], 5),
                       note='algorithmic bytes = read x + read eps + write x (12 B/element, in-kernel Philox); in-loop launch '
                            'time from HIP events (the state was last touched a whole UNet forward earlier: HBM, not cache)')

        # the standalone update launch (dlpm_update_f32: k_update_rows, or k_update<VEC> under the clip / DLIM flags): every variant that cannot take the fused head -- --clip, --deterministic, LIM,
        # START_X / Z / PREVIOUS_X -- and the path all bounded parity fixtures run
        uu = prof.get('update') or prof.get('lim_update')
        if uu and ((hf or hu) is None or args.clip or args.deterministic or args.lim):
            gbs = uu['bytes'] / (uu['ms'] * 1e-3) / 1e9
            upd_unfused = dict(kernel='k_update_lim' if args.lim else ('%s (x_{t-1} update as its own launch behind the head convolution, Philox noise%s)'
                                      % ('k_update<VEC>' if (args.clip or args.deterministic) else 'k_update_rows',
                                         ', clip_denoised: x0 predicted, clamped, eps recomputed' if args.clip else ', DLIM' if args.deterministic else '')),
                               bound='hbm', achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit='GB/s', frac=round(gbs / PEAK_HBM_GBS, 4),
                               traffic=None, bytes_per_launch=uu['bytes'] / uu['launches'], avg_launch_ms=round(uu['ms'] / uu['launches'], 5),
                               note='algorithmic bytes = read x + read eps + write x (12 B/element, in-kernel Philox); in-loop launch time from '
                                    'HIP events of the instrumented eager pass (eps was written by the head convolution just before: partly L2 / MALL)')

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not dry:
        cpu = cpu_baseline(cfg_name, T, alpha, threads=args.cpu_threads or None)

    if rank == 0:
        step_tflops = run.flops_per_sample * B * world / (ms_per_step * 1e-3) / 1e12
        out = {
            'metric': METRIC[cfg_name], 'value': round(value, 4),
            'unit': 'samples/s', 'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': round(ms_per_step, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.gemm == 'f32' or cfg_name == 'mnist' else 'f32 (1x1 / stride-2 convolutions: fp32 operands split exactly into 3 bf16 planes, bf16 MFMA products, fp32 accumulate)',
            'data': 'synthetic',
            'config': {'workload': args.workload + ('+non_iso' if args.non_iso else '') + ('+lim_sde' if args.lim else '') + ('+clip' if args.clip else '') + ('+dlim' if args.deterministic else ''), 'state_shape_per_gpu': shape, 'global_batch': B * world,
                       'reverse_steps': T, 'alpha': alpha, 'timed_steps': K, 'trajectory_steps': nsteps,
                       'init_ms': round(init_s * 1e3, 3), 'allgather_ms': None if gather_s is None else round(gather_s * 1e3, 3),
                       'net': 'reference cifar10.yml UNet (mc=128, 39.6M params), random init + re-drawn zero tensors'
                       if cfg_name == 'cifar10' else cfg_name,
                       'rng': 'philox (device, keyed by global sample index)', 'hip_graph': not args.no_graph,
                       'conv_generation': args.conv, 'gemm_1x1_and_downsample': args.gemm + (' (= bf16x3: fp32 operands split exactly into 3 bf16 planes, 6 partial products per multiply, fp32 accumulate; fp32-grade error, tests/test_gpu_kernels.py)' if args.gemm == 'auto' else ''), 'conv_dispatch_batch': args.dispatch_batch if args.dispatch_batch >= 0 else (args.batch or WORKLOADS[args.workload][1]),
                       'parallelism': ('batch-sharded x%d, one RCCL all-gather at the end' % world) if world > 1 else 'single GPU (no collective)'},
            'value_source': value_source,
            'full_trajectory_s': None if full_s is None else round(full_s, 4),
            'extrapolated_value': round(B * world / (init_s + nsteps * ms_per_step / 1e3 + (gather_s or 0.0)), 4),
            'gflop_per_sample_step': round(run.flops_per_sample / 1e9, 4),
            'whole_step_tflops': round(step_tflops, 3),
            'algorithmic_tflops_over_fp32_peak': round(step_tflops / (PEAK_FP32_MFMA_TFLOPS * world), 4),
            'algorithmic_tflops_note': 'whole_step_tflops counts the DIRECT-convolution FLOPs of the step (SURVEY 8d); the Winograd kernels execute 36/144 '
                                       '(F(4x4)) or 16/36 (F(2x2)) of those multiplies, so this ratio may exceed 1 -- the fraction of a hardware '
                                       'peak is roofline.frac',
            'samples_finite': finite,
            'ranks': ranks_info, 'backend': dist_info['backend'], 'rccl_world_size': dist_info['world_size'],
            'allgather_bytes': dist_info['allgather_bytes'], 'distributed': dist_info,
            'board_power_w': None if not board_block else board_block['board_power_w'], 'sclk_mhz': None if not board_block else board_block['sclk_mhz'],
            'package_limit_w': None if not board_block else board_block['package_limit_w'], 'board': board_block,
            'roofline': roofline, 'split_gemm_kernel': split, 'update_kernel': upd, 'update_kernel_unfused': upd_unfused, 'attention_kernel': att, 'ms_per_step_by_kernel_class': breakdown, 'cpu_baseline': cpu,
        }
        if dry:
            out['dry_run'] = 'control-flow test only (DLPM_BENCH_DRY_RUN=1): stub sampler on CPU, NOT a measurement'
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
