#!/usr/bin/env python3
"""bench.py -- samples/sec of DLPM's reverse-sampling loop at T=1000 on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]        (N > 1: launched by torch.distributed.run)

Workload (N = 1 and per GPU for N > 1, weak scaling): BASELINE.json configs[2] -- CIFAR-10 shaped
state [1024, 3, 32, 32], the reference's cifar10.yml UNet (39.6 M parameters, attention at 8x8 and
4x4), random init with the zero-initialised tensors re-drawn (dlpm_amd/weights.py), T = 1000,
alpha = 1.7, clamp_a = 10, clamp_eps = 50, fp32, Philox noise keyed by the global sample index.

A "step" is ONE reverse step (UNet forward + fused update) over the whole batch: W warm-up steps,
then exactly K timed steps between barrier + synchronize pairs, max over ranks.  The trajectory has
T-1 = 999 identical steps, so
    value = B_total / (init_s + 999 * ms_per_step / 1000 + allgather_s)        [samples/s at T=1000]
with init (A draws + tables + x_T) and the final RCCL all-gather measured in the same run.  With
--steps 999 --warmup 0 the timed region IS the whole trajectory.
"""
import argparse
import ctypes as C
import json
import os
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
# HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes (separate --pmc runs,
# FETCH_SIZE doubled per the gfx950 guide): profiles/r01/pmc_hbm_traffic_*.txt
PMC_TRAFFIC_BYTES_PER_LAUNCH = {'k_conv3x3_halo_ws<128,2,2,2,2,2>': (0.6663 + 0.2194) * 1e9,  # fetch + write, mean over all three instantiations
                                'k_conv3x3_wino_q': (0.7141 + 0.2238) * 1e9,   # fetch + write, mean over both instantiations (47 launches/step), v14
                                'k_conv3x3_wino4': (1.0377 + 0.3134) * 1e9}    # fetch + write, mean over both instantiations (33 launches/step)
PMC_TRAFFIC_FILE = 'profiles/r01/pmc_hbm_traffic_v18.txt'
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32
PEAK_HBM_GBS = 8000.0


def parse_prof(txt):
    out = {}
    for line in txt.strip().splitlines():
        name, n, ms, fl, by = line.split()
        out[name] = dict(launches=int(n), ms=float(ms), flops=float(fl), bytes=float(by))
    return out


def cpu_baseline(cfg_name, T, alpha, budget_s=20.0):
    """The oracle (a torch-CPU port of the reference loop) on this box's host cores, bounded sample."""
    from oracle import nets, sampler as osampler, process as P
    import dlpm_amd
    p = dlpm_amd.load_config(cfg_name)
    torch.manual_seed(1234)
    net = dlpm_amd.rerandomize_(dlpm_amd.init_model_by_parameter(p), 4321)
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    heads = p['model']['num_heads']
    B, steps = 8, 4
    shape = [B, p['data']['channels'], p['data']['image_size'], p['data']['image_size']]
    g = torch.Generator().manual_seed(0)
    model = lambda x, t: nets.unet_forward(sd, x, t, heads)
    cores = torch.get_num_threads()
    with torch.inference_mode():
        x = torch.randn(shape, generator=g)
        t0 = time.time()
        model(x, torch.full((B,), 0.5))           # warm-up + cost probe
        probe = time.time() - t0
        steps = max(2, min(64, int(budget_s / max(probe, 1e-3))))
        Tshort = steps + 1
        A = torch.rand(Tshort, B, generator=g) + 0.5
        zs = [torch.randn(shape, generator=g) for _ in range(steps)]
        t0 = time.time()
        osampler.sample_with_tables(model, shape, Tshort, alpha, A, x, zs)
        dt = time.time() - t0
    per_step = dt / steps
    return dict(value=B / (per_step * (T - 1)), unit='samples/s at T=1000', cores=cores, kind='port',
                sample='oracle (torch-CPU port of the reference loop), same UNet, B=%d, %d reverse steps timed '
                       '(%.1f s), extrapolated linearly to 999 steps' % (B, steps, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--workload', default='cifar10_unet_b1024_T1000', choices=list(WORKLOADS))
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default: the workload\'s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-prof', action='store_true', help='skip the instrumented eager pass (roofline = null)')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--non-iso', action='store_true',
                    help='non-isotropic noise variant (--non_iso of the reference): [T,B,D] tables; not the headline config')
    ap.add_argument('--lim', action='store_true',
                    help='LIM sampler variant (--method lim of the reference, SDE updates): T network evaluations; '
                         'not the headline config')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert world == args.gpus, 'WORLD_SIZE=%d but --gpus %d (use torch.distributed.run for N > 1)' % (world, args.gpus)
    # test hooks (not used by the driver): run the N-rank control flow on a 1-GPU box -- every rank on cuda:0 with gloo
    if os.environ.get('DLPM_BENCH_SINGLE_DEVICE') == '1':
        local = 0
    backend = os.environ.get('DLPM_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import dlpm_amd
    from dlpm_amd import _lib
    from dlpm_amd.dist import all_gather_samples
    L = _lib.lib()

    cfg_name, B, T, alpha = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    K, W = args.steps, args.warmup
    nsteps = T if args.lim else T - 1        # network evaluations of one sample() call
    assert 1 <= K and W + K <= nsteps, 'at most %d steps exist' % nsteps
    p = dlpm_amd.load_config(cfg_name)
    torch.manual_seed(1234)
    net = dlpm_amd.rerandomize_(dlpm_amd.init_model_by_parameter(p), 4321)
    shape = [B, p['data']['channels'], p['data']['image_size'], p['data']['image_size']]
    ev = p['eval']['dlpm']
    meth = dlpm_amd.GenerativeLevyProcess(alpha, str(dev), T, rescale_timesteps=True, seed=0, sample_offset=rank * B,
                                          use_graph=not args.no_graph, isotropic=not args.non_iso, LIM=args.lim)
    st = _lib.stream_ptr()
    h = meth._native_sampler(net, shape, _lib.SMP_LIM if args.lim else 0, 0.0, ev['clamp_a'], ev['clamp_eps'], 0)
    flops_per_sample = net.flops_per_sample(shape[2])

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- init (A, tables, x_T)
    barrier()
    t0 = time.perf_counter()
    _lib.check(L.dlpm_sampler_begin(h, st))
    torch.cuda.synchronize()
    init_s = time.perf_counter() - t0
    # ---- warm-up (the first step runs eagerly, the second is captured into the graph)
    _lib.check(L.dlpm_sampler_steps(h, W, st))
    # ---- timed region: exactly K steps
    barrier()
    t0 = time.perf_counter()
    _lib.check(L.dlpm_sampler_steps(h, K, st))
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt, init_s], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt, init_s = tmax.tolist()
    ms_per_step = dt / K * 1e3
    # ---- final gather of the finished samples (single RCCL all-gather)
    x = torch.empty(shape, device=dev)
    _lib.check(L.dlpm_sampler_copy_state(h, x.data_ptr(), st))
    barrier()
    t0 = time.perf_counter()
    full = all_gather_samples(x, B * world)
    barrier()
    gather_s = time.perf_counter() - t0
    finite = bool(torch.isfinite(full).all().item())

    total_s = init_s + nsteps * ms_per_step / 1e3 + gather_s
    value = B * world / total_s

    roofline, upd, breakdown, att = None, None, None, None
    if not args.no_prof and rank == 0:
        # instrumented eager pass on the same stream: HIP events around every launch, by kernel class
        _lib.check(L.dlpm_prof_enable(1))
        nprof = 3
        _lib.check(L.dlpm_sampler_steps(h, nprof, st))
        buf = C.create_string_buffer(1 << 16)
        _lib.check(L.dlpm_prof_report(buf, len(buf)))
        _lib.check(L.dlpm_prof_enable(0))
        prof = parse_prof(buf.value.decode())
        breakdown = {k: round(v['ms'] / nprof, 4) for k, v in prof.items()}
        wino4, wino = prof.get('conv3x3_wino4'), prof.get('conv3x3_wino')
        c = wino4 or wino or prof.get('conv3x3_halo') or prof.get('conv3x3_igemm')
        if c:
            ach = c['flops'] / (c['ms'] * 1e-3) / 1e12
            executed = None
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
                kname = 'k_conv3x3_halo_ws<128,2,2,2,2,2>'
                kdesc = ('k_conv3x3_halo_ws<128,2,2,2,2,2> (3x3 stride-1 conv, fp32 MFMA 32x32x2, LDS halo tile + register-streamed '
                         'weights, fused GN+SiLU/bias/residual/GN-stats)')
            roofline = dict(kernel=kdesc,
                            bound='mfma', achieved=round(ach, 3), peak=PEAK_FP32_MFMA_TFLOPS, unit='TFLOP/s',
                            frac=round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                            traffic=(PMC_TRAFFIC_BYTES_PER_LAUNCH.get(kname) if cfg_name == 'cifar10' and B == 1024 else None),
                            traffic_note='HBM bytes per launch (mean over the launches of a step) from committed rocprofv3 PMC passes, %s; algorithmic bytes per launch = %.4g' % (PMC_TRAFFIC_FILE, c['bytes'] / c['launches']),
                            launches_per_step=c['launches'] // nprof, avg_launch_ms=round(c['ms'] / c['launches'], 5),
                            flops_per_launch_avg=c['flops'] / c['launches'],
                            share_of_step_ms=round(c['ms'] / nprof, 3),
                            all_mfma_conv_classes_tflops=round(sum(v['flops'] for k, v in prof.items() if k.startswith('conv') and 'stem' not in k and 'direct' not in k)
                                                               / (sum(v['ms'] for k, v in prof.items() if k.startswith('conv') and 'stem' not in k and 'direct' not in k) * 1e-3) / 1e12, 3))
            if executed:
                # `achieved` counts ALGORITHMIC flops (2*9*Cin*Cout per output pixel, the direct-convolution count of
                # SURVEY 8d); a Winograd kernel executes a fraction of them on the matrix pipe (36/144 for F(4x4,3x3),
                # 16/36 for F(2x2,3x3)), so `frac` can exceed 1: the MFMA pipe's own utilisation is reported next to it
                roofline['mfma_executed_tflops'] = round(ach * executed, 3)
                roofline['mfma_utilisation'] = round(ach * executed / PEAK_FP32_MFMA_TFLOPS, 4)
                roofline['note'] = ('achieved = algorithmic (direct-convolution) FLOP/s; the Winograd kernel executes %.4f of them on '
                                    'the matrix pipe, mfma_utilisation = executed MFMA FLOP/s over the fp32 MFMA peak' % executed)
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
        u = prof.get('update')
        if u:
            gbs = u['bytes'] / (u['ms'] * 1e-3) / 1e9
            upd = dict(kernel='k_update_rows (fused x_{t-1} update, Philox noise)', bound='hbm', achieved=round(gbs, 1),
                       peak=PEAK_HBM_GBS, unit='GB/s', frac=round(gbs / PEAK_HBM_GBS, 4), traffic=None,
                       bytes_per_launch=u['bytes'] / u['launches'], avg_launch_ms=round(u['ms'] / u['launches'], 5))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg_name, T, alpha)

    if rank == 0:
        step_tflops = flops_per_sample * B * world / (ms_per_step * 1e-3) / 1e12
        out = {
            'metric': METRIC[cfg_name], 'value': round(value, 4),
            'unit': 'samples/s', 'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': round(ms_per_step, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': args.workload + ('+non_iso' if args.non_iso else '') + ('+lim_sde' if args.lim else ''), 'state_shape_per_gpu': shape, 'global_batch': B * world,
                       'reverse_steps': T, 'alpha': alpha, 'timed_steps': K, 'trajectory_steps': nsteps,
                       'init_ms': round(init_s * 1e3, 3), 'allgather_ms': round(gather_s * 1e3, 3),
                       'net': 'reference cifar10.yml UNet (mc=128, 39.6M params), random init + re-drawn zero tensors'
                       if cfg_name == 'cifar10' else cfg_name,
                       'rng': 'philox (device, keyed by global sample index)', 'hip_graph': not args.no_graph,
                       'parallelism': 'batch-sharded x%d, one RCCL all-gather at the end' % world},
            'gflop_per_sample_step': round(flops_per_sample / 1e9, 4),
            'whole_step_tflops': round(step_tflops, 3),
            'whole_step_frac_of_fp32_peak': round(step_tflops / (PEAK_FP32_MFMA_TFLOPS * world), 4),
            'samples_finite': finite,
            'roofline': roofline, 'update_kernel': upd, 'attention_kernel': att, 'ms_per_step_by_kernel_class': breakdown, 'cpu_baseline': cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
