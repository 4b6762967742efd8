#!/usr/bin/env python3
"""Image-dump leg on one MI355X: quantisation kernel, native PNG writer, and the chunk loop
with and without overlap, beside the reference's way (fp32 D2H, then one PIL save per sample on the host).

    python tools/bench_dump.py [--G 512] [--batch 64] [--steps 20]
"""
import argparse
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dlpm_amd
from dlpm_amd import _lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--G', type=int, default=512)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--steps', type=int, default=20)
    a = ap.parse_args()
    L = _lib.lib()
    dev = 'cuda'
    # 1. quantisation kernel
    for shape in [(1024, 3, 32, 32), (256, 3, 64, 64), (4096, 1, 32, 32)]:
        x = torch.rand(shape, device=dev)
        out = torch.empty((shape[0], shape[2], shape[3], 3), dtype=torch.uint8, device=dev)
        for _ in range(3):
            _lib.check(L.dlpm_images_to_rgb8(x.data_ptr(), out.data_ptr(), *shape, _lib.stream_ptr()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            _lib.check(L.dlpm_images_to_rgb8(x.data_ptr(), out.data_ptr(), *shape, _lib.stream_ptr()))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        by = shape[0] * shape[2] * shape[3] * (4 * shape[1] + 3)
        print('images_to_rgb8 %-20s %7.1f us  %7.1f GB/s (algorithmic %d B/pixel)' % (shape, us, by / us / 1e3, 4 * shape[1] + 3))
    # 2. PNG writer alone (sample-like content: smooth + noise)
    g = np.random.default_rng(0)
    for H in (32, 64):
        n = 2048 if H == 32 else 512
        yy, xx = np.mgrid[0:H, 0:H]
        base = (128 + 80 * np.sin(xx / 5.0) * np.cos(yy / 7.0))[None, :, :, None]
        batch = np.clip(base + g.normal(0, 12, (n, H, H, 3)), 0, 255).astype(np.uint8)
        for threads in (1, 2, 4, 8):
            d = tempfile.mkdtemp()
            t0 = time.perf_counter()
            _lib.check(L.dlpm_png_write_rgb8(batch.ctypes.data, n, H, H, d.encode(), 0, 6, threads))
            dt = time.perf_counter() - t0
            sz = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d)) / n
            shutil.rmtree(d)
            print('png_write %dx%d  threads=%d  %8.0f images/s  (%.0f B/file)' % (H, H, threads, n / dt, sz))
        try:
            from PIL import Image
            d = tempfile.mkdtemp()
            t0 = time.perf_counter()
            for i in range(n):
                Image.fromarray(batch[i]).save(os.path.join(d, '%d.png' % i))
            dt = time.perf_counter() - t0
            shutil.rmtree(d)
            print('PIL save  %dx%d  (python loop)  %8.0f images/s' % (H, H, n / dt))
        except ImportError:
            pass
    # 3. chunk loop end to end (reference CIFAR UNet, short trajectories so the dump is visible next to the sampling)
    p = dlpm_amd.load_config('cifar10')
    p['device'] = dev
    torch.manual_seed(1234)
    net = dlpm_amd.init_model_by_parameter(p)
    dlpm_amd.rerandomize_(net, 4321)
    kw = dict(p['eval']['dlpm'])
    kw['reverse_steps'] = a.steps
    shape = [3, 32, 32]

    def run(mode):
        method = dlpm_amd.GenerativeLevyProcess(1.7, dev, a.steps, rescale_timesteps=True, seed=0)
        gm = dlpm_amd.GenerationManager(method, dlpm_amd.ShapeProbe(shape), True, **kw)
        d = tempfile.mkdtemp()
        gm.generate({'default': net}, 512 if mode == 'overlap+chunk512' else a.batch)   # warm: handle, workspace, graph
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == 'reference-style':
            from PIL import Image
            total, remaining = 0, a.G
            while remaining > 0:
                n = min(a.batch, remaining)
                gm.generate({'default': net}, n)                    # fp32 D2H inside
                s = gm.samples
                for i in range(n):                                  # what save_image does, per sample
                    arr = s[i].mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
                    Image.fromarray(arr).save(os.path.join(d, '%d.png' % (i + total)))
                total += n
                remaining -= n
        else:
            ev = dlpm_amd.EvaluationManager(method, gm, None, verbose=False, is_image=True, gen_data_path=d,
                                            overlap=(mode != 'serial'), device_batch=512 if mode == 'overlap+chunk512' else None)
            ev.evaluate_model({'default': net}, data_to_generate=a.G, batch_size=a.batch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        nfiles = len(os.listdir(d))
        shutil.rmtree(d)
        method.close()
        return dt, nfiles

    for mode in ('reference-style', 'serial', 'overlap', 'overlap+chunk512'):
        dt, nf = run(mode)
        print('chunk loop G=%d batch=%d T=%d  %-16s %7.3f s  (%d files, %.1f images/s)' % (a.G, a.batch, a.steps, mode, dt, nf, a.G / dt))


if __name__ == '__main__':
    main()
