"""EvaluationManager: the generate-and-dump half of bem/evaluate/EvaluationManager.py.

The reference generates `data_to_generate` images in chunks of `eval.batch_size`; after every chunk it copies
the fp32 samples to the host and calls `torchvision.utils.save_image` once per sample, serially, before the next
chunk starts (EvaluationManager.py:174-196).  Here a chunk's samples stay on the GPU, are quantised to 8-bit RGB
by `dlpm_images_to_rgb8`, cross PCIe on a side stream into pinned memory (3 bytes/pixel), and are encoded and
written by native threads (`dlpm_png_write_rgb8`) while the next chunk is sampling.  File names and pixel values
are the reference's (`<gen_data_path>/<i>.png`, i counting over all chunks).  With the Philox generator the whole
dump is one `dataset_stream()`: sample i is the same whatever the chunk size, so `device_batch` may enlarge the
chunks beyond `eval.batch_size` without changing a pixel.

The metrics half (FID / PRDC / Wasserstein / MMD) needs third-party packages and Inception weights that are not
available offline and is out of scope: `evals` keeps the reference's keys but nothing is appended to them.
"""
import copy
import ctypes as C
import os
import queue
import threading

import numpy as np
import torch

from . import _lib


class ImageDump:
    """[n,C,H,W] fp32 images in [0,1] on the GPU -> `<out_dir>/<first_index+i>.png`, pipelined:
    stream of the caller: quantise -> side stream: D2H into a pinned slot -> worker thread: PNG encode + write."""

    def __init__(self, out_dir, C_, H, W, max_batch, level=6, threads=4, overlap=True, slots=2):
        os.makedirs(out_dir, exist_ok=True)
        self.out_dir, self.shape, self.max_batch = out_dir, (C_, H, W), max_batch
        self.level, self.threads, self.overlap = level, threads, overlap
        self.dev = [torch.empty((max_batch, H, W, 3), dtype=torch.uint8, device='cuda') for _ in range(slots)]
        self.host = [torch.empty((max_batch, H, W, 3), dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.copy_stream = torch.cuda.Stream()
        self.free = queue.Queue()
        for s in range(slots):
            self.free.put(s)
        self.pending = queue.Queue()
        self.error = None
        self.written = 0
        self.worker = None
        if overlap:
            self.worker = threading.Thread(target=self._run, name='dlpm-png-writer', daemon=True)
            self.worker.start()

    def _write(self, slot, n, first_index):
        _lib.check(_lib.lib().dlpm_png_write_rgb8(self.host[slot].data_ptr(), n, self.shape[1], self.shape[2],
                                                 self.out_dir.encode(), first_index, self.level, self.threads))
        self.written += n

    def _run(self):
        while True:
            item = self.pending.get()
            if item is None:
                return
            slot, n, first_index, done = item
            try:
                done.synchronize()
                if self.error is None:
                    self._write(slot, n, first_index)
            except Exception as e:            # surfaced by submit()/close() on the caller's thread
                self.error = e
            finally:
                self.free.put(slot)

    def submit(self, x, first_index):
        if self.error is not None:
            raise self.error
        n = x.shape[0]
        assert x.is_cuda and x.dtype == torch.float32 and tuple(x.shape[1:]) == self.shape and n <= self.max_batch, x.shape
        x = x.contiguous()
        slot = self.free.get()                # waits while both slots are still being written
        _lib.check(_lib.lib().dlpm_images_to_rgb8(x.data_ptr(), self.dev[slot].data_ptr(), n, *self.shape,
                                                 _lib.stream_ptr()))
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(ready)
            self.host[slot][:n].copy_(self.dev[slot][:n], non_blocking=True)
            done = torch.cuda.Event()
            done.record()
        if self.overlap:
            self.pending.put((slot, n, first_index, done))
        else:
            done.synchronize()
            try:
                self._write(slot, n, first_index)
            finally:
                self.free.put(slot)

    def close(self):
        if self.worker is not None:
            self.pending.put(None)
            self.worker.join()
            self.worker = None
        if self.error is not None:
            raise self.error


DEVICE_BATCH = 1024   # images per chunk under device_batch='auto' (the per-sample rate of the image nets is flat beyond it)


def auto_device_batch(models, chw, data_to_generate, reserve=0.5, device=None):
    """Largest power of two <= DEVICE_BATCH (and not beyond the dump) whose network workspaces + state fit in `reserve` of the
    free HBM OF `device` (the GPU the method and its nets live on -- not the process's current device: a sharded evaluation
    runs the method on cuda:k with current device 0).  Nets that cannot tell their workspace (`workspace_bytes(B, image_size)`)
    count as 64 x the state.  The sampler's [T,B] tables, graph pools and the dump's pinned buffers live in the reserve; the
    caller halves the chunk if the first one still runs out of memory (`_evaluate_model`)."""
    free, _ = torch.cuda.mem_get_info(device)
    b = 1
    while b * 2 <= min(DEVICE_BATCH, max(1, int(data_to_generate))):
        b *= 2
    state = 4 * chw[0] * chw[1] * chw[2]
    while b > 1:
        need = 4 * b * state
        for m in (models or {}).values():
            wb = getattr(m, 'workspace_bytes', None)
            need += wb(b, chw[1]) if wb is not None else 64 * b * state
        if need <= reserve * free:
            break
        b //= 2
    return b


class EvaluationManager:
    """Constructor and entry points of bem/evaluate/EvaluationManager.py:33-118; `_evaluate_model` covers the
    generation + image-dump part (:174-196) and the 2-D generation call (:135)."""

    def __init__(self, method, gen_manager, dataloader, verbose=True, logger=None, is_image=False, gen_data_path=None,
                 real_data_path=None, overlap=True, png_level=6, png_threads=4, device_batch='auto', **kwargs):
        self.method, self.gen_manager, self.dataloader = method, gen_manager, dataloader
        self.verbose, self.logger, self.is_image = verbose, logger, is_image
        self.gen_data_path, self.real_data_path = gen_data_path, real_data_path
        self.overlap, self.png_level, self.png_threads = overlap, png_level, png_threads
        self.device_batch = device_batch
        self.kwargs = kwargs
        self.evals = None
        self.reset()

    def reset(self, keep_losses=False, keep_evals=False):
        old = self.evals or {}
        e = {}
        for k in ('losses', 'losses_batch'):                                        # EvaluationManager.py:60-77
            e[k] = old[k] if keep_losses else np.array([], dtype=np.float32)
        for k in ('wass', 'mmd', 'precision', 'recall', 'density', 'coverage', 'f_1_pr', 'f_1_dc', 'fid', 'fig'):
            e[k] = old[k] if keep_evals else []
        e['grad_norm'] = old['grad_norm'] if keep_evals else np.array([], dtype=np.float32)
        self.evals = e

    def generate_default(self, models, nsamples, **kwargs):
        self.gen_manager.generate(models, nsamples, **kwargs)
        return self.gen_manager

    def evaluate_model(self, models, **kwargs):
        tmp_kwargs = copy.deepcopy(self.kwargs)
        tmp_kwargs.update(kwargs)
        return self._evaluate_model(models, **tmp_kwargs)

    def _evaluate_model(self, models, data_to_generate, batch_size, fig_lim=1.5, callback_on_logging=None, **kwargs):
        if not self.is_image:
            self.gen_manager.generate(models, data_to_generate, **kwargs)          # EvaluationManager.py:135
            return {'generated': data_to_generate, 'gen_data_path': None}
        assert self.gen_data_path is not None, 'gen_data_path is needed to save the generated images'
        total = 0
        if data_to_generate != 0:
            _, (data, _) = next(enumerate(self.gen_manager.original_data))
            Cc, H, W = data.shape[1:]
            stream = getattr(self.method, 'dataset_stream', None)
            min_batch = None
            if self.device_batch and stream is not None and getattr(self.method, 'rng', None) == 'philox':
                # inside dataset_stream() the i-th sample does not depend on the chunking, so the chunk can be
                # sized for the GPU (eval.batch_size = 64 leaves an MI355X half idle: the CIFAR net runs at 46 % of its
                # B = 1024 per-sample rate there) without changing any pixel.  'auto' (the default since round 5: a caller
                # that drops the classes in unchanged gets the fast shape) = up to DEVICE_BATCH images, as many as the nets'
                # workspaces leave room for in free HBM; device_batch=0 / None keeps the reference's chunking.
                dev_b = self.device_batch
                if dev_b == 'auto':
                    mdev = torch.device(getattr(self.method, 'device', 'cuda'))
                    dev_b = auto_device_batch(models, [Cc, H, W], data_to_generate, device=mdev if mdev.type == 'cuda' else None)
                    min_batch = batch_size                                          # an OOM on the first chunk halves down to the config's own
                batch_size = max(batch_size, int(dev_b))
            dump = ImageDump(self.gen_data_path, Cc, H, W, min(batch_size, data_to_generate), level=self.png_level,
                             threads=self.png_threads, overlap=self.overlap)
            remaining = data_to_generate
            # (the chunks are sampled with declare_batch=False: the nets keep whatever batch their owner declared -- none by default --
            #  so that a pixel depends neither on the chunking nor on the last chunk's size; tests/test_image_dump.py)
            if self.verbose:
                print('generating {} images for fid computation'.format(remaining))
            ctx = stream() if stream is not None else _null()
            try:
                with ctx:
                    while remaining > 0:                                           # EvaluationManager.py:181-193
                        n = min(batch_size, remaining)
                        try:
                            self.gen_manager.generate(models, n, to_host=False, declare_batch=False, **kwargs)
                        except (torch.cuda.OutOfMemoryError, RuntimeError) as e:
                            # 'auto' sized the chunk from free HBM with a reserve; should the FIRST chunk not fit after all, halve it
                            # (the pixels do not depend on the chunking inside dataset_stream()) instead of failing the dump
                            oom = isinstance(e, torch.cuda.OutOfMemoryError) or 'out of memory' in str(e).lower()
                            if not (oom and total == 0 and min_batch is not None and batch_size // 2 >= max(1, min_batch)):
                                raise
                            batch_size //= 2
                            torch.cuda.empty_cache()
                            if self.verbose:
                                print('device_batch=auto: out of memory, chunk halved to {}'.format(batch_size))
                            continue
                        dump.submit(self.gen_manager.samples, total)
                        total += n
                        remaining -= n
                        if self.verbose:
                            print(remaining, end=' ', flush=True)
            finally:
                dump.close()
            assert data_to_generate == total == dump.written
            if self.verbose:
                print('saved generated data in {}.'.format(self.gen_data_path))
        return {'generated': total, 'gen_data_path': self.gen_data_path}


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
