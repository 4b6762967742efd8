"""GenerationManager: drop-in for bem/GenerationManager.py:13-63 (the part eval.py --generate uses).

`generate(models, nsamples, ...)` probes the dataloader for the per-sample shape, calls
`method.sample`, clamps to +-1 (images) / +-6 (2-D), applies the inverse affine transform (x+1)/2 for
images (bem/datasets/__init__.py:108-109) on the GPU, and leaves CPU tensors in `.samples` /
`.history`.  Plotting / animation helpers of the reference file are out of scope.
"""
import copy

import torch

from . import _lib


class ShapeProbe:
    """Stands in for the torch DataLoader the reference passes (only its first batch's shape is
    ever used: GenerationManager.py:40-42)."""

    def __init__(self, sample_shape):
        self.sample_shape = list(sample_shape)

    def __iter__(self):
        yield torch.empty([1] + self.sample_shape), torch.empty(1)


class GenerationManager:
    def __init__(self, method, dataloader, is_image, **kwargs):
        self.method = method
        self.original_data = dataloader
        self.is_image = is_image
        self.kwargs = kwargs
        self.samples = []
        self.history = []

    def _post(self, x, to_host=True):
        """clamp + (x+1)/2 in one HIP kernel, then the D2H copy (skipped when the caller keeps the samples on
        the GPU, e.g. dlpm_amd.EvaluationManager's image dump)."""
        clamp = 1.0 if self.is_image else 6.0
        if x.is_cuda:
            x = x.contiguous().float()
            out = torch.empty_like(x)
            _lib.check(_lib.lib().dlpm_postprocess_f32(x.data_ptr(), out.data_ptr(), x.numel(), clamp,
                                                      1 if self.is_image else 0, _lib.stream_ptr()))
            return out.cpu() if to_host else out
        raise _lib.DlpmError('GenerationManager expects samples on the GPU; there is no CPU fallback')

    def generate(self, models, nsamples, get_sample_history=False, print_progression=False, to_host=True, declare_batch=True, **kwargs):
        assert nsamples > 0, 'nsamples must be greater than 0, got {}'.format(nsamples)
        tmp_kwargs = copy.deepcopy(self.kwargs)
        tmp_kwargs.update(kwargs)
        if declare_batch:
            # nets whose caller never declared a batch learn the one this call samples (UNetModel.declare_batch): the kernel choice stays a
            # function of the layer and of a DECLARATION -- here "generate(models, nsamples)" -- never of the batch a launch happens to
            # carry.  Callers that sample one evaluation in chunks declare once and pass declare_batch=False (EvaluationManager).
            for m in (models or {}).values():
                if hasattr(m, 'declare_batch'):
                    m.declare_batch(nsamples)
        _, (data, y) = next(enumerate(self.original_data))
        size = list(data.size())
        size[0] = nsamples
        x = self.method.sample(shape=size, models=models, print_progression=print_progression,
                               get_sample_history=get_sample_history, **tmp_kwargs)
        nfeat = data.shape[-1]
        if get_sample_history:
            _, hist = x
            self.samples = self._post(hist[-1, ..., :nfeat], to_host)
            self.history = self._post(hist[..., :nfeat], to_host)
        else:
            self.samples = self._post(x[..., :nfeat], to_host)
            self.history = []
        return self.samples
