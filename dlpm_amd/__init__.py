"""dlpm_amd -- MI355X-native (gfx950) implementation of DLPM's reverse-time sampling loop.

Host-side mirror of the reference's entry points over libdlpm_amd.so (include/dlpm_amd.h):

    GenerationManager(method, dataloader, is_image, **eval_kwargs).generate(models, nsamples)
    GenerativeLevyProcess(alpha, device, reverse_steps, ...).sample(models, shape, reverse_steps, ...)
    UNetModel(...) / MLPModel(p)   with   model(x, t) -> eps

See DESIGN.md for the scope and INTEGRATION.md for the binding.
"""
from .method import GenerativeLevyProcess, ReferenceStreams, init_method_by_parameter  # noqa: F401
from .process import DLPM  # noqa: F401
from .unet import UNetModel, unet_from_config  # noqa: F401
from .mlp import MLPModel  # noqa: F401
from .generation import GenerationManager, ShapeProbe  # noqa: F401
from .evaluation import EvaluationManager, ImageDump  # noqa: F401
from .weights import rerandomize_  # noqa: F401
from .config import load_config, is_image_dataset, init_model_by_parameter  # noqa: F401
from . import checkpoint  # noqa: F401
