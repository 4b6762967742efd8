"""Reference checkpoints: where they live on disk and how their tensors reach the native nets.

The reference stores one `.pt` per experiment under `models/<name>/<dataset>/model_<exphash>[_<epoch>].pt`
(bem/utils_exp.py:52-151) holding

    {'epoch', 'steps', 'model_parameters': state_dict, 'optimizer': ..., 'learning_schedule': ...,
     'ema_models': [shadow_dict(mu_0), shadow_dict(mu_1), ...]}          (bem/TrainingManager.py:267-285)

where each EMA entry is `EMAHelper.state_dict()` = the bare `{param_name: tensor}` shadow of the
parameters with `requires_grad` (bem/utils_ema.py:15-21,58-62): buffers and frozen tensors are NOT in
it and must come from `model_parameters`.  The key names are those of `UNetModel` / `MLPModel`
(`input_blocks.7.1.qkv.weight`, `out.2.weight`, ...), which this build's parameter containers share.

This module only reads/writes the files and resolves the paths; the tensors are handed to
`UNetModel.load_state_dict` / `MLPModel.load_state_dict`, whose native handle is rebuilt lazily.
"""
import glob
import hashlib
import os

import torch


# ---------------------------------------------------------------------------------------- path hashing
def exp_hash_dict(p):
    """The sub-dictionary the reference hashes for DLPM experiments (dlpm/dlpm_experiment.py:11-19)."""
    m = p['method']
    return {'data': {k: v for k, v in p['data'].items() if k in ['dataset', 'channels', 'image_size']},
            m: {k: v for k, v in p[m].items()},
            'model': {k: v for k, v in p['model'].items()}}


def eval_hash_dict(p):
    """FileHandler.default_eval_hash (bem/utils_exp.py:48-50)."""
    return {'eval': p['eval'][p['method']]}


def _digest(d, n):
    return hashlib.sha256(str(d).encode('utf-8')).hexdigest()[:n]


def get_exp_hash(p):
    """16 hex characters (bem/utils_exp.py:53-61); depends on key ORDER, like the reference."""
    return _digest(exp_hash_dict(p), 16)


def get_eval_hash(p):
    """8 hex characters (bem/utils_exp.py:64-70)."""
    return _digest(eval_hash_dict(p), 8)


def _iteration_of(path):
    """The reference's test for a trailing epoch number: last '_' field of the name without '.pt' is a
    digit string shorter than 8 characters (bem/utils_exp.py:124-126)."""
    tail = str(path)[:-3].split('_')[-1]
    return int(tail) if tail.isdigit() and len(tail) < 8 else None


def find_checkpoint(p, folder_path, epoch=None):
    """Path of the model file `eval.py` would load (get_paths_from_param, bem/utils_exp.py:96-139):
    `folder/<dataset>/model_<hash>_<epoch>.pt` when an epoch is given, else the highest saved epoch,
    else the file without an epoch suffix."""
    h = get_exp_hash(p)
    d = os.path.join(folder_path, p['data']['dataset'])
    base = os.path.join(d, 'model_' + h)
    if epoch is not None:
        return base + '_' + str(epoch) + '.pt'
    found = glob.glob(base + '*')
    assert len(found) > 0, 'no models to load in {}, with hash {}'.format(d, h)
    best, best_path = 0, None
    for f in found:
        it = _iteration_of(f)
        if it is not None and it > best:
            best, best_path = it, f
    if best_path is not None:
        return best_path
    if os.path.exists(base + '.pt'):
        return base + '.pt'
    raise Exception('Did not find a model to load at location {} with hash {}'.format(d, h))


def eval_folder(p, folder_path):
    """`<folder>/<dataset>/new_eval_<exphash>_<evalhash>` (bem/utils_exp.py:85-94)."""
    return os.path.join(folder_path, p['data']['dataset'], '_'.join(('new_eval', get_exp_hash(p), get_eval_hash(p))))


# ---------------------------------------------------------------------------------------- tensors
def _names(name):
    """Checkpoint keys for a model name; 'default' keeps the historic un-suffixed keys
    (bem/TrainingManager.py:249-252)."""
    if name == 'default':
        return 'model_parameters', 'ema_models'
    return 'model_{}_parameters'.format(name), 'ema_models_{}'.format(name)


def _strip_module(sd):
    # nn.DataParallel wrappers prefix every key with 'module.'
    if sd and all(k.startswith('module.') for k in sd):
        return {k[len('module.'):]: v for k, v in sd.items()}
    return sd


def read_checkpoint(path):
    """torch.load on CPU.  The reference pickles optimizer/schedule state next to the tensors, so the
    file is a general pickle: only load files you trust (same as the reference's own loader)."""
    return torch.load(path, map_location='cpu', weights_only=False)


def model_state(ckpt, name='default', ema=None, model=None):
    """The state_dict to evaluate: raw weights (`ema=None`) or the `ema`-th EMA shadow overlaid on them.

    EvaluationManager evaluates an EMA by copying the shadow into the trainable parameters of a
    model copy (`param.data.copy_(shadow[name])` over `named_parameters()`, bem/utils_ema.py:32-37,54-56);
    everything the shadow lacks keeps the raw value.  `named_parameters()` lists a shared parameter
    once, so for nets that register one tensor under several names (the toy MLP: `time_emb.weight` is
    also `time_mlp.0.weight`, Model.py:70-100) the shadow holds only the first name; pass `model` so the
    other names of the same parameter receive the EMA value too, as they do in the reference."""
    k_model, k_ema = _names(name)
    assert k_model in ckpt, 'no {} in checkpoint (keys: {})'.format(k_model, list(ckpt))
    sd = dict(_strip_module(ckpt[k_model]))
    if ema is not None:
        assert k_ema in ckpt and ckpt[k_ema] is not None, 'no ema model in checkpoint'
        shadows = ckpt[k_ema]
        assert 0 <= ema < len(shadows), 'checkpoint holds {} ema models, asked for #{}'.format(len(shadows), ema)
        shadow = _strip_module(shadows[ema])
        unknown = [k for k in shadow if k not in sd]
        assert not unknown, 'ema shadow has parameters the model lacks: {}'.format(unknown[:4])
        alias = {k: k for k in shadow}
        if model is not None:
            first = {id(p): n for n, p in model.named_parameters()}
            for k, v in model.state_dict(keep_vars=True).items():
                if first.get(id(v)) in shadow:
                    alias[k] = first[id(v)]
        for k, src in alias.items():
            v = shadow[src]
            assert tuple(v.shape) == tuple(sd[k].shape), (k, tuple(v.shape), tuple(sd[k].shape))
            sd[k] = v
    return {k: v.detach().to(torch.float32) if torch.is_floating_point(v) else v for k, v in sd.items()}


def load_into(model, path_or_ckpt, name='default', ema=None):
    """Load a reference checkpoint into a dlpm_amd.UNetModel / MLPModel (strict: every key must match in
    name and shape, as `nn.Module.load_state_dict` does for the reference).  Returns the checkpoint's
    (epoch, steps)."""
    ckpt = read_checkpoint(path_or_ckpt) if isinstance(path_or_ckpt, (str, os.PathLike)) else path_or_ckpt
    model.load_state_dict(model_state(ckpt, name, ema, model), strict=True)   # also drops the stale native handle
    return ckpt.get('epoch'), ckpt.get('steps')


def save_checkpoint(path, models, epoch=0, steps=0, ema_shadows=None):
    """Write a file `TrainingManager.load` accepts (no optimizer / schedule state: `safe_load_state_dict`
    skips `None` destinations, bem/TrainingManager.py:242-244, so such a file loads when the manager was
    built for evaluation only).  `ema_shadows`: {name: [shadow_dict, ...]}."""
    ck = {'epoch': epoch, 'steps': steps}
    for name, model in models.items():
        k_model, k_ema = _names(name)
        sfx = '' if name == 'default' else '_' + name
        ck[k_model] = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        ck['optimizer' + sfx] = None
        ck[('learning_schedule' if name == 'default' else 'learnin_schedule') + sfx] = None
        if ema_shadows and name in ema_shadows:
            ck[k_ema] = [{k: v.detach().cpu() for k, v in s.items()} for s in ema_shadows[name]]
    torch.save(ck, path)
    return path
