"""Reference checkpoints: path hashing, file selection and the tensors that reach the
nets, against answers produced by the reference's own FileHandler / TrainingManager / EMAHelper
(tests/golden/f9_*; generator: tools/make_fixtures.py f9)."""
import json
import os

import numpy as np
import pytest
import torch
import yaml

from conftest import GOLDEN, golden
import dlpm_amd
from dlpm_amd import checkpoint as ck

CKPT = os.path.join(GOLDEN, 'f9_checkpoint_mlp.pt')
with open(os.path.join(GOLDEN, 'f9_paths.json')) as _f:
    PATHS = json.load(_f)


@pytest.mark.parametrize('name', ['cifar10', 'mnist', '2d_data', 'cifar10_lt'])
def test_experiment_and_eval_hashes_match_reference(name):
    p = dlpm_amd.load_config(name)
    assert [ck.get_exp_hash(p), ck.get_eval_hash(p)] == PATHS['hashes'][name]
    p['dlpm']['alpha'] = 1.7                       # what --alpha 1.7 does before the paths are derived
    assert [ck.get_exp_hash(p), ck.get_eval_hash(p)] == PATHS['hashes'][name + '@alpha1.7']
    p = dlpm_amd.load_config(name)
    p['method'] = 'lim'                            # --method lim
    assert [ck.get_exp_hash(p), ck.get_eval_hash(p)] == PATHS['hashes'][name + '@lim']


@pytest.mark.parametrize('i', range(len(PATHS['layouts'])))
def test_find_checkpoint_picks_the_file_the_reference_picks(tmp_path, i):
    lay = PATHS['layouts'][i]
    p = dlpm_amd.load_config(lay['config'])
    d = tmp_path / p['data']['dataset']
    d.mkdir()
    for f in lay['files']:
        (d / f).touch()
    got = ck.find_checkpoint(p, str(tmp_path), epoch=lay['epoch'])
    assert os.path.relpath(got, str(tmp_path)) == lay['model']
    assert os.path.relpath(ck.eval_folder(p, str(tmp_path)), str(tmp_path)) == lay['eval_dir']


def test_find_checkpoint_errors_like_the_reference(tmp_path):
    p = dlpm_amd.load_config('mnist')
    (tmp_path / 'mnist').mkdir()
    with pytest.raises(AssertionError, match='no models to load'):
        ck.find_checkpoint(p, str(tmp_path))


def test_reference_written_file_layout():
    c = ck.read_checkpoint(CKPT)
    assert c['epoch'] == 3 and c['steps'] == 3
    assert {'model_parameters', 'optimizer', 'learning_schedule', 'ema_models'} <= set(c)
    assert len(c['ema_models']) == 2
    raw = ck.model_state(c)
    net = dlpm_amd.MLPModel(dlpm_amd.load_config('2d_data'))
    assert list(raw) == list(net.state_dict())                       # same keys, same order
    for k, v in net.state_dict().items():
        assert tuple(raw[k].shape) == tuple(v.shape), k
    for i in range(2):
        sd = ck.model_state(c, ema=i)
        shadow = c['ema_models'][i]
        assert all(torch.equal(sd[k], shadow[k]) for k in shadow)
        assert any(not torch.equal(sd[k], raw[k]) for k in shadow)
    with pytest.raises(AssertionError, match='ema models'):
        ck.model_state(c, ema=2)
    with pytest.raises(AssertionError, match='no model_vae_parameters'):
        ck.model_state(c, name='vae')


def test_load_into_is_strict_and_drops_the_native_handle():
    net = dlpm_amd.MLPModel(dlpm_amd.load_config('2d_data'))
    assert ck.load_into(net, CKPT, ema=1) == (3, 3)
    c = ck.read_checkpoint(CKPT)
    assert all(torch.equal(net.state_dict()[k], v) for k, v in c['ema_models'][1].items())
    # names the shadow lacks because they alias a listed parameter carry the EMA value as well
    assert 'time_mlp.0.weight' not in c['ema_models'][1]
    assert torch.equal(net.state_dict()['time_mlp.0.weight'], c['ema_models'][1]['time_emb.weight'])
    unet = dlpm_amd.UNetModel(1, 32, 1, 1, [1], channel_mult=(1,), num_heads=4, use_scale_shift_norm=True)
    with pytest.raises(RuntimeError):
        ck.load_into(unet, CKPT)


def test_unet_round_trip_through_the_reference_file_format(tmp_path):
    torch.manual_seed(5)
    a = dlpm_amd.UNetModel(3, 32, 3, 1, [2], channel_mult=(1, 2), num_heads=4, use_scale_shift_norm=True)
    dlpm_amd.rerandomize_(a, 9)
    shadow = {k: v.detach() * 0.5 for k, v in a.named_parameters()}
    path = ck.save_checkpoint(str(tmp_path / 'model_abc_12.pt'), {'default': a}, epoch=12, steps=340,
                              ema_shadows={'default': [shadow]})
    torch.manual_seed(6)
    b = dlpm_amd.UNetModel(3, 32, 3, 1, [2], channel_mult=(1, 2), num_heads=4, use_scale_shift_norm=True)
    assert ck.load_into(b, path) == (12, 340)
    assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), b.state_dict().values()))
    ck.load_into(b, path, ema=0)
    assert all(torch.equal(b.state_dict()[k], v) for k, v in shadow.items())
    # DataParallel-style prefixes are accepted
    c = ck.read_checkpoint(path)
    c['model_parameters'] = {'module.' + k: v for k, v in c['model_parameters'].items()}
    ck.load_into(b, c)
    assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), b.state_dict().values()))


@pytest.mark.gpu
@pytest.mark.parametrize('ema', [None, 0, 1])
def test_checkpointed_mlp_runs_on_the_gpu_like_the_reference(ema):
    f = golden('f9_checkpoint_mlp_io')
    net = dlpm_amd.MLPModel(dlpm_amd.load_config('2d_data'))
    ck.load_into(net, CKPT, ema=ema)
    y = net(torch.from_numpy(f['x']).cuda(), torch.from_numpy(f['t']).cuda()).cpu().numpy()
    want = f['y_raw' if ema is None else 'y_ema%d' % ema]
    assert np.abs(y - want).max() < 2e-5 * max(1.0, np.abs(want).max())
    other = f['y_ema0' if ema is None else 'y_raw']
    assert np.abs(y - other).max() > 1e-3          # the three weight sets are distinguishable
