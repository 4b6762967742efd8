"""LIM sampler (`method: lim`): oracle/lim.py against the reference's VPSDE values and
sample() trajectories (tests/golden/f10_*), then the HIP path against both."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import lim as olim, nets, sampler as osampler

T_ = torch.from_numpy


class Synth:
    def __call__(self, x, t):
        return 0.5 * x + t.view(-1, *([1] * (x.dim() - 1)))


def _mlp_sd():
    f = golden('f10_lim_sde_mlp')
    return {k[3:]: T_(f[k]) for k in f.files if k.startswith('w__')}


@pytest.mark.parametrize('alpha', [1.5, 1.7, 1.8, 2.0])
@pytest.mark.parametrize('steps', [10, 100, 1000])
def test_vpsde_functions_bit_exact(alpha, steps):
    f = golden('f10_lim_vpsde')
    tag = 'a%s_n%d_' % (str(alpha).replace('.', 'p'), steps)
    sde = olim.VPSDE(alpha)
    ts = olim.timesteps(sde, steps)
    assert np.array_equal(ts.numpy(), f[tag + 'ts'])
    assert np.array_equal(sde.beta(ts).numpy(), f[tag + 'beta'])
    assert np.array_equal(sde.marginal_log_mean_coeff(ts).numpy(), f[tag + 'logmean'])
    assert np.array_equal(sde.diffusion_coeff(ts).numpy(), f[tag + 'diff'])
    assert np.array_equal(sde.marginal_std(ts).numpy(), f[tag + 'std'], equal_nan=True)


ORACLE_TRAJ = ['f10_lim_sde_toy', 'f10_lim_ode_toy', 'f10_lim_sde_img', 'f10_lim_ode_img', 'f10_lim_sde_img_b20',
               'f10_lim_sde_gauss', 'f10_lim_sde_mlp', 'f10_lim_ode_mlp']


@pytest.mark.parametrize('name', ORACLE_TRAJ)
def test_oracle_trajectories_same_seeds(name):
    f = golden(name)
    steps, alpha, ode, ce = f['meta']
    shape = [int(v) for v in f['shape']]
    if name.endswith('mlp'):
        sd = _mlp_sd()
        model = lambda x, t: nets.mlp_forward(sd, x, t)
    else:
        model = Synth()
    x, hist = olim.sample(model, shape, int(steps), float(alpha), osampler.Streams(0, 0), ode=bool(ode),
                          clamp_eps=None if ce < 0 else float(ce), get_sample_history=True)
    want = f['history']
    assert hist.shape == want.shape
    scale = np.abs(want).max(axis=tuple(range(1, want.ndim)), keepdims=True) + 1e-6
    assert np.max(np.abs(hist.numpy() - want) / scale) < 2e-5
    np.testing.assert_allclose(x.numpy(), f['final'], rtol=2e-4, atol=2e-4 * float(np.abs(f['final']).max()))


# ------------------------------------------------------------------------------------------------ host mirror
@pytest.mark.parametrize('alpha', [1.5, 1.8, 2.0])
@pytest.mark.parametrize('ode', [False, True])
def test_lim_tables_mirror_and_native(alpha, ode):
    """dlpm_amd.lim.lim_tables (torch fp32 op sequence over the [steps] grid) == the oracle's per-step coefficients
    (the same ops on [B]-vectors of equal values, as the reference evaluates them) to a few ulp -- torch's CPU pow/exp
    take a vector or a scalar-tail code path depending on the length, so the reference's own values move by an
    ulp with B; libdlpm_amd's dlpm_lim_tables_f32 (double arithmetic) within a few fp32 ulp."""
    from dlpm_amd.lim import VPSDE, lim_tables
    steps = 50
    ts, tmp, cx, cs, cn = lim_tables(VPSDE(alpha), steps, ode)
    sde = olim.VPSDE(alpha)
    ots = olim.timesteps(sde, steps)
    assert torch.equal(ts, ots)
    for i in (0, 1, 17, 48, 49):                 # per-step evaluation on [B]-vectors of equal values, as the reference does
        s, t = torch.ones(32) * ots[i], torch.ones(32) * ots[i + 1]
        otmp, ocx, ocs, ocn = olim.step_coefficients(sde, s, t, ode)
        for got, want in ((tmp[i], otmp[0]), (cx[i], ocx[0]), (cs[i], ocs[0]), (cn[i], torch.tensor(0.0) if ode else ocn[0])):
            assert abs(float(got) - float(want)) <= 4e-7 * abs(float(want)), (i, float(got), float(want))
    nts, ntmp, ncx, ncs, ncn = lim_tables(VPSDE(alpha), steps, ode, native=True)
    np.testing.assert_allclose(nts.numpy(), ts.numpy(), rtol=2.4e-7)      # torch's vectorised linspace rounds differently
    # cx - 1 and cn are differences of nearly equal numbers in fp32; double arithmetic is the more accurate side
    np.testing.assert_allclose(ntmp.numpy(), tmp.numpy(), rtol=5e-5)      # 1 - exp(small) cancels in fp32 near t = 0
    np.testing.assert_allclose(ncx.numpy(), cx.numpy(), rtol=5e-5)        # log(cos) near t = T (cos -> 0) likewise
    np.testing.assert_allclose(ncs.numpy(), cs.numpy(), rtol=5e-4, atol=1e-7)
    np.testing.assert_allclose(ncn.numpy(), cn.numpy(), rtol=5e-4, atol=1e-7)


def test_lim_constructor_contract():
    import dlpm_amd
    with pytest.raises(AssertionError, match='rescaled timesteps'):
        dlpm_amd.GenerativeLevyProcess(1.8, 'cpu', 10, LIM=True, rescale_timesteps=False)
    p = dlpm_amd.load_config('cifar10')
    p['method'], p['device'] = 'lim', 'cpu'
    m = dlpm_amd.init_method_by_parameter(p)
    assert m.LIM and m.reverse_steps == 1000 and m.sde.T == 0.9946


# ------------------------------------------------------------------------------------------------ GPU
class SynthT(torch.nn.Module):
    def forward(self, x, t):
        return 0.5 * x + t.view(-1, *([1] * (x.dim() - 1)))


def _native_mlp():
    import dlpm_amd
    m = dlpm_amd.MLPModel(dlpm_amd.load_config('2d_data'))
    m.load_state_dict(_mlp_sd())
    return m


def _check_traj(name, model, **kw):
    import dlpm_amd
    f = golden(name)
    steps, alpha, ode, ce = f['meta']
    shape = [int(v) for v in f['shape']]
    meth = dlpm_amd.GenerativeLevyProcess(float(alpha), 'cuda', int(steps), rescale_timesteps=True, LIM=True,
                                          rng='reference', seed=0, **kw)
    x, hist = meth.sample({'default': model}, shape, int(steps), deterministic=bool(ode),
                          clamp_eps=None if ce < 0 else float(ce), get_sample_history=True)
    want = f['history']
    got = hist.cpu().numpy()
    assert got.shape == want.shape
    scale = np.abs(want).max(axis=tuple(range(1, want.ndim)), keepdims=True) + 1e-6
    assert np.max(np.abs(got - want) / scale) < 5e-5
    fin = f['final']
    assert np.abs(x.cpu().numpy() - fin).max() < 1e-4 * max(1.0, np.abs(fin).max())
    meth.close()


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['f10_lim_sde_toy', 'f10_lim_ode_toy', 'f10_lim_sde_img', 'f10_lim_ode_img',
                                  'f10_lim_sde_img_b20', 'f10_lim_sde_gauss'])
def test_lim_trajectory_identical_seeds_callable_model(name):
    _check_traj(name, SynthT().cuda())


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['f10_lim_sde_mlp', 'f10_lim_ode_mlp'])
def test_lim_trajectory_identical_seeds_native_mlp(name):
    _check_traj(name, _native_mlp())


@pytest.mark.gpu
def test_lim_trajectory_identical_seeds_native_unet():
    from test_host_mirror import build_unet
    from dlpm_amd.weights import state_digest
    f = golden('f10_lim_sde_unet_tiny')
    net, _ = build_unet('tiny')
    assert state_digest(net) == str(f['weight_digest'])
    _check_traj('f10_lim_sde_unet_tiny', net)


@pytest.mark.gpu
def test_lim_philox_graph_eager_shards_history():
    import dlpm_amd
    from test_host_mirror import build_unet
    net, _ = build_unet('tiny2')
    steps, alpha = 12, 1.8

    def run(B, offset, graph, ode=False, hist=False):
        m = dlpm_amd.GenerativeLevyProcess(alpha, 'cuda', steps, rescale_timesteps=True, LIM=True, seed=7,
                                           sample_offset=offset, use_graph=graph)
        out = m.sample({'default': net}, [B, 1, 16, 16], steps, deterministic=ode, clamp_eps=50.0, get_sample_history=hist)
        m.close()
        return out
    full_g, full_e = run(6, 0, True).cpu(), run(6, 0, False).cpu()
    assert torch.equal(full_g, full_e) and torch.isfinite(full_g).all()
    assert torch.equal(torch.cat([run(2, 0, True).cpu(), run(4, 2, True).cpu()]), full_g)
    x, hist = run(6, 0, True, hist=True)
    assert hist.shape == (steps + 1, 6, 1, 16, 16) and torch.equal(hist[-1], x) and torch.equal(x.cpu(), full_g)
    # the ODE is deterministic given x_0: two keys with the same x_0 ... different keys give different x_0
    ode_a, ode_b = run(6, 0, True, ode=True).cpu(), run(6, 0, False, ode=True).cpu()
    assert torch.equal(ode_a, ode_b) and not torch.equal(ode_a, full_g)


@pytest.mark.gpu
def test_lim_generation_manager_with_config_kwargs():
    """eval.lim kwargs ({deterministic, reverse_steps, clip_denoised}) through GenerationManager, toy data."""
    import dlpm_amd
    p = dlpm_amd.load_config('2d_data')
    p['method'], p['device'] = 'lim', 'cuda'
    kw = dict(p['eval']['lim'])
    kw['reverse_steps'] = 20
    m = dlpm_amd.init_method_by_parameter(p, seed=3)
    m.reverse_steps = 20
    m.dlpm.rescale_diffusion(20)
    gm = dlpm_amd.GenerationManager(m, dlpm_amd.ShapeProbe([1, 2]), False, **kw)
    s = gm.generate({'default': _native_mlp()}, 64)
    assert s.shape == (64, 1, 2) and torch.isfinite(s).all() and s.abs().max() <= 6.0
