"""CPU-side checks of the product: the C ABI loads and exports every symbol of include/dlpm_amd.h,
the host functions (schedule, MT19937 streams) match the golden vectors, the parameter containers
are seed-identical to the reference, and the oracle UNet restatement matches the reference outputs
when fed those weights.  No GPU compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import golden, ROOT
import dlpm_amd
from dlpm_amd import _lib
from dlpm_amd.weights import rerandomize_, state_digest
from oracle import nets

UNETS = {
    'tiny': dict(in_ch=3, mc=32, mult=[1, 2], attn=[2], heads=4, res=1),
    'tiny2': dict(in_ch=1, mc=32, mult=[1, 2, 2], attn=[2, 4], heads=4, res=2),
    'mnist': dict(in_ch=1, mc=32, mult=[1, 2, 2, 2], attn=[2, 4], heads=4, res=2),
    'cifar': dict(in_ch=3, mc=128, mult=[1, 2, 2, 2], attn=[4, 8, 16], heads=4, res=2),
    # mc = 128 at 16x16: every ResBlock / Upsample conv qualifies for the Winograd F(4x4,3x3) kernel
    'wide': dict(in_ch=3, mc=128, mult=[1, 2], attn=[2], heads=4, res=2),
}


def build_unet(name, seed=1234, reseed=4321):
    c = UNETS[name]
    torch.manual_seed(seed)
    net = dlpm_amd.UNetModel(c['in_ch'], c['mc'], c['in_ch'], c['res'], c['attn'], channel_mult=c['mult'],
                             num_heads=c['heads'], use_scale_shift_norm=True)
    d0 = state_digest(net)
    rerandomize_(net, reseed)
    return net, d0


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'dlpm_amd.h')).read()
    declared = set(re.findall(r'\b(dlpm_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'dlpm_status', 'dlpm_update_flags'}
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), 'libdlpm_amd.so does not export %s' % name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert L.dlpm_abi_version() == _lib.ABI_VERSION == 6


def test_error_channel():
    L = _lib.lib()
    out = np.empty(4, np.float32)
    with pytest.raises(ValueError, match='alpha'):
        _lib.check(L.dlpm_schedule_f32(10, 2.5, *[out.ctypes.data] * 4))
    st = _lib.MT19937()
    _lib.check(L.dlpm_mt19937_seed(C.byref(st), 0))
    with pytest.raises(ValueError, match='Wrong value of alpha'):
        _lib.check(L.dlpm_skewed_levy_host_f32(C.byref(st), 0.0, 4, -1.0, out.ctypes.data))


@pytest.mark.parametrize('T,alpha', [(100, 1.7), (1000, 1.7), (1000, 1.8), (4000, 1.8), (20, 1.5)])
def test_native_schedule_close_and_mirror_schedule_exact(T, alpha):
    f = golden('f1_schedule')
    tag = 'T%d_a%s' % (T, str(alpha).replace('.', 'p'))
    out = [np.empty(T, np.float32) for _ in range(4)]
    _lib.check(_lib.lib().dlpm_schedule_f32(T, alpha, *[o.ctypes.data for o in out]))
    # gammas / bargammas within 1-2 fp32 ulp; sigmas are (1 - g^alpha)^(1/alpha) of g ~ 1, i.e. a
    # cancellation that amplifies a 1-ulp difference in g: bounded in ABSOLUTE terms
    np.testing.assert_allclose(out[0], f[tag + '_g'], rtol=2.5e-7)
    np.testing.assert_allclose(out[1], f[tag + '_bg'], rtol=2e-6)
    np.testing.assert_allclose(out[2], f[tag + '_s'], atol=2e-5)
    np.testing.assert_allclose(out[3], f[tag + '_bs'], atol=2e-5)
    # the Python mirror (default path) is bit-exact
    d = dlpm_amd.DLPM(alpha, 'cpu', T)
    for name, v in zip(['g', 'bg', 's', 'bs'], d.host_schedule):
        assert np.array_equal(v.numpy(), f[tag + '_' + name])


def test_rescale_diffusion():
    d = dlpm_amd.DLPM(1.7, 'cpu', 4000)
    d.rescale_diffusion(100)
    assert np.array_equal(d.bargammas.numpy(), golden('f1_schedule')['rescaled_4000_to_100_a1p7_bg'])


@pytest.mark.parametrize('tag', ['s0_a1p7', 's1_a1p5', 's2_a1p8', 's3_a1p9', 's5_a1p2', 's4_a2p0'])
def test_host_skewed_levy_stream(tag):
    f = golden('f2_skewed_levy')
    seed, alpha, n, clamp = f[tag + '_meta']
    s = dlpm_amd.ReferenceStreams(int(seed), 0)
    a = s.skewed_levy(alpha, int(n), None if clamp < 0 else clamp).numpy()
    a7 = s.skewed_levy(alpha, 7, None if clamp < 0 else clamp).numpy()
    assert np.mean(a == f[tag + '_a']) > 0.99
    np.testing.assert_allclose(a, f[tag + '_a'], rtol=2e-7)
    np.testing.assert_allclose(a7, f[tag + '_a_next7'], rtol=2e-7)


def test_host_randn_stream():
    f = golden('f2_randn')
    for seed in (0, 7, 123):
        s = dlpm_amd.ReferenceStreams(0, seed)
        keys = sorted([k for k in f.files if k.startswith('s%d_call' % seed)],
                      key=lambda k: int(k.split('call')[1].split('_')[0]))
        for k in keys:
            n = int(k.split('_n')[1])
            np.testing.assert_allclose(s.randn([n]).numpy(), f[k], rtol=0, atol=4e-7 if n >= 16 else 0, err_msg=k)


def test_numpy_global_state_roundtrip():
    import scipy.stats
    np.random.seed(77)
    np.random.random_sample(5)
    s = dlpm_amd.ReferenceStreams.from_global_numpy(0)
    mine = s.skewed_levy(1.7, 9).numpy()
    theirs = scipy.stats.levy_stable.rvs(1.7 / 2, 1, loc=0, scale=2 * np.cos(np.pi * 1.7 / 4) ** (2 / 1.7), size=9)
    np.testing.assert_allclose(mine, theirs.astype(np.float32), rtol=2e-7)
    s.store_global_numpy()
    ref_next = np.random.random_sample(3)
    np.random.seed(77)
    np.random.random_sample(5 + 18)
    assert np.array_equal(ref_next, np.random.random_sample(3))


def test_reference_prologue_matches_trajectory_fixture():
    f = golden('f5_traj_synth_img')
    m = dlpm_amd.GenerativeLevyProcess(1.7, 'cpu', 50, rescale_timesteps=True, rng='reference', seed=0)
    A, xT = m._host_noise_prologue([2, 3, 4, 4], 10.0, 50.0, None)
    np.testing.assert_allclose(A.numpy(), f['A'], rtol=2e-7)
    np.testing.assert_allclose(xT.numpy(), f['xT'], rtol=1e-6, atol=1e-6)
    z = m._streams().randn([2, 3, 4, 4]).numpy()
    np.testing.assert_allclose(z, f['z'][0], atol=4e-7)


@pytest.mark.parametrize('name', list(UNETS))
def test_seed_identical_weights_and_oracle_unet(name):
    f = golden('f6_unet_' + name)
    net, d0 = build_unet(name)
    assert d0 == bytes(f['digest_init']).hex(), 'default init differs from the reference under the same seed'
    assert state_digest(net) == bytes(f['digest_final']).hex()
    assert sum(p.numel() for p in net.parameters()) == int(f['nparams'])
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    x, t = torch.from_numpy(f['x']), torch.from_numpy(f['t'])
    with torch.no_grad():
        y, feats = nets.unet_forward(sd, x, t, UNETS[name]['heads'], return_feats=True)
        y2 = nets.unet_forward(sd, x, torch.from_numpy(f['t_same']), UNETS[name]['heads'])
    np.testing.assert_allclose(y.numpy(), f['y'], atol=2e-5)
    np.testing.assert_allclose(y2.numpy(), f['y_same_t'], atol=2e-5)
    stats = []
    for grp in ('down', 'up'):
        for ft in feats[grp]:
            stats.append([ft.mean().item(), ft.abs().mean().item()] + ft.flatten()[:4].tolist())
    ft = feats['middle']
    stats.append([ft.mean().item(), ft.abs().mean().item()] + ft.flatten()[:4].tolist())
    np.testing.assert_allclose(np.array(stats), f['block_stats'], atol=2e-5)


def test_mlp_container_matches_reference_state_dict():
    f = golden('f6_mlp_forward')
    p = dlpm_amd.load_config('2d_data')
    torch.manual_seed(1)
    m = dlpm_amd.MLPModel(p)
    ref = {k[3:]: f[k] for k in f.files if k.startswith('w__')}
    sd = m.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    assert all(np.array_equal(sd[k].numpy(), ref[k]) for k in ref)


def test_no_cpu_fallback():
    net, _ = build_unet('tiny')
    with pytest.raises(_lib.DlpmError, match='no CPU fallback'):
        net(torch.zeros(1, 3, 16, 16), torch.zeros(1))
    gm = dlpm_amd.GenerationManager(None, dlpm_amd.ShapeProbe([3, 4, 4]), True)
    with pytest.raises(_lib.DlpmError):
        gm._post(torch.zeros(2, 3, 4, 4))


@pytest.mark.parametrize('T,alpha', [(100, 1.7), (1000, 1.8), (30, 1.5)])
def test_exploding_schedule_mirror_exact_native_close(T, alpha):
    f = golden('f1_schedule')
    tag = 'expl_T%d_a%s' % (T, str(alpha).replace('.', 'p'))
    d = dlpm_amd.DLPM(alpha, 'cpu', T, scale='scale_exploding')
    for name, v in zip(['g', 'bg', 's', 'bs'], d.host_schedule):
        assert np.array_equal(v.numpy(), f[tag + '_' + name]), name
    n = dlpm_amd.DLPM(alpha, 'cpu', T, scale='scale_exploding', native_schedule=True)
    for name, v in zip(['g', 'bg', 's', 'bs'], n.host_schedule):
        # sigmas are differences of nearly equal fp32 powers accumulated in fp32 by the reference: the double
        # recurrence is the accurate side; agreement degrades towards t = 0 where sigma_t^alpha is ~1e-5 of the sum
        np.testing.assert_allclose(v.numpy(), f[tag + '_' + name], rtol=2e-3 if name == 's' else 1e-6, err_msg=name)
    # reference quirk: rescale_diffusion regenerates a scale_PRESERVING schedule
    d.rescale_diffusion(20)
    assert d.scale == 'scale_exploding' and float(d.host_schedule[0][5]) < 1.0
    m = dlpm_amd.GenerativeLevyProcess(alpha, 'cpu', T, scale='scale_exploding', input_scaling=True)
    assert torch.equal(m._input_scale(), 1 / (1 + m.dlpm.host_schedule[3]))
    assert dlpm_amd.GenerativeLevyProcess(alpha, 'cpu', T, input_scaling=True)._input_scale() is None


def test_header_is_plain_c(tmp_path):
    """include/dlpm_amd.h is the drop-in boundary: it must be consumable by a C compiler (C99, no C++-isms) and by a
    C++ one, and a C program must link against the library with it."""
    import subprocess
    src = tmp_path / 'use.c'
    src.write_text('#include "dlpm_amd.h"\n#include <stdio.h>\n'
                   'int main(void) { dlpm_update_args a; dlpm_sampler_config c; (void)a; (void)c;\n'
                   '  float g[8], bg[8], s[8], bs[8];\n'
                   '  if (dlpm_schedule_f32(8, 1.7, g, bg, s, bs) != DLPM_OK) return 2;\n'
                   '  printf("%d %.6f\\n", dlpm_abi_version(), bg[7]); return 0; }\n')
    inc = os.path.join(ROOT, 'include')
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-fsyntax-only', '-I', inc, str(src)])
    subprocess.check_call(['g++', '-std=c++17', '-Wall', '-fsyntax-only', '-x', 'c++', '-I', inc, str(src)])
    exe = tmp_path / 'use'
    libdir = os.path.join(ROOT, 'dlpm_amd', 'lib')
    subprocess.check_call(['gcc', '-std=c99', '-I', inc, str(src), '-o', str(exe), '-L', libdir, '-ldlpm_amd',
                           '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib'])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert int(out[0]) == _lib.ABI_VERSION
    d = dlpm_amd.DLPM(1.7, 'cpu', 8)
    assert abs(float(out[1]) - float(d.host_schedule[1][7])) < 1e-6


def test_product_never_imports_oracle():
    import subprocess, sys
    code = ("import sys; import dlpm_amd; "
            "bad=[m for m in sys.modules if m=='oracle' or m.startswith('oracle.')]; assert not bad, bad")
    subprocess.check_call([sys.executable, '-c', code], cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'dlpm_amd')):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.cpp', '.h')):
                src = open(os.path.join(dirpath, fn)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, fn


def test_unsupported_paths_fail_loudly():
    with pytest.raises(NotImplementedError):
        dlpm_amd.GenerativeLevyProcess(1.7, 'cpu', 10, LIM=True, rescale_timesteps=True, isotropic=False)
    with pytest.raises(AssertionError, match='Unknown scale'):
        dlpm_amd.GenerativeLevyProcess(1.7, 'cpu', 10, scale='scale_imploding')
    with pytest.raises(Exception, match='Wrong value of alpha'):
        dlpm_amd.DLPM(2.5, 'cpu', 10)
    m = dlpm_amd.GenerativeLevyProcess(1.7, 'cpu', 10, rescale_timesteps=False)
    with pytest.raises(AssertionError, match='Rescaling only works'):
        m.sample({'default': lambda x, t: x}, [1, 1, 2], 20)


def test_activation_arena_recycles_buffers():
    """dlpm_unet_workspace_bytes is a host-side dry run of the launch plan: with liveness-based reuse the CIFAR net's
    activations at B = 8192 (BASELINE config 4 on ONE GPU) must fit the 288 GB of an MI355X with room to spare, and
    the footprint must be well under the no-reuse sum (what dlpm_unet_keep_features(1) allocates)."""
    L = _lib.lib()
    c = UNETS['cifar']
    cfg = _lib.UNetConfig()
    cfg.in_channels, cfg.model_channels, cfg.out_channels = c['in_ch'], c['mc'], c['in_ch']
    cfg.num_res_blocks, cfg.num_heads, cfg.image_size = c['res'], c['heads'], 32
    cfg.n_mult, cfg.n_attn = len(c['mult']), len(c['attn'])
    for i, m in enumerate(c['mult']):
        cfg.channel_mult[i] = m
    for i, a in enumerate(c['attn']):
        cfg.attention_resolutions[i] = a
    h = C.c_void_p()
    _lib.check(L.dlpm_unet_create(C.byref(cfg), C.byref(h)))
    try:
        need = L.dlpm_unet_workspace_bytes(h, 8192)
        assert 0 < need < 200e9, need
        _lib.check(L.dlpm_unet_keep_features(h, 1))
        total = L.dlpm_unet_workspace_bytes(h, 8192)
        _lib.check(L.dlpm_unet_keep_features(h, 0))
        assert total > 2.5 * need, (need, total)
        assert L.dlpm_unet_workspace_bytes(h, 8192) == need            # deterministic
        assert abs(L.dlpm_unet_workspace_bytes(h, 1024) * 8 - need) < 0.01 * need
    finally:
        L.dlpm_unet_destroy(h)


def test_split_epilogue_ragged_tile_addresses_stay_inside_the_tensor():
    """conv_split.hip, split_store_from_registers<MASKED>: on a ragged last tile (B * HW not a multiple of 128) every lane's
    residual READ and output WRITE must land on a row below M, also for lanes whose first row already lies beyond it (round-2
    advisor finding: the clamp was relative to the lane's first row and read up to 48 rows behind the tensor).  The kernel's
    index arithmetic, restated: row0 = m0 + wm * 64 + 4 * kh, lane rows i * 32 + (r & 3) + 8 * (r >> 2); residual rows are
    clamped to min(row, rlim) >= 0 from the lane's clamped first row min(wm * 64 + 4 * kh, mrem - 1); writes are masked by row <= rlim."""
    for M in (16, 48, 64, 100, 127, 129, 192, 3 * 64 + 16, 5 * 16):
        m0 = (M // 128) * 128
        mrem = min(128, M - m0)
        if mrem == 128 or mrem <= 0:
            continue
        for wm in (0, 1):
            for kh in (0, 1):
                first = wm * 64 + 4 * kh
                rlim = mrem - 1 - first
                rres0 = min(first, mrem - 1)
                for i in (0, 1):
                    for r in range(16):
                        row = i * 32 + (r & 3) + 8 * (r >> 2)
                        read_row = m0 + rres0 + max(min(row, rlim), 0)
                        assert 0 <= read_row < M, (M, wm, kh, i, r, read_row)
                        if row <= rlim:
                            assert m0 + first + row < M


def test_declared_batch_semantics_of_the_host_classes():
    """Round 6: `GenerationManager.generate(models, n)` declares n as the nets' dispatch batch (UNetModel.declare_batch) unless their owner
    declared a policy explicitly; `declare_batch=False` (what EvaluationManager passes for the chunks of one evaluation) leaves it alone.
    Host logic only: no native handle is created."""
    net, _ = build_unet('tiny')
    assert net._conv_policy == (_lib.CONV_AUTO, 0)
    net.declare_batch(64)
    assert net._conv_policy == (_lib.CONV_AUTO, 64)
    net.declare_batch(8)                                   # the latest undeclared caller wins
    assert net._conv_policy == (_lib.CONV_AUTO, 8)
    net.set_conv_policy('f2', 1024)                        # an explicit declaration ...
    net.declare_batch(8)                                   # ... is never overridden
    assert net._conv_policy == (_lib.CONV_F2, 1024)

    class Method:
        device = 'cpu'

        def __init__(self):
            self.shapes = []

        def sample(self, shape, models, **kw):
            self.shapes.append(list(shape))
            raise RuntimeError('stop here')                # (the sampler itself needs the GPU)

    net2, _ = build_unet('tiny')
    m = Method()
    gm = dlpm_amd.GenerationManager(m, dlpm_amd.ShapeProbe([3, 16, 16]), True)
    with pytest.raises(RuntimeError, match='stop here'):
        gm.generate({'default': net2}, 5)
    assert net2._conv_policy == (_lib.CONV_AUTO, 5) and m.shapes == [[5, 3, 16, 16]]
    with pytest.raises(RuntimeError, match='stop here'):
        gm.generate({'default': net2}, 3, declare_batch=False)
    assert net2._conv_policy == (_lib.CONV_AUTO, 5)
