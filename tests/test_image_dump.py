"""Generated-image dump: quantisation kernel, native PNG writer and the chunk loop of
bem/evaluate/EvaluationManager.py:174-196, against oracle/images.py and independent PNG decoders."""
import ctypes as C
import io
import os

import numpy as np
import pytest
import torch

import dlpm_amd
from dlpm_amd import _lib
from oracle import images as oimg
from test_host_mirror import build_unet


def _encode(img, level=6):
    L = _lib.lib()
    H, W = img.shape[:2]
    cap = L.dlpm_png_bound(H, W)
    buf = np.empty(cap, np.uint8)
    n = C.c_int64()
    _lib.check(L.dlpm_png_encode_rgb8(img.ctypes.data, H, W, level, buf.ctypes.data, cap, C.byref(n)))
    return bytes(buf[:n.value])


def _pil_decode(data):
    from PIL import Image
    im = Image.open(io.BytesIO(data))
    assert im.mode == 'RGB'
    return np.asarray(im)


@pytest.mark.parametrize('H,W,kind', [(32, 32, 'noise'), (32, 32, 'smooth'), (64, 64, 'smooth'), (1, 1, 'noise'),
                                      (5, 7, 'noise'), (3, 200, 'flat')])
def test_png_encoder_round_trips_through_two_decoders(H, W, kind):
    g = np.random.default_rng(H * 1000 + W)
    if kind == 'noise':
        img = g.integers(0, 256, (H, W, 3), dtype=np.uint8)
    elif kind == 'flat':
        img = np.full((H, W, 3), 200, np.uint8)
    else:
        yy, xx = np.mgrid[0:H, 0:W]
        img = np.stack([(yy * 4) % 256, (xx * 3 + yy) % 256, (xx * yy) % 256], -1).astype(np.uint8)
    for level in (0, 1, 6, 9):
        data = _encode(np.ascontiguousarray(img), level)
        assert np.array_equal(oimg.png_decode_rgb8(data), img)
        assert np.array_equal(_pil_decode(data), img)
    if kind != 'noise' and H * W > 100:
        assert len(_encode(np.ascontiguousarray(img), 6)) < img.size // 2      # filtering + deflate do compress


def test_png_encoder_argument_errors():
    L = _lib.lib()
    img = np.zeros((4, 4, 3), np.uint8)
    out = np.zeros(16, np.uint8)
    n = C.c_int64()
    assert L.dlpm_png_encode_rgb8(img.ctypes.data, 4, 4, 6, out.ctypes.data, 16, C.byref(n)) == -1
    assert b'too small' in L.dlpm_last_error()
    assert L.dlpm_png_encode_rgb8(img.ctypes.data, 4, 4, 11, out.ctypes.data, 16, C.byref(n)) == -1
    assert L.dlpm_png_bound(0, 4) == -1


@pytest.mark.parametrize('threads', [1, 4])
def test_png_batch_writer_names_files_like_the_reference(tmp_path, threads):
    g = np.random.default_rng(5)
    batch = g.integers(0, 256, (37, 8, 8, 3), dtype=np.uint8)
    _lib.check(_lib.lib().dlpm_png_write_rgb8(batch.ctypes.data, 37, 8, 8, str(tmp_path).encode(), 100, 6, threads))
    assert sorted(os.listdir(tmp_path), key=lambda s: int(s[:-4])) == ['%d.png' % i for i in range(100, 137)]
    for i in (0, 17, 36):
        with open(tmp_path / ('%d.png' % (100 + i)), 'rb') as f:
            assert np.array_equal(oimg.png_decode_rgb8(f.read()), batch[i])
    rc = _lib.lib().dlpm_png_write_rgb8(batch.ctypes.data, 2, 8, 8, str(tmp_path / 'missing').encode(), 0, 6, threads)
    assert rc == -6 and b'cannot write' in _lib.lib().dlpm_last_error()


def test_oracle_quantisation_matches_torchvision_formula_known_answers():
    # hand-checked values of floor(clamp(x*255 + 0.5, 0, 255))
    x = torch.tensor([0.0, 1.0, 0.5, 0.0019607, 0.00196, 0.998, 1.5, -0.2, 127.5 / 255]).view(1, 1, 3, 3)
    want = np.array([0, 255, 128, 0, 0, 254, 255, 0, 128], np.uint8)
    got = oimg.to_rgb8(x)
    assert got.shape == (1, 3, 3, 3)
    assert np.array_equal(got[0, :, :, 0].ravel(), want) and np.array_equal(got[..., 0], got[..., 2])


def test_oracle_quantisation_and_decoder_against_pil_vectors():
    """tests/golden/f11_pil_quantise.npz was produced by PIL (the library torchvision.utils.save_image hands its array
    to; torchvision itself is absent from the build image): PIL's own float -> 8-bit conversion of 255 x + 0.5 at every
    rounding boundary, and a PIL-encoded PNG.  The oracle's quantisation and its PNG decoder must agree with both."""
    from conftest import golden
    f = golden('f11_pil_quantise')
    x = torch.from_numpy(f['x']).view(1, 1, 64, 64)
    got = oimg.to_rgb8(x)
    assert np.array_equal(got[0, :, :, 0], f['q']) and np.array_equal(got[0, :, :, 1], f['q'])
    assert np.array_equal(oimg.png_decode_rgb8(bytes(f['png_bytes'])), f['png_rgb'])


@pytest.mark.gpu
def test_images_to_rgb8_against_pil_vectors():
    from conftest import golden
    f = golden('f11_pil_quantise')
    xd = torch.from_numpy(f['x']).view(1, 1, 64, 64).cuda()
    out = torch.empty((1, 64, 64, 3), dtype=torch.uint8, device='cuda')
    _lib.check(_lib.lib().dlpm_images_to_rgb8(xd.data_ptr(), out.data_ptr(), 1, 1, 64, 64, _lib.stream_ptr()))
    o = out.cpu().numpy()
    assert np.array_equal(o[0, :, :, 0], f['q']) and np.array_equal(o[0, :, :, 2], f['q'])


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(5, 3, 32, 32), (3, 1, 32, 32), (2, 3, 64, 64), (1, 1, 1, 1), (7, 3, 5, 9)])
def test_images_to_rgb8_bit_exact(shape):
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.rand(shape, generator=g) * 1.2 - 0.1
    flat = x.view(-1)
    k = torch.arange(min(flat.numel(), 512))
    flat[:k.numel()] = (k.float() % 256 + 0.5) / 255          # exact rounding boundaries and their neighbours
    if flat.numel() > 1024:
        flat[512:1024] = torch.nextafter(flat[:512], torch.tensor(0.0))
    out = torch.empty((shape[0], shape[2], shape[3], 3), dtype=torch.uint8, device='cuda')
    xd = x.cuda()
    _lib.check(_lib.lib().dlpm_images_to_rgb8(xd.data_ptr(), out.data_ptr(), shape[0], shape[1], shape[2], shape[3],
                                             _lib.stream_ptr()))
    assert np.array_equal(out.cpu().numpy(), oimg.to_rgb8(x))


@pytest.mark.gpu
def test_images_to_rgb8_rejects_other_channel_counts():
    x = torch.zeros((1, 2, 4, 4), device='cuda')
    out = torch.empty((1, 4, 4, 3), dtype=torch.uint8, device='cuda')
    assert _lib.lib().dlpm_images_to_rgb8(x.data_ptr(), out.data_ptr(), 1, 2, 4, 4, _lib.stream_ptr()) == -1


def _read_dir(d, n):
    imgs = []
    for i in range(n):
        with open(os.path.join(d, '%d.png' % i), 'rb') as f:
            imgs.append(oimg.png_decode_rgb8(f.read()))
    return np.stack(imgs)


@pytest.mark.gpu
@pytest.mark.parametrize('overlap', [True, False])
def test_evaluation_manager_dumps_the_chunks_it_generates(tmp_path, overlap):
    net, _ = build_unet('tiny2')
    shape = [net.in_channels, 16, 16]
    method = dlpm_amd.GenerativeLevyProcess(1.7, 'cuda', 6, rescale_timesteps=True, seed=3)
    gm = dlpm_amd.GenerationManager(method, dlpm_amd.ShapeProbe(shape), True, reverse_steps=6, clamp_a=10, clamp_eps=50)
    ev = dlpm_amd.EvaluationManager(method, gm, None, verbose=False, is_image=True, gen_data_path=str(tmp_path / 'a'),
                                    overlap=overlap, data_to_generate=11, batch_size=4)
    r = ev.evaluate_model({'default': net})
    assert r['generated'] == 11 and sorted(os.listdir(tmp_path / 'a'), key=lambda s: int(s[:-4])) == ['%d.png' % i for i in range(11)]
    got = _read_dir(tmp_path / 'a', 11)
    # the same 11 samples in ONE chunk through the float path the reference takes (samples -> host -> quantise)
    method2 = dlpm_amd.GenerativeLevyProcess(1.7, 'cuda', 6, rescale_timesteps=True, seed=3)
    gm2 = dlpm_amd.GenerationManager(method2, dlpm_amd.ShapeProbe(shape), True, reverse_steps=6, clamp_a=10, clamp_eps=50)
    with method2.dataset_stream():
        want = oimg.to_rgb8(gm2.generate({'default': net}, 11, declare_batch=False))   # (the dump path declares no batch either)
    assert np.array_equal(got, want)                       # chunking-invariant and bit-identical pixels
    assert got.std() > 1                                   # not a constant image
    assert method.calls == 1                               # one dataset = one stream
    # GPU-sized chunks (device_batch) do not change a pixel either
    ev.gen_data_path, ev.device_batch = str(tmp_path / 'b'), 8
    method.calls = 0
    ev.evaluate_model({'default': net})
    assert np.array_equal(_read_dir(tmp_path / 'b', 11), want)


@pytest.mark.gpu
def test_evaluation_manager_one_channel_and_2d(tmp_path):
    net, _ = build_unet('mnist')
    method = dlpm_amd.GenerativeLevyProcess(1.7, 'cuda', 3, rescale_timesteps=True, seed=1)
    gm = dlpm_amd.GenerationManager(method, dlpm_amd.ShapeProbe([1, 32, 32]), True, reverse_steps=3)
    ev = dlpm_amd.EvaluationManager(method, gm, None, verbose=False, is_image=True, gen_data_path=str(tmp_path))
    ev.evaluate_model({'default': net}, data_to_generate=3, batch_size=2)
    got = _read_dir(tmp_path, 3)
    assert got.shape == (3, 32, 32, 3) and np.array_equal(got[..., 0], got[..., 1]) and np.array_equal(got[..., 0], got[..., 2])
    assert set(ev.evals) >= {'fid', 'precision', 'recall', 'density', 'coverage', 'wass', 'mmd', 'losses'}


def test_auto_device_batch_sizes_the_chunk_from_free_hbm(monkeypatch):
    """EvaluationManager(device_batch='auto') (the default): the largest power of two <= 1024 -- and not beyond the dump --
    whose workspaces fit in half of the free HBM.  No GPU needed: mem_get_info and the net's workspace are stubbed."""
    from dlpm_amd import evaluation as E

    class Net:
        def workspace_bytes(self, B, image_size):
            return B * 5 * 2 ** 20                      # 5 MiB per image (the CIFAR net's is 4.8)

    monkeypatch.setattr(torch.cuda, 'mem_get_info', lambda *a: (280 * 2 ** 30, 288 * 2 ** 30))
    assert E.auto_device_batch({'default': Net()}, [3, 32, 32], 5000) == 1024          # a free MI355X
    assert E.auto_device_batch({'default': Net()}, [3, 32, 32], 300) == 256            # never beyond the dump
    assert E.auto_device_batch({'default': Net()}, [3, 32, 32], 1) == 1
    monkeypatch.setattr(torch.cuda, 'mem_get_info', lambda *a: (3 * 2 ** 30, 288 * 2 ** 30))
    assert E.auto_device_batch({'default': Net()}, [3, 32, 32], 5000) == 256           # 1.5 GiB to spend: 256 x 5 MiB fits
    assert E.auto_device_batch({'default': object()}, [3, 32, 32], 5000) == 1024       # unknown nets: 64 x the state per image
    ev = E.EvaluationManager(None, None, None, is_image=True)
    assert ev.device_batch == 'auto'
