"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the image dump the reference performs per generated sample.

bem/evaluate/EvaluationManager.py:188-190 calls torchvision.utils.save_image(samples[i], "<i>.png") on the
[0,1] CPU tensors GenerationManager leaves in `.samples`.  torchvision is a third-party dependency that is NOT
vendored under /root/reference, is not pinned by it (bem/requirements.txt lists a bare `torchvision`) and is not
installed in this image, so this restates its published algorithm (torchvision/utils.py, unchanged across
0.9 ... 0.2x): make_grid turns a single 1-channel image into 3 equal channels, then
    ndarr = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to("cpu", torch.uint8).numpy()
is handed to PIL.  Pinning: torchvision cannot produce vectors here, so the anchor is the library it delegates to --
tests/golden/f11_pil_quantise.npz holds PIL's own float -> 8-bit conversion (mode "F" -> "L") of 255 x + 0.5 at every
rounding boundary plus a PIL-encoded PNG (tools/make_fixtures.py f11); to_rgb8 and png_decode_rgb8 are checked against
both.  What stays unpinned by any reference run: that save_image is this op sequence (its published source).
"""
import struct
import zlib

import numpy as np
import torch


def to_rgb8(samples01):
    """[n,C,H,W] float tensor in [0,1] -> [n,H,W,3] uint8, per-sample save_image quantisation."""
    x = samples01.detach().to('cpu', torch.float32)
    if x.shape[1] == 1:
        x = torch.cat((x, x, x), 1)
    return x.mul(255).add_(0.5).clamp_(0, 255).permute(0, 2, 3, 1).to(torch.uint8).numpy()


def png_decode_rgb8(data):
    """Minimal PNG reader (8-bit truecolour, non-interlaced; all five filter types) -> [H,W,3] uint8.
    Checks the signature and every chunk CRC."""
    assert data[:8] == b'\x89PNG\r\n\x1a\n', 'bad signature'
    pos, idat, hdr = 8, b'', None
    while pos < len(data):
        n, typ = struct.unpack('>I4s', data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack('>I', data[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(typ + body) & 0xFFFFFFFF == crc, 'bad CRC in %s' % typ
        if typ == b'IHDR':
            hdr = struct.unpack('>IIBBBBB', body)
        elif typ == b'IDAT':
            idat += body
        elif typ == b'IEND':
            break
        pos += 12 + n
    W, H, depth, ctype, comp, flt, inter = hdr
    assert (depth, ctype, comp, flt, inter) == (8, 2, 0, 0, 0), hdr
    raw = zlib.decompress(idat)
    rb = 3 * W
    assert len(raw) == H * (rb + 1)
    out = np.zeros((H, rb), np.uint8)
    prev = np.zeros(rb, np.int32)
    for y in range(H):
        t = raw[y * (rb + 1)]
        line = np.frombuffer(raw, np.uint8, rb, y * (rb + 1) + 1).astype(np.int32)
        cur = np.zeros(rb, np.int32)
        for i in range(rb):
            a = cur[i - 3] if i >= 3 else 0
            b = prev[i]
            c = prev[i - 3] if i >= 3 else 0
            if t == 0:
                p = 0
            elif t == 1:
                p = a
            elif t == 2:
                p = b
            elif t == 3:
                p = (a + b) >> 1
            else:
                q = a + b - c
                pa, pb, pc = abs(q - a), abs(q - b), abs(q - c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            cur[i] = (line[i] + p) & 255
        out[y] = cur
        prev = cur
    return out.reshape(H, W, 3)
