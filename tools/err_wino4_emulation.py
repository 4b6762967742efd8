#!/usr/bin/env python3
"""Where does the F(4x4,3x3) kernel's error come from?  NumPy emulation of k_conv3x3_wino4's arithmetic (CPU only, ~1 min).

256 input channels, SiLU-shaped activations, one 4x4 output tile per sample, against the float64 direct convolution.  Each stage can be
switched to float64 on its own: the input transform V = B^T d B, the channel accumulation M = sum_c V_c U_c (sequential, one fp32
rounding per product as the pessimistic model of the MFMA chain), the output transform Y = A^T M A.  U = G g G^T is computed in float64
and rounded once, as the kernel's relayout does.  Also: accumulation in 4 / 16 independent chains (what a second accumulator set
would buy), and inputs with their mean removed.

    python tools/err_wino4_emulation.py > profiles/r03/err_wino4_emulation.txt"""
import numpy as np

rng = np.random.default_rng(0)
BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)
Cin, Cout, N = 256, 16, 64
z = rng.standard_normal((N, Cin, 6, 6))
g = rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)
U32 = np.einsum('ia,ocab,jb->ocij', G, g, G).astype(np.float32)


def direct(d):
    out = np.zeros((N, Cout, 4, 4))
    for i in range(4):
        for j in range(4):
            out[:, :, i, j] = np.einsum('ncab,ocab->no', d[:, :, i:i + 3, j:j + 3], g)
    return out


def seq_transform(M, T, n_out):       # fp32, one rounding per term, rows then columns
    T32 = T.astype(np.float32)
    Z = np.zeros(M.shape[:2] + (n_out, M.shape[3]), np.float32)
    for i in range(n_out):
        acc = np.zeros(M.shape[:2] + (M.shape[3],), np.float32)
        for a in range(T.shape[1]):
            if T32[i, a] != 0:
                acc = (acc + T32[i, a] * M[:, :, a, :]).astype(np.float32)
        Z[:, :, i, :] = acc
    Y = np.zeros(M.shape[:2] + (n_out, n_out), np.float32)
    for j in range(n_out):
        acc = np.zeros(M.shape[:2] + (n_out,), np.float32)
        for b in range(T.shape[1]):
            if T32[j, b] != 0:
                acc = (acc + T32[j, b] * Z[:, :, :, b]).astype(np.float32)
        Y[:, :, :, j] = acc
    return Y


def wino(d, in_dt, acc_dt, out_dt, chains=1):
    d32 = d.astype(np.float32)
    V = (np.einsum('ia,ncab,jb->ncij', BT, d32.astype(np.float64), BT) if in_dt == np.float64 else seq_transform(d32, BT, 6)).astype(np.float32)
    parts = []
    cs = Cin // chains
    for k in range(chains):
        M = np.zeros((N, Cout, 6, 6), acc_dt)
        for c in range(k * cs, (k + 1) * cs):
            M = (M + V[:, None, c].astype(acc_dt) * U32[None, :, c].astype(acc_dt)).astype(acc_dt)
        parts.append(M)
    M = parts[0]
    for m in parts[1:]:
        M = (M + m).astype(acc_dt)
    M = M.astype(np.float32)
    return np.einsum('ia,noab,jb->noij', AT, M.astype(np.float64), AT) if out_dt == np.float64 else seq_transform(M, AT, 4)


def report(tag, Y, ref):
    print('%-58s max %.3g   rms %.3g' % (tag, np.abs(Y - ref).max(), np.sqrt(((Y - ref) ** 2).mean())))


f32, f64 = np.float32, np.float64
d = z / (1 + np.exp(-z))
ref = direct(d.astype(f32).astype(f64))
print('F(4x4,3x3), %d channels, silu(N(0,1)) inputs, |Y| rms %.3f; error against the float64 direct convolution' % (Cin, np.sqrt((ref ** 2).mean())))
report('everything fp32', wino(d, f32, f32, f32), ref)
report('input transform in fp64', wino(d, f64, f32, f32), ref)
report('output transform in fp64', wino(d, f32, f32, f64), ref)
report('both transforms in fp64', wino(d, f64, f32, f64), ref)
report('channel accumulation in fp64 (transforms fp32)', wino(d, f32, f64, f32), ref)
report('everything but the rounding of U in fp64', wino(d, f64, f64, f64), ref)
report('fp32, accumulation in 4 chains', wino(d, f64, f32, f64, 4), ref)
report('fp32, accumulation in 16 chains', wino(d, f64, f32, f64, 16), ref)
acc = np.zeros((N, Cout, 4, 4), f32)
d32, g32 = d.astype(f32), g.astype(f32)
for c in range(Cin):
    for a in range(3):
        for b in range(3):
            acc = (acc + d32[:, None, c, a:a + 4, b:b + 4] * g32[None, :, c, a, b, None, None]).astype(f32)
report('direct convolution, fp32, sequential', acc, ref)
d0 = d - 0.2066
report('everything fp32, inputs with their mean removed', wino(d0, f32, f32, f32), direct(d0.astype(f32).astype(f64)))
