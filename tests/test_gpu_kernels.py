"""GPU parity tests of the hand-written spliced-GEMM kernel against a plain PyTorch fp32 reference of the
same op, called through the C ABI (xv_kernel_tdnn_gemm).  Floating point: tolerances are stated per mode."""
import ctypes

import numpy as np
import pytest

from helpers import pkg as _pkg

pytestmark = pytest.mark.gpu

HALO = 32   # the engine keeps 32 zero rows around every plane (kHalo)


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


F16_MODES = (2, 3, 4)          # fp16, fp16x3, fp16x2
X_SPLIT = (0, 3)               # modes whose activations carry a residual plane
W_SPLIT = (0, 3, 4)            # modes whose weights carry one (fp16x2: fp16 activations x split weights)


def _split(t, prec, torch, split):
    """fp32 tensor -> (hi, lo) 16-bit planes the way the kernels define them."""
    dt = torch.float16 if prec in F16_MODES else torch.bfloat16
    hi = t.to(dt)
    if not split:
        return hi, None
    return hi, (t - hi.float()).to(dt)


def _planes_value(hi, lo):
    return hi.float() + (lo.float() if lo is not None else 0)


def _run_case(prec, epi, rows, n_pad, segs, relu, bn, seed=0, m_valid=None):
    """segs: list of (source index, ld, row_shift, k_len).  Returns (kernel output, reference output)."""
    torch = _torch()
    P = _pkg()
    g = torch.Generator(device="cpu").manual_seed(seed)
    dev = torch.device("cuda:0")
    nsrc = max(s[0] for s in segs) + 1
    src_ld = {}
    for s in segs:
        src_ld[s[0]] = max(src_ld.get(s[0], 0), s[1])
    # split-fp16 residuals are 2^-12 of the value: operands are scaled up so that they stay fp16 normals (the engine
    # scales its packed weights the same way, PackModel)
    amp = 16.0 if prec in (3, 4) else 1.0
    X = [(torch.randn(rows + 2 * HALO, src_ld[i], generator=g) * amp).to(dev) for i in range(nsrc)]
    K = sum(s[3] for s in segs)
    W = (torch.randn(n_pad, K, generator=g) * (amp * amp / np.sqrt(K))).to(dev)
    bias = (torch.randn(n_pad, generator=g) * 0.1).to(dev)
    scale = (torch.rand(n_pad, generator=g) + 0.5).to(dev)
    offset = (torch.randn(n_pad, generator=g) * 0.1).to(dev)
    Xp = [_split(x, prec, torch, prec in X_SPLIT) for x in X]
    Wp = _split(W, prec, torch, prec in W_SPLIT)

    d = P.GemmDesc()
    d.precision, d.epilogue, d.nseg = prec, epi, len(segs)
    for j, (si, ld, shift, klen) in enumerate(segs):
        hi, lo = Xp[si]
        esz = 2
        d.seg[j].hi = hi.data_ptr() + HALO * src_ld[si] * esz
        d.seg[j].lo = (lo.data_ptr() + HALO * src_ld[si] * esz) if lo is not None else None
        d.seg[j].ld, d.seg[j].row_shift, d.seg[j].k_len = src_ld[si], shift, klen
    d.w_hi, d.w_lo, d.ldw = Wp[0].data_ptr(), (Wp[1].data_ptr() if Wp[1] is not None else None), K
    d.rows, d.n_pad = rows, n_pad
    d.bias, d.scale, d.offset = bias.data_ptr(), scale.data_ptr(), offset.data_ptr()
    d.relu, d.bn = int(relu), int(bn)
    d.hip_stream = None

    # reference on the values the kernel actually sees (quantised planes), accumulated in fp32/fp64
    Xq = [_planes_value(*p).double() for p in Xp]
    Wq = _planes_value(*Wp).double()
    z = torch.zeros(rows, n_pad, dtype=torch.float64, device=dev)
    k0 = 0
    for (si, ld, shift, klen) in segs:
        xs = Xq[si][HALO + shift: HALO + shift + rows, :klen]
        z += xs @ Wq[:, k0:k0 + klen].T
        k0 += klen
    z = z + bias.double()
    if relu:
        z = torch.clamp(z, min=0)
    if bn:
        z = z * scale.double() + offset.double()

    torch.cuda.synchronize()
    if epi == P.EPI_ACT:
        oh = torch.zeros(rows, n_pad, dtype=torch.float16 if prec in F16_MODES else torch.bfloat16, device=dev)
        ol = torch.zeros_like(oh)
        d.out_hi, d.out_lo, d.ldo = oh.data_ptr(), (ol.data_ptr() if prec in X_SPLIT else None), n_pad
        P.kernel_tdnn_gemm(d)
        torch.cuda.synchronize()
        return _planes_value(oh, ol if prec in X_SPLIT else None).double().cpu().numpy(), z.cpu().numpy()
    if epi == P.EPI_F32:
        of = torch.full((rows, n_pad), -7.0, dtype=torch.float32, device=dev)
        d.out_f32, d.ldf, d.m_valid = of.data_ptr(), n_pad, rows if m_valid is None else m_valid
        P.kernel_tdnn_gemm(d)
        torch.cuda.synchronize()
        return of.double().cpu().numpy(), z.cpu().numpy()
    # stats
    ngrp = rows // 16
    rng = np.random.default_rng(seed)
    first = rng.integers(0, 10, ngrp).astype(np.int8)
    last = np.minimum(16, first + rng.integers(0, 17, ngrp)).astype(np.int8)
    gr = torch.from_numpy(np.stack([first, last], 1).copy()).to(dev)
    part = torch.zeros(ngrp, 2, n_pad, dtype=torch.float32, device=dev)
    d.partial, d.ldp, d.grp_range = part.data_ptr(), n_pad, gr.data_ptr()
    P.kernel_tdnn_gemm(d)
    torch.cuda.synchronize()
    zz = z.cpu().numpy().reshape(ngrp, 16, n_pad)
    mask = (np.arange(16)[None, :] >= first[:, None]) & (np.arange(16)[None, :] < last[:, None])
    ref = np.stack([(zz * mask[:, :, None]).sum(1), (zz * zz * mask[:, :, None]).sum(1)], 1)
    return part.double().cpu().numpy(), ref


TDNN2 = [(0, 512, -2, 512), (0, 512, 0, 512), (0, 512, 2, 512)]          # tdnn2 of run_xvector_new.sh:96
TDNN1 = [(0, 32, o, 32) for o in (-2, -1, 0, 1, 2)]                        # tdnn1, 23-dim input padded to 32
CVEC5 = [(0, 512, 0, 512), (1, 128, 0, 128)]                               # tdnn5_xvec: Append(tdnn4_xvec, tdnn5)
AM5 = [(0, 768, -6, 672), (0, 768, -3, 672), (0, 768, 0, 672)]             # AM tdnn5 (650 wide, padded)

# tolerance on max|err| / max|ref| : the reference already uses the quantised operands, so what is left is
# the dropped lo*lo term (split mode, ~2^-16 relative per product) and fp32 accumulation order.
TOL = {0: 2e-5, 1: 2e-5, 2: 2e-5, 3: 2e-5, 4: 2e-5}
OUT_Q = {0: 2.0 ** -16, 1: 2.0 ** -8, 2: 2.0 ** -10, 3: 2.0 ** -16, 4: 2.0 ** -10}   # re-quantising the OUTPUT planes
ALL_PREC = [0, 1, 2, 3, 4]


@pytest.mark.parametrize("prec", ALL_PREC)
@pytest.mark.parametrize("segs", [TDNN2, TDNN1, CVEC5, AM5], ids=["tdnn2", "tdnn1", "cvec5", "am5"])
def test_gemm_f32_epilogue(prec, segs):
    out, ref = _run_case(prec, 1, 256, 256, segs, relu=True, bn=True)
    err = np.abs(out - ref).max() / np.abs(ref).max()
    assert err < TOL[prec], err


@pytest.mark.parametrize("prec", ALL_PREC)
def test_gemm_act_epilogue(prec):
    out, ref = _run_case(prec, 0, 384, 512, TDNN2, relu=True, bn=True, seed=1)
    err = np.abs(out - ref).max() / np.abs(ref).max()
    assert err < TOL[prec] + OUT_Q[prec], err


@pytest.mark.parametrize("relu,bn", [(False, False), (True, False), (False, True)])
def test_gemm_epilogue_flags(relu, bn):
    out, ref = _run_case(0, 1, 128, 128, [(0, 64, 0, 64)], relu=relu, bn=bn, seed=2)
    assert np.abs(out - ref).max() / np.abs(ref).max() < 2e-5


def test_gemm_m_valid_rows_untouched():
    out, ref = _run_case(0, 1, 128, 128, [(0, 64, 0, 64)], relu=False, bn=False, seed=3, m_valid=70)
    assert np.abs(out[:70] - ref[:70]).max() / np.abs(ref).max() < 2e-5
    assert np.all(out[70:] == -7.0)


@pytest.mark.parametrize("prec", ALL_PREC)
def test_gemm_stats_epilogue(prec):
    out, ref = _run_case(prec, 2, 256, 1536, [(0, 512, 0, 512)], relu=True, bn=True, seed=4)
    scale = np.abs(ref).max(axis=(0, 2), keepdims=True)
    assert (np.abs(out - ref) / scale).max() < 3e-5


@pytest.mark.parametrize("prec", [0, 1, 3, 4])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_even_tile_count_uses_256_row_variant(prec, epi):
    # 22 row tiles of 128 -> 11 tiles of 256 (not a multiple of 8 either); three K segments with shifts
    out, ref = _run_case(prec, epi, 22 * 128, 256, [(0, 64, -3, 64), (0, 64, 0, 64), (1, 32, 3, 32)], relu=True, bn=True,
                         seed=6)
    if epi == 2:
        scale = np.abs(ref).max(axis=(0, 2), keepdims=True)
        assert (np.abs(out - ref) / scale).max() < 3e-5
    else:
        assert np.abs(out - ref).max() / np.abs(ref).max() < TOL[prec] + (OUT_Q[prec] if epi == 0 else 0)


def test_gemm_many_tiles_xcd_mapping():
    # 21 row tiles (not a multiple of 8) x 3 column tiles: exercises the XCD-aware block -> tile map
    out, ref = _run_case(1, 1, 21 * 128, 384, [(0, 64, -1, 64), (0, 64, 1, 64)], relu=True, bn=False, seed=5)
    assert np.abs(out - ref).max() / np.abs(ref).max() < 2e-5


@pytest.mark.parametrize("prec", [1, 3, 4])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_stream_k_variant(prec, epi):
    # 66 x 512 rows x 4 column tiles = 264 tiles of 512 x 128 (528 of 256 x 128) for 256 workgroups: every workgroup
    # gets a head part, whole tiles and a tail part; three K segments with time offsets (partial starts inside a group)
    out, ref = _run_case(prec, epi, 66 * 512, 512, [(0, 64, -3, 64), (0, 64, 0, 64), (0, 64, 3, 64), (1, 32, 1, 32)],
                         relu=True, bn=True, seed=8)
    if epi == 2:
        scale = np.abs(ref).max(axis=(0, 2), keepdims=True)
        assert (np.abs(out - ref) / scale).max() < 3e-5
    else:
        assert np.abs(out - ref).max() / np.abs(ref).max() < TOL[prec] + (OUT_Q[prec] if epi == 0 else 0)
