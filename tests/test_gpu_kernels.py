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


F16_MODES = (2, 3, 4, 8)       # fp16, fp16x3, fp16x2, fp16x3e
X_SPLIT = (0, 3, 8)            # modes whose activations carry a residual plane
W_SPLIT = (0, 3, 4, 8)         # modes whose weights carry one (fp16x2: fp16 activations x split weights)
E2M1_GRID = np.array([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])


def _decode_lo4(lo4, lo4s, n_cols):
    """4-bit residual plane [rows, n_cols / 2] + E8M0 scales [rows, n_cols / 64] -> float64 [rows, n_cols]"""
    b = lo4.cpu().numpy()
    nib = np.stack([b & 15, b >> 4], axis=-1).reshape(b.shape[0], n_cols)
    val = np.where(nib & 8, -1.0, 1.0) * E2M1_GRID[nib & 7]
    sc = 2.0 ** (lo4s.cpu().numpy()[:, :n_cols // 64].astype(np.float64) - 127)
    return val * np.repeat(sc, 64, axis=1)


def _split(t, prec, torch, split):
    """fp32 tensor -> (hi, lo) 16-bit planes the way the kernels define them."""
    dt = torch.float16 if prec in F16_MODES else torch.bfloat16
    hi = t.to(dt)
    if not split:
        return hi, None
    return hi, (t - hi.float()).to(dt)


def _planes_value(hi, lo):
    return hi.float() + (lo.float() if lo is not None else 0)


def _run_case(prec, epi, rows, n_pad, segs, relu, bn, seed=0, m_valid=None, p8=0):
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
    amp = 16.0 if prec in (3, 4, 8) else 1.0
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
    d.p8 = p8

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
        if prec == 8:   # fp16 plane + 4-bit image of what its rounding dropped
            o4 = torch.zeros(rows, n_pad // 2, dtype=torch.uint8, device=dev)
            o4s = torch.zeros(rows, (n_pad // 64 + 3) // 4 * 4, dtype=torch.uint8, device=dev)
            d.out_lo, d.out_lo4, d.out_lo4_scale = None, o4.data_ptr(), o4s.data_ptr()
            P.kernel_tdnn_gemm(d)
            torch.cuda.synchronize()
            return oh.double().cpu().numpy() + _decode_lo4(o4, o4s, n_pad), z.cpu().numpy()
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


@pytest.mark.parametrize("segs", [TDNN2, TDNN1], ids=["tdnn2", "tdnn1"])
def test_gemm_planes_epilogue_with_4bit_residual(segs):
    """XV_PREC_FP16X3E: the planes epilogue writes the fp16 plane and the e2m1 image of its rounding residual (one scale
    per row and 64 columns, 2^-13 of the binade of the block's largest value): together 2-3 bits better than the fp16
    plane alone - worst case half a grid step (0.5 between 2 and 4) at that scale."""
    out, ref = _run_case(8, 0, 384, 512, segs, relu=True, bn=True, seed=1)
    err = np.abs(out - ref)
    blk = np.abs(ref).reshape(ref.shape[0], -1, 64).max(axis=2, keepdims=True)        # per (row, 64 columns)
    rel = (err.reshape(ref.shape[0], -1, 64) / np.maximum(blk, 1e-30)).max()
    assert rel < 2.0 ** -13 * 1.05 + TOL[3], rel
    assert np.sqrt((err ** 2).mean()) < 0.25 * 2.0 ** -12 * np.sqrt((ref ** 2).mean())   # fp16 alone: ~0.29 * 2^-11


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


# ---------------------------------------------------------------------------------------------------------------------
# XV_PREC_FP16MX: fp16 product + block-scaled 4-bit residual product.  The reference below restates the arithmetic
# exactly (what is quantised how, with which scale), so the tolerance is again fp32 accumulation order only.
E2M1 = [0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0]


def _q_e2m1(t, torch):
    """round to nearest (ties to even code) onto the e2m1 grid, saturating at 6 - v_cvt_scalef32_pk_fp4_f16"""
    a = t.abs()
    idx = ((a > 0.25).long() + (a >= 0.75).long() + (a > 1.25).long() + (a >= 1.75).long() + (a > 2.5).long() +
           (a >= 3.5).long() + (a > 5.0).long())
    grid = torch.tensor(E2M1, dtype=t.dtype, device=t.device)
    return torch.sign(t) * grid[idx]


def _lo4_plane(x32, xh, torch):
    """What the planes epilogue of XV_PREC_FP16MX2 / FP16X3E writes next to the fp16 plane xh of the fp32 values x32: the
    e2m1 codes of r = x32 - xh with one scale 2^(E - 13) per row and 64 columns (E = exponent of the block's largest
    |x32|: every residual then lies in [-4, 4]), packed two per byte, the E8M0 scale bytes, and the decoded values."""
    r = x32.double() - xh.double()
    rows, n = r.shape
    b = r.reshape(rows, n // 64, 64)
    m = x32.float().abs().reshape(rows, n // 64, 64).amax(dim=2, keepdim=True)
    ebits = (m.view(torch.int32) >> 23) & 255
    e8 = torch.where(ebits < 113, torch.full_like(ebits, 100), torch.where(ebits > 254, torch.full_like(ebits, 241), ebits - 13))
    sc = torch.pow(torch.tensor(2.0, dtype=torch.float64, device=r.device), e8.double() - 127)
    q = _q_e2m1(b / sc, torch)
    grid = torch.tensor(E2M1, dtype=torch.float64, device=r.device)
    code = (q.abs()[..., None] == grid).long().argmax(dim=-1) | ((b < 0).long() << 3)
    code = code.reshape(rows, n)
    packed = (code[:, 0::2] | (code[:, 1::2] << 4)).to(torch.uint8).contiguous()
    pitch = (n // 64 + 3) // 4 * 4        # scale rows are padded to whole dwords
    e8p = torch.zeros(rows, pitch, dtype=torch.uint8, device=r.device)
    e8p[:, :n // 64] = e8.reshape(rows, n // 64).to(torch.uint8)
    return packed, e8p, (q * sc).reshape(rows, n)


def _w4_image(W, torch):
    """4-bit image of the weights for the second walk: per row and 32 consecutive columns the smallest power of two 2^E
    with max |w| / 2^E <= 6, codes to nearest even - what xv_pack_mx_weights packs (values only)"""
    n, K = W.shape
    b = W.double().reshape(n, K // 32, 32)
    m = b.abs().amax(dim=2, keepdim=True)
    E = torch.ceil(torch.log2(torch.clamp(m, min=1e-300) / 6.0))
    sc = torch.pow(torch.tensor(2.0, dtype=torch.float64, device=W.device), E)
    return (_q_e2m1(b / sc, torch) * sc).reshape(n, K)


def _gmax_bits(plane_abs_max, torch):
    """float32 bit patterns of the group maxima, as the producing epilogue records them"""
    return plane_abs_max.float().view(torch.int32)


def _run_mx_case(epi, rows, n_pad, segs, seed=0, variant_rows=None, prec=6, p8=0):
    torch = _torch()
    P = _pkg()
    g = torch.Generator(device="cpu").manual_seed(seed)
    dev = torch.device("cuda:0")
    nsrc = max(s[0] for s in segs) + 1
    src_ld = {}
    for s in segs:
        src_ld[s[0]] = max(src_ld.get(s[0], 0), s[1])
    amp = 16.0
    # activations with a different magnitude in every 16-row group (the per-group scales matter), all of ld columns
    X = []
    for i in range(nsrc):
        x = torch.randn(rows + 2 * HALO, src_ld[i], generator=g) * amp
        gs = torch.exp(torch.randn((rows + 2 * HALO) // 16, generator=g).clamp(-2, 2) * 1.5).repeat_interleave(16)[:, None]
        X.append((x * gs).to(dev))
    K = sum(s[3] for s in segs)
    W = (torch.randn(n_pad, K, generator=g) * (amp * amp / np.sqrt(K))).to(dev)
    bias = (torch.randn(n_pad, generator=g) * 0.1).to(dev)
    scale = ((torch.rand(n_pad, generator=g) + 0.5) / 256).to(dev)   # keeps the fp16 output plane far from its range limit
    offset = (torch.randn(n_pad, generator=g) * 0.1).to(dev)
    Xh = [x.to(torch.float16) for x in X]
    Wh = W.to(torch.float16)
    # residual plane + row scales from the library's own packer (the same code xv_model_pack runs)
    w4, w4s = P.pack_mx_residual(W.cpu().numpy(), Wh.cpu().view(torch.int16).numpy().view(np.uint16),
                                 [(s[0], s[2], s[3]) for s in segs], walk64=bool(p8))
    w4_d, w4s_d = torch.from_numpy(w4).to(dev), torch.from_numpy(w4s).to(dev)
    w4t_d = torch.from_numpy(P.tile_mx_scales(w4s, epi)).to(dev)    # the same scales in the kernels' staging order
    # group maxima of the source planes (rows relative to logical row 0)
    gmax = []
    for i in range(nsrc):
        body = Xh[i][HALO:HALO + rows].float().abs().reshape(rows // 16, 16, -1).amax(dim=(1, 2))
        gmax.append(_gmax_bits(body, torch).contiguous())

    lo4 = [_lo4_plane(X[i], Xh[i], torch) for i in range(nsrc)] if prec == 7 else None
    d = P.GemmDesc()
    d.precision, d.epilogue, d.nseg = prec, epi, len(segs)
    for j, (si, ld, shift, klen) in enumerate(segs):
        d.seg[j].hi = Xh[si].data_ptr() + HALO * src_ld[si] * 2
        d.seg[j].lo = None
        d.seg[j].ld, d.seg[j].row_shift, d.seg[j].k_len = src_ld[si], shift, klen
        d.seg[j].gmax = gmax[si].data_ptr()
        if prec == 7:
            d.seg[j].lo4 = lo4[si][0].data_ptr() + HALO * (src_ld[si] // 2)
            d.seg[j].lo4_scale = lo4[si][1].data_ptr() + HALO * lo4[si][1].shape[1]
    if prec == 7:
        w4b, w4bs = P.pack_mx_weights(W.cpu().numpy(), [(s[0], s[2], s[3]) for s in segs], walk64=bool(p8))
        w4b_d = torch.from_numpy(w4b).to(dev)
        w4bs_d = torch.from_numpy(P.tile_mx_scales(w4bs, epi)).to(dev)
        d.w4b, d.ldw4b, d.w4b_scale = w4b_d.data_ptr(), (2 * K if p8 else K // 2), w4bs_d.data_ptr()
    d.w_hi, d.w_lo, d.ldw = Wh.data_ptr(), None, K
    d.w4, d.ldw4, d.w4_scale = w4_d.data_ptr(), K // 128 * 64, w4t_d.data_ptr()
    run_rows = variant_rows or rows     # variant_rows: launch only the first rows of the same data (another tile partition)
    d.rows, d.n_pad = run_rows, n_pad
    d.bias, d.scale, d.offset = bias.data_ptr(), scale.data_ptr(), offset.data_ptr()
    d.relu, d.bn = 1, 1
    d.hip_stream = None
    d.p8 = p8

    # ---- reference: fp16 x . fp16 w_hi + q4(x / sx) sx . q4((w - w_hi) / sw) sw, in fp64
    Whd = Wh.double()
    R = W.double() - Whd
    # scale of weight element (n, k): byte 4 * block + lane group of row n, where k is column c of K step `step` of the
    # walk (consecutive segments over one source = one group, walked chunk by chunk, offset by offset), block = step // 4
    # and lane group = (c % 32) // 8
    step_of_col = np.empty(K, dtype=np.int64)
    k0 = base = j = 0
    while j < len(segs):
        ns = 1
        while j + ns < len(segs) and segs[j + ns][0] == segs[j][0] and segs[j + ns][3] == segs[j][3]:
            ns += 1
        klen = segs[j][3]
        for jj in range(ns):
            c = np.arange(klen)
            if p8:   # tdnn_gemm_kernel_p8: 64-column chunk -> offset -> the chunk's two halves
                step_of_col[k0 + jj * klen + c] = base + ((c // 64) * ns + jj) * 2 + (c % 64) // 32
            else:
                step_of_col[k0 + jj * klen + c] = base + (c // 32) * ns + jj
        base += ns * klen // 32
        k0 += ns * klen
        j += ns
    sidx = torch.from_numpy(4 * (step_of_col // 4) + (np.arange(K) % 32) // 8).to(dev)
    e8 = w4s_d.double()[:, sidx]
    sw = torch.pow(torch.tensor(2.0, dtype=torch.float64, device=dev), e8 - 127)
    Rq = _q_e2m1(R / sw, torch) * sw
    W4q = _w4_image(W, torch) if prec == 7 else None
    z = torch.zeros(rows, n_pad, dtype=torch.float64, device=dev)
    k0 = 0
    for (si, ld, shift, klen) in segs:
        xs = Xh[si][HALO + shift: HALO + shift + rows, :klen].double()
        # scale of OUTPUT row r = 2^(e - 2 - 127), e = exponent field of the maximum of r's 16-row group of the source plane
        ebits = ((gmax[si] >> 23) & 255).clamp(16, 200).double().repeat_interleave(16)[:, None]
        sx = torch.pow(torch.tensor(2.0, dtype=torch.float64, device=dev), ebits - 2 - 127)
        x4 = _q_e2m1(xs / sx, torch) * sx
        z += xs @ Whd[:, k0:k0 + klen].T + x4 @ Rq[:, k0:k0 + klen].T
        if prec == 7:   # + q4(x - fp16(x)) . q4(w)
            z += lo4[si][2][HALO + shift: HALO + shift + rows, :klen] @ W4q[:, k0:k0 + klen].T
        k0 += klen
    z = torch.clamp(z + bias.double(), min=0) * scale.double() + offset.double()

    torch.cuda.synchronize()
    if epi == P.EPI_ACT:
        oh = torch.zeros(rows, n_pad, dtype=torch.float16, device=dev)
        gm_out = torch.zeros(rows // 16, dtype=torch.int32, device=dev)
        d.out_hi, d.out_lo, d.ldo = oh.data_ptr(), None, n_pad
        d.gmax_out = gm_out.data_ptr()
        if prec in (7, 9):
            o4 = torch.zeros(rows, n_pad // 2, dtype=torch.uint8, device=dev)
            o4s = torch.zeros(rows, (n_pad // 64 + 3) // 4 * 4, dtype=torch.uint8, device=dev)
            d.out_lo4, d.out_lo4_scale = o4.data_ptr(), o4s.data_ptr()
        P.kernel_tdnn_gemm(d)
        torch.cuda.synchronize()
        # the recorded group maxima are those of the fp32 results before the fp16 rounding of the plane
        want = z.float().abs().reshape(rows // 16, 16, -1).amax(dim=(1, 2))[:run_rows // 16]
        got = gm_out.view(torch.float32)[:run_rows // 16]
        assert torch.allclose(got, want, rtol=1e-4, atol=0), (got[:4], want[:4])
        if prec in (7, 9):   # the output plane with its own 4-bit residual: two to three bits better than fp16 alone
            if prec == 9:
                return oh.double().cpu().numpy() + _decode_lo4(o4, o4s, n_pad), z.cpu().numpy(), oh.cpu().numpy()
            return oh.double().cpu().numpy() + _decode_lo4(o4, o4s, n_pad), z.cpu().numpy()
        return oh.double().cpu().numpy(), z.cpu().numpy()
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


TDNN3 = [(0, 512, -3, 512), (0, 512, 0, 512), (0, 512, 3, 512)]
CVEC5_MX = [(0, 512, 0, 512), (1, 128, 0, 128)]      # two sources, each a whole number of 128-column blocks


@pytest.mark.parametrize("segs", [TDNN2, TDNN3, [(0, 512, 0, 512)], CVEC5_MX], ids=["tdnn2", "tdnn3", "tdnn4", "cvec5"])
def test_gemm_mx_act_per_tile_kernel(segs):
    # 3 x 256 rows: too few tiles for the persistent grid -> the 256-row per-tile kernel
    out, ref = _run_mx_case(0, 768, 512, segs, seed=11)
    assert np.abs(out - ref).max() / np.abs(ref).max() < TOL[4] + OUT_Q[4]


def test_gemm_mx_stats_per_tile_kernel():
    out, ref = _run_mx_case(2, 512, 1536, [(0, 512, 0, 512)], seed=12)
    scale = np.abs(ref).max(axis=(0, 2), keepdims=True)
    assert (np.abs(out - ref) / scale).max() < 3e-5


@pytest.mark.parametrize("epi", [0, 2])
@pytest.mark.parametrize("segs", [TDNN3, CVEC5_MX], ids=["tdnn3", "cvec5"])
def test_gemm_mx_stream_k(epi, segs):
    # 66 x 512 rows x 4 column tiles on 256 workgroups: heads, whole tiles and tails, cut at multiples of four steps
    out, ref = _run_mx_case(epi, 66 * 512, 512, segs, seed=13)
    if epi == 2:
        scale = np.abs(ref).max(axis=(0, 2), keepdims=True)
        assert (np.abs(out - ref) / scale).max() < 3e-5
    else:
        assert np.abs(out - ref).max() / np.abs(ref).max() < TOL[4] + OUT_Q[4]


def test_gemm_mx_refuses_unsuitable_launch():
    P = _pkg()
    with pytest.raises(P.XvError):
        # two sources of two K steps each: no 128-column block of the walk lies inside one source
        _run_mx_case(0, 768, 512, [(0, 64, 0, 64), (1, 64, 0, 64)], seed=1)


AM_MX = [(0, 768, -3, 768), (0, 768, 0, 768), (0, 768, 3, 768)]     # phonetic branch: 650-wide sources padded to 768
CVEC5_ODD = [(0, 512, 0, 512), (1, 128, 0, 128), (2, 384, 0, 384)]   # an odd number of 128-column chunks per source


@pytest.mark.parametrize("epi", [0, 2])
@pytest.mark.parametrize("segs", [TDNN3, CVEC5_MX, [(0, 512, 0, 512)], AM_MX, CVEC5_ODD], ids=["tdnn3", "cvec5", "tdnn4", "am768", "odd"])
def test_gemm_mx2_stream_k(epi, segs):
    """XV_PREC_FP16MX2 on the persistent kernel: the second K walk (4-bit residual of the activations x 4-bit image of the
    weights) against an exact emulation of both 4-bit products; parts cut inside both walks."""
    out, ref = _run_mx_case(epi, 66 * 512, 512, segs, seed=17, prec=7)
    if epi == 2:
        scale = np.abs(ref).max(axis=(0, 2), keepdims=True)
        assert (np.abs(out - ref) / scale).max() < 3e-5
    else:
        assert np.abs(out - ref).max() / np.abs(ref).max() < TOL[4] + 2.0 ** -13


@pytest.mark.parametrize("epi", [0, 2])
@pytest.mark.parametrize("segs", [TDNN3, CVEC5_MX, AM_MX, CVEC5_ODD], ids=["tdnn3", "cvec5", "am768", "odd"])
def test_gemm_mx2_per_tile_kernel(epi, segs):
    # 3 x 256 rows: too few tiles for the persistent grid -> the 256-row per-tile kernel
    out, ref = _run_mx_case(epi, 768, 512, segs, seed=19, prec=7)
    if epi == 2:
        scale = np.abs(ref).max(axis=(0, 2), keepdims=True)
        assert (np.abs(out - ref) / scale).max() < 3e-5
    else:
        assert np.abs(out - ref).max() / np.abs(ref).max() < TOL[4] + 2.0 ** -13


@pytest.mark.parametrize("rows", [66 * 512, 768], ids=["stream_k", "per_tile"])
@pytest.mark.parametrize("segs", [TDNN3, AM_MX], ids=["tdnn3", "am768"])
def test_gemm_mxe_is_the_mx_product_with_the_residual_plane_of_its_output(rows, segs):
    """Precision 9 (kPrecFp16MxE, the lite layers of a calibrated fp16mx2 context in front of a consumer that walks the
    residual plane): the 1.25-pass product - its fp16 plane is bit-identical to precision 6's - and, next to it, the 4-bit
    residual of that plane, which brings plane + residual two to three bits closer to the exact result."""
    full, ref, hi = _run_mx_case(0, rows, 512, segs, seed=23, prec=9)
    plain, ref6 = _run_mx_case(0, rows, 512, segs, seed=23, prec=6)
    assert np.array_equal(hi.astype(np.float64), plain)
    assert np.abs(full - ref).max() / np.abs(ref).max() < TOL[4] + 2.0 ** -13


# ---------------------------------------------------------------------------------------------------------------------
# tdnn_gemm_kernel_p8: 256 x 256 tiles, K tiles of 64 columns (xv_gemm_desc.p8).  Same references as above - the kernel
# forms its sums in another order (64-column chunk -> offset), which a tolerance on fp32 accumulation order does not see.
P8_TDNN2 = [(0, 512, -2, 512), (0, 512, 0, 512), (0, 512, 2, 512)]
P8_TDNN3 = [(0, 512, -3, 512), (0, 512, 0, 512), (0, 512, 3, 512)]
P8_PLAIN = [(0, 512, 0, 512)]
P8_TWO = [(0, 512, 0, 512), (1, 128, 0, 128)]


def _p8_check(out, ref, epi, tol):
    if epi == 2:
        scale = np.abs(ref).max(axis=(0, 2), keepdims=True)
        assert (np.abs(out - ref) / scale).max() < 3e-5
    else:
        assert np.abs(out - ref).max() / np.abs(ref).max() < tol


@pytest.mark.parametrize("epi", [0, 2])
@pytest.mark.parametrize("segs", [P8_TDNN2, P8_PLAIN, P8_TWO], ids=["tdnn2", "tdnn4", "cvec5"])
def test_gemm_p8_fp16_small_launch(epi, segs):
    # 3 row tiles x 2 column tiles on a grid of a few workgroups: whole tiles only, XCD blocks without tiles return at once
    out, ref = _run_case(2, epi, 3 * 256, 512, segs, relu=True, bn=True, seed=21, p8=1)
    _p8_check(out, ref, epi, TOL[2] + OUT_Q[2])


@pytest.mark.parametrize("epi", [0, 2])
def test_gemm_p8_fp16_stream_k(epi):
    # 100 row tiles x 2 column tiles = 200 tiles of 24 K tiles on 256 workgroups: 12.5 row tiles per XCD block, 16 groups of 2
    # column lanes, every workgroup a head part, whole tiles and a tail part cut at an even K tile inside an offset group
    out, ref = _run_case(2, epi, 100 * 256, 512, P8_TDNN3, relu=True, bn=True, seed=22, p8=1)
    _p8_check(out, ref, epi, TOL[2] + OUT_Q[2])


@pytest.mark.parametrize("epi", [0, 2])
@pytest.mark.parametrize("segs", [P8_TDNN2, P8_TDNN3, P8_PLAIN, P8_TWO], ids=["tdnn2", "tdnn3", "tdnn4", "cvec5"])
def test_gemm_p8_mx_small_launch(epi, segs):
    out, ref = _run_mx_case(epi, 3 * 256, 512, segs, seed=23, p8=1)
    _p8_check(out, ref, epi, 2e-5 + 2.0 ** -10)


@pytest.mark.parametrize("epi", [0, 2])
@pytest.mark.parametrize("segs", [P8_TDNN3, P8_TWO], ids=["tdnn3", "cvec5"])
def test_gemm_p8_mx_stream_k(epi, segs):
    out, ref = _run_mx_case(epi, 100 * 256, 768 if segs is P8_TWO else 512, segs, seed=24, p8=1)
    _p8_check(out, ref, epi, 2e-5 + 2.0 ** -10)


def test_gemm_p8_is_independent_of_the_cut():
    """The same rows as part of a large launch (stream-K cuts inside tiles) and as a launch of their own (whole tiles): same
    bits - a chunk's activations do not depend on what it is batched with."""
    torch = _torch()
    big, _ = _run_mx_case(0, 100 * 256, 512, P8_TDNN2, seed=25, p8=1)
    small, _ = _run_mx_case(0, 100 * 256, 512, P8_TDNN2, seed=25, p8=1, variant_rows=2 * 256)
    assert np.array_equal(big[:2 * 256], small[:2 * 256])


def test_gemm_p8_refuses_unsuitable_launch():
    P = _pkg()
    with pytest.raises(P.XvError):
        _run_case(2, 0, 3 * 128, 512, P8_PLAIN, relu=True, bn=True, seed=1, p8=1)          # rows not a multiple of 256
    with pytest.raises(P.XvError):
        _run_case(2, 0, 2 * 256, 384, P8_PLAIN, relu=True, bn=True, seed=1, p8=1)          # n_pad not a multiple of 256
    with pytest.raises(P.XvError):
        _run_case(3, 0, 2 * 256, 512, P8_PLAIN, relu=True, bn=True, seed=1, p8=1)          # three-pass arithmetic
    with pytest.raises(P.XvError):
        _run_case(2, 0, 2 * 256, 512, [(0, 96, 0, 96)], relu=True, bn=True, seed=1, p8=1)  # K not in whole 64-column tiles


# ---- the 1.5-pass arithmetic (kPrecFp16Mx2) on tdnn_gemm_kernel_p8: behind the fp16 tiles a second walk over the activations'
# 4-bit residual planes and the 4-bit image of the weights, in tiles of 256 4-bit columns (sources of whole 256-column tiles)
P8_AM768 = [(0, 768, -3, 768), (0, 768, 0, 768), (0, 768, 3, 768)]
P8_TWO256 = [(0, 512, 0, 512), (1, 256, 0, 256)]
P8_MX2_CASES = [P8_TDNN2, P8_TDNN3, P8_PLAIN, P8_AM768, P8_TWO256]
P8_MX2_IDS = ["tdnn2", "tdnn3", "tdnn4", "am768", "two"]


def _p8_mx2(epi, rows, n_pad, segs, seed, **kw):
    return _run_mx_case(epi, rows, n_pad, segs, seed=seed, prec=7, p8=1, **kw)


@pytest.mark.parametrize("epi", [0, 2])
@pytest.mark.parametrize("segs", P8_MX2_CASES, ids=P8_MX2_IDS)
def test_gemm_p8_mx2_small_launch(epi, segs):
    out, ref = _p8_mx2(epi, 3 * 256, 768 if segs is P8_AM768 else 512, segs, 27)
    _p8_check(out, ref, epi, TOL[4] + 2.0 ** -13)


@pytest.mark.parametrize("epi", [0, 2])
@pytest.mark.parametrize("segs", P8_MX2_CASES, ids=P8_MX2_IDS)
def test_gemm_p8_mx2_stream_k(epi, segs):
    # 100 row tiles: every workgroup a head part, whole tiles and a tail part, cut inside either walk
    out, ref = _p8_mx2(epi, 100 * 256, 768 if segs is P8_AM768 else 512, segs, 28)
    _p8_check(out, ref, epi, TOL[4] + 2.0 ** -13)


@pytest.mark.parametrize("epi", [0, 2])
@pytest.mark.parametrize("rows", [100 * 256, 3 * 256], ids=["stream_k", "small"])
def test_gemm_p8_mx2_is_bit_stable_under_load(epi, rows):
    """The launches that differed from run to run in round 4 (every one with time offsets; tools/repeat_mx_case.py): the scales
    of a second-walk tile were staged into the buffer the late wave group was still reading.  The same launch 40 times beside a
    stream that keeps the CUs and the memory system busy: bit-identical."""
    import threading
    torch = _torch()
    stop = []

    def noise():
        st = torch.cuda.Stream()
        a = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
        b = torch.randn(1 << 25, device="cuda")
        with torch.cuda.stream(st):
            while not stop:
                for _ in range(4):
                    (a @ a)
                    b.add_(1.0)
                st.synchronize()
    tn = threading.Thread(target=noise)
    tn.start()
    try:
        ref, bad = None, 0
        for i in range(40):
            out = _p8_mx2(epi, rows, 512, P8_TDNN3, 29)[0]
            if ref is None:
                ref = out
            elif not np.array_equal(ref, out):
                bad += 1
    finally:
        stop.append(1)
        tn.join()
    assert bad == 0, "%d of 39 repeats differ" % bad


@pytest.mark.parametrize("rows", [100 * 256, 3 * 256], ids=["stream_k", "small"])
@pytest.mark.parametrize("segs", [P8_TDNN3, P8_AM768, P8_PLAIN], ids=["tdnn3", "am768", "tdnn4"])
def test_gemm_p8_mxe_is_the_p8_mx_product_with_the_residual_plane_of_its_output(rows, segs):
    """Precision 9 on tdnn_gemm_kernel_p8 (a lite layer of a calibrated mixture whose consumer still walks its residual plane):
    the fp16 plane has the bits of precision 6 on the same kernel, and plane + 4-bit residual is two to three bits closer to the
    exact result."""
    n_pad = 768 if segs is P8_AM768 else 512
    full, ref, hi = _run_mx_case(0, rows, n_pad, segs, seed=31, prec=9, p8=1)
    plain, _ = _run_mx_case(0, rows, n_pad, segs, seed=31, prec=6, p8=1)
    assert np.array_equal(hi.astype(np.float64), plain)
    assert np.abs(full - ref).max() / np.abs(ref).max() < TOL[4] + 2.0 ** -13
