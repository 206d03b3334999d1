"""CPU tests of the product's host side through the C ABI and the command-line tools (no compute calls: there is
no GPU here, and the product has no CPU compute path - which is itself asserted)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import helpers as H

P = H.pkg()
BIN = os.path.join(H.ROOT, H.PKG_NAME, "bin")
HEADER = os.path.join(H.ROOT, "include", "xvec_hip.h")


def test_library_exports_every_declared_symbol():
    text = open(HEADER).read()
    declared = set(re.findall(r"\b(xv_[a-z_0-9]+)\s*\(", text))
    declared -= {"xv_status"}
    lib = ctypes.CDLL(P.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(P.ABI_SYMBOLS), declared ^ set(P.ABI_SYMBOLS)
    P.lib().xv_version.restype = ctypes.c_char_p
    assert b"xvec_hip" in P.lib().xv_version()


EXPECT = {  # topology -> (left, right, min_frames, layers in the cone, MACs at T=400)  
    "v2_xvector": (7, 7, 15, 6, 1034332160),
    "v3_multitask": (7, 7, 15, 6, 1034332160),
    "v4_cvector": (13, 7, 21, 11, 2701279832),
    "v5_cvector": (13, 7, 21, 11, 2701279832),
    "pa_wo_pretrain": (13, 7, 21, 11, 2701279832),
}


@pytest.mark.parametrize("topology", sorted(EXPECT))
@pytest.mark.parametrize("binary", [True, False])
def test_model_load_and_lowering(topology, binary):
    net, line = H.synth_model(topology)
    if not binary and topology not in ("v2_xvector", "v5_cvector"):
        pytest.skip("text flavour covered on two topologies")
    m = P.Model(raw=net.to_bytes(binary), nnet_config=line)
    i = m.info
    left, right, minf, nl, macs = EXPECT[topology]
    assert (i.input_dim, i.output_dim, i.left_context, i.right_context, i.min_frames, i.num_layers, i.output_is_segment) == \
        (23, 512, left, right, minf, nl, 1)
    assert m.macs(400) == macs
    d = m.describe()
    assert "mean+stddev pooling" in d and "(segment)" in d
    # the oracle's independent context derivation agrees
    n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True))
    n2.apply_nnet_config(line)
    assert H.xo.GraphEvaluator(n2, np.float32).context() == (left, right)


def test_binary_and_text_models_pack_identically():
    net, line = H.synth_model("v2_xvector")
    a = P.Model(raw=net.to_bytes(True), nnet_config=line).pack(P.PREC_BF16X3)
    b = P.Model(raw=net.to_bytes(False), nnet_config=line).pack(P.PREC_BF16X3)
    assert a == b
    c = P.Model(raw=net.to_bytes(True), nnet_config=line).pack(P.PREC_FP16)
    assert len(c) < len(a)        # single plane instead of hi+lo


def _bn_mantissa(net, bn_name):
    """FoldBatchNormIntoConsumers (csrc/program.cc): the BatchNorm of a relu-batchnorm layer that only feeds other layers moves
    into its consumers - s = m * 2^e, the producer keeps 2^e, the consumer's weight columns take m in [1, 2) (and its bias W . o).
    Returns m per column of the producing layer, as the library computes it (float scale, double arithmetic)."""
    c = net.components[bn_name]
    s = (np.float64(c.f.get("target_rms", 1.0)) * (np.asarray(c.f["stats_var"], np.float64) + c.f.get("epsilon", 1e-3)) ** -0.5).astype(np.float32)
    fr, _ = np.frexp(s.astype(np.float64))
    return 2.0 * fr


def test_split_fp16_modes_share_one_weight_image_scaled_by_a_power_of_two():
    """fp16x3 / fp16x2 are kernel policies over the same packed weights: fp16 hi + lo planes of W * 2^S, S chosen so that
    max |w| * 2^S lies in [2^13, 2^14) (the residual plane then holds fp16 normals); the exact inverse sits in the
    epilogue parameters (bias * 2^S, scale * 2^-S).  (The modes with 4-bit products - auto, fp16mx, fp16mx2 - hold the same
    values with the K of inter-layer sources padded to blocks of 128 columns, plus their 4-bit planes.)"""
    net = H.nm.synthesize(H.tiny_config(), seed=3)
    m = P.Model(raw=net.to_bytes(True))
    blobs = [m.pack(p) for p in (P.PREC_FP16X3, P.PREC_FP16X2)]
    assert len(set(len(b) for b in blobs)) == 1
    # identical up to the precision field of the header and the fingerprint of the image (which covers that field)
    diff = [i for i in range(len(blobs[0])) if blobs[0][i] != blobs[1][i]]
    assert 0 < len(diff) <= 4 + 8 and max(diff) < 128
    assert len(m.pack(P.PREC_AUTO)) >= len(blobs[0]) and len(m.pack(P.PRECISIONS["fp16mx2"])) >= len(blobs[0])
    w = np.asarray(net.components["tdnn4.affine"].f["linear"], np.float32)     # 12 x 12
    scale = 2.0 ** (14 - np.frexp(np.abs(w).max())[1])
    assert 2 ** 13 <= np.abs(w).max() * scale < 2 ** 14
    u16 = np.frombuffer(blobs[0], dtype=np.uint16)
    row_hi = (w[0] * scale).astype(np.float16)
    first_row = row_hi.view(np.uint16)
    pos = [i for i in range(len(u16) - 12) if np.array_equal(u16[i:i + 12], first_row)]
    assert pos, "fp16(hi) image of tdnn4's first scaled weight row not found in the blob"
    n_pad, k_pad = 128, 32
    hi = u16[pos[0]:pos[0] + n_pad * k_pad].view(np.float16).reshape(n_pad, k_pad)[:12, :12].astype(np.float64)
    lo_all = u16[pos[0] + n_pad * k_pad:pos[0] + 2 * n_pad * k_pad].view(np.float16).reshape(n_pad, k_pad)
    lo = lo_all[:12, :12].astype(np.float64)
    rec = (hi + lo) / scale
    assert np.max(np.abs(rec - w) / np.abs(w).max()) < 2.0 ** -20
    assert np.all(np.abs(lo[lo != 0]) >= 2.0 ** -14) or np.mean(np.abs(lo[lo != 0]) >= 2.0 ** -14) > 0.9   # normals


def test_packed_weights_are_the_split_of_the_fp32_weights():
    net = H.nm.synthesize(H.tiny_config(), seed=3)
    blob = P.Model(raw=net.to_bytes(True)).pack(P.PREC_BF16X3)
    # hi + lo reproduces every weight to ~2^-17 relative: find tdnn2's [12 x 24] block by value search
    w = np.asarray(net.components["tdnn4.affine"].f["linear"], np.float32)     # 12x12, K padded to 32, N to 128
    u16 = np.frombuffer(blob, dtype=np.uint16)
    f = (u16.astype(np.uint32) << 16).view(np.float32)
    hi = (np.frombuffer(w.tobytes(), np.uint32) + 0x7FFF + ((np.frombuffer(w.tobytes(), np.uint32) >> 16) & 1)) >> 16
    first_row = hi[:12].astype(np.uint16)
    pos = [i for i in range(len(u16) - 12) if np.array_equal(u16[i:i + 12], first_row)]
    assert pos, "bf16(hi) image of tdnn4's first weight row not found in the blob"
    p = pos[0]
    n_pad, k_pad = 128, 32
    hi_plane = f[p:p + n_pad * k_pad].reshape(n_pad, k_pad)
    # the lo plane follows the hi plane (256-byte aligned; 128*32*2 bytes is already aligned)
    lo_plane = f[p + n_pad * k_pad:p + 2 * n_pad * k_pad].reshape(n_pad, k_pad)
    rec = hi_plane[:12, :12].astype(np.float64) + lo_plane[:12, :12].astype(np.float64)
    assert np.abs(rec - w).max() <= 2.0 ** -16 * np.abs(w).max()
    assert not hi_plane[12:, :].any() and not hi_plane[:, 12:].any()            # zero padding of rows / K


def test_output_node_selection_and_nnet_config():
    net, _ = H.synth_model("v2_xvector")
    raw = net.to_bytes(True)
    a = P.Model(raw=raw, nnet_config="output-node name=output input=tdnn6.affine")
    assert "tdnn6.affine" in a.describe() and a.info.num_layers == 6
    b = P.Model(raw=raw, nnet_config="output-node name=output input=tdnn7.affine")    # the "other" embedding layer
    assert b.info.num_layers == 7 and "tdnn6.batchnorm" in b.describe()
    # the unedited model's output is the speaker log-softmax: pooled, 5139 classes
    c = P.Model(raw=raw)
    assert c.info.output_dim == 5139 and "log-softmax" in c.describe()


@pytest.mark.parametrize("mutation,needle", [
    (lambda s: s.replace("Append(Offset(tdnn1.batchnorm, -2)", "Sum(Offset(tdnn1.batchnorm, -2)"), "not supported"),
    (lambda s: s.replace("input=tdnn3.batchnorm", "input=nosuchnode"), "unknown node"),
    (lambda s: s.replace("output-node name=output", "output-node name=out2"), "no output-node"),
])
def test_unsupported_graphs_fail_with_model_error(mutation, needle):
    net = H.nm.synthesize(H.tiny_config(), seed=1)
    net.config_lines = [mutation(l) for l in net.config_lines]
    with pytest.raises(P.XvError) as e:
        P.Model(raw=net.to_bytes(True))
    assert e.value.status in (1, 2) and needle in str(e.value)


def test_truncated_model_is_an_io_error():
    net, _ = H.synth_model("v2_xvector")
    raw = net.to_bytes(True)
    with pytest.raises(P.XvError) as e:
        P.Model(raw=raw[:len(raw) // 2])
    assert e.value.status == 1
    with pytest.raises(P.XvError):
        P.Model(rxfilename="/nonexistent/final.raw")


def test_no_cpu_compute_path():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    net, line = H.synth_model("v2_xvector")
    m = P.Model(raw=net.to_bytes(True), nnet_config=line)
    with pytest.raises(P.XvError) as e:
        P.Context(m)
    assert e.value.status == 3 and "no CPU path" in str(e.value)


def _py_plan(T, chunk, minc, pad):
    """Chunk list per SURVEY.md App. B.5, written independently of csrc/extractor.cc."""
    if T == 0 or (not pad and T < minc):
        return None
    this = T if (chunk <= 0 or T < chunk) else chunk
    out = []
    for ci in range(-(-T // this)):
        ln = min(this, T - ci * this)
        if ln < minc:
            if not pad:
                continue
            left = (minc - ln) // 2
            out.append((ci * this, ln, left, minc - ln - left))
        else:
            out.append((ci * this, ln, 0, 0))
    return out or None


@pytest.mark.parametrize("T", [0, 1, 14, 15, 24, 25, 26, 299, 300, 301, 920, 10001])
@pytest.mark.parametrize("chunk,minc", [(-1, 25), (300, 25), (10000, 25), (300, 100)])
@pytest.mark.parametrize("pad", [True, False])
def test_chunk_planning(T, chunk, minc, pad):
    got = P.plan_chunks(T, chunk, minc, pad, min_net_frames=15)
    want = _py_plan(T, chunk, minc, pad)
    if want is not None and any(ln + l + r < 15 for _, ln, l, r in want):
        want = None       # a chunk shorter than the network context cannot be computed: utterance fails
    assert got == want


# ------------------------------------------------------------------------------------- command-line tools
def _run(tool, *args, **kw):
    return subprocess.run([os.path.join(BIN, tool)] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)


def test_nnet3_copy_shim_pipe_form(tmp_path):
    net, line = H.synth_model("v2_xvector")
    raw = net.to_bytes(True)
    (tmp_path / "final.raw").write_bytes(raw)
    (tmp_path / "extract.config").write_text(line + "\n")
    # exactly the rxfilename extract_xvectors_new.sh:59 builds
    rx = "%s --nnet-config=%s/extract.config %s/final.raw - |" % (os.path.join(BIN, "nnet3-copy"), tmp_path, tmp_path)
    m = P.Model(rxfilename=rx)
    assert m.info.num_layers == 6 and m.info.output_dim == 512
    assert m.pack() == P.Model(raw=raw, nnet_config=line).pack()
    # file -> file, components byte-identical, readable by the independent Python parser
    r = _run("nnet3-copy", "--nnet-config=%s/extract.config" % tmp_path, str(tmp_path / "final.raw"), str(tmp_path / "out.raw"))
    assert r.returncode == 0, r.stderr
    n2 = H.nm.Nnet3.from_bytes((tmp_path / "out.raw").read_bytes())
    assert any("output-node name=output input=tdnn6.affine" in l for l in n2.config_lines)
    assert np.array_equal(n2.components["tdnn3.affine"].f["linear"], net.components["tdnn3.affine"].f["linear"])
    r = _run("nnet3-copy", "--edits=foo", str(tmp_path / "final.raw"), "-")
    assert r.returncode == 1 and b"not supported" in r.stderr


def test_cli_contract_without_gpu(tmp_path):
    r = _run("nnet3-xvector-compute", "--help")
    assert r.returncode == 0 and b"Usage: nnet3-xvector-compute" in r.stderr
    r = _run("nnet3-xvector-compute", "only-one-arg")
    assert r.returncode == 1
    r = _run("nnet3-xvector-compute", "--chunk-size=abc", "a", "b", "c")
    assert r.returncode == 1 and b"invalid integer" in r.stderr
    r = _run("nnet3-xvector-compute", "--use-gpu=no", "/nonexistent.raw", "ark:/dev/null", "ark:/dev/null")
    assert r.returncode == 255 and b"cannot open" in r.stderr
    # upstream options that only tune the computation are ignored with a warning (compute_output.sh:117 passes them to
    # nnet3-compute); options that change the numbers are refused rather than ignored
    r = _run("nnet3-compute", "--frames-per-chunk=50", "--extra-left-context=10", "--frame-subsampling-factor=1", "/nonexistent.raw",
             "ark:/dev/null", "ark:/dev/null")
    assert r.returncode == 255 and r.stderr.count(b"ignoring option") == 3
    for opt in ("--frame-subsampling-factor=3", "--online-ivectors=scp:iv.scp", "--ivectors=scp:iv.scp", "--use-priors=true"):
        r = _run("nnet3-compute", opt, "/nonexistent.raw", "ark:/dev/null", "ark:/dev/null")
        assert r.returncode == 1 and b"not supported" in r.stderr, opt
    import torch
    if not torch.cuda.is_available():
        net, line = H.synth_model("v2_xvector")
        (tmp_path / "final.raw").write_bytes(net.to_bytes(True))
        r = _run("nnet3-xvector-compute", "--use-gpu=no", "--output-node=tdnn6.affine", str(tmp_path / "final.raw"),
                 "ark:/dev/null", "ark:/dev/null")
        # fails loudly: there is no CPU fallback behind --use-gpu=no
        assert r.returncode == 255 and b"no CPU path" in r.stderr and b"--use-gpu=no requested" in r.stderr


E2M1 = np.array([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])


def _swap_fields(rho):
    p, g, r = (rho >> 4) & 3, (rho >> 2) & 3, rho & 3
    return ((p >> 1) << 5) | (g << 3) | ((p & 1) << 2) | r


@pytest.mark.parametrize("segs", [[(0, 0, 512)], [(0, -2, 256), (0, 0, 256), (0, 2, 256)], [(0, 0, 256), (1, 0, 128)]],
                         ids=["plain", "spliced", "two_sources"])
def test_mx_residual_packing_and_scale_tiling(segs):
    """XV_PREC_FP16MX packer (host code, no GPU): every lane-group chunk of 32 residuals gets the smallest power-of-two
    scale that keeps it inside the e2m1 range, the nibbles are the nearest grid values, and xv_tile_mx_scales puts the
    scale of LDS row rho / lane group g where the kernels' dword read expects it (both operand orientations)."""
    rng = np.random.default_rng(3)
    n_pad, K = 256, sum(s[2] for s in segs)
    w = (rng.standard_normal((n_pad, K)) * rng.uniform(0.01, 4.0, (n_pad, 1))).astype(np.float32)
    hi = w.astype(np.float16)
    res = w.astype(np.float64) - hi.astype(np.float64)
    w4, sc = P.pack_mx_residual(w, hi.view(np.uint16), segs)
    assert w4.shape == (n_pad, K // 2) and sc.shape == (n_pad, K // 32)
    # walk order: consecutive segments over one source form a group, walked 32-column chunk by chunk, offset by offset
    step_cols = []
    k0 = j = 0
    while j < len(segs):
        ns = 1
        while j + ns < len(segs) and segs[j + ns][0] == segs[j][0] and segs[j + ns][2] == segs[j][2]:
            ns += 1
        klen = segs[j][2]
        for kk in range(klen // 32):
            for jj in range(ns):
                step_cols.append(k0 + jj * klen + kk * 32)
        k0 += ns * klen
        j += ns
    assert len(step_cols) == K // 32
    nib = np.stack([w4 & 15, w4 >> 4], axis=-1).reshape(n_pad, K // 128, 4, 32)      # [row][block][lane group][element]
    val = np.where(nib & 8, -1.0, 1.0) * E2M1[nib & 7]
    for b in range(K // 128):
        for g in range(4):
            cols = np.array([step_cols[4 * b + e // 8] + 8 * g + e % 8 for e in range(32)])
            r = res[:, cols]
            s = 2.0 ** (sc[:, 4 * b + g].astype(np.float64) - 127)[:, None]
            m = np.abs(r).max(axis=1)
            assert np.all(m <= 6 * s[:, 0] * (1 + 1e-12)) and np.all((m > 3 * s[:, 0]) | (m == 0))
            # nearest grid value: the reconstruction error is at most half the local grid spacing (1 above 4, ...)
            err = np.abs(val[:, b, g, :] * s - r) / s
            a = np.abs(r / s)
            half_gap = np.where(a >= 4, 1.0, np.where(a >= 2, 0.5, 0.25))
            assert np.all(err <= half_gap + 1e-9)
    for epi, swap in ((0, True), (2, False)):
        tiled = P.tile_mx_scales(sc, epi)
        nblk = K // 128
        for t in range(n_pad // 128):
            for rho in (0, 1, 17, 63, 64, 100, 127):
                row = t * 128 + ((rho & 64) | _swap_fields(rho & 63) if swap else rho)
                h, wf, i = rho >> 6, (rho >> 4) & 3, rho & 15
                for b in (0, nblk - 1):
                    for g in range(4):
                        assert tiled[(t * nblk + b) * 512 + h * 256 + (i * 4 + g) * 4 + wf] == sc[row, 4 * b + g]


def test_mx_weight_image_packing():
    """xv_pack_mx_weights (host code): the 4-bit image of the weights for the second K walk of XV_PREC_FP16MX2 - per row
    and lane-group chunk of 32 consecutive columns the smallest power-of-two scale that keeps the chunk inside the e2m1
    range, nearest grid values, 64 bytes per 128-column step in walk order (chunk -> offset)."""
    rng = np.random.default_rng(5)
    segs = [(0, -2, 256), (0, 0, 256), (0, 2, 256), (1, 0, 128)]
    n_pad, K = 128, sum(s[2] for s in segs)
    w = (rng.standard_normal((n_pad, K)) * rng.uniform(0.01, 40.0, (n_pad, 1))).astype(np.float32)
    w4b, sc = P.pack_mx_weights(w, segs)
    assert w4b.shape == (n_pad, K // 2) and sc.shape == (n_pad, K // 32)
    lo_col = [kq * 128 + j * 256 for kq in range(2) for j in range(3)] + [768]      # steps of the second walk
    nib = np.stack([w4b & 15, w4b >> 4], axis=-1).reshape(n_pad, K // 128, 4, 32)      # [row][step][lane group][element]
    val = np.where(nib & 8, -1.0, 1.0) * E2M1[nib & 7]
    for t, c0 in enumerate(lo_col):
        for g in range(4):
            x = w[:, c0 + 32 * g: c0 + 32 * g + 32].astype(np.float64)
            s = 2.0 ** (sc[:, 4 * t + g].astype(np.float64) - 127)[:, None]
            m = np.abs(x).max(axis=1)
            assert np.all(m <= 6 * s[:, 0] * (1 + 1e-12)) and np.all(m > 3 * s[:, 0])
            a = np.abs(x / s)
            half_gap = np.where(a >= 4, 1.0, np.where(a >= 2, 0.5, 0.25))
            assert np.all(np.abs(val[:, t, g, :] * s - x) / s <= half_gap + 1e-9)


def test_default_precision_policy_is_one_rule_for_every_entry_point():
    """XV_PREC_DEFAULT (what the command line, dist_extract.py, bench.py and Context() use without being told anything
    else; engine.cc PackModelPolicy): fp16mx2 for a pooled output whose layers can all run it, fp16x3 for frame-level
    outputs - resolved when the model is packed, so the image a rank broadcasts already says what it is."""
    import struct
    def packed_precision(blob):     # BlobHeader: char magic[8]; uint32 version; int32 precision
        assert blob[:8] == b"XVHIPBLB"
        return struct.unpack_from("<i", blob, 12)[0]
    net, line = H.synth_model("v2_xvector")
    pooled = P.Model(raw=net.to_bytes(True), nnet_config=line)
    assert pooled.pack() == pooled.pack(P.PREC_DEFAULT) == pooled.pack(P.PRECISIONS["fp16mx2"])
    assert packed_precision(pooled.pack()) == P.PREC_FP16MX2
    frames = P.Model(raw=net.to_bytes(True), nnet_config="output-node name=output input=tdnn4.batchnorm")
    assert frames.info.output_is_segment == 0
    assert packed_precision(frames.pack()) == P.PREC_FP16X3
    cnet, cline = H.synth_model("v5_cvector")     # the c-vector network's 650-wide branch is padded to whole 128-column blocks
    assert packed_precision(P.Model(raw=cnet.to_bytes(True), nnet_config=cline).pack()) == P.PREC_FP16MX2
    assert P.PRECISIONS["default"] == P.PREC_DEFAULT == -1 and P.PRECISION_NAMES[P.PREC_FP16MX2] == "fp16mx2"


def test_blob_layer_table_is_validated_before_anything_becomes_a_device_pointer():
    """every offset of the layer table that the engine turns into base + offset device pointers - the images in the
    K-walk order of tdnn_gemm_kernel_p8 (w4p / w4bp and their scales) included - is range-checked when the blob is parsed,
    i.e. before the first HIP call: a corrupt or foreign image is XV_ERR_*, "layer i is inconsistent", never a GPU fault."""
    import struct
    net, line = H.synth_model("v2_xvector")
    blob = bytearray(P.Model(raw=net.to_bytes(True), nnet_config=line).pack(P.PRECISIONS["fp16mx2"]))
    HDR, LAYER = 112, 320                      # sizeof(BlobHeader), sizeof(BlobLayer) (engine.cc)
    off = {"w4": 248, "w4_scale": 256, "w4p": 288, "w4p_scale": 296, "w4bp": 304, "w4bp_scale": 312}
    n_layers = struct.unpack_from("<i", blob, 20)[0]
    data_offset, total = struct.unpack_from("<QQ", blob, 96)
    assert total == len(blob) and HDR + n_layers * LAYER <= data_offset
    names = [bytes(blob[HDR + i * LAYER:HDR + i * LAYER + 64]).split(b"\0")[0].decode() for i in range(n_layers)]
    i = names.index("tdnn2.batchnorm")
    base = HDR + i * LAYER
    NONE = (1 << 64) - 1
    w4p, w4p_scale = struct.unpack_from("<QQ", blob, base + off["w4p"])
    assert w4p != NONE and w4p_scale != NONE, "tdnn2 of the x-vector network carries the p8 residual image"

    def message(mutate):
        b = bytearray(blob)
        mutate(b)
        with pytest.raises(P.XvError) as e:
            P.Context(blob=bytes(b), device=0)
        return str(e.value)
    data_bytes = total - data_offset
    bad = {
        "w4p beyond the image": lambda b: struct.pack_into("<Q", b, base + off["w4p"], data_bytes - 16),
        "w4p without its scales": lambda b: struct.pack_into("<Q", b, base + off["w4p_scale"], NONE),
        "w4p scales beyond the image": lambda b: struct.pack_into("<Q", b, base + off["w4p_scale"], data_bytes - 16),
        "w4p without w4": lambda b: struct.pack_into("<QQ", b, base + off["w4"], NONE, NONE),
        "w4bp without w4b": lambda b: (struct.pack_into("<QQ", b, base + off["w4bp"], 0, 0),
                                       struct.pack_into("<QQ", b, base + 272, NONE, NONE)),      # w4b, w4b_scale
        "w4bp scales without w4bp": lambda b: struct.pack_into("<QQ", b, base + off["w4bp"], NONE, 0),
    }
    for what, mut in bad.items():
        assert "inconsistent" in message(mut), what
    # the untouched image gets past the parser: with a GPU it loads, without one the failure is the device's, not the blob's
    try:
        P.Context(blob=bytes(blob), device=0).close()
    except P.XvError as e:
        assert "inconsistent" not in str(e) and e.status == 3, str(e)


def test_bn_fold_is_opt_in_and_moves_the_mantissa_into_the_consumer(monkeypatch):
    """XVEC_DEBUG=bn_fold=1 (csrc/program.cc FoldBatchNormIntoConsumers; OFF by default - measured in round 5, it costs the fast
    arithmetics the averaging of their weight-side errors): the packed image of tdnn4 then holds W * diag(m) with m the mantissa
    of tdnn3.batchnorm's scale, and the describe table says which layers gave their BatchNorm away."""
    net = H.nm.synthesize(H.tiny_config(), seed=3)
    plain = P.Model(raw=net.to_bytes(True))
    assert "folded" not in plain.describe()
    monkeypatch.setenv("XVEC_DEBUG", "bn_fold=1")
    folded = P.Model(raw=net.to_bytes(True))
    monkeypatch.delenv("XVEC_DEBUG")
    d = folded.describe()
    assert d.count("bn(folded)") == 4 and "tdnn5.batchnorm" in d     # tdnn1-4 feed other layers; tdnn5 feeds the pooling
    blob = folded.pack(P.PREC_BF16X3)
    assert blob != plain.pack(P.PREC_BF16X3)
    w = (np.asarray(net.components["tdnn4.affine"].f["linear"], np.float64) * _bn_mantissa(net, "tdnn3.batchnorm")[None, :]).astype(np.float32)
    u16 = np.frombuffer(blob, dtype=np.uint16)
    bits = np.frombuffer(w.tobytes(), np.uint32)
    first_row = ((bits + 0x7FFF + ((bits >> 16) & 1)) >> 16)[:12].astype(np.uint16)      # bf16(hi) of the folded row
    assert any(np.array_equal(u16[i:i + 12], first_row) for i in range(len(u16) - 12))


def _publish_worker(args):
    path, k = args
    import importlib
    pkg = importlib.import_module(H.PKG_NAME)
    won, got = pkg.calibration_file_publish(path, 0x1234 + 0, "fp16mx2", lite_mask=1 << k, note="job %d" % k)
    return won, got["lite_mask"]


def test_calibration_file_round_trip_first_publisher_wins_and_bad_files_are_errors(tmp_path):
    """The shared choice of a recipe (csrc/calib_file.h, VERDICT r05 "missing" 1): what a job publishes is what every job reads;
    of several jobs that publish at once (run.pl JOB=1:nj starts them together, extract_xvectors_new.sh:91) exactly one wins and
    ALL hold the winner's choice afterwards; a file that cannot be parsed is an error, never silently the default."""
    path = str(tmp_path / "xvec.calib")
    assert P.calibration_file_read(path) is None
    won, got = P.calibration_file_publish(path, 0xfeedbeefcafe0123, "fp16mx2", lite_mask=0x2e6, tol=7.5e-5, note="64 chunks\nworst 6.5e-05")
    assert won and got == {"model": 0xfeedbeefcafe0123, "precision": "fp16mx2", "lite_mask": 0x2e6}
    assert P.calibration_file_read(path) == got
    text = open(path).read().splitlines()
    assert text[0] == "xvec-calibration 1" and "model feedbeefcafe0123" in text and "precision fp16mx2" in text and "lite-mask 2e6" in text
    assert any(ln.startswith("note 64 chunks worst") for ln in text)        # one line, whatever the note held
    won2, got2 = P.calibration_file_publish(path, 0x1, "fp16mx")             # too late: the first choice stands
    assert not won2 and got2 == got
    assert [f for f in os.listdir(tmp_path) if ".tmp." in f] == []          # no temporary left behind
    # eight jobs at once, each with a choice of its own
    import multiprocessing as mp
    race = str(tmp_path / "race.calib")
    with mp.get_context("spawn").Pool(8) as pool:
        res = pool.map(_publish_worker, [(race, k) for k in range(8)])
    assert sum(w for w, _ in res) == 1
    assert len(set(m for _, m in res)) == 1 and res[[w for w, _ in res].index(True)][1] == P.calibration_file_read(race)["lite_mask"]
    # malformed files
    for bad in ("", "something else\n", "xvec-calibration 2\nmodel 1\nprecision fp16mx\n", "xvec-calibration 1\nprecision fp16mx\n",
                "xvec-calibration 1\nmodel 12\nprecision bf16\n", "xvec-calibration 1\nmodel 12\nprecision fp16mx\nlite-mask 3\n"):
        open(str(tmp_path / "bad.calib"), "w").write(bad)
        with pytest.raises(P.XvError) as e:
            P.calibration_file_read(str(tmp_path / "bad.calib"))
        assert "calibration file" in str(e.value)
    with pytest.raises(P.XvError):          # a choice that cannot be shared must not be run on
        P.calibration_file_publish(str(tmp_path / "no_such_dir" / "x.calib"), 1, "fp16mx")
    with pytest.raises(P.XvError):
        P.calibration_file_publish(path, 1, "fp16mx", lite_mask=1)


def test_packed_image_carries_its_fingerprint():
    """Blob version 8: the header names the image (FNV-1a over the whole image with the field zero) - what a calibration file is
    keyed by.  Two packs of one model agree; another precision, another model or another output node give another fingerprint."""
    import struct

    def fp(blob):       # BlobHeader: magic[8] u32 version i32 precision 6 x i32 f32 6 x i32 reserved[7]
        assert struct.unpack_from("<I", blob, 8)[0] == 8
        lo, hi = struct.unpack_from("<II", blob, 8 + 4 + 4 + 6 * 4 + 4 + 6 * 4)
        return lo | (hi << 32)
    net, line = H.synth_model("v2_xvector")
    a = P.Model(raw=net.to_bytes(True), nnet_config=line).pack()
    b = P.Model(raw=net.to_bytes(False), nnet_config=line).pack()
    assert a == b and fp(a) != 0
    zeroed = bytearray(a)
    off = 8 + 4 + 4 + 6 * 4 + 4 + 6 * 4
    zeroed[off:off + 8] = bytes(8)
    h = 0xcbf29ce484222325
    words = np.frombuffer(bytes(zeroed[:len(zeroed) // 8 * 8]), dtype="<u8")
    for w in words[:4096]:                      # (the first 32 KiB in pure Python: the rule, not the speed)
        h = ((h ^ int(w)) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    assert h != fp(a)                           # the fingerprint covers the planes too, not just the head
    others = [P.Model(raw=net.to_bytes(True), nnet_config=line).pack(P.PREC_FP16X3),
              P.Model(raw=H.synth_model("v2_xvector", seed=124)[0].to_bytes(True), nnet_config=line).pack(),
              P.Model(raw=net.to_bytes(True), nnet_config="output-node name=output input=tdnn7.affine").pack()]
    assert len({fp(a)} | {fp(o) for o in others}) == 4


def test_the_reference_feature_pipeline_is_recognised_as_text_and_nothing_else_is():
    """csrc/fuse_pipe.h: the one rspecifier string every extraction script of the reference builds
    (extract_xvectors_new.sh:79, extract_xvectors.sh:73, extract_output_new.sh:75, extract_cvectors_with_am.sh:92,
    extract_cvectors_with_embedding.sh:81, extract_log_post.sh:68-70) is recognised - so that nnet3-xvector-compute can run its
    two stages on the device - and anything that is not EXACTLY that pipeline is left to the shell."""
    R = P.recognize_feature_pipeline
    ref = ("ark:apply-cmvn-sliding --norm-vars=false --center=true --cmn-window=300 scp:data/sre10/split8/3/feats.scp ark:- | "
           "select-voiced-frames ark:- scp,s,cs:data/sre10/split8/3/vad.scp ark:- |")
    assert R(ref) == {"feats": "scp:data/sre10/split8/3/feats.scp", "vad": "scp,s,cs:data/sre10/split8/3/vad.scp", "cmn_window": 300,
                      "min_cmn_window": 100, "center": True}
    # the text the scripts of the reference hold, with their shell variables filled in: every one of them is recognised
    import glob
    import re
    root = "/root/reference/egs/sre/v2/sid"
    if os.path.isdir(root):
        seen = 0
        for f in glob.glob(root + "/nnet3*/**/*.sh", recursive=True):
            for m in re.finditer(r'feats?="(ark:apply-cmvn-sliding[^"]*)"', open(f).read()):
                s = m.group(1)
                for var, val in (("${sdata}", "data/x/split4/1"), ("$sdata", "data/x/split4/1"), ("$norm_vars", "false"),
                                 ("$center", "true"), ("$cmn_window", "300")):
                    s = s.replace(var, val)
                got = R(s)
                assert got and got["feats"] == "scp:data/x/split4/1/feats.scp" and got["cmn_window"] == 300 and got["center"], (f, s, got)
                assert got["vad"] in ("", "scp,s,cs:data/x/split4/1/vad.scp"), (f, got)
                seen += 1
        assert seen >= 10, seen
    # variations the device front-end implements
    assert R("ark:apply-cmvn-sliding --norm-vars=false --center=false --cmn-window=200 --min-cmn-window=50 ark:/x/raw.ark ark:- |") == \
        {"feats": "ark:/x/raw.ark", "vad": "", "cmn_window": 200, "min_cmn_window": 50, "center": False}
    assert R("ark:/opt/kaldi/src/featbin/apply-cmvn-sliding --cmn_window=300 --norm_vars=false scp:f.scp ark:- | "
             "/opt/kaldi/src/ivectorbin/select-voiced-frames ark:- ark:vad.ark ark:- |")["vad"] == "ark:vad.ark"
    assert R("ark:apply-cmvn-sliding scp:f.scp ark:- |")["cmn_window"] == 600      # the tool's own defaults
    # ... and everything else is a command line
    for other in ("scp:feats.scp", "ark:feats.ark", "ark:cat feats.ark |",
                  ref.replace("--norm-vars=false", "--norm-vars=true"),               # variance normalisation: not implemented
                  ref.replace("--cmn-window=300", "--cmn-window=300 --max-warnings=5"),   # an option that is not understood
                  ref.replace("--center=true", "--center=maybe"),
                  ref.replace("scp:data/sre10/split8/3/feats.scp", "scp:-"),           # reads another stream
                  ref.replace("scp:data/sre10/split8/3/feats.scp", "scp:filter.pl a b |"),
                  ref.replace("select-voiced-frames", "select-voiced-frames --verbose=2"),
                  ref.replace("select-voiced-frames", "some-other-tool"),
                  ref[:-1] + "| copy-feats ark:- ark:- |",                             # a third stage
                  ref.replace("ark:- |", "ark:- 2>/dev/null |"),                       # a redirection
                  ref.replace("scp:data", "scp:$sdata"), ref.replace("apply-cmvn-sliding", "'apply-cmvn-sliding'"),
                  ref.replace("ark:apply", "ark,s,cs:apply"), ref[:-1], "ark:apply-cmvn-sliding --norm-vars=false ark:- |"):
        assert R(other) is None, other
