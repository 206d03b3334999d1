"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle (numpy restatement
of the nnet3 semantics) on the same seeded inputs.  Floating point; tolerance from BASELINE.json north_star:
embeddings within 1e-4 relative in the split-precision modes (bf16x3, fp16x3, fp16x2 for long chunks, auto = the
default).  Single-pass bf16 / fp16 are opt-in modes whose measured error is bounded loosely here and reported by bench.py."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

TOL_PARITY = 1e-4           # north_star: "within 1e-4 relative on the embedding vector"
TOL_SINGLE = {1: 6e-2, 2: 8e-3}   # bf16 (8-bit mantissa) / fp16 (11-bit) single pass: sanity bounds only


def _oracle(net, cfg_line, dtype=np.float32):
    n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True))
    n2.apply_nnet_config(cfg_line)
    return H.xo.GraphEvaluator(n2, dtype)


@pytest.fixture(scope="module")
def v2():
    P = H.pkg()
    net, line = H.synth_model("v2_xvector")
    model = P.Model(raw=net.to_bytes(True), nnet_config=line)
    return P, net, line, model


def test_v2_single_chunk_parity(v2):
    P, net, line, model = v2
    ctx = P.Context(model, precision=P.PREC_BF16X3)
    ev32, ev64 = _oracle(net, line, np.float32), _oracle(net, line, np.float64)
    for T in (400, 15, 16, 25, 137):
        x = H.features(T, T)
        out = ctx.forward_batch(x, [0, T])
        ref32 = ev32.compute(x)
        ref64 = ev64.compute(x)
        assert H.rel_err(out, ref32) < TOL_PARITY, (T, H.rel_err(out, ref32))
        assert H.rel_err(out, ref64) < TOL_PARITY, (T, H.rel_err(out, ref64))


def test_v2_ragged_batch_parity_and_batch_invariance(v2):
    P, net, line, model = v2
    ctx = P.Context(model, precision=P.PREC_BF16X3)
    ev32 = _oracle(net, line, np.float32)
    lens = [400, 215, 16, 33, 400, 601, 25, 128, 129, 127, 15, 310]
    utts = [H.features(100 + i, T) for i, T in enumerate(lens)]
    feats, offs = H.pack(utts)
    out = ctx.forward_batch(feats, offs)
    ref = np.stack([ev32.compute(u)[0] for u in utts])
    assert H.rel_err(out, ref) < TOL_PARITY, H.rel_err(out, ref)
    # results must not depend on what else is in the batch: bit-for-bit (no cross-utterance arithmetic exists,
    # and the pooling reduction order depends only on the position inside the utterance)
    for i in (0, 3, 5, 10):
        solo = ctx.forward_batch(utts[i], [0, lens[i]])
        assert np.array_equal(solo[0], out[i]), i
    perm = [5, 0, 11, 2, 7]
    f2, o2 = H.pack([utts[i] for i in perm])
    out2 = ctx.forward_batch(f2, o2)
    assert np.array_equal(out2, out[perm])


def test_v2_fp16x3_parity(v2):
    P, net, line, model = v2
    ctx = P.Context(model, precision=P.PREC_FP16X3)
    ev64 = _oracle(net, line, np.float64)
    for T in (400, 15, 16, 25, 137):
        x = H.features(T, T)
        err = H.rel_err(ctx.forward_batch(x, [0, T]), ev64.compute(x))
        assert err < TOL_PARITY, (T, err)


def test_v2_auto_mode_parity_and_batch_invariance(v2):
    """XV_PREC_AUTO: chunks that pool >= 300 frames take the fast kernels (fp16 activations x fp16 weights + the
    block-scaled 4-bit residual product, XV_PREC_FP16MX), shorter ones the three-pass ones.  Every chunk stays within the parity tolerance, and which arithmetic a chunk
    gets depends on its own length only: solo == batched == permuted, bit for bit."""
    P, net, line, model = v2
    ctx = P.Context(model, precision=P.PREC_AUTO)
    ev64 = _oracle(net, line, np.float64)
    lens = [400, 215, 16, 33, 400, 601, 25, 128, 313, 314, 15, 1000]
    utts = [H.features(100 + i, T) for i, T in enumerate(lens)]
    feats, offs = H.pack(utts)
    out = ctx.forward_batch(feats, offs)
    ref = np.stack([ev64.compute(u)[0] for u in utts])
    for i, T in enumerate(lens):
        err = H.rel_err(out[i:i + 1], ref[i:i + 1])
        assert err < TOL_PARITY, (T, err)
    for i in (0, 3, 5, 9, 10):
        solo = ctx.forward_batch(utts[i], [0, lens[i]])
        assert np.array_equal(solo[0], out[i]), i
    perm = [5, 0, 11, 2, 7]
    f2, o2 = H.pack([utts[i] for i in perm])
    assert np.array_equal(ctx.forward_batch(f2, o2), out[perm])
    # all-long and all-short batches (one region only)
    for sel in ([0, 4, 5, 11], [2, 3, 6, 10]):
        f3, o3 = H.pack([utts[i] for i in sel])
        assert np.array_equal(ctx.forward_batch(f3, o3), out[sel])
    # the long chunks really took the fast arithmetic, the short ones the three-pass one
    x3 = P.Context(model, precision=P.PREC_FP16X3)
    x2 = P.Context(model, precision=P.PREC_FP16MX)
    assert np.array_equal(x2.forward_batch(utts[0], [0, 400])[0], out[0])
    assert np.array_equal(x3.forward_batch(utts[1], [0, 215])[0], out[1])
    assert not np.array_equal(x3.forward_batch(utts[0], [0, 400])[0], out[0])


@pytest.mark.parametrize("seed", [123, 7, 2024])
def test_auto_mode_error_over_models_and_utterances(seed):
    """The two-pass arithmetic's error comes from rounding activations to fp16 and shrinks with the pooled frames; its
    size depends on the model and the data.  Three independently drawn synthetic models x 12 utterances at the headline
    length (T = 400, two-pass) and at the threshold (T = 314): every embedding within the 1e-4 bar, the worst one
    recorded in the assertion message."""
    P = H.pkg()
    net, line = H.synth_model("v2_xvector", seed=seed)
    model = P.Model(raw=net.to_bytes(True), nnet_config=line)
    ctx = P.Context(model, precision=P.PREC_AUTO)
    ev64 = _oracle(net, line, np.float64)
    utts = [H.features(9000 + seed + i, 400 if i % 2 == 0 else 314) for i in range(12)]
    feats, offs = H.pack(utts)
    out = ctx.forward_batch(feats, offs)
    errs = [H.rel_err(out[i:i + 1], ev64.compute(u)) for i, u in enumerate(utts)]
    assert max(errs) < TOL_PARITY, errs
    assert max(errs) > 5e-6      # these chunks really ran the two-pass kernels
    print("seed %d: max %.3e  mean %.3e" % (seed, max(errs), float(np.mean(errs))))


@pytest.mark.parametrize("seed", [11, 12])
def test_trained_like_model_needs_the_three_pass_mode(seed):
    """Heavy-tailed weights (Student-t, nu = 3) and BatchNorm statistics that match the activations, with StatsVar
    spread over 1e-3 .. 10 (helpers.trained_like_model): the error of every mode that keeps activations in ONE fp16
    plane is set by the activation rounding, and that depends on the model - on this one fp16x2 and auto (fp16mx) land
    at 1 - 2.5e-4, above the 1e-4 bar that they meet on Kaldi's own initialisation distribution (the other tests and
    the benchmark model).  fp16x3 keeps fp32-grade results; the one-plane fast modes are opt-in, the
    command line's default is fp16mx2, which carries a 4-bit second plane for the activations
    (test_fp16mx2_is_model_independent)."""
    P = H.pkg()
    net, line = H.trained_like_model("v2_xvector", seed)
    model = P.Model(raw=net.to_bytes(True), nnet_config=line)
    ev64 = _oracle(net, line, np.float64)
    utts = [H.features(50 + i, 400) for i in range(4)]
    feats, offs = H.pack(utts)
    ref = np.stack([ev64.compute(u)[0] for u in utts])
    errs = {}
    for name in ("fp16x3", "fp16x2", "auto"):
        out = P.Context(model, precision=P.PRECISIONS[name]).forward_batch(feats, offs)
        errs[name] = max(H.rel_err(out[i:i + 1], ref[i:i + 1]) for i in range(len(utts)))
    print("trained-like model seed %d: %s" % (seed, ", ".join("%s %.2e" % kv for kv in errs.items())))
    assert errs["fp16x3"] < 1e-5, errs
    assert errs["fp16x2"] < 5e-4 and errs["auto"] < 5e-4, errs      # same order as plain fp16: bounded, but not parity-grade
    assert errs["auto"] > 2e-5                                      # the fast kernels really ran


@pytest.mark.parametrize("which", ["init_123", "trained_11", "trained_12", "v5_trained_11", "v5_trained_12", "v5_trained_13"])
def test_fp16mx2_is_model_independent(which):
    """XV_PREC_FP16MX2 corrects the fp16 rounding of the activations with a 4-bit residual plane (1.5 MFMA passes per
    product): on the heavy-tailed, BatchNorm-calibrated models where the one-plane modes land at 1.2 - 1.7e-4 it stays
    below 6e-5 at every chunk length (chunks that pool fewer than 160 frames take the three-pass arithmetic), and a
    chunk's embedding does not depend on its neighbours in the batch."""
    P = H.pkg()
    if which == "init_123":
        net, line = H.synth_model("v2_xvector", 123)
    elif which.startswith("v5_"):   # the c-vector network (phonetic branch, two-source Append: train_cvector_with_am.sh:65-89)
        net, line = H.trained_like_model("v5_cvector", int(which[-2:]))
    else:
        net, line = H.trained_like_model("v2_xvector", int(which[-2:]))
    model = P.Model(raw=net.to_bytes(True), nnet_config=line)
    ev64 = _oracle(net, line, np.float64)
    lens = [400, 137, 400, 314, 60, 400, 25, 200, 115, 400]
    utts = [H.features(9100 + i, T) for i, T in enumerate(lens)]
    feats, offs = H.pack(utts)
    ctx = P.Context(model, precision=P.PRECISIONS["fp16mx2"])
    out = ctx.forward_batch(feats, offs)
    errs = [H.rel_err(out[i:i + 1], ev64.compute(u)) for i, u in enumerate(utts)]
    print("%s: fp16mx2 max %.2e mean %.2e" % (which, max(errs), float(np.mean(errs))))
    # the deeper c-vector network accumulates more of what is left (measured 7.9e-5 at 117 pooled frames with the first
    # threshold of 100, hence 160 now): 7e-5 there, 6e-5 on the x-vector network and on every chunk of >= 300 frames
    assert max(errs) < (7e-5 if which.startswith("v5_") else 6e-5), errs
    assert max(errs[i] for i, T in enumerate(lens) if T >= 300) < 6e-5, errs
    assert max(errs[i] for i, T in enumerate(lens) if T >= 190) > 5e-6       # the long chunks really took the fast kernels
    # the short ones the three-pass ones (2e-5 on the deep c-vector network at 5 pooled frames: fp32 cancellation in
    # E[x^2] - mean^2 against the fp64 oracle, the same in every mode)
    assert max(errs[i] for i, T in enumerate(lens) if T < 160) < (3e-5 if which.startswith("v5_") else 5e-6)
    for i in (0, 1, 6):
        solo = ctx.forward_batch(*H.pack(utts[i:i + 1]))
        assert np.array_equal(solo[0], out[i])
    order = [9, 3, 7, 1, 5, 0, 8, 2, 6, 4]
    perm = ctx.forward_batch(*H.pack([utts[k] for k in order]))
    for pos, k in enumerate(order):
        assert np.array_equal(perm[pos], out[k])


def test_fp16mx2_other_topologies():
    P = H.pkg()
    for topology in ("v5_cvector", "v3_multitask"):
        net, line = H.synth_model(topology)
        model = P.Model(raw=net.to_bytes(True), nnet_config=line)
        try:
            ctx = P.Context(model, precision=P.PRECISIONS["fp16mx2"])
        except P.XvError as e:       # a layer whose sources are not whole 128-column blocks: the mode says so at pack time
            assert "fp16mx2" in str(e), str(e)
            continue
        ev64 = _oracle(net, line, np.float64)
        utts = [H.features(300 + i, T) for i, T in enumerate([400, 21, 330])]
        out = ctx.forward_batch(*H.pack(utts))
        for i, u in enumerate(utts):
            assert H.rel_err(out[i:i + 1], ev64.compute(u)) < TOL_PARITY, (topology, i)


@pytest.mark.parametrize("topology", ["v5_cvector", "v3_multitask"])
def test_other_topologies_auto_mode(topology):
    P = H.pkg()
    net, line = H.synth_model(topology)
    model = P.Model(raw=net.to_bytes(True), nnet_config=line)
    ctx = P.Context(model, precision=P.PREC_AUTO)
    ev64 = _oracle(net, line, np.float64)
    lens = [400, 21, 330]
    utts = [H.features(300 + i, T) for i, T in enumerate(lens)]
    feats, offs = H.pack(utts)
    out = ctx.forward_batch(feats, offs)
    ref = np.stack([ev64.compute(u)[0] for u in utts])
    for i, T in enumerate(lens):
        err = H.rel_err(out[i:i + 1], ref[i:i + 1])
        assert err < TOL_PARITY, (topology, T, err)


@pytest.mark.parametrize("prec", [1, 2])
def test_v2_single_pass_modes(v2, prec):
    P, net, line, model = v2
    ctx = P.Context(model, precision=prec)
    ev32 = _oracle(net, line, np.float32)
    utts = [H.features(7 + i, 400) for i in range(4)]
    feats, offs = H.pack(utts)
    out = ctx.forward_batch(feats, offs)
    ref = np.stack([ev32.compute(u)[0] for u in utts])
    err = H.rel_err(out, ref)
    assert err < TOL_SINGLE[prec], err
    assert err > 1e-6   # it really is a different arithmetic from the parity mode


@pytest.mark.parametrize("topology", ["v5_cvector", "v4_cvector", "v3_multitask", "pa_wo_pretrain", "v3_2share", "v3_3share",
                                      "v3_4share"])
@pytest.mark.parametrize("mode", ["bf16x3", "default"])
def test_other_topologies_parity(topology, mode):
    P = H.pkg()
    net, line = H.synth_model(topology)
    model = P.Model(raw=net.to_bytes(True), nnet_config=line)
    ctx = P.Context(model, precision=P.PRECISIONS[mode])
    ev32 = _oracle(net, line, np.float64)
    lens = [400, 21 if "cvector" in topology or topology == "pa_wo_pretrain" else 15, 77]
    utts = [H.features(300 + i, T) for i, T in enumerate(lens)]
    feats, offs = H.pack(utts)
    out = ctx.forward_batch(feats, offs)
    ref = np.stack([ev32.compute(u)[0] for u in utts])
    assert H.rel_err(out, ref) < TOL_PARITY, H.rel_err(out, ref)


def test_tiny_net_with_segment_level_chain():
    # tiny widths (padding paths), output taken after relu+batchnorm of the segment-level layer
    P = H.pkg()
    net = H.nm.synthesize(H.tiny_config(), seed=5)
    model = P.Model(raw=net.to_bytes(False))
    ctx = P.Context(model, precision=P.PREC_BF16X3)
    ev = H.xo.GraphEvaluator(net, np.float32)
    utts = [H.features(i, T, 5) for i, T in enumerate([15, 16, 25, 40])]
    feats, offs = H.pack(utts)
    out = ctx.forward_batch(feats, offs)
    ref = np.stack([ev.compute(u)[0] for u in utts])
    assert H.rel_err(out, ref) < TOL_PARITY, H.rel_err(out, ref)


def test_variance_floor_branch():
    # a constant pooled column: E[x^2] - mean^2 cancels, the 1e-10 floor decides (SURVEY.md §8(c) KAT 5)
    P = H.pkg()
    net = H.nm.synthesize(H.tiny_config(), seed=6)
    c = net.components["tdnn5.affine"]
    c.f["linear"][3, :] = 0.0
    c.f["bias"][3] = 0.75
    model = P.Model(raw=net.to_bytes(True))
    ctx = P.Context(model, precision=P.PREC_BF16X3)
    ev = H.xo.GraphEvaluator(net, np.float64)
    x = H.features(9, 60, 5)
    out = ctx.forward_batch(x, [0, 60])
    assert H.rel_err(out, ev.compute(x)) < TOL_PARITY


def test_chunk_loop_matches_oracle(v2):
    P, net, line, model = v2
    ctx = P.Context(model, precision=P.PREC_BF16X3)
    ev32 = _oracle(net, line, np.float32)
    lens = [1000, 20, 0, 260, 10, 14]
    utts = [H.features(500 + i, T) for i, T in enumerate(lens)]
    feats, offs = H.pack(utts)
    for chunk, minc, pad in ((300, 25, True), (300, 25, False), (-1, 25, True), (10000, 25, False)):
        out, ok = ctx.extract_utterances(feats, offs, chunk, minc, pad)
        for i, u in enumerate(utts):
            ref = H.xo.extract_xvector(ev32, u, chunk, minc, pad)
            assert ok[i] == (ref is not None), (i, chunk, minc, pad)
            if ref is not None:
                assert H.rel_err(out[i:i + 1], ref[None]) < TOL_PARITY, (i, chunk, pad)


def test_short_chunk_is_an_argument_error(v2):
    P, net, line, model = v2
    ctx = P.Context(model)
    with pytest.raises(P.XvError) as e:
        ctx.forward_batch(H.features(1, 14), [0, 14])
    assert e.value.status == 4


def test_context_from_packed_blob_is_identical(v2):
    P, net, line, model = v2
    a = P.Context(model, precision=P.PREC_BF16X3)
    b = P.Context(blob=model.pack(P.PREC_BF16X3))
    x = H.features(42, 400)
    assert np.array_equal(a.forward_batch(x, [0, 400]), b.forward_batch(x, [0, 400]))


def _kernels_that_ran(ctx, feats, offs):
    """Names of the GEMM instantiations one forward pass launches (engine profile labels: `tdnn_gemm<act>:tdnn2.batchnorm
    tdnn_gemm_kernel_sk<fp16mx2,act,8>`)."""
    ctx.set_profiling(True)
    ctx.forward_batch(feats, offs)
    rep = ctx.profile_report()
    ctx.set_profiling(False)
    return " ".join(l for (l, _, _) in rep)


@pytest.mark.parametrize("topology,mode", [("v2_xvector", "bf16x3"), ("v2_xvector", "default"), ("v2_xvector", "auto"),
                                           ("v5_cvector", "default"), ("v5_cvector", "auto")])
def test_full_size_batch_properties(topology, mode):
    """BASELINE configs 2 and 3 at full size (256 chunks x 400 frames, v2 x-vector and v5 c-vector), in the arithmetic
    that ships (`default` = XV_PREC_DEFAULT -> fp16mx2) and in the opt-in fast mode (`auto` -> fp16mx), plus the
    three-pass reference mode: size-independent properties instead of a slow oracle run -
      * identical inputs give identical outputs wherever they sit in the batch (32 distinct chunks, each 8 times);
      * every distinct chunk computed ALONE gives the same bits as inside the full batch (the stream-K kernels of the
        full batch against the per-tile kernels of a one-chunk launch);
      * 8 rows are checked against the fp64 oracle at the parity tolerance;
      * the profile report proves that the stream-K instantiation of the mode under test really ran."""
    P = H.pkg()
    net, line = H.synth_model(topology)
    model = P.Model(raw=net.to_bytes(True), nnet_config=line)
    ctx = P.Context(model, precision=P.PRECISIONS[mode])
    pool = [H.features(1000 + i, 400) for i in range(32)]
    utts = [pool[i % 32] for i in range(256)]
    feats, offs = H.pack(utts)
    out = ctx.forward_batch(feats, offs)
    assert np.all(np.isfinite(out))
    for i in range(32, 256):
        assert np.array_equal(out[i], out[i % 32]), i
    for i in (0, 5, 31):
        solo = ctx.forward_batch(*H.pack(pool[i:i + 1]))
        assert np.array_equal(solo[0], out[i]), i
    ev64 = _oracle(net, line, np.float64)
    ref = np.stack([ev64.compute(pool[i])[0] for i in range(8)])
    errs = [H.rel_err(out[i:i + 1], ref[i:i + 1]) for i in range(8)]
    print("%s %s at 256 x 400: max %.2e mean %.2e" % (topology, mode, max(errs), float(np.mean(errs))))
    assert max(errs) < TOL_PARITY, errs
    ran = _kernels_that_ran(ctx, feats, offs)
    want = {"bf16x3": "bf16x3", "default": "fp16mx2", "auto": "fp16mx"}[mode]
    # the 1.25-pass launches of these layers run the 256 x 256 x 64 kernel, the 1.5-pass ones the stream-K kernel
    want_kernel = "tdnn_gemm_kernel_p8<fp16mx,act>" if want == "fp16mx" else "tdnn_gemm_kernel_sk<%s,act,8>" % want
    assert want_kernel in ran or mode == "bf16x3", ran
    assert ("<%s," % want) in ran, ran
    if mode == "default":
        assert ctx.precision == P.PREC_FP16MX2
        # 1.5 passes: the layers without time offsets on the 256 x 256 kernel (x-vector: tdnn4, tdnn5 - K = 512), the spliced
        # ones and those whose sources are not whole 256-column tiles (c-vector tdnn5_xvec: 512 + 128) on the stream-K kernel
        want_stats = "tdnn_gemm_kernel_p8<fp16mx2,stats>" if topology == "v2_xvector" else "tdnn_gemm_kernel_sk<fp16mx2,stats,8>"
        assert want_stats in ran, ran
        if topology == "v2_xvector":
            assert "tdnn4.batchnorm tdnn_gemm_kernel_p8<fp16mx2,act>" in ran, ran


def test_calibration_sample_covers_the_whole_table(tmp_path):
    """VERDICT r03 item 4.  A table whose first 64 utterances are benign (long: the pooling averages the activation rounding of
    fp16mx over 1500 frames) and whose later ones are a different "speaker" (10x louder, 320 frames: the error of the same
    arithmetic is ~2x larger).  The reference's lists are speaker-sorted (utils/data/split_data.sh:18-21), so a head-of-list
    sample is exactly this trap: calibrated on the head alone fp16mx passes a tolerance that the rest of the job exceeds.
    xv_calibrate_table samples the whole list and keeps fp16mx2."""
    P = H.pkg()
    from oracle import kaldi_io as kio
    net, line = H.synth_model("v2_xvector", 123)
    model = P.Model(raw=net.to_bytes(True), nnet_config=line)
    head = [("spkA-%03d" % i, H.features(3000 + i, 1500)) for i in range(64)]
    tail = [("spkB-%03d" % i, 10.0 * H.features(4000 + i, 320)) for i in range(192)]
    ark = tmp_path / "feats.ark"
    kio.write_ark_matrices(str(ark), head + tail)
    ctx = P.Context(model)
    fh, oh = H.pack([x for _, x in head])
    cal_head = ctx.calibrate(fh, oh, tol=1.0)                    # what a head-of-list sample measures
    e_head = cal_head["err_mx"]
    assert cal_head["chosen"] == "fp16mx" and cal_head["tail"] >= e_head
    ctx.set_fast_mode("fp16mx2")
    cal_all = ctx.calibrate_table("ark:%s" % ark, tol=1.0)       # 64 utterances spread over all 256: 16 of A, 48 of B
    e_all = cal_all["err_mx"]
    print("head %.3g  whole list %.3g" % (e_head, e_all), cal_all)
    assert cal_all["checked_mx"] == 64
    assert e_all > 1.2 * e_head, (e_head, e_all)
    # a tolerance the head passes - on its worst chunk and on the tail its errors project (mean + 6 sd within 1.10 x the tolerance) -
    # and the job does not
    lo = max(e_head, cal_head["tail"] / 1.10)
    assert lo < e_all, (lo, e_all)
    tol = float(np.sqrt(lo * e_all))
    ctx.set_fast_mode("fp16mx2")
    assert ctx.calibrate(fh, oh, tol=tol)["chosen"] == "fp16mx"
    ctx.set_fast_mode("fp16mx2")
    assert ctx.calibrate_table("ark:%s" % ark, tol=tol)["chosen"] == "fp16mx2" and ctx.fast_mode == "fp16mx2"
    # the same through a stream: only the head can be sampled (and the log line says so)
    ctx.set_fast_mode("fp16mx2")
    assert ctx.calibrate_table("ark:cat %s |" % ark, tol=tol)["chosen"] == "fp16mx"


def test_calibration_picks_the_lighter_mode_only_where_it_is_accurate():
    """XV_PREC_DEFAULT + xv_ctx_calibrate (what the command-line tools do on the head of their table): on the
    initialisation-like model fp16mx stays within 7.5e-5 of the three-pass arithmetic and is chosen; on the heavy-tailed,
    BatchNorm-calibrated model it is at 1 - 2e-4 and the context stays in fp16mx2.  After the call the context computes
    exactly what a context of the chosen mode computes, and every embedding is within the parity bar of the fp64 oracle."""
    P = H.pkg()
    for which, want in (("init", "fp16mx"), ("trained", "fp16mx2")):
        net, line = H.synth_model("v2_xvector", 123) if which == "init" else H.trained_like_model("v2_xvector", 11)
        model = P.Model(raw=net.to_bytes(True), nnet_config=line)
        # 18 chunks the lighter mode runs fast (>= 300 pooled frames; it takes at least 16 of them to be chosen), one that only
        # fp16mx2 runs fast (200 frames: 186 pooled) and one that every mode runs three-pass (120 frames)
        utts = [H.features(60 + i, T) for i, T in enumerate([400, 400, 333, 400, 120, 400, 200] + [400, 333, 350] * 4 + [400])]
        feats, offs = H.pack(utts)
        ctx = P.Context(model)                          # XV_PREC_DEFAULT -> packed as fp16mx2
        assert ctx.precision == P.PREC_FP16MX2 and ctx.fast_mode == "fp16mx2"
        cal = ctx.calibrate(feats, offs)                # default tolerance: 7.5e-5 on the worst chunk
        print(which, cal)
        assert cal["chosen"] == want and ctx.fast_mode == want, cal
        assert cal["checked"] == 19 and cal["checked_mx"] == 18   # the 120-frame chunk runs three-pass in every mode: not compared
        assert cal["err_mx2"] < 7e-5 and (cal["err_mx"] <= 7.5e-5) == (want == "fp16mx"), cal
        out = ctx.forward_batch(feats, offs)
        twin = P.Context(model, precision=P.PRECISIONS["auto" if want == "fp16mx" else "fp16mx2"])
        if cal.get("lite_mask"):   # fp16mx2 with some layers in 1.25 passes (the next test): still within the tolerance it measured
            assert cal["err_lite"] <= 7.5e-5 and ctx.lite_mask == cal["lite_mask"], cal
            twin.set_lite_mask(cal["lite_mask"])
        assert np.array_equal(out, twin.forward_batch(feats, offs))
        ev64 = _oracle(net, line, np.float64)
        for i, u in enumerate(utts):
            assert H.rel_err(out[i:i + 1], ev64.compute(u)) < TOL_PARITY, (which, i)
        # a choice made elsewhere (the other ranks of a multi-GPU job) is applied with set_fast_mode
        other = P.Context(model)
        other.set_fast_mode(want)
        if cal.get("lite_mask"):
            other.set_lite_mask(cal["lite_mask"])
        assert np.array_equal(other.forward_batch(feats, offs), out)
    # a handful of qualifying chunks is not a measurement: the packed mode stays, whatever they show
    net, line = H.synth_model("v2_xvector", 123)
    few = P.Context(P.Model(raw=net.to_bytes(True), nnet_config=line))
    f5, o5 = H.pack([H.features(60 + i, 400) for i in range(5)])
    cal = few.calibrate(f5, o5)
    assert cal["checked_mx"] == 5 and cal["err_mx"] <= 7.5e-5 and cal["chosen"] == "fp16mx2" and few.fast_mode == "fp16mx2", cal
    # contexts that cannot switch say so and stay what they are
    c3 = P.Context(model, precision=P.PREC_FP16X3)
    cal = c3.calibrate(feats, offs)
    assert cal["checked"] == 0 and cal["chosen"] == "fp16x3"
    with pytest.raises(P.XvError):
        c3.set_fast_mode("fp16mx")


def test_calibration_mixes_the_two_fast_arithmetics_on_the_cvector_network():
    """Where fp16mx as a whole misses the tolerance (the c-vector network: 8 - 9e-5 against 7.5e-5), the calibration keeps the
    1.5-pass context and takes the second walk off the most expensive layers the tolerance allows (xv_calibration.lite_mask:
    on this network the 650-wide phonetic branch, train_am.sh:30-38).  The mixture is a property of the context like the
    mode: applied elsewhere with set_lite_mask it computes the same bits, whatever the batch."""
    P = H.pkg()
    net, line = H.synth_model("v5_cvector", 123)
    model = P.Model(raw=net.to_bytes(True), nnet_config=line)
    NL = 60   # a sample of the tools' size: the acceptance rule also looks at the spread of the errors (xv_calibration.tail)
    utts = [H.features(80 + i, T) for i, T in enumerate([400, 333, 350] * (NL // 3) + [200, 120])]
    feats, offs = H.pack(utts)
    ctx = P.Context(model)
    tol = 7.5e-5
    cal = ctx.calibrate(feats, offs, tol)
    print(cal)
    print(model.describe())
    assert cal["checked_mx"] == NL
    if cal["chosen"] == "fp16mx":
        # should the lighter mode pass outright on this sample (8.0 - 8.3e-5 on the bench's 64): the same question with a
        # tolerance it misses
        assert cal["err_mx"] <= tol and ctx.lite_mask == 0
        tol = 0.9 * cal["err_mx"]
        ctx = P.Context(model)
        cal = ctx.calibrate(feats, offs, tol)
        print(cal)
    assert cal["chosen"] == "fp16mx2" and cal["err_mx"] > tol and cal["err_mx2"] <= tol, cal
    mask = cal.get("lite_mask", 0)
    assert mask and ctx.lite_mask == mask and 0 < cal["err_lite"] <= tol, cal
    # what was adopted projects a tail (mean + 6 sd of the per-chunk error over the held-out half) inside 1.10 x the tolerance
    assert cal["err_holdout"] <= cal["tail"] <= 1.10 * tol * (1 + 1e-6), cal
    out = ctx.forward_batch(feats, offs)
    ev64 = _oracle(net, line, np.float64)
    for i in (0, 1, 2, NL, NL + 1):
        assert H.rel_err(out[i:i + 1], ev64.compute(utts[i])) < TOL_PARITY, i
    # not the plain 1.5-pass arithmetic, and not the 1.25-pass one either
    plain = P.Context(model, precision=P.PRECISIONS["fp16mx2"])
    assert not np.array_equal(plain.forward_batch(feats, offs)[:NL], out[:NL])
    # the same mixture on another context; an utterance's embedding does not depend on its batch
    other = P.Context(model)
    other.set_lite_mask(mask)
    assert other.lite_mask == mask and np.array_equal(other.forward_batch(feats, offs), out)
    f1, o1 = H.pack(utts[3:5])
    assert np.array_equal(other.forward_batch(f1, o1), out[3:5])
    # bits of layers that cannot run the 1.25-pass arithmetic are dropped; a change of mode clears the mixture
    other.set_lite_mask((1 << 62) - 1)
    kept = other.lite_mask
    assert kept and kept & mask == mask and kept != (1 << 62) - 1
    assert H.rel_err(other.forward_batch(f1, o1), out[3:5]) < 2e-4
    other.set_fast_mode("fp16mx2")          # the mode it is in: the mixture goes, the plain mode stays
    assert other.lite_mask == 0 and np.array_equal(other.forward_batch(f1, o1), plain.forward_batch(f1, o1))
    other.set_fast_mode("fp16mx")
    assert other.lite_mask == 0
    with pytest.raises(P.XvError):
        other.set_lite_mask(mask)          # only inside the 1.5-pass context
    # a job with many chunks between the two thresholds (160 .. 299 pooled frames: fast in plain fp16mx2, three-pass in a
    # mixture) keeps the plain mode: the mixture would slow those down by more than it saves on the others
    ragged = utts[:NL] + [H.features(300 + i, 200) for i in range(NL // 3)]
    fr, orr = H.pack(ragged)
    c2 = P.Context(model)
    cal2 = c2.calibrate(fr, orr, tol)
    assert cal2["checked"] == NL + NL // 3 and cal2["checked_mx"] == NL and cal2["chosen"] == "fp16mx2" and not cal2.get("lite_mask"), cal2


def test_bn_fold_opt_in_is_the_same_function(monkeypatch):
    """XVEC_DEBUG=bn_fold=1 (csrc/program.cc FoldBatchNormIntoConsumers, profiles/r05_bn_fold.md): the folded program computes the same
    function - the parity-grade arithmetic agrees with the fp64 oracle as closely as the unfolded one, the shipped default stays
    within the bar - on the x-vector and on the two-branch c-vector graph (a consumer with two folded sources)."""
    P = H.pkg()
    for topology in ("v2_xvector", "v5_cvector"):
        net, line = H.synth_model(topology)
        monkeypatch.setenv("XVEC_DEBUG", "bn_fold=1")
        model = P.Model(raw=net.to_bytes(True), nnet_config=line)
        monkeypatch.delenv("XVEC_DEBUG")
        assert "bn(folded)" in model.describe()
        ev64 = _oracle(net, line, np.float64)
        utts = [H.features(4400 + i, T) for i, T in enumerate([400, 21, 137, 333])]
        feats, offs = H.pack(utts)
        ref = np.stack([ev64.compute(u)[0] for u in utts])
        e3 = H.rel_err(P.Context(model, precision=P.PREC_FP16X3).forward_batch(feats, offs), ref)
        ed = H.rel_err(P.Context(model).forward_batch(feats, offs), ref)
        print("%s folded: fp16x3 %.2e, default %.2e" % (topology, e3, ed))
        assert e3 < 3e-6 and ed < TOL_PARITY, (topology, e3, ed)
