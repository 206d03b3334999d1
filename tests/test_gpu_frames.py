"""GPU parity tests of frame-level outputs (what the reference obtains from `nnet3-compute`): senone log-posteriors of
the AM / multitask heads and bottleneck features, one output row per input frame with edge-frame replication."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _case(cfgs, out_node, head_stddev=1.0, seed=11):
    P = H.pkg()
    net = H.nm.synthesize([H.config_text(c) for c in cfgs], seed=seed, head_stddev=head_stddev)
    line = "output-node name=output input=%s" % out_node
    model = P.Model(raw=net.to_bytes(True), nnet_config=line)
    assert model.info.output_is_segment == 0
    n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True))
    n2.apply_nnet_config(line)
    return P, model, H.xo.GraphEvaluator(n2, np.float32)


@pytest.mark.parametrize("cfgs,node,dim", [
    (["am"], "output.log-softmax", 5139),                 # train_am.sh:31-37 senone head (fixture: 5139 targets)
    (["am"], "tdnn5.batchnorm", 128),                     # extract_bn.sh: 128-d phonetic bottleneck features
    (["v3_multitask"], "output_am.log-softmax", 3856),    # config 5: multitask senone head
    (["v3_multitask"], "tdnn4_xvec.batchnorm", 512),      # an inner frame-level node
])
def test_frame_level_parity(cfgs, node, dim):
    P, model, ev = _case(cfgs, node)
    assert model.info.output_dim == dim
    ctx = P.Context(model, precision=P.PREC_BF16X3)
    lens = [200, 1, 37, 600, 16]
    utts = [H.features(40 + i, T) for i, T in enumerate(lens)]
    feats, offs = H.pack(utts)
    out = ctx.forward_batch(feats, offs)
    assert out.shape == (sum(lens), dim)
    for i, u in enumerate(utts):
        ref = H.xo.compute_all_frames(ev, u)
        got = out[offs[i]:offs[i + 1]]
        if node.endswith("log-softmax"):
            # log-posteriors (values of order -100 with these synthetic heads): relative to the row's largest magnitude,
            # and each row must still be a normalised distribution
            assert H.rel_err(got, ref) < TOL, (i, H.rel_err(got, ref))
            assert np.abs(np.exp(got.astype(np.float64)).sum(axis=1) - 1).max() < 1e-3
        else:
            assert H.rel_err(got, ref) < TOL, (i, H.rel_err(got, ref))


def test_frame_level_fp16_mode_config5():
    # BASELINE config 5: multitask net incl. senone head, fp16 MFMA, variable-length 200-600 frame chunks
    P, model, ev = _case(["v3_multitask"], "output_am.log-softmax")
    ctx = P.Context(model, precision=P.PREC_FP16)
    rng = np.random.default_rng(5)
    lens = [int(t) for t in rng.integers(200, 601, 6)]
    utts = [H.features(90 + i, T) for i, T in enumerate(lens)]
    feats, offs = H.pack(utts)
    out = ctx.forward_batch(feats, offs)
    ref = np.concatenate([H.xo.compute_all_frames(ev, u) for u in utts])
    # single-pass fp16: not the parity mode; bound loosely, the measured value is what bench reports
    assert H.rel_err(out, ref) < 5e-3
    assert (out.argmax(1) == ref.argmax(1)).mean() > 0.97
    # in this mode the head's logits are a 16-bit plane (engine.h logits16); the normalisation itself is fp32: every row of
    # log-posteriors sums to one as exactly as fp32 allows, whatever the rounding of the logits
    lse = np.log(np.exp(out.astype(np.float64)).sum(1))
    assert np.max(np.abs(lse)) < 2e-5, np.max(np.abs(lse))


def test_pooled_log_softmax_output():
    # the unedited x-vector net: speaker log-posteriors after tdnn7 (pooled, LogSoftmax over 5139 classes)
    P = H.pkg()
    net = H.nm.synthesize(H.config_text("v2_xvector"), seed=3, head_stddev=1.0)
    model = P.Model(raw=net.to_bytes(True))
    assert model.info.output_is_segment == 1 and model.info.output_dim == 5139
    ctx = P.Context(model)
    ev = H.xo.GraphEvaluator(net, np.float32)
    utts = [H.features(70 + i, T) for i, T in enumerate([300, 150])]
    feats, offs = H.pack(utts)
    out = ctx.forward_batch(feats, offs)
    ref = np.stack([ev.compute(u)[0] for u in utts])
    assert H.rel_err(out, ref) < TOL
    assert np.abs(np.exp(out.astype(np.float64)).sum(axis=1) - 1).max() < 1e-3


def test_first_forward_pass_of_a_context_on_a_busy_gpu():
    """A context's first forward pass allocates and zero-fills its activation planes.  hipMemset returns before the fill has
    run, on the null stream, which the engine's non-blocking streams are not ordered behind: with the chip busy (another
    stream, another context - the recipes start four processes per GPU) the first pass could run beside the clearing of its
    own buffers (round 4, tools/stress_frames.py: 2 runs in 5, in every arithmetic, errors up to 390 in bottleneck features).
    Fresh contexts from two threads under a noise stream: the first result of each equals its later ones."""
    import threading
    torch = pytest.importorskip("torch")
    P, model, ev = _case(["v3_multitask"], "tdnn5_am.batchnorm")
    lens = [int(t) for t in np.random.default_rng(5).integers(200, 601, 24)]
    feats, offs = H.pack([H.features(700 + i, T) for i, T in enumerate(lens)])
    stop, bad = [], []

    def noise():
        st = torch.cuda.Stream()
        a = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
        with torch.cuda.stream(st):
            while not stop:
                for _ in range(4):
                    (a @ a)
                st.synchronize()

    def worker(tag):
        for rep in range(8):
            ctx = P.Context(model, precision=P.PREC_FP16X3 if rep % 2 else P.PREC_FP16)
            first = ctx.forward_batch(feats, offs)
            for k in range(2):
                if not np.array_equal(first, ctx.forward_batch(feats, offs)):
                    bad.append((tag, rep, k))
            ctx.close()
    tn = threading.Thread(target=noise)
    tn.start()
    try:
        ts = [threading.Thread(target=worker, args=("t%d" % i,)) for i in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    finally:
        stop.append(1)
        tn.join()
    assert not bad, bad


@pytest.mark.parametrize("prec", ["fp16", "fp16x3"])
def test_config5_full_size_properties(prec):
    """BASELINE config 5 at full size on one GPU: the multitask net with its senone head (prepare_nnet3_xconfig.sh:53-59),
    256 chunks of 200-600 frames (seed 5, as bench.py draws them), frame-level log-posteriors - ~100 k rows x 3856 columns.
    Too large for the oracle, so size-independent properties, on the device buffers:
      * 32 distinct chunks, each 8 times at shuffled positions: equal inputs give equal rows wherever they sit in the batch;
      * three chunks computed ALONE give the bits they have inside the full batch (other tile counts, other kernel variants);
      * every row is a normalised distribution (logsumexp = 0 to fp32 accuracy), finite, with no positive entry;
      * two chunks against the fp32 oracle: parity tolerance in fp16x3 (nnet3-compute's default arithmetic here), the loose
        single-pass bound in fp16 (the mode config 5 names)."""
    torch = pytest.importorskip("torch")
    P, model, ev = _case(["v3_multitask"], "output_am.log-softmax")
    ctx = P.Context(model, precision=P.PRECISIONS[prec])
    rng = np.random.default_rng(5)
    lens32 = [int(t) for t in rng.integers(200, 601, 32)]
    pool = [H.features(900 + i, T) for i, T in enumerate(lens32)]
    order = np.concatenate([rng.permutation(32) for _ in range(8)])
    utts = [pool[i] for i in order]
    feats, offs = H.pack(utts)
    S = model.info.output_dim
    f_dev = torch.from_numpy(feats).cuda()
    out = torch.empty(int(offs[-1]), S, dtype=torch.float32, device="cuda")
    ctx.forward_batch_device(f_dev.data_ptr(), offs, out.data_ptr(), S, None)
    ctx.synchronize()
    assert bool(torch.isfinite(out).all()) and float(out.max()) <= 0.0
    lse = torch.logsumexp(out.double(), dim=1)
    assert float(lse.abs().max()) < 2e-5, float(lse.abs().max())
    first = {}
    for pos, i in enumerate(order):
        rows = out[int(offs[pos]):int(offs[pos + 1])]
        if i in first:
            assert torch.equal(rows, first[i]), (pos, int(i))
        else:
            first[int(i)] = rows
    for i in (0, 7, 31):
        solo = ctx.forward_batch(pool[i], [0, lens32[i]])
        assert np.array_equal(solo, first[i].cpu().numpy()), i
    for i in (3, 20):
        ref = H.xo.compute_all_frames(ev, pool[i])
        e = H.rel_err(first[i].cpu().numpy(), ref)
        assert e < (TOL if prec == "fp16x3" else 5e-3), (prec, i, e)
