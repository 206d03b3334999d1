"""CPU tests: the oracle against the committed golden fixtures.

* tests/golden/configs/*.config - graph text emitted by the REFERENCE's own xconfig library (the only piece of
  reference code that can run; it pins topology, not arithmetic) - must parse, and the derived contexts / pooled
  frame counts / MAC counts must equal the tables of SURVEY.md App. A.3 / BASELINE.md §2.
* tests/golden/numeric_goldens.npz - fp64 embeddings cross-checked at generation time against an independent
  torch-conv1d formulation (tests/golden/make_numeric_goldens.py).
Parity with Kaldi itself is UNPINNED (Kaldi is not vendored and cannot run here); these tests pin the oracle to
what *can* be pinned."""
import os

import numpy as np
import pytest

import helpers as H

G = np.load(os.path.join(H.GOLDEN, "numeric_goldens.npz"))


@pytest.mark.parametrize("name", ["v2_xvector", "am", "v4_cvector", "v5_cvector", "v3_multitask", "v3_2share",
                                  "v3_3share", "v3_4share", "pa_wo_pretrain"])
def test_config_goldens_parse(name):
    text = H.config_text(name)
    kinds = {}
    for line in text.splitlines():
        p = H.nm.parse_config_line(line)
        assert p is not None
        kind, kv = p
        kinds[kind] = kinds.get(kind, 0) + 1
        if "input" in kv and kind in ("component-node", "output-node"):
            d = H.nm.parse_descriptor(kv["input"])
            assert repr(d).replace(" ", "") == kv["input"].replace(" ", "")   # round trip of the descriptor text
    assert kinds.get("output-node", 0) >= 1 and kinds["component"] == kinds["component-node"]


CONTEXT = {"v2_xvector": (7, 7), "v3_multitask": (7, 7), "v4_cvector": (13, 7), "v5_cvector": (13, 7),
           "pa_wo_pretrain": (13, 7)}


@pytest.mark.parametrize("topology", sorted(CONTEXT))
def test_context_and_pooled_frames(topology):
    net, line = H.synth_model(topology)
    n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True))
    n2.apply_nnet_config(line)
    ev = H.xo.GraphEvaluator(n2, np.float32)
    assert ev.context() == CONTEXT[topology]
    T = 60
    ev.compute(H.features(1, T))
    assert ev.pooled_frames == T - sum(CONTEXT[topology])          # T-14 / T-20
    with pytest.raises(Exception):
        ev.compute(H.features(1, sum(CONTEXT[topology])))         # one frame short: not computable, never padded


def test_numeric_goldens_tiny():
    net = H.nm.synthesize(H.tiny_config(), seed=5)
    net.apply_nnet_config("output-node name=output input=tdnn6.affine")
    ev = H.xo.GraphEvaluator(net, np.float64)
    for T in (15, 16, 25, 40):
        got = ev.compute(H.features(T, T, 5))[0]
        assert np.abs(got - G["tiny_T%d" % T]).max() < 1e-12 * np.abs(got).max()


@pytest.mark.parametrize("topology,key", [("v2_xvector", "v2"), ("v5_cvector", "v5")])
def test_numeric_goldens_full_size(topology, key):
    net, line = H.synth_model(topology)
    n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True))
    n2.apply_nnet_config(line)
    ev64, ev32 = H.xo.GraphEvaluator(n2, np.float64), H.xo.GraphEvaluator(n2, np.float32)
    for T in (25, 400):
        x = H.features(T, T)
        ref = G["%s_T%d" % (key, T)]
        assert H.rel_err(ev64.compute(x), ref[None]) < 1e-11
        assert H.rel_err(ev32.compute(x), ref[None]) < 2e-5       # what fp32 (Kaldi's BaseFloat) arithmetic costs


def test_chunk_loop_goldens():
    net, line = H.synth_model("v2_xvector")
    n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True))
    n2.apply_nnet_config(line)
    ev = H.xo.GraphEvaluator(n2, np.float64)
    x = H.features(920, 920)
    a = H.xo.extract_xvector(ev, x, 300, 25, True)
    b = H.xo.extract_xvector(ev, x, 300, 25, False)
    assert H.rel_err(a[None], G["v2_chunk300_pad"][None]) < 1e-11
    assert H.rel_err(b[None], G["v2_chunk300_nopad"][None]) < 1e-11
    assert H.rel_err(a[None], b[None]) > 1e-4                      # the padded 20-frame tail does change the average
    # degenerate cases (App. B.5)
    assert H.xo.extract_xvector(ev, x[:0], 300, 25, True) is None           # zero-length
    assert H.xo.extract_xvector(ev, x[:20], 300, 25, False) is None         # shorter than min, no padding
    assert H.xo.extract_xvector(ev, x[:20], 300, 25, True) is not None      # padded to 25 by edge replication


def test_mac_formulas_match_baseline_md():
    assert H.xo.xvector_macs(400) == 1034332160      # 2.0687 GFLOP
    assert H.xo.cvector_macs(400) == 2701279832      # 5.4026 GFLOP


def test_model_text_and_binary_round_trip():
    net = H.nm.synthesize(H.tiny_config(), seed=9)
    for binary in (True, False):
        n2 = H.nm.Nnet3.from_bytes(net.to_bytes(binary))
        assert list(n2.components) == list(net.components)
        for k, c in net.components.items():
            for f in ("linear", "bias", "stats_mean", "stats_var"):
                if f in c.f:
                    assert np.array_equal(np.asarray(c.f[f], np.float32), np.asarray(n2.components[k].f[f], np.float32))
        x = H.features(3, 30, 5)
        a = H.xo.GraphEvaluator(net, np.float64).compute(x)
        b = H.xo.GraphEvaluator(n2, np.float64).compute(x)
        # scalars such as BatchNorm's epsilon are stored as float32 in the file (0.001 -> 0.0010000000475)
        assert H.rel_err(a, b) < 1e-8
