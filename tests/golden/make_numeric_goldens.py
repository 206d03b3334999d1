#!/usr/bin/env python3
"""Generate tests/golden/numeric_goldens.npz - numeric known-answer vectors for the hot path.

The reference holds no numeric fixtures and Kaldi cannot be run (parity
UNPINNED).  These goldens therefore pin the oracle against an INDEPENDENT formulation of the same published
semantics, written here with torch primitives (conv1d with dilation for the spliced affines, mean / biased std
for the pooling) rather than with the generic graph evaluator of oracle/xvector_oracle.py:

  * tiny nets (feature dim 5, widths 8/12, T in {15,16,25,40}) - both formulations in fp64 agree to <= 1e-12;
  * full-size v2 x-vector and v5 c-vector models (synthetic weights, seed 123), T in {25, 400}: fp64 embeddings;
  * a chunk-loop case (T=920, chunk 300, min 25, both --pad-input values).

Everything is regenerated from seeds (models: oracle/nnet3_model.synthesize(seed); features:
oracle/xvector_oracle.synthetic_features(i, T)), so the .npz stores only the expected outputs.
Run:  python tests/golden/make_numeric_goldens.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402


def conv_formulation(net, feats, layers, pooled, emb):
    """x-vector style chain via torch.conv1d (valid convolution == nnet3's 'computable frames only')."""
    torch.set_default_dtype(torch.float64)
    x = torch.from_numpy(np.asarray(feats, np.float64)).T[None]  # [1, D, T]

    def affine_conv(x, name, offsets):
        c = net.components[name + ".affine"]
        W = torch.from_numpy(np.asarray(c.f["linear"], np.float64))
        b = torch.from_numpy(np.asarray(c.f["bias"], np.float64))
        N, K = W.shape
        D = x.shape[1]
        assert K == D * len(offsets)
        if len(offsets) == 1:
            w = W.reshape(N, 1, D).permute(0, 2, 1)
            dil = 1
        else:
            dil = offsets[1] - offsets[0]
            assert all(offsets[i + 1] - offsets[i] == dil for i in range(len(offsets) - 1))
            w = W.reshape(N, len(offsets), D).permute(0, 2, 1)  # [N, D, taps]; tap j multiplies frame t+off_j
        return torch.nn.functional.conv1d(x, w.contiguous(), b, dilation=dil)

    def relu_bn(x, name):
        c = net.components[name + ".batchnorm"]
        var = torch.from_numpy(np.asarray(c.f["stats_var"], np.float64))
        mean = torch.from_numpy(np.asarray(c.f["stats_mean"], np.float64))
        s = c.f.get("target_rms", 1.0) / torch.sqrt(var + c.f.get("epsilon", 1e-3))
        return (torch.relu(x) - mean[None, :, None]) * s[None, :, None]

    for name, offsets in layers:
        x = relu_bn(affine_conv(x, name, offsets), name)
    mu = x.mean(dim=2)
    sd = torch.sqrt(torch.clamp(x.var(dim=2, unbiased=False), min=1e-10))
    st = torch.cat([mu, sd], dim=1)
    c = net.components[emb + ".affine"]
    out = st @ torch.from_numpy(np.asarray(c.f["linear"], np.float64)).T + torch.from_numpy(np.asarray(c.f["bias"], np.float64))
    return out.numpy()[0]


XVEC_LAYERS = [("tdnn1", [-2, -1, 0, 1, 2]), ("tdnn2", [-2, 0, 2]), ("tdnn3", [-3, 0, 3]), ("tdnn4", [0]), ("tdnn5", [0])]


def main():
    out = {}
    # ---- tiny nets ---------------------------------------------------------------------------------
    net = H.nm.synthesize(H.tiny_config(), seed=5)
    net.apply_nnet_config("output-node name=output input=tdnn6.affine")
    ev = H.xo.GraphEvaluator(net, np.float64)
    for T in (15, 16, 25, 40):
        x = H.features(T, T, 5)
        a = ev.compute(x)[0]
        b = conv_formulation(net, x, XVEC_LAYERS, "tdnn5", "tdnn6")
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(a).max()), (T, np.abs(a - b).max())
        out["tiny_T%d" % T] = a
    # ---- full-size v2 ------------------------------------------------------------------------------
    net2, line = H.synth_model("v2_xvector")
    n2 = H.nm.Nnet3.from_bytes(net2.to_bytes(True))
    n2.apply_nnet_config(line)
    ev2 = H.xo.GraphEvaluator(n2, np.float64)
    for T in (25, 400):
        x = H.features(T, T)
        a = ev2.compute(x)[0]
        b = conv_formulation(n2, x, XVEC_LAYERS, "tdnn5", "tdnn6")
        assert np.abs(a - b).max() <= 1e-11 * np.abs(a).max(), (T, np.abs(a - b).max())
        out["v2_T%d" % T] = a
    # chunk loop (App. B.5)
    x = H.features(920, 920)   # chunks of 300,300,300 and a 20-frame tail (< min_chunk_size 25)
    out["v2_chunk300_pad"] = H.xo.extract_xvector(ev2, x, 300, 25, True)
    out["v2_chunk300_nopad"] = H.xo.extract_xvector(ev2, x, 300, 25, False)
    # ---- full-size v5 c-vector (two branches, Append(tdnn4_xvec, tdnn5)) --------------------------------
    net5, line5 = H.synth_model("v5_cvector")
    n5 = H.nm.Nnet3.from_bytes(net5.to_bytes(True))
    n5.apply_nnet_config(line5)
    ev5 = H.xo.GraphEvaluator(n5, np.float64)
    for T in (25, 400):
        out["v5_T%d" % T] = ev5.compute(H.features(T, T))[0]
    np.savez_compressed(os.path.join(HERE, "numeric_goldens.npz"), **out)
    for k, v in out.items():
        print("%-20s dim %d  |max| %.4f" % (k, v.shape[0], np.abs(v).max()))


if __name__ == "__main__":
    main()
