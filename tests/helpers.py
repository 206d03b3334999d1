"""Shared test helpers: synthetic models from the golden graph texts, synthetic features."""
import functools
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import nnet3_model as nm  # noqa: E402
from oracle import xvector_oracle as xo  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
PKG_NAME = "speaker-embedding-with-phonetic-information_amd"


def pkg():
    return importlib.import_module(PKG_NAME)


def config_text(name):
    return open(os.path.join(GOLDEN, "configs", name + ".config")).read()


# embedding node per topology (v2/run_sre10.sh:201 "tdnn6.affine"; v5/run_sre10.sh:83 "tdnn6_xvec.affine")
TOPOLOGIES = {
    "v2_xvector": (["v2_xvector"], "tdnn6.affine"),
    "v3_multitask": (["v3_multitask"], "tdnn6_xvec.affine"),
    "v4_cvector": (["am", "v4_cvector"], "tdnn6_xvec.affine"),
    "v5_cvector": (["am", "v5_cvector"], "tdnn6_xvec.affine"),
    "pa_wo_pretrain": (["pa_wo_pretrain"], "tdnn6.affine"),
}


@functools.lru_cache(maxsize=None)
def synth_model(topology="v2_xvector", seed=123):
    cfgs, node = TOPOLOGIES[topology]
    net = nm.synthesize([config_text(c) for c in cfgs], seed=seed)
    return net, "output-node name=output input=%s" % node


def tiny_config(feat_dim=5, w1=8, w2=12, pool=16, emb=8):
    """A small net in the same grammar as the reference graphs (for fast oracle comparisons)."""
    def layer(name, inp, k, n):
        return ("component name={0}.affine type=NaturalGradientAffineComponent input-dim={2} output-dim={3} max-change=0.75\n"
                "component-node name={0}.affine component={0}.affine input={1}\n"
                "component name={0}.relu type=RectifiedLinearComponent dim={3} self-repair-scale=1e-05\n"
                "component-node name={0}.relu component={0}.relu input={0}.affine\n"
                "component name={0}.batchnorm type=BatchNormComponent dim={3} target-rms=1.0\n"
                "component-node name={0}.batchnorm component={0}.batchnorm input={0}.relu\n").format(name, inp, k, n)
    t = "input-node name=input dim=%d\n" % feat_dim
    t += layer("tdnn1", "Append(Offset(input, -2), Offset(input, -1), input, Offset(input, 1), Offset(input, 2))",
               5 * feat_dim, w1)
    t += layer("tdnn2", "Append(Offset(tdnn1.batchnorm, -2), tdnn1.batchnorm, Offset(tdnn1.batchnorm, 2))", 3 * w1, w2)
    t += layer("tdnn3", "Append(Offset(tdnn2.batchnorm, -3), tdnn2.batchnorm, Offset(tdnn2.batchnorm, 3))", 3 * w2, w2)
    t += layer("tdnn4", "tdnn3.batchnorm", w2, w2)
    t += layer("tdnn5", "tdnn4.batchnorm", w2, pool)
    t += ("component name=stats-extraction-0-10000 type=StatisticsExtractionComponent input-dim=%d input-period=1 "
          "output-period=1 include-variance=true\n"
          "component-node name=stats-extraction-0-10000 component=stats-extraction-0-10000 input=tdnn5.batchnorm\n"
          "component name=stats-pooling-0-10000 type=StatisticsPoolingComponent input-dim=%d input-period=1 "
          "left-context=0 right-context=10000 num-log-count-features=0 output-stddevs=true\n"
          "component-node name=stats-pooling-0-10000 component=stats-pooling-0-10000 input=stats-extraction-0-10000\n"
          ) % (pool, 2 * pool + 1)
    t += layer("tdnn6", "Round(stats-pooling-0-10000, 1)", 2 * pool, emb)
    t += "output-node name=output input=tdnn6.batchnorm objective=linear\n"
    return t


def features(i, T, dim=23):
    return xo.synthetic_features(i, T, dim)


def pack(utts):
    offs = np.zeros(len(utts) + 1, dtype=np.int32)
    offs[1:] = np.cumsum([u.shape[0] for u in utts])
    return np.concatenate(utts, axis=0).astype(np.float32), offs


def rel_err(a, b):
    """max |a-b| over max |b| per row - the 'relative on the embedding vector' measure used everywhere."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b).max(axis=-1) / np.abs(b).max(axis=-1)))
