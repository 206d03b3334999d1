"""CPU study (fp64, no GPU, no product code): does folding each BatchNorm into its CONSUMER change the error of the 1.25-pass
arithmetic (fp16mx)?  VERDICT r04 item 3.

Today a frame-level layer stores y = s * relu(z) + o as its fp16 plane (.affine -> .relu -> .batchnorm,
steps/libs/nnet3/xconfig/basic_layers.py:778-816).  Folded, it would store r = relu(z) * 2^e (e = the power of two of s, per
column: exact) and the consumer would use W' = W * diag(s / 2^e), b' = b + sum_j W_j . o (legal because nnet3 never pads: every
consumed frame is a computed one).  Planes then hold exact zeros where the ReLU cut (~half the entries), which the chip rewards
with clock (+5 % measured on such data); the planes epilogue loses an FMA per value.  Open question before building it: the fp16
ROUNDING of the plane - relative to s * r it is the same 2^-12, but y = s * r + o is the centred variable, so |y| < |s r| for
most entries above the cut (smaller absolute error today) while entries at the cut carry 2^-12 |o| today and nothing folded.

Schemes (tdnn1's input layer and everything behind the pooling are three-pass = exact here, like the product; the pooled
layer's activations are never rounded to fp16 in either scheme - its statistics come from the fp32 accumulators):
  mx        plane = fp16(s relu(z) + o);   z' = plane . fp16(W) + q4(plane) . q4(W - fp16(W)) + b
  mx_fold   plane = fp16(relu(z) 2^e);     z' = plane . fp16(W') + q4(plane) . q4(W' - fp16(W')) + b'
  x2 / x2_fold   the same without weight rounding (fp16x2: activations' rounding alone)

usage: sim_bn_fold.py [chunks]"""
import os
import struct
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import helpers as H  # noqa: E402
from oracle.export_program import export_program  # noqa: E402
from sim_precision2 import GRIDS, q16, q_block_scaled  # noqa: E402


def load_program(path):
    b = open(path, "rb").read()
    assert b[:8] == b"XVORACLE"
    input_dim, n_layers, pooled, out_layer, floor = struct.unpack_from("<iiiif", b, 8)
    p = 28
    layers = []
    for _ in range(n_layers):
        nsrc, = struct.unpack_from("<i", b, p)
        p += 4
        src = []
        for _ in range(nsrc):
            src.append(struct.unpack_from("<iii", b, p))
            p += 12
        k, n, relu, bn, seg = struct.unpack_from("<iiiii", b, p)
        p += 20
        w = np.frombuffer(b, np.float32, n * k, p).reshape(n, k).astype(np.float64)
        p += 4 * n * k
        vec = []
        for _ in range(3):
            vec.append(np.frombuffer(b, np.float32, n, p).astype(np.float64))
            p += 4 * n
        layers.append(dict(src=src, w=w, b=vec[0], scale=vec[1], offset=vec[2], relu=relu, bn=bn, seg=seg))
    return layers, pooled, out_layer, floor


def q4_group16(x):
    """The product's 4-bit copy of a frame fragment set: e2m1, one power-of-two scale per 16-row group (over every column of
    the source), 2^(e - 2) with e the exponent of the group's maximum, saturating at the e2m1 maximum."""
    n = x.shape[0]
    pad = (-n) % 16
    xp = np.pad(x, [(0, pad), (0, 0)])
    g = xp.reshape(-1, 16, x.shape[1])
    m = np.abs(g).max(axis=(1, 2), keepdims=True)
    e = np.floor(np.log2(np.where(m > 0, m, 1.0)))
    sc = 2.0 ** (e - 2)
    q = (np.sign(g) * np.minimum(np.abs(_qgrid(g / sc)), 6.0)) * sc
    return q.reshape(-1, x.shape[1])[:n]


def _qgrid(v):
    from sim_precision2 import qgrid
    return qgrid(v, GRIDS["4"])


def plane_residual4(y, yq):
    """The 1.5-pass mode's 4-bit image of what the fp16 rounding of a plane dropped: e2m1, one scale per row and 64 columns,
    2^(E - 13) with E the exponent of the block's largest |y|."""
    n, c = y.shape
    pad = (-c) % 64
    yp = np.pad(y, [(0, 0), (0, pad)]).reshape(n, -1, 64)
    rp = np.pad(y - yq, [(0, 0), (0, pad)]).reshape(n, -1, 64)
    m = np.abs(yp).max(axis=2, keepdims=True)
    E = np.floor(np.log2(np.where(m > 0, m, 1.0)))
    sc = 2.0 ** (E - 13)
    return (_qgrid(rp / sc) * sc).reshape(n, -1)[:, :c]


def forward(prog, feats, scheme, tau=1e30):
    layers, pooled, out_layer, floor = prog
    T = feats.shape[0]
    fold = scheme.endswith("_fold")
    wround = scheme.startswith("mx")
    second = scheme.startswith("mx2")      # + q4(y - fp16(y)) . q4(W): the 1.5-pass arithmetic
    planes = {-1: (0, T - 1, feats.astype(np.float64), None, None)}     # idx -> (lo, hi, stored plane, (mant, offset) when folded, 4-bit residual)
    stats = None
    cache = forward.cache.setdefault((id(prog), scheme, tau), {})
    for i, L in enumerate(layers):
        if L["seg"]:
            x = np.concatenate([stats if s[0] == -2 else planes[s[0]][2] for s in L["src"]])
            z = L["w"] @ x + L["b"]
            y = (np.maximum(z, 0) if L["relu"] else z) * L["scale"] + L["offset"]
            planes[i] = (0, 0, y, None, None)
            continue
        lo = max(planes[s[0]][0] - s[1] for s in L["src"])
        hi = min(planes[s[0]][1] - s[1] for s in L["src"])
        exact_in = all(s[0] == -1 for s in L["src"])        # the layers on the network input run three passes
        z = np.tile(L["b"], (hi - lo + 1, 1))
        k0 = 0
        for j, (si, off, dim) in enumerate(L["src"]):
            plo, phi, P, fo, R4 = planes[si]
            x = P[lo + off - plo: hi + off - plo + 1]
            W = L["w"][:, k0:k0 + dim]
            if fo is not None:      # folded source: its BatchNorm moves into this consumer
                mant, o = fo
                z += W @ o
                W = W * mant[None, :]
            if exact_in or scheme == "exact":
                z += x @ W.T
            else:
                key = (i, j)
                if key not in cache:
                    wh = q16(W) if wround else W
                    wl = q_block_scaled(W - wh, GRIDS["4"], block=32, axis=1) if wround else None
                    w4 = q_block_scaled(W, GRIDS["4"], block=32, axis=1) if second else None
                    cache[key] = (wh, wl, w4)
                wh, wl, w4 = cache[key]
                z += x @ wh.T
                if wround:
                    z += q4_group16(x) @ wl.T
                if second and R4 is not None:
                    z += R4[lo + off - plo: hi + off - plo + 1] @ w4.T
            k0 += dim
        r = np.maximum(z, 0) if L["relu"] else z
        if i == pooled:
            y = r * L["scale"] + L["offset"]
            n = y.shape[0]
            mean = y.sum(0) / n
            var = np.maximum((y * y).sum(0) / n - mean * mean, floor)
            stats = np.concatenate([mean, np.sqrt(var)])
            planes[i] = (lo, hi, y, None, None)
            continue
        if scheme == "exact":
            planes[i] = (lo, hi, r * L["scale"] + L["offset"], None, None)
            continue
        sc, of = L["scale"].copy(), L["offset"].copy()
        fo = None
        if fold and L["relu"] and L["bn"]:
            # per column: fold where the offset is moderate (|o| <= tau); a column that sits far from zero relative to its spread
            # (|o| >> 1) would put values of size |o| / m into the plane where the centred variable is of size 1
            sel = np.abs(of) <= tau
            e = np.floor(np.log2(sc))
            mant = np.where(sel, sc / 2.0 ** e, 1.0)
            fo = (mant, np.where(sel, of, 0.0))
            sc = np.where(sel, 2.0 ** e, sc)
            of = np.where(sel, 0.0, of)
        y = r * sc + of
        yq = q16(y)
        planes[i] = (lo, hi, yq, fo, plane_residual4(y, yq) if second else None)
    return planes[out_layer][2]


forward.cache = {}


def main():
    nchunks = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    models = [("v2 init 123", lambda: H.synth_model("v2_xvector", 123)), ("v2 init 7", lambda: H.synth_model("v2_xvector", 7)),
              ("v2 trained-like 11", lambda: H.trained_like_model("v2_xvector", 11)),
              ("v2 trained-like 12", lambda: H.trained_like_model("v2_xvector", 12)),
              ("v5 init 123", lambda: H.synth_model("v5_cvector", 123)),
              ("v5 trained-like 11", lambda: H.trained_like_model("v5_cvector", 11))]
    d = tempfile.mkdtemp()
    only = os.environ.get("SIM_MODELS")
    if only:
        models = [m for m in models if any(o in m[0] for o in only.split(","))]
    for name, mk in models:
        net, line = mk()
        n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True))
        n2.apply_nnet_config(line)
        export_program(n2, d + "/p.bin")
        prog = load_program(d + "/p.bin")
        forward.cache.clear()     # (keyed by id(prog): a new model must not find the previous one's weight images)
        ev = H.xo.GraphEvaluator(n2, np.float64)
        taus = [float(a) for a in sys.argv[2:]] or [1e30]
        keys = [("exact", 1e30), ("mx", 1e30), ("mx2", 1e30)] + [(k, t) for t in taus for k in ("mx_fold", "mx2_fold")]
        errs = {k: [] for k in keys}
        zeros = []
        for c in range(nchunks):
            x = H.features(20000 + c, 400)
            ref = ev.compute(x)[0]
            for k in errs:
                forward.cache_tag = k
                got = forward(prog, x, k[0], k[1])
                errs[k].append(float(np.abs(got - ref).max() / np.abs(ref).max()))
        off_all = np.concatenate([np.abs(L["offset"]) for L in prog[0] if L["bn"] and not L["seg"]])
        print("%-20s |o| median %.2f, 90%% %.2f, max %.2f" % (name, np.median(off_all), np.quantile(off_all, 0.9), off_all.max()))
        for k, v in errs.items():
            print("      %-10s tau %-6s worst %.2e mean %.2e" % (k[0], "inf" if k[1] > 1e29 else "%g" % k[1], max(v), float(np.mean(v))), flush=True)


if __name__ == "__main__":
    main()
