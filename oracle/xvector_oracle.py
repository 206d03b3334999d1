"""TEST INFRASTRUCTURE - not a product path.  PARITY UNPINNED (see oracle/README.md).

CPU restatement (numpy) of what `nnet3-xvector-compute` computes for the graphs the reference defines.

The arithmetic of this path lives in upstream Kaldi (github.com/kaldi-asr/kaldi, ~v5.3/5.4, Q1 2018,
NOT vendored under /root/reference and unpinned - SURVEY.md §8(c)); the reference only holds the call
sites (egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:86-93) and the graph definitions
(egs/sre/v2/local/nnet3/xvector/run_xvector_new.sh:94-114 etc.).  This file restates the published
nnet3 semantics as a *generic graph evaluator* - it does not pattern-match
TDNN layers, so it is an independent formulation from the product's lowering in csrc/program.cc:

  * request = input frames t in [0,T), output index t=0 (x-vector) or every computable t (frame-level);
  * a node index is computable iff all Append() parts are; no zero padding, ever (App. B.4);
  * Affine: y = x W^T + b;  ReLU;  BatchNorm test mode: y = x*s + o, s = target_rms/sqrt(var+eps), o = -mean*s;
  * StatisticsExtraction -> [1, x, x^2]; StatisticsPooling over the available frames of [t-L, t+R]:
    mean = S/n, sigma = sqrt(max(Q/n - mean^2, floor));
  * chunk loop + length-weighted average exactly as App. B.5 (both --pad-input behaviours).

dtype=np.float32 mimics Kaldi's BaseFloat arithmetic (sgemm via numpy/OpenBLAS - also the cpu_baseline
"port"); dtype=np.float64 is the accuracy ground truth.
"""
import math

import numpy as np

from .nnet3_model import Nnet3, parse_config_line, parse_descriptor

_SIMPLE = ("NaturalGradientAffineComponent", "AffineComponent", "FixedAffineComponent", "LinearComponent",
           "RectifiedLinearComponent", "BatchNormComponent", "LogSoftmaxComponent", "SoftmaxComponent",
           "SigmoidComponent", "TanhComponent", "NoOpComponent")


class GraphEvaluator:
    def __init__(self, net: Nnet3, dtype=np.float64, output_name="output"):
        self.net = net
        self.dtype = dtype
        self.output_name = output_name
        self.nodes = {}
        for line in net.config_lines:
            p = parse_config_line(line)
            if p is None:
                continue
            kind, kv = p
            entry = {"kind": kind}
            if kind == "input-node":
                entry["dim"] = int(kv["dim"])
            elif kind == "component-node":
                entry["component"] = kv["component"]
                entry["desc"] = parse_descriptor(kv["input"])
            elif kind == "output-node":
                entry["desc"] = parse_descriptor(kv["input"])
            else:
                raise ValueError("unsupported node type " + kind)
            # the same name may exist as output-node and component-node in principle; outputs win for lookup by
            # output_name only
            key = kv["name"]
            if kind == "output-node":
                self.nodes.setdefault("__out__" + key, entry)
                if key not in self.nodes:
                    self.nodes[key] = entry
            else:
                self.nodes[key] = entry
        self._w = {}
        for name, c in net.components.items():
            self._w[name] = self._prep_component(c)
        self.input_dim = next(e["dim"] for e in self.nodes.values() if e["kind"] == "input-node")

    # ------------------------------------------------------------------ components
    def _prep_component(self, c):
        dt = self.dtype
        t = c.type
        if "linear" in c.f:
            w = np.asarray(c.f["linear"], dtype=dt)
            b = np.asarray(c.f["bias"], dtype=dt) if "bias" in c.f else np.zeros(w.shape[0], dt)
            return ("affine", np.ascontiguousarray(w.T), b)
        if t == "RectifiedLinearComponent":
            return ("relu",)
        if t == "BatchNormComponent":
            eps = dt(c.f.get("epsilon", 1e-3))
            rms = dt(c.f.get("target_rms", 1.0))
            var = np.asarray(c.f["stats_var"], dtype=dt)
            mean = np.asarray(c.f["stats_mean"], dtype=dt)
            scale = rms * (var + eps) ** dt(-0.5)
            offset = -mean * scale
            return ("scaleoffset", scale.astype(dt), offset.astype(dt))
        if t == "LogSoftmaxComponent":
            return ("logsoftmax",)
        if t == "StatisticsExtractionComponent":
            assert c.f.get("input_period", 1) == 1 and c.f.get("output_period", 1) == 1
            return ("stats_extract", bool(c.f.get("include_variance", True)))
        if t == "StatisticsPoolingComponent":
            assert c.f.get("input_period", 1) == 1 and c.f.get("num_log_count_features", 0) == 0
            return ("stats_pool", int(c.f.get("left_context", 0)), int(c.f.get("right_context", 0)),
                    bool(c.f.get("output_stddevs", True)), float(c.f.get("variance_floor", 1e-10)))
        if t == "NoOpComponent":
            return ("noop",)
        raise ValueError("oracle: unsupported component " + t)

    def _apply(self, w, x):
        k = w[0]
        if k == "affine":
            return x @ w[1] + w[2]
        if k == "relu":
            return np.maximum(x, 0)
        if k == "scaleoffset":
            return x * w[1] + w[2]
        if k == "logsoftmax":
            m = x.max(axis=1, keepdims=True)
            return x - m - np.log(np.exp(x - m).sum(axis=1, keepdims=True))
        if k == "stats_extract":
            ones = np.ones((x.shape[0], 1), dtype=x.dtype)
            return np.concatenate([ones, x, x * x], axis=1) if w[1] else np.concatenate([ones, x], axis=1)
        if k == "noop":
            return x
        raise ValueError(k)

    # ------------------------------------------------------------------ ranges (computable frames)
    def _desc_range(self, d):
        if d.kind == "node":
            return self._node_range(d.name)
        if d.kind == "offset":
            lo, hi = self._desc_range(d.args[0])
            return lo - d.value, hi - d.value
        if d.kind in ("round", "scale"):
            return self._desc_range(d.args[0])
        los, his = zip(*[self._desc_range(a) for a in d.args])
        return max(los), min(his)

    def _node_range(self, name):
        if name in self._range:
            return self._range[name]
        e = self.nodes[name]
        if e["kind"] == "input-node":
            r = (0, self._T - 1)
        else:
            lo, hi = self._desc_range(e["desc"])
            if e["kind"] == "component-node":
                w = self._w[e["component"]]
                if w[0] == "stats_pool":
                    L, R = w[1], w[2]
                    # computable at t iff [t-L, t+R] meets [lo, hi]
                    lo, hi = (lo - R, hi + L) if hi >= lo else (0, -1)
            r = (lo, hi)
        self._range[name] = r
        return r

    def _segment_level(self, name):
        e = self.nodes[name]
        if e["kind"] == "input-node":
            return False
        if e["kind"] == "component-node" and self._w[e["component"]][0] == "stats_pool":
            return True
        return any(self._segment_level(n) for n in e["desc"].nodes())

    # ------------------------------------------------------------------ values
    def _desc_value(self, d, t0, t1):
        if d.kind == "node":
            return self._node_value(d.name, t0, t1)
        if d.kind == "offset":
            return self._desc_value(d.args[0], t0 + d.value, t1 + d.value)
        if d.kind == "round":
            return self._desc_value(d.args[0], t0, t1)
        if d.kind == "scale":
            return self._desc_value(d.args[0], t0, t1) * self.dtype(d.value)
        parts = [self._desc_value(a, t0, t1) for a in d.args]
        if d.kind == "append":
            return np.concatenate(parts, axis=1)
        return sum(parts[1:], parts[0])

    def _node_value(self, name, t0, t1):
        e = self.nodes[name]
        if e["kind"] == "input-node":
            assert 0 <= t0 and t1 < self._T, "input frame out of range (no padding in nnet3)"
            return self._feats[t0:t1 + 1]
        if not self._segment_level(name):
            if name not in self._full:
                lo, hi = self._node_range(name)
                if hi < lo:
                    raise ValueError("node %s is not computable for T=%d" % (name, self._T))
                x = self._desc_value(e["desc"], lo, hi)
                if e["kind"] == "component-node":
                    x = self._apply(self._w[e["component"]], x)
                self._full[name] = (lo, hi, x)
            lo, hi, x = self._full[name]
            assert lo <= t0 and t1 <= hi, "frame %d..%d of %s not computable" % (t0, t1, name)
            return x[t0 - lo:t1 - lo + 1]
        # segment-level
        if e["kind"] == "component-node" and self._w[e["component"]][0] == "stats_pool":
            _, L, R, stddevs, floor = self._w[e["component"]]
            src_lo, src_hi = self._desc_range(e["desc"])
            rows = []
            for t in range(t0, t1 + 1):
                a, b = max(t - L, src_lo), min(t + R, src_hi)
                if b < a:
                    raise ValueError("stats pooling at t=%d has no input" % t)
                st = self._desc_value(e["desc"], a, b)
                s = st.sum(axis=0)
                self.pooled_frames = b - a + 1
                n = s[0]
                dim = (st.shape[1] - 1) // 2
                mean = s[1:1 + dim] / n
                if stddevs:
                    var = s[1 + dim:] / n - mean * mean
                    var = np.maximum(var, self.dtype(floor))
                    rows.append(np.concatenate([mean, var ** self.dtype(0.5)]))
                else:
                    rows.append(np.concatenate([mean, s[1 + dim:] / n]))
            return np.stack(rows).astype(self.dtype)
        x = self._desc_value(e["desc"], t0, t1)
        if e["kind"] == "component-node":
            x = self._apply(self._w[e["component"]], x)
        return x

    # ------------------------------------------------------------------ API
    def context(self):
        """(left, right) context of the frame-level part feeding the output (frames lost on each side)."""
        self._T, self._range, self._full = 1000, {}, {}
        out = self.nodes["__out__" + self.output_name]
        names = set()

        def walk(d):
            for n in d.nodes():
                if n in names:
                    continue
                names.add(n)
                e = self.nodes[n]
                if e["kind"] != "input-node":
                    walk(e["desc"])
        walk(out["desc"])
        frame_nodes = [n for n in names if not self._segment_level(n)]
        lo = max(self._node_range(n)[0] for n in frame_nodes)
        hi = min(self._node_range(n)[1] for n in frame_nodes)
        return lo, self._T - 1 - hi

    def compute(self, feats):
        """One chunk: returns [1, dim] for a pooled (segment-level) output, else [frames, dim]."""
        feats = np.asarray(feats, dtype=self.dtype)
        self._feats, self._T = feats, feats.shape[0]
        self._range, self._full = {}, {}
        out = self.nodes["__out__" + self.output_name]
        seg = any(self._segment_level(n) for n in out["desc"].nodes())
        if seg:
            return self._desc_value(out["desc"], 0, 0)
        lo, hi = self._desc_range(out["desc"])
        if hi < lo:
            raise ValueError("output not computable for T=%d" % self._T)
        return self._desc_value(out["desc"], lo, hi)


def extract_xvector(ev: GraphEvaluator, feats, chunk_size=-1, min_chunk_size=100, pad_input=True):
    """The per-utterance loop of nnet3-xvector-compute (parameters come from
    run_xvector_new.sh:83,88 -> 10000 / 25 and extract_xvectors_new.sh:62-68).
    Returns the embedding, or None when the utterance is counted as failed."""
    feats = np.asarray(feats)
    num_rows = feats.shape[0]
    if num_rows == 0:
        return None
    this_chunk = chunk_size
    if not pad_input and num_rows < min_chunk_size:
        return None
    elif num_rows < chunk_size:
        this_chunk = num_rows
    elif chunk_size == -1:
        this_chunk = num_rows
    num_chunks = int(math.ceil(num_rows / float(this_chunk)))
    avg = None
    tot = 0.0
    for ci in range(num_chunks):
        offset = min(this_chunk, num_rows - ci * this_chunk)
        if not pad_input and offset < min_chunk_size:
            continue
        sub = feats[ci * this_chunk: ci * this_chunk + offset]
        tot += offset
        if pad_input and offset < min_chunk_size:
            left = (min_chunk_size - offset) // 2
            right = min_chunk_size - offset - left
            sub = np.concatenate([np.repeat(sub[:1], left, axis=0), sub, np.repeat(sub[-1:], right, axis=0)], axis=0)
        xv = ev.compute(sub)[0]
        avg = offset * xv if avg is None else avg + offset * xv
    if avg is None:
        return None
    return (avg / ev.dtype(tot)).astype(ev.dtype)


def compute_all_frames(ev: GraphEvaluator, feats):
    """Frame-level output for EVERY input frame, the way `nnet3-compute` produces it (reference call sites:
    sid/nnet3_cvector/cvector/extract_log_post.sh:77-84, sid/nnet3_cvector/am/extract_bn.sh:68): frames the network
    needs outside [0, T) are the first / last frame repeated [UPSTREAM Kaldi DecodableNnetSimple, recalled]."""
    feats = np.asarray(feats)
    left, right = ev.context()
    padded = np.concatenate([np.repeat(feats[:1], left, axis=0), feats, np.repeat(feats[-1:], right, axis=0)], axis=0)
    out = ev.compute(padded)
    assert out.shape[0] == feats.shape[0]
    return out


# ------------------------------------------------------------------ synthetic data
def synthetic_features(i, T, dim=23):
    """Utterance i: N(0,1)*sigma_d, sigma_d = 8*0.9^d, seeded 20180101+i."""
    rng = np.random.default_rng(20180101 + i)
    sigma = 8.0 * 0.9 ** np.arange(dim)
    return (rng.standard_normal((T, dim)) * sigma).astype(np.float32)


def xvector_macs(T):
    """Algorithmic MACs of the v2 x-vector path (SURVEY.md App. A.3 / BASELINE.md §2)."""
    return 58880 * (T - 4) + 786432 * (T - 8) + 1816576 * (T - 14) + 1536000


def cvector_macs(T):
    am = 74750 * (T - 4) + 1267500 * ((T - 6) + (T - 8) + (T - 14)) + 249600 * (T - 20)
    x = 58880 * (T - 10) + 786432 * (T - 14) + 2008576 * (T - 20) + 1536000
    return am + x
