"""TEST INFRASTRUCTURE - not a product path.  PARITY UNPINNED (see oracle/README.md).

Flattens an nnet3 model (oracle/nnet3_model.Nnet3) into the layer list that oracle/xvec_oracle.c reads.
Only the grammar of the reference's graphs is handled: spliced affine [-> ReLU] [-> BatchNorm], one
mean+stddev pooling, affine layers after it."""
import struct

import numpy as np

from .nnet3_model import parse_config_line, parse_descriptor

_AFFINE = ("NaturalGradientAffineComponent", "AffineComponent", "FixedAffineComponent")


def _terms(d, shift=0):
    if d.kind == "node":
        return [(d.name, shift)]
    if d.kind == "offset":
        return _terms(d.args[0], shift + d.value)
    if d.kind == "round":
        return _terms(d.args[0], shift)
    if d.kind == "append":
        out = []
        for a in d.args:
            out += _terms(a, shift)
        return out
    raise ValueError("unsupported descriptor " + repr(d))


def export_program(net, path, output_name="output"):
    nodes = {}
    out_desc = None
    for line in net.config_lines:
        p = parse_config_line(line)
        if p is None:
            continue
        kind, kv = p
        if kind == "output-node" and kv["name"] == output_name:
            out_desc = parse_descriptor(kv["input"])
        elif kind != "output-node":
            nodes[kv["name"]] = (kind, kv)
    layers, memo, state = [], {}, {"pooled": -1, "floor": 1e-10}

    def materialise(name):
        if name in memo:
            return memo[name]
        kind, kv = nodes[name]
        if kind == "input-node":
            memo[name] = -1
            return -1
        ops, cur = [], name
        while True:
            kind, kv = nodes[cur]
            comp = net.components[kv["component"]]
            if comp.type in _AFFINE:
                break
            (src, sh), = _terms(parse_descriptor(kv["input"]))
            if comp.type == "StatisticsPoolingComponent":
                ek, ekv = nodes[src]
                (fsrc, _), = _terms(parse_descriptor(ekv["input"]))
                state["pooled"] = materialise(fsrc)
                state["floor"] = comp.f.get("variance_floor", 1e-10)
                memo[name] = -2
                return -2
            ops.append(comp)
            cur = src
        ops.reverse()
        L = {"relu": 0, "bn": 0, "w": np.asarray(comp.f["linear"], np.float32), "b": np.asarray(comp.f["bias"], np.float32)}
        n = L["w"].shape[0]
        L["scale"], L["offset"] = np.ones(n, np.float32), np.zeros(n, np.float32)
        for c in ops:
            if c.type == "RectifiedLinearComponent" and not L["bn"]:
                L["relu"] = 1
            elif c.type == "BatchNormComponent" and not L["bn"]:
                s = (np.float32(c.f.get("target_rms", 1.0)) *
                     (np.asarray(c.f["stats_var"], np.float32) + np.float32(c.f.get("epsilon", 1e-3))) ** np.float32(-0.5))
                L["bn"], L["scale"], L["offset"] = 1, s.astype(np.float32), (-np.asarray(c.f["stats_mean"], np.float32) * s).astype(np.float32)
            else:
                raise ValueError("unsupported chain at " + name)
        L["src"] = []
        seg = False
        for (sname, sh) in _terms(parse_descriptor(kv["input"])):
            idx = materialise(sname)
            dim = int(nodes[sname][1]["dim"]) if idx == -1 else (2 * layers[state["pooled"]]["w"].shape[0] if idx == -2 else layers[idx]["w"].shape[0])
            seg = seg or idx == -2 or (idx >= 0 and layers[idx]["segment"])
            L["src"].append((idx, 0 if seg else sh, dim))
        L["segment"] = 1 if seg else 0
        layers.append(L)
        memo[name] = len(layers) - 1
        return memo[name]

    (oname, _), = _terms(out_desc)
    out_layer = materialise(oname)
    input_dim = next(int(kv["dim"]) for k, kv in nodes.values() if k == "input-node")
    with open(path, "wb") as f:
        f.write(b"XVORACLE")
        f.write(struct.pack("<iiiif", input_dim, len(layers), state["pooled"], out_layer, state["floor"]))
        for L in layers:
            f.write(struct.pack("<i", len(L["src"])))
            for s in L["src"]:
                f.write(struct.pack("<iii", *s))
            n, k = L["w"].shape
            f.write(struct.pack("<iiiii", k, n, L["relu"], L["bn"], L["segment"]))
            f.write(np.ascontiguousarray(L["w"], np.float32).tobytes())
            f.write(L["b"].tobytes())
            f.write(L["scale"].tobytes())
            f.write(L["offset"].tobytes())
    return len(layers)
