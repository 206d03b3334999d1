#!/usr/bin/env python3
"""How far out does the error of the calibrated default go on MANY chunks?  (GPU box.)

The default arithmetic is chosen on 64 chunks (32 choose, 32 confirm) at a tolerance of 7.5e-5 against the three-pass
arithmetic; the full-batch parity tests assert the worst of 256 chunks against the fp64 oracle below 1e-4.  BASELINE config 4
is a million utterances: this tool measures the distribution of the per-chunk error over N distinct chunks (default 32768)
so that the tail is a measurement instead of an extrapolation.  Reference = the fp16x3 context of the same model on the same
chunks (its own error against the fp64 oracle is 5-7e-6 - tests/test_gpu_full_batch_parity.py - i.e. the figures below are good
to about a tenth of the tolerance; the oracle itself takes ~0.15 s per chunk and is used for a spot check of the worst chunks).

usage: tail_error.py [N] [--models v2,v2t11,v2t12,v5,v5t11] [--oracle K] [--frames LO-HI] [--tol T]
       (K worst chunks re-checked against the fp64 oracle; --frames: chunk lengths uniform over LO..HI instead of 400 - the fast
       arithmetics take chunks from 160 (fp16mx2) / 300 (fp16mx, mixtures) pooled frames, shorter ones run three-pass)
One JSON line per model: what the calibration chose, mean / percentiles / worst of the per-chunk relative error (max |d| over
max |ref| of the embedding, the measure of every parity test), how many chunks lie above 9e-5 and above 1e-4."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402

MODELS = {"v2": ("v2_xvector", None), "v2t11": ("v2_xvector", 11), "v2t12": ("v2_xvector", 12), "v5": ("v5_cvector", None),
          "v5t11": ("v5_cvector", 11)}


def main():
    args = sys.argv[1:]
    N = int(args[0]) if args and not args[0].startswith("--") else 32768
    names = args[args.index("--models") + 1].split(",") if "--models" in args else list(MODELS)
    n_oracle = int(args[args.index("--oracle") + 1]) if "--oracle" in args else 4
    tol = float(args[args.index("--tol") + 1]) if "--tol" in args else 7.5e-5
    P = H.pkg()
    B = 256
    lo, hi = (int(v) for v in args[args.index("--frames") + 1].split("-")) if "--frames" in args else (400, 400)
    lens = np.random.default_rng(99).integers(lo, hi + 1, N)
    for name in names:
        if name[:3] in ("v2s", "v5s"):   # another draw of the initialisation-like model: v2s7 = synth_model("v2_xvector", seed=7)
            topo, seed = ("v2_xvector" if name[1] == "2" else "v5_cvector"), None
            net, line = H.synth_model(topo, seed=int(name[3:]))
        elif name[:3] in ("v2t", "v5t") and name not in MODELS:   # another heavy-tailed one
            topo, seed = ("v2_xvector" if name[1] == "2" else "v5_cvector"), int(name[3:])
            net, line = H.trained_like_model(topo, seed)
        else:
            topo, seed = MODELS[name]
            net, line = H.synth_model(topo) if seed is None else H.trained_like_model(topo, seed)
        model = P.Model(raw=net.to_bytes(True), nnet_config=line)
        ctx = P.Context(model)                                   # XV_PREC_DEFAULT
        ref_ctx = P.Context(model, precision=P.PRECISIONS["fp16x3"])
        # the tools' sample: 64 chunks spread evenly over the whole list
        picks = sorted({((2 * i + 1) * N) // 128 for i in range(64)})
        sub = [H.features(700000 + k, int(lens[k])) for k in picks]
        f, o = H.pack(sub)
        cal = ctx.calibrate(f, o, tol)
        # --project: the two plain arithmetics next to what was adopted, each with two projections of its tail from the SAMPLE's
        # per-chunk errors - linear (mean + 6 sd, the rule of Engine::Calibrate) and log-normal (exp(mean + 6 sd of the logarithms)) -
        # to be held against the worst of the N chunks
        extra = {}
        if "--project" in args:
            ref_s = np.asarray(ref_ctx.forward_batch(f, o), np.float64)
            modes = {"adopted": ctx}
            for m in ("fp16mx", "fp16mx2"):
                modes[m] = P.Context(model, precision=P.PRECISIONS[m])
            for m, c2 in modes.items():
                g = np.asarray(c2.forward_batch(f, o), np.float64)
                e = np.abs(g - ref_s).max(axis=1) / np.abs(ref_s).max(axis=1)
                le = np.log(np.maximum(e, 1e-12))
                extra[m] = {"ctx": c2, "sample_worst": float(e.max()), "lin6": float(e.mean() + 6 * e.std(ddof=1)),
                            "log6": float(np.exp(le.mean() + 6 * le.std(ddof=1))), "errs": np.empty(N)}
        errs = np.empty(N)
        for b0 in range(0, N, B):
            utts = [H.features(700000 + k, int(lens[k])) for k in range(b0, min(N, b0 + B))]
            f, o = H.pack(utts)
            got = np.asarray(ctx.forward_batch(f, o), np.float64)
            ref = np.asarray(ref_ctx.forward_batch(f, o), np.float64)
            errs[b0:b0 + len(utts)] = np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)
            for m, x in extra.items():
                if m != "adopted":
                    g = np.asarray(x["ctx"].forward_batch(f, o), np.float64)
                    x["errs"][b0:b0 + len(utts)] = np.abs(g - ref).max(axis=1) / np.abs(ref).max(axis=1)
        worst = np.argsort(errs)[::-1][:n_oracle]
        spot = []
        if n_oracle:
            n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True))
            n2.apply_nnet_config(line)
            ev64 = H.xo.GraphEvaluator(n2, np.float64)
            for k in worst:
                u = H.features(700000 + int(k), int(lens[k]))
                f, o = H.pack([u])
                r64 = ev64.compute(u)[0]
                g = np.asarray(ctx.forward_batch(f, o), np.float64)[0]
                spot.append({"chunk": int(k), "frames": int(lens[k]), "vs_fp16x3": float(errs[k]), "vs_fp64_oracle": float(np.abs(g - r64).max() / np.abs(r64).max())})
        proj = {}
        for m, x in extra.items():
            e = errs if m == "adopted" else x["errs"]
            proj[m] = {"sample_worst": x["sample_worst"], "lin6": x["lin6"], "log6": x["log6"], "worst": float(e.max()), "mean": float(e.mean()),
                       "p99.9": float(np.quantile(e, 0.999))}
        q = lambda p: float(np.quantile(errs, p))   # noqa: E731
        print(json.dumps({"model": name, "chunks": N, "frames": [lo, hi], "calibration_tol": tol, "chosen": cal["chosen"], "lite_mask": cal.get("lite_mask", 0),
                          "lite_dropped": cal.get("lite_dropped", 0), "err_sample": {k: cal[k] for k in ("err_mx", "err_mx2", "err_lite", "err_holdout", "tail") if k in cal},
                          "mean": float(errs.mean()), "p50": q(0.5), "p99": q(0.99), "p99.9": q(0.999), "p99.99": q(0.9999),
                          "worst": float(errs.max()), "above_9e-5": int((errs > 9e-5).sum()), "above_1e-4": int((errs > 1e-4).sum()),
                          "worst_chunks_against_the_fp64_oracle": spot, **({"projections": proj} if proj else {})}), flush=True)


if __name__ == "__main__":
    main()
