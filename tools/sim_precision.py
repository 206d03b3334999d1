"""CPU study behind DESIGN.md section 3.0 (no GPU, no product code): embedding error of the x-vector / c-vector nets when the
operands of the frame-level affine layers are rounded to 16 bits, evaluated with the fp64 graph evaluator of the test
oracle.  Shows that the weight rounding error survives the statistics pooling (it is the same in every frame) while the
activation rounding error averages out like 1/sqrt(pooled frames) - the reason the fp16x2 kernel mode (fp16 activations
x split-fp16 weights, two MFMAs per product) meets the 1e-4 bar for long chunks.
usage: sim_precision.py [topology] [T ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402


def q16(x):
    return x.astype(np.float16).astype(np.float64)


def qb16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)
    return r.astype(np.float64)


class QEval(H.xo.GraphEvaluator):
    """fp64 evaluation with the activations (xq) and / or weights (wq) of the frame-level affines rounded; `layers`
    restricts the activation rounding to the listed affine indices (per-layer sensitivity)."""

    def __init__(self, net, xq=None, wq=None, layers=None):
        super().__init__(net, np.float64)
        self.xq, self.wq, self.layers, self.idx = xq, wq, layers, 0

    def compute(self, f):
        self.idx = 0
        return super().compute(f)

    def _apply(self, w, x):
        if w[0] == "affine":
            i = self.idx
            self.idx += 1
            if x.shape[0] > 1:      # frame-level; the layers after the pooling run three-pass arithmetic
                if self.xq and (self.layers is None or i in self.layers):
                    x = self.xq(x)
                return x @ (self.wq(w[1]) if self.wq else w[1]) + w[2]
        return super()._apply(w, x)


def main():
    topo = sys.argv[1] if len(sys.argv) > 1 else "v2_xvector"
    lens = [int(a) for a in sys.argv[2:]] or [400, 100, 25, 1000]
    net, line = H.synth_model(topo)
    n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True))
    n2.apply_nnet_config(line)
    ev64 = H.xo.GraphEvaluator(n2, np.float64)
    modes = {"fp16 x, exact w": (q16, None), "exact x, fp16 w": (None, q16), "fp16 x, fp16 w": (q16, q16),
             "bf16 x, exact w": (qb16, None), "bf16 x, bf16 w": (qb16, qb16)}
    for T in lens:
        errs = {k: [] for k in modes}
        for i in range(3):
            x = H.features(i, T)
            ref = ev64.compute(x)
            for k, (xq, wq) in modes.items():
                errs[k].append(H.rel_err(QEval(n2, xq, wq).compute(x), ref))
        print("T=%d" % T)
        for k in modes:
            print("   %-18s max %.3e  mean %.3e" % (k, max(errs[k]), float(np.mean(errs[k]))))
        if T == lens[0]:
            for layers in ([0], [1], [2], [3], [4]):
                e = [H.rel_err(QEval(n2, q16, None, layers).compute(H.features(i, T)), ev64.compute(H.features(i, T)))
                     for i in range(2)]
                print("   fp16 x in affine %d only: %.3e" % (layers[0], max(e)))


if __name__ == "__main__":
    main()
