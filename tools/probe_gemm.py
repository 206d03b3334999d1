#!/usr/bin/env python3
"""K sweep of the spliced-GEMM kernel through xv_kernel_tdnn_gemm (fixed cost per tile vs cost per K step).
Usage: python tools/probe_gemm.py [rows] [n_pad]   (GPU box only; timing with events on torch's current stream)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402

HALO = 32


def run(P, rows, n_pad, segs, epi, reps=20, separate_sources=False):
    dev = torch.device("cuda:0")
    ld = max(s[1] for s in segs)
    nsrc = len(segs) if separate_sources else 1
    X = torch.randn(nsrc, rows + 2 * HALO, ld, device=dev)
    xh = X.to(torch.bfloat16); xl = (X - xh.float()).to(torch.bfloat16)
    K = sum(s[3] for s in segs)
    W = torch.randn(n_pad, K, device=dev) / np.sqrt(K)
    wh = W.to(torch.bfloat16); wl = (W - wh.float()).to(torch.bfloat16)
    bias = torch.zeros(n_pad, device=dev); scale = torch.ones(n_pad, device=dev); offset = torch.zeros(n_pad, device=dev)
    oh = torch.empty(rows + 2 * HALO, n_pad, dtype=torch.bfloat16, device=dev); ol = torch.empty_like(oh)
    partial = torch.empty(rows // 16, 2, n_pad, device=dev)
    rng = torch.zeros(rows // 16, 2, dtype=torch.int8, device=dev); rng[:, 1] = 16
    d = P.GemmDesc()
    d.precision, d.epilogue, d.nseg = 0, epi, len(segs)
    for j, (si, l, shift, klen) in enumerate(segs):
        src = j if separate_sources else 0
        d.seg[j].hi = xh[src].data_ptr() + HALO * ld * 2
        d.seg[j].lo = xl[src].data_ptr() + HALO * ld * 2
        d.seg[j].ld, d.seg[j].row_shift, d.seg[j].k_len = ld, shift, klen
    d.w_hi, d.w_lo, d.ldw = wh.data_ptr(), wl.data_ptr(), K
    d.rows, d.n_pad = rows, n_pad
    d.bias, d.scale, d.offset = bias.data_ptr(), scale.data_ptr(), offset.data_ptr()
    d.relu, d.bn = 1, 1
    d.out_hi, d.out_lo, d.ldo = oh.data_ptr() + HALO * n_pad * 2, ol.data_ptr() + HALO * n_pad * 2, n_pad
    d.partial, d.ldp, d.grp_range = partial.data_ptr(), n_pad, rng.data_ptr()
    st = torch.cuda.current_stream()
    d.hip_stream = st.cuda_stream
    for _ in range(3):
        P.kernel_tdnn_gemm(d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        P.kernel_tdnn_gemm(d)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 102400
    n_pad = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    P = H.pkg()
    # warm the clocks
    run(P, rows, n_pad, [(0, 512, 0, 512)], 0, reps=50)
    print("rows %d n_pad %d (bf16x3)" % (rows, n_pad))
    for epi, name in ((0, "act"), (2, "stats")):
        for K in (32, 64, 128, 256, 512, 1024):
            t = run(P, rows, n_pad, [(0, max(K, 32), 0, K)], epi)
            print("  %-5s nshift=1 K=%4d steps=%2d : %.4f ms" % (name, K, K // 32, t))
        t = run(P, rows, n_pad, [(0, 256, 0, 256)], epi)
        print("  %-5s K=256 one strided source (64-B row segments)       : %.4f ms" % (name, t))
        t = run(P, rows, n_pad, [(j, 32, 0, 32) for j in range(8)], epi, separate_sources=True)
        print("  %-5s K=256 as 8 planes of 32 columns (contiguous 1 KiB)  : %.4f ms" % (name, t))
        for kseg in (32, 128, 512):
            t = run(P, rows, n_pad, [(0, kseg, -2, kseg), (0, kseg, 0, kseg), (0, kseg, 2, kseg)], epi)
            print("  %-5s nshift=3 K=%4d steps=%2d : %.4f ms" % (name, 3 * kseg, 3 * kseg // 32, t))


if __name__ == "__main__":
    main()
