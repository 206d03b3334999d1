"""GPU box: run-to-run reproducibility of one GEMM launch through the C ABI (tests/test_gpu_kernels.py::_run_mx_case): the same
launch N times, outputs compared bit for bit.  usage: repeat_mx_case.py [repeats] [rows] [precisions, e.g. 7,6]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_kernels as K  # noqa: E402

import threading  # noqa: E402

import torch  # noqa: E402

# a second stream that keeps the memory system and the CUs busy meanwhile (REPEAT_NOISE=0 turns it off): a race that needs a
# late LDS-DMA or a co-resident workgroup does not show in a launch that has the chip to itself
_stop = False


def _noise():
    st = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
    b = torch.randn(1 << 26, device="cuda")
    with torch.cuda.stream(st):
        while not _stop:
            for _ in range(4):
                (a @ a)
                b.add_(1.0)
            st.synchronize()


if os.environ.get("REPEAT_NOISE", "1") != "0":
    threading.Thread(target=_noise, daemon=True).start()

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ROWS = int(sys.argv[2]) if len(sys.argv) > 2 else 768
cases = {"tdnn3": K.TDNN3, "cvec5": K.CVEC5_MX, "tdnn4": [(0, 512, 0, 512)], "am768": K.AM_MX, "odd": K.CVEC5_ODD}
PRECS = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [7, 6, 9]
for prec in PRECS:
    for epi in ((0, 2) if prec != 9 else (0,)):
        for name, segs in cases.items():
            for n_pad in (512, 128, 768):
                ref = None
                bad = 0
                try:
                    for i in range(N):
                        out = K._run_mx_case(epi, ROWS, n_pad, segs, seed=19, prec=prec, p8=int(os.environ.get("REPEAT_P8", "0")))[0]
                        if ref is None:
                            ref = out
                        elif not np.array_equal(ref, out):
                            bad += 1
                except Exception as e:  # noqa: BLE001
                    print("prec %d epi %d %-6s n_pad %4d: %s" % (prec, epi, name, n_pad, str(e)[:80]))
                    continue
                print("prec %d epi %d %-6s n_pad %4d: %d of %d runs differ" % (prec, epi, name, n_pad, bad, N - 1), flush=True)
_stop = True
os._exit(0)
