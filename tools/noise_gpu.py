"""GPU box: keep the chip busy for N seconds (fp16 matmuls + a streaming add on a stream of its own) - background load for
running the GPU test suite or the stress tools under contention.  usage: noise_gpu.py [seconds]"""
import sys
import time

import torch

T = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
a = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
b = torch.randn(1 << 26, device="cuda")
t0 = time.time()
while time.time() - t0 < T:
    for _ in range(8):
        (a @ a)
        b.add_(1.0)
    torch.cuda.synchronize()
