"""GPU box: background load that raises MEMORY LATENCY for everything else on the chip - random 4-byte gathers over a buffer
far larger than the TLB reach and the L2 / MALL (default 48 GiB), beside a streaming add and fp16 matmuls, each on a stream of
its own.  tools/noise_gpu.py keeps the CUs and the HBM bandwidth busy; a hazard that needs a LATE LDS-DMA (a miscounted vmcnt
wait) shows when the loads of the kernel under test take several times their usual time, which is what page-table walks and
row conflicts of another process's gathers do to them.  usage: noise_latency.py [seconds] [GiB]"""
import sys
import time

import torch

T = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
GIB = float(sys.argv[2]) if len(sys.argv) > 2 else 48.0
n = int(GIB * (1 << 30) / 4)
big = torch.empty(n, device="cuda", dtype=torch.int32)
big.zero_()
idx = torch.randint(0, n, (1 << 25,), device="cuda", dtype=torch.int64)
a = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
b = torch.randn(1 << 26, device="cuda")
s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
t0 = time.time()
while time.time() - t0 < T:
    with torch.cuda.stream(s1):
        for _ in range(8):
            big[idx].sum()
            idx = (idx * 1103515245 + 12345) % n
    with torch.cuda.stream(s2):
        for _ in range(4):
            b.add_(1.0)
    with torch.cuda.stream(s3):
        for _ in range(2):
            (a @ a)
    torch.cuda.synchronize()
