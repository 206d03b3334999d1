import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_kernels as T
import torch
if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        d = a[k] != b[k]
        rows = np.unique(np.where(d)[0])
        print(k, "differing elements:", int(d.sum()), "of", d.size, " rows/groups:", rows[:20], "..." if len(rows) > 20 else "", " n rows", len(rows))
    sys.exit(0)
got = {}
orig = torch.allclose
def grab(a, b, rtol=0, atol=0):
    got["gmax"] = a.cpu().numpy().copy()
    return True
torch.allclose = grab
res = {}
for name, segs, epi in (("tdnn3", T.TDNN3, 0), ("tdnn4", [(0, 512, 0, 512)], 0), ("stats", [(0, 512, 0, 512)], 2)):
    out, ref = T._run_mx_case(epi, 66 * 512, 512 if epi == 0 else 1536, segs, seed=13)
    res[name] = out
    if epi == 0:
        res[name + "_gmax"] = got["gmax"]
np.savez(sys.argv[2], **res)
