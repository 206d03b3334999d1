"""GPU box: cost of the per-launch HIP events (set_profiling) and of a second engine lane on the headline workload.
usage: prof_cost.py <lanes>  (measured r01: profiling <= 2 %, second lane +1.5 % once stream-K removed the tile tails)"""
import os, sys, time, importlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
P = importlib.import_module("speaker-embedding-with-phonetic-information_amd")
os.environ["XVEC_LANES"] = sys.argv[1]
net, line = H.synth_model("v2_xvector")
model = P.Model(raw=net.to_bytes(True), nnet_config=line)
ctx = P.Context(model, device=0)
B, T, D = 256, 400, 23
dev = torch.device("cuda:0")
feats = torch.randn(B * T, D, device=dev) * (8.0 * 0.9 ** torch.arange(D, device=dev))
offs = (np.arange(B + 1) * T).astype(np.int32)
outs = [torch.empty(B, 512, device=dev) for _ in range(4)]
def run(n):
    for i in range(n):
        o = outs[i % 4]; ctx.forward_batch_device(feats.data_ptr(), offs, o.data_ptr(), 512, None)
    torch.cuda.synchronize()
run(200)
for prof in (False, True, False, True):
    ctx.set_profiling(prof)
    t = time.perf_counter(); run(100); dt = time.perf_counter() - t
    if prof: ctx.profile_report()
    print("lanes", sys.argv[1], "profiling", prof, "utt/s %.0f" % (B * 100 / dt))
