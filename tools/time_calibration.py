import sys, os, time
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0,R); sys.path.insert(0,R+"/tests")
import numpy as np, helpers as H
P=H.pkg()
for topo in ("v2_xvector","v5_cvector"):
    net,line=H.synth_model(topo)
    model=P.Model(raw=net.to_bytes(True), nnet_config=line)
    utts=[H.features(i,400) for i in range(64)]
    f,o=H.pack(utts)
    for rep in range(3):
        ctx=P.Context(model)
        ctx.forward_batch(f,o)
        t0=time.perf_counter(); cal=ctx.calibrate(f,o); t1=time.perf_counter()
        t2=time.perf_counter(); ctx.forward_batch(f,o); t3=time.perf_counter()
        print(topo, rep, "calibrate %.1f ms" % ((t1-t0)*1e3), "one forward_batch %.1f ms" % ((t3-t2)*1e3), cal.get("lite_mask"), flush=True)
