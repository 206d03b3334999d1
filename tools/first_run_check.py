import sys, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0,R); sys.path.insert(0,R+"/tests")
import numpy as np, helpers as H
P=H.pkg()
cfgs,_=H.TOPOLOGIES["v3_multitask"]
net=H.nm.synthesize([H.config_text(c) for c in cfgs], seed=123, head_stddev=1.0)
for node, prec in (("tdnn5_am.batchnorm","fp16"),("tdnn5_am.batchnorm","bf16"),("output_am.log-softmax","fp16"),("tdnn4_xvec.batchnorm","fp16")):
    model=P.Model(raw=net.to_bytes(True), nnet_config="output-node name=output input=%s" % node)
    lens=np.random.default_rng(5).integers(200,601,24)
    f,o=H.pack([H.features(700+i,int(T)) for i,T in enumerate(lens)])
    nbad=0
    for rep in range(6):
        ctx=P.Context(model, device=0, precision=P.PRECISIONS[prec])
        a=ctx.forward_batch(f,o); b=ctx.forward_batch(f,o); c=ctx.forward_batch(f,o)
        if not np.array_equal(a,b) or not np.array_equal(b,c):
            nbad+=1
            d=np.abs(a-b); rows=np.where(d.max(1)>0)[0]
            print(node,prec,"rep",rep,"first vs second differ: rows",rows[:10],len(rows),"of",len(a),"max",d.max(), "b==c", np.array_equal(b,c), "nan in a", np.isnan(a).sum(), flush=True)
            offs=np.asarray(o); 
            print("   chunk starts near:", [int(offs[np.searchsorted(offs, r, side='right')-1]) for r in rows[:5]], "row-in-chunk", [int(r-offs[np.searchsorted(offs, r, side='right')-1]) for r in rows[:5]])
        del ctx
    print(node,prec,"fresh contexts with a differing first run:",nbad,"of 6",flush=True)
