"""The RCCL leg of the multi-GPU path (it replaces the per-job model reads of
egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:59,72) executed on the ONE GPU a test box has: a one-rank
communicator is enough to run ncclCommInitAll / ncclBroadcast and the device-resident blob path behind them."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import helpers as H
from oracle import kaldi_io as kio

pytestmark = pytest.mark.gpu


def test_ctx_create_broadcast_one_rank_is_bit_identical():
    """xv_ctx_create_broadcast(model, {0}, 1): packs on the host, uploads, runs a real ncclBroadcast on a one-rank
    communicator and builds the context from the device image it left behind (no host round trip).  The forward pass
    must equal xv_ctx_create's bit for bit."""
    P = H.pkg()
    net, line = H.synth_model("v2_xvector")
    model = P.Model(raw=net.to_bytes(True), nnet_config=line)
    utts = [H.features(i, T) for i, T in enumerate((400, 57, 333))]
    feats, offs = H.pack(utts)
    want = P.Context(model, precision=P.PREC_AUTO).forward_batch(feats, offs)
    ctxs = P.create_broadcast(model, [0], precision=P.PREC_AUTO)
    assert len(ctxs) == 1 and ctxs[0].device == 0 and ctxs[0].precision == P.PREC_AUTO
    assert np.array_equal(ctxs[0].forward_batch(feats, offs), want)


def test_context_from_device_blob():
    """xv_ctx_create_from_device_blob: the packed image handed over as device memory (what a rank holds after the
    broadcast)."""
    import torch
    P = H.pkg()
    net, line = H.synth_model("v2_xvector")
    model = P.Model(raw=net.to_bytes(True), nnet_config=line)
    blob = model.pack(P.PREC_AUTO)
    wt = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
    ctx = P.Context(device_blob=(wt.data_ptr(), wt.numel()), device=0)
    x = H.features(3, 400)
    assert np.array_equal(ctx.forward_batch(x, [0, 400]), P.Context(blob=blob).forward_batch(x, [0, 400]))
    # a damaged image is refused, not dereferenced
    bad = wt.clone()
    bad[200:264] = 255
    with pytest.raises(P.XvError):
        P.Context(device_blob=(bad.data_ptr(), bad.numel()), device=0)


def test_dist_extract_nccl_one_rank(tmp_path):
    """dist_extract.py --backend nccl under torch.distributed.run with one rank: process group on RCCL, the weight
    broadcast, the context from the broadcast buffer, the table job, the merged scp - against the oracle."""
    net, line = H.synth_model("v2_xvector")
    (tmp_path / "final.raw").write_bytes(net.to_bytes(True))
    lens = [400, 137, 25, 333, 1000, 64]
    utts = [("utt%d" % i, H.features(700 + i, T)) for i, T in enumerate(lens)]
    kio.write_ark_matrices(str(tmp_path / "feats.ark"), utts, scp_path=str(tmp_path / "feats.scp"))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tmp_path / "out"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(H.ROOT, H.PKG_NAME, "dist_extract.py"), "--nnet", str(tmp_path / "final.raw"),
           "--output-node", "tdnn6.affine", "--feats-scp", str(tmp_path / "feats.scp"), "--out-dir", str(out), "--name", "t",
           "--backend", "nccl", "--min-chunk-size", "25", "--chunk-size", "10000", "--precision", "fp16x3"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-2000:]
    assert "Done 6 utterances, failed for 0 (over 1 ranks)" in r.stdout
    got = dict(kio.read_scp(str(out / "xvector_t.scp"), "vector"))
    n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True))
    n2.apply_nnet_config(line)
    ev = H.xo.GraphEvaluator(n2, np.float32)
    for k, x in utts:
        ref = H.xo.extract_xvector(ev, x, 10000, 25, True)
        assert H.rel_err(got[k][None], ref[None]) < 1e-4, k
