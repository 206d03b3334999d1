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
    # an image of another format version (an older build's pack) is refused by name, from the host and from the device
    old = bytearray(blob)
    old[8:12] = (5).to_bytes(4, "little")     # BlobHeader: char magic[8]; uint32 version
    with pytest.raises(P.XvError, match="format version 5"):
        P.Context(blob=bytes(old))
    oldt = torch.frombuffer(old, dtype=torch.uint8).cuda()
    with pytest.raises(P.XvError, match="format version 5"):
        P.Context(device_blob=(oldt.data_ptr(), oldt.numel()), device=0)


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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_dist_extract(tmp_path, nproc, out_name, extra):
    out = tmp_path / out_name
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(H.ROOT, H.PKG_NAME, "dist_extract.py"),
           "--nnet", str(tmp_path / "final.raw"), "--output-node", "tdnn6.affine", "--feats-scp", str(tmp_path / "feats.scp"),
           "--out-dir", str(out), "--name", "t", "--min-chunk-size", "25", "--chunk-size", "10000"] + extra
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-3000:]
    return out, r.stdout


def test_two_ranks_on_one_gpu_equal_one_rank_byte_for_byte(tmp_path):
    """Shard equivalence with real compute at N > 1 ("outputs of N-way sharded run == 1-way run,
    bit-for-bit per utterance"; sharding rule utils/split_scp.pl:208-217, concatenation
    egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:99): dist_extract.py with two ranks that share device 0
    (gloo carries the weight broadcast; the launcher starts before any GPU call) against a one-rank run of the same
    job.  The merged scp must list the same keys in the same order and every vector must be byte-identical - no
    utterance's embedding may depend on which rank, batch or neighbours it was computed with."""
    net, _ = H.synth_model("v2_xvector")
    (tmp_path / "final.raw").write_bytes(net.to_bytes(True))
    lens = [400, 137, 25, 333, 1000, 64, 400, 400, 211, 15, 399, 640, 87]
    utts = [("utt%02d" % i, H.features(900 + i, T)) for i, T in enumerate(lens)]
    kio.write_ark_matrices(str(tmp_path / "feats.ark"), utts, scp_path=str(tmp_path / "feats.scp"))
    one, log1 = _run_dist_extract(tmp_path, 1, "one", ["--backend", "gloo", "--force-device", "0"])
    two, log2 = _run_dist_extract(tmp_path, 2, "two", ["--backend", "gloo", "--force-device", "0"])
    assert "Done 13 utterances, failed for 0 (over 1 ranks)" in log1
    assert "Done 13 utterances, failed for 0 (over 2 ranks)" in log2
    k1 = [l.split()[0] for l in open(one / "xvector_t.scp")]
    k2 = [l.split()[0] for l in open(two / "xvector_t.scp")]
    assert k1 == k2 == [k for k, _ in utts]
    # rank shards: 7 + 6, contiguous (split_scp.pl rule), each in its own ark
    assert [l.split()[0] for l in open(two / "xvector_t.1.scp")] == k1[:7]
    assert [l.split()[0] for l in open(two / "xvector_t.2.scp")] == k1[7:]
    v1 = dict(kio.read_scp(str(one / "xvector_t.scp"), "vector"))
    v2 = dict(kio.read_scp(str(two / "xvector_t.scp"), "vector"))
    for k in k1:
        assert v1[k].tobytes() == v2[k].tobytes(), k
    # the concatenated per-rank arks ARE the one-rank ark (same records, same order)
    cat = (two / "xvector_t.1.ark").read_bytes() + (two / "xvector_t.2.ark").read_bytes()
    assert cat == (one / "xvector_t.1.ark").read_bytes()


def test_two_ranks_apply_the_mixture_rank_0_calibrated(tmp_path):
    """The c-vector network, where the calibration mixes the two fast arithmetics (xv_calibration.lite_mask): rank 0 measures on
    a sample of the whole list, arithmetic AND layer mask travel to the other rank, and the two-rank job writes the bytes of
    the one-rank job."""
    import ast
    import re
    net, line = H.synth_model("v5_cvector", 123)
    (tmp_path / "final.raw").write_bytes(net.to_bytes(True))
    lens = [400, 333, 350] * 6 + [200, 120, 400, 64]
    utts = [("utt%02d" % i, H.features(80 + i, T)) for i, T in enumerate(lens)]
    kio.write_ark_matrices(str(tmp_path / "feats.ark"), utts, scp_path=str(tmp_path / "feats.scp"))
    node = ["--output-node", line.split("input=")[1]]     # (argparse keeps the last --output-node)

    def cal_of(log):
        return ast.literal_eval(re.search(r"rank 0 calibration: (\{.*\})", log).group(1))
    extra = ["--backend", "gloo", "--force-device", "0", "--calibrate", "true"] + node
    one, log1 = _run_dist_extract(tmp_path, 1, "probe", extra)
    cal = cal_of(log1)
    if not cal.get("lite_mask"):      # fp16mx passed outright on this sample: ask again with a tolerance it misses
        assert cal["chosen"] == "fp16mx", cal
        extra += ["--calibrate-tol", "%.3e" % (0.9 * cal["err_mx"])]
        one, log1 = _run_dist_extract(tmp_path, 1, "one", extra)
        cal = cal_of(log1)
    assert cal["chosen"] == "fp16mx2" and cal["lite_mask"], cal
    two, log2 = _run_dist_extract(tmp_path, 2, "two", extra)
    assert cal_of(log2) == cal
    assert "Done 22 utterances, failed for 0 (over 2 ranks)" in log2
    v1 = dict(kio.read_scp(str(one / "xvector_t.scp"), "vector"))
    v2 = dict(kio.read_scp(str(two / "xvector_t.scp"), "vector"))
    assert list(v1) == list(v2) == [k for k, _ in utts]
    for k in v1:
        assert v1[k].tobytes() == v2[k].tobytes(), k
    # the same choice as a FILE (--calibration, csrc/calib_file.h): a one-rank job publishes it, a two-rank job and the plain
    # command-line tool read it - three launchers, one arithmetic, the same bytes
    calib = str(tmp_path / "xvec.calib")
    shared = [a for a in extra if a not in ("--calibrate", "true")] + ["--calibration", calib]
    f1, flog1 = _run_dist_extract(tmp_path, 1, "file_one", shared)
    assert "published" in flog1 and os.path.exists(calib), flog1[-1500:]
    f2, flog2 = _run_dist_extract(tmp_path, 2, "file_two", shared)
    assert "(read %s)" % calib in flog2 and "rank 0 calibration" not in flog2, flog2[-1500:]
    w1 = dict(kio.read_scp(str(f1 / "xvector_t.scp"), "vector"))
    w2 = dict(kio.read_scp(str(f2 / "xvector_t.scp"), "vector"))
    exe = os.path.join(H.ROOT, H.PKG_NAME, "bin", "nnet3-xvector-compute")
    r = subprocess.run([exe, "--min-chunk-size=25", "--chunk-size=10000", "--calibration=" + calib] + ["--output-node=" + node[1]] +
                       [str(tmp_path / "final.raw"), "scp:%s/feats.scp" % tmp_path, "ark:%s/cli.ark" % tmp_path],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0 and "read from" in r.stderr, r.stderr[-1500:]
    w3 = dict(kio.read_ark(str(tmp_path / "cli.ark"), "vector"))
    for k in w1:
        assert w1[k].tobytes() == w2[k].tobytes() == w3[k].tobytes() == v1[k].tobytes(), k


def test_bench_two_ranks_on_one_gpu_prints_one_line():
    """bench.py's N > 1 path (process group, size + weight broadcast, per-rank context from the broadcast image, barrier +
    max-over-ranks timing) executed for real: two ranks sharing device 0 over gloo.  The throughput of such a run means
    nothing; the contract does: exactly one JSON line, from rank 0, with n_gpus = 2."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(H.ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--no-extra-modes", "--no-cpu-baseline"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900,
                       env=dict(os.environ, BENCH_DIST_BACKEND="gloo", BENCH_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
                                OMP_NUM_THREADS="1"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    assert d["roofline"]["frac"] > 0 and d["parity_rel_err_vs_oracle_fp32"] < 1e-4


def test_ctx_create_broadcast_over_two_devices():
    """xv_ctx_create_broadcast with n = 2: ncclCommInitAll over two devices, one ncclBroadcast of the packed image over
    xGMI, one context per GPU built from the image the broadcast left there; both must reproduce xv_ctx_create bit for
    bit.  Needs two visible GPUs (the test boxes have one: skipped there, runs on the driver's 8-GPU node)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    P = H.pkg()
    net, line = H.synth_model("v2_xvector")
    model = P.Model(raw=net.to_bytes(True), nnet_config=line)
    utts = [H.features(i, T) for i, T in enumerate((400, 57, 333))]
    feats, offs = H.pack(utts)
    want = P.Context(model).forward_batch(feats, offs)
    ctxs = P.create_broadcast(model, [0, 1])
    assert [c.device for c in ctxs] == [0, 1]
    for c in ctxs:
        assert np.array_equal(c.forward_batch(feats, offs), want)


def _two_gpus():
    import torch
    return torch.cuda.device_count() >= 2


def test_bench_two_gpus_over_rccl():
    """bench.py --gpus 2 as the driver launches it: fresh child processes under torch.distributed.run, one rank per GPU, the
    `nccl` (= RCCL) backend, the weight image broadcast over xGMI, max-over-ranks timing, one JSON line whose value is the
    whole job's.  Needs two visible GPUs (the one-GPU test boxes skip it; the driver's 8-GPU node runs it)."""
    import json
    if not _two_gpus():
        pytest.skip("needs two GPUs")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(H.ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
           "--no-extra-modes", "--no-cpu-baseline"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["parity_rel_err_vs_oracle_fp32"] < 1e-4
    # two GPUs process two batches per step: well above what one of them delivers alone on this workload
    assert d["value"] > 1.5 * 256 / (d["ms_per_step"] * 1e-3) / 2 * 0.99


def test_dist_extract_two_gpus_over_rccl_equal_one_rank(tmp_path):
    """dist_extract.py --backend nccl with two ranks on two devices against a one-rank run of the same job: same keys in the
    same order, every vector byte-identical, the concatenated per-rank arks are the one-rank ark (utils/split_scp.pl:208-217,
    extract_xvectors_new.sh:99) - with the weights carried by a real two-rank RCCL broadcast."""
    if not _two_gpus():
        pytest.skip("needs two GPUs")
    net, _ = H.synth_model("v2_xvector")
    (tmp_path / "final.raw").write_bytes(net.to_bytes(True))
    lens = [400, 137, 25, 333, 1000, 64, 400, 400, 211, 15, 399, 640, 87] + [400] * 20
    utts = [("utt%02d" % i, H.features(900 + i, T)) for i, T in enumerate(lens)]
    kio.write_ark_matrices(str(tmp_path / "feats.ark"), utts, scp_path=str(tmp_path / "feats.scp"))
    one, log1 = _run_dist_extract(tmp_path, 1, "one", ["--backend", "nccl"])
    two, log2 = _run_dist_extract(tmp_path, 2, "two", ["--backend", "nccl"])
    assert "Done 33 utterances, failed for 0 (over 2 ranks)" in log2, log2[-2000:]
    k1 = [l.split()[0] for l in open(one / "xvector_t.scp")]
    assert k1 == [l.split()[0] for l in open(two / "xvector_t.scp")] == [k for k, _ in utts]
    v1 = dict(kio.read_scp(str(one / "xvector_t.scp"), "vector"))
    v2 = dict(kio.read_scp(str(two / "xvector_t.scp"), "vector"))
    for k in k1:
        assert v1[k].tobytes() == v2[k].tobytes(), k
    cat = (two / "xvector_t.1.ark").read_bytes() + (two / "xvector_t.2.ark").read_bytes()
    assert cat == (one / "xvector_t.1.ark").read_bytes()


def _cli_job(tmp_path, n_utts=150):
    net, line = H.synth_model("v2_xvector")
    (tmp_path / "final.raw").write_bytes(net.to_bytes(True))
    lens = [400, 137, 25, 333, 1000, 64, 400, 400, 211, 15, 399, 640, 87, 0, 10]
    utts = [("utt%03d" % i, H.features(900 + i, lens[i % len(lens)]) if lens[i % len(lens)] else np.zeros((0, 23), np.float32))
            for i in range(n_utts)]
    kio.write_ark_matrices(str(tmp_path / "feats.ark"), utts, scp_path=str(tmp_path / "feats.scp"))
    return utts


def _cli(tmp_path, tag, extra, env=None):
    exe = os.path.join(H.ROOT, H.PKG_NAME, "bin", "nnet3-xvector-compute")
    # small batches, so that a 150-utterance job is many batches and several devices all get some
    cmd = [exe, "--use-gpu=yes", "--min-chunk-size=25", "--chunk-size=10000", "--output-node=tdnn6.affine", "--batch-frames=4096", "--calibrate=true"] + extra + \
          [str(tmp_path / "final.raw"), "scp:%s/feats.scp" % tmp_path, "ark,scp:%s/%s.ark,%s/%s.scp" % (tmp_path, tag, tmp_path, tag)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env or {})))
    return r


def test_cli_devices_option_one_device_is_the_one_gpu_job_byte_for_byte(tmp_path):
    """nnet3-xvector-compute --devices=0 (and XVEC_DEVICES=all on a one-GPU box): the in-binary multi-GPU path with one device -
    a real ncclBroadcast on a one-rank communicator, the engine built from the device copy, the round-robin dealer with one
    engine - writes the archive and the script file of the plain one-GPU job, byte for byte (extract_xvectors_new.sh:83-93
    at --nj 1; SURVEY.md section 7 hard part 8)."""
    _cli_job(tmp_path)
    r0 = _cli(tmp_path, "plain", [])
    assert r0.returncode == 0, r0.stderr[-2000:]
    r1 = _cli(tmp_path, "dev0", ["--devices=0"])
    assert r1.returncode == 0 and "one RCCL broadcast" in r1.stderr, r1.stderr[-2000:]
    assert (tmp_path / "dev0.ark").read_bytes() == (tmp_path / "plain.ark").read_bytes()
    assert (tmp_path / "dev0.scp").read_text().replace("dev0.ark", "plain.ark") == (tmp_path / "plain.scp").read_text()
    import torch
    if torch.cuda.device_count() == 1:
        r2 = _cli(tmp_path, "all", [], env={"XVEC_DEVICES": "all"})
        assert r2.returncode == 0 and "1 of 1 devices" in r2.stderr, r2.stderr[-2000:]
        assert (tmp_path / "all.ark").read_bytes() == (tmp_path / "plain.ark").read_bytes()
    # a bad list is an error before anything is written
    r3 = _cli(tmp_path, "bad", ["--devices=0,0"])
    assert r3.returncode == 255 and "twice" in r3.stderr, r3.stderr[-1000:]
    r4 = _cli(tmp_path, "bad", ["--devices=97"])
    assert r4.returncode == 255 and "visible devices" in r4.stderr, r4.stderr[-1000:]


def test_cli_devices_all_on_a_multi_gpu_node_equals_one_gpu(tmp_path):
    """The same with every GPU of the node (skipped on one-GPU boxes; first runs on the driver's 8-GPU node): batches dealt
    round-robin to N engines, one ordered writer - byte-identical to the one-GPU job, whatever N."""
    if not _two_gpus():
        pytest.skip("needs two GPUs")
    _cli_job(tmp_path, 600)
    r0 = _cli(tmp_path, "plain", ["--device=0"])
    assert r0.returncode == 0, r0.stderr[-2000:]
    r1 = _cli(tmp_path, "two", ["--devices=0,1"])
    assert r1.returncode == 0 and "2 of" in r1.stderr, r1.stderr[-2000:]
    assert (tmp_path / "two.ark").read_bytes() == (tmp_path / "plain.ark").read_bytes()
    r2 = _cli(tmp_path, "all", ["--devices=all"])
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert (tmp_path / "all.ark").read_bytes() == (tmp_path / "plain.ark").read_bytes()


def test_cli_several_engines_threaded_consumers_write_the_one_engine_archive(tmp_path):
    """The several-engine table loop (a consumer thread per engine, batches dealt round-robin, the calling thread writes in
    table order) on a ONE-GPU box: XVEC_DEBUG=engines_on_one_device=3 builds three engines on device 0 (test knob, no RCCL).  Archive
    and script file are those of the one-engine job byte for byte - with calibration on (the choice of engine 0 is shared), with
    utterances that fail in different places (empty, too short, wrong dimension), for two batch sizes."""
    utts = _cli_job(tmp_path, 400)
    # a wrong-dimension utterance and an empty one in the middle of the table: rejected by the consumer, the rest unaffected
    bad = [("bad_dim", np.ones((300, 24), np.float32)), ("zero_len", np.zeros((0, 23), np.float32))]
    allu = utts[:200] + bad + utts[200:]
    kio.write_ark_matrices(str(tmp_path / "feats.ark"), allu, scp_path=str(tmp_path / "feats.scp"))
    for bf in ("4096", "20000"):
        r0 = _cli(tmp_path, "one" + bf, ["--batch-frames=" + bf])
        assert r0.returncode == 0, r0.stderr[-2000:]
        r1 = _cli(tmp_path, "three" + bf, ["--batch-frames=" + bf], env={"XVEC_DEBUG": "engines_on_one_device=3", "XVEC_TIMING": "1"})
        assert r1.returncode == 0 and "3 engines on device" in r1.stderr, r1.stderr[-2000:]
        assert "summed over the engines' threads" in r1.stderr
        assert (tmp_path / ("three%s.ark" % bf)).read_bytes() == (tmp_path / ("one%s.ark" % bf)).read_bytes()
        assert (tmp_path / ("three%s.scp" % bf)).read_text().replace("three", "one") == (tmp_path / ("one%s.scp" % bf)).read_text()
        done = lambda r: [l for l in r.stderr.splitlines() if "Done " in l][-1]   # noqa: E731
        assert done(r0).split("Done")[-1] == done(r1).split("Done")[-1], (done(r0), done(r1))
        assert "bad_dim" in r1.stderr
    # the device front-end (sliding CMN + voiced-frame selection, the two pipe stages of extract_xvectors_new.sh:79) and the speaker-
    # level back-end under the consumer threads: the VAD table is shared by them (one reader, locked), an utterance without voiced
    # frames and one without a VAD entry fail in the consumer that meets them
    from oracle import frontend as fe
    vads = [(k, fe.synthetic_vad(i, x.shape[0])) for i, (k, x) in enumerate(allu) if x.shape[0] > 0 and k not in ("bad_dim", "utt007")]
    vads[3] = (vads[3][0], np.zeros_like(vads[3][1]))
    kio.write_ark_vectors(str(tmp_path / "vad.ark"), vads, scp_path=str(tmp_path / "vad.scp"))
    rng = np.random.default_rng(3)
    with open(tmp_path / "mean.vec", "wb") as f:
        f.write(b"\0B")
        kio.write_vector(f, rng.standard_normal(512).astype(np.float32) * 0.1)
    fe_args = ["--cmn-window=300", "--vad-rspecifier=scp:%s/vad.scp" % tmp_path, "--backend-mean=%s/mean.vec" % tmp_path,
               "--backend-normalize-length=true"]
    q0 = _cli(tmp_path, "fe_one", fe_args)
    assert q0.returncode == 0, q0.stderr[-2000:]
    q1 = _cli(tmp_path, "fe_three", fe_args, env={"XVEC_DEBUG": "engines_on_one_device=3"})
    assert q1.returncode == 0 and "3 engines on device" in q1.stderr, q1.stderr[-2000:]
    assert (tmp_path / "fe_three.ark").read_bytes() == (tmp_path / "fe_one.ark").read_bytes()
    assert done(q0).split("Done")[-1] == done(q1).split("Done")[-1], (done(q0), done(q1))
    assert "No VAD input found for utterance utt007" in q1.stderr and "voiced" in q1.stderr
    # a damaged archive is fatal in the threaded path too: non-zero exit, no hang, what was written before is intact vectors
    blob = (tmp_path / "feats.ark").read_bytes()
    (tmp_path / "cut.ark").write_bytes(blob[:len(blob) // 2 + 7])
    exe = os.path.join(H.ROOT, H.PKG_NAME, "bin", "nnet3-xvector-compute")
    r = subprocess.run([exe, "--use-gpu=yes", "--min-chunk-size=25", "--chunk-size=10000", "--output-node=tdnn6.affine",
                        "--batch-frames=4096", str(tmp_path / "final.raw"), "ark:%s/cut.ark" % tmp_path, "ark:%s/cut_out.ark" % tmp_path],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600,
                       env=dict(os.environ, XVEC_DEBUG="engines_on_one_device=2"))
    assert r.returncode != 0, r.stderr[-1500:]
    got = list(kio.read_ark(str(tmp_path / "cut_out.ark"), "vector")) if os.path.getsize(tmp_path / "cut_out.ark") else []
    ref = dict(kio.read_ark(str(tmp_path / "one4096.ark"), "vector"))
    for k, v in got:
        np.testing.assert_array_equal(v, ref[k])
