"""GPU parity tests of the device front-end (sliding CMN + voiced-frame selection) against oracle/frontend.py, and
of the extractor CLI with the front-end fused in (replacing the two pipe stages of extract_xvectors_new.sh:79)."""
import os
import subprocess

import numpy as np
import pytest

import helpers as H
from oracle import frontend as fe
from oracle import kaldi_io as kio

pytestmark = pytest.mark.gpu
BIN = os.path.join(H.ROOT, H.PKG_NAME, "bin")


@pytest.fixture(scope="module")
def ctx():
    P = H.pkg()
    net, line = H.synth_model("v2_xvector")
    return P.Context(P.Model(raw=net.to_bytes(True), nnet_config=line))


@pytest.mark.parametrize("center", [True, False])
def test_sliding_cmn_and_vad_selection(ctx, center):
    lens = [1000, 299, 300, 301, 50, 1, 640]
    utts = [H.features(60 + i, T) + 3.0 for i, T in enumerate(lens)]       # non-zero mean so CMN matters
    vads = [fe.synthetic_vad(i, T) for i, T in enumerate(lens)]
    vads[5][:] = 1.0
    raw, offs = H.pack(utts)
    out, out_off = ctx.frontend(raw, offs, np.concatenate(vads), cmn_window=300, center=center)
    for i, (u, v) in enumerate(zip(utts, vads)):
        ref = fe.select_voiced(fe.sliding_cmn(u, 300, center), v)
        got = out[out_off[i]:out_off[i + 1]]
        if ref is None:
            assert got.shape[0] == 0
            continue
        assert got.shape == ref.shape, i
        # fp32 features of magnitude ~10: the subtraction is done in double on both sides
        assert np.abs(got - ref).max() <= 2e-6 * max(1.0, np.abs(u).max()), (i, np.abs(got - ref).max())
    # no VAD, no CMN: identity
    out2, off2 = ctx.frontend(raw, offs, None, cmn_window=0)
    assert np.array_equal(out2, raw) and np.array_equal(off2, offs)


@pytest.mark.parametrize("chunk", [10000, 150])   # 150: utterances are cut into chunks -> the host round-trip path
def test_cli_with_fused_front_end(tmp_path, chunk):
    net, line = H.synth_model("v2_xvector")
    (tmp_path / "final.raw").write_bytes(net.to_bytes(True))
    lens = [500, 120, 64, 900]
    utts = [("utt%d" % i, H.features(900 + i, T) + 1.5) for i, T in enumerate(lens)]
    vads = [("utt%d" % i, fe.synthetic_vad(30 + i, T)) for i, T in enumerate(lens)]
    vads[2] = ("utt2", np.zeros(64, np.float32))                          # nothing voiced -> skipped with a warning
    kio.write_ark_matrices(str(tmp_path / "feats.ark"), utts, scp_path=str(tmp_path / "feats.scp"))
    kio.write_ark_vectors(str(tmp_path / "vad.ark"), vads, scp_path=str(tmp_path / "vad.scp"))
    r = subprocess.run([os.path.join(BIN, "nnet3-xvector-compute"), "--use-gpu=yes", "--min-chunk-size=25", "--chunk-size=%d" % chunk,
                        "--output-node=tdnn6.affine", "--cmn-window=300", "--cmn-center=true",
                        "--vad-rspecifier=scp,s,cs:%s/vad.scp" % tmp_path, str(tmp_path / "final.raw"),
                        "scp:%s/feats.scp" % tmp_path, "ark,scp:%s/x.ark,%s/x.scp" % (tmp_path, tmp_path)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    err = r.stderr.decode()
    assert r.returncode == 0, err
    got = dict(kio.read_scp(str(tmp_path / "x.scp"), "vector"))
    n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True))
    n2.apply_nnet_config(line)
    ev = H.xo.GraphEvaluator(n2, np.float32)
    for (k, x), (_, v) in zip(utts, vads):
        f = fe.select_voiced(fe.sliding_cmn(x, 300, True), v)
        if f is None:
            assert k not in got
            continue
        ref = H.xo.extract_xvector(ev, f, chunk, 25, True)
        assert H.rel_err(got[k][None], ref[None]) < 1e-4, k
    assert "No features were judged as voiced for utterance utt2" in err
    assert "Done 3 utterances, failed for 1" in err


@pytest.mark.parametrize("chunk", [10000, 150])   # 150: cut into chunks -> the views are expanded on the host, same floats
def test_compressed_raw_features_are_expanded_on_the_device(tmp_path, chunk):
    """The recipes' raw features are stored compressed (steps/make_mfcc.sh: compress=true -> Kaldi "CM" matrices, one byte per
    element).  A table job with the fused front-end maps the archive, uploads the compressed objects as they are and expands them
    on the GPU (cm_expand_kernel) with the host reader's float operations: the archive it writes is BYTE-IDENTICAL to the one of
    the same job with the expansion done by the reader threads (XVEC_DEBUG=cm_on_device=0), and matches the oracle - which reads
    the compressed archive with its own reader - at the parity tolerance."""
    net, line = H.synth_model("v2_xvector")
    (tmp_path / "final.raw").write_bytes(net.to_bytes(True))
    lens = [500, 120, 333, 900, 64, 401, 77, 260, 640, 40]
    utts = [("utt%d" % i, H.features(900 + i, T) + 1.5) for i, T in enumerate(lens)]
    vads = [("utt%d" % i, fe.synthetic_vad(30 + i, T)) for i, T in enumerate(lens)]
    vads[4] = ("utt4", np.zeros(64, np.float32))                          # nothing voiced -> skipped with a warning
    few = np.zeros(40, np.float32)
    few[[3, 4, 5, 9, 17, 18, 19, 20, 30, 31, 33, 39]] = 1.0               # 12 voiced frames < --min-chunk-size: ONE chunk, padded by
    vads[9] = ("utt9", few)                                               # edge replication to 25 (--pad-input) - on the device too
    kio.write_ark_matrices(str(tmp_path / "feats.ark"), utts, scp_path=str(tmp_path / "feats.scp"), compressed="CM")
    kio.write_ark_vectors(str(tmp_path / "vad.ark"), vads, scp_path=str(tmp_path / "vad.scp"))
    stored = dict(kio.read_scp(str(tmp_path / "feats.scp"), "matrix"))     # what a conforming reader reconstructs
    outs = {}
    for tag, dbg in (("device", None), ("host", "cm_on_device=0")):
        env = dict(os.environ, XVEC_TIMING="1")
        if dbg:
            env["XVEC_DEBUG"] = dbg
        r = subprocess.run([os.path.join(BIN, "nnet3-xvector-compute"), "--use-gpu=yes", "--min-chunk-size=25", "--chunk-size=%d" % chunk,
                            "--output-node=tdnn6.affine", "--cmn-window=300", "--cmn-center=true", "--batch-frames=2000",
                            "--vad-rspecifier=scp,s,cs:%s/vad.scp" % tmp_path, str(tmp_path / "final.raw"),
                            "scp:%s/feats.scp" % tmp_path, "ark:%s/x_%s.ark" % (tmp_path, tag)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        err = r.stderr.decode()
        assert r.returncode == 0, err[-1500:]
        assert "Done 9 utterances, failed for 1" in err, err[-800:]
        outs[tag] = (open(tmp_path / ("x_%s.ark" % tag), "rb").read(), err)
    assert outs["device"][0] == outs["host"][0]
    # chunk = 150 cuts every utterance into several chunks and pads the short last ones: still selections of the kept rows, still
    # the device path (rounds 2-5 sent such batches back through the host)
    assert "utterances went to the device compressed" in outs["device"][1], outs["device"][1][-1500:]
    assert "took the host round trip" not in outs["device"][1] and "took the host round trip" not in outs["host"][1]
    assert "went to the device compressed" not in outs["host"][1]
    n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True))
    n2.apply_nnet_config(line)
    ev = H.xo.GraphEvaluator(n2, np.float32)
    got = dict(kio.read_ark(str(tmp_path / "x_device.ark"), "vector"))
    for (k, _), (_, v) in zip(utts, vads):
        x = fe.select_voiced(fe.sliding_cmn(stored[k], 300, True), v)
        if x is None:
            assert k not in got
            continue
        ref = H.xo.extract_xvector(ev, x, chunk, 25, True)
        assert H.rel_err(got[k][None], ref[None]) < 1e-4, k


def test_the_recipes_feature_pipeline_runs_on_the_device_without_a_script_change(tmp_path):
    """extract_xvectors_new.sh:79 hands the binary `ark:apply-cmvn-sliding --norm-vars=false --center=true --cmn-window=300
    scp:feats.scp ark:- | select-voiced-frames ark:- scp,s,cs:vad.scp ark:- |`.  nnet3-xvector-compute recognises exactly that
    string (csrc/fuse_pipe.h) and runs both stages on the device, reading the stored (compressed) features itself: neither tool
    exists on this box, the job succeeds, says so, and writes byte for byte what the explicit --cmn-window / --vad-rspecifier form
    writes (which is compared with the oracle above).  A pipeline it does not implement (--norm-vars=true) is left to the shell -
    where it fails here, for want of the tools; XVEC_DEBUG=fuse_pipe=0 does the same to the recognised one."""
    net, line = H.synth_model("v2_xvector")
    (tmp_path / "final.raw").write_bytes(net.to_bytes(True))
    (tmp_path / "extract.config").write_text(line + "\n")
    lens = [500, 120, 333, 900, 64, 401]
    utts = [("utt%d" % i, H.features(900 + i, T) + 1.5) for i, T in enumerate(lens)]
    vads = [("utt%d" % i, fe.synthetic_vad(30 + i, T)) for i, T in enumerate(lens)]
    vads[4] = ("utt4", np.zeros(64, np.float32))
    kio.write_ark_matrices(str(tmp_path / "feats.ark"), utts, scp_path=str(tmp_path / "feats.scp"), compressed="CM")
    kio.write_ark_vectors(str(tmp_path / "vad.ark"), vads, scp_path=str(tmp_path / "vad.scp"))
    model = "%s --nnet-config=%s/extract.config %s/final.raw - |" % (os.path.join(BIN, "nnet3-copy"), tmp_path, tmp_path)
    pipe = ("ark:apply-cmvn-sliding --norm-vars=false --center=true --cmn-window=300 scp:%s/feats.scp ark:- | "
            "select-voiced-frames ark:- scp,s,cs:%s/vad.scp ark:- |" % (tmp_path, tmp_path))

    def run(tag, feat, extra=(), env=None):
        # the script's own argv (extract_xvectors_new.sh:92-93), --use-gpu=no included
        return subprocess.run([os.path.join(BIN, "nnet3-xvector-compute"), "--use-gpu=no", "--min-chunk-size=25", "--chunk-size=10000"] +
                              list(extra) + [model, feat, "ark,scp:%s/x_%s.ark,%s/x_%s.scp" % (tmp_path, tag, tmp_path, tag)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, XVEC_TIMING="1", **(env or {})))
    a = run("pipe", pipe)
    err = a.stderr.decode()
    assert a.returncode == 0, err[-1500:]
    assert "feature pipeline recognised" in err and "runs on the device" in err, err[-1500:]
    assert "Done 5 utterances, failed for 1" in err and "No features were judged as voiced for utterance utt4" in err, err[-1500:]
    assert "utterances went to the device compressed" in err            # the stored bytes go up as they are
    b = run("flags", "scp:%s/feats.scp" % tmp_path, ["--cmn-window=300", "--cmn-center=true", "--vad-rspecifier=scp,s,cs:%s/vad.scp" % tmp_path])
    assert b.returncode == 0, b.stderr.decode()[-1500:]
    assert (tmp_path / "x_pipe.ark").read_bytes() == (tmp_path / "x_flags.ark").read_bytes()
    # not this pipeline: the shell gets it (and has no such tools here)
    c = run("other", pipe.replace("--norm-vars=false", "--norm-vars=true"))
    assert c.returncode != 0 and "feature pipeline recognised" not in c.stderr.decode()
    d = run("off", pipe, env={"XVEC_DEBUG": "fuse_pipe=0"})
    assert d.returncode != 0 and "feature pipeline recognised" not in d.stderr.decode()
