"""GPU tests of the speaker-level back-end: the device kernels through the C ABI and the four
drop-in tools chained exactly as egs/sre/v2/run_sre10.sh:238-241 chains Kaldi's, against oracle/backend.py."""
import os
import subprocess

import numpy as np
import pytest

import helpers as H
from oracle import backend as B
from oracle import kaldi_io as kio

pytestmark = pytest.mark.gpu
BIN = os.path.join(H.ROOT, H.PKG_NAME, "bin")
TOL = 1e-5   # fp32-grade: v_mfma_f32_16x16x4_f32 products with fp32 accumulation (K up to 3000) against the fp64 oracle


def _x(n, dim, seed):
    return (np.random.default_rng(seed).standard_normal((n, dim)) * 3).astype(np.float32)


@pytest.mark.parametrize("n,dim,rows,affine", [(1, 512, 150, False), (37, 512, 150, False), (200, 600, 200, True), (5, 23, 7, True),
                                               (16, 3000, 64, False), (70, 512, 300, True), (130, 100, 257, False)])
def test_backend_chain_matches_oracle(n, dim, rows, affine):
    P = H.pkg()
    x, mean = _x(n, dim, 1), _x(1, dim, 2)[0]
    t = (_x(rows, dim + (1 if affine else 0), 3) / np.sqrt(dim)).astype(np.float32)
    got, ratio = P.backend_apply(x, mean=mean, transform=t, normalize=True, return_ratio=True)
    ref, rref = B.backend_chain(x, mean, t, normalize=True)
    assert got.shape == (n, rows)
    assert H.rel_err(got, ref) < TOL
    np.testing.assert_allclose(ratio, rref, rtol=1e-5)
    np.testing.assert_allclose(np.linalg.norm(got, axis=1), np.sqrt(rows), rtol=1e-5)


def test_backend_stages_are_optional_and_zero_vectors_survive():
    P = H.pkg()
    x = _x(33, 512, 5)
    x[7] = 0
    np.testing.assert_array_equal(P.backend_apply(x), x)                                    # nothing to do: a copy
    np.testing.assert_array_equal(P.backend_apply(x, mean=x[3]), x - x[3])                  # bit-exact subtraction
    y, r = P.backend_apply(x, normalize=True, return_ratio=True)
    ref, rref = B.normalize_length(x)
    assert r[7] == 0 and not y[7].any()                                                     # "Zero iVector": left alone
    assert H.rel_err(np.delete(y, 7, 0), np.delete(ref, 7, 0)) < TOL
    y2 = P.backend_apply(x, normalize=True, scaleup=False)
    np.testing.assert_allclose(np.linalg.norm(np.delete(y2, 7, 0), axis=1), 1.0, rtol=1e-5)
    # same vector -> same bits whatever else is in the batch
    np.testing.assert_array_equal(P.backend_apply(x[:5], normalize=True), y[:5])


def test_backend_argument_errors():
    P = H.pkg()
    with pytest.raises(P.XvError, match="Dimension mismatch: input vector has dimension 8 and transform has 11 columns"):
        P.backend_apply(_x(2, 8, 1), transform=_x(3, 11, 2))
    with pytest.raises(P.XvError):
        P.backend_apply(_x(2, 8, 1), mean=np.zeros(9, np.float32))
    assert P.backend_apply(np.zeros((0, 8), np.float32), normalize=True).shape == (0, 8)


def test_segment_mean_is_bit_exact_with_the_sequential_fp32_loop():
    P = H.pkg()
    x = (_x(300, 512, 9) * 100).astype(np.float32)
    rng = np.random.default_rng(4)
    segs = [list(rng.integers(0, 300, k)) for k in (1, 2, 7, 64, 150)] + [[]]
    got = P.segment_mean(x, segs)
    for s, g in zip(segs, got):
        if not s:
            assert not g.any()
            continue
        acc = x[s[0]].copy()
        for i in s[1:]:
            acc = (acc + x[i]).astype(np.float32)
        np.testing.assert_array_equal(g, acc * np.float32(1.0 / len(s)))
    g64 = P.segment_mean(x, [list(range(300))], acc64=True)[0]
    np.testing.assert_array_equal(g64, B.global_mean(x))
    with pytest.raises(P.XvError):
        P.segment_mean(x, [[0, 300]])


@pytest.fixture(scope="module")
def job(tmp_path_factory):
    d = tmp_path_factory.mktemp("backend")
    rng = np.random.default_rng(21)
    utts = [("spk%02d-utt%d" % (s, u), (rng.standard_normal(512) * 2 + s).astype(np.float32)) for s in range(6) for u in range(1 + s % 4)]
    kio.write_ark_vectors(str(d / "xvector.ark"), utts, scp_path=str(d / "xvector.scp"))
    spk2utt = [("spk%02d" % s, [k for k, _ in utts if k.startswith("spk%02d-" % s)]) for s in range(6)]
    spk2utt[2][1].append("spk02-missing")                 # an utterance without a vector
    spk2utt.append(("spk99", ["spk99-utt0"]))             # a speaker without any
    (d / "spk2utt").write_text("".join("%s %s\n" % (s, " ".join(u)) for s, u in spk2utt))
    lda = (rng.standard_normal((150, 512)) / 20).astype(np.float32)
    with open(d / "transform.mat", "wb") as f:
        f.write(b"\0B")
        kio.write_matrix(f, lda)
    return d, utts, spk2utt, lda


def _run(args, **kw):
    return subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)


def test_ivector_mean_tools(job):
    d, utts, spk2utt, lda = job
    r = _run([os.path.join(BIN, "ivector-mean"), "ark:%s/spk2utt" % d, "scp:%s/xvector.scp" % d,
              "ark,scp:%s/spk.ark,%s/spk.scp" % (d, d), "ark,t:%s/num_utts.ark" % d])
    err = r.stderr.decode()
    assert r.returncode == 0, err
    ref, counts, missing, empty = B.speaker_means(spk2utt, dict(utts))
    got = list(kio.read_scp(str(d / "spk.scp"), "vector"))
    assert [k for k, _ in got] == [k for k, _ in ref]
    for (k, g), (_, m) in zip(got, ref):
        np.testing.assert_array_equal(g, m)                # fp32 accumulation in spk2utt order: bit-exact
    assert (d / "num_utts.ark").read_text() == "".join("%s %d \n" % (k, counts[k]) for k, _ in ref)
    assert "No iVector present in input for utterance spk02-missing" in err
    assert "Not producing output for speaker spk99 since no utterances had iVectors" in err
    assert "Computed mean of 6 speakers (1 with no utterances), consisting of %d utterances (2 absent from input)." % len(utts) in err
    # global mean -> a Kaldi vector object, binary by default, text on request
    r = _run([os.path.join(BIN, "ivector-mean"), "scp:%s/xvector.scp" % d, str(d / "mean.vec")])
    assert r.returncode == 0, r.stderr.decode()
    with open(d / "mean.vec", "rb") as f:
        assert f.read(2) == b"\0B"
        mean = kio.read_vector(f)
    np.testing.assert_array_equal(mean, B.global_mean(np.stack([v for _, v in utts])))
    r = _run([os.path.join(BIN, "ivector-mean"), "--binary=false", "ark:%s/xvector.ark" % d, str(d / "mean.txt")])
    assert r.returncode == 0 and (d / "mean.txt").read_text().startswith(" [ ")


def test_the_scoring_pipe_of_run_sre10(job):
    """ivector-mean | ivector-subtract-global-mean mean.vec | transform-vec | ivector-normalize-length, run_sre10.sh:238-240."""
    d, utts, spk2utt, lda = job
    assert _run([os.path.join(BIN, "ivector-mean"), "scp:%s/xvector.scp" % d, str(d / "mean.vec")]).returncode == 0
    pipe = ("%(b)s/ivector-mean ark:%(d)s/spk2utt scp:%(d)s/xvector.scp ark:- | "
            "%(b)s/ivector-subtract-global-mean %(d)s/mean.vec ark:- ark:- | "
            "%(b)s/transform-vec %(d)s/transform.mat ark:- ark:- | "
            "%(b)s/ivector-normalize-length ark:- ark,t:%(d)s/enroll.txt") % {"b": BIN, "d": d}
    r = _run(["bash", "-c", "set -o pipefail; " + pipe])
    err = r.stderr.decode()
    assert r.returncode == 0, err
    got = list(kio.read_ark(str(d / "enroll.txt"), "vector"))
    means, _, _, _ = B.speaker_means(spk2utt, dict(utts))
    gmean = B.global_mean(np.stack([v for _, v in utts]))
    ref, ratios = B.backend_chain(np.stack([m for _, m in means]), gmean, lda, normalize=True)
    assert [k for k, _ in got] == [k for k, _ in means]
    assert H.rel_err(np.stack([v for _, v in got]), ref) < 1e-5          # text output: 6-7 significant digits
    assert "Applied transform to 6 vectors." in err and "Processed 6 iVectors." in err
    assert "Wrote 6 mean-subtracted iVectors" in err
    assert "Average ratio of iVector to expected length was" in err
    # the test-side pipe (run_sre10.sh:240) with the mean of the input itself, straight from the scp
    r = _run(["bash", "-c", "set -o pipefail; %(b)s/ivector-subtract-global-mean scp:%(d)s/xvector.scp ark:- | "
              "%(b)s/ivector-normalize-length --scaleup=false ark:- ark,scp:%(d)s/t.ark,%(d)s/t.scp" % {"b": BIN, "d": d}])
    assert r.returncode == 0, r.stderr.decode()
    x = np.stack([v for _, v in utts])
    ref2, _ = B.backend_chain(x, B.global_mean(x), None, normalize=True, scaleup=False)
    got2 = list(kio.read_scp(str(d / "t.scp"), "vector"))
    assert [k for k, _ in got2] == [k for k, _ in utts]
    assert H.rel_err(np.stack([v for _, v in got2]), ref2) < TOL


def test_tool_errors(job):
    d, utts, spk2utt, lda = job
    with open(d / "bad.mat", "wb") as f:
        f.write(b"\0B")
        kio.write_matrix(f, np.zeros((4, 100), np.float32))
    r = _run([os.path.join(BIN, "transform-vec"), str(d / "bad.mat"), "ark:%s/xvector.ark" % d, "ark:/dev/null"])
    assert r.returncode == 255 and b"Dimension mismatch: input vector has dimension 512 and transform has 100 columns" in r.stderr
    (d / "empty.ark").write_bytes(b"")
    r = _run([os.path.join(BIN, "ivector-normalize-length"), "ark:%s/empty.ark" % d, "ark:/dev/null"])
    assert r.returncode == 1 and b"Processed 0 iVectors." in r.stderr
    r = _run([os.path.join(BIN, "ivector-normalize-length"), "ark:%s/xvector.ark" % d])
    assert r.returncode == 1 and b"Usage: ivector-normalize-length" in r.stderr
    r = _run([os.path.join(BIN, "transform-vec"), "--nosuch=1", "a", "b", "c"])
    assert r.returncode == 255 and b"Invalid option" in r.stderr


def test_fused_backend_options_of_the_extractor(tmp_path):
    """nnet3-xvector-compute --backend-*: the test-side pipe of run_sre10.sh:240 fused behind the extraction."""
    net, line = H.synth_model("v2_xvector")
    (tmp_path / "final.raw").write_bytes(net.to_bytes(True))
    utts = [("u%d" % i, H.features(900 + i, T)) for i, T in enumerate([150, 31, 400])]
    kio.write_ark_matrices(str(tmp_path / "f.ark"), utts)
    rng = np.random.default_rng(8)
    mean = rng.standard_normal(512).astype(np.float32)
    lda = (rng.standard_normal((150, 512)) / 20).astype(np.float32)
    with open(tmp_path / "mean.vec", "wb") as f:
        f.write(b"\0B")
        kio.write_vector(f, mean)
    with open(tmp_path / "transform.mat", "wb") as f:
        f.write(b"\0B")
        kio.write_matrix(f, lda)
    common = [os.path.join(BIN, "nnet3-xvector-compute"), "--min-chunk-size=25", "--output-node=tdnn6.affine"]
    r = _run(common + [str(tmp_path / "final.raw"), "ark:%s/f.ark" % tmp_path, "ark:%s/plain.ark" % tmp_path])
    assert r.returncode == 0, r.stderr.decode()
    r = _run(common + ["--backend-mean=%s/mean.vec" % tmp_path, "--backend-transform=%s/transform.mat" % tmp_path,
                       "--backend-normalize-length=true", str(tmp_path / "final.raw"), "ark:%s/f.ark" % tmp_path,
                       "ark:%s/fused.ark" % tmp_path])
    assert r.returncode == 0, r.stderr.decode()
    plain = list(kio.read_ark(str(tmp_path / "plain.ark"), "vector"))
    fused = list(kio.read_ark(str(tmp_path / "fused.ark"), "vector"))
    assert [k for k, _ in fused] == [k for k, _ in plain] == ["u0", "u1", "u2"]
    ref, _ = B.backend_chain(np.stack([v for _, v in plain]), mean, lda, normalize=True)
    got = np.stack([v for _, v in fused])
    assert got.shape == (3, 150) and H.rel_err(got, ref) < TOL
    # a mean of the wrong dimension is fatal
    with open(tmp_path / "bad.vec", "wb") as f:
        f.write(b"\0B")
        kio.write_vector(f, mean[:100])
    r = _run(common + ["--backend-mean=%s/bad.vec" % tmp_path, str(tmp_path / "final.raw"), "ark:%s/f.ark" % tmp_path, "ark:/dev/null"])
    assert r.returncode == 255 and b"--backend-mean has dimension 100" in r.stderr


def test_transform_vec_reads_the_text_matrix_the_reference_helpers_write(tmp_path):
    """tests/golden/text_matrix/cases.json `object_by_reference`: what steps/libs/common.py:333-352 write_kaldi_matrix wrote for a
    2 x 3 integer matrix ("[ 1 -2 3\\n40 5 -600 ]") is read as the transform of transform-vec: y = M x."""
    import base64
    import json
    c = json.load(open(os.path.join(H.ROOT, "tests", "golden", "text_matrix", "cases.json")))
    (tmp_path / "m.txt").write_bytes(base64.b64decode(c["object_by_reference"]))
    M = np.array(c["matrices"][1][1], np.float32)
    vecs = [("v%d" % i, H.features(i, 1, 3)[0]) for i in range(5)]
    kio.write_ark_vectors(str(tmp_path / "v.ark"), vecs)
    r = _run([os.path.join(BIN, "transform-vec"), str(tmp_path / "m.txt"), "ark:%s/v.ark" % tmp_path, "ark:%s/y.ark" % tmp_path])
    assert r.returncode == 0, r.stderr
    y = dict(kio.read_ark(str(tmp_path / "y.ark"), "vector"))
    for k, v in vecs:
        assert np.allclose(y[k], M @ v, rtol=1e-6), k
