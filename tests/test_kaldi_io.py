"""CPU tests of the Kaldi I/O layers: hand-assembled format KATs, the text form that the
reference's own Python helpers use (egs/sre/v2/steps/libs/common.py:354-470), and the C++ `kio` implementation
(through the copy-feats / copy-vector tools built from csrc/) against the independent Python one in oracle/."""
import io
import os
import struct
import subprocess

import numpy as np
import pytest

import helpers as H
from oracle import kaldi_io as kio

BIN = os.path.join(H.ROOT, H.PKG_NAME, "bin")


def _run(tool, *args, **kw):
    return subprocess.run([os.path.join(BIN, tool)] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)


def test_kat_binary_float_matrix_and_vector():
    blob = b"utt1 \x00BFM \x04\x02\x00\x00\x00\x04\x03\x00\x00\x00" + struct.pack("<6f", 1, 2, 3, 4, 5, 6)
    (key, m), = list(kio.read_ark(io.BytesIO(blob)))
    assert key == "utt1" and m.shape == (2, 3) and np.array_equal(m, [[1, 2, 3], [4, 5, 6]])
    vb = b"spk \x00BFV \x04\x03\x00\x00\x00" + struct.pack("<3f", 0.5, -1.0, 2.0)
    (key, v), = list(kio.read_ark(io.BytesIO(vb), "vector"))
    assert key == "spk" and np.array_equal(v, [0.5, -1.0, 2.0])
    out = io.BytesIO()
    out.write(b"spk ")
    out.write(b"\x00B")
    kio.write_vector(out, v)
    assert out.getvalue() == vb


def test_kat_compressed_matrix_formats():
    # CM2: uint16 row-major, value = min + range*u/65535 ; CM3: uint8, /255 ; CM: per-column percentile headers
    hdr = struct.pack("<ffii", -1.0, 2.0, 2, 2)
    cm2 = b"CM2 " + hdr + struct.pack("<4H", 0, 65535, 32768, 16384)
    m = kio.read_matrix(io.BytesIO(cm2))
    assert np.allclose(m, [[-1.0, 1.0], [-1 + 2 * 32768 / 65535, -1 + 2 * 16384 / 65535]], atol=1e-6)
    cm3 = b"CM3 " + hdr + bytes([0, 255, 51, 102])
    m = kio.read_matrix(io.BytesIO(cm3))
    assert np.allclose(m, [[-1.0, 1.0], [-1 + 2 * 51 / 255, -1 + 2 * 102 / 255]], atol=1e-6)
    # CM: one column, percentiles (0, 16384, 49152, 65535) -> (-1, -0.5, 0.5, 1); bytes 0/64/192/255 hit them
    cm = b"CM " + struct.pack("<ffii", -1.0, 2.0, 4, 1) + struct.pack("<4H", 0, 16384, 49152, 65535) + bytes([0, 64, 192, 255])
    m = kio.read_matrix(io.BytesIO(cm))
    assert np.allclose(m[:, 0], [-1.0, -0.5, 0.5, 1.0], atol=2e-5)


def test_text_matrix_form_matches_reference_python_helpers():
    # what steps/libs/common.py:write_matrix_ascii emits / read_matrix_ascii accepts: "key [\n  a b\n  c d ]\n"
    text = b"utt [\n  1 2 3\n  4 5 6 ]\n"
    (key, m), = list(kio.read_ark(io.BytesIO(text)))
    assert key == "utt" and np.array_equal(m, [[1, 2, 3], [4, 5, 6]])
    out = io.BytesIO()
    kio.write_matrix(out, m, binary=False)
    (k2, m2), = list(kio.read_ark(io.BytesIO(b"utt" + out.getvalue())))
    assert np.array_equal(m, m2)


@pytest.fixture(scope="module")
def feats(tmp_path_factory):
    d = tmp_path_factory.mktemp("io")
    utts = [("utt%03d" % i, H.features(i, 7 + 5 * i, 6)) for i in range(5)]
    utts.append(("empty", np.zeros((0, 0), np.float32)))
    return d, utts


@pytest.mark.parametrize("form", ["binary", "text", "CM", "CM2", "CM3", "double"])
def test_cpp_reader_all_matrix_formats(feats, form):
    d, utts = feats
    utts = [u for u in utts if u[1].size] if form.startswith("CM") else utts
    src = str(d / ("in_%s.ark" % form))
    expect = dict(utts)
    if form == "binary":
        kio.write_ark_matrices(src, utts)
    elif form == "text":
        kio.write_ark_matrices(src, utts, binary=False)
    elif form == "double":
        with open(src, "wb") as f:
            for k, m in utts:
                f.write(k.encode() + b" \x00B")
                kio.write_matrix(f, m, True, double=True)
    else:
        with open(src, "wb") as f:
            for k, m in utts:
                f.write(k.encode() + b" \x00B")
                expect[k] = kio.write_compressed_matrix(f, m, form)
    dst = str(d / ("out_%s.ark" % form))
    r = _run("copy-feats", "ark:" + src, "ark:" + dst)
    assert r.returncode == 0, r.stderr
    got = dict(kio.read_ark(dst))
    assert list(got) == [k for k, _ in utts]
    for k, m in expect.items():
        assert got[k].shape == m.shape or (m.size == 0 and got[k].size == 0)
        if m.size:
            assert np.allclose(got[k], m, rtol=0, atol=2e-6 * max(1.0, np.abs(m).max())), (form, k)
            if form in ("binary", "text"):
                assert np.array_equal(got[k], m)


def test_cpp_ark_scp_writer_offsets_pipes_and_text(feats):
    d, utts = feats
    utts = [u for u in utts if u[1].size]
    src = str(d / "w_in.ark")
    kio.write_ark_matrices(src, utts)
    ark, scp = str(d / "w_out.ark"), str(d / "w_out.scp")
    # feature rspecifier is a pipe (the form extract_xvectors_new.sh:79 uses), output is ark,scp
    r = _run("copy-feats", "ark:cat %s |" % src, "ark,scp:%s,%s" % (ark, scp))
    assert r.returncode == 0, r.stderr
    lines = open(scp).read().split("\n")[:-1]
    assert [l.split()[0] for l in lines] == [k for k, _ in utts]
    raw = open(ark, "rb").read()
    for line, (k, m) in zip(lines, utts):
        path, off = line.split()[1].rsplit(":", 1)
        assert path == ark and raw[int(off):int(off) + 2] == b"\x00B"        # offset points at the binary marker
        assert raw[int(off) - len(k) - 1:int(off)] == k.encode() + b" "
    got = dict(kio.read_scp(scp))
    for k, m in utts:
        assert np.array_equal(got[k], m)
    # scp in -> text ark out -> python reader
    txt = str(d / "w_out.txt")
    r = _run("copy-feats", "scp:" + scp, "ark,t:" + txt)
    assert r.returncode == 0, r.stderr
    got = dict(kio.read_ark(txt))
    for k, m in utts:
        assert np.allclose(got[k], m, rtol=2e-7)
    # vectors: binary -> scp+ark -> text
    vsrc = str(d / "v.ark")
    vecs = [("spk%d" % i, H.features(i, 1, 9)[0]) for i in range(4)]
    kio.write_ark_vectors(vsrc, vecs)
    r = _run("copy-vector", "ark:" + vsrc, "ark,scp:%s,%s" % (d / "v2.ark", d / "v2.scp"))
    assert r.returncode == 0, r.stderr
    got = dict(kio.read_scp(str(d / "v2.scp"), "vector"))
    for k, v in vecs:
        assert np.array_equal(got[k], v)


def test_cpp_reader_error_behaviour(feats):
    d, _ = feats
    r = _run("copy-feats", "ark:/nonexistent/file.ark", "ark:/dev/null")
    assert r.returncode == 255 and b"cannot open" in r.stderr
    bad = str(d / "bad.ark")
    open(bad, "wb").write(b"utt1 \x00BFM \x04\x02\x00\x00\x00\x04\x03\x00\x00\x00abc")     # truncated payload
    r = _run("copy-feats", "ark:" + bad, "ark:/dev/null")
    assert r.returncode == 255 and b"unexpected end" in r.stderr
    r = _run("copy-feats", "foo:" + bad, "ark:/dev/null")
    assert r.returncode == 255


def test_archive_through_a_pipe_is_drained_by_a_second_thread(tmp_path):
    """`ark:cmd |` - the form extract_xvectors_new.sh:79 feeds features in: a drain thread does the read(2) calls into a ring of
    1 MiB blocks, the parser consumes from it (kio.cc PipeDrain).  A large archive (many ring wrap-arounds, matrices that
    straddle blocks), compressed and text objects, a slow producer and an early close must all behave exactly like the plain
    stdio path (XVEC_DEBUG=pipe_drain=0) and like the file."""
    rng = np.random.default_rng(3)
    utts = [("u%04d" % i, rng.standard_normal((int(rng.integers(1, 900)), 23)).astype(np.float32)) for i in range(700)]   # ~29 MB
    src = str(tmp_path / "big.ark")
    kio.write_ark_matrices(src, utts)
    want = open(src, "rb").read()
    for env in ({}, {"XVEC_DEBUG": "pipe_drain=0"}):
        dst = str(tmp_path / ("out%d.ark" % len(env)))
        r = _run("copy-feats", "ark:cat %s |" % src, "ark:" + dst, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr.decode()[-500:]
        assert open(dst, "rb").read() == want
    # a producer that trickles (small writes, pauses): blocks arrive partly filled
    trickle = "python3 -c \"import sys,time; d=open('%s','rb').read()[:3000000]\nfor i in range(0,len(d),70001):\n  sys.stdout.buffer.write(d[i:i+70001]); sys.stdout.buffer.flush(); time.sleep(0.002)\"" % src
    head = [u for u in utts]
    n_bytes, n_head = 0, 0
    for k, m in head:
        n_bytes += len(k) + 1 + 2 + 15 + m.size * 4
        if n_bytes > 3000000:
            break
        n_head += 1
    dst = str(tmp_path / "trickle.ark")
    r = _run("copy-feats", "ark:%s |" % trickle, "ark:" + dst)
    # the stream ends inside a matrix: everything in front of it was copied, the cut is an error like in Kaldi
    assert r.returncode != 0 and b"unexpected end of file" in r.stderr, r.stderr.decode()[-300:]
    got = [k for k, _ in kio.read_ark(dst, "matrix")]
    assert got == [k for k, _ in utts[:n_head]], (len(got), n_head)
    # text archive through the pipe (byte-wise Peek / Get across block boundaries)
    txt = str(tmp_path / "t.ark")
    assert _run("copy-feats", "ark:" + src, "ark,t:" + txt).returncode == 0
    back = str(tmp_path / "back.ark")
    assert _run("copy-feats", "ark:cat %s |" % txt, "ark:" + back).returncode == 0
    b = dict(kio.read_ark(back, "matrix"))
    assert len(b) == len(utts) and all(np.allclose(b[k], m, rtol=1e-5, atol=1e-6) for k, m in utts[:50])


def _text_cases():
    import base64
    import json
    c = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "text_matrix", "cases.json")))
    for k in ("written_by_reference", "object_by_reference", "ours_oracle", "ours_cpp"):
        c[k] = base64.b64decode(c[k])
    c["matrices"] = [(k, np.array(m, np.float32)) for k, m in c["matrices"]]
    return c


def test_text_matrices_written_by_the_reference_helpers_are_read(tmp_path):
    """tests/golden/text_matrix/cases.json (made by tests/golden/make_text_matrix_goldens.py from the reference's own
    steps/libs/common.py:333-470, the one place in its tree that writes a Kaldi table): the archive text its write_matrix_ascii
    produced ("key [" + rows of "%f" + " ]" - one space, no indentation, where Kaldi's tools write "key  [" and two-space rows)
    is read by the oracle's reader and by the C++ one to the six decimals it carries; its write_kaldi_matrix object
    ("[ 1 -2 3\\n40 5 -600 ]") is read as a whole-file matrix by both."""
    c = _text_cases()
    got = dict(kio.read_ark(io.BytesIO(c["written_by_reference"])))
    assert list(got) == [k for k, _ in c["matrices"]]
    for k, m in c["matrices"]:
        assert got[k].shape == m.shape and np.abs(got[k] - m).max() <= 5.1e-7 * max(1.0, np.abs(m).max()), k
    (tmp_path / "ref.txt").write_bytes(c["written_by_reference"])
    r = _run("copy-feats", "ark:%s/ref.txt" % tmp_path, "ark:%s/ref.ark" % tmp_path)
    assert r.returncode == 0, r.stderr
    got_cpp = dict(kio.read_ark(str(tmp_path / "ref.ark")))
    for k, m in c["matrices"]:
        assert np.array_equal(got_cpp[k], got[k]), k          # both readers parse the same decimal strings to the same floats
    # the whole-file object
    M = kio.read_matrix(io.BytesIO(c["object_by_reference"]), binary=False)
    assert np.array_equal(M, c["matrices"][1][1])
    # (the C++ reader of whole-file objects sits behind the back-end tools, which need the GPU: tests/test_gpu_backend.py)


def test_text_matrices_we_write_are_what_the_reference_helpers_read(tmp_path):
    """The other direction, pinned at generation time: the bytes our two writers emit for the fixture's matrices (oracle/kaldi_io.py,
    bin/copy-feats ark,t:) are still the recorded ones, and what the reference's read_mat_ark parsed from those bytes equals the
    matrices (to float print precision)."""
    c = _text_cases()
    o = io.BytesIO()
    for k, m in c["matrices"]:
        o.write(k.encode() + b" ")
        kio.write_matrix(o, m, binary=False)
    assert o.getvalue() == c["ours_oracle"]
    kio.write_ark_matrices(str(tmp_path / "in.ark"), c["matrices"])
    r = _run("copy-feats", "ark:%s/in.ark" % tmp_path, "ark,t:%s/out.txt" % tmp_path)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "out.txt").read_bytes() == c["ours_cpp"]
    for name in ("reference_read_ours_oracle", "reference_read_ours_cpp"):
        parsed = c[name]
        assert [k for k, _ in parsed] == [k for k, _ in c["matrices"]]
        for (k, rows), (_, m) in zip(parsed, c["matrices"]):
            assert np.allclose(np.array(rows, np.float64), m.astype(np.float64), rtol=3e-7, atol=0), (name, k)


@pytest.mark.parametrize("form", ["FM", "CM", "CM2", "scp"])
def test_mapped_files_read_like_stdio(tmp_path, form):
    """Regular files are mapped read-only since round 6 (kio.cc Input::Open: the index pass of a table job skips through the
    mapping, sequential readers copy out of the page cache); XVEC_DEBUG=mmap=0 is the stdio path of rounds 1-5.  Same bytes out
    for float and compressed archives and for a script file of offsets - and the same error for an archive cut off inside its
    last object, at the same place (everything in front of it is written first)."""
    rng = np.random.default_rng(11)
    utts = [("u%03d" % i, rng.standard_normal((int(rng.integers(1, 300)), 23)).astype(np.float32)) for i in range(120)]
    src = str(tmp_path / "a.ark")
    kio.write_ark_matrices(src, utts, scp_path=str(tmp_path / "a.scp"), compressed=None if form in ("FM", "scp") else form)
    spec = "scp:%s/a.scp" % tmp_path if form == "scp" else "ark:" + src
    outs = []
    for dbg in ("mmap=1", "mmap=0", "mmap=0,readers=1"):
        dst = str(tmp_path / ("o_%s.ark" % dbg.replace("=", "").replace(",", "_")))
        r = _run("copy-feats", spec, "ark:" + dst, env=dict(os.environ, XVEC_DEBUG=dbg))
        assert r.returncode == 0, r.stderr.decode()[-500:]
        outs.append(open(dst, "rb").read())
    assert outs[0] == outs[1] == outs[2]
    got = list(kio.read_ark(str(tmp_path / "o_mmap1.ark"), "matrix"))
    assert [k for k, _ in got] == [k for k, _ in utts]
    if form in ("FM", "scp"):
        for (k, m), (_, want) in zip(got, utts):
            assert np.array_equal(m, want), k
    # truncated inside the last object
    blob = open(src, "rb").read()
    open(str(tmp_path / "cut.ark"), "wb").write(blob[:-7])
    errs = []
    for dbg in ("mmap=1", "mmap=0"):
        dst = str(tmp_path / ("cut_%s.ark" % dbg[-1]))
        r = _run("copy-feats", "ark:%s/cut.ark" % tmp_path, "ark:" + dst, env=dict(os.environ, XVEC_DEBUG=dbg))
        assert r.returncode == 255 and b"unexpected end" in r.stderr, r.stderr.decode()[-300:]
        errs.append([k for k, _ in kio.read_ark(dst, "matrix")])
    assert errs[0] == errs[1] == [k for k, _ in utts[:-1]]


def test_file_truncated_while_it_is_mapped_ends_like_an_input_error(tmp_path):
    """Mapped input (round 6): a file that someone truncates while a tool reads it raises SIGBUS where read(2) would have come
    back short.  The tools turn that into their usual ending - ERROR on stderr, exit status 255 - instead of dying of the signal.
    (A FIFO holds the writer side of the copy so that the truncation lands while the archive is still mapped and unread.)"""
    import signal
    import time
    rng = np.random.default_rng(5)
    utts = [("u%04d" % i, rng.standard_normal((400, 23)).astype(np.float32)) for i in range(2000)]     # 74 MB: many pages unread
    src = str(tmp_path / "big.ark")
    kio.write_ark_matrices(src, utts)
    fifo = str(tmp_path / "out.fifo")
    os.mkfifo(fifo)
    p = subprocess.Popen([os.path.join(H.ROOT, H.PKG_NAME, "bin", "copy-feats"), "ark:" + src, "ark:" + fifo], stderr=subprocess.PIPE)
    rd = os.open(fifo, os.O_RDONLY)          # the tool can open its output now; it blocks once the FIFO's buffer is full
    time.sleep(0.5)
    os.truncate(src, 4096)                   # everything behind the first page is gone
    got = 0
    while True:                              # drain: the tool reads on, into pages that no longer exist
        b = os.read(rd, 1 << 20)
        if not b:
            break
        got += len(b)
    os.close(rd)
    err = p.communicate(timeout=60)[1].decode()
    assert p.returncode == 255, (p.returncode, err[-300:])
    assert p.returncode != -signal.SIGBUS and "ERROR" in err and ("changed" in err or "unexpected end" in err), err[-300:]
