"""CPU test: the plain-C restatement (oracle/xvec_oracle.c) against the numpy graph evaluator - two independent
formulations of the same published semantics (both test infrastructure; parity with Kaldi itself is unpinned)."""
import os
import subprocess

import numpy as np
import pytest

import helpers as H
from oracle.export_program import export_program

EXE = os.path.join(H.ROOT, "oracle", "_build", "xvec_oracle_c")


@pytest.fixture(scope="module", autouse=True)
def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(H.ROOT, "oracle")])


@pytest.mark.parametrize("topology,T", [("v2_xvector", 15), ("v2_xvector", 64), ("v5_cvector", 21), ("v5_cvector", 50),
                                        ("pa_wo_pretrain", 33)])
def test_c_oracle_matches_numpy_oracle(tmp_path, topology, T):
    net, line = H.synth_model(topology)
    n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True))
    n2.apply_nnet_config(line)
    export_program(n2, str(tmp_path / "prog.bin"))
    x = H.features(T, T)
    x.tofile(str(tmp_path / "x.f32"))
    subprocess.check_call([EXE, str(tmp_path / "prog.bin"), str(tmp_path / "x.f32"), str(T), str(tmp_path / "o.f32")])
    got = np.fromfile(str(tmp_path / "o.f32"), np.float32)
    ref = H.xo.GraphEvaluator(n2, np.float64).compute(x)
    assert H.rel_err(got[None], ref) < 2e-5


def test_c_oracle_refuses_short_chunks(tmp_path):
    net, line = H.synth_model("v2_xvector")
    n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True))
    n2.apply_nnet_config(line)
    export_program(n2, str(tmp_path / "prog.bin"))
    H.features(1, 14).tofile(str(tmp_path / "x.f32"))
    r = subprocess.run([EXE, str(tmp_path / "prog.bin"), str(tmp_path / "x.f32"), "14", str(tmp_path / "o.f32")],
                       stderr=subprocess.PIPE)
    assert r.returncode == 2 and b"never pads" in r.stderr


def test_cpu_baseline_program_matches_numpy_oracle(tmp_path):
    """oracle/xvec_cpu_baseline (B0 of BASELINE.md section 3: blocked GEMM + OpenMP, ark in / ark out) against the numpy
    graph evaluator, ragged lengths, one and several threads; utterances shorter than the context count as failed."""
    from oracle import kaldi_io as kio
    exe = os.path.join(H.ROOT, "oracle", "_build", "xvec_cpu_baseline")
    net, line = H.synth_model("v2_xvector")
    n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True))
    n2.apply_nnet_config(line)
    export_program(n2, str(tmp_path / "prog.bin"))
    utts = [("u%d" % i, H.features(40 + i, T)) for i, T in enumerate((64, 15, 14, 137, 31))]
    kio.write_ark_matrices(str(tmp_path / "f.ark"), utts)
    ev = H.xo.GraphEvaluator(n2, np.float64)
    for threads in (1, 3):
        r = subprocess.run([exe, str(tmp_path / "prog.bin"), str(tmp_path / "f.ark"), str(tmp_path / "o.ark"), str(threads)],
                           stdout=subprocess.PIPE, check=True)
        assert b"4 utterances" in r.stdout and ("threads %d" % threads).encode() in r.stdout
        got = dict(kio.read_ark(str(tmp_path / "o.ark"), "vector"))
        assert sorted(got) == ["u0", "u1", "u3", "u4"]          # u2 (14 frames) is shorter than the 15-frame context
        for k, x in utts:
            if k in got:
                assert H.rel_err(got[k][None], ev.compute(x)) < 2e-5, k
