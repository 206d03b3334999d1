"""The host-side parsers (model files, feature archives: untrusted input) under AddressSanitizer + UBSan.

`make sanitize` builds csrc/host_selftest_main.cc with plain g++ -fsanitize=address,undefined (no HIP involved; GPU
sanitizers are not available on the target pool).  The driver round-trips a model and an archive and then replays them
truncated at 60 positions and with 90 single-bit flips each (200 / 300 when run by hand): every variant must parse or be rejected with KioError;
a crash, a hang or a sanitizer report fails the test."""
import os
import subprocess

import numpy as np
import pytest

import helpers as H
from oracle import kaldi_io as kio

CSRC = os.path.join(H.ROOT, H.PKG_NAME, "csrc")
EXE = os.path.join(CSRC, "build", "host_selftest_asan")


@pytest.fixture(scope="module")
def selftest():
    r = subprocess.run(["make", "-C", CSRC, "sanitize"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and os.path.exists(EXE), r.stdout[-2000:]
    return EXE


@pytest.mark.parametrize("binary_model,ark_form", [(True, "FM"), (False, "CM2"), (True, "text"), (True, "CM")])
def test_parsers_survive_damaged_input_under_asan_ubsan(selftest, tmp_path, binary_model, ark_form):
    net = H.nm.synthesize([H.tiny_config()], seed=3)
    (tmp_path / "tiny.raw").write_bytes(net.to_bytes(binary_model))
    rng = np.random.default_rng(2)
    utts = [("u%d" % i, rng.standard_normal((T, 5)).astype(np.float32)) for i, T in enumerate([40, 3, 17])]
    if ark_form == "text":
        kio.write_ark_matrices(str(tmp_path / "f.ark"), utts, binary=False)
    else:
        kio.write_ark_matrices(str(tmp_path / "f.ark"), utts, compressed=None if ark_form == "FM" else ark_form)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([selftest, str(tmp_path / "tiny.raw"), "tdnn6.affine", str(tmp_path / "f.ark"), "60"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "host_selftest: ok" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
