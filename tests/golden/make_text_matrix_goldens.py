#!/usr/bin/env python3
"""Generates tests/golden/text_matrix/cases.json: the Kaldi TEXT matrix form as the reference's own Python helpers write and read
it (egs/sre/v2/steps/libs/common.py:333-470 - the only code in the reference tree that touches a Kaldi table format; SURVEY.md
section 8(c)).  Run in the build container (imports common.py from /root/reference; nothing of it is copied - the fixture holds
matrices, the bytes those helpers WROTE for them, and what those helpers READ back from the bytes our writers emit):

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_text_matrix_goldens.py

  written_by_reference   archive text of write_matrix_ascii(fd, mat, key) for every matrix, one after the other
  object_by_reference    write_kaldi_matrix(file, mat) of the first matrix with integer entries ("[ a b\\n c d ]", whole-file object)
  ours_oracle / ours_cpp the text archive oracle/kaldi_io.py and bin/copy-feats (`ark,t:`) write for the same matrices
  reference_read_*       read_mat_ark(...) of those two, as the reference parsed them"""
import base64
import io
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True
from oracle import kaldi_io as kio  # noqa: E402
import helpers as H  # noqa: E402


def main():
    # the reference's helpers are Python 2 (`except IOError, ValueError:`): converted in memory with lib2to3, the way SURVEY.md
    # section 8(c) imports the xconfig library; nothing is written next to the reference or into this repository
    import types
    from lib2to3 import refactor
    src = open("/root/reference/egs/sre/v2/steps/libs/common.py").read()
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
    ref = types.ModuleType("ref_common")
    exec(compile(str(tool.refactor_string(src, "common.py")), "common.py(2to3)", "exec"), ref.__dict__)
    rng = np.random.default_rng(20180105)
    mats = [("one", np.array([[1.5]], np.float32)),
            ("ints", np.array([[1, -2, 3], [40, 5, -600]], np.float32)),
            ("mfcc", (rng.standard_normal((7, 23)) * 8 * 0.9 ** np.arange(23)).astype(np.float32)),
            ("tiny-and-big", np.array([[1e-7, -3.25e-5, 123456.75, -0.0], [2.5e6, 1.0 / 3.0, -7e-4, 9.0]], np.float32))]
    d = tempfile.mkdtemp()
    # the reference writes
    p = os.path.join(d, "ref.txt")
    with open(p, "w") as fd:
        for k, m in mats:
            ref.write_matrix_ascii(fd, m.tolist(), key=k)
    written = open(p, "rb").read()
    p2 = os.path.join(d, "obj.txt")
    ref.write_kaldi_matrix(p2, mats[1][1].astype(int).tolist())
    obj = open(p2, "rb").read()
    # ours: the python oracle's writer and the C++ tool's
    o = io.BytesIO()
    for k, m in mats:
        o.write(k.encode() + b" ")
        kio.write_matrix(o, m, binary=False)
    ours_oracle = o.getvalue()
    kio.write_ark_matrices(os.path.join(d, "in.ark"), mats)
    exe = os.path.join(ROOT, H.PKG_NAME, "bin", "copy-feats")
    subprocess.check_call([exe, "ark:%s/in.ark" % d, "ark,t:%s/cpp.txt" % d], stderr=subprocess.DEVNULL)
    ours_cpp = open(os.path.join(d, "cpp.txt"), "rb").read()

    def ref_read(b):
        q = os.path.join(d, "x.txt")
        open(q, "wb").write(b)
        return [[k, m] for k, m in ref.read_mat_ark(q)]

    out = {"matrices": [[k, m.tolist()] for k, m in mats],
           "written_by_reference": base64.b64encode(written).decode(), "object_by_reference": base64.b64encode(obj).decode(),
           "ours_oracle": base64.b64encode(ours_oracle).decode(), "ours_cpp": base64.b64encode(ours_cpp).decode(),
           "reference_read_ours_oracle": ref_read(ours_oracle), "reference_read_ours_cpp": ref_read(ours_cpp)}
    os.makedirs(os.path.join(HERE, "text_matrix"), exist_ok=True)
    json.dump(out, open(os.path.join(HERE, "text_matrix", "cases.json"), "w"))
    print(written.decode()[:300])
    print(ours_cpp.decode()[:300])


if __name__ == "__main__":
    main()
