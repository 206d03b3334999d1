#!/usr/bin/env python3
"""Regenerate tests/golden/configs/*.config from the REFERENCE's own xconfig library.

Runs only in the build container (needs /root/reference); never on the GPU box.  It follows
SURVEY.md Appendix D: the reference's `steps/libs/nnet3/xconfig` is Python 2, so a scratch
copy is converted with lib2to3 (plus the two integer divisions at xconfig/utils.py:564,585)
under a temp dir, the xconfig heredocs are read out of the reference recipe scripts at
generation time, and the `final` config lines the library emits are written as fixtures.
Nothing from the reference is copied into the repository: the fixtures are the library's
*output* for a given topology (graph text only - no arithmetic).

The heredoc sources (reference file:lines):
  v2_xvector      egs/sre/v2/local/nnet3/xvector/run_xvector_new.sh:90-115
  am              egs/sre/v5/local/nnet3_cvector/cvector/train_am.sh:30-38
  v4_cvector      egs/sre/v4/local/nnet3_cvector/cvector/train_xvector_with_am.sh:43-56  (existing = am)
  v5_cvector      egs/sre/v5/local/nnet3_cvector/cvector/train_cvector_with_am.sh:65-89   (existing = am)
  v3_multitask    egs/sre/v3/local/nnet3_cvector/cvector/prepare_nnet3_xconfig.sh:46-70
  v3_{2,3,4}share egs/sre/v3/local/nnet3_cvector/cvector/prepare_nnet3_xconfig_{2,3,4}share.sh
  pa_wo_pretrain  egs/sre/v4/local/nnet3/xvector/run_xvector_pa_wo_pretrain.sh:94-122
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")

SUBST = {
    "feat_dim": "23", "max_chunk_size": "10000", "num_targets": "5139", "num_speakers": "5139",
    "num_senones": "3856", "am_node": "tdnn5.batchnorm",
}


def heredoc(path, which=0):
    """Return the body of the `which`-th `cat <<EOF > ...xconfig*` heredoc of a shell script."""
    text = open(os.path.join(REF, path)).read()
    bodies = re.findall(r"cat <<EOF > [^\n]*xconfig[^\n]*\n(.*?)\nEOF", text, flags=re.S)
    body = bodies[which]
    body = re.sub(r"\$\{?(\w+)\}?", lambda m: SUBST[m.group(1)], body)
    return body


def prepare_lib(tmp):
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    dst = os.path.join(tmp, "libs")
    shutil.copytree(os.path.join(REF, "egs/sre/v2/steps/libs"), dst)
    subprocess.check_call(["chmod", "-R", "u+w", tmp])
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", "libs"], cwd=tmp,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    up = os.path.join(dst, "nnet3/xconfig/utils.py")
    s = open(up).read()
    s = s.replace("len(positions) / 2", "len(positions) // 2").replace("len(fields) / 2", "len(fields) // 2")
    open(up, "w").write(s)
    sys.path.insert(0, tmp)


def final_config(xconfig_text, tmp, existing_nodes=None):
    import libs.nnet3.xconfig.parser as xp
    import libs.nnet3.xconfig.layers as xl
    path = os.path.join(tmp, "network.xconfig")
    open(path, "w").write(xconfig_text + "\n")
    existing = []
    for name, dim in (existing_nodes or []):
        existing.append(xl.XconfigExistingLayer("existing", {"name": name, "dim": dim}, existing))
    layers = xp.read_xconfig_file(path, existing)
    lines = []
    for layer in layers:
        for tag, line in layer.get_full_config():
            if tag == "final":
                lines.append(line.rstrip())
    return "\n".join(lines) + "\n"


def am_existing(senones=3856):
    nodes = [("input", 23)]
    for i in range(1, 5):
        for part in ("affine", "relu", "batchnorm"):
            nodes.append(("tdnn%d.%s" % (i, part), 650))
    for part in ("affine", "relu", "batchnorm"):
        nodes.append(("tdnn5.%s" % part, 128))
    nodes += [("output.affine", senones), ("output.log-softmax", senones), ("output", senones)]
    return nodes


def main():
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="xc_")
    try:
        prepare_lib(tmp)
        jobs = {
            "v2_xvector": (heredoc("egs/sre/v2/local/nnet3/xvector/run_xvector_new.sh"), None),
            "am": (heredoc("egs/sre/v5/local/nnet3_cvector/cvector/train_am.sh"), None),
            "v4_cvector": (heredoc("egs/sre/v4/local/nnet3_cvector/cvector/train_xvector_with_am.sh"), am_existing()),
            "v5_cvector": (heredoc("egs/sre/v5/local/nnet3_cvector/cvector/train_cvector_with_am.sh")
                           .replace("output_xvec", "output"), am_existing()),
            "v3_multitask": (heredoc("egs/sre/v3/local/nnet3_cvector/cvector/prepare_nnet3_xconfig.sh")
                             .replace("output_xvec", "output"), None),
            "v3_2share": (heredoc("egs/sre/v3/local/nnet3_cvector/cvector/prepare_nnet3_xconfig_2share.sh")
                          .replace("output_xvec", "output"), None),
            "v3_3share": (heredoc("egs/sre/v3/local/nnet3_cvector/cvector/prepare_nnet3_xconfig_3share.sh")
                          .replace("output_xvec", "output"), None),
            "v3_4share": (heredoc("egs/sre/v3/local/nnet3_cvector/cvector/prepare_nnet3_xconfig_4share.sh")
                          .replace("output_xvec", "output"), None),
            "pa_wo_pretrain": (heredoc("egs/sre/v4/local/nnet3/xvector/run_xvector_pa_wo_pretrain.sh"), None),
        }
        for name, (xc, existing) in jobs.items():
            text = final_config(xc, tmp, existing)
            open(os.path.join(OUT, name + ".config"), "w").write(text)
            print("%-16s %4d lines" % (name, text.count("\n")))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
