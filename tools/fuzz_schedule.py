#!/usr/bin/env python3
"""Schedule fuzzing of the GEMM kernels (GPU box).  A systematic hazard check next to the repeat-under-noise tools.

The kernels of csrc/kernels.hip order their LDS traffic by hand (counted waits, one or two barriers per phase, two wave
groups a barrier interval apart, a flag exchange between workgroups).  Every output element is accumulated in ONE fixed
order, so a launch's result does not depend on timing - unless a wait or a barrier is missing.  The fuzz build
(`-DXVEC_SCHED_FUZZ`, kernels.hip `sched_fuzz`) sleeps every wave for a pseudo-random time at every phase boundary,
differently in every launch; this tool runs the same forward passes on the product library and on the fuzz build and
compares the results BIT FOR BIT:

    tools/fuzz_schedule.py [--repeats R] [--quick] [--lib PATH] [--inject-lib PATH] [--smoke] [--no-forward] [--no-kernel-tests]

* reference: the product library (speaker-embedding-with-phonetic-information_amd/libxvec_hip.so), one forward per case;
* fuzzed: `--lib` (default <package>/fuzz/libxvec_hip.so, built by `make fuzz`), R forwards per case in one process;
* forward cases: topologies x arithmetic modes (incl. mixtures with lite layers) x kernel families (the 256 x 256 kernel,
  the 512 x 128 stream-K kernel, the per-tile kernels) x batch shapes (the bench shape, a ragged one, a small one);
* kernel launch tests: tests/test_gpu_kernels.py (single launches against the bit-exact emulations, shapes the engine does
  not dispatch included) run on the fuzz build;
* `--inject-lib`: a fuzz build with round 4's race put back on purpose (`make fuzz-inject`, -DXVEC_FUZZ_INJECT_HAZARD; the
  default path is used when it exists): its kernel launch tests must FAIL - that the method sees a hazard of this kind is
  part of what the tool reports.

Prints one line per case and a JSON summary; exit status 1 when a fuzzed result differs from the reference or a kernel
launch test fails on the fuzz build."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(out, topo, prec, lite, n, T, ragged, repeats):
    import helpers as H
    P = H.pkg()
    if ":" in topo:   # "config:node" - a frame-level output (nnet3-compute: senone log-posteriors, bottleneck features)
        cfg, node = topo.split(":")
        net = H.nm.synthesize([H.config_text(cfg)], seed=11, head_stddev=1.0)
        line = "output-node name=output input=%s" % node
    else:
        net, line = H.synth_model(topo)
    model = P.Model(raw=net.to_bytes(True), nnet_config=line)
    ctx = P.Context(model, precision=P.PRECISIONS[prec])
    if lite:
        ctx.set_lite_mask(lite)
    rng = np.random.default_rng(11)
    lens = rng.integers(max(T // 2, 30), T + T // 2, n) if ragged else np.full(n, T)
    pool = [H.features(2000 + i, int(t)) for i, t in enumerate(lens[:16])]
    utts = [pool[i % 16] for i in range(n)]
    feats, offs = H.pack(utts)
    res = [np.asarray(ctx.forward_batch(feats, offs)).copy() for _ in range(repeats)]
    np.save(out, np.stack(res))


def run_child(lib, env_extra, args, out):
    env = dict(os.environ)
    env.pop("XVEC_LIB", None)
    if lib:
        env["XVEC_LIB"] = lib
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", out] + [str(a) for a in args], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    if r.returncode != 0:
        raise RuntimeError("child failed (%s): %s" % (lib or "product", r.stderr[-1500:]))
    return np.load(out)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        out, topo, prec, lite, n, T, ragged, repeats = sys.argv[2:10]
        child(out, topo, prec, int(lite), int(n), int(T), int(ragged), int(repeats))
        return 0
    args = sys.argv[1:]
    repeats = int(args[args.index("--repeats") + 1]) if "--repeats" in args else 6
    quick = "--quick" in args
    import helpers as H
    pkg_dir = os.path.join(ROOT, H.PKG_NAME)
    lib = os.path.abspath(args[args.index("--lib") + 1]) if "--lib" in args else os.path.join(pkg_dir, "fuzz", "libxvec_hip.so")
    inject = os.path.abspath(args[args.index("--inject-lib") + 1]) if "--inject-lib" in args else os.path.join(pkg_dir, "fuzz_inject", "libxvec_hip.so")
    if not os.path.exists(inject):
        inject = None
    if not os.path.exists(lib):
        raise SystemExit("no fuzz build at %s (make -C speaker-embedding-with-phonetic-information_amd/csrc fuzz)" % lib)
    families = [("p8", {}), ("sk", {"XVEC_DEBUG": "p8=0"}), ("v2", {"XVEC_DEBUG": "gemm_variant=2"})]
    # (mode, lite mask): the mixtures run fp16mxe (a lite layer that still writes its residual plane) and fp16mx layers inside fp16mx2
    modes = [("fp16mx", 0), ("fp16mx2", 0), ("fp16mx2", 0b0101010), ("fp16mx2", 0b1111100), ("fp16x3", 0), ("fp16x2", 0), ("fp16", 0), ("bf16", 0)]
    shapes = [(256, 400, 0), (97, 300, 1), (5, 137, 0)]
    topos = ["v2_xvector", "v5_cvector"]
    if quick:
        modes = modes[:3] + modes[4:5]
        shapes = shapes[:2]
    if "--smoke" in args:   # what tests/test_gpu_fuzz.py runs inside the GPU suite: under a minute
        modes, shapes, topos, families = modes[:3], shapes[1:2], topos[:1], families[:2]
    tmp = tempfile.mkdtemp(prefix="xvfuzz")
    cases = bad = launches = 0
    report = []
    # ---- single launches against the emulations, on the fuzz build (and on the build with the race put back: must fail) ----
    kt = {}
    if "--no-kernel-tests" not in args:
        for tag, l in (("fuzz", lib), ("inject", inject)):
            if not l:
                continue
            env = dict(os.environ, XVEC_LIB=l)
            r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_kernels.py"), "-m", "gpu", "-q",
                                "-p", "no:cacheprovider"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1800)
            last = [x for x in r.stdout.strip().splitlines() if " passed" in x or " failed" in x][-1:]
            kt[tag] = {"rc": r.returncode, "summary": last[0].strip() if last else r.stdout[-300:],
                       "failed": [x.split(" ")[1] for x in r.stdout.splitlines() if x.startswith("FAILED ")]}
            print(json.dumps({"kernel_launch_tests": tag, **kt[tag]}), flush=True)
    if "--no-forward" in args:
        topos = []
    for topo in topos:
        for prec, lite in modes:
            for fam, env in families:
                for n, T, ragged in shapes:
                    a = [topo, prec, lite, n, T, ragged]
                    ref = run_child(None, env, a + [1], os.path.join(tmp, "ref.npy"))[0]
                    row = {"topology": topo, "mode": prec + ("+lite%x" % lite if lite else ""), "kernels": fam, "shape": [n, T, ragged]}
                    # (the race of the inject build sits in launches the engine does not dispatch: it is judged by the launch tests only)
                    for tag, l in (("fuzz", lib),):
                        got = run_child(l, env, a + [repeats], os.path.join(tmp, tag + ".npy"))
                        diff = [int(np.sum(g.view(np.uint32) != ref.view(np.uint32))) for g in got]
                        row[tag + "_differing_launches"] = sum(1 for d in diff if d)
                        row[tag + "_differing_values_max"] = max(diff)
                        if tag == "fuzz":
                            launches += repeats
                            bad += sum(1 for d in diff if d)
                    cases += 1
                    report.append(row)
                    print(json.dumps(row), flush=True)
    # frame-level outputs (config 5's senone head, the phonetic bottleneck): three-pass and single-pass fp16, fewer and shorter chunks
    # (one output row per frame: 3856 log-posteriors each)
    if "--no-forward" not in args and "--smoke" not in args:
        for topo in (["v3_multitask:output_am.log-softmax"] if quick else ["v3_multitask:output_am.log-softmax", "am:tdnn5.batchnorm"]):
            for prec in ("fp16x3", "fp16"):
                for fam, env in families:
                    for n, T, ragged in ((48, 400, 1), (3, 137, 0)):
                        a = [topo, prec, 0, n, T, ragged]
                        ref = run_child(None, env, a + [1], os.path.join(tmp, "ref.npy"))[0]
                        got = run_child(lib, env, a + [repeats], os.path.join(tmp, "fuzz.npy"))
                        diff = [int(np.sum(g.view(np.uint32) != ref.view(np.uint32))) for g in got]
                        row = {"topology": topo, "mode": prec, "kernels": fam, "shape": [n, T, ragged],
                               "fuzz_differing_launches": sum(1 for d in diff if d), "fuzz_differing_values_max": max(diff)}
                        launches += repeats
                        bad += row["fuzz_differing_launches"]
                        cases += 1
                        report.append(row)
                        print(json.dumps(row), flush=True)
    print(json.dumps({"cases": cases, "fuzzed_forward_passes": launches, "passes_that_differ_from_the_product_build": bad,
                      "kernel_launch_tests_on_the_fuzz_build": kt.get("fuzz", {}).get("summary"),
                      "kernel_launch_tests_with_the_race_put_back": kt.get("inject", {}).get("summary")}))
    return 1 if bad or kt.get("fuzz", {}).get("rc", 0) != 0 else 0


if __name__ == "__main__":
    sys.exit(main())
