#!/usr/bin/env python3
"""bench.py - throughput of the x-vector / c-vector extraction hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one rank per
GPU with torch.distributed.run.  One "step" = one pass of the hot path (prep -> TDNN GEMMs -> pooling -> embedding
affine) over one batch of synthetic feature chunks per GPU; inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line.

  metric   utterance-embeddings/sec on 400-frame chunks (BASELINE.json), whole job over all N GPUs
  workload BASELINE.json configs[1]: v2 x-vector TDNN, batch = 256 chunks x 400 frames per GPU ("weak" scaling:
           per-GPU work is fixed, utterances are sharded, the only collective is ONE RCCL broadcast of the packed
           weights at start-up - SURVEY.md §8(e))
  dtype    the arithmetic the dominant kernel computed in.  Default `--precision default` = what every entry point of the
           library ships (XV_PREC_DEFAULT): the model is packed for fp16mx2 (fp16 MFMA + two block-scaled 4-bit products for the
           rounding residuals of weights and activations, 1.5 passes - what the command-line tools run when nothing else is
           said, a function of the model alone: `other_modes.fp16mx2`) and then MEASURED like a recipe's shared calibration
           file is (xv_ctx_calibrate = `nnet3-xvector-compute --calibration=<file>` on this table):
           64 chunks spread evenly over the workload, outside the timed region: the lighter fp16mx (1.25 passes)
           runs only if its worst embedding stays within 7.5e-5 of the three-pass fp16x3 result, which it does on this model
           (Kaldi's initialisation distribution, what BASELINE.json asks for) and does not on the heavy-tailed model of
           `parity_trained_like_model`, where the same policy keeps fp16mx2.  `config.calibration` holds what was measured and
           chosen; `other_modes` the other arithmetics on the same workload, each with its error.
  roofline dominant kernel = the GEMM instantiation with the largest share of the step (name as the engine's profile
           report observed it); achieved = algorithmic FLOPs of its launches / their duration, HIP events stamped by the
           dispatches on the launch stream over the same K steps repeated right after the timed region (the event
           bookkeeping costs ~4 % of a step, so it stays out of `value`); peak = 2.5 PFLOP/s dense fp16 MFMA
           (MI355X_MICROARCH.md).  "mfma_per_alg_mac" = MFMA issue time per algorithmic product in fp16-pass units.
  cpu_baseline  BASELINE.md section 3 "B0": oracle/xvec_cpu_baseline.c (C + OpenMP restatement of Kaldi's semantics -
           "port", NOT Kaldi) file in / file out, one thread and every host core; rank 0, N=1 only, ~12 s.
"""
import argparse
import importlib
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# precision mode -> MFMAs executed per algorithmic product in the frame-level GEMMs
PRECISION_NOTES = {"default": None, "bf16x3": 3, "fp16x3": 3, "fp16x2": 2, "fp16mx": 1.25, "fp16mx2": 1.5, "auto": None, "bf16": 1, "fp16": 1}
KERNEL_PASSES = {"bf16x3": 3, "fp16x3": 3, "fp16x2": 2, "fp16mx": 1.25, "fp16mx2": 1.5, "fp16x3e": 3, "bf16": 1, "fp16": 1}
PEAK_TFLOPS = 2500.0  # dense bf16/fp16 MFMA, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"



def arithmetic_name(ctx):
    """fast-chunk arithmetic of a context; "fp16mx2+lite<k>": k of its layers run the 1.25-pass arithmetic (xv_calibration.lite_mask)"""
    m = ctx.lite_mask
    return ctx.fast_mode + ("+lite%d" % bin(m).count("1") if m else "")

def cpu_baseline(topology, frames, seconds):
    """CPU baseline B0 of BASELINE.md section 3: oracle/xvec_cpu_baseline.c (this repo's C + OpenMP restatement of
    Kaldi's semantics - "port", NOT Kaldi, which is neither vendored by the reference nor installed), file in / file
    out, on the host cores of this box: once with one thread, once with every core (the reference spreads utterances
    over `nj` single-threaded processes, egs/sre/v2/run_sre10.sh:24,200).  The numpy/OpenBLAS figure of round 1 (one
    single-threaded oracle process per core) is kept beside it.  Bounded sample, started as child processes."""
    import shutil
    import subprocess
    import tempfile
    import numpy as np
    import helpers as H
    from oracle import kaldi_io as kio
    from oracle.export_program import export_program
    # cores this process may actually use: the affinity mask and the cgroup CPU quota, not the host's core count
    host = os.cpu_count() or 1
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else host
    cores, why = aff, ("every CPU of the host" if aff == host else
                       "sched_getaffinity: this process may run on %d of the host's %d CPUs" % (aff, host))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(per) + 0.5))
            if quota < cores:
                cores, why = quota, "cgroup cpu.max = %s %s: a quota of %d CPUs (affinity mask: %d, host: %d)" % (q, per, quota, aff, host)
    except Exception:   # noqa: BLE001 - no cgroup v2 quota file
        pass
    res = {"unit": "utt/s", "cores": cores, "host_cpus": host, "cores_limited_by": why, "kind": "port"}
    d = tempfile.mkdtemp(prefix="xvb0", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)   # archives (memory)
    dx = tempfile.mkdtemp(prefix="xvb0x")      # the host-built executable (/dev/shm may be mounted noexec)
    try:
        # candidates: the binary built in-tree (portable AVX2 + FMA) and, when a compiler is here, one built for this
        # host's ISA; the faster one on a short calibration run is used
        cands = []
        exe0 = os.path.join(ROOT, "oracle", "_build", "xvec_cpu_baseline")
        if os.path.exists(exe0):
            cands.append((exe0, "gcc -O3 -march=x86-64-v3 -fopenmp (prebuilt)"))
        native = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ARCH=native", "OUT=" + dx,
                                 dx + "/xvec_cpu_baseline"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if native.returncode == 0 and os.path.exists(dx + "/xvec_cpu_baseline"):
            cands.append((dx + "/xvec_cpu_baseline", "gcc -O3 -march=native -fopenmp, built on this host"))
        if not cands:
            raise RuntimeError("oracle/_build/xvec_cpu_baseline is missing (run __graft_entry__.build())")
        net, line = H.synth_model(topology)
        n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True))
        n2.apply_nnet_config(line)
        export_program(n2, d + "/prog.bin")
        pool = [H.features(2000 + i, frames) for i in range(16)]

        def run(n_utts, threads, exe=None):
            exe = exe or best[0]
            with open(d + "/f.ark", "wb") as f:
                for i in range(n_utts):
                    f.write(("u%06d " % i).encode() + b"\0B")
                    kio.write_matrix(f, pool[i % 16])
            r = subprocess.run([exe, d + "/prog.bin", d + "/f.ark", d + "/o.ark", str(threads)], stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE)
            m = re.search(r"(\d+) utterances, (\d+) frames, threads (\d+), read ([0-9.]+) s, compute ([0-9.]+) s, write ([0-9.]+) s",
                          r.stdout.decode())
            if r.returncode != 0 or not m:
                raise RuntimeError("xvec_cpu_baseline failed: " + r.stderr.decode()[-300:])
            n, comp = int(m.group(1)), float(m.group(5))
            total = float(m.group(4)) + comp + float(m.group(6))
            return n, comp, total
        # one thread: calibrate every candidate on 8 utterances, then a sample worth ~seconds/3 with the faster one
        best, per_core = None, 0.0
        for c in cands:
            n, comp, _ = run(8, 1, c[0])
            if n / comp > per_core:
                best, per_core = c, n / comp
        res["build"] = best[1]
        n1 = max(8, int(per_core * seconds / 3))
        n, comp, total = run(n1, 1)
        res["one_thread"] = {"value": n / comp, "utterances": n, "compute_s": comp, "file_to_file_s": total}
        # every core: calibrate on 2 utterances per thread (threads do not scale like cores on a shared host), then a
        # sample worth ~seconds/2 of wall
        n, comp, total = run(2 * cores, cores)
        nall = max(2 * cores, min(40000, int(n / comp * seconds / 2)))
        if nall > 3 * cores:
            n, comp, total = run(nall, cores)
        res["value"] = n / comp
        res["all_cores"] = {"value": n / comp, "threads": cores, "utterances": n, "compute_s": comp, "file_to_file_s": total,
                            "file_to_file_utt_s": n / total}
        res["sample"] = ("oracle/xvec_cpu_baseline (C, register-blocked fp32 GEMM, OpenMP over utterances; binary ark in, ark out): "
                         "%d x %d-frame utterances on %d threads in %.1f s of compute (value), %d on one thread in %.1f s"
                         % (n, frames, cores, comp, res["one_thread"]["utterances"], res["one_thread"]["compute_s"]))
    finally:
        shutil.rmtree(d, ignore_errors=True)
        shutil.rmtree(dx, ignore_errors=True)
    # round-1 figure for continuity: the numpy/OpenBLAS fp32 oracle, one single-threaded process per core
    try:
        worker = os.path.join(ROOT, "tools", "cpu_baseline_worker.py")
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        procs = [subprocess.Popen([sys.executable, worker, topology, str(frames), str(seconds / 3), str(i)], stdout=subprocess.PIPE,
                                  stderr=subprocess.DEVNULL, env=env) for i in range(cores)]
        n_tot, rate = 0, 0.0
        for p in procs:
            out = p.communicate()[0].decode().split()
            if len(out) == 2:
                n_tot += int(out[0])
                rate += int(out[0]) / float(out[1])
        res["numpy_oracle"] = {"value": rate, "processes": cores, "utterances": n_tot}
    except Exception as e:   # noqa: BLE001
        res["numpy_oracle"] = {"error": str(e)}
    if isinstance(res.get("numpy_oracle", {}).get("value"), float) and res["numpy_oracle"]["value"] > res.get("value", 0.0):
        res["note"] = ("the numpy/OpenBLAS oracle (one single-threaded process per core) is the faster CPU figure on this host; "
                       "value stays the C + OpenMP program BASELINE.md section 3 names as B0")
    return res


def layer_table(model):
    """{out node: (K, N, left, right, segment)} from the library's layer table (unpadded dims; left / right = frames not
    computable at the chunk edges)."""
    tab = {}
    for line in model.describe().splitlines()[1:]:
        parts = line.split()
        if len(parts) < 3 or "->" not in parts[2]:
            continue
        k, n = parts[2].split("->")
        m = re.search(r"ctx (\d+)/(\d+)", line)
        tab[parts[1]] = (int(k), int(n), int(m.group(1)) if m else 0, int(m.group(2)) if m else 0, "(segment)" in line)
    return tab


def gemm_groups(prof, tab, lens, ctx_pad):
    """Groups the profiled GEMM launches by the kernel instantiation that ran them.  Label format (engine.cc):
    'tdnn_gemm<epi>:<layer> <kernel name> [<kernel name of the second row region>]'.  Returns
    {kernel: {"ms": per step, "launches": per step, "flops": algorithmic FLOPs per step, "layers": [...]}}."""
    groups = {}
    for label, calls, ms in prof:
        if not label.startswith("tdnn_gemm<"):
            continue
        parts = label.split()
        layer = parts[0].split(":", 1)[1]
        kern = parts[1] if len(parts) > 1 else parts[0].split(":")[0]
        if len(parts) > 2:
            kern += "+" + parts[2]
        k, n, left, right, seg = tab.get(layer, (0, 0, 0, 0, False))
        frames = len(lens) if seg else float(sum(max(0, int(t) + ctx_pad - left - right) for t in lens))
        g = groups.setdefault(kern, {"ms": 0.0, "launches": 0, "flops": 0.0, "layers": []})
        g["ms"] += ms / max(1, calls)
        g["launches"] += 1
        g["flops"] += 2.0 * k * n * frames
        g["layers"].append(layer)
    return groups


def make_inputs(torch, ctx, lens, dev, seed):
    D = ctx.info.input_dim
    total_rows = int(lens.sum())
    g = torch.Generator(device=dev).manual_seed(seed)
    sigma = (8.0 * 0.9 ** torch.arange(D, dtype=torch.float32)).to(dev)
    feats = torch.randn(total_rows, D, generator=g, device=dev, dtype=torch.float32) * sigma
    import numpy as np
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    return feats, offs


def prewarm(torch, fn, seconds=0.6):
    """Keep the GPU busy for a moment before a timed region: after an idle gap (model set-up, the CPU oracle) the
    first ~100 ms of kernels run at a reduced clock, which once halved a measured rate.  Not counted as steps."""
    t = time.perf_counter()
    while time.perf_counter() - t < seconds:
        for _ in range(8):
            fn()
        torch.cuda.synchronize()


def time_steps(torch, fn, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def extra_config(torch, P, H, np, dev, local_rank, topology, precision, output_node, ragged, batch, steps, trained_seed=None):
    """One of the other BASELINE.json configurations, measured the same way as the headline (never `value`).
    trained_seed: the topology with heavy-tailed weights and calibrated BatchNorm statistics (helpers.trained_like_model) - what
    a trained Kaldi model looks like to the arithmetic, where the random-initialisation model of SURVEY.md section 8(d) is benign."""
    cfgs, _ = H.TOPOLOGIES[topology]
    if trained_seed is not None:
        net, line = H.trained_like_model(topology, trained_seed)
    elif output_node:
        net = H.nm.synthesize([H.config_text(c) for c in cfgs], seed=123, head_stddev=1.0)   # non-zero senone head
        line = "output-node name=output input=%s" % output_node
    else:
        net, line = H.synth_model(topology)
    model = P.Model(raw=net.to_bytes(True), nnet_config=line)
    ctx = P.Context(model, device=local_rank, precision=P.PRECISIONS[precision])
    cal = None
    if ragged:
        lens = np.random.default_rng(5).integers(ragged[0], ragged[1] + 1, batch).astype(np.int64)
    else:
        lens = np.full(batch, 400, dtype=np.int64)
    feats, offs = make_inputs(torch, ctx, lens, dev, 777)
    frame_level = not ctx.info.output_is_segment
    if precision == "default" and not frame_level:
        picks = list(range(batch)) if batch <= 64 else sorted({((2 * i + 1) * batch) // 128 for i in range(64)})
        fh = feats.cpu().numpy()
        sub = [fh[int(offs[k]):int(offs[k + 1])] for k in picks]
        cal = ctx.calibrate(np.concatenate(sub), np.concatenate([[0], np.cumsum([len(x) for x in sub])]).astype(np.int32), 7.5e-5)
    out = torch.empty(int(lens.sum()) if frame_level else batch, ctx.info.output_dim, dtype=torch.float32, device=dev)

    def step():
        ctx.forward_batch_device(feats.data_ptr(), offs, out.data_ptr(), out.shape[1], None)
    prewarm(torch, step, 0.3)
    dt = time_steps(torch, step, steps)
    mi = model.info
    ctx_pad = 0 if mi.output_is_segment else mi.left_context + mi.right_context
    macs = float(np.mean([model.macs(int(t) + ctx_pad) for t in lens]))
    n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True))
    n2.apply_nnet_config(line)
    ev = H.xo.GraphEvaluator(n2, np.float32)
    # chunks compared with the fp32 oracle: 64 spread evenly over the batch for a pooled output, two for a
    # frame-level one (each is 4096 columns x every frame)
    nchk = min(2 if frame_level else 64, batch)
    worst_x3 = None
    if frame_level:
        f_host = feats[:int(offs[nchk])].cpu().numpy()
        ref = np.concatenate([H.xo.compute_all_frames(ev, f_host[offs[i]:offs[i + 1]]) for i in range(nchk)])
        err = H.rel_err(out[:int(offs[nchk])].cpu().numpy(), ref)
        err_mean = None
    else:
        rows = sorted({(i * batch) // nchk for i in range(nchk)})
        f_host = feats.cpu().numpy()
        ref = np.stack([ev.compute(f_host[offs[i]:offs[i + 1]])[0] for i in rows])
        e_ = np.abs(out.cpu().numpy()[rows].astype(np.float64) - ref).max(axis=1) / np.abs(ref).max(axis=1)
        err, err_mean = float(e_.max()), float(e_.mean())
        # and EVERY chunk of the step against the three-pass arithmetic on the same inputs
        cx = P.Context(model, device=local_rank, precision=P.PRECISIONS["fp16x3"])
        ox = torch.empty_like(out)
        cx.forward_batch_device(feats.data_ptr(), offs, ox.data_ptr(), ox.shape[1], None)
        torch.cuda.synchronize()
        d_ = (out - ox).abs().amax(dim=1) / ox.abs().amax(dim=1)
        worst_x3 = {"value": float(d_.max()), "mean": float(d_.mean()), "chunks": int(d_.numel())}
        del cx, ox
    return {"workload": "%s%s, %s, %d chunks x %s frames, output %s" % (topology, " (trained-like weights, seed %d)" % trained_seed if trained_seed is not None else "",
                                                                     precision, batch,
                                                                     "%d-%d" % ragged if ragged else "400", output_node or "embedding"),
            "arithmetic": arithmetic_name(ctx), "calibration": cal,
            "value": batch * steps / dt, "unit": "utt/s", "frames_per_sec": float(lens.sum()) * steps / dt,
            "ms_per_step": dt / steps * 1e3, "alg_gflop_per_utt": 2.0 * macs / 1e9,
            "alg_tflops": 2.0 * macs * batch * steps / dt / 1e12, "rel_err_vs_oracle_fp32": err,
            "rel_err_vs_oracle_fp32_mean": err_mean, "parity_chunks_vs_oracle": nchk, "parity_worst_vs_fp16x3": worst_x3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="default", choices=sorted(PRECISION_NOTES),
                    help="default = what the command-line tools ship (XV_PREC_DEFAULT): the context is packed as fp16mx2 and, like a "
                         "recipe's shared calibration file (nnet3-xvector-compute --calibration), measured on 64 chunks spread over the workload "
                         "(xv_ctx_calibrate, outside the timed region): fp16mx if its error against fp16x3 is within 7.5e-5 on the worst chunk, else fp16mx2")
    ap.add_argument("--no-parity-sweep", action="store_true",
                    help="skip the every-chunk comparison against fp16x3 (profiling runs: the last forward pass of the process is then the timed workload's)")
    ap.add_argument("--no-calibrate", action="store_true", help="with --precision default: keep fp16mx2 whatever the model")
    ap.add_argument("--topology", default="v2_xvector")
    ap.add_argument("--batch", type=int, default=256, help="chunks per GPU per step")
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-modes", action="store_true", help="skip pipelined / sustained / other modes / other configs")
    ap.add_argument("--output-node", default=None,
                    help="compute this node instead of the topology's embedding node (e.g. output_am.log-softmax: the "
                         "senone head of BASELINE config 5 -> frame-level output, one row per frame)")
    ap.add_argument("--ragged", default=None, help="LO-HI: chunk lengths drawn uniformly from {LO..HI} (seed 5) instead of --frames")
    ap.add_argument("--lanes", type=int, default=2,
                    help="batches in flight inside the engine during the timed region (2 = the engine's default, what the command-line "
                         "tools run: consecutive steps alternate between two streams, so the tail kernels of one batch overlap the next "
                         "batch's; the per-kernel durations of the roofline always come from a one-lane context)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import helpers as H
    P = importlib.import_module("speaker-embedding-with-phonetic-information_amd")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a visible MI355X; there is no CPU path to measure")
    # BENCH_DIST_BACKEND=gloo + BENCH_FORCE_DEVICE=0 exist only to exercise the multi-rank code path on a 1-GPU box
    # (ranks then share one GPU, so the throughput of such a run means nothing)
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("BENCH_FORCE_DEVICE"):
        local_rank = int(os.environ["BENCH_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if backend == "nccl" else torch.device("cpu")    # where collective payloads live
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # ---- model: read/packed ONCE on rank 0, broadcast over RCCL -------------------------------------------
    prec = P.PRECISIONS[args.precision]
    net = cfg_line = None
    ragged = tuple(int(v) for v in args.ragged.split("-")) if args.ragged else None
    if rank == 0:
        net, cfg_line = H.synth_model(args.topology)
        if args.output_node:
            cfgs, _ = H.TOPOLOGIES[args.topology]
            net = H.nm.synthesize([H.config_text(c) for c in cfgs], seed=123, head_stddev=1.0)   # non-zero senone head
            cfg_line = "output-node name=output input=%s" % args.output_node
        model = P.Model(raw=net.to_bytes(True), nnet_config=cfg_line)
        blob = model.pack(prec)
        meta = torch.tensor([len(blob)], dtype=torch.int64, device=cdev)
    else:
        meta = torch.zeros(1, dtype=torch.int64, device=cdev)
    # bounded: a rank that never joins the broadcast becomes an error and exit status 3 after XVEC_BCAST_TIMEOUT (here 300 s unless
    # set: the other ranks wait in it while rank 0 synthesises and packs the model on a box whose page cache may be cold)
    with P.Watchdog("the broadcast of the packed weights (%d ranks, backend %s)" % (world, backend),
                    None if os.environ.get("XVEC_BCAST_TIMEOUT") else 300.0):
        if world > 1:
            dist.broadcast(meta, 0)
        nbytes = int(meta[0].item())
        if rank == 0:
            wt = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(cdev)
        else:
            wt = torch.empty(nbytes, dtype=torch.uint8, device=cdev)
        if world > 1:
            dist.broadcast(wt, 0)       # the ONE collective of this path (weights over xGMI)
            if wt.is_cuda:
                torch.cuda.synchronize()
    # The timed region runs the engine as the command-line tools run it: two lanes (XVEC_LANES=2, the engine's default) - two
    # consecutive steps are in flight on two streams, the small tail kernels of one batch overlap the next batch's GEMMs.  The
    # per-kernel HIP-event durations of the roofline come from a SECOND context with one lane (`ctx_prof`), run right after the
    # timed region on the same inputs: with two batches sharing the GPU a kernel's duration would include its neighbour's.
    def make_ctx(lanes):
        os.environ["XVEC_LANES"] = str(lanes)
        if wt.is_cuda:   # the broadcast buffer is used where it is: device to device, no host round trip
            torch.cuda.synchronize()
            return P.Context(device_blob=(wt.data_ptr(), wt.numel()), device=local_rank)
        return P.Context(blob=wt.numpy().tobytes(), device=local_rank)
    ctx = make_ctx(args.lanes)
    ctx_prof = make_ctx(1) if args.lanes != 1 else ctx
    os.environ["XVEC_LANES"] = str(args.lanes)
    del wt

    def calibrate_ctx(c, f_dev, o, what):
        """The CLI's policy: rank 0 measures on 64 chunks spread evenly over the workload (xv_calibrate_table's rule), every
        rank runs what it chose."""
        import numpy as _np
        nb = len(o) - 1
        n = min(64, nb)
        picks = list(range(nb)) if nb <= 64 else sorted({((2 * i + 1) * nb) // 128 for i in range(64)})
        choice = torch.tensor([-1, 0, 0], dtype=torch.int64, device=cdev)   # arithmetic, lite-layer mask (uint64) as two 32-bit halves
        cal = None
        if rank == 0:
            fh = f_dev.cpu().numpy()
            sub = [fh[int(o[k]):int(o[k + 1])] for k in picks]
            so = _np.concatenate([[0], _np.cumsum([len(x) for x in sub])]).astype(_np.int32)
            cal = c.calibrate(_np.concatenate(sub), so, 7.5e-5)
            cal["sample"] = "%d chunks spread evenly over the %d of %s" % (n, nb, what)
            choice[0] = P.PRECISIONS[cal["chosen"]]
            choice[1] = cal.get("lite_mask", 0) & 0xFFFFFFFF
            choice[2] = cal.get("lite_mask", 0) >> 32
        if world > 1:
            dist.broadcast(choice, 0)
            if rank != 0:
                c.set_fast_mode(P.PRECISION_NAMES[int(choice[0])])
                lite = int(choice[1]) | (int(choice[2]) << 32)
                if lite:
                    c.set_lite_mask(lite)
        return cal

    # ---- synthetic inputs resident in HBM (N(0,1)*sigma_d, sigma_d = 8*0.9^d) ---------------
    B, T = args.batch, args.frames
    if ragged:
        lens = np.random.default_rng(5 + rank).integers(ragged[0], ragged[1] + 1, B).astype(np.int64)
    else:
        lens = np.full(B, T, dtype=np.int64)
    total_rows = int(lens.sum())
    frame_level = not ctx.info.output_is_segment
    feats, offs = make_inputs(torch, ctx, lens, dev, 20180101 + rank)
    if os.environ.get("BENCH_ZERO_FEATS") == "1":   # diagnostic only (DVFS study): constant activations
        feats.zero_()
    calibration = None
    if args.precision == "default" and not args.no_calibrate and not frame_level:
        calibration = calibrate_ctx(ctx, feats, offs, "this workload")
    out_rows = total_rows if frame_level else B
    outs = [torch.empty(out_rows, ctx.info.output_dim, dtype=torch.float32, device=dev) for _ in range(2 if frame_level else 4)]
    out = outs[0]
    step_no = [0]

    def step():
        o = outs[step_no[0] % len(outs)]
        step_no[0] += 1
        ctx.forward_batch_device(feats.data_ptr(), offs, o.data_ptr(), o.shape[1], None)

    prewarm(torch, step)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # Per-kernel durations: the same K steps once more, now with a (start, stop) HIP event pair stamped by every dispatch
    # on the launch stream.  Kept out of the timed region above because the event bookkeeping itself costs ~4 % of a step
    # (18 events per step: the profiled steps run at `profiled_ms_per_step`); kernel durations are not affected by it.
    if ctx_prof is not ctx and ctx.info.output_is_segment and ctx_prof.fast_mode != ctx.fast_mode:
        ctx_prof.set_fast_mode(ctx.fast_mode)
    if ctx_prof is not ctx and ctx.lite_mask:
        ctx_prof.set_lite_mask(ctx.lite_mask)
    o_prof = torch.empty_like(outs[0])

    def step1():
        ctx_prof.forward_batch_device(feats.data_ptr(), offs, o_prof.data_ptr(), o_prof.shape[1], None)
    # one batch in flight, untimed warm-up then K steps without the event bookkeeping (`single_lane`), then K steps with it
    for _ in range(max(2, args.warmup)):
        step1()
    dt_one = time_steps(torch, step1, args.steps)
    ctx_prof.set_profiling(True)
    tp0 = time.perf_counter()
    for _ in range(args.steps):
        step1()
    torch.cuda.synchronize()
    dt_prof = time.perf_counter() - tp0
    ctx_prof.set_profiling(False)
    prof = ctx_prof.profile_report()
    tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())

    if rank == 0:
        value = world * B * args.steps / dt
        mi = model.info
        ctx_pad = 0 if mi.output_is_segment else mi.left_context + mi.right_context
        macs = float(np.mean([model.macs(int(t) + ctx_pad) for t in lens]))      # average per chunk
        # ---- roofline of the dominant kernel: the GEMM instantiation with the largest share of the step ----------
        groups = gemm_groups(prof, layer_table(model), lens, ctx_pad)
        dom = max(groups, key=lambda k: groups[k]["ms"]) if groups else None
        total_prof_ms = sum(ms / max(1, c) for (_, c, ms) in prof)
        roofline = {"bound": "mfma", "achieved": 0.0, "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": 0.0, "traffic": None}
        if dom:
            g = groups[dom]
            achieved = g["flops"] / (g["ms"] * 1e-3) / 1e12
            pname = re.search(r"<(\w+),", dom)
            passes = KERNEL_PASSES.get(pname.group(1) if pname else "", None)
            # HBM-side bytes per launch of that kernel: PMC counters cannot be read by the run that is being timed, so they come
            # from the last rocprofv3 --pmc passes over this same command (tools/profile_gpu.sh -> tools/parse_rocprof.py ->
            # profiles/pmc_traffic.json), STAMPED with the hash of the kernel sources they were collected on: counters of another
            # kernel build are refused (traffic = null) instead of going stale silently
            traffic, traffic_note = None, "profiles/pmc_traffic.json not found"
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    from parse_rocprof import kernels_sha16
                    pj = json.load(open(pmc))
                    have, want = pj.get("kernels_sha16"), kernels_sha16()
                    if have == want:
                        traffic = pj.get("hbm_bytes_per_launch", {}).get(dom)
                        traffic_note = ("profiles/pmc_traffic.json (%s): rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE per launch of this kernel, "
                                        "separate passes (MI355X_MICROARCH.md, HBM section), collected on this kernel build (kernels_sha16 %s); "
                                        "not measured by this run" % (pj.get("source"), want))
                    else:
                        traffic_note = ("profiles/pmc_traffic.json is stale: collected on kernels_sha16 %s, this tree is %s - re-run "
                                        "tools/profile_gpu.sh + tools/parse_rocprof.py" % (have, want))
                except Exception as e:   # noqa: BLE001
                    traffic_note = "profiles/pmc_traffic.json unreadable: %s" % e
            roofline = {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / PEAK_TFLOPS, "traffic": traffic,
                        "traffic_source": traffic_note,
                        "kernel": dom, "layers": g["layers"], "launches_per_step": g["launches"],
                        "avg_launch_ms": g["ms"] / g["launches"], "alg_flops_per_launch": g["flops"] / g["launches"],
                        "mfma_per_alg_mac": passes,
                        "mfma_executed_frac": (achieved * passes / PEAK_TFLOPS) if passes else None,
                        "whole_step_alg_tflops": 2.0 * macs * B * args.steps / dt / 1e12,
                        "all_kernels_ms_per_step": total_prof_ms, "profiled_ms_per_step": dt_prof / args.steps * 1e3,
                        "measured": "HIP events stamped by the dispatches on the launch stream, over %d steps run right "
                                    "after the timed region (same process, same inputs)" % args.steps,
                        "by_kernel": {k: {"ms_per_step": v["ms"], "launches_per_step": v["launches"], "layers": v["layers"],
                                          "alg_tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12,
                                          "frac": v["flops"] / (v["ms"] * 1e-3) / 1e12 / PEAK_TFLOPS}
                                      for k, v in groups.items()}}
        # ---- parity spot check against the oracle on the same inputs (not timed) ------------------------------
        n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True))
        n2.apply_nnet_config(cfg_line)
        ev = H.xo.GraphEvaluator(n2, np.float32)
        # pooled output: EVERY chunk of the step against the fp32 oracle (256 x 400 frames: ~3 s of numpy); frame-level: two chunks
        nchk = min(2, B) if frame_level else B
        f_host = feats[:int(offs[nchk])].cpu().numpy()
        parity_mean = None
        if frame_level:
            ref = np.concatenate([H.xo.compute_all_frames(ev, f_host[offs[i]:offs[i + 1]]) for i in range(nchk)])
            parity = H.rel_err(outs[0][:int(offs[nchk])].cpu().numpy(), ref)
        else:
            ref = np.stack([ev.compute(f_host[offs[i]:offs[i + 1]])[0] for i in range(nchk)])
            e_ = np.abs(out[:nchk].cpu().numpy().astype(np.float64) - ref).max(axis=1) / np.abs(ref).max(axis=1)
            parity, parity_mean = float(e_.max()), float(e_.mean())
        kernel_precs = sorted(set(m for k in groups for m in re.findall(r"<(\w+),", k)))
        res = {
            "metric": "utterance-embeddings/sec (400-frame chunks)", "value": value, "unit": "utt/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # the arithmetic the dominant kernel computes in (the kernel names carry the mode of every launch)
            "dtype": (re.search(r"<(\w+),", dom).group(1) if dom else args.precision),
            "data": "synthetic",
            "config": {"workload": "%s TDNN, %d chunks x %s frames per GPU per step, utterance-sharded, weights broadcast once over RCCL"
                                   % (args.topology, B, args.ragged or T), "topology": args.topology, "batch_chunks_per_gpu": B,
                       "frames_per_chunk": T if not ragged else None, "precision": args.precision,
                       "arithmetic": arithmetic_name(ctx), "calibration": calibration, "kernel_precisions": kernel_precs,
                       "lanes": args.lanes, "alg_gflop_per_utt": 2.0 * macs / 1e9, "output_node": args.output_node or "embedding",
                       "chunk_lengths": args.ragged or str(T), "frames_per_step": total_rows,
                       # which build of the library ran (it names the kernel sources it was compiled from: the hash profiles/pmc_traffic.json
                       # is stamped with)
                       "library": P.lib().xv_version().decode()},
            "frames_per_sec": world * total_rows * args.steps / dt,
            "roofline": roofline,
            "parity_rel_err_vs_oracle_fp32": parity,          # worst chunk
            "parity_rel_err_vs_oracle_fp32_mean": parity_mean,
            "parity_chunks_vs_oracle": nchk,
            "kernels_ms_per_step": {l: ms / max(1, c) for (l, c, ms) in prof},
            # the same K steps with ONE batch in flight (the context the per-kernel durations come from; rounds 1-4 reported this as
            # `value`): what the second lane's overlap of one batch's tail kernels with the next batch's GEMMs is worth
            "single_lane": {"lanes": 1, "value": B * args.steps / dt_one, "unit": "utt/s (this rank)", "ms_per_step": dt_one / args.steps * 1e3},
        }
        if not frame_level and PRECISION_NOTES[args.precision] != 1 and not args.no_parity_sweep:
            # every chunk of the step against the three-pass arithmetic on the same inputs: worst chunk of max|d| / max|ref|
            cx = P.Context(model, device=local_rank, precision=P.PRECISIONS["fp16x3"])
            ox = torch.empty_like(out)
            cx.forward_batch_device(feats.data_ptr(), offs, ox.data_ptr(), ox.shape[1], None)
            torch.cuda.synchronize()
            d_ = (out - ox).abs().amax(dim=1) / ox.abs().amax(dim=1)
            res["parity_worst_vs_fp16x3"] = {"value": float(d_.max()), "mean": float(d_.mean()), "chunks": int(d_.numel())}
            del cx, ox
        extras = world == 1 and not args.no_extra_modes
        if extras:
            # sustained: the same step for >= 3 s of wall clock (the timed region above is ~30 ms)
            n_sus = max(args.steps, int(3.2 / (dt / args.steps)))
            ds = time_steps(torch, step, n_sus)
            res["sustained"] = {"seconds": ds, "steps": n_sus, "value": B * n_sus / ds, "unit": "utt/s",
                                "ratio_to_value": (B * n_sus / ds) / value}
        if extras and PRECISION_NOTES[args.precision] != 1 and not frame_level and not ragged:
            extra = {}
            # the other arithmetic modes on the same workload, each with its measured error (never `value`)
            for pname in ("fp16x3", "fp16mx2", "auto", "fp16x2", "bf16", "fp16"):
                if pname == args.precision or (pname == "fp16mx2" and ctx.fast_mode == "fp16mx2") or \
                        (pname == "auto" and ctx.fast_mode == "fp16mx"):
                    continue
                c2 = P.Context(model, device=local_rank, precision=P.PRECISIONS[pname])
                o2 = torch.empty_like(out)
                f2 = lambda: c2.forward_batch_device(feats.data_ptr(), offs, o2.data_ptr(), o2.shape[1], None)  # noqa: E731
                prewarm(torch, f2, 0.3)
                d2 = time_steps(torch, f2, args.steps)
                extra[pname] = {"value": B * args.steps / d2, "unit": "utt/s",
                                "alg_tflops": 2.0 * macs * B * args.steps / d2 / 1e12,
                                "rel_err_vs_oracle_fp32": H.rel_err(o2[:nchk].cpu().numpy(), ref), "parity_chunks_vs_oracle": nchk}
                del c2
            res["other_modes"] = extra
            # the same arithmetic with tdnn_gemm_kernel_p8 dealing out WHOLE output tiles (XVEC_DEBUG=p8_whole=1, read per context): every
            # launch is longer on its own (no even split of the K tiles), but there is no exchange of partial tiles, and with two
            # batches in flight the other lane's kernels fill the CUs that finish early (profiles/r05_p8_whole_tiles.md)
            try:
                os.environ["XVEC_DEBUG"] = "p8_whole=1"
                cw = P.Context(model, device=local_rank, precision=prec)
                os.environ.pop("XVEC_DEBUG", None)
                if calibration:
                    cw.set_fast_mode(calibration["chosen"])
                    if calibration.get("lite_mask"):
                        cw.set_lite_mask(calibration["lite_mask"])
                ow = torch.empty_like(out)
                fw = lambda: cw.forward_batch_device(feats.data_ptr(), offs, ow.data_ptr(), ow.shape[1], None)  # noqa: E731
                prewarm(torch, fw, 0.3)
                dw = time_steps(torch, fw, args.steps)
                res["whole_tiles"] = {"value": B * args.steps / dw, "unit": "utt/s", "lanes": args.lanes, "ms_per_step": dw / args.steps * 1e3,
                                      "bit_identical_to_the_timed_run": bool(torch.equal(ow, out)),
                                      "note": "XVEC_DEBUG=p8_whole=1: not the default - the dominant kernel's own launch is ~13 % longer this way"}
                del cw
            except Exception as e:   # noqa: BLE001
                os.environ.pop("XVEC_DEBUG", None)
                res["whole_tiles"] = {"error": str(e)}
            # the fast modes on a model closer to a trained one (heavy-tailed weights, calibrated BatchNorm): see docstring
            try:
                tnet, tline = H.trained_like_model(args.topology, 11)
                tmodel = P.Model(raw=tnet.to_bytes(True), nnet_config=tline)
                tn2 = H.nm.Nnet3.from_bytes(tnet.to_bytes(True))
                tn2.apply_nnet_config(tline)
                tev = H.xo.GraphEvaluator(tn2, np.float64)
                tu = [H.features(50 + i, 400) for i in range(4)]
                tf, to = H.pack(tu)
                tref = np.stack([tev.compute(u)[0] for u in tu])
                ptl = {pn: H.rel_err(P.Context(tmodel, device=local_rank, precision=P.PRECISIONS[pn]).forward_batch(tf, to), tref)
                       for pn in ("fp16x3", "fp16mx2", "fp16x2", "auto")}
                if args.precision == "default":
                    # the shipped policy on this model: calibrate on its own utterances, then run what was chosen
                    tc = P.Context(tmodel, device=local_rank)
                    tcal = tc.calibrate(tf, to, 7.5e-5)
                    ptl["default"] = H.rel_err(tc.forward_batch(tf, to), tref)
                    ptl["default_chose"] = tcal
                else:
                    ptl[args.precision] = H.rel_err(P.Context(tmodel, device=local_rank, precision=prec).forward_batch(tf, to), tref)
                res["parity_trained_like_model"] = ptl
            except Exception as e:   # noqa: BLE001
                res["parity_trained_like_model"] = {"error": str(e)}
            # BASELINE.json configs 3 and 5 on this GPU
            oc = {}
            for key, kw in (("v5_cvector", dict(topology="v5_cvector", precision=args.precision, output_node=None, ragged=None)),
                            # the headline's network with heavy-tailed weights (what a trained model looks like to the arithmetic)
                            ("v2_trained_like", dict(topology="v2_xvector", precision=args.precision, output_node=None, ragged=None,
                                                     trained_seed=11)),
                            ("v3_senone_fp16_ragged", dict(topology="v3_multitask", precision="fp16",
                                                           output_node="output_am.log-softmax", ragged=(200, 600))),
                            # the same job in the parity-grade arithmetic nnet3-compute defaults to (config 5 names fp16)
                            ("v3_senone_fp16x3_ragged", dict(topology="v3_multitask", precision="fp16x3",
                                                             output_node="output_am.log-softmax", ragged=(200, 600)))):
                try:
                    oc[key] = extra_config(torch, P, H, np, dev, local_rank, batch=B, steps=max(5, args.steps // 3), **kw)
                except Exception as e:   # noqa: BLE001
                    oc[key] = {"error": str(e)}
            res["other_configs"] = oc
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(args.topology, int(np.mean(lens)), args.cpu_seconds)
            except Exception as e:   # noqa: BLE001
                res["cpu_baseline"] = {"error": str(e)}
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
