#!/usr/bin/env python3
"""Multi-GPU embedding extraction: one process per GPU, utterances sharded, weights broadcast ONCE.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        speaker-embedding-with-phonetic-information_amd/dist_extract.py \
        --nnet "nnet3-copy --nnet-config=exp/xvectors/extract.config exp/nnet/final.raw - |" \
        --feats-scp data/sre10/feats.scp --feat-pipe "apply-cmvn-sliding ... scp:SCP ark:- | select-voiced-frames ... |" \
        --out-dir exp/xvectors --name sre10 --min-chunk-size 25 --chunk-size 10000

The reference parallelises extraction with `nj` independent processes that each re-read the model through their
own `nnet3-copy` pipe and get a contiguous slice of feats.scp (egs/sre/v2/sid/nnet3/xvector/
extract_xvectors_new.sh:59,72,91-93; utils/split_scp.pl:193-221).  This launcher is the MI355X-node form of the
same thing: rank 0 reads + lowers + packs the model once, ONE broadcast moves the packed weights
to the other ranks (RCCL over xGMI with --backend nccl; gloo in CPU tests), every rank takes the contiguous slice
split_scp.pl would give job rank+1 and writes xvector_<name>.<rank+1>.{ark,scp}; rank 0 concatenates the scp files
exactly like extract_xvectors_new.sh:99.  torch.distributed is plumbing only: all compute is libxvec_hip.so.
"""
import argparse
import importlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def shard_bounds(n, world):
    """Contiguous split of n items over `world` jobs, the first n % world jobs get one extra item
    (utils/split_scp.pl:208-217)."""
    base, extra = divmod(n, world)
    bounds, start = [], 0
    for r in range(world):
        size = base + (1 if r < extra else 0)
        bounds.append((start, start + size))
        start += size
    return bounds


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--nnet", required=True, help="raw nnet3 model rxfilename (file or 'command |')")
    ap.add_argument("--nnet-config", default=None)
    ap.add_argument("--output-node", default=None, help="e.g. tdnn6.affine (instead of an nnet3-copy pipe)")
    ap.add_argument("--feats-scp", required=True)
    ap.add_argument("--feat-pipe", default=None,
                    help="feature pipeline with the literal SCP standing for this rank's scp slice; default: scp:SCP")
    ap.add_argument("--out-dir", required=True)
    ap.add_argument("--name", default="xvector")
    ap.add_argument("--chunk-size", type=int, default=-1)
    ap.add_argument("--min-chunk-size", type=int, default=100)
    ap.add_argument("--pad-input", default="true")
    ap.add_argument("--precision", default="auto", choices=["bf16x3", "bf16", "fp16", "fp16x3", "fp16x2", "auto"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--dry-run", action="store_true", help="shard + broadcast only (no device; used by CPU tests)")
    args = ap.parse_args(argv)

    import numpy as np
    import torch
    import torch.distributed as dist
    P = importlib.import_module(os.path.basename(HERE))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = args.backend == "nccl"
    if use_cuda:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if use_cuda else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)

    # ---- model: read ONCE on rank 0, one broadcast of the packed image ------------------------------------------
    prec = P.PRECISIONS[args.precision]
    if rank == 0:
        cfg = open(args.nnet_config).read() if args.nnet_config else ""
        if args.output_node:
            cfg += "\noutput-node name=output input=%s\n" % args.output_node
        model = P.Model(rxfilename=args.nnet, nnet_config=cfg or None)
        blob = model.pack(prec)
        size = torch.tensor([len(blob)], dtype=torch.int64, device=dev)
    else:
        size = torch.zeros(1, dtype=torch.int64, device=dev)
    if world > 1:
        dist.broadcast(size, 0)
    if rank == 0:
        wt = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    else:
        wt = torch.empty(int(size.item()), dtype=torch.uint8, device=dev)
    if world > 1:
        dist.broadcast(wt, 0)
    blob = wt.cpu().numpy().tobytes()

    # ---- this rank's contiguous slice of the utterance list -----------------------------------------------------
    lines = [l for l in open(args.feats_scp) if l.strip()]
    lo, hi = shard_bounds(len(lines), world)[rank]
    os.makedirs(args.out_dir, exist_ok=True)
    job = rank + 1
    my_scp = os.path.join(args.out_dir, "feats_%s.%d.scp" % (args.name, job))
    with open(my_scp, "w") as f:
        f.writelines(lines[lo:hi])
    rspec = ("ark:" + args.feat_pipe.replace("SCP", my_scp)) if args.feat_pipe else ("scp:" + my_scp)
    ark = os.path.join(args.out_dir, "xvector_%s.%d.ark" % (args.name, job))
    scp = os.path.join(args.out_dir, "xvector_%s.%d.scp" % (args.name, job))
    done = failed = 0
    if args.dry_run:
        import hashlib
        print("rank %d/%d: blob %d bytes sha1 %s, utterances [%d, %d)" % (rank, world, len(blob),
                                                                      hashlib.sha1(blob).hexdigest(), lo, hi), flush=True)
        open(scp, "w").writelines("%s DRYRUN\n" % l.split()[0] for l in lines[lo:hi])
    elif hi > lo:
        ctx = P.Context(blob=blob, device=local_rank)
        done, failed = ctx.extract_table(rspec, "ark,scp:%s,%s" % (ark, scp), args.chunk_size, args.min_chunk_size,
                                         args.pad_input.lower() in ("true", "t", "1"))
    else:
        open(scp, "w").close()
    counts = torch.tensor([done, failed], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(counts)          # bookkeeping only (2 integers); no data-path collective exists
        dist.barrier()
    if rank == 0:
        with open(os.path.join(args.out_dir, "xvector_%s.scp" % args.name), "w") as out:
            for j in range(1, world + 1):
                out.write(open(os.path.join(args.out_dir, "xvector_%s.%d.scp" % (args.name, j))).read())
        print("Done %d utterances, failed for %d (over %d ranks)" % (int(counts[0]), int(counts[1]), world), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if (args.dry_run or int(counts[0]) > 0) else 1


if __name__ == "__main__":
    sys.exit(main())
