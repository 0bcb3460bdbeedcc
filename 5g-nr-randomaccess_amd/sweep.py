"""BASELINE config 5: the `--times` x nUE sweep sharded over the GPUs of one node.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \\
        5g-nr-randomaccess_amd/sweep.py --times 1000 --program beta --out results_dir

One process per GPU.  Philox trials are independent: they are dealt to ranks by descending cost
(`dist.shard_trials`), each rank runs its shard in ONE `prach_run_trials` call (one workgroup cluster per
trial), then ONE sum all-reduce (RCCL over xGMI; payload < 1 KB) merges the per-nUE aggregates and the
per-trial rows are gathered to rank 0, which writes `results.csv` with AveragePerformance.py's arithmetic
(sum of the per-seed 2-decimal values in seed order — not recoverable from the summed raw aggregates).
The reference runs the same grid serially (RandomAccessWithNOMA.c:216-221).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--times", type=int, default=100)
    ap.add_argument("--program", choices=("beta", "withnoma"), default="beta")
    ap.add_argument("--sweep", default="10000:100000:10000")
    ap.add_argument("--out", default=".")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--same-device", action="store_true", help="rehearsal on one GPU: every rank uses cuda:0")
    args = ap.parse_args(argv)

    import torch
    import __graft_entry__ as g
    pkg = g.load_package()
    distmod = importlib.import_module(pkg.__name__ + ".dist")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    lo, hi, step = map(int, args.sweep.split(":"))
    points = list(range(lo, hi + 1, step))
    variant = pkg.VARIANT_BETA_C if args.program == "beta" else pkg.VARIANT_WITHNOMA_C
    cfgs = [pkg.make_cfg(n, variant=variant, rng_mode=pkg.RNG_PHILOX, seed=s) for s in range(args.times) for n in points]
    mine = distmod.shard_trials(cfgs, rank, world)

    eng = pkg.Engine(local_rank)
    t0 = time.perf_counter()
    res = []
    CH = 1024  # trials per call: bounds the device arena (2.7 GB per 1024 trials of the sweep)
    for a in range(0, len(mine), CH):
        r, _ = eng.run_trials([cfgs[i] for i in mine[a:a + CH]])
        res.extend(r)
    dt = time.perf_counter() - t0
    agg = distmod.aggregate_rows([cfgs[i] for i in mine], res, points)
    tot = distmod.allreduce_aggregates(agg, device=dev if (world > 1 and args.backend == "nccl") else None)
    # per-trial Results.txt texts (Beta.c's six lines; latency 0) travel to rank 0 for the exact results.csv — only where they are
    # consumed: AveragePerformance.py reads the Beta.c program's files
    allrows = None
    if variant == pkg.VARIANT_BETA_C:
        rows = [(i, pkg.format_results(cfgs[i], r, 0.0).decode()) for i, r in zip(mine, res)]
        allrows = distmod.gather_trial_rows(rows, dst=0, device=dev if (world > 1 and args.backend == "nccl") else None)
    if rank == 0:
        fi = {n: k for k, n in enumerate(distmod.AGG_FIELDS)}
        summary = {"program": args.program, "times": args.times, "points": points, "world": world,
                   "updates": int(tot[:, fi["updates"]].sum()), "rank0_seconds": dt,
                   "success_ratio": {str(p): float(tot[k, fi["nSuccessUE"]]) / (args.times * p) for k, p in enumerate(points)}}
        if variant == pkg.VARIANT_BETA_C:
            by = dict(allrows)
            per_point = [[by[s * len(points) + k] for s in range(args.times)] for k in range(len(points))]
            os.makedirs(args.out, exist_ok=True)
            with open(os.path.join(args.out, "results.csv"), "wb") as f:
                f.write(pkg.results_csv(per_point))
            summary["results_csv"] = os.path.join(args.out, "results.csv")
        print(json.dumps(summary), flush=True)
    eng.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
