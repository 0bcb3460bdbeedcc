#!/usr/bin/env python3
"""bench.py — UE-subframe updates/s of the PRACH random-access hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched through
torch.distributed.run, one rank per GPU (RCCL).  One JSON line on rank 0.

  step      one pass of the hot path over one batch = ONE trial per GPU of BASELINE config 2:
            RandomAccessSimulatorBeta.c as committed — nUE=100 000, Beta(3,4) arrivals, 54 preambles,
            nGrantUL=54, backoff 20, retx limit 10 (maxMsg2TxCount=9), --times 1 — i.e. 10 000
            dependent subframes x 100 000 UEs = 1e9 UE-subframe updates.  Production RNG (Philox).
  value     (updates processed by all ranks in the K timed steps) / (max over ranks of the wall time
            of those K steps, barrier + device sync on both sides).  Trial inputs (parameter block,
            arrival table) are tiny and staged by the call; there is no host-resident data set.
  roofline  dominant kernel = trial_kernel; achieved = 32 B (5 int32 fields read + 3
            written per UE per subframe) x updates per launch / its mean duration, measured with HIP
            events on the engine's own stream (prach_last_timing).  The single-trial workload is
            latency-bound (1e4 dependent subframes), not HBM-bound: the fraction is reported as asked.
  cpu_baseline  the real reference binary (oracle/_ref, built from /root/reference in the build
            container) on one host core for a bounded sample of its own hard-coded sweep; plus
            `cpu_port`: the oracle's O(N)-per-subframe restatement on the full workload.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_UPDATE = 32.0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_reference_baseline(budget_s: float):
    """Time the REAL reference program (RandomAccessSimulatorBeta, single thread like the reference)
    for ~budget_s seconds of its hard-coded nUE sweep; its per-point Results.txt carries the cumulative
    clock() seconds (Beta.c:481) and the golden fixture the exit subframe of each point."""
    exe = os.path.join(ROOT, "oracle", "_ref", "RandomAccessSimulatorBeta")
    gold = os.path.join(ROOT, "tests", "golden", "beta.json")
    if not (os.path.exists(exe) and os.path.exists(gold)):
        return None
    g = json.load(open(gold))
    exit_time = {}
    for blk in g["stdout"].split("-------- ")[1:]:
        lines = blk.split("\n")
        n = int([l for l in lines if l.startswith("Number of UEs:")][0].split(":")[1])
        t = int([l for l in lines if l.startswith("Total simulation time:")][0].split(":")[1].replace("ms", ""))
        exit_time[n] = t
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "BasicBetaSimulationResults"))
        p = subprocess.Popen([exe], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        try:
            p.wait(timeout=budget_s)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
        done = []
        for n in sorted(exit_time):
            f = os.path.join(d, "BasicBetaSimulationResults", f"0_54_{n}_Results.txt")
            if os.path.exists(f):
                lines = open(f).read().split("\n")
                if len(lines) >= 6 and lines[5]:
                    done.append((n, float(lines[5])))
        if not done:
            return None
        updates = sum(n * min(10000, exit_time[n] + 1) for n, _ in done)
        secs = done[-1][1]
        return {"value": updates / secs, "unit": "UE-subframe updates/s", "cores": 1, "kind": "reference",
                "sample": "oracle/_ref/RandomAccessSimulatorBeta (reference as committed, 54 grants, seed 0): nUE points "
                          + ",".join(str(n) for n, _ in done) + f" of its own sweep, {secs:.1f} s of clock()"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nue", type=int, default=100000)
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of reference-CPU timing (rank 0, N=1 only)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl == RCCL over xGMI; gloo only to rehearse on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as g
    pkg = g.load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if args.same_device:
        local_rank = 0
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")  # where the collective's tensors live
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    eng = pkg.Engine(local_rank)
    # the workload: RandomAccessSimulatorBeta.c as committed (Beta.c:47-57), one trial per GPU
    def trial(seed):
        return pkg.make_cfg(args.nue, variant=pkg.VARIANT_BETA_C, rng_mode=pkg.RNG_PHILOX, seed=seed)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        eng.run_trials([trial(1000 + rank)])
    barrier()
    t0 = time.perf_counter()
    updates = 0
    kernel_ms = 0.0
    agg_succ = 0
    for k in range(args.steps):
        (res,), _ = eng.run_trials([trial(rank + world * k)])
        assert res.status == 0
        updates += args.nue * res.steps
        agg_succ += res.nSuccessUE
        kernel_ms += eng.timing().kernel_ms
    barrier()
    dt = time.perf_counter() - t0

    tot_updates, max_dt = updates, dt
    if dist is not None:
        # the one collective of the job: final aggregates (success counts, updates) summed over ranks (RCCL)
        t = torch.tensor([updates, agg_succ], dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        tm = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        tot_updates, agg_succ, max_dt = int(t[0]), int(t[1]), float(tm[0])

    if rank == 0:
        value = tot_updates / max_dt
        per_launch_updates = updates / args.steps
        k_ms = kernel_ms / args.steps
        achieved = ALGO_BYTES_PER_UPDATE * per_launch_updates / (k_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "UE-subframe updates/sec at nUE=100k Beta; bit-exact success-ratio vs ref",
            "value": value, "unit": "UE-subframe updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * max_dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"configs[1]: nUE={args.nue}, -d 2 (Beta), 54 preambles, maxRetx=10, --times 1 per GPU "
                                   f"(RandomAccessSimulatorBeta.c as committed: nGrantUL=54, backoff 20, 10 000 subframes)",
                       "rng": "philox4x32-10 (production mode)", "trials_per_step_per_gpu": 1,
                       "updates_per_step_per_gpu": per_launch_updates, "parallelism": f"trials sharded over {world} GPU(s), one RCCL sum all-reduce of the aggregates"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "prach::cluster_kernel (G=32 workgroups per trial)", "kernel_ms": k_ms,
                         "note": "single-trial workload: 1e4 dependent subframes over <=1.6 MB of L2-resident state; bounded by per-subframe latency, not HBM (DESIGN.md §5)"},
            "success_ratio_mean": agg_succ / (args.steps * world * args.nue),
        }
        if world == 1 and not args.no_extras:
            sys.path.insert(0, ROOT)
            extras = {}
            # (1) reference-bit-exact mode: the 100k point of the reference's own chained sweep (glibc rand stream)
            try:
                offs = json.load(open(os.path.join(ROOT, "tests", "golden", "stream_offsets.json")))
                gold = json.load(open(os.path.join(ROOT, "tests", "golden", "beta.json")))
                tr = [t_ for t_ in gold["trials"] if t_["nUE"] == 100000][0]
                cfg = pkg.make_cfg(100000, variant=pkg.VARIANT_BETA_C, rng_mode=pkg.RNG_GLIBC, seed=0,
                                   stream_offset=offs["beta"]["100000"])
                t1 = time.perf_counter()
                (r,), (logs,) = eng.run_trials([cfg], want_logs=True)
                wall = time.perf_counter() - t1
                import hashlib
                ok = (pkg.format_results(cfg, r, 0.0).decode()[:-8] == tr["results_text"]
                      and hashlib.sha256(pkg.format_logs(logs, 100000)).hexdigest() == tr["logs_sha256"])
                tm_ = eng.timing()
                extras["glibc_mode_reference_point"] = {
                    "bit_exact_vs_reference_files": bool(ok), "nSuccessUE": r.nSuccessUE, "success_ratio": r.nSuccessUE / 1e5,
                    "kernel_updates_per_s": 1e5 * r.steps / (tm_.kernel_ms * 1e-3),
                    "wall_updates_per_s_incl_host_stream_and_log_dump": 1e5 * r.steps / wall}
            except Exception as e:  # fixtures missing: report, do not fail the bench
                extras["glibc_mode_reference_point"] = {"error": repr(e)}
            # (2) the RandomAccessWithNOMA default (12 grants, overload) single trial
            cfg = pkg.make_cfg(args.nue, variant=pkg.VARIANT_WITHNOMA_C, rng_mode=pkg.RNG_PHILOX, seed=0)
            (r,), _ = eng.run_trials([cfg])
            extras["withnoma_g12_single_trial_kernel_updates_per_s"] = args.nue * r.steps / (eng.timing().kernel_ms * 1e-3)
            # (3) BASELINE config 3: nUE sweep 10k..100k x --times 100, all 1000 trials concurrently (one workgroup per
            #     trial): the regime where the state of the in-flight trials streams through HBM every subframe
            cfgs = [pkg.make_cfg(n, variant=pkg.VARIANT_WITHNOMA_C, rng_mode=pkg.RNG_PHILOX, seed=s)
                    for s in range(100) for n in range(10000, 100001, 10000)]
            rs, _ = eng.run_trials(cfgs)
            upd = sum(c.nUE * r_.steps for c, r_ in zip(cfgs, rs))
            kms = eng.timing().kernel_ms
            extras["config3_sweep_x100_1000_trials"] = {
                "kernel_updates_per_s": upd / (kms * 1e-3), "kernel_ms": kms, "updates": upd,
                "roofline": {"bound": "hbm", "achieved": 32.0 * upd / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": 32.0 * upd / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "note": "algorithmic bytes of the reference's dense formulation (32 B per UE-subframe update, SURVEY 8d); "
                                     "the kernel skips finished / not yet arrived groups, keeps 8+4 B hot records and does not rewrite a UE "
                                     "in steady contention, so the algorithmic rate can exceed the HBM peak; PMC traffic of this "
                                     "launch: profiles/r01f_summary.md"},
                "mean_success_ratio_100k": sum(r_.nSuccessUE for c, r_ in zip(cfgs, rs) if c.nUE == 100000) / 100 / 1e5}
            # (4) BASELINE config 4: NOMA.c power-level grouping, nUE=100 000, one trial
            cfg = pkg.make_cfg(args.nue, variant=pkg.VARIANT_NOMA_C, rng_mode=pkg.RNG_PHILOX, seed=0)
            (r,), _ = eng.run_trials([cfg])
            extras["noma_c_single_trial"] = {"kernel_updates_per_s": args.nue * r.steps / (eng.timing().kernel_ms * 1e-3),
                                             "nSuccessUE": r.nSuccessUE, "upload_ms_activation_table": eng.timing().upload_ms}
            out["extras"] = extras
        if world == 1 and not args.no_cpu:
            from oracle import binding as ob
            # parity of THIS bench's workload against the oracle + the O(N)/subframe CPU port, 1 core
            ocfg = ob.make_cfg(args.nue, variant=ob.VARIANT_BETA_C)
            t1 = time.perf_counter()
            ores, _ = ob.run_trial(ocfg, ob.Rng(ob.RNG_PHILOX, 0), want_ues=False)
            osec = time.perf_counter() - t1
            (r0,), _ = eng.run_trials([trial(0)])
            out["parity"] = {"vs": "oracle (pinned to the compiled reference)", "bit_exact":
                             (r0.nSuccessUE, r0.time_exit, r0.collisionPreambles, r0.totalPreambleTxop, r0.sumTimer, r0.draws)
                             == (ores.nSuccessUE, ores.time_exit, ores.collisionPreambles, ores.totalPreambleTxop, ores.sumTimer, ores.draws),
                             "nSuccessUE": r0.nSuccessUE}
            out["cpu_port"] = {"value": args.nue * ores.steps / osec, "unit": "UE-subframe updates/s", "cores": 1, "kind": "port",
                               "sample": f"oracle O(N)/subframe restatement, the full workload (nUE={args.nue}, {ores.steps} subframes), {osec:.1f} s"}
            ref = cpu_reference_baseline(args.cpu_budget)
            out["cpu_baseline"] = ref if ref is not None else out["cpu_port"]
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
