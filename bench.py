#!/usr/bin/env python3
"""bench.py — UE-subframe updates/s of the PRACH random-access hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched through
torch.distributed.run, one rank per GPU (RCCL).  One JSON line on rank 0.

  N = 1   step = ONE trial of BASELINE configs[1]: RandomAccessSimulatorBeta.c as committed — nUE=100 000,
          Beta(3,4) arrivals, 54 preambles, nGrantUL=54, backoff 20, retx limit 10 (maxMsg2TxCount=9),
          --times 1 — i.e. 10 000 dependent subframes x 100 000 UEs = 1e9 UE-subframe updates, production RNG
          (Philox).  The dominant kernel is prach::lcluster_kernel (a cluster of workgroups per trial, UE
          state resident in LDS); `value` = updates / wall time of the K timed calls.
  N > 1   step = BASELINE configs[4]: the `--times T` x nUE sweep (10 points, 10k..100k) grid of the Beta.c
          program, dealt to the ranks by descending cost (dist.shard_trials: STRONG scaling, the total work
          does not depend on N), each rank runs its shard in ONE call (one launch), then ONE sum
          all-reduce of the int64 aggregate block (RCCL) and the gather of the per-trial rows to rank 0
          (results.csv needs them: AveragePerformance.py:10-24).  `value` = total updates / max over ranks
          of the step time, collective and gather INCLUDED.  The reference runs this grid serially
          (RandomAccessWithNOMA.c:216-221).  `--workload grid` runs the same grid on one GPU.
  roofline  achieved = 32 B (5 int32 fields read + 3 written per UE per subframe in the
          reference's dense formulation) x updates per launch / the kernel's mean duration, measured with
          HIP events on the engine's own stream (prach_last_timing) — an ALGORITHMIC rate: a single trial is
          10 000 dependent subframes over LDS/L2-resident state, bounded by per-subframe latency, not by HBM
          (measured HBM traffic: `traffic_from_profile`, a separate rocprofv3 --pmc pass, profiles/).
  cpu_baseline  the real reference program AT THE METRIC'S OWN SIZE, timed in this run: oracle/_ref/RandomAccessSimulatorBeta_100k
          (RandomAccessSimulatorBeta.c compiled with its hard-coded sweep started at nUE = 100 000, built from /root/reference in the
          build container) is started as a child process on one host core right behind the timed region, runs while the extras and the
          other CPU legs do, and is joined at the end (about 67 s on a GPU box);
          `cpu_reference_small_points`: the unmodified binary on the first points (10k..) of its own sweep for --cpu-budget seconds;
          `cpu_port`: the oracle's O(N)-per-subframe restatement on the full N=1 workload, one core;
          `cpu_port_all_cores`: the same restatement on 100 000-UE trials fanned over all host cores.
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
KERNEL_NAMES = {0: "prach::cluster_kernel (records in global memory)", 1: "prach::cluster_kernel (8+4 B records, one workgroup per trial)",
                2: "prach::cluster_kernel (LDS-resident records)", 3: "prach::lcluster_kernel (LDS-resident UE state)",
                4: "prach::batch_kernel (one workgroup per trial: the 32-byte records travel with the events, in 2 KB chunks)"}


def own_bytes(tm):
    """The bytes the one-workgroup-per-trial kernel itself asks for, from its own counters (prach_timing.group_visits / .event_ues).
    batch_kernel (round 4: the state travels with the event): an event UE's 32-byte record is read from the chunk of the subframe's event
    list it sits in and written into the chunk of the subframe of its next event (whole chunks are read: event_ues counts 64 per chunk),
    64 B; a contention window costs one 4-byte join-list entry written and one read (group_visits counts the joins), 8 B.  The general
    kernel's 8 + 4 byte form: 8 B per lane and 64-UE visit, ~40 B per event."""
    if tm.rec_mode == 4:
        return tm.event_ues * 64 + tm.group_visits * 8
    return tm.group_visits * 512 + tm.event_ues * 40


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
                          + ",".join(str(n) for n, _ in done) + f" of its own hard-coded sweep, {secs:.1f} s of clock(); the program cannot "
                          "be started at nUE=100 000 — its O(N^2) collision scan makes that point slower per update",
                "at_nUE_100000": None}  # (filled in by the caller: cpu_reference_at_100k)


REF100K_RECORD = os.path.join(ROOT, "profiles", "r03_cpu_reference_100k.json")


def start_reference_100k():
    """The reference's CPU path AT THE METRIC'S OWN SIZE: oracle/_ref/RandomAccessSimulatorBeta_100k is RandomAccessSimulatorBeta.c
    compiled with its hard-coded sweep started at nUE = 100 000 (oracle/Makefile SED_BETA_100K: one token of line 71), i.e. the
    reference's own 100 000-UE point from srand(0) (Beta.c:71,111-183,205-206).  Started as a CHILD process (one core, like the
    reference) and joined later: about 67 s on a GPU box's host, during which this process runs its extras and other CPU legs."""
    exe = os.path.join(ROOT, "oracle", "_ref", "RandomAccessSimulatorBeta_100k")
    if not os.path.exists(exe):
        return None
    d = tempfile.mkdtemp(prefix="prach_ref100k_")
    os.makedirs(os.path.join(d, "BasicBetaSimulationResults"))
    p = subprocess.Popen([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    return {"p": p, "dir": d, "t0": time.perf_counter()}


def join_reference_100k(h, timeout_s: float):
    """Wait for the child of start_reference_100k and turn its Results.txt (cumulative clock() seconds, Beta.c:205-206,481) and its
    'Total simulation time' line into a cpu_baseline record; None when it did not finish (it is killed then)."""
    import shutil
    if h is None:
        return None
    p = h["p"]
    try:
        so, _ = p.communicate(timeout=max(1.0, timeout_s))
    except subprocess.TimeoutExpired:
        p.kill()
        p.communicate()
        shutil.rmtree(h["dir"], ignore_errors=True)
        return None
    wall = time.perf_counter() - h["t0"]
    try:
        lines = open(os.path.join(h["dir"], "BasicBetaSimulationResults", "0_54_100000_Results.txt")).read().split("\n")
        tsim = [int(l.split(":")[1].replace("ms", "")) for l in so.split("\n") if l.startswith("Total simulation time:")][0]
        steps = min(10000, tsim + 1)
        secs = float(lines[5])
        rec = {"value": 100000.0 * steps / secs, "unit": "UE-subframe updates/s", "cores": 1, "kind": "reference",
               "sample": f"oracle/_ref/RandomAccessSimulatorBeta_100k: the reference's own nUE = 100 000 point from srand(0) (54 preambles, nGrantUL 54, "
                         f"retx 10), all {steps} subframes, {secs:.1f} s of clock() on one core ({wall:.1f} s wall, as a child process beside this "
                         f"run's extras and CPU legs), success ratio {lines[1]} %",
               "measured_in_this_run": True, "host": os.uname().nodename, "host_cores": host_cores()}
    except Exception as e:  # the child wrote nothing usable
        rec = None
        print(f"bench.py: reference 100k child: {e!r}", file=sys.stderr)
    shutil.rmtree(h["dir"], ignore_errors=True)
    return rec


def reference_100k_record():
    """(only when the child could not run: the record of an earlier run on a GPU box, marked as not measured here)"""
    if os.path.exists(REF100K_RECORD):
        rec = json.load(open(REF100K_RECORD))
        rec["measured_in_this_run"] = False
        rec["source"] = "profiles/r03_cpu_reference_100k.json: `python bench.py --cpu-full` on a GPU box in round 3"
        return rec
    return None


def cpu_port_all_cores(ob, nue: int, ntrials: int):
    """The oracle's O(N)/subframe restatement on `ntrials` trials of the N=1 workload's size, fanned over all host cores
    (plain C behind ctypes: the GIL is released)."""
    from concurrent.futures import ThreadPoolExecutor
    cores = host_cores()

    def one(seed):
        r, _ = ob.run_trial(ob.make_cfg(nue, variant=ob.VARIANT_BETA_C), ob.Rng(ob.RNG_PHILOX, seed), want_ues=False)
        return nue * r.steps

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        upd = sum(ex.map(one, range(ntrials)))
    dt = time.perf_counter() - t0
    return {"value": upd / dt, "unit": "UE-subframe updates/s", "cores": cores, "kind": "port",
            "sample": f"oracle O(N)/subframe restatement, {ntrials} trials of nUE={nue} (Beta.c program) on {cores} host threads, {dt:.1f} s"}


def host_cores() -> int:
    """Host threads this process may use (the GPU box gives one GPU's share of the host, not all of it)."""
    try:
        return max(1, min(len(os.sched_getaffinity(0)), 32))
    except AttributeError:
        return max(1, min(os.cpu_count() or 1, 32))


def traffic_from_profile():
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tf):
        return None
    try:
        return json.load(open(tf))
    except Exception:
        return None


def checked_traffic(prof, rec_mode: int, k_ms: float):
    """roofline.traffic = counter-measured HBM bytes per launch (FETCH_SIZE x the calibrated 2 + WRITE_SIZE, separate rocprofv3 --pmc
    passes of this same command: profiles/traffic.json) — carried only while the profiled kernel IS this run's kernel and its mean
    duration under the profiler is within 5 % of this run's; otherwise null and the reason."""
    if prof is None:
        return None, "no profiles/traffic.json"
    name = prof.get("kernel", "")
    mine = "lcluster_kernel" if rec_mode == 3 else "batch_kernel" if rec_mode == 4 else "cluster_kernel"
    if ("prach::" + mine) not in name:
        return None, f"profiles/traffic.json was taken on `{name}`, this run's kernel is prach::{mine}: not carried"
    prof_ms = float(prof.get("avg_kernel_ns", 0.0)) * 1e-6
    if k_ms <= 0 or abs(prof_ms - k_ms) / k_ms > 0.05:
        return None, f"profiles/traffic.json: kernel mean {prof_ms:.2f} ms under the profiler vs {k_ms:.2f} ms in this run (> 5 %): stale, not carried"
    return float(prof["hbm_bytes_per_launch"]), (f"profiles/traffic.json ({prof.get('tag', '?')}): {prof.get('method', '')}; kernel `{name}`, "
                                                  f"{prof_ms:.2f} ms under the profiler vs {k_ms:.2f} ms here")


def batch_traffic(workload: str, rec_mode: int, updates: float, kernel_ms: float):
    """Counter-measured HBM bytes of a batch_kernel launch, scaled from the profiled launch of the same regime (profiles/traffic_batch.json: bytes per UE-subframe
    update of the 1000-trial grid / of config 3) to this launch's updates — carried only while the kernel is prach::batch_kernel and its update rate is within
    15 % of the profiled launch's (a launch of another size has another tail, hence the wider band than for the single-trial line); else null and the reason."""
    tf = os.path.join(ROOT, "profiles", "traffic_batch.json")
    if not os.path.exists(tf):
        return None, "no profiles/traffic_batch.json"
    prof = json.load(open(tf))
    w = prof.get("workloads", {}).get(workload)
    if rec_mode != 4 or w is None or "batch_kernel" not in prof.get("kernel", ""):
        return None, f"profiles/traffic_batch.json holds prach::batch_kernel's {list(prof.get('workloads', {}))}: not this launch"
    rate = updates / (kernel_ms * 1e-3)
    if abs(rate / w["kernel_updates_per_s"] - 1.0) > 0.15:
        return None, f"profiles/traffic_batch.json: {w['kernel_updates_per_s']:.3e} updates/s under the profiler vs {rate:.3e} here (> 15 %): stale, not carried"
    return w["hbm_bytes_per_update"] * updates, (f"profiles/traffic_batch.json ({prof.get('tag', '?')}, {workload}): {w['hbm_bytes_per_update']:.3f} counter bytes per update "
                                                  f"x this launch's updates; {prof.get('method', '')}")


def relaunch_under_torchrun(args) -> int:
    """`python bench.py --gpus N` (N > 1) without a torch.distributed rendezvous in the environment: run the same command line as
    N ranks in a child process.  Returns the child's exit code; a node with fewer than N GPUs is an error, not a smaller run."""
    import socket
    import torch  # (torch.cuda.device_count() does not initialise the GPU on this image; nothing else is called)
    if not args.same_device and torch.cuda.device_count() < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but this node has {torch.cuda.device_count()} visible GPU(s)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(p.stdout)
    sys.stdout.flush()
    if p.returncode == 0 and not any(l.startswith("{") and '"n_gpus": %d' % args.gpus in l for l in p.stdout.splitlines()):
        print(f"bench.py: the {args.gpus}-rank child printed no JSON line with n_gpus = {args.gpus}", file=sys.stderr)
        return 3
    return p.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nue", type=int, default=100000)
    ap.add_argument("--times", type=int, default=1000, help="N>1 (or --workload grid): seeds of the --times x sweep grid (BASELINE configs[4]: 1000)")
    ap.add_argument("--workload", choices=("auto", "single", "grid"), default="auto", help="auto: single trial at N=1, the sharded grid at N>1")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of reference-CPU timing (rank 0, N=1 only)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-full", action="store_true", help="(kept for old command lines: the reference's own nUE = 100 000 point now runs by default, as a child process)")
    ap.add_argument("--cpu-ref-timeout", type=float, default=420.0, help="seconds after which the reference's 100 000-UE child process is given up (67 s on a GPU box, 265 s in the build container)")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl == RCCL over xGMI; gloo only to rehearse on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain `python bench.py --gpus N`: nothing here has touched the GPU yet, so start the N ranks as a fresh
        # child (one process per GPU under torch.distributed.run), relay its one JSON line and leave with its exit code.  It never
        # degrades to a one-GPU run.
        sys.exit(relaunch_under_torchrun(args))
    import torch
    import __graft_entry__ as g
    pkg = g.load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` or under "
                         f"`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    if not args.same_device and torch.cuda.device_count() < args.gpus:  # (counting devices does not initialise the GPU)
        raise SystemExit(f"bench.py: --gpus {args.gpus} but this node has {torch.cuda.device_count()} visible GPU(s)")
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
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    eng = pkg.Engine(local_rank)
    grid = args.workload == "grid" or (args.workload == "auto" and world > 1)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if grid:
        out = run_grid(args, pkg, eng, torch, dist, rank, world, dev, cdev, barrier)
    else:
        out = run_single(args, pkg, eng, torch, dist, rank, world, cdev, barrier)
    if rank == 0:
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_grid(args, pkg, eng, torch, dist, rank, world, dev, cdev, barrier):
    """BASELINE configs[4]: --times x nUE sweep sharded over the ranks; one all-reduce + one row gather per step."""
    import importlib
    import numpy as np
    distmod = importlib.import_module(pkg.__name__ + ".dist")
    points = list(range(10000, 100001, 10000))
    cfgs = [pkg.make_cfg(n, variant=pkg.VARIANT_BETA_C, rng_mode=pkg.RNG_PHILOX, seed=s) for s in range(args.times) for n in points]
    mine = distmod.shard_trials(cfgs, rank, world)
    my_cfgs = [cfgs[i] for i in mine]
    CH = 16384  # trials per call: the whole shard in one launch (10 000 trials of the sweep = 27 GB of the 288 GB; fewer launch tails)

    def step():
        t0 = time.perf_counter()
        res, kms, own = [], 0.0, 0
        for a in range(0, len(my_cfgs), CH):
            r, _ = eng.run_trials(my_cfgs[a:a + CH])
            res.extend(r)
            tmc = eng.timing()
            kms += tmc.kernel_ms
            own += own_bytes(tmc)
        t_sim = time.perf_counter() - t0
        agg = distmod.aggregate_rows(my_cfgs, res, points)  # raises if a trial did not return PRACH_OK
        tot = distmod.allreduce_aggregates(agg, device=cdev if (dist is not None and args.backend == "nccl") else None)
        rows = [(i, pkg.format_results(cfgs[i], r, 0.0).decode()) for i, r in zip(mine, res)]
        allrows = distmod.gather_trial_rows(rows, dst=0, device=cdev if (dist is not None and args.backend == "nccl") else None)
        return tot, allrows, t_sim, kms, time.perf_counter() - t0, own

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    sims, kmss, steps_t, owns = [], [], [], []
    tot = allrows = None
    for _ in range(args.steps):
        tot, allrows, t_sim, kms, t_step, own = step()
        sims.append(t_sim); kmss.append(kms); steps_t.append(t_step); owns.append(own)
    barrier()
    dt = time.perf_counter() - t0
    fi = {n: k for k, n in enumerate(distmod.AGG_FIELDS)}
    max_dt, per_rank = dt, [sum(sims) / args.steps]
    if dist is not None:
        tm = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        max_dt = float(tm[0])
        pr = torch.zeros(world, dtype=torch.float64, device=cdev)
        pr[rank] = sum(sims) / args.steps
        dist.all_reduce(pr, op=dist.ReduceOp.SUM)
        per_rank = [float(x) for x in pr.cpu()]
    if rank != 0:
        return None
    updates_per_step = int(tot[:, fi["updates"]].sum())
    by = dict(allrows)
    per_point = [[by[s * len(points) + k] for s in range(args.times)] for k in range(len(points))]
    csv_bytes = pkg.results_csv(per_point)
    value = updates_per_step * args.steps / max_dt
    mean_rank = sum(per_rank) / len(per_rank)
    tmr = eng.timing()
    g_traffic, g_note = batch_traffic("grid", tmr.rec_mode, updates_per_step / world, sum(kmss) / args.steps)
    return {
        "metric": "UE-subframe updates/sec at nUE=100k Beta; bit-exact success-ratio vs ref",
        "value": value, "unit": "UE-subframe updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * max_dt / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic",
        "config": {"workload": f"configs[4]: --times {args.times} x nUE sweep 10k..100k (10 points) = {len(cfgs)} Beta.c trials (54 preambles, "
                               f"nGrantUL=54, retx 10), sharded over {world} GPU(s) by descending cost; one RCCL sum all-reduce of the int64 "
                               f"aggregates + per-trial row gather for results.csv per step",
                   "rng": "philox4x32-10 (production mode)", "trials_per_step": len(cfgs), "updates_per_step": updates_per_step,
                   "parallelism": f"trials sharded over {world} GPU(s), no data-path collective", "collective_backend": args.backend if world > 1 else None},
        "per_rank_sim_seconds": per_rank, "imbalance": (max(per_rank) / mean_rank - 1.0) if mean_rank > 0 else 0.0,
        "rank0_step_seconds": {"simulation": sum(sims) / args.steps, "whole_step_incl_allreduce_and_gather": sum(steps_t) / args.steps},
        # rank 0's kernels.  In this regime (one workgroup per trial, thousands in flight) the kernel skips finished / not yet arrived 64-UE
        # groups, reads ONE 4-byte pass word per visited UE and touches a UE's 32-byte record only on its events, so the 32 B per update of
        # the reference's dense formulation is not what it moves: the fraction is built from the kernel's OWN bytes (counted by the kernel:
        # own_bytes()) and the dense-formulation rate is carried beside it, labelled.
        "roofline": {"bound": "hbm", "achieved": sum(owns) / (sum(kmss) * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": sum(owns) / (sum(kmss) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": g_traffic, "traffic_unit": "HBM bytes per step on rank 0 (FETCH_SIZE x 2.000 + WRITE_SIZE)", "traffic_provenance": g_note,
                     "kernel": "prach::batch_kernel (event lists of 32-byte records in 2 KB chunks, one workgroup per trial)", "kernel_ms_rank0": sum(kmss) / args.steps,
                     "bytes": "the kernel's own bytes: 64 B per event UE (its 32-byte record streamed in and out, whole 64-record chunks) + 8 B per contention window (join-list entry), rank 0",
                     "own_bytes_per_update": sum(owns) / max(1, updates_per_step * args.steps / world),
                     "dense_formulation_GBps_32B_per_update": ALGO_BYTES_PER_UPDATE * updates_per_step / world / (sum(kmss) / args.steps * 1e-3) / 1e9,
                     "note": "the launch is bound by instruction issue (0.58-0.66 of the SIMDs' vector issue slots over the whole launch, profiles/r04_grid.md section 5), "
                             "not by HBM (counter traffic 0.78 B per update = 0.95 x the kernel's own bytes, 1.9 TB/s); the 32 B per update of the reference's dense formulation is not a lower bound for a "
                             "kernel that only touches a UE at its events (dense_formulation_GBps is carried for the record: several times the chip's peak)"},
        # The N = 1 line of the driver's scaling run is the single-trial workload (configs[1]), a DIFFERENT workload: the one-GPU figure of this
        # grid regime travels in that same N = 1 line as extras.grid_one_gpu (measured in that run, --times 100), and every N > 1 line carries
        # value_per_gpu, so that a scaling efficiency can be formed from measured records only (no constant is pasted in here).
        "value_per_gpu": value / world,
        "one_gpu_reference": "extras.grid_one_gpu.kernel_updates_per_s / .wall_updates_per_s of the N = 1 record of the same run (or: python bench.py --gpus 1 --workload grid)",
        "success_ratio": {str(p): float(tot[k, fi["nSuccessUE"]]) / (args.times * p) for k, p in enumerate(points)},
        "results_csv_sha256": __import__("hashlib").sha256(csv_bytes).hexdigest(),
    }


def run_single(args, pkg, eng, torch, dist, rank, world, cdev, barrier):
    # the workload: RandomAccessSimulatorBeta.c as committed (Beta.c:47-57), one trial per GPU
    def trial(seed):
        return pkg.make_cfg(args.nue, variant=pkg.VARIANT_BETA_C, rng_mode=pkg.RNG_PHILOX, seed=seed)

    for w in range(args.warmup):
        eng.run_trials([trial(1000 + rank)])
    barrier()
    t0 = time.perf_counter()
    updates = 0
    kernel_ms = 0.0
    agg_succ = 0
    fallbacks = 0
    tm = None
    for k in range(args.steps):
        (res,), _ = eng.run_trials([trial(rank + world * k)])
        assert res.status == 0
        updates += args.nue * res.steps
        agg_succ += res.nSuccessUE
        tm = eng.timing()
        kernel_ms += tm.kernel_ms
        fallbacks += tm.fallback_trials
    barrier()
    dt = time.perf_counter() - t0
    assert fallbacks == 0, "a timed trial was rerun on the fallback kernel"
    # the reference's own 100 000-UE point: a child process on one host core, started behind the timed region, joined at the very end
    ref100k = start_reference_100k() if (world == 1 and rank == 0 and not args.no_cpu) else None

    tot_updates, max_dt = updates, dt
    if dist is not None:
        t = torch.tensor([updates, agg_succ], dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        tmx = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmx, op=dist.ReduceOp.MAX)
        tot_updates, agg_succ, max_dt = int(t[0]), int(t[1]), float(tmx[0])
    if rank != 0:
        return None

    value = tot_updates / max_dt
    per_launch_updates = updates / args.steps
    k_ms = kernel_ms / args.steps
    achieved = ALGO_BYTES_PER_UPDATE * per_launch_updates / (k_ms * 1e-3) / 1e9
    prof = traffic_from_profile()
    traffic, traffic_note = checked_traffic(prof, tm.rec_mode, k_ms)
    out = {
        "metric": "UE-subframe updates/sec at nUE=100k Beta; bit-exact success-ratio vs ref",
        "value": value, "unit": "UE-subframe updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * max_dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic",
        "config": {"workload": f"configs[1]: nUE={args.nue}, -d 2 (Beta), 54 preambles, maxRetx=10, --times 1 per GPU "
                               f"(RandomAccessSimulatorBeta.c as committed: nGrantUL=54, backoff 20, 10 000 subframes)",
                   "rng": "philox4x32-10 (production mode)", "trials_per_step_per_gpu": 1,
                   "updates_per_step_per_gpu": per_launch_updates, "parallelism": f"one trial per GPU x {world} GPU(s)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_unit": "HBM bytes per launch (FETCH_SIZE x 2.000 + WRITE_SIZE)", "traffic_provenance": traffic_note,
                     "traffic_GBps": (traffic / (k_ms * 1e-3) / 1e9) if traffic else None, "algorithmic_bytes_per_launch": ALGO_BYTES_PER_UPDATE * per_launch_updates,
                     "kernel": f"{KERNEL_NAMES.get(tm.rec_mode, '?')}, {tm.cluster_size} workgroups per trial", "kernel_ms": k_ms,
                     "us_per_subframe": 1e3 * k_ms / (per_launch_updates / args.nue),
                     "note": "achieved = ALGORITHMIC bytes (32 B per UE-subframe update, SURVEY 8d) / kernel time.  The single-trial workload is 1e4 dependent "
                             "subframes over LDS-resident state: bounded by the per-subframe latency chain (one cross-CU exchange + dependent phases), not by "
                             "HBM; `traffic` is this kernel's counter-measured HBM traffic per launch (separate rocprofv3 --pmc passes, profiles/): far BELOW the algorithmic bytes "
                             "because the UE state lives in LDS for the whole trial"},
        "success_ratio_mean": agg_succ / (args.steps * world * args.nue),
        "host_cores": {"os_cpu_count": os.cpu_count(), "usable_by_this_process": host_cores()},
    }
    if world == 1 and not args.no_extras:
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
        # (1b) the reference's stream with many seeds in flight: what `prach_sim -t 100` issues for the sweep's last point (the ten points of a seed are
        # chained through its rand() stream, so a call holds one point of every seed): 100 seeds x nUE = 100 000, Beta.c, one workgroup per trial
        cfgs = [pkg.make_cfg(args.nue, variant=pkg.VARIANT_BETA_C, rng_mode=pkg.RNG_GLIBC, seed=s) for s in range(100)]
        eng.run_trials([pkg.make_cfg(args.nue, variant=pkg.VARIANT_BETA_C, rng_mode=pkg.RNG_GLIBC, seed=s, max_steps=5) for s in range(100)])  # (the call's arena — a one-time hipMalloc of several GB — and the code object)
        t1 = time.perf_counter()
        rs, _ = eng.run_trials(cfgs)
        wall = time.perf_counter() - t1
        tm_ = eng.timing()
        upd = sum(c.nUE * r_.steps for c, r_ in zip(cfgs, rs))
        extras["glibc_mode_100_seeds_one_sweep_point"] = {"kernel": KERNEL_NAMES.get(tm_.rec_mode, "?") + " in the reference's rand() stream", "trials": len(cfgs),
                                                          "kernel_updates_per_s": upd / (tm_.kernel_ms * 1e-3), "wall_updates_per_s": upd / wall, "kernel_ms": tm_.kernel_ms,
                                                          "bad": sum(r_.status != 0 for r_ in rs), "fallback_trials": tm_.fallback_trials}
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
        tm3 = eng.timing()
        kms = tm3.kernel_ms
        own = own_bytes(tm3)  # the kernel's OWN bytes
        c3_traffic, c3_note = batch_traffic("config3", tm3.rec_mode, upd, kms)
        extras["config3_sweep_x100_1000_trials"] = {
            "kernel_updates_per_s": upd / (kms * 1e-3), "kernel_ms": kms, "updates": upd,
            "algorithmic_GBps_32B_per_update": 32.0 * upd / (kms * 1e-3) / 1e9,
            "own_traffic": {"group_visits": tm3.group_visits, "event_ues": tm3.event_ues, "own_bytes": own, "own_bytes_per_update": own / upd,
                            "own_GBps": own / (kms * 1e-3) / 1e9, "frac_of_hbm_peak": own / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "counter_traffic_bytes": c3_traffic, "counter_traffic_provenance": c3_note},
            "kernel": KERNEL_NAMES.get(tm3.rec_mode, "?"),
            "note": "32 B per update are the algorithmic bytes of the reference's dense formulation; the kernel touches a UE only at its events (its 32-byte record "
                    "streamed in and out of the event lists), so this is NOT an HBM fraction: profiles/r04_grid.md",
            "mean_success_ratio_100k": sum(r_.nSuccessUE for c, r_ in zip(cfgs, rs) if c.nUE == 100000) / 100 / 1e5}
        # (3b) the sharded grid's regime on ONE GPU (the denominator of the N > 1 lines' scaling): the Beta.c program's sweep x --times 100,
        #      1000 trials in one call
        cfgs = [pkg.make_cfg(n, variant=pkg.VARIANT_BETA_C, rng_mode=pkg.RNG_PHILOX, seed=s) for s in range(100) for n in range(10000, 100001, 10000)]
        t1 = time.perf_counter()
        rs, _ = eng.run_trials(cfgs)
        wall = time.perf_counter() - t1
        upd = sum(c.nUE * r_.steps for c, r_ in zip(cfgs, rs))
        tmg = eng.timing()
        gg_traffic, gg_note = batch_traffic("grid", tmg.rec_mode, upd, tmg.kernel_ms)
        extras["grid_one_gpu"] = {"workload": "configs[4]'s grid with --times 100: 1000 Beta.c trials (nUE 10k..100k), one call, one launch", "kernel": KERNEL_NAMES.get(tmg.rec_mode, "?"),
                                  "own_bytes": own_bytes(tmg), "counter_traffic_bytes": gg_traffic, "counter_traffic_provenance": gg_note,
                                  "kernel_updates_per_s": upd / (tmg.kernel_ms * 1e-3), "wall_updates_per_s": upd / wall, "kernel_ms": tmg.kernel_ms,
                                  "updates": upd, "trials": len(cfgs), "bad": sum(r_.status != 0 for r_ in rs), "fallback_trials": tmg.fallback_trials}
        # (4) BASELINE config 4: NOMA.c power-level grouping, nUE=100 000, one trial.  kernel_ms covers noma_activation_kernel (activeUE, NOMA.c:131-192, on the
        # device), the host's recomputation of the UEs it flagged (noma_host_ues) and noma_kernel
        cfg = pkg.make_cfg(args.nue, variant=pkg.VARIANT_NOMA_C, rng_mode=pkg.RNG_PHILOX, seed=0)
        (r,), _ = eng.run_trials([cfg])
        tmn = eng.timing()
        def noma_roofline(upd, ue_slots, kms, tm_):
            # prach::noma_kernel loads and stores a UE's 16-byte record ONCE per 5 ms access slot (NOMA.c:665-711 runs the grouping per slot
            # and the four subframes behind it on registers), so its own bytes are per UE-SLOT, not per update: 16 B in + 16 B out.
            own = 32.0 * ue_slots
            return {"bound": "hbm", "achieved": own / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": own / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                    "kernel": f"prach::noma_kernel, {tm_.cluster_size} workgroup(s) per trial", "kernel_ms": kms,
                    "bytes": "the kernel's own bytes: 32 B per UE and 5 ms access slot (record in + out once per slot)", "own_bytes": own, "own_bytes_per_update": own / upd,
                    "dense_formulation_GBps_32B_per_update": ALGO_BYTES_PER_UPDATE * upd / (kms * 1e-3) / 1e9,
                    "note": "counter-measured HBM traffic of both launches: profiles/r03_noma.md (a trial's state stays in L2: 50 MB of HBM traffic per "
                            "100 000-UE trial, 0.7 TB/s for the batched experiment) — the kernel is bound by one cross-CU exchange + the per-sector resolve per slot, not by bytes"}
        slots = args.nue * ((r.steps + 4) // 5)
        extras["noma_c_single_trial"] = {"kernel_updates_per_s": args.nue * r.steps / (tmn.kernel_ms * 1e-3),
                                         "wall_updates_per_s": args.nue * r.steps / (tmn.total_ms * 1e-3),
                                         "nSuccessUE": r.nSuccessUE, "activation": "device (noma_activation_kernel, inside kernel_ms)", "ues_recomputed_on_host": tmn.noma_host_ues,
                                         "roofline": noma_roofline(args.nue * r.steps, slots, tmn.kernel_ms, tmn)}
        # (4b) NOMA.c's OWN experiment (NOMA.c:637-719: 10 seeds x the ten sweep points), all 100 trials in one call
        cfgs = [pkg.make_cfg(n, variant=pkg.VARIANT_NOMA_C, rng_mode=pkg.RNG_PHILOX, seed=s) for s in range(10) for n in range(10000, 100001, 10000)]
        t1 = time.perf_counter()
        rs, _ = eng.run_trials(cfgs)
        wall = time.perf_counter() - t1
        tmb = eng.timing()
        upd = sum(c.nUE * r_.steps for c, r_ in zip(cfgs, rs))
        slots = sum(c.nUE * ((r_.steps + 4) // 5) for c, r_ in zip(cfgs, rs))
        extras["noma_c_experiment_batched"] = {"trials": len(cfgs), "kernel_updates_per_s": upd / (tmb.kernel_ms * 1e-3), "wall_updates_per_s": upd / wall,
                                               "activation": "device (noma_activation_kernel, inside kernel_ms)", "ues_recomputed_on_host": tmb.noma_host_ues, "updates": upd, "bad": sum(r_.status != 0 for r_ in rs), "roofline": noma_roofline(upd, slots, tmb.kernel_ms, tmb)}
        # (4c) NOMA.c in ITS OWN rand() stream (the parity mode of config 4): the ten seeds of the sweep's last point, one launch (noma_glibc_trial_kernel)
        cfgs = [pkg.make_cfg(args.nue, variant=pkg.VARIANT_NOMA_C, rng_mode=pkg.RNG_GLIBC, seed=s) for s in range(10)]
        t1 = time.perf_counter()
        rs, _ = eng.run_trials(cfgs)
        wall = time.perf_counter() - t1
        tmg2 = eng.timing()
        upd = sum(c.nUE * r_.steps for c, r_ in zip(cfgs, rs))
        extras["noma_c_reference_stream_10_seeds"] = {"trials": len(cfgs), "kernel_updates_per_s": upd / (tmg2.kernel_ms * 1e-3), "wall_updates_per_s": upd / wall, "kernel_ms": tmg2.kernel_ms,
                                                      "launches": tmg2.launches, "rerun_with_host_activation": tmg2.fallback_trials, "bad": sum(r_.status != 0 for r_ in rs),
                                                      "nSuccessUE_seed0": rs[0].nSuccessUE, "reference_nSuccessUE_seed0_at_its_own_stream_offset": "tests/golden/noma_c.json (the chained run is test_gpu_noma_glibc_reproduces_reference_lines)"}
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
        ref_small = cpu_reference_baseline(args.cpu_budget)  # (one more single-threaded child beside the 100k one: the host has the cores)
        at100k = join_reference_100k(ref100k, args.cpu_ref_timeout - (time.perf_counter() - ref100k["t0"])) if ref100k is not None else None
        out["cpu_port_all_cores"] = cpu_port_all_cores(ob, args.nue, 2 * host_cores())  # (behind the join: it takes every core)
        if ref_small is not None:
            ref_small.pop("at_nUE_100000", None)
            out["cpu_reference_small_points"] = ref_small
        if at100k is not None:
            out["cpu_baseline"] = at100k
        else:  # no binary on this box, or it did not finish: say so, carry what there is
            fb = ref_small if ref_small is not None else dict(out["cpu_port"])
            fb["at_nUE_100000"] = reference_100k_record()
            fb["note"] = "oracle/_ref/RandomAccessSimulatorBeta_100k was not available or did not finish within --cpu-ref-timeout: this is NOT the nUE = 100 000 figure"
            out["cpu_baseline"] = fb
    return out


if __name__ == "__main__":
    main()
