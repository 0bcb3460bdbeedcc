"""N>1 path on CPU: world_size-2 gloo run of the trial sharding + the single aggregate all-reduce +
the row gather (the same code bench.py / the sweep driver use over RCCL).  The oracle stands in for
the GPU engine here (tests may do that; the product never does)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    from oracle import binding as ob
    pkg = g.load_package()
    import importlib
    distmod = importlib.import_module("nr_randomaccess_amd.dist")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    points = [600, 900, 1200]
    cfgs = [pkg.make_cfg(n, variant=1, rng_mode=pkg.RNG_PHILOX, seed=s) for s in range(3) for n in points]
    mine = distmod.shard_trials(cfgs, rank, world)
    res = []
    for i in mine:
        c = cfgs[i]
        r, _ = ob.run_trial(ob.make_cfg(c.nUE, variant=1), ob.Rng(ob.RNG_PHILOX, int(c.seed)), want_ues=False)
        res.append(r)
    agg = distmod.aggregate_rows([cfgs[i] for i in mine], res, points)
    tot = distmod.allreduce_aggregates(agg)
    rows = distmod.gather_trial_rows([(i, res[k].nSuccessUE) for k, i in enumerate(mine)], dst=0)
    if rank == 0:
        q.put((mine, tot.tolist(), sorted(rows)))
    else:
        q.put((mine, None, None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_sharded_sweep(ob, pkg, world):
    """world 2, and world 8 — the size of the node the driver's scaling run uses (nine trials over eight ranks: one rank runs two)."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shards = [o[0] for o in outs]
    assert sorted(i for sh in shards for i in sh) == list(range(9)) and all(len(sh) >= 1 for sh in shards)
    tot = next(o[1] for o in outs if o[1] is not None)
    rows = next(o[2] for o in outs if o[2] is not None)
    # single-process ground truth
    points = [600, 900, 1200]
    exp = np.zeros((3, 10), dtype=np.int64)
    exp_rows = []
    k = 0
    for s in range(3):
        for j, n in enumerate(points):
            r, _ = ob.run_trial(ob.make_cfg(n, variant=1), ob.Rng(ob.RNG_PHILOX, s), want_ues=False)
            exp[j] += np.array([1, r.nSuccessUE, r.preambleTxCount, r.sumTimer, r.collisionPreambles, r.totalPreambleTxop,
                                r.continueFaliedUEs, r.finalSuccessUEs, r.steps, n * r.steps], dtype=np.int64)
            exp_rows.append((k, r.nSuccessUE))
            k += 1
    assert tot == exp.tolist()
    assert rows == exp_rows


def _rows_worker(rank, world, port, q, mode):
    """gather_trial_rows with text rows: RandomAccessWithNOMA's eight-line Results.txt (the longest row the drivers produce) and, in
    mode "too_long", one row on rank 1 only that does not fit — every rank must raise, none may hang in the payload collective."""
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    from oracle import binding as ob
    g.load_package()
    import importlib
    distmod = importlib.import_module("nr_randomaccess_amd.dist")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ocfg = ob.make_cfg(1500, variant=1)
    r, _ = ob.run_trial(ocfg, ob.Rng(ob.RNG_PHILOX, rank), want_ues=False)
    r.nSuccessUE, r.totalPreambleTxop, r.continueFaliedUEs = 99999, 2147483647, 2147483647  # (the widest numbers the format can print)
    text = ob.format_results(ocfg, r).decode() + "0.000000"
    rows = [(10 * rank + k, text) for k in range(2 + rank)]
    if mode == "too_long" and rank == 1:
        rows.append((99, "x" * (distmod.ROW_PAYLOAD + 1)))
    try:
        out = distmod.gather_trial_rows(rows, dst=0)
        q.put((rank, "ok", out, text))
    except ValueError as e:
        q.put((rank, "ValueError", str(e), text))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["withnoma_text", "too_long"])
def test_two_rank_gloo_text_rows(ob, pkg, mode):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rows_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if mode == "too_long":
        assert [o[1] for o in outs] == ["ValueError", "ValueError"]  # together, before the payload collective
        return
    assert [o[1] for o in outs] == ["ok", "ok"] and outs[1][2] is None
    texts = {o[0]: o[3] for o in outs}
    assert len(texts[0]) > 110  # (the old 120-byte record could not carry this row)
    assert outs[0][2] == [(0, texts[0]), (1, texts[0]), (10, texts[1]), (11, texts[1]), (12, texts[1])]


def test_shard_trials_properties(pkg):
    import importlib
    distmod = importlib.import_module("nr_randomaccess_amd.dist")
    cfgs = [pkg.make_cfg(n, seed=s) for s in range(7) for n in range(10000, 100001, 10000)]
    for world in (1, 2, 3, 8):
        parts = [distmod.shard_trials(cfgs, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(len(cfgs)))
        loads = [sum(distmod.trial_cost(cfgs[i]) for i in p) for p in parts]
        assert max(loads) - min(loads) <= 100000 * 10000  # within one largest trial
        chained = [distmod.shard_trials(cfgs, r, world, chain_by_seed=True) for r in range(world)]
        assert sorted(sum(chained, [])) == list(range(len(cfgs)))
        for p in chained:
            for s in {int(cfgs[i].seed) for i in p}:
                assert [i for i in p if int(cfgs[i].seed) == s] == [i for i in range(len(cfgs)) if int(cfgs[i].seed) == s]


def test_trial_cost_table_and_modelled_imbalance(pkg):
    """prach_trial_cost: the measured weights (profiles/r04_cost_table.json, compiled into csrc/prach_host.c) the multi-GPU dealing balances — monotone in nUE,
    exact at the sweep's points, linear between them, scaled for shortened trials; and dealing BASELINE configs[4] (--times 1000 x the ten-point sweep) to
    world = 8 by them leaves a modelled imbalance of at most 2 % (max over ranks of the dealt cost against the mean), every trial dealt exactly once."""
    import importlib
    import json
    distmod = importlib.import_module("nr_randomaccess_amd.dist")
    table = json.load(open(os.path.join(ROOT, "profiles", "r04_cost_table.json")))["programs"]
    for variant, name in ((pkg.VARIANT_BETA_C, "beta_c"), (pkg.VARIANT_WITHNOMA_C, "withnoma_c")):
        costs = []
        for n in range(10000, 100001, 10000):
            c = distmod.trial_cost(pkg.make_cfg(n, variant=variant, rng_mode=pkg.RNG_PHILOX, seed=0))
            assert abs(c - table[name][str(n)]) < 1e-6, (name, n, c)
            costs.append(c)
        assert costs == sorted(costs)
        mid = distmod.trial_cost(pkg.make_cfg(15000, variant=variant, rng_mode=pkg.RNG_PHILOX, seed=0))
        assert abs(mid - 0.5 * (costs[0] + costs[1])) < 1e-6
        half = distmod.trial_cost(pkg.make_cfg(50000, variant=variant, rng_mode=pkg.RNG_PHILOX, seed=0, max_steps=5000))
        assert abs(half - 0.5 * costs[4]) < 1e-6
    world, times = 8, 1000
    cfgs = [pkg.make_cfg(n, variant=pkg.VARIANT_BETA_C, rng_mode=pkg.RNG_PHILOX, seed=s) for s in range(times) for n in range(10000, 100001, 10000)]
    cost = [distmod.trial_cost(c) for c in cfgs]
    seen = np.zeros(len(cfgs), dtype=np.int64)
    loads = []
    for rank in range(world):
        mine = distmod.shard_trials(cfgs, rank, world)
        seen[mine] += 1
        loads.append(sum(cost[i] for i in mine))
    assert (seen == 1).all()
    assert max(loads) / (sum(loads) / world) - 1.0 <= 0.02, loads
