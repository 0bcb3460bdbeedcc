"""Trial sharding over the GPUs of one node (one process per GPU, torch.distributed; backend "nccl"
is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference runs its `--times` seeds x nUE sweep serially (RandomAccessWithNOMA.c:216-221).
Trials are independent in philox mode, so they are dealt to ranks by descending cost with NO
data-path collective; after the simulation ONE sum all-reduce merges the per-nUE-point aggregates
(int64, < 1 KB: latency-bound) and the per-trial rows are gathered to rank 0 for the per-seed files
and the exact results.csv.  In glibc mode the nUE points of one seed are chained
through the draw-stream offset, so whole seeds are the sharding unit.
"""
from __future__ import annotations

import numpy as np

AGG_FIELDS = ("trials", "nSuccessUE", "preambleTxCount", "sumTimer", "collisionPreambles", "totalPreambleTxop",
              "continueFaliedUEs", "finalSuccessUEs", "steps", "updates")


def trial_cost(cfg) -> float:
    """Kernel microseconds of this trial inside a batched launch: the library's measured table (prach_trial_cost, csrc/prach_host.c;
    scripts/gpu_cost_table.py) — the same weights `prach_sim --gpus N` deals with."""
    import ctypes
    from . import lib
    return float(lib().prach_trial_cost(ctypes.byref(cfg)))


def shard_trials(cfgs, rank: int, world: int, chain_by_seed: bool = False):
    """Indices of the trials this rank runs. Greedy longest-processing-time dealing (deterministic,
    identical on every rank).  chain_by_seed keeps all trials of one seed on one rank, in order."""
    n = len(cfgs)
    if world <= 1:
        return list(range(n))
    if chain_by_seed:
        seeds = sorted({int(c.seed) for c in cfgs})
        units = [[i for i in range(n) if int(cfgs[i].seed) == s] for s in seeds]
    else:
        units = [[i] for i in range(n)]
    cost = [sum(trial_cost(cfgs[i]) for i in u) for u in units]  # (floats from one C function: identical on every rank)
    order = sorted(range(len(units)), key=lambda k: (-cost[k], k))
    load = [0.0] * world
    mine = []
    for k in order:
        r = min(range(world), key=lambda q: (load[q], q))
        load[r] += cost[k]
        if r == rank:
            mine.extend(units[k])
    return sorted(mine)


def aggregate_rows(cfgs, results, points):
    """int64 [len(points), len(AGG_FIELDS)] sums of this rank's results per nUE point."""
    agg = np.zeros((len(points), len(AGG_FIELDS)), dtype=np.int64)
    pos = {p: k for k, p in enumerate(points)}
    for c, r in zip(cfgs, results):
        if getattr(r, "status", 0) != 0:  # (a trial that did not return PRACH_OK must never be averaged in)
            raise RuntimeError(f"trial (seed {int(c.seed)}, nUE {int(c.nUE)}) returned status {r.status}: refusing to aggregate it")
        k = pos[int(c.nUE)]
        row = (1, r.nSuccessUE, r.preambleTxCount, r.sumTimer, r.collisionPreambles, r.totalPreambleTxop,
               r.continueFaliedUEs, r.finalSuccessUEs, r.steps, int(c.nUE) * int(r.steps))
        agg[k] += np.array(row, dtype=np.int64)
    return agg


def allreduce_aggregates(agg: np.ndarray, device=None):
    """ONE sum all-reduce of the aggregate block across ranks (RCCL on GPUs, gloo on CPU)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return agg
    t = torch.from_numpy(agg.copy())
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


ROW_BYTES = 256  # a row = trial index (8 bytes) + kind (1) + payload length (2) + up to ROW_PAYLOAD bytes of text
ROW_PAYLOAD = ROW_BYTES - 11  # (Beta.c's six-line Results.txt is ~45 bytes, RandomAccessWithNOMA's eight lines ~115 at nUE = 100 000)


def _row_payload(val):
    if isinstance(val, (int, np.integer)):
        return str(int(val)).encode(), b"i"
    return (val if isinstance(val, bytes) else str(val).encode()), b"s"


def gather_trial_rows(rows, dst: int = 0, device=None):
    """Per-trial rows [(trial index, text or int), ...] gathered to `dst` in rank order; None elsewhere.
    Carried by ONE tensor all-gather of fixed-size byte records (a native collective of RCCL and gloo alike — no pickled-object
    collective on the critical path of the multi-GPU bench); every rank pads to the largest shard.  A row that does not fit its record
    fails on EVERY rank together: the sizes travel with the shard lengths in the first all-gather, before any payload moves."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        for _, val in rows:
            if len(_row_payload(val)[0]) > ROW_PAYLOAD:
                raise ValueError(f"row text of {len(_row_payload(val)[0])} bytes does not fit the {ROW_BYTES}-byte record")
        return list(rows)
    world = dist.get_world_size()
    packed = [(int(idx),) + _row_payload(val) for idx, val in rows]
    longest = max((len(p) for _, p, _ in packed), default=0)
    n = torch.tensor([len(rows), longest], dtype=torch.int64)
    if device is not None:
        n = n.to(device)
    heads = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(heads, n)
    heads = [h.cpu() for h in heads]
    worst = max(int(h[1]) for h in heads)
    if worst > ROW_PAYLOAD:  # every rank sees the same numbers: all of them stop here, none is left waiting in the next collective
        raise ValueError(f"row text of {worst} bytes (on some rank) does not fit the {ROW_BYTES}-byte record")
    counts = [int(h[0]) for h in heads]
    buf = np.zeros((max(max(counts), 1), ROW_BYTES), dtype=np.uint8)
    for k, (idx, payload, kind) in enumerate(packed):
        rec = idx.to_bytes(8, "little") + kind + len(payload).to_bytes(2, "little") + payload
        buf[k, :len(rec)] = np.frombuffer(rec, dtype=np.uint8)
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    parts = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    if dist.get_rank() != dst:
        return None
    out = []
    for r in range(world):
        a = parts[r].cpu().numpy()
        for k in range(counts[r]):
            raw = a[k].tobytes()
            idx = int.from_bytes(raw[:8], "little")
            ln = int.from_bytes(raw[9:11], "little")
            payload = raw[11:11 + ln]
            out.append((idx, int(payload) if raw[8:9] == b"i" else payload.decode()))
    return out
