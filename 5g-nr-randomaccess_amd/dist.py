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


def trial_cost(cfg) -> int:
    return int(cfg.nUE) * (60000 if cfg.uniform else 10000)


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
    cost = [sum(trial_cost(cfgs[i]) for i in u) for u in units]
    order = sorted(range(len(units)), key=lambda k: (-cost[k], k))
    load = [0] * world
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


ROW_BYTES = 120  # a row = (trial index, short text): index in 8 bytes + up to 112 bytes of text


def gather_trial_rows(rows, dst: int = 0, device=None):
    """Per-trial rows [(trial index, text or int), ...] gathered to `dst` in rank order; None elsewhere.
    Carried by ONE tensor all-gather of fixed-size byte records (a native collective of RCCL and gloo alike — no pickled-object
    collective on the critical path of the multi-GPU bench); every rank pads to the largest shard."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(rows)
    world = dist.get_world_size()
    n = torch.tensor([len(rows)], dtype=torch.int64)
    if device is not None:
        n = n.to(device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    nmax = max(int(c.item()) for c in counts)
    buf = np.zeros((max(nmax, 1), ROW_BYTES), dtype=np.uint8)
    kinds = set()
    for k, (idx, val) in enumerate(rows):
        if isinstance(val, (int, np.integer)):
            payload, kind = str(int(val)).encode(), b"i"
        else:
            payload, kind = (val if isinstance(val, bytes) else str(val).encode()), b"s"
        if len(payload) > ROW_BYTES - 10:
            raise ValueError(f"row text of {len(payload)} bytes does not fit the {ROW_BYTES}-byte record")
        kinds.add(kind)
        rec = int(idx).to_bytes(8, "little") + kind + bytes([len(payload)]) + payload
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
        for k in range(int(counts[r].item())):
            raw = a[k].tobytes()
            idx = int.from_bytes(raw[:8], "little")
            ln = raw[9]
            payload = raw[10:10 + ln]
            out.append((idx, int(payload) if raw[8:9] == b"i" else payload.decode()))
    return out
