#!/usr/bin/env python3
"""rand() calls consumed before each nUE point of the reference's chained sweeps (one srand per seed,
Beta.c:69-71), computed with the oracle while re-verifying every point against the reference
fixture (Results.txt bytes + Logs SHA-256).  Lets the GPU tests / bench start a chain at any point
without replaying the earlier ones.  Output: tests/golden/stream_offsets.json."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "stream_offsets.json")
out = json.load(open(OUT)) if os.path.exists(OUT) else {}
for case in sys.argv[1:] or ["beta", "noma_default", "noma_uniform", "noma_odd", "noma_g54"]:
    path = os.path.join(ROOT, "tests", "golden", f"{case}.json")
    if not os.path.exists(path):
        continue
    g = json.load(open(path))
    variant = ob.VARIANT_BETA_C if g["variant"] == "BETA_C" else ob.VARIANT_WITHNOMA_C
    rngs, offs = {}, {}
    for tr in g["trials"]:
        if tr["seed"] != 0:
            continue
        rng = rngs.setdefault(0, ob.Rng(ob.RNG_GLIBC, 0))
        offs[str(tr["nUE"])] = rng.consumed()
        cfg = ob.make_cfg(tr["nUE"], variant=variant, **g["cfg_overrides"])
        res, ues = ob.run_trial(cfg, rng)
        assert ob.format_results(cfg, res).decode() == tr["results_text"], (case, tr["nUE"])
        assert hashlib.sha256(ob.format_logs(ues, cfg.nUE)).hexdigest() == tr["logs_sha256"], (case, tr["nUE"])
        print(case, tr["nUE"], "verified; offset", offs[str(tr["nUE"])], flush=True)
    offs["end"] = rngs[0].consumed()
    out[case] = offs
    json.dump(out, open(OUT, "w"), indent=1)
