#!/usr/bin/env bash
# scripts/gpu_profile.sh TAG — run ON THE GPU BOX (through gpurun): the round's rocprofv3 evidence, written under gpurun_out/prof_TAG/.
#   1. kernel trace + stats of the bench command (the headline kernel's average duration must agree with bench.py's roofline)
#   2. FETCH_SIZE and WRITE_SIZE of the same command, separate --pmc passes (MI355X_MICROARCH.md, HBM / rocprofv3 PMC slots)
#   3. the same three passes for BASELINE config 3 (1000 concurrent trials: scripts/gpu_batch.py 100)
#   4. FETCH_SIZE / WRITE_SIZE calibration of 8-byte-per-lane accesses (profiles/tools/copy_probe.hip)
# Never combines --pmc with a trace domain; the profiled program is python3 / the probe binary itself.
set -euo pipefail
TAG="${1:?tag}"
OUT="gpurun_out/prof_${TAG}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/bench_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/bench_write.log" 2>&1
C3="python3 scripts/gpu_batch.py 100 1 0"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c3_trace" -- $C3 > "$OUT/c3_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/c3_fetch" -- $C3 > "$OUT/c3_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/c3_write" -- $C3 > "$OUT/c3_write.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$OUT/c3_sq" -- $C3 > "$OUT/c3_sq.log" 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 profiles/tools/copy_probe.hip -o /tmp/copy_probe
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/probe_fetch" -- /tmp/copy_probe > "$OUT/probe_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/probe_write" -- /tmp/copy_probe > "$OUT/probe_write.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/probe_trace" -- /tmp/copy_probe > "$OUT/probe_trace.log" 2>&1
tail -2 "$OUT/bench_trace.log"; tail -1 "$OUT/c3_trace.log"
echo "profiles in $OUT"
