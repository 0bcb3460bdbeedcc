#!/usr/bin/env bash
# scripts/gpu_pmc.sh TAG NAME -- <python script + args>   (run ON THE GPU BOX through gpurun)
# One rocprofv3 pass per counter group (never --pmc together with a trace domain; the profiled program is python3 itself) for the
# given command, written under gpurun_out/prof_TAG/NAME_*.  Groups: kernel trace + stats; FETCH_SIZE; WRITE_SIZE; SQ wait / issue;
# VMEM level (average memory latency = SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM); LDS level; L2 hit / miss; memory-side requests by size.
set -euo pipefail
TAG="${1:?tag}"; NAME="${2:?name}"; shift 2
[ "$1" = "--" ] && shift
OUT="gpurun_out/prof_${TAG}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { # group-name counters...   (PMC_GROUPS="fetch write": only those passes)
    local grp="$1"; shift
    if [ -n "${PMC_GROUPS:-}" ] && ! [[ " $PMC_GROUPS " == *" $grp "* ]]; then return 0; fi
    rocprofv3 --pmc "$@" --output-format csv -d "$OUT/${NAME}_${grp}" -- python3 "${CMD[@]}" > "$OUT/${NAME}_${grp}.log" 2>&1 || { echo "pass $grp failed"; tail -5 "$OUT/${NAME}_${grp}.log"; }
    echo "pass $grp done"
}
CMD=("$@")
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${NAME}_trace" -- python3 "${CMD[@]}" > "$OUT/${NAME}_trace.log" 2>&1
tail -2 "$OUT/${NAME}_trace.log"
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES
run vmem SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH
run lds SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_INST_LEVEL_SMEM
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum
run tcpw TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
run ea_rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run ea_wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_ATOMIC_LEVEL_sum
echo "profiles in $OUT"
