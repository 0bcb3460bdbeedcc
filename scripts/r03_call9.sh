set -e
export PRACH_PRINT_STAMPS=1
PRACH_LIB=$GRAFT_REPO_ROOT/5g-nr-randomaccess_amd/libprach_hip_diag.so python3 tests/tools/gpu_single.py "cluster=1" 1 100000 2>&1 | grep -E "fine stamps|cluster=1" | tail -2
PRACH_LIB=$GRAFT_REPO_ROOT/5g-nr-randomaccess_amd/libprach_hip_diag.so python3 tests/tools/gpu_single.py "cluster=1" 1 30000 2>&1 | grep -E "fine stamps|cluster=1" | tail -2
PRACH_LIB=$GRAFT_REPO_ROOT/5g-nr-randomaccess_amd/libprach_hip_diag.so python3 tests/tools/gpu_single.py "cluster=1" 0 100000 2>&1 | grep -E "fine stamps|cluster=1" | tail -2
PRACH_LIB=$GRAFT_REPO_ROOT/5g-nr-randomaccess_amd/libprach_hip_diag.so python3 tests/tools/gpu_single.py "cluster=1" 0 10000 2>&1 | grep -E "fine stamps|cluster=1" | tail -2
unset PRACH_PRINT_STAMPS
PMC_GROUPS="fetch write sq vmem tcc ea_rd ea_wr tcpw lds" scripts/gpu_pmc.sh r03b c3 -- scripts/gpu_batch.py 100 1 0
