#!/usr/bin/env bash
# scripts/gpu_cli_times.sh — run ON THE GPU BOX: wall time of the prach_sim command lines the README quotes (outputs into a scratch directory).
set -uo pipefail
S=$GRAFT_REPO_ROOT/5g-nr-randomaccess_amd/prach_sim
D=$(mktemp -d); cd "$D"
t() { local name="$1"; shift; local a=$(date +%s%N); "$@" > /dev/null 2> err.txt || { echo "$name: FAILED"; tail -2 err.txt; }; local b=$(date +%s%N); printf "%-52s %d ms\n" "$name" $(( (b - a) / 1000000 )); }
t "warm-up (beta -g 12 -t 10 philox)" $S --program beta -g 12 -t 10 --rng philox --logs 0 --csv w.csv
t "beta -g 12 -t 100 --rng philox --logs 0 --csv" $S --program beta -g 12 -t 100 --rng philox --logs 0 --csv results.csv
t "beta -g 12 -t 100 (glibc default) --logs 0 --csv" $S --program beta -g 12 -t 100 --logs 0 --csv results2.csv
t "beta -t 100 --logs 0 (as committed, glibc)" $S --program beta -t 100 --logs 0
t "withnoma -t 100 --logs 0 (glibc)" $S --program withnoma -t 100 --logs 0
t "beta (ten-point sweep, per-UE logs)" $S --program beta
t "withnoma (ten-point sweep, per-UE logs)" $S --program withnoma
t "noma --rng philox" $S --program noma --rng philox
t "noma --rng glibc" $S --program noma --rng glibc
head -3 results.csv
