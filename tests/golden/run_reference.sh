#!/usr/bin/env bash
# Generates the raw reference outputs the golden fixtures are digested from.
# Runs the REAL reference programs (compiled by `make -C oracle ref` from /root/reference into
# oracle/_ref/, nothing copied) each in its own scratch directory under oracle/_ref/runs/.
# Only works in the build container (needs /root/reference to have been compiled); the digests
# (tests/golden/*.json, made by digest_reference.py) are what is committed and what travels.
#
#   usage: tests/golden/run_reference.sh <case>     (cases below; each is independent)
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
BIN="$ROOT/oracle/_ref"
RUNS="$BIN/runs"
case "${1:?case name}" in
  beta)          d="$RUNS/beta";          mkdir -p "$d/BasicBetaSimulationResults" "$d/BasicUniformSimulationResults"
                 (cd "$d" && "$BIN/RandomAccessSimulatorBeta" > stdout.txt) ;;
  noma_default)  d="$RUNS/noma_default";  mkdir -p "$d"; (cd "$d" && "$BIN/RandomAccessWithNOMA" > stdout.txt) ;;
  noma_uniform)  d="$RUNS/noma_uniform";  mkdir -p "$d"; (cd "$d" && "$BIN/RandomAccessWithNOMA" -d 1 -t 1 > stdout.txt) ;;
  noma_odd)      d="$RUNS/noma_odd";      mkdir -p "$d"; (cd "$d" && "$BIN/RandomAccessWithNOMA" -p 64 -b 10 -g 8 -rc 4 -mrc 5 -s 10 -t 1 > stdout.txt) ;;
  noma_g54)      d="$RUNS/noma_g54";      mkdir -p "$d"; (cd "$d" && "$BIN/RandomAccessWithNOMA" -g 54 -t 1 > stdout.txt) ;;
  noma_seed2)    d="$RUNS/noma_seed2";    mkdir -p "$d"; (cd "$d" && "$BIN/RandomAccessWithNOMA" -d 0 -p 30 -b 40 -g 20 -rc 3 -mrc 20 -t 3 > stdout.txt) ;;
  noma_c)        d="$RUNS/noma_c";        mkdir -p "$d/TestResults"; (cd "$d" && "$BIN/NOMA" > stdout.txt) ;;
  *) echo "unknown case $1" >&2; exit 2 ;;
esac
echo "done $1"
