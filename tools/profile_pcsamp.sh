#!/bin/bash
# Stochastic PC sampling (rocprofv3, beta) of the DEFAULT pipelined run - the one measurement that sees the chip with a dozen
# proofs in flight: counter collection (--pmc) serialises kernels, a kernel trace has no issue/stall information.  Every sample
# says whether the sampled wave issued an instruction that cycle, of which type, or why it stalled, and - on gfx950 - what the
# SIMD's arbiters were doing; tools/pcsamp_summary.py reduces the samples per kernel family.
# Best effort: the feature is beta; run it under a short timeout, after the other measurements are safe.
# usage: tools/profile_pcsamp.sh <out-dir> [interval-cycles (power of two)] [bench.py flags...]
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/$1"; shift
IV=${1:-8388608}; [ $# -gt 0 ] && shift
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
FLAGS="--steps 20 --warmup 4 --blocks 2 --headline-only --no-sweep --no-cpu-baseline --no-host-witness --no-check --no-clock-probe --witness device $*"
timeout 420 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit cycles --pc-sampling-method stochastic --pc-sampling-interval $IV \
  --kernel-trace --output-format csv -d "$OUT/raw" -o pcs -- python3 "$ROOT/bench.py" $FLAGS > "$OUT/line.json" 2> "$OUT/run.log"
echo "rc=$?" >> "$OUT/run.log"
find "$OUT/raw" -type f | head -20 >> "$OUT/run.log"
python3 "$ROOT/tools/pcsamp_summary.py" "$OUT/raw" "$OUT/pc_sampling_summary.md" >> "$OUT/run.log" 2>&1
# keep the head of every raw file for the record (the full sample files are tens of MB), drop the rest
for f in $(find "$OUT/raw" -type f -name '*.csv'); do head -40 "$f" > "$OUT/head_$(basename $f)"; done
rm -rf "$OUT/raw"
tail -5 "$OUT/run.log"
