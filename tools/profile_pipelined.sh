#!/bin/bash
# Kernel trace of the default pipelined run (4 proofs in flight): per-kernel durations UNDER concurrency.
# usage: tools/profile_pipelined.sh <out-dir> [bench.py flags...]
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/$1"; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace -d "$OUT/trace" -o t -- python3 "$ROOT/bench.py" --steps 40 --warmup 8 --blocks 3 --headline-only --no-sweep --no-cpu-baseline --no-host-witness --no-check --no-clock-probe "$@" > "$OUT/trace_line.json" 2> "$OUT/trace.log"
T=$(find "$OUT/trace" -name '*.db' | head -1)
python3 "$ROOT/tools/rocpd_stats.py" "$T" "$OUT/kernel_stats.md" > /dev/null
python3 "$ROOT/tools/rocpd_busy.py" "$T" > "$OUT/busy.txt" 2>&1
rm -rf "$OUT/trace"
head -30 "$OUT/kernel_stats.md"
