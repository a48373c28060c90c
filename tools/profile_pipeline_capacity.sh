#!/bin/bash
# Kernel trace of the default pipelined run (16 proofs in flight) reduced to where the chip's capacity goes
# (tools/rocpd_pipeline.py), plus the plain per-kernel table.   usage: tools/profile_pipeline_capacity.sh <out-dir> [bench.py flags...]
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/$1"; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace -d "$OUT/trace" -o t -- python3 "$ROOT/bench.py" --steps 40 --warmup 8 --blocks 6 --headline-only --no-sweep --no-cpu-baseline --no-host-witness --no-clock-probe --no-check "$@" > "$OUT/trace_line.json" 2> "$OUT/trace.log"
T=$(find "$OUT/trace" -name '*.db' | head -1)
python3 -c "import sqlite3,sys; print([r[1] for r in sqlite3.connect(sys.argv[1]).execute('pragma table_info(kernels)')])" "$T" > "$OUT/kernels_columns.txt"
python3 "$ROOT/tools/rocpd_pipeline.py" "$T" > "$OUT/pipeline_capacity.md" 2> "$OUT/pipeline_capacity.err"
python3 "$ROOT/tools/rocpd_stats.py" "$T" "$OUT/kernel_stats.md" > /dev/null
rm -rf "$OUT/trace"
cat "$OUT/kernels_columns.txt"; head -60 "$OUT/pipeline_capacity.md"; tail -3 "$OUT/pipeline_capacity.err"
