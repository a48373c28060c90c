#!/bin/bash
# Stand-alone kernel durations and VALU instruction counts of one proof at a time, every kernel on one stream
# (`--inflight 1 --mode throughput`: CG_FLAG_THROUGHPUT_MODE on a one-slot context): two rocprofv3 runs of the same command (kernel trace; SQ_INSTS_VALU), summarised by
# tools/rocpd_efficiency.py.   usage: tools/profile_serial.sh <out-dir> [bench.py flags...]
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/$1"; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
FLAGS="--steps 8 --warmup 2 --blocks 1 --headline-only --no-sweep --no-cpu-baseline --no-host-witness --no-check --no-clock-probe --inflight 1 --mode throughput --witness device $*"
rocprofv3 --kernel-trace -d "$OUT/trace" -o t -- python3 "$ROOT/bench.py" $FLAGS > "$OUT/trace_line.json" 2> "$OUT/trace.log"
rocprofv3 --pmc SQ_INSTS_VALU -d "$OUT/pmc" -o p -- python3 "$ROOT/bench.py" $FLAGS > "$OUT/pmc_line.json" 2> "$OUT/pmc.log"
T=$(find "$OUT/trace" -name '*.db' | head -1); P=$(find "$OUT/pmc" -name '*.db' | head -1)
python3 "$ROOT/tools/rocpd_efficiency.py" "$T" "$P" "$OUT/efficiency.md" > /dev/null
python3 "$ROOT/tools/rocpd_stats.py" "$T" "$OUT/kernel_stats.md" > /dev/null
python3 "$ROOT/tools/rocpd_accum_launches.py" "$T" "$P" "$OUT/accum_launches.md" > /dev/null
rm -rf "$OUT/trace" "$OUT/pmc"
tail -3 "$OUT/efficiency.md"
