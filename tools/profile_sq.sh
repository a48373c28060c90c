#!/bin/bash
# SQ counter passes for the prove path's kernels, one proof at a time on one stream (stand-alone launches): one rocprofv3
# --pmc run per counter set (no trace domains next to --pmc), summarised per kernel and launch by tools/rocpd_counters.py.
# usage: tools/profile_sq.sh <out-dir> [bench.py flags...]
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/$1"; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
FLAGS="--steps 4 --warmup 2 --blocks 1 --headline-only --no-sweep --no-cpu-baseline --no-host-witness --no-check --no-clock-probe --inflight 1 --mode throughput --witness device $*"
i=0
: > "$OUT/sq_counters.md"
echo "SQ counters per launch (steady state), serial streams, one proof in flight: \`rocprofv3 --pmc <set> -- python3 bench.py $FLAGS\` (one pass per set), summarised by tools/rocpd_counters.py" >> "$OUT/sq_counters.md"
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $SET -d "$OUT/s$i" -o p -- python3 "$ROOT/bench.py" $FLAGS > "$OUT/s$i.line.json" 2> "$OUT/s$i.log"
  DB=$(find "$OUT/s$i" -name '*.db' | head -1)
  if [ -n "$DB" ]; then echo >> "$OUT/sq_counters.md"; python3 "$ROOT/tools/rocpd_counters.py" "$DB" | head -14 >> "$OUT/sq_counters.md"; echo >> "$OUT/sq_counters.md"; python3 "$ROOT/tools/rocpd_counters_big_launch.py" "$DB" >> "$OUT/sq_counters.md"; else echo "set $i ($SET): no database" >> "$OUT/sq_counters.md"; tail -3 "$OUT/s$i.log" >> "$OUT/sq_counters.md"; fi
  rm -rf "$OUT/s$i"
done
cat "$OUT/sq_counters.md"
