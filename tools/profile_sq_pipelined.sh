#!/bin/bash
# SQ counter passes over the DEFAULT run (twelve proofs in flight, throughput contexts): the launches the pipelined run makes,
# with its segment lengths and its one-stream ordering.  NOTE: rocprofv3 serialises kernels while it collects counters, so
# these are per-launch figures of the pipelined run's launches, not of kernels overlapping each other (tools/profile_pcsamp.sh
# is the concurrent view).  One pass per counter set (no trace domain next to --pmc); reduced per kernel by
# tools/rocpd_counters.py and per kernel FAMILY, with derived fractions, by tools/rocpd_families.py.
# usage: tools/profile_sq_pipelined.sh <out-dir> [bench.py flags...]
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/$1"; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
FLAGS="--steps 20 --warmup 4 --blocks 2 --headline-only --no-sweep --no-cpu-baseline --no-host-witness --no-check --no-clock-probe --witness device $*"
i=0
DBS=""
for SET in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $SET -d "$OUT/s$i" -o p -- python3 "$ROOT/bench.py" $FLAGS > "$OUT/s$i.line.json" 2> "$OUT/s$i.log"
  DB=$(find "$OUT/s$i" -name '*.db' | head -1)
  [ -n "$DB" ] && { python3 "$ROOT/tools/rocpd_counters.py" "$DB" "$OUT/set$i.md" > /dev/null; DBS="$DBS $DB"; }
done
python3 "$ROOT/tools/rocpd_families.py" "$OUT/sq_families_pipelined_run.md" $DBS
rm -rf "$OUT"/s1 "$OUT"/s2 "$OUT"/s3
cat "$OUT/sq_families_pipelined_run.md"
