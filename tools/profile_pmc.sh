#!/bin/bash
# The three counter passes behind profiles/pmc_counters.json (one counter per pass, kernel trace off), every kernel of a
# proof on one stream.   usage: tools/profile_pmc.sh <out-dir> <workload-key> [bench.py flags...]
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/$1"; KEY="$2"; shift; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
FLAGS="--steps 6 --warmup 2 --blocks 1 --headline-only --no-sweep --no-cpu-baseline --no-host-witness --no-check --no-clock-probe --inflight 1 --mode throughput --witness device $*"
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  rocprofv3 --pmc $C -d "$OUT/$C" -o p -- python3 "$ROOT/bench.py" $FLAGS > "$OUT/$C.line.json" 2> "$OUT/$C.log"
done
F=$(find "$OUT/FETCH_SIZE" -name '*.db' | head -1); W=$(find "$OUT/WRITE_SIZE" -name '*.db' | head -1); V=$(find "$OUT/SQ_INSTS_VALU" -name '*.db' | head -1)
python3 "$ROOT/tools/make_pmc_json.py" "$KEY" "$F" "$W" "$V" "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU> -- python3 bench.py $FLAGS" "$OUT/pmc_counters.json"
python3 "$ROOT/tools/rocpd_pmc.py" "$V" SQ_INSTS_VALU "$OUT/valu_per_proof.md" > /dev/null
# where the HBM traffic of a proof goes, kernel by kernel (KB per proof, raw counters)
python3 "$ROOT/tools/rocpd_pmc.py" "$F" FETCH_SIZE "$OUT/fetch_kb_per_proof.md" > /dev/null
python3 "$ROOT/tools/rocpd_pmc.py" "$W" WRITE_SIZE "$OUT/write_kb_per_proof.md" > /dev/null
rm -rf "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE" "$OUT/SQ_INSTS_VALU"
