#!/bin/bash
# Kernel timeline of ONE proof alone on the GPU (latency mode).  usage: tools/timeline_lone_proof.sh <out-dir> [shards]
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/$1"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace -d "$OUT/trace" -o t -- python3 "$ROOT/tools/probe_latency.py" ${2:-1} > "$OUT/probe.txt" 2> "$OUT/trace.log"
T=$(find "$OUT/trace" -name '*.db' | head -1)
python3 "$ROOT/tools/rocpd_timeline.py" "$T" 90 > "$OUT/timeline.txt"
rm -rf "$OUT/trace"
cat "$OUT/probe.txt"
