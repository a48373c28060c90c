set -u
mkdir -p gpurun_out/ab2
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  for sh in 1 8; do
    CG_GATE_ACCUM=$1 CG_CHAIN_PRIORITY=$2 python tools/probe_latency.py $sh 2>/dev/null | sed "s/^/gate=$1 prio=$2 /"
  done
done | tee gpurun_out/ab2/latency.txt
for g in 0 1 0 1; do
  CG_GATE_ACCUM=$g python bench.py --steps 80 --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gate=$g', d['value'])"
done | tee gpurun_out/ab2/throughput.txt
