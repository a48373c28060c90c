set -u
mkdir -p gpurun_out/ab3
for n in 2 3 4 5 4 3; do
  python bench.py --steps 80 --inflight $n --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight=$n', d['value'])"
done | tee gpurun_out/ab3/inflight.txt
tools/profile_pipelined.sh gpurun_out/ab3/pipe > /dev/null 2>&1
cat gpurun_out/ab3/pipe/busy.txt
