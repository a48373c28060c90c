# quick GPU confidence run: the GPU tests except the full-size ones, a lone-proof latency probe (1 and 8 shards), a short bench
set -u
mkdir -p gpurun_out/chk
(time python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py) > gpurun_out/chk/tests.log 2>&1; tail -3 gpurun_out/chk/tests.log
python tools/probe_latency.py 1 2>/dev/null | tee gpurun_out/chk/lat1.txt
python tools/probe_latency.py 8 2>/dev/null | tee gpurun_out/chk/lat8.txt
python bench.py --steps 80 --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['phase_ms'])" | tee gpurun_out/chk/bench.txt
