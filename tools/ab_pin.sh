set -u
mkdir -p gpurun_out/pin
cd tools/ubench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include f29_rates.hip -o f29_pin 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -DCG_NO_PIN f29_rates.hip -o f29_nopin 2>/dev/null
cd ../..
for v in pin nopin pin nopin; do echo "== $v"; tools/ubench/f29_$v | grep -v gather | awk '$2==4 || $3==4 || /blk/'; done | tee gpurun_out/pin/ubench.txt
export CG_BUILD_JOBS=16
run() { python bench.py --steps 80 --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['phase_ms']['accum_g1_ms'], d['phase_ms']['witness_map_ms'])"; }
run pin; run pin
CG_HIPCC_EXTRA="-DCG_NO_PIN" python crescent-credentials_amd/build.py > /dev/null 2>&1
run nopin; run nopin
python crescent-credentials_amd/build.py > /dev/null 2>&1
run pin; run pin
CG_HIPCC_EXTRA="-DCG_NO_PIN" python crescent-credentials_amd/build.py > /dev/null 2>&1
run nopin
python crescent-credentials_amd/build.py > /dev/null 2>&1
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
