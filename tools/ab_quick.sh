# quick check on one box: parity tests, the default headline measurement three times, the per-proof instruction count
set -u
python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e_files.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3; do python bench.py --witness device --steps 80 --headline-only --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['phase_ms']['accum_g1_ms'], d['phase_ms']['witness_map_ms'], d['phase_ms']['total_ms'])"; done
tools/profile_pmc.sh gpurun_out/pmcq "rs256-sd/gates/bits=0.90" > /dev/null 2>&1; grep "^|\|total" gpurun_out/pmcq/valu_per_proof.md
