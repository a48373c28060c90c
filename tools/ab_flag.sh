# usage: tools/ab_flag.sh "<compiler flag>": parity subset, then the 2^21 G1 MSM probe and the pipelined rate with the default build and with the flagged build, alternating on one box (the default build is restored on exit)
export CG_BUILD_JOBS=16
trap 'python crescent-credentials_amd/build.py > /dev/null 2>&1' EXIT
FLAG="$1"
probe() { python tools/probe_msm.py --group 1 --k 21 --reps 5 2>/dev/null | tail -1 | python -c "
import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('$1', 'accum_g1_ms', round(d['accum_g1_ms'],3), 'msm_h_ms', round(d['msm_h_ms'],3))"; }
run() { python bench.py --witness device --steps 100 --blocks 5 --headline-only --no-sweep --no-cpu-baseline --no-host-witness --no-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['timing']['median_block']['spread_pct'])"; }
(timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "msm or prove_golden" 2>&1 | tail -1)
for i in 1 2; do
  python crescent-credentials_amd/build.py > /dev/null 2>&1; probe default; run default
  CG_HIPCC_EXTRA="$FLAG" python crescent-credentials_amd/build.py > /dev/null 2>&1; probe "$FLAG"; run "$FLAG"
done
CG_HIPCC_EXTRA="$FLAG" python crescent-credentials_amd/build.py > /dev/null 2>&1
(timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "msm or prove_golden or edge" 2>&1 | tail -1)
