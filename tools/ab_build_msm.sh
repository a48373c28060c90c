# usage: tools/ab_build_msm.sh REPS "<extra hipcc flags>": a resident 2^21-point uniform G1 MSM (tools/probe_msm.py) and the headline
# bench with the default build and with the flagged build, alternating on one box
set -u
N=$1; FLAGS=$2
export CG_BUILD_JOBS=16
run() { python tools/probe_msm.py --k 21 --group 1 --reps 5 2>/dev/null | tail -1 | python -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('$1 msm accum_g1_ms', round(d['accum_g1_ms'],3), 'total', round(d['total_ms'],3))"
        python bench.py --steps 200 --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 bench', d['value'])"; }
for i in $(seq $N); do
  python crescent-credentials_amd/build.py > /dev/null 2>&1; run default
  CG_HIPCC_EXTRA="$FLAGS" python crescent-credentials_amd/build.py > /dev/null 2>&1; run "[$FLAGS]"
done
python crescent-credentials_amd/build.py > /dev/null 2>&1
