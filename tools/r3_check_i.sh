set -u
run() { env "$@" python bench.py --steps 100 --blocks 5 --no-sweep --no-cpu-baseline --no-host-witness $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$* $EXTRA', d['value'], d['timing']['spread_pct'], d['phase_ms']['accum_g2_ms'])"; }
for i in 1 2; do
EXTRA="" run X=1
EXTRA="" run CG_G2_ONE_LANE=1
EXTRA="--bits 0" run X=1
EXTRA="--bits 0" run CG_G2_ONE_LANE=1
done
