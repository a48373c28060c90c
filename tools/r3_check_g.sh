set -u
O=gpurun_out/r3h; mkdir -p $O
tools/profile_serial.sh $O/serial > /dev/null 2>&1; cat $O/serial/accum_launches.md
run() { env "$@" python bench.py --steps 100 --blocks 5 --no-sweep --no-cpu-baseline --no-host-witness $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$* $EXTRA', d['value'], d['timing']['spread_pct'], d['roofline_valu']['sustained_clock_ghz'])"; }
for i in 1 2; do
EXTRA="" run X=1
EXTRA="--inflight 8" run X=1
EXTRA="--inflight 16" run X=1
EXTRA="--inflight 20" run X=1
EXTRA="" run CG_MIN_SEGMENT=32
EXTRA="" run CG_MIN_SEGMENT=128
done
