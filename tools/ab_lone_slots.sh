# the headline rate and the rate / latency against proofs in flight with 0, 1, 2, 3 lone slots (tuning build: CG_LONE_SLOTS), alternating on one box
# usage: tools/ab_lone_slots.sh "0 1 2 3" [reps]
set -u
VALS=${1:-"0 1 2"}; N=${2:-2}
for i in $(seq $N); do
  for x in $VALS; do
    env CG_LONE_SLOTS=$x CRESCENT_GPU_LIB=$PWD/crescent-credentials_amd/libcrescent_gpu_tuning.so python bench.py --no-cpu-baseline --no-sweep --no-shapes --no-cold-start --no-check --no-host-witness --no-ubench 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lone_slots=$x value %.2f  ' % d['value'] + '  '.join('%d: %.1f/s p50 %.2f ms' % (p['in_flight'], p['proofs_per_s'], p['latency_ms']['p50']) for p in d['inflight_curve']['points']))"
  done
done
