# usage: tools/ab_env.sh VAR [reps]: the default headline measurement with VAR unset / VAR=1, alternating, on one box
set -u
V=$1; N=${2:-4}
for i in $(seq $N); do
  python bench.py --steps 100 --no-sweep --no-cpu-baseline --no-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'])"
  env $V=1 python bench.py --steps 100 --no-sweep --no-cpu-baseline --no-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=1', d['value'])"
done
