# usage: tools/ab_envs.sh REPS "bench args" "ENV1=a ENV2=b" "ENV1=c" ...: round-robin over environment settings ("-" = none)
set -u
N=$1; ARGS=$2; shift; shift
for i in $(seq $N); do
  for e in "$@"; do
    if [ "$e" = "-" ]; then E=""; else E="$e"; fi
    env $E python bench.py --steps 100 --no-sweep --no-cpu-baseline --no-check $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$e]', d['value'])"
  done
done
