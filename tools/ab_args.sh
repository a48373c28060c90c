# usage: tools/ab_args.sh REPS "args A" "args B" ...: the headline measurement with each argument set, round-robin on one box
set -u
N=$1; shift
for i in $(seq $N); do
  for a in "$@"; do
    python bench.py --witness device --steps 100 --headline-only --no-sweep --no-cpu-baseline $a 2>/dev/null | python tools/line_value.py "[$a]"
  done
done
