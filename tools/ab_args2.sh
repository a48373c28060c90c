set -u
N=$1; shift
for i in $(seq $N); do
  for a in "$@"; do
    python bench.py --no-sweep --no-cpu-baseline $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$a]', d['value'])"
  done
done
