set -u
run() { env "$@" python bench.py --steps 200 --blocks 5 --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['timing']['host_cpus_busy'], d['host_witness']['pinned']['proofs_per_s'], d['host_witness']['pinned']['host_cpus_busy'], d['host_witness']['pageable']['proofs_per_s'], d['host_witness']['pageable']['host_cpus_busy'])"; }
for i in 1 2; do run X=1; run CG_SPIN_WAIT=1; done
