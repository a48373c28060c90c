set -u
O=gpurun_out/r3final_d; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_and_ranks.py tests/test_gpu_e2e_files.py -x -q 2>&1 | tail -2)
python bench.py --shape medium --steps 400 --blocks 5 --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('medium', d['value'], d['timing']['host_cpus_busy'], d['host_witness']['pinned']['proofs_per_s'])"
CG_SPIN_WAIT=1 python bench.py --shape medium --steps 400 --blocks 5 --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('medium spin', d['value'], d['timing']['host_cpus_busy'], d['host_witness']['pinned']['proofs_per_s'])"
tools/profile_pmc.sh $O/pmc "rs256-sd/gates/bits=0.90" > /dev/null 2>&1
cp $O/pmc/pmc_counters.json profiles/pmc_counters.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 200 $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
python -c "
import json
for f in ('bench_default','bench_driver_args'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['timing']['spread_pct'], d['timing']['host_cpus_busy'], d['roofline_valu']['frac'], d['roofline_valu']['counters']['current'], d['host_witness']['pinned']['proofs_per_s'], d['vs_baseline'])"
