set -u
O=gpurun_out/r3f; mkdir -p $O
hw() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); h=d.get('host_witness',{})
print('$2', 'device', d['value'], 'pinned', h.get('pinned',{}).get('proofs_per_s'), 'pageable', h.get('pageable',{}).get('proofs_per_s'), 'clock', d['roofline_valu']['sustained_clock_ghz'], 'spread', d['timing']['spread_pct'])"; }
for i in 1 2; do
  python bench.py --steps 100 --blocks 5 --no-sweep --no-cpu-baseline > $O/hw_$i.json 2>$O/hw_$i.err; hw $O/hw_$i.json new
done
python tools/exp_shard_idle.py 2>&1 | grep "gap" | tail -6
(time timeout 900 python bench.py --gpus 2 --backend gloo --steps 24 --no-host-witness) > $O/bench_2rank.json 2> $O/bench_2rank.err; tail -2 $O/bench_2rank.err
python -c "
import json; d=json.load(open('$O/bench_2rank.json')); print(d['value'], json.dumps(d['sharded']))"
(time timeout 1700 python -m pytest tests/test_gpu_host_and_ranks.py tests/test_gpu_e2e_files.py -x -q) > $O/tests.log 2>&1; tail -5 $O/tests.log
