set -u
O=gpurun_out/r3shapes; mkdir -p $O
for sh in rs256 mdl1 rs256-db; do
  python bench.py --shape $sh --no-sweep > $O/bench_$sh.json 2> $O/bench_$sh.err
  python -c "
import json; d=json.load(open('$O/bench_$sh.json')); print('$sh', d['value'], d['timing']['spread_pct'], d.get('host_witness',{}).get('pinned',{}).get('proofs_per_s'), d['cpu_baseline'].get('value'), d['cpu_baseline'].get('proof_bytes_identical_to_gpu'))"
done
python tools/sweep.py --g1 10:22 --g2 10:20 --ntt 10:24 --reps 5 --out $O/unit_sweep_uniform.md > /dev/null 2>&1; tail -5 $O/unit_sweep_uniform.md
python tools/sweep.py --g1 10:22 --g2 10:20 --ntt 21:21 --reps 5 --bits 0.9 --out $O/unit_sweep_circom.md > /dev/null 2>&1; tail -3 $O/unit_sweep_circom.md
