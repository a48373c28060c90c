set -u
O=gpurun_out/r05_c; mkdir -p $O
T=crescent-credentials_amd/libcrescent_gpu_tuning.so
(time python -m pytest tests/test_gpu_parity.py -m gpu -q -x) > $O/parity.log 2>&1; tail -3 $O/parity.log
for i in 1 2 3; do
  CRESCENT_GPU_LIB=$T CG_ACCUM_UNSIGNED=1 python bench.py --witness device --steps 100 --no-sweep --no-cpu-baseline --no-check --no-host-witness 2>/dev/null | python tools/line_value.py "unsigned"
  CRESCENT_GPU_LIB=$T python bench.py --witness device --steps 100 --no-sweep --no-cpu-baseline --no-check --no-host-witness 2>/dev/null | python tools/line_value.py "signed"
done 2>&1 | tee $O/ab_signed.txt
tools/profile_serial.sh $O/serial > /dev/null 2>&1; tail -3 $O/serial/efficiency.md; head -9 $O/serial/accum_launches.md
python bench.py --gpus 2 --backend gloo --steps 24 --warmup 4 --no-host-witness --no-check > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; tail -c 300 $O/bench_2rank_gloo.err
python -c "
import json; d=json.load(open('$O/bench_2rank_gloo.json')); print(d['value'], json.dumps(d['sharded'].get('arrangements'))[:900])"
(time python -m pytest tests/test_gpu_host_and_ranks.py -m gpu -q -x) > $O/ranks.log 2>&1; tail -3 $O/ranks.log
rocprofv3 --list-avail 2>/dev/null | grep -i -B2 -A12 "pc.sampl" | head -60 > $O/pcsamp_avail.txt; head -30 $O/pcsamp_avail.txt
