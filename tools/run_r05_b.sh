set -u
O=gpurun_out/r05_b; mkdir -p $O
T=crescent-credentials_amd/libcrescent_gpu_tuning.so
# parity first: the MSM + prove suites on the new kernel
(time python -m pytest tests/test_gpu_parity.py -m gpu -q -x) > $O/parity.log 2>&1; tail -3 $O/parity.log
# A/B on one box, alternating: round 4's unsigned accumulation (tuning build, CG_ACCUM_UNSIGNED=1) against the signed one
for i in 1 2 3; do
  CRESCENT_GPU_LIB=$T CG_ACCUM_UNSIGNED=1 python bench.py --witness device --steps 100 --no-sweep --no-cpu-baseline --no-check --no-host-witness 2>/dev/null | python tools/line_value.py "unsigned"
  CRESCENT_GPU_LIB=$T python bench.py --witness device --steps 100 --no-sweep --no-cpu-baseline --no-check --no-host-witness 2>/dev/null | python tools/line_value.py "signed"
done 2>&1 | tee $O/ab_signed.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err; echo
python tools/line_value.py default < $O/bench_default.json
tools/profile_serial.sh $O/serial > /dev/null 2>&1; tail -3 $O/serial/efficiency.md
(time python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py) > $O/gputests_rest.log 2>&1; tail -5 $O/gputests_rest.log
