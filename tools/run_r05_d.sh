set -u
O=gpurun_out/r05_d; mkdir -p $O
T=crescent-credentials_amd/libcrescent_gpu_tuning.so
B="python bench.py --witness device --steps 100 --no-sweep --no-cpu-baseline --no-check --no-host-witness"
# parity of every flush variant on the MSM + prove suites (tuning build)
for how in 1 2; do
  CRESCENT_GPU_LIB=$T CG_FLUSH=$how python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "msm or prove_equals or prove_golden or sharded" > $O/parity_flush$how.log 2>&1; tail -1 $O/parity_flush$how.log
done
(for i in 1 2 3; do
  CRESCENT_GPU_LIB=$T CG_ACCUM_UNSIGNED=1 $B 2>/dev/null | python tools/line_value.py "unsigned (round 4)"
  for how in 0 1 2; do CRESCENT_GPU_LIB=$T CG_FLUSH=$how $B 2>/dev/null | python tools/line_value.py "signed, flush $how"; done
done) 2>&1 | tee $O/flush_variants.txt
# stand-alone h launch per variant
for how in 0 1 2; do
  CRESCENT_GPU_LIB=$T CG_FLUSH=$how tools/profile_serial.sh $O/serial$how > /dev/null 2>&1; echo "flush $how"; sed -n 5,6p $O/serial$how/accum_launches.md; tail -1 $O/serial$how/efficiency.md
done 2>&1 | tee $O/flush_variants_serial.txt
(time python -m pytest tests/test_gpu_host_and_ranks.py -m gpu -q -x) > $O/ranks.log 2>&1; tail -3 $O/ranks.log
rocprofv3 --list-avail 2>/dev/null | grep -i -B2 -A12 "pc.sampl" | head -60 > $O/pcsamp_avail.txt; head -30 $O/pcsamp_avail.txt
