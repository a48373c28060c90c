set -u
O=gpurun_out/r05_f; mkdir -p $O
T=$PWD/crescent-credentials_amd/libcrescent_gpu_tuning.so
B="python bench.py --witness device --steps 100 --no-sweep --no-cpu-baseline --no-check --no-host-witness"
(for i in 1 2 3 4 5 6; do
  CRESCENT_GPU_LIB=$T CG_ACCUM_UNSIGNED=1 $B 2>/dev/null | python tools/line_value.py "unsigned (round 4)"
  CRESCENT_GPU_LIB=$T CG_FLUSH=0 $B 2>/dev/null | python tools/line_value.py "signed, flush 0"
  CRESCENT_GPU_LIB=$T CG_FLUSH=1 $B 2>/dev/null | python tools/line_value.py "signed, flush 1"
done) 2>&1 | tee $O/signed_vs_unsigned.txt
(for i in 1 2; do
  CRESCENT_GPU_LIB=$T CG_ACCUM_UNSIGNED=1 $B --bits 0 2>/dev/null | python tools/line_value.py "uniform unsigned"
  CRESCENT_GPU_LIB=$T CG_FLUSH=0 $B --bits 0 2>/dev/null | python tools/line_value.py "uniform signed flush 0"
  CRESCENT_GPU_LIB=$T CG_FLUSH=1 $B --bits 0 2>/dev/null | python tools/line_value.py "uniform signed flush 1"
done) 2>&1 | tee $O/uniform.txt
for how in 0 1; do
  CRESCENT_GPU_LIB=$T CG_FLUSH=$how tools/profile_serial.sh $O/serial$how > /dev/null 2>&1; echo "flush $how"; sed -n 5,6p $O/serial$how/accum_launches.md; tail -1 $O/serial$how/efficiency.md
done 2>&1 | tee $O/flush_variants_serial.txt
