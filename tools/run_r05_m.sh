set -u
O=gpurun_out/r05_m; mkdir -p $O
D=$PWD/crescent-credentials_amd
B="python bench.py --witness device --steps 100 --no-sweep --no-cpu-baseline --no-check --no-host-witness"
for v in sell2 sell4; do CRESCENT_GPU_LIB=$D/libcrescent_gpu_$v.so python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "prove_equals or prove_golden or witness or edge_shapes" 2>&1 | tail -1; done
(for i in 1 2 3 4; do
  CRESCENT_GPU_LIB=$D/libcrescent_gpu_prev.so $B 2>/dev/null | python tools/line_value.py "round-5 kernels before the transform change"
  CRESCENT_GPU_LIB=$D/libcrescent_gpu_tw.so $B 2>/dev/null | python tools/line_value.py "first twiddle of the next pair ahead"
  CRESCENT_GPU_LIB=$D/libcrescent_gpu_sell2.so $B 2>/dev/null | python tools/line_value.py "... + sparse product gathers 2 at a time"
  CRESCENT_GPU_LIB=$D/libcrescent_gpu_sell4.so $B 2>/dev/null | python tools/line_value.py "... + sparse product gathers 4 at a time"
done) 2>&1 | tee $O/sell_batch.txt
for v in tw sell2 sell4; do CRESCENT_GPU_LIB=$D/libcrescent_gpu_$v.so tools/profile_serial.sh $O/serial_$v > /dev/null 2>&1; echo $v; grep "k_sell29\|total" $O/serial_$v/efficiency.md; done 2>&1 | tee $O/sell_batch_serial.txt
