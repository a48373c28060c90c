set -u
O=gpurun_out/r05_o; mkdir -p $O
P=$PWD/crescent-credentials_amd/libcrescent_gpu_prev.so
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ntt or prove_equals or prove_golden or witness or edge_shapes or strided" > $O/parity_ntt.log 2>&1; tail -1 $O/parity_ntt.log
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "transforms" > $O/fullsize_ntt.log 2>&1; tail -1 $O/fullsize_ntt.log
B="python bench.py --witness device --steps 100 --no-sweep --no-cpu-baseline --no-check --no-host-witness"
(for i in 1 2 3 4; do
  CRESCENT_GPU_LIB=$P $B 2>/dev/null | python tools/line_value.py "a block barrier after every stage pair"
  $B 2>/dev/null | python tools/line_value.py "no block barrier between wave-local stage pairs"
done) 2>&1 | tee $O/ntt_wave_local.txt
tools/profile_serial.sh $O/serial > /dev/null 2>&1; grep "k_ntt29\|total" $O/serial/efficiency.md | tee -a $O/ntt_wave_local.txt
