set -u
O=gpurun_out/r05_i; mkdir -p $O
T=$PWD/crescent-credentials_amd/libcrescent_gpu_tuning.so
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "signed_accumulation or context_flags or h_scalars" > $O/new_tests.log 2>&1; tail -2 $O/new_tests.log
tools/profile_pipelined.sh $O/pipelined > /dev/null 2>&1; head -8 $O/pipelined/kernel_stats.md | cut -c1-150
tools/profile_sq_pipelined.sh $O/sqpipe > $O/sqpipe.out 2>&1; tail -3 $O/sqpipe.out | cut -c1-200
B="python bench.py --steps 100 --no-sweep --no-cpu-baseline --no-check --no-host-witness"
(for i in 1 2; do
  $B 2>/dev/null | python tools/line_value.py "host witness, 12 in flight"
  $B --inflight 16 2>/dev/null | python tools/line_value.py "host witness, 16 in flight"
  $B --inflight 10 2>/dev/null | python tools/line_value.py "host witness, 10 in flight"
  CRESCENT_GPU_LIB=$T CG_NTT_TILE=10 $B 2>/dev/null | python tools/line_value.py "1024-element transform tiles"
done) 2>&1 | tee $O/inflight_and_tiles.txt
