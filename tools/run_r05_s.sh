set -u
O=gpurun_out/r05_s; mkdir -p $O
L=$PWD/crescent-credentials_amd/libcrescent_gpu_cap32.so
B="python bench.py --steps 100 --no-sweep --no-cpu-baseline --no-check --no-host-witness"
(for i in 1 2; do
  for n in 16 20 24; do CRESCENT_GPU_LIB=$L $B --inflight $n 2>/dev/null | python tools/line_value.py "host witness, $n in flight (16 hardware queues)"; done
  CRESCENT_GPU_LIB=$L GPU_MAX_HW_QUEUES=24 $B --inflight 24 2>/dev/null | python tools/line_value.py "host witness, 24 in flight, 24 hardware queues"
done) 2>&1 | tee $O/more_in_flight.txt
