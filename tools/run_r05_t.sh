set -u
O=gpurun_out/r05_t; mkdir -p $O
L=$PWD/crescent-credentials_amd/libcrescent_gpu_cap32.so
B="python bench.py --steps 100 --no-sweep --no-cpu-baseline --no-check --no-host-witness"
(for i in 1 2 3; do
  for n in 16 18 20 22; do CRESCENT_GPU_LIB=$L $B --inflight $n 2>/dev/null | python tools/line_value.py "host witness, $n in flight"; done
done) 2>&1 | tee $O/more_in_flight.txt
