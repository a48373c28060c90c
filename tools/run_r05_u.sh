set -u
O=gpurun_out/r05_u; mkdir -p $O
P=$PWD/crescent-credentials_amd/libcrescent_gpu_prev.so
B="python bench.py --steps 100 --no-sweep --no-cpu-baseline --no-check --no-host-witness"
(for i in 1 2 3; do
  CRESCENT_GPU_LIB=$P GPU_MAX_HW_QUEUES=16 $B 2>/dev/null | python tools/line_value.py "host witness, a copy stream per upload buffer, 16 queues (before)"
  GPU_MAX_HW_QUEUES=16 $B 2>/dev/null | python tools/line_value.py "host witness, four copy streams, 16 queues"
  GPU_MAX_HW_QUEUES=20 $B 2>/dev/null | python tools/line_value.py "host witness, four copy streams, 20 queues"
  GPU_MAX_HW_QUEUES=20 $B --witness device 2>/dev/null | python tools/line_value.py "device witness, 20 queues"
  GPU_MAX_HW_QUEUES=16 $B --witness device 2>/dev/null | python tools/line_value.py "device witness, 16 queues"
done) 2>&1 | tee $O/copy_streams.txt
