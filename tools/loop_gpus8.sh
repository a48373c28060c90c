# the eight-processes-on-one-GPU bench command of tests/test_gpu_host_and_ranks.py, N times, the device memory dirtied before each
set -u
N=${1:-10}; BYTE=${2:-0xA5}
fails=0
for i in $(seq $N); do
  python tools/dirty_vram.py $BYTE > /dev/null 2>&1
  if ! python bench.py --gpus 8 --backend gloo --allow-shared-gpu --steps 8 --warmup 2 --inflight 2 --blocks 3 --shape medium --no-host-witness --sharded-steps 6 --sharded-inflight 2 --sharded-stream 16 --no-check --leg-timeout 900 > /tmp/g8.out 2> /tmp/g8.err; then
    fails=$((fails+1)); echo "run $i FAILED"; grep "AssertionError" /tmp/g8.err | cut -c1-330
  fi
done
echo "dirty byte $BYTE: $fails of $N runs failed"
