set -u
O=gpurun_out/r05_ab; mkdir -p $O
export CRESCENT_GPU_LIB=$PWD/crescent-credentials_amd/libcrescent_gpu_tuning.so
B="python bench.py --steps 100 --no-sweep --no-cpu-baseline --no-check --no-host-witness"
run() { label="$1"; shift; env "$@" $B 2>>$O/err.log | python tools/line_value.py "$label"; }
(for i in 1 2 3; do
 run "accumulation in 256-thread workgroups (shipped)" X=1
 run "accumulation in 64-thread workgroups" CG_ACCUM_BLOCK=64
 run "accumulation in 128-thread workgroups" CG_ACCUM_BLOCK=128
done) 2>&1 | tee $O/accum_block.txt
