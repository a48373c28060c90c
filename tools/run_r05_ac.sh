set -u
O=gpurun_out/r05_ac; mkdir -p $O
export CRESCENT_GPU_LIB=$PWD/crescent-credentials_amd/libcrescent_gpu_tuning.so
B="python bench.py --steps 100 --no-sweep --no-cpu-baseline --no-host-witness"
run() { label="$1"; shift; env "$@" $B 2>>$O/err.log | python tools/line_value.py "$label"; }
(for i in 1 2 3; do
 run "combine / bucket reduction waves one per workgroup (shipped)" X=1
 run "G1: four waves per workgroup" CG_TAIL_BLOCK=256
 run "G1 and G2: four waves per workgroup" CG_TAIL_BLOCK=256 CG_TAIL_BLOCK_G2=256
done) 2>&1 | tee $O/tail_block.txt
grep -c "verifies: True" $O/err.log
