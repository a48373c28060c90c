set -u
O=gpurun_out/r05_w; mkdir -p $O
export CRESCENT_GPU_LIB=$PWD/crescent-credentials_amd/libcrescent_gpu_tuning.so
B="python bench.py --steps 100 --no-sweep --no-cpu-baseline --no-check --no-host-witness"
run() { label="$1"; shift; env "$@" $B 2>/dev/null | python tools/line_value.py "$label"; }
(run "full" X=1
 run "without l" CG_KNOCK=1
 run "without a" CG_KNOCK=2
 run "without b1 (b2 groups its own entries)" CG_KNOCK=4
 run "without b2" CG_KNOCK=8
 run "full" X=1
 run "without the witness map" CG_KNOCK=16
 run "without h" CG_KNOCK=32
 run "without l, a, b1" CG_KNOCK=7
 run "without the bucket reductions" CG_KNOCK_TAIL=1
 run "without the combine levels" CG_KNOCK_COMBINE=1
 run "full" X=1
 run "without accumulation and combine levels" CG_KNOCK_ACCUM=1
 run "without accumulation, combine levels, bucket reductions (grouping + witness map left)" CG_KNOCK_ACCUM=1 CG_KNOCK_TAIL=1
 run "h MSM alone (scalars reused)" CG_KNOCK=31
 run "witness map alone" CG_KNOCK=47
 run "b2 alone" CG_KNOCK=55
 run "l, a, b1 alone" CG_KNOCK=56
 run "full" X=1
) 2>&1 | tee $O/knock_outs.txt
