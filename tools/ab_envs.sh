# (the environment switches exist in the tuning build only: `python crescent-credentials_amd/build.py --tuning`)
# usage: tools/ab_envs.sh REPS "bench args" "ENV1=a ENV2=b" "ENV1=c" ...: round-robin over environment settings ("-" = none)
set -u
N=$1; ARGS=$2; shift; shift
for i in $(seq $N); do
  for e in "$@"; do
    if [ "$e" = "-" ]; then E=""; else E="$e"; fi
    env $E CRESCENT_GPU_LIB=crescent-credentials_amd/libcrescent_gpu_tuning.so python bench.py --witness device --steps 100 --headline-only --no-sweep --no-cpu-baseline --no-check $ARGS 2>/dev/null | python tools/line_value.py "[$e]"
  done
done
