set -u
export TMPDIR=/tmp
R=$PWD
cd /tmp
for mode in new plain; do
  if [ $mode = plain ]; then export CG_SELL_PLAIN=1; else unset CG_SELL_PLAIN; fi
  CG_SERIAL_STREAMS=1 CG_LATENCY_MODE=0 rocprofv3 --pmc SQ_INSTS_VALU -d $R/gpurun_out/r3o/$mode -o p -- python3 $R/bench.py --steps 6 --warmup 2 --blocks 1 --no-sweep --no-cpu-baseline --no-host-witness --no-clock-probe --inflight 1 > /dev/null 2>&1
  DB=$(find $R/gpurun_out/r3o/$mode -name '*.db' | head -1)
  echo "== $mode"; python3 $R/tools/rocpd_pmc.py $DB SQ_INSTS_VALU | grep -E "sell|total|TOTAL|ntt29_pass<0, 0>" | head -6
  rm -rf $R/gpurun_out/r3o/$mode
done
