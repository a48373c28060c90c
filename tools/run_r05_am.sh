set -u
O=$PWD/gpurun_out/r05_am; mkdir -p $O
ROOT=$PWD
export TMPDIR=/tmp
cd /tmp
FLAGS="--steps 4 --warmup 2 --blocks 1 --no-sweep --no-cpu-baseline --no-host-witness --no-check --no-clock-probe --inflight 1 --mode throughput --witness device"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS -d $O/s1 -o p -- python3 $ROOT/bench.py $FLAGS > $O/s1.line.json 2> $O/s1.log
DB=$(find $O/s1 -name '*.db' | head -1)
if [ -n "$DB" ]; then python3 $ROOT/tools/rocpd_counters.py "$DB" | head -30 > $O/lds_counters.md; else tail -5 $O/s1.log > $O/lds_counters.md; fi
rm -rf $O/s1
cat $O/lds_counters.md | cut -c1-220
