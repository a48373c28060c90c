set -u
O=gpurun_out/r05_an; mkdir -p $O
P=$PWD/crescent-credentials_amd/libcrescent_gpu_prev.so
B="python bench.py --steps 100 --no-sweep --no-cpu-baseline --no-host-witness"
(for i in 1 2 3; do
  CRESCENT_GPU_LIB=$P $B 2>>$O/err.log | python tools/line_value.py "a pad word per 16 elements (before)"
  $B 2>>$O/err.log | python tools/line_value.py "a pad word per 32 elements"
done) 2>&1 | tee $O/lds_pad.txt
grep -c "verifies: True" $O/err.log
# LDS counters again
ROOT=$PWD; OO=$PWD/$O
(export TMPDIR=/tmp; cd /tmp
FLAGS="--steps 4 --warmup 2 --blocks 1 --no-sweep --no-cpu-baseline --no-host-witness --no-check --no-clock-probe --inflight 1 --mode throughput --witness device"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS -d $OO/s1 -o p -- python3 $ROOT/bench.py $FLAGS > $OO/s1.line.json 2> $OO/s1.log
DB=$(find $OO/s1 -name '*.db' | head -1)
python3 $ROOT/tools/rocpd_counters.py "$DB" | head -8 > $OO/lds_counters.md; rm -rf $OO/s1)
cut -c1-200 $O/lds_counters.md
# the full validation of the tree
(time python -m pytest tests -m gpu -q) > $O/gputests.log 2>&1; tail -3 $O/gputests.log
tools/profile_pmc.sh $O/pmc "rs256-sd/gates/bits=0.90" > /dev/null 2>&1
cp $O/pmc/pmc_counters.json profiles/pmc_counters.json && cp profiles/pmc_counters.json $O/pmc_counters.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err; echo
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
for f in bench_default bench_driver_args; do python tools/line_value.py $f < $O/$f.json; done
python -c "import json; d=json.load(open('$O/bench_default.json')); print(d['roofline_valu']['counters'], d['roofline']['traffic'])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
