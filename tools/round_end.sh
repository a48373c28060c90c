# end-of-round artefacts on one box: full GPU test suite, default bench line and the line with the driver's arguments, serial
# efficiency + per-launch accumulation table, pipelined kernel stats + where the chip's capacity goes in the pipelined run,
# SQ counter passes (serial launches, and ONE pass over the default 16-in-flight run), the PMC passes behind
# profiles/pmc_counters.json, the 2-rank run as the driver invokes it (gloo data plane, both ranks on this one GPU), the same
# with --backend nccl (RCCL refuses two ranks on one device: the fallback path), lone-proof / shard latencies, a soak.
# usage: tools/round_end.sh [out-dir]
set -u
O=${1:-gpurun_out/final}
mkdir -p $O
(time python -m pytest tests -m gpu -q) > $O/gputests.log 2>&1; tail -3 $O/gputests.log
tools/profile_pmc.sh $O/pmc "rs256-sd/gates/bits=0.90" > /dev/null 2>&1
cp $O/pmc/pmc_counters.json profiles/pmc_counters.json          # the lines below then carry the counter-derived figures
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
tools/profile_serial.sh $O/serial > /dev/null 2>&1
tools/profile_pipeline_capacity.sh $O/pipe > /dev/null 2>&1
tools/profile_sq.sh $O/sq > /dev/null 2>&1
# one counter pass over the PIPELINED run (program directly after --; kernels are serialised by the counter collection,
# so these are stand-alone figures of the launches the pipelined run makes, segment lengths included)
( export TMPDIR=/tmp; cd /tmp; rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OLDPWD/$O/sqpipe -o p -- python3 $OLDPWD/bench.py --steps 20 --warmup 4 --blocks 2 --headline-only --no-sweep --no-cpu-baseline --no-host-witness --no-check --no-clock-probe > $OLDPWD/$O/sqpipe.line.json 2> $OLDPWD/$O/sqpipe.log )
DB=$(find $O/sqpipe -name '*.db' | head -1); [ -n "$DB" ] && python3 tools/rocpd_counters.py "$DB" $O/sq_counters_pipelined_run.md > /dev/null; rm -rf $O/sqpipe
python bench.py --gpus 2 --backend gloo --steps 24 --warmup 4 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
python bench.py --gpus 2 --backend nccl --allow-shared-gpu --steps 24 --warmup 4 --no-host-witness > $O/bench_2rank_nccl_one_gpu.json 2> $O/bench_2rank_nccl_one_gpu.err
for s in 1 2 4 8; do python tools/probe_latency.py $s 2>/dev/null | cut -c1-330; done > $O/latency.txt
python tools/soak.py 12000 16 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
# round 6: the clock the chip holds under each kind of load, the issue rates in shader cycles, the cold start through the C caller
python tools/probe_clock_vs_load.py 3 > $O/clock_vs_load.txt 2>&1
tools/ubench/valu_rates --json > $O/valu_rates.json 2>&1
tools/ubench/f29_rates --cycles > $O/f29_cycles.json 2>&1
python -m pytest tests/test_gpu_cold_start.py -q -s -k "files_to_client_state" 2>&1 | grep "cold start" > $O/cold_start_c_caller.txt
# every bench line of this run goes through tools/line_value.py: a line marked `incomplete` (printed by the watchdog) is refused
for f in $O/bench_*.json; do python tools/line_value.py $(basename $f .json) < $f; done | tee $O/lines_checked.txt
ls -la $O $O/serial $O/pipe $O/pmc $O/sq
