# After the last kernel-source change of a round: re-take the PMC passes behind profiles/pmc_counters.json (its entries are
# stamped with the source fingerprint, and bench.py nulls what it derives from a stale one), then the default line and the
# line with the driver's arguments on the same box.  usage: tools/refresh_counters.sh [out-dir]
set -u
O=${1:-gpurun_out/refresh}; mkdir -p $O
(time python -m pytest tests -m gpu -q) > $O/gputests.log 2>&1; tail -2 $O/gputests.log
tools/profile_pmc.sh $O/pmc "rs256-sd/gates/bits=0.90" > /dev/null 2>&1
cp $O/pmc/pmc_counters.json profiles/pmc_counters.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 200 $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
python - <<PY
import json
for f in ("bench_default", "bench_driver_args"):
    d = json.load(open("$O/" + f + ".json"))
    print(f, d["value"], d["timing"]["median_block"]["spread_pct"], d["roofline_valu"]["frac"], d["roofline_valu"]["counters"]["current"], d["host_witness"]["pinned"]["proofs_per_s"], d["cpu_baseline"]["gpu_over_cpu"], d["proof_verifies"], d["key_check"]["ok"])
PY
