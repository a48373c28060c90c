for rep in 1 2; do for m in 0 1; do for s in 1 8; do
CG_LATENCY_MODE=$m python tools/probe_latency.py $s 2>/dev/null | cut -c1-60 | sed "s/^/latency_mode=$m /"
done; done; done
O=gpurun_out/final
tools/profile_serial.sh $O/serial > /dev/null 2>&1
tools/profile_pmc.sh $O/pmc "rs256-sd/gates/bits=0.90" > /dev/null 2>&1
tail -3 $O/serial/efficiency.md; grep valu_wave $O/pmc/pmc_counters.json
