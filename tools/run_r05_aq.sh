set -u
O=gpurun_out/r05_aq; mkdir -p $O
(for n in 1 2 4 8; do python tools/probe_latency.py $n 2>/dev/null | cut -c1-330; done) | tee $O/latency.txt
