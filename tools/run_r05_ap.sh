set -u
O=gpurun_out/r05_ap; mkdir -p $O
python tools/soak.py 10000 16 > $O/soak.txt 2>&1; tail -1 $O/soak.txt
