set -u
O=gpurun_out/r05_ao; mkdir -p $O
tools/profile_serial.sh $O/serial > /dev/null 2>&1; tail -1 $O/serial/efficiency.md
tools/profile_pipelined.sh $O/pipelined > /dev/null 2>&1; head -8 $O/pipelined/kernel_stats.md | cut -c1-150
python tools/line_value.py traced < $O/pipelined/trace_line.json
