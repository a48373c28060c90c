set -u
O=gpurun_out/r3p; mkdir -p $O
tools/profile_sq.sh $O/sq > /dev/null 2>&1
tools/profile_serial.sh $O/serial > /dev/null 2>&1
tools/timeline_lone_proof.sh $O/lone 1 > /dev/null 2>&1
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
ls $O/*
