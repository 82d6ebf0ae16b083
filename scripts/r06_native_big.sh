#!/bin/bash
# round 6: reproduce BENCH_r05's native_host.big130 failure (tools/stream_driver, config 5's size) with the child's stderr kept
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r06a
free -g > gpurun_out/r06a/mem.txt; nproc >> gpurun_out/r06a/mem.txt; ulimit -a >> gpurun_out/r06a/mem.txt
cat /sys/fs/cgroup/memory.max >> gpurun_out/r06a/mem.txt 2>&1
python scripts/write_program.py big130 /tmp/big130.bin 2 > gpurun_out/r06a/write.log 2>&1
for i in 1 2 3; do
  ( timeout 600 tools/stream_driver /tmp/big130.bin ) > gpurun_out/r06a/native_big130_$i.out 2> gpurun_out/r06a/native_big130_$i.err
  echo "rc=$?" >> gpurun_out/r06a/native_big130_$i.err
done
python scripts/write_program.py mixed /tmp/mixed.bin 64 >> gpurun_out/r06a/write.log 2>&1
( timeout 600 tools/stream_driver /tmp/mixed.bin ) > gpurun_out/r06a/native_mixed.out 2> gpurun_out/r06a/native_mixed.err
echo "rc=$?" >> gpurun_out/r06a/native_mixed.err
tail -n 3 gpurun_out/r06a/*.err gpurun_out/r06a/*.out
