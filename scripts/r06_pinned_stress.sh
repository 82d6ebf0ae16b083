#!/bin/bash
# round 6: the pinned read-buffer pass of tools/stream_driver on big130, many times over (BENCH_r05's native_host.big130 failure:
# "a cooperative one-instance pass lost a workgroup and could not be repeated on the device", once in ~10 runs)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/${1:-r06c}; mkdir -p $OUT
python scripts/write_program.py big130 /tmp/big130.bin 2
for i in 1 2 3 4; do
  GC_DRIVER_PINNED_PASSES=${2:-30} timeout 900 tools/stream_driver /tmp/big130.bin > $OUT/stress_$i.out 2> $OUT/stress_$i.err
  echo "rc=$?" >> $OUT/stress_$i.err
done
tail -n 4 $OUT/stress_*.err; cut -c1-300 $OUT/stress_*.out
