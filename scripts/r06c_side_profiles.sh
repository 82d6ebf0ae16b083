#!/bin/bash
# kernel stats of the side rows on the shipped tree: the streamed programs and config 3 (rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash scripts/profile_stream.sh r06c "ed25519like:1024 ssa23:64 mixed:64" > gpurun_out/r06c_side_profiles.log 2>&1
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d /tmp/kt_c3 -o kt -- python $R/scripts/bench_config3.py > $R/gpurun_out/prof_r06c/bench_config3.log 2>&1
find /tmp/kt_c3 -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof_r06c/config3_kernel_stats.csv \;
tail -1 $R/gpurun_out/prof_r06c/bench_config3.log | cut -c1-300
head -14 $R/gpurun_out/prof_r06c/config3_kernel_stats.csv | cut -c1-200
ls $R/gpurun_out/prof_r06c
