#!/bin/bash
# kernel stats of the level-launch row (schedule 0) under both forms of its hash workgroups
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
python $R/scripts/level_launch_probe.py 2>&1 | tee $OUT/r06c_level_probe.txt
cd /tmp && export TMPDIR=/tmp
for form in classic dual; do
  if [ $form = classic ]; then export GC_LEVEL_CLASSIC=1; else unset GC_LEVEL_CLASSIC; fi
  rm -rf /tmp/prof_$form
  rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_$form -o lv -- python $R/scripts/level_launch_probe.py one > /tmp/prof_$form.log 2>&1
  f=$(find /tmp/prof_$form -name '*kernel_stats.csv' | head -1)
  echo "== $form" >> $OUT/r06c_level_kernel_stats.txt
  head -8 "$f" | cut -c1-220 >> $OUT/r06c_level_kernel_stats.txt
  t=$(find /tmp/prof_$form -name '*kernel_trace.csv' | head -1)
  python $R/scripts/r06c_level_gaps.py "$t" >> $OUT/r06c_level_kernel_stats.txt
done
cat $OUT/r06c_level_kernel_stats.txt
