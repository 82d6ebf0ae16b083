#!/bin/bash
# threads per workgroup of the flattened kernels on deep, narrow circuits: parity under each forced size, then the rows they bound
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
for t in 256 512; do
  echo "== parity, GC_FLAT_THREADS=$t"
  GC_FLAT_THREADS=$t python -m pytest tests/test_gpu_garble_eval.py tests/test_gpu_go_transcript.py tests/test_gpu_config3.py -x -q -m gpu 2>&1 | tail -3
done
echo "== parity, default choice"
python -m pytest tests/test_gpu_garble_eval.py tests/test_gpu_go_transcript.py tests/test_gpu_config3.py -x -q -m gpu 2>&1 | tail -3
for t in 1024 512 256 auto; do
  if [ $t = auto ]; then unset GC_FLAT_THREADS; else export GC_FLAT_THREADS=$t; fi
  echo "== config3, threads $t"; python scripts/bench_config3.py
  echo "== and_chain_10000 / add64 x 256, threads $t"; python scripts/r06c_tf_rows.py
done 2>&1 | tee $OUT/r06c_tf_probe.txt
