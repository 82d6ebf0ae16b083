#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02_3
mkdir -p $OUT
cd $REPO
python -m pytest tests/test_gpu_garble_eval.py -x -q -m gpu -k "pinned or host or concurrent" > $OUT/pytest.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/pytest.log
python scripts/bench_host_api.py 1024 > $OUT/host_api.json 2> $OUT/host_api.err; echo "host api rc=$?"; cat $OUT/host_api.json; tail -3 $OUT/host_api.err
timeout 300 tools/issue_rate_ubench > $OUT/issue_rate.txt 2>&1; echo "ubench rc=$?"
grep -E "VMEM|vgpr shift|v_and_b32|v_or_b32|literal|clock" $OUT/issue_rate.txt
