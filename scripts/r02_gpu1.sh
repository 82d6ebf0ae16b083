#!/bin/bash
# round 2, GPU call 1: new parity tests (config 4, stream fixes), full GPU suite, bench (plain and with the RCCL
# gather forced on one rank), instruction issue-rate micro-benchmark
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02_1
mkdir -p $OUT
cd $REPO
python -m pytest tests/test_gpu_config4.py tests/test_gpu_stream.py -x -q -m gpu > $OUT/pytest_new.log 2>&1; echo "new tests rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/pytest_new.log
python -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; echo "all gpu tests rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_all.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
python bench.py --force-collective --no-cpu-baseline --no-iknp > $OUT/bench_coll.json 2> $OUT/bench_coll.err; echo "bench coll rc=$?" | tee -a $OUT/summary.txt
python bench.py --batch 8192 --steps 40 --warmup 5 --no-cpu-baseline --no-iknp > $OUT/bench_8192.json 2> $OUT/bench_8192.err; echo "bench 8192 rc=$?" | tee -a $OUT/summary.txt
timeout 300 tools/issue_rate_ubench > $OUT/issue_rate.txt 2>&1; echo "ubench rc=$?" | tee -a $OUT/summary.txt
cat $OUT/issue_rate.txt
cat $OUT/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','garble_ms','eval_ms')}); print(d['roofline']); print(d['aes_core'])"
tail -2 $OUT/bench_coll.json | cut -c1-400
tail -2 $OUT/bench_8192.json | cut -c1-400
