#!/bin/bash
# round 2: collect every number and profile the docs quote (one gpurun call)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02_collect
mkdir -p $OUT
cd $REPO
python bench.py > $OUT/bench_b1024.json 2> $OUT/bench_b1024.err; echo "bench rc=$?"
python bench.py --key-bytes 16 --no-cpu-baseline --no-iknp --no-host-api --no-stream --no-config3 > $OUT/bench_b1024_key16.json 2>/dev/null; echo "bench key16 rc=$?"
python bench.py --batch 8192 --steps 40 --warmup 5 --no-cpu-baseline --no-iknp --no-host-api --no-stream --no-config3 > $OUT/bench_b8192.json 2>/dev/null; echo "bench 8192 rc=$?"
python bench.py --force-collective --no-cpu-baseline --no-iknp --no-host-api --no-stream --no-config3 > $OUT/bench_b1024_rccl_1rank.json 2>/dev/null; echo "bench coll rc=$?"
python bench.py --circuit tests/golden/sha256xor.gcf --batch 256 --steps 50 --warmup 5 --no-cpu-baseline --no-iknp --no-host-api --no-stream --no-config3 > $OUT/bench_sha256xor_b256.json 2>/dev/null; echo "bench sha rc=$?"
python scripts/bench_config3.py > $OUT/config3.json 2> $OUT/config3.err; echo "config3 rc=$?"
python scripts/bench_stream.py 130000000 > $OUT/stream_1e8.json 2> $OUT/stream.err; echo "stream rc=$?"
python scripts/bench_host_api.py 1024 > $OUT/host_api.json 2>/dev/null; echo "host rc=$?"
timeout 300 tools/issue_rate_ubench > $OUT/issue_rate_ubench.txt 2>&1; echo "ubench rc=$?"
bash scripts/profile.sh r02_flat > $OUT/profile.log 2>&1; echo "profile rc=$?"
bash scripts/profile_sq.sh > $OUT/profile_sq.log 2>&1; echo "profile sq rc=$?"
cp gpurun_out/prof_sq/summary.txt $OUT/sq_summary.txt 2>/dev/null
for f in bench_b1024 bench_b1024_key16 bench_b8192 bench_b1024_rccl_1rank bench_sha256xor_b256; do python - <<P
import json
d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1])
print('$f', '%.4g AND/s'%d['value'], 'g %.3f e %.3f'%(d['garble_ms'],d['eval_ms']), 'frac',round(d['roofline']['frac'],3))
P
done
tail -3 $OUT/config3.json | cut -c1-600; tail -1 $OUT/stream_1e8.json | cut -c1-700; cat $OUT/host_api.json | cut -c1-900
