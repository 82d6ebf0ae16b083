#!/bin/bash
# Run on the GPU box (through gpurun): N rank processes of bench.py on ONE GPU with tests/standin_rccl under gc_comm_* (RCCL
# refuses two ranks on one device).  Shows the N > 1 path end to end at config 4's shape; the gather times are the stand-in's
# (synchronous, through shared memory), and N processes share one GPU: the value is NOT a scaling number.
# usage: scripts/ranks_on_one_gpu.sh <N> [bench.py arguments]
set -u
N=${1:-2}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ranks_one_gpu
mkdir -p $OUT
SO=/tmp/librccl_standin.so
hipcc -shared -fPIC -O2 -o $SO $REPO/tests/standin_rccl/standin_rccl.cpp || exit 1
RDV=$(mktemp -d)
export GC_LAUNCH_NONCE=$(date +%s%N)p$$  # (one value per launch, in the name of the communicator-id file: mpc_amd/dist.py)
export WORLD_SIZE=$N MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 TORCHELASTIC_RUN_ID=one_gpu_$$ GC_RENDEZVOUS_DIR=$RDV GC_RCCL_PATH=$SO GC_BENCH_DEVICE=0
pids=()
for r in $(seq 0 $((N - 1))); do
    RANK=$r LOCAL_RANK=$r python $REPO/bench.py --gpus $N --no-cpu-baseline "$@" > $OUT/rank$r.out 2> $OUT/rank$r.err &
    pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
echo "exit $rc"
cat $OUT/rank0.out
tail -2 $OUT/rank1.err
