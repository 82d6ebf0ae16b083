#!/bin/bash
# ONE command for the first box that shows more than one MI355X (the reference shards nothing —
# circuit/garble.go:253-278 is why instances are independent): the in-process N-device test (gc_comm_init_all + one all-gather,
# tests/test_gpu_config4.py::test_two_devices_sharded_gather), then bench.py at N = 1, 2, 4, 8 as the driver launches it (one
# rank per GPU over RCCL, 8 192 instances per rank = BASELINE config 4's share), every line kept, and a table: job value,
# per-rank min / max, scaling against the SAME RUN's N = 1 figure at 8 192 instances (`n1_batch8192` of the N = 1 line).
# usage: scripts/first_multi_gpu.sh [steps] [warmup]     (writes gpurun_out/multi/)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
STEPS=${1:-20}; WARMUP=${2:-5}
OUT=gpurun_out/multi; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0   # dmabuf IPC: RCCL across processes needs it on this driver
[ -f mpc_amd/csrc/libgcengine.so ] || python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
NDEV=$(python -c "from mpc_amd import engine; print(engine.device_count())" 2>/dev/null || echo 0)
echo "devices visible: $NDEV" | tee $OUT/devices.txt
rocm-smi --showtopo >> $OUT/devices.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_config4.py tests/test_gpu_two_ranks.py -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/tests.log
timeout 900 python bench.py --gpus 1 --steps $STEPS --warmup $WARMUP --no-stream --no-config3 --no-host-api > $OUT/n1.json 2> $OUT/n1.err; echo "N=1 rc=$?"
for N in 2 4 8; do
  [ "$N" -le "$NDEV" ] || continue
  timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
      bench.py --gpus $N --steps $STEPS --warmup $WARMUP > $OUT/n$N.json 2> $OUT/n$N.err
  echo "N=$N rc=$?"
done
python - "$OUT" <<'PY' | tee $OUT/summary.txt
import json, os, sys
out = sys.argv[1]
rows, base = [], None
for n in (1, 2, 4, 8):
    p = os.path.join(out, "n%d.json" % n)
    if not os.path.exists(p):
        continue
    lines = [l for l in open(p).read().splitlines() if l.startswith("{")]
    if not lines:
        rows.append((n, "no JSON line (see n%d.err)" % n)); continue
    j = json.loads(lines[-1])
    if j.get("error"):
        rows.append((n, "error at stage %s: %s" % (j.get("stage"), j["error"][:200]))); continue
    if n == 1:
        base = (j.get("n1_batch8192") or {}).get("and_gates_per_s")
        rows.append((1, "value %.3e AND/s at 1 024 instances; n1_batch8192 %s" % (j["value"], "%.3e" % base if base else "missing")))
        continue
    pr, c4 = j.get("per_rank", {}), j.get("config4", {})
    eff = (j["value"] / (n * base)) if base else None
    rows.append((n, "value %.3e AND/s (8 192 per rank), per rank min %.3e max %.3e, vs N x n1_batch8192: %s, ranks seen %s on %s distinct GPUs, "
                    "gather %.0f us (%s GB/s received per GPU), outputs ok %s" % (
                        j["value"], pr.get("min", 0), pr.get("max", 0), "%.3f" % eff if eff else "n/a", j.get("n_ranks_seen"),
                        c4.get("distinct_gpus"), c4.get("gather_us") or 0, "%.1f" % c4["gather_GBs_per_gpu_received"] if c4.get("gather_GBs_per_gpu_received") else "n/a",
                        c4.get("gathered_outputs_ok"))))
for n, t in rows:
    print("N=%d: %s" % (n, t))
PY
