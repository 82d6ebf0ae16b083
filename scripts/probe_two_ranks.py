"""How far does the N > 1 path get on a box with ONE GPU?  `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2
--master-addr 127.0.0.1 --master-port P scripts/probe_two_ranks.py` puts both ranks on device 0.  RCCL refuses that
communicator — but only AFTER its bootstrap: with NCCL_DEBUG=WARN both ranks print "Duplicate GPU detected : rank 0 and rank 1
both on CUDA device ...", i.e. the launcher environment, the hand-over of the ncclUniqueId through the file and
gc_comm_init_rank with world 2 in two processes all worked (profiles/r03_two_rank_probe.txt).  On a box with two GPUs change
Context(0) to Context(LOCAL_RANK) and it gathers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_amd import dist as gdist, engine
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
ctx = engine.Context(0)
try:
    comm = gdist.open_comm(ctx, rank, world)
    a = ctx.to_device(np.full(1024, rank + 1, np.uint8))
    b = ctx.zeros((world, 1024))
    comm.allgather(a, b, 1024)
    ctx.sync()
    print("rank", rank, "gathered", b.numpy()[:, 0].tolist(), "max", comm.allreduce_max(float(rank)))
    comm.barrier(); comm.close()
except Exception as e:
    print("rank", rank, "FAILED:", str(e)[:300])
ctx.close()
