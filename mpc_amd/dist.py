"""Multi-GPU host logic: independent circuit instances shard embarrassingly across ranks (one process per GPU);
the only exchange is a terminal all-gather of the decoded output bits / output labels.

The data-path collective is gc_comm_allgather of libgcengine.so (ncclAllGather of RCCL over xGMI, called from the
C ABI — what a Go host binds).  The ranks of bench.py import no torch at all: `python -m torch.distributed.run` is only
the process launcher (RANK / LOCAL_RANK / WORLD_SIZE in the environment), the 128-byte ncclUniqueId travels from rank 0
to the other ranks of the node through a file (exchange_unique_id_file — the job a Go host does over its own
p2p.Conn), barrier and max-over-ranks timing are gc_comm_* calls.  In the CPU tests a gloo group stands in as the
gather transport (tests/gloo_transport.py: GlooGather) so that the sharding / padding / reassembly logic, the bench's step
loop (StepLoop, run_timed) and bench.py's own rank entry (on tests/stub_engine.py) are exercised with world_size 2 and 3
without a GPU."""
import os
import time

import numpy as np


def shard_range(total, rank, world):
    """contiguous instance range [lo, hi) of `rank` when `total` instances are split over `world` ranks"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rows(total, world):
    """rows every rank contributes to the gather: the largest shard (smaller shards are zero-padded, the
    collective needs equal sizes on all ranks)"""
    return -(-total // world) if world else 0


def reassemble(gathered, total, world):
    """[world, shard_rows, ...] as gathered -> [total, ...] in instance order, padding dropped"""
    parts = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        parts.append(gathered[r][: hi - lo])
    return np.concatenate(parts, axis=0) if parts else gathered.reshape((0,) + gathered.shape[2:])


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_control():
    """CPU control plane of a torch.distributed.run launch (gloo): rendezvous only, no GPU collective"""
    import torch.distributed as dist

    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    return rank, local_rank, world


def exchange_unique_id(make_id, rank, world):
    """rank 0 draws the communicator id (make_id() -> bytes), every rank returns it"""
    if world == 1:
        return make_id()
    import torch.distributed as dist

    box = [make_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def _rendezvous_path(directory=None):
    """The file of this launch: in GC_RENDEZVOUS_DIR (a directory the launcher made for the job) if set, else in a
    directory of this user's own under TMPDIR (created 0700, refused if somebody else owns it); the name carries the
    launcher's pid — the common parent of the ranks —, MASTER_PORT and the launcher's run id, so neither concurrent nor
    earlier launches collide."""
    directory = directory or os.environ.get("GC_RENDEZVOUS_DIR")
    if not directory:
        directory = os.path.join(os.environ.get("TMPDIR") or "/tmp", "gc_comm.%d" % os.getuid())
        try:
            os.mkdir(directory, 0o700)
        except FileExistsError:
            pass
        st = os.lstat(directory)
        import stat as _stat
        if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise PermissionError("rendezvous directory %s is not a private directory of this user" % directory)
    # GC_LAUNCH_NONCE: a value the launcher draws per launch and hands to every rank (scripts/ranks_on_one_gpu.sh does): launches
    # from ONE shell share the parent pid and the port, and a run that crashed a moment ago may have left its file behind
    nonce = os.environ.get("GC_LAUNCH_NONCE") or (os.environ.get("TORCHELASTIC_RUN_ID", "") + "r" + os.environ.get("TORCHELASTIC_RESTART_COUNT", ""))
    nonce = "".join(ch for ch in nonce if ch.isalnum())[:48]
    return os.path.join(directory, "gc_comm_id.%d.%s.%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"), nonce or "x"))


def exchange_unique_id_file(make_id, rank, world, timeout=180.0, directory=None):
    """The same hand-over without any torch: the ranks of ONE node share a file.  Rank 0 removes whatever an earlier launch
    left under the name, draws the id and publishes it atomically (exclusive create of a temporary, no symlink followed,
    then rename); the others poll for a regular file of this user that is not older than their own start, for at most
    `timeout` seconds (TimeoutError).  The readers remove nothing."""
    if world == 1:
        return make_id()
    name = _rendezvous_path(directory)
    if rank == 0:
        for stale in (name, name + ".tmp"):
            try:
                os.unlink(stale)
            except OSError:
                pass
        uid = make_id()
        fd = os.open(name + ".tmp", os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(uid)
        os.replace(name + ".tmp", name)
        return uid
    t0 = time.monotonic()
    start = time.time()
    while True:
        try:
            fd = os.open(name, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
            try:
                st = os.fstat(fd)
                import stat as _stat
                # rank 0 starts with this process (same launcher): a file older than this rank by more than the launcher's
                # spread is a leftover (GC_LAUNCH_NONCE in the name rules leftovers out altogether)
                if _stat.S_ISREG(st.st_mode) and st.st_uid == os.getuid() and st.st_size > 0 and st.st_mtime >= start - 30:
                    return os.read(fd, 4096)
            finally:
                os.close(fd)
        except OSError:
            pass
        if time.monotonic() - t0 > timeout:
            raise TimeoutError("rank %d: no communicator id from rank 0 within %.0f s (%s)" % (rank, timeout, name))
        time.sleep(0.01)


def cleanup_unique_id_file(rank, directory=None):
    """rank 0, after the communicator exists on every rank (i.e. after a barrier)"""
    if rank != 0:
        return
    try:
        os.unlink(_rendezvous_path(directory))
    except OSError:
        pass


class Watchdog:
    """Bounds a stage of a multi-rank run from OUTSIDE the main thread: ncclCommInitRank and the collectives are C calls that
    never return when a peer is missing, so a timer thread ends the process instead — on_timeout(message) first (rank 0
    prints its one JSON error line there), then os._exit(3).  arm() re-arms for the next stage, disarm() when done.  The same
    thread catches the launcher's SIGTERM (sent to the surviving ranks when one rank fails): exit code 4 after the line."""

    def __init__(self, on_timeout, catch_sigterm=True):
        import signal
        import threading
        self._on_timeout = on_timeout
        self._lock = threading.Lock()
        self._deadline, self._what = None, ""
        # The launcher ends the surviving ranks with SIGTERM when one rank fails.  A Python handler would only run once the
        # main thread leaves the C call it may be stuck in, so the signal is BLOCKED here (the timer thread inherits the mask)
        # and the timer thread waits for it synchronously: the rank says why it ends (rank 0: its one error line) and leaves.
        self._sigs = None
        if catch_sigterm and hasattr(signal, "pthread_sigmask") and hasattr(signal, "sigtimedwait"):
            signal.pthread_sigmask(signal.SIG_BLOCK, {signal.SIGTERM})
            self._sigs = {signal.SIGTERM}
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def arm(self, seconds, what):
        with self._lock:
            self._deadline, self._what = time.monotonic() + float(seconds), what

    def disarm(self):
        with self._lock:
            self._deadline = None

    def _run(self):
        import signal
        import sys
        while True:
            if self._sigs:
                if signal.sigtimedwait(self._sigs, 0.05) is not None:
                    try:
                        sys.stderr.write("bench watchdog: terminated by the launcher (another rank failed)\n")
                        self._on_timeout("terminated by the launcher: another rank failed (its message is on stderr)")
                        sys.stdout.flush()
                    finally:
                        os._exit(4)
            else:
                time.sleep(0.05)
            with self._lock:
                late = self._deadline is not None and time.monotonic() > self._deadline
                what = self._what
            if late:
                try:
                    sys.stderr.write("bench watchdog: %s\n" % what)
                    self._on_timeout(what)
                    sys.stdout.flush()
                finally:
                    os._exit(3)


def open_comm(ctx, rank, world, timeout=180.0, engine=None):
    """the RCCL communicator of this rank's gc_ctx (gc_comm_init_rank); no torch involved.  The id hand-over waits at most
    `timeout` seconds; ncclCommInitRank itself is bounded by the caller's Watchdog."""
    if engine is None:
        from . import engine

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: the only mode the host driver supports
    uid = exchange_unique_id_file(engine.comm_unique_id, rank, world, timeout=timeout)
    comm = engine.Comm(ctx, uid, world, rank)
    if world > 1:
        comm.barrier()
        cleanup_unique_id_file(rank)
    return comm


class StepLoop:
    """The step / gather bookkeeping of bench.py, free of any device type so that the CPU suite runs it at world 2 and 3.

    The decoded outputs of `slots` steps are collected in ONE accumulator and gathered with ONE all-gather (fewer, larger
    collectives); a tail of fewer than `slots` steps is flushed before the clock stops.
      launch(j)      enqueue one step of the hot path whose outputs land in slot j of the accumulator
      gather(nfresh) enqueue the all-gather of the accumulator; the first nfresh slots hold outputs that no earlier
                     gather carried (None: one rank, nothing to gather)
    Invariant checked by the tests: after flush() every step's outputs were gathered exactly once."""

    def __init__(self, slots, launch, gather=None):
        self.slots = max(1, int(slots)) if gather is not None else 1
        self.launch, self.gather = launch, gather
        self.pos = 0            # next free slot
        self.steps_done = 0
        self.steps_gathered = 0
        self.gathers = 0

    def step(self):
        self.launch(self.pos)
        self.pos += 1
        self.steps_done += 1
        if self.pos == self.slots:
            self._gather()

    def _gather(self):
        if self.gather is not None and self.pos:
            self.gather(self.pos)
            self.gathers += 1
        self.steps_gathered += self.pos
        self.pos = 0

    def flush(self):
        """outputs of the steps since the last full accumulator"""
        if self.pos:
            self._gather()


def run_timed(loop, fence, steps, warmup, allreduce_max=None, clock=time.perf_counter):
    """W untimed steps, then EXACTLY `steps` steps between two fences (barrier + device sync on both sides); the tail
    gather is inside the timed region; the job's time is the slowest rank's."""
    for _ in range(warmup):
        loop.step()
    loop.flush()
    fence()
    t0 = clock()
    for _ in range(steps):
        loop.step()
    loop.flush()
    fence()
    elapsed = clock() - t0
    loop.local_elapsed = elapsed  # (this rank's own time; bench.py reports per-rank values beside the job's)
    if allreduce_max is not None:
        elapsed = allreduce_max(elapsed)
    return elapsed


def gather_sharded(transport, local_rows, total, world):
    """One rank's rows of a [total, ...] result (its shard_range) -> the whole result on every rank.
    transport.allgather_host(padded [shard_rows, ...]) -> [world, shard_rows, ...]"""
    per = shard_rows(total, world)
    local_rows = np.asarray(local_rows)
    pad = np.zeros((per,) + local_rows.shape[1:], local_rows.dtype)
    pad[: len(local_rows)] = local_rows
    return reassemble(transport.allgather_host(pad), total, world)
