// A stand-in for librccl.so, for ONE purpose: to let two rank processes of bench.py run the whole N > 1 path — launcher
// environment, id hand-over, gc_comm_init_rank, the gathers behind the decode on the ctx stream, barrier, max-over-ranks,
// the JSON line — on a box with ONE GPU, where RCCL itself refuses a communicator ("Duplicate GPU detected").  TEST
// INFRASTRUCTURE (tests/test_gpu_two_ranks.py selects it with GC_RCCL_PATH): it implements the ten entry points comm.cpp binds,
// for the ranks of one node, through a shared mapping of a file under TMPDIR named by the unique id (a file, not /dev/shm:
// containers keep that one small); every collective is synchronous
// (stream sync, device -> segment, barrier, segment -> device, barrier), so its TIMES mean nothing — what it shows is that
// everything around the collective library works with more than one rank.  Not RCCL, not a transport: no xGMI involved.
//
//   hipcc -shared -fPIC -O2 -o /tmp/librccl_standin.so tests/standin_rccl/standin_rccl.cpp
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {
constexpr size_t kSlot = (size_t)32 << 20;  // bytes a rank can contribute to one collective
constexpr int kWaitSeconds = 120;

struct Seg {
    std::atomic<uint32_t> attached, count, gen, left;
    uint32_t nranks;
    uint8_t pad[4096 - 5 * sizeof(uint32_t)];
    uint8_t data[1];
};

struct Comm {
    Seg *seg = nullptr;
    size_t bytes = 0;
    int rank = 0, nranks = 0;
    char name[256] = "";
};

bool wait_until(const std::atomic<uint32_t> &v, uint32_t not_equal_to_then_go, bool until_changes) {
    const auto t0 = std::chrono::steady_clock::now();
    while (until_changes ? v.load(std::memory_order_acquire) == not_equal_to_then_go
                         : v.load(std::memory_order_acquire) != not_equal_to_then_go) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(kWaitSeconds)) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    return true;
}

bool barrier(Comm *c) {
    Seg *s = c->seg;
    const uint32_t g = s->gen.load(std::memory_order_acquire);
    if (s->count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->nranks) {
        s->count.store(0, std::memory_order_relaxed);
        s->gen.fetch_add(1, std::memory_order_acq_rel);
        return true;
    }
    return wait_until(s->gen, g, true);
}

size_t size_of(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}
}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int *v) {
    if (v) *v = 1;  // (no RCCL release carries this number)
    return ncclSuccess;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    FILE *f = std::fopen("/dev/urandom", "rb");
    if (!f || std::fread(id->internal, 1, sizeof id->internal, f) != sizeof id->internal) {
        if (f) std::fclose(f);
        return ncclSystemError;
    }
    std::fclose(f);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || nranks > 16 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm *c = new Comm;
    c->rank = rank, c->nranks = nranks;
    char hex[25];
    for (int i = 0; i < 12; i++) std::snprintf(hex + 2 * i, 3, "%02x", (unsigned char)id.internal[i]);
    const char *tmp = std::getenv("TMPDIR");
    std::snprintf(c->name, sizeof c->name, "%s/gc_standin_%s", tmp && *tmp ? tmp : "/tmp", hex);
    c->bytes = offsetof(Seg, data) + kSlot * (size_t)nranks;
    const int fd = open(c->name, O_CREAT | O_RDWR | O_NOFOLLOW, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) {
        if (fd >= 0) close(fd);
        delete c;
        return ncclSystemError;
    }
    void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        delete c;
        return ncclSystemError;
    }
    c->seg = (Seg *)p;  // (a fresh segment is all zeros: the counters start at 0)
    c->seg->nranks = (uint32_t)nranks;
    c->seg->attached.fetch_add(1, std::memory_order_acq_rel);
    if (!wait_until(c->seg->attached, (uint32_t)nranks, false)) {  // every rank of the communicator has to show up
        munmap(p, c->bytes);
        unlink(c->name);
        delete c;
        return ncclSystemError;
    }
    *out = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *, int, const int *) { return ncclInvalidUsage; }  // (processes only)

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm *c = (Comm *)comm;
    if (!c) return ncclSuccess;
    const bool last = c->seg->left.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->nranks;
    munmap(c->seg, c->bytes);
    if (last) unlink(c->name);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream) {
    Comm *c = (Comm *)comm;
    const size_t bytes = count * size_of(type);
    if (!c || !size_of(type) || bytes > kSlot) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;  // everything enqueued before the collective
    if (hipMemcpy(c->seg->data + kSlot * (size_t)c->rank, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    for (int r = 0; r < c->nranks; r++)
        if (hipMemcpy((uint8_t *)recv + bytes * (size_t)r, c->seg->data + kSlot * (size_t)r, bytes, hipMemcpyHostToDevice) != hipSuccess)
            return ncclUnhandledCudaError;
    return barrier(c) ? ncclSuccess : ncclSystemError;  // (the slots are free again)
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
    Comm *c = (Comm *)comm;
    if (!c || type != ncclFloat64 || op != ncclMax || count * 8 > kSlot) return ncclInvalidArgument;  // what comm.cpp asks for
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    double *mine = (double *)(c->seg->data + kSlot * (size_t)c->rank);
    if (hipMemcpy(mine, send, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    double *res = new double[count];
    for (size_t i = 0; i < count; i++) {
        double m = ((double *)(c->seg->data))[i];
        for (int r = 1; r < c->nranks; r++) {
            const double v = ((double *)(c->seg->data + kSlot * (size_t)r))[i];
            m = v > m ? v : m;
        }
        res[i] = m;
    }
    const hipError_t e = hipMemcpy(recv, res, count * 8, hipMemcpyHostToDevice);
    delete[] res;
    if (e != hipSuccess) return ncclUnhandledCudaError;
    return barrier(c) ? ncclSuccess : ncclSystemError;
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "stand-in: HIP call failed";
        case ncclSystemError: return "stand-in: shared segment / a rank did not arrive";
        case ncclInvalidArgument: return "stand-in: invalid argument";
        case ncclInvalidUsage: return "stand-in: not supported";
        default: return "stand-in: error";
    }
}

}  // extern "C"
