// stream_lanes.cpp — the deep lanes of a ctx (stream_internal.h: DeepLanes): creating and PROBING the extra HIP streams.
#include "stream_internal.h"

namespace gcs {

namespace {
__global__ void k_lane_probe(uint32_t *flag, uint32_t *seen, unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();  // 100 MHz
    uint32_t v = 0;
    while (!(v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) && wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    *seen = v;
}
__global__ void k_lane_flag(uint32_t *flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}  // namespace

// the ctx's lanes, created and probed on first use (under ctx->mu)
void DeepLanes::setup_ctx(gc_ctx *ctx) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->lanes_state != 0) return;
    ctx->lanes_state = -1;
    const char *v = std::getenv("GC_STREAM_DEEP_LANES");
    int want = v && *v ? std::atoi(v) : (int)kDeepLanesDefault;
    if (want <= 0) return;
    want = std::min(want, 8);
    if (hipSetDevice(ctx->device) != hipSuccess) return;
    uint32_t *h = nullptr, *d_probe = nullptr;
    if (hipMalloc((void **)&d_probe, 2 * sizeof(uint32_t)) != hipSuccess ||
        hipHostMalloc((void **)&h, 2 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        if (d_probe) (void)hipFree(d_probe);
        return;
    }
    // does a kernel on `a` run while one on `b`, enqueued EARLIER, is still spinning?  (b's kernel waits up to 0.4 ms
    // for the flag a's kernel raises)
    auto beside = [&](hipStream_t b, hipStream_t a) -> bool {
        h[0] = h[1] = 0;
        if (hipMemcpyAsync(d_probe, h, 8, hipMemcpyHostToDevice, a) != hipSuccess || hipStreamSynchronize(a) != hipSuccess) return false;
        hipLaunchKernelGGL(k_lane_probe, dim3(1), dim3(1), 0, b, d_probe, d_probe + 1, 40000ull);
        hipLaunchKernelGGL(k_lane_flag, dim3(1), dim3(1), 0, a, d_probe);
        if (hipStreamSynchronize(b) != hipSuccess || hipStreamSynchronize(a) != hipSuccess) return false;
        if (hipMemcpy(h, d_probe, 8, hipMemcpyDeviceToHost) != hipSuccess) return false;
        return h[1] == 1;
    };
    const bool trace = std::getenv("GC_TRACE") != nullptr;
    for (int tries = 0; (int)ctx->lanes.size() < want && tries < want + 8; tries++) {
        hipStream_t c = nullptr;
        if (hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) break;
        bool ok = beside(c, ctx->stream);
        for (size_t i = 0; ok && i < ctx->lanes.size(); i++) ok = beside(c, ctx->lanes[i]);
        if (trace) std::fprintf(stderr, "[gc trace] deep lane candidate %d: %s\n", tries, ok ? "runs beside the ctx stream" : "shares a hardware queue: set aside");
        (ok ? ctx->lanes : ctx->lanes_aside).push_back(c);
    }
    (void)hipGetLastError();
    (void)hipHostFree(h);
    (void)hipFree(d_probe);
    if (!ctx->lanes.empty()) ctx->lanes_state = 1;
}

}  // namespace gcs
