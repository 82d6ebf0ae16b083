// graph_node_ubench — what does ONE dependent kernel node of a hipGraph cost on this runtime / GPU?
// Behind bench.py's `level_launch` row (schedule 0, the north star's literal shape: one launch per dependency level): aes_128 is
// 308 + 308 dependent launches per step, 15.9 us per level pair in BENCH_r05.  This prints the floor under that: a chain of N
// dependent nodes that do (a) nothing, (b) the HBM traffic of an average aes_128 level at 1 024 instances (~6.5 MB read + write),
// replayed from a graph and launched directly.
// build: hipcc --offload-arch=gfx950 -O2 tools/graph_node_ubench.hip -o tools/graph_node_ubench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void k_empty(unsigned *sink) {
    if (threadIdx.x == 0xffffffffu) *sink = 1;
}
// y[i] ^= x[i] over n uint4: n * 48 bytes of traffic (two reads and a write per element: a free gate's model)
__global__ void k_xor(const uint4 *__restrict__ a, const uint4 *__restrict__ b, uint4 *__restrict__ o, unsigned n) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint4 x = a[i], y = b[i];
        o[i] = make_uint4(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w);
    }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const int nodes = argc > 1 ? std::atoi(argv[1]) : 308, reps = 20;
    const unsigned n = 135000;  // x 48 B = 6.5 MB per node
    uint4 *a, *b, *o;
    unsigned *sink;
    CK(hipMalloc(&a, n * 16));
    CK(hipMalloc(&b, n * 16));
    CK(hipMalloc(&o, n * 16));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, n * 16));
    CK(hipMemset(b, 2, n * 16));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int mode = 0; mode < 2; mode++) {
        auto launch_all = [&]() {
            for (int i = 0; i < nodes; i++) {
                if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, sink);
                else hipLaunchKernelGGL(k_xor, dim3((n + 255) / 256), dim3(256), 0, s, a, b, o, n);
            }
        };
        launch_all();
        CK(hipStreamSynchronize(s));
        double t0 = now_us();
        for (int r = 0; r < reps; r++) launch_all();
        CK(hipStreamSynchronize(s));
        const double direct = (now_us() - t0) / reps / nodes;
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        launch_all();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        t0 = now_us();
        for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        const double graph = (now_us() - t0) / reps / nodes;
        std::printf("%-46s %d dependent nodes: %.2f us per node launched directly, %.2f us per node in a hipGraph\n",
                    mode == 0 ? "empty kernel (1 workgroup):" : "6.5 MB of label traffic (528 workgroups):", nodes, direct, graph);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
