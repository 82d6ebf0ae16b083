// queue_overlap_ubench — do kernels on different HIP streams of one process really run side by side on this runtime?
// (developer aid behind the streaming engine's "deep lanes": a long one-workgroup kernel on a lane stream must not hold up
// the short kernels of the main stream).  Prints wall times for: main alone, lane alone, both; with the lane stream created
// at default or high priority, after `extra` other streams were created and used.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void spin(unsigned long long ticks, unsigned *sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned v = 0;
    while (wall_clock64() - t0 < ticks) v++;
    if (threadIdx.x == 0 && v == 0xffffffffu) *sink = v;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const int extra = argc > 1 ? std::atoi(argv[1]) : 2, nlanes = argc > 2 ? std::atoi(argv[2]) : 4, prio = argc > 3 ? std::atoi(argv[3]) : 0;
    unsigned *sink;
    CK(hipMalloc(&sink, 4));
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    std::printf("priority range: least %d greatest %d; extra streams %d, lanes %d, lane priority %s\n", lo, hi, extra, nlanes, prio ? "high" : "default");
    hipStream_t mainS;
    CK(hipStreamCreateWithFlags(&mainS, hipStreamNonBlocking));
    std::vector<hipStream_t> ex(extra), lanes(nlanes);
    for (auto &s : ex) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (auto &s : lanes) {
        if (prio) CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
        else CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    }
    // wall_clock64 ticks at 100 MHz
    const unsigned long long us = 100;
    auto warm = [&]() {
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, mainS, 10 * us, sink);
        for (auto &s : ex) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 10 * us, sink);
        for (auto &s : lanes) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 10 * us, sink);
        return hipDeviceSynchronize();
    };
    CK(warm());
    auto run_main = [&]() { for (int i = 0; i < 100; i++) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, mainS, 20 * us, sink); };
    auto run_lanes = [&](int n) { for (int l = 0; l < n; l++) for (int i = 0; i < 2; i++) hipLaunchKernelGGL(spin, dim3(1), dim3(1024), 0, lanes[l], 1000 * us, sink); };
    double t0 = now();
    run_main();
    CK(hipDeviceSynchronize());
    const double t_main = now() - t0;
    t0 = now();
    run_lanes(1);
    CK(hipDeviceSynchronize());
    const double t_lane = now() - t0;
    std::printf("main alone (100 x 20 us, 256 wg): %.3f ms; one lane alone (2 x 1 ms, 1 wg): %.3f ms\n", t_main, t_lane);
    for (int n = 1; n <= nlanes; n++) {
        t0 = now();
        run_lanes(n);
        run_main();
        CK(hipDeviceSynchronize());
        std::printf("%d lane(s) + main: %.3f ms (perfect overlap: %.3f, serial: %.3f)\n", n, now() - t0, t_lane > t_main ? t_lane : t_main, n * t_lane + t_main);
    }
    // the same with the extra streams busy too (copy / serialiser streams of the engine)
    t0 = now();
    run_lanes(nlanes);
    for (auto &s : ex) for (int i = 0; i < 20; i++) hipLaunchKernelGGL(spin, dim3(8), dim3(256), 0, s, 50 * us, sink);
    run_main();
    CK(hipDeviceSynchronize());
    std::printf("%d lanes + %d busy extra streams + main: %.3f ms\n", nlanes, extra, now() - t0);
    return 0;
}
