// What the set-up and tear-down calls of a drop-in run cost on this box: tools/hipcost_probe.hip
// hipcc --offload-arch=gfx950 -O2 -o /tmp/hipcost tools/hipcost_probe.hip && /tmp/hipcost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_nop(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define T(label, reps, body) { double t0 = now(); for (int r_ = 0; r_ < reps; ++r_) { body; } printf("%-44s %8.3f ms\n", label, (now() - t0) / reps); }
int main() {
    hipSetDevice(0);
    void* w = nullptr; hipMalloc(&w, 1 << 20); hipFree(w);
    hipStream_t st;
    T("hipStreamCreateWithFlags + Destroy", 10, { hipStreamCreateWithFlags(&st, hipStreamNonBlocking); hipStreamDestroy(st); });
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (size_t mb : {1, 8, 17, 88, 1200}) {
        char l[64]; void* p;
        snprintf(l, sizeof l, "hipMalloc %zu MB", mb);
        double tm = 0, tf = 0, tfu = 0;
        for (int r = 0; r < 5; ++r) {
            double t0 = now(); hipMalloc(&p, mb << 20); tm += now() - t0;
            t0 = now(); hipFree(p); tf += now() - t0;
            hipMalloc(&p, mb << 20); hipMemsetAsync(p, 0, mb << 20, st); hipStreamSynchronize(st);
            t0 = now(); hipFree(p); tfu += now() - t0;
        }
        printf("%-44s %8.3f ms, hipFree untouched %8.3f ms, hipFree after use %8.3f ms\n", l, tm / 5, tf / 5, tfu / 5);
    }
    void* h;
    T("hipHostMalloc 4 MB + hipHostFree", 5, { hipHostMalloc(&h, 4 << 20, hipHostMallocPortable); hipHostFree(h); });
    hipEvent_t e;
    T("hipEventCreateWithFlags + Destroy", 20, { hipEventCreateWithFlags(&e, hipEventDisableTiming); hipEventDestroy(e); });
    T("hipFuncSetAttribute(max dynamic LDS)", 20, { hipFuncSetAttribute((const void*)k_nop, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); });
    int occ;
    T("hipOccupancyMaxActiveBlocksPerMultiprocessor", 20, { hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_nop, 256, 65536); });
    hipDeviceProp_t prop;
    T("hipGetDeviceProperties", 5, { hipGetDeviceProperties(&prop, 0); });
    void* d; hipMalloc(&d, 16 << 20);
    T("hipMemsetAsync 1 MB + sync", 20, { hipMemsetAsync(d, 0, 1 << 20, st); hipStreamSynchronize(st); });
    T("kernel launch + sync", 20, { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, (int*)nullptr); hipStreamSynchronize(st); });
    std::vector<int> hb(1 << 20);
    T("hipMemcpyAsync H2D 4 MB pageable + sync", 10, { hipMemcpyAsync(d, hb.data(), 4 << 20, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); });
    T("hipMemcpyAsync D2H 4 MB pageable + sync", 10, { hipMemcpyAsync(hb.data(), d, 4 << 20, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); });
    return 0;
}
