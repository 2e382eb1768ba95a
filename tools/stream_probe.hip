// stream_probe.hip -- what HBM rate does the X access pattern itself sustain on this chip?
// Reads an N x P int32 column-major matrix the way k_resample does (each wave reads a run of
// consecutive observations of one feature per load, 16 loads in flight), with 1, 2 or 4
// observations per lane, and only ORs the values.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int R, int NT>
__global__ __launch_bounds__(NT) void probe(const int* __restrict__ X, long N, int P, unsigned* out) {
    const int lane = threadIdx.x & 63;
    const long ntiles = N / ((long)NT * R);
    unsigned acc = 0;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long wave_base = tile * NT * R + (long)__builtin_amdgcn_readfirstlane(threadIdx.x & ~63) * R;
        const char* base = (const char*)(X + wave_base);
        const int voff = lane * 4 * R;
        for (int d0 = 0; d0 < P; d0 += 16) {
            unsigned v[16][R];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int d = d0 + u < P ? d0 + u : P - 1;
                __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (long)d * N * 4), 0, 0x7fffffff, 0x00020000);
                if constexpr (R == 1) v[u][0] = __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0);
                if constexpr (R == 2) { auto t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0); v[u][0] = t[0]; v[u][1] = t[1]; }
                if constexpr (R == 4) { auto t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0); v[u][0] = t[0]; v[u][1] = t[1]; v[u][2] = t[2]; v[u][3] = t[3]; }
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int r = 0; r < R; ++r) acc |= v[u][r] << u;
        }
    }
    if (acc == 0xdeadbeef) out[0] = acc;
}

template <int R, int NT>
void run(const int* X, long N, int P, unsigned* out, int grid, const char* name) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((probe<R, NT>), dim3(grid), dim3(NT), 0, 0, X, N, P, out);
    CK(hipEventRecord(a));
    const int reps = 10;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((probe<R, NT>), dim3(grid), dim3(NT), 0, 0, X, N, P, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-28s grid %5d  %.3f ms/pass  %.0f GB/s\n", name, grid, ms / reps, (double)N * P * 4 / (ms / reps * 1e-3) / 1e9);
}

int main() {
    const long N = 10000000 / 4096 * 4096; const int P = 100;
    int* X; unsigned* out;
    CK(hipMalloc(&X, (size_t)N * P * 4)); CK(hipMalloc(&out, 64));
    CK(hipMemset(X, 1, (size_t)N * P * 4));
    for (int g : {256, 512, 1024}) {
        run<1, 1024>(X, N, P, out, g, "1 obs/lane (dword) 1024thr");
        run<1, 512>(X, N, P, out, g, "1 obs/lane (dword) 512thr");
        run<2, 512>(X, N, P, out, g, "2 obs/lane (dwordx2) 512thr");
        run<4, 512>(X, N, P, out, g, "4 obs/lane (dwordx4) 512thr");
        run<4, 256>(X, N, P, out, g, "4 obs/lane (dwordx4) 256thr");
    }
    return 0;
}
