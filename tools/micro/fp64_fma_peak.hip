// Measures the vector fp64 FMA issue rate of the device (v_fma_f64, wave64): N independent chains per lane, W waves per
// SIMD.  Build: hipcc -O3 --offload-arch=gfx950 tools/micro/fp64_fma_peak.hip -o tools/micro/fp64_fma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int CH>
__global__ __launch_bounds__(256) void k(double* out, double a, double b, int iters)
{
    double acc[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < CH; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    int ncu = 256;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); ncu = p.multiProcessorCount;
    double* out; hipMalloc(&out, sizeof(double) * 256 * ncu * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    for (int wpc = 1; wpc <= 2; ++wpc) {          // workgroups (of 4 waves) per CU: 1 or 2 waves per SIMD
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<16>, dim3(ncu * wpc), dim3(256), 0, 0, out, 1.0000001, 1e-9, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fmas = (double)ncu * wpc * 256 * iters * 8 * 16;
            if (rep == 2) printf("CUs %d, %d waves/SIMD: %.3f ms, %.2f TFLOP/s (2 flop per FMA), %.2f cycles per wave-FMA per SIMD at %.2f GHz nominal\n",
                                 ncu, wpc, ms, 2 * fmas / ms / 1e9, (ms * 1e-3 * p.clockRate * 1e3) / ((double)wpc * iters * 8 * 16), p.clockRate / 1e6);
        }
    }
    return 0;
}
