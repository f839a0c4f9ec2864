// Latency of the instructions the serial chain of a tick is made of, ONE wave on a SIMD (as in k_tick_chain / k_tick_front's
// workgroup 0): dependent fp64 FMA, division, sqrt, v_readlane, an LDS round trip, a workgroup barrier among 4 waves.
// Build: hipcc -O3 --offload-arch=gfx950 -o latency latency.hip ; run on the GPU box.  Prints shader cycles per operation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(long long* out, double* sink, double seed, int reps)
{
    __shared__ double lds[512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lds[threadIdx.x] = seed + threadIdx.x;
    __syncthreads();
    double a = seed + lane * 1e-3, b = 1.0000001, c = 1e-9, acc = a;
    long long t0, t1;
    long long r[12] = { 0 };
    // 1. dependent FMA chain
    t0 = clock64();
    for (int i = 0; i < reps; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = fma(acc, b, c);
    }
    t1 = clock64(); r[0] = t1 - t0;
    // 2. four independent FMA chains
    double a0 = acc, a1 = acc + 1, a2 = acc + 2, a3 = acc + 3;
    t0 = clock64();
    for (int i = 0; i < reps; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { a0 = fma(a0, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c); }
    }
    t1 = clock64(); r[1] = t1 - t0;
    acc = a0 + a1 + a2 + a3;
    // 3. dependent divisions
    t0 = clock64();
    for (int i = 0; i < reps; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = b / (acc + c);
    }
    t1 = clock64(); r[2] = t1 - t0;
    // 4. dependent sqrt
    acc = fabs(acc) + 2.0;
    t0 = clock64();
    for (int i = 0; i < reps; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = sqrt(acc + b);
    }
    t1 = clock64(); r[3] = t1 - t0;
    // 5. readlane + use (broadcast of a double, consumed by a VALU op)
    t0 = clock64();
    for (int i = 0; i < reps; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(acc), u), hi = __builtin_amdgcn_readlane(__double2hiint(acc), u);
            acc = acc + __hiloint2double(hi, lo) * c;
        }
    }
    t1 = clock64(); r[4] = t1 - t0;
    // 6. dependent LDS round trips (pointer chase through LDS)
    int idx = lane;
    t0 = clock64();
    for (int i = 0; i < reps; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) idx = ((int)lds[idx & 255]) & 255;
    }
    t1 = clock64(); r[5] = t1 - t0;
    acc += idx;
    // 7. barrier among the workgroup's waves (all arrive together)
    t0 = clock64();
    for (int i = 0; i < reps; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    t1 = clock64(); r[6] = t1 - t0;
    // 8. 25 broadcasts to every lane (50 readlanes) feeding 70 FMAs: the S = H P H^T block of the chain's head
    double e = acc;
    t0 = clock64();
    for (int i = 0; i < reps; ++i) {
        double pb[25];
#pragma unroll
        for (int u = 0; u < 25; ++u) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(e), u), hi = __builtin_amdgcn_readlane(__double2hiint(e), u);
            pb[u] = __hiloint2double(hi, lo);
        }
        double hp[10];
#pragma unroll
        for (int q = 0; q < 5; ++q)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                double s = 0.0;
#pragma unroll
                for (int q2 = 0; q2 < 5; ++q2) s = fma(b + rr + q2, pb[5 * q + q2], s);
                hp[2 * q + rr] = s;
            }
        double S = 0.0;
#pragma unroll
        for (int q = 0; q < 10; ++q) S = fma(hp[q], b, S);
        e = e + S * c;
    }
    t1 = clock64(); r[7] = t1 - t0;
    acc += e;
    // 9. LDS write by one lane then read by all after a barrier (the hand-off hd[] of the chain)
    t0 = clock64();
    for (int i = 0; i < reps; ++i) {
        if (threadIdx.x == 128) lds[300] = acc;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        acc = acc + lds[300] * c;
    }
    t1 = clock64(); r[8] = t1 - t0;
    if (lane == 0) for (int q = 0; q < 12; ++q) out[wave * 12 + q] = r[q];
    sink[threadIdx.x] = acc;
}

int main()
{
    long long* out; double* sink;
    hipMalloc(&out, 4 * 12 * sizeof(long long));
    hipMalloc(&sink, 256 * sizeof(double));
    const int reps = 64;
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, out, sink, 1.5, reps);
    hipDeviceSynchronize();
    std::vector<long long> h(48);
    hipMemcpy(h.data(), out, 48 * sizeof(long long), hipMemcpyDeviceToHost);
    const char* names[9] = { "dependent fp64 FMA", "fp64 FMA, 4 independent chains (per FMA)", "dependent fp64 division (+1 add)",
                             "dependent fp64 sqrt (+1 add)", "double broadcast (2 readlanes) + mul + add", "dependent LDS read (+cvt, and)",
                             "workgroup barrier, 4 waves", "25 broadcasts + 60 FMAs (the head's S block)", "LDS write, barrier, read" };
    const double per[9] = { 16.0 * reps, 16.0 * reps, 4.0 * reps, 4.0 * reps, 16.0 * reps, 8.0 * reps, 8.0 * reps, 1.0 * reps, 1.0 * reps };
    for (int q = 0; q < 9; ++q)
        printf("%-48s %8.1f cycles  (wave 0; waves 1-3: %.1f %.1f %.1f)\n", names[q], h[q] / per[q], h[12 + q] / per[q], h[24 + q] / per[q],
               h[36 + q] / per[q]);
    return 0;
}
