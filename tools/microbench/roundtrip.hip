// roundtrip.hip -- what one host <-> device round trip costs on this box, for the call-by-call (api-driven) path:
// associateLandmark() must hand an id back to the caller before the next call can be made (slam.cpp:291).
//   A  launch + hipStreamSynchronize
//   B  launch, the kernel stores a sequence word into mapped pinned host memory, the host spins on it
//   C  two dependent launches + the same spin (what a separate "finish" kernel costs)
//   D  a resident kernel polling a mapped mailbox: host writes a command word, device answers (bounded: exits by itself)
//   E  launch of a kernel with N busy microseconds + spin (launch latency does not hide behind kernel time)
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench/roundtrip.hip -o tools/microbench/roundtrip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_nop(int* out, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = v; }
__global__ void k_seq(volatile int* host_word, int v)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        __hip_atomic_store((int*)host_word, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void k_mid(int* tmp, int v) { if (threadIdx.x == 0) tmp[blockIdx.x] = v; }
__global__ void k_busy(volatile int* host_word, int v, long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store((int*)host_word, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// resident server: waits for cmd == expected (1, 2, ...), answers with the same number; leaves after `rounds` or on timeout
__global__ void k_serve(volatile int* cmd, volatile int* ans, int rounds, long long timeout_ticks)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int r = 1; r <= rounds; ++r) {
        const long long t0 = wall_clock64();
        bool ok = false;
        while (wall_clock64() - t0 < timeout_ticks) {
            if (__hip_atomic_load((int*)cmd, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= r) { ok = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) { __hip_atomic_store((int*)ans, -r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); return; }
        __hip_atomic_store((int*)ans, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int* d; CK(hipMalloc(&d, 4096));
    int* hw; CK(hipHostMalloc((void**)&hw, 4096, hipHostMallocMapped));
    volatile int* vw = hw;
    hw[0] = 0; hw[16] = 0; hw[32] = 0;
    const int N = 2000;
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, d, i);
    CK(hipStreamSynchronize(st));
    double t0 = now_us();
    for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, d, i); CK(hipStreamSynchronize(st)); }
    printf("A launch + hipStreamSynchronize            : %.2f us per round trip\n", (now_us() - t0) / N);
    t0 = now_us();
    for (int i = 1; i <= N; ++i) { hipLaunchKernelGGL(k_seq, dim3(1), dim3(64), 0, st, vw, i); while (vw[0] != i) { } }
    printf("B launch + spin on mapped host word        : %.2f us\n", (now_us() - t0) / N);
    CK(hipStreamSynchronize(st));
    t0 = now_us();
    for (int i = 1; i <= N; ++i) {
        hipLaunchKernelGGL(k_mid, dim3(34), dim3(256), 0, st, d, i);
        hipLaunchKernelGGL(k_seq, dim3(1), dim3(64), 0, st, vw, N + i);
        while (vw[0] != N + i) { }
    }
    printf("C two dependent launches + spin            : %.2f us\n", (now_us() - t0) / N);
    CK(hipStreamSynchronize(st));
    for (long long busy : { 300LL, 600LL }) {            // 100 MHz ticks: 3 us, 6 us
        t0 = now_us();
        for (int i = 1; i <= N; ++i) { hipLaunchKernelGGL(k_busy, dim3(34), dim3(256), 0, st, vw, 3 * N + i, busy); while (vw[0] != 3 * N + i) { } }
        printf("E launch of a %.0f us kernel + spin          : %.2f us\n", busy / 100.0, (now_us() - t0) / N);
        CK(hipStreamSynchronize(st));
    }
    // D: mailbox
    volatile int* cmd = hw + 16; volatile int* ans = hw + 32;
    const int R = 2000;
    hipLaunchKernelGGL(k_serve, dim3(1), dim3(64), 0, st, cmd, ans, R, 100LL * 1000 * 100);   // 100 ms timeout per command
    t0 = now_us();
    int bad = 0;
    for (int r = 1; r <= R; ++r) {
        __atomic_store_n((int*)cmd, r, __ATOMIC_RELEASE);
        int a;
        while ((a = __atomic_load_n((int*)ans, __ATOMIC_ACQUIRE)) != r) { if (a < 0) { bad = a; break; } }
        if (bad) break;
    }
    const double dt = (now_us() - t0) / R;
    CK(hipStreamSynchronize(st));
    printf("D resident mailbox round trip              : %.2f us%s\n", dt, bad ? " (TIMEOUT)" : "");
    // launch throughput: enqueue only
    t0 = now_us();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, d, i);
    const double enq = (now_us() - t0) / N;
    CK(hipStreamSynchronize(st));
    printf("F enqueue only (host cost per launch)      : %.2f us;  drained in %.2f us per launch\n", enq, (now_us() - t0) / N);
    // small D2H copy
    int hv = 0;
    t0 = now_us();
    for (int i = 0; i < N; ++i) CK(hipMemcpy(&hv, d, 4, hipMemcpyDeviceToHost));
    printf("G hipMemcpy D2H of 4 bytes                 : %.2f us\n", (now_us() - t0) / N);
    return 0;
}
