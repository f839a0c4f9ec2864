// Is v_mfma_f64_16x16x4_f64 the k-ordered fma chain  d = fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, c))))  bit for bit?
// (What k_tick_rank's "same k-ordered fma chain" statement and any VALU replica of its sums rest on.)
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/microbench/mfma_f64_order.hip -o /tmp/mfma_order && /tmp/mfma_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k(const double* A, const double* B, const double* C, double* D)
{
    // operand layout of v_mfma_f64_16x16x4_f64: lane l holds A[i = l % 16][k = l / 16], B[k = l / 16][j = l % 16];
    // the result: lane l holds D[i = l / 16 + 4 * r][j = l % 16], r = 0..3  (ekf_rank.h: col = g4 + 4 r)
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16], b = B[(l / 16) * 16 + l % 16];
    d4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[(l / 16 + 4 * r) * 16 + l % 16];
    const d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(l / 16 + 4 * r) * 16 + l % 16] = d[r];
}

int main()
{
    const int trials = 2000;
    long long bad_seq = 0, bad_rev = 0, bad_tree = 0, total = 0;
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dC, 256 * 8); hipMalloc(&dD, 256 * 8);
    srand(7);
    for (int t = 0; t < trials; ++t) {
        double A[64], B[64], C[256], D[256];
        for (int i = 0; i < 64; ++i) {
            A[i] = (rand() / (double)RAND_MAX - 0.5) * std::ldexp(1.0, rand() % 40 - 20);
            B[i] = (rand() / (double)RAND_MAX - 0.5) * std::ldexp(1.0, rand() % 40 - 20);
            if (t % 5 == 0 && i % 4 >= 2) A[i] = 0.0;                 // (the pass pads two of four k with exact zeros when a correction is masked)
        }
        for (int i = 0; i < 256; ++i) C[i] = (rand() / (double)RAND_MAX - 0.5) * std::ldexp(1.0, rand() % 40 - 20);
        hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
        hipMemcpy(dC, C, sizeof(C), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D, dD, sizeof(D), hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double s = C[i * 16 + j], r = C[i * 16 + j];
                for (int kk = 0; kk < 4; ++kk) s = std::fma(A[i * 4 + kk], B[kk * 16 + j], s);
                for (int kk = 3; kk >= 0; --kk) r = std::fma(A[i * 4 + kk], B[kk * 16 + j], r);
                const double tr = std::fma(A[i * 4 + 0], B[0 * 16 + j], A[i * 4 + 1] * B[1 * 16 + j]) + std::fma(A[i * 4 + 2], B[2 * 16 + j], A[i * 4 + 3] * B[3 * 16 + j]) + C[i * 16 + j];
                const double d = D[i * 16 + j];
                bad_seq += std::memcmp(&d, &s, 8) != 0; bad_rev += std::memcmp(&d, &r, 8) != 0; bad_tree += std::memcmp(&d, &tr, 8) != 0;
                ++total;
            }
    }
    std::printf("v_mfma_f64_16x16x4_f64 against CPU fma chains over %lld elements: k ascending (fma(a3,b3, .. fma(a0,b0,c))) %lld differ; k descending %lld differ; pairwise tree %lld differ\n",
                total, bad_seq, bad_rev, bad_tree);
    return 0;
}
