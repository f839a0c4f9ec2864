// dense_predict.hip -- P <- F P F^T + Qbar as two LDS-tiled MFMA GEMMs for gfx950 (wave64).
//
// Reference: the dense covariance propagation `A * covariance * A.t() + Q_bar` of nuslam/src/slam_library.cpp:104
// (4 len^3 flop).  For the reference's own motion model A = I + B has two non-zeros and k_predict does the same
// job in O(len); this path serves a caller-supplied dense Jacobian (nuslam_ekf_predict_dense) and is the one
// MFMA-bound kernel of the engine.
//
// Matrix cores used: v_mfma_f32_32x32x2_f32 (exact f32 fma chain) / v_mfma_f64_16x16x4_f64.
// Everything is column-major.  Output tiles are produced with the MFMA "N" index (the lane) running along the
// rows of C, so stores are 128-byte coalesced along the contiguous dimension: the MFMA B operand is our A tile,
// the MFMA A operand is our B tile.
#include "dense_predict.h"

#include <hip/hip_ext.h>

namespace {

template <typename T> struct Mfma;

template <> struct Mfma<float> {
    static constexpr int TM = 32;   // tile edge
    static constexpr int TK = 2;    // k per instruction
    static constexpr int NACC = 16; // accumulator registers per lane
    typedef float acc_t __attribute__((ext_vector_type(16)));
    static __device__ inline acc_t run(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
    static __device__ inline int idx(int lane) { return lane & 31; }
    static __device__ inline int kk(int lane) { return lane >> 5; }
    static __device__ inline int row(int lane, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
};

template <> struct Mfma<double> {
    static constexpr int TM = 16;
    static constexpr int TK = 4;
    static constexpr int NACC = 4;
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ inline acc_t run(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ inline int idx(int lane) { return lane & 15; }
    static __device__ inline int kk(int lane) { return lane >> 4; }
    static __device__ inline int row(int lane, int r) { return (lane >> 4) + 4 * r; }   // f64 C/D map differs from f32
};

template <typename T> struct alignas(16) V16 { T v[16 / sizeof(T)]; };

constexpr int BK = 16;

// C(i,j) = sum_k A(i,k) * Bop(k,j) [+ Q(i,j) on the 3x3 corner]
//   A   : (i,k) at A[i + k*ld]                         (contiguous along i)
//   Bop : B_KMAJOR ? B[k + j*ld] : B[j + k*ld]         (first GEMM: B = P, k-major; second: B = F used as F^T)
// 256 threads = 4 waves in a 2x2 grid, each wave owns a 64x64 sub-tile.
//
// Pipeline: two LDS stages; while the MFMAs of K-tile t run from stage t&1, the global loads of tile t+1 are in
// flight into registers and are written to stage (t+1)&1 after the compute -- one barrier per K-tile.
// Workgroup order: the linear id is remapped so that the workgroups dealt to one XCD (ids congruent mod 8) cover a
// contiguous band of block rows and share their B strips in that XCD's L2.
//
// TILE = 128 (a wave owns 64 x 64) or 64 (32 x 32).  The grid is one-dimensional: this launch covers the 128 x 128 tiles
// tile0 .. tile0 + n - 1 of the gx-wide tile grid, as n workgroups (TILE = 128) or 4 n (TILE = 64, four quadrants per
// tile).  The small tiles exist for the TAIL: 6241 tiles on 768 resident workgroups leave a ninth generation that is one
// eighth full -- a whole tile time with most CUs idle (10 % of the product).  Those last tiles are launched as quadrants
// instead: four times the workgroups, a quarter of the time each, the same sum for every element (k ascending), so the
// result does not change by a bit.
template <typename T, bool B_KMAJOR, bool ADD_Q, int TILE>
__global__ __launch_bounds__(256) void k_gemm(int L, int ld, int gx, int tile0, const T* __restrict__ A, const T* __restrict__ B,
                                              T* __restrict__ C, double q00, double q10, double q20, double q01,
                                              double q11, double q21, double q02, double q12, double q22)
{
    typedef Mfma<T> M;
    typedef typename M::acc_t acc_t;
    constexpr int BM = TILE, BN = TILE;
    constexpr int VEC = 16 / sizeof(T);
    constexpr int NT = (TILE / 2) / M::TM;    // MFMA tiles per wave edge
    constexpr int PB = B_KMAJOR ? 1 : 0;      // odd row stride -> conflict-free transposing stores
    constexpr int AV = BM / VEC;              // 16-byte vectors per k-row of the A tile
    constexpr int NA = (BK * AV) / 256;       // vectors per thread, A tile
    constexpr int KV = BK / VEC;              // vectors per column of a k-major B tile
    constexpr int BV = BN / VEC;
    constexpr int NB = B_KMAJOR ? (BN * KV) / 256 : (BK * BV) / 256;
    __shared__ T As[2][BK][BM];
    __shared__ T Bs[2][BK][BN + PB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int it = (wave & 1) * (TILE / 2), jt = (wave >> 1) * (TILE / 2);
    // XCD-aware order (bijective for any grid size)
    const int nwg = gridDim.x;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    constexpr int SUB = 128 / TILE;           // quadrants per tile edge
    const int tile = tile0 + wg / (SUB * SUB), quad = wg % (SUB * SUB);
    const int i0 = (tile % gx) * 128 + (quad % SUB) * TILE, j0 = (tile / gx) * 128 + (quad / SUB) * TILE;
    const int idx = M::idx(lane), kk = M::kk(lane);

    acc_t acc[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < M::NACC; ++r) acc[a][b][r] = 0;

    V16<T> ra[NA], rb[NB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int s = 0; s < NA; ++s) {
            const int v = tid + s * 256;
            const int k = v / AV, iv = (v % AV) * VEC;
#pragma unroll
            for (int e = 0; e < VEC; ++e) ra[s].v[e] = 0;
            if (k0 + k < L && i0 + iv < ld) ra[s] = *reinterpret_cast<const V16<T>*>(A + (size_t)(k0 + k) * ld + i0 + iv);
        }
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            const int v = tid + s * 256;
#pragma unroll
            for (int e = 0; e < VEC; ++e) rb[s].v[e] = 0;
            if (B_KMAJOR) {
                const int kq = (v % KV) * VEC, j = v / KV;
                if (j0 + j < L && k0 + kq < ld) rb[s] = *reinterpret_cast<const V16<T>*>(B + (size_t)(j0 + j) * ld + k0 + kq);
            } else {
                const int k = v / BV, jv = (v % BV) * VEC;
                if (k0 + k < L && j0 + jv < ld) rb[s] = *reinterpret_cast<const V16<T>*>(B + (size_t)(k0 + k) * ld + j0 + jv);
            }
        }
    };
    auto lstore = [&](int st) {
#pragma unroll
        for (int s = 0; s < NA; ++s) {
            const int v = tid + s * 256;
            *reinterpret_cast<V16<T>*>(&As[st][v / AV][(v % AV) * VEC]) = ra[s];
        }
#pragma unroll
        for (int s = 0; s < NB; ++s) {
            const int v = tid + s * 256;
            if (B_KMAJOR) {
                const int kq = (v % KV) * VEC, j = v / KV;
#pragma unroll
                for (int e = 0; e < VEC; ++e) Bs[st][kq + e][j] = rb[s].v[e];
            } else {
                *reinterpret_cast<V16<T>*>(&Bs[st][v / BV][(v % BV) * VEC]) = rb[s];
            }
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();
    const int nk = (L + BK - 1) / BK;
    for (int t = 0; t < nk; ++t) {
        const int st = t & 1;
        if (t + 1 < nk) gload((t + 1) * BK);              // in flight during the MFMAs below
#pragma unroll
        for (int ks = 0; ks < BK; ks += M::TK) {
            T af[NT], bf[NT];
#pragma unroll
            for (int a = 0; a < NT; ++a) af[a] = Bs[st][ks + kk][jt + a * M::TM + idx];   // MFMA A operand <- our B tile
#pragma unroll
            for (int b = 0; b < NT; ++b) bf[b] = As[st][ks + kk][it + b * M::TM + idx];   // MFMA B operand <- our A tile
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) acc[a][b] = M::run(af[a], bf[b], acc[a][b]);
        }
        if (t + 1 < nk) lstore(st ^ 1);                   // stage st^1 was last read in iteration t-1 (barrier since)
        __syncthreads();
    }

    // ---- epilogue: lane <-> row i (contiguous), register <-> column j
    const double q[9] = { q00, q10, q20, q01, q11, q21, q02, q12, q22 };
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int i = i0 + it + b * M::TM + idx;
#pragma unroll
            for (int r = 0; r < M::NACC; ++r) {
                const int j = j0 + jt + a * M::TM + M::row(lane, r);
                if (i < L && j < L) {
                    T val = acc[a][b][r];
                    if (ADD_Q && i < 3 && j < 3) val = (T)((double)val + q[i + 3 * j]);
                    C[(size_t)j * ld + i] = val;
                }
            }
        }
}

template <typename T>
int run_t(int L, int ld, const T* F, T* P, T* Tw, const double Q[9], hipStream_t stream, hipEvent_t ev[4])
{
    const int gx = (L + 127) / 128, ntiles = gx * gx;
    // the tail (see k_gemm): what is left over after whole generations of resident workgroups, if that is less than half
    // a generation, goes out as quadrants behind the big tiles
    static int n_cu = 0;                                                 // (one device type per process)
    if (!n_cu) {
        int dev = 0, cus = 0;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) ? cus : 256;
    }
    const int gen = n_cu * (sizeof(T) == 4 ? 3 : 1);                    // resident workgroups: 3 per CU (f32), 1 (f64: 65 KB of LDS, 206 + 128 registers)
    int tail = ntiles % gen;
    if (tail * 2 > gen || ntiles < gen) tail = 0;
    const int big = ntiles - tail;
    dim3 block(256);
    // T = F * P            (slam_library.cpp:104, left product first)
    if (big)
        hipExtLaunchKernelGGL((k_gemm<T, true, false, 128>), dim3(big), block, 0, stream, ev ? ev[0] : nullptr, tail ? nullptr : (ev ? ev[1] : nullptr), 0,
                              L, ld, gx, 0, F, (const T*)P, Tw, 0., 0., 0., 0., 0., 0., 0., 0., 0.);
    if (tail)
        hipExtLaunchKernelGGL((k_gemm<T, true, false, 64>), dim3(4 * tail), block, 0, stream, big ? nullptr : (ev ? ev[0] : nullptr), ev ? ev[1] : nullptr, 0,
                              L, ld, gx, big, F, (const T*)P, Tw, 0., 0., 0., 0., 0., 0., 0., 0., 0.);
    if (hipGetLastError() != hipSuccess) return 1;
    // P = T * F^T + Qbar
    if (big)
        hipExtLaunchKernelGGL((k_gemm<T, false, true, 128>), dim3(big), block, 0, stream, ev ? ev[2] : nullptr, tail ? nullptr : (ev ? ev[3] : nullptr), 0,
                              L, ld, gx, 0, (const T*)Tw, F, P, Q[0], Q[1], Q[2], Q[3], Q[4], Q[5], Q[6], Q[7], Q[8]);
    if (tail)
        hipExtLaunchKernelGGL((k_gemm<T, false, true, 64>), dim3(4 * tail), block, 0, stream, big ? nullptr : (ev ? ev[2] : nullptr), ev ? ev[3] : nullptr, 0,
                              L, ld, gx, big, (const T*)Tw, F, P, Q[0], Q[1], Q[2], Q[3], Q[4], Q[5], Q[6], Q[7], Q[8]);
    if (hipGetLastError() != hipSuccess) return 1;
    return 0;
}

} // namespace

int dense_predict_launch(int dtype, int L, int ld, const void* F, void* P, void* T, const double Q[9],
                         hipStream_t stream, hipEvent_t ev[4])
{
    hipEvent_t* e = (ev && ev[0]) ? ev : nullptr;
    if (dtype == 1) return run_t<float>(L, ld, (const float*)F, (float*)P, (float*)T, Q, stream, e);
    return run_t<double>(L, ld, (const double*)F, (double*)P, (double*)T, Q, stream, e);
}
