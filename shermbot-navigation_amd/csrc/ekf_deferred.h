// ekf_deferred.h -- deferred application of the corrections of one tick.
//
// The reference applies every correction to the whole covariance at once:  P_i = (I - K_i H_i) P_{i-1}
// (slam_library.cpp:279), one pass over len^2 entries per marker.  Written out, P_i = P_{i-1} - K_i (H_i P_{i-1}),
// so after j corrections
//
//      P_j = P_0 - sum_{i<=j} U_i V_i,      U_i = K_i (len x 2),   V_i = H_i P_{i-1} (2 x len).
//
// Everything correction j+1 needs from P_j -- its five rows and five columns at {0,1,2,c,c+1} -- can be formed from
// P_0 and the pending factors in O(j len) work, so the corrections of a tick only touch O(len) data each
// (k_update_deferred) and the covariance itself is rewritten ONCE, by a rank-2J pass (k_flush): 2 len^2 w bytes per
// tick instead of per correction.  This is the reference's algebra re-associated; it is exact in exact arithmetic
// and agrees with the eager kernel to rounding (asserted at 1e-9 from a warm snapshot in tests/test_gpu_deferred.py),
// but it is NOT bit-identical to the oracle, so it is opt-in (nuslam_ekf_set_deferred) and the eager k_update stays
// the default.  Known association only: associateLandmark needs every candidate's 5x5 block of the current P.
#pragma once

namespace nuslam {

constexpr int kMaxPending = 16;   // factors kept before a flush is forced

// One correction in factor form.  Thread t owns index t: row t of K (-> U), column t of H P (-> V), state entry t.
// U, V: [kMaxPending * 2][ld] per filter; factor i, component r lives at row 2i + r.
template <typename T, bool INLINE_ID>
__global__ __launch_bounds__(256) void k_update_deferred(View v, ObsArg o, int mode, int total_landmarks, int J,
                                                         const T* __restrict__ P0, double* __restrict__ U,
                                                         double* __restrict__ V)
{
    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const int t = blockIdx.x * 256 + tid;
    const int ld = v.ld, L = v.L;
    const int* ci = v.c_in + b * C_WORDS;
    const int seen = ci[C_SEEN], cached = ci[C_SEEN_CACHED], brk = ci[C_BRK], status0 = ci[C_STATUS];
    const int id_raw = INLINE_ID ? o.id0 : o.ids[b * o.stride + o.off];
    const Decision d = resolve(v.n, id_raw, seen, cached, brk, status0, mode, total_landmarks);
    const int c = d.c;
    const int set[5] = { 0, 1, 2, c, c + 1 };
    const double* s = v.s_in + (size_t)b * ld;
    double* so = v.s_out + (size_t)b * ld;
    const T* Pb = P0 + (size_t)b * v.p_stride;
    double* Ub = U + (size_t)b * 2 * kMaxPending * ld;
    double* Vb = V + (size_t)b * 2 * kMaxPending * ld;

    __shared__ double sUs[2 * kMaxPending][5];   // U_i(set[q], r)
    __shared__ double sVs[2 * kMaxPending][5];   // V_i(r, set[q])
    __shared__ double sBlk[5][5];                // P_j(set[q2], set[q]) at [q][q2]
    __shared__ double sShare[20];                // Hc[10], Sinv[4], lx, ly, dz0, dz1
    __shared__ int sFlag[2];                     // skip, status

    // the 20 J wave-uniform factor entries, cooperatively
    for (int e = tid; e < 2 * J * 5; e += 256) {
        const int f = e / 5, q = e % 5;
        sUs[f][q] = Ub[(size_t)f * ld + set[q]];
        sVs[f][q] = Vb[(size_t)f * ld + set[q]];
    }
    __syncthreads();
    if (tid < 25) {
        const int q = tid / 5, q2 = tid % 5;
        double acc = (double)Pb[(size_t)set[q] * ld + set[q2]];
        for (int f = 0; f < 2 * J; ++f) acc = fma(-sUs[f][q2], sVs[f][q], acc);
        sBlk[q][q2] = acc;
    }
    __syncthreads();
    if (tid < 64) {       // wave 0: the shared scalars (redundantly on every lane), published by lane 0
        bool skip0 = d.skip;
        int st = d.new_status;
        double Hc0[10], Si0[4], lx0 = 0, ly0 = 0, dz0 = 0, dz1 = 0;
#pragma unroll
        for (int q = 0; q < 10; ++q) Hc0[q] = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) Si0[q] = 0;
        if (!skip0) {
            const double th = s[0], x = s[1], y = s[2];
            double r, phi;
            fetch_obs(o, b, r, phi);
            if (d.init) {                                 // initializeLandmark, slam_library.cpp:255-261
                lx0 = x + r * cos(phi + th);
                ly0 = y + r * sin(phi + th);
            } else { lx0 = s[c]; ly0 = s[c + 1]; }
            double pb[5][5], S[4], zr, zb;
#pragma unroll
            for (int q = 0; q < 5; ++q)
#pragma unroll
                for (int q2 = 0; q2 < 5; ++q2) pb[q][q2] = sBlk[q][q2];
            jacobian_compact(x, y, lx0, ly0, Hc0);        // :268
            innovation_cov_block(pb, Hc0, v.R, S);        // :270
            if (inv2(S, Si0)) { skip0 = true; if (st == 0) st = kStatusSingular; }
            measurement(th, x, y, lx0, ly0, zr, zb);      // :265
            dz0 = r - zr;                                 // :272
            dz1 = phi - zb;
        }
        if (tid == 0) {
#pragma unroll
            for (int q = 0; q < 10; ++q) sShare[q] = Hc0[q];
#pragma unroll
            for (int q = 0; q < 4; ++q) sShare[10 + q] = Si0[q];
            sShare[14] = lx0; sShare[15] = ly0; sShare[16] = dz0; sShare[17] = dz1;
            sFlag[0] = skip0 ? 1 : 0; sFlag[1] = st;
        }
    }
    __syncthreads();
    const bool skip = sFlag[0] != 0;
    const double lx = sShare[14], ly = sShare[15];
    if (blockIdx.x == 0 && tid == 0) {
        int* co = v.c_out + b * C_WORDS;
        co[C_SEEN] = d.new_seen; co[C_SEEN_CACHED] = cached; co[C_BRK] = d.new_brk; co[C_STATUS] = sFlag[1];
        if (v.id_log && o.log_slot >= 0) v.id_log[(size_t)b * v.log_stride + o.log_slot] = d.id;
    }
    if (t >= ld) return;
    double* Un = Ub + (size_t)(2 * J) * ld;
    double* Vn = Vb + (size_t)(2 * J) * ld;
    double sv = (t < L) ? s[t] : 0.0;
    if (d.init && t == c) sv = lx;
    if (d.init && t == c + 1) sv = ly;
    if (skip || t >= L) {                                 // a skipped correction is a zero factor
        Un[t] = 0.0; Un[ld + t] = 0.0; Vn[t] = 0.0; Vn[ld + t] = 0.0;
        so[t] = sv;
        return;
    }
    double Hc[10], Sinv[4];
#pragma unroll
    for (int q = 0; q < 10; ++q) Hc[q] = sShare[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) Sinv[q] = sShare[10 + q];
    // columns set[q] of P_j at row t, rows set[q] of P_j at column t
    double Cq[5], Rq[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        Cq[q] = (double)Pb[(size_t)set[q] * ld + t];
        Rq[q] = (double)Pb[(size_t)t * ld + set[q]];
    }
    for (int f = 0; f < 2 * J; ++f) {
        const double u = Ub[(size_t)f * ld + t], w = Vb[(size_t)f * ld + t];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            Cq[q] = fma(-u, sVs[f][q], Cq[q]);
            Rq[q] = fma(-sUs[f][q], w, Rq[q]);
        }
    }
    double ph[2], K[2], G[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        double a = 0.0, g = 0.0;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            a = fma(Cq[q], Hc[rr + 2 * q], a);            // (P H^T)(t, rr)
            g = fma(Hc[rr + 2 * q], Rq[q], g);            // (H P)(rr, t)
        }
        ph[rr] = a;
        G[rr] = g;
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        double a = 0.0;
        a = fma(ph[0], Sinv[0 + 2 * s2], a);
        a = fma(ph[1], Sinv[1 + 2 * s2], a);
        K[s2] = a;
    }
    Un[t] = K[0]; Un[ld + t] = K[1];
    Vn[t] = G[0]; Vn[ld + t] = G[1];
    double acc = 0.0;
    acc = fma(K[0], sShare[16], acc);
    acc = fma(K[1], sShare[17], acc);
    sv = sv + acc;                                        // state += K (z - z_hat), :275
    if (t == 0) sv = normalize_angle(sv);                 // :276
    so[t] = sv;
}

// P <- P - sum_f U_f V_f (f < 2J), in place: one streaming pass, 2 len^2 w bytes, 2J FMA per element.
// Same tiling as k_update: a wave owns 64*VEC rows x 16 columns; the V strip of its columns goes through LDS.
template <typename T>
__global__ __launch_bounds__(256) void k_flush(View v, int J, T* __restrict__ P, const double* __restrict__ U,
                                               const double* __restrict__ V)
{
    typedef Pack16<T> vec_t;
    constexpr int VEC = 16 / sizeof(T);
    constexpr int CW = 16;
    const int b = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ld = v.ld, L = v.L;
    const int row0 = (blockIdx.x * 64 + lane) * VEC;
    const int strip = blockIdx.y * 4 + wave;
    const bool active = strip * CW < L;
    const int j0 = active ? strip * CW : 0;
    const bool rows_ok = row0 < ld;
    const int rowc = rows_ok ? row0 : 0;
    const int ncol = (L - j0) < CW ? (L - j0) : CW;
    const double* Ub = U + (size_t)b * 2 * kMaxPending * ld;
    const double* Vb = V + (size_t)b * 2 * kMaxPending * ld;
    __shared__ double sV[4][2 * kMaxPending][CW];

    T* Pw = P + (size_t)b * v.p_stride + (size_t)j0 * ld + rowc;
    vec_t p[CW];
#pragma unroll
    for (int jj = 0; jj < CW; ++jj) p[jj] = *reinterpret_cast<const vec_t*>(Pw + (size_t)(jj < ncol ? jj : 0) * ld);
    for (int e = lane; e < 2 * J * CW; e += 64) {
        const int f = e / CW, jj = e % CW;
        sV[wave][f][jj] = Vb[(size_t)f * ld + j0 + (jj < ncol ? jj : 0)];
    }
    __syncthreads();
    if (!active || !rows_ok) return;
    // fp64 accumulators: all 16 columns at once for T = double (the tile registers themselves), four at a time
    // for T = float so that the 2J-term chain is rounded to fp32 only once
    constexpr int GC = sizeof(T) == 8 ? CW : 4;
#pragma unroll
    for (int g0 = 0; g0 < CW; g0 += GC) {
        double acc[GC][VEC];
#pragma unroll
        for (int jj = 0; jj < GC; ++jj)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[jj][e] = (double)p[g0 + jj].v[e];
        for (int f = 0; f < 2 * J; ++f) {
            double u[VEC];
#pragma unroll
            for (int e = 0; e < VEC; e += 2) {
                const double2 uu = *reinterpret_cast<const double2*>(Ub + (size_t)f * ld + row0 + e);
                u[e] = -uu.x; u[e + 1] = -uu.y;
            }
#pragma unroll
            for (int jj = 0; jj < GC; ++jj) {
                const double w = sV[wave][f][g0 + jj];
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[jj][e] = fma(u[e], w, acc[jj][e]);
            }
        }
#pragma unroll
        for (int jj = 0; jj < GC; ++jj)
#pragma unroll
            for (int e = 0; e < VEC; ++e) p[g0 + jj].v[e] = (T)acc[jj][e];
    }
#pragma unroll
    for (int jj = 0; jj < CW; ++jj)
        if (jj < ncol) *reinterpret_cast<vec_t*>(Pw + (size_t)jj * ld) = p[jj];
}

} // namespace nuslam
