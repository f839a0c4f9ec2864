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
//
// The grid is only ceil(ld/256) workgroups, so the kernel is one dependent chain and its length is what counts:
//   ONE burst of global loads at the top (own U/V rows, P0 row/column entries, the 20 J wave-uniform factor entries,
//   the 5x5 block entry of the lanes that sum it) -- all unconditional with clamped indices and multiplicative masks,
//   because hipcc sinks a load whose value is only selected into a branch that ends in s_waitcnt vmcnt(0);
//   -> barrier -> in parallel: all threads form C_q(t), R_q(t) of P_j; wave 3 sums the 5x5 block of P_j (2 lanes per
//   entry); wave 1 evaluates cartesian2polar and z_hat (atan2/sin/cos), which need the state only
//   -> barrier -> wave 0: H, S, S^-1 -> barrier -> K, H P, state.
template <typename T, bool INLINE_ID>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_update_deferred(View v, ObsArg o, int mode, int total_landmarks, int J,
                                                         const T* __restrict__ P0, double* __restrict__ U,
                                                         double* __restrict__ V)
{
    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
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
    const int F2 = 2 * J;
    const bool live = t < L;
    const int tc = live ? t : 0;

    __shared__ double sUs[2 * kMaxPending][5];   // U_i(set[q], r)
    __shared__ double sVs[2 * kMaxPending][5];   // V_i(r, set[q])
    __shared__ double sBlk[5][5];                // P_j(set[q2], set[q]) at [q][q2]
    __shared__ double sShare[20];                // Hc[10], Sinv[4], lx, ly, zr|dz0, zb|dz1, r, phi
    __shared__ int sFlag[2];                     // skip, status

    // ---- the burst
    double uo[2 * kMaxPending], wo[2 * kMaxPending];     // own row of U, own column of V (zero beyond F2)
#pragma unroll
    for (int f = 0; f < 2 * kMaxPending; ++f) {
        const int fc = f < F2 ? f : 0;
        const double mk = f < F2 ? 1.0 : 0.0;
        uo[f] = Ub[(size_t)fc * ld + tc] * mk;
        wo[f] = Vb[(size_t)fc * ld + tc] * mk;
    }
    double Cq[5], Rq[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        Cq[q] = (double)Pb[(size_t)set[q] * ld + tc];
        Rq[q] = (double)Pb[(size_t)tc * ld + set[q]];
    }
    // 2 F2 x 5 wave-uniform factor entries: thread e < 160 fetches U(f, set[q]) and V(f, set[q])
    const int ef = (tid / 5) < 2 * kMaxPending ? tid / 5 : 0, eq = tid % 5;
    const double mke = (tid < 2 * kMaxPending * 5 && ef < F2) ? 1.0 : 0.0;
    const int efc = ef < F2 ? ef : 0;
    const double stU = Ub[(size_t)efc * ld + set[eq]] * mke;
    const double stV = Vb[(size_t)efc * ld + set[eq]] * mke;
    // the 5x5 block: lanes 0..49 of wave 3, entry 5q + q2 split over two lanes (even / odd factors)
    const int be = (lane >> 1) < 25 ? lane >> 1 : 0, bhalf = lane & 1;
    const int bq = be / 5, bq2 = be % 5;
    const double blk0 = (double)Pb[(size_t)set[bq] * ld + set[bq2]] * (bhalf == 0 ? 1.0 : 0.0);
    // state entries (uniform)
    const double th = s[0], x = s[1], y = s[2], slx = s[c], sly = s[c + 1];

    if (tid < 2 * kMaxPending * 5) { sUs[ef][eq] = stU; sVs[ef][eq] = stV; }
    __syncthreads();

    // ---- phase A, three roles in parallel
    // The innovation needs ~8 dependent double-precision transcendentals (cartesian2polar of the marker, z_hat of
    // the landmark): a ~4 us latency chain on one wave.  Its two halves are independent unless the landmark is being
    // initialised, so wave 1 converts the marker while wave 2 evaluates the expected measurement.
    if (wave == 1 && !d.skip) {
        double r_obs, phi_obs;
        fetch_obs(o, b, r_obs, phi_obs);                  // slam.cpp:286
        if (lane == 0) { sShare[18] = r_obs; sShare[19] = phi_obs; }
        if (d.init) {                                     // initializeLandmark, slam_library.cpp:255-261
            const double lx1 = x + r_obs * cos(phi_obs + th), ly1 = y + r_obs * sin(phi_obs + th);
            double zr, zb;
            measurement(th, x, y, lx1, ly1, zr, zb);      // :265
            if (lane == 0) { sShare[14] = lx1; sShare[15] = ly1; sShare[16] = zr; sShare[17] = zb; }
        }
    }
    if (wave == 2 && !d.skip && !d.init) {
        double zr, zb;
        measurement(th, x, y, slx, sly, zr, zb);          // :265
        if (lane == 0) { sShare[14] = slx; sShare[15] = sly; sShare[16] = zr; sShare[17] = zb; }
    }
    if (wave == 3 && lane < 50) {
        double acc = blk0;
        for (int f = bhalf; f < F2; f += 2) acc = fma(-sUs[f][bq2], sVs[f][bq], acc);
        acc = acc + __shfl_xor(acc, 1, 64);
        if (bhalf == 0) sBlk[bq][bq2] = acc;
    }
    // C_q(t) = P_j(t, set[q]),  R_q(t) = P_j(set[q], t); factors beyond F2 are zeros in uo / wo
#pragma unroll
    for (int f = 0; f < 2 * kMaxPending; ++f) {
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            Cq[q] = fma(-uo[f], sVs[f][q], Cq[q]);
            Rq[q] = fma(-sUs[f][q], wo[f], Rq[q]);
        }
    }
    __syncthreads();
    if (wave == 0) {      // H, S, S^-1 (redundantly on every lane), published by lane 0
        bool skip0 = d.skip;
        int st = d.new_status;
        double Hc0[10], Si0[4];
#pragma unroll
        for (int q = 0; q < 10; ++q) Hc0[q] = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) Si0[q] = 0;
        if (!skip0) {
            double pb[5][5], S[4];
#pragma unroll
            for (int q = 0; q < 5; ++q)
#pragma unroll
                for (int q2 = 0; q2 < 5; ++q2) pb[q][q2] = sBlk[q][q2];
            jacobian_compact(x, y, sShare[14], sShare[15], Hc0);   // :268
            innovation_cov_block(pb, Hc0, v.R, S);                 // :270
            if (inv2(S, Si0)) { skip0 = true; if (st == 0) st = kStatusSingular; }
        }
        const double dz0 = sShare[18] - sShare[16];       // :272, bearing innovation not wrapped
        const double dz1 = sShare[19] - sShare[17];
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 10; ++q) sShare[q] = Hc0[q];
#pragma unroll
            for (int q = 0; q < 4; ++q) sShare[10 + q] = Si0[q];
            sFlag[0] = skip0 ? 1 : 0; sFlag[1] = st;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): every lane has read sShare[16..19]
        if (lane == 0) { sShare[16] = dz0; sShare[17] = dz1; }
    }
    __syncthreads();
    const bool skip = sFlag[0] != 0;
    if (blockIdx.x == 0 && tid == 0) {
        int* co = v.c_out + b * C_WORDS;
        co[C_SEEN] = d.new_seen; co[C_SEEN_CACHED] = cached; co[C_BRK] = d.new_brk; co[C_STATUS] = sFlag[1];
        if (v.id_log && o.log_slot >= 0) v.id_log[(size_t)b * v.log_stride + o.log_slot] = d.id;
    }
    if (t >= ld) return;
    double* Un = Ub + (size_t)F2 * ld;
    double* Vn = Vb + (size_t)F2 * ld;
    double sv = live ? s[t] : 0.0;
    if (!d.skip && d.init && t == c) sv = sShare[14];
    if (!d.skip && d.init && t == c + 1) sv = sShare[15];
    if (skip || !live) {                                  // a skipped correction is a zero factor
        Un[t] = 0.0; Un[ld + t] = 0.0; Vn[t] = 0.0; Vn[ld + t] = 0.0;
        so[t] = sv;
        return;
    }
    double Hc[10], Sinv[4];
#pragma unroll
    for (int q = 0; q < 10; ++q) Hc[q] = sShare[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) Sinv[q] = sShare[10 + q];
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

// Pout <- Pin - sum_f U_f V_f (f < 2J) (ping-pong like k_update): one streaming pass over P (2 len^2 w bytes) that is a rank-2J GEMM
// update, so it runs on the matrix cores: v_mfma_f64_16x16x4_f64 with the P tile as the accumulator.  (With VALU
// FMAs every product needs a broadcast operand; the LDS/readlane traffic for that made the pass LDS-bound at 20 us.)
//
// MFMA roles: D[m][n] += A[m][k] B[k][n] with n (lane & 15) <-> a ROW of P, m (lane >> 4) + 4 reg <-> a COLUMN, so a
// lane's 16-byte load/store covers VEC consecutive rows of one column: lane l of row block rb holds rows
// i0 + 16 VEC rb + VEC (l & 15) + s (s < VEC: VEC interleaved 16-row sub-tiles) of columns j0 + (l >> 4) + 4 r.
//   A operand = V(f = k0 + (l >> 4), column j0 + (l & 15))        8-byte global load, shared by all row blocks
//   B operand = -U(f = k0 + (l >> 4), row of (l & 15), s)          from the workgroup's LDS copy of its U rows
// A workgroup's four waves share the same 64 VEC rows and own four adjacent 16-column strips.
typedef double mfma_d4 __attribute__((ext_vector_type(4)));

template <typename T>
__global__ __launch_bounds__(256) void k_flush(View v, int J, const T* __restrict__ Pin, T* __restrict__ Pout,
                                               const double* __restrict__ U, const double* __restrict__ V)
{
    typedef Pack16<T> vec_t;
    constexpr int VEC = 16 / sizeof(T);
    constexpr int CW = 16, RB = 4, ROWS = 64 * VEC, KS = 2 * kMaxPending / 4;
    constexpr int UPAD = 2;                                // de-phase consecutive factor rows in the LDS banks
    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int ld = v.ld, L = v.L;
    const int rbase = blockIdx.x * ROWS;
    const int strip = blockIdx.y * 4 + wave;
    const bool active = strip * CW < L;
    const int j0 = active ? strip * CW : 0;
    const int n16 = lane & 15, g4 = lane >> 4;
    const int F2 = 2 * J;
    const double* Ub = U + (size_t)b * 2 * kMaxPending * ld;
    const double* Vb = V + (size_t)b * 2 * kMaxPending * ld;
    __shared__ double sU[2 * kMaxPending][ROWS + UPAD];

    // the tile: RB x 4 loads of 16 bytes per lane (clamped addresses; stores are guarded)
    const T* Pb = Pin + (size_t)b * v.p_stride;
    T* Po = Pout + (size_t)b * v.p_stride;
    vec_t p[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rbase + 16 * VEC * rb + VEC * n16;
            const int col = j0 + g4 + 4 * r;
            p[rb][r] = *reinterpret_cast<const vec_t*>(Pb + (size_t)(col < L ? col : 0) * ld + (row < ld ? row : 0));
        }
    // A operands for all k-steps (V is contiguous along the column index)
    // (every load below is unconditional with a clamped index and masked afterwards: a load inside a branch or a
    // runtime-trip loop is not hoisted by the compiler and each one then exposes its full latency)
    double vop[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int f = 4 * ks + g4;
        const int col = j0 + n16;
        // masked by multiplication, not by a select: hipcc turns `cond ? load : 0` back into a branch around the
        // load, and every such branch ends in s_waitcnt vmcnt(0) -- eight serialised round trips behind the tile loads
        const double w = Vb[(size_t)(f < F2 ? f : 0) * ld + (col < L ? col : 0)];
        vop[ks] = w * ((f < F2 && col < L) ? 1.0 : 0.0);
    }
    // the workgroup's rows of U (zero beyond the pending factors / beyond ld)
    constexpr int NSTAGE = 2 * kMaxPending * (ROWS / 2) / 256;
    double2 ust[NSTAGE];
#pragma unroll
    for (int k = 0; k < NSTAGE; ++k) {
        const int e = tid + k * 256;
        const int f = e / (ROWS / 2), i2 = (e % (ROWS / 2)) * 2;
        const bool ok = f < F2 && rbase + i2 < ld;
        ust[k] = *reinterpret_cast<const double2*>(Ub + (size_t)(ok ? f : 0) * ld + (ok ? rbase + i2 : 0));
        const double mk = ok ? 1.0 : 0.0;
        ust[k].x *= mk; ust[k].y *= mk;
    }
#pragma unroll
    for (int k = 0; k < NSTAGE; ++k) {
        const int e = tid + k * 256;
        const int f = e / (ROWS / 2), i2 = (e % (ROWS / 2)) * 2;
        sU[f][i2] = ust[k].x; sU[f][i2 + 1] = ust[k].y;
    }
    __syncthreads();
    if (!active) return;

#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        mfma_d4 acc[VEC];
#pragma unroll
        for (int sidx = 0; sidx < VEC; ++sidx)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[sidx][r] = (double)p[rb][r].v[sidx];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {     // factors beyond the pending ones are zero operands: no branch
            const double* up = &sU[4 * ks + g4][16 * VEC * rb + VEC * n16];
#pragma unroll
            for (int sidx = 0; sidx < VEC; ++sidx)
                acc[sidx] = __builtin_amdgcn_mfma_f64_16x16x4f64(vop[ks], -up[sidx], acc[sidx], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            vec_t out;
#pragma unroll
            for (int sidx = 0; sidx < VEC; ++sidx) out.v[sidx] = (T)acc[sidx][r];
            const int row = rbase + 16 * VEC * rb + VEC * n16;
            const int col = j0 + g4 + 4 * r;
            if (row < ld && col < L) *reinterpret_cast<vec_t*>(Po + (size_t)col * ld + row) = out;
        }
    }
}

} // namespace nuslam
