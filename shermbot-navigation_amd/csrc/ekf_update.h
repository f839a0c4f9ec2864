// ekf_update.h -- ExtendedKalman::update (nuslam/src/slam_library.cpp:263-282) fused with the caller's decision
// chain (nuslam/src/slam.cpp:295-318) as ONE streaming kernel over the covariance.  Included by ekf_kernels.h.
#pragma once

namespace nuslam {

// The caller's decision chain for one marker, slam.cpp:295-318, plus id validation.  Every workgroup evaluates it
// from the same read-only inputs (ctrl words, cur_id, the observation), so no hand-off between workgroups is needed.
struct Decision {
    int id, c;
    bool skip, init;
    int new_seen, new_brk, new_status;
};

__device__ inline Decision resolve(int n, int id, int seen, int cached, int brk, int status, int mode,
                                   int total_landmarks)
{
    Decision d;
    d.id = id;
    d.new_seen = seen; d.new_brk = brk; d.new_status = status;
    d.skip = false; d.init = false;
    if (mode == MODE_FORCE) {
        if (d.id < 1 || d.id > n) { d.skip = true; if (d.new_status == 0) d.new_status = kStatusBounds; }
    } else if (brk) {
        d.skip = true; d.id = 0;                                              // marker loop already left (:315)
    } else if (d.id > n) {
        d.skip = true; if (d.new_status == 0) d.new_status = kStatusBounds;   // initializeLandmark would index out of bounds
    } else {
        if (mode == MODE_KNOWN && d.id > seen) d.new_seen = d.id;             // what associateLandmark would have counted
        if (d.id > cached) d.init = true;                                     // :295-297
        else if (d.id < 0) d.skip = true;                                     // :298-300
        else if (d.id > total_landmarks) { d.skip = true; d.new_brk = 1; }    // :301-316
        if (!d.skip && d.id < 1) { d.skip = true; if (d.new_status == 0) d.new_status = kStatusBounds; }
    }
    d.c = d.skip ? 3 : 3 + 2 * (d.id - 1);
    return d;
}

// S = H P H^T + R from a preloaded 5x5 block pb[q][q2] = P(set[q2], set[q])  (slam_library.cpp:270), ascending-k sums
__device__ inline void innovation_cov_block(const double pb[5][5], const double Hc[10], const double R[4], double S[4])
{
    double HPs[2][5];
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            double acc = 0.0;
#pragma unroll
            for (int q2 = 0; q2 < 5; ++q2) acc = fma(Hc[r + 2 * q2], pb[q][q2], acc);
            HPs[r][q] = acc;
        }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 5; ++q) acc = fma(HPs[r][q], Hc[s2 + 2 * q], acc);
            S[r + 2 * s2] = acc + R[r + 2 * s2];
        }
}

// broadcast lane `src` (compile-time constant after unrolling) of a per-lane double to the whole wave
__device__ inline double lane_bcast(double val, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(val), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(val), src);
    return __hiloint2double(hi, lo);
}

// P is ping-ponged: the kernel reads Pin (never written here) and writes Pout, so every wave can recompute the
// small shared quantities itself -- H, S = H P H^T + R from the 5x5 block P[set,set], S^-1 -- and derive what its
// own tile needs: the Kalman-gain rows K(i,:) = (P H^T)(i,:) S^-1 for its rows (five 16-byte column loads per
// lane) and the five prior rows R_q(j) = P(set[q], j) for its columns.
//
//   P'(i,j) = sum_k M(i,k) P(k,j),  M = I - K H,  k ascending over {0,1,2} U {i} U {c,c+1}
//           = M0 R0 + M1 R1 + M2 R2 [+ P(i,j) if 2 < i < c] + Mc Rc + Mc1 Rc1 [+ P(i,j) if i > c+1]
//
// which is the reference's (eye - K*H) * covariance with the exactly-zero terms dropped (bit-identical to the
// oracle).  HBM traffic: every element of P is read once and written once = 2 len^2 w bytes (the roofline figure);
// the strips add 10 len w.  One wave owns 64*VEC consecutive rows x CW columns; a lane moves 16 bytes per column.
//
// Everything here is latency (the arithmetic is small), so ALL global loads of a wave are issued in one burst
// before the first wait, as vector loads: the wave-uniform inputs (pose, landmark, the 5x5 block, the 5 x CW
// prior-row strip) are fetched one element per lane and broadcast with v_readlane instead of chains of dependent
// scalar loads.  With the landmark id passed inline every address depends only on kernel arguments.
// MODE and INLINE_ID are compile-time so that the known-association fast path (id inside the kernel arguments)
// contains no load at all in front of the burst.
// WAVES per workgroup: 8 = one head chain per CU when the grid is one resident generation, else 4 (see ekf_update2.h)
template <typename T, int CW, int MODE, bool INLINE_ID, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_update(View v, ObsArg o, int total_landmarks,
                                                const T* __restrict__ Pin, T* __restrict__ Pout)
{
    constexpr int mode = MODE;
    static_assert(CW == 16, "the prior-row strip is fetched as 16 columns x {3, 2} rows per vector load");
    typedef Pack16<T> vec_t;
    constexpr int VEC = 16 / sizeof(T);
    const int b = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ld = v.ld, L = v.L;
    const int row0 = (blockIdx.x * 64 + lane) * VEC;
    const int strip = blockIdx.y * WAVES + wave;
    const bool active = strip * CW < L;                   // wave-uniform; idle waves only keep the barrier company
    const int j0 = active ? strip * CW : 0;
    const bool rows_ok = row0 < ld;
    const int rowc = rows_ok ? row0 : 0;                  // clamped: loads stay unconditional, stores are guarded
    const int ncol = (L - j0) < CW ? (L - j0) : CW;       // wave-uniform

    // what wave 0 of the workgroup computes once for all four waves: Hc[10], Sinv[4], lx, ly, th, x, y
    __shared__ double sh_d[24];                           // [20..23]: marker (r, phi) and z_hat of a plain correction
    __shared__ int sh_i[2];                               // skip, latched status

    // (1) control words and the landmark column.  cg is always safe to read; it equals the landmark's column
    // whenever the decision below keeps the update.
    const int* ci = v.c_in + b * C_WORDS;
    const int seen = ci[C_SEEN], cached = ci[C_SEEN_CACHED], brk = ci[C_BRK], status0 = ci[C_STATUS];
    int seen_now = seen, status_now = status0;
    int id_raw;
    if (MODE == MODE_DA) {                                // associateLandmark's verdict, from the reduced key
        const Assoc a = decode_association(v.n, seen, brk, status0, v.akey[2 * b + v.aslot]);
        id_raw = a.id; seen_now = a.new_seen; status_now = a.new_status;
        // a generated trace has a fixed number of marker slots per tick; a slot without a marker (presence word < 0)
        // is a marker the node never received: associateLandmark is not called for it (slam.cpp:279), nothing moves
        if (o.ids != nullptr && o.ids[b * o.stride + o.off] < 0) { id_raw = -1; seen_now = seen; status_now = status0; }
    } else {
        id_raw = INLINE_ID ? o.id0 : o.ids[b * o.stride + o.off];
    }
    const int cg = (id_raw >= 1 && id_raw <= v.n) ? 3 + 2 * (id_raw - 1) : 3;
    const double* s = v.s_in + (size_t)b * ld;
    const T* Pb = Pin + (size_t)b * v.p_stride;

    // (2) one burst of vector loads, in the order they are needed (vmcnt retires in order): first, in straight-line
    // code, the few values the serial head chain consumes; nothing bulky is requested before they are back (the gate),
    // because all workgroups are resident at once and the memory system serves requests roughly in arrival order.
    //  (a) lanes 32..36 of every wave: th, x, y, lx, ly; the head wave's lanes 0..24: the 5x5 block, lane = 5q + q2
    //      holds P(set[q2], set[q]) (the other waves fetch one element instead: no branch around the load)
    const bool blk_lane = wave == 0 && lane < 25;
    const int bq = blk_lane ? lane / 5 : 0, bq2 = blk_lane ? lane % 5 : 0;
    const int cq = bq < 3 ? bq : cg + (bq - 3), cq2 = bq2 < 3 ? bq2 : cg + (bq2 - 3);
    const double v_blk = (double)Pb[(size_t)cq * ld + cq2];
    const int sl = lane - 32;
    const int si = (sl >= 0 && sl < 3) ? sl : (sl == 3 ? cg : (sl == 4 ? cg + 1 : 0));
    const double v_st = s[si];
    double raw_a, raw_b;                                  // the marker as it travels (x, y or range, bearing)
    load_obs_raw(o.a, o.b, o.stride, o.off, o.a0, o.b0, b, s, raw_a, raw_b);
    //      this lane's state rows (fp64 storage only; the fp32 variant has four rows per lane and no registers to spare)
    constexpr bool kEarlyRows = (VEC == 2);
    double s_rows[VEC];
    if (kEarlyRows) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) s_rows[e] = s[rowc + e];
    }
    __builtin_amdgcn_s_waitcnt(0x0070);                   // the gate: vmcnt(0)
    //  (b) the prior rows of this wave's columns: lane = 16q + jj holds P(set[q], j0 + jj), q < 3 in vA, q - 3 < 2 in vB
    const int sj = lane & 15, sq = lane >> 4;
    const int sjc = sj < ncol ? sj : 0;
    const T* colp = Pb + (size_t)(j0 + sjc) * ld;
    const double vA = (double)colp[sq < 3 ? sq : 0];
    const double vB = (double)colp[cg + (sq & 1)];
    //  (c) the five gain columns, then the tile: CW independent 16-byte loads per lane
    vec_t pc[5];
#pragma unroll
    for (int q = 0; q < 5; ++q)
        pc[q] = *reinterpret_cast<const vec_t*>(Pb + (size_t)(q < 3 ? q : cg + (q - 3)) * ld + rowc);
    const T* Pr = Pb + (size_t)j0 * ld + rowc;
    vec_t p[CW];
#pragma unroll
    for (int jj = 0; jj < CW; ++jj) p[jj] = load_stream(Pr + (size_t)(jj < ncol ? jj : 0) * ld);   // read once: streaming

    // (3) decision (every wave: a few scalar compares) + the shared quantities (wave 0, then LDS)
    const Decision d = resolve(v.n, id_raw, seen_now, cached, brk, status_now, mode, total_landmarks);
    double* so = v.s_out + (size_t)b * ld;
    const int c = d.c;                                    // == cg unless skipped
    const int set[5] = { 0, 1, 2, c, c + 1 };
    if (wave == 0) {
        bool skip0 = d.skip;
        int st = d.new_status;
        double Hc0[10], Si0[4], lx0 = 0, ly0 = 0;
#pragma unroll
        for (int q = 0; q < 10; ++q) Hc0[q] = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) Si0[q] = 0;
        const double th = lane_bcast(v_st, 32), x = lane_bcast(v_st, 33), y = lane_bcast(v_st, 34);
        if (!skip0) {
            if (d.init) {                                 // initializeLandmark, slam_library.cpp:255-261
                double r, phi;
                obs_polar(o, raw_a, raw_b, r, phi);
                lx0 = x + r * cos(phi + th);
                ly0 = y + r * sin(phi + th);
            } else { lx0 = lane_bcast(v_st, 35); ly0 = lane_bcast(v_st, 36); }
            double pb[5][5], S[4];
#pragma unroll
            for (int q = 0; q < 5; ++q)
#pragma unroll
                for (int q2 = 0; q2 < 5; ++q2) pb[q][q2] = lane_bcast(v_blk, 5 * q + q2);
            jacobian_compact(x, y, lx0, ly0, Hc0);        // :268
            innovation_cov_block(pb, Hc0, v.R, S);        // H P H^T + R, :270
            if (inv2(S, Si0)) { skip0 = true; if (st == 0) st = kStatusSingular; }
        }
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 10; ++q) sh_d[q] = Hc0[q];
#pragma unroll
            for (int q = 0; q < 4; ++q) sh_d[10 + q] = Si0[q];
            sh_d[14] = lx0; sh_d[15] = ly0; sh_d[16] = th; sh_d[17] = x; sh_d[18] = y;
            sh_i[0] = skip0 ? 1 : 0; sh_i[1] = st;
        }
    } else if (blockIdx.y == 0 && !d.skip && !d.init) {
        // The innovation z - z_hat (five atan2, sin/cos pairs) only feeds the state correction, which column strip 0's
        // waves apply.  For a plain correction it needs nothing the head produces, so two otherwise idle waves of
        // those workgroups form it while the head chain runs (it used to be their tail, after the barrier).
        if (wave == 1) {
            double r, phi;
            obs_polar(o, raw_a, raw_b, r, phi);
            if (lane == 0) { sh_d[20] = r; sh_d[21] = phi; }
        } else if (wave == 2) {
            double zr, zb;                                // :265
            measurement(lane_bcast(v_st, 32), lane_bcast(v_st, 33), lane_bcast(v_st, 34), lane_bcast(v_st, 35),
                        lane_bcast(v_st, 36), zr, zb);
            if (lane == 0) { sh_d[22] = zr; sh_d[23] = zb; }
        }
    }
    __syncthreads();
    if (!active) return;
    const bool skip = sh_i[0] != 0;
    const int new_status = sh_i[1];
    double Hc[10], Sinv[4];
#pragma unroll
    for (int q = 0; q < 10; ++q) Hc[q] = sh_d[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) Sinv[q] = sh_d[10 + q];
    const double lx = sh_d[14], ly = sh_d[15];
    // The innovation z - z_hat (five atan2, sin/cos pairs) only feeds the state correction, which the waves of
    // column strip 0 apply; every other wave skips the transcendentals (wave-uniform branch).
    const bool owns_state = (strip == 0);
    double dz0 = 0, dz1 = 0;
    if (owns_state && !skip) {
        double r, phi, zr, zb;
        if (d.init) {                                     // z_hat at the landmark the head wave just initialised
            obs_polar(o, raw_a, raw_b, r, phi);
            measurement(sh_d[16], sh_d[17], sh_d[18], lx, ly, zr, zb);   // :265
        } else { r = sh_d[20]; phi = sh_d[21]; zr = sh_d[22]; zb = sh_d[23]; }
        dz0 = r - zr;                                     // :272, bearing innovation not wrapped
        dz1 = phi - zb;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        int* co = v.c_out + b * C_WORDS;
        co[C_SEEN] = d.new_seen; co[C_SEEN_CACHED] = cached; co[C_BRK] = d.new_brk; co[C_STATUS] = new_status;
        if (v.id_log && o.log_slot >= 0) v.id_log[(size_t)b * v.log_stride + o.log_slot] = d.id;
        if (MODE == MODE_DA) v.akey[2 * b + (v.aslot ^ 1)] = kNoKey;     // re-arm the slot the next association uses
    }
    T* Pw = Pout + (size_t)b * v.p_stride + (size_t)j0 * ld + row0;

    if (skip) {
        // no correction: the ping-pong still has to carry P (and the state) across
        if (!rows_ok) return;
#pragma unroll
        for (int jj = 0; jj < CW; ++jj)
            if (jj < ncol) store_stream(Pw + (size_t)jj * ld, p[jj]);
        if (owns_state) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int i = row0 + e;
                double sv = kEarlyRows ? s_rows[e] : s[i];
                if (d.init && i == c) sv = lx;            // the landmark was initialised before update() threw
                if (d.init && i == c + 1) sv = ly;
                so[i] = sv;
            }
        }
        return;
    }

    // (4) this lane's rows: K(i,:) = (P H^T)(i,:) S^-1 and M(i, set[q]) = eye - K H      (:270, :279)
    double m[5][VEC];
    double before[VEC], after[VEC];       // 1.0 where P(i,j) itself enters the sum before / after the landmark terms
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const int i = row0 + e;
        double ph[2];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 5; ++q) acc = fma((double)pc[q].v[e], Hc[rr + 2 * q], acc);
            ph[rr] = acc;
        }
        double K[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            double acc = 0.0;
            acc = fma(ph[0], Sinv[0 + 2 * s2], acc);
            acc = fma(ph[1], Sinv[1 + 2 * s2], acc);
            K[s2] = acc;
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            double kh = 0.0;
            kh = fma(K[0], Hc[0 + 2 * q], kh);
            kh = fma(K[1], Hc[1 + 2 * q], kh);
            m[q][e] = (i == set[q] ? 1.0 : 0.0) - kh;
        }
        before[e] = ((i > 2) && (i < c)) ? 1.0 : 0.0;
        after[e] = (i > c + 1) ? 1.0 : 0.0;
        if (owns_state && rows_ok) {
            // state += K (z - z_hat); heading re-normalised  (:275-276).  Rows [len, ld) are zero padding: K = 0.
            double acc = 0.0;
            acc = fma(K[0], dz0, acc);
            acc = fma(K[1], dz1, acc);
            double sv = (d.init && i == c) ? lx : (d.init && i == c + 1) ? ly : (kEarlyRows ? s_rows[e] : s[i]);
            sv = sv + acc;
            if (i == 0) sv = normalize_angle(sv);
            so[i] = sv;
        }
    }

    // (5) the sweep
#pragma unroll
    for (int jj = 0; jj < CW; ++jj) {
        const double r0 = lane_bcast(vA, jj), r1 = lane_bcast(vA, 16 + jj), r2 = lane_bcast(vA, 32 + jj),
                     r3 = lane_bcast(vB, jj), r4 = lane_bcast(vB, 16 + jj);
        vec_t out;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const double pij = (double)p[jj].v[e];
            double acc = m[0][e] * r0;
            acc = fma(m[1][e], r1, acc);
            acc = fma(m[2][e], r2, acc);
            acc = fma(before[e], pij, acc);               // + 1.0 * P(i,j) or + 0.0 * P(i,j): exact either way
            acc = fma(m[3][e], r3, acc);
            acc = fma(m[4][e], r4, acc);
            acc = fma(after[e], pij, acc);
            out.v[e] = (T)acc;
        }
        if (jj < ncol && rows_ok) store_stream(Pw + (size_t)jj * ld, out);     // written once, read next by other XCDs
    }
}

} // namespace nuslam
