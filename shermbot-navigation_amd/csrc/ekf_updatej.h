// ekf_updatej.h -- J consecutive corrections in one pass over the covariance (k_update2 generalised).
//
// Markers i .. i+J-1 of a known-id tick, each a plain ExtendedKalman::update (slam_library.cpp:263-282) of an
// initialised landmark: P_J = (I - K_J H_J) ... (I - K_1 H_1) P_0.  Correction s only needs, from P_{s-1}, the rows and
// columns at set_s = {0,1,2,c_s,c_s+1} and the 5x5 block there.  With U = {0,1,2} U {c_s, c_s+1 : s} (NU = 3 + 2J
// indices) every wave carries, besides its tile, the NU columns of P at its rows and the NU rows of P at its columns,
// and moves ALL of them from P_{s-1} to P_s with the one sweep formula (p1_entry: the sequential arithmetic, rounded to
// the storage type exactly where k_update would have stored).  Wave 0 does the same for the NU x NU block and the NU
// state entries and produces each correction's head (H_s, S_s^-1).  One read and one write of P per J corrections;
// results bit-identical to J k_update launches (tests/test_gpu_pair.py).
//
// Per correction s:  phase A  wave 3: z_hat_s at the current state | wave 0: H_s, S_s^-1, K_s and M_s at the rows U
//                    barrier  phase B  wave 0: innovation, state_s at U, block -> P_s, tables to LDS
//                    barrier  phase C  all waves: M_s at their rows; their U columns, U rows and tile -> P_s
// The LDS tables are double-buffered by the parity of s, so phase A/B of correction s+1 never overwrites what a slower
// wave is still reading in phase C of correction s.
#pragma once

namespace nuslam {

template <int J> struct UJ {
    static constexpr int NU = 3 + 2 * J;
    static constexpr int NREG = (NU + 3) / 4;         // lane-distributed row-strip registers (4 rows x 16 columns each)
    static constexpr int NBLK = (NU * NU + 63) / 64;  // lane-distributed block registers (wave 0)
};

struct ObsJ {                      // up to 4 inline markers + ids (host passes them in the kernel arguments)
    const double* a; const double* b;   // trace arrays (marker x / y) or null -> inline a0/b0
    long long stride, off;              // per-filter stride, offset of the FIRST marker (the others follow consecutively)
    double a0[4], b0[4];
    int id[4];
    int cartesian;
    int log_slot;                       // slot of the first marker in id_log, or -1
};

// per-parity LDS tables (doubles)
template <int J> struct TJ {
    static constexpr int NU = UJ<J>::NU;
    static constexpr int HC = 0, SI = 10, DZ = 14, ZH = 16, MU = 18 /* [NU][5] */, RU = MU + 5 * NU /* [5][NU] */,
                         WORDS = RU + 5 * NU;
};

template <typename T, int J>
__global__ __launch_bounds__(256, 2) void k_updatej(View v, ObsJ o, const T* __restrict__ Pin, T* __restrict__ Pout)
{
    constexpr int CW = 16;
    constexpr int NU = UJ<J>::NU, NREG = UJ<J>::NREG, NBLK = UJ<J>::NBLK;
    typedef TJ<J> L_;
    typedef Pack16<T> vec_t;
    constexpr int VEC = 16 / sizeof(T);
    const int b = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ld = v.ld, L = v.L;
    const int row0 = (blockIdx.x * 64 + lane) * VEC;
    const int strip = blockIdx.y * 4 + wave;
    const bool active = strip * CW < L;
    const int j0 = active ? strip * CW : 0;
    const bool rows_ok = row0 < ld;
    const int rowc = rows_ok ? row0 : 0;
    const int ncol = (L - j0) < CW ? (L - j0) : CW;

    __shared__ double tab[2][L_::WORDS];
    __shared__ double sh_state[NU];                // state at U, kept current by wave 0
    __shared__ double sh_obs[2 * J];               // markers in polar form
    __shared__ int sh_flag[J + 1];                 // singular flag per correction, latched status

    int U[NU];
    U[0] = 0; U[1] = 1; U[2] = 2;
#pragma unroll
    for (int s = 0; s < J; ++s) { U[3 + 2 * s] = 3 + 2 * (o.id[s] - 1); U[4 + 2 * s] = U[3 + 2 * s] + 1; }
    const double* sv_in = v.s_in + (size_t)b * ld;
    double* so = v.s_out + (size_t)b * ld;
    const T* Pb = Pin + (size_t)b * v.p_stride;

    // ---- one burst of loads
    double blk[NBLK], st_u = 0;                    // wave 0: entry e = NU a + b' of the block P(U[a], U[b']) lives in
    if (wave == 0) {                               //         register e / 64, lane e % 64; lane k < NU holds state[U[k]]
#pragma unroll
        for (int r = 0; r < NBLK; ++r) {
            const int e = 64 * r + lane;
            const int a = e < NU * NU ? e / NU : 0, bb = e < NU * NU ? e % NU : 0;
            blk[r] = (double)Pb[(size_t)U[bb] * ld + U[a]];
        }
        st_u = sv_in[U[lane < NU ? lane : 0]];
    }
    // the NU rows of P at this wave's columns: register r, lane 16 q + jj holds P(U[4 r + q], j0 + jj)
    const int sj = lane & 15, sq = lane >> 4;
    const int sjc = sj < ncol ? sj : 0;
    const T* colp = Pb + (size_t)(j0 + sjc) * ld;
    double rs[NREG];
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
        const int a = 4 * r + sq;
        rs[r] = (double)colp[U[a < NU ? a : 0]];
    }
    const T* Pr = Pb + (size_t)j0 * ld + rowc;
    vec_t p[CW];
#pragma unroll
    for (int jj = 0; jj < CW; ++jj) p[jj] = *reinterpret_cast<const vec_t*>(Pr + (size_t)(jj < ncol ? jj : 0) * ld);
    vec_t pcU[NU];                                 // the NU columns of P at this lane's rows
#pragma unroll
    for (int k = 0; k < NU; ++k) pcU[k] = *reinterpret_cast<const vec_t*>(Pb + (size_t)U[k] * ld + rowc);
    double svr[VEC];                               // running state entries of this lane's rows (state owners only)
    const bool owns_state = (strip == 0);
#pragma unroll
    for (int e = 0; e < VEC; ++e) svr[e] = (owns_state && rows_ok) ? sv_in[row0 + e] : 0.0;

    // markers in polar form: waves 1 and 2 share them (only the trace is needed); state at U to LDS
    if (wave == 1 || wave == 2) {
#pragma unroll
        for (int s = 0; s < J; ++s)
            if ((s & 1) == wave - 1) {
                const double xa = o.a ? o.a[b * o.stride + o.off + s] : o.a0[s];
                const double xb = o.b ? o.b[b * o.stride + o.off + s] : o.b0[s];
                double r, f;
                if (o.cartesian) cartesian2polar(xa, xb, r, f); else { r = xa; f = xb; }
                if (lane == 0) { sh_obs[2 * s] = r; sh_obs[2 * s + 1] = f; }
            }
    }
    if (wave == 0 && lane < NU) sh_state[lane] = st_u;
    int status = (wave == 0) ? v.c_in[b * C_WORDS + C_STATUS] : 0;
    __syncthreads();

#pragma unroll
    for (int s = 0; s < J; ++s) {
        double* tb = tab[s & 1];
        const int cs = U[3 + 2 * s];
        int us[5];                                  // positions of set_s in U
        us[0] = 0; us[1] = 1; us[2] = 2; us[3] = 3 + 2 * s; us[4] = 4 + 2 * s;
        const int sets[5] = { 0, 1, 2, cs, cs + 1 };

        // ---------------- phase A
        double Ka[2] = { 0, 0 };
        int sing = 0;
        if (wave == 3) {
            double zr, zb;
            measurement(sh_state[0], sh_state[1], sh_state[2], sh_state[3 + 2 * s], sh_state[4 + 2 * s], zr, zb);   // :265
            if (lane == 0) { tb[L_::ZH] = zr; tb[L_::ZH + 1] = zb; }
        } else if (wave == 0) {
            auto bget = [&](int a, int bb) {        // P_cur(U[a], U[bb]) broadcast from the lane-distributed block
                const int e = NU * a + bb;
                return lane_bcast(blk[e / 64], e % 64);
            };
            auto bshfl = [&](int e) {               // entry e of the block, per-lane index (every lane executes it)
                double x = __shfl(blk[0], e % 64, 64);
#pragma unroll
                for (int r = 1; r < NBLK; ++r) { const double xr = __shfl(blk[r], e % 64, 64); x = (e / 64 == r) ? xr : x; }
                return x;
            };
            double Hc[10], Si[4], pb[5][5], S[4];
#pragma unroll
            for (int q = 0; q < 5; ++q)
#pragma unroll
                for (int q2 = 0; q2 < 5; ++q2) pb[q][q2] = bget(us[q2], us[q]);
            jacobian_compact(sh_state[1], sh_state[2], sh_state[3 + 2 * s], sh_state[4 + 2 * s], Hc);            // :268
            innovation_cov_block(pb, Hc, v.R, S);                                                                 // :270
            sing = inv2(S, Si);
            if (sing) {                             // this correction becomes a no-op: K = 0
                if (status == 0) status = kStatusSingular;
#pragma unroll
                for (int q = 0; q < 4; ++q) Si[q] = 0.0;
            }
            // K_s and M_s at row U[lane]; the prior rows R_s(k, U[lane]) for everybody's column updates
            const int a = lane < NU ? lane : 0;
            double pc[5], m[5], rk[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) pc[q] = bshfl(NU * a + us[q]);          // P_cur(U[a], set_s[q])
#pragma unroll
            for (int k = 0; k < 5; ++k) rk[k] = bshfl(NU * us[k] + a);          // P_cur(set_s[k], U[a])
            gain_row(pc, Hc, Si, U[a], sets, Ka, m);
            if (lane < NU) {
#pragma unroll
                for (int q = 0; q < 5; ++q) { tb[L_::MU + 5 * lane + q] = m[q]; tb[L_::RU + NU * q + lane] = rk[q]; }
            }
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 10; ++q) tb[L_::HC + q] = Hc[q];
#pragma unroll
                for (int q = 0; q < 4; ++q) tb[L_::SI + q] = Si[q];
                sh_flag[s] = sing;
            }
        }
        __syncthreads();

        // ---------------- phase B (wave 0): innovation, state at U, block -> P_s
        if (wave == 0) {
            const double dz0 = sing ? 0.0 : sh_obs[2 * s] - tb[L_::ZH], dz1 = sing ? 0.0 : sh_obs[2 * s + 1] - tb[L_::ZH + 1];   // :272
            double acc = 0.0;
            acc = fma(Ka[0], dz0, acc);
            acc = fma(Ka[1], dz1, acc);
            double sn = st_u + acc;                                  // :275
            if (lane == 0 && !sing) sn = normalize_angle(sn);        // :276
            st_u = sn;
            if (lane < NU) sh_state[lane] = sn;
            if (lane == 0) { tb[L_::DZ] = dz0; tb[L_::DZ + 1] = dz1; }
            // every block entry moves to P_s with the sweep formula
#pragma unroll
            for (int r = 0; r < NBLK; ++r) {
                const int e = 64 * r + lane;
                const bool ok = e < NU * NU;
                const int a = ok ? e / NU : 0, bb = ok ? e % NU : 0;
                double mrow[5], rcol[5];
#pragma unroll
                for (int q = 0; q < 5; ++q) { mrow[q] = tb[L_::MU + 5 * a + q]; rcol[q] = tb[L_::RU + NU * q + bb]; }
                const int i = U[a];
                const double bef = (i > 2 && i < cs) ? 1.0 : 0.0, aft = (i > cs + 1) ? 1.0 : 0.0;
                const double nv = p1_entry<T>(mrow, rcol, blk[r], bef, aft);
                blk[r] = ok ? nv : blk[r];
            }
        }
        __syncthreads();
        if (!active) continue;

        // ---------------- phase C (all waves with a tile)
        const double* Hc = tb + L_::HC;             // wave-uniform LDS reads keep 28 doubles out of the VGPRs
        const double* Si = tb + L_::SI;
        const int sing_s = sh_flag[s];
        double m[VEC][5], bef[VEC], aft[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int i = row0 + e;
            double pc[5], K[2];
#pragma unroll
            for (int q = 0; q < 5; ++q) pc[q] = (double)pcU[us[q]].v[e];
            gain_row(pc, Hc, Si, i, sets, K, m[e]);
            bef[e] = ((i > 2) && (i < cs)) ? 1.0 : 0.0;
            aft[e] = (i > cs + 1) ? 1.0 : 0.0;
            if (owns_state) {
                double acc = 0.0;
                acc = fma(K[0], tb[L_::DZ], acc);
                acc = fma(K[1], tb[L_::DZ + 1], acc);
                svr[e] = svr[e] + acc;
                if (i == 0 && !sing_s) svr[e] = normalize_angle(svr[e]);
            }
            // this row's U columns -> P_s (only those a later correction still reads, and 0, 1, 2)
#pragma unroll
            for (int k = 0; k < NU; ++k) {
                if (k < 3 || k >= 3 + 2 * (s + 1)) {
                    double rcol[5];
#pragma unroll
                    for (int q = 0; q < 5; ++q) rcol[q] = tb[L_::RU + NU * q + k];
                    pcU[k].v[e] = (T)sweep_entry(m[e], rcol, (double)pcU[k].v[e], bef[e], aft[e]);
                }
            }
        }
        // prior rows of this wave's columns for correction s, then the U rows -> P_s
        double r_of[5];                              // this lane's column sj: R_s(q, j)
#pragma unroll
        for (int q = 0; q < 5; ++q) r_of[q] = __shfl(rs[us[q] / 4], 16 * (us[q] % 4) + sj, 64);
        double rs_new[NREG];
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            const int a = 4 * r + sq;
            const int ac = a < NU ? a : 0;
            double mrow[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) mrow[q] = tb[L_::MU + 5 * ac + q];
            const int i = U[ac];
            const double bf = (i > 2 && i < cs) ? 1.0 : 0.0, af = (i > cs + 1) ? 1.0 : 0.0;
            rs_new[r] = p1_entry<T>(mrow, r_of, rs[r], bf, af);
        }
        // the tile
#pragma unroll
        for (int jj = 0; jj < CW; ++jj) {
            double rcol[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) rcol[q] = lane_bcast(rs[us[q] / 4], 16 * (us[q] % 4) + jj);
#pragma unroll
            for (int e = 0; e < VEC; ++e) p[jj].v[e] = (T)sweep_entry(m[e], rcol, (double)p[jj].v[e], bef[e], aft[e]);
        }
#pragma unroll
        for (int r = 0; r < NREG; ++r) rs[r] = rs_new[r];
    }

    if (wave == 0 && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
        const int* ci = v.c_in + b * C_WORDS;
        int* co = v.c_out + b * C_WORDS;
        co[C_SEEN] = ci[C_SEEN]; co[C_SEEN_CACHED] = ci[C_SEEN_CACHED]; co[C_BRK] = ci[C_BRK]; co[C_STATUS] = status;
        if (v.id_log && o.log_slot >= 0)
            for (int s = 0; s < J; ++s) v.id_log[(size_t)b * v.log_stride + o.log_slot + s] = o.id[s];
    }
    if (!active || !rows_ok) return;
    if (owns_state)
#pragma unroll
        for (int e = 0; e < VEC; ++e) so[row0 + e] = svr[e];
    T* Pw = Pout + (size_t)b * v.p_stride + (size_t)j0 * ld + row0;
#pragma unroll
    for (int jj = 0; jj < CW; ++jj)
        if (jj < ncol) *reinterpret_cast<vec_t*>(Pw + (size_t)jj * ld) = p[jj];
}

} // namespace nuslam
