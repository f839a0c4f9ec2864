// ekf_update2.h -- TWO consecutive corrections in one pass over the covariance (temporal blocking of k_update).
//
// ExtendedKalman::update (slam_library.cpp:263-282) applied twice, for markers i and i+1 of a tick whose landmark ids
// the host knows: P2 = (I - K2 H2) (I - K1 H1) P0.  Everything the second correction needs from P1 -- its five rows,
// its five columns and the 5x5 block at {0,1,2,c2,c2+1}, and the state after the first correction -- is a handful of
// ENTRIES of P1, and each entry of P1 is produced by the same sweep formula from P0, M1 and the prior rows R1.  A wave
// therefore recomputes those entries itself (from strips of P0 it loads anyway) and applies both corrections to its
// tile in registers: every element of P is read once and written once per PAIR of corrections -- half the HBM bytes
// per correction -- and the arithmetic of every entry is exactly the sequential one, operation for operation, so the
// result is bit-identical to two k_update launches (asserted in tests/test_gpu_pair.py) and hence to the oracle.
//
// Restrictions (checked by the host, which falls back to k_update otherwise): known association with ids the host
// holds (inline, or per filter in the resident trace), both landmarks already initialised in every filter (no
// initializeLandmark, no skip / break branch).
#pragma once

namespace nuslam {

// the sweep formula of k_update for one entry: sum_k M(i,k) P(k,j), k ascending over {0,1,2} U {i} U {c,c+1}
__device__ inline double sweep_entry(const double m[5], const double r[5], double pij, double before, double after)
{
    double acc = m[0] * r[0];
    acc = fma(m[1], r[1], acc);
    acc = fma(m[2], r[2], acc);
    acc = fma(before, pij, acc);
    acc = fma(m[3], r[3], acc);
    acc = fma(m[4], r[4], acc);
    acc = fma(after, pij, acc);
    return acc;
}

// K(i,:) = (P H^T)(i,:) S^-1 and M(i, set[q]) = delta - (K H)(i, set[q]) from the five column entries pc[q] = P(i, set[q])
__device__ inline void gain_row(const double pc[5], const double Hc[10], const double Sinv[4], int i, const int set[5],
                                double K[2], double m[5])
{
    double ph[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 5; ++q) acc = fma(pc[q], Hc[rr + 2 * q], acc);
        ph[rr] = acc;
    }
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
        m[q] = (i == set[q] ? 1.0 : 0.0) - kh;
    }
}

// an entry of P1 as k_update leaves it in memory: rounded to the storage type
template <typename T>
__device__ inline double p1_entry(const double m[5], const double r[5], double pij, double before, double after)
{
    return (double)(T)sweep_entry(m, r, pij, before, after);
}

// Debug-only phase clock (variant builds with -DNUSLAM_PHASE_CLOCK, tools/exp_phase_clock.py): wave 0 of one
// mid-grid workgroup stamps the 100 MHz wall clock at each phase boundary.  Compiled out of the product.
#ifdef NUSLAM_PHASE_CLOCK
__device__ long long g_phase[32];
__device__ unsigned g_hwid[4096][8][2];      // HW_ID and XCC_ID of every wave
__device__ long long g_wg[4096][2];          // entry / exit stamp of every workgroup (wave 0), index = y * gridDim.x + x
#define PHASE(k)                                                                                        \
    do {                                                                                                \
        if (blockIdx.x == 3 && (blockIdx.y == 40 / WAVES || blockIdx.y == 104 / WAVES) &&   \
            blockIdx.z == 0 && (threadIdx.x & 63) == 0 &&                                                      \
            (threadIdx.x >> 6) == 0)                                                                            \
            g_phase[(blockIdx.y == 104 / WAVES ? 16 : 0) + k] = (long long)wall_clock64();                       \
    } while (0)
#else
#define PHASE(k) do { } while (0)
#endif

// Waves per workgroup.  Every workgroup recomputes the two heads (a ~4 us serial chain of wave64 fp64 instructions);
// a CU holds 8 of these waves.  As two 4-wave workgroups the CU runs two chains, and the later-dispatched workgroup
// (its waves lose instruction arbitration to the older one's sweep) finishes 4 us after the first; as ONE 8-wave
// workgroup the CU runs one chain and all eight tiles are swept together.
// The host picks WAVES = 8 when the whole grid is then resident in one generation (the single-filter, latency-bound
// case) and WAVES = 4 otherwise (many filters: throughput-bound, and 8-wave groups idle more waves at small L).

// LDS layout (doubles)
enum { S2_HC1 = 0, S2_SI1 = 10, S2_HC2 = 14, S2_SI2 = 24, S2_M1S = 28 /* [a][q] 5x5: M1(set2[a], set1[q]) */,
       S2_R1S = 53 /* [q][b] 5x5: P0(set1[q], set2[b]) */, S2_DZ1 = 78, S2_ST1 = 80 /* th,x,y,lx2,ly2 after correction 1 */,
       S2_OBS = 85 /* r1, phi1, r2, phi2 */, S2_K1S = 89 /* [a][2]: K1(set2[a], :) */, S2_NU1 = 100 /* z1 - z_hat1 */,
       S2_ZH2 = 102 /* range and un-rotated bearing of z_hat2 */, S2_WORDS = 104 };

template <typename T, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 8 / WAVES) void k_update2(View v, ObsArg o1, ObsArg o2, const T* __restrict__ Pin,
                                                 T* __restrict__ Pout)
{
    constexpr int CW = 16;
    typedef Pack16<T> vec_t;
    constexpr int VEC = 16 / sizeof(T);
    const int b = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ld = v.ld, L = v.L;
    const int row0 = (blockIdx.x * 64 + lane) * VEC;
#ifdef NUSLAM_FLIP_Y
    const int strip = (gridDim.y - 1 - blockIdx.y) * 4 + wave;       // experiment: which half is slow, the data's or the dispatch's?
#else
    const int strip = blockIdx.y * WAVES + wave;
#endif
    // Which wave carries which phase-A/B role rotates with the workgroup's dispatch generation.  Wave k of every
    // workgroup lands on SIMD k, the head chain is ~3000 wave64 fp64 instructions at 4 cycles each whatever the EXEC
    // mask, and the CU holds two of these workgroups: with the head always on wave 0 both chains queue on SIMD 0 and
    // the second workgroup of every CU finishes 4 us after the first (phase clock: exits at 10.3 / 14.6 us) while
    // SIMDs 1-3 idle.  256 = CUs of an MI355X; a different mapping costs speed, never correctness.
    const int role = wave;   // (rotation tried and measured: no gain)
    // Column strip 0 is half as wide as the others (8 columns): its wave also carries the state rows and, in workgroup
    // (0,0), the heading's second re-normalisation; with a full tile it was the last wave of every launch to finish.
    const int jstart = strip == 0 ? 0 : strip * CW - CW / 2;
    const bool active = jstart < L;
    const int j0 = active ? jstart : 0;
    const bool rows_ok = row0 < ld;
    const int rowc = rows_ok ? row0 : 0;
    const int wcol = strip == 0 ? CW / 2 : CW;
    const int ncol = (L - j0) < wcol ? (L - j0) : wcol;

    __shared__ double sh[S2_WORDS];
    __shared__ int sh_i[4];                       // singular flag of correction 1, of correction 2, status
    PHASE(0);
#ifdef NUSLAM_PHASE_CLOCK
    if (threadIdx.x == 0 && blockIdx.z == 0) g_wg[blockIdx.y * gridDim.x + blockIdx.x][0] = (long long)wall_clock64();
    if ((threadIdx.x & 63) == 0 && blockIdx.z == 0) {
        g_hwid[blockIdx.y * gridDim.x + blockIdx.x][threadIdx.x >> 6][0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        g_hwid[blockIdx.y * gridDim.x + blockIdx.x][threadIdx.x >> 6][1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif

    // ids: inline when every filter corrects the same landmarks, else this filter's entries of the resident trace
    const int id1 = o1.ids ? o1.ids[b * o1.stride + o1.off] : o1.id0;
    const int id2 = o2.ids ? o2.ids[b * o2.stride + o2.off] : o2.id0;
    const int c1 = 3 + 2 * (id1 - 1), c2 = 3 + 2 * (id2 - 1);
    const int set1[5] = { 0, 1, 2, c1, c1 + 1 };
    const int set2[5] = { 0, 1, 2, c2, c2 + 1 };
    const int U[7] = { 0, 1, 2, c1, c1 + 1, c2, c2 + 1 };        // set1[q] = U[q]; set2[q] = U[q < 3 ? q : q + 2]
    const double* s = v.s_in + (size_t)b * ld;
    double* so = v.s_out + (size_t)b * ld;
    const T* Pb = Pin + (size_t)b * v.p_stride;

    // ---- one burst of loads, in the order they are needed: vmcnt retires in order, so everything the serial head
    // chain consumes (block, state, status, the markers) is issued FIRST, in straight-line code, and is waited for with
    // vmcnt(N > 0) while the strips, the gain columns and the tile are still in flight.  (A load issued after the
    // tile, or inside a role branch, costs the head chain the whole read phase: 4.6 us at N = 1000, measured with the
    // phase clock.)
    PHASE(11);
    //   every wave: lanes 0..6 hold state[U[k]]; the head wave (role 0) also: lane 7a + b holds P0(U[a], U[b])
    auto Uat = [&](int k) { return k < 3 ? k : (k < 5 ? c1 + (k - 3) : c2 + (k - 5)); };   // U[k] for a per-lane k
    const double v_st = s[Uat(lane < 7 ? lane : 0)];
    const bool blk_lane = role == 0 && lane < 49;  // the other waves fetch one element (no branch around the load)
    const int a_blk = blk_lane ? lane / 7 : 0, b_blk = blk_lane ? lane % 7 : 0;
    const double v_blk = (double)Pb[(size_t)Uat(b_blk) * ld + Uat(a_blk)];
    const int* ci = v.c_in + b * C_WORDS;          // wave-uniform: scalar loads
    const int status_in = ci[C_STATUS], seen_in = ci[C_SEEN], cached_in = ci[C_SEEN_CACHED], brk_in = ci[C_BRK];
    //   this lane's state rows (used by column strip 0's waves only, but a late load would cost them a round trip)
    //   (fp64 storage only: with four rows per lane the fp32 variant has no registers to spare and loads them late)
    constexpr bool kEarlyRows = (VEC == 2);
    double s_rows[VEC];
    if (kEarlyRows) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) s_rows[e] = s[rowc + e];
    }
    //   the marker this wave turns into polar form (role 2: marker 2; the others: marker 1)
    double raw_a, raw_b;
    {
        const bool w2 = role == 2;                 // field by field: selecting between the two by-value structs
        load_obs_raw(w2 ? o2.a : o1.a, w2 ? o2.b : o1.b, w2 ? o2.stride : o1.stride, w2 ? o2.off : o1.off,   // would put them on the stack
                     w2 ? o2.a0 : o1.a0, w2 ? o2.b0 : o1.b0, b, s, raw_a, raw_b);
    }
#ifndef NUSLAM_NO_GATE
    // Gate: nothing bulky is requested before these few values are back.  All 512 workgroups are resident at once and
    // the memory system serves requests roughly in arrival order, so without the gate the later-dispatched half of the
    // grid gets its head inputs only after the earlier half's 16 MB of tile loads (measured with the phase clock:
    // exits at 10 us / 14.3 us); with it every head chain starts after one unloaded round trip and overlaps the tile
    // traffic of the whole grid.
    __builtin_amdgcn_s_waitcnt(0x0070);            // vmcnt(0)
#endif
    // prior rows of this wave's columns, one element per lane (lane = 16q + jj):
    //   vA rows {0, 1, 2, c1},  vB rows {c1+1, c2, c2+1, c2+1}
    const int sj = lane & 15, sq = lane >> 4;
    const int sjc = sj < ncol ? sj : 0;
    const T* colp = Pb + (size_t)(j0 + sjc) * ld;
    const double vA = (double)colp[sq < 3 ? sq : c1];
    const double vB = (double)colp[sq == 0 ? c1 + 1 : (sq == 1 ? c2 : c2 + 1)];
    vec_t pcU[7];                                  // columns U[k] of P0 at this lane's rows
#pragma unroll
    for (int k = 0; k < 7; ++k) pcU[k] = *reinterpret_cast<const vec_t*>(Pb + (size_t)U[k] * ld + rowc);
    const T* Pr = Pb + (size_t)j0 * ld + rowc;
    vec_t p[CW];
#pragma unroll
    // streaming: every tile element is read once and written once (the strips and gain columns, which are shared
    // between workgroups, stay ordinary cached loads)
    for (int jj = 0; jj < CW; ++jj) p[jj] = load_stream(Pr + (size_t)(jj < ncol ? jj : 0) * ld);
    PHASE(12);

    // ---- phase A, four roles in parallel (the transcendental chains need the state and the trace only, not P):
    //   role 1: marker 1 in polar form      role 2: marker 2 in polar form      role 3: z_hat of correction 1
    //   role 0: head of correction 1 (H1, S1^-1) and K1 / M1 at the rows of set2
    auto blk = [&](int a, int bb) { return lane_bcast(v_blk, 7 * a + bb); };   // P0(U[a], U[b]) (head wave only)
    auto u2 = [](int q) { return q < 3 ? q : q + 2; };                          // index of set2[q] in U
    double Hc1r[10], Si1r[4];
    int sing1 = 0, status = 0;
    if (role == 1) {
        double r1, f1;
        obs_polar(o1, raw_a, raw_b, r1, f1);
        if (lane == 0) { sh[S2_OBS] = r1; sh[S2_OBS + 1] = f1; }
    } else if (role == 2) {
        if (blockIdx.y == 0) {                     // marker 2 only enters the state update: strip 0's workgroups
            double r2, f2;
            obs_polar(o2, raw_a, raw_b, r2, f2);
            if (lane == 0) { sh[S2_OBS + 2] = r2; sh[S2_OBS + 3] = f2; }
        }
    } else if (role == 3) {
        double zr, zb;                                                  // :265, at the state before correction 1
        measurement(lane_bcast(v_st, 0), lane_bcast(v_st, 1), lane_bcast(v_st, 2), lane_bcast(v_st, 3),
                    lane_bcast(v_st, 4), zr, zb);
        if (lane == 0) { sh[S2_DZ1] = zr; sh[S2_DZ1 + 1] = zb; }
    } else if (role == 0) {
        const double x = lane_bcast(v_st, 1), y = lane_bcast(v_st, 2);
        const double l1x = lane_bcast(v_st, 3), l1y = lane_bcast(v_st, 4);
        status = status_in;
        double pb[5][5], S[4];
#pragma unroll
        for (int q = 0; q < 5; ++q)
#pragma unroll
            for (int q2 = 0; q2 < 5; ++q2) pb[q][q2] = blk(q2, q);     // pb[q][q2] = P(set[q2], set[q])
        PHASE(1);                                                       // (v_blk / v_st have arrived)
        jacobian_compact(x, y, l1x, l1y, Hc1r);
        innovation_cov_block(pb, Hc1r, v.R, S);
        sing1 = inv2(S, Si1r);
        PHASE(2);
        if (sing1) {                                                    // correction 1 becomes a no-op: K1 = 0
            if (status == 0) status = kStatusSingular;
#pragma unroll
            for (int q = 0; q < 4; ++q) Si1r[q] = 0.0;
        }
        // K1 and M1 at the five rows of set2, one row per lane (lane a < 5 forms row set2[a]), published through LDS
        {
            const int la = lane < 5 ? lane : 0;
            const int ua = la < 3 ? la : la + 2;                        // index of set2[la] in U
            const int ia = la < 3 ? la : c2 + (la - 3);                 // set2[la]
            double pcl[5], K1l[2], M1l[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) pcl[q] = __shfl(v_blk, 7 * ua + q, 64);   // P0(set2[a], set1[q])
            gain_row(pcl, Hc1r, Si1r, ia, set1, K1l, M1l);
            if (lane < 5) {
#pragma unroll
                for (int q = 0; q < 5; ++q) sh[S2_M1S + 5 * lane + q] = M1l[q];
                sh[S2_K1S + 2 * lane] = K1l[0];
                sh[S2_K1S + 2 * lane + 1] = K1l[1];
            }
            // R1S[q][a] = P0(set1[q], set2[a]) sits in lane 7q + u2(a): every such lane stores its own element
            const int aa = lane / 7, bb = lane % 7;
            if (lane < 35 && (bb < 3 || bb > 4)) sh[S2_R1S + 5 * aa + (bb < 3 ? bb : bb - 2)] = v_blk;
            if (lane == 0) sh_i[0] = sing1;
        }
    }
    PHASE(3);
    __syncthreads();
    PHASE(4);

    // ---- phase B, two chains side by side:
    //   head wave: first innovation, position and landmark after correction 1, head of correction 2 (H2, S2^-1)
    //   role 3   : the same state update including the heading, its re-normalisation (:276: sin, cos, atan2 -- a third
    //              of the old single chain) and the second innovation z2 - z_hat2, which only the state rows need
    if (role == 0) {
        const double x = lane_bcast(v_st, 1), y = lane_bcast(v_st, 2);
        const double l2x = lane_bcast(v_st, 5), l2y = lane_bcast(v_st, 6);
        const double dz10 = sing1 ? 0.0 : sh[S2_OBS] - sh[S2_DZ1], dz11 = sing1 ? 0.0 : sh[S2_OBS + 1] - sh[S2_DZ1 + 1];
        double st1[5];
        const double s0v[5] = { 0.0, x, y, l2x, l2y };
#pragma unroll
        for (int a = 1; a < 5; ++a) {
            double acc = 0.0;
            acc = fma(sh[S2_K1S + 2 * a], dz10, acc);
            acc = fma(sh[S2_K1S + 2 * a + 1], dz11, acc);
            st1[a] = s0v[a] + acc;
        }
        // P1(set2, set2), one entry per lane (lane = 5q + q2: row set2[q2], column set2[q]), exactly as the sweep forms it
        double ent = 0.0;
        {
            const int e = lane < 25 ? lane : 0;
            const int q = e / 5, q2 = e % 5;
            double mrow[5], r1v[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                mrow[k] = sh[S2_M1S + 5 * q2 + k];                      // written by this wave before the barrier
                r1v[k] = __shfl(v_blk, 7 * k + u2(q), 64);              // P0(set1[k], set2[q])
            }
            const int i = set2[q2];
            const double bef = (i > 2 && i < c1) ? 1.0 : 0.0, aft = (i > c1 + 1) ? 1.0 : 0.0;
            ent = p1_entry<T>(mrow, r1v, __shfl(v_blk, 7 * u2(q2) + u2(q), 64), bef, aft);
        }
        double Hc2[10], Si2[4], pb[5][5], S[4];
#pragma unroll
        for (int q = 0; q < 5; ++q)
#pragma unroll
            for (int q2 = 0; q2 < 5; ++q2) pb[q][q2] = lane_bcast(ent, 5 * q + q2);
        jacobian_compact(st1[1], st1[2], st1[3], st1[4], Hc2);         // :268 at the corrected state
        innovation_cov_block(pb, Hc2, v.R, S);
        int sing2 = inv2(S, Si2);
        if (sing2) {
            if (status == 0) status = kStatusSingular;
#pragma unroll
            for (int q = 0; q < 4; ++q) Si2[q] = 0.0;
        }
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 10; ++q) { sh[S2_HC1 + q] = Hc1r[q]; sh[S2_HC2 + q] = Hc2[q]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) { sh[S2_SI1 + q] = Si1r[q]; sh[S2_SI2 + q] = Si2[q]; }
            sh[S2_NU1] = dz10; sh[S2_NU1 + 1] = dz11;
            sh_i[1] = sing2; sh_i[2] = status;
        }
    } else if ((role == 3 || role == 2) && blockIdx.y == 0) {
        // consumed by the state rows only (column strip 0's workgroups).  Two waves share the chain: role 3 re-normalises
        // the heading (sin, cos, atan2), role 2 turns the corrected landmark offset into polar form (sqrt; atan2, sin,
        // cos, atan2); the last step of computeTheoreticalMeasurement, normalize(bearing - heading), needs both and is
        // done by the state-owning wave after the barrier.
        const int sg1 = sh_i[0];
        const double dz10 = sg1 ? 0.0 : sh[S2_OBS] - sh[S2_DZ1], dz11 = sg1 ? 0.0 : sh[S2_OBS + 1] - sh[S2_DZ1 + 1];
        const double s0v[5] = { lane_bcast(v_st, 0), lane_bcast(v_st, 1), lane_bcast(v_st, 2), lane_bcast(v_st, 5),
                                lane_bcast(v_st, 6) };
        double st1[5];
#pragma unroll
        for (int a = 0; a < 5; ++a) {
            double acc = 0.0;
            acc = fma(sh[S2_K1S + 2 * a], dz10, acc);
            acc = fma(sh[S2_K1S + 2 * a + 1], dz11, acc);
            st1[a] = s0v[a] + acc;
        }
        if (role == 3) {
            if (!sg1) st1[0] = normalize_angle(st1[0]);                // update() re-normalises the heading (:276)
            if (lane == 0) sh[S2_ST1] = st1[0];                        // the heading row's value after correction 1
        } else {
            double zr, zb;                                             // computeTheoreticalMeasurement (:150-160) up to
            cartesian2polar(st1[3] - st1[1], st1[4] - st1[2], zr, zb); // the polar form of the offset
            if (lane == 0) { sh[S2_ZH2] = zr; sh[S2_ZH2 + 1] = zb; }
        }
    }
    PHASE(5);
    __syncthreads();
    PHASE(6);
    if (!active) return;
    sing1 = sh_i[0];
    const int sing2 = sh_i[1];
    // H and S^-1 of both corrections stay in LDS (wave-uniform broadcast reads): holding 28 doubles per lane in VGPRs
    // pushed the kernel past 256 registers, i.e. down to one wave per SIMD
    const double* Hc1 = sh + S2_HC1; const double* Si1 = sh + S2_SI1;
    const double* Hc2 = sh + S2_HC2; const double* Si2 = sh + S2_SI2;
    const double dz10 = sh[S2_NU1], dz11 = sh[S2_NU1 + 1];

    // the innovations only feed the state: column strip 0's waves
    const bool owns_state = (strip == 0);
    double dz20 = 0.0, dz21 = 0.0;
    if (owns_state && !sing2) {
        const double zb = normalize_angle(sh[S2_ZH2 + 1] - sh[S2_ST1]);     // the last step of z_hat2 (:159)
        dz20 = sh[S2_OBS + 2] - sh[S2_ZH2];
        dz21 = sh[S2_OBS + 3] - zb;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        int* co = v.c_out + b * C_WORDS;
        co[C_SEEN] = seen_in; co[C_SEEN_CACHED] = cached_in; co[C_BRK] = brk_in; co[C_STATUS] = sh_i[2];
        if (v.id_log && o1.log_slot >= 0) v.id_log[(size_t)b * v.log_stride + o1.log_slot] = id1;
        if (v.id_log && o2.log_slot >= 0) v.id_log[(size_t)b * v.log_stride + o2.log_slot] = id2;
    }

    // ---- this lane's rows: M1, then the columns set2 of P1 at these rows, then M2
    double m1[VEC][5], m2[VEC][5], bef1[VEC], aft1[VEC], bef2[VEC], aft2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const int i = row0 + e;
        double pc[5], K1[2], K2[2];
#pragma unroll
        for (int q = 0; q < 5; ++q) pc[q] = (double)pcU[q].v[e];                     // P0(i, set1[q])
        gain_row(pc, Hc1, Si1, i, set1, K1, m1[e]);
        bef1[e] = ((i > 2) && (i < c1)) ? 1.0 : 0.0;
        aft1[e] = (i > c1 + 1) ? 1.0 : 0.0;
        bef2[e] = ((i > 2) && (i < c2)) ? 1.0 : 0.0;
        aft2[e] = (i > c2 + 1) ? 1.0 : 0.0;
#pragma unroll
        for (int q = 0; q < 5; ++q) {                                                // P1(i, set2[q])
            double r1v[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) r1v[k] = sh[S2_R1S + 5 * k + q];
            pc[q] = p1_entry<T>(m1[e], r1v, (double)pcU[q < 3 ? q : q + 2].v[e], bef1[e], aft1[e]);
        }
        gain_row(pc, Hc2, Si2, i, set2, K2, m2[e]);
        if (owns_state && rows_ok) {
            // state after both corrections: s + K1 nu1, heading re-normalised, then + K2 nu2, re-normalised (:275-276)
            double acc = 0.0;
            acc = fma(K1[0], dz10, acc);
            acc = fma(K1[1], dz11, acc);
            double sv = (kEarlyRows ? s_rows[e] : s[i]) + acc;
            if (i == 0) sv = sh[S2_ST1];           // the heading: role 3 formed exactly this sum and re-normalised it (:276)
            acc = 0.0;
            acc = fma(K2[0], dz20, acc);
            acc = fma(K2[1], dz21, acc);
            sv = sv + acc;
            if (i == 0 && !sing2) sv = normalize_angle(sv);
            so[i] = sv;
        }
    }

    PHASE(7);
    // ---- this wave's columns: the rows set2 of P1, lane-distributed like vA / vB (wA rows {0,1,2,c2}, wB row c2+1)
    double wA, wB;
    {
        double r1v[5];
        r1v[0] = __shfl(vA, sj, 64); r1v[1] = __shfl(vA, 16 + sj, 64); r1v[2] = __shfl(vA, 32 + sj, 64);
        r1v[3] = __shfl(vA, 48 + sj, 64); r1v[4] = __shfl(vB, sj, 64);
        const double p0c2 = __shfl(vB, 16 + sj, 64), p0c21 = __shfl(vB, 32 + sj, 64);   // P0(c2, j), P0(c2+1, j)
        double mrow[5];
        {
            const int a = sq;                                        // set2[a], a = 0..3
#pragma unroll
            for (int q = 0; q < 5; ++q) mrow[q] = sh[S2_M1S + 5 * a + q];
            const int i = set2[a];
            const double bef = (i > 2 && i < c1) ? 1.0 : 0.0, aft = (i > c1 + 1) ? 1.0 : 0.0;
            wA = p1_entry<T>(mrow, r1v, a < 3 ? vA : p0c2, bef, aft);
        }
        {
#pragma unroll
            for (int q = 0; q < 5; ++q) mrow[q] = sh[S2_M1S + 5 * 4 + q];
            const int i = set2[4];
            const double bef = (i > 2 && i < c1) ? 1.0 : 0.0, aft = (i > c1 + 1) ? 1.0 : 0.0;
            wB = p1_entry<T>(mrow, r1v, p0c21, bef, aft);
        }
    }

    // ---- the sweep: both corrections on the tile in registers.  The ten prior-row values of a column (R1, R2) are
    // wave-uniform; they go through a per-wave LDS strip and come back as broadcast ds_read_b128 -- 5 LDS reads per
    // column instead of 20 v_readlane, which were a sixth of this kernel's VALU instructions.
    __shared__ double sR[WAVES][CW][10];
    {
        double(*R)[10] = sR[wave];
        R[sj][sq] = vA;                               // r1[0..3] = rows 0, 1, 2, c1
        R[sj][5 + sq] = wA;                           // r2[0..3] = rows 0, 1, 2, c2 of P1
        if (sq == 0) { R[sj][4] = vB; R[sj][9] = wB; }    // r1[4] = row c1+1, r2[4] = row c2+1 of P1
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): the strip is this wave's own, no barrier needed
    T* Pw = Pout + (size_t)b * v.p_stride + (size_t)j0 * ld + row0;
    PHASE(8);
    auto sweep_column = [&](int jj) {
        double r1v[5], r2v[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) { r1v[q] = sR[wave][jj][q]; r2v[q] = sR[wave][jj][5 + q]; }
        vec_t out;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            // k_update stores P1 in T; with fp32 storage the second correction must start from that rounded value
            const double p1 = p1_entry<T>(m1[e], r1v, (double)p[jj].v[e], bef1[e], aft1[e]);
            out.v[e] = (T)sweep_entry(m2[e], r2v, p1, bef2[e], aft2[e]);
        }
        if (jj < ncol && rows_ok) store_stream(Pw + (size_t)jj * ld, out);
    };
#pragma unroll
    for (int jj = 0; jj < CW / 2; ++jj) sweep_column(jj);
    if (ncol > CW / 2) {                               // wave-uniform: strip 0 has only the first half
#pragma unroll
        for (int jj = CW / 2; jj < CW; ++jj) sweep_column(jj);
    }
#ifdef NUSLAM_PHASE_CLOCK
    PHASE(9);
    __builtin_amdgcn_s_waitcnt(0x0070);               // vmcnt(0): the tile's stores have been acknowledged
    PHASE(10);
    if (threadIdx.x == 0 && blockIdx.z == 0) g_wg[blockIdx.y * gridDim.x + blockIdx.x][1] = (long long)wall_clock64();
#endif
}

} // namespace nuslam
