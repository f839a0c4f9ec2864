// ekf_da.h -- a tick with UNKNOWN data association (slam.cpp:279-318 with associateLandmark, slam_library.cpp:188-253,
// in front of every correction) as m small launches and ONE pass over the covariance.
//
// The per-correction path pays a full pass over P for every marker (k_associate + k_update, 2 len^2 w bytes each) although
// associateLandmark itself reads only O(len) numbers of P: for every candidate k the 5 x 5 block P(set_k, set_k),
// set_k = {0, 1, 2, c_k, c_k + 1}, i.e.
//
//     TR[q][j] = P(q, j)   rows 0..2            TC[q][i] = P(i, q)   columns 0..2
//     TD[e][k] = P(c_k + (e & 1), c_k + (e >> 1))   the 2 x 2 diagonal blocks
//
// and those obey the sweep formula of ekf_update.h restricted to themselves:  P_s(i,j) = sum_k M_s(i,k) P_{s-1}(k,j)
// needs, besides the entry, the gain row K_s(i,:) and the five prior entries R_s(:, j) = P_{s-1}(set_s, j) -- the strips
// the tick pipeline (ekf_tick.h) forms anyway.  With known ids all strips of a tick come out of one launch (k_tick_panels);
// here the landmark of correction s is known only after associating against the result of correction s-1, so the strips
// are formed one correction per launch:
//
//   k_da_begin     TR / TC / TD / state / control words out of the covariance after predict; candidates of marker 0.
//   k_da_step(s)   every workgroup: reduce the candidates' keys -> id (decode_association + the caller's chain); H, S,
//                  S^-1, z_hat (redundantly, from tracked entries);  rows c, c+1 and columns c, c+1 of P_{s-1} = the tick's
//                  starting covariance (one gather) replayed through corrections 0..s-1 from their stored strips;
//                  R_s (5 x len), K_s (len x 2), the state;  TR / TC / TD after the correction;  the candidates of marker
//                  s + 1 against exactly that.  One thread per (state index, role), 60 indices per workgroup.
//   k_tick_apply   the pass over P with all strips of the round (ekf_tick.h), unchanged.
//
// Every entry goes through the operations k_associate / k_update perform on it, in the same order (tests/test_gpu_da.py
// compares bit for bit), so the decisions -- match, new landmark, gray zone -- are the per-correction path's.
#pragma once

namespace nuslam {

constexpr int kDaSlots = 64;     // index slots of a workgroup: slots 0..2 = the pose indices (carried by EVERY workgroup,
constexpr int kDaOwn = 60;       // stored by workgroup 0), slots 3..62 = 60 own indices = 30 landmarks, slot 63 idle
constexpr int kDaLm = kDaOwn / 2;
constexpr int kDaSlotStride = 16; // 8-byte words between two workgroups' key slots: a 128-byte line each

struct DaBuf {
    double* TR[2];   // [B][3][ld]
    double* TC[2];   // [B][3][ld]
    double* TD[2];   // [B][4][n]
    double* DS[2];   // [B][ld]       state
    int* DC[2];      // [B][C_WORDS]  control words
    double* Z;       // [B][2][kTickJ] the round's markers in polar form (slam.cpp:286)
    int* keyp;       // [B][kTickJ][nwg] association key of marker s as seen by workgroup w (min over its candidates)
    int nwg;
    // resident round only:
    double* AP[2];          // [B][n][16]  what associateLandmark formed for a MATCHING candidate: H (10), psi^-1 (4), z_hat (2)
    long long* keyt;        // [B][kTickJ][nwg][kDaSlotStride]  (tag << 32) | key: the key slot doubles as the workgroup's arrival flag
    long long* fwd;         // [B][8]  served rounds: the command workgroup 0 took from the host's mailbox, for the other workgroups
};

// associateLandmark's test of one candidate (slam_library.cpp:209-246) from pb[q][q2] = P(set[q2], set[q]):
// 0 = match, 1 = gray zone, 2 = psi singular, -1 = neither
__device__ inline int assoc_code(const double pb[5][5], const double R[4], double th, double x, double y, double lx,
                                 double ly, double r, double phi)
{
    const double min_threshold = 0.01;   // :193
    const double max_threshold = 60;     // :194
    double Hc[10], psi[4], psi_inv[4], zr, zb;
    jacobian_compact(x, y, lx, ly, Hc);                    // :212
    innovation_cov_block(pb, Hc, R, psi);                  // :215
    measurement(th, x, y, lx, ly, zr, zb);                 // :218
    const double dz0 = r - zr, dz1 = phi - zb;             // :229 (bearing difference not wrapped)
    if (inv2(psi, psi_inv)) return 2;                      // psi_k.i() would throw
    double w0 = 0.0, w1 = 0.0, d = 0.0;                    // (dz^T psi^-1) dz, :231
    w0 = fma(dz0, psi_inv[0], w0); w0 = fma(dz1, psi_inv[1], w0);
    w1 = fma(dz0, psi_inv[2], w1); w1 = fma(dz1, psi_inv[3], w1);
    d = fma(w0, dz0, d); d = fma(w1, dz1, d);
    if (d < min_threshold) return 0;                                   // :238
    if ((d > min_threshold) && (d < max_threshold)) return 1;          // :243
    return -1;
}

__device__ inline int da_index(int wg, int l) { return l < 3 ? l : 3 + wg * kDaOwn + (l - 3); }

// ------------------------------------------------------------------------------------------------ begin
template <typename T>
__global__ __launch_bounds__(256) void k_da_begin(View v, TickObs o, const T* __restrict__ P, DaBuf d)
{
    const int b = blockIdx.y, wg = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ld = v.ld, L = v.L, n = v.n;
    const T* Pb = P + (size_t)b * v.p_stride;
    const double* s = v.s_in + (size_t)b * ld;
    const int t = da_index(wg, lane);
    const bool own = lane < 63 && (lane >= 3 || wg == 0);
    if (wave == 0) {
        if (own && t < L) {
#pragma unroll
            for (int q = 0; q < 3; ++q) d.TR[0][((size_t)b * 3 + q) * ld + t] = (double)Pb[(size_t)t * ld + q];
        }
    } else if (wave == 1) {
        if (own && t < ld) {
#pragma unroll
            for (int q = 0; q < 3; ++q) d.TC[0][((size_t)b * 3 + q) * ld + t] = (double)Pb[(size_t)q * ld + t];
            d.DS[0][(size_t)b * ld + t] = s[t];
        }
    } else if (wave == 2) {
        const int k = wg * kDaLm + lane;                   // 0-based landmark
        if (lane < kDaLm && k < n) {
            const int c = 3 + 2 * k;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                d.TD[0][((size_t)b * 4 + e) * n + k] = (double)Pb[(size_t)(c + (e >> 1)) * ld + c + (e & 1)];
        }
    } else {
        if (wg == 0 && lane < C_WORDS) d.DC[0][b * C_WORDS + lane] = v.c_in[b * C_WORDS + lane];
        if (wg == 0 && lane < kTickJ) {
            double a = 0.0, bb = 0.0;
            if (lane < o.J) {
                a = o.a ? o.a[b * o.stride + o.off + lane] : o.a0[lane];
                bb = o.b ? o.b[b * o.stride + o.off + lane] : o.b0[lane];
            }
            double r, phi;
            if (o.cartesian) cartesian2polar(a, bb, r, phi);
            else { r = a; phi = bb; }
            d.Z[((size_t)b * 2 + 0) * kTickJ + lane] = r;
            d.Z[((size_t)b * 2 + 1) * kTickJ + lane] = phi;
        }
        // the candidates of marker 0, exactly k_associate
        const int* ci = v.c_in + b * C_WORDS;
        const int seen = ci[C_SEEN];
        int key = kNoKey;
        const int k1 = wg * kDaLm + lane + 1;
        if (!(ci[C_BRK] || seen == 0 || seen >= n) && lane < kDaLm && k1 <= seen) {
            const double a = o.a ? o.a[b * o.stride + o.off] : o.a0[0];
            const double bb = o.b ? o.b[b * o.stride + o.off] : o.b0[0];
            double r, phi;
            if (o.cartesian) cartesian2polar(a, bb, r, phi);
            else { r = a; phi = bb; }
            const int c = 3 + 2 * (k1 - 1);
            const int set[5] = { 0, 1, 2, c, c + 1 };
            double pb[5][5];
#pragma unroll
            for (int q = 0; q < 5; ++q)
#pragma unroll
                for (int q2 = 0; q2 < 5; ++q2) pb[q][q2] = (double)Pb[(size_t)set[q] * ld + set[q2]];
            const int code = assoc_code(pb, v.R, s[0], s[1], s[2], s[c], s[c + 1], r, phi);
            if (code >= 0) key = k1 * 4 + code;
        }
        key = wave_min(key);
        if (lane == 0) d.keyp[((size_t)b * kTickJ + 0) * d.nwg + wg] = key;
    }
}

// ------------------------------------------------------------------------------------------------ one correction
// wave 0: ROW role, lane = slot: column t of the prior rows (R_s), of TR
// wave 1: COLUMN role: row t of the gain (K_s), of TC, state entry t
// wave 2: the head: H, S, S^-1, the gain rows of the pose (what k_update's wave 0 does)
// wave 3: the replay coefficients of rows / columns c, c+1 (uniform over the workgroup), z_hat and the innovation
template <typename T>
__global__ __launch_bounds__(256) void k_da_step(View v, TickObs o, int st, int last, int total_landmarks,
                                                 const T* __restrict__ P, DaBuf d, TickStep* __restrict__ plan,
                                                 double* __restrict__ Kbuf, double* __restrict__ Rbuf, double* __restrict__ Vbuf)
{
    // Vbuf (may be null): V_s = H_s R_s beside R_s, the second factor of the rank-2m pass (ekf_rank.h) -- hp_entry's sum
    const int b = blockIdx.y, wg = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ld = v.ld, L = v.L, n = v.n;
    const int par = st & 1;
    const double* TRc = d.TR[par] + (size_t)b * 3 * ld;
    double* TRn = d.TR[par ^ 1] + (size_t)b * 3 * ld;
    const double* TCc = d.TC[par] + (size_t)b * 3 * ld;
    double* TCn = d.TC[par ^ 1] + (size_t)b * 3 * ld;
    const double* TDc = d.TD[par] + (size_t)b * 4 * n;
    double* TDn = d.TD[par ^ 1] + (size_t)b * 4 * n;
    const double* sc = d.DS[par] + (size_t)b * ld;
    double* sn = d.DS[par ^ 1] + (size_t)b * ld;
    const int* cc = d.DC[par] + b * C_WORDS;
    int* cn = d.DC[par ^ 1] + b * C_WORDS;
    const T* Pb = P + (size_t)b * v.p_stride;
    TickStep* pl = plan + (size_t)b * kTickJ;
    double* Kb = Kbuf + (size_t)b * kTickJ * 2 * ld;
    double* Rb = Rbuf + (size_t)b * kTickJ * 5 * ld;
    const int t = da_index(wg, lane);
    const bool own = lane < 63 && (lane >= 3 || wg == 0);

    __shared__ double hd[20];                 // Hc[10], Sinv[4], lx, ly, dz0, dz1
    __shared__ int hi[2];                     // no correction (skip / singular), status after
    __shared__ double Mpose[3][6];            // M_s(q, set_s), q = 0..2
    __shared__ double hist[kTickJ][12];       // correction t' < s: Hc[10], c, skip
    __shared__ double mcL[kTickJ][2][8];      // M_t'(c + e, set_t'), before, after
    __shared__ double rcL[kTickJ][2][6];      // R_t'(:, c + e)
    __shared__ double rcolL[kDaSlots][5];     // R_s(:, t) of the workgroup's columns
    __shared__ double mrowL[kDaSlots][8];     // M_s(t, set_s), before, after of its rows
    __shared__ double nTR[kDaSlots][3], nTC[kDaSlots][3], nS[kDaSlots], nTD[kDaLm][4];

    // ---- loads that depend on kernel arguments only: tracked entries and this thread's strips of corrections 0..s-1
    const int tr_ = t < L ? t : 0, tc_ = t < ld ? t : 0;
    double e3[3] = { 0.0, 0.0, 0.0 }, sv = 0.0;
    double strip[kTickJ][5];                                            // wave 0: R_t'(:, t); wave 1: [t'][0..1] = K_t'(t, :)
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) e3[q] = TRc[(size_t)q * ld + tr_];
#pragma unroll
        for (int tp = 0; tp < kTickJ; ++tp) {
            const int tq = tp < st ? tp : 0;
#pragma unroll
            for (int q = 0; q < 5; ++q) strip[tp][q] = Rb[(size_t)(tq * 5 + q) * ld + tr_];
        }
    } else if (wave == 1) {
#pragma unroll
        for (int q = 0; q < 3; ++q) e3[q] = TCc[(size_t)q * ld + tc_];
        sv = sc[tc_];
#pragma unroll
        for (int tp = 0; tp < kTickJ; ++tp) {
            const int tq = tp < st ? tp : 0;
            strip[tp][0] = Kb[(size_t)(tq * 2 + 0) * ld + tc_];
            strip[tp][1] = Kb[(size_t)(tq * 2 + 1) * ld + tc_];
        }
    }

    // ---- the decision, by every wave: associateLandmark's verdict from the candidates' keys, then slam.cpp:295-316
    int key = kNoKey;
    for (int i = lane; i < d.nwg; i += 64) {
        const int kk = d.keyp[((size_t)b * kTickJ + st) * d.nwg + i];
        key = kk < key ? kk : key;
    }
    key = wave_min(key);
    const int seen = cc[C_SEEN], cached = cc[C_SEEN_CACHED], brk = cc[C_BRK], status0 = cc[C_STATUS];
    int id_raw, seen_now, status_now;
    {
        const Assoc a = decode_association(n, seen, brk, status0, key);
        id_raw = a.id; seen_now = a.new_seen; status_now = a.new_status;
        // a marker slot without a marker (presence word < 0): associateLandmark is not called for it (slam.cpp:279)
        if (o.ids != nullptr && o.ids[b * o.stride + o.off + st] < 0) { id_raw = -1; seen_now = seen; status_now = status0; }
    }
    const Decision dd = resolve(n, id_raw, seen_now, cached, brk, status_now, MODE_DA, total_landmarks);
    const int c = dd.c;
    const int setv[5] = { 0, 1, 2, c, c + 1 };

    // ---- loads that depend on the landmark
    double g0 = 0.0, g1 = 0.0;
    if (wave == 0) {                                                    // P0(c, t), P0(c+1, t): one line per column
        g0 = (double)Pb[(size_t)tr_ * ld + c];
        g1 = (double)Pb[(size_t)tr_ * ld + c + 1];
    } else if (wave == 1) {                                             // P0(t, c), P0(t, c+1): contiguous
        g0 = (double)Pb[(size_t)c * ld + tc_];
        g1 = (double)Pb[(size_t)(c + 1) * ld + tc_];
    } else if (wave == 2) {
        // lane 5 q + q2 < 25: P_{s-1}(set[q2], set[q]); lanes 32..36: th, x, y, lx, ly; lanes 40, 41: the marker
        const int e = lane < 25 ? lane : 0;
        const int q = e / 5, q2 = e % 5;
        const double* src;
        if (q2 < 3) src = TRc + (size_t)q2 * ld + setv[q];
        else if (q < 3) src = TCc + (size_t)q * ld + setv[q2];
        else src = TDc + (size_t)((q2 - 3) + 2 * (q - 3)) * n + (c - 3) / 2;
        const int sl = lane - 32;
        if (sl >= 0 && sl < 5) src = sc + setv[sl];
        if (sl == 8) src = d.Z + ((size_t)b * 2 + 0) * kTickJ + st;
        if (sl == 9) src = d.Z + ((size_t)b * 2 + 1) * kTickJ + st;
        const double val = *src;
        const double th = lane_bcast(val, 32), x = lane_bcast(val, 33), y = lane_bcast(val, 34);
        double lx = lane_bcast(val, 35), ly = lane_bcast(val, 36);
        const double r = lane_bcast(val, 40), phi = lane_bcast(val, 41);
        bool skip0 = dd.skip;
        int stt = dd.new_status;
        double Hc[10], Si[4];
#pragma unroll
        for (int k = 0; k < 10; ++k) Hc[k] = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) Si[k] = 0.0;
        if (!skip0) {
            if (dd.init) {                                              // initializeLandmark, slam_library.cpp:255-261
                lx = x + r * cos(phi + th);
                ly = y + r * sin(phi + th);
            }
            double pb[5][5], S[4];
#pragma unroll
            for (int a = 0; a < 5; ++a)
#pragma unroll
                for (int a2 = 0; a2 < 5; ++a2) pb[a][a2] = lane_bcast(val, 5 * a + a2);
            jacobian_compact(x, y, lx, ly, Hc);                         // :268
            innovation_cov_block(pb, Hc, v.R, S);                       // :270
            if (inv2(S, Si)) { skip0 = true; if (stt == 0) stt = kStatusSingular; }
            // the gain rows of the pose: pc[a2] = P(lane, set[a2]) (shuffled by all lanes: a lane masked out returns nothing)
            double pc[5];
#pragma unroll
            for (int a2 = 0; a2 < 5; ++a2) pc[a2] = __shfl(val, 5 * a2 + (lane < 3 ? lane : 0), 64);
            if (!skip0 && lane < 3) {
                double K[2], m[5];
                gain_row(pc, Hc, Si, lane, setv, K, m);
#pragma unroll
                for (int a2 = 0; a2 < 5; ++a2) Mpose[lane][a2] = m[a2];
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 10; ++k) hd[k] = Hc[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) hd[10 + k] = Si[k];
            hd[14] = lx; hd[15] = ly;
            hi[0] = skip0 ? 1 : 0; hi[1] = stt;
        }
    } else {
        // lanes (t', e): the coefficients of rows / columns c + e through correction t'
        const int tp = lane >> 1, e = lane & 1;
        if (lane < 2 * kTickJ) {
            const bool livep = tp < st;
            const TickStep* ps = pl + (livep ? tp : 0);
            const int skp = livep ? ps->skip : 1, cp = ps->c;
            double Hp[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) Hp[k] = ps->Hc[k];
            const int i = c + e;
            const double K0 = Kb[(size_t)((livep ? tp : 0) * 2 + 0) * ld + i], K1 = Kb[(size_t)((livep ? tp : 0) * 2 + 1) * ld + i];
            double rr[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) rr[q] = Rb[(size_t)((livep ? tp : 0) * 5 + q) * ld + i];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                double kh = 0.0;                                        // gain_row's M(i, set[q]) = delta - (K H)(i, set[q])
                kh = fma(K0, Hp[0 + 2 * q], kh);
                kh = fma(K1, Hp[1 + 2 * q], kh);
                const int sidx = q < 3 ? q : cp + (q - 3);
                mcL[tp][e][q] = (i == sidx ? 1.0 : 0.0) - kh;
                rcL[tp][e][q] = rr[q];
            }
            mcL[tp][e][5] = (i > 2 && i < cp) ? 1.0 : 0.0;
            mcL[tp][e][6] = (i > cp + 1) ? 1.0 : 0.0;
            if (e == 0) {
#pragma unroll
                for (int k = 0; k < 10; ++k) hist[tp][k] = Hp[k];
                hist[tp][10] = (double)cp;
                hist[tp][11] = (double)skp;
            }
        }
        // z_hat at the (possibly just initialised) landmark and the innovation, :265, :272
        if (!dd.skip) {
            const double th = sc[0], x = sc[1], y = sc[2];
            const double r = d.Z[((size_t)b * 2 + 0) * kTickJ + st], phi = d.Z[((size_t)b * 2 + 1) * kTickJ + st];
            double lx, ly;
            if (dd.init) { lx = x + r * cos(phi + th); ly = y + r * sin(phi + th); }
            else { lx = sc[c]; ly = sc[c + 1]; }
            double zr, zb;
            measurement(th, x, y, lx, ly, zr, zb);
            if (lane == 0) { hd[16] = r - zr; hd[17] = phi - zb; }
        }
    }
    __syncthreads();

    const bool nocorr = hi[0] != 0;
    const int new_status = hi[1];
    const double lx = hd[14], ly = hd[15];
    double Hc[10], Si[4];
#pragma unroll
    for (int k = 0; k < 10; ++k) Hc[k] = hd[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) Si[k] = hd[10 + k];
    double n3[3] = { e3[0], e3[1], e3[2] };                            // TR / TC of this slot after the correction
    double m5[5] = { 0.0, 0.0, 0.0, 0.0, 0.0 }, bef = 0.0, aft = 0.0;  // wave 1: M_s(t, set_s)

    if (wave == 0) {
        if (!nocorr) {
            // rows c, c+1 of P_{s-1} at column t: the gathered entries replayed through corrections 0..s-1
#pragma unroll
            for (int tp = 0; tp < kTickJ; ++tp) {
                if (tp < st && hist[tp][11] == 0.0) {
                    g0 = p1_entry<T>(mcL[tp][0], strip[tp], g0, mcL[tp][0][5], mcL[tp][0][6]);
                    g1 = p1_entry<T>(mcL[tp][1], strip[tp], g1, mcL[tp][1][5], mcL[tp][1][6]);
                }
            }
            const double rs[5] = { e3[0], e3[1], e3[2], g0, g1 };
            if (own && t < L) {
#pragma unroll
                for (int q = 0; q < 5; ++q) Rb[(size_t)(st * 5 + q) * ld + t] = rs[q];
                if (Vbuf) {
                    double* Vb = Vbuf + (size_t)b * kTickJ * 2 * ld;
                    Vb[(size_t)(st * 2 + 0) * ld + t] = hp_entry(Hc, rs, 0);
                    Vb[(size_t)(st * 2 + 1) * ld + t] = hp_entry(Hc, rs, 1);
                }
            }
#pragma unroll
            for (int q = 0; q < 5; ++q) rcolL[lane][q] = rs[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) n3[q] = p1_entry<T>(Mpose[q], rs, e3[q], 0.0, 0.0);
        }
        if (own && t < L) {
#pragma unroll
            for (int q = 0; q < 3; ++q) TRn[(size_t)q * ld + t] = n3[q];
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) nTR[lane][q] = n3[q];
    } else if (wave == 1) {
        double snew = sv;
        if (dd.init && !dd.skip) {                                      // initializeLandmark ran (also when update() then threw)
            if (t == c) snew = lx;
            if (t == c + 1) snew = ly;
        }
        if (!nocorr) {
#pragma unroll
            for (int tp = 0; tp < kTickJ; ++tp) {
                if (tp < st && hist[tp][11] == 0.0) {
                    const int cp = (int)hist[tp][10];
                    double mt[5];
#pragma unroll
                    for (int q = 0; q < 5; ++q) {
                        double kh = 0.0;
                        kh = fma(strip[tp][0], hist[tp][0 + 2 * q], kh);
                        kh = fma(strip[tp][1], hist[tp][1 + 2 * q], kh);
                        const int sidx = q < 3 ? q : cp + (q - 3);
                        mt[q] = (t == sidx ? 1.0 : 0.0) - kh;
                    }
                    const double bp = (t > 2 && t < cp) ? 1.0 : 0.0, ap = (t > cp + 1) ? 1.0 : 0.0;
                    g0 = p1_entry<T>(mt, rcL[tp][0], g0, bp, ap);
                    g1 = p1_entry<T>(mt, rcL[tp][1], g1, bp, ap);
                }
            }
            const double pc[5] = { e3[0], e3[1], e3[2], g0, g1 };
            double K[2];
            gain_row(pc, Hc, Si, t, setv, K, m5);
            bef = (t > 2 && t < c) ? 1.0 : 0.0;
            aft = (t > c + 1) ? 1.0 : 0.0;
            if (own && t < ld) {
                Kb[(size_t)(st * 2 + 0) * ld + t] = K[0];
                Kb[(size_t)(st * 2 + 1) * ld + t] = K[1];
            }
            double acc = 0.0;                                           // state += K (z - z_hat), heading re-normalised (:275-276)
            acc = fma(K[0], hd[16], acc);
            acc = fma(K[1], hd[17], acc);
            snew = snew + acc;
            if (t == 0) snew = normalize_angle(snew);
#pragma unroll
            for (int q = 0; q < 5; ++q) mrowL[lane][q] = m5[q];
            mrowL[lane][5] = bef; mrowL[lane][6] = aft;
        }
        if (own && t < ld) {
            sn[t] = snew;
            if (last) v.s_out[(size_t)b * ld + t] = snew;
        }
        nS[lane] = snew;
    } else if (wave == 3 && wg == 0 && lane == 0) {
        // the correction's record for the pass over P, the control words, the id log
        TickStep* ps = pl + st;
        ps->skip = nocorr ? 1 : 0; ps->init = (dd.init && !dd.skip) ? 1 : 0; ps->c = c; ps->id = dd.id;
#pragma unroll
        for (int k = 0; k < 10; ++k) ps->Hc[k] = Hc[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) ps->Sinv[k] = Si[k];
        ps->dz[0] = hd[16]; ps->dz[1] = hd[17]; ps->lxy[0] = lx; ps->lxy[1] = ly;
        cn[C_SEEN] = dd.new_seen; cn[C_SEEN_CACHED] = cached; cn[C_BRK] = dd.new_brk; cn[C_STATUS] = new_status;
        if (last) {
            int* co = v.c_out + b * C_WORDS;
            co[C_SEEN] = dd.new_seen; co[C_SEEN_CACHED] = cached; co[C_BRK] = dd.new_brk; co[C_STATUS] = new_status;
        }
        if (v.id_log && o.log_slot0 >= 0) v.id_log[(size_t)b * v.log_stride + o.log_slot0 + st] = dd.id;
    }
    __syncthreads();

    // ---- columns 0..2 at this row (needs R_s(:, q) = the row role's strips of slots 0..2), the diagonal blocks
    if (wave == 1) {
        if (!nocorr) {
#pragma unroll
            for (int q = 0; q < 3; ++q) n3[q] = p1_entry<T>(m5, rcolL[q], e3[q], bef, aft);
        }
        if (own && t < ld) {
#pragma unroll
            for (int q = 0; q < 3; ++q) TCn[(size_t)q * ld + t] = n3[q];
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) nTC[lane][q] = n3[q];
    } else if (wave == 0 || wave == 2) {
        // entry e of landmark lm: P(c_lm + (e & 1), c_lm + (e >> 1))
        const int idx = (wave == 0 ? 0 : 64) + lane;
        const int lm = idx >> 2, e = idx & 3;
        const int k = wg * kDaLm + lm;
        if (lm < kDaLm) {
            double val = 0.0;
            if (k < n) {
                val = TDc[(size_t)e * n + k];
                const int si = 3 + 2 * lm + (e & 1), sj = 3 + 2 * lm + (e >> 1);
                if (!nocorr) val = p1_entry<T>(mrowL[si], rcolL[sj], val, mrowL[si][5], mrowL[si][6]);
                TDn[(size_t)e * n + k] = val;
            }
            nTD[lm][e] = val;
        }
    }
    if (st + 1 >= o.J) return;                                          // the round's last correction: nothing to associate
    __syncthreads();

    // ---- the candidates of marker s+1 against the state and covariance this correction leaves (k_associate)
    if (wave == 0) {
        const int seen1 = dd.new_seen;
        int key1 = kNoKey;
        const int k1 = wg * kDaLm + lane + 1;
        if (!(dd.new_brk || seen1 == 0 || seen1 >= n) && lane < kDaLm && k1 <= seen1) {
            const double r = d.Z[((size_t)b * 2 + 0) * kTickJ + st + 1], phi = d.Z[((size_t)b * 2 + 1) * kTickJ + st + 1];
            const int s0 = 3 + 2 * lane;                                // slot of c_k
            double pb[5][5];                                            // pb[q][q2] = P(set[q2], set[q])
#pragma unroll
            for (int q = 0; q < 5; ++q)
#pragma unroll
                for (int q2 = 0; q2 < 5; ++q2) {
                    double val;
                    if (q2 < 3) val = nTR[q < 3 ? q : s0 + (q - 3)][q2];                   // row q2 at column set[q]
                    else if (q < 3) val = nTC[s0 + (q2 - 3)][q];                          // column q at row set[q2]
                    else val = nTD[lane][(q2 - 3) + 2 * (q - 3)];
                    pb[q][q2] = val;
                }
            const int code = assoc_code(pb, v.R, nS[0], nS[1], nS[2], nS[s0], nS[s0 + 1], r, phi);
            if (code >= 0) key1 = k1 * 4 + code;
        }
        key1 = wave_min(key1);
        if (lane == 0) d.keyp[((size_t)b * kTickJ + st + 1) * d.nwg + wg] = key1;
    }
}

// ------------------------------------------------------------------------------------------------ a whole round, resident
// The same corrections as k_da_begin + J x k_da_step, as ONE launch whose workgroups stay resident and meet between
// corrections (grid of nwg x B workgroups, all of which must be on the chip at once: the host takes this path only while
// nwg * B fits the CUs, see do_da_rounds).  What a kernel boundary cost per correction -- dispatch, the re-load of every
// thread's tracked entries and of its strips of corrections 0..s-1 -- stays in registers and LDS; what other workgroups
// need (the tracked entries of whichever landmark is associated next, the strips at its two indices, the candidates'
// keys) still travels through global memory -- every such store and every such load at agent scope (`sc1`: write-through,
// L1-bypassing), every storing wave drained (vmcnt(0)) in front of the workgroup barrier behind which ONE lane stores the
// workgroup's key slot, tagged with the step: the slot is the arrival flag, polled with sc1 loads, a workgroup barrier
// between the poll and the loads (the fence-free hand-off of MI355X_MICROARCH.md "Valid forms", first table row).  An
// agent-scope release + acquire pair per correction instead (L2 write-back + L1 invalidate, ~1.7 us each) made the
// resident kernel no faster than one launch per marker.  Every wait is bounded; an expired one latches NUSLAM_E_SYNC in
// the filter's status word instead of hanging the device.
//
// What the phase clock (make daclock, tools/exp_da_clock.py) showed per correction, and what was done about it:
//  * update() recomputes H, S, S^-1 and z_hat that associateLandmark has just formed for the matching candidate from the
//    same state and covariance (slam_library.cpp:212-218 / :265-270: the same functions on the same inputs, the same
//    bits).  The candidate lane that matches publishes them (16 numbers); the head of the next correction loads them
//    instead of redoing the transcendental chain (1.2 us) -- except for a first sighting, whose landmark is initialised
//    between the two.
//  * the candidates' own transcendentals (z_hat: three atan2, two sincos) need the new STATE only, not the covariance:
//    they run on an idle wave while the tracked covariance entries are still being updated; likewise the heading's
//    re-normalisation, on the head wave instead of one lane of the gain wave.
//  * the replay of rows / columns c, c+1 through corrections 0..s-1 reads its coefficients from LDS one correction ahead.
constexpr int kStatusSync = 9;      // NUSLAM_E_SYNC

// ------------------------------------------------------------------------------------------------ a round SERVED to the host
// The class API driven call by call (slam.cpp:279-318): associateLandmark(z) must hand its id back to the caller before the
// caller decides (initializeLandmark / skip / stop) and calls update() -- one host round trip per marker.  As a kernel launch
// per call that is ~6.7 us of launch latency plus the kernel on this box (tools/microbench/roundtrip.hip); a RESIDENT round
// that takes its commands from a mailbox in mapped pinned host memory answers in ~2.9 us plus its own work.  k_da_round<T, true>
// is that: the same trips as the trace-driven round, but
//   * at the top of every trip workgroup 0 publishes the verdict of the marker scanned in the previous trip (id, seen,
//     status: two 8-byte system-scope stores, each carrying the sequence number) and waits for the next command
//     { the correction the caller has decided on for the previous marker (update(z, id) [+ initializeLandmark]: MODE_FORCE;
//       none if the caller skipped it), the next marker to scan (associateLandmark(z')), or END };
//   * workgroup 0 alone reads the mailbox and hands the command on through device memory (agent scope): it is the one arbiter
//     of "no command came" -- after sv.timeout_ticks it closes the round by itself (PARK: nothing is applied, the trip ends the
//     round as END would), so a caller that stops calling never leaves waves spinning;  every other wait is bounded too;
//   * a command is eight 8-byte words read by eight lanes in one instruction and validated by its sequence number and an XOR
//     checksum instead of a second dependent read over PCIe.
struct DaServe {
    const long long* cmd;    // host (mapped, pinned): [0] (seq << 32) | (flags << 24) | id   [1] r  [2] phi  [3] checksum
                             //                        [4] [5] the correction's own (r, phi) when DA_F_ZOVR   [6] [7] unused
    long long* ans;          // host (mapped, pinned): [0] (seq << 32) | (unsigned)id   [1] (seq << 32) | kind << 28 | status << 24 | seen
    long long* keys;         // host (mapped, pinned): [nwg] (seq << 32) | key -- every workgroup's verdict on the marker of command seq goes
                             // STRAIGHT to the host (which takes the minimum and decodes it, decode_association's rule), before the
                             // workgroups meet: the meet, the drain in front of it and the next trip's top then run beside the host's turn
    int seq0;                // sequence number of the first command of this launch
    int timeout_ticks;       // 100 MHz ticks workgroup 0 waits for a command before it closes the round by itself
    double* mirror;          // host (mapped, pinned), may be null: the state vector the round leaves, stored by the workgroups as they
    long long* mtags;        // write it to the device's copy; mtags[workgroup] = mseq behind each workgroup's entries (TickTagged, ekf_tick.h)
    long long mseq;
};
enum { DA_F_CORR = 1, DA_F_INIT = 2, DA_F_SCAN = 4, DA_F_END = 8, DA_F_ZOVR = 16, DA_F_CLEAR = 32, DA_F_PARK = 64 };
constexpr long long kDaCmdMagic = 0x5a17c0de5a17c0dell;

__device__ inline long long ld_system(const long long* p)
{
    return __hip_atomic_load(const_cast<long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ inline void st_system(long long* p, long long x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline void da_answer(const DaServe& sv, int seq, int id, int seen, int status, int kind)
{
    st_system(sv.ans + 1, ((long long)seq << 32) | (long long)(((unsigned)kind << 28) | ((unsigned)(status & 15) << 24) | ((unsigned)seen & 0xffffffu)));
    st_system(sv.ans + 0, ((long long)seq << 32) | (long long)(unsigned)id);
}
// Every thread of the workgroup calls it; returns through cmd_sh[0..5] (LDS): word 0 (flags, id), r, phi, -, the correction's
// own (r, phi).  Wave 0 does the waiting; the barrier at the end publishes cmd_sh.
__device__ inline void da_wait_cmd(const DaServe& sv, int seq, int wg, long long* fwd, long long* cmd_sh, int* expired_sh)
{
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const long long t0 = (long long)wall_clock64();
        // (the other workgroups give workgroup 0 its whole timeout plus the time a healthy chip needs to hand a command on)
        const long long limit = wg == 0 ? (long long)sv.timeout_ticks : (long long)sv.timeout_ticks + 200000ll;
        long long w = 0;
        int ok = 0, expired = 0;
        for (;;) {
            if (lane < 8) w = wg == 0 ? ld_system(sv.cmd + lane) : ld_agent(fwd + lane);
            long long x = (lane < 6 && lane != 3) ? w : 0ll;
            x ^= __shfl_xor(x, 1, 64); x ^= __shfl_xor(x, 2, 64); x ^= __shfl_xor(x, 4, 64);
            const long long w0 = __shfl(w, 0, 64), w3 = __shfl(w, 3, 64), xs = __shfl(x, 0, 64);
            ok = (int)(w0 >> 32) == seq && (xs ^ kDaCmdMagic) == w3;
            if (ok) break;
            if ((long long)wall_clock64() - t0 > limit) { expired = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (expired) {
            // nothing came: the round ends here, nothing applied (workgroup 0 tells the host that command `seq` was NOT taken)
            const long long w0 = ((long long)seq << 32) | ((long long)(DA_F_PARK | DA_F_END) << 24);
            w = lane == 0 ? w0 : (lane == 3 ? (w0 ^ kDaCmdMagic) : 0ll);
        }
        if (wg == 0) {
            if (lane < 8) st_agent(fwd + lane, w);
            if (expired && lane == 0) da_answer(sv, seq, 0, 0, 0, 1);
        }
        if (lane < 6) cmd_sh[lane] = w;
        if (lane == 0) *expired_sh = (expired && wg != 0) ? 1 : 0;     // (a follower that never heard from workgroup 0: NUSLAM_E_SYNC)
    }
    __syncthreads();
}

constexpr int kDaRoundLds = (kTickJ * 5 + kTickJ * 2 + kTickJ * 5) * kDaSlots * (int)sizeof(double);

// (st_agent / ld_agent: ekf_tick.h)

#ifdef NUSLAM_DA_CLOCK
__device__ long long g_da_clock[4][16];       // debug builds (make daclock): per wave of workgroup 0, 100 MHz ticks per phase
#define DCK(k) do { const long long n__ = (long long)wall_clock64(); dck[k] += n__ - dct; dct = n__; } while (0)
#else
#define DCK(k) do { } while (0)
#endif
template <typename T, bool SERVED>
__global__ __launch_bounds__(256) void k_da_round(View v, TickObs o, int total_landmarks, const T* __restrict__ P, DaBuf d,
                                                  TickStep* __restrict__ plan, double* __restrict__ Kbuf,
                                                  double* __restrict__ Rbuf, double* __restrict__ Vbuf, int round_tag, DaServe srv)
{
    const int b = blockIdx.y, wg = blockIdx.x, nwg = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ld = v.ld, L = v.L, n = v.n, J = o.J;
    const T* Pb = P + (size_t)b * v.p_stride;
    TickStep* pl = plan + (size_t)b * kTickJ;
    double* Kb = Kbuf + (size_t)b * kTickJ * 2 * ld;
    double* Rb = Rbuf + (size_t)b * kTickJ * 5 * ld;
    const int t = da_index(wg, lane);
    const bool own = lane < 63 && (lane >= 3 || wg == 0);
    const int tr_ = t < L ? t : 0, tc_ = t < ld ? t : 0;

    extern __shared__ double da_lds[];
    double* Rl = da_lds;                                   // [kTickJ][5][64]  R_t'(:, t) of this workgroup's columns
    double* Kl = Rl + kTickJ * 5 * kDaSlots;               // [kTickJ][2][64]  K_t'(t, :) of its rows
    double* Ml = Kl + kTickJ * 2 * kDaSlots;               // [kTickJ][5][64]  M_t'(t, set_t') of its rows
    __shared__ double hd[20];                 // Hc[10], Sinv[4], lx, ly, dz0, dz1
    __shared__ int hi[2];
    __shared__ double Mpose[3][6];
    __shared__ double hist[kTickJ][12];       // Hc[10], c, no-correction flag of the corrections so far
    __shared__ double mcL[kTickJ][2][8];
    __shared__ double rcL[kTickJ][2][6];
    __shared__ double rcolL[kDaSlots][5];
    __shared__ double mrowL[kDaSlots][8];
    __shared__ double nTR[kDaSlots][3], nTC[kDaSlots][3], nS[kDaSlots], nTD[kDaLm][4];
    __shared__ double Zl[2][kTickJ];
    __shared__ double candZ[kDaLm][2];
    __shared__ int meet_sh[2];                // the reduced key of the next marker, whether every workgroup arrived
    __shared__ long long cmd_sh[6];           // SERVED: the command of this trip
    __shared__ int cmd_expired_sh;

    // ---- prologue (k_da_begin): the tracked entries out of the covariance after predict
    double e3[3] = { 0.0, 0.0, 0.0 }, sv = 0.0;            // wave 0: TR[.][t]; wave 1: TC[.][t] and the state entry
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            e3[q] = (double)Pb[(size_t)tr_ * ld + q];
            nTR[lane][q] = e3[q];
            if (own && t < L) st_agent(&d.TR[0][((size_t)b * 3 + q) * ld + t], e3[q]);
        }
    } else if (wave == 1) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            e3[q] = (double)Pb[(size_t)q * ld + tc_];
            nTC[lane][q] = e3[q];
            if (own && t < ld) st_agent(&d.TC[0][((size_t)b * 3 + q) * ld + t], e3[q]);
        }
        sv = v.s_in[(size_t)b * ld + tc_];
        nS[lane] = sv;
        if (own && t < ld) st_agent(&d.DS[0][(size_t)b * ld + t], sv);
    } else if (wave == 2) {
        const int k = wg * kDaLm + lane;
        if (lane < kDaLm) {
            const int c0 = 3 + 2 * k;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double val = 0.0;
                if (k < n) {
                    val = (double)Pb[(size_t)(c0 + (e >> 1)) * ld + c0 + (e & 1)];
                    st_agent(&d.TD[0][((size_t)b * 4 + e) * n + k], val);
                }
                nTD[lane][e] = val;
            }
        }
    } else if (!SERVED && lane < kTickJ) {
        double a = 0.0, bb = 0.0;
        if (lane < J) {
            a = o.a ? o.a[b * o.stride + o.off + lane] : o.a0[lane];
            bb = o.b ? o.b[b * o.stride + o.off + lane] : o.b0[lane];
        }
        double r, phi;
        if (o.cartesian) cartesian2polar(a, bb, r, phi);
        else { r = a; phi = bb; }
        Zl[0][lane] = r;
        Zl[1][lane] = phi;
    }
#ifdef NUSLAM_DA_CLOCK
    long long dck[16] = { 0 }, dct = (long long)wall_clock64();
#endif
    int seen, cached, brk, status;
    {
        const int* ci = v.c_in + b * C_WORDS;
        seen = ci[C_SEEN]; cached = ci[C_SEEN_CACHED]; brk = ci[C_BRK]; status = ci[C_STATUS];
    }
    unsigned live = 0;                                     // corrections so far that changed P
    int key = kNoKey;                                      // the reduced key of marker st (from the meet)
    int scanned = 0, match_id = 0;                         // SERVED: the previous trip scanned a marker; the landmark it matched
    if (SERVED) brk = 0;                                   // (the caller's own chain decides about the break, slam.cpp:301-316)

    for (int st = -1; st < (SERVED ? kTickJ : J); ++st) {
        bool last = st + 1 == J;
        int f_flags = 0, f_id = 0;                         // SERVED: this trip's command
        if (SERVED) {
            // the verdict of the marker scanned in the previous trip (slam_library.cpp:238-252) goes to the caller ...
            int ans_id = 0;
            match_id = 0;                                  // (what a scan formed for its match is good for the very next correction only)
            if (scanned) {
                const Assoc a = decode_association(n, seen, 0, status, key);
                ans_id = a.id; seen = a.new_seen; status = a.new_status;
                if (key != kNoKey && (key & 3) == 0) match_id = key >> 2;
            }
            const int seq = srv.seq0 + st + 1;
            // (a scanning command is answered by the workgroups' keys themselves, see below; any other by this word)
            if (st >= 0 && wg == 0 && tid == 0 && !scanned) da_answer(srv, seq - 1, ans_id, seen, status, 0);
            // ... and the caller's next call comes back
            da_wait_cmd(srv, seq, wg, d.fwd + 8 * (size_t)b, cmd_sh, &cmd_expired_sh);
            const long long w0 = cmd_sh[0];
            f_flags = (int)((w0 >> 24) & 0xff);
            f_id = (int)(w0 & 0xffffffll);
            if (cmd_expired_sh && status == 0) status = kStatusSync;
            if (f_flags & DA_F_CLEAR) status = 0;
            if (st < 0) f_flags &= ~DA_F_CORR;             // (the first command of a round can only scan)
            const bool scan = (f_flags & DA_F_SCAN) != 0 && st + 1 < kTickJ;
            last = !scan && ((f_flags & DA_F_END) != 0 || st + 1 >= kTickJ);
            if (tid == 0) {
                if (scan) { Zl[0][st + 1] = __longlong_as_double(cmd_sh[1]); Zl[1][st + 1] = __longlong_as_double(cmd_sh[2]); }
                if (st >= 0 && (f_flags & DA_F_ZOVR)) { Zl[0][st] = __longlong_as_double(cmd_sh[4]); Zl[1][st] = __longlong_as_double(cmd_sh[5]); }
            }
            scanned = scan ? 1 : 0;
            __syncthreads();
        }
        bool nocorr = true;
        int c = 3;
        double m5[5] = { 0.0, 0.0, 0.0, 0.0, 0.0 }, bef = 0.0, aft = 0.0;  // wave 1: M_s(t, set_s)
        if (st >= 0) {
            const int par = st & 1;
            const double* TRc = d.TR[par] + (size_t)b * 3 * ld;
            const double* TCc = d.TC[par] + (size_t)b * 3 * ld;
            const double* TDc = d.TD[par] + (size_t)b * 4 * n;
            const double* sc = d.DS[par] + (size_t)b * ld;
            DCK(0);
            // ---- the decision, by every thread: associateLandmark's verdict, then slam.cpp:295-316
            Decision dd;
            if (SERVED) {
                // the caller has decided (slam.cpp:295-316 ran on the host): update(z, id) was called, or it was not
                dd.id = f_id; dd.new_seen = seen; dd.new_brk = brk; dd.new_status = status;
                dd.skip = (f_flags & DA_F_CORR) == 0;
                dd.init = !dd.skip && (f_flags & DA_F_INIT) != 0;
                if (!dd.skip && (f_id < 1 || f_id > n)) { dd.skip = true; dd.init = false; if (dd.new_status == 0) dd.new_status = kStatusBounds; }
                dd.c = dd.skip ? 3 : 3 + 2 * (f_id - 1);
            } else {
                int id_raw, seen_now, status_now;
                const Assoc a = decode_association(n, seen, brk, status, key);
                id_raw = a.id; seen_now = a.new_seen; status_now = a.new_status;
                if (o.ids != nullptr && o.ids[b * o.stride + o.off + st] < 0) { id_raw = -1; seen_now = seen; status_now = status; }
                dd = resolve(n, id_raw, seen_now, cached, brk, status_now, MODE_DA, total_landmarks);
            }
            c = dd.c;
            const int setv[5] = { 0, 1, 2, c, c + 1 };
            // update() of a landmark associateLandmark has just examined (SERVED: of the very landmark it matched, with the marker it
            // was given)
            const bool matched = !dd.skip && !dd.init && (!SERVED || (f_id == match_id && (f_flags & DA_F_ZOVR) == 0));
            DCK(1);

            // ---- loads that depend on the landmark
            double g0 = 0.0, g1 = 0.0;
            double val = 0.0;                              // wave 2: one tracked entry / state entry / packet entry per lane
            double th_old = 0.0, kp0 = 0.0, kp1 = 0.0;     // wave 2: the heading before, K_s(0, :)
            if (wave == 0) {
                g0 = (double)Pb[(size_t)tr_ * ld + c];
                g1 = (double)Pb[(size_t)tr_ * ld + c + 1];
            } else if (wave == 1) {
                g0 = (double)Pb[(size_t)c * ld + tc_];
                g1 = (double)Pb[(size_t)(c + 1) * ld + tc_];
            } else if (wave == 2) {
                // lane 5 q + q2 < 25: P_{s-1}(set[q2], set[q]); lanes 32..36: th, x, y, lx, ly; lanes 40..55: the matching
                // candidate's H, psi^-1, z_hat
                const int e = lane < 25 ? lane : 0;
                const int q = e / 5, q2 = e % 5;
                const double* src;
                if (q2 < 3) src = TRc + (size_t)q2 * ld + setv[q];
                else if (q < 3) src = TCc + (size_t)q * ld + setv[q2];
                else src = TDc + (size_t)((q2 - 3) + 2 * (q - 3)) * n + (c - 3) / 2;
                const int sl = lane - 32;
                if (sl >= 0 && sl < 5) src = sc + setv[sl];
                if (matched && sl >= 8 && sl < 24) src = d.AP[par] + ((size_t)b * n + (c - 3) / 2) * 16 + (sl - 8);
                val = ld_agent(src);
            } else {
                // lanes (t', e): the coefficients of rows / columns c + e through correction t'
                const int tp = lane >> 1, e = lane & 1;
                if (lane < 2 * kTickJ && tp < st && !dd.skip) {
                    const int cp = (int)hist[tp][10];
                    const int i = c + e;
                    const double K0 = ld_agent(&Kb[(size_t)(tp * 2 + 0) * ld + i]), K1 = ld_agent(&Kb[(size_t)(tp * 2 + 1) * ld + i]);
                    double rr[5];
#pragma unroll
                    for (int q = 0; q < 5; ++q) rr[q] = ld_agent(&Rb[(size_t)(tp * 5 + q) * ld + i]);
#pragma unroll
                    for (int q = 0; q < 5; ++q) {
                        double kh = 0.0;
                        kh = fma(K0, hist[tp][0 + 2 * q], kh);
                        kh = fma(K1, hist[tp][1 + 2 * q], kh);
                        const int sidx = q < 3 ? q : cp + (q - 3);
                        mcL[tp][e][q] = (i == sidx ? 1.0 : 0.0) - kh;
                        rcL[tp][e][q] = rr[q];
                    }
                    mcL[tp][e][5] = (i > 2 && i < cp) ? 1.0 : 0.0;
                    mcL[tp][e][6] = (i > cp + 1) ? 1.0 : 0.0;
                }
            }
            // ---- the replay coefficients are in LDS: the replay (waves 0, 1) runs beside the head (wave 2), whose loads
            // are still in flight (an LDS-only barrier: nobody waits for vector memory here)
            lds_barrier();
            if (wave == 0 && !dd.skip) {
                // rows c, c+1 of P_{s-1} at column t: the gathered entries replayed through corrections 0..s-1, two
                // corrections per trip so that the second one's coefficients are on their way from LDS while the first
                // is applied (a register-rotating prefetch cost more in moves than it hid)
                unsigned todo = live;
                while (todo) {
                    const int ta = __builtin_ctz(todo);
                    todo &= todo - 1;
                    const bool two = todo != 0;
                    const int tb = two ? __builtin_ctz(todo) : ta;
                    if (two) todo &= todo - 1;
                    double ra[5], rb[5], ma[7], mb[7], na[7], nb[7];
#pragma unroll
                    for (int q = 0; q < 5; ++q) { ra[q] = Rl[(ta * 5 + q) * kDaSlots + lane]; rb[q] = Rl[(tb * 5 + q) * kDaSlots + lane]; }
#pragma unroll
                    for (int q = 0; q < 7; ++q) { ma[q] = mcL[ta][0][q]; mb[q] = mcL[ta][1][q]; na[q] = mcL[tb][0][q]; nb[q] = mcL[tb][1][q]; }
                    g0 = p1_entry<T>(ma, ra, g0, ma[5], ma[6]);
                    g1 = p1_entry<T>(mb, ra, g1, mb[5], mb[6]);
                    if (two) {
                        g0 = p1_entry<T>(na, rb, g0, na[5], na[6]);
                        g1 = p1_entry<T>(nb, rb, g1, nb[5], nb[6]);
                    }
                }
            } else if (wave == 1 && !dd.skip) {
                unsigned todo = live;
                while (todo) {
                    const int ta = __builtin_ctz(todo);
                    todo &= todo - 1;
                    const bool two = todo != 0;
                    const int tb = two ? __builtin_ctz(todo) : ta;
                    if (two) todo &= todo - 1;
                    double mta[5], mtb[5], ra0[5], ra1[5], rb0[5], rb1[5];
#pragma unroll
                    for (int q = 0; q < 5; ++q) {
                        mta[q] = Ml[(ta * 5 + q) * kDaSlots + lane]; mtb[q] = Ml[(tb * 5 + q) * kDaSlots + lane];
                        ra0[q] = rcL[ta][0][q]; ra1[q] = rcL[ta][1][q]; rb0[q] = rcL[tb][0][q]; rb1[q] = rcL[tb][1][q];
                    }
                    const int cpa = (int)hist[ta][10], cpb = (int)hist[tb][10];
                    const double bpa = (t > 2 && t < cpa) ? 1.0 : 0.0, apa = (t > cpa + 1) ? 1.0 : 0.0;
                    const double bpb = (t > 2 && t < cpb) ? 1.0 : 0.0, apb = (t > cpb + 1) ? 1.0 : 0.0;
                    g0 = p1_entry<T>(mta, ra0, g0, bpa, apa);
                    g1 = p1_entry<T>(mta, ra1, g1, bpa, apa);
                    if (two) {
                        g0 = p1_entry<T>(mtb, rb0, g0, bpb, apb);
                        g1 = p1_entry<T>(mtb, rb1, g1, bpb, apb);
                    }
                }
            } else if (wave == 2) {
                const double th = lane_bcast(val, 32), x = lane_bcast(val, 33), y = lane_bcast(val, 34);
                double lx = lane_bcast(val, 35), ly = lane_bcast(val, 36);
                const double r = Zl[0][st], phi = Zl[1][st];
                bool skip0 = dd.skip;
                int stt = dd.new_status;
                double Hc[10], Si[4], dz0 = 0.0, dz1 = 0.0;
#pragma unroll
                for (int k = 0; k < 10; ++k) Hc[k] = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) Si[k] = 0.0;
                if (matched) {
#pragma unroll
                    for (int k = 0; k < 10; ++k) Hc[k] = lane_bcast(val, 40 + k);
#pragma unroll
                    for (int k = 0; k < 4; ++k) Si[k] = lane_bcast(val, 50 + k);
                    dz0 = r - lane_bcast(val, 54);                      // :272, bearing innovation not wrapped
                    dz1 = phi - lane_bcast(val, 55);
                } else if (!skip0) {
                    if (!SERVED || dd.init) {                           // (trace-driven: not matched means a first sighting)
                        lx = x + r * cos(phi + th);                     // initializeLandmark, slam_library.cpp:255-261
                        ly = y + r * sin(phi + th);
                    }
                    double pb[5][5], S[4];
#pragma unroll
                    for (int a = 0; a < 5; ++a)
#pragma unroll
                        for (int a2 = 0; a2 < 5; ++a2) pb[a][a2] = lane_bcast(val, 5 * a + a2);
                    jacobian_compact(x, y, lx, ly, Hc);                 // :268
                    innovation_cov_block(pb, Hc, v.R, S);               // :270
                    if (inv2(S, Si)) { skip0 = true; if (stt == 0) stt = kStatusSingular; }
                    double zr, zb;
                    measurement(th, x, y, lx, ly, zr, zb);              // :265
                    dz0 = r - zr;
                    dz1 = phi - zb;
                }
                if (!skip0) {
                    double pc[5], K[2], m[5];                           // the gain rows of the pose: pc[a2] = P(lane, set[a2])
#pragma unroll
                    for (int a2 = 0; a2 < 5; ++a2) pc[a2] = __shfl(val, 5 * a2 + (lane < 3 ? lane : 0), 64);
                    gain_row(pc, Hc, Si, lane < 3 ? lane : 0, setv, K, m);
                    if (lane < 3) {
#pragma unroll
                        for (int a2 = 0; a2 < 5; ++a2) Mpose[lane][a2] = m[a2];
                    }
                    kp0 = lane_bcast(K[0], 0);
                    kp1 = lane_bcast(K[1], 0);
                }
                th_old = th;
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < 10; ++k) hd[k] = Hc[k];
#pragma unroll
                    for (int k = 0; k < 4; ++k) hd[10 + k] = Si[k];
                    hd[14] = lx; hd[15] = ly; hd[16] = dz0; hd[17] = dz1;
                    hi[0] = skip0 ? 1 : 0; hi[1] = stt;
                }
            }
            DCK(2);
            __syncthreads();
            DCK(3);

            nocorr = hi[0] != 0;
            const int new_status = hi[1];
            const double lx = hd[14], ly = hd[15];
            double Hc[10], Si[4];
#pragma unroll
            for (int k = 0; k < 10; ++k) Hc[k] = hd[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) Si[k] = hd[10 + k];
            double n3[3] = { e3[0], e3[1], e3[2] };

            if (wave == 0) {
                if (!nocorr) {
                    const double rs[5] = { e3[0], e3[1], e3[2], g0, g1 };
#pragma unroll
                    for (int q = 0; q < 5; ++q) {
                        Rl[(st * 5 + q) * kDaSlots + lane] = rs[q];
                        rcolL[lane][q] = rs[q];
                    }
#pragma unroll
                    for (int q = 0; q < 3; ++q) n3[q] = p1_entry<T>(Mpose[q], rs, e3[q], 0.0, 0.0);
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    nTR[lane][q] = n3[q];
                    e3[q] = n3[q];
                }
            } else if (wave == 1) {
                double snew = sv;
                if (dd.init && !dd.skip) {                              // initializeLandmark ran (also when update() then threw)
                    if (t == c) snew = lx;
                    if (t == c + 1) snew = ly;
                }
                if (!nocorr) {
                    const double pc[5] = { e3[0], e3[1], e3[2], g0, g1 };
                    double K[2];
                    gain_row(pc, Hc, Si, t, setv, K, m5);
                    bef = (t > 2 && t < c) ? 1.0 : 0.0;
                    aft = (t > c + 1) ? 1.0 : 0.0;
                    Kl[(st * 2 + 0) * kDaSlots + lane] = K[0];
                    Kl[(st * 2 + 1) * kDaSlots + lane] = K[1];
#pragma unroll
                    for (int q = 0; q < 5; ++q) { Ml[(st * 5 + q) * kDaSlots + lane] = m5[q]; mrowL[lane][q] = m5[q]; }
                    mrowL[lane][5] = bef; mrowL[lane][6] = aft;
                    double acc = 0.0;                                   // state += K (z - z_hat)  (:275)
                    acc = fma(K[0], hd[16], acc);
                    acc = fma(K[1], hd[17], acc);
                    snew = snew + acc;
                }
                if (t != 0) {                                           // (the heading: wave 2, which re-normalises it, :276)
                    nS[lane] = snew;
                    sv = snew;
                }
            } else if (wave == 2) {
                double th = th_old;
                if (!nocorr) {
                    double acc = 0.0;
                    acc = fma(kp0, hd[16], acc);
                    acc = fma(kp1, hd[17], acc);
                    th = normalize_angle(th + acc);                     // :275-276
                }
                if (lane == 0) nS[0] = th;
            } else if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 10; ++k) hist[st][k] = Hc[k];
                hist[st][10] = (double)c;
                hist[st][11] = nocorr ? 1.0 : 0.0;
                if (wg == 0) {
                    // the correction's record for the pass over P, the control words, the id log
                    TickStep* ps = pl + st;
                    ps->skip = nocorr ? 1 : 0; ps->init = (dd.init && !dd.skip) ? 1 : 0; ps->c = c; ps->id = dd.id;
#pragma unroll
                    for (int k = 0; k < 10; ++k) ps->Hc[k] = Hc[k];
#pragma unroll
                    for (int k = 0; k < 4; ++k) ps->Sinv[k] = Si[k];
                    ps->dz[0] = hd[16]; ps->dz[1] = hd[17]; ps->lxy[0] = lx; ps->lxy[1] = ly;
                    if (last) {
                        int* co = v.c_out + b * C_WORDS;
                        co[C_SEEN] = dd.new_seen; co[C_SEEN_CACHED] = cached; co[C_BRK] = dd.new_brk; co[C_STATUS] = new_status;
                    }
                    if (v.id_log && o.log_slot0 >= 0) v.id_log[(size_t)b * v.log_stride + o.log_slot0 + st] = dd.id;
                }
            }
            if (!nocorr) live |= 1u << st;
            seen = dd.new_seen; brk = dd.new_brk; status = new_status;
            DCK(4);
        }
        __syncthreads();
        DCK(5);
        if (wave == 1 && t == 0) sv = nS[0];

        // ---- wave 1: columns 0..2 at this row (needs the row role's strips of slots 0..2), the diagonal blocks, and --
        // everything a candidate's psi needs being then in THIS wave's hands -- the next marker's candidates up to
        // psi^-1 (H from the new state, psi = H P H^T + R);  wave 3: their z_hat (three atan2, two sincos: the long pole
        // of this phase);  waves 0, 2: the stores of the correction.
        const bool cand = !last && (!SERVED || scanned) && !(brk || seen == 0 || seen >= n) && lane < kDaLm && wg * kDaLm + lane + 1 <= seen;
        const int s0 = 3 + 2 * (lane < kDaLm ? lane : 0);               // slot of the candidate's first index
        double cH[10], cPi[4];                                          // wave 1: the candidate's H and psi^-1
        int csing = 0;
        if (wave == 1) {
            if (st >= 0) {
                double n3[3] = { e3[0], e3[1], e3[2] };
                if (!nocorr) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) n3[q] = p1_entry<T>(m5, rcolL[q], e3[q], bef, aft);
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    nTC[lane][q] = n3[q];
                    e3[q] = n3[q];
                }
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int idx = 64 * h2 + lane;                     // entry e of landmark lm: P(c_lm + (e & 1), c_lm + (e >> 1))
                    const int lm = idx >> 2, e = idx & 3;
                    const int k = wg * kDaLm + lm;
                    if (lm < kDaLm && k < n) {
                        double val = nTD[lm][e];
                        const int si = 3 + 2 * lm + (e & 1), sj = 3 + 2 * lm + (e >> 1);
                        if (!nocorr) val = p1_entry<T>(mrowL[si], rcolL[sj], val, mrowL[si][5], mrowL[si][6]);
                        nTD[lm][e] = val;
                    }
                }
            }
            if (cand) {
                double pb[5][5], psi[4];                                // pb[q][q2] = P(set[q2], set[q])
                jacobian_compact(nS[1], nS[2], nS[s0], nS[s0 + 1], cH);                   // :212
#pragma unroll
                for (int q = 0; q < 5; ++q)
#pragma unroll
                    for (int q2 = 0; q2 < 5; ++q2) {
                        double val;
                        if (q2 < 3) val = nTR[q < 3 ? q : s0 + (q - 3)][q2];
                        else if (q < 3) val = nTC[s0 + (q2 - 3)][q];                      // (written by this wave just above)
                        else val = nTD[lane][(q2 - 3) + 2 * (q - 3)];
                        pb[q][q2] = val;
                    }
                innovation_cov_block(pb, cH, v.R, psi);                 // :215
                csing = inv2(psi, cPi);
            }
        } else if (wave == 0) {
            if (st >= 0 && own) {
                const int par = st & 1;
                double* sn = d.DS[par ^ 1] + (size_t)b * ld;
                if (t < L && !nocorr) {
#pragma unroll
                    for (int q = 0; q < 5; ++q) st_agent(&Rb[(size_t)(st * 5 + q) * ld + t], rcolL[lane][q]);
                    if (Vbuf) {                                         // V_s = H_s R_s beside R_s (read by the NEXT kernel only: plain stores)
                        double* Vb = Vbuf + (size_t)b * kTickJ * 2 * ld;
                        const double rs[5] = { rcolL[lane][0], rcolL[lane][1], rcolL[lane][2], rcolL[lane][3], rcolL[lane][4] };
                        Vb[(size_t)(st * 2 + 0) * ld + t] = hp_entry(hist[st], rs, 0);
                        Vb[(size_t)(st * 2 + 1) * ld + t] = hp_entry(hist[st], rs, 1);
                    }
                }
                if (t < ld) {
                    if (!nocorr) {
                        st_agent(&Kb[(size_t)(st * 2 + 0) * ld + t], Kl[(st * 2 + 0) * kDaSlots + lane]);
                        st_agent(&Kb[(size_t)(st * 2 + 1) * ld + t], Kl[(st * 2 + 1) * kDaSlots + lane]);
                    }
                    const double sx = nS[lane];
                    st_agent(&sn[t], sx);
                    if (last) v.s_out[(size_t)b * ld + t] = sx;
                    if (SERVED && last && srv.mirror) st_sys(srv.mirror + t, sx);
                }
            }
        } else if (wave == 2) {
            if (st >= 0 && own && t < L) {
                const int par = st & 1;
                double* TRn = d.TR[par ^ 1] + (size_t)b * 3 * ld;
#pragma unroll
                for (int q = 0; q < 3; ++q) st_agent(&TRn[(size_t)q * ld + t], nTR[lane][q]);
            }
        } else {
            if (cand) {
                double zr, zb;
                measurement(nS[0], nS[1], nS[2], nS[s0], nS[s0 + 1], zr, zb);              // :218
                candZ[lane][0] = zr; candZ[lane][1] = zb;
            }
        }
        DCK(6);
        if (SERVED && last && srv.mirror && st >= 0) {                  // (uniform) this workgroup's share of the state is in the host's mirror
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) st_sys(srv.mtags + wg, srv.mseq);
        }
        if (last) {
            // (SERVED: the END command has been carried out -- a round that parked has said so already)
            if (SERVED && wg == 0 && tid == 0 && (f_flags & DA_F_PARK) == 0) da_answer(srv, srv.seq0 + st + 1, 0, seen, status, 0);
            break;
        }
        __syncthreads();
        DCK(7);

        // ---- wave 1: the candidates' distances against the marker (k_associate), the key;  waves 0, 2: what wave 1 left
        // in LDS for the other workgroups (columns 0..2, the diagonal blocks) goes out
        if (wave == 0) {
            if (st >= 0) {
                double* TDn = d.TD[(st & 1) ^ 1] + (size_t)b * 4 * n;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int idx = 64 * h2 + lane;
                    const int lm = idx >> 2, e = idx & 3;
                    const int k = wg * kDaLm + lm;
                    if (lm < kDaLm && k < n) st_agent(&TDn[(size_t)e * n + k], nTD[lm][e]);
                }
            }
        } else if (wave == 2) {
            if (st >= 0 && own && t < ld) {
                double* TCn = d.TC[(st & 1) ^ 1] + (size_t)b * 3 * ld;
#pragma unroll
                for (int q = 0; q < 3; ++q) st_agent(&TCn[(size_t)q * ld + t], nTC[lane][q]);
            }
        } else if (wave == 1) {
            int key1 = kNoKey;
            if (cand) {
                const double r = Zl[0][st + 1], phi = Zl[1][st + 1];
                const double zr = candZ[lane][0], zb = candZ[lane][1];
                const double dz0 = r - zr, dz1 = phi - zb;             // :229
                int code = -1;
                if (csing) code = 2;
                else {
                    double w0 = 0.0, w1 = 0.0, dd2 = 0.0;               // (dz^T psi^-1) dz, :231
                    w0 = fma(dz0, cPi[0], w0); w0 = fma(dz1, cPi[1], w0);
                    w1 = fma(dz0, cPi[2], w1); w1 = fma(dz1, cPi[3], w1);
                    dd2 = fma(w0, dz0, dd2); dd2 = fma(w1, dz1, dd2);
                    if (dd2 < 0.01) code = 0;                                           // :238
                    else if ((dd2 > 0.01) && (dd2 < 60)) code = 1;                      // :243
                }
                if (code >= 0) key1 = (wg * kDaLm + lane + 1) * 4 + code;
                if (code == 0) {
                    // a match: update() will want exactly these (same state, same covariance, same functions)
                    double* ap = d.AP[(st + 1) & 1] + ((size_t)b * n + wg * kDaLm + lane) * 16;
#pragma unroll
                    for (int k = 0; k < 10; ++k) st_agent(ap + k, cH[k]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) st_agent(ap + 10 + k, cPi[k]);
                    st_agent(ap + 14, zr);
                    st_agent(ap + 15, zb);
                }
            }
            key1 = wave_min(key1);
            meet_sh[0] = key1;                                          // (every lane holds the minimum)
            if (SERVED && scanned && lane == 0) st_system(srv.keys + wg, ((long long)(srv.seq0 + st + 1) << 32) | (long long)(unsigned)key1);
        }
        DCK(8);
        // ---- meet the other workgroups of this filter: everything stored above has left, then the tagged key slot; the
        // slots of all workgroups are polled until they carry this step's tag
        {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int tag = round_tag + st + 2;
            long long* slots = d.keyt + ((size_t)b * kTickJ + st + 1) * nwg * kDaSlotStride;
            if (wave == 0) {
                if (lane == 0) st_agent(slots + (size_t)wg * kDaSlotStride, ((long long)tag << 32) | (unsigned)meet_sh[0]);
                int kmin = kNoKey, ok = 1;
                for (int i = lane; i < nwg; i += 64) {
                    long long w = 0;
                    int got = 0;
                    for (int it = 0; it < (1 << 20); ++it) {
                        w = ld_agent(slots + (size_t)i * kDaSlotStride);
                        if ((int)(w >> 32) == tag) { got = 1; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    ok &= got;
                    const int kk = (int)(unsigned)(w & 0xffffffffll);
                    kmin = kk < kmin ? kk : kmin;
                }
                kmin = wave_min(kmin);
                ok = __all(ok);
                if (lane == 0) { meet_sh[0] = kmin; meet_sh[1] = ok; }
            }
            __syncthreads();
            key = meet_sh[0];
            if (!meet_sh[1] && status == 0) status = kStatusSync;
        }
        DCK(9);
    }
#ifdef NUSLAM_DA_CLOCK
    if (lane == 0 && b == 0 && wg == 0)
        for (int k = 0; k < 16; ++k) g_da_clock[wave][k] = dck[k];
#endif
}

} // namespace nuslam
