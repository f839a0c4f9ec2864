// ekf_rank.h -- the pass over P as a rank-2m update on the matrix cores.
//
// update() writes P_s = (I - K_s H_s) P_{s-1} (slam_library.cpp:279).  The same product re-associated is
//
//      P_s = P_{s-1} - K_s (H_s P_{s-1}) = P_{s-1} - K_s V_s,          V_s = H_s R_s   (2 x len),
//
// and the strips of ekf_tick.h are exactly its factors: K_s (len x 2) and the five prior rows R_s = P_{s-1}(set_s, :) of
// which V_s is a 2 x 5 combination (hp_entry).  A round of m corrections is therefore ONE rank-2m update
//
//      P_m = P_0 - [K_1 ... K_m] [V_1; ...; V_m]                        (len x 2m) (2m x len),
//
// 2 FMAs per element and correction where the (I - K H) P chain of k_tick_apply spends 7: the pass becomes a streaming
// kernel again (2 len^2 w bytes, 4 m len^2 flop).  It runs on v_mfma_f64_16x16x4_f64 with the tile of P as the accumulator:
// operands in MFMA layout need no per-FMA broadcast (a VALU version of the same sum was LDS-bound), and the
// f64 MFMA is a k-ordered fma chain, so every element is   acc = fma(V_f(col), -K_f(row), acc),  f = 2 s + r ascending --
// a definition tick_carry reproduces bit for bit for the overlapped run.
//
// Not bit-identical to the oracle's (I - K H) P (same algebra, different rounding: 1e-13-ish per entry from a warm
// state; contract 1e-6, tests/test_gpu_depth.py).  A round that holds a FIRST SIGHTING keeps the exact chain: there
// P(c, c) = INT_MAX (slam_library.cpp:30) is cancelled to ~1e-3 and the two forms lose different digits.  The plan says
// which rounds those are (round_flags); this kernel skips them and k_tick_apply(only_if_init) takes them.
//
// Tiling.  MFMA roles as k_flush: D[m][n], n = lane & 15 <-> a ROW of P (VEC interleaved 16-row sub-tiles, so a lane's
// 16-byte access covers VEC consecutive rows of one column), m = (lane >> 4) + 4 reg <-> a COLUMN.
//   row group = 16 VEC rows, column group = 16 columns; a wave owns RB x CB groups, a workgroup WR x WC waves.
//   A operand = V_f(col),  f = 4 ks + (lane >> 4), col = column group base + (lane & 15)      8 bytes per lane
//   B operand = -K_f(row .. row + VEC)                                                       8 VEC bytes per lane
// Operands shared by several waves of the workgroup are staged in LDS once (K if WC > 1, V if WR > 1), the others are
// loaded straight into registers in operand layout.  Skipped corrections and factors beyond the round are zero operands.
#pragma once

namespace nuslam {

typedef double rank_d4 __attribute__((ext_vector_type(4)));

constexpr int kRankNF = 2 * kTickJ;        // factors of a full round
constexpr int kRankKS = kRankNF / 4;       // MFMA k-steps

template <typename T, int RB, int CB, int WR, int WC>
struct RankTile {
    static constexpr int VEC = 16 / (int)sizeof(T);
    static constexpr int RG = 16 * VEC;                    // rows of a row group
    static constexpr int WROWS = WR * RB * RG, WCOLS = WC * CB * 16;
    static constexpr int NT = 64 * WR * WC;
    static constexpr bool K_LDS = WC > 1, V_LDS = WR > 1;
    static constexpr int KLD = WROWS + 2;                  // de-phases consecutive factor rows in the LDS banks
    static constexpr int VLD = WCOLS + 16;                 // row stride = 128 bytes mod 256: the four k-rows of an A read hit disjoint banks
    static constexpr size_t lds_bytes = sizeof(double) * ((K_LDS ? (size_t)kRankNF * KLD : 0) + (V_LDS ? (size_t)kRankNF * VLD : 0));
};

// tiles_r x tiles_c workgroup tiles per filter.  by_filter: blockIdx.x = 8 k + x puts filter (k / tiles) * 8 + x on the
// XCD that round-robin placement gives residue x (speed only): a filter's K / V strips are then fetched into ONE L2
// instead of all eight.  Otherwise (few filters) blockIdx.y is the filter and residue x owns the tile rows = x mod 8,
// so an XCD fetches only its rows of K.
template <typename T, int RB, int CB, int WR, int WC>
__global__ __launch_bounds__(64 * WR * WC) void k_tick_rank(View v, int J, const TickStep* __restrict__ plan,
                                                            const double* __restrict__ Kbuf, const double* __restrict__ Vbuf,
                                                            const T* __restrict__ Pin, T* __restrict__ Pout, int check_init,
                                                            int tiles_r, int tiles_c, int by_filter)
{
    typedef RankTile<T, RB, CB, WR, WC> TL;
    typedef Pack16<T> vec_t;
    typedef Pack16<double> d2_t;
    constexpr int VEC = TL::VEC, RG = TL::RG, KS = kRankKS, NF = kRankNF, NT = TL::NT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int n16 = lane & 15, g4 = lane >> 4;
    const int ld = v.ld, L = v.L;
    int b, tr, tc;
    if (by_filter) {
        const int x = blockIdx.x & 7, k = blockIdx.x >> 3, tpf = tiles_r * tiles_c;
        b = (k / tpf) * 8 + x;
        if (b >= v.B) return;
        const int tile = k % tpf;
        tr = tile % tiles_r; tc = tile / tiles_r;
    } else {
        const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
        b = blockIdx.y;
        tr = (k / tiles_c) * 8 + x; tc = k % tiles_c;
        if (tr >= tiles_r) return;
    }
    const TickStep* pl = plan + (size_t)b * kTickJ;
    unsigned actmask; bool any_init;
    round_flags(pl, J, actmask, any_init);
    if (check_init && any_init) return;                   // this filter's round goes through the exact chain

    extern __shared__ double lds_rank[];
    double* sK = lds_rank;                                // [NF][KLD]   -K, masked
    double* sV = lds_rank + (TL::K_LDS ? NF * TL::KLD : 0);   // [NF][VLD]   V, masked
    const double* Kb = Kbuf + (size_t)b * NF * ld;
    const double* Vb = Vbuf + (size_t)b * NF * ld;
    const int row_g0 = tr * TL::WROWS, col_g0 = tc * TL::WCOLS;
    const int row_w0 = row_g0 + wr * RB * RG, col_w0 = col_g0 + wc * CB * 16;

    // ---- every load of the kernel, in the order the results are needed (vmcnt retires in order): staged strips, the
    // operands that go straight to registers, then the tile.  Addresses are clamped, never branched around; masks are
    // multiplicative (a `cond ? load : 0` becomes a branch that ends in s_waitcnt vmcnt(0), ekf_deferred.h).
    constexpr int KCH = TL::K_LDS ? NF * (TL::WROWS / 2) / NT : 1;
    constexpr int VCH = TL::V_LDS ? NF * (TL::WCOLS / 2) / NT : 1;
    static_assert(!TL::K_LDS || NF * (TL::WROWS / 2) % NT == 0, "K staging: whole chunks per thread");
    static_assert(!TL::V_LDS || NF * (TL::WCOLS / 2) % NT == 0, "V staging: whole chunks per thread");
    d2_t kst[KCH], vst[VCH];
    if (TL::K_LDS) {
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int e = tid + i * NT;
            const int f = e / (TL::WROWS / 2), i2 = (e % (TL::WROWS / 2)) * 2;
            const int row = row_g0 + i2 < ld ? row_g0 + i2 : 0;
            kst[i] = *reinterpret_cast<const d2_t*>(Kb + (size_t)f * ld + row);
        }
    }
    if (TL::V_LDS) {
#pragma unroll
        for (int i = 0; i < VCH; ++i) {
            const int e = tid + i * NT;
            const int f = e / (TL::WCOLS / 2), i2 = (e % (TL::WCOLS / 2)) * 2;
            const int col = col_g0 + i2 < ld ? col_g0 + i2 : 0;
            vst[i] = *reinterpret_cast<const d2_t*>(Vb + (size_t)f * ld + col);
        }
    }
    double kreg[TL::K_LDS ? 1 : RB][TL::K_LDS ? 1 : KS][VEC];     // B operands in registers (rows owned by this wave alone)
    if (!TL::K_LDS) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int row = row_w0 + rb * RG + VEC * n16;
                const double* src = Kb + (size_t)(4 * ks + g4) * ld + (row < ld ? row : 0);
#pragma unroll
                for (int e = 0; e < VEC; e += 2) {
                    const d2_t x = *reinterpret_cast<const d2_t*>(src + e);
                    kreg[rb][ks][e] = x.v[0]; kreg[rb][ks][e + 1] = x.v[1];
                }
            }
    }
    double vreg[TL::V_LDS ? 1 : CB][TL::V_LDS ? 1 : KS];          // A operands in registers
    if (!TL::V_LDS) {
#pragma unroll
        for (int cg = 0; cg < CB; ++cg)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int col = col_w0 + cg * 16 + n16;
                vreg[cg][ks] = Vb[(size_t)(4 * ks + g4) * ld + (col < ld ? col : 0)];
            }
    }
    const T* Pb = Pin + (size_t)b * v.p_stride;
    vec_t p[RB][CB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cg = 0; cg < CB; ++cg)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row_w0 + rb * RG + VEC * n16;
                const int col = col_w0 + cg * 16 + g4 + 4 * r;
                p[rb][cg][r] = load_stream(Pb + (size_t)(col < L ? col : 0) * ld + (row < ld ? row : 0));
            }

    // ---- masks (factor f belongs to correction f >> 1), staging
    if (TL::K_LDS) {
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int e = tid + i * NT;
            const int f = e / (TL::WROWS / 2), i2 = (e % (TL::WROWS / 2)) * 2;
            const double mk = ((actmask >> (f >> 1)) & 1u) ? -1.0 : 0.0;
            d2_t x;
            x.v[0] = kst[i].v[0] * mk; x.v[1] = kst[i].v[1] * mk;
            *reinterpret_cast<d2_t*>(sK + (size_t)f * TL::KLD + i2) = x;
        }
    }
    if (TL::V_LDS) {
#pragma unroll
        for (int i = 0; i < VCH; ++i) {
            const int e = tid + i * NT;
            const int f = e / (TL::WCOLS / 2), i2 = (e % (TL::WCOLS / 2)) * 2;
            const double mk = ((actmask >> (f >> 1)) & 1u) ? 1.0 : 0.0;
            d2_t x;
            x.v[0] = vst[i].v[0] * mk; x.v[1] = vst[i].v[1] * mk;
            *reinterpret_cast<d2_t*>(sV + (size_t)f * TL::VLD + i2) = x;
        }
    }
    if (!TL::K_LDS) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const double mk = ((actmask >> ((4 * ks + g4) >> 1)) & 1u) ? -1.0 : 0.0;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int e = 0; e < VEC; ++e) kreg[rb][ks][e] *= mk;
        }
    }
    if (!TL::V_LDS) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const double mk = ((actmask >> ((4 * ks + g4) >> 1)) & 1u) ? 1.0 : 0.0;
#pragma unroll
            for (int cg = 0; cg < CB; ++cg) vreg[cg][ks] *= mk;
        }
    }
    if (TL::K_LDS || TL::V_LDS) lds_barrier();            // (LDS only: __syncthreads() would also wait for the tile)
    if (row_w0 >= ld || col_w0 >= L) return;

    T* Po = Pout + (size_t)b * v.p_stride;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        double kb[KS][VEC];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (TL::K_LDS) {
                const double* up = sK + (size_t)(4 * ks + g4) * TL::KLD + (wr * RB + rb) * RG + VEC * n16;
#pragma unroll
                for (int e = 0; e < VEC; e += 2) {
                    const d2_t x = *reinterpret_cast<const d2_t*>(up + e);
                    kb[ks][e] = x.v[0]; kb[ks][e + 1] = x.v[1];
                }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) kb[ks][e] = kreg[rb][ks][e];
            }
        }
#pragma unroll
        for (int cg = 0; cg < CB; ++cg) {
            rank_d4 acc[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[e][r] = (double)p[rb][cg][r].v[e];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const double va = TL::V_LDS ? sV[(size_t)(4 * ks + g4) * TL::VLD + (wc * CB + cg) * 16 + n16] : vreg[cg][ks];
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(va, kb[ks][e], acc[e], 0, 0, 0);
            }
            const int row = row_w0 + rb * RG + VEC * n16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                vec_t out;
#pragma unroll
                for (int e = 0; e < VEC; ++e) out.v[e] = (T)acc[e][r];
                const int col = col_w0 + cg * 16 + g4 + 4 * r;
                if (row < ld && col < L) store_stream(Po + (size_t)col * ld + row, out);
            }
        }
    }
}

} // namespace nuslam
