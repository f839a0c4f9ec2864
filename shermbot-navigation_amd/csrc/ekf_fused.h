// ekf_fused.h -- a known-id tick of ONE filter as ONE launch: predict || chain || strips || the pass over P.
//
// k_tick_front (ekf_tick.h) ends with the strips, and the rank-2m pass (k_tick_rank, ekf_rank.h) used to start behind it: 13 us
// of HBM traffic strictly after 35 us of a latency-bound chain, although the pass's TILE LOADS depend on nothing the chain
// produces -- only its MFMA operands do (K_s, V_s: slam_library.cpp:279 re-associated as P - K (H P)), and the strips already
// form those entry by entry.  Here the pass runs as workgroups of the same launch, behind the front's in blockIdx order (the
// producers are dispatched, and resident, before any consumer):
//   * a pass workgroup loads its tile of P into the MFMA accumulators at entry -- 32 MB of reads under the chain's prologue.
//     Tiles that hold rows / columns 0..2 wait for the predict workgroups first and read with agent-scope loads (the chain's
//     XCD has pre-predict lines of them in its L2);
//   * k-step ks (corrections 2 ks, 2 ks + 1) is applied when its operands have ARRIVED, and the operands say so themselves: the
//     strips store every K_s / V_s value as two 8-byte words { half of the double, this round's tag } (TickPublish::tagK / tagV;
//     agent-scope stores, no drain, no counter -- a drain per entry in front of a progress counter made the strips 5 us per
//     entry), and a pass wave loads the four factor rows it needs straight into registers in operand layout with agent-scope
//     loads until every word carries the tag.  One lane per workgroup first waits for the chain's announcement of the entry and for
//     a probe word of the strips, so that nobody polls data that cannot be there yet; the operands are staged in LDS once per
//     workgroup (every element fetched once: agent-scope loads are not served by the local L2); the round's last k-step is split
//     into its two corrections.  No fences;
//   * after the last k-step the tile is stored (nt) to the other P buffer: what is left behind the chain is the last entry's
//     strips, one k-step and the stores.
// Every element goes through k_tick_rank's own chain  acc = fma(V_f(col), -K_f(row), acc),  f ascending: the same bits as the
// two-launch form (tests/test_gpu_rank.py), whatever the tile shape.  Rounds the host cannot prove free of a first sighting
// keep the two launches (the exact chain may have to take them).  Every wait is bounded; an expired one is counted in
// `timeouts` (NUSLAM_E_SYNC, the handle is poisoned).
//
// Tile: <RB, CB> row / column groups per wave, 2 x 2 waves: 128 x 96 for fp64 -> 16 x 21 = 336 workgroups at N = 1000, which
// with the 72 of the front are resident together at two waves per SIMD (the chain's 225 registers set the kernel's allocation).
#pragma once

namespace nuslam {

template <typename T>
struct FusedTile {
    static constexpr int VEC = 16 / (int)sizeof(T);
    // (48 accumulator doubles per lane for either storage type: with the chain's 225 registers the kernel stays at two waves per SIMD)
    static constexpr int RB = VEC == 2 ? 2 : 1, CB = 3, WR = 2, WC = 2;
    static constexpr int RG = 16 * VEC;
    static constexpr int WROWS = WR * RB * RG, WCOLS = WC * CB * 16;
    static int tiles_r(int ld) { return (ld + WROWS - 1) / WROWS; }
    static int tiles_c(int L) { return (L + WCOLS - 1) / WCOLS; }
    __device__ static int tiles_r_dev(int ld) { return (ld + WROWS - 1) / WROWS; }
    __device__ static int tiles_c_dev(int L) { return (L + WCOLS - 1) / WCOLS; }
    static int blocks(int ld, int L) { return ((tiles_r(ld) + 7) / 8) * 8 * tiles_c(L); }   // (tile rows dealt over the XCDs, as k_tick_rank)
};

template <typename T>
__device__ inline void tick_pass_role(const int b, const int idx, View v, const int J, const TickStep* __restrict__ plan,
                                      const T* __restrict__ Pin, T* __restrict__ Pout, TickPublish pub, TickTagged tg, int* __restrict__ timeouts)
{
    typedef FusedTile<T> TL;
    typedef Pack16<T> vec_t;
    constexpr int RB = TL::RB, CB = TL::CB, WC = TL::WC, VEC = TL::VEC, RG = TL::RG, NF = kRankNF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int n16 = lane & 15, g4 = lane >> 4;
    const int ld = v.ld, L = v.L;
    const int tiles_r = (ld + TL::WROWS - 1) / TL::WROWS, tiles_c = (L + TL::WCOLS - 1) / TL::WCOLS;
    const int x = idx & 7, k = idx >> 3;
    const int tr = (k / tiles_c) * 8 + x, tc = k % tiles_c;
    if (tr >= tiles_r) return;
    const int row_w0 = tr * TL::WROWS + wr * RB * RG, col_w0 = tc * TL::WCOLS + wc * CB * 16;
#if defined(NUSLAM_FUSED_BISECT) && NUSLAM_FUSED_BISECT == 1
    return;                                                             // (measurement builds only: what the chain costs beside IDLE pass workgroups)
#endif
    const T* Pb = Pin + (size_t)b * v.p_stride;
    const TickStep* pl = plan + (size_t)b * kTickJ;

    // ---- the tile.  Rows / columns 1, 2 and the pose corner are rewritten by the predict workgroups of this launch
    const bool edge = pub.predict != 0 && (tr == 0 || tc == 0);         // (uniform)
    TL(11, idx == 100 && tid == 0);                                      // a pass workgroup: entry
    if (pub.predict) {
        // Nobody loads before the chain has gathered its 35 x 35 block (its 1225 scattered reads and the predict hand-offs took 2.8 us
        // longer with 32 MB of tile loads in flight beside them); the edge tiles also wait for the predict workgroups
        if (tid == 0) {
            int ok = 0;
            for (int it = 0; it < (1 << 18); ++it) {
                if (edge ? seq_reached(ld_agent(pub.flag + kPubWords * b + 2), pub.pbase)
                         : seq_reached(ld_agent(pub.flag + kPubWords * b + 1), pub.gbase)) { ok = 1; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (!ok) atomicAdd(timeouts, 1);
        }
        __syncthreads();
    }
    rank_d4 acc[RB][CB][VEC];
    if (edge) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int cg = 0; cg < CB; ++cg)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row_w0 + rb * RG + VEC * n16;
                    const int col = col_w0 + cg * 16 + g4 + 4 * r;
                    const T* src = Pb + (size_t)(col < L ? col : 0) * ld + (row < ld ? row : 0);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[rb][cg][e][r] = (double)ld_agent(src + e);
                }
    } else {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int cg = 0; cg < CB; ++cg)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row_w0 + rb * RG + VEC * n16;
                    const int col = col_w0 + cg * 16 + g4 + 4 * r;
                    const T* src = Pb + (size_t)(col < L ? col : 0) * ld + (row < ld ? row : 0);
                    const vec_t q = load_stream(src);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[rb][cg][e][r] = (double)q.v[e];
                }
    }

    // ---- the k-steps, each when its corrections' strips have arrived (cmask: bit 0 = correction 2 ks, bit 1 = 2 ks + 1 belongs to the
    // round).  (Splitting the round's last k-step into its two corrections, so that only one correction's operands are fetched behind
    // the chain's end, measured WORSE: a unit costs a pass workgroup ~3 us of dependent round trips whatever it holds.)
    const bool idle = row_w0 >= ld || col_w0 >= L;                      // (a wave beyond the matrix keeps the barriers company)
    const long long* tK = tg.tagK + (size_t)b * NF * ld * 2;
    const long long* tV = tg.tagV + (size_t)b * NF * ld * 2;
    const int tag = tg.tag;
    __shared__ double sK[4][TL::WROWS + 2];                               // -K_f at the workgroup's rows, masked (staged once per unit:
    __shared__ double sV[4][TL::WCOLS + 16];                               //  V_f at its columns           the waves share them)
    __shared__ int live_sh;                                             // the unit's live corrections (the plan's skip flags, read by the poller)
    int announced = 0;                                                  // (thread 0) plan entries the chain had announced when last asked
    bool lost = false;
    // the probe of a unit: ONE tagged word of its last live correction (this workgroup's first column of V): when it has arrived the
    // strips are at that entry, and only then are the operands loaded -- 1344 waves re-loading their operands while they waited cost
    // the chain's own store acknowledgements ~0.5 us per correction
    const long long* probe0 = tV + (size_t)(tc * TL::WCOLS) * 2;
    const int ksl = (J - 1) >> 1;                                       // the last k-step that holds a correction
#if defined(NUSLAM_FUSED_BISECT) && NUSLAM_FUSED_BISECT == 2
    const int units = 0;                                                // (measurement builds only: tile loads and stores, no k-steps)
#else
    const int units = ksl + 1;
#endif
#pragma unroll 1
    for (int u = 0; u < units; ++u) {
        const int ks = u;
        const int cmask = 2 * ks + 1 < J ? 3 : 1;
        const int need = 2 * ks + (cmask >> 1);                         // the last entry this unit reads
        if (tid == 0) {
            int ok = 0, lv = 0;
            const int* w = pub.flag + kPubWords * b;
            const bool tail = need + 2 >= J;                            // the round's last entries: worth asking for often
            for (int it = 0; it < (1 << 18); ++it) {
                // (A) the chain has announced the entry (asked again only when the last answer does not cover it)
                if (announced < need + 1) {
                    announced = (int)((unsigned)ld_agent(w) - (unsigned)pub.base);
                    if (announced < need + 1 || announced > 2 * kTickJ) {
                        announced = 0;
                        if (tail) __builtin_amdgcn_s_sleep(4);
                        else __builtin_amdgcn_s_sleep(48);              // (a correction takes the chain ~1.75 us)
                        continue;
                    }
                }
                // (B) ... and the strips have reached it: the probe word of the unit's last live correction carries the tag
                const int s0 = (cmask & 1) ? (int)(ld_agent(reinterpret_cast<const long long*>(&pl[2 * ks].skip)) & 0xffffffffll) : 1;
                const int s1 = (cmask & 2) ? (int)(ld_agent(reinterpret_cast<const long long*>(&pl[2 * ks + 1].skip)) & 0xffffffffll) : 1;
                lv = (s0 == 0 ? 1 : 0) | (s1 == 0 ? 2 : 0);
                if (lv == 0) { ok = 1; break; }
                if (tail) { ok = 1; break; }                            // (the round's end: the operands are asked for directly, below)
                const int cl = (lv & 2) ? 2 * ks + 1 : 2 * ks;
                const long long pw = ld_agent(probe0 + (size_t)(2 * cl + 1) * ld * 2);
                if ((int)(pw >> 32) == tag) { ok = 1; break; }
                if (tail) __builtin_amdgcn_s_sleep(2);
                else __builtin_amdgcn_s_sleep(24);
            }
            if (!ok) atomicAdd(timeouts, 1);
            live_sh = lv;
        }
        __syncthreads();
        TL(12, idx == 100 && tid == 0 && u == 0);                       // ... its first unit's entries have arrived, its last one's
        TL(13, idx == 100 && tid == 0 && u == units - 1);
        const int lv = live_sh;
        if (lv != 0 && !lost) {
            // the operands say themselves when they have arrived: every 8-byte word carries this round's tag (strip workgroups other than
            // the probed one may be a moment behind: rarely more than one trip).  Each element is fetched ONCE per workgroup.
            bool done = false;
            for (int it = 0; it < (1 << 16) && !done; ++it) {
                bool ok = true;
#pragma unroll
                for (int q = 0; q < (4 * TL::WROWS + 4 * TL::WCOLS + 255) / 256; ++q) {
                    const int e = tid + 256 * q;
                    if (e < 4 * TL::WROWS) {
                        const int r4 = e / TL::WROWS, i = e % TL::WROWS;
                        const bool on = ((lv >> (r4 >> 1)) & 1) != 0;
                        const int row = tr * TL::WROWS + i;
                        double x = 0.0;
                        if (on) ok = ld_tagged(tK + ((size_t)(4 * ks + r4) * ld + (row < ld ? row : 0)) * 2, tag, x) && ok;
                        sK[r4][i] = on ? -x : 0.0;
                    } else if (e < 4 * TL::WROWS + 4 * TL::WCOLS) {
                        const int e2 = e - 4 * TL::WROWS;
                        const int r4 = e2 / TL::WCOLS, i = e2 % TL::WCOLS;
                        const bool on = ((lv >> (r4 >> 1)) & 1) != 0;
                        const int col = tc * TL::WCOLS + i;
                        double x = 0.0;
                        if (on) ok = ld_tagged(tV + ((size_t)(4 * ks + r4) * ld + (col < L ? col : 0)) * 2, tag, x) && ok;   // (V exists for columns < L)
                        sV[r4][i] = on ? x : 0.0;
                    }
                }
                done = __syncthreads_and(ok ? 1 : 0) != 0;
                if (!done) __builtin_amdgcn_s_sleep(2);
            }
            if (!done) { lost = true; if (tid == 0) atomicAdd(timeouts, 1); }
            else {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    double kb[VEC];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) kb[e] = sK[g4][(wr * RB + rb) * RG + VEC * n16 + e];
#pragma unroll
                    for (int cg = 0; cg < CB; ++cg) {
                        const double va = sV[g4][(wc * CB + cg) * 16 + n16];
#pragma unroll
                        for (int e = 0; e < VEC; ++e)
                            acc[rb][cg][e] = __builtin_amdgcn_mfma_f64_16x16x4f64(va, kb[e], acc[rb][cg][e], 0, 0, 0);
                    }
                }
            }
        }
        lds_barrier();                                                  // (live_sh, sK, sV are rewritten by the next unit)
    }
    if (idle) return;
    T* Po = Pout + (size_t)b * v.p_stride;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cg = 0; cg < CB; ++cg) {
            const int row = row_w0 + rb * RG + VEC * n16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                vec_t out;
#pragma unroll
                for (int e = 0; e < VEC; ++e) out.v[e] = (T)acc[rb][cg][e][r];
                const int col = col_w0 + cg * 16 + g4 + 4 * r;
                if (row < ld && col < L) store_stream(Po + (size_t)col * ld + row, out);
            }
        }
    TL(14, idx == 100 && tid == 0);                                      // ... its stores issued
    TL(15, idx == (int)gridDim.x - 80 && tid == 0);
}

// grid.x = 1 (the chain) + n_pred (predict workgroups) + n_strip (strip workgroups) + FusedTile<T>::blocks (pass workgroups);
// grid.y = 1: ONE filter (with several, filter 1's producers would be dispatched behind filter 0's consumers)
template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_tick_fused(View v, TickObs o, int total_landmarks, T* __restrict__ P, T* __restrict__ Pout,
                                                    TickStep* __restrict__ plan, double* __restrict__ Kbuf,
                                                    double* __restrict__ Rbuf, double* __restrict__ Vbuf,
                                                    TickPublish pub, TickTagged tg, int n_pred, int n_strip, int J, int* __restrict__ timeouts)
{
    // (J == o.J, as an argument of its own: reading a field of `o` in the kernel body itself made the compiler keep a private copy of
    // the whole struct in scratch memory for the chain and the strips, whose loops then wait on scratch loads)
    const int b = blockIdx.y;
    const int x = blockIdx.x;
    if (x == 0) tick_chain<T, false, true>(b, v, o, total_landmarks, P, plan, TickCarry{}, nullptr, nullptr, pub);
    else if (x <= n_pred) {
        TL(4, x == 1 && threadIdx.x == 0);
        tick_predict_role<T>(b, x - 1, v, pub, P, timeouts);
        TL(5, x == 1 && threadIdx.x == 0);
    } else if (x <= n_pred + n_strip) {
        TL(6, x == n_pred + 1 + n_strip / 2 && threadIdx.x == 0);          // strips: a middle workgroup and the last one, entry / exit
        TL(8, x == n_pred + n_strip && threadIdx.x == 0);
        tick_panels_stream<T, true>(b, x - 1 - n_pred, v, o, P, plan, Kbuf, Rbuf, Vbuf, pub, timeouts, nullptr, nullptr, nullptr, nullptr, tg);
        TL(7, x == n_pred + 1 + n_strip / 2 && threadIdx.x == 0);
        TL(9, x == n_pred + n_strip && threadIdx.x == 0);
    } else tick_pass_role<T>(b, x - 1 - n_pred - n_strip, v, J, plan, P, Pout, pub, tg, timeouts);
}

// ------------------------------------------------------------------------------------------------ the ticks of a resident trace in ONE launch
// nuslam_batch_run on one filter with known ids: k_tick_fused's four roles, each LOOPING over the run's ticks, and the covariance
// RESIDENT in the pass workgroups' accumulators in between (N = 1000's P is 32 MB, a quarter of the chip's vector registers).  A tick
// then ends not with 32 MB of stores but with what the NEXT tick's chain and strips gather: a pass workgroup stores, after its last
// k-step, only the entries of its tile whose row or column is in the next tick's index set (~1 MB, at their natural addresses in the
// other P buffer: the gathers stay as they are); the tiles that hold rows / columns 0..2 -- the predict role rewrites those in place
// every tick -- store whole and reload.  The last tick of the launch stores every tile: P is complete in memory when the run ends, and
// only then.  Same sums in the same order as a launch per tick: the same bits (tests/test_gpu_run.py).
//
// Hand-offs across ticks: every strip workgroup (state stored) and every live pass workgroup (exports stored) adds one to a count
// at the end of a tick; the next tick's chain waits for it before it gathers, and everything else follows from the chain as in one
// tick (predict behind the gather, strips and edge tiles behind the predict, k-steps behind the tagged operands).  What crosses from
// one workgroup to another inside the launch -- exports, edge tiles, the state -- is stored and loaded at agent scope (the reader's
// XCD may hold the previous tick's line).  Tags, announcement bases and the two counters advance per tick; K / V words and plan
// entries are rewritten only after their last reader has counted itself done.
struct RunArg {
    int ticks;             // ticks of this launch
    int m;                 // markers per tick in the resident trace (TickObs::off advances by it)
    int* done;             // TickRun::done: a word per strip workgroup, then a word per pass workgroup
    int* done_all;         // TickRun::done_all: a word per pass workgroup
    int done_base;         // the running tick number before the launch: tick t of the launch is number done_base + t + 1
    int n_strip, n_pass;   // the words
    const int* ids;        // the trace's ids (TickObs::ids / stride / off of the launch's first tick: as arguments of their own -- a field of the
    long long ids_stride, ids_off;   // by-value TickObs read in a role makes the compiler keep a private copy of the struct in scratch)
};

// 16 bytes at agent scope in one store (the edge tiles of k_run_fused: 32 KB per workgroup and tick, read by other workgroups of the launch)
template <typename V>
__device__ inline void st_agent16(void* p, const V& x)
{
    static_assert(sizeof(V) == 16, "16 bytes");
    typedef int i4 __attribute__((ext_vector_type(4)));
    i4 w;
    __builtin_memcpy(&w, &x, 16);
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(w) : "memory");
}

template <typename T>
__device__ inline void run_pass_role(const int b, const int idx, View v, const int J, const TickStep* __restrict__ plan,
                                     T* __restrict__ P0, T* __restrict__ P1, TickPublish pub0, TickTagged tg0, const int n_pred,
                                     int* __restrict__ timeouts, RunArg ra)
{
    typedef FusedTile<T> TL;
    typedef Pack16<T> vec_t;
    constexpr int RB = TL::RB, CB = TL::CB, WC = TL::WC, VEC = TL::VEC, RG = TL::RG, NF = kRankNF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int n16 = lane & 15, g4 = lane >> 4;
    const int ld0 = v.ld, L = v.L;
    const int ld = ld0;
    const int tiles_r = (ld + TL::WROWS - 1) / TL::WROWS, tiles_c = (L + TL::WCOLS - 1) / TL::WCOLS;
    const int x = idx & 7, k = idx >> 3;
    const int tr = (k / tiles_c) * 8 + x, tc = k % tiles_c;
    if (tr >= tiles_r) return;                                          // (not a live pass workgroup: not counted in RunArg::per_tick)
    const int row_g0 = tr * TL::WROWS, col_g0 = tc * TL::WCOLS;
    const int row_w0 = row_g0 + wr * RB * RG, col_w0 = col_g0 + wc * CB * 16;
    const TickStep* pl = plan + (size_t)b * kTickJ;
    const bool edge = tr == 0 || tc == 0;                               // holds rows / columns 0..2: rewritten by the predict role every tick
    const bool idle = row_w0 >= ld || col_w0 >= L;                      // (a wave beyond the matrix keeps the barriers company)
    const long long* tK = tg0.tagK + (size_t)b * NF * ld * 2;
    const long long* tV = tg0.tagV + (size_t)b * NF * ld * 2;
    __shared__ double sK[4][TL::WROWS + 2];
    __shared__ double sV[4][TL::WCOLS + 16];
    __shared__ int live_sh;
    __shared__ unsigned char rowf[TL::WROWS], colf[TL::WCOLS];          // the next tick's index set inside this tile
    __shared__ int skipc[kTickJ];                                       // (thread 0's) the plan entries' skip flags, as far as announced
    const long long* probe0 = tV + (size_t)(tc * TL::WCOLS) * 2;
    const int ksl = (J - 1) >> 1;
    const int units = ksl + 1;
    bool lost = false;
    rank_d4 acc[RB][CB][VEC];

#pragma unroll 1
    for (int t = 0; t < ra.ticks; ++t) {
        int ld = ld0;
        asm volatile("" : "+s"(ld));                                    // (as in k_run_fused: nothing of a tick's address arithmetic is hoisted out of the loop)
        const T* Pb = ((t & 1) ? P1 : P0) + (size_t)b * v.p_stride;
        T* Po = ((t & 1) ? P0 : P1) + (size_t)b * v.p_stride;
        const int base = pub0.base + t * 2 * kTickJ, gbase = pub0.gbase + t, pbase = pub0.pbase + t * n_pred;
        const int tag = tg0.tag + t;
        // ---- the tile: from memory in the launch's first tick; the edge tiles every tick, behind the predict role
        if (t == 0 || edge) {
            if (tid == 0) {
                int ok = 0;
                for (int it = 0; it < (1 << 18); ++it) {
                    if (edge ? seq_reached(ld_agent(pub0.flag + kPubWords * b + 2), pbase)
                             : seq_reached(ld_agent(pub0.flag + kPubWords * b + 1), gbase)) { ok = 1; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
                if (!ok) atomicAdd(timeouts, 1);
            }
            __syncthreads();
            if (edge) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int cg = 0; cg < CB; ++cg)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = row_w0 + rb * RG + VEC * n16;
                            const int col = col_w0 + cg * 16 + g4 + 4 * r;
                            const T* src = Pb + (size_t)(col < L ? col : 0) * ld + (row < ld ? row : 0);
#pragma unroll
                            for (int e = 0; e < VEC; ++e) acc[rb][cg][e][r] = (double)ld_agent(src + e);
                        }
            } else {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int cg = 0; cg < CB; ++cg)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = row_w0 + rb * RG + VEC * n16;
                            const int col = col_w0 + cg * 16 + g4 + 4 * r;
                            const T* src = Pb + (size_t)(col < L ? col : 0) * ld + (row < ld ? row : 0);
                            const vec_t q = load_stream(src);
#pragma unroll
                            for (int e = 0; e < VEC; ++e) acc[rb][cg][e][r] = (double)q.v[e];
                        }
            }
        }
        // ---- what this workgroup will export behind the tick's last k-step: the entries of its tile in a row or a column of the NEXT
        // tick's index set (the chain's rule: an id outside the map reads index 3).  Found now, under the chain's loop.
        const bool last = t + 1 == ra.ticks;
        unsigned rmask = 0u, cmask = 0u;                                 // this lane's rows (rb, e) / columns (cg, r) to export
        if (!last) {                                                    // (uniform)
            // (positions 0..2 -- the pose -- are in every index set: the chain gathers those rows / columns too)
            if (tid < TL::WROWS) rowf[tid] = (row_g0 + tid < 3) ? 1 : 0;
            if (tid < TL::WCOLS) colf[tid] = (col_g0 + tid < 3) ? 1 : 0;
            __syncthreads();
            if (tid < kTickJ) {
                int id = 0;
                if (tid < J) id = ra.ids[b * ra.ids_stride + ra.ids_off + (long long)(t + 1) * ra.m + tid];
                const int c = (id >= 1 && id <= v.n) ? 3 + 2 * (id - 1) : 3;
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const int i = c + d;
                    if (i >= row_g0 && i < row_g0 + TL::WROWS) rowf[i - row_g0] = 1;
                    if (i >= col_g0 && i < col_g0 + TL::WCOLS) colf[i - col_g0] = 1;
                }
            }
            __syncthreads();
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (rowf[(wr * RB + rb) * RG + VEC * n16 + e]) rmask |= 1u << (rb * VEC + e);
#pragma unroll
            for (int cg = 0; cg < CB; ++cg)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (colf[(wc * CB + cg) * 16 + g4 + 4 * r]) cmask |= 1u << (cg * 4 + r);
        }
        // ---- the k-steps (tick_pass_role's, with this tick's tag and announcement base)
        int announced = 0, cached = 0;                                   // (thread 0) entries announced / entries whose skip flag is in skipc
        const bool stamp = idx == 100 && tid == 0 && t + 2 == ra.ticks;  // (debug builds: the timeline of the launch's second-to-last tick)
#pragma unroll 1
        for (int u = 0; u < units; ++u) {
            const int ks = u;
            const int cmask = 2 * ks + 1 < J ? 3 : 1;
            const int need = 2 * ks + (cmask >> 1);
            if (tid == 0) {
                int ok = 0, lv = 0;
                const int* w = pub0.flag + kPubWords * b;
                const bool tail = need + 2 >= J;
                for (int it = 0; it < (1 << 18); ++it) {
                    if (announced < need + 1) {
                        announced = (int)((unsigned)ld_agent(w) - (unsigned)base);
                        if (announced < need + 1 || announced > 2 * kTickJ) {
                            announced = 0;
                            if (tail) __builtin_amdgcn_s_sleep(4);
                            else __builtin_amdgcn_s_sleep(48);
                            continue;
                        }
                    }
                    // the skip flags of EVERY entry announced so far, asked for together (a workgroup that is behind -- an edge tile starts
                    // its tick behind the predict role -- then pays one round trip for several units, not one per unit)
                    const int upto = announced < J ? announced : J;
                    if (cached < upto) {
                        long long fw[kTickJ];
#pragma unroll
                        for (int e = 0; e < kTickJ; ++e)
                            fw[e] = (e >= cached && e < upto) ? ld_agent(reinterpret_cast<const long long*>(&pl[e].skip)) : 0ll;
#pragma unroll
                        for (int e = 0; e < kTickJ; ++e)
                            if (e >= cached && e < upto) skipc[e] = (int)(fw[e] & 0xffffffffll);
                        cached = upto;
                    }
                    const int s0 = (cmask & 1) ? skipc[2 * ks] : 1;
                    const int s1 = (cmask & 2) ? skipc[2 * ks + 1] : 1;
                    lv = (s0 == 0 ? 1 : 0) | (s1 == 0 ? 2 : 0);
                    if (lv == 0) { ok = 1; break; }
                    if (tail || announced >= need + 3) { ok = 1; break; }   // (the strips are past an entry two announcements old)
                    const int cl = (lv & 2) ? 2 * ks + 1 : 2 * ks;
                    const long long pw = ld_agent(probe0 + (size_t)(2 * cl + 1) * ld * 2);
                    if ((int)(pw >> 32) == tag) { ok = 1; break; }
                    if (tail) __builtin_amdgcn_s_sleep(2);
                    else __builtin_amdgcn_s_sleep(24);
                }
                if (!ok) atomicAdd(timeouts, 1);
                live_sh = lv;
            }
            __syncthreads();
            TL(17, stamp && u == units - 1);                            // run: the last unit's entries are announced
            const int lv = live_sh;
            if (lv != 0 && !lost) {
                bool done = false;
                for (int it = 0; it < (1 << 16) && !done; ++it) {
                    bool ok = true;
#pragma unroll
                    for (int q = 0; q < (4 * TL::WROWS + 4 * TL::WCOLS + 255) / 256; ++q) {
                        const int e = tid + 256 * q;
                        if (e < 4 * TL::WROWS) {
                            const int r4 = e / TL::WROWS, i = e % TL::WROWS;
                            const bool on = ((lv >> (r4 >> 1)) & 1) != 0;
                            const int row = row_g0 + i;
                            double xv = 0.0;
                            if (on) ok = ld_tagged(tK + ((size_t)(4 * ks + r4) * ld + (row < ld ? row : 0)) * 2, tag, xv) && ok;
                            sK[r4][i] = on ? -xv : 0.0;
                        } else if (e < 4 * TL::WROWS + 4 * TL::WCOLS) {
                            const int e2 = e - 4 * TL::WROWS;
                            const int r4 = e2 / TL::WCOLS, i = e2 % TL::WCOLS;
                            const bool on = ((lv >> (r4 >> 1)) & 1) != 0;
                            const int col = col_g0 + i;
                            double xv = 0.0;
                            if (on) ok = ld_tagged(tV + ((size_t)(4 * ks + r4) * ld + (col < L ? col : 0)) * 2, tag, xv) && ok;
                            sV[r4][i] = on ? xv : 0.0;
                        }
                    }
                    done = __syncthreads_and(ok ? 1 : 0) != 0;
                    if (!done) __builtin_amdgcn_s_sleep(2);
                }
                if (!done) { lost = true; if (tid == 0) atomicAdd(timeouts, 1); }
                else {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        double kb[VEC];
#pragma unroll
                        for (int e = 0; e < VEC; ++e) kb[e] = sK[g4][(wr * RB + rb) * RG + VEC * n16 + e];
#pragma unroll
                        for (int cg = 0; cg < CB; ++cg) {
                            const double va = sV[g4][(wc * CB + cg) * 16 + n16];
#pragma unroll
                            for (int e = 0; e < VEC; ++e)
                                acc[rb][cg][e] = __builtin_amdgcn_mfma_f64_16x16x4f64(va, kb[e], acc[rb][cg][e], 0, 0, 0);
                        }
                    }
                }
            }
            lds_barrier();
        }
        // ---- what the tick leaves in memory.  The accumulators hold the tile as stored: rounded to the storage type, as a launch
        // per tick would reload it
        if constexpr (sizeof(T) != sizeof(double)) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int cg = 0; cg < CB; ++cg)
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[rb][cg][e][r] = (double)(T)acc[rb][cg][e][r];
        }
        TL(18, stamp);                                                  // run: k-steps done
        TLMAX(edge ? 26 : 27, tid == 0 && t + 2 == ra.ticks);          // (debug timeline: the last edge / interior workgroup through its k-steps)
        if (last) {
            if (!idle) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int cg = 0; cg < CB; ++cg) {
                        const int row = row_w0 + rb * RG + VEC * n16;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            vec_t out;
#pragma unroll
                            for (int e = 0; e < VEC; ++e) out.v[e] = (T)acc[rb][cg][e][r];
                            const int col = col_w0 + cg * 16 + g4 + 4 * r;
                            if (row < ld && col < L) store_stream(Po + (size_t)col * ld + row, out);
                        }
                    }
            }
        } else {
            // first what the next tick's CHAIN gathers -- row and column both in the index set: a handful of entries per workgroup --, and
            // the word that says so; the rest behind it
            if (!idle && __any(rmask != 0u && cmask != 0u)) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int cg = 0; cg < CB; ++cg)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int col = col_w0 + cg * 16 + g4 + 4 * r;
                            const bool cf = ((cmask >> (cg * 4 + r)) & 1u) != 0u;
#pragma unroll
                            for (int e = 0; e < VEC; ++e) {
                                const int row = row_w0 + rb * RG + VEC * n16 + e;
                                const bool rf = cf && ((rmask >> (rb * VEC + e)) & 1u) != 0u;
                                if (rf && row < ld && col < L) st_agent(Po + (size_t)col * ld + row, (T)acc[rb][cg][e][r]);
                            }
                        }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) st_agent(ra.done + ra.n_strip + idx, ra.done_base + t + 1);
            TLMAX(edge ? 24 : 22, tid == 0 && t + 2 == ra.ticks);   // (debug timeline: the last interior / edge workgroup to say so)
            if (edge) {
                if (!idle) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int cg = 0; cg < CB; ++cg) {
                            const int row = row_w0 + rb * RG + VEC * n16;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                vec_t out;
#pragma unroll
                                for (int e = 0; e < VEC; ++e) out.v[e] = (T)acc[rb][cg][e][r];
                                const int col = col_w0 + cg * 16 + g4 + 4 * r;
                                if (row < ld && col < L) st_agent16(Po + (size_t)col * ld + row, out);
                            }
                        }
                }
            } else if (__any((rmask | cmask) != 0u)) {                  // (most waves of most tiles hold nothing of the next index set)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int cg = 0; cg < CB; ++cg)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int col = col_w0 + cg * 16 + g4 + 4 * r;
                            const bool cf = ((cmask >> (cg * 4 + r)) & 1u) != 0u;
#pragma unroll
                            for (int e = 0; e < VEC; ++e) {
                                const int row = row_w0 + rb * RG + VEC * n16 + e;
                                const bool rf = cf || ((rmask >> (rb * VEC + e)) & 1u) != 0u;
                                if (rf && row < ld && col < L) st_agent(Po + (size_t)col * ld + row, (T)acc[rb][cg][e][r]);
                            }
                        }
            }
            // ... and this workgroup is done with the tick
            TL(19, stamp);                                              // run: exports issued
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            TL(20, stamp);                                              // run: exports acknowledged
            if (tid == 0) st_agent(ra.done_all + idx, ra.done_base + t + 1);
            TLMAX(25, tid == 0 && t + 2 == ra.ticks);               // (debug timeline: the last workgroup with everything stored)
            TL(21, stamp);                                              // run: counted
        }
    }
}

// grid as k_tick_fused.  P0 holds the covariance when the launch begins; it is complete in P1 (odd number of ticks) or P0 (even) when it ends
template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_run_fused(View v, TickObs o, int total_landmarks, T* __restrict__ P0, T* __restrict__ P1,
                                                    TickStep* __restrict__ plan, double* __restrict__ Kbuf,
                                                    double* __restrict__ Rbuf, double* __restrict__ Vbuf,
                                                    TickPublish pub, TickTagged tg, int n_pred, int n_strip, int J, int* __restrict__ timeouts, RunArg ra)
{
    const int b = blockIdx.y;
    const int x = blockIdx.x;
    if (x > n_pred + n_strip) {
        run_pass_role<T>(b, x - 1 - n_pred - n_strip, v, J, plan, P0, P1, pub, tg, n_pred, timeouts, ra);
        return;
    }
    // (a loop per role: one loop around the three roles' text made the register allocator carry every role's address arithmetic through
    // all of them)
    auto at_tick = [&](int t, View& vt, TickPublish& pt, TickTagged& tt, TickRun& rn) {
        vt = v;
        // (what does not change from tick to tick must not look so: hoisted out of the loop, a role's address arithmetic -- all of it a
        // function of ld -- is kept live across the whole launch and spills)
        asm volatile("" : "+s"(vt.ld));
        if (t & 1) {
            vt.s_in = v.s_out; vt.s_out = const_cast<double*>(v.s_in);
            vt.c_in = v.c_out; vt.c_out = const_cast<int*>(v.c_in);
        }
        pt = pub;
        pt.base = pub.base + t * 2 * kTickJ; pt.gbase = pub.gbase + t; pt.pbase = pub.pbase + t * n_pred;
        pt.tw.off = pub.tw.off + 2 * t;
        tt = tg;
        tt.tag = tg.tag + t;
        rn.done = ra.done; rn.done_all = ra.done_all; rn.stamp = ra.done_base + t + 1; rn.first = t == 0 ? 1 : 0; rn.toff = (long long)t * ra.m; rn.timeouts = timeouts;
        rn.n_strip = ra.n_strip; rn.n_pass = ra.n_pass; rn.stamp_dbg_second_last = t + 2 == ra.ticks ? 1 : 0;
        rn.tiles_r = FusedTile<T>::tiles_r_dev(v.ld); rn.tiles_c = FusedTile<T>::tiles_c_dev(v.L);
    };
    View vt; TickPublish pt; TickTagged tt; TickRun rn;
    if (x == 0) {
#pragma unroll 1
        for (int t = 0; t < ra.ticks; ++t) {
            at_tick(t, vt, pt, tt, rn);
            tick_chain<T, false, true, true>(b, vt, o, total_landmarks, (t & 1) ? P1 : P0, plan, TickCarry{}, nullptr, nullptr, pt, rn);
            __syncthreads();                                            // (the role's LDS is rewritten by the next tick)
        }
    } else if (x <= n_pred) {
#pragma unroll 1
        for (int t = 0; t < ra.ticks; ++t) {
            at_tick(t, vt, pt, tt, rn);
            tick_predict_role<T, true>(b, x - 1, vt, pt, (t & 1) ? P1 : P0, timeouts, rn);
            __syncthreads();
        }
    } else {
#pragma unroll 1
        for (int t = 0; t < ra.ticks; ++t) {
            at_tick(t, vt, pt, tt, rn);
            tick_panels_stream<T, true, true>(b, x - 1 - n_pred, vt, o, (t & 1) ? P1 : P0, plan, Kbuf, Rbuf, Vbuf, pt, timeouts, nullptr, nullptr, nullptr, nullptr, tt, rn);
            __syncthreads();
        }
    }
}

} // namespace nuslam
