// ekf_tick.h -- one tick's corrections as THREE kernels and ONE pass over the covariance.
//
// The m corrections of a tick (slam.cpp:279-318: update() per marker, each relinearised at the state the previous one
// left) are sequential, but what makes them sequential is small.  Correction s touches every element of P through
//
//     P_s(i,j) = sum_k M_s(i,k) P_{s-1}(k,j),   k ascending over {0,1,2} U {i} U {c_s, c_s+1},   M_s = I - K_s H_s
//
// (slam_library.cpp:279 with the exactly-zero terms dropped -- the sweep formula of ekf_update.h), and the only inputs
// of that formula that are not the element itself are, per ROW i, the gain K_s(i,:) (two numbers) and, per COLUMN j,
// the five prior entries R_s(q,j) = P_{s-1}(set_s[q], j).  Those are O(len) numbers per correction, and they obey the
// same recurrence restricted to a PANEL: with U = {0,1,2} U {c_t, c_t+1 : t = 1..m} (3 + 2m indices),
//
//     rows    RP[p][j] = P(U[p], j)      columns CP[i][p] = P(i, U[p])      block BK[p][p'] = P(U[p], U[p'])
//
// every entry of RP / CP / BK after correction s is the sweep formula applied to entries of RP / CP / BK after
// correction s-1.  So:
//
//   k_tick_chain   one workgroup per filter: the serial part.  Carries BK (35 x 35) and the 35 state entries at U
//                  through the m corrections: decision chain (slam.cpp:295-316), initializeLandmark (:255-261), z_hat
//                  (:150-160), H (:162-186), S, S^-1 (:270), K and M at the rows of U, the state at U -- and writes a
//                  PLAN per correction: H, S^-1, the innovation, M(U[p],:), P_{s-1}(set_s, U[p]).   Latency-bound
//                  (transcendentals); nothing in it is O(len).
//   k_tick_panels  one quad of lanes per (state index, role), no serial dependency left: replays the plan (staged in
//                  LDS) on its column of RP and its row of CP and emits, per correction, R_s(:, j) (5 x len) and
//                  K_s(i,:) (len x 2), and the new state vector.
//   k_tick_apply   THE pass over P: every tile is read once, carried through all m corrections in registers with the
//                  K / R strips staged in LDS, and written once -- 2 len^2 w bytes per TICK instead of per correction
//                  (or per pair).  k_tick_apply_units: the same for one large fp64 filter, a wave's columns in two
//                  units so that loads and stores hide behind the arithmetic.
//   (overlapped runs: k_tick_prep, tick_carry inside the next tick's k_tick_chain, k_tick_wait / k_tick_signal)
//
// Every floating-point operation on every element is the one k_update performs, in the same order, with the same
// rounding to the storage type after each correction; only where it is executed has changed.  The result is therefore
// bit-identical to m launches of k_update (tests/test_gpu_tick.py) and hence to the oracle's update().
//
// Round 3, on top of that:
//   k_tick_rank    (ekf_rank.h) the DEFAULT pass over P: the same product re-associated as a rank-2m update on the matrix
//                  cores, P -= [K_1 .. K_m][V_1; ..; V_m] with V_s = H_s R_s formed beside R_s (hp_entry) -- 2 FMAs per
//                  element and correction instead of 7, a streaming kernel; agrees with the chain above to rounding, not bit
//                  for bit.  Rounds that cancel an INT_MAX diagonal stay with k_tick_apply (round_flags; the plan's `init`).
//   k_tick_front   one filter: predict, chain and strips as ONE launch.  The chain (workgroup 0) gathers its block from the
//                  covariance before predict and applies predict to it itself (predict_block); the predict workgroups
//                  rewrite rows / columns 1, 2 of P; the strip workgroups follow the chain entry by entry as it announces
//                  each plan entry (TickPublish: agent-scope stores and loads, no fences) instead of starting when it ends,
//                  fetching entry s+1 while they apply entry s.  Same arithmetic as k_predict + k_tick_chain + k_tick_panels:
//                  same bits.  (k_tick_strips / k_tick_chain_pub: the same hand-off between two launches on two streams,
//                  streamed overlapped runs.)
//   the chain      ONE barrier-separated phase per correction (it had three): every wave forms for itself what it needs of what
//                  the previous correction leaves (gain rows, state), the head's wave-uniform work runs on lanes side by side,
//                  the round's decisions are taken once, one lane per marker (tick_chain).
//   rank-form strips  in rounds the rank-2m pass applies, the strips carry their panels the way the pass will apply the round to P:
//                  P(U[p], t) -= K_s(U[p], :) V_s(:, t), two FMAs per entry and correction instead of seven.  The plan entry's
//                  HEAD holds what that takes (KV rows); k_tick_panels decides the form per filter from the round's flags (or is
//                  told by the host that every filter qualifies: LDS for the heads only), k_tick_front's strips take the form
//                  the host can prove -- so a filter's bits do not depend on the launch form it runs in.
#pragma once
#include <type_traits>

namespace nuslam {

constexpr int kTickJ = 16;                 // corrections per round (a tick with more markers runs several rounds)
constexpr int kTickNU = 3 + 2 * kTickJ;    // panel indices
constexpr int kTickDump = 512;             // doubles behind each strip buffer: where k_tick_panels' lanes that own nothing store

// the markers of one round, for every filter
struct TickObs {
    const double* a;       // marker x (cartesian != 0) or range, [b * stride + off + s]; null -> a0[s]
    const double* b;       // marker y or bearing
    const int* ids;        // known 1-based ids, same indexing; null -> id0[s]
    long long stride, off;
    double a0[kTickJ], b0[kTickJ];
    int id0[kTickJ];
    int cartesian;
    int log_slot0;         // id_log index of marker 0 of this round, or -1
    int J;                 // markers in this round, 1..kTickJ
    int forced;            // != 0: the CALLER has taken the decisions of slam.cpp:295-316 (the class driven call by call: every marker
                           // here is an update() that was really called, MODE_FORCE): no skip / break / `seen` bookkeeping on the device
    unsigned init_mask;    // forced: bit s = initializeLandmark(z_s, id_s) was called in front of update(z_s, id_s)
};

// what one correction leaves for k_tick_panels and k_tick_apply (per filter, per marker)
struct alignas(16) TickStep {
    int skip;              // the marker changes nothing in P (gray zone, break, bad id, singular S)
    int init;              // bit 0: initializeLandmark ran: state rows c, c+1 become lxy before the correction (or instead of it);
                           // bit 1: the landmark's covariance still carried the INT_MAX diagonal of slam_library.cpp:30 (with ids
                           // in order the same corrections; a known id BELOW an earlier one is never initialised by the
                           // caller's chain, slam.cpp:295, yet its first update cancels INT_MAX all the same)
    int c;                 // first state index of the landmark
    int id;                // resolved id
    double Hc[10], Sinv[4], dz[2], lxy[2];
    double heading, pad_;          // the wrapped heading after the correction (slam_library.cpp:276)
    // ---- the PREFIX ends here (kPlanPrefixWords).  Strips in the exact chain's form stage / fetch prefix + MP + BR (kPlanExactWords),
    // strips in rank form prefix + KV, packed as kPlanHeadWords words (plan_kv below)
    double MP[kTickNU][8];         // M_s(U[p], set_s[0..4]), before-flag, after-flag of row U[p]
    double BR[kTickNU][8];         // BR[p][q] = P_{s-1}(set_s[q], U[p]), q = 0..4
    double KV[kTickNU][4];         // K_s(U[p], 0..1), V_s(0..1, U[p]): what the strips need of the index set when they carry their panels in
                                   // RANK form (rounds the rank-2m pass applies: P -= K_s V_s, two FMAs per entry instead of seven)
};
constexpr int kPlanPrefixWords = 2 + 18 + 2;                             // 8-byte words: the four ints, Hc .. lxy, heading + pad
constexpr int kPlanExactWords = kPlanPrefixWords + kTickNU * 16;        // ... + MP + BR
constexpr int kPlanHeadWords = kPlanPrefixWords + kTickNU * 4;          // the packed image of a rank-form entry: prefix, then KV
static_assert(kPlanPrefixWords % 2 == 0 && kPlanHeadWords % 2 == 0 && kPlanExactWords % 2 == 0, "whole 16-byte pieces");
static_assert(sizeof(TickStep) == 8 * (kPlanExactWords + kTickNU * 4), "TickStep layout");
// the KV rows of a PACKED rank-form entry (LDS images: ps points at the prefix, the rows follow it)
__device__ inline const double (*plan_kv(const TickStep* ps))[4]
{
    return reinterpret_cast<const double (*)[4]>(reinterpret_cast<const double*>(ps) + kPlanPrefixWords);
}

// What an overlapped run's chain starts from instead of the covariance (FUSED): the strips of the PREVIOUS tick at this
// tick's index set, dropped into compact arrays by k_tick_panels, the 35 x 35 block k_tick_prep gathered from the
// covariance the previous pass reads, that tick's plan and control words, this tick's twist.
// dynamic LDS of a FUSED chain (tick_carry): the previous tick's R (5), K (2), V (2) strips at this tick's index set, the plan's scalars, the state
constexpr int kTickCarryLds = (kTickJ * 9 * (kTickNU + 1) + kTickJ * 12 + kTickNU + 1) * (int)sizeof(double);
struct TickCarry {
    const double* blk;         // [B][NU][NU]   P(U'[p], U'[q]) before the previous tick's corrections
    const double* KU;          // [B][J][2][NU] K_s at U'
    const double* RU;          // [B][J][5][NU] R_s at U'
    const double* SU;          // [B][NU]       state at U' after the previous tick
    const TickStep* plan_prev; // the previous tick's plan
    const int* wait_cnt;       // bumped by the kernel behind the previous tick's k_tick_panels
    int wait_target;
    int* timeouts;
    int Jt;                    // markers of the previous tick's round
    int rank;                  // != 0: that round's pass is the rank-2m one unless its plan holds a first sighting (round_flags)
    TwistArg tw;               // this tick's twist
};

// Agent-scope (sc1) stores and loads: what one workgroup hands to workgroups on other CUs / XCDs INSIDE a launch travels as
// these on both sides (MI355X_MICROARCH.md, "Valid forms": 8-byte agent atomics both sides, the signal after every storing
// wave's vmcnt(0) and a workgroup barrier) -- no fences, whose L2 write-back / L1 invalidate cost ~1.7 us each.
#ifdef NUSLAM_DA_EXP_PLAIN
__device__ inline void st_agent(double* p, double x) { *p = x; }
#else
__device__ inline void st_agent(double* p, double x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
__device__ inline void st_agent(long long* p, long long x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_agent(int* p, int x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_agent(float* p, float x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline float ld_agent(const float* p)
{
    return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline double ld_agent(const double* p)
{
    return __hip_atomic_load(const_cast<double*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline long long ld_agent(const long long* p)
{
    return __hip_atomic_load(const_cast<long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline int ld_agent(const int* p)
{
    return __hip_atomic_load(const_cast<int*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sequence words of the in-launch hand-offs count up for the life of a handle (a run of an hour wraps them): "has the word
// reached the target" is decided on the wrapped difference, in unsigned arithmetic.
__device__ inline bool seq_reached(int word, int target) { return (int)((unsigned)word - (unsigned)target) >= 0; }

// Workgroup barrier for hand-offs through LDS only: waits for this wave's LDS traffic, not for its global stores
// (__syncthreads() also drains vmcnt, i.e. waits ~1 us for the acknowledgement of every plan store in flight).
__device__ inline void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// (H_s P_{s-1})(r, j) from the five prior rows at column j: ascending-q FMA chain -- the ONE definition every producer
// and consumer of V strips uses (k_tick_panels, k_tick_front, the association kernels, tick_carry), so that they agree bit for bit
__device__ inline double hp_entry(const double* Hc, const double rs[5], int r)
{
    double a = 0.0;
#pragma unroll
    for (int q = 0; q < 5; ++q) a = fma(Hc[r + 2 * q], rs[q], a);
    return a;
}

// Which corrections of a round change P (bit s: correction s is live) and whether the round holds a first sighting
// (`init` != 0: the INT_MAX diagonal of slam_library.cpp:30 is about to be cancelled) -- from the plan's flags,
// wave-uniform.  A round with a first sighting is applied by the exact chain (k_tick_apply), every other round by the
// rank-2m pass (ekf_rank.h); both kernels and tick_carry evaluate this one predicate.
__device__ inline void round_flags(const TickStep* pl, int J, unsigned& actmask, bool& any_init)
{
    const int lane = threadIdx.x & 63;
    const int l16 = lane < kTickJ ? lane : 0;
    const int sk = pl[l16].skip, in = pl[l16].init;
    actmask = (unsigned)__ballot(lane < J && sk == 0) & 0xffffu;
    any_init = ((unsigned)__ballot(lane < J && in != 0) & 0xffffu) != 0u;
}

// Hand-offs BETWEEN kernels of the two streams of an overlapped run (nuslam_batch_set_overlap): a monotonic counter in
// global memory, agent scope.  Producer: every workgroup, after a barrier behind all its stores, one thread: release
// fence, drain, add.  Consumer: one thread polls (bounded), acquire fence (this CU's L1 may hold the buffer's lines of
// two ticks ago), drain, workgroup barrier -- the forms of cdna_hip_programming.md Guideline 16.  Waiting is bounded:
// if the producer never arrives the consumer gives up after ~0.2 s and the results are flagged through `ok`.
__device__ inline void tick_signal(int* cnt)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline bool tick_wait(const int* cnt, int target, int iters = 1 << 18)      // every thread of the workgroup calls it
{
    __shared__ int ok_sh;
    if (threadIdx.x == 0) {
        int ok = 0;
        for (int it = 0; it < iters; ++it) {
            if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target >= 0) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ok_sh = ok;
    }
    __syncthreads();
    return ok_sh != 0;
}

// The consumer side as a kernel of its own (one wave): the kernel enqueued behind it on the same stream starts when the
// counter has arrived, and starts with the usual kernel-boundary acquire.  (The wait used to sit at the top of
// k_tick_panels: the mere presence of the agent-scope fence in that kernel, never executed in one-stream runs, took it
// from 20 to 37 us.)
__global__ __launch_bounds__(64) void k_tick_wait(const int* __restrict__ cnt, int target, int* __restrict__ timeouts, int iters)
{
    if (!tick_wait(cnt, target, iters) && threadIdx.x == 0) atomicAdd(timeouts, 1);
}
// ... and the producer side as one: everything the kernels in front of it on its stream stored is published
__global__ __launch_bounds__(64) void k_tick_signal(int* __restrict__ cnt)
{
    if (threadIdx.x == 0) tick_signal(cnt);
}

// predict() on a 35 x 35 block of P at the index set U and on the state there, exactly as k_predict writes them into the
// full covariance (slam_library.cpp:65-148: pose advance, A = I + B at the advanced heading, A P A^T + Qbar restricted to the
// block; positions 0, 1, 2 of U are the pose).  FBs / SF: block and state before, BKo / SMo: after.  256 threads.
template <typename T>
__device__ inline void predict_block(const View& v, const MotionStep& ms, const double (*FBs)[kTickNU + 1],
                                     double (*BKo)[kTickNU + 1], const double* SF, double* SMo)
{
    // ms: motion_step(SF[0], this tick's twist) -- predictEstimate :71-94, getA :127-148 (formed by the caller, who can do it
    // while the block is still on its way)
    constexpr int NU = kTickNU, NT = 256;
    const int tid = threadIdx.x;
    const double dq_x = ms.dq_x, dq_y = ms.dq_y, th1 = ms.th1, a1 = ms.a1, a2 = ms.a2;
    if (tid < NU) {
        const int p = tid;
        SMo[p] = p == 0 ? th1 : p == 1 ? SF[1] + dq_x : p == 2 ? SF[2] + dq_y : SF[p];
    }
    if (tid == 0) {
        double pp[3][3], tt[3][3], u[3][3];
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) pp[i][j] = FBs[i][j];
        for (int j = 0; j < 3; ++j) {
            tt[0][j] = pp[0][j];
            tt[1][j] = a1 * pp[0][j] + pp[1][j];
            tt[2][j] = a2 * pp[0][j] + pp[2][j];
        }
        for (int i = 0; i < 3; ++i) {
            u[i][0] = tt[i][0];
            u[i][1] = tt[i][0] * a1 + tt[i][1];
            u[i][2] = tt[i][0] * a2 + tt[i][2];
        }
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) BKo[i][j] = (double)(T)(u[i][j] + v.Q[i + 3 * j]);
    }
    for (int e = tid; e < NU * NU; e += NT) {
        const int p = e / NU, q = e % NU;
        if (p < 3 && q < 3) continue;                                   // the corner: thread 0 above
        double val = FBs[p][q];
        if (q >= 3 && (p == 1 || p == 2)) {                             // column role: rows 1, 2 of a landmark column
            const double p0 = FBs[0][q];
            val = (double)(T)((p == 1 ? a1 : a2) * p0 + val);
        } else if (p >= 3 && (q == 1 || q == 2)) {                      // row role: columns 1, 2 of a landmark row
            const double t0 = FBs[p][0];
            val = (double)(T)(t0 * (q == 1 ? a1 : a2) + val);
        }
        BKo[p][q] = val;
    }
}

#ifdef NUSLAM_CHAIN_CLOCK
__device__ long long g_front_tl[32];          // debug builds: absolute 100 MHz stamps of one k_tick_front launch (TL below)
#define TL(k, cond) do { if (cond) g_front_tl[k] = (long long)wall_clock64(); } while (0)
#define TLMAX(k, cond) do { if (cond) atomicMax(reinterpret_cast<unsigned long long*>(&g_front_tl[k]), (unsigned long long)wall_clock64()); } while (0)
__device__ long long g_chain_clock[32];       // debug builds: per wave, 100 MHz ticks spent in each phase of the loop
#else
#define TL(k, cond) do { } while (0)
#define TLMAX(k, cond) do { } while (0)
#endif
// ------------------------------------------------------------------------------------------------ the next tick's start
// P after the previous tick's corrections and this tick's predict, restricted to this tick's index set U' -- WITHOUT
// waiting for the pass over P.  Correction s changes entry (a, b) of that 35 x 35 block through K_s(U'[a], :) and
// R_s(:, U'[b]) only, and those are numbers k_tick_panels produces anyway (it forms K_s for every row and R_s for every
// column); it drops the ones at U' into compact arrays (a position map says which threads own them), and the block from
// the covariance the pass will READ (gathered by k_tick_prep) is replayed through the sixteen corrections here, then
// predict exactly as k_predict writes it (slam_library.cpp:65-148).  Same arithmetic on every entry as the pass itself.
// Called by all 512 threads of a FUSED chain; U = the index set (already in LDS), FBs = scratch block, out: BKo, SMo and
// the control words the chain starts from.  (First versions: a 70 x 70 block carried with its own gain rows, 50-60 us on
// one CU; then this replay as a kernel of its own between the strips and the pass, ~16 us with its launch.)
template <typename T>
__device__ inline void tick_carry(const View& v, int b, const TickCarry& cy, const int* U, double (*FBs)[kTickNU + 1],
                                  double (*BKo)[kTickNU + 1], double* SMo, const int* ctrl4, int& seen, int& cached, int& brk,
                                  int& status)
{
    // 256 threads; a thread owns five entries of ONE row of the block.  What does not depend on the previous tick's strips is fetched
    // BEFORE the wait for them (the plan's scalars); the strips at U' then come in as whole contiguous
    // arrays, 16 bytes per lane (a CU's address path moves ~one lane-address per 2-3 cycles whatever the pattern: 12 k lane-loads --
    // every thread its row's 32 gains -- were 12 of this function's 21 us), and V_s = H_s R_s at U' is formed ONCE per column
    // (hp_entry, the strips' own definition) instead of once per entry.
    constexpr int NU = kTickNU, NT = 256, NE = 5, RW = NU + 1;
    static_assert(NU % NE == 0 && (NU / NE) * NU <= NT, "five entries of one row per thread");
    const int tid = threadIdx.x;
    const TickStep* pl = cy.plan_prev + (size_t)b * kTickJ;
    const int Jt = cy.Jt;
    extern __shared__ double carry_lds[];
    double* Rl = carry_lds;                                             // [kTickJ][5][RW]  R_s(q, U'[p])
    double* Kl = Rl + kTickJ * 5 * RW;                                  // [kTickJ][2][RW]  K_s(U'[p], r)
    double* Vl = Kl + kTickJ * 2 * RW;                                  // [kTickJ][2][RW]  V_s(r, U'[p])
    double (*PS)[12] = reinterpret_cast<double (*)[12]>(Vl + kTickJ * 2 * RW);
    double* SF = Vl + kTickJ * 2 * RW + kTickJ * 12;                    // [RW]
    __shared__ int canon[NU + 1];
    __shared__ int ndup_sh;
    // the arithmetic of the pass that is rewriting the covariance meanwhile: the exact chain (p1_entry) or the rank-2m sum
    const bool rank = cy.rank != 0 && !__syncthreads_or(tid < Jt && pl[tid < kTickJ ? tid : 0].init != 0);

    for (int e = tid; e < kTickJ * 12; e += NT) {
        const int st = e / 12, f = e % 12;
        const TickStep* ps = pl + (st < Jt ? st : 0);
        PS[st][f] = f < 10 ? ps->Hc[f] : (f == 10 ? (double)ps->c : (double)(st < Jt ? ps->skip : 1));
    }
    if (tid < NU) {                                                     // the position whose strips stand for this one
        int cp = tid;
        for (int k = tid - 1; k >= 0; --k)
            if (U[k] == U[tid]) cp = k;
        canon[tid] = cp;
    }
    {
        const unsigned long long dup = __ballot(tid < NU && canon[tid < NU ? tid : 0] != tid);      // (wave 0 holds all of them)
        if (tid == 0) ndup_sh = __popcll(dup);
    }
    const bool mine = tid < (NU / NE) * NU;
    const int a = mine ? tid / (NU / NE) : 0, b0 = mine ? (tid % (NU / NE)) * NE : 0;
    const int ia = U[a];
    double E[NE];
    TL(10, b == 0 && tid == 0);

    // ---- the previous tick's strips at this tick's index set are complete
    if (!tick_wait(cy.wait_cnt, cy.wait_target) && tid == 0) atomicAdd(cy.timeouts, 1);
    TL(13, b == 0 && tid == 0);
    // (the block prep gathered for this tick is on the handle's stream IN FRONT of those strips: complete as well)
#pragma unroll
    for (int k = 0; k < NE; ++k) E[k] = cy.blk[(size_t)b * NU * NU + a * NU + b0 + k];
    {
        const Pack16<double>* r2 = reinterpret_cast<const Pack16<double>*>(cy.RU + (size_t)b * kTickJ * 5 * NU);
        for (int e2 = tid; e2 < kTickJ * 5 * NU / 2; e2 += NT) {
            const Pack16<double> w = r2[e2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 2 * e2 + h, row = e / NU, p = e % NU;     // row = st * 5 + q
                Rl[row * RW + p] = w.v[h];
            }
        }
        const Pack16<double>* k2 = reinterpret_cast<const Pack16<double>*>(cy.KU + (size_t)b * kTickJ * 2 * NU);
        for (int e2 = tid; e2 < kTickJ * 2 * NU / 2; e2 += NT) {
            const Pack16<double> w = k2[e2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 2 * e2 + h, row = e / NU, p = e % NU;     // row = st * 2 + r
                Kl[row * RW + p] = w.v[h];
            }
        }
        if (tid < NU) SF[tid] = cy.SU[(size_t)b * NU + canon[tid]];
    }
    __syncthreads();
    if (ndup_sh != 0) {
        // an index that stands at several positions of U' (the same landmark seen twice, or ids out of range) has its strips at
        // the first of them only (the position map keeps the first)
        for (int p = 1; p < NU; ++p) {
            const int cp = canon[p];
            if (cp == p) continue;                                      // (uniform)
            for (int row = tid; row < kTickJ * 7; row += NT) {
                if (row < kTickJ * 5) Rl[row * RW + p] = Rl[row * RW + cp];
                else Kl[(row - kTickJ * 5) * RW + p] = Kl[(row - kTickJ * 5) * RW + cp];
            }
        }
        __syncthreads();
    }
    if (rank) {
        for (int e = tid; e < kTickJ * 2 * NU; e += NT) {
            const int row = e / NU, p = e % NU, st = row >> 1, r = row & 1;
            double rs[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) rs[q] = Rl[(st * 5 + q) * RW + p];
            Vl[row * RW + p] = hp_entry(PS[st], rs, r);
        }
        __syncthreads();
    }
    TL(11, b == 0 && tid == 0);

    if (rank) {
        // k_tick_rank's sum on these entries: acc = fma(V_f(column), -K_f(row), acc), f = 2 s + r ascending, as the f64 MFMA
        // accumulates it (a k-ordered fma chain); rounded to the storage type once, as that kernel stores it
#pragma unroll
        for (int st = 0; st < kTickJ; ++st) {
            if (st < Jt && PS[st][11] == 0.0) {
                const double K0 = Kl[(st * 2 + 0) * RW + a], K1 = Kl[(st * 2 + 1) * RW + a];
#pragma unroll
                for (int k = 0; k < NE; ++k) {
                    E[k] = fma(Vl[(st * 2 + 0) * RW + b0 + k], -K0, E[k]);
                    E[k] = fma(Vl[(st * 2 + 1) * RW + b0 + k], -K1, E[k]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NE; ++k) E[k] = (double)(T)E[k];
    } else {
#pragma unroll
        for (int st = 0; st < kTickJ; ++st) {
            const double* ps = PS[st];
            if (st < Jt && ps[11] == 0.0) {                             // (a skipped marker moves nothing in P)
                const int c = (int)ps[10];
                const double K0 = Kl[(st * 2 + 0) * RW + a], K1 = Kl[(st * 2 + 1) * RW + a];
                double m[5];
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    double kh = 0.0;                                    // gain_row's M(i, set[q]) = delta - (K H)(i, set[q])
                    kh = fma(K0, ps[0 + 2 * q], kh);
                    kh = fma(K1, ps[1 + 2 * q], kh);
                    const int sidx = q < 3 ? q : c + (q - 3);
                    m[q] = (ia == sidx ? 1.0 : 0.0) - kh;
                }
                const double bef = (ia > 2 && ia < c) ? 1.0 : 0.0, aft = (ia > c + 1) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < NE; ++k) {
                    double r[5];
#pragma unroll
                    for (int q = 0; q < 5; ++q) r[q] = Rl[(st * 5 + q) * RW + b0 + k];
                    E[k] = p1_entry<T>(m, r, E[k], bef, aft);
                }
            }
        }
    }
    TL(12, b == 0 && tid == 0);
    if (mine) {
#pragma unroll
        for (int k = 0; k < NE; ++k) FBs[a][b0 + k] = E[k];
    }
    __syncthreads();

    // ---- this tick's predict on the block and the pose, as k_predict does it (slam_library.cpp:65-148)
    {
        const TwistArg& tw = cy.tw;
        const double dth = tw.tw ? tw.tw[b * tw.stride + tw.off + 0] : tw.dth0;
        const double dx = tw.tw ? tw.tw[b * tw.stride + tw.off + 1] : tw.dx0;
        predict_block<T>(v, motion_step(SF[0], dth, dx), FBs, BKo, SF, SMo);
    }
    const int* c4 = ctrl4 + 4 * b;
    seen = c4[0]; cached = c4[0]; brk = 0; status = c4[3];              // slam.cpp:250-251 at this tick's top
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ the serial chain
// FUSED: the workgroup first waits for the previous tick's strips, replays that tick's corrections on the 35 x 35 block
// (what used to be a kernel of its own, k_tick_next: ~5 us of launch and one more hand-off per tick on the critical path
// of the run) and applies this tick's predict to it (tick_carry), then runs the chain.
// PUBLISH (k_tick_front): the strips are formed by workgroups of the SAME launch, correction by correction, while the chain
// runs on: every plan entry is stored with agent-scope stores and announced through *pub_flag = pub_base + (entries
// complete) -- one correction late, at a point where the storing waves' vmcnt(0) waits are free (their stores are >1 us old).
struct TickPublish {
    int* flag;             // per filter, kPubWords ints: [0] pub_base + number of complete plan entries of this round;
    int base;              // (predict fused in, below) [1] the chain has gathered its block, [2] predict workgroups done
    int predict;           // != 0: this launch also holds the tick's predict (k_tick_front's middle workgroups)
    int gbase, pbase;      // the values [1] / [2] reach when that has happened
    TwistArg tw;
    int rank_panels;       // != 0: the strips that follow this chain carry their panels in rank form (the host has proven the round free of
                           // first sightings and the pass is the rank-2m one): the chain stores the entry's KV rows instead of MP / BR
};
// k_tick_fused (ekf_fused.h): the pass over P runs as workgroups of the SAME launch and takes K_s / V_s from the strips as they are
// formed.  The strips then store every value as TWO self-validating 8-byte words { half of the double, tag } (agent-scope
// stores, no drain, no counter): a consumer accepts an element when both of its words carry this round's tag.
struct TickTagged {
    long long* tagK;       // [B][2 kTickJ][ld][2]
    long long* tagV;
    int tag;               // this round's tag (never 0: the buffers start zeroed)
    // the state vector the round leaves, mirrored into mapped pinned HOST memory as the strips form it (getStateVector() of the
    // caller's loop, slam.cpp:184,250, then needs neither a stream synchronisation nor a device-to-host copy): every strip
    // workgroup stores its entries (system scope), drains, and stamps mtags[workgroup] = mseq
    double* mirror;        // null: no mirror
    long long* mtags;
    long long mseq;
};
// k_run_fused (ekf_fused.h): the ticks of a resident trace inside ONE launch, the covariance resident in the pass workgroups' registers
// in between.  A role is then called once per tick (RUN): what another workgroup left in the PREVIOUS tick of the same launch -- the
// state, the rows / columns of P the pass workgroups exported -- is read with agent-scope loads, behind a count of the workgroups that
// have finished that tick (strips: state stored; pass: exports stored).
struct TickRun {
    int* done;             // one word per strip workgroup, then one per pass workgroup: the last tick (a running number) the workgroup has
                           // finished -- state stored / exports stored.  (A shared counter serialises ~400 device-scope atomics per tick.)
    int* done_all;         // one word per pass workgroup: ALL of its exports of that tick are stored (`done`'s pass words say so of the entries
                           // the chain gathers -- row AND column in the next index set --, which a workgroup stores first: the rest, the
                           // rows / columns the strips read, is waited for by the predict role, off the chain's path)
    int stamp;             // this tick's number: what a workgroup leaves in its word; the chain waits for every word to reach stamp - 1
    int first;             // != 0: the launch's first tick -- nothing to wait for
    int stamp_dbg_second_last;   // (debug timeline: the launch's second-to-last tick)
    long long toff;        // this tick's offset into the resident trace, added to TickObs::off
    int* timeouts;         // expired waits (NUSLAM_E_SYNC)
    int n_strip, n_pass, tiles_r, tiles_c;   // the words that exist; pass workgroup idx is live iff ((idx >> 3) / tiles_c) * 8 + (idx & 7) < tiles_r
};
__device__ inline void st_sys(double* p, double x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline void st_sys(long long* p, long long x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// { half, tag } words of a tagged strip element
__device__ inline void st_tagged(long long* p, double x, int tag)
{
    const long long bits = __double_as_longlong(x);
    st_agent(p, (long long)(((unsigned long long)(unsigned)tag << 32) | (unsigned long long)(unsigned)(bits >> 32)));
    st_agent(p + 1, (long long)(((unsigned long long)(unsigned)tag << 32) | (unsigned long long)(unsigned)(bits & 0xffffffffll)));
}
// returns false while the element has not arrived
__device__ inline bool ld_tagged(const long long* p, int tag, double& x)
{
    const long long w0 = ld_agent(p), w1 = ld_agent(p + 1);
    x = __longlong_as_double((long long)(((unsigned long long)w0 << 32) | ((unsigned long long)w1 & 0xffffffffull)));
    return (int)(w0 >> 32) == tag && (int)(w1 >> 32) == tag;
}
constexpr int kPubWords = 4;
__device__ inline void plan_store(bool publish, double* p, double x)
{
    if (publish) st_agent(p, x);
    else *p = x;
}
__device__ inline void plan_store_head(bool publish, TickStep* ps, int skip, int init, int c, int id)
{
    if (publish) {                                                      // (four ints = two 8-byte words)
        st_agent(reinterpret_cast<long long*>(&ps->skip), (long long)(unsigned)skip | ((long long)init << 32));
        st_agent(reinterpret_cast<long long*>(&ps->c), (long long)(unsigned)c | ((long long)id << 32));
    } else { ps->skip = skip; ps->init = init; ps->c = c; ps->id = id; }
}

template <typename T, bool FUSED, bool PUBLISH, bool RUN = false>
__device__ inline void tick_chain(const int b, View v, TickObs o, int total_landmarks, const T* __restrict__ P,
                                  TickStep* __restrict__ plan, TickCarry cy,
                                  int* __restrict__ ctrl_out4, int* __restrict__ done_cnt, TickPublish pub, TickRun rn = TickRun{})
{
    const long long ooff = RUN ? o.off + rn.toff : o.off;
    constexpr int NU = kTickNU;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ld = v.ld;
    const double* s = v.s_in + (size_t)b * ld;
    const T* Pb = P + (size_t)b * v.p_stride;
    TickStep* pl = plan + (size_t)b * kTickJ;
    const int J = o.J;
    TL(0, PUBLISH && b == 0 && tid == 0);                               // chain: entry
    // The chain is the tick's critical path and one wave per SIMD; in overlapped runs waves of the pass over P share its CU: its
    // instructions go first
    __builtin_amdgcn_s_setprio(3);

    __shared__ double BK[2][NU][NU + 1];          // P(U[p], U[p']) before / after the current correction
    __shared__ double SM[2][NU + 1];              // state at U
    __shared__ double MPl[2][NU][8];              // M at the rows of U: of the correction in flight / of the one whose block update is pending
    __shared__ double zr[kTickJ], zphi[kTickJ];   // the markers in polar form (slam.cpp:286)
    __shared__ double hd[2][20];                  // Hc[10], Sinv[4], lx, ly, dz0, dz1 of the correction whose head is being formed / was formed last
    __shared__ int hi[2];
    __shared__ int Ush[NU + 1];
    __shared__ int idsh[kTickJ];
    __shared__ double KVl[NU][4];                 // the plan entry's KV rows (K_s at the rows U[p], V_s at the columns U[p]) on their way out
    __shared__ int4 dlive[kTickJ + 1];            // the corrections that change P, in order: { marker, first state index, resolved id, init }
    __shared__ int dsum[4];                       // landmarks seen / break flag after the round, first marker with a bounds error, live count

    if (tid < kTickJ) {
        int id = 0;
        if (tid < J) id = o.ids ? o.ids[b * o.stride + ooff + tid] : o.id0[tid];
        idsh[tid] = id;
        const int c = (id >= 1 && id <= v.n) ? 3 + 2 * (id - 1) : 3;      // always a readable index
        Ush[3 + 2 * tid] = c;
        Ush[4 + 2 * tid] = c + 1;
    }
    if (tid < 3) Ush[tid] = tid;
    __syncthreads();
    int seen = 0, brk = 0, status = 0, cached = 0;
    if (FUSED) {
        tick_carry<T>(v, b, cy, Ush, BK[1], BK[0], SM[0], ctrl_out4, seen, cached, brk, status);    // (waits, inside, for the previous tick's strips)
        TL(14, PUBLISH && b == 0 && tid == 0);                          // ... and replayed on this tick's block
    } else {
        // Everything the round starts from is REQUESTED first -- this tick's twist and heading, the markers, the control words, the
        // 35 x 35 block -- and what does not need the block is computed while it is on its way: the motion step's sines and
        // cosines (motion_step), the markers' polar forms.
        const bool fpred = PUBLISH && pub.predict;                      // the tick's predict runs in THIS launch (k_tick_front)
        if constexpr (RUN) {
            // the previous tick of this launch is complete: every strip workgroup has stored its state entries, every pass workgroup the
            // rows / columns of this tick's index set (bounded: an expired wait goes on and is reported, as everywhere)
            if (!rn.first) {                                            // (uniform) every thread looks at its share of the words
                // (a thread's words are asked for TOGETHER: an agent-scope load is a ~1.5 us round trip)
                constexpr int kWordsPerThread = 4;                      // (up to 1024 words)
                const int nw = rn.n_strip + rn.n_pass;
                int wi[kWordsPerThread];
                bool need[kWordsPerThread];
#pragma unroll
                for (int q = 0; q < kWordsPerThread; ++q) {
                    const int w = tid + 256 * q, idx = w - rn.n_strip;
                    need[q] = w < nw && !(idx >= 0 && ((idx >> 3) / rn.tiles_c) * 8 + (idx & 7) >= rn.tiles_r);      // (no such tile)
                    wi[q] = need[q] ? w : 0;
                }
                bool ok = false;
                for (int it = 0; it < (1 << 20) && !ok; ++it) {
                    int val[kWordsPerThread];
#pragma unroll
                    for (int q = 0; q < kWordsPerThread; ++q) val[q] = need[q] ? ld_agent(rn.done + wi[q]) : 0;
                    ok = true;
#pragma unroll
                    for (int q = 0; q < kWordsPerThread; ++q) {
                        if (need[q] && seq_reached(val[q], rn.stamp - 1)) need[q] = false;
                        ok = ok && !need[q];
                    }
                }
                if (!__syncthreads_and(ok ? 1 : 0) && tid == 0) atomicAdd(rn.timeouts, 1);
            }
            TL(16, b == 0 && tid == 0);                                 // run: the previous tick's state and block entries are stored
        }
        double dth = 0.0, dxx = 0.0, theta = 0.0;
        if (fpred) {
            const TwistArg& tw = pub.tw;
            dth = tw.tw ? tw.tw[b * tw.stride + tw.off + 0] : tw.dth0;
            dxx = tw.tw ? tw.tw[b * tw.stride + tw.off + 1] : tw.dx0;
            theta = RUN ? ld_agent(&s[0]) : s[0];
        }
        double oa = 0.0, ob = 0.0;
        if (wave == 3 && lane < J) {
            oa = o.a ? o.a[b * o.stride + ooff + lane] : o.a0[lane];
            ob = o.b ? o.b[b * o.stride + ooff + lane] : o.b0[lane];
        }
        const int* ci = v.c_in + b * C_WORDS;
        seen = ci[C_SEEN]; brk = ci[C_BRK]; status = ci[C_STATUS]; cached = ci[C_SEEN_CACHED];
        constexpr int NG = (NU * NU + 255) / 256;
        double g[NG], gp[NG];
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            // consecutive lanes walk DOWN a column: the two rows of a landmark share a cache line (each scattered 8-byte
            // read costs ~16 cycles of this CU's one address path: 1225 of them were 8 us of every tick)
            const int e = tid + 256 * u, ec = e < NU * NU ? e : 0;
            const int q = ec / NU, p = ec % NU;
            g[u] = RUN ? (double)ld_agent(&Pb[(size_t)Ush[q] * ld + Ush[p]]) : (double)Pb[(size_t)Ush[q] * ld + Ush[p]];
            // fpred: the entry of row 0 (column 0) predict combines it with (predict_block's column / row role), fetched by the
            // thread itself so that predict can be applied before anything passes through LDS
            gp[u] = 0.0;
            {
                const bool colr = q >= 3 && (p == 1 || p == 2), rowr = p >= 3 && (q == 1 || q == 2);
                if (fpred && (colr || rowr)) {                          // (128 of the 1225)
                    const T* gsrc = &Pb[(size_t)(rowr ? 0 : Ush[q]) * ld + (colr ? 0 : Ush[p])];
                    gp[u] = RUN ? (double)ld_agent(gsrc) : (double)*gsrc;
                }
            }
        }
        double gc = 0.0;                                                // the 3 x 3 pose corner, entry (i, j) on lane i + 3 j of wave 0
        if (fpred && tid < 9) gc = RUN ? (double)ld_agent(&Pb[(size_t)(tid / 3) * ld + tid % 3]) : (double)Pb[(size_t)(tid / 3) * ld + tid % 3];
        const double gs = RUN ? ld_agent(&s[Ush[tid < NU ? tid : 0]]) : s[Ush[tid < NU ? tid : 0]];
        MotionStep ms{};
        if (fpred) ms = motion_step(theta, dth, dxx);
        if (wave == 3 && lane < kTickJ) {         // all markers' polar forms at once, one lane each
            double r, phi;
            if (o.cartesian) cartesian2polar(oa, ob, r, phi);
            else { r = oa; phi = ob; }
            zr[lane] = r;
            zphi[lane] = phi;
        }
        // What was gathered is the covariance BEFORE predict (the middle workgroups of k_tick_front rewrite rows / columns 1, 2 of P
        // in place once told that it has arrived): predict is applied to the block and the pose on the way into LDS, with
        // k_predict's own arithmetic entry by entry (the cases of predict_block; slam_library.cpp:65-148)
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int e = tid + 256 * u;
            if (e < NU * NU) {
                const int q = e / NU, p = e % NU;
                double val = g[u];
                if (fpred) {
                    if (q >= 3 && (p == 1 || p == 2)) val = (double)(T)((p == 1 ? ms.a1 : ms.a2) * gp[u] + val);
                    else if (p >= 3 && (q == 1 || q == 2)) val = (double)(T)(gp[u] * (q == 1 ? ms.a1 : ms.a2) + val);
                }
                if (!(fpred && p < 3 && q < 3)) BK[0][p][q] = val;      // (the corner: thread 0 below)
            }
        }
        if (fpred && wave == 0) {
            double pp[3][3], tt[3][3], uu[3][3];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i) pp[i][j] = lane_bcast(gc, i + 3 * j);
            const double a1 = ms.a1, a2 = ms.a2;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                tt[0][j] = pp[0][j];
                tt[1][j] = a1 * pp[0][j] + pp[1][j];
                tt[2][j] = a2 * pp[0][j] + pp[2][j];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                uu[i][0] = tt[i][0];
                uu[i][1] = tt[i][0] * a1 + tt[i][1];
                uu[i][2] = tt[i][0] * a2 + tt[i][2];
            }
            if (tid == 0) {
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int i = 0; i < 3; ++i) BK[0][i][j] = (double)(T)(uu[i][j] + v.Q[i + 3 * j]);
            }
        }
        if (tid < NU) {
            double sv = gs;
            if (fpred) sv = tid == 0 ? ms.th1 : tid == 1 ? gs + ms.dq_x : tid == 2 ? gs + ms.dq_y : gs;
            SM[0][tid] = sv;
        }
        if (fpred) {
            __syncthreads();                                            // (every gathered value has arrived: the predict workgroups may go)
            TL(10, b == 0 && tid == 0);                                 // chain: block gathered, predict applied
            if (tid == 0) st_agent(pub.flag + kPubWords * b + 1, pub.gbase);
            cached = seen; brk = 0;                                     // the tick's bookkeeping (slam.cpp:250-251)
        }
    }
    if (FUSED && wave == 3 && lane < kTickJ) {    // (overlapped runs: the markers' polar forms)
        double a = 0.0, bb = 0.0;
        if (lane < J) {
            a = o.a ? o.a[b * o.stride + o.off + lane] : o.a0[lane];
            bb = o.b ? o.b[b * o.stride + o.off + lane] : o.b0[lane];
        }
        double r, phi;
        if (o.cartesian) cartesian2polar(a, bb, r, phi);
        else { r = a; phi = bb; }
        zr[lane] = r;
        zphi[lane] = phi;
    }
    if (wave == 2) {
        // The caller's decision chain for every marker of the round (slam.cpp:295-316), once, one lane per marker, beside the
        // polar forms: it depends on the ids and the counters only (a singular S changes the status word, nothing else), so the
        // loop below walks a list of live corrections and has no decision on its critical path.  What one marker hands to the
        // next is the break flag (set by the first marker that takes the branch of :301-316; which one does not depend on
        // it) and the running maximum of `seen`, which no decision reads.  Markers that change nothing get their plan entry here.
        const int st = lane;
        const bool in = st < J;
        const int id_st = in ? idsh[st] : 0;
        const int dmode = o.forced ? MODE_FORCE : MODE_KNOWN;          // forced: the caller's own chain ran on the host (lazy class API)
        const Decision d0 = resolve(v.n, id_st, seen, cached, brk, 0, dmode, total_landmarks);   // with the round's incoming break flag
        const unsigned long long brk_set = __ballot(in && d0.new_brk != 0 && brk == 0);
        const bool brk_here = brk != 0 || (brk_set & ((1ull << st) - 1ull)) != 0ull;
        Decision d = resolve(v.n, id_st, seen, cached, brk_here ? 1 : 0, 0, dmode, total_landmarks);
        if (o.forced && !d.skip) d.init = ((o.init_mask >> st) & 1u) != 0u;
        const unsigned long long live = __ballot(in && !d.skip), bad = __ballot(in && d.new_status != 0);
        int smax = in ? d.new_seen : seen;                              // (new_seen = max(seen, id) where the marker counts)
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) smax = max(smax, __shfl_xor(smax, off, 64));
        if (in) {
            if (v.id_log && o.log_slot0 >= 0) v.id_log[(size_t)b * v.log_stride + o.log_slot0 + st] = d.id;
            if (d.skip) plan_store_head(PUBLISH, pl + st, 1, 0, 3, d.id);
            else dlive[__popcll(live & ((1ull << st) - 1ull))] = make_int4(st, d.c, d.id, d.init ? 1 : 0);
        }
        if (lane == 0) {
            const int nl = __popcll(live);
            dlive[nl] = make_int4(0, 3, 0, 0);
            dsum[0] = smax; dsum[1] = (brk != 0 || brk_set != 0ull) ? 1 : 0;     // (forced: both pass through unchanged)
            dsum[2] = bad ? __ffsll((long long)bad) - 1 : 255; dsum[3] = nl;
        }
    }
    __syncthreads();
    seen = dsum[0]; brk = dsum[1];
    const int nlive = dsum[3];

    // ONE phase (one workgroup barrier) per correction.  Between the barrier that ends the head of correction s-1 and the
    // one that ends the head of correction s, every wave works from the same three things -- the block and the state BEFORE
    // correction s-1 (BK[bcur], SM[scur]) and that correction's head (H, S^-1, innovation: hd[hb]) -- and forms for itself
    // whatever it needs of what correction s-1 leaves:
    //   wave 2  the rows of M_{s-1} and the state at set_s (five gain rows), with them the 25 entries P_{s-1}(set_s, set_s)
    //           (p1_entry, the block update's own formula), then H_s, S_s, S_s^-1                    -> hd[hb ^ 1]
    //   wave 1  the prior rows and the scalars of correction s-1 to the plan; the state at set_s likewise, the polar form of the
    //           landmark offset and the innovation of correction s                                     -> hd[hb ^ 1]
    //   wave 0  K, M at all rows of U and the state there after correction s-1 (the plan's MP rows), a third of the block update
    //   wave 3  the rows of M once more (its own copy), two thirds of the block update
    // (Until round 3 the gain rows had a phase of their own between head and head: 0.55 us and a barrier per correction on
    // the critical path; now only the head is on it.)  The arithmetic of every value is unchanged.
    auto advance_entry = [&](const double (*Bo)[NU + 1], const double (*Mp)[8], const int spP[5], int p, int pp) {
        double mrow[5], r[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) { mrow[q] = Mp[p][q]; r[q] = Bo[spP[q]][pp]; }
        return p1_entry<T>(mrow, r, Bo[p][pp], Mp[p][5], Mp[p][6]);
    };
    // the block update: 192 slots (wave 0 one, wave 3 two)
    auto run_pending = [&](const double (*Bo)[NU + 1], double (*Bn)[NU + 1], const double (*Mp)[8], int posP, int me) {
        // only the rows / columns a later correction still reads: positions 0, 1, 2 and those behind marker posP (the
        // same liveness as in the strips), walked as a 16 x 12 thread grid over the live index list -- no division
        const int spP[5] = { 0, 1, 2, posP, posP + 1 };
        const int nl = 3 + (NU - (posP + 2));
        const int ty = me / 12, tx = me - 12 * ty;
        for (int a = ty; a < nl; a += 16) {
            const int p = a < 3 ? a : posP + 2 + (a - 3);
            for (int c2 = tx; c2 < nl; c2 += 12) {
                const int pp = c2 < 3 ? c2 : posP + 2 + (c2 - 3);
                Bn[p][pp] = advance_entry(Bo, Mp, spP, p, pp);
            }
        }
    };
    int bcur = 0, scur = 0, hb = 0;
    bool have = false;                    // correction pv_* has its head in hd[hb]; its gain rows, state and block update are pending
    int pv_st = 0, pv_pos = 3, pv_c = 3, pv_id = 0;
    bool pv_init = false;
#ifdef NUSLAM_CHAIN_CLOCK
    long long ck[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, ct = (long long)wall_clock64();
// (a stamp costs ~0.05 us and pins the code around it: with NUSLAM_CHAIN_CLOCK == 1 only the two around the barrier are taken --
// busy time per wave --, with 2 all of them)
#define CK(k) do { if (NUSLAM_CHAIN_CLOCK > 1 || (k) < 2) { const long long n__ = (long long)wall_clock64(); ck[k] += n__ - ct; ct = n__; } } while (0)
#else
#define CK(k) do { } while (0)
#endif
    // the head of correction pv, in registers; != 0: its S was singular (read in one LDS round trip, in front of the branch on it)
    double hP[18];
    auto load_head = [&]() {
        const int sg = hi[hb];
#pragma unroll
        for (int q = 0; q < 18; ++q) hP[q] = hd[hb][q];
        return sg;
    };
    int4 dn = dlive[0];                   // the next live correction: { marker, first state index, resolved id, init }
    // one phase: what correction pv leaves (if `have`), and the head of correction st (if has_cur)
    auto phase = [&](const bool has_cur, const int st, const bool cur_init, const int k_next) {
        const double (*B0)[NU + 1] = BK[bcur];                          // the block before correction pv, complete
        const double* S0 = SM[scur];
        double* hN = hd[hb ^ 1];
        const int posP = pv_pos, cP = pv_c;
        const int spP[5] = { 0, 1, 2, posP, posP + 1 };                 // positions of set_pv in U
        const int setP[5] = { 0, 1, 2, cP, cP + 1 };
        TickStep* psP = pl + pv_st;
        // K, M of correction pv at row U[p] and the state there after it (slam_library.cpp:271-275; the heading still raw)
        auto row_after = [&](int p, double m[5], double& bef, double& aft, double KP[2]) {
            const int i = Ush[p];
            double pc[5], Hc[10], Si[4];
#pragma unroll
            for (int q = 0; q < 5; ++q) pc[q] = B0[p][spP[q]];          // P(U[p], set[q])
#pragma unroll
            for (int q = 0; q < 10; ++q) Hc[q] = hP[q];
#pragma unroll
            for (int q = 0; q < 4; ++q) Si[q] = hP[10 + q];
            gain_row(pc, Hc, Si, i, setP, KP, m);
            bef = (i > 2 && i < cP) ? 1.0 : 0.0;
            aft = (i > cP + 1) ? 1.0 : 0.0;
            double sv = (pv_init && i == cP) ? hP[14] : (pv_init && i == cP + 1) ? hP[15] : S0[p];
            double acc = 0.0;
            acc = fma(KP[0], hP[16], acc);
            acc = fma(KP[1], hP[17], acc);
            return sv + acc;                                            // :275
        };
        const int pos = 3 + 2 * st;
        const int sp[5] = { 0, 1, 2, pos, pos + 1 };                    // positions of set_st in U
        if (wave == 0) {
            if (have) {
                if (lane < NU) {
                    const int p = lane;
                    double m[5], bef, aft;
                    double KP[2];
                    double sv = row_after(p, m, bef, aft, KP);
                    double (*Mn)[8] = MPl[0];
#pragma unroll
                    for (int q = 0; q < 5; ++q) { Mn[p][q] = m[q]; if (!PUBLISH) psP->MP[p][q] = m[q]; }
                    Mn[p][5] = bef; Mn[p][6] = aft;
                    if (!PUBLISH) {                                     // (the strip workgroups of k_tick_front form the two flags themselves)
                        psP->MP[p][5] = bef; psP->MP[p][6] = aft;
                        psP->MP[p][7] = 0.0;
                    }
                    if (p == 0) {                                       // the heading row after that correction, wrapped (:276)
                        sv = normalize_angle(sv);
                        plan_store(PUBLISH, &psP->heading, sv);
                    }
                    SM[scur ^ 1][p] = sv;
                    // what rank-form strips need of this position: K_s(U[p], :) and V_s(:, U[p]) = H_s P_{s-1}(set_s, U[p]) (hp_entry)
                    if (!PUBLISH || pub.rank_panels) {
                        double rsv[5];
#pragma unroll
                        for (int q = 0; q < 5; ++q) rsv[q] = B0[spP[q]][p];
                        const double V0 = hp_entry(hP, rsv, 0), V1 = hp_entry(hP, rsv, 1);
                        if (PUBLISH) { KVl[p][0] = KP[0]; KVl[p][1] = KP[1]; KVl[p][2] = V0; KVl[p][3] = V1; }
                        else { psP->KV[p][0] = KP[0]; psP->KV[p][1] = KP[1]; psP->KV[p][2] = V0; psP->KV[p][3] = V1; }
                    }
                }
                if (PUBLISH && pub.rank_panels) {
                    // (coalesced, read back from this wave's own LDS copy, as the rows of M below)
#pragma unroll
                    for (int k5 = 0; k5 < (NU * 4 + 63) / 64; ++k5) {
                        const int idx = 64 * k5 + lane;
                        if (idx < NU * 4) st_agent(&psP->KV[0][0] + idx, (&KVl[0][0])[idx]);
                    }
                } else if (PUBLISH) {
                    // the rows of M to the plan, COALESCED: lane l of store k writes word 64 k + l of MP (rows of 8 words), read
                    // back from the LDS copy this wave has just written (a wave's LDS accesses execute in order).  With one row
                    // per lane every store instruction touched 35 different cache lines through the CU's one address path.
                    const double (*Mr)[8] = MPl[0];
#pragma unroll
                    for (int k5 = 0; k5 < (NU * 8 + 63) / 64; ++k5) {
                        const int idx = 64 * k5 + lane, p = idx >> 3, q = idx & 7;
                        if (p < NU && q < 5) st_agent(&psP->MP[0][0] + idx, Mr[p][q]);
                    }
                }
                if (has_cur) run_pending(B0, BK[bcur ^ 1], MPl[0], posP, lane);
            }
        } else if (wave == 3) {
            if (have && has_cur) {
                if (lane < NU) {
                    double m[5], bef, aft;
                    double KP_[2];
                    (void)row_after(lane, m, bef, aft, KP_);
                    double (*Mn)[8] = MPl[1];
#pragma unroll
                    for (int q = 0; q < 5; ++q) Mn[lane][q] = m[q];
                    Mn[lane][5] = bef; Mn[lane][6] = aft;
                }
                run_pending(B0, BK[bcur ^ 1], MPl[1], posP, 64 + lane);
                run_pending(B0, BK[bcur ^ 1], MPl[1], posP, 128 + lane);
            }
        } else {
            if (wave == 1 && have) {
                // the plan entry of correction pv: the five prior rows at the columns of U and the scalars -- issued first, so that
                // the wait for them in front of the barrier finds them acknowledged
                const int fresh = B0[posP][posP] > 1.0e9 ? 2 : 0;       // INT_MAX (2.1e9) still on the landmark's diagonal
                if (PUBLISH) {
                    if (!pub.rank_panels) {
#pragma unroll
                        for (int k5 = 0; k5 < (NU * 8 + 63) / 64; ++k5) {
                            const int idx = 64 * k5 + lane, p = idx >> 3, q = idx & 7;
                            if (p < NU && q < 5) st_agent(&psP->BR[0][0] + idx, B0[spP[q]][p]);
                        }
                    }
                    // the entry's scalars as ONE store instruction, a word per lane: {skip, init}, {c, id}, then Hc[10], Sinv[4],
                    // dz[2], lxy[2], contiguous behind them
                    const int w = lane;
                    if (w < 20) {
                        long long word;
                        if (w == 0) word = (long long)(unsigned)0 | ((long long)((pv_init ? 1 : 0) | fresh) << 32);
                        else if (w == 1) word = (long long)(unsigned)cP | ((long long)pv_id << 32);
                        else {
                            const int f = w - 2;                        // 0..13: Hc, Sinv = hd[0..13]; 14, 15: dz = hd[16], hd[17]; 16, 17: lxy = hd[14], hd[15]
                            word = __double_as_longlong(hd[hb][f < 14 ? f : (f < 16 ? f + 2 : f - 2)]);
                        }
                        st_agent(reinterpret_cast<long long*>(psP) + w, word);
                    }
                } else {
                    if (lane < NU) {
#pragma unroll
                        for (int q = 0; q < 5; ++q) psP->BR[lane][q] = B0[spP[q]][lane];
                    }
                    if (lane == 0) {
                        plan_store_head(false, psP, 0, (pv_init ? 1 : 0) | fresh, cP, pv_id);
#pragma unroll
                        for (int q = 0; q < 10; ++q) psP->Hc[q] = hP[q];
#pragma unroll
                        for (int q = 0; q < 4; ++q) psP->Sinv[q] = hP[10 + q];
                        psP->dz[0] = hP[16]; psP->dz[1] = hP[17]; psP->lxy[0] = hP[14]; psP->lxy[1] = hP[15];
                    }
                }
            }
            if (has_cur) {
                // the state at set_st after correction pv: lane j (mod 5) forms the row of position sp[j]
                const int j = lane % 5, p = sp[j];
                double m[5], bef = 0.0, aft = 0.0, sv;
                CK(2);
                double KP_[2];
                if (have) sv = row_after(p, m, bef, aft, KP_);
                else sv = S0[p];
                CK(3);
                const double th_raw = lane_bcast(sv, 0), x = lane_bcast(sv, 1), y = lane_bcast(sv, 2);
                double lx, ly;
                if (cur_init) {                                         // initializeLandmark, slam_library.cpp:255-261: wrapped heading
                    const double th = have ? normalize_angle(th_raw) : th_raw, r = zr[st], phi = zphi[st];
                    lx = x + r * cos(phi + th);
                    ly = y + r * sin(phi + th);
                } else { lx = lane_bcast(sv, 3); ly = lane_bcast(sv, 4); }
                if (wave == 2) {
                    // P(set[q2], set[q]) after correction pv: one entry per lane (lane e = 5 q + q2 holds the row of M it needs:
                    // q2 == j), then broadcast
                    double ent;
                    {
                        const int e = lane < 25 ? lane : j;
                        const int q = e / 5, q2 = e % 5;
                        const double pij = B0[sp[q2]][sp[q]];
                        if (have) {
                            double r[5];
#pragma unroll
                            for (int k = 0; k < 5; ++k) r[k] = B0[spP[k]][sp[q]];
                            ent = p1_entry<T>(m, r, pij, bef, aft);
                        } else ent = pij;
                    }
                    CK(4);
                    // H (:268), S = H P H^T + R and S^-1 (:270) with the lanes of this wave side by side: a wave issues one fp64
                    // instruction per ~6.6 cycles whether 1 or 64 lanes take part (tools/microbench/latency.hip), so the four
                    // quotients of the Jacobian are ONE division on four lanes, the ten entries of H P one 5-FMA chain on ten
                    // lanes, S one on four, S^-1 one division on four.  Every value is formed by the operations of
                    // jacobian_compact / innovation_cov_block / inv2 in their order.
                    const double dx = lx - x, dy = ly - y;
                    const double dd = (dx * dx) + (dy * dy);
                    const double sd = sqrt(dd);
                    const int k4 = lane & 3;
                    const double qv = ((k4 & 1) ? dy : dx) / ((k4 & 2) ? dd : sd);        // lanes 0..3: dx/sd, dy/sd, dx/d, dy/d
                    const double xs = lane_bcast(qv, 0), ys = lane_bcast(qv, 1), xd = lane_bcast(qv, 2), yd = lane_bcast(qv, 3);
                    // Hc[r + 2 q] = H(r, set[q]) = { 0, -1 | -xs, yd | -ys, -xd | xs, -yd | ys, xd }: named scalars, selected per lane
                    // (an array indexed by a lane-dependent row would live in scratch memory)
                    const double nxs = -xs, nys = -ys, nxd = -xd, nyd = -yd;
#define NUSLAM_HROW(rr, k) ((k) == 0 ? ((rr) ? -1.0 : 0.0) : (k) == 1 ? ((rr) ? yd : nxs) : (k) == 2 ? ((rr) ? nxd : nys) \
                            : (k) == 3 ? ((rr) ? nyd : xs) : ((rr) ? xd : ys))
                    CK(5);
                    // (H P)(r, set[q]) on lane 2 q + r: the sum over q2 of H(r, set[q2]) P(set[q2], set[q]), ascending
                    double hp = 0.0;
                    {
                        const int L = lane < 10 ? lane : 0, q = L >> 1, r = L & 1;
#pragma unroll
                        for (int q2 = 0; q2 < 5; ++q2) {
                            const double pv = __shfl(ent, 5 * q + q2, 64);
                            hp = fma(NUSLAM_HROW(r, q2), pv, hp);
                        }
                    }
                    // S(r, s2) on lane r + 2 s2: the sum over q of (H P)(r, set[q]) H(s2, set[q]), ascending, then + R
                    double Sv;
                    {
                        const int M = lane & 3, r = M & 1, s2 = M >> 1;
                        double sacc = 0.0;
#pragma unroll
                        for (int q = 0; q < 5; ++q) {
                            const double hv = __shfl(hp, 2 * q + r, 64);
                            sacc = fma(hv, NUSLAM_HROW(s2, q), sacc);
                        }
                        const double R0 = v.R[0], R1 = v.R[1], R2 = v.R[2], R3 = v.R[3];
                        Sv = sacc + (M == 0 ? R0 : M == 1 ? R1 : M == 2 ? R2 : R3);
                    }
                    CK(6);
                    // S^-1 (inv2: Armadillo's tiny-matrix inverse; general LU when |det| < epsilon), column-major: entry M on lane M
                    const double sa = lane_bcast(Sv, 0), sc = lane_bcast(Sv, 1), sb = lane_bcast(Sv, 2), sdd = lane_bcast(Sv, 3);
                    const double det = (sa * sdd) - (sb * sc);
                    double Siv;
                    int sing = 0;
                    if (fabs(det) >= DBL_EPSILON) {
                        const int M = lane & 3;
                        Siv = (M == 0 ? sdd : M == 1 ? -sc : M == 2 ? -sb : sa) / det;
                    } else {
                        const double S4[4] = { sa, sc, sb, sdd };
                        double Si[4];
                        sing = inv2(S4, Si);
                        const int M = lane & 3;
                        Siv = M == 0 ? Si[0] : M == 1 ? Si[1] : M == 2 ? Si[2] : Si[3];
                    }
                    if (lane < 4) hN[10 + lane] = Siv;
                    if (lane == 0) {
                        hN[0] = 0.0; hN[1] = -1.0; hN[2] = nxs; hN[3] = yd; hN[4] = nys; hN[5] = nxd; hN[6] = xs; hN[7] = nyd;
                        hN[8] = ys; hN[9] = xd;
                        hN[14] = lx; hN[15] = ly;
                        hi[hb ^ 1] = sing;
                    }
#undef NUSLAM_HROW
                } else {
                    const double mx = lx - x, my = ly - y;              // measurement(): :152-156
                    double zr_h, zb_h;
                    cartesian2polar(mx, my, zr_h, zb_h);
                    CK(4);
                    // ... and the innovation, against the WRAPPED heading (:157-159, :276)
                    const double th_w = normalize_angle(th_raw);
                    const double zb = normalize_angle(zb_h - th_w);
                    if (lane == 0) { hN[16] = zr[st] - zr_h; hN[17] = zphi[st] - zb; }   // :272, bearing not wrapped
                }
            }
        }
        // (PUBLISH: the stores of entry pv -- wave 0's rows of M and heading, wave 1's prior rows and scalars, issued at the top
        // of this phase -- are waited for here)
        if (PUBLISH && wave != 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (wave 2: the entries of the skipped markers)
        dn = dlive[k_next];                                             // (for the next trip; it arrives while this wave waits below)
        CK(0);
        lds_barrier();
        CK(1);
        if (have) { scur ^= 1; bcur ^= 1; }
        hb ^= 1;
    };
    int sing_at = 255;                    // first marker whose S was singular
    // the singular S of correction pv (update() throws behind initializeLandmark, slam_library.cpp:263-270): nothing but the first
    // sighting's state rows changes
    auto settle_singular = [&]() {
        TickStep* psP = pl + pv_st;
        const double lx = hP[14], ly = hP[15];
        const int fresh = BK[bcur][pv_pos][pv_pos] > 1.0e9 ? 2 : 0;
        if (sing_at == 255) sing_at = pv_st;
        if (tid == 0) {
            plan_store_head(PUBLISH, psP, 1, (pv_init ? 1 : 0) | fresh, pv_c, pv_id);
            plan_store(PUBLISH, &psP->lxy[0], lx); plan_store(PUBLISH, &psP->lxy[1], ly);
        }
        if (pv_init && tid < NU) {
            const int i = Ush[tid];
            if (i == pv_c) SM[scur][tid] = lx;
            if (i == pv_c + 1) SM[scur][tid] = ly;
        }
        lds_barrier();
        have = false;
    };
#ifdef NUSLAM_CHAIN_CLOCK
    const long long sclk0 = (long long)clock64(), wclk0 = (long long)wall_clock64();
#endif
    TL(1, PUBLISH && b == 0 && tid == 0);                               // chain: loop start
    for (int k = 0; k < nlive; ++k) {
        CK(7);
        const int4 dc = dn;
        const int st = dc.x;
        if (have && load_head() != 0) settle_singular();
        phase(true, st, dc.w != 0, k + 1);
        // entries 0 .. st-1 are complete: the storing waves waited for their stores in front of the barrier just passed
        if (PUBLISH && tid == 192) st_agent(pub.flag + kPubWords * b, (int)((unsigned)pub.base + (unsigned)st));
        have = true;
        pv_st = st; pv_pos = 3 + 2 * st; pv_c = dc.y; pv_id = dc.z; pv_init = dc.w != 0;
    }
    if (have && load_head() != 0) settle_singular();
    if (status == 0) status = dsum[2] < sing_at ? kStatusBounds : (sing_at != 255 ? kStatusSingular : 0);
    TL(2, PUBLISH && b == 0 && tid == 0);                               // chain: loop end
    if (have) phase(false, 0, false, 0);                                   // what the last correction leaves: its plan entry
#ifdef NUSLAM_CHAIN_CLOCK
    if (wave == 3) {                                                    // shader cycles per 100 MHz tick over the loop: the clock the CU ran at
        ck[2] = (long long)clock64() - sclk0;
        ck[3] = (long long)wall_clock64() - wclk0;
    }
    if (lane == 0 && b == 0)
        for (int k = 0; k < 8; ++k) g_chain_clock[wave * 8 + k] = ck[k];
#endif
    if (PUBLISH) {                                                      // the whole round is complete
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) st_agent(pub.flag + kPubWords * b, (int)((unsigned)pub.base + (unsigned)J));
    }
    if (tid == 0) {
        int* co = v.c_out + b * C_WORDS;
        co[C_SEEN] = seen; co[C_SEEN_CACHED] = cached; co[C_BRK] = brk; co[C_STATUS] = status;
        if (ctrl_out4) { int* c4 = ctrl_out4 + 4 * b; c4[0] = seen; c4[1] = cached; c4[2] = brk; c4[3] = status; }
    }
    if (done_cnt) {                                                     // overlapped runs: the plan is complete
        __syncthreads();
        if (tid == 0) tick_signal(done_cnt);
    }
    TL(3, PUBLISH && b == 0 && tid == 0);                               // chain: exit
}

template <typename T, bool FUSED>
__global__ __launch_bounds__(256) void k_tick_chain(View v, TickObs o, int total_landmarks, const T* __restrict__ P,
                                                    TickStep* __restrict__ plan, TickCarry cy,
                                                    int* __restrict__ ctrl_out4, int* __restrict__ done_cnt)
{
    tick_chain<T, FUSED, false>(blockIdx.x, v, o, total_landmarks, P, plan, cy, ctrl_out4, done_cnt, TickPublish{});
}

// ------------------------------------------------------------------------------------------------ overlapped runs: prep
// behind predict(t) on the handle's stream: the position map of U' (posmap[i] = first position of state index i in U',
// else -1; the entries set for the previous tick's map -- U_t -- are cleared first) and the block P(U'[p], U'[q]).
// One more workgroup (blockIdx.x == NU + 1, filter 0) waits for the chain's plan of THIS tick, so that the strips can be
// enqueued right behind this kernel without a launch of their own for the wait.
template <typename T>
__global__ __launch_bounds__(64) void k_tick_prep(View v, TickObs ot, TickObs on, const T* __restrict__ P,
                                                  int* __restrict__ posmap, double* __restrict__ blk,
                                                  const int* __restrict__ wait_cnt, int wait_target, int* __restrict__ timeouts)
{
    constexpr int NU = kTickNU;
    const int b = blockIdx.y, q = blockIdx.x, p = threadIdx.x;
    if (q == NU + 1) {
        if (b == 0 && !tick_wait(wait_cnt, wait_target) && p == 0) atomicAdd(timeouts, 1);
        return;
    }
    auto index_of = [&](const TickObs& o, int pos) {
        if (pos < 3) return pos;
        const int st = (pos - 3) >> 1;
        int id = 0;
        if (st < o.J) id = o.ids ? o.ids[b * o.stride + o.off + st] : o.id0[st];
        return ((id >= 1 && id <= v.n) ? 3 + 2 * (id - 1) : 3) + ((pos - 3) & 1);
    };
    if (q < NU) {
        if (p < NU) blk[((size_t)b * NU + p) * NU + q] = (double)P[(size_t)b * v.p_stride + (size_t)index_of(on, q) * v.ld + index_of(on, p)];
        return;
    }
    int* pm = posmap + (size_t)b * v.ld;                                // the last workgroup of the row: the map
    if (p < NU) pm[index_of(ot, p)] = -1;
    __syncthreads();
    if (p == 0)
        for (int k = NU - 1; k >= 0; --k) pm[index_of(on, k)] = k;      // descending: the FIRST position of an index stays
}

// ------------------------------------------------------------------------------------------------ the panels
// 512 threads = 64 state indices x 2 roles x 4 lanes: the four lanes of a QUAD share one index t and split the panel
// positions p = 4 j + k between them (k = lane & 3).  Role 0 (waves 0-3) carries COLUMN t of the row panel
// (RP[p] = P(U[p], t)), role 1 (waves 4-7) ROW t of the column panel (CP[p] = P(t, U[p])) and state entry t.
// Why quads: with one lane per index a wave had 35 dependent coefficient reads per correction and nothing to hide their
// latency behind (60 % of the wave's cycles parked on lgkmcnt, 2 us per correction); a quarter of the rows per lane and
// eight waves per workgroup bring it to ~0.3 us.  The five entries every lane of the quad needs (rows / columns
// set_s) are broadcast inside the quad by DPP.  The plan of the round (61 KB) is staged in LDS once; only the panel
// entries a later correction still reads are carried (positions 0, 1, 2 and those of LATER markers).
__device__ inline double quad_bcast(double x, int k)     // lane k of every quad's value to its four lanes; k is a literal
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    switch (k) {                                          // (the DPP control is an immediate)
    case 0: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x00, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x00, 0xf, 0xf, false); break;
    case 1: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x55, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x55, 0xf, 0xf, false); break;
    case 2: lo = __builtin_amdgcn_update_dpp(lo, lo, 0xaa, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xaa, 0xf, 0xf, false); break;
    default: lo = __builtin_amdgcn_update_dpp(lo, lo, 0xff, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xff, 0xf, 0xf, false); break;
    }
    return __hiloint2double(hi, lo);
}

constexpr int kQuadRows = (kTickNU + 3) / 4;              // panel positions per lane
#ifdef NUSLAM_CHAIN_CLOCK
__device__ long long g_panels_clock[2][20];               // debug builds: per role, 100 MHz ticks: prologue, 16 corrections, tail
#define PCK(r, k) do { if (blockIdx.x == 1 && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) & 1) == 0) { \
        const long long n__ = (long long)wall_clock64(); g_panels_clock[r][k] = n__ - pct; pct = n__; } } while (0)
#else
#define PCK(r, k) do { } while (0)
#endif

// IDX = state indices per workgroup (64: 8 waves; 32: 4 waves, one per SIMD -- the kernel is VALU-issue-bound, so a single
// filter, whose 2003 indices fill only a fraction of the chip anyway, takes the smaller groups on twice the CUs).
template <typename T, int IDX>
__global__ __launch_bounds__(IDX * 8) __attribute__((amdgpu_waves_per_eu(1, IDX == 32 ? 2 : 4))) void k_tick_panels(View v, TickObs o, const T* __restrict__ P,
                                                     const TickStep* __restrict__ plan, double* __restrict__ Kbuf,
                                                     double* __restrict__ Rbuf, double* __restrict__ Vbuf, const int* __restrict__ posmap,
                                                     double* __restrict__ KU, double* __restrict__ RU, double* __restrict__ SU, int rank_ok)
{
    // rank_ok (the pass is the rank-2m one): a filter whose round holds no first sighting (round_flags: the predicate the pass itself
    // evaluates) carries its panels in RANK form, P -= K_s V_s -- two FMAs per entry and correction instead of the exact chain's seven,
    // a quarter of the plan staged (the entries' heads) -- as the pass will apply the round to the covariance
    // Vbuf != null: also V_s = H_s R_s (2 x len, slam_library.cpp:279 written as P - K_s (H_s P)), the second factor of the
    // rank-2m pass (k_tick_rank)
    // posmap != null (overlapped runs): the strips at the NEXT tick's index set are also dropped into the compact arrays
    // KU [J][2][NU], RU [J][5][NU], SU [NU] (state after the round) for the next tick's chain (tick_carry)
    constexpr int NU = kTickNU;
    const int b = blockIdx.y;
    const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x / (IDX * 4));
    const int k = threadIdx.x & 3;                                      // which quarter of the positions
    const int t = blockIdx.x * IDX + ((threadIdx.x % (IDX * 4)) >> 2);
    const int ld = v.ld, L = v.L;
    const T* Pb = P + (size_t)b * v.p_stride;
    const int J = o.J;
    double* Kb = Kbuf + (size_t)b * kTickJ * 2 * ld;
    double* Rb = Rbuf ? Rbuf + (size_t)b * kTickJ * 5 * ld : nullptr;       // (null: nobody reads the R strips of this round)

#ifdef NUSLAM_CHAIN_CLOCK
    long long pct = (long long)wall_clock64();
#endif
    extern __shared__ double plan_l[];                                  // [J] TickStep, or [J] heads of kPlanHeadWords words (rank form)
    // (rank_ok == 2: the host has proven the round free of first sightings in every filter -- no look at the flags, and the launch
    // provides LDS for the heads only: twice the workgroups per CU for batches)
    bool rankp = rank_ok == 2;
    if (rank_ok == 1) {
        unsigned am;
        bool any_init;
        round_flags(plan + (size_t)b * kTickJ, J, am, any_init);
        rankp = !any_init;
    }
    const int ent_words = rankp ? kPlanHeadWords : kPlanExactWords;             // an entry's stride in LDS
    {
        Pack16<double>* dst = reinterpret_cast<Pack16<double>*>(plan_l);
        PCK(0, 17);
        // (4.6 us of this kernel's ~18 at N = 1000 in the exact form: a CU moves a 1 KB wave-load in ~0.25 us here; copying only
        // the rows that are read -- half of them -- through predicated pieces cost more in address arithmetic than it saved)
        if (rankp) {
            constexpr int per16 = kPlanHeadWords / 2;                   // (compile-time divisors): prefix, then the KV rows from the entry's end
            for (int e = threadIdx.x; e < J * per16; e += IDX * 8) {
                const int st = e / per16, w = e % per16;
                dst[e] = reinterpret_cast<const Pack16<double>*>(plan + (size_t)b * kTickJ + st)[w < kPlanPrefixWords / 2 ? w : w + (kPlanExactWords - kPlanPrefixWords) / 2];
            }
        } else {
            constexpr int per16 = kPlanExactWords / 2;                  // prefix + MP + BR: the entry without its KV rows
            for (int e = threadIdx.x; e < J * per16; e += IDX * 8) {
                const int st = e / per16, w = e % per16;
                dst[e] = reinterpret_cast<const Pack16<double>*>(plan + (size_t)b * kTickJ + st)[w];
            }
        }
    }
    auto entry = [&](int st) { return reinterpret_cast<const TickStep*>(plan_l + (size_t)st * ent_words); };
    PCK(0, 18);
    // this lane's positions p = 4 j + k and the state indices behind them
    int Uk[kQuadRows];
#pragma unroll
    for (int j = 0; j < kQuadRows; ++j) {
        const int p = 4 * j + k;
        int u = 3;
        if (p < 3) u = p;
        else if (p < NU) {
            const int st = (p - 3) >> 1;
            int id = 0;
            if (st < J) id = o.ids ? o.ids[b * o.stride + o.off + st] : o.id0[st];
            u = ((id >= 1 && id <= v.n) ? 3 + 2 * (id - 1) : 3) + ((p - 3) & 1);
        }
        Uk[j] = u;
    }

    // (the two forms as two instantiations of the same text: a run-time switch inside the sixteen unrolled corrections would cut them
    // into basic blocks the scheduler cannot move LDS reads across)
    auto body = [&](auto rk_) {
    constexpr bool RANKP = decltype(rk_)::value;
    if (role == 0) {
        // ---- column t of the five-row strips R_s and of the row panel
        const bool live = t < L;
        const int pm = (posmap && live && k == 0) ? posmap[(size_t)b * ld + t] : -1;
        // Stores without branches: a lane that owns no column / no compact slot stores into a dump area behind the buffer.
        // (An `if` around the stores ends the basic block: with sixteen unrolled corrections the scheduler could then not
        // move a correction's LDS reads of the plan above the previous correction's arithmetic, and every correction
        // started with exposed LDS round trips -- one wave per SIMD here, nothing else to run meanwhile.)
        double* const rdump = v.dump + (threadIdx.x & (kTickDump - 1));
        double* const rdst = (Rb && live && k == 0) ? Rb + t : rdump;
        const size_t rstep = (Rb && live && k == 0) ? (size_t)ld : 0;
        double* const vdst = (Vbuf && live && k == 0) ? Vbuf + (size_t)b * kTickJ * 2 * ld + t : rdump;
        const size_t vstep = (Vbuf && live && k == 0) ? (size_t)ld : 0;
        double* const rudump = RU ? RU + (size_t)v.B * kTickJ * 5 * NU + (threadIdx.x & (kTickDump - 1)) : nullptr;
        double* const rudst = pm >= 0 ? RU + (size_t)b * kTickJ * 5 * NU + pm : rudump;
        const size_t rustep = pm >= 0 ? (size_t)NU : 0;
        const T* col = Pb + (size_t)(live ? t : 0) * ld;
        double RP[kQuadRows];
#pragma unroll
        for (int j = 0; j < kQuadRows; ++j) RP[j] = (double)col[Uk[j]];
        PCK(0, 19);
        __syncthreads();
        PCK(0, 0);
        // which corrections change P: one LDS read and a ballot instead of a flag read (and its latency) at the top of
        // every correction; the corrections themselves run WITHOUT a branch around them -- a skipped one computes on its
        // (stale) plan entry and keeps nothing -- so that the sixteen of them are one basic block to the scheduler
        const unsigned actmask = (unsigned)__ballot((int)(threadIdx.x & 63) < J && entry((threadIdx.x & 63) < kTickJ ? (threadIdx.x & 63) : 0)->skip == 0) & 0xffffu;
#pragma unroll
        for (int st = 0; st < kTickJ; ++st) {             // (no break / continue: the loop must unroll, RP is indexed by st)
            const TickStep* ps = entry(st < J ? st : 0);
            PCK(0, 1 + st);
            const bool act = (actmask >> st) & 1u;
            if (IDX == 32 || act) {                       // (IDX == 64, batches: a branch per correction keeps the register count at 110)
                const int pos = 3 + 2 * st;
                // P_{s-1}(set_s[q], t): positions 0, 1, 2, pos, pos + 1 live in lanes 0, 1, 2, pos & 3, (pos + 1) & 3
                const double rs[5] = { quad_bcast(RP[0], 0), quad_bcast(RP[0], 1), quad_bcast(RP[0], 2),
                                       quad_bcast(RP[pos >> 2], pos & 3), quad_bcast(RP[(pos + 1) >> 2], (pos + 1) & 3) };
                double* const rd = act ? rdst : rdump;
                const size_t rsp = act ? rstep : 0;
#pragma unroll
                for (int q = 0; q < 5; ++q) rd[(size_t)(st * 5 + q) * rsp] = rs[q];
                double V0 = 0.0, V1 = 0.0;                              // V_s(r, t) = sum_q H_s(r, set[q]) R_s(q, t)
                if (Vbuf || RANKP) { V0 = hp_entry(ps->Hc, rs, 0); V1 = hp_entry(ps->Hc, rs, 1); }
                if (Vbuf) {                                             // (uniform)
                    double* const vd = act ? vdst : rdump;
                    const size_t vsp = act ? vstep : 0;
                    vd[(size_t)(st * 2 + 0) * vsp] = V0;
                    vd[(size_t)(st * 2 + 1) * vsp] = V1;
                }
                if (posmap) {                                           // (uniform)
                    double* const rud = act ? rudst : rudump;
                    const size_t rusp = act ? rustep : 0;
#pragma unroll
                    for (int q = 0; q < 5; ++q) rud[(size_t)(st * 5 + q) * rusp] = rs[q];
                }
#pragma unroll
                for (int j = 0; j < kQuadRows; ++j) {
                    if (4 * j + 3 < 3 + 0 || (4 * j >= 3 && 4 * j + 3 < pos + 2)) continue;      // no lane of the quad is live here
                    const int p = 4 * j + k;
                    const bool on = p < NU && !(p >= 3 && p < pos + 2);  // rows of this and earlier markers are not read again
                    const int pc = on ? p : 0;
                    if constexpr (RANKP) {                                        // (uniform) P(U[p], t) -= K_s(U[p], :) V_s(:, t), the pass's own sum
                        const Pack16<double> kk = *reinterpret_cast<const Pack16<double>*>(&plan_kv(ps)[pc][0]);
                        const double nv = fma(V1, -kk.v[1], fma(V0, -kk.v[0], RP[j]));
                        RP[j] = (on && act) ? nv : RP[j];
                        continue;
                    }
                    const Pack16<double> m01 = *reinterpret_cast<const Pack16<double>*>(&ps->MP[pc][0]);
                    const Pack16<double> m23 = *reinterpret_cast<const Pack16<double>*>(&ps->MP[pc][2]);
                    const Pack16<double> m45 = *reinterpret_cast<const Pack16<double>*>(&ps->MP[pc][4]);
                    const double m6 = ps->MP[pc][6];
                    const double m[5] = { m01.v[0], m01.v[1], m23.v[0], m23.v[1], m45.v[0] };
                    const double nv = p1_entry<T>(m, rs, RP[j], m45.v[1], m6);
                    RP[j] = (on && act) ? nv : RP[j];
                }
            }
        }
    } else {
        // ---- row t of the gains K_s and of the column panel; state entry t
        const bool live = t < ld;
        const int tr = live ? t : 0;
        const int pm = (posmap && live && k == 0) ? posmap[(size_t)b * ld + t] : -1;
        double* const kdump = v.dump + (threadIdx.x & (kTickDump - 1));
        double* const kdst = (live && k == 0) ? Kb + t : kdump;
        const size_t kstep = (live && k == 0) ? (size_t)ld : 0;
        double* const kudump = KU ? KU + (size_t)v.B * kTickJ * 2 * NU + (threadIdx.x & (kTickDump - 1)) : nullptr;
        double* const kudst = pm >= 0 ? KU + (size_t)b * kTickJ * 2 * NU + pm : kudump;
        const size_t kustep = pm >= 0 ? (size_t)NU : 0;
        double CP[kQuadRows];
#pragma unroll
        for (int j = 0; j < kQuadRows; ++j) CP[j] = (double)Pb[(size_t)Uk[j] * ld + tr];
        double sv = v.s_in[(size_t)b * ld + tr];
        __syncthreads();
        PCK(1, 0);
        const int l16 = (threadIdx.x & 63) < kTickJ ? (threadIdx.x & 63) : 0;
        const unsigned actmask = (unsigned)__ballot((int)(threadIdx.x & 63) < J && entry(l16)->skip == 0) & 0xffffu;
        const unsigned initmask = (unsigned)__ballot((int)(threadIdx.x & 63) < J && (entry(l16)->init & 1) != 0) & 0xffffu;
        const int c_lane = entry(l16)->c;                               // correction (lane & 15)'s landmark index
#pragma unroll
        for (int st = 0; st < kTickJ; ++st) {
            const TickStep* ps = entry(st < J ? st : 0);
            PCK(1, 1 + st);
            const int c = __builtin_amdgcn_readlane(c_lane, st);
            const bool act = (actmask >> st) & 1u;
            const bool init = (initmask >> st) & 1u;
            if (st < J && !act && init) {                               // the landmark was initialised before update() threw
                if (t == c) sv = ps->lxy[0];
                if (t == c + 1) sv = ps->lxy[1];
            }
            if (IDX == 32 || act) {
                const int pos = 3 + 2 * st;
                const int setv[5] = { 0, 1, 2, c, c + 1 };
                const double pc[5] = { quad_bcast(CP[0], 0), quad_bcast(CP[0], 1), quad_bcast(CP[0], 2),
                                       quad_bcast(CP[pos >> 2], pos & 3), quad_bcast(CP[(pos + 1) >> 2], (pos + 1) & 3) };
                double Hc[10], Si[4], K[2], m[5];
#pragma unroll
                for (int q = 0; q < 10; ++q) Hc[q] = ps->Hc[q];
#pragma unroll
                for (int q = 0; q < 4; ++q) Si[q] = ps->Sinv[q];
                gain_row(pc, Hc, Si, t, setv, K, m);
                double* const kd = act ? kdst : kdump;
                const size_t ksp = act ? kstep : 0;
                kd[(size_t)(st * 2 + 0) * ksp] = K[0];
                kd[(size_t)(st * 2 + 1) * ksp] = K[1];
                if (posmap) {                                           // (uniform)
                    double* const kud = act ? kudst : kudump;
                    const size_t kusp = act ? kustep : 0;
                    kud[(size_t)(st * 2 + 0) * kusp] = K[0];
                    kud[(size_t)(st * 2 + 1) * kusp] = K[1];
                }
                const double bef = (t > 2 && t < c) ? 1.0 : 0.0, aft = (t > c + 1) ? 1.0 : 0.0;
                double s0 = (init && t == c) ? ps->lxy[0] : (init && t == c + 1) ? ps->lxy[1] : sv;
                double acc = 0.0;
                acc = fma(K[0], ps->dz[0], acc);
                acc = fma(K[1], ps->dz[1], acc);
                s0 = s0 + acc;                                          // :275
                // :276 -- the chain formed exactly this sum for the heading and wrapped it (one wave there instead of a
                // 450-instruction straggler here, sixteen times)
                if (t == 0) s0 = ps->heading;
                sv = act ? s0 : sv;
#pragma unroll
                for (int j = 0; j < kQuadRows; ++j) {
                    if (4 * j >= 3 && 4 * j + 3 < pos + 2) continue;    // no lane of the quad is live here
                    const int p = 4 * j + k;
                    const bool on = p < NU && !(p >= 3 && p < pos + 2);  // columns of this and earlier markers are not read again
                    const int pcx = on ? p : 0;
                    if constexpr (RANKP) {                                        // (uniform) P(t, U[p]) -= K_s(t, :) V_s(:, U[p])
                        const Pack16<double> vv = *reinterpret_cast<const Pack16<double>*>(&plan_kv(ps)[pcx][2]);
                        const double nv = fma(vv.v[1], -K[1], fma(vv.v[0], -K[0], CP[j]));
                        CP[j] = (on && act) ? nv : CP[j];
                        continue;
                    }
                    const Pack16<double> r01 = *reinterpret_cast<const Pack16<double>*>(&ps->BR[pcx][0]);
                    const Pack16<double> r23 = *reinterpret_cast<const Pack16<double>*>(&ps->BR[pcx][2]);
                    const double r4 = ps->BR[pcx][4];
                    const double r[5] = { r01.v[0], r01.v[1], r23.v[0], r23.v[1], r4 };
                    const double nv = p1_entry<T>(m, r, CP[j], bef, aft);
                    CP[j] = (on && act) ? nv : CP[j];
                }
            }
        }
        if (live && k == 0) v.s_out[(size_t)b * ld + t] = sv;
        if (pm >= 0) SU[(size_t)b * NU + pm] = sv;
    }
    };
    if (rankp) body(std::true_type{});
    else body(std::false_type{});
}

// ------------------------------------------------------------------------------------------------ strips of a batch, rank form
// Many filters, a round the host has proven free of first sightings (rank_ok == 2 above), nobody reading the R strips: ONE LANE per
// (state index, role) instead of a quad, and no LDS.  What a correction needs besides the lane's own 35 panel entries -- H_s, S_s^-1,
// the K / V rows of the index set -- is the same for every lane of a wave (one filter per workgroup): it comes through the SCALAR cache
// (the plan read through the constant address space) and enters the FMAs as SGPR operands: per index and correction 10 + 2 x |live
// positions| FMAs -- k_tick_panels<T, 64> spends ten DPP moves, nine 16-byte LDS reads and eighteen selects per LANE of the quad on the
// same work.  185 us against 287 at 1024 x N = 200 (DESIGN 3c has the breakdown: the kernel moves ~0.8 GB -- the row panel is read one
// 8-byte entry per 64-byte line of a column-major matrix -- and that, not the arithmetic, is what is left).  Latency is hidden by
// occupancy (5 waves per SIMD), not by scheduling across corrections, so the corrections may branch.
// Same sums in the same order as k_tick_panels' rank form: same bits (tests/test_gpu_rank.py).
#include "ekf_strips_blocks.inc"
// The K / V rows of the index set enter the panel FMAs as scalar operands, a BLOCK of up to 15 positions per scalar round trip
// (ekf_strips_blocks.inc, written by tools/gen_strips_asm.py: s_load_dwordx4 per position into a clobbered SGPR window, one wait, two FMAs
// per position).  Positions 0..2 are always live; of 3..34 those of this and earlier markers ([3, 5 + 2 st)) are dead -- never read
// again --: runs [3..14], [15..26], [27..34] are dropped once entirely dead, a partly dead run is computed whole (its dead entries take
// values nobody reads).  ROLE picks the half of a KV row (K: bytes 0..15, V: bytes 16..31) and the operand order of the FMAs.
#define LANE_PANEL_UPDATE(ROLE, R, st, ps_addr, a, b) do { \
        const unsigned long long kv__ = (ps_addr) + offsetof(TickStep, KV) + 16 * (ROLE); \
        if ((st) <= 4) { \
            LANE_BLOCK_A12_R##ROLE(R, 3, a, b, kv__, kv__ + 32 * 3); \
            LANE_BLOCK_B12_R##ROLE(R, 15, a, b, kv__, kv__ + 32 * 15); \
            LANE_BLOCK_B8_R##ROLE(R, 27, a, b, kv__, kv__ + 32 * 27); \
        } else if ((st) <= 10) { \
            LANE_BLOCK_A12_R##ROLE(R, 15, a, b, kv__, kv__ + 32 * 15); \
            LANE_BLOCK_B8_R##ROLE(R, 27, a, b, kv__, kv__ + 32 * 27); \
        } else if ((st) <= 14) { \
            LANE_BLOCK_A8_R##ROLE(R, 27, a, b, kv__, kv__ + 32 * 27); \
        } else { \
            LANE_BLOCK_A0_R##ROLE(R, 0, a, b, kv__, kv__); \
        } \
    } while (0)
template <typename T>
__global__ __launch_bounds__(128) void k_tick_strips_lane(View v, TickObs o, const T* __restrict__ P, const TickStep* __restrict__ plan,
                                                          double* __restrict__ Kbuf, double* __restrict__ Vbuf)
{
    constexpr int NU = kTickNU;
    typedef const __attribute__((address_space(4))) TickStep* ConstPlan;
    // workgroup -> (filter, group of 64 indices): consecutive workgroups go to the eight XCDs in turn, so XCD x takes filters x, x + 8, ...
    // and the groups of one filter follow each other on ONE XCD -- its plan is fetched into one L2, once
    const int ld = v.ld, L = v.L, J = o.J;
    const int G = (ld + 63) >> 6;
    const int slot = (int)blockIdx.x >> 3;
    const int b = ((int)blockIdx.x & 7) + 8 * (slot / G);
    if (b >= v.B) return;
    const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int t = (slot % G) * 64 + ((int)threadIdx.x & 63);
    const T* Pb = P + (size_t)b * v.p_stride;
    ConstPlan pl = (ConstPlan)(unsigned long long)(plan + (size_t)b * kTickJ);
    const unsigned long long plan_addr = (unsigned long long)(plan + (size_t)b * kTickJ);
    // the scalar loads below come one correction at a time and would each wait for HBM (the plan is 94 MB at 1024 filters, written by the
    // chain kernel before): every line this wave is going to read is touched here, all at once, and is in the L2 by the time it is asked for
    {
        const int lane = (int)threadIdx.x & 63;
        const char* base = reinterpret_cast<const char*>(plan + (size_t)b * kTickJ);
        const size_t kv0 = offsetof(TickStep, KV);
        // lanes 0..3: the prefix (176 bytes); lanes 4..23: the KV rows (1120 bytes), 64 bytes apart
        const size_t off = lane < 4 ? (size_t)lane * 64 : kv0 + (size_t)(lane - 4) * 64;
        const bool on = lane < 4 ? off < (size_t)kPlanPrefixWords * 8 : (lane < 24 && off < sizeof(TickStep));
        for (int st = 0; st < J; ++st) {
            if (on) {
                const double x = *reinterpret_cast<const double*>(base + (size_t)st * sizeof(TickStep) + (off & ~(size_t)7));
                asm volatile("" :: "v"(x));
            }
        }
    }
    // the index set of the round (wave-uniform): lane s fetches marker s's id -- ONE load --, the lanes' values go to scalars
    int U[NU];
    {
        const int lane = (int)threadIdx.x & 63;
        int id = 0;
        if (o.ids) {
            if (lane < J) id = o.ids[b * o.stride + o.off + lane];
        } else {
#pragma unroll
            for (int st = 0; st < kTickJ; ++st) id = (lane == st && st < J) ? o.id0[st] : id;
        }
        const int cl = (id >= 1 && id <= v.n) ? 3 + 2 * (id - 1) : 3;
#pragma unroll
        for (int p = 0; p < NU; ++p) U[p] = p < 3 ? p : __builtin_amdgcn_readlane(cl, (p - 3) >> 1) + ((p - 3) & 1);
    }
    if (role == 0) {
        // ---- column t of the row panel; V_s(:, t)
        const bool live = t < L;
        const T* col = Pb + (size_t)(live ? t : 0) * ld;
        double* const Vb = Vbuf + (size_t)b * kTickJ * 2 * ld + t;
        double RP[NU];
#pragma unroll
        for (int p = 0; p < NU; ++p) RP[p] = (double)col[U[p]];
        // (the panel has ARRIVED before the corrections begin, and the compiler knows it: left to its own bookkeeping across the branches
        // below it waits for vmcnt(0) -- the correction's own stores -- at the top of every correction)
#pragma unroll
        for (int p = 0; p < NU; ++p) asm volatile("" :: "v"(RP[p]));
#pragma unroll
        for (int st = 0; st < kTickJ; ++st) {
            ConstPlan ps = pl + (st < J ? st : 0);
            if (st < J && ps->skip == 0) {
            const int pos = 3 + 2 * st;
            const double rs[5] = { RP[0], RP[1], RP[2], RP[pos], RP[pos + 1] };
            double Hc[10];
#pragma unroll
            for (int q = 0; q < 10; ++q) Hc[q] = ps->Hc[q];
            const double V0 = hp_entry(Hc, rs, 0), V1 = hp_entry(Hc, rs, 1);
            if (live) {
                __builtin_nontemporal_store(V0, &Vb[(size_t)(st * 2 + 0) * ld]);
                __builtin_nontemporal_store(V1, &Vb[(size_t)(st * 2 + 1) * ld]);
            }
            // RP[p] = fma(V1, -K_s(U[p], 1), fma(V0, -K_s(U[p], 0), RP[p])) for the live p
            LANE_PANEL_UPDATE(0, RP, st, plan_addr + (unsigned long long)st * sizeof(TickStep), V0, V1);
            }
        }
    } else {
        // ---- row t of the column panel; K_s(t, :); state entry t
        const bool live = t < ld;
        const int tr = live ? t : 0;
        double* const Kb = Kbuf + (size_t)b * kTickJ * 2 * ld + tr;
        double CP[NU];
#pragma unroll
        for (int p = 0; p < NU; ++p) CP[p] = (double)Pb[(size_t)U[p] * ld + tr];
        double sv = v.s_in[(size_t)b * ld + tr];
#pragma unroll
        for (int p = 0; p < NU; ++p) asm volatile("" :: "v"(CP[p]));
        asm volatile("" :: "v"(sv));
#pragma unroll
        for (int st = 0; st < kTickJ; ++st) {
            ConstPlan ps = pl + (st < J ? st : 0);
            const int c = ps->c;
            const bool act = st < J && ps->skip == 0, init = (ps->init & 1) != 0;
            // (the state's scalars come with the entry's head, unconditionally, and enter through selects: a branch around them made every
            // correction wait for its own stores)
            const double lx = ps->lxy[0], ly = ps->lxy[1], head = ps->heading, dz0 = ps->dz[0], dz1 = ps->dz[1];
            const bool at0 = init && t == c, at1 = init && t == c + 1;
            if (st < J && !act) {                                       // the landmark was initialised before update() threw
                sv = at0 ? lx : sv;
                sv = at1 ? ly : sv;
            }
            if (act) {                                                  // (uniform)
            const int pos = 3 + 2 * st;
            const int setv[5] = { 0, 1, 2, c, c + 1 };
            const double pc[5] = { CP[0], CP[1], CP[2], CP[pos], CP[pos + 1] };
            double Hc[10], Si[4], K[2], m[5];
#pragma unroll
            for (int q = 0; q < 10; ++q) Hc[q] = ps->Hc[q];
#pragma unroll
            for (int q = 0; q < 4; ++q) Si[q] = ps->Sinv[q];
            gain_row(pc, Hc, Si, t, setv, K, m);
            if (live) {
                __builtin_nontemporal_store(K[0], &Kb[(size_t)(st * 2 + 0) * ld]);
                __builtin_nontemporal_store(K[1], &Kb[(size_t)(st * 2 + 1) * ld]);
            }
            double s0 = at0 ? lx : at1 ? ly : sv;
            double acc = 0.0;
            acc = fma(K[0], dz0, acc);
            acc = fma(K[1], dz1, acc);
            s0 = s0 + acc;                                              // :275
            sv = t == 0 ? head : s0;                                    // :276 (wrapped by the chain)
            // CP[p] = fma(V_s(1, U[p]), -K[1], fma(V_s(0, U[p]), -K[0], CP[p])) for the live p
            LANE_PANEL_UPDATE(1, CP, st, plan_addr + (unsigned long long)st * sizeof(TickStep), K[0], K[1]);
            }
        }
        if (live) v.s_out[(size_t)b * ld + t] = sv;
    }
}

// ------------------------------------------------------------------------------------------------ chain and strips in one launch
// One filter (or few): the strips need nothing of correction s but plan entry s, and a correction of the chain takes ~2.2 us
// where the strips' share of it takes ~0.6.  k_tick_front runs the chain on workgroup 0 and the strip workgroups BESIDE it
// in the same launch: they pick each plan entry up as the chain announces it (TickPublish; agent-scope stores and loads,
// no fences) and have the round's K / R / V strips and the new state ready ~one correction after the chain ends, instead of
// 17 us (N = 1000) after it as a kernel of their own.  The whole grid is resident at once (the host launches this form
// only while 1 + the strip workgroups fit the chip), the chain is block 0 -- dispatched first --, every wait is bounded
// and an expired one is reported through `timeouts` (NUSLAM_E_SYNC).  Same arithmetic as k_tick_panels<T, 32>, entry by
// entry: same bits.
constexpr int kPlanWords = (int)(sizeof(TickStep) / 8);

// TAGGED (k_tick_fused only): rank-form strips whose K / V go out as tagged words (tg) instead of into Kbuf / Vbuf
template <typename T, bool TAGGED = false, bool RUN = false>
__device__ inline void tick_panels_stream(const int b, const int wg, View v, TickObs o, const T* __restrict__ P,
                                          const TickStep* __restrict__ plan, double* __restrict__ Kbuf,
                                          double* __restrict__ Rbuf, double* __restrict__ Vbuf, TickPublish pub,
                                          int* __restrict__ timeouts, const int* __restrict__ posmap = nullptr,
                                          double* __restrict__ KU = nullptr, double* __restrict__ RU = nullptr,
                                          double* __restrict__ SU = nullptr, TickTagged tg = TickTagged{}, TickRun rn = TickRun{})
{
    const long long ooff = RUN ? o.off + rn.toff : o.off;
    // posmap != null (streamed overlapped runs, k_tick_strips): the strips at the NEXT tick's index set are also dropped into the
    // compact arrays KU, RU, SU for that tick's chain (tick_carry), as k_tick_panels does
    constexpr int NU = kTickNU, IDX = 32;
    const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x / (IDX * 4));
    const int k = threadIdx.x & 3;
    const int t = wg * IDX + ((threadIdx.x % (IDX * 4)) >> 2);
    const int ld = v.ld, L = v.L;
    const T* Pb = P + (size_t)b * v.p_stride;
    const int J = o.J;
    double* Kb = Kbuf + (size_t)b * kTickJ * 2 * ld;
    double* Rb = Rbuf + (size_t)b * kTickJ * 5 * ld;
    __shared__ long long slot[2][kPlanWords];                           // the plan entry being applied / the next one
    __shared__ int ok_sh, fail_sh;
    if (threadIdx.x == 0) fail_sh = 0;
    __syncthreads();

    int Uk[kQuadRows];
#pragma unroll
    for (int j = 0; j < kQuadRows; ++j) {
        const int p = 4 * j + k;
        int u = 3;
        if (p < 3) u = p;
        else if (p < NU) {
            const int st = (p - 3) >> 1;
            int id = 0;
            if (st < J) id = o.ids ? o.ids[b * o.stride + ooff + st] : o.id0[st];
            u = ((id >= 1 && id <= v.n) ? 3 + 2 * (id - 1) : 3) + ((p - 3) & 1);
        }
        Uk[j] = u;
    }
    // this lane's panel entries and where its strips go (role 0: column t of the row panel, R and V; role 1: row t of the
    // column panel, K, the state entry)
    const bool live = role == 0 ? t < L : t < ld;
    const int tr = live ? t : 0;
    const bool owner = live && k == 0;
    double* const dump = v.dump + (threadIdx.x & (kTickDump - 1));
    const int pm = (posmap && owner) ? posmap[(size_t)b * ld + t] : -1;
    double* const cu_dst = pm < 0 ? dump : role == 0 ? RU + (size_t)b * kTickJ * 5 * NU + pm : KU + (size_t)b * kTickJ * 2 * NU + pm;
    const size_t cu_step = pm >= 0 ? (size_t)NU : 0;
    double PN[kQuadRows];
    double sv = 0.0;
    bool failed = false;
    if (pub.predict) {
        // the tick's predict is being applied by workgroups of this launch: wait for all of them, then read what they wrote
        // (rows / columns 1, 2 of P, the advanced state in s_out) with agent-scope loads -- every load of handed-off bytes
        if (threadIdx.x == 0) {
            int ok = 0;
            for (int it = 0; it < (1 << 18); ++it) {
                if (seq_reached(ld_agent(pub.flag + kPubWords * b + 2), pub.pbase)) { ok = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            ok_sh = ok;
        }
        __syncthreads();
        if (!ok_sh) {
            failed = true;
            if (threadIdx.x == 0) atomicAdd(timeouts, 1);
        }
        __syncthreads();                                                // (ok_sh is rewritten by the entry loop)
#pragma unroll
        for (int j = 0; j < kQuadRows; ++j)
            PN[j] = role == 0 ? (double)ld_agent(&Pb[(size_t)tr * ld + Uk[j]]) : (double)ld_agent(&Pb[(size_t)Uk[j] * ld + tr]);
        if (role == 1) sv = ld_agent(&v.s_out[(size_t)b * ld + tr]);
    } else {
#pragma unroll
        for (int j = 0; j < kQuadRows; ++j)
            PN[j] = role == 0 ? (double)Pb[(size_t)tr * ld + Uk[j]] : (double)Pb[(size_t)Uk[j] * ld + tr];
        if (role == 1) sv = v.s_in[(size_t)b * ld + tr];
    }
    const long long* src = reinterpret_cast<const long long*>(plan + (size_t)b * kTickJ);
    // Entry st+1 is FETCHED WHILE entry st is applied whenever the chain has already announced it (a workgroup that has fallen
    // behind the chain catches up at the pace of its arithmetic, ~0.6 us per entry, instead of flag + fetch + arithmetic in a
    // row): each wave looks at the flag itself and fetches its own words of the entry -- every load of handed-off bytes an
    // agent-scope load, issued behind the flag's --, one barrier per entry makes the slot whole.  Waits are bounded as before.
    const int* const flagp = pub.flag + kPubWords * b;
    constexpr int kW = (kPlanWords + 255) / 256;
    const bool rankp = pub.rank_panels != 0;                            // panels in rank form: only the entries' heads are fetched
    const int ent_words = rankp ? kPlanHeadWords : kPlanExactWords;     // (the slot holds the packed image)
    auto published = [&](int st) {
        const int f = __builtin_amdgcn_readfirstlane(ld_agent(flagp));
        return seq_reached(f, (int)((unsigned)pub.base + (unsigned)(st + 1)));
    };
    auto wait_entry = [&](int st) {
        for (int it = 0; it < (1 << 18); ++it) {
            if (published(st)) return true;
            __builtin_amdgcn_s_sleep(1);
        }
        return false;
    };
    long long rw[kW];
    auto fetch = [&](int st) {
        const long long* e = src + (size_t)st * kPlanWords;
#pragma unroll
        for (int u = 0; u < kW; ++u) {
            const int w = (int)threadIdx.x + 256 * u;
            const int wc = w < ent_words ? w : 0;
            // (rank form: < 256 words, one load; the KV rows sit at the entry's end)
            if (u == 0 || !rankp) rw[u] = ld_agent(e + (rankp && wc >= kPlanPrefixWords ? wc + (kPlanExactWords - kPlanPrefixWords) : wc));
        }
    };
    auto stash = [&](int st) {
#pragma unroll
        for (int u = 0; u < kW; ++u) {
            const int w = (int)threadIdx.x + 256 * u;
            if (w < ent_words) slot[st & 1][w] = rw[u];
        }
    };
    if (!failed && J > 0) {
        if (wait_entry(0)) { fetch(0); stash(0); }
        else if ((threadIdx.x & 63) == 0) fail_sh = 1;
    }
    __syncthreads();
    if (!failed && fail_sh != 0) {
        failed = true;
        if (threadIdx.x == 0) atomicAdd(timeouts, 1);
    }

    auto loop = [&](auto rk_) {
    constexpr bool RANKP = decltype(rk_)::value;
#pragma unroll
    for (int st = 0; st < kTickJ; ++st) {
        if (st < J && !failed) {                                        // (uniform)
            const TickStep* ps = reinterpret_cast<const TickStep*>(slot[st & 1]);
            bool pre = false;
            if (st + 1 < J) {
                pre = published(st + 1);
                if (pre) fetch(st + 1);
            }
            {
                const bool act = ps->skip == 0;
                const bool init = (ps->init & 1) != 0;
                const int c = ps->c;
                const int pos = 3 + 2 * st;
                if (role == 0) {
                    if (act) {
                        const double rs[5] = { quad_bcast(PN[0], 0), quad_bcast(PN[0], 1), quad_bcast(PN[0], 2),
                                               quad_bcast(PN[pos >> 2], pos & 3), quad_bcast(PN[(pos + 1) >> 2], (pos + 1) & 3) };
                        double* const rd = owner ? Rb + t : dump;
                        const size_t rsp = owner ? (size_t)ld : 0;
#pragma unroll
                        for (int q = 0; q < 5; ++q) rd[(size_t)(st * 5 + q) * rsp] = rs[q];
                        if (posmap) {                                   // (uniform)
#pragma unroll
                            for (int q = 0; q < 5; ++q) cu_dst[(size_t)(st * 5 + q) * cu_step] = rs[q];
                        }
                        double V0 = 0.0, V1 = 0.0;
                        if (Vbuf || RANKP) { V0 = hp_entry(ps->Hc, rs, 0); V1 = hp_entry(ps->Hc, rs, 1); }
                        if constexpr (TAGGED) {
                            // EVERY lane stores to its index's slot -- the four lanes of a quad hold the same value, lanes beyond the
                            // matrix index 0's -- so the wave's write-through stores merge into whole lines (with three lanes of four
                            // writing a dump area through to memory the strips took four times as long)
                            long long* const vt = tg.tagV + (((size_t)b * kTickJ * 2 + st * 2) * ld + tr) * 2;
                            st_tagged(vt, V0, tg.tag);
                            st_tagged(vt + (size_t)ld * 2, V1, tg.tag);
                        } else if (Vbuf) {
                            double* const vd = owner ? Vbuf + (size_t)b * kTickJ * 2 * ld + t : dump;
                            vd[(size_t)(st * 2 + 0) * rsp] = V0;
                            vd[(size_t)(st * 2 + 1) * rsp] = V1;
                        }
#pragma unroll
                        for (int j = 0; j < kQuadRows; ++j) {
                            if (4 * j >= 3 && 4 * j + 3 < pos + 2) continue;
                            const int p = 4 * j + k;
                            const bool on = p < NU && !(p >= 3 && p < pos + 2);
                            const int pc = on ? p : 0;
                            if constexpr (RANKP) {                                // (uniform) P(U[p], t) -= K_s(U[p], :) V_s(:, t), the pass's own sum
                                const double nv = fma(V1, -plan_kv(ps)[pc][1], fma(V0, -plan_kv(ps)[pc][0], PN[j]));
                                PN[j] = on ? nv : PN[j];
                                continue;
                            }
                            const double m[5] = { ps->MP[pc][0], ps->MP[pc][1], ps->MP[pc][2], ps->MP[pc][3], ps->MP[pc][4] };
                            const int iu = Uk[j];                       // the row of P this entry lives in
                            const double nv = p1_entry<T>(m, rs, PN[j], (iu > 2 && iu < c) ? 1.0 : 0.0, (iu > c + 1) ? 1.0 : 0.0);
                            PN[j] = on ? nv : PN[j];
                        }
                    }
                } else {
                    if (!act && init) {                                 // the landmark was initialised before update() threw
                        if (t == c) sv = ps->lxy[0];
                        if (t == c + 1) sv = ps->lxy[1];
                    }
                    if (act) {
                        const int setv[5] = { 0, 1, 2, c, c + 1 };
                        const double pc[5] = { quad_bcast(PN[0], 0), quad_bcast(PN[0], 1), quad_bcast(PN[0], 2),
                                               quad_bcast(PN[pos >> 2], pos & 3), quad_bcast(PN[(pos + 1) >> 2], (pos + 1) & 3) };
                        double Hc[10], Si[4], K[2], m[5];
#pragma unroll
                        for (int q = 0; q < 10; ++q) Hc[q] = ps->Hc[q];
#pragma unroll
                        for (int q = 0; q < 4; ++q) Si[q] = ps->Sinv[q];
                        gain_row(pc, Hc, Si, t, setv, K, m);
                        if constexpr (TAGGED) {
                            long long* const kt = tg.tagK + (((size_t)b * kTickJ * 2 + st * 2) * ld + tr) * 2;
                            st_tagged(kt, K[0], tg.tag);
                            st_tagged(kt + (size_t)ld * 2, K[1], tg.tag);
                        } else {
                            double* const kd = owner ? Kb + t : dump;
                            const size_t ksp = owner ? (size_t)ld : 0;
                            kd[(size_t)(st * 2 + 0) * ksp] = K[0];
                            kd[(size_t)(st * 2 + 1) * ksp] = K[1];
                        }
                        if (posmap) {                                   // (uniform)
                            cu_dst[(size_t)(st * 2 + 0) * cu_step] = K[0];
                            cu_dst[(size_t)(st * 2 + 1) * cu_step] = K[1];
                        }
                        const double bef = (t > 2 && t < c) ? 1.0 : 0.0, aft = (t > c + 1) ? 1.0 : 0.0;
                        double s0 = (init && t == c) ? ps->lxy[0] : (init && t == c + 1) ? ps->lxy[1] : sv;
                        double acc = 0.0;
                        acc = fma(K[0], ps->dz[0], acc);
                        acc = fma(K[1], ps->dz[1], acc);
                        s0 = s0 + acc;                                  // :275
                        if (t == 0) s0 = ps->heading;                   // :276, wrapped by the chain
                        sv = s0;
#pragma unroll
                        for (int j = 0; j < kQuadRows; ++j) {
                            if (4 * j >= 3 && 4 * j + 3 < pos + 2) continue;
                            const int p = 4 * j + k;
                            const bool on = p < NU && !(p >= 3 && p < pos + 2);
                            const int pcx = on ? p : 0;
                            if constexpr (RANKP) {                                // (uniform) P(t, U[p]) -= K_s(t, :) V_s(:, U[p])
                                const double nv = fma(plan_kv(ps)[pcx][3], -K[1], fma(plan_kv(ps)[pcx][2], -K[0], PN[j]));
                                PN[j] = on ? nv : PN[j];
                                continue;
                            }
                            const double r[5] = { ps->BR[pcx][0], ps->BR[pcx][1], ps->BR[pcx][2], ps->BR[pcx][3], ps->BR[pcx][4] };
                            const double nv = p1_entry<T>(m, r, PN[j], bef, aft);
                            PN[j] = on ? nv : PN[j];
                        }
                    }
                }
            }
            if (st + 1 < J) {
                if (!pre) {
                    if (wait_entry(st + 1)) { fetch(st + 1); pre = true; }
                    else if ((threadIdx.x & 63) == 0) fail_sh = 1;
                }
                if (pre) stash(st + 1);
                __syncthreads();
                if (fail_sh != 0) {
                    failed = true;
                    if (threadIdx.x == 0) atomicAdd(timeouts, 1);
                }
            }
        }
    }
    };
    if constexpr (TAGGED) loop(std::true_type{});
    else {
        if (rankp) loop(std::true_type{});
        else loop(std::false_type{});
    }
    if constexpr (RUN) {
        // the next tick's chain (another workgroup, this launch) reads the state: agent-scope store, then this workgroup counts itself done
        if (role == 1 && owner) st_agent(&v.s_out[(size_t)b * ld + t], sv);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) st_agent(rn.done + wg, rn.stamp);
        TLMAX(23, threadIdx.x == 0 && !rn.first && rn.stamp_dbg_second_last);
    } else if (role == 1 && owner) v.s_out[(size_t)b * ld + t] = sv;
    if constexpr (TAGGED) {
        if (tg.mirror) {                                                // (uniform)
            if (role == 1 && owner) st_sys(tg.mirror + t, sv);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) st_sys(tg.mtags + wg, tg.mseq);
        }
    }
    if (role == 1 && pm >= 0) SU[(size_t)b * NU + pm] = sv;
}

// The strips as a launch of their own that FOLLOWS a chain running in another launch (on the chain stream of a streamed overlapped
// run, nuslam_hip.hip): the same consumer as k_tick_front's strip workgroups, behind predict(t) and prep(t) on the handle's
// stream; drops the strips at the next tick's index set for that tick's chain and counts itself done for it.
template <typename T>
__global__ __launch_bounds__(256) void k_tick_strips(View v, TickObs o, const T* __restrict__ P, const TickStep* __restrict__ plan,
                                                     double* __restrict__ Kbuf, double* __restrict__ Rbuf, double* __restrict__ Vbuf,
                                                     TickPublish pub, int* __restrict__ timeouts, const int* __restrict__ posmap,
                                                     double* __restrict__ KU, double* __restrict__ RU, double* __restrict__ SU,
                                                     int* __restrict__ done_cnt)
{
    TL(6, blockIdx.y == 0 && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0);
    TL(8, blockIdx.y == 0 && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0);
    tick_panels_stream<T>(blockIdx.y, blockIdx.x, v, o, P, plan, Kbuf, Rbuf, Vbuf, pub, timeouts, posmap, KU, RU, SU);
    if (done_cnt) {
        __syncthreads();
        if (threadIdx.x == 0) tick_signal(done_cnt);
    }
    TL(7, blockIdx.y == 0 && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0);
    TL(9, blockIdx.y == 0 && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0);
}

// ... and the chain it follows: k_tick_chain announcing its plan entry by entry (FUSED: started from the previous tick's strips)
template <typename T, bool FUSED>
__global__ __launch_bounds__(256) void k_tick_chain_pub(View v, TickObs o, int total_landmarks, const T* __restrict__ P,
                                                        TickStep* __restrict__ plan, TickCarry cy, int* __restrict__ ctrl_out4,
                                                        int* __restrict__ done_cnt, TickPublish pub)
{
    tick_chain<T, FUSED, true>(blockIdx.x, v, o, total_landmarks, P, plan, cy, ctrl_out4, done_cnt, pub);
}

// The tick's predict as workgroups of k_tick_front (thread t: column t of rows 1, 2, row t of columns 1, 2, state entry t; thread
// 0 the 3 x 3 corner + Q -- k_predict's arithmetic, slam_library.cpp:65-148).  They start when the chain has gathered its block
// from the covariance BEFORE predict, store with agent-scope stores, and count themselves done for the strip workgroups.
template <typename T, bool RUN = false>
__device__ inline void tick_predict_role(const int b, const int blk, View v, TickPublish pub, T* __restrict__ P,
                                         int* __restrict__ timeouts, TickRun rn = TickRun{})
{
    // RUN (k_run_fused): state and covariance were left by other workgroups in the previous tick of this launch: agent-scope loads
    auto ldp = [](const T* q) -> double { if constexpr (RUN) return (double)ld_agent(q); else return (double)*q; };
    auto lds = [](const double* q) -> double { if constexpr (RUN) return ld_agent(q); else return *q; };
    if (threadIdx.x == 0) {
        int ok = 0;
        for (int it = 0; it < (1 << 18); ++it) {
            if (seq_reached(ld_agent(pub.flag + kPubWords * b + 1), pub.gbase)) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) atomicAdd(timeouts, 1);               // (goes on all the same: every wait of this launch is bounded)
    }
    __syncthreads();
    if constexpr (RUN) {
        // the previous tick's pass workgroups have stored everything they export (the edge tiles this role rewrites, the rows / columns
        // the strips will read once this role has counted itself done)
        if (!rn.first) {
            constexpr int kWordsPerThread = 4;
            int wi[kWordsPerThread];
            bool need[kWordsPerThread];
#pragma unroll
            for (int q = 0; q < kWordsPerThread; ++q) {
                const int idx = (int)threadIdx.x + 256 * q;
                need[q] = idx < rn.n_pass && ((idx >> 3) / rn.tiles_c) * 8 + (idx & 7) < rn.tiles_r;
                wi[q] = need[q] ? idx : 0;
            }
            bool ok = false;
            for (int it = 0; it < (1 << 20) && !ok; ++it) {
                int val[kWordsPerThread];
#pragma unroll
                for (int q = 0; q < kWordsPerThread; ++q) val[q] = need[q] ? ld_agent(rn.done_all + wi[q]) : 0;
                ok = true;
#pragma unroll
                for (int q = 0; q < kWordsPerThread; ++q) {
                    if (need[q] && seq_reached(val[q], rn.stamp - 1)) need[q] = false;
                    ok = ok && !need[q];
                }
            }
            if (!__syncthreads_and(ok ? 1 : 0) && threadIdx.x == 0) atomicAdd(timeouts, 1);
        }
    }
    const int t = blk * 256 + threadIdx.x;
    const int ld = v.ld;
    const double* s = v.s_in + (size_t)b * ld;
    double* so = v.s_out + (size_t)b * ld;
    const TwistArg& tw = pub.tw;
    const double dth = tw.tw ? tw.tw[b * tw.stride + tw.off + 0] : tw.dth0;
    const double dx = tw.tw ? tw.tw[b * tw.stride + tw.off + 1] : tw.dx0;
    const double theta = lds(&s[0]);
    const MotionStep ms = motion_step(theta, dth, dx);                 // predictEstimate :71-94, getA :127-148
    const double dq_x = ms.dq_x, dq_y = ms.dq_y, th1 = ms.th1, a1 = ms.a1, a2 = ms.a2;
    if (t < ld) st_agent(&so[t], t == 0 ? th1 : t == 1 ? lds(&s[1]) + dq_x : t == 2 ? lds(&s[2]) + dq_y : lds(&s[t]));
    T* Pb = P + (size_t)b * v.p_stride;
    if (t == 0) {
        double p[3][3], tt[3][3], u[3][3];
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) p[i][j] = ldp(&Pb[(size_t)j * ld + i]);
        for (int j = 0; j < 3; ++j) {
            tt[0][j] = p[0][j];
            tt[1][j] = a1 * p[0][j] + p[1][j];
            tt[2][j] = a2 * p[0][j] + p[2][j];
        }
        for (int i = 0; i < 3; ++i) {
            u[i][0] = tt[i][0];
            u[i][1] = tt[i][0] * a1 + tt[i][1];
            u[i][2] = tt[i][0] * a2 + tt[i][2];
        }
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) st_agent(&Pb[(size_t)j * ld + i], (T)(u[i][j] + v.Q[i + 3 * j]));
    } else if (t >= 3 && t < v.L) {
        T* col = Pb + (size_t)t * ld;
        const double p0 = ldp(&col[0]);
        const double p1 = ldp(&col[1]);
        const double p2 = ldp(&col[2]);
        const double t0 = ldp(&Pb[t]);
        const double r1 = ldp(&Pb[(size_t)1 * ld + t]), r2 = ldp(&Pb[(size_t)2 * ld + t]);
        st_agent(&col[1], (T)(a1 * p0 + p1));
        st_agent(&col[2], (T)(a2 * p0 + p2));
        st_agent(&Pb[(size_t)1 * ld + t], (T)(t0 * a1 + r1));
        st_agent(&Pb[(size_t)2 * ld + t], (T)(t0 * a2 + r2));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(pub.flag + kPubWords * b + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// grid.x = 1 (the chain) + n_pred (predict workgroups, when the tick's predict rides along) + the strip workgroups; grid.y = filter
template <typename T>
__global__ __launch_bounds__(256) void k_tick_front(View v, TickObs o, int total_landmarks, T* __restrict__ P,
                                                    TickStep* __restrict__ plan, double* __restrict__ Kbuf,
                                                    double* __restrict__ Rbuf, double* __restrict__ Vbuf,
                                                    TickPublish pub, int n_pred, int* __restrict__ timeouts)
{
    const int b = blockIdx.y;
    const int x = blockIdx.x;
    if (x == 0) tick_chain<T, false, true>(b, v, o, total_landmarks, P, plan, TickCarry{}, nullptr, nullptr, pub);
    else if (x <= n_pred) {
        TL(4, b == 0 && x == 1 && threadIdx.x == 0);                    // predict role: entry / exit
        tick_predict_role<T>(b, x - 1, v, pub, P, timeouts);
        TL(5, b == 0 && x == 1 && threadIdx.x == 0);
    } else {
        const int last = (int)gridDim.x - 1, mid = (n_pred + 1 + last) / 2;
        (void)last; (void)mid;
        TL(6, b == 0 && x == mid && threadIdx.x == 0);                  // strips: a middle workgroup and the last one, entry / exit
        TL(8, b == 0 && x == last && threadIdx.x == 0);
        tick_panels_stream<T>(b, x - 1 - n_pred, v, o, P, plan, Kbuf, Rbuf, Vbuf, pub, timeouts);
        TL(7, b == 0 && x == mid && threadIdx.x == 0);
        TL(9, b == 0 && x == last && threadIdx.x == 0);
    }
}

// ------------------------------------------------------------------------------------------------ the pass over P
// Tiling as k_update: a wave owns 64*RPL consecutive rows x 16 (or 8) columns, a lane moves 16 (or 32) bytes per column;
// WAVES column strips per workgroup.  Everything the J corrections need besides the tile is staged in LDS ONCE, in one burst of loads
// issued ahead of the tile's: K_s at the workgroup's rows (J x 2 x 64*VEC doubles, shared by its waves) and R_s at each
// wave's 16 columns (J x 5 x 16 doubles per wave).  The loop over the corrections then touches only LDS and registers:
// per correction and lane two 16-byte reads of K, ten FMAs per row for M(i, set), and per column five broadcast reads of
// R and seven FMAs per element -- the FMA chain of k_update, operation for operation.
// the chain's last operation, p <- after * p + acc: for fp64 storage written onto the tile's own register (the loop over
// the corrections carries the tile in fixed registers; the compiler's v_fmac form lands in the accumulator and cost one
// v_mov_b64 per element and correction to bring it home: 32 of ~300 vector instructions)
template <typename T>
__device__ inline void last_term(T& p, double after, double pij, double acc) { p = (T)fma(after, pij, acc); }
template <>
__device__ inline void last_term<double>(double& p, double after, double, double acc)
{
    asm("v_fma_f64 %0, %1, %0, %2" : "+v"(p) : "v"(after), "v"(acc));
}

// RPL = rows per lane: 16 bytes' worth (2 for fp64, 4 for fp32; 16 columns per wave); the kernel also instantiates with
// 4 rows of fp64 (two 16-byte loads per column, 8 columns per wave: half the LDS broadcast reads of R per FMA).  That
// variant measured SLOWER at len = 2003, J = 16 (36.3 against 33.7 us): the pass is not LDS-bound.  What it is: ~15 us
// that do not depend on J (staging K and R, the tile's loads, its stores -- none of it overlapped with arithmetic,
// because the eight waves of a CU move in step) plus ~1.4 us per correction (fp64 FMA issue: 224 chain FMAs + ~80 other
// vector instructions per wave and correction, two waves per SIMD), against 11.4 us of pure FMA issue for J = 16.
// KLDS = false: the gain rows are not staged in LDS; every lane fetches K_s of its own rows from the strips (L2) two
// corrections ahead.  The workgroup then holds only its waves' R (10 KB per wave) and nothing is shared between waves: no
// barrier, three 4-wave workgroups per CU instead of two -- for batches of small filters, where many generations of
// workgroups pass through a CU and the ones in their load / store phase hide behind the ones doing arithmetic.
template <typename T, int WAVES, int RPL, bool KLDS>
__global__ __launch_bounds__(64 * WAVES, KLDS ? 1 : 3) void k_tick_apply(View v, int J, const TickStep* __restrict__ plan,
                                                           const double* __restrict__ Kbuf, const double* __restrict__ Rbuf,
                                                           const T* __restrict__ Pin, T* __restrict__ Pout, int only_if_init)
{
    // only_if_init: launched behind k_tick_rank, which leaves the filters whose round holds a first sighting to this kernel
    if (only_if_init) {
        unsigned am; bool any_init;
        round_flags(plan + (size_t)blockIdx.z * kTickJ, J, am, any_init);
        if (!any_init) return;
    }
    typedef Pack16<T> vec_t;
    typedef Pack16<double> d2_t;
    constexpr int VEC = 16 / sizeof(T);                   // elements per 16-byte vector
    constexpr int NV = RPL / VEC;                         // vectors per column and lane
    static_assert(NV == 1 || NV == 2, "a lane owns 16 or 32 bytes of every column");
    constexpr int CW = NV == 1 ? 16 : 8;                  // columns of a wave
    constexpr int ROWS = 64 * RPL;                        // rows of a workgroup
    const int b = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ld = v.ld, L = v.L;
    const int rbase = blockIdx.x * ROWS;
    const int row0 = rbase + lane * RPL;
    const int strip = blockIdx.y * WAVES + wave;
    const bool active = strip * CW < L;
    const int j0 = active ? strip * CW : 0;
    const bool rows_ok = row0 < ld;                       // (ld is a multiple of 32 >= RPL: a lane's rows are all inside or all outside)
    const int rowc = rows_ok ? row0 : 0;
    const int ncol = (L - j0) < CW ? (L - j0) : CW;
    const TickStep* pl = plan + (size_t)b * kTickJ;

    extern __shared__ double lds_apply[];
    double* Ks = lds_apply;                               // [J][2][ROWS]  (KLDS only)
    double* Rs = lds_apply + (KLDS ? (size_t)J * 2 * ROWS : 0) + (size_t)wave * J * 5 * CW;   // this wave's [J][5][CW]

    // ---- one burst of loads: K (workgroup-cooperative), R (per wave), then the tile
    constexpr int KCH = KLDS ? kTickJ * 2 * ROWS / 2 / (64 * WAVES) : 1;     // 16-byte chunks of K per thread, at most
    d2_t kst[KCH];
    const double* Kb = Kbuf + (size_t)b * kTickJ * 2 * ld;
    const int nk = J * 2 * (ROWS / 2);                    // chunks: [J*2 rows][ROWS/2]
    if (KLDS) {
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int k = threadIdx.x + i * 64 * WAVES;
            const int kk = k < nk ? k : 0;
            const int row = kk / (ROWS / 2), piece = kk % (ROWS / 2);
            int gr = rbase + 2 * piece;
            gr = gr < ld ? gr : 0;                        // rows past the padded length: any readable address (never used)
            kst[i] = *reinterpret_cast<const d2_t*>(Kb + (size_t)row * ld + gr);
        }
    }
    constexpr int RCH = (kTickJ * 5 * CW / 2 + 63) / 64;  // 16-byte chunks of R per lane, at most
    d2_t rst[RCH];
    const double* Rb = Rbuf + (size_t)b * kTickJ * 5 * ld + j0;
    const int nr = J * 5 * (CW / 2);
#pragma unroll
    for (int i = 0; i < RCH; ++i) {
        const int k = lane + i * 64;
        const int kk = k < nr ? k : 0;
        const int row = kk / (CW / 2), piece = kk % (CW / 2);
        // (columns past len in the last strip: still inside the row's ld entries; they feed only columns never stored)
        rst[i] = *reinterpret_cast<const d2_t*>(Rb + (size_t)row * ld + 2 * piece);
    }
    const T* Pr = Pin + (size_t)b * v.p_stride + (size_t)j0 * ld + rowc;
    vec_t p[CW][NV];
#pragma unroll
    for (int jj = 0; jj < CW; ++jj)
#pragma unroll
        for (int nv = 0; nv < NV; ++nv) p[jj][nv] = load_stream(Pr + (size_t)(jj < ncol ? jj : 0) * ld + nv * VEC);
    if (KLDS) {
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int k = threadIdx.x + i * 64 * WAVES;
            if (k < nk) *reinterpret_cast<d2_t*>(Ks + 2 * (size_t)k) = kst[i];
        }
    }
#pragma unroll
    for (int i = 0; i < RCH; ++i) {
        const int k = lane + i * 64;
        if (k < nr) *reinterpret_cast<d2_t*>(Rs + 2 * (size_t)k) = rst[i];
    }
    if (KLDS) __syncthreads();                            // (without K in LDS a wave reads back only what it wrote itself)
    if (!active) return;
    // KLDS == false: this lane's gain rows of corrections st and st + 1, fetched two ahead
    constexpr int KV = RPL / 2;                           // 16-byte pieces per K component
    const double* Kl0 = Kb + rowc;
    d2_t kq[2][2][KV];                                    // [which of the two in flight][component][piece]
    if (!KLDS) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int e = 0; e < KV; ++e)
                    kq[a][r][e] = *reinterpret_cast<const d2_t*>(Kl0 + (size_t)((a < J ? a : 0) * 2 + r) * ld + 2 * e);
    }

    // the step's scalars (skip, c, H) are fetched one correction ahead: a scalar load issued at the top of the
    // iteration that needs it would expose its latency sixteen times
    int nskip = pl[0].skip, nc = pl[0].c;
    double nH[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) nH[q] = pl[0].Hc[q];
    for (int st = 0; st < J; ++st) {
        const int skip = nskip, c = nc;
        double Hc[10];
#pragma unroll
        for (int q = 0; q < 10; ++q) Hc[q] = nH[q];
        {
            const TickStep* pn = pl + (st + 1 < J ? st + 1 : st);
            nskip = pn->skip; nc = pn->c;
#pragma unroll
            for (int q = 0; q < 10; ++q) nH[q] = pn->Hc[q];
        }
        d2_t kc[2][KV];                                   // KLDS == false: K_s of this lane's rows; refill its slot with K_{s+2}
        if (!KLDS) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int e = 0; e < KV; ++e) {
                    kc[r][e] = kq[0][r][e];
                    kq[0][r][e] = kq[1][r][e];
                    kq[1][r][e] = *reinterpret_cast<const d2_t*>(Kl0 + (size_t)((st + 2 < J ? st + 2 : 0) * 2 + r) * ld + 2 * e);
                }
        }
        if (skip) continue;                               // wave-uniform
        double m[RPL][5], bef[RPL], aft[RPL];
#pragma unroll
        for (int e = 0; e < RPL; e += 2) {
            const d2_t k0 = KLDS ? *reinterpret_cast<const d2_t*>(Ks + (size_t)(st * 2 + 0) * ROWS + lane * RPL + e) : kc[0][e / 2];
            const d2_t k1 = KLDS ? *reinterpret_cast<const d2_t*>(Ks + (size_t)(st * 2 + 1) * ROWS + lane * RPL + e) : kc[1][e / 2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int i = row0 + e + h2;
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    double kh = 0.0;                      // exactly gain_row's M(i, set[q]) = delta - (K H)(i, set[q])
                    kh = fma(k0.v[h2], Hc[0 + 2 * q], kh);
                    kh = fma(k1.v[h2], Hc[1 + 2 * q], kh);
                    const int sidx = q < 3 ? q : c + (q - 3);
                    m[e + h2][q] = (i == sidx ? 1.0 : 0.0) - kh;
                }
                bef[e + h2] = ((i > 2) && (i < c)) ? 1.0 : 0.0;
                aft[e + h2] = (i > c + 1) ? 1.0 : 0.0;
            }
        }
        const double* Rst = Rs + (size_t)st * 5 * CW;
        // the prior rows of eight columns at a time, as 20 broadcast ds_read_b128 issued together (one dependent
        // read-then-FMA group per column left the loop waiting on LDS latency sixteen times per correction)
#pragma unroll
        for (int half = 0; half < CW / 8; ++half) {
            d2_t rr[5][4];
#pragma unroll
            for (int q = 0; q < 5; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) rr[q][k] = *reinterpret_cast<const d2_t*>(Rst + q * CW + half * 8 + 2 * k);
            // sweep_entry's chain, written stage by stage over the 8 x RPL independent elements: seven dependent FMAs
            // per element, sixteen (or thirty-two) chains in flight
            double acc[8][RPL], pij[8][RPL];
#pragma unroll
            for (int j8 = 0; j8 < 8; ++j8)
#pragma unroll
                for (int e = 0; e < RPL; ++e) {
                    pij[j8][e] = (double)p[half * 8 + j8][e / VEC].v[e % VEC];
                    acc[j8][e] = m[e][0] * rr[0][j8 >> 1].v[j8 & 1];
                }
#pragma unroll
            for (int q = 1; q < 3; ++q)
#pragma unroll
                for (int j8 = 0; j8 < 8; ++j8)
#pragma unroll
                    for (int e = 0; e < RPL; ++e) acc[j8][e] = fma(m[e][q], rr[q][j8 >> 1].v[j8 & 1], acc[j8][e]);
#pragma unroll
            for (int j8 = 0; j8 < 8; ++j8)
#pragma unroll
                for (int e = 0; e < RPL; ++e) acc[j8][e] = fma(bef[e], pij[j8][e], acc[j8][e]);
#pragma unroll
            for (int q = 3; q < 5; ++q)
#pragma unroll
                for (int j8 = 0; j8 < 8; ++j8)
#pragma unroll
                    for (int e = 0; e < RPL; ++e) acc[j8][e] = fma(m[e][q], rr[q][j8 >> 1].v[j8 & 1], acc[j8][e]);
#pragma unroll
            for (int j8 = 0; j8 < 8; ++j8)
#pragma unroll
                for (int e = 0; e < RPL; ++e) last_term(p[half * 8 + j8][e / VEC].v[e % VEC], aft[e], pij[j8][e], acc[j8][e]);
        }
    }
    if (!rows_ok) return;
    T* Pw = Pout + (size_t)b * v.p_stride + (size_t)j0 * ld + row0;
#pragma unroll
    for (int jj = 0; jj < CW; ++jj)
        if (jj < ncol) {
#pragma unroll
            for (int nv = 0; nv < NV; ++nv) store_stream(Pw + (size_t)jj * ld + nv * VEC, p[jj][nv]);
        }
}

// ------------------------------------------------------------------------------------------------ the pass, one big filter
// fp64, one resident generation of 8-wave workgroups (a single large filter).  What k_tick_apply left on the table there
// (phase counters: waves parked 45 % of their life, 34 % of it in the load / store phases, which nothing overlapped
// because the eight waves of a CU move in step):
//  * a wave's 16 columns are two UNITS of 8: everything is requested up front, unit A is carried through the J
//    corrections while unit B's loads are still landing, A's stores drain under B's arithmetic.  For that no vector-memory
//    result may be waited for inside the loops (vmcnt retires in order: waiting for anything would wait for unit B), so
//  * the loops read only LDS: M_s(i, set_s) of the workgroup's 128 rows, formed ONCE by the workgroup (wave w forms
//    corrections w and w + 8) instead of by each of its eight waves, and the unit's prior rows R_s (unit B's travel in
//    registers until A is done).  LDS: J * (5 * 128 + 8 * 5 * 8) doubles = 120 KB at J = 16.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_tick_apply_units(View v, int J, const TickStep* __restrict__ plan,
                                                                 const double* __restrict__ Kbuf, const double* __restrict__ Rbuf,
                                                                 const double* __restrict__ Pin, double* __restrict__ Pout, int only_if_init)
{
    if (only_if_init) {
        unsigned am; bool any_init;
        round_flags(plan + (size_t)blockIdx.z * kTickJ, J, am, any_init);
        if (!any_init) return;
    }
    typedef Pack16<double> d2_t;
    constexpr int ROWS = 128, CW = 16, UW = 8;
    const int b = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ld = v.ld, L = v.L;
    const int rbase = blockIdx.x * ROWS;
    const int row0 = rbase + lane * 2;
    const int strip = blockIdx.y * WAVES + wave;
    const bool active = strip * CW < L;
    const int j0 = active ? strip * CW : 0;
    const bool rows_ok = row0 < ld;
    const int rowc = rows_ok ? row0 : 0;
    const int ncol = (L - j0) < CW ? (L - j0) : CW;
    const TickStep* pl = plan + (size_t)b * kTickJ;

    extern __shared__ double lds_apply[];
    double* Ms = lds_apply;                                             // [J][5][ROWS]
    double* Rs = lds_apply + (size_t)J * 5 * ROWS + (size_t)wave * J * 5 * UW;   // this wave's [J][5][UW], one unit at a time

    // ---- everything is requested here
    constexpr int NP = (kTickJ + WAVES - 1) / WAVES;                    // corrections whose M this wave forms
    const double* Kb = Kbuf + (size_t)b * kTickJ * 2 * ld + rowc;
    d2_t kk[NP][2];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int st = wave + WAVES * i;
        const int sq = st < J ? st : 0;
        kk[i][0] = *reinterpret_cast<const d2_t*>(Kb + (size_t)(sq * 2 + 0) * ld);
        kk[i][1] = *reinterpret_cast<const d2_t*>(Kb + (size_t)(sq * 2 + 1) * ld);
    }
    constexpr int RCH = (kTickJ * 5 * (UW / 2) + 63) / 64;              // 16-byte chunks of one unit's R per lane
    d2_t ra[RCH], rb[RCH];
    const double* Rb = Rbuf + (size_t)b * kTickJ * 5 * ld + j0;
    const int nr = J * 5 * (UW / 2);
#pragma unroll
    for (int i = 0; i < RCH; ++i) {
        const int k = lane + i * 64;
        const int kq = k < nr ? k : 0;
        const int row = kq / (UW / 2), piece = kq % (UW / 2);
        ra[i] = *reinterpret_cast<const d2_t*>(Rb + (size_t)row * ld + 2 * piece);
        rb[i] = *reinterpret_cast<const d2_t*>(Rb + (size_t)row * ld + UW + 2 * piece);
    }
    const double* Pr = Pin + (size_t)b * v.p_stride + (size_t)j0 * ld + rowc;
    d2_t pa[UW], pb[UW];
#pragma unroll
    for (int jj = 0; jj < UW; ++jj) pa[jj] = load_stream(Pr + (size_t)(jj < ncol ? jj : 0) * ld);
#pragma unroll
    for (int jj = 0; jj < UW; ++jj) pb[jj] = load_stream(Pr + (size_t)(UW + jj < ncol ? UW + jj : 0) * ld);

    // ---- M of corrections wave, wave + WAVES, ... at this lane's two rows: exactly gain_row's M(i, set[q])
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int st = wave + WAVES * i;
        if (st < J) {
            const TickStep* ps = pl + st;
            const int c = ps->c;
            double Hc[10];
#pragma unroll
            for (int q = 0; q < 10; ++q) Hc[q] = ps->Hc[q];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                d2_t mq;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    double kh = 0.0;
                    kh = fma(kk[i][0].v[h2], Hc[0 + 2 * q], kh);
                    kh = fma(kk[i][1].v[h2], Hc[1 + 2 * q], kh);
                    const int sidx = q < 3 ? q : c + (q - 3);
                    mq.v[h2] = (row0 + h2 == sidx ? 1.0 : 0.0) - kh;
                }
                *reinterpret_cast<d2_t*>(Ms + (size_t)(st * 5 + q) * ROWS + 2 * lane) = mq;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < RCH; ++i) {
        const int k = lane + i * 64;
        if (k < nr) *reinterpret_cast<d2_t*>(Rs + 2 * (size_t)k) = ra[i];
    }
    lds_barrier();                                                      // (__syncthreads() would also wait for unit B's loads)
    if (!active) return;

    auto run_unit = [&](d2_t (&p)[UW]) {
        int nskip = pl[0].skip, nc = pl[0].c;
        for (int st = 0; st < J; ++st) {
            const int skip = nskip, c = nc;
            {
                const TickStep* pn = pl + (st + 1 < J ? st + 1 : st);
                nskip = pn->skip; nc = pn->c;
            }
            if (skip) continue;                                         // wave-uniform
            double m[2][5], bef[2], aft[2];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const d2_t mq = *reinterpret_cast<const d2_t*>(Ms + (size_t)(st * 5 + q) * ROWS + 2 * lane);
                m[0][q] = mq.v[0]; m[1][q] = mq.v[1];
            }
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int i = row0 + h2;
                bef[h2] = ((i > 2) && (i < c)) ? 1.0 : 0.0;
                aft[h2] = (i > c + 1) ? 1.0 : 0.0;
            }
            const double* Rst = Rs + (size_t)st * 5 * UW;
            d2_t rr[5][4];
#pragma unroll
            for (int q = 0; q < 5; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) rr[q][k] = *reinterpret_cast<const d2_t*>(Rst + q * UW + 2 * k);
            double acc[8][2];
#pragma unroll
            for (int j8 = 0; j8 < 8; ++j8)
#pragma unroll
                for (int e = 0; e < 2; ++e) acc[j8][e] = m[e][0] * rr[0][j8 >> 1].v[j8 & 1];
#pragma unroll
            for (int q = 1; q < 3; ++q)
#pragma unroll
                for (int j8 = 0; j8 < 8; ++j8)
#pragma unroll
                    for (int e = 0; e < 2; ++e) acc[j8][e] = fma(m[e][q], rr[q][j8 >> 1].v[j8 & 1], acc[j8][e]);
#pragma unroll
            for (int j8 = 0; j8 < 8; ++j8)
#pragma unroll
                for (int e = 0; e < 2; ++e) acc[j8][e] = fma(bef[e], p[j8].v[e], acc[j8][e]);
#pragma unroll
            for (int q = 3; q < 5; ++q)
#pragma unroll
                for (int j8 = 0; j8 < 8; ++j8)
#pragma unroll
                    for (int e = 0; e < 2; ++e) acc[j8][e] = fma(m[e][q], rr[q][j8 >> 1].v[j8 & 1], acc[j8][e]);
#pragma unroll
            for (int j8 = 0; j8 < 8; ++j8)
#pragma unroll
                for (int e = 0; e < 2; ++e) last_term(p[j8].v[e], aft[e], p[j8].v[e], acc[j8][e]);
        }
    };

    double* Pw = Pout + (size_t)b * v.p_stride + (size_t)j0 * ld + row0;
    run_unit(pa);
    if (rows_ok) {
#pragma unroll
        for (int jj = 0; jj < UW; ++jj)
            if (jj < ncol) store_stream(Pw + (size_t)jj * ld, pa[jj]);
    }
    if (ncol <= UW) return;                                             // (the last strip may end inside unit A)
#pragma unroll
    for (int i = 0; i < RCH; ++i) {
        const int k = lane + i * 64;
        if (k < nr) *reinterpret_cast<d2_t*>(Rs + 2 * (size_t)k) = rb[i];
    }
    run_unit(pb);
    if (rows_ok) {
#pragma unroll
        for (int jj = 0; jj < UW; ++jj)
            if (UW + jj < ncol) store_stream(Pw + (size_t)(UW + jj) * ld, pb[jj]);
    }
}

} // namespace nuslam
