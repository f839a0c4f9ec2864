// ekf_pipe.h -- pipelined runs (nuslam_hip.hip, run_pipelined): the panels tick t starts from, WITHOUT the pass over P of tick t-1.
//
// A tick's serial part (k_tick_front: chain + strips) reads of the covariance only the 35 rows and 35 columns at its index set U
// (3 pose indices + 2 per marker).  Entry (i, j) of what tick t starts from is
//     predict_t( P_{t-1|t-2}(i, j) - sum_s K_s(i, :) V_s(:, j) )
// -- the input of the pass of tick t-1 (ping-ponged: stable while that pass runs), the strips tick t-1 left (which the pass
// reads too), and the motion model's two Jacobian entries.  k_tick_panel_carry forms exactly these 70 x len entries, with
// k_tick_rank's own sum per entry (acc = fma(V_f(col), -K_f(row), acc), f = 2 s + r ascending over the live corrections, rounded
// to the storage type once) and k_predict's own arithmetic (slam_library.cpp:96-148), so that k_tick_front started from the
// panels computes bit for bit what it computes from the covariance after pass and predict.  The pass of tick t-1 and the predict
// of tick t then run on a second stream BESIDE the front launch of tick t instead of in front of it.
// Only rounds the rank-2m pass applies (no first sighting: round_flags) can be carried this way; the host proves that per tick.
#pragma once

namespace nuslam {

struct PanelSet {
    double* row;       // [B][kTickNU][ld]  row[p][t] = P(U[p], t)
    double* col;       // [B][kTickNU][ld]  col[p][t] = P(t, U[p])
    double* state;     // [B][ld]           the state vector after predict
};

// grid (ceil(ld / 64) + 1, 2, B), 64 threads.  blockIdx.y: 0 = row panel (thread = column t of the rows U[p]), 1 = column panel
// (thread = row t of the columns U[p]).  The last workgroup of x holds t = 0, 1, 2, whose predict needs entries of other rows /
// columns (one lane per position there); the others skip t < 3.
// Jp == 0: nothing to carry (the run's first tick); apply_predict == 0: the covariance is already predicted (likewise).
template <typename T>
__global__ __launch_bounds__(64) void k_tick_panel_carry(View v, TickObs on, const T* __restrict__ Pin, const TickStep* __restrict__ plan_prev,
                                                         int Jp, const double* __restrict__ Kbuf, const double* __restrict__ Vbuf,
                                                         TwistArg tw, int apply_predict, const double* __restrict__ s_prev, PanelSet ps,
                                                         const int* __restrict__ wait_cnt, int wait_target, int* __restrict__ timeouts)
{
    // wait_cnt (may be null): bumped by the other stream behind the predict that completed Pin; every workgroup waits for it (bounded)
    if (wait_cnt && !tick_wait(wait_cnt, wait_target) && threadIdx.x == 0) atomicAdd(timeouts, 1);
    constexpr int NU = kTickNU, NF = 2 * kTickJ;
    const int b = blockIdx.z, role = blockIdx.y, lane = threadIdx.x;
    const int ld = v.ld, L = v.L;
    const bool special = blockIdx.x == gridDim.x - 1;
    const T* Pb = Pin + (size_t)b * v.p_stride;
    const double* Kb = Kbuf + (size_t)b * kTickJ * 2 * ld;
    const double* Vb = Vbuf + (size_t)b * kTickJ * 2 * ld;
    const double* sp = s_prev + (size_t)b * ld;
    double* out = (role == 0 ? ps.row : ps.col) + (size_t)b * NU * ld;

    __shared__ int Ush[NU + 1];
    __shared__ double stage[NF][NU + 1];      // role 0: K_f(U[p]); role 1: V_f(U[p])
    if (lane < NU) {
        int u = lane;
        if (lane >= 3) {
            const int st = (lane - 3) >> 1;
            int id = 0;
            if (st < on.J) id = on.ids ? on.ids[b * on.stride + on.off + st] : on.id0[st];
            u = ((id >= 1 && id <= v.n) ? 3 + 2 * (id - 1) : 3) + ((lane - 3) & 1);
        }
        Ush[lane] = u;
    }
    unsigned actmask = 0u;
    if (Jp > 0) {
        bool any_init;
        round_flags(plan_prev + (size_t)b * kTickJ, Jp, actmask, any_init);
    }
    __syncthreads();
    {
        const double* src = role == 0 ? Kb : Vb;
        for (int e = lane; e < NF * NU; e += 64) {
            const int f = e / NU, p = e % NU;
            stage[f][p] = ((actmask >> (f >> 1)) & 1u) ? src[(size_t)f * ld + Ush[p]] : 0.0;
        }
    }
    double a1 = 0.0, a2 = 0.0;
    MotionStep ms{};
    if (apply_predict) {
        const double dth = tw.tw ? tw.tw[b * tw.stride + tw.off + 0] : tw.dth0;
        const double dx = tw.tw ? tw.tw[b * tw.stride + tw.off + 1] : tw.dx0;
        ms = motion_step(sp[0], dth, dx);
        a1 = ms.a1; a2 = ms.a2;
    }
    __syncthreads();
    // the entry (i, j) of the covariance after the previous tick's corrections, as k_tick_rank leaves it in memory
    auto carried = [&](int i, int j) {
        double acc = (double)Pb[(size_t)j * ld + i];
#pragma unroll
        for (int f = 0; f < NF; ++f)
            if ((actmask >> (f >> 1)) & 1u) acc = fma(Vb[(size_t)f * ld + j], -Kb[(size_t)f * ld + i], acc);
        return (double)(T)acc;
    };

    if (!special) {
        const int t = blockIdx.x * 64 + lane;
        const bool live = t >= 3 && (role == 0 ? t < L : t < ld);
        const int tc = live ? t : 3;
        // this thread's own factor of every correction: V_f(t) (row panel) / K_f(t) (column panel), and its 35 entries
        double own[NF];
        {
            const double* src = role == 0 ? Vb : Kb;
#pragma unroll
            for (int f = 0; f < NF; ++f) own[f] = src[(size_t)f * ld + tc];
        }
        double E[NU];
#pragma unroll
        for (int p = 0; p < NU; ++p) E[p] = role == 0 ? (double)Pb[(size_t)tc * ld + Ush[p]] : (double)Pb[(size_t)Ush[p] * ld + tc];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            if ((actmask >> (f >> 1)) & 1u) {                           // (uniform)
                if (role == 0) {
#pragma unroll
                    for (int p = 0; p < NU; ++p) E[p] = fma(own[f], -stage[f][p], E[p]);    // V_f(column t), -K_f(row U[p])
                } else {
#pragma unroll
                    for (int p = 0; p < NU; ++p) E[p] = fma(stage[f][p], -own[f], E[p]);    // V_f(column U[p]), -K_f(row t)
                }
            }
        }
#pragma unroll
        for (int p = 0; p < NU; ++p) E[p] = (double)(T)E[p];
        if (apply_predict) {
            // k_predict at t >= 3: rows 1, 2 of column t take a * row 0 (column role), columns 1, 2 of row t take column 0 * a (row role)
            if (role == 0) { const double p0 = E[0]; E[1] = (double)(T)(a1 * p0 + E[1]); E[2] = (double)(T)(a2 * p0 + E[2]); }
            else { const double t0 = E[0]; E[1] = (double)(T)(t0 * a1 + E[1]); E[2] = (double)(T)(t0 * a2 + E[2]); }
        }
        if (live) {
#pragma unroll
            for (int p = 0; p < NU; ++p) out[(size_t)p * ld + t] = E[p];
        }
        // the state vector after predictEstimate (slam_library.cpp:71-94)
        if (role == 0) {
            const int ts = blockIdx.x * 64 + lane;
            if (ts < ld) {
                double sv = sp[ts];
                if (apply_predict) sv = ts == 0 ? ms.th1 : ts == 1 ? sv + ms.dq_x : ts == 2 ? sv + ms.dq_y : sv;
                ps.state[(size_t)b * ld + ts] = sv;
            }
        }
        return;
    }
    // ---- t = 0, 1, 2: one lane per position p; the pose corner first (k_predict's thread 0)
    __shared__ double corner[3][3];
    if (lane < 9) corner[lane % 3][lane / 3] = carried(lane % 3, lane / 3);
    __syncthreads();
    double u[3][3];
    {
        double pp[3][3], tt[3][3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) pp[i][j] = corner[i][j];
        if (apply_predict) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                tt[0][j] = pp[0][j];
                tt[1][j] = a1 * pp[0][j] + pp[1][j];
                tt[2][j] = a2 * pp[0][j] + pp[2][j];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                u[i][0] = tt[i][0];
                u[i][1] = tt[i][0] * a1 + tt[i][1];
                u[i][2] = tt[i][0] * a2 + tt[i][2];
            }
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i) u[i][j] = (double)(T)(u[i][j] + v.Q[i + 3 * j]);
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i) u[i][j] = pp[i][j];
        }
    }
    if (lane < NU) {
        const int p = lane, iu = Ush[p];
        for (int t = 0; t < 3; ++t) {
            double val;
            if (p < 3) val = role == 0 ? u[p][t] : u[t][p];             // row panel: P(p, t); column panel: P(t, p)
            else if (role == 0) {
                // P(U[p], t), a landmark row: column 0 stays, columns 1, 2 take column 0 * a (row role)
                const double c0 = carried(iu, 0);
                val = t == 0 ? c0 : (apply_predict ? (double)(T)(c0 * (t == 1 ? a1 : a2) + carried(iu, t)) : carried(iu, t));
            } else {
                // P(t, U[p]), a landmark column: row 0 stays, rows 1, 2 take a * row 0 (column role)
                const double r0 = carried(0, iu);
                val = t == 0 ? r0 : (apply_predict ? (double)(T)((t == 1 ? a1 : a2) * r0 + carried(t, iu)) : carried(t, iu));
            }
            out[(size_t)p * ld + t] = val;
        }
    }
}

}  // namespace nuslam
