// nuslam_hip.hip -- host side of the C ABI declared in include/nuslam_hip.h: owns the device memory of a batch
// of filters, flips the state/control double buffers, and launches the kernels of ekf_kernels.h on the
// handle's own stream.  A single filter (nuslam_ekf_t) is a batch of one.
//
// There is no CPU fallback anywhere in this file: every entry point either runs HIP kernels or fails.
#include "../../include/nuslam_hip.h"
#include "ekf_kernels.h"
#include "dense_predict.h"
#include "ekf_sim.h"
#include "build/build_info.h"

#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "ekf_comm.h"

using namespace nuslam;

namespace nuslam {
// circle_fit.hip: simulated scans -> marker slots, the landmarks node's chain on the device
int scan_to_markers(hipStream_t stream, const float* d_scans, int n_scans, double min_range, double max_range, int m,
                    double* d_mx, double* d_my, int* d_ids, unsigned long long* d_empty, int* overflow_out);
}

namespace {

thread_local std::string g_hip_err;

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess) {                                                                       \
            g_hip_err = std::string(#expr) + ": " + hipGetErrorString(e__);                            \
            return NUSLAM_E_HIP;                                                                       \
        }                                                                                              \
    } while (0)

constexpr int kSweepCW = 16;  // columns per wave in the sweep
constexpr int kNotConcurrent = -1000;   // run_overlapped -> nuslam_batch_run: the two streams do not run side by side here

inline int roundup(int x, int m) { return (x + m - 1) / m * m; }

} // namespace

struct nuslam_batch {
    int device = 0;
    int n = 0, L = 0, ld = 0, B = 0, dtype = 0;
    hipStream_t stream = nullptr;
    int n_cu = 256;            // compute units of the device (MI355X: 256)
    double* state[2] = { nullptr, nullptr };
    int* ctrl[2] = { nullptr, nullptr };
    int sidx = 0, cidx = 0;
    void* Pbuf[2] = { nullptr, nullptr };   // covariance ping-pong; Pbuf[pidx] is current
    int pidx = 0;
    long long p_stride = 0;
    int* cur_id = nullptr;
    int* host_word = nullptr;  // pinned host memory: {id, status, expired waits} of the last stand-alone associateLandmark
    int* akey = nullptr;       // [B][2] association key slots
    int aslot = 0;
    double* tr = nullptr;      // per-filter trace scratch
    double* stats = nullptr;   // 2L + 6
    double* pose_err = nullptr; // [B][4]: squared pose error vs the simulated truth + NEES
    int last_tick = -1;        // last tick of the resident trace that nuslam_batch_run applied
    double Q[9], R[4];
    // resident trace
    double* tr_tw = nullptr; double* tr_mx = nullptr; double* tr_my = nullptr; int* tr_ids = nullptr;
    int tr_ticks = 0, tr_m = 0, tr_bcast = 0;
    bool tr_presence_only = false;   // data-association trace: tr_ids only says which marker slots are filled (id > 0)
    double* tr_truth = nullptr;   // generated traces only: the simulated robot's pose after each tick
    float* tr_scan = nullptr;     // generated with lidar != 0: [B][ticks][360] simulated ranges
    std::vector<int> h_ids;    // host copy of a broadcast trace's ids: passed inline so k_update needs no id load
    std::vector<int> h_ids_pf; // host copy of a per-filter trace's ids [B][ticks][m]: only to decide pairing per tick
    // staging for nuslam_ekf_tick (one filter, m observations)
    double* st_mx = nullptr; double* st_my = nullptr; int* st_ids = nullptr; int* id_log = nullptr;
    int st_cap = 0, log_stride = 0;
    // dense predict: staged Jacobian (element type = dtype); the product T = F P goes to the idle P buffer
    void* wF = nullptr;
    bool f_staged = false;
    // deferred application: pending rank-2 factors U_i = K_i, V_i = H_i P_{i-1}
    // tick pipeline (ekf_tick.h): plan + K / R strips of one round of up to kTickJ corrections
    int tick_mode = -1;        // 1: known-id ticks run as chain + panels + one pass over P; 0: one sweep per correction / pair;
                               // -1: whichever is faster for this handle (see tick_pipeline_pays)
    TickStep* tk_plan = nullptr; double* tk_K = nullptr; double* tk_R = nullptr;
    bool tk_ready = false;     // every buffer of the tick pipeline allocated and initialised (ensure_tick_buffers)
    int* tk_pub = nullptr;     // [B]: the chain's announcements to the strip workgroups of the same launch (k_tick_front)
    unsigned seq_pub = 0, seq_gather = 0, seq_pred = 0;   // (sequence words: they wrap, the device compares wrapped differences)
    int front = 1;             // 1: one filter's chain and strips as ONE launch (k_tick_front) while its grid fits the chip
    int run_fused = 1;         // nuslam_batch_run on one filter, known ids, no first sighting possible: the run's ticks as ONE launch (k_run_fused)
    unsigned run_done = 0;     // k_run_fused: the running number of the last tick enqueued (tk_run: the workgroups' words)
    int* tk_run = nullptr;
    int fuse_pass = 1;         // 1: ... and the rank-2m pass over P as workgroups of that launch too (k_tick_fused, ekf_fused.h) in rounds
                               // the host can prove free of first sightings; 0: the pass as a launch of its own behind it (tick mode 4)
    // the state vector mirrored into mapped pinned host memory by the kernels that produce it (fused ticks, served rounds): get_state
    // then answers without a stream synchronisation and without a device-to-host copy
    double* st_host = nullptr; long long* st_tags = nullptr;
    long long st_seq = 0;      // the stamp of the last mirrored launch
    int st_ntags = 0;          // ... and how many workgroups stamp
    unsigned long long state_epoch = 1, mirror_epoch = 0;   // mirror valid <=> nothing has touched the state since that launch
    long long* tk_tagK = nullptr;   // [B][2 kTickJ][ld][2]: the K / V strips of a fused round as self-validating { half, tag } words
    long long* tk_tagV = nullptr;
    unsigned seq_tag = 0;      // the last round's tag (0 is never one)
    double* tk_V = nullptr;    // V_s = H_s R_s strips [B][kTickJ][2][ld]: the second factor of the rank-2m pass (ekf_rank.h)
    // the pass over P of a tick pipeline (nuslam_batch_set_pass_variant): 0 = rank-2m on the matrix cores, the exact chain
    // for rounds with a first sighting; 1 = always the exact chain, plain kernel; 2 = always the exact chain, two-unit
    // kernel where it applies; 10 + k = as 0 with tile shape k (experiments)
    int pass_mode = 0, rank_tile = 0;
    int strips_lane = 1;       // batches, rank form proven by the host: k_tick_strips_lane; 0 (variant 20): k_tick_panels<T, 64> as before
    std::vector<unsigned char> touched;   // [B][n + 1]: landmarks the host can PROVE have been corrected at least once (known-id
                               // ticks, restore): a round of such ids cannot cancel an INT_MAX diagonal, so the exact-chain
                               // pass need not be launched behind the rank-2m one
    // unknown association (ekf_da.h): tracked rows / columns / diagonal blocks of P, one launch per correction
    DaBuf da = {};
    void* da_mem = nullptr;
    int apply_units = 1;       // exact chain, fp64, one resident generation: the two-unit pass (k_tick_apply_units); 0: k_tick_apply
    unsigned da_round_tag = 0; // resident round kernel: tags of the key slots, kTickJ + 1 per round (slots are never reset)
    // cross-tick overlap (nuslam_batch_run on a resident trace): the chain of tick t+1 runs on its own stream while
    // strips and pass of tick t run on the handle's
    int predict_bookkeeping = 1;   // 0 while an overlapped run carries the control words on the chain stream
    int overlap = -1;          // 1 / 0: nuslam_batch_set_overlap; -1: off with the rank-2m pass (on one stream a tick at N = 1000 takes
                               // 77.7 us against 82.1 overlapped: the pass left the critical path's price range); with the exact chain
                               // on for a single filter, off for batches, whose pass fills every CU and only delays the chain
    bool ov_same_stream = false;   // test hook (set_overlap(2)): the chain "stream" IS the handle's, nothing can run beside it
    int ov_ok = -1;            // -1: not probed; 1: the two streams were seen running side by side; 0: they were not (a tool or
                               // environment serialises dispatches): overlapped runs then take the one-stream path
    bool poisoned = false;     // a bounded device-side wait expired (NUSLAM_E_SYNC read back): state / covariance are not to be
                               // trusted; ticks are refused until EVERY filter has been restored (restore()) or re-initialised
    std::vector<unsigned char> needs_restore;   // [B], while poisoned: filters not restored yet
    hipStream_t stream2 = nullptr;
    TickStep* tk_plan2 = nullptr; int* tk_ctrl4 = nullptr; double* tk_blk = nullptr;
    int* tk_sync = nullptr;                            // {chain, next} completion counters, timeouts
    int* tk_posmap = nullptr; double* tk_KU = nullptr; double* tk_RU = nullptr; double* tk_SU = nullptr;
    int seq_chain = 0, seq_next = 0;                   // the counters' values after everything enqueued so far
    hipEvent_t ov_start = nullptr;
    std::vector<hipEvent_t> ov_events;
    // Groups (nuslam_batch_set_interleave): the known-id ticks of a LARGE batch's resident trace run as G groups of filters, each on a
    // stream of its own, the groups' passes over P taking turns: one group's HBM-bound pass then runs beside the other group's
    // latency-bound chain and its strips.  Same kernels on the same per-filter data: same bits.  -1 (default): 2 groups for 512
    // filters and more (1024 x N = 200: 728 us per tick against 781), else 1.
    int groups = -1;
    hipStream_t gstream[4] = { nullptr, nullptr, nullptr, nullptr };
    hipEvent_t gev[4] = { nullptr, nullptr, nullptr, nullptr }, gev0 = nullptr;
    hipEvent_t gpass[4] = { nullptr, nullptr, nullptr, nullptr };   // group g's last pass over P is done
    bool gpass_set[4] = { false, false, false, false };
    int group_ring = 1;        // the groups' passes take turns (measurement: nuslam_batch_set_interleave(h, 10 + G) lets them run free)
    bool pairing = true;       // k_update2 for consecutive plain corrections of a known-id tick
    std::vector<int> host_seen;   // host mirror of every filter's `seen`; valid while only known-id calls were made
    bool host_seen_valid = true;
    bool deferred = false;
    double* dU = nullptr; double* dV = nullptr;
    int J = 0;
    // The class API driven call by call (slam.cpp:269-318): predict() / initializeLandmark() / update() of ONE filter are recorded
    // here and reach the device as one tick -- k_tick_front + the rank-2m pass, the kernels nuslam_ekf_tick launches -- when the
    // next predict(), a getter, associateLandmark(), a copy, sync() or anything else that looks at the filter arrives (lazy_flush).
    struct Lazy {
        bool on = false;           // nuslam_ekf_create turns it on for single filters (nuslam_ekf_set_lazy)
        bool has_predict = false;
        double dth = 0.0, dx = 0.0;
        std::vector<double> r, phi;            // the recorded update() calls, in order
        std::vector<int> id;
        std::vector<unsigned char> init;       // ... and whether initializeLandmark(z, id) with the same z and id came right before
        bool pend_init = false;                // an initializeLandmark whose update() has not arrived yet
        double pi_r = 0.0, pi_phi = 0.0;
        int pi_id = 0;
        bool busy = false;                     // lazy_flush is running (its own launches must not re-enter it)
        bool empty() const { return !has_predict && id.empty() && !pend_init; }
    } lazy;
    // ... with associateLandmark() in the loop (slam.cpp:291): a resident round kernel that takes the caller's calls from a mailbox in
    // mapped pinned host memory (k_da_round<T, true>, csrc/ekf_da.h "a round SERVED to the host")
    struct Serve {
        long long* mail = nullptr;             // [0..7] the command, [8..9] the answer (two cache lines), [16 .. 16 + nwg) the workgroups' keys
        int seq = 0;                           // sequence number of the last command sent
        bool open = false;                     // a served round is resident on the device
        int trips = 0;                         // commands of this round the device has carried out
        bool any_init = false;                 // one of its corrections was a first sighting (the caller initialised the landmark)
        bool pend = false;                     // the correction the caller has decided on for the last marker: sent with the next command
        int p_id = 0; bool p_init = false; double p_r = 0.0, p_phi = 0.0;
        bool z_valid = false; double z_r = 0.0, z_phi = 0.0;    // the marker the device scanned last (its correction's default z)
        bool clear_status = false;             // the latched status was returned to the caller: the next command clears it
        long long mirror_seq = 0;              // the open round stamps the state mirror with this when it ends
        int timeout_us = 1000;                 // the device closes a round by itself when no command comes for this long
        bool used = false;                     // associateLandmark was called since the last predict: the caller's loop is slam.cpp's, and
                                               // the next predict() opens the next served round at once (predict kernel and round kernel
                                               // queue up behind the pass over P instead of waiting for the first associateLandmark)
    } srv;
    bool dense_predict = false;   // do_predict: state-only kernel + the two MFMA products with the staged Jacobian
    bool dense_getA = false;      // ... and the staged Jacobian is getA(tw) itself: I + B, B(1,0), B(2,0) rewritten every predict
    // profiling
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[NUSLAM_K_COUNT];
    std::vector<hipEvent_t> pool;
    double prof_ms[NUSLAM_K_COUNT] = { 0 };
    long long prof_n[NUSLAM_K_COUNT] = { 0 };
    hipEvent_t t0 = nullptr, t1 = nullptr;

    size_t esize() const { return dtype == NUSLAM_F32 ? 4 : 8; }
    void* P() const { return Pbuf[pidx]; }
    void* Palt() const { return Pbuf[pidx ^ 1]; }

    View view() const
    {
        View v;
        v.n = n; v.L = L; v.ld = ld; v.B = B;
        v.s_in = state[sidx]; v.s_out = state[sidx ^ 1];
        v.c_in = ctrl[cidx]; v.c_out = ctrl[cidx ^ 1];
        v.p_stride = p_stride;
        v.cur_id = cur_id; v.akey = akey; v.aslot = aslot; v.id_log = id_log; v.log_stride = log_stride;
        v.dump = tk_R ? tk_R + (size_t)B * kTickJ * 5 * ld : nullptr;
        memcpy(v.Q, Q, sizeof(Q));
        memcpy(v.R, R, sizeof(R));
        return v;
    }
};

struct nuslam_ekf {
    nuslam_batch* core;
};

namespace {

hipEvent_t get_event(nuslam_batch* h)
{
    if (!h->pool.empty()) {
        hipEvent_t e = h->pool.back();
        h->pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

// Launch on `st` (the handle's stream, or its chain stream); with profiling on, the dispatch carries its own start/stop events.
template <typename... KArgs, typename... Args>
int launch_on(nuslam_batch* h, hipStream_t st, int kid, void (*kern)(KArgs...), dim3 grid, dim3 block, size_t lds, Args... args)
{
    if (h->prof && kid >= 0) {
        hipEvent_t e0 = get_event(h), e1 = get_event(h);
        if (!e0 || !e1) { g_hip_err = "hipEventCreate failed"; return NUSLAM_E_HIP; }
        hipExtLaunchKernelGGL(kern, grid, block, lds, st, e0, e1, 0, args...);
        h->pending[kid].emplace_back(e0, e1);
    } else {
        hipLaunchKernelGGL(kern, grid, block, lds, st, args...);
    }
    HIPCHK(hipGetLastError());
    return NUSLAM_OK;
}
template <typename... KArgs, typename... Args>
int launch(nuslam_batch* h, int kid, void (*kern)(KArgs...), dim3 grid, dim3 block, Args... args)
{
    return launch_on(h, h->stream, kid, kern, grid, block, 0, args...);
}

// the same with dynamic LDS
template <typename... KArgs, typename... Args>
int launch_lds(nuslam_batch* h, int kid, void (*kern)(KArgs...), dim3 grid, dim3 block, size_t lds, Args... args)
{
    return launch_on(h, h->stream, kid, kern, grid, block, lds, args...);
}

int drain_profile(nuslam_batch* h)
{
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int k = 0; k < NUSLAM_K_COUNT; ++k) {
        for (auto& pr : h->pending[k]) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, pr.first, pr.second));
            h->prof_ms[k] += ms;
            h->prof_n[k] += 1;
            h->pool.push_back(pr.first);
            h->pool.push_back(pr.second);
        }
        h->pending[k].clear();
    }
    return NUSLAM_OK;
}

#define DISPATCH_T(h, CALL)                                   \
    do {                                                      \
        if ((h)->dtype == NUSLAM_F32) { typedef float T; CALL; } \
        else { typedef double T; CALL; }                      \
    } while (0)

// apply the pending factors: P <- P - sum U_i V_i
int flush_pending(nuslam_batch* h)
{
    if (h->J == 0) return NUSLAM_OK;
    View v = h->view();
    const int vec = 16 / (int)h->esize();
    const int strips = (h->L + kSweepCW - 1) / kSweepCW;
    dim3 grid((h->ld + 64 * vec - 1) / (64 * vec), (strips + 3) / 4, h->B), block(256);
    int rc = NUSLAM_OK;
    DISPATCH_T(h, rc = (launch(h, NUSLAM_K_FLUSH, k_flush<T>, grid, block, v, h->J, (const T*)h->P(), (T*)h->Palt(),
                               (const double*)h->dU, (const double*)h->dV)));
    if (rc) return rc;
    h->pidx ^= 1;
    h->J = 0;
    return NUSLAM_OK;
}

int launch_dense(nuslam_batch* c)
{
    hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };
    if (c->prof)
        for (auto& e : ev) { e = get_event(c); if (!e) { g_hip_err = "hipEventCreate failed"; return NUSLAM_E_HIP; } }
    // T = F P into the idle covariance buffer (every row < len of every column is written), then P = T F^T + Qbar
    int rc = dense_predict_launch(c->dtype, c->L, c->ld, c->wF, c->P(), c->Palt(), c->Q, c->stream, ev);
    if (c->prof) {
        c->pending[NUSLAM_K_DENSE_GEMM].emplace_back(ev[0], ev[1]);
        c->pending[NUSLAM_K_DENSE_GEMM].emplace_back(ev[2], ev[3]);
    }
    if (rc) { g_hip_err = "dense_predict_launch failed"; return NUSLAM_E_HIP; }
    HIPCHK(hipGetLastError());
    return NUSLAM_OK;
}

int do_predict(nuslam_batch* h, const TwistArg& tw)
{
    if (h->poisoned) return NUSLAM_E_SYNC;
    h->last_tick = -1;                    // (nuslam_batch_run sets it again behind its ticks: the truth-dependent statistics
                                          // only ever refer to a tick of the resident trace that was really the last one applied)
    int frc = flush_pending(h);
    if (frc) return frc;
    View v = h->view();
    dim3 grid((h->ld + 255) / 256, 1, h->B), block(256);
    int rc = NUSLAM_OK;
    if (h->dense_predict) {
        DISPATCH_T(h, rc = (launch(h, NUSLAM_K_PREDICT, k_predict<T, true>, grid, block, v, tw, (T*)h->P(), h->predict_bookkeeping,
                                   (T*)(h->dense_getA ? h->wF : nullptr))));
        if (!rc) rc = launch_dense(h);
    } else {
        DISPATCH_T(h, rc = (launch(h, NUSLAM_K_PREDICT, k_predict<T, false>, grid, block, v, tw, (T*)h->P(), h->predict_bookkeeping,
                                   (T*)nullptr)));
    }
    if (rc) return rc;
    h->sidx ^= 1; h->state_epoch++;
    h->cidx ^= 1;
    return NUSLAM_OK;
}

// Launches the candidate scan only; the key is consumed by the next do_update(MODE_DA) or by associate_finish().
int do_associate(nuslam_batch* h, const ObsArg& o)
{
    int frc = flush_pending(h);
    if (frc) return frc;
    View v = h->view();
    dim3 grid((h->n + 63) / 64 > 0 ? (h->n + 63) / 64 : 1, h->B), block(64);
    int rc = NUSLAM_OK;
    DISPATCH_T(h, rc = launch(h, NUSLAM_K_ASSOCIATE, k_associate<T>, grid, block, v, o, (const T*)h->P()));
    return rc;
}

int associate_finish(nuslam_batch* h)
{
    h->host_seen_valid = false;
    View v = h->view();
    int rc = launch(h, -1, k_associate_finish, dim3(h->B), dim3(1), v, h->host_word, (const int*)(h->tk_sync ? h->tk_sync + 2 : nullptr));
    if (rc) return rc;
    h->cidx ^= 1;
    h->aslot ^= 1;
    return NUSLAM_OK;
}

// Waves per workgroup of the sweep kernels.  Every workgroup recomputes the correction's head, a serial chain of wave64
// fp64 instructions, and a CU holds eight of these waves: with 8-wave workgroups a CU runs ONE chain and sweeps eight
// tiles together.  That pays when the whole grid is resident at once (one generation: the single-filter case, where a
// launch is latency-bound); with more workgroups than CUs the launch is throughput-bound and 4-wave groups idle fewer
// waves at small L (L = 403: 7 % against 19 %).
int sweep_waves(const nuslam_batch* h, int vec, int strips)
{
    const long long wgs8 = (long long)((h->ld + 64 * vec - 1) / (64 * vec)) * ((strips + 7) / 8) * h->B;
#ifdef NUSLAM_PHASE_CLOCK
    if (const char* f = getenv("NUSLAM_FORCE_WAVES")) return atoi(f) == 4 ? 4 : 8;     // experiments only (debug builds)
#endif
    return wgs8 <= h->n_cu ? 8 : 4;
}

int do_update(nuslam_batch* h, const ObsArg& o, int mode, int total)
{
    if (h->poisoned) return NUSLAM_E_SYNC;
    if (h->deferred && mode != MODE_DA) {
        if (h->J == kMaxPending) { int frc = flush_pending(h); if (frc) return frc; }
        View vd = h->view();
        dim3 gridd((h->ld + 255) / 256, 1, h->B), blockd(256);
        int rcd = NUSLAM_OK;
        if (o.ids == nullptr)
            DISPATCH_T(h, rcd = (launch(h, NUSLAM_K_UPDATE_DEFERRED, k_update_deferred<T, true>, gridd, blockd, vd, o, mode,
                                        total, h->J, (const T*)h->P(), h->dU, h->dV)));
        else
            DISPATCH_T(h, rcd = (launch(h, NUSLAM_K_UPDATE_DEFERRED, k_update_deferred<T, false>, gridd, blockd, vd, o, mode,
                                        total, h->J, (const T*)h->P(), h->dU, h->dV)));
        if (rcd) return rcd;
        h->J += 1;
        h->sidx ^= 1; h->state_epoch++;
        h->cidx ^= 1;
        return NUSLAM_OK;
    }
    View v = h->view();
    const int vec = 16 / (int)h->esize();
    const int strips = (h->L + kSweepCW - 1) / kSweepCW;
    const int waves = sweep_waves(h, vec, strips);
    dim3 grid((h->ld + 64 * vec - 1) / (64 * vec), (strips + waves - 1) / waves, h->B), block(64 * waves);
    int rc = NUSLAM_OK;
    const bool inl = (o.ids == nullptr);
#define LAUNCH_UPDATE(MODE_, INL_)                                                                              \
    do {                                                                                                        \
        if (waves == 8)                                                                                         \
            DISPATCH_T(h, rc = (launch(h, NUSLAM_K_UPDATE, k_update<T, kSweepCW, MODE_, INL_, 8>, grid, block, v, o, \
                                       total, (const T*)h->P(), (T*)h->Palt())));                               \
        else                                                                                                    \
            DISPATCH_T(h, rc = (launch(h, NUSLAM_K_UPDATE, k_update<T, kSweepCW, MODE_, INL_, 4>, grid, block, v, o, \
                                       total, (const T*)h->P(), (T*)h->Palt())));                               \
    } while (0)
    if (mode == MODE_DA) LAUNCH_UPDATE(MODE_DA, true);
    else if (mode == MODE_FORCE) { if (inl) LAUNCH_UPDATE(MODE_FORCE, true); else LAUNCH_UPDATE(MODE_FORCE, false); }
    else { if (inl) LAUNCH_UPDATE(MODE_KNOWN, true); else LAUNCH_UPDATE(MODE_KNOWN, false); }
#undef LAUNCH_UPDATE
    if (rc) return rc;
    h->sidx ^= 1; h->state_epoch++;
    h->cidx ^= 1;
    h->pidx ^= 1;
    if (mode == MODE_DA) h->aslot ^= 1;
    return NUSLAM_OK;
}

int do_update2(nuslam_batch* h, const ObsArg& o1, const ObsArg& o2)
{
    View v = h->view();
    const int vec = 16 / (int)h->esize();
    const int strips = 1 + (h->L - kSweepCW / 2 + kSweepCW - 1) / kSweepCW;    // strip 0 is half as wide (ekf_update2.h)
    const int waves = sweep_waves(h, vec, strips);
    dim3 grid((h->ld + 64 * vec - 1) / (64 * vec), (strips + waves - 1) / waves, h->B), block(64 * waves);
    int rc = NUSLAM_OK;
    if (waves == 8)
        DISPATCH_T(h, rc = (launch(h, NUSLAM_K_UPDATE2, k_update2<T, 8>, grid, block, v, o1, o2, (const T*)h->P(), (T*)h->Palt())));
    else
        DISPATCH_T(h, rc = (launch(h, NUSLAM_K_UPDATE2, k_update2<T, 4>, grid, block, v, o1, o2, (const T*)h->P(), (T*)h->Palt())));
    if (rc) return rc;
    h->sidx ^= 1; h->state_epoch++;
    h->cidx ^= 1;
    h->pidx ^= 1;
    return NUSLAM_OK;
}

// The tick pipeline moves P once per tick instead of once per pair of corrections, at the price of a serial chain
// (~2.4 us per correction, one workgroup per filter) and the strip kernel.  With many filters those run side by side and
// the pass over P dominates (measured 2x at 1024 x N = 200); for ONE filter the chain is exposed and the gain is
// smaller (N = 1000, 16 markers: 105 us per tick against 121 us for eight pair launches).
bool tick_pipeline_pays(const nuslam_batch* h, int m)
{
    if (h->tick_mode >= 0) return h->tick_mode >= 1;
    return m >= 4;                                  // (fewer markers: the fixed cost of three launches is not recovered)
}

int set_rank_attributes();
int ensure_state_mirror(nuslam_batch* h);

int ensure_tick_buffers(nuslam_batch* h)
{
    if (h->tk_ready) return NUSLAM_OK;
    // (a failure partway leaves tk_ready unset: the next call frees what the failed one got and starts over)
    {
        void* part[] = { h->tk_plan, h->tk_K, h->tk_R, h->tk_V, h->tk_pub, h->tk_tagK, h->tk_tagV };
        for (void* p : part)
            if (p) (void)hipFree(p);
        h->tk_plan = nullptr; h->tk_K = h->tk_R = h->tk_V = nullptr; h->tk_pub = nullptr; h->tk_tagK = h->tk_tagV = nullptr;
    }
    const int big = 160 * 1024 - 1024;          // gfx950: 160 KB of LDS per CU
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tick_apply<double, 8, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tick_apply_units<8>), hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tick_apply<double, 4, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tick_apply<float, 8, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tick_apply<float, 4, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tick_panels<double, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tick_panels<float, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tick_panels<double, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tick_panels<float, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIPCHK(hipMalloc(&h->tk_plan, sizeof(TickStep) * (size_t)h->B * kTickJ));
    HIPCHK(hipMalloc(&h->tk_K, sizeof(double) * ((size_t)h->B * kTickJ * 2 * h->ld + kTickDump)));
    HIPCHK(hipMalloc(&h->tk_R, sizeof(double) * ((size_t)h->B * kTickJ * 5 * h->ld + kTickDump)));
    HIPCHK(hipMalloc(&h->tk_V, sizeof(double) * (size_t)h->B * kTickJ * 2 * h->ld));
    // the rank-2m pass masks the strips of skipped corrections by multiplication: what it reads must always be finite
    HIPCHK(hipMemsetAsync(h->tk_K, 0, sizeof(double) * ((size_t)h->B * kTickJ * 2 * h->ld + kTickDump), h->stream));
    HIPCHK(hipMemsetAsync(h->tk_V, 0, sizeof(double) * (size_t)h->B * kTickJ * 2 * h->ld, h->stream));
    { int rc = set_rank_attributes(); if (rc) return rc; }
    HIPCHK(hipMalloc(&h->tk_pub, sizeof(int) * kPubWords * (size_t)h->B));
    HIPCHK(hipMemsetAsync(h->tk_pub, 0, sizeof(int) * kPubWords * (size_t)h->B, h->stream));
    if (h->B == 1) {                                 // (the fused launch is for one filter)
        const size_t tb = sizeof(long long) * 2 * kTickJ * (size_t)h->ld * 2;
        HIPCHK(hipMalloc(&h->tk_tagK, tb));
        HIPCHK(hipMalloc(&h->tk_tagV, tb));
        HIPCHK(hipMemsetAsync(h->tk_tagK, 0, tb, h->stream));
        HIPCHK(hipMemsetAsync(h->tk_tagV, 0, tb, h->stream));
    }
    if (!h->tk_sync) {                             // {chain, next} completion counters, expired waits, probe words
        HIPCHK(hipMalloc(&h->tk_sync, sizeof(int) * 8));
        HIPCHK(hipMemsetAsync(h->tk_sync, 0, sizeof(int) * 8, h->stream));
    }
    h->seq_pub = h->seq_gather = h->seq_pred = 0;
    h->seq_tag = 0;
    h->tk_ready = true;
    return NUSLAM_OK;
}

// forced_init != null: the markers are update() calls the caller really made (lazy class API): the device takes no decision of
// slam.cpp:295-316 for them; forced_init[i] != 0: initializeLandmark came in front of marker i
TickObs make_tick_obs(const nuslam_batch* h, const ObsArg& base, int i0, int m, const int* host_ids, const double* host_mx,
                      const double* host_my, const unsigned char* forced_init = nullptr)
{
    TickObs o;
    o.forced = forced_init ? 1 : 0;
    o.init_mask = 0u;
    o.J = (m - i0) < kTickJ ? (m - i0) : kTickJ;
    o.a = host_mx ? nullptr : base.a; o.b = host_mx ? nullptr : base.b;
    o.ids = host_ids ? nullptr : base.ids;
    o.stride = base.stride; o.off = base.off + i0;
    o.cartesian = base.cartesian;
    o.log_slot0 = h->id_log ? i0 : -1;
    for (int k = 0; k < kTickJ; ++k) {
        const bool in = k < o.J;
        o.a0[k] = (in && host_mx) ? host_mx[i0 + k] : (in ? base.a0 : 0.0);
        o.b0[k] = (in && host_my) ? host_my[i0 + k] : (in ? base.b0 : 0.0);
        o.id0[k] = (in && host_ids) ? host_ids[i0 + k] : (in ? base.id0 : 0);
        if (in && forced_init && forced_init[i0 + k]) o.init_mask |= 1u << k;
    }
    return o;
}

// A contiguous range of the handle's filters and the stream its launches go to; default: all of them
struct Sub {
    int g0, Bg;
    hipStream_t st;
};
Sub whole(const nuslam_batch* h) { return Sub{ 0, h->B, h->stream }; }
// the view of filters [g0, g0 + Bg): every per-filter pointer moved up, B = Bg (kernels index their filters from 0)
View sub_view(const View& v, const Sub& sb)
{
    View w = v;
    w.B = sb.Bg;
    w.s_in += (size_t)sb.g0 * v.ld; w.s_out += (size_t)sb.g0 * v.ld;
    w.c_in += (size_t)sb.g0 * C_WORDS; w.c_out += (size_t)sb.g0 * C_WORDS;
    w.cur_id += sb.g0; w.akey += 2 * (size_t)sb.g0;
    if (w.id_log) w.id_log += (size_t)sb.g0 * v.log_stride;
    return w;
}
template <typename T> T* filt(void* base, const nuslam_batch* h, const Sub& sb) { return (T*)base + (size_t)sb.g0 * h->p_stride; }

// ---- the rank-2m pass (ekf_rank.h): tile shapes <RB, CB, WR, WC> per storage type
template <typename T, int RB, int CB, int WR, int WC>
int launch_rank_t(nuslam_batch* h, const View& v, int J, const TickStep* plan, int check_init, const Sub& sb)
{
    typedef RankTile<T, RB, CB, WR, WC> TL;
    const int tiles_r = (h->ld + TL::WROWS - 1) / TL::WROWS, tiles_c = (h->L + TL::WCOLS - 1) / TL::WCOLS;
    const bool by_filter = sb.Bg >= 8;                // a filter per XCD (its strips in one L2), else tile rows per XCD
    const dim3 grid = by_filter ? dim3((unsigned)(((sb.Bg + 7) / 8) * 8 * tiles_r * tiles_c))
                                : dim3((unsigned)(((tiles_r + 7) / 8) * 8 * tiles_c), (unsigned)sb.Bg);
    const size_t so = (size_t)sb.g0 * kTickJ * 2 * h->ld;
    return launch_on(h, sb.st, NUSLAM_K_TICK_RANK, k_tick_rank<T, RB, CB, WR, WC>, grid, dim3(TL::NT), TL::lds_bytes, sub_view(v, sb), J,
                     plan + (size_t)sb.g0 * kTickJ, (const double*)h->tk_K + so, (const double*)h->tk_V + so,
                     (const T*)filt<T>(h->P(), h, sb), filt<T>(h->Palt(), h, sb), check_init, tiles_r, tiles_c, by_filter ? 1 : 0);
}
#define RANK_TILES(X)                                                                           \
    X(double, 4, 1, 1, 4) X(double, 1, 4, 4, 1) X(double, 2, 2, 2, 2) X(double, 2, 2, 1, 4)       \
    X(double, 1, 2, 2, 2) X(double, 1, 1, 2, 2) X(double, 2, 1, 2, 2) X(double, 1, 2, 1, 4) X(double, 2, 3, 2, 2)   \
    X(float, 2, 2, 1, 4) X(float, 1, 4, 4, 1) X(float, 1, 4, 2, 2) X(float, 2, 1, 1, 4)
int set_rank_attributes()
{
#define X(T, RB, CB, WR, WC)                                                                                         \
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tick_rank<T, RB, CB, WR, WC>),                        \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)RankTile<T, RB, CB, WR, WC>::lds_bytes));
    RANK_TILES(X)
#undef X
    return NUSLAM_OK;
}
int launch_rank(nuslam_batch* h, const View& v, int J, const TickStep* plan, int check_init, const Sub& sb)
{
    // tile 0 is the measured best per storage type (profiles/r03/pass_tiles.txt); 1..3 stay selectable for measurement
    // (nuslam_batch_set_pass_variant(10 + k)) -- every tile computes the same bits
    if (h->dtype == NUSLAM_F64) {
        switch (h->rank_tile) {
        case 1: return launch_rank_t<double, 4, 1, 1, 4>(h, v, J, plan, check_init, sb);
        case 2: return launch_rank_t<double, 1, 4, 4, 1>(h, v, J, plan, check_init, sb);
        case 3: return launch_rank_t<double, 2, 2, 1, 4>(h, v, J, plan, check_init, sb);
        // smaller workgroup tiles (a 403 x 403 matrix under 128 x 64 tiles is covered 1.41 times) and the fused launch's 128 x 96
        case 4: return launch_rank_t<double, 1, 2, 2, 2>(h, v, J, plan, check_init, sb);     // 64 x 64
        case 5: return launch_rank_t<double, 1, 1, 2, 2>(h, v, J, plan, check_init, sb);     // 64 x 32
        case 6: return launch_rank_t<double, 2, 1, 2, 2>(h, v, J, plan, check_init, sb);     // 128 x 32
        case 7: return launch_rank_t<double, 1, 2, 1, 4>(h, v, J, plan, check_init, sb);     // 32 x 128
        case 8: return launch_rank_t<double, 2, 3, 2, 2>(h, v, J, plan, check_init, sb);     // 128 x 96
        default:
            // 128 x 64 tiles cover a 403 x 403 matrix (N = 200) 1.41 times, 64 x 64 tiles 1.24 times: 3 % faster there (profiles/r04/pass_tiles.txt)
            if (h->L <= 600) return launch_rank_t<double, 1, 2, 2, 2>(h, v, J, plan, check_init, sb);
            return launch_rank_t<double, 2, 2, 2, 2>(h, v, J, plan, check_init, sb);
        }
    }
    switch (h->rank_tile) {
    case 1: return launch_rank_t<float, 2, 2, 1, 4>(h, v, J, plan, check_init, sb);
    case 2: return launch_rank_t<float, 1, 4, 4, 1>(h, v, J, plan, check_init, sb);
    case 3: return launch_rank_t<float, 1, 4, 2, 2>(h, v, J, plan, check_init, sb);
    default: return launch_rank_t<float, 2, 1, 1, 4>(h, v, J, plan, check_init, sb);
    }
}

// The pass over P of one round from the plan and the strips in tk_K / tk_R / tk_V.  may_init: the host cannot rule out a
// first sighting in this round (it can for known ids at or below its proven lower bound of `seen`): then the rank-2m pass
// checks the plan's flags per filter and the exact chain is launched behind it for the filters it left.
int launch_pass(nuslam_batch* h, const View& v, int J, const TickStep* plan, bool shares_chip, bool may_init, const Sub& sb)
{
    int only_if_init = 0;
    if (h->pass_mode == 0) {
        int rc = launch_rank(h, v, J, plan, may_init ? 1 : 0, sb);
        if (rc || !may_init) return rc;
        only_if_init = 1;
    }
    const int vec = 16 / (int)h->esize();                      // rows per lane: 16 bytes' worth
    const int strips = (h->L + kSweepCW - 1) / kSweepCW;
    const int waves = sweep_waves(h, vec, strips);             // (by the handle's size: every group takes the same kernel)
    int rc = NUSLAM_OK;
    dim3 grid((h->ld + 64 * vec - 1) / (64 * vec), (strips + waves - 1) / waves, sb.Bg), block(64 * waves);
    // one resident generation (waves == 8): the gain rows staged in LDS, shared by the workgroup's waves; many
    // generations (waves == 4): gain rows from L2, R alone in LDS, three workgroups per CU (see k_tick_apply)
    const bool k_in_lds = waves == 8 || h->dtype != NUSLAM_F64;
    const size_t lds = sizeof(double) * ((k_in_lds ? (size_t)J * 2 * 64 * vec : 0) + (size_t)waves * J * 5 * kSweepCW);
    const View w = sub_view(v, sb);
    const TickStep* pl = plan + (size_t)sb.g0 * kTickJ;
    const double* Kp = (const double*)h->tk_K + (size_t)sb.g0 * kTickJ * 2 * h->ld;
    const double* Rp = (const double*)h->tk_R + (size_t)sb.g0 * kTickJ * 5 * h->ld;
    if (h->dtype == NUSLAM_F64) {
        const double* Pi = filt<double>(h->P(), h, sb);
        double* Po = filt<double>(h->Palt(), h, sb);
        // (an overlapped run keeps k_tick_apply: its 164 registers leave room for the chain's workgroup on the same SIMDs; beside
        // the two-unit kernel's 206 the chain had to wait for a CU of its own and the pass took 51 us instead of 40)
        if (waves == 8 && h->apply_units && !shares_chip)
            rc = launch_on(h, sb.st, NUSLAM_K_TICK_APPLY, k_tick_apply_units<8>, grid, block,
                           sizeof(double) * (size_t)J * (5 * 128 + 8 * 5 * 8), w, J, pl, Kp, Rp, Pi, Po, only_if_init);
        else if (waves == 8)
            rc = launch_on(h, sb.st, NUSLAM_K_TICK_APPLY, k_tick_apply<double, 8, 2, true>, grid, block, lds, w, J, pl, Kp, Rp, Pi, Po, only_if_init);
        else
            rc = launch_on(h, sb.st, NUSLAM_K_TICK_APPLY, k_tick_apply<double, 4, 2, false>, grid, block, lds, w, J, pl, Kp, Rp, Pi, Po, only_if_init);
    } else {
        const float* Pi = filt<float>(h->P(), h, sb);
        float* Po = filt<float>(h->Palt(), h, sb);
        if (waves == 8)
            rc = launch_on(h, sb.st, NUSLAM_K_TICK_APPLY, k_tick_apply<float, 8, 4, true>, grid, block, lds, w, J, pl, Kp, Rp, Pi, Po, only_if_init);
        else
            rc = launch_on(h, sb.st, NUSLAM_K_TICK_APPLY, k_tick_apply<float, 4, 4, true>, grid, block, lds, w, J, pl, Kp, Rp, Pi, Po, only_if_init);
    }
    return rc;
}

// Known-id markers of one tick of filter b, in order: can one of them be the FIRST correction of its landmark (the one that
// cancels the INT_MAX diagonal; rounds holding one go through the exact chain)?  Decided against what the host can prove:
// a landmark counts as corrected once a known-id marker for it has certainly reached update() -- resolve(): 1 <= id <= n,
// the marker loop not broken, id <= total_landmarks (above it the marker is either a first sighting or the `break`,
// slam.cpp:295-316, which the host cannot tell apart: it then stops marking for the rest of the tick).
bool tick_may_init(nuslam_batch* h, int b, const int* ids, int m, int total)
{
    unsigned char* tb = h->touched.data() + (size_t)b * (h->n + 1);
    bool may = false, maybe_brk = false;
    for (int i = 0; i < m; ++i) {
        const int id = ids[i];
        if (id < 1 || id > h->n) continue;                 // skipped marker / bad id: nothing is applied
        if (!tb[id]) may = true;
        if (maybe_brk) continue;
        if (id > total) { maybe_brk = true; continue; }
        tb[id] = 1;
    }
    return may;
}
// the same for every filter of the handle: one id list for all (host_ids) or filter b's at pf_ids + b * pf_stride
bool tick_may_init_all(nuslam_batch* h, const int* host_ids, const int* pf_ids, long long pf_stride, int m, int total)
{
    if (!host_ids && !pf_ids) return true;
    bool may = false;
    for (int b = 0; b < h->B; ++b)
        may = tick_may_init(h, b, host_ids ? host_ids : pf_ids + (size_t)b * pf_stride, m, total) || may;
    return may;
}

// strips + the pass over P of one round, on the handle's stream, from `plan`
// `between`, if given, is enqueued between the strips and the pass (overlapped runs: the signal for the next chain)
template <typename F>
int launch_strips_and_pass(nuslam_batch* h, const View& v, const TickObs& o, const TickStep* plan, bool compact, bool may_init, F between)
{
    int rc = NUSLAM_OK;
    double* vbuf = h->pass_mode == 0 ? h->tk_V : nullptr;       // the rank-2m pass's second factor, formed beside R
    // the five prior rows R_s themselves are read by the exact chain and by the next tick's carry only: when the host can prove
    // that the rank-2m pass takes every filter of this round, they are not stored at all (30 of 394 us at 1024 x N = 200)
    double* rbuf = (h->pass_mode == 0 && !may_init && !compact) ? nullptr : h->tk_R;
    // the strips' form (ekf_tick.h): exact chain / rank form where the round's flags allow it (1) / rank form, proven by the host (2: LDS
    // for the entries' heads only)
    const int rank_ok = h->pass_mode != 0 ? 0 : (may_init ? 1 : 2);
    const size_t plan_lds = sizeof(double) * (rank_ok == 2 ? kPlanHeadWords : kPlanExactWords) * (size_t)o.J;
    if (rank_ok == 2 && !rbuf && vbuf && !compact && h->strips_lane && (long long)((h->ld + 63) / 64) * h->B > 4 * h->n_cu)
        // many filters, rank form proven, no R strips wanted: a lane per (index, role), coefficients through the scalar cache
        DISPATCH_T(h, rc = (launch(h, NUSLAM_K_TICK_PANELS, k_tick_strips_lane<T>, dim3(8 * ((h->B + 7) / 8) * ((h->ld + 63) / 64)), dim3(128),
                                   v, o, (const T*)h->P(), plan, h->tk_K, vbuf)));
    else if ((long long)((h->ld + 31) / 32) * h->B <= h->n_cu)   // few filters: 4-wave groups, one wave per SIMD, on twice the CUs
        DISPATCH_T(h, rc = (launch_lds(h, NUSLAM_K_TICK_PANELS, k_tick_panels<T, 32>, dim3((h->ld + 31) / 32, h->B), dim3(256),
                                       plan_lds, v, o, (const T*)h->P(), plan, h->tk_K, rbuf, vbuf, (const int*)(compact ? h->tk_posmap : nullptr),
                                       h->tk_KU, h->tk_RU, h->tk_SU, rank_ok)));
    else
        DISPATCH_T(h, rc = (launch_lds(h, NUSLAM_K_TICK_PANELS, k_tick_panels<T, 64>, dim3((h->ld + 63) / 64, h->B), dim3(512),
                                       plan_lds, v, o, (const T*)h->P(), plan, h->tk_K, rbuf, vbuf, (const int*)(compact ? h->tk_posmap : nullptr),
                                       h->tk_KU, h->tk_RU, h->tk_SU, rank_ok)));
    if (rc) return rc;
    rc = between();
    if (rc) return rc;
    return launch_pass(h, v, o.J, plan, compact, may_init, whole(h));
}

int ensure_da_buffers(nuslam_batch* h)
{
    if (h->da_mem) return NUSLAM_OK;
    const size_t B = h->B, ld = h->ld, n = h->n;
    const int nwg = (h->ld - 3 + kDaOwn - 1) / kDaOwn > 0 ? (h->ld - 3 + kDaOwn - 1) / kDaOwn : 1;
    // one allocation: 2 x (TR, TC) [B][3][ld], 2 x TD [B][4][n], 2 x DS [B][ld], Z [B][2][kTickJ]; then the int arrays
    // ... AP 2 x [B][n][16], keyt [B][kTickJ][nwg] (8-byte words)
    const size_t nd = 2 * (2 * B * 3 * ld) + 2 * B * 4 * n + 2 * B * ld + B * 2 * kTickJ + 2 * B * n * 16 + B * kTickJ * (size_t)nwg * kDaSlotStride + 8 * B;
    const size_t ni = 2 * B * C_WORDS + B * kTickJ * (size_t)nwg;
    HIPCHK(hipMalloc(&h->da_mem, nd * sizeof(double) + ni * sizeof(int)));
    HIPCHK(hipMemsetAsync(h->da_mem, 0, nd * sizeof(double) + ni * sizeof(int), h->stream));
    double* p = (double*)h->da_mem;
    for (int k = 0; k < 2; ++k) { h->da.TR[k] = p; p += B * 3 * ld; }
    for (int k = 0; k < 2; ++k) { h->da.TC[k] = p; p += B * 3 * ld; }
    for (int k = 0; k < 2; ++k) { h->da.TD[k] = p; p += B * 4 * n; }
    for (int k = 0; k < 2; ++k) { h->da.DS[k] = p; p += B * ld; }
    h->da.Z = p; p += B * 2 * kTickJ;
    for (int k = 0; k < 2; ++k) { h->da.AP[k] = p; p += B * n * 16; }
    h->da.keyt = (long long*)p; p += B * kTickJ * (size_t)nwg * kDaSlotStride;
    h->da.fwd = (long long*)p; p += 8 * B;
    int* q = (int*)p;
    for (int k = 0; k < 2; ++k) { h->da.DC[k] = q; q += B * C_WORDS; }
    h->da.keyp = q;
    h->da.nwg = nwg;
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_da_round<double, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kDaRoundLds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_da_round<float, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kDaRoundLds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_da_round<double, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kDaRoundLds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_da_round<float, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kDaRoundLds));
    return NUSLAM_OK;
}

// The markers of an unknown-association tick in rounds of up to kTickJ (ekf_da.h): k_da_begin, one k_da_step per marker,
// then the one pass over P.
int do_da_rounds(nuslam_batch* h, const ObsArg& base, int m, int total, const double* host_mx, const double* host_my)
{
    { int erc = ensure_tick_buffers(h); if (erc) return erc; }
    { int erc = ensure_da_buffers(h); if (erc) return erc; }
    for (int i0 = 0; i0 < m; i0 += kTickJ) {
        const TickObs o = make_tick_obs(h, base, i0, m, nullptr, host_mx, host_my);
        View v = h->view();
        int rc = NUSLAM_OK;
        double* vbuf = h->pass_mode == 0 ? h->tk_V : nullptr;   // V = H R beside R, for the rank-2m pass
        const dim3 grid(h->da.nwg, h->B), block(256);
        // all workgroups of the resident kernel must be on the chip together (they wait for each other): one per CU at most
        const bool resident = h->tick_mode != 2 && (long long)h->da.nwg * h->B <= h->n_cu;
        if (resident) {
            DISPATCH_T(h, rc = (launch_lds(h, NUSLAM_K_DA_STEP, k_da_round<T, false>, grid, block, (size_t)kDaRoundLds, v, o, total,
                                           (const T*)h->P(), h->da, h->tk_plan, h->tk_K, h->tk_R, vbuf, (int)h->da_round_tag, DaServe{})));
            if (rc) return rc;
            h->da_round_tag += kTickJ + 1;
        } else {
            DISPATCH_T(h, rc = (launch(h, NUSLAM_K_DA_BEGIN, k_da_begin<T>, grid, block, v, o, (const T*)h->P(), h->da)));
            if (rc) return rc;
            for (int st = 0; st < o.J; ++st) {
                DISPATCH_T(h, rc = (launch(h, NUSLAM_K_DA_STEP, k_da_step<T>, grid, block, v, o, st, st + 1 == o.J ? 1 : 0, total,
                                           (const T*)h->P(), h->da, h->tk_plan, h->tk_K, h->tk_R, vbuf)));
                if (rc) return rc;
            }
        }
        rc = launch_pass(h, v, o.J, h->tk_plan, false, true, whole(h));  // (association decides on the device whether a landmark is new)
        if (rc) return rc;
        h->sidx ^= 1; h->state_epoch++;
        h->cidx ^= 1;
        h->pidx ^= 1;
    }
    return NUSLAM_OK;
}

// The markers of a known-id tick in rounds of up to kTickJ: k_tick_chain (serial part, one workgroup per filter),
// k_tick_panels (the O(len) strips, one quad per state index), k_tick_apply (the one pass over P).
// Can this handle's known-id tick take chain + strips (+ the tick's predict) as ONE launch?  Its whole grid must be resident.
bool front_fits(const nuslam_batch* h, bool with_predict)
{
    const long long per_filter = 1 + (h->ld + 31) / 32 + (with_predict ? (h->ld + 255) / 256 : 0);
    return h->front && per_filter * h->B <= h->n_cu;
}

// k_tick_fused / k_run_fused: every workgroup of the grid must be resident at once (two waves per SIMD: two workgroups per CU)
bool fused_fits(const nuslam_batch* h)
{
    int n_pass = 0;
    DISPATCH_T(h, n_pass = FusedTile<T>::blocks(h->ld, h->L));
    return 1 + (h->ld + 255) / 256 + (h->ld + 31) / 32 + n_pass <= 2 * h->n_cu;
}

// fused_predict != null: no predict kernel ran for this tick -- its predict rides in the first round's k_tick_front launch
int do_tick_rounds(nuslam_batch* h, const ObsArg& base, int m, int total, const int* host_ids, const double* host_mx,
                   const double* host_my, const int* pf_ids, long long pf_stride, const TwistArg* fused_predict = nullptr,
                   const unsigned char* forced_init = nullptr)
{
    { int erc = ensure_tick_buffers(h); if (erc) return erc; }
    bool may_init = tick_may_init_all(h, host_ids, pf_ids, pf_stride, m, total);   // (`seen` is cached per TICK, not per round)
    if (forced_init)                                    // an initializeLandmark the caller made: the round goes through the exact chain
        for (int i = 0; i < m; ++i) may_init = may_init || forced_init[i] != 0;
    // With the rank-2m pass the strips carry their panels in rank form in every round without a first sighting.  k_tick_panels decides
    // that per filter from the round's own flags; the strips that FOLLOW the chain (k_tick_front) cannot -- they start before the
    // round's flags exist -- and take the form the host can prove: the one-launch form is used only when it can (every filter then
    // takes the form it would take in any other launch form: same bits).
    const bool rank_strips = h->pass_mode == 0;
    const bool front_ok = !rank_strips || !may_init;
    if (fused_predict && !(front_ok && front_fits(h, true))) {          // (the caller left the predict to the first round's launch)
        int prc = do_predict(h, *fused_predict);
        if (prc) return prc;
        fused_predict = nullptr;
    }
    for (int i0 = 0; i0 < m; i0 += kTickJ) {
        const TickObs o = make_tick_obs(h, base, i0, m, host_ids, host_mx, host_my, forced_init);
        View v = h->view();
        int rc = NUSLAM_OK;
        const int strip_wgs = (h->ld + 31) / 32;
        const bool with_predict = fused_predict != nullptr && i0 == 0;
        if (front_ok && front_fits(h, with_predict)) {
            // chain and strips in ONE launch: the strip workgroups follow the chain entry by entry (k_tick_front); in the
            // tick's first round the predict rides along as workgroups of its own
            double* vbuf = h->pass_mode == 0 ? h->tk_V : nullptr;
            const int n_pred = with_predict ? (h->ld + 255) / 256 : 0;
            TickPublish pub;
            pub.flag = h->tk_pub; pub.base = (int)h->seq_pub; pub.predict = with_predict ? 1 : 0;
            pub.gbase = (int)(h->seq_gather + 1u); pub.pbase = (int)(h->seq_pred + (unsigned)n_pred);
            pub.rank_panels = rank_strips ? 1 : 0;
            if (with_predict) pub.tw = *fused_predict;
            else { pub.tw.tw = nullptr; pub.tw.stride = pub.tw.off = 0; pub.tw.dth0 = pub.tw.dx0 = 0.0; }
            // ONE filter, the rank-2m pass, no first sighting possible: the pass rides in the same launch (ekf_fused.h)
            const bool fused = h->fuse_pass && h->B == 1 && h->pass_mode == 0 && h->rank_tile == 0 && !may_init && fused_fits(h);
            if (fused) {
                h->seq_tag += 1u;
                if (h->seq_tag == 0u) h->seq_tag = 1u;
                TickTagged tg;
                tg.tagK = h->tk_tagK; tg.tagV = h->tk_tagV; tg.tag = (int)h->seq_tag;
                tg.mirror = nullptr; tg.mtags = nullptr; tg.mseq = 0;
                if (h->lazy.on && i0 + kTickJ >= m && strip_wgs <= 512 && ensure_state_mirror(h) == NUSLAM_OK) {
                    // (the tick's last round: its strips leave the state the caller's next getStateVector() reads)
                    h->st_seq += 1;
                    tg.mirror = h->st_host; tg.mtags = h->st_tags; tg.mseq = h->st_seq;
                }
                int n_pass = 0;
                DISPATCH_T(h, n_pass = FusedTile<T>::blocks(h->ld, h->L));
                DISPATCH_T(h, rc = (launch(h, NUSLAM_K_TICK_CHAIN, k_tick_fused<T>, dim3(1 + n_pred + strip_wgs + n_pass, 1), dim3(256), v, o, total,
                                           (T*)h->P(), (T*)h->Palt(), h->tk_plan, h->tk_K, h->tk_R, vbuf, pub, tg, n_pred, strip_wgs, o.J, h->tk_sync + 2)));
                if (rc) return rc;
                if (tg.mirror) { h->st_ntags = strip_wgs; h->mirror_epoch = h->state_epoch + 1; }    // (+1: the flip below)
            } else {
                DISPATCH_T(h, rc = (launch(h, NUSLAM_K_TICK_CHAIN, k_tick_front<T>, dim3(1 + n_pred + strip_wgs, h->B), dim3(256), v, o, total,
                                           (T*)h->P(), h->tk_plan, h->tk_K, h->tk_R, vbuf, pub, n_pred, h->tk_sync + 2)));
                if (rc) return rc;
            }
            h->seq_pub += 2u * kTickJ;
            if (with_predict) { h->seq_gather += 1u; h->seq_pred += (unsigned)n_pred; }
            if (!fused) {
                rc = launch_pass(h, v, o.J, h->tk_plan, false, may_init, whole(h));
                if (rc) return rc;
            }
        } else {
            DISPATCH_T(h, rc = (launch(h, NUSLAM_K_TICK_CHAIN, k_tick_chain<T, false>, dim3(h->B), dim3(256), v, o, total,
                                       (const T*)h->P(), h->tk_plan, TickCarry{}, (int*)nullptr, (int*)nullptr)));
            if (rc) return rc;
            rc = launch_strips_and_pass(h, v, o, h->tk_plan, false, may_init, [] { return (int)NUSLAM_OK; });
            if (rc) return rc;
        }
        h->sidx ^= 1; h->state_epoch++;
        h->cidx ^= 1;
        h->pidx ^= 1;
    }
    return NUSLAM_OK;
}

// ---- groups of filters on streams of their own
int group_count(const nuslam_batch* h)
{
    int g = h->groups < 0 ? (h->B >= 512 ? 2 : 1) : h->groups;        // (measured at 1024 x N = 200: 2 groups 728 us per tick, 1: 781, 3: 752, 4: 799)
    if (g > 4) g = 4;
    if (g > h->B) g = h->B;
    return g < 1 ? 1 : g;
}
Sub group_of(nuslam_batch* h, int g, int G)
{
    const int g0 = (int)((long long)h->B * g / G), g1 = (int)((long long)h->B * (g + 1) / G);
    return Sub{ g0, g1 - g0, g == 0 ? h->stream : h->gstream[g] };
}
int ensure_groups(nuslam_batch* h, int G)
{
    for (int g = 1; g < G; ++g)
        if (!h->gstream[g]) {
            HIPCHK(hipStreamCreateWithFlags(&h->gstream[g], hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&h->gev[g], hipEventDisableTiming));
        }
    if (!h->gev0) HIPCHK(hipEventCreateWithFlags(&h->gev0, hipEventDisableTiming));
    for (int g = 0; g < G; ++g)
        if (!h->gpass[g]) HIPCHK(hipEventCreateWithFlags(&h->gpass[g], hipEventDisableTiming));
    return NUSLAM_OK;
}
// the group streams start behind everything the handle's stream holds / the handle's stream goes on behind all of them
int groups_fork(nuslam_batch* h, int G)
{
    HIPCHK(hipEventRecord(h->gev0, h->stream));
    for (int g = 1; g < G; ++g) HIPCHK(hipStreamWaitEvent(h->gstream[g], h->gev0, 0));
    for (int g = 0; g < 4; ++g) h->gpass_set[g] = false;
    return NUSLAM_OK;
}
int groups_join(nuslam_batch* h, int G)
{
    for (int g = 1; g < G; ++g) {
        HIPCHK(hipEventRecord(h->gev[g], h->gstream[g]));
        HIPCHK(hipStreamWaitEvent(h->stream, h->gev[g], 0));
    }
    return NUSLAM_OK;
}

// One known-id tick of a large batch as G groups of filters on G streams: predict, chain, strips and the pass of each group are
// enqueued on the group's own stream, so the groups' phases slide against each other on the chip.  Same kernels, same per-filter
// data as the ungrouped tick: same bits.  (The caller forks / joins the streams around a run of such ticks.)
int do_tick_grouped(nuslam_batch* h, int G, const TwistArg& tw, const ObsArg& base, int m, int total, const int* host_ids,
                    const int* pf_ids, long long pf_stride)
{
    if (h->poisoned) return NUSLAM_E_SYNC;
    h->last_tick = -1;
    { int erc = ensure_tick_buffers(h); if (erc) return erc; }
    int rc = NUSLAM_OK;
    {   // ---- predict (slam_library.cpp:65-148)
        const View v = h->view();
        for (int g = 0; g < G && !rc; ++g) {
            const Sub sb = group_of(h, g, G);
            TwistArg twg = tw;
            if (twg.tw) twg.off += (long long)sb.g0 * twg.stride;
            dim3 grid((h->ld + 255) / 256, 1, sb.Bg), block(256);
            DISPATCH_T(h, rc = (launch_on(h, sb.st, NUSLAM_K_PREDICT, k_predict<T, false>, grid, block, 0, sub_view(v, sb), twg,
                                          filt<T>(h->P(), h, sb), h->predict_bookkeeping, (T*)nullptr)));
        }
        if (rc) return rc;
        h->sidx ^= 1; h->state_epoch++;
        h->cidx ^= 1;
    }
    const bool may_init = tick_may_init_all(h, host_ids, pf_ids, pf_stride, m, total);
    double* vbuf = h->pass_mode == 0 ? h->tk_V : nullptr;
    const int rank_ok = h->pass_mode != 0 ? 0 : (may_init ? 1 : 2);
    for (int i0 = 0; i0 < m; i0 += kTickJ) {
        const TickObs o = make_tick_obs(h, base, i0, m, host_ids, nullptr, nullptr);
        const View v = h->view();
        for (int g = 0; g < G && !rc; ++g) {
            const Sub sb = group_of(h, g, G);
            const View w = sub_view(v, sb);
            TickObs og = o;
            og.off += (long long)sb.g0 * og.stride;                     // (a broadcast trace has stride 0)
            TickStep* pl = h->tk_plan + (size_t)sb.g0 * kTickJ;
            DISPATCH_T(h, rc = (launch_on(h, sb.st, NUSLAM_K_TICK_CHAIN, k_tick_chain<T, false>, dim3(sb.Bg), dim3(256), 0, w, og, total,
                                          (const T*)filt<T>(h->P(), h, sb), pl, TickCarry{}, (int*)nullptr, (int*)nullptr)));
            if (rc) break;
            double* Kp = h->tk_K + (size_t)sb.g0 * kTickJ * 2 * h->ld;
            double* Rp = (h->pass_mode == 0 && !may_init) ? nullptr : h->tk_R + (size_t)sb.g0 * kTickJ * 5 * h->ld;
            double* Vp = vbuf ? vbuf + (size_t)sb.g0 * kTickJ * 2 * h->ld : nullptr;
            if (rank_ok == 2 && !Rp && Vp && h->strips_lane && (long long)((h->ld + 63) / 64) * sb.Bg > 4 * h->n_cu)
                DISPATCH_T(h, rc = (launch_on(h, sb.st, NUSLAM_K_TICK_PANELS, k_tick_strips_lane<T>, dim3(8 * ((sb.Bg + 7) / 8) * ((h->ld + 63) / 64)), dim3(128), 0,
                                              w, og, (const T*)filt<T>(h->P(), h, sb), (const TickStep*)pl, Kp, Vp)));
            else
                DISPATCH_T(h, rc = (launch_on(h, sb.st, NUSLAM_K_TICK_PANELS, k_tick_panels<T, 64>, dim3((h->ld + 63) / 64, sb.Bg), dim3(512),
                                              sizeof(double) * (rank_ok == 2 ? kPlanHeadWords : kPlanExactWords) * (size_t)og.J, w, og, (const T*)filt<T>(h->P(), h, sb), (const TickStep*)pl, Kp, Rp, Vp,
                                              (const int*)nullptr, (double*)nullptr, (double*)nullptr, (double*)nullptr, rank_ok)));
            if (rc) break;
            // The passes take turns: group g's pass starts when group g-1's (for group 0: the last group's of the previous round) has
            // ended.  Left to themselves the groups run in step -- pass beside pass, strips beside strips --; with the passes in a ring one
            // group's HBM-bound pass runs beside the other groups' chains and strips.
            const int prev = (g + G - 1) % G;
            if (h->group_ring && h->gpass_set[prev]) HIPCHK(hipStreamWaitEvent(sb.st, h->gpass[prev], 0));
            rc = launch_pass(h, v, og.J, h->tk_plan, false, may_init, sb);
            if (rc) break;
            if (h->group_ring) {
                HIPCHK(hipEventRecord(h->gpass[g], sb.st));
                h->gpass_set[g] = true;
            }
        }
        if (rc) return rc;
        h->sidx ^= 1; h->state_epoch++;
        h->cidx ^= 1;
        h->pidx ^= 1;
    }
    h->host_seen_valid = false;
    return NUSLAM_OK;
}

// nuslam_batch_run, ticks [t0, t1) of ONE filter's resident known-id trace, as ONE launch (k_run_fused, ekf_fused.h): the covariance
// stays in the pass workgroups' registers between the ticks.  kRunNotApplicable: this handle / this run takes a launch per tick
// (anything but the default known-id pipeline, a tick that may hold a first sighting, ids the host does not have).
constexpr int kRunNotApplicable = -2000;
int run_fused(nuslam_batch* h, int t0, int t1, int total)
{
    const int nt = t1 - t0, m = h->tr_m;
    if (!(h->run_fused && h->fuse_pass && h->front && h->B == 1 && h->pass_mode == 0 && h->rank_tile == 0 && nt >= 2 && m >= 1 && m <= kTickJ &&
          h->tr_ids && !h->tr_presence_only && tick_pipeline_pays(h, m) && !h->deferred && !h->dense_predict && h->predict_bookkeeping &&
          front_fits(h, true) && fused_fits(h) && !h->id_log))
        return kRunNotApplicable;
    const int* hid = h->h_ids.empty() ? nullptr : h->h_ids.data();
    const int* pfid = h->h_ids_pf.empty() ? nullptr : h->h_ids_pf.data();
    if (!hid && !pfid) return kRunNotApplicable;
    {   // no tick of the run may hold a first sighting: proven on a COPY of the host's record (a launch per tick proves it again, tick by tick)
        const std::vector<unsigned char> keep = h->touched;
        bool may = false;
        for (int t = t0; t < t1 && !may; ++t)
            may = tick_may_init_all(h, hid ? hid + (size_t)t * m : nullptr, pfid ? pfid + (size_t)t * m : nullptr, (long long)h->tr_ticks * m, m, total);
        if (may) { h->touched = keep; return kRunNotApplicable; }
    }
    { int erc = ensure_tick_buffers(h); if (erc) return erc; }
    if (h->poisoned) return NUSLAM_E_SYNC;
    h->last_tick = -1;
    ObsArg base;
    base.a = h->tr_mx; base.b = h->tr_my; base.ids = h->tr_ids;
    base.stride = h->tr_bcast ? 0 : (long long)h->tr_ticks * m;
    base.off = (long long)t0 * m;
    base.a0 = base.b0 = 0.0; base.id0 = 0; base.cartesian = 1; base.log_slot = -1;
    const TickObs o = make_tick_obs(h, base, 0, m, nullptr, nullptr, nullptr);
    View v = h->view();
    const int strip_wgs = (h->ld + 31) / 32, n_pred = (h->ld + 255) / 256;
    TickPublish pub;
    pub.flag = h->tk_pub; pub.base = (int)h->seq_pub; pub.predict = 1;
    pub.gbase = (int)(h->seq_gather + 1u); pub.pbase = (int)(h->seq_pred + (unsigned)n_pred);
    pub.rank_panels = 1;
    pub.tw.tw = h->tr_tw; pub.tw.stride = h->tr_bcast ? 0 : (long long)h->tr_ticks * 2; pub.tw.off = (long long)t0 * 2;
    pub.tw.dth0 = pub.tw.dx0 = 0.0;
    if (h->seq_tag > 0x3fffffffu) {                      // (tags advance by one per tick: start over long before they wrap)
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemsetAsync(h->tk_tagK, 0, sizeof(long long) * 2 * kTickJ * 2 * (size_t)h->ld * h->B, h->stream));
        HIPCHK(hipMemsetAsync(h->tk_tagV, 0, sizeof(long long) * 2 * kTickJ * 2 * (size_t)h->ld * h->B, h->stream));
        h->seq_tag = 0u;
    }
    TickTagged tg;
    tg.tagK = h->tk_tagK; tg.tagV = h->tk_tagV; tg.tag = (int)(h->seq_tag + 1u);
    tg.mirror = nullptr; tg.mtags = nullptr; tg.mseq = 0;
    int n_pass = 0, live = 0;
    DISPATCH_T(h, n_pass = FusedTile<T>::blocks(h->ld, h->L));
    DISPATCH_T(h, live = FusedTile<T>::tiles_r(h->ld) * FusedTile<T>::tiles_c(h->L));
    RunArg ra;
    if (!h->tk_run) {
        // [0, 1024) the strips' / pass workgroups' words, [1024, 2048) the pass workgroups' second words
        const size_t bytes = sizeof(int) * 2048;
        HIPCHK(hipMalloc(&h->tk_run, bytes));
        HIPCHK(hipMemsetAsync(h->tk_run, 0, bytes, h->stream));
    }
    if (strip_wgs + n_pass > 1024) return kRunNotApplicable;
    (void)live;
    ra.ticks = nt; ra.m = m; ra.done = h->tk_run; ra.done_all = h->tk_run + 1024; ra.done_base = (int)h->run_done; ra.n_strip = strip_wgs; ra.n_pass = n_pass;
    ra.ids = o.ids; ra.ids_stride = o.stride; ra.ids_off = o.off;
    int rc = NUSLAM_OK;
    DISPATCH_T(h, rc = (launch(h, NUSLAM_K_TICK_CHAIN, k_run_fused<T>, dim3(1 + n_pred + strip_wgs + n_pass, 1), dim3(256), v, o, total,
                               (T*)h->P(), (T*)h->Palt(), h->tk_plan, h->tk_K, h->tk_R, h->tk_V, pub, tg, n_pred, strip_wgs, o.J, h->tk_sync + 2, ra)));
    if (rc) return rc;
    h->run_done += (unsigned)nt;
    h->seq_tag += (unsigned)nt;
    h->seq_pub += 2u * kTickJ * (unsigned)nt;
    h->seq_gather += (unsigned)nt; h->seq_pred += (unsigned)n_pred * (unsigned)nt;
    if (nt & 1) { h->sidx ^= 1; h->cidx ^= 1; h->pidx ^= 1; }
    h->state_epoch += (unsigned)nt;
    h->host_seen_valid = false;
    return NUSLAM_OK;
}

ObsArg inline_obs(double a, double b, int id, int cartesian)
{
    ObsArg o;
    o.a = nullptr; o.b = nullptr; o.ids = nullptr; o.stride = 0; o.off = 0;
    o.a0 = a; o.b0 = b; o.id0 = id; o.cartesian = cartesian; o.log_slot = -1;
    return o;
}

// One loop body of slam.cpp:250-319 for every filter of the batch.
int do_tick(nuslam_batch* h, const TwistArg& tw, ObsArg base, int m, bool known, int total,
            const int* host_ids = nullptr, const double* host_mx = nullptr, const double* host_my = nullptr,
            const int* pf_ids = nullptr, long long pf_stride = 0, bool no_predict = false)
{
    // (no_predict: the markers continue a tick whose predict has already run -- nuslam_ekf_tick_ex without a twist)
    // One filter, known ids, the tick pipeline: no predict kernel -- the predict rides in the first round's launch (k_tick_front)
    const bool fuse_predict = !no_predict && known && m > 0 && tick_pipeline_pays(h, m) && !h->deferred && !h->dense_predict && h->predict_bookkeeping &&
                              front_fits(h, true);
    int rc = NUSLAM_OK;
    if (fuse_predict) {
        if (h->poisoned) return NUSLAM_E_SYNC;
        h->last_tick = -1;
        rc = do_tick_rounds(h, base, m, total, host_ids, host_mx, host_my, pf_ids, pf_stride, &tw);
        if (rc) return rc;
        h->host_seen_valid = false;
        return NUSLAM_OK;
    }
    if (no_predict) { if (h->poisoned) return NUSLAM_E_SYNC; h->last_tick = -1; }
    else rc = do_predict(h, tw);
    if (rc) return rc;
    // Pairing needs every marker of the tick to be a plain correction of an already-initialised landmark, in every
    // filter: then the caller's chain (slam.cpp:295-316) takes the `update` branch for each of them and `seen` does
    // not move.  host_ids: one id list for all filters; pf_ids: filter b's list at pf_ids + b * pf_stride.
    if (known && tick_pipeline_pays(h, m) && !h->deferred && m > 0) {
        rc = do_tick_rounds(h, base, m, total, host_ids, host_mx, host_my, pf_ids, pf_stride);
        if (rc) return rc;
        h->host_seen_valid = false;                 // (the mirror only serves the per-correction path's pairing)
        return NUSLAM_OK;
    }
    if (!known && tick_pipeline_pays(h, m) && !h->deferred && m > 0 && h->n >= 1) {
        rc = do_da_rounds(h, base, m, total, host_mx, host_my);
        if (rc) return rc;
        h->host_seen_valid = false;
        return NUSLAM_OK;
    }
    auto id_of = [&](int b, int i) { return host_ids ? host_ids[i] : pf_ids[(size_t)b * pf_stride + i]; };
    const bool have_ids = host_ids != nullptr || pf_ids != nullptr;
    bool plain = known && have_ids && h->host_seen_valid;
    if (plain)
        for (int b = 0; b < h->B && plain; ++b)
            for (int i = 0; i < m; ++i) {
                const int id = id_of(b, i);
                if (id < 1 || id > h->n || id > total || id > h->host_seen[b]) { plain = false; break; }
            }
    const bool pair = plain && h->pairing && !h->deferred;
    auto obs_at = [&](int i) {
        ObsArg o = base;
        o.off = base.off + i;
        o.log_slot = h->id_log ? i : -1;
        if (host_ids) { o.ids = nullptr; o.id0 = host_ids[i]; }        // same id for every filter: pass it inline
        if (host_mx) { o.a = nullptr; o.b = nullptr; o.a0 = host_mx[i]; o.b0 = host_my[i]; }
        return o;
    };
    for (int i = 0; i < m; ++i) {
        if (pair && i + 1 < m) {
            rc = do_update2(h, obs_at(i), obs_at(i + 1));
            if (rc) return rc;
            ++i;
            continue;
        }
        ObsArg o = obs_at(i);
        if (!known) {
            rc = do_associate(h, o);
            if (rc) return rc;
        }
        rc = do_update(h, o, known ? MODE_KNOWN : MODE_DA, total);
        if (rc) return rc;
    }
    // keep the host mirror of `seen` exact, or drop it
    if (!known || !have_ids) h->host_seen_valid = false;
    else if (h->host_seen_valid && !plain) {                            // plain ticks leave `seen` where it was
        for (int b = 0; b < h->B && h->host_seen_valid; ++b)
            for (int i = 0; i < m; ++i) {
                const int id = id_of(b, i);
                if (id < 0) continue;                                   // skipped marker
                if (id < 1 || id > h->n || id > total) { h->host_seen_valid = false; break; }   // error / break paths: stop mirroring
                if (id > h->host_seen[b]) h->host_seen[b] = id;
            }
    }
    return NUSLAM_OK;
}

int ensure_state_mirror(nuslam_batch* h)
{
    if (h->st_host) return NUSLAM_OK;
    HIPCHK(hipHostMalloc((void**)&h->st_host, sizeof(double) * (size_t)h->ld, hipHostMallocMapped));
    HIPCHK(hipHostMalloc((void**)&h->st_tags, sizeof(long long) * 512, hipHostMallocMapped));
    for (int i = 0; i < 512; ++i) h->st_tags[i] = 0;
    return NUSLAM_OK;
}

// ---- the served round (unknown association driven call by call): host side of csrc/ekf_da.h "a round SERVED to the host"
int get_seen(nuslam_batch* h, int b, int* seen);
bool serve_usable(const nuslam_batch* h)
{
    return h->lazy.on && h->B == 1 && h->n >= 1 && h->tick_mode != 0 && h->tick_mode != 2 && !h->deferred && !h->dense_predict &&
           (h->ld - 3 + kDaOwn - 1) / kDaOwn <= h->n_cu;
}
int serve_open(nuslam_batch* h)
{
    nuslam_batch::Serve& sv = h->srv;
    { int erc = ensure_tick_buffers(h); if (erc) return erc; }
    { int erc = ensure_da_buffers(h); if (erc) return erc; }
    if (!sv.mail) {
        const int words = 16 + roundup(h->da.nwg, 8);
        HIPCHK(hipHostMalloc((void**)&sv.mail, sizeof(long long) * words, hipHostMallocMapped));
        for (int i = 0; i < words; ++i) sv.mail[i] = 0;
    }
    if (!h->host_seen_valid) {                              // the host decodes the verdicts itself: it needs `seen` exactly
        int seen = 0;
        int grc = get_seen(h, 0, &seen);
        if (grc) return grc;
    }
    if (h->poisoned) return NUSLAM_E_SYNC;
    h->last_tick = -1;
    TickObs o = make_tick_obs(h, inline_obs(0.0, 0.0, 0, 0), 0, kTickJ, nullptr, nullptr, nullptr);
    o.log_slot0 = -1;
    DaServe ds;
    ds.cmd = sv.mail; ds.ans = sv.mail + 8; ds.keys = sv.mail + 16; ds.seq0 = sv.seq + 1; ds.timeout_ticks = sv.timeout_us * 100;
    ds.mirror = nullptr; ds.mtags = nullptr; ds.mseq = 0;
    if (h->da.nwg <= 512 && ensure_state_mirror(h) == NUSLAM_OK) {
        h->st_seq += 1;
        ds.mirror = h->st_host; ds.mtags = h->st_tags; ds.mseq = h->st_seq;
        sv.mirror_seq = h->st_seq;
    }
    const View v = h->view();
    double* vbuf = h->pass_mode == 0 ? h->tk_V : nullptr;
    int rc = NUSLAM_OK;
    DISPATCH_T(h, rc = (launch_lds(h, NUSLAM_K_DA_STEP, k_da_round<T, true>, dim3(h->da.nwg, 1), dim3(256), (size_t)kDaRoundLds, v, o, h->n,
                                   (const T*)h->P(), h->da, h->tk_plan, h->tk_K, h->tk_R, vbuf, (int)h->da_round_tag, ds)));
    if (rc) return rc;
    h->da_round_tag += kTickJ + 1;
    sv.open = true; sv.trips = 0; sv.any_init = false; sv.z_valid = false;
    return NUSLAM_OK;
}
void serve_send(nuslam_batch* h, int flags, int id, double r, double phi, double cr, double cphi)
{
    nuslam_batch::Serve& sv = h->srv;
    if (sv.clear_status) { flags |= DA_F_CLEAR; sv.clear_status = false; }
    volatile long long* m = sv.mail;
    long long w1, w2, w4, w5;
    memcpy(&w1, &r, 8); memcpy(&w2, &phi, 8); memcpy(&w4, &cr, 8); memcpy(&w5, &cphi, 8);
    sv.seq += 1;
    const long long w0 = ((long long)sv.seq << 32) | ((long long)(flags & 0xff) << 24) | (long long)(id & 0xffffff);
    m[1] = w1; m[2] = w2; m[4] = w4; m[5] = w5;
    m[3] = w0 ^ w1 ^ w2 ^ w4 ^ w5 ^ kDaCmdMagic;
    __atomic_store_n((long long*)&m[0], w0, __ATOMIC_RELEASE);
}
// the device's answer to command sv.seq: 0 = carried out (id / seen / status filled in), 1 = the round had parked: the command was
// NOT taken; < 0: nothing came back (NUSLAM_E_SYNC)
int serve_wait(nuslam_batch* h, bool scan, int* id, int* seen, int* status)
{
    nuslam_batch::Serve& sv = h->srv;
    volatile long long* a = sv.mail + 8;
    const auto t0 = std::chrono::steady_clock::now();
    if (scan) {
        // the verdict comes as every workgroup's key (min over its candidates of 4 k + outcome): the minimum, decoded as
        // decode_association does on the device (slam_library.cpp:197-200, 206-207, 238-252) -- the device does the same for itself
        volatile long long* keys = sv.mail + 16;
        const int nwg = h->da.nwg;
        int got = 0, kmin = kNoKey;
        for (unsigned spin = 1;; ++spin) {
            while (got < nwg) {
                const long long w = keys[got];
                if ((int)(w >> 32) != sv.seq) break;
                const int k = (int)(unsigned)(w & 0xffffffffll);
                kmin = k < kmin ? k : kmin;
                ++got;
            }
            if (got == nwg) break;
            const long long a1 = a[1];
            if ((int)(a1 >> 32) == sv.seq && (((unsigned)(a1 & 0xffffffffll) >> 28) & 1u)) return 1;      // parked
            __builtin_ia32_pause();
            if ((spin & 0x3fff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) return -1;
        }
        const int n = h->n, s0 = h->host_seen[0];
        int rid = -1, ns = s0, st = 0;
        if (s0 == 0) { ns = 1; rid = 1; }
        else if (s0 >= n) { rid = -1; st = NUSLAM_E_BOUNDS; }
        else if (kmin == kNoKey) { ns = s0 + 1; rid = ns; }
        else if ((kmin & 3) == 0) rid = kmin >> 2;
        else if ((kmin & 3) == 1) rid = -1;
        else { rid = -1; st = NUSLAM_E_SINGULAR; }
        if (id) *id = rid;
        if (seen) *seen = ns;
        if (status) *status = st;
        return 0;
    }
    for (unsigned spin = 1;; ++spin) {
        const long long a0 = a[0], a1 = a[1];
        if ((int)(a0 >> 32) == sv.seq && (int)(a1 >> 32) == sv.seq) {
            const unsigned lo = (unsigned)(a1 & 0xffffffffll);
            if (id) *id = (int)(a0 & 0xffffffffll);
            if (seen) *seen = (int)(lo & 0xffffffu);
            if (status) *status = (int)((lo >> 24) & 15u);
            return (int)(lo >> 28) & 1;
        }
        __builtin_ia32_pause();
        if ((spin & 0x3fff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) return -1;
    }
}
// the round is over on the device (END carried out, or parked): the ONE pass over P for its `slots` corrections
int serve_close(nuslam_batch* h, int slots)
{
    nuslam_batch::Serve& sv = h->srv;
    sv.open = false; sv.z_valid = false;
    if (slots <= 0) return NUSLAM_OK;                       // (the round ended before its first correction slot: nothing was written)
    const View v = h->view();
    int rc = launch_pass(h, v, slots, h->tk_plan, false, sv.any_init, whole(h));
    if (rc) return rc;
    h->sidx ^= 1; h->state_epoch++;
    h->cidx ^= 1;
    h->pidx ^= 1;
    if (sv.mirror_seq == h->st_seq && sv.mirror_seq != 0) { h->st_ntags = h->da.nwg; h->mirror_epoch = h->state_epoch; }
    if (sv.any_init) for (auto& t : h->touched) t = 0;     // (which landmarks: not tracked for served rounds -- drop the proof)
    return NUSLAM_OK;
}
int serve_poison(nuslam_batch* h)
{
    if (!h->poisoned) h->needs_restore.assign((size_t)h->B, 1);
    h->poisoned = true;
    h->srv.open = false;
    return NUSLAM_E_SYNC;
}
// One command and its answer.  parked_out: the device had closed the round by itself before the command arrived (nothing of it was
// applied; the round has been closed here as well).
int serve_roundtrip(nuslam_batch* h, int flags, int id, double r, double phi, double cr, double cphi, int* id_out, int* status_out,
                    bool* parked_out)
{
    nuslam_batch::Serve& sv = h->srv;
    serve_send(h, flags, id, r, phi, cr, cphi);
    int seen = 0, st = 0, rid = 0;
    const int k = serve_wait(h, (flags & DA_F_SCAN) != 0, &rid, &seen, &st);
    if (k < 0) return serve_poison(h);
    *parked_out = k == 1;
    if (k == 1) return serve_close(h, sv.trips);           // trips executed incl. the parking one = trips + 1; slots = that - 1
    sv.trips += 1;
    if (flags & DA_F_SCAN) { h->host_seen[0] = seen; h->host_seen_valid = true; }
    if (id_out) *id_out = rid;
    if (status_out) *status_out = st;
    if (flags & DA_F_END) return serve_close(h, sv.trips - 1);
    return NUSLAM_OK;
}
// flags + payload of the correction the caller has decided on (none: 0)
int serve_pending_flags(nuslam_batch* h, double* cr, double* cphi)
{
    nuslam_batch::Serve& sv = h->srv;
    *cr = *cphi = 0.0;
    if (!sv.pend) return 0;
    int f = DA_F_CORR | (sv.p_init ? DA_F_INIT : 0);
    if (!(sv.z_valid && sv.z_r == sv.p_r && sv.z_phi == sv.p_phi)) { f |= DA_F_ZOVR; *cr = sv.p_r; *cphi = sv.p_phi; }
    if (sv.p_init) sv.any_init = true;
    return f;
}
int lazy_flush(nuslam_batch* h);
// the served round's end: the correction still pending goes with the END command
int serve_end(nuslam_batch* h)
{
    nuslam_batch::Serve& sv = h->srv;
    if (!sv.open) return NUSLAM_OK;
    double cr, cphi;
    const int f = serve_pending_flags(h, &cr, &cphi);
    bool parked = false;
    int st = 0;
    int rc = serve_roundtrip(h, f | DA_F_END, sv.pend ? sv.p_id : 0, 0.0, 0.0, cr, cphi, nullptr, &st, &parked);
    if (rc) return rc;
    if (parked && sv.pend) {
        // the device had closed the round before this arrived: the correction goes through the per-call kernels
        const ObsArg o = inline_obs(sv.p_r, sv.p_phi, sv.p_id, 0);
        if (sv.p_init) rc = launch(h, -1, k_init_landmark, dim3(1), dim3(1), (h->state_epoch++, h->view()), o, h->state[h->sidx]);
        if (!rc) rc = do_update(h, o, MODE_FORCE, h->n);
    }
    sv.pend = false;
    return rc;
}
// associateLandmark(z) through the served round
int serve_associate(nuslam_batch* h, double r, double phi, int* id_out)
{
    nuslam_batch::Serve& sv = h->srv;
    for (int attempt = 0; attempt < 3; ++attempt) {
        if (sv.open && sv.trips >= kTickJ) { int rc = serve_end(h); if (rc) return rc; }    // every slot of the round is used
        if (!sv.open) {
            int rc = lazy_flush(h);                         // what was recorded without association (the tick's predict) goes first
            if (!rc) rc = serve_open(h);
            if (rc) return rc;
        }
        double cr, cphi;
        const int f = serve_pending_flags(h, &cr, &cphi);
        bool parked = false;
        int st = 0, id = 0;
        int rc = serve_roundtrip(h, f | DA_F_SCAN, sv.pend ? sv.p_id : 0, r, phi, cr, cphi, &id, &st, &parked);
        if (rc) return rc;
        if (parked) {
            // the round had closed itself (no call for srv.timeout_us): the pending correction through the per-call kernels, then a new round
            if (sv.pend) {
                const ObsArg o = inline_obs(sv.p_r, sv.p_phi, sv.p_id, 0);
                if (sv.p_init) rc = launch(h, -1, k_init_landmark, dim3(1), dim3(1), (h->state_epoch++, h->view()), o, h->state[h->sidx]);
                if (!rc) rc = do_update(h, o, MODE_FORCE, h->n);
                sv.pend = false;
                if (rc) return rc;
            }
            continue;
        }
        sv.pend = false;
        sv.z_valid = true; sv.z_r = r; sv.z_phi = phi;
        sv.used = true;
        *id_out = id;
        if (st != 0) { sv.clear_status = true; return st; }     // a full map / singular psi: reported from inside this very call
        return NUSLAM_OK;
    }
    return serve_poison(h);                                 // (three rounds in a row closed under the caller's feet)
}
// update() while a served round is open: the decision travels with the caller's NEXT call (the next associateLandmark, or the end of
// the round); a second update() in a row sends the first one on its own
int serve_update(nuslam_batch* h, double r, double phi, int id, bool init)
{
    nuslam_batch::Serve& sv = h->srv;
    if (sv.trips == 0) {
        // The round was opened ahead of its first associateLandmark (predict() of a caller whose last tick associated) and has not
        // scanned a marker yet: its first command cannot carry a correction.  End it (nothing was written) and let the caller record
        // this update the lazy way -- the tick's predict has already run.
        sv.pend = false;
        int rc = serve_end(h);
        if (rc) return rc;
        return -1;
    }
    if (sv.pend) {
        if (sv.trips >= kTickJ) { int rc = serve_end(h); if (rc) return rc; }
        else {
            double cr, cphi;
            const int f = serve_pending_flags(h, &cr, &cphi);
            bool parked = false;
            int st = 0;
            int rc = serve_roundtrip(h, f, sv.p_id, 0.0, 0.0, cr, cphi, nullptr, &st, &parked);
            if (!rc && parked) {
                const ObsArg o = inline_obs(sv.p_r, sv.p_phi, sv.p_id, 0);
                if (sv.p_init) rc = launch(h, -1, k_init_landmark, dim3(1), dim3(1), (h->state_epoch++, h->view()), o, h->state[h->sidx]);
                if (!rc) rc = do_update(h, o, MODE_FORCE, h->n);
            }
            sv.pend = false;
            sv.z_valid = false;                             // (the scanned marker has had its correction: a further one brings its own z)
            if (rc) return rc;
        }
    }
    if (!sv.open) return -1;                                // (the round is gone: the caller records the update the lazy way)
    sv.pend = true; sv.p_id = id; sv.p_init = init; sv.p_r = r; sv.p_phi = phi;
    return NUSLAM_OK;
}

// ---- the class API driven call by call (nuslam_ekf_predict / _init_landmark / _update), recorded and applied as ONE tick
// What the caller's loop (slam.cpp:269-318) has asked for since the last flush goes to the device: the recorded predict rides in
// the first round's k_tick_front launch, the recorded update() calls are its corrections in MODE_FORCE (the caller has taken the
// decisions of slam.cpp:295-316 itself; an initializeLandmark in front of an update is that correction's `init` flag), then the
// rank-2m pass -- the kernels, and the bits, of nuslam_ekf_tick_ex on the same inputs.  Fewer than four corrections (where the
// tick pipeline does not pay, tick_pipeline_pays), the deferred and the dense-predict modes take the per-call kernels.
// A failure drops what was recorded and is returned to the call that triggered the flush.
int lazy_flush(nuslam_batch* h)
{
    nuslam_batch::Lazy& z = h->lazy;
    if (z.busy) return NUSLAM_OK;
    if (h->srv.open) {                                  // the served round ends here: its last correction and the pass over P
        z.busy = true;
        const int src = serve_end(h);
        z.busy = false;
        if (src) return src;
    }
    if (z.empty()) return NUSLAM_OK;
    z.busy = true;
    struct Done {
        nuslam_batch::Lazy& z;
        ~Done() { z.has_predict = false; z.pend_init = false; z.r.clear(); z.phi.clear(); z.id.clear(); z.init.clear(); z.busy = false; }
    } done{ z };
    HIPCHK(hipSetDevice(h->device));
    const int m = (int)z.id.size();
    TwistArg tw;
    tw.tw = nullptr; tw.stride = 0; tw.off = 0; tw.dth0 = z.dth; tw.dx0 = z.dx;
    int rc = NUSLAM_OK;
    if (m > 0 && tick_pipeline_pays(h, m) && !h->deferred && !h->dense_predict && h->B == 1) {
        if (h->poisoned) return NUSLAM_E_SYNC;
        h->last_tick = -1;
        int* saved_log = h->id_log;
        h->id_log = nullptr;
        rc = do_tick_rounds(h, inline_obs(0.0, 0.0, 0, 0), m, h->n, z.id.data(), z.r.data(), z.phi.data(), nullptr, 0,
                            z.has_predict ? &tw : nullptr, z.init.data());
        h->id_log = saved_log;
        // (forced corrections do not move `seen`: the host's mirror of it stays what it was)
    } else {
        if (z.has_predict) rc = do_predict(h, tw);
        for (int i = 0; i < m && !rc; ++i) {
            const ObsArg o = inline_obs(z.r[i], z.phi[i], z.id[i], 0);
            if (z.init[i]) rc = launch(h, -1, k_init_landmark, dim3(1), dim3(1), (h->state_epoch++, h->view()), o, h->state[h->sidx]);
            if (!rc) rc = do_update(h, o, MODE_FORCE, h->n);
            if (!rc) h->touched[z.id[i]] = 1;         // (B == 1: the landmark has been corrected)
        }
    }
    if (!rc && z.pend_init)                             // an initializeLandmark no update() followed
        rc = launch(h, -1, k_init_landmark, dim3(1), dim3(1), (h->state_epoch++, h->view()), inline_obs(z.pi_r, z.pi_phi, z.pi_id, 0), h->state[h->sidx]);
    return rc;
}

// nuslam_batch_run with the chains running ahead: the ticks of a resident known-id trace, where the host knows the next
// tick's markers while it enqueues this one.  Two free-running streams, no event between them inside the loop (a
// cross-stream hipEvent cost 10-20 us per use here); the two hand-offs per tick are counters in device memory
// (tick_signal / tick_wait):
//   handle's stream   predict(t), prep(t) -> [plan(t)] strips(t) -> signal -> pass(t)         (P, state vector)
//   chain stream      [strips(t-1)] chain(t) -> [strips(t)] chain(t+1) -> ...                 (a 35 x 35 block)
// prep(t) gathers the 35 x 35 block at the NEXT tick's index set out of the covariance pass(t) will read -- into the
// block buffer of tick t+1's parity: chain(t+1) reads it at its start, while prep(t+1) (which only its LAST workgroup holds
// back until chain(t+1) is done) already writes the other one; strips(t)
// also drops the gain / prior-row strips at that set into compact arrays; chain(t+1) -- one kernel, waiting inside for
// the signal behind strips(t) -- first replays the round on the block and applies predict(t+1) (tick_carry), then runs
// its corrections: all while pass(t), predict(t+1), prep(t+1) run.  Same arithmetic, same bits as the one-stream order
// (tests/test_gpu_tick.py::test_overlapped_run_is_bit_identical).
int run_overlapped(nuslam_batch* h, int t_begin, int t_end, int total)
{
    { int erc = ensure_tick_buffers(h); if (erc) return erc; }
    const size_t B = (size_t)h->B;
    if (!h->stream2) {
        // its own hardware queue, served first: a chain is one workgroup per filter
        int prio_lo = 0, prio_hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        if (h->ov_same_stream) h->stream2 = h->stream;
        else HIPCHK(hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, prio_hi));
        HIPCHK(hipMalloc(&h->tk_plan2, sizeof(TickStep) * B * kTickJ));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tick_chain<double, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kTickCarryLds));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tick_chain<float, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kTickCarryLds));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tick_chain_pub<double, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kTickCarryLds));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tick_chain_pub<float, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kTickCarryLds));
        HIPCHK(hipMalloc(&h->tk_ctrl4, sizeof(int) * 4 * B));
        HIPCHK(hipMalloc(&h->tk_blk, sizeof(double) * 2 * B * kTickNU * kTickNU));      // two: by the parity of the tick that reads it
        HIPCHK(hipMalloc(&h->tk_posmap, sizeof(int) * B * h->ld));
        HIPCHK(hipMalloc(&h->tk_KU, sizeof(double) * (B * kTickJ * 2 * kTickNU + kTickDump)));
        HIPCHK(hipMalloc(&h->tk_RU, sizeof(double) * (B * kTickJ * 5 * kTickNU + kTickDump)));
        HIPCHK(hipMalloc(&h->tk_SU, sizeof(double) * B * kTickNU));
        HIPCHK(hipEventCreateWithFlags(&h->ov_start, hipEventDisableTiming));
        h->seq_chain = h->seq_next = 0;
    }
    if (h->ov_ok < 0) {
        // Do the two streams really run side by side here?  (Under rocprofv3 --pmc, a serialising debug environment or a CU
        // mask they do not, and every hand-off of the run would expire.)  One probe per handle: a one-wave consumer on the
        // chain stream waits -- briefly -- for a counter that a one-wave producer, enqueued AFTER it on the handle's stream,
        // bumps.  If it gives up, overlapped runs of this handle take the one-stream path: same bits, no dependency.
        HIPCHK(hipStreamSynchronize(h->stream));
        int rcp = launch_on(h, h->stream2, -1, k_tick_wait, dim3(1), dim3(64), 0, (const int*)(h->tk_sync + 3), 1, h->tk_sync + 4, 1 << 13);
        if (!rcp) rcp = launch(h, -1, k_tick_signal, dim3(1), dim3(64), h->tk_sync + 3);
        if (rcp) return rcp;
        HIPCHK(hipStreamSynchronize(h->stream2));
        HIPCHK(hipStreamSynchronize(h->stream));
        int gave_up = 0;
        HIPCHK(hipMemcpy(&gave_up, h->tk_sync + 4, sizeof(int), hipMemcpyDeviceToHost));
        h->ov_ok = gave_up ? 0 : 1;
    }
    if (!h->ov_ok) return kNotConcurrent;             // (internal: the caller falls back to the one-stream loop)
    HIPCHK(hipMemsetAsync(h->tk_posmap, 0xff, sizeof(int) * B * h->ld, h->stream));      // every index: not in the next tick's set
    auto obs_of = [&](int t) {
        ObsArg o;
        o.a = h->tr_mx; o.b = h->tr_my; o.ids = h->tr_ids;
        o.stride = h->tr_bcast ? 0 : (long long)h->tr_ticks * h->tr_m;
        o.off = (long long)t * h->tr_m;
        o.a0 = o.b0 = 0.0; o.id0 = 0; o.cartesian = 1; o.log_slot = -1;
        const int* hid = h->h_ids.empty() ? nullptr : h->h_ids.data() + (size_t)t * h->tr_m;
        return make_tick_obs(h, o, 0, h->tr_m, hid, nullptr, nullptr);
    };
    auto twist_of = [&](int t) {
        TwistArg tw;
        tw.tw = h->tr_tw; tw.stride = h->tr_bcast ? 0 : (long long)h->tr_ticks * 2; tw.off = (long long)t * 2;
        tw.dth0 = tw.dx0 = 0.0;
        return tw;
    };
    int* cnt_chain = h->tk_sync + 0;
    int* cnt_next = h->tk_sync + 1;
    int* timeouts = h->tk_sync + 2;
    int rc = NUSLAM_OK;
    int prev_J = 0;
    // STREAMED (few filters: the chains' and the strips' workgroups fit the chip together): the strips of tick t are a launch on the
    // handle's stream that FOLLOWS chain(t), running on the chain stream, plan entry by plan entry (TickPublish, as inside
    // k_tick_front) instead of starting when it has ended, and counts itself done for chain(t+1), which is already waiting
    // on its CU.  The loop that bounds a tick is then chain -> (last entry's strips) -> chain, ~38 us at N = 1000, with the pass
    // over P, the next predict and prep entirely beside it.
    const int n_strip = (h->ld + 31) / 32;
    const bool streamed_fits = h->front && (long long)(n_strip + 1) * h->B <= h->n_cu;
    for (int t = t_begin; t < t_end; ++t) {
        TickStep* plan = ((t - t_begin) & 1) ? h->tk_plan2 : h->tk_plan;
        const bool more = t + 1 < t_end;
        // ---- handle's stream: predict(t) (the control words travel with the chains after the first tick), prep(t)
        h->predict_bookkeeping = (t == t_begin) ? 1 : 0;
        rc = do_predict(h, twist_of(t));
        h->predict_bookkeeping = 1;
        if (rc) return rc;
        const TickObs o = obs_of(t);
        const View v = h->view();
        const bool may_init = tick_may_init_all(h, h->h_ids.empty() ? nullptr : h->h_ids.data() + (size_t)t * h->tr_m,
                                                h->h_ids_pf.empty() ? nullptr : h->h_ids_pf.data() + (size_t)t * h->tr_m,
                                                (long long)h->tr_ticks * h->tr_m, h->tr_m, total);

        TickPublish pub;
        pub.flag = h->tk_pub; pub.base = (int)h->seq_pub; pub.predict = 0; pub.gbase = pub.pbase = 0;
        pub.rank_panels = h->pass_mode == 0 ? 1 : 0;
        // (rank-form strips that follow the chain need the host's proof that the round holds no first sighting: do_tick_rounds)
        const bool streamed = streamed_fits && (h->pass_mode != 0 || !may_init);
        pub.tw.tw = nullptr; pub.tw.stride = pub.tw.off = 0; pub.tw.dth0 = pub.tw.dx0 = 0.0;
        // ---- chain stream: chain(t): from P for the first tick (behind predict), from the strips of tick t-1 afterwards
        if (t == t_begin) {
            HIPCHK(hipEventRecord(h->ov_start, h->stream));               // once per run
            HIPCHK(hipStreamWaitEvent(h->stream2, h->ov_start, 0));
            if (streamed)
                DISPATCH_T(h, rc = (launch_on(h, h->stream2, NUSLAM_K_TICK_CHAIN, k_tick_chain_pub<T, false>, dim3(h->B), dim3(256), 0, v, o,
                                              total, (const T*)h->P(), plan, TickCarry{}, h->tk_ctrl4, cnt_chain, pub)));
            else
                DISPATCH_T(h, rc = (launch_on(h, h->stream2, NUSLAM_K_TICK_CHAIN, k_tick_chain<T, false>, dim3(h->B), dim3(256), 0, v, o,
                                              total, (const T*)h->P(), plan, TickCarry{}, h->tk_ctrl4, cnt_chain)));
        } else {
            // (it waits, inside, for the strips of tick t-1 at this tick's index set: cnt_next counts the signal kernels
            // enqueued behind k_tick_panels)
            TickCarry cy;
            cy.blk = h->tk_blk + (size_t)(t & 1) * B * kTickNU * kTickNU; cy.KU = h->tk_KU; cy.RU = h->tk_RU; cy.SU = h->tk_SU;
            cy.plan_prev = ((t - t_begin) & 1) ? h->tk_plan : h->tk_plan2;
            cy.wait_cnt = cnt_next; cy.wait_target = h->seq_next; cy.timeouts = timeouts;
            cy.Jt = prev_J; cy.tw = twist_of(t);
            cy.rank = h->pass_mode == 0 ? 1 : 0;
            // Many filters: their chains' workgroups must not sit on the CUs spinning while the strips they wait for still
            // need room to run -- a one-wave kernel does the waiting for them (their own wait then falls through).
            if ((long long)h->B * 4 > h->n_cu) {
                rc = launch_on(h, h->stream2, -1, k_tick_wait, dim3(1), dim3(64), 0, (const int*)cnt_next, h->seq_next, timeouts, 1 << 18);
                if (rc) return rc;
            }
            if (streamed)
                DISPATCH_T(h, rc = (launch_on(h, h->stream2, NUSLAM_K_TICK_CHAIN, k_tick_chain_pub<T, true>, dim3(h->B), dim3(256),
                                              (size_t)kTickCarryLds, v, o, total, (const T*)h->P(), plan, cy, h->tk_ctrl4, cnt_chain, pub)));
            else
                DISPATCH_T(h, rc = (launch_on(h, h->stream2, NUSLAM_K_TICK_CHAIN, k_tick_chain<T, true>, dim3(h->B), dim3(256),
                                              (size_t)kTickCarryLds, v, o, total, (const T*)h->P(), plan, cy, h->tk_ctrl4, cnt_chain)));
        }
        prev_J = o.J;
        if (rc) return rc;
        h->seq_chain += h->B;
        if (streamed) {
            // ---- handle's stream: prep(t), the strips following chain(t) (every workgroup counts itself done for chain(t+1)), the pass
            h->seq_pub += 2u * kTickJ;
            if (more) {
                DISPATCH_T(h, rc = (launch(h, -1, k_tick_prep<T>, dim3(kTickNU + 1, h->B), dim3(64), v, o, obs_of(t + 1),
                                           (const T*)h->P(), h->tk_posmap, h->tk_blk + (size_t)((t + 1) & 1) * B * kTickNU * kTickNU,
                                           (const int*)cnt_chain, h->seq_chain, timeouts)));
                if (rc) return rc;
            }
            double* vbuf = h->pass_mode == 0 ? h->tk_V : nullptr;
            DISPATCH_T(h, rc = (launch(h, NUSLAM_K_TICK_PANELS, k_tick_strips<T>, dim3(n_strip, h->B), dim3(256), v, o, (const T*)h->P(),
                                       (const TickStep*)plan, h->tk_K, h->tk_R, vbuf, pub, timeouts, (const int*)(more ? h->tk_posmap : nullptr),
                                       h->tk_KU, h->tk_RU, h->tk_SU, more ? cnt_next : (int*)nullptr)));
            if (rc) return rc;
            if (more) h->seq_next += n_strip * h->B;
            else {
                // (the run's last chain has written the control words too before anything else on this stream goes on)
                rc = launch(h, -1, k_tick_wait, dim3(1), dim3(64), (const int*)cnt_chain, h->seq_chain, timeouts, 1 << 18);
                if (rc) return rc;
            }
            rc = launch_pass(h, v, o.J, plan, more, may_init, whole(h));
            if (rc) return rc;
            h->sidx ^= 1; h->state_epoch++;
            h->cidx ^= 1;
            h->pidx ^= 1;
            continue;
        }
        // ---- handle's stream: prep(t) (its last workgroup waits for plan(t)), strips(t), the signal, the pass over P
        if (more)
            DISPATCH_T(h, rc = (launch(h, -1, k_tick_prep<T>, dim3(kTickNU + 2, h->B), dim3(64), v, o, obs_of(t + 1),
                                       (const T*)h->P(), h->tk_posmap, h->tk_blk + (size_t)((t + 1) & 1) * B * kTickNU * kTickNU,
                                       (const int*)cnt_chain, h->seq_chain, timeouts)));
        else
            rc = launch(h, -1, k_tick_wait, dim3(1), dim3(64), (const int*)cnt_chain, h->seq_chain, timeouts, 1 << 18);
        if (rc) return rc;
        rc = launch_strips_and_pass(h, v, o, plan, more, may_init, [&]() -> int {
            if (!more) return NUSLAM_OK;
            h->seq_next += 1;                       // the strips at the next tick's index set are complete: chain(t+1) may start
            return launch(h, -1, k_tick_signal, dim3(1), dim3(64), cnt_next);
        });
        if (rc) return rc;
        h->sidx ^= 1; h->state_epoch++;
        h->cidx ^= 1;
        h->pidx ^= 1;
    }
    h->host_seen_valid = false;
    h->last_tick = t_end - 1;
    return NUSLAM_OK;
}

void free_batch(nuslam_batch* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->srv.open) {                                     // a resident served round: tell it to end (it would park by itself otherwise)
        h->srv.pend = false;
        (void)serve_end(h);
    }
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (int g = 1; g < 4; ++g) {
        if (h->gstream[g]) { (void)hipStreamSynchronize(h->gstream[g]); (void)hipStreamDestroy(h->gstream[g]); }
        if (h->gev[g]) (void)hipEventDestroy(h->gev[g]);
    }
    if (h->gev0) (void)hipEventDestroy(h->gev0);
    for (int g = 0; g < 4; ++g)
        if (h->gpass[g]) (void)hipEventDestroy(h->gpass[g]);
    if (h->srv.mail) (void)hipHostFree(h->srv.mail);
    if (h->st_host) (void)hipHostFree(h->st_host);
    if (h->st_tags) (void)hipHostFree(h->st_tags);
    void* ptrs[] = { h->state[0], h->state[1], h->ctrl[0], h->ctrl[1], h->Pbuf[0], h->Pbuf[1], h->cur_id, h->akey, h->dU, h->dV, h->tr,
                     h->stats, h->pose_err, h->da_mem, h->tk_plan, h->tk_K, h->tk_R, h->tk_V, h->tk_pub, h->tk_tagK, h->tk_tagV, h->tk_plan2, h->tk_ctrl4, h->tk_blk, h->tk_sync, h->tk_posmap, h->tk_KU, h->tk_RU, h->tk_SU, h->tk_run, h->tr_tw, h->tr_mx, h->tr_my, h->tr_ids, h->tr_truth, h->tr_scan, h->st_mx, h->st_my, h->st_ids, h->id_log,
                     h->wF };
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (int k = 0; k < NUSLAM_K_COUNT; ++k)
        for (auto& pr : h->pending[k]) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto e : h->pool) (void)hipEventDestroy(e);
    for (auto e : h->ov_events) (void)hipEventDestroy(e);
    if (h->ov_start) (void)hipEventDestroy(h->ov_start);
    if (h->stream2 && h->stream2 != h->stream) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); }
    if (h->t0) (void)hipEventDestroy(h->t0);
    if (h->t1) (void)hipEventDestroy(h->t1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->host_word) (void)hipHostFree(h->host_word);
    delete h;
}

int alloc_batch(int B, int n, int dtype, int device, nuslam_batch** out)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return NUSLAM_E_NODEV;
    }
    if (device < 0 || device >= count) return NUSLAM_E_ARG;
    nuslam_batch* h = new (std::nothrow) nuslam_batch();
    if (!h) return NUSLAM_E_NOMEM;
    h->device = device; h->B = B; h->n = n; h->L = 3 + 2 * n; h->ld = roundup(h->L, 32); h->dtype = dtype;
    h->p_stride = (long long)h->ld * h->L;
    h->touched.assign((size_t)B * (n + 1), 0);
    int rc = [&]() -> int {
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->n_cu = cus;
        const size_t sb = sizeof(double) * (size_t)B * h->ld;
        HIPCHK(hipMalloc(&h->state[0], sb));
        HIPCHK(hipMalloc(&h->state[1], sb));
        HIPCHK(hipMalloc(&h->ctrl[0], sizeof(int) * B * C_WORDS));
        HIPCHK(hipMalloc(&h->ctrl[1], sizeof(int) * B * C_WORDS));
        HIPCHK(hipMalloc(&h->Pbuf[0], h->esize() * (size_t)B * h->p_stride));
        HIPCHK(hipMalloc(&h->Pbuf[1], h->esize() * (size_t)B * h->p_stride));
        HIPCHK(hipMalloc(&h->cur_id, sizeof(int) * B));
        HIPCHK(hipHostMalloc((void**)&h->host_word, sizeof(int) * 4, hipHostMallocMapped));
        HIPCHK(hipMalloc(&h->akey, sizeof(int) * 2 * B));
        HIPCHK(hipMalloc(&h->tr, sizeof(double) * B));
        HIPCHK(hipMalloc(&h->stats, sizeof(double) * (2 * h->L + 6)));
        HIPCHK(hipMalloc(&h->pose_err, sizeof(double) * 4 * B));
        HIPCHK(hipMemsetAsync(h->cur_id, 0, sizeof(int) * B, h->stream));
        HIPCHK(hipEventCreate(&h->t0));
        HIPCHK(hipEventCreate(&h->t1));
        return NUSLAM_OK;
    }();
    if (rc) { free_batch(h); return rc; }
    *out = h;
    return NUSLAM_OK;
}

// every stream the handle may have work on (its own, the chain stream of overlapped runs)
int sync_all_streams(nuslam_batch* h)
{
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->stream2 && h->stream2 != h->stream) HIPCHK(hipStreamSynchronize(h->stream2));
    for (int g = 1; g < 4; ++g)
        if (h->gstream[g]) HIPCHK(hipStreamSynchronize(h->gstream[g]));
    return NUSLAM_OK;
}
// The rank-2m pass reads all 32 factor rows of a filter's K / V strips every round and masks the unused ones by multiplication:
// what a diverged or abandoned run left there (a NaN) must not survive into the next one (0 * NaN = NaN across all of P).
int zero_strips(nuslam_batch* h, int b)
{
    if (!h->tk_K) return NUSLAM_OK;
    const size_t per = (size_t)kTickJ * 2 * h->ld;
    HIPCHK(hipMemsetAsync(h->tk_K + (size_t)b * per, 0, sizeof(double) * per, h->stream));
    HIPCHK(hipMemsetAsync(h->tk_V + (size_t)b * per, 0, sizeof(double) * per, h->stream));
    return NUSLAM_OK;
}

int init_batch(nuslam_batch* h, const double* robot, const double* map, const double Q[9], const double R[4])
{
    memcpy(h->Q, Q, sizeof(h->Q));
    memcpy(h->R, R, sizeof(h->R));
    struct Staged {                                   // freed on every exit path, after the stream has drained
        double* robot = nullptr; double* map = nullptr; hipStream_t stream;
        ~Staged() { (void)hipStreamSynchronize(stream); if (robot) (void)hipFree(robot); if (map) (void)hipFree(map); }
    } st;
    st.stream = h->stream;
    HIPCHK(hipSetDevice(h->device));
    if (robot) {
        HIPCHK(hipMalloc(&st.robot, sizeof(double) * 3 * h->B));
        HIPCHK(hipMemcpyAsync(st.robot, robot, sizeof(double) * 3 * h->B, hipMemcpyHostToDevice, h->stream));
    }
    if (map && h->n > 0) {
        HIPCHK(hipMalloc(&st.map, sizeof(double) * 2 * h->n * (size_t)h->B));
        HIPCHK(hipMemcpyAsync(st.map, map, sizeof(double) * 2 * h->n * (size_t)h->B, hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(hipMemsetAsync(h->Pbuf[0], 0, h->esize() * (size_t)h->B * h->p_stride, h->stream));
    HIPCHK(hipMemsetAsync(h->Pbuf[1], 0, h->esize() * (size_t)h->B * h->p_stride, h->stream));
    View v = h->view();
    dim3 grid((h->ld + 255) / 256, 1, h->B), block(256);
    int rc = NUSLAM_OK;
    DISPATCH_T(h, rc = launch(h, -1, k_init<T>, grid, block, v, (const double*)st.robot, (const double*)st.map, (T*)h->P(),
                              h->state[0], h->state[1], h->ctrl[0], h->ctrl[1]));
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    h->sidx = 0; h->cidx = 0; h->pidx = 0; h->aslot = 0;
    h->state_epoch++;
    h->host_seen.assign((size_t)h->B, 0); h->host_seen_valid = true;
    h->touched.assign((size_t)h->B * (h->n + 1), 0);
    for (int b = 0; b < h->B; ++b) { int zrc = zero_strips(h, b); if (zrc) return zrc; }
    h->poisoned = false;
    h->needs_restore.clear();
    h->last_tick = -1;
    return NUSLAM_OK;
}

int read_status(nuslam_batch* h, int clear, int* first_bad, int* status_out)
{
    HIPCHK(hipSetDevice(h->device));
    { int lrc = lazy_flush(h); if (lrc) return lrc; }
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<int> c((size_t)h->B * C_WORDS);
    HIPCHK(hipMemcpy(c.data(), h->ctrl[h->cidx], sizeof(int) * c.size(), hipMemcpyDeviceToHost));
    int st = 0, bad = -1;
    for (int b = 0; b < h->B; ++b)
        if (c[(size_t)b * C_WORDS + C_STATUS] != 0) { st = c[(size_t)b * C_WORDS + C_STATUS]; bad = b; break; }
    if (h->tk_sync) {                              // overlapped runs: a hand-off between the two streams that never arrived
        int expired = 0;
        HIPCHK(hipMemcpy(&expired, h->tk_sync + 2, sizeof(int), hipMemcpyDeviceToHost));
        if (expired) {
            if (!st) { st = NUSLAM_E_SYNC; bad = 0; }
            if (!h->poisoned) h->needs_restore.assign((size_t)h->B, 1);
            h->poisoned = true;                    // the run went on from a hand-off that never arrived: nothing after it is valid
            if (clear) HIPCHK(hipMemset(h->tk_sync + 2, 0, sizeof(int)));
        }
    }
    // a correction was skipped for a singular S: if it was a first sighting the landmark still carries INT_MAX although the
    // host marked it corrected -- drop the proof for the filter (its rounds go through the checked path until a restore)
    for (int b = 0; b < h->B; ++b)
        if (c[(size_t)b * C_WORDS + C_STATUS] == kStatusSingular)
            for (int id = 0; id <= h->n; ++id) h->touched[(size_t)b * (h->n + 1) + id] = 0;
    if (clear && st) {
        for (int b = 0; b < h->B; ++b) c[(size_t)b * C_WORDS + C_STATUS] = 0;
        HIPCHK(hipMemcpy(h->ctrl[h->cidx], c.data(), sizeof(int) * c.size(), hipMemcpyHostToDevice));
    }
    if (st == NUSLAM_E_SYNC) {                       // (the resident association round latches its expired waits in the status word)
        if (!h->poisoned) h->needs_restore.assign((size_t)h->B, 1);
        h->poisoned = true;
    }
    if (first_bad) *first_bad = bad;
    if (status_out) *status_out = st;
    return NUSLAM_OK;
}

int get_state(nuslam_batch* h, int b, double* out, int len)
{
    if (!h || !out || b < 0 || b >= h->B || len < h->L) return NUSLAM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    { int lrc = lazy_flush(h); if (lrc) return lrc; }
    if (h->st_host && h->mirror_epoch == h->state_epoch && h->B == 1) {
        // the launch that formed this state also wrote it into mapped host memory, workgroup by workgroup: wait for their stamps --
        // no stream synchronisation (the pass over P may still be running), no device-to-host copy
        volatile long long* tags = h->st_tags;
        const auto t0 = std::chrono::steady_clock::now();
        int got = 0;
        for (unsigned spin = 1; got < h->st_ntags; ++spin) {
            while (got < h->st_ntags && tags[got] == h->st_seq) ++got;
            if (got == h->st_ntags) break;
            __builtin_ia32_pause();
            if ((spin & 0x3fff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
        }
        if (got == h->st_ntags) {
            std::atomic_thread_fence(std::memory_order_acquire);
            memcpy(out, h->st_host, sizeof(double) * h->L);
            return NUSLAM_OK;
        }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, h->state[h->sidx] + (size_t)b * h->ld, sizeof(double) * h->L, hipMemcpyDeviceToHost));
    return NUSLAM_OK;
}

int get_cov(nuslam_batch* h, int b, double* out, int ld)
{
    if (!h || !out || b < 0 || b >= h->B || ld < h->L) return NUSLAM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    { int lrc = lazy_flush(h); if (lrc) return lrc; }
    { int frc = flush_pending(h); if (frc) return frc; }
    HIPCHK(hipStreamSynchronize(h->stream));
    const char* src = (const char*)h->P() + h->esize() * (size_t)b * h->p_stride;
    if (h->dtype == NUSLAM_F64) {
        HIPCHK(hipMemcpy2D(out, sizeof(double) * ld, src, sizeof(double) * h->ld, sizeof(double) * h->L, h->L,
                           hipMemcpyDeviceToHost));
    } else {
        std::vector<float> tmp((size_t)h->L * h->L);
        HIPCHK(hipMemcpy2D(tmp.data(), sizeof(float) * h->L, src, sizeof(float) * h->ld, sizeof(float) * h->L, h->L,
                           hipMemcpyDeviceToHost));
        for (int j = 0; j < h->L; ++j)
            for (int i = 0; i < h->L; ++i) out[i + (size_t)j * ld] = (double)tmp[i + (size_t)j * h->L];
    }
    return NUSLAM_OK;
}

int get_seen(nuslam_batch* h, int b, int* seen)
{
    if (!h || !seen || b < 0 || b >= h->B) return NUSLAM_E_ARG;
    // The lazy class API: predict / initializeLandmark / update do not move `seen` (slam_library.cpp: only associateLandmark does),
    // so while the host's mirror is exact the getter answers from it -- no flush, no device round trip (a 4-byte device-to-host
    // copy costs ~19 us on this box: tools/microbench/roundtrip.hip -- per tick of the caller's loop, slam.cpp:251)
    if (h->lazy.on && h->B == 1 && h->host_seen_valid) { *seen = h->host_seen[0]; return NUSLAM_OK; }
    HIPCHK(hipSetDevice(h->device));
    { int lrc = lazy_flush(h); if (lrc) return lrc; }
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(seen, h->ctrl[h->cidx] + (size_t)b * C_WORDS + C_SEEN, sizeof(int), hipMemcpyDeviceToHost));
    if (h->B == 1) { h->host_seen[0] = *seen; h->host_seen_valid = true; }   // (nothing is in flight: the mirror is exact again)
    return NUSLAM_OK;
}

int restore(nuslam_batch* h, int b, const double* state, const double* cov, int ld, int seen)
{
    if (!h || !state || !cov || b < 0 || b >= h->B || ld < h->L || seen < 0 || seen > h->n) return NUSLAM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    (void)lazy_flush(h);                                   // (whatever was recorded is about to be overwritten; its errors with it)
    { int frc = flush_pending(h); if (frc) return frc; }
    // (kernels of a failed overlapped run may still be in flight on the chain stream: nothing of theirs may land
    // behind what is written here)
    { int src = sync_all_streams(h); if (src) return src; }
    { int zrc = zero_strips(h, b); if (zrc) return zrc; }
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<double> s(h->ld, 0.0);
    memcpy(s.data(), state, sizeof(double) * h->L);
    HIPCHK(hipMemcpy(h->state[h->sidx] + (size_t)b * h->ld, s.data(), sizeof(double) * h->ld, hipMemcpyHostToDevice));
    char* dst = (char*)h->P() + h->esize() * (size_t)b * h->p_stride;
    if (h->dtype == NUSLAM_F64) {
        HIPCHK(hipMemcpy2D(dst, sizeof(double) * h->ld, cov, sizeof(double) * ld, sizeof(double) * h->L, h->L,
                           hipMemcpyHostToDevice));
    } else {
        std::vector<float> tmp((size_t)h->L * h->L);
        for (int j = 0; j < h->L; ++j)
            for (int i = 0; i < h->L; ++i) tmp[i + (size_t)j * h->L] = (float)cov[i + (size_t)j * ld];
        HIPCHK(hipMemcpy2D(dst, sizeof(float) * h->ld, tmp.data(), sizeof(float) * h->L, sizeof(float) * h->L, h->L,
                           hipMemcpyHostToDevice));
    }
    int c[C_WORDS] = { seen, seen, 0, 0 };
    HIPCHK(hipMemcpy(h->ctrl[h->cidx] + (size_t)b * C_WORDS, c, sizeof(c), hipMemcpyHostToDevice));
    h->state_epoch++;
    if (h->poisoned) {                                     // a restored filter is a valid one again -- the handle once ALL of them are
        if (h->needs_restore.size() != (size_t)h->B) h->needs_restore.assign((size_t)h->B, 1);
        h->needs_restore[b] = 0;
        bool any = false;
        for (unsigned char nr : h->needs_restore) any = any || nr != 0;
        if (!any) { h->poisoned = false; h->needs_restore.clear(); }
    }
    h->last_tick = -1;
    for (int id = 1; id <= h->n; ++id) {                   // which landmarks still carry INT_MAX: the diagonal says
        const size_t c = 3 + 2 * (size_t)(id - 1);
        // (id <= seen: above it the device's chain initialises the landmark at its next known-id marker, slam.cpp:295, and flags
        // the round as a first sighting whatever the diagonal holds -- the host's proof must say the same)
        h->touched[(size_t)b * (h->n + 1) + id] = (cov[c + c * (size_t)ld] < 1.0e9 && id <= seen) ? 1 : 0;
    }
    if (h->B == 1) { h->host_seen[0] = seen; h->host_seen_valid = true; }
    else if (h->host_seen_valid) h->host_seen[b] = seen;
    return NUSLAM_OK;
}

// the statistics vector of nuslam_batch_stats into h->stats (device), on the handle's stream
int launch_stats(nuslam_batch* h)
{
    { int lrc = lazy_flush(h); if (lrc) return lrc; }
    { int frc = flush_pending(h); if (frc) return frc; }
    View v = h->view();
    int rc = NUSLAM_OK;
    DISPATCH_T(h, rc = launch(h, -1, k_trace<T>, dim3(h->B), dim3(256), v, (const T*)h->P(), h->tr));
    if (rc) return rc;
    // truth of the last tick applied from a generated trace (tube_world's own robot); absent otherwise
    const bool have_truth = h->tr_truth != nullptr && h->last_tick >= 0 && h->last_tick < h->tr_ticks;
    DISPATCH_T(h, rc = launch(h, -1, k_pose_error<T>, dim3((h->B + 63) / 64), dim3(64), v, (const T*)h->P(),
                              (const double*)h->state[h->sidx], (const double*)(have_truth ? h->tr_truth : nullptr),
                              (long long)h->tr_ticks * 3, (long long)h->last_tick * 3, h->pose_err));
    if (rc) return rc;
    return launch(h, -1, k_stats, dim3((h->L + 255) / 256), dim3(256), v, (const double*)h->state[h->sidx],
                  (const double*)h->tr, (const double*)h->pose_err, h->stats);
}

// latched device status -> return code of a synchronising call
int sync_status(nuslam_batch* h)
{
    int st = 0;
    int rc = read_status(h, 0, nullptr, &st);
    return rc ? rc : st;
}

int ensure_stage(nuslam_batch* h, int m)
{
    if (m <= h->st_cap) return NUSLAM_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    void* olds[] = { h->st_mx, h->st_my, h->st_ids, h->id_log };
    for (void* p : olds)
        if (p) (void)hipFree(p);
    h->st_mx = h->st_my = nullptr; h->st_ids = nullptr; h->id_log = nullptr; h->st_cap = 0;
    const int cap = roundup(m, 64);
    HIPCHK(hipMalloc(&h->st_mx, sizeof(double) * cap));
    HIPCHK(hipMalloc(&h->st_my, sizeof(double) * cap));
    HIPCHK(hipMalloc(&h->st_ids, sizeof(int) * cap));
    HIPCHK(hipMalloc(&h->id_log, sizeof(int) * cap * (size_t)h->B));
    h->st_cap = cap;
    h->log_stride = cap;
    return NUSLAM_OK;
}

} // namespace

// =============================================================================================== C ABI
extern "C" {

const char* nuslam_strerror(int status)
{
    switch (status) {
    case NUSLAM_OK: return "ok";
    case NUSLAM_E_ARG: return "invalid argument";
    case NUSLAM_E_BOUNDS: return "landmark index out of bounds (Armadillo: std::logic_error)";
    case NUSLAM_E_SINGULAR: return "innovation covariance singular (Armadillo inv(): std::runtime_error)";
    case NUSLAM_E_HIP: return "HIP runtime error";
    case NUSLAM_E_NODEV: return "no HIP device";
    case NUSLAM_E_NOMEM: return "out of memory";
    case NUSLAM_E_CAPACITY: return "a fixed-size device table is too small for this input";
    case NUSLAM_E_COMM: return "RCCL error; nuslam_last_hip_error() has the text";
    case NUSLAM_E_SYNC: return "a bounded device-side wait expired (overlapped run / resident round): results invalid";
    default: return "unknown status";
    }
}

const char* nuslam_last_hip_error(void) { return g_hip_err.c_str(); }
int nuslam_abi_version(void) { return NUSLAM_HIP_ABI_VERSION; }
const char* nuslam_build_info(void) { return "csrc=" NUSLAM_CSRC_HASH; }

int nuslam_device_count(int* count)
{
    if (!count) return NUSLAM_E_ARG;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) { (void)hipGetLastError(); c = 0; }
    *count = c;
    return NUSLAM_OK;
}

// ---- host-side pure helpers (public methods of the reference class that take a caller-supplied vector)
static double h_normalize_angle(double rad) { return std::atan2(std::sin(rad), std::cos(rad)); }

int nuslam_cartesian2polar(double x, double y, double out[2])
{
    if (!out) return NUSLAM_E_ARG;
    out[0] = std::sqrt((x * x) + (y * y));
    out[1] = h_normalize_angle(std::atan2(y, x));
    return NUSLAM_OK;
}

// EKFSlam::broadcast_map2odom_tf, slam.cpp:175-210, on the host (Transform2D algebra of rigid2d.cpp:170-209)
int nuslam_map_to_odom(const double odom[3], const double state[3], double out[3])
{
    if (!odom || !state || !out) return NUSLAM_E_ARG;
    const double cob = std::cos(odom[2]), sob = std::sin(odom[2]);                 // T_ob, :179-182
    const double cmb = std::cos(state[0]), smb = std::sin(state[0]);               // T_mb, :186-188
    const double ci = cob, si = -sob;                                              // T_ob.inv(), rigid2d.cpp:187-196
    const double xi = (-odom[0] * cob) + (-odom[1] * sob);
    const double yi = (odom[0] * sob) + (-odom[1] * cob);
    const double m10 = (smb * ci) + (cmb * si);                                    // T_mb * T_bo, rigid2d.cpp:198-209
    out[0] = (cmb * xi) - (smb * yi) + state[1];
    out[1] = (smb * xi) + (cmb * yi) + state[2];
    out[2] = h_normalize_angle(std::asin(m10));                                    // :194
    return NUSLAM_OK;
}

int nuslam_measurement(const double* s, int len, int j, double out[2])
{
    if (!s || !out) return NUSLAM_E_ARG;
    if (j < 1 || 4 + 2 * (j - 1) >= len) return NUSLAM_E_BOUNDS;
    nuslam_cartesian2polar(s[3 + 2 * (j - 1)] - s[1], s[4 + 2 * (j - 1)] - s[2], out);
    out[1] = h_normalize_angle(out[1] - s[0]);
    return NUSLAM_OK;
}

int nuslam_jacobian(const double* s, int len, int j, double* H)
{
    if (!s || !H) return NUSLAM_E_ARG;
    if (j < 1 || 4 + 2 * (j - 1) >= len) return NUSLAM_E_BOUNDS;
    for (int i = 0; i < 2 * len; ++i) H[i] = 0.0;
    const int c = 3 + 2 * (j - 1);
    const double dx = s[c] - s[1], dy = s[c + 1] - s[2];
    const double d = (dx * dx) + (dy * dy);
    H[1 + 2 * 0] = -1;
    H[0 + 2 * 1] = -dx / std::sqrt(d);  H[1 + 2 * 1] = dy / d;
    H[0 + 2 * 2] = -dy / std::sqrt(d);  H[1 + 2 * 2] = -dx / d;
    H[0 + 2 * c] = dx / std::sqrt(d);   H[1 + 2 * c] = -dy / d;
    H[0 + 2 * (c + 1)] = dy / std::sqrt(d);  H[1 + 2 * (c + 1)] = dx / d;
    return NUSLAM_OK;
}

// ---- batch
int nuslam_batch_create(int n_filters, const double* robot, const double* map, int n_landmarks, const double Q[9],
                        const double R[4], int dtype, int device, nuslam_batch_t** out)
{
    if (!out || !Q || !R || n_filters < 1 || n_landmarks < 0 || (dtype != NUSLAM_F64 && dtype != NUSLAM_F32))
        return NUSLAM_E_ARG;
    nuslam_batch* h = nullptr;
    int rc = alloc_batch(n_filters, n_landmarks, dtype, device, &h);
    if (rc) return rc;
    rc = init_batch(h, robot, map, Q, R);
    if (rc) { free_batch(h); return rc; }
    *out = h;
    return NUSLAM_OK;
}

int nuslam_batch_destroy(nuslam_batch_t* h)
{
    free_batch(h);
    return NUSLAM_OK;
}

int nuslam_batch_size(const nuslam_batch_t* h, int* n_filters, int* len)
{
    if (!h) return NUSLAM_E_ARG;
    if (n_filters) *n_filters = h->B;
    if (len) *len = h->L;
    return NUSLAM_OK;
}

namespace {

void free_trace(nuslam_batch* h)
{
    void* olds[] = { h->tr_tw, h->tr_mx, h->tr_my, h->tr_ids, h->tr_truth, h->tr_scan };
    for (void* p : olds)
        if (p) (void)hipFree(p);
    h->tr_tw = h->tr_mx = h->tr_my = h->tr_truth = nullptr; h->tr_ids = nullptr; h->tr_scan = nullptr;
    h->tr_ticks = h->tr_m = h->tr_bcast = 0;
    h->last_tick = -1;
    h->tr_presence_only = false;
    h->h_ids.clear();
    h->h_ids_pf.clear();
}

__global__ void k_philox_probe(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                               unsigned* out)
{
    unsigned r[4];
    philox4x32_10(c0, c1, c2, c3, k0, k1, r);
    for (int q = 0; q < 4; ++q) out[q] = r[q];
}

__global__ void k_normalize_probe(const double* __restrict__ in, int n, double* __restrict__ out)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t < n) out[t] = normalize_angle(in[t]);
}

} // namespace

int nuslam_device_normalize_angle(const double* in, int n, double* out, int device)
{
    if (!in || !out || n < 1) return NUSLAM_E_ARG;
    HIPCHK(hipSetDevice(device));
    double* d = nullptr;
    HIPCHK(hipMalloc(&d, 2 * sizeof(double) * (size_t)n));
    hipError_t e = hipMemcpy(d, in, sizeof(double) * (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_normalize_probe, dim3((n + 63) / 64), dim3(64), 0, 0, (const double*)d, n, d + n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, d + n, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    HIPCHK(e);
    return NUSLAM_OK;
}

int nuslam_philox4x32_10(const unsigned ctr[4], const unsigned key[2], unsigned out[4], int device)
{
    if (!ctr || !key || !out) return NUSLAM_E_ARG;
    HIPCHK(hipSetDevice(device));
    unsigned* d = nullptr;
    HIPCHK(hipMalloc(&d, 4 * sizeof(unsigned)));
    hipLaunchKernelGGL(k_philox_probe, dim3(1), dim3(1), 0, 0, ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out, d, 4 * sizeof(unsigned), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    HIPCHK(e);
    return NUSLAM_OK;
}

static int simulate_into_trace(nuslam_batch_t* h, const nuslam_sim_params* p, const double* landmarks, int n_world,
                               const double* cmd, int ticks, int m, unsigned long long seed, unsigned first_filter,
                               int known_ids, long long* empty_slots)
{
    const size_t B = (size_t)h->B, T = (size_t)ticks, M = (size_t)m;
    struct Scratch {                                  // inputs of the two generator kernels; freed on every exit path
        double* lm = nullptr; double* cmd = nullptr; unsigned long long* empty = nullptr;
        ~Scratch() { if (lm) (void)hipFree(lm); if (cmd) (void)hipFree(cmd); if (empty) (void)hipFree(empty); }
    } d;
    HIPCHK(hipMalloc(&h->tr_tw, sizeof(double) * B * T * 2));
    HIPCHK(hipMalloc(&h->tr_mx, sizeof(double) * B * T * M));
    HIPCHK(hipMalloc(&h->tr_my, sizeof(double) * B * T * M));
    HIPCHK(hipMalloc(&h->tr_ids, sizeof(int) * B * T * M));
    HIPCHK(hipMalloc(&h->tr_truth, sizeof(double) * B * T * 3));
    HIPCHK(hipMalloc(&d.lm, sizeof(double) * 2 * n_world));
    HIPCHK(hipMalloc(&d.cmd, sizeof(double) * 2 * T));
    HIPCHK(hipMalloc(&d.empty, sizeof(unsigned long long)));
    HIPCHK(hipMemcpy(d.lm, landmarks, sizeof(double) * 2 * n_world, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d.cmd, cmd, sizeof(double) * 2 * T, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(d.empty, 0, sizeof(unsigned long long)));
    SimArg a;
    a.p = *p; a.lm = d.lm; a.cmd = d.cmd; a.n_world = n_world; a.ticks = ticks; a.m = m; a.B = h->B;
    a.seed = seed; a.first_filter = first_filter;
    a.tw = h->tr_tw; a.mx = h->tr_mx; a.my = h->tr_my; a.ids = h->tr_ids; a.truth = h->tr_truth;
    a.empty = d.empty; a.raw = nullptr;
    hipLaunchKernelGGL(k_sim_path, dim3((h->B + 63) / 64), dim3(64), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    const size_t lds = (size_t)n_world * 9;
    if (lds > 48 * 1024)
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sim_markers),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (p->lidar != 0.0) {
        // markers from the simulated lidar through the landmarks node's chain; they carry no identity
        if (known_ids) return NUSLAM_E_ARG;
        HIPCHK(hipMalloc(&h->tr_scan, sizeof(float) * B * T * 360));
        hipLaunchKernelGGL(k_sim_scan, dim3(ticks, h->B), dim3(256), 0, h->stream, a, h->tr_scan);
        HIPCHK(hipGetLastError());
        int overflow[2] = { 0, 0 };
        const int src = scan_to_markers(h->stream, h->tr_scan, (int)(B * T), p->lidar_min_range, p->lidar_max_range, m,
                                        h->tr_mx, h->tr_my, h->tr_ids, d.empty, overflow);
        if (src) { g_hip_err = "scan_to_markers failed"; return src; }
        if (overflow[0]) return NUSLAM_E_CAPACITY;                 // a scan with more clusters than the per-scan table holds
        if (overflow[1]) return NUSLAM_E_ARG;                      // more accepted markers in a scan than slots: raise m
    } else {
        hipLaunchKernelGGL(k_sim_markers, dim3(ticks, h->B), dim3(256), lds, h->stream, a);
        HIPCHK(hipGetLastError());
    }
    unsigned long long empty = 0;
    HIPCHK(hipMemcpyAsync(&empty, d.empty, sizeof(empty), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (empty_slots) *empty_slots = (long long)empty;
    h->tr_ticks = ticks; h->tr_m = m; h->tr_bcast = 0;
    h->tr_presence_only = !known_ids;
    if (known_ids) {
        // the host keeps the ids only to decide, tick by tick, whether every correction is a plain one (pairing)
        std::vector<int>& dst = h->B == 1 ? h->h_ids : h->h_ids_pf;
        dst.resize(B * T * M);
        HIPCHK(hipMemcpy(dst.data(), h->tr_ids, sizeof(int) * B * T * M, hipMemcpyDeviceToHost));
    }
    return NUSLAM_OK;
}

int nuslam_batch_simulate(nuslam_batch_t* h, const nuslam_sim_params* p, const double* landmarks, int n_world,
                          const double* cmd, int ticks, int m, unsigned long long seed, unsigned first_filter,
                          int known_ids, long long* empty_slots)
{
    if (!h || !p || !landmarks || !cmd || n_world < 1 || n_world > kSimMaxWorld || ticks < 1 || m < 1)
        return NUSLAM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    free_trace(h);
    const int rc = simulate_into_trace(h, p, landmarks, n_world, cmd, ticks, m, seed, first_filter, known_ids, empty_slots);
    if (rc) {                                     // no half-made trace survives a failure, whatever step it came from
        (void)hipStreamSynchronize(h->stream);
        free_trace(h);
    }
    return rc;
}

int nuslam_batch_get_scan(nuslam_batch_t* h, int b, int tick, float out[360])
{
    if (!h || !h->tr_scan || !out || b < 0 || b >= h->B || tick < 0 || tick >= h->tr_ticks) return NUSLAM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, h->tr_scan + ((size_t)b * h->tr_ticks + tick) * 360, sizeof(float) * 360, hipMemcpyDeviceToHost));
    return NUSLAM_OK;
}

int nuslam_batch_get_trace(nuslam_batch_t* h, int b, double* tw, double* mx, double* my, int* ids, double* truth)
{
    if (!h || !h->tr_tw || b < 0 || b >= h->B) return NUSLAM_E_ARG;
    if ((ids && !h->tr_ids) || (truth && !h->tr_truth)) return NUSLAM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t bb = h->tr_bcast ? 0 : (size_t)b, T = (size_t)h->tr_ticks, M = (size_t)h->tr_m;
    if (tw) HIPCHK(hipMemcpy(tw, h->tr_tw + bb * T * 2, sizeof(double) * T * 2, hipMemcpyDeviceToHost));
    if (mx && M) HIPCHK(hipMemcpy(mx, h->tr_mx + bb * T * M, sizeof(double) * T * M, hipMemcpyDeviceToHost));
    if (my && M) HIPCHK(hipMemcpy(my, h->tr_my + bb * T * M, sizeof(double) * T * M, hipMemcpyDeviceToHost));
    if (ids && M) HIPCHK(hipMemcpy(ids, h->tr_ids + bb * T * M, sizeof(int) * T * M, hipMemcpyDeviceToHost));
    if (truth) HIPCHK(hipMemcpy(truth, h->tr_truth + bb * T * 3, sizeof(double) * T * 3, hipMemcpyDeviceToHost));
    return NUSLAM_OK;
}

int nuslam_batch_load_trace(nuslam_batch_t* h, int ticks, int m, const double* tw, const double* mx, const double* my,
                            const int* ids, int bcast)
{
    if (!h || ticks < 1 || m < 0 || !tw || (m > 0 && (!mx || !my))) return NUSLAM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    free_trace(h);
    const size_t nb = bcast ? 1 : (size_t)h->B;
    const size_t mm = m > 0 ? (size_t)m : 1;
    HIPCHK(hipMalloc(&h->tr_tw, sizeof(double) * nb * ticks * 2));
    HIPCHK(hipMemcpy(h->tr_tw, tw, sizeof(double) * nb * ticks * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&h->tr_mx, sizeof(double) * nb * ticks * mm));
    HIPCHK(hipMalloc(&h->tr_my, sizeof(double) * nb * ticks * mm));
    if (m > 0) {
        HIPCHK(hipMemcpy(h->tr_mx, mx, sizeof(double) * nb * ticks * m, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->tr_my, my, sizeof(double) * nb * ticks * m, hipMemcpyHostToDevice));
        if (ids) {
            HIPCHK(hipMalloc(&h->tr_ids, sizeof(int) * nb * ticks * m));
            HIPCHK(hipMemcpy(h->tr_ids, ids, sizeof(int) * nb * ticks * m, hipMemcpyHostToDevice));
        }
    }
    h->tr_ticks = ticks; h->tr_m = m; h->tr_bcast = bcast ? 1 : 0;
    h->h_ids.clear();
    if (ids && m > 0 && (bcast || h->B == 1)) h->h_ids.assign(ids, ids + (size_t)ticks * m);
    else if (ids && m > 0) h->h_ids_pf.assign(ids, ids + (size_t)h->B * ticks * m);
    return NUSLAM_OK;
}

int nuslam_batch_run(nuslam_batch_t* h, int t_begin, int t_end, int total_landmarks)
{
    if (!h || !h->tr_tw || t_begin < 0 || t_end > h->tr_ticks || t_begin > t_end) return NUSLAM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    { int lrc = lazy_flush(h); if (lrc) return lrc; }
    int* saved_log = h->id_log;
    h->id_log = nullptr;  // resident traces do not log resolved ids
    int rc = NUSLAM_OK;
    if (h->poisoned) { h->id_log = saved_log; return NUSLAM_E_SYNC; }
    const bool known_trace = h->tr_ids != nullptr && !h->tr_presence_only;
    const bool want_overlap = h->overlap < 0 ? (h->B == 1 && h->pass_mode != 0) : h->overlap != 0;
    if (want_overlap && h->ov_ok != 0 && known_trace && t_end - t_begin >= 2 && h->tr_m >= 1 && h->tr_m <= kTickJ && !h->deferred && !h->dense_predict &&
        (h->tick_mode != 0) && (h->tr_bcast ? !h->h_ids.empty() : true)) {
        rc = run_overlapped(h, t_begin, t_end, total_landmarks);
        if (rc != kNotConcurrent) {              // (else: the probe found the streams serialised -> the loop below)
            h->id_log = saved_log;
            return rc;
        }
        rc = NUSLAM_OK;
    }
    if (known_trace) {
        // one filter: the run as ONE launch (k_run_fused) -- in stretches of at most kRunMaxTicks ticks (~0.2 s of one kernel), and tick
        // by tick from the first stretch that does not qualify (a possible first sighting)
        constexpr int kRunMaxTicks = 4096;
        bool ran = false;
        while (t_begin < t_end) {
            const int t1 = t_end - t_begin <= kRunMaxTicks + 1 ? t_end : t_begin + kRunMaxTicks;
            rc = run_fused(h, t_begin, t1, total_landmarks);
            if (rc == kRunNotApplicable) { rc = NUSLAM_OK; break; }
            if (rc) { h->id_log = saved_log; return rc; }
            t_begin = t1;
            ran = true;
        }
        if (ran && t_begin >= t_end) {
            h->id_log = saved_log;
            h->last_tick = t_end - 1;
            return NUSLAM_OK;
        }
    }
    // a large batch on a known-id trace: G groups of filters on G streams (do_tick_grouped)
    const int G = group_count(h);
    const bool grouped = G > 1 && known_trace && h->tr_m >= 1 && tick_pipeline_pays(h, h->tr_m) && !h->deferred && !h->dense_predict &&
                         !front_fits(h, false) && t_end > t_begin && (!h->h_ids.empty() || !h->h_ids_pf.empty());
    if (grouped) {
        rc = ensure_groups(h, G);
        if (!rc) rc = ensure_tick_buffers(h);
        if (!rc) rc = groups_fork(h, G);
    }
    for (int t = t_begin; t < t_end && !rc; ++t) {
        TwistArg tw;
        tw.tw = h->tr_tw; tw.stride = h->tr_bcast ? 0 : (long long)h->tr_ticks * 2; tw.off = (long long)t * 2;
        tw.dth0 = tw.dx0 = 0.0;
        ObsArg o;
        o.a = h->tr_mx; o.b = h->tr_my; o.ids = h->tr_ids;
        o.stride = h->tr_bcast ? 0 : (long long)h->tr_ticks * h->tr_m;
        o.off = (long long)t * h->tr_m;
        o.a0 = o.b0 = 0.0; o.id0 = 0; o.cartesian = 1; o.log_slot = -1;
        const int* hid = h->h_ids.empty() ? nullptr : h->h_ids.data() + (size_t)t * h->tr_m;
        const int* pfid = h->h_ids_pf.empty() ? nullptr : h->h_ids_pf.data() + (size_t)t * h->tr_m;
        if (grouped)
            rc = do_tick_grouped(h, G, tw, o, h->tr_m, total_landmarks, hid, pfid, (long long)h->tr_ticks * h->tr_m);
        else
            rc = do_tick(h, tw, o, h->tr_m, h->tr_ids != nullptr && !h->tr_presence_only, total_landmarks, hid, nullptr, nullptr, pfid,
                         (long long)h->tr_ticks * h->tr_m);
    }
    if (grouped) { int jrc = groups_join(h, G); if (!rc) rc = jrc; }
    h->id_log = saved_log;
    if (!rc && t_end > t_begin) h->last_tick = t_end - 1;
    return rc;
}

int nuslam_batch_get_state(nuslam_batch_t* h, int b, double* out, int len) { return get_state(h, b, out, len); }
int nuslam_batch_get_cov(nuslam_batch_t* h, int b, double* out, int ld) { return get_cov(h, b, out, ld); }
int nuslam_batch_get_seen(nuslam_batch_t* h, int b, int* seen) { return get_seen(h, b, seen); }
int nuslam_batch_restore(nuslam_batch_t* h, int b, const double* state, const double* cov, int ld, int seen)
{
    return restore(h, b, state, cov, ld, seen);
}

#ifdef NUSLAM_DA_CLOCK
extern "C" int nuslam_debug_da_clock(long long out[64])
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(nuslam::g_da_clock), sizeof(long long) * 64));
    return NUSLAM_OK;
}
#endif
#ifdef NUSLAM_CHAIN_CLOCK
extern "C" int nuslam_debug_panels_clock(long long out[40])
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(nuslam::g_panels_clock), sizeof(long long) * 40));
    return NUSLAM_OK;
}
extern "C" int nuslam_debug_front_timeline(long long out[32])
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(nuslam::g_front_tl), sizeof(long long) * 32));
    const long long zero[32] = { 0 };                        // (the stamps that are maxima start over)
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(nuslam::g_front_tl), zero, sizeof(zero)));
    return NUSLAM_OK;
}
extern "C" int nuslam_debug_chain_clock(long long out[32])
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(nuslam::g_chain_clock), sizeof(long long) * 32));
    return NUSLAM_OK;
}
#endif
#ifdef NUSLAM_PHASE_CLOCK
extern "C" int nuslam_debug_phase(long long out[32])
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(nuslam::g_phase), sizeof(long long) * 32));
    return NUSLAM_OK;
}
extern "C" int nuslam_debug_hwid(unsigned* out, int n_wg)
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(nuslam::g_hwid), sizeof(unsigned) * 16 * n_wg));
    return NUSLAM_OK;
}
extern "C" int nuslam_debug_wg(long long* out, int n_wg)
{
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(nuslam::g_wg), sizeof(long long) * 2 * n_wg));
    return NUSLAM_OK;
}
#endif

int nuslam_batch_sync(nuslam_batch_t* h)
{
    if (!h) return NUSLAM_E_ARG;
    return sync_status(h);
}

int nuslam_batch_status(nuslam_batch_t* h, int clear, int* first_bad_filter, int* status_out)
{
    if (!h) return NUSLAM_E_ARG;
    return read_status(h, clear, first_bad_filter, status_out);
}

int nuslam_batch_stats(nuslam_batch_t* h, double* out, int out_len)
{
    if (!h || !out || out_len < 2 * h->L + 6) return NUSLAM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    int rc = launch_stats(h);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, h->stats, sizeof(double) * (2 * h->L + 6), hipMemcpyDeviceToHost));
    return NUSLAM_OK;
}

int nuslam_batch_set_deferred(nuslam_batch_t* h, int enable)
{
    if (!h) return NUSLAM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    { int lrc = lazy_flush(h); if (lrc) return lrc; }
    int rc = flush_pending(h);
    if (rc) return rc;
    if (enable && !h->dU) {
        const size_t bytes = sizeof(double) * (size_t)h->B * 2 * kMaxPending * h->ld;
        HIPCHK(hipMalloc(&h->dU, bytes));
        HIPCHK(hipMalloc(&h->dV, bytes));
        HIPCHK(hipMemsetAsync(h->dU, 0, bytes, h->stream));
        HIPCHK(hipMemsetAsync(h->dV, 0, bytes, h->stream));
    }
    h->deferred = enable != 0;
    return NUSLAM_OK;
}

int nuslam_ekf_set_deferred(nuslam_ekf_t* h, int enable) { return h ? nuslam_batch_set_deferred(h->core, enable) : NUSLAM_E_ARG; }

int nuslam_batch_set_pairing(nuslam_batch_t* h, int enable)
{
    if (!h) return NUSLAM_E_ARG;
    { int lrc = lazy_flush(h); if (lrc) return lrc; }
    h->pairing = enable != 0;
    return NUSLAM_OK;
}

int nuslam_batch_set_tick_mode(nuslam_batch_t* h, int mode)
{
 if (!h || mode < -1 || mode > 5) return NUSLAM_E_ARG;
    { int lrc = lazy_flush(h); if (lrc) return lrc; }
    h->run_fused = mode != 5;                      // 5: as 1 with every tick of a nuslam_batch_run a launch of its own (k_tick_fused)
    if (mode == 5) mode = 1;
    h->front = mode != 3;                          // 3: as 1 with the chain and the strips as two launches (measurement)
    h->fuse_pass = mode != 3 && mode != 4;         // 4: as 1 with the pass over P as a launch of its own behind the front (round 3's default)
    h->tick_mode = (mode == 3 || mode == 4) ? 1 : mode;
    return NUSLAM_OK;
}

int nuslam_batch_set_pass_variant(nuslam_batch_t* h, int variant)
{
    if (!h || variant < 0 || (variant > 2 && variant < 10) || (variant > 18 && variant != 20)) return NUSLAM_E_ARG;
    { int lrc = lazy_flush(h); if (lrc) return lrc; }
    h->strips_lane = variant != 20;
    if (variant == 20) variant = 0;
    if (variant >= 10) { h->pass_mode = 0; h->rank_tile = variant - 10; return NUSLAM_OK; }
    h->pass_mode = variant;
    h->rank_tile = 0;
    h->apply_units = variant != 1;
    return NUSLAM_OK;
}

int nuslam_batch_set_interleave(nuslam_batch_t* h, int groups)
{
    if (!h || (groups > 4 && (groups < 11 || groups > 14))) return NUSLAM_E_ARG;
    h->group_ring = groups < 10;                   // 10 + G: G groups whose passes do not take turns (measurement)
    if (groups >= 10) groups -= 10;
    h->groups = groups < 0 ? -1 : (groups < 1 ? 1 : groups);
    return NUSLAM_OK;
}

int nuslam_batch_set_overlap(nuslam_batch_t* h, int enable)
{
    if (!h || enable > 2) return NUSLAM_E_ARG;
    if (enable == 2 && h->stream2 && h->stream2 != h->stream) return NUSLAM_E_ARG;     // (the hook must be set before the first overlapped run)
    h->overlap = enable < 0 ? -1 : (enable != 0);
    if (enable == 2) h->ov_same_stream = true;
    return NUSLAM_OK;
}

int nuslam_batch_profile(nuslam_batch_t* h, int enable)
{
    if (!h) return NUSLAM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    { int lrc = lazy_flush(h); if (lrc) return lrc; }
    int rc = drain_profile(h);
    if (rc) return rc;
    for (int k = 0; k < NUSLAM_K_COUNT; ++k) { h->prof_ms[k] = 0; h->prof_n[k] = 0; }
    h->prof = enable != 0;
    return NUSLAM_OK;
}

int nuslam_batch_profile_read(nuslam_batch_t* h, int kernel, double* total_ms, long long* launches)
{
    if (!h || kernel < 0 || kernel >= NUSLAM_K_COUNT) return NUSLAM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    { int lrc = lazy_flush(h); if (lrc) return lrc; }
    int rc = drain_profile(h);
    if (rc) return rc;
    if (total_ms) *total_ms = h->prof_ms[kernel];
    if (launches) *launches = h->prof_n[kernel];
    h->prof_ms[kernel] = 0;
    h->prof_n[kernel] = 0;
    return NUSLAM_OK;
}

int nuslam_batch_timer_start(nuslam_batch_t* h)
{
    if (!h) return NUSLAM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    { int lrc = lazy_flush(h); if (lrc) return lrc; }
    HIPCHK(hipEventRecord(h->t0, h->stream));
    return NUSLAM_OK;
}

int nuslam_batch_timer_stop(nuslam_batch_t* h, double* elapsed_ms)
{
    if (!h || !elapsed_ms) return NUSLAM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    { int lrc = lazy_flush(h); if (lrc) return lrc; }
    HIPCHK(hipEventRecord(h->t1, h->stream));
    HIPCHK(hipEventSynchronize(h->t1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->t0, h->t1));
    *elapsed_ms = ms;
    return NUSLAM_OK;
}

namespace {
__global__ void k_fill_nan(double* __restrict__ p, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = __longlong_as_double(0x7ff8000000000000ll);
}
} // namespace

int nuslam_batch_inject_fault(nuslam_batch_t* h, int kind)
{
    if (!h || kind < 1 || kind > 2) return NUSLAM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    { int lrc = lazy_flush(h); if (lrc) return lrc; }
    { int erc = ensure_tick_buffers(h); if (erc) return erc; }
    if (kind == 1) {
        const int one = 1;
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(h->tk_sync + 2, &one, sizeof(int), hipMemcpyHostToDevice));
        return NUSLAM_OK;
    }
    const size_t n = (size_t)h->B * kTickJ * 2 * h->ld;
    hipLaunchKernelGGL(k_fill_nan, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->tk_K, n);
    hipLaunchKernelGGL(k_fill_nan, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->tk_V, n);
    HIPCHK(hipGetLastError());
    return NUSLAM_OK;
}

// ---- one filter
int nuslam_ekf_create(const double robot[3], const double* map, int n_landmarks, const double Q[9], const double R[4],
                      int dtype, int device, nuslam_ekf_t** out)
{
    if (!out || !robot || (n_landmarks > 0 && !map)) return NUSLAM_E_ARG;
    nuslam_batch* core = nullptr;
    int rc = nuslam_batch_create(1, robot, map, n_landmarks, Q, R, dtype, device, &core);
    if (rc) return rc;
    nuslam_ekf* h = new (std::nothrow) nuslam_ekf();
    if (!h) { free_batch(core); return NUSLAM_E_NOMEM; }
    h->core = core;
    core->lazy.on = true;                           // the class API driven call by call reaches the device tick by tick (lazy_flush)
    *out = h;
    return NUSLAM_OK;
}

int nuslam_ekf_set_lazy(nuslam_ekf_t* h, int enable)
{
    if (!h) return NUSLAM_E_ARG;
    int rc = lazy_flush(h->core);
    h->core->lazy.on = enable != 0;
    h->core->srv.timeout_us = enable > 1 ? enable : 1000;
    return rc;
}

int nuslam_ekf_destroy(nuslam_ekf_t* h)
{
    if (!h) return NUSLAM_OK;
    free_batch(h->core);
    delete h;
    return NUSLAM_OK;
}

int nuslam_ekf_clone(const nuslam_ekf_t* src, nuslam_ekf_t** out)
{
    if (!src || !out) return NUSLAM_E_ARG;
    nuslam_batch* s = src->core;
    HIPCHK(hipSetDevice(s->device));              // before the flush: it launches on s->stream
    { int lrc = lazy_flush(s); if (lrc) return lrc; }
    { int frc = flush_pending(s); if (frc) return frc; }
    nuslam_batch* d = nullptr;
    int rc = alloc_batch(1, s->n, s->dtype, s->device, &d);
    if (rc) return rc;
    memcpy(d->Q, s->Q, sizeof(d->Q));
    memcpy(d->R, s->R, sizeof(d->R));
    d->host_seen = s->host_seen; d->host_seen_valid = s->host_seen_valid; d->pairing = s->pairing; d->tick_mode = s->tick_mode; d->front = s->front; d->fuse_pass = s->fuse_pass;
    d->touched = s->touched; d->pass_mode = s->pass_mode; d->rank_tile = s->rank_tile; d->apply_units = s->apply_units;
    d->lazy.on = s->lazy.on;
    rc = [&]() -> int {
        HIPCHK(hipSetDevice(s->device));
        HIPCHK(hipStreamSynchronize(s->stream));
        HIPCHK(hipMemcpy(d->state[0], s->state[s->sidx], sizeof(double) * s->ld, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(d->state[1], s->state[s->sidx], sizeof(double) * s->ld, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(d->ctrl[0], s->ctrl[s->cidx], sizeof(int) * C_WORDS, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(d->ctrl[1], s->ctrl[s->cidx], sizeof(int) * C_WORDS, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(d->P(), s->P(), s->esize() * (size_t)s->p_stride, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemset(d->akey, 0x7f, sizeof(int) * 2));      // overwritten just below with the exact sentinel
        const int nokey[2] = { 0x7fffffff, 0x7fffffff };
        HIPCHK(hipMemcpy(d->akey, nokey, sizeof(nokey), hipMemcpyHostToDevice));
        return NUSLAM_OK;
    }();
    if (rc) { free_batch(d); return rc; }
    nuslam_ekf* h = new (std::nothrow) nuslam_ekf();
    if (!h) { free_batch(d); return NUSLAM_E_NOMEM; }
    h->core = d;
    *out = h;
    return NUSLAM_OK;
}

int nuslam_ekf_as_batch(nuslam_ekf_t* h, nuslam_batch_t** out)
{
    if (!h || !out) return NUSLAM_E_ARG;
    { int lrc = lazy_flush(h->core); if (lrc) return lrc; }
    *out = h->core;
    return NUSLAM_OK;
}

int nuslam_ekf_predict(nuslam_ekf_t* h, double dth, double dx, double dy)
{
    (void)dy;  // Twist2D::dy is never read by the filter (slam_library.cpp:71-148)
    if (!h) return NUSLAM_E_ARG;
    nuslam_batch* c = h->core;
    HIPCHK(hipSetDevice(c->device));
    if (c->lazy.on) {
        // the tick recorded so far goes to the device; this predict opens the next one (slam.cpp:269)
        int lrc = lazy_flush(c);
        if (lrc) return lrc;
        if (c->poisoned) return NUSLAM_E_SYNC;
        if (c->srv.used && serve_usable(c)) {
            // the last tick associated its markers: this one will too (slam.cpp:291) -- predict and the next served round are
            // launched now, behind the pass over P, so that the first associateLandmark finds the round waiting for it
            c->srv.used = false;
            TwistArg tw;
            tw.tw = nullptr; tw.stride = 0; tw.off = 0; tw.dth0 = dth; tw.dx0 = dx;
            int rc = do_predict(c, tw);
            if (!rc) rc = serve_open(c);
            return rc;
        }
        c->lazy.has_predict = true; c->lazy.dth = dth; c->lazy.dx = dx;
        return NUSLAM_OK;
    }
    TwistArg tw;
    tw.tw = nullptr; tw.stride = 0; tw.off = 0; tw.dth0 = dth; tw.dx0 = dx;
    return do_predict(c, tw);
}

int nuslam_ekf_update(nuslam_ekf_t* h, double range, double bearing, int id)
{
    if (!h) return NUSLAM_E_ARG;
    nuslam_batch* c = h->core;
    if (id < 1 || id > c->n) return NUSLAM_E_BOUNDS;
    HIPCHK(hipSetDevice(c->device));
    if (c->lazy.on) {
        if (c->poisoned) return NUSLAM_E_SYNC;
        nuslam_batch::Lazy& z = c->lazy;
        unsigned char init = 0;
        if (z.pend_init) {
            if (z.pi_id == id && z.pi_r == range && z.pi_phi == bearing) { init = 1; z.pend_init = false; }   // slam.cpp:295-297, :318
            else { int lrc = lazy_flush(c); if (lrc) return lrc; }        // an unrelated initializeLandmark: apply it on its own first
        }
        if (c->srv.open) {                                 // in a served round (associateLandmark in the loop): with the next command
            const int src = serve_update(c, range, bearing, id, init != 0);
            if (src >= 0) return src;
        }
        z.r.push_back(range); z.phi.push_back(bearing); z.id.push_back(id); z.init.push_back(init);
        if (z.id.size() >= 4 * (size_t)kTickJ) return lazy_flush(c);      // (bounded record: four rounds)
        return NUSLAM_OK;
    }
    return do_update(c, inline_obs(range, bearing, id, 0), MODE_FORCE, c->n);
}

int nuslam_ekf_associate(nuslam_ekf_t* h, double range, double bearing, int* id_out)
{
    if (!h || !id_out) return NUSLAM_E_ARG;
    nuslam_batch* c = h->core;
    HIPCHK(hipSetDevice(c->device));
    if (serve_usable(c)) {
        if (c->lazy.pend_init) { int lrc = lazy_flush(c); if (lrc) return lrc; }   // (an initializeLandmark no update() followed)
        return serve_associate(c, range, bearing, id_out);
    }
    { int lrc = lazy_flush(c); if (lrc) return lrc; }
    int rc = do_associate(c, inline_obs(range, bearing, 0, 0));
    if (!rc) rc = associate_finish(c);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    // the id, the latched status and the expired-wait count came back through pinned host memory with the kernel itself
    *id_out = c->host_word[0];
    if (c->host_word[1] == 0 && c->host_word[2] == 0) return NUSLAM_OK;
    // a full map / singular psi is what the reference reports from inside this very call
    int st = 0;
    rc = read_status(c, 1, nullptr, &st);
    return rc ? rc : st;
}

int nuslam_ekf_init_landmark(nuslam_ekf_t* h, double range, double bearing, int id)
{
    if (!h) return NUSLAM_E_ARG;
    nuslam_batch* c = h->core;
    if (id < 1 || id > c->n) return NUSLAM_E_BOUNDS;
    HIPCHK(hipSetDevice(c->device));
    if (c->lazy.on) {
        nuslam_batch::Lazy& z = c->lazy;
        if (z.pend_init) { int lrc = lazy_flush(c); if (lrc) return lrc; }   // two in a row: the first one on its own
        z.pend_init = true; z.pi_r = range; z.pi_phi = bearing; z.pi_id = id;
        return NUSLAM_OK;
    }
    View v = c->view();
    c->state_epoch++;
    return launch(c, -1, k_init_landmark, dim3(1), dim3(1), v, inline_obs(range, bearing, id, 0), c->state[c->sidx]);
}

int nuslam_ekf_tick(nuslam_ekf_t* h, double dth, double dx, double dy, int m, const double* mx, const double* my,
                    const int* known_ids, int total_landmarks, int* ids_out)
{
    (void)dy;
    const double tw[2] = { dth, dx };
    return nuslam_ekf_tick_ex(h, tw, m, mx, my, 0, known_ids, total_landmarks, ids_out);
}

int nuslam_ekf_tick_ex(nuslam_ekf_t* h, const double* twist, int m, const double* mx, const double* my, int polar,
                       const int* known_ids, int total_landmarks, int* ids_out)
{
    if (!h || m < 0 || (m > 0 && (!mx || !my))) return NUSLAM_E_ARG;
    nuslam_batch* c = h->core;
    HIPCHK(hipSetDevice(c->device));
    { int lrc = lazy_flush(c); if (lrc) return lrc; }
    const double dth = twist ? twist[0] : 0.0, dx = twist ? twist[1] : 0.0;
    int* saved_log = c->id_log;
    if (ids_out && m > 0) {
        int rc = ensure_stage(c, m);
        if (rc) return rc;
        saved_log = c->id_log;
        HIPCHK(hipMemsetAsync(c->id_log, 0, sizeof(int) * m, c->stream));
    } else {
        c->id_log = nullptr;
    }
    TwistArg tw;
    tw.tw = nullptr; tw.stride = 0; tw.off = 0; tw.dth0 = dth; tw.dx0 = dx;
    // one filter: markers and ids travel inside the kernel arguments, nothing is staged
    ObsArg o = inline_obs(0.0, 0.0, 0, polar ? 0 : 1);
    int rc = do_tick(c, tw, o, m, known_ids != nullptr, total_landmarks, known_ids, mx, my, nullptr, 0, twist == nullptr);
    c->id_log = saved_log;
    if (rc) return rc;
    if (ids_out) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (m > 0) HIPCHK(hipMemcpy(ids_out, c->id_log, sizeof(int) * m, hipMemcpyDeviceToHost));
        return sync_status(c);
    }
    return NUSLAM_OK;
}

int nuslam_ekf_len(const nuslam_ekf_t* h, int* len)
{
    if (!h || !len) return NUSLAM_E_ARG;
    *len = h->core->L;
    return NUSLAM_OK;
}

int nuslam_ekf_get_state(nuslam_ekf_t* h, double* out, int len) { return h ? get_state(h->core, 0, out, len) : NUSLAM_E_ARG; }
int nuslam_ekf_get_cov(nuslam_ekf_t* h, double* out, int ld) { return h ? get_cov(h->core, 0, out, ld) : NUSLAM_E_ARG; }
int nuslam_ekf_get_seen(nuslam_ekf_t* h, int* seen) { return h ? get_seen(h->core, 0, seen) : NUSLAM_E_ARG; }
int nuslam_ekf_restore(nuslam_ekf_t* h, const double* state, const double* cov, int ld, int seen)
{
    return h ? restore(h->core, 0, state, cov, ld, seen) : NUSLAM_E_ARG;
}
int nuslam_ekf_sync(nuslam_ekf_t* h) { return h ? sync_status(h->core) : NUSLAM_E_ARG; }
int nuslam_ekf_status(nuslam_ekf_t* h, int clear, int* status_out)
{
    return h ? read_status(h->core, clear, nullptr, status_out) : NUSLAM_E_ARG;
}

int nuslam_ekf_predict_dense(nuslam_ekf_t* h, const double* F, int ldf)
{
    if (!h || (F && ldf < h->core->L)) return NUSLAM_E_ARG;
    nuslam_batch* c = h->core;
    if (!F && !c->f_staged) return NUSLAM_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    { int lrc = lazy_flush(c); if (lrc) return lrc; }
    { int frc = flush_pending(c); if (frc) return frc; }
    if (F) {
        const size_t bytes = c->esize() * (size_t)c->p_stride;
        if (!c->wF) HIPCHK(hipMalloc(&c->wF, bytes));
        // stage F into the padded device layout in the covariance's element type.  F is caller-owned pageable host
        // memory and may be released as soon as this call returns, so the copy is synchronous.
        HIPCHK(hipMemsetAsync(c->wF, 0, bytes, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->dtype == NUSLAM_F64) {
            HIPCHK(hipMemcpy2D(c->wF, sizeof(double) * c->ld, F, sizeof(double) * ldf, sizeof(double) * c->L, c->L,
                               hipMemcpyHostToDevice));
        } else {
            std::vector<float> tmp((size_t)c->L * c->L);
            for (int j = 0; j < c->L; ++j)
                for (int i = 0; i < c->L; ++i) tmp[i + (size_t)j * c->L] = (float)F[i + (size_t)j * ldf];
            HIPCHK(hipMemcpy2D(c->wF, sizeof(float) * c->ld, tmp.data(), sizeof(float) * c->L, sizeof(float) * c->L, c->L,
                               hipMemcpyHostToDevice));
        }
        c->f_staged = true;
        c->dense_getA = false;                   // the resident matrix is the caller's again
    }
    return launch_dense(c);
}

int nuslam_ekf_use_dense_predict(nuslam_ekf_t* h, int enable)
{
    if (!h || enable < 0 || enable > 2) return NUSLAM_E_ARG;
    nuslam_batch* c = h->core;
    { int lrc = lazy_flush(c); if (lrc) return lrc; }
    if (enable == 2) {
        // the reference's own predict on the matrix cores: A = I + B resident in HBM, its two non-zeros of B rewritten by
        // every predict from (theta', twist) on the device (k_predict), then P <- A P A^T + Qbar as two dense products
        HIPCHK(hipSetDevice(c->device));
        const size_t elems = (size_t)c->p_stride;
        if (!c->wF) HIPCHK(hipMalloc(&c->wF, c->esize() * elems));
        int rc = NUSLAM_OK;
        DISPATCH_T(c, rc = launch(c, -1, k_fill_identity<T>, dim3((unsigned)((elems + 255) / 256)), dim3(256), c->L, c->ld, (T*)c->wF));
        if (rc) return rc;
        c->f_staged = true;
        c->dense_getA = true;
        c->dense_predict = true;
        return NUSLAM_OK;
    }
    if (enable && !c->f_staged) return NUSLAM_E_ARG;
    if (enable && c->dense_getA) return NUSLAM_E_ARG;        // the resident matrix is getA's now: stage a caller's F again first
    c->dense_predict = enable != 0;
    if (!enable) c->dense_getA = false;
    return NUSLAM_OK;
}

// ---- the batch reduction over RCCL / xGMI
#define NCCLCHK(expr)                                                                                  \
    do {                                                                                               \
        nuslam_comm_detail::Result r__ = (expr);                                                       \
        if (r__ != 0) {                                                                                \
            g_hip_err = std::string(#expr) + ": " + nuslam_comm_detail::api().GetErrorString(r__);    \
            return NUSLAM_E_COMM;                                                                      \
        }                                                                                              \
    } while (0)

int nuslam_comm_unique_id(unsigned char id[NUSLAM_COMM_ID_BYTES])
{
    if (!id) return NUSLAM_E_ARG;
    auto& a = nuslam_comm_detail::api();
    if (!a.so) { g_hip_err = a.err; return NUSLAM_E_COMM; }
    nuslam_comm_detail::UniqueId u;
    NCCLCHK(a.GetUniqueId(&u));
    memcpy(id, u.internal, NUSLAM_COMM_ID_BYTES);
    return NUSLAM_OK;
}

int nuslam_comm_create(const unsigned char id[NUSLAM_COMM_ID_BYTES], int world, int rank, int device, nuslam_comm_t** out)
{
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return NUSLAM_E_ARG;
    auto& a = nuslam_comm_detail::api();
    if (!a.so) { g_hip_err = a.err; return NUSLAM_E_COMM; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); return NUSLAM_E_NODEV; }
    if (device < 0 || device >= count) return NUSLAM_E_ARG;
    HIPCHK(hipSetDevice(device));
    nuslam_comm* c = new (std::nothrow) nuslam_comm();
    if (!c) return NUSLAM_E_NOMEM;
    c->world = world; c->rank = rank; c->device = device;
    nuslam_comm_detail::UniqueId u;
    memcpy(u.internal, id, NUSLAM_COMM_ID_BYTES);
    nuslam_comm_detail::Result r = a.CommInitRank(&c->comm, world, u, rank);
    if (r != 0) {
        g_hip_err = std::string("ncclCommInitRank: ") + a.GetErrorString(r);
        delete c;
        return NUSLAM_E_COMM;
    }
    *out = c;
    return NUSLAM_OK;
}

int nuslam_comm_destroy(nuslam_comm_t* c)
{
    if (!c) return NUSLAM_OK;
    (void)hipSetDevice(c->device);
    if (c->gathered) (void)hipFree(c->gathered);
    if (c->total) (void)hipFree(c->total);
    if (c->comm) (void)nuslam_comm_detail::api().CommDestroy(c->comm);
    delete c;
    return NUSLAM_OK;
}

int nuslam_comm_size(const nuslam_comm_t* c, int* world, int* rank)
{
    if (!c) return NUSLAM_E_ARG;
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    return NUSLAM_OK;
}

int nuslam_batch_reduce_stats(nuslam_batch_t* h, nuslam_comm_t* c, double* total, int total_len, double* per_rank)
{
    if (!h || !c || !total || total_len < 2 * h->L + 6 || c->device != h->device) return NUSLAM_E_ARG;
    const int len = 2 * h->L + 6;
    HIPCHK(hipSetDevice(h->device));
    if (c->cap < len) {
        HIPCHK(hipStreamSynchronize(h->stream));
        if (c->gathered) (void)hipFree(c->gathered);
        if (c->total) (void)hipFree(c->total);
        c->gathered = c->total = nullptr; c->cap = 0;
        HIPCHK(hipMalloc(&c->gathered, sizeof(double) * (size_t)c->world * len));
        HIPCHK(hipMalloc(&c->total, sizeof(double) * len));
        c->cap = len;
    }
    int rc = launch_stats(h);                       // this rank's vector into h->stats, on the handle's stream
    if (rc) return rc;
    // every rank contributes `len` doubles; rank r's row lands at gathered + r * len on every rank
    NCCLCHK(nuslam_comm_detail::api().AllGather(h->stats, c->gathered, (size_t)len, nuslam_comm_detail::kDouble, c->comm,
                                                h->stream));
    hipLaunchKernelGGL(nuslam_comm_detail::k_rank_ordered_sum, dim3((len + 255) / 256), dim3(256), 0, h->stream,
                       (const double*)c->gathered, c->world, len, c->total);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(total, c->total, sizeof(double) * len, hipMemcpyDeviceToHost));
    if (per_rank) HIPCHK(hipMemcpy(per_rank, c->gathered, sizeof(double) * (size_t)c->world * len, hipMemcpyDeviceToHost));
    return NUSLAM_OK;
}

// getStateVector + getCovariance + getSeenLandmarks in one call: the checkpoint that nuslam_ekf_restore takes back
int nuslam_ekf_snapshot(nuslam_ekf_t* h, double* state, int len, double* cov, int ld, int* seen)
{
    if (!h || !state || !cov || !seen) return NUSLAM_E_ARG;
    int rc = get_state(h->core, 0, state, len);
    if (!rc) rc = get_cov(h->core, 0, cov, ld);
    if (!rc) rc = get_seen(h->core, 0, seen);
    return rc;
}

} // extern "C"
