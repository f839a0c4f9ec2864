/*
 * nuslam_hip.h -- C ABI of the MI355X-native EKF-SLAM predict/update engine.
 *
 * This is the drop-in boundary for ONE path of sziselman/Shermbot-Navigation: the nuslam package's
 * slam_library::ExtendedKalman (nuslam/include/nuslam/slam_library.hpp:23-113, implementation
 * nuslam/src/slam_library.cpp) as driven by the slam node's loop (nuslam/src/slam.cpp:246-319).
 * Every entry point names the reference interface it replaces.  Plain pointers and sizes only; the
 * library owns all device memory.  The shared object is shermbot-navigation_amd/libnuslam_hip.so.
 *
 * Conventions
 *   - state order (theta, x, y, m1x, m1y, ...), length len = 3 + 2n; landmark ids are 1-based
 *     (slam_library.cpp:152).
 *   - matrices cross this boundary COLUMN-MAJOR, like arma::mat: X(i,j) = X[i + j*ld].
 *   - every function returns a status (0 = NUSLAM_OK).  Mutating calls are enqueue-and-return on the
 *     handle's own HIP stream; getters, nuslam_ekf_associate() and nuslam_ekf_sync() wait for it.
 *     Failures that the reference would raise as an Armadillo exception inside a call (a full map in
 *     associateLandmark -> std::logic_error, a singular innovation covariance -> std::runtime_error) are
 *     detected on the device; they make the offending step a no-op, are latched, and are returned by the
 *     next synchronising call (and by nuslam_ekf_status()).
 *   - a handle is not thread-safe; distinct handles may be driven from distinct threads.
 *   - there is no CPU fallback: without a HIP device every create call fails with NUSLAM_E_NODEV.
 */
#ifndef NUSLAM_HIP_H
#define NUSLAM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* 2: nuslam_batch_stats is 2*len + 6 doubles (trace(P) at [2*len + 4]); the status enum gained NUSLAM_E_SYNC / _COMM /
 *    _CAPACITY; nuslam_batch_set_pass_variant selects between the rank-2m pass and the exact chain; NUSLAM_K_TICK_RANK */
/* 3: nuslam_ekf_predict / _init_landmark / _update of a single filter are RECORDED and applied tick by tick (see "Lazy ticks"
 *    below; nuslam_ekf_set_lazy); nuslam_ekf_tick_ex; nuslam_batch_inject_fault; a poisoned handle (NUSLAM_E_SYNC) comes back only
 *    when every filter has been restored; nuslam_batch_set_interleave is now ON by default for large batches (see there); RETIRED:
 *    the four-corrections-per-pass kernel behind nuslam_batch_set_pairing(h, 4) (slower than pairs).  (Also since 2,
 *    not listed then: nuslam_sim_params grew by `fov` and `min_range` -- a caller compiled against the version-1 struct must be
 *    rebuilt.) */
#define NUSLAM_HIP_ABI_VERSION 3

typedef enum {
    NUSLAM_OK = 0,
    NUSLAM_E_ARG = 1,       /* null pointer / bad size / bad dtype */
    NUSLAM_E_BOUNDS = 2,    /* landmark id outside 1..n, or associateLandmark on a full map
                               (Armadillo: std::logic_error, slam_library.cpp:206-207) */
    NUSLAM_E_SINGULAR = 3,  /* innovation covariance not invertible (Armadillo inv(): std::runtime_error, :270) */
    NUSLAM_E_HIP = 4,       /* a HIP runtime call failed; nuslam_last_hip_error() has the text */
    NUSLAM_E_NODEV = 5,     /* no HIP device available */
    NUSLAM_E_NOMEM = 6,
    NUSLAM_E_COMM = 8,      /* RCCL could not be loaded or a collective failed; nuslam_last_hip_error() has the text */
    NUSLAM_E_SYNC = 9,      /* a bounded device-side wait between workgroups / streams expired (resident unknown-association
                               round, overlapped run): latched like the others; the results of that run are invalid */
    NUSLAM_E_CAPACITY = 7   /* a fixed-size device table is too small for this input (nuslam_batch_simulate with lidar:
                               a scan left more clusters than the per-scan table of 64 holds) */
} nuslam_status;

typedef enum { NUSLAM_F64 = 0, NUSLAM_F32 = 1 } nuslam_dtype; /* storage type of the covariance in HBM */

typedef struct nuslam_ekf nuslam_ekf_t;     /* one filter            == slam_library::ExtendedKalman */
typedef struct nuslam_batch nuslam_batch_t; /* B independent filters (Monte-Carlo trials), same n     */

const char* nuslam_strerror(int status);
const char* nuslam_last_hip_error(void);
int nuslam_abi_version(void);
/* "csrc=<16 hex digits>": a content hash of the kernel sources this library was built from (profiles/ records carry
 * it so that a counter measurement is only ever quoted beside the code it was taken from) */
const char* nuslam_build_info(void);
int nuslam_device_count(int* count);

/* ------------------------------------------------------------------ host-side helpers (pure functions) */
/* slam_library::cartesian2polar, slam_library.cpp:16-22 */
int nuslam_cartesian2polar(double x, double y, double out_range_bearing[2]);
/* ExtendedKalman::computeTheoreticalMeasurement(j, state_vec), slam_library.cpp:150-160 */
int nuslam_measurement(const double* state, int len, int j, double out_range_bearing[2]);
/* ExtendedKalman::linearizedMeasurementModel(j, state_vec), slam_library.cpp:162-186; H is 2 x len, ld 2 */
int nuslam_jacobian(const double* state, int len, int j, double* H);
/* The map -> odom transform the slam node broadcasts every tick, EKFSlam::broadcast_map2odom_tf, slam.cpp:175-210:
 * T_mo = T_mb * T_ob^-1 with T_mb from the filter's pose (state[0..2] = theta, x, y) and T_ob from the odometry
 * model's pose odom = {x, y, theta}; out = {x, y, yaw}, yaw = normalize_angle(asin(sin theta_mo)) (:194). */
int nuslam_map_to_odom(const double odom_xyth[3], const double state[3], double out_xyyaw[3]);

/* ------------------------------------------------------------------ one filter */
/* ExtendedKalman(colvec robotState, colvec mapState, mat Q, mat R), slam_library.cpp:39-63 (+ initCov :24-33).
 * map has 2*n_landmarks entries.  dtype selects how the covariance is stored in HBM; state and all
 * O(len) arithmetic stay fp64. */
int nuslam_ekf_create(const double robot[3], const double* map, int n_landmarks, const double Q[9],
                      const double R[4], int dtype, int device, nuslam_ekf_t** out);
int nuslam_ekf_destroy(nuslam_ekf_t* h);
/* copy construction / copy assignment of the value type (slam.cpp:157) */
int nuslam_ekf_clone(const nuslam_ekf_t* h, nuslam_ekf_t** out);

/* Lazy ticks (default ON for handles made by nuslam_ekf_create).  The reference's node drives the filter call by call
 * (slam.cpp:269 predict, :296 initializeLandmark, :318 update); one pass over the covariance per update() is 2 len^2 w bytes each.
 * With lazy ticks nuslam_ekf_predict / _init_landmark / _update only RECORD the call; what has been recorded reaches the device as
 * ONE tick -- the kernels of nuslam_ekf_tick_ex (predict, serial chain, strips and the one rank-2m pass over the covariance as ONE
 * launch, csrc/ekf_fused.h), bit-identical to that entry on the same inputs -- when the next nuslam_ekf_predict arrives or when anything looks at
 * the filter: a getter, nuslam_ekf_associate, _clone, _sync, _status, _snapshot, _restore, _tick, any nuslam_batch_* call on the
 * handle nuslam_ekf_as_batch returned, _destroy (discards).  Consequences a caller can observe:
 *   - argument errors (a landmark id outside 1..n) are still returned by the call itself; failures detected on the device (a
 *     singular innovation covariance) are latched as before and surface at the next synchronising call;
 *   - fewer than four recorded corrections, deferred mode and the dense predict go through the per-call kernels, as with lazy off;
 *   - an initializeLandmark(z, id) directly followed by update(z, id) with the same z and id (slam.cpp:295-318) becomes that
 *     correction's first-sighting flag; any other initializeLandmark is applied on its own, in order;
 *   - nuslam_ekf_get_seen answers from the host's mirror of `seen` while that is exact (predict / initializeLandmark / update
 *     never move it, slam_library.cpp:188-253) without touching the device; nuslam_ekf_get_state applies what was recorded and reads
 *     the state from mapped host memory that the tick's launch itself writes (no stream synchronisation, no device-to-host copy:
 *     the pass over the covariance may still be running when it returns);
 *   - with nuslam_ekf_associate in the loop (slam.cpp:291) the tick runs as a round SERVED to the host: a resident kernel that takes
 *     the caller's calls from a mailbox in mapped pinned host memory -- associate(z) is one ~3 us round trip plus the O(len) work of
 *     the association and of the correction the caller decided on for the previous marker (nuslam_ekf_update / _init_landmark only
 *     record that decision; it travels with the next call), the pass over the covariance follows once per tick.  Verdicts, ids and
 *     `seen` are those of the per-call kernels.  The round ends with the next predict / getter / sync, after 16 markers, or by itself
 *     when no call arrives for 1 ms (it can be re-opened at once: nothing is lost, a correction not yet sent takes the per-call
 *     kernels); no wave ever waits unboundedly for the host.
 * enable == 0: every call launches its own kernels again (one pass over the covariance per update()); enable > 1: on, and a served
 * round closes itself after `enable` microseconds without a call (default 1000; tests use small values to exercise that path). */
int nuslam_ekf_set_lazy(nuslam_ekf_t* h, int enable);

/* ExtendedKalman::predict(const Twist2D&), slam_library.cpp:65-148.  dy is accepted and ignored, as there. */
int nuslam_ekf_predict(nuslam_ekf_t* h, double dth, double dx, double dy);
/* The covariance propagation of slam_library.cpp:104 for a caller-supplied dense Jacobian:
 * P <- F P F^T + Qbar, two len^3 products on the matrix cores (MFMA).  F is host memory, len x len,
 * column-major, leading dimension ldf.  The state is not touched. */
int nuslam_ekf_predict_dense(nuslam_ekf_t* h, const double* F, int ldf);
/* F stays resident in HBM after the call above; F == NULL there re-uses the resident Jacobian.
 * nuslam_ekf_use_dense_predict(h, 2): every later predict of this filter (nuslam_ekf_predict, _tick, nuslam_batch_run)
 * IS the reference's predict on the matrix cores (BASELINE config "MFMA F P F^T path enabled"): the state advances as
 * slam_library.cpp:71-94, A = I + B of getA (:127-148, evaluated at the advanced heading) is kept resident -- the device
 * rewrites its two non-zeros of B from this tick's twist -- and the covariance becomes A P A^T + Qbar as two dense len^3
 * products (:104), not the two-non-zero shortcut.  For fp64 storage the result is bit-identical to the shortcut (the f64
 * MFMA is a k-ordered fma chain and the extra terms are exact zeros).
 * enable == 1: as 2 but with whatever Jacobian the last nuslam_ekf_predict_dense staged, unchanged from tick to tick (a
 * GEMM measurement, not the reference's predict); 0: back to the shortcut. */
int nuslam_ekf_use_dense_predict(nuslam_ekf_t* h, int enable);
/* ExtendedKalman::update(const Twist2D&, colvec z, int id), slam_library.cpp:263-282 (the twist is unused there). */
int nuslam_ekf_update(nuslam_ekf_t* h, double range, double bearing, int id);
/* ExtendedKalman::associateLandmark(colvec z), slam_library.cpp:188-253.  Synchronises. */
int nuslam_ekf_associate(nuslam_ekf_t* h, double range, double bearing, int* id_out);
/* ExtendedKalman::initializeLandmark(colvec z, int id), slam_library.cpp:255-261 */
int nuslam_ekf_init_landmark(nuslam_ekf_t* h, double range, double bearing, int id);

/* One iteration of the slam node's loop body, slam.cpp:250-251 + 269-319, entirely on the device:
 * predict(twist), then for each of the m markers (x, y in the robot frame): cartesian2polar ->
 * associateLandmark (or the caller's known id) -> initializeLandmark if id > seen-at-tick-start /
 * skip if id < 0 / stop if id > total_landmarks -> update.  known_ids == NULL selects data association.
 * ids_out (m ints) may be NULL; when given the call synchronises and reports the id each marker resolved to. */
int nuslam_ekf_tick(nuslam_ekf_t* h, double dth, double dx, double dy, int m, const double* mx,
                    const double* my, const int* known_ids, int total_landmarks, int* ids_out);

/* nuslam_ekf_tick with two more degrees of freedom: twist == NULL continues a tick whose predict has already run (no predict, the
 * cached `seen` of slam.cpp:251 and the break flag stay as they are); polar != 0: a / b hold (range, bearing) pairs -- markers the
 * caller has already put through cartesian2polar (slam.cpp:286) -- instead of (x, y). */
int nuslam_ekf_tick_ex(nuslam_ekf_t* h, const double twist_dth_dx[2], int m, const double* a, const double* b, int polar,
                       const int* known_ids, int total_landmarks, int* ids_out);

/* getStateVector / getCovariance / getSeenLandmarks, slam_library.cpp:284-297.  Synchronise. */
int nuslam_ekf_len(const nuslam_ekf_t* h, int* len);
int nuslam_ekf_get_state(nuslam_ekf_t* h, double* out, int len);
int nuslam_ekf_get_cov(nuslam_ekf_t* h, double* out, int ld);
int nuslam_ekf_get_seen(nuslam_ekf_t* h, int* seen);
/* overwrite (state, covariance, seen): checkpoint restore / warm-start fixtures.  Waits for every stream of the handle, clears
 * the filter's strip scratch; a handle poisoned by NUSLAM_E_SYNC accepts ticks again once EVERY one of its filters was restored. */
int nuslam_ekf_restore(nuslam_ekf_t* h, const double* state, const double* cov, int ld, int seen);
/* checkpoint: state (len), covariance (column-major, leading dimension ld) and seen in one call -- the reference
 * keeps these three members (slam_library.hpp:26-33); nuslam_ekf_restore takes them back.  Synchronises. */
int nuslam_ekf_snapshot(nuslam_ekf_t* h, double* state, int len, double* cov, int ld, int* seen);
int nuslam_ekf_sync(nuslam_ekf_t* h);
/* latched device-side status (see Conventions); clear != 0 resets it */
int nuslam_ekf_status(nuslam_ekf_t* h, int clear, int* status_out);

/* ------------------------------------------------------------------ B independent filters */
/* All filters share n, Q, R and dtype; robot is 3*B, map is 2*n*B (filter-major); either may be NULL (zeros). */
int nuslam_batch_create(int n_filters, const double* robot, const double* map, int n_landmarks,
                        const double Q[9], const double R[4], int dtype, int device, nuslam_batch_t** out);
int nuslam_batch_destroy(nuslam_batch_t* h);
int nuslam_batch_size(const nuslam_batch_t* h, int* n_filters, int* len);
/* Make a trace resident in HBM: for filter b, tick t: twist tw[(b*ticks + t)*2 + {0,1}] = (dth, dx) and m
 * markers mx/my[(b*ticks + t)*m + i] with ids ids[(b*ticks + t)*m + i] (ids == NULL: data association).
 * bcast != 0: the arrays describe ONE filter and every filter replays them. */
int nuslam_batch_load_trace(nuslam_batch_t* h, int ticks, int m, const double* tw, const double* mx,
                            const double* my, const int* ids, int bcast);
/* Run ticks [t_begin, t_end) of the resident trace on every filter (the loop body of slam.cpp:250-319).
 * ONE filter, known ids, no tick of the run able to hold a first sighting (the host proves it from the ids), the default tick mode:
 * the run is ONE launch (csrc/ekf_fused.h, k_run_fused) -- the covariance is read when it begins, stays in the pass workgroups'
 * registers from tick to tick (a tick stores only the rows / columns of the next tick's index set) and is complete in memory again
 * when the call's work has finished, i.e. for every getter, copy and later call.  Same bits as a launch per tick. */
int nuslam_batch_run(nuslam_batch_t* h, int t_begin, int t_end, int total_landmarks);
/* one filter's results (synchronise) */
int nuslam_batch_get_state(nuslam_batch_t* h, int b, double* out, int len);
int nuslam_batch_get_cov(nuslam_batch_t* h, int b, double* out, int ld);
int nuslam_batch_get_seen(nuslam_batch_t* h, int b, int* seen);
int nuslam_batch_restore(nuslam_batch_t* h, int b, const double* state, const double* cov, int ld, int seen);
int nuslam_batch_sync(nuslam_batch_t* h);
int nuslam_batch_status(nuslam_batch_t* h, int clear, int* first_bad_filter, int* status_out);
/* Batch statistics for the Monte-Carlo reduction: out = { sum_b state (len), sum_b state^2 (len),
 * sum_b (estimate - truth)^2 of the pose (theta wrapped, x, y: 3), sum_b NEES (e^T P[0:3,0:3]^-1 e), sum_b trace(P),
 * count } -- 2*len + 6 doubles, every sum accumulated in filter order (deterministic).  The truth is the simulated
 * robot's pose after the last tick nuslam_batch_run applied (traces made by nuslam_batch_simulate keep it); with any
 * other trace the four truth-dependent entries are 0. */
int nuslam_batch_stats(nuslam_batch_t* h, double* out, int out_len);
/* The batch reduction over RCCL / xGMI (one process per GPU, the filters sharded over the ranks, no
 * collective in the data path).  Rank 0 makes an id and hands it to the other ranks by any channel the host program
 * has (a file, MPI, torch.distributed ...); every rank then creates its communicator on its own device.
 * nuslam_batch_reduce_stats: ncclAllGather of every rank's nuslam_batch_stats vector on the batch's stream, then
 * the rows added in rank order on the device -- the same bits on every rank and for every ring order.  total gets
 * 2*len + 6 doubles; per_rank (may be NULL) world x (2*len + 6).  Collective: every rank must call it. */
#define NUSLAM_COMM_ID_BYTES 128
typedef struct nuslam_comm nuslam_comm_t;
int nuslam_comm_unique_id(unsigned char id[NUSLAM_COMM_ID_BYTES]);
int nuslam_comm_create(const unsigned char id[NUSLAM_COMM_ID_BYTES], int world, int rank, int device, nuslam_comm_t** out);
int nuslam_comm_destroy(nuslam_comm_t* c);
int nuslam_comm_size(const nuslam_comm_t* c, int* world, int* rank);
int nuslam_batch_reduce_stats(nuslam_batch_t* h, nuslam_comm_t* c, double* total, int total_len, double* per_rank);
/* a one-filter view for the single-filter API above is the batch of size 1: */
int nuslam_ekf_as_batch(nuslam_ekf_t* h, nuslam_batch_t** out); /* borrowed; do not destroy */

/* Deferred application (opt-in; known association only).  The corrections of a tick are kept as rank-2 factors,
 * P_j = P_0 - sum U_i V_i with U_i = K_i and V_i = H_i P_{i-1} -- the reference's update (slam_library.cpp:270-279)
 * re-associated -- and the covariance is rewritten once per tick (before the next predict, or when a getter,
 * association or 16 pending corrections force it): 2 len^2 w bytes per TICK instead of per correction.
 * Results agree with the default eager path to rounding, not bit for bit. */
int nuslam_batch_set_deferred(nuslam_batch_t* h, int enable);
/* Pairing (default on): inside a tick with known ids, consecutive corrections of already-initialised landmarks are
 * applied two at a time by one pass over the covariance (k_update2) -- same arithmetic, same bits, half the HBM
 * bytes per correction.  enable = 0 forces one k_update launch per correction. */
int nuslam_batch_set_pairing(nuslam_batch_t* h, int enable);
int nuslam_ekf_set_deferred(nuslam_ekf_t* h, int enable);
/* How a known-id tick (nuslam_ekf_tick with known_ids, nuslam_batch_run on a trace with ids) applies its corrections.
 * mode 1 (default): the tick pipeline -- the serial part of all corrections first (a 35 x 35 block and 35 state
 * entries carried through the corrections: H, S^-1, K, the innovation), then the O(len) gain / prior-row strips,
 * then ONE pass over the covariance that carries every tile through all corrections: 2 len^2 w bytes per tick.
 * For ONE filter the whole tick is one launch where it can be (csrc/ekf_fused.h: predict, chain, strips and the pass over the
 * covariance as workgroups of one grid -- the pass's tile loads run under the serial chain); mode 4: as 1 with the pass as a launch
 * of its own behind the front launch; mode 3: chain, strips and pass as three launches (measurement); mode 5: as 1 with every tick
 * of a nuslam_batch_run a launch of its own (mode 1 runs the ticks of one filter's known-id run as ONE launch, see nuslam_batch_run).
 * Same bits in 1, 3, 4 and 5.
 * mode 0: one pass per correction (or per pair, see nuslam_batch_set_pairing).  Same arithmetic on every element in
 * the same order: the two modes produce identical bits.  mode -1 (default): the library picks per handle.
 * Ticks with UNKNOWN association follow the same switch: mode 1 = tracked rows / columns / diagonal blocks of the
 * covariance, one O(len) step per marker and one pass over the covariance per tick (csrc/ekf_da.h) -- as ONE resident
 * launch per round while the handle's workgroups fit the chip, else one launch per marker; mode 2 = always one launch
 * per marker; mode 0 = associate + one pass per marker.  Identical bits in every mode. */
int nuslam_batch_set_tick_mode(nuslam_batch_t* h, int mode);
/* nuslam_batch_run on a known-id trace in tick-pipeline mode: enable > 0 lets the serial chain of tick t+1 run on a
 * second stream while the strips and the pass over P of tick t run on the handle's (the host knows the next tick's
 * markers from the resident trace; the streams hand over through device counters, every wait bounded).  Same bits
 * either way.  (While a handle's chains and strips fit the chip together, the strips are a launch that FOLLOWS the chain
 * of the other stream plan entry by plan entry instead of starting when it has ended.)  Before its first overlapped run a
 * handle PROBES whether the two streams really execute side by side (a
 * profiler's counter pass or a serialising environment makes them take turns, and every hand-off would expire): if not,
 * its runs take the one-stream path -- same bits, no dependency between streams.  enable == 2: test hook, as 1 but the
 * "second stream" is the handle's own, so the probe must find them serialised.  enable < 0 (default): off with the
 * rank-2m pass (one stream is the faster order there), on for one filter / off for batches with the exact chain.
 * Should a hand-off expire all the same (NUSLAM_E_SYNC), the handle refuses further ticks until it is restored. */
int nuslam_batch_set_overlap(nuslam_batch_t* h, int enable);
/* nuslam_batch_run of a LARGE batch on a known-id trace: the filters run as `groups` (1..4) groups, each on a stream of its own, the
 * groups' passes over the covariance taking turns, so that one group's HBM-bound pass runs beside the other groups' latency-bound
 * chains and their strips.  Same kernels on the same per-filter data: same bits for every group count.  groups < 0 (default): 2
 * groups for 512 filters and more, else one (1024 x N = 200 on one MI355X: 2 groups 728 us per tick, one 781, 3: 752, 4: 799).
 * 10 + G: G groups whose passes do not take turns (measurement).  Every call that looks at the handle afterwards waits for all of
 * its streams. */
int nuslam_batch_set_interleave(nuslam_batch_t* h, int groups);
/* How a tick pipeline's ONE pass over the covariance applies the round's corrections.
 *   0 (default)  as a rank-2m update on the matrix cores: update()'s P <- (I - K H) P (slam_library.cpp:279) re-associated
 *                as P - K (H P), all corrections of the round in one v_mfma_f64 accumulation per tile -- 2 FMAs per element
 *                and correction instead of 7, a streaming kernel.  Same algebra, different rounding: agrees with the exact
 *                chain to ~1e-13 per entry from a warm state, NOT bit for bit.  A round that holds a first sighting (the
 *                INT_MAX diagonal of slam_library.cpp:30 is being cancelled) is applied by the exact chain, decided per
 *                filter on the device from the round's plan.
 *   1            always the exact chain (bit-identical to one update() per marker), plain kernel
 *   2            always the exact chain, the two-unit kernel for one large fp64 filter that has the chip to itself
 *   10 + k       as 0 with tile shape k = 0..8 (measurement only)
 *   20           as 0 with a batch's strips by k_tick_panels (a quad of lanes per index, the round's plan in LDS) where the default takes
 *                k_tick_strips_lane (a lane per index, the plan through the scalar cache): same bits, measurement / test only */
int nuslam_batch_set_pass_variant(nuslam_batch_t* h, int variant);

/* ------------------------------------------------------------------ Monte-Carlo trace generator */
/* The simulator's loop, nuturtlesim/src/tube_world.cpp:509-533, run on the device for every filter of a batch and
 * written straight into the resident trace nuslam_batch_run() replays -- per-filter worlds without host involvement.
 * Parameter names and defaults follow nuturtlesim/config/tube_world_params.yaml and
 * nuturtle_description/config/diff_params.yaml. */
typedef struct nuslam_sim_params {
    double wheel_base, wheel_radius;   /* diff_params.yaml:2-3 */
    double dt;                         /* loop period, 1 / frequency (tube_world.cpp:66: 50 Hz) */
    double twist_noise;                /* sigma of the Gaussian added to the commanded dth and dx, :177-189 */
    double slip_min, slip_max;         /* wheel-slip noise N((min + max) / 2, max - mean), :480-483 */
    double tube_radius, robot_radius;  /* collision slide, :371-389 */
    double tube_var;                   /* constant offset added to both marker coordinates, :311-312 */
    double marker_sigma;               /* extension: Gaussian marker noise on top (0 = the reference's behaviour) */
    double max_range;                  /* range gate :300-307; <= 0: every tube every tick (what the slam node sees) */
    double lidar;                      /* != 0: the markers of a tick come from the simulated lidar instead of set_rel_markers:
                                          simulate_lidar_scanner (tube_world.cpp:405-471) -> the landmarks node's chain
                                          clusterPoints -> classifyCluster -> circleFit (nuslam/src/landmarks.cpp:63, 82-108),
                                          all on the device; data association only (scan markers carry no identity) */
    double lidar_min_range, lidar_max_range;   /* nuturtlesim/config/scan_params.yaml: minimum_range 0.05, maximum_range 1.0 */
    double fov;                        /* extension: half-angle (rad) of the marker sensor's field of view; <= 0: all around (the
                                          reference's behaviour).  The reference's update() does not wrap the bearing innovation
                                          (slam_library.cpp:272), so a tube behind the robot yields a 2 pi innovation: Monte-Carlo
                                          worlds that are to stay in the filter's working regime use a limited field of view */
    double min_range;                  /* extension, with fov > 0: tubes nearer than this are not reported either (at a few cm the
                                          marker noise alone carries a bearing across the +-pi cut) */
} nuslam_sim_params;
/* Generate `ticks` ticks for every filter: landmarks = {x0, y0, x1, y1, ...} (n_world tubes, shared by all filters),
 * cmd = ticks x (dth, dx) commanded body twists (the /cmd_vel stream, shared).  Filter b draws from the random streams
 * of global filter index first_filter + b, so a sharded batch reproduces the unsharded one.  Each tick keeps the (at
 * most m) nearest tubes within max_range in tube order; unused slots get id -1, which the tick's decision chain skips
 * (slam.cpp:298-300).  known_ids != 0: the trace carries the tube index + 1 as the landmark id; 0: data association
 * (the ids then only mark which slots hold a marker; an empty slot is a marker the node never received, so
 * associateLandmark is not called for it).  empty_slots (may be NULL) receives the number of unused slots.
 * Synchronises. */
int nuslam_batch_simulate(nuslam_batch_t* h, const nuslam_sim_params* p, const double* landmarks, int n_world,
                          const double* cmd, int ticks, int m, unsigned long long seed, unsigned first_filter,
                          int known_ids, long long* empty_slots);
/* Read filter b's resident trace back (any pointer may be NULL): tw ticks x 2, mx / my / ids ticks x m, truth
 * ticks x 3 (theta, x, y of the simulated robot after each tick; only for generated traces). */
int nuslam_batch_get_trace(nuslam_batch_t* h, int b, double* tw, double* mx, double* my, int* ids, double* truth);
/* the simulated lidar scan (360 ranges, robot frame) of filter b at tick t; only after a generation with lidar != 0 */
int nuslam_batch_get_scan(nuslam_batch_t* h, int b, int tick, float out_ranges[360]);
/* hook for the device's rigid2d::normalize_angle (rigid2d/src/rigid2d.cpp:9-13; on the device a two-constant range
 * reduction, csrc/ekf_device.h): out[i] = normalize_angle(in[i]), computed on the device */
int nuslam_device_normalize_angle(const double* in, int n, double* out, int device);
/* bit-exact hook for the generator's RNG: the Philox4x32-10 block of (seed, counter), computed on the device */
int nuslam_philox4x32_10(const unsigned ctr[4], const unsigned key[2], unsigned out[4], int device);

/* ------------------------------------------------------------------ landmark extraction */
/* circle_fit::circleFit (nuslam/src/circle_fit_library.cpp:15-134) and circle_fit::classifyCluster (:208-250) for a
 * batch of lidar clusters, one wave per cluster.  Cluster c owns points [offsets[c], offsets[c+1]) of xs / ys (host
 * memory).  Per cluster: centre, radius (the marker's scale.x is 2 * radius, :124), status 0 ok / 1 fewer than four
 * points (the reference returns marker.id = -1, :73-77) / 2 singular 4x4 system; optionally classifyCluster's verdict
 * and the standard deviation of the inscribed angles in degrees.  kernel_ms (may be NULL) receives the device time. */
int nuslam_circle_fit_batch(int n_clusters, const int* offsets, const double* xs, const double* ys, double* centre_x,
                            double* centre_y, double* radius, int* status, int* is_circle, double* angle_std_dev,
                            int device, double* kernel_ms);

/* ------------------------------------------------------------------ measurement hooks */
typedef enum {
    NUSLAM_K_PREDICT = 0,
    NUSLAM_K_ASSOCIATE = 1,
    NUSLAM_K_UPDATE = 2,       /* the streaming correction sweep: the HBM-bound kernel */
    NUSLAM_K_DENSE_GEMM = 3,   /* the two MFMA products of nuslam_ekf_predict_dense */
    NUSLAM_K_UPDATE_DEFERRED = 4, /* one correction in factor form, O(len) (deferred mode) */
    NUSLAM_K_FLUSH = 5,        /* the rank-2J pass that applies a tick's pending corrections (deferred mode) */
    NUSLAM_K_UPDATE2 = 6,      /* two consecutive corrections in one pass over P (bit-identical to two NUSLAM_K_UPDATE) */
    NUSLAM_K_TICK_CHAIN = 7,   /* tick pipeline: the serial part of a round of corrections (one workgroup per filter) */
    NUSLAM_K_TICK_PANELS = 8,  /* tick pipeline: the O(len) gain / prior-row strips of the round */
    NUSLAM_K_TICK_APPLY = 9,   /* tick pipeline: the one pass over P that applies the whole round -- the HBM-bound kernel */
    NUSLAM_K_TICK_NEXT = 10,   /* (no longer launched: the replay of the next tick's starting block is part of NUSLAM_K_TICK_CHAIN) */
    NUSLAM_K_DA_BEGIN = 11,    /* unknown-association tick pipeline: tracked rows / columns / diagonal blocks out of P */
    NUSLAM_K_DA_STEP = 12,     /* ... one correction: association verdict, strips, tracked entries, next marker's candidates */
    NUSLAM_K_TICK_RANK = 13,   /* tick pipeline: the one pass over P as a rank-2m update on the matrix cores -- the HBM-bound kernel */
    NUSLAM_K_COUNT = 14
} nuslam_kernel_id;
/* When enabled, every launch of the listed kernels carries its own pair of HIP events on the handle's
 * stream (hipExtLaunchKernelGGL start/stop events: the dispatch's own begin/end timestamps). */
int nuslam_batch_profile(nuslam_batch_t* h, int enable);
/* Synchronises; returns the summed duration and launch count since the last read, then resets. */
int nuslam_batch_profile_read(nuslam_batch_t* h, int kernel, double* total_ms, long long* launches);
/* HIP-event stopwatch on the handle's stream (for whole-region timing from a host language) */
int nuslam_batch_timer_start(nuslam_batch_t* h);
int nuslam_batch_timer_stop(nuslam_batch_t* h, double* elapsed_ms); /* synchronises */
/* Fault injection for the recovery paths (tests).  kind 1: count one expired device-side wait, as a hand-off between workgroups
 * or streams that never arrived would: the next status read reports NUSLAM_E_SYNC and poisons the handle.  kind 2: put NaN into
 * every filter's gain / factor strips, as a diverged run would leave them. */
int nuslam_batch_inject_fault(nuslam_batch_t* h, int kind);

#ifdef __cplusplus
}
#endif
#endif
