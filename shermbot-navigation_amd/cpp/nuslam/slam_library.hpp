// slam_library.hpp -- the host-side mirror of the reference's EKF-SLAM class
// (nuslam/include/nuslam/slam_library.hpp:18,23-113) over the C ABI of include/nuslam_hip.h.
//
// Same namespace, class name, method names, argument meaning and error behaviour as the reference, so that
// nuslam/src/slam.cpp builds against it unchanged; every method is a call into libnuslam_hip.so (HIP kernels on
// the MI355X).  The object is a handle to device-resident state:
//   - copy construction / assignment clone the device state (the node copy-assigns a temporary, slam.cpp:157);
//   - a default-constructed object is empty, as in the reference (slam_library.cpp:35-37);
//   - the three getters return references to host mirrors that are refreshed lazily (one device-to-host copy
//     per change) and stay valid until the next mutating call -- the lifetime the reference's member
//     references have;
//   - failures the reference surfaces as Armadillo exceptions keep their type: an out-of-range landmark index
//     -> std::logic_error, a singular innovation covariance -> std::runtime_error;
//   - LAZY TICKS (include/nuslam_hip.h, nuslam_ekf_set_lazy; default on): predict() / initializeLandmark() / update() are
//     recorded by the library and reach the device as ONE tick -- predict, the serial chain, the strips and the one pass over
//     the covariance in a single launch -- at the next predict(), getter, associateLandmark(), copy, sync() or destruction, so
//     the loop of nuslam/src/slam.cpp:250-319, unchanged, runs at the rate of the one-call tick() extension.
//     associateLandmark() is one command to a resident round kernel (a mailbox in mapped host memory), getSeenLandmarks()
//     the library's host mirror, getStateVector() mapped host memory the kernels write.  A bad landmark id is still thrown by
//     the call itself; a device-side failure (singular innovation covariance) surfaces at the next synchronising call.
//     setLazy(false) restores one kernel launch -- one pass over the covariance -- per call.
//
// Vector / matrix types: arma::colvec / arma::mat when <armadillo> is available (the ROS build), otherwise the
// two small column-major value types below, which provide just what this interface touches.
#ifndef NUSLAM_HIP_SLAM_LIBRARY_HPP
#define NUSLAM_HIP_SLAM_LIBRARY_HPP

#include <cstddef>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "nuslam_hip.h"
#include "rigid2d/rigid2d.hpp"
#include "rigid2d/diff_drive.hpp"

#if !defined(NUSLAM_HIP_NO_ARMADILLO) && defined(__has_include)
#if __has_include(<armadillo>)
#include <armadillo>
#define NUSLAM_HIP_WITH_ARMADILLO 1
#endif
#endif

namespace slam_library
{
#ifdef NUSLAM_HIP_WITH_ARMADILLO
    using namespace arma;
    inline double* raw(mat& m) { return m.memptr(); }
    inline const double* raw(const mat& m) { return m.memptr(); }
#else
    /// column-major dense matrix of doubles: element (i, j) at data[i + j * n_rows]
    class mat
    {
    public:
        std::size_t n_rows = 0, n_cols = 0, n_elem = 0;
        mat() = default;
        mat(std::size_t r, std::size_t c) : n_rows(r), n_cols(c), n_elem(r * c), d_(r * c, 0.0) {}
        double& operator()(std::size_t i, std::size_t j) { return d_.at(i + j * n_rows); }
        const double& operator()(std::size_t i, std::size_t j) const { return d_.at(i + j * n_rows); }
        double& operator()(std::size_t i) { return d_.at(i); }                 // linear, column-major
        const double& operator()(std::size_t i) const { return d_.at(i); }
        double* memptr() { return d_.data(); }
        const double* memptr() const { return d_.data(); }
        const double* begin() const { return d_.data(); }
        const double* end() const { return d_.data() + d_.size(); }

    private:
        std::vector<double> d_;
    };
    class colvec : public mat
    {
    public:
        colvec() = default;
        explicit colvec(std::size_t n) : mat(n, 1) {}
    };
    typedef colvec vec;
    inline double* raw(mat& m) { return m.memptr(); }
    inline const double* raw(const mat& m) { return m.memptr(); }
#endif

    using namespace rigid2d;

    namespace detail
    {
        inline void check(int status, const char* where)
        {
            if (status == NUSLAM_OK) return;
            std::string msg = std::string(where) + ": " + nuslam_strerror(status);
            if (status == NUSLAM_E_HIP) msg += std::string(" (") + nuslam_last_hip_error() + ")";
            if (status == NUSLAM_E_BOUNDS || status == NUSLAM_E_ARG) throw std::logic_error(msg);   // arma: index out of bounds
            throw std::runtime_error(msg);                                                          // arma: inv() singular
        }
    }

    /// range-bearing of a marker at (x, y) in the robot frame (slam_library.cpp:16-22)
    inline colvec cartesian2polar(double x, double y)
    {
        colvec rb(2);
        detail::check(nuslam_cartesian2polar(x, y, raw(rb)), "cartesian2polar");
        return rb;
    }

    /// Extended Kalman Filter SLAM; the (3+2n) x (3+2n) covariance lives in the GPU's HBM.
    class ExtendedKalman
    {
    private:
        nuslam_ekf_t* h_ = nullptr;
        int len = 0;
        int n = 0;
        int dtype_ = NUSLAM_F64;
        int device_ = 0;
        // host mirrors for the const& getters
        mutable colvec state_vector;
        mutable mat covariance;
        mutable int seen_landmarks = 0;
        mutable bool state_stale_ = true, cov_stale_ = true, seen_stale_ = true;

        void touched() const { state_stale_ = cov_stale_ = seen_stale_ = true; }
        void require() const
        {
            if (!h_) throw std::logic_error("ExtendedKalman: default-constructed (empty) filter");
        }
        void adopt(nuslam_ekf_t* h)
        {
            h_ = h;
            detail::check(nuslam_ekf_len(h_, &len), "ExtendedKalman");
            n = (len - 3) / 2;
            touched();
        }

    public:
        ExtendedKalman() = default;

        /// \param robotState 3x1 (theta, x, y); \param mapState 2n x 1; \param Q 3x3 process noise; \param R 2x2 sensor noise
        /// (slam_library.cpp:39-63).  dtype/device select the covariance storage type in HBM and the GPU.
        ExtendedKalman(colvec robotState, colvec mapState, mat Q, mat R, int dtype = NUSLAM_F64, int device = 0)
            : dtype_(dtype), device_(device)
        {
            if (robotState.n_elem != 3 || mapState.n_elem % 2 != 0 || Q.n_elem != 9 || R.n_elem != 4)
                throw std::logic_error("ExtendedKalman: robotState 3x1, mapState 2nx1, Q 3x3, R 2x2 expected");
            nuslam_ekf_t* h = nullptr;
            detail::check(nuslam_ekf_create(raw(robotState), mapState.n_elem ? raw(mapState) : nullptr,
                                            static_cast<int>(mapState.n_elem / 2), raw(Q), raw(R), dtype, device, &h),
                          "ExtendedKalman");
            adopt(h);
        }

        ExtendedKalman(const ExtendedKalman& o) : dtype_(o.dtype_), device_(o.device_)
        {
            if (o.h_) {
                nuslam_ekf_t* h = nullptr;
                detail::check(nuslam_ekf_clone(o.h_, &h), "ExtendedKalman(copy)");
                adopt(h);
            }
        }
        ExtendedKalman(ExtendedKalman&& o) noexcept { swap(o); }
        ExtendedKalman& operator=(ExtendedKalman o) noexcept   // copy-and-swap: covers copy and move assignment
        {
            swap(o);
            return *this;
        }
        ~ExtendedKalman()
        {
            if (h_) nuslam_ekf_destroy(h_);
        }
        void swap(ExtendedKalman& o) noexcept
        {
            std::swap(h_, o.h_); std::swap(len, o.len); std::swap(n, o.n);
            std::swap(dtype_, o.dtype_); std::swap(device_, o.device_);
            touched(); o.touched();
        }

        /// prediction step (slam_library.cpp:65-148); recorded: opens the next tick (what was recorded before goes to the device)
        void predict(const Twist2D& tw)
        {
            require();
            detail::check(nuslam_ekf_predict(h_, tw.dth, tw.dx, tw.dy), "predict");
            touched();
        }

        /// expected range-bearing of landmark j for a caller-supplied state vector (slam_library.cpp:150-160)
        colvec computeTheoreticalMeasurement(int j, colvec state_vec)
        {
            colvec z(2);
            detail::check(nuslam_measurement(raw(state_vec), static_cast<int>(state_vec.n_elem), j, raw(z)),
                          "computeTheoreticalMeasurement");
            return z;
        }

        /// 2 x (3+2n) measurement Jacobian of landmark j at a caller-supplied state vector (slam_library.cpp:162-186)
        mat linearizedMeasurementModel(int j, colvec state_vec)
        {
            const int l = h_ ? len : static_cast<int>(state_vec.n_elem);
            if (static_cast<int>(state_vec.n_elem) < l) throw std::logic_error("linearizedMeasurementModel: short state vector");
            mat H(2, static_cast<std::size_t>(l));
            detail::check(nuslam_jacobian(raw(state_vec), l, j, raw(H)), "linearizedMeasurementModel");
            return H;
        }

        /// Mahalanobis data association (slam_library.cpp:188-253): landmark id, -1 for the gray zone, or a new id
        int associateLandmark(colvec z_i)
        {
            require();
            int id = 0;
            detail::check(nuslam_ekf_associate(h_, z_i(0), z_i(1), &id), "associateLandmark");
            touched();
            return id;
        }

        /// place landmark id from a range-bearing measurement (slam_library.cpp:255-261)
        void initializeLandmark(colvec z_id, int id)
        {
            require();
            detail::check(nuslam_ekf_init_landmark(h_, z_id(0), z_id(1), id), "initializeLandmark");
            touched();
        }

        /// measurement update for landmark id (slam_library.cpp:263-282; the twist is unused, as in the reference); recorded
        void update(const Twist2D& tw, colvec z_id, int id)
        {
            (void)tw;
            require();
            detail::check(nuslam_ekf_update(h_, z_id(0), z_id(1), id), "update");
            touched();
        }

        /// One whole iteration of the slam node's loop body (slam.cpp:250-251, 269-319) without a host round trip:
        /// predict, then per marker (x, y in the robot frame) cartesian2polar -> associate (or the given id) ->
        /// initialise / skip / stop -> update.  known_ids may be empty (data association).
        std::vector<int> tick(const Twist2D& tw, const std::vector<double>& mx, const std::vector<double>& my,
                              const std::vector<int>& known_ids, int total_landmarks, bool want_ids = true)
        {
            require();
            if (mx.size() != my.size() || (!known_ids.empty() && known_ids.size() != mx.size()))
                throw std::logic_error("tick: marker arrays differ in length");
            std::vector<int> ids(mx.size(), 0);
            detail::check(nuslam_ekf_tick(h_, tw.dth, tw.dx, tw.dy, static_cast<int>(mx.size()), mx.data(), my.data(),
                                          known_ids.empty() ? nullptr : known_ids.data(), total_landmarks,
                                          want_ids ? ids.data() : nullptr), "tick");
            touched();
            return ids;
        }

        /// P <- F P F^T + Qbar for a dense len x len Jacobian on the matrix cores (the algebra of slam_library.cpp:104)
        void predictDense(const mat& F)
        {
            require();
            if (static_cast<int>(F.n_rows) != len || static_cast<int>(F.n_cols) != len)
                throw std::logic_error("predictDense: F must be len x len");
            detail::check(nuslam_ekf_predict_dense(h_, raw(F), len), "predictDense");
            touched();
        }

        const colvec& getStateVector() const   // slam_library.cpp:284-287
        {
            if (!h_) return state_vector;
            if (state_stale_) {
                if (static_cast<int>(state_vector.n_elem) != len) state_vector = colvec(static_cast<std::size_t>(len));
                detail::check(nuslam_ekf_get_state(h_, raw(state_vector), len), "getStateVector");
                state_stale_ = false;
            }
            return state_vector;
        }

        const mat& getCovariance() const       // slam_library.cpp:289-292
        {
            if (!h_) return covariance;
            if (cov_stale_) {
                if (static_cast<int>(covariance.n_rows) != len)
                    covariance = mat(static_cast<std::size_t>(len), static_cast<std::size_t>(len));
                detail::check(nuslam_ekf_get_cov(h_, raw(covariance), len), "getCovariance");
                cov_stale_ = false;
            }
            return covariance;
        }

        const int& getSeenLandmarks() const    // slam_library.cpp:294-297
        {
            if (h_ && seen_stale_) {
                detail::check(nuslam_ekf_get_seen(h_, &seen_landmarks), "getSeenLandmarks");
                seen_stale_ = false;
            }
            return seen_landmarks;
        }

        /// apply what was recorded, wait for everything enqueued so far; surfaces latched device-side failures as exceptions
        void sync() const
        {
            if (h_) detail::check(nuslam_ekf_sync(h_), "sync");
        }

        /// lazy ticks on (default) / off: see the header comment
        void setLazy(bool on)
        {
            require();
            detail::check(nuslam_ekf_set_lazy(h_, on ? 1 : 0), "setLazy");
            touched();
        }

        /// checkpoint restore: overwrite (state, covariance, seen)
        void restore(const colvec& state, const mat& cov, int seen)
        {
            require();
            detail::check(nuslam_ekf_restore(h_, raw(state), raw(cov), len, seen), "restore");
            touched();
        }

        nuslam_ekf_t* handle() const { return h_; }
    };
}

#endif
