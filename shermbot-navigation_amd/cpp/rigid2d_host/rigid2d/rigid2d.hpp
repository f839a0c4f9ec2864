// rigid2d.hpp -- host-side SE(2) helpers with the interface of the reference's rigid2d library
// (rigid2d/include/rigid2d/rigid2d.hpp:71,150,168,254; rigid2d/src/rigid2d.cpp:9-13,187-209,254-261,294-328).
//
// Only needed where the reference's own rigid2d package is not on the include path: in a catkin workspace the
// slam node keeps linking the reference's rigid2d and only `nuslam` is replaced.  O(1) scalar math, header-only.
#ifndef NUSLAM_HOST_RIGID2D_HPP
#define NUSLAM_HOST_RIGID2D_HPP
#include <cmath>

namespace rigid2d {

constexpr double PI = 3.14159265358979323846;

constexpr bool almost_equal(double d1, double d2, double epsilon = 1.0e-12)
{
    return (d1 - d2 < epsilon) && (d2 - d1 < epsilon);
}
constexpr double deg2rad(double deg) { return (PI / 180.0) * deg; }
constexpr double rad2deg(double rad) { return (180.0 / PI) * rad; }

/// wrap to (-pi, pi] the way the reference does: atan2(sin, cos), not fmod   (rigid2d.cpp:9-13)
inline double normalize_angle(double rad) { return std::atan2(std::sin(rad), std::cos(rad)); }

struct Vector2D {
    double x = 0.0;
    double y = 0.0;
};

/// body twist; the filter reads dth and dx only (rigid2d.hpp:150-155)
struct Twist2D {
    double dth;
    double dx;
    double dy;
};

/// planar rigid transform stored as (cos, sin, x, y)
class Transform2D {
    double c_ = 1.0, s_ = 0.0, x_ = 0.0, y_ = 0.0;

public:
    Transform2D() = default;
    explicit Transform2D(const Vector2D& t) : x_(t.x), y_(t.y) {}
    explicit Transform2D(double rad) : c_(std::cos(rad)), s_(std::sin(rad)) {}
    Transform2D(const Vector2D& t, double rad) : c_(std::cos(rad)), s_(std::sin(rad)), x_(t.x), y_(t.y) {}

    const double& getCosTh() const { return c_; }
    const double& getSinTh() const { return s_; }
    const double& getX() const { return x_; }
    const double& getY() const { return y_; }

    Vector2D operator()(Vector2D v) const
    {
        Vector2D out;
        out.x = (v.x * c_) + (v.y * (-s_)) + x_;
        out.y = (v.x * s_) + (v.y * c_) + y_;
        return out;
    }

    /// adjoint map of a twist into this frame (rigid2d.cpp:254-261)
    Twist2D operator()(Twist2D tw) const
    {
        Twist2D out;
        out.dth = tw.dth;
        out.dx = (y_ * tw.dth) + (c_ * tw.dx) - (s_ * tw.dy);
        out.dy = -(x_ * tw.dth) + (s_ * tw.dx) + (c_ * tw.dy);
        return out;
    }

    Transform2D inv() const   // rigid2d.cpp:187-196
    {
        Transform2D r;
        r.c_ = c_;
        r.s_ = -s_;
        r.x_ = (-x_ * c_) + (-y_ * s_);
        r.y_ = (x_ * s_) + (-y_ * c_);
        return r;
    }

    Transform2D& operator*=(const Transform2D& rhs)   // rigid2d.cpp:198-209
    {
        const double m00 = (c_ * rhs.c_) - (s_ * rhs.s_);
        const double m10 = (s_ * rhs.c_) + (c_ * rhs.s_);
        const double m02 = (c_ * rhs.x_) - (s_ * rhs.y_) + x_;
        const double m12 = (s_ * rhs.x_) + (c_ * rhs.y_) + y_;
        c_ = m00; s_ = m10; x_ = m02; y_ = m12;
        return *this;
    }
};

inline Transform2D operator*(Transform2D lhs, const Transform2D& rhs) { return lhs *= rhs; }

/// follow a constant twist for unit time (rigid2d.cpp:294-328)
inline Transform2D integrateTwist(Twist2D& tw)
{
    if (tw.dth == 0) {
        Vector2D t;
        t.x = tw.dx;
        t.y = tw.dy;
        return Transform2D(t);
    }
    Vector2D centre;                       // centre of rotation seen from the body
    centre.x = tw.dy / tw.dth;
    centre.y = -(tw.dx / tw.dth);
    const Transform2D T_sb(centre);
    const Transform2D T_ss(tw.dth);
    const Transform2D T_bs = T_sb.inv();
    return T_bs * T_ss * T_sb;
}

} // namespace rigid2d
#endif
