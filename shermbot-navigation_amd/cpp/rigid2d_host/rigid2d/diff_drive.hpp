// diff_drive.hpp -- host-side differential-drive odometry with the interface of the reference's
// rigid2d::DiffDrive (rigid2d/include/rigid2d/diff_drive.hpp:13-97, rigid2d/src/diff_drive.cpp:20-146).
// It feeds the filter its Twist2D; O(1) scalar math, header-only.  See rigid2d.hpp for when this is used.
#ifndef NUSLAM_HOST_DIFF_DRIVE_HPP
#define NUSLAM_HOST_DIFF_DRIVE_HPP
#include <cmath>

#include "rigid2d.hpp"

namespace rigid2d {

struct wheelVel {
    double uL;
    double uR;
};

class DiffDrive {
    double wheelBase = 0.0, wheelRad = 0.0;
    double x = 0.0, y = 0.0, th = 0.0;
    double thL = 0.0, thR = 0.0;

    Twist2D wheelsToTwist(double thLnew, double thRnew) const   // diff_drive.cpp:83-90 == :114-121
    {
        const double dUL = thLnew - thL;
        const double dUR = thRnew - thR;
        Twist2D t;
        t.dth = (wheelRad / wheelBase) * (dUR - dUL);
        t.dx = (wheelRad / 2) * (dUL + dUR);
        t.dy = 0.0;
        return t;
    }

public:
    DiffDrive() = default;
    DiffDrive(double base, double rad, double xx, double yy, double theta, double left, double right)
        : wheelBase(base), wheelRad(rad), x(xx), y(yy), th(theta), thL(left), thR(right)
    {
    }

    const double& getWheelBase() const { return wheelBase; }
    const double& getWheelRad() const { return wheelRad; }
    const double& getX() const { return x; }
    const double& getY() const { return y; }
    const double& getTh() const { return th; }
    const double& getThL() const { return thL; }
    const double& getThR() const { return thR; }

    wheelVel convertTwist(const Twist2D& tw)   // diff_drive.cpp:66-78
    {
        const double d = wheelBase / 2;
        const double r = wheelRad;
        wheelVel u;
        u.uL = (-(d / r) * tw.dth) + (tw.dx / r);
        u.uR = ((d / r) * tw.dth) + (tw.dx / r);
        return u;
    }

    /// body twist for new wheel angles; does NOT store them (diff_drive.cpp:80-110)
    Twist2D getTwist(double thLnew, double thRnew) { return wheelsToTwist(thLnew, thRnew); }

    /// integrate the pose and store the wheel angles (diff_drive.cpp:111-146); th is never normalised
    DiffDrive& operator()(double thLnew, double thRnew)
    {
        Twist2D twb = wheelsToTwist(thLnew, thRnew);
        const Transform2D Tbb = integrateTwist(twb);
        Twist2D dqb;
        dqb.dth = std::atan(Tbb.getSinTh() / Tbb.getCosTh());
        dqb.dx = Tbb.getX();
        dqb.dy = Tbb.getY();
        const Twist2D dq = Transform2D(th)(dqb);
        th += dq.dth;
        x += dq.dx;
        y += dq.dy;
        thL = thLnew;
        thR = thRnew;
        return *this;
    }

    DiffDrive& changeConfig(double dx, double dy)
    {
        x += dx;
        y += dy;
        return *this;
    }
};

} // namespace rigid2d
#endif
