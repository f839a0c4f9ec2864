// ref_shim.cpp -- extern "C" doorway onto the REFERENCE's own rigid2d / DiffDrive code.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it includes the reference's headers
// from where they lie (-I$(REF)/rigid2d/include) and is linked with the reference's rigid2d.cpp and
// diff_drive.cpp compiled from /root/reference by oracle/Makefile into oracle/_ref/librigid2d_ref.so
// (git-ignored; exists only where /root/reference exists or where that .so was shipped).
// It is used to validate the oracle's rigid2d/DiffDrive restatement and to generate tests/golden/rigid2d_ref.npz.
//
// The reference's EKF translation unit (nuslam/src/slam_library.cpp) is NOT built: it needs Armadillo,
// which this image does not have.
#include "rigid2d/rigid2d.hpp"
#include "rigid2d/diff_drive.hpp"

extern "C" {

double ref_normalize_angle(double rad) { return rigid2d::normalize_angle(rad); }

// T = {cos, sin, x, y}
void ref_integrate_twist(const double tw[3], double T_out[4])
{
    rigid2d::Twist2D t;
    t.dth = tw[0]; t.dx = tw[1]; t.dy = tw[2];
    rigid2d::Transform2D T = rigid2d::integrateTwist(t);
    T_out[0] = T.getCosTh(); T_out[1] = T.getSinTh(); T_out[2] = T.getX(); T_out[3] = T.getY();
}

// adjoint of T(trans=(x,y), rot=rad) applied to a twist
void ref_transform_twist(double x, double y, double rad, const double tw[3], double out[3])
{
    rigid2d::Vector2D v; v.x = x; v.y = y;
    rigid2d::Transform2D T(v, rad);
    rigid2d::Twist2D t; t.dth = tw[0]; t.dx = tw[1]; t.dy = tw[2];
    rigid2d::Twist2D r = T(t);
    out[0] = r.dth; out[1] = r.dx; out[2] = r.dy;
}

// dd = {base, rad, x, y, th, thL, thR}
static rigid2d::DiffDrive make(const double dd[7])
{ return rigid2d::DiffDrive(dd[0], dd[1], dd[2], dd[3], dd[4], dd[5], dd[6]); }

void ref_dd_convert_twist(const double dd[7], const double tw[3], double u_out[2])
{
    rigid2d::DiffDrive d = make(dd);
    rigid2d::Twist2D t; t.dth = tw[0]; t.dx = tw[1]; t.dy = tw[2];
    rigid2d::wheelVel u = d.convertTwist(t);
    u_out[0] = u.uL; u_out[1] = u.uR;
}

void ref_dd_get_twist(const double dd[7], double thL, double thR, double tw_out[3])
{
    rigid2d::DiffDrive d = make(dd);
    rigid2d::Twist2D t = d.getTwist(thL, thR);
    tw_out[0] = t.dth; tw_out[1] = t.dx; tw_out[2] = t.dy;
}

void ref_dd_step(double dd[7], double thL, double thR)
{
    rigid2d::DiffDrive d = make(dd);
    d(thL, thR);
    dd[2] = d.getX(); dd[3] = d.getY(); dd[4] = d.getTh(); dd[5] = d.getThL(); dd[6] = d.getThR();
}

// Transform2D algebra: T = {cos, sin, x, y} built through the reference's (trans, radians) constructor
static rigid2d::Transform2D tf(double x, double y, double rad)
{
    rigid2d::Vector2D v; v.x = x; v.y = y;
    return rigid2d::Transform2D(v, rad);
}
static void put(const rigid2d::Transform2D& T, double out[4])
{ out[0] = T.getCosTh(); out[1] = T.getSinTh(); out[2] = T.getX(); out[3] = T.getY(); }

void ref_tf_inv(double x, double y, double rad, double out[4]) { put(tf(x, y, rad).inv(), out); }
void ref_tf_mul(double x1, double y1, double rad1, double x2, double y2, double rad2, double out[4])
{ put(tf(x1, y1, rad1) * tf(x2, y2, rad2), out); }
void ref_tf_point(double x, double y, double rad, double px, double py, double out[2])
{
    rigid2d::Vector2D p; p.x = px; p.y = py;
    rigid2d::Vector2D q = tf(x, y, rad)(p);
    out[0] = q.x; out[1] = q.y;
}
// the algebra of broadcast_map2odom_tf (nuslam/src/slam.cpp:179-194) on the reference's own Transform2D
void ref_map_to_odom(const double odom[3], const double state[3], double out[3])
{
    rigid2d::Transform2D T_mo = tf(state[1], state[2], state[0]) * tf(odom[0], odom[1], odom[2]).inv();
    out[0] = T_mo.getX(); out[1] = T_mo.getY();
    out[2] = rigid2d::normalize_angle(asin(T_mo.getSinTh()));
}

}
