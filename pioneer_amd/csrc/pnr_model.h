// pnr_model.h — compile-time model table of the Pioneer 6-DoF arm.
//
// The engine's own compact description of the robot (not a copy of the URDF
// file): the kinematic chain after merging fixed joints, per moving body the
// joint axis, the joint origin in the parent body frame, limits, and the
// lumped inertial data used by dynamics mode.  Source of the numbers:
// reference assets/pioneer_knm_6dof.urdf:27-275.
// Everything is constexpr so kernels unroll over it and fold the constants.
#pragma once

namespace pnr {

constexpr int kDof = 6;
constexpr int kObsDim = 137;
constexpr int kStatePlanes = 3;   // float4 planes per half-env record (2 x 12 = 24 words per env)
constexpr int kDynPlanes = 9;     // float4 planes of dynamics-mode extra state (36 words)

enum Axis : int { AX = 0, AY = 1, AZ = 2 };

struct JointDef {
    Axis axis;          // revolute axis in the child frame (== parent frame at q = 0)
    double ox, oy, oz;  // joint origin in the parent body frame
    double limit;       // symmetric position limit, rad (URDF text value)
};

// Moving bodies 1..6 (fixed joints merged into their parents):
//   body1 = rotator1 + hinge1          urdf:209-219   q1 about z
//   body2 = arm1                       urdf:221-227   q2 about y, origin (0,0,3)
//   body3 = arm2                       urdf:229-235   q3 about y, origin (0,0,11)
//   body4 = rotator2 + hinge2          urdf:237-248   q4 about x, origin (0,1,0)
//   body5 = arm3                       urdf:250-256   q5 about y, origin (11,0,0)
//   body6 = rotator3+effector+pointer  urdf:258-275   q6 about x, origin (0,0,0)
constexpr JointDef kJoints[kDof] = {
    {AZ, 0.0, 0.0, 0.0, 3.1416},
    {AY, 0.0, 0.0, 3.0, 1.309},
    {AY, 0.0, 0.0, 11.0, 1.309},
    {AX, 0.0, 1.0, 0.0, 3.1416},
    {AY, 11.0, 0.0, 0.0, 1.5708},
    {AX, 0.0, 0.0, 0.0, 3.1416},
};

// float32 joint limits as the env stores them (joint_limits(), pioneer_knm_env.py:217-220)
// and float32(cos/sin) of them — the 36 per-run constant observation entries
// (pioneer_knm_env.py:196-197).  r_lo = -r_hi exactly, so cos(r_lo) = cos(r_hi),
// sin(r_lo) = -sin(r_hi).  pnr_create re-derives these with libm and refuses
// to run on a mismatch.
__host__ __device__ constexpr float limit_hi(int j) { return (float)kJoints[j].limit; }
__host__ __device__ constexpr float limit_lo(int j) { return -(float)kJoints[j].limit; }
constexpr float kCos3_1416 = -0x1.000000p+0f, kSin3_1416 = -0x1.e5ddeap-18f;
constexpr float kCos1_309 = 0x1.090714p-2f, kSin1_309 = 0x1.ee8df0p-1f;
constexpr float kCos1_5708 = -0x1.e5ddeap-19f, kSin1_5708 = 0x1.000000p+0f;
constexpr float kLimitCos[kDof] = {kCos3_1416, kCos1_309, kCos1_309, kCos3_1416, kCos1_5708, kCos3_1416};
constexpr float kLimitSin[kDof] = {kSin3_1416, kSin1_309, kSin1_309, kSin3_1416, kSin1_5708, kSin3_1416};

// robot:pointer frame origin in body 6 (urdf:271-275)
constexpr double kTipX = 3.6, kTipY = 0.0, kTipZ = 1.9;

// URDF links lumped into each moving body, in URDF order (for per-link mass
// randomisation): index into the 11 non-world links
//   0 base(static) 1 rotator1 2 hinge1 3 arm1 4 arm2 5 rotator2 6 hinge2
//   7 arm3 8 rotator3 9 effector 10 pointer
constexpr int kNumLinks = 11;
constexpr int kLinkBody[kNumLinks] = {-1, 0, 0, 1, 2, 3, 3, 4, 5, 5, 5};
// every link: mass 1, inertia diag(1,1,1) about its own frame origin, COM at the
// origin (urdf inertial blocks, e.g. :29-34); the pointer link sits at kTip in body 6.
constexpr double kLinkMass = 1.0;
constexpr double kLinkInertia = 1.0;

// Contact sample spheres of dynamics mode.  The reference URDF has <visual> but no <collision> elements, so only the
// pointer's sphere (urdf:190-196) is the reference's own; the link samples are capsules fitted to the visual boxes
// (arm1 urdf:78-90, arm2 :92-104, rotator2 + hinge2 :106-132, arm3 :134-153, effector :169-188): n spheres of the
// given radius from a to b in the frame of moving body `body` (0-based).  radius < 0 = pointer_radius; the last
// sample of the last capsule IS the pointer sphere.  Spacing <= sphere diameter + the demo obstacle's width.
struct CapsuleDef { int body; float ax, ay, az, bx, by, bz; int n; float radius; };
constexpr int kNumCapsules = 5;
constexpr CapsuleDef kCapsules[kNumCapsules] = {
    {1, 0.f, 0.f, 0.f, 0.f, 0.f, 11.f, 8, 0.7f},
    {2, -1.f, 1.f, 0.f, 9.f, 1.f, 0.f, 7, 0.7f},
    {3, 9.f, 0.f, 0.f, 11.f, 0.f, 0.f, 2, 0.7f},
    {4, -0.5f, 0.f, 0.f, 2.5f, 0.f, 0.f, 3, 0.6f},
    {5, 3.6f, 0.f, -0.75f, 3.6f, 0.f, 1.9f, 3, -1.0f},
};

}  // namespace pnr
