/*
 * pnr_dyn_oracle.h — float64 CPU statement of the DYNAMICS-MODE step
 * (Featherstone articulated-body algorithm + PD joint torques + joint limits +
 * pointer/ground penalty contact, 10 sub-steps at 1/240 s).
 *
 * TEST INFRASTRUCTURE ONLY (see pnr_oracle.h).  PARITY UNPINNED: the reference
 * never exercises dynamics (its World.step is a physical no-op)
 * and pins nothing here; this oracle is validated by physics (tests/test_dyn_oracle.py:
 * ABA == M^-1(tau - C) from an independent NumPy CRBA/RNEA, energy conservation,
 * teleport mode == kinematic mode) and then used as the checker of the HIP kernel.
 *
 * Model: the six moving bodies obtained by merging the URDF's fixed joints
 * (assets/pioneer_knm_6dof.urdf:27-275): every link has mass 1 and inertia
 * diag(1,1,1) about its own frame origin; per-link mass scales s[11] (domain
 * randomisation) multiply mass and inertia of link l.
 */
#ifndef PNR_DYN_ORACLE_H
#define PNR_DYN_ORACLE_H

#include <stdint.h>
#include "pnr_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_LINKS 11

/* a static body of the scene: World/Scene.create_body_plane / create_body_box / create_body_sphere with mass 0 and a
 * collision shape (bullet_scene.py:193-246); orientation is Bullet's quaternion (x, y, z, w) */
#define ORC_MAX_SCENE 8
enum { ORC_SHAPE_NONE = 0, ORC_SHAPE_PLANE = 1, ORC_SHAPE_BOX = 2, ORC_SHAPE_SPHERE = 3 };
typedef struct orc_scene_body {
    int32_t shape;
    int32_t pad;
    double position[3];
    double orientation[4];
    double size[3];           /* plane: planeNormal in the body frame; box: halfExtents; sphere: radius in size[0] */
} orc_scene_body;

typedef struct orc_dyn_params {
    double kp, kd;            /* PD gains on (r - q), (v - qd): Joint.control_position's positionGain/velocityGain role, bullet_scene.py:123-142 */
    double torque_limit;      /* `force` cap; <= 0 = unlimited */
    double gravity;           /* SimulationConfig.gravity, bullet_env.py:41 (setGravity(0,0,-g), bullet_scene.py:270) */
    double timestep;          /* 1/240 */
    int32_t frame_skip;       /* 10 sub-steps (World.step, bullet_scene.py:273-275) */
    int32_t teleport;         /* 1: q := r, qd := 0 before the sub-steps and no motor torque (the reference's resetJointState semantics) */
    int32_t randomize;        /* per-env draws at reset */
    int32_t pad;
    double joint_damping, joint_friction;     /* defaults when not randomised (URDF: 0) */
    double rand_mass_lo, rand_mass_hi;
    double rand_friction_lo, rand_friction_hi;
    double rand_damping_lo, rand_damping_hi;
    double ground_z;          /* NaN = no contact plane */
    double contact_kp, contact_kd;
    double obstacle_position[3], obstacle_half_extents[3];   /* half extent <= 0: no box */
    double pointer_radius;
    /* the rest of the reference's motor surface (bullet_scene.py:123-155) */
    int32_t control_mode;     /* 0: POSITION_CONTROL (Joint.control_position), 1: VELOCITY_CONTROL (Joint.control_velocity: the
                                 motor tracks the commanded velocity only, torque_limit is its `force`) */
    int32_t link_contacts;    /* 0: only the pointer sphere collides; 1: sample spheres along every moving link too
                                 (capsules fitted to the URDF's visual boxes, which carry no <collision> of their own) */
    double max_velocity;      /* control_position's maxVelocity: cap on the velocity the motor asks for; <= 0 = none */
    int32_t n_scene;          /* static scene bodies every contact sample sphere collides with (surface to surface) */
    int32_t pd_inertia_scaled;/* 1: kp / kd are per unit of each joint's articulated-body inertia D_i at the current pose (the motor
                                 asks for an ACCELERATION kp (r - q) + kd (v* - qd); torque = clip(D_i x that, +-torque_limit)), i.e.
                                 kp = omega^2 and kd = 2 zeta omega for every joint and pose alike; 0: plain torque gains */
    orc_scene_body scene[ORC_MAX_SCENE];
} orc_dyn_params;

#define ORC_CONTACT_SAMPLES 23
/* the contact sample spheres: body (0-based moving body), position in that body's frame, radius (< 0: pointer_radius) */
typedef struct orc_contact_sample { int32_t body; double c[3]; double radius; } orc_contact_sample;
int orc_dyn_contact_samples(const orc_dyn_params* d, orc_contact_sample* out);   /* returns the number active (1 or 23) */

typedef struct orc_dyn_state {
    double q[ORC_DOF], qd[ORC_DOF];
    double mass_scale[ORC_LINKS];
    double friction[ORC_DOF], damping[ORC_DOF];
} orc_dyn_state;

void orc_dyn_params_default(orc_dyn_params* d);

/* forward dynamics: qdd = ABA(q, qd, tau) with gravity g along -z and an optional
 * external force on the pointer (world frame, applied at the pointer origin) */
void orc_dyn_aba(const orc_dyn_state* s, const double tau[ORC_DOF], double gravity,
                 const double f_tip_world[3], double qdd[ORC_DOF]);

/* the same with an external spatial force on every body (body coordinates at the body origin, [n; f]) or NULL */
void orc_dyn_aba_ext(const orc_dyn_state* s, const double tau[ORC_DOF], double gravity,
                     const double fext[ORC_DOF][6], double qdd[ORC_DOF]);

/* the motor torque of one joint (PD / velocity servo with the maxVelocity and force caps), before damping and friction */
double orc_dyn_motor_torque(const orc_dyn_params* d, double r_ref, double v_ref, double q, double qd);
/* J_i: rotational inertia about joint i's axis of everything outboard of it at the zero pose, unscaled link masses
 * (= the mass-matrix diagonal there; a yardstick for gains, not used by the motor) */
void orc_dyn_nominal_inertia(double J[ORC_DOF]);
/* ABA with an acceleration-level motor term: + clip(D_i ades[i], +-tcap) on joint i (ades NULL: plain ABA) */
void orc_dyn_aba_motor(const orc_dyn_state* s, const double tau[ORC_DOF], const double ades[ORC_DOF], double tcap, double gravity,
                       const double fext[ORC_DOF][6], double qdd[ORC_DOF]);

/* penalty contact forces of all active sample spheres as per-body spatial forces; returns 1 if any contact is active */
int orc_dyn_contact_wrenches(const orc_dyn_params* d, const orc_dyn_state* s, double fext[ORC_DOF][6]);

/* kinetic energy and gravitational potential energy (for conservation tests) */
void orc_dyn_energy(const orc_dyn_state* s, double gravity, double* kinetic, double* potential);

/* penalty contact force on the pointer (world frame) from the ground plane and the box obstacle;
 * returns 1 if any contact is active */
int orc_dyn_contact_force(const orc_dyn_params* d, const double pos[3], const double vel[3], double f[3]);

/* pointer position and linear velocity in the world frame */
void orc_dyn_tip(const orc_dyn_state* s, double pos[3], double vel[3]);

/* one sub-step (semi-implicit Euler at d->timestep) tracking (r_ref, v_ref) */
/* a joint's motor as Joint.control_position / control_velocity set it (bullet_scene.py:123-155); enabled 0: the env-wide law */
typedef struct orc_joint_motor {
    int32_t enabled, control_mode;                    /* control_mode 0 position, 1 velocity */
    double target_position, target_velocity, position_gain, velocity_gain, max_force, max_velocity;
} orc_joint_motor;
void orc_dyn_world_step(const orc_dyn_params* d, const orc_params* p, const orc_state* ks, orc_dyn_state* s, const orc_joint_motor* m);

void orc_dyn_substep(const orc_dyn_params* d, const orc_params* p, orc_dyn_state* s,
                     const double r_ref[ORC_DOF], const double v_ref[ORC_DOF]);

/* reset of the dynamics state after orc_reset: q = r, qd = 0, parameter draws */
void orc_dyn_reset(const orc_dyn_params* d, const orc_params* p, const orc_state* ks, orc_dyn_state* s,
                   uint64_t global_env_id, uint32_t episode_drawn);

/* BulletEnv.step in dynamics mode: kinematic command integration, frame_skip sub-steps of
 * ABA + PD, reward/obs from the simulated q, qd; auto-reset as in orc_step */
void orc_dyn_step(const orc_dyn_params* d, const orc_params* p, orc_state* ks, orc_dyn_state* s,
                  uint64_t global_env_id, const float action[ORC_DOF], double obs[ORC_OBS],
                  double* reward, uint8_t* done, uint8_t* truncated, double info[4]);

void orc_dyn_reset_batch(const orc_dyn_params* d, const orc_params* p, orc_state* ks, orc_dyn_state* s,
                         int64_t n, int64_t env_id_offset, const uint8_t* mask, const double* joint_pos,
                         const double* target_pos, double* obs, int nthreads);
void orc_dyn_step_batch(const orc_dyn_params* d, const orc_params* p, orc_state* ks, orc_dyn_state* s,
                        int64_t n, int64_t env_id_offset, const float* actions, double* obs,
                        double* reward, uint8_t* done, uint8_t* truncated, double* info, int nthreads);

int orc_dyn_sizeof_params(void);
int orc_dyn_sizeof_state(void);

#ifdef __cplusplus
}
#endif
#endif
