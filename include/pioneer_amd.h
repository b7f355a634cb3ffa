/*
 * pioneer_amd.h — C ABI of the MI355X-native Pioneer-arm step/rollout engine.
 *
 * The reference (xdralex/pioneer) has no FFI layer: its boundary is the Python
 * gym.Env protocol of pioneer/envs/pioneer/pioneer_knm_env.py as consumed by
 * RLlib.  This header is the C-ABI a maintainer would bind in place of that
 * Python hot path; every entry point names the reference interface it
 * replaces (file:line, relative to the reference tree).
 *
 * Conventions
 *   - every function returns an int status (PNR_OK == 0, < 0 on error); no
 *     exceptions cross the boundary; pnr_last_error() gives the message;
 *   - all I/O buffers are caller-owned DEVICE pointers (plain pointers and
 *     sizes; no torch types); the library owns only the per-env state;
 *   - calls are asynchronous on the given HIP stream (passed as void*, i.e.
 *     a hipStream_t; NULL = the default stream) and never synchronise — with
 *     ONE exception: pnr_create zero-fills the state planes on the NULL stream
 *     and waits for that fill (hipStreamSynchronize(NULL)) before it returns,
 *     so that a first pnr_reset on ANY stream, blocking or not, finds them
 *     zero (pnr_destroy frees memory and synchronises as hipFree does);
 *   - a handle is not thread-safe; distinct handles are independent;
 *   - there is NO CPU backend: pnr_create fails with PNR_ERR_NODEVICE when no
 *     gfx950 device is usable.
 */
#ifndef PIONEER_AMD_H
#define PIONEER_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an exported signature or struct changes incompatibly; a caller checks pnr_abi_version() == the
 * PNR_ABI_VERSION it was compiled against before any other call (pnr_config carries it too).
 *   1  round 1
 *   2  pnr_ppo_loss gained `idx` (argument 2) and `means` (before `stream`); pnr_config grew (guarded by struct_size)
 *   3  pnr_mlp_step gained first_net / n_nets (the two nets as two chains) and w3_partials / w3_partial_floats (layer 3's
 *      weight-gradient partials made by the fused kernel); new entry points pnr_ppo_rollout (the sampler's T steps as one
 *      resident launch), pnr_filter_prepare, pnr_mlp_w3_partial_floats
 *   4  float32-accurate operands for the MLP kernels: pnr_mlp_pack / pnr_mlp_forward / pnr_mlp_act / pnr_mlp_gather gained `planes`
 *      (before `stream`), pnr_mlp_step gained `planes`; pnr_create waits for its zero fill (Conventions)
 *   5  new entry points pnr_world_step, pnr_set_joint_motor, pnr_build_fingerprint; `planes` == 2 now means two SCALED FP16 planes
 *      (was: two bf16 planes) in every pnr_mlp_* call; pnr_mlp_train_step checks every argument before its first launch and
 *      accepts g_head == NULL with w3_partials */
#define PNR_ABI_VERSION 5

#define PNR_DOF 6          /* revolute joints of pioneer_knm_6dof.urdf:209-264 */
#define PNR_OBS_DIM 137    /* pioneer_knm_env.py:194-211 (26 pieces)         */
#define PNR_STATE_WORDS 24 /* a[6] v[6] r[6] target[3] potential step episode */
#define PNR_INFO_DIM 4     /* r_pot, r_step, r_done, dist (numeric subset of   */
                           /* the info dict, pioneer_knm_env.py:167-179)      */
#define PNR_DYN_STATE_WORDS 36 /* dynamics mode: q[6] qd[6] + 24 params       */

enum pnr_status {
    PNR_OK = 0,
    PNR_ERR_INVALID = -1,     /* bad argument (AssertionError in the reference) */
    PNR_ERR_HIP = -2,         /* a HIP runtime call failed                      */
    PNR_ERR_NOMEM = -3,
    PNR_ERR_NODEVICE = -4,    /* no usable gfx950 device                        */
    PNR_ERR_UNSUPPORTED = -5
};

/* Memory layout of a [num_envs x F] batch. */
enum pnr_layout {
    PNR_ENV_MAJOR = 0,     /* row-major [num_envs][F] — what gym/RLlib see      */
    PNR_FEATURE_MAJOR = 1  /* [F][num_envs] — feature planes (first Linear as W . X^T)      */
};

/* The reference's motor forms (Joint.control_position / control_velocity, bullet_scene.py:123-155). */
enum pnr_control {
    PNR_CONTROL_POSITION = 0, /* POSITION_CONTROL: positionGain pd_kp, velocityGain pd_kd, force torque_limit, maxVelocity */
    PNR_CONTROL_VELOCITY = 1  /* VELOCITY_CONTROL: the motor tracks the commanded velocity only (gain pd_kd, force torque_limit) */
};

enum pnr_mode {
    PNR_MODE_KINEMATIC = 0, /* the reference's live semantics ("parity mode")   */
    PNR_MODE_DYNAMIC = 1    /* ABA forward dynamics + PD torque tracking         */
};

/*
 * A static body of the scene: what World/Scene.create_body_plane / create_body_box / create_body_sphere make with mass 0
 * and a collision shape (bullet_scene.py:193-246).  Dynamics mode only: every contact sample sphere of the arm (the
 * pointer; with link_contacts the 22 link samples too) collides with it, surface to surface, by the penalty law of
 * contact_kp / contact_kd.  orientation is Bullet's quaternion (x, y, z, w).
 */
#define PNR_MAX_SCENE 8
enum pnr_shape { PNR_SHAPE_NONE = 0, PNR_SHAPE_PLANE = 1, PNR_SHAPE_BOX = 2, PNR_SHAPE_SPHERE = 3 };
typedef struct pnr_scene_body {
    int32_t shape;                /* enum pnr_shape */
    int32_t reserved;
    double position[3];           /* basePosition */
    double orientation[4];        /* baseOrientation (x, y, z, w) */
    double size[3];               /* plane: planeNormal in the body frame; box: halfExtents; sphere: radius in size[0] */
} pnr_scene_body;

/*
 * Tunables.  Field names and defaults follow PioneerKinematicConfig
 * (pioneer_knm_env.py:19-34) and SimulationConfig (bullet_env.py:36-44);
 * max_episode_steps is gym.wrappers.TimeLimit's argument
 * (pioneer/launch/pioneer_knm_train.py:27).  Fill with pnr_config_default().
 */
typedef struct pnr_config {
    uint32_t struct_size; /* sizeof(pnr_config), checked by pnr_create */
    uint32_t abi_version; /* PNR_ABI_VERSION */

    /* PioneerKinematicConfig */
    double max_v_to_r;            /* 2       */
    double max_a_to_v;            /* 10      */
    double done_distance;         /* 0.1     */
    double award_max;             /* 100.0   */
    double award_done;            /* 5.0     */
    double award_potential_slope; /* 10.0    */
    double penalty_step;          /* 1/100   */
    double target_lo[3];          /* (15,-10,2) */
    double target_hi[3];          /* (25, 10,6) */
    double target_radius;         /* 0.2 (visual only in the reference) */

    /* SimulationConfig */
    double timestep;              /* 1/240 */
    int32_t frame_skip;           /* 10    */
    double gravity;               /* 0     */

    /* TimeLimit; 0 disables truncation */
    int32_t max_episode_steps;    /* 500   */

    /* engine options (no reference counterpart) */
    int32_t auto_reset;           /* 1: done|truncated envs are re-drawn in-kernel */
    int32_t obs_layout;           /* enum pnr_layout */
    int32_t action_layout;        /* enum pnr_layout */
    int32_t mode;                 /* enum pnr_mode */

    /* dynamics mode only (PD surface of bullet_scene.py:123-155; unpinned) */
    double pd_kp;                 /* position gain  [torque/rad]         */
    double pd_kd;                 /* velocity gain  [torque/(rad/s)]     */
    double torque_limit;          /* |tau| cap; <= 0 = unlimited         */
    double joint_damping;         /* viscous, URDF default 0             */
    double joint_friction;        /* Coulomb (smoothed), URDF default 0  */
    int32_t teleport;             /* 1: reference semantics — q:=r, qd:=0 before the sub-steps */
    int32_t randomize;            /* 1: per-env link-mass / friction / damping draws at reset  */
    double rand_mass_lo, rand_mass_hi;         /* scale on every link mass, U(lo,hi) */
    double rand_friction_lo, rand_friction_hi; /* per-joint Coulomb friction         */
    double rand_damping_lo, rand_damping_hi;   /* per-joint viscous damping          */
    double ground_z;              /* contact plane height for the pointer; NaN = no plane */
    double contact_kp, contact_kd;/* penalty contact stiffness / damping  */
    /* static box obstacle for the pointer sphere (the reference demo's create_body_box,
     * pioneer_knm_env.py:249-255: half extents (0.5,0.5,5) at (10,5,0)); half extent <= 0 = none */
    double obstacle_position[3];
    double obstacle_half_extents[3];
    double pointer_radius;        /* 0.2: the pointer's sphere (urdf:190-196) */
    int32_t control_mode;         /* enum pnr_control; dynamics mode, teleport 0 */
    int32_t link_contacts;        /* 1: sample spheres along every moving link collide with the plane / box too (capsules
                                   * fitted to the URDF's visual boxes; the URDF itself has no <collision>) */
    double max_velocity;          /* control_position's maxVelocity (bullet_scene.py:126,136): cap on the velocity the
                                   * motor asks for, rad/s; <= 0 = none */
    int32_t n_scene;              /* static scene bodies in use, 0 .. PNR_MAX_SCENE */
    int32_t pd_inertia_scaled;    /* 1: the motor asks for an ACCELERATION pd_kp (r - q) + pd_kd (v* - qd) and applies the torque
                                   * clip(D_i x that, +-torque_limit), D_i being joint i's articulated-body inertia at the current
                                   * pose (the ABA forms it anyway): pd_kp = omega^2 and pd_kd = 2 zeta omega then hold for every
                                   * joint and pose alike, and the explicit motor is stable whenever pd_kd x timestep < 2 — on the
                                   * light wrist (inertia 6.6) as on the shoulder (1477).  Bullet's own motors are implicit
                                   * constraints and need no such care.  0: plain torque gains, the same for all joints */
    pnr_scene_body scene[PNR_MAX_SCENE];
} pnr_config;

typedef struct pnr_env_s* pnr_handle;

/* Derived per-joint constants of PioneerKinematicEnv.__init__
 * (pioneer_knm_env.py:56-61, :72, :217-220). */
typedef struct pnr_constants {
    float r_lo[PNR_DOF], r_hi[PNR_DOF]; /* joint_limits(), float32 */
    float v_max[PNR_DOF];               /* max_v_to_r * (r_hi - r_lo) */
    float a_max[PNR_DOF];               /* max_a_to_v * v_max == action_space bound */
    double dt;                          /* world.step_time = timestep * frame_skip */
    double eps;                         /* 1e-5 */
} pnr_constants;

/* PioneerKinematicConfig() / SimulationConfig() defaults. */
int pnr_config_default(pnr_config* cfg);

/* Constants derived from a config without creating a device handle. */
int pnr_get_constants(const pnr_config* cfg, pnr_constants* out);

/*
 * Replaces PioneerKinematicEnv.__init__ (pioneer_knm_env.py:39-74) +
 * BulletEnv.__init__/reset_simulator/load_scene (bullet_env.py:66-148) for
 * num_envs independent envs on HIP device `device_id`.  `env_id_offset` is the
 * global index of local env 0: the reset RNG is keyed by (seed, global env
 * id, episode#), so trajectories do not depend on how a batch is sharded
 * across GPUs.  The envs hold no valid state until the first full pnr_reset (mask == NULL) or
 * pnr_set_state: pnr_step / pnr_rollout / pnr_observe before that fail with PNR_ERR_INVALID
 * (the reference's constructor calls reset_world() itself, pioneer_knm_env.py:69; the Python
 * façade PioneerKinematicEnv does the same).
 */
int pnr_create(const pnr_config* cfg, int64_t num_envs, int64_t env_id_offset,
               int device_id, uint64_t seed, pnr_handle* out);

int pnr_destroy(pnr_handle h);

/* seed(): pioneer_knm_env.py:107-109.  Takes effect at the next reset. */
int pnr_seed(pnr_handle h, uint64_t seed);

/*
 * Replaces BulletEnv.reset (bullet_env.py:187-190) + reset_world
 * (pioneer_knm_env.py:76-105).
 *   mask        [num_envs] bytes, non-zero = reset this env; NULL = all
 *   joint_pos   [num_envs][6] env-major float32 or NULL — the reference's
 *               `joint_positions` override; NULL draws r ~ U(r_lo, r_hi)
 *   target_pos  [num_envs][3] env-major float32 or NULL — `target_position`
 *               override; NULL draws target ~ U(target_lo, target_hi)
 *   obs_out     obs batch in cfg.obs_layout, or NULL; rows of envs that are
 *               not reset are left untouched
 */
int pnr_reset(pnr_handle h, const uint8_t* mask, const float* joint_pos,
              const float* target_pos, float* obs_out, void* stream);

/*
 * Replaces BulletEnv.step (bullet_env.py:192-197) = act
 * (pioneer_knm_env.py:111-182) + observe (:184-211) + TimeLimit.step for every
 * env of the batch, in one kernel launch.
 *   actions    [num_envs x 6] float32 in cfg.action_layout
 *   obs        [num_envs x 137] float32 in cfg.obs_layout
 *   reward     [num_envs] float32
 *   done       [num_envs] bytes — the env's own `done` (distance < done_distance)
 *   truncated  [num_envs] bytes — TimeLimit.truncated (elapsed >= max && !done); may be NULL
 *   info       [num_envs][4] float32 (r_pot, r_step, r_done, dist) or NULL
 */
int pnr_step(pnr_handle h, const float* actions, float* obs, float* reward,
             uint8_t* done, uint8_t* truncated, float* info, void* stream);

/*
 * T consecutive steps in ONE launch with open-loop actions (state stays in
 * registers between steps).  Buffers are the pnr_step ones with a leading
 * [T] axis: actions [T][N x 6], obs [T][N x 137], reward/done/truncated [T][N].
 * Same results as T calls of pnr_step: bit for bit in kinematic mode; in dynamics mode
 * the simulated quantities agree to float32 rounding (two kernels, contraction allowed),
 * command state, counters, targets and per-env draws bit for bit.
 */
int pnr_rollout(pnr_handle h, int32_t T, const float* actions, float* obs,
                float* reward, uint8_t* done, uint8_t* truncated, void* stream);

/* observe() without stepping (pioneer_knm_env.py:184-211). */
int pnr_observe(pnr_handle h, float* obs_out, void* stream);

/*
 * World.step() (bullet_scene.py:273-275: frame_skip x stepSimulation) on its own: the simulator advances by step_time and
 * NOTHING else happens — no command integration, reward, TimeLimit, reset or observation.  What the reference's demo loop
 * drives (pioneer_knm_env.py:277-296: reset_state(position(), velocity); world.step()).
 *   dynamics-mode handle (joint_state must be NULL): frame_skip articulated-body sub-steps on the handle's simulated joints
 *     (q, qd of pnr_get_dyn_state) under gravity, contacts, joint limits, friction / damping and each joint's motor — the one
 *     pnr_set_joint_motor gave it, else the handle's own motor law tracking the env's command state r, v (teleport handles:
 *     no motor: the joints coast).  Parity unpinned, like all of dynamics mode.
 *   kinematic-mode handle: there is no simulated state; joint_state [num_envs][12] (q[6] | qd[6], float32, caller-owned
 *     device memory) is the joints as Bullet holds them after resetJointState(position, velocity), and with the reference's
 *     defaults (no gravity, no motor torque, no collision shapes) a step carries each on at its velocity: q += qd * step_time,
 *     stopped at its limit with the velocity zeroed.  The env's own state is not touched (the reference's self.r / self.v
 *     are not either).
 */
int pnr_world_step(pnr_handle h, float* joint_state, void* stream);

/*
 * Joint.control_position / Joint.control_velocity (bullet_scene.py:123-155 -> setJointMotorControl2) for joint `joint`
 * (0..5, URDF order) of every env of a dynamics-mode handle: control_mode PNR_CONTROL_POSITION (targetPosition,
 * targetVelocity, positionGain, velocityGain, force, maxVelocity) or PNR_CONTROL_VELOCITY (targetVelocity, force).  An
 * optional argument the caller leaves out is passed as NaN and takes the handle's configured value (pd_kp, pd_kd,
 * torque_limit, max_velocity; targetVelocity: 0).  The motor is the engine's one law,
 *   tau = clip(Kp (r* - q) + kd (clamp(v* + c (r* - q), +-maxVelocity) - qd), +-force)
 * (pnr_config.control_mode's folding, per joint), honoured by pnr_world_step; pnr_step keeps driving every joint from the
 * env's own command state (the reference's act() teleports the joints each step, pioneer_knm_env.py:148).  Host-side and
 * synchronous: takes effect with the next pnr_world_step.  Parity unpinned (Bullet's motor is a velocity-level constraint).
 */
int pnr_set_joint_motor(pnr_handle h, int32_t joint, int32_t control_mode, double target_position, double target_velocity,
                        double position_gain, double velocity_gain, double max_force, double max_velocity);

/*
 * Raw state for checkpoint / tests, as planar 32-bit words [24][num_envs]:
 * words 0-5 a, 6-11 v, 12-17 r, 18-20 target, 21 potential (float32),
 * 22 step_index, 23 episode (uint32).  Device pointers.
 */
int pnr_get_state(pnr_handle h, uint32_t* words_out, void* stream);
int pnr_set_state(pnr_handle h, const uint32_t* words_in, void* stream);

/* Dynamics-mode extra state, planar float32 [36][num_envs]: q[6], qd[6],
 * link-mass scale[11], friction[6], damping[6], 1 pad.  PNR_ERR_UNSUPPORTED in
 * kinematic mode. */
int pnr_get_dyn_state(pnr_handle h, float* words_out, void* stream);
int pnr_set_dyn_state(pnr_handle h, const float* words_in, void* stream);

/* Diagnostic: the engine's float32 sin/cos (the np.sin/np.cos replacement used
 * for obs entries, pioneer_knm_env.py:195-203) over a device array x[n].
 * bounded != 0 selects the Cody-Waite path used for r, v and limit distances;
 * 0 the path used for raw actions (falls back to full reduction above 2^17). */
int pnr_diag_sincos(const float* x, float* sin_out, float* cos_out, int64_t n,
                    int bounded, void* stream);

/* Host-driver helper (not part of the env surface): the element-wise part of the PPO loss with the
 * hyper-parameters of pioneer/launch/pioneer_knm_train.py:45-67 (clip_param, vf_clip_param,
 * vf_loss_coeff; kl_coeff and entropy_coeff as device scalars because they change between captured
 * replays), forward and backward in one launch.  head_policy / head_value are the two nets' raw
 * outputs as rows of 16 floats (means 0..5, log-stds 6..11 | value 0); the gradients of the batch-mean
 * loss come back in the same layout; partial_sums [partial_rows][8] receives per-block sums of
 * (policy_loss, vf_loss, kl, entropy, total) with partial_rows >= ceil(batch / 256).  idx (int64 [batch], or NULL) is
 * the minibatch gather: sample i's rollout record (actions .. value_old) is row idx[i] of those arrays, while the
 * head rows and their gradients are indexed by i.  means (8 floats, or NULL) receives the five batch means, summed from
 * partial_sums in row order by a second small launch.  Device pointers. */
int pnr_ppo_loss(int64_t batch, const int64_t* idx, const float* head_policy, const float* head_value, const float* actions,
                 const float* logp_old, const float* mean_old, const float* log_std_old, const float* adv,
                 const float* value_target, const float* value_old, const float* kl_coeff,
                 const float* entropy_coeff, float clip_param, float vf_clip_param, float vf_loss_coeff,
                 float* grad_head_policy, float* grad_head_value, float* partial_sums, int64_t partial_rows,
                 float* means, void* stream);

/*
 * Host-driver helper: after the sampler's T steps, one launch for the log-probabilities of the taken actions (diagonal
 * Gaussian, actions / mean / log_std [T][n][6]; pass actions NULL to skip) and GAE(lambda) advantages and value targets
 * [T][n] from reward / values [T][n], last_value [n] (the bootstrap value) and the done / truncated bytes of pnr_step
 * (terminal = done | truncated, as RLlib 0.8's postprocessing treats the TimeLimit cut; truncated may be NULL).
 * terminals [T][n] (1.0 / 0.0) is optional.  gamma / lambda: the reference leaves RLlib's defaults (0.99 / 1.0).
 * stats (optional): the rollout's bookkeeping in the same launch — the episode statistics behind the reference's result
 * columns episode_reward_{max,min,mean} / episode_len_mean (cli.py:32-38): ep_ret / ep_len [n] are each env's running return
 * and length (in / out, carried across rollouts), w_sum / w_len / w_cnt (float64) and w_max / w_min (float32) the window
 * accumulators of the episodes that ended, updated in place; adv_stats [3] (float64) receives the advantages' sum, sum of
 * squares and count (PPO standardises them over the global batch).  scratch: pnr_ppo_gae_scratch(n) doubles.
 */
typedef struct pnr_rollout_stats {
    float* ep_ret; float* ep_len;
    double* scratch; int64_t scratch_doubles;
    double* w_sum; double* w_len; double* w_cnt; float* w_max; float* w_min;
    double* adv_stats;
} pnr_rollout_stats;
int64_t pnr_ppo_gae_scratch(int64_t n);
int pnr_ppo_gae(int32_t T, int64_t n, const float* reward, const float* values, const float* last_value, const uint8_t* done,
                const uint8_t* truncated, const float* actions, const float* mean, const float* log_std, double gamma,
                double lambda, float* logp, float* adv, float* value_target, float* terminals, const pnr_rollout_stats* stats,
                void* stream);

/*
 * Host-driver helper: the moment pass of the observation filter ('observation_filter': 'ConcurrentMeanStdFilter',
 * pioneer_knm_train.py:66) over obs [rows][137]: dsum[c] += sum_r (obs[r][c] - pivot[c]), dsq[c] += sum_r (...)^2 (float64
 * accumulators [137]; float32 partial sums over 512 rows each, added in order), *dn += rows; one read of the buffer.
 * scratch: pnr_filter_moments_scratch(rows) floats.
 */
int64_t pnr_filter_moments_scratch(int64_t rows);
int pnr_filter_moments(int64_t rows, const float* obs, const float* pivot, float* scratch, int64_t scratch_floats, double* dsum,
                       double* dsq, double* dn, void* stream);

/* The filter's merge on one rank (MeanStdFilter.sync): the pending delta *dn, dsum / dsq [137] (zeroed on return) about
 * pivot [137] into the running *n, mean / m2 [137] by Chan's update, float64, one launch. */
int pnr_filter_merge(double* dn, double* dsum, double* dsq, const float* pivot, double* n, double* mean, double* m2, void* stream);

/* MeanStdFilter.prepare() in one launch ('observation_filter': 'ConcurrentMeanStdFilter', pioneer_knm_train.py:66): the float32 vectors
 * the kernels filter with, x' = clamp((x - loc) * inv, lo, hi), from the
 * running statistics *n, mean / m2 [137] (RLlib MeanStdFilter: (x - mean) / (std + 1e-8), clipped to +-clip; clip = +inf: no clipping;
 * the identity until two samples exist). */
int pnr_filter_prepare(const double* n, const double* mean, const double* m2, double clip, float* loc, float* inv, float* lo, float* hi,
                       void* stream);

/*
 * Host-driver helper: out[0..n) = a pseudo-random permutation of 0..n-1 keyed by (seed, stream_id) — a Feistel network
 * with cycle walking, one launch and no sort; the SGD epochs' minibatch shuffle (RLlib sgd.py shuffles each epoch).
 */
int pnr_permutation(int64_t n, uint64_t seed, uint64_t stream_id, int64_t* out, void* stream);

/*
 * Host-driver helpers (not part of the env surface): the two MLPs of the reference's PPO config — 'fcnet_hiddens':
 * [256, 256] (pioneer/launch/pioneer_knm_train.py:59-61), tanh, separate policy (12 outputs: 6 means + 6 log-stds) and
 * value (1 output) nets, RLlib's FullyConnectedNetwork with vf_share_layers False — as bf16 MFMA kernels.
 *   params / grads  12 device pointers, net-major: policy w1 [256][137], b1 [256], w2 [256][256], b2 [256],
 *                   w3 [n3][256], b3 [n3], then the value net's six; float32, row-major [out][in] as the host keeps them
 *   wpack           pnr_mlp_pack_elems() bf16 values; bias: pnr_mlp_bias_elems() floats (written by pnr_mlp_pack)
 *   obs             [rows][137] float32; idx [batch] int64 row gather or NULL; f_loc/f_inv/f_lo/f_hi [137] or all NULL:
 *                   the nets see clamp((obs - loc) * inv, lo, hi), the MeanStdFilter of the reference's config (:66)
 *   head            [2][batch][16] float32: policy rows = means 0..5, raw log-stds 6..11; value rows = v at column 0
 *   xs [batch][144], h1 / h2 / dz1 / dz2 [2][batch][256] bf16: activations kept for / made by the backward pass
 *   slabs           >= pnr_mlp_slab_floats(batch) floats of scratch (per-slice partial gradients, summed in order)
 * pnr_mlp_backward = backward-data + weight gradients + reduction; gradients of the batch as given by g_head
 * [2][batch][16] (d loss / d head), times the device scalar *scale when scale != NULL, written (accumulate = 0) or
 * added (1) to `grads`.
 */
int64_t pnr_mlp_pack_elems(void);
int64_t pnr_mlp_bias_elems(void);
int64_t pnr_mlp_slab_floats(int64_t batch);
/*
 * `planes` (1, 2 or 3) selects the precision of every MFMA operand of the pnr_mlp_* kernels; accumulation is float32 always.
 *   1: bf16 operands (8 significant bits) — the reduced-precision fast variant.
 *   2: each float32 operand as TWO FP16 PLANES (ABI 5): p0 = fp16(s x), p1 = fp16(s x - p0) with a power-of-two scale s per tensor
 *      (weights 2^8, activations 2^8, net inputs 2^4, gradients 4 * 2^ceil(log2 batch); the accumulators are divided by the product
 *      of the two scales, exactly) — 22 significant bits per operand, the three fp16 MFMAs (0,0), (0,1), (1,0) per product.  Measured
 *      at a float32 framework GEMM's own distance from float64 (heads 2.9e-7 against 3.8e-7; gradients 9.6e-7 against 2.4e-6 of float32
 *      autograd): the accuracy of a float32 GEMM, i.e. what the reference's float32 learner computes in (pioneer_knm_train.py:47).
 *      Range: scaled values are clamped to +-65504 — weights |w| < 255, inputs |x| < 4094, per-sample gradients |dL/dz| < 16 384.
 *   3: three bf16 planes (24 bits, exact split, no range caveat), the six bf16 MFMAs of the plane pairs i + j < 3 per product (ABI 4).
 * Every 16-bit operand buffer then exists `planes` times, plane-major: wpack [planes][pnr_mlp_pack_elems()], xs / xs_out / xs_in
 * [planes][batch][144], h1 / h2 / dz1 / dz2 [planes][2][batch][256].  Biases, heads, slabs, Adam state and master weights are float32 either way.
 * planes > 1: pnr_mlp_forward / pnr_mlp_act save no activations (xs, h1, h2 / xs_out must be NULL: pnr_mlp_backward is bf16-only and
 * the learner gathers its inputs from the float32 observations), pnr_mlp_gather takes no xs_rows, pnr_mlp_train_step needs w3_partials.
 */
int pnr_mlp_pack(const float* const* params, int32_t n3_policy, int32_t n3_value, void* wpack, float* bias, int32_t planes, void* stream);
int pnr_mlp_forward(int64_t batch, const float* obs, const int64_t* idx, const float* f_loc, const float* f_inv,
                    const float* f_lo, const float* f_hi, const void* wpack, const float* bias, float* head,
                    void* xs, void* h1, void* h2, int32_t first_net, int32_t n_nets, int32_t planes, void* stream);
/*
 * The sampler's per-step launch: both nets forward on `obs` [batch][137] and, in the policy net's last epilogue, the
 * action draw of RLlib's DiagGaussian (what the reference's PPO config samples with): log_std = clamp(raw, -20, 2),
 * actions = mean + exp(log_std) * noise (noise [batch][6] standard-normal draws supplied by the caller), env_actions =
 * clamp(actions, -a_max, a_max) (a_max [6]; the env's action space, pioneer_knm_env.py:60-61; NULL: no clipping and
 * env_actions is not written).  mean / log_std / actions / env_actions [batch][6], values [batch]; head [2][batch][16] or NULL;
 * xs_out [batch][144] bf16 or NULL: the nets' input as they saw it (filtered, rounded), which the learner's epoch gather
 * (pnr_mlp_gather's xs_rows) copies instead of re-making it from the float32 observations.
 */
int pnr_mlp_act(int64_t batch, const float* obs, const float* f_loc, const float* f_inv, const float* f_lo, const float* f_hi,
                const void* wpack, const float* bias, const float* noise, const float* a_max, float* head, float* mean,
                float* log_std, float* values, float* actions, float* env_actions, void* xs_out, int32_t planes, void* stream);
/*
 * The sampler's closed loop as ONE resident launch: T x (pnr_mlp_act, pnr_step) for every env of handle `h` — per step both nets
 * on the observation in slot t of `obs`, the DiagGaussian draw and clip as in pnr_mlp_act, then BulletEnv.step
 * (bullet_env.py:192-197) with that action: reward / done / truncated [T][n] and the next observation into slot t + 1.  A
 * workgroup owns 64 envs for all T steps (env state, observation tile and both nets' W2 stay on the CU).  Replaces the RLlib
 * rollout worker's act -> env.step loop (the reference: one env per worker process, pioneer_knm_train.py:49) for kinematic-mode
 * handles with env-major layouts; results equal the per-step calls bit for bit.
 *   obs [T + 1][n][137]: slot 0 = where the rollout starts (what the last pnr_step / pnr_reset / rollout left), slots 1..T written
 *   noise [T][n][6] standard-normal draws; a_max [6] or NULL (no clipping); f_loc..f_hi [137] each or all NULL
 *   mean / log_std / actions [T][n][6] (8-byte aligned), values [T][n], xs_out [T][n][144] bf16 or NULL
 */
int pnr_ppo_rollout(pnr_handle h, int32_t T, const float* f_loc, const float* f_inv, const float* f_lo, const float* f_hi,
                    const void* wpack, const float* bias, const float* noise, const float* a_max, float* obs, float* mean,
                    float* log_std, float* values, float* actions, void* xs_out, float* reward, uint8_t* done, uint8_t* truncated,
                    void* stream);
int pnr_mlp_backward(int64_t batch, const float* g_head, const void* wpack, const void* xs, const void* h1, const void* h2,
                     void* dz1, void* dz2, float* slabs, int64_t slab_floats, float* const* grads, int32_t n3_policy,
                     int32_t n3_value, int32_t accumulate, const float* scale, void* stream);

/*
 * One PPO minibatch update of both nets in three launches, nothing of it on the host: (1) per 64-sample tile and net the
 * forward pass (activations saved for the weight gradients), the tile's share of the loss and its backward-data pass in
 * ONE kernel (which also counts the update in *adam_step); (2) weight gradients per batch slice; (3) EITHER the fused
 * slab-reduction + Adam + bf16 repacking, one extra block of which sums the loss means (flat_grad == NULL), OR the
 * loss means in a small launch of their own and the reduction into flat_grad
 * [pnr_mlp_grad_floats()] with no update: a multi-GPU run all-reduces that bucket and calls pnr_mlp_adam(s, flat_grad,
 * 1 / world_size, stream).  The update is Adam with the arithmetic of the optimiser the reference trains with ('lr' of its
 * config, pioneer_knm_train.py:64; betas 0.9 / 0.999, eps 1e-8; no weight decay) with its state m, v kept as
 * [pnr_mlp_grad_floats()] floats each.  wpack / bias must hold the CURRENT weights on entry (pnr_mlp_pack once, then
 * every update refreshes them).  All pointers are device pointers; `means` receives (policy_loss, vf_loss, kl,
 * entropy, total, 0, 0, 0).  partials: scratch of partial_rows >= 2 * ceil(batch / 64) rows of 8 floats; `head` is not
 * written (the head rows never leave the chip); g_head [2][batch][16] is written ONLY when w3_partials is NULL (with
 * w3_partials nothing outside the tile reads the head gradients, they stay on the chip, g_head is left as it was and may be
 * NULL).  wpack, bias, adam_m, adam_v, slabs and flat_grad must be 16-byte aligned (PNR_ERR_INVALID otherwise: the optimiser
 * kernel moves them four floats at a time); so must xs_in, and with planes > 1 xs_in_plane must be 0 or a multiple of 8
 * elements >= batch * 144 (the kernels read 16-byte vectors at xs_in + plane * xs_in_plane).  Every argument is checked
 * before the first launch: an error return has launched nothing and counted no update.
 */
typedef struct pnr_mlp_step {
    uint32_t struct_size;   /* sizeof(pnr_mlp_step) */
    int64_t batch;
    const float* obs; const int64_t* idx;
    const float* f_loc; const float* f_inv; const float* f_lo; const float* f_hi;
    const float* actions; const float* logp_old; const float* mean_old; const float* log_std_old;
    const float* adv; const float* value_target; const float* value_old;
    const float* kl_coeff; const float* entropy_coeff;
    float clip_param, vf_clip_param, vf_loss_coeff;
    float* params[12]; int32_t n3_policy, n3_value;
    void* wpack; float* bias;
    float* adam_m; float* adam_v; float* adam_step;
    float lr, beta1, beta2, eps;
    float* head; float* g_head; void* xs; void* h1; void* h2; void* dz1; void* dz2;
    float* partials; int64_t partial_rows; float* slabs; int64_t slab_floats;
    float* means;
    float* flat_grad;
    const void* xs_in;      /* optional [batch][144] bf16: the nets' input already gathered, filtered and rounded by
                             * pnr_mlp_gather.  Then obs / idx / the filter vectors are not read, the record arrays (actions ..
                             * value_old) are read row by row, and `xs` is not written */
    int32_t first_net, n_nets; /* the nets this call works on: 0, 0 (or 0, 2) = both; 0, 1 = the policy net; 1, 1 = the value net.
                             * The two nets share nothing but their input (vf_share_layers False), so a multi-GPU run may drive them
                             * as two independent chains on two streams — each train_step -> all-reduce of ITS half of flat_grad
                             * -> pnr_mlp_adam — and one net's all-reduce overlaps the other's kernels.  Every call counts one update
                             * in *adam_step: two chains need two counters.  `means` of a one-net call holds that net's terms only
                             * (policy: policy_loss, kl, entropy and their share of total; value: vf_loss and its share): the
                             * update's means are the element-wise sum of the two rows.  partial_rows >= n_nets * ceil(batch / 64) */
    float* w3_partials;     /* optional scratch, w3_partial_floats >= pnr_mlp_w3_partial_floats(batch): the fused kernel then leaves layer 3's */
    int64_t w3_partial_floats; /* weight-gradient products per 64-sample tile there (17 KB) instead of storing H2 (32 KB per tile) for the
                             * weight-gradient kernel, which adds them in tile order: the same sums bit for bit, 30 % fewer bytes in that
                             * kernel.  NULL: H2 is stored to h2 and read back (h2 must be given either way) */
    int32_t planes;         /* 0 or 1: bf16 operands; 2, 3: split float32 operands (see pnr_mlp_pack): wpack, xs_in / xs, h1, dz1, dz2 hold
                             * `planes` planes, w3_partials must be given, Adam refreshes every plane of wpack */
    int64_t xs_in_plane;    /* planes > 1: elements between two planes of xs_in; 0 = batch * 144.  (A minibatch inside an epoch's
                             * gathered planes [planes][rows][144]: xs_in = plane 0's first row of it, xs_in_plane = rows * 144) */
} pnr_mlp_step;
int64_t pnr_mlp_w3_partial_floats(int64_t batch);
/*
 * An SGD epoch's shuffle applied once: row i of every output is row idx[i] of the corresponding input — the observation
 * filtered and rounded to the nets' input layout (xs_out [batch][144] bf16) and the rollout record — so that the epoch's
 * minibatch updates read contiguous rows (pnr_mlp_step.xs_in = xs_out + 144 * first_row, record pointers likewise).
 * record_rows (optional): the same record as ONE row of 24 floats per sample, made once per iteration by
 * pnr_ppo_pack_record (actions 0..5 | mean 6..11 | log_std 12..17 | logp, adv, value_target, value_old | 2 pad; adv
 * standardised as (adv - *adv_mu) / *adv_den when the two device scalars are given): the gather then reads one or two cache
 * lines per sample for the record instead of seven, and the seven input arrays may be NULL.  xs_rows (optional, [rows][144]
 * bf16): the net inputs saved by pnr_mlp_act; then obs and the filter vectors are not read (288-byte rows instead of 548).
 */
int pnr_ppo_pack_record(int64_t rows, const float* actions, const float* logp_old, const float* mean_old, const float* log_std_old,
                        const float* adv, const float* value_target, const float* value_old, const float* adv_mu, const float* adv_den,
                        float* record_rows, void* stream);
int pnr_mlp_gather(int64_t batch, const int64_t* idx, const float* obs, const float* f_loc, const float* f_inv, const float* f_lo,
                   const float* f_hi, const float* actions, const float* logp_old, const float* mean_old, const float* log_std_old,
                   const float* adv, const float* value_target, const float* value_old, void* xs_out, float* actions_out,
                   float* logp_out, float* mean_out, float* log_std_out, float* adv_out, float* value_target_out,
                   float* value_old_out, const float* record_rows, const void* xs_rows, int32_t planes, void* stream);
int64_t pnr_mlp_grad_floats(void);
int pnr_mlp_train_step(const pnr_mlp_step* s, void* stream);
int pnr_mlp_adam(const pnr_mlp_step* s, const float* flat_grad, float grad_scale, void* stream);

int64_t pnr_num_envs(pnr_handle h);

/* Last error message of `h`, or of the calling thread when h == NULL. */
const char* pnr_last_error(pnr_handle h);

int pnr_abi_version(void);

/* What this binary was built from: "api=<sha16>;learn=<sha16>;" — per translation unit, the first 16 hex digits of sha256 over
 * the compile flags, the unit's source files and this header, baked in at compile time (-DPNR_UNIT_FINGERPRINT).  The Python
 * loader (pioneer_amd/_lib.py load_library) recomputes it from the tree and refuses a library that differs; bench.py prints
 * it next to the fingerprints of the committed counter passes.  (No reference counterpart: build hygiene.) */
const char* pnr_build_fingerprint(void);

#ifdef __cplusplus
}
#endif
#endif /* PIONEER_AMD_H */
