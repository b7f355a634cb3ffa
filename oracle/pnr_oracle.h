/*
 * pnr_oracle.h — CPU restatement of the reference's Pioneer-arm step path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pioneer_amd/ may include, link or
 * call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (xdralex/pioneer) holds no tests, golden
 * vectors or fixtures for this path, and its own code cannot run here
 * (pybullet / gym / ray are absent: ordinary ModuleNotFoundError).  This file
 * restates the algorithm from the reference's source text; the only
 * reference-held data it is checked against is the URDF
 * (tests/golden/urdf_chain.json, extracted by tests/golden/make_urdf_chain.py)
 * and the analytic KATs of SURVEY.md Appendix C.
 */
#ifndef PNR_ORACLE_H
#define PNR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_DOF 6
#define ORC_OBS 137

/* Where values are rounded when they are stored between steps. */
enum orc_precision {
    ORC_REF = 0, /* exactly the reference: r and target are float64 right after
                    reset (pioneer_knm_env.py:81,90,94), potential is a Python
                    float; r/v become float32 at the first step (:117-118)   */
    ORC_DEV = 1  /* device storage model: r, target and potential are rounded
                    to float32 whenever they are stored; arithmetic unchanged */
};

typedef struct orc_params {
    /* PioneerKinematicConfig, pioneer_knm_env.py:19-34 */
    double max_v_to_r, max_a_to_v, done_distance;
    double award_max, award_done, award_potential_slope, penalty_step;
    double target_lo[3], target_hi[3];
    /* SimulationConfig, bullet_env.py:36-44 */
    double timestep;
    int32_t frame_skip;
    /* TimeLimit(max_episode_steps), pioneer_knm_train.py:27; 0 = off */
    int32_t max_episode_steps;
    int32_t precision; /* enum orc_precision */
    int32_t auto_reset;
    uint64_t seed;
    /* derived, pioneer_knm_env.py:56-61 (filled by orc_params_default/derive) */
    float r_lo[ORC_DOF], r_hi[ORC_DOF], v_max[ORC_DOF], a_max[ORC_DOF];
    double dt, eps;
} orc_params;

typedef struct orc_state {
    float a[ORC_DOF];   /* acceleration commanded on the previous step  */
    float v[ORC_DOF];
    double r[ORC_DOF];  /* float64 after reset (ORC_REF), float32-valued after a step */
    int32_t r_is_f64;   /* 1 between reset and the first step in ORC_REF */
    double target[3];
    double potential;
    uint32_t step_index;
    uint32_t episode;   /* number of resets drawn so far (RNG counter) */
} orc_state;

void orc_params_default(orc_params* p);
/* recompute r_lo..a_max, dt, eps from the tunables */
void orc_params_derive(orc_params* p);

/* Forward kinematics of link robot:pointer, float64
 * (bullet_scene.py:53-59 getLinkState -> link_world_position;
 *  chain pioneer_knm_6dof.urdf:204-275). */
void orc_fk_pointer(const double q[ORC_DOF], double out_xyz[3]);

/* compute_potential, pioneer_knm_env.py:232-236 */
double orc_potential(const orc_params* p, double distance);

/* Philox4x32-10 (Salmon et al., SC'11): the counter-based generator both the
 * oracle and the engine use for reset draws (the reference uses numpy's
 * MT19937 RandomState, whose stream is not reproduced: reset parity is
 * distributional + through the explicit override arguments). */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* reset_world, pioneer_knm_env.py:76-105.  joint_pos / target_pos may be NULL
 * (then drawn from Philox(seed, global_env_id, episode)). */
void orc_reset(const orc_params* p, orc_state* s, uint64_t global_env_id,
               const double* joint_pos, const double* target_pos);

/* observe, pioneer_knm_env.py:184-211 -> float64[137] */
void orc_observe(const orc_params* p, const orc_state* s, double obs[ORC_OBS]);

/* pieces of step(), exported for the dynamics-mode oracle */
void orc_integrate(const orc_params* p, orc_state* s, const float action[ORC_DOF]);
int orc_reward(const orc_params* p, orc_state* s, const double q[ORC_DOF], double* reward, double info[4]);
void orc_observe_qv(const orc_params* p, const orc_state* s, const double q[ORC_DOF],
                    const double qd[ORC_DOF], int q_is_f64, double obs[ORC_OBS]);
void orc_draw_block(const orc_params* p, uint64_t genv, uint32_t episode, uint32_t block, double u[4]);

/* BulletEnv.step, bullet_env.py:192-197: act (pioneer_knm_env.py:111-182) +
 * observe + TimeLimit + optional auto-reset.
 * info[4] = r_pot, r_step, r_done, dist.  Any output pointer may be NULL. */
void orc_step(const orc_params* p, orc_state* s, uint64_t global_env_id,
              const float action[ORC_DOF], double obs[ORC_OBS], double* reward,
              uint8_t* done, uint8_t* truncated, double info[4]);

/* Batched (OpenMP over envs when nthreads > 1).  actions [n][6] float32,
 * obs [n][137] float64 (may be NULL), reward [n] float64. */
void orc_reset_batch(const orc_params* p, orc_state* s, int64_t n, int64_t env_id_offset,
                     const uint8_t* mask, const double* joint_pos, const double* target_pos,
                     double* obs, int nthreads);
void orc_step_batch(const orc_params* p, orc_state* s, int64_t n, int64_t env_id_offset,
                    const float* actions, double* obs, double* reward, uint8_t* done,
                    uint8_t* truncated, double* info, int nthreads);

/* Pack / unpack the engine's planar state words [24][n] (see
 * include/pioneer_amd.h pnr_get_state). */
void orc_state_to_words(const orc_state* s, int64_t n, uint32_t* words);
void orc_state_from_words(orc_state* s, int64_t n, const uint32_t* words);

int orc_sizeof_state(void);
int orc_sizeof_params(void);

#ifdef __cplusplus
}
#endif
#endif
