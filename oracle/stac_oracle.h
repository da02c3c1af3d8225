/*
 * stac_oracle.h -- CPU restatement of the STAC hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the checker, not the product: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product path (stac_mjx_amd/) never links,
 * imports or calls anything in oracle/.
 *
 * What it restates (reference = talmolab/stac-mjx, citations relative to /root/reference):
 *   orc_fk            mujoco.mjx._src.smooth.kinematics as called from stac_mjx/utils.py:49-60
 *                     (un-vendored dependency mujoco-mjx, unpinned in pyproject.toml:23-24;
 *                     algorithm restated in SURVEY.md appendix A1, validated against the
 *                     stored reference output demos/demo_viz.p to 2e-7)
 *   orc_q_loss        stac_mjx/stac_core.py:27-63   (+ analytic gradient replacing jax.grad)
 *   orc_q_opt         stac_mjx/stac_core.py:66-99,182-235 -> jaxopt==0.8.5 ProjectedGradient.run
 *                     (un-vendored, pyproject.toml:35; restated from its published algorithm:
 *                     FISTA-accelerated proximal gradient with backtracking line search,
 *                     SURVEY.md appendix A2)
 *   orc_m_opt         stac_mjx/stac_core.py:102-172
 *   orc_root_optimization / orc_pose_optimization / orc_ik_clips
 *                     stac_mjx/compute_stac.py:17-104,170-278 and stac_mjx/stac.py:356-454
 *
 * PARITY STATUS: FK is pinned against demos/demo_viz.p (real reference output, 9e-8) and m_opt against the
 * known-answer cases of tests/unit/test_m_opt.py.  The reference holds NO numeric test or fixture for
 * q_opt / pose_optimization (tests mock it: tests/unit/test_compute_stac.py:32-51) and jaxopt/jax/mujoco cannot
 * be imported here, so a bit-level replay of jaxopt's iterates is unverifiable.  What CAN be tied to a real
 * MJX + jaxopt run is, and is tested (tests/test_pin_demo_viz.py): the stored fit of demo_viz.p satisfies this
 * oracle's stopping rule for the reference's last solve of every frame, is a fixed point of its part solves to
 * 3-5e-5, and the whole driver fits the same 50 frames to 0.80-0.82 mm against the stored fit's 0.92 mm.
 * The q_phase is therefore pinned IN MARKER SPACE AND AT THE STOPPING RULE, not iterate by iterate.
 *
 * All arithmetic is IEEE float32 with explicit fmaf() in the kinematics (the operation sequence the HIP
 * kernels execute), compiled with -ffp-contract=off so nothing else is contracted.
 */
#ifndef STAC_ORACLE_H
#define STAC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_JNT_FREE = 0, ORC_JNT_BALL = 1, ORC_JNT_SLIDE = 2, ORC_JNT_HINGE = 3 };

typedef struct {
    int32_t nbody, njnt, nq, nsite;
    const int32_t *body_parentid; /* [nbody] */
    const float *body_pos;        /* [nbody,3] */
    const float *body_quat;       /* [nbody,4] */
    const int32_t *body_jntadr;   /* [nbody] */
    const int32_t *body_jntnum;   /* [nbody] */
    const int32_t *jnt_type;      /* [njnt] */
    const int32_t *jnt_qposadr;   /* [njnt] */
    const int32_t *jnt_bodyid;    /* [njnt] */
    const float *jnt_pos;         /* [njnt,3] */
    const float *jnt_axis;        /* [njnt,3] */
    const float *qpos0;           /* [nq] */
    const int32_t *site_bodyid;   /* [nsite] */
    const float *site_pos;        /* [nsite,3] (the marker offsets) */
} orc_model;

typedef struct {
    float tol;       /* FTOL */
    int32_t maxiter; /* N_ITER_Q */
    int32_t maxls;   /* jaxopt default 15 */
} orc_pg_params;

typedef struct {
    int32_t iter_num;
    float stepsize;
    float error;
    float t;
    int32_t ls_evals;   /* number of loss-only evaluations (line-search candidates) */
    int32_t grad_evals; /* number of value+gradient evaluations */
    float loss;         /* q_loss(params) -- not part of the reference state; for reports */
} orc_pg_state;

/* Forward kinematics.  qpos is in/out (free/ball quaternions are normalised and written back,
 * like MJX).  Any output pointer may be NULL. */
void orc_fk(const orc_model *m, float *qpos, float *xpos, float *xquat, float *xanchor,
            float *xaxis, float *site_xpos);

/* q_loss (stac_core.py:27-63).  grad may be NULL.  Returns the loss. */
float orc_q_loss(const orc_model *m, const float *q, const float *kp, const uint8_t *qs_to_opt,
                 const uint8_t *kps_to_opt, const float *initial_q, float *grad);

double orc_q_loss_d(const orc_model *m, const float *q, const float *kp, const uint8_t *qs_to_opt,
                    const uint8_t *kps_to_opt, const float *initial_q, float *grad);

/* One StacCore.q_opt call: params_out[nq] = res.params (NOT blended with q0; the caller applies
 * make_qs like compute_stac.py does). */
void orc_q_opt(const orc_model *m, const orc_pg_params *p, const float *kp,
               const uint8_t *qs_to_opt, const uint8_t *kps_to_opt, const float *q0,
               const float *lb, const float *ub, float *params_out, orc_pg_state *state_out);

/* ---- optional fast solver (NOT the reference's algorithm) --------------------------------------------------
 * Projected Levenberg-Marquardt on the same objective, bounds and masks as orc_q_opt (the "LM qpos update" of
 * BASELINE.json's north star).  It does not reproduce the reference's truncated
 * projected-gradient iterates and is judged in marker space against them.  It IS the operation sequence of the HIP kernel
 * (stac_mjx_amd/csrc/stac_lm.hip), so that the GPU tests compare the two bit for bit.  Per iteration: the loss as the sum
 * of the per-site terms four at a time in site order; Gauss-Newton entries A[b][a] = sum over the DFS-ordered sites below
 * coordinate b of dot3(J_b, J_a) for the coordinates a on b's root path (others are structurally zero), b = -g / 2, a gauge
 * term on every raw quaternion (free root, ball joints); multiplicative damping fma(A_ii, lambda, A_ii) + 1e-9, bound-active coordinates frozen;
 * Featherstone's L^T D L on the root paths (pivots in decreasing qpos order: no fill-in), y = L^-T b on the fly,
 * z = D^-1 y, d = L^-1 z ancestors first; step clipped to the box, accepted if the loss decreases (lambda /= 2) else
 * lambda *= 4 (at most 8 times in a row; a matrix that is not positive definite counts as a rejected evaluation of the
 * point itself AND quadruples lambda once more on that turn: x 16 -- kernel and oracle alike, stac_lm.hip: the factorisation failed, so
 * the damping is raised faster than after an ordinary rejected step; tests/test_gpu_parity.py keeps a check of the LM answer that does
 * not go through this file: the marker-space comparison against the PG solver).  Stops on the same residual as the PG solver (||clip(x - grad) - x|| <= tol) or after maxiter accepted
 * steps.  A ball joint's four raw quaternion components are coordinates like the free root's (columns in the frame the ball
 * rotation is applied in, the same gauge term). */
typedef struct {
    float tol;       /* same stopping residual as the PG solver */
    int32_t maxiter; /* accepted steps, e.g. 40 */
    float lambda0;   /* initial damping, e.g. 1e-2 */
} orc_lm_params;

void orc_q_opt_lm(const orc_model *m, const orc_lm_params *p, const float *kp, const uint8_t *qs_to_opt,
                  const uint8_t *kps_to_opt, const float *q0, const float *lb, const float *ub,
                  float *params_out, orc_pg_state *state_out);

/* Self-check of the LM Jacobian columns against the analytic gradient (all coordinates, all sites weighted 1): max |J^T f - grad|. */
double orc_lm_jac_check(const orc_model *m, const float *q, const float *kp);

/* orc_ik_clips with the LM solver in place of every PG solve (same sequencing, masks and replace_qs). */
void orc_ik_clips_lm(const orc_model *m, const orc_lm_params *p, const float *kp, int32_t C, int32_t F,
                     const float *lb, const float *ub, const uint8_t *part_masks, int32_t P,
                     const uint8_t *trunk_kps, int32_t root_kp_idx, int32_t root_dims, int32_t do_root_opt,
                     const float *q_init, float *qposes, float *xposes, float *xquats, float *markers,
                     float *frame_error, uint32_t *counters, int32_t nthreads);

/* Cross-frame partial sums of the offset phase: partial[3K+2] = {s[K,3], z2, T}. */
void orc_m_partial(const orc_model *m, const float *keypoints, const float *q, int32_t T,
                   float *partial);
/* Closed form from the (all-reduced) partial sums. */
void orc_m_finish(int32_t K, const float *partial, const float *initial_offsets,
                  const float *is_regularized, float reg_coef, float *params_out,
                  float *error_out);
/* _m_opt = partial + finish. */
void orc_m_opt(const orc_model *m, const float *keypoints, const float *q, int32_t T,
               const float *initial_offsets, const float *is_regularized, float reg_coef,
               float *params_out, float *error_out);

/* compute_stac.root_optimization on frame `frame` of one clip; qpos[nq] is the carried
 * mjx_data.qpos (in/out). */
void orc_root_optimization(const orc_model *m, const orc_pg_params *p, const float *kp_clip,
                           int32_t frame, int32_t root_kp_idx, int32_t root_dims, const float *lb,
                           const float *ub, const uint8_t *trunk_kps, float *qpos,
                           orc_pg_state *last_state);

/* compute_stac.pose_optimization over one clip of F frames (sequential warm start).
 * Outputs may be NULL: qposes[F,nq], xposes[F,nbody,3], xquats[F,nbody,4], markers[F,K,3],
 * frame_error[F], counters[F,4] = {sum iters, sum ls_evals, sum grad_evals, n_solves}. */
void orc_pose_optimization(const orc_model *m, const orc_pg_params *p, const float *kp_clip,
                           int32_t F, const float *lb, const float *ub,
                           const uint8_t *part_masks, int32_t P, float *qpos, float *qposes,
                           float *xposes, float *xquats, float *markers, float *frame_error,
                           uint32_t *counters);

/* Stac.ik_only over C clips (stac.py:356-454): per clip qpos <- qpos0, optional root
 * optimisation on frame 0, then pose optimisation.  OpenMP-parallel over clips.
 * Layouts are clip-major: kp[C,F,3K] -> qposes[C,F,nq] etc. */
void orc_ik_clips(const orc_model *m, const orc_pg_params *p, const float *kp, int32_t C,
                  int32_t F, const float *lb, const float *ub, const uint8_t *part_masks,
                  int32_t P, const uint8_t *trunk_kps, int32_t root_kp_idx, int32_t root_dims,
                  int32_t do_root_opt, const float *q_init, float *qposes, float *xposes,
                  float *xquats, float *markers, float *frame_error, uint32_t *counters,
                  int32_t nthreads);

int32_t orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
