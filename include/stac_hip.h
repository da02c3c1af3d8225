/*
 * stac_hip.h -- C ABI of the MI355X-native STAC pose-fitting engine (libstac_hip.so).
 *
 * This is the drop-in boundary for the hot path of talmolab/stac-mjx: the solver seam
 * `StacCore.q_opt / StacCore.m_opt` (stac_mjx/stac_core.py:175-275) and, one level up, the three
 * phase drivers of stac_mjx/compute_stac.py that call it once per frame (:17-104, :107-167,
 * :170-278) and that `Stac.ik_only` vmaps over clips (stac_mjx/stac.py:405-440).
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in the signatures (`stream` is a hipStream_t passed
 *     as void*; NULL = the default stream).
 *   - every `float*` / `uint32_t*` data argument is a DEVICE pointer owned by the caller (e.g. a
 *     PyTorch-ROCm tensor) unless the comment says "host".  The library never frees caller memory.
 *   - all calls are asynchronous on `stream`.  Host-side waits, complete list: (1) a stac_q_solve / stac_q_phase
 *     call with a NEW set of masks (or stac_q_solve with new lb / ub) waits once for its own small table uploads
 *     on that stream (the host arrays are the caller's temporaries); (2) stac_fk / stac_q_phase called WITHOUT
 *     xpos / xquat outputs grow an internal scratch block to its high-water mark -- growing frees the old block,
 *     which waits for the device.  Launch control words are set by a kernel on the stream, every other buffer is
 *     allocated at stac_model_create.
 *   - a model belongs to the device that was current at stac_model_create; every entry point makes that device
 *     current for its duration and restores the caller's.
 *   - developer switches (STAC_HIP_* environment variables) are read once, at
 *     stac_model_create.
 *   - return value: 0 = OK, negative = error (see stac_last_error()); no exceptions cross the ABI.
 *   - layouts are C-contiguous float32; clip-major: kp[C][F][3K], qpos[C][F][nq].
 *   - re-entrant per model: one call at a time on a given stac_model, which owns scratch buffers for the launch in
 *     flight;
 *     no global state besides the thread-local error string.
 */
#ifndef STAC_HIP_H
#define STAC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STAC_HIP_ABI_VERSION 3

/* mjtJoint values (same as MuJoCo's, so tables from a MuJoCo compile can be passed as they are). */
enum { STAC_JNT_FREE = 0, STAC_JNT_BALL = 1, STAC_JNT_SLIDE = 2, STAC_JNT_HINGE = 3 };

enum {
    STAC_OK = 0,
    STAC_ERR_INVALID = -1,  /* bad argument */
    STAC_ERR_HIP = -2,      /* a HIP runtime call failed */
    STAC_ERR_CAPACITY = -3, /* model exceeds the compiled kernel limits */
    STAC_ERR_NO_DEVICE = -4 /* no usable GPU */
};

/* Flat kinematic model, HOST pointers, copied at stac_model_create.
 * Replaces what the reference gets from mjx.put_model(mj_model) (stac_mjx/utils.py:34-46) plus the
 * bounds of stac_mjx/stac.py:54-88.  Field meaning = MuJoCo's mjModel fields of the same name. */
typedef struct stac_model_tables {
    int32_t nbody, njnt, nq, nsite;  /* nsite = K fit sites (one per keypoint) */
    const int32_t *body_parentid;    /* [nbody]   */
    const float *body_pos;           /* [nbody,3] */
    const float *body_quat;          /* [nbody,4] w,x,y,z */
    const int32_t *body_jntadr;      /* [nbody]   (-1 if none) */
    const int32_t *body_jntnum;      /* [nbody]   */
    const int32_t *jnt_type;         /* [njnt]    */
    const int32_t *jnt_qposadr;      /* [njnt]    */
    const int32_t *jnt_bodyid;       /* [njnt]    */
    const float *jnt_pos;            /* [njnt,3]  */
    const float *jnt_axis;           /* [njnt,3]  unit */
    const float *qpos0;              /* [nq]      */
    const int32_t *site_bodyid;      /* [K]       */
    const float *site_pos;           /* [K,3]     initial marker offsets */
    const float *lb;                 /* [nq]  box bounds of the q_phase (stac.py:54-88) */
    const float *ub;                 /* [nq]  */
} stac_model_tables;

/* Solver hyper-parameters (host struct).  Replaces StacCore.__init__(tol, n_iter_q)
 * (stac_mjx/stac_core.py:182-191) and the jaxopt defaults it relies on. */
typedef struct stac_q_params {
    float tol;               /* FTOL: stop when ||clip(x - grad) - x||_2 <= tol */
    int32_t maxiter;         /* N_ITER_Q (>= 1) */
    int32_t maxls;           /* line-search halvings, jaxopt default 15 */
    int32_t lanes_per_chain; /* 0 = auto; else 8, 16, 32 or 64 lanes of a wavefront per chain (a width that is not built for
                                this model -- 4 always -- runs on the next wider one; results do not depend on it) */
    int32_t solver;          /* STAC_SOLVER_PG (the reference's algorithm, parity mode) or STAC_SOLVER_LM */
    float lm_lambda0;        /* LM only: initial damping (0 -> 1e-2); maxiter then counts accepted LM steps */
} stac_q_params;

/* STAC_SOLVER_LM is an optional fast solver, NOT the reference's algorithm: projected Levenberg-Marquardt on the
 * same objective, masks, bounds, stopping residual and sequencing (analytic site Jacobians, J^T J + damping,
 * Cholesky in LDS).  It fits the markers at least as well as the truncated projected gradient but does not
 * reproduce its iterates; it is available in stac_q_phase only. */
enum { STAC_SOLVER_PG = 0, STAC_SOLVER_LM = 1 };

typedef struct stac_model stac_model; /* opaque */

/* Thread-local description of the last error returned on this thread ("" if none). */
const char *stac_last_error(void);
int32_t stac_abi_version(void);
/* Number of visible GPUs (0 if none); does not initialise a device. */
int32_t stac_device_count(void);

/* Uploads the tables to the current device.  Returns NULL on failure (see stac_last_error). */
stac_model *stac_model_create(const stac_model_tables *host_tables);
void stac_model_destroy(stac_model *m);
/* info[8] (host) = {nbody, njnt, nq, K, n_active_bodies, n_active_joints, n_levels, max_lanes_hint} */
int32_t stac_model_info(const stac_model *m, int32_t *info);

/* utils.set_site_pos / get_site_pos (stac_mjx/utils.py:93-126): offsets[K,3] device pointer. */
int32_t stac_set_site_pos(stac_model *m, const float *offsets, void *stream);
int32_t stac_get_site_pos(const stac_model *m, float *offsets_out, void *stream);

/* utils.kinematics on N poses (stac_mjx/utils.py:49-60 -> mjx smooth.kinematics).
 * qpos[N,nq] is read; outputs may be NULL: qpos_norm_out[N,nq] (quaternions normalised, as MJX
 * writes back), xpos[N,nbody,3], xquat[N,nbody,4], site_xpos[N,K,3]. */
int32_t stac_fk(const stac_model *m, const float *qpos, int32_t N, float *qpos_norm_out, float *xpos,
                float *xquat, float *site_xpos, void *stream);

/* Batched StacCore.q_opt (stac_mjx/stac_core.py:193-235): N independent solves with the same masks.
 *   kp[N,3K], q0[N,nq]; qs_to_opt[nq] and kps_to_opt[3K] are HOST uint8 masks.
 *   lb[nq], ub[nq]: HOST float box of THIS call, as q_opt takes it per call (hyperparams_proj, stac_core.py:83);
 *   both NULL = the bounds given at stac_model_create.
 *   params_out[N,nq]  = res.params (not blended with q0 -- the caller applies make_qs)
 *   state_out[N,4]    = {error, stepsize, t, loss(params)}   (res.state)
 *   counters_out[N,4] = {iter_num, ls_evals, grad_evals, 1}  (may be NULL) */
int32_t stac_q_solve(const stac_model *m, const stac_q_params *p, const float *kp, const float *q0,
                     const uint8_t *qs_to_opt, const uint8_t *kps_to_opt, const float *lb, const float *ub,
                     int32_t N, float *params_out, float *state_out, uint32_t *counters_out, void *stream);

/* The q_phase of C clips of F frames each, one warm-started chain per clip: per clip
 * [root_optimization on frame 0 (compute_stac.py:17-104) if do_root_opt] then pose_optimization
 * (compute_stac.py:170-278: per frame one full-body solve + P part solves, each followed by
 * replace_qs).  This is what Stac.ik_only vmaps over clips (stac.py:405-440) and what
 * Stac.fit_offsets runs on its single chain (stac.py:298-311, C = 1, q_init = carried qpos).
 *   kp[C,F,3K]; q_init[C,nq] or NULL (-> qpos0, like mjx.make_data)
 *   part_masks: HOST uint8 [P,nq]; trunk_kps: HOST uint8 [K] (used only when do_root_opt)
 *   qpos_out[C,F,nq], err_out[C,F] (PG residual of the frame's LAST solve, compute_stac.py:252),
 *   counters_out[C,F,4] = {sum iter, sum ls_evals, sum grad_evals, n_solves} (may be NULL),
 *   q_carry_out[C,nq] final qpos of every chain (may be NULL),
 *   xpos_out[C,F,nbody,3], xquat_out[C,F,nbody,4], markers_out[C,F,K,3] (each may be NULL). */
int32_t stac_q_phase(const stac_model *m, const stac_q_params *p, const float *kp,
                     const float *q_init, const uint8_t *part_masks, const uint8_t *trunk_kps,
                     int32_t C, int32_t F, int32_t P, int32_t root_kp_idx, int32_t root_dims,
                     int32_t do_root_opt, float *qpos_out, float *err_out, uint32_t *counters_out,
                     float *q_carry_out, float *xpos_out, float *xquat_out, float *markers_out,
                     void *stream);

/* Cross-frame sums of the offset phase (_m_opt, stac_mjx/stac_core.py:148-160) over T frames:
 * partial[3K+2] = { s[K,3] = sum_t R^T (y - p),  z2 = sum |y - p|^2,  T }.
 * With several GPUs each rank calls this on its shard and the host all-reduces `partial`
 * (one RCCL all-reduce of 3K+2 floats) before stac_m_phase_finish.
 * workspace: device scratch of at least stac_m_phase_workspace_floats(m, T) floats. */
int64_t stac_m_phase_workspace_floats(const stac_model *m, int32_t T);
int32_t stac_m_phase_partial(const stac_model *m, const float *kp, const float *q, int32_t T,
                             float *workspace, float *partial, void *stream);
/* Closed form (stac_core.py:162-170): offsets_out[K,3], err_out[1].  All device pointers. */
int32_t stac_m_phase_finish(const stac_model *m, const float *partial, const float *initial_offsets,
                            const float *is_regularized, float reg_coef, float *offsets_out,
                            float *err_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* STAC_HIP_H */
