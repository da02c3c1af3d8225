// stac_plan.hpp -- data shared between the host side of libstac_hip.so and its kernels.
//
// The "plan" is the marker-ancestor subtree of the body tree (rodent 31 of 67 bodies,
// 39 of 68 joints) laid out level by level, i.e. exactly what one q_loss evaluation
// (stac_mjx/stac_core.py:27-63) touches.  It is one contiguous blob of 32-bit words in device
// memory which every workgroup copies into LDS once.
#pragma once
#include <cstdint>

namespace stac {

constexpr int kMaxKinds = 40;  // root pass x2 + full + up to 37 part groups

struct PlanHeader {
    int32_t nbody, njnt, nq, K;
    int32_t nab;     // active bodies (ancestors-or-self of a fit site), sorted by (depth, id)
    int32_t naj;     // joints of active bodies, in slot order
    int32_t nlev;    // levels of the active tree
    int32_t nquat;   // quaternion joints (free / ball) in the WHOLE model
    int32_t nqpad;   // nq rounded up to a multiple of 4
    int32_t has_ball;
    // word offsets into the blob -------------------------------------------------------------
    int32_t off_lev_adr;     // [nlev+1] first slot of each level
    int32_t off_ab_parent;   // [nab] parent's index in the per-chain transform array (0 = world, s+1 = slot s)
    int32_t off_ab_jadr;     // [nab] first active joint
    int32_t off_ab_jnum;     // [nab]
    int32_t off_ab_sadr;     // [nab] own sites: start in site_list
    int32_t off_ab_snum;     // [nab]
    int32_t off_ab_cadr;     // [nab] children: start in child_list
    int32_t off_ab_cnum;     // [nab]
    int32_t off_site_list;   // [K]  site ids grouped by body slot, increasing id inside a body
    int32_t off_child_list;  // [nab] child slots, DEcreasing body id inside a parent (oracle order)
    int32_t off_ab_pos;      // [nab*3] float
    int32_t off_ab_quat;     // [nab*4] float
    int32_t off_aj_type;     // [naj]
    int32_t off_aj_qadr;     // [naj]
    int32_t off_aj_slot;     // [naj] body slot
    int32_t off_aj_pos;      // [naj*3] float
    int32_t off_aj_axis;     // [naj*3] float
    int32_t off_aj_q0;       // [naj] float: qpos0[qadr] (hinge / slide reference)
    int32_t off_site_slot;   // [K]
    int32_t off_site_pos;    // [K*3] float -- the marker offsets; mutable (stac_set_site_pos)
    int32_t off_lb;          // [nqpad] float
    int32_t off_ub;          // [nqpad] float
    int32_t off_qpos0;       // [nqpad] float
    int32_t off_quat_adr;    // [nquat] qpos address of every quaternion (free: adr+3, ball: adr)
    int32_t total_words;
    // per-chain LDS layout (float offsets inside one chain's region) ------------------------------
    int32_t c_bx;      // [(nab+1)*7] pos(3) quat(4); entry 0 = world
    int32_t c_ja;      // [naj*6] anchor(3) axis(3)
    int32_t c_jq;      // [naj*4] quaternion before a ball joint (only if has_ball)
    int32_t c_jn;      // [naj]   |q| of free/ball quaternions
    int32_t c_sw;      // [K*6] site wrench f(3) t(3); aliased by gg[nqpad] (gradient out)
    int32_t c_bw;      // [nab*6] body wrench; aliased by r2[K] (before) and red[2*nqpad] (after)
    int32_t c_qe;      // [nqpad] evaluation point (quaternions normalised in place)
    int32_t c_kp;      // [3K] keypoints of the current frame
    int32_t chain_stride;
};

// Full-model tables for the stand-alone FK / offset-phase kernels (device pointers).
struct FullModel {
    int32_t nbody, njnt, nq, K;
    const int32_t *body_parentid, *body_jntadr, *body_jntnum;
    const float *body_pos, *body_quat;
    const int32_t *jnt_type, *jnt_qposadr;
    const float *jnt_pos, *jnt_axis, *qpos0;
    const int32_t *site_bodyid;
    const float *site_pos;  // points INTO the plan blob (single source of truth for the offsets)
};

struct QArgs {
    const PlanHeader *hdr;  // device copy of the header
    const float *plan;      // device blob
    PlanHeader h;           // by-value copy (kernel argument, scalar registers)
    // problem
    const float *kp;        // [C,F,3K]
    const float *q_init;    // [C,nq] or null
    const uint8_t *masks;   // device [nkinds, nqpad] bytes: kind 0,1 = root passes, 2 = full, 3.. = parts
    const uint8_t *kpw;     // device [2, K] bytes: row 0 = trunk mask (root passes), row 1 = all ones / single-solve mask
    const uint8_t *kpw3;    // device [3K] per-coordinate mask for single-solve mode (or null)
    int32_t C, F, P;
    int32_t root_kp_idx, do_root_opt;
    int32_t single;         // 1 = stac_q_solve mode (one solve, outputs x unblended + state)
    int32_t mb_words;       // LDS words reserved for the per-kind mask bit table (multiple of 4)
    float tol;
    int32_t maxiter, maxls;
    // outputs
    float *qpos_out;        // [C,F,nq]   (single: params_out [N,nq])
    float *err_out;         // [C,F]      (single: state_out [N,4])
    uint32_t *counters_out; // [C,F,4] or null
    float *q_carry_out;     // [C,nq] or null
};

}  // namespace stac
