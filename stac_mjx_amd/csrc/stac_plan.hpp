// stac_plan.hpp -- data shared between the host side of libstac_hip.so and its kernels.
//
// The "plan" is the marker-ancestor subtree of the body tree (rodent 31 of 67 bodies,
// 39 of 68 joints) laid out level by level, i.e. exactly what one q_loss evaluation
// (stac_mjx/stac_core.py:27-63) touches.  It is one contiguous blob of 32-bit words in device
// memory which every workgroup copies into LDS once.
#pragma once
#include <cstdint>

namespace stac {

constexpr int kPlaceHdr = 16, kPlaceWords = kPlaceHdr + 16 * 1024;  // QArgs::place: header, then one counter per (XCC, SE/SH/CU, SIMD)
constexpr int kMaxKinds = 40;  // root pass x2 + full + up to 37 part groups
// A lane of a G-lane group takes the sites k = r * G + lane, r < kSiteRounds: their keypoints and loss terms stay in
// registers when K <= kSiteRounds * G (host and kernel evaluate the same condition); else they go through LDS.
constexpr int kSiteRounds = 3;
// FISTA's momentum sequence t_0 = 1, t_(k+1) = (1 + sqrt(1 + 4 t_k^2)) / 2 depends on the iteration number alone: the plan holds
// {t_(k+1), (t_k - 1) / t_(k+1)} for k < kTTab right in front of the joint records (PlanHeader::off_joint - 2 kTTab: a launch that
// wants it stages from there on), computed on the host
// with the kernel's own float32 expressions; later iterations compute them (stac_kernels.hip)
constexpr int kTTab = 256;
// The lean kernels (stac_kernels.hip) take the sites in as many rounds as a rodent-sized marker set needs at their group width
// (K <= 32: two rounds of 16 lanes, one of 32) instead of three of which the last ones are empty: the host checks K against it
// (launch_q_phase).  The loss tree is the same sum: the registers that fall away held +0, and a sum of squares is never -0.
// (NQR: solver registers per lane of the instantiation -- the wide models' shapes, more than four, take two rounds at 32 lanes)
constexpr int lean_site_rounds(int G, int NQR) { return G >= 32 ? (NQR > 4 ? 2 : 1) : (G >= 16 ? 2 : kSiteRounds); }
// A transform entry in a chain's LDS region: position (3 words) then quaternion (4 words, in the order x, y, z, w).  Used for the body
// transforms (c_bx), the per-joint {anchor, pre-joint quaternion} entries (c_ja; the pre-pass parks the joint-local
// quaternion in the quaternion words) and, as {f, t}, for the site wrenches and range sums.  Packed (7 words) by
// default; -DSTAC_XF_WORDS=8 pads the entries to 16 bytes so that they move with two ds_read_b128 / ds_write_b128
// instead of four scalar-pair instructions.  Measured: the aligned form is neutral in the latency modes and costs the
// throughput modes 3-15 % (12 % more LDS per chain = fewer resident chains), so packed it is.
#ifndef STAC_XF_WORDS
#define STAC_XF_WORDS 7
#endif
constexpr int kXf = STAC_XF_WORDS;     // 8 (aligned, ds_*_b128) or 7 (packed: 12 % less LDS per chain, scalar LDS accesses)
constexpr int kXq = kXf == 8 ? 4 : 3;  // word of the quaternion (second vector) inside an entry
constexpr int kXw = kXq + 3;           // a quaternion is stored (x, y, z, w): lane c of a quad then finds its quaternion
                                       // component kXq words behind its position component (stac_device.hpp, FkQuadLane)

// Records are 16-byte aligned so the kernel fetches them with ds_read_b128.
struct BodyRec {      // 12 words
    int32_t parent;   // transform index of the parent in the per-chain array bx: 0 = world
    int32_t jadr;     // first joint record
    int32_t jnum;
    int32_t flags;    // bit 0: body_quat is the identity (product with it is exact, skipped)
                      // bit 1: the parent sits at the same position of the previous level (its transform is still
                      //        in the registers of the lane that now takes this body)
                      // bits 16-31: transform index this body is stored at (0xFFFF: nobody reads it back: not stored)
    float pos[3];
    int32_t jzero;    // bit i: joint i of the body has jnt_pos == 0 (rotate(0, q) = 0: anchor = pos, exact)
    float quat[4];
};
struct JointRec {     // 12 words
    int32_t type;     // mjtJoint
    int32_t qadr;
    int32_t slo;      // sorted-site range [slo, shi) of the joint's body subtree
    int32_t shi;
    float pos[3];
    float q0;         // qpos0[qadr] (hinge / slide reference); free / ball: int32 ordinal among the quaternion joints
    float axis[3];
    int32_t rid;      // index of the joint's site range among the DISTINCT subtree ranges (RangeRec)
};
// The subtree wrench a joint needs is the sum, left to right from zero, of the site wrenches at the sorted-site
// positions [lo, hi) of its body's subtree.  Joints on one body, and bodies that carry no site of their own above a
// single child, share their range (rodent: 39 joints, 19 distinct ranges; mouse: 181 joints, 27), so every distinct
// range is summed ONCE -- longest first -- and the joints read the result.
struct RangeRec { int32_t lo, hi; };
// One step of the FK "program": the work of one lane position in one micro-level.  A level of the tree takes
// max(1, most joints of a body in it) micro-levels; a body's first step composes it with its parent and applies
// its first joint, further joints are further steps.  Fixed-size records at (micro_level * max_width + position)
// make every address a function of the loop counter, so the kernel fetches step k+1 while it computes step k.
//
// Every step runs the SAME straight-line arithmetic, with neutral data where a part does not apply:
//     pos  = pos + rotate(bpos, quat)          bpos = 0 when the step does not start a body      (exact no-op)
//     quat = quat * bquat                      only in models with oriented bodies; identity else (exact)
//     anchor = rotate(jpos, quat) + pos;  prequat = quat
//     quat = quat * ql                         ql = joint-local quaternion; the identity when the step has no joint
//     pos  = anchor - rotate(jpos, quat)       jpos = 0: anchor = pos stays (exact)
// which is a hinge / ball joint or nothing at all.  Free and slide joints and parents that another lane produced are
// the exceptions; the per-micro-level flag words in front of the records say whether ANY position has one, so the
// common step has no divergent branch and no select.  All offsets are word offsets into the chain's LDS region.
struct FkStep {       // 12 words (+4 when some active body has a non-identity body_quat)
    float bpos[3];    // body_pos, or zeros
    int32_t kind;     // FK_KIND_*
    int32_t par_off;  // transform of the parent (c_bx + kXf * index) when another lane produced it, else -1 (the
                      // lane's running transform is the parent's)
    int32_t ja_off;   // this joint's anchor / pre-joint quaternion entry (c_ja + kXf * j); no joint in this step: c_sink
    int32_t xf_off;   // where the body's transform goes after this step (c_bx + kXf * index); not stored: c_sink
    int32_t ql_next;  // joint-local quaternion of this position's NEXT step (fetched one step ahead), or the
                      // identity quaternion of the world entry (c_bx + kXq)
    float jpos[3];    // jnt_pos, or zeros
    int32_t aux;      // free: qpos address; slide: active joint index
    // float bquat[4] (w, x, y, z) follows when PlanHeader::fk_rec_words == 16
};
enum : int32_t { FK_KIND_PLAIN = 0, FK_KIND_FREE = 1, FK_KIND_SLIDE = 2 };
// per-micro-level flags (wave-uniform: every chain of a wavefront runs the same program in lockstep): SOME position ...
enum : int32_t {
    FK_ML_PARENT_LDS = 1,  // ... starts a body whose parent another lane (or nobody) produced
    FK_ML_SPECIAL = 2,     // ... has a free or a slide joint
    FK_ML_BODY = 4,        // ... starts a body (else the compose with body_pos = 0 is skipped: exact)
    FK_ML_JOINT = 8,       // ... has a joint (else the whole joint part is skipped)
    FK_ML_JPOS = 16,       // ... has a joint with jnt_pos != 0 (else anchor = pos and pos stays: exact)
    FK_ML_BQUAT = 32,      // ... starts a body with a non-identity body_quat
};

// The form of a micro-level's step, next to its flags (flags | form << 8, 16 bits per micro-level), for programs that are
// not uniform (PlanHeader::fk_uniform): the plain hinge / ball step on neutral data, with or without parents from LDS,
// or the general step (free joints below the top level, slide joints, oriented bodies).  More forms -- one per flag
// combination -- were each faster alone and slower together.
enum : int32_t {
    FK_FORM_GENERAL = 0,            // by the flags
    FK_FORM_BODY_JOINT = 1,         // every position composes with a parent it holds and applies a joint
    FK_FORM_PARENT_BODY_JOINT = 3,  // as BODY_JOINT, some parents come from LDS
};

struct SiteRec {      // 4 words
    float pos[3];     // the marker offset -- mutable (stac_set_site_pos writes the blob)
    int32_t slot_sortpos;  // transform index of the site's body | sorted position << 16
};

struct PlanHeader {
    int32_t nbody, njnt, nq, K;
    int32_t nab;     // active bodies (ancestors-or-self of a fit site), sorted by (depth, id)
    int32_t naj;     // joints of active bodies, in slot order
    int32_t nlev;    // levels of the active tree
    int32_t nquat;   // quaternion joints (free / ball) in the WHOLE model
    int32_t nqpad;   // nq rounded up to a multiple of 4
    int32_t has_ball;
    // word offsets into the blob (all multiples of 4) ---------------------------------------------
    int32_t off_lev_adr;   // [nlev+1] first slot of each level
    int32_t off_body;      // BodyRec[nab]
    int32_t off_joint;     // JointRec[naj]
    int32_t off_site;      // SiteRec[K]
    int32_t off_lb;        // [nqpad] float
    int32_t off_ub;        // [nqpad] float
    int32_t off_qpos0;     // [nqpad] float
    int32_t off_quat_adr;  // [nquat] qpos address of every quaternion (free: adr+3, ball: adr)
    int32_t off_active;    // [nqpad] 1 where the coordinate belongs to a joint of the active subtree (only those get a gradient entry)
    int32_t total_words;   // end of what a launch stages in LDS (the per-launch copy may stop at core_words, or inside the root program)
    int32_t plan_skip;     // ... and its beginning (per launch): a kernel that runs the FK program does not need the level tables and
                           // body records in front of the joint records, so the LDS copy starts at off_joint; the launch's header copy then carries
                           // every off_* of a staged area minus plan_skip (rebase_plan_offsets, stac_abi.hip)
    int32_t core_words;    // blob without the FK program (the program is last)
    // per-chain LDS layout (float offsets inside one chain's region) ------------------------------
    int32_t c_bx;      // [(nst+1)*kXf] transform entries of the stored bodies; entry 0 = world
    int32_t c_ja;      // [naj*kXf] anchor + quaternion before the joint (the joint pass rotates the axis)
    int32_t c_jn;      // [nqj] |q| of the active free / ball quaternions, by quaternion ordinal (JointRec::q0)
    int32_t c_sw;      // [K*kXf] site wrench entries {f(3), -, t(3), -} by sorted-site position
    int32_t c_sink;    // one entry at the tail of the site-wrench region, dead during the kinematics: where the FK program's
                       // steps store what nobody reads (instead of predicating the store)
    int32_t c_gg;      // [nqpad] gradient out: the site-wrench region (dead once the range sums are done)
    int32_t c_rw;      // [nrange*kXf] wrench sums of the distinct ranges {F(3), -, T(3), -}: the body-transform region
                       // when they fit (the transforms are dead once the sites have read them), else an own region
    int32_t nrange, off_range;  // RangeRec[nrange] in the blob
    int32_t c_qe;      // [nqpad] evaluation point (quaternions normalised in place); aliases c_sw in the PG kernel
    int32_t c_kp;      // [3K] keypoints of the current frame      } only when a lane group has more than kSiteRounds
    int32_t c_r2;      // [K padded] per-site loss terms           } sites per lane: else both stay in registers
    int32_t stride_regs, stride_lds;  // chain stride without / with those two regions (odd)
    int32_t chain_stride;  // the one of this launch (set per launch by the host)
    int32_t stride_forced; // STAC_HIP_STRIDE_ADD given: the strides are taken as they are (profiling)
    int32_t max_width; // bodies in the widest level
    int32_t nst;       // bodies whose transform is stored in LDS (a site or a child on another lane reads it)
    int32_t nqj;       // active quaternion joints (free / ball)
    int32_t kpow2;     // K rounded up to a power of two (the LDS loss tree of models with more than 64 sites)
    int32_t c_qsv;     // [4*nqj] their normalised quaternions, kept for the gradient pass
    int32_t off_fkstep;    // header (fk_hdr_words: one word per PAIR of micro-levels (FK_ML_* | FK_FORM_* << 8 of step ml, the same of step ml + 1 << 16), then the ql offset of
                           // each position's first step), then FkStep[n_mlev * max_width] (word offset into the blob)
    int32_t off_fkroot;    // same size: the pruned program of the root passes (filled per call from the trunk keypoints)
    int32_t fk_hdr_words;  // words in front of the records of either program
    int32_t n_mlev_hdr;    // micro-levels of the full program: the header holds n_mlev_hdr / 2 flag words, then the
                           // first-step ql offsets
    int32_t n_mlev;        // micro-levels of the FK program
    int32_t fk_rec_words;  // 12, or 16 when the records carry body_quat
    int32_t fk_uniform;    // 1: no step of the program needs the general form (no slide joint, no oriented body, free joints
                           // only on top-level bodies): the kernel runs ONE straight-line step without any dispatch
    // ---- split kinematics of the lean kernels ("FK3", stac_device.hpp: fk3_*) ------------------------------------------------
    // One free root at qpos 0 .. 6, hinges below it, no oriented body: the kinematics of a chain then fall apart into
    //   P1  the quaternion chain      q_j = q_pre(j) * ql_j           one product per hinge, four lanes per product (DPP)
    //   P2  the rotations             rotate(body_pos | jnt_pos, q)   one lane per rotation, all of them side by side
    //   P3  the position chain        p = p + r                       one addition per rotation, three lanes per path
    // -- the same operations on the same operands in the same order per result as the step program above (bit-identical), but
    // only the additions and the 4-instruction products stay on the serial path.  P1 / P3 are list-scheduled on fk3_w lane
    // positions by the host; a (step, position) of P3 owns one 3-word slot of the chain's c3_pb array: P2 writes the rotation
    // there, P3 adds the running position onto it in place, sites / children / the gradient pass read it from there.
    int32_t fk3;           // 1: the tables below exist (else the lean kernels are not used for this model)
    int32_t off3_site;     // [K] per site: word of its body's position | word of its body's quaternion << 16 (chain region), full program;
                           // then [naj] per joint: word of its anchor | word of its pre-joint quaternion << 16
    int32_t off3_prog;     // full program: T1 [16 (cap1 + 2)] = cap1 / 2 + 3 records of 24 words, a record per block of two steps (layout:
                           // build_fk3_program, stac_abi.hip), T2 [cap2][4] {v.x, v.y, v.z, q word | out word << 16}, T3 [cap3 * 4]: per block of
                           // four steps and position the restart entry (word of the value, or bit 31 | a valid word), then the site and joint words
    int32_t off3_root;     // same layout: the pruned program of the root passes (filled per call)
    int32_t fk3_n;         // full program: steps of P1 (even) | steps of P3 (a multiple of 4) << 8 | tasks of P2 (padded to a multiple of 32 with no-ops) << 16
    int32_t fk3_cap1, fk3_cap2, fk3_cap3;  // capacity of either program area (steps / tasks): where T2, T3 and the site words start
    int32_t c3_ql;         // [naj * 4] joint-local quaternions (w, x, y, z) by active joint; the range sums of a full trip alias it
    int32_t c3_qb;         // [naj * 4] quaternion AFTER every active joint (w, x, y, z); entry 0 = the free root's
    int32_t c3_pb;         // [(cap3 * 4 + 4) * 3] position slots by (step, position) of P3; then the root position, a sink, slack
    int32_t c3_rw0;        // [kXf] range sum of the root joint in a root fast trip (which keeps c3_ql intact)
    int32_t stride3;       // chain stride of a lean launch (odd)
    int32_t nbq, c3_bq, off3_bq;  // oriented bodies below the root body (body_quat not the identity): how many; their constant quaternions
                           // (w, x, y, z) in a chain's region (the right factors of their products in P1; written once, by the prologue);
                           // the same in the blob
    // Range sums by component (16 lanes per chain): the first rsplit of the distinct site ranges (they are sorted longest first) are
    // summed one COMPONENT {F.x .. T.z} per lane -- the same left-to-right sums, a sixth of the serial length --, the others one range
    // per lane as before.  The host picks the split that makes the two passes shortest (build_plan); 0 = every range whole.
    int32_t rsplit;
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
    // packed tables of fk_kernel (stac_abi.hip, build_fk_tables): per body 16 words {parent slot or -1 = the body before, slot to
    // park in or -1, jntadr, jntnum | pos xyz, first site | quat wxyz | end of sites, (body 0 only: qposadr of the first joint
    // visited, of the second), 0}; per joint 12 words {type, qposadr, qpos0[qposadr], qposadr of the joint visited two steps on |
    // pos xyz, 0 | axis xyz, 0}; the site ids sorted by body
    const int32_t *fk_brec, *fk_jrec, *fk_sites;
    int32_t fk_nslots;
};

struct QArgs {
    const PlanHeader *hdr;  // unused (reserved)
    const float *plan;      // device blob
    PlanHeader h;           // by-value copy (kernel argument, scalar registers)
    // problem
    const float *kp;        // [C,F,3K]
    const float *q_init;    // [C,nq] or null
    const uint8_t *masks;   // device [nkinds, nqpad] bytes: kind 0,1 = root passes, 2 = full, 3.. = parts
    const uint8_t *kpw;     // device [K] bytes: trunk mask (weights of the root passes)
    const uint8_t *kpw3;    // device [3K] per-coordinate mask for single-solve mode (or null)
    const uint8_t *kpw_sorted;  // device [K] trunk mask by sorted-site position (LM solver)
    const float *bounds;    // device [2 * nqpad] lb then ub overriding the plan's for this call (stac_q_solve), or null
    const int32_t *perm;    // device [C] or null: slot / queue position -> chain, the chains in the order of their expected length
                            // (stac_kernels.hip, root_key_kernel); results do not depend on it
    int32_t *place;         // device [kPlaceWords] or null: placement work space {arrived, crowded tickets, other tickets, ...,
                            // wavefronts per SIMD}: the wavefronts of crowded SIMDs take the short end of perm (q_phase_kernel)
    int32_t place_crowded;  // a SIMD with this many wavefronts of the launch or more counts as crowded
    int32_t C, F, P;
    int32_t root_kp_idx, do_root_opt;
    int32_t single;         // 1 = stac_q_solve mode (one solve, outputs x unblended + state)
    int32_t mb_words;       // LDS words reserved for the per-kind mask bit table (multiple of 4)
    int32_t n_mlev_root;    // micro-levels of the root-pass FK program at h.off_fkroot; 0 = none (never prune)
    int32_t n_run_root;     // steps of that program that have work (<= n_mlev_root, which is padded to an even count)
    int32_t n_root_joints;  // leading active joints that carry the root passes' coordinates
    int32_t fk3r_n;  // the pruned FK3 program at h.off3_root, packed like PlanHeader::fk3_n (0: none)
    // Root fast trips (throughput kernels).  While every live chain of a wavefront is in a root solve, only the first
    // root_fast coordinates move: the joint-local quaternions of all other joints, computed once, stay valid (the root
    // program parks their anchor / pre-joint entries in the sink), the gradient is the root joint's alone and its subtree
    // wrench sums the weighted (trunk) sites only -- the others contribute exact zeros, and a running sum that starts at
    // +0 is never -0, so leaving them out changes no bit.  0 = off (conditions: stac_abi.hip, stac_q_phase).  A chain
    // that finishes its root solves waits (ST_WAIT) until the other chains of its wavefront have, so that the root trips
    // of a wavefront are ALL fast ones and its chains start their pose solves in the same trip.
    int32_t root_fast;      // number of leading coordinates the root passes optimise (= root_dims), or 0
    int32_t free0p;         // > 0: active joint 0 is a free joint at qpos 0 .. 6 and free0p - 1 its ordinal among the quaternion
                            // joints: lanes 0 .. 6 of a group (8 lanes or more) then run its pre-pass and its gradient out of
                            // their registers, one component each, instead of one lane doing all of it (set per launch)
    uint32_t root_trunk_lo, root_trunk_hi;  // sorted-site positions inside the root joint's range that carry a weight
    // Straggler hand-off (null = off).  Chains take very different numbers of iterations; once most of a launch's
    // chains are done the rest would drag on at a few waves per CU.  ctl = {finished chains, threshold, handed-off
    // chains, capacity}: a chain of the throughput kernel that starts an iteration (state VG_Y) after `finished`
    // has reached `threshold` writes its solver state to hand[] and stops; a second launch of the latency kernel
    // (resume = 1, one chain per wavefront) picks the states up and finishes them, one trip per iteration.
    int32_t *ctl;           // {finished, threshold, handed off, capacity, next chain of the queue}
    float *hand;            // [capacity][3 * nqpad + 12]: x, y, q0, then {chain, kind, frame, iter, stepsize, t, c_iter, c_ls, c_grad, c_solves, fx, error}
    int32_t resume;
    int32_t resume_slots;   // grid size of the resume launch (= capacity)
    int32_t queue_slots;    // > 0: chain queue -- the grid covers this many chain slots (the resident ones); a group that
                            // finishes a chain takes the next unstarted one (ctl[4]) instead of leaving its slot idle
    int32_t flags;          // bit 0: do NOT fuse the x_next gradient into accepted line-search evaluations; bit 1: level-loop FK instead of the FK program; bit 2 (host only): a uniform FK program runs through the per-step dispatch; bit 3 (host only): no lean kernel (A/B switches); bit 4 (set by the host per launch): every active joint but the free root is a hinge
    float tol;
    int32_t maxiter, maxls;
    // outputs
    float *qpos_out;        // [C,F,nq]   (single: params_out [N,nq])
    float *err_out;         // [C,F]      (single: state_out [N,4])
    uint32_t *counters_out; // [C,F,4] or null
    float *q_carry_out;     // [C,nq] or null
    unsigned long long *prof;  // diagnostic builds (-DSTAC_PROFILE) only: [16] per-phase cycle sums
};

// ---- optional LM solver (stac_lm.hip): per solve-kind tables in one global-memory blob of 32-bit words -----
struct LmKind {            // 12 words, at blob[kind * 12]
    int32_t nd;            // optimised coordinates with structural support ("dofs"), ancestors first
    int32_t ne;            // structurally non-zero entries (row >= col) of J^T J
    int32_t ni;            // (site, dof) pairs with a non-zero Jacobian block
    int32_t maxpd;         // longest root path, counted in dofs (= row stride of the path-compressed matrices)
    int32_t off_dof;       // LmDof[nd]            -- offset inside the HOT section (staged in LDS)
    int32_t off_ent;       // LmEnt[ne], longest site range first   -- absolute, read from global memory
    int32_t off_item;      // LmItem[ni]                            -- absolute, read from global memory
    int32_t nquat;         // raw quaternions (free root, ball joints) of which all four components are optimised
    int32_t off_anc;       // int32[nd * maxpd]: dof index at every position of the dof's root path (-1 beyond);
                           // offset inside the HOT section
    int32_t off_quat;      // int32[nquat]: index of each such quaternion's first dof (HOT section)
    int32_t pad1, pad2;
};
constexpr int kLmKindWords = 12;
struct LmDof { int32_t qadr, joint, comp, pd; };                 // qpos index, active joint, component, path depth
struct LmEnt { int32_t row, col, pds, range; };                  // pds = pd_row | pd_col << 8; range = lo | hi << 16
struct LmItem { int32_t sitepos, dof, pd, pad; };
struct LmArgs {
    const int32_t *tab;    // device blob: LmKind[nkinds], HOT section (dof + path tables of every kind), then the rest
    int32_t nkinds;
    int32_t hot_off, hot_words;  // the part every workgroup stages in LDS (it is walked in dependent loops)
    int32_t n_max;         // max nd over kinds
    int32_t npk;           // max over kinds of nd * maxpd (path-compressed matrix), rounded up to a multiple of 4
    int32_t maxpd;         // max over kinds
    float lambda0;
    // extra per-chain LDS regions (float offsets inside the chain region, after the PG layout)
    int32_t c_qe;          // [nqpad] evaluation point (overrides PlanHeader::c_qe for this kernel)
    int32_t c_gg;          // [max(nqpad, K padded)] gradient out / per-site loss terms (overrides PlanHeader::c_gg, which
                           // the PG kernel places inside the body transforms)
    int32_t c_sx;          // [3K] site world positions by sorted-site position
    int32_t c_jp;          // [K * maxpd * 3] weighted Jacobian blocks; aliased by the factor H[npk]
    int32_t c_A;           // [npk] J^T W J (+ gauge term), row b holds its root-path columns: A[b * maxpd + pd(a)]
    int32_t c_b;           // [n_max] J^T W (kp - x)
    int32_t c_d;           // [n_max + 2 maxpd] step, then two scratch rows of the pivot loop
    int32_t c_fz;          // [n_max] 1.0 where the coordinate is frozen at a bound
    int32_t chain_stride;  // PG stride + extras (odd)
};

}  // namespace stac
