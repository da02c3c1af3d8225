// stac_abi.hip -- host side of libstac_hip.so: plan construction and the C ABI of include/stac_hip.h.
//
// Builds the "plan" (marker-ancestor subtree, level schedule, CSR site/child lists) from the flat
// model tables, keeps it in device memory and launches the kernels of stac_kernels.hip.
// No CPU fallback exists: every compute entry point needs a GPU and fails loudly without one.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/stac_hip.h"
#include "stac_plan.hpp"

namespace stac {
bool q_phase_has_variant(int G, int nq, int wpe);
bool q_phase_has_lean_variant(int G, int nq, int wpe, int spec);
bool q_phase_lean_conditions(const QArgs &a, int G);
bool q_phase_lean_holds(int G, int nq, int K);
hipError_t launch_q_phase(const QArgs &a, int G, int wpb, int wpe, int spec, size_t lds_bytes, hipStream_t s,
                          int *capacity_out, bool lean);
hipError_t launch_q_phase_lm(const QArgs &a, const LmArgs &L, int G, int wpb, size_t lds_bytes, hipStream_t s,
                             int *capacity_out);
int lm_waves_per_simd(int G, int nq);
hipError_t launch_fk(const FullModel &M, const float *qpos, int N, float *qn, float *xpos, float *xquat,
                     float *site_xpos, int normalize, hipStream_t s);
hipError_t launch_m_partial(const FullModel &M, const float *kp, const float *xpos, const float *xquat, int T,
                            float *contrib, float *partial, hipStream_t s);
hipError_t launch_m_finish(int K, const float *partial, const float *m0, const float *dreg, float lam, float *out,
                           float *err, hipStream_t s);
hipError_t launch_ctl_init(int32_t *ctl, int v0, int v1, int v2, int v3, int v4, hipStream_t s);
hipError_t launch_chain_order(const float *kp, int C, int F, int K, const float *rest_sites, const uint8_t *kpw, int root_kp_idx,
                              float rx, float ry, float rz, uint32_t *hist, uint32_t *keybits, int32_t *perm, int32_t *place,
                              hipStream_t s);
}  // namespace stac

using namespace stac;

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return fail(STAC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));       \
    } while (0)

// Developer switches: read from the environment ONCE, at stac_model_create, so that a variable
// set later cannot change the launch shape of a model in use.  -1 = not set.
struct DebugSwitches {
    int flags = -1, spec = -1, specg = -1, specr = -1, wpe = -1, wpb = -1, handoff = -1, queue = -1, stride_add = -1, rsplit = -1;
    bool noprune = false, verbose = false, nofast = false, nofree0 = false, noorder = false, nodiet = false, nolean = false;
    static int geti(const char *name) {
        const char *v = getenv(name);
        return v ? atoi(v) : -1;
    }
    void read_env() {
        flags = geti("STAC_HIP_FLAGS"); spec = geti("STAC_HIP_SPEC"); wpe = geti("STAC_HIP_WPE"); wpb = geti("STAC_HIP_WPB");
        handoff = geti("STAC_HIP_HANDOFF"); queue = geti("STAC_HIP_QUEUE"); specg = geti("STAC_HIP_SPECG"); specr = geti("STAC_HIP_SPECR"); stride_add = geti("STAC_HIP_STRIDE_ADD"); rsplit = geti("STAC_HIP_RSPLIT");
        noprune = getenv("STAC_HIP_NOPRUNE") != nullptr; nofast = getenv("STAC_HIP_NOFAST") != nullptr; nofree0 = getenv("STAC_HIP_NOFREE0") != nullptr; noorder = getenv("STAC_HIP_NOORDER") != nullptr; nolean = getenv("STAC_HIP_NOLEAN") != nullptr; nodiet = getenv("STAC_HIP_NODIET") != nullptr; verbose = getenv("STAC_HIP_VERBOSE") != nullptr;
    }
};

struct stac_model {
    int device = 0;   // the HIP device the model lives on: every entry point runs with it current (and restores)
    int cus = 256;    // its compute units (hipDeviceAttributeMultiprocessorCount: 256 on a whole MI355X; fewer on a partition or under a CU mask)
    DebugSwitches dbg;
    float *d_bounds = nullptr;          // [2 nqpad] per-call lb / ub of stac_q_solve
    std::vector<float> bounds_cache;    // what d_bounds holds
    PlanHeader h{};
    std::vector<float> blob_host;  // plan blob (host mirror)
    float *d_blob = nullptr;       // device plan blob
    // full tables on device
    int32_t *d_body_parentid = nullptr, *d_body_jntadr = nullptr, *d_body_jntnum = nullptr;
    float *d_body_pos = nullptr, *d_body_quat = nullptr;
    int32_t *d_jnt_type = nullptr, *d_jnt_qposadr = nullptr;
    float *d_jnt_pos = nullptr, *d_jnt_axis = nullptr, *d_qpos0 = nullptr;
    int32_t *d_site_bodyid = nullptr;
    int32_t *d_fk_brec = nullptr, *d_fk_jrec = nullptr, *d_fk_sites = nullptr;  // packed tables of fk_kernel (build_fk_tables)
    int fk_nslots = 0;
    float *d_site_pos = nullptr;  // [K,3] offsets for the stand-alone FK / m-phase kernels (mirrors the plan's SiteRec.pos)
    uint32_t *d_order = nullptr;  // chain order / placement work space: [3K floats rest sites][4096 buckets][kPlaceWords][C key bits][C perm]
    size_t order_chains = 0;
    uint8_t *d_masks = nullptr;  // [kMaxKinds, nqpad] + [K] + [3K]
    std::vector<uint8_t> masks_cache;  // what d_masks currently holds (uploads + their sync happen only on change)
    // host copies of the plan's structure, used to build the LM solver's per-kind tables
    std::vector<int> h_ab_parent, h_aj_type, h_aj_qadr, h_aj_slot, h_aj_slo, h_aj_shi, h_sortpos;
    // what build_fk_program needs (the root passes get a pruned program per call)
    std::vector<BodyRec> h_brec;
    std::vector<int> h_lev_adr, h_ab_jadr, h_ab_jnum, h_xf, h_site_slot;
    std::vector<float> h_aj_pos;
    int n_mlev_root = 0;        // micro-levels of the root-pass program currently in the blob (0 = none)
    int n_run_root = 0;         // of which the leading ones have work (n_mlev_root is padded to an even count)
    PlanHeader h3{};            // h with the chain layout of a lean launch (split kinematics, PlanHeader::fk3); valid when h.fk3
    int fk3r = 0;  // the pruned FK3 program currently in the blob: its counts packed like PlanHeader::fk3_n (0: none)
    std::vector<float> h_bpos;  // body_pos of the active bodies, by slot
    int32_t *d_lm_tab = nullptr;
    size_t lm_tab_words = 0;
    std::vector<int32_t> lm_tab_cache;
    LmArgs lm_args{};
    size_t masks_bytes = 0;
    int32_t *d_ctl = nullptr;    // straggler hand-off: {finished, threshold, handed off, capacity}
    float *d_hand = nullptr;     // [hand_cap][3 nqpad + 12] solver states in transit
    int hand_cap = 0;
    float *d_scratch = nullptr;  // grown on demand (xpos/xquat when the caller does not want them)
    size_t scratch_floats = 0;
    int max_depth = 0;
    FullModel full() const {
        FullModel M{};
        M.nbody = h.nbody; M.njnt = h.njnt; M.nq = h.nq; M.K = h.K;
        M.body_parentid = d_body_parentid; M.body_jntadr = d_body_jntadr; M.body_jntnum = d_body_jntnum;
        M.body_pos = d_body_pos; M.body_quat = d_body_quat;
        M.jnt_type = d_jnt_type; M.jnt_qposadr = d_jnt_qposadr;
        M.jnt_pos = d_jnt_pos; M.jnt_axis = d_jnt_axis; M.qpos0 = d_qpos0;
        M.site_bodyid = d_site_bodyid;
        M.site_pos = d_site_pos;
        M.fk_brec = d_fk_brec; M.fk_jrec = d_fk_jrec; M.fk_sites = d_fk_sites; M.fk_nslots = fk_nslots;
        return M;
    }
};

// The packed tables of fk_kernel (stac_kernels.hip; layout: FullModel in stac_plan.hpp).  A lane carries the transform of the
// body it has just computed in registers; a body whose children do not ALL follow it directly parks its transform in an LDS slot
// until its last child has read it.  Slots are handed out like registers: a slot is free again from the step of its owner's
// last child on (that step reads before it writes).
struct FkTables {
    std::vector<int32_t> brec, jrec, sites;
    int nslots = 0;
};
static FkTables build_fk_tables(const stac_model_tables *t) {
    const int nb = t->nbody, nj = t->njnt, K = t->nsite;
    FkTables ft;
    ft.brec.assign((size_t)16 * std::max(nb, 1), 0);
    ft.jrec.assign((size_t)12 * std::max(nj, 1), 0);
    auto f2i = [](float f) { int32_t i; std::memcpy(&i, &f, 4); return i; };
    std::vector<int> last_child(nb, -1), slot_of(nb, -1), owner;  // owner[slot] = body parked there
    std::vector<char> far_child(nb, 0);
    for (int b = 1; b < nb; ++b) {
        const int p = t->body_parentid[b];
        last_child[p] = b;
        if (p != b - 1) far_child[p] = 1;
    }
    for (int b = 0; b < nb; ++b) {
        int32_t *r = &ft.brec[(size_t)16 * b];
        const int p = b ? t->body_parentid[b] : -1;
        r[0] = (b && p != b - 1) ? slot_of[p] : -1;
        r[1] = -1;
        if (far_child[b]) {
            int sl = -1;
            for (int k = 0; k < (int)owner.size() && sl < 0; ++k)
                if (last_child[owner[k]] <= b) sl = k;
            if (sl < 0) { sl = (int)owner.size(); owner.push_back(b); }
            owner[sl] = b;
            slot_of[b] = sl;
            r[1] = sl;
        }
        r[2] = t->body_jntadr[b] < 0 ? 0 : t->body_jntadr[b];
        r[3] = t->body_jntnum[b];
        for (int c = 0; c < 3; ++c) r[4 + c] = f2i(t->body_pos[3 * b + c]);
        for (int c = 0; c < 4; ++c) r[8 + c] = f2i(t->body_quat[4 * b + c]);
        r[7] = (int32_t)ft.sites.size();
        for (int k = 0; k < K; ++k)
            if (t->site_bodyid[k] == b) ft.sites.push_back(k);
        r[12] = (int32_t)ft.sites.size();
    }
    ft.nslots = (int)owner.size();
    for (int j = 0; j < nj; ++j) {
        int32_t *r = &ft.jrec[(size_t)12 * j];
        r[0] = t->jnt_type[j];
        r[1] = t->jnt_qposadr[j];
        r[2] = f2i(t->qpos0[t->jnt_qposadr[j]]);
        r[3] = t->jnt_qposadr[j];
        for (int c = 0; c < 3; ++c) { r[4 + c] = f2i(t->jnt_pos[3 * j + c]); r[8 + c] = f2i(t->jnt_axis[3 * j + c]); }
    }
    // the kernel requests the first coordinate of the joint it will visit TWO steps on while it works on the current one: the order
    // of the visits is the bodies' (body 0 has none); brec[13], brec[14] of body 0 = the coordinates of the first two joints visited
    std::vector<int> visit;
    for (int b = 1; b < nb; ++b)
        for (int j = t->body_jntadr[b]; j < t->body_jntadr[b] + t->body_jntnum[b]; ++j) visit.push_back(j);
    for (size_t i = 0; i < visit.size(); ++i) {
        if (i < 2) ft.brec[13 + i] = t->jnt_qposadr[visit[i]];
        if (i + 2 < visit.size()) ft.jrec[(size_t)12 * visit[i] + 3] = t->jnt_qposadr[visit[i + 2]];
    }
    if (ft.sites.empty()) ft.sites.push_back(0);
    return ft;
}

template <typename T>
static hipError_t upload(T **dst, const T *src, size_t n) {
    hipError_t e = hipMalloc(reinterpret_cast<void **>(dst), std::max<size_t>(n, 1) * sizeof(T));
    if (e != hipSuccess) return e;
    if (n) e = hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice);
    return e;
}

// Makes the model's device current for the duration of an entry point (a caller may have switched devices since
// stac_model_create: allocations and launches must still land on the model's GPU), then restores the caller's.
// Compute units of the device the running entry point works on: the launch heuristics (resident chains, chain queue,
// placement by SIMD load) count with the device's own CUs, not with a compile-time 256 (set by DeviceGuard)
static thread_local int t_cus = 256;
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const stac_model *m) {
        if (m && hipGetDevice(&prev) == hipSuccess && prev != m->device) switched = hipSetDevice(m->device) == hipSuccess;
        if (m) t_cus = m->cus;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

// Scratch for stac_fk / stac_q_phase when the caller does not want xpos / xquat: kept at its high-water mark (grown
// by half again, so repeated slightly larger calls do not reallocate).  Growing frees the old block, which waits for
// the device -- the only such wait of the library, documented in stac_hip.h.
static int ensure_scratch(stac_model *m, size_t floats) {
    if (floats <= m->scratch_floats) return STAC_OK;
    if (m->d_scratch) (void)hipFree(m->d_scratch);
    m->d_scratch = nullptr;
    m->scratch_floats = 0;
    floats += floats / 2;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&m->d_scratch), floats * sizeof(float)));
    m->scratch_floats = floats;
    return STAC_OK;
}

extern "C" const char *stac_last_error(void) { return g_err.c_str(); }
extern "C" int32_t stac_abi_version(void) { return STAC_HIP_ABI_VERSION; }
extern "C" int32_t stac_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// The FK program of the bodies in `need` (null: all active bodies): a header (one FK_ML_* flag word per micro-level,
// then per position the ql offset of its first step) and the FkStep records at (micro_level * max_width + position in
// the level).  Positions are those of the full layout, so a child still follows its parent on one lane.
static std::vector<int32_t> build_fk_program(const stac_model *m, const char *need, int *n_mlev_out, int *uniform_out = nullptr,
                                              int ja_keep = 1 << 30,  // joints >= ja_keep park their anchor / pre-joint entry in the sink
                                              int *n_run_out = nullptr) {
    const PlanHeader &h = m->h;
    const int W = h.max_width, rw = h.fk_rec_words, nlev = h.nlev, hw = h.fk_hdr_words;
    const std::vector<int> &lev_adr = m->h_lev_adr;
    // Start step of every body: as soon as its parent is finished and its lane position is free (positions are those of
    // the level layout, so a position runs its bodies in level order and a child that takes over its parent's position
    // starts in the step after the parent's last).  A level no longer waits for its slowest body.
    std::vector<int> tstart(m->h_brec.size(), 0), tfin(m->h_brec.size(), 0), free_t(W, 0);
    int t_end = 0;
    for (int l = 0; l < nlev; ++l)
        for (int s = lev_adr[l]; s < lev_adr[l + 1]; ++s) {
            if (need && !need[s]) continue;
            const int pp = s - lev_adr[l], ps = m->h_ab_parent[s];  // parent slot + 1, 0 = world
            tstart[s] = std::max(ps ? tfin[ps - 1] : 0, free_t[pp]);
            tfin[s] = tstart[s] + std::max(1, m->h_ab_jnum[s]);
            free_t[pp] = tfin[s];
            t_end = std::max(t_end, tfin[s]);
        }
    const int n_mlev = std::max((t_end + 1) & ~1, 2);  // even: the kernel runs two steps per loop trip
    auto f2i = [](float f) { int32_t i; std::memcpy(&i, &f, 4); return i; };
    const int32_t ident_ql = h.c_bx + kXq;  // the world entry's quaternion (1, 0, 0, 0)
    std::vector<int32_t> prog((size_t)hw + (size_t)n_mlev * W * rw, 0);
    for (int ml = 0; ml < n_mlev; ++ml)
        for (int pp = 0; pp < W; ++pp) {  // a position without work: neutral data, nothing stored
            int32_t *r = prog.data() + hw + ((size_t)ml * W + pp) * rw;
            r[4] = -1; r[5] = h.c_sink; r[6] = h.c_sink; r[7] = ident_ql; r[3] = FK_KIND_PLAIN;
            if (rw == 16) r[12] = f2i(1.0f);
        }
    std::vector<int32_t> ql_of((size_t)n_mlev * W, ident_ql);  // ql offset of every (micro-level, position)
    for (int l = 0; l < nlev; ++l)
        for (int s = lev_adr[l]; s < lev_adr[l + 1]; ++s) {
            if (need && !need[s]) continue;
            const BodyRec &br = m->h_brec[s];
            const int pp = s - lev_adr[l], njs = m->h_ab_jnum[s], nsteps = std::max(1, njs), xfs = m->h_xf[s];
            for (int i = 0; i < nsteps; ++i) {
                const int ml = tstart[s] + i;
                int32_t *r = prog.data() + hw + ((size_t)ml * W + pp) * rw;
                const int fsh = 16 * (ml & 1);
                if (i == 0) {
                    prog[ml >> 1] |= FK_ML_BODY << fsh;
                    if (!(br.flags & 1)) prog[ml >> 1] |= FK_ML_BQUAT << fsh;
                    if (!(br.flags & 2)) {  // the parent's transform comes from LDS (another lane produced it, or the world)
                        r[4] = h.c_bx + kXf * br.parent;
                        prog[ml >> 1] |= FK_ML_PARENT_LDS << fsh;
                    }
                    for (int c = 0; c < 3; ++c) r[c] = f2i(br.pos[c]);
                    if (rw == 16) for (int c = 0; c < 4; ++c) r[12 + c] = f2i(br.quat[c]);
                }
                if (i < njs) {
                    const int j = m->h_ab_jadr[s] + i, ty = m->h_aj_type[j];
                    const float *jp = m->h_aj_pos.data() + 3 * j;
                    for (int c = 0; c < 3; ++c) r[8 + c] = f2i(jp[c]);
                    r[5] = j < ja_keep ? h.c_ja + kXf * j : h.c_sink;
                    const bool free_as_parent = ty == STAC_JNT_FREE && i == 0 && br.parent == 0 && (br.flags & 1);
                    if (!free_as_parent) {
                        ql_of[(size_t)ml * W + pp] = h.c_ja + kXf * j + kXq;
                        prog[ml >> 1] |= FK_ML_JOINT << fsh;
                        if (jp[0] != 0.0f || jp[1] != 0.0f || jp[2] != 0.0f) prog[ml >> 1] |= FK_ML_JPOS << fsh;
                    }
                    if (free_as_parent) {
                        // A free joint on a top-level body without orientation is a plain step on a synthetic parent: the
                        // pre-pass leaves {qpos[0:3], normalised quaternion} in the joint's own entry, and with body_pos = 0,
                        // an identity joint quaternion and jnt_pos = 0 the step reproduces exactly that transform
                        // (anchor = position; the free joint's gradient does not look at the pre-joint quaternion).
                        r[4] = h.c_ja + kXf * j;
                        prog[ml >> 1] |= FK_ML_PARENT_LDS << fsh;
                        for (int c = 0; c < 3; ++c) { r[c] = 0; r[8 + c] = 0; }
                    } else if (ty == STAC_JNT_FREE) { r[3] = FK_KIND_FREE; r[11] = m->h_aj_qadr[j]; prog[ml >> 1] |= FK_ML_SPECIAL << fsh; }
                    if (ty == STAC_JNT_SLIDE) { r[3] = FK_KIND_SLIDE; r[11] = j; prog[ml >> 1] |= (FK_ML_SPECIAL | FK_ML_JPOS) << fsh; }
                }
                if (i == nsteps - 1 && xfs >= 0) r[6] = h.c_bx + kXf * xfs;
            }
        }
    for (int ml = 0; ml + 1 < n_mlev; ++ml)
        for (int pp = 0; pp < W; ++pp) prog[hw + ((size_t)ml * W + pp) * rw + 7] = ql_of[(size_t)(ml + 1) * W + pp];
    for (int pp = 0; pp < W; ++pp) prog[(h.n_mlev_hdr >> 1) + pp] = ql_of[pp];  // first steps: fetched by the prologue
    bool uniform = true;
    for (int ml = 0; ml < n_mlev; ++ml) {  // step form (FK_FORM_*) next to the flags
        const int fsh = 16 * (ml & 1), fl = (prog[ml >> 1] >> fsh) & 255;
        // hinge / ball steps on neutral data where a part does not apply (body_pos = 0, jnt_pos = 0, identity joint
        // quaternion: exact no-ops; a position without a joint stores to the sink); free and slide joints and oriented
        // bodies take the general step
        int form = FK_FORM_GENERAL;
        if (!(fl & (FK_ML_SPECIAL | FK_ML_BQUAT))) form = (fl & FK_ML_PARENT_LDS) ? FK_FORM_PARENT_BODY_JOINT : FK_FORM_BODY_JOINT;
        if (form == FK_FORM_GENERAL) uniform = false;
        prog[ml >> 1] |= form << (fsh + 8);
    }
    if (m->dbg.verbose) {
        fprintf(stderr, "[stac] FK program%s: %d steps, forms", need ? " (root passes)" : "", n_mlev);
        for (int ml = 0; ml < n_mlev; ++ml) fprintf(stderr, " %d(%x)", (prog[ml >> 1] >> (16 * (ml & 1) + 8)) & 255, (prog[ml >> 1] >> (16 * (ml & 1))) & 255);
        fprintf(stderr, "\n");
    }
    *n_mlev_out = n_mlev;
    if (n_run_out) *n_run_out = std::max(t_end, 1);  // steps with work (the records are padded to an even count)
    if (uniform_out) *uniform_out = uniform ? 1 : 0;
    return prog;
}

// ---- split kinematics (PlanHeader::fk3): the three tables of a program ---------------------------------------------------------
// P1 runs one quaternion product per active hinge, P3 one addition per rotation that is not identically zero; both are list-
// scheduled (longest remaining path first) on four lane positions.  A position continues with the successor of its last
// operation where there is one -- the running value stays in its registers --, any other operation "restarts" from the LDS
// entry of its predecessor.  Returns false if the program does not fit (more than 32 steps, offsets beyond 16 bits).
struct Fk3Program {
    std::vector<int32_t> words;   // T1 [16 (cap1 + 2)] (cap1 / 2 + 3 records of 24 words), T2 [cap2][4], T3 [cap3 * 4] (cap3 + 12 used), site words [K], joint words [naj]
    int n1 = 0, n2 = 0, n3 = 0;
};
struct Fk3Op { int prev; int height; int t = -1, pp = -1; bool restart = false; };
// B, D: an operation may restart only in a step t with t % B == 0, and only from a predecessor finished in a step <= t - D (the
// kernel requests the restart value that far ahead: fk3_p3); a restart from the root position (prev < 0) is always possible.
static int fk3_schedule(std::vector<Fk3Op> &ops, const int B, const int D) {  // returns the number of steps
    const int n = (int)ops.size(), W = 4;
    for (int i = 0; i < n; ++i) ops[i].height = 1;
    for (bool changed = true; changed;) {  // (copies made by fk3_unshare stand behind their successors: no index order to rely on)
        changed = false;
        for (int i = 0; i < n; ++i) {
            const int p = ops[i].prev;
            if (p >= 0 && ops[p].height < ops[i].height + 1) { ops[p].height = ops[i].height + 1; changed = true; }
        }
    }
    std::vector<int> last(W, -2);  // last operation of every position (-2: none yet)
    int done = 0, t = 0;
    for (; done < n; ++t) {
        std::vector<char> taken(W, 0);
        for (int pp = 0; pp < W; ++pp) {  // continue where a successor exists
            if (last[pp] < 0) continue;
            int best = -1;
            for (int i = 0; i < n; ++i)
                if (ops[i].t < 0 && ops[i].prev == last[pp] && (best < 0 || ops[i].height > ops[best].height)) best = i;
            if (best >= 0) { ops[best].t = t; ops[best].pp = pp; taken[pp] = 1; last[pp] = best; ++done; }
        }
        for (int pp = 0; pp < W && t % B == 0; ++pp) {  // free positions take the most urgent operation whose predecessor is finished
            if (taken[pp]) continue;
            int best = -1;
            for (int i = 0; i < n; ++i)
                if (ops[i].t < 0 && (ops[i].prev < 0 || (ops[ops[i].prev].t >= 0 && ops[ops[i].prev].t <= t - D)) &&
                    (best < 0 || ops[i].height > ops[best].height))
                    best = i;
            if (best >= 0) { ops[best].t = t; ops[best].pp = pp; ops[best].restart = true; taken[pp] = 1; last[pp] = best; ++done; }
        }
        if (t > 4 * n + 8) return -1;
    }
    return t;
}
// A restart waits for its source (fk3_schedule: D steps, then the next block), so a branch close to the root would hold its whole
// subtree back.  An operation at depth <= L with several successors therefore gives every successor but the highest its own copy of
// the chain from the root (same operands, same order: the same bits, computed once more by a position that would idle anyway).
// copy(i) appends a copy of operation i's payload and is called once per copied operation, in chain order.
template <class Copy>
static void fk3_unshare(std::vector<Fk3Op> &ops, const int L, Copy &&copy) {
    const int n = (int)ops.size();
    std::vector<int> depth(n, 0), height(n, 1);
    for (int i = 0; i < n; ++i) depth[i] = (ops[i].prev >= 0 ? depth[ops[i].prev] : 0) + 1;  // (prev < i here)
    for (int i = n - 1; i >= 0; --i)
        if (ops[i].prev >= 0) height[ops[i].prev] = std::max(height[ops[i].prev], height[i] + 1);
    for (int i = 0; i < n; ++i) {
        if (depth[i] > L) continue;
        std::vector<int> kids;
        for (int k = i + 1; k < n; ++k)
            if (ops[k].prev == i) kids.push_back(k);
        if (kids.size() < 2) continue;
        std::stable_sort(kids.begin(), kids.end(), [&](int x, int y) { return height[x] > height[y]; });
        std::vector<int> chain;
        for (int j = i; j >= 0; j = ops[j].prev) chain.push_back(j);
        for (size_t c = 1; c < kids.size(); ++c) {
            int last = -1;
            for (int q = (int)chain.size() - 1; q >= 0; --q) {
                ops.push_back(Fk3Op{last, 0});
                copy(chain[q]);
                last = (int)ops.size() - 1;
            }
            ops[kids[c]].prev = last;
        }
    }
}
static bool build_fk3_program(const stac_model *m, const PlanHeader &h3, const char *need, int cap1, int cap2, int cap3, Fk3Program &out) {
    const int nab = h3.nab, naj = h3.naj, K = h3.K;
    auto f2i = [](float f) { int32_t i; std::memcpy(&i, &f, 4); return i; };
    auto nz3 = [](const float *v) { return v[0] != 0.0f || v[1] != 0.0f || v[2] != 0.0f; };
    // Quaternion nodes (entries of the chain's c3_qb array): node j = the quaternion AFTER active joint j (node 0: the free root's, from the
    // pre-pass), node naj = the sink of idle positions, node naj + 1 + i = the quaternion of the i-th ORIENTED body (body_quat not the
    // identity, below the root body) before its joints: q_parent * body_quat, the oracle's own order (oracle/stac_oracle.c: fk_chain).
    // fin[s]: node of body s after its last joint (else its oriented-body node, else its parent's); pre[j]: node in front of joint j.
    std::vector<int> fin(nab, 0), pre(naj, 0), bnode(nab, -1), bpre(nab, 0);
    int nbq = 0;
    for (int s = 0; s < nab; ++s) {
        const int ps = m->h_ab_parent[s] - 1;
        int cur = ps >= 0 ? fin[ps] : 0;
        if (s != 0 && !(m->h_brec[s].flags & 1)) { bpre[s] = cur; bnode[s] = naj + 1 + nbq; cur = bnode[s]; ++nbq; }
        for (int i = 0; i < m->h_ab_jnum[s]; ++i) { const int j = m->h_ab_jadr[s] + i; pre[j] = cur; cur = j; }
        fin[s] = cur;
    }
    if (nbq != h3.nbq) return false;  // (the layout was made for the model's own count: build_plan)
    // P1: one product per needed hinge, and one per needed oriented body (its factor: the constant body_quat, which every chain region
    // holds in c3_bq -- written once per launch, q_phase_kernel's prologue)
    struct Prod { int qlw, node, prenode; };  // word of the right factor, node of the result, node of the left factor
    std::vector<Fk3Op> o1;
    std::vector<Prod> p_of1;
    std::vector<int> op_of_node(naj + 1 + nbq, -1);
    for (int s = 0; s < nab; ++s) {
        if (need && !need[s]) continue;
        if (bnode[s] >= 0) {
            op_of_node[bnode[s]] = (int)o1.size();
            o1.push_back(Fk3Op{bpre[s] == 0 ? -1 : op_of_node[bpre[s]], 0});
            p_of1.push_back(Prod{h3.c3_bq + 4 * (bnode[s] - naj - 1), bnode[s], bpre[s]});
        }
        for (int i = 0; i < m->h_ab_jnum[s]; ++i) {
            const int j = m->h_ab_jadr[s] + i;
            if (j == 0) continue;  // the free root: its quaternion comes from the pre-pass
            op_of_node[j] = (int)o1.size();
            o1.push_back(Fk3Op{pre[j] == 0 ? -1 : op_of_node[pre[j]], 0});
            p_of1.push_back(Prod{h3.c3_ql + 4 * j, j, pre[j]});
        }
    }
    // P3: the additions, in the order of the step program (body_pos, then per joint anchor and position)
    struct Rot { int qnode; float v[3]; };
    std::vector<Fk3Op> o3;
    std::vector<Rot> rot;  // one per P3 operation
    std::vector<int> body_op(nab, -1), anchor_op(naj, -1);
    for (int s = 0; s < nab; ++s) {
        if (need && !need[s]) continue;
        const int ps = m->h_ab_parent[s] - 1;
        int cur = ps >= 0 ? body_op[ps] : -1;
        const float *bp = m->h_bpos.data() + 3 * s;
        if (s != 0 && nz3(bp)) {
            o3.push_back(Fk3Op{cur, 0});
            rot.push_back(Rot{ps >= 0 ? fin[ps] : 0, {bp[0], bp[1], bp[2]}});
            cur = (int)o3.size() - 1;
        }
        for (int i = 0; i < m->h_ab_jnum[s]; ++i) {
            const int j = m->h_ab_jadr[s] + i;
            if (j == 0) { cur = -1; anchor_op[j] = -1; continue; }  // free root: position = qpos[0:3]
            const float *jp = m->h_aj_pos.data() + 3 * j;
            if (nz3(jp)) {
                o3.push_back(Fk3Op{cur, 0});
                rot.push_back(Rot{pre[j], {jp[0], jp[1], jp[2]}});        // anchor = rotate(jpos, prequat) + pos
                cur = (int)o3.size() - 1;
                anchor_op[j] = cur;
                o3.push_back(Fk3Op{cur, 0});
                rot.push_back(Rot{j, {-jp[0], -jp[1], -jp[2]}});          // pos = anchor - rotate(jpos, quat): rotate is odd in v
                cur = (int)o3.size() - 1;
            } else {
                anchor_op[j] = cur;
            }
        }
        body_op[s] = cur;
    }
    fk3_unshare(o1, 2, [&](int i) { p_of1.push_back(p_of1[i]); });
    fk3_unshare(o3, 4, [&](int i) { rot.push_back(rot[i]); });
    int n1 = fk3_schedule(o1, 2, 3);
    if (n1 > 0) n1 = (n1 + 1) & ~1;  // P1 runs whole blocks of two steps
    int n3 = fk3_schedule(o3, 4, 5);
    if (n3 > 0) n3 = (n3 + 3) & ~3;  // P3 runs whole blocks of four steps
    const int n2 = ((int)o3.size() + 31) & ~31;
    if (n1 < 0 || n3 < 0 || n1 > 254 || n3 > 252 || n1 > cap1 || n3 > cap3 || n2 > cap2) return false;  // (eight bits each in PlanHeader::fk3_n)
    const int root_slot = cap3 * 4, sink_slot = cap3 * 4 + 1;
    auto pbw = [&](int op) { return h3.c3_pb + 3 * (op < 0 ? root_slot : o3[op].t * 4 + o3[op].pp); };
    auto qbw = [&](int node) { return h3.c3_qb + 4 * node; };
    if (h3.stride3 >= 65536) return false;
    out.n1 = std::max(n1, 0); out.n2 = n2; out.n3 = n3;
    // T1, blocks of two steps, 24 words per record: [24 r + 4 pp] = {ql word of step 2 r, of step 2 r + 1, out word of step 2 r - 2, of
    // 2 r - 1}, [24 r + 16 + 2 pp] = {restart entry of block r - 1, of block r} -- record r + 1 is what block r works from (its out words,
    // its restart flag, the next block's ql and restart words), record 0 the prologue.  A restart entry: the word of the quaternion the
    // position starts the block from, or bit 31 | a valid word if it keeps its running value.  cap1 / 2 + 3 records (P1 fetches one
    // record ahead and runs pairs of blocks); an idle step multiplies by any quaternion into the sink.
    out.words.assign((size_t)16 * (cap1 + 2) + 4 * (size_t)cap2 + 4 * (size_t)cap3 + K + naj, 0);
    int32_t *T1 = out.words.data(), *T2 = T1 + 16 * (cap1 + 2), *T3 = T2 + 4 * cap2, *SW = T3 + 4 * cap3, *JW = SW + K;
    const int nrec1 = cap1 / 2 + 3;
    if (24 * nrec1 > 16 * (cap1 + 2)) return false;  // (cap1 >= 10: build_plan)
    for (int r = 0; r < nrec1; ++r)
        for (int pp = 0; pp < 4; ++pp) {
            int32_t *w = T1 + 24 * r + 4 * pp, *e = T1 + 24 * r + 16 + 2 * pp;
            w[0] = w[1] = h3.c3_ql; w[2] = w[3] = qbw(naj);
            e[0] = e[1] = (int32_t)(0x80000000u | (uint32_t)qbw(0));
        }
    for (size_t i = 0; i < o1.size(); ++i) {
        const Prod &pr = p_of1[i];
        const int blk = o1[i].t >> 1, k = o1[i].t & 1, pp = o1[i].pp;
        T1[24 * blk + 4 * pp + k] = pr.qlw;
        T1[24 * (blk + 1) + 4 * pp + 2 + k] = qbw(pr.node);
        if (o1[i].restart) {  // (only in the first step of a block: fk3_schedule)
            T1[24 * blk + 16 + 2 * pp + 1] = qbw(pr.prenode);
            T1[24 * (blk + 1) + 16 + 2 * pp] = qbw(pr.prenode);
        }
    }
    for (int i = 0; i < cap2; ++i) { T2[4 * i + 3] = qbw(0) | (h3.c3_pb + 3 * sink_slot) << 16; }  // no-op: rotate(0, root quaternion) into the sink
    for (size_t i = 0; i < o3.size(); ++i) {
        int32_t *r = T2 + 4 * i;
        for (int c = 0; c < 3; ++c) r[c] = f2i(rot[i].v[c]);
        r[3] = qbw(rot[i].qnode) | pbw((int)i) << 16;
    }
    // T3[4 b + pp]: what position pp starts block b (steps 4 b .. 4 b + 3) from -- the word of a restart value, or bit 31 | any valid
    // word if it keeps its running value; cap3 / 4 + 3 blocks (P3 requests three blocks ahead) in the table's 4 cap3 words
    for (int i = 0; i < 4 * cap3; ++i) T3[i] = (int32_t)(0x80000000u | (uint32_t)pbw(-1));
    for (size_t i = 0; i < o3.size(); ++i)
        if (o3[i].restart) T3[o3[i].t + o3[i].pp] = pbw(o3[i].prev);  // (t % 4 == 0: entry 4 (t / 4) + pp)
    for (int k = 0; k < K; ++k) {
        const int sl = m->h_site_slot[k];
        if (need && !need[sl]) { SW[k] = pbw(-1) | qbw(0) << 16; continue; }  // (a site the root passes do not weigh: any valid entry)
        SW[k] = pbw(body_op[sl]) | qbw(fin[sl]) << 16;
    }
    // per joint: word of its anchor | word of its pre-joint quaternion << 16 (the gradient pass; full program only)
    for (int j = 0; j < naj; ++j) JW[j] = need ? 0 : (pbw(anchor_op[j]) | qbw(pre[j]) << 16);
    if (m->dbg.verbose)
        fprintf(stderr, "[stac] FK3 program%s: %d products in %d steps, %d rotations, %d additions in %d steps\n", need ? " (root passes)" : "",
                (int)o1.size(), n1, (int)o3.size(), (int)o3.size(), n3);
    return true;
}

// ------------------------------------------------------------------------------------------------
// plan construction
// ------------------------------------------------------------------------------------------------
static int build_plan(stac_model *m, const stac_model_tables *t) {
    const int nb = t->nbody, nj = t->njnt, nq = t->nq, K = t->nsite;
    if (nb < 2 || nq < 1 || K < 1) return fail(STAC_ERR_INVALID, "model needs >= 1 body, qpos and fit site");
    std::vector<int> depth(nb, 0), active(nb, 0);
    for (int b = 1; b < nb; ++b) {
        const int p = t->body_parentid[b];
        if (p < 0 || p >= b) return fail(STAC_ERR_INVALID, "bodies must be in depth-first order (parent id < body id)");
        depth[b] = depth[p] + 1;
    }
    for (int k = 0; k < K; ++k) {
        const int sb = t->site_bodyid[k];
        if (sb < 0 || sb >= nb) return fail(STAC_ERR_INVALID, "site_bodyid out of range");
        for (int b = sb; b > 0 && !active[b]; b = t->body_parentid[b]) active[b] = 1;
    }
    // Active bodies level by level.  Inside a level a body takes its parent's position wherever that position
    // exists in its own level (the kernel lane that finished the parent then continues with the child and
    // keeps the transform in registers); the deepest subtrees choose first, so long limbs stay on one lane.
    std::vector<int> height(nb, 0);
    for (int b = nb - 1; b >= 1; --b)
        if (active[b]) height[t->body_parentid[b]] = std::max(height[t->body_parentid[b]], height[b] + 1);
    int max_depth_active = 0;
    for (int b = 1; b < nb; ++b)
        if (active[b]) max_depth_active = std::max(max_depth_active, depth[b]);
    std::vector<int> slots, pos_in_level(nb, -1);
    for (int d = 1; d <= max_depth_active; ++d) {
        std::vector<int> bodies;
        for (int b = 1; b < nb; ++b)
            if (active[b] && depth[b] == d) bodies.push_back(b);
        std::stable_sort(bodies.begin(), bodies.end(), [&](int a, int b) { return height[a] > height[b]; });
        const int w = (int)bodies.size();
        std::vector<int> at(w, -1);
        std::vector<char> placed(w, 0);
        for (int i = 0; i < w; ++i) {  // deepest first: inherit the parent's position if it is free
            const int pp = pos_in_level[t->body_parentid[bodies[i]]];
            if (pp >= 0 && pp < w && at[pp] < 0) { at[pp] = bodies[i]; placed[i] = 1; }
        }
        int free_pos = 0;
        for (int i = 0; i < w; ++i) {
            if (placed[i]) continue;
            while (at[free_pos] >= 0) ++free_pos;
            at[free_pos] = bodies[i];
        }
        for (int i = 0; i < w; ++i) { pos_in_level[at[i]] = i; slots.push_back(at[i]); }
    }
    const int nab = (int)slots.size();
    if (nab == 0) return fail(STAC_ERR_INVALID, "all fit sites are attached to the world body");
    std::vector<int> slot_of(nb, -1);
    for (int s = 0; s < nab; ++s) slot_of[slots[s]] = s;
    // levels
    std::vector<int> lev_adr;
    for (int s = 0; s < nab; ++s)
        if (s == 0 || depth[slots[s]] != depth[slots[s - 1]]) lev_adr.push_back(s);
    const int nlev = (int)lev_adr.size();
    lev_adr.push_back(nab);
    // contiguous depth levels are required (an active body's parent is active, one level up)
    // joints
    std::vector<int> ab_jadr(nab), ab_jnum(nab), aj_type, aj_qadr, aj_slot;
    std::vector<float> aj_pos, aj_axis, aj_q0;
    int has_ball = 0;
    for (int s = 0; s < nab; ++s) {
        const int b = slots[s];
        ab_jadr[s] = (int)aj_type.size();
        ab_jnum[s] = t->body_jntnum[b];
        for (int j = t->body_jntadr[b]; j < t->body_jntadr[b] + t->body_jntnum[b]; ++j) {
            const int ty = t->jnt_type[j];
            if (ty < 0 || ty > 3) return fail(STAC_ERR_INVALID, "unknown joint type");
            if (ty == STAC_JNT_BALL) has_ball = 1;
            aj_type.push_back(ty);
            aj_qadr.push_back(t->jnt_qposadr[j]);
            aj_slot.push_back(s);
            for (int i = 0; i < 3; ++i) { aj_pos.push_back(t->jnt_pos[3 * j + i]); aj_axis.push_back(t->jnt_axis[3 * j + i]); }
            aj_q0.push_back((ty == STAC_JNT_HINGE || ty == STAC_JNT_SLIDE) ? t->qpos0[t->jnt_qposadr[j]] : 0.0f);
        }
    }
    const int naj = (int)aj_type.size();
    // sites sorted by (body id, site id); per joint the range of sorted positions inside its body's subtree
    std::vector<int> ab_parent(nab), sortpos(K), aj_slo(naj), aj_shi(naj), sorted;
    for (int b = 0; b < nb; ++b)
        for (int k = 0; k < K; ++k)
            if (t->site_bodyid[k] == b) { sortpos[k] = (int)sorted.size(); sorted.push_back(k); }
    std::vector<int> sub_end(nb);
    for (int b = 0; b < nb; ++b) sub_end[b] = b;
    for (int b = nb - 1; b >= 1; --b) sub_end[t->body_parentid[b]] = std::max(sub_end[t->body_parentid[b]], sub_end[b]);
    for (int s = 0; s < nab; ++s) {
        const int b = slots[s];
        const int p = t->body_parentid[b];
        ab_parent[s] = p == 0 ? 0 : slot_of[p] + 1;
        int lo = 0;
        while (lo < K && t->site_bodyid[sorted[lo]] < b) ++lo;
        int hi = lo;
        while (hi < K && t->site_bodyid[sorted[hi]] <= sub_end[b]) ++hi;
        for (int j = ab_jadr[s]; j < ab_jadr[s] + ab_jnum[s]; ++j) { aj_slo[j] = lo; aj_shi[j] = hi; }
    }
    // quaternion addresses of the whole model
    std::vector<int> quat_adr;
    for (int j = 0; j < nj; ++j) {
        if (t->jnt_type[j] == STAC_JNT_FREE) quat_adr.push_back(t->jnt_qposadr[j] + 3);
        if (t->jnt_type[j] == STAC_JNT_BALL) quat_adr.push_back(t->jnt_qposadr[j]);
    }

    PlanHeader &h = m->h;
    h.nbody = nb; h.njnt = nj; h.nq = nq; h.K = K;
    h.nab = nab; h.naj = naj; h.nlev = nlev; h.nquat = (int)quat_adr.size();
    h.nqpad = (nq + 3) & ~3;
    h.has_ball = has_ball;
    m->max_depth = nlev;
    h.max_width = 0;
    for (int l = 0; l < nlev; ++l) h.max_width = std::max(h.max_width, lev_adr[l + 1] - lev_adr[l]);
    m->h_ab_parent = ab_parent; m->h_aj_type = aj_type; m->h_aj_qadr = aj_qadr; m->h_aj_slot = aj_slot;
    m->h_aj_slo = aj_slo; m->h_aj_shi = aj_shi; m->h_sortpos = sortpos;
    m->h_site_slot.resize(K);
    for (int k = 0; k < K; ++k) m->h_site_slot[k] = slot_of[t->site_bodyid[k]];
    if (nab >= 65535 || K >= 65535) return fail(STAC_ERR_CAPACITY, "too many bodies / sites");
    if (K > 4096) return fail(STAC_ERR_CAPACITY, "more than 4096 fit sites");

    std::vector<float> &B = m->blob_host;
    B.clear();
    auto align4 = [&]() { while (B.size() & 3) B.push_back(0.f); };
    auto put_raw = [&](const void *src, size_t words) {
        align4();
        int off = (int)B.size();
        B.resize(B.size() + words);
        std::memcpy(B.data() + off, src, words * 4);
        align4();
        return off;
    };
    auto put_fpad = [&](const float *src, int n, int padded) {
        std::vector<float> v(src, src + n);
        v.resize(padded, 0.f);
        return put_raw(v.data(), v.size());
    };
    // Which body transforms go to LDS: those a site or a child on ANOTHER lane reads back (a child at its parent's
    // position continues from the lane's registers).  Levels wider than the narrowest lane group (4) can take
    // several passes per level, where nothing is carried: then every transform is stored.
    int max_width0 = 0;
    for (int l = 0; l < nlev; ++l) max_width0 = std::max(max_width0, lev_adr[l + 1] - lev_adr[l]);
    std::vector<int> stored(nab, max_width0 > 4 ? 1 : 0), xf(nab, -1);
    stored[0] = 1;  // moments are taken about the first active body
    for (int k = 0; k < K; ++k) stored[slot_of[t->site_bodyid[k]]] = 1;
    for (int s = 0; s < nab; ++s) {
        const int b = slots[s], pb = t->body_parentid[b];
        if (pb > 0 && pos_in_level[pb] != pos_in_level[b]) stored[slot_of[pb]] = 1;
    }
    int nst = 0;
    for (int s = 0; s < nab; ++s)
        if (stored[s]) xf[s] = ++nst;  // transform index; 0 = world
    std::vector<BodyRec> brec(nab);
    for (int s = 0; s < nab; ++s) {
        const int b = slots[s];
        BodyRec &r = brec[s];
        const int pb = t->body_parentid[b];
        r.parent = pb == 0 ? 0 : std::max(xf[slot_of[pb]], 0);  // transform index (only read when the parent is stored)
        r.jadr = ab_jadr[s]; r.jnum = ab_jnum[s];
        const float *q = t->body_quat + 4 * b;
        r.flags = (q[0] == 1.0f && q[1] == 0.0f && q[2] == 0.0f && q[3] == 0.0f) ? 1 : 0;
        if (pb > 0 && pos_in_level[pb] == pos_in_level[b]) r.flags |= 2;  // parent transform is in the lane's registers
        r.flags |= (xf[s] < 0 ? 0xFFFF : xf[s]) << 16;                    // where this body's transform is stored
        for (int i = 0; i < 3; ++i) r.pos[i] = t->body_pos[3 * b + i];
        r.jzero = 0;
        for (int jj = 0; jj < ab_jnum[s] && jj < 31; ++jj) {
            const float *jp = aj_pos.data() + 3 * (ab_jadr[s] + jj);
            if (jp[0] == 0.0f && jp[1] == 0.0f && jp[2] == 0.0f) r.jzero |= (1 << jj);
        }
        for (int i = 0; i < 4; ++i) r.quat[i] = q[i];
    }
    // distinct site ranges, longest first (the lanes of a round then run loops of similar length)
    std::vector<RangeRec> ranges;
    std::vector<int> aj_rid(naj, 0);
    {
        std::vector<std::pair<int, int>> uniq;
        for (int j = 0; j < naj; ++j) {
            const std::pair<int, int> r{aj_slo[j], aj_shi[j]};
            if (std::find(uniq.begin(), uniq.end(), r) == uniq.end()) uniq.push_back(r);
        }
        std::stable_sort(uniq.begin(), uniq.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) {
            return a.second - a.first > b.second - b.first;
        });
        for (const auto &r : uniq) ranges.push_back(RangeRec{r.first, r.second});
        for (int j = 0; j < naj; ++j)
            aj_rid[j] = (int)(std::find(uniq.begin(), uniq.end(), std::pair<int, int>{aj_slo[j], aj_shi[j]}) - uniq.begin());
        if (ranges.empty()) ranges.push_back(RangeRec{0, 0});
    }
    h.nrange = (int)ranges.size();
    {   // PlanHeader::rsplit: how many of the longest ranges the 16-lane kernels sum one component per lane.  Cost of a choice, in
        // instructions of the wavefront (every round of lanes is as long as its longest loop): a site of a component task is one read and
        // one addition, of a whole range six of each; a round has a fixed part (its record, the stores, the loop).
        auto cost = [&](int ns) {
            const double c1 = 2.5, c6 = 13.0, fixed = 15.0;
            double c = 0.0;
            for (int t0 = 0; t0 < 6 * ns; t0 += 16) c += fixed + c1 * (ranges[t0 / 6].hi - ranges[t0 / 6].lo);  // (sorted: a round's first task is its longest)
            for (int r0 = ns; r0 < h.nrange; r0 += 16) c += fixed + c6 * (ranges[r0].hi - ranges[r0].lo);
            return c;
        };
        int best = 0;
        for (int ns = 1; ns <= h.nrange; ++ns)
            if (cost(ns) < cost(best)) best = ns;
        h.rsplit = m->dbg.rsplit >= 0 ? std::min(m->dbg.rsplit, h.nrange) : best;
    }
    std::vector<JointRec> jrec(std::max(naj, 1));
    int nqj = 0;
    for (int j = 0; j < naj; ++j) {
        JointRec &r = jrec[j];
        r.type = aj_type[j]; r.qadr = aj_qadr[j]; r.slo = aj_slo[j]; r.shi = aj_shi[j];
        for (int i = 0; i < 3; ++i) { r.pos[i] = aj_pos[3 * j + i]; r.axis[i] = aj_axis[3 * j + i]; }
        r.q0 = aj_q0[j]; r.rid = aj_rid[j];
        if (aj_type[j] == STAC_JNT_FREE || aj_type[j] == STAC_JNT_BALL) {  // no reference angle: the ordinal among the
            const int32_t qi = nqj++;                                       // quaternion joints (saved-quaternion slot)
            std::memcpy(&r.q0, &qi, 4);
        }
    }
    std::vector<SiteRec> srec(K);
    for (int k = 0; k < K; ++k) {
        for (int i = 0; i < 3; ++i) srec[k].pos[i] = t->site_pos[3 * k + i];
        srec[k].slot_sortpos = xf[slot_of[t->site_bodyid[k]]] | (sortpos[k] << 16);
    }
    h.off_lev_adr = put_raw(lev_adr.data(), lev_adr.size());
    h.off_body = put_raw(brec.data(), brec.size() * sizeof(BodyRec) / 4);
    {   // momentum table (stac_plan.hpp, kTTab): the kernel's expressions, float32, no contraction (build flags)
        std::vector<float> tt(2 * kTTab);
        float tk = 1.0f;
        for (int k = 0; k < kTTab; ++k) {
            const float tn = 0.5f * (1.0f + std::sqrt(1.0f + 4.0f * tk * tk));
            tt[2 * k] = tn;
            tt[2 * k + 1] = (tk - 1.0f) / tn;
            tk = tn;
        }
        (void)put_raw(tt.data(), tt.size());
    }
    h.off_joint = put_raw(jrec.data(), jrec.size() * sizeof(JointRec) / 4);
    h.off_site = put_raw(srec.data(), srec.size() * sizeof(SiteRec) / 4);
    h.off_range = put_raw(ranges.data(), ranges.size() * sizeof(RangeRec) / 4);
    h.off_lb = put_fpad(t->lb, nq, h.nqpad);
    h.off_ub = put_fpad(t->ub, nq, h.nqpad);
    h.off_qpos0 = put_fpad(t->qpos0, nq, h.nqpad);
    quat_adr.push_back(0);
    h.off_quat_adr = put_raw(quat_adr.data(), quat_adr.size());
    {
        std::vector<int32_t> act(h.nqpad, 0);
        for (int j = 0; j < naj; ++j) {
            const int dims = aj_type[j] == STAC_JNT_FREE ? 7 : (aj_type[j] == STAC_JNT_BALL ? 4 : 1);
            for (int c = 0; c < dims; ++c) act[aj_qadr[j] + c] = 1;
        }
        h.off_active = put_raw(act.data(), act.size());
    }
    h.core_words = (int)B.size();  // what a kernel that walks the levels itself stages in LDS
    // per-chain LDS layout
    int o = 0;
    h.nst = nst;
    h.nqj = nqj;
    h.c_bx = o; o += (nst + 1) * kXf;
    h.c_ja = o; o += naj * kXf;
    h.c_jn = o; o += std::max(nqj, 1);
    h.c_qsv = o; o += 4 * nqj;
    o = (o + 3) & ~3;
    h.c_sw = o; o += std::max(K * kXf, h.nqpad + kXf);  // (+ kXf: the tail entry is the FK program's store sink, c_sink)
    h.c_sink = o - kXf;
    o = (o + 3) & ~3;
    h.kpow2 = 1;
    while (h.kpow2 < K) h.kpow2 <<= 1;
    // Joint pass: (A) the distinct range sums read the site wrenches and go where the body transforms were (dead once the
    // site pass is over; the root position the moments refer to is read before); (B) the joints read those sums and write
    // the gradient where the site wrenches were.  Own region for the sums if they do not fit.
    h.c_gg = h.c_sw;
    if ((nst + 1) * kXf >= h.nrange * kXf) {
        h.c_rw = h.c_bx;
    } else {
        h.c_rw = o; o += h.nrange * kXf;
    }
    h.c_qe = h.c_sw;  // the evaluation point is dead once the site pass writes the wrenches (the LM kernel, which reads it
                      // later in the trip, moves it into its own region)
    // strides of 4 x odd words: regions stay 16-byte aligned (ds_read_b128 / ds_write_b128) and the chains of one
    // wavefront start in different banks
    auto stride_of = [](int words) { const int q = (words + 3) / 4; return kXf == 8 ? 4 * (q | 1) : (words | 1); };
    o = (o + 3) & ~3;
    h.stride_regs = stride_of(o) + std::max(m->dbg.stride_add, 0);
    h.stride_forced = m->dbg.stride_add >= 0;
    h.c_r2 = o; o += K > 64 ? h.kpow2 : ((K + 3) & ~3);
    h.c_kp = o; o += 3 * K;
    h.stride_lds = stride_of(o) + std::max(m->dbg.stride_add, 0);

    m->h_brec = brec; m->h_lev_adr = lev_adr; m->h_ab_jadr = ab_jadr; m->h_ab_jnum = ab_jnum; m->h_xf = xf;
    m->h_aj_pos = aj_pos;
    m->h_bpos.resize((size_t)3 * nab);
    for (int s = 0; s < nab; ++s)
        for (int i = 0; i < 3; ++i) m->h_bpos[3 * s + i] = brec[s].pos[i];
    bool any_bquat = false;
    for (int s = 0; s < nab; ++s) any_bquat = any_bquat || !(brec[s].flags & 1);
    {   // Split kinematics (PlanHeader::fk3; stac_plan.hpp): a free root at qpos 0 .. 6 on the ONE top-level active body, hinges
        // below it, no oriented body.  The lean kernels run nothing else, so a model without it takes the generic kernels.
        h.fk3 = 0;
        // (the free root's own body may be oriented: a free joint sets the body's pose, body_pos / body_quat do not enter -- mouse;
        //  an oriented body below it is one more product of P1, q_parent * body_quat, with the constant as its right factor -- fly)
        std::vector<float> bq_const;
        for (int s2 = 1; s2 < nab; ++s2)
            if (!(brec[s2].flags & 1)) bq_const.insert(bq_const.end(), brec[s2].quat, brec[s2].quat + 4);
        const int nbq = (int)bq_const.size() / 4;
        bool ok = naj >= 1 && aj_type[0] == STAC_JNT_FREE && aj_qadr[0] == 0 && ab_jnum[0] >= 1 && ab_jadr[0] == 0 && nqj == 1 && !has_ball &&
                  getenv("STAC_HIP_NOFK3") == nullptr && !(nbq > 0 && getenv("STAC_HIP_NOFK3BQ") != nullptr);
        for (int j = 1; j < naj && ok; ++j) ok = aj_type[j] == STAC_JNT_HINGE;
        for (int s = 1; s < nab && ok; ++s) ok = ab_parent[s] != 0;
        if (ok) {
            PlanHeader g = h;  // the lean chain layout: regions of whole 4-word groups, stride 4 x odd (16-byte aligned entries)
            // capacities of a program area: the full program's own sizes (a pruned program is a sub-DAG: it is checked against them)
            Fk3Program probe;
            g.c3_qb = 0; g.c3_pb = 0; g.c3_ql = 0; g.c3_bq = 0; g.nbq = nbq; g.stride3 = 1;
            ok = build_fk3_program(m, g, nullptr, 254, 4096, 252, probe);
            if (ok) {
                // (cap1 even, >= 10: the table area 16 (cap1 + 2) holds cap1 / 2 + 3 records of 24 words; cap3: whole blocks)
                g.fk3_cap1 = std::max(probe.n1, 10); g.fk3_cap2 = std::max(probe.n2, 32); g.fk3_cap3 = std::max(probe.n3, 4);
                int o3 = 0;
                g.c3_qb = o3; o3 += (naj + 1 + nbq) * 4;                 // (+ 1: the sink of idle P1 positions; then the oriented bodies' nodes)
                g.c3_pb = o3; o3 += ((g.fk3_cap3 * 4 + 4) * 3 + 3) & ~3;  // (step, position) slots, root position, sink, slack of the prefetch
                g.c3_ql = o3; o3 += std::max(naj * 4, (h.nrange * kXf + 3) & ~3);
                g.c_rw = g.c3_ql;                                         // a full trip's range sums go where the joint-local quaternions were
                g.c3_rw0 = o3; o3 += 8;
                g.c3_bq = o3; o3 += 4 * nbq;                              // body_quat (w, x, y, z) of the oriented bodies: constants, written by the prologue
                g.c_jn = o3; o3 += 4;
                g.c_qsv = o3; o3 += 4 * nqj;
                g.c_sw = o3; o3 += (std::max(K * kXf, h.nqpad + kXf) + 3) & ~3;
                g.c_sink = g.c_sw + std::max(K * kXf, h.nqpad + kXf) - kXf;
                g.c_gg = g.c_qe = g.c_sw;
                g.c_bx = g.c_ja = 0;  // (unused by the lean kernels)
                const int q4 = (o3 + 3) / 4;
                g.stride3 = 4 * (q4 | 1);
                if ((g.stride3 / 4) % 8 == 0) g.stride3 += 8;  // (never a multiple of 32 words: the chains of a wavefront on different banks)
                g.stride_regs = g.stride_lds = g.stride3;
                g.stride_forced = 1;  // (q_chain_stride takes it as it is)
                Fk3Program full;
                ok = build_fk3_program(m, g, nullptr, g.fk3_cap1, g.fk3_cap2, g.fk3_cap3, full);
                if (ok) {
                    g.fk3 = 1;
                    g.fk3_n = full.n1 | full.n3 << 8 | full.n2 << 16;
                    g.off3_bq = put_raw(bq_const.data(), bq_const.size());
                    g.off3_prog = put_raw(full.words.data(), full.words.size());
                    g.off3_site = g.off3_prog + 16 * (g.fk3_cap1 + 2) + 4 * g.fk3_cap2 + 4 * g.fk3_cap3;
                    std::vector<int32_t> blank3(full.words.size(), 0);
                    g.off3_root = put_raw(blank3.data(), blank3.size());
                    // what both layouts share
                    const int32_t keep[] = {g.fk3, g.off3_site, g.off3_prog, g.off3_root, g.fk3_n,
                                            g.fk3_cap1, g.fk3_cap2, g.fk3_cap3};
                    h.fk3 = keep[0]; h.off3_site = keep[1]; h.off3_prog = keep[2]; h.off3_root = keep[3]; h.fk3_n = keep[4];
                    h.fk3_cap1 = keep[5]; h.fk3_cap2 = keep[6]; h.fk3_cap3 = keep[7];
                    h.c3_ql = g.c3_ql; h.c3_qb = g.c3_qb; h.c3_pb = g.c3_pb; h.c3_rw0 = g.c3_rw0; h.stride3 = g.stride3;
                    h.c3_bq = g.c3_bq; h.off3_bq = g.off3_bq; h.nbq = g.nbq;
                    m->h3 = g;
                }
            }
        }
    }
    {   // FK program (FkStep records, see stac_plan.hpp); a second area of the same size takes the pruned program of
        // the root passes, which depends on the call's trunk keypoints (fill_root_program)
        h.fk_rec_words = any_bquat ? 16 : 12;
        {   // header of a program area: a flag word per micro-level (of the full program: a pruned one has fewer),
            // then the first step's ql offset per position
            int mm = 0;
            for (int l = 0; l < nlev; ++l) {
                int ml = 1;
                for (int s2 = lev_adr[l]; s2 < lev_adr[l + 1]; ++s2) ml = std::max(ml, ab_jnum[s2]);
                mm += ml;
            }
            h.n_mlev_hdr = std::max((mm + 1) & ~1, 2);
            h.fk_hdr_words = ((h.n_mlev_hdr >> 1) + h.max_width + 3) & ~3;  // one flag word per pair of micro-levels
        }
        int n_mlev = 0;
        const std::vector<int32_t> prog = build_fk_program(m, nullptr, &n_mlev, &h.fk_uniform);  // (a pruned program is a subset)
        h.off_fkstep = put_raw(prog.data(), prog.size());
        h.n_mlev = n_mlev;
        std::vector<int32_t> blank(prog.size(), 0);
        h.off_fkroot = put_raw(blank.data(), blank.size());
        m->n_mlev_root = 0;
    }
    h.total_words = (int)B.size();

    h.chain_stride = h.stride_lds;
    if (h.fk3) {  // the lean layout's copy of everything that was settled after it was made
        PlanHeader &g = m->h3;
        g.fk_rec_words = h.fk_rec_words; g.n_mlev_hdr = h.n_mlev_hdr; g.fk_hdr_words = h.fk_hdr_words; g.off_fkstep = h.off_fkstep;
        g.off_fkroot = h.off_fkroot; g.n_mlev = h.n_mlev; g.fk_uniform = h.fk_uniform; g.total_words = h.total_words;
        g.chain_stride = g.stride3;
    }
    return STAC_OK;
}

// LDS diet (PlanHeader::plan_skip): the offsets of the staged areas as the kernel sees them, from the first staged word
static void rebase_plan_offsets(PlanHeader &h) {
    const int sk = h.plan_skip;
    if (sk <= 0) return;
    int32_t *offs[] = {&h.off_joint, &h.off_site, &h.off_range, &h.off_lb, &h.off_ub, &h.off_qpos0, &h.off_quat_adr,
                       &h.off_active, &h.off_fkstep, &h.off_fkroot, &h.off3_site, &h.off3_prog, &h.off3_root, &h.off3_bq};
    for (int32_t *o : offs) *o -= sk;
    h.off_lev_adr = h.off_body = 0;  // not staged (nothing of a program launch reads them)
}

static int q_mb_words(int nkinds, int G) { return ((nkinds + 1) * G + 3) & ~3; }  // per-kind mask bits + one row of active-coordinate bits
// (G = 16: two chains share a 32-lane half of the wavefront, and the 32-bit LDS accesses bank by word address mod 32 per
// half: a stride of 16 mod 32 puts the second chain's run of 16 consecutive words on the other 16 banks.  Measured on the
// bench, profiles/r03/stride_sweep.txt: conflict cycles 25 % -> 16 % of the LDS-active ones; time within the noise,
// as it is for a stride of 0 mod 32 with 46 % -- the conflicts are not what the kernel waits for.)
static int q_chain_stride(const PlanHeader &h, int G) {
    int s = h.K <= kSiteRounds * G ? h.stride_regs : h.stride_lds;
    if (G == 16 && kXf == 7 && !h.stride_forced) s += (16 - s % 32 + 32) % 32;
    return s;
}
static size_t q_lds_bytes(const PlanHeader &h, int G, int nkinds, int wpb) {
    const int plan_words = (h.total_words - h.plan_skip + 3) & ~3;  // h is the per-launch copy: [plan_skip, total_words) = what this launch stages
    return (size_t)(plan_words + q_mb_words(nkinds, G) + wpb * (64 / G) * q_chain_stride(h, G)) * sizeof(float);
}
constexpr size_t kLdsPerCu = 160 * 1024;
constexpr int kFullCus = 256;  // a whole MI355X (the placement kernel's SIMD census is laid out for it)
#define kCus t_cus
// latency mode: up to this many chains a chain is spread over 8 / 4 wavefronts of a workgroup (else one per chain)
// (measured, rodent, 250-frame clips, profiles/r02/lat_sweep.txt: with the four-lanes-per-position kinematics four
// wavefronts per chain are 6-12 % ahead of eight from 40 to 256 chains and hold twice as many chains: eight are kept
// behind STAC_HIP_SPECG=64)
constexpr long kSpec64MaxChains = 0, kSpec32MaxChains = 512;
// one wavefront per chain: FOUR roles of 16 lanes (two candidates and their momentum points per trip: 1.16 trips per
// iteration, but the four-lanes-per-position kinematics and fewer rounds in every per-item phase): measured 229 k against
// 179 k frames/s on 1 000 x 250 clips for eight roles of 8 lanes (which stay behind STAC_HIP_SPECG=8)
constexpr int kLatG = 16, kLatR = 4;
constexpr int kOrderMinChains = 2048, kOrderMaxFrames = 8;  // stac_q_phase: chains ordered by expected length and placed by SIMD load
// latency mode with 4 roles per chain from this many chains on (never auto-selected below: developer switch STAC_HIP_SPECR)
constexpr long kSpec4MinChains = 1L << 40;

// Wavefronts per workgroup: the waves of a block share one copy of the plan, so more chains fit the
// 160 KiB of a CU.  Returns the wpb (1..8) that maximises resident chains per CU for this G.
// Launch shape for a given G: wavefronts per workgroup (the waves of a block share one plan copy) and the
// register-cap variant.  wpe = 2 keeps everything in registers (<= 256 VGPRs, 8 waves per CU); wpe = 3
// (<= 168 VGPRs, 16-lane groups only) holds up to twelve waves as ONE workgroup (eleven: 3,3,3,2 over the SIMDs; two 5-wave
// workgroups could stack four waves on a SIMD -- the lean rodent layout fits eleven: 100 000 frames 649 -> 720 k frames/s); wpe = 4 (<= 128 VGPRs, more spills to scratch) admits 16 waves
// per CU.  LDS is allocated in granules (1280 B observed on gfx950: five 32 224-B workgroups do not
// fit a CU).
struct QShape { int wpb, wpe, waves_per_cu; };
static QShape pick_shape(const PlanHeader &h, int G, int nkinds, long waves_needed = -1, bool lean = false) {
    constexpr size_t kGranule = 1280;
    QShape best{0, 2, 0};
    for (int wpe = 2; wpe <= 4; ++wpe) {
        if (!(lean ? q_phase_has_lean_variant(G, h.nq, wpe, 0) : q_phase_has_variant(G, h.nq, wpe))) continue;  // (stac_kernels.hip, STAC_Q_SHAPES: only shapes that pass the spill gate are built)
        for (int wpb = 1; wpb <= (wpe == 3 ? 12 : 8); ++wpb) {
            size_t lds = q_lds_bytes(h, G, nkinds, wpb);
            if (lds > kLdsPerCu) break;
            lds = (lds + kGranule - 1) / kGranule * kGranule;
            int blocks = (int)(kLdsPerCu / lds);
            if (wpb <= 4) {
                // up to four waves per workgroup spread over the SIMDs: only the CU total matters
                if (blocks * wpb > 4 * wpe) blocks = 4 * wpe / wpb;
            } else {
                // a bigger workgroup puts ceil(wpb/4) waves on some SIMD; assume the worst stacking
                const int per_simd = (wpb + 3) / 4;
                if (blocks * per_simd > wpe) blocks = wpe / per_simd;
            }
            const int waves = blocks * wpb;
            // prefer more resident waves, then the smaller workgroup; a variant with a tighter register cap (more
            // spills) has to bring at least 20 % more waves than the best roomier one
            if (wpe == best.wpe ? waves > best.waves_per_cu : waves * 5 >= best.waves_per_cu * 6) best = QShape{wpb, wpe, waves};
            // few chains: the first shape that holds every wave at once is enough -- small workgroups spread
            // over more CUs and the 2-waves-per-SIMD variant does not spill
            if (waves_needed >= 0 && (long)waves * kCus >= waves_needed) return QShape{wpb, wpe, waves};
        }
    }
    return best;
}

// ---- latency (speculative) mode: eight evaluation roles of G lanes per chain -------------------------------------------
// G = 8: one chain per wavefront, `chains` wavefronts per workgroup; G = 32 / 64: one chain per workgroup of 4 / 8
// wavefronts.  A chain's LDS block = 8 role regions + the exchange area (accept flags, losses, two gradients).
static int spec_xch_words(const PlanHeader &h) { return 64 + 4 * h.nqpad + 4; }
// nr = evaluation roles per chain: 8, or 4 (two chains per wavefront at 8 lanes per role: large batches)
static size_t spec_lds_bytes(const PlanHeader &h, int G, int nkinds, int chains, int nr = 8) {
    const int plan_words = (h.total_words - h.plan_skip + 3) & ~3;
    // (+ 1 KB: the range sums of the latency kernels read up to 23 site entries behind a range's end -- unclamped addresses, the values
    //  dropped --, which behind the workgroup's last chain would leave the allocation)
    return (size_t)(plan_words + q_mb_words(nkinds, G) + chains * (nr * q_chain_stride(h, G) + spec_xch_words(h))) * sizeof(float) + 1024;
}
struct SpecShape { int G, chains_per_block, waves_per_block; long resident; };  // resident = chains the chip holds at once
// The one-wavefront-per-chain latency kernel (kLatG lanes per role: straggler hand-off, large clip counts) exists up to 8
// solver registers per lane (stac_kernels.hip, launch_q_phase); wider models run four wavefronts per chain and keep their stragglers
static bool lat_one_wave_ok(const PlanHeader &h) { return h.nq <= 16 * 8; }
static SpecShape pick_spec_shape(const PlanHeader &h, int G, int nkinds, long nchains = -1, int nr = 8) {
    constexpr size_t kGranule = 1280;
    SpecShape best{G, 0, 0, 0};
    if ((G == 8 || G == 16) && G * nr <= 64) {  // 256-VGPR kernel: two waves per SIMD, eight per CU; 64 / (G nr) chains per wavefront
        const int cw = 64 / (G * nr);
        for (int w = 1; w <= 8; ++w) {
            size_t lds = spec_lds_bytes(h, G, nkinds, w * cw, nr);
            if (lds > kLdsPerCu) break;
            lds = (lds + kGranule - 1) / kGranule * kGranule;
            const int blocks = std::min((int)(kLdsPerCu / lds), 8 / w);
            const long res = (long)blocks * w * cw * kCus;
            if (res > best.resident) best = SpecShape{G, w * cw, w, res};
            if (nchains >= 0 && res >= nchains) return SpecShape{G, w * cw, w, res};  // small workgroups spread over more CUs
        }
        return best;
    }
    const int nw = G * nr / 64;  // wavefronts per chain; 256-VGPR kernels as well (the 128-VGPR builds spill > 100 registers): eight waves per CU
    size_t lds = spec_lds_bytes(h, G, nkinds, 1);
    if (lds > kLdsPerCu) return best;
    lds = (lds + kGranule - 1) / kGranule * kGranule;
    const int blocks = std::min((int)(kLdsPerCu / lds), 8 / nw);
    return SpecShape{G, 1, nw, (long)blocks * kCus};
}

extern "C" stac_model *stac_model_create(const stac_model_tables *t) {
    if (!t) { fail(STAC_ERR_INVALID, "null tables"); return nullptr; }
    if (stac_device_count() <= 0) {
        fail(STAC_ERR_NO_DEVICE, "no HIP device visible: the STAC engine has no CPU fallback");
        return nullptr;
    }
    stac_model *m = new stac_model();
    (void)hipGetDevice(&m->device);
    {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, m->device) == hipSuccess && n > 0) m->cus = n;
        t_cus = m->cus;
    }
    m->dbg.read_env();
    if (build_plan(m, t) != STAC_OK) { delete m; return nullptr; }
    if (m->dbg.flags >= 0 && (m->dbg.flags & 4)) m->h.fk_uniform = 0;  // A/B switch: run the program step by step through its flags and forms
    const int nb = t->nbody, nj = t->njnt, nq = t->nq, K = t->nsite;
    hipError_t e = hipSuccess;
    auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    chk(upload(&m->d_blob, m->blob_host.data(), m->blob_host.size()));
    chk(upload(&m->d_body_parentid, t->body_parentid, (size_t)nb));
    chk(upload(&m->d_body_jntadr, t->body_jntadr, (size_t)nb));
    chk(upload(&m->d_body_jntnum, t->body_jntnum, (size_t)nb));
    chk(upload(&m->d_body_pos, t->body_pos, (size_t)nb * 3));
    chk(upload(&m->d_body_quat, t->body_quat, (size_t)nb * 4));
    chk(upload(&m->d_jnt_type, t->jnt_type, (size_t)nj));
    chk(upload(&m->d_jnt_qposadr, t->jnt_qposadr, (size_t)nj));
    chk(upload(&m->d_jnt_pos, t->jnt_pos, (size_t)nj * 3));
    chk(upload(&m->d_jnt_axis, t->jnt_axis, (size_t)nj * 3));
    chk(upload(&m->d_qpos0, t->qpos0, (size_t)nq));
    chk(upload(&m->d_site_bodyid, t->site_bodyid, (size_t)K));
    chk(upload(&m->d_site_pos, t->site_pos, (size_t)K * 3));
    {
        const FkTables ft = build_fk_tables(t);
        m->fk_nslots = ft.nslots;
        chk(upload(&m->d_fk_brec, ft.brec.data(), ft.brec.size()));
        chk(upload(&m->d_fk_jrec, ft.jrec.data(), ft.jrec.size()));
        chk(upload(&m->d_fk_sites, ft.sites.data(), ft.sites.size()));
    }
    m->masks_bytes = (size_t)kMaxKinds * m->h.nqpad + 4 * (size_t)K + 64;
    chk(hipMalloc(reinterpret_cast<void **>(&m->d_masks), m->masks_bytes));
    chk(hipMalloc(reinterpret_cast<void **>(&m->d_bounds), 2 * (size_t)m->h.nqpad * sizeof(float)));
    chk(hipMalloc(reinterpret_cast<void **>(&m->d_ctl), 8 * sizeof(int32_t)));
    {   // hand-off buffer of the straggler hand-off at its maximum size (one entry per wavefront the latency kernel
        // can hold resident), so that no launch ever reallocates (= waits for the device)
        m->hand_cap = lat_one_wave_ok(m->h) ? (int)pick_spec_shape(m->h, kLatG, 1, -1, kLatR).resident : 0;
        if (m->hand_cap > 0)
            chk(hipMalloc(reinterpret_cast<void **>(&m->d_hand), (size_t)m->hand_cap * (3 * (size_t)m->h.nqpad + 12) * sizeof(float)));
    }
    if (e != hipSuccess) {
        fail(STAC_ERR_HIP, std::string("model upload failed: ") + hipGetErrorString(e));
        stac_model_destroy(m);
        return nullptr;
    }
    g_err.clear();
    return m;
}

extern "C" void stac_model_destroy(stac_model *m) {
    if (!m) return;
    DeviceGuard dg(m);
    void *ptrs[] = {m->d_bounds, m->d_blob, m->d_body_parentid, m->d_body_jntadr, m->d_body_jntnum, m->d_body_pos,
                    m->d_body_quat, m->d_jnt_type, m->d_jnt_qposadr, m->d_jnt_pos, m->d_jnt_axis,
                    m->d_qpos0, m->d_site_bodyid, m->d_site_pos, m->d_fk_brec, m->d_fk_jrec, m->d_fk_sites, m->d_masks, m->d_scratch, m->d_lm_tab, m->d_ctl, m->d_hand, m->d_order};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete m;
}

extern "C" int32_t stac_model_info(const stac_model *m, int32_t *info) {
    if (!m || !info) return fail(STAC_ERR_INVALID, "null argument");
    info[0] = m->h.nbody; info[1] = m->h.njnt; info[2] = m->h.nq; info[3] = m->h.K;
    info[4] = m->h.nab; info[5] = m->h.naj; info[6] = m->h.nlev;
    int width = 0;
    const int *lev = reinterpret_cast<const int *>(m->blob_host.data()) + m->h.off_lev_adr;
    for (int l = 0; l < m->h.nlev; ++l) width = std::max(width, lev[l + 1] - lev[l]);
    info[7] = width;
    return STAC_OK;
}

extern "C" int32_t stac_set_site_pos(stac_model *m, const float *offsets, void *stream) {
    if (!m || !offsets) return fail(STAC_ERR_INVALID, "null argument");
    DeviceGuard dg(m);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(m->d_site_pos, offsets, sizeof(float) * 3 * m->h.K, hipMemcpyDeviceToDevice, s));
    // scatter into the plan's SiteRec.pos (stride 4 words)
    HIP_TRY(hipMemcpy2DAsync(m->d_blob + m->h.off_site, sizeof(SiteRec), offsets, 3 * sizeof(float), 3 * sizeof(float),
                             m->h.K, hipMemcpyDeviceToDevice, s));
    return STAC_OK;
}

extern "C" int32_t stac_get_site_pos(const stac_model *m, float *out, void *stream) {
    if (!m || !out) return fail(STAC_ERR_INVALID, "null argument");
    DeviceGuard dg(m);
    HIP_TRY(hipMemcpyAsync(out, m->d_site_pos, sizeof(float) * 3 * m->h.K, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return STAC_OK;
}

static int fk_impl(const stac_model *mc, const float *qpos, int32_t N, float *qn, float *xpos, float *xquat,
                   float *site_xpos, int normalize, void *stream) {
    stac_model *m = const_cast<stac_model *>(mc);
    if (!m || !qpos || N < 0) return fail(STAC_ERR_INVALID, "stac_fk: bad argument");
    if (N == 0) return STAC_OK;
    DeviceGuard dg(m);
    // (outputs the caller does not want are not written: the body transforms live in the kernel's LDS rows)
    HIP_TRY(launch_fk(m->full(), qpos, N, qn, xpos, xquat, site_xpos, normalize, (hipStream_t)stream));
    return STAC_OK;
}

// Diagnostics, not part of include/stac_hip.h: {G, NQR, WPE, SPECP} of the most recent q_phase_kernel launch of the calling thread
// (SPECP bit 0 = a lean kernel, SPECP & ~1 = evaluation roles of the latency mode; tests/test_gpu_parity.py checks that the shapes
// the bench runs in are the lean ones).
namespace stac { extern thread_local int g_last_q_shape[4]; }
extern "C" void stac_debug_last_q_kernel(int32_t *out4) {
    for (int i = 0; i < 4; ++i) out4[i] = stac::g_last_q_shape[i];
}

extern "C" int32_t stac_fk(const stac_model *m, const float *qpos, int32_t N, float *qn, float *xpos,
                           float *xquat, float *site_xpos, void *stream) {
    return fk_impl(m, qpos, N, qn, xpos, xquat, site_xpos, 1, stream);
}

// Lanes per chain.  Fewer lanes = more chains per wave instruction (throughput); more lanes = fewer idle lanes
// when there are few chains (latency).  Measured on the rodent (20-frame clips, frames/s in thousands):
//   chains      500   1000   1500   2000   3000   4000   5000   8000
//   latency     63    123    165    126    167    168    160      -      (speculative kernel, 1536 chains resident)
//   G = 32      31     59     91    122    168    223    250      -
//   G = 16      25     51     76    102    149    199    239    362
// i.e. the latency mode up to about 2.3 x its resident capacity, then 32 lanes until the 16-lane kernel has
// about 45 % of its slots filled (8 and 4 lanes only on request: with the LDS footprint of a chain they cannot
// keep two waves per SIMD resident).  Returns 0 for the latency mode.
static int pick_lanes(const stac_model *m, const PlanHeader *lean_h, int requested, int nchains, int nkinds, bool spec_allowed, int frames) {
    // (lean_h: the header a lean launch of this call would stage -- chain layout of the split kinematics, plan from the joint records
    //  on --, or null when the call cannot be lean.  Every candidate width is sized with the header ITS launch would get: the lean one
    //  where a lean instantiation holds nq at that width, the generic one elsewhere (mouse: lean at 32 lanes only))
    const int nq = m->h.nq;
    // (a lean width must also hold the model's sites in registers -- what q_phase_lean_conditions will ask of the launch --: else the width
    //  would be sized with a header the launch does not get)
    const int K = m->h.K;
    auto thr_lean = [&](int G) { return lean_h && q_phase_lean_holds(G, nq, K) && (q_phase_has_lean_variant(G, nq, 2, 0) || q_phase_has_lean_variant(G, nq, 3, 0)); };
    auto thr_shape = [&](int G) { const bool ln = thr_lean(G); return pick_shape(ln ? *lean_h : m->h, G, nkinds, -1, ln); };
    PlanHeader lean_spec = lean_h ? *lean_h : m->h;  // (a lean latency launch stages from the joint records on: no momentum table)
    if (lean_h) lean_spec.plan_skip += 2 * kTTab;
    auto spec_hdr = [&](int G, int nr) -> const PlanHeader & { return lean_h && q_phase_lean_holds(G, nq, K) && q_phase_has_lean_variant(G, nq, 2, nr) ? lean_spec : m->h; };
    if (requested == 4 || requested == 8 || requested == 16 || requested == 32 || requested == 64) return requested;
    if (spec_allowed) {
        const SpecShape ss = lat_one_wave_ok(m->h) ? pick_spec_shape(spec_hdr(kLatG, kLatR), kLatG, nkinds, -1, kLatR)  // one chain per wave
                                                  : pick_spec_shape(spec_hdr(32, 8), 32, nkinds);
        // Measured crossover against the throughput kernel (rodent, profiles/r03/shape_sweep.txt): 1.8 x the resident chains
        // for single-frame clips, 2.3 x for two or three frames, 2.6 x from four frames on (the longer a clip, the less
        // of it are the root solves that the throughput kernel runs as fast trips)
        // (four wavefronts per chain -- models too wide for one wavefront per chain: mouse --: 3.2 x from four frames on, measured
        //  with the lean kernels: 700 chains 51.6 k against 45.7 k frames/s, 1 000 chains 54.8 k against 65.2 k)
        const long x10 = frames <= 1 ? 18 : (frames < 4 ? 23 : (lat_one_wave_ok(m->h) ? 26 : 32));
        if (ss.resident && (long)nchains * 10 <= ss.resident * x10) return 0;
    }
    const QShape s16 = thr_shape(16);
    {   // Models whose chains are so large that 16-lane groups leave a wavefront or less per SIMD (mouse: 6.9 KB of LDS per chain,
        // three 4-chain wavefronts per CU): 32-lane groups hold MORE chains per CU there (seven 2-chain wavefronts) and halve
        // the rounds of every per-coordinate and per-joint phase -- 27.3 k -> 36.4 k frames/s on 10 000 mouse frames.
        const QShape s32 = thr_shape(32);
        if (s16.wpb && s32.wpb && s32.waves_per_cu * 2 > s16.waves_per_cu * 4 && m->h.max_width * 4 <= 32 &&
            (long)nchains * 100 > (long)s32.waves_per_cu * kCus * 2 * 45)
            return 32;
    }
    // (a model whose only lean width is 32 -- more coordinates than the 16-lane lean shapes hold: mouse -- stays on it: the generic
    //  kernels run the step-program kinematics, 2.2 x slower on its 84-product spine)
    if (thr_lean(32) && !thr_lean(16) && thr_shape(32).wpb) return 32;
    if (s16.wpb && (long)nchains * 100 > (long)s16.waves_per_cu * kCus * 4 * 25) return 16;  // (from a quarter of the resident slots on: 16 lanes beat 32 at every measured size above the latency kernel's range)
    if (nchains > 2500 && thr_shape(32).wpb) return 32;
    return 64;
}

// QArgs::free0p for a launch with G lanes per group: active joint 0 a free joint at qpos 0 .. 6 (the first quaternion
// joint: ordinal 0, JointRec::q0 in build_plan) and its seven coordinates in register 0 of lanes 0 .. 6
static int free0_ordinal_p1(const stac_model *m, int G) {
    return (G >= 8 && m->h.naj > 0 && m->h_aj_type[0] == STAC_JNT_FREE && m->h_aj_qadr[0] == 0 && !m->dbg.nofree0) ? 1 : 0;
}

// QArgs::flags bit 4 for a launch whose free0p is set: every active joint but the free root (which lanes 0 .. 6 handle apart) is a
// hinge -- the joint loops of the generic kernels then run the hinge formulas without looking at the joint type (mouse, fly), and
// the lean kernels, which have it compiled in, may be chosen (launch_q_phase)
static void set_hinges_flag(const stac_model *m, QArgs &a) {
    bool hinges = a.free0p == 1;
    for (int j = 1; j < m->h.naj && hinges; ++j) hinges = m->h_aj_type[j] == STAC_JNT_HINGE;
    a.flags = (a.flags & ~16) | (hinges ? 16 : 0);
}

static int run_q(const stac_model *mc, const stac_q_params *p, QArgs &a, int nchains, hipStream_t s) {
    stac_model *m = const_cast<stac_model *>(mc);
    if (p->maxiter < 1) return fail(STAC_ERR_INVALID, "maxiter must be >= 1");
    if (p->maxls < 1) return fail(STAC_ERR_INVALID, "maxls must be >= 1");
    a.hdr = nullptr;
    a.plan = m->d_blob;
    a.h = m->h;
    a.tol = p->tol; a.maxiter = p->maxiter; a.maxls = p->maxls;
    const int nkinds = a.single ? 1 : a.P + 3;
    const DebugSwitches &dbg = m->dbg;
    if (dbg.flags >= 0) a.flags = dbg.flags;  // developer A/B switches (see stac_plan.hpp)
    if (dbg.nolean) a.flags |= 8;  // (host only: the generic kernels even where a lean one fits)
    if (a.h.c_gg != a.h.c_sw || a.h.c_qe != a.h.c_sw) return fail(STAC_ERR_INVALID, "plan layout: the PG kernel expects c_qe == c_gg == c_sw");
#ifdef STAC_PROFILE
    static unsigned long long *d_prof = nullptr;
    if (!d_prof) { (void)hipMalloc(reinterpret_cast<void **>(&d_prof), 16 * sizeof(unsigned long long)); }
    (void)hipMemsetAsync(d_prof, 0, 16 * sizeof(unsigned long long), s);
    a.prof = d_prof;
#endif
    // (clip length for the latency / throughput crossover; without root optimisation -- the tethered fly -- a single-frame clip
    //  has no root solves for the throughput kernel to run as fast trips: counted like a two-frame clip, measured)
    // Which chain layout a launch gets: the lean kernels (split kinematics) have their own (PlanHeader::fk3, stac_model::h3).  Decided per
    // launch from the launch's own flags and lanes (lean_for); `likely` only serves the shape heuristics before the lanes are known.
    auto lean_for = [&](QArgs &q, int lanes, int wpe_or_roles, bool spec_launch) {
        q.h.fk3 = m->h.fk3; q.h.K = m->h.K; q.h.nqj = m->h.nqj; q.h.has_ball = m->h.has_ball;
        if (dbg.nolean || !m->h.fk3 || !q_phase_lean_conditions(q, lanes)) return false;
        if (spec_launch) return q_phase_has_lean_variant(lanes, m->h.nq, 2, wpe_or_roles);
        return q_phase_has_lean_variant(lanes, m->h.nq, 2, 0) || q_phase_has_lean_variant(lanes, m->h.nq, 3, 0);
    };
    // a lean launch stages the plan from the joint records up to the split-kinematics tables (the root program only when it is used)
    // (thr: a throughput launch -- it stages the momentum table, which stands right in front of the joint records; the latency kernels
    //  compute the pair and leave it out: 2 KB that decide whether 1 000 two-wavefront chains are resident)
    auto lean_header = [&](const QArgs &q, bool thr) {
        PlanHeader hh = m->h3;
        hh.plan_skip = m->h.off_joint - (thr ? 2 * kTTab : 0);
        const int area = 16 * (hh.fk3_cap1 + 2) + 4 * hh.fk3_cap2 + 4 * hh.fk3_cap3 + hh.K + hh.naj;
        hh.total_words = (q.do_root_opt && q.fk3r_n != 0) ? hh.off3_root + area : hh.off3_root;
        return hh;
    };
    const bool likely_lean = m->h.fk3 && !a.single && !a.bounds && !dbg.nolean && !(dbg.flags >= 0 && dbg.flags != 0);
    const PlanHeader lean_h = likely_lean ? lean_header(a, true) : m->h;
    // (clip length for the latency / throughput crossover; without root optimisation -- the tethered fly -- a single-frame clip
    //  has no root solves for the throughput kernel to run as fast trips: counted like a two-frame clip, measured)
    int G = pick_lanes(m, likely_lean ? &lean_h : nullptr, p->lanes_per_chain, nchains, nkinds, !a.single, a.single ? 1 : (a.do_root_opt ? a.F : std::max(a.F, 2)));
    hipError_t e = hipErrorInvalidValue;
    int cap = 0;
    // Latency mode: when there are so few chains that each would get a whole wavefront anyway (G = 64), let
    // the wavefront's eight 8-lane groups evaluate the line-search candidates and next momentum points of
    // that ONE chain speculatively (q_phase_kernel<8, ., ., true>): one trip per iteration instead of three.
    int spec = G == 0 ? 1 : 0;
    if (dbg.spec >= 0) spec = a.single ? 0 : dbg.spec;
    if (G == 0) G = spec ? 64 : 32;
    if (spec) {
        // lanes per evaluation role: with very few chains every chain gets a whole workgroup of 4 or 8 wavefronts
        // (measured, rodent, 250-frame clips: DESIGN.md 2.1), else one wavefront per chain
        int sg = kLatG;
        {
            const bool l32 = likely_lean && q_phase_lean_holds(32, m->h.nq, m->h.K) && q_phase_has_lean_variant(32, m->h.nq, 2, 8);
            const SpecShape s64 = pick_spec_shape(m->h, 64, nkinds), s32 = pick_spec_shape(l32 ? lean_header(a, false) : m->h, 32, nkinds);
            if (s64.resident && (long)nchains <= kSpec64MaxChains) sg = 64;
            else if (s32.resident && (long)nchains <= kSpec32MaxChains) sg = 32;
        }
        if (dbg.specg == 8 || dbg.specg == 16 || dbg.specg == 32 || dbg.specg == 64) sg = dbg.specg;
        // no instantiation wider than 10 registers per vector at 8 lanes per role, 8 at 16 (stac_kernels.hip): more lanes per role
        if (sg == 8 && m->h.nq > 80) sg = 16;
        if (sg == 16 && m->h.nq > 128) sg = 32;
        // roles per chain: four (two candidates + their momentum points, two chains per wavefront) once the batch is so
        // large that throughput counts, not the latency of one chain
        int sr = (sg == 8 && (long)nchains >= kSpec4MinChains) ? 4 : 8;
        if (sg == 8 && (dbg.specr == 4 || dbg.specr == 8)) sr = dbg.specr;
        if (sg == 16) sr = 4;  // four roles of 16 lanes: one chain per wavefront
        a.free0p = free0_ordinal_p1(m, sg);
        set_hinges_flag(m, a);
        // Between the four-wavefront kernel's resident chains (512) and twice that: TWO wavefronts per chain, eight roles of 16 lanes
        // (lean kernel only) -- one trip per iteration like the four-wavefront kernel instead of 1.16 longer ones (measured, 1 000 chains
        // of 250 frames: profiles/r05/NOTES.md)
        if (sg == 16 && dbg.specg < 0 && lean_for(a, 16, 8, true)) {
            const SpecShape s2 = pick_spec_shape(lean_header(a, false), 16, nkinds, -1, 8);
            if (s2.resident && (long)nchains <= s2.resident) sr = 8;
        }
        if (sg == 16 && dbg.specr == 8 && lean_for(a, 16, 8, true)) sr = 8;
        const bool lean = lean_for(a, sg, sr, true);
        const PlanHeader hh = lean ? lean_header(a, false) : m->h;
        const SpecShape sh = pick_spec_shape(hh, sg, nkinds, nchains, sr);
        if (sh.chains_per_block) {
            const size_t lds = spec_lds_bytes(hh, sg, nkinds, sh.chains_per_block, sr);
            if (dbg.verbose)
                fprintf(stderr, "[stac] q_phase: chains=%d speculative%s, %d roles of %d lanes per chain (%d wavefront(s) per chain), %d chain(s) per workgroup, lds=%zu B/block, resident=%ld\n",
                        nchains, lean ? " (lean)" : "", sr, sg, std::max(sg / 8, 1), sh.chains_per_block, lds, sh.resident);
            a.mb_words = q_mb_words(nkinds, sg);
            a.h = hh;
            a.h.chain_stride = q_chain_stride(hh, sg);
            // chain queue (see below): more clips than resident chain slots -> the roles of a finished clip take the next
            long resident = pick_spec_shape(hh, sg, nkinds, -1, sr).resident / sh.chains_per_block * sh.chains_per_block;
            if (dbg.queue > 0 && dbg.queue < nchains) resident = (long)(dbg.queue + sh.chains_per_block - 1) / sh.chains_per_block * sh.chains_per_block;
            if ((long)nchains > resident && dbg.queue != 0) {
                HIP_TRY(launch_ctl_init(m->d_ctl, 0, 0x7fffffff, 0, 0, (int)resident, s));
                a.ctl = m->d_ctl; a.queue_slots = (int)resident;
            }
            a.perm = nullptr; a.place = nullptr;  // (latency mode: few chains, nothing to balance)
            rebase_plan_offsets(a.h);
            e = launch_q_phase(a, sg, sh.waves_per_block, 2, sr, lds, s, &cap, lean);
            a.h = m->h;
            a.ctl = nullptr; a.queue_slots = 0;
        }
    }
    for (; !cap && G <= 64; G *= 2) {
        a.free0p = free0_ordinal_p1(m, G);
        set_hinges_flag(m, a);
        const int flags_in = a.flags;
        const bool lean = lean_for(a, G, 0, false);
        const PlanHeader &gh = lean ? m->h3 : m->h;
        // The FK program (fixed-size step records, prefetched) costs LDS: stage it only if every level fits the
        // lane group and the workgroups that fit a CU stay the same; else the kernel walks the levels itself.
        const long waves_needed = ((long)nchains * G + 63) / 64;
        a.h = gh;
        a.h.total_words = m->h.core_words;
        a.h.plan_skip = 0;
        QShape sh{0, 2, 0};
        if (lean) {
            a.h = lean_header(a, true);
            sh = pick_shape(a.h, G, nkinds, waves_needed, true);
            if (!sh.wpb) { a.h = m->h; a.flags = flags_in; continue; }
        } else {
        sh = pick_shape(a.h, G, nkinds, waves_needed);
        if (!sh.wpb) continue;  // does not fit the LDS with this many chains per wave: widen the group
        a.flags |= 2;
        if (m->h.max_width <= G && !(dbg.flags >= 0 && (dbg.flags & 2))) {
            PlanHeader hp = m->h;
            // LDS diet of a program launch: no level tables / body records (they come first in the blob), and of the root
            // program (last) only the steps this call's trunk keypoints need -- or nothing without root optimisation
            if (m->h.off_lev_adr == 0 && m->h.off_body < m->h.off_joint && !dbg.nodiet) {
                hp.plan_skip = m->h.off_joint - 2 * kTTab;  // (from the momentum table on)
                const int root_words = (a.do_root_opt && a.n_mlev_root > 0) ? m->h.fk_hdr_words + a.n_mlev_root * m->h.max_width * m->h.fk_rec_words : 0;
                hp.total_words = m->h.off_fkroot + root_words;
            }
            const QShape shp = pick_shape(hp, G, nkinds, waves_needed);
            // same residency, or every wave resident anyway -- or at least 60 % of the waves: a wave on the program runs
            // its kinematics about twice as fast as one that walks the level tables (mouse, 85 levels: 3 waves per CU on
            // the program 24.0 k frames/s, 5 on the level loop 22.0 k)
            if (shp.wpb && (shp.waves_per_cu >= sh.waves_per_cu || (long)shp.waves_per_cu * kCus >= waves_needed ||
                            shp.waves_per_cu * 10 >= sh.waves_per_cu * 6)) {
                sh = shp; a.h.total_words = hp.total_words; a.h.plan_skip = hp.plan_skip; a.flags &= ~2;
            }
        }
        }
        if (dbg.wpe >= 0) {
            const int want = dbg.wpe >= 4 ? 4 : (dbg.wpe == 3 && G == 16) ? 3 : 2;
            if (!lean || q_phase_has_lean_variant(G, m->h.nq, want, 0)) sh.wpe = want;
        }
        if (dbg.wpb >= 0) {  // developer overrides
            const int ww = dbg.wpb;
            if (ww >= 1 && ww <= (sh.wpe == 3 ? 12 : 8) && q_lds_bytes(a.h, G, nkinds, ww) <= kLdsPerCu) sh.wpb = ww;
        }
        if (dbg.verbose)
            fprintf(stderr, "[stac] q_phase: chains=%d G=%d%s wpb=%d wpe=%d waves/CU=%d lds=%zu B/block chain_stride=%d floats plan=%d words\n",
                    nchains, G, lean ? " (lean)" : "", sh.wpb, sh.wpe, sh.waves_per_cu, q_lds_bytes(a.h, G, nkinds, sh.wpb), q_chain_stride(gh, G), a.h.total_words - a.h.plan_skip);
        a.mb_words = q_mb_words(nkinds, G);
        a.h.chain_stride = q_chain_stride(gh, G);
        // the latency kernel that takes the stragglers (hand-off below): lean like this launch where it can be
        QArgs b0 = a;
        b0.flags = a.flags & ~2;
        b0.free0p = free0_ordinal_p1(m, kLatG);
        set_hinges_flag(m, b0);
        const bool lean_b = lean_for(b0, kLatG, kLatR, true);
        const PlanHeader hb = lean_b ? lean_header(b0, false) : m->h;
        // Straggler hand-off: chains take very different numbers of iterations (the slowest of 10 000 about 1.6x the
        // mean), so the launch would end on a few waves per CU.  Once all but `hcap` chains are done, the rest move to
        // the latency kernel at their next iteration boundary (QArgs::ctl).
        int hcap = 0;
        if (!a.single && !(a.flags & 3) && (lean_b || m->h.max_width <= kLatG) && lat_one_wave_ok(m->h)) {
            const int spec_cap = (int)pick_spec_shape(hb, kLatG, nkinds, -1, kLatR).resident;
            // worth it while the tail is a sizeable part of the launch: up to about three rounds of resident chains
            // (15 % of the chains: 1 500 of 10 000 run like 2 000 -- 564-575 k against 567-572 k frames/s -- and move 1 MB less through HBM)
            if (nchains >= 4096 && (long)nchains <= 3L * sh.waves_per_cu * kCus * (64 / G)) hcap = std::min(spec_cap, nchains * 3 / 20);
            if (dbg.handoff >= 0) hcap = spec_cap ? std::min(dbg.handoff, nchains) : 0;
        }
        // Chain queue: when the launch has more chains than resident slots, the grid covers the resident slots only and
        // a group that finishes a chain takes the next unstarted one -- no slot idles until its whole workgroup is done.
        const long resident = (long)sh.waves_per_cu * kCus * (64 / G);
        int qslots = 0;
        if (!a.single && (long)nchains > resident && dbg.queue != 0) qslots = (int)resident;
        if (dbg.queue > 0) {  // developer / test override: this many slots (whole workgroups)
            const int per_block = sh.wpb * (64 / G), want = dbg.queue;
            if (!a.single && want < nchains) qslots = (want + per_block - 1) / per_block * per_block;
        }
        if (qslots > 0 && hcap == 0 && !(a.flags & 3) && (lean_b || m->h.max_width <= kLatG) && dbg.handoff < 0 && lat_one_wave_ok(m->h)) {
            hcap = std::min((int)pick_spec_shape(hb, kLatG, nkinds, -1, kLatR).resident, nchains / 5);  // with a queue the tail is one round: hand off
        }
        hcap = std::min(hcap, m->hand_cap);
        // A group that hands its chain off stops taking chains from the queue, so hand-off must not begin while the
        // queue could still hold more unstarted chains than the groups that stay: the threshold nchains - hcap is
        // only reached after every chain has STARTED when hcap <= qslots (at most qslots chains are in flight).
        if (qslots > 0) hcap = std::min(hcap, qslots);
        if (hcap > 0 || qslots > 0) {
            HIP_TRY(launch_ctl_init(m->d_ctl, 0, hcap > 0 ? nchains - hcap : 0x7fffffff, 0, hcap, qslots, s));
            a.ctl = m->d_ctl;
            a.queue_slots = qslots;
        }
        if (hcap > 0) a.hand = m->d_hand;
        const int root_fast = a.root_fast;
        if ((a.flags & 2) || G < root_fast) a.root_fast = 0;  // root fast trips need the FK program and the root coordinates in register 0
        const int32_t *perm_in = a.perm;
        int32_t *place_in = a.place;
        {
            const long waves = ((long)nchains * G + 63) / 64, simds = 4L * kCus;
            const bool all_resident = qslots == 0 && waves <= (long)sh.waves_per_cu * kCus;
            // (from ONE wavefront per SIMD upward: 6 144 chains = 1.5 per SIMD 23.5 -> 21.0 ms, profiles/r03/shape_sweep.txt)
            // (placement counts wavefronts per SIMD of a whole device: off on a partition / under a CU mask, where the census
            //  would wait for wavefronts that are not resident and every launch would sit out the time-out)
            if (all_resident && kCus == kFullCus && waves > simds && waves % simds != 0) a.place_crowded = (int)(waves / simds) + 1;
            else {
                a.place = nullptr;
                // with a chain queue the order still serves: the queue hands the chains out longest first, four of similar
                // length per wavefront (which takes its next four together: q_phase_kernel, ST_NEXT)
                if (!(qslots > 0 && a.root_fast > 0)) a.perm = nullptr;
            }
            if (dbg.verbose && a.place) fprintf(stderr, "[stac] q_phase: chains placed by SIMD load (crowded = %d wavefronts or more)\n", a.place_crowded);
        }
        {
            const size_t lds_bytes = q_lds_bytes(a.h, G, nkinds, sh.wpb);
            rebase_plan_offsets(a.h);
            e = launch_q_phase(a, G, sh.wpb, sh.wpe, 0, lds_bytes, s, &cap, lean);
            a.h = m->h;
        }
        a.root_fast = root_fast;
        a.perm = perm_in; a.place = place_in;
        if (cap && e == hipSuccess && hcap > 0) {
            // second launch: the latency kernel resumes whatever was handed off (waves without an entry exit at once)
            QArgs b = b0;
            b.ctl = a.ctl; b.hand = a.hand; b.root_fast = root_fast;
            b.queue_slots = 0; b.perm = nullptr; b.place = nullptr;
            b.resume = 1; b.resume_slots = hcap;
            b.h = hb;
            const SpecShape ss = pick_spec_shape(hb, kLatG, nkinds, hcap, kLatR);
            b.mb_words = q_mb_words(nkinds, kLatG);
            b.h.chain_stride = q_chain_stride(hb, kLatG);
            int cap2 = 0;
            const size_t lds2 = spec_lds_bytes(hb, kLatG, nkinds, ss.chains_per_block, kLatR);
            rebase_plan_offsets(b.h);
            e = launch_q_phase(b, kLatG, ss.waves_per_block, 2, kLatR, lds2, s, &cap2, lean_b);
            if (!cap2) return fail(STAC_ERR_CAPACITY, "hand-off: the latency kernel does not hold this model");
            if (dbg.verbose) fprintf(stderr, "[stac] q_phase: hand-off of up to %d stragglers to the latency kernel%s (wpb=%d)\n", hcap, lean_b ? " (lean)" : "", ss.waves_per_block);
        }
        a.ctl = nullptr; a.hand = nullptr; a.queue_slots = 0;
        if (cap) break;  // an instantiation with this many lanes holds nq
        a.flags = flags_in;
    }
    if (!cap) return fail(STAC_ERR_CAPACITY, "model exceeds the q_phase kernel limits (160 KiB LDS per CU, nq <= 256)");
    if (e != hipSuccess) return fail(STAC_ERR_HIP, std::string("q_phase launch: ") + hipGetErrorString(e));
#ifdef STAC_PROFILE
    {
        unsigned long long h[16];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, a.prof, sizeof(h), hipMemcpyDeviceToHost);
        static const char *names[] = {"loop", "stage", "fk", "sites", "loss_sum", "zero_gg", "joint_grad", "trans_sums", "accept_fused", "end_solve", "prepass", "-"};
        unsigned long long tot = 0;
        for (int i = 0; i < 11; ++i) tot += h[i];
        fprintf(stderr, "[stac profile] G=%d", G);
        for (int i = 0; i < 11; ++i) fprintf(stderr, " %s=%.1f%%", names[i], 100.0 * (double)h[i] / (double)(tot ? tot : 1));
        fprintf(stderr, " total_wave_cycles=%.3g wave_trips=%.4g cycles_per_wave_trip=%.0f fk_cycles_per_wave_trip=%.0f pure_root_trips=%.4g cycles_per_pure_root_trip=%.0f cycles_per_other_trip=%.0f\n", (double)tot, (double)h[11],
                (double)tot / (double)(h[11] ? h[11] : 1), (double)h[2] / (double)(h[11] ? h[11] : 1), (double)h[13],
                (double)h[12] / (double)(h[13] ? h[13] : 1), (double)(tot - h[12]) / (double)(h[11] - h[13] ? h[11] - h[13] : 1));
    }
#endif
    return STAC_OK;
}


// ---- LM solver: per-kind tables (dofs, non-zero J^T J entries, (site, dof) Jacobian items) ------------------------
static int build_lm_tables(stac_model *m, const uint8_t *masks /*[nkinds, nqpad]*/, int nkinds, float lambda0, hipStream_t s) {
    const PlanHeader &h = m->h;
    const int naj = h.naj, nab = h.nab;
    std::vector<int32_t> tab((size_t)nkinds * kLmKindWords, 0), hot, cold;
    int n_max = 0, maxpd_all = 1, npk_max = 4;
    auto is_anc_or_self = [&](int sa, int sb) {  // slot sa ancestor-or-equal of slot sb
        for (int s2 = sb; s2 >= 0; s2 = m->h_ab_parent[s2] - 1) if (s2 == sa) return true;
        return false;
    };
    for (int kind = 0; kind < nkinds; ++kind) {
        const uint8_t *mask = masks + (size_t)kind * h.nqpad;
        struct Dof { int qadr, joint, comp, pd; };
        std::vector<Dof> D;
        for (int j = 0; j < naj; ++j) {
            if (m->h_aj_shi[j] <= m->h_aj_slo[j]) continue;  // no fit site below this joint
            const int dims = m->h_aj_type[j] == STAC_JNT_FREE ? 7 : (m->h_aj_type[j] == STAC_JNT_BALL ? 4 : 1);
            for (int c = 0; c < dims; ++c)
                if (mask[m->h_aj_qadr[j] + c]) D.push_back({m->h_aj_qadr[j] + c, j, c, 0});
        }
        // qpos order (= the model's joint order: parents before children), whatever the order of the active joints in the plan:
        // the L^T D L pivots run through the coordinates backwards, and rounding makes their order part of the result --
        // oracle/stac_oracle.c::q_opt_lm_ws states the same order
        std::stable_sort(D.begin(), D.end(), [](const Dof &x, const Dof &y) { return x.qadr < y.qadr; });
        const int nd = (int)D.size();
        std::vector<int32_t> ents, items;
        std::vector<std::vector<int>> paths(nd);
        int maxpd = 1;
        struct Ent { int row, col, pds, range, len; };
        std::vector<Ent> E;
        for (int b = 0; b < nd; ++b) {
            const int jb = D[b].joint, sb = m->h_aj_slot[jb];
            int pd = 0;
            std::vector<int> path;
            for (int a2 = 0; a2 <= b; ++a2)
                if (a2 == b || is_anc_or_self(m->h_aj_slot[D[a2].joint], sb)) path.push_back(a2);
            pd = (int)path.size() - 1;
            D[b].pd = pd;
            paths[b] = path;
            maxpd = std::max(maxpd, pd + 1);
            const int lo = m->h_aj_slo[jb], hi = m->h_aj_shi[jb];
            for (int pi = 0; pi < (int)path.size(); ++pi)
                E.push_back({b, path[pi], pd | (pi << 8), lo | (hi << 16), hi - lo});
            for (int i = lo; i < hi; ++i) { items.push_back(i); items.push_back(b); items.push_back(pd); items.push_back(0); }
        }
        if (maxpd > 255) return fail(STAC_ERR_CAPACITY, "kinematic path too deep for the LM solver");
        std::stable_sort(E.begin(), E.end(), [](const Ent &x, const Ent &y) { return x.len > y.len; });
        for (const Ent &e : E) { ents.push_back(e.row); ents.push_back(e.col); ents.push_back(e.pds); ents.push_back(e.range); }
        // raw quaternions (the free root's, every ball joint's) of which all four components are optimised: index of their first dof
        std::vector<int32_t> quats;
        for (int b = 0; b + 3 < nd; ++b) {
            const int ty = m->h_aj_type[D[b].joint], c0 = ty == STAC_JNT_FREE ? 3 : 0;
            if ((ty == STAC_JNT_FREE || ty == STAC_JNT_BALL) && D[b].comp == c0 && D[b + 3].joint == D[b].joint && D[b + 3].comp == c0 + 3)
                quats.push_back(b);
        }
        const size_t kho = (size_t)kind * kLmKindWords;
        tab[kho + 0] = nd; tab[kho + 1] = (int)E.size(); tab[kho + 2] = (int)items.size() / 4; tab[kho + 3] = maxpd;
        tab[kho + 7] = (int)quats.size();
        // hot section (LDS): dof records + path table; cold section (global): entries + items
        tab[kho + 4] = (int)hot.size();
        for (const Dof &d : D) { hot.push_back(d.qadr); hot.push_back(d.joint); hot.push_back(d.comp); hot.push_back(d.pd); }
        tab[kho + 8] = (int)hot.size();
        for (int b = 0; b < nd; ++b)
            for (int pi = 0; pi < maxpd; ++pi) hot.push_back(pi < (int)paths[b].size() ? paths[b][pi] : -1);
        tab[kho + 9] = (int)hot.size();
        hot.insert(hot.end(), quats.begin(), quats.end());
        while (hot.size() & 3) hot.push_back(0);
        tab[kho + 5] = (int)cold.size();
        cold.insert(cold.end(), ents.begin(), ents.end());
        tab[kho + 6] = (int)cold.size();
        cold.insert(cold.end(), items.begin(), items.end());
        n_max = std::max(n_max, nd);
        maxpd_all = std::max(maxpd_all, maxpd);
        npk_max = std::max(npk_max, nd * maxpd);
    }
    (void)nab;
    const int hot_off = (int)tab.size(), hot_words = (int)hot.size();
    const int cold_off = hot_off + hot_words;
    for (int kind = 0; kind < nkinds; ++kind) {  // cold offsets become absolute
        tab[(size_t)kind * kLmKindWords + 5] += cold_off;
        tab[(size_t)kind * kLmKindWords + 6] += cold_off;
    }
    tab.insert(tab.end(), hot.begin(), hot.end());
    tab.insert(tab.end(), cold.begin(), cold.end());
    if (tab != m->lm_tab_cache) {
        if (tab.size() > m->lm_tab_words) {
            if (m->d_lm_tab) (void)hipFree(m->d_lm_tab);
            m->d_lm_tab = nullptr;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&m->d_lm_tab), tab.size() * sizeof(int32_t)));
            m->lm_tab_words = tab.size();
        }
        HIP_TRY(hipMemcpyAsync(m->d_lm_tab, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        m->lm_tab_cache = tab;
    }
    LmArgs &L = m->lm_args;
    L.tab = m->d_lm_tab;
    L.nkinds = nkinds;
    L.hot_off = hot_off;
    L.hot_words = hot_words;
    L.n_max = std::max(n_max, 1);
    if (L.n_max > 192) return fail(STAC_ERR_CAPACITY, "more than 192 optimised coordinates: not supported by the LM solver");
    L.npk = (npk_max + 3) & ~3;
    L.maxpd = maxpd_all;
    L.lambda0 = lambda0 > 0.0f ? lambda0 : 1e-2f;
    int o = h.stride_lds;  // the PG layout with the keypoints in LDS
    o = (o + 3) & ~3;
    L.c_qe = o; o += h.nqpad;  // the LM kernel reads the evaluation point after the site pass: not aliased
    L.c_gg = o; o += std::max(h.nqpad, (h.K + 3) & ~3);  // and keeps the per-site loss terms and the gradient apart from the transforms
    o = (o + 3) & ~3;
    L.c_sx = o; o += 3 * h.K;
    o = (o + 3) & ~3;
    L.c_jp = o; o += std::max(h.K * L.maxpd * 3, L.npk);
    o = (o + 3) & ~3;
    L.c_A = o; o += L.npk;
    L.c_b = o; o += L.n_max;
    L.c_d = o; o += L.n_max + 2 * L.maxpd;  // step vector + two scratch rows of the L^T D L pivot loop
    L.c_fz = o; o += L.n_max;
    L.chain_stride = kXf == 8 ? 4 * (((o + 3) / 4) | 1) : (o | 1);  // 16-byte aligned regions, chains spread over the banks
    return STAC_OK;
}

static int run_q_lm(stac_model *m, const stac_q_params *p, QArgs &a, int nchains, const uint8_t *host_masks, hipStream_t s) {
    if (p->maxiter < 1) return fail(STAC_ERR_INVALID, "maxiter must be >= 1");
    const int nkinds = a.P + 3;
    const int rc = build_lm_tables(m, host_masks, nkinds, p->lm_lambda0, s);
    if (rc != STAC_OK) return rc;
    a.hdr = nullptr;
    a.plan = m->d_blob;
    a.h = m->h;  // total_words (what the launch stages in LDS) is settled with the launch shape below
    a.h.c_qe = m->lm_args.c_qe;
    a.h.c_gg = m->lm_args.c_gg;
    a.tol = p->tol; a.maxiter = p->maxiter; a.maxls = p->maxls;
    const LmArgs &L = m->lm_args;
#ifdef STAC_PROFILE
    static unsigned long long *d_prof = nullptr;
    if (!d_prof) { (void)hipMalloc(reinterpret_cast<void **>(&d_prof), 16 * sizeof(unsigned long long)); }
    (void)hipMemsetAsync(d_prof, 0, 16 * sizeof(unsigned long long), s);
    a.prof = d_prof;
#endif
    int G = (p->lanes_per_chain == 16 || p->lanes_per_chain == 32 || p->lanes_per_chain == 64) ? p->lanes_per_chain
            : 64;   // measured: the whole wave on one chain is fastest at every batch size
    hipError_t e = hipErrorInvalidValue;
    int cap = 0;
    for (; !cap && G <= 64; G *= 2) {
        const int cpw = 64 / G;
        const int mbw = (2 * nkinds * G + 3) & ~3, khw = ((nkinds * kLmKindWords + 3) & ~3) + ((L.hot_words + 3) & ~3) +
                                                   ((((L.maxpd * (L.maxpd + 1)) >> 1) + 3) & ~3);
        int plan_words = m->h.core_words;
        auto lds_for = [&](int wpb) { return (size_t)(((plan_words + 3) & ~3) + mbw + khw + wpb * cpw * L.chain_stride) * sizeof(float); };
        const int wps = lm_waves_per_simd(G, m->h.nq);
        auto best_shape = [&](int &wpb_out) {
            int best = 0;
            wpb_out = 0;
            for (int w = 1; w <= 8; ++w) {
                size_t lds = lds_for(w);
                if (lds > kLdsPerCu) break;
                lds = (lds + 1279) / 1280 * 1280;
                int blocks = (int)(kLdsPerCu / lds);
                if (w <= 4) { if (blocks * w > 4 * wps) blocks = 4 * wps / w; }
                else { const int per_simd = (w + 3) / 4; if (blocks * per_simd > wps) blocks = wps / per_simd; }
                if (blocks * w > best) { best = blocks * w; wpb_out = w; }
            }
            return best;
        };
        int wpb = 0;
        int best_waves = best_shape(wpb);
        if (!wpb) continue;
        // the FK program (stac_plan.hpp) is staged only if every level fits the lane group and it costs no occupancy
        a.flags |= 2;
        a.h.total_words = m->h.core_words;
        if (m->h.max_width <= G && !(m->dbg.flags >= 0 && (m->dbg.flags & 2))) {
            plan_words = m->h.total_words;
            int wpb2 = 0;
            const int waves2 = best_shape(wpb2);
            if (wpb2 && waves2 >= best_waves) { wpb = wpb2; best_waves = waves2; a.flags &= ~2; a.h.total_words = m->h.total_words; }
            else plan_words = m->h.core_words;
        }
        a.mb_words = mbw;
        if (m->dbg.verbose)
            fprintf(stderr, "[stac] q_phase LM: chains=%d G=%d wpb=%d waves/CU=%d lds=%zu B/block chain_stride=%d n_max=%d maxpd=%d\n",
                    nchains, G, wpb, best_waves, lds_for(wpb), L.chain_stride, L.n_max, L.maxpd);
        // chain queue (see run_q): the grid covers the resident slots, finished groups take the next chain
        a.ctl = nullptr; a.queue_slots = 0;
        long resident = (long)best_waves * kCus * cpw;
        if (m->dbg.queue > 0) {  // developer / test override: this many slots (whole workgroups)
            const int per_block = wpb * cpw, want = m->dbg.queue;
            if (want < nchains) resident = (long)(want + per_block - 1) / per_block * per_block;
        }
        if ((long)nchains > resident && m->dbg.queue != 0) {
            HIP_TRY(launch_ctl_init(m->d_ctl, 0, 0, 0, 0, (int)resident, s));
            a.ctl = m->d_ctl; a.queue_slots = (int)resident;
        }
        e = launch_q_phase_lm(a, L, G, wpb, lds_for(wpb), s, &cap);
        a.ctl = nullptr; a.queue_slots = 0;
    }
    if (!cap) return fail(STAC_ERR_CAPACITY, "model exceeds the LM q_phase kernel limits (LDS per CU / nq)");
    if (e != hipSuccess) return fail(STAC_ERR_HIP, std::string("q_phase LM launch: ") + hipGetErrorString(e));
#ifdef STAC_PROFILE
    {
        unsigned long long hh[16];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(hh, a.prof, sizeof(hh), hipMemcpyDeviceToHost);
        static const char *names[] = {"loop", "stage_fk", "sites_loss", "gradient", "accept", "jac_blocks", "jtj", "damp", "ltdl", "fwd", "trial", "end"};
        unsigned long long tot = 0;
        for (int i = 0; i < 12; ++i) tot += hh[i];
        fprintf(stderr, "[stac profile LM]");
        for (int i = 0; i < 12; ++i) fprintf(stderr, " %s=%.1f%%", names[i], 100.0 * (double)hh[i] / (double)(tot ? tot : 1));
        fprintf(stderr, " total_wave_cycles=%.3g\n", (double)tot);
    }
#endif
    return STAC_OK;
}

// Root passes weigh the trunk keypoints only: every other site contributes exact zeros there, so the kinematics of
// the bodies that carry no trunk keypoint are not needed.  Builds the FK program of the needed bodies into the
// blob's second program area; the kernel runs it when every chain of a wavefront is in a root pass.
static int fill_root_program(stac_model *m, const uint8_t *trunk_kps, bool enable, int n_root_joints, hipStream_t s) {
    const PlanHeader &h = m->h;
    m->n_mlev_root = 0;
    m->fk3r = 0;
    if (!enable || m->dbg.noprune) return STAC_OK;
    std::vector<char> need(h.nab, 0);
    int n_need = 0;
    for (int k = 0; k < h.K; ++k)
        if (trunk_kps[k])
            for (int sl = m->h_site_slot[k]; sl >= 0 && !need[sl]; sl = m->h_ab_parent[sl] - 1) { need[sl] = 1; ++n_need; }
    if (n_need == 0 || n_need == h.nab) return STAC_OK;  // nothing to prune (or nothing weighted at all)
    int n_mlev = 0;
    // Only the root joints' gradients are evaluated in a pruned trip: every other joint's {anchor, pre-joint quaternion}
    // goes to the sink, so the joint-local quaternion the pre-pass parked in its entry survives the trip (root fast trips)
    int n_run = 0;
    const std::vector<int32_t> prog = build_fk_program(m, need.data(), &n_mlev, nullptr, n_root_joints, &n_run);
    int32_t *dst = reinterpret_cast<int32_t *>(m->blob_host.data()) + h.off_fkroot;
    std::memcpy(dst, prog.data(), prog.size() * 4);
    HIP_TRY(hipMemcpyAsync(m->d_blob + h.off_fkroot, dst, prog.size() * 4, hipMemcpyHostToDevice, s));
    // the four-lanes-per-position program stops after its last step with work; the other forms run whole pairs
    m->n_mlev_root = n_mlev;
    m->n_run_root = n_run;
    if (h.fk3) {  // the same bodies as a split-kinematics program (lean launches)
        Fk3Program rp;
        if (build_fk3_program(m, m->h3, need.data(), h.fk3_cap1, h.fk3_cap2, h.fk3_cap3, rp) && rp.n3 > 0) {
            int32_t *d3 = reinterpret_cast<int32_t *>(m->blob_host.data()) + h.off3_root;
            std::memcpy(d3, rp.words.data(), rp.words.size() * 4);
            HIP_TRY(hipMemcpyAsync(m->d_blob + h.off3_root, d3, rp.words.size() * 4, hipMemcpyHostToDevice, s));
            m->fk3r = rp.n1 | rp.n3 << 8 | rp.n2 << 16;
        }
    }
    return STAC_OK;
}

extern "C" int32_t stac_q_solve(const stac_model *mc, const stac_q_params *p, const float *kp, const float *q0,
                                const uint8_t *qs_to_opt, const uint8_t *kps_to_opt, const float *lb, const float *ub,
                                int32_t N, float *params_out, float *state_out, uint32_t *counters_out, void *stream) {
    stac_model *m = const_cast<stac_model *>(mc);
    if (!m || !p || !qs_to_opt || !kps_to_opt || N < 0) return fail(STAC_ERR_INVALID, "stac_q_solve: bad argument");
    if ((lb == nullptr) != (ub == nullptr)) return fail(STAC_ERR_INVALID, "stac_q_solve: lb and ub must be given together");
    if (N == 0) return STAC_OK;
    if (!kp || !q0 || !params_out || !state_out) return fail(STAC_ERR_INVALID, "stac_q_solve: bad argument");
    if (p->solver != STAC_SOLVER_PG) return fail(STAC_ERR_INVALID, "stac_q_solve implements the reference's projected gradient only");
    DeviceGuard dg(m);
    hipStream_t s = (hipStream_t)stream;
    const int nqpad = m->h.nqpad, K = m->h.K, nq = m->h.nq;
    const float *d_bounds = nullptr;
    if (lb) {  // per-call box (StacCore.q_opt takes lb / ub per call, stac_core.py:193-235); NULL = the model's
        std::vector<float> hb(2 * (size_t)nqpad, 0.0f);
        std::memcpy(hb.data(), lb, nq * sizeof(float));
        std::memcpy(hb.data() + nqpad, ub, nq * sizeof(float));
        for (int i = 0; i < nq; ++i)
            if (!(hb[i] <= hb[nqpad + i])) return fail(STAC_ERR_INVALID, "stac_q_solve: lb > ub (or NaN) at coordinate " + std::to_string(i));
        if (hb != m->bounds_cache) {
            HIP_TRY(hipMemcpyAsync(m->d_bounds, hb.data(), hb.size() * sizeof(float), hipMemcpyHostToDevice, s));
            HIP_TRY(hipStreamSynchronize(s));  // hb is a temporary
            m->bounds_cache = hb;
        }
        d_bounds = m->d_bounds;
    }
    std::vector<uint8_t> hostm((size_t)nqpad + 3 * K + 1, 0);
    std::memcpy(hostm.data(), qs_to_opt, nq);
    std::memcpy(hostm.data() + nqpad, kps_to_opt, 3 * K);
    hostm.back() = 0xA5;  // tag: single-solve layout
    uint8_t *d_kpw3 = m->d_masks + (size_t)kMaxKinds * nqpad + K;
    if (hostm != m->masks_cache) {
        HIP_TRY(hipMemcpyAsync(m->d_masks, hostm.data(), nqpad, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_kpw3, hostm.data() + nqpad, 3 * K, hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));  // hostm is a temporary
        m->masks_cache = hostm;
    }
    QArgs a{};
    a.kp = kp; a.q_init = q0; a.masks = m->d_masks; a.kpw = nullptr; a.kpw3 = d_kpw3;
    a.C = N; a.F = 1; a.P = 0; a.root_kp_idx = 0; a.do_root_opt = 0; a.single = 1;
    a.bounds = d_bounds;
    a.qpos_out = params_out; a.err_out = state_out; a.counters_out = counters_out; a.q_carry_out = nullptr;
    return run_q(m, p, a, N, s);
}

extern "C" int32_t stac_q_phase(const stac_model *mc, const stac_q_params *p, const float *kp, const float *q_init,
                                const uint8_t *part_masks, const uint8_t *trunk_kps, int32_t C, int32_t F, int32_t P,
                                int32_t root_kp_idx, int32_t root_dims, int32_t do_root_opt, float *qpos_out,
                                float *err_out, uint32_t *counters_out, float *q_carry_out, float *xpos_out,
                                float *xquat_out, float *markers_out, void *stream) {
    stac_model *m = const_cast<stac_model *>(mc);
    if (!m || !p || C < 0 || F < 0 || P < 0) return fail(STAC_ERR_INVALID, "stac_q_phase: bad argument");
    if (C == 0 || F == 0) return STAC_OK;  // nothing to do (a rank of a sharded run that owns no clip): arrays may be empty
    if (!kp || !qpos_out || !err_out) return fail(STAC_ERR_INVALID, "stac_q_phase: bad argument");
    if (P > 0 && !part_masks) return fail(STAC_ERR_INVALID, "stac_q_phase: part_masks is null");
    if (P + 3 > kMaxKinds) return fail(STAC_ERR_CAPACITY, "too many part groups");
    if (do_root_opt && (!trunk_kps || root_kp_idx < 0 || root_kp_idx >= m->h.K || root_dims < 1 || root_dims > m->h.nq))
        return fail(STAC_ERR_INVALID, "stac_q_phase: bad root optimisation arguments");
    DeviceGuard dg(m);
    hipStream_t s = (hipStream_t)stream;
    const int nqpad = m->h.nqpad, K = m->h.K, nq = m->h.nq;
    // mask table: rows 0,1 = root passes (first root_dims qpos), row 2 = all, rows 3.. = parts
    std::vector<uint8_t> hostm((size_t)(P + 3) * nqpad + K, 0);
    for (int i = 0; i < nq; ++i) {
        hostm[i] = hostm[nqpad + i] = (do_root_opt && i < root_dims) ? 1 : 0;
        hostm[2 * (size_t)nqpad + i] = 1;
    }
    for (int pi = 0; pi < P; ++pi) std::memcpy(hostm.data() + (size_t)(3 + pi) * nqpad, part_masks + (size_t)pi * nq, nq);
    uint8_t *d_kpw = m->d_masks + (size_t)kMaxKinds * nqpad;
    for (int k = 0; k < K; ++k) hostm[(size_t)(P + 3) * nqpad + k] = (trunk_kps && trunk_kps[k]) ? 1 : 0;
    for (int k = 0; k < K; ++k) hostm.push_back((trunk_kps && trunk_kps[k]) ? 1 : 0);  // placeholder, reordered below
    for (int k = 0; k < K; ++k) hostm[(size_t)(P + 3) * nqpad + K + m->h_sortpos[k]] = (trunk_kps && trunk_kps[k]) ? 1 : 0;
    hostm.push_back(0x5A);  // tag: phase layout
    // joints whose coordinates the root passes optimise: a prefix of the active joints (the root body comes first);
    // the gradient pass of a pruned root-pass trip stops there (the other gradients are masked out anyway)
    int n_root_joints = m->h.naj;
    if (do_root_opt) {
        int nrj = 0;
        while (nrj < m->h.naj && m->h_aj_qadr[nrj] < root_dims) ++nrj;
        bool prefix = true;
        for (int j = nrj; j < m->h.naj; ++j) prefix = prefix && m->h_aj_qadr[j] >= root_dims;
        if (prefix) n_root_joints = nrj;
    }
    if (hostm != m->masks_cache) {
        HIP_TRY(hipMemcpyAsync(m->d_masks, hostm.data(), (size_t)(P + 3) * nqpad, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_kpw, hostm.data() + (size_t)(P + 3) * nqpad, 2 * K, hipMemcpyHostToDevice, s));
        m->masks_cache.clear();
        const int rc = fill_root_program(m, hostm.data() + (size_t)(P + 3) * nqpad, do_root_opt != 0, n_root_joints, s);
        if (rc != STAC_OK) return rc;
        HIP_TRY(hipStreamSynchronize(s));  // hostm is a temporary
        m->masks_cache = hostm;
    }
    QArgs a{};
    a.kp = kp; a.q_init = q_init; a.masks = m->d_masks; a.kpw = d_kpw; a.kpw3 = nullptr;
    a.C = C; a.F = F; a.P = P; a.root_kp_idx = root_kp_idx; a.do_root_opt = do_root_opt ? 1 : 0; a.single = 0;
    a.n_mlev_root = do_root_opt ? m->n_mlev_root : 0;
    a.n_run_root = do_root_opt ? m->n_run_root : 0;
    a.n_root_joints = n_root_joints;
    if (do_root_opt && m->h.fk3) a.fk3r_n = m->fk3r;
    // Root fast trips (QArgs::root_fast): the root coordinates are the first root_dims (<= 8: register 0 of every lane
    // group of 8 or more lanes) and belong to leading joints that share ONE subtree range, whose weighted sites fit a 64-bit mask
    if (do_root_opt && m->n_mlev_root > 0 && n_root_joints >= 1 && n_root_joints < m->h.naj && root_dims <= 8 && K <= 64 && !m->dbg.nofast) {
        bool same = true;
        for (int j = 1; j < n_root_joints; ++j) same = same && m->h_aj_slo[j] == m->h_aj_slo[0] && m->h_aj_shi[j] == m->h_aj_shi[0];
        int covered = 0;
        for (int j = 0; j < n_root_joints; ++j) covered += m->h_aj_type[j] == STAC_JNT_FREE ? 7 : (m->h_aj_type[j] == STAC_JNT_BALL ? 4 : 1);
        if (same && covered == root_dims) {
            uint64_t mask = 0;
            const uint8_t *ts = hostm.data() + (size_t)(P + 3) * nqpad + K;  // trunk mask by sorted-site position
            for (int i = m->h_aj_slo[0]; i < m->h_aj_shi[0]; ++i)
                if (ts[i]) mask |= 1ull << i;
            a.root_fast = root_dims;
            a.root_trunk_lo = (uint32_t)mask; a.root_trunk_hi = (uint32_t)(mask >> 32);

        }
    }
    // Chain order + placement (QArgs::perm / place): large batches of short clips (the root solves are where chain lengths
    // differ), rest pose as the start (no q_init), a free root at qpos 0 .. 6.  run_q keeps them only for a throughput launch
    // whose wavefronts are all resident and unevenly spread over the SIMDs.
    if (a.root_fast > 0 && m->h_aj_type[0] == STAC_JNT_FREE && m->h_aj_qadr[0] == 0 && !q_init && C >= kOrderMinChains &&
        F <= kOrderMaxFrames && !m->dbg.noorder && p->solver == STAC_SOLVER_PG) {
        const size_t words = (size_t)3 * K + 4096 + kPlaceWords + 2 * (size_t)C;
        if ((size_t)C > m->order_chains) {
            if (m->d_order) (void)hipFree(m->d_order);
            m->d_order = nullptr; m->order_chains = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&m->d_order), words * sizeof(uint32_t)));
            m->order_chains = (size_t)C;
        }
        float *rest = reinterpret_cast<float *>(m->d_order);
        uint32_t *hist = m->d_order + 3 * K;
        int32_t *place = reinterpret_cast<int32_t *>(hist + 4096);
        uint32_t *keybits = reinterpret_cast<uint32_t *>(place + kPlaceWords);
        int32_t *perm = reinterpret_cast<int32_t *>(keybits + m->order_chains);
        const int rc2 = fk_impl(m, m->d_qpos0, 1, nullptr, nullptr, nullptr, rest, 1, stream);  // marker sites of the rest pose
        if (rc2 != STAC_OK) return rc2;
        const float *q0h = m->blob_host.data() + m->h.off_qpos0;  // rest position of the root body = its free joint's qpos0
        HIP_TRY(launch_chain_order(kp, C, F, K, rest, d_kpw, root_kp_idx, q0h[0], q0h[1], q0h[2], hist, keybits, perm, place, s));
        a.perm = perm; a.place = place;
    }
    a.qpos_out = qpos_out; a.err_out = err_out; a.counters_out = counters_out; a.q_carry_out = q_carry_out;
    a.kpw_sorted = d_kpw + K;
    const int rc = p->solver == STAC_SOLVER_LM ? run_q_lm(m, p, a, C, hostm.data(), s) : run_q(m, p, a, C, s);
    if (rc != STAC_OK) return rc;
    if (xpos_out || xquat_out || markers_out)
        // qpos_out already holds kinematics' normalised quaternions: use them as they are
        return fk_impl(m, qpos_out, C * F, nullptr, xpos_out, xquat_out, markers_out, 0, stream);
    return STAC_OK;
}

extern "C" int64_t stac_m_phase_workspace_floats(const stac_model *m, int32_t T) {
    if (!m || T < 0) return -1;
    return (int64_t)T * ((int64_t)m->h.nbody * 7 + 3 * m->h.K + 1);
}

extern "C" int32_t stac_m_phase_partial(const stac_model *m, const float *kp, const float *q, int32_t T,
                                        float *workspace, float *partial, void *stream) {
    if (!m || !workspace || !partial || T < 0 || (T > 0 && (!kp || !q)))  // T = 0: kp / q may be empty (null) arrays
        return fail(STAC_ERR_INVALID, "stac_m_phase_partial: bad argument");
    DeviceGuard dg(m);
    hipStream_t s = (hipStream_t)stream;
    const size_t nb = m->h.nbody;
    float *xpos = workspace, *xquat = workspace + (size_t)T * nb * 3, *contrib = workspace + (size_t)T * nb * 7;
    if (T > 0) HIP_TRY(launch_fk(m->full(), q, T, nullptr, xpos, xquat, nullptr, 1, s));
    HIP_TRY(launch_m_partial(m->full(), kp, xpos, xquat, T, contrib, partial, s));
    return STAC_OK;
}

extern "C" int32_t stac_m_phase_finish(const stac_model *m, const float *partial, const float *initial_offsets,
                                       const float *is_regularized, float reg_coef, float *offsets_out,
                                       float *err_out, void *stream) {
    if (!m || !partial || !initial_offsets || !is_regularized || !offsets_out)
        return fail(STAC_ERR_INVALID, "stac_m_phase_finish: bad argument");
    DeviceGuard dg(m);
    HIP_TRY(launch_m_finish(m->h.K, partial, initial_offsets, is_regularized, reg_coef, offsets_out, err_out,
                            (hipStream_t)stream));
    return STAC_OK;
}
