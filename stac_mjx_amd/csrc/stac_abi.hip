// stac_abi.hip -- host side of libstac_hip.so: plan construction and the C ABI of include/stac_hip.h.
//
// Builds the "plan" (marker-ancestor subtree, level schedule, CSR site/child lists) from the flat
// model tables, keeps it in device memory and launches the kernels of stac_kernels.hip.
// No CPU fallback exists: every compute entry point needs a GPU and fails loudly without one.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/stac_hip.h"
#include "stac_plan.hpp"

namespace stac {
hipError_t launch_q_phase(const QArgs &a, int G, size_t lds_bytes, hipStream_t s, int *capacity_out);
hipError_t launch_fk(const FullModel &M, const float *qpos, int N, float *qn, float *xpos, float *xquat,
                     float *site_xpos, int normalize, hipStream_t s);
hipError_t launch_m_partial(const FullModel &M, const float *kp, const float *xpos, const float *xquat, int T,
                            float *contrib, float *partial, hipStream_t s);
hipError_t launch_m_finish(int K, const float *partial, const float *m0, const float *dreg, float lam, float *out,
                           float *err, hipStream_t s);
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

struct stac_model {
    PlanHeader h{};
    std::vector<float> blob_host;  // plan blob (host mirror)
    float *d_blob = nullptr;       // device plan blob
    // full tables on device
    int32_t *d_body_parentid = nullptr, *d_body_jntadr = nullptr, *d_body_jntnum = nullptr;
    float *d_body_pos = nullptr, *d_body_quat = nullptr;
    int32_t *d_jnt_type = nullptr, *d_jnt_qposadr = nullptr;
    float *d_jnt_pos = nullptr, *d_jnt_axis = nullptr, *d_qpos0 = nullptr;
    int32_t *d_site_bodyid = nullptr;
    uint8_t *d_masks = nullptr;  // [kMaxKinds, nqpad] + [K] + [3K]
    size_t masks_bytes = 0;
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
        M.site_pos = d_blob + h.off_site_pos;
        return M;
    }
};

template <typename T>
static hipError_t upload(T **dst, const T *src, size_t n) {
    hipError_t e = hipMalloc(reinterpret_cast<void **>(dst), std::max<size_t>(n, 1) * sizeof(T));
    if (e != hipSuccess) return e;
    if (n) e = hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice);
    return e;
}

static int ensure_scratch(stac_model *m, size_t floats) {
    if (floats <= m->scratch_floats) return STAC_OK;
    if (m->d_scratch) (void)hipFree(m->d_scratch);
    m->d_scratch = nullptr;
    m->scratch_floats = 0;
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
    std::vector<int> slots;  // body ids sorted by (depth, id)
    for (int b = 1; b < nb; ++b)
        if (active[b]) slots.push_back(b);
    std::stable_sort(slots.begin(), slots.end(), [&](int a, int b) { return depth[a] < depth[b]; });
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
    // own sites (ascending id) and children (descending body id)
    std::vector<int> ab_sadr(nab), ab_snum(nab), site_list, ab_cadr(nab), ab_cnum(nab), child_list, ab_parent(nab);
    for (int s = 0; s < nab; ++s) {
        const int b = slots[s];
        ab_sadr[s] = (int)site_list.size();
        for (int k = 0; k < K; ++k)
            if (t->site_bodyid[k] == b) site_list.push_back(k);
        ab_snum[s] = (int)site_list.size() - ab_sadr[s];
        ab_cadr[s] = (int)child_list.size();
        for (int c = nb - 1; c >= 1; --c)
            if (active[c] && t->body_parentid[c] == b) child_list.push_back(slot_of[c]);
        ab_cnum[s] = (int)child_list.size() - ab_cadr[s];
        const int p = t->body_parentid[b];
        ab_parent[s] = p == 0 ? 0 : slot_of[p] + 1;
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

    std::vector<float> &B = m->blob_host;
    B.clear();
    auto put_i = [&](const std::vector<int> &v, size_t pad_to = 0) {
        int off = (int)B.size();
        for (int x : v) { float f; std::memcpy(&f, &x, 4); B.push_back(f); }
        while (B.size() < off + pad_to) B.push_back(0.f);
        return off;
    };
    auto put_f = [&](const std::vector<float> &v, size_t pad_to = 0) {
        int off = (int)B.size();
        for (float x : v) B.push_back(x);
        while (B.size() < off + pad_to) B.push_back(0.f);
        return off;
    };
    std::vector<float> ab_pos, ab_quat;
    for (int s = 0; s < nab; ++s) {
        for (int i = 0; i < 3; ++i) ab_pos.push_back(t->body_pos[3 * slots[s] + i]);
        for (int i = 0; i < 4; ++i) ab_quat.push_back(t->body_quat[4 * slots[s] + i]);
    }
    std::vector<int> site_slot(K);
    for (int k = 0; k < K; ++k) site_slot[k] = slot_of[t->site_bodyid[k]];
    h.off_lev_adr = put_i(lev_adr);
    h.off_ab_parent = put_i(ab_parent);
    h.off_ab_jadr = put_i(ab_jadr);
    h.off_ab_jnum = put_i(ab_jnum);
    h.off_ab_sadr = put_i(ab_sadr);
    h.off_ab_snum = put_i(ab_snum);
    h.off_ab_cadr = put_i(ab_cadr);
    h.off_ab_cnum = put_i(ab_cnum);
    h.off_site_list = put_i(site_list);
    h.off_child_list = put_i(child_list, 1);
    h.off_ab_pos = put_f(ab_pos);
    h.off_ab_quat = put_f(ab_quat);
    h.off_aj_type = put_i(aj_type, 1);
    h.off_aj_qadr = put_i(aj_qadr, 1);
    h.off_aj_slot = put_i(aj_slot, 1);
    h.off_aj_pos = put_f(aj_pos, 1);
    h.off_aj_axis = put_f(aj_axis, 1);
    h.off_aj_q0 = put_f(aj_q0, 1);
    h.off_site_slot = put_i(site_slot);
    h.off_site_pos = put_f(std::vector<float>(t->site_pos, t->site_pos + 3 * K));
    h.off_lb = put_f(std::vector<float>(t->lb, t->lb + nq), h.nqpad);
    h.off_ub = put_f(std::vector<float>(t->ub, t->ub + nq), h.nqpad);
    h.off_qpos0 = put_f(std::vector<float>(t->qpos0, t->qpos0 + nq), h.nqpad);
    h.off_quat_adr = put_i(quat_adr, 1);
    h.total_words = (int)B.size();

    // per-chain LDS layout
    int o = 0;
    h.c_bx = o; o += (nab + 1) * 7;
    h.c_ja = o; o += naj * 6;
    h.c_jq = o; o += has_ball ? naj * 4 : 0;
    h.c_jn = o; o += naj;
    h.c_sw = o; o += std::max(K * 6, h.nqpad);
    h.c_bw = o; o += std::max(std::max(nab * 6, 2 * h.nqpad), K);
    o = (o + 3) & ~3;
    h.c_qe = o; o += h.nqpad;
    h.c_kp = o; o += 3 * K;
    // odd stride (mod 32 banks) so that the chains of one wavefront hit different LDS banks
    if ((o & 1) == 0) o += 1;
    h.chain_stride = o;
    return STAC_OK;
}

static int q_mb_words(int nkinds, int G) { return (nkinds * G + 3) & ~3; }
static size_t q_lds_bytes(const PlanHeader &h, int G, int nkinds) {
    const int plan_words = (h.total_words + 3) & ~3;
    return (size_t)(plan_words + q_mb_words(nkinds, G) + (64 / G) * h.chain_stride) * sizeof(float);
}

extern "C" stac_model *stac_model_create(const stac_model_tables *t) {
    if (!t) { fail(STAC_ERR_INVALID, "null tables"); return nullptr; }
    if (stac_device_count() <= 0) {
        fail(STAC_ERR_NO_DEVICE, "no HIP device visible: the STAC engine has no CPU fallback");
        return nullptr;
    }
    stac_model *m = new stac_model();
    if (build_plan(m, t) != STAC_OK) { delete m; return nullptr; }
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
    m->masks_bytes = (size_t)kMaxKinds * m->h.nqpad + 4 * (size_t)K + 64;
    chk(hipMalloc(reinterpret_cast<void **>(&m->d_masks), m->masks_bytes));
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
    void *ptrs[] = {m->d_blob, m->d_body_parentid, m->d_body_jntadr, m->d_body_jntnum, m->d_body_pos,
                    m->d_body_quat, m->d_jnt_type, m->d_jnt_qposadr, m->d_jnt_pos, m->d_jnt_axis,
                    m->d_qpos0, m->d_site_bodyid, m->d_masks, m->d_scratch};
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
    HIP_TRY(hipMemcpyAsync(m->d_blob + m->h.off_site_pos, offsets, sizeof(float) * 3 * m->h.K,
                           hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return STAC_OK;
}

extern "C" int32_t stac_get_site_pos(const stac_model *m, float *out, void *stream) {
    if (!m || !out) return fail(STAC_ERR_INVALID, "null argument");
    HIP_TRY(hipMemcpyAsync(out, m->d_blob + m->h.off_site_pos, sizeof(float) * 3 * m->h.K,
                           hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return STAC_OK;
}

static int fk_impl(const stac_model *mc, const float *qpos, int32_t N, float *qn, float *xpos, float *xquat,
                   float *site_xpos, int normalize, void *stream) {
    stac_model *m = const_cast<stac_model *>(mc);
    if (!m || !qpos || N < 0) return fail(STAC_ERR_INVALID, "stac_fk: bad argument");
    if (N == 0) return STAC_OK;
    const size_t nb = m->h.nbody;
    if (!xpos || !xquat) {
        const int rc = ensure_scratch(m, (size_t)N * nb * 7);
        if (rc != STAC_OK) return rc;
        if (!xpos) xpos = m->d_scratch;
        if (!xquat) xquat = m->d_scratch + (size_t)N * nb * 3;
    }
    HIP_TRY(launch_fk(m->full(), qpos, N, qn, xpos, xquat, site_xpos, normalize, (hipStream_t)stream));
    return STAC_OK;
}

extern "C" int32_t stac_fk(const stac_model *m, const float *qpos, int32_t N, float *qn, float *xpos,
                           float *xquat, float *site_xpos, void *stream) {
    return fk_impl(m, qpos, N, qn, xpos, xquat, site_xpos, 1, stream);
}

static int pick_lanes(const stac_model *m, int requested, int nchains) {
    if (requested == 4 || requested == 8 || requested == 16 || requested == 32 || requested == 64) return requested;
    (void)m;
    // Heuristic: more lanes per chain when there are few chains (latency), fewer when there are many
    // (throughput).  256 CUs x 4 SIMDs; aim at >= 2 wavefronts per SIMD.
    if (nchains >= 16384) return 8;
    if (nchains >= 4096) return 16;
    if (nchains >= 1024) return 32;
    return 64;
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
    int G = pick_lanes(m, p->lanes_per_chain, nchains);
    size_t lds = q_lds_bytes(m->h, G, nkinds);
    while (lds > 160 * 1024 && G < 64) { G *= 2; lds = q_lds_bytes(m->h, G, nkinds); }
    if (lds > 160 * 1024) return fail(STAC_ERR_CAPACITY, "model does not fit the 160 KiB LDS of a CU");
    int cap = 0;
    a.mb_words = q_mb_words(nkinds, G);
    hipError_t e = launch_q_phase(a, G, lds, s, &cap);
    if (e != hipSuccess && cap == 0) {
        // no instantiation with this many lanes holds nq: widen the group
        for (int g2 = G * 2; g2 <= 64 && cap == 0; g2 *= 2) {
            lds = q_lds_bytes(m->h, g2, nkinds);
            if (lds > 160 * 1024) continue;
            a.mb_words = q_mb_words(nkinds, g2);
            e = launch_q_phase(a, g2, lds, s, &cap);
        }
        if (cap == 0) return fail(STAC_ERR_CAPACITY, "nq exceeds the compiled q_phase kernel capacity (256)");
    }
    if (e != hipSuccess) return fail(STAC_ERR_HIP, std::string("q_phase launch: ") + hipGetErrorString(e));
    return STAC_OK;
}

extern "C" int32_t stac_q_solve(const stac_model *mc, const stac_q_params *p, const float *kp, const float *q0,
                                const uint8_t *qs_to_opt, const uint8_t *kps_to_opt, int32_t N,
                                float *params_out, float *state_out, uint32_t *counters_out, void *stream) {
    stac_model *m = const_cast<stac_model *>(mc);
    if (!m || !p || !kp || !q0 || !qs_to_opt || !kps_to_opt || !params_out || !state_out || N < 0)
        return fail(STAC_ERR_INVALID, "stac_q_solve: bad argument");
    if (N == 0) return STAC_OK;
    hipStream_t s = (hipStream_t)stream;
    const int nqpad = m->h.nqpad, K = m->h.K, nq = m->h.nq;
    std::vector<uint8_t> hostm((size_t)nqpad + 3 * K, 0);
    std::memcpy(hostm.data(), qs_to_opt, nq);
    std::memcpy(hostm.data() + nqpad, kps_to_opt, 3 * K);
    uint8_t *d_kpw3 = m->d_masks + (size_t)kMaxKinds * nqpad + K;
    HIP_TRY(hipMemcpyAsync(m->d_masks, hostm.data(), nqpad, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_kpw3, hostm.data() + nqpad, 3 * K, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));  // hostm is a temporary
    QArgs a{};
    a.kp = kp; a.q_init = q0; a.masks = m->d_masks; a.kpw = nullptr; a.kpw3 = d_kpw3;
    a.C = N; a.F = 1; a.P = 0; a.root_kp_idx = 0; a.do_root_opt = 0; a.single = 1;
    a.qpos_out = params_out; a.err_out = state_out; a.counters_out = counters_out; a.q_carry_out = nullptr;
    return run_q(m, p, a, N, s);
}

extern "C" int32_t stac_q_phase(const stac_model *mc, const stac_q_params *p, const float *kp, const float *q_init,
                                const uint8_t *part_masks, const uint8_t *trunk_kps, int32_t C, int32_t F, int32_t P,
                                int32_t root_kp_idx, int32_t root_dims, int32_t do_root_opt, float *qpos_out,
                                float *err_out, uint32_t *counters_out, float *q_carry_out, float *xpos_out,
                                float *xquat_out, float *markers_out, void *stream) {
    stac_model *m = const_cast<stac_model *>(mc);
    if (!m || !p || !kp || !qpos_out || !err_out || C < 0 || F < 0 || P < 0)
        return fail(STAC_ERR_INVALID, "stac_q_phase: bad argument");
    if (P > 0 && !part_masks) return fail(STAC_ERR_INVALID, "stac_q_phase: part_masks is null");
    if (P + 3 > kMaxKinds) return fail(STAC_ERR_CAPACITY, "too many part groups");
    if (do_root_opt && (!trunk_kps || root_kp_idx < 0 || root_kp_idx >= m->h.K || root_dims < 1 || root_dims > m->h.nq))
        return fail(STAC_ERR_INVALID, "stac_q_phase: bad root optimisation arguments");
    if (C == 0 || F == 0) return STAC_OK;
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
    HIP_TRY(hipMemcpyAsync(m->d_masks, hostm.data(), (size_t)(P + 3) * nqpad, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_kpw, hostm.data() + (size_t)(P + 3) * nqpad, K, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));  // hostm is a temporary
    QArgs a{};
    a.kp = kp; a.q_init = q_init; a.masks = m->d_masks; a.kpw = d_kpw; a.kpw3 = nullptr;
    a.C = C; a.F = F; a.P = P; a.root_kp_idx = root_kp_idx; a.do_root_opt = do_root_opt ? 1 : 0; a.single = 0;
    a.qpos_out = qpos_out; a.err_out = err_out; a.counters_out = counters_out; a.q_carry_out = q_carry_out;
    const int rc = run_q(m, p, a, C, s);
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
    if (!m || !kp || !q || !workspace || !partial || T < 0) return fail(STAC_ERR_INVALID, "stac_m_phase_partial: bad argument");
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
    HIP_TRY(launch_m_finish(m->h.K, partial, initial_offsets, is_regularized, reg_coef, offsets_out, err_out,
                            (hipStream_t)stream));
    return STAC_OK;
}
