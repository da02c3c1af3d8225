// stac_device.hpp -- device-side helpers shared by the kernels of libstac_hip.so (math that mirrors
// oracle/stac_oracle.c operation for operation, cross-lane sums, LDS access helpers).
#pragma once
#include <hip/hip_runtime.h>

#include "stac_plan.hpp"

namespace stac {

// ------------------------------------------------------------------------------------------------
// math (mirrors oracle/stac_oracle.c: dot3, cross3, rotate, qmul, normalize4, sincos_, quat_to_mat)
// ------------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
struct Q4 { float w, x, y, z; };

#define FMA(a, b, c) __builtin_fmaf((a), (b), (c))
__device__ __forceinline__ float dot3(V3 a, V3 b) { return FMA(a.z, b.z, FMA(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) {
    return {FMA(a.y, b.z, -(a.z * b.y)), FMA(a.z, b.x, -(a.x * b.z)), FMA(a.x, b.y, -(a.y * b.x))};
}
__device__ __forceinline__ V3 add3(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 sub3(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }

// rotate(vec, quat): r = 2(u.v)u + (s^2 - u.u)v + 2 s (u x v)
__device__ __forceinline__ V3 rotate(V3 v, Q4 q) {
    const float s = q.w;
    const V3 u = {q.x, q.y, q.z};
    const float uv = dot3(u, v), uu = dot3(u, u);
    const V3 c = cross3(u, v);
    const float k = FMA(s, s, -uu), t = uv + uv, s2 = s + s;
    return {FMA(s2, c.x, FMA(k, v.x, t * u.x)), FMA(s2, c.y, FMA(k, v.y, t * u.y)), FMA(s2, c.z, FMA(k, v.z, t * u.z))};
}
__device__ __forceinline__ Q4 qmul(Q4 u, Q4 v) {
    Q4 r;
    r.w = FMA(-u.z, v.z, FMA(-u.y, v.y, FMA(-u.x, v.x, u.w * v.w)));
    r.x = FMA(-u.z, v.y, FMA(u.y, v.z, FMA(u.x, v.w, u.w * v.x)));
    r.y = FMA(u.z, v.x, FMA(u.y, v.w, FMA(-u.x, v.z, u.w * v.y)));
    r.z = FMA(u.z, v.w, FMA(-u.y, v.x, FMA(u.x, v.y, u.w * v.z)));
    return r;
}
// normalize(x) = x / (|x| + 1e-6 [|x| == 0]); returns |x| through *n
__device__ __forceinline__ Q4 normalize4(Q4 q, float *n_out) {
    const float n = __builtin_sqrtf(FMA(q.z, q.z, FMA(q.y, q.y, FMA(q.x, q.x, q.w * q.w))));
    const float d = n + (n == 0.0f ? 1e-6f : 0.0f);
    *n_out = n;
    return {q.w / d, q.x / d, q.y / d, q.z / d};
}
// Cody-Waite + cephes minimax sin/cos as an explicit mul/fma sequence: the oracle's sincos_ operation
// for operation.
__device__ __forceinline__ void sincos_(float x, float *sn, float *cs) {
    const float k = __builtin_rintf(x * 0.636619772367581343f);
    float r = FMA(-k, 1.5703125f, x);
    r = FMA(-k, 4.837512969970703125e-4f, r);
    r = FMA(-k, 7.54978995489188216e-8f, r);
    const float z = r * r;
    const float ps = FMA(FMA(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    const float s0 = FMA(r * z, ps, r);
    const float pc = FMA(FMA(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    const float c0 = FMA(z * z, pc, FMA(-0.5f, z, 1.0f));
    const int q = ((int)k) & 3;
    const float ss = (q & 1) ? c0 : s0, cc = (q & 1) ? s0 : c0;
    *sn = (q & 2) ? -ss : ss;
    *cs = ((q + 1) & 2) ? -cc : cc;
}
__device__ __forceinline__ float clipf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ V3 ld3(const float *p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ Q4 ld4(const float *p) { return {p[0], p[1], p[2], p[3]}; }
__device__ __forceinline__ void st3(float *p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ void st4(float *p, Q4 q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }

// ---- cross-lane sums (oracle: tree_sum) ---------------------------------------------------------------
// The value held by the lane whose index differs in bit H.  Levels are applied in increasing H, so when
// H >= 4 every lane of the 4- (8-, 16-) lane block already holds the same partial sum and a mirrored
// read from the neighbouring block returns exactly the xor-partner's value: all of it stays on DPP.
template <int H>
__device__ __forceinline__ float xor_partner(float v) {
    const int i = __builtin_bit_cast(int, v);
    int r;
    if constexpr (H == 1) r = __builtin_amdgcn_update_dpp(i, i, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (H == 2) r = __builtin_amdgcn_update_dpp(i, i, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (H == 4) r = __builtin_amdgcn_update_dpp(i, i, 0x141, 0xF, 0xF, true);  // row_half_mirror
    else if constexpr (H == 8) r = __builtin_amdgcn_update_dpp(i, i, 0x140, 0xF, 0xF, true);  // row_mirror
    else if constexpr (H == 16) r = __builtin_amdgcn_ds_swizzle(i, 0x401F);                   // xor 16 inside 32 lanes
    else r = __shfl_xor(i, 32, 64);
    return __builtin_bit_cast(float, r);
}
// Pairwise-tree sum over the elements e = r*G + lane_in_group of a striped vector (zeros beyond nq):
// lane butterflies for the levels below G, then the registers.  Every lane of the group gets the sum.
// (Round 6 tried gfx950's v_permlane16_swap / v_permlane32_swap for the two levels that cross a 16-lane row, instead of ds_swizzle /
//  ds_bpermute: 3 % SLOWER on 250-frame clips, 13 % on the LM solver -- and not the xor-partner exchange this tree needs.  Dropped.)
template <int G, int NQR>
__device__ __forceinline__ float group_tree_sum(const float (&v)[NQR]) {
    constexpr int PR = NQR <= 1 ? 1 : NQR <= 2 ? 2 : NQR <= 4 ? 4 : NQR <= 8 ? 8 : NQR <= 16 ? 16 : 32;
    float t[PR];
#pragma unroll
    for (int r = 0; r < PR; ++r) {
        float x = r < NQR ? v[r] : 0.0f;
        if constexpr (G >= 2) x = x + xor_partner<1>(x);
        if constexpr (G >= 4) x = x + xor_partner<2>(x);
        if constexpr (G >= 8) x = x + xor_partner<4>(x);
        if constexpr (G >= 16) x = x + xor_partner<8>(x);
        if constexpr (G >= 32) x = x + xor_partner<16>(x);
        if constexpr (G >= 64) x = x + xor_partner<32>(x);
        t[r] = x;
    }
#pragma unroll
    for (int h = 1; h < PR; h *= 2)
#pragma unroll
        for (int i = 0; i < PR; i += 2 * h) t[i] = t[i] + t[i + h];
    return t[0];
}

// ---- launch arguments as the trip loop sees them ---------------------------------------------------------------------------
// QArgs is 424 bytes of kernel argument.  Read once in the prologue, every field -- and every address or lane mask derived
// from one -- stays live across the whole trip loop: the kernels ran with 100 to 310 scalars spilled into vector-register
// lanes.  Inside the loop the arguments are therefore seen through two views with the members' own names:
//   * the fields that every trip uses are scalars PINNED before the loop (made opaque, so that the compiler neither
//     re-derives nor re-loads them: they simply occupy ~40 of the 102 scalar registers);
//   * everything else is a reference into the kernarg segment through a pointer that is laundered inside the loop: a cold
//     field is fetched by a scalar load where it is used (the constant cache holds the segment) and is dead afterwards.
template <class T>
using KRef = const __attribute__((address_space(4))) T &;
template <class T>
__device__ __forceinline__ T pin_scalar(T v) {
    asm volatile("" : "+s"(v));
    return v;
}
template <class T>
__device__ __forceinline__ T pin_vector(T v) {
    asm volatile("" : "+v"(v));
    return v;
}
// How a field is kept across the trip loop: S = pinned in a scalar register; V = pinned in a vector register where the kernel
// has vector registers to spare (VPIN: the latency kernels; an offset that is only ever added to a lane's address costs
// nothing there), else re-read; N = re-read from the kernarg segment where it is used.
#ifndef STAC_PIN_ADDR
#define STAC_PIN_ADDR V
#endif
#ifndef STAC_PIN_CTL
#define STAC_PIN_CTL C
#endif
enum PinClass { PIN_N = 0, PIN_S = 1, PIN_V = 2, PIN_C = 3 };
#define STAC_PINCLASS_(c) PIN_##c
#define STAC_PINCLASS(c) STAC_PINCLASS_(c)
// (field, class): loop bounds, flags and sizes that feed scalar compares and branches; then LDS offsets
#ifndef STAC_PIN_CTL2
#define STAC_PIN_CTL2 STAC_PIN_CTL
#endif
#define STAC_HOT_HEADER_FIELDS(X)                                                                                              \
    X(nq, STAC_PIN_CTL2) X(K, STAC_PIN_CTL2) X(nqpad, N) X(naj, STAC_PIN_CTL) X(nrange, STAC_PIN_CTL) X(max_width, STAC_PIN_CTL) \
    X(fk_hdr_words, STAC_PIN_CTL2) X(n_mlev_hdr, STAC_PIN_CTL2) X(n_mlev, STAC_PIN_CTL2) X(fk_rec_words, STAC_PIN_CTL2)         \
    X(fk_uniform, STAC_PIN_CTL2)                                                                                               \
    X(off_joint, STAC_PIN_ADDR) X(off_site, STAC_PIN_ADDR) X(off_lb, STAC_PIN_ADDR) X(off_ub, STAC_PIN_ADDR)                    \
    X(off_range, STAC_PIN_ADDR) X(off_fkstep, STAC_PIN_ADDR) X(off_fkroot, STAC_PIN_ADDR) X(c_bx, STAC_PIN_ADDR)                \
    X(c_ja, STAC_PIN_ADDR) X(c_jn, STAC_PIN_ADDR) X(c_sw, STAC_PIN_ADDR) X(c_sink, STAC_PIN_ADDR) X(c_rw, STAC_PIN_ADDR)        \
    X(c_qe, STAC_PIN_ADDR) X(c_qsv, STAC_PIN_ADDR) X(chain_stride, STAC_PIN_ADDR)                                              \
    X(c3_ql, STAC_PIN_ADDR) X(c3_qb, STAC_PIN_ADDR) X(c3_pb, STAC_PIN_ADDR) X(c3_rw0, STAC_PIN_ADDR) X(off3_prog, STAC_PIN_ADDR) \
    X(off3_root, STAC_PIN_ADDR) X(off3_site, STAC_PIN_ADDR) X(fk3_n, STAC_PIN_CTL2) \
    X(fk3_cap1, STAC_PIN_CTL2) X(fk3_cap2, STAC_PIN_CTL2) X(fk3_cap3, STAC_PIN_CTL2) X(rsplit, STAC_PIN_CTL)
#define STAC_HOT_ARGS_FIELDS(X)                                                                                                \
    X(single, STAC_PIN_CTL) X(P, STAC_PIN_CTL2) X(flags, STAC_PIN_CTL) X(free0p, STAC_PIN_CTL) X(root_fast, STAC_PIN_CTL)       \
    X(n_mlev_root, STAC_PIN_CTL2) X(n_run_root, STAC_PIN_CTL2) X(n_root_joints, STAC_PIN_CTL2) X(maxls, STAC_PIN_CTL)           \
    X(maxiter, STAC_PIN_CTL2) X(queue_slots, N) X(resume, N) X(root_trunk_lo, N) X(root_trunk_hi, N) X(tol, STAC_PIN_CTL2)
// (VPIN: 1 = class V fields are pinned in vector registers; bit 1 set = class C fields are pinned in scalar registers)
template <int VPIN, PinClass C, class T>
__device__ __forceinline__ T pin_as(T v) {
    if constexpr (C == PIN_S || (C == PIN_C && (VPIN & 2))) return pin_scalar(v);
    else if constexpr (C == PIN_V && (VPIN & 1)) return pin_vector(v);
    else return v;
}
template <int VPIN, PinClass C>
constexpr bool is_pinned() { return C == PIN_S || (C == PIN_V && (VPIN & 1)) || (C == PIN_C && (VPIN & 2)); }
struct HotHeader {  // PlanHeader fields of every trip (values)
    int32_t nq, K, nqpad, naj, nrange, max_width, off_joint, off_site, off_lb, off_ub, off_range, off_fkstep, off_fkroot, fk_hdr_words,
        n_mlev_hdr, n_mlev, fk_rec_words, fk_uniform, c_bx, c_ja, c_jn, c_sw, c_sink, c_rw, c_qe, c_qsv, chain_stride;
    int32_t c3_ql, c3_qb, c3_pb, c3_rw0, off3_prog, off3_root, off3_site, fk3_n, fk3_cap1, fk3_cap2, fk3_cap3;  // split kinematics
    int32_t rsplit;
};
template <int VPIN, class KH>
__device__ __forceinline__ HotHeader pin_header(const KH &h) {  // before the loop: the pinned ones
    HotHeader o = {};
#define STAC_PIN(f, c) if constexpr (is_pinned<VPIN, STAC_PINCLASS(c)>()) o.f = pin_as<VPIN, STAC_PINCLASS(c)>(h.f);
    STAC_HOT_HEADER_FIELDS(STAC_PIN)
#undef STAC_PIN
    return o;
}
struct TripHeader : HotHeader {  // + the other PlanHeader fields (references into the kernarg segment)
    KRef<int32_t> nbody, njnt, nab, nlev, nquat, has_ball, off_lev_adr, off_body, off_qpos0, off_quat_adr, off_active, total_words,
        plan_skip, core_words, c_gg, c_kp, c_r2, stride_regs, stride_lds, stride_forced, nst, nqj, kpow2;
    KRef<int32_t> fk3, stride3, nbq, c3_bq, off3_bq;
};
template <int VPIN, class KH>
__device__ __forceinline__ TripHeader trip_header(const HotHeader &hot, const KH &k) {  // inside the loop: pinned values + fresh reads
    HotHeader t = hot;
#define STAC_PIN(f, c) if constexpr (!is_pinned<VPIN, STAC_PINCLASS(c)>()) t.f = k.f;
    STAC_HOT_HEADER_FIELDS(STAC_PIN)
#undef STAC_PIN
    return TripHeader{t, k.nbody, k.njnt, k.nab, k.nlev, k.nquat, k.has_ball, k.off_lev_adr, k.off_body, k.off_qpos0, k.off_quat_adr,
                      k.off_active, k.total_words, k.plan_skip, k.core_words, k.c_gg, k.c_kp, k.c_r2, k.stride_regs, k.stride_lds,
                      k.stride_forced, k.nst, k.nqj, k.kpow2,
                      k.fk3, k.stride3, k.nbq, k.c3_bq, k.off3_bq};
}
static_assert(sizeof(PlanHeader) == (27 + 15 + 19 + 2 + 1 + 3) * 4, "TripHeader must list every PlanHeader field");
struct HotArgs {  // QArgs fields of every trip (values)
    int32_t single, P, flags, free0p, root_fast, n_mlev_root, n_run_root, n_root_joints, maxls, maxiter, queue_slots, resume;
    uint32_t root_trunk_lo, root_trunk_hi;
    float tol;
};
template <int VPIN, class KA>
__device__ __forceinline__ HotArgs pin_args(const KA &a) {
    HotArgs o = {};
#define STAC_PIN(f, c) if constexpr (is_pinned<VPIN, STAC_PINCLASS(c)>()) o.f = pin_as<VPIN, STAC_PINCLASS(c)>(a.f);
    STAC_HOT_ARGS_FIELDS(STAC_PIN)
#undef STAC_PIN
    return o;
}
struct TripArgs : HotArgs {
    KRef<int32_t> fk3r_n;
    KRef<const float *> kp, q_init;
    KRef<const uint8_t *> kpw, kpw3;
    KRef<const int32_t *> perm;
    KRef<int32_t> C, F, root_kp_idx, do_root_opt;
    KRef<int32_t *> ctl;
    KRef<float *> hand, qpos_out, err_out;
    KRef<uint32_t *> counters_out;
    KRef<float *> q_carry_out;
};
template <int VPIN, class KA>
__device__ __forceinline__ TripArgs trip_args(const HotArgs &hot, const KA &k) {
    HotArgs t = hot;
#define STAC_PIN(f, c) if constexpr (!is_pinned<VPIN, STAC_PINCLASS(c)>()) t.f = k.f;
    STAC_HOT_ARGS_FIELDS(STAC_PIN)
#undef STAC_PIN
    return TripArgs{t, k.fk3r_n, k.kp, k.q_init, k.kpw, k.kpw3, k.perm, k.C, k.F, k.root_kp_idx, k.do_root_opt, k.ctl, k.hand, k.qpos_out, k.err_out,
                    k.counters_out, k.q_carry_out};
}

enum : int { ST_VG_Y = 0, ST_LS = 1, ST_VG_X = 2, ST_DONE = 3, ST_SPEC = 4, ST_WAIT = 5, ST_NEXT = 6 };

// In-kernel phase stamps: diagnostic build only (-DSTAC_PROFILE -> libstac_hip_prof.so); the stamps
// go to a buffer of their own and feed no output.  Read the SHARES, not the run time.
#ifdef STAC_PROFILE
// -DSTAC_PROFILE_SEL=1: only root fast (lite) trips are accumulated; =0: only the others; undefined: all trips
#define PROF_DECL unsigned long long pt0 = __builtin_readcyclecounter(), pacc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ptmp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long ptrip0 = pt0
#define PROF_TICK(i)                                             \
    do {                                                         \
        const unsigned long long pt1 = __builtin_readcyclecounter(); \
        ptmp[i] += pt1 - pt0;                                    \
        pt0 = pt1;                                               \
    } while (0)
#define PROF_TRIP
#ifdef STAC_PROFILE_SEL
#define PROF_KEEP(isroot) ((isroot) == (STAC_PROFILE_SEL != 0))
#else
#define PROF_KEEP(isroot) true
#endif
#define PROF_ROOT(isroot) do { const unsigned long long pq = __builtin_readcyclecounter(); if (isroot) { pacc[12] += pq - ptrip0; pacc[13] += 1; } ptrip0 = pq; \
        const bool keep_ = PROF_KEEP(isroot); for (int i_ = 0; i_ < 11; ++i_) { if (keep_) pacc[i_] += ptmp[i_]; ptmp[i_] = 0; } if (keep_) pacc[11] += 1; } while (0)
// (the LM kernel keeps the plain accumulation: it has no trip classes)
#define PROF_LM_END do { for (int i_ = 0; i_ < 12; ++i_) { pacc[i_] += ptmp[i_]; ptmp[i_] = 0; } } while (0)
#define PROF_FLUSH(a)                                                                     \
    do {                                                                                  \
        if (a.prof && lane == 0)                                                          \
            for (int i = 0; i < 14; ++i) atomicAdd(a.prof + i, pacc[i]);                  \
    } while (0)
#elif defined(STAC_MARK)  // developer builds: phase boundaries as comments in the assembly listing (no instruction)
#define PROF_DECL
#define PROF_TRIP
#define PROF_ROOT(isroot)
#define PROF_LM_END
#define PROF_TICK(i) asm volatile("; TICK " #i)
#define PROF_FLUSH(a)
#else
#define PROF_DECL
#define PROF_TRIP
#define PROF_ROOT(isroot)
#define PROF_LM_END
#define PROF_TICK(i)
#define PROF_FLUSH(a)
#endif
enum : int { JFREE = 0, JBALL = 1, JSLIDE = 2, JHINGE = 3 };

// Chains never span wavefronts, so ordering this wave's own LDS traffic is enough: LDS operations of
// one wave execute in issue order; the fence stops the compiler from moving or caching LDS accesses
// across the point (it lowers to s_waitcnt lgkmcnt(0)), the wave barrier pins the schedule.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}


__device__ __forceinline__ float4 lds4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ int4 lds4i(const float *p) { return *reinterpret_cast<const int4 *>(p); }
// transform entries (kXf words: stac_plan.hpp): position in words 0-2, quaternion (x, y, z, w) in words kXq .. kXq + 3
__device__ __forceinline__ V3 ld_tpos(const float *p) {
    if constexpr (kXf == 8) { const float4 v = lds4(p); return {v.x, v.y, v.z}; }
    else return ld3(p);
}
__device__ __forceinline__ Q4 ld_tquat(const float *p) {
    if constexpr (kXf == 8) { const float4 v = lds4(p + 4); return {v.w, v.x, v.y, v.z}; }
    else return {p[6], p[3], p[4], p[5]};
}
__device__ __forceinline__ void st_tpos(float *p, V3 v) {
    if constexpr (kXf == 8) *reinterpret_cast<float4 *>(p) = float4{v.x, v.y, v.z, 0.0f};
    else st3(p, v);
}
__device__ __forceinline__ void st_tquat(float *p, Q4 q) {
    if constexpr (kXf == 8) *reinterpret_cast<float4 *>(p + 4) = float4{q.x, q.y, q.z, q.w};
    else { p[3] = q.x; p[4] = q.y; p[5] = q.z; p[6] = q.w; }
}
// second vector of a wrench entry {f, t}
__device__ __forceinline__ V3 ld_tvec2(const float *p) { return ld_tpos(p + kXq); }
__device__ __forceinline__ void st_tvec2(float *p, V3 v) { st_tpos(p + kXq, v); }

// ------------------------------------------------------------------------------------------------
// forward kinematics of one chain out of LDS (shared by the PG and the LM kernel)
// ------------------------------------------------------------------------------------------------
// Joint-local transforms: everything of the kinematics that depends on q alone, one lane per joint (hinge: the
// half-angle sincos and the local quaternion; free / ball: the normalised quaternion, written back like MJX does;
// slide: the displacement).  This keeps the sincos off the serial chain of the FK that follows.  The result sits in
// the ja slot that the joint's pre-joint quaternion takes once FK has consumed it, so it needs no LDS of its own.
// naj_lim: the leading joints to do (all, or -- root fast trips -- only the joints of the root passes' coordinates).
// HINGES: every joint of the loop is a hinge (lean kernels: a uniform program below a free root that is handled apart) -- no dispatch.
// hinges_rt: the same, known at launch time (QArgs::flags bit 4: generic kernels on models whose joints are all hinges below the root).
template <bool HINGES = false, class HT>
__device__ __forceinline__ void joint_local_prepass(const HT &H, const float *P, float *CBc, const int lf, const int gf,
                                                    const int naj_lim, const int j_first = 0, const bool hinges_rt = false) {
    const float *jrec = P + H.off_joint;
    float *qe = CBc + H.c_qe, *jn = CBc + H.c_jn, *ja = CBc + H.c_ja, *qsv = CBc + H.c_qsv;
    if (!HINGES && hinges_rt) {  // (wave-uniform) one straight loop, no dispatch
        for (int j = lf + j_first; j < naj_lim; j += gf) {
            const float *jr = jrec + 12 * j;
            const int ad = reinterpret_cast<const int *>(jr)[1];
            const float4 jp4 = lds4(jr + 4);  // pos, q0
            const float4 ja4 = lds4(jr + 8);  // axis, slot
            const float angle = qe[ad] - jp4.w;
            float sn, cs;
            sincos_(angle * 0.5f, &sn, &cs);
            st_tquat(ja + kXf * j, Q4{cs, ja4.x * sn, ja4.y * sn, ja4.z * sn});
        }
        return;
    }
    for (int j = lf + j_first; j < naj_lim; j += gf) {
        const float *jr = jrec + 12 * j;
        const int4 ji = lds4i(jr);  // type, qadr, slo, shi
        const int ty = ji.x, ad = ji.y;
        if (HINGES || ty == JHINGE) {
            const float4 jp4 = lds4(jr + 4);  // pos, q0
            const float4 ja4 = lds4(jr + 8);  // axis, slot
            const float angle = qe[ad] - jp4.w;
            float sn, cs;
            sincos_(angle * 0.5f, &sn, &cs);
            st_tquat(ja + kXf * j, Q4{cs, ja4.x * sn, ja4.y * sn, ja4.z * sn});
        } else if (ty == JSLIDE) {
            const float4 jp4 = lds4(jr + 4);
            ja[kXf * j + kXw] = qe[ad] - jp4.w;  // (the w word of the entry's quaternion)
        } else {
            const int qa = ty == JFREE ? ad + 3 : ad;
            float n;
            const Q4 qn = normalize4(ld4(qe + qa), &n);
            st4(qe + qa, qn);  // written back, like MJX
            st_tquat(ja + kXf * j, qn);
            if (ty == JFREE) st_tpos(ja + kXf * j, ld3(qe + ad));  // {position, quaternion}: the FK program's synthetic parent of the body
            // the gradient pass needs |q| and the unit quaternion after qe's region has been reused: they are kept by
            // the joint's ordinal among the quaternion joints (JointRec::q0 of a free / ball joint)
            const int qord = __builtin_bit_cast(int, lds4(jr + 4).w);
            jn[qord] = n;
            st4(qsv + 4 * qord, qn);
        }
    }
}

// Forward kinematics of one chain, level by level (mjx smooth.kinematics), by gf lanes (lf = this
// lane's index among them): reads the evaluation point qe (quaternions already normalised) and the joint-local
// quaternions ql, writes the body transforms bx and, if store_ja, every joint's anchor and pre-joint quaternion.
// A lane keeps the transform of the body it has just finished: the host lays the levels out so that a body sits
// at its parent's position in the level wherever it can (flag bit 1), and then the parent transform never makes
// the LDS round trip.  All lanes of the wavefront must call it together (wave-level synchronisation per level).
template <class HT>
__device__ __forceinline__ void fk_levels(const HT &H, const float *P, float *CBc, const int lf, const int gf,
                                          const bool active, const bool store_ja) {
    const int *lev_adr = reinterpret_cast<const int *>(P + H.off_lev_adr);
    const float *brec = P + H.off_body, *jrec = P + H.off_joint;
    float *bx = CBc + H.c_bx, *ja = CBc + H.c_ja;
    const float *qe = CBc + H.c_qe;
    const bool carry_ok = H.max_width <= gf;  // every level is a single pass
    V3 cpos = {0.f, 0.f, 0.f};
    Q4 cquat = {1.f, 0.f, 0.f, 0.f};
    for (int lev = 0; lev < H.nlev; ++lev) {
        const int s_end = active ? lev_adr[lev + 1] : 0;
        for (int s = lev_adr[lev] + lf; s < s_end; s += gf) {
            const float *br = brec + 12 * s;
            const int4 bi = lds4i(br);       // parent, jadr, jnum, flags
            const float4 bp = lds4(br + 4);  // pos, zero-jnt_pos bits
            V3 ppos = cpos;
            Q4 pquat = cquat;
            if (!(carry_ok && (bi.w & 2))) {
                const float *pp = bx + bi.x * kXf;
                ppos = ld_tpos(pp);
                pquat = ld_tquat(pp);
            }
            V3 pos = add3(ppos, rotate(V3{bp.x, bp.y, bp.z}, pquat));
            Q4 quat = pquat;  // product with an identity body_quat is exact: skipped
            if (!(bi.w & 1)) {
                const float4 bq = lds4(br + 8);
                quat = qmul(pquat, Q4{bq.x, bq.y, bq.z, bq.w});
            }
            const int jz = __builtin_bit_cast(int, bp.w);
            for (int jj = 0; jj < bi.z; ++jj) {
                const int j = bi.y + jj;
                const float *jr = jrec + 12 * j;
                const int4 ji = lds4i(jr);        // type, qadr, slo, shi
                const float4 jp4 = lds4(jr + 4);  // pos, q0
                const int ty = ji.x, ad = ji.y;
                const V3 jp = {jp4.x, jp4.y, jp4.z};
                // jnt_pos == 0: rotate(0, q) is a zero vector, so anchor = pos and pos stays (exact)
                const bool jzero = jj < 31 && ((jz >> jj) & 1);
                const Q4 prequat = quat;  // xaxis = rotate(jnt_axis, prequat) is evaluated by the joint pass
                V3 anchor = pos;
                if (ty == JHINGE || ty == JBALL) {
                    const Q4 qloc = ld_tquat(ja + kXf * j);
                    if (!jzero) anchor = add3(rotate(jp, quat), pos);
                    quat = qmul(quat, qloc);
                    if (!jzero) pos = sub3(anchor, rotate(jp, quat));
                } else if (ty == JFREE) {
                    anchor = ld3(qe + ad);
                    pos = anchor;
                    quat = ld4(qe + ad + 3);  // normalised by the pre-pass
                } else {  // slide
                    if (!jzero) anchor = add3(rotate(jp, quat), pos);
                    const float4 ja4 = lds4(jr + 8);
                    const V3 axis = rotate(V3{ja4.x, ja4.y, ja4.z}, quat);
                    const float d = ja[kXf * j + kXw];
                    pos = {FMA(axis.x, d, pos.x), FMA(axis.y, d, pos.y), FMA(axis.z, d, pos.z)};
                }
                if (store_ja) {
                    st_tpos(ja + kXf * j, anchor);
                    st_tquat(ja + kXf * j, prequat);
                }
            }
            const int xf = (int)((unsigned)bi.w >> 16);
            if (xf != 0xFFFF) {  // somebody reads it back (a site, or a child on another lane)
                st_tpos(bx + xf * kXf, pos);
                st_tquat(bx + xf * kXf, quat);
            }
            cpos = pos;
            cquat = quat;
        }
        wave_sync();
    }
}

// The same kinematics driven by the FK "program" (FkStep records, stac_plan.hpp): one fixed-size record per
// (micro-level, lane position), fetched one step ahead together with the step's joint-local quaternion.  Every step
// is the same straight-line hinge arithmetic on neutral data where a part does not apply (zeros / identity: exact
// no-ops), so the common step has no divergent branch, no select and no address arithmetic beyond base + offset; a
// per-micro-level flag word (wave-uniform) says when some position needs a parent that another lane produced or has
// a free / slide joint.  Requires max_width <= gf.  Bit-identical to fk_levels.
struct FkRegs {
    float4 r0;  // bpos | kind
    int4 r1;    // par_off, ja_off, xf_off, ql_next
    float4 r2;  // jpos | aux
    float4 r3;  // body_quat (records of 16 words)
    Q4 ql;      // joint-local quaternion of the step's joint (identity: none)
};
template <int RW>
__device__ __forceinline__ void fk_fetch(FkRegs &R, const float *rec, const float *CBc, const int ql_off) {
    R.r0 = lds4(rec);
    R.r1 = lds4i(rec + 4);
    R.r2 = lds4(rec + 8);
    if constexpr (RW == 16) R.r3 = lds4(rec + 12);
    R.ql = ld_tquat(CBc + ql_off - kXq);  // ql_off = the quaternion words of a transform entry
}
// One step: fetch the next step's record into N (its ql offset is in this record), then run this one.
// `on`: the lane has a position in the program (the others run position 0's data, with nothing loaded or stored).
template <int RW>
__device__ __forceinline__ void fk_step(const FkRegs &R, FkRegs &N, const float *next_rec, const int mlf, const bool on,
                                        V3 &pos, Q4 &quat, float *CBc, const float *qe, const float *jrec, const bool store_ja) {
    fk_fetch<RW>(N, next_rec, CBc, R.r1.w);
    if (mlf & FK_ML_PARENT_LDS) {  // wave-uniform: some position starts a body whose parent another lane (or nobody) produced
        const int po = R.r1.x;
        if (on && po >= 0) {
            pos = ld_tpos(CBc + po);
            quat = ld_tquat(CBc + po);
        }
    }
    // every part below is skipped when NO position of this micro-level needs it (wave-uniform flags): the skipped
    // arithmetic would be an exact no-op on neutral data
    if (mlf & FK_ML_BODY) pos = add3(pos, rotate(V3{R.r0.x, R.r0.y, R.r0.z}, quat));
    if constexpr (RW == 16) {
        if (mlf & FK_ML_BQUAT) quat = qmul(quat, Q4{R.r3.x, R.r3.y, R.r3.z, R.r3.w});
    }
    const V3 jp = {R.r2.x, R.r2.y, R.r2.z};
    const V3 pos0 = pos;
    const Q4 prequat = quat;  // xaxis = rotate(jnt_axis, prequat) is evaluated by the joint pass
    V3 anchor = pos;
    if (mlf & FK_ML_JOINT) {
        if (mlf & FK_ML_JPOS) anchor = add3(rotate(jp, quat), pos);
        quat = qmul(quat, R.ql);
        if (mlf & FK_ML_JPOS) pos = sub3(anchor, rotate(jp, quat));
    }
    if (mlf & FK_ML_SPECIAL) {  // wave-uniform: some position has a free or a slide joint in this micro-level
        const int kind = __builtin_bit_cast(int, R.r0.w), aux = __builtin_bit_cast(int, R.r2.w);
        if (on && kind == FK_KIND_FREE) {
            anchor = ld3(qe + aux);
            pos = anchor;
            quat = R.ql;  // normalised by the pre-pass
        } else if (on && kind == FK_KIND_SLIDE) {
            const float4 ja4 = lds4(jrec + 12 * aux + 8);
            const V3 axis = rotate(V3{ja4.x, ja4.y, ja4.z}, prequat);
            const float d = R.ql.w;
            quat = prequat;
            pos = {FMA(axis.x, d, pos0.x), FMA(axis.y, d, pos0.y), FMA(axis.z, d, pos0.z)};
        }
    }
    if (store_ja && on) {  // (a step without a joint aims at the sink entry)
        st_tpos(CBc + R.r1.y, anchor);
        st_tquat(CBc + R.r1.y, prequat);
    }
    if (on) {
        st_tpos(CBc + R.r1.z, pos);
        st_tquat(CBc + R.r1.z, quat);
    }
    wave_sync();
}
// The step of a uniform program (PlanHeader::fk_uniform), one lane per position: the same straight-line arithmetic for
// every position and step, on neutral data where a part does not apply; the only branch is wave-uniform (`par`: some
// position of this step loads its parent).  Bit-identical to fk_step.
template <int RW>
__device__ __forceinline__ void fk_step_uniform(const FkRegs &R, FkRegs &N, const float *next_rec, const bool par, V3 &pos, Q4 &quat,
                                                float *CBc) {
    if (par) {
        const int src = R.r1.x >= 0 ? R.r1.x : R.r1.z;  // (any valid entry: the loaded values are dropped)
        const V3 lp = ld_tpos(CBc + src);
        const Q4 lq = ld_tquat(CBc + src);
        fk_fetch<RW>(N, next_rec, CBc, R.r1.w);
        if (R.r1.x >= 0) { pos = lp; quat = lq; }
    } else {
        fk_fetch<RW>(N, next_rec, CBc, R.r1.w);
    }
    pos = add3(pos, rotate(V3{R.r0.x, R.r0.y, R.r0.z}, quat));
    const V3 jp = {R.r2.x, R.r2.y, R.r2.z};
    const Q4 prequat = quat;
    const V3 anchor = add3(rotate(jp, quat), pos);
    quat = qmul(quat, R.ql);
    pos = sub3(anchor, rotate(jp, quat));
    st_tpos(CBc + R.r1.y, anchor);
    st_tquat(CBc + R.r1.y, prequat);
    st_tpos(CBc + R.r1.z, pos);
    st_tquat(CBc + R.r1.z, quat);
    wave_sync();
}
template <int RW, class HT>
__device__ __forceinline__ void fk_program(const HT &H, const float *P, float *CBc, const int lf, const int gf,
                                           const bool active, const bool store_ja, const int prog_off, const int n_ml) {
    const int W = H.max_width;
    const bool on = active && lf < W;
    // header: one word per PAIR of micro-levels (16 bits each: flags | form << 8), then the ql offset of every
    // position's first step
    const int *hdr = reinterpret_cast<const int *>(P + prog_off);
    const float *sp = P + prog_off + H.fk_hdr_words + RW * (on ? lf : 0);
    const float *jrec = P + H.off_joint;
    const float *qe = CBc + H.c_qe;
    V3 pos = {0.f, 0.f, 0.f};  // the lane's running transform: a body that follows its parent on the same lane
    Q4 quat = {1.f, 0.f, 0.f, 0.f};  // starts from it without touching LDS
    FkRegs A, B;
    A.r3 = B.r3 = float4{1.f, 0.f, 0.f, 0.f};
    fk_fetch<RW>(A, sp, CBc, hdr[(H.n_mlev_hdr >> 1) + (on ? lf : 0)]);
    const int stride = RW * W;
    int fl_v = hdr[0];
    if (H.fk_uniform) {
        if (!on) return;  // (wave_sync needs no company: LDS operations of a wave execute in order)
        for (int ml = 0; ml < n_ml; ml += 2) {
            const int fl = __builtin_amdgcn_readfirstlane(fl_v);
            fl_v = hdr[(ml >> 1) + 1];
            sp += stride;
            fk_step_uniform<RW>(A, B, sp, (fl & FK_ML_PARENT_LDS) != 0, pos, quat, CBc);
            if (ml + 2 < n_ml) sp += stride;
            fk_step_uniform<RW>(B, A, sp, ((fl >> 16) & FK_ML_PARENT_LDS) != 0, pos, quat, CBc);
        }
        return;
    }
    // two steps per trip so that the fetched record never has to be copied (n_mlev is even: padded by the host)
    for (int ml = 0; ml < n_ml; ml += 2) {
        const int fl = __builtin_amdgcn_readfirstlane(fl_v);
        fl_v = hdr[(ml >> 1) + 1];  // next pair (one word of slack behind the last pair: the first-step table follows)
        sp += stride;
        fk_step<RW>(A, B, sp, fl & 255, on, pos, quat, CBc, qe, jrec, store_ja);
        if (ml + 2 < n_ml) sp += stride;
        fk_step<RW>(B, A, sp, (fl >> 16) & 255, on, pos, quat, CBc, qe, jrec, store_ja);
    }
}

// ---- the same program, FOUR lanes per position ------------------------------------------------------------------
// Lane c = 0..3 of a quad owns component c of the running quaternion (w, x, y, z) and, for c >= 1, component c - 1
// of the running position.  Every result component is the scalar code's own mul / fma sequence (same operands,
// same order: bit-identical), with the other components fetched through quad_perm DPP operands:
//   qmul   r_c = u.w v_c +- u.x v_p1(c) +- u.y v_p2(c) +- u.z v_p3(c)           4 instructions instead of 16
//   rotate r_c = s2 (u x v)_c + (k v_c + t u_c)                                 9 instead of 24
// plus, once per new quaternion, its broadcasts and the v-independent terms (u.u, k, s2).  A step costs about half
// the instructions of the one-lane-per-position form, and in the trees at hand (at most four bodies side by side)
// the lanes were idle anyway.  Needs 4 * max_width <= lanes of the group.
template <int CTRL>
__device__ __forceinline__ float quad_dpp(float v) {
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, CTRL, 0xF, 0xF, true));
}
struct QuadQuat {
    float qc;              // own component
    float S, UX, UY, UZ;   // broadcasts of w, x, y, z
    float U1, U2;          // u_(c+1), u_(c+2) in the cyclic order x -> y -> z -> x (lanes 1-3)
    float K, S2;           // s^2 - u.u, 2 s
    float SX, SY, SZ;      // +-u.x, +-u.y, +-u.z with the signs of this lane's row of the quaternion product
};
__device__ __forceinline__ void quad_derive(QuadQuat &Q, const float qc, const int m1, const int m2, const int m3) {
    Q.qc = qc;
    Q.S = quad_dpp<0x00>(qc);
    Q.UX = quad_dpp<0x55>(qc);
    Q.UY = quad_dpp<0xAA>(qc);
    Q.UZ = quad_dpp<0xFF>(qc);
    Q.U1 = quad_dpp<0x78>(qc);  // [0,2,3,1]
    Q.U2 = quad_dpp<0x9C>(qc);  // [0,3,1,2]
    const float uu = FMA(Q.UZ, Q.UZ, FMA(Q.UY, Q.UY, Q.UX * Q.UX));
    Q.K = FMA(Q.S, Q.S, -uu);
    Q.S2 = Q.S + Q.S;
    Q.SX = __builtin_bit_cast(float, __builtin_bit_cast(int, Q.UX) ^ m1);
    Q.SY = __builtin_bit_cast(float, __builtin_bit_cast(int, Q.UY) ^ m2);
    Q.SZ = __builtin_bit_cast(float, __builtin_bit_cast(int, Q.UZ) ^ m3);
}
// The compiler folds a quad_perm operand into v_mul_f32 but not into v_fmac_f32 (it leaves a v_mov_b32_dpp in front of
// every fma), so the two kernels of the step are written out.  Hazards the assembler does not see inside a block: a DPP
// operand needs two wait states after a VALU write of its register (the leading s_nop: the operand may have just been
// copied), and so do the compiler's DPP moves that consume a block's result (trailing s_nop of quad_qmul, whose
// result feeds quad_derive).
#define STAC_DPP(p) " quad_perm:[" p "] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
// component c - 1 of rotate(v, q) in lane c >= 1 (vd = the lane's component of v); lane 0 holds no meaning
__device__ __forceinline__ float quad_rotate(const float vd, const QuadQuat &Q) {
    float uv, m, r;
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %3, %4" STAC_DPP("1,1,1,1")    // uv = v.x u.x
        "v_mul_f32_dpp %1, %3, -%8" STAC_DPP("0,2,3,1")   // m = -(u_(c+2) v_(c+1))
        "v_fmac_f32_dpp %0, %3, %5" STAC_DPP("2,2,2,2")   //    + v.y u.y
        "v_fmac_f32_dpp %1, %3, %7" STAC_DPP("0,3,1,2")   // m = u_(c+1) v_(c+2) - u_(c+2) v_(c+1)
        "v_fmac_f32_dpp %0, %3, %6" STAC_DPP("3,3,3,3")   //    + v.z u.z
        "v_add_f32 %0, %0, %0\n\t"                        // t = 2 u.v
        "v_mul_f32 %2, %0, %9\n\t"                        // t u_c
        "v_fmac_f32 %2, %10, %3\n\t"                      // + k v_c
        "v_fmac_f32 %2, %11, %1"                          // + s2 (u x v)_c
        : "=&v"(uv), "=&v"(m), "=&v"(r)
        : "v"(vd), "v"(Q.UX), "v"(Q.UY), "v"(Q.UZ), "v"(Q.U1), "v"(Q.U2), "v"(Q.qc), "v"(Q.K), "v"(Q.S2));
    return r;
}
// component c of qmul(q, v) (vq = the lane's component of v)
__device__ __forceinline__ float quad_qmul(const QuadQuat &Q, const float vq) {
    float r;
    asm("s_nop 0\n\t"
        "v_mul_f32 %0, %1, %2\n\t"
        "v_fmac_f32_dpp %0, %2, %3" STAC_DPP("1,0,3,2")
        "v_fmac_f32_dpp %0, %2, %4" STAC_DPP("2,3,0,1")
        "v_fmac_f32_dpp %0, %2, %5" STAC_DPP("3,2,1,0")
        "s_nop 1"
        : "=&v"(r)
        : "v"(Q.S), "v"(vq), "v"(Q.SX), "v"(Q.SY), "v"(Q.SZ));
    return r;
}
// The common step in one block: the three chains that depend on the incoming quaternion only -- body offset
// rotate(bpos, q), anchor offset rotate(jpos, q), and q * ql with the broadcasts / u.u / k of the product -- are
// interleaved by hand (a lone wavefront issues a DEPENDENT VALU instruction only every ~8 cycles, an independent
// one every 4; the compiler schedules an asm block as a unit).  Same operations, same operands, same order per
// result as quad_rotate / quad_qmul / quad_derive.  On return: pcb = pc + rotate(bpos, q) (BODY) or pc;
// ra = rotate(jpos, q); Qn = everything of the new quaternion except S2 and the signed copies.
__device__ __forceinline__ void quad_joint_fused(const float pc, const QuadQuat &Q, const float vb, const float vj, const float ql,
                                                 float &pcb, float &ra, QuadQuat &Qn) {
    float r, nS, nUX, nUY, nUZ, nU1, nU2, nK, tB, tA, mB;
    asm("s_nop 0\n\t"
        "v_mul_f32 %0, %13, %27\n\t"                          // M  r = s ql_c
        "v_mul_f32_dpp %8, %25, %14" STAC_DPP("1,1,1,1")      // B  uv = bpos.x u.x
        "v_mul_f32_dpp %9, %26, %14" STAC_DPP("1,1,1,1")      // A  uv = jpos.x u.x
        "v_fmac_f32_dpp %0, %27, %22" STAC_DPP("1,0,3,2")     // M
        "v_mul_f32_dpp %10, %25, -%18" STAC_DPP("0,2,3,1")    // B  m = -(u_(c+2) bpos_(c+1))
        "v_mul_f32_dpp %11, %26, -%18" STAC_DPP("0,2,3,1")    // A
        "v_fmac_f32_dpp %0, %27, %23" STAC_DPP("2,3,0,1")     // M
        "v_fmac_f32_dpp %8, %25, %15" STAC_DPP("2,2,2,2")     // B
        "v_fmac_f32_dpp %9, %26, %15" STAC_DPP("2,2,2,2")     // A
        "v_fmac_f32_dpp %0, %27, %24" STAC_DPP("3,2,1,0")     // M  r = component c of q * ql
        "v_fmac_f32_dpp %10, %25, %17" STAC_DPP("0,3,1,2")    // B
        "v_fmac_f32_dpp %11, %26, %17" STAC_DPP("0,3,1,2")    // A
        "v_fmac_f32_dpp %8, %25, %16" STAC_DPP("3,3,3,3")     // B
        "v_fmac_f32_dpp %9, %26, %16" STAC_DPP("3,3,3,3")     // A
        "v_mov_b32_dpp %1, %0" STAC_DPP("0,0,0,0")            // M  broadcasts of the product
        "v_mov_b32_dpp %2, %0" STAC_DPP("1,1,1,1")
        "v_mov_b32_dpp %3, %0" STAC_DPP("2,2,2,2")
        "v_mov_b32_dpp %4, %0" STAC_DPP("3,3,3,3")
        "v_add_f32 %8, %8, %8\n\t"                            // B  t = 2 u.v
        "v_add_f32 %9, %9, %9\n\t"                            // A
        "v_mul_f32 %7, %2, %2\n\t"                            // M  u.u
        "v_mul_f32 %8, %8, %12\n\t"                           // B  t u_c
        "v_mul_f32 %9, %9, %12\n\t"                           // A
        "v_fmac_f32 %7, %3, %3\n\t"                           // M
        "v_fmac_f32 %8, %19, %25\n\t"                         // B  + k v_c
        "v_fmac_f32 %9, %19, %26\n\t"                         // A
        "v_fmac_f32 %7, %4, %4\n\t"                           // M
        "v_fmac_f32 %8, %20, %10\n\t"                         // B  + s2 (u x v)_c
        "v_fmac_f32 %9, %20, %11\n\t"                         // A
        "v_fma_f32 %7, %1, %1, -%7\n\t"                       // M  k = s^2 - u.u
        "v_mov_b32_dpp %5, %0" STAC_DPP("0,2,3,1")            // M  u_(c+1)
        "v_mov_b32_dpp %6, %0" STAC_DPP("0,3,1,2")            // M  u_(c+2)
        "v_add_f32 %10, %21, %8"                              // B  pcb = pc + rotate(bpos, q)
        : "=&v"(r), "=&v"(nS), "=&v"(nUX), "=&v"(nUY), "=&v"(nUZ), "=&v"(nU1), "=&v"(nU2), "=&v"(nK), "=&v"(tB), "=&v"(tA),
          "=&v"(mB), "=&v"(ra)
        : "v"(Q.qc), "v"(Q.S), "v"(Q.UX), "v"(Q.UY), "v"(Q.UZ), "v"(Q.U1), "v"(Q.U2), "v"(Q.K), "v"(Q.S2), "v"(pc), "v"(Q.SX),
          "v"(Q.SY), "v"(Q.SZ), "v"(vb), "v"(vj), "v"(ql));
    pcb = mB;
    ra = tA;  // (the block kept the anchor chain's cross term in `ra`)
    Qn.qc = r; Qn.S = nS; Qn.UX = nUX; Qn.UY = nUY; Qn.UZ = nUZ; Qn.U1 = nU1; Qn.U2 = nU2; Qn.K = nK;
}
struct FkQuadRegs {
    float vb, vj;  // the lane's component of bpos / jpos
    int4 o;        // par_off, ja_off, xf_off, ql_next
    float bq;      // the lane's component of body_quat (records of 16 words)
    float ql;      // the lane's component of the step's joint-local quaternion
};
struct FkQuadLane {
    int c;           // 0 .. 3 = w, x, y, z
    int voff;        // component offset inside a record vector: c - 1 (lane 0: 0, unused)
    int poff;        // word of the lane's position component inside a transform entry; its quaternion component is kXq
                     // words behind (entries hold x, y, z, w).  Lane 0 has no position component: it aims at word 3,
                     // which the quaternion half of the same ds_write2_b32 -- lane 1's x, or its own w with padded
                     // entries -- overwrites (data0 of all lanes is written before data1: build/micro/w2order.hip)
    int m1, m2, m3;  // sign masks of the lane's row of the quaternion product
};
// {position component, quaternion component} of the lane into the entry at word offset `off`: one LDS instruction
__device__ __forceinline__ void quad_store(float *CBc, const int off, const FkQuadLane &L, const float p, const float q) {
    // two plain stores, position component first: the compiler merges them into one ds_write2_b32 (data0 = the lower
    // offset).  An asm statement here would make every wave_sync() drain the LDS queue (an opaque memory clobber).
    float *e = CBc + off + L.poff;
    e[0] = p;
    e[kXq] = q;
}
template <int RW>
__device__ __forceinline__ void fk_fetch_quad(FkQuadRegs &R, const float *rec, const float *CBc, const int ql_off, const FkQuadLane &L) {
    R.vb = rec[L.voff];
    R.vj = rec[8 + L.voff];
    R.o = lds4i(rec + 4);
    if constexpr (RW == 16) R.bq = rec[12 + L.c];
    R.ql = CBc[ql_off + L.poff];  // (ql_off = the quaternion words x, y, z, w of an entry)
}
// A step by its flags (any combination).  Only lanes with a position run the program (the caller masks the rest), and
// a step stores to the sink entry what nobody reads, so nothing here is predicated per lane but the parent load.
template <int RW>
__device__ __forceinline__ void fk_step_quad_general(const FkQuadRegs &R, const float *rec, const int mlf, float &pc, QuadQuat &Q,
                                                     float *CBc, const float *qe, const float *jrec, const FkQuadLane &L) {
    if (mlf & FK_ML_PARENT_LDS) {
        float qc = Q.qc;
        if (R.o.x >= 0) {
            pc = CBc[R.o.x + L.poff];
            qc = CBc[R.o.x + L.poff + kXq];
        }
        quad_derive(Q, qc, L.m1, L.m2, L.m3);
    }
    if (mlf & FK_ML_BODY) pc = pc + quad_rotate(R.vb, Q);
    if constexpr (RW == 16) {
        if (mlf & FK_ML_BQUAT) quad_derive(Q, quad_qmul(Q, R.bq), L.m1, L.m2, L.m3);
    }
    const float pos0 = pc, prequat = Q.qc;
    float anchor = pc;
    if (mlf & FK_ML_SPECIAL) {  // free / slide joints somewhere in this micro-level
        const QuadQuat Qpre = Q;
        if (mlf & FK_ML_JOINT) {
            if (mlf & FK_ML_JPOS) anchor = quad_rotate(R.vj, Q) + pc;
            quad_derive(Q, quad_qmul(Q, R.ql), L.m1, L.m2, L.m3);
            if (mlf & FK_ML_JPOS) pc = anchor - quad_rotate(R.vj, Q);
        }
        float qc = Q.qc;
        const int kind = __builtin_bit_cast(int, rec[3]), aux = __builtin_bit_cast(int, rec[11]);  // (not part of the prefetch)
        const float axis = quad_rotate(jrec[12 * (kind == FK_KIND_SLIDE ? aux : 0) + 8 + L.voff], Qpre);
        const float d = quad_dpp<0x00>(R.ql);
        if (kind == FK_KIND_FREE) {
            anchor = qe[aux + L.voff];
            pc = anchor;
            qc = R.ql;  // normalised by the pre-pass
        } else if (kind == FK_KIND_SLIDE) {
            qc = prequat;
            pc = FMA(axis, d, pos0);
        }
        quad_derive(Q, qc, L.m1, L.m2, L.m3);
    } else if (mlf & FK_ML_JOINT) {
        if (mlf & FK_ML_JPOS) anchor = quad_rotate(R.vj, Q) + pc;
        quad_derive(Q, quad_qmul(Q, R.ql), L.m1, L.m2, L.m3);
        if (mlf & FK_ML_JPOS) pc = anchor - quad_rotate(R.vj, Q);
    }
    quad_store(CBc, R.o.y, L, anchor, prequat);
    quad_store(CBc, R.o.z, L, pc, Q.qc);
}
// The plain step, straight-line: hinge / ball joints on every position, on neutral data where a part does not apply
// (body_pos = 0: the step does not start a body; jnt_pos = 0 and an identity joint quaternion: no joint).  PARENT: some
// positions load their parent's transform (issued ahead of the next record's fetch: the loads return in order).
template <int RW, bool PARENT>
__device__ __forceinline__ void fk_step_quad_joint(const FkQuadRegs &R, FkQuadRegs &N, const float *next_rec, float &pc, QuadQuat &Q,
                                                   float *CBc, const FkQuadLane &L) {
    if constexpr (PARENT) {
        const int po = R.o.x >= 0 ? R.o.x : R.o.z;  // (any valid entry: the loaded values are dropped)
        const float pl = CBc[po + L.poff], ql = CBc[po + L.poff + kXq];
        fk_fetch_quad<RW>(N, next_rec, CBc, R.o.w, L);
        pc = R.o.x >= 0 ? pl : pc;
        quad_derive(Q, R.o.x >= 0 ? ql : Q.qc, L.m1, L.m2, L.m3);
    } else {
        fk_fetch_quad<RW>(N, next_rec, CBc, R.o.w, L);
    }
    float pcb, ra;
    QuadQuat Qn;
    quad_joint_fused(pc, Q, R.vb, R.vj, R.ql, pcb, ra, Qn);
    Qn.S2 = Qn.S + Qn.S;
    Qn.SX = __builtin_bit_cast(float, __builtin_bit_cast(int, Qn.UX) ^ L.m1);
    Qn.SY = __builtin_bit_cast(float, __builtin_bit_cast(int, Qn.UY) ^ L.m2);
    Qn.SZ = __builtin_bit_cast(float, __builtin_bit_cast(int, Qn.UZ) ^ L.m3);
    const float anchor = ra + pcb;
    quad_store(CBc, R.o.y, L, anchor, Q.qc);
    pc = anchor - quad_rotate(R.vj, Qn);
    quad_store(CBc, R.o.z, L, pc, Qn.qc);
    Q = Qn;
}
template <int RW>
__device__ __forceinline__ void fk_step_quad(const FkQuadRegs &R, FkQuadRegs &N, const float *rec, const float *next_rec, const int code,
                                             float &pc, QuadQuat &Q, float *CBc, const float *qe, const float *jrec, const FkQuadLane &L) {
    // Every form issues its LDS operations in the same pattern -- [parent loads] fetch of the next record, two stores --
    // because the forms join again before the next step: s_waitcnt counts are merged over all predecessors, and one form
    // with fewer operations behind its fetch would make EVERY step wait for the stores of the step before to complete.
    const int form = code >> 8;
    // Two straight-line forms and the general one.  More forms -- one per flag combination, without the arithmetic a
    // step does not need -- are each faster alone (a program of BODY steps only: 195 cycles per step, of QJOINT steps
    // 208, against 385 / 486 for the two below), but every alternative is one more block of the structurised control
    // flow that EVERY step walks through: ten forms cost the rodent's 14-step program 10.1 k cycles, against 5.9 k
    // for the sum of its steps.  The arithmetic is not what a step costs: without it the program takes 3 % less.
    if (form == FK_FORM_BODY_JOINT) fk_step_quad_joint<RW, false>(R, N, next_rec, pc, Q, CBc, L);
    else if (form == FK_FORM_PARENT_BODY_JOINT) fk_step_quad_joint<RW, true>(R, N, next_rec, pc, Q, CBc, L);
    else {
        fk_fetch_quad<RW>(N, next_rec, CBc, R.o.w, L);
        fk_step_quad_general<RW>(R, rec, code & 255, pc, Q, CBc, qe, jrec, L);
    }
    wave_sync();
}
template <int RW, bool PSEL, class HT>
__device__ __forceinline__ void fk_program_quad(const HT &H, const float *P, float *CBc, const int lf, const int gf,
                                                const bool active, const int prog_off, const int n_ml_even, const int n_run) {
    const int W = H.max_width;
    const int pp = lf >> 2, c = lf & 3;
    if (!(active && pp < W)) return;  // whole quads: the DPP operands below stay inside a quad
    FkQuadLane L;
    L.c = c;
    L.voff = c ? c - 1 : 0;
    L.poff = c ? c - 1 : 3;
    const int sgn = (int)0x80000000u;
    L.m1 = (c == 0 || c == 2) ? sgn : 0;
    L.m2 = (c == 0 || c == 3) ? sgn : 0;
    L.m3 = (c == 0 || c == 1) ? sgn : 0;
    const int *hdr = reinterpret_cast<const int *>(P + prog_off);
    const float *sp = P + prog_off + H.fk_hdr_words + RW * pp;
    const float *jrec = P + H.off_joint;
    const float *qe = CBc + H.c_qe;
    float pc = 0.f;
    QuadQuat Q;
    quad_derive(Q, c == 0 ? 1.f : 0.f, L.m1, L.m2, L.m3);
    FkQuadRegs A, B;
    A.bq = B.bq = c == 0 ? 1.f : 0.f;
    int fl_v = hdr[0];
    fk_fetch_quad<RW>(A, sp, CBc, hdr[(H.n_mlev_hdr >> 1) + pp], L);
    quad_store(CBc, H.c_sink, L, pc, Q.qc);  // (the pattern of a step: see fk_step_quad)
    quad_store(CBc, H.c_sink, L, pc, Q.qc);
    const int stride = RW * W;
    if (H.fk_uniform) {
        // every step is the same straight-line code (parents by select, body / joint parts on neutral data): no flags, no
        // dispatch, nothing for s_waitcnt to be conservative about
        const int n_ml = n_run > 0 ? n_run : n_ml_even;  // (stops after the last step with work)
        int ml = 0;
        for (; ml + 1 < n_ml; ml += 2) {
            const int fl = __builtin_amdgcn_readfirstlane(fl_v);
            fl_v = hdr[(ml >> 1) + 1];
            sp += stride;
            // PSEL: the parent part only in the steps that have one, behind one wave-uniform branch (measured: +1.5 % where
            // several chains share the wavefront's instruction stream, -2.7 % in latency mode, where the step without any
            // branch wins)
            if (!PSEL || (fl & FK_ML_PARENT_LDS)) fk_step_quad_joint<RW, true>(A, B, sp, pc, Q, CBc, L);
            else fk_step_quad_joint<RW, false>(A, B, sp, pc, Q, CBc, L);
            wave_sync();
            if (ml + 2 < n_ml_even) sp += stride;  // (no record behind the last one)
            if (!PSEL || ((fl >> 16) & FK_ML_PARENT_LDS)) fk_step_quad_joint<RW, true>(B, A, sp, pc, Q, CBc, L);
            else fk_step_quad_joint<RW, false>(B, A, sp, pc, Q, CBc, L);
            wave_sync();
        }
        if (ml < n_ml) {  // an odd number of steps with work (root-pass programs): the last one, outside the loop
            const int fl = __builtin_amdgcn_readfirstlane(fl_v);
            sp += stride;
            if (!PSEL || (fl & FK_ML_PARENT_LDS)) fk_step_quad_joint<RW, true>(A, B, sp, pc, Q, CBc, L);
            else fk_step_quad_joint<RW, false>(A, B, sp, pc, Q, CBc, L);
            wave_sync();
        }
        return;
    }
    const int n_ml = n_ml_even;
    for (int ml = 0; ml < n_ml; ml += 2) {
        const int fl = __builtin_amdgcn_readfirstlane(fl_v);
        fl_v = hdr[(ml >> 1) + 1];
        const float *r0 = sp;
        sp += stride;
        const float *r1 = sp;
        fk_step_quad<RW>(A, B, r0, sp, fl & 0xFFFF, pc, Q, CBc, qe, jrec, L);
        if (ml + 2 < n_ml) sp += stride;
        fk_step_quad<RW>(B, A, r1, sp, (fl >> 16) & 0xFFFF, pc, Q, CBc, qe, jrec, L);
    }
}

// FK of one chain by gf lanes: the program when every level fits the lanes (four lanes per position when QUAD and
// they fit four times over), else the level loop.
// n_ml_root > 0 (wave-uniform): every chain of the wavefront is in a root pass -- run the pruned program at off_fkroot.
template <bool QUAD, bool PSEL = false, class HT>
__device__ __forceinline__ void fk_chain(const HT &H, const float *P, float *CBc, const int lf, const int gf,
                                         const bool active, const bool store_ja, const bool use_levels,
                                         const int n_ml_root = 0, const int n_run_root = 0) {
    if (H.max_width <= gf && !use_levels) {
        const int prog_off = n_ml_root > 0 ? H.off_fkroot : H.off_fkstep;
        const int n_ml = n_ml_root > 0 ? n_ml_root : H.n_mlev;
        if constexpr (QUAD) {
            if (4 * H.max_width <= gf) {
                const int n_run = n_ml_root > 0 ? n_run_root : 0;
                if (H.fk_rec_words == 16) fk_program_quad<16, PSEL>(H, P, CBc, lf, gf, active, prog_off, n_ml, n_run);
                else fk_program_quad<12, PSEL>(H, P, CBc, lf, gf, active, prog_off, n_ml, n_run);
                return;
            }
        }
        if (H.fk_rec_words == 16) fk_program<16>(H, P, CBc, lf, gf, active, store_ja, prog_off, n_ml);
        else fk_program<12>(H, P, CBc, lf, gf, active, store_ja, prog_off, n_ml);
    } else {
        fk_levels(H, P, CBc, lf, gf, active, store_ja);
    }
}

// ------------------------------------------------------------------------------------------------
// split kinematics of the lean kernels (PlanHeader::fk3; tables: stac_abi.hip, build_fk3_program)
// ------------------------------------------------------------------------------------------------
// The step program above runs, per body and joint, pos += rotate(bpos, q); anchor = rotate(jpos, q) + pos; q = q * ql;
// pos = anchor - rotate(jpos, q) as ONE serial chain of ~60-instruction steps.  With a free root and only hinges below it the
// quaternions do not depend on the positions, so the same operations -- same operands, same order per result: bit-identical --
// are issued as three passes of which only the cheap ones are serial:
//   P1  q_j = q_pre(j) * ql_j                 four lanes per product; the running quaternion is the DPP operand, the signed
//                                             permutations of ql (known before the chain starts) are prepared beside it
//   P2  r = rotate(v, q_node)                 one lane per rotation (the scalar code's own sequence), all lanes busy
//   P3  p = p + r                             one addition per rotation, in place in the rotation's slot; three lanes per path
// P1 and P3 run uniform blocks of steps (two / four): no flagged steps, no runs, nothing a block waits for that it requested itself.
// A position that has no work in a step multiplies / adds whatever is there and stores to a sink / its own unused slot: its running
// value is garbage until its next path begins, and a path always begins with a restart at the head of a block (a select).
struct Fk3Lane {
    int pp, c;       // lane position (0 .. 3) and component (P1: w, x, y, z; P3: -, x, y, z)
    int m1, m2, m3;  // sign masks of the lane's row of the quaternion product (as FkQuadLane)
};
__device__ __forceinline__ Fk3Lane fk3_lane(const int lf) {
    Fk3Lane L;
    L.pp = lf >> 2;
    L.c = lf & 3;
    const int sgn = (int)0x80000000u;
    L.m1 = (L.c == 0 || L.c == 2) ? sgn : 0;
    L.m2 = (L.c == 0 || L.c == 3) ? sgn : 0;
    L.m3 = (L.c == 0 || L.c == 1) ? sgn : 0;
    return L;
}
// component c of q * ql from the lane's component of each: r_c = q.w R0 + q.x R1 + q.y R2 + q.z R3 with R_k = +-ql_(c ^ k)
// (qmul's own products in qmul's own order: the signs sit on the ql factors, fma(-a, b, c) == fma(a, -b, c)).  Hazards: ql comes
// from LDS; qc -- the DPP operand of the products -- may have been written by the VALU instruction in front of the block: the
// three permutations of ql stand between (tests/test_isa_hazards.py checks the distance in the binary).
__device__ __forceinline__ float fk3_qmul(const float qc, const float ql, const Fk3Lane &L) {
    float r, R1, R2, R3;
    asm("v_xor_b32_dpp %1, %5, %6" STAC_DPP("1,0,3,2")
        "v_xor_b32_dpp %2, %5, %7" STAC_DPP("2,3,0,1")
        "v_xor_b32_dpp %3, %5, %8" STAC_DPP("3,2,1,0")
        "v_mul_f32_dpp %0, %4, %5" STAC_DPP("0,0,0,0")
        "v_fmac_f32_dpp %0, %4, %1" STAC_DPP("1,1,1,1")
        "v_fmac_f32_dpp %0, %4, %2" STAC_DPP("2,2,2,2")
        "v_fmac_f32_dpp %0, %4, %3 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(r), "=&v"(R1), "=&v"(R2), "=&v"(R3)
        : "v"(qc), "v"(ql), "v"(L.m1), "v"(L.m2), "v"(L.m3));
    return r;
}
// P1.  Blocks of two steps, every block the same instructions (as P3 below): a position keeps its running quaternion or starts the
// block from a restart value.  T1: records of 24 words (layout: build_fk3_program, stac_abi.hip); block r works from record r + 1 --
// its out words and restart flag, the ql and restart words of block r + 1 --, so a block's joint-local quaternions and its restart
// value are requested while the block before it runs (the host schedules a restart at least three steps behind the step that wrote
// the value) and its record a block before that.  Two blocks per trip, the two sets of registers in turn: nothing is copied.
__device__ __forceinline__ float fk3_restart_value(const float *base, const int e) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + ((unsigned)e << 2));  // (bits 30, 31 leave: one v_lshl_add)
}
__device__ __forceinline__ int2 lds2i(const float *p) { return *reinterpret_cast<const int2 *>(p); }
// (Fk3Head: the first two records of the lane's position, which the lean latency kernels keep in registers -- the pass then starts with
//  its quaternions' round trip, not with the records' and then theirs)
struct Fk3Head { int ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3, pa0, pa1, pb0, pb1; };  // (plain words: a struct of vectors went through scratch)
__device__ __forceinline__ Fk3Head fk3_p1_head(const float *T1, const Fk3Lane &L) {
    const float *rp = T1 + 4 * L.pp, *ep = T1 + 16 + 2 * L.pp;
    const int4 a = lds4i(rp), b = lds4i(rp + 24);
    const int2 c = lds2i(ep), d = lds2i(ep + 24);
    return Fk3Head{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, d.x, d.y};
}
__device__ __forceinline__ void fk3_p1(const float *T1, const int nb, float *CBc, const Fk3Lane &L, const bool use_head, const Fk3Head head) {
    float *base = CBc + L.c;
    const float *rp = T1 + 4 * L.pp, *ep = T1 + 16 + 2 * L.pp;
    int4 Ra, Rb;
    int2 Pa, Pb;
    if (use_head) {  // (wave-uniform)
        Ra = int4{head.ra0, head.ra1, head.ra2, head.ra3}; Rb = int4{head.rb0, head.rb1, head.rb2, head.rb3};
        Pa = int2{head.pa0, head.pa1}; Pb = int2{head.pb0, head.pb1};
    }
    else { Ra = lds4i(rp); Pa = lds2i(ep); Rb = lds4i(rp + 24); Pb = lds2i(ep + 24); }
    float qa = base[Ra.x], qb = base[Ra.y];
    float rl = fk3_restart_value(base, Pa.y);
    float qc = 0.0f;
    for (int b = 0; b + 2 <= nb; b += 2) {
        rp += 48; ep += 48;
        {   // block b: record b + 1 in Rb / Pb, its quaternions in qa, qb, rl
            Ra = lds4i(rp);
            Pa = lds2i(ep);
            const float na = base[Rb.x], nq = base[Rb.y];
            const float rn = fk3_restart_value(base, Pb.y);  // (before this block's stores: it sees the blocks before this one)
            qc = Pb.x >= 0 ? rl : qc;
            const float q1 = fk3_qmul(qc, qa, L);
            base[Rb.z] = q1;
            qc = fk3_qmul(q1, qb, L);
            base[Rb.w] = qc;
            qa = na; qb = nq; rl = rn;
        }
        {   // block b + 1: record b + 2 in Ra / Pa
            Rb = lds4i(rp + 24);
            Pb = lds2i(ep + 24);
            const float na = base[Ra.x], nq = base[Ra.y];
            const float rn = fk3_restart_value(base, Pa.y);
            qc = Pa.x >= 0 ? rl : qc;
            const float q1 = fk3_qmul(qc, qa, L);
            base[Ra.z] = q1;
            qc = fk3_qmul(q1, qb, L);
            base[Ra.w] = qc;
            qa = na; qb = nq; rl = rn;
        }
    }
    if (nb & 1) {  // the last block of an odd count: nothing left to request
        qc = Pb.x >= 0 ? rl : qc;
        const float q1 = fk3_qmul(qc, qa, L);
        base[Rb.z] = q1;
        base[Rb.w] = fk3_qmul(q1, qb, L);
    }
}
// P2.  T2: tasks {v.x, v.y, v.z, quaternion word | result word << 16}, n2 a multiple of 32 (no-op tasks at the end).  PAIR (throughput
// kernels at 16 lanes): two rounds of lanes at once -- their loads in flight together, their arithmetic, chains of dependent
// instructions, interleaved (a lone wavefront of the latency kernels pays per instruction, dependent or not: one round at a time).
template <bool PAIR>
__device__ __forceinline__ void fk3_p2(const float *T2, const int n2, float *CBc, const int lf, const int gf) {
    if constexpr (PAIR) {
        for (int i0 = 0; i0 < n2; i0 += 2 * gf) {  // (whole rounds: no lane-dependent trip count; 2 gf = 32 divides n2)
            const float4 ta = lds4(T2 + 4 * (i0 + lf));
            const float4 tb = lds4(T2 + 4 * (i0 + gf + lf));
            const int wa = __builtin_bit_cast(int, ta.w), wb = __builtin_bit_cast(int, tb.w);
            const float4 qa = lds4(CBc + (wa & 0xFFFF)), qb = lds4(CBc + (wb & 0xFFFF));  // (w, x, y, z)
            const V3 ra = rotate(V3{ta.x, ta.y, ta.z}, Q4{qa.x, qa.y, qa.z, qa.w});
            const V3 rb = rotate(V3{tb.x, tb.y, tb.z}, Q4{qb.x, qb.y, qb.z, qb.w});
            float *oa = CBc + (int)((unsigned)wa >> 16), *ob = CBc + (int)((unsigned)wb >> 16);
            oa[0] = ra.x; oa[1] = ra.y; oa[2] = ra.z;
            ob[0] = rb.x; ob[1] = rb.y; ob[2] = rb.z;
        }
    } else {
        // (a lone wavefront waits out every LDS round trip: a round's task comes two rounds ahead, its quaternion one; the rounds
        //  behind the last one read the last round again)
        const float *tp = T2 + 4 * lf;
        const int last = n2 - gf;
        float4 tk = lds4(tp), tn = lds4(tp + 4 * min(gf, last));
        float4 q4 = lds4(CBc + (__builtin_bit_cast(int, tk.w) & 0xFFFF));  // (w, x, y, z)
        for (int i0 = 0; i0 < n2; i0 += gf) {
            const float4 tnn = lds4(tp + 4 * min(i0 + 2 * gf, last));
            const float4 qn = lds4(CBc + (__builtin_bit_cast(int, tn.w) & 0xFFFF));
            const V3 r = rotate(V3{tk.x, tk.y, tk.z}, Q4{q4.x, q4.y, q4.z, q4.w});
            float *o = CBc + (int)((unsigned)__builtin_bit_cast(int, tk.w) >> 16);
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            tk = tn; tn = tnn; q4 = qn;
        }
    }
}
// P2 of the lean latency kernels (round 6): the lane's tasks of the first NR rounds come out of registers (they never change: the
// kernel reads them once), so a round is one LDS round trip -- its quaternion -- instead of two dependent ones; the NR quaternions are
// requested together.  Rounds behind the NR-th run as before.
template <int NR>
__device__ __forceinline__ void fk3_p2_pinned(const float4 (&tk)[NR], const float *T2, const int n2, float *CBc, const int lf, const int gf) {
    float4 q[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r)
        if (r * gf < n2) q[r] = lds4(CBc + (__builtin_bit_cast(int, tk[r].w) & 0xFFFF));  // (w, x, y, z)
#pragma unroll
    for (int r = 0; r < NR; ++r)
        if (r * gf < n2) {
            const V3 rt = rotate(V3{tk[r].x, tk[r].y, tk[r].z}, Q4{q[r].x, q[r].y, q[r].z, q[r].w});
            float *o = CBc + (int)((unsigned)__builtin_bit_cast(int, tk[r].w) >> 16);
            o[0] = rt.x; o[1] = rt.y; o[2] = rt.z;
        }
    if (NR * gf < n2) fk3_p2<false>(T2 + 4 * NR * gf, n2 - NR * gf, CBc, lf, gf);
}
// P3.  The slot of (step t, position pp) is pb + 3 * (4 t + pp).  Blocks of four steps, every block the same instructions: a position
// either keeps its running value or starts the block from a restart value, T3[4 b + pp] = word of that value, or bit 31 | a valid word.
// Everything a block reads is requested while the block before it runs -- its four operands, its restart value (the host schedules a
// restart at least five steps behind the step that wrote the value: fk3_schedule); the restart words come three blocks ahead -- so a
// lone wavefront pays the LDS round trip once per pass, not once per step.
__device__ __forceinline__ void fk3_p1(const float *T1, const int nb, float *CBc, const Fk3Lane &L) {
    fk3_p1(T1, nb, CBc, L, false, Fk3Head{});
}
__device__ __forceinline__ void fk3_p3(const int *T3, const int nb, float *CBc, const int pb, const Fk3Lane &L, const bool use_head = false,
                                       const int h0 = 0, const int h1 = 0, const int h2 = 0) {
    const int cc = L.c ? L.c - 1 : 0;  // (lane 0 of a quad doubles lane 1: the same loads, the same values stored to the same words)
    const float *base = CBc + cc;
    float *slot = CBc + pb + 3 * L.pp + cc;
    const int *rp = T3 + L.pp;
    // (the words two blocks ahead of their use: no block waits for the one it requested last; head3: the first three out of registers)
    int e0 = h0, e1 = h1, e2 = h2;
    if (!use_head) { e0 = rp[0]; e1 = rp[4]; e2 = rp[8]; }  // (wave-uniform)
    float v0 = slot[0], v1 = slot[12], v2 = slot[24], v3 = slot[36];
    float rl = fk3_restart_value(base, e0);
    float p = 0.0f;
#pragma unroll 2  // (measured 1 / 2 / 4: 554 / 575 / 570 k frames/s, 18.8 / 19.3 / 18.9 k on 40 x 250)
    for (int b = 0; b < nb; ++b) {
        const int e3 = rp[4 * b + 12];
        const float rn = fk3_restart_value(base, e1);  // (before this block's stores: it sees the blocks before this one)
        const float *ns = slot + (b + 1 < nb ? 48 : 0);
        const float n0 = ns[0], n1 = ns[12], n2 = ns[24], n3 = ns[36];
        p = e0 >= 0 ? rl : p;
        const float s0 = p + v0;
        const float s1 = s0 + v1;
        const float s2 = s1 + v2;
        const float s3 = s2 + v3;
        slot[0] = s0; slot[12] = s1; slot[24] = s2; slot[36] = s3;
        p = s3;
        slot += 48;
        e0 = e1; e1 = e2; e2 = e3; rl = rn;
        v0 = n0; v1 = n1; v2 = n2; v3 = n3;
    }
}
struct Fk3Prog { const float *T1, *T2; const int *T3, *site; int n1, n2, n3; };
// all three passes; lf = lane in the group of gf lanes (16 or 32; the first 16 run P1 and P3); THROUGHPUT: not a latency kernel
template <bool THROUGHPUT>
__device__ __forceinline__ void fk3_run(const Fk3Prog &G3, float *CBc, const int pb, const int lf, const int gf) {
    const Fk3Lane L = fk3_lane(lf & 15);
    if (gf == 16 || lf < 16) fk3_p1(G3.T1, G3.n1 >> 1, CBc, L);
    wave_sync();
    fk3_p2<THROUGHPUT>(G3.T2, G3.n2, CBc, lf, gf);
    wave_sync();
    if (gf == 16 || lf < 16) fk3_p3(G3.T3, G3.n3 >> 2, CBc, pb, L);
    wave_sync();
}

}  // namespace stac
