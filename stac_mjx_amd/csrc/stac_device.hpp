// stac_device.hpp -- device-side helpers shared by the kernels of libstac_hip.so (math that mirrors
// oracle/stac_oracle.c operation for operation, cross-lane sums, LDS access helpers).
#pragma once
#include <hip/hip_runtime.h>

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

enum : int { ST_VG_Y = 0, ST_LS = 1, ST_VG_X = 2, ST_DONE = 3, ST_SPEC = 4 };

// In-kernel phase stamps: diagnostic build only (-DSTAC_PROFILE -> libstac_hip_prof.so); the stamps
// go to a buffer of their own and feed no output.  Read the SHARES, not the run time.
#ifdef STAC_PROFILE
#define PROF_DECL unsigned long long pt0 = __builtin_readcyclecounter(), pacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PROF_TICK(i)                                             \
    do {                                                         \
        const unsigned long long pt1 = __builtin_readcyclecounter(); \
        pacc[i] += pt1 - pt0;                                    \
        pt0 = pt1;                                               \
    } while (0)
#define PROF_FLUSH(a)                                                                     \
    do {                                                                                  \
        if (a.prof && lane == 0)                                                          \
            for (int i = 0; i < 12; ++i) atomicAdd(a.prof + i, pacc[i]);                  \
    } while (0)
#else
#define PROF_DECL
#define PROF_TICK(i)
#define PROF_FLUSH(a)
#endif
enum : int { JFREE = 0, JBALL = 1, JSLIDE = 2, JHINGE = 3 };

// Chains never span wavefronts, so ordering this wave's own LDS traffic is enough: LDS operations of
// one wave execute in issue order; the fence stops the compiler from moving or caching LDS accesses
// across the point (it lowers to s_waitcnt lgkmcnt(0)), the wave barrier pins the schedule.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}


__device__ __forceinline__ float4 lds4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ int4 lds4i(const float *p) { return *reinterpret_cast<const int4 *>(p); }

}  // namespace stac
