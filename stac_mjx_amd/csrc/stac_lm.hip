// stac_lm.hip -- optional fast q_phase solver: projected Levenberg-Marquardt (STAC_SOLVER_LM).
//
// NOT the reference's algorithm (the reference runs jaxopt.ProjectedGradient, stac_mjx/stac_core.py:182-191,
// which stac_kernels.hip reproduces bit for bit).  This is the "LM qpos update" of BASELINE.json's north star
// (SURVEY.md section 7 step 6): same objective, masks, bounds, stopping residual and phase sequencing
// (compute_stac.py:17-104,170-278), judged in marker space against the reference's algorithm.  oracle/stac_oracle.c::q_opt_lm_ws
// is its CPU statement, operation for operation (loss sum, Gauss-Newton entries, pivots, substitutions, accept / reject turns):
// the GPU tests compare the two bit for bit.
//
// Mapping: like the PG kernel a wavefront is split into groups of G lanes, one chain per group, every trip
// round the main loop evaluates one point per group (FK over the marker-ancestor subtree, site residuals,
// analytic gradient).  An accepted point additionally gets its Gauss-Newton system: weighted site Jacobian
// blocks J[site][dof on its root path] in LDS, the structurally non-zero entries of J^T W J summed by one lane
// per entry over a contiguous range of DFS-ordered sites, damping A_ii (1 + lambda) + mu, coordinates at an
// active bound frozen, dense packed Cholesky + substitutions in LDS, step clipped to the box.
#include <hip/hip_runtime.h>

#include "stac_device.hpp"
#include "stac_plan.hpp"

namespace stac {

enum : int { LM_EVAL_X = 0, LM_TRIAL = 1, LM_DONE = 3 };


// Jacobian block d(site position)/d(dof): one column of the 3 x nd site Jacobian.
struct DofGeom {
    int ty, comp;
    V3 anchor, axis;  // joint anchor and world axis (hinge / slide)
    Q4 qh;            // normalised raw quaternion (free / ball joint)
    Q4 pre;           // ball joint: the quaternion in front of the joint (its rotation is applied in that frame)
    float dn;         // |q| (+1e-6 if 0)
};
// raw quaternion (s, u), q^ = q / |q|, d = the rotated vector: dx/ds = -2 (u^ x d) / |q|, dx/du_j = 2 (s^ (e_j x d) - (e_j x u^) x d) / |q|
__device__ __forceinline__ V3 quat_col(const Q4 qh, const float dn, const int c4, const V3 d) {
    const V3 u = {qh.x, qh.y, qh.z};
    if (c4 == 0) {
        const V3 c = cross3(u, d);
        return {(-2.0f * c.x) / dn, (-2.0f * c.y) / dn, (-2.0f * c.z) / dn};
    }
    const V3 e = {c4 == 1 ? 1.f : 0.f, c4 == 2 ? 1.f : 0.f, c4 == 3 ? 1.f : 0.f};
    const V3 exd = cross3(e, d), t2 = cross3(cross3(e, u), d);
    return {(2.0f * (qh.w * exd.x - t2.x)) / dn, (2.0f * (qh.w * exd.y - t2.y)) / dn, (2.0f * (qh.w * exd.z - t2.z)) / dn};
}
__device__ __forceinline__ V3 jac_col(const DofGeom &g, V3 sx) {
    const V3 d = sub3(sx, g.anchor);
    if (g.ty == JHINGE) return cross3(g.axis, d);
    if (g.ty == JSLIDE) return g.axis;
    if (g.ty == JBALL) {  // in the frame the ball rotation is applied in, and back
        const V3 dl = rotate(d, Q4{g.pre.w, -g.pre.x, -g.pre.y, -g.pre.z});
        return rotate(quat_col(g.qh, g.dn, g.comp, dl), g.pre);
    }
    if (g.comp < 3) return {g.comp == 0 ? 1.f : 0.f, g.comp == 1 ? 1.f : 0.f, g.comp == 2 ? 1.f : 0.f};
    return quat_col(g.qh, g.dn, g.comp - 3, d);
}

template <int G, int NQR, int WPE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void q_phase_lm_kernel(const QArgs a, const LmArgs L) {
    extern __shared__ float lds[];
    constexpr int CPW = 64 / G;
    const PlanHeader &H = a.h;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int grp = lane / G, lg = lane % G;
    const int nq = H.nq, K = H.K, nqpad = H.nqpad;

    // ---- plan, mask bits, LM kind headers into LDS ----------------------------------------------------------
    float *P = lds;
    for (int i = threadIdx.x; i < H.total_words; i += blockDim.x) P[i] = a.plan[i];
    const int plan_words = (H.total_words + 3) & ~3;
    uint32_t *MB = reinterpret_cast<uint32_t *>(lds + plan_words);  // [2][nkinds][G]: qs_to_opt bits, dof bits
    const int nkinds = a.P + 3;
    int *KH = reinterpret_cast<int *>(lds + plan_words + a.mb_words);  // LmKind[nkinds]
    for (int i = threadIdx.x; i < nkinds * kLmKindWords; i += blockDim.x) KH[i] = L.tab[i];
    const int khh_words = (nkinds * kLmKindWords + 3) & ~3;
    int *KT = KH + khh_words;  // hot tables of every kind: dof records and root-path tables (walked in dependent loops)
    for (int i = threadIdx.x; i < L.hot_words; i += blockDim.x) KT[i] = L.tab[L.hot_off + i];
    // TRI[q] = (row << 8 | col) of the q-th entry of a packed lower triangle (pair enumeration of the L^T D L step)
    int *TRI = KT + ((L.hot_words + 3) & ~3);
    const int ntri = (L.maxpd * (L.maxpd + 1)) >> 1;
    for (int r = threadIdx.x; r < L.maxpd; r += blockDim.x)
        for (int c = 0; c <= r; ++c) TRI[((r * (r + 1)) >> 1) + c] = (r << 8) | c;
    __syncthreads();
    for (int i = threadIdx.x; i < nkinds * G; i += blockDim.x) {
        const int kind = i / G, l = i % G;
        uint32_t bits = 0, dbits = 0;
        for (int r = 0; r < NQR; ++r) {
            const int e = r * G + l;
            if (e < nq && a.masks[kind * nqpad + e]) bits |= (1u << r);
        }
        const LmKind *kh = reinterpret_cast<const LmKind *>(KH + kLmKindWords * kind);
        for (int d = 0; d < kh->nd; ++d) {
            const int e = KT[kh->off_dof + 4 * d];
            if (e % G == l) dbits |= (1u << (e / G));
        }
        MB[i] = bits;
        MB[nkinds * G + i] = dbits;
    }
    const int kh_words = khh_words + ((L.hot_words + 3) & ~3) + ((ntri + 3) & ~3);
    float *CB = lds + plan_words + a.mb_words + kh_words + (wave * CPW + grp) * L.chain_stride;
    float *bx = CB + H.c_bx, *ja = CB + H.c_ja, *jn = CB + H.c_jn;
    float *sw = CB + H.c_sw, *gg = CB + H.c_gg, *r2 = CB + H.c_gg;
    float *qe = CB + H.c_qe, *kpl = CB + H.c_kp;
    float *sxs = CB + L.c_sx, *Jp = CB + L.c_jp, *Hc = CB + L.c_jp, *Ap = CB + L.c_A;
    float *bv = CB + L.c_b, *dv = CB + L.c_d, *fz = CB + L.c_fz;
    __syncthreads();

    const int *lev_adr = reinterpret_cast<const int *>(P + H.off_lev_adr);
    const float *brec = P + H.off_body, *jrec = P + H.off_joint, *srec = P + H.off_site;
    const float *lbv = P + H.off_lb, *ubv = P + H.off_ub, *qpos0 = P + H.off_qpos0;
    const int *quat_adr = reinterpret_cast<const int *>(P + H.off_quat_adr);

    int chain = (blockIdx.x * wpb + wave) * CPW + grp;  // with a chain queue (QArgs::queue_slots): the first of several
    int st = chain < a.C ? LM_EVAL_X : LM_DONE;
    int kind = a.do_root_opt ? 0 : 2;
    int frame = 0, iter = 0, tries = 0;
    float f = 0.0f, lam = L.lambda0, error = __builtin_inff();
    uint32_t c_iter = 0, c_ls = 0, c_grad = 0, c_solves = 0;
    float x[NQR], g[NQR], tr[NQR], q0[NQR];

    if (lg == 0) { st_tpos(bx, V3{0.f, 0.f, 0.f}); st_tquat(bx, Q4{1.f, 0.f, 0.f, 0.f}); }
    size_t kp_chain = (size_t)(chain < a.C ? chain : 0) * a.F * 3 * K;
#pragma unroll
    for (int r = 0; r < NQR; ++r) {
        const int e = r * G + lg;
        float v = 0.f;
        if (e < nq && st != LM_DONE) v = a.q_init ? a.q_init[(size_t)chain * nq + e] : qpos0[e];
        q0[r] = v;
    }
    if (st != LM_DONE) {
        for (int i = lg; i < 3 * K; i += G) kpl[i] = a.kp[kp_chain + i];
        if (kind < 2) {
#pragma unroll
            for (int r = 0; r < NQR; ++r) {
                const int e = r * G + lg;
                if (e < 3) q0[r] = a.kp[kp_chain + 3 * a.root_kp_idx + e];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < NQR; ++r) { x[r] = q0[r]; tr[r] = q0[r]; g[r] = 0.f; }
    wave_sync();

    PROF_DECL;
    // Launch arguments inside the trip loop: read from the kernarg segment again in every trip, through a pointer that is
    // opaque per trip, together with everything derived from them and from the lane index (stac_kernels.hip, q_phase_kernel:
    // "Launch arguments inside the trip loop"): nothing of QArgs / LmArgs stays live across the loop, where it used to
    // cost 103-152 scalars spilled into vector-register lanes and up to 308 B of scratch per lane.
    struct LmKernArgs { QArgs a; LmArgs L; };
    static_assert(offsetof(LmKernArgs, L) == sizeof(QArgs), "kernarg layout: LmArgs follows QArgs without padding");
    typedef const __attribute__((address_space(4))) LmKernArgs KArgs;
    KArgs *const ak_base = (KArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    const int lg_outer = lg, cb_words = (int)(CB - lds);
    while (__any(st != LM_DONE)) {
        PROF_TICK(0);
        KArgs *ak_t = ak_base;
        asm volatile("" : "+s"(ak_t));
        const auto &a = ak_t->a;
        const auto &L = ak_t->L;
        const auto &H = a.h;
        int lg_t = lg_outer, cb_t = cb_words;
        asm volatile("" : "+v"(lg_t), "+v"(cb_t));
        const int lg = lg_t;
        // (the names below shadow the prologue's)
        const int nq = H.nq, K = H.K, nqpad = H.nqpad;
        const int plan_words = (H.total_words + 3) & ~3;
        uint32_t *const MB = reinterpret_cast<uint32_t *>(lds + plan_words);
        const int nkinds = a.P + 3;
        int *const KH = reinterpret_cast<int *>(lds + plan_words + a.mb_words);
        int *const KT = KH + ((nkinds * kLmKindWords + 3) & ~3);
        int *const TRI = KT + ((L.hot_words + 3) & ~3);
        float *const CB = lds + cb_t;
        float *const bx = CB + H.c_bx, *const ja = CB + H.c_ja, *const jn = CB + H.c_jn;
        float *const sw = CB + H.c_sw, *const gg = CB + H.c_gg, *const r2 = CB + H.c_gg;
        float *const qe = CB + H.c_qe, *const kpl = CB + H.c_kp;
        float *const sxs = CB + L.c_sx, *const Jp = CB + L.c_jp, *const Hc = CB + L.c_jp, *const Ap = CB + L.c_A;
        float *const bv = CB + L.c_b, *const dv = CB + L.c_d, *const fz = CB + L.c_fz;
        const float *const jrec = P + H.off_joint, *const srec = P + H.off_site;
        const float *const lbv = P + H.off_lb, *const ubv = P + H.off_ub, *const qpos0 = P + H.off_qpos0;
        const int *const quat_adr = reinterpret_cast<const int *>(P + H.off_quat_adr);
        const int st_in = st;
        const uint32_t mbits = MB[kind * G + lg], dbits = MB[nkinds * G + kind * G + lg];
        const LmKind kh = *reinterpret_cast<const LmKind *>(KH + kLmKindWords * kind);

        // ---- stage the point: x at the start of a solve, the trial point afterwards ---------------------------
#pragma unroll
        for (int r = 0; r < NQR; ++r) {
            const int e = r * G + lg;
            if (e < nq) {
                const float pt = st_in == LM_TRIAL ? tr[r] : x[r];
                const float mi = ((mbits >> r) & 1u) ? 1.0f : 0.0f;
                qe[e] = (1.0f - mi) * q0[r] + mi * pt;
            }
        }
        wave_sync();

        // ---- forward kinematics: the PG kernel's joint-local pre-pass + FK program (stac_device.hpp) -------------------
        joint_local_prepass(H, P, CB, lg, G, H.naj);
        wave_sync();
        fk_chain<(G >= 16)>(H, P, CB, lg, G, true, true, (a.flags & 2) != 0);

        PROF_TICK(1);  // stage + FK
        // ---- sites: world position (kept for the Jacobian), residual, loss term, wrench -------------------------------
        const V3 cref = ld_tpos(bx + kXf);
        const bool trunk_w = kind < 2;
        const int Kpad = (K + 3) & ~3;
        for (int k = K + lg; k < Kpad; k += G) r2[k] = 0.0f;
        for (int k = lg; k < K; k += G) {
            const float4 sr = lds4(srec + 4 * k);
            const int ss = __builtin_bit_cast(int, sr.w);
            const float *bp = bx + (ss & 0xFFFF) * kXf;
            const V3 sx = add3(ld_tpos(bp), rotate(V3{sr.x, sr.y, sr.z}, ld_tquat(bp)));
            const float w = trunk_w ? (a.kpw[k] ? 1.f : 0.f) : 1.f;
            const float rx = (kpl[3 * k] - sx.x) * w, ry = (kpl[3 * k + 1] - sx.y) * w, rz = (kpl[3 * k + 2] - sx.z) * w;
            const V3 fv = {-2.0f * rx, -2.0f * ry, -2.0f * rz};
            const int sp = ss >> 16;
            st_tpos(sw + kXf * sp, fv);
            st_tvec2(sw + kXf * sp, cross3(sub3(sx, cref), fv));
            st3(sxs + 3 * sp, sx);
            r2[k] = FMA(rz, rz, FMA(ry, ry, rx * rx));
        }
        wave_sync();
        float loss = 0.0f;
        for (int i = 0; i < (Kpad >> 2); ++i) {
            const float4 q4 = lds4(r2 + 4 * i);
            loss += (q4.x + q4.y) + (q4.z + q4.w);
        }
        wave_sync();

        PROF_TICK(2);  // sites + loss
        // ---- gradient (joint pass); the world axis replaces the pre-joint quaternion in ja for the Jacobian --------------
        for (int e = lg; e < nqpad; e += G) gg[e] = 0.0f;
        wave_sync();
        for (int j = lg; j < H.naj; j += G) {
            const float *jr = jrec + 12 * j;
            const int4 ji = lds4i(jr);
            const int ty = ji.x, ad = ji.y;
            V3 Fs = {0.f, 0.f, 0.f}, T0 = {0.f, 0.f, 0.f};
            for (int i = ji.z; i < ji.w; ++i) {
                Fs = add3(Fs, ld_tpos(sw + kXf * i));
                T0 = add3(T0, ld_tvec2(sw + kXf * i));
            }
            const V3 anchor = ld_tpos(ja + kXf * j);
            const Q4 prequat = ld_tquat(ja + kXf * j);
            const V3 tau = sub3(T0, cross3(sub3(anchor, cref), Fs));
            if (ty == JHINGE || ty == JSLIDE) {
                const float4 ja4 = lds4(jr + 8);
                const V3 axis = rotate(V3{ja4.x, ja4.y, ja4.z}, prequat);
                st3(ja + kXf * j + kXq, axis);
                gg[ad] = ty == JHINGE ? dot3(axis, tau) : dot3(axis, Fs);
            } else {  // free / ball (the PG kernel's formulas: stac_kernels.hip, joint_gradient)
                int qa = ad;
                V3 tl = tau;
                if (ty == JFREE) {
                    st3(gg + ad, Fs);
                    qa = ad + 3;
                } else {
                    tl = rotate(tau, Q4{prequat.w, -prequat.x, -prequat.y, -prequat.z});  // in the frame the ball rotation is applied in
                }
                const Q4 qh = ld4(qe + qa);
                const V3 u = {qh.x, qh.y, qh.z};
                const V3 uxt = cross3(u, tl);
                const float n = jn[__builtin_bit_cast(int, lds4(jr + 4).w)];  // by quaternion ordinal
                const float dn = n + (n == 0.0f ? 1e-6f : 0.0f);
                gg[qa] = (-2.0f * dot3(tl, u)) / dn;
                gg[qa + 1] = (2.0f * FMA(qh.w, tl.x, -uxt.x)) / dn;
                gg[qa + 2] = (2.0f * FMA(qh.w, tl.y, -uxt.y)) / dn;
                gg[qa + 3] = (2.0f * FMA(qh.w, tl.z, -uxt.z)) / dn;
            }
        }
        wave_sync();
        float gnew[NQR];
#pragma unroll
        for (int r = 0; r < NQR; ++r) {
            const int e = r * G + lg;
            gnew[r] = (e < nq && ((mbits >> r) & 1u)) ? gg[e] : 0.0f;
        }

        PROF_TICK(3);  // gradient
        // ---- accept / reject, stopping test --------------------------------------------------------------------------------
        bool ending = false, need_system = false;
        if (st_in != LM_DONE) {
            c_grad++;
            const bool first = st_in == LM_EVAL_X;
            if (!first) c_ls++;
            if (first || loss < f) {
#pragma unroll
                for (int r = 0; r < NQR; ++r) { if (!first) x[r] = tr[r]; g[r] = gnew[r]; }
                f = loss;
                tries = 0;
                if (!first) {
                    lam = lam * 0.5f;
                    if (lam < 1e-9f) lam = 1e-9f;
                    iter++;
                }
                need_system = true;
            } else {
                lam = lam * 4.0f;
                tries++;
                if (tries >= 8) ending = true;  // no decrease found: keep x
            }
        }
        {
            float t0[NQR];
#pragma unroll
            for (int r = 0; r < NQR; ++r) {
                const int e = r * G + lg;
                float a0 = 0.0f;
                if (e < nq) {
                    const float d = clipf(x[r] - g[r], lbv[e], ubv[e]) - x[r];
                    a0 = d * d;
                }
                t0[r] = a0;
            }
            const float e2 = group_tree_sum<G, NQR>(t0);
            if (need_system) {
                error = __builtin_sqrtf(e2);
                if (!(error > a.tol) || iter >= a.maxiter || kh.nd == 0) { ending = true; need_system = false; }
            }
        }

        PROF_TICK(4);  // accept + residual
        // ---- Gauss-Newton system of the accepted point ----------------------------------------------------------------------
        const int n = kh.nd;
        if (__any(need_system)) {
            if (need_system) {
                auto geom = [&](int d) {
                    const int4 dr = *reinterpret_cast<const int4 *>(KT + kh.off_dof + 4 * d);
                    const float *jr = jrec + 12 * dr.y;
                    DofGeom g2;
                    g2.ty = reinterpret_cast<const int *>(jr)[0];
                    g2.comp = dr.z;
                    g2.anchor = ld_tpos(ja + kXf * dr.y);
                    g2.axis = ld3(ja + kXf * dr.y + kXq);
                    g2.qh = Q4{1.f, 0.f, 0.f, 0.f};
                    g2.pre = Q4{1.f, 0.f, 0.f, 0.f};
                    g2.dn = 1.0f;
                    if (g2.ty == JFREE || g2.ty == JBALL) {
                        const int ad = reinterpret_cast<const int *>(jr)[1];
                        g2.qh = ld4(qe + (g2.ty == JFREE ? ad + 3 : ad));
                        if (g2.ty == JBALL) g2.pre = ld_tquat(ja + kXf * dr.y);  // (a ball's entry keeps its pre-joint quaternion: no world axis)
                        const float nn = jn[__builtin_bit_cast(int, lds4(jr + 4).w)];  // by quaternion ordinal
                        g2.dn = nn + (nn == 0.0f ? 1e-6f : 0.0f);
                    }
                    return g2;
                };
                // weighted Jacobian blocks of every (site, dof on its root path)
                for (int it = lg; it < kh.ni; it += G) {
                    const int4 rec = *reinterpret_cast<const int4 *>(L.tab + kh.off_item + 4 * it);
                    const V3 col = jac_col(geom(rec.y), ld3(sxs + 3 * rec.x));
                    // site weight (0/1): trunk mask in the root passes, all ones otherwise
                    const float w = trunk_w ? (a.kpw_sorted[rec.x] ? 1.f : 0.f) : 1.0f;
                    st3(Jp + (rec.x * kh.maxpd + rec.z) * 3, V3{col.x * w, col.y * w, col.z * w});
                }
                for (int i = lg; i < kh.nd * kh.maxpd; i += G) Ap[i] = 0.0f;
            }
            wave_sync();
            PROF_TICK(5);  // Jacobian blocks
            if (need_system) {
                for (int en = lg; en < kh.ne; en += G) {
                    const int4 rec = *reinterpret_cast<const int4 *>(L.tab + kh.off_ent + 4 * en);
                    const int pr = rec.z & 0xFF, pc = rec.z >> 8, lo = rec.w & 0xFFFF, hi = rec.w >> 16;
                    float s = 0.0f;
                    for (int i = lo; i < hi; ++i)
                        s += dot3(ld3(Jp + (i * kh.maxpd + pr) * 3), ld3(Jp + (i * kh.maxpd + pc) * 3));
                    Ap[rec.x * kh.maxpd + pc] = s;
                }
                for (int d = lg; d < n; d += G) {
                    const int e = KT[kh.off_dof + 4 * d];
                    const float b = -0.5f * gg[e];
                    bv[d] = b;
                    const float xe = qe[e];  // the accepted point (root quaternion normalised: never at +-1 unless axis-aligned)
                    fz[d] = ((xe <= lbv[e] && b < 0.0f) || (xe >= ubv[e] && b > 0.0f)) ? 1.0f : 0.0f;
                }
            }
            wave_sync();
            if (need_system) {
                // gauge: the length of a raw quaternion (the free root's, a ball joint's) does not change the pose -- make that direction stiff
                for (int idx = lg; idx < 10 * kh.nquat; idx += G) {
                    const int qi = idx / 10;
                    int c = 0, e = idx - 10 * qi;
                    while (e > c) { e -= c + 1; ++c; }  // -> (c, e), e <= c < 4
                    // the four raw-quaternion dofs are consecutive on one root path: pd(b0 + c) = pd(b0) + c
                    const int b0 = KT[kh.off_quat + qi];
                    const int ad = KT[kh.off_dof + 4 * b0], pd0 = KT[kh.off_dof + 4 * b0 + 3];
                    Ap[(b0 + c) * kh.maxpd + pd0 + e] += qe[ad + c] * qe[ad + e];
                }
            }
            wave_sync();
        }

        PROF_TICK(6);  // J^T J entries, b, gauge
        // ---- damped, bound-aware system -> L^T D L factorisation -> step -> trial point -------------------------------------
        // J^T J of a kinematic tree is non-zero only between dofs on a common root path.  Rows hold their own
        // path (row b, column pd(a)); eliminating the dofs leaves-first (Featherstone's L^T D L) creates no
        // fill-in: dof k only touches the rows of its ancestors.
        const bool solving = (st_in != LM_DONE) && !ending;
        const int mp = kh.maxpd;
        bool bad = false;
        if (__any(solving)) {
            if (solving) {
                for (int en = lg; en < kh.ne; en += G) {
                    const int4 rec = *reinterpret_cast<const int4 *>(L.tab + kh.off_ent + 4 * en);
                    const int pc = rec.z >> 8;
                    float v = Ap[rec.x * mp + pc];
                    const bool fr = fz[rec.x] != 0.0f, fc = fz[rec.y] != 0.0f;
                    if (rec.x == rec.y) v = fr ? 1.0f : FMA(v, lam, v) + 1e-9f;
                    else if (fr || fc) v = 0.0f;
                    Hc[rec.x * mp + pc] = v;
                }
                for (int i = lg; i < n; i += G) dv[i] = fz[i] != 0.0f ? 0.0f : bv[i];
            }
            wave_sync();
            PROF_TICK(7);  // damped matrix
            if (solving) {
                const int *anc = KT + kh.off_anc;
                // Per pivot k (leaves first): (A) for p < pd(k): l_p = H[k][p] / H[k][k] into a scratch row (the pivot
                // row itself is still read by (B)), y = L^-T b on the fly (b[anc_p] -= l_p b[k]), and the previous
                // pivot's scratch row goes to its final place L[k+1][.]; (B) pairs (p_i >= p_j):
                // H[anc_i][p_j] -= l_i H[k][p_j].  Two wave-level syncs and pd(k) divisions per pivot.
                float *lrow = dv + L.n_max;  // two scratch rows of maxpd floats, used alternately
                int kprev = -1, pdprev = 0;
                // (what does not depend on the numbers -- the next pivot's path depth, this lane's first ancestor and pair of either phase --
                //  is read a phase ahead: the pivots are serial and every dependent LDS round trip of a pivot counts)
                int pdn = n > 0 ? KT[kh.off_dof + 4 * (n - 1) + 3] : 0;
                for (int k = n - 1; k >= 0; --k) {
                    const int pdk = pdn;
                    pdn = k > 0 ? KT[kh.off_dof + 4 * (k - 1) + 3] : 0;
                    const float *Hk = Hc + k * mp;
                    const float dkk = Hk[pdk];
                    const int ai0 = anc[k * mp + min(lg, mp - 1)];  // (phase A's first ancestor of this lane)
                    const int pij0 = TRI[min(lg, ((L.maxpd * (L.maxpd + 1)) >> 1) - 1)];  // (phase B's first pair)
                    if (!(dkk > 0.0f)) bad = true;
                    float *lcur = lrow + (k & 1) * mp;
                    const float *lold = lrow + ((k + 1) & 1) * mp;
                    for (int p = lg; p < pdprev; p += G) Hc[kprev * mp + p] = lold[p];
                    const float bk = dv[k];
                    for (int p = lg; p < pdk; p += G) {
                        const float l = Hk[p] / dkk;
                        lcur[p] = l;
                        const int ai = p == lg ? ai0 : anc[k * mp + p];
                        dv[ai] = FMA(-l, bk, dv[ai]);
                    }
                    const int pi0 = pij0 >> 8;
                    const int aj0 = anc[k * mp + min(max(pi0, 0), mp - 1)];  // (phase B's first ancestor: before the fence)
                    wave_sync();
                    // ((B) dividing H[k][p_i] by the pivot itself instead of waiting for (A)'s quotients -- one fence per pivot -- measured
                    //  slower: 1.92 against 1.99 M frames/s)
                    const int T = (pdk * (pdk + 1)) >> 1;
                    for (int q = lg; q < T; q += G) {
                        const int pij = q == lg ? pij0 : TRI[q], pi = pij >> 8, pj = pij & 0xFF;
                        const int ai = q == lg ? aj0 : anc[k * mp + pi];
                        Hc[ai * mp + pj] = FMA(-lcur[pi], Hk[pj], Hc[ai * mp + pj]);
                    }
                    wave_sync();
                    kprev = k; pdprev = pdk;
                }
                if (kprev >= 0) {
                    const float *lold = lrow + (kprev & 1) * mp;
                    for (int p = lg; p < pdprev; p += G) Hc[kprev * mp + p] = lold[p];
                }
                wave_sync();
                PROF_TICK(8);  // L^T D L + backward pass
                // z = D^-1 y, then d = L^-1 z.  A dof only depends on the dofs of its own root path: every lane
                // keeps the running value of its dofs in registers; depth by depth the finished dofs are published
                // and the deeper ones subtract L[i][depth] * d[ancestor at that depth].
                constexpr int NDL = (192 + G - 1) / G;
                float tv[NDL];
                int pdv[NDL];
#pragma unroll
                for (int r = 0; r < NDL; ++r) {
                    const int i = r * G + lg;
                    pdv[r] = i < n ? KT[kh.off_dof + 4 * i + 3] : -1;
                    tv[r] = i < n ? dv[i] / Hc[i * mp + pdv[r]] : 0.0f;
                }
                wave_sync();
                for (int lev = 0; lev < mp; ++lev) {
#pragma unroll
                    for (int r = 0; r < NDL; ++r)
                        if (pdv[r] == lev) dv[r * G + lg] = tv[r];
                    wave_sync();
#pragma unroll
                    for (int r = 0; r < NDL; ++r) {
                        const int i = r * G + lg;
                        if (pdv[r] > lev) tv[r] = FMA(-Hc[i * mp + lev], dv[anc[i * mp + lev]], tv[r]);
                    }
                    wave_sync();
                }
            }
            wave_sync();
            PROF_TICK(9);  // forward pass
            if (solving) {
                if (bad) {  // not positive definite at this damping: treat as a rejected step
                    lam = lam * 4.0f;
#pragma unroll
                    for (int r = 0; r < NQR; ++r) tr[r] = x[r];
                } else {
                    // scatter the step to qpos order through gg (consumed), then the clipped trial point
                    for (int e = lg; e < nqpad; e += G) gg[e] = 0.0f;
                }
            }
            wave_sync();
            if (solving && !bad)
                for (int d = lg; d < n; d += G) gg[KT[kh.off_dof + 4 * d]] = dv[d];
            wave_sync();
            if (solving && !bad) {
#pragma unroll
                for (int r = 0; r < NQR; ++r) {
                    const int e = r * G + lg;
                    tr[r] = (e < nq && ((dbits >> r) & 1u)) ? clipf(x[r] + gg[e], lbv[e], ubv[e]) : x[r];
                }
            }
            if (solving) st = LM_TRIAL;
            wave_sync();
        }

        PROF_TICK(10);  // trial point
        // ---- end of a solve: replace_qs, next solve / next frame (same sequencing as the PG kernel) ----------------------------
        if (__any(ending)) {
            if (ending) {
#pragma unroll
                for (int r = 0; r < NQR; ++r) {
                    const int e = r * G + lg;
                    if (e < nq) {
                        const float mi = ((mbits >> r) & 1u) ? 1.0f : 0.0f;
                        qe[e] = (kind != 2) ? ((1.0f - mi) * q0[r] + mi * x[r]) : x[r];
                    }
                }
            }
            wave_sync();
            if (ending) {
                for (int qi = lg; qi < H.nquat; qi += G) {
                    const int ad = quat_adr[qi];
                    float nn;
                    st4(qe + ad, normalize4(ld4(qe + ad), &nn));
                }
            }
            wave_sync();
            if (ending) {
                c_iter += iter;
                c_solves++;
#pragma unroll
                for (int r = 0; r < NQR; ++r) {
                    const int e = r * G + lg;
                    if (e < nq) q0[r] = qe[e];
                }
                if (kind < 2) c_iter = c_ls = c_grad = c_solves = 0;
                kind++;
                if (kind > a.P + 2) {
                    const size_t fo = (size_t)chain * a.F + frame;
                    for (int e = lg; e < nq; e += G) a.qpos_out[fo * nq + e] = qe[e];
                    if (lg == 0) {
                        a.err_out[fo] = error;
                        if (a.counters_out) {
                            uint32_t *co = a.counters_out + fo * 4;
                            co[0] = c_iter; co[1] = c_ls; co[2] = c_grad; co[3] = c_solves;
                        }
                    }
                    c_iter = c_ls = c_grad = c_solves = 0;
                    frame++;
                    kind = 2;
                    if (frame >= a.F) {
                        if (a.q_carry_out)
                            for (int e = lg; e < nq; e += G) a.q_carry_out[(size_t)chain * nq + e] = qe[e];
                        st = LM_DONE;
                        if (a.queue_slots > 0) {  // chain queue: take the next unstarted chain (ctl[4])
                            int nxt = 0;
                            if (lg == 0) nxt = atomicAdd(a.ctl + 4, 1);
                            nxt = __shfl(nxt, grp * G, 64);
                            if (nxt < a.C) {
                                chain = nxt;
                                kp_chain = (size_t)nxt * a.F * 3 * K;
                                kind = a.do_root_opt ? 0 : 2;
                                frame = 0;
#pragma unroll
                                for (int r = 0; r < NQR; ++r) {
                                    const int e = r * G + lg;
                                    float v = 0.f;
                                    if (e < nq) v = a.q_init ? a.q_init[(size_t)nxt * nq + e] : qpos0[e];
                                    if (kind < 2 && e < 3) v = a.kp[kp_chain + 3 * a.root_kp_idx + e];
                                    q0[r] = v;
                                    g[r] = 0.f;
                                }
                                for (int i = lg; i < 3 * K; i += G) kpl[i] = a.kp[kp_chain + i];
                                st = LM_EVAL_X;  // x, tr and the solver scalars are reset just below
                            }
                        }
                    } else {
                        for (int i = lg; i < 3 * K; i += G) kpl[i] = a.kp[kp_chain + (size_t)frame * 3 * K + i];
                    }
                } else if (kind < 2) {
#pragma unroll
                    for (int r = 0; r < NQR; ++r) {
                        const int e = r * G + lg;
                        if (e < 3) q0[r] = kpl[3 * a.root_kp_idx + e];
                    }
                }
                if (st != LM_DONE) {
#pragma unroll
                    for (int r = 0; r < NQR; ++r) { x[r] = q0[r]; tr[r] = q0[r]; }
                    lam = L.lambda0; iter = 0; tries = 0;
                    error = __builtin_inff();
                    st = LM_EVAL_X;
                }
            }
            wave_sync();
        }
        PROF_TICK(11);  // end of solve
        PROF_LM_END;
    }
    PROF_FLUSH(a);
}

template <int G, int NQR, int WPE>
static hipError_t launch_lm(const QArgs &a, const LmArgs &L, int wpb, size_t lds_bytes, hipStream_t s) {
    constexpr int CPW = 64 / G;
    const int per_block = CPW * wpb;
    const int slots = a.queue_slots > 0 ? (a.queue_slots < a.C ? a.queue_slots : a.C) : a.C;  // chain queue: resident slots only
    const int blocks = (slots + per_block - 1) / per_block;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&q_phase_lm_kernel<G, NQR, WPE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((q_phase_lm_kernel<G, NQR, WPE>), dim3(blocks), dim3(64 * wpb), lds_bytes, s, a, L);
    return hipGetLastError();
}

// Register caps: the 64-lane instantiations fit 168 VGPRs (3 waves per SIMD), the narrower ones need 2 per SIMD.
hipError_t launch_q_phase_lm(const QArgs &a, const LmArgs &L, int G, int wpb, size_t lds_bytes, hipStream_t s,
                             int *capacity_out) {
    const int nq = a.h.nq;
    *capacity_out = 0;
#define STAC_TRY(GG, RR, WW)                                    \
    if (G == GG && nq <= GG * RR) {                             \
        *capacity_out = GG * RR;                                \
        return launch_lm<GG, RR, WW>(a, L, wpb, lds_bytes, s);  \
    }
    STAC_TRY(16, 5, 2) STAC_TRY(16, 8, 2) STAC_TRY(16, 16, 2)
    STAC_TRY(32, 3, 2) STAC_TRY(32, 8, 2)
    STAC_TRY(64, 2, 3) STAC_TRY(64, 4, 2)
#undef STAC_TRY
    return hipErrorInvalidValue;
}

int lm_waves_per_simd(int G, int nq) { return (G == 64 && nq <= 128) ? 3 : 2; }

}  // namespace stac
