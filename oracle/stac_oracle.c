/*
 * stac_oracle.c -- CPU restatement of the STAC hot path (see stac_oracle.h).
 * TEST INFRASTRUCTURE ONLY: never linked into or called from the product path.
 * Parity status: FK, bounds and m_opt pinned exactly to the reference's stored outputs / known-answer tests;
 * the q_phase is pinned to the reference's stored fit demos/demo_viz.p in marker space, at its stopping rule and
 * as a fixed point (tests/test_pin_demo_viz.py).  The iterate-level trajectory of jaxopt's ProjectedGradient is
 * "PARITY UNPINNED": no jaxopt here, and the one printed reference run (demos/rodent_demo.ipynb) came from older
 * reference source and is reproduced in distribution only (tests/tools/jaxopt_variant_sweep.py).
 *
 * Build: oracle/Makefile (gcc -O3 -mfma -ffp-contract=off -fopenmp).  Compiling with
 * -DORC_REAL=double gives a float64 twin used by tests for finite-difference and
 * rounding-sensitivity checks (same symbols, separate .so).
 */
#include "stac_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef ORC_REAL
#define ORC_REAL float
#endif
typedef ORC_REAL real;

/* The public ABI is float32; the float64 twin converts at the boundary. */
#define R(x) ((real)(x))

static inline real rsqrt_(real x) { return sizeof(real) == 4 ? (real)sqrtf((float)x) : (real)sqrt((double)x); }
/* Fused multiply-add: one rounding, exactly what v_fma_f32 does on the GPU (build with -mfma so gcc
 * emits the hardware instruction; glibc's fmaf is exact too, only slower). */
static inline real FMA(real a, real b, real c) { return sizeof(real) == 4 ? (real)fmaf((float)a, (float)b, (float)c) : (real)fma((double)a, (double)b, (double)c); }

/* sin/cos of the hinge half-angle.  The float32 build uses its own Cody-Waite reduction + minimax
 * polynomial (cephes sinf/cosf coefficients) written as an explicit sequence of mul/fma, so that the
 * HIP kernels -- which contain the same operation sequence -- reproduce it bit for bit (libm's and
 * OCML's sinf differ in the last ulp).  Max error ~1.5 ulp for |a| < 100. */
static inline void sincos_(real a, real *sn, real *cs) {
    if (sizeof(real) != 4) { *sn = (real)sin((double)a); *cs = (real)cos((double)a); return; }
    const float x = (float)a;
    const float k = rintf(x * 0.636619772367581343f);            /* x * 2/pi, round-half-even */
    float r = fmaf(-k, 1.5703125f, x);                            /* pi/2 split in three */
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.54978995489188216e-8f, r);
    const float z = r * r;
    const float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    const float s0 = fmaf(r * z, ps, r);
    const float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    const float c0 = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
    const int q = ((int)k) & 3;
    const float ss = (q & 1) ? c0 : s0, cc = (q & 1) ? s0 : c0;
    *sn = (real)((q & 2) ? -ss : ss);
    *cs = (real)(((q + 1) & 2) ? -cc : cc);
}

/* ---- mujoco.mjx._src.math restated, with explicit fma ------------------------ */
static inline real dot3(const real *a, const real *b) { return FMA(a[2], b[2], FMA(a[1], b[1], a[0] * b[0])); }
static inline void cross3(const real *a, const real *b, real *r) {
    const real r0 = FMA(a[1], b[2], -(a[2] * b[1]));
    const real r1 = FMA(a[2], b[0], -(a[0] * b[2]));
    const real r2 = FMA(a[0], b[1], -(a[1] * b[0]));
    r[0] = r0; r[1] = r1; r[2] = r2;
}
/* rotate(vec, quat): r = 2(u.v)u + (s^2 - u.u)v + 2 s (u x v) */
static inline void rotate(const real *v, const real *q, real *r) {
    const real s = q[0];
    const real *u = q + 1;
    real c[3];
    const real uv = dot3(u, v), uu = dot3(u, u);
    cross3(u, v, c);
    const real k = FMA(s, s, -uu), t = uv + uv, s2 = s + s;
    const real r0 = FMA(s2, c[0], FMA(k, v[0], t * u[0]));
    const real r1 = FMA(s2, c[1], FMA(k, v[1], t * u[1]));
    const real r2 = FMA(s2, c[2], FMA(k, v[2], t * u[2]));
    r[0] = r0; r[1] = r1; r[2] = r2;
}
static inline void qmul(const real *u, const real *v, real *r) {
    real t[4];
    t[0] = FMA(-u[3], v[3], FMA(-u[2], v[2], FMA(-u[1], v[1], u[0] * v[0])));
    t[1] = FMA(-u[3], v[2], FMA(u[2], v[3], FMA(u[1], v[0], u[0] * v[1])));
    t[2] = FMA(u[3], v[1], FMA(u[2], v[0], FMA(-u[1], v[3], u[0] * v[2])));
    t[3] = FMA(u[3], v[0], FMA(-u[2], v[1], FMA(u[1], v[2], u[0] * v[3])));
    r[0] = t[0]; r[1] = t[1]; r[2] = t[2]; r[3] = t[3];
}
/* normalize(x) = x / (|x| + 1e-6 [|x| == 0]); returns |x| */
static inline real normalize4(real *q) {
    real n = rsqrt_(FMA(q[3], q[3], FMA(q[2], q[2], FMA(q[1], q[1], q[0] * q[0]))));
    real d = n + (n == R(0) ? R(1e-6) : R(0));
    for (int i = 0; i < 4; ++i) q[i] = q[i] / d;
    return n;
}
static inline void quat_to_mat(const real *q, real *m /* row-major 3x3 */) {
    const real q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
    const real q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
    const real q12 = q[1] * q[2], q13 = q[1] * q[3], q23 = q[2] * q[3];
    m[0] = q00 + q11 - q22 - q33; m[1] = R(2) * (q12 - q03);      m[2] = R(2) * (q13 + q02);
    m[3] = R(2) * (q12 + q03);    m[4] = q00 - q11 + q22 - q33;   m[5] = R(2) * (q23 - q01);
    m[6] = R(2) * (q13 - q02);    m[7] = R(2) * (q23 + q01);      m[8] = q00 - q11 - q22 + q33;
}

/* Pairwise (binary-tree) sum of n terms padded with zeros to a power of two: level h adds element
 * i+h into element i for i = 0, 2h, 4h, ...  XLA's reduction order is unspecified; this is the order
 * the HIP kernels use (lane butterflies, then registers) and it does not depend on how many lanes
 * share a chain.  `buf` must hold next_pow2(n) elements; it is clobbered. */
static real tree_sum(real *buf, int n) {
    int P = 1;
    while (P < n) P <<= 1;
    for (int i = n; i < P; ++i) buf[i] = R(0);
    for (int h = 1; h < P; h <<= 1)
        for (int i = 0; i < P; i += 2 * h) buf[i] = buf[i] + buf[i + h];
    return buf[0];
}

/* ---- workspace ------------------------------------------------------------------------- */
typedef struct {
    real *mp;                               /* model tables converted to `real` */
    real *body_pos, *body_quat, *jnt_pos, *jnt_axis, *qpos0, *site_pos;
    real *qf, *xpos, *xquat, *xanchor, *xaxis, *jprequat, *jnorm, *sx, *F, *Tq;
    real *x, *y, *g, *cand, *xn, *gn, *tmp, *kp, *q0, *lb, *ub;
    real *tb0, *tb1;  /* tree_sum scratch (power-of-two padded) */
    real *sf, *st;    /* per-site force / moment */
    int *sord;        /* site ids sorted by (body id, site id): subtrees are contiguous ranges */
    int *blo, *bhi;   /* per body: range [lo, hi) of sorted-site positions inside its subtree */
    int ref_body; /* moments of the gradient wrenches are taken about xpos[ref_body] */
} ws_t;

static ws_t *ws_new(const orc_model *m) {
    ws_t *w = (ws_t *)calloc(1, sizeof(ws_t));
    const int nb = m->nbody, nj = m->njnt, nq = m->nq, K = m->nsite;
    size_t n_model = (size_t)nb * 7 + (size_t)nj * 6 + nq + (size_t)K * 3;
    size_t n_fk = (size_t)nq + nb * 7 + nj * 6 + nj * 4 + nj + K * 3 + nb * 6;
    int Pq = 1, Pk = 1;
    while (Pq < nq) Pq <<= 1;
    while (Pk < K) Pk <<= 1;
    const int Pm = Pq > Pk ? Pq : Pk;
    size_t n_pg = (size_t)nq * 10 + K * 3 + 2 * (size_t)Pm + 6 * (size_t)K;
    real *p = (real *)calloc(n_model + n_fk + n_pg + 64, sizeof(real));
    w->mp = p;
    w->body_pos = p; p += nb * 3;
    w->body_quat = p; p += nb * 4;
    w->jnt_pos = p; p += nj * 3;
    w->jnt_axis = p; p += nj * 3;
    w->qpos0 = p; p += nq;
    w->site_pos = p; p += K * 3;
    for (int i = 0; i < nb * 3; ++i) w->body_pos[i] = R(m->body_pos[i]);
    for (int i = 0; i < nb * 4; ++i) w->body_quat[i] = R(m->body_quat[i]);
    for (int i = 0; i < nj * 3; ++i) w->jnt_pos[i] = R(m->jnt_pos[i]);
    for (int i = 0; i < nj * 3; ++i) w->jnt_axis[i] = R(m->jnt_axis[i]);
    for (int i = 0; i < nq; ++i) w->qpos0[i] = R(m->qpos0[i]);
    for (int i = 0; i < K * 3; ++i) w->site_pos[i] = R(m->site_pos[i]);
    w->qf = p; p += nq;
    w->xpos = p; p += nb * 3;
    w->xquat = p; p += nb * 4;
    w->xanchor = p; p += nj * 3;
    w->xaxis = p; p += nj * 3;
    w->jprequat = p; p += nj * 4;
    w->jnorm = p; p += nj;
    w->sx = p; p += K * 3;
    w->F = p; p += nb * 3;
    w->Tq = p; p += nb * 3;
    w->x = p; p += nq; w->y = p; p += nq; w->g = p; p += nq; w->cand = p; p += nq;
    w->xn = p; p += nq; w->gn = p; p += nq; w->tmp = p; p += nq; w->q0 = p; p += nq;
    w->lb = p; p += nq; w->ub = p; p += nq;
    w->kp = p; p += K * 3;
    w->tb0 = p; p += Pm; w->tb1 = p; p += Pm;
    w->sf = p; p += 3 * K; w->st = p; p += 3 * K;
    /* sites sorted by (body, id); bodies are in DFS pre-order, so the sites of a subtree are contiguous */
    {
        w->sord = (int *)calloc((size_t)K + 2 * (size_t)nb + 2, sizeof(int));
        w->blo = w->sord + K; w->bhi = w->blo + nb;
        int *send = (int *)calloc(nb, sizeof(int)); /* last body id of each subtree */
        for (int b = 0; b < nb; ++b) send[b] = b;
        for (int b = nb - 1; b >= 1; --b) { const int pa = m->body_parentid[b]; if (send[b] > send[pa]) send[pa] = send[b]; }
        int n = 0;
        for (int b = 0; b < nb; ++b)
            for (int k = 0; k < K; ++k)
                if (m->site_bodyid[k] == b) w->sord[n++] = k;
        for (int b = 0; b < nb; ++b) {
            int lo = 0, hi = 0;
            while (lo < K && m->site_bodyid[w->sord[lo]] < b) ++lo;
            hi = lo;
            while (hi < K && m->site_bodyid[w->sord[hi]] <= send[b]) ++hi;
            w->blo[b] = lo; w->bhi[b] = hi;
        }
        free(send);
    }
    /* ref_body = first body, ordered by (depth, id), that is an ancestor-or-self of a fit site
     * (the root body for every model of the reference). */
    {
        int *depth = (int *)calloc(nb, sizeof(int));
        unsigned char *act = (unsigned char *)calloc(nb, 1);
        for (int b = 1; b < nb; ++b) depth[b] = depth[m->body_parentid[b]] + 1;
        for (int k = 0; k < K; ++k)
            for (int b = m->site_bodyid[k]; b > 0 && !act[b]; b = m->body_parentid[b]) act[b] = 1;
        w->ref_body = nb > 1 ? 1 : 0;
        int best = 1 << 30;
        for (int b = 1; b < nb; ++b)
            if (act[b] && depth[b] < best) { best = depth[b]; w->ref_body = b; }
        free(depth); free(act);
    }
    return w;
}
static void ws_free(ws_t *w) {
    if (w) { free(w->mp); free(w->sord); free(w); }
}

/* ---- forward kinematics (mjx smooth.kinematics) ---------------------------- */
static void fk_ws(const orc_model *m, ws_t *w, real *qpos) {
    const int nb = m->nbody;
    real *xpos = w->xpos, *xquat = w->xquat;
    xpos[0] = xpos[1] = xpos[2] = R(0);
    xquat[0] = R(1); xquat[1] = xquat[2] = xquat[3] = R(0);
    for (int b = 1; b < nb; ++b) {
        const int p = m->body_parentid[b];
        real pos[3], quat[4], r[3];
        /* pos = xpos[p] + rotate(body_pos, xquat[p]); quat = xquat[p] * body_quat */
        rotate(w->body_pos + 3 * b, xquat + 4 * p, r);
        for (int i = 0; i < 3; ++i) pos[i] = xpos[3 * p + i] + r[i];
        qmul(xquat + 4 * p, w->body_quat + 4 * b, quat);
        const int j0 = m->body_jntadr[b], nj = m->body_jntnum[b];
        for (int j = j0; j < j0 + nj; ++j) {
            const int a = m->jnt_qposadr[j];
            real *anchor = w->xanchor + 3 * j, *axis = w->xaxis + 3 * j;
            const real *jpos = w->jnt_pos + 3 * j, *jax = w->jnt_axis + 3 * j;
            for (int i = 0; i < 4; ++i) w->jprequat[4 * j + i] = quat[i];
            w->jnorm[j] = R(1);
            switch (m->jnt_type[j]) {
            case ORC_JNT_FREE:
                for (int i = 0; i < 3; ++i) { anchor[i] = qpos[a + i]; pos[i] = qpos[a + i]; }
                axis[0] = R(0); axis[1] = R(0); axis[2] = R(1);
                w->jnorm[j] = normalize4(qpos + a + 3); /* written back, like MJX */
                for (int i = 0; i < 4; ++i) quat[i] = qpos[a + 3 + i];
                break;
            case ORC_JNT_BALL: {
                rotate(jpos, quat, r);
                for (int i = 0; i < 3; ++i) anchor[i] = r[i] + pos[i];
                rotate(jax, quat, axis);
                w->jnorm[j] = normalize4(qpos + a);
                qmul(quat, qpos + a, quat);
                rotate(jpos, quat, r);
                for (int i = 0; i < 3; ++i) pos[i] = anchor[i] - r[i];
            } break;
            case ORC_JNT_HINGE: {
                rotate(jpos, quat, r);
                for (int i = 0; i < 3; ++i) anchor[i] = r[i] + pos[i];
                rotate(jax, quat, axis);
                const real angle = qpos[a] - w->qpos0[a];
                real s, c;
                sincos_(angle * R(0.5), &s, &c);
                real qloc[4] = {c, jax[0] * s, jax[1] * s, jax[2] * s};
                qmul(quat, qloc, quat);
                rotate(jpos, quat, r);
                for (int i = 0; i < 3; ++i) pos[i] = anchor[i] - r[i];
            } break;
            case ORC_JNT_SLIDE: {
                rotate(jpos, quat, r);
                for (int i = 0; i < 3; ++i) anchor[i] = r[i] + pos[i];
                rotate(jax, quat, axis);
                const real d = qpos[a] - w->qpos0[a];
                for (int i = 0; i < 3; ++i) pos[i] = FMA(axis[i], d, pos[i]);
            } break;
            default: break;
            }
        }
        for (int i = 0; i < 3; ++i) xpos[3 * b + i] = pos[i];
        for (int i = 0; i < 4; ++i) xquat[4 * b + i] = quat[i];
    }
    for (int k = 0; k < m->nsite; ++k) {
        const int b = m->site_bodyid[k];
        real r[3];
        rotate(w->site_pos + 3 * k, xquat + 4 * b, r);
        for (int i = 0; i < 3; ++i) w->sx[3 * k + i] = xpos[3 * b + i] + r[i];
    }
}

void orc_fk(const orc_model *m, float *qpos, float *xpos, float *xquat, float *xanchor,
            float *xaxis, float *site_xpos) {
    ws_t *w = ws_new(m);
    for (int i = 0; i < m->nq; ++i) w->qf[i] = R(qpos[i]);
    fk_ws(m, w, w->qf);
    for (int i = 0; i < m->nq; ++i) qpos[i] = (float)w->qf[i];
    if (xpos) for (int i = 0; i < m->nbody * 3; ++i) xpos[i] = (float)w->xpos[i];
    if (xquat) for (int i = 0; i < m->nbody * 4; ++i) xquat[i] = (float)w->xquat[i];
    if (xanchor) for (int i = 0; i < m->njnt * 3; ++i) xanchor[i] = (float)w->xanchor[i];
    if (xaxis) for (int i = 0; i < m->njnt * 3; ++i) xaxis[i] = (float)w->xaxis[i];
    if (site_xpos) for (int i = 0; i < m->nsite * 3; ++i) site_xpos[i] = (float)w->sx[i];
    ws_free(w);
}

/* ---- q_loss and its analytic gradient (stac_core.py:27-63) --------------- */
static real q_loss_ws(const orc_model *m, ws_t *w, const real *q, const real *kp,
                      const uint8_t *qs_to_opt, const uint8_t *kps_to_opt, const real *initial_q,
                      real *grad) {
    const int nq = m->nq, K = m->nsite;
    /* make_qs (utils.py:129-144): (1 - mask) * q0 + mask * q */
    for (int i = 0; i < nq; ++i) {
        const real mi = qs_to_opt[i] ? R(1) : R(0);
        w->qf[i] = (R(1) - mi) * initial_q[i] + mi * q[i];
    }
    fk_ws(m, w, w->qf);
    /* residual = (kp - markers) * kps_to_opt; loss = sum(residual^2)  (stac_core.py:57-61) */
    /* summation order: per site fma(rz,rz, fma(ry,ry, rx*rx)), then the pairwise tree over the sites. */
    for (int k = 0; k < K; ++k) {
        real r[3];
        for (int i = 0; i < 3; ++i) {
            const real wi = kps_to_opt[3 * k + i] ? R(1) : R(0);
            r[i] = (kp[3 * k + i] - w->sx[3 * k + i]) * wi;
        }
        w->tb0[k] = FMA(r[2], r[2], FMA(r[1], r[1], r[0] * r[0]));
    }
    const real loss = tree_sum(w->tb0, K);
    if (!grad) return loss;

    /* dL/dx_k = -2 w (kp - x_k); moment of each site force about c = xpos[ref_body] (root body). */
    const real *c = w->xpos + 3 * w->ref_body;
    for (int k = 0; k < K; ++k) {
        real f[3], d[3];
        for (int i = 0; i < 3; ++i) {
            const real wi = kps_to_opt[3 * k + i] ? R(1) : R(0);
            f[i] = R(-2) * ((kp[3 * k + i] - w->sx[3 * k + i]) * wi);
            d[i] = w->sx[3 * k + i] - c[i];
        }
        cross3(d, f, w->st + 3 * k);
        for (int i = 0; i < 3; ++i) w->sf[3 * k + i] = f[i];
    }
    for (int i = 0; i < nq; ++i) grad[i] = R(0);
    for (int j = 0; j < m->njnt; ++j) {
        const int b = m->jnt_bodyid[j], a = m->jnt_qposadr[j];
        /* subtree wrench of the joint's body: its sites in (body id, site id) order, from zero */
        real F[3] = {R(0), R(0), R(0)}, T0[3] = {R(0), R(0), R(0)};
        for (int i = w->blo[b]; i < w->bhi[b]; ++i) {
            const int k = w->sord[i];
            for (int c3 = 0; c3 < 3; ++c3) { F[c3] += w->sf[3 * k + c3]; T0[c3] += w->st[3 * k + c3]; }
        }
        const real *anchor = w->xanchor + 3 * j, *axis = w->xaxis + 3 * j;
        real d[3], t[3], tau[3];
        for (int i = 0; i < 3; ++i) d[i] = anchor[i] - c[i];
        cross3(d, F, t);
        for (int i = 0; i < 3; ++i) tau[i] = T0[i] - t[i]; /* torque about the joint anchor */
        switch (m->jnt_type[j]) {
        case ORC_JNT_HINGE: grad[a] = dot3(axis, tau); break;
        case ORC_JNT_SLIDE: grad[a] = dot3(axis, F); break;
        case ORC_JNT_FREE:
        case ORC_JNT_BALL: {
            int qa = a;
            real tl[3];
            if (m->jnt_type[j] == ORC_JNT_FREE) {
                for (int i = 0; i < 3; ++i) grad[a + i] = F[i];
                qa = a + 3;
                for (int i = 0; i < 3; ++i) tl[i] = tau[i];
            } else {
                /* torque expressed in the frame the ball rotation is applied in */
                real qc[4] = {w->jprequat[4 * j], -w->jprequat[4 * j + 1], -w->jprequat[4 * j + 2], -w->jprequat[4 * j + 3]};
                rotate(tau, qc, tl);
            }
            const real *qh = w->qf + qa; /* normalised quaternion (s, u) */
            real uxt[3];
            cross3(qh + 1, tl, uxt);
            const real n = w->jnorm[j];
            const real dn = n + (n == R(0) ? R(1e-6) : R(0));
            grad[qa] = (R(-2) * dot3(tl, qh + 1)) / dn;
            for (int i = 0; i < 3; ++i) grad[qa + 1 + i] = (R(2) * FMA(qh[0], tl[i], -uxt[i])) / dn;
        } break;
        default: break;
        }
    }
    for (int i = 0; i < nq; ++i) grad[i] = qs_to_opt[i] ? grad[i] : R(0);
    return loss;
}

static double q_loss_d(const orc_model *m, const float *q, const float *kp, const uint8_t *qs_to_opt,
                       const uint8_t *kps_to_opt, const float *initial_q, float *grad);

float orc_q_loss(const orc_model *m, const float *q, const float *kp, const uint8_t *qs_to_opt,
                 const uint8_t *kps_to_opt, const float *initial_q, float *grad) {
    return (float)q_loss_d(m, q, kp, qs_to_opt, kps_to_opt, initial_q, grad);
}

/* same, loss returned at the build's working precision (for finite-difference tests on the f64 twin) */
double orc_q_loss_d(const orc_model *m, const float *q, const float *kp, const uint8_t *qs_to_opt,
                    const uint8_t *kps_to_opt, const float *initial_q, float *grad) {
    return q_loss_d(m, q, kp, qs_to_opt, kps_to_opt, initial_q, grad);
}

static double q_loss_d(const orc_model *m, const float *q, const float *kp, const uint8_t *qs_to_opt,
                       const uint8_t *kps_to_opt, const float *initial_q, float *grad) {
    ws_t *w = ws_new(m);
    for (int i = 0; i < m->nq; ++i) { w->x[i] = R(q[i]); w->q0[i] = R(initial_q[i]); }
    for (int i = 0; i < 3 * m->nsite; ++i) w->kp[i] = R(kp[i]);
    real l = q_loss_ws(m, w, w->x, w->kp, qs_to_opt, kps_to_opt, w->q0, grad ? w->g : NULL);
    if (grad) for (int i = 0; i < m->nq; ++i) grad[i] = (float)w->g[i];
    ws_free(w);
    return (double)l;
}

/* ---- jaxopt 0.8.5 ProjectedGradient.run ----------------------------------- */
static inline real clipr(real v, real lo, real hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void q_opt_ws(const orc_model *m, ws_t *w, const orc_pg_params *p, const uint8_t *qs_to_opt,
                     const uint8_t *kps_to_opt, orc_pg_state *st) {
    /* inputs in w->q0 (= init params AND initial_q, stac_core.py:83,90), w->kp, w->lb, w->ub;
     * result in w->x. */
    const int nq = m->nq;
    const real eps = sizeof(real) == 4 ? R(1.1920929e-7) : R(2.220446049250313e-16);
    real *x = w->x, *y = w->y, *g = w->g, *cand = w->cand, *gn = w->gn;
    const real *lb = w->lb, *ub = w->ub;
    for (int i = 0; i < nq; ++i) { x[i] = w->q0[i]; y[i] = w->q0[i]; }
    real stepsize = R(1), t = R(1), error = (real)INFINITY;
    int iter = 0, ls_evals = 0, grad_evals = 0;
    if (p->maxiter > 0) {
        do {
            /* (f_y, g_y) = value_and_grad(fun)(y) */
            const real fy = q_loss_ws(m, w, y, w->kp, qs_to_opt, kps_to_opt, w->q0, g);
            ++grad_evals;
            /* backtracking line search from the current stepsize */
            real eta = stepsize;
            for (int i = 0; i < nq; ++i) cand[i] = clipr(FMA(-eta, g[i], y[i]), lb[i], ub[i]);
            int n = 0;
            for (;;) {
                if (n >= p->maxls) break;
                const real fc = q_loss_ws(m, w, cand, w->kp, qs_to_opt, kps_to_opt, w->q0, NULL);
                ++ls_evals;
                for (int i = 0; i < nq; ++i) {
                    const real d = cand[i] - y[i];
                    w->tb0[i] = d * d;
                    w->tb1[i] = d * g[i];
                }
                const real sq = tree_sum(w->tb0, nq), vd = tree_sum(w->tb1, nq);
                const real lhs = eta * (fc - fy);
                const real rhs = eta * vd + R(0.5) * sq + eps;
                if (!(lhs > rhs)) break;
                eta = eta * R(0.5);
                for (int i = 0; i < nq; ++i) cand[i] = clipr(FMA(-eta, g[i], y[i]), lb[i], ub[i]);
                ++n;
            }
            const real next_step = (eta <= R(1e-6)) ? R(1) : eta / R(0.5);
            const real tn = R(0.5) * (R(1) + rsqrt_(R(1) + R(4) * t * t));
            const real beta = (t - R(1)) / tn;
            for (int i = 0; i < nq; ++i) {
                const real d = cand[i] - x[i];
                y[i] = FMA(beta, d, cand[i]);
                x[i] = cand[i];
            }
            /* error = || clip(x_next - grad(x_next)) - x_next ||_2 */
            (void)q_loss_ws(m, w, x, w->kp, qs_to_opt, kps_to_opt, w->q0, gn);
            ++grad_evals;
            for (int i = 0; i < nq; ++i) {
                const real d = clipr(x[i] - gn[i], lb[i], ub[i]) - x[i];
                w->tb0[i] = d * d;
            }
            error = rsqrt_(tree_sum(w->tb0, nq));
            stepsize = next_step;
            t = tn;
            ++iter;
        } while (error > p->tol && iter < p->maxiter);
    }
    if (st) {
        st->iter_num = iter;
        st->stepsize = (float)stepsize;
        st->error = (float)error;
        st->t = (float)t;
        st->ls_evals = ls_evals;
        st->grad_evals = grad_evals;
        st->loss = (float)q_loss_ws(m, w, x, w->kp, qs_to_opt, kps_to_opt, w->q0, NULL);
    }
}

static void ws_set_bounds(const orc_model *m, ws_t *w, const float *lb, const float *ub) {
    for (int i = 0; i < m->nq; ++i) { w->lb[i] = R(lb[i]); w->ub[i] = R(ub[i]); }
}

void orc_q_opt(const orc_model *m, const orc_pg_params *p, const float *kp,
               const uint8_t *qs_to_opt, const uint8_t *kps_to_opt, const float *q0,
               const float *lb, const float *ub, float *params_out, orc_pg_state *state_out) {
    ws_t *w = ws_new(m);
    ws_set_bounds(m, w, lb, ub);
    for (int i = 0; i < m->nq; ++i) w->q0[i] = R(q0[i]);
    for (int i = 0; i < 3 * m->nsite; ++i) w->kp[i] = R(kp[i]);
    q_opt_ws(m, w, p, qs_to_opt, kps_to_opt, state_out);
    for (int i = 0; i < m->nq; ++i) params_out[i] = (float)w->x[i];
    ws_free(w);
}

/* ---- offset phase (stac_core.py:102-172) -------------------------------------------------- */
void orc_m_partial(const orc_model *m, const float *keypoints, const float *q, int32_t T,
                   float *partial) {
    const int K = m->nsite, nq = m->nq;
    ws_t *w = ws_new(m);
    real *s = (real *)calloc((size_t)3 * K + 2, sizeof(real));
    real z2 = R(0);
    /* Summation order (XLA's is unspecified): per frame the K-site subtotal of |z|^2, then frames
     * accumulated in index order per component -- the order the HIP kernels reproduce exactly. */
    for (int t = 0; t < T; ++t) {
        for (int i = 0; i < nq; ++i) w->qf[i] = R(q[(size_t)t * nq + i]);
        fk_ws(m, w, w->qf);
        real z2t = R(0);
        for (int k = 0; k < K; ++k) {
            const int b = m->site_bodyid[k];
            real mat[9], z[3];
            quat_to_mat(w->xquat + 4 * b, mat);
            for (int i = 0; i < 3; ++i) z[i] = R(keypoints[(size_t)t * 3 * K + 3 * k + i]) - w->xpos[3 * b + i];
            /* s_k += R^T z */
            for (int i = 0; i < 3; ++i) s[3 * k + i] += mat[0 + i] * z[0] + mat[3 + i] * z[1] + mat[6 + i] * z[2];
            z2t += z[0] * z[0] + z[1] * z[1] + z[2] * z[2];
        }
        z2 += z2t;
    }
    for (int i = 0; i < 3 * K; ++i) partial[i] = (float)s[i];
    partial[3 * K] = (float)z2;
    partial[3 * K + 1] = (float)T;
    free(s);
    ws_free(w);
}

void orc_m_finish(int32_t K, const float *partial, const float *initial_offsets,
                  const float *is_regularized, float reg_coef, float *params_out,
                  float *error_out) {
    const real T = R(partial[3 * K + 1]), z2 = R(partial[3 * K]), lam = R(reg_coef);
    real ms = R(0), mm = R(0), reg = R(0);
    for (int i = 0; i < 3 * K; ++i) {
        const real d = R(is_regularized[i]), s = R(partial[i]), m0 = R(initial_offsets[i]);
        const real denom = T + lam * d;
        const real numer = s + lam * d * m0;
        /* no frame at all and an unregularised coordinate: keep the previous offset (the reference would divide 0 by 0) */
        const real ms_i = denom == (real)0 ? m0 : numer / denom;
        params_out[i] = (float)ms_i;
        ms += ms_i * s;
        mm += ms_i * ms_i;
        const real dr = d * (ms_i - m0);
        reg += dr * dr;
    }
    if (error_out) *error_out = (float)((z2 - R(2) * ms + T * mm) + lam * reg);
}

void orc_m_opt(const orc_model *m, const float *keypoints, const float *q, int32_t T,
               const float *initial_offsets, const float *is_regularized, float reg_coef,
               float *params_out, float *error_out) {
    float *partial = (float *)calloc((size_t)3 * m->nsite + 2, sizeof(float));
    orc_m_partial(m, keypoints, q, T, partial);
    orc_m_finish(m->nsite, partial, initial_offsets, is_regularized, reg_coef, params_out, error_out);
    free(partial);
}


/* ============================================================================================================
 * Optional fast solver: projected Levenberg-Marquardt (see stac_oracle.h).  NOT the reference's algorithm.
 * ============================================================================================================ */
typedef struct {
    int nd;          /* optimised coordinates with structural support (ancestor of a fit site), in qpos order */
    int maxpd;       /* longest root path, counted in such coordinates */
    int *dof;        /* [nd] qpos indices, increasing */
    int *dof_jnt;    /* [nd] joint of each coordinate */
    int *pd;         /* [nd] position of the coordinate on its own root path = number of earlier coordinates on it */
    int *path;       /* [nd, nq] the coordinates of its root path, itself last (path[a][pd[a]] == a) */
    real *J;         /* [K sorted positions, nd, 3] weighted Jacobian blocks d(site)/d(coordinate) */
    real *A, *H;     /* [nd, nd]; only entries (row, column on the row's root path) are used */
    real *b, *d, *l; /* [nd] right-hand side, step, the pivot's multipliers */
    real *term;      /* [K + 4] per-site loss terms */
    unsigned char *frozen;
} lm_ws_t;

static lm_ws_t *lm_new(const orc_model *m) {
    lm_ws_t *l = (lm_ws_t *)calloc(1, sizeof(lm_ws_t));
    const int nq = m->nq, K = m->nsite;
    l->dof = (int *)calloc(3 * (size_t)nq + (size_t)nq * nq, sizeof(int));
    l->dof_jnt = l->dof + nq;
    l->pd = l->dof_jnt + nq;
    l->path = l->pd + nq;
    l->J = (real *)calloc((size_t)3 * K * nq + 2 * (size_t)nq * nq + 3 * (size_t)nq + K + 4, sizeof(real));
    l->A = l->J + (size_t)3 * K * nq;
    l->H = l->A + (size_t)nq * nq;
    l->b = l->H + (size_t)nq * nq;
    l->d = l->b + nq;
    l->l = l->d + nq;
    l->term = l->l + nq;
    l->frozen = (unsigned char *)calloc(nq, 1);
    return l;
}
static void lm_free(lm_ws_t *l) {
    if (l) { free(l->dof); free(l->J); free(l->frozen); free(l); }
}

/* One column of the 3 x nd site Jacobian at the pose of the last fk_ws (w->qf normalised, anchors, axes, jnorm filled):
 * stac_lm.hip, jac_col, operation for operation. */
/* raw quaternion (s,u), q^ = q/|q|, d = the rotated vector: dx/ds = -2 (u^ x d)/|q|, dx/du_j = 2 (s^ (e_j x d) - (e_j x u^) x d)/|q| */
static void lm_quat_col(const real *qh, real n, int c4, const real *d, real *col) {
    const real dn = n + (n == R(0) ? R(1e-6) : R(0));
    if (c4 == 0) {
        cross3(qh + 1, d, col);
        for (int t = 0; t < 3; ++t) col[t] = (R(-2) * col[t]) / dn;
    } else {
        real e[3] = {R(0), R(0), R(0)}, exd[3], exu[3], t2[3];
        e[c4 - 1] = R(1);
        cross3(e, d, exd);
        cross3(e, qh + 1, exu);
        cross3(exu, d, t2);
        for (int t = 0; t < 3; ++t) col[t] = (R(2) * (qh[0] * exd[t] - t2[t])) / dn;
    }
}
static void lm_jac_col(const orc_model *m, const ws_t *w, int j, int comp, const real *sx, real *col) {
    const real *anchor = w->xanchor + 3 * j, *axis = w->xaxis + 3 * j;
    real dvec[3];
    for (int t = 0; t < 3; ++t) dvec[t] = sx[t] - anchor[t];
    switch (m->jnt_type[j]) {
    case ORC_JNT_HINGE: cross3(axis, dvec, col); break;
    case ORC_JNT_SLIDE: col[0] = axis[0]; col[1] = axis[1]; col[2] = axis[2]; break;
    case ORC_JNT_FREE:
        if (comp < 3) { col[0] = comp == 0; col[1] = comp == 1; col[2] = comp == 2; }
        else lm_quat_col(w->qf + m->jnt_qposadr[j] + 3, w->jnorm[j], comp - 3, dvec, col);
        break;
    case ORC_JNT_BALL: { /* in the frame the ball rotation is applied in, and back (as the gradient's torque) */
        const real *pre = w->jprequat + 4 * j;
        const real qc[4] = {pre[0], -pre[1], -pre[2], -pre[3]};
        real dl[3], cl[3];
        rotate(dvec, qc, dl);
        lm_quat_col(w->qf + m->jnt_qposadr[j], w->jnorm[j], comp, dl, cl);
        rotate(cl, pre, col);
    } break;
    default: col[0] = col[1] = col[2] = R(0); break;
    }
}

/* The LM kernel's loss of the pose of the last fk_ws: the per-site terms in site order, summed four at a time (zeros behind the last). */
static real lm_loss(const orc_model *m, const ws_t *w, lm_ws_t *l, const uint8_t *kps_to_opt) {
    const int K = m->nsite, Kpad = (K + 3) & ~3;
    for (int k = 0; k < K; ++k) {
        real r[3];
        for (int i = 0; i < 3; ++i) r[i] = (w->kp[3 * k + i] - w->sx[3 * k + i]) * (kps_to_opt[3 * k + i] ? R(1) : R(0));
        l->term[k] = FMA(r[2], r[2], FMA(r[1], r[1], r[0] * r[0]));
    }
    for (int k = K; k < Kpad; ++k) l->term[k] = R(0);
    real loss = R(0);
    for (int i = 0; i < Kpad; i += 4) loss += (l->term[i] + l->term[i + 1]) + (l->term[i + 2] + l->term[i + 3]);
    return loss;
}

/* stac_lm.hip::q_phase_lm_kernel, one solve, operation for operation: the same evaluation (FK, gradient), the kernel's loss sum, the
 * Gauss-Newton entries over DFS-ordered site ranges, the damping, the L^T D L factorisation on root paths (leaves first, no fill-in), the
 * substitutions, the clipped step and the accept / reject turns -- including what the kernel does when the damped matrix is not
 * positive definite (the step is the point itself: one more evaluation, one more rejection). */
static void q_opt_lm_ws(const orc_model *m, ws_t *w, lm_ws_t *l, const orc_lm_params *p, const uint8_t *qs_to_opt,
                        const uint8_t *kps_to_opt, orc_pg_state *st) {
    /* inputs in w->q0 (start AND initial_q), w->kp, w->lb, w->ub; result in w->x */
    const int nq = m->nq, K = m->nsite;
    real *x = w->x, *xt = w->cand, *g = w->g;
    for (int i = 0; i < nq; ++i) x[i] = w->q0[i];
    /* coordinates: optimised AND on the root path of some fit site, in qpos order; their root paths */
    l->nd = 0;
    for (int j = 0; j < m->njnt; ++j) {
        const int b = m->jnt_bodyid[j], a = m->jnt_qposadr[j];
        const int dims = m->jnt_type[j] == ORC_JNT_FREE ? 7 : (m->jnt_type[j] == ORC_JNT_BALL ? 4 : 1);
        if (w->bhi[b] <= w->blo[b]) continue;
        for (int c = 0; c < dims; ++c)
            if (qs_to_opt[a + c]) { l->dof[l->nd] = a + c; l->dof_jnt[l->nd] = j; l->nd++; }
    }
    const int nd = l->nd;
    l->maxpd = 1;
    for (int b2 = 0; b2 < nd; ++b2) {
        const int bb = m->jnt_bodyid[l->dof_jnt[b2]];
        int np = 0;
        for (int a2 = 0; a2 <= b2; ++a2) {
            int on_path = a2 == b2;
            for (int s2 = bb; s2 > 0 && !on_path; s2 = m->body_parentid[s2]) on_path = s2 == m->jnt_bodyid[l->dof_jnt[a2]];
            if (on_path) l->path[(size_t)b2 * nq + np++] = a2;
        }
        l->pd[b2] = np - 1;
        if (np > l->maxpd) l->maxpd = np;
    }
    real lam = R(p->lambda0);
    int iter = 0, evals = 0, gevals = 0, tries = 0;
    (void)q_loss_ws(m, w, x, w->kp, qs_to_opt, kps_to_opt, w->q0, g);
    real f = lm_loss(m, w, l, kps_to_opt);
    ++gevals;
    real error = (real)INFINITY;
    for (;;) {
        /* stopping residual of the accepted point, same definition as the PG solver */
        for (int i = 0; i < nq; ++i) { const real dd = clipr(x[i] - g[i], w->lb[i], w->ub[i]) - x[i]; w->tb0[i] = dd * dd; }
        error = rsqrt_(tree_sum(w->tb0, nq));
        if (!(error > R(p->tol)) || iter >= p->maxiter || nd == 0) break;
        /* Gauss-Newton system at x (fk_ws state is that of x: the last evaluation was the accepted point) */
        for (int b2 = 0; b2 < nd; ++b2) {
            const int j = l->dof_jnt[b2], bb = m->jnt_bodyid[j];
            for (int i = w->blo[bb]; i < w->bhi[bb]; ++i) { /* sites of the joint's body subtree, by sorted position */
                const int k = w->sord[i];
                real col[3];
                lm_jac_col(m, w, j, l->dof[b2] - m->jnt_qposadr[j], w->sx + 3 * k, col);
                const real wk = kps_to_opt[3 * k] ? R(1) : R(0); /* site weight (0 / 1): the trunk mask in the root passes */
                for (int t = 0; t < 3; ++t) l->J[((size_t)i * nd + b2) * 3 + t] = col[t] * wk;
            }
        }
        for (int b2 = 0; b2 < nd; ++b2) {
            const int bb = m->jnt_bodyid[l->dof_jnt[b2]];
            for (int pi = 0; pi <= l->pd[b2]; ++pi) {
                const int a2 = l->path[(size_t)b2 * nq + pi];
                real s = R(0);
                for (int i = w->blo[bb]; i < w->bhi[bb]; ++i)
                    s += dot3(l->J + ((size_t)i * nd + b2) * 3, l->J + ((size_t)i * nd + a2) * 3);
                l->A[(size_t)b2 * nd + a2] = s;
            }
            l->b[b2] = R(-0.5) * g[l->dof[b2]];
            const int e = l->dof[b2];
            const real xe = w->qf[e]; /* the accepted point as staged (root quaternion normalised) */
            l->frozen[b2] = (xe <= w->lb[e] && l->b[b2] < R(0)) || (xe >= w->ub[e] && l->b[b2] > R(0));
        }
        /* gauge of a raw quaternion (the free root's, a ball joint's): its length does not change the pose; make that direction stiff */
        for (int b2 = 0; b2 + 3 < nd; ++b2) {
            const int j = l->dof_jnt[b2], ty = m->jnt_type[j];
            const int qa = m->jnt_qposadr[j] + (ty == ORC_JNT_FREE ? 3 : 0);
            if ((ty != ORC_JNT_FREE && ty != ORC_JNT_BALL) || l->dof[b2] != qa || l->dof_jnt[b2 + 3] != j || l->dof[b2 + 3] != qa + 3)
                continue;
            const real *qh = w->qf + qa;
            for (int c = 0; c < 4; ++c)
                for (int e = 0; e <= c; ++e) l->A[(size_t)(b2 + c) * nd + b2 + e] += qh[c] * qh[e];
        }
        int accepted = 0;
        while (!accepted) {
            /* damped, bound-aware matrix on the root paths; right-hand side */
            for (int b2 = 0; b2 < nd; ++b2) {
                for (int pi = 0; pi <= l->pd[b2]; ++pi) {
                    const int a2 = l->path[(size_t)b2 * nq + pi];
                    real v = l->A[(size_t)b2 * nd + a2];
                    if (a2 == b2) v = l->frozen[b2] ? R(1) : FMA(v, lam, v) + R(1e-9);
                    else if (l->frozen[b2] || l->frozen[a2]) v = R(0);
                    l->H[(size_t)b2 * nd + a2] = v;
                }
                l->d[b2] = l->frozen[b2] ? R(0) : l->b[b2];
            }
            /* L^T D L, leaves first: pivot k scales its row, y = L^-T b on the fly, and updates the rows of its ancestors */
            int bad = 0;
            for (int k = nd - 1; k >= 0; --k) {
                const int pdk = l->pd[k];
                const int *pk = l->path + (size_t)k * nq;
                const real dkk = l->H[(size_t)k * nd + k];
                if (!(dkk > R(0))) bad = 1;
                const real bk = l->d[k];
                for (int pi = 0; pi < pdk; ++pi) {
                    const real lp = l->H[(size_t)k * nd + pk[pi]] / dkk;
                    l->l[pi] = lp;
                    l->d[pk[pi]] = FMA(-lp, bk, l->d[pk[pi]]);
                }
                for (int pi = 0; pi < pdk; ++pi)
                    for (int pj = 0; pj <= pi; ++pj) {
                        const size_t at = (size_t)pk[pi] * nd + pk[pj];
                        l->H[at] = FMA(-l->l[pi], l->H[(size_t)k * nd + pk[pj]], l->H[at]);
                    }
                for (int pi = 0; pi < pdk; ++pi) l->H[(size_t)k * nd + pk[pi]] = l->l[pi];
            }
            /* z = D^-1 y, d = L^-1 z: a coordinate only depends on the ones of its own root path (ancestors first) */
            for (int i = 0; i < nd; ++i) {
                const int *pi_ = l->path + (size_t)i * nq;
                real tv = l->d[i] / l->H[(size_t)i * nd + i];
                for (int lev = 0; lev < l->pd[i]; ++lev) tv = FMA(-l->H[(size_t)i * nd + pi_[lev]], l->d[pi_[lev]], tv);
                l->d[i] = tv;
            }
            real ft;
            if (bad) { /* not positive definite at this damping: the kernel evaluates the point itself and rejects it */
                lam = lam * R(4);
                ft = f;
            } else {
                for (int i = 0; i < nq; ++i) xt[i] = x[i];
                for (int a2 = 0; a2 < nd; ++a2) { const int i = l->dof[a2]; xt[i] = clipr(x[i] + l->d[a2], w->lb[i], w->ub[i]); }
                (void)q_loss_ws(m, w, xt, w->kp, qs_to_opt, kps_to_opt, w->q0, w->gn);
                ft = lm_loss(m, w, l, kps_to_opt);
            }
            ++evals; ++gevals;
            if (!bad && ft < f) {
                for (int i = 0; i < nq; ++i) { x[i] = xt[i]; g[i] = w->gn[i]; }
                f = ft;
                tries = 0;
                lam = lam * R(0.5);
                if (lam < R(1e-9)) lam = R(1e-9);
                ++iter;
                accepted = 1;
            } else {
                lam = lam * R(4);
                if (++tries >= 8) break; /* no decrease found: keep x */
            }
        }
        if (!accepted) {
            /* restore the fk_ws state of x for the caller and stop */
            (void)q_loss_ws(m, w, x, w->kp, qs_to_opt, kps_to_opt, w->q0, g);
            break;
        }
    }
    if (st) {
        st->iter_num = iter;
        st->stepsize = (float)lam;
        st->error = (float)error;
        st->t = R(0);
        st->ls_evals = evals;
        st->grad_evals = gevals;
        st->loss = (float)f;
    }
}

void orc_q_opt_lm(const orc_model *m, const orc_lm_params *p, const float *kp, const uint8_t *qs_to_opt,
                  const uint8_t *kps_to_opt, const float *q0, const float *lb, const float *ub,
                  float *params_out, orc_pg_state *state_out) {
    ws_t *w = ws_new(m);
    lm_ws_t *l = lm_new(m);
    ws_set_bounds(m, w, lb, ub);
    for (int i = 0; i < m->nq; ++i) w->q0[i] = R(q0[i]);
    for (int i = 0; i < 3 * m->nsite; ++i) w->kp[i] = R(kp[i]);
    q_opt_lm_ws(m, w, l, p, qs_to_opt, kps_to_opt, state_out);
    for (int i = 0; i < m->nq; ++i) params_out[i] = (float)w->x[i];
    lm_free(l);
    ws_free(w);
}

/* Self-check of the LM solver's Jacobian columns (tests/test_oracle.py): with f_k = -2 w_k (kp_k - x_k) the analytic gradient of
 * the loss is grad[a] = sum over the sites of dot3(dx_k/dq_a, f_k) for every coordinate a.  Returns the largest difference between
 * that sum over lm_jac_col's columns and q_loss_ws's gradient over all coordinates (hinge, slide, free and ball joints). */
double orc_lm_jac_check(const orc_model *m, const float *q, const float *kp) {
    ws_t *w = ws_new(m);
    const int nq = m->nq, K = m->nsite;
    uint8_t *ones = (uint8_t *)malloc((size_t)(nq > 3 * K ? nq : 3 * K));
    memset(ones, 1, (size_t)(nq > 3 * K ? nq : 3 * K));
    real *x = (real *)malloc(sizeof(real) * (size_t)nq), *g = (real *)malloc(sizeof(real) * (size_t)nq);
    for (int i = 0; i < nq; ++i) { x[i] = R(q[i]); w->q0[i] = R(q[i]); }
    for (int i = 0; i < 3 * K; ++i) w->kp[i] = R(kp[i]);
    (void)q_loss_ws(m, w, x, w->kp, ones, ones, w->q0, g);
    double worst = 0.0;
    for (int j = 0; j < m->njnt; ++j) {
        const int dims = m->jnt_type[j] == ORC_JNT_FREE ? 7 : (m->jnt_type[j] == ORC_JNT_BALL ? 4 : 1);
        for (int c = 0; c < dims; ++c) {
            double s = 0.0;
            for (int k = 0; k < K; ++k) {
                int below = 0;
                for (int b = m->site_bodyid[k]; b > 0 && !below; b = m->body_parentid[b]) below = b == m->jnt_bodyid[j];
                if (!below) continue;
                real col[3];
                lm_jac_col(m, w, j, c, w->sx + 3 * k, col);
                for (int t = 0; t < 3; ++t) s += (double)col[t] * (double)(R(-2) * (w->kp[3 * k + t] - w->sx[3 * k + t]));
            }
            const double d = fabs(s - (double)g[m->jnt_qposadr[j] + c]);
            if (d > worst) worst = d;
        }
    }
    free(x); free(g); free(ones);
    ws_free(w);
    return worst;
}

/* ---- phase drivers (compute_stac.py) -------------------------------------------------------- */
/* `lm` != NULL swaps every PG solve for the LM solver (orc_ik_clips_lm); sequencing is unchanged. */
typedef struct { const orc_lm_params *p; lm_ws_t *l; } lm_opt_t;
static void solve_ws(const orc_model *m, ws_t *w, const orc_pg_params *p, const lm_opt_t *lm, const uint8_t *qs,
                     const uint8_t *ks, orc_pg_state *st) {
    if (lm && lm->p) q_opt_lm_ws(m, w, lm->l, lm->p, qs, ks, st);
    else q_opt_ws(m, w, p, qs, ks, st);
}

/* utils.replace_qs: qpos <- make_qs(q0, mask, params); kinematics (normalises quaternions). */
static void replace_qs_ws(const orc_model *m, ws_t *w, const uint8_t *mask, real *qpos) {
    for (int i = 0; i < m->nq; ++i) {
        const real mi = mask ? (mask[i] ? R(1) : R(0)) : R(1);
        qpos[i] = mask ? (R(1) - mi) * w->q0[i] + mi * w->x[i] : w->x[i];
    }
    fk_ws(m, w, qpos);
}

static void root_opt_ws(const orc_model *m, ws_t *w, const orc_pg_params *p, const lm_opt_t *lm, const float *kp_clip,
                        int frame, int root_kp_idx, int root_dims, const uint8_t *trunk_kps,
                        real *qpos, orc_pg_state *st) {
    const int nq = m->nq, K = m->nsite;
    uint8_t *qs = (uint8_t *)calloc(nq, 1), *kps = (uint8_t *)calloc(3 * K, 1);
    for (int i = 0; i < root_dims && i < nq; ++i) qs[i] = 1;
    for (int k = 0; k < K; ++k) kps[3 * k] = kps[3 * k + 1] = kps[3 * k + 2] = trunk_kps[k];
    const float *kpf = kp_clip + (size_t)frame * 3 * K;
    for (int i = 0; i < 3 * K; ++i) w->kp[i] = R(kpf[i]);
    for (int pass = 0; pass < 2; ++pass) { /* compute_stac.py:57-98: two identical solves */
        for (int i = 0; i < nq; ++i) w->q0[i] = qpos[i];
        for (int i = 0; i < 3; ++i) w->q0[i] = R(kpf[3 * root_kp_idx + i]);
        solve_ws(m, w, p, lm, qs, kps, st);
        replace_qs_ws(m, w, qs, qpos);
    }
    free(qs); free(kps);
}

void orc_root_optimization(const orc_model *m, const orc_pg_params *p, const float *kp_clip,
                           int32_t frame, int32_t root_kp_idx, int32_t root_dims, const float *lb,
                           const float *ub, const uint8_t *trunk_kps, float *qpos,
                           orc_pg_state *last_state) {
    ws_t *w = ws_new(m);
    ws_set_bounds(m, w, lb, ub);
    real *qp = (real *)malloc(sizeof(real) * m->nq);
    for (int i = 0; i < m->nq; ++i) qp[i] = R(qpos[i]);
    root_opt_ws(m, w, p, NULL, kp_clip, frame, root_kp_idx, root_dims, trunk_kps, qp, last_state);
    for (int i = 0; i < m->nq; ++i) qpos[i] = (float)qp[i];
    free(qp);
    ws_free(w);
}

static void pose_opt_ws(const orc_model *m, ws_t *w, const orc_pg_params *p, const lm_opt_t *lm, const float *kp_clip,
                        int F, const uint8_t *part_masks, int P, real *qpos, float *qposes,
                        float *xposes, float *xquats, float *markers, float *frame_error,
                        uint32_t *counters) {
    const int nq = m->nq, K = m->nsite, nb = m->nbody;
    uint8_t *all_q = (uint8_t *)malloc(nq), *all_k = (uint8_t *)malloc(3 * K);
    memset(all_q, 1, nq);
    memset(all_k, 1, 3 * K);
    for (int f = 0; f < F; ++f) {
        orc_pg_state st;
        uint32_t cnt[4] = {0, 0, 0, 0};
        const float *kpf = kp_clip + (size_t)f * 3 * K;
        for (int i = 0; i < 3 * K; ++i) w->kp[i] = R(kpf[i]);
        /* full-body solve (compute_stac.py:217-231) */
        for (int i = 0; i < nq; ++i) w->q0[i] = qpos[i];
        solve_ws(m, w, p, lm, all_q, all_k, &st);
        cnt[0] += st.iter_num; cnt[1] += st.ls_evals; cnt[2] += st.grad_evals; cnt[3] += 1;
        replace_qs_ws(m, w, NULL, qpos);
        /* individual part solves (compute_stac.py:233-250) */
        for (int pi = 0; pi < P; ++pi) {
            const uint8_t *mask = part_masks + (size_t)pi * nq;
            for (int i = 0; i < nq; ++i) w->q0[i] = qpos[i];
            solve_ws(m, w, p, lm, mask, all_k, &st);
            cnt[0] += st.iter_num; cnt[1] += st.ls_evals; cnt[2] += st.grad_evals; cnt[3] += 1;
            replace_qs_ws(m, w, mask, qpos);
        }
        if (qposes) for (int i = 0; i < nq; ++i) qposes[(size_t)f * nq + i] = (float)qpos[i];
        if (xposes) for (int i = 0; i < nb * 3; ++i) xposes[(size_t)f * nb * 3 + i] = (float)w->xpos[i];
        if (xquats) for (int i = 0; i < nb * 4; ++i) xquats[(size_t)f * nb * 4 + i] = (float)w->xquat[i];
        if (markers) for (int i = 0; i < K * 3; ++i) markers[(size_t)f * K * 3 + i] = (float)w->sx[i];
        if (frame_error) frame_error[f] = st.error; /* PG residual of the LAST solve (compute_stac.py:252) */
        if (counters) for (int i = 0; i < 4; ++i) counters[(size_t)f * 4 + i] = cnt[i];
    }
    free(all_q); free(all_k);
}

void orc_pose_optimization(const orc_model *m, const orc_pg_params *p, const float *kp_clip,
                           int32_t F, const float *lb, const float *ub,
                           const uint8_t *part_masks, int32_t P, float *qpos, float *qposes,
                           float *xposes, float *xquats, float *markers, float *frame_error,
                           uint32_t *counters) {
    ws_t *w = ws_new(m);
    ws_set_bounds(m, w, lb, ub);
    real *qp = (real *)malloc(sizeof(real) * m->nq);
    for (int i = 0; i < m->nq; ++i) qp[i] = R(qpos[i]);
    pose_opt_ws(m, w, p, NULL, kp_clip, F, part_masks, P, qp, qposes, xposes, xquats, markers, frame_error, counters);
    for (int i = 0; i < m->nq; ++i) qpos[i] = (float)qp[i];
    free(qp);
    ws_free(w);
}

int32_t orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static void ik_clips_impl(const orc_model *m, const orc_pg_params *p, const orc_lm_params *lmp, const float *kp, int32_t C,
                  int32_t F, const float *lb, const float *ub, const uint8_t *part_masks,
                  int32_t P, const uint8_t *trunk_kps, int32_t root_kp_idx, int32_t root_dims,
                  int32_t do_root_opt, const float *q_init, float *qposes, float *xposes,
                  float *xquats, float *markers, float *frame_error, uint32_t *counters,
                  int32_t nthreads) {
    const int nq = m->nq, K = m->nsite, nb = m->nbody;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    if (nthreads > C) nthreads = C > 0 ? C : 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        /* one workspace per thread (not per clip): the CPU baseline should not time malloc */
        ws_t *w = ws_new(m);
        ws_set_bounds(m, w, lb, ub);
        real *qp = (real *)malloc(sizeof(real) * nq);
        lm_opt_t lmo = {lmp, lmp ? lm_new(m) : NULL};
        const lm_opt_t *lm = lmp ? &lmo : NULL;
#ifdef _OPENMP
#pragma omp for schedule(dynamic)
#endif
        for (int c = 0; c < C; ++c) {
            for (int i = 0; i < nq; ++i) qp[i] = q_init ? R(q_init[(size_t)c * nq + i]) : w->qpos0[i];
            const float *kpc = kp + (size_t)c * F * 3 * K;
            if (do_root_opt) root_opt_ws(m, w, p, lm, kpc, 0, root_kp_idx, root_dims, trunk_kps, qp, NULL);
            pose_opt_ws(m, w, p, lm, kpc, F, part_masks, P, qp,
                        qposes ? qposes + (size_t)c * F * nq : NULL,
                        xposes ? xposes + (size_t)c * F * nb * 3 : NULL,
                        xquats ? xquats + (size_t)c * F * nb * 4 : NULL,
                        markers ? markers + (size_t)c * F * K * 3 : NULL,
                        frame_error ? frame_error + (size_t)c * F : NULL,
                        counters ? counters + (size_t)c * F * 4 : NULL);
        }
        free(qp);
        lm_free(lmo.l);
        ws_free(w);
    }
    (void)nthreads;
}

void orc_ik_clips(const orc_model *m, const orc_pg_params *p, const float *kp, int32_t C,
                  int32_t F, const float *lb, const float *ub, const uint8_t *part_masks,
                  int32_t P, const uint8_t *trunk_kps, int32_t root_kp_idx, int32_t root_dims,
                  int32_t do_root_opt, const float *q_init, float *qposes, float *xposes,
                  float *xquats, float *markers, float *frame_error, uint32_t *counters,
                  int32_t nthreads) {
    ik_clips_impl(m, p, NULL, kp, C, F, lb, ub, part_masks, P, trunk_kps, root_kp_idx, root_dims, do_root_opt, q_init,
                  qposes, xposes, xquats, markers, frame_error, counters, nthreads);
}

void orc_ik_clips_lm(const orc_model *m, const orc_lm_params *p, const float *kp, int32_t C, int32_t F,
                     const float *lb, const float *ub, const uint8_t *part_masks, int32_t P,
                     const uint8_t *trunk_kps, int32_t root_kp_idx, int32_t root_dims, int32_t do_root_opt,
                     const float *q_init, float *qposes, float *xposes, float *xquats, float *markers,
                     float *frame_error, uint32_t *counters, int32_t nthreads) {
    ik_clips_impl(m, NULL, p, kp, C, F, lb, ub, part_masks, P, trunk_kps, root_kp_idx, root_dims, do_root_opt, q_init,
                  qposes, xposes, xquats, markers, frame_error, counters, nthreads);
}
