// stac_kernels.hip -- gfx950 (CDNA4) kernels of the STAC pose-fitting hot path.
//
// What runs here (reference = talmolab/stac-mjx):
//   q_phase_kernel   per-frame q_phase: q_loss (stac_mjx/stac_core.py:27-63) = MJX forward kinematics
//                    down the body tree -> marker site_xpos -> masked keypoint residual, its analytic
//                    gradient (replaces jax.grad), and the jaxopt ProjectedGradient iteration around it
//                    (stac_core.py:66-99,182-191), sequenced like compute_stac.root_optimization /
//                    pose_optimization (stac_mjx/compute_stac.py:17-104,170-278): one warm-started
//                    chain per clip, persistent over the clip's frames.
//   fk_kernel        utils.kinematics on N poses (stac_mjx/utils.py:49-60) for the xpos/xquat/marker
//                    outputs of pose_optimization (compute_stac.py:261-264).
//   m_* kernels      the cross-frame sums and closed form of _m_opt (stac_core.py:102-172).
//
// Execution model: a workgroup holds 1-10 independent 64-lane wavefronts that only share one LDS copy of
// the model plan.  A wavefront is split into 64/G groups of G lanes; each group owns one chain (clip) and
// cooperates on it: lanes of a group take the joints (joint-local pre-pass, gradient pass), the marker
// sites, 1/G of every nq-vector (held in registers) and -- max_width of them -- the bodies of the tree
// levels (FK program / level loop, stac_device.hpp).  The model "plan" (active subtree tables, FK step
// records) and the per-chain arrays live in LDS.  Every group runs its own solver state machine, so chains
// of one wavefront may be in different solves, line searches or frames: each trip round the main loop is
// one q_loss evaluation for every group.
//
// Arithmetic contract: every float operation sequence here is the one of oracle/stac_oracle.c
// (same expression trees with the same explicit fma placement, same summation orders; build with
// -ffp-contract=off so nothing else is contracted), so results are bit-identical to the CPU oracle;
// tests/test_gpu_parity.py checks exactly that.  There is no dense contraction on this path, hence no MFMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "stac_plan.hpp"
#include "stac_device.hpp"

#ifndef STAC_RT
#define STAC_RT 12  // sites per trip of a latency kernel's range sum (measured 6 / 8 / 12 / 16 / 24: 19.24 / 19.20 / 19.33 / 18.72 / 17.62 k frames/s on 40 x 250)
#endif

namespace stac {

// ------------------------------------------------------------------------------------------------
// q_phase kernel
// ------------------------------------------------------------------------------------------------

// WPE = waves per SIMD the register allocation is capped for (2 -> 256 VGPRs, 3 -> 168 VGPRs in workgroups of
// up to ten waves, 4 -> 128 VGPRs): the host picks the variant that lets the most chains be resident.
//
// SPEC (latency mode): EIGHT lane groups of G lanes work on ONE chain -- the eight groups of one wavefront
// (G = 8), or, when there are so few chains that the chip would stay empty, the groups of G / 8 wavefronts of
// one workgroup (G = 32: 4 waves, G = 64: 8 waves per chain; every phase outside the serial FK then gets
// 4-8 x the lanes and the waves overlap each other's LDS waits).  After a bootstrap evaluation of f, grad f at
// y, every trip evaluates in parallel the four line-search candidates cand_c = clip(y - (eta / 2^c) g),
// c = 0..3 (roles 0-3) and the four momentum points y_next(c) they would lead to (roles 4-7).  The first
// acceptable candidate c* is taken exactly as the sequential algorithm would; its gradient gives the stopping
// residual and role 4+c* already holds f, grad f at the next y: one trip per PG iteration instead of three,
// identical arithmetic per evaluation.  The joint pass of such a trip runs after the choice, for those two
// evaluations only, by all 64 lanes of the wave(s) that own them.  The solver state is replicated in every
// role; roles exchange {accept flag, loss} and the two gradients through a small per-chain LDS area
// (double-buffered by trip parity, so two workgroup barriers per trip suffice).
// SPEC = number of evaluation roles per chain in latency mode (0 = throughput mode): 8 = four candidates + their four
// momentum points; 4 = two + two (84 % of the iterations accept one of the first two candidates; the others take a
// second trip with candidates 2 and 3): two chains per wavefront at 8 lanes per role, for large batches.
// SPECP = SPEC | 1: the LEAN kernels -- the same kernel with the launch-wide choices of the common case (phase mode, not a single
// solve; a free root joint at qpos 0 .. 6; the uniform four-lanes-per-position FK program with 12-word records; every site in
// registers; no developer flags) as compile-time constants inside the trip loop: the other paths drop out of the loop body
// (10 % fewer instructions, no scalar spills).  The host takes it when all of that holds (launch_q_phase); same bits.
template <int G, int NQR, int WPE, int SPECP>
__global__ __launch_bounds__(WPE == 3 ? 768 : 512) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void q_phase_kernel(const QArgs a_in) {
    constexpr int SPEC = SPECP & ~1;
    constexpr bool LEAN = (SPECP & 1) != 0;
    const QArgs &a = a_in;
    static_assert(SPEC == 0 || (SPEC == 8 && (G == 8 || G == 16 || G == 32 || G == 64)) || (SPEC == 4 && (G == 8 || G == 16)), "speculative mode: 8 (or 4) roles of G lanes");
    extern __shared__ float lds[];
    constexpr int CPW = 64 / G;
    constexpr int NR = SPEC ? SPEC : 1, NC = SPEC ? SPEC / 2 : 1;  // roles per chain, of which candidates
    constexpr int LC = (G * NR >= 64) ? 64 : G * NR;                // lanes of one wavefront that work on one chain
    constexpr int NW = SPEC ? (G * NR >= 64 ? G * NR / 64 : 1) : 1; // SPEC: wavefronts per chain (one chain per workgroup when > 1)
    constexpr int CW = SPEC ? 64 / LC : 1;                          // SPEC, NW == 1: chains per wavefront
    const PlanHeader &H = a.h;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int grp = lane / G, lg = lane % G;
    const int nq = H.nq, K = H.K, nqpad = H.nqpad;

    // ---- stage the plan into LDS (shared by the block's wavefronts) ------------------------------------
    // The launch stages the words [plan_skip, total_words) of the blob; the host has rebased every off_* of this launch's
    // header to the first staged word (run_q), so P is the LDS base itself -- never a pointer in front of it: a lambda
    // that the compiler does not inline reads through a flat pointer, and below the LDS aperture that is a fault.
    float *P = lds;
    for (int i = threadIdx.x; i < H.total_words - H.plan_skip; i += blockDim.x) {
        float v = a.plan[H.plan_skip + i];
        if (a.bounds) {  // per-call lb / ub of stac_q_solve (StacCore.q_opt takes them per call, stac_core.py:193-235)
            if (i >= H.off_lb && i < H.off_lb + nqpad) v = a.bounds[i - H.off_lb];
            else if (i >= H.off_ub && i < H.off_ub + nqpad) v = a.bounds[nqpad + i - H.off_ub];
        }
        P[i] = v;
    }
    const int plan_words = (H.total_words - H.plan_skip + 3) & ~3;
    // per-kind qs_to_opt bit masks, one 32-bit word per (kind, lane-in-group): bit r <-> element r*G+lg
    uint32_t *MB = reinterpret_cast<uint32_t *>(lds + plan_words);
    const int nkinds = a.single ? 1 : a.P + 3;
    for (int i = threadIdx.x; i < nkinds * G; i += blockDim.x) {
        const int kind = i / G, l = i % G;
        uint32_t bits = 0;
        for (int r = 0; r < NQR; ++r) {
            const int e = r * G + l;
            if (e < nq && a.masks[kind * nqpad + e]) bits |= (1u << r);
        }
        MB[i] = bits;
    }
    // one more row: the coordinates that belong to a joint of the active subtree (bit r <-> element r*G+lg): only those have
    // a gradient entry, so the gradient vector needs no zeroing before the joint pass
    for (int l = threadIdx.x; l < G; l += blockDim.x) {
        uint32_t bits = 0;
        for (int r = 0; r < NQR; ++r) {
            const int e = r * G + l;
            if (e < nq && __builtin_bit_cast(int, a.plan[H.plan_skip + H.off_active + e])) bits |= (1u << r);
        }
        MB[nkinds * G + l] = bits;
    }
    // SPEC: a chain's block = 8 role regions + the exchange area; the block's waves share it (NW > 1) or own one each
    const int xch_words = 64 + 4 * nqpad + 4;
    const int cblock_words = NR * H.chain_stride + xch_words;
    const int cidx = SPEC ? (NW > 1 ? 0 : grp / NR) : 0;  // SPEC, NW == 1: this lane's chain among the wavefront's
    float *CBw = lds + plan_words + a.mb_words + (SPEC ? (NW > 1 ? 0 : wave * CW + cidx) * cblock_words : wave * CPW * H.chain_stride);
    const int role = SPEC ? (NW > 1 ? wave * CPW + grp : grp % NR) : grp;
    float *CB = CBw + (SPEC ? role : grp) * H.chain_stride;  // this evaluation's (chain's) region
    float *XB = CBw + NR * H.chain_stride;  // SPEC: [2][8][4] {accept, loss}, [2][2][nqpad] gradients, [4] queue word
    int trip_parity = 0;
    auto chain_sync = [&]() {  // all lanes that work on this chain
        if constexpr (NW > 1) __syncthreads();
        else wave_sync();
    };
    float *bx = CB + H.c_bx, *ja = CB + H.c_ja, *jn = CB + H.c_jn, *qsv = CB + H.c_qsv;
    // (the evaluation point and the gradient vector live in the site-wrench region: c_qe == c_gg == c_sw in this kernel, run_q checks)
    float *sw = CB + H.c_sw, *gg = sw, *r2 = CB + H.c_r2;
    float *qe = sw, *kpl = CB + H.c_kp;
    // Sites k = r * G + lg, r < kSiteRounds, belong to this lane: their keypoints and loss terms stay in registers when
    // that covers all K sites (the host then lays the chain out without the c_kp / c_r2 regions); else through LDS.
    constexpr int NSR = LEAN ? lean_site_rounds(G, NQR) : kSiteRounds;
    const bool site_regs = K <= NSR * G;
    float kpr[NSR][3];
#pragma unroll
    for (int r = 0; r < NSR; ++r) kpr[r][0] = kpr[r][1] = kpr[r][2] = 0.0f;
    __syncthreads();  // the only workgroup-wide barrier: the plan is shared by the block's waves

    const int *lev_adr = reinterpret_cast<const int *>(P + H.off_lev_adr);
    const float *brec = P + H.off_body, *jrec = P + H.off_joint, *srec = P + H.off_site;
    const float *lbv = P + H.off_lb, *ubv = P + H.off_ub, *qpos0 = P + H.off_qpos0;
    const int *quat_adr = reinterpret_cast<const int *>(P + H.off_quat_adr);

    // ---- per-chain solver state (uniform inside a group) ------------------------------------------
    // Placement (QArgs::place; throughput launches whose wavefronts are all resident at once).  2 500 wavefronts on 1 024
    // SIMDs: some SIMDs hold three, the others two, and a wavefront that shares its SIMD with two others runs its trips a
    // quarter slower than one that shares it with one -- for the whole launch, the wavefronts never move.  With the chains
    // in the order of their expected length (QArgs::perm) the wavefronts of the crowded SIMDs take the short chains and
    // the others the long ones, so that both kinds finish together.  Every wavefront counts itself on its SIMD (HW_ID /
    // XCC_ID), waits until all have (bounded: after the time-out it goes on with what it sees -- any outcome is a valid
    // assignment), and draws its position from its kind's end of the order.
    int wave_pos = (int)(blockIdx.x * wpb + wave);
    if constexpr (!SPEC) {
        if (a.place) {
            int pos = 0;
            if (lane == 0) {
                const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u;
                const int key = (int)((xcc << 10) | (((hw >> 8) & 0xFFu) << 2) | ((hw >> 4) & 3u));  // XCC | SE, SH, CU | SIMD
                int32_t *pl = a.place;
                atomicAdd(pl + kPlaceHdr + key, 1);
                __threadfence();
                atomicAdd(pl, 1);
                const int total = (int)(gridDim.x * wpb);
                const unsigned long long t0 = __builtin_readcyclecounter();
                while (__hip_atomic_load(pl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < total &&
                       __builtin_readcyclecounter() - t0 < 2000000ull)
                    __builtin_amdgcn_s_sleep(32);
                const int n = __hip_atomic_load(pl + kPlaceHdr + key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool crowded = n >= a.place_crowded;
                const int t = atomicAdd(pl + (crowded ? 1 : 2), 1);
                pos = crowded ? t : total - 1 - t;
                // the grid's padding wavefronts (positions without chains: whole workgroups are launched) sit at the SHORT end
                // of the order, below the shortest chains, so that it is the crowded SIMDs that draw them
                const int nreal = (a.C + CPW - 1) / CPW, npad = total - nreal;
                pos = pos >= npad ? pos - npad : nreal + pos;
            }
            wave_pos = __builtin_amdgcn_readfirstlane(pos);
        }
    }
    const int slot_id = SPEC ? (NW > 1 ? (int)blockIdx.x : (int)(blockIdx.x * wpb + wave) * CW + cidx) : wave_pos * CPW + grp;
    const int hstride = 3 * nqpad + 12;
    // resume = 1: this launch continues the chains that the throughput kernel handed off (QArgs::ctl / hand)
    const bool resuming = a.resume != 0 && slot_id < a.ctl[2] && slot_id < a.ctl[3];
    const float *hs = a.hand + (size_t)(resuming ? slot_id : 0) * hstride;
    const int *hi = reinterpret_cast<const int *>(hs + 3 * nqpad);
    // (QArgs::perm: the launch's chains in the order the host wants them on the slots / in the queue)
    // queue position -> chain: with a queue the chains go out longest first (perm is ascending), the short ones fill the end
    auto queue_chain = [&](const auto &a, const int pos) { return a.perm ? a.perm[(!SPEC && a.queue_slots > 0) ? a.C - 1 - pos : pos] : pos; };
    int chain = a.resume ? (resuming ? hi[0] : a.C) : (slot_id < a.C && a.perm ? queue_chain(a, slot_id) : slot_id);  // with a chain queue
                                                                 // (QArgs::queue_slots) a group takes further chains when it has finished one
    int st = chain < a.C ? ST_VG_Y : ST_DONE;
    int kind = a.single ? 0 : (a.do_root_opt ? 0 : 2);  // index into the mask table
    int frame = 0, iter = 0, nls = 0;
    float stepsize = 1.0f, t = 1.0f, eta = 1.0f, fy = 0.0f, fx = 0.0f;
    float error = __builtin_inff();
    uint32_t c_iter = 0, c_ls = 0, c_grad = 0, c_solves = 0;
    if (resuming) {
        kind = hi[1]; frame = hi[2]; iter = hi[3];
        stepsize = hs[3 * nqpad + 4]; t = hs[3 * nqpad + 5];
        c_iter = (uint32_t)hi[6]; c_ls = (uint32_t)hi[7]; c_grad = (uint32_t)hi[8]; c_solves = (uint32_t)hi[9];
        fx = hs[3 * nqpad + 10]; error = hs[3 * nqpad + 11];
    }

    float x[NQR], y[NQR], g[NQR], q0[NQR];
    // the lane's bounds: from the plan in LDS where registers are short (throughput kernels), in registers in latency mode
    // (a trip reads them four times per coordinate)
    // (latency kernels; throughput kernels where the register cap leaves room: +3 % on the 10 000-frame bench at 164 of 168 VGPRs)
    constexpr bool BREG = SPEC != 0 || (WPE == 2 && NQR <= 8) || (WPE == 3 && NQR <= 5);
    float lbr[BREG ? NQR : 1], ubr[BREG ? NQR : 1];
    if constexpr (BREG) {
#pragma unroll
        for (int r = 0; r < NQR; ++r) {
            const int e = r * G + lg;
            lbr[r] = e < nq ? lbv[e] : 0.0f;
            ubr[r] = e < nq ? ubv[e] : 0.0f;
        }
    }
#define LB(r, e) (BREG ? lbr[BREG ? (r) : 0] : lbv[e])
#define UB(r, e) (BREG ? ubr[BREG ? (r) : 0] : ubv[e])
    // the line-search candidate clip(y - eta * g) is recomputed where it is needed (same bits, fewer registers)
#define CAND(r, e) clipf(FMA(-eta, g[r], y[r]), LB(r, e), UB(r, e))
    const float eps = 1.1920929e-7f;

    // bx[0] = world (the lean kernels' split kinematics have no world entry)
    if (!LEAN && lg == 0) { st_tpos(bx, V3{0.f, 0.f, 0.f}); st_tquat(bx, Q4{1.f, 0.f, 0.f, 0.f}); }
    // lean: the oriented bodies' constant quaternions, the right factors of their products in P1 (PlanHeader::nbq; once per launch --
    // nothing else writes the region)
    if constexpr (LEAN) { for (int i = lg; i < 4 * H.nbq; i += G) CB[H.c3_bq + i] = P[H.off3_bq + i]; }

    // initial qpos, keypoints of frame 0, first solve
    size_t kp_chain = (size_t)(chain < a.C ? chain : 0) * a.F * 3 * K;
    // keypoints of a frame: this lane's sites into registers, or the whole frame into LDS
    // (TripCtx: what the chain-level helpers below need of the launch and the lane, handed in by the caller, so that nothing
    //  they use has to stay live across the trip loop: see "Launch arguments inside the trip loop" below)
    struct TripCtx { int lg, nq, K; float *kpl; const float *lbv, *ubv, *qpos0; const uint32_t *MB; };
    const TripCtx cx0{lg, nq, K, kpl, lbv, ubv, qpos0, MB};
    auto load_kp = [&](const auto &a, const TripCtx &cx, const size_t base) {
        const int lg = cx.lg, K = cx.K;
        float *const kpl = cx.kpl;
        if (K <= NSR * G) {
#pragma unroll
            for (int r = 0; r < NSR; ++r) {
                const int k = r * G + lg;
                if (k < K) { kpr[r][0] = a.kp[base + 3 * k]; kpr[r][1] = a.kp[base + 3 * k + 1]; kpr[r][2] = a.kp[base + 3 * k + 2]; }
            }
        } else {
            for (int i = lg; i < 3 * K; i += G) kpl[i] = a.kp[base + i];
        }
    };
#pragma unroll
    for (int r = 0; r < NQR; ++r) {
        const int e = r * G + lg;
        float v = 0.f;
        if (e < nq && st != ST_DONE) v = a.q_init ? a.q_init[(size_t)chain * nq + e] : qpos0[e];
        q0[r] = v;
    }
    if (st != ST_DONE) {
        load_kp(a, cx0, kp_chain + (size_t)frame * 3 * K);
        if (!a.single && kind < 2 && !resuming) {  // root pass: q0[:3] = keypoint of the root marker (compute_stac.py:57-59)
#pragma unroll
            for (int r = 0; r < NQR; ++r) {
                const int e = r * G + lg;
                if (e < 3) q0[r] = a.kp[kp_chain + 3 * a.root_kp_idx + e];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < NQR; ++r) { x[r] = q0[r]; y[r] = q0[r]; g[r] = 0.f; }
    // root fast trips (QArgs::root_fast): are the joint-local quaternions of this group's chain still in its ja entries?
    bool ql_fresh = false;
    // ... and do this lane's coordinates outside the root passes' lie inside the box?  Then clip(y - eta * 0) = y for them:
    // they drop out of every norm and never move, and the solver transition of a root fast trip looks at register 0 only.
    // (They are constant over a chain's root solves, so this is settled when the chain starts.)
    bool tail_ok = true;
    // the qs_to_opt bits of the running solve for this lane's coordinates (MB[kind][lane]): carried across the trips of a solve, read
    // again where the kind changes (end of a solve, next chain) -- not at the top of every trip, where the staging waits for it
    uint32_t mbits_c = 0;
    auto check_tail = [&](const auto &a, const TripCtx &cx) {
        const int lg = cx.lg, nq = cx.nq;
        const float *const lbv = cx.lbv, *const ubv = cx.ubv;
        tail_ok = true;
#pragma unroll
        for (int r = 0; r < NQR; ++r) {
            const int e = r * G + lg;
            if (e < nq && e >= a.root_fast && !(q0[r] >= lbv[e] && q0[r] <= ubv[e])) tail_ok = false;
        }
    };
    if (!SPEC && a.root_fast > 0) check_tail(a, cx0);
    // the same for the next chain a group takes from the queue (all lanes of the group call it together)
    auto begin_chain = [&](const auto &a, const TripCtx &cx, const int c) {
        const int lg = cx.lg, nq = cx.nq, K = cx.K;
        const float *const qpos0 = cx.qpos0;
        chain = c;
        kp_chain = (size_t)c * a.F * 3 * K;
        kind = a.do_root_opt ? 0 : 2;
        frame = 0; iter = 0; nls = 0;
        stepsize = 1.0f; t = 1.0f; error = __builtin_inff();
        c_iter = c_ls = c_grad = c_solves = 0;
#pragma unroll
        for (int r = 0; r < NQR; ++r) {
            const int e = r * G + lg;
            float v = 0.f;
            if (e < nq) v = a.q_init ? a.q_init[(size_t)c * nq + e] : qpos0[e];
            if (kind < 2 && e < 3) v = a.kp[kp_chain + 3 * a.root_kp_idx + e];
            q0[r] = v; x[r] = v; y[r] = v; g[r] = 0.f;
        }
        load_kp(a, cx, kp_chain);
        st = ST_VG_Y;
        ql_fresh = false;
        mbits_c = cx.MB[kind * G + lg];
        if (!SPEC && a.root_fast > 0) check_tail(a, cx);
    };
    if (resuming) {
#pragma unroll
        for (int r = 0; r < NQR; ++r) {
            const int e = r * G + lg;
            if (e < nq) { x[r] = hs[e]; y[r] = hs[nqpad + e]; q0[r] = hs[2 * nqpad + e]; }
        }
    }
    if constexpr (SPEC != 0 && NW > 1) {
        // the exchanged gradient vectors are zeroed ONCE: every pass rewrites the entries of all active joints, and the
        // others -- coordinates of joints outside the marker-ancestor subtree -- are never written (two waves fill a vector)
        for (int e = threadIdx.x; e < 4 * nqpad; e += blockDim.x) XB[64 + e] = 0.0f;
        __syncthreads();
    }
    wave_sync();

    // root fast trips: the lane's one weighted (trunk) site, if no lane has more than one -- then a single round of the
    // site pass covers them all (sites1)
    int trunk_r = -1;
    bool sites1 = false;
    // (the trunk weights of the lane's sites as bits: the root passes' site terms would fetch them from global memory in every trip)
    uint32_t kpw_bits = 0;
    if (!a.single && site_regs && a.kpw) {
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < NSR; ++r) {
            const int k = r * G + lg;
            if (k < K && a.kpw[k]) { trunk_r = r; ++cnt; kpw_bits |= 1u << r; }
        }
        sites1 = !SPEC && a.root_fast > 0 && !__any(cnt > 1);
    }
    mbits_c = MB[kind * G + lg];
    // (the coordinates that have a gradient entry at all -- the row behind the kinds' --: constant for the launch)
    // (lean kernels; the generic ones, some of them at their register cap, read it where they use it)
    uint32_t act_bits = 0;
    if constexpr (LEAN) {
        act_bits = MB[nkinds * G + lg];
        asm volatile("" : "+v"(act_bits));
    }
    // ---- lean latency kernels: what a lane reads from the plan in EVERY trip, kept in registers (round 6) ---------------------------
    // A lone wavefront waits out every LDS round trip, and the per-item phases of a trip begin with two or three DEPENDENT ones: the item's
    // record (which joint / site / rotation is this lane's?), then the addresses in it, then the data.  The lane's items never change, and
    // these kernels have a hundred vector registers to spare: the records of the lane's sites and of its joints of the pre-pass are read
    // once, here (the full program's; a root pass -- the first frame of a clip only -- reads the pruned program's site words as before).
#ifndef STAC_NO_LATPIN
    constexpr bool LATPIN = LEAN && SPEC != 0 && NQR <= 5;  // (the wide shapes -- eight solver registers per lane -- have no registers to spare)
#else
    constexpr bool LATPIN = false;
#endif
    // (the 16-lane lean throughput kernels have a dozen registers to spare under their cap: the site records as well, +1 %)
#ifndef STAC_NO_THRPIN
    constexpr bool SITEPIN = LATPIN || (LEAN && SPEC == 0 && G == 16);
#else
    constexpr bool SITEPIN = LATPIN;
#endif
#ifndef STAC_NO_THRPIN2
    constexpr bool PREPIN = LATPIN || (LEAN && SPEC == 0 && G == 16);  // (and the pre-pass joints: +0.8 % on 100 000 frames, 164 of 168 registers)
#else
    constexpr bool PREPIN = LATPIN;
#endif
    constexpr int PJ = PREPIN ? (G == 32 ? 2 : 3) : 1;  // rounds of the pre-pass whose joints are pinned (rodent: all of them)
    float4 pin_sr[NSR];      // SiteRec of the lane's sites
    int pin_s3[NSR];         // their body position / quaternion words under the full program
    int pin_jad[PJ], pin_jout[PJ];
    float pin_jq0[PJ], pin_jax[PJ][3];
    if constexpr (SITEPIN) {
        const int *site3f = reinterpret_cast<const int *>(P + H.off3_prog) + 16 * (H.fk3_cap1 + 2) + 4 * H.fk3_cap2 + 4 * H.fk3_cap3;
#pragma unroll
        for (int r = 0; r < NSR; ++r) {
            const int k = min(r * G + lg, K - 1);
            pin_sr[r] = lds4(srec + 4 * k);
            pin_s3[r] = site3f[k];
            asm volatile("" : "+v"(pin_sr[r].x), "+v"(pin_sr[r].y), "+v"(pin_sr[r].z), "+v"(pin_sr[r].w), "+v"(pin_s3[r]));
        }
    }
    if constexpr (PREPIN) {
#pragma unroll
        for (int u = 0; u < PJ; ++u) {
            const int j = min(lg + 1 + u * G, H.naj - 1);  // (a lane without a joint in this round repeats the last one: the same value to the same words)
            const float *jr = jrec + 12 * j;
            pin_jad[u] = reinterpret_cast<const int *>(jr)[1];
            pin_jq0[u] = jr[7];
            pin_jax[u][0] = jr[8]; pin_jax[u][1] = jr[9]; pin_jax[u][2] = jr[10];
            pin_jout[u] = H.c3_ql + 4 * j;
            asm volatile("" : "+v"(pin_jad[u]), "+v"(pin_jq0[u]), "+v"(pin_jax[u][0]), "+v"(pin_jax[u][1]), "+v"(pin_jax[u][2]), "+v"(pin_jout[u]));
        }
    }
    // ... the rotations of the lane's first PR2 rounds of P2 (full program), the joint of the gradient pass's first round and the range
    // tasks of its first two (the lanes of a chain's wavefronts take joint `lane` / task `lane`, `lane + 64`)
    constexpr int PR2 = 3;
    float4 pin_t2[LATPIN ? PR2 : 1];
    Fk3Head pin_h1 = {};   // P1's first two records of the lane's position, P3's first three restart words (full program)
    int pin_h3[3] = {0, 0, 0};
    struct GPin { int ad, rw, jw, j; float ax, ay, az; };  // qpos address, word of the range sum, anchor / pre-joint quaternion words, joint
    struct TPin { int lo, hi, src, dst, maxn; };            // site range, word of the first site's component, word of the sum; the round's longest range
    GPin pin_g = {};
    TPin pin_t[LATPIN ? 3 : 1] = {};
    if constexpr (LATPIN) {
        const float *T2f = P + H.off3_prog + 16 * (H.fk3_cap1 + 2);
        const int n2f = (int)((unsigned)H.fk3_n >> 16);
        {
            const Fk3Lane L3 = fk3_lane(lg & 15);
            pin_h1 = fk3_p1_head(P + H.off3_prog, L3);
            const int *T3f = reinterpret_cast<const int *>(T2f + 4 * H.fk3_cap2) + L3.pp;
            pin_h3[0] = T3f[0]; pin_h3[1] = T3f[4]; pin_h3[2] = T3f[8];
            asm volatile("" : "+v"(pin_h3[0]), "+v"(pin_h3[1]), "+v"(pin_h3[2]));
        }
#pragma unroll
        for (int r = 0; r < PR2; ++r) {
            pin_t2[r] = lds4(T2f + 4 * min(r * G + lg, n2f - 1));
            asm volatile("" : "+v"(pin_t2[r].x), "+v"(pin_t2[r].y), "+v"(pin_t2[r].z), "+v"(pin_t2[r].w));
        }
        constexpr int LCp = (G * NR >= 64) ? 64 : G * NR;
        {   // the joint this lane takes in the first round of the gradient pass: `lane` of the chain's lanes (NW == 1: the two evaluations' joints one after the other)
            const int ll = lane % LCp;
            const int j = NW > 1 ? min(ll, H.naj - 1) : (ll >= H.naj ? min(ll - H.naj, H.naj - 1) : ll);
            const float *jr = jrec + 12 * j;
            pin_g.j = j;
            pin_g.ad = reinterpret_cast<const int *>(jr)[1];
            pin_g.ax = jr[8]; pin_g.ay = jr[9]; pin_g.az = jr[10];
            pin_g.rw = H.c_rw + kXf * reinterpret_cast<const int *>(jr)[11];
            pin_g.jw = reinterpret_cast<const int *>(P + H.off3_site)[K + j];
            asm volatile("" : "+v"(pin_g.ad), "+v"(pin_g.rw), "+v"(pin_g.jw), "+v"(pin_g.j), "+v"(pin_g.ax), "+v"(pin_g.ay), "+v"(pin_g.az));
        }
        if constexpr (NW > 1) {  // range tasks: [0], [1] = the owner's first two (t = base + lane, + 64), [2] = the helper's (component `lane` of range 0)
            const RangeRec *rrec_p = reinterpret_cast<const RangeRec *>(P + H.off_range);
            const int base = NW == 2 ? 0 : 6;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int t = u < 2 ? min(base + lane + 64 * u, 6 * H.nrange - 1) : min(lane, 5);
                const int r = t / 6, k = t - 6 * r;
                const int co = k < 3 ? k : kXq + k - 3;
                // (the ranges are sorted longest first: a round's first task has its longest range)
                const int r_first = u < 2 ? min((base + 64 * u) / 6, H.nrange - 1) : 0;
                pin_t[u] = TPin{rrec_p[r].lo, rrec_p[r].hi, H.c_sw + co, H.c_rw + kXf * r + co, rrec_p[r_first].hi - rrec_p[r_first].lo};
                asm volatile("" : "+v"(pin_t[u].lo), "+v"(pin_t[u].hi), "+v"(pin_t[u].src), "+v"(pin_t[u].dst), "+v"(pin_t[u].maxn));
            }
        }
    }
    PROF_DECL;
    // ================================= main loop: one q_loss evaluation per trip ==================
    const int lg_outer = lg;
    // Launch arguments inside the trip loop.  QArgs is 424 bytes of kernel argument: read once in the prologue, every field
    // -- and every address or lane mask derived from one -- stays live across the whole trip loop, and the kernel ran with 190
    // to 310 scalars spilled into vector-register lanes (v_readlane at every use: a tenth of its vector instructions) on top of
    // which the vector registers themselves spilled to scratch.  So the loop reads what it needs from the kernarg segment again,
    // where it needs it (scalar loads through `a`, below; the constant cache holds the segment): throughput kernels through a
    // pointer that is made opaque once per trip, so that no load can be hoisted out of the loop; latency kernels (a lone
    // wavefront per SIMD has nothing to hide a scalar load behind) leave the choice to the compiler and launder only the
    // pointer of the cold end-of-solve block.  Measured (10 000-frame bench / 40 x 250 clips): +3.5 % / -1.5 %; SGPR spills of the
    // shipped instantiations 103-313 -> 0-35, the headline kernel from 19 spilled VGPRs to none at 154 of 168.
    typedef const __attribute__((address_space(4))) QArgs KQArgs;
    KQArgs *const ak_base = (KQArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr bool kReloadPerTrip = SPEC == 0;
#ifndef STAC_VPIN
#define STAC_VPIN (SPEC != 0 && G >= 16)
#endif
#ifndef STAC_SPIN
#define STAC_SPIN false
#endif
#ifndef STAC_NO_PIN
    constexpr int kVPin = (STAC_VPIN ? 1 : 0) | (STAC_SPIN ? 2 : 0);
    const HotHeader hot_h = pin_header<kVPin>(a.h);
    const HotArgs hot_a = pin_args<kVPin>(a);
#endif
    const int cb_words = (int)(CB - lds);
    while (__any(st != ST_DONE)) {
        PROF_TICK(0);  // loop control
        PROF_TRIP;
        KQArgs *ak_t = ak_base;
        if constexpr (kReloadPerTrip) asm volatile("" : "+s"(ak_t));
#ifndef STAC_NO_PIN
        TripArgs a_t = trip_args<kVPin>(hot_a, *ak_t);
        TripHeader H_t = trip_header<kVPin>(hot_h, ak_t->h);
        if constexpr (LEAN) {  // (what the host has checked for this launch: launch_q_phase)
            a_t.single = 0; a_t.flags = 0; a_t.free0p = 1;
            H_t.fk_uniform = 1; H_t.fk_rec_words = 12;
            __builtin_assume(4 * H_t.max_width <= G);
            __builtin_assume(H_t.K <= NSR * G);
        }
        const TripArgs &a = a_t;
        const TripHeader &H = H_t;
#else
        KQArgs &a = *ak_t;
        const auto &H = a.h;
#endif
        // ... and neither does anything derived from it: the names below shadow the prologue's
        const int nq = H.nq, K = H.K, nqpad = H.nqpad;
        const int plan_words = (H.total_words - H.plan_skip + 3) & ~3;
        uint32_t *const MB = reinterpret_cast<uint32_t *>(lds + plan_words);
        const int nkinds = a.single ? 1 : a.P + 3;
        const int hstride = 3 * nqpad + 12;
        const bool site_regs = LEAN || K <= NSR * G;
        const float *const jrec = P + H.off_joint, *const srec = P + H.off_site;
        const float *const lbv = P + H.off_lb, *const ubv = P + H.off_ub;
        const int *const quat_adr = reinterpret_cast<const int *>(P + H.off_quat_adr);
        // The lane index and the chain's region likewise: opaque per trip, so that every address and every lane mask that
        // depends on them is recomputed inside the trip (one VALU instruction each) instead of being hoisted out of the loop,
        // where dozens of them lived for the whole launch and were spilled.
        int lg_t = lg_outer, cb_t = cb_words;
        asm volatile("" : "+v"(lg_t), "+v"(cb_t));
        const int lg = lg_t;
        float *const CB = lds + cb_t;
        float *const bx = CB + H.c_bx, *const ja = CB + H.c_ja, *const jn = CB + H.c_jn, *const qsv = CB + H.c_qsv;
        float *const sw = CB + H.c_sw, *const gg = sw, *const r2 = CB + H.c_r2;
        float *const qe = sw, *const kpl = CB + H.c_kp;
        const TripCtx cx{lg, nq, K, kpl, lbv, ubv, P + H.off_qpos0, MB};
        // lean kernels (split kinematics, PlanHeader::fk3): word of the root position inside a chain's region (behind the slots of P3)
        const int root_w = LEAN ? H.c3_pb + 12 * H.fk3_cap3 : 0;
        if (!SPEC && a.ctl && !a.resume) {
            // hand-off: a chain about to start an iteration after most chains of the launch are done goes to the
            // latency kernel (its state is complete at this point: x, y, q0 and a dozen scalars)
            // (looked at every eighth iteration: the counter lives in L2)
            if (__any(st == ST_VG_Y && (iter & 7) == 0)) {
                // (using the count requested at the previous look, so that the trip that looks does not wait for L2: measured 1.7 % SLOWER)
                const int done_cnt = __hip_atomic_load(a.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (st == ST_VG_Y && (iter & 7) == 0 && done_cnt >= a.ctl[1]) {
                    int slot = 0;
                    if (lg == 0) slot = atomicAdd(a.ctl + 2, 1);
                    slot = __shfl(slot, grp * G, 64);
                    if (slot < a.ctl[3]) {
                        float *hd = a.hand + (size_t)slot * hstride;
#pragma unroll
                        for (int r = 0; r < NQR; ++r) {
                            const int e = r * G + lg;
                            if (e < nq) { hd[e] = x[r]; hd[nqpad + e] = y[r]; hd[2 * nqpad + e] = q0[r]; }
                        }
                        if (lg == 0) {
                            int *hdi = reinterpret_cast<int *>(hd + 3 * nqpad);
                            hdi[0] = chain; hdi[1] = kind; hdi[2] = frame; hdi[3] = iter;
                            hd[3 * nqpad + 4] = stepsize; hd[3 * nqpad + 5] = t;
                            hdi[6] = (int)c_iter; hdi[7] = (int)c_ls; hdi[8] = (int)c_grad; hdi[9] = (int)c_solves;
                            hd[3 * nqpad + 10] = fx; hd[3 * nqpad + 11] = error;
                        }
                        st = ST_DONE;
                    } else if (lg == 0) {
                        atomicAdd(a.ctl + 2, -1);  // no room: stay
                    }
                }
            }
            if (!__any(st != ST_DONE)) break;
        }
        if constexpr (!SPEC) {
            // wave-synchronous root phase: chains that have finished their root solves go on when no chain of the
            // wavefront is in one any more
            const bool root_live = st != ST_DONE && st != ST_WAIT && st != ST_NEXT && !a.single && kind < 2;
            const bool any_root_live = __any(root_live);  // (all lanes vote: not inside the condition below)
            if (st == ST_WAIT && !any_root_live) st = ST_VG_Y;
            // chain queue with root fast trips: the groups of a wavefront take their next chains TOGETHER, when all of them
            // have finished (ST_NEXT), so that the root solves of the new chains run as fast trips too (the queue hands the
            // chains out in the order of their expected length: the four of a wavefront finish close to each other)
            if (a.queue_slots > 0 && a.root_fast > 0) {
                const bool busy = st != ST_DONE && st != ST_NEXT;
                if (!__any(busy) && __any(st == ST_NEXT)) {
                    if (st == ST_NEXT) {
                        int nxt = 0;
                        if (lg == 0) nxt = atomicAdd(a.ctl + 4, 1);
                        nxt = __shfl(nxt, grp * G, 64);
                        if (nxt < a.C) begin_chain(a, cx, queue_chain(a, nxt));
                        else st = ST_DONE;
                    }
                    if (!__any(st != ST_DONE)) break;
                }
            }
        }
        const int st_in = st;
        const bool live_in = st_in != ST_DONE && st_in != ST_WAIT && (SPEC || st_in != ST_NEXT);
        const uint32_t mbits = mbits_c;
        // A line-search candidate that is accepted becomes x_next, whose gradient the stopping test
        // needs (the oracle's separate VG_X evaluation runs the very same FK).  The step size doubles
        // after every iteration, so the first candidate is almost always rejected and the second
        // accepted: evaluate the gradient together with every candidate after the first.
        // (the latency kernels never enter ST_LS / ST_VG_X -- a solve goes VG_Y, then SPEC trips to its end --: what belongs to
        //  those states is compiled out of them here and in the transition below)
        const bool ls_wants_grad = SPEC == 0 && (st_in == ST_LS) && (nls >= 1) && !(a.flags & 1);
        const bool want_grad = (st_in == ST_VG_Y) || (st_in == ST_VG_X) || ls_wants_grad || (SPEC && st_in == ST_SPEC);
        const bool any_grad = __any(want_grad);
        // The gradient pass is issued for the whole wavefront as soon as one group needs it, so a FIRST candidate gets
        // its gradient for free whenever a neighbour chain asks for one -- and if it is accepted (one iteration in six)
        // the separate evaluation of x_next disappears as well.
        const bool ls_with_grad = SPEC == 0 && (st_in == ST_LS) && !(a.flags & 1) && (nls >= 1 || any_grad);
        // t_next and the momentum coefficient of the running iteration (functions of t only; recomputed every
        // trip instead of being carried); SPEC: this group's candidate scale 2^-c
        const float spec_pow = SPEC ? ((role % NC) == 0 ? 1.0f : (role % NC) == 1 ? 0.5f : (role % NC) == 2 ? 0.25f : 0.125f) : 1.0f;
        // (t == t_iter: the throughput kernels take both from the plan's table while iter < kTTab -- a square root and a division less per
        //  trip: +1.5-2 % on the 10 000-frame bench --; the latency kernels, which would wait for the read at once (measured: -0.5 %),
        //  and the 128-register variants, which have no register left for it, compute them, like everyone beyond the table)
        float spec_tn, spec_beta;
        if constexpr (SPEC == 0 && WPE != 4) {
            const float2 e = *reinterpret_cast<const float2 *>(jrec - 2 * kTTab + 2 * min(iter, kTTab - 1));
            spec_tn = e.x;
            spec_beta = e.y;
            if (__any(iter >= kTTab)) {
                const float tn = 0.5f * (1.0f + __builtin_sqrtf(1.0f + 4.0f * t * t));
                const float be = (t - 1.0f) / tn;
                spec_tn = iter >= kTTab ? tn : spec_tn;
                spec_beta = iter >= kTTab ? be : spec_beta;
            }
        } else {
            spec_tn = 0.5f * (1.0f + __builtin_sqrtf(1.0f + 4.0f * t * t));
            spec_beta = (t - 1.0f) / spec_tn;
        }

        // root passes weigh the trunk keypoints only: when every live chain of the wave is in one, the kinematics stop at
        // the ancestors of those keypoints (the other sites contribute exact zeros, written as such below)
        const bool root_pass = !a.single && kind < 2;
        const int n_mlev_root_a = LEAN ? a.fk3r_n : a.n_mlev_root;  // (lean: the pruned split-kinematics program, if the call has one)
        const int n_ml_root = (n_mlev_root_a > 0 && !__any(live_in && !root_pass)) ? n_mlev_root_a : 0;
        // root fast trip: only the root coordinates are staged and only the root joint's local transform is refreshed
        // (lite) once the other joints' local quaternions sit untouched in their ja entries
        const bool fast_trip = !SPEC && a.root_fast > 0 && n_ml_root > 0;
        if (fast_trip && !tail_ok) {
            // A chain whose start pose has coordinates outside the box (mouse: ranges that do not contain the rest angle)
            // moves them onto the box in the first iteration of the solve; from then on x = y sits inside and the lite
            // transition is exact for it as well (the kinematics never see x or y of a coordinate that is not optimised)
            bool ok = true;
#pragma unroll
            for (int r = 0; r < NQR; ++r) {
                const int e = r * G + lg;
                if (e < nq && e >= a.root_fast && !(x[r] == y[r] && x[r] >= lbv[e] && x[r] <= ubv[e])) ok = false;
            }
            tail_ok = ok;
        }
        const bool lite = fast_trip && !__any(live_in && !ql_fresh);

        // the world entry of the transform array (the gradient pass of the previous trip left its range sums there)
        if (!LEAN && lg == 0) { st_tpos(bx, V3{0.f, 0.f, 0.f}); st_tquat(bx, Q4{1.f, 0.f, 0.f, 0.f}); }
        // ---- make_qs (utils.py:129-144): qf = (1 - mask) * q0 + mask * point ---------------------
        // A free joint at qpos 0 .. 6 (active joint 0: QArgs::free0p): its pre-pass out of the registers of lanes 0 .. 6, which
        // hold the staged coordinates v.  Every lane fetches the four raw quaternion components from lanes 3 .. 6 (one
        // cross-lane round trip) and computes |q| itself -- the same fma chain as normalize4 -- so the four components are
        // divided side by side and stored by their own lanes: {position, unit quaternion} into the joint's entry (the FK
        // program's synthetic parent of the root body), the unit quaternion and |q| for the gradient pass, and the
        // evaluation point as MJX writes it back.  (One lane doing all of it, next to fifteen hinges, was the long pole
        // of the pre-pass's first round.)
        auto free0_prepass = [&](const float v, const int qord) {
            const int gb = lane - lg;
            const float qw = __shfl(v, gb + 3, 64), qx = __shfl(v, gb + 4, 64), qy = __shfl(v, gb + 5, 64), qz = __shfl(v, gb + 6, 64);
            const float n = __builtin_sqrtf(FMA(qz, qz, FMA(qy, qy, FMA(qx, qx, qw * qw))));
            const float dn = n + (n == 0.0f ? 1e-6f : 0.0f);
            if constexpr (LEAN) {
                // the same stores without a branch: every lane divides, the lanes that hold nothing aim at the sink entry
                const float qn = v / dn;
                const int c = lg - 3;  // 0 .. 3 = w, x, y, z on lanes 3 .. 6
                const float val = lg < 3 ? v : qn;
                float *const sink = CB + H.c_sink;
                float *d_qe = lg < 7 ? qe + lg : sink;
                // (split kinematics: the root position slot of P3 and entry 0 of the quaternion array, which holds (w, x, y, z))
                float *d_ja = lg < 3 ? CB + root_w + lg : (lg < 7 ? CB + H.c3_qb + c : sink + 1);
                float *d_qs = (lg >= 3 && lg < 7) ? qsv + 4 * qord + c : sink + 2;
                float *d_jn = lg == 3 ? jn + qord : sink + 3;
                *d_qe = val;
                *d_ja = val;
                *d_qs = qn;
                *d_jn = n;
            } else if (lg < 3) {
                qe[lg] = v;
                ja[lg] = v;
            } else if (lg < 7) {
                const float qn = v / dn;
                const int c = lg - 3;  // 0 .. 3 = w, x, y, z
                qe[lg] = qn;
                ja[kXq + (c == 0 ? 3 : c - 1)] = qn;  // entries hold (x, y, z, w)
                qsv[4 * qord + c] = qn;
                if (c == 0) jn[qord] = n;
            }
        };
        const int j0 = a.free0p ? 1 : 0;  // first joint of the per-joint loops
        if (lite && a.free0p) {
            // root fast trip on a free root: staging and the pre-pass in one go
            const float pt = (st_in == ST_VG_Y) ? y[0] : ((st_in == ST_LS) ? CAND(0, lg) : x[0]);
            const float mi = (mbits & 1u) ? 1.0f : 0.0f;
            free0_prepass((1.0f - mi) * q0[0] + mi * pt, a.free0p - 1);
        } else if (lite) {
            if (lg < a.root_fast) {  // (the root coordinates are elements 0 .. root_fast - 1 <= G: register 0)
                const float pt = (st_in == ST_VG_Y) ? y[0] : ((st_in == ST_LS) ? CAND(0, lg) : x[0]);
                const float mi = (mbits & 1u) ? 1.0f : 0.0f;
                qe[lg] = (1.0f - mi) * q0[0] + mi * pt;
            }
        } else {
        // straight-line: the bounds of all the lane's coordinates are fetched together (one LDS round trip, not one per
        // coordinate behind a branch each), the point is chosen by selects
        float lbs[NQR], ubs[NQR];
#pragma unroll
        for (int r = 0; r < NQR; ++r) {
            const int e = r * G + lg, eb = e < nq ? e : 0;
            lbs[r] = LB(r, eb);
            ubs[r] = UB(r, eb);
        }
        const float eta_s = (SPEC && st_in == ST_SPEC) ? eta * spec_pow : eta;  // eta / 2^c (exact)
        float v0 = 0.0f;
#pragma unroll
        for (int r = 0; r < NQR; ++r) {
            const int e = r * G + lg;
            const float cr = clipf(FMA(-eta_s, g[r], y[r]), lbs[r], ubs[r]);
            float pt = (st_in == ST_VG_Y) ? y[r] : ((SPEC == 0 && st_in == ST_LS) ? cr : x[r]);
            if constexpr (SPEC != 0) {  // (selects, not branches: every lane computes the momentum point, the roles it is for take it)
                const float mom = FMA(spec_beta, cr - x[r], cr);
                const float ps = role < NC ? cr : mom;
                pt = st_in == ST_SPEC ? ps : pt;
            }
            const float mi = ((mbits >> r) & 1u) ? 1.0f : 0.0f;
            const float v = (1.0f - mi) * q0[r] + mi * pt;
            if (r == 0) v0 = v;
            if constexpr (LEAN) {  // (a select on the address instead of a branch round the store: coordinates nobody stages go to the sink)
                float *dst = (e < nq && !(e < 7)) ? qe + e : CB + H.c_sink;
                *dst = v;
            } else {
                if (e < nq && !(a.free0p && e < 7)) qe[e] = v;
            }
        }
        if (a.free0p) free0_prepass(v0, a.free0p - 1);
        }
        wave_sync();
        PROF_TICK(1);  // stage

        if (!(lite && a.free0p)) {  // (a root fast trip on a free root has nothing else to prepare)
            if constexpr (LEAN) {
                // every hinge's joint-local quaternion (w, x, y, z) into its own array: a pruned trip keeps what it does not use
                // Three rounds of lanes at once (the rodent's 38 hinges on 16 lanes: all of them): the rounds are independent chains of
                // dependent instructions, which the scheduler interleaves.  A lane whose round has no joint left takes the last
                // joint again: the same value to the same words.
                // (throughput kernels: a lone wavefront of the latency kernels pays per instruction, dependent or not)
                constexpr int U = SPEC != 0 ? (G == 32 ? 2 : 1) : (G == 16 ? 3 : 2);
                if constexpr (PREPIN) {  // (the lane's joints of the first PJ rounds out of registers: LATPIN, above)
#pragma unroll
                    for (int u = 0; u < PJ; ++u) {
                        if (u > 0 && 1 + u * G >= H.naj) break;  // (wave-uniform: no joint left for any lane)
                        const float angle = qe[pin_jad[u]] - pin_jq0[u];
                        float sn, cs;
                        sincos_(angle * 0.5f, &sn, &cs);
                        *reinterpret_cast<float4 *>(CB + pin_jout[u]) = float4{cs, pin_jax[u][0] * sn, pin_jax[u][1] * sn, pin_jax[u][2] * sn};
                    }
                }
                for (int j0 = lg + 1 + (PREPIN ? PJ * G : 0); j0 < H.naj; j0 += U * G) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int j = min(j0 + u * G, H.naj - 1);
                        const float *jr = jrec + 12 * j;
                        const int ad = reinterpret_cast<const int *>(jr)[1];
                        const float4 jp4 = lds4(jr + 4);  // pos, q0
                        const float4 ja4 = lds4(jr + 8);  // axis, range id
                        const float angle = qe[ad] - jp4.w;
                        float sn, cs;
                        sincos_(angle * 0.5f, &sn, &cs);
                        *reinterpret_cast<float4 *>(CB + H.c3_ql + 4 * j) = float4{cs, ja4.x * sn, ja4.y * sn, ja4.z * sn};
                    }
                }
            } else {
                joint_local_prepass<LEAN>(H, P, CB, lg, G, lite ? a.n_root_joints : H.naj, j0, (a.flags & 16) != 0 && j0 == 1);
            }
            wave_sync();
        }
        PROF_TICK(10);  // joint-local pre-pass

        // ---- forward kinematics, level by level (mjx smooth.kinematics) -------------
        const int *site3 = nullptr;  // lean: per site, the words of its body's position and quaternion under the program that ran
        if constexpr (LEAN) {
            const bool rootp = n_ml_root > 0;
            const float *pg = P + (rootp ? H.off3_root : H.off3_prog);
            Fk3Prog G3;
            G3.T1 = pg;
            G3.T2 = pg + 16 * (H.fk3_cap1 + 2);
            G3.T3 = reinterpret_cast<const int *>(G3.T2 + 4 * H.fk3_cap2);
            G3.site = G3.T3 + 4 * H.fk3_cap3;
            const int n123 = rootp ? n_ml_root : H.fk3_n;  // (one word for the three counts: one select, no branch round three loads)
            G3.n1 = n123 & 0xFF;
            G3.n3 = (n123 >> 8) & 0xFF;
            G3.n2 = (int)((unsigned)n123 >> 16);
            site3 = G3.site;
#if defined(STAC_PROFILE) && defined(STAC_PROF_FK3)  // (diagnostic: the three passes charged to stamps 2, 4 and 9)
            {
                const Fk3Lane L3 = fk3_lane(lg & 15);
                if (G == 16 || lg < 16) fk3_p1(G3.T1, G3.n1 >> 1, CB, L3);
                wave_sync();
                PROF_TICK(2);
                fk3_p2<(SPEC == 0 && G == 16)>(G3.T2, G3.n2, CB, lg, G);
                wave_sync();
                PROF_TICK(4);
                if (G == 16 || lg < 16) fk3_p3(G3.T3, G3.n3 >> 2, CB, H.c3_pb, L3);
                wave_sync();
                PROF_TICK(9);
            }
#else
            if constexpr (LATPIN) {
                const Fk3Lane L3 = fk3_lane(lg & 15);
                if (G == 16 || lg < 16) fk3_p1(G3.T1, G3.n1 >> 1, CB, L3, !rootp, pin_h1);
                wave_sync();
                if (rootp) fk3_p2<false>(G3.T2, G3.n2, CB, lg, G);  // (a root pass runs the pruned program's rotations)
                else fk3_p2_pinned<PR2>(pin_t2, G3.T2, G3.n2, CB, lg, G);
                wave_sync();
                if (G == 16 || lg < 16) fk3_p3(G3.T3, G3.n3 >> 2, CB, H.c3_pb, L3, !rootp, pin_h3[0], pin_h3[1], pin_h3[2]);
                wave_sync();
            } else {
                fk3_run<(SPEC == 0 && G == 16)>(G3, CB, H.c3_pb, lg, G);
            }
#endif
        } else {
            fk_chain<(G >= 16), (G == 16 && !SPEC)>(H, P, CB, lg, G, true, any_grad, (a.flags & 2) != 0, n_ml_root, a.n_run_root);
        }
        // after a root fast trip the local quaternions are still where the pre-pass put them (sunk ja stores); after any
        // other trip the pre-joint quaternions have replaced them
        ql_fresh = fast_trip && tail_ok;  // (a chain whose other coordinates leave the box never takes the lite path)

        PROF_TICK(2);  // FK
        // ---- marker sites: residual, per-site loss term, per-site wrench ----------------------------
        const V3 cref = LEAN ? ld3(CB + root_w) : ld_tpos(bx + kXf);  // entry 1 = first active body (the root): moments are taken about it
        const bool trunk_w = (!a.single) && kind < 2;
        const int Kpad = (K + 3) & ~3;
        // one site: world position, weighted residual against the keypoint (kx, ky, kz), loss term; the wrench
        // (f, (x - c) x f) goes to its place in DFS-site order
        auto site_term = [&](const int k, const float kx, const float ky, const float kz, const bool tw, const int rr = -1) -> float {
            float4 sr;
            if (SITEPIN && rr >= 0) sr = pin_sr[rr < 0 ? 0 : rr];  // (called with the round: the record out of registers)
            else sr = lds4(srec + 4 * k);
            const int ss = __builtin_bit_cast(int, sr.w);
            V3 bpos_w;
            Q4 bquat_w;
            if constexpr (LEAN) {
                int s3;
                if (SITEPIN && rr >= 0) {
                    s3 = pin_s3[rr < 0 ? 0 : rr];
                    if (n_ml_root > 0) s3 = site3[k];  // (wave-uniform: a root pass ran the pruned program)
                } else {
                    s3 = site3[k];
                }
                bpos_w = ld3(CB + (s3 & 0xFFFF));
                const float4 q4 = lds4(CB + (int)((unsigned)s3 >> 16));
                bquat_w = Q4{q4.x, q4.y, q4.z, q4.w};
            } else {
                const float *bp = bx + (ss & 0xFFFF) * kXf;
                bpos_w = ld_tpos(bp);
                bquat_w = ld_tquat(bp);
            }
            const V3 sx = add3(bpos_w, rotate(V3{sr.x, sr.y, sr.z}, bquat_w));
            float w0, w1, w2;
            if (a.single) {
                w0 = a.kpw3[3 * k] ? 1.f : 0.f; w1 = a.kpw3[3 * k + 1] ? 1.f : 0.f; w2 = a.kpw3[3 * k + 2] ? 1.f : 0.f;
            } else {
                w0 = w1 = w2 = (trunk_w ? (tw ? 1.f : 0.f) : 1.f);
            }
            // a site without weight: exact zeros, written without looking at its body (which a pruned FK has skipped)
            const bool wz = (w0 == 0.0f) && (w1 == 0.0f) && (w2 == 0.0f);
            const float rx = wz ? 0.0f : (kx - sx.x) * w0, ry = wz ? 0.0f : (ky - sx.y) * w1,
                        rz = wz ? 0.0f : (kz - sx.z) * w2;
            if (any_grad) {
                const V3 f = {-2.0f * rx, -2.0f * ry, -2.0f * rz};
                V3 tq = cross3(sub3(sx, cref), f);
                if (wz) tq = V3{0.0f, 0.0f, 0.0f};
                const int sp = ss >> 16;
                st_tpos(sw + kXf * sp, f);
                st_tvec2(sw + kXf * sp, tq);
            }
            return FMA(rz, rz, FMA(ry, ry, rx * rx));
        };
        float loss;
        if (site_regs) {
            // pairwise tree over the sites (oracle: tree_sum) as lane butterflies + registers: no LDS, no barrier
            float term[NSR];
            if (lite && sites1) {
                // root fast trip: only the weighted sites have a term, at most one per lane: one round instead of NSR
                // (the others' terms are exact zeros, and nobody reads their wrench entries: see the range sum below)
                float v = 0.0f;
                if (trunk_r >= 0) {
                    const float kx = trunk_r == 0 ? kpr[0][0] : (trunk_r == 1 ? kpr[NSR > 1 ? 1 : 0][0] : kpr[NSR - 1][0]);
                    const float ky = trunk_r == 0 ? kpr[0][1] : (trunk_r == 1 ? kpr[NSR > 1 ? 1 : 0][1] : kpr[NSR - 1][1]);
                    const float kz = trunk_r == 0 ? kpr[0][2] : (trunk_r == 1 ? kpr[NSR > 1 ? 1 : 0][2] : kpr[NSR - 1][2]);
                    v = site_term(trunk_r * G + lg, kx, ky, kz, true);
                }
#pragma unroll
                for (int r = 0; r < NSR; ++r) term[r] = r == trunk_r ? v : 0.0f;
            } else {
#pragma unroll
            for (int r = 0; r < NSR; ++r) {
                const int k = r * G + lg;
                term[r] = k < K ? site_term(k, kpr[r][0], kpr[r][1], kpr[r][2], ((kpw_bits >> r) & 1u) != 0, r) : 0.0f;
            }
            }
            loss = group_tree_sum<G, NSR>(term);
            wave_sync();
            PROF_TICK(3);  // sites + loss
        } else {
        for (int k = K + lg; k < (K > 64 ? H.kpow2 : Kpad); k += G) r2[k] = 0.0f;  // zero padding of the loss tree
        for (int k = lg; k < K; k += G) r2[k] = site_term(k, kpl[3 * k], kpl[3 * k + 1], kpl[3 * k + 2], trunk_w && a.kpw[k] != 0);
        wave_sync();
        PROF_TICK(3);  // sites
        // pairwise tree over the sites (oracle: tree_sum), every lane redundantly from broadcast LDS reads:
        // r2 is padded with zeros to a multiple of 4; levels 1 and 2 inside each float4, then across them.
        {
            const int n4 = Kpad >> 2;
            if (n4 > 16) {
                // more than 64 sites: the same pairwise tree, in place in LDS (r2 is padded to a power of two)
                for (int h = 1; h < H.kpow2; h <<= 1) {
                    for (int i = lg * 2 * h; i < H.kpow2; i += G * 2 * h) r2[i] = r2[i] + r2[i + h];
                    wave_sync();
                }
                loss = r2[0];
            } else if (n4 <= 8) {
                float acc[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc[i] = 0.0f;
                    if (i < n4) {
                        const float4 q4 = lds4(r2 + 4 * i);
                        acc[i] = (q4.x + q4.y) + (q4.z + q4.w);
                    }
                }
                // adding the zero padding is exact, so the tree over 8 float4 equals the oracle's tree
                loss = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
            } else {
                float acc[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    acc[i] = 0.0f;
                    if (i < n4) {
                        const float4 q4 = lds4(r2 + 4 * i);
                        acc[i] = (q4.x + q4.y) + (q4.z + q4.w);
                    }
                }
#pragma unroll
                for (int h = 1; h < 16; h *= 2)
#pragma unroll
                    for (int i = 0; i < 16; i += 2 * h) acc[i] = acc[i] + acc[i + h];
                loss = acc[0];
            }
        }
        wave_sync();
        PROF_TICK(4);  // loss sum
        }
        // ---- gradient pass, two steps over the evaluation whose arrays start at CBx ------------------------
        // (A) subtree wrench of every DISTINCT site range: the site wrenches in (body id, site id) order, summed left to
        //     right from zero (RangeRec, stac_plan.hpp); written where the body transforms were
        const RangeRec *rrec = reinterpret_cast<const RangeRec *>(P + H.off_range);
        auto range_sum = [&](const int r, float *CBx) {
            const float *swx = CBx + H.c_sw;
            const RangeRec rr = rrec[r];
            V3 Fs = {0.f, 0.f, 0.f}, T0 = {0.f, 0.f, 0.f};
            int i = rr.lo;
            for (; i + 4 <= rr.hi; i += 4) {  // same left-to-right order; four wrenches in flight per LDS round trip
                V3 f4[4], t4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { f4[u] = ld_tpos(swx + kXf * (i + u)); t4[u] = ld_tvec2(swx + kXf * (i + u)); }
#pragma unroll
                for (int u = 0; u < 4; ++u) { Fs = add3(Fs, f4[u]); T0 = add3(T0, t4[u]); }
            }
            if (i < rr.hi) {  // the last one to three sites in one trip instead of one each (reads behind the end repeat the last site, their
                              // additions are dropped: the same sum): +0.9 % on the 10 000-frame bench
                const int last = rr.hi - 1;
                V3 f4[3], t4[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) { f4[u] = ld_tpos(swx + kXf * min(i + u, last)); t4[u] = ld_tvec2(swx + kXf * min(i + u, last)); }
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const V3 fn = add3(Fs, f4[u]), tn = add3(T0, t4[u]);
                    const bool on = i + u <= last;
                    Fs = V3{on ? fn.x : Fs.x, on ? fn.y : Fs.y, on ? fn.z : Fs.z};
                    T0 = V3{on ? tn.x : T0.x, on ? tn.y : T0.y, on ? tn.z : T0.z};
                }
            }
            st_tpos(CBx + H.c_rw + kXf * r, Fs);
            st_tvec2(CBx + H.c_rw + kXf * r, T0);
        };
        // With 32 or 64 lanes on one evaluation (latency mode) the sums are split further: one task = one of the six
        // components {F, T} of one range -- the same left-to-right sums, but a lane's loop is one LDS read and one add per
        // site instead of four reads and six adds, and the six tasks of a range sit on neighbouring lanes (equal trip
        // counts; the ranges are sorted longest first).  The longest range -- every site, for the root's joint -- sets
        // the length of this phase.  (With 16 lanes the extra rounds cost more than they save: measured -8 %.)
        // One component of a range's sum in the latency kernels: the n values from p0 on (a site every kXf words), left to right from zero,
        // RT values per trip.  The values behind the range's end are READ (one address register and immediate offsets instead of a
        // clamped index per value: the words behind a chain's last site are the chain's own slack, the next chain's region or the slack
        // behind the workgroup's last chain -- spec_lds_bytes) and their additions dropped: the same sum.
        auto range_acc = [&](auto rt_c, const float *p0, const int n) -> float {
            constexpr int RT = decltype(rt_c)::value;
            float acc = 0.f;
            const float *p = p0;
            for (int m = n; m > 0; m -= RT, p += kXf * RT) {  // (m: values left -- one compare against a constant per value)
                float v[RT];
#pragma unroll
                for (int u = 0; u < RT; ++u) v[u] = p[kXf * u];
#pragma unroll
                for (int u = 0; u < RT; ++u) acc = u < m ? acc + v[u] : acc;
            }
            return acc;
        };
        // by the longest range of the round (wave-uniform): four, twelve or twenty-four values per trip
        auto range_acc_by = [&](const int maxn, const float *p0, const int n) -> float {
            if (maxn <= 4) return range_acc(std::integral_constant<int, 4>{}, p0, n);
            if (maxn <= STAC_RT) return range_acc(std::integral_constant<int, STAC_RT>{}, p0, n);
            return range_acc(std::integral_constant<int, 24>{}, p0, n);
        };
        // (a task out of registers -- latency kernels, the lane's first tasks: LATPIN; maxn: the round's longest range)
        auto range_task_pinned = [&](const auto &tp, float *CBx, const int maxn) {
            CBx[tp.dst] = range_acc_by(maxn, CBx + tp.src + kXf * tp.lo, tp.hi - tp.lo);
        };
        auto range_task = [&](const int t, float *CBx) {
            const int r = t / 6, k = t - 6 * r;
            const RangeRec rr = rrec[r];
            const int co = k < 3 ? k : kXq + k - 3;
            const float *src = CBx + H.c_sw + co;
            float acc = 0.f;
            if constexpr (SPEC != 0) {
                // (latency kernels: a lone wavefront waits out every LDS round trip, so twelve sites per trip and no remainder loop)
                acc = range_acc(std::integral_constant<int, STAC_RT>{}, src + kXf * rr.lo, rr.hi - rr.lo);
            } else {
                int i = rr.lo;
                for (; i + 4 <= rr.hi; i += 4) {  // four in flight per LDS round trip
                    const float v0 = src[kXf * i], v1 = src[kXf * (i + 1)], v2 = src[kXf * (i + 2)], v3 = src[kXf * (i + 3)];
                    acc = acc + v0; acc = acc + v1; acc = acc + v2; acc = acc + v3;
                }
                if (i < rr.hi) {  // the last one to three sites in one trip (as range_sum)
                    const int last = rr.hi - 1;
                    const float v0 = src[kXf * i], v1 = src[kXf * min(i + 1, last)], v2 = src[kXf * min(i + 2, last)];
                    acc = acc + v0;
                    acc = i + 1 <= last ? acc + v1 : acc;
                    acc = i + 2 <= last ? acc + v2 : acc;
                }
            }
            CBx[H.c_rw + kXf * r + co] = acc;
        };
        // (B) one joint: its range's wrench, then the joint formulas; crefx = the root position the moments refer to
        // (lean kernels: the joints this is called for -- all but the free root, which has its own path -- are hinges: the host has
        //  checked that the model has no ball joint and that its FK program is uniform, launch_q_phase)
        constexpr bool LEAN_HINGES = LEAN && SPEC == 0;  // (the latency kernels pass the free root through here as well)
        auto joint_gradient = [&](const int j, float *CBx, const V3 crefx, float *ggx, const bool pinned = false) {
            const float *jax_ = CBx + H.c_ja, *qsvx = CBx + H.c_qsv, *jnx = CBx + H.c_jn;
            const float *jr = jrec + 12 * j;
            int ty = JHINGE, ad;
            float4 ja4;
            const float *rw;
            if (LATPIN && pinned) {  // (the lane's joint of the first round: its record out of registers)
                ad = pin_g.ad;
                ja4 = float4{pin_g.ax, pin_g.ay, pin_g.az, 0.0f};
                rw = CBx + pin_g.rw;
            } else {
                const int4 ji = lds4i(jr);  // type, qadr, slo, shi
                ty = ji.x; ad = ji.y;
                ja4 = lds4(jr + 8);  // axis, range id
                rw = CBx + H.c_rw + kXf * __builtin_bit_cast(int, ja4.w);
            }
            const V3 Fs = ld_tpos(rw), T0 = ld_tvec2(rw);
                V3 anchor;
                Q4 prequat;
                if constexpr (LEAN) {  // (split kinematics: where the joint's anchor and its pre-joint quaternion are, from the full program's joint words)
                    const int jw = (LATPIN && pinned) ? pin_g.jw : reinterpret_cast<const int *>(P + H.off3_site)[K + j];
                    anchor = ld3(CBx + (jw & 0xFFFF));
                    const float4 q4 = lds4(CBx + (int)((unsigned)jw >> 16));
                    prequat = Q4{q4.x, q4.y, q4.z, q4.w};
                } else {
                    anchor = ld_tpos(jax_ + kXf * j);
                    prequat = ld_tquat(jax_ + kXf * j);
                }
                const V3 tau = sub3(T0, cross3(sub3(anchor, crefx), Fs));
                // (lean kernels know the types: joint 0 is the free root, every other one a hinge)
                const bool is_hinge = LEAN_HINGES ? true : (LEAN ? j != 0 : ty == JHINGE);
                if (is_hinge) {
                    ggx[ad] = dot3(rotate(V3{ja4.x, ja4.y, ja4.z}, prequat), tau);
                } else if (!LEAN && ty == JSLIDE) {
                    ggx[ad] = dot3(rotate(V3{ja4.x, ja4.y, ja4.z}, prequat), Fs);
                } else {
                    int qa = ad;
                    V3 tl = tau;
                    if (LEAN || ty == JFREE) {
                        st3(ggx + ad, Fs);
                        qa = ad + 3;
                    } else {
                        tl = rotate(tau, Q4{prequat.w, -prequat.x, -prequat.y, -prequat.z});
                    }
                    const int qord = LEAN ? 0 : __builtin_bit_cast(int, lds4(jr + 4).w);  // ordinal among the quaternion joints
                    const Q4 qh = ld4(qsvx + 4 * qord);  // saved by the pre-pass
                    const V3 u = {qh.x, qh.y, qh.z};
                    const V3 uxt = cross3(u, tl);
                    const float n = jnx[qord];
                    const float dn = n + (n == 0.0f ? 1e-6f : 0.0f);
                    ggx[qa] = (-2.0f * dot3(tl, u)) / dn;
                    ggx[qa + 1] = (2.0f * FMA(qh.w, tl.x, -uxt.x)) / dn;
                    ggx[qa + 2] = (2.0f * FMA(qh.w, tl.y, -uxt.y)) / dn;
                    ggx[qa + 3] = (2.0f * FMA(qh.w, tl.z, -uxt.z)) / dn;
                }
        };
        // The gradient of a free joint at qpos 0 .. 6 (active joint 0; QArgs::root_free / free0p), component lg on lane lg
        // < 7: the formulas of joint_gradient's free branch -- same operations, same order per component -- but the four
        // divisions side by side instead of one lane doing all seven components while the others wait.
        auto free0_gradient = [&](float *CBx, const V3 crefx, const float *rw, const int qord, const int lg) -> float {  // (lg: the component)
            const V3 Fs = ld_tpos(rw), T0 = ld_tvec2(rw);
            const V3 anchor = LEAN ? ld3(CBx + root_w) : ld_tpos(CBx + H.c_ja);
            const V3 tau = sub3(T0, cross3(sub3(anchor, crefx), Fs));
            const Q4 qh = ld4(CBx + H.c_qsv + 4 * qord);
            const V3 u = {qh.x, qh.y, qh.z};
            const V3 uxt = cross3(u, tau);
            const float n = CBx[H.c_jn + qord];
            const float dn = n + (n == 0.0f ? 1e-6f : 0.0f);
            const float tl_i = lg == 4 ? tau.x : (lg == 5 ? tau.y : tau.z);
            const float ux_i = lg == 4 ? uxt.x : (lg == 5 ? uxt.y : uxt.z);
            const float num = lg == 3 ? -2.0f * dot3(tau, u) : 2.0f * FMA(qh.w, tl_i, -ux_i);
            const float gq = num / dn;
            const float gf = lg == 0 ? Fs.x : (lg == 1 ? Fs.y : Fs.z);
            return lg < 3 ? gf : gq;
        };
        float gnew[NQR];
#pragma unroll
        for (int r = 0; r < NQR; ++r) gnew[r] = 0.f;
        // (latency mode: a speculative trip computes gradients only for the two evaluations it ends up using, below)
        if (any_grad && fast_trip) {
            // root fast trip: the root joint's subtree wrench over the weighted sites only (in the order of the full sum),
            // one component per lane, then the root joint's gradient
            const int rid0 = __builtin_bit_cast(int, jrec[11]);
            if (lg < 6) {
                const int co = lg < 3 ? lg : kXq + lg - 3;
                const float *src = sw + co;
                float acc = 0.f;
                for (uint32_t m = a.root_trunk_lo; m; m &= m - 1) acc = acc + src[kXf * __builtin_ctz(m)];
                for (uint32_t m = a.root_trunk_hi; m; m &= m - 1) acc = acc + src[kXf * (32 + __builtin_ctz(m))];
                CB[(LEAN ? H.c3_rw0 : H.c_rw + kXf * rid0) + co] = acc;  // (lean: beside the joint-local quaternions, which a fast trip keeps)
            }
            wave_sync();
            PROF_TICK(5);  // range sums
            if (a.free0p) {
                const float gv = free0_gradient(CB, cref, CB + (LEAN ? H.c3_rw0 : H.c_rw + kXf * rid0), a.free0p - 1, lg);
                if (lg < 7 && (mbits & 1u)) gnew[0] = gv;
            } else {
                for (int j = lg; j < a.n_root_joints; j += G) joint_gradient(j, CB, cref, gg);
                wave_sync();
                if (lg < a.root_fast && (mbits & 1u)) gnew[0] = gg[lg];
            }
            wave_sync();
            PROF_TICK(6);  // joint gradients
        } else if (any_grad && !(SPEC && st_in == ST_SPEC)) {
            // the range sums go where the body transforms were (the site pass, their last reader, is over; cref is in a
            // register), the gradient where the site wrenches were (dead once the range sums are done)
            if constexpr (G >= 32) { for (int t = lg; t < 6 * H.nrange; t += G) range_task(t, CB); }
            else if constexpr (LEAN && SPEC == 0 && G == 16) {
                // the longest ranges one component per lane, the others one range per lane (PlanHeader::rsplit: the rodent's 23-site
                // range of the root joint is 23 reads and additions on each of six lanes instead of 138 of each on one)
                const int ns = H.rsplit;
                for (int t = lg; t < 6 * ns; t += G) range_task(t, CB);
                for (int r = ns + lg; r < H.nrange; r += G) range_sum(r, CB);
            }
            else { for (int r = lg; r < H.nrange; r += G) range_sum(r, CB); }
            wave_sync();
            if constexpr (SPEC != 0) {  // (latency kernels: the bootstrap evaluation of a solve is rare; they keep the zeroed vector)
                for (int e = lg; e < nqpad; e += G) gg[e] = 0.0f;
                wave_sync();
            }
            PROF_TICK(5);  // range sums
            const int naj_g = n_ml_root > 0 ? a.n_root_joints : H.naj;  // pruned root-pass trip: only the root's joints
            if (!LEAN && (a.flags & 16) && j0 == 1) {
                // (launch-wide: every joint but the free root is a hinge -- QArgs::flags bit 4: the hinge formula without the dispatch)
                for (int j = lg + 1; j < naj_g; j += G) {
                    const float *jr = jrec + 12 * j;
                    const int ad = reinterpret_cast<const int *>(jr)[1];
                    const float4 ja4 = lds4(jr + 8);  // axis, range id
                    const float *rw = CB + H.c_rw + kXf * __builtin_bit_cast(int, ja4.w);
                    const V3 Fs = ld_tpos(rw), T0 = ld_tvec2(rw);
                    const V3 anchor = ld_tpos(ja + kXf * j);
                    const Q4 prequat = ld_tquat(ja + kXf * j);
                    const V3 tau = sub3(T0, cross3(sub3(anchor, cref), Fs));
                    gg[ad] = dot3(rotate(V3{ja4.x, ja4.y, ja4.z}, prequat), tau);
                }
            } else {
                if constexpr (LEAN && SPEC == 0) {
                    // three rounds of lanes at once, as in the pre-pass (a lane without a joint left repeats the last one)
                    constexpr int U = G == 16 ? 3 : 2;
                    for (int jb = lg + j0; jb < naj_g; jb += U * G) {
#pragma unroll
                        for (int u = 0; u < U; ++u) joint_gradient(min(jb + u * G, naj_g - 1), CB, cref, gg);
                    }
                } else {
                    for (int j = lg + j0; j < naj_g; j += G) joint_gradient(j, CB, cref, gg);
                }
            }
            wave_sync();
            const uint32_t abits = SPEC ? mbits : ((LEAN ? act_bits : MB[nkinds * G + lg]) & mbits);  // optimised coordinates that HAVE a gradient entry
#pragma unroll
            for (int r = 0; r < NQR; ++r) {
                const int e = r * G + lg;
                if constexpr (LEAN) {  // (read, then select: no branch per register; gg has nqpad + 7 words)
                    const float gv = gg[e < nq ? e : 0];
                    gnew[r] = (e < nq && ((abits >> r) & 1u)) ? gv : gnew[r];
                } else {
                    if (e < nq && ((abits >> r) & 1u)) gnew[r] = gg[e];
                }
            }
            if (a.free0p) {  // the free root joint: one component per lane, its four divisions side by side
                const float gv = free0_gradient(CB, cref, CB + H.c_rw + kXf * __builtin_bit_cast(int, jrec[11]), a.free0p - 1, lg);
                if (lg < 7 && (mbits & 1u)) gnew[0] = gv;
            }
            wave_sync();
            PROF_TICK(6);  // joint gradients
        }

        // ---- solver transitions (jaxopt ProjectedGradient) ------------------------------------
        bool ending = false;
        float sum0 = 0.0f, sum1 = 0.0f;
        if (lite) {
            // Root fast trip: every live chain is in a root solve whose other coordinates sit inside the box (tail_ok), so
            // registers 1 .. NQR - 1 hold g = 0 and x = y: their terms of every norm are exact zeros and the accepted point
            // leaves them where they are.  Same operations as below on register 0; the trees over the registers reduce to
            // "+ 0" (a sum of squares / products plus zeros: one addition of +0 is all that the zero registers do to it).
            const bool in0 = lg < nq;
            if (st_in == ST_VG_Y) {
                fy = loss; eta = stepsize; nls = 0;
                g[0] = gnew[0];
                c_grad++;
                st = ST_LS;
            }
            float a0 = 0.0f, a1 = 0.0f;
            if (in0) {
                if (st_in == ST_VG_X) {
                    const float d = clipf(x[0] - gnew[0], LB(0, lg), UB(0, lg)) - x[0];
                    a0 = d * d;
                } else if (st_in == ST_LS) {
                    const float d = CAND(0, lg) - y[0];
                    a0 = d * d;
                    a1 = d * g[0];
                }
            }
            const float one0[1] = {a0}, one1[1] = {a1};
            sum0 = group_tree_sum<G, 1>(one0) + 0.0f;
            sum1 = group_tree_sum<G, 1>(one1) + 0.0f;
            PROF_TICK(7);  // transition terms + sums
            bool fused = false;
            if (st_in == ST_LS) {
                c_ls++;
                const float lhs = eta * (loss - fy);
                const float rhs = eta * sum1 + 0.5f * sum0 + eps;
                bool accept = !(lhs > rhs);
                const bool evaluated_point_accepted = accept;
                if (!accept) {
                    eta = eta * 0.5f;
                    nls++;
                    if (nls >= a.maxls) accept = true;
                }
                if (accept) {
                    const float cr = in0 ? CAND(0, lg) : x[0];
                    const float d = cr - x[0];
                    y[0] = FMA(spec_beta, d, cr);
                    x[0] = cr;
                    st = ST_VG_X;
                    fused = ls_with_grad && evaluated_point_accepted;
                }
            }
            if (__any(fused)) {
                float f0 = 0.0f;
                if (in0) {
                    const float d = clipf(x[0] - gnew[0], LB(0, lg), UB(0, lg)) - x[0];
                    f0 = d * d;
                }
                const float onef[1] = {f0};
                const float e2 = group_tree_sum<G, 1>(onef) + 0.0f;
                if (fused) sum0 = e2;
            }
            if (st_in == ST_VG_X || fused) {
                fx = loss;
                error = __builtin_sqrtf(sum0);
                stepsize = (eta <= 1e-6f) ? 1.0f : eta / 0.5f;
                t = spec_tn;
                iter++;
                c_grad++;
                if (error > a.tol && iter < a.maxiter) st = ST_VG_Y;
                else ending = true;
            }
        } else {
        if (st_in == ST_VG_Y) {
            fy = loss;
            eta = stepsize;
            nls = 0;
#pragma unroll
            for (int r = 0; r < NQR; ++r) g[r] = gnew[r];
            c_grad++;
            st = SPEC ? ST_SPEC : ST_LS;
        }
        // nq-sums as pairwise trees over the striped registers (oracle: tree_sum); all groups compute
        // them every trip (a few dozen DPP adds), only the groups in the matching state use them.  Straight-line: the
        // bounds come in one LDS round trip for the whole transition, both candidates of a term are computed and one is selected.
        float lbs[NQR], ubs[NQR];
#pragma unroll
        for (int r = 0; r < NQR; ++r) {
            const int e = r * G + lg, eb = e < nq ? e : 0;
            lbs[r] = LB(r, eb);
            ubs[r] = UB(r, eb);
        }
        {
            const float eta_s = (SPEC && st_in == ST_SPEC) ? eta * spec_pow : eta;
            const bool isx = SPEC == 0 && st_in == ST_VG_X, isl = SPEC ? st_in == ST_SPEC : st_in == ST_LS;
            float t0[NQR], t1[NQR];
#pragma unroll
            for (int r = 0; r < NQR; ++r) {
                const int e = r * G + lg;
                const float dx = clipf(x[r] - gnew[r], lbs[r], ubs[r]) - x[r];
                const float dl = clipf(FMA(-eta_s, g[r], y[r]), lbs[r], ubs[r]) - y[r];
                const float d = isx ? dx : dl;
                const bool on = e < nq && (isx || isl);
                t0[r] = on ? d * d : 0.0f;
                t1[r] = (on && isl) ? d * g[r] : 0.0f;
            }
            sum0 = group_tree_sum<G, NQR>(t0);
            sum1 = group_tree_sum<G, NQR>(t1);
        }
        PROF_TICK(7);  // transition terms + nq sums
        bool fused = false;  // accepted a candidate whose gradient is already in gnew
        if (SPEC == 0 && st_in == ST_LS) {
            c_ls++;
            const float lhs = eta * (loss - fy);
            const float rhs = eta * sum1 + 0.5f * sum0 + eps;
            bool accept = !(lhs > rhs);
            const bool evaluated_point_accepted = accept;
            if (!accept) {
                eta = eta * 0.5f;
                nls++;
                if (nls >= a.maxls) accept = true;  // taken without evaluation, like jaxopt's loop bound
            }
            if (accept) {
                const float beta = spec_beta;  // (t - 1) / t_next of the running iteration
#pragma unroll
                for (int r = 0; r < NQR; ++r) {
                    const int e = r * G + lg;
                    const float cr = e < nq ? clipf(FMA(-eta, g[r], y[r]), lbs[r], ubs[r]) : x[r];
                    const float d = cr - x[r];
                    y[r] = FMA(beta, d, cr);
                    x[r] = cr;
                }
                st = ST_VG_X;
                fused = ls_with_grad && evaluated_point_accepted;
            }
        }
        // stopping residual of x_next: either this trip evaluated x (VG_X) or the accepted candidate
        // came with its gradient (fused)
        if (SPEC == 0 && __any(fused)) {
            float t0[NQR];
#pragma unroll
            for (int r = 0; r < NQR; ++r) {
                const int e = r * G + lg;
                const float d = clipf(x[r] - gnew[r], lbs[r], ubs[r]) - x[r];
                t0[r] = e < nq ? d * d : 0.0f;
            }
            const float e2 = group_tree_sum<G, NQR>(t0);
            if (fused) sum0 = e2;
        }
        if (SPEC == 0 && (st_in == ST_VG_X || fused)) {
            fx = loss;
            error = __builtin_sqrtf(sum0);
            stepsize = (eta <= 1e-6f) ? 1.0f : eta / 0.5f;  // eta still holds the accepted step
            t = spec_tn;
            iter++;
            c_grad++;
            if (error > a.tol && iter < a.maxiter) st = ST_VG_Y;
            else ending = true;
        }


        }
        if (SPEC && st_in == ST_SPEC) {  // every lane of the chain is in this state: the solver state is replicated
            // (1) which candidate would the sequential line search take?  Candidate n = nls + c is evaluated
            //     only while n < maxls; candidate n == maxls is taken without evaluation (jaxopt's loop bound).
            const float ec = eta * spec_pow;
            const bool ok = !(ec * (loss - fy) > ec * sum1 + 0.5f * sum0 + eps);  // sufficient decrease (own candidate)
            const bool take = role < NC && (ok || nls + role >= a.maxls);
            float *xc = XB + trip_parity * 32;
            if (lg == 0) { xc[4 * role] = take ? 1.0f : 0.0f; xc[4 * role + 1] = loss; }
            PROF_TICK(8);
            chain_sync();
            PROF_TICK(4);  // latency mode: wait for the other roles' losses
            int cs = -1;  // first accepting role = c*
#pragma unroll
            for (int c = NC - 1; c >= 0; --c)
                if (xc[4 * c] != 0.0f) cs = c;
            if (cs < 0) {  // none of the NC candidates: halve NC more times
                eta = eta * (NC == 4 ? 0.0625f : 0.25f);
                nls += NC;
                c_ls += NC;
            } else {
                const int evaluated = (nls + cs >= a.maxls) ? cs : cs + 1;
                const float pw = cs == 0 ? 1.0f : cs == 1 ? 0.5f : cs == 2 ? 0.25f : 0.125f;
                const float eacc = eta * pw;
                // (2) Only two of the eight evaluations need their gradient: the accepted candidate (role c*, for the
                //     stopping residual) and the momentum point it leads to (role 4 + c*, the next iteration's grad f).
                //     The wave(s) owning them run the joint pass with all 64 lanes; every role then reads both vectors.
                float *gxa = XB + 64 + trip_parity * 2 * nqpad, *gxn = gxa + nqpad;
                float *CBa = CBw + cs * H.chain_stride, *CBn = CBw + (NC + cs) * H.chain_stride;
                if constexpr (NW == 1) {  // the LC lanes of this wavefront that work on the chain share both gradient passes
                    const int ll = lane % LC;
                    const V3 crefa = LEAN ? ld3(CBa + root_w) : ld_tpos(CBa + H.c_bx + kXf), crefn = LEAN ? ld3(CBn + root_w) : ld_tpos(CBn + H.c_bx + kXf);
                    for (int e = ll; e < nqpad; e += LC) { gxa[e] = 0.0f; gxn[e] = 0.0f; }
                    // (the longest ranges one component per lane, as in the throughput kernels: PlanHeader::rsplit)
                    const int nt = 6 * H.rsplit, nw = H.nrange - H.rsplit;
                    for (int i = ll; i < 2 * nt; i += LC) {
                        const bool nx = i >= nt;
                        range_task(nx ? i - nt : i, nx ? CBn : CBa);
                    }
                    for (int i = ll; i < 2 * nw; i += LC) {
                        const bool nx = i >= nw;
                        range_sum(H.rsplit + (nx ? i - nw : i), nx ? CBn : CBa);
                    }
                    wave_sync();
                    if constexpr (LATPIN) {
                        if (ll < 2 * H.naj) {  // first round: the lane's joint out of registers
                            const bool nx = ll >= H.naj;
                            joint_gradient(pin_g.j, nx ? CBn : CBa, nx ? crefn : crefa, nx ? gxn : gxa, true);
                        }
                    }
                    for (int i = ll + (LATPIN ? LC : 0); i < 2 * H.naj; i += LC) {
                        const bool nx = i >= H.naj;
                        joint_gradient(nx ? i - H.naj : i, nx ? CBn : CBa, nx ? crefn : crefa, nx ? gxn : gxa);
                    }
                } else {
                    // Two waves per evaluation: the owner of the role takes every range but the longest and the joints that
                    // read them, its neighbour (wave ^ 1: a wave of the same half of the roles) the longest range -- all
                    // sites, for the root -- and the joints on it (the free joint's formulas are the long ones).  No range or
                    // gradient entry is shared between the two, so they only meet at the barrier that follows anyway.
                    // (TWO wavefronts per chain -- eight roles of 16 lanes --: wave 0 holds the candidates, wave 1 the momentum points,
                    //  each runs the whole gradient pass of its one evaluation)
                    constexpr bool SOLO = NW == 2;
                    const int wa = cs / CPW, wn = (NC + cs) / CPW;  // wave-uniform
                    const bool mine = wa == wave || wn == wave, help = !SOLO && (wa == (wave ^ 1) || wn == (wave ^ 1));
                    if (mine || help) {
                        const bool first = mine ? wa == wave : wa == (wave ^ 1);
                        float *gx = first ? gxa : gxn, *CBx = first ? CBa : CBn;
                        const V3 crefx = LEAN ? ld3(CBx + root_w) : ld_tpos(CBx + H.c_bx + kXf);
                        if constexpr (LATPIN) {  // (the lane's first two tasks / the helper's one out of registers)
                            const int base = SOLO ? 0 : 6;
                            if (mine) {
                                if (base + lane < 6 * H.nrange) range_task_pinned(pin_t[0], CBx, pin_t[0].maxn);
                                if (base + lane + 64 < 6 * H.nrange) range_task_pinned(pin_t[1], CBx, pin_t[1].maxn);
                                for (int t = base + lane + 128; t < 6 * H.nrange; t += 64) range_task(t, CBx);
                            } else if (lane < 6) range_task_pinned(pin_t[2], CBx, pin_t[2].maxn);
                        } else {
                        if (mine) { for (int t = (SOLO ? 0 : 6) + lane; t < 6 * H.nrange; t += 64) range_task(t, CBx); }
                        else if (lane < 6) range_task(lane, CBx);
                        }
                        wave_sync();
#if defined(STAC_PROFILE) && defined(STAC_PROF_GRAD)  // (diagnostic: the range sums of the latency kernels' gradient pass charged to stamp 9)
                        PROF_TICK(9);
#endif
                        if constexpr (LATPIN) {
                            // (the free root -- lean kernels: joint 0, qpos 0 .. 6, range 0 --: one component per lane, its four divisions
                            //  side by side, where the wavefront that takes it has nothing else in this round: the neighbour wavefront of
                            //  the four-wavefront kernels, whose other lanes idle)
                            const bool free0_lanes = !SOLO && !mine;
                            if (free0_lanes && lane < 7) gx[lane] = free0_gradient(CBx, crefx, CBx + H.c_rw, 0, lane);
                            if (lane < H.naj) {  // first round: the lane's joint out of registers
                                const bool on_longest = pin_g.rw == H.c_rw;  // (range 0)
                                if ((SOLO || on_longest != mine) && !(free0_lanes && pin_g.j == 0)) joint_gradient(pin_g.j, CBx, crefx, gx, true);
                            }
                        }
                        for (int j = lane + (LATPIN ? 64 : 0); j < H.naj; j += 64) {
                            const bool on_longest = __builtin_bit_cast(int, jrec[12 * j + 11]) == 0;
                            if (SOLO || on_longest != mine) joint_gradient(j, CBx, crefx, gx);
                        }
                    }
                }
                PROF_TICK(5);  // latency mode: gradient pass of the two chosen evaluations (owner waves)
                chain_sync();
                PROF_TICK(6);  // latency mode: wait for the gradients
                // (straight-line: both vectors are read at a valid index and selected, the accepted point is computed once -- the branches
                //  round every register's loads and the second evaluation of the clip cost a lone wavefront ~50 instructions per trip)
                float gnext[NQR], xacc[NQR];
#pragma unroll
                for (int r = 0; r < NQR; ++r) {
                    const int e = r * G + lg;
                    const bool in = e < nq, on = in && ((mbits >> r) & 1u);
                    const float ga = gxa[in ? e : 0], gb = gxn[in ? e : 0];
                    gnew[r] = on ? ga : 0.0f;
                    gnext[r] = on ? gb : 0.0f;
                }
                float t0[NQR];
#pragma unroll
                for (int r = 0; r < NQR; ++r) {
                    const int e = r * G + lg;
                    xacc[r] = clipf(FMA(-eacc, g[r], y[r]), LB(r, e), UB(r, e));
                    const float d = clipf(xacc[r] - gnew[r], LB(r, e), UB(r, e)) - xacc[r];
                    t0[r] = e < nq ? d * d : 0.0f;
                }
                const float e2 = group_tree_sum<G, NQR>(t0);  // the same value in every role
                const float fx_c = xc[4 * cs + 1];
                // (3) f at the next momentum point comes from role 4 + c*
                const float fy_next = xc[4 * (NC + cs) + 1];
                c_ls += evaluated;
                c_grad += 1;  // the gradient at x_next (the oracle's VG_X evaluation)
                const float next_step = (eacc <= 1e-6f) ? 1.0f : eacc / 0.5f;
#pragma unroll
                for (int r = 0; r < NQR; ++r) {
                    const int e = r * G + lg;
                    const float cr = e < nq ? xacc[r] : x[r];
                    const float d = cr - x[r];
                    y[r] = FMA(spec_beta, d, cr);
                    x[r] = cr;
                    g[r] = gnext[r];
                }
                fx = fx_c;
                error = __builtin_sqrtf(e2);
                stepsize = next_step;
                t = spec_tn;
                iter++;
                fy = fy_next;
                eta = stepsize;
                nls = 0;
                if (!(error > a.tol && iter < a.maxiter)) ending = true;
                else { c_grad += 1; }  // f, grad f at the next y: already evaluated (role 4 + c*)
            }
            trip_parity ^= 1;
        }

        PROF_TICK(8);  // accept / fused residual
        // ---- end of a solve: replace_qs (utils.py:147-169), next solve / next frame ------------------------
        if (__any(ending)) {
            // (cold: once per solve.  What it needs of the launch is read here, whatever the trip keeps in registers)
            KQArgs *ak_c = ak_base;
            asm volatile("" : "+s"(ak_c));
#ifndef STAC_NO_PIN
            TripArgs a_c = trip_args<kVPin>(hot_a, *ak_c);
            if constexpr (LEAN) { a_c.single = 0; a_c.flags = 0; a_c.free0p = 1; }
            const TripArgs &a = a_c;
            const TripHeader H = trip_header<kVPin>(hot_h, ak_c->h);
#else
            KQArgs &a = *ak_c;
            const auto &H = a.h;
#endif
            const int nq = H.nq, K = H.K;
            const int *const quat_adr = reinterpret_cast<const int *>(P + H.off_quat_adr);
            const TripCtx cx{lg, nq, K, CB + H.c_kp, P + H.off_lb, P + H.off_ub, P + H.off_qpos0,
                             reinterpret_cast<const uint32_t *>(lds + ((H.total_words - H.plan_skip + 3) & ~3))};
            if (ending) {
#pragma unroll
                for (int r = 0; r < NQR; ++r) {
                    const int e = r * G + lg;
                    if (e < nq) {
                        const float mi = ((mbits >> r) & 1u) ? 1.0f : 0.0f;
                        // full-body solve: qpos <- params; root / part solves: make_qs(q0, mask, params)
                        const bool blend = a.single ? false : (kind != 2);
                        qe[e] = blend ? ((1.0f - mi) * q0[r] + mi * x[r]) : x[r];
                    }
                }
            }
            wave_sync();
            if (ending && !a.single) {  // kinematics normalises quaternions in qpos
                for (int qi = lg; qi < H.nquat; qi += G) {
                    const int ad = quat_adr[qi];
                    float n;
                    st4(qe + ad, normalize4(ld4(qe + ad), &n));
                }
            }
            wave_sync();
            if (ending) {
                c_iter += iter;
                c_solves++;
                if (a.single) {
                    for (int e = lg; e < nq; e += G) a.qpos_out[(size_t)chain * nq + e] = qe[e];
                    if (lg == 0) {
                        float *so = a.err_out + (size_t)chain * 4;
                        so[0] = error; so[1] = stepsize; so[2] = t; so[3] = fx;
                        if (a.counters_out) {
                            uint32_t *co = a.counters_out + (size_t)chain * 4;
                            co[0] = (uint32_t)iter; co[1] = c_ls; co[2] = c_grad; co[3] = 1u;
                        }
                    }
                    st = ST_DONE;
                } else {
#pragma unroll
                    for (int r = 0; r < NQR; ++r) {
                        const int e = r * G + lg;
                        if (e < nq) q0[r] = qe[e];
                    }
                    if (kind < 2) {
                        // root optimisation is not part of the frame's counters (compute_stac.py:17-104)
                        c_iter = c_ls = c_grad = c_solves = 0;
                    }
                    const bool was_root = kind < 2;
                    kind++;
                    if (kind > a.P + 2) {  // frame finished: record it (compute_stac.py:261-267)
                        const size_t fo = (size_t)chain * a.F + frame;
                        if (!SPEC || role == 0)
                            for (int e = lg; e < nq; e += G) a.qpos_out[fo * nq + e] = qe[e];
                        if (lg == 0 && (!SPEC || role == 0)) {
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
                            if (a.q_carry_out && (!SPEC || role == 0))
                                for (int e = lg; e < nq; e += G) a.q_carry_out[(size_t)chain * nq + e] = qe[e];
                            st = ST_DONE;
                            if (a.ctl && !a.resume) {
                                if (!SPEC && lg == 0) atomicAdd(a.ctl, 1);  // finished chains (hand-off threshold)
                                if (a.queue_slots > 0) {  // chain queue: take the next unstarted chain (latency mode: all
                                    int nxt = 0;          // eight roles of the chain move on together)
                                    if constexpr (SPEC && NW > 1) {
                                        int *xq = reinterpret_cast<int *>(XB + 64 + 4 * nqpad);
                                        if (threadIdx.x == 0) xq[0] = atomicAdd(a.ctl + 4, 1);
                                        __syncthreads();
                                        nxt = xq[0];
                                        __syncthreads();
                                    } else if (!SPEC && a.root_fast > 0) {
                                        st = ST_NEXT;  // with its wavefront's other groups, at the top of the loop
                                        nxt = a.C;
                                    } else {
                                        if (SPEC ? (lane % LC) == 0 : lg == 0) nxt = atomicAdd(a.ctl + 4, 1);
                                        nxt = __shfl(nxt, SPEC ? (lane / LC) * LC : grp * G, 64);
                                    }
                                    if (nxt < a.C) begin_chain(a, cx, queue_chain(a, nxt));
                                }
                            }
                        } else {
                            load_kp(a, cx, kp_chain + (size_t)frame * 3 * K);
                        }
                    } else if (kind < 2) {  // second root pass: seed the translation again (compute_stac.py:80-81)
#pragma unroll
                        for (int r = 0; r < NQR; ++r) {
                            const int e = r * G + lg;
                            if (e < 3) q0[r] = a.kp[kp_chain + 3 * a.root_kp_idx + e];  // root passes run on frame 0
                        }
                    }
                    if (st != ST_DONE && (SPEC || st != ST_NEXT)) {
#pragma unroll
                        for (int r = 0; r < NQR; ++r) { x[r] = q0[r]; y[r] = q0[r]; }
                        stepsize = 1.0f; t = 1.0f; iter = 0;
                        error = __builtin_inff();
                        st = ST_VG_Y;
                        if (!SPEC && a.root_fast > 0 && kind < 2) check_tail(a, cx);  // the next root solve starts from q0 again
                        // root solves over: the pose solves begin when the wavefront's other chains are there too (top of the loop)
                        if (!SPEC && a.root_fast > 0 && was_root && kind == 2) st = ST_WAIT;
                    }
                }
            }
            mbits_c = cx.MB[kind * G + lg];  // (the next solve's bits: kind has moved on for the lanes that ended one)
            wave_sync();
        }
        PROF_TICK(9);  // end of solve
        PROF_ROOT(lite);
    }
    PROF_FLUSH(a);
}

// ------------------------------------------------------------------------------------------------
// stand-alone forward kinematics: one lane per pose, whole body tree (utils.kinematics)
// ------------------------------------------------------------------------------------------------
// A lane walks the bodies of ITS pose in index order (parents before children) and keeps the current body's transform in
// registers: a child that follows its parent directly -- the chains that make up most of every tree (the rodent's tail: 26
// bodies) -- reads nothing.  A body with a child further down the list parks its transform in one of a few LDS slots (7 words per
// lane, lane-private columns: no bank conflicts, no synchronisation; the host allots the slots like registers, FkTables in
// stac_abi.hip: rodent 3, mouse 4: 1.8 KB per slot and wavefront).  The model tables are packed records (16 words per body, 12
// per joint) read through the SCALAR cache: every lane of a wavefront is at the same body, so body offsets, joint axes and site
// offsets arrive in SGPRs and cost no vector memory traffic at all.  The coordinate of the joint two visits AHEAD is requested
// before the current body's results are stored (VMEM counts loads and stores in one in-order counter on gfx9: a load issued
// after a store waits for that store's acknowledgement).
// Body transforms leave through an LDS staging row per lane (kFkChunk bodies: 56 words, odd stride; 14.6 KB per wavefront, which
// with the slots makes eight wavefronts per CU): every kFkChunk bodies the wavefront turns the rows round -- half a wavefront per
// pose, consecutive lanes to consecutive words of the pose-major arrays -- so a store instruction touches 2 x 128 B instead of
// 64 lines (measured: scattered or partial-line stores run at 60 G transactions/s chip-wide whatever their size: 268 us per
// 100 000 rodent poses with every transform stored straight from the registers, 137 us like this).  The marker sites still go
// out one by one (12 B each; staging them as well would halve the resident wavefronts).  Same operations per pose as the oracle,
// in the same order: bit-identical.
// normalize = 0: quaternions in qpos are used as they are (the q_phase kernel has already applied
// kinematics' write-back normalisation; MJX's xquat IS that stored quaternion).
typedef const __attribute__((address_space(4))) int32_t *KInt;
typedef const __attribute__((address_space(4))) float *KFlt;
constexpr int kFkChunk = 8, kFkStride = 7 * kFkChunk + 1;  // bodies per staged chunk; words per lane's staging row (odd)
__global__ __launch_bounds__(64) void fk_kernel(FullModel M, const float *qpos, int N, float *qpos_norm_out,
                                                 float *__restrict__ xpos, float *__restrict__ xquat,
                                                 float *__restrict__ site_xpos, int normalize) {
    extern __shared__ float lds[];
    const int t = threadIdx.x;
    const int n0 = blockIdx.x * 64, n = n0 + t;
    const bool live = n < N;  // (the lanes past the end walk the last pose again: they help with the copies, their own results go nowhere)
    const int nc = live ? n : N - 1;
    const float *q = qpos + (size_t)nc * M.nq;
    float *qn = qpos_norm_out && live ? qpos_norm_out + (size_t)n * M.nq : nullptr;
    float *sx = site_xpos && live ? site_xpos + (size_t)n * 3 * M.K : nullptr;
    float *xp_blk = xpos ? xpos + (size_t)n0 * 3 * M.nbody : nullptr;
    float *xq_blk = xquat ? xquat + (size_t)n0 * 4 * M.nbody : nullptr;
    float *stage = lds + max(M.fk_nslots, 1) * (7 * 64);
    float *mine = stage + t * kFkStride;
    const int nvalid = min(64, N - n0);
    if (qpos_norm_out) {
        // the normalised copy of the coordinates: the wavefront's 64 rows are one contiguous block, copied as such (full lines);
        // the quaternion joints then overwrite their four words below (same wavefront, same address: in order).
        const float *src = qpos + (size_t)n0 * M.nq;
        float *dst = qpos_norm_out + (size_t)n0 * M.nq;
        if (dst != src) {  // (in place there is nothing to copy; partially overlapping arrays are not an input)
            const float *__restrict__ sr = src;
            float *__restrict__ dr = dst;
#pragma unroll 8
            for (int i = t, tot = nvalid * M.nq; i < tot; i += 64) dr[i] = sr[i];
        }
    }
    KInt brec = (KInt)M.fk_brec, jrec = (KInt)M.fk_jrec, sites = (KInt)M.fk_sites;
    KFlt spos = (KFlt)M.site_pos;  // (constant for the duration of a launch: stac_set_site_pos is stream-ordered)
    float *slot = lds + t;
    V3 pos = {0.f, 0.f, 0.f};
    Q4 quat = {1.f, 0.f, 0.f, 0.f};
    // the first two joints visited (a model without joints -- wave-uniform -- has no qpos row to read from)
    float qnext = 0.0f, qnext2 = 0.0f;
    if (M.nq > 0) { qnext = q[brec[13]]; qnext2 = q[brec[14]]; }
    for (int b = 0; b < M.nbody; ++b) {
        KInt br = brec + 16 * b;
        KFlt bf = (KFlt)br;
        if (b > 0) {
            const int psrc = br[0];
            if (psrc >= 0) {  // the parent is not the body before this one: its parked transform
                const float *sp = slot + psrc * (7 * 64);
                pos = {sp[0], sp[64], sp[128]};
                quat = {sp[192], sp[256], sp[320], sp[384]};
            }
            const Q4 pquat = quat;
            pos = add3(pos, rotate(V3{bf[4], bf[5], bf[6]}, pquat));
            quat = qmul(pquat, Q4{bf[8], bf[9], bf[10], bf[11]});
            const int j0 = br[2], j1 = j0 + br[3];
            for (int j = j0; j < j1; ++j) {
                KInt jr = jrec + 12 * j;
                KFlt jf = (KFlt)jr;
                const int ty = jr[0], ad = jr[1];
                const float qv = qnext, q0 = jf[2];
                qnext = qnext2;
                qnext2 = q[jr[3]];  // first coordinate of the joint two visits on (the last ones: their own again)
                const V3 jp = {jf[4], jf[5], jf[6]}, jax = {jf[8], jf[9], jf[10]};
                if (ty == JFREE) {
                    pos = {qv, q[ad + 1], q[ad + 2]};
                    float nn;
                    quat = normalize ? normalize4(ld4(q + ad + 3), &nn) : ld4(q + ad + 3);
                    if (qn) st4(qn + ad + 3, quat);
                } else if (ty == JHINGE) {
                    const V3 anchor = add3(rotate(jp, quat), pos);
                    float sn, cs;
                    sincos_((qv - q0) * 0.5f, &sn, &cs);
                    quat = qmul(quat, Q4{cs, jax.x * sn, jax.y * sn, jax.z * sn});
                    pos = sub3(anchor, rotate(jp, quat));
                } else if (ty == JSLIDE) {
                    const V3 axis = rotate(jax, quat);
                    const float d = qv - q0;
                    pos = {FMA(axis.x, d, pos.x), FMA(axis.y, d, pos.y), FMA(axis.z, d, pos.z)};
                } else {
                    const V3 anchor = add3(rotate(jp, quat), pos);
                    float nn;
                    const Q4 raw = {qv, q[ad + 1], q[ad + 2], q[ad + 3]};
                    const Q4 qloc = normalize ? normalize4(raw, &nn) : raw;
                    if (qn) st4(qn + ad, qloc);
                    quat = qmul(quat, qloc);
                    pos = sub3(anchor, rotate(jp, quat));
                }
            }
        }
        const int save = br[1];
        if (save >= 0) {
            float *sp = slot + save * (7 * 64);
            sp[0] = pos.x; sp[64] = pos.y; sp[128] = pos.z;
            sp[192] = quat.w; sp[256] = quat.x; sp[320] = quat.y; sp[384] = quat.z;
        }
        const int cb = b & (kFkChunk - 1);
        st4(mine + 4 * cb, quat);
        st3(mine + 4 * kFkChunk + 3 * cb, pos);
        if (cb == kFkChunk - 1 || b == M.nbody - 1) {  // turn the staged chunk round: lanes = (pose parity, word)
            const int b0 = b - cb, h = t >> 5, w = t & 31;
            wave_sync();
            if (xq_blk && w < 4 * (cb + 1))
#pragma unroll 4
                for (int p = h; p < nvalid; p += 2) xq_blk[(size_t)p * 4 * M.nbody + 4 * b0 + w] = stage[p * kFkStride + w];
            if (xp_blk && w < 3 * (cb + 1))
#pragma unroll 4
                for (int p = h; p < nvalid; p += 2) xp_blk[(size_t)p * 3 * M.nbody + 3 * b0 + w] = stage[p * kFkStride + 4 * kFkChunk + w];
            wave_sync();
        }
        if (sx)
            for (int s = br[7], s1 = br[12]; s < s1; ++s) {  // the marker sites of this body
                const int k = sites[s];
                st3(sx + 3 * k, add3(pos, rotate(V3{spos[3 * k], spos[3 * k + 1], spos[3 * k + 2]}, quat)));
            }
    }
}

// ------------------------------------------------------------------------------------------------
// offset phase (_m_opt, stac_core.py:148-170)
// ------------------------------------------------------------------------------------------------
// per (frame, site): R^T z and |z|^2 ; contrib[t][3K+1] with the K-site subtotal of |z|^2 last.
__global__ __launch_bounds__(64) void m_contrib_kernel(FullModel M, const float *kp, const float *xpos,
                                                        const float *xquat, int T, float *contrib) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int K = M.K;
    float *out = contrib + (size_t)t * (3 * K + 1);
    float z2t = 0.0f;
    for (int k = 0; k < K; ++k) {
        const int b = M.site_bodyid[k];
        const float *q = xquat + ((size_t)t * M.nbody + b) * 4;
        const float q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
        const float q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
        const float q12 = q[1] * q[2], q13 = q[1] * q[3], q23 = q[2] * q[3];
        const float m0 = q00 + q11 - q22 - q33, m1 = 2.0f * (q12 - q03), m2 = 2.0f * (q13 + q02);
        const float m3 = 2.0f * (q12 + q03), m4 = q00 - q11 + q22 - q33, m5 = 2.0f * (q23 - q01);
        const float m6 = 2.0f * (q13 - q02), m7 = 2.0f * (q23 + q01), m8 = q00 - q11 - q22 + q33;
        const float *p = xpos + ((size_t)t * M.nbody + b) * 3;
        const float *yk = kp + (size_t)t * 3 * K + 3 * k;
        const float z0 = yk[0] - p[0], z1 = yk[1] - p[1], z2 = yk[2] - p[2];
        out[3 * k + 0] = m0 * z0 + m3 * z1 + m6 * z2;
        out[3 * k + 1] = m1 * z0 + m4 * z1 + m7 * z2;
        out[3 * k + 2] = m2 * z0 + m5 * z1 + m8 * z2;
        z2t += z0 * z0 + z1 * z1 + z2 * z2;
    }
    out[3 * K] = z2t;
}

// partial[c] = sum over frames in index order (one thread per component), partial[3K+1] = T
__global__ __launch_bounds__(128) void m_reduce_kernel(const float *contrib, int T, int K, float *partial) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int W = 3 * K + 1;
    if (c < W) {
        float s = 0.0f;
        for (int t = 0; t < T; ++t) s += contrib[(size_t)t * W + c];
        partial[c] = s;
    } else if (c == W) {
        partial[W] = (float)T;
    }
}

// closed form; one thread (3K terms, scalar reductions in index order like the oracle)
__global__ void m_finish_kernel(int K, const float *partial, const float *m0, const float *dreg, float lam,
                                float *out, float *err) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const float T = partial[3 * K + 1], z2 = partial[3 * K];
    float ms = 0.0f, mm = 0.0f, reg = 0.0f;
    for (int i = 0; i < 3 * K; ++i) {
        const float d = dreg[i], s = partial[i], m0i = m0[i];
        const float denom = T + lam * d;
        const float numer = s + lam * d * m0i;
        // no frame anywhere and an unregularised coordinate: nothing determines it -- keep the previous offset
        const float v = denom == 0.0f ? m0i : numer / denom;
        out[i] = v;
        ms += v * s;
        mm += v * v;
        const float dr = d * (v - m0i);
        reg += dr * dr;
    }
    if (err) *err = ((z2 - 2.0f * ms) + T * mm) + lam * reg;
}

// control words of the chain queue / straggler hand-off (QArgs::ctl), set on the stream: no host buffer involved
__global__ void ctl_init_kernel(int32_t *ctl, int v0, int v1, int v2, int v3, int v4) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { ctl[0] = v0; ctl[1] = v1; ctl[2] = v2; ctl[3] = v3; ctl[4] = v4; ctl[5] = 0; ctl[6] = 0; ctl[7] = 0; }
}

// ---- chain order of a root-optimised launch (QArgs::perm) ----------------------------------------------------------------
// How long a chain runs is mostly a matter of its root solves, and those of how far the clip's first frame is turned away
// from the rest pose.  Its loss at the starting point -- rest pose moved to the root keypoint, trunk keypoints only; no
// kinematics needed -- ranks the chains well enough (Spearman 0.88 with the number of evaluations on the bench batch): a
// counting sort on the upper bits of that loss (4096 buckets; the order inside a bucket does not matter, and no result
// depends on where a chain runs).  What the order is for: QArgs::place.
constexpr int kKeyBuckets = 4096;
__global__ void root_key_kernel(const float *kp, int C, int F, int K, const float *rest_sites, const uint8_t *kpw, int root_kp_idx,
                                float rx, float ry, float rz, uint32_t *hist, uint32_t *keybits) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float *k0 = kp + (size_t)c * F * 3 * K;
    const float sx = k0[3 * root_kp_idx] - rx, sy = k0[3 * root_kp_idx + 1] - ry, sz = k0[3 * root_kp_idx + 2] - rz;
    float acc = 0.0f;
    for (int k = 0; k < K; ++k) {
        if (!kpw[k]) continue;
        const float dx = k0[3 * k] - (rest_sites[3 * k] + sx), dy = k0[3 * k + 1] - (rest_sites[3 * k + 1] + sy),
                    dz = k0[3 * k + 2] - (rest_sites[3 * k + 2] + sz);
        acc += dx * dx + dy * dy + dz * dz;
    }
    uint32_t b = __builtin_bit_cast(uint32_t, acc) >> 19;  // sign 0: exponent + four mantissa bits, monotone in acc
    if (!(acc == acc)) b = kKeyBuckets - 1;                 // NaN keypoints: last
    b = b < (uint32_t)kKeyBuckets ? b : (uint32_t)kKeyBuckets - 1;
    keybits[c] = b;
    atomicAdd(hist + b, 1u);
}
__global__ __launch_bounds__(1024) void key_scan_kernel(uint32_t *hist) {  // exclusive prefix sum over the 4096 buckets, in place
    __shared__ uint32_t part[1024];
    const int t = threadIdx.x;
    uint32_t v[4], sum = 0;
    for (int i = 0; i < 4; ++i) { v[i] = hist[4 * t + i]; sum += v[i]; }
    part[t] = sum;
    __syncthreads();
    for (int h = 1; h < 1024; h <<= 1) {
        const uint32_t add = t >= h ? part[t - h] : 0u;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (int i = 0; i < 4; ++i) { hist[4 * t + i] = run; run += v[i]; }
}
__global__ void key_scatter_kernel(const uint32_t *keybits, uint32_t *offs, int C, int32_t *perm) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    perm[atomicAdd(offs + keybits[c], 1u)] = c;
}

// ------------------------------------------------------------------------------------------------
// launchers (called from stac_abi.hip)
// ------------------------------------------------------------------------------------------------
hipError_t launch_chain_order(const float *kp, int C, int F, int K, const float *rest_sites, const uint8_t *kpw, int root_kp_idx,
                              float rx, float ry, float rz, uint32_t *hist, uint32_t *keybits, int32_t *perm, int32_t *place,
                              hipStream_t s) {
    hipError_t e = hipMemsetAsync(hist, 0, kKeyBuckets * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(place, 0, kPlaceWords * sizeof(int32_t), s);
    if (e != hipSuccess) return e;
    const int blocks = (C + 255) / 256;
    hipLaunchKernelGGL(root_key_kernel, dim3(blocks), dim3(256), 0, s, kp, C, F, K, rest_sites, kpw, root_kp_idx, rx, ry, rz, hist, keybits);
    hipLaunchKernelGGL(key_scan_kernel, dim3(1), dim3(1024), 0, s, hist);
    hipLaunchKernelGGL(key_scatter_kernel, dim3(blocks), dim3(256), 0, s, keybits, hist, C, perm);
    return hipGetLastError();
}

hipError_t launch_ctl_init(int32_t *ctl, int v0, int v1, int v2, int v3, int v4, hipStream_t s) {
    hipLaunchKernelGGL(ctl_init_kernel, dim3(1), dim3(64), 0, s, ctl, v0, v1, v2, v3, v4);
    return hipGetLastError();
}

// the instantiation of the most recent q_phase launch of this thread (tests: which kernel did the host choose?  stac_abi.hip,
// stac_debug_last_q_kernel)
thread_local int g_last_q_shape[4] = {0, 0, 0, 0};

template <int G, int NQR, int WPE, int SPECP>
static hipError_t launch_q(const QArgs &a, int wpb, size_t lds_bytes, hipStream_t s) {
    constexpr int SPEC = SPECP & ~1;  // (bit 0 = the lean kernel)
    g_last_q_shape[0] = G; g_last_q_shape[1] = NQR; g_last_q_shape[2] = WPE; g_last_q_shape[3] = SPECP;
    constexpr int NR = SPEC ? SPEC : 1;
    constexpr int NW = SPEC ? (G * NR >= 64 ? G * NR / 64 : 1) : 1;  // SPEC: wavefronts per chain; more than one -> one chain per workgroup
    constexpr int CPW = SPEC ? (G * NR >= 64 ? 1 : 64 / (G * NR)) : 64 / G;  // chains per wavefront
    if (NW > 1) wpb = NW;
    const int per_block = NW > 1 ? 1 : CPW * wpb;
    // a resume launch has one slot per hand-off entry; with a chain queue the grid covers the resident slots only
    const int slots = a.resume ? a.resume_slots : (a.queue_slots > 0 ? std::min(a.queue_slots, a.C) : a.C);
    const int blocks = (slots + per_block - 1) / per_block;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&q_phase_kernel<G, NQR, WPE, SPECP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((q_phase_kernel<G, NQR, WPE, SPECP>), dim3(blocks), dim3(64 * wpb), lds_bytes, s, a);
    return hipGetLastError();
}

// ---- the instantiations that ship -------------------------------------------------------------------------------------------
// (lanes per chain G, solver registers per lane NQR: nq <= G * NQR, register cap WPE).  Every shape here passes the resource
// gate of tests/test_isa_hazards.py (scratch <= 64 B per lane, <= 40 scalars spilled into vector lanes; table:
// profiles/r04/resource_usage.txt).  What does not is not built: the 128-VGPR variants of the 8- and 16-lane kernels (74 to 750
// spilled vector registers) and of the 32- / 64-lane kernels with four or more solver registers per lane (17 to 80), 32 solver registers per lane at 4 or 8 lanes and the 4-lane kernels altogether (160 B to 1.3 KB of
// scratch; never chosen automatically, slower than 16 lanes at every batch size) -- a request for them runs on the next wider
// group, results are the same bit for bit.  Latency kernels stop at 10 solver registers per lane at 8 lanes and 8 at 16 or 32
// (round 3: the wider ones, 300+ B of scratch, read spill slots before writing them); wider models take more lanes per role.
#ifdef STAC_INST_SUBSET  // developer builds (experiments): only the shapes of the default bench and of its 250-frame-clip leg
#define STAC_Q_SHAPES(X) X(16, 5, 2) X(16, 5, 3)
#define STAC_Q_LEAN_SHAPES(X) X(16, 5, 3)
#define STAC_Q_SPEC_LEAN_SHAPES(X) X(16, 5, 4) X(16, 5, 8) X(32, 3, 8)
#define STAC_Q_SPEC_SHAPES(X) X(16, 5, 4) X(32, 3, 8)
#else
// lean kernels (SPECP bit 0): the shapes that rodent-sized models run in -- large batches, the straggler hand-off, few long clips
// (the first shape of a width that holds nq is taken: narrower ones first.  Three solver registers per lane at 16 lanes, two at 32: models of
//  up to 48 / 64 coordinates -- the fruit fly's 43 --, whose nq-sums and staging then run over three registers instead of five)
#define STAC_Q_LEAN_SHAPES(X) X(16, 3, 3) X(16, 5, 2) X(16, 5, 3) X(32, 3, 2) X(32, 8, 2)
#define STAC_Q_SPEC_LEAN_SHAPES(X) X(16, 3, 4) X(16, 5, 4) X(16, 5, 8) X(32, 2, 8) X(32, 3, 8) X(32, 8, 8)
#define STAC_Q_SHAPES(X)                                                        \
    X(8, 10, 2) X(8, 16, 2)                                                      \
    X(16, 5, 2) X(16, 5, 3) X(16, 8, 2) X(16, 8, 3) X(16, 16, 2)                 \
    X(32, 3, 2) X(32, 3, 4) X(32, 4, 2) X(32, 8, 2)                              \
    X(64, 2, 2) X(64, 2, 4) X(64, 4, 2)
// (G lanes per role, NQR, roles per chain)
#define STAC_Q_SPEC_SHAPES(X)                                                   \
    X(8, 10, 4) X(8, 10, 8) X(16, 5, 4) X(16, 8, 4) X(32, 3, 8) X(32, 8, 8) X(64, 2, 8) X(64, 4, 8)
#endif

// Is there a throughput instantiation with G lanes per chain and register cap wpe that holds nq coordinates?
bool q_phase_has_variant(int G, int nq, int wpe) {
#define STAC_HAS(GG, RR, WW) if (G == GG && wpe == WW && nq <= GG * RR) return true;
    STAC_Q_SHAPES(STAC_HAS)
#undef STAC_HAS
    return false;
}

// Is there a lean instantiation (SPECP bit 0) for this shape?  spec = 0: throughput kernel with register cap wpe; else the latency
// kernel with `spec` roles of G lanes.
bool q_phase_has_lean_variant(int G, int nq, int wpe, int spec) {
    if (spec) {
#define STAC_HAS(GG, RR, NRR) if (G == GG && spec == NRR && nq <= GG * RR) return true;
        STAC_Q_SPEC_LEAN_SHAPES(STAC_HAS)
#undef STAC_HAS
        return false;
    }
#define STAC_HAS(GG, RR, WW) if (G == GG && wpe == WW && nq <= GG * RR) return true;
    STAC_Q_LEAN_SHAPES(STAC_HAS)
#undef STAC_HAS
    return false;
}
// The launch-wide choices the lean kernels have compiled in (q_phase_kernel, SPECP bit 0): phase mode with the model's own box, only
// hinges below a free root at qpos 0 .. 6 (QArgs::flags == 16: set_hinges_flag, no developer flag), the split kinematics
// (PlanHeader::fk3: stac_plan.hpp), every site in registers at this group width.  The host decides with it which chain layout
// the launch gets (run_q) and passes its decision to launch_q_phase.
// solver registers per lane of the lean instantiation that holds nq at this width (the same for every register cap / role count); 0: none
int q_phase_lean_nqr(int G, int nq) {
    int nqr = 0;
#define STAC_NQR(GG, RR, WW) if (G == GG && nq <= GG * RR && (nqr == 0 || RR < nqr)) nqr = RR;
    STAC_Q_LEAN_SHAPES(STAC_NQR)
    STAC_Q_SPEC_LEAN_SHAPES(STAC_NQR)
#undef STAC_NQR
    return nqr;
}
// ... and does it hold the model's K sites in registers?  (what the width heuristics of the host need to know before a launch exists)
bool q_phase_lean_holds(int G, int nq, int K) {
    const int nqr = q_phase_lean_nqr(G, nq);
    return nqr > 0 && K <= lean_site_rounds(G, nqr) * G;
}
bool q_phase_lean_conditions(const QArgs &a, int G) {
    const int nqr = q_phase_lean_nqr(G, a.h.nq);
    return !a.single && !a.bounds && a.flags == 16 && a.free0p == 1 && a.h.fk3 == 1 && nqr > 0 && a.h.K <= lean_site_rounds(G, nqr) * G &&
           a.h.nqj == 1 && !a.h.has_ball && G >= 16;
}

// wpb = wavefronts per workgroup (they share the plan copy), wpe = register-cap variant (2, 3 or 4 wavefronts per SIMD; the
// nearest one that exists for G is taken), spec = evaluation roles per chain in latency mode (0 = throughput mode).
// lean: the launch carries the lean chain layout (the host has checked q_phase_lean_conditions and that the shape exists).
// *capacity_out = G * NQR of the instantiation that ran, 0 if none holds nq at this G.
hipError_t launch_q_phase(const QArgs &a, int G, int wpb, int wpe, int spec, size_t lds_bytes, hipStream_t s,
                          int *capacity_out, bool lean) {
    const int nq = a.h.nq;
    *capacity_out = 0;
    if (lean && !(q_phase_lean_conditions(a, G) && q_phase_has_lean_variant(G, nq, wpe, spec))) return hipErrorInvalidValue;
    if (spec) {
#define STAC_TRY_SPEC_LEAN(GG, RR, NRR)                             \
    if (lean && G == GG && spec == NRR && nq <= GG * RR) {          \
        *capacity_out = GG * RR;                                    \
        return launch_q<GG, RR, 2, NRR | 1>(a, wpb, lds_bytes, s);  \
    }
        STAC_Q_SPEC_LEAN_SHAPES(STAC_TRY_SPEC_LEAN)
#undef STAC_TRY_SPEC_LEAN
#define STAC_TRY_SPEC(GG, RR, NRR)                                  \
    if (G == GG && spec == NRR && nq <= GG * RR) {                  \
        *capacity_out = GG * RR;                                    \
        return launch_q<GG, RR, 2, NRR>(a, wpb, lds_bytes, s);      \
    }
        STAC_Q_SPEC_SHAPES(STAC_TRY_SPEC)
#undef STAC_TRY_SPEC
        return hipErrorInvalidValue;
    }
    if (!lean && !q_phase_has_variant(G, nq, wpe)) wpe = 2;  // (every G has its 2-per-SIMD variants)
#define STAC_TRY_LEAN(GG, RR, WW)                                   \
    if (lean && G == GG && wpe == WW && nq <= GG * RR) {            \
        *capacity_out = GG * RR;                                    \
        return launch_q<GG, RR, WW, 1>(a, wpb, lds_bytes, s);       \
    }
    STAC_Q_LEAN_SHAPES(STAC_TRY_LEAN)
#undef STAC_TRY_LEAN
#define STAC_TRY(GG, RR, WW)                                        \
    if (G == GG && wpe == WW && nq <= GG * RR) {                    \
        *capacity_out = GG * RR;                                    \
        return launch_q<GG, RR, WW, 0>(a, wpb, lds_bytes, s);       \
    }
    STAC_Q_SHAPES(STAC_TRY)
#undef STAC_TRY
    return hipErrorInvalidValue;
}

hipError_t launch_fk(const FullModel &M, const float *qpos, int N, float *qn, float *xpos, float *xquat,
                     float *site_xpos, int normalize, hipStream_t s) {
    if (N <= 0) return hipSuccess;
    const size_t lds_bytes = ((size_t)std::max(M.fk_nslots, 1) * 7 + kFkStride) * 64 * sizeof(float);  // (rodent: 5.4 + 14.6 KB per wavefront)
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;  // (a tree with more than 83 open branch points)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fk_kernel, dim3((N + 63) / 64), dim3(64), lds_bytes, s, M, qpos, N, qn, xpos, xquat, site_xpos, normalize);
    return hipGetLastError();
}

hipError_t launch_m_partial(const FullModel &M, const float *kp, const float *xpos, const float *xquat, int T,
                            float *contrib, float *partial, hipStream_t s) {
    // T == 0 (a rank of a sharded fit that owns no sampled frame): no contributions, the sums below come out as
    // zeros with partial[3K+1] = 0 -- a zero-block grid would be an invalid launch
    if (T > 0) {
        hipLaunchKernelGGL(m_contrib_kernel, dim3((T + 63) / 64), dim3(64), 0, s, M, kp, xpos, xquat, T, contrib);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    const int W = 3 * M.K + 2;
    hipLaunchKernelGGL(m_reduce_kernel, dim3((W + 127) / 128), dim3(128), 0, s, contrib, T, M.K, partial);
    return hipGetLastError();
}

hipError_t launch_m_finish(int K, const float *partial, const float *m0, const float *dreg, float lam, float *out,
                           float *err, hipStream_t s) {
    hipLaunchKernelGGL(m_finish_kernel, dim3(1), dim3(1), 0, s, K, partial, m0, dreg, lam, out, err);
    return hipGetLastError();
}

}  // namespace stac

