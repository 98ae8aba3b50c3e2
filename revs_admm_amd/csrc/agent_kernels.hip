// Residence ("Home") side of one ADMM iteration, all homes at once -- gfx950.
//
// Replaces, per iteration k of lpsolver.solve_ADMM (reference lpsolver.py:262-284):
//   H_obj = Home(cost, homes[h], P_est[k][h], P_sch[k][h], G[k][h]); H_obj.solve()
//   check = P_est[k+1][h] - P_sch[k+1][h];  G[k+1][h] = G[k][h] + (kappa/2) check
//   diff[k+1][h] = |check| / T
// for every home h, in ONE kernel: the profiles of a home are read once, the
// schedule is solved in registers, and P_sch, G, S, C, diff are written once.
//
// Mapping.  A home's T slots are spread over a group of LPA lanes, SPL
// consecutive slots per lane (LPA*SPL >= T), so a 64-wide wavefront carries
// 64/LPA homes: T=24 -> 8 lanes x 3 slots, 8 homes per wavefront, every lane
// busy.  (One whole wavefront per home would idle 40 of 64 lanes at T=24.)
// Rows are slot-contiguous in HBM, so the 64 lanes of a wavefront read one
// contiguous 64/LPA * T * 4 byte span per profile (768 B at T=24): coalesced.
// The SOC operator K (prefix sum over slots, lpsolver.py:105-108) and its
// transpose (suffix sum) are an in-lane scan plus a log2(LPA)-step DPP scan
// across the group -- no LDS, no T x T matrix is ever formed.
//
// Arithmetic: float (BASELINE north star); the per-home residual terms (diff, dsq) are
// summed in double by the residual kernels on request, which also take the maximum of
// diff over all homes (the reference's only convergence measure).
#define REVS_AGENT_TU
#include "common.h"
#include "select_body.h"
#include "tree_body.h"
#include "internal.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace revs {

struct AgentArgs {
    int64_t n;
    int32_t T;
    const float *cost;
    const revs_home_t *homes;
    const float *load;
    const float *pe_old;
    const float *pe_new;
    float *ps;             // P_sch[k] in
    float *gam;            // G[k] in
    float *ps_out;         // P_sch[k+1] out (== ps for the in-place form)
    float *gam_out;        // G[k+1] out
    float *s_out;
    float *c_out;
    float *diff;
    float *dsq;            // per home: sum_t (P_sch[k+1] - P_sch[k])^2
    int32_t *status;
    float *y_state;
    float kappa;
    revs_pdhg_t pd;
    // nsel > 0: the first nsel workgroups of the launch run the operator's candidate
    // selection (one slot each, select_body.h) instead of homes -- independent work that
    // overlaps the sweep and delivers the operator's verdict while the sweep is running
    int32_t nsel;
    SelectArgs sel;
    // p_next != NULL: the sweep also does the home pass of the NEXT operator evaluation for
    // multipliers y = 0 (revs_op_dual_eval with dsl = NULL on the state it has just produced):
    // pe2_out = max(g0', 0) as float, g0' = (P_est[k+1] + P_sch[k+1])/2 - G[k+1]/kappa, and
    // p_next[node][t] += g (double; must be zero on entry) -- one pass over the homes per
    // ADMM iteration instead of two while the operator's rows stay slack
    const int32_t *node_of;
    double *p_next;
    float *pe2_out;
    // Streaming steady state (revs_plan_stream_run).  ctl != NULL: every workgroup first reads
    // ctl->bad_seq and returns at once when an EARLIER launch's verdict failed (nothing this
    // launch would write may then be written).  tree.n > 0: the first nsel = T workgroups judge
    // the voltage rows of slot blockIdx.x from the node sums p_in by the tree form of R p
    // (tree_body.h) instead of running the candidate selection, clear their share of p_zero,
    // and the last of them to finish leaves the launch's record {rmax, failed, seq} in `rec`.
    StreamCtl *ctl;
    unsigned int seq, base_seq;   // this launch's number; the first number of the current call
    TreeArgs tree;
    const double *p_in;
    double *p_zero;
    double vtol;           // eps * scale: the verdict fails when a row is further out
    double *rec;           // this launch's record slot, double[4] (pinned host memory)
    unsigned int *flags;   // OR of status bits over all residences (pinned host memory), or NULL
    int32_t m;
    // kin > 1: this launch does kin consecutive ADMM iterations of every residence, the state kept in
    // registers (multipliers zero: the operator's answer of iteration g + 1 is a function of the
    // residence's own state after iteration g) -- profiles read and written once per kin iterations.
    // Inner iteration i accumulates its node sums into p_next + i * slice_stride, writes its diff to
    // diff + i * diff_stride and the largest diff of this launch's residences into the REVS_DMAX_SLOTS
    // doubles at dmax_out + i * slice_stride (atomic max on the bit pattern of a non-negative double,
    // workgroup b into slot b % REVS_DMAX_SLOTS; NULL: not recorded).
    // pe_out: P_est[g + kin] (the estimate the last inner iteration consumed); y_out: the carried
    // PDHG multiplier after the last inner iteration (== y_state: in place).
    int32_t kin;
    float *pe_out;
    float *y_out;
    long long slice_stride, diff_stride;
    double *dmax_out;
    // The folded chain (revs_plan_chain_fold_run, template argument CHAIN): the operator's multipliers
    // are NOT zero.  The sweep forms the operator's answer itself, pen = max(g0 - d[node], 0) with the
    // shifts d = R^T y / kappa of the trial multipliers (sh_a, double[T][sh_m], slot-major: the evaluation
    // kernel's arithmetic, bit for bit -- the operator launch before this sweep computes them) and
    // writes it to pe_out; it accumulates that evaluation's node sums p | N | -(kappa/2) sum g^2 into
    // fold_a and, after the dual update, those of the evaluation of the SAME multipliers on the state
    // it has just produced into fold_b (both double[T][sh_m][4] = {p, N, q, 0}, slot-major, zero on entry; shifts sh_b: the
    // same sums taken in the order that evaluation's own home pass would take them): one pass over
    // the residences per ADMM iteration while rows keep binding.
    const double *sh_a, *sh_b;
    int32_t sh_m;
    double sh_kappa;
    double *fold_a, *fold_b;
    // wg_order != NULL (nsel == 0): workgroup b of the launch takes the residences of workgroup wg_order[b] -- the
    // workgroups that hold the most residences with an EV first, so that the launch's last round, which cannot fill
    // the chip, is made of the quick ones (runtime.cpp: plan_wg_order)
    const int32_t *wg_order;
};
constexpr int kMaxInner = REVS_AGENT_MAX_INNER;
// nodes whose sums a workgroup accumulates in LDS (residences are sorted by node; the others' go straight to
// global memory): 4 -- but 2 where a lane group is 16 lanes wide with more than 3 slots each (16 residences
// per workgroup: two nodes unless nodes hold fewer than 8 residences), which buys T = 96 its 16 inner iterations
constexpr int shape_node_loc(int slots) { return slots > 48 ? 2 : 4; }
constexpr int shape_max_inner(int slots) {      // inner iterations whose node-sum accumulators fit 24 KB of LDS
    const int fit = 24576 / (shape_node_loc(slots) * slots * 8);
    return fit >= kMaxInner ? kMaxInner : (fit >= 16 ? 16 : (fit >= 8 ? 8 : (fit >= 4 ? 4 : (fit >= 2 ? 2 : 1))));
}

#ifndef REVS_AGENT_MULTI_WAVES
#define REVS_AGENT_MULTI_WAVES 5     // wavefronts per SIMD of the multi-iteration sweep (tuning: build with -D)
#endif

// SPL consecutive floats of one lane as ONE global_load/store_dwordxSPL: the 64 lanes of a
// wavefront then cover one contiguous 64*SPL*4-byte span per instruction instead of SPL
// interleaved stride-SPL passes over it (global memory only needs dword alignment).
template <int SPL>
struct alignas(4) PackF { float v[SPL]; };
template <int SPL>
__device__ __forceinline__ void ld_pack(const float *p, float (&o)[SPL]) {
    const PackF<SPL> t = *reinterpret_cast<const PackF<SPL> *>(p);
#pragma unroll
    for (int j = 0; j < SPL; ++j) o[j] = t.v[j];
}
template <int SPL>
__device__ __forceinline__ void st_pack(float *p, const float (&v)[SPL]) {
    PackF<SPL> t;
#pragma unroll
    for (int j = 0; j < SPL; ++j) t.v[j] = v[j];
    *reinterpret_cast<PackF<SPL> *>(p) = t;
}

constexpr int kBlock = 256;
constexpr float kSocTarget = 0.9f;   // lpsolver.py:109
constexpr float kSocMax = 1.0f;      // lpsolver.py:102-103

// rank of each of this lane's SPL keys among the group's LPA*SPL keys, ties to
// the earlier slot: rank_j = #{tau : key_tau < key_j or (key_tau == key_j and tau < t_j)}
// = #{tau : (key_tau, tau) < (key_j, t_j)} lexicographically: the key mapped to an integer of the same order in the
// high word, the slot in the low word, ONE 64-bit compare and an add-with-carry per pair (the two float compares,
// the slot compare and the mask logic were five vector and two scalar instructions per pair: 360 + 144 per
// wavefront, most of the on/off charger's solve).
template <int LPA, int SPL>
__device__ __forceinline__ void group_rank(const float (&key)[SPL], int t0, int (&rank)[SPL]) {
    int ord[SPL];
    long long mine[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const int b = __float_as_int(key[j]);
        ord[j] = b ^ ((b >> 31) & 0x7fffffff);           // floats -> integers of the same order (-0 < +0: keys are sums, never -0)
        mine[j] = ((long long)ord[j] << 32) | (unsigned int)(t0 + j);
        rank[j] = 0;
    }
#pragma unroll 1
    for (int sl = 0; sl < LPA; ++sl) {
#pragma unroll
        for (int sj = 0; sj < SPL; ++sj) {
            const long long other = ((long long)__shfl(ord[sj], sl, LPA) << 32) | (unsigned int)(sl * SPL + sj);
#pragma unroll
            for (int j = 0; j < SPL; ++j) rank[j] += other < mine[j] ? 1 : 0;
        }
    }
}

// The same ranks from DOUBLE keys (revs_pdhg_t::keys64, and the individual mode):
// rank_j = #{tau : key_tau < key_j or (key_tau == key_j and tau < t_j)} -- numpy's stable argsort, which is what the
// oracle ranks with (oracle/revs_oracle.py home_solve_binary; the reference's Gurobi picks among exactly tied slots
// by an order of its own).  The keys are doubles formed from the float inputs in the oracle's
// own order of operations (round 5): mapped to integers of the same order, (key, slot) pairs compare
// lexicographically as ord_tau < ord_j + [tau < t_j] -- every slot of an earlier lane precedes every slot of this
// one, none of a later lane does, and the lane's own slots are settled behind the loop.
template <int LPA, int SPL>
__device__ __forceinline__ void group_rank(const double (&key)[SPL], int lig, int (&rank)[SPL]) {
    long long ord[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const long long b = __double_as_longlong(key[j]);
        ord[j] = b ^ ((b >> 63) & 0x7fffffffffffffffLL);  // doubles -> integers of the same order (keys are sums and products of a positive rating: never -0)
        rank[j] = 0;
    }
#pragma unroll 1
    for (int sl = 0; sl < LPA; ++sl) {
        const long long c = sl < lig ? 1 : 0;
        long long thr[SPL];
#pragma unroll
        for (int j = 0; j < SPL; ++j) thr[j] = ord[j] + c;          // (+inf + 1 does not overflow)
#pragma unroll
        for (int sj = 0; sj < SPL; ++sj) {
            const long long other = __shfl(ord[sj], sl, LPA);
#pragma unroll
            for (int j = 0; j < SPL; ++j) rank[j] += other < thr[j] ? 1 : 0;
        }
    }
    // (sl == lig compared strictly: an equal key at an EARLIER slot of this lane precedes as well)
#pragma unroll
    for (int j = 1; j < SPL; ++j)
#pragma unroll
        for (int sj = 0; sj < j; ++sj) rank[j] += ord[sj] == ord[j] ? 1 : 0;
}

// One slot's verdict inside the streaming sweep's launch (workgroup t < T).
__device__ __forceinline__ void stream_verdict_body(const AgentArgs &a, const int t) {
    extern __shared__ double tree_lds[];
    __builtin_amdgcn_s_setprio(3);       // few wavefronts among the sweep's, and on the critical path
    const int tid = threadIdx.x;
    // clear this workgroup's share of the node-sum array the launch AFTER this one accumulates into
    if (a.p_zero) {
        const int64_t tot = (int64_t)a.m * a.T, per = (tot + a.nsel - 1) / a.nsel;
        const int64_t i0 = (int64_t)t * per, i1 = i0 + per < tot ? i0 + per : tot;
        for (int64_t i = i0 + tid; i < i1; i += kBlock) a.p_zero[i] = 0.0;
    }
    const double rmax = tree_rmax(a.tree, a.p_in, a.T, t, a.sel.vlo, a.sel.vhi, tree_lds, nullptr);
    if (tid != 0) return;
    // (every later launch of the call is silenced, so a call writes at most this one value; a
    // number left by an earlier call is below base_seq and ignored: no re-arming between calls)
    if (!(rmax <= a.vtol))
        __hip_atomic_store(&a.ctl->bad_seq, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_max(&a.ctl->rmax_bits, (unsigned long long)__double_as_longlong(rmax),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // both performed before this slot is counted
    const unsigned int old = __hip_atomic_fetch_add(&a.ctl->arrive, 1u, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
    if (old != (unsigned int)a.nsel - 1) return;
    // last slot of this launch: the record the host polls, then reset for the next launch
    const unsigned long long bits = __hip_atomic_load(&a.ctl->rmax_bits, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
    const unsigned int bad = __hip_atomic_load(&a.ctl->bad_seq, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&a.ctl->arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&a.ctl->rmax_bits, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    volatile double *rec = a.rec;
    rec[0] = __longlong_as_double((long long)bits);
    rec[1] = (bad >= a.base_seq && bad <= a.seq) ? 1.0 : 0.0;
    __threadfence_system();
    rec[2] = (double)a.seq;
}

// 64 VGPRs = 8 wavefronts per SIMD: the sweep's own path needs 56; the verdict workgroups that
// share its launch are written to fit the same budget (tree_body.h) and must not lower it.
// (Measured and rejected, round 2: a wavefront working through 2-4 groups of residences with the
// next group's loads in flight -- the second input set costs 118-128 VGPRs, occupancy 4, and the
// launch takes 24-25 us against 20; without the prefetch the loop spills at the 64-register cap;
// 16 homes per wavefront (4 lanes x 6 slots) squeezed to 72 VGPRs so that all 6 250 wavefronts are
// resident at once: 22.7 us against 18.5.)
// MULTI: the loop over AgentArgs::kin inner iterations exists (its loop-carried state -- five
// profiles, the PDHG constants of the residence -- does not fit 64 VGPRs: the multi-iteration
// form trades occupancy for registers, it is bound by instruction issue, not by memory latency);
// without it the body runs once and compiles to the one-iteration kernel.
// CHAIN: the folded chain's sweep (see AgentArgs::sh_R): one iteration, shifts and two sets of
// node-sum accumulators in LDS, doubles in flight -- like MULTI it gives up occupancy for registers.
#ifndef REVS_AGENT_CHAIN_WAVES
#define REVS_AGENT_CHAIN_WAVES 6     // ... of the folded chain's sweep (tuning: build with -D; 5 / 6 / 7 / 8: 0.0469 / 0.0462 / 0.0466 / 0.0497 ms per binding iteration)
#endif
// (on/off chargers, one iteration per launch: 7 -- the double ranking keys of round 5 do not fit 64 registers without scratch)
constexpr int agent_waves(int spl, bool full_rows, bool multi, bool chain, int mode) {
    return (spl <= 4 && !full_rows) ? (multi ? REVS_AGENT_MULTI_WAVES : (chain ? REVS_AGENT_CHAIN_WAVES : (mode == REVS_MODE_BINARY ? 7 : 8))) : 4;
}
template <int LPA, int SPL, int MODE, bool FULL_ROWS = false, bool MULTI = false, bool CHAIN = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(agent_waves(SPL, FULL_ROWS, MULTI, CHAIN, MODE))))
void agent_step_kernel(const AgentArgs a) {
    constexpr int kHomesPerBlock = kBlock / LPA;
    // An earlier launch of this call failed its verdict: this launch must write nothing.  The
    // control word is requested here and tested where the residences' own loads have been issued
    // (reading is harmless): one loaded memory round trip per workgroup less in front of them.
    unsigned int bad = 0u;
    if (a.ctl) bad = a.ctl->bad_seq;
    const bool silenced = a.ctl && bad >= a.base_seq && bad < a.seq;
    if (a.nsel > 0 && (int)blockIdx.x < a.nsel) {          // uniform per workgroup
        if (silenced) return;
        if (a.tree.n > 0) stream_verdict_body(a, (int)blockIdx.x);
        else dual_select_body(a.sel, (int)blockIdx.x);
        return;
    }
    const int bid = a.wg_order ? a.wg_order[blockIdx.x] : (int)blockIdx.x - a.nsel;
    const int tid = threadIdx.x;
    const int lig = tid & (LPA - 1);
    const int T = a.T;
    const int t0 = lig * SPL;
    const float kappa = a.kappa;
    const int kin = MULTI ? a.kin : 1;
    const bool tfull = t0 + SPL <= T;        // this lane's SPL slots all exist: one dwordxSPL per profile
    // pe_new == NULL (multipliers all zero): the operator's answer max(g0, 0) is recomputed
    // below from the three profiles it is a function of, instead of being read
    const bool rec_pen = a.pe_new == nullptr;
    const float *pen_src = rec_pen ? a.pe_old : a.pe_new;   // (a mapped address for the else branch)
    float cst[SPL];
    bool tval[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        tval[j] = t0 + j < T;
        cst[j] = tval[j] ? a.cost[t0 + j] : 0.f;
    }
    // node sums of the next home pass: residences are sorted by node, so a workgroup's homes
    // sit on a few consecutive nodes -- accumulate in LDS (one set per inner iteration), flush one
    // global add per (iteration, node, slot) at the end of the launch.
    // (Measured and rejected, round 2: per-wavefront xor-shuffle reduction and one global add per
    // slot, no LDS and no workgroup barrier -- 19.5 us against 18.4 at T = 24, and 149 us against
    // 74 at 125 000 x 96, where a wavefront holds 4 residences and the memory-side f64 atomics
    // quadruple.)
    constexpr int kSlots = LPA * SPL, kNodeLoc = shape_node_loc(kSlots);
    // (at most 24 KB of accumulators per workgroup: 16 iterations up to 32 slots per group, 8 up to 96, 4 beyond)
    constexpr int kAcc = MULTI ? shape_max_inner(kSlots) : 1;
    __shared__ double nacc[kAcc][kNodeLoc][kSlots];
    __shared__ unsigned int dmx[kAcc];
    __shared__ unsigned int arrived;      // wavefronts of this workgroup that have finished their residences (the last one flushes)
    const int64_t first = (int64_t)bid * kHomesPerBlock;
    // Every global load of the kernel is issued before the first use of any of them (one exposed
    // memory latency per wavefront, not three): profiles, then the home record and the carried
    // PDHG multiplier.
    const int64_t agent = first + tid / LPA;
    const bool live = agent < a.n;
    const int64_t row = agent * (int64_t)T;
    float L[SPL], pe[SPL], pso[SPL], gm[SPL], pen[SPL];
    if (live && tfull) {
        ld_pack<SPL>(a.load + row + t0, L);
        ld_pack<SPL>(a.pe_old + row + t0, pe);
        if (!rec_pen) ld_pack<SPL>(a.pe_new + row + t0, pen);
        ld_pack<SPL>(a.ps + row + t0, pso);
        ld_pack<SPL>(a.gam + row + t0, gm);
    } else {
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            // out-of-range lanes read element 0 (always mapped) and discard it: straight-line
            // loads instead of one exec-masked branch per element
            const bool v = live && tval[j];
            const int64_t o = v ? row + t0 + j : 0;
            const float vL = a.load[o], vpe = a.pe_old[o], vpn = pen_src[o], vps = a.ps[o], vg = a.gam[o];
            L[j] = v ? vL : 0.f;
            pe[j] = v ? vpe : 0.f;
            pen[j] = v ? vpn : 0.f;
            pso[j] = v ? vps : 0.f;
            gm[j] = v ? vg : 0.f;
        }
    }
    revs_home_t h;
    float yy_in = 0.f;
    int node_ld = 0;            // (with the state: the node sums' accumulators are picked by it right behind the barrier)
    if ((a.p_next || CHAIN) && live) node_ld = a.node_of[agent];
    if (live) {
        h = a.homes[agent];
        if constexpr (MODE == REVS_MODE_RELAXED_PDHG && !FULL_ROWS)
            if (a.y_state) yy_in = a.y_state[agent];
    } else {
        h.ev = 0; h.start = 0; h.end = 0; h.nmin = 0; h.nmax = 0;
        h.rating = 0.f; h.capacity = 1.f; h.initial = 0.f;
    }
    if (silenced) return;
    int base = 0;
    __shared__ double dsh[CHAIN ? 2 : 1][CHAIN ? kNodeLoc : 1][CHAIN ? kSlots : 1];   // shifts of this workgroup's nodes: list order, row order
    // fold_a, fold_b: p | N | sum g^2 -- the addends rounded so that the sums are exact in any order
    // (revs_q36 / revs_q32, common.h): the bits the evaluation kernel's fixed-order sums give
    __shared__ double facc[CHAIN ? 2 : 1][3][CHAIN ? kNodeLoc : 1][CHAIN ? kSlots : 1];
    if (a.p_next || CHAIN) {      // (behind the loads: the barrier does not wait for them)
        if (a.p_next) {
            for (int i = tid; i < kin * kNodeLoc * kSlots; i += kBlock) (&nacc[0][0][0])[i] = 0.0;
            if (tid < kAcc) dmx[tid] = 0u;
        }
        if (tid == 0) arrived = 0u;
        base = a.node_of[first < a.n ? first : a.n - 1];
        if constexpr (CHAIN) {
            for (int i = tid; i < 2 * 3 * kNodeLoc * kSlots; i += kBlock) (&facc[0][0][0][0])[i] = 0.0;
            for (int i = tid; i < kNodeLoc * T; i += kBlock) {
                const int l = i / T, t = i - l * T;
                dsh[0][l][t] = base + l < a.sh_m ? a.sh_a[(int64_t)t * a.sh_m + base + l] : 0.0;
                dsh[1][l][t] = base + l < a.sh_m ? a.sh_b[(int64_t)t * a.sh_m + base + l] : 0.0;
            }
        }
        __syncthreads();
    }
    const bool full = live && tfull;
    bool valid[SPL], win[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) valid[j] = live && tval[j];
    const float inv_kf = 1.0f / kappa;
    if (rec_pen) {      // same arithmetic as the folded home pass that would have stored it
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const float g0 = revs_g0f(pe[j], pso[j], gm[j], inv_kf);
            pen[j] = (valid[j] && g0 > 0.f) ? g0 : 0.f;
        }
    }
    const int node = node_ld;
    double dl[CHAIN ? SPL : 1];      // the shifts of this lane's slots
    // p | N | sum g^2 of one evaluation into accumulator set f (fold_a / fold_b behind it); zero addends where the
    // residence is clamped.  The sweep's time follows its instruction count one to one (r04: summing the addends over
    // the wavefront's lane groups first -- 8 lanes on 8 LDS addresses instead of 64 -- cost 160 VALU instructions per
    // wavefront and 3 us per sweep, the LDS atomics' serialisation is not what bounds it): the local case is nine
    // straight-line LDS adds per lane, one branch in front of them.
    auto fold_add = [&](const int f, const double (&gq)[CHAIN ? SPL : 1], const double (&cn)[CHAIN ? SPL : 1],
                        const double (&g2)[CHAIN ? SPL : 1]) {
        if constexpr (CHAIN) {
            const int loc = node - base;
            if (loc < kNodeLoc) {           // (lanes without a residence: node = 0 <= base, addends zero)
                const int lc = loc < 0 ? 0 : loc;
#pragma unroll
                for (int j = 0; j < SPL; ++j) {
                    unsafeAtomicAdd(&facc[f][0][lc][t0 + j], gq[j]);
                    unsafeAtomicAdd(&facc[f][1][lc][t0 + j], cn[j]);
                    unsafeAtomicAdd(&facc[f][2][lc][t0 + j], g2[j]);
                }
            } else {
                double *const glob = f ? a.fold_b : a.fold_a;
#pragma unroll
                for (int j = 0; j < SPL; ++j) {
                    if (cn[j] != 0.0) {
                        const int64_t o = 4 * ((int64_t)(t0 + j) * a.sh_m + node);      // [T][m][4]: {p, N, q, 0} per slot and node
                        unsafeAtomicAdd(&glob[o], gq[j]);
                        unsafeAtomicAdd(&glob[o + 1], 1.0);
                        unsafeAtomicAdd(&glob[o + 2], -0.5 * a.sh_kappa * g2[j]);
                    }
                }
            }
        }
    };
    if constexpr (CHAIN) {
        // the operator's answer of this iteration, as op_dual_eval_kernel forms it, and its node sums
        const int loc = node - base;
        double gq[SPL], cn[SPL], g2[SPL];
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            dl[j] = valid[j] ? (loc < kNodeLoc ? dsh[0][loc][t0 + j] : a.sh_a[(int64_t)(t0 + j) * a.sh_m + node]) : 0.0;
            const double g0 = (double)revs_g0f(pe[j], pso[j], gm[j], inv_kf);
            const bool fr = valid[j] && g0 > dl[j];
            const double g = fr ? g0 - dl[j] : 0.0;
            pen[j] = (float)g;
            gq[j] = revs_q36(g);            // (g = 0 where clamped: zero addends)
            cn[j] = fr ? 1.0 : 0.0;
            g2[j] = revs_q32(g * g);
        }
        fold_add(0, gq, cn, g2);
    }
    const bool ev = h.ev != 0;
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const int t = t0 + j;
        win[j] = ev && valid[j] && (t >= h.start) && (t < h.end);
    }
    const ScanMasks<LPA> mk(lig);

    // ---- what a residence's PDHG passes need and no ADMM iteration changes (hoisted out of the
    // loop over the inner iterations) ----
    // Per-home scalars are computed redundantly by the LPA lanes of the group: use the 1-ulp
    // hardware reciprocal / square root instead of IEEE division (10 instructions each) -- step
    // sizes need no exactness, and b, x0 only to the PDHG tolerance.
    float inv_rate = 0.f, delta = 0.f, hi = 0.f, lo_last = 0.f, inv_kr = 0.f, wsum = 0.f;
    float tau = 0.f, sig = 0.f, inv1pt = 0.f, sd = 0.f;              // FULL_ROWS
    float tau1 = 0.f, inv1 = 0.f, sd1 = 0.f, ts = 0.f, sig1 = 0.f, inv_sig1 = 0.f, inv_d2 = 0.f;   // presolved
    float w[SPL];
    bool pd_infeasible = false;
    // (pd.tol: > 0 as given, 0 = automatic, < 0 = never before max_iter)
    const float pd_tol = a.pd.tol != 0.f ? a.pd.tol : ((FULL_ROWS || !(a.pd.polish & 1)) ? 1e-6f : 1e-4f);
    if constexpr (MODE == REVS_MODE_RELAXED_PDHG) {
        const float rate = ev ? h.rating : 1.f;
        inv_rate = __builtin_amdgcn_rcpf(rate);
        delta = ev ? h.rating * __builtin_amdgcn_rcpf(h.capacity) : 1.f;
        const int ws = max(h.start, 0), we = min(h.end, T);
        const float Tw = (float)max(we - ws, 1);
        hi = kSocMax - h.initial;
        lo_last = fmaxf(kSocTarget, h.initial) - h.initial;
        inv_kr = __builtin_amdgcn_rcpf(kappa * rate);
        if constexpr (FULL_ROWS) {
            const float inv_nK = __builtin_amdgcn_rcpf(delta * 0.63661977236758134f * (Tw + 1.0f));
            tau = (a.pd.tau_scale > 0.f ? a.pd.tau_scale : 0.25f) * inv_nK;
            sig = (a.pd.sigma_scale > 0.f ? a.pd.sigma_scale : 4.0f) * inv_nK;
            inv1pt = __builtin_amdgcn_rcpf(1.0f + tau);
            sd = sig * delta;
        } else {
            // Presolved form: with p >= 0 the SOC is nondecreasing, so of the rows
            // init <= s_t <= 1, s_T >= 0.9 only the terminal one can bind.  K is then the
            // single row delta * 1^T (||K|| = delta sqrt(T_w)), its dual one scalar per home,
            // and K x a group sum -- no scans.
            const float inv_nK1 = __builtin_amdgcn_rcpf(delta * __builtin_amdgcn_sqrtf(Tw));
            tau1 = (a.pd.tau_scale > 0.f ? a.pd.tau_scale : 0.5f) * inv_nK1;
            sig1 = (a.pd.sigma_scale > 0.f ? a.pd.sigma_scale : 2.0f) * inv_nK1;
            inv1 = __builtin_amdgcn_rcpf(1.0f + tau1);
            sd1 = sig1 * delta;
            ts = tau1 * inv1 * sd1;
            inv_sig1 = __builtin_amdgcn_rcpf(sig1);
            inv_d2 = __builtin_amdgcn_rcpf(delta * delta);
        }
#pragma unroll
        for (int j = 0; j < SPL; ++j) { w[j] = win[j] ? 1.f : 0.f; wsum += w[j]; }
        // the window cannot deliver the energy to 90 % SOC: "No solution found"
        wsum = group_sum<LPA>(wsum);
        pd_infeasible = ev && (lo_last > delta * wsum * (1.f + 1e-6f));
    }
    float yrow[SPL];      // FULL_ROWS: the carried multipliers of the SOC rows
    if constexpr (MODE == REVS_MODE_RELAXED_PDHG && FULL_ROWS) {
#pragma unroll
        for (int j = 0; j < SPL; ++j)
            yrow[j] = (a.y_state && valid[j] && ev) ? a.y_state[row + t0 + j] : 0.f;
    }
    float yy = ev ? yy_in : 0.f;      // presolved PDHG: the carried multiplier of the terminal row

    float p[SPL], gn[SPL], gmn[SPL], pe2[SPL];
    int status = 0, sticky = 0;
    float ddg = 0.f, dfh = 0.f;
    unsigned dmx_mine = 0u;                                  // lane i: the wavefront's max diff of inner iteration i
    float *diff_it = a.diff + (live ? agent : 0);
    double *pnext_it = a.p_next ? a.p_next + (int64_t)node * T + t0 : nullptr;     // this lane's first slot, this iteration's slice
    const bool need_pe2 = a.p_next != nullptr;
#pragma unroll 1
    for (int it = 0;; ++it) {
    float q[SPL];
    // No FMA contraction in the linear term and the keys: every slot must go through the
    // same rounding steps, so that equal inputs give equal keys ("ties to the earlier slot"
    // is the reference-visible rule) whatever the unrolled code looks like.
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
#pragma clang fp contract(off)
        // lpsolver.py:118-119  a_t = gamma_t + (kappa/2)(p_util_t + p_res_t)
        const float at = gm[j] + 0.5f * kappa * (pe[j] + pso[j]);
        // objective in p:  (kappa/2) p^2 + q p,  q = kappa*LOAD + c - a
        q[j] = kappa * L[j] + cst[j] - at;
        p[j] = 0.f;
    }
    status = 0;

    // A wavefront whose residences have no EV at all (the engine sorts like with like: 43 % of
    // the wavefronts of the bench workload) has nothing to solve: p = 0, status = 0 stand.
    if (!__any(ev)) {
    } else if constexpr (MODE == REVS_MODE_BINARY) {
        // p_t = e_t * rating, e_t binary (lpsolver.py:92-98).  Switching slot t on
        // costs delta_t = rating ((kappa/2) rating + q_t); take the nmin cheapest,
        // then more while delta < 0, up to nmax.
        int rank[SPL];
        int nwin = 0;
        bool neg[SPL];               // delta_t < 0: worth taking beyond the nmin slots the SOC rows demand
#pragma unroll
        for (int j = 0; j < SPL; ++j) nwin += win[j] ? 1 : 0;
        nwin = (int)group_sum<LPA>((float)nwin);
        if (a.pd.keys64) {
            // delta_t in DOUBLE, from the float inputs, in the oracle's order of operations (home_linear_term,
            // home_solve_binary): on identical inputs the on/off decision is the float64 restatement's for every
            // residence (tests/test_gpu_agent.py).  NOT the closed loop's default: there the exactly tied optima of the
            // reference's data (flat tariff blocks x repeated loads) stay tied only while a_t is formed by the very
            // float operations that formed the state -- G = -(kappa/2) g and (kappa/2)(P_est + P_sch) cancel EXACTLY in
            // float, and differ by G's own rounding in double: ties are then redrawn by that noise every iteration
            // and the 121144 feeder's run leaves the reference's band (mean diff[k] off by 4 x, measured in round 5).
            double key[SPL];
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
#pragma clang fp contract(off)
                const double kd = (double)kappa, rt = (double)h.rating;
                const double at = (double)gm[j] + 0.5 * kd * ((double)pe[j] + (double)pso[j]);
                const double qd = kd * (double)L[j] + (double)cst[j] - at;
                key[j] = win[j] ? rt * (0.5 * kd * rt + qd) : (double)INFINITY;
                neg[j] = key[j] < 0.0;
            }
            group_rank<LPA, SPL>(key, lig, rank);
        } else {
            float key[SPL];
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
#pragma clang fp contract(off)
                key[j] = win[j] ? h.rating * (0.5f * kappa * h.rating + q[j]) : INFINITY;
                neg[j] = key[j] < 0.f;
            }
            group_rank<LPA, SPL>(key, t0, rank);
        }
        const bool infeasible = ev && (h.nmin > h.nmax || h.nmin > nwin);
        status = infeasible ? 1 : 0;
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const bool take = win[j] & !infeasible &
                              ((rank[j] < h.nmin) | ((rank[j] < h.nmax) & neg[j]));
            p[j] = take ? h.rating : 0.f;
        }
    } else if constexpr (MODE == REVS_MODE_RELAXED_PDHG) {
        // PDHG on  min x^2/2 + b x,  0 <= x <= w,  lo <= K x <= hi,  x = p/rating,
        // K = delta * inclusive prefix sum (rows j: s_{j+1} - initial).
        float b[SPL], x[SPL], lo[SPL];
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            b[j] = q[j] * inv_kr;
            lo[j] = (t0 + j == T - 1) ? lo_last : 0.f;
            // warm start: primal from the previous schedule (P_sch[k] - LOAD), dual from
            // the previous iteration's multipliers when the caller keeps them
            x[j] = a.y_state ? clip3((pso[j] - L[j]) * inv_rate, 0.f, w[j]) : 0.f;
        }
        const bool infeasible = pd_infeasible;
        bool done = !ev | infeasible;
        bool unpolished = false;
        int iters = 0;
        const int check = max(a.pd.check, 1);
        if constexpr (!FULL_ROWS) {
            // KKT Newton steps on the terminal row's multiplier (described at the polish below): one routine,
            // run from the carried multiplier BEFORE PDHG (pd.polish bit 1: in the closed loop the multiplier
            // moves little between ADMM iterations, the active piece is the old one and one or two steps
            // land on the optimum -- a residence settled here skips PDHG, a wavefront of them the whole loop)
            // and from PDHG's iterate after it (bit 0).  PDHG stays the globally convergent solver behind
            // both: whatever the steps do not settle in 6 rounds goes through it.
            auto kkt_newton = [&](float &mu, bool &fine) {
#pragma unroll 1
                for (int r = 0;; ++r) {
                    float sxm = 0.f, u[SPL];
#pragma unroll
                    for (int j = 0; j < SPL; ++j) {
                        u[j] = fmaf(-delta, mu, -b[j]);
                        sxm += clip3(u[j], 0.f, w[j]);
                    }
                    sxm = group_sum<LPA>(sxm);
                    const float S = delta * sxm;
                    // (bitwise, not short-circuit: `a || (b && c)` on per-lane values compiles to nested exec-mask branches --
                    // four of them per round here, ~25 scalar instructions; as mask arithmetic it is three compares and two ops)
                    const bool mz = mu == 0.f;
                    const bool up = (mu > 0.f) | (mz & (S > hi));
                    const bool dn = (mu < 0.f) | (mz & (S < lo_last));
                    const float tgt = up ? hi : (dn ? lo_last : S);
                    const float resid = fabsf(S - tgt), tscale = fmaxf(1.f, fabsf(tgt));
                    fine = fine | (resid <= 5e-7f * tscale);
                    if (r == 6) {
                        // (round 5) S is a float sum over the window: at 48 free slots its own rounding reaches ~1e-6, and a
                        // residence whose row sits 10 ulps off its bound can neither pass the test above nor move (the
                        // step is below mu's resolution).  Its iterate here is Newton's, two orders closer than the
                        // PDHG iterate it would otherwise fall back to: accepted.
                        fine = fine | (resid <= 4e-6f * tscale);
                        break;
                    }
                    if (__all(fine)) break;
                    // The step: S is piecewise linear in mu with slope -delta^2 x (slots strictly inside their box),
                    // counted in the direction the step has to go (a slot sitting exactly on a bound moves one way only).
                    const bool rise = S > tgt;               // mu has to rise (S falls with mu)
                    // (falling: the same test on w - u -- "0 <= u < w" is "0 < w - u <= w" -- so that both directions
                    // are one select and two compares per slot, no branch)
                    float nf = 0.f, uu[SPL];
#pragma unroll
                    for (int j = 0; j < SPL; ++j) {
                        uu[j] = rise ? u[j] : w[j] - u[j];
                        nf += ((uu[j] > 0.f) & (uu[j] <= w[j])) ? 1.f : 0.f;
                    }
                    nf = group_sum<LPA>(nf);
                    float mun = nf > 0.f ? fmaf((S - tgt) * inv_d2, __builtin_amdgcn_rcpf(nf), mu) : mu;
                    if (__any(!fine & (nf == 0.f))) {
                        // (rare) flat in that direction: no slot moves until mu reaches the nearest breakpoint there --
                        // where the first slot leaves w (mu rising: mu_j = -(b_j + w_j) / delta) or leaves 0 (falling:
                        // mu_j = -b_j / delta)
                        float bp = -INFINITY;                // max over the candidates of -mu_j delta (rising) / +mu_j delta (falling)
#pragma unroll
                        for (int j = 0; j < SPL; ++j) {
                            const float m = rise ? b[j] + w[j] : -b[j];
                            bp = ((uu[j] > w[j]) & (w[j] > 0.f)) ? fmaxf(bp, m) : bp;
                        }
                        bp = group_max<LPA>(bp) * (delta * inv_d2);
                        // (a hair beyond it, so that the slot counts as inside its box whatever the rounding)
                        const float hair = 1e-6f * (fabsf(bp) + 1.f);
                        const float jump = rise ? hair - bp : bp - hair;
                        mun = ((nf == 0.f) & (bp > -INFINITY)) ? jump : mun;
                    }
                    mun = up ? fmaxf(mun, 0.f) : (dn ? fminf(mun, 0.f) : mun);     // a multiplier keeps its sign
                    mu = fine ? mu : mun;
                }
            };
            auto kkt_take = [&](float mu, bool fine) {
#pragma unroll
                for (int j = 0; j < SPL; ++j) x[j] = fine ? clip3(fmaf(-delta, mu, -b[j]), 0.f, w[j]) : x[j];
                yy = fine ? mu * inv_sig1 : yy;
            };
            float mu_pre = sig1 * yy;
            bool pre = false;
            if (a.pd.polish & 2) {
                pre = !ev | infeasible;
                kkt_newton(mu_pre, pre);
                kkt_take(mu_pre, pre);
                done = done | pre;
            }
            // The sweep is VALU-issue bound and half of its instructions are this loop, so the
            // iteration is arranged for the fewest operations:
            //   x+ = clip((x - tau (sd y + b)) / (1 + tau), 0, w) = clip(a x - cb - s, 0, w),
            //        a = 1/(1+tau), cb = tau a b (per slot, once), s = tau a sd y (one multiply)
            //   K (2 x+ - x) = delta (2 sum x+ - sum x), with sum x carried from the last pass
            float cb[SPL], sx = 0.f;
#pragma unroll
            for (int j = 0; j < SPL; ++j) { cb[j] = tau1 * inv1 * b[j]; sx += x[j]; }
            auto iterate1 = [&](auto res_tag) -> float {
                constexpr bool RES = decltype(res_tag)::value;
                const float s = ts * yy;
                float sn = 0.f, dmax = 0.f;
#pragma unroll
                for (int j = 0; j < SPL; ++j) {
                    const float xn = clip3(fmaf(inv1, x[j], -cb[j]) - s, 0.f, w[j]);
                    sn += xn;
                    if constexpr (RES) dmax = fmaxf(dmax, fabsf(xn - x[j]));
                    x[j] = xn;
                }
                const float acc = fmaf(2.0f, sn, -sx);
                sx = sn;
                const float v = fmaf(delta, group_sum<LPA>(acc), yy);
                const float yn = v - clip3(v, lo_last, hi);
                if constexpr (RES) dmax = fmaxf(dmax, fabsf(yn - yy));
                yy = yn;
                return dmax;
            };
            for (int k = 0; k < a.pd.max_iter; k += check) {
                if (__all(done)) break;      // wave-uniform exit every wave reaches
                for (int c = 1; c < check; ++c) iterate1(std::false_type{});
                float res = iterate1(std::true_type{});
                iters += done ? 0 : check;
                res = group_max_nonneg<LPA>(res);
                done = done | (res <= pd_tol);
            }
            // KKT polish on the piece PDHG has identified.  For the terminal row's multiplier mu
            // (= sigma yy) the minimiser is x(mu) = clip(-b - delta mu, 0, w) and delta sum x(mu) is
            // piecewise linear and nonincreasing in mu: semismooth Newton steps on "the row sits on
            // the bound its multiplier's sign names" (or: mu = 0 and the row inside its bounds) land
            // on the exact optimum of the identified piece -- the first-order method's tail
            // (schedules 2e-5 kW off per solve at a 1e-6 step test, 1e-3 kW in the closed loop) is
            // gone for the price of ~2 passes, and PDHG itself may stop much earlier (pd.tol).
            if ((a.pd.polish & 1) && !__all(pre)) {
                float mu = pre ? mu_pre : sig1 * yy;
                bool fine = pre | !ev | infeasible;       // the row's KKT conditions hold at x(mu) to float rounding
                // (a Newton step is exact while the free slots stay free: one or two steps from where
                // PDHG stopped; the loop ends as soon as every residence of the wavefront is there)
                kkt_newton(mu, fine);
                // (a residence the steps did not settle keeps PDHG's iterate -- at PDHG's own, looser tolerance:
                // status bit 2 says so)
                kkt_take(mu, fine);
                unpolished = !fine;
            }
            yy = ev ? yy : 0.f;
        } else {
        // y of padded slots must stay 0: give them lo = hi = 0 ... no: v - clip(v,0,0) = v.
        // Instead padded slots get lo = -inf, hi = +inf, so y = v - v = 0.
        float hiv[SPL];
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            hiv[j] = valid[j] ? hi : INFINITY;
            lo[j] = valid[j] ? lo[j] : -INFINITY;
        }
        // one PDHG iteration; with RES the largest step of this iteration is returned
        auto iterate = [&](auto res_tag) -> float {
            constexpr bool RES = decltype(res_tag)::value;
            // K^T y : inclusive suffix sum of y
            float sfx[SPL], acc = 0.f;
#pragma unroll
            for (int j = SPL - 1; j >= 0; --j) { acc += yrow[j]; sfx[j] = acc; }
            const float so = group_excl_suffix<LPA>(acc, lig, mk);
            float xb[SPL], dmax = 0.f;
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
                const float kty = sd * (sfx[j] + so);
                const float xn = clip3((x[j] - tau * (kty + b[j])) * inv1pt, 0.f, w[j]);
                xb[j] = xn + (xn - x[j]);
                if constexpr (RES) dmax = fmaxf(dmax, fabsf(xn - x[j]));
                x[j] = xn;
            }
            // K xbar : inclusive prefix sum
            float pfx[SPL];
            acc = 0.f;
#pragma unroll
            for (int j = 0; j < SPL; ++j) { acc += xb[j]; pfx[j] = acc; }
            const float po = group_excl_prefix<LPA>(acc, lig, mk);
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
                const float v = yrow[j] + delta * (pfx[j] + po);
                const float yn = v - clip3(v, lo[j], hiv[j]);
                if constexpr (RES) dmax = fmaxf(dmax, fabsf(yn - yrow[j]));
                yrow[j] = yn;
            }
            return dmax;
        };
        // Homes of one wavefront iterate together until all of them have converged (a
        // converged home keeps iterating: it only moves closer to its optimum); `iters`
        // records when each home first met the tolerance.
        for (int k = 0; k < a.pd.max_iter; k += check) {
            if (__all(done)) break;          // wave-uniform exit every wave reaches
            for (int c = 1; c < check; ++c) iterate(std::false_type{});
            float res = iterate(std::true_type{});
            iters += done ? 0 : check;
            res = group_max<LPA>(res);
            done = done | (res <= pd_tol);
        }
#pragma unroll
        for (int j = 0; j < SPL; ++j) yrow[j] = ev ? yrow[j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < SPL; ++j) p[j] = (ev & !infeasible) ? x[j] * h.rating : 0.f;
        // bit 1: the iteration cap was reached before the tolerance (the schedule is then only
        // as good as max_iter passes make it: surfaced, not silently accepted)
        // bit 2: the KKT polish did not settle (the schedule is PDHG's iterate at its step tolerance)
        status = (iters << 8) | (unpolished ? 4 : 0) | ((ev & !infeasible & !done) ? 2 : 0) | (infeasible ? 1 : 0);
    } else {
        // closed form: p_t = clip(u_t + nu, 0, ub_t), u = -q/kappa; nu is the
        // multiplier of the terminal SOC rows (the only ones that can bind when
        // p >= 0), found by bisection then solved exactly on the identified piece.
        float u[SPL], ub[SPL];
        float cap_sum = 0.f, umax = 0.f;
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            u[j] = -q[j] / kappa;
            ub[j] = win[j] ? h.rating : 0.f;
            cap_sum += ub[j];
            umax = fmaxf(umax, win[j] ? fabsf(u[j]) : 0.f);
        }
        cap_sum = group_sum<LPA>(cap_sum);
        umax = group_max<LPA>(umax);
        const float Elo = ev ? (fmaxf(kSocTarget, h.initial) - h.initial) * h.capacity : 0.f;
        const float Ehi = ev ? (kSocMax - h.initial) * h.capacity : 0.f;
        auto fsum = [&](float nu) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < SPL; ++j) s += fminf(fmaxf(u[j] + nu, 0.f), ub[j]);
            return group_sum<LPA>(s);
        };
        const float s0 = fsum(0.f);
        const bool infeasible = ev && (Elo > cap_sum * (1.f + 1e-6f));
        status = infeasible ? 1 : 0;
        float nu = 0.f;
        const bool need_lo = s0 < Elo, need_hi = s0 > Ehi;
        if (ev && !infeasible && (need_lo || need_hi)) {
            const float tgt = need_lo ? Elo : Ehi;
            float lo_ = -(umax + h.rating + 1.f), hi_ = -lo_;
            for (int i = 0; i < 40; ++i) {
                const float mid = 0.5f * (lo_ + hi_);
                const bool below = fsum(mid) < tgt;
                lo_ = below ? mid : lo_;
                hi_ = below ? hi_ : mid;
            }
            nu = 0.5f * (lo_ + hi_);
            float nfree = 0.f, fixed = 0.f;
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
                const float xv = u[j] + nu;
                const bool fr = xv > 0.f && xv < ub[j];
                nfree += fr ? 1.f : 0.f;
                fixed += fr ? u[j] : ((xv >= ub[j]) ? ub[j] : 0.f);
            }
            nfree = group_sum<LPA>(nfree);
            fixed = group_sum<LPA>(fixed);
            if (nfree > 0.f) nu = (tgt - fixed) / nfree;
        }
#pragma unroll
        for (int j = 0; j < SPL; ++j)
            p[j] = (ev && !infeasible) ? fminf(fmaxf(u[j] + nu, 0.f), ub[j]) : 0.f;
    }

    // ---- g, dual update, residual terms of this iteration (lpsolver.py:275-284) ----
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const float g = p[j] + L[j];                        // lpsolver.py:64-65
        const float chk = pen[j] - g;                       // lpsolver.py:280
        gn[j] = g;
        gmn[j] = gm[j] + 0.5f * kappa * chk;                // lpsolver.py:282
        ss += valid[j] ? chk * chk : 0.f;
    }
    // next evaluation's home pass (same arithmetic as op_dual_eval_kernel with d = 0)
    if (need_pe2) {
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const float g0 = revs_g0f(pen[j], gn[j], gmn[j], inv_kf);
            pe2[j] = (valid[j] & (g0 > 0.f)) ? g0 : 0.f;
        }
    }
    if constexpr (CHAIN) {      // the evaluation of the same multipliers on the state just produced
        const int loc = node - base;
        double gq[SPL], cn[SPL], g2[SPL];
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const double g0 = (double)revs_g0f(pen[j], gn[j], gmn[j], inv_kf);
            const double ds = valid[j] ? (loc < kNodeLoc ? dsh[1][loc][t0 + j] : a.sh_b[(int64_t)(t0 + j) * a.sh_m + node]) : 0.0;
            const bool fr = valid[j] && g0 > ds;
            const double g = fr ? g0 - ds : 0.0;
            gq[j] = revs_q36(g);
            cn[j] = fr ? 1.0 : 0.0;
            g2[j] = revs_q32(g * g);
        }
        fold_add(1, gq, cn, g2);
    }
    // per-home residual terms; the norms over all homes are folded by revs_residual_finalize
    // only when somebody asks for them (no workgroup reduction in the sweep, which is VALU
    // issue bound)
    const float ssg = group_sum<LPA>(ss);
    // lpsolver.py:284; hardware sqrt and reciprocal (1 ulp each) instead of the IEEE sequences
    dfh = live ? __builtin_amdgcn_sqrtf(ssg) * __builtin_amdgcn_rcpf((float)T) : 0.f;
    sticky |= status & 7;
    if (live && lig == 0) *diff_it = dfh;
    diff_it += a.diff_stride;
    if (a.dmax_out) {
        // the wavefront's largest diff of this iteration, kept by lane `it` (dfh >= 0: the bit patterns order; every
        // lane of a group holds its residence's value): no LDS traffic and no scalar code inside the loop -- an
        // atomicMax from the eight leading lanes compiles to a scalar loop over them, 56 SALU + 8 v_readlane per
        // iteration -- and one LDS atomic per lane and launch behind it
        const unsigned m = wave_max_groups_u<LPA>(__float_as_uint(dfh));
        dmx_mine = (tid & 63) == it ? m : dmx_mine;
    }
    if (a.p_next && live) {
        // (floats widened to double: the node sums are exact, whatever the order; a slot that does not exist or an
        // answer of zero adds +0.0 to an accumulator nobody reads / exactly nothing: straight-line LDS adds)
        const int loc = node - base;
        if (loc < kNodeLoc) {
            // (round 5, tuning builds: without these adds the launch is 4-5 % shorter -- they are not what bounds it; summing
            // the wavefront's residences across its lane groups first, one add per slot from 8 lanes instead of 64: +14 %)
#pragma unroll
            for (int j = 0; j < SPL; ++j) unsafeAtomicAdd(&nacc[it][loc][t0 + j], (double)pe2[j]);
        } else {
#pragma unroll
            for (int j = 0; j < SPL; ++j)
                if (pe2[j] > 0.f) unsafeAtomicAdd(pnext_it + j, (double)pe2[j]);
        }
    }
    if (a.p_next) pnext_it += a.slice_stride;
    if (!MULTI || it + 1 >= kin) break;
    // the next iteration of the same residences: P_est[g+1] = this iteration's estimate, P_est[g+2]
    // the one just prepared -- exactly the floats a launch per iteration would write and read back
#pragma unroll
    for (int j = 0; j < SPL; ++j) { pe[j] = pen[j]; pen[j] = pe2[j]; pso[j] = gn[j]; gm[j] = gmn[j]; }
    }

    // ---- stores: the state after the last inner iteration ----
    {   // kappa |dP_sch|'s per-home term of the last iteration (the dual residual, on request)
        float dd = 0.f;
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const float dg = gn[j] - pso[j];
            dd += valid[j] ? dg * dg : 0.f;
        }
        ddg = group_sum<LPA>(dd);
    }
    const int64_t crow = agent * (int64_t)(T + 1);
    float socv[SPL];
    if (a.c_out) {      // the SOC trajectory (a prefix sum over the slots) only where it is returned
        float pacc = 0.f, pfx[SPL];
#pragma unroll
        for (int j = 0; j < SPL; ++j) { pacc += p[j]; pfx[j] = pacc; }
        const float poff = group_excl_prefix<LPA>(pacc, lig, mk);
        const float invcap = ev ? __builtin_amdgcn_rcpf(h.capacity) : 0.f;    // (1 ulp; SOC only)
#pragma unroll
        for (int j = 0; j < SPL; ++j) socv[j] = ev ? h.initial + (pfx[j] + poff) * invcap : 0.f;
    }
    if (full) {
        st_pack<SPL>(a.ps_out + row + t0, gn);
        st_pack<SPL>(a.gam_out + row + t0, gmn);
        if (a.pe_out) st_pack<SPL>(a.pe_out + row + t0, pen);
        if (a.s_out) st_pack<SPL>(a.s_out + row + t0, p);
        if (a.c_out) st_pack<SPL>(a.c_out + crow + t0 + 1, socv);
        if (a.pe2_out) st_pack<SPL>(a.pe2_out + row + t0, pe2);
    } else {
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const int t = t0 + j;
            if (valid[j]) {
                const int64_t o = row + t;
                a.ps_out[o] = gn[j];
                a.gam_out[o] = gmn[j];
                if (a.pe_out) a.pe_out[o] = pen[j];
                if (a.s_out) a.s_out[o] = p[j];
                if (a.c_out) a.c_out[crow + t + 1] = socv[j];
                if (a.pe2_out) a.pe2_out[o] = pe2[j];
            }
        }
    }
    if constexpr (MODE == REVS_MODE_RELAXED_PDHG && FULL_ROWS) {
#pragma unroll
        for (int j = 0; j < SPL; ++j)
            if (a.y_out && valid[j]) a.y_out[row + t0 + j] = yrow[j];
    }
    if (live && lig == 0) {
        if constexpr (MODE == REVS_MODE_RELAXED_PDHG && !FULL_ROWS)
            if (a.y_out) a.y_out[agent] = yy;
        if (a.c_out) a.c_out[crow] = ev ? h.initial : 0.f;
        a.dsq[agent] = ddg;
        const int st = status | sticky;
        if (a.status) a.status[agent] = st;
        if (a.flags && (st & 3))         // rare: straight into the host's word
            __hip_atomic_fetch_or(a.flags, (unsigned int)(st & 3), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (a.p_next) {
        if (a.dmax_out && (tid & 63) < kin && dmx_mine != 0u) atomicMax(&dmx[tid & 63], dmx_mine);
        // No workgroup barrier in front of the flush (round 5): a wavefront that is done counts itself and leaves; the LAST
        // of the workgroup's wavefronts to arrive flushes the accumulators for all.  (A wavefront of residences without an
        // EV runs a third of the instructions of one that solves QPs -- the engine's order puts like with like, so most
        // workgroups hold both kinds; at a barrier the quick ones sat on their registers until the slow ones came, and
        // nearly half of all wave-cycles were spent parked: SQ_WAIT_ANY 187 M of SQ_WAVE_CYCLES 415 M, r04.)  Ordering:
        // a wavefront's LDS operations execute in issue order, so its adds into nacc / dmx are performed before its
        // increment of `arrived` is; the last arriver's reads are issued after its own increment has returned.
        unsigned int before = 0u;
        if ((tid & 63) == 0) before = atomicAdd(&arrived, 1u);
        before = (unsigned int)__builtin_amdgcn_readfirstlane((int)before);
        if (before != kBlock / 64 - 1) return;
        const int ln = tid & 63;
        for (int r = ln; r < kNodeLoc * T; r += 64) {
            const int l = r / T, t = r - l * T;
            double *dst = a.p_next + (int64_t)(base + l) * T + t;
            for (int itq = 0; itq < kin; ++itq, dst += a.slice_stride) {
                const double v = nacc[itq][l][t];
                if (v != 0.0) unsafeAtomicAdd(dst, v);
            }
        }
        if (a.dmax_out && ln < kin && dmx[ln] != 0u)
            // (REVS_DMAX_SLOTS addresses per iteration: thousands of workgroups on ONE address serialise
            // in the L2 -- measured 27 us per launch at 3 125 workgroups)
            __hip_atomic_fetch_max((unsigned long long *)(a.dmax_out + (long long)ln * a.slice_stride +
                                                          (bid & (REVS_DMAX_SLOTS - 1))),
                                   (unsigned long long)__double_as_longlong((double)__uint_as_float(dmx[ln])),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if constexpr (CHAIN) {
#ifdef REVS_CHAIN_BARRIER_FLUSH       // (tuning builds: the form of rounds 3-4, every thread flushes behind a workgroup barrier)
        __syncthreads();
        for (int i = tid; i < 2 * T * kNodeLoc * 3; i += kBlock) {
            const int f = i / (T * kNodeLoc * 3), r0 = i - f * (T * kNodeLoc * 3);
            const int tl = r0 / 3, q = r0 - tl * 3;
            const int t = tl / kNodeLoc, l = tl - t * kNodeLoc;
            const double v = facc[f][q][l][t];
            if (v != 0.0 && base + l < a.sh_m)
                unsafeAtomicAdd((f ? a.fold_b : a.fold_a) + 4 * ((int64_t)t * a.sh_m + base + l) + q,
                                q == 2 ? -0.5 * a.sh_kappa * v : v);
        }
#else
        // as the multi-iteration sweep above: no barrier, the last wavefront of the workgroup to arrive flushes for all
        // (fold_a / fold_b are slot-major, [T][m][4] = {p, N, q, 0}: a slot's block is contiguous for the operator launch's
        // coalesced read; this workgroup's kNodeLoc nodes x 3 quantities of a slot sit on one or two cache lines)
        unsigned int before = 0u;
        if ((tid & 63) == 0) before = atomicAdd(&arrived, 1u);
        before = (unsigned int)__builtin_amdgcn_readfirstlane((int)before);
        if (before != kBlock / 64 - 1) return;
        const int ln = tid & 63;
        // lane -> (node l, quantity q) of a slot: kNodeLoc * 3 <= 12 lanes per slot, 64 / 12 = 5 slots per trip
        constexpr int kPerSlot = kNodeLoc * 3, kSlotsPerTrip = 64 / kPerSlot;
        const int sl = ln / kPerSlot, lq = ln - sl * kPerSlot, l = lq / 3, q = lq - 3 * l;
        if (sl < kSlotsPerTrip && base + l < a.sh_m) {
            for (int t = sl; t < T; t += kSlotsPerTrip) {
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const double v = facc[f][q][l][t];
                    if (v != 0.0)
                        unsafeAtomicAdd((f ? a.fold_b : a.fold_a) + 4 * ((int64_t)t * a.sh_m + base + l) + q,
                                        q == 2 ? -0.5 * a.sh_kappa * v : v);
                }
            }
        }
#endif
    }
}

// Residual norms from the per-home terms, on request only (not part of the sweep):
// stage 1, workgroup b folds the contiguous chunk b of homes into scratch[b] =
// {sum (T diff)^2, sum dsq, max diff} in double, fixed order; stage 2 folds the chunks.
// Bitwise reproducible.
__global__ __launch_bounds__(256) void residual_chunks_kernel(
        const float *__restrict__ diff, const float *__restrict__ dsq, int64_t n, int32_t T,
        double *__restrict__ scratch) {
    const int64_t chunk = (n + gridDim.x - 1) / gridDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * chunk, i1 = i0 + chunk < n ? i0 + chunk : n;
    double s0 = 0.0, s1 = 0.0, mx = 0.0;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const double d = (double)diff[i] * T;
        s0 += d * d;
        s1 += (double)dsq[i];
        mx = fmax(mx, (double)diff[i]);
    }
    for (int d = 32; d >= 1; d >>= 1) {
        s0 += __shfl_xor(s0, d, 64);
        s1 += __shfl_xor(s1, d, 64);
        mx = fmax(mx, __shfl_xor(mx, d, 64));
    }
    __shared__ double red[3][4];
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; red[2][threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        scratch[3 * blockIdx.x + 0] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        scratch[3 * blockIdx.x + 1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        scratch[3 * blockIdx.x + 2] = fmax(fmax(red[2][0], red[2][1]), fmax(red[2][2], red[2][3]));
    }
}

__global__ __launch_bounds__(64) void residual_finalize_kernel(
        const double *__restrict__ scratch, int nchunk, float kappa, float eps, float *out) {
    if (threadIdx.x == 0) {
        double s0 = 0.0, s1 = 0.0, mx = 0.0;
        for (int b = 0; b < nchunk; ++b) {
            s0 += scratch[3 * b]; s1 += scratch[3 * b + 1]; mx = fmax(mx, scratch[3 * b + 2]);
        }
        out[0] = (float)sqrt(s0);
        out[1] = kappa * (float)sqrt(s1);
        out[2] = (float)mx;
        out[3] = ((float)mx <= eps) ? 1.0f : 0.0f;
    }
}

// Individual mode (lpsolver.py:407-460): switching slot t on changes
// 0.01 tariff.g + 0.99 (1 - s_T) by 0.01 c_t rating - 0.99 rating/capacity;
// take the cheapest slots while that is negative, up to nmax.
template <int LPA, int SPL>
__global__ __launch_bounds__(kBlock) void residence_kernel(
        int64_t n, int32_t T, const float *tariff, const revs_home_t *homes,
        const float *load, float *p_out, float *soc_out, float *g_out) {
    const int tid = threadIdx.x;
    const int lig = tid & (LPA - 1);
    const int64_t agent = (int64_t)blockIdx.x * (kBlock / LPA) + tid / LPA;
    const bool live = agent < n;
    const int t0 = lig * SPL;
    revs_home_t h;
    if (live) h = homes[agent];
    else { h.ev = 0; h.start = 0; h.end = 0; h.nmin = 0; h.nmax = 0; h.rating = 0.f; h.capacity = 1.f; h.initial = 0.f; }
    const bool ev = h.ev != 0;
    double key[SPL];
    float p[SPL], pfx[SPL];
    int rank[SPL];
    bool win[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
#pragma clang fp contract(off)
        const int t = t0 + j;
        win[j] = ev && live && t < T && t >= h.start && t < h.end;
        const double c = (t < T) ? (double)tariff[t] : 0.0;
        // (double, the oracle's order of operations: solve_residence)
        key[j] = win[j] ? 0.01 * c * (double)h.rating - 0.99 * ((double)h.rating / (double)h.capacity) : (double)INFINITY;
    }
    group_rank<LPA, SPL>(key, lig, rank);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        p[j] = (win[j] && rank[j] < h.nmax && key[j] < 0.0) ? h.rating : 0.f;
        acc += p[j];
        pfx[j] = acc;
    }
    const ScanMasks<LPA> mk(lig);
    const float poff = group_excl_prefix<LPA>(acc, lig, mk);
    const float invcap = ev ? 1.f / h.capacity : 0.f;
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const int t = t0 + j;
        if (live && t < T) {
            const int64_t o = agent * (int64_t)T + t;
            p_out[o] = p[j];
            g_out[o] = p[j] + load[o];
            soc_out[agent * (int64_t)(T + 1) + t + 1] = ev ? h.initial + (pfx[j] + poff) * invcap : 0.f;
        }
    }
    if (live && lig == 0) soc_out[agent * (int64_t)(T + 1)] = ev ? h.initial : 0.f;
}

// ---- dispatch ---------------------------------------------------------------
struct Shape { int lpa, spl; };
static Shape pick_shape(int T, int lanes = 0) {
    // revs_pdhg_t::lanes: the wide shapes for a GPU that holds few residences
    if (T <= 32 && lanes == 32) return {32, 1};
    if (T <= 32 && lanes == 16) return {16, 2};
    // REVS_AGENT_SHAPE=LPAxSPL overrides the mapping for T <= 24 and 64 < T <= 96 (tuning
    // hook; read once: this runs twice per launch on the host's critical path)
    static const Shape env = [] {
        Shape e{0, 0};
        if (const char *v = getenv("REVS_AGENT_SHAPE")) {
            int l = 0, p = 0;
            if (sscanf(v, "%dx%d", &l, &p) == 2) e = Shape{l, p};
        }
        return e;
    }();
    if (env.lpa > 0 && env.lpa * env.spl >= T) {
        const int l = env.lpa, p = env.spl;
        if (T <= 24 && ((l == 8 && p == 3) || (l == 4 && p == 6) || (l == 2 && p == 12))) return env;
        if (T > 64 && T <= 96 && ((l == 32 && p == 3) || (l == 16 && p == 6) || (l == 8 && p == 12)))
            return env;
    }
    if (T <= 8) return {8, 1};
    if (T <= 16) return {8, 2};
    if (T <= 24) return {8, 3};
    if (T <= 32) return {8, 4};
    if (T <= 48) return {16, 3};
    if (T <= 64) return {16, 4};
    if (T <= 96) return {16, 6};     // (measured at T = 96: 78.8 us against 82.7 / 82.4 for 32x3 / 8x12)
    if (T <= 128) return {32, 4};
    return {64, 3};
}

template <int LPA, int SPL>
static void launch_agent(const AgentArgs &a, int mode, dim3 grid, hipStream_t s) {
    // dynamic LDS: only a launch whose first workgroups run the tree form of R p needs any
    const size_t lds = a.tree.n > 0 ? tree_lds_bytes(a.tree.n) : 0;
    if (a.sh_a) {        // the folded chain's sweep
        switch (mode) {
            case REVS_MODE_BINARY:
                hipLaunchKernelGGL((agent_step_kernel<LPA, SPL, REVS_MODE_BINARY, false, false, true>), grid, dim3(kBlock), 0, s, a);
                break;
            case REVS_MODE_RELAXED_PDHG:
                hipLaunchKernelGGL((agent_step_kernel<LPA, SPL, REVS_MODE_RELAXED_PDHG, false, false, true>), grid, dim3(kBlock), 0, s, a);
                break;
            default:
                hipLaunchKernelGGL((agent_step_kernel<LPA, SPL, REVS_MODE_RELAXED_EXACT, false, false, true>), grid, dim3(kBlock), 0, s, a);
                break;
        }
        return;
    }
    if (a.kin > 1) {     // several iterations per launch (no verdict workgroups in these launches)
        switch (mode) {
            case REVS_MODE_BINARY:
                hipLaunchKernelGGL((agent_step_kernel<LPA, SPL, REVS_MODE_BINARY, false, true>), grid, dim3(kBlock), 0, s, a);
                break;
            case REVS_MODE_RELAXED_PDHG:
                if (a.pd.full_rows)
                    hipLaunchKernelGGL((agent_step_kernel<LPA, SPL, REVS_MODE_RELAXED_PDHG, true, true>), grid, dim3(kBlock), 0, s, a);
                else
                    hipLaunchKernelGGL((agent_step_kernel<LPA, SPL, REVS_MODE_RELAXED_PDHG, false, true>), grid, dim3(kBlock), 0, s, a);
                break;
            default:
                hipLaunchKernelGGL((agent_step_kernel<LPA, SPL, REVS_MODE_RELAXED_EXACT, false, true>), grid, dim3(kBlock), 0, s, a);
                break;
        }
        return;
    }
    switch (mode) {
        case REVS_MODE_BINARY:
            hipLaunchKernelGGL((agent_step_kernel<LPA, SPL, REVS_MODE_BINARY>), grid, dim3(kBlock), lds, s, a);
            break;
        case REVS_MODE_RELAXED_PDHG:
            if (a.pd.full_rows)
                hipLaunchKernelGGL((agent_step_kernel<LPA, SPL, REVS_MODE_RELAXED_PDHG, true>), grid, dim3(kBlock), lds, s, a);
            else
                hipLaunchKernelGGL((agent_step_kernel<LPA, SPL, REVS_MODE_RELAXED_PDHG, false>), grid, dim3(kBlock), lds, s, a);
            break;
        default:
            hipLaunchKernelGGL((agent_step_kernel<LPA, SPL, REVS_MODE_RELAXED_EXACT>), grid, dim3(kBlock), lds, s, a);
            break;
    }
}

#define REVS_FOR_SHAPE(sh, CALL)                                   \
    do {                                                           \
        if (sh.lpa == 8 && sh.spl == 1) { CALL(8, 1); }            \
        else if (sh.lpa == 32 && sh.spl == 1) { CALL(32, 1); }     \
        else if (sh.lpa == 16 && sh.spl == 2) { CALL(16, 2); }     \
        else if (sh.lpa == 8 && sh.spl == 2) { CALL(8, 2); }       \
        else if (sh.lpa == 8 && sh.spl == 3) { CALL(8, 3); }       \
        else if (sh.lpa == 4 && sh.spl == 6) { CALL(4, 6); }       \
        else if (sh.lpa == 2 && sh.spl == 12) { CALL(2, 12); }     \
        else if (sh.lpa == 8 && sh.spl == 4) { CALL(8, 4); }       \
        else if (sh.lpa == 16 && sh.spl == 3) { CALL(16, 3); }     \
        else if (sh.lpa == 16 && sh.spl == 4) { CALL(16, 4); }     \
        else if (sh.lpa == 32 && sh.spl == 3) { CALL(32, 3); }     \
        else if (sh.lpa == 32 && sh.spl == 4) { CALL(32, 4); }     \
        else if (sh.lpa == 16 && sh.spl == 6) { CALL(16, 6); }     \
        else if (sh.lpa == 8 && sh.spl == 12) { CALL(8, 12); }     \
        else { CALL(64, 3); }                                      \
    } while (0)

}  // namespace revs

using namespace revs;

extern "C" void revs_pdhg_defaults(revs_pdhg_t *o) {
    o->max_iter = 4000;
    o->check = 4;
    o->tol = 0.f;           // 0 = automatic: 1e-4 with the polish behind it, 1e-6 without
    o->tau_scale = 0.f;      // 0 = automatic: 0.5 / 2.0 presolved, 0.25 / 4.0 with full_rows
    o->sigma_scale = 0.f;
    o->full_rows = 0;
    o->polish = 1;
    o->lanes = 0;
    o->keys64 = 0;
}

static int64_t agent_num_blocks(int64_t n_homes, int32_t T, int lanes) {
    if (n_homes <= 0 || T <= 0 || T > REVS_MAX_T) return 0;
    const Shape sh = pick_shape(T, lanes);
    const int64_t per = kBlock / sh.lpa;
    return (n_homes + per - 1) / per;
}

extern "C" int32_t revs_residual_num_chunks(int64_t n_homes) {
    if (n_homes <= 0) return 0;
    const int64_t c = (n_homes + 4095) / 4096;
    return (int32_t)(c < 256 ? c : 256);
}

static int agent_step_impl(int64_t n_homes, int32_t T, const float *cost,
                           const revs_home_t *homes, const float *load,
                           const float *p_est_old, const float *p_est_new,
                           const float *p_sch, const float *gamma, float *p_sch_out,
                           float *gamma_out, float *s_out, float *c_out, float *diff,
                           float *dsq, int32_t *status, float *pdhg_dual,
                           float kappa, int32_t mode, const revs_pdhg_t *pdhg_host,
                           const SelectArgs *sel, const int32_t *node_of, double *p_next,
                           float *pe2_out, void *stream, const StreamExtra *sx = nullptr,
                           unsigned int *flags = nullptr, const ChainFold *cf = nullptr) {
    REVS_REQUIRE(n_homes > 0, "revs_agent_step: n_homes=%lld", (long long)n_homes);
    REVS_REQUIRE(T > 0 && T <= REVS_MAX_T, "revs_agent_step: T=%d outside 1..%d", T, REVS_MAX_T);
    REVS_REQUIRE(cost && homes && load && p_est_old && p_sch && gamma && p_sch_out &&
                 gamma_out && diff && dsq, "revs_agent_step: null pointer argument");
    REVS_REQUIRE(mode >= 0 && mode <= 2, "revs_agent_step: mode=%d", mode);
    REVS_REQUIRE(kappa > 0.f, "revs_agent_step: kappa=%g must be positive", (double)kappa);
    AgentArgs a;
    a.n = n_homes; a.T = T; a.cost = cost; a.homes = homes; a.load = load;
    a.pe_old = p_est_old; a.pe_new = p_est_new;
    a.ps = const_cast<float *>(p_sch); a.gam = const_cast<float *>(gamma);
    a.ps_out = p_sch_out; a.gam_out = gamma_out;
    a.s_out = s_out; a.c_out = c_out; a.diff = diff; a.dsq = dsq;
    a.status = status; a.y_state = pdhg_dual; a.kappa = kappa;
    a.nsel = 0;
    a.sel = SelectArgs{};
    if (sel) { a.nsel = sel->T; a.sel = *sel; }
    a.node_of = node_of; a.p_next = p_next; a.pe2_out = pe2_out;
    REVS_REQUIRE(!p_next || (node_of && (pe2_out || (sx && !sx->verdict))), "revs_agent_step: node_of / pe2_out missing");
    a.ctl = nullptr; a.seq = 0; a.base_seq = 0; a.tree = TreeArgs{}; a.p_in = nullptr; a.p_zero = nullptr;
    a.vtol = 0.0; a.rec = nullptr; a.flags = flags; a.m = 0;
    a.kin = 1; a.pe_out = nullptr; a.y_out = pdhg_dual; a.slice_stride = 0; a.diff_stride = 0; a.dmax_out = nullptr;
    a.sh_a = nullptr; a.sh_b = nullptr; a.sh_m = 0; a.sh_kappa = 0.0;
    a.fold_a = nullptr; a.fold_b = nullptr;
    a.wg_order = nullptr;
    if (cf) {
        REVS_REQUIRE(cf->sh_a && cf->sh_b && cf->m > 0 && cf->kappa > 0 && cf->fold_a && cf->fold_b &&
                     cf->fold_a != cf->fold_b && cf->pe_out && node_of && !p_est_new && !p_next && !sel && !sx &&
                     !(pdhg_host && pdhg_host->full_rows),
                     "revs_agent_step: bad argument of the folded chain's sweep");
        a.sh_a = cf->sh_a; a.sh_b = cf->sh_b; a.sh_m = cf->m; a.sh_kappa = cf->kappa;
        a.fold_a = cf->fold_a; a.fold_b = cf->fold_b; a.pe_out = cf->pe_out;
        if (cf->y_out) a.y_out = cf->y_out;
        a.wg_order = cf->wg_order;
    }
    if (sx && !sx->verdict) {         // judged by blocks (stream_block_verdict): silencing only
        REVS_REQUIRE(!sel, "revs_agent_step: bad streaming argument");      // (ctl == NULL: never silenced)
        a.ctl = sx->ctl; a.seq = sx->seq; a.base_seq = sx->base_seq; a.flags = sx->flags;
        const int lanes_ = pdhg_host ? pdhg_host->lanes : 0;
        REVS_REQUIRE(sx->kin >= 1 && sx->kin <= revs_agent_max_inner(T, lanes_) && sx->slice_stride >= 0 && sx->diff_stride >= 0,
                     "revs_agent_step: kin=%d outside 1..%d (T = %d)", sx->kin, revs_agent_max_inner(T, lanes_), T);
        REVS_REQUIRE(sx->kin == 1 || (!p_est_new && p_next && sx->pe_out && !s_out && !c_out &&
                                      sx->slice_stride >= (int64_t)0),
                     "revs_agent_step: several iterations per launch need the recomputed estimate, node sums "
                     "and pe_out, and write no S / C");
        a.kin = sx->kin; a.pe_out = sx->pe_out; a.slice_stride = sx->slice_stride; a.diff_stride = sx->diff_stride;
        a.dmax_out = sx->dmax_out;
        if (sx->y_out) a.y_out = sx->y_out;
        a.wg_order = sx->wg_order;          // (as many entries as this launch has workgroups: plan_wg_order)
    } else if (sx) {
        REVS_REQUIRE(sx->ctl && sx->rec && sx->p_in && sx->tree.n > 0 && sx->tree.n <= REVS_TREE_SWEEP_MAX && sx->tree.n % 8 == 0 &&
                     sx->tree.pack && sx->tree.w &&
                     sx->m > 0 && sx->vlo <= sx->vhi && sx->vtol >= 0.0 && !sel,
                     "revs_agent_step: bad streaming argument");
        a.ctl = sx->ctl; a.seq = sx->seq; a.base_seq = sx->base_seq; a.tree = sx->tree; a.p_in = sx->p_in; a.p_zero = sx->p_zero;
        a.vtol = sx->vtol; a.rec = sx->rec; a.flags = sx->flags; a.m = sx->m;
        a.nsel = T;
        a.sel.vlo = sx->vlo; a.sel.vhi = sx->vhi;
    }
    if (pdhg_host) a.pd = *pdhg_host; else revs_pdhg_defaults(&a.pd);
    REVS_REQUIRE(a.pd.max_iter > 0 && a.pd.check > 0 && a.pd.tau_scale >= 0 && a.pd.sigma_scale >= 0,
                 "revs_agent_step: bad PDHG parameters");
    const Shape sh = pick_shape(T, a.pd.lanes);
    const int64_t nblk = agent_num_blocks(n_homes, T, a.pd.lanes);
    REVS_REQUIRE(nblk < (1ll << 31), "revs_agent_step: too many homes for one launch");
    const dim3 grid((unsigned)(nblk + a.nsel));
    hipStream_t s = (hipStream_t)stream;
#define CALL(LPA, SPL) launch_agent<LPA, SPL>(a, mode, grid, s)
    REVS_FOR_SHAPE(sh, CALL);
#undef CALL
    REVS_CHECK_LAUNCH("revs_agent_step");
    return REVS_OK;
}

namespace revs {
int64_t agent_homes_per_block(int32_t T, int32_t lanes) {
    if (T <= 0 || T > REVS_MAX_T) return 0;
    return kBlock / pick_shape(T, lanes).lpa;
}

int agent_step_stream(int64_t n_homes, int32_t T, const float *cost, const revs_home_t *homes,
                      const float *load, const float *p_est_old, const float *p_est_new,
                      const float *p_sch, const float *gamma, float *p_sch_out, float *gamma_out,
                      float *diff, float *dsq, int32_t *status, float *pdhg_dual, float kappa,
                      int32_t mode, const revs_pdhg_t *pdhg_host, const int32_t *node_of,
                      double *p_next, float *p_est_next, const StreamExtra &sx, void *stream) {
    return agent_step_impl(n_homes, T, cost, homes, load, p_est_old, p_est_new, p_sch, gamma,
                           p_sch_out, gamma_out, nullptr, nullptr, diff, dsq, status, pdhg_dual,
                           kappa, mode, pdhg_host, nullptr, node_of, p_next, p_est_next, stream, &sx);
}

int agent_step_chain(int64_t n_homes, int32_t T, const float *cost, const revs_home_t *homes,
                     const float *load, const float *p_est, const float *p_sch, const float *gamma,
                     float *p_sch_out, float *gamma_out, float *s_out, float *c_out, float *diff, float *dsq,
                     int32_t *status, float *pdhg_dual, float kappa, int32_t mode, const revs_pdhg_t *pdhg_host,
                     const int32_t *node_of, const ChainFold &cf, unsigned int *flags, void *stream) {
    return agent_step_impl(n_homes, T, cost, homes, load, p_est, nullptr, p_sch, gamma, p_sch_out, gamma_out,
                           s_out, c_out, diff, dsq, status, pdhg_dual, kappa, mode, pdhg_host, nullptr,
                           node_of, nullptr, nullptr, stream, nullptr, flags, &cf);
}

// ---- verdicts by blocks (sharded streaming steady state, runtime.cpp) --------------------
// With residences sharded the node sums of an iteration are only known after an all-reduce, and
// one collective per 18 us sweep would be the whole step.  So a block of B sweeps runs without
// verdicts, each accumulating its sums into its own slice of a ring; ONE all-reduce then sums
// the B slices over the ranks, and one launch of B x T workgroups judges them all.  A failed
// verdict silences everything behind the block; the sweeps behind the failed iteration that
// did run worked from an estimate that is not the operator's answer: the state a block starts
// from is never overwritten while its verdicts are pending (the sweeps of a block rotate through
// the other sets of buffers, runtime.cpp), and the host goes back to it.
// more than 64 KB of dynamic LDS (the big tree shapes) has to be granted per kernel, once
template <int NT, int IPT, typename K>
static bool tree_big_lds(K kernel, size_t lds) {
    return grant_lds(reinterpret_cast<const void *>(kernel), lds, "tree form");
}

struct BlockVerdict {
    StreamCtl *ctl;
    unsigned int base_seq, gate_seq;      // no-op when a launch numbered base_seq..gate_seq failed
    unsigned int first_seq;               // slice g (counting `pre` as slice 0) is judged for iteration first_seq + g
    int32_t nb, T;                        // slices judged, `pre` included
    int32_t ndmax;                        // records written: nb, or nb + 1 (the call's last slice: its tail only)
    TreeArgs tree;
    const double *pre;                    // NULL, or the node sums of the call's first iteration (the caller's array)
    double *ring;                         // the ring slices follow: slice g - (pre != NULL) at ring + that * stride
    long long stride;                     // doubles from one ring slice to the next
    int32_t mt, ntail;                    // node sums, then ntail partial maxima of diff (all ranks') per ring slice
    double *hand_over;                    // NULL, or where the call's last slice (ring slice nb - (pre != NULL)) is copied to
    double vlo, vhi, vtol;
    unsigned long long *grp_bits;         // [nb] device words, zero on entry and on exit
    unsigned long long *grp_arrive;       // [nb + 1] device words (low halves: arrival counts), zero on entry and on exit
    double *grp_dmax;                     // [ndmax] device words
    double *rec;                          // the record ring (device address of pinned memory)
};
constexpr int kHandOverGroups = 32;       // workgroups that copy the call's last slice to the caller
constexpr int kGrpWords = REVS_STREAM_BLOCK_MAX + 4;   // per-slice words (+ the call's first iteration, + the handed-over slice)
// Workgroup (g, t): slot t of slice g -- it clears what it has read, so the slice is ready for the
// block that accumulates into it next; workgroup t == 0 of a ring slice also folds (and clears)
// the slice's tail -- the partial maxima of diff that the sweep which produced the slice left,
// every rank in its own REVS_DMAX_SLOTS words: the all-reduce's sum IS the gather -- into max_h
// diff[h] of that sweep's iteration.  With hand_over, kHandOverGroups more workgroups move the
// call's last slice (its rows are the next call's to judge) to the caller's array, clear it, and
// fold its tail.  The last workgroup to finish writes the records {rmax, failed, seq, max diff
// of the iteration before} and the lowest failed number into the control word.
// PAIR: a workgroup judges the slots 2 t and 2 t + 1 of its slice (tree_rmax_pair: half the scattered requests); the
// host picks it where T is even and the arrays are 16-byte aligned.
template <int NT, int IPT, bool PAIR = false>
__global__ __launch_bounds__(NT) void stream_block_verdict_kernel(const BlockVerdict b) {
    extern __shared__ double tree_lds[];
    const int tid = threadIdx.x;
    VD_STAMP(0);
    const int per_slice = PAIR ? b.T / 2 : b.T;                 // workgroups per slice
    const int njudge = b.nb * per_slice, npre = b.pre ? 1 : 0;
    const bool extra = (int)blockIdx.x >= njudge;
    // (the tree's static data are requested in front of the control word's test: one round trip, not two)
    unsigned long long pk0[IPT];
    double wgt0[IPT];
#pragma unroll
    for (int i = 0; i < IPT; ++i) pk0[i] = 0ull;
    if (!extra) {
        if (IPT * tid < b.tree.n) {
#pragma unroll
            for (int i = 0; i < IPT; i += 2) {
                const TreeU2 u = *reinterpret_cast<const TreeU2 *>(b.tree.pack + IPT * tid + i);
                pk0[i] = u.v[0]; pk0[i + 1] = u.v[1];
            }
        }
        tree_fetch_w<NT, IPT>(b.tree, wgt0);
    }
    {
        const unsigned int bad = b.ctl->bad_seq;
        if (bad >= b.base_seq && bad <= b.gate_seq) return;
    }
    VD_STAMP(1);
    const int g = extra ? b.nb : (int)blockIdx.x / per_slice, t = extra ? (int)blockIdx.x - njudge : (int)blockIdx.x - g * per_slice;
    double *slice = b.ring + (long long)(g - npre) * b.stride;       // (g == 0 with pre: not a ring slice)
    const bool ring_slice = !(b.pre && g == 0);
    __shared__ double dm_s[NT / 64];
    if (t == 0 && ring_slice && b.ntail > 0) {
        double v = 0.0;
        for (int i = tid; i < b.ntail; i += NT) { v = fmax(v, slice[b.mt + i]); slice[b.mt + i] = 0.0; }
        v = wave_max_d(v);
        if ((tid & 63) == 0) dm_s[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) {
            double dm = dm_s[0];
#pragma unroll
            for (int w = 1; w < NT / 64; ++w) dm = fmax(dm, dm_s[w]);
            __hip_atomic_store(&b.grp_dmax[g], dm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    double rmax = 0.0;
    if (!extra) {
        if constexpr (PAIR)
            // (the slice is cleared whole, with coalesced stores, by the last of its workgroups to arrive: 2048 scattered
            // 16-byte stores per workgroup and the wait for them were ~5.7 us of every workgroup's chain)
            rmax = tree_rmax_pair<NT, IPT>(b.tree, ring_slice ? slice : b.pre, b.T, 2 * t, b.vlo, b.vhi, tree_lds, nullptr, pk0, wgt0);
        else
            rmax = ring_slice ? tree_rmax<NT, IPT, true>(b.tree, slice, b.T, t, b.vlo, b.vhi, tree_lds, nullptr, slice, pk0, wgt0)
                              : tree_rmax<NT, IPT, true>(b.tree, b.pre, b.T, t, b.vlo, b.vhi, tree_lds, nullptr, nullptr, pk0, wgt0);
    } else {
        // (eight loads in flight per trip: one element per trip made this loop the launch -- 24 dependent round
        // trips, ~20 us of the 29 us a burst's verdict launch took)
        const long long per = (b.mt + kHandOverGroups - 1) / kHandOverGroups;
        const long long i0 = t * per, i1 = i0 + per < b.mt ? i0 + per : b.mt;
        for (long long ib = i0 + tid; ib < i1; ib += 8 * NT) {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = ib + q * NT < i1 ? slice[ib + q * NT] : 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (ib + q * NT < i1) { b.hand_over[ib + q * NT] = v[q]; slice[ib + q * NT] = 0.0; }
        }
    }
    __shared__ int last_s, slice_last_s;
    __shared__ unsigned int bad_s;
    VD_STAMP(2);
    if (tid == 0) {
        if (!extra)
            __hip_atomic_fetch_max(&b.grp_bits[g], (unsigned long long)__double_as_longlong(rmax),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // performed before this workgroup is counted
        // Arrivals in two levels: the T workgroups of a slice (the hand-over groups: a group of their own) count
        // on the slice's word, the last of them on the launch's -- several hundred adds on ONE word are a serial
        // chain of ~12 ns each at the end of a launch that a short burst has nothing to hide behind (776
        // workgroups for a block of 32: 9 us; now 24 + 33 in a row).
        unsigned int *const mine = reinterpret_cast<unsigned int *>(b.grp_arrive + g);
        const unsigned int members = extra ? (unsigned int)kHandOverGroups : (unsigned int)per_slice;
        const unsigned int old1 = __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        slice_last_s = old1 == members - 1u;
        if (slice_last_s) __hip_atomic_store(mine, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_s = 0;
        bad_s = 0xFFFFFFFFu;
    }
    __syncthreads();
    if (!slice_last_s) return;              // (uniform)
    if constexpr (PAIR) {
        if (!extra && ring_slice) {         // every workgroup of the slice has read it: clear it whole (mt is even here)
            TreeD2 *const z = reinterpret_cast<TreeD2 *>(slice);
            for (int i = tid; i < b.mt / 2; i += NT) z[i] = TreeD2{{0.0, 0.0}};
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    if (tid == 0) {
        const unsigned int old = __hip_atomic_fetch_add(&b.ctl->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_s = old == (unsigned int)(b.nb + (b.hand_over ? 1 : 0)) - 1u;
    }
    VD_STAMP(3);
    __syncthreads();
    if (!last_s) return;
    VD_STAMP(4);
    for (int q = tid; q < b.ndmax; q += NT) {
        const bool judged = q < b.nb;
        double r = 0.0;
        if (judged) {
            const unsigned long long bits = __hip_atomic_load(&b.grp_bits[q], __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&b.grp_bits[q], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            r = __longlong_as_double((long long)bits);
        }
        const unsigned int seq = b.first_seq + (unsigned int)q;
        const bool failed = judged && !(r <= b.vtol);
        if (failed) atomicMin(&bad_s, seq);
        volatile double *rec = b.rec + 4 * (seq % kRecRing);
        rec[0] = r;
        rec[1] = failed ? 1.0 : 0.0;
        rec[3] = (b.ntail > 0 && !(b.pre && q == 0))
                     ? __hip_atomic_load(&b.grp_dmax[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1.0;
        __threadfence_system();
        rec[2] = (double)seq;
    }
    __syncthreads();
    if (tid == 0) {
        if (bad_s != 0xFFFFFFFFu)
            __hip_atomic_store(&b.ctl->bad_seq, bad_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&b.ctl->arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    VD_STAMP(5);
}
#ifdef REVS_VD_STAMPS
}  // namespace revs
extern "C" int revs_tuning_verdict_stamps(double *out_host) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(revs::g_vd_stamps), sizeof(double) * 1024 * 8) == hipSuccess ? 0 : -1;
}
namespace revs {
#endif

int stream_block_verdict(StreamCtl *ctl, unsigned int base_seq, unsigned int gate_seq,
                         unsigned int first_seq, int32_t nb, int32_t T, const TreeArgs &tree,
                         const double *pre, double *ring, int64_t stride, int32_t mt, int32_t ntail,
                         double *hand_over, double vlo, double vhi, double vtol,
                         unsigned long long *grp_bits, double *grp_dmax, double *rec, void *stream) {
    REVS_REQUIRE(ctl && nb >= 0 && nb < kRecRing && (nb > 0 || hand_over) && T > 0 && tree.n > 0 &&
                 tree.n <= REVS_TREE_MAX && tree.n % tree_shape(tree.n).ipt == 0 && tree.pack && tree.w && ring && stride >= mt + ntail &&
                 mt > 0 && ntail >= 0 && vlo <= vhi && vtol >= 0.0 && grp_bits && grp_dmax && rec &&
                 (!pre || nb >= 1), "stream_block_verdict: bad argument");
    // (the arrival words follow the maxima: revs_plan_set_stream_block allocates both halves)
    const BlockVerdict b{ctl, base_seq, gate_seq, first_seq, nb, T, nb + (hand_over ? 1 : 0), tree, pre, ring,
                         (long long)stride, mt, ntail, hand_over, vlo, vhi, vtol, grp_bits, grp_bits + kGrpWords, grp_dmax, rec};
    const size_t lds = tree_lds_bytes(tree.n);
    const TreeShape sh = tree_shape(tree.n);
    // two slots per workgroup where a 16-byte request can fetch them (the 256 x 8 shape: registers)
    const bool pair = sh.nt == 256 && T % 2 == 0 && stride % 2 == 0 && ((uintptr_t)ring & 15u) == 0 && (!pre || ((uintptr_t)pre & 15u) == 0);
    const dim3 grid((unsigned)(nb * (pair ? T / 2 : T) + (hand_over ? kHandOverGroups : 0)));
    if (pair) {
        hipLaunchKernelGGL((stream_block_verdict_kernel<256, 8, true>), grid, dim3(256), lds, (hipStream_t)stream, b);
        REVS_CHECK_LAUNCH("stream_block_verdict");
        return REVS_OK;
    }
#define VK(NT, IPT)                                                                                            \
    do {                                                                                                       \
        if (!tree_big_lds<NT, IPT>(stream_block_verdict_kernel<NT, IPT>, lds)) return REVS_ELAUNCH;            \
        hipLaunchKernelGGL((stream_block_verdict_kernel<NT, IPT>), grid, dim3(NT), lds, (hipStream_t)stream, b); \
    } while (0)
    if (sh.nt == 256) VK(256, 8);
    else if (sh.nt == 512) VK(512, 8);
    else if (sh.ipt == 8) VK(1024, 8);
    else VK(1024, 16);
#undef VK
    REVS_CHECK_LAUNCH("stream_block_verdict");
    return REVS_OK;
}

template <int NT, int IPT>
__global__ __launch_bounds__(NT) void tree_voltage_kernel(TreeArgs tr, const double *p, int T,
                                                          double vlo, double vhi, double *v_out,
                                                          double *rmax_out) {
    extern __shared__ double tree_lds[];
    const double r = tree_rmax<NT, IPT>(tr, p, T, (int)blockIdx.x, vlo, vhi, tree_lds, v_out);
    if (threadIdx.x == 0 && rmax_out) rmax_out[blockIdx.x] = r;
}
}  // namespace revs

extern "C" int revs_tree_voltage(int32_t m, int32_t T, const revs_tree_t *tree, const double *p,
                                 double vlo, double vhi, double *v_out, double *rmax_out,
                                 void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && tree && p && tree->n > 0 && tree->n <= REVS_TREE_MAX &&
                 tree->n % tree_shape(tree->n).ipt == 0 && tree->pack && tree->w && vlo <= vhi,
                 "revs_tree_voltage: bad argument (tree nodes <= %d, a multiple of 8; of 16 beyond 8192)", REVS_TREE_MAX);
    const TreeArgs tr{tree->n, (const unsigned long long *)tree->pack, tree->w};
    const size_t lds = tree_lds_bytes(tree->n);
    const TreeShape sh = tree_shape(tree->n);
#define VK(NT, IPT)                                                                                    \
    do {                                                                                               \
        if (!tree_big_lds<NT, IPT>(tree_voltage_kernel<NT, IPT>, lds)) return REVS_ELAUNCH;            \
        hipLaunchKernelGGL((tree_voltage_kernel<NT, IPT>), dim3(T), dim3(NT), lds, (hipStream_t)stream, \
                           tr, p, T, vlo, vhi, v_out, rmax_out);                                       \
    } while (0)
    if (sh.nt == 256) VK(256, 8);
    else if (sh.nt == 512) VK(512, 8);
    else if (sh.ipt == 8) VK(1024, 8);
    else VK(1024, 16);
#undef VK
    REVS_CHECK_LAUNCH("revs_tree_voltage");
    return REVS_OK;
}

extern "C" int revs_agent_step_out(int64_t n_homes, int32_t T, const float *cost,
                                   const revs_home_t *homes, const float *load,
                                   const float *p_est_old, const float *p_est_new,
                                   const float *p_sch, const float *gamma, float *p_sch_out,
                                   float *gamma_out, float *s_out, float *c_out, float *diff,
                                   float *dsq, int32_t *status, float *pdhg_dual,
                                   float kappa, int32_t mode, const revs_pdhg_t *pdhg_host,
                                   void *stream) {
    return agent_step_impl(n_homes, T, cost, homes, load, p_est_old, p_est_new, p_sch, gamma,
                           p_sch_out, gamma_out, s_out, c_out, diff, dsq, status, pdhg_dual,
                           kappa, mode, pdhg_host, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int32_t revs_agent_max_inner(int32_t T, int32_t lanes) {
    if (T <= 0 || T > REVS_MAX_T) return 0;
    const Shape sh = pick_shape(T, lanes);
    return shape_max_inner(sh.lpa * sh.spl);
}

extern "C" int revs_agent_step_multi(int64_t n_homes, int32_t T, const float *cost, const revs_home_t *homes,
                                     const float *load, const float *p_est, const float *p_sch,
                                     const float *gamma, float *p_est_out, float *p_sch_out,
                                     float *gamma_out, float *p_est_next, float *diff, int64_t diff_stride,
                                     float *dsq, int32_t *status, float *pdhg_dual, float *pdhg_dual_out,
                                     float kappa, int32_t mode, const revs_pdhg_t *pdhg_host,
                                     const int32_t *node_of, double *p_next, int64_t slice_stride,
                                     double *dmax_out, int32_t kin, void *stream) {
    const int lanes_ = pdhg_host ? pdhg_host->lanes : 0;
    REVS_REQUIRE(p_est_out && node_of && p_next && kin >= 1 && kin <= revs_agent_max_inner(T, lanes_),
                 "revs_agent_step_multi: bad argument (kin = %d, at most %d at T = %d)", kin, revs_agent_max_inner(T, lanes_), T);
    REVS_REQUIRE(p_est_out != p_est && p_sch_out != p_sch && gamma_out != gamma,
                 "revs_agent_step_multi: the state is not updated in place");
    StreamExtra sx{};
    sx.verdict = false;
    sx.kin = kin;
    sx.pe_out = p_est_out;
    sx.y_out = pdhg_dual_out;
    sx.slice_stride = slice_stride;
    sx.diff_stride = diff_stride;
    sx.dmax_out = dmax_out;
    return agent_step_impl(n_homes, T, cost, homes, load, p_est, nullptr, p_sch, gamma, p_sch_out, gamma_out,
                           nullptr, nullptr, diff, dsq, status, pdhg_dual, kappa, mode, pdhg_host, nullptr,
                           node_of, p_next, p_est_next, stream, &sx);
}

extern "C" int revs_agent_step_select(int64_t n_homes, int32_t T, const float *cost,
                                      const revs_home_t *homes, const float *load,
                                      const float *p_est_old, const float *p_est_new,
                                      const float *p_sch, const float *gamma, float *p_sch_out,
                                      float *gamma_out, float *s_out, float *c_out, float *diff,
                                      float *dsq, int32_t *status, float *pdhg_dual,
                                      float kappa, int32_t mode, const revs_pdhg_t *pdhg_host,
                                      int32_t m, const double *sel_partial, const double *y,
                                      double vlo, double vhi, int32_t kadd, const double *vfull,
                                      const double *viol, int64_t *cand_idx, int32_t *cand_cnt,
                                      double *cand_val, double *stats, double seq,
                                      const int32_t *node_of, double *p_next, float *p_est_next,
                                      int32_t sel_nblk, void *stream) {
    REVS_REQUIRE(m > 0 && m <= 16384 && sel_partial && y && vfull && viol && cand_idx && cand_cnt &&
                 cand_val && stats && vlo <= vhi && kadd >= 0,
                 "revs_agent_step_select: bad selection argument");
    REVS_REQUIRE(sel_nblk >= 0 && sel_nblk <= 256, "revs_agent_step_select: sel_nblk=%d", sel_nblk);
    const SelectArgs sa{m, T, sel_nblk ? sel_nblk : revs_op_dual_blocks(m), kadd, sel_partial, y, vfull, viol, vlo, vhi,
                        seq, cand_idx, cand_cnt, cand_val, stats};
    return agent_step_impl(n_homes, T, cost, homes, load, p_est_old, p_est_new, p_sch, gamma,
                           p_sch_out, gamma_out, s_out, c_out, diff, dsq, status, pdhg_dual,
                           kappa, mode, pdhg_host, &sa, node_of, p_next, p_est_next, stream);
}

extern "C" int revs_agent_step(int64_t n_homes, int32_t T, const float *cost,
                               const revs_home_t *homes, const float *load,
                               const float *p_est_old, const float *p_est_new, float *p_sch,
                               float *gamma, float *s_out, float *c_out, float *diff,
                               float *dsq, int32_t *status, float *pdhg_dual, float kappa,
                               int32_t mode, const revs_pdhg_t *pdhg_host, void *stream) {
    return revs_agent_step_out(n_homes, T, cost, homes, load, p_est_old, p_est_new, p_sch, gamma,
                               p_sch, gamma, s_out, c_out, diff, dsq, status, pdhg_dual,
                               kappa, mode, pdhg_host, stream);
}

extern "C" int revs_residual_finalize(const float *diff, const float *dsq, int64_t n_homes,
                                      int32_t T, float kappa, float eps, double *scratch,
                                      float *out, void *stream) {
    REVS_REQUIRE(diff && dsq && scratch && out && n_homes > 0 && T > 0,
                 "revs_residual_finalize: bad argument");
    const int nchunk = revs_residual_num_chunks(n_homes);
    hipLaunchKernelGGL(residual_chunks_kernel, dim3(nchunk), dim3(256), 0, (hipStream_t)stream, diff,
                       dsq, n_homes, T, scratch);
    hipLaunchKernelGGL(residual_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scratch,
                       nchunk, kappa, eps, out);
    REVS_CHECK_LAUNCH("revs_residual_finalize");
    return REVS_OK;
}

// OR of the low three bits of every residence's status word into *out (a word the host can read: pinned memory).
__global__ __launch_bounds__(256) void status_or_kernel(int64_t n, const int32_t *__restrict__ status,
                                                        unsigned int *__restrict__ out) {
    unsigned int v = 0u;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) v |= (unsigned int)status[i] & 7u;
    const unsigned long long any = __ballot(v != 0u);
    if (any == 0ull) return;                                   // (the usual case: nothing to report)
    if (v) __hip_atomic_fetch_or(out, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int revs_status_or(int64_t n_homes, const int32_t *status, uint32_t *out, void *stream) {
    REVS_REQUIRE(n_homes > 0 && status && out, "revs_status_or: bad argument");
    const int64_t nb = (n_homes + 255) / 256;
    hipLaunchKernelGGL(status_or_kernel, dim3((unsigned)(nb < 1024 ? nb : 1024)), dim3(256), 0, (hipStream_t)stream, n_homes,
                       status, out);
    REVS_CHECK_LAUNCH("revs_status_or");
    return REVS_OK;
}

extern "C" int revs_residence_solve(int64_t n_homes, int32_t T, const float *tariff,
                                    const revs_home_t *homes, const float *load, float *p_out,
                                    float *soc_out, float *g_out, void *stream) {
    REVS_REQUIRE(n_homes > 0 && T > 0 && T <= REVS_MAX_T, "revs_residence_solve: bad size");
    REVS_REQUIRE(tariff && homes && load && p_out && soc_out && g_out,
                 "revs_residence_solve: null pointer argument");
    const Shape sh = pick_shape(T);
    const int64_t per = kBlock / sh.lpa;
    const dim3 grid((unsigned)((n_homes + per - 1) / per));
    hipStream_t s = (hipStream_t)stream;
#define CALL(LPA, SPL)                                                                         \
    hipLaunchKernelGGL((residence_kernel<LPA, SPL>), grid, dim3(kBlock), 0, s, n_homes, T,     \
                       tariff, homes, load, p_out, soc_out, g_out)
    REVS_FOR_SHAPE(sh, CALL);
#undef CALL
    REVS_CHECK_LAUNCH("revs_residence_solve");
    return REVS_OK;
}
