// Error reporting and version string of librevs_admm.so.
#include "common.h"
#include "internal.h"
#include <dlfcn.h>
#include <stdarg.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <vector>
#include <cstring>

#include <map>
#include <mutex>
namespace revs {
bool grant_lds(const void *kernel, size_t bytes, const char *who) {
    if (bytes <= 64 * 1024) return true;
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> granted;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    size_t &g = granted[{kernel, dev}];
    if (g >= bytes) return true;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        set_error("%s: %zu bytes of LDS refused: %s", who, bytes, hipGetErrorString(e));
        return false;
    }
    g = bytes;
    return true;
}
}  // namespace revs

namespace revs {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace revs

extern "C" const char *revs_last_error(void) { return revs::g_err; }
extern "C" const char *revs_version(void) { return "revs_admm_amd 0.1 (gfx950)"; }

// Device-side address of pinned host memory (hipHostMalloc / torch pin_memory): lets a
// kernel write its few result words where the host reads them, without a copy kernel.
extern "C" int revs_host_device_ptr(void *host_ptr, void **dev_ptr) {
    REVS_REQUIRE(host_ptr && dev_ptr, "revs_host_device_ptr: null argument");
    const hipError_t e = hipHostGetDevicePointer(dev_ptr, host_ptr, 0);
    if (e != hipSuccess) {
        revs::set_error("revs_host_device_ptr: %s", hipGetErrorString(e));
        return REVS_EINVAL;
    }
    return REVS_OK;
}

// ---- steady-state ADMM iteration as one host call (see revs_admm.h) -------------------
struct revs_comm;
struct revs_plan {
    revs_plan_desc_t d;
    hipEvent_t ev;
    double seq;
    uint32_t *counters;     // device, one per 32-row tile: K-split workgroups of R p done
    double t_launch = 0.0, t_wait = 0.0;   // host time in launches / waiting (REVS_PLAN_TRACE)
    // streaming steady state (revs_plan_stream_run)
    revs::StreamCtl *ctl = nullptr;        // device
    double *rec_host = nullptr;            // pinned: double[kRecRing][4] = {rmax, failed, seq, max diff of the iteration before}
    double *rec_dev = nullptr;             // its device-side address
    unsigned int *flags_host = nullptr;    // pinned: OR of the residences' status bits
    unsigned int *flags_dev = nullptr;
    unsigned int stream_seq = 0;           // sequence number of the last streaming launch
    revs::TreeArgs tree{};                 // tree.n == 0: no tree form
    revs_comm *comm = nullptr;
    // verdicts by blocks (revs_plan_set_stream_block)
    int32_t block = 0;                     // iterations judged together; <= 1: every launch judges itself
    int32_t overlap = 0;                   // all-reduce + verdicts of a block on `side`, beside the next block's sweeps
    int32_t inner = 1;                     // ADMM iterations per sweep launch (revs_plan_set_stream_inner)
    int32_t fold_redo = 2;                 // Newton steps beyond the first inside the folded chain (revs_plan_set_fold_redo)
    int32_t kadd_cold = 0, kadd_cold_at = 0;
    int32_t *wg_order = nullptr;           // the sweep's workgroups, heaviest first (plan_wg_order), device; built on first use
    bool wg_order_tried = false;   // revs_plan_set_kadd_cold: rows admitted per Newton iteration while many are violated
    double *ring = nullptr;                // device: node sums (+ diff tails) of two blocks, double[2][block][stride]
    size_t ring_cap = 0;                   // ... doubles allocated
    bool ring_dirty = true;                // the ring is not known to be all zero (fresh, or a call failed)
    unsigned long long *grp_bits = nullptr;                // device: per-slice maxima, zero between launches
    double *grp_dmax = nullptr;            // device: per-slice max diff
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> events;        // pool: sweeps-done / verdicts-done per block, end of call
    // optional timing of the bursts on their own stream (revs_plan_stream_timing)
    hipEvent_t tev[2] = {nullptr, nullptr};
    hipEvent_t cev[4] = {nullptr, nullptr, nullptr, nullptr};   // around a block's all-reduce / around that block's sweeps
    bool cev_valid = false;
    int32_t cev_nb = 0;                    // iterations of the timed block
    // folded chain (revs_plan_chain_fold_run): sums of the trial's evaluation E2 / of the next
    // iteration's evaluation E1 by iteration parity, the E2 side's row scratch, the odd parity's
    // candidate sets and stats blocks ([0]: the evaluation's, [1]: the trial's)
    double *fold_e2[2] = {nullptr, nullptr}, *fold_e1[2] = {nullptr, nullptr};
    double *fold_v[3] = {nullptr, nullptr, nullptr};
    int32_t *fold_info[2] = {nullptr, nullptr};        // the models' pivot counts, by iteration parity
    double *fold_sh[2] = {nullptr, nullptr};           // the trial's shifts R^T y / kappa, list order / row order
    int64_t *fold_ci[2] = {nullptr, nullptr};
    int32_t *fold_cc[2] = {nullptr, nullptr};
    double *fold_cv[2] = {nullptr, nullptr};
    double *fold_st_host[2] = {nullptr, nullptr}, *fold_st_dev[2] = {nullptr, nullptr};
    revs_newton_opts_t newton{};           // revs_plan_set_newton
    double *fold_st_local[2] = {nullptr, nullptr};     // device: stats of the next-iteration half, by the parity of the set they belong to
    bool fold_st_local_valid = false;      // ... hold the stats of the evaluation the next verdict belongs to
    int32_t fold_par = 0;                  // parity of the iteration a resumed call starts with
    bool fold_ready = false;               // ... whose rows / model / step the last call has already run
    int32_t timing = 0;                    // 0 off, 1 armed (next burst records tev[0]), 2 open
    int64_t timed_launches = 0;            // residence-sweep launches between the two events
};

// Host-side acceptance test of a chained Newton iteration (operator_newton.py: _chain_launch): the
// checks AdmmEngine._operator_solve_newton would make on the two evaluations' stats, for the
// one outcome that needs no further launch.  See include/revs_admm.h.
// why: 0 accepted | 1 everything holds but the rows after the step are still above the tolerance (the
// step itself is a good Newton step: another iteration from it) | 2 anything else
static int chain_accept_impl(int32_t T, const double *s0, const double *s1, double scale, double eps,
                             int32_t amax, int32_t kadd, int32_t chain_few, int32_t *nsup_sum,
                             int32_t *nsup_max, int *why) {
    *why = 2;
    if (!s0 || !s1 || T <= 0 || !(scale > 0.0) || !nsup_sum || !nsup_max) return 0;
    double rmax0 = 0.0, ns_max = 0.0, ncand_max = 0.0;
    for (int t = 0; t < T; ++t) {
        const double *a = s0 + 8 * t;
        if (a[2] > amax) return 0;                       // more multipliers than a model holds
        const double r = a[0] / scale;
        rmax0 = r > rmax0 ? r : rmax0;
        if (a[2] >= amax && a[3] > 0 && r > eps) return 0;
        const double room = kadd < amax - a[2] ? kadd : amax - a[2];
        const double nc = a[2] + (a[3] < room ? a[3] : room);
        ncand_max = nc > ncand_max ? nc : ncand_max;
        ns_max = a[2] > ns_max ? a[2] : ns_max;
    }
    if (!(rmax0 > eps)) return 0;                        // already converged: the general path
    if (ncand_max > 8) return 0;                         // not the small model
    if ((ns_max + kadd <= REVS_DUAL_FEW) != (chain_few != 0)) return 0;
    double rmax1 = 0.0, sum = 0.0, mx = 0.0;
    for (int t = 0; t < T; ++t) {
        const double *a = s0 + 8 * t, *b = s1 + 8 * t;
        const double D = a[1];
        if (a[0] / scale > eps &&                        // pending slot: Armijo on the full step
            !(b[1] >= D + 1e-4 * b[4] - 1e-11 * (D < 0 ? -D : D)))
            return 0;
        if (b[2] > amax) return 0;
        const double r = b[0] / scale;
        rmax1 = r > rmax1 ? r : rmax1;
        sum += b[2];
        mx = b[2] > mx ? b[2] : mx;
    }
    *nsup_sum = (int32_t)sum;
    *nsup_max = (int32_t)mx;
    if (!(rmax1 <= eps)) { *why = 1; return 0; }         // needs another iteration
    *why = 0;
    return 1;
}

extern "C" int revs_newton_chain_accept(int32_t T, const double *s0, const double *s1, double scale,
                                        double eps, int32_t amax, int32_t kadd, int32_t chain_few,
                                        int32_t *nsup_sum, int32_t *nsup_max) {
    int why = 0;
    return chain_accept_impl(T, s0, s1, scale, eps, amax, kadd, chain_few, nsup_sum, nsup_max, &why);
}

static const int32_t *plan_wg_order(revs_plan_t *plan);
extern "C" revs_plan_t *revs_plan_create(const revs_plan_desc_t *desc) {
    if (!desc || !desc->stats || !desc->stats_host || !desc->pnq || desc->T <= 0 || desc->m <= 0) {
        revs::set_error("revs_plan_create: bad descriptor");
        return nullptr;
    }
    revs_plan *p = new revs_plan{*desc, nullptr, 0.0, nullptr};
    const size_t nb = sizeof(uint32_t) * ((desc->m + 31) / 32);
    hipError_t e = hipEventCreateWithFlags(&p->ev, hipEventDisableTiming);
    const char *what = "hipEventCreateWithFlags";
    if (e == hipSuccess) { e = hipMalloc((void **)&p->counters, nb); what = "hipMalloc"; }
    if (e == hipSuccess) { e = hipMemset(p->counters, 0, nb); what = "hipMemset"; }
    if (e != hipSuccess) {
        revs::set_error("revs_plan_create: %s: %s", what, hipGetErrorString(e));
        if (p->ev) (void)hipEventDestroy(p->ev);
        if (p->counters) (void)hipFree(p->counters);
        delete p;
        return nullptr;
    }
    // control block, record ring and status word of the streaming steady state
    const revs::StreamCtl ctl0{0u, 0u, 0ull};
    void *host = nullptr;
    what = "hipMalloc";
    e = hipMalloc((void **)&p->ctl, sizeof(revs::StreamCtl));
    if (e == hipSuccess) { e = hipMemcpy(p->ctl, &ctl0, sizeof(ctl0), hipMemcpyHostToDevice); what = "hipMemcpy"; }
    if (e == hipSuccess) {
        e = hipHostMalloc(&host, sizeof(double) * 4 * revs::kRecRing + 64, hipHostMallocMapped);
        what = "hipHostMalloc";
    }
    if (e == hipSuccess) {
        p->rec_host = (double *)host;
        p->flags_host = (unsigned int *)(p->rec_host + 4 * revs::kRecRing);
        for (int i = 0; i < 4 * revs::kRecRing; ++i) p->rec_host[i] = -1.0;
        *p->flags_host = 0u;
        void *dp = nullptr;
        e = hipHostGetDevicePointer(&dp, host, 0);
        what = "hipHostGetDevicePointer";
        p->rec_dev = (double *)dp;
        p->flags_dev = (unsigned int *)(p->rec_dev + 4 * revs::kRecRing);
    }
    if (e != hipSuccess) {
        revs::set_error("revs_plan_create: %s: %s", what, hipGetErrorString(e));
        revs_plan_destroy(p);
        return nullptr;
    }
    (void)plan_wg_order(p);       // (here, not inside the first burst: a copy of the residence records to the host and a sort)
    return p;
}

// The order in which a multi-iteration sweep's launch takes its workgroups of residences: those that hold the most
// residences with an EV first (a residence without one has no QP to solve: its wavefront runs a third of the
// instructions).  A launch of 100 000 residences is 3 125 workgroups on a chip that holds 1 280 at a time: its last round
// cannot fill the chip, and with the heavy workgroups in front that round is made of the quick ones.  Nothing else
// changes: the same workgroups do the same work (node sums are exact: order-independent).  NULL when the descriptor
// carries no residence records (then: launch order = residence order).
static const int32_t *plan_wg_order(revs_plan_t *plan) {
    if (plan->wg_order_tried) return plan->wg_order;
    plan->wg_order_tried = true;
#ifdef REVS_TUNING        // (tuning builds: A/B of the order inside one job)
    if (getenv("REVS_NO_WG_ORDER")) return nullptr;
#endif
    const revs_plan_desc_t &d = plan->d;
    const int64_t per = revs::agent_homes_per_block(d.T, d.pdhg.lanes);
    if (!d.homes || d.n_homes <= 0 || per <= 0) return nullptr;
    const int64_t nb = (d.n_homes + per - 1) / per;
    if (nb < 2 || nb >= (1ll << 31)) return nullptr;
    std::vector<revs_home_t> h((size_t)d.n_homes);
    if (hipMemcpy(h.data(), d.homes, sizeof(revs_home_t) * (size_t)d.n_homes, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
    std::vector<int32_t> w((size_t)nb, 0), order((size_t)nb);
    for (int64_t i = 0; i < d.n_homes; ++i) w[(size_t)(i / per)] += h[(size_t)i].ev != 0;
    for (int64_t b = 0; b < nb; ++b) order[(size_t)b] = (int32_t)b;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return w[(size_t)x] > w[(size_t)y]; });
    if (hipMalloc((void **)&plan->wg_order, sizeof(int32_t) * (size_t)nb) != hipSuccess) { plan->wg_order = nullptr; return nullptr; }
    if (hipMemcpy(plan->wg_order, order.data(), sizeof(int32_t) * (size_t)nb, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(plan->wg_order);
        plan->wg_order = nullptr;
    }
    return plan->wg_order;
}

extern "C" void revs_plan_destroy(revs_plan_t *plan) {
    if (!plan) return;
    if (plan->wg_order) (void)hipFree(plan->wg_order);
    if (plan->ev) (void)hipEventDestroy(plan->ev);
    if (plan->counters) (void)hipFree(plan->counters);
    if (plan->ctl) (void)hipFree(plan->ctl);
    if (plan->rec_host) (void)hipHostFree(plan->rec_host);
    if (plan->ring) (void)hipFree(plan->ring);
    if (plan->grp_bits) (void)hipFree(plan->grp_bits);
    if (plan->grp_dmax) (void)hipFree(plan->grp_dmax);
    for (int i = 0; i < 2; ++i) {
        if (plan->fold_e2[i]) (void)hipFree(plan->fold_e2[i]);      // (fold_e1[i ^ 1] is its second half)
        if (plan->fold_ci[i]) (void)hipFree(plan->fold_ci[i]);
        if (plan->fold_cc[i]) (void)hipFree(plan->fold_cc[i]);
        if (plan->fold_cv[i]) (void)hipFree(plan->fold_cv[i]);
        if (plan->fold_st_host[i]) (void)hipHostFree(plan->fold_st_host[i]);
    }
    for (double *v : plan->fold_v) if (v) (void)hipFree(v);
    for (double *v : plan->fold_st_local) if (v) (void)hipFree(v);
    for (int32_t *v : plan->fold_info) if (v) (void)hipFree(v);
    for (double *v : plan->fold_sh) if (v) (void)hipFree(v);
    for (hipEvent_t e : plan->events) (void)hipEventDestroy(e);
    for (hipEvent_t e : plan->tev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : plan->cev) if (e) (void)hipEventDestroy(e);
    if (plan->side) (void)hipStreamDestroy(plan->side);
    delete plan;
}

// v = R p_in with the row bookkeeping (one launch when the tile form applies); clears p_out.
static int plan_product(revs_plan_t *plan, const double *y, const double *pin, double *pout,
                        void *stream) {
    const revs_plan_desc_t &d = plan->d;
    if (d.T <= 32 && (d.m + 31) / 32 <= 256)
        return revs_op_dual_product_rows(d.m, d.T, d.Rt, pin, d.pnq, y, d.vlo, d.vhi, d.ksplit,
                                         d.v_slabs, d.vfull, d.viol, d.partial, pout, plan->counters,
                                         stream);
    const int r = revs_gemm_tn_f64_split(d.m, d.T, d.m, d.Rt, pin, d.v_slabs, d.ksplit, stream);
    if (r != REVS_OK) return r;
    return revs_op_dual_rows(d.m, d.T, d.ksplit, d.v_slabs, d.pnq, y, d.vlo, d.vhi, d.vfull, d.viol,
                             d.partial, pout, stream);
}

extern "C" int revs_plan_spec_step(revs_plan_t *plan, int32_t phase, const double *y,
                                   int32_t use_y, const float *p_est, float *p_est_new,
                                   const float *p_sch, const float *gamma, float *p_sch_out,
                                   float *gamma_out, float *s_out, float *c_out, int32_t fused_in,
                                   const double *p_in, double *p_out, float *p_est_next,
                                   double *rmax_out, void *ev_mid, void *ev_end, void *stream) {
    if (phase == 64) {                       // a product run ahead, nothing else
        REVS_REQUIRE(plan && y && p_in && p_out && p_in != p_out, "revs_plan_spec_step: bad argument");
        return plan_product(plan, y, p_in, p_out, stream);
    }
    REVS_REQUIRE(plan && phase >= 1 && phase <= 63 && (!(phase & 28) || (phase & 2)) &&
                 (!(phase & 32) || phase == 32) && y && p_est && p_est_new && p_sch && gamma &&
                 p_sch_out && gamma_out && rmax_out && p_in, "revs_plan_spec_step: bad argument");
    REVS_REQUIRE(!(phase & 8) || p_out, "revs_plan_spec_step: running ahead needs p_out");
    const revs_plan_desc_t &d = plan->d;
    const bool fuse_out = p_out != nullptr;
    REVS_REQUIRE(!(fuse_out || fused_in) || (!use_y && d.node_of && (!fuse_out || p_est_next)),
                 "revs_plan_spec_step: fused home pass needs y = 0, node_of and p_est_next");
    REVS_REQUIRE(p_out != p_in && (fused_in || p_in == d.pnq),
                 "revs_plan_spec_step: p_in / p_out inconsistent");
    hipStream_t s = (hipStream_t)stream;
    const auto t_enter = std::chrono::steady_clock::now();
    int rc;
    if ((phase & 1) && !fused_in) {  // home pass of this evaluation (else: the last sweep did it)
        rc = revs_op_dual_evaluate(1, d.m, d.T, d.node_ptr, p_est, p_sch, gamma, d.R, d.Rt, y, use_y,
                                   d.kappa, d.vlo, d.vhi, d.kadd, d.ksplit, d.d_slabs, d.v_slabs,
                                   d.pnq, p_est_new, d.vfull, d.viol, d.partial, d.cand_idx,
                                   d.cand_cnt, d.cand_val, d.stats, 0.0, nullptr, stream);
        if (rc != REVS_OK) return rc;
    }
    if (!(phase & (2 | 32))) return REVS_OK;
    const double seq = (phase & 32) ? plan->seq : (plan->seq += 1.0);
    if (!(phase & 32)) {
    // node sums p_in: this evaluation's (from the home pass above, or from the last fused
    // sweep; all-reduced by a sharded caller between the phases); p_out: where this sweep
    // accumulates the next ones -- never the same array, so that clearing the latter cannot
    // race with the product reading the former
    const bool one_launch = d.T <= 32 && (d.m + 31) / 32 <= 256;
    const int sel_nblk = one_launch ? (d.m + 31) / 32 : 0;
    auto product = [&](const double *pin, double *pout) -> int {
        return plan_product(plan, y, pin, pout, stream);
    };
    if (!(phase & 4)) {                              // (else: the previous call ran it ahead)
        rc = product(p_in, p_out);
        if (rc != REVS_OK) return rc;
    }
    if (ev_mid) (void)hipEventRecord((hipEvent_t)ev_mid, s);
    // the candidate selection rides in the sweep's launch (its first T workgroups)
    rc = revs_agent_step_select(d.n_homes, d.T, d.cost, d.homes, d.load, p_est,
                                (d.recompute_pe_new && !use_y) ? nullptr : p_est_new, p_sch,
                                gamma, p_sch_out, gamma_out, s_out, c_out, d.diff, d.dsq,
                                d.status, d.pdhg_dual, (float)d.kappa, d.mode, &d.pdhg, d.m,
                                d.partial, y, d.vlo, d.vhi, d.kadd, d.vfull, d.viol, d.cand_idx,
                                d.cand_cnt, d.cand_val, d.stats, seq, fuse_out ? d.node_of : nullptr,
                                p_out, fuse_out ? p_est_next : nullptr, sel_nblk, stream);
    if (rc != REVS_OK) return rc;
    if (ev_end) (void)hipEventRecord((hipEvent_t)ev_end, s);
    if (phase & 8) {
        // The NEXT iteration's product, before this one's verdict is known: it needs only the
        // node sums this sweep leaves in p_out, and it keeps the queue from running dry while
        // the host turns around (a restart costs the stream ~6 us).  It clears the array that
        // held this evaluation's sums.  If this sweep is discarded it has computed nothing
        // anyone reads: the caller's next evaluation rewrites every array it touches.
        rc = product(p_out, const_cast<double *>(p_in));
        if (rc != REVS_OK) return rc;
    }
    if (phase & 16) return REVS_OK;          // the caller waits with a phase-32 call
    }
    // Wait for the evaluation, not the sweep: poll the sequence tag the select kernel writes
    // into the pinned stats block of every slot (lower latency than an event wait).
    const volatile double *st = d.stats_host;
    const auto t0 = std::chrono::steady_clock::now();
    plan->t_launch += std::chrono::duration<double, std::micro>(t0 - t_enter).count();
    for (int t = 0; t < d.T; ++t) {
        unsigned spins = 0;
        while (st[8 * t + 5] != seq) {
            if ((++spins & 0xFFFF) == 0) {
                if (hipStreamQuery(s) == hipSuccess && st[8 * t + 5] != seq) {
                    revs::set_error("revs_plan_spec_step: stream idle but stats tag missing");
                    return REVS_ELAUNCH;
                }
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
                    (void)hipStreamSynchronize(s);   // nothing of ours may still be writing
                    revs::set_error("revs_plan_spec_step: timed out waiting for the evaluation");
                    return REVS_ELAUNCH;
                }
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    plan->t_wait += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    double mx = 0.0;
    for (int t = 0; t < d.T; ++t) mx = st[8 * t] > mx ? st[8 * t] : mx;
    *rmax_out = mx;
    return REVS_OK;
}

extern "C" int revs_plan_chain_step(revs_plan_t *plan, const double *y, double *y_trial,
                                    int32_t use_y, int32_t sup0, int32_t chain_few,
                                    const float *p_est, float *p_est_new, const float *p_sch,
                                    const float *gamma, float *p_sch_out, float *gamma_out,
                                    float *s_out, float *c_out, int32_t *accepted,
                                    int32_t *nsup_sum, int32_t *nsup_max, void *ev_mid,
                                    void *ev_end, void *stream) {
    REVS_REQUIRE(plan && y && y_trial && y != y_trial && p_est && p_est_new && p_sch && gamma &&
                 p_sch_out && gamma_out && accepted && nsup_sum && nsup_max && sup0 >= -1 && sup0 <= 1,
                 "revs_plan_chain_step: bad argument");
    const revs_plan_desc_t &d = plan->d;
    REVS_REQUIRE(d.cand_idx1 && d.cand_cnt1 && d.cand_val1 && d.stats1 && d.stats1_host && d.yhat &&
                 d.k_full && d.info && d.max_pivots > 0 && d.eps > 0,
                 "revs_plan_chain_step: the plan was created without the chain's buffers");
    hipStream_t s = (hipStream_t)stream;
    int64_t *const ci[2] = {d.cand_idx, d.cand_idx1};
    int32_t *const cc[2] = {d.cand_cnt, d.cand_cnt1};
    double *const cv[2] = {d.cand_val, d.cand_val1};
    double *const st[2] = {d.stats, d.stats1};
    const double scale = std::max(std::max(std::fabs(d.vlo), std::fabs(d.vhi)), 1e-300);
    const int nb32 = (d.m + 31) / 32;
    int sel_nblk = (d.T <= 32 && nb32 <= 256) ? nb32 : 0;
    // home pass of an evaluation of multipliers yy: row-wise from the lists of set `sup`, or dense
    auto home_pass = [&](const double *yy, int uy, int sup) -> int {
        if (uy && sup >= 0)
            return revs_op_dual_eval_rows(d.m, d.T, d.node_ptr, p_est, p_sch, gamma, d.R, ci[sup],
                                          cc[sup], yy, d.kappa, d.pnq, p_est_new, stream);
        return revs_op_dual_evaluate(1, d.m, d.T, d.node_ptr, p_est, p_sch, gamma, d.R, d.Rt, yy, uy,
                                     d.kappa, d.vlo, d.vhi, d.kadd, d.ksplit, d.d_slabs, d.v_slabs,
                                     d.pnq, p_est_new, d.vfull, d.viol, d.partial, ci[0], cc[0], cv[0],
                                     st[0], 0.0, nullptr, stream);
    };
    // product R p and the row bookkeeping; the selection is left to the next launch
    // (a feeder of more than REVS_TREE_SWEEP_MAX nodes: the fused launches below do not hold it, its rows still come
    // from the tree form -- one block of partial sums per slot)
    const bool big_tree = plan->tree.n > REVS_TREE_SWEEP_MAX;
    const revs_tree_t trb{plan->tree.n, (const uint64_t *)plan->tree.pack, plan->tree.w};
    if (big_tree) sel_nblk = 1;
    auto rows = [&](const double *yy, int uy, int k) -> int {
        if (big_tree)
            return revs_op_dual_rows_tree(d.m, d.T, &trb, d.pnq, yy, d.vlo, d.vhi, d.kadd, d.vfull, d.viol, d.partial,
                                          nullptr, ci[k], cc[k], cv[k], st[k], 0.0, 0, stream);
        return revs_op_dual_evaluate(2 | 4, d.m, d.T, d.node_ptr, p_est, p_sch, gamma, d.R, d.Rt, yy, uy,
                                     d.kappa, d.vlo, d.vhi, d.kadd, d.ksplit, d.d_slabs, d.v_slabs,
                                     d.pnq, p_est_new, d.vfull, d.viol, d.partial, ci[k], cc[k], cv[k],
                                     st[k], 0.0, plan->counters, stream);
    };
    // With the feeder as a tree the operator side between the home passes is the tree form of R p:
    // rows, selection, small model and step of every slot in ONE launch of T workgroups, the trial's
    // rows in another (its selection rides in the sweep's launch) -- no matrix stream at all.
    const bool tf = plan->tree.n > 0 && plan->tree.n <= REVS_TREE_SWEEP_MAX;
    const revs_tree_t trh{plan->tree.n, (const uint64_t *)plan->tree.pack, plan->tree.w};
    int rc;
    if ((rc = home_pass(y, use_y, sup0)) != REVS_OK) return rc;
    if (tf) {
        rc = revs_op_dual_tree_select_model_step(d.m, d.T, &trh, d.pnq, y, d.vlo, d.vhi, d.kadd, d.vfull, d.viol,
                                                 d.partial, ci[0], cc[0], cv[0], st[0], 0.0, d.R, d.kappa, d.delta,
                                                 d.max_pivots, d.k_full, d.yhat, d.info, scale, d.eps, y_trial,
                                                 st[1] + 4, stream);
    } else {
        if ((rc = rows(y, use_y, 0)) != REVS_OK) return rc;
        rc = revs_op_dual_select_model_step(d.m, d.T, d.partial, sel_nblk, y, d.vlo, d.vhi, d.kadd, d.vfull,
                                            d.viol, ci[0], cc[0], cv[0], st[0], 0.0, d.R,
                                            d.pnq + (int64_t)d.m * d.T, d.kappa, d.delta, d.max_pivots,
                                            d.k_full, d.yhat, d.info, scale, d.eps, y_trial, st[1] + 4,
                                            stream);
    }
    if (rc != REVS_OK) return rc;
    if ((rc = home_pass(y_trial, 1, chain_few ? 0 : -1)) != REVS_OK) return rc;
    if (tf)
        rc = revs_op_dual_rows_tree(d.m, d.T, &trh, d.pnq, y_trial, d.vlo, d.vhi, d.kadd, d.vfull, d.viol, d.partial,
                                    nullptr, ci[1], cc[1], cv[1], st[1], 0.0, 0, stream);
    else
        rc = rows(y_trial, 1, 1);
    if (rc != REVS_OK) return rc;
    if (ev_mid) (void)hipEventRecord((hipEvent_t)ev_mid, s);
    const double seq = -(plan->seq += 1.0);          // (negative: not a spec-step tag)
    rc = revs_agent_step_select(d.n_homes, d.T, d.cost, d.homes, d.load, p_est, p_est_new, p_sch,
                                gamma, p_sch_out, gamma_out, s_out, c_out, d.diff, d.dsq, d.status,
                                d.pdhg_dual, (float)d.kappa, d.mode, &d.pdhg, d.m, d.partial, y_trial,
                                d.vlo, d.vhi, d.kadd, d.vfull, d.viol, ci[1], cc[1], cv[1], st[1], seq,
                                nullptr, nullptr, nullptr, tf ? 1 : sel_nblk, stream);
    if (rc != REVS_OK) return rc;
    if (ev_end) (void)hipEventRecord((hipEvent_t)ev_end, s);
    const volatile double *tg = d.stats1_host;
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < d.T; ++t) {
        unsigned spins = 0;
        while (tg[8 * t + 5] != seq) {
            if ((++spins & 0xFFFF) == 0) {
                if (hipStreamQuery(s) == hipSuccess && tg[8 * t + 5] != seq) {
                    revs::set_error("revs_plan_chain_step: stream idle but stats tag missing");
                    return REVS_ELAUNCH;
                }
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
                    (void)hipStreamSynchronize(s);
                    revs::set_error("revs_plan_chain_step: timed out waiting for the evaluation");
                    return REVS_ELAUNCH;
                }
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    *accepted = revs_newton_chain_accept(d.T, d.stats_host, d.stats1_host, scale, d.eps,
                                         REVS_DUAL_AMAX, d.kadd, chain_few, nsup_sum, nsup_max);
    return REVS_OK;
}

extern "C" int revs_plan_spec_run(revs_plan_t *plan, int32_t max_steps, const double *y,
                                  revs_spec_state_t *st, double scale, double eps,
                                  int32_t *kept_steps, int32_t *last_fused_in, double *rmax_out,
                                  void *stream) {
    REVS_REQUIRE(plan && max_steps >= 0 && y && st && kept_steps && last_fused_in && rmax_out &&
                 scale > 0.0 && st->p_est && st->p_est_new && st->p_est_alt && st->p_sch &&
                 st->p_sch_alt && st->gamma && st->gamma_alt && st->p0 && st->p_alt &&
                 st->p0 != st->p_alt && st->p0 == plan->d.pnq &&
                 (!st->fused_ready || st->fused_p == st->p0 || st->fused_p == st->p_alt),
                 "revs_plan_spec_run: bad argument");
    *kept_steps = 0;
    *last_fused_in = 0;
    *rmax_out = 0.0;
    bool ahead = false;                    // this iteration's product is already in the queue
    static const bool trace = getenv("REVS_PLAN_TRACE") != nullptr;
    const auto tr0 = std::chrono::steady_clock::now();
    for (int32_t k = 0; k < max_steps; ++k) {
        const int32_t fused_in = st->fused_ready;
        const double *p_in = fused_in ? st->fused_p : st->p0;
        double *p_out = p_in == st->p0 ? st->p_alt : st->p0;
        double rm = 0.0;
        const int32_t phase = 3 | (ahead ? 4 : 0) | (k + 1 < max_steps ? 8 : 0);
        ahead = (phase & 8) != 0;
        const int rc = revs_plan_spec_step(plan, phase, y, 0, st->p_est, st->p_est_new, st->p_sch, st->gamma,
                                           st->p_sch_alt, st->gamma_alt, nullptr, nullptr, fused_in, p_in,
                                           p_out, st->p_est_alt, &rm, nullptr, nullptr, stream);
        if (rc != REVS_OK) return rc;
        *rmax_out = rm;
        if (!(rm / scale <= eps)) {          // discard: the caller finishes this iteration
            *last_fused_in = fused_in;
            return REVS_OK;
        }
        std::swap(st->p_sch, st->p_sch_alt);
        std::swap(st->gamma, st->gamma_alt);
        st->fused_ready = 1;
        st->fused_p = p_out;
        std::swap(st->p_est, st->p_est_new);
        std::swap(st->p_est_new, st->p_est_alt);
        ++*kept_steps;
    }
    if (trace && *kept_steps > 0) {
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr0).count();
        fprintf(stderr, "[revs_plan_spec_run] %d steps, %.2f us per step on the host; launches %.2f us, "
                "waiting %.2f us per step\n", *kept_steps, us / *kept_steps, plan->t_launch / *kept_steps,
                plan->t_wait / *kept_steps);
        plan->t_launch = plan->t_wait = 0.0;
    }
    return REVS_OK;
}

// ---- the operator's Newton solve as one native call (see revs_admm.h) ---------------------------
extern "C" int revs_plan_set_newton(revs_plan_t *plan, const revs_newton_opts_t *o) {
    REVS_REQUIRE(plan && o && o->k_slabs && o->nks >= 1 && o->nks <= 64 && o->alpha_host && o->alpha_dev && o->info_host &&
                 o->newton_max >= 1 && o->ls_max >= 1, "revs_plan_set_newton: bad argument");
    plan->newton = *o;
    return REVS_OK;
}

extern "C" int revs_plan_newton_solve(revs_plan_t *plan, revs_newton_state_t *st, void *stream) {
    REVS_REQUIRE(plan && st && st->y && st->y_trial && st->y != st->y_trial && st->p_est && st->p_sch && st->gamma &&
                 st->p_est_new && st->sup >= -1 && st->sup <= 1, "revs_plan_newton_solve: bad argument");
    const revs_plan_desc_t &d = plan->d;
    const revs_newton_opts_t &o = plan->newton;
    REVS_REQUIRE(o.k_slabs && d.cand_idx1 && d.cand_cnt1 && d.cand_val1 && d.stats && d.stats_host && d.stats1 && d.stats1_host && d.yhat && d.k_full &&
                 d.info && d.max_pivots > 0 && d.eps > 0, "revs_plan_newton_solve: revs_plan_set_newton / the chain's buffers are missing");
    const int T = d.T, A = REVS_DUAL_AMAX;
    REVS_REQUIRE(T <= 256, "revs_plan_newton_solve: T = %d", T);
    // the blocks the caller refers to still hold the evaluations it saw (every slot's record carries the evaluation's tag)
    for (int blk = 0; blk < 2; ++blk) {
        const double want = blk ? st->pre_tag : st->first_tag;
        if (!(blk ? st->have_pre : st->have_first) || want == 0.0) continue;
        const volatile double *b = blk ? d.stats1_host : d.stats_host;
        for (int t = 0; t < T; ++t)
            REVS_REQUIRE(b[8 * t + 5] == want, "revs_plan_newton_solve: stats block %d no longer holds evaluation %g (slot %d carries %g)",
                         blk, want, t, (double)b[8 * t + 5]);
    }
    hipStream_t s = (hipStream_t)stream;
    int64_t *const ci[2] = {d.cand_idx, d.cand_idx1};
    int32_t *const cc[2] = {d.cand_cnt, d.cand_cnt1};
    double *const cv[2] = {d.cand_val, d.cand_val1};
    double *const sd[2] = {d.stats, d.stats1};
    const double *const sh[2] = {d.stats_host, d.stats1_host};
    const double scale = std::max(std::max(std::fabs(d.vlo), std::fabs(d.vhi)), 1e-300);
    const int64_t mt = (int64_t)d.m * T;
    const bool tf = plan->tree.n > 0;     // rows by the tree form of R p (every shape: revs_op_dual_rows_tree)
    const revs_tree_t trh{plan->tree.n, (const uint64_t *)plan->tree.pack, plan->tree.w};
    double *ycur = st->y, *ytrial = st->y_trial;
    // One evaluation of multipliers yy (p, N, D, the voltage rows, candidate lists and stats into set k; P_est_new =
    // the answer for yy), waited for: the selection tags the pinned stats block behind a system-scope fence.
    auto evaluate = [&](const double *yy, int uy, int k, int sup, double *out /* [T][8] */, int kadd) -> int {
        const double tag = (plan->seq += 1.0) + 0.25;      // (Python's evaluations: n + 0.5; the other native loops: whole numbers)
        auto phase = [&](int ph) -> int {
            if (tf)
                return revs_op_dual_evaluate_tree(ph, d.m, T, d.node_ptr, st->p_est, st->p_sch, st->gamma, d.R, &trh, yy, uy,
                                                  d.kappa, d.vlo, d.vhi, kadd, d.ksplit, d.d_slabs, d.pnq, st->p_est_new, d.vfull,
                                                  d.viol, d.partial, ci[k], cc[k], cv[k], sd[k], tag, stream);
            return revs_op_dual_evaluate(ph, d.m, T, d.node_ptr, st->p_est, st->p_sch, st->gamma, d.R, d.Rt, yy, uy, d.kappa,
                                         d.vlo, d.vhi, kadd, d.ksplit, d.d_slabs, d.v_slabs, d.pnq, st->p_est_new, d.vfull, d.viol,
                                         d.partial, ci[k], cc[k], cv[k], sd[k], tag, plan->counters, stream);
        };
        int rc;
        if (uy && sup >= 0) {             // few multipliers: shifts straight from their rows of R, no dense product
            rc = revs_op_dual_eval_rows(d.m, T, d.node_ptr, st->p_est, st->p_sch, st->gamma, d.R, ci[sup], cc[sup], yy, d.kappa,
                                        d.pnq, st->p_est_new, stream);
            if (rc == REVS_OK && plan->comm) rc = revs_comm_allreduce_f64(plan->comm, d.pnq, 3 * mt, 0, stream);
            if (rc == REVS_OK) rc = phase(2);
        } else if (!plan->comm) {
            rc = phase(3);
        } else {
            rc = phase(1);
            if (rc == REVS_OK) rc = revs_comm_allreduce_f64(plan->comm, d.pnq, 3 * mt, 0, stream);      // the only exchange
            if (rc == REVS_OK) rc = phase(2);
        }
        if (rc != REVS_OK) return rc;
        const volatile double *tg = sh[k];
        const auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < T; ++t) {
            unsigned spins = 0;
            while (tg[8 * t + 5] != tag) {
                if ((++spins & 0xFFFF) == 0) {
                    if (hipStreamQuery(s) == hipSuccess && tg[8 * t + 5] != tag) {
                        revs::set_error("revs_plan_newton_solve: stream idle but stats tag missing");
                        return REVS_ELAUNCH;
                    }
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
                        (void)hipStreamSynchronize(s);
                        revs::set_error("revs_plan_newton_solve: timed out waiting for an evaluation");
                        return REVS_ELAUNCH;
                    }
                }
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        for (int i = 0; i < 8 * T; ++i) out[i] = tg[i];
        return REVS_OK;
    };
    std::vector<double> stt(8 * (size_t)T), stn(8 * (size_t)T), alpha((size_t)T), Dv((size_t)T);
    std::vector<char> pending((size_t)T);
    int cur = 0, rc = REVS_OK;
    if (st->have_first) for (int i = 0; i < 8 * T; ++i) stt[i] = d.stats_host[i];
    else if ((rc = evaluate(ycur, st->use_y, 0, st->use_y ? st->sup : -1, stt.data(), d.kadd)) != REVS_OK) return rc;
    // Rows admitted to a slot's model per Newton iteration: d.kadd (2: the warm solves' models stay small) -- but
    // plan->kadd_cold while some slot still shows more than plan->kadd_cold_at violated rows without a multiplier (a cold
    // solve: admitting two at a time makes it as many Newton iterations as half the rows that end up binding).
    // kadd_stt: what the evaluation behind `stt` admitted with (its candidate lists are that long).
    // ... and only while the rows admitted last time nearly all kept a multiplier (kept >= 0.5) or some slot already carries
    // 16 of them: rows that bind one by one, the 121144 feeder; on long laterals a handful of multipliers clears hundreds of
    // violated rows, most admitted rows end without one and a slot ends with 3-4 multipliers: there the small lists stay.
    int kadd_stt = d.kadd;
    double ns_prev = 0.0, adm_prev = 0.0;
    bool have_prev = false;
    int evals = 1, newton = 0, pivots = 0, stall = 0, n_small = 0, n_general = 0;
    bool ok_all = false, last_small = false, few = false, from_pre = st->have_pre != 0, big_needed = false;
    double best = INFINITY;
    for (;;) {
        double rmax = 0.0, ns_max = 0.0, nc_max = 0.0, nv_max = 0.0, ns_sum = 0.0, adm_now = 0.0;
        bool over = false, full = false;
        for (int t = 0; t < T; ++t) {
            const double *a = &stt[8 * t];
            if (a[2] > A) over = true;                           // more multipliers than a model holds
            const double r = a[0] / scale;
            rmax = std::max(rmax, r);
            // a slot whose model is full of multipliers while rows are still violated cannot take them in
            if (a[2] >= A && a[3] > 0 && r > d.eps) full = true;
            ns_max = std::max(ns_max, a[2]);
            nc_max = std::max(nc_max, a[2] + std::min(a[3], std::min((double)kadd_stt, A - a[2])));
            nv_max = std::max(nv_max, a[3]);
            ns_sum += a[2];
            adm_now += std::min(a[3], std::min((double)kadd_stt, A - a[2]));
        }
        const double kept = have_prev ? (ns_sum - ns_prev) / std::max(adm_prev, 1.0) : 0.0;
        if (over || full) big_needed = true;
        const int kadd_next = (plan->kadd_cold > d.kadd && nv_max > plan->kadd_cold_at && (kept >= 0.5 || ns_max >= 16.0)) ? plan->kadd_cold : d.kadd;
        ns_prev = ns_sum; adm_prev = adm_now; have_prev = true;
        if (over) break;
        if (rmax <= d.eps) { ok_all = true; break; }
        if (newton >= o.newton_max || full) break;
        // ... and a solve that stopped improving is not worth more iterations
        if (rmax < 0.5 * best) { best = rmax; stall = 0; }
        else if (++stall >= 10) break;
        ++newton;
        last_small = nc_max <= 8;
        few = ns_max + kadd_stt <= REVS_DUAL_FEW;
        // (the chain guessed how its trial's home pass gets d = R^T y / kappa -- row-wise or dense; another choice here
        // would differ in the last bits: then the trial is made again)
        const bool use_pre = st->have_pre && newton == 1 && last_small && few == (st->chain_few_in != 0);
        if (!use_pre) {
            if (last_small) {
                ++n_small;
                rc = revs_op_dual_model_small(d.m, T, d.R, d.pnq + mt, ci[cur], cc[cur], cv[cur], d.kappa, d.delta, d.max_pivots,
                                              d.k_full, d.yhat, d.info, stream);
            } else {
                ++n_general;
                rc = revs_op_dual_model(d.m, T, d.R, d.pnq + mt, ci[cur], cc[cur], cv[cur], d.kappa, d.delta, d.max_pivots, o.nks,
                                        o.k_slabs, d.k_full, d.yhat, d.info, stream);
            }
            if (rc != REVS_OK) return rc;
        } else {
            ++n_small;                                            // (the chain ran this model on this set)
        }
        bool any_pending = false;
        for (int t = 0; t < T; ++t) {
            Dv[t] = stt[8 * t + 1];
            pending[t] = stt[8 * t] / scale > d.eps;
            alpha[t] = pending[t] ? 1.0 : 0.0;
        }
        const int nxt = 1 - cur;
        int kadd_stn = kadd_next;
        for (int ls = 0; ls < o.ls_max; ++ls) {
            kadd_stn = (use_pre && ls == 0) ? d.kadd : kadd_next;      // (the chain's trial admitted with the plan's own)
            if (use_pre && ls == 0) {
                for (int i = 0; i < 8 * T; ++i) stn[i] = d.stats1_host[i];      // that trial and its evaluation: already there
            } else {
                from_pre = false;
                for (int t = 0; t < T; ++t) o.alpha_host[t] = alpha[t];        // read by the step kernel through its mapping
                // (the trial starts from the current multipliers: copied by the step's own launch)
                rc = revs::dual_step_copy(T, ci[cur], cc[cur], cv[cur], d.yhat, o.alpha_dev, ycur, d.m, ytrial, sd[nxt] + 4, stream);
                if (rc == REVS_OK) rc = evaluate(ytrial, 1, nxt, few ? cur : -1, stn.data(), kadd_next);
                if (rc != REVS_OK) return rc;
            }
            ++evals;
            // (slack 1e-11 |D|: the evaluations sum the squares rounded to 2^-32 so that the sums do not depend on their
            // order -- a rounding of ~1e-13 |D| per evaluation)
            any_pending = false;
            for (int t = 0; t < T; ++t) {
                const bool okk = stn[8 * t + 1] >= Dv[t] + 1e-4 * stn[8 * t + 4] - 1e-11 * std::fabs(Dv[t]);
                if (okk) pending[t] = 0;
                if (pending[t]) { any_pending = true; alpha[t] *= 0.5; }
            }
            if (!any_pending) break;
        }
        for (int t = 0; t < T; ++t) pivots += std::abs(o.info_host[t]);     // (the evaluation was waited for)
        if (any_pending) break;                                   // no ascent found: leave it to the ADMM forms
        std::swap(ycur, ytrial);
        cur = nxt;
        stt.swap(stn);
        kadd_stt = kadd_stn;
    }
    st->y = ycur;
    st->y_trial = ytrial;
    st->ok = ok_all;
    st->newton = newton;
    st->evals = evals;
    st->pivots = pivots;
    st->models_small = n_small;
    st->models_general = n_general;
    st->last_small = last_small;
    st->few = newton >= 1 ? (few ? 1 : 0) : 0;
    st->pre_kept = ok_all && from_pre && newton <= 1;
    st->cur = cur;
    double sum = 0.0, mx = 0.0;
    for (int t = 0; t < T; ++t) { sum += stt[8 * t + 2]; mx = std::max(mx, stt[8 * t + 2]); }
    st->nsup_sum = (int32_t)sum;
    st->nsup_max = (int32_t)mx;
    st->big_needed = big_needed ? 1 : 0;
    st->reserved_ = 0;
    if (!ok_all && !big_needed && hipMemsetAsync(ycur, 0, sizeof(double) * mt, s) != hipSuccess) {
        revs::set_error("revs_plan_newton_solve: clearing the multipliers failed");
        return REVS_ELAUNCH;
    }
    return REVS_OK;
}

extern "C" int revs_plan_chain_run(revs_plan_t *plan, int32_t max_steps, revs_chain_state_t *st,
                                   int32_t chain_few, int32_t *kept_steps, void *stream) {
    REVS_REQUIRE(plan && max_steps >= 0 && st && kept_steps && st->y && st->y_trial && st->p_est &&
                 st->p_est_new && st->p_sch && st->p_sch_alt && st->gamma && st->gamma_alt,
                 "revs_plan_chain_run: bad argument");
    *kept_steps = 0;
    for (int32_t k = 0; k < max_steps; ++k) {
        int32_t acc = 0, nsum = 0, nmax = 0;
        const int rc = revs_plan_chain_step(plan, st->y, st->y_trial, st->use_y, st->sup0, chain_few,
                                            st->p_est, st->p_est_new, st->p_sch, st->gamma,
                                            st->p_sch_alt, st->gamma_alt, nullptr, nullptr, &acc, &nsum,
                                            &nmax, nullptr, nullptr, stream);
        if (rc != REVS_OK) return rc;
        if (!acc) return REVS_OK;              // the caller's general loop takes this iteration
        std::swap(st->y, st->y_trial);
        st->use_y = nsum > 0;
        st->sup0 = (nsum > 0 && nmax + plan->d.kadd <= REVS_DUAL_FEW) ? 1 : -1;
        std::swap(st->p_sch, st->p_sch_alt);
        std::swap(st->gamma, st->gamma_alt);
        std::swap(st->p_est, st->p_est_new);
        ++*kept_steps;
    }
    return REVS_OK;
}


// The binding steady state with ONE pass over the residences per ADMM iteration (see revs_admm.h).
// Per iteration k (parity par = k & 1, candidate sets / stats S0[par], S1[par]):
//   sweep     the residences' iteration with the operator's answer for the trial multipliers formed
//             inside (shifts from S0[par]'s lists), P_sch / G to the spares, pen to p_est_new; folds
//             the trial's node sums into fold_e2[par] and the sums of the same multipliers on the
//             new state into fold_e1[par ^ 1]
//   KV        [0, T): rows + selection of the trial -> S1[par] (the verdict the host polls);
//             [T, 2T): rows, selection, small model, step of iteration k + 1 -> S0[par ^ 1], the
//             next trial in y_spare; clears fold_e2[par ^ 1], fold_e1[par]
// then revs_newton_chain_accept on S0[par], S1[par]; accepted: roles rotate and iteration k + 1
// starts with its sweep -- its operator work is done.  The first iteration of a call that does not
// resume evaluates the multipliers with the evaluation kernel first.
static int fold_alloc(revs_plan_t *plan) {
    const revs_plan_desc_t &d = plan->d;
    if (plan->fold_e2[0]) return REVS_OK;
    const size_t mt = (size_t)d.m * d.T;
    hipError_t e = hipSuccess;
    auto dev = [&](void **p, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(p, bytes);
        if (e == hipSuccess) e = hipMemset(*p, 0, bytes);
    };
    for (int i = 0; i < 2; ++i) {
        // the two sum arrays one sweep accumulates into -- fold_e2[par] | fold_e1[par ^ 1] -- are one
        // allocation: sharded, ONE all-reduce per iteration covers both
        // (double[m][T][4] = {p, N, q, 0} each)
        dev((void **)&plan->fold_e2[i], sizeof(double) * 8 * mt);
        if (e == hipSuccess) plan->fold_e1[i ^ 1] = plan->fold_e2[i] + 4 * mt;
        dev((void **)&plan->fold_ci[i], sizeof(int64_t) * (size_t)d.T * REVS_DUAL_AMAX);
        dev((void **)&plan->fold_cc[i], sizeof(int32_t) * (size_t)d.T);
        dev((void **)&plan->fold_cv[i], sizeof(double) * (size_t)d.T * 3 * REVS_DUAL_AMAX);
        if (e == hipSuccess) {
            void *h = nullptr, *dp = nullptr;
            e = hipHostMalloc(&h, sizeof(double) * 8 * (size_t)d.T, hipHostMallocMapped);
            if (e == hipSuccess) {
                memset(h, 0, sizeof(double) * 8 * (size_t)d.T);
                e = hipHostGetDevicePointer(&dp, h, 0);
            }
            plan->fold_st_host[i] = (double *)h;
            plan->fold_st_dev[i] = (double *)dp;
        }
    }
    dev((void **)&plan->fold_st_local[0], sizeof(double) * 8 * (size_t)d.T);
    dev((void **)&plan->fold_st_local[1], sizeof(double) * 8 * (size_t)d.T);
    dev((void **)&plan->fold_v[0], sizeof(double) * mt);
    dev((void **)&plan->fold_v[1], sizeof(double) * mt);
    dev((void **)&plan->fold_v[2], sizeof(double) * (size_t)d.T * 4);
    dev((void **)&plan->fold_info[0], sizeof(int32_t) * (size_t)d.T);
    dev((void **)&plan->fold_info[1], sizeof(int32_t) * (size_t)d.T);
    dev((void **)&plan->fold_sh[0], sizeof(double) * mt);
    dev((void **)&plan->fold_sh[1], sizeof(double) * (mt + 32 * (size_t)d.T));     // (+ the tuning build's stage stamps)
    if (e != hipSuccess) {
        revs::set_error("revs_plan_chain_fold_run: allocating the folded chain's buffers: %s", hipGetErrorString(e));
        return REVS_ELAUNCH;
    }
    return REVS_OK;
}

extern "C" int revs_plan_chain_fold_run(revs_plan_t *plan, int32_t max_steps, revs_chain_fold_state_t *st,
                                        int32_t *kept_steps, void *stream) {
    REVS_REQUIRE(plan && max_steps >= 0 && st && kept_steps && st->y && st->y_trial && st->y_spare &&
                 st->y != st->y_trial && st->y != st->y_spare && st->y_trial != st->y_spare && st->p_est &&
                 st->p_est_new && st->p_sch && st->p_sch_alt && st->gamma && st->gamma_alt,
                 "revs_plan_chain_fold_run: bad argument");
    const revs_plan_desc_t &d = plan->d;
    REVS_REQUIRE(plan->tree.n > 0 && plan->tree.n <= REVS_TREE_SWEEP_MAX && d.node_of && d.cand_idx1 && d.cand_cnt1 &&
                 d.cand_val1 && d.stats1 && d.stats1_host && d.yhat && d.k_full && d.info && d.max_pivots > 0 &&
                 d.eps > 0 && !d.pdhg.full_rows,
                 "revs_plan_chain_fold_run: needs the feeder as a tree (at most %d nodes), node_of, the chain's "
                 "buffers and the presolved PDHG form", REVS_TREE_SWEEP_MAX);
    REVS_REQUIRE(d.m <= REVS_CHAIN_FOLD_MAX_M, "revs_plan_chain_fold_run: m = %d constraint nodes, the folded chain's "
                 "operator launch holds %d (use revs_plan_chain_run)", d.m, REVS_CHAIN_FOLD_MAX_M);
    *kept_steps = 0;
    if (fold_alloc(plan) != REVS_OK) return REVS_ELAUNCH;
    hipStream_t s = (hipStream_t)stream;
    struct Set { int64_t *ci; int32_t *cc; double *cv; double *st; const double *st_host; };
    auto set_of = [&](int par, int which) -> Set {
        if (par == 0)
            return which == 0 ? Set{d.cand_idx, d.cand_cnt, d.cand_val, d.stats, d.stats_host}
                              : Set{d.cand_idx1, d.cand_cnt1, d.cand_val1, d.stats1, d.stats1_host};
        return Set{plan->fold_ci[which], plan->fold_cc[which], plan->fold_cv[which], plan->fold_st_dev[which],
                   plan->fold_st_host[which]};
    };
    const double scale = std::max(std::max(std::fabs(d.vlo), std::fabs(d.vhi)), 1e-300);
    const revs_tree_t trh{plan->tree.n, (const uint64_t *)plan->tree.pack, plan->tree.w};
    const int64_t mt = (int64_t)d.m * d.T;
    bool have_k1 = st->resume != 0 && plan->fold_ready;
    int par = have_k1 ? plan->fold_par : 0;
    plan->fold_ready = false;
    st->resume = 0;
    int rc = REVS_OK;
    st->redone = 0;
    st->pivots = 0;
    // redo: Newton steps beyond the first that the current iteration has taken.  A trial that passes the
    // line search but leaves the rows above the tolerance IS the general loop's next Newton iterate, and its
    // evaluation on the current state is what the sweep has just folded (fold_e2[par]): the operator launch
    // without a verdict half runs rows / selection / model / step on those sums, and the iteration's sweep
    // and operator launch are made again from there -- the general loop's iterates, without its round trips.
    int redo = 0;
    bool redo_pending = false;
    const int kMaxRedo = plan->fold_redo;
    // Sweeps enqueued ahead of their iteration's turn (st->p_est_3 ...): `swept` = this iteration's sweep is in the
    // queue already.  Every sweep ORs its residences' status bits into its own host-visible word (three rotate: at
    // most two sweeps are unjudged at any time); a word joins the sticky one when its iteration is kept.
    REVS_REQUIRE((st->p_est_3 != nullptr) == (st->p_sch_3 != nullptr) && (st->p_est_3 != nullptr) == (st->gamma_3 != nullptr),
                 "revs_plan_chain_fold_run: the third set of state buffers is all three or none");
#ifdef REVS_TUNING        // (debugging aids of tuning builds; the product build has no process-wide toggles in this loop)
    static const bool no_spec = getenv("REVS_FOLD_NO_SPEC") != nullptr;
#else
    constexpr bool no_spec = false;
#endif
    const bool can_spec = st->p_est_3 != nullptr && !no_spec;
    bool swept = false;
    unsigned int sweep_no = 0;
    volatile unsigned int *const fwords = plan->flags_host ? (volatile unsigned int *)plan->flags_host + 1 : nullptr;
    if (fwords) fwords[0] = fwords[1] = fwords[2] = 0u;
    const bool warm = d.mode == REVS_MODE_RELAXED_PDHG && d.pdhg_dual != nullptr;
    REVS_REQUIRE(!warm || !st->pdhg_dual || (st->pdhg_dual == d.pdhg_dual && st->pdhg_dual_new && st->pdhg_dual_new != st->pdhg_dual &&
                                              (!st->p_est_3 || (st->pdhg_dual_3 && st->pdhg_dual_3 != st->pdhg_dual &&
                                                                st->pdhg_dual_3 != st->pdhg_dual_new))),
                 "revs_plan_chain_fold_run: pdhg_dual must be the plan's, with distinct spares");
    const bool ybuf = warm && st->pdhg_dual != nullptr;      // (else: updated in place, as before round 4)
    auto sweep = [&](int parity, const float *pe, const float *ps, const float *gm, float *pe_out, float *ps_out, float *gm_out,
                     float *s_out, float *c_out, float *y_in, float *y_out) -> int {
        revs::ChainFold cf{plan->fold_sh[0], plan->fold_sh[1], d.m, d.kappa, plan->fold_e2[parity],
                           plan->fold_e1[parity ^ 1], pe_out};
        cf.y_out = ybuf ? y_out : nullptr;
        cf.wg_order = plan_wg_order(plan);
        int r = revs::agent_step_chain(d.n_homes, d.T, d.cost, d.homes, d.load, pe, ps, gm, ps_out, gm_out, s_out, c_out, d.diff,
                                       d.dsq, d.status, ybuf ? y_in : d.pdhg_dual, (float)d.kappa, d.mode, &d.pdhg, d.node_of, cf,
                                       plan->flags_dev ? plan->flags_dev + 1 + sweep_no % 3u : nullptr, stream);
        ++sweep_no;
        // Residences sharded: every rank's sweep has folded its own residences' addends -- exact and order-independent
        // (revs_q36 / revs_q32), so the all-reduced sums are the one-process sums bit for bit.  Both arrays in ONE
        // collective per iteration (8 M T doubles: {p, N, q, 0} per slot and node, twice); everything behind it is
        // replicated and deterministic.
        if (r == REVS_OK && plan->comm) r = revs_comm_allreduce_f64(plan->comm, plan->fold_e2[parity], 8 * mt, 0, stream);
        return r;
    };
    for (int32_t k = 0; k < max_steps; ++k) {
        const Set S0 = set_of(par, 0), S1 = set_of(par, 1), S0n = set_of(par ^ 1, 0), S1n = set_of(par ^ 1, 1);
        if (redo_pending) {
            redo_pending = false;
            revs::ChainKv c0{};
            c0.m = d.m; c0.T = d.T; c0.kadd = d.kadd; c0.has_e2 = 0;
            c0.tree = plan->tree;
            c0.vlo = d.vlo; c0.vhi = d.vhi; c0.kappa = d.kappa; c0.delta = d.delta; c0.scale = scale; c0.eps = d.eps;
            c0.max_pivots = d.max_pivots;
            c0.e1 = revs::ChainKvSide{plan->fold_e2[par], st->y, d.vfull, d.viol, d.partial, S0.ci, S0.cc, S0.cv, S0.st, 0.0, 4};
            plan->fold_st_local_valid = false;      // (this launch writes the host block itself)
            c0.R = d.R; c0.k_full = d.k_full; c0.yhat = d.yhat; c0.info = plan->fold_info[par];
            c0.y_trial = st->y_trial;
            c0.lin_out = S1.st + 4;
            c0.clr0 = nullptr; c0.clr1 = nullptr;
            c0.sh_a = plan->fold_sh[0]; c0.sh_b = plan->fold_sh[1];
            rc = revs::chain_kv_launch(c0, stream);
            if (rc != REVS_OK) return rc;
            if (hipMemsetAsync(plan->fold_e2[par], 0, sizeof(double) * 4 * mt, s) != hipSuccess ||
                hipMemsetAsync(plan->fold_e1[par ^ 1], 0, sizeof(double) * 4 * mt, s) != hipSuccess) {
                revs::set_error("revs_plan_chain_fold_run: hipMemsetAsync failed");
                return REVS_ELAUNCH;
            }
        } else if (!have_k1) {
            // entry: the multipliers' evaluation by the evaluation kernel (row-wise shifts from the
            // caller's list `sup0` when it has one), rows / selection / model / step in one launch
            if (st->use_y && st->sup0 >= 0) {
                int64_t *const ci[2] = {d.cand_idx, d.cand_idx1};
                int32_t *const cc[2] = {d.cand_cnt, d.cand_cnt1};
                rc = revs_op_dual_eval_rows(d.m, d.T, d.node_ptr, st->p_est, st->p_sch, st->gamma, d.R, ci[st->sup0],
                                            cc[st->sup0], st->y, d.kappa, d.pnq, st->p_est_new, stream);
            } else {
                rc = revs_op_dual_evaluate(1, d.m, d.T, d.node_ptr, st->p_est, st->p_sch, st->gamma, d.R, d.Rt, st->y,
                                           st->use_y, d.kappa, d.vlo, d.vhi, d.kadd, d.ksplit, d.d_slabs, d.v_slabs,
                                           d.pnq, st->p_est_new, d.vfull, d.viol, d.partial, d.cand_idx, d.cand_cnt,
                                           d.cand_val, d.stats, 0.0, nullptr, stream);
            }
            if (rc != REVS_OK) return rc;
            if (plan->comm && (rc = revs_comm_allreduce_f64(plan->comm, d.pnq, 3 * mt, 0, stream)) != REVS_OK) return rc;
            {   // rows, selection, small model, step and the trial's shifts: the operator launch without a trial to judge
                revs::ChainKv c0{};
                c0.m = d.m; c0.T = d.T; c0.kadd = d.kadd; c0.has_e2 = 0;
                c0.tree = plan->tree;
                c0.vlo = d.vlo; c0.vhi = d.vhi; c0.kappa = d.kappa; c0.delta = d.delta; c0.scale = scale; c0.eps = d.eps;
                c0.max_pivots = d.max_pivots;
                c0.e1 = revs::ChainKvSide{d.pnq, st->y, d.vfull, d.viol, d.partial, S0.ci, S0.cc, S0.cv, S0.st, 0.0, 1};
                plan->fold_st_local_valid = false;
                c0.R = d.R; c0.k_full = d.k_full; c0.yhat = d.yhat; c0.info = plan->fold_info[par];
                c0.y_trial = st->y_trial;
                c0.lin_out = S1.st + 4;
                c0.clr0 = nullptr; c0.clr1 = nullptr;
                c0.sh_a = plan->fold_sh[0]; c0.sh_b = plan->fold_sh[1];
                rc = revs::chain_kv_launch(c0, stream);
            }
            if (rc != REVS_OK) return rc;
            // (the arrays this iteration's sweep accumulates into start from zero)
            if (hipMemsetAsync(plan->fold_e2[par], 0, sizeof(double) * 4 * mt, s) != hipSuccess ||
                hipMemsetAsync(plan->fold_e1[par ^ 1], 0, sizeof(double) * 4 * mt, s) != hipSuccess) {
                revs::set_error("revs_plan_chain_fold_run: hipMemsetAsync failed");
                return REVS_ELAUNCH;
            }
        }
        const unsigned int word_k = swept ? (sweep_no - 1u) % 3u : sweep_no % 3u;      // this iteration's sweep's status word
        if (!swept) {
            rc = sweep(par, st->p_est, st->p_sch, st->gamma, st->p_est_new, st->p_sch_alt, st->gamma_alt, st->s_out, st->c_out,
                       st->pdhg_dual, st->pdhg_dual_new);
            if (rc != REVS_OK) return rc;
        }
        swept = false;
        const double seq = -(plan->seq += 1.0);
        revs::ChainKv c{};
        c.m = d.m; c.T = d.T; c.kadd = d.kadd; c.has_e2 = 1;
        c.tree = plan->tree;
        c.vlo = d.vlo; c.vhi = d.vhi; c.kappa = d.kappa; c.delta = d.delta; c.scale = scale; c.eps = d.eps;
        c.max_pivots = d.max_pivots;
        c.e2 = revs::ChainKvSide{plan->fold_e2[par], st->y_trial, plan->fold_v[0], plan->fold_v[1], plan->fold_v[2],
                                 S1.ci, S1.cc, S1.cv, S1.st, seq, 4};
        // (the next iteration's stats stay on the device; this launch's verdict half hands the host the ones the launch
        // before left there for THIS iteration's acceptance test)
        c.e1 = revs::ChainKvSide{plan->fold_e1[par ^ 1], st->y_trial, d.vfull, d.viol, d.partial,
                                 S0n.ci, S0n.cc, S0n.cv, plan->fold_st_local[par ^ 1], 0.0, 4};
        if (plan->fold_st_local_valid) { c.fwd_src = plan->fold_st_local[par]; c.fwd_dst = S0.st; }
        plan->fold_st_local_valid = true;
        c.R = d.R; c.k_full = d.k_full; c.yhat = d.yhat; c.info = plan->fold_info[par ^ 1];     // (iteration k + 1's model)
        c.y_trial = st->y_spare;
        c.lin_out = S1n.st + 4;
        c.clr0 = plan->fold_e2[par ^ 1];
        c.clr1 = plan->fold_e1[par];
        c.sh_a = plan->fold_sh[0];
        c.sh_b = plan->fold_sh[1];
        c.prev_cidx = S0.ci;                  // the lists of the launch whose step wrote y_trial
        c.prev_ccnt = S0.cc;
        rc = revs::chain_kv_launch(c, stream);
        if (rc != REVS_OK) return rc;
        // The next iteration's sweep, unjudged: it needs this launch's shifts and cleared sum arrays (stream order) and
        // the state this iteration's sweep wrote; its own output goes to the third set.  (Not behind an iteration that
        // took extra Newton steps: the call returns behind that one.)
        const bool spec = can_spec && k + 1 < max_steps && redo == 0;
        if (spec) {
            rc = sweep(par ^ 1, st->p_est_new, st->p_sch_alt, st->gamma_alt, st->p_est_3, st->p_sch_3, st->gamma_3, nullptr, nullptr,
                       st->pdhg_dual_new, st->pdhg_dual_3);
            if (rc != REVS_OK) return rc;
        }
        // the trial's verdict: poll its tags (pinned memory), then the driver's own acceptance test
        const volatile double *tg = S1.st_host;
        const auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < d.T; ++t) {
            unsigned spins = 0;
            while (tg[8 * t + 5] != seq) {
                if ((++spins & 0xFFFF) == 0) {
                    if (hipStreamQuery(s) == hipSuccess && tg[8 * t + 5] != seq) {
                        revs::set_error("revs_plan_chain_fold_run: stream idle but stats tag missing");
                        return REVS_ELAUNCH;
                    }
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
                        (void)hipStreamSynchronize(s);
                        revs::set_error("revs_plan_chain_fold_run: timed out waiting for the trial's verdict");
                        return REVS_ELAUNCH;
                    }
                }
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        int32_t nsum = 0, nmax = 0;
        int why = 0;
        const int acc = chain_accept_impl(d.T, S0.st_host, S1.st_host, scale, d.eps, REVS_DUAL_AMAX, d.kadd, 1,
                                          &nsum, &nmax, &why);
#ifdef REVS_TUNING
        static const bool ftrace = getenv("REVS_FOLD_TRACE") != nullptr;
#else
        constexpr bool ftrace = false;
#endif
        if (ftrace && !acc) {
            double r0 = 0, r1 = 0, ncm = 0;
            int arm = 0;
            for (int t = 0; t < d.T; ++t) {
                const double *a = S0.st_host + 8 * t, *b = S1.st_host + 8 * t;
                r0 = std::max(r0, a[0] / scale);
                r1 = std::max(r1, b[0] / scale);
                ncm = std::max(ncm, a[2] + std::min(a[3], (double)d.kadd));
                if (a[0] / scale > d.eps && !(b[1] >= a[1] + 1e-4 * b[4] - 1e-11 * std::fabs(a[1]))) {
                    ++arm;
                    fprintf(stderr, "   slot %d: D0 %.17g D1 %.17g lin %.6g gain %.6g rows0 %.3g rows1 %.3g ns %g nv %g\n", t, a[1], b[1],
                            b[4], b[1] - a[1], a[0] / scale, b[0] / scale, a[2], a[3]);
                }
            }
            fprintf(stderr, "[fold] iteration %d (par %d, resumed %d) rejected: rows before %.3g after %.3g candidates %g armijo failures %d\n",
                    k, par, (int)have_k1, r0, r1, ncm, arm);
        }
        if (!acc) {
            // The speculative sweep of a rejected or redone iteration does not stand, and neither does what it
            // said about its own problems: "a PDHG residence stopped at its cap" is dropped from the sticky
            // status word, as revs_plan_stream_run_blocks does behind a roll-back (the sweep that replaces it
            // sets the bit again if it is true of the problem that counts; "no solution" does not depend on
            // the estimate: kept).  The carried PDHG multipliers ARE left where that sweep put them: another
            // warm start of the same problems.
            // (each sweep has its own word: this one's and the unjudged next one's are dropped -- the latter once it
            // has run; "no solution" does not depend on the estimate: kept)
            if (spec && hipStreamSynchronize(s) != hipSuccess) {
                revs::set_error("revs_plan_chain_fold_run: waiting for the unjudged sweep failed");
                return REVS_ELAUNCH;
            }
            if (fwords) {
                *(volatile unsigned int *)plan->flags_host |= (fwords[word_k] | (spec ? fwords[(word_k + 1u) % 3u] : 0u)) & 1u;
                fwords[word_k] = 0u;
                if (spec) fwords[(word_k + 1u) % 3u] = 0u;
            }
            // The caller's general loop takes this iteration (state untouched).  When the trial is a good
            // Newton step that merely left the rows above the tolerance -- the usual rejection with on/off
            // chargers -- the multipliers are handed back AT the trial (resume = 2): the caller goes on
            // from it instead of making the same step again.
            if (why == 1 && redo < kMaxRedo) {
                double *y_old = st->y;          // y := the step; the next trial goes where the speculative one went
                st->y = st->y_trial;
                st->y_trial = st->y_spare;
                st->y_spare = y_old;
                st->use_y = 1;
                st->sup0 = -1;
                ++redo;
                redo_pending = true;
                --k;
                continue;
            }
            st->redone = redo;
            if (why == 1) {
                std::swap(st->y, st->y_trial);
                st->use_y = 1;
                st->sup0 = -1;
                st->resume = 2;
                // (the pivots the step's model took: the caller's books count them with the solve it finishes)
                std::vector<int32_t> inf((size_t)d.T, 0);
                if (hipMemcpyAsync(inf.data(), plan->fold_info[par], sizeof(int32_t) * inf.size(), hipMemcpyDeviceToHost, s) != hipSuccess ||
                    hipStreamSynchronize(s) != hipSuccess) {
                    revs::set_error("revs_plan_chain_fold_run: reading the pivot counts failed");
                    return REVS_ELAUNCH;
                }
                st->pivots = 0;
                for (int32_t v : inf) st->pivots += v < 0 ? -v : v;
            }
            return REVS_OK;
        }
        if (fwords) {                         // this iteration's sweep stands: its status bits join the sticky word
            *(volatile unsigned int *)plan->flags_host |= fwords[word_k];
            fwords[word_k] = 0u;
        }
        double *y_old = st->y;
        st->y = st->y_trial;
        st->y_trial = st->y_spare;
        st->y_spare = y_old;
        st->use_y = nsum > 0;
        st->sup0 = -1;
        std::swap(st->p_sch, st->p_sch_alt);
        std::swap(st->gamma, st->gamma_alt);
        std::swap(st->p_est, st->p_est_new);
        if (ybuf) { std::swap(st->pdhg_dual, st->pdhg_dual_new); plan->d.pdhg_dual = st->pdhg_dual; }
        if (spec) {                           // (state k + 1 is current; the unjudged sweep read it and wrote the third set)
            std::swap(st->p_sch_alt, st->p_sch_3);
            std::swap(st->gamma_alt, st->gamma_3);
            std::swap(st->p_est_new, st->p_est_3);
            if (ybuf) std::swap(st->pdhg_dual_new, st->pdhg_dual_3);
            swept = true;
        }
        st->s_out = nullptr;                  // (schedules are written by the call's first iteration only)
        st->c_out = nullptr;
        ++*kept_steps;
        par ^= 1;
#ifdef REVS_TUNING
        static const bool no_pipe = getenv("REVS_FOLD_NO_PIPE") != nullptr;
#else
        constexpr bool no_pipe = false;
#endif
        have_k1 = !no_pipe;
        if (redo > 0) {                       // (the caller books this iteration's extra Newton steps: it is the call's last)
            st->redone = redo;
            break;
        }
    }
    plan->fold_ready = have_k1;
    plan->fold_par = par;
    st->resume = have_k1 ? 1 : 0;
#ifdef REVS_KV_STAMPS
    {
        std::vector<double> h(32 * (size_t)d.T);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), plan->fold_sh[1] + mt, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
        int worst = 0;
        for (int t = 0; t < d.T; ++t)
            if (h[32 * t + 20] - h[32 * t] > h[32 * worst + 20] - h[32 * worst]) worst = t;
        // 0 start | 1-6 rows | 7-10 selection | 11-16 model | 17 step | 18-20 shifts: microseconds since the slot's start
        fprintf(stderr, "[kv stamps, us since start] slowest slot %d:", worst);
        for (int i = 1; i <= 20; ++i) fprintf(stderr, " %d:%.1f", i, (h[32 * worst + i] - h[32 * worst]) * 0.01);
        fprintf(stderr, " | violated %g support %g room %g | fast body at %.1f, list read %.1f", h[32 * worst + 24], h[32 * worst + 25], h[32 * worst + 26],
                (h[32 * worst + 27] - h[32 * worst]) * 0.01, (h[32 * worst + 28] - h[32 * worst]) * 0.01);
        fprintf(stderr, " | prologue: round-1 loads issued %.1f, LDS cleared %.1f, list in %.1f, gathers issued %.1f, multipliers in %.1f, rows of R requested %.1f",
                (h[32 * worst + 21] - h[32 * worst]) * 0.01, (h[32 * worst + 22] - h[32 * worst]) * 0.01, (h[32 * worst + 23] - h[32 * worst]) * 0.01,
                (h[32 * worst + 29] - h[32 * worst]) * 0.01, (h[32 * worst + 30] - h[32 * worst]) * 0.01, (h[32 * worst + 31] - h[32 * worst]) * 0.01);
        fprintf(stderr, "\n[kv stamps, mean over slots]            ");
        for (int i = 1; i <= 20; ++i) {
            double acc = 0;
            for (int t = 0; t < d.T; ++t) acc += (h[32 * t + i] - h[32 * t]) * 0.01;
            fprintf(stderr, " %d:%.1f", i, acc / d.T);
        }
        fprintf(stderr, "\n");
    }
#endif
    return REVS_OK;
}

// ---- RCCL communicator owned by the library (see revs_admm.h) ---------------------------
// librccl.so.1 is opened at run time: the copy already mapped into the process when there is
// one (PyTorch's), else the system's.  Only the handful of entry points used here is bound.
struct Id128 { char b[128]; };          // ncclUniqueId: 128 bytes, passed by value
namespace {
struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id128, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
}  // namespace
static Rccl g_rccl;

static bool rccl_load() {
    if (g_rccl.h) return true;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW);
    if (!h) h = dlopen("librccl.so", RTLD_NOW);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW);
    if (!h) {
        revs::set_error("revs_comm: cannot open librccl.so.1: %s", dlerror());
        return false;
    }
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce) {
        revs::set_error("revs_comm: librccl.so.1 lacks an expected entry point");
        return false;
    }
    g_rccl.h = h;
    return true;
}

struct revs_comm {
    void *nccl;                             // RCCL communicator, or NULL: the caller's own transport
    int rank, nranks;
    revs_host_allreduce_fn fn = nullptr;    // host-staged all-reduce supplied by the caller
    void *ctx = nullptr;
    double *stage = nullptr;                // pinned staging buffer of the hook form
    size_t stage_count = 0;
};

extern "C" int revs_comm_unique_id(void *id128_out) {
    REVS_REQUIRE(id128_out, "revs_comm_unique_id: null argument");
    if (!rccl_load()) return REVS_ELAUNCH;
    const int rc = g_rccl.GetUniqueId(id128_out);
    if (rc != 0) {
        revs::set_error("ncclGetUniqueId: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
        return REVS_ELAUNCH;
    }
    return REVS_OK;
}

extern "C" revs_comm_t *revs_comm_create(const void *id128, int32_t rank, int32_t nranks) {
    if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) {
        revs::set_error("revs_comm_create: bad argument");
        return nullptr;
    }
    if (!rccl_load()) return nullptr;
    Id128 id;
    memcpy(id.b, id128, sizeof(id.b));
    void *c = nullptr;
    const int rc = g_rccl.CommInitRank(&c, nranks, id, rank);
    if (rc != 0 || !c) {
        revs::set_error("ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
        return nullptr;
    }
    return new revs_comm{c, rank, nranks};
}

extern "C" revs_comm_t *revs_comm_create_hook(revs_host_allreduce_fn fn, void *ctx, int32_t rank,
                                              int32_t nranks) {
    if (!fn || nranks < 1 || rank < 0 || rank >= nranks) {
        revs::set_error("revs_comm_create_hook: bad argument");
        return nullptr;
    }
    revs_comm *c = new revs_comm{nullptr, rank, nranks};
    c->fn = fn;
    c->ctx = ctx;
    return c;
}

extern "C" void revs_comm_destroy(revs_comm_t *comm) {
    if (!comm) return;
    if (comm->nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comm->nccl);
    if (comm->stage) (void)hipHostFree(comm->stage);
    delete comm;
}

// The hook form: everything enqueued on `stream` so far is waited for, the buffer goes through
// pinned host memory to the caller's function and back.  Synchronous by construction -- it is
// the transport of a caller that has no RCCL path between its ranks (two ranks sharing one
// device, MPI over the host, a test harness), not a fast path.
static int comm_allreduce_hook(revs_comm_t *comm, double *buf, int64_t count, int32_t op, hipStream_t s) {
    if (comm->stage_count < (size_t)count) {
        if (comm->stage) (void)hipHostFree(comm->stage);
        comm->stage = nullptr;
        comm->stage_count = 0;
        void *h = nullptr;
        if (hipHostMalloc(&h, sizeof(double) * (size_t)count, hipHostMallocDefault) != hipSuccess) {
            revs::set_error("revs_comm_allreduce_f64: hipHostMalloc of the staging buffer failed");
            return REVS_ELAUNCH;
        }
        comm->stage = (double *)h;
        comm->stage_count = (size_t)count;
    }
    hipError_t e = hipMemcpyAsync(comm->stage, buf, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        revs::set_error("revs_comm_allreduce_f64: staging to the host: %s", hipGetErrorString(e));
        return REVS_ELAUNCH;
    }
    const int rc = comm->fn(comm->ctx, comm->stage, count, op);
    if (rc != 0) {
        revs::set_error("revs_comm_allreduce_f64: the caller's all-reduce returned %d", rc);
        return REVS_ELAUNCH;
    }
    e = hipMemcpyAsync(buf, comm->stage, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);          // (the staging buffer is reused)
    if (e != hipSuccess) {
        revs::set_error("revs_comm_allreduce_f64: staging back to the device: %s", hipGetErrorString(e));
        return REVS_ELAUNCH;
    }
    return REVS_OK;
}

extern "C" int revs_comm_allreduce_f64(revs_comm_t *comm, double *buf, int64_t count, int32_t op,
                                       void *stream) {
    REVS_REQUIRE(comm && buf && count > 0 && (op == 0 || op == 2 || op == 3),
                 "revs_comm_allreduce_f64: bad argument");
    if (comm->fn) return comm_allreduce_hook(comm, buf, count, op, (hipStream_t)stream);
    // ncclFloat64 = 8; ncclSum / ncclMax / ncclMin = 0 / 2 / 3 (rccl.h)
    const int rc = g_rccl.AllReduce(buf, buf, (size_t)count, 8, op, comm->nccl, (hipStream_t)stream);
    if (rc != 0) {
        revs::set_error("ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
        return REVS_ELAUNCH;
    }
    return REVS_OK;
}

// ---- streaming steady state (see revs_admm.h) ------------------------------------------
extern "C" int revs_plan_set_tree(revs_plan_t *plan, const revs_tree_t *t) {
    REVS_REQUIRE(plan, "revs_plan_set_tree: null plan");
    if (!t || t->n == 0) { plan->tree = revs::TreeArgs{}; return REVS_OK; }
    REVS_REQUIRE(t->n > 0 && t->n <= REVS_TREE_MAX && t->n % revs::tree_shape(t->n).ipt == 0 && t->pack && t->w,
                 "revs_plan_set_tree: bad tree (at most %d nodes, a multiple of 8; of 16 beyond 8192)", REVS_TREE_MAX);
    plan->tree = revs::TreeArgs{t->n, (const unsigned long long *)t->pack, t->w};
    return REVS_OK;
}

extern "C" int revs_plan_set_comm(revs_plan_t *plan, revs_comm_t *comm) {
    REVS_REQUIRE(plan, "revs_plan_set_comm: null plan");
    plan->comm = comm;
    return REVS_OK;
}

extern "C" int revs_plan_set_stream_block(revs_plan_t *plan, int32_t block, int32_t overlap) {
    REVS_REQUIRE(plan && block >= 0 && block <= REVS_STREAM_BLOCK_MAX,
                 "revs_plan_set_stream_block: block=%d outside 0..%d", block, REVS_STREAM_BLOCK_MAX);
    if (block <= 1) { plan->block = 0; return REVS_OK; }
    const revs_plan_desc_t &d = plan->d;
    REVS_REQUIRE(d.n_homes > 0 && d.node_of, "revs_plan_set_stream_block: the plan has no residences / node_of");
    REVS_REQUIRE(d.recompute_pe_new, "revs_plan_set_stream_block: verdicts by blocks need recompute_pe_new");
    if (!plan->grp_bits) {
        const size_t gb = sizeof(unsigned long long) * (REVS_STREAM_BLOCK_MAX + 4);   // (+ the call's first iteration, + the handed-over slice)
        hipError_t e = hipMalloc((void **)&plan->grp_bits, 2 * gb);                    // (maxima, then the slices' arrival words)
        if (e == hipSuccess) e = hipMemset(plan->grp_bits, 0, 2 * gb);
        if (e == hipSuccess) e = hipMalloc((void **)&plan->grp_dmax, gb);
        if (e == hipSuccess) e = hipMemset(plan->grp_dmax, 0, gb);
        if (e == hipSuccess && !plan->side) e = hipStreamCreateWithFlags(&plan->side, hipStreamNonBlocking);
        if (e != hipSuccess) {
            revs::set_error("revs_plan_set_stream_block: %s", hipGetErrorString(e));
            plan->block = 0;
            return REVS_ELAUNCH;
        }
    }
    plan->block = block;
    plan->overlap = overlap != 0;
    return REVS_OK;
}

// Everything the run loops would otherwise allocate the first time they need it: the ring of node-sum slices and the
// event pool of the block form (for the block size, stream count and communicator the plan has NOW -- a later
// change is picked up by the loops as before), the folded chain's buffers.  A fresh engine's first run then makes no
// allocation, no synchronising memset and no event between its launches (round 5: ~0.5 ms of a 2.2 ms transient).
extern "C" int revs_plan_prepare(revs_plan_t *plan) {
    REVS_REQUIRE(plan, "revs_plan_prepare: null plan");
    const revs_plan_desc_t &d = plan->d;
    if (d.cand_idx1 && d.stats1_host && plan->tree.n > 0) {
        const int rc = fold_alloc(plan);
        if (rc != REVS_OK) return rc;
    }
    if (plan->block > 1 && d.n_homes > 0) {
        const int nranks = plan->comm ? plan->comm->nranks : 1;
        const size_t stride = (size_t)d.m * d.T + (size_t)REVS_DMAX_SLOTS * nranks;
        const size_t need = (size_t)2 * plan->block * stride;
        hipError_t e = hipSuccess;
        if (plan->ring_cap < need) {
            if (plan->ring) { (void)hipDeviceSynchronize(); (void)hipFree(plan->ring); }
            plan->ring = nullptr;
            plan->ring_cap = 0;
            e = hipMalloc((void **)&plan->ring, sizeof(double) * need);
            if (e == hipSuccess) plan->ring_cap = need;
        }
        if (e == hipSuccess && plan->ring_dirty) {
            e = hipMemset(plan->ring, 0, sizeof(double) * plan->ring_cap);
            if (e == hipSuccess) plan->ring_dirty = false;
        }
        while (e == hipSuccess && plan->events.size() < 2 * 8 + 1) {      // (a burst of eight blocks: 256 iterations)
            hipEvent_t ev;
            e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e == hipSuccess) plan->events.push_back(ev);
        }
        if (e != hipSuccess) {
            revs::set_error("revs_plan_prepare: %s", hipGetErrorString(e));
            return REVS_ELAUNCH;
        }
    }
    return REVS_OK;
}

extern "C" int revs_plan_set_kadd_cold(revs_plan_t *plan, int32_t kadd_cold, int32_t cold_at) {
    REVS_REQUIRE(plan && kadd_cold >= 0 && kadd_cold <= 64 && cold_at >= 0, "revs_plan_set_kadd_cold: kadd_cold=%d (0..64), cold_at=%d", kadd_cold, cold_at);
    plan->kadd_cold = kadd_cold;
    plan->kadd_cold_at = cold_at;
    return REVS_OK;
}

extern "C" int revs_plan_set_fold_redo(revs_plan_t *plan, int32_t steps) {
    REVS_REQUIRE(plan && steps >= 0 && steps <= 8, "revs_plan_set_fold_redo: steps=%d outside 0..8", steps);
    plan->fold_redo = steps;
    return REVS_OK;
}

extern "C" int revs_plan_set_stream_inner(revs_plan_t *plan, int32_t inner) {
    REVS_REQUIRE(plan && inner >= 1 && inner <= REVS_AGENT_MAX_INNER,
                 "revs_plan_set_stream_inner: inner=%d outside 1..%d", inner, REVS_AGENT_MAX_INNER);
    plan->inner = inner;
    return REVS_OK;
}

extern "C" int revs_plan_set_pdhg_dual(revs_plan_t *plan, float *pdhg_dual) {
    REVS_REQUIRE(plan && (pdhg_dual != nullptr) == (plan->d.pdhg_dual != nullptr),
                 "revs_plan_set_pdhg_dual: the plan was created %s carried multipliers",
                 plan && plan->d.pdhg_dual ? "with" : "without");
    plan->d.pdhg_dual = pdhg_dual;
    return REVS_OK;
}

extern "C" int revs_plan_stream_timing(revs_plan_t *plan, int32_t enable) {
    REVS_REQUIRE(plan, "revs_plan_stream_timing: null plan");
    for (hipEvent_t &e : plan->tev)
        if (enable && !e && hipEventCreate(&e) != hipSuccess) {
            revs::set_error("revs_plan_stream_timing: hipEventCreate failed");
            return REVS_ELAUNCH;
        }
    for (hipEvent_t &e : plan->cev)
        if (enable && plan->comm && !e && hipEventCreate(&e) != hipSuccess) {
            revs::set_error("revs_plan_stream_timing: hipEventCreate failed");
            return REVS_ELAUNCH;
        }
    plan->timing = enable ? 1 : 0;
    plan->cev_valid = false;
    return REVS_OK;
}

extern "C" int revs_plan_collective_ms(revs_plan_t *plan, double *collective_ms, double *block_ms, int32_t *iterations) {
    REVS_REQUIRE(plan && collective_ms && block_ms && iterations, "revs_plan_collective_ms: null argument");
    REVS_REQUIRE(plan->cev_valid, "revs_plan_collective_ms: no block's all-reduce has been timed (one GPU, or revs_plan_stream_timing not armed)");
    float a = 0.f, b = 0.f;
    hipError_t e = hipEventElapsedTime(&a, plan->cev[0], plan->cev[1]);
    if (e == hipSuccess) e = hipEventElapsedTime(&b, plan->cev[2], plan->cev[3]);
    if (e != hipSuccess) {
        revs::set_error("revs_plan_collective_ms: %s (synchronise the streams first)", hipGetErrorString(e));
        return REVS_ELAUNCH;
    }
    *collective_ms = (double)a;
    *block_ms = (double)b;
    *iterations = plan->cev_nb;
    return REVS_OK;
}

extern "C" int revs_plan_stream_elapsed_ms(revs_plan_t *plan, double *ms) {
    REVS_REQUIRE(plan && ms, "revs_plan_stream_elapsed_ms: null argument");
    REVS_REQUIRE(plan->timing == 2, "revs_plan_stream_elapsed_ms: no burst since revs_plan_stream_timing");
    float f = 0.f;
    const hipError_t e = hipEventElapsedTime(&f, plan->tev[0], plan->tev[1]);
    if (e != hipSuccess) {
        revs::set_error("revs_plan_stream_elapsed_ms: %s (synchronise the stream first)", hipGetErrorString(e));
        return REVS_ELAUNCH;
    }
    *ms = (double)f;
    plan->timing = 1;
    return REVS_OK;
}

extern "C" int64_t revs_plan_stream_launches(revs_plan_t *plan) {
    return plan ? plan->timed_launches : 0;
}

// two HIP events around the bursts since revs_plan_stream_timing, on the bursts' own stream
static void timing_begin(revs_plan_t *plan, hipStream_t s) {
    if (plan->timing == 1 && hipEventRecord(plan->tev[0], s) == hipSuccess) { plan->timing = 2; plan->timed_launches = 0; }
}
static void timing_end(revs_plan_t *plan, hipStream_t s) {
    if (plan->timing == 2) (void)hipEventRecord(plan->tev[1], s);
}

extern "C" int32_t revs_plan_status_flags(revs_plan_t *plan, int32_t clear) {
    if (!plan || !plan->flags_host) return 0;
    const unsigned int f = *(volatile unsigned int *)plan->flags_host;
    if (clear) *(volatile unsigned int *)plan->flags_host = 0u;
    return (int32_t)f;
}

// Wait for the record of launch `seq`; 0 = kept, 1 = its verdict failed, < 0 = error.
static int stream_wait(revs_plan_t *plan, unsigned int seq, hipStream_t s, double *rmax) {
    const volatile double *r = plan->rec_host + 4 * (seq % revs::kRecRing);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    static const bool trace = getenv("REVS_PLAN_TRACE") != nullptr;
    auto prev = t0;
    while (r[2] != (double)seq) {
        if (trace) {        // the longest the host itself was away from this loop
            const auto now = std::chrono::steady_clock::now();
            plan->t_wait = std::max(plan->t_wait, std::chrono::duration<double, std::micro>(now - prev).count());
            prev = now;
        }
        // (no HIP call in this loop: a hipStreamQuery here was measured to stop the host for
        // milliseconds now and then -- the runtime retires its finished commands inside it --
        // while the queue behind the awaited launch ran dry)
        if ((++spins & 0xFFFFF) == 0) {
            const auto waited = std::chrono::steady_clock::now() - t0;
            if (waited > std::chrono::seconds(2) && hipStreamQuery(s) == hipSuccess && r[2] != (double)seq) {
                revs::set_error("revs_plan_stream_run: stream idle but record %u missing", seq);
                return REVS_ELAUNCH;
            }
            if (waited > std::chrono::seconds(120)) {
                (void)hipStreamSynchronize(s);
                revs::set_error("revs_plan_stream_run: timed out waiting for record %u", seq);
                return REVS_ELAUNCH;
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    *rmax = r[0];
    return r[1] != 0.0 ? 1 : 0;
}

static void stream_rotate(revs_stream_state_t *st, int kept, int64_t n_homes) {
    if (kept <= 0) return;
    revs_stream_state_t r = *st;
    for (int i = 0; i < 3; ++i) { r.p_est[i] = st->p_est[(kept + i) % 3]; r.p[i] = st->p[(kept + i) % 3]; }
    for (int i = 0; i < 2; ++i) { r.p_sch[i] = st->p_sch[(kept + i) % 2]; r.gamma[i] = st->gamma[(kept + i) % 2]; }
    if (st->diff_hist) r.diff_hist = st->diff_hist + (int64_t)kept * n_homes;
    *st = r;
}

// The streaming loop with the verdicts taken by blocks (see include/revs_admm.h).  Iteration k of
// the call has number seq0 + k and consumes the node sums "of iteration k".  Enqueued in one burst:
//   verdict of iteration 0 (the caller's st->p0);
//   per block [k0, k0 + nb):  sweep launches of up to `inner` iterations each, iteration k
//       accumulating the sums of iteration k + 1 (and the partial maxima of its own diff) into
//       ring slice k - k0 | ONE all-reduce of the nb slices (sharded) | verdicts of iterations
//       k0+1 .. k0+nb (the last block: .. max_steps - 1, its last slice is the next call's st->p0;
//       its diff tail is folded into the extra record seq0 + max_steps) | the slices cleared;
// every launch is a no-op once an iteration at or before its own has failed.  With plan->overlap
// the all-reduce and the verdicts of block b go to the plan's second stream while the caller's
// stream runs block b + 1 (two ring halves; block b + 2 waits for block b's verdicts).  The
// blocks are B long, the last B iterations of a call split 3 : 1 so that the all-reduce nobody
// can hide -- the last one -- is a short one.  Then the records are read in order.
// Roll-back without copies: the residences' state lives in FOUR sets of buffers.  A block reads
// its entry set E_b and its launches alternate between the two sets that are neither E_b nor
// E_{b-1}, so the state a block started from survives until the block AFTER it has been enqueued
// -- and that one cannot start before this block's verdicts are in.  A failed iteration j means
// that sweeps behind j ran on an estimate that was not the operator's answer: the sweeps from the
// entry of the block that judged j up to j - 1 (all judged good) are run again from E_b, then sweep
// j itself (outputs to a spare set, carried multipliers in place): bit for bit the memory that the
// loop judging every launch leaves behind a failed verdict.
extern "C" int revs_plan_stream_run_blocks(revs_plan_t *plan, int32_t max_steps, revs_stream_sets_t *st,
                                           double scale, double eps, int32_t *kept_steps, double *rmax_last,
                                           double *dmax_out, void *stream) {
    REVS_REQUIRE(plan && st && kept_steps && rmax_last && max_steps >= 0 && max_steps < revs::kRecRing - 1 &&
                 scale > 0.0 && eps > 0.0, "revs_plan_stream_run_blocks: bad argument (at most %d steps per call)",
                 revs::kRecRing - 2);
    const revs_plan_desc_t &d = plan->d;
    REVS_REQUIRE(plan->tree.n > 0 && d.node_of && plan->block > 1 && d.recompute_pe_new,
                 "revs_plan_stream_run_blocks: needs a tree, node_of, recompute_pe_new and revs_plan_set_stream_block");
    const bool warm = d.mode == REVS_MODE_RELAXED_PDHG && d.pdhg_dual != nullptr;
    for (int i = 0; i < 4; ++i) {
        REVS_REQUIRE(st->p_est[i] && st->p_sch[i] && st->gamma[i] && (!warm || st->pdhg_dual[i]),
                     "revs_plan_stream_run_blocks: null buffer in set %d", i);
        for (int j = 0; j < i; ++j)
            REVS_REQUIRE(st->p_est[i] != st->p_est[j] && st->p_sch[i] != st->p_sch[j] && st->gamma[i] != st->gamma[j] &&
                         (!warm || st->pdhg_dual[i] != st->pdhg_dual[j]),
                         "revs_plan_stream_run_blocks: the four sets must be distinct buffers");
        REVS_REQUIRE(st->p_est_next != st->p_est[i], "revs_plan_stream_run_blocks: p_est_next must not be in a set");
    }
    REVS_REQUIRE(st->p0 && st->p0_out && st->p0 != st->p0_out && st->p_est_next,
                 "revs_plan_stream_run_blocks: p0 / p0_out / p_est_next missing (or p0 == p0_out)");
    REVS_REQUIRE(!warm || st->pdhg_dual[0] == d.pdhg_dual,
                 "revs_plan_stream_run_blocks: set 0 does not hold the plan's carried multipliers (%p, the plan's: %p)",
                 (void *)st->pdhg_dual[0], (void *)d.pdhg_dual);
    hipStream_t s = (hipStream_t)stream;
    *kept_steps = 0;
    *rmax_last = 0.0;
    if (max_steps == 0) return REVS_OK;
    if (plan->stream_seq > 0xFFFF0000u) {                // wrap, once in 4e9 launches: start over
        const revs::StreamCtl ctl0{0u, 0u, 0ull};
        if (hipStreamSynchronize(s) != hipSuccess ||
            hipMemcpy(plan->ctl, &ctl0, sizeof(ctl0), hipMemcpyHostToDevice) != hipSuccess) {
            revs::set_error("revs_plan_stream_run_blocks: resetting the control block failed");
            return REVS_ELAUNCH;
        }
        plan->stream_seq = 0;
    }
    const unsigned int seq0 = plan->stream_seq + 1;
    const int64_t mt = (int64_t)d.m * d.T, nt = d.n_homes * (int64_t)d.T;
    (void)nt;
    const int B = plan->block, K = std::min(plan->inner, revs_agent_max_inner(d.T, d.pdhg.lanes));
    const int nranks = plan->comm ? plan->comm->nranks : 1, rank = plan->comm ? plan->comm->rank : 0;
    const int ntail = REVS_DMAX_SLOTS * nranks;
    const int64_t stride = mt + ntail;                   // doubles per ring slice: node sums, then every rank's partial maxima
    // (the second stream costs a burst ~0.15 ms of host time in event and cross-stream calls: a
    // burst of one block has nothing to hide behind and stays on the caller's stream)
    const bool ov = plan->overlap != 0 && max_steps > B;
    const double vtol = eps * scale;
    auto hip_ok = [&](hipError_t e, const char *what) -> int {
        if (e == hipSuccess) return REVS_OK;
        revs::set_error("revs_plan_stream_run_blocks: %s: %s", what, hipGetErrorString(e));
        return REVS_ELAUNCH;
    };
    {   // the ring: two halves of B slices (one half without the second stream)
        const size_t need = (size_t)2 * B * stride;
        if (plan->ring_cap < need) {
            if (plan->ring) { (void)hipStreamSynchronize(s); (void)hipFree(plan->ring); }
            plan->ring = nullptr;
            plan->ring_cap = 0;
            if (hip_ok(hipMalloc((void **)&plan->ring, sizeof(double) * need), "hipMalloc(ring)") != REVS_OK)
                return REVS_ELAUNCH;
            plan->ring_cap = need;
            plan->ring_dirty = true;
        }
    }
    // block starts: k0[b], b = 0 .. nblocks (k0[nblocks] = max_steps)
    std::vector<int> k0s;
    for (int k = 0; k < max_steps;) {
        k0s.push_back(k);
        const int rem = max_steps - k;
        k += rem > B ? B : (ov && rem >= 8 ? rem - (rem + 3) / 4 : rem);
    }
    const int nblocks = (int)k0s.size();
    k0s.push_back(max_steps);
    auto block_of = [&](int j) {       // the block whose verdicts cover iteration j >= 1: k0 < j <= k0 + nb
        int b = 0;
        while (k0s[b + 1] < j) ++b;
        return b;
    };
    auto ring_of = [&](int b) { return plan->ring + (ov ? (int64_t)(b & 1) * B * stride : 0); };
    // One launch: iterations k .. k + kin - 1 from set `in` to set `out`; their node sums and diff
    // tails to slices (k - k0) .. of `ring` (replay: one scratch region, no tails).
    const int32_t *const wg_order = plan_wg_order(plan);
    auto sweep = [&](int k, int kin, int in, int out, double *slice0, bool replay, float *pe_next,
                     bool y_in_place) -> int {
        revs::StreamExtra sx{};
        sx.ctl = plan->ctl;
        sx.seq = seq0 + (unsigned int)k + 1u;            // (the kernel skips when bad < seq: at or before k)
        sx.base_seq = replay ? sx.seq : seq0;            // (a replayed sweep is never skipped)
        sx.verdict = false;
        sx.flags = plan->flags_dev;
        sx.kin = kin;
        sx.pe_out = st->p_est[out];
        sx.y_out = warm ? (y_in_place ? st->pdhg_dual[in] : st->pdhg_dual[out]) : nullptr;
        sx.slice_stride = stride;
        sx.diff_stride = st->diff_hist ? d.n_homes : 0;
        sx.dmax_out = replay ? nullptr : slice0 + mt + (int64_t)rank * REVS_DMAX_SLOTS;
        sx.wg_order = wg_order;
        return revs::agent_step_stream(
            d.n_homes, d.T, d.cost, d.homes, d.load, st->p_est[in], nullptr, st->p_sch[in], st->gamma[in],
            st->p_sch[out], st->gamma[out],
            st->diff_hist ? st->diff_hist + (int64_t)k * d.n_homes : d.diff, d.dsq, d.status,
            warm ? st->pdhg_dual[in] : nullptr, (float)d.kappa, d.mode, &d.pdhg, d.node_of, slice0, pe_next, sx, s);
    };
    // events of the overlapped form: [2 b] = block b's sweeps are done, [2 b + 1] = its verdicts are
    // in, [2 nblocks] = the side stream has finished this call
    if (ov)
        while ((int)plan->events.size() < 2 * nblocks + 1) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
                revs::set_error("revs_plan_stream_run_blocks: hipEventCreate failed");
                return REVS_ELAUNCH;
            }
            plan->events.push_back(e);
        }
    int rc = REVS_OK, launched = 0, checked = 0, failed_at = -1;
    double rm = 0.0;
    static const bool trace = getenv("REVS_PLAN_TRACE") != nullptr;
    const auto tr0 = std::chrono::steady_clock::now();
    timing_begin(plan, s);
    // The slices a call accumulates into are zero: every verdict launch clears what it has judged.
    // Only a fresh ring, or one a failed call left behind (silenced launches judge nothing), is cleared here.
    if (plan->ring_dirty) {
        rc = hip_ok(hipMemsetAsync(plan->ring, 0, sizeof(double) * plan->ring_cap, s), "hipMemsetAsync(ring)");
        plan->ring_dirty = false;
    }
    hipStream_t q = ov ? plan->side : s;                 // where the collective and the verdicts go
    std::vector<int> entry(nblocks + 1, 0);              // the set a block starts from
    int cur = 0, prev = 1;                               // prev: the entry of the block before (kept intact as well)
    for (int b = 0; b < nblocks && rc == REVS_OK; ++b) {
        const int k0 = k0s[b], nb = k0s[b + 1] - k0;
        double *ring = ring_of(b);
        entry[b] = cur;
        int wk[2], nw = 0;
        for (int i = 0; i < 4; ++i) if (i != cur && i != prev) wk[nw++] = i;
        // (overlapped: this block reuses the ring half of block b - 2, whose verdicts must be in
        // and the half cleared -- they also decide whether this block is a no-op -- and rewrites
        // the set block b - 2 started from)
        if (ov && b >= 2) rc = hip_ok(hipStreamWaitEvent(s, plan->events[2 * (b - 2) + 1], 0), "hipStreamWaitEvent");
        int in = cur, w = 0;
        const bool time_block = plan->timing != 0 && plan->comm && plan->cev[0] && b == 0;      // (a call's first block)
        if (time_block && rc == REVS_OK) rc = hip_ok(hipEventRecord(plan->cev[2], s), "hipEventRecord");
        for (int k = k0; k < k0 + nb && rc == REVS_OK;) {
            const int kin = std::min(K, k0 + nb - k);
            const bool last = (k + kin == max_steps);    // the call's last launch also prepares P_est[k+n+1]
            rc = sweep(k, kin, in, wk[w], ring + (int64_t)(k - k0) * stride, false, last ? st->p_est_next : nullptr, false);
            ++plan->timed_launches;
            in = wk[w];
            w ^= 1;
            k += kin;
            launched += kin;
        }
        if (time_block && rc == REVS_OK) rc = hip_ok(hipEventRecord(plan->cev[3], s), "hipEventRecord");
        prev = cur;
        cur = in;
        // The call's last block has nothing to run beside: without a collective its verdicts go behind its
        // sweeps on the caller's stream -- no hop between streams in front of the launch the host waits for
        // (a burst of one block never leaves the stream) -- once the block before has been judged (the
        // verdict launches share their arrival counter).
        const bool lastb = b + 1 == nblocks;
        const bool on_main = ov && lastb && !plan->comm;
        hipStream_t vq = on_main ? s : q;
        if (on_main) {
            if (b >= 1 && rc == REVS_OK)
                rc = hip_ok(hipStreamWaitEvent(s, plan->events[2 * (b - 1) + 1], 0), "hipStreamWaitEvent");
        } else {
            if (ov && rc == REVS_OK) rc = hip_ok(hipEventRecord(plan->events[2 * b], s), "hipEventRecord");
            if (ov && rc == REVS_OK) rc = hip_ok(hipStreamWaitEvent(q, plan->events[2 * b], 0), "hipStreamWaitEvent");
        }
        if (rc == REVS_OK && plan->comm) {
            if (time_block) rc = hip_ok(hipEventRecord(plan->cev[0], q), "hipEventRecord");
            if (rc == REVS_OK) rc = revs_comm_allreduce_f64(plan->comm, ring, (int64_t)nb * stride, 0, q);
            if (time_block && rc == REVS_OK) {
                rc = hip_ok(hipEventRecord(plan->cev[1], q), "hipEventRecord");
                plan->cev_valid = rc == REVS_OK;
                plan->cev_nb = nb;
            }
        }
        // block 0's launch also judges the call's first iteration (the caller's st->p0: its sweep ran
        // unjudged, like every other sweep of the block); the last block's also hands the call's last
        // slice over to the caller (st->p0_out) and folds its tail into the extra record
        const int judged = (lastb ? nb - 1 : nb) + (b == 0 ? 1 : 0);
        if (rc == REVS_OK)
            rc = revs::stream_block_verdict(plan->ctl, seq0, seq0 + (unsigned int)k0,
                                            seq0 + (unsigned int)k0 + (b == 0 ? 0u : 1u), judged, d.T, plan->tree,
                                            b == 0 ? st->p0 : nullptr, ring, stride, (int32_t)mt, ntail,
                                            lastb ? st->p0_out : nullptr, d.vlo, d.vhi, vtol,
                                            plan->grp_bits, plan->grp_dmax, plan->rec_dev, vq);
        if (ov && rc == REVS_OK) rc = hip_ok(hipEventRecord(plan->events[2 * b + 1], vq), "hipEventRecord");
    }
    entry[nblocks] = cur;
    if (ov) {        // the caller's stream is done when the side stream is (also after an error above)
        int r2 = hip_ok(hipEventRecord(plan->events[2 * nblocks], q), "hipEventRecord");
        if (r2 == REVS_OK) r2 = hip_ok(hipStreamWaitEvent(s, plan->events[2 * nblocks], 0), "hipStreamWaitEvent");
        if (rc == REVS_OK) rc = r2;
    }
    timing_end(plan, s);
    const auto tr1 = std::chrono::steady_clock::now();
    // records 0 .. launched - 1 are the iterations' verdicts; record `launched` carries the last
    // iteration's max diff only
    for (; rc == REVS_OK && checked <= launched && failed_at < 0; ++checked) {
        double r = 0.0;
        const int v = stream_wait(plan, seq0 + (unsigned int)checked, s, &r);
        if (v < 0) rc = v;
        else if (v == 1) failed_at = checked;
        if (checked < launched) rm = r;
        if (v >= 0 && dmax_out && checked >= 1)
            dmax_out[checked - 1] = plan->rec_host[4 * ((seq0 + (unsigned int)checked) % revs::kRecRing) + 3];
    }
    if (trace) {
        fprintf(stderr, "[revs_plan_stream_run_blocks] %d blocks of at most %d, %d iterations per launch%s: %d iterations "
                "enqueued in %.1f us, records read %.1f us later (host away from the wait loop for at most %.1f us), "
                "failed at %d\n", nblocks, B, K, ov ? ", overlapped" : "", launched,
                std::chrono::duration<double, std::micro>(tr1 - tr0).count(),
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr1).count(),
                plan->t_wait, failed_at);
        plan->t_wait = 0.0;
    }
    plan->stream_seq = seq0 + (unsigned int)std::max(launched, 1);
    *rmax_last = rm;
    if (failed_at >= 0 || rc != REVS_OK) {
        if (ov) (void)hipStreamSynchronize(plan->side);
        (void)hipStreamSynchronize(s);
        plan->ring_dirty = true;
        // The sticky status word has collected bits from sweeps that are now undone (they ran on an
        // estimate that was not the operator's answer): "a PDHG residence stopped at its cap" is a
        // statement about such a sweep's problem, not about the trajectory -- dropped here and set
        // again by the replay below for the sweeps that stand.  ("No solution" does not depend on
        // the estimate: kept.)
        if (plan->flags_host) *(volatile unsigned int *)plan->flags_host &= ~2u;
    }
    int kept = rc != REVS_OK ? 0 : (failed_at >= 0 ? failed_at : launched);
    int fin = cur;                                       // the set that holds the state at return
    if (rc == REVS_OK && failed_at >= 0) {
        // Block bf judged it.  If it is the first iteration of block bf + 1, that block's entry set IS
        // the state wanted (a block never writes the set it started from).  Otherwise go back to bf's
        // own entry and run the good sweeps k0 .. failed_at - 1 again, alternating between two sets
        // that are not the entry.
        const int bf = failed_at > 0 ? block_of(failed_at) : -1;
        int k = failed_at, in = entry[0];
        if (bf >= 0) {
            if (failed_at == k0s[bf + 1]) in = entry[bf + 1];
            else { k = k0s[bf]; in = entry[bf]; }
        }
        int wk[2], nw = 0;
        for (int i = 0; i < 4 && nw < 2; ++i) if (i != in) wk[nw++] = i;
        int w = 0;
        while (k < failed_at && rc == REVS_OK) {
            const int kin = std::min(K, failed_at - k);
            rc = sweep(k, kin, in, wk[w], plan->ring, true, nullptr, false);
            in = wk[w];
            w ^= 1;
            k += kin;
        }
        // ... and the failed iteration's own sweep, as the loop that judges every launch runs it:
        // outputs to a spare set, the carried multipliers updated in place
        int spare = 0;
        while (spare == in) ++spare;
        if (rc == REVS_OK) rc = sweep(failed_at, 1, in, spare, plan->ring, true, st->p_est_next, true);
        fin = in;
        if (rc != REVS_OK || hipStreamSynchronize(s) != hipSuccess) {
            if (rc == REVS_OK) revs::set_error("revs_plan_stream_run_blocks: replaying the block failed");
            rc = REVS_ELAUNCH;
            kept = 0;
        }
    }
    *kept_steps = kept;
    if (rc == REVS_OK && fin != 0) {                     // roles: set 0 = the state at return
        std::swap(st->p_est[0], st->p_est[fin]);
        std::swap(st->p_sch[0], st->p_sch[fin]);
        std::swap(st->gamma[0], st->gamma[fin]);
        std::swap(st->pdhg_dual[0], st->pdhg_dual[fin]);
    }
    if (rc == REVS_OK && warm) plan->d.pdhg_dual = st->pdhg_dual[0];
    if (st->diff_hist) st->diff_hist += (int64_t)kept * d.n_homes;
    return rc;
}

extern "C" int revs_plan_stream_run(revs_plan_t *plan, int32_t max_steps, revs_stream_state_t *st,
                                    double scale, double eps, int32_t *kept_steps,
                                    double *rmax_last, void *stream) {
    REVS_REQUIRE(plan && st && kept_steps && rmax_last && max_steps >= 0 && max_steps < revs::kRecRing &&
                 scale > 0.0 && eps > 0.0, "revs_plan_stream_run: bad argument (at most %d steps per call)",
                 revs::kRecRing - 1);
    const revs_plan_desc_t &d = plan->d;
    REVS_REQUIRE(plan->tree.n > 0 && plan->tree.n <= REVS_TREE_SWEEP_MAX && d.node_of,
                 "revs_plan_stream_run: the plan has no tree / node_of, or a tree of more than %d nodes (those are "
                 "judged by blocks: revs_plan_stream_run_blocks)", REVS_TREE_SWEEP_MAX);
    for (int i = 0; i < 3; ++i)
        REVS_REQUIRE(st->p_est[i] && st->p[i] && (i == 2 || (st->p_sch[i] && st->gamma[i])),
                     "revs_plan_stream_run: null buffer");
    REVS_REQUIRE(st->p[0] != st->p[1] && st->p[1] != st->p[2] && st->p[0] != st->p[2] &&
                 st->p_est[0] != st->p_est[1] && st->p_est[1] != st->p_est[2] && st->p_est[0] != st->p_est[2] &&
                 st->p_sch[0] != st->p_sch[1] && st->gamma[0] != st->gamma[1],
                 "revs_plan_stream_run: buffers must be distinct");
    hipStream_t s = (hipStream_t)stream;
    *kept_steps = 0;
    *rmax_last = 0.0;
    if (max_steps == 0) return REVS_OK;
    // (sequence numbers only grow: what an earlier call left in the control word is below this
    // call's first number and ignored by the kernels -- nothing to re-arm, no copy on the stream)
    if (plan->stream_seq > 0xFFFF0000u) {                // wrap, once in 4e9 launches: start over
        const revs::StreamCtl ctl0{0u, 0u, 0ull};
        if (hipStreamSynchronize(s) != hipSuccess ||
            hipMemcpy(plan->ctl, &ctl0, sizeof(ctl0), hipMemcpyHostToDevice) != hipSuccess) {
            revs::set_error("revs_plan_stream_run: resetting the control block failed");
            return REVS_ELAUNCH;
        }
        plan->stream_seq = 0;
    }
    const unsigned int seq0 = plan->stream_seq + 1;
    const int64_t mt = (int64_t)d.m * d.T;
    REVS_REQUIRE(plan->block <= 1, "revs_plan_stream_run: verdicts by blocks go through revs_plan_stream_run_blocks");
    auto launch = [&](int k) -> int {               // step k of this call (roles by rotation)
        revs::StreamExtra sx;
        sx.ctl = plan->ctl;
        sx.seq = seq0 + (unsigned int)k;
        sx.base_seq = seq0;
        sx.verdict = true;
        sx.tree = plan->tree;
        sx.p_in = st->p[k % 3];
        sx.p_zero = st->p[(k + 2) % 3];
        sx.vlo = d.vlo; sx.vhi = d.vhi; sx.vtol = eps * scale;
        sx.rec = plan->rec_dev + 4 * (sx.seq % revs::kRecRing);
        sx.flags = plan->flags_dev;
        sx.m = d.m;
        double *p_next = st->p[(k + 1) % 3];
        int rc = revs::agent_step_stream(
            d.n_homes, d.T, d.cost, d.homes, d.load, st->p_est[k % 3],
            d.recompute_pe_new ? nullptr : st->p_est[(k + 1) % 3], st->p_sch[k % 2], st->gamma[k % 2],
            st->p_sch[(k + 1) % 2], st->gamma[(k + 1) % 2],
            st->diff_hist ? st->diff_hist + (int64_t)k * d.n_homes : d.diff, d.dsq, d.status, d.pdhg_dual,
            (float)d.kappa, d.mode, &d.pdhg, d.node_of, p_next, st->p_est[(k + 2) % 3], sx, stream);
        if (rc != REVS_OK) return rc;
        if (plan->comm) rc = revs_comm_allreduce_f64(plan->comm, p_next, mt, 0, stream);
        return rc;
    };
    // All max_steps launches (and, sharded, their collectives) are enqueued in ONE burst, then
    // the records are read in order.  No decision is taken between launches -- a failed verdict
    // silences the launches behind it on the device -- so every rank of a sharded run issues the
    // same collectives whatever its timing, and the host never pauses between submissions (a
    // launch submitted after a pause was measured to start late: ~6 us always, milliseconds now
    // and then, whatever the queue holds).  The caller bounds max_steps by how many silenced
    // launches it is willing to waste behind a failure (AdmmEngine._stream_run).
    int launched = 0, checked = 0, failed_at = -1, rc = REVS_OK;
    double rm = 0.0;
    static const bool trace = getenv("REVS_PLAN_TRACE") != nullptr;
    const auto tr0 = std::chrono::steady_clock::now();
    auto tr1 = tr0;
    double slowest = 0.0;
    timing_begin(plan, s);
    for (; launched < max_steps; ++launched) {
        const auto a = trace ? std::chrono::steady_clock::now() : tr0;
        if ((rc = launch(launched)) != REVS_OK) goto out;
        ++plan->timed_launches;
        if (trace)
            slowest = std::max(slowest, std::chrono::duration<double, std::micro>(
                                            std::chrono::steady_clock::now() - a).count());
    }
    timing_end(plan, s);
    tr1 = std::chrono::steady_clock::now();
    for (; checked < launched && failed_at < 0; ++checked) {
        const int v = stream_wait(plan, seq0 + (unsigned int)checked, s, &rm);
        if (v < 0) { rc = v; goto out; }
        if (v == 1) failed_at = checked;
    }
    if (trace) {
        const auto tr2 = std::chrono::steady_clock::now();
        fprintf(stderr, "[revs_plan_stream_run] %d launches in %.1f us (slowest %.1f us), records read "
                "%.1f us later (host away from the wait loop for at most %.1f us), %d kept\n", launched,
                std::chrono::duration<double, std::micro>(tr1 - tr0).count(), slowest,
                std::chrono::duration<double, std::micro>(tr2 - tr1).count(), plan->t_wait,
                failed_at >= 0 ? failed_at : launched);
        plan->t_wait = 0.0;
    }
out:
    plan->stream_seq = seq0 + (unsigned int)std::max(launched, 1) - 1;
    *rmax_last = rm;
    const int kept = rc != REVS_OK ? 0 : (failed_at >= 0 ? failed_at : launched);
    if (failed_at >= 0 || rc != REVS_OK)
        (void)hipStreamSynchronize(s);                   // the launches behind the failed one are no-ops
    *kept_steps = kept;
    stream_rotate(st, kept, d.n_homes);   // the roles, by the kept steps
    return rc;
}
