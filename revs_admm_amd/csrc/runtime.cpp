// Error reporting and version string of librevs_admm.so.
#include "common.h"
#include <stdarg.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>

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
struct revs_plan {
    revs_plan_desc_t d;
    hipEvent_t ev;
    double seq;
    uint32_t *counters;     // device, one per 32-row tile: K-split workgroups of R p done
    double t_launch = 0.0, t_wait = 0.0;   // host time in launches / waiting (REVS_PLAN_TRACE)
};

// Host-side acceptance test of a chained Newton iteration (engine.py: _chain_launch): the
// checks AdmmEngine._operator_solve_newton would make on the two evaluations' stats, for the
// one outcome that needs no further launch.  See include/revs_admm.h.
extern "C" int revs_newton_chain_accept(int32_t T, const double *s0, const double *s1, double scale,
                                        double eps, int32_t amax, int32_t kadd, int32_t chain_few,
                                        int32_t *nsup_sum, int32_t *nsup_max) {
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
    if ((ns_max + kadd <= 48) != (chain_few != 0)) return 0;
    double rmax1 = 0.0, sum = 0.0, mx = 0.0;
    for (int t = 0; t < T; ++t) {
        const double *a = s0 + 8 * t, *b = s1 + 8 * t;
        const double D = a[1];
        if (a[0] / scale > eps &&                        // pending slot: Armijo on the full step
            !(b[1] >= D + 1e-4 * b[4] - 1e-13 * (D < 0 ? -D : D)))
            return 0;
        if (b[2] > amax) return 0;
        const double r = b[0] / scale;
        rmax1 = r > rmax1 ? r : rmax1;
        sum += b[2];
        mx = b[2] > mx ? b[2] : mx;
    }
    if (!(rmax1 <= eps)) return 0;                       // needs another iteration
    *nsup_sum = (int32_t)sum;
    *nsup_max = (int32_t)mx;
    return 1;
}

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
    return p;
}

extern "C" void revs_plan_destroy(revs_plan_t *plan) {
    if (!plan) return;
    (void)hipEventDestroy(plan->ev);
    (void)hipFree(plan->counters);
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
    const int sel_nblk = (d.T <= 32 && nb32 <= 256) ? nb32 : 0;
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
    auto rows = [&](const double *yy, int uy, int k) -> int {
        return revs_op_dual_evaluate(2 | 4, d.m, d.T, d.node_ptr, p_est, p_sch, gamma, d.R, d.Rt, yy, uy,
                                     d.kappa, d.vlo, d.vhi, d.kadd, d.ksplit, d.d_slabs, d.v_slabs,
                                     d.pnq, p_est_new, d.vfull, d.viol, d.partial, ci[k], cc[k], cv[k],
                                     st[k], 0.0, plan->counters, stream);
    };
    int rc;
    if ((rc = home_pass(y, use_y, sup0)) != REVS_OK) return rc;
    if ((rc = rows(y, use_y, 0)) != REVS_OK) return rc;
    rc = revs_op_dual_select_model_step(d.m, d.T, d.partial, sel_nblk, y, d.vlo, d.vhi, d.kadd, d.vfull,
                                        d.viol, ci[0], cc[0], cv[0], st[0], 0.0, d.R,
                                        d.pnq + (int64_t)d.m * d.T, d.kappa, d.delta, d.max_pivots,
                                        d.k_full, d.yhat, d.info, scale, d.eps, y_trial, st[1] + 4,
                                        stream);
    if (rc != REVS_OK) return rc;
    if ((rc = home_pass(y_trial, 1, chain_few ? 0 : -1)) != REVS_OK) return rc;
    if ((rc = rows(y_trial, 1, 1)) != REVS_OK) return rc;
    if (ev_mid) (void)hipEventRecord((hipEvent_t)ev_mid, s);
    const double seq = -(plan->seq += 1.0);          // (negative: not a spec-step tag)
    rc = revs_agent_step_select(d.n_homes, d.T, d.cost, d.homes, d.load, p_est, p_est_new, p_sch,
                                gamma, p_sch_out, gamma_out, s_out, c_out, d.diff, d.dsq, d.status,
                                d.pdhg_dual, (float)d.kappa, d.mode, &d.pdhg, d.m, d.partial, y_trial,
                                d.vlo, d.vhi, d.kadd, d.vfull, d.viol, ci[1], cc[1], cv[1], st[1], seq,
                                nullptr, nullptr, nullptr, sel_nblk, stream);
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
        st->sup0 = (nsum > 0 && nmax + plan->d.kadd <= 48) ? 1 : -1;
        std::swap(st->p_sch, st->p_sch_alt);
        std::swap(st->gamma, st->gamma_alt);
        std::swap(st->p_est, st->p_est_new);
        ++*kept_steps;
    }
    return REVS_OK;
}
