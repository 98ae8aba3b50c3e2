/*
 * revs_admm.h -- C ABI of librevs_admm.so, the MI355X (gfx950) engine behind the
 * distributed ("ADMM") path of REVS, rounak-meyur/revs-admm.
 *
 * Boundary: everything `lpsolver.solve_ADMM` (reference lpsolver.py:242-290) does
 * per iteration -- the operator ("Utility") QP, the per-residence ("Home")
 * problem for every home, the dual update and the `diff` residual -- on device
 * buffers the caller owns.  Plain pointers and sizes only; `stream` is a
 * hipStream_t passed as void* (NULL = the null stream).  All pointers are DEVICE
 * pointers unless the name ends in `_host`.  Every entry point returns 0 on
 * success and a negative REVS_E* code otherwise; revs_last_error() gives the text.
 * Nothing here allocates, frees or synchronises: every entry point only enqueues
 * kernels on `stream` and is safe under hipGraph capture -- except the revs_plan_*
 * functions at the end (they own one hipEvent and revs_plan_spec_step waits on it).
 *
 * Layout in HBM: home-major, slot-contiguous.  A "profile" is float[n_homes][T].
 */
#ifndef REVS_ADMM_H
#define REVS_ADMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REVS_OK          0
#define REVS_EINVAL     -1   /* bad argument (null pointer, T out of range, ...) */
#define REVS_ELAUNCH    -2   /* hip launch / runtime error */
#define REVS_ENOTCONV   -3   /* operator solve hit max_iter before eps */

/* Largest T the agent kernels are instantiated for. */
#define REVS_MAX_T 192
/* consecutive ADMM iterations one launch of the residence sweep can carry in registers
 * (streaming steady state, multipliers zero; revs_plan_set_stream_inner); a launch's LDS holds one
 * set of node-sum accumulators per iteration, so the library uses at most
 * revs_agent_max_inner(T, lanes) <= REVS_AGENT_MAX_INNER of them: 32 up to T = 24 (16 with the wide lane
 * shapes), 16 up to 96, 8 up to 192, 4 beyond */
#define REVS_AGENT_MAX_INNER 32
/* the sweep leaves the largest diff of an iteration (the convergence measure, lpsolver.py:284) as
 * REVS_DMAX_SLOTS partial maxima, workgroup b into slot b % REVS_DMAX_SLOTS */
#define REVS_DMAX_SLOTS 64
/* the radial feeder as a tree (documented at revs_tree_voltage below) */
#define REVS_TREE_MAX 16384
/* ... of which a launch that judges its own rows (revs_plan_stream_run with block <= 1) and the chained
 * iteration's fused launches (rows + selection + model + step in one workgroup of 256 threads, 8 positions each:
 * revs_plan_chain_step, revs_plan_chain_fold_run) hold this many.  The Newton evaluations' row launches
 * (revs_plan_newton_solve, revs_op_dual_rows_tree) take every tree up to REVS_TREE_MAX since round 4. */
#define REVS_TREE_SWEEP_MAX 2048
/* rows (constraint nodes) the folded chain's operator launch holds: a slot's multipliers, voltages and
 * violations are staged in one workgroup's LDS (revs_plan_chain_fold_run) */
#define REVS_CHAIN_FOLD_MAX_M 2048
/* candidate rows per slot in the dual Newton model of the operator QP (the plan's candidate buffers: T x this) */
#define REVS_DUAL_AMAX 128
/* While no slot's list (multipliers + the rows an evaluation may admit) is longer than this, an evaluation forms the
 * shifts R^T y / kappa straight from the listed rows of R; beyond it by the dense f64 product. */
#ifndef REVS_DUAL_FEW
#define REVS_DUAL_FEW 48
#endif
typedef struct {
    int32_t n;
    const uint64_t *pack;
    const double *w;
} revs_tree_t;

/* ---- per-residence data -------------------------------------------------
 * One record per home; replaces homes[h]["EV"] of the reference
 * (extract.py:122-131, consumed by Home.add_EV, lpsolver.py:68-110).
 * nmin/nmax: number of full-rate slots the SOC rows allow, i.e. the integers n
 * with initial + n*rating/capacity in [0.9, 1.0] (lpsolver.py:101-109).  The
 * host computes them in double so that the device never rounds a borderline
 * case differently from the model (revs_admm_amd/lpsolver.py: pack_homes).   */
typedef struct {
    int32_t ev;        /* 0: no EV (p = 0, s = 0: lpsolver.py:70-79) */
    int32_t start;     /* first slot charging is allowed  (lpsolver.py:97) */
    int32_t end;       /* one past the last allowed slot                  */
    int32_t nmin;
    int32_t nmax;
    float   rating;    /* kW   */
    float   capacity;  /* kWh  */
    float   initial;   /* SOC at slot 0 */
} revs_home_t;

/* how the Home problem treats the charger */
#define REVS_MODE_BINARY        0  /* p_t in {0, rating}: the reference MIQP, exact     */
#define REVS_MODE_RELAXED_PDHG  1  /* 0 <= p_t <= rating, batched PDHG (north star)     */
#define REVS_MODE_RELAXED_EXACT 2  /* same QP, closed-form multiplier search            */

typedef struct {
    int32_t max_iter;     /* PDHG iteration cap (multiple of check)           */
    int32_t check;        /* convergence test every `check` iterations        */
    float   tol;          /* stop when max(|dx|, |dy|/sigma) <= tol; 0 (default) = automatic: 1e-4
                             where the polish below finishes the solve, 1e-6 with polish = 0 or
                             full_rows; < 0: never before max_iter */
    float   tau_scale;    /* tau   = tau_scale   / ||K||   (0 = automatic)    */
    float   sigma_scale;  /* sigma = sigma_scale / ||K||   (0 = automatic)    */
    int32_t full_rows;    /* 0 (default): presolved -- with p >= 0 the SOC is
                             nondecreasing, so only the terminal rows s_T in [0.9, 1]
                             of lpsolver.py:101-109 can bind; K is that single row and
                             the dual one scalar per home (automatic scales 0.5 / 2).
                             1: keep all T SOC rows (K = prefix sum; scales 0.25 / 4) */
    int32_t polish;       /* presolved form only, bits.  1 (default): after PDHG has stopped, semismooth Newton
                             steps on the terminal row's multiplier with the schedule in closed form
                             x(mu) = clip(-b - delta mu, 0, w) -- the exact optimum of the piece PDHG identified
                             (the KKT conditions of lpsolver.py:83-128's relaxation to float rounding).
                             2: the same steps BEFORE PDHG, from the carried multiplier (ydual; zero without):
                             a residence they settle (at most 6 steps) does not enter PDHG, a wavefront of such
                             residences skips the loop -- in the closed loop that is nearly every solve after the
                             first iterations (status >> 8, the PDHG passes, reads 0 for them); the rest go
                             through PDHG and bit 0's steps as before.  Same optimum either way. */
    int32_t lanes;        /* lanes of a wavefront that share one residence's T slots (every mode, not only PDHG):
                             0 (default) = by T alone (8 lanes x 3 slots at T = 24: the fewest instructions per
                             residence); 16 or 32 with T <= 32 = 2 / 1 slot(s) per lane -- a third of the per-slot work
                             in every wavefront's instruction chain and 2 - 4 x the wavefronts: the shape for a GPU
                             that holds few residences (BASELINE config 2 over eight GPUs: 12 500 each, 391 workgroups
                             of the default shape on 256 compute units).  Other values, or T > 32: as 0. */
    int32_t keys64;       /* on/off chargers (REVS_MODE_BINARY): 1 = the slots' switching costs delta_t are formed and
                             ranked in double, in the float64 restatement's order of operations -- the decision is then
                             the oracle's for every residence on identical inputs.  0 (default): in float, by the same
                             operations that update the state, so that the closed loop's exactly tied optima
                             (lpsolver.py:92-110 on flat tariff blocks x repeated loads) stay tied from one ADMM
                             iteration to the next, as they do in the reference's double arithmetic. */
} revs_pdhg_t;

const char *revs_version(void);
const char *revs_last_error(void);

/* Device-side address of PINNED host memory (hipHostMalloc, torch pin_memory), so that a
 * kernel can write a few result words where the host reads them (revs_op_dual_select's
 * stats).  Fails with REVS_EINVAL when host_ptr is not mapped pinned memory. */
int revs_host_device_ptr(void *host_ptr, void **dev_ptr_out);

/* Defaults used when `pdhg` is NULL: 4000, 4, automatic tolerance and scales, presolved rows, polish */
void revs_pdhg_defaults(revs_pdhg_t *out_host);

/* Number of chunk records (3 doubles each) revs_residual_finalize uses for n_homes; size its
 * `scratch` as double[3 * that]. */
int32_t revs_residual_num_chunks(int64_t n_homes);

/* One ADMM iteration of the residence side for ALL homes:
 *   Home(cost, homes[h], P_est[k][h], P_sch[k][h], G[k][h]).solve()  lpsolver.py:273-277
 *   check = P_est[k+1][h] - P_sch[k+1][h]                            lpsolver.py:280-281
 *   G[k+1][h] = G[k][h] + (kappa/2) check                            lpsolver.py:282-283
 *   diff[k+1][h] = |check|_2 / T                                     lpsolver.py:284
 * fused in one kernel.
 *   cost        float[T]          tariff
 *   load        float[n][T]       homes[h]["LOAD"]
 *   p_est_old   float[n][T]       P_est[k]     (read)
 *   p_est_new   float[n][T]       P_est[k+1]   (read; may alias p_est_old)
 *   p_sch       float[n][T]       in: P_sch[k]   out: P_sch[k+1] = g_opt
 *   gamma       float[n][T]       in: G[k]       out: G[k+1]
 *   s_out       float[n][T]       p_opt  (reference's S)   or NULL to skip
 *   c_out       float[n][T+1]     s_opt  (reference's C)   or NULL to skip
 *   diff        float[n]          diff[k+1]
 *   dsq         float[n]          per home sum_t (P_sch[k+1] - P_sch[k])^2 (the dual residual's
 *                                 terms; folded with diff by revs_residual_finalize on request --
 *                                 the sweep itself does no reduction across homes)
 *   status      int32[n]          0 ok, 1 infeasible ("No solution found", lpsolver.py:153-155),
 *                                 for PDHG: iterations used in bits 8.. ; or NULL
 *   pdhg_dual   float[n] (float[n][T] with full_rows)   REVS_MODE_RELAXED_PDHG only, or NULL:
 *                                 the PDHG multipliers of the SOC rows, kept between ADMM
 *                                 iterations.  When given, PDHG starts from them and from the
 *                                 previous schedule (P_sch[k] - LOAD) instead of from zero;
 *                                 zero-initialise it.
 */
int revs_agent_step(int64_t n_homes, int32_t T,
                    const float *cost, const revs_home_t *homes, const float *load,
                    const float *p_est_old, const float *p_est_new,
                    float *p_sch, float *gamma,
                    float *s_out, float *c_out, float *diff,
                    float *dsq, int32_t *status, float *pdhg_dual,
                    float kappa, int32_t mode, const revs_pdhg_t *pdhg_host,
                    void *stream);

/* The same iteration with P_sch[k+1] and G[k+1] written to separate buffers (p_sch and
 * gamma are only read), so a caller may launch it speculatively -- e.g. before it knows
 * whether the operator's first answer will stand -- and discard the result.
 * revs_agent_step is this call with p_sch_out = p_sch, gamma_out = gamma. */
int revs_agent_step_out(int64_t n_homes, int32_t T,
                        const float *cost, const revs_home_t *homes, const float *load,
                        const float *p_est_old, const float *p_est_new,
                        const float *p_sch, const float *gamma,
                        float *p_sch_out, float *gamma_out,
                        float *s_out, float *c_out, float *diff,
                        float *dsq, int32_t *status, float *pdhg_dual,
                        float kappa, int32_t mode, const revs_pdhg_t *pdhg_host,
                        void *stream);

/* revs_agent_step_out with the operator's candidate selection (the second kernel of
 * revs_op_dual_select; arguments as there, T slots) running as the first T workgroups of the
 * same launch: independent of the sweep, it overlaps it and its stats -- the operator's
 * verdict -- reach the host while the sweep is still running.  The caller must have run
 * revs_op_dual_evaluate with phase bits 2|4 (rows done, selection deferred) before.
 * With p_next != NULL the sweep also performs the home pass of the NEXT evaluation for
 * multipliers y = 0 on the state it has just produced (revs_op_dual_eval with dsl = NULL):
 * p_est_next float[n][T] = max(g0', 0), g0' = (P_est[k+1] + P_sch[k+1])/2 - G[k+1]/kappa, and
 * p_next double[m][T] += node sums of it (ZERO on entry: revs_op_dual_rows can clear it);
 * node_of int32[n] = node of every residence (residences sorted by node).
 * p_est_new may be NULL when the operator's multipliers are all zero: its answer is then
 * max(g0, 0), g0 = (P_est[k] + P_sch[k])/2 - G[k]/kappa, a function of three profiles the
 * sweep reads anyway, and is recomputed (same arithmetic, same bits) instead of loaded --
 * one input stream less where the sweep is bandwidth-bound.
 * sel_nblk: number of partial blocks the selection folds (0 = revs_op_dual_blocks(m), the
 * count revs_op_dual_select / _rows write; (m + 31) / 32 after revs_op_dual_product_rows). */
int revs_agent_step_select(int64_t n_homes, int32_t T,
                           const float *cost, const revs_home_t *homes, const float *load,
                           const float *p_est_old, const float *p_est_new,
                           const float *p_sch, const float *gamma,
                           float *p_sch_out, float *gamma_out,
                           float *s_out, float *c_out, float *diff,
                           float *dsq, int32_t *status, float *pdhg_dual,
                           float kappa, int32_t mode, const revs_pdhg_t *pdhg_host,
                           int32_t m, const double *sel_partial, const double *y, double vlo,
                           double vhi, int32_t kadd, const double *vfull, const double *viol,
                           int64_t *cand_idx, int32_t *cand_cnt, double *cand_val,
                           double *stats, double seq,
                           const int32_t *node_of, double *p_next, float *p_est_next,
                           int32_t sel_nblk, void *stream);

/* The global ADMM residuals from the per-home terms of the last sweep (two small kernels,
 * double accumulation in a fixed order: bitwise reproducible):
 *   out[0] = |P_est[k+1] - P_sch[k+1]|_2 = sqrt(sum (T diff)^2)   (primal residual)
 *   out[1] = kappa * |P_sch[k+1] - P_sch[k]|_2 = kappa sqrt(sum dsq)   (dual residual)
 *   out[2] = max_h diff[k+1][h]   -- the reference's convergence measure (lpsolver.py:284)
 *   out[3] = 1.0f if out[2] <= eps else 0.0f       (convergence flag, stays on device)
 * out: float[4] on the device; scratch: double[3 * revs_residual_num_chunks(n_homes)].   */
int revs_residual_finalize(const float *diff, const float *dsq, int64_t n_homes, int32_t T,
                           float kappa, float eps, double *scratch, float *out, void *stream);

/* OR of bits 0-2 of every residence's status word (1: no solution -- lpsolver.py:153-155's message --, 2: PDHG stopped at its
 * cap, 4: the KKT polish did not settle) into *out, a word of PINNED host memory by its device-side address
 * (revs_host_device_ptr) that the caller has cleared: one small launch and a stream synchronise instead of a read-back of
 * the status array.  *out is complete when the stream has passed the launch. */
int revs_status_or(int64_t n_homes, const int32_t *status, uint32_t *out, void *stream);

/* Individual mode, lpsolver.py:430-460 (solve_residence): min 0.01 tariff.g +
 * 0.99 (1 - s_T), binary charger, SOC box, no s_T >= 0.9 row.
 *   p_out float[n][T], soc_out float[n][T+1], g_out float[n][T]                 */
int revs_residence_solve(int64_t n_homes, int32_t T,
                         const float *tariff, const revs_home_t *homes, const float *load,
                         float *p_out, float *soc_out, float *g_out, void *stream);

/* ---- operator ("Utility") side: see also revs_admm_ops.h -----------------------------
 * The drop-in binds the calls of THIS header: the home sweep (revs_agent_step*), the individual mode
 * (revs_residence_solve), the voltage check below, and the plan calls further down, behind which the whole
 * operator QP runs natively (revs_plan_newton_solve, revs_plan_chain_fold_run, revs_plan_stream_run_blocks).
 * The operator's building blocks -- the launches those plan calls are made of, which the Python driver's
 * general paths (the ADMM forms of the operator QP, the chain issued in phases) and the kernel-level tests
 * issue one by one -- are declared in include/revs_admm_ops.h. */

/* V = R P : the operator's LinDistFlow voltage-sensitivity check -- `R_res @ g[:,t]`
 * of lpsolver.py:191-193 and `R@P` of drawing.py:75 -- for all slots at once.
 * Rt = R transposed, float[m][m] row-major (R itself: it is symmetric);
 * P float[m][T] node injections; V float[m][T]. */
int revs_voltage_f32(int32_t m, int32_t T, const float *Rt, const float *P, float *V,
                     void *stream);

/* `kin` (1..REVS_AGENT_MAX_INNER) consecutive ADMM iterations of every residence in ONE launch,
 * for the regime in which the operator's multipliers are zero: its answer of iteration g + 1,
 * P_est[g+1] = max((P_est[g] + P_sch[g])/2 - G[g]/kappa, 0) (lpsolver.py:196-207 with slack rows),
 * is a function of the residence's own state, so the recurrence of lpsolver.py:254-287 touches no
 * other residence -- only the voltage verdict does, and it is taken afterwards from the node sums
 * this launch leaves (the streaming loop judges them by blocks and rolls back, see
 * revs_plan_set_stream_block).  The state (P_est[g], P_sch[g], G[g], carried PDHG multipliers) is
 * read once, kept in registers for kin iterations and written once: HBM bytes per iteration / kin.
 * Bit for bit what kin launches of revs_agent_step_select's folded form compute.
 *   p_est / p_sch / gamma      state at g (float[n][T]);  *_out: state at g + kin (other arrays)
 *   p_est_next                 NULL, or P_est[g + kin + 1] (the estimate the next iteration consumes)
 *   diff                       row i (stride diff_stride floats; 0: one row) = diff of iteration g + i
 *   pdhg_dual / pdhg_dual_out  carried multipliers in / out (may be the same array; PDHG only)
 *   p_next                     double, slice i at p_next + i * slice_stride: += node sums of
 *                              P_est[g + i + 2] (zero on entry); node_of: node of every residence
 *   dmax_out                   NULL, or (same stride) REVS_DMAX_SLOTS doubles per iteration: atomic max
 *                              (on the bit pattern of a non-negative double) of the residences' diff
 *                              of iteration g + i; the maximum over the slots is max_h diff[h] */
int32_t revs_agent_max_inner(int32_t T, int32_t lanes);       /* lanes: revs_pdhg_t::lanes */
int revs_agent_step_multi(int64_t n_homes, int32_t T, const float *cost, const revs_home_t *homes,
                          const float *load, const float *p_est, const float *p_sch, const float *gamma,
                          float *p_est_out, float *p_sch_out, float *gamma_out, float *p_est_next,
                          float *diff, int64_t diff_stride, float *dsq, int32_t *status,
                          float *pdhg_dual, float *pdhg_dual_out, float kappa, int32_t mode,
                          const revs_pdhg_t *pdhg_host, const int32_t *node_of, double *p_next,
                          int64_t slice_stride, double *dmax_out, int32_t kin, void *stream);

/* ---- steady-state ADMM iteration as one host call ---------------------------------
 * The driver's loop of lpsolver.py:254-287 for the case "the operator's multipliers are
 * expected to stand" (revs_admm_amd/engine.py: AdmmEngine.step): enqueue one evaluation of
 * the operator's dual (revs_op_dual_evaluate, phase 3) and, right behind it, the home sweep
 * with P_sch[k+1], G[k+1] going to spare buffers (revs_agent_step_out); then wait for the
 * evaluation's stats only and report the largest row residual.  The caller keeps the
 * sweep's output iff that residual is within tolerance, else finishes the Newton solve and
 * runs the sweep again.  The plan holds the pointers that do not change between iterations;
 * the ping-pong buffers are passed per call. */
typedef struct {
    int64_t n_homes; int32_t m; int32_t T;
    const int64_t *node_ptr;
    const double *R; const double *Rt;
    double kappa, vlo, vhi;
    int32_t kadd, ksplit;
    double *d_slabs, *v_slabs, *pnq, *vfull, *viol, *partial;
    int64_t *cand_idx; int32_t *cand_cnt; double *cand_val;
    double *stats;               /* device-side address of stats_host */
    const double *stats_host;    /* pinned host memory, double[T][8] */
    const float *cost; const revs_home_t *homes; const float *load;
    float *diff, *dsq; int32_t *status; float *pdhg_dual;
    int32_t mode;
    revs_pdhg_t pdhg;
    const int32_t *node_of;      /* node of every residence, or NULL (no fused home pass) */
    int32_t recompute_pe_new;    /* revs_plan_spec_step with use_y = 0: the sweep does not read
                                  * p_est_new (see revs_agent_step_select) */
    /* revs_plan_chain_step only (may be zero otherwise): the second set of candidate lists
     * and stats, the model's scratch, and the Newton parameters */
    int64_t *cand_idx1;
    int32_t *cand_cnt1;
    double *cand_val1, *stats1;
    const double *stats1_host;
    double *yhat, *k_full;
    int32_t *info;
    double delta, eps;
    int32_t max_pivots;
} revs_plan_desc_t;
typedef struct revs_plan revs_plan_t;
revs_plan_t *revs_plan_create(const revs_plan_desc_t *desc_host);
void revs_plan_destroy(revs_plan_t *plan);
/* phase bit 0: the home pass of the evaluation (revs_op_dual_eval into pnq / p_est_new),
 * skipped when fused_in != 0: the previous call's sweep (kept by the caller) has already done
 * it -- p_est_new and the node sums are in place.  phase bit 1: product R p_in with the row
 * bookkeeping, the sweep with the selection in its launch, then wait for the verdict:
 * rmax_out_host = largest entry [t][0] of stats (row residual, absolute; stats[t][1] is not
 * meaningful after a fused home pass).  A driver that shards residences calls phase 1,
 * all-reduces p_in, calls phase 2.
 *   p_in    node sums of this evaluation: pnq[0] unless fused_in
 *   p_out   NULL, or (needs use_y = 0) the array -- not p_in -- into which this call's sweep
 *           accumulates the NEXT evaluation's node sums while writing p_est_next; it is
 *           cleared on the way
 * ev_mid / ev_end: optional hipEvent_t handles recorded between evaluation and sweep / after
 * the sweep.
 * phase bit 3 (value 8, with bit 1; used by revs_plan_spec_run): after the sweep and BEFORE
 * waiting for the verdict, the product + rows of the NEXT iteration are enqueued too (node sums
 * p_out, clearing p_in's array): they depend on this sweep only, and the queue does not run
 * dry while the host turns around.  phase bit 2 (value 4): this call's product was enqueued
 * that way by the previous call -- skip it.  If the sweep is discarded the work run ahead is
 * never read: the caller's next evaluation rewrites every array it touched.
 * For a driver that has to act between the stages (the sharded one exchanges node sums):
 * phase bit 4 (value 16, with bit 1) returns after the launches without waiting; phase = 32
 * only waits for the verdict of the launches made last; phase = 64 only enqueues a product
 * R p_in + rows (clearing p_out) -- the one a later call skips with bit 2. */
int revs_plan_spec_step(revs_plan_t *plan, int32_t phase, const double *y, int32_t use_y,
                        const float *p_est, float *p_est_new, const float *p_sch,
                        const float *gamma, float *p_sch_out, float *gamma_out,
                        float *s_out, float *c_out, int32_t fused_in, const double *p_in,
                        double *p_out, float *p_est_next, double *rmax_out_host,
                        void *ev_mid, void *ev_end, void *stream);

/* Up to max_steps consecutive steady-state iterations in one host call (one GPU, multipliers
 * all zero, the next evaluation's home pass folded into every sweep, no S / C output): each
 * is revs_plan_spec_step(phase 3, use_y = 0, p_out != NULL) on the buffer roles in `st`,
 * judged by rmax / scale <= eps, and -- if kept -- followed by the role rotation of
 * AdmmEngine.step (P_sch / G with their spares; P_est <- P_est_new <- P_est_alt <- P_est;
 * node sums alternating between p0 and p_alt).  The call returns after max_steps kept
 * iterations or after the first one whose sweep must be discarded: that iteration's launches
 * have been made, `st` still holds ITS roles, *last_fused_in tells whether its evaluation
 * took the node sums from the previous sweep, *rmax_out its row residual; *kept_steps counts
 * the kept ones before it. */
typedef struct {
    float *p_est, *p_est_new, *p_est_alt;
    float *p_sch, *p_sch_alt, *gamma, *gamma_alt;
    double *p0, *p_alt;          /* the two node-sum arrays (p0 = pnq[0] of the plan) */
    double *fused_p;             /* where the last kept sweep left the next node sums */
    int32_t fused_ready;         /* ... if it did */
} revs_spec_state_t;
int revs_plan_spec_run(revs_plan_t *plan, int32_t max_steps, const double *y,
                       revs_spec_state_t *st, double scale, double eps, int32_t *kept_steps,
                       int32_t *last_fused_in, double *rmax_out, void *stream);

/* The binding steady state as ONE host call (operator_newton.py: AdmmEngine._chain_launch /
 * _chain_accept; one GPU): evaluation of the multipliers y into candidate set 0 (home pass
 * row-wise from the lists of set `sup0` when use_y and sup0 >= 0, else dense),
 * revs_op_dual_select_model_step into y_trial, evaluation of y_trial into set 1 (home pass
 * row-wise from set 0 when chain_few), the home sweep on its answer p_est_new with the
 * trial's selection in its launch (revs_agent_step_select, P_sch / G into the _out
 * buffers); then the sequence tag of set 1 is polled and both stats blocks are judged by
 * revs_newton_chain_accept: *accepted = its verdict, nsup_sum / nsup_max as there.  The
 * stats blocks stay in stats_host / stats1_host for a caller that has to go on. */
int revs_plan_chain_step(revs_plan_t *plan, const double *y, double *y_trial, int32_t use_y,
                         int32_t sup0, int32_t chain_few, const float *p_est, float *p_est_new,
                         const float *p_sch, const float *gamma, float *p_sch_out,
                         float *gamma_out, float *s_out, float *c_out, int32_t *accepted,
                         int32_t *nsup_sum, int32_t *nsup_max, void *ev_mid, void *ev_end,
                         void *stream);

/* Up to max_steps consecutive iterations of the binding steady state in one host call:
 * revs_plan_chain_step on the roles in `st`; an accepted iteration is followed by the role
 * rotation of AdmmEngine.step / _chain_book (y <-> y_trial; use_y = any multiplier left;
 * sup0 = 1 if the row-wise home pass applies to them -- at most 48 - kadd per slot -- else
 * -1; P_sch / G with their spares; P_est <-> P_est_new).  Returns after max_steps accepted
 * iterations or after the first one that is not (its launches made, `st` holding ITS roles,
 * the stats blocks in place for the caller's general loop); *kept_steps counts the accepted
 * ones. */
typedef struct {
    double *y, *y_trial;
    int32_t use_y, sup0;
    float *p_est, *p_est_new;
    float *p_sch, *p_sch_alt, *gamma, *gamma_alt;
} revs_chain_state_t;
int revs_plan_chain_run(revs_plan_t *plan, int32_t max_steps, revs_chain_state_t *st,
                        int32_t chain_few, int32_t *kept_steps, void *stream);

/* The binding steady state with ONE pass over the residences per ADMM iteration (one GPU, the
 * feeder as a tree of at most REVS_TREE_SWEEP_MAX nodes, few multipliers per slot: the row-wise
 * form of the shifts).  revs_plan_chain_step makes three passes over the residences per iteration
 * (two evaluations' home passes and the sweep) and five launches; here the sweep forms the
 * operator's answer for the trial multipliers itself -- pen = max(g0 - d[node], 0), the evaluation
 * kernel's arithmetic -- and folds BOTH evaluations' node sums into its own pass: those of the trial
 * on the current state (its verdict) and those of the same multipliers on the state it has just
 * produced (the next iteration's first evaluation).  The operator side is ONE launch of 2 T
 * workgroups behind the sweep: [0, T) judge the trial (rows by the tree form, selection: the stats
 * the host polls), [T, 2T) already run rows, selection, small model and step of the NEXT iteration.
 * The host accepts iteration k on the same test as revs_plan_chain_step
 * (revs_newton_chain_accept) while that launch is still busy; the next sweep is already in the queue
 * behind it (st->p_est_3 ...: enqueued unjudged, round 4): no gap, two launches per iteration.  A
 * rejected iteration leaves the state untouched (the sweeps wrote to the spares only) for the caller's
 * general loop.
 *   st   y / y_trial / y_spare: three multiplier arrays (double[m][T]); roles rotate by the kept
 *        iterations (y = the accepted multipliers at return); use_y, sup0 as revs_chain_state_t
 *        (sup0 = -1 at return); state buffers as there; s_out / c_out: NULL, or where the FIRST
 *        iteration of the call writes schedules and SOC;
 *        resume: in, 1 = the previous call kept all its iterations and nothing has touched the
 *        state or the multipliers since (its last launch has already prepared this call's first
 *        iteration); out, 1 = this call ended that way; out, 2 = the call stopped at an iteration whose
 *        full Newton step passed the line search but left the rows above the tolerance: y IS that step
 *        (y_trial the multipliers before it), the state is untouched, and the caller's loop continues the
 *        operator's solve from y.  (Up to two such steps per iteration are taken inside the call: the
 *        trial's evaluation on the current state is what the sweep has folded, so the operator launch runs
 *        on those sums and sweep and launch are made again: `redone`.) */
typedef struct {
    double *y, *y_trial, *y_spare;
    int32_t use_y, sup0;
    float *p_est, *p_est_new;
    float *p_sch, *p_sch_alt, *gamma, *gamma_alt;
    float *s_out, *c_out;
    int32_t resume;
    int32_t pivots;     /* out, with resume = 2: pivots taken by the model of the step y */
    int32_t redone;     /* out: Newton steps beyond the first that the call's LAST iteration took inside the call
                           (a kept one: the call returns behind it; or, with resume = 2, the one handed back) */
    /* A third set of state buffers (all three or none).  With it the sweep of iteration k + 1 is enqueued right
     * behind the operator launch of iteration k, BEFORE the host has seen that iteration's verdict (it reads the
     * spares iteration k wrote and writes this set): the host's poll - accept - launch turnaround (~6 us) is off the
     * GPU's critical path.  A rejected iteration k leaves that sweep's output unused (the call waits for it);
     * the three sets' roles rotate by the kept iterations. */
    float *p_est_3, *p_sch_3, *gamma_3;
    /* The residences' carried PDHG multipliers (float[n]; NULL x 3 when the plan has none): a sweep reads
     * pdhg_dual and writes pdhg_dual_new (the unjudged one: reads that, writes pdhg_dual_3); roles rotate with the
     * state, so a rejected iteration leaves the multipliers it started from untouched -- the sweep that replaces it
     * starts from the same warm start, bit for bit.  At return the plan points at pdhg_dual. */
    float *pdhg_dual, *pdhg_dual_new, *pdhg_dual_3;
} revs_chain_fold_state_t;
int revs_plan_chain_fold_run(revs_plan_t *plan, int32_t max_steps, revs_chain_fold_state_t *st,
                             int32_t *kept_steps, void *stream);
/* Newton steps beyond the first that revs_plan_chain_fold_run takes inside the call per iteration
 * (default 2; 0: every such iteration is handed back at its step, resume = 2). */
int revs_plan_set_fold_redo(revs_plan_t *plan, int32_t steps);
/* revs_plan_newton_solve admits desc.kadd violated rows per slot and Newton iteration -- and kadd_cold of them (0: off)
 * in an evaluation that follows one in which some slot showed more than cold_at violated rows without a multiplier AND
 * either at least half of the rows admitted the time before kept a multiplier or some slot carries 16 multipliers already
 * (a cold solve on a feeder whose rows bind one by
 * one: lpsolver.py:183-194 hands Gurobi every row at once; here two at a time would be as many Newton iterations as half
 * the rows that end up binding). */
int revs_plan_set_kadd_cold(revs_plan_t *plan, int32_t kadd_cold, int32_t cold_at);
/* Allocates now what the run loops (revs_plan_stream_run_blocks, revs_plan_chain_fold_run) otherwise allocate the first
 * time they are entered -- the ring of node-sum slices and the event pool for the block size, overlap and communicator
 * set so far, the folded chain's buffers -- so that a fresh plan's first run makes no allocation between its launches.
 * Optional; call after revs_plan_set_tree / set_comm / set_stream_block. */
int revs_plan_prepare(revs_plan_t *plan);

/* ---- the operator's Newton solve as ONE native call -----------------------------------------
 * Utility(...).solve() (reference lpsolver.py:163-238, called at lpsolver.py:256-259) through its dual:
 * the loop of revs_admm_amd/operator_newton.py:_operator_solve_newton -- evaluate the multipliers, and
 * while rows are beyond eps: model of every slot on its candidate set (revs_op_dual_model_small when
 * every slot has at most 8 candidates, else revs_op_dual_model), Armijo backtracking per slot on the
 * dual value (step, evaluation of the trial; halve the pending slots' steps), the same stopping, stalling
 * and hand-off tests -- with the launches, the waits on the pinned stats blocks, the line search and the
 * bookkeeping in the library: one read of a stats block per evaluation and no interpreter in between
 * (round 3 measured ~80 us per evaluation from Python for ~45 us of kernels).  Residences sharded: the
 * plan's communicator (revs_plan_set_comm) all-reduces p | N | q between an evaluation's phases.
 * Everything is deterministic: the iterates are those of the Python loop, bit for bit.
 *   revs_plan_set_newton: the buffers the plan descriptor does not carry -- the general model's Gram
 *        slabs (double[T][nks][128][128]), the step lengths (pinned host double[T] and its device
 *        address: written by the host, read by the step kernel through the mapping), the host view of
 *        desc.info (pinned int32[T]: the models' pivot counts) -- and the loop's limits.
 *   st   in: y = the current multipliers, y_trial = scratch (double[m][T] each); use_y: y is not all zero;
 *        sup: candidate set (0 / 1) that lists every row with y != 0 (few of them: row-wise shifts), or
 *        -1; the state; have_first: stats block 0 already holds the evaluation of y on this state (first_tag);
 *        have_pre: ... and block 1 the evaluation of the chain's trial (small model on set 0, full step
 *        for the slots not within tolerance), made with the row-wise shifts iff chain_few_in;
 *        out: ok (P_est_new holds the answer to tolerance; else y has been cleared and the caller's ADMM
 *        forms take the iteration), y / y_trial swapped so that y is the accepted array, newton / evals /
 *        pivots / models_small / models_general, last_small, few, pre_kept (the accepted state is exactly
 *        what the chain's launches left), cur (the candidate set of the accepted evaluation),
 *        nsup_sum / nsup_max (rows with a multiplier: all slots / the fullest slot). */
typedef struct {
    double *k_slabs;
    int32_t nks;
    double *alpha_host;
    const double *alpha_dev;
    const int32_t *info_host;
    int32_t newton_max, ls_max;
} revs_newton_opts_t;
int revs_plan_set_newton(revs_plan_t *plan, const revs_newton_opts_t *opts);
typedef struct {
    double *y, *y_trial;
    int32_t use_y, sup;
    const float *p_est, *p_sch, *gamma;
    float *p_est_new;
    int32_t have_first, have_pre, chain_few_in;
    int32_t ok, newton, evals, pivots, models_small, models_general, last_small, few, pre_kept, cur, nsup_sum, nsup_max;
    /* in: the sequence tags ([8 t + 5] of every slot's record) of the evaluations have_first / have_pre refer to, as the
     * caller saw them (0: not checked).  A block that no longer carries its tag -- a launch wrote it since -- is
     * REVS_EINVAL, not a solve from other numbers. */
    double first_tag, pre_tag;
    /* out: the solve stopped because a slot holds more rows than a model of REVS_DUAL_AMAX (128) takes -- more multipliers
     * than that, or a full model with rows still violated.  y is NOT cleared then (ok = 0): the caller goes on from it on
     * lists of up to REVS_DUAL_AMAX_BIG rows (include/revs_admm_ops.h: revs_op_dual_*_big). */
    int32_t big_needed, reserved_;
} revs_newton_state_t;
int revs_plan_newton_solve(revs_plan_t *plan, revs_newton_state_t *st, void *stream);

/* ---- the feeder as a tree: R p in O(nodes) ------------------------------------------
 * The reference forms the LinDistFlow sensitivity matrix R = 2 F D F^T densely
 * (lpsolver.py:17-26) and checks R_res g[:,t] against the limits (lpsolver.py:188-193).  On
 * a radial feeder R[i][j] = 2 * (sum of r over the edges shared by the root->i and root->j
 * paths), so v = R p is two tree passes: P_sub(e) = load below edge e, then
 * v_i = sum over the edges e on the root->i path of 2 r_e P_sub(e).  With the tree nodes in
 * DFS preorder (a subtree = a contiguous range [j, end_j)) both passes are prefix sums:
 *     C      = exclusive prefix of the injections                  P_sub(j) = C[end_j] - C[j]
 *     w'_j   = w_j P_sub(j),  w_j = 2 r(edge to the parent)
 *     v_j    = sum_{a <= j} w'_a - sum_{a : end_a <= j} w'_a      (= over the ancestors a of j)
 * the second sum taken from a prefix over the nodes sorted by `end`.  All arrays below live
 * on the device and are indexed by preorder position; n <= REVS_TREE_MAX and n % 8 == 0 (pad with
 * weightless nodes hanging off the substation: src = -1, w = 0, end = j + 1):
 *   src[j]  constraint row (0..m-1) whose node sum p[src][t] is injected at -- and whose
 *           voltage row is checked at -- position j; -1: a node without residences
 *   end[j]  one past the last position of j's subtree
 *   eo[k]   the positions sorted by end (stable)
 *   cle[j]  number of positions a with end[a] <= j
 * packed 16 bits each into pack[j] = (src[j] + 1) | end[j] << 16 | eo[j] << 32 | cle[j] << 48 (one
 * 64-bit load per position: the evaluation is latency-bound), and
 *   w[j]    2 r of the edge from j to its parent (double)
 * revs_tree_voltage: one workgroup per slot; v_out double[m][T] (rows with src; may be NULL),
 * rmax_out double[T] = largest violation max(v - vhi, vlo - v, 0) over the checked rows. */
int revs_tree_voltage(int32_t m, int32_t T, const revs_tree_t *tree_host, const double *p,
                      double vlo, double vhi, double *v_out, double *rmax_out, void *stream);

/* ---- RCCL communicator owned by the library ---------------------------------------------
 * Residences shard over ranks; the only data-path collective is the all-reduce (sum, f64) of
 * the M x T node sums, enqueued by the library itself on the compute stream between two
 * sweeps (no host in between, no second stream).  librccl.so.1 is opened at run time
 * (the copy already mapped into the process, i.e. PyTorch's, when there is one): a
 * single-GPU process never needs it.  The 128-byte unique id is made on rank 0 and carried
 * to the other ranks by the caller (any host channel). */
typedef struct revs_comm revs_comm_t;
int revs_comm_unique_id(void *id128_out);
revs_comm_t *revs_comm_create(const void *id128, int32_t rank, int32_t nranks);
void revs_comm_destroy(revs_comm_t *comm);
int revs_comm_allreduce_f64(revs_comm_t *comm, double *buf, int64_t count, int32_t op /* 0 sum, 2 max, 3 min */,
                            void *stream);
/* The same communicator over the CALLER's transport: `fn(ctx, host_buf, count, op)` must leave in
 * host_buf (pinned host memory owned by the library, `count` doubles) the element-wise reduction
 * over all ranks and return 0.  revs_comm_allreduce_f64 then waits for the stream, stages the
 * buffer to the host, calls fn, and stages the result back before it returns -- synchronous,
 * for ranks that have no RCCL path between them (RCCL refuses two ranks on one device; MPI or
 * gloo over the host; the two-process tests of the sharded loop, tests/test_gpu_sharded.py).
 * Every rank must make the same sequence of calls, as with RCCL. */
typedef int (*revs_host_allreduce_fn)(void *ctx, double *host_buf, int64_t count, int32_t op);
revs_comm_t *revs_comm_create_hook(revs_host_allreduce_fn fn, void *ctx, int32_t rank, int32_t nranks);

/* ---- streaming steady state: one launch per ADMM iteration, no host in the loop ---------
 * While the operator's multipliers are zero an iteration of lpsolver.py:254-287 is ONE
 * launch: its first T workgroups judge the voltage rows of the estimate P_est[k+1] the
 * previous sweep prepared (tree form above, node sums p[0]), every other workgroup solves
 * its residences from that estimate, updates G, writes the next estimate P_est[k+2] and
 * accumulates its node sums into p[1]; p[2] is cleared on the way for the launch after.
 * A launch whose rows are NOT within eps * scale records its sequence number in the plan's
 * control block, and every later launch returns at once (all workgroups read that word
 * first) -- so the host can keep the queue full without waiting for any verdict, and
 * nothing written after a failed verdict has to be undone beyond that one sweep's spare
 * buffers.  With a communicator set (revs_plan_set_comm) the node sums are all-reduced on
 * the same stream after every sweep (block <= 1 only: see revs_plan_stream_run_blocks below for
 * the form with one collective per block).  All max_steps launches (at most REVS_STREAM_MAX) are
 * enqueued in one burst and no decision is taken in between, so every rank of a sharded run
 * issues the same collectives; the caller bounds max_steps by the silenced launches it accepts
 * to waste behind a failed verdict.
 *   st    roles at entry and, rotated by the kept steps, at return:
 *         p_est[0] = P_est[k], p_est[1] = P_est[k+1] (prepared by the previous sweep),
 *         p_est[2] spare;  p_sch[0], gamma[0] current, [1] spare;  p[0] = node sums of
 *         p_est[1] (already all-reduced), p[1] = ZERO, p[2] = any
 * Returns after max_steps kept iterations, or after the first whose verdict failed
 * (*kept_steps < max_steps; the stream has drained; roles are those of the failed
 * iteration, whose sweep wrote only to the spares). */
typedef struct {
    float *p_est[3];
    float *p_sch[2];
    float *gamma[2];
    double *p[3];
    float *diff_hist;       /* NULL, or float[max_steps][n_homes]: launch k writes the residences' diff
                             * (lpsolver.py:284) to row k instead of the plan's diff array; advanced
                             * by the kept steps at return -- the per-iteration diff of solve_ADMM
                             * without a host round trip per iteration */
} revs_stream_state_t;
int revs_plan_set_tree(revs_plan_t *plan, const revs_tree_t *tree_host);
int revs_plan_set_comm(revs_plan_t *plan, revs_comm_t *comm);
#define REVS_STREAM_MAX 1023
int revs_plan_stream_run(revs_plan_t *plan, int32_t max_steps, revs_stream_state_t *st,
                         double scale, double eps, int32_t *kept_steps, double *rmax_last,
                         void *stream);
/* Verdicts by blocks -- the form of the loop above that the engine uses by default, and the only one
 * that makes sense with residences sharded: the node sums of an iteration are only known after the
 * all-reduce, and one collective per sweep would make the collective's latency the step.  With
 * block = B > 1 (revs_plan_set_stream_block), revs_plan_stream_run_blocks lets B iterations run
 * without a verdict, each accumulating its node sums into its own slice of a ring owned by the
 * plan; ONE all-reduce sums the B slices over the ranks and one launch of B x T workgroups judges
 * them (tree form, as above).  Because no verdict is needed between them, `inner` consecutive
 * iterations (revs_plan_set_stream_inner, <= REVS_AGENT_MAX_INNER) are ONE launch that keeps every
 * residence's state in registers (revs_agent_step_multi): profiles read and written once per
 * `inner` iterations.  Every launch of the burst is still a no-op once an iteration at or before
 * its own has failed; the sweeps that ran behind a failed iteration are undone WITHOUT copies: the
 * caller hands over FOUR sets of state buffers, a block's launches alternate between the two sets
 * that are neither its own entry set nor the entry set of the block before, so the state a block
 * started from is intact until its verdicts are in; after a failure the call runs the good
 * iterations of that block again from its entry set, then the failed iteration's own sweep (outputs
 * to a spare set, carried multipliers in place) -- memory is then bit for bit what the loop that
 * judges every launch leaves.
 *   st    sets [0..3] of {P_est, P_sch, G, carried PDHG multipliers (PDHG with warm start only)};
 *         at entry set 0 is the state (P_est[k], P_sch[k], G[k]; its pdhg_dual must be the plan's),
 *         the others are scratch; at return set 0 is the state after the kept iterations (the roles
 *         are permuted, the plan's pdhg_dual follows -- see revs_plan_set_pdhg_dual).
 *         p0: node sums of the estimate P_est[k+1] (all-reduced); p0_out (another array): those of
 *         P_est[k+kept+1] when every iteration was kept; p_est_next (a buffer outside the sets):
 *         P_est[k+kept+1] itself then; diff_hist as in revs_stream_state_t.
 *   dmax_out  NULL, or double[max_steps]: [i] = max_h diff[h] of the i-th iteration of this call
 *         (lpsolver.py:284 -- the convergence measure), for the kept iterations: folded on the
 *         device by the sweeps (REVS_DMAX_SLOTS partial maxima per rank in the tail of every ring
 *         slice, so that the all-reduce of the sums also gathers them) and the verdict launches.
 * overlap != 0 (and more than `block` iterations in the call): the all-reduce and the verdicts of
 * block b run on a second stream owned by the plan while the caller's stream already runs the
 * sweeps of block b + 1 (two ring halves; block b + 2 waits for block b's verdicts) -- the
 * collective then costs the step nothing as long as it is shorter than a block of sweeps; the last
 * `block` iterations are split 3 : 1 so that the last collective is a short one.  The caller's
 * stream waits for the second stream before the call returns, so synchronising the caller's
 * stream is still enough.
 * Memory: 2 B (M T + REVS_DMAX_SLOTS ranks) doubles.  Set block, inner and overlap on every rank alike. */
#define REVS_STREAM_BLOCK_MAX 256
typedef struct {
    float *p_est[4];
    float *p_sch[4];
    float *gamma[4];
    float *pdhg_dual[4];
    const double *p0;
    double *p0_out;
    float *p_est_next;
    float *diff_hist;
} revs_stream_sets_t;
int revs_plan_set_stream_block(revs_plan_t *plan, int32_t block, int32_t overlap);
int revs_plan_set_stream_inner(revs_plan_t *plan, int32_t inner);
int revs_plan_stream_run_blocks(revs_plan_t *plan, int32_t max_steps, revs_stream_sets_t *st,
                                double scale, double eps, int32_t *kept_steps, double *rmax_last,
                                double *dmax_out, void *stream);
/* The array of carried PDHG multipliers the plan's other entry points (revs_plan_spec_step, ...)
 * use -- revs_plan_stream_run_blocks rotates it with the sets and leaves the plan pointing at set 0's. */
int revs_plan_set_pdhg_dual(revs_plan_t *plan, float *pdhg_dual);
/* Two HIP events around the bursts of revs_plan_stream_run, recorded by the library on the
 * bursts' own stream: revs_plan_stream_timing(plan, 1) arms them (the next burst records the
 * first one, every burst re-records the second after its last launch);
 * revs_plan_stream_elapsed_ms returns the time between them once the stream has been
 * synchronised, and re-arms.  What bench.py prices the sweep kernel's launch duration with. */
int revs_plan_stream_timing(revs_plan_t *plan, int32_t enable);
int revs_plan_stream_elapsed_ms(revs_plan_t *plan, double *ms);
/* ... and the number of residence-sweep launches enqueued between the two events */
int64_t revs_plan_stream_launches(revs_plan_t *plan);
/* ... and, with residences sharded and the timing armed, two more events around the all-reduce of the node sums of
 * the last call's FIRST block (*iterations slices in one collective) on the stream it is issued on -- the second stream
 * when the blocks overlap: *collective_ms = that all-reduce's duration, *block_ms = the duration of that block's sweep
 * launches on the caller's stream (a collective that runs beside the next block's sweeps is hidden as long as it is
 * the shorter one).  REVS_EINVAL when none has run since the timing was armed (one GPU). */
int revs_plan_collective_ms(revs_plan_t *plan, double *collective_ms, double *block_ms, int32_t *iterations);
/* Bits OR-ed by the sweeps launched through the plan since the last clear: 1 = a residence's
 * window cannot reach 90 % SOC (the reference prints "No solution found", lpsolver.py:153-155),
 * 2 = a PDHG residence hit max_iter before its tolerance.  Meaningful after the stream has
 * been synchronised. */
int32_t revs_plan_status_flags(revs_plan_t *plan, int32_t clear);

#ifdef __cplusplus
}
#endif
#endif /* REVS_ADMM_H */
