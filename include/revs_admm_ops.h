/*
 * revs_admm_ops.h -- the operator QP's building blocks in librevs_admm.so (same library, same conventions
 * as revs_admm.h): the launches that revs_plan_newton_solve / revs_plan_chain_fold_run /
 * revs_plan_stream_run_blocks are made of.  NOT part of the drop-in boundary: a reference-side
 * binding needs revs_admm.h only.  They are exported because revs_admm_amd's Python driver issues them one by
 * one on its general paths (operator_admm.py: the ADMM forms of the operator QP; operator_newton.py: the chain
 * issued in phases, the Python Newton loop kept for comparison) and because the kernel-level parity tests
 * (tests/test_gpu_newton.py, test_gpu_operator.py) call each of them against its numpy restatement.
 */
#ifndef REVS_ADMM_OPS_H
#define REVS_ADMM_OPS_H

#include "revs_admm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- operator ("Utility") side -------------------------------------------
 * lpsolver.py:163-238:  min (kappa/2)|g - g0|^2  s.t.  g >= 0,
 *                        vlo <= R (A g) <= vhi   per slot,
 * g0 = (P_est + P_sch)/2 - G/kappa, A = home->node aggregation, R the LinDistFlow
 * sensitivity matrix of the constrained nodes (compute_Rmat, lpsolver.py:17-26).
 * Solved by ADMM in OSQP form with the KKT matrix applied through the
 * eigendecomposition D^1/2 R D^1/2 = Q L Q^T (D = diag(n_m); voltage row m is scaled by
 * sqrt(n_m), which keeps the operator matrix symmetric), so rho can be re-tuned per
 * slot without refactoring.  All operator arithmetic is double (the QP is ill conditioned:
 * cond(R)^2 ~ 5e7 on the 121144 feeder); the two (concatenated) products with Q per inner
 * iteration run on v_mfma_f64_16x16x4_f64.
 */

/* C[m][n] (+)= At^T * B on the matrix cores.  At is stored k-major (double
 * At[k][lda], element At[kk][i] = A[i][kk]); B is double[k][ldb]; C double[m][ldc].
 * n is small (T, up to 192).  accumulate != 0 adds into C.  f64 uses
 * v_mfma_f64_16x16x4_f64, f32 v_mfma_f32_16x16x4_f32 (exact f32 fma chain). */
int revs_gemm_tn_f64(int32_t m, int32_t n, int32_t k, const double *At, int32_t lda,
                     const double *B, int32_t ldb, double *C, int32_t ldc,
                     int32_t accumulate, void *stream);
int revs_gemm_tn_f32(int32_t m, int32_t n, int32_t k, const float *At, int32_t lda,
                     const float *B, int32_t ldb, float *C, int32_t ldc,
                     int32_t accumulate, void *stream);
/* C = At^T B with dense leading dimensions, K cut over `ksplit` groups of workgroups:
 * C holds ksplit partial slabs of m*n doubles (see _x2). */
int revs_gemm_tn_f64_split(int32_t m, int32_t n, int32_t k, const double *At, const double *B,
                           double *C, int32_t ksplit, void *stream);
/* [C0 | C1] = At^T [B0 | B1]: one product whose right-hand side and result are each the
 * horizontal concatenation of two double[k][T] / double[m][T] arrays (2T <= 192), so
 * the matrix is streamed once for both.  With the voltage rows scaled by sqrt(n_m) the
 * operator's matrix D^1/2 R D^1/2 = Q L Q^T is symmetric and one inner iteration is
 *   [ta | tb] = Q^T [rhat | w]   and   [va | usa] = Q [a | sa]
 * -- two launches instead of four products.  ksplit / slabs as for _x2. */
int revs_gemm_tn_f64_cat(int32_t m, int32_t T, int32_t k, const double *At, const double *B0,
                         const double *B1, double *C0, double *C1, int32_t ksplit,
                         void *stream);
/* Two independent products of the same shape in ONE launch (dense leading
 * dimensions lda = m, ldb = ldc = n, no accumulate): C0 = At0^T B0, C1 = At1^T B1.
 * ksplit (1..8) cuts K over ksplit groups of workgroups so a small m still fills the
 * chip; group s writes its PARTIAL product to slab s, i.e. C0 and C1 must each hold
 * ksplit slabs of m*n doubles and the product is the sum of the slabs (the
 * revs_op_node_* consumers add them, in slab order). */
int revs_gemm_tn_f64_x2(int32_t m, int32_t n, int32_t k, const double *At0, const double *B0,
                        double *C0, const double *At1, const double *B1, double *C1,
                        int32_t ksplit, void *stream);

/* Segmented home->node aggregation: out[node][t] = scale[node] * sum over the
 * homes of that node of in[home][t].  Homes are sorted by node; node_ptr is the
 * CSR offset array int64[m+1].  scale may be NULL (=1).  Deterministic. */
int revs_aggregate_f64(int32_t m, int32_t T, const int64_t *node_ptr,
                       const double *in_home, const double *scale, double *out_node,
                       void *stream);
int revs_aggregate_f32(int32_t m, int32_t T, const int64_t *node_ptr,
                       const float *in_home, float *out_node, void *stream);

/* g0 = (P_est + P_sch)/2 - G/kappa : the unconstrained minimiser of the Utility
 * objective (lpsolver.py:196-207).  float in, double out. */
int revs_op_g0(int64_t n_homes, int32_t T, const float *p_est, const float *p_sch,
               const float *gamma, float kappa, double *g0, void *stream);

/* Operator ADMM state.  Per home and slot ONE double s_b = z_b + y_b is kept
 * (z_b = max(s_b,0) and y_b = min(s_b,0) are complementary); per node z_v, y_v.
 * Cold start: s_b = max(g0,0); z_v = clip(cx, vlo, vhi), y_v = 0, w = rho_v z_v
 * (cx = C_v x for that start, formed by the driver). */
int revs_op_init_home(int64_t n_homes, int32_t T, const double *g0, double *sb, void *stream);
int revs_op_init_node(int32_t m, int32_t T, const double *cx, const double *rho_v,
                      const double *bound_scale, double vlo, double vhi, double *zv, double *yv,
                      double *w, void *stream);

/* Home pass of one inner iteration (sb, g0 double[n][T]; rho_b double[T];
 * c = kappa + rho_b[t]; node m owns homes node_ptr[m]..node_ptr[m+1]-1):
 *   z = max(s_b,0), y = min(s_b,0)
 *   if xc != NULL (double[m][T], node correction from revs_op_node_update):
 *       xt  = (kappa g0 + rho_b z - y) / c + inv_sqrt_n[m] xc[m]      x-update
 *       u   = alpha xt + (1-alpha) z + y/rho_b
 *       z   = max(u,0), y = rho_b min(u,0), s_b = z + y
 *       if res != NULL (double[8][T], see revs_op_node_update; needs cty_node =
 *       V S U^T y_v, double[m][T]): rows 1,2,5,6,7 get the per-slot maxima of
 *       |xt - z|, |kappa (xt-g0) + C^T y|, |xt|, |C^T y|, |kappa g0|
 *   rhat[m] = inv_sqrt_n[m] * sum_homes (kappa g0 + rho_b z - y)
 * With homes sharded over GPUs rhat is this rank's partial sum: all-reduce it. */
int revs_op_home_pass(int32_t m, int32_t T, const int64_t *node_ptr,
                      const double *inv_sqrt_n, double *sb, const double *g0,
                      const double *xc, const double *rho_b, double kappa, double alpha,
                      double *rhat, const double *cty_node, double *res, void *stream);

/* revs_op_node_update (below) followed by revs_op_home_pass, in ONE launch: the workgroup
 * of node m first updates row m of (xc, z_v, y_v, w) from the products va, usa (nslab
 * slabs each) and the current rhat, then runs the home pass with that xc.  No residuals
 * in this form: the checking iteration of a block uses the two separate calls. */
int revs_op_home_pass_fused(int32_t m, int32_t T, const int64_t *node_ptr,
                            const double *inv_sqrt_n, double *sb, const double *g0,
                            const double *rho_b, double kappa, double alpha, double *rhat,
                            int32_t nslab, const double *va, const double *usa,
                            const double *rho_v, const double *bound_scale, double vlo,
                            double vhi, double *xc, double *zv, double *yv, double *w,
                            void *stream);

/* Node passes (double[m][T]; s double[m] singular values; rho_v, rho_b double[T]):
 *   revs_op_node_w:      w  = rho_v z_v - y_v
 *   revs_op_row_scale:   out = s (per row) * in
 *   revs_op_node_scale:  a  = (ta + s tb) / (c + rho_v s^2),  sa = s a
 *                        with ta = V^T rhat, tb = U^T w from revs_gemm_tn_f64_x2, each
 *                        given as nslab K-split slabs (double[nslab][m][T]) that are summed
 *   revs_op_node_update: (va, usa as nslab slabs, like ta/tb)
 *                        xc = va - rhat/c            (va = V a)
 *                        h  = alpha usa + (1-alpha) z_v   (usa = U sa = C_v xt)
 *                        z_v = clip(h + y_v/rho_v, b vlo, b vhi); y_v += rho_v (h - z_v)
 *                        (b = bound_scale[m], or 1 when NULL: row m of the voltage block
 *                        is stored scaled by sqrt(n_m), see below)
 *                        w = rho_v z_v - y_v
 *                        if res != NULL: rows 0,3,4 get max|usa - z_v|, |usa|, |z_v|
 * res is double[8][T], must be ZERO before the checking iteration (maxima are merged
 * with atomicMax) and feeds the driver's stopping test and per-slot rho update. */
int revs_op_node_w(int32_t m, int32_t T, const double *zv, const double *yv,
                   const double *rho_v, double *w, void *stream);
int revs_op_row_scale(int32_t m, int32_t T, const double *s, const double *in, double *out,
                      void *stream);
int revs_op_node_scale(int32_t m, int32_t T, int32_t nslab, const double *ta, const double *tb,
                       const double *s, const double *rho_v, const double *rho_b,
                       double kappa, double *a, double *sa, void *stream);
int revs_op_node_update(int32_t m, int32_t T, int32_t nslab, const double *va,
                        const double *rhat, const double *usa, const double *rho_v,
                        const double *rho_b, const double *bound_scale, double kappa,
                        double alpha, double vlo, double vhi,
                        double *xc, double *zv, double *yv, double *w, double *res,
                        void *stream);

/* ---- node-space fast path of the operator QP ------------------------------------
 * While no residence is pushed to g = 0 the problem collapses to the M constrained nodes:
 * g = g0 + A~^T d, cost (kappa/2)|d|^2, rows b vlo <= Rs (p0 + d) <= b vhi, p0 = A~ g0,
 * Rs = D^1/2 R D^1/2 = Q L Q^T.  ADMM on (x = p0 + d, z = Rs x) in the eigenbasis needs two
 * T-column products per iteration, no home-space traffic and no communication; per OUTER
 * iteration the ranks exchange only p0 (sum) and gmin (min).  The driver accepts the
 * answer iff slack = gmin + isn d >= 0 everywhere, else it runs the general path above.
 *
 *   revs_op_node_prep      p0[m] = isn[m] sum_i g0_i, gmin[m] = min_i g0_i (double[m][T]),
 *                          g0 = (P_est + P_sch)/2 - G/kappa from the float state;
 *                          g0_out (double[n][T]) or NULL.  preclamp != 0 uses max(g0, 0):
 *                          an exact presolve when R >= 0 entrywise and vlo <= 0 (only upper
 *                          rows can bind, so every node shift is <= 0 and a residence with
 *                          g0 < 0 sits at zero whatever the voltage rows do)
 *   revs_op_nodefast_feas   the operator's voltage check proper: v0 = Rs p0 (nslab slabs of
 *                           Q (l ph0)) against the bounds; cx = v0; stats (double[2], ZERO on
 *                           entry): [0] = largest row violation (0 <=> g0 already respects
 *                           every voltage row), [1] = max(0, -min gmin) (> 0 <=> some
 *                           residence has g0 < 0).  Both 0 <=> the answer is g0 itself
 *   revs_op_nodefast_scale  xh = (kappa ph0 + l wh)/(kappa + rho_v l^2), sx = l xh
 *                           (wh = Q^T w as nslab slabs, ph0 = Q^T p0)
 *   revs_op_nodefast_update z_v, y_v, w from zt = Q sx (nslab slabs); res rows 0,3,4
 *   revs_op_nodefast_dualres res rows 2,5,6,7 from yh = Q^T y_v (nslab slabs), in the
 *                           eigenbasis: |kappa (xh - ph0) + l yh|, kappa|xh| ...
 *   revs_op_nodefast_finish d = Q xh - p0 (x as nslab slabs), slack = gmin + isn d;
 *                           stats (double[2], ZERO on entry) = {max(0, -min slack), max|p0|}
 *   revs_op_node_apply      P_est_i = max(g0_i + isn[m] d[m], 0) as float (g0 as in prep)   */
int revs_op_node_prep(int32_t m, int32_t T, const int64_t *node_ptr, const double *inv_sqrt_n,
                      const float *p_est, const float *p_sch, const float *gamma, double kappa,
                      int32_t preclamp, double *p0, double *gmin, double *g0_out, void *stream);
int revs_op_nodefast_feas(int32_t m, int32_t T, int32_t nslab, const double *v0,
                          const double *bound_scale, const double *gmin, double vlo, double vhi,
                          double *cx, double *stats, void *stream);
int revs_op_nodefast_scale(int32_t m, int32_t T, int32_t nslab, const double *wh,
                           const double *ph0, const double *lam, const double *rho_v,
                           double kappa, double *xh, double *sx, void *stream);
int revs_op_nodefast_update(int32_t m, int32_t T, int32_t nslab, const double *zt,
                            const double *rho_v, const double *bound_scale, double alpha,
                            double vlo, double vhi, double *zv, double *yv, double *w,
                            double *res, void *stream);
int revs_op_nodefast_dualres(int32_t m, int32_t T, int32_t nslab, const double *xh,
                             const double *ph0, const double *lam, const double *yh,
                             double kappa, double *res, void *stream);
int revs_op_nodefast_finish(int32_t m, int32_t T, int32_t nslab, const double *x,
                            const double *p0, const double *gmin, const double *inv_sqrt_n,
                            double *d, double *slack, double *stats, void *stream);
int revs_op_node_apply(int32_t m, int32_t T, const int64_t *node_ptr, const double *inv_sqrt_n,
                       const float *p_est, const float *p_sch, const float *gamma, double kappa,
                       int32_t preclamp, const double *d, float *p_est_new, void *stream);

/* P_est = max(s_b, 0) as float: the operator's answer handed to the homes
 * (U_obj.g_opt, lpsolver.py:236-237, 259). */
int revs_op_export(int64_t n_homes, int32_t T, const double *sb, float *p_est, void *stream);

/* ---- dual Newton path of the operator QP (the default) --------------------------
 * Utility.solve (lpsolver.py:163-238) through its dual.  With y (double[m][T]) the
 * multipliers of the voltage rows, every residence of node m answers
 *     g_i = max(g0_i - d_m, 0),   d = R^T y / kappa,
 * so the dual function of slot t is
 *     D_t(y) = -(kappa/2) sum_i g_i^2 - sum_m max(vhi y_m, vlo y_m)   (+ a constant),
 * concave and C^1 with piecewise-linear gradient dD/dy_m = (R p)_m - b_m, p = A g,
 * b_m = vhi / vlo for an upper / lower row.  Each Newton iteration solves the
 * sign-constrained quadratic model on a candidate set of at most REVS_DUAL_AMAX (128) rows per
 * slot (rows with y != 0 plus the most violated ones) with generalised Hessian
 * K = R_F N R_F^T / kappa (N_m = residences of node m not clamped at zero) by block
 * principal pivoting, then takes an Armijo step on D_t.  The slots are independent
 * problems that share R and run side by side.  Per evaluation the ranks exchange pnq
 * (one all-reduce of 3 m T doubles); everything else is replicated and deterministic.
 *
 *   revs_op_dual_eval    d = (sum of the nslab slabs of R^T y)/kappa, or 0 when dsl is NULL;
 *                        g0 = (P_est + P_sch)/2 - G/kappa from the float state;
 *                        pnq[0] = p (node sums of g), pnq[1] = N (free residences),
 *                        pnq[2] = -(kappa/2) sum g^2 (double[3][m][T]); P_est_new = g (float)
 *                        when p_est_new != NULL
 *   revs_op_dual_select  v = sum of the nslab slabs of R p; per slot t: candidate rows
 *                        (cand_idx int64[T][AMAX], cand_cnt int32[T], -1 when more than AMAX
 *                        rows carry a multiplier), cand_val double[T][3][AMAX] = sign (+1
 *                        upper row, -1 lower row), gradient v - b, current y;
 *                        stats double[T][8]: [0] largest |v - b| over rows with y != 0 and
 *                        bound violation over the others, [1] D_t, [2] rows with y != 0,
 *                        [3] violated rows with y = 0 ([4] is left to revs_op_dual_step),
 *                        [5] = seq, written last behind a system-scope fence: when stats is
 *                        pinned host memory the host may poll it instead of an event.
 *                        vfull double[m][T] receives v; viol double[m][T] and partial
 *                        double[revs_op_dual_blocks(m)][T][4] are workspace.
 *   revs_op_dual_model   the model of every slot: K = R_F N R_F^T / kappa over its candidates
 *                        (R double[m][m] row-major, n_free = pnq[1]; accumulated as nks
 *                        column slabs k_slabs double[T][nks][AMAX][AMAX], summed in order
 *                        into k_full double[T][AMAX][AMAX]), then its maximiser over the sign
 *                        constraints by block principal pivoting in LDS: yhat double[T][AMAX];
 *                        info int32[T] = pivots (negative: limit hit)
 *   revs_op_dual_step    y_trial = y at the candidates moved by alpha[t] towards yhat (all
 *                        other entries of y_trial must already equal y);
 *                        lin_out[8 t] = gradient . (y_trial - y)                         */
int revs_op_dual_eval(int32_t m, int32_t T, const int64_t *node_ptr, const float *p_est,
                      const float *p_sch, const float *gamma, int32_t nslab, const double *dsl,
                      double kappa, double *pnq, float *p_est_new, void *stream);
/* revs_op_dual_eval with d = R^T y / kappa taken from the few rows of R that carry a
 * multiplier instead of the slabs of a dense product: sup_idx int64[T][AMAX] / sup_cnt
 * int32[T] (all >= 0) list, per slot, rows that include every row with y != 0 -- e.g. the
 * candidate lists of the selection that produced or last judged this y. */
int revs_op_dual_eval_rows(int32_t m, int32_t T, const int64_t *node_ptr, const float *p_est,
                           const float *p_sch, const float *gamma, const double *R,
                           const int64_t *sup_idx, const int32_t *sup_cnt, const double *y,
                           double kappa, double *pnq, float *p_est_new, void *stream);
int32_t revs_op_dual_blocks(int32_t m);
int revs_op_dual_select(int32_t m, int32_t T, int32_t nslab, const double *vsl,
                        const double *pnq, const double *y, double vlo, double vhi,
                        int32_t kadd, double *vfull, double *viol, double *partial,
                        int64_t *cand_idx, int32_t *cand_cnt, double *cand_val, double *stats,
                        double seq, void *stream);
/* The first kernel of revs_op_dual_select alone (the selection left to
 * revs_agent_step_select); zero_out double[m][T] or NULL is cleared on the way. */
int revs_op_dual_rows(int32_t m, int32_t T, int32_t nslab, const double *vsl, const double *pnq,
                      const double *y, double vlo, double vhi, double *vfull, double *viol,
                      double *partial, double *zero_out, void *stream);
/* v_slabs = R p (Rt = R^T row-major, p double[m][T]) as `ksplit` K-split slabs AND the row
 * bookkeeping of revs_op_dual_rows in one launch: the last K-split workgroup of every
 * 32-row tile sums the tile's slabs and writes vfull, viol, partial (here
 * double[(m + 31) / 32][T][4]: pass that block count to revs_agent_step_select as
 * sel_nblk; the dual-value terms come from pnq[2]) and clears its rows of zero_out (NULL, or
 * a double[m][T] array other than p, which every workgroup reads to the end).  counters:
 * uint32[(m + 31) / 32], zero before the first use, left zero.  T <= 32. */
int revs_op_dual_product_rows(int32_t m, int32_t T, const double *Rt, const double *p,
                              const double *pnq, const double *y, double vlo, double vhi,
                              int32_t ksplit,
                              double *v_slabs, double *vfull, double *viol, double *partial,
                              double *zero_out, uint32_t *counters, void *stream);
int revs_op_dual_model(int32_t m, int32_t T, const double *R, const double *n_free,
                       const int64_t *cand_idx, const int32_t *cand_cnt, const double *cand_val,
                       double kappa, double delta, int32_t max_pivots, int32_t nks,
                       double *k_slabs, double *k_full, double *yhat, int32_t *info,
                       void *stream);
/* One evaluation as a single host call: phase bit 0 = [R^T y into d_slabs when use_y,]
 * revs_op_dual_eval; phase bit 1 = R p into v_slabs (Rt = R^T, row-major) and
 * revs_op_dual_select -- product and row bookkeeping as ONE launch when tile_counters
 * (uint32[(m + 31) / 32], zero before the first use, left zero) is given and T <= 32, see
 * revs_op_dual_product_rows (`partial` then holds (m + 31) / 32 blocks);
 * with phase bit 2 (value 4) the selection kernel is left to revs_agent_step_select.
 * A driver that shards residences runs phase 1, all-reduces pnq, then phase 2. */
int revs_op_dual_evaluate(int32_t phase, int32_t m, int32_t T, const int64_t *node_ptr,
                          const float *p_est, const float *p_sch, const float *gamma,
                          const double *R, const double *Rt, const double *y, int32_t use_y,
                          double kappa, double vlo, double vhi, int32_t kadd, int32_t ksplit,
                          double *d_slabs, double *v_slabs, double *pnq, float *p_est_new,
                          double *vfull, double *viol, double *partial, int64_t *cand_idx,
                          int32_t *cand_cnt, double *cand_val, double *stats, double seq,
                          uint32_t *tile_counters, void *stream);
/* revs_op_dual_model for slots with at most 8 candidates each (the caller knows the counts
 * from the stats: rows with y != 0 plus the violated rows admitted), Gram matrix and pivoting
 * in one small kernel.  A slot with more than 8 candidates is left unmoved and flagged
 * info = -999.  k_full as in revs_op_dual_model (top-left block written). */
int revs_op_dual_model_small(int32_t m, int32_t T, const double *R, const double *n_free,
                             const int64_t *cand_idx, const int32_t *cand_cnt,
                             const double *cand_val, double kappa, double delta,
                             int32_t max_pivots, double *k_full, double *yhat, int32_t *info,
                             void *stream);
int revs_op_dual_step(int32_t T, const int64_t *cand_idx, const int32_t *cand_cnt,
                      const double *cand_val, const double *yhat, const double *alpha,
                      double *y_trial, double *lin_out, void *stream);
/* revs_op_dual_step with alpha_t decided on the device from the stats of the evaluation the
 * model was built on: alpha_t = 1 if stats_prev[8 t] / scale > eps (rows of slot t not yet
 * within tolerance), else 0 -- what a driver that had read those stats would pass.  Lets a
 * driver enqueue evaluation, model, step and the next evaluation without reading anything in
 * between (operator_newton.py: the binding steady state).  y (double[m][T]) != NULL: y_trial = y is
 * copied by the same launch before the candidates are written (else the caller has done so). */
int revs_op_dual_step_pending(int32_t T, const int64_t *cand_idx, const int32_t *cand_cnt,
                              const double *cand_val, const double *yhat,
                              const double *stats_prev, double scale, double eps,
                              const double *y, int32_t m,
                              double *y_trial, double *lin_out, void *stream);
/* The selection of revs_op_dual_select (after revs_op_dual_rows / _product_rows; arguments as
 * revs_agent_step_select's), revs_op_dual_model_small on the lists it builds and
 * revs_op_dual_step_pending (y_trial = y, then the full step in the slots whose rows this
 * selection finds beyond eps; lin_out as there) in ONE launch, one workgroup per slot.  A
 * slot with more than 8 candidates gets info = -999 (and an unchanged y_trial column), as
 * from revs_op_dual_model_small: the caller then runs the general model. */
int revs_op_dual_select_model_step(int32_t m, int32_t T, const double *sel_partial, int32_t sel_nblk,
                                   const double *y, double vlo, double vhi, int32_t kadd,
                                   const double *vfull, const double *viol, int64_t *cand_idx,
                                   int32_t *cand_cnt, double *cand_val, double *stats, double seq,
                                   const double *R, const double *n_free, double kappa, double delta,
                                   int32_t max_pivots, double *k_full, double *yhat, int32_t *info,
                                   double scale, double eps, double *y_trial, double *lin_out,
                                   void *stream);
/* ---- the same evaluation behind the tree form of R p (see "the feeder as a tree" below) ----
 * On a radial feeder v = R p is three prefix sums over the nodes in DFS preorder: one workgroup per
 * slot computes its slot's voltages in O(nodes) and judges its rows on the spot -- no 33 MB matrix
 * stream, no K-split slabs, one block of partial sums per slot (the selection then runs with
 * sel_nblk = 1).  Same outputs as revs_op_dual_rows / revs_op_dual_select (rows of nodes without
 * residences have no position in the tree: their v, violation and multiplier stay zero).
 * revs_op_dual_rows_tree: pnq = the node sums p | N | q of the home pass; zero_out: an array (not
 * pnq) cleared on the way, or NULL; with_select != 0: the candidate selection of every slot in the
 * same launch (else the caller runs it: revs_agent_step_select with sel_nblk = 1) -- the slot's rows then reach
 * the selection through LDS where 3 m doubles fit beside the tree's scan buffer (m <= ~4 000), and vfull / viol
 * are scratch of the call: NOT written.
 * revs_op_dual_evaluate_tree: revs_op_dual_evaluate with phase bit 1 done this way (phase bit 0 --
 * the product R^T y for the home pass's shifts, when use_y -- is unchanged).
 * revs_op_dual_tree_select_model_step: rows, selection, small model and step of every slot in ONE
 * launch (revs_op_dual_select_model_step with the rows in front; n_free = pnq + m T). */
int revs_op_dual_rows_tree(int32_t m, int32_t T, const revs_tree_t *tree_host, const double *pnq,
                           const double *y, double vlo, double vhi, int32_t kadd, double *vfull,
                           double *viol, double *partial, double *zero_out, int64_t *cand_idx,
                           int32_t *cand_cnt, double *cand_val, double *stats, double seq,
                           int32_t with_select, void *stream);
int revs_op_dual_evaluate_tree(int32_t phase, int32_t m, int32_t T, const int64_t *node_ptr,
                               const float *p_est, const float *p_sch, const float *gamma,
                               const double *R, const revs_tree_t *tree_host, const double *y,
                               int32_t use_y, double kappa, double vlo, double vhi, int32_t kadd,
                               int32_t ksplit, double *d_slabs, double *pnq, float *p_est_new,
                               double *vfull, double *viol, double *partial, int64_t *cand_idx,
                               int32_t *cand_cnt, double *cand_val, double *stats, double seq,
                               void *stream);
int revs_op_dual_tree_select_model_step(int32_t m, int32_t T, const revs_tree_t *tree_host,
                                        const double *pnq, const double *y, double vlo, double vhi,
                                        int32_t kadd, double *vfull, double *viol, double *partial,
                                        int64_t *cand_idx, int32_t *cand_cnt, double *cand_val,
                                        double *stats, double seq, const double *R, double kappa,
                                        double delta, int32_t max_pivots, double *k_full, double *yhat,
                                        int32_t *info, double scale, double eps, double *y_trial,
                                        double *lin_out, void *stream);
/* Host only (no GPU work): the acceptance test of such a chained iteration on the two stats
 * blocks (double[T][8], as revs_op_dual_select writes them; s1[8 t + 4] = the step kernel's
 * linear term) -- returns 1 iff the driver's own checks (operator_newton.py:
 * AdmmEngine._operator_solve_newton) would find: evaluation 0 not yet within eps, at most 8
 * candidates per slot (the small model), the row-wise/dense choice `chain_few` right, the
 * Armijo test passed by the full step in every pending slot, and evaluation 1 within eps.
 * Then *nsup_sum / *nsup_max = total / largest number of multipliers per slot in s1.  Any
 * other outcome returns 0 and is left to the driver's general loop. */
int revs_newton_chain_accept(int32_t T, const double *s0, const double *s1, double scale, double eps,
                             int32_t amax, int32_t kadd, int32_t chain_few, int32_t *nsup_sum,
                             int32_t *nsup_max);

/* ---- the model problem beyond REVS_DUAL_AMAX rows per slot (csrc/newton_big.hip) ------------------------------
 * The dual Newton path above holds REVS_DUAL_AMAX = 128 candidate rows per slot (its factor lives in LDS).  The reference
 * hands Gurobi every row (lpsolver.py:183-194); a slot with 129 ... REVS_DUAL_AMAX_BIG binding rows stays on the Newton
 * path through these three launches -- lists, Gram slabs, Hessian and factor in global memory (cand_idx
 * int64[T][BIG], cand_cnt int32[T] (-1: more rows with a multiplier than BIG), cand_val double[T][3][BIG] = sign |
 * gradient | current multiplier, k_slabs double[T][nks][BIG][BIG], k_full / l_factor double[T][BIG][BIG] each,
 * yhat double[T][BIG], info int32[T]: pivoting rounds, negative when the limit was hit, -998 for a slot flagged -1):
 *   revs_op_dual_select_big  the rows with y != 0 in row order + the `kadd` most violated rows without one (ties to the
 *                            lower row), from the multipliers and the row arrays vfull / viol ([m][T]) an evaluation by
 *                            the dense path (revs_op_dual_evaluate) left
 *   revs_op_dual_model_big   K_t = R_F N_t R_F^T / kappa and the LCP u >= 0, K'u - c >= 0, u.(K'u - c) = 0 by block
 *                            principal pivoting (same regularisation, tolerances and rule as revs_op_dual_model)
 *   revs_op_dual_step_big    y_trial = y, then y + alpha[t] (yhat - y) on slot t's listed rows; lin_out[8 t] = gradient . step */
#define REVS_DUAL_AMAX_BIG 512
int revs_op_dual_select_big(int32_t m, int32_t T, const double *y, const double *vfull, const double *viol, double vlo,
                            double vhi, int32_t kadd, int64_t *cand_idx, int32_t *cand_cnt, double *cand_val, void *stream);
int revs_op_dual_model_big(int32_t m, int32_t T, const double *R, const double *n_free, const int64_t *cand_idx,
                           const int32_t *cand_cnt, const double *cand_val, double kappa, double delta, int32_t max_pivots,
                           int32_t nks, double *k_slabs, double *k_full, double *l_factor, double *yhat, int32_t *info,
                           void *stream);
int revs_op_dual_step_big(int32_t T, const int64_t *cand_idx, const int32_t *cand_cnt, const double *cand_val,
                          const double *yhat, const double *alpha, const double *y, int32_t m, double *y_trial,
                          double *lin_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* REVS_ADMM_OPS_H */
