// Operator ("Utility") side of one ADMM iteration by ADMM in OSQP form -- gfx950, double
// precision.  FALLBACK: the default path is the dual Newton solver of newton_kernels.hip;
// these kernels run under OperatorOptions(solver="admm") or when a Newton solve hands an
// iteration over (more than 128 binding rows in a slot, ...).
//
// Reference: class Utility, lpsolver.py:163-238.  The operator's problem
//     min  sum_i (kappa/2)|g_i|^2 + g_i . a_i           a_i = gamma_i - (kappa/2)(P_est_i + P_sch_i)
//     s.t. g >= 0 (Gurobi default lb),  vlo <= R_res g[:,t] <= vhi  for every slot t
// is the projection of g0 = -a/kappa onto the voltage-feasible set.  Here it is
// solved by ADMM in OSQP form (x, z = Cx, y) on C = [C_v ; I]:
//     A~  = diag(1/sqrt(n_m)) A      (A = home->node aggregation, n_m homes on node m)
//     C_v = D^1/2 R D^1/2 A~         (voltage row m scaled by sqrt(n_m), bounds too)
//     D^1/2 R D^1/2 = Q L Q^T        symmetric PSD, eigendecomposed once on the host
// so that (kappa I + rho_b I + rho_v C_v^T C_v)^-1 is applied through Q and L: per-slot
// rho can change without refactoring anything.  One inner iteration of the GENERAL path =
//     home pass   (this file)   node update for the workgroup's own node row (fused),
//                               then z_b / y_b of every home of that node; form rhs and
//                               aggregate it to the node
//     2 products  (gemm_kernels.hip)   Q^T [rhat | w]  and  Q [a | l a]
//     1 node pass (this file)   a = (ta + l tb) / (c + rho_v l^2)
// With homes sharded over GPUs its only exchange is the all-reduce of rhat (m x T
// doubles) after the home pass.  The NODE-SPACE FAST PATH further down needs no home
// pass inside the loop at all.
//
// State kept per home and slot is ONE double: s_b = z_b + y_b.  The bound rows'
// z_b = max(u,0) and y_b = rho_b min(u,0) are complementary (one of them is zero), so
// z_b = max(s_b,0), y_b = min(s_b,0) decode it exactly; sigma = 0 and Boyd-style
// over-relaxation (z/y use alpha xt + (1-alpha) z, x itself is not stored) remove the
// x array.  The home pass therefore moves 3 arrays (read s_b, g0; write s_b), not 7.
//
// All arrays are [rows][T] row-major doubles; one thread per (row, slot).
#include "common.h"

namespace revs {

__device__ __forceinline__ void atomic_max_nonneg(double *addr, double v) {
    // non-negative IEEE doubles order like their bit patterns
    atomicMax(reinterpret_cast<unsigned long long *>(addr),
              (unsigned long long)__double_as_longlong(v));
}

// g0 = (P_est + P_sch)/2 - G/kappa      (lpsolver.py:202-204: g0 = -a/kappa)
__global__ void op_g0_kernel(int64_t total, const float *pe, const float *ps, const float *gm,
                             double kappa, double *g0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) g0[i] = (double)revs_g0f(pe[i], ps[i], gm[i], 1.0f / (float)kappa);
}

// cold start: s_b = max(g0, 0)  (z_b = that, y_b = 0)
__global__ void op_init_home_kernel(int64_t total, const double *g0, double *sb) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) sb[i] = fmax(g0[i], 0.0);
}
// cold start, node side: z_v = clip(Cx), y_v = 0, w = rho_v z_v
// (row m of the voltage block is scaled by bscale[m] = sqrt(n_m), and so are its bounds)
__global__ void op_init_node_kernel(int total, int T, const double *cx, const double *rho_v,
                                    const double *bscale, double vlo, double vhi, double *zv,
                                    double *yv, double *w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const double bs = bscale ? bscale[i / T] : 1.0;
        const double z = fmin(fmax(cx[i], bs * vlo), bs * vhi);
        zv[i] = z; yv[i] = 0.0; w[i] = rho_v[i % T] * z;
    }
}

// out[node][t] = scale[node] * sum_{homes of node} in[home][t]
template <typename T>
__global__ void aggregate_kernel(int m, int Ts, const int64_t *node_ptr, const T *in,
                                 const T *scale, T *out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * Ts) return;
    const int node = idx / Ts, t = idx - node * Ts;
    T acc = 0;
    for (int64_t i = node_ptr[node]; i < node_ptr[node + 1]; ++i) acc += in[i * Ts + t];
    out[idx] = scale ? scale[node] * acc : acc;
}

// Home pass.  For node m, slot t (c = kappa + rho_b[t]), every home of the node:
//   z = max(s_b,0), y = min(s_b,0)
//   if xc:  rhs = kappa g0 + rho_b z - y                 (state before the update)
//           xt  = rhs / c + inv_sqrt_n[m] xc[m]          (the x-update of ADMM)
//           h   = alpha xt + (1-alpha) z ;  u = h + y / rho_b
//           z   = max(u,0) ;  y = rho_b min(u,0) ;  s_b = z + y
//           if res: per-slot maxima of |xt - z|, |kappa (xt - g0) + isn cty[m] + y|,
//                   |xt|, |isn cty[m] + y|, |kappa g0|  -> res rows 1,2,5,6,7
//   rhat[m] = inv_sqrt_n[m] * sum_homes (kappa g0 + rho_b z - y)        (new state)
//
// Mapping: one 256-thread workgroup per node; TL = pow2 >= T lanes walk the slots
// (contiguous doubles of one home), HS = 256/TL "home lanes" stride over the node's
// homes, so a node with 49 homes keeps 8 loads per array in flight per slot lane instead
// of one; the HS partial sums are combined through LDS in a fixed order (reproducible).
// With `nu` (node-update operands) the workgroup first performs op_node_update for ITS
// node row -- xc, z_v, y_v, w from the products va, usa -- and keeps xc in LDS: the node
// pass and its launch disappear from the iteration.
struct NodeUpd {
    const double *va, *usa, *rho_v, *bscale;
    double *zv, *yv, *w, *xc_out;
    double vlo, vhi;
    int nslab;
};

template <int TL>
__global__ __launch_bounds__(256) void op_home_pass_kernel(
        int m, int T, const int64_t *__restrict__ node_ptr, const double *__restrict__ inv_sqrt_n,
        double *__restrict__ sb, const double *__restrict__ g0, const double *__restrict__ xc,
        const double *__restrict__ rho_b, double kappa, double alpha, double *__restrict__ rhat,
        const double *__restrict__ cty, double *__restrict__ res, const NodeUpd nu) {
    constexpr int HS = 256 / TL;
    const int node = blockIdx.x;
    const int t = threadIdx.x % TL, hs = threadIdx.x / TL;
    const bool tok = t < T;
    const int tc = tok ? t : 0;
    const int idx = node * T + tc;
    const double rb = rho_b[tc];
    const double inv_c = 1.0 / (kappa + rb), inv_rb = 1.0 / rb;
    const double isn = inv_sqrt_n[node];
    __shared__ double xc_s[TL];
    if (nu.va) {
        if (hs == 0 && tok) {
            const int64_t total = (int64_t)m * T;
            double vav = nu.va[idx], zt = nu.usa[idx];
            for (int q = 1; q < nu.nslab; ++q) { vav += nu.va[idx + q * total]; zt += nu.usa[idx + q * total]; }
            const double rv = nu.rho_v[t];
            const double xcv = vav - rhat[idx] * inv_c;
            const double h = alpha * zt + (1.0 - alpha) * nu.zv[idx];
            double y = nu.yv[idx];
            const double bs = nu.bscale ? nu.bscale[node] : 1.0;
            const double zn = fmin(fmax(h + y / rv, bs * nu.vlo), bs * nu.vhi);
            y += rv * (h - zn);
            nu.zv[idx] = zn;
            nu.yv[idx] = y;
            nu.w[idx] = rv * zn - y;
            nu.xc_out[idx] = xcv;
            xc_s[t] = xcv;
        }
        __syncthreads();
    }
    const double corr = nu.va ? isn * xc_s[tc] : (xc ? isn * xc[idx] : 0.0);
    const bool upd = nu.va || xc;
    const double ct = res ? isn * cty[idx] : 0.0;
    double acc = 0.0, r1 = 0, r2 = 0, r5 = 0, r6 = 0, r7 = 0;
    const int64_t i0 = node_ptr[node], i1 = node_ptr[node + 1];
    if (tok) {
        for (int64_t i = i0 + hs; i < i1; i += HS) {
            const int64_t o = i * T + t;
            const double sv = sb[o];
            const double kg = kappa * g0[o];
            double z = fmax(sv, 0.0), y = fmin(sv, 0.0);
            if (upd) {
                const double xt = (kg + rb * z - y) * inv_c + corr;
                const double u = alpha * xt + (1.0 - alpha) * z + y * inv_rb;
                z = fmax(u, 0.0);
                y = rb * fmin(u, 0.0);
                sb[o] = z + y;
                if (res) {
                    const double cy = ct + y;
                    r1 = fmax(r1, fabs(xt - z));
                    r2 = fmax(r2, fabs(kappa * xt - kg + cy));
                    r5 = fmax(r5, fabs(xt));
                    r6 = fmax(r6, fabs(cy));
                    r7 = fmax(r7, fabs(kg));
                }
            }
            acc += kg + rb * z - y;
        }
    }
    __shared__ double red[6][HS][TL];
    red[0][hs][t] = acc;
    if (res) { red[1][hs][t] = r1; red[2][hs][t] = r2; red[3][hs][t] = r5; red[4][hs][t] = r6; red[5][hs][t] = r7; }
    __syncthreads();
    if (hs == 0 && tok) {
        double a = red[0][0][t];
#pragma unroll
        for (int k = 1; k < HS; ++k) a += red[0][k][t];
        rhat[idx] = isn * a;
        if (res && upd) {
            double q1 = 0, q2 = 0, q5 = 0, q6 = 0, q7 = 0;
#pragma unroll
            for (int k = 0; k < HS; ++k) {
                q1 = fmax(q1, red[1][k][t]); q2 = fmax(q2, red[2][k][t]); q5 = fmax(q5, red[3][k][t]);
                q6 = fmax(q6, red[4][k][t]); q7 = fmax(q7, red[5][k][t]);
            }
            atomic_max_nonneg(res + 1 * T + t, q1);
            atomic_max_nonneg(res + 2 * T + t, q2);
            atomic_max_nonneg(res + 5 * T + t, q5);
            atomic_max_nonneg(res + 6 * T + t, q6);
            atomic_max_nonneg(res + 7 * T + t, q7);
        }
    }
}

// out = s (row scale) * in
__global__ void op_row_scale_kernel(int total, int T, const double *s, const double *in,
                                    double *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) out[i] = s[i / T] * in[i];
}

// w = rho_v z_v - y_v
__global__ void op_node_w_kernel(int total, int T, const double *zv, const double *yv,
                                 const double *rho_v, double *w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) w[i] = rho_v[i % T] * zv[i] - yv[i];
}

// t1 = ta + s tb ;  a = t1 / (c + rho_v s^2) ;  sa = s a        (row j <-> singular value s_j)
// ta, tb arrive as `nslab` K-split partial products (slab stride = total): summed here
__global__ void op_node_scale_kernel(int total, int T, int nslab, const double *ta,
                                     const double *tb, const double *s, const double *rho_v,
                                     const double *rho_b, double kappa, double *a, double *sa) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = i / T, t = i - j * T;
    const double sj = s[j];
    const double c = kappa + rho_b[t];
    double tav = ta[i], tbv = tb[i];
    for (int q = 1; q < nslab; ++q) { tav += ta[i + (int64_t)q * total]; tbv += tb[i + (int64_t)q * total]; }
    const double av = (tav + sj * tbv) / (c + rho_v[t] * sj * sj);
    a[i] = av;
    sa[i] = sj * av;
}

// xc = va - rhat / c ;  h = alpha usa + (1-alpha) z_v ;  z_v = clip(h + y_v/rho_v) ;
// y_v += rho_v (h - z_v) ;  w = rho_v z_v - y_v ;  usa = C_v xt.
// if res: per-slot maxima |usa - z_v|, |usa|, |z_v| -> res rows 0, 3, 4
__global__ void op_node_update_kernel(int total, int T, int nslab, const double *va,
                                      const double *rhat, const double *usa, const double *rho_v,
                                      const double *rho_b, const double *bscale, double kappa,
                                      double alpha, double vlo, double vhi, double *xc,
                                      double *zv, double *yv, double *w, double *res) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // residual maxima are merged per workgroup in LDS first (49k threads hammering 3 x T
    // global words cost 45 us), then one global atomic per (workgroup, slot)
    __shared__ unsigned long long smax[3][256];
    if (res) {
        smax[0][threadIdx.x] = 0; smax[1][threadIdx.x] = 0; smax[2][threadIdx.x] = 0;
        __syncthreads();
    }
    const bool live = i < total;
    const int t = live ? i % T : 0;
    double r0 = 0, r3 = 0, r4 = 0;
    if (live) {
        const double rv = rho_v[t];
        const double c = kappa + rho_b[t];
        double vav = va[i], zt = usa[i];
        for (int q = 1; q < nslab; ++q) { vav += va[i + (int64_t)q * total]; zt += usa[i + (int64_t)q * total]; }
        xc[i] = vav - rhat[i] / c;
        const double h = alpha * zt + (1.0 - alpha) * zv[i];
        double y = yv[i];
        const double bs = bscale ? bscale[i / T] : 1.0;
        const double zn = fmin(fmax(h + y / rv, bs * vlo), bs * vhi);
        y += rv * (h - zn);
        zv[i] = zn;
        yv[i] = y;
        w[i] = rv * zn - y;
        r0 = fabs(zt - zn); r3 = fabs(zt); r4 = fabs(zn);
    }
    if (res) {
        const int slot = t % 256;
        if (live) {
            atomicMax(&smax[0][slot], (unsigned long long)__double_as_longlong(r0));
            atomicMax(&smax[1][slot], (unsigned long long)__double_as_longlong(r3));
            atomicMax(&smax[2][slot], (unsigned long long)__double_as_longlong(r4));
        }
        __syncthreads();
        const int tt = threadIdx.x;
        if (tt < T && tt < 256) {
            atomicMax(reinterpret_cast<unsigned long long *>(res + 0 * T + tt), smax[0][tt]);
            atomicMax(reinterpret_cast<unsigned long long *>(res + 3 * T + tt), smax[1][tt]);
            atomicMax(reinterpret_cast<unsigned long long *>(res + 4 * T + tt), smax[2][tt]);
        }
    }
}

// P_est = z_b = max(s_b, 0), as float
__global__ void op_export_kernel(int64_t total, const double *sb, float *pe) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) pe[i] = (float)fmax(sb[i], 0.0);
}

// ---------------------------------------------------------------------------
// Node-space fast path.  While no residence is pushed to g = 0, the operator QP
// collapses to the constrained nodes: g = g0 + A~^T d with node correction d (M x T),
// cost (kappa/2)|d|^2, rows blo <= Rs (p0 + d) <= bhi, p0 = A~ g0.  ADMM on (x = p0 + d,
// z = Rs x) in the eigenbasis of Rs = Q L Q^T:
//     xh  = (kappa ph0 + l wh) / (kappa + rho l^2)        wh = Q^T (rho z - y)
//     zt  = Q (l xh)
//     z   = clip(alpha zt + (1-alpha) z + y/rho), y += rho (alpha zt + (1-alpha) z_old - z)
// Two T-column products per iteration, no home-space traffic and -- with residences
// sharded -- no communication: the only exchange per OUTER iteration is the all-reduce
// of p0 (sum) and gmin (min).  Afterwards min_i g0_i + isn d >= 0 is checked per node;
// if a clamp would be active the caller falls back to the general home-space ADMM.

// p0[m][t] = isn[m] * sum_i g0_i[t],  gmin[m][t] = min_i g0_i[t],  g0 from the float state.
// g0_out (or NULL) also stores g0 in double for the general path.
template <int TL>
__global__ __launch_bounds__(256) void op_node_prep_kernel(
        int m, int T, const int64_t *__restrict__ node_ptr, const double *__restrict__ inv_sqrt_n,
        const float *__restrict__ pe, const float *__restrict__ ps, const float *__restrict__ gm,
        double kappa, int preclamp, double *__restrict__ p0, double *__restrict__ gmin,
        double *__restrict__ g0_out) {
    constexpr int HS = 256 / TL;
    const int node = blockIdx.x;
    const int t = threadIdx.x % TL, hs = threadIdx.x / TL;
    const bool tok = t < T;
    double acc = 0.0, mn = INFINITY;
    const int64_t i0 = node_ptr[node], i1 = node_ptr[node + 1];
    const float inv_kf = 1.0f / (float)kappa;
    if (tok) {
        for (int64_t i = i0 + hs; i < i1; i += HS) {
            const int64_t o = i * T + t;
            double g = (double)revs_g0f(pe[o], ps[o], gm[o], inv_kf);
            if (g0_out) g0_out[o] = g;
            // R >= 0 entrywise and vlo <= 0: only upper rows can bind, the node shift is
            // <= 0, so max(g0 - theta, 0) = max(max(g0,0) - theta, 0): residences with
            // g0 < 0 are at zero whatever the voltage rows do (exact presolve)
            if (preclamp) g = fmax(g, 0.0);
            acc += g;
            mn = fmin(mn, g);
        }
    }
    __shared__ double red[2][HS][TL];
    red[0][hs][t] = acc;
    red[1][hs][t] = mn;
    __syncthreads();
    if (hs == 0 && tok) {
        double a = red[0][0][t], b = red[1][0][t];
#pragma unroll
        for (int k = 1; k < HS; ++k) { a += red[0][k][t]; b = fmin(b, red[1][k][t]); }
        p0[node * T + t] = inv_sqrt_n[node] * a;
        gmin[node * T + t] = b;
    }
}

// xh = (kappa ph0 + l wh) / (kappa + rho_v l^2) ;  sx = l xh      (wh as nslab slabs)
__global__ void op_nodefast_scale_kernel(int total, int T, int nslab, const double *wh,
                                         const double *ph0, const double *lam,
                                         const double *rho_v, double kappa, double *xh,
                                         double *sx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = i / T, t = i - j * T;
    const double l = lam[j];
    double w = wh[i];
    for (int q = 1; q < nslab; ++q) w += wh[i + (int64_t)q * total];
    const double x = (kappa * ph0[i] + l * w) / (kappa + rho_v[t] * l * l);
    xh[i] = x;
    sx[i] = l * x;
}

// z / y / w update from zt = Q (l xh) (nslab slabs); with res: rows 0,3,4 = max|zt - z|, |zt|, |z|
__global__ void op_nodefast_update_kernel(int total, int T, int nslab, const double *zt_s,
                                          const double *rho_v, const double *bscale,
                                          double alpha, double vlo, double vhi, double *zv,
                                          double *yv, double *w, double *res) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ unsigned long long smax[3][256];
    if (res) {
        smax[0][threadIdx.x] = 0; smax[1][threadIdx.x] = 0; smax[2][threadIdx.x] = 0;
        __syncthreads();
    }
    const bool live = i < total;
    const int t = live ? i % T : 0;
    double r0 = 0, r3 = 0, r4 = 0;
    if (live) {
        double zt = zt_s[i];
        for (int q = 1; q < nslab; ++q) zt += zt_s[i + (int64_t)q * total];
        const double rv = rho_v[t];
        const double h = alpha * zt + (1.0 - alpha) * zv[i];
        double y = yv[i];
        const double bs = bscale ? bscale[i / T] : 1.0;
        const double zn = fmin(fmax(h + y / rv, bs * vlo), bs * vhi);
        y += rv * (h - zn);
        zv[i] = zn;
        yv[i] = y;
        w[i] = rv * zn - y;
        r0 = fabs(zt - zn); r3 = fabs(zt); r4 = fabs(zn);
    }
    if (res) {
        if (live) {
            atomicMax(&smax[0][t], (unsigned long long)__double_as_longlong(r0));
            atomicMax(&smax[1][t], (unsigned long long)__double_as_longlong(r3));
            atomicMax(&smax[2][t], (unsigned long long)__double_as_longlong(r4));
        }
        __syncthreads();
        const int tt = threadIdx.x;
        if (tt < T) {
            atomicMax(reinterpret_cast<unsigned long long *>(res + 0 * T + tt), smax[0][tt]);
            atomicMax(reinterpret_cast<unsigned long long *>(res + 3 * T + tt), smax[1][tt]);
            atomicMax(reinterpret_cast<unsigned long long *>(res + 4 * T + tt), smax[2][tt]);
        }
    }
}

// dual residual in the eigenbasis: rows 2,5,6,7 = max |kappa (xh - ph0) + l yh|, kappa|xh|,
// |l yh|, kappa|ph0|   (yh = Q^T y as nslab slabs).  Row 1 (bound rows) stays 0.
__global__ void op_nodefast_dualres_kernel(int total, int T, int nslab, const double *xh,
                                           const double *ph0, const double *lam,
                                           const double *yh_s, double kappa, double *res) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ unsigned long long smax[4][256];
    smax[0][threadIdx.x] = 0; smax[1][threadIdx.x] = 0; smax[2][threadIdx.x] = 0; smax[3][threadIdx.x] = 0;
    __syncthreads();
    const bool live = i < total;
    const int t = live ? i % T : 0;
    if (live) {
        double yh = yh_s[i];
        for (int q = 1; q < nslab; ++q) yh += yh_s[i + (int64_t)q * total];
        const double ly = lam[i / T] * yh;
        atomicMax(&smax[0][t], (unsigned long long)__double_as_longlong(fabs(kappa * (xh[i] - ph0[i]) + ly)));
        atomicMax(&smax[1][t], (unsigned long long)__double_as_longlong(fabs(xh[i])));
        atomicMax(&smax[2][t], (unsigned long long)__double_as_longlong(fabs(ly)));
        atomicMax(&smax[3][t], (unsigned long long)__double_as_longlong(fabs(kappa * ph0[i])));
    }
    __syncthreads();
    const int tt = threadIdx.x;
    if (tt < T) {
        atomicMax(reinterpret_cast<unsigned long long *>(res + 2 * T + tt), smax[0][tt]);
        atomicMax(reinterpret_cast<unsigned long long *>(res + 5 * T + tt), smax[1][tt]);
        atomicMax(reinterpret_cast<unsigned long long *>(res + 6 * T + tt), smax[2][tt]);
        atomicMax(reinterpret_cast<unsigned long long *>(res + 7 * T + tt), smax[3][tt]);
    }
}

// v0 = Rs p0 (nslab slabs) against the scaled bounds: stats[0] = max over rows of the
// violation max(v0 - b vhi, b vlo - v0, 0); cx = v0.  Zero violation <=> d = 0 is optimal.
// stats[1] = max(0, -min gmin): > 0 <=> some residence has g0 < 0 (a clamp is certain).
__global__ void op_nodefast_feas_kernel(int total, int T, int nslab, const double *v_s,
                                        const double *bscale, const double *gmin, double vlo,
                                        double vhi, double *cx, double *stats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double viol = 0.0, neg = 0.0;
    if (i < total) {
        double v = v_s[i];
        for (int q = 1; q < nslab; ++q) v += v_s[i + (int64_t)q * total];
        cx[i] = v;
        const double bs = bscale ? bscale[i / T] : 1.0;
        viol = fmax(fmax(v - bs * vhi, bs * vlo - v), 0.0);
        neg = fmax(-gmin[i], 0.0);
    }
    for (int o = 32; o >= 1; o >>= 1) {
        viol = fmax(viol, __shfl_xor(viol, o, 64));
        neg = fmax(neg, __shfl_xor(neg, o, 64));
    }
    __shared__ double red[2][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = viol; red[1][threadIdx.x >> 6] = neg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        viol = fmax(fmax(red[0][0], red[0][1]), fmax(red[0][2], red[0][3]));
        neg = fmax(fmax(red[1][0], red[1][1]), fmax(red[1][2], red[1][3]));
        if (viol > 0.0) atomic_max_nonneg(stats, viol);
        if (neg > 0.0) atomic_max_nonneg(stats + 1, neg);
    }
}

// d = x - p0 (x as nslab slabs of Q xh); slack[m][t] = gmin + isn d  (>= 0 <=> no clamp)
// stats[0] = max(0, -min slack), stats[1] = max |p0|   (stats zeroed by the caller)
__global__ void op_nodefast_finish_kernel(int total, int T, int nslab, const double *x_s,
                                          const double *p0, const double *gmin,
                                          const double *inv_sqrt_n, double *d, double *slack,
                                          double *stats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double viol = 0.0, pabs = 0.0;
    if (i < total) {
        double x = x_s[i];
        for (int q = 1; q < nslab; ++q) x += x_s[i + (int64_t)q * total];
        const double dv = x - p0[i];
        d[i] = dv;
        const double sl = gmin[i] + inv_sqrt_n[i / T] * dv;
        slack[i] = sl;
        viol = fmax(-sl, 0.0);
        pabs = fabs(p0[i]);
    }
    for (int o = 32; o >= 1; o >>= 1) {
        viol = fmax(viol, __shfl_xor(viol, o, 64));
        pabs = fmax(pabs, __shfl_xor(pabs, o, 64));
    }
    __shared__ double red[2][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = viol; red[1][threadIdx.x >> 6] = pabs; }
    __syncthreads();
    if (threadIdx.x == 0) {       // one pair of global atomics per workgroup, and only if needed
        viol = fmax(fmax(red[0][0], red[0][1]), fmax(red[0][2], red[0][3]));
        pabs = fmax(fmax(red[1][0], red[1][1]), fmax(red[1][2], red[1][3]));
        if (viol > 0.0) atomic_max_nonneg(stats + 0, viol);
        atomic_max_nonneg(stats + 1, pabs);
    }
}

// P_est_i = max(g0_i + isn[m] d[m], 0) as float, g0 recomputed from the float state
template <int TL>
__global__ __launch_bounds__(256) void op_node_apply_kernel(
        int m, int T, const int64_t *__restrict__ node_ptr, const double *__restrict__ inv_sqrt_n,
        const float *__restrict__ pe, const float *__restrict__ ps, const float *__restrict__ gm,
        double kappa, int preclamp, const double *__restrict__ d, float *__restrict__ pe_new) {
    constexpr int HS = 256 / TL;
    const int node = blockIdx.x;
    const int t = threadIdx.x % TL, hs = threadIdx.x / TL;
    if (t >= T) return;
    const double corr = inv_sqrt_n[node] * d[node * T + t];
    const float inv_kf = 1.0f / (float)kappa;
    const int64_t i0 = node_ptr[node], i1 = node_ptr[node + 1];
    for (int64_t i = i0 + hs; i < i1; i += HS) {
        const int64_t o = i * T + t;
        double g = (double)revs_g0f(pe[o], ps[o], gm[o], inv_kf);
        if (preclamp) g = fmax(g, 0.0);
        pe_new[o] = (float)fmax(g + corr, 0.0);
    }
}

static inline dim3 grid1(int64_t total) { return dim3((unsigned)((total + 255) / 256)); }

}  // namespace revs

using namespace revs;
#define S_(stream) ((hipStream_t)(stream))

extern "C" int revs_op_g0(int64_t n_homes, int32_t T, const float *p_est, const float *p_sch,
                          const float *gamma, float kappa, double *g0, void *stream) {
    REVS_REQUIRE(n_homes > 0 && T > 0 && p_est && p_sch && gamma && g0 && kappa > 0,
                 "revs_op_g0: bad argument");
    const int64_t total = n_homes * T;
    hipLaunchKernelGGL(op_g0_kernel, grid1(total), dim3(256), 0, S_(stream), total, p_est, p_sch,
                       gamma, (double)kappa, g0);
    REVS_CHECK_LAUNCH("revs_op_g0");
    return REVS_OK;
}

extern "C" int revs_op_init_home(int64_t n_homes, int32_t T, const double *g0, double *sb,
                                 void *stream) {
    REVS_REQUIRE(n_homes > 0 && T > 0 && g0 && sb, "revs_op_init_home: bad argument");
    const int64_t total = n_homes * T;
    hipLaunchKernelGGL(op_init_home_kernel, grid1(total), dim3(256), 0, S_(stream), total, g0, sb);
    REVS_CHECK_LAUNCH("revs_op_init_home");
    return REVS_OK;
}

extern "C" int revs_op_init_node(int32_t m, int32_t T, const double *cx, const double *rho_v,
                                 const double *bound_scale, double vlo, double vhi, double *zv,
                                 double *yv, double *w, void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && cx && rho_v && zv && yv && w, "revs_op_init_node: bad argument");
    hipLaunchKernelGGL(op_init_node_kernel, grid1((int64_t)m * T), dim3(256), 0, S_(stream),
                       m * T, T, cx, rho_v, bound_scale, vlo, vhi, zv, yv, w);
    REVS_CHECK_LAUNCH("revs_op_init_node");
    return REVS_OK;
}

extern "C" int revs_aggregate_f64(int32_t m, int32_t T, const int64_t *node_ptr,
                                  const double *in_home, const double *scale, double *out_node,
                                  void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && node_ptr && in_home && out_node, "revs_aggregate_f64: bad argument");
    hipLaunchKernelGGL((aggregate_kernel<double>), grid1((int64_t)m * T), dim3(256), 0, S_(stream),
                       m, T, node_ptr, in_home, scale, out_node);
    REVS_CHECK_LAUNCH("revs_aggregate_f64");
    return REVS_OK;
}

extern "C" int revs_aggregate_f32(int32_t m, int32_t T, const int64_t *node_ptr,
                                  const float *in_home, float *out_node, void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && node_ptr && in_home && out_node, "revs_aggregate_f32: bad argument");
    hipLaunchKernelGGL((aggregate_kernel<float>), grid1((int64_t)m * T), dim3(256), 0, S_(stream),
                       m, T, node_ptr, in_home, (const float *)nullptr, out_node);
    REVS_CHECK_LAUNCH("revs_aggregate_f32");
    return REVS_OK;
}

extern "C" int revs_op_home_pass(int32_t m, int32_t T, const int64_t *node_ptr,
                                 const double *inv_sqrt_n, double *sb, const double *g0,
                                 const double *xc, const double *rho_b, double kappa,
                                 double alpha, double *rhat, const double *cty_node,
                                 double *res, void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && node_ptr && inv_sqrt_n && sb && g0 && rho_b && rhat,
                 "revs_op_home_pass: bad argument");
    REVS_REQUIRE(!res || (cty_node && xc), "revs_op_home_pass: res needs cty_node and xc");
    REVS_REQUIRE(T <= 256, "revs_op_home_pass: T=%d exceeds 256", T);
    const NodeUpd nu{};
#define HP(TL)                                                                                  \
    hipLaunchKernelGGL((op_home_pass_kernel<TL>), dim3(m), dim3(256), 0, S_(stream), m, T,      \
                       node_ptr, inv_sqrt_n, sb, g0, xc, rho_b, kappa, alpha, rhat, cty_node, res, nu)
    if (T <= 32) HP(32);
    else if (T <= 64) HP(64);
    else if (T <= 128) HP(128);
    else HP(256);
#undef HP
    REVS_CHECK_LAUNCH("revs_op_home_pass");
    return REVS_OK;
}

extern "C" int revs_op_home_pass_fused(int32_t m, int32_t T, const int64_t *node_ptr,
                                       const double *inv_sqrt_n, double *sb, const double *g0,
                                       const double *rho_b, double kappa, double alpha,
                                       double *rhat, int32_t nslab, const double *va,
                                       const double *usa, const double *rho_v,
                                       const double *bound_scale, double vlo, double vhi,
                                       double *xc, double *zv, double *yv, double *w,
                                       void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && T <= 256 && node_ptr && inv_sqrt_n && sb && g0 && rho_b && rhat,
                 "revs_op_home_pass_fused: bad argument");
    REVS_REQUIRE(nslab >= 1 && va && usa && rho_v && xc && zv && yv && w && vlo <= vhi,
                 "revs_op_home_pass_fused: bad node-update argument");
    const NodeUpd nu{va, usa, rho_v, bound_scale, zv, yv, w, xc, vlo, vhi, nslab};
#define HP(TL)                                                                                  \
    hipLaunchKernelGGL((op_home_pass_kernel<TL>), dim3(m), dim3(256), 0, S_(stream), m, T,      \
                       node_ptr, inv_sqrt_n, sb, g0, (const double *)nullptr, rho_b, kappa,     \
                       alpha, rhat, (const double *)nullptr, (double *)nullptr, nu)
    if (T <= 32) HP(32);
    else if (T <= 64) HP(64);
    else if (T <= 128) HP(128);
    else HP(256);
#undef HP
    REVS_CHECK_LAUNCH("revs_op_home_pass_fused");
    return REVS_OK;
}

extern "C" int revs_op_row_scale(int32_t m, int32_t T, const double *s, const double *in,
                                 double *out, void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && s && in && out, "revs_op_row_scale: bad argument");
    hipLaunchKernelGGL(op_row_scale_kernel, grid1((int64_t)m * T), dim3(256), 0, S_(stream),
                       m * T, T, s, in, out);
    REVS_CHECK_LAUNCH("revs_op_row_scale");
    return REVS_OK;
}

extern "C" int revs_op_node_w(int32_t m, int32_t T, const double *zv, const double *yv,
                              const double *rho_v, double *w, void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && zv && yv && rho_v && w, "revs_op_node_w: bad argument");
    hipLaunchKernelGGL(op_node_w_kernel, grid1((int64_t)m * T), dim3(256), 0, S_(stream), m * T, T,
                       zv, yv, rho_v, w);
    REVS_CHECK_LAUNCH("revs_op_node_w");
    return REVS_OK;
}

extern "C" int revs_op_node_scale(int32_t m, int32_t T, int32_t nslab, const double *ta,
                                  const double *tb, const double *s, const double *rho_v,
                                  const double *rho_b, double kappa, double *a, double *sa,
                                  void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && nslab >= 1 && ta && tb && s && rho_v && rho_b && a && sa,
                 "revs_op_node_scale: bad argument");
    hipLaunchKernelGGL(op_node_scale_kernel, grid1((int64_t)m * T), dim3(256), 0, S_(stream),
                       m * T, T, nslab, ta, tb, s, rho_v, rho_b, kappa, a, sa);
    REVS_CHECK_LAUNCH("revs_op_node_scale");
    return REVS_OK;
}

extern "C" int revs_op_node_update(int32_t m, int32_t T, int32_t nslab, const double *va,
                                   const double *rhat, const double *usa, const double *rho_v,
                                   const double *rho_b, const double *bound_scale,
                                   double kappa, double alpha, double vlo, double vhi,
                                   double *xc, double *zv, double *yv, double *w, double *res,
                                   void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && nslab >= 1 && va && rhat && usa && rho_v && rho_b && xc && zv &&
                 yv && w, "revs_op_node_update: bad argument");
    REVS_REQUIRE(vlo <= vhi, "revs_op_node_update: vlo > vhi");
    REVS_REQUIRE(T <= 256, "revs_op_node_update: T=%d exceeds 256", T);
    hipLaunchKernelGGL(op_node_update_kernel, grid1((int64_t)m * T), dim3(256), 0, S_(stream),
                       m * T, T, nslab, va, rhat, usa, rho_v, rho_b, bound_scale, kappa, alpha, vlo,
                       vhi, xc, zv, yv, w, res);
    REVS_CHECK_LAUNCH("revs_op_node_update");
    return REVS_OK;
}

extern "C" int revs_op_export(int64_t n_homes, int32_t T, const double *sb, float *p_est,
                              void *stream) {
    REVS_REQUIRE(n_homes > 0 && T > 0 && sb && p_est, "revs_op_export: bad argument");
    const int64_t total = n_homes * T;
    hipLaunchKernelGGL(op_export_kernel, grid1(total), dim3(256), 0, S_(stream), total, sb, p_est);
    REVS_CHECK_LAUNCH("revs_op_export");
    return REVS_OK;
}

#define REVS_TL_DISPATCH(T, KERNEL, ...)                                                        \
    do {                                                                                        \
        if ((T) <= 32) hipLaunchKernelGGL((KERNEL<32>), __VA_ARGS__);                           \
        else if ((T) <= 64) hipLaunchKernelGGL((KERNEL<64>), __VA_ARGS__);                      \
        else if ((T) <= 128) hipLaunchKernelGGL((KERNEL<128>), __VA_ARGS__);                    \
        else hipLaunchKernelGGL((KERNEL<256>), __VA_ARGS__);                                    \
    } while (0)

extern "C" int revs_op_node_prep(int32_t m, int32_t T, const int64_t *node_ptr,
                                 const double *inv_sqrt_n, const float *p_est, const float *p_sch,
                                 const float *gamma, double kappa, int32_t preclamp, double *p0,
                                 double *gmin, double *g0_out, void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && T <= 256 && node_ptr && inv_sqrt_n && p_est && p_sch && gamma &&
                 p0 && gmin && kappa > 0, "revs_op_node_prep: bad argument");
    REVS_TL_DISPATCH(T, op_node_prep_kernel, dim3(m), dim3(256), 0, S_(stream), m, T, node_ptr,
                     inv_sqrt_n, p_est, p_sch, gamma, kappa, preclamp, p0, gmin, g0_out);
    REVS_CHECK_LAUNCH("revs_op_node_prep");
    return REVS_OK;
}

extern "C" int revs_op_nodefast_scale(int32_t m, int32_t T, int32_t nslab, const double *wh,
                                      const double *ph0, const double *lam, const double *rho_v,
                                      double kappa, double *xh, double *sx, void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && nslab >= 1 && wh && ph0 && lam && rho_v && xh && sx,
                 "revs_op_nodefast_scale: bad argument");
    hipLaunchKernelGGL(op_nodefast_scale_kernel, grid1((int64_t)m * T), dim3(256), 0, S_(stream),
                       m * T, T, nslab, wh, ph0, lam, rho_v, kappa, xh, sx);
    REVS_CHECK_LAUNCH("revs_op_nodefast_scale");
    return REVS_OK;
}

extern "C" int revs_op_nodefast_update(int32_t m, int32_t T, int32_t nslab, const double *zt,
                                       const double *rho_v, const double *bound_scale,
                                       double alpha, double vlo, double vhi, double *zv,
                                       double *yv, double *w, double *res, void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && T <= 256 && nslab >= 1 && zt && rho_v && zv && yv && w &&
                 vlo <= vhi, "revs_op_nodefast_update: bad argument");
    hipLaunchKernelGGL(op_nodefast_update_kernel, grid1((int64_t)m * T), dim3(256), 0, S_(stream),
                       m * T, T, nslab, zt, rho_v, bound_scale, alpha, vlo, vhi, zv, yv, w, res);
    REVS_CHECK_LAUNCH("revs_op_nodefast_update");
    return REVS_OK;
}

extern "C" int revs_op_nodefast_dualres(int32_t m, int32_t T, int32_t nslab, const double *xh,
                                        const double *ph0, const double *lam, const double *yh,
                                        double kappa, double *res, void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && T <= 256 && nslab >= 1 && xh && ph0 && lam && yh && res,
                 "revs_op_nodefast_dualres: bad argument");
    hipLaunchKernelGGL(op_nodefast_dualres_kernel, grid1((int64_t)m * T), dim3(256), 0, S_(stream),
                       m * T, T, nslab, xh, ph0, lam, yh, kappa, res);
    REVS_CHECK_LAUNCH("revs_op_nodefast_dualres");
    return REVS_OK;
}

extern "C" int revs_op_nodefast_feas(int32_t m, int32_t T, int32_t nslab, const double *v0,
                                     const double *bound_scale, const double *gmin, double vlo,
                                     double vhi, double *cx, double *stats, void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && nslab >= 1 && v0 && gmin && cx && stats && vlo <= vhi,
                 "revs_op_nodefast_feas: bad argument");
    hipLaunchKernelGGL(op_nodefast_feas_kernel, grid1((int64_t)m * T), dim3(256), 0, S_(stream),
                       m * T, T, nslab, v0, bound_scale, gmin, vlo, vhi, cx, stats);
    REVS_CHECK_LAUNCH("revs_op_nodefast_feas");
    return REVS_OK;
}

extern "C" int revs_op_nodefast_finish(int32_t m, int32_t T, int32_t nslab, const double *x,
                                       const double *p0, const double *gmin,
                                       const double *inv_sqrt_n, double *d, double *slack,
                                       double *stats, void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && nslab >= 1 && x && p0 && gmin && inv_sqrt_n && d && slack &&
                 stats, "revs_op_nodefast_finish: bad argument");
    hipLaunchKernelGGL(op_nodefast_finish_kernel, grid1((int64_t)m * T), dim3(256), 0, S_(stream),
                       m * T, T, nslab, x, p0, gmin, inv_sqrt_n, d, slack, stats);
    REVS_CHECK_LAUNCH("revs_op_nodefast_finish");
    return REVS_OK;
}

extern "C" int revs_op_node_apply(int32_t m, int32_t T, const int64_t *node_ptr,
                                  const double *inv_sqrt_n, const float *p_est, const float *p_sch,
                                  const float *gamma, double kappa, int32_t preclamp,
                                  const double *d, float *p_est_new, void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && T <= 256 && node_ptr && inv_sqrt_n && p_est && p_sch && gamma &&
                 d && p_est_new && kappa > 0, "revs_op_node_apply: bad argument");
    REVS_TL_DISPATCH(T, op_node_apply_kernel, dim3(m), dim3(256), 0, S_(stream), m, T, node_ptr,
                     inv_sqrt_n, p_est, p_sch, gamma, kappa, preclamp, d, p_est_new);
    REVS_CHECK_LAUNCH("revs_op_node_apply");
    return REVS_OK;
}
