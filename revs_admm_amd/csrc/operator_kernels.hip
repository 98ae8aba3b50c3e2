// Operator ("Utility") side of one ADMM iteration -- gfx950, double precision.
//
// Reference: class Utility, lpsolver.py:163-238.  The operator's problem
//     min  sum_i (kappa/2)|g_i|^2 + g_i . a_i           a_i = gamma_i - (kappa/2)(P_est_i + P_sch_i)
//     s.t. g >= 0 (Gurobi default lb),  vlo <= R_res g[:,t] <= vhi  for every slot t
// is the projection of g0 = -a/kappa onto the voltage-feasible set.  Here it is
// solved by ADMM in OSQP form (x, z = Cx, y) on C = [R' A~ ; I]:
//     A~ = diag(1/sqrt(n_m)) A   (A = home->node aggregation, n_m homes on node m)
//     R' = R diag(sqrt(n_m)) = U S V^T
// so that (P + sigma I + C^T rho C)^-1 is applied through U, S, V: per-slot rho can
// change without refactoring anything.  One inner iteration =
//     home pass   (this file)   update x, z_b, y_b of every home from the node
//                               correction xc; form rhs and aggregate it to nodes
//     4 skinny GEMMs (gemm_kernels.hip)   V^T rhat, U^T w, V a, U (s a)
//     2 node passes (this file)
// With homes sharded over GPUs the only exchange is the all-reduce of rhat
// (m x T doubles) between the home pass and the GEMMs.
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
    if (i < total) g0[i] = 0.5 * ((double)pe[i] + (double)ps[i]) - (double)gm[i] / kappa;
}

// cold start: x = z_b = max(g0, 0), y_b = 0
__global__ void op_init_home_kernel(int64_t total, const double *g0, double *x, double *zb,
                                    double *yb) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const double v = fmax(g0[i], 0.0);
        x[i] = v; zb[i] = v; yb[i] = 0.0;
    }
}
// cold start, node side: z_v = clip(Cx), y_v = 0, w = rho_v z_v
__global__ void op_init_node_kernel(int total, int T, const double *cx, const double *rho_v,
                                    double vlo, double vhi, double *zv, double *yv, double *w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const double z = fmin(fmax(cx[i], vlo), vhi);
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

// Home pass.  For node m, slot t (c = kappa + sigma + rho_b[t]):
//   if xc:  rhs  = sigma x + kappa g0 + rho_b z_b - y_b          (state before update)
//           xt   = rhs / c + inv_sqrt_n[m] xc[m]
//           x    = alpha xt + (1-alpha) x
//           h    = alpha xt + (1-alpha) z_b
//           z_b  = max(h + y_b / rho_b, 0);   y_b += rho_b (h - z_b)
//   rhat[m] = inv_sqrt_n[m] * sum_homes (sigma x + kappa g0 + rho_b z_b - y_b)   (new state)
__global__ __launch_bounds__(256) void op_home_pass_kernel(
        int m, int T, const int64_t *node_ptr, const double *inv_sqrt_n, double *x, double *zb,
        double *yb, const double *g0, const double *xc, const double *rho_b, double kappa,
        double sigma, double alpha, double *rhat) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * T) return;
    const int node = idx / T, t = idx - node * T;
    const double rb = rho_b[t];
    const double c = kappa + sigma + rb;
    const double isn = inv_sqrt_n[node];
    const double corr = xc ? isn * xc[idx] : 0.0;
    double acc = 0.0;
    for (int64_t i = node_ptr[node]; i < node_ptr[node + 1]; ++i) {
        const int64_t o = i * T + t;
        double xv = x[o], zv = zb[o], yv = yb[o];
        const double kg = kappa * g0[o];
        if (xc) {
            const double rhs = sigma * xv + kg + rb * zv - yv;
            const double xt = rhs / c + corr;
            xv = alpha * xt + (1.0 - alpha) * xv;
            const double h = alpha * xt + (1.0 - alpha) * zv;
            const double zn = fmax(h + yv / rb, 0.0);
            yv += rb * (h - zn);
            zv = zn;
            x[o] = xv; zb[o] = zv; yb[o] = yv;
        }
        acc += sigma * xv + kg + rb * zv - yv;
    }
    rhat[idx] = isn * acc;
}

// w = rho_v z_v - y_v
__global__ void op_node_w_kernel(int total, int T, const double *zv, const double *yv,
                                 const double *rho_v, double *w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) w[i] = rho_v[i % T] * zv[i] - yv[i];
}

// t1 = ta + s tb ;  a = t1 / (c + rho_v s^2) ;  sa = s a        (row j <-> singular value s_j)
__global__ void op_node_scale_kernel(int total, int T, const double *ta, const double *tb,
                                     const double *s, const double *rho_v, const double *rho_b,
                                     double kappa, double sigma, double *a, double *sa) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = i / T, t = i - j * T;
    const double sj = s[j];
    const double c = kappa + sigma + rho_b[t];
    const double av = (ta[i] + sj * tb[i]) / (c + rho_v[t] * sj * sj);
    a[i] = av;
    sa[i] = sj * av;
}

// xc = va - rhat / c ;  h = alpha usa + (1-alpha) z_v ;  z_v = clip(h + y_v/rho_v) ;
// y_v += rho_v (h - z_v) ;  cx = alpha usa + (1-alpha) cx ;  w = rho_v z_v - y_v
__global__ void op_node_update_kernel(int total, int T, const double *va, const double *rhat,
                                      const double *usa, const double *rho_v,
                                      const double *rho_b, double kappa, double sigma,
                                      double alpha, double vlo, double vhi, double *xc,
                                      double *zv, double *yv, double *cx, double *w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int t = i % T;
    const double rv = rho_v[t];
    const double c = kappa + sigma + rho_b[t];
    xc[i] = va[i] - rhat[i] / c;
    const double zt = usa[i];
    const double zo = zv[i];
    const double h = alpha * zt + (1.0 - alpha) * zo;
    double y = yv[i];
    const double zn = fmin(fmax(h + y / rv, vlo), vhi);
    y += rv * (h - zn);
    zv[i] = zn;
    yv[i] = y;
    cx[i] = alpha * zt + (1.0 - alpha) * cx[i];
    w[i] = rv * zn - y;
}

// Per-slot residual maxima (out[8][T], zeroed by the caller, all entries >= 0):
//  0 max_m |cx - z_v|     1 max_i |x - z_b|     2 max_i |kappa (x - g0) + isn cty[m] + y_b|
//  3 max_m |cx|           4 max_m |z_v|         5 max_i |x|
//  6 max_i |isn cty[m] + y_b|                   7 max_i |kappa g0|
__global__ __launch_bounds__(256) void op_residuals_kernel(
        int m, int T, const int64_t *node_ptr, const double *inv_sqrt_n, const double *x,
        const double *zb, const double *yb, const double *g0, const double *cty,
        const double *cx, const double *zv, double kappa, double *out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * T) return;
    const int node = idx / T, t = idx - node * T;
    const double ct = inv_sqrt_n[node] * cty[idx];
    double r1 = 0, r2 = 0, r5 = 0, r6 = 0, r7 = 0;
    for (int64_t i = node_ptr[node]; i < node_ptr[node + 1]; ++i) {
        const int64_t o = i * T + t;
        const double xv = x[o], gv = g0[o];
        const double cy = ct + yb[o];
        r1 = fmax(r1, fabs(xv - zb[o]));
        r2 = fmax(r2, fabs(kappa * (xv - gv) + cy));
        r5 = fmax(r5, fabs(xv));
        r6 = fmax(r6, fabs(cy));
        r7 = fmax(r7, fabs(kappa * gv));
    }
    atomic_max_nonneg(out + 0 * T + t, fabs(cx[idx] - zv[idx]));
    atomic_max_nonneg(out + 1 * T + t, r1);
    atomic_max_nonneg(out + 2 * T + t, r2);
    atomic_max_nonneg(out + 3 * T + t, fabs(cx[idx]));
    atomic_max_nonneg(out + 4 * T + t, fabs(zv[idx]));
    atomic_max_nonneg(out + 5 * T + t, r5);
    atomic_max_nonneg(out + 6 * T + t, r6);
    atomic_max_nonneg(out + 7 * T + t, r7);
}

__global__ void op_export_kernel(int64_t total, const double *zb, float *pe) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) pe[i] = (float)zb[i];
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

extern "C" int revs_op_init_home(int64_t n_homes, int32_t T, const double *g0, double *x,
                                 double *zb, double *yb, void *stream) {
    REVS_REQUIRE(n_homes > 0 && T > 0 && g0 && x && zb && yb, "revs_op_init_home: bad argument");
    const int64_t total = n_homes * T;
    hipLaunchKernelGGL(op_init_home_kernel, grid1(total), dim3(256), 0, S_(stream), total, g0, x,
                       zb, yb);
    REVS_CHECK_LAUNCH("revs_op_init_home");
    return REVS_OK;
}

extern "C" int revs_op_init_node(int32_t m, int32_t T, const double *cx, const double *rho_v,
                                 double vlo, double vhi, double *zv, double *yv, double *w,
                                 void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && cx && rho_v && zv && yv && w, "revs_op_init_node: bad argument");
    hipLaunchKernelGGL(op_init_node_kernel, grid1((int64_t)m * T), dim3(256), 0, S_(stream),
                       m * T, T, cx, rho_v, vlo, vhi, zv, yv, w);
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
                                 const double *inv_sqrt_n, double *x, double *zb, double *yb,
                                 const double *g0, const double *xc, const double *rho_b,
                                 double kappa, double sigma, double alpha, double *rhat,
                                 void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && node_ptr && inv_sqrt_n && x && zb && yb && g0 && rho_b && rhat,
                 "revs_op_home_pass: bad argument");
    hipLaunchKernelGGL(op_home_pass_kernel, grid1((int64_t)m * T), dim3(256), 0, S_(stream), m, T,
                       node_ptr, inv_sqrt_n, x, zb, yb, g0, xc, rho_b, kappa, sigma, alpha, rhat);
    REVS_CHECK_LAUNCH("revs_op_home_pass");
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

extern "C" int revs_op_node_scale(int32_t m, int32_t T, const double *ta, const double *tb,
                                  const double *s, const double *rho_v, const double *rho_b,
                                  double kappa, double sigma, double *a, double *sa,
                                  void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && ta && tb && s && rho_v && rho_b && a && sa,
                 "revs_op_node_scale: bad argument");
    hipLaunchKernelGGL(op_node_scale_kernel, grid1((int64_t)m * T), dim3(256), 0, S_(stream),
                       m * T, T, ta, tb, s, rho_v, rho_b, kappa, sigma, a, sa);
    REVS_CHECK_LAUNCH("revs_op_node_scale");
    return REVS_OK;
}

extern "C" int revs_op_node_update(int32_t m, int32_t T, const double *va, const double *rhat,
                                   const double *usa, const double *rho_v, const double *rho_b,
                                   double kappa, double sigma, double alpha, double vlo,
                                   double vhi, double *xc, double *zv, double *yv, double *cx,
                                   double *w, void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && va && rhat && usa && rho_v && rho_b && xc && zv && yv && cx && w,
                 "revs_op_node_update: bad argument");
    REVS_REQUIRE(vlo <= vhi, "revs_op_node_update: vlo > vhi");
    hipLaunchKernelGGL(op_node_update_kernel, grid1((int64_t)m * T), dim3(256), 0, S_(stream),
                       m * T, T, va, rhat, usa, rho_v, rho_b, kappa, sigma, alpha, vlo, vhi, xc,
                       zv, yv, cx, w);
    REVS_CHECK_LAUNCH("revs_op_node_update");
    return REVS_OK;
}

extern "C" int revs_op_residuals(int32_t m, int32_t T, const int64_t *node_ptr,
                                 const double *inv_sqrt_n, const double *x, const double *zb,
                                 const double *yb, const double *g0, const double *cty_node,
                                 const double *cx, const double *zv, double kappa, double *out,
                                 void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && node_ptr && inv_sqrt_n && x && zb && yb && g0 && cty_node &&
                 cx && zv && out, "revs_op_residuals: bad argument");
    hipLaunchKernelGGL(op_residuals_kernel, grid1((int64_t)m * T), dim3(256), 0, S_(stream), m, T,
                       node_ptr, inv_sqrt_n, x, zb, yb, g0, cty_node, cx, zv, kappa, out);
    REVS_CHECK_LAUNCH("revs_op_residuals");
    return REVS_OK;
}

extern "C" int revs_op_export(int64_t n_homes, int32_t T, const double *zb, float *p_est,
                              void *stream) {
    REVS_REQUIRE(n_homes > 0 && T > 0 && zb && p_est, "revs_op_export: bad argument");
    const int64_t total = n_homes * T;
    hipLaunchKernelGGL(op_export_kernel, grid1(total), dim3(256), 0, S_(stream), total, zb, p_est);
    REVS_CHECK_LAUNCH("revs_op_export");
    return REVS_OK;
}
