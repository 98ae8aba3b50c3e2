// Operator ("Utility") QP through its dual: semismooth Newton on the voltage-row
// multipliers -- gfx950, double precision.
//
// Reference: class Utility, lpsolver.py:163-238 (Gurobi's barrier solves it there).
//     min (kappa/2)|g - g0|^2   s.t.  g >= 0,  vlo <= R (A g)[:,t] <= vhi  for every slot t
// With y[m][t] the multiplier of row (m,t), the residences of node m answer
//     g_i = max(g0_i - d_m, 0),   d = R^T y / kappa
// -- the g >= 0 rows are handled exactly inside this closed form, they never enter the
// iteration.  The dual of slot t,
//     D_t(y) = -(kappa/2) sum_i g_i^2 - sum_m max(vhi y_m, vlo y_m),
// is concave, C^1, piecewise quadratic; dD/dy_m = (R p)_m - b_m with p = A g.  A Newton
// iteration maximises the quadratic model with generalised Hessian -K, K = R_F N R_F^T/kappa
// (N_m = number of unclamped residences of node m), over the sign constraints of a
// candidate set F_t of at most 128 rows (rows with y != 0 and the most violated ones), and
// takes an Armijo step on D_t.  Only a handful of rows of a radial feeder bind, so the
// small dense problems live in one workgroup's LDS; the heavy parts are
//     2 products  R^T y, R p      gemm_kernels.hip, f64 matrix cores
//     1 home pass                 op_dual_eval_kernel, HBM bound (3 float reads + 1 write)
// per evaluation, and a run needs 5-15 evaluations where the ADMM form needed 10^2-10^4
// iterations of the same cost.
#define REVS_KVS_TU
#include "common.h"
#include "select_body.h"
#include "tree_body.h"
#include "internal.h"
#include <type_traits>

namespace revs {

// ---- home pass: p, N, -(kappa/2) sum g^2 per node and slot; P_est_new = g ----------
// Mapping as op_home_pass_kernel: one workgroup per node, TL slot lanes x HS home lanes,
// partial sums combined through LDS in a fixed order.
// d = R^T y / kappa either from the K-split slabs of the dense product (dsl) or, while only a
// few rows carry a multiplier, straight from those rows of R (SparseD: the candidate list of
// the last selection covers every row with y != 0) -- a dozen loads per node and slot
// instead of a 12 us product.
struct SparseD {
    const double *R;          // row-major [m][m]
    const int64_t *sidx;      // [T][kAmax] candidate rows
    const int32_t *scnt;      // [T]
    const double *y;          // [m][T]
};

template <int TL>
__global__ __launch_bounds__(256) void op_dual_eval_kernel(
        int m, int T, const int64_t *__restrict__ node_ptr, const float *__restrict__ pe,
        const float *__restrict__ ps, const float *__restrict__ gm, int nslab,
        const double *__restrict__ dsl, double kappa, double *__restrict__ pnq,
        float *__restrict__ pe_new, const SparseD sp) {
    constexpr int HS = 256 / TL;
    const int node = blockIdx.x;
    const int t = threadIdx.x % TL, hs = threadIdx.x / TL;
    const bool tok = t < T;
    const int64_t total = (int64_t)m * T;
    const int idx = node * T + (tok ? t : 0);
    const double inv_k = 1.0 / kappa;
    const float inv_kf = 1.0f / (float)kappa;
    double d = 0.0;
    if (dsl) {
        for (int q = 0; q < nslab; ++q) d += dsl[idx + q * total];
        d *= inv_k;
    } else if (sp.R && tok) {
        const int cnt = sp.scnt[t];
        const int64_t *si = sp.sidx + (int64_t)t * kAmax;
        // eight rows at a time, predicated: the index loads, then the R and y loads of a batch
        // are all in flight together (two memory round trips per batch, not two per row)
        for (int i0 = 0; i0 < cnt; i0 += 8) {
            int64_t f[8];
            double rv[8], yv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = i0 + k < cnt ? si[i0 + k] : 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                rv[k] = sp.R[f[k] * m + node];
                yv[k] = i0 + k < cnt ? sp.y[f[k] * T + t] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) d = __builtin_fma(rv[k], yv[k], d);    // (explicit: chain_shifts_body repeats it)
        }
        d *= inv_k;
    }
    double ap = 0.0, an = 0.0, aq = 0.0;
    const int64_t i0 = node_ptr[node], i1 = node_ptr[node + 1];
    if (tok) {
        for (int64_t i = i0 + hs; i < i1; i += HS) {
            const int64_t o = i * T + t;
            const double g0 = (double)revs_g0f(pe[o], ps[o], gm[o], inv_kf);
            const bool fr = g0 > d;
            const double g = fr ? g0 - d : 0.0;
            ap += revs_q36(g);              // (order-independent sums: common.h)
            an += fr ? 1.0 : 0.0;
            aq += revs_q32(g * g);
            if (pe_new) pe_new[o] = (float)g;
        }
    }
    __shared__ double red[3][HS][TL];
    red[0][hs][t] = ap; red[1][hs][t] = an; red[2][hs][t] = aq;
    __syncthreads();
    if (hs == 0 && tok) {
        double a = red[0][0][t], b = red[1][0][t], c = red[2][0][t];
#pragma unroll
        for (int k = 1; k < HS; ++k) { a += red[0][k][t]; b += red[1][k][t]; c += red[2][k][t]; }
        pnq[idx] = a;
        pnq[idx + total] = b;
        pnq[idx + 2 * total] = -0.5 * kappa * c;
    }
}

// ---- per-slot bookkeeping: v, residual, D_t, candidate rows ------------------------
// Stage 1, all rows in parallel: v = sum of the product's K-split slabs, the residual /
// violation of every row, and per-(row block, slot) partial reductions in a fixed order.
// Workgroup = 8 rows x TL slot lanes; partial[blk][t] = {max residual, sum D terms,
// rows with y != 0, violated rows with y = 0}.
template <int TL>
__global__ __launch_bounds__(256) void op_dual_rows_kernel(
        int m, int T, int nslab, const double *__restrict__ vsl, const double *__restrict__ pnq,
        const double *__restrict__ y, double vlo, double vhi, double *__restrict__ vfull,
        double *__restrict__ viol, double *__restrict__ partial, double *__restrict__ zero_out) {
    constexpr int HS = 256 / TL;
    const int t = threadIdx.x % TL, hs = threadIdx.x / TL;
    const int rows_per = HS * ((m + HS * (int)gridDim.x - 1) / (HS * (int)gridDim.x));
    const int rb = blockIdx.x * rows_per;
    const int64_t total = (int64_t)m * T;
    double rmax = 0.0, dsum = 0.0, nsup = 0.0, nvio = 0.0;
    if (t < T) {
        for (int r = rb + hs; r < min(m, rb + rows_per); r += HS) {
            const int64_t i = (int64_t)r * T + t;
            double v = 0.0;
            for (int q = 0; q < nslab; ++q) v += vsl[i + q * total];
            vfull[i] = v;
            const double yv = y[i];
            const bool up = yv > 0.0 || (yv == 0.0 && v > vhi);
            const double b = up ? vhi : vlo;
            const double vi = fmax(fmax(v - vhi, vlo - v), 0.0);
            rmax = fmax(rmax, yv != 0.0 ? fabs(v - b) : vi);
            dsum += pnq[i + 2 * total] - fmax(vhi * yv, vlo * yv);
            nsup += yv != 0.0 ? 1.0 : 0.0;
            nvio += (yv == 0.0 && vi > 0.0) ? 1.0 : 0.0;
            viol[i] = yv != 0.0 ? 0.0 : vi;
            if (zero_out) zero_out[i] = 0.0;
        }
    }
    __shared__ double red[4][HS][TL];
    red[0][hs][t] = rmax; red[1][hs][t] = dsum; red[2][hs][t] = nsup; red[3][hs][t] = nvio;
    __syncthreads();
    if (hs == 0 && t < T) {
        double a = red[0][0][t], b = red[1][0][t], c = red[2][0][t], d = red[3][0][t];
#pragma unroll
        for (int k = 1; k < HS; ++k) {
            a = fmax(a, red[0][k][t]); b += red[1][k][t]; c += red[2][k][t]; d += red[3][k][t];
        }
        double *o = partial + ((int64_t)blockIdx.x * T + t) * 4;
        o[0] = a; o[1] = b; o[2] = c; o[3] = d;
    }
}

// The same bookkeeping behind the tree form of R p (tree_body.h): one workgroup per slot computes
// v = R p of its slot in O(nodes) -- the node sums of a slot are 16 KB where the dense product
// streams the 33 MB of R -- and judges its rows on the spot; partial has ONE block per slot
// (nblk = 1 for the selection).  Rows of nodes without residences carry no position in the tree:
// their v, violation and multiplier are zero and stay so (the arrays start from zero).
struct TreeRowsArgs {
    TreeArgs tree;
    int m, T;
    const double *p;          // node sums p[m][T]
    const double *qn;         // -(kappa/2) sum g^2 per node and slot (pnq + 2 m T)
    const double *y;
    double vlo, vhi;
    double *vfull, *viol, *partial, *zero_out;
    double *ycopy_out = nullptr;   // != NULL: the slot's multipliers are copied there, row by row (the next trial's start)
    int es = 1;                    // layout of p / qn (ChainKvSide::es): 1 = planar [m][T] arrays; 4 = the folded sweep's [T][m][4] =
                                   // {p, N, q, 0} per slot and node, slot-major (p = base, qn = base + 2)
};
// rows_lds != NULL (double[3 m + 4] of LDS): the slot's multipliers, voltages and violations by row and
// the four sums are left THERE for a selection that follows in the same workgroup
// (SelectArgs::rows_lds) instead of in the global columns vfull / viol.
struct NoPrefetch { __device__ void operator()() const {} };
// after_pack(): the caller's own loads, issued behind the packed indices (the first thing every gather
// waits for) and in front of everything else
// NT x IPT: the workgroup's shape (tree_body.h; 256 x 8 inside the fused launches, the bigger ones in
// op_tree_rows_big_kernel for feeders of more than 2048 nodes)
template <class F = NoPrefetch, int NT = 256, int IPT = 8>
__device__ __forceinline__ void tree_rows_body(const TreeRowsArgs &a, const int t, double *lds,
                                               double *rows_lds = nullptr, F &&after_pack = F()) {
    const int tid = threadIdx.x, j0 = IPT * tid, T = a.T;
    const bool act = j0 < a.tree.n;
    // the positions' packed indices first: every gather below -- multipliers, dual terms, and the node
    // sums inside tree_voltage -- waits for them and for nothing else
    unsigned long long pk[IPT];
#pragma unroll
    for (int i = 0; i < IPT; ++i) pk[i] = 0ull;
    if (act) {
#pragma unroll
        for (int i = 0; i < IPT; i += 2) {
            const TreeU2 u = *reinterpret_cast<const TreeU2 *>(a.tree.pack + j0 + i);
            pk[i] = u.v[0]; pk[i + 1] = u.v[1];
        }
    }
    after_pack();
    REVS_KVS(t, 21);
    if (rows_lds)      // (rows without a position in the tree: zero; the scans' barriers order this)
        for (int i = tid; i < 3 * a.m; i += NT) rows_lds[i] = 0.0;
    REVS_KVS(t, 22);
#ifdef REVS_KV_STAMPS
    asm volatile("" :: "v"(pk[0]), "v"(pk[IPT - 1]));
    REVS_KVS(t, 23);
#endif
    if (a.zero_out)
        for (int r = tid; r < a.m; r += NT) a.zero_out[(int64_t)r * T + t] = 0.0;
    double yv[IPT], qv[IPT];
#pragma unroll
    for (int i = 0; i < IPT; ++i) {           // (positions without a row fetch row 0: masked behind the scans)
        const int s = (int)(pk[i] & 0xFFFFu) - 1;
        yv[i] = a.y[(int64_t)(s >= 0 ? s : 0) * T + t];
        qv[i] = a.es == 4 ? a.qn[4 * ((int64_t)t * a.m + (s >= 0 ? s : 0))] : a.qn[(int64_t)(s >= 0 ? s : 0) * T + t];
    }
    double v8[IPT];
    REVS_KVS(t, 1);
    {   // (tree_voltage<NT, IPT, true, true>, with the node sums' element stride)
        double wb[IPT];
        tree_fetch_w<NT, IPT>(a.tree, wb);
        if (a.es == 4) tree_gather_p<NT, IPT, true>(a.tree, a.p, 4, 4 * t * a.m, pk, v8, nullptr);     // (row stride 4, the slot's block)
        else tree_gather_p<NT, IPT, true>(a.tree, a.p, T, t, pk, v8, nullptr);
        tree_scan<NT, IPT, true>(a.tree, t, lds, v8, wb, pk);
    }
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const bool has = (pk[i] & 0xFFFFu) != 0ull;
        yv[i] = has ? yv[i] : 0.0;
        qv[i] = has ? qv[i] : 0.0;
    }
    double rmax = 0.0, dsum = 0.0, nsup = 0.0, nvio = 0.0;
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const int s = (int)(pk[i] & 0xFFFFu) - 1;
        if (s >= 0) {
            const int64_t o = (int64_t)s * T + t;
            const double v = v8[i], y1 = yv[i];
            const bool up = y1 > 0.0 || (y1 == 0.0 && v > a.vhi);
            const double b = up ? a.vhi : a.vlo;
            const double vi = fmax(fmax(v - a.vhi, a.vlo - v), 0.0);
            rmax = fmax(rmax, y1 != 0.0 ? fabs(v - b) : vi);
            dsum += qv[i] - fmax(a.vhi * y1, a.vlo * y1);
            nsup += y1 != 0.0 ? 1.0 : 0.0;
            nvio += (y1 == 0.0 && vi > 0.0) ? 1.0 : 0.0;
            if (a.ycopy_out) a.ycopy_out[o] = y1;
            if (rows_lds) {
                rows_lds[s] = y1;
                rows_lds[a.m + s] = v;
                rows_lds[2 * a.m + s] = y1 != 0.0 ? 0.0 : vi;
            } else {
                a.vfull[o] = v;
                a.viol[o] = y1 != 0.0 ? 0.0 : vi;
            }
        }
    }
    REVS_KVS(t, 6);
    rmax = wave_max_d(rmax); dsum = wave_sum_d(dsum); nsup = wave_sum_d(nsup); nvio = wave_sum_d(nvio);
    constexpr int NW = NT / 64;
    __shared__ double rr[4][NW];
    if ((tid & 63) == 0) { rr[0][tid >> 6] = rmax; rr[1][tid >> 6] = dsum; rr[2][tid >> 6] = nsup; rr[3][tid >> 6] = nvio; }
    __syncthreads();
    if (tid == 0) {
        double *o = a.partial + (int64_t)t * 4;
        double r0 = rr[0][0], r1 = rr[1][0], r2 = rr[2][0], r3 = rr[3][0];        // (wavefront order: fixed)
#pragma unroll
        for (int w = 1; w < NW; ++w) { r0 = fmax(r0, rr[0][w]); r1 += rr[1][w]; r2 += rr[2][w]; r3 += rr[3][w]; }
        o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
        if (rows_lds) { rows_lds[3 * a.m] = o[0]; rows_lds[3 * a.m + 1] = o[1]; rows_lds[3 * a.m + 2] = o[2]; rows_lds[3 * a.m + 3] = o[3]; }
    }
}

// Rows by the tree form for feeders of more than 2048 nodes (the shapes of tree_body.h: 512 x 8, 1024 x 8, 1024 x 16):
// v, violations and the slot's four sums to global memory; the selection follows as its own launch
// (op_dual_select_kernel, nblk = 1).
template <int NT, int IPT>
__global__ __launch_bounds__(NT) void op_tree_rows_big_kernel(const TreeRowsArgs ta) {
    extern __shared__ double tree_lds[];
    tree_rows_body<NoPrefetch, NT, IPT>(ta, blockIdx.x, tree_lds);
}

// The shifts of an evaluation, d = R^T y, by the tree form (R of a radial feeder is symmetric: R^T y = R y is the same
// three prefix sums with the multipliers in the place of the node sums), one workgroup per slot, into slab 0 of the
// dense product's output (rows that carry no checked position -- nodes without a residence -- are not written: the
// home pass reads d only where residences are).  Replaces the f64 matrix-core product of the evaluations on radial
// feeders: 22.8 us per evaluation at M = 1 126, T = 96 (0.14 of the MFMA peak: launch- and latency-bound), 11.8 at T = 24;
// this launch: 6.4 us.  (Folded into the step's launch in front of it -- the same T workgroups -- the feeder's 15 iterations
// took 8.7-8.9 ms against 8.6: measured and dropped, r05.)
template <int NT, int IPT>
__global__ __launch_bounds__(NT) void op_tree_shift_kernel(const TreeArgs tr, const int T, const double *__restrict__ y,
                                                           double *__restrict__ d_out) {
    extern __shared__ double tree_lds[];
    const int t = blockIdx.x;
    double a[IPT];
    unsigned long long pk[IPT];
    tree_voltage<NT, IPT, false, true>(tr, y, T, t, tree_lds, a, pk, nullptr);
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const int s = (int)(pk[i] & 0xFFFFu) - 1;
        if (s >= 0) d_out[(int64_t)s * T + t] = a[i];
    }
}

// rows by the tree form, then (SELECT) the candidate selection of the same slot in the same workgroup
// (SELECT with `staged`: the slot's multipliers, voltages and violations reach the selection through LDS -- double[3 m + 4]
// behind the tree's scan buffer -- instead of the global columns vfull / viol, which are then NOT written: two column
// stores and three column loads of m cache lines each off a latency-bound workgroup)
template <bool SELECT>
__global__ __launch_bounds__(256) void op_tree_rows_kernel(const TreeRowsArgs ta, const SelectArgs sa, const int staged) {
#ifdef REVS_TUNING
    REVS_KVS_BEGIN(SELECT ? REVS_ROWS_STAMP_PTR : nullptr);
    __syncthreads();
    REVS_KVS(blockIdx.x, 0);
#else
    REVS_KVS_BEGIN(nullptr);
#endif
    extern __shared__ double tree_lds[];
    if constexpr (SELECT) {
        if (staged) {
            double *rows_lds = tree_lds + (tree_lds_bytes(ta.tree.n) / sizeof(double) + 1) / 2 * 2;
            tree_rows_body(ta, blockIdx.x, tree_lds, rows_lds);
            __syncthreads();
            SelectArgs s2 = sa;
            s2.rows_lds = rows_lds;
            dual_select_body<true>(s2, blockIdx.x);
            REVS_KVS(blockIdx.x, 11);
            return;
        }
    }
    tree_rows_body(ta, blockIdx.x, tree_lds);
    if constexpr (SELECT) {
        __syncthreads();                    // this workgroup's v, violations and sums, in global memory
        dual_select_body<true>(sa, blockIdx.x);
    }
}

// Stage 2 (select_body.h), one workgroup per slot, as its own kernel.
__global__ __launch_bounds__(256) void op_dual_select_kernel(const SelectArgs sa) {
    REVS_KVS_BEGIN(nullptr);
    dual_select_body<true>(sa, blockIdx.x);
}

// ---- model Hessian: K_t = R_F N_t R_F^T (candidates x candidates), K-split slabs ------
// Workgroup (t, ks, tile) computes one 64 x 64 tile (tile = 2 bi + bj) over the columns
// [ks, ks+1) * m/nks of R in chunks of 32: the candidate rows of R (and the same rows scaled
// by N[.,t]) are staged in LDS with coalesced loads, and thread (ti, tj) accumulates a 4 x 4
// block.  Tiles beyond the slot's candidate count exit at once.  Slabs are summed (fixed
// order) by op_dual_bpp_kernel.
__global__ __launch_bounds__(256) void op_dual_gram_kernel(
        int m, int T, const double *__restrict__ R, const double *__restrict__ Nn,
        const int64_t *__restrict__ cidx, const int32_t *__restrict__ ccnt, int nks,
        double *__restrict__ Kslab) {
    const int t = blockIdx.x, ks = blockIdx.y, tid = threadIdx.x;
    const int bi = blockIdx.z >> 1, bj = blockIdx.z & 1;
    const int a = ccnt[t];
    if (a <= 0 || bi * 64 >= a || bj * 64 >= a) return;
    constexpr int KC = 32;
    // (row stride 36 doubles: the matrix cores' operand reads below -- lane = row % 16 + 16 (column % 4) -- then fall on
    // 4 row + column, every bank pair once per half wavefront)
    constexpr int KS = KC + 4;
    __shared__ double Ws[64][KS], Rs_[64][KS];
    const int ai = min(64, (a - bi * 64 + 3) & ~3), aj = min(64, (a - bj * 64 + 3) & ~3);
    // The products on the matrix cores (round 5): wavefront w forms rows 16 w .. 16 w + 15 of the tile against its 64
    // columns, four v_mfma_f64_16x16x4 per four columns of the chunk -- five LDS reads per lane for 4096 multiply-adds.
    // (As 4 x 4 register blocks on the vector pipe every multiply-add pair took an LDS read: the launch was bound by
    // LDS bandwidth, ~24 us on the 121144 feeder whatever the loads did.)
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;              // operand row (A) / column (B) inside a 16-tile, k inside a step
    const bool mine = wave * 16 < ai;
    d4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = d4{0.0, 0.0, 0.0, 0.0};
    const int chunk = (m + nks - 1) / nks;
    const int k0 = ks * chunk, k1 = min(m, k0 + chunk);
    // stage: 256 threads = 8 rows x 32 columns per pass.  All sixteen entries of a chunk are requested before the first is
    // used (clamped addresses, masked afterwards: with the loads inside the tests the loop was sixteen dependent round trips
    // per chunk -- 29 us per launch on the 121144 feeder), and the next chunk's before this chunk's products are formed.
    // (the rows' numbers straight from the lists into the registers of the threads that fetch them -- they went through LDS
    // and a barrier before the first fetch could be issued; the diagonal tile, the only one up to 64 candidates, fetches
    // its rows once, not once for either side of the product: r05)
    const bool diag = bi == bj;
    int64_t rwi[8], rwj[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int li = bi * 64 + (tid >> 5) + 8 * q, lj = bj * 64 + (tid >> 5) + 8 * q;
        rwi[q] = li < a ? cidx[(int64_t)t * kAmax + li] : -1;
        rwj[q] = diag ? rwi[q] : (lj < a ? cidx[(int64_t)t * kAmax + lj] : -1);
    }
    double rvi[8], rvj[8], nv = 0.0;
    auto fetch = [&](int kk) {
        const int c = kk + (tid & 31);
        const int cc = c < k1 ? c : k1 - 1;
        nv = Nn[(int64_t)cc * T + t];
#pragma unroll
        for (int q = 0; q < 8; ++q) rvi[q] = R[(rwi[q] >= 0 ? rwi[q] : 0) * m + cc];
        if (diag) {
#pragma unroll
            for (int q = 0; q < 8; ++q) rvj[q] = rvi[q];
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) rvj[q] = R[(rwj[q] >= 0 ? rwj[q] : 0) * m + cc];
        }
    };
    if (k0 < k1) fetch(k0);
    for (int kk = k0; kk < k1; kk += KC) {
        const bool cin = kk + (tid & 31) < k1;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int rr = (tid >> 5) + 8 * q;
            const double wv_ = rvi[q] * nv;
            Ws[rr][tid & 31] = (rwi[q] >= 0 && cin) ? wv_ : 0.0;
            Rs_[rr][tid & 31] = (rwj[q] >= 0 && cin) ? rvj[q] : 0.0;
        }
        __syncthreads();
        if (kk + KC < k1) fetch(kk + KC);                   // (uniform; in flight behind the products below)
        if (mine) {                                         // (uniform per wavefront)
#pragma unroll
            for (int st = 0; st < KC / 4; ++st) {
                const double av = Ws[wave * 16 + fr][4 * st + fk];
                double bv[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) bv[n] = Rs_[16 * n + fr][4 * st + fk];
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[n], acc[n], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (mine) {
        // C / D map of v_mfma_f64_16x16x4_f64 (gemm_kernels.hip): column = lane & 15, row = (lane >> 4) + 4 reg
        double *o = Kslab + ((int64_t)t * nks + ks) * kAmax * kAmax;
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int li = wave * 16 + fk + 4 * reg, lj = 16 * n + fr;
                if (li < ai && lj < aj) o[(bi * 64 + li) * kAmax + bj * 64 + lj] = acc[n][reg];
            }
    }
}

// ---- the model problem of one slot: block principal pivoting in LDS ------------------
// In sign-normalised variables u = s y >= 0 the model is the LCP
//     u >= 0,  w = K' u - c >= 0,  u . w = 0,   K' = S K S + delta I,  c = s grad + K' u_cur.
// Block principal pivoting (Judice & Pires) with the single-exchange backup rule: basic
// set B (u free, w = 0) as two 64-bit words; solve K'_BB u_B = c_B by Cholesky in LDS (the
// workgroup's dynamic LDS holds the 128 x 129 factor: 129 KB of the CU's 160 KB), swap
// every infeasible index (u_i < 0 in B, w_i < 0 outside) while that shrinks their number,
// else only the highest one -- finite for a positive definite K'.
__global__ __launch_bounds__(256) void op_dual_bpp_kernel(
        const double *__restrict__ Kslab, int nks, double inv_kappa, double *__restrict__ Kall,
        const int32_t *__restrict__ ccnt, const double *__restrict__ cval, double delta,
        int max_pivots, double *__restrict__ yhat, int32_t *__restrict__ info) {
    const int t = blockIdx.x, tid = threadIdx.x;
    const int a = ccnt[t];
    const double *cs = cval + (int64_t)t * 3 * kAmax, *cg = cs + kAmax, *cy = cg + kAmax;
    double *yo = yhat + (int64_t)t * kAmax;
    if (a <= 0) {                           // uniform
        if (tid < kAmax) yo[tid] = cy[tid];
        if (tid == 0) info[t] = 0;
        return;
    }
    BPP_STAMP(0);
#ifdef REVS_BPP_STAMPS
    if (tid == 0) g_bpp_stamps[blockIdx.x][28] = (double)clock64();
#endif
    extern __shared__ double Ls_dyn[];                    // [kAmax][kAmax + 1]
    constexpr int kKl = 65;                               // row stride of the LDS copy of K (a <= 64)
    // K = (sum of the K-split slabs) / kappa, rows and columns < a (rounded up to 4)
    double *Kt = Kall + (int64_t)t * kAmax * kAmax;
    {
        // (four elements per thread and pass, slab by slab: 4 loads in flight per round trip, same order of the sum)
        const int a4 = (a + 3) & ~3, tot = a4 * a4;
        for (int e0 = tid; e0 < tot; e0 += 4 * 256) {
            const double *src[4];
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            int off[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int e = e0 + 256 * x, ec = e < tot ? e : 0;
                const int i = ec / a4, j = ec - i * a4;
                off[x] = i * kAmax + j;
                src[x] = Kslab + (int64_t)t * nks * kAmax * kAmax + off[x];
            }
            // (all slabs of the four elements requested before the first is added: one memory round trip per pass,
            // not one per slab -- the slabs' 25 MB do not sit in a cache)
            for (int q0 = 0; q0 < nks; q0 += 8) {
                double v[8][4];
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int x = 0; x < 4; ++x) v[q][x] = src[x][(int64_t)(q0 + q < nks ? q0 + q : 0) * kAmax * kAmax];
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int x = 0; x < 4; ++x) acc[x] += q0 + q < nks ? v[q][x] : 0.0;
            }
#pragma unroll
            for (int x = 0; x < 4; ++x)
                if (e0 + 256 * x < tot) {
                    const double kv_ = acc[x] * inv_kappa;
                    Kt[off[x]] = kv_;
                    // (at most 64 candidates: a copy of K in the half of the factor's LDS that 64 rows leave free -- the
                    // products K' u of every round and the factorisations' first reads then take LDS trips, not
                    // trips to the L2 behind the stores above)
                    if (a <= 64) Ls_dyn[64 * (kAmax + 1) + (off[x] >> 7) * kKl + (off[x] & (kAmax - 1))] = kv_;
                }
        }
    }
    __syncthreads();
    BPP_STAMP(1);
    auto Ls = [&](int i, int j) -> double & { return Ls_dyn[i * (kAmax + 1) + j]; };
    const double *const Kp = a <= 64 ? (const double *)(Ls_dyn + 64 * (kAmax + 1)) : (const double *)Kt;
    const int ks = a <= 64 ? kKl : kAmax;
    // (the factorisations read K through two pointers in their own address spaces, on two code paths: through the one
    // above -- LDS or global by `a`, a generic pointer -- every entry was a flat load the compiler waited for on its own)
    const bool klds = a <= 64;
    typedef const __attribute__((address_space(3))) double *lds_cdp;
    typedef const __attribute__((address_space(1))) double *glb_cdp;
    const lds_cdp Kl = (lds_cdp)(Ls_dyn + 64 * (kAmax + 1));
    const glb_cdp Kg = (glb_cdp)Kt;
    __shared__ double s_s[kAmax], c_s[kAmax], u_s[kAmax], w_s[kAmax], wred[4], idk_s[kAmax], col_s[2][4][kAmax];
    __shared__ int bl[kAmax];
    __shared__ unsigned long long Bsh[kWords], Vsh[kWords];
    __shared__ double dl_s, tolw_s, tolu_s;
    __shared__ int done_s, piv_s;
    const int lane = tid & 63, wave = tid >> 6;

    if (tid < kAmax) {
        const bool in = tid < a;
        const double s = in ? cs[tid] : 1.0;
        s_s[tid] = s;
        u_s[tid] = in ? fmax(s * cy[tid], 0.0) : 0.0;
        const double tr = wave_sum_d(in ? Kp[tid * ks + tid] : 0.0);
        if (lane == 0) wred[wave] = tr;
    }
    __syncthreads();
    if (tid == 0) {
        double tr = 0.0;
        for (int w = 0; w < kWords; ++w) tr += wred[w];
        dl_s = delta * tr / a + 1e-300; done_s = 0; piv_s = 0;
        if (!(tr > 0.0)) done_s = 3;       // K = 0: no residence answers to these rows
    }
    __syncthreads();
    if (done_s == 3) {                      // uniform: leave the multipliers where they are
        if (tid < kAmax) yo[tid] = tid < a ? cy[tid] : 0.0;
        if (tid == 0) info[t] = 0;
        return;
    }
    const double dl = dl_s;
    auto kp = [&](int i, int j) -> double {
        return s_s[i] * s_s[j] * Kp[i * ks + j] + (i == j ? dl : 0.0);
    };
    // out_i = sum_j K'_ij u_j : 2 lanes per row, kAmax/2 columns each
    const int mi = tid >> 1, mp = tid & 1;
    // (sixteen entries of the row in flight per round trip: one load per trip through the loop made this the
    // kernel -- 64 dependent L2 latencies per product, a product per pivoting round; same order of the sum)
    auto matvec_row = [&]() -> double {
        double acc = 0.0;
        if (mi < a) {
            const int j0 = mp * (kAmax / 2);
            for (int jb = j0; jb < j0 + kAmax / 2 && jb < a; jb += 16) {
                double kv[16];
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) kv[jj] = Kp[mi * ks + (jb + jj < a ? jb + jj : mi)];
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    const int j = jb + jj;
                    if (j < a) acc += (s_s[mi] * s_s[j] * kv[jj] + (mi == j ? dl : 0.0)) * u_s[j];
                }
            }
        }
        acc += __shfl_xor(acc, 1, 64);
        return acc;
    };
    {
        const double ku = matvec_row();
        if (mp == 0) c_s[mi] = mi < a ? s_s[mi] * cg[mi] + ku : 0.0;
    }
    if (tid < kAmax) {
        // (started from every candidate basic instead -- the newly admitted rows are all violated at u_cur -- the
        // feeder's 15 iterations took 12.6 ms instead of 10.1: more rounds, not fewer; r05)
        const unsigned long long B0 = __ballot(tid < a && u_s[tid] > 0.0);
        if (lane == 0) Bsh[wave] = B0;
    }
    __syncthreads();
    if (tid < kAmax) {
        const double cm = wave_max_d(fabs(c_s[tid]));
        if (lane == 0) wred[wave] = cm;
    }
    __syncthreads();
    if (tid == 0) {
        double cm = 0.0;
        for (int w = 0; w < kWords; ++w) cm = fmax(cm, wred[w]);
        tolw_s = 1e-13 * cm;
    }
    int ninf = kAmax + 1, pcount = 3;       // thread 0's pivoting state
    BPP_STAMP(2);
    int round_ = 0;
    for (;;) {
        __syncthreads();
        BPP_STAMP(3 + 4 * round_);
        unsigned long long B[kWords];
        int nb = 0;
#pragma unroll
        for (int w = 0; w < kWords; ++w) { B[w] = Bsh[w]; nb += __popcll(B[w]); }
        if (tid < kAmax && ((B[wave] >> lane) & 1ull)) {
            int pos = __popcll(B[wave] & ((1ull << lane) - 1ull));
            for (int w = 0; w < wave; ++w) pos += __popcll(B[w]);
            bl[pos] = tid;
        }
        __syncthreads();
        // K'_BB = L D L^T (unit lower L below the diagonal of Ls, 1 / D in idk_s), pivots floored at dl 1e-6 as the
        // Cholesky form floored its squares.  The trailing matrix lives in REGISTERS -- thread (ti, tj) owns the elements
        // (i, j) with i % 16 = ti, j % 16 = tj -- and is eliminated FOUR columns per barrier: the threads that own a
        // block's columns publish them (two alternating LDS buffers), every thread factors the 4 x 4 diagonal block for
        // itself, forms the panel entries of its own rows and columns (independent LDS reads) and applies the rank-4
        // update to its block with straight-line FMAs; the finished columns go to Ls, scaled, for the solves.
        // Reciprocals by v_rcp_f64 + two Newton steps.  (Stage stamps, 50 rows: in-LDS Cholesky with three barriers, a
        // square root and a division per column ~45 us per round; one column per barrier with the matrix in registers
        // 24 us -- the barrier, the broadcast of the pivot and its reciprocal chain per column; r04.)
        auto factor = [&](auto nt_tag) {
            constexpr int NT = decltype(nt_tag)::value;
            const int ti = tid >> 4, tj = tid & 15;
            // This thread's block of K'_BB: the candidates' numbers and signs of its rows and columns, then EVERY entry
            // requested (clamped addresses, no condition around a load) before the first is used, masked afterwards.
            // (As `cond ? kp(bl[i], bl[j]) : 0` each entry was a branch around two dependent LDS reads and a flat load
            // with its own s_waitcnt: 16 or 64 memory round trips one after the other at the top of every factorisation.)
            double am[NT][NT];
            {
                int bi[NT], bj[NT];
#pragma unroll
                for (int x = 0; x < NT; ++x) {
                    const int i = ti + 16 * x, j = tj + 16 * x;
                    bi[x] = bl[i < nb ? i : 0];
                    bj[x] = bl[j < nb ? j : 0];
                }
                if (klds) {
#pragma unroll
                    for (int x = 0; x < NT; ++x)
#pragma unroll
                        for (int y = 0; y < NT; ++y) am[x][y] = y <= x ? Kl[bi[x] * kKl + bj[y]] : 0.0;
                } else {
#pragma unroll
                    for (int x = 0; x < NT; ++x)
#pragma unroll
                        for (int y = 0; y < NT; ++y) am[x][y] = y <= x ? Kg[bi[x] * kAmax + bj[y]] : 0.0;
                }
                double si[NT], sj[NT];
#pragma unroll
                for (int x = 0; x < NT; ++x) { si[x] = s_s[bi[x]]; sj[x] = s_s[bj[x]]; }
#pragma unroll
                for (int x = 0; x < NT; ++x)
#pragma unroll
                    for (int y = 0; y < NT; ++y) {
                        const int i = ti + 16 * x, j = tj + 16 * y;
                        am[x][y] = (i < nb && j <= i) ? si[x] * sj[y] * am[x][y] + (bi[x] == bj[y] ? dl : 0.0) : 0.0;
                    }
            }
            const double floor_ = dl * 1e-6;
#pragma unroll
            for (int yb = 0; yb < NT; ++yb) {
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int k0 = 16 * yb + 4 * q4;
                    if (k0 >= nb) break;                              // uniform
                    double (*const col)[kAmax] = col_s[(k0 >> 2) & 1];
                    if ((tj >> 2) == q4) {                            // the owners of columns k0 .. k0 + 3: rows ti + 16 x
#pragma unroll
                        for (int x = 0; x < NT; ++x) col[tj & 3][ti + 16 * x] = am[x][yb];
                    }
                    __syncthreads();
                    // the diagonal block, factored by every thread for itself: D4 = L4 diag(d) L4^T
                    double D4[4][4], L4[4][4], r4[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c <= r; ++c) {
                            const bool in = k0 + r < nb;              // (columns beyond the set: identity)
                            D4[r][c] = in ? col[c][k0 + r] : (r == c ? 1.0 : 0.0);
                        }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const double dk = fmax(D4[c][c], floor_);
                        double r = __builtin_amdgcn_rcp(dk);
                        r = __builtin_fma(__builtin_fma(-dk, r, 1.0), r, r);
                        r = __builtin_fma(__builtin_fma(-dk, r, 1.0), r, r);
                        r4[c] = r;
#pragma unroll
                        for (int rr = c + 1; rr < 4; ++rr) L4[rr][c] = D4[rr][c] * r;
#pragma unroll
                        for (int rr = c + 1; rr < 4; ++rr)
#pragma unroll
                            for (int cc = c + 1; cc <= rr; ++cc) D4[rr][cc] -= L4[rr][c] * D4[cc][c];
                    }
                    // the panel: for a row i below the block, w_i = L4^-1 A_i,block (the unscaled entries l_ic d_c) and
                    // l_i = w_i / d; zero for rows inside or above the block
                    double li[NT][4], wj[NT][4];
#pragma unroll
                    for (int x = 0; x < NT; ++x) {
                        const int i = ti + 16 * x, j = tj + 16 * x;
                        double pi[4], pj[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            // (masked loads: the compiler's form -- a branch around each -- measured faster here than
                            // all reads unconditional and waited for together: rows beyond the set are skipped whole)
                            pi[c] = (i > k0 + 3 && i < nb) ? col[c][i] : 0.0;
                            pj[c] = (j > k0 + 3 && j < nb) ? col[c][j] : 0.0;
                        }
#pragma unroll
                        for (int c = 1; c < 4; ++c)
#pragma unroll
                            for (int cc = 0; cc < c; ++cc) { pi[c] -= L4[c][cc] * pi[cc]; pj[c] -= L4[c][cc] * pj[cc]; }
#pragma unroll
                        for (int c = 0; c < 4; ++c) { li[x][c] = pi[c] * r4[c]; wj[x][c] = pj[c]; }
                    }
#pragma unroll
                    for (int x = 0; x < NT; ++x)
#pragma unroll
                        for (int y = 0; y < NT; ++y)
#pragma unroll
                            for (int c = 0; c < 4; ++c) am[x][y] -= li[x][c] * wj[y][c];
                    // the finished columns for the solves: l_ic of the rows below the block (their owners' threads hold
                    // them), the block's own sub-diagonal, 1 / d
                    if ((tj >> 2) == q4) {
                        const int c = tj & 3;
#pragma unroll
                        for (int x = 0; x < NT; ++x) {
                            const int i = ti + 16 * x;
                            // (li[x][c] by selects: a register array indexed by a lane's own value lives in scratch
                            // memory -- every panel step paid sixteen scratch stores and a dependent load)
                            double lv = li[x][0];
#pragma unroll
                            for (int cc = 1; cc < 4; ++cc) lv = cc == c ? li[x][cc] : lv;
                            if (i > k0 + 3 && i < nb) Ls(i, k0 + c) = lv;
                        }
                    }
                    if (tid < 16) {
                        const int r = tid >> 2, c = tid & 3;
                        if (c < r && k0 + r < nb) {
                            double v = 0.0;
#pragma unroll
                            for (int rr = 1; rr < 4; ++rr)
#pragma unroll
                                for (int cc = 0; cc < rr; ++cc) v = (rr == r && cc == c) ? L4[rr][cc] : v;
                            Ls(k0 + r, k0 + c) = v;
                        }
                        if (c == r && k0 + r < nb) {
                            double v = r4[0];
#pragma unroll
                            for (int cc = 1; cc < 4; ++cc) v = cc == c ? r4[cc] : v;
                            idk_s[k0 + r] = v;
                        }
                    }
                }
            }
        };
        if (nb <= 64) factor(std::integral_constant<int, 4>{});
        else factor(std::integral_constant<int, 8>{});
        __syncthreads();
        BPP_STAMP(4 + 4 * round_);
        // triangular solves in wavefront 0: lane l holds components l and l + 64; the pivot component reaches the
        // others through v_readlane (k is uniform: no trip through the LDS crossbar per step), and the entries of the
        // next four columns of L are requested before the current four are used.  Branch-free: every LDS read is
        // made (clamped address) and masked afterwards, steps beyond the set multiply zeros, and the components
        // below / from 64 are two loops -- as ternaries and `if (k < nb)` the compiler turned each read, each
        // readlane and each step into a branch of its own: 326 cycles per column, 6.8 us per solve at 50 rows
        // (stage stamps 20-22, r04).
        if (tid < 64) {
            double v0 = lane < nb ? c_s[bl[lane]] : 0.0;
            double v1 = lane + 64 < nb ? c_s[bl[lane + 64]] : 0.0;
            auto lane_of = [](double v, int k) -> double {            // v of lane k (k uniform, 0..63)
                const long long bb = __double_as_longlong(v);
                const int lo = __builtin_amdgcn_readlane((int)bb, k);
                const int hi = __builtin_amdgcn_readlane((int)(bb >> 32), k);
                return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
            };
            // (Four pivots per step -- their components and the block's six entries of L read across with twenty
            // independent readlanes, the 4 x 4 triangle solved by every lane for itself -- measured slower: 8.4 us
            // against 6.5 for both solves at 59 rows; r05.)
            // L y = c_B (unit diagonal): column k of L, rows lane and lane + 64
            auto colf = [&](int k, double (&a0)[4], double (&a1)[4]) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int kc = k + c < kAmax ? k + c : kAmax - 1;
                    const double x0 = Ls(lane, kc), x1 = Ls(lane + 64, kc);      // (always in bounds; masked below)
                    a0[c] = (k + c < nb && lane > k + c && lane < nb) ? x0 : 0.0;
                    a1[c] = (k + c < nb && lane + 64 < nb) ? x1 : 0.0;      // (lane + 64 > k + c below 64; tested from 64 on)
                }
            };
            double c0[4], c1[4], n0[4], n1[4];
            if (round_ == 0) BPP_STAMP(20);
            colf(0, c0, c1);
            const int nlo = nb < 64 ? nb : 64;
            for (int kb = 0; kb < nlo; kb += 4) {                     // pivots among the components below 64
                colf(kb + 4, n0, n1);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double yk = lane_of(v0, (kb + c) & 63);     // (a step beyond the set: its column is masked to zero)
                    v0 -= c0[c] * yk;                                 // (zero where lane <= k)
                    v1 -= c1[c] * yk;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) { c0[c] = n0[c]; c1[c] = n1[c]; }
            }
            for (int kb = 64; kb < nb; kb += 4) {                     // ... from 64 on: they move components from 64 on only
#pragma unroll
                for (int c = 0; c < 4; ++c) c1[c] = lane + 64 > kb + c ? c1[c] : 0.0;
                colf(kb + 4, n0, n1);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double yk = lane_of(v1, (kb + c) & 63);
                    v1 -= c1[c] * yk;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) { c0[c] = n0[c]; c1[c] = n1[c]; }
            }
            if (round_ == 0) BPP_STAMP(21);
            // D^-1, then L^T x = y: row k of L, columns lane and lane + 64
            v0 *= lane < nb ? idk_s[lane] : 0.0;
            v1 *= lane + 64 < nb ? idk_s[lane + 64] : 0.0;
            auto rowb = [&](int k, double (&a0)[4], double (&a1)[4]) {      // rows k, k - 1, k - 2, k - 3
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int kc = k - c > 0 ? k - c : 0;
                    const double x0 = Ls(kc, lane), x1 = Ls(kc, lane + 64);
                    a0[c] = (k - c >= 0 && k - c < nb && lane < k - c) ? x0 : 0.0;
                    a1[c] = (k - c >= 0 && k - c < nb && lane + 64 < k - c) ? x1 : 0.0;
                }
            };
            // (kb starts at the last row of the block of four that holds nb - 1, so that no block straddles 64; rows
            // beyond the set hold whatever an earlier round left: masked)
            const int ktop = ((nb - 1) | 3);                          // last row of the block of four that holds nb - 1
            rowb(ktop, c0, c1);
            for (int kb = ktop; kb >= 64; kb -= 4) {                  // pivots among the components from 64 on
                rowb(kb - 4, n0, n1);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double xk = lane_of(v1, (kb - c) & 63);     // (a row beyond the set: masked to zero, x_k = 0 anyway)
                    v0 -= c0[c] * xk;                                 // (zero where lane >= k)
                    v1 -= c1[c] * xk;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) { c0[c] = n0[c]; c1[c] = n1[c]; }
            }
            for (int kb = ktop < 63 ? ktop : 63; kb >= 0; kb -= 4) {  // ... below 64: they move components below 64 only
                rowb(kb - 4, n0, n1);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double xk = lane_of(v0, (kb - c) & 63);
                    v0 -= c0[c] * xk;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) { c0[c] = n0[c]; c1[c] = n1[c]; }
            }
            if (round_ == 0) BPP_STAMP(22);
            u_s[lane] = 0.0; u_s[lane + 64] = 0.0;
            __builtin_amdgcn_wave_barrier();
            if (lane < nb) u_s[bl[lane]] = v0;
            if (lane + 64 < nb) u_s[bl[lane + 64]] = v1;
        }
        __syncthreads();
        BPP_STAMP(5 + 4 * round_);
        {
            const double ku = matvec_row();
            if (mp == 0) w_s[mi] = mi < a ? ku - c_s[mi] : 0.0;
        }
        BPP_STAMP(6 + 4 * round_);
        ++round_;
        if (tid < kAmax) {
            const double um = wave_max_d(fabs(u_s[tid]));
            if (lane == 0) wred[wave] = um;
        }
        __syncthreads();
        if (tid == 0) {
            double um = 0.0;
            for (int w = 0; w < kWords; ++w) um = fmax(um, wred[w]);
            tolu_s = 1e-13 * um;
        }
        __syncthreads();
        if (tid < kAmax) {
            const bool inB = (B[wave] >> lane) & 1ull;
            const bool bad = tid < a && (inB ? (u_s[tid] < -tolu_s) : (w_s[tid] < -tolw_s));
            const unsigned long long V = __ballot(bad);
            if (lane == 0) Vsh[wave] = V;
        }
        __syncthreads();
        if (tid == 0) {
            int nv = 0, top = -1;
            for (int w = 0; w < kWords; ++w) {
                nv += __popcll(Vsh[w]);
                if (Vsh[w]) top = w * 64 + 63 - __clzll((long long)Vsh[w]);
            }
            const int pv = ++piv_s;
            if (nv == 0) done_s = 1;
            else if (pv >= max_pivots) done_s = 2;
            else if (nv < ninf || pcount > 0) {
                if (nv < ninf) { ninf = nv; pcount = 3; } else --pcount;
                for (int w = 0; w < kWords; ++w) Bsh[w] = B[w] ^ Vsh[w];
            } else {
                Bsh[top >> 6] = B[top >> 6] ^ (1ull << (top & 63));
            }
        }
        __syncthreads();
        if (done_s) break;
    }
    if (tid < kAmax) yo[tid] = tid < a ? s_s[tid] * fmax(u_s[tid], 0.0) : 0.0;
    if (tid == 0) info[t] = done_s == 1 ? piv_s : -piv_s;
    BPP_STAMP(31);
#ifdef REVS_BPP_STAMPS
    if (tid == 0) { g_bpp_stamps[blockIdx.x][30] = (double)a; g_bpp_stamps[blockIdx.x][29] = (double)piv_s;
                    g_bpp_stamps[blockIdx.x][27] = (double)clock64(); }     // (shader clock ticks: [28] at the start)
#endif
}
#if defined(REVS_KV_STAMPS) && defined(REVS_ROWS_STAMPS)
extern "C" int revs_tuning_rows_stamps(double *out_host) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_rows_stamps), sizeof(double) * 256 * 32) == hipSuccess ? 0 : -1;
}
#endif
#ifdef REVS_BPP_STAMPS
extern "C" int revs_tuning_bpp_stamps(double *out_host) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_bpp_stamps), sizeof(double) * 256 * 32) == hipSuccess ? 0 : -1;
}
#endif

// The LCP of at most 8 candidates by one thread, everything in registers (loops sized by the
// template parameter): same regularisation, tolerances and pivoting rule as op_dual_bpp_kernel.
template <int A>
__device__ __forceinline__ void small_bpp(const double (&Ks)[8][8], const double *cs,
                                          const double *cg, const double *cy, double delta,
                                          int max_pivots, double *yo, int32_t *info, const int nfill = kAmax) {
    // One thread, everything in registers: rows >= a and non-basic rows are masked to the
    // identity, so every loop below has compile-time bounds.
    double s[A], c[A], u[A], cyv[A], K[A][A];
    double tr = 0.0;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        s[i] = cs[i];
        cyv[i] = cy[i];
        u[i] = fmax(s[i] * cyv[i], 0.0);
        tr += Ks[i][i];
    }
    if (!(tr > 0.0)) {                      // K = 0: leave the multipliers where they are
        for (int i = 0; i < nfill; ++i) yo[i] = i < A ? cy[i] : 0.0;
        *info = 0;
        return;
    }
    const double dl = delta * tr / A + 1e-300;
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
        for (int j = 0; j < A; ++j)
            K[i][j] = s[i] * s[j] * Ks[i][j] + (i == j ? dl : 0.0);
    double cmax = 0.0;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        double ku = 0.0;
#pragma unroll
        for (int j = 0; j < A; ++j) ku += K[i][j] * u[j];
        c[i] = s[i] * cg[i] + ku;
        cmax = fmax(cmax, fabs(c[i]));
    }
    const double tolw = 1e-13 * cmax;
    unsigned B = 0;
#pragma unroll
    for (int i = 0; i < A; ++i) B |= (u[i] > 0.0) ? (1u << i) : 0u;
    int ninf = A + 1, pcount = 3, piv = 0, done = 0;
    while (!done) {
        // K'_BB = L D L^T (rows outside B masked to the identity), pivots floored as the Cholesky form floors them
        // (d >= dl 1e-6).  No square roots, and the reciprocals by v_rcp_f64 + two Newton steps: the one thread that
        // runs this is a chain of dependent f64 instructions, and an IEEE sqrt + division per pivot (37 of them) was
        // half of it.
        double L[A][A], z[A], idk[A];
#pragma unroll
        for (int k = 0; k < A; ++k)
#pragma unroll
            for (int l = 0; l <= k; ++l)
                L[k][l] = (((B >> k) & 1u) && ((B >> l) & 1u)) ? K[k][l] : (k == l ? 1.0 : 0.0);
#pragma unroll
        for (int k = 0; k < A; ++k) {
            const double dk = fmax(L[k][k], dl * 1e-6);
            double r = __builtin_amdgcn_rcp(dk);
            r = __builtin_fma(__builtin_fma(-dk, r, 1.0), r, r);
            r = __builtin_fma(__builtin_fma(-dk, r, 1.0), r, r);
            idk[k] = r;
            double ck[A];                    // column k below the diagonal, unscaled (= l_ik d_k)
#pragma unroll
            for (int i = k + 1; i < A; ++i) { ck[i] = L[i][k]; L[i][k] = ck[i] * r; }
#pragma unroll
            for (int i = k + 1; i < A; ++i)
#pragma unroll
                for (int j = k + 1; j <= i; ++j) L[i][j] -= L[i][k] * ck[j];
        }
#pragma unroll
        for (int k = 0; k < A; ++k) {        // L y = c_B
            double v = ((B >> k) & 1u) ? c[k] : 0.0;
#pragma unroll
            for (int l = 0; l < k; ++l) v -= L[k][l] * z[l];
            z[k] = v;
        }
#pragma unroll
        for (int k = A - 1; k >= 0; --k) {   // L^T x = D^-1 y
            double v = z[k] * idk[k];
#pragma unroll
            for (int l = k + 1; l < A; ++l) v -= L[l][k] * z[l];
            z[k] = v;
        }
        double umax = 0.0;
#pragma unroll
        for (int i = 0; i < A; ++i) { u[i] = ((B >> i) & 1u) ? z[i] : 0.0; umax = fmax(umax, fabs(u[i])); }
        unsigned V = 0;
#pragma unroll
        for (int i = 0; i < A; ++i) {
            double ku = 0.0;
#pragma unroll
            for (int j = 0; j < A; ++j) ku += K[i][j] * u[j];
            const double wi = ku - c[i];
            const bool inB = (B >> i) & 1u;
            if (inB ? (u[i] < -1e-13 * umax) : (wi < -tolw)) V |= 1u << i;
        }
        ++piv;
        const int nv = __popc(V);
        if (nv == 0) done = 1;
        else if (piv >= max_pivots) done = 2;
        else if (nv < ninf) { ninf = nv; pcount = 3; B ^= V; }
        else if (pcount > 0) { --pcount; B ^= V; }
        else B ^= 1u << (31 - __clz((int)V));
    }
#pragma unroll
    for (int i = 0; i < A; ++i) yo[i] = s[i] * fmax(u[i], 0.0);
    for (int i = A; i < nfill; ++i) yo[i] = 0.0;
    *info = done == 1 ? piv : -piv;
}


// ---- the model problem when a slot has at most 8 candidates (the binding steady state:
// one to three multipliers per slot) -- Gram matrix and pivoting in ONE small kernel instead
// of the tiled Gram kernel + the 128-row pivoting kernel (37 us -> ~8 us per Newton step).
// The whole workgroup accumulates the 36 upper-triangular entries of R_F N R_F^T over the
// columns (coalesced row reads), one thread then runs the same block principal pivoting
// (same regularisation, tolerances and backup rule) on the 8 x 8 problem in LDS.
constexpr int kSmall = 8;
__device__ __forceinline__ void small_model_body(
        const int t, int m, int T, const double *__restrict__ R, const double *__restrict__ Nn,
        const int64_t *__restrict__ cidx, const int32_t *__restrict__ ccnt,
        const double *__restrict__ cval, double inv_kappa, double delta, int max_pivots,
        double *__restrict__ Kall, double *__restrict__ yhat, int32_t *__restrict__ info, const int nes = 1) {
    const int tid = threadIdx.x;
    const int a = ccnt[t];
    int64_t f[kSmall];                      // (entries beyond the count hold row 0: loadable)
#pragma unroll
    for (int i = 0; i < kSmall; ++i) f[i] = cidx[(int64_t)t * kAmax + i];
    const double *cs = cval + (int64_t)t * 3 * kAmax, *cg = cs + kAmax, *cy = cg + kAmax;
    double *yo = yhat + (int64_t)t * kAmax;
    if (a <= 0 || a > kSmall) {             // uniform; a > 8 means the caller's count was wrong
        if (tid < kAmax) yo[tid] = cy[tid];
        if (tid == 0) info[t] = a <= 0 ? 0 : -999;
        return;
    }
    // (the solve's inputs are fetched now, not when thread 0 gets to them)
    __shared__ double cvs[3][kSmall];
    if (tid < 3 * kSmall) cvs[tid / kSmall][tid % kSmall] = cs[(tid / kSmall) * kAmax + tid % kSmall];
#pragma unroll
    for (int i = 0; i < kSmall; ++i) f[i] = i < a ? f[i] : -1;
    double acc[kSmall * (kSmall + 1) / 2];
#pragma unroll
    for (int p = 0; p < kSmall * (kSmall + 1) / 2; ++p) acc[p] = 0.0;
    REVS_KVS(t, 12);
#pragma unroll 8
    for (int mm = tid; mm < m; mm += 256) {
        const double nv = nes == 4 ? Nn[4 * ((int64_t)t * m + mm)] : Nn[(int64_t)mm * T + t];
        double r[kSmall];
#pragma unroll
        for (int i = 0; i < kSmall; ++i) r[i] = f[i] >= 0 ? R[f[i] * m + mm] : 0.0;
        int p = 0;
#pragma unroll
        for (int i = 0; i < kSmall; ++i) {
            const double rn = r[i] * nv;
#pragma unroll
            for (int j = i; j < kSmall; ++j) acc[p++] += rn * r[j];
        }
    }
    __shared__ double part[4][kSmall * (kSmall + 1) / 2];
    __shared__ double Ks[kSmall][kSmall];
    REVS_KVS(t, 13);
#pragma unroll
    for (int p = 0; p < kSmall * (kSmall + 1) / 2; ++p) {
        const double v = wave_sum_d(acc[p]);
        if ((tid & 63) == 0) part[tid >> 6][p] = v;
    }
    __syncthreads();
    REVS_KVS(t, 14);
    if (tid < kSmall * kSmall) {
        const int i = tid / kSmall, j = tid % kSmall;
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        const int p = lo * kSmall - lo * (lo - 1) / 2 + (hi - lo);
        const double v = (((part[0][p] + part[1][p]) + part[2][p]) + part[3][p]) * inv_kappa;
        Ks[i][j] = v;
        if (i < a && j < a) Kall[(int64_t)t * kAmax * kAmax + i * kAmax + j] = v;
    }
    __syncthreads();
    REVS_KVS(t, 15);
    if (tid != 0) return;
    switch (a) {                            // one thread; loops sized by the candidate count
        case 1: small_bpp<1>(Ks, cvs[0], cvs[1], cvs[2], delta, max_pivots, yo, info + t); break;
        case 2: small_bpp<2>(Ks, cvs[0], cvs[1], cvs[2], delta, max_pivots, yo, info + t); break;
        case 3: small_bpp<3>(Ks, cvs[0], cvs[1], cvs[2], delta, max_pivots, yo, info + t); break;
        case 4: small_bpp<4>(Ks, cvs[0], cvs[1], cvs[2], delta, max_pivots, yo, info + t); break;
        case 5: small_bpp<5>(Ks, cvs[0], cvs[1], cvs[2], delta, max_pivots, yo, info + t); break;
        case 6: small_bpp<6>(Ks, cvs[0], cvs[1], cvs[2], delta, max_pivots, yo, info + t); break;
        case 7: small_bpp<7>(Ks, cvs[0], cvs[1], cvs[2], delta, max_pivots, yo, info + t); break;
        default: small_bpp<8>(Ks, cvs[0], cvs[1], cvs[2], delta, max_pivots, yo, info + t); break;
    }
    REVS_KVS(t, 16);
}

__global__ __launch_bounds__(256) void op_dual_model_small_kernel(
        int m, int T, const double *__restrict__ R, const double *__restrict__ Nn,
        const int64_t *__restrict__ cidx, const int32_t *__restrict__ ccnt,
        const double *__restrict__ cval, double inv_kappa, double delta, int max_pivots,
        double *__restrict__ Kall, double *__restrict__ yhat, int32_t *__restrict__ info) {
    REVS_KVS_BEGIN(nullptr);
    small_model_body(blockIdx.x, m, T, R, Nn, cidx, ccnt, cval, inv_kappa, delta, max_pivots, Kall,
                     yhat, info);
}

// y_trial[cand] = y + alpha_t (yhat - y); lin_out[8 t] = grad . (y_trial - y)
// (first wavefront of the workgroup only past the copy; NT = the workgroup's threads)
template <int NT>
__device__ __forceinline__ void dual_step_body(
        const int t, int T, const int64_t *__restrict__ cidx, const int32_t *__restrict__ ccnt,
        const double *__restrict__ cval, const double *__restrict__ yhat, const double al,
        double *__restrict__ ytrial, double *__restrict__ lin_out,
        const double *__restrict__ ycopy, int m) {
    const int a = ccnt[t];
    if (ycopy) {                            // y_trial[., t] = y[., t] first (this slot's column only)
        for (int r = threadIdx.x; r < m; r += NT) ytrial[(int64_t)r * T + t] = ycopy[(int64_t)r * T + t];
        __syncthreads();
    }
    if (threadIdx.x >= 64) return;
    const double *cs = cval + (int64_t)t * 3 * kAmax, *cg = cs + kAmax, *cy = cg + kAmax;
    double lin = 0.0;
    for (int i = threadIdx.x; i < a; i += 64) {
        const double yo = cy[i], yh = yhat[(int64_t)t * kAmax + i];
        // a full step lands exactly on yhat, so a multiplier that leaves is exactly zero
        const double yn = al == 1.0 ? yh : (al == 0.0 ? yo : yo + al * (yh - yo));
        ytrial[cidx[(int64_t)t * kAmax + i] * T + t] = yn;
        lin += cg[i] * (yn - yo);
    }
    lin = wave_sum_d(lin);
    if (threadIdx.x == 0) lin_out[t * 8] = lin;
}

__global__ __launch_bounds__(64) void op_dual_step_kernel(
        int T, const int64_t *__restrict__ cidx, const int32_t *__restrict__ ccnt,
        const double *__restrict__ cval, const double *__restrict__ yhat,
        const double *__restrict__ alpha, double *__restrict__ ytrial,
        double *__restrict__ lin_out, const double *__restrict__ stats_prev, double scale,
        double eps, const double *__restrict__ ycopy, int m) {
    const int t = blockIdx.x;
    // (stats_prev: full step for the slots whose rows are not yet within tolerance, decided
    // here exactly as the host would: the caller has not read those stats yet)
    const double al = stats_prev ? (stats_prev[t * 8] / scale > eps ? 1.0 : 0.0) : alpha[t];
    dual_step_body<64>(t, T, cidx, ccnt, cval, yhat, al, ytrial, lin_out, ycopy, m);
}

// The step with the copy of the slot's column of multipliers in front (y_trial[., t] = y[., t], eight loads in flight per
// thread), one launch: the native Newton loop made a device-to-device copy of all of y and then the step -- a dispatch
// and a kernel boundary per Newton iteration.
__global__ __launch_bounds__(256) void op_dual_step_copy_kernel(
        int T, const int64_t *__restrict__ cidx, const int32_t *__restrict__ ccnt,
        const double *__restrict__ cval, const double *__restrict__ yhat,
        const double *__restrict__ alpha, double *__restrict__ ytrial,
        double *__restrict__ lin_out, const double *__restrict__ ycopy, int m) {
    const int t = blockIdx.x;
    for (int r0 = threadIdx.x; r0 < m; r0 += 8 * 256) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = r0 + 256 * q < m ? ycopy[(int64_t)(r0 + 256 * q) * T + t] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (r0 + 256 * q < m) ytrial[(int64_t)(r0 + 256 * q) * T + t] = v[q];
    }
    __syncthreads();
    dual_step_body<256>(t, T, cidx, ccnt, cval, yhat, alpha[t], ytrial, lin_out, nullptr, m);
}

// Selection, small model and full/zero step of one slot in ONE workgroup (the chained Newton
// iteration of the binding steady state: three launches and two reloads of the candidate
// lists less).  A slot with more than 8 candidates gets info = -999 from the model part, as
// from op_dual_model_small_kernel, and the caller falls back.
struct FusedArgs {
    const double *R, *Nn, *ycopy;
    double inv_kappa, delta, scale, eps;
    int max_pivots;
    double *Kall, *yhat, *ytrial, *lin_out;
    int32_t *info;
    int nes = 1;                   // element stride of Nn (ChainKvSide::es)
};
__global__ __launch_bounds__(256) void op_dual_select_model_step_kernel(const SelectArgs sa,
                                                                        const FusedArgs fa) {
    REVS_KVS_BEGIN(nullptr);
    const int t = blockIdx.x;
    const double rmax = dual_select_body<true>(sa, t);
    __syncthreads();                        // this workgroup's lists, written to global memory
    small_model_body(t, sa.m, sa.T, fa.R, fa.Nn, sa.cidx, sa.ccnt, sa.cval, fa.inv_kappa, fa.delta,
                     fa.max_pivots, fa.Kall, fa.yhat, fa.info);
    __syncthreads();                        // yhat (thread 0)
    dual_step_body<256>(t, sa.T, sa.cidx, sa.ccnt, sa.cval, fa.yhat,
                        rmax / fa.scale > fa.eps ? 1.0 : 0.0, fa.ytrial, fa.lin_out, fa.ycopy, sa.m);
}

// ... and with the slot's rows by the tree form in front (the whole operator side of a chained
// Newton iteration between two home passes is then ONE launch of T workgroups)
__global__ __launch_bounds__(256) void op_tree_select_model_step_kernel(const TreeRowsArgs ta, const SelectArgs sa,
                                                                        const FusedArgs fa) {
    REVS_KVS_BEGIN(nullptr);
    extern __shared__ double tree_lds[];
    const int t = blockIdx.x;
    tree_rows_body(ta, t, tree_lds);
    __syncthreads();
    const double rmax = dual_select_body<true>(sa, t);
    __syncthreads();
    small_model_body(t, sa.m, sa.T, fa.R, fa.Nn, sa.cidx, sa.ccnt, sa.cval, fa.inv_kappa, fa.delta,
                     fa.max_pivots, fa.Kall, fa.yhat, fa.info);
    __syncthreads();
    dual_step_body<256>(t, sa.T, sa.cidx, sa.ccnt, sa.cval, fa.yhat,
                        rmax / fa.scale > fa.eps ? 1.0 : 0.0, fa.ytrial, fa.lin_out, fa.ycopy, sa.m);
}

// The folded chain's operator launch (revs_plan_chain_fold_run): behind the sweep of iteration k,
// workgroups [0, T) judge the trial of iteration k (rows by the tree form and selection on the sums
// the sweep folded: the verdict the host polls), workgroups [T, 2T) already run rows, selection,
// small model and step of iteration k + 1 on the sums of the SAME multipliers on the new state --
// speculative only in that the host may reject iteration k, in which case nobody reads them.  All
// workgroups clear their share of the two sum arrays the NEXT sweep accumulates into.
// d[t][node] (slot-major: a slot's workgroup writes its column contiguously) = (R^T y)[node][t] / kappa
// from the rows of slot t's candidate list, in the list's order
// -- op_dual_eval_kernel's SparseD loop, term for term -- and with the rows that carry a multiplier
// taken in ascending row order: the order of the list the NEXT evaluation's home pass would read (a
// selection lists the rows with y != 0 first, by row).  (Lists of more than 8 rows never pass the
// chain's acceptance test: list order will do for the second one.)
__device__ __forceinline__ void chain_shifts_body(const int t, int m, int T, const double *__restrict__ R,
                                                  const int64_t *__restrict__ cidx, const int32_t *__restrict__ ccnt,
                                                  const double *__restrict__ y, double inv_kappa,
                                                  double *__restrict__ sh_a, double *__restrict__ sh_b) {
    const int cnt = ccnt[t];
    const int64_t *si = cidx + (int64_t)t * kAmax;
    if (cnt <= 8) {
        // The list and its multipliers do not depend on the node: fetched once, compacted to the rows
        // that carry a multiplier (a term with y = 0 adds exactly nothing) and sorted once -- all
        // uniform over the workgroup; per node only the loads of R are left, all independent.
        int64_t f[8], fl[8], fs[8];
        double yv[8], yl[8], ys[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = k < cnt ? si[k] : 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) yv[k] = k < cnt ? y[f[k] * T + t] : 0.0;
        int ns = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { fl[k] = 0; yl[k] = 0.0; }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (yv[k] != 0.0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) if (q == ns) { fl[q] = f[k]; yl[q] = yv[k]; }
                ++ns;
            }
        }
        int64_t last = -1;
        bool same = true;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            int64_t best = 0x7FFFFFFFFFFFFFFFll;
            double yb = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < ns && fl[k] > last && fl[k] < best) { best = fl[k]; yb = yl[k]; }
            fs[r] = r < ns ? best : 0;
            ys[r] = r < ns ? yb : 0.0;
            same = same && (r >= ns || fs[r] == fl[r]);
            last = best;
        }
        // (eight nodes per pass with every load of the pass in flight before the first use: the loop is
        // latency, not bandwidth)
        REVS_KVS(t, 19);
        for (int n0 = threadIdx.x; n0 < m; n0 += 8 * 256) {
            double ra[8][4], rb[8][4];
            const int kk = ns < 4 ? ns : 4;         // rows beyond four (rare) go through the second loop below
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int node = n0 + 256 * i;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ra[i][k] = (k < kk && node < m) ? R[fl[k] * m + node] : 0.0;
                    rb[i][k] = (!same && k < kk && node < m) ? R[fs[k] * m + node] : 0.0;
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int node = n0 + 256 * i;
                if (node >= m) continue;
                double d = 0.0, ds = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < kk) { d = __builtin_fma(ra[i][k], yl[k], d); ds = __builtin_fma(rb[i][k], ys[k], ds); }
#pragma unroll
                for (int k = 4; k < 8; ++k)
                    if (k < ns) {
                        d = __builtin_fma(R[fl[k] * m + node], yl[k], d);
                        if (!same) ds = __builtin_fma(R[fs[k] * m + node], ys[k], ds);
                    }
                if (same) ds = d;
                sh_a[(int64_t)t * m + node] = d * inv_kappa;
                sh_b[(int64_t)t * m + node] = ds * inv_kappa;
            }
        }
        return;
    }
    for (int node = threadIdx.x; node < m; node += 256) {
        double d = 0.0;
        for (int i0 = 0; i0 < cnt; i0 += 8) {
            int64_t f[8];
            double rv[8], yv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = i0 + k < cnt ? si[i0 + k] : 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                rv[k] = R[f[k] * m + node];
                yv[k] = i0 + k < cnt ? y[f[k] * T + t] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) d = __builtin_fma(rv[k], yv[k], d);
        }
        sh_a[(int64_t)t * m + node] = d * inv_kappa;
        sh_b[(int64_t)t * m + node] = d * inv_kappa;
    }
}


// ---- rows and selection of one slot when the multipliers' support is known beforehand ----------------
// (the folded chain past its entry).  The multipliers a launch evaluates were produced by the previous
// launch's step from ITS candidate list, so every row with y != 0 is on that list (at most 8 rows here):
//  * no column of y is gathered -- a column of a [m][T] array is m cache lines, ~1.7 us of a
//    workgroup's L1 fill rate each -- its few entries are fetched through the list;
//  * the new list's head (the rows with a multiplier, ascending) is known before the voltages are: the
//    caller can fetch those rows of R at once (`after_pack`), a memory latency off the model step;
//  * the violated rows are collected as the voltages are judged and placed by rank (the order the
//    arg-max rounds of dual_select_body would take them), the support's entries are written straight
//    from the judging threads: no row-indexed staging of v / violations, no compaction scan.
// Same lists, stats and partial sums, bit for bit, as tree_rows_body + dual_select_body.

// (what is the same for every lane is told to the compiler: scalar registers, scalar arithmetic)
__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni_d(double v) {
    const long long b = __double_as_longlong(v);
    return __longlong_as_double(((long long)uni_i((int)(b >> 32)) << 32) | (unsigned int)uni_i((int)b));
}

// The previous list's rows with a multiplier, ascending: sup / ysup (uniform), their number.  Lane k < 8 of every
// wavefront holds entry k of the list (row my_f, multiplier my_y; nz: it counts): each ranks its row among the
// others by eight scalar broadcasts, and the q-th smallest is read back from the lane that holds it -- ~90
// instructions, no LDS, no barrier.  (The same selection as a uniform 8 x 8 sort of 64-bit rows: ~2.5 us of a
// latency-bound workgroup, r04 stamps.)
__device__ __forceinline__ int chain_support_rank(const int my_f, const double my_y, const bool nz,
                                                  long long (&sup)[8], double (&ysup)[8]) {
    const unsigned long long nzm = __ballot(nz);
    int rank = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int fj = __builtin_amdgcn_readlane(my_f, j);
        rank += (((nzm >> j) & 1ull) && fj < my_f) ? 1 : 0;      // (rows are distinct)
    }
    const long long yb = __double_as_longlong(my_y);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const unsigned long long mq = __ballot(nz && rank == q);
        const int src = __builtin_amdgcn_readfirstlane(mq ? __builtin_ctzll(mq) : 0);
        const int row = __builtin_amdgcn_readlane(my_f, src);
        const int lo = __builtin_amdgcn_readlane((int)yb, src), hi = __builtin_amdgcn_readlane((int)(yb >> 32), src);
        sup[q] = mq ? (long long)row : -1;
        ysup[q] = mq ? __longlong_as_double(((long long)hi << 32) | (unsigned int)lo) : 0.0;
    }
    return (int)__builtin_popcountll(nzm);
}

constexpr int kRankMax = 256;
struct ChainSelScratch {        // LDS of chain_rows_select_body
    double vval[kRankMax], vv[kRankMax];
    int vrow[kRankMax];
    int chosen[kAmax];
    double vsup[8];
    double rr[4][4];
    double best_v[2][4];
    int best_r[2][4];
    int vcount;
};

// Returns false (uniform; nothing published yet) when the previous list is unusable (overflowed, or longer
// than 8) or more than kRankMax rows are violated: the caller runs the general bodies instead.
// rmax_out: the slot's largest row residual; sup / ns_out: the support (the new list's head).
// Loads: two round trips -- {tree indices and weights, the previous list, first()'s} then {the list's
// multipliers, the gathers of node sums and dual terms, second()'s (which knows the support)}.
// The sums are the folded sweep's (TreeRowsArgs::es == 4: [T][m][4], slot-major): the slot's block -- m x {p, N, q, 0},
// contiguous -- is fetched in ROW order with coalesced 16-byte loads in the first round trip (it does not wait for
// the tree's indices) and staged in LDS by row (rows4: y | p | N | q, m doubles each); the tree-ordered gather reads
// LDS.  (Gathering the columns from memory cost the texture path 2.7 cycles per lane and request: 4.6 us of a
// 10 us prologue, r04 stamps.)  N stays in rows4 + 2 m for the model step.
template <class F1, class F2>
__device__ __forceinline__ bool chain_rows_select_body(const TreeRowsArgs &a, const SelectArgs &sa, const int t,
                                                       double *lds, double *rows4, ChainSelScratch &cs_,
                                                       const int64_t *__restrict__ pci, const int32_t *__restrict__ pcc,
                                                       long long (&sup)[8], int &ns_out,
                                                       F1 &&first, F2 &&second, double &rmax_out) {
    const int tid = threadIdx.x, j0 = 8 * tid, T = a.T, m = a.m;
    if (a.es != 4 || m > 2048) return false;        // (uniform)
    double *const ylds = rows4, *const plds = rows4 + m, *const nlds = rows4 + 2 * m, *const qlds = rows4 + 3 * m;
    const bool act = j0 < a.tree.n;
    unsigned long long pk[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) pk[i] = 0ull;
    if (act) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            const TreeU2 u = *reinterpret_cast<const TreeU2 *>(a.tree.pack + j0 + i);
            pk[i] = u.v[0]; pk[i + 1] = u.v[1];
        }
    }
    const int pc_raw = pcc[t];
    const int lane = tid & 63;
    const int my_f = (int)pci[(int64_t)t * kAmax + (lane < 8 ? lane : 0)];      // lane k < 8: entry k (beyond the count: row 0)
    double wgt[8];
    tree_fetch_w<256, 8>(a.tree, wgt);
    // the slot's block, 16 bytes per lane and load: chunk c = row c / 2, half c % 2 ({p, N} | {q, 0})
    TreeD2 ch[16];
    {
        const TreeD2 *__restrict__ blk = reinterpret_cast<const TreeD2 *>(a.p + 4 * (int64_t)t * m);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int c = tid + 256 * k;
            ch[k] = blk[c < 2 * m ? c : 0];
        }
    }
    first();
    REVS_KVS(t, 21);
    for (int i = tid; i < m; i += 256) ylds[i] = 0.0;
    if (tid == 0) cs_.vcount = 0;
    REVS_KVS(t, 22);
    const int pc = uni_i(pc_raw);
    REVS_KVS(t, 23);
    if (pc < 0 || pc > 8) return false;
    const double my_y = a.y[(int64_t)my_f * T + t];
    // (a thread's chunks are all the same half: c % 2 == tid % 2)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = tid + 256 * k, row = c >> 1;
        if (c < 2 * m) {
            if (tid & 1) qlds[row] = ch[k].v[0];
            else { plds[row] = ch[k].v[0]; nlds[row] = ch[k].v[1]; }
        }
    }
    REVS_KVS(t, 29);
    double ysup[8];
    const int ns = chain_support_rank(my_f, my_y, lane < 8 && lane < pc && my_y != 0.0, sup, ysup);
    REVS_KVS(t, 30);
    ns_out = ns;
    second();
    REVS_KVS(t, 31);
    __syncthreads();                        // the block by row, and the cleared multipliers' column
    double v8[8], qv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {           // (positions without a row: masked by the scan / zero)
        const int s = (int)(pk[i] & 0xFFFFu) - 1;
        v8[i] = (act && s >= 0) ? plds[s] : 0.0;
        qv[i] = s >= 0 ? qlds[s] : 0.0;
    }
    if (tid < 8 && tid < ns) {
        double mine = 0.0;
        long long row = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) { mine = q == tid ? ysup[q] : mine; row = q == tid ? sup[q] : row; }
        ylds[row] = mine;                   // (read behind the scans' barriers)
    }
    REVS_KVS(t, 1);
    tree_scan<256, 8, true>(a.tree, t, lds, v8, wgt, pk);
    double rmax = 0.0, dsum = 0.0, nsup = 0.0, nvio = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int s = (int)(pk[i] & 0xFFFFu) - 1;
        if (s >= 0) {
            const double v = v8[i], y1 = ylds[s];
            const bool up = y1 > 0.0 || (y1 == 0.0 && v > a.vhi);
            const double b = up ? a.vhi : a.vlo;
            const double vi = fmax(fmax(v - a.vhi, a.vlo - v), 0.0);
            rmax = fmax(rmax, y1 != 0.0 ? fabs(v - b) : vi);
            dsum += qv[i] - fmax(a.vhi * y1, a.vlo * y1);
            nsup += y1 != 0.0 ? 1.0 : 0.0;
            nvio += (y1 == 0.0 && vi > 0.0) ? 1.0 : 0.0;
            if (a.ycopy_out) a.ycopy_out[(int64_t)s * T + t] = y1;
            if (y1 != 0.0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) if (sup[q] == (long long)s) cs_.vsup[q] = v;
            } else if (vi > 0.0) {
                const int q = atomicAdd(&cs_.vcount, 1);
                if (q < kRankMax) { cs_.vval[q] = vi; cs_.vv[q] = v; cs_.vrow[q] = s; }
            }
        }
    }
    REVS_KVS(t, 6);
    rmax = wave_max_d(rmax); dsum = wave_sum_d(dsum); nsup = wave_sum_d(nsup); nvio = wave_sum_d(nvio);
    if ((tid & 63) == 0) { cs_.rr[0][tid >> 6] = rmax; cs_.rr[1][tid >> 6] = dsum; cs_.rr[2][tid >> 6] = nsup; cs_.rr[3][tid >> 6] = nvio; }
    __syncthreads();
    REVS_KVS(t, 7);
    const double o0 = fmax(fmax(cs_.rr[0][0], cs_.rr[0][1]), fmax(cs_.rr[0][2], cs_.rr[0][3]));
    const double o1 = ((cs_.rr[1][0] + cs_.rr[1][1]) + cs_.rr[1][2]) + cs_.rr[1][3];
    const double o2 = ((cs_.rr[2][0] + cs_.rr[2][1]) + cs_.rr[2][2]) + cs_.rr[2][3];
    const double o3 = ((cs_.rr[3][0] + cs_.rr[3][1]) + cs_.rr[3][2]) + cs_.rr[3][3];
    const int nv = (int)o3;
    if (nv > kRankMax || (int)o2 != ns) return false;      // (uniform; the second never happens)
    // what dual_select_body folds from the one block of partials: max(0, .) and 0 + .
    const double rmax_t = fmax(0.0, o0);
    if (tid == 0) {
        double *o = a.partial + (int64_t)t * 4;
        o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
        double *stats = sa.stats;
        stats[t * 8 + 0] = rmax_t;
        stats[t * 8 + 1] = 0.0 + o1;
        stats[t * 8 + 2] = (double)ns;
        stats[t * 8 + 3] = (double)nv;
        if (sa.fwd_src) {                   // (SelectArgs::fwd_src)
#pragma unroll
            for (int i = 0; i < 4; ++i) sa.fwd_dst[t * 8 + i] = sa.fwd_src[t * 8 + i];
        }
        if (!sa.lazy) __threadfence_system();
        if (!sa.lazy) reinterpret_cast<volatile double *>(stats)[t * 8 + 5] = sa.seq;
        else stats[t * 8 + 5] = sa.seq;     // (device memory, nobody polls: a plain store)
    }
    rmax_out = rmax_t;
    int64_t *ci = sa.cidx + (int64_t)t * kAmax;
    double *cs = sa.cval + (int64_t)t * 3 * kAmax, *cg = cs + kAmax, *cy = cg + kAmax;
    SlotLists *const ll = sa.ll;
    const int room = (ns == 0 && nv == 0) ? 0 : min(min(sa.kadd, kAmax - ns), nv);
    const int nq = min(cs_.vcount, kRankMax);
    const int added = min(room, nq);
    REVS_KVS(t, 8);
    REVS_KVV(t, 24, nq); REVS_KVV(t, 25, ns); REVS_KVV(t, 26, room);
    if (room > 0) {
        // the violated rows sit one per thread: `added` rounds of a block-wide arg-max over registers, one barrier
        // each (counting every row's rank instead costs a pass over all of them through LDS)
        double x = tid < nq ? cs_.vval[tid] : 0.0;
        const int r = tid < nq ? cs_.vrow[tid] : 0x7FFFFFFF;
        for (int k2 = 0; k2 < added; ++k2) {
            const double wv = wave_max_d(x);
            const int wr = wave_min_i(x == wv ? r : 0x7FFFFFFF);
            const int pp = k2 & 1;
            if ((tid & 63) == 0) { cs_.best_v[pp][tid >> 6] = wv; cs_.best_r[pp][tid >> 6] = wr; }
            __syncthreads();
            double bv = cs_.best_v[pp][0];
            int br = cs_.best_r[pp][0];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const double ov = cs_.best_v[pp][w];
                const int orow = cs_.best_r[pp][w];
                const bool take = ov > bv || (ov == bv && orow < br);
                bv = take ? ov : bv;
                br = take ? orow : br;
            }
            if (tid < nq && r == br) { x = 0.0; cs_.chosen[k2] = tid; }
        }
    }
    REVS_KVS(t, 9);
    __syncthreads();
    REVS_KVS(t, 10);
    const int cnt = ns + added;
    if (tid < kAmax) {
        long long e_ci = 0;
        double e_cs = 1.0, e_cg = 0.0, e_cy = 0.0;
        if (tid < ns) {
            double yv = 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) if (q == tid) { yv = ysup[q]; e_ci = sup[q]; }
            e_cs = yv > 0.0 ? 1.0 : -1.0;
            e_cg = cs_.vsup[tid] - (yv > 0.0 ? a.vhi : a.vlo);
            e_cy = yv;
        } else if (tid < cnt) {
            const int q = cs_.chosen[tid - ns];
            const double v = cs_.vv[q];
            const bool up = v > a.vhi;
            e_ci = cs_.vrow[q];
            e_cs = up ? 1.0 : -1.0;
            e_cg = v - (up ? a.vhi : a.vlo);
        }
        ci[tid] = e_ci; cs[tid] = e_cs; cg[tid] = e_cg; cy[tid] = e_cy;
        if (ll) { ll->ci[tid] = e_ci; ll->cs[tid] = e_cs; ll->cg[tid] = e_cg; ll->cy[tid] = e_cy; }
    }
    if (tid == 0) { sa.ccnt[t] = cnt; if (ll) ll->cnt = cnt; }
    return true;
}

// Small model, step and the trial's shifts of one slot in one piece -- the folded chain's usual case: at
// most 8 candidates (the list in LDS, left there by the selection: SlotLists) and at most 2048 rows.
// The same arithmetic, term for term, as small_model_body + dual_step_body + chain_shifts_body (the
// stand-alone launches of the general loop give the same bits), without their trips through memory:
// the candidates' rows of R are fetched ONCE -- the Gram sums and the shifts R^T y read the same
// elements, 8 nodes x 8 rows per thread, kept in registers across the pivoting -- the column of N is
// requested by the caller before the rows are even judged (nn[j] = N[tid + 256 j][t]), the wavefront
// sums run over the a (a + 1) / 2 entries that exist, and multipliers, step and list stay in LDS.
template <int A>
__device__ __forceinline__ void chain_fast_body(
        const int t, const int m, const int T, const double *__restrict__ R, const double (&nn)[8],
        double (&r)[8][kSmall], const int npre,
        const SlotLists *ll, const double inv_kappa, const double delta, const int max_pivots, const double al,
        double *__restrict__ Kall, double *__restrict__ yhat, int32_t *__restrict__ info,
        double *__restrict__ ytrial, double *__restrict__ lin_out, double *__restrict__ sh_a,
        double *__restrict__ sh_b) {
    // A = the candidate count: every loop below has compile-time bounds and no branch on it -- straight-line
    // code (a taken branch costs a wavefront ~20 cycles, and these workgroups are nothing but latency)
    const int tid = threadIdx.x;
    REVS_KVS(t, 27);
    long long f[A];
#pragma unroll
    for (int i = 0; i < A; ++i) f[i] = (long long)uni_i((int)ll->ci[i]);
    REVS_KVS(t, 28);
    // (the first npre rows -- the support known before the voltages were -- are here already)
#pragma unroll
    for (int i = 0; i < A; ++i) {
        if (i >= npre) {                    // uniform
#pragma unroll
            // (nodes beyond m read element 0 of the row and are never used: their N is zero in the Gram sums,
            // their shifts are not stored -- no select on the loaded value, which would make every row wait
            // for its own loads before the next row's are issued)
            for (int j = 0; j < 8; ++j) {
                const int mm = tid + 256 * j;
                r[j][i] = R[f[i] * m + (mm < m ? mm : 0)];
            }
        }
    }
    REVS_KVS(t, 12);
    constexpr int kTri = kSmall * (kSmall + 1) / 2;
    auto tri = [](int i, int jj) { return i * kSmall - i * (i - 1) / 2 + (jj - i); };   // (i <= jj)
    double acc[kTri];
#pragma unroll
    for (int p = 0; p < kTri; ++p) acc[p] = 0.0;
    // (nodes beyond m carry N = 0 and a finite r: their terms add exactly nothing)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int i = 0; i < A; ++i) {
            const double rn = r[j][i] * nn[j];
#pragma unroll
            for (int jj = i; jj < A; ++jj) acc[tri(i, jj)] += rn * r[j][jj];
        }
    }
    __shared__ double part[4][kTri];
    __shared__ double Ks[kSmall][kSmall];
    __shared__ double yo_s[kSmall];
    REVS_KVS(t, 13);
    {
        // the A (A + 1) / 2 sums level by level (their chains interleave), then ONE store: lane e keeps entry e
        constexpr int kN = A * (A + 1) / 2;
        double red[kN];
        int e = 0;
#pragma unroll
        for (int i = 0; i < A; ++i)
#pragma unroll
            for (int jj = i; jj < A; ++jj) red[e++] = acc[tri(i, jj)];
        wave_sum_multi_d<kN>(red);
        double mine = 0.0;
        int slot = 0;
        e = 0;
#pragma unroll
        for (int i = 0; i < A; ++i)
#pragma unroll
            for (int jj = i; jj < A; ++jj, ++e) {
                const bool me = (tid & 63) == e;
                mine = me ? red[e] : mine;
                slot = me ? tri(i, jj) : slot;
            }
        if ((tid & 63) < kN) part[tid >> 6][slot] = mine;
    }
    __syncthreads();
    REVS_KVS(t, 14);
    if (tid < kSmall * kSmall) {
        const int i = tid / kSmall, j = tid % kSmall;
        if (i < A && j < A) {
            const int lo = i < j ? i : j, hi = i < j ? j : i;
            const int p = lo * kSmall - lo * (lo - 1) / 2 + (hi - lo);
            const double v = (((part[0][p] + part[1][p]) + part[2][p]) + part[3][p]) * inv_kappa;
            Ks[i][j] = v;
            Kall[(int64_t)t * kAmax * kAmax + i * kAmax + j] = v;
        }
    }
    __syncthreads();
    REVS_KVS(t, 15);
    if (tid == 0) {
        int32_t inf;
        small_bpp<A>(Ks, ll->cs, ll->cg, ll->cy, delta, max_pivots, yo_s, &inf, kSmall);
        info[t] = inf;
        REVS_KVS(t, 16);
    }
    __syncthreads();
    if (tid < kAmax) yhat[(int64_t)t * kAmax + tid] = tid < kSmall ? yo_s[tid] : 0.0;
    // the step (dual_step_body), every thread for itself: y_trial at the candidates, in list order
    double yn[A];
#pragma unroll
    for (int i = 0; i < A; ++i) {
        const double yo = ll->cy[i], yh = yo_s[i];
        yn[i] = uni_d(al == 1.0 ? yh : (al == 0.0 ? yo : yo + al * (yh - yo)));
    }
    if (tid < 64) {
        double lin = 0.0;
        if (tid < A) {
            double mine = 0.0;
#pragma unroll
            for (int i = 0; i < A; ++i) mine = i == tid ? yn[i] : mine;
            ytrial[ll->ci[tid] * T + t] = mine;
            lin += ll->cg[tid] * (mine - ll->cy[tid]);
        }
        lin = wave_sum_d(lin);
        if (tid == 0) lin_out[t * 8] = lin;
    }
    REVS_KVS(t, 18);
    // the shifts (chain_shifts_body): list order -- a term with y = 0 adds exactly nothing, so the rows
    // need no compaction -- and, where it differs, ascending row order of the rows that carry a multiplier
    bool same = true;
    {   // list order restricted to the rows with a multiplier ascending already (the usual case)?
        long long last = -1;
#pragma unroll
        for (int k2 = 0; k2 < A; ++k2) {
            const bool nz = yn[k2] != 0.0;
            same = same && (!nz || f[k2] > last);
            last = nz ? f[k2] : last;
        }
    }
    REVS_KVS(t, 19);
    if (same) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int node = tid + 256 * j;
            double d = 0.0;
#pragma unroll
            for (int k2 = 0; k2 < A; ++k2) d = __builtin_fma(r[j][k2], yn[k2], d);
            if (node < m) {
                sh_a[(int64_t)t * m + node] = d * inv_kappa;
                sh_b[(int64_t)t * m + node] = d * inv_kappa;
            }
        }
        return;
    }
    int ord[A], ns = 0;
    {
        long long last = -1;
#pragma unroll
        for (int q = 0; q < A; ++q) {
            long long best = 0x7FFFFFFFFFFFFFFFll;
            int bk = -1;
#pragma unroll
            for (int k2 = 0; k2 < A; ++k2)
                if (yn[k2] != 0.0 && f[k2] > last && f[k2] < best) { best = f[k2]; bk = k2; }
            ord[q] = bk;
            if (bk >= 0) { ++ns; last = best; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int node = tid + 256 * j;
        if (node >= m) continue;
        double d = 0.0, ds = 0.0;
#pragma unroll
        for (int k2 = 0; k2 < A; ++k2) d = __builtin_fma(r[j][k2], yn[k2], d);
#pragma unroll
        for (int q = 0; q < A; ++q) {
            if (q < ns) {
                double rv = 0.0, yq = 0.0;
#pragma unroll
                for (int k2 = 0; k2 < A; ++k2)
                    if (k2 == ord[q]) { rv = r[j][k2]; yq = yn[k2]; }
                ds = __builtin_fma(rv, yq, ds);
            }
        }
        sh_a[(int64_t)t * m + node] = d * inv_kappa;
        sh_b[(int64_t)t * m + node] = ds * inv_kappa;
    }
}

struct ChainKvArgs {
    int has_e2;
    TreeRowsArgs e2;
    SelectArgs s2;
    TreeRowsArgs e1;
    SelectArgs s1;
    FusedArgs f1;
    double *clr0, *clr1;
    long long clr_count;
    double *sh_a, *sh_b;
    const int64_t *prev_ci;
    const int32_t *prev_cc;
    double *stamps;             // tuning build only
};
__global__ __launch_bounds__(256) void op_chain_kv_kernel(const ChainKvArgs k) {
    extern __shared__ double tree_lds[];
    const int T = k.s1.T, m = k.s1.m;
    REVS_KVS_BEGIN((k.has_e2 && (int)blockIdx.x < T) ? nullptr : k.stamps);
    // (the rows of a slot go to its selection through LDS: double[4 m + 4] behind the tree's scan buffer)
    double *rows_lds = tree_lds + (tree_lds_bytes(k.e1.tree.n) / sizeof(double) + 1) / 2 * 2;
    __shared__ ChainSelScratch sel_s;
    long long sup[8];
    if (k.has_e2 && (int)blockIdx.x < T) {
        const int t2 = blockIdx.x;
        double rm2;
        int ns2;
        bool done = false;
        if (k.prev_cc)
            done = chain_rows_select_body(k.e2, k.s2, t2, tree_lds, rows_lds, sel_s, k.prev_ci, k.prev_cc, sup, ns2,
                                          NoPrefetch(), NoPrefetch(), rm2);
        if (!done) {
            __syncthreads();
            tree_rows_body(k.e2, t2, tree_lds, rows_lds);
            __syncthreads();
            SelectArgs s2 = k.s2;
            s2.rows_lds = rows_lds;
            dual_select_body<true>(s2, t2);
        }
        // the verdict is out: these workgroups have time left -- they clear the two sum arrays the NEXT
        // sweep accumulates into (behind everything the other half of the launch has to fetch)
        const long long per = (k.clr_count + T - 1) / T;
        const long long i0 = blockIdx.x * per, i1 = i0 + per < k.clr_count ? i0 + per : k.clr_count;
        for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
            if (k.clr0) k.clr0[i] = 0.0;
            if (k.clr1) k.clr1[i] = 0.0;
        }
        return;
    }
    const int t = (int)blockIdx.x - (k.has_e2 ? T : 0);
    // (tuning build: stamps 0 | 1-6 rows | 7-10 selection | 11-16 model | 17 step | 18-20 shifts)
#define KV_STAMP(i) REVS_KVS(t, i)
    KV_STAMP(0);
    __shared__ SlotLists lists;
    double nn[8], r[8][kSmall];
    int npre = 0;
    // this slot's column of N (the model's weights) and the rows of R of the multipliers' support: needed
    // after the selection, requested in front of it
    auto fetch_nn = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int mm = threadIdx.x + 256 * j;
            const double v = k.f1.nes == 4 ? k.f1.Nn[4 * ((int64_t)t * m + (mm < m ? mm : 0))] : k.f1.Nn[(int64_t)(mm < m ? mm : 0) * T + t];
            nn[j] = mm < m ? v : 0.0;
        }
    };
    double rmax = 0.0;
    bool done = false;
    // (the folded sweep's sums: N arrives with p and q in the slot's block and reaches the model through LDS, by row)
    if (k.prev_cc) {
        SelectArgs s1 = k.s1;
        s1.ll = &lists;
        int ns1 = 0;
        done = chain_rows_select_body(k.e1, s1, t, tree_lds, rows_lds, sel_s, k.prev_ci, k.prev_cc, sup, ns1, NoPrefetch(),
                                      [&]() {
            if (m <= 2048) {
#pragma unroll
                for (int q = 0; q < kSmall; ++q) {
                    if (q < ns1) {          // uniform
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int mm = threadIdx.x + 256 * j;
                            r[j][q] = k.f1.R[sup[q] * m + (mm < m ? mm : 0)];
                        }
                    }
                }
            }
        }, rmax);
        if (done && m <= 2048) npre = ns1;
        if (done) {                         // (staged before the selection's barriers)
            const double *const nlds = rows_lds + 2 * m;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int mm = threadIdx.x + 256 * j;
                nn[j] = mm < m ? nlds[mm] : 0.0;
            }
        }
    }
    if (!done) {
        __syncthreads();
        tree_rows_body(k.e1, t, tree_lds, rows_lds, fetch_nn);
        __syncthreads();
        KV_STAMP(7);
        SelectArgs s1 = k.s1;
        s1.rows_lds = rows_lds;
        s1.ll = &lists;
        rmax = dual_select_body<true>(s1, t);
    }
    __syncthreads();
    KV_STAMP(11);
    const double al = rmax / k.f1.scale > k.f1.eps ? 1.0 : 0.0;
    const int cnt1 = uni_i(lists.cnt);
    if (cnt1 >= 1 && cnt1 <= kSmall && m <= 2048) {
#define REVS_FAST(A) chain_fast_body<A>(t, m, T, k.f1.R, nn, r, npre, &lists, k.f1.inv_kappa, k.f1.delta, k.f1.max_pivots, al, \
                                        k.f1.Kall, k.f1.yhat, k.f1.info, k.f1.ytrial, k.f1.lin_out, k.sh_a, k.sh_b)
        switch (cnt1) {
            case 1: REVS_FAST(1); break;
            case 2: REVS_FAST(2); break;
            case 3: REVS_FAST(3); break;
            case 4: REVS_FAST(4); break;
            case 5: REVS_FAST(5); break;
            case 6: REVS_FAST(6); break;
            case 7: REVS_FAST(7); break;
            default: REVS_FAST(8); break;
        }
#undef REVS_FAST
        KV_STAMP(20);
        return;
    }
    if (cnt1 == 0) {
        // a slot without multipliers or violated rows: what the three bodies below would do with an empty
        // list -- yhat = the (zero) multipliers, no step, lin = 0, shifts zero -- without their loops
        if (threadIdx.x < kAmax) k.f1.yhat[(int64_t)t * kAmax + threadIdx.x] = 0.0;
        if (threadIdx.x == 0) { k.f1.info[t] = 0; k.f1.lin_out[t * 8] = 0.0; }
        for (int node = threadIdx.x; node < m; node += 256) {
            k.sh_a[(int64_t)t * m + node] = 0.0;
            k.sh_b[(int64_t)t * m + node] = 0.0;
        }
        KV_STAMP(20);
        return;
    }
    small_model_body(t, m, T, k.f1.R, k.f1.Nn, k.s1.cidx, k.s1.ccnt, k.s1.cval, k.f1.inv_kappa, k.f1.delta,
                     k.f1.max_pivots, k.f1.Kall, k.f1.yhat, k.f1.info, k.f1.nes);
    __syncthreads();
    KV_STAMP(17);
    dual_step_body<256>(t, T, k.s1.cidx, k.s1.ccnt, k.s1.cval, k.f1.yhat, al, k.f1.ytrial, k.f1.lin_out, nullptr, m);
    __syncthreads();                        // the trial's column (first wavefront), in global memory
    KV_STAMP(18);
    chain_shifts_body(t, m, T, k.f1.R, k.s1.cidx, k.s1.ccnt, k.f1.ytrial, k.f1.inv_kappa, k.sh_a, k.sh_b);
    __syncthreads();
    KV_STAMP(20);
#undef KV_STAMP
}

int chain_kv_launch(const ChainKv &c, void *stream) {
    REVS_REQUIRE(c.m > 0 && c.m <= 16384 && c.T > 0 && c.T <= 256 && c.tree.n > 0 && c.tree.n <= REVS_TREE_SWEEP_MAX &&
                 c.e1.pnq && c.e1.y && c.e1.vfull && c.e1.viol && c.e1.partial && c.e1.cidx && c.e1.ccnt && c.e1.cval &&
                 c.e1.stats && c.R && c.k_full && c.yhat && c.info && c.y_trial && c.lin_out && c.y_trial != c.e1.y &&
                 c.sh_a && c.sh_b && c.sh_a != c.sh_b &&
                 (!c.has_e2 || (c.e2.pnq && c.e2.y && c.e2.vfull && c.e2.viol && c.e2.partial && c.e2.cidx &&
                                c.e2.ccnt && c.e2.cval && c.e2.stats && c.e2.vfull != c.e1.vfull)),
                 "chain_kv_launch: bad argument");
    const int64_t mt = (int64_t)c.m * c.T;
    ChainKvArgs k;
    k.has_e2 = c.has_e2;
    REVS_REQUIRE((c.e1.es == 1 || c.e1.es == 4) && (!c.has_e2 || c.e2.es == 1 || c.e2.es == 4), "chain_kv_launch: bad layout of the sums");
    auto rows = [&](const ChainKvSide &s) {
        TreeRowsArgs r{c.tree, c.m, c.T, s.pnq, s.pnq + (s.es == 4 ? 2 : 2 * mt), s.y, c.vlo, c.vhi, s.vfull, s.viol, s.partial, nullptr};
        r.es = s.es;
        return r;
    };
    auto sel = [&](const ChainKvSide &s) {
        return SelectArgs{c.m, c.T, 1, c.kadd, s.partial, s.y, s.vfull, s.viol, c.vlo, c.vhi, s.seq, s.cidx, s.ccnt, s.cval, s.stats};
    };
    k.e1 = rows(c.e1);
    k.e1.ycopy_out = c.y_trial;     // the next trial starts from this evaluation's multipliers: copied by the rows' pass
    k.s1 = sel(c.e1);
    k.s1.lazy = c.has_e2 != 0;      // (its stats stay on the device: c.e1.stats; the next launch's verdict half forwards them)
    REVS_REQUIRE((c.fwd_src == nullptr) == (c.fwd_dst == nullptr) && (!c.fwd_src || c.has_e2), "chain_kv_launch: bad stats forwarding");
    k.e2 = rows(c.has_e2 ? c.e2 : c.e1);
    k.s2 = sel(c.has_e2 ? c.e2 : c.e1);
    k.s2.fwd_src = c.fwd_src; k.s2.fwd_dst = c.fwd_dst;
    k.f1 = FusedArgs{c.R, c.e1.pnq + (c.e1.es == 4 ? 1 : mt), c.e1.y, 1.0 / c.kappa, c.delta, c.scale, c.eps, c.max_pivots,
                     c.k_full, c.yhat, c.y_trial, c.lin_out, c.info, c.e1.es};
    k.clr0 = c.clr0; k.clr1 = c.clr1; k.clr_count = 4 * mt;      // (the folded sweep's arrays: {p, N, q, 0} per node and slot)
    k.sh_a = c.sh_a; k.sh_b = c.sh_b;
    k.prev_ci = c.prev_cidx; k.prev_cc = c.prev_cidx ? c.prev_ccnt : nullptr;
    const size_t lds = ((tree_lds_bytes(c.tree.n) / sizeof(double) + 1) / 2 * 2 + 4 * (size_t)c.m + 4) * sizeof(double);
    // (a slot's rows are staged in LDS: 4 m doubles beside the tree's scan buffer and ~30 KB of static LDS)
    REVS_REQUIRE(c.m <= REVS_CHAIN_FOLD_MAX_M && lds <= 128 * 1024,
                 "chain_kv_launch: m = %d rows do not fit the operator launch's LDS (at most %d)", c.m, REVS_CHAIN_FOLD_MAX_M);
    if (!grant_lds(reinterpret_cast<const void *>(&op_chain_kv_kernel), lds > 64 * 1024 ? 128 * 1024 : lds, "chain_kv_launch"))
        return REVS_ELAUNCH;
    k.stamps = c.sh_b + mt;
    hipLaunchKernelGGL(op_chain_kv_kernel, dim3((c.has_e2 ? 2 : 1) * c.T), dim3(256), lds, (hipStream_t)stream, k);
    REVS_CHECK_LAUNCH("chain_kv_launch");
    return REVS_OK;
}

}  // namespace revs

using namespace revs;
#define S_(stream) ((hipStream_t)(stream))

static int dual_eval_impl(int32_t m, int32_t T, const int64_t *node_ptr, const float *p_est,
                          const float *p_sch, const float *gamma, int32_t nslab, const double *dsl,
                          double kappa, double *pnq, float *p_est_new, const SparseD &sp,
                          void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && T <= 256 && node_ptr && p_est && p_sch && gamma && pnq &&
                 kappa > 0, "revs_op_dual_eval: bad argument");
    REVS_REQUIRE(!dsl || nslab >= 1, "revs_op_dual_eval: nslab=%d", nslab);
#define EV(TL)                                                                                 \
    hipLaunchKernelGGL((op_dual_eval_kernel<TL>), dim3(m), dim3(256), 0, S_(stream), m, T,     \
                       node_ptr, p_est, p_sch, gamma, nslab, dsl, kappa, pnq, p_est_new, sp)
    if (T <= 32) EV(32);
    else if (T <= 64) EV(64);
    else if (T <= 128) EV(128);
    else EV(256);
#undef EV
    REVS_CHECK_LAUNCH("revs_op_dual_eval");
    return REVS_OK;
}

extern "C" int revs_op_dual_eval(int32_t m, int32_t T, const int64_t *node_ptr,
                                 const float *p_est, const float *p_sch, const float *gamma,
                                 int32_t nslab, const double *dsl, double kappa, double *pnq,
                                 float *p_est_new, void *stream) {
    return dual_eval_impl(m, T, node_ptr, p_est, p_sch, gamma, nslab, dsl, kappa, pnq, p_est_new,
                          SparseD{nullptr, nullptr, nullptr, nullptr}, stream);
}

extern "C" int revs_op_dual_eval_rows(int32_t m, int32_t T, const int64_t *node_ptr,
                                      const float *p_est, const float *p_sch, const float *gamma,
                                      const double *R, const int64_t *sup_idx,
                                      const int32_t *sup_cnt, const double *y, double kappa,
                                      double *pnq, float *p_est_new, void *stream) {
    REVS_REQUIRE(R && sup_idx && sup_cnt && y, "revs_op_dual_eval_rows: null argument");
    return dual_eval_impl(m, T, node_ptr, p_est, p_sch, gamma, 0, nullptr, kappa, pnq, p_est_new,
                          SparseD{R, sup_idx, sup_cnt, y}, stream);
}

extern "C" int32_t revs_op_dual_blocks(int32_t m) {
    return m <= 0 ? 0 : (m + 7) / 8 < 256 ? (m + 7) / 8 : 256;
}

static int dual_rows_select(int32_t m, int32_t T, int32_t nslab, const double *vsl,
                            const double *pnq, const double *y, double vlo, double vhi,
                            int32_t kadd, double *vfull, double *viol, double *partial,
                            int64_t *cand_idx, int32_t *cand_cnt, double *cand_val, double *stats,
                            double seq, bool defer_select, double *zero_out, void *stream) {
    REVS_REQUIRE(m <= 16384, "revs_op_dual_select: m=%d exceeds 16384 rows", m);
    REVS_REQUIRE(m > 0 && T > 0 && T <= 256 && nslab >= 1 && vsl && pnq && y && vfull && viol &&
                 partial && cand_idx && cand_cnt && cand_val && stats && vlo <= vhi && kadd >= 0,
                 "revs_op_dual_select: bad argument");
    const int nblk = revs_op_dual_blocks(m);
#define RW(TL)                                                                                 \
    hipLaunchKernelGGL((op_dual_rows_kernel<TL>), dim3(nblk), dim3(256), 0, S_(stream), m, T,  \
                       nslab, vsl, pnq, y, vlo, vhi, vfull, viol, partial, zero_out)
    if (T <= 32) RW(32);
    else if (T <= 64) RW(64);
    else if (T <= 128) RW(128);
    else RW(256);
#undef RW
    if (!defer_select) {
        const SelectArgs sa{m, T, nblk, kadd, partial, y, vfull, viol, vlo, vhi, seq,
                            cand_idx, cand_cnt, cand_val, stats};
        hipLaunchKernelGGL(op_dual_select_kernel, dim3(T), dim3(256), 0, S_(stream), sa);
    }
    REVS_CHECK_LAUNCH("revs_op_dual_select");
    return REVS_OK;
}

extern "C" int revs_op_dual_select(int32_t m, int32_t T, int32_t nslab, const double *vsl,
                                   const double *pnq, const double *y, double vlo, double vhi,
                                   int32_t kadd, double *vfull, double *viol, double *partial,
                                   int64_t *cand_idx, int32_t *cand_cnt, double *cand_val,
                                   double *stats, double seq, void *stream) {
    return dual_rows_select(m, T, nslab, vsl, pnq, y, vlo, vhi, kadd, vfull, viol, partial, cand_idx,
                            cand_cnt, cand_val, stats, seq, false, nullptr, stream);
}

extern "C" int revs_op_dual_rows(int32_t m, int32_t T, int32_t nslab, const double *vsl,
                                 const double *pnq, const double *y, double vlo, double vhi,
                                 double *vfull, double *viol, double *partial, double *zero_out,
                                 void *stream) {
    int64_t ci = 0; int32_t cc = 0; double cv = 0.0, st = 0.0;   // unused: selection deferred
    return dual_rows_select(m, T, nslab, vsl, pnq, y, vlo, vhi, 0, vfull, viol, partial, &ci, &cc,
                            &cv, &st, 0.0, true, zero_out, stream);
}

extern "C" int revs_op_dual_model(int32_t m, int32_t T, const double *R, const double *n_free,
                                  const int64_t *cand_idx, const int32_t *cand_cnt,
                                  const double *cand_val, double kappa, double delta,
                                  int32_t max_pivots, int32_t nks, double *k_slabs, double *k_full,
                                  double *yhat, int32_t *info, void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && R && n_free && cand_idx && cand_cnt && cand_val && k_slabs &&
                 k_full && yhat && info && kappa > 0 && delta >= 0 && max_pivots > 0 &&
                 nks >= 1 && nks <= 64, "revs_op_dual_model: bad argument");
    static const size_t lds = sizeof(double) * kAmax * (kAmax + 1);
    if (!grant_lds(reinterpret_cast<const void *>(&op_dual_bpp_kernel), lds, "revs_op_dual_model")) return REVS_ELAUNCH;
    hipLaunchKernelGGL(op_dual_gram_kernel, dim3(T, nks, kWords * kWords), dim3(256), 0, S_(stream),
                       m, T, R, n_free, cand_idx, cand_cnt, nks, k_slabs);
    hipLaunchKernelGGL(op_dual_bpp_kernel, dim3(T), dim3(256), lds, S_(stream), k_slabs, nks,
                       1.0 / kappa, k_full, cand_cnt, cand_val, delta, max_pivots, yhat, info);
    REVS_CHECK_LAUNCH("revs_op_dual_model");
    return REVS_OK;
}

extern "C" int revs_op_dual_model_small(int32_t m, int32_t T, const double *R, const double *n_free,
                                        const int64_t *cand_idx, const int32_t *cand_cnt,
                                        const double *cand_val, double kappa, double delta,
                                        int32_t max_pivots, double *k_full, double *yhat,
                                        int32_t *info, void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && R && n_free && cand_idx && cand_cnt && cand_val && k_full && yhat &&
                 info && kappa > 0 && delta >= 0 && max_pivots > 0,
                 "revs_op_dual_model_small: bad argument");
    hipLaunchKernelGGL(op_dual_model_small_kernel, dim3(T), dim3(256), 0, S_(stream), m, T, R, n_free,
                       cand_idx, cand_cnt, cand_val, 1.0 / kappa, delta, max_pivots, k_full, yhat, info);
    REVS_CHECK_LAUNCH("revs_op_dual_model_small");
    return REVS_OK;
}

extern "C" int revs_op_dual_select_model_step(
        int32_t m, int32_t T, const double *sel_partial, int32_t sel_nblk, const double *y, double vlo,
        double vhi, int32_t kadd, const double *vfull, const double *viol, int64_t *cand_idx,
        int32_t *cand_cnt, double *cand_val, double *stats, double seq, const double *R,
        const double *n_free, double kappa, double delta, int32_t max_pivots, double *k_full,
        double *yhat, int32_t *info, double scale, double eps, double *y_trial, double *lin_out,
        void *stream) {
    REVS_REQUIRE(m > 0 && m <= 16384 && T > 0 && sel_partial && y && vfull && viol && cand_idx &&
                 cand_cnt && cand_val && stats && vlo <= vhi && kadd >= 0 && sel_nblk >= 0 &&
                 sel_nblk <= 256 && R && n_free && k_full && yhat && info && kappa > 0 && delta >= 0 &&
                 max_pivots > 0 && scale > 0.0 && y_trial && y_trial != y && lin_out,
                 "revs_op_dual_select_model_step: bad argument");
    const SelectArgs sa{m, T, sel_nblk ? sel_nblk : revs_op_dual_blocks(m), kadd, sel_partial, y, vfull,
                        viol, vlo, vhi, seq, cand_idx, cand_cnt, cand_val, stats};
    const FusedArgs fa{R, n_free, y, 1.0 / kappa, delta, scale, eps, max_pivots, k_full, yhat, y_trial,
                       lin_out, info};
    hipLaunchKernelGGL(op_dual_select_model_step_kernel, dim3(T), dim3(256), 0, S_(stream), sa, fa);
    REVS_CHECK_LAUNCH("revs_op_dual_select_model_step");
    return REVS_OK;
}

extern "C" int revs_op_dual_step(int32_t T, const int64_t *cand_idx, const int32_t *cand_cnt,
                                 const double *cand_val, const double *yhat, const double *alpha,
                                 double *y_trial, double *lin_out, void *stream) {
    REVS_REQUIRE(T > 0 && cand_idx && cand_cnt && cand_val && yhat && alpha && y_trial && lin_out,
                 "revs_op_dual_step: bad argument");
    hipLaunchKernelGGL(op_dual_step_kernel, dim3(T), dim3(64), 0, S_(stream), T, cand_idx,
                       cand_cnt, cand_val, yhat, alpha, y_trial, lin_out, nullptr, 1.0, 0.0, nullptr, 0);
    REVS_CHECK_LAUNCH("revs_op_dual_step");
    return REVS_OK;
}

namespace revs {
int dual_step_copy(int32_t T, const int64_t *cand_idx, const int32_t *cand_cnt, const double *cand_val, const double *yhat,
                   const double *alpha, const double *y, int32_t m, double *y_trial, double *lin_out, void *stream) {
    REVS_REQUIRE(T > 0 && cand_idx && cand_cnt && cand_val && yhat && alpha && y && m > 0 && y_trial && y_trial != y && lin_out,
                 "dual_step_copy: bad argument");
    hipLaunchKernelGGL(op_dual_step_copy_kernel, dim3(T), dim3(256), 0, S_(stream), T, cand_idx, cand_cnt, cand_val, yhat,
                       alpha, y_trial, lin_out, y, m);
    REVS_CHECK_LAUNCH("dual_step_copy");
    return REVS_OK;
}
}  // namespace revs

extern "C" int revs_op_dual_step_pending(int32_t T, const int64_t *cand_idx, const int32_t *cand_cnt,
                                         const double *cand_val, const double *yhat,
                                         const double *stats_prev, double scale, double eps,
                                         const double *y, int32_t m, double *y_trial,
                                         double *lin_out, void *stream) {
    REVS_REQUIRE(T > 0 && cand_idx && cand_cnt && cand_val && yhat && stats_prev && y_trial &&
                 lin_out && scale > 0.0 && (!y || m > 0) && y != y_trial,
                 "revs_op_dual_step_pending: bad argument");
    hipLaunchKernelGGL(op_dual_step_kernel, dim3(T), dim3(64), 0, S_(stream), T, cand_idx,
                       cand_cnt, cand_val, yhat, nullptr, y_trial, lin_out, stats_prev, scale, eps,
                       y, m);
    REVS_CHECK_LAUNCH("revs_op_dual_step_pending");
    return REVS_OK;
}

static bool tree_ok(const revs_tree_t *tree) {       // the fused launches: one workgroup of 256 x 8 positions per slot
    return tree && tree->n > 0 && tree->n <= REVS_TREE_SWEEP_MAX && tree->n % 8 == 0 && tree->pack && tree->w;
}
static bool tree_ok_big(const revs_tree_t *tree) {   // the evaluations' row launches: every shape of tree_body.h
    return tree && tree->n > 0 && tree->n <= REVS_TREE_MAX && tree->n % tree_shape(tree->n).ipt == 0 && tree->pack && tree->w;
}
template <int NT, int IPT, typename K>
static bool rows_big_lds(K kernel, size_t lds) {     // more than 64 KB of dynamic LDS is granted per kernel and device
    return grant_lds(reinterpret_cast<const void *>(kernel), lds, "revs_op_dual_rows_tree");
}

static int dual_shift_tree(int32_t m, int32_t T, const revs_tree_t *tree, const double *y, double *d_out, void *stream) {
    REVS_REQUIRE(m > 0 && m <= 16384 && T > 0 && T <= 256 && tree_ok_big(tree) && y && d_out, "revs_op_dual_evaluate_tree: bad argument");
    const TreeArgs tr{tree->n, (const unsigned long long *)tree->pack, tree->w};
    const size_t lds = tree_lds_bytes(tree->n);
    const TreeShape sh = tree_shape(tree->n);
#define SK(NT, IPT)                                                                                                       \
    do {                                                                                                                  \
        if (!grant_lds(reinterpret_cast<const void *>(&op_tree_shift_kernel<NT, IPT>), lds, "revs_op_dual_evaluate_tree")) \
            return REVS_ELAUNCH;                                                                                          \
        hipLaunchKernelGGL((op_tree_shift_kernel<NT, IPT>), dim3(T), dim3(NT), lds, S_(stream), tr, T, y, d_out);         \
    } while (0)
    if (sh.nt == 256) SK(256, 8);
    else if (sh.nt == 512) SK(512, 8);
    else if (sh.ipt == 8) SK(1024, 8);
    else SK(1024, 16);
#undef SK
    REVS_CHECK_LAUNCH("revs_op_dual_evaluate_tree");
    return REVS_OK;
}

extern "C" int revs_op_dual_rows_tree(int32_t m, int32_t T, const revs_tree_t *tree, const double *pnq,
                                      const double *y, double vlo, double vhi, int32_t kadd, double *vfull,
                                      double *viol, double *partial, double *zero_out, int64_t *cand_idx,
                                      int32_t *cand_cnt, double *cand_val, double *stats, double seq,
                                      int32_t with_select, void *stream) {
    REVS_REQUIRE(m > 0 && m <= 16384 && T > 0 && T <= 256 && tree_ok_big(tree) && pnq && y && vfull && viol &&
                 partial && vlo <= vhi && kadd >= 0 && zero_out != pnq &&
                 (!with_select || (cand_idx && cand_cnt && cand_val && stats)),
                 "revs_op_dual_rows_tree: bad argument (tree nodes <= %d, a multiple of 8; of 16 beyond 8192)", REVS_TREE_MAX);
    const TreeArgs tr{tree->n, (const unsigned long long *)tree->pack, tree->w};
    const TreeRowsArgs ta{tr, m, T, pnq, pnq + 2 * (int64_t)m * T, y, vlo, vhi, vfull, viol, partial, zero_out};
    const SelectArgs sa{m, T, 1, kadd, partial, y, vfull, viol, vlo, vhi, seq, cand_idx, cand_cnt, cand_val, stats};
    if (tree->n > REVS_TREE_SWEEP_MAX) {
        // more than 2048 nodes: the rows by a workgroup of 512 or 1024 threads per slot, the selection behind it
        const size_t lds = tree_lds_bytes(tree->n);
        const TreeShape sh = tree_shape(tree->n);
#define RK(NT, IPT)                                                                                           \
        do {                                                                                                  \
            if (!rows_big_lds<NT, IPT>(op_tree_rows_big_kernel<NT, IPT>, lds)) return REVS_ELAUNCH;           \
            hipLaunchKernelGGL((op_tree_rows_big_kernel<NT, IPT>), dim3(T), dim3(NT), lds, S_(stream), ta);   \
        } while (0)
        if (sh.nt == 512) RK(512, 8);
        else if (sh.ipt == 8) RK(1024, 8);
        else RK(1024, 16);
#undef RK
        REVS_CHECK_LAUNCH("revs_op_dual_rows_tree");
        if (with_select) {
            hipLaunchKernelGGL(op_dual_select_kernel, dim3(T), dim3(256), 0, S_(stream), sa);
            REVS_CHECK_LAUNCH("revs_op_dual_rows_tree (selection)");
        }
        return REVS_OK;
    }
    if (with_select) {
        // (the rows go to the selection through LDS where they fit: vfull / viol are scratch of this call then)
        const size_t staged_lds = ((tree_lds_bytes(tree->n) / sizeof(double) + 1) / 2 * 2 + 3 * (size_t)m + 4) * sizeof(double);
        const bool staged = staged_lds <= 120 * 1024;
        if (staged && !grant_lds(reinterpret_cast<const void *>(&op_tree_rows_kernel<true>), staged_lds > 64 * 1024 ? 120 * 1024 : staged_lds,
                                 "revs_op_dual_rows_tree"))
            return REVS_ELAUNCH;
        hipLaunchKernelGGL((op_tree_rows_kernel<true>), dim3(T), dim3(256), staged ? staged_lds : tree_lds_bytes(tree->n), S_(stream),
                           ta, sa, staged ? 1 : 0);
    } else {
        hipLaunchKernelGGL((op_tree_rows_kernel<false>), dim3(T), dim3(256), tree_lds_bytes(tree->n), S_(stream), ta, sa, 0);
    }
    REVS_CHECK_LAUNCH("revs_op_dual_rows_tree");
    return REVS_OK;
}

extern "C" int revs_op_dual_tree_select_model_step(
        int32_t m, int32_t T, const revs_tree_t *tree, const double *pnq, const double *y, double vlo,
        double vhi, int32_t kadd, double *vfull, double *viol, double *partial, int64_t *cand_idx,
        int32_t *cand_cnt, double *cand_val, double *stats, double seq, const double *R, double kappa,
        double delta, int32_t max_pivots, double *k_full, double *yhat, int32_t *info, double scale,
        double eps, double *y_trial, double *lin_out, void *stream) {
    REVS_REQUIRE(m > 0 && m <= 16384 && T > 0 && T <= 256 && tree_ok(tree) && pnq && y && vfull && viol &&
                 partial && cand_idx && cand_cnt && cand_val && stats && vlo <= vhi && kadd >= 0 && R && k_full &&
                 yhat && info && kappa > 0 && delta >= 0 && max_pivots > 0 && scale > 0.0 && y_trial &&
                 y_trial != y && lin_out, "revs_op_dual_tree_select_model_step: bad argument");
    const TreeArgs tr{tree->n, (const unsigned long long *)tree->pack, tree->w};
    const TreeRowsArgs ta{tr, m, T, pnq, pnq + 2 * (int64_t)m * T, y, vlo, vhi, vfull, viol, partial, nullptr};
    const SelectArgs sa{m, T, 1, kadd, partial, y, vfull, viol, vlo, vhi, seq, cand_idx, cand_cnt, cand_val, stats};
    const FusedArgs fa{R, pnq + (int64_t)m * T, y, 1.0 / kappa, delta, scale, eps, max_pivots, k_full, yhat,
                       y_trial, lin_out, info};
    hipLaunchKernelGGL(op_tree_select_model_step_kernel, dim3(T), dim3(256), tree_lds_bytes(tree->n), S_(stream),
                       ta, sa, fa);
    REVS_CHECK_LAUNCH("revs_op_dual_tree_select_model_step");
    return REVS_OK;
}

// One evaluation of the dual function as a single host call (the driver's steady state is
// host-bound otherwise: five launches of 3-12 us each).  phase bit 0: R^T y (when use_y)
// and the home pass; phase bit 1: R p, row bookkeeping (one launch with tile_counters and
// T <= 32, see revs_op_dual_product_rows) and candidate lists; with phase bit 2 the candidate-list kernel is left to the
// caller (revs_agent_step_select runs it inside the home sweep's launch).  A driver that shards residences runs phase 1, all-reduces
// pnq, then runs phase 2.
static int dual_evaluate_impl(int32_t phase, int32_t m, int32_t T, const int64_t *node_ptr,
                                     const float *p_est, const float *p_sch, const float *gamma,
                                     const double *R, const double *Rt, const double *y,
                                     int32_t use_y, double kappa, double vlo, double vhi,
                                     int32_t kadd, int32_t ksplit, double *d_slabs,
                                     double *v_slabs, double *pnq, float *p_est_new,
                                     double *vfull, double *viol, double *partial,
                                     int64_t *cand_idx, int32_t *cand_cnt, double *cand_val,
                                     double *stats, double seq, uint32_t *tile_counters,
                                     const revs_tree_t *tree, void *stream) {
    REVS_REQUIRE(phase >= 1 && phase <= 7 && (!(phase & 4) || (phase & 2)) && y && pnq,
                 "revs_op_dual_evaluate: bad argument");
    int rc;
    if (phase & 1) {
        int nslab = ksplit;
        if (use_y && tree) {           // d = R^T y by the tree form: one slab
            REVS_REQUIRE(d_slabs, "revs_op_dual_evaluate_tree: d_slabs missing");
            rc = dual_shift_tree(m, T, tree, y, d_slabs, stream);
            if (rc != REVS_OK) return rc;
            nslab = 1;
        } else if (use_y) {
            REVS_REQUIRE(R && d_slabs, "revs_op_dual_evaluate: R / d_slabs missing");
            rc = revs_gemm_tn_f64_split(m, T, m, R, y, d_slabs, ksplit, stream);
            if (rc != REVS_OK) return rc;
        }
        rc = revs_op_dual_eval(m, T, node_ptr, p_est, p_sch, gamma, nslab,
                               use_y ? d_slabs : nullptr, kappa, pnq, p_est_new, stream);
        if (rc != REVS_OK) return rc;
    }
    if ((phase & 2) && tree) {      // R p of a radial feeder in O(nodes), rows (and selection) per slot
        return revs_op_dual_rows_tree(m, T, tree, pnq, y, vlo, vhi, kadd, vfull, viol, partial, nullptr,
                                      cand_idx, cand_cnt, cand_val, stats, seq, !(phase & 4), stream);
    }
    if (phase & 2) {
        REVS_REQUIRE(Rt && v_slabs, "revs_op_dual_evaluate: Rt / v_slabs missing");
        if (tile_counters && T <= 32 && (m + 31) / 32 <= 256) {
            // product and row bookkeeping in one launch, then the selection over its tiles
            REVS_REQUIRE(m <= 16384 && vfull && viol && partial && cand_idx && cand_cnt && cand_val &&
                         stats && vlo <= vhi && kadd >= 0, "revs_op_dual_evaluate: bad argument");
            rc = revs_op_dual_product_rows(m, T, Rt, pnq, pnq, y, vlo, vhi, ksplit, v_slabs, vfull,
                                           viol, partial, nullptr, tile_counters, stream);
            if (rc != REVS_OK) return rc;
            if (!(phase & 4)) {
                const SelectArgs sa{m, T, (m + 31) / 32, kadd, partial, y, vfull, viol, vlo, vhi, seq,
                                    cand_idx, cand_cnt, cand_val, stats};
                hipLaunchKernelGGL(op_dual_select_kernel, dim3(T), dim3(256), 0, S_(stream), sa);
                REVS_CHECK_LAUNCH("revs_op_dual_evaluate");
            }
        } else {
            rc = revs_gemm_tn_f64_split(m, T, m, Rt, pnq, v_slabs, ksplit, stream);
            if (rc != REVS_OK) return rc;
            rc = dual_rows_select(m, T, ksplit, v_slabs, pnq, y, vlo, vhi, kadd, vfull, viol, partial,
                                  cand_idx, cand_cnt, cand_val, stats, seq, (phase & 4) != 0, nullptr,
                                  stream);
            if (rc != REVS_OK) return rc;
        }
    }
    return REVS_OK;
}

extern "C" int revs_op_dual_evaluate(int32_t phase, int32_t m, int32_t T, const int64_t *node_ptr,
                                     const float *p_est, const float *p_sch, const float *gamma,
                                     const double *R, const double *Rt, const double *y,
                                     int32_t use_y, double kappa, double vlo, double vhi,
                                     int32_t kadd, int32_t ksplit, double *d_slabs,
                                     double *v_slabs, double *pnq, float *p_est_new,
                                     double *vfull, double *viol, double *partial,
                                     int64_t *cand_idx, int32_t *cand_cnt, double *cand_val,
                                     double *stats, double seq, uint32_t *tile_counters,
                                     void *stream) {
    return dual_evaluate_impl(phase, m, T, node_ptr, p_est, p_sch, gamma, R, Rt, y, use_y, kappa, vlo, vhi, kadd,
                              ksplit, d_slabs, v_slabs, pnq, p_est_new, vfull, viol, partial, cand_idx, cand_cnt,
                              cand_val, stats, seq, tile_counters, nullptr, stream);
}

extern "C" int revs_op_dual_evaluate_tree(int32_t phase, int32_t m, int32_t T, const int64_t *node_ptr,
                                          const float *p_est, const float *p_sch, const float *gamma,
                                          const double *R, const revs_tree_t *tree, const double *y,
                                          int32_t use_y, double kappa, double vlo, double vhi,
                                          int32_t kadd, int32_t ksplit, double *d_slabs, double *pnq,
                                          float *p_est_new, double *vfull, double *viol, double *partial,
                                          int64_t *cand_idx, int32_t *cand_cnt, double *cand_val,
                                          double *stats, double seq, void *stream) {
    REVS_REQUIRE(tree_ok_big(tree), "revs_op_dual_evaluate_tree: bad tree (at most %d nodes, a multiple of 8; of 16 beyond 8192)",
                 REVS_TREE_MAX);
    return dual_evaluate_impl(phase, m, T, node_ptr, p_est, p_sch, gamma, R, nullptr, y, use_y, kappa, vlo, vhi,
                              kadd, ksplit, d_slabs, nullptr, pnq, p_est_new, vfull, viol, partial, cand_idx,
                              cand_cnt, cand_val, stats, seq, nullptr, tree, stream);
}
