// The operator's model problem for slots with MORE candidate rows than the fast path holds -- gfx950.
//
// The dual Newton path (newton_kernels.hip) solves, per slot, a sign-constrained quadratic model over the rows that carry
// a multiplier plus the most violated rows without one.  Its kernels hold REVS_DUAL_AMAX = 128 rows per slot: the factor
// of the model's Hessian lives in one workgroup's LDS.  The reference hands Gurobi EVERY row (lpsolver.py:183-194) and
// always gets an answer; here a slot with more than 128 binding rows used to hand the ADMM iteration to the OSQP-form
// fallback, which is slow there and ends in REVS_ENOTCONV when it stops above tolerance.  This file is the same Newton
// iteration for up to REVS_DUAL_AMAX_BIG = 512 rows per slot, with everything that was sized by 128 in global memory:
//   op_big_select_kernel   candidate list: rows with y != 0 in row order, then the `kadd` most violated rows
//   op_big_gram_kernel     K_t = R_F N_t R_F^T over the list, K-split slabs (op_dual_gram_kernel with 8 x 8 tiles of 64)
//   op_big_bpp_kernel      the LCP u >= 0, K'u - c >= 0, u.(K'u - c) = 0 by block principal pivoting (Judice-Pires, the
//                          rule of op_dual_bpp_kernel), K'_BB = L D L^T by a blocked left-looking factorisation whose
//                          factor sits in global memory (2 MB per slot: L2-resident), 16 columns per panel, the panel's
//                          16 x 16 diagonal block factored in the registers of one wavefront (v_readlane broadcasts)
//   op_big_step_kernel     y_trial = y + alpha (yhat - y) on the listed rows
// A slow path by construction (one workgroup of 1024 threads per slot, ~0.2 ms per pivoting round at 400 rows): what it
// buys is that the iteration STAYS on the Newton path -- same stopping test, same line search, same answer to 1e-8.
#include "common.h"
#include "select_body.h"

namespace revs {

constexpr int kBig = REVS_DUAL_AMAX_BIG;
constexpr int kBigWords = kBig / 64;
static_assert(kBig % 64 == 0 && kBig <= 512 && 64 % 16 == 0, "the pivoting kernel keeps a row per thread pair: 1024 threads");

__device__ __forceinline__ double bcast_d(double v, int k) {          // v of lane k (k uniform, 0..63)
    const long long bb = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)bb, k);
    const int hi = __builtin_amdgcn_readlane((int)(bb >> 32), k);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// ---- candidate lists of up to kBig rows ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void op_big_select_kernel(
        int m, int T, const double *__restrict__ y, const double *__restrict__ vfull, const double *__restrict__ viol,
        double vlo, double vhi, int kadd, int64_t *__restrict__ cidx, int32_t *__restrict__ ccnt, double *__restrict__ cval) {
    const int t = blockIdx.x, tid = threadIdx.x;
    int64_t *ci = cidx + (int64_t)t * kBig;
    double *cs = cval + (int64_t)t * 3 * kBig, *cg = cs + kBig, *cy = cg + kBig;
    __shared__ int cnt_s[4], nv_s[4];
    __shared__ double best_v[2][4];
    __shared__ int best_i[2][4], chosen[kBig];
    __shared__ unsigned int taken[16384 / 32];
    for (int i = tid; i < 16384 / 32; i += 256) taken[i] = 0u;
    const int per = (m + 255) / 256, r0 = min(m, tid * per), r1 = min(m, r0 + per);
    int nsup = 0, nvl = 0;
    for (int r = r0; r < r1; ++r) {
        const double yv = y[(int64_t)r * T + t];
        nsup += yv != 0.0 ? 1 : 0;
        nvl += (yv == 0.0 && viol[(int64_t)r * T + t] > 0.0) ? 1 : 0;
    }
    const int incl = wave_incl_scan_i(nsup);
    const int nvw = (int)wave_sum_d((double)nvl);
    if ((tid & 63) == 63) cnt_s[tid >> 6] = incl;
    if ((tid & 63) == 0) nv_s[tid >> 6] = nvw;
    __syncthreads();
    int pos = incl - nsup;
    for (int w = 0; w < (tid >> 6); ++w) pos += cnt_s[w];
    const int ns = cnt_s[0] + cnt_s[1] + cnt_s[2] + cnt_s[3], nv = nv_s[0] + nv_s[1] + nv_s[2] + nv_s[3];
    if (ns > kBig) {                                  // (uniform) more multipliers than even this path holds
        if (tid == 0) ccnt[t] = -1;
        return;
    }
    for (int r = r0; r < r1 && nsup > 0; ++r) {
        const double yv = y[(int64_t)r * T + t];
        if (yv != 0.0) {
            ci[pos] = r;
            cs[pos] = yv > 0.0 ? 1.0 : -1.0;
            cg[pos] = vfull[(int64_t)r * T + t] - (yv > 0.0 ? vhi : vlo);
            cy[pos] = yv;
            ++pos;
        }
    }
    const int room = min(min(kadd, kBig - ns), nv);
    int added = 0;
    for (int k = 0; k < room; ++k) {                  // block-wide arg-max rounds: larger violation first, ties to the lower row
        double bv = 0.0;
        int bi = 0x7FFFFFFF;
        for (int r = tid; r < m; r += 256) {
            if ((taken[r >> 5] >> (r & 31)) & 1u) continue;
            const double x = y[(int64_t)r * T + t] == 0.0 ? viol[(int64_t)r * T + t] : 0.0;
            if (x > bv) { bv = x; bi = r; }           // (ascending r: ties keep the lower row)
        }
        const double wv = wave_max_d(bv);
        bi = wave_min_i(bv == wv && bv > 0.0 ? bi : 0x7FFFFFFF);
        const int pp = k & 1;
        if ((tid & 63) == 0) { best_v[pp][tid >> 6] = wv; best_i[pp][tid >> 6] = bi; }
        __syncthreads();
        bv = best_v[pp][0]; bi = best_i[pp][0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (best_v[pp][w] > bv || (best_v[pp][w] == bv && best_i[pp][w] < bi)) { bv = best_v[pp][w]; bi = best_i[pp][w]; }
        if (!(bv > 0.0)) break;                       // uniform
        if (tid == 0) { chosen[k] = bi; taken[bi >> 5] |= 1u << (bi & 31); }
        ++added;
        __syncthreads();
    }
    __syncthreads();
    if (tid < added) {
        const int bi = chosen[tid];
        const double v = vfull[(int64_t)bi * T + t];
        const bool up = v > vhi;
        ci[ns + tid] = bi;
        cs[ns + tid] = up ? 1.0 : -1.0;
        cg[ns + tid] = v - (up ? vhi : vlo);
        cy[ns + tid] = 0.0;
    }
    const int cnt = ns + added;
    if (tid == 0) ccnt[t] = cnt;
    for (int i = cnt + tid; i < kBig; i += 256) { ci[i] = 0; cs[i] = 1.0; cg[i] = 0.0; cy[i] = 0.0; }
}

// ---- K_t = R_F N_t R_F^T over the list: workgroup (t, ks, tile) computes one 64 x 64 tile over a slab of R's columns ----
__global__ __launch_bounds__(256) void op_big_gram_kernel(
        int m, int T, const double *__restrict__ R, const double *__restrict__ Nn, const int64_t *__restrict__ cidx,
        const int32_t *__restrict__ ccnt, int nks, double *__restrict__ Kslab) {
    constexpr int NTD = kBig / 64;
    const int t = blockIdx.x, ks = blockIdx.y, tid = threadIdx.x;
    const int bi = blockIdx.z / NTD, bj = blockIdx.z % NTD;
    const int a = ccnt[t];
    if (a <= 0 || bi * 64 >= a || bj * 64 >= a) return;
    constexpr int KC = 32;
    __shared__ double Ws[64][KC + 1], Rs_[64][KC + 1];
    __shared__ int64_t rows_i[64], rows_j[64];
    if (tid < 64) {
        rows_i[tid] = bi * 64 + tid < a ? cidx[(int64_t)t * kBig + bi * 64 + tid] : -1;
        rows_j[tid] = bj * 64 + tid < a ? cidx[(int64_t)t * kBig + bj * 64 + tid] : -1;
    }
    __syncthreads();
    const int ti = tid >> 4, tj = tid & 15;
    double acc[4][4] = {};
    const int chunk = (m + nks - 1) / nks;
    const int k0 = ks * chunk, k1 = min(m, k0 + chunk);
    for (int kk = k0; kk < k1; kk += KC) {
        const int c = kk + (tid & 31);
        const bool cin = c < k1;
        const double nv = cin ? Nn[(int64_t)c * T + t] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int rr = (tid >> 5) + 8 * q;
            const int64_t ri = rows_i[rr], rj = rows_j[rr];
            Ws[rr][tid & 31] = (ri >= 0 && cin) ? R[ri * m + c] * nv : 0.0;
            Rs_[rr][tid & 31] = (rj >= 0 && cin) ? R[rj * m + c] : 0.0;
        }
        __syncthreads();
#pragma unroll 8
        for (int cc = 0; cc < KC; ++cc) {
            double wv[4], rv[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) { wv[x] = Ws[ti * 4 + x][cc]; rv[x] = Rs_[tj * 4 + x][cc]; }
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int z = 0; z < 4; ++z) acc[x][z] += wv[x] * rv[z];
        }
        __syncthreads();
    }
    double *o = Kslab + ((int64_t)t * nks + ks) * kBig * kBig;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int z = 0; z < 4; ++z)
            o[(int64_t)(bi * 64 + ti * 4 + x) * kBig + bj * 64 + tj * 4 + z] = acc[x][z];
}

// ---- the model problem of one slot: block principal pivoting, factor in global memory --------------------------------
constexpr int kBigNT = 1024, kPanel = 16, kTPR = kBigNT / kPanel;      // kTPR threads per row of a panel (<= 64: one wavefront)
__global__ __launch_bounds__(kBigNT) void op_big_bpp_kernel(
        const double *__restrict__ Kslab, int nks, double inv_kappa, double *__restrict__ Kall, double *__restrict__ Lall,
        const int32_t *__restrict__ ccnt, const double *__restrict__ cval, double delta, int max_pivots,
        double *__restrict__ yhat, int32_t *__restrict__ info) {
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int a = ccnt[t];
    const double *cs = cval + (int64_t)t * 3 * kBig, *cg = cs + kBig, *cy = cg + kBig;
    double *yo = yhat + (int64_t)t * kBig;
    if (a <= 0) {                                     // (uniform; a == -1: more rows than this path holds -- nothing moves)
        for (int i = tid; i < kBig; i += kBigNT) yo[i] = cy[i];
        if (tid == 0) info[t] = a < 0 ? -998 : 0;
        return;
    }
    double *Kt = Kall + (int64_t)t * kBig * kBig, *Lf = Lall + (int64_t)t * kBig * kBig;
    __shared__ double s_s[kBig], c_s[kBig], u_s[kBig], w_s[kBig], d_s[kBig], idk_s[kBig], v_s[kBig], red_s[kBigNT / 64];
    __shared__ double Lb[kPanel][kPanel + 1], r_s[kPanel];
    __shared__ int bl[kBig];
    __shared__ unsigned long long Bsh[kBigWords], Vsh[kBigWords];
    __shared__ double dl_s, tolw_s, tolu_s;
    __shared__ int done_s, piv_s;
    // every thread gets the workgroup's sum / maximum of v (two barriers)
    auto block_sum = [&](double v) -> double {
        v = wave_sum_d(v);
        if (lane == 0) red_s[wave] = v;
        __syncthreads();
        double r = 0.0;
        for (int w = 0; w < kBigNT / 64; ++w) r += red_s[w];
        __syncthreads();
        return r;
    };
    auto block_max = [&](double v) -> double {
        v = wave_max_d(v);
        if (lane == 0) red_s[wave] = v;
        __syncthreads();
        double r = red_s[0];
        for (int w = 1; w < kBigNT / 64; ++w) r = fmax(r, red_s[w]);
        __syncthreads();
        return r;
    };
    // K = (sum of the K-split slabs) / kappa, fixed order
    for (int e = tid; e < a * a; e += kBigNT) {
        const int i = e / a, j = e - i * a;
        const double *src = Kslab + (int64_t)t * nks * kBig * kBig + (int64_t)i * kBig + j;
        double acc = 0.0;
        for (int q = 0; q < nks; ++q) acc += src[(int64_t)q * kBig * kBig];
        Kt[(int64_t)i * kBig + j] = acc * inv_kappa;
    }
    for (int i = tid; i < kBig; i += kBigNT) {
        const bool in = i < a;
        const double s = in ? cs[i] : 1.0;
        s_s[i] = s;
        u_s[i] = in ? fmax(s * cy[i], 0.0) : 0.0;
        w_s[i] = 0.0;
    }
    __syncthreads();
    const double tr = block_sum(tid < a ? Kt[(int64_t)tid * kBig + tid] : 0.0);
    if (!(tr > 0.0)) {                                // K = 0: no residence answers to these rows (uniform)
        for (int i = tid; i < kBig; i += kBigNT) yo[i] = i < a ? cy[i] : 0.0;
        if (tid == 0) info[t] = 0;
        return;
    }
    const double dl = delta * tr / a + 1e-300;
    auto kp = [&](int i, int j) -> double { return s_s[i] * s_s[j] * Kt[(int64_t)i * kBig + j] + (i == j ? dl : 0.0); };
    // out_i = sum_j K'_ij u_j: two threads per row, half of the columns each
    auto matvec_row = [&]() -> double {
        const int i = tid >> 1, h = tid & 1;
        double acc = 0.0;
        if (i < a) {
            const int j0 = h * ((a + 1) / 2), j1 = h ? a : (a + 1) / 2;
            for (int j = j0; j < j1; ++j) acc += kp(i, j) * u_s[j];
        }
        acc += __shfl_xor(acc, 1, 64);
        return acc;
    };
    {
        const double ku = matvec_row();
        if ((tid & 1) == 0 && (tid >> 1) < kBig) c_s[tid >> 1] = (tid >> 1) < a ? s_s[tid >> 1] * cg[tid >> 1] + ku : 0.0;
    }
    if (tid < kBig) {
        const unsigned long long B0 = __ballot(tid < a && u_s[tid] > 0.0);
        if (lane == 0) Bsh[wave] = B0;
    }
    if (tid == 0) { done_s = 0; piv_s = 0; }
    __syncthreads();
    const double cm = block_max(tid < kBig ? fabs(c_s[tid]) : 0.0);
    const double tolw = 1e-13 * cm, floor_ = dl * 1e-6;
    int ninf = kBig + 1, pcount = 3;                  // thread 0's pivoting state
    for (;;) {
        __syncthreads();
        unsigned long long B[kBigWords];
        int nb = 0;
#pragma unroll
        for (int w = 0; w < kBigWords; ++w) { B[w] = Bsh[w]; nb += __popcll(B[w]); }
        const unsigned long long Bw = Bsh[wave & (kBigWords - 1)];       // this wavefront's word (wavefronts 0..7)
        if (tid < kBig && ((Bw >> lane) & 1ull)) {
            int pos = __popcll(Bw & ((1ull << lane) - 1ull));
#pragma unroll
            for (int w = 0; w < kBigWords; ++w) pos += w < wave ? __popcll(B[w]) : 0;
            bl[pos] = tid;
        }
        __syncthreads();
        // ---- K'_BB = L D L^T, left-looking, kPanel columns at a time; Lf[i][k] = l_ik (i > k), d_s / idk_s = D, 1 / D ----
        for (int k0 = 0; k0 < nb; k0 += kPanel) {
            const int kb = min(kPanel, nb - k0);
            // 1. the panel's entries with the contributions of the columns before it taken out:
            //    P_ic = K'(i, k0 + c) - sum_{k < k0} l_ik d_k l_{k0+c, k}     (rows i >= k0 + c)
            const int npairs = (nb - k0) * kb;
            for (int e = tid; e < npairs; e += kBigNT) {
                const int ii = e / kb, c = e - ii * kb, i = k0 + ii, j = k0 + c;
                if (j > i) continue;
                double acc = kp(bl[i], bl[j]);
                const double *li = Lf + (int64_t)i * kBig, *lj = Lf + (int64_t)j * kBig;
                for (int k = 0; k < k0; ++k) acc -= li[k] * (d_s[k] * lj[k]);
                Lf[(int64_t)i * kBig + j] = acc;
            }
            __syncthreads();
            // 2. the diagonal block, in the registers of wavefront 0: lane r holds row r (columns <= r); column by column the
            //    pivot and the column's unscaled entries reach the other lanes through v_readlane
            if (wave == 0) {
                double row[kPanel];
#pragma unroll
                for (int c = 0; c < kPanel; ++c)
                    row[c] = (lane < kb && c <= lane) ? Lf[(int64_t)(k0 + lane) * kBig + k0 + c] : (c == lane ? 1.0 : 0.0);
#pragma unroll
                for (int c = 0; c < kPanel; ++c) {
                    const double dk = fmax(bcast_d(row[c], c), floor_);
                    const double rinv = 1.0 / dk;
                    const double l = lane > c ? row[c] * rinv : 0.0;
#pragma unroll
                    for (int cc = c + 1; cc < kPanel; ++cc) {
                        const double wcc = bcast_d(row[c], cc);          // lane cc's unscaled entry of column c
                        row[cc] -= lane >= cc ? l * wcc : 0.0;
                    }
                    row[c] = lane > c ? l : (lane == c ? dk : 0.0);
                }
                if (lane < kPanel) {
#pragma unroll
                    for (int c = 0; c < kPanel; ++c) Lb[lane][c] = c < lane ? row[c] : 0.0;
                    if (lane < kb) {
                        double dk = row[0];
#pragma unroll
                        for (int c = 1; c < kPanel; ++c) dk = c == lane ? row[c] : dk;
                        d_s[k0 + lane] = dk;
                        idk_s[k0 + lane] = 1.0 / dk;
#pragma unroll
                        for (int c = 0; c < kPanel; ++c)
                            if (c < lane) Lf[(int64_t)(k0 + lane) * kBig + k0 + c] = row[c];
                    }
                }
            }
            __syncthreads();
            // 3. the rows below the block, one thread each: w_ic = P_ic - sum_{cc < c} w_i,cc l_{c,cc}, l_ic = w_ic / d_c
            for (int i = k0 + kb + tid; i < nb; i += kBigNT) {
                double x[kPanel];
#pragma unroll
                for (int c = 0; c < kPanel; ++c) x[c] = c < kb ? Lf[(int64_t)i * kBig + k0 + c] : 0.0;
#pragma unroll
                for (int c = 1; c < kPanel; ++c)
#pragma unroll
                    for (int cc = 0; cc < c; ++cc) x[c] -= Lb[c][cc] * x[cc];
#pragma unroll
                for (int c = 0; c < kPanel; ++c)
                    if (c < kb) Lf[(int64_t)i * kBig + k0 + c] = x[c] * idk_s[k0 + c];
            }
            __syncthreads();
        }
        // ---- L D L^T x = c_B ----
        for (int i = tid; i < kBig; i += kBigNT) v_s[i] = i < nb ? c_s[bl[i]] : 0.0;
        __syncthreads();
        for (int k0 = 0; k0 < nb; k0 += kPanel) {     // forward: L z = c_B
            const int kb = min(kPanel, nb - k0);
            {   // the block's rows minus what the components before the block contribute: kTPR threads per row
                const int r = tid / kTPR, part = tid % kTPR, i = k0 + r;
                double acc = 0.0;
                if (r < kb)
                    for (int k = part; k < k0; k += kTPR) acc += Lf[(int64_t)i * kBig + k] * v_s[k];
#pragma unroll
                for (int dd = kTPR / 2; dd >= 1; dd >>= 1) acc += __shfl_xor(acc, dd, 64);
                if (part == 0 && r < kPanel) r_s[r] = r < kb ? v_s[i] - acc : 0.0;
            }
            __syncthreads();
            if (wave == 0) {                          // inside the block: lane r holds row r of L's block
                double lrow[kPanel], val = lane < kPanel ? r_s[lane & (kPanel - 1)] : 0.0;
#pragma unroll
                for (int c = 0; c < kPanel; ++c)
                    lrow[c] = (lane < kb && c < lane) ? Lf[(int64_t)(k0 + lane) * kBig + k0 + c] : 0.0;
#pragma unroll
                for (int c = 0; c < kPanel; ++c) {
                    const double zc = bcast_d(val, c);
                    val -= lrow[c] * zc;              // (zero for lanes <= c)
                }
                if (lane < kb) v_s[k0 + lane] = val;
            }
            __syncthreads();
        }
        for (int i = tid; i < nb; i += kBigNT) v_s[i] *= idk_s[i];
        __syncthreads();
        for (int k0 = ((nb - 1) / kPanel) * kPanel; k0 >= 0; k0 -= kPanel) {      // backward: L^T x = D^-1 z
            const int kb = min(kPanel, nb - k0);
            if (wave == 0) {                          // inside the block: lane c holds column c of L's block
                double lcol[kPanel], val = lane < kb ? v_s[k0 + lane] : 0.0;
#pragma unroll
                for (int cc = 0; cc < kPanel; ++cc)
                    lcol[cc] = (lane < kb && cc < kb && cc > lane) ? Lf[(int64_t)(k0 + cc) * kBig + k0 + lane] : 0.0;
#pragma unroll
                for (int cc = kPanel - 1; cc >= 0; --cc) {
                    const double xc = bcast_d(val, cc);                   // (final: every later component has been taken out)
                    val -= lcol[cc] * xc;             // (zero for lanes >= cc)
                }
                if (lane < kb) v_s[k0 + lane] = val;
            }
            __syncthreads();
            for (int i = tid; i < k0; i += kBigNT) {  // ... and out of the components before the block
                double acc = 0.0;
                for (int c = 0; c < kb; ++c) acc += Lf[(int64_t)(k0 + c) * kBig + i] * v_s[k0 + c];
                v_s[i] -= acc;
            }
            __syncthreads();
        }
        for (int i = tid; i < kBig; i += kBigNT) u_s[i] = 0.0;
        __syncthreads();
        for (int i = tid; i < nb; i += kBigNT) u_s[bl[i]] = v_s[i];
        __syncthreads();
        {
            const double ku = matvec_row();
            if ((tid & 1) == 0 && (tid >> 1) < kBig) w_s[tid >> 1] = (tid >> 1) < a ? ku - c_s[tid >> 1] : 0.0;
        }
        __syncthreads();
        const double um = block_max(tid < kBig ? fabs(u_s[tid]) : 0.0);
        const double tolu = 1e-13 * um;
        if (tid < kBig) {
            const bool inB = (Bw >> lane) & 1ull;
            const bool bad = tid < a && (inB ? (u_s[tid] < -tolu) : (w_s[tid] < -tolw));
            const unsigned long long V = __ballot(bad);
            if (lane == 0) Vsh[wave] = V;
        }
        __syncthreads();
        if (tid == 0) {
            int nv = 0, top = -1;
            for (int w = 0; w < kBigWords; ++w) {
                nv += __popcll(Vsh[w]);
                if (Vsh[w]) top = w * 64 + 63 - __clzll((long long)Vsh[w]);
            }
            const int pv = ++piv_s;
            if (nv == 0) done_s = 1;
            else if (pv >= max_pivots) done_s = 2;
            else if (nv < ninf || pcount > 0) {
                if (nv < ninf) { ninf = nv; pcount = 3; } else --pcount;
                for (int w = 0; w < kBigWords; ++w) Bsh[w] = B[w] ^ Vsh[w];
            } else {
                Bsh[top >> 6] ^= 1ull << (top & 63);              // (Bsh still holds this round's set)
            }
        }
        __syncthreads();
        if (done_s) break;
    }
    for (int i = tid; i < kBig; i += kBigNT) yo[i] = i < a ? s_s[i] * fmax(u_s[i], 0.0) : 0.0;
    if (tid == 0) info[t] = done_s == 1 ? piv_s : -piv_s;
}

// ---- y_trial = y, then y_trial = y + alpha (yhat - y) on the slot's listed rows; lin_out[8 t] = gradient . step ------
__global__ __launch_bounds__(256) void op_big_step_kernel(
        int T, const int64_t *__restrict__ cidx, const int32_t *__restrict__ ccnt, const double *__restrict__ cval,
        const double *__restrict__ yhat, const double *__restrict__ alpha, const double *__restrict__ ycopy, int m,
        double *__restrict__ ytrial, double *__restrict__ lin_out) {
    const int t = blockIdx.x, tid = threadIdx.x;
    for (int r = tid; r < m; r += 256) ytrial[(int64_t)r * T + t] = ycopy[(int64_t)r * T + t];
    __syncthreads();
    const int a = ccnt[t];
    const double al = alpha[t];
    const double *cg = cval + (int64_t)t * 3 * kBig + kBig, *cy = cg + kBig;
    double lin = 0.0;
    for (int i = tid; i < a; i += 256) {
        const double yo = cy[i], yh = yhat[(int64_t)t * kBig + i];
        const double yn = al == 1.0 ? yh : (al == 0.0 ? yo : yo + al * (yh - yo));     // (a full step lands exactly on yhat)
        ytrial[cidx[(int64_t)t * kBig + i] * T + t] = yn;
        lin += cg[i] * (yn - yo);
    }
    lin = wave_sum_d(lin);
    __shared__ double ls[4];
    if ((tid & 63) == 0) ls[tid >> 6] = lin;
    __syncthreads();
    if (tid == 0) lin_out[t * 8] = ((ls[0] + ls[1]) + ls[2]) + ls[3];
}

}  // namespace revs

using namespace revs;
#define S_(stream) ((hipStream_t)(stream))

extern "C" int revs_op_dual_select_big(int32_t m, int32_t T, const double *y, const double *vfull, const double *viol,
                                       double vlo, double vhi, int32_t kadd, int64_t *cand_idx, int32_t *cand_cnt,
                                       double *cand_val, void *stream) {
    REVS_REQUIRE(m > 0 && m <= 16384 && T > 0 && y && vfull && viol && cand_idx && cand_cnt && cand_val && vlo <= vhi &&
                 kadd >= 0 && kadd <= REVS_DUAL_AMAX_BIG, "revs_op_dual_select_big: bad argument");
    hipLaunchKernelGGL(op_big_select_kernel, dim3(T), dim3(256), 0, S_(stream), m, T, y, vfull, viol, vlo, vhi, kadd,
                       cand_idx, cand_cnt, cand_val);
    REVS_CHECK_LAUNCH("revs_op_dual_select_big");
    return REVS_OK;
}

extern "C" int revs_op_dual_model_big(int32_t m, int32_t T, const double *R, const double *n_free, const int64_t *cand_idx,
                                      const int32_t *cand_cnt, const double *cand_val, double kappa, double delta,
                                      int32_t max_pivots, int32_t nks, double *k_slabs, double *k_full, double *l_factor,
                                      double *yhat, int32_t *info, void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && R && n_free && cand_idx && cand_cnt && cand_val && kappa > 0 && delta >= 0 &&
                 max_pivots > 0 && nks >= 1 && nks <= 64 && k_slabs && k_full && l_factor && l_factor != k_full && yhat && info,
                 "revs_op_dual_model_big: bad argument");
    constexpr int NTD = REVS_DUAL_AMAX_BIG / 64;
    hipLaunchKernelGGL(op_big_gram_kernel, dim3(T, nks, NTD * NTD), dim3(256), 0, S_(stream), m, T, R, n_free, cand_idx,
                       cand_cnt, nks, k_slabs);
    hipLaunchKernelGGL(op_big_bpp_kernel, dim3(T), dim3(kBigNT), 0, S_(stream), k_slabs, nks, 1.0 / kappa, k_full,
                       l_factor, cand_cnt, cand_val, delta, max_pivots, yhat, info);
    REVS_CHECK_LAUNCH("revs_op_dual_model_big");
    return REVS_OK;
}

extern "C" int revs_op_dual_step_big(int32_t T, const int64_t *cand_idx, const int32_t *cand_cnt, const double *cand_val,
                                     const double *yhat, const double *alpha, const double *y, int32_t m, double *y_trial,
                                     double *lin_out, void *stream) {
    REVS_REQUIRE(T > 0 && cand_idx && cand_cnt && cand_val && yhat && alpha && y && m > 0 && y_trial && y_trial != y && lin_out,
                 "revs_op_dual_step_big: bad argument");
    hipLaunchKernelGGL(op_big_step_kernel, dim3(T), dim3(256), 0, S_(stream), T, cand_idx, cand_cnt, cand_val, yhat, alpha, y,
                       m, y_trial, lin_out);
    REVS_CHECK_LAUNCH("revs_op_dual_step_big");
    return REVS_OK;
}
