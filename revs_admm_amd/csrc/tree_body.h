// v = R p on a radial feeder as three prefix sums (include/revs_admm.h, "the feeder as a
// tree"), one workgroup of 256 threads per slot: as its own kernel (revs_tree_voltage) or as
// the first T workgroups of the streaming sweep's launch, where it judges the voltage rows of
// the estimate the previous sweep prepared while the residences are being solved.
#pragma once
#include "common.h"

namespace revs {

constexpr int kTreeIpt = REVS_TREE_MAX / 256;        // positions per thread (strided: j = tid + 256 i)
static_assert(kTreeIpt * 256 == REVS_TREE_MAX, "REVS_TREE_MAX must be a multiple of 256");

// dynamic LDS of a launch that carries the tree workgroups: the gather buffer (n + 1 doubles)
// and the wave totals of one block scan (kTreeIpt chunks x 4 waves)
__host__ __device__ inline size_t tree_lds_bytes(int n) {
    return sizeof(double) * ((size_t)n + 1 + kTreeIpt * 4 + 4);
}

__device__ __forceinline__ double wave_incl_scan_d(double v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// Block-wide scan of x[i] at positions j = tid + 256 i (i < kTreeIpt), in position order.
// On return x[i] is the INCLUSIVE prefix at its position; the function returns the grand
// total.  Fixed order: bitwise reproducible.  `tot` = kTreeIpt * 4 doubles of LDS.
__device__ __forceinline__ double block_scan_strided(double (&x)[kTreeIpt], double *tot) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < kTreeIpt; ++i) {
        x[i] = wave_incl_scan_d(x[i], lane);
        if (lane == 63) tot[i * 4 + wave] = x[i];
    }
    __syncthreads();
    double run = 0.0;
#pragma unroll
    for (int i = 0; i < kTreeIpt; ++i) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
            if (w < wave) x[i] += tot[i * 4 + w];   // the waves before mine in this chunk
        x[i] += run;                            // the chunks before this one
        run += ((tot[i * 4 + 0] + tot[i * 4 + 1]) + tot[i * 4 + 2]) + tot[i * 4 + 3];
    }
    __syncthreads();                            // `tot` may be reused
    return run;
}

struct TreeArgs {
    int32_t n;
    const int32_t *src, *end, *eo, *cle;
    const double *w;
};

// Largest violation max(v - vhi, vlo - v, 0) over the checked rows of slot t (valid in
// thread 0); v_out[src][t] = v when v_out != NULL.  `lds`: tree_lds_bytes(n) bytes.
__device__ __forceinline__ double tree_rmax(const TreeArgs &tr, const double *__restrict__ p,
                                            int T, int t, double vlo, double vhi, double *lds,
                                            double *__restrict__ v_out) {
    const int tid = threadIdx.x, n = tr.n;
    double *buf = lds, *tot = lds + n + 1;
    int sj[kTreeIpt];
    double a[kTreeIpt], b[kTreeIpt];
    // injections in preorder
#pragma unroll
    for (int i = 0; i < kTreeIpt; ++i) {
        const int j = tid + 256 * i;
        sj[i] = j < n ? tr.src[j] : -1;
    }
#pragma unroll
    for (int i = 0; i < kTreeIpt; ++i) a[i] = sj[i] >= 0 ? p[(int64_t)sj[i] * T + t] : 0.0;
#pragma unroll
    for (int i = 0; i < kTreeIpt; ++i) b[i] = a[i];
    const double total = block_scan_strided(b, tot);            // inclusive
#pragma unroll
    for (int i = 0; i < kTreeIpt; ++i) {
        const int j = tid + 256 * i;
        b[i] -= a[i];                                           // exclusive: C[j]
        if (j < n) buf[j] = b[i];
    }
    if (tid == 0) buf[n] = total;
    __syncthreads();
    // w'_j = w_j (C[end_j] - C[j])
#pragma unroll
    for (int i = 0; i < kTreeIpt; ++i) {
        const int j = tid + 256 * i;
        a[i] = j < n ? tr.w[j] * (buf[tr.end[j]] - b[i]) : 0.0;
    }
    __syncthreads();                                            // every read of C is done
#pragma unroll
    for (int i = 0; i < kTreeIpt; ++i) {
        const int j = tid + 256 * i;
        if (j < n) buf[j] = a[i];
    }
    __syncthreads();
    // the same values in end-order, exclusive prefix F
#pragma unroll
    for (int i = 0; i < kTreeIpt; ++i) {
        const int k = tid + 256 * i;
        b[i] = k < n ? buf[tr.eo[k]] : 0.0;
    }
    __syncthreads();                                            // every read of w' is done
    {
        double c[kTreeIpt];
#pragma unroll
        for (int i = 0; i < kTreeIpt; ++i) c[i] = b[i];
        const double tf = block_scan_strided(c, tot);
#pragma unroll
        for (int i = 0; i < kTreeIpt; ++i) {
            const int k = tid + 256 * i;
            if (k < n) buf[k] = c[i] - b[i];
        }
        if (tid == 0) buf[n] = tf;
    }
    // inclusive prefix of w' in preorder (registers), then v_j = Pre[j] - F[cle[j]]
    block_scan_strided(a, tot);                                 // (its barriers publish F too)
    double rmax = 0.0;
#pragma unroll
    for (int i = 0; i < kTreeIpt; ++i) {
        const int j = tid + 256 * i;
        if (j < n && sj[i] >= 0) {
            const double v = a[i] - buf[tr.cle[j]];
            rmax = fmax(rmax, fmax(fmax(v - vhi, vlo - v), 0.0));
            if (v_out) v_out[(int64_t)sj[i] * T + t] = v;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) rmax = fmax(rmax, __shfl_xor(rmax, d, 64));
    __syncthreads();
    if ((tid & 63) == 0) tot[tid >> 6] = rmax;
    __syncthreads();
    return fmax(fmax(tot[0], tot[1]), fmax(tot[2], tot[3]));
}

// Control block of the streaming steady state (device memory, owned by the plan).
struct StreamCtl {
    unsigned int bad_seq;                 // smallest sequence number whose verdict failed; ~0u: none
    unsigned int arrive;                  // tree workgroups of the current launch that are done
    unsigned long long rmax_bits;         // max over their slots (bit pattern of a double >= 0)
};
constexpr int kRecRing = 64;              // records double[kRecRing][4] in pinned host memory

}  // namespace revs
