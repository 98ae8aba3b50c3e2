// v = R p on a radial feeder as three prefix sums (include/revs_admm.h, "the feeder as a
// tree"), one workgroup of 256 threads per slot: as its own kernel (revs_tree_voltage) or as
// the first T workgroups of the streaming sweep's launch, where it judges the voltage rows of
// the estimate the previous sweep prepared while the residences are being solved.
#pragma once
#include "common.h"

namespace revs {

constexpr int kTreeIpt = REVS_TREE_MAX / 256;        // positions per thread (strided: j = tid + 256 i)
static_assert(kTreeIpt * 256 == REVS_TREE_MAX, "REVS_TREE_MAX must be a multiple of 256");

// dynamic LDS of a launch that carries the tree workgroups: one leading zero, the scan /
// gather buffer (n doubles), and two sets of wave totals
__host__ __device__ inline size_t tree_lds_bytes(int n) {
    return sizeof(double) * ((size_t)n + 1 + 8);
}

__device__ __forceinline__ double wave_incl_scan_d(double v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// In-place inclusive prefix sum of base[0..n) in LDS by the whole workgroup (256 threads),
// 256 positions per round with a running carry; fixed order: bitwise reproducible.  One
// barrier per round (the wave totals ping-pong between two sets) and one at the end.
// Deliberately NOT unrolled: this runs inside the residence sweep's kernel and must stay far
// below its register budget (see tree_rmax).
__device__ __forceinline__ void lds_scan_inplace(double *base, int n, double *tot) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double carry = 0.0;
    int pp = 0;
#pragma unroll 1
    for (int c0 = 0; c0 < n; c0 += 256, pp ^= 4) {
        const int j = c0 + tid;
        double x = j < n ? base[j] : 0.0;
        x = wave_incl_scan_d(x, lane);
        if (lane == 63) tot[pp + wave] = x;
        __syncthreads();
        const double t0 = tot[pp], t1 = tot[pp + 1], t2 = tot[pp + 2], t3 = tot[pp + 3];
        x += carry + (wave > 0 ? t0 : 0.0) + (wave > 1 ? t1 : 0.0) + (wave > 2 ? t2 : 0.0);
        carry += ((t0 + t1) + t2) + t3;
        if (j < n) base[j] = x;
    }
    __syncthreads();
}

struct TreeArgs {
    int32_t n;
    const int32_t *src, *end, *eo, *cle;
    const double *w;
};

// Largest violation max(v - vhi, vlo - v, 0) over the checked rows of slot t (every thread
// gets it); v_out[src][t] = v when v_out != NULL.  `lds`: tree_lds_bytes(n) bytes.
// Register budget: this body runs inside the residence sweep's kernel, whose occupancy (8
// wavefronts per SIMD, 64 VGPRs) it must not lower.  So the prefix sums run in place in LDS
// (base[-1] = 0 makes an exclusive prefix a read at j - 1), and only two 8-double arrays ever
// live in registers across a barrier: the values being re-ordered, and Pre.
__device__ __forceinline__ double tree_rmax(const TreeArgs &tr, const double *p, int T, int t,
                                            double vlo, double vhi, double *lds, double *v_out) {
    const int tid = threadIdx.x, n = tr.n;
    double *base = lds + 1, *tot = lds + 1 + n;
    if (tid == 0) lds[0] = 0.0;
    // C: prefix of the injections in preorder
#pragma unroll
    for (int i = 0; i < kTreeIpt; ++i) {
        const int j = tid + 256 * i;
        if (j < n) {
            const int s = tr.src[j];
            base[j] = s >= 0 ? p[(int64_t)s * T + t] : 0.0;
        }
    }
    __syncthreads();
    lds_scan_inplace(base, n, tot);
    // w'_j = w_j (C[end_j] - C[j]),  C[j] = base[j - 1]
    double a[kTreeIpt];
#pragma unroll
    for (int i = 0; i < kTreeIpt; ++i) {
        const int j = tid + 256 * i;
        a[i] = j < n ? tr.w[j] * (base[tr.end[j] - 1] - base[j - 1]) : 0.0;
    }
    __syncthreads();                                            // every read of C is done
#pragma unroll
    for (int i = 0; i < kTreeIpt; ++i) {
        const int j = tid + 256 * i;
        if (j < n) base[j] = a[i];
    }
    __syncthreads();
    // the same values in end-order (held in registers), then Pre = prefix of w' in preorder
#pragma unroll
    for (int i = 0; i < kTreeIpt; ++i) {
        const int k = tid + 256 * i;
        a[i] = k < n ? base[tr.eo[k]] : 0.0;
    }
    __syncthreads();                                            // every read of w' is done
    lds_scan_inplace(base, n, tot);
    double pre[kTreeIpt];
#pragma unroll
    for (int i = 0; i < kTreeIpt; ++i) {
        const int j = tid + 256 * i;
        pre[i] = j < n ? base[j] : 0.0;
    }
    __syncthreads();
    // F: prefix of w' in end-order
#pragma unroll
    for (int i = 0; i < kTreeIpt; ++i) {
        const int k = tid + 256 * i;
        if (k < n) base[k] = a[i];
    }
    __syncthreads();
    lds_scan_inplace(base, n, tot);
    // v_j = Pre[j] - F_excl[cle[j]] on the checked rows
    double rmax = 0.0;
#pragma unroll
    for (int i = 0; i < kTreeIpt; ++i) {
        const int j = tid + 256 * i;
        const int s = j < n ? tr.src[j] : -1;
        if (s >= 0) {
            const double v = pre[i] - base[tr.cle[j] - 1];
            rmax = fmax(rmax, fmax(fmax(v - vhi, vlo - v), 0.0));
            if (v_out) v_out[(int64_t)s * T + t] = v;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) rmax = fmax(rmax, __shfl_xor(rmax, d, 64));
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
