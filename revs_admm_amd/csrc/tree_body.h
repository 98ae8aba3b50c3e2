// v = R p on a radial feeder as three prefix sums (include/revs_admm.h, "the feeder as a
// tree"), one workgroup of 256 threads per slot: as its own kernel (revs_tree_voltage) or as
// the first T workgroups of the streaming sweep's launch, where it judges the voltage rows of
// the estimate the previous sweep prepared while the residences are being solved.
#pragma once
#include "common.h"

namespace revs {

constexpr int kTreeIpt = REVS_TREE_MAX / 256;        // consecutive positions per thread: j = 8 tid + i
static_assert(kTreeIpt == 8, "thread-local vectors below are written for 8 positions");

// dynamic LDS of a launch that carries the tree workgroups: two leading zeros (the second is
// element -1 of the 16-byte-aligned scan / gather buffer), the buffer, two sets of wave totals
__host__ __device__ inline size_t tree_lds_bytes(int) {
    return sizeof(double) * (2 + REVS_TREE_MAX + 8);
}

// Inclusive prefix sum over the 64 lanes of a wavefront, doubles, on the DPP network: four
// row_shr steps inside each row of 16 lanes, then row_bcast:15 / row_bcast:31 carry the row
// totals across (gfx9 DPP controls) -- six steps of two 32-bit moves and one add, no LDS
// permute (a __shfl_up of a double is two ds_bpermute round trips per step).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_incl_scan_d(double v) {
    v += dpp_d<0x111, 0xf>(v);        // row_shr:1 (lanes without a source read 0)
    v += dpp_d<0x112, 0xf>(v);        // row_shr:2
    v += dpp_d<0x114, 0xf>(v);        // row_shr:4
    v += dpp_d<0x118, 0xf>(v);        // row_shr:8
    v += dpp_d<0x142, 0xa>(v);        // row_bcast:15 into rows 1 and 3 (others add 0)
    v += dpp_d<0x143, 0xc>(v);        // row_bcast:31 into rows 2 and 3
    return v;
}

// Sum of `tot` over all threads before this one in the workgroup (256 threads, fixed order:
// bitwise reproducible).  One barrier; `red` (4 doubles) must not be rewritten before the
// caller's next barrier.
__device__ __forceinline__ double block_excl_offset(double tot, double *red) {
    const int tid = threadIdx.x, wave = tid >> 6;
    const double incl = wave_incl_scan_d(tot);
    if ((tid & 63) == 63) red[wave] = incl;
    __syncthreads();
    double off = incl - tot;
    if (wave > 0) off += red[0];
    if (wave > 1) off += red[1];
    if (wave > 2) off += red[2];
    return off;
}

struct TreeArgs {
    int32_t n;                              // a multiple of 8 (the host pads with weightless roots)
    const unsigned long long *pack;         // per position: src + 1 | end << 16 | eo << 32 | cle << 48
    const double *w;
};

struct alignas(16) TreeU2 { unsigned long long v[2]; };
struct alignas(16) TreeD2 { double v[2]; };

// The voltages themselves: v[i] = (R p)[src] at this thread's positions 8 tid + i (0 where the
// position carries no checked row), pk[i] = the positions' packed indices (row = (pk & 0xFFFF) - 1).
// p_clear != NULL (the same array as p, writable): every node sum read is set to zero behind the
// read -- the block verdicts leave the ring slice they judged ready for the next accumulation.
// Ends with the workgroup past a barrier; lds[2 ..] may be reused by the caller after its own barrier.
__device__ __forceinline__ void tree_voltage(const TreeArgs &tr, const double *p, int T, int t,
                                             double *lds, double (&a)[8], unsigned long long (&pk)[8],
                                             double *p_clear) {
    const int tid = threadIdx.x, n = tr.n, j0 = 8 * tid;
    const bool act = j0 < n;
    double *base = lds + 2, *red0 = lds + 2 + REVS_TREE_MAX, *red1 = red0 + 4;
    if (tid == 0) lds[1] = 0.0;                                 // base[-1]
    double b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { pk[i] = 0ull; a[i] = 0.0; b[i] = 0.0; }
    if (act) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            const TreeU2 u = *reinterpret_cast<const TreeU2 *>(tr.pack + j0 + i);
            const TreeD2 wv = *reinterpret_cast<const TreeD2 *>(tr.w + j0 + i);
            pk[i] = u.v[0]; pk[i + 1] = u.v[1];
            b[i] = wv.v[0]; b[i + 1] = wv.v[1];                 // (b holds the weights until phase 2)
        }
        // C: inclusive prefix of the injections in preorder
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int s = (int)(pk[i] & 0xFFFFu) - 1;
            a[i] = s >= 0 ? p[(int64_t)s * T + t] : 0.0;
        }
        if (p_clear) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int s = (int)(pk[i] & 0xFFFFu) - 1;
                if (s >= 0) p_clear[(int64_t)s * T + t] = 0.0;
            }
        }
    }
#pragma unroll
    for (int i = 1; i < 8; ++i) a[i] += a[i - 1];
    const double cex = block_excl_offset(a[7], red0);           // C_excl at j0
    if (act) {
#pragma unroll
        for (int i = 0; i < 8; i += 2)
            *reinterpret_cast<TreeD2 *>(base + j0 + i) = TreeD2{{a[i] + cex, a[i + 1] + cex}};
    }
    __syncthreads();
    // w'_j = w_j (C[end_j] - C[j]),  C[j] = base[j - 1] (own positions: registers)
    if (act) {
#pragma unroll
        for (int i = 7; i >= 1; --i)
            a[i] = b[i] * (base[(int)((pk[i] >> 16) & 0xFFFFu) - 1] - (a[i - 1] + cex));
        a[0] = b[0] * (base[(int)((pk[0] >> 16) & 0xFFFFu) - 1] - cex);
    }
    __syncthreads();                                            // every read of C is done
    if (act) {
#pragma unroll
        for (int i = 0; i < 8; i += 2)
            *reinterpret_cast<TreeD2 *>(base + j0 + i) = TreeD2{{a[i], a[i + 1]}};
    }
    __syncthreads();
    // the same values in end-order, then both prefixes: Pre over preorder (a), F over end-order (b)
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = act ? base[(int)((pk[i] >> 32) & 0xFFFFu)] : 0.0;
#pragma unroll
    for (int i = 1; i < 8; ++i) { a[i] += a[i - 1]; b[i] += b[i - 1]; }
    const double pex = block_excl_offset(a[7], red1);           // (its barrier: every read of w' is done)
    const double fex = block_excl_offset(b[7], red0);
    if (act) {
#pragma unroll
        for (int i = 0; i < 8; i += 2)
            *reinterpret_cast<TreeD2 *>(base + j0 + i) = TreeD2{{b[i] + fex, b[i + 1] + fex}};
    }
    __syncthreads();
    // v_j = Pre[j] - F_excl[cle[j]] on the checked rows,  F_excl[c] = base[c - 1]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int s = (int)(pk[i] & 0xFFFFu) - 1;
        a[i] = (act && s >= 0) ? (a[i] + pex) - base[(int)(pk[i] >> 48) - 1] : 0.0;
    }
}

// Largest violation max(v - vhi, vlo - v, 0) over the checked rows of slot t (every thread
// gets it); v_out[src][t] = v when v_out != NULL.  `lds`: tree_lds_bytes() bytes, 16-byte aligned.
// Two things shape this body.  Registers: it runs inside the residence sweep's kernel, whose
// occupancy (8 wavefronts per SIMD, 64 VGPRs) it must not lower -- two 8-double vectors per
// thread (thread tid owns positions 8 tid .. 8 tid + 7) beside the static data.  Latency: its 24
// or 96 workgroups are the launch's critical path when memory is saturated by the sweep (every
// dependent global load costs 2-3 us there), so ALL the static data of a thread -- four 16-bit
// indices per position packed in one 64-bit word, and the weights -- are requested at the very
// top, and the only dependent global access is the gather of the node sums behind them.
__device__ __forceinline__ double tree_rmax(const TreeArgs &tr, const double *p, int T, int t,
                                            double vlo, double vhi, double *lds, double *v_out,
                                            double *p_clear = nullptr) {
    const int tid = threadIdx.x;
    double *red1 = lds + 2 + REVS_TREE_MAX + 4;
    unsigned long long pk[8];
    double a[8];
    tree_voltage(tr, p, T, t, lds, a, pk, p_clear);
    double rmax = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int s = (int)(pk[i] & 0xFFFFu) - 1;
        if (s >= 0) {
            const double v = a[i];
            rmax = fmax(rmax, fmax(fmax(v - vhi, vlo - v), 0.0));
            if (v_out) v_out[(int64_t)s * T + t] = v;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) rmax = fmax(rmax, __shfl_xor(rmax, d, 64));
    if ((tid & 63) == 0) red1[tid >> 6] = rmax;
    __syncthreads();
    return fmax(fmax(red1[0], red1[1]), fmax(red1[2], red1[3]));
}

// Control block of the streaming steady state (device memory, owned by the plan).
struct StreamCtl {
    unsigned int bad_seq;                 // sequence number of the last launch whose verdict failed (0: none yet)
    unsigned int arrive;                  // tree workgroups of the current launch that are done
    unsigned long long rmax_bits;         // max over their slots (bit pattern of a double >= 0)
};
constexpr int kRecRing = 1024;             // records double[kRecRing][4] in pinned host memory

}  // namespace revs
