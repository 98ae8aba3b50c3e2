// v = R p on a radial feeder as three prefix sums (include/revs_admm.h, "the feeder as a
// tree"), one workgroup per slot: as its own kernel (revs_tree_voltage), inside the verdict
// launches of the streaming steady state, in the Newton evaluations' row kernels, or as the first T
// workgroups of a sweep's launch (the form in which every launch judges itself).
// Shapes: NT threads own IPT consecutive positions each -- 256 x 8 up to 2048 tree nodes (the only
// shape inside a sweep's launch, whose workgroups are 256 threads and whose registers and LDS set
// the sweep's occupancy), 512 x 8, 1024 x 8 and 1024 x 16 up to REVS_TREE_MAX = 16 384 in the
// stand-alone launches.
#pragma once
#include "common.h"

namespace revs {

constexpr int kTreeSweepMax = REVS_TREE_SWEEP_MAX;   // 256 threads x 8 positions

struct TreeShape { int nt, ipt; };
__host__ __device__ inline TreeShape tree_shape(int n) {
    if (n <= 2048) return {256, 8};
    if (n <= 4096) return {512, 8};
    if (n <= 8192) return {1024, 8};
    return {1024, 16};
}
// dynamic LDS of a launch that carries tree workgroups: two leading zeros (the second is
// element -1 of the 16-byte-aligned scan / gather buffer), the buffer, two sets of wave totals
__host__ __device__ inline size_t tree_lds_bytes(int n) {
    const TreeShape sh = tree_shape(n);
    return sizeof(double) * (2 + (size_t)sh.nt * sh.ipt + 2 * (sh.nt / 64));
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

// Sum of `tot` over all threads before this one in the workgroup (NT threads, fixed order:
// bitwise reproducible).  One barrier; `red` (NT / 64 doubles) must not be rewritten before the
// caller's next barrier.
template <int NT>
__device__ __forceinline__ double block_excl_offset(double tot, double *red) {
    const int tid = threadIdx.x, wave = tid >> 6;
    const double incl = wave_incl_scan_d(tot);
    if ((tid & 63) == 63) red[wave] = incl;
    __syncthreads();
    double off = incl - tot;
#pragma unroll
    for (int w = 0; w < NT / 64 - 1; ++w)
        if (wave > w) off += red[w];
    return off;
}

struct TreeArgs {
    int32_t n;                              // a multiple of the shape's IPT (the host pads with weightless roots)
    const unsigned long long *pack;         // per position: src + 1 | end << 16 | eo << 32 | cle << 48
    const double *w;
};

struct alignas(16) TreeU2 { unsigned long long v[2]; };
struct alignas(16) TreeD2 { double v[2]; };

// The voltages themselves: a[i] = (R p)[src] at this thread's positions IPT tid + i (0 where the
// position carries no checked row), pk[i] = the positions' packed indices (row = (pk & 0xFFFF) - 1).
// p_clear != NULL (the same array as p, writable): every node sum read is set to zero behind the
// read -- the block verdicts leave the ring slice they judged ready for the next accumulation.
// `lds`: tree_lds_bytes() bytes, 16-byte aligned.  Ends without a barrier behind its last LDS reads.
// PK_LOADED: the caller has fetched pk[] already (it gathers other columns by the same rows).
// Two things shape this body.  Registers: in the 256 x 8 shape it runs inside the residence sweep's
// kernel, whose occupancy (8 wavefronts per SIMD, 64 VGPRs) it must not lower -- two IPT-double
// vectors per thread beside the static data.  Latency: its workgroups are a launch's critical
// path when memory is saturated by a sweep (every dependent global load costs 2-3 us there), so
// ALL the static data of a thread -- four 16-bit indices per position packed in one 64-bit word,
// and the weights -- are requested at the very top, and the only dependent global access is the
// gather of the node sums behind them.
// The three pieces of tree_voltage, for callers that order their own loads around them:
// the weights (static), the gather of the node sums (waits for the packed indices), the scans.
template <int NT, int IPT>
__device__ __forceinline__ void tree_fetch_w(const TreeArgs &tr, double (&b)[IPT]) {
    const int j0 = IPT * threadIdx.x;
#pragma unroll
    for (int i = 0; i < IPT; ++i) b[i] = 0.0;
    if (j0 < tr.n) {
#pragma unroll
        for (int i = 0; i < IPT; i += 2) {
            const TreeD2 wv = *reinterpret_cast<const TreeD2 *>(tr.w + j0 + i);
            b[i] = wv.v[0]; b[i + 1] = wv.v[1];
        }
    }
}
// STRAIGHT: positions without a row fetch row 0 -- straight-line loads, nothing waits or branches here --
// and are masked where the values are first used (tree_scan<.., true>).  For the latency-bound operator
// launches; inside the sweep's kernel the eight values in flight at once would cost registers (spills at
// its 64-register cap).
template <int NT, int IPT, bool STRAIGHT = false>
__device__ __forceinline__ void tree_gather_p(const TreeArgs &tr, const double *p, int T, int t,
                                              const unsigned long long (&pk)[IPT], double (&a)[IPT], double *p_clear) {
    const int j0 = IPT * threadIdx.x;
#pragma unroll
    for (int i = 0; i < IPT; ++i) a[i] = 0.0;
    if (j0 < tr.n) {
        // C: inclusive prefix of the injections in preorder
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int s = (int)(pk[i] & 0xFFFFu) - 1;
            if constexpr (STRAIGHT) a[i] = p[(int64_t)(s >= 0 ? s : 0) * T + t];
            else a[i] = s >= 0 ? p[(int64_t)s * T + t] : 0.0;
        }
        if (p_clear) {
#pragma unroll
            for (int i = 0; i < IPT; ++i) {
                const int s = (int)(pk[i] & 0xFFFFu) - 1;
                if (s >= 0) p_clear[(int64_t)s * T + t] = 0.0;
            }
        }
    }
}
// a: the gathered node sums, b: the weights (both zero beyond the tree); out: a = the voltages
template <int NT, int IPT, bool MASK = false>
__device__ __forceinline__ void tree_scan(const TreeArgs &tr, int t, double *lds, double (&a)[IPT], double (&b)[IPT],
                                          const unsigned long long (&pk)[IPT]) {
    const int tid = threadIdx.x, n = tr.n, j0 = IPT * tid;
    const bool act = j0 < n;
    double *base = lds + 2, *red0 = lds + 2 + NT * IPT, *red1 = red0 + NT / 64;
    if (tid == 0) lds[1] = 0.0;                                 // base[-1]
    if constexpr (MASK) {
#pragma unroll
        for (int i = 0; i < IPT; ++i) a[i] = (act && (pk[i] & 0xFFFFu) != 0ull) ? a[i] : 0.0;
    }
#pragma unroll
    for (int i = 1; i < IPT; ++i) a[i] += a[i - 1];
    REVS_KVS(t, 2);
    const double cex = block_excl_offset<NT>(a[IPT - 1], red0);   // C_excl at j0
    REVS_KVS(t, 3);
    if (act) {
#pragma unroll
        for (int i = 0; i < IPT; i += 2)
            *reinterpret_cast<TreeD2 *>(base + j0 + i) = TreeD2{{a[i] + cex, a[i + 1] + cex}};
    }
    __syncthreads();
    // w'_j = w_j (C[end_j] - C[j]),  C[j] = base[j - 1] (own positions: registers)
    if (act) {
#pragma unroll
        for (int i = IPT - 1; i >= 1; --i)
            a[i] = b[i] * (base[(int)((pk[i] >> 16) & 0xFFFFu) - 1] - (a[i - 1] + cex));
        a[0] = b[0] * (base[(int)((pk[0] >> 16) & 0xFFFFu) - 1] - cex);
    }
    __syncthreads();                                            // every read of C is done
    if (act) {
#pragma unroll
        for (int i = 0; i < IPT; i += 2)
            *reinterpret_cast<TreeD2 *>(base + j0 + i) = TreeD2{{a[i], a[i + 1]}};
    }
    __syncthreads();
    REVS_KVS(t, 4);
    // the same values in end-order, then both prefixes: Pre over preorder (a), F over end-order (b)
#pragma unroll
    for (int i = 0; i < IPT; ++i) b[i] = act ? base[(int)((pk[i] >> 32) & 0xFFFFu)] : 0.0;
#pragma unroll
    for (int i = 1; i < IPT; ++i) { a[i] += a[i - 1]; b[i] += b[i - 1]; }
    const double pex = block_excl_offset<NT>(a[IPT - 1], red1);   // (its barrier: every read of w' is done)
    const double fex = block_excl_offset<NT>(b[IPT - 1], red0);
    if (act) {
#pragma unroll
        for (int i = 0; i < IPT; i += 2)
            *reinterpret_cast<TreeD2 *>(base + j0 + i) = TreeD2{{b[i] + fex, b[i + 1] + fex}};
    }
    __syncthreads();
    REVS_KVS(t, 5);
    // v_j = Pre[j] - F_excl[cle[j]] on the checked rows,  F_excl[c] = base[c - 1]
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const int s = (int)(pk[i] & 0xFFFFu) - 1;
        a[i] = (act && s >= 0) ? (a[i] + pex) - base[(int)(pk[i] >> 48) - 1] : 0.0;
    }
}

template <int NT, int IPT, bool PK_LOADED = false, bool STRAIGHT = false>
__device__ __forceinline__ void tree_voltage(const TreeArgs &tr, const double *p, int T, int t,
                                             double *lds, double (&a)[IPT], unsigned long long (&pk)[IPT],
                                             double *p_clear) {
    const int j0 = IPT * threadIdx.x;
    if (!PK_LOADED) {
#pragma unroll
        for (int i = 0; i < IPT; ++i) pk[i] = 0ull;
        if (j0 < tr.n) {
#pragma unroll
            for (int i = 0; i < IPT; i += 2) {
                const TreeU2 u = *reinterpret_cast<const TreeU2 *>(tr.pack + j0 + i);
                pk[i] = u.v[0]; pk[i + 1] = u.v[1];
            }
        }
    }
    double b[IPT];                                              // (b holds the weights until phase 2)
    tree_fetch_w<NT, IPT>(tr, b);
    tree_gather_p<NT, IPT, STRAIGHT>(tr, p, T, t, pk, a, p_clear);
    tree_scan<NT, IPT, STRAIGHT>(tr, t, lds, a, b, pk);
}

// Largest violation max(v - vhi, vlo - v, 0) over the checked rows of slot t (every thread
// gets it); v_out[src][t] = v when v_out != NULL.
// PRE: the caller has fetched pk[] and the weights wgt[] already (tree_fetch_w), e.g. in front of a test
// it has to wait for anyway.
template <int NT = 256, int IPT = 8, bool PRE = false>
__device__ __forceinline__ double tree_rmax(const TreeArgs &tr, const double *p, int T, int t,
                                            double vlo, double vhi, double *lds, double *v_out,
                                            double *p_clear = nullptr, unsigned long long *pk_pre = nullptr,
                                            double *wgt_pre = nullptr) {
    const int tid = threadIdx.x;
    double *red1 = lds + 2 + NT * IPT + NT / 64;
    unsigned long long pk[IPT];
    double a[IPT];
    if constexpr (PRE) {
        double b[IPT];
#pragma unroll
        for (int i = 0; i < IPT; ++i) { pk[i] = pk_pre[i]; b[i] = wgt_pre[i]; }
        tree_gather_p<NT, IPT, true>(tr, p, T, t, pk, a, nullptr);
        tree_scan<NT, IPT, true>(tr, t, lds, a, b, pk);
    } else {
        tree_voltage<NT, IPT>(tr, p, T, t, lds, a, pk, p_clear);
    }
    double rmax = 0.0;
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const int s = (int)(pk[i] & 0xFFFFu) - 1;
        if (s >= 0) {
            const double v = a[i];
            rmax = fmax(rmax, fmax(fmax(v - vhi, vlo - v), 0.0));
            if (v_out) v_out[(int64_t)s * T + t] = v;
        }
    }
    rmax = wave_max_d(rmax);
    __syncthreads();                                            // (red1 was the scans': every read is done)
    if ((tid & 63) == 0) red1[tid >> 6] = rmax;
    __syncthreads();
    double r = red1[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) r = fmax(r, red1[w]);
    if constexpr (PRE) {
        if (p_clear && IPT * tid < tr.n) {      // (behind everything: the column's stores are off the verdict's path)
#pragma unroll
            for (int i = 0; i < IPT; ++i) {
                const int s = (int)(pk[i] & 0xFFFFu) - 1;
                if (s >= 0) p_clear[(int64_t)s * T + t] = 0.0;
            }
        }
    }
    return r;
}

// tree_rmax for the slots t and t + 1 (t even, T even, p 16-byte aligned) of one array in one workgroup: ONE 16-byte
// request per position fetches both slots' node sums, and one clears them -- the verdict launches are bound by the
// rate of scattered lane requests (2048 gathers and 2048 clearing stores of 8 bytes per slot: ~2.7 cycles each through
// a compute unit's texture path; 800 workgroups of a block of 32 took 42 us), not by the scans.  The scans of the
// two slots run one after the other on the same LDS.  Returns max(rmax_t, rmax_t+1).
template <int NT, int IPT>
__device__ __forceinline__ double tree_rmax_pair(const TreeArgs &tr, const double *p, int T, int t, double vlo, double vhi,
                                                 double *lds, double *p_clear, const unsigned long long *pk_pre,
                                                 const double *wgt_pre) {
    const int tid = threadIdx.x;
    const bool act = IPT * tid < tr.n;
    double *red1 = lds + 2 + NT * IPT + NT / 64;
    unsigned long long pk[IPT];
    double a0[IPT], a1[IPT], b[IPT];
#pragma unroll
    for (int i = 0; i < IPT; ++i) { pk[i] = pk_pre[i]; a0[i] = 0.0; a1[i] = 0.0; }
    if (act) {
#pragma unroll
        for (int i = 0; i < IPT; ++i) {       // (positions without a row fetch row 0: masked by the scans)
            const int s = (int)(pk[i] & 0xFFFFu) - 1;
            const TreeD2 v = *reinterpret_cast<const TreeD2 *>(p + (int64_t)(s >= 0 ? s : 0) * T + t);
            a0[i] = v.v[0]; a1[i] = v.v[1];
        }
    }
    double rmax = 0.0;
#pragma unroll
    for (int i = 0; i < IPT; ++i) b[i] = wgt_pre[i];
    tree_scan<NT, IPT, true>(tr, t, lds, a0, b, pk);
    VD_STAMP(6);
#pragma unroll
    for (int i = 0; i < IPT; ++i)
        if ((pk[i] & 0xFFFFu) != 0ull) rmax = fmax(rmax, fmax(fmax(a0[i] - vhi, vlo - a0[i]), 0.0));
    __syncthreads();                                            // (the first slot's last LDS reads are done)
#pragma unroll
    for (int i = 0; i < IPT; ++i) b[i] = wgt_pre[i];
    tree_scan<NT, IPT, true>(tr, t + 1, lds, a1, b, pk);
    VD_STAMP(7);
#pragma unroll
    for (int i = 0; i < IPT; ++i)
        if ((pk[i] & 0xFFFFu) != 0ull) rmax = fmax(rmax, fmax(fmax(a1[i] - vhi, vlo - a1[i]), 0.0));
    rmax = wave_max_d(rmax);
    __syncthreads();                                            // (red1 was the scans': every read is done)
    if ((tid & 63) == 0) red1[tid >> 6] = rmax;
    __syncthreads();
    double r = red1[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) r = fmax(r, red1[w]);
    if (p_clear && act) {                       // (behind everything: the stores are off the verdict's path)
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int s = (int)(pk[i] & 0xFFFFu) - 1;
            if (s >= 0) *reinterpret_cast<TreeD2 *>(p_clear + (int64_t)s * T + t) = TreeD2{{0.0, 0.0}};
        }
    }
    return r;
}

// Control block of the streaming steady state (device memory, owned by the plan).
struct StreamCtl {
    unsigned int bad_seq;                 // sequence number of the last launch whose verdict failed (0: none yet)
    unsigned int arrive;                  // tree workgroups of the current launch that are done
    unsigned long long rmax_bits;         // max over their slots (bit pattern of a double >= 0)
};
constexpr int kRecRing = 1024;             // records double[kRecRing][4] in pinned host memory

}  // namespace revs
