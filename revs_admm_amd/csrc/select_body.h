// Candidate selection of the operator's dual Newton path as a device function: it runs as
// its own kernel (op_dual_select_kernel) or as the first T workgroups of the home sweep's
// launch (agent_step_kernel with AgentArgs::nsel > 0), where it overlaps the sweep.
#pragma once
#include "common.h"

namespace revs {

constexpr int kAmax = REVS_DUAL_AMAX;
static_assert(kAmax == 128, "candidate sets are two 64-bit words / two wavefronts");
constexpr int kWords = kAmax / 64;

// A slot's candidate list in LDS (SelectArgs::ll): what the selection also writes to cidx / ccnt / cval,
// for a model step that follows in the same workgroup (the folded chain) without a trip through memory.
struct SlotLists {
    long long ci[kAmax];
    double cs[kAmax], cg[kAmax], cy[kAmax];
    int cnt;
};

struct SelectArgs {
    int m, T, nblk, kadd;
    const double *partial, *y, *vfull, *viol;
    double vlo, vhi, seq;
    int64_t *cidx;
    int32_t *ccnt;
    double *cval, *stats;
    // lazy: nobody polls this evaluation's tag (the folded chain's speculative next iteration: the
    // host reads its stats only after a later launch's verdict) -- the stats are stored without the
    // system-scope fence in front of the tag, which otherwise holds thread 0 (and at the next
    // barrier the whole workgroup) for a round trip to host memory
    bool lazy = false;
    // fwd_src != NULL: before its own tag this selection copies another evaluation's stats {rmax, D, multipliers,
    // violated rows} of the same slot from device memory (fwd_src[8 t ..]) into the host-visible block fwd_dst --
    // the folded chain's next-iteration half keeps its stats on the device (a store to pinned host memory in front
    // of a volatile tag store holds thread 0, and at the next barrier the workgroup, for a round trip over the bus:
    // ~1.5 us of the launch's critical path), and the verdict half of the NEXT launch hands them to the host
    const double *fwd_src = nullptr;
    double *fwd_dst = nullptr;
    // rows_lds != NULL: the rows' multipliers, voltages and violations of THIS slot staged in LDS by
    // the workgroup's own row kernel (double[3][m], by row: y | v | violation) and its four sums in
    // rows_lds[3 m ..]: read instead of the global columns (a column of a [m][T] array is m cache
    // lines: each column read or write costs a latency-bound workgroup ~2 us)
    const double *rows_lds = nullptr;
    SlotLists *ll = nullptr;
};

// Stage 2, one workgroup per slot: fold the partials (fixed order), and -- only for a slot
// that has multipliers or violated rows -- build its candidate list.  Thread j owns the
// contiguous rows [j R, (j+1) R), R = ceil(m/256), so the rows with a multiplier are
// compacted in row order (deterministic: every rank builds the same lists); the most
// violated rows without a multiplier are appended by `kadd` rounds of a block-wide
// arg-max (ties to the lower row).
// EAGER (the stand-alone kernel, which is a chain of memory round trips otherwise): every load
// whose address does not depend on data -- partials, this thread's multipliers, voltages and
// violations -- is issued at the top.  Not inside the sweep's launch: the registers would set
// the sweep's occupancy, and a slot with nothing to do (its usual case there) needs none.
template <bool EAGER = false>
__device__ __forceinline__ double dual_select_body(const SelectArgs &sa, const int t) {
    const int m = sa.m, T = sa.T, nblk = sa.nblk, kadd = sa.kadd;
    const double *__restrict__ partial = sa.partial, *__restrict__ y = sa.y;
    const double *__restrict__ vfull = sa.vfull, *__restrict__ viol = sa.viol;
    const double vlo = sa.vlo, vhi = sa.vhi, seq = sa.seq;
    const double *const rl = sa.rows_lds;
    auto y_at = [&](int r) { return rl ? rl[r] : y[(int64_t)r * T + t]; };
    auto v_at = [&](int r) { return rl ? rl[m + r] : vfull[(int64_t)r * T + t]; };
    auto viol_at = [&](int r) { return rl ? rl[2 * m + r] : viol[(int64_t)r * T + t]; };
    int64_t *__restrict__ cidx = sa.cidx;
    int32_t *__restrict__ ccnt = sa.ccnt;
    double *__restrict__ cval = sa.cval, *__restrict__ stats = sa.stats;
    const int tid = threadIdx.x;
    // inside the sweep's launch these few wavefronts share their SIMDs with the sweep's, which
    // keep the vector pipe busy, and the host is waiting for their verdict: raised priority
    if (!EAGER) __builtin_amdgcn_s_setprio(3);
    __shared__ int cnt_s[4];
    __shared__ int vcount;
    if (EAGER && tid == 0) vcount = 0;     // (ordered by the first barrier below)
    __shared__ double red_s[4][4];
    __shared__ double best_v[4];
    __shared__ int best_i[4];
    const int per = (m + 255) / 256;
    const int r0 = min(m, tid * per), r1 = min(m, r0 + per);
    constexpr int kLoc = 16, kPer = 8;
    const bool inreg = m <= 256 * kLoc;
    const bool eager = EAGER && inreg && per <= kPer;
    double loc[kLoc], ysv[kPer], vsv[kPer];
    {
        double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
        for (int k = tid; k < nblk; k += 256) {
            const double *o = rl ? rl + 3 * m : partial + ((int64_t)k * T + t) * 4;
            a = fmax(a, o[0]); b += o[1]; c += o[2]; d += o[3];
        }
        if (eager) {
#pragma unroll
            for (int j = 0; j < kPer; ++j) {
                const int r = r0 + j;
                ysv[j] = r < r1 ? y_at(r) : 0.0;
                vsv[j] = r < r1 ? v_at(r) : 0.0;
            }
#pragma unroll
            for (int i = 0; i < kLoc; ++i) {
                const int r = tid + 256 * i;
                loc[i] = r < m ? viol_at(r) : 0.0;
            }
        }
        a = wave_max_d(a); b = wave_sum_d(b); c = wave_sum_d(c); d = wave_sum_d(d);
        if ((tid & 63) == 0) {
            red_s[0][tid >> 6] = a; red_s[1][tid >> 6] = b; red_s[2][tid >> 6] = c; red_s[3][tid >> 6] = d;
        }
    }
    __syncthreads();
    REVS_KVS(t, 8);
    const int ns = (int)(((red_s[2][0] + red_s[2][1]) + red_s[2][2]) + red_s[2][3]);
    const int nv = (int)(((red_s[3][0] + red_s[3][1]) + red_s[3][2]) + red_s[3][3]);
    const double rmax_t = fmax(fmax(red_s[0][0], red_s[0][1]), fmax(red_s[0][2], red_s[0][3]));
    if (tid == 0) {
        stats[t * 8 + 0] = rmax_t;
        stats[t * 8 + 1] = ((red_s[1][0] + red_s[1][1]) + red_s[1][2]) + red_s[1][3];
        stats[t * 8 + 2] = (double)ns;
        stats[t * 8 + 3] = (double)nv;
        if (sa.fwd_src) {       // (all four read before the first is stored: the arrays may alias as far as the compiler knows)
            double f[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) f[i] = sa.fwd_src[t * 8 + i];
#pragma unroll
            for (int i = 0; i < 4; ++i) sa.fwd_dst[t * 8 + i] = f[i];
        }
        // stats may live in pinned host memory: a host that polls [5] for this evaluation's
        // sequence number sees [0..3] complete (system-scope release before the tag)
        if (!sa.lazy) __threadfence_system();
        reinterpret_cast<volatile double *>(stats)[t * 8 + 5] = seq;
    }
    int64_t *ci = cidx + (int64_t)t * kAmax;
    double *cs = cval + (int64_t)t * 3 * kAmax, *cg = cs + kAmax, *cy = cg + kAmax;
    SlotLists *const ll = sa.ll;
    if (ns > kAmax || (ns == 0 && nv == 0)) {   // uniform: too many multipliers / nothing to do
        if (tid == 0) { ccnt[t] = ns > kAmax ? -1 : 0; if (ll) ll->cnt = ns > kAmax ? -1 : 0; }
        if (tid < kAmax) { ci[tid] = 0; cs[tid] = 1.0; cg[tid] = 0.0; cy[tid] = 0.0; }
        return rmax_t;
    }
    int nsup = 0;
    if (eager) {
#pragma unroll
        for (int j = 0; j < kPer; ++j) nsup += ysv[j] != 0.0 ? 1 : 0;
    } else {
        for (int r = r0; r < r1; ++r) nsup += y_at(r) != 0.0 ? 1 : 0;
    }
    // exclusive prefix over the workgroup: scan inside the wavefront, then the wavefronts' totals
    const int incl = wave_incl_scan_i(nsup);
    if ((tid & 63) == 63) cnt_s[tid >> 6] = incl;
    __syncthreads();
    int pos = incl - nsup;
    for (int w = 0; w < (tid >> 6); ++w) pos += cnt_s[w];
    if (eager) {
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const double yv = ysv[j];
            if (yv != 0.0) {
                const double sg = yv > 0.0 ? 1.0 : -1.0, gr = vsv[j] - (yv > 0.0 ? vhi : vlo);
                ci[pos] = r0 + j;
                cs[pos] = sg;
                cg[pos] = gr;
                cy[pos] = yv;
                if (ll) { ll->ci[pos] = r0 + j; ll->cs[pos] = sg; ll->cg[pos] = gr; ll->cy[pos] = yv; }
                ++pos;
            }
        }
    } else {
        for (int r = r0; r < r1 && nsup > 0; ++r) {
            const double yv = y_at(r);
            if (yv != 0.0) {
                const double sg = yv > 0.0 ? 1.0 : -1.0, gr = v_at(r) - (yv > 0.0 ? vhi : vlo);
                ci[pos] = r;
                cs[pos] = sg;
                cg[pos] = gr;
                cy[pos] = yv;
                if (ll) { ll->ci[pos] = r; ll->cs[pos] = sg; ll->cg[pos] = gr; ll->cy[pos] = yv; }
                ++pos;
            }
        }
    }
    const int room = min(min(kadd, kAmax - ns), nv);
    int added = 0;
    REVS_KVS(t, 9);
    // Row r is always scanned by thread r % 256.  Up to 4096 rows a thread keeps its (at most
    // 16) violations in registers and zeroes the one that is taken; beyond that it re-reads
    // them and remembers the taken ones in a register mask -- either way no global store has
    // to become visible between rounds.  One barrier per round (results ping-pong in LDS).
    if (inreg && !eager) {
#pragma unroll
        for (int i = 0; i < kLoc; ++i) {
            const int r = tid + 256 * i;
            loc[i] = r < m ? viol_at(r) : 0.0;
        }
    }
    unsigned long long took = 0ull;
    __shared__ double best_v2[2][4];
    __shared__ int best_i2[2][4];
    __shared__ int chosen[kAmax];
    constexpr int kRank = 256;
    __shared__ double vval[EAGER ? kRank : 1];
    __shared__ int vrow[EAGER ? kRank : 1];
    const bool collected = EAGER && room > 0 && nv <= kRank;
    if (collected) {
        // Few violated rows (nv of them carry a positive entry): collected (any order) one per thread, so
        // that the `room` rounds of the block-wide arg-max below compare registers instead of scanning
        // every thread's share of the rows -- larger violation first, ties to the lower row.
        auto put = [&](double x, int r) {
            if (x > 0.0) {
                const int q = atomicAdd(&vcount, 1);
                if (q < kRank) { vval[q] = x; vrow[q] = r; }
            }
        };
        if (inreg) {
#pragma unroll
            for (int i = 0; i < kLoc; ++i) put(loc[i], tid + 256 * i);
        } else {
            for (int r = tid; r < m; r += 256) put(viol_at(r), r);
        }
        __syncthreads();
        const int nq = min(vcount, kRank);
        double x = tid < nq ? vval[tid] : 0.0;
        const int r = tid < nq ? vrow[tid] : 0x7FFFFFFF;
        for (int k = 0; k < min(room, nq); ++k) {      // (one candidate per thread: the rounds scan nothing)
            const double wv = wave_max_d(x);
            const int wr = wave_min_i(x == wv ? r : 0x7FFFFFFF);
            const int pp = k & 1;
            if ((tid & 63) == 0) { best_v2[pp][tid >> 6] = wv; best_i2[pp][tid >> 6] = wr; }
            __syncthreads();
            double bv = best_v2[pp][0];
            int br = best_i2[pp][0];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const double ov = best_v2[pp][w];
                const int orow = best_i2[pp][w];
                const bool take = ov > bv || (ov == bv && orow < br);
                bv = take ? ov : bv;
                br = take ? orow : br;
            }
            if (tid < nq && r == br) { x = 0.0; chosen[k] = r; }
        }
        added = min(room, nq);
    }
    for (int k = 0; k < (collected ? 0 : room); ++k) {
        double bv = 0.0;
        int bi = m;
        if (inreg) {
#pragma unroll
            for (int i = 0; i < kLoc; ++i)
                if (loc[i] > bv) { bv = loc[i]; bi = tid + 256 * i; }   // ascending r: ties keep the lower row
        } else {
            for (int r = tid, i = 0; r < m; r += 256, ++i) {
                const double x = viol_at(r);
                if (x > bv && !((took >> i) & 1ull)) { bv = x; bi = r; }
            }
        }
        {   // the wavefront's largest violation, ties to the lower row
            const double wv = wave_max_d(bv);
            bi = wave_min_i(bv == wv ? bi : 0x7FFFFFFF);
            bv = wv;
        }
        const int pp = k & 1;
        if ((tid & 63) == 0) { best_v2[pp][tid >> 6] = bv; best_i2[pp][tid >> 6] = bi; }
        __syncthreads();
        bv = best_v2[pp][0]; bi = best_i2[pp][0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (best_v2[pp][w] > bv || (best_v2[pp][w] == bv && best_i2[pp][w] < bi)) {
                bv = best_v2[pp][w]; bi = best_i2[pp][w];
            }
        if (!(bv > 0.0)) break;             // uniform
        if (tid == (bi & 255)) {
            if (inreg) {
#pragma unroll
                for (int i = 0; i < kLoc; ++i) if (i == (bi >> 8)) loc[i] = 0.0;
            } else {
                took |= 1ull << (bi >> 8);
            }
        }
        if (tid == 0) chosen[k] = bi;         // (visible after the next barrier)
        ++added;
    }
    // the chosen rows' entries, one thread each: a global load inside the round loop would
    // cost every round a memory latency (the other wavefronts wait at the barrier)
    __syncthreads();
    REVS_KVS(t, 10);
    if (tid < added) {
        const int bi = chosen[tid];
        const double v = v_at(bi);
        const bool up = v > vhi;
        ci[ns + tid] = bi;
        cs[ns + tid] = up ? 1.0 : -1.0;
        cg[ns + tid] = v - (up ? vhi : vlo);
        cy[ns + tid] = 0.0;
        if (ll) { ll->ci[ns + tid] = bi; ll->cs[ns + tid] = up ? 1.0 : -1.0; ll->cg[ns + tid] = v - (up ? vhi : vlo); ll->cy[ns + tid] = 0.0; }
    }
    const int cnt = ns + added;
    if (tid == 0) { ccnt[t] = cnt; if (ll) ll->cnt = cnt; }
    if (tid >= cnt && tid < kAmax) {
        ci[tid] = 0; cs[tid] = 1.0; cg[tid] = 0.0; cy[tid] = 0.0;
        if (ll) { ll->ci[tid] = 0; ll->cs[tid] = 1.0; ll->cg[tid] = 0.0; ll->cy[tid] = 0.0; }
    }
    return rmax_t;
}

}  // namespace revs
