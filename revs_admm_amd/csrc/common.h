// Shared helpers for librevs_admm.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "revs_admm_ops.h"      // (includes revs_admm.h, the boundary)

namespace revs {

void set_error(const char *fmt, ...);

#define REVS_REQUIRE(cond, ...)                                  \
    do {                                                         \
        if (!(cond)) {                                           \
            revs::set_error(__VA_ARGS__);                        \
            return REVS_EINVAL;                                  \
        }                                                        \
    } while (0)

#define REVS_CHECK_LAUNCH(what)                                                  \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            revs::set_error("%s: %s", what, hipGetErrorString(e__));             \
            return REVS_ELAUNCH;                                                 \
        }                                                                        \
    } while (0)

// ---- DPP cross-lane moves (64-wide wavefront, rows of 16 lanes) ----------
// dpp_ctrl encodings: quad_perm 0x00-0xFF, row_shl:n 0x100+n, row_shr:n 0x110+n,
// row_mirror 0x140, row_half_mirror 0x141.  bound_ctrl=1: lanes whose source is
// outside the row read 0.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(
        __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}

// ---- reductions over the 64 lanes of a wavefront without the LDS permute network -------------
// __shfl_xor of a double is two ds_bpermute_b32 (a round trip through the LDS crossbar each, ~40
// cycles) per step and the six steps depend on each other: ~260 cycles per reduction, which is what
// a latency-bound workgroup of the operator's launches spends its time on.  Here: v_permlane32_swap /
// v_permlane16_swap (gfx950) for the two steps across rows, DPP row rotations inside the rows --
// ~70 cycles, and the SAME association as the xor butterfly 32, 16, 8, 4, 2, 1 (after the step at
// distance 2d a lane's value depends on its index mod 2d only, so a rotation by d inside the row of
// 16 delivers exactly the butterfly partner's value): the sums keep their bits
// (tools/probes/wave_reduce.hip).
template <int CTRL>
__device__ __forceinline__ double dpp_rot_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// (a, b) = (the value of this lane's partner half, own half) in some order: lanes i and i ^ 32 (SWAP32)
// or i ^ 16 both get the pair {v_i, v_partner}
template <bool SWAP32>
__device__ __forceinline__ void swap_pair_d(double v, double &a0, double &a1) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    if constexpr (SWAP32) {
        const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        a0 = __longlong_as_double(((long long)h[0] << 32) | l[0]);
        a1 = __longlong_as_double(((long long)h[1] << 32) | l[1]);
    } else {
        const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        a0 = __longlong_as_double(((long long)h[0] << 32) | l[0]);
        a1 = __longlong_as_double(((long long)h[1] << 32) | l[1]);
    }
}
__device__ __forceinline__ double wave_sum_d(double v) {
    double a0, a1;
    swap_pair_d<true>(v, a0, a1);  v = a0 + a1;
    swap_pair_d<false>(v, a0, a1); v = a0 + a1;
    v += dpp_rot_d<0x128>(v);      // row_ror:8
    v += dpp_rot_d<0x124>(v);
    v += dpp_rot_d<0x122>(v);
    v += dpp_rot_d<0x121>(v);
    return v;
}
// Sum over the 64 / LPA aligned lane groups of a wavefront, lane by lane (lane i of every group gets the sum of
// lane i of all groups): the butterfly's steps at distance >= LPA only.
template <int LPA>
__device__ __forceinline__ double wave_sum_over_groups_d(double v) {
    double a0, a1;
    if constexpr (LPA <= 32) { swap_pair_d<true>(v, a0, a1);  v = a0 + a1; }
    if constexpr (LPA <= 16) { swap_pair_d<false>(v, a0, a1); v = a0 + a1; }
    if constexpr (LPA <= 8) v += dpp_rot_d<0x128>(v);      // row_ror:8
    if constexpr (LPA <= 4) v += dpp_rot_d<0x124>(v);
    if constexpr (LPA <= 2) v += dpp_rot_d<0x122>(v);
    if constexpr (LPA <= 1) v += dpp_rot_d<0x121>(v);
    return v;
}
// N independent sums at once, level by level: the chains of different values interleave (one after the
// other each is ~18 dependent instructions with nothing to fill the gaps).  Same association as wave_sum_d.
template <int N>
__device__ __forceinline__ void wave_sum_multi_d(double (&v)[N]) {
    double a0, a1;
#pragma unroll
    for (int e = 0; e < N; ++e) { swap_pair_d<true>(v[e], a0, a1); v[e] = a0 + a1; }
#pragma unroll
    for (int e = 0; e < N; ++e) { swap_pair_d<false>(v[e], a0, a1); v[e] = a0 + a1; }
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] += dpp_rot_d<0x128>(v[e]);
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] += dpp_rot_d<0x124>(v[e]);
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] += dpp_rot_d<0x122>(v[e]);
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] += dpp_rot_d<0x121>(v[e]);
}
__device__ __forceinline__ double wave_max_d(double v) {
    double a0, a1;
    swap_pair_d<true>(v, a0, a1);  v = fmax(a0, a1);
    swap_pair_d<false>(v, a0, a1); v = fmax(a0, a1);
    v = fmax(v, dpp_rot_d<0x128>(v));
    v = fmax(v, dpp_rot_d<0x124>(v));
    v = fmax(v, dpp_rot_d<0x122>(v));
    v = fmax(v, dpp_rot_d<0x121>(v));
    return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
    const auto a = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    v = min((int)a[0], (int)a[1]);
    const auto b = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    v = min((int)b[0], (int)b[1]);
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, true));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, true));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, true));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, true));
    return v;
}
// inclusive prefix sum of an int over the wavefront (row_shr steps, then the row totals across)
__device__ __forceinline__ int wave_incl_scan_i(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

// Sum / max over an aligned group of LPA lanes, result in every lane of the group.
// (the steps across rows of 16 lanes: v_permlane16_swap / v_permlane32_swap (gfx950) hand lanes i and i ^ 16 (i ^ 32) the
// pair {own, partner} -- one VALU instruction where __shfl_xor is a trip through the LDS crossbar)
__device__ __forceinline__ void swap_pair_f(float v, bool swap32, float &a0, float &a1) {
    const unsigned b = __float_as_uint(v);
    if (swap32) {
        const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
        a0 = __uint_as_float(r[0]); a1 = __uint_as_float(r[1]);
    } else {
        const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
        a0 = __uint_as_float(r[0]); a1 = __uint_as_float(r[1]);
    }
}
template <int LPA>
__device__ __forceinline__ float group_sum(float v) {
    float a0, a1;
    if constexpr (LPA >= 2) v += dpp_f<0xB1>(v);            // quad_perm [1,0,3,2]
    if constexpr (LPA >= 4) v += dpp_f<0x4E>(v);            // quad_perm [2,3,0,1]
    if constexpr (LPA >= 8) v += dpp_f<0x141>(v);           // row_half_mirror
    if constexpr (LPA >= 16) v += dpp_f<0x140>(v);          // row_mirror
    if constexpr (LPA >= 32) { swap_pair_f(v, false, a0, a1); v = a0 + a1; }
    if constexpr (LPA >= 64) { swap_pair_f(v, true, a0, a1); v = a0 + a1; }
    return v;
}
template <int LPA>
__device__ __forceinline__ float group_max(float v) {
    float a0, a1;
    if constexpr (LPA >= 2) v = fmaxf(v, dpp_f<0xB1>(v));
    if constexpr (LPA >= 4) v = fmaxf(v, dpp_f<0x4E>(v));
    if constexpr (LPA >= 8) v = fmaxf(v, dpp_f<0x141>(v));
    if constexpr (LPA >= 16) v = fmaxf(v, dpp_f<0x140>(v));
    if constexpr (LPA >= 32) { swap_pair_f(v, false, a0, a1); v = fmaxf(a0, a1); }
    if constexpr (LPA >= 64) { swap_pair_f(v, true, a0, a1); v = fmaxf(a0, a1); }
    return v;
}

// max over a group of NON-NEGATIVE floats (their bit patterns order as unsigned integers): v_max_u32 with the DPP
// move folded in -- fmaxf of a DPP-moved value costs a canonicalising v_max_f32 x, x on top of the move and the max
template <int LPA>
__device__ __forceinline__ float group_max_nonneg(float v) {
    unsigned u = __float_as_uint(v);
    if constexpr (LPA >= 2) u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0xB1, 0xf, 0xf, true));
    if constexpr (LPA >= 4) u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x4E, 0xf, 0xf, true));
    if constexpr (LPA >= 8) u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x141, 0xf, 0xf, true));
    if constexpr (LPA >= 16) u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x140, 0xf, 0xf, true));
    if constexpr (LPA >= 32) { const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false); u = max((unsigned)r[0], (unsigned)r[1]); }
    if constexpr (LPA >= 64) { const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false); u = max((unsigned)r[0], (unsigned)r[1]); }
    return __uint_as_float(u);
}
// max of a non-negative float over the whole wavefront, in EVERY lane, when the LPA lanes of each aligned group
// already agree: DPP rotations inside the rows of 16, then v_permlane16_swap / v_permlane32_swap (gfx950) across
template <int LPA>
__device__ __forceinline__ unsigned wave_max_groups_u(unsigned u) {
    if constexpr (LPA < 2) u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x121, 0xf, 0xf, true));   // row_ror:1
    if constexpr (LPA < 4) u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x122, 0xf, 0xf, true));
    if constexpr (LPA < 8) u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x124, 0xf, 0xf, true));
    if constexpr (LPA < 16) u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x128, 0xf, 0xf, true));
    if constexpr (LPA < 32) {
        const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        u = max((unsigned)a[0], (unsigned)a[1]);
    }
    if constexpr (LPA < 64) {
        const auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        u = max((unsigned)b[0], (unsigned)b[1]);
    }
    return u;
}

// Exclusive prefix / suffix sums over the lanes of a group (lig = lane index inside
// the group).  Each Hillis-Steele step is written as s = fma(dpp(s), mask, s) with a
// 0/1 lane mask so that the compiler can fold the DPP move into one v_fmac_f32_dpp;
// the masks keep a row_shr/row_shl from pulling in the neighbouring group's lanes.
template <int LPA>
struct ScanMasks {
    float up[4];    // lig >= 1, 2, 4, 8
    float dn[4];    // lig + d < LPA for d = 1, 2, 4, 8
    __device__ __forceinline__ explicit ScanMasks(int lig) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            up[k] = (lig >= (1 << k)) ? 1.0f : 0.0f;
            dn[k] = (lig + (1 << k) < LPA) ? 1.0f : 0.0f;
        }
    }
};

template <int LPA>
__device__ __forceinline__ float group_excl_prefix(float v, int lig, const ScanMasks<LPA> &mk) {
    float s;
    if constexpr (LPA <= 16) {
        s = dpp_f<0x111>(v) * mk.up[0];                      // row_shr:1
        if constexpr (LPA >= 4) s = fmaf(dpp_f<0x111>(s), mk.up[0], s);
        if constexpr (LPA >= 4) s = fmaf(dpp_f<0x112>(s), mk.up[1], s);
        if constexpr (LPA >= 8) s = fmaf(dpp_f<0x114>(s), mk.up[2], s);
        if constexpr (LPA >= 16) s = fmaf(dpp_f<0x118>(s), mk.up[3], s);
    } else {
        float t;
        s = __shfl_up(v, 1, LPA);
        s = (lig >= 1) ? s : 0.0f;
#pragma unroll
        for (int d = 1; d < LPA; d <<= 1) {
            t = __shfl_up(s, d, LPA);
            s += (lig >= d) ? t : 0.0f;
        }
    }
    return s;
}
template <int LPA>
__device__ __forceinline__ float group_excl_suffix(float v, int lig, const ScanMasks<LPA> &mk) {
    float s;
    if constexpr (LPA <= 16) {
        s = dpp_f<0x101>(v) * mk.dn[0];                      // row_shl:1
        if constexpr (LPA >= 4) s = fmaf(dpp_f<0x101>(s), mk.dn[0], s);
        if constexpr (LPA >= 4) s = fmaf(dpp_f<0x102>(s), mk.dn[1], s);
        if constexpr (LPA >= 8) s = fmaf(dpp_f<0x104>(s), mk.dn[2], s);
        if constexpr (LPA >= 16) s = fmaf(dpp_f<0x108>(s), mk.dn[3], s);
    } else {
        float t;
        s = __shfl_down(v, 1, LPA);
        s = (lig + 1 < LPA) ? s : 0.0f;
#pragma unroll
        for (int d = 1; d < LPA; d <<= 1) {
            t = __shfl_down(s, d, LPA);
            s += (lig + d < LPA) ? t : 0.0f;
        }
    }
    return s;
}

// g0 = (P_est + P_sch)/2 - G/kappa, the unconstrained minimiser of the operator's objective
// (lpsolver.py:196-207: g0 = -a/kappa), in FLOAT: the sum rounded once, then one fused multiply-add
// (inv_kf = 1.0f / (float)kappa in every caller).  The operator's state and its answer are float
// arrays; every kernel that forms g0 -- the dual evaluations, the ADMM forms, the home pass folded
// into the residence sweep -- goes through this function, so the answer max(g0 - d, 0) is the same
// float whichever kernel computes it.  (Round 1 formed g0 in double in each of them: 71 f64
// instructions per wavefront of the sweep, which is VALU-issue bound: -3 us per launch.)
__device__ __forceinline__ float revs_g0f(float pe, float ps, float gm, float inv_kf) {
    return __builtin_fmaf(-gm, inv_kf, 0.5f * (pe + ps));
}

// Node sums that do not depend on the order of summation.  The operator's evaluations sum, per node
// and slot, the residences' answers g (kW) and their squares in double.  Rounded to a multiple of
// 2^-36 (2^-32 for the squares) first, every partial sum below 2^17 (2^21) is a 53-bit number and
// each floating-point addition is EXACT: the evaluation kernel's fixed-order sums and the atomics of
// a sweep that folds the same evaluation give the same bits -- and the rounding (1.5e-11 kW) is far
// below the 1e-8 relative tolerance of the voltage rows (rounding g to float first, 1e-7 kW, is not).
// (round to nearest even by adding and subtracting 3 x 2^15 / 3 x 2^19, whose unit in the last place is 2^-36 / 2^-32 --
// for 0 <= g < 2^15 kW, 0 <= g^2 < 2^19: the same values as rint(g 2^36) 2^-36, two additions instead of three operations)
__device__ __forceinline__ double revs_q36(double g) {
    double r = g + 98304.0;
    asm volatile("" : "+v"(r));             // (the compiler must not fold the pair away)
    return r - 98304.0;
}
__device__ __forceinline__ double revs_q32(double g2) {
    double r = g2 + 1572864.0;
    asm volatile("" : "+v"(r));
    return r - 1572864.0;
}

// clip(v, lo, hi) as one v_med3_f32 (lo <= hi)
__device__ __forceinline__ float clip3(float v, float lo, float hi) {
    return __builtin_amdgcn_fmed3f(v, lo, hi);
}

}  // namespace revs

// More than 64 KB of dynamic LDS has to be granted per kernel -- and per device: one grant per (kernel, current device)
// and size (a `static const` result was per process).  false + revs_last_error when the runtime refuses.
namespace revs {
bool grant_lds(const void *kernel, size_t bytes, const char *who);
}

// Stage stamps of the latency-bound operator launches exist in tuning builds only (csrc/tuning.h, never included by
// the product build: python -m revs_admm_amd.build --out tune/lib.so -DREVS_TUNING -DREVS_KV_STAMPS | -DREVS_BPP_STAMPS | -DREVS_VD_STAMPS)
#ifdef REVS_TUNING
#include "tuning.h"
#else
#define REVS_KVS_BEGIN(ptr) do { } while (0)
#define REVS_KVS(t, i) do { } while (0)
#define REVS_KVV(t, i, val) do { } while (0)
#define BPP_STAMP(i) do { } while (0)
#define VD_STAMP(i) do { } while (0)
#endif
