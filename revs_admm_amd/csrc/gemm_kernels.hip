// Skinny dense products on the matrix cores -- gfx950.
//
//   C[m][n] = At^T[m][k] * B[k][n]        (At stored k-major: At[kk][i], row-major)
//
// with n = T (24..192 slots) and m, k = number of constrained feeder nodes.  These
// are the operator side's only dense contractions: the LinDistFlow voltage
// sensitivity R.P the reference forms as `R_res @ g[:,t]` (lpsolver.py:191-193,
// drawing.py:75) and, inside the operator QP, the products with the singular
// vectors of R.  Both operands are k-major so that every MFMA operand fetch is a
// 16-element contiguous run (128 B of f64 / 64 B of f32) straight from global
// memory; R is symmetric, so At = R needs no transpose.
//
// The product is bound by streaming At (m*k elements, each used for only n <= 192
// columns): 6 flop/byte at T=24 in f64.  Tiling for that:
//   * one workgroup = one 16-row tile of C and ALL its column tiles (NT = ceil(n/16)
//     accumulator tiles per wavefront), so At is read exactly once;
//   * the workgroup's 16 wavefronts split K; each issues its At/B loads a whole
//     unrolled batch (8 k-steps) ahead of the MFMAs that consume them, which keeps
//     ~64 KB per CU in flight;
//   * the 16 partial tiles are summed through LDS in a fixed order (bitwise
//     reproducible), one column tile per wavefront.
// v_mfma_f64_16x16x4_f64 for double, v_mfma_f32_16x16x4_f32 (exact f32 fma chain)
// for float.  `batch` independent products can share one launch (gridDim.y), which
// is how the operator issues V^T rhat with U^T w, and V a with U (s a).
#include "common.h"

namespace revs {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mfma;
template <> struct Mfma<double> {
    using acc_t = d4;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // C/D map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct Mfma<float> {
    using acc_t = f4;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // C/D map of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg
    static __device__ __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

// NT <= 2 (T <= 32): 16 wavefronts, loads 8 k-steps ahead (123 VGPRs in f64).
// Wider T needs NT*8 accumulator registers per lane, so the workgroup drops to 8
// wavefronts (256-VGPR budget) and a shorter look-ahead.

template <typename T>
struct GemmOperands {
    const T *At;
    const T *B;
    T *C;
};
template <typename T>
struct GemmBatch {
    GemmOperands<T> op[2];
};

template <typename T, int NT, int kGemmWaves, int kUnroll>
__global__ __launch_bounds__(kGemmWaves * 64) void gemm_tn_kernel(
        int m, int n, int k, GemmBatch<T> batch, int lda, int ldb, int ldc, int accumulate) {
    using M = Mfma<T>;
    using acc_t = typename M::acc_t;
    const GemmOperands<T> op = batch.op[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 15;          // A: output row inside the tile / B: output column
    const int kk = lane >> 4;         // k index inside a 4-deep step
    const int row0 = blockIdx.x * 16;
    const bool arow_ok = (row0 + r) < m;
    // this wave's K range in steps of 4
    const int ksteps = (k + 3) >> 2;
    const int per = (ksteps + kGemmWaves - 1) / kGemmWaves;
    const int s_begin = min(wave * per, ksteps);
    const int s_end = min(s_begin + per, ksteps);

    acc_t acc[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) acc[c] = acc_t{0, 0, 0, 0};

    // clamp out-of-range rows/columns to a valid address and zero the value instead
    const int arow = arow_ok ? row0 + r : 0;
    int bcol[NT];
    bool bok[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        bok[c] = (c * 16 + r) < n;
        bcol[c] = bok[c] ? c * 16 + r : 0;
    }

    int s = s_begin;
    // full batches: every k index in range, loads issued ahead of the MFMAs
    for (; s + kUnroll <= s_end && (s + kUnroll) * 4 <= k; s += kUnroll) {
        T av[kUnroll], bv[kUnroll][NT];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t kidx = (int64_t)(s + u) * 4 + kk;
            av[u] = op.At[kidx * lda + arow];
#pragma unroll
            for (int c = 0; c < NT; ++c) bv[u][c] = op.B[kidx * ldb + bcol[c]];
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const T a = arow_ok ? av[u] : T(0);
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[c] = M::mma(a, bok[c] ? bv[u][c] : T(0), acc[c]);
        }
    }
    // remainder (and the ragged last k-step)
    for (; s < s_end; ++s) {
        const int kidx = s * 4 + kk;
        const bool kok = kidx < k;
        const int64_t kc = kok ? kidx : 0;
        const T a = (kok && arow_ok) ? op.At[kc * lda + arow] : T(0);
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            const T b = (kok && bok[c]) ? op.B[kc * ldb + bcol[c]] : T(0);
            acc[c] = M::mma(a, b, acc[c]);
        }
    }

    // fixed-order reduction of the 16 partial tiles, column tile c by wavefront c % 16
    __shared__ T red[kGemmWaves][4][64];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wave][i][lane] = acc[c][i];
        __syncthreads();
        if (wave == (c % kGemmWaves)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                T v = red[0][i][lane];
#pragma unroll
                for (int w = 1; w < kGemmWaves; ++w) v += red[w][i][lane];
                const int orow = row0 + M::row(lane, i);
                const int ocol = c * 16 + (lane & 15);
                if (orow < m && ocol < n) {
                    T *cp = op.C + (int64_t)orow * ldc + ocol;
                    *cp = accumulate ? (*cp + v) : v;
                }
            }
        }
        __syncthreads();
    }
}

template <typename T>
static int launch_gemm(int m, int n, int k, int nbatch, const GemmBatch<T> &b, int lda, int ldb,
                       int ldc, int accumulate, void *stream, const char *what) {
    REVS_REQUIRE(m > 0 && n > 0 && k > 0, "%s: m=%d n=%d k=%d", what, m, n, k);
    REVS_REQUIRE(n <= 192, "%s: n=%d exceeds 192 columns", what, n);
    for (int i = 0; i < nbatch; ++i)
        REVS_REQUIRE(b.op[i].At && b.op[i].B && b.op[i].C, "%s: null pointer argument", what);
    REVS_REQUIRE(lda >= m && ldb >= n && ldc >= n, "%s: leading dimension too small", what);
    const dim3 grid((m + 15) / 16, nbatch);
    const int nt = (n + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(NT, W, U)                                                                   \
    hipLaunchKernelGGL((gemm_tn_kernel<T, NT, W, U>), grid, dim3(W * 64), 0, s, m, n, k, b, lda, \
                       ldb, ldc, accumulate)
    if (nt <= 1) LAUNCH(1, 16, 8);
    else if (nt <= 2) LAUNCH(2, 16, 8);
    else if (nt <= 4) LAUNCH(4, 8, 4);
    else if (nt <= 6) LAUNCH(6, 8, 4);
    else if (nt <= 8) LAUNCH(8, 8, 2);
    else LAUNCH(12, 8, 2);
#undef LAUNCH
    REVS_CHECK_LAUNCH(what);
    return REVS_OK;
}

}  // namespace revs

using namespace revs;

extern "C" int revs_gemm_tn_f64(int32_t m, int32_t n, int32_t k, const double *At, int32_t lda,
                                const double *B, int32_t ldb, double *C, int32_t ldc,
                                int32_t accumulate, void *stream) {
    GemmBatch<double> b{{{At, B, C}, {nullptr, nullptr, nullptr}}};
    return launch_gemm<double>(m, n, k, 1, b, lda, ldb, ldc, accumulate, stream, "revs_gemm_tn_f64");
}

extern "C" int revs_gemm_tn_f64_x2(int32_t m, int32_t n, int32_t k, const double *At0,
                                   const double *B0, double *C0, const double *At1,
                                   const double *B1, double *C1, void *stream) {
    GemmBatch<double> b{{{At0, B0, C0}, {At1, B1, C1}}};
    return launch_gemm<double>(m, n, k, 2, b, m, n, n, 0, stream, "revs_gemm_tn_f64_x2");
}

extern "C" int revs_gemm_tn_f32(int32_t m, int32_t n, int32_t k, const float *At, int32_t lda,
                                const float *B, int32_t ldb, float *C, int32_t ldc,
                                int32_t accumulate, void *stream) {
    GemmBatch<float> b{{{At, B, C}, {nullptr, nullptr, nullptr}}};
    return launch_gemm<float>(m, n, k, 1, b, lda, ldb, ldc, accumulate, stream, "revs_gemm_tn_f32");
}

extern "C" int revs_voltage_f32(int32_t m, int32_t T, const float *Rt, const float *P, float *V,
                                void *stream) {
    GemmBatch<float> b{{{Rt, P, V}, {nullptr, nullptr, nullptr}}};
    return launch_gemm<float>(m, T, m, 1, b, m, T, T, 0, stream, "revs_voltage_f32");
}
