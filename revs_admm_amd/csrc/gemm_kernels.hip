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
// Tiling: a 256-thread workgroup owns one 16 x 16 tile of C; its 4 wavefronts
// split K, each accumulating with v_mfma_f64_16x16x4_f64 (f32:
// v_mfma_f32_16x16x4_f32, exact f32 fma chain), and the 4 partial tiles are
// summed through LDS in a fixed order (deterministic).  m/16 x ceil(n/16)
// workgroups: 142 at m=1126, T=24.  The matrices (<= 32 MiB at m=2048) stay
// resident in L2 / Infinity Cache across the inner iterations of the QP.
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

template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_kernel(int m, int n, int k, const T *__restrict__ At,
                                                      int lda, const T *__restrict__ B, int ldb,
                                                      T *__restrict__ C, int ldc, int accumulate) {
    using M = Mfma<T>;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 15;          // A: output row inside the tile / B: output column
    const int kk = lane >> 4;         // k index inside a 4-deep step
    const int row0 = blockIdx.x * 16;
    const int col0 = blockIdx.y * 16;
    const bool arow_ok = (row0 + r) < m;
    const bool bcol_ok = (col0 + r) < n;
    // this wave's K range, in steps of 4, rounded so the 4 waves cover [0, k)
    const int ksteps = (k + 3) >> 2;
    const int per = (ksteps + 3) >> 2;
    const int s_begin = wave * per;
    const int s_end = min(s_begin + per, ksteps);

    typename M::acc_t acc = {0, 0, 0, 0};
    const T *ap = At + (int64_t)(s_begin * 4 + kk) * lda + row0 + r;
    const T *bp = B + (int64_t)(s_begin * 4 + kk) * ldb + col0 + r;
    int kidx = s_begin * 4 + kk;
#pragma unroll 4
    for (int s = s_begin; s < s_end; ++s) {
        const bool kok = kidx < k;
        const T av = (kok && arow_ok) ? *ap : T(0);
        const T bv = (kok && bcol_ok) ? *bp : T(0);
        acc = M::mma(av, bv, acc);
        ap += (int64_t)4 * lda;
        bp += (int64_t)4 * ldb;
        kidx += 4;
    }
    __shared__ T red[4][4][64];
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][i][lane] = acc[i];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const T v = ((red[0][i][lane] + red[1][i][lane]) + red[2][i][lane]) + red[3][i][lane];
            const int orow = row0 + M::row(lane, i);
            const int ocol = col0 + (lane & 15);
            if (orow < m && ocol < n) {
                T *cp = C + (int64_t)orow * ldc + ocol;
                *cp = accumulate ? (*cp + v) : v;
            }
        }
    }
}

template <typename T>
static int launch_gemm(int m, int n, int k, const T *At, int lda, const T *B, int ldb, T *C,
                       int ldc, int accumulate, void *stream, const char *what) {
    REVS_REQUIRE(m > 0 && n > 0 && k > 0, "%s: m=%d n=%d k=%d", what, m, n, k);
    REVS_REQUIRE(At && B && C, "%s: null pointer argument", what);
    REVS_REQUIRE(lda >= m && ldb >= n && ldc >= n, "%s: leading dimension too small", what);
    const dim3 grid((m + 15) / 16, (n + 15) / 16);
    hipLaunchKernelGGL((gemm_tn_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, m, n, k, At,
                       lda, B, ldb, C, ldc, accumulate);
    REVS_CHECK_LAUNCH(what);
    return REVS_OK;
}

}  // namespace revs

extern "C" int revs_gemm_tn_f64(int32_t m, int32_t n, int32_t k, const double *At, int32_t lda,
                                const double *B, int32_t ldb, double *C, int32_t ldc,
                                int32_t accumulate, void *stream) {
    return revs::launch_gemm<double>(m, n, k, At, lda, B, ldb, C, ldc, accumulate, stream,
                                     "revs_gemm_tn_f64");
}

extern "C" int revs_gemm_tn_f32(int32_t m, int32_t n, int32_t k, const float *At, int32_t lda,
                                const float *B, int32_t ldb, float *C, int32_t ldc,
                                int32_t accumulate, void *stream) {
    return revs::launch_gemm<float>(m, n, k, At, lda, B, ldb, C, ldc, accumulate, stream,
                                    "revs_gemm_tn_f32");
}

extern "C" int revs_voltage_f32(int32_t m, int32_t T, const float *Rt, const float *P, float *V,
                                void *stream) {
    return revs::launch_gemm<float>(m, T, m, Rt, m, P, T, V, T, 0, stream, "revs_voltage_f32");
}
