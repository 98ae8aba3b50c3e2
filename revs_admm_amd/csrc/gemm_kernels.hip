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
//   * one workgroup = one (T<=32: two interleaved) 16-row tile(s) of C and ALL column
//     tiles (NT = ceil(n/16) accumulator tiles per wavefront), so At is read exactly once;
//   * the workgroup's 16 wavefronts split K; each issues its At/B loads a whole
//     unrolled batch (8 k-steps) ahead of the MFMAs that consume them, which keeps
//     ~64 KB per CU in flight;
//   * the 16 partial tiles are summed through LDS in a fixed order (bitwise
//     reproducible), one column tile per wavefront.
// v_mfma_f64_16x16x4_f64 for double, v_mfma_f32_16x16x4_f32 (exact f32 fma chain)
// for float.  `batch` independent products can share one launch (gridDim.y), which
// is how the operator issues V^T rhat with U^T w, and V a with U (s a).
#include "common.h"
#include <type_traits>

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

// NT <= 2 (T <= 32): 16 wavefronts, two register stages of 4 k-steps.  Wider T needs NT*8
// accumulator registers per lane, so the workgroup drops to 8 wavefronts (256-VGPR
// budget) and a shorter look-ahead.
//
// RT = 2 makes the workgroup own TWO interleaved 16-row tiles (rows row0+2i and
// row0+2i+1): every lane fetches its two A values as one 16-byte load and each B
// fragment feeds two MFMAs, halving the B traffic out of L2 (at T=24 every workgroup
// re-reads all of B, which otherwise outweighs the matrix itself).  `ksplit` cuts K
// across gridDim.z so that the launch still fills 256 CUs; split s writes its partial
// product to slab s of C (C + s*m*ldc) and the elementwise consumers add the slabs in
// a fixed order, so the result stays reproducible without atomics.

// One product C = At^T B.  B and C may each be the horizontal concatenation of two
// arrays [X0 | X1] with X0 `csplit` columns wide (csplit = n: a single array), which is
// how the operator multiplies Q^T by [rhat | w] without first packing them.
template <typename T>
struct GemmOperands {
    const T *At;
    const T *B;
    T *C;
    const T *B1;
    T *C1;
    int csplit;
};
template <typename T>
struct GemmBatch {
    GemmOperands<T> op[2];
};

// Optional epilogue of the dual Newton path's product R p (ROWS instantiation): the LAST of
// the gridDim.z K-split workgroups of a row tile to finish sums the slabs of its 32 rows and
// does the row bookkeeping of op_dual_rows_kernel for them on the spot -- v, residual /
// violation, per-(tile, slot) partials -- so that no separate rows kernel is launched.  The
// slabs are handed over inside the launch through the agent-wide coherence point: sc1
// (write-through) stores, an explicit s_waitcnt vmcnt(0) in every storing wave, the
// workgroup barrier, ONE agent-scope counter add per workgroup; the workgroup whose add
// returns gridDim.z - 1 reads the slabs with sc1 loads behind a barrier.  No L2 write-back.
struct GemmRows {
    const double *qpart, *y;      // pnq + 2 m T; multipliers
    double vlo, vhi;
    double *vfull, *viol, *partial, *zero_out;
    unsigned int *counters;       // one per row tile, zero between launches
};

template <typename T> struct Vec2;
template <> struct Vec2<double> { using type = double2; };
template <> struct Vec2<float> { using type = float2; };

template <typename T, int NT, int RT, int kGemmWaves, int kUnroll, bool ROWS = false>
__global__ __launch_bounds__(kGemmWaves * 64) void gemm_tn_kernel(
        int m, int n, int k, GemmBatch<T> batch, int lda, int ldb, int ldc, int accumulate,
        GemmRows rw = GemmRows{}) {
    using M = Mfma<T>;
    using acc_t = typename M::acc_t;
    using vec2 = typename Vec2<T>::type;
    const GemmOperands<T> op = batch.op[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 15;          // A: output row inside the tile / B: output column
    const int kk = lane >> 4;         // k index inside a 4-deep step
    const int row0 = blockIdx.x * 16 * RT;
    // this block's K range (gridDim.z splits), then this wave's share, in steps of 4
    const int ksteps = (k + 3) >> 2;
    const int kper_blk = (ksteps + gridDim.z - 1) / gridDim.z;
    const int b_begin = min((int)blockIdx.z * kper_blk, ksteps);
    const int b_end = min(b_begin + kper_blk, ksteps);
    const int per = (b_end - b_begin + kGemmWaves - 1) / kGemmWaves;
    const int s_begin = min(b_begin + wave * per, b_end);
    const int s_end = min(s_begin + per, b_end);
    const int64_t slab = (int64_t)blockIdx.z * m * ldc;

    acc_t acc[RT][NT];
#pragma unroll
    for (int q = 0; q < RT; ++q)
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[q][c] = acc_t{0, 0, 0, 0};

    // clamp out-of-range rows/columns to a valid address and zero the value instead
    bool aok[RT];
#pragma unroll
    for (int q = 0; q < RT; ++q) aok[q] = (row0 + RT * r + q) < m;
    const int arow = aok[0] ? row0 + RT * r : 0;
    // the 16-byte A fetch needs both rows in range and an even element offset
    const bool pair_ok = (RT == 2) && aok[RT - 1] && ((lda & 1) == 0);
    const T *bptr[NT];      // column base of this lane's B element, per column tile
    bool bok[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const int j = c * 16 + r;
        bok[c] = j < n;
        const bool hi = j >= op.csplit;
        bptr[c] = bok[c] ? (hi ? op.B1 + (j - op.csplit) : op.B + j) : op.B;
    }

    auto load_a = [&](int64_t kidx, T (&out)[RT]) {
        const T *p = op.At + kidx * lda + arow;
        if constexpr (RT == 2) {
            if (pair_ok) {
                const vec2 v = *reinterpret_cast<const vec2 *>(p);
                out[0] = v.x;
                out[1] = v.y;
            } else {
                out[0] = p[0];
                out[1] = aok[1] ? p[1] : T(0);
            }
        } else {
            out[0] = p[0];
        }
    };

    int s = s_begin;
    // Full batches of kUnroll k-steps, software-pipelined with two register stages: the
    // loads of batch b+1 are in flight while the MFMAs of batch b issue (the compiler's
    // counted vmcnt waits release stage b without draining stage b+1).
    const int nfull = (min(s_end, k >> 2) - s_begin) / kUnroll;   // batches with every k in range
    T av[2][kUnroll][RT], bv[2][kUnroll][NT];
    auto load_batch = [&](int st, int sb) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t kidx = (int64_t)(sb + u) * 4 + kk;
            load_a(kidx, av[st][u]);
#pragma unroll
            for (int c = 0; c < NT; ++c) bv[st][u][c] = bptr[c][kidx * ldb];
        }
    };
    auto mma_batch = [&](int st) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                const T b = bok[c] ? bv[st][u][c] : T(0);
#pragma unroll
                for (int q = 0; q < RT; ++q)
                    acc[q][c] = M::mma(aok[q] ? av[st][u][q] : T(0), b, acc[q][c]);
            }
        }
    };
    if (nfull > 0) load_batch(0, s);
    int bidx = 0;
    for (; bidx + 2 <= nfull; bidx += 2) {
        load_batch(1, s + kUnroll);
        mma_batch(0);
        if (bidx + 2 < nfull) load_batch(0, s + 2 * kUnroll);
        mma_batch(1);
        s += 2 * kUnroll;
    }
    if (bidx < nfull) {
        mma_batch(0);
        s += kUnroll;
    }
    // remainder (and the ragged last k-step)
    for (; s < s_end; ++s) {
        const int kidx = s * 4 + kk;
        const bool kok = kidx < k;
        T a[RT];
        load_a(kok ? kidx : 0, a);
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            const T b = (kok && bok[c]) ? bptr[c][(int64_t)kidx * ldb] : T(0);
#pragma unroll
            for (int q = 0; q < RT; ++q)
                acc[q][c] = M::mma((kok && aok[q]) ? a[q] : T(0), b, acc[q][c]);
        }
    }

    // fixed-order reduction of the partial tiles, (row tile q, column tile c) by
    // wavefront (q*NT + c) % kGemmWaves
    __shared__ T red[kGemmWaves][RT][4][64];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
#pragma unroll
        for (int q = 0; q < RT; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wave][q][i][lane] = acc[q][c][i];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < RT; ++q) {
            if (wave == ((q * NT + c) % kGemmWaves)) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    T v = red[0][q][i][lane];
#pragma unroll
                    for (int w = 1; w < kGemmWaves; ++w) v += red[w][q][i][lane];
                    const int orow = row0 + RT * M::row(lane, i) + q;
                    const int ocol = c * 16 + (lane & 15);
                    if (orow < m && ocol < n) {
                        const bool hi = ocol >= op.csplit;
                        T *cp = (hi ? op.C1 + (ocol - op.csplit) : op.C + ocol) + slab +
                                (int64_t)orow * ldc;
                        if constexpr (ROWS)
                            __hip_atomic_store(cp, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else
                            *cp = accumulate ? (*cp + v) : v;
                    }
                }
            }
        }
        // ROWS: the slab stores above are handed to another workgroup below.  A barrier does
        // not drain VMEM (the ISA of this kernel showed `global_store ... sc1; s_barrier;
        // global_atomic_add` with no wait in between), so every storing wave waits for its own
        // write-through stores to complete before it joins the barrier that precedes the
        // counter increment (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 stores +
        // vmcnt(0) in every storing wave + barrier + one agent-scope add; consumer: sc1 loads
        // behind the returned add and a barrier).
        if constexpr (ROWS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if constexpr (ROWS && std::is_same_v<T, double>) {
        __shared__ int last_s;
        if (threadIdx.x == 0) {
            const unsigned int old = __hip_atomic_fetch_add(rw.counters + blockIdx.x, 1u, __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT);
            last_s = old == gridDim.z - 1;
            if (last_s)
                __hip_atomic_store(rw.counters + blockIdx.x, 0u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!last_s) return;                                  // uniform
        constexpr int kRows = 16 * RT;
        double(*buf)[kRows][32] = reinterpret_cast<double(*)[kRows][32]>(&red[0][0][0][0]);
        static_assert(sizeof(red) >= 4 * kRows * 32 * sizeof(double), "epilogue scratch");
        const int64_t total = (int64_t)m * ldc;
        for (int e = threadIdx.x; e < kRows * 32; e += kGemmWaves * 64) {
            const int rr = e >> 5, t = e & 31;
            const int r = row0 + rr;
            double res = 0.0, dterm = 0.0, sup = 0.0, vio = 0.0;
            if (r < m && t < n) {
                const int64_t i = (int64_t)r * ldc + t;
                double v = 0.0;
                for (unsigned q = 0; q < gridDim.z; ++q)
                    v += __hip_atomic_load(op.C + i + q * total, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                rw.vfull[i] = v;
                const double yv = rw.y[i];
                const bool up = yv > 0.0 || (yv == 0.0 && v > rw.vhi);
                const double b = up ? rw.vhi : rw.vlo;
                const double vi = fmax(fmax(v - rw.vhi, rw.vlo - v), 0.0);
                res = yv != 0.0 ? fabs(v - b) : vi;
                dterm = rw.qpart[i] - fmax(rw.vhi * yv, rw.vlo * yv);
                sup = yv != 0.0 ? 1.0 : 0.0;
                vio = (yv == 0.0 && vi > 0.0) ? 1.0 : 0.0;
                rw.viol[i] = yv != 0.0 ? 0.0 : vi;
                if (rw.zero_out) rw.zero_out[i] = 0.0;
            }
            buf[0][rr][t] = res; buf[1][rr][t] = dterm; buf[2][rr][t] = sup; buf[3][rr][t] = vio;
        }
        __syncthreads();
        if ((int)threadIdx.x < 4 * 32) {                      // one thread per (quantity, slot)
            const int q = threadIdx.x >> 5, t = threadIdx.x & 31;
            double acc = 0.0;
            if (q == 0) {
#pragma unroll 8
                for (int rr = 0; rr < kRows; ++rr) acc = fmax(acc, buf[0][rr][t]);
            } else {
#pragma unroll 8
                for (int rr = 0; rr < kRows; ++rr) acc += buf[q][rr][t];    // fixed order
            }
            if (t < n) rw.partial[((int64_t)blockIdx.x * n + t) * 4 + q] = acc;
        }
    }
}

template <typename T>
static int launch_gemm(int m, int n, int k, int nbatch, int ksplit, const GemmBatch<T> &b,
                       int lda, int ldb, int ldc, int accumulate, void *stream,
                       const char *what) {
    REVS_REQUIRE(m > 0 && n > 0 && k > 0, "%s: m=%d n=%d k=%d", what, m, n, k);
    REVS_REQUIRE(n <= 192, "%s: n=%d exceeds 192 columns", what, n);
    REVS_REQUIRE(ksplit >= 1 && ksplit <= 8, "%s: ksplit=%d", what, ksplit);
    REVS_REQUIRE(ksplit == 1 || !accumulate, "%s: accumulate with ksplit", what);
    for (int i = 0; i < nbatch; ++i)
        REVS_REQUIRE(b.op[i].At && b.op[i].B && b.op[i].C, "%s: null pointer argument", what);
    REVS_REQUIRE(lda >= m && ldb >= b.op[0].csplit && ldc >= b.op[0].csplit,
                 "%s: leading dimension too small", what);
    const int nt = (n + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(NT, RT, W, U)                                                                   \
    hipLaunchKernelGGL((gemm_tn_kernel<T, NT, RT, W, U>),                                      \
                       dim3((m + 16 * RT - 1) / (16 * RT), nbatch, ksplit), dim3(W * 64), 0, s, \
                       m, n, k, b, lda, ldb, ldc, accumulate)
    if (nt <= 1) LAUNCH(1, 2, 16, 4);
    else if (nt <= 2) LAUNCH(2, 2, 16, 4);
    else if (nt <= 3) LAUNCH(3, 2, 16, 2);
    else if (nt <= 4) LAUNCH(4, 1, 8, 2);
    else if (nt <= 6) LAUNCH(6, 1, 8, 2);
    else if (nt <= 8) LAUNCH(8, 1, 8, 1);
    else LAUNCH(12, 1, 8, 1);
#undef LAUNCH
    REVS_CHECK_LAUNCH(what);
    return REVS_OK;
}

}  // namespace revs

using namespace revs;

extern "C" int revs_gemm_tn_f64(int32_t m, int32_t n, int32_t k, const double *At, int32_t lda,
                                const double *B, int32_t ldb, double *C, int32_t ldc,
                                int32_t accumulate, void *stream) {
    GemmBatch<double> b{{{At, B, C, nullptr, nullptr, n}, {}}};
    return launch_gemm<double>(m, n, k, 1, 1, b, lda, ldb, ldc, accumulate, stream,
                               "revs_gemm_tn_f64");
}

extern "C" int revs_gemm_tn_f64_split(int32_t m, int32_t n, int32_t k, const double *At,
                                      const double *B, double *C, int32_t ksplit, void *stream) {
    GemmBatch<double> b{{{At, B, C, nullptr, nullptr, n}, {}}};
    return launch_gemm<double>(m, n, k, 1, ksplit, b, m, n, n, 0, stream, "revs_gemm_tn_f64_split");
}

extern "C" int revs_op_dual_product_rows(int32_t m, int32_t T, const double *Rt, const double *p,
                                         const double *pnq, const double *y, double vlo,
                                         double vhi, int32_t ksplit,
                                         double *v_slabs, double *vfull, double *viol,
                                         double *partial, double *zero_out, uint32_t *counters,
                                         void *stream) {
    REVS_REQUIRE(m > 0 && T > 0 && T <= 32 && Rt && p && pnq && y && v_slabs && vfull && viol &&
                 partial && counters && ksplit >= 1 && ksplit <= 8 && vlo <= vhi,
                 "revs_op_dual_product_rows: bad argument (T <= 32)");
    REVS_REQUIRE(zero_out != p, "revs_op_dual_product_rows: zero_out must not be the operand p");
    GemmBatch<double> b{{{Rt, p, v_slabs, nullptr, nullptr, T}, {}}};
    const GemmRows rw{pnq + 2 * (int64_t)m * T, y, vlo, vhi, vfull, viol, partial, zero_out, counters};
    hipLaunchKernelGGL((gemm_tn_kernel<double, 2, 2, 16, 4, true>), dim3((m + 31) / 32, 1, ksplit),
                       dim3(16 * 64), 0, (hipStream_t)stream, m, T, m, b, m, T, T, 0, rw);
    REVS_CHECK_LAUNCH("revs_op_dual_product_rows");
    return REVS_OK;
}

extern "C" int revs_gemm_tn_f64_x2(int32_t m, int32_t n, int32_t k, const double *At0,
                                   const double *B0, double *C0, const double *At1,
                                   const double *B1, double *C1, int32_t ksplit, void *stream) {
    GemmBatch<double> b{{{At0, B0, C0, nullptr, nullptr, n}, {At1, B1, C1, nullptr, nullptr, n}}};
    return launch_gemm<double>(m, n, k, 2, ksplit, b, m, n, n, 0, stream, "revs_gemm_tn_f64_x2");
}

extern "C" int revs_gemm_tn_f64_cat(int32_t m, int32_t T, int32_t k, const double *At,
                                    const double *B0, const double *B1, double *C0, double *C1,
                                    int32_t ksplit, void *stream) {
    REVS_REQUIRE(B1 && C1, "revs_gemm_tn_f64_cat: null pointer argument");
    GemmBatch<double> b{{{At, B0, C0, B1, C1, T}, {}}};
    return launch_gemm<double>(m, 2 * T, k, 1, ksplit, b, m, T, T, 0, stream,
                               "revs_gemm_tn_f64_cat");
}

extern "C" int revs_gemm_tn_f32(int32_t m, int32_t n, int32_t k, const float *At, int32_t lda,
                                const float *B, int32_t ldb, float *C, int32_t ldc,
                                int32_t accumulate, void *stream) {
    GemmBatch<float> b{{{At, B, C, nullptr, nullptr, n}, {}}};
    return launch_gemm<float>(m, n, k, 1, 1, b, lda, ldb, ldc, accumulate, stream,
                              "revs_gemm_tn_f32");
}

extern "C" int revs_voltage_f32(int32_t m, int32_t T, const float *Rt, const float *P, float *V,
                                void *stream) {
    GemmBatch<float> b{{{Rt, P, V, nullptr, nullptr, T}, {}}};
    return launch_gemm<float>(m, T, m, 1, 1, b, m, T, T, 0, stream, "revs_voltage_f32");
}
