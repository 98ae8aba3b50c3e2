// What does a workgroup pay for a COLUMN of a [m][T] array of doubles (m = 2048, T = 24: every element in
// its own 128-byte line), freshly written by another kernel's atomics -- and does the kind of load matter?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/column_probe.hip -o tools/probes/column_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
constexpr int M = 2048, T = 24;

__global__ void touch(double *a, int n) {      // the producer: device-scope atomics, as the sweep's flush
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) unsafeAtomicAdd(&a[i], 1.0);
}
template <int KIND>
__device__ __forceinline__ double ld(const double *p) {
    if constexpr (KIND == 0) return *p;
    else if constexpr (KIND == 1) return __builtin_nontemporal_load(p);
    else if constexpr (KIND == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <int KIND, int NCOL, int W, int PAD = 0, int SEQ = 0>
__global__ __launch_bounds__(256) void gather(const double *a, double *out, double *us) {
    const int t = blockIdx.x % T;
    double acc = 0.0;
    const long long t0 = wall_clock64();
    double v[NCOL][8];
#pragma unroll
    for (int c = 0; c < NCOL; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = threadIdx.x + 256 * j;
            // W = 1: [m][T]; W > 1: tiles [m / W][T][W]
            const long long idx = W == 1 ? (long long)n * T + t : ((long long)(n / W) * T + t) * W + n % W;
            v[c][j] = ld<KIND>(a + (long long)c * (M * T + PAD) + idx);
            if (SEQ && j == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
    for (int c = 0; c < NCOL; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[c][j];
    __syncthreads();
    const long long t1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) us[blockIdx.x] = (double)(t1 - t0) * 0.01;
}
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
template <int KIND, int NCOL, int W, int PAD = 0, int SEQ = 0>
void run(double *a, double *out, double *us, const char *name) {
    const int wgs = 48;
    std::vector<double> h(wgs);
    double tot = 0, mx = 0;
    for (int r = 0; r < 5; ++r) {
        touch<<<1024, 256>>>(a, 4 * M * T);
        gather<KIND, NCOL, W, PAD, SEQ><<<wgs, 256>>>(a, out, us);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), us, wgs * 8, hipMemcpyDeviceToHost);
        tot += med(h); mx += *std::max_element(h.begin(), h.end());
    }
    printf("%-28s pad %4d seq %d %d column(s), tile width %2d: median %.2f us, slowest workgroup %.2f us\n", name, PAD, SEQ, NCOL, W, tot / 5, mx / 5);
}
int main() {
    double *a, *out, *us;
    (void)hipMalloc(&a, 5 * M * T * 8); (void)hipMemset(a, 0, 5 * M * T * 8);
    (void)hipMalloc(&out, 48 * 256 * 8); (void)hipMalloc(&us, 48 * 8);
    run<0, 1, 1>(a, out, us, "plain load");
    run<0, 3, 1>(a, out, us, "plain load");
    run<1, 3, 1>(a, out, us, "nontemporal load");
    run<2, 3, 1>(a, out, us, "atomic load, agent scope");
    run<3, 3, 1>(a, out, us, "atomic load, system scope");
    run<0, 3, 2>(a, out, us, "plain load");
    run<0, 3, 4>(a, out, us, "plain load");
    run<0, 3, 8>(a, out, us, "plain load");
    run<0, 3, 16>(a, out, us, "plain load");
    run<0, 2, 1>(a, out, us, "plain load");
    run<0, 3, 1, 16>(a, out, us, "plain load");
    run<0, 3, 1, 48>(a, out, us, "plain load");
    run<0, 3, 1, 272>(a, out, us, "plain load");
    run<0, 3, 1, 1040>(a, out, us, "plain load");
    run<0, 3, 1, 0, 1>(a, out, us, "plain load");
    run<0, 4, 1>(a, out, us, "plain load");
    return 0;
}
