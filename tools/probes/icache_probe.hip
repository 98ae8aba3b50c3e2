// How much does cold straight-line code cost?  A workgroup of the operator's launches runs ~10^4
// instructions ONCE, on a compute unit whose instruction cache the sweep has just refilled with its own
// code.  Kernel A: N dependent f64 fma's per thread, fully unrolled (8 bytes of code each) -- timed inside
// the kernel (s_memrealtime, 100 MHz) cold (behind a different, larger kernel on every CU) and warm
// (launched again at once); kernel L: the same chain as a loop.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/icache_probe.hip -o tools/probes/icache_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

template <int N>
__global__ __launch_bounds__(256) void chain_unrolled(double *out, double a, double b, double *stamps) {
    double x = threadIdx.x;
    const long long t0 = wall_clock64();
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fma(x, a, b);
    const long long t1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = x;
    if (threadIdx.x == 0) stamps[blockIdx.x] = (double)(t1 - t0) * 0.01;
}
__global__ __launch_bounds__(256) void chain_loop(double *out, double a, double b, double *stamps, int n) {
    double x = threadIdx.x;
    const long long t0 = wall_clock64();
#pragma unroll 1
    for (int i = 0; i < n; ++i) x = __builtin_fma(x, a, b);
    const long long t1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = x;
    if (threadIdx.x == 0) stamps[blockIdx.x] = (double)(t1 - t0) * 0.01;
}
// the evictor: a different kernel with > 64 KB of code, on every compute unit
template <int N>
__global__ __launch_bounds__(256) void evictor(double *out, double a, double b) {
    double x = threadIdx.x, y = blockIdx.x;
#pragma unroll
    for (int i = 0; i < N; ++i) { x = __builtin_fma(x, a, y); y = __builtin_fma(y, b, x); }
    out[blockIdx.x * 256 + threadIdx.x] = x + y;
}

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

template <int N>
void run(double *out, double *st, int wgs) {
    std::vector<double> h(wgs);
    double cold = 0, warm = 0;
    const int reps = 5;
    for (int r = 0; r < reps; ++r) {
        evictor<6000><<<1024, 256>>>(out, 1.0000001, 0.5);
        chain_unrolled<N><<<wgs, 256>>>(out, 1.0000001, 1e-9, st);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), st, wgs * 8, hipMemcpyDeviceToHost);
        cold += med(h);
        chain_unrolled<N><<<wgs, 256>>>(out, 1.0000001, 1e-9, st);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), st, wgs * 8, hipMemcpyDeviceToHost);
        warm += med(h);
    }
    chain_loop<<<wgs, 256>>>(out, 1.0000001, 1e-9, st, N);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), st, wgs * 8, hipMemcpyDeviceToHost);
    printf("N = %5d fma (%4d KB of code), %d workgroups: cold %.2f us, warm %.2f us, loop %.2f us  -> cold %.1f / warm %.1f / loop %.1f cycles per instruction at 2.4 GHz\n",
           N, N * 8 / 1024, wgs, cold / reps, warm / reps, med(h), cold / reps * 2400 / N, warm / reps * 2400 / N, med(h) * 2400 / N);
}

int main() {
    double *out, *st;
    (void)hipMalloc(&out, 1024 * 256 * 8);
    (void)hipMalloc(&st, 1024 * 8);
    run<512>(out, st, 48);
    run<2048>(out, st, 48);
    run<8192>(out, st, 48);
    run<16384>(out, st, 48);
    run<8192>(out, st, 256);
    return 0;
}
