// VALU issue rate on gfx950, measured: how many wave64 instructions per SIMD-cycle for v_fma_f32,
// v_pk_fma_f32, a DPP add and v_med3_f32, at 1..8 wavefronts per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float *out, int iters, float a, float b) {
    float x[8];
    float2v y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = float2v{x[i], x[i] + 1.f}; }
    const float2v a2{a, a}, b2{b, b};
    long long t0 = clock64();
    for (int k = 0; k < iters; ++k) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(a2), "v"(b2));
            if (KIND == 2) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
            if (KIND == 3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            if (KIND == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        }
    }
    long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y;
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (float)(t1 - t0);
}

int main() {
    float *d;
    hipMalloc(&d, 64);
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_add_f32_dpp", "v_med3_f32", "v_add_f32"};
    const int iters = 4000;
    for (int kind = 0; kind < 5; ++kind) {
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = 256 * wps;      // 256 CUs x wps workgroups of 4 wavefronts = wps wavefronts per SIMD
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&] {
                if (kind == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
                if (kind == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
                if (kind == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
                if (kind == 3) hipLaunchKernelGGL(rate_kernel<3>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
                if (kind == 4) hipLaunchKernelGGL(rate_kernel<4>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
            };
            launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0.f, h[2];
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
            const double instr_per_simd = (double)wps * iters * 8;
            printf("%-14s %d waves/SIMD: %8.1f us, %.2f ns per instruction per SIMD, %.2f clock64 ticks per instr of one wave\n",
                   names[kind], wps, ms * 1e3, ms * 1e6 / instr_per_simd, h[1] / (iters * 8.0));
        }
    }
    return 0;
}
