// VALU issue rate on gfx950, measured in SHADER CYCLES: how many cycles one SIMD needs per wave64 VALU
// instruction -- v_fma_f32, v_pk_fma_f32, v_add_f32, a DPP add, v_med3_f32, v_fma_f64 -- with 1, 2, 4 and 8
// wavefronts resident per SIMD, every wavefront running 8 or 16 INDEPENDENT dependency chains (so that a
// chain's latency never gates issue), on the whole chip at once (as the sweep runs) and on one CU alone.
//
// Why cycles: the wall-clock rate (ns per instruction) folds the clock the chip sustains under the load
// into the figure.  Every wavefront brackets its loop with s_memtime (the shader-clock counter) and
// s_memrealtime (the constant 100 MHz counter): cycles per instruction per SIMD = the wavefront's cycles x
// (wavefronts sharing its SIMD) / instructions issued on that SIMD, and the ratio of the two counters is the
// clock the loop ran at.  Where each wavefront sat (XCC, SE, CU, SIMD: HW_REG_HW_ID / HW_REG_XCC_ID) is
// recorded, and the host prints how many wavefronts really shared a SIMD -- the "waves per SIMD" column is
// measured, not assumed from the grid.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/valu_rate.hip -o tools/probes/valu_rate.bin && tools/probes/valu_rate.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

struct WaveRec { unsigned long long c0, c1, r0, r1; unsigned hw, xcc; };

template <int KIND, int CH>
__global__ __launch_bounds__(1024) void rate_kernel(float *out, WaveRec *rec, int iters, float a, float b) {
    float x[CH];
    float2v y[CH];
    double z[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = float2v{x[i], x[i] + 1.f}; z[i] = x[i]; }
    const float2v a2{a, a}, b2{b, b};
    const double ad = a, bd = b;
    __syncthreads();
    unsigned long long c0, c1, r0, r1;
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    for (int k = 0; k < iters; ++k) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(a2), "v"(b2));
            if (KIND == 2) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
            if (KIND == 3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            if (KIND == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (KIND == 5) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(z[i]) : "v"(ad), "v"(bd));
        }
    }
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {      // (only the kind's own registers live across the loop)
        if (KIND == 1) s += y[i].x + y[i].y;
        else if (KIND == 5) s += (float)z[i];
        else s += x[i];
    }
    if (s == 123.456f) out[0] = s;
    if ((threadIdx.x & 63) == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        rec[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = WaveRec{c0, c1, r0, r1, hw, xcc};
    }
}

template <int KIND, int CH>
static void run(const char *name, int blocks, const char *scope, float *d, WaveRec *drec, int iters, int threads = 256) {
    std::vector<WaveRec> h(blocks * (threads / 64));
    hipLaunchKernelGGL((rate_kernel<KIND, CH>), dim3(blocks), dim3(threads), 0, 0, d, drec, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate_kernel<KIND, CH>), dim3(blocks), dim3(threads), 0, 0, d, drec, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), drec, sizeof(WaveRec) * h.size(), hipMemcpyDeviceToHost);
    // wavefronts per SIMD as placed: key = (xcc, se, sh, cu, simd) from HW_ID [simd 5:4, cu 11:8, sh 12, se 15:13]
    std::map<unsigned, int> per;
    for (auto &w : h) per[((w.xcc & 0xf) << 16) | (w.hw & 0xfff0) >> 4 << 0] += 1;
    std::vector<int> occ;
    for (auto &p : per) occ.push_back(p.second);
    std::sort(occ.begin(), occ.end());
    // cycles per instruction per SIMD, rigorously: on every SIMD, (last wavefront's end - first wavefront's start) by the
    // shader clock, over the instructions its wavefronts issued there (s_memtime is one counter per XCC: the wavefronts
    // of a SIMD share it).  "steady": the same over the window in which ALL of the SIMD's wavefronts were inside their
    // loops (latest start .. earliest end), with the instructions pro-rated -- what the SIMD sustains with `occ`
    // wavefronts resident, free of the ramp at either end.
    std::vector<double> cpi, cpi_steady, mhz;
    const double n = (double)iters * CH;
    std::map<unsigned, std::vector<const WaveRec *>> by;
    for (auto &w : h) by[((w.xcc & 0xf) << 16) | (w.hw & 0xfff0) >> 4].push_back(&w);
    for (auto &kv : by) {
        unsigned long long a0 = ~0ull, a1 = 0, b0 = 0, b1 = ~0ull;
        for (auto *w : kv.second) { a0 = std::min(a0, w->c0); a1 = std::max(a1, w->c1); b0 = std::max(b0, w->c0); b1 = std::min(b1, w->c1); }
        cpi.push_back((double)(a1 - a0) / (n * kv.second.size()));
        if (b1 > b0) {
            double issued = 0.0;
            for (auto *w : kv.second) issued += n * (double)(b1 - b0) / (double)(w->c1 - w->c0);
            cpi_steady.push_back((double)(b1 - b0) / issued);
        }
    }
    for (auto &w : h) mhz.push_back((double)(w.c1 - w.c0) / (double)(w.r1 - w.r0) * 100.0);
    std::sort(cpi.begin(), cpi.end());
    std::sort(cpi_steady.begin(), cpi_steady.end());
    std::sort(mhz.begin(), mhz.end());
    const double steady = cpi_steady.empty() ? 0.0 : cpi_steady[cpi_steady.size() / 2];
    const double instr_per_simd = n * occ[occ.size() / 2];
    printf("%-14s %-9s chains %2d  SIMDs used %4zu  waves/SIMD min %d med %d max %d | cycles per instr per SIMD: span med %.2f (p10 %.2f p90 %.2f), "
           "all-resident window med %.2f | shader clock %.0f MHz | launch %.1f us = %.2f ns per instr per SIMD\n",
           name, scope, CH, occ.size(), occ.front(), occ[occ.size() / 2], occ.back(), cpi[cpi.size() / 2], cpi[cpi.size() / 10],
           cpi[cpi.size() * 9 / 10], steady, mhz[mhz.size() / 2], ms * 1e3, ms * 1e6 / instr_per_simd);
}

int main() {
    float *d;
    WaveRec *drec;
    hipMalloc(&d, 64);
    hipMalloc(&drec, sizeof(WaveRec) * 4 * 256 * 8);
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    printf("device %s, %d CUs, clockRate %d kHz\n", pr.gcnArchName, pr.multiProcessorCount, pr.clockRate);
    const int iters = 4000;
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_add_f32_dpp", "v_med3_f32", "v_add_f32", "v_fma_f64"};
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = pr.multiProcessorCount * wps;   // workgroups of 4 wavefronts: wps wavefronts per SIMD when spread evenly
        char scope[32];
        snprintf(scope, sizeof scope, "chip x%d", wps);
        run<0, 8>(names[0], blocks, scope, d, drec, iters);
        run<0, 16>(names[0], blocks, scope, d, drec, iters);
        run<0, 32>(names[0], blocks, scope, d, drec, iters);
        run<4, 32>(names[4], blocks, scope, d, drec, iters);
        run<1, 8>(names[1], blocks, scope, d, drec, iters);
        run<2, 8>(names[2], blocks, scope, d, drec, iters);
        run<3, 8>(names[3], blocks, scope, d, drec, iters);
        run<4, 8>(names[4], blocks, scope, d, drec, iters);
        run<5, 8>(names[5], blocks, scope, d, drec, iters);
    }
    // ONE workgroup alone on the chip (no power / clock effect of 1024 busy SIMDs): 256 x wps threads = wps wavefronts
    // on each SIMD of one CU
    for (int wps : {1, 2, 4}) {
        char scope[32];
        snprintf(scope, sizeof scope, "one CU x%d", wps);
        run<0, 8>(names[0], 1, scope, d, drec, iters, 256 * wps);
        run<0, 32>(names[0], 1, scope, d, drec, iters, 256 * wps);
        run<1, 8>(names[1], 1, scope, d, drec, iters, 256 * wps);
        run<4, 8>(names[4], 1, scope, d, drec, iters, 256 * wps);
    }
    return 0;
}
