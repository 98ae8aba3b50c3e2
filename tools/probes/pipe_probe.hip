// Does fetching a wavefront's NEXT group of residences with gfx950's direct global -> LDS loads
// (no registers held while the current group is being solved) pay for a sweep-like kernel?
// Stand-alone probe, not part of the library:
//     hipcc -O3 --offload-arch=gfx950 tools/probes/pipe_probe.hip -o gpurun_out/pipe_probe && gpurun_out/pipe_probe
// Shape of the real sweep at T = 24: 8 lanes per residence, 3 slots per lane, 8 residences per
// wavefront; per residence-slot 4 floats read (load, P_est, P_sch, Gamma) and 3 written; a
// PDHG-like loop of ITERS passes (fma, clip, 8-lane DPP sum) in between.
//   A  one group per wavefront, 3125 workgroups (as the library does)
//   B  persistent workgroups (8 per CU), every wavefront walks its groups, plain loads
//   C  as B, the next group's profiles requested into LDS before the current one is solved
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int T = 24, SPL = 3, LPA = 8, ITERS = 12;
struct alignas(4) P3 { float v[3]; };

__device__ __forceinline__ float group_sum8(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    return v;
}

__device__ __forceinline__ void solve_store(const float (&L)[3], const float (&pe)[3], const float (&ps)[3],
                                            const float (&gm)[3], float *ps_out, float *gm_out, float *pe2,
                                            long long o, float kappa) {
    float x[3], b[3], w[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { x[j] = ps[j] * 0.1f; b[j] = gm[j] - kappa * pe[j]; w[j] = 1.0f + L[j] * 0.f; }
    float yy = 0.f, sx = x[0] + x[1] + x[2];
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
        const float s = 0.3f * yy;
        float sn = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float xn = __builtin_amdgcn_fmed3f(fmaf(0.6f, x[j], -0.01f * b[j]) - s, 0.f, w[j]);
            sn += xn; x[j] = xn;
        }
        const float acc = fmaf(2.0f, sn, -sx);
        sx = sn;
        const float v = fmaf(0.2f, group_sum8(acc), yy);
        yy = v - __builtin_amdgcn_fmed3f(v, 0.5f, 1.0f);
    }
    P3 a, c, d;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float p = x[j] * 4.8f + L[j];
        a.v[j] = p;
        c.v[j] = gm[j] + 0.5f * kappa * (pe[j] - p);
        d.v[j] = fmaxf(0.5f * (pe[j] + p) - c.v[j] / kappa, 0.f);
    }
    *reinterpret_cast<P3 *>(ps_out + o) = a;
    *reinterpret_cast<P3 *>(gm_out + o) = c;
    *reinterpret_cast<P3 *>(pe2 + o) = d;
}

struct Args { const float *load, *pe, *ps, *gm; float *ps_out, *gm_out, *pe2; long long n; int groups; };

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) void kernel_a(const Args a) {
    const long long agent = (long long)blockIdx.x * 32 + threadIdx.x / LPA;
    if (agent >= a.n) return;
    const long long o = agent * T + (threadIdx.x & 7) * SPL;
    float L[3], pe[3], ps[3], gm[3];
    const P3 l = *reinterpret_cast<const P3 *>(a.load + o), p = *reinterpret_cast<const P3 *>(a.pe + o),
             s = *reinterpret_cast<const P3 *>(a.ps + o), g = *reinterpret_cast<const P3 *>(a.gm + o);
#pragma unroll
    for (int j = 0; j < 3; ++j) { L[j] = l.v[j]; pe[j] = p.v[j]; ps[j] = s.v[j]; gm[j] = g.v[j]; }
    solve_store(L, pe, ps, gm, a.ps_out, a.gm_out, a.pe2, o, 5.0f);
}

// A plus, one at a time, what the real sweep also does (F bit mask):
//   1  the residence record (32 bytes, the 8 lanes of a residence read the same words) and the
//      carried multiplier (one float per residence, read and written back)
//   2  per-residence outputs: diff, dsq (8-lane sums), status -- 12 bytes written by lane 0
//   4  convergence test every 4 passes (8-lane max, wave-wide vote) instead of a fixed count
//   8  node sums of the third output: LDS accumulation per workgroup, one f64 atomic per node and slot
struct Rec { int ev, start, end, nmin, nmax; float rating, capacity, initial; };
struct ArgsD { Args a; const Rec *homes; float *yst; float *diff, *dsq; int *status; const int *node_of; double *pnext; };
template <int F>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) void kernel_d(const ArgsD d) {
    const Args &a = d.a;
    const int lig = threadIdx.x & 7;
    const long long first = (long long)blockIdx.x * 32;
    const long long agent = first + threadIdx.x / LPA;
    const bool live = agent < a.n;
    const long long ag = live ? agent : a.n - 1;
    const long long o = ag * T + lig * SPL;
    float L[3], pe[3], ps[3], gm[3];
    const P3 l = *reinterpret_cast<const P3 *>(a.load + o), p = *reinterpret_cast<const P3 *>(a.pe + o),
             s = *reinterpret_cast<const P3 *>(a.ps + o), g = *reinterpret_cast<const P3 *>(a.gm + o);
#pragma unroll
    for (int j = 0; j < 3; ++j) { L[j] = l.v[j]; pe[j] = p.v[j]; ps[j] = s.v[j]; gm[j] = g.v[j]; }
    Rec h{1, 0, 24, 0, 24, 4.8f, 20.f, 0.2f};
    float yy = 0.f;
    if (F & 1) { h = d.homes[ag]; yy = d.yst[ag]; }
    __shared__ double nacc[4][24];
    int base = 0;
    if (F & 8) {
        for (int i = threadIdx.x; i < 96; i += 256) (&nacc[0][0])[i] = 0.0;
        base = d.node_of[first < a.n ? first : a.n - 1];
        __syncthreads();
    }
    float x[3], b[3], w[3];
    const float kappa = 5.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int t = lig * 3 + j;
        x[j] = ps[j] * 0.1f; b[j] = gm[j] - kappa * pe[j];
        w[j] = (t >= h.start && t < h.end && h.ev) ? 1.0f : 0.0f;
    }
    float sx = x[0] + x[1] + x[2];
    bool done = false;
    int iters = 0;
    auto pass = [&](bool res) -> float {
        const float sc = 0.3f * yy;
        float sn = 0.f, dmax = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float xn = __builtin_amdgcn_fmed3f(fmaf(0.6f, x[j], -0.01f * b[j]) - sc, 0.f, w[j]);
            sn += xn;
            if (res) dmax = fmaxf(dmax, fabsf(xn - x[j]));
            x[j] = xn;
        }
        const float acc = fmaf(2.0f, sn, -sx);
        sx = sn;
        const float v = fmaf(0.2f, group_sum8(acc), yy);
        const float yn = v - __builtin_amdgcn_fmed3f(v, 0.5f, 1.0f);
        if (res) dmax = fmaxf(dmax, fabsf(yn - yy));
        yy = yn;
        return dmax;
    };
    if (F & 4) {
        for (int it = 0; it < 4000; it += 4) {
            if (__all(done)) break;
            for (int c = 1; c < 4; ++c) pass(false);
            float r = pass(true);
            iters += done ? 0 : 4;
            r = fmaxf(r, __shfl_xor(r, 1, 64)); r = fmaxf(r, __shfl_xor(r, 2, 64)); r = fmaxf(r, __shfl_xor(r, 4, 64));
            done = done || (r <= 1e-6f) || iters >= ITERS;      // (the probe's data: stop where A stops)
        }
    } else {
#pragma unroll 1
        for (int it = 0; it < ITERS; ++it) pass(false);
    }
    P3 pa, pc, pd;
    float dsum = 0.f, qsum = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float pp = x[j] * h.rating + L[j];
        pa.v[j] = pp;
        const float chk = pe[j] - pp;
        pc.v[j] = gm[j] + 0.5f * kappa * chk;
        pd.v[j] = fmaxf(0.5f * (pe[j] + pp) - pc.v[j] / kappa, 0.f);
        dsum += chk * chk; qsum += (pp - ps[j]) * (pp - ps[j]);
    }
    if (live) {
        *reinterpret_cast<P3 *>(a.ps_out + o) = pa;
        *reinterpret_cast<P3 *>(a.gm_out + o) = pc;
        *reinterpret_cast<P3 *>(a.pe2 + o) = pd;
    }
    if (F & 2) {
        dsum = group_sum8(dsum); qsum = group_sum8(qsum);
        if (live && lig == 0) { d.diff[ag] = sqrtf(dsum) / T; d.dsq[ag] = qsum; d.status[ag] = iters << 8; }
    }
    if ((F & 1) && live && lig == 0) d.yst[ag] = yy;
    if (F & 8) {
        if (live) {
            const int ln = d.node_of[ag] - base;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (ln < 4) atomicAdd(&nacc[ln][lig * 3 + j], (double)pd.v[j]);
                else atomicAdd(&d.pnext[(long long)(base + ln) * T + lig * 3 + j], (double)pd.v[j]);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 96; i += 256) {
            const double v = (&nacc[0][0])[i];
            if (v != 0.0) atomicAdd(&d.pnext[(long long)(base + i / 24) * T + i % 24], v);
        }
    }
}

// a "group" = the 8 residences of one wavefront; wavefront w of the grid walks groups w, w + W, ...
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) void kernel_b(const Args a) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), W = gridDim.x * 4;
    for (int g = wave; g < a.groups; g += W) {
        const long long agent = (long long)g * 8 + lane / LPA;
        if (agent >= a.n) continue;
        const long long o = agent * T + (lane & 7) * SPL;
        float L[3], pe[3], ps[3], gm[3];
        const P3 l = *reinterpret_cast<const P3 *>(a.load + o), p = *reinterpret_cast<const P3 *>(a.pe + o),
                 s = *reinterpret_cast<const P3 *>(a.ps + o), q = *reinterpret_cast<const P3 *>(a.gm + o);
#pragma unroll
        for (int j = 0; j < 3; ++j) { L[j] = l.v[j]; pe[j] = p.v[j]; ps[j] = s.v[j]; gm[j] = q.v[j]; }
        solve_store(L, pe, ps, gm, a.ps_out, a.gm_out, a.pe2, o, 5.0f);
    }
}

typedef __attribute__((address_space(3))) float lds_f;
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) void kernel_c(const Args a) {
    __shared__ float buf[4][4][64 * 3];           // [wavefront][profile][lane x 3 floats]
    const int lane = threadIdx.x & 63, wl = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wl, W = gridDim.x * 4;
    auto request = [&](int g) {                   // the four profiles of group g -> this wavefront's LDS
        long long agent = (long long)g * 8 + lane / LPA;
        agent = agent < a.n ? agent : a.n - 1;    // (clamped: always a mapped address)
        const long long o = agent * T + (lane & 7) * SPL;
        __builtin_amdgcn_global_load_lds(a.load + o, (lds_f *)&buf[wl][0][0], 12, 0, 0);
        __builtin_amdgcn_global_load_lds(a.pe + o, (lds_f *)&buf[wl][1][0], 12, 0, 0);
        __builtin_amdgcn_global_load_lds(a.ps + o, (lds_f *)&buf[wl][2][0], 12, 0, 0);
        __builtin_amdgcn_global_load_lds(a.gm + o, (lds_f *)&buf[wl][3][0], 12, 0, 0);
    };
    int g = wave;
    if (g < a.groups) request(g);
    for (; g < a.groups; g += W) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float L[3], pe[3], ps[3], gm[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            L[j] = buf[wl][0][lane * 3 + j]; pe[j] = buf[wl][1][lane * 3 + j];
            ps[j] = buf[wl][2][lane * 3 + j]; gm[j] = buf[wl][3][lane * 3 + j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the buffer is free again
        if (g + W < a.groups) request(g + W);
        const long long agent = (long long)g * 8 + lane / LPA;
        if (agent < a.n)
            solve_store(L, pe, ps, gm, a.ps_out, a.gm_out, a.pe2, agent * T + (lane & 7) * SPL, 5.0f);
    }
}

int main() {
    const long long n = 100000, nt = n * T;
    std::vector<float> h(nt);
    for (long long i = 0; i < nt; ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f;
    float *d[10];
    for (int i = 0; i < 10; ++i) { CHECK(hipMalloc((void **)&d[i], nt * 4)); CHECK(hipMemcpy(d[i], h.data(), nt * 4, hipMemcpyHostToDevice)); }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int groups = (int)((n + 7) / 8);
    auto run = [&](const char *name, int which, int grid) -> int {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            for (int k = 0; k < 200; ++k) {        // outputs of one launch are inputs of the next, as in the loop
                const int s = k & 1;
                Args a{d[0], d[1 + s], d[3 + s], d[5 + s], d[3 + (s ^ 1)], d[5 + (s ^ 1)], d[1 + (s ^ 1)], n, groups};
                if (which == 0) hipLaunchKernelGGL(kernel_a, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, 0, a);
                else if (which == 1) hipLaunchKernelGGL(kernel_b, dim3(grid), dim3(256), 0, 0, a);
                else hipLaunchKernelGGL(kernel_c, dim3(grid), dim3(256), 0, 0, a);
            }
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("%-58s %6.2f us per launch (%.2f TB/s of 67.2 MB)\n", name, best / 200 * 1e3, 67.2e6 / (best / 200 * 1e-3) / 1e12);
        return 0;
    };
    if (run("A  one group per wavefront, 3125 workgroups", 0, 0)) return 1;
    {   // D: features of the real sweep on top of A
        std::vector<Rec> hr(n);
        std::vector<int> nof(n);
        for (long long i = 0; i < n; ++i) {
            hr[i] = Rec{(int)(i * 7919 % 2 == 0), (int)(i % 11), 24 - (int)(i % 5), 3, 20, 4.8f, 20.f, 0.2f};
            nof[i] = (int)(i * 2048 / n);
        }
        Rec *dh; float *yst, *df, *dq; int *stt, *dn; double *pn;
        CHECK(hipMalloc((void **)&dh, n * sizeof(Rec))); CHECK(hipMemcpy(dh, hr.data(), n * sizeof(Rec), hipMemcpyHostToDevice));
        CHECK(hipMalloc((void **)&dn, n * 4)); CHECK(hipMemcpy(dn, nof.data(), n * 4, hipMemcpyHostToDevice));
        CHECK(hipMalloc((void **)&yst, n * 4)); CHECK(hipMemset(yst, 0, n * 4));
        CHECK(hipMalloc((void **)&df, n * 4)); CHECK(hipMalloc((void **)&dq, n * 4)); CHECK(hipMalloc((void **)&stt, n * 4));
        CHECK(hipMalloc((void **)&pn, 2048 * T * 8)); CHECK(hipMemset(pn, 0, 2048 * T * 8));
        auto rund = [&](const char *name, int F) -> int {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                for (int k = 0; k < 200; ++k) {
                    const int s = k & 1;
                    ArgsD a{{d[0], d[1 + s], d[3 + s], d[5 + s], d[3 + (s ^ 1)], d[5 + (s ^ 1)], d[1 + (s ^ 1)], n, groups},
                            dh, yst, df, dq, stt, dn, pn};
                    const dim3 grid((unsigned)((n + 31) / 32));
                    switch (F) {
                        case 0: hipLaunchKernelGGL(kernel_d<0>, grid, dim3(256), 0, 0, a); break;
                        case 1: hipLaunchKernelGGL(kernel_d<1>, grid, dim3(256), 0, 0, a); break;
                        case 3: hipLaunchKernelGGL(kernel_d<3>, grid, dim3(256), 0, 0, a); break;
                        case 7: hipLaunchKernelGGL(kernel_d<7>, grid, dim3(256), 0, 0, a); break;
                        case 8: hipLaunchKernelGGL(kernel_d<8>, grid, dim3(256), 0, 0, a); break;
                        default: hipLaunchKernelGGL(kernel_d<15>, grid, dim3(256), 0, 0, a); break;
                    }
                }
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            printf("%-58s %6.2f us per launch\n", name, best / 200 * 1e3);
            return 0;
        };
        if (rund("D  A restated with windows (no extra feature)", 0)) return 1;
        if (rund("D  + residence record and carried multiplier", 1)) return 1;
        if (rund("D  + diff / dsq / status per residence", 3)) return 1;
        if (rund("D  + convergence test every 4 passes", 7)) return 1;
        if (rund("D  node sums only (LDS + f64 atomics)", 8)) return 1;
        if (rund("D  all of them", 15)) return 1;
    }
    for (int grid : {1024, 2048, 3125})
        { char nm[96]; snprintf(nm, 96, "B  persistent, plain loads, %d workgroups", grid); if (run(nm, 1, grid)) return 1; }
    for (int grid : {1024, 1536, 2048})
        { char nm[96]; snprintf(nm, 96, "C  persistent, next group -> LDS, %d workgroups", grid); if (run(nm, 2, grid)) return 1; }
    return 0;
}
