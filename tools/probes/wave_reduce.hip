#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
template <int CTRL>
__device__ __forceinline__ double dppd(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double swap32_sum(double v) {
    const long long b = __double_as_longlong(v);
    auto l = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
    auto h = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    const double a0 = __longlong_as_double(((long long)h[0] << 32) | l[0]);
    const double a1 = __longlong_as_double(((long long)h[1] << 32) | l[1]);
    return a0 + a1;
}
__device__ __forceinline__ double swap16_sum(double v) {
    const long long b = __double_as_longlong(v);
    auto l = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
    auto h = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    const double a0 = __longlong_as_double(((long long)h[0] << 32) | l[0]);
    const double a1 = __longlong_as_double(((long long)h[1] << 32) | l[1]);
    return a0 + a1;
}
__device__ __forceinline__ double wave_sum_new(double v) {
    v = swap32_sum(v);
    v = swap16_sum(v);
    v += dppd<0x128>(v);
    v += dppd<0x124>(v);
    v += dppd<0x122>(v);
    v += dppd<0x121>(v);
    return v;
}
__device__ __forceinline__ double wave_sum_old(double v) {
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__global__ void k(const double *x, double *o, double *n, long long *tk) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    double v = x[i];
    long long t0 = clock64();
    double a = wave_sum_old(v);
    long long t1 = clock64();
    double b = wave_sum_new(v);
    long long t2 = clock64();
    o[i] = a; n[i] = b;
    if (threadIdx.x == 0) { tk[2 * blockIdx.x] = t1 - t0; tk[2 * blockIdx.x + 1] = t2 - t1; }
}
int main() {
    const int N = 64 * 1024;
    double *hx = (double *)malloc(N * 8), *ho = (double *)malloc(N * 8), *hn = (double *)malloc(N * 8);
    srand(1);
    for (int i = 0; i < N; ++i) hx[i] = (rand() / (double)RAND_MAX - 0.5) * exp2((double)(rand() % 40 - 20));
    double *dx, *dO, *dn; long long *tk;
    hipMalloc(&dx, N * 8); hipMalloc(&dO, N * 8); hipMalloc(&dn, N * 8); hipMalloc(&tk, 2 * 1024 * 8);
    hipMemcpy(dx, hx, N * 8, hipMemcpyHostToDevice);
    k<<<1024, 64>>>(dx, dO, dn, tk);
    hipMemcpy(ho, dO, N * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hn, dn, N * 8, hipMemcpyDeviceToHost);
    long long ht[2048]; hipMemcpy(ht, tk, sizeof(ht), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < N; ++i) if (memcmp(&ho[i], &hn[i], 8)) ++bad;
    printf("mismatches %d of %d; cycles old %lld new %lld\n", bad, N, ht[2000], ht[2001]);
    return bad != 0;
}
