// The wavefront reductions of csrc/common.h (v_permlane32_swap / v_permlane16_swap + DPP row rotations,
// no LDS permute) against the xor-shuffle butterfly they replace: same bits for the sums (same
// association), same values for max / min / prefix sums -- and what each costs.
//   hipcc -O3 --offload-arch=gfx950 -I include -I revs_admm_amd/csrc tools/probes/wave_reduce.hip -o tools/probes/wave_reduce.bin
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

__device__ __forceinline__ double sum_shfl(double v) {
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ double max_shfl(double v) {
    for (int d = 32; d >= 1; d >>= 1) v = fmax(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ int min_shfl(int v) {
    for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ int scan_shfl(int v) {
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(v, d, 64); if ((threadIdx.x & 63) >= d) v += o; }
    return v;
}
__global__ void k(const double *x, double *o, double *n, int *bad, long long *tk) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const double v = x[i];
    const long long t0 = clock64();
    const double a = sum_shfl(v);
    const long long t1 = clock64();
    const double b = revs::wave_sum_d(v);
    const long long t2 = clock64();
    o[i] = a; n[i] = b;
    const int key = (int)(fabs(v) * 1e6) % 1000;
    int wrong = 0;
    wrong += max_shfl(v) != revs::wave_max_d(v);
    wrong += min_shfl(key) != revs::wave_min_i(key);
    wrong += scan_shfl(key) != revs::wave_incl_scan_i(key);
    if (wrong) atomicAdd(bad, wrong);
    if (threadIdx.x == 0) { tk[2 * blockIdx.x] = t1 - t0; tk[2 * blockIdx.x + 1] = t2 - t1; }
}
int main() {
    const int N = 64 * 1024;
    double *hx = (double *)malloc(N * 8), *ho = (double *)malloc(N * 8), *hn = (double *)malloc(N * 8);
    srand(1);
    for (int i = 0; i < N; ++i) hx[i] = (rand() / (double)RAND_MAX - 0.5) * exp2((double)(rand() % 40 - 20));
    double *dx, *dO, *dn; long long *tk; int *bad, hbad = 0;
    (void)hipMalloc(&dx, N * 8); (void)hipMalloc(&dO, N * 8); (void)hipMalloc(&dn, N * 8); (void)hipMalloc(&tk, 2 * 1024 * 8);
    (void)hipMalloc(&bad, 4); (void)hipMemset(bad, 0, 4);
    (void)hipMemcpy(dx, hx, N * 8, hipMemcpyHostToDevice);
    k<<<1024, 64>>>(dx, dO, dn, bad, tk);
    (void)hipMemcpy(ho, dO, N * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hn, dn, N * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
    long long ht[2048]; (void)hipMemcpy(ht, tk, sizeof(ht), hipMemcpyDeviceToHost);
    int mism = 0;
    for (int i = 0; i < N; ++i) if (memcmp(&ho[i], &hn[i], 8)) ++mism;
    // (the cycle counts are only meaningful when nothing else is between the clock reads: -DWAVE_REDUCE_TIME
    //  builds of round 3 measured 260 cycles for the shuffles and 68 for swaps + DPP)
    printf("mismatches %d of %d; max / min / scan disagreements %d; (clocks around each form: %lld, %lld)\n", mism, N, hbad,
           ht[2000], ht[2001]);
    return (mism != 0 || hbad != 0) ? 1 : 0;
}
