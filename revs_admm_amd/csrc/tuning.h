// Tuning builds only (-DREVS_TUNING; the product build never includes this header): stage stamps inside the
// latency-bound operator launches.
//   -DREVS_KV_STAMPS   wall-clock ticks (100 MHz) at up to 32 points of a slot's way through the folded chain's operator
//                      launch (op_chain_kv_kernel), written by thread 0 of its second half's workgroups; printed by
//                      revs_plan_chain_fold_run at the end of a call (tools/regime_run.py with REVS_LIB=<the build>)
//   -DREVS_BPP_STAMPS  the same inside op_dual_bpp_kernel (tools/bpp_stamps.py)
//   -DREVS_VD_STAMPS   the same inside stream_block_verdict_kernel, every workgroup (tools/verdict_stamps.py)
#pragma once
#if defined(REVS_KV_STAMPS) && defined(REVS_KVS_TU)      // (the operator kernels' translation unit only)
namespace revs {
// (the pointer sits in LDS, put there by thread 0 of a stamped workgroup: reading it waits for no
// outstanding vector load -- a stamp must not order the code it measures)
static __shared__ double *kvs_lds;
}
#define REVS_KVS_BEGIN(ptr) do { if (threadIdx.x == 0) revs::kvs_lds = (ptr); } while (0)
#define REVS_KVS(t, i) do { if (threadIdx.x == 0 && revs::kvs_lds) \
        revs::kvs_lds[32 * (t) + (i)] = (double)wall_clock64(); } while (0)
#define REVS_KVV(t, i, val) do { if (threadIdx.x == 0 && revs::kvs_lds) revs::kvs_lds[32 * (t) + (i)] = (double)(val); } while (0)
#else
#define REVS_KVS_BEGIN(ptr) do { } while (0)
#define REVS_KVS(t, i) do { } while (0)
#define REVS_KVV(t, i, val) do { } while (0)
#endif
//   -DREVS_KV_STAMPS -DREVS_ROWS_STAMPS   the evaluations' rows + selection launch (op_tree_rows_kernel<true>) writes the
//                      same stamps into g_rows_stamps (tools/rows_stamps.py)
#if defined(REVS_KV_STAMPS) && defined(REVS_ROWS_STAMPS) && defined(REVS_KVS_TU)
namespace revs { __device__ double g_rows_stamps[256 * 32]; }
#define REVS_ROWS_STAMP_PTR revs::g_rows_stamps
#else
#define REVS_ROWS_STAMP_PTR nullptr
#endif
#if defined(REVS_BPP_STAMPS) && defined(REVS_KVS_TU)
namespace revs { __device__ double g_bpp_stamps[256][32]; }
#define BPP_STAMP(i) do { if (threadIdx.x == 0 && (i) < 32) revs::g_bpp_stamps[blockIdx.x][i] = (double)wall_clock64(); } while (0)
#else
#define BPP_STAMP(i) do { } while (0)
#endif
#if defined(REVS_VD_STAMPS) && defined(REVS_AGENT_TU)
namespace revs { __device__ double g_vd_stamps[1024][8]; }
#define VD_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) revs::g_vd_stamps[blockIdx.x][i] = (double)wall_clock64(); } while (0)
#else
#define VD_STAMP(i) do { } while (0)
#endif
