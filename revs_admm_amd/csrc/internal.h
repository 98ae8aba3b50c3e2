// Calls between the translation units of librevs_admm.so that are not part of the C ABI.
#pragma once
#include "common.h"
#include "tree_body.h"

namespace revs {

// What the streaming steady state adds to a sweep's launch (AgentArgs in agent_kernels.hip).
struct StreamExtra {
    StreamCtl *ctl;          // control block (device); NULL: an ordinary launch
    unsigned int seq, base_seq;   // this launch; the first launch of the call it belongs to
    bool verdict;            // false: silencing only (seq, base_seq, flags) -- verdicts by blocks
    TreeArgs tree;           // tree.n > 0: the first T workgroups judge the voltage rows
    const double *p_in;      // node sums they judge
    double *p_zero;          // array they clear for the launch after this one (or NULL)
    double vlo, vhi, vtol;
    double *rec;             // record slot of this launch (device address of pinned memory)
    unsigned int *flags;     // status bits (device address of pinned memory) or NULL
    int32_t m;
    // several ADMM iterations per launch (AgentArgs::kin in agent_kernels.hip; verdict == false only)
    int32_t kin = 1;         // inner iterations, 1..REVS_AGENT_MAX_INNER
    float *pe_out = nullptr; // P_est after the last inner iteration (kin > 1: required)
    float *y_out = nullptr;  // carried PDHG multipliers out (NULL: in place)
    int64_t slice_stride = 0;   // doubles between the node-sum slices of consecutive inner iterations
    int64_t diff_stride = 0;    // floats between their diff rows (0: one row, the last iteration's)
    double *dmax_out = nullptr; // per inner iteration (stride slice_stride): max diff, bits of a double
    const int32_t *wg_order = nullptr;   // NULL, or the residence workgroup each workgroup of the launch takes (verdict == false)
};
int64_t agent_homes_per_block(int32_t T, int32_t lanes);     // residences per workgroup of the sweep's launch

// revs_agent_step_select's sweep with the next home pass folded in, plus `sx` (see above).
int agent_step_stream(int64_t n_homes, int32_t T, const float *cost, const revs_home_t *homes,
                      const float *load, const float *p_est_old, const float *p_est_new,
                      const float *p_sch, const float *gamma, float *p_sch_out, float *gamma_out,
                      float *diff, float *dsq, int32_t *status, float *pdhg_dual, float kappa,
                      int32_t mode, const revs_pdhg_t *pdhg_host, const int32_t *node_of,
                      double *p_next, float *p_est_next, const StreamExtra &sx, void *stream);

// The folded chain's sweep (AgentArgs::sh_a in agent_kernels.hip): the residences' iteration with
// the operator's answer formed inside, pen = max(g0 - d[node], 0), d = sh_a (double[m][T]: the shifts
// R^T y / kappa of the trial multipliers, written by the operator launch before); pe_out receives
// it; fold_a / fold_b (double[T][m][4] = {p, N, -(kappa/2) sum g^2, 0} per slot and node, zero on entry)
// receive the node sums of that evaluation and of the evaluation of the same multipliers on the new state
// (shifts sh_b).
struct ChainFold {
    const double *sh_a, *sh_b;
    int32_t m;
    double kappa;
    double *fold_a, *fold_b;
    float *pe_out;
    float *y_out = nullptr;     // the carried PDHG multipliers after this sweep (NULL: in place)
    const int32_t *wg_order = nullptr;   // as StreamExtra::wg_order
};
int agent_step_chain(int64_t n_homes, int32_t T, const float *cost, const revs_home_t *homes,
                     const float *load, const float *p_est, const float *p_sch, const float *gamma,
                     float *p_sch_out, float *gamma_out, float *s_out, float *c_out, float *diff, float *dsq,
                     int32_t *status, float *pdhg_dual, float kappa, int32_t mode, const revs_pdhg_t *pdhg_host,
                     const int32_t *node_of, const ChainFold &cf, unsigned int *flags, void *stream);

// The folded chain's operator launch (newton_kernels.hip: op_chain_kv_kernel).  One side = one
// evaluation judged by the tree form: its node sums pnq (p | N | q), the multipliers, scratch for
// the rows (vfull, viol, partial -- the two sides run in ONE launch: separate scratch), the
// candidate set and stats block its selection fills, tagged `seq`.  e2 (has_e2): the trial of the
// iteration before -- rows and selection only; e1: rows, selection, small model and step into
// y_trial (lin_out: the linear term, into the trial's stats block).  clr0 / clr1: double[4 m T]
// arrays cleared on the way (NULL: none).
struct ChainKvSide {
    const double *pnq, *y;
    double *vfull, *viol, *partial;
    int64_t *cidx;
    int32_t *ccnt;
    double *cval, *stats;
    double seq;
    // layout of pnq: 1 = planar double[3][m][T] (the evaluation kernel's), 4 = double[T][m][4] = {p, N, q, 0} per
    // slot and node, slot-major (the folded sweep's: a slot's block is contiguous -- the operator launch reads it with
    // coalesced loads in its first round trip and gathers by tree position from LDS)
    int32_t es = 1;
};
struct ChainKv {
    int32_t m, T, kadd, has_e2;
    TreeArgs tree;
    double vlo, vhi, kappa, delta, scale, eps;
    int32_t max_pivots;
    ChainKvSide e1, e2;
    const double *R;
    double *k_full, *yhat;
    int32_t *info;
    double *y_trial, *lin_out;
    double *clr0, *clr1;
    double *sh_a, *sh_b;     // out: the shifts R^T y_trial / kappa (double[m][T]) in list order / in row order
    // the candidate lists of the launch whose step produced e1.y (= e2.y): every row that carries a
    // multiplier is on them, so the slots need not gather the multipliers' columns (NULL: unknown)
    const int64_t *prev_cidx = nullptr;
    const int32_t *prev_ccnt = nullptr;
    // has_e2 launches keep e1's stats in device memory (e1.stats): fwd_src = where the PREVIOUS launch left the stats
    // of the evaluation this launch's verdict belongs to, fwd_dst = the host-visible block the driver reads them from
    // (copied by the verdict half in front of its tag; NULL: that evaluation wrote the host block itself)
    const double *fwd_src = nullptr;
    double *fwd_dst = nullptr;
};
int chain_kv_launch(const ChainKv &c, void *stream);
// revs_op_dual_step with y_trial = y (all rows of every slot) copied in front, one launch (newton_kernels.hip)
int dual_step_copy(int32_t T, const int64_t *cand_idx, const int32_t *cand_cnt, const double *cand_val, const double *yhat,
                   const double *alpha, const double *y, int32_t m, double *y_trial, double *lin_out, void *stream);

// Verdicts by blocks (agent_kernels.hip).  A no-op when a launch numbered base_seq..gate_seq has
// failed its verdict.  Judges `nb` slices of node sums -- `pre` (the caller's array: the sums of
// the call's first iteration; NULL: none) and then ring slices at ring + g * stride, each `mt`
// node sums followed by `ntail` partial maxima of diff left by the sweep that produced the slice
// -- for the iterations first_seq, first_seq + 1, ...; clears every ring slice it has read; with
// hand_over, copies the ring slice after the judged ones (the call's last: its rows are the next
// call's to judge) there, clears it and folds its tail; writes the records {rmax, failed, seq,
// max diff of the iteration before} and the lowest failed number into the control word.
int stream_block_verdict(StreamCtl *ctl, unsigned int base_seq, unsigned int gate_seq,
                         unsigned int first_seq, int32_t nb, int32_t T, const TreeArgs &tree,
                         const double *pre, double *ring, int64_t stride, int32_t mt, int32_t ntail,
                         double *hand_over, double vlo, double vhi, double vtol,
                         unsigned long long *grp_bits, double *grp_dmax, double *rec, void *stream);

}  // namespace revs
