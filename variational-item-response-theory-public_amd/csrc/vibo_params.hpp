// vibo_params.hpp -- host/device shared launch parameters and workspace layout.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace vibo {

constexpr int kTilePersons = 64;   // persons per tile = lanes per wave

// bytes per person row of the fp8 code tile for a workgroup of `waves` waves (I <= 1024 / 512 / 304 / 144):
// an odd multiple of 16 B, so the per-person ds_read_b128 / ds_read_u8 accesses spread over all LDS banks
constexpr int code_tile_stride(int waves) { return waves >= 16 ? 1040 : waves == 8 ? 528 : waves == 4 ? 304 : 144; }

// template ability width for a runtime ability_dim (1,2,4,8)
inline int padded_ability_dim(int a) { return a <= 1 ? 1 : a <= 2 ? 2 : a <= 4 ? 4 : 8; }

// floats per prepped item row (see item_prep_kernel):
//   1PL: [+1'.. (A times), b', 0..]  2PL: [-a'_0..-a'_{AT-1}, b', 0..]
//   3PL: [-a'_0..-a'_{AT-1}, b', guess, 1-guess, 0..]      (x' = x * log2(e): logits in log2 units)
// i.e. logit' = row[0..AT] . [theta_0..theta_{AT-1}, 1]  for every model
inline int prepped_item_width(int irt, int at) {
    if (at == 1) return irt != 3 ? 2 : 4;
    if (at == 2) return irt != 3 ? 4 : 8;
    if (at == 4) return 8;
    return 12;
}

// per-block partial record, in floats:
//   [0,8)  scalars: ll, kl, logq0, logp, ladj, nobs, 0, 0
//   table grads  [set 2][c 2][2*A]          (unconditional posterior)
//   flow grads   [set 2][n_flows][2*A+1]
//   item grads   [D][I_pad]  (structure of arrays; I_pad = 64 * item blocks)
struct PartialLayout {
    int off_table, off_flow, off_item, stride, i_pad;
};
inline PartialLayout partial_layout(int A, int D, int I, int n_flows) {
    PartialLayout L;
    L.i_pad = ((I + 63) / 64) * 64;
    L.off_table = 8;
    L.off_flow = L.off_table + 8 * A;
    L.off_item = L.off_flow + 2 * n_flows * (2 * A + 1);
    L.stride = L.off_item + D * L.i_pad;
    L.stride = (L.stride + 3) & ~3;
    return L;
}

struct ElboParams {
    const float* response;
    const void* mask;
    const int64_t* row_index;
    const float* table;       // [2][2A]
    const float* item_prep;   // [I][DP] prepped item rows (workspace)
    const float* item_raw;    // [I][D] the caller's item sample (the matrix row-split kernel preps its own operands)
    const float* eps;         // [B][A]
    float* ability_mu;
    float* ability_logvar;
    float* ability;
    float* partial;           // [nblk][stride]
    const float* flow;        // [n_flows][2A+1] = uhat | w | b   (row-split kernel only)
    float* ability_k;         // [B][A] sample after the flows     (row-split kernel, n_flows > 0)
    float* ability_ladj;      // [B] sum of log|det| of the flows  (row-split kernel, n_flows > 0)
    // panel mode of the row-split kernel (more than 1024 items): one launch per panel of <= 1024 items
    const int* row_cnt;       // [B] packed counts (n1 << 16 | nobs) over the WHOLE row, or null (single panel)
    int item0;                // first item of this panel
    int I_total;              // items of the whole row (PoE prior experts, nmiss)
    int primary;              // 1: this launch writes the per-person outputs and owns the KL / REG terms
    int panel_count;          // matrix row-split kernel: > 1 = ALL panels in this launch (grid = panel_count x workgroups per panel;
                              // item0 / I / primary / post_coef / the record slot follow from blockIdx.x, see the kernel)
    // conditional posterior (vibo_cond.hip): product-of-experts sums come from cond_pre_kernel, the backward
    // hands per-person coefficients to cond_post_kernel instead of accumulating the 2-row table gradient
    const float* pre_stats;   // [pre_panels][B][2A+1] = lam[A] | s[A] | nobs, or null
    float* post_coef;         // [B][2 sets][2][A]: P1 = gmu / lam, P2 = -(gmu amu + glv) / lam   (this launch's share)
    int pre_panels;
    // VIBO_POSTERIOR_GIVEN, one panel: the caller's posterior read by the slot lanes themselves (lam = exp(-logvar), s = mu lam,
    // nobs = I_total: given_pre_kernel's statements) and its gradient written by them (d/d mu, d/d logvar of the two sets:
    // the posterior IS (mu, logvar) there) -- no pre_stats / post_coef round trip, two launches less per call
    const float* given_post;  // [B][2A] = mu | logvar, or null
    float* given_grad;        // [2 sets][B][2A], or null
    // conditional posterior, one panel, ability_dim 1, fp32 rows: the matrix row-split kernel forms the experts' sums itself (its XM == 3)
    const float* cond_table;  // [2][I_total][2] = mu | logvar per (response code, item), or null
    uint8_t* codes_out;       // [B][codes_stride] the rows' 1-byte cell codes (minibatch order) for the table-gradient pass, or null
    long long codes_stride;
    long long resp_stride, mask_stride;
    int B, I, A, D, DP;
    int n_tiles, lds_stride, lds_main;
    int mask_dtype, missing_mode, reg_mode, vec_ok, n_flows;
    PartialLayout lay;
    int32_t* step_tick;       // non-null: workgroup 0 increments it (the train step's Adam counter, vibo_elbo_fwd_bwd_step)
    unsigned long long* insitu;   // non-null: the in-situ launch timer's eight words (vibo_set_insitu_timer; matrix row-split kernel)
};

// the conditional posterior's table-gradient finalize riding in the ELBO finalize launch (vibo_cond_finalize.hpp)
struct CondFinTail {
    int kind;                 // 0: none, 1: cond_post_kernel's records (VALU pass), 2: cm_backward_kernel's (matrix-pipe pass)
    int gx, gy;               // its grid
    const float* rec;
    const float* table;       // kind 2: the expert table (chain rule through the product of experts)
    float* grad_table;
    int I, A;
    int bpp, rec_stride;      // kind 1
    int nR, N;                // kind 2
    int packed_cols;          // kind 2: > 0 = the records' columns are [piece][packed_cols] (cm_backward_body's PK): the pieces' sums add up
};

struct FinalizeParams {
    const float* partial;
    float* out_scalars;
    float* grad_table;
    float* grad_item;
    float* grad_flow;
    int nblk, I, A, D, n_flows, reg_mode, irt, want_grad;
    int panel_items, bpp;     // item grads of item i live in blocks [i / panel_items * bpp, +bpp) (panel mode)
    PartialLayout lay;
    int n_fin;                // workgroups of the finalize proper; the ones past them run `tail`
    CondFinTail tail;
};

// launch parameters of train_epilogue_fused_kernel (vibo_trainer.hip; host side: vibo_train_epilogue_fused in vibo_capi.hip)
struct EpiParams {
    int H, O, n_item_entries, I, D;
    const float* flat_in;     // person-sharded: the all-reduced [8 scalars | grads]; else == flat_out
    float* flat_out;
    float* saved_h;           // this step's activations in, the next step's out
    float* kl_parts;          // [2][kl_part_count]: double-buffered by step parity
    float* eps_item;          // this step's in, the next step's out (in place)
    const float* beta;
    const float* lr;
    int32_t* step_count;
    float *P, *M, *V, *mu, *lv, *im, *iv, *loss_out;
    const float* partial;     // non-null: the ELBO kernel's partial records (fused finalize)
    int nblk;
    PartialLayout lay;
    float* item_feat;         // out: the next step's item sample
    float* table;             // out: the next step's expert table
    uint32_t seed_lo, seed_hi;
    float* eps_ab;            // out: the next step's ability noise [n_ab]
    long long n_ab;
    uint32_t ab_stream;
    int n_item_blocks;
};

}  // namespace vibo
