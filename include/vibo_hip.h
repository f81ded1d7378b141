/*
 * vibo_hip.h -- C ABI of the MI355X-native VIBO ELBO engine (libvibo_hip.so).
 *
 * The reference (mhw32/variational-item-response-theory-public) is pure Python
 * and has no FFI of its own; the seam this library plugs into is the method
 * surface of VIBO_1PL/2PL/3PL that src/torch_core/vibo.py calls (SURVEY.md
 * §8b).  Each entry point below names the reference code it replaces.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer unless it
 *     says "host"; the caller owns every buffer including the workspace; the
 *     library allocates nothing and keeps no state between calls;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*); no host
 *     synchronisation inside; safe to capture into a hipGraph;
 *   - outputs are fully overwritten (never accumulated into);
 *   - return value: 0 ok; <0 invalid argument / unsupported combination (see
 *     vibo_last_error_string); >0 a hipError_t from the launch.  No C++
 *     exception crosses the boundary.
 *   - all floating point is IEEE fp32 (the reference computes in fp32).
 */
#ifndef VIBO_HIP_H
#define VIBO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIBO_ABI_VERSION 2

enum { VIBO_IRT_1PL = 1, VIBO_IRT_2PL = 2, VIBO_IRT_3PL = 3 };
enum { VIBO_POSTERIOR_UNCONDITIONAL = 0, VIBO_POSTERIOR_CONDITIONAL = 1,
       VIBO_POSTERIOR_GIVEN = 2 };   /* q(theta_p) = N(mu_p, exp(logvar_p)) computed by the caller: `table` is [B][2A]
                                        (mu | logvar per person) and `grad_table` comes back as [2][B][2A].  This is
                                        the --ability-merge mean encoder (models.py:584-594, 631-650): its per-person
                                        input is the mean of two feature vectors weighted by the counts of
                                        vibo_row_counts, followed by a dense [B,H]x[H,H] MLP that stays with the
                                        caller's GEMM library.  Row-split path only (4..32767 items, chunkable rows). */
enum { VIBO_MISSING_PRIOR = 0,   /* missing cell -> N(0,1) prior expert (models.py:613-620) */
       VIBO_MISSING_DROP = 1 };  /* --drop-missing: expert removed      (vibo.py:217)       */
enum { VIBO_MASK_U8 = 0,         /* torch.bool / uint8, 1 byte per cell (datasets.py:938)   */
       VIBO_MASK_I64 = 1,        /* mask.long() as the reference train loop passes it (vibo.py:240) */
       VIBO_MASK_NONE = 2,       /* mask == NULL: every cell observed                        */
       VIBO_MASK_CODES = 3 };    /* "Format P" (SURVEY 8f-3): `mask` points at one byte per cell holding the whole cell,
                                    0 = answered wrong, 1 = answered right, 2 = missing; `response` is ignored (NULL
                                    allowed) and mask_row_stride is the stride of the code rows.  1 B/cell of HBM
                                    traffic instead of 5.  Produced by vibo_pack_codes from the reference's layout
                                    (fp32 responses with -1 for missing + a separate mask, datasets.py:928-940).
                                    Row-split paths only: 4..32767 items, code rows 4-byte aligned with a stride that
                                    is a multiple of 4 (cells past num_item are ignored).  Anything else returns -8.  */
enum { VIBO_REG_KL = 0,          /* elbo(use_kl_divergence=True): analytic KL (models.py:427-430) */
       VIBO_REG_SAMPLED = 1 };   /* use_kl_divergence=False / flows: log q - log p at the sample
                                    (models.py:406-424, 432-441)                             */

#define VIBO_MAX_ABILITY_DIM 8          /* every row-split / matrix-pipe / trainer path */
#define VIBO_MAX_ABILITY_DIM_WIDE 16    /* vibo_elbo_fwd_bwd / vibo_encode / vibo_decode[_mean]: ability_dim 9..16 run on the
                                           wave-per-person kernel (fp32 rows, u8 / int64 / no mask; atomically reduced: not
                                           bitwise reproducible, ~30x below the row-split kernels) */
#define VIBO_MAX_FLOWS 8
#define VIBO_NUM_SCALARS 8
/* indices into out_scalars */
enum { VIBO_S_LL = 0,        /* sum_{p,i} mask * log Bernoulli(response | p_pi)   (utils.py:46-49)   */
       VIBO_S_REG = 1,       /* KL mode: sum_p KL(q(theta_p) || N(0,1))            (utils.py:85-88)
                                sampled:  sum_p [log q0(theta_0) - ladj - log p(theta_K)]            */
       VIBO_S_KL = 2,        /* sum_p KL, always                                                    */
       VIBO_S_LOGQ0 = 3,     /* sum_p log N(theta_0; mu_p, var_p)                 (utils.py:59-61)   */
       VIBO_S_LOGP = 4,      /* sum_p log N(theta_K; 0, 1)                        (utils.py:64-67)   */
       VIBO_S_LADJ = 5,      /* sum_p sum_k log|det J_k|                          (flows.py:39)      */
       VIBO_S_NOBS = 6,      /* number of observed cells                                             */
       VIBO_S_RESERVED = 7 };

/* Problem descriptor (host memory, POD).  B persons of this call x I items. */
typedef struct vibo_desc {
    int32_t abi_version;      /* VIBO_ABI_VERSION */
    int32_t num_person;       /* B: rows processed by this call */
    int32_t num_item;         /* I */
    int32_t ability_dim;      /* A, 1..VIBO_MAX_ABILITY_DIM_WIDE (9..16: see above)  */
    int32_t irt_model;        /* VIBO_IRT_*; item_feat_dim D = 1 | A+1 | A+2        */
    int32_t posterior;        /* VIBO_POSTERIOR_*                                    */
    int32_t missing_mode;     /* VIBO_MISSING_*                                      */
    int32_t mask_dtype;       /* VIBO_MASK_*                                         */
    int32_t reg_mode;         /* VIBO_REG_* (must be SAMPLED when n_flows > 0)       */
    int32_t n_flows;          /* planar flows on the ability sample, 0..VIBO_MAX_FLOWS */
    int32_t want_grad;        /* 0: forward heads only; 1: also gradients            */
    int32_t deterministic;    /* 1: fixed-order reductions (bitwise reproducible)    */
    int64_t response_row_stride; /* elements between consecutive rows of `response`  */
    int64_t mask_row_stride;     /* elements between consecutive rows of `mask`      */
    int32_t flags;            /* VIBO_FLAG_* (0 = let the planner choose)            */
    int32_t reserved;         /* 0                                                   */
} vibo_desc;

/* vibo_desc.flags: planner overrides (A/B measurements, tests that pin one kernel).  The planner reads nothing but the
 * descriptor -- no environment variables, no state between calls. */
enum { VIBO_FLAG_KERNEL_VALU = 1,     /* row-split work goes to the VALU kernel (vibo_split_kernel.hpp)              */
       VIBO_FLAG_KERNEL_MATRIX = 2,   /* ... to the matrix-pipe kernel (vibo_msplit_kernel.hpp) whatever the size    */
                                      /* (either flag also keeps narrow rows -- 4..128 items, ability_dim <= 4, plain model -- off the
                                       *  narrow-row kernel, vibo_narrow.hip, which the planner picks for them otherwise)            */
       VIBO_FLAG_NO_EMIT_CODES = 4,   /* multi-pass paths re-read the fp32 rows instead of the first pass's cell codes */
       VIBO_FLAG_COND_VALU = 8,       /* conditional posterior: the VALU passes (vibo_cond.hip) instead of the matrix-pipe ones */
       VIBO_FLAG_COND_MATRIX = 16,    /* ... the matrix-pipe passes (vibo_cmean.hip) whatever the size, wherever the rows allow  */
       VIBO_FLAG_COND_THREE_PASS = 32 };  /* conditional posterior at ability_dim 1: keep the separate first pass (cond_pre) instead of
                                         folding it into the matrix row-split kernel */

/* Which fused kernel vibo_elbo_fwd_bwd would launch for `d` (pointers assumed aligned): VIBO_KERNEL_*, or <0 on a bad
 * descriptor.  Lets a benchmark state which kernel its numbers belong to. */
enum { VIBO_KERNEL_MATRIX = 1,        /* msplit_kernel: contractions as f16 hi/lo MFMAs (default above 2 048 persons) */
       VIBO_KERNEL_VALU = 2,          /* split_kernel: VALU row-split kernel                                          */
       VIBO_KERNEL_ROW = 3,           /* wave-per-row kernel (int64 masks)                                            */
       VIBO_KERNEL_TILED = 4,         /* tiled fp32-MFMA fallback (ragged rows)                                       */
       VIBO_KERNEL_GENERAL = 5,       /* wave-per-person fallback                                                     */
       VIBO_KERNEL_NARROW = 6 };      /* narrow_kernel: 4..128 items, ability_dim <= 4, plain model: a row per 16 lanes */
int vibo_plan_kernel(const vibo_desc* d);
/* Conditional posterior: which of its two extra passes would run on the matrix pipe (csrc/vibo_cmean.hip) for `d` -- bit 0 the
 * experts' per-person sums, bit 1 the scatter of the table gradient; bit 2: there is no separate first pass at all -- the matrix
 * row-split kernel forms the experts' sums itself (ability_dim 1, at most 1024 items, fp32 rows; VIBO_FLAG_COND_THREE_PASS clears it);
 * 0 = both on the VALU kernels (csrc/vibo_cond.hip, or not a conditional-posterior call), <0 on a bad descriptor. */
int vibo_plan_cond_passes(const vibo_desc* d);

/* Library / ABI version (VIBO_ABI_VERSION of the build). */
int vibo_version(void);

/* Message for the last non-zero return on this thread (host pointer, static storage). */
const char* vibo_last_error_string(void);

/*
 * Measurement hook: the fused kernel's own duration, measured by the kernel (no reference counterpart; bench.py's
 * `roofline.kernel_ms_insitu`).  HIP events cannot be recorded inside a replayed hipGraph and a tracer changes what it traces;
 * with a timer block set, every matrix row-split or narrow-row launch (VIBO_KERNEL_MATRIX, VIBO_KERNEL_NARROW) enqueued FROM THIS HOST THREAD afterwards -- eagerly
 * or while a stream capture records it into a graph -- stamps its earliest workgroup entry and its latest workgroup exit on the
 * chip-wide 100 MHz clock (s_memrealtime) and the last workgroup to leave adds the difference to the block:
 *   block[3] sum of the launches' durations | block[4] launches | block[5] shortest | block[6] longest | block[7] the last one
 *   (ticks of 10 ns; block[0..2] are the launch in flight).  `block` = 8 uint64 of device memory the caller owns, armed by
 * vibo_insitu_timer_reset (two byte-fills on `stream`) and read back by the caller after synchronising.  NULL switches the hook off
 * (the default).  The pointer is thread-local host state -- the library's only one besides the error string; the data path never
 * reads it.  Cost: two device-scope atomics per workgroup and launch.  Launches that share a block must not overlap in time.
 */
int vibo_set_insitu_timer(uint64_t* block);
int vibo_insitu_timer_reset(uint64_t* block, void* stream);

/* Self-test of the kernels' cross-row sums (csrc/vibo_device.hpp: xor16_add / xor32_add = v_permlane16_swap / v_permlane32_swap
 * through inline asm) against the __shfl_xor form they replace: `in` = 64 floats (one per lane of a wave), `out` = 6 x 64 floats:
 * [0] swap form of v + v[lane ^ 16], [1] of v + v[lane ^ 32], [2] [3] the same two sums by __shfl_xor, [4] the chained swap form
 * (16 then 32) as the kernels use it, [5] the chained shuffle form.  Rows 0/2, 1/3 and 4/5 must agree bit for bit. */
int vibo_selftest_lane_swaps(const float* in, float* out, void* stream);

/* Workspace bytes vibo_elbo_fwd_bwd / vibo_encode need for `d` (0 on bad desc). */
size_t vibo_workspace_bytes(const vibo_desc* d);

/*
 * Fused ELBO step over a minibatch of B persons: product-of-experts ability
 * posterior, reparameterised sample (+ planar flows), 1PL/2PL/3PL link, masked
 * Bernoulli log-likelihood and the ability-side regulariser, with the full
 * backward pass in the same sweep (each response row is read from HBM once).
 *
 * Replaces, on the reference hot path (vibo.py:237-268):
 *   AbilityInferenceNetwork / ConditionalAbilityInferenceNetwork.forward
 *                                   (models.py:596-629, 652-661, 695-710)
 *   product_of_experts              (utils.py:105-113)
 *   reparameterize_gaussian         (models.py:506-510; eps supplied by caller)
 *   NormalizingFlows.forward on the ability sample (flows.py:21-41, 58-66)
 *   irt_model_1pl/2pl/3pl, decode   (models.py:729-766, 373-378)
 *   masked_bernoulli_log_pdf(...).sum(), kl_divergence_standard_normal_prior,
 *   normal_log_pdf / standard_normal_log_pdf of the ability
 *                                   (models.py:399, 412-418, 428, 433-435)
 *   and autograd's backward of all of the above (vibo.py:267).
 *
 * The per-(person,item) encoder MLP is NOT evaluated per cell: Bernoulli
 * responses take two observed values, so the caller passes the encoder's
 * outputs for inputs 0 and 1 as `table` and receives d/d table; the (tiny) MLP,
 * the item-side reparameterisation / flows / KL and the optimizer stay with the
 * caller.
 *
 *  response   [B rows x I] fp32, 1.0 = correct, 0.0 = wrong (missing cells: any value)
 *  mask       [B rows x I] u8 or i64 per d->mask_dtype, nonzero = observed; NULL iff MASK_NONE
 *  row_index  [B] int64 rows of response/mask to process, or NULL for rows 0..B-1
 *  table      unconditional: [2][2A]    (row c = encoder([c]):      mu[0..A) | logvar[A..2A))
 *             given:         [B][2A]    (the caller's posterior of every person of this call)
 *             conditional:   [2][I][2A] (entry = encoder([c, item_i]))
 *  item       [I][D] item sample d_i (after item flows if any): 2PL/3PL cols 0..A-1
 *             discrimination, col A difficulty, col A+1 guess logit (3PL); 1PL col 0 difficulty
 *  eps        [B][A] standard-normal draws for the ability reparameterisation
 *  flow       [n_flows][2A+1]: uhat[A] | w[A] | b  (uhat = flows.py:23-25, computed by caller); NULL if none
 *
 *  out_scalars    [VIBO_NUM_SCALARS] fp32, see VIBO_S_*
 *  ability_mu     [B][A]   posterior mean          (models.py:364-366)
 *  ability_logvar [B][A]   posterior log-variance
 *  ability        [B][A]   theta_0 = mu + exp(.5 logvar) * eps
 *  ability_k      [B][A]   theta after flows (NULL allowed when n_flows == 0)
 *  ability_ladj   [B]      sum_k log|det J_k|  (NULL allowed when n_flows == 0)
 *  grad_table     [2] x table-shape: [0] = d LL / d table, [1] = d REG / d table
 *  grad_item      [I][D]   d LL / d item
 *  grad_flow      [2][n_flows][2A+1]: d LL / d flow, d REG / d flow (NULL if no flows)
 *                 (grad_* may be NULL when d->want_grad == 0)
 *
 * The caller composes  loss = -LL + beta * REG + (item-side terms)  and the
 * parameter gradients  -grad[0] + beta * grad[1]  (models.py:427-443).
 */
int vibo_elbo_fwd_bwd(const vibo_desc* d,
                      const float* response, const void* mask, const int64_t* row_index,
                      const float* table, const float* item, const float* eps,
                      const float* flow,
                      float* out_scalars,
                      float* ability_mu, float* ability_logvar, float* ability,
                      float* ability_k, float* ability_ladj,
                      float* grad_table, float* grad_item, float* grad_flow,
                      void* workspace, size_t workspace_bytes, void* stream);

/*
 * vibo_elbo_fwd_bwd with the rows' whole-row counts supplied: row_counts[r] = vibo_row_counts of SOURCE row r (the row a call
 * without row_index reads as its row r, the row row_index[k] = r selects) -- statistics of the data alone, which a caller with a
 * resident matrix computes once.  The paths that would count first (unconditional posterior, more than 1024 items: a 5 B/cell
 * pass in front of the panels) skip that pass; every other path ignores the argument.  Results are those of vibo_elbo_fwd_bwd bit
 * for bit.
 */
int vibo_elbo_fwd_bwd_counts(const vibo_desc* d,
                             const float* response, const void* mask, const int64_t* row_index, const int32_t* row_counts,
                             const float* table, const float* item, const float* eps,
                             const float* flow,
                             float* out_scalars,
                             float* ability_mu, float* ability_logvar, float* ability,
                             float* ability_k, float* ability_ladj,
                             float* grad_table, float* grad_item, float* grad_flow,
                             void* workspace, size_t workspace_bytes, void* stream);

/*
 * Forward-only ability posterior q(theta | responses[, items]) for B persons
 * (model.encode under no_grad: vibo.py:363-364, 406-407, 434-435; models.py:356-371).
 * Arguments as above; writes ability_mu / ability_logvar [B][A].  With a workspace of vibo_workspace_bytes(d) and
 * 16-byte chunkable rows it runs as a row-statistics pass at HBM speed + a per-person finish; without (workspace
 * null / too small, int64 masks, ...) as one wave per person.
 */
int vibo_encode(const vibo_desc* d,
                const float* response, const void* mask, const int64_t* row_index,
                const float* table,
                float* ability_mu, float* ability_logvar,
                void* workspace, size_t workspace_bytes, void* stream);

/*
 * counts[p] = n_correct << 16 | n_observed over the whole row of person p (any row layout / mask dtype, up to 32767
 * items): the sufficient statistics of a Bernoulli response row for the unconditional encoders -- the product of
 * experts (models.py:596-629) and the masked mean of --ability-merge mean (models.py:631-650: observed cells only).
 */
int vibo_row_counts(const vibo_desc* d, const float* response, const void* mask, const int64_t* row_index, int32_t* counts,
                    void* stream);

/*
 * The --ability-merge mean encoder per person (models.py:584-594, 631-650).  With Bernoulli responses the mean of the
 * per-term features over a person's observed items is h0 + w (h1 - h0), w = n_correct / n_observed (counts from
 * vibo_row_counts), so the first layer of mlp2 is affine in w:  z = u + w v  with the two H-vectors
 * u = W1 h0 + b1, v = W1 (h1 - h0) the caller forms from mlp1 / mlp2[0] (autograd on 2 x H numbers).  These entry points
 * do the per-person rest:  posterior[p] = W2 elu(u + w_p v) + b2  = (mu | logvar) [2A]   (W2 = mlp2[2].weight [2A][H]),
 * and its backward: given d loss / d posterior [B][2A] (what vibo_elbo_fwd_bwd returns in VIBO_POSTERIOR_GIVEN mode,
 * combined by the caller), `partials` receives n_partials records [ d/du (H) | d/dv (H) | d/dW2 (2A x H) | d/db2 (2A) ]
 * that the caller sums (fixed order).  n_partials = vibo_mean_encoder_partials(d); hidden <= 256.
 * The forward runs a wave per person up to 2 048 persons (sums over the hidden units as wave totals) and a thread per person
 * beyond: a call is bit-identical to itself, calls on either side of that size agree to fp32 rounding.
 */
int vibo_mean_encoder_partials(const vibo_desc* d);
int vibo_mean_encoder_forward(const vibo_desc* d, int hidden, const int32_t* counts, const float* u, const float* v,
                              const float* w2, const float* b2, float* posterior, void* stream);
int vibo_mean_encoder_backward(const vibo_desc* d, int hidden, const int32_t* counts, const float* u, const float* v,
                               const float* w2, const float* grad_posterior, float* partials, int n_partials, void* stream);

/*
 * Repack B rows from the reference's layout (fp32 `response`, `mask` per d->mask_dtype U8 / I64 / NONE, strides from d)
 * into 1-byte cell codes for VIBO_MASK_CODES: codes[row * codes_row_stride + i], i < num_item; the cells between
 * num_item and codes_row_stride are written as "missing".  One pass, done once for a device-resident dataset
 * (replaces the per-step `.float()` / `.long()` conversions of vibo.py:239-240 for good).
 */
int vibo_pack_codes(const vibo_desc* d, const float* response, const void* mask, uint8_t* codes, int64_t codes_row_stride,
                    void* stream);

/*
 * decode(): P(response = 1) for every (person, item) -> response_mu [B][I]
 * (models.py:373-378, 529-533, 544-548; callers vibo.py:379, 409).
 *  ability [B][A], item [I][D].
 */
int vibo_decode(const vibo_desc* d, const float* ability, const float* item,
                float* response_mu, void* stream);

/*
 * Posterior-predictive mean: the mean over num_samples posterior draws of decode() (the S decode calls of
 * sample_posterior_predictive, vibo.py:363-390, reduced the way the imputation-accuracy step consumes them,
 * vibo.py:504-548) without materialising the [S][B][I] stack:
 *     response_mu_mean[b][i] = (1/S) sum_s P(response = 1 | ability[s][b], item[s][i])
 *  ability [S][B][A], item [S][I][D], response_mu_mean [B][I].
 */
int vibo_decode_mean(const vibo_desc* d, int num_samples, const float* ability, const float* item,
                     float* response_mu_mean, void* stream);


/*
 * Fused O(I) part of one VIBO train step (unconditional posterior, no flows), for trainers that want the whole
 * step as ~5 kernels instead of ~80 tiny PyTorch launches.  Semantics = the PyTorch statements they replace:
 *
 * vibo_train_prologue   (models.py:356-361, 575-582, 713-726, 506-510; utils.py:85-88)
 *     step_count += 1
 *     item_feat = item_mu + exp(0.5 * item_logvar) * eps_item                       [I][D]
 *     kl_parts[g] = partial sums of  -0.5 (1 + logvar - mu^2 - exp(logvar)), one per 64 entries   (summed by the epilogue)
 *     table[c] = W2 . elu(W1 . elu(W0 * c + b0) + b1) + b2   for c in {0,1}          [2][2A]
 *     saved activations h1, h2 [2][H]
 * vibo_train_epilogue   (models.py:427-443; vibo.py:267-268 = loss.backward(); optimizer.step())
 *     loss = -LL + beta * (REG + KL_item)                with LL, REG, d/dtable, d/ditem read from `flat`
 *            (the buffer vibo_elbo_fwd_bwd filled: [8 scalars | grad_table[2][2][2A] | grad_item[I][D]])
 *     gradients of the encoder MLP (2-row backward by hand) and of item_mu / item_logvar, immediately
 *     applied with Adam (betas 0.9/0.999, eps 1e-8, no weight decay, torch.optim.Adam's update formula).
 *     Parameters and Adam moments are updated IN PLACE.
 *
 *  mlp_params / adam_m / adam_v: one flat fp32 buffer each laid out  W0[H] | b0[H] | W1[H][H] | b1[H] | W2[2A][H] | b2[2A]
 *  item_mu, item_logvar, item_m/v (mu then logvar): [I][D] each
 *  beta, lr: device scalars (fp32) so that an annealing schedule stays hipGraph-capturable
 *  step_count: device int32[2].  [0] = Adam's step number t, incremented by the prologue and read by the epilogue;
 *              [1] = number of COMPLETED steps, incremented by the epilogue: the step component of the noise counters
 *              (vibo_train_prologue_noise below, or vibo_fill_normal called with step_count + 1)
 * vibo_train_prologue_noise = vibo_train_prologue that draws its own reparameterisation noise in the same launch:
 *     eps_item [I][D]  <- stream 0, eps_ability [B][A] <- stream `ability_stream_id`, both exactly the values
 *     vibo_fill_normal(out, n, seed, step_count + 1, stream_id) produces (torch.randn_like in models.py:506-510);
 *     two launches fewer per step, which is a quarter of a step at the reference's default batch size of 16.
 *  kl_parts: workspace of at least ceil(I*D/64) floats;  hidden_dim H <= 256
 */
int vibo_train_prologue(const vibo_desc* d, int hidden_dim, const float* mlp_params, const float* item_mu,
                        const float* item_logvar, const float* eps_item, float* item_feat, float* table,
                        float* saved_h, float* kl_parts, int32_t* step_count, void* stream);

int vibo_train_prologue_noise(const vibo_desc* d, int hidden_dim, const float* mlp_params, const float* item_mu,
                              const float* item_logvar, float* eps_item, float* item_feat, float* table,
                              float* saved_h, float* kl_parts, int32_t* step_count, uint64_t seed, float* eps_ability,
                              uint32_t ability_stream_id, void* stream);
int vibo_train_epilogue(const vibo_desc* d, int hidden_dim, const float* flat, const float* saved_h,
                        const float* kl_parts, const float* eps_item, const float* beta, const float* lr,
                        const int32_t* step_count, float* mlp_params, float* mlp_m, float* mlp_v,
                        float* item_mu, float* item_logvar, float* item_m, float* item_v, float* loss_out,
                        void* stream);

/*
 * The folded train step of the plain model (unconditional posterior, no flows): TWO launches per step instead of four.
 * The step is software-pipelined across its own iterations -- the epilogue of step t leaves everything the ELBO kernel of
 * step t + 1 reads (noise, item sample, item KL parts, expert table) in memory:
 *
 * vibo_elbo_fwd_bwd_step = vibo_elbo_fwd_bwd (table / item as left behind by the previous epilogue, or by vibo_train_prime)
 *     that also ticks step_count[0] += 1 (Adam's t; what vibo_train_prologue does in the four-launch form) and, with
 *     skip_finalize != 0, leaves the per-workgroup partial records in `workspace` (out_scalars / grad_* are not written)
 *     for vibo_train_epilogue_fused.  vibo_train_step_supported(d): bit 0 = this descriptor can take the folded step
 *     (single-launch row-split path: 4..1024 items, chunkable rows, no int64 mask, KL regulariser, gradients; otherwise the
 *     call returns -8), bit 1 = skip_finalize too.
 * vibo_train_epilogue_fused = [finalize] + vibo_train_epilogue for THIS step + vibo_train_prologue_noise for the NEXT one.
 *     workspace != NULL: the workspace a vibo_elbo_fwd_bwd_step(skip_finalize) call with the same descriptor just filled;
 *         the fixed-order sums over its partial records happen here (same order as the stand-alone finalize: bit-identical)
 *         and are also written to `flat` ([8 scalars | grad_table[2][2][2A] | grad_item[I][D]]).
 *     workspace == NULL: `flat` already holds the sums (person-sharded: finalize ran before the all-reduce).
 *     Then, from the updated parameters: eps_item (in place) and eps_ability[0 .. n_eps_ability) are redrawn with the
 *     vibo_fill_normal streams 0 / ability_stream_id at counter step_count[0] (= what the next step's
 *     vibo_train_prologue_noise would draw), item_feat / the next half of kl_parts / table / saved_h are recomputed --
 *     the same statements in the same order as vibo_train_prologue: the two forms of the step agree bit for bit.
 *     kl_parts: [2][ceil(I*D/64)] floats, double-buffered by the parity of step_count[0].  step_count[1] += 1.
 * vibo_train_prime = the head of the FIRST folded step (and of the first one after the parameters were changed from
 *     outside): vibo_train_prologue without the tick, writing the kl_parts half the coming step reads.  The caller fills
 *     eps_item / eps_ability with vibo_fill_normal(step_count + 1) before it.
 */
int vibo_train_step_supported(const vibo_desc* d);
int vibo_elbo_fwd_bwd_step(const vibo_desc* d, int32_t* step_count, int skip_finalize, const float* response, const void* mask,
                           const int64_t* row_index, const float* table, const float* item, const float* eps, float* out_scalars,
                           float* ability_mu, float* ability_logvar, float* ability, float* grad_table, float* grad_item,
                           void* workspace, size_t workspace_bytes, void* stream);
int vibo_train_epilogue_fused(const vibo_desc* d, int hidden_dim, const void* workspace, float* flat, float* saved_h,
                              float* kl_parts, float* eps_item, const float* beta, const float* lr, int32_t* step_count,
                              float* mlp_params, float* mlp_m, float* mlp_v, float* item_mu, float* item_logvar, float* item_m,
                              float* item_v, float* loss_out, uint64_t seed, float* item_feat, float* table, float* eps_ability,
                              int64_t n_eps_ability, uint32_t ability_stream_id, void* stream);
int vibo_train_prime(const vibo_desc* d, int hidden_dim, const float* mlp_params, const float* item_mu,
                     const float* item_logvar, const float* eps_item, float* item_feat, float* table,
                     float* saved_h, float* kl_parts, int32_t* step_count, void* stream);

/*
 * The same two halves of a train step for --conditional-posterior and / or --n-norm-flows models (product-of-experts
 * encoder, IRT decoder; vibo.py:243-268 with models.py:337-354, 380-443, 664-710 and flows.py:21-66):
 *
 * vibo_ctrain_prologue   item_feat = item_mu + exp(.5 item_logvar) * eps_item          (models.py:361-362, 506-510)
 *                        item_k    = item-side planar flows of item_feat                (models.py:346-348; = item_feat without flows)
 *                        flow_packed [F][2A+1] = uhat | w | b of the ability-side flows (flows.py:23-26), the `flow` input
 *                                    of vibo_elbo_fwd_bwd
 *                        table     = encoder MLP on the rows [c, item_feat_i] -> [2][I][2A]  (conditional posterior,
 *                                    models.py:695-710) or on [c] -> [2][2A]
 *                        draw_noise != 0: eps_item / eps_ability are WRITTEN with the vibo_fill_normal streams 0 /
 *                                    ability_stream_id of step step_count[1] (as vibo_train_prologue_noise does); else eps_item is read
 * vibo_ctrain_epilogue   loss (KL mode: -LL + beta (REG + KL_item), models.py:427-430; flows: -(LL + log p(d_K) - REG -
 *                        log q(d_K)), models.py:406-424, annealing ignored), then the backward through the table MLP
 *                        (the conditional encoder's input gradient reaches the item sample), the item-side flows and
 *                        the sample, and Adam on every parameter, IN PLACE (torch.optim.Adam's update formula).
 *                        `flat` = the buffer vibo_elbo_fwd_bwd filled with `table` / item = item_k / flow = flow_packed:
 *                        [8 scalars | grad_table x 2 | grad_item [I][D] | grad_flow x 2].
 *
 *  params / adam_m / adam_v: one flat fp32 buffer each,
 *      W0 [H][xin] | b0 [H] | W1 [H][H] | b1 [H] | W2 [2A][H] | b2 [2A] | ability flows F x (u[A] | w[A] | b) | item flows F x (u[D] | w[D] | b)
 *      with xin = 1 + D (conditional) or 1; vibo_ctrain_param_floats(d, H) floats.  hidden_dim H <= 64 (narrower widths run zero-padded on the 64-wide matrix-pipe tile).
 *  scratch: vibo_ctrain_scratch_floats(d, H) floats, handed to both calls of a step unchanged in between.
 *  step_count, beta, lr, item_m / item_v: as for vibo_train_prologue / vibo_train_epilogue.
 * Every reduction is a fixed-order sum of per-workgroup records: bitwise reproducible, hipGraph-capturable (no host sync).
 */
int64_t vibo_ctrain_param_floats(const vibo_desc* d, int hidden_dim);
int64_t vibo_ctrain_scratch_floats(const vibo_desc* d, int hidden_dim);
int vibo_ctrain_prologue(const vibo_desc* d, int hidden_dim, const float* params, const float* item_mu, const float* item_logvar,
                         float* eps_item, uint64_t seed, int draw_noise, float* eps_ability, uint32_t ability_stream_id,
                         float* item_feat, float* item_k, float* table, float* flow_packed, float* scratch, int32_t* step_count,
                         void* stream);
int vibo_ctrain_epilogue(const vibo_desc* d, int hidden_dim, const float* flat, const float* eps_item, const float* item_feat,
                         const float* item_k, const float* beta, const float* lr, int32_t* step_count, float* params, float* adam_m,
                         float* adam_v, float* item_mu, float* item_logvar, float* item_m, float* item_v, float* scratch,
                         float* loss_out, void* stream);

/*
 * The O(I) + O(1) part of a train step of the --ability-merge mean encoder (unconditional posterior, IRT decoder, no flows;
 * models.py:584-594, 631-650; vibo.py:243-268) -- what vibo_train_prologue / vibo_train_epilogue are for the product-of-experts
 * encoder.  A step is
 *     vibo_mtrain_prologue      step_count[0] += 1; item sample, item KL parts (+ the Philox noise with draw_noise, as
 *                               vibo_train_prologue_noise); the 2-row mlp1 forward and the u, v vectors of vibo_mean_encoder_forward
 *                               (uv = u [H] | v [H]); saved = the activations the backward needs (4 H floats)
 *     vibo_mean_encoder_forward (counts of the minibatch's rows, u, v, W22, b22) -> posterior [B][2A]
 *     vibo_elbo_fwd_bwd         in VIBO_POSTERIOR_GIVEN mode (table = that posterior) -> flat = [8 scalars | grad_table [2][B][2A] | grad_item]
 *     vibo_mean_encoder_backward_sets   = vibo_mean_encoder_backward on d loss / d posterior = -grad_table[0] + beta grad_table[1]
 *                               (the combination done in the kernel: no [B][2A] temporary)
 *     vibo_mtrain_epilogue      fixed-order sums of those per-wave records (grad_sums: 2H + 2A H + 2A floats of scratch), loss =
 *                               -LL + beta (REG + KL_item), the backward through u, v, mlp2[0] and the 2-row mlp1 by hand, Adam
 *                               on every parameter IN PLACE (torch.optim.Adam's update), item backward + Adam; step_count[1] += 1
 *  params / adam_m / adam_v: one flat fp32 buffer each,
 *      mlp1[0].weight [H] | .bias [H] | mlp1[2].weight [H][H] | .bias [H] | mlp2[0].weight [H][H] | .bias [H] | mlp2[2].weight [2A][H] | .bias [2A]
 *      (vibo_mtrain_param_floats(d, H) floats); hidden_dim H <= 128; d->posterior must be VIBO_POSTERIOR_GIVEN.
 *  kl_parts: ceil(I*D/64) floats; the other arguments as for vibo_train_prologue / vibo_train_epilogue.
 */
int64_t vibo_mtrain_param_floats(const vibo_desc* d, int hidden_dim);
int vibo_mtrain_prologue(const vibo_desc* d, int hidden_dim, const float* params, const float* item_mu,
                         const float* item_logvar, float* eps_item, uint64_t seed, int draw_noise, float* eps_ability,
                         uint32_t ability_stream_id, float* item_feat, float* uv, float* saved, float* kl_parts,
                         int32_t* step_count, void* stream);
int vibo_mean_encoder_backward_sets(const vibo_desc* d, int hidden, const int32_t* counts, const float* u, const float* v,
                                    const float* w2, const float* grad_sets, const float* beta, float* partials,
                                    int n_partials, void* stream);
int vibo_mtrain_epilogue(const vibo_desc* d, int hidden_dim, const float* flat, const float* partials, int n_partials,
                         float* grad_sums, const float* saved, const float* kl_parts, const float* eps_item,
                         const float* beta, const float* lr, int32_t* step_count, float* params, float* adam_m,
                         float* adam_v, float* item_mu, float* item_logvar, float* item_m, float* item_v, float* loss_out,
                         void* stream);

/*
 * --ability-merge mean WITH --conditional-posterior (models.py:664-710 with _forward_mean :631-650): the per-term feature
 * depends on the item, feature[c][i][:] = elu(mlp1([c, item_i])) (the caller's 2 x I-row MLP), and the encoder's input is its mean
 * over a person's observed cells.  The sum over the cells -- one-hot(codes) [B, 2I] x feature [2I, H], the path's one dense
 * encoder contraction -- and its transpose for the backward run on the matrix pipe from the 1-byte cell codes
 * (VIBO_MASK_CODES layout: 0 wrong / 1 right / 2 missing; rows 4-byte aligned, stride % 4 == 0), the dense operand as
 * hi + lo f16 pieces with fp32 accumulation (fp32-grade):
 *     vibo_code_table_sum_forward    out_sum [B][H]          = sum_i [observed] feature[code_pi][i][:]
 *     vibo_code_table_sum_backward   grad_feature [2][I][H]  = sum_p [code_pi == c] grad_sum[p][:]      (fixed-order: reproducible)
 * H = 64.  scratch: vibo_code_table_scratch_bytes(B, I, H) bytes, 256-byte aligned, need not survive between the two calls.
 * The division by the number of observed cells (vibo_row_counts) and mlp2 -- a plain [B, 64] x [64, 64] GEMM -- stay with the caller.
 */
size_t vibo_code_table_scratch_bytes(int64_t num_person, int num_item, int hidden_dim);
int vibo_code_table_sum_forward(int64_t num_person, int num_item, int hidden_dim, const uint8_t* codes, int64_t codes_row_stride,
                                const float* feature, float* out_sum, void* scratch, size_t scratch_bytes, void* stream);
int vibo_code_table_sum_backward(int64_t num_person, int num_item, int hidden_dim, const uint8_t* codes, int64_t codes_row_stride,
                                 const float* grad_sum, float* grad_feature, void* scratch, size_t scratch_bytes, void* stream);

/*
 * Standard-normal fill for the reparameterisation noise (replaces the torch.randn_like calls of utils.py:85-88 as
 * used at models.py:361,368 when the caller does not need PyTorch's generator stream):
 *     out[i] ~ N(0,1),  Philox4x32-10 keyed by `seed`, counter (i / 4, *step_count, stream_id), Box-Muller.
 * The step number is read on the device, so a captured hipGraph draws fresh noise on every replay (the prologue
 * increments step_count).  Deterministic: same (seed, step, stream_id, i) -> same value.
 */
int vibo_fill_normal(float* out, int64_t n, uint64_t seed, const int32_t* step_count, uint32_t stream_id, void* stream);

/*
 * num_samples forward evaluations of the ELBO heads in one pass over the response rows: the loop body of
 * log_marginal (models.py:445-504, callers vibo.py:322-347, 550-556) under no_grad.  Per sample s the caller
 * supplies a fresh item sample item[s] ([I][D], after the item flows if any) and ability noise eps[s] ([B][A]); the
 * row loads, codes, counts and the product of experts are shared by the samples (2 or 4 per pass).
 *     out_scalars[s][VIBO_NUM_SCALARS]   same heads as vibo_elbo_fwd_bwd's out_scalars, per sample
 * Posterior outputs and gradients are not produced.  Returns -8 when nothing can be shared between the samples or
 * the descriptor is not on the row-split path (conditional posterior: its table depends on the item sample; int64
 * masks; unaligned rows): loop over vibo_elbo_fwd_bwd instead.
 * Workspace: vibo_multi_workspace_bytes(d, num_samples) bytes, 256-byte aligned.
 */
size_t vibo_multi_workspace_bytes(const vibo_desc* d, int num_samples);
int vibo_elbo_multi_forward(const vibo_desc* d, int num_samples, const float* response, const void* mask,
                            const int64_t* row_index, const float* table, const float* item, const float* eps,
                            const float* flow, float* out_scalars, void* workspace, size_t workspace_bytes, void* stream);

/*
 * The per-term MLP decoders of --generative-model link | deep | residual (LinkedIRT / DeepIRT / ResidualIRT,
 * models.py:769-919) -- masked Bernoulli log-likelihood, forward and backward, of
 *     z1[p][i][:] = U[i][:] + V[p][:] + w1[:] L[p][i]
 *     o[p][i]     = w3 . elu(W2 . elu(z1) + b2) + b3 + resid L[p][i]
 *     P(response[p][i] = 1) = sigmoid(o)   or   guess[i] + (1 - guess[i]) sigmoid(o)   when guess != NULL
 * with hidden_dim = 64.  The caller (PyTorch autograd) forms the per-item / per-person halves of the first layer:
 *   deep / residual:  U = mlp_concat[0].weight[:, :64] . mlp_item_feat(item) [I][64],
 *                     V = mlp_concat[0].weight[:, 64:] . mlp_ability(ability) + bias [B][64];  residual: L = the IRT logit
 *                     (irt_model_*pl(return_logit=True), models.py:729-766), resid = 1, w1 = NULL;  deep: L = NULL
 *   link:             U = NULL, V = link[0].bias broadcast [B][64], w1 = link[0].weight [64], L = the IRT logit, resid = 0
 * and backpropagates the returned gradients through them.  response [B][I] fp32 (row stride in d), mask u8 or NULL.
 * Outputs are partial records in a fixed order (sum them; n_ib = ceil(I / 64), n_wave = 4 n_ib person_chunks):
 *   ll_part [n_wave]; dW2_part [n_wave][64][64]; dvec_part [n_wave][4][64] = d b2 | d w3 | d w1 | (d b3, 0...);
 *   dU_part [person_chunks][I][64]; dV_part [4 n_ib][B][64]; dL [B][I] (complete, not partial);
 *   dguess_part [person_chunks][I] (w.r.t. the guess probability).
 * want_grad = 0: only ll_part and, when prob_out != NULL, prob_out [B][I] = P(response = 1) (decode()).
 * person_chunks: vibo_decoder_person_chunks(B, I), or any value in 1..B.
 */
/*
 * A stack of planar flows on the rows of z [n_rows][dim] (PlanarFlow.forward / NormalizingFlows.forward, flows.py:21-41,
 * 58-66; the item-side stack of models.py:342-348 and the ability-side stack of the MLP-decoder models):
 *     z <- z + uhat_k tanh(w_k . z + b_k),   ladj[row] = sum_k log(|1 + (1 - tanh^2)(w_k . uhat_k)| + 1e-8)
 * packed [n_flows][2 dim + 1] = uhat | w | b per flow (uhat from (u, w) by the caller, flows.py:24-26); dim <= 18,
 * n_flows <= VIBO_MAX_FLOWS.  forward: z_out [n_rows][dim], ladj [n_rows], tanh_out [n_rows][n_flows] (kept for backward).
 * backward: given d/d z_out and d/d ladj, writes d/d z [n_rows][dim] and ceil(n_rows / 256) partial records
 * [n_flows][2 dim + 1] = d/d uhat | d/d w | d/d b that the caller sums (fixed order).
 */
int vibo_flow_stack_forward(int n_rows, int dim, int n_flows, const float* z, const float* packed, float* z_out, float* ladj,
                            float* tanh_out, void* stream);
int vibo_flow_stack_backward(int n_rows, int dim, int n_flows, const float* z_out, const float* packed, const float* tanh_saved,
                             const float* g_zout, const float* g_ladj, float* g_z, float* partials, void* stream);

typedef struct vibo_decoder_desc {
    int32_t num_person, num_item, hidden_dim, want_grad, person_chunks;
    float resid;
    int64_t response_row_stride, mask_row_stride;
} vibo_decoder_desc;
int vibo_decoder_person_chunks(int num_person, int num_item);
int vibo_decoder_fwd_bwd(const vibo_decoder_desc* d, const float* response, const uint8_t* mask,
                         const float* U, const float* V, const float* L, const float* guess, const float* w1,
                         const float* W2, const float* b2, const float* w3, const float* b3,
                         float* ll_part, float* dU_part, float* dV_part, float* dL, float* dguess_part,
                         float* dW2_part, float* dvec_part, float* prob_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VIBO_HIP_H */
