/*
 * pfn_hip.h -- C ABI of libpfn_hip.so, the MI355X (gfx950) implementation of the PFN training
 * hot path of automl/TransformersCanDoBayesianInference.
 *
 * The reference is pure Python and has no FFI of its own; the drop-in
 * boundary is its Python module surface.  Every entry point below names the reference code it
 * replaces (paths relative to the reference checkout).  A binding needs nothing but dlopen /
 * ctypes.CDLL: plain pointers, sizes, and a HIP stream handle passed as void*.
 *
 * Contract (all entry points):
 *   - every pointer is DEVICE memory owned by the caller unless stated otherwise; the library
 *     never allocates or frees device memory and never synchronises the device;
 *   - work is enqueued asynchronously on `stream` (a hipStream_t; NULL = the legacy default stream);
 *   - return value: PFN_OK (0) or a negative PFN_ERR_* code; pfn_last_error_string() gives detail;
 *     no C++ exception crosses the boundary;
 *   - re-entrant for distinct streams.  Everything that decides what a call computes or how its workspace is laid out travels IN the call
 *     (pfn_model_desc incl. its `schedule` bits, shapes, pointers).  The only process-wide mutable state is (a) the per-thread error
 *     string, (b) the TEST / PROFILING hooks pfn_set_tuning and pfn_profile_* below -- kernel-selection and instrumentation switches that
 *     never change a result beyond rounding order or a buffer layout; they are not synchronised and are meant to be set while no call is
 *     in flight (tests, bench.py --tune); production callers never touch them.
 *
 * Internal activation layout is batch-major [B, S, E] (one synthetic dataset = one contiguous
 * block); the reference layout [S, B, ...] is converted at the boundary kernels.
 */
#ifndef PFN_HIP_H
#define PFN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFN_ABI_VERSION 8

enum {
  PFN_OK = 0,
  PFN_ERR_UNSUPPORTED = -1, /* shape / option outside what the kernels implement */
  PFN_ERR_ALIGNMENT = -2,   /* pointer or leading dimension not 16-byte aligned */
  PFN_ERR_LAUNCH = -3,      /* hipGetLastError() after a launch */
  PFN_ERR_ARGUMENT = -4,    /* NULL pointer, negative size, workspace too small ... */
};

enum { PFN_PREC_BF16 = 0, /* bf16 MFMA operands, f32 accumulate / residual / statistics */
       PFN_PREC_F32 = 1,  /* exact-f32 MFMA (v_mfma_f32_32x32x2_f32): parity / debugging mode */
       PFN_PREC_FP16 = 2  /* ABI 8: fp16 MFMA operands (v_mfma_f32_32x32x16_f16: the same rate and bytes as bf16, 11-bit significand -- the format whose
                           * TRAINING forward meets the reference's outputs to 1e-3, profiles/r06_operand_format_simulation.json); f32 accumulate / residual /
                           * statistics as in bf16.  The backward runs under a power-of-two loss scale chosen on the device from max|dlogits| and taken out again where
                           * the gradients are written (csrc/pfn_device.h LossScale): `grads` holds unscaled f32 gradients, as in the other modes. */ };

/* Architecture of TransformerModel (transformer.py:14-26) with the default Linear encoders
 * (encoders.py:8), NoPositionalEncoding (positional_encodings.py:12-18) and the default
 * decoder Linear-GELU-Linear (transformer.py:23). */
typedef struct pfn_model_desc {
  int32_t num_features; /* x-encoder input width (train.py:33) */
  int32_t emsize;       /* ninp */
  int32_t nhead;
  int32_t nhid;         /* FFN width and decoder hidden width (transformer.py:17,23) */
  int32_t nlayers;
  int32_t n_out;        /* decoder output width (criterion.num_bars for bar losses, train.py:39) */
  int32_t precision;    /* PFN_PREC_* */
  float ln_eps;         /* 1e-5 (torch LayerNorm default) */
  float dropout;        /* TransformerEncoderLayer's dropout probability (transformer.py:17; train.py:22 default 0.2).  0: no dropout
                         * buffers are carved.  > 0: pfn_stack_forward_dropout / pfn_stack_backward_split(use_dropout = 1) apply it at the
                         * layer's four sites (attention probabilities, after out_proj, after the FFN activation, after linear2);
                         * pfn_stack_forward is the inference pass (model.eval()) and applies none. */
  int32_t schedule;     /* PFN_SCHED_* bits, 0 = the measured defaults.  The schedule is part of the descriptor because it decides the row
                         * layout of some workspace buffers: a forward and the backward that reads its workspace must see the same bits, and
                         * they do when both are given the same descriptor (ABI 6; it was process-global state before). */
} pfn_model_desc;
enum { PFN_SCHED_TOP_LAYER_ALL_ROWS = 1, /* run the TOP encoder layer on every row like the others.  Default (bit clear): everything behind its K / V
                                          * projection runs on the test rows only -- the reference returns output[single_eval_pos:] (transformer.py:91),
                                          * so that layer's train rows feed nothing (not with live dropout, not when sep < S / 4) */
       PFN_SCHED_FUSE_LN_WIDE = 2,       /* emsize 1024: LayerNorm-fused GEMMs on 64-row x 1024-column tiles (correct, measured slower: default off) */
       PFN_SCHED_SEPARATE_LNBWD = 4,     /* LayerNorm backward as its own kernels instead of inside the data-gradient GEMMs that feed it */
       PFN_SCHED_NO_KEY_CENTERING = 16,  /* KEY CENTRING (ABI 8): the keys of every dataset are centred before they are rounded to 16 bits: k' = k - W_k xbar with xbar a sample mean
                                          * of the dataset's TRAIN rows of the layer input.  softmax_j(q_i . k_j) is invariant to one vector subtracted from every key, so outputs and
                                          * gradients are the reference's (transformer.py:84) while the rounding of K stops being relative to the keys' common component (nine
                                          * times their spread on a trained model).  Default: ON with PFN_PREC_FP16 -- this bit turns it off -- and OFF with PFN_PREC_BF16 (the
                                          * arithmetic of rounds 1-5) -- PFN_SCHED_KEY_CENTERING turns it on.  Costs one small kernel + a shifted GEMM epilogue per layer: 0.5 % of the step. */
       PFN_SCHED_KEY_CENTERING = 64,
       PFN_SCHED_F32_RESIDUAL = 128,     /* PFN_PREC_FP16 keeps the pre-LayerNorm sums (the residual the next block adds, the LayerNorm backward's input) in f32 as bf16 does.
                                          * Default (bit clear, fp16, no dropout, emsize 128 / 256 / 512): they are stored in fp16 -- the LayerNorm-fused GEMMs are bound by their
                                          * epilogue's HBM streams and this halves the two f32 ones; LayerNorm itself, its statistics and the operand copy still come from f32 registers.
                                          * Other widths (the LayerNorm as its own kernel): the GEMM ahead of it adds the residual from the operand-precision copy and stores an fp16 sum */
       PFN_SCHED_FUSE_Q_PROJECTION = 32, /* the Q projection runs INSIDE the attention forward kernel (north_star: "QKV projection + scaled-dot-product attention + softmax ... as one
                                          * fused kernel"): a workgroup forms its 256 queries' head slice x W_q[h]^T + b_q[h] on the matrix cores in its prologue, the GEMM in
                                          * front projects k | v only (shared by every query block of a head: they stay a GEMM).  Same Q bits as the GEMM's; 16-bit operands,
                                          * head dim <= 128, emsize % 128 == 0, no live dropout -- otherwise ignored.  Built and measured in round 6 (ABI 8); default off:
                                          * profiles/r06_fused_q_projection.txt */
       PFN_SCHED_DETERMINISTIC = 8       /* bit-reproducible gradients (the reference's CPU loop is deterministic, train.py:58-110): every gradient element has
                                          * exactly ONE writer per launch and launches are stream-ordered -- weight-gradient GEMMs without token splits, LayerNorm
                                          * gamma / beta / bias gradients through per-workgroup partials summed in a fixed order (implies SEPARATE_LNBWD), the
                                          * embedding gradient from one workgroup per column block.  The caller runs ONE backward at a time into a gradient buffer
                                          * (no concurrent micro-batch streams).  Slower (weight gradients under-fill the chip); default off. */ };
/* schedule bits a binding should put into new descriptors: 0 unless a test / profiling run changed the defaults through pfn_set_tuning
 * (PFN_TUNE_FUSE_LNBWD, PFN_TUNE_FUSE_LN_WIDE, PFN_TUNE_TOP_LAYER_TEST_ROWS) */
int pfn_default_schedule(void);

int pfn_abi_version(void);
const char* pfn_last_error_string(void);
/* TEST / PROFILING ONLY: process-wide kernel-selection knobs (results are identical up to rounding order; not synchronised -- set them
 * while no call is in flight).  Keys 2, 5, 6, 12 and 13 do not act on calls directly: they change what pfn_default_schedule() hands to NEW
 * descriptors (pfn_model_desc::schedule), so a forward / backward pair can never disagree about them.
 * PFN_TUNE_GEMM_NT_KERNEL: 0 automatic (default), 1 always the 128x128 register-staged
 * kernel, 2 the 256x256 LDS-DMA kernel whenever the shape is legal for it, 3 the 128x256 one.
 * PFN_TUNE_FUSE_LNBWD: 1 (default) the stack backward runs LayerNorm backward inside the data-gradient GEMMs that feed it
 * (pfn_op_gemm_lnbwd), 0 as separate kernels.  (Key 3, the persistent NT GEMM of round 5, is gone from the product library: csrc/experiments/.) */
enum { PFN_TUNE_GEMM_NT_KERNEL = 0,
       PFN_TUNE_GEMM_TN_WRAP = 1, /* profiling only: > 0 makes the grouped TN kernel re-read its first `value` token rows (wrong results, cache-resident operands) */
       PFN_TUNE_FUSE_LNBWD = 2,
       PFN_TUNE_ATTN_PINGPONG = 4, /* bit 0: attention forward, bit 1: backward key-block pass -- the two waves of a SIMD run half a tile apart (default: see attention.hip) */
       PFN_TUNE_FUSE_LN_WIDE = 5,  /* 1: emsize 1024 runs the LayerNorm-fused GEMMs on 64-row x 1024-column tiles (gemm_nt_ln_wide / lnbwd_wide); 0 (default): the
                                    * 256 x 256 GEMM + LayerNorm kernels -- measured faster at that width */
       PFN_TUNE_GP_PLANES = 8,     /* 1 (default): the GP sampler's rank-256 trailing update reads pre-split fp16 planes (hi, lo on a power-of-two scale) by LDS-DMA; 0: it splits the f32 panel into three bf16 terms per tile (rounds 2-3) */
       PFN_TUNE_FUSE_DELTA = 9,    /* 1 (default): on the full-sequence layers of the bf16 stack the attention backward's delta = rowsum(dO . O) is taken in the epilogue of the GEMM that
                                    * produces dO (f32 atomics into a zeroed scratch: two addends per element at head dim 128); 0: a pass of its own over dO and O (attn_delta_kernel) */
       PFN_TUNE_GEMM_LN_ROWS = 7,  /* 1: the LayerNorm-fused GEMMs at emsize 512 run on 64-row tiles, two workgroups per CU (gemm.hip g_ln_rows64); 0 (default): 128-row tiles */
       PFN_TUNE_ATTN_BWD_GROUP = 10, /* > 0: the attention backward's key-block / query-block pass pair runs for that many datasets at a time, every group through the front of
                                      * the dS^T scratch (which then stays in the 256 MB memory-side cache between the two passes); 0 (default): one pair per call */
       PFN_TUNE_WGRAD_SPLITS = 11,   /* > 0: token-axis splits of the grouped weight-gradient launch where the stack leaves them automatic; 0 (default): the occupancy rule */
       PFN_TUNE_WGRAD_WAVES = 14,       /* waves per 256 x 256 tile of the grouped weight-gradient launch: 8 (128 x 64 each) or 4 (128 x 128 each, the whole register file per wave) */
       PFN_TUNE_LOSS_SCALE_TARGET = 15, /* PFN_PREC_FP16 backward: log2 of the value max|dlogits| is scaled to (power-of-two loss scale chosen on the device per call; default 2,
                                         * range -8 .. 12).  16 - target binades of headroom for what the chain adds; overflowing elements saturate at +-65504 */
       PFN_TUNE_RESIDUAL16 = 16,        /* 0: new descriptors carry PFN_SCHED_F32_RESIDUAL; 1 (default): they do not */
       PFN_TUNE_FUSE_Q_PROJECTION = 12, /* 1: new descriptors carry PFN_SCHED_FUSE_Q_PROJECTION (default 0) */
       PFN_TUNE_KEY_CENTERING = 13,     /* 1: new descriptors carry PFN_SCHED_KEY_CENTERING, 0: PFN_SCHED_NO_KEY_CENTERING, -1 (default): neither (centred with fp16, not with bf16) */
       PFN_TUNE_TOP_LAYER_TEST_ROWS = 6 /* 1 (default): the top encoder layer runs everything behind its K / V projection on the test rows only -- the reference
                                    * returns output[single_eval_pos:] (transformer.py:91), so that layer's train rows feed nothing; 0: every layer on every row */ };
int pfn_set_tuning(int key, int value);
/* TEST / PROFILING ONLY: in-step kernel timing.  pfn_profile_enable(1) makes the stack entry points (and pfn_op_attention_*) bracket every
 * launch of the kernel classes below with a pair of HIP events ON THE LAUNCH STREAM; pfn_profile_read(slot, &ms, &n) waits for the pairs
 * recorded for that class so far, returns their summed duration and count and forgets them.  bench.py uses it to report how long the
 * dominant kernel runs INSIDE the step (two micro-batch streams and the prior sampler share the chip), next to its duration on an idle chip.
 * Slots with +1 = the same kernel launched for the top encoder layer's test rows only. */
enum { PFN_PROF_ATTN_FWD = 0, PFN_PROF_ATTN_BWD_DELTA = 2, PFN_PROF_ATTN_BWD_KV = 4, PFN_PROF_ATTN_BWD_DQ = 6, PFN_PROF_GEMM_QKV = 8,
       PFN_PROF_GEMM_OUT_LN = 10, PFN_PROF_GEMM_LIN1 = 12, PFN_PROF_GEMM_LIN2_LN = 14, PFN_PROF_GEMM_DHPRE = 16, PFN_PROF_GEMM_DY1 = 18,
       PFN_PROF_GEMM_DCTX = 20, PFN_PROF_GEMM_DX = 22, PFN_PROF_WGRAD = 24, PFN_PROF_SLOTS = 26 };
int pfn_profile_enable(int on);
int pfn_profile_read(int slot, double* total_ms, int64_t* launches);

/* ---- parameter packing ------------------------------------------------------------------------
 * All parameters live in ONE flat f32 buffer (and one flat f32 gradient buffer) in state-dict
 * order: encoder.{weight,bias}, y_encoder.{weight,bias}, then per layer
 * self_attn.in_proj_{weight,bias}, self_attn.out_proj.{weight,bias}, linear1.{weight,bias},
 * linear2.{weight,bias}, norm1.{weight,bias}, norm2.{weight,bias}, then decoder.0.{weight,bias},
 * decoder.2.{weight,bias} (SURVEY.md 8(b) key list).  Each tensor starts at a multiple of 64
 * elements.  pfn_param_layout fills offsets[i] (elements) for tensor i; returns the tensor count
 * (4 + 12*nlayers + 4) or a negative error.  offsets may be NULL to query the count. */
int pfn_param_layout(const pfn_model_desc* d, int64_t* offsets, int64_t* numels, int max_tensors);
int64_t pfn_param_count(const pfn_model_desc* d);
/* operand-precision shadow of the weights (+ transposed copies for the data-gradient GEMMs) */
int64_t pfn_shadow_bytes(const pfn_model_desc* d);
int pfn_prepare_params(const pfn_model_desc* d, const float* params, void* shadow, void* stream);

/* ---- encoder stack: replaces TransformerModel.forward (transformer.py:55-91) -------------------
 * x: [T,B,F] f32 with element strides (x_st, x_sb, 1); y: [T,B] f32 with strides (y_st, y_sb).
 * sep = single_eval_pos, already normalised to [0,T].  Attention mask semantics of
 * generate_D_q_matrix (transformer.py:34-41): query i may attend key j iff j < sep or j == i.
 * logits: [(T-sep)*B, n_out] f32, row (t-sep)*B + b  (== output[single_eval_pos:] of the reference).
 * workspace: pfn_workspace_bytes() bytes; holds the activations pfn_stack_backward needs.
 * src_sbe: optional [T,B,E] f32 pre-embedded input (custom encoders / positional encodings run
 * in PyTorch); when non-NULL x/y are ignored and pfn_stack_backward returns d(src) in dsrc_sbe. */
int64_t pfn_workspace_bytes(const pfn_model_desc* d, int B, int S);
/* Rows the TOP encoder layer runs on behind its K / V projection for this call shape: (S - sep) * B when the stack drops that layer's train rows
 * (they feed nothing: the reference returns output[single_eval_pos:], transformer.py:91 -- the default; not with PFN_SCHED_TOP_LAYER_ALL_ROWS, not with live
 * dropout, not when sep < S / 4), else B * S.  Same results either way; bench.py counts FLOPs and launches with it. */
int64_t pfn_top_layer_rows(const pfn_model_desc* d, int B, int S, int sep, int use_dropout);
int pfn_stack_forward(const pfn_model_desc* d, const float* params, const void* shadow,
                      const float* x, int64_t x_st, int64_t x_sb,
                      const float* y, int64_t y_st, int64_t y_sb,
                      const float* src_sbe,
                      int B, int S, int sep,
                      void* workspace, int64_t workspace_bytes,
                      float* logits, void* stream);
/* The training forward with dropout (desc.dropout > 0).  Masks are counter-based functions of (dropout_seed, layer, site, element)
 * -- csrc/pfn_device.h `dropout_keep`; the backward regenerates them from the same seed, nothing is stored.  torch's own Philox
 * stream cannot be reproduced, so parity is against the oracle evaluating the SAME masks (oracle/pfn_oracle.py dropout_masks). */
int pfn_stack_forward_dropout(const pfn_model_desc* d, const float* params, const void* shadow,
                              const float* x, int64_t x_st, int64_t x_sb,
                              const float* y, int64_t y_st, int64_t y_sb,
                              const float* src_sbe,
                              int B, int S, int sep,
                              void* workspace, int64_t workspace_bytes,
                              float* logits, void* stream, uint64_t dropout_seed);
/* replaces loss.backward() through the model (train.py:93).  dlogits: [(T-sep)*B, n_out] f32.
 * grads: flat f32 buffer in pfn_param_layout order; gradients are ACCUMULATED into it
 * (train.py:92-97 sums micro-batch gradients). */
int pfn_stack_backward(const pfn_model_desc* d, const float* params, const void* shadow,
                       const float* x, int64_t x_st, int64_t x_sb,
                       const float* y, int64_t y_st, int64_t y_sb,
                       int B, int S, int sep,
                       void* workspace, int64_t workspace_bytes,
                       const float* dlogits, float* grads, float* dsrc_sbe, void* stream);
/* The same backward for data-parallel runs (no counterpart in the reference, which is single-process: train.py:29): the weight
 * gradients of the TOP `first_group_layers` encoder layers are launched as their own group as soon as the data-gradient chain has
 * passed those layers, and `on_first_group(user)` is called on the host right after that launch has been enqueued on `stream` --
 * every gradient of the parameters from layer (nlayers - first_group_layers) to the end of the flat buffer (those layers and the
 * decoder) is then complete in stream order, so the caller can record an event there and start the all-reduce of that part of the
 * buffer while the lower layers' backward still runs.  first_group_layers <= 0 or >= nlayers, or a NULL callback: exactly
 * pfn_stack_backward. */
typedef void (*pfn_host_callback)(void* user);
int pfn_stack_backward_split(const pfn_model_desc* d, const float* params, const void* shadow,
                             const float* x, int64_t x_st, int64_t x_sb,
                             const float* y, int64_t y_st, int64_t y_sb,
                             int B, int S, int sep,
                             void* workspace, int64_t workspace_bytes,
                             const float* dlogits, float* grads, float* dsrc_sbe, void* stream,
                             int first_group_layers, pfn_host_callback on_first_group, void* user,
                             int use_dropout, uint64_t dropout_seed);   /* use_dropout: the forward was pfn_stack_forward_dropout(dropout_seed) */

/* ---- RAGGED BATCH (ABI 7): the micro-batches of one optimizer step as ONE launch set.  The reference runs the k batches of an optimizer step one after the
 * other, each with its own single_eval_pos (train.py:66-97; the notebooks train at batch_size 4 x aggregate_k_gradients 25); a batch of 4 datasets fills a
 * fraction of the chip.  Here the datasets of several batches are stacked along B and every dataset carries its own eval position:
 *   sep_of  [B]     int32, device: eval position of dataset b (0 <= sep_of[b] <= S);  sep_min / sep_max = min / max over b (host values)
 *   row_off [B + 1] int64, device: first compact test row of dataset b; row_off[B] = test_rows = sum_b (S - sep_of[b])
 * Compact test rows (logits / dlogits [test_rows, n_out]) are DATASET-MAJOR: row_off[b] + (t - sep_of[b]) -- not the (t - sep) * B + b of the uniform entry
 * points (a micro-batch's rows are one contiguous slice [b][t]).  Fused embedding only (x / y given, no src_sbe); the top encoder layer runs on the test rows
 * only when every dataset leaves room for it (4 sep_min >= S), on every row otherwise.
 * Same workspace (pfn_workspace_bytes(d, B, S)); gradients are accumulated into `grads` as in pfn_stack_backward; first_group_layers / on_first_group as in
 * pfn_stack_backward_split.  The caller forms each micro-batch's loss from its slice, so the summed gradient equals the reference's sequential accumulation. */
int pfn_stack_forward_ragged(const pfn_model_desc* d, const float* params, const void* shadow,
                             const float* x, int64_t x_st, int64_t x_sb,
                             const float* y, int64_t y_st, int64_t y_sb,
                             int B, int S, const int32_t* sep_of, const int64_t* row_off, int sep_min, int sep_max, int64_t test_rows,
                             void* workspace, int64_t workspace_bytes, float* logits, void* stream,
                             int use_dropout, uint64_t dropout_seed);
int pfn_stack_backward_ragged(const pfn_model_desc* d, const float* params, const void* shadow,
                              const float* x, int64_t x_st, int64_t x_sb,
                              const float* y, int64_t y_st, int64_t y_sb,
                              int B, int S, const int32_t* sep_of, const int64_t* row_off, int sep_min, int sep_max, int64_t test_rows,
                              void* workspace, int64_t workspace_bytes,
                              const float* dlogits, float* grads, void* stream,
                              int first_group_layers, pfn_host_callback on_first_group, void* user,
                              int use_dropout, uint64_t dropout_seed);

/* ---- bar distribution: replaces BarDistribution / FullSupportBarDistribution.forward and .mean
 * (bar_distribution.py:19-38, 83-117).  logits [R, nbars] f32 (row stride ld), y [R], borders
 * [nbars+1] sorted.  nll [R].  lse [R] and bucket [R] are saved for the backward. */
int pfn_bar_nll_forward(const float* logits, int64_t ld, const float* y, const float* borders,
                        int64_t R, int nbars, int full_support,
                        float* nll, float* lse, int32_t* bucket, void* stream);
/* dlogits[r,:] = gout[r] * (softmax(logits[r,:]) - onehot(bucket[r])) */
int pfn_bar_nll_backward(const float* logits, int64_t ld, const float* lse, const int32_t* bucket,
                         const float* gout, int64_t R, int nbars, float* dlogits, void* stream);
int pfn_bar_mean(const float* logits, int64_t ld, const float* borders, int64_t R, int nbars,
                 int full_support, float* mean, void* stream);

/* ---- optimizer: replaces clip_grad_norm_(params, 1.) + Adam.step() + zero_grad()
 * (train.py:55,95-97).  One fused pass over the flat buffers; the clip coefficient is computed on
 * the device (no host sync).  grad_scale multiplies g first (1/world for DP averaging).
 * scratch: >= 8192 bytes, zeroed once by the caller; scratch[0] (f32) receives the global gradient norm before clipping.
 * A non-finite norm (inf / NaN anywhere in g) SKIPS the step: params and moments are left as they are, g is still cleared when zero_grad != 0, and
 * scratch[1025] (f32) counts the skipped steps -- GradScaler's found_inf behaviour; the reference's loop would write NaN into every parameter.
 * zero_grad != 0 clears g after the update. */
int pfn_clip_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                       float lr, float beta1, float beta2, float eps, float max_norm,
                       float grad_scale, int step, int zero_grad, float* scratch, void* stream);

/* ---- GP prior sampler: replaces priors.fast_gp.get_batch (priors/fast_gp.py:35-58) and the
 * sampling half of priors.fast_gp_mix.get_batch (priors/fast_gp_mix.py:85-99).
 * y_b ~ N(0, outputscale_b * k(x_b, x_b; lengthscale_b) + noise_b * I) by Gram -> Cholesky -> L z.
 * x [B,S,nf] f32: filled with U[0,1) from the counter-based generator when gen_x != 0, else input.
 * z [B,S] f32 base normals: generated when gen_z != 0 (and written back), else input.
 * lengthscale [B,nf], outputscale [B], noise [B].  kernel: 0 = RBF, 1 / 2 / 3 = Matern nu = 2.5 / 1.5 / 0.5 (gpytorch MaternKernel's three closed forms).
 * K_ws: workspace of pfn_gp_workspace_bytes(B, S) bytes: the [B,S,S] f32 matrix, factored in place (SCRATCH: with the plane scratch attached only the
 * 256 x 256 diagonal blocks of the factor are left in it -- the panels below them are consumed from the planes and never stored), followed by the scratch of
 * the trailing update (TWO sets -- the solved panels of an even and an odd outer block, the delayed rank-512 update -- of two scaled fp16 planes each: + 23 % at S = 2000, + 39 % at 1000, + 50 % at 512; always size with pfn_gp_workspace_bytes).  K_ws_bytes (ABI 7): the size the caller allocated --
 * below B*S*S*4 the call returns PFN_ERR_ARGUMENT; between that and pfn_gp_workspace_bytes(B, S) the trailing update runs without the plane scratch
 * (same arithmetic, slower), so a caller sized for an older ABI cannot be written past.  info [B]: 0 or (index+1) of the first non-positive pivot. */
int64_t pfn_gp_workspace_bytes(int B, int S);
int pfn_gp_prior_sample(float* x, float* z, float* y, float* K_ws, int64_t K_ws_bytes,
                        const float* lengthscale, const float* outputscale, const float* noise,
                        int B, int S, int nf, int kernel, int gen_x, int gen_z,
                        uint64_t seed, uint64_t offset, int32_t* info, void* stream);

/* ---- Exact-GP sequential predictions: replaces priors.fast_gp.evaluate (priors/fast_gp.py:88-120), which refits a
 * gpytorch ExactGP on points 0..t-1 and predicts point t for every t (one Cholesky per t).  Here ONE factorisation of
 * the full covariance gives all of them (gp_prior.hip): for every dataset b and position t, the posterior at x[b,t]
 * given (x[b,:t], y[b,:t]) under the GP with the given hyper-parameters:
 *   mean[b,t], var[b,t] (predictive, observation noise included), nll[b,t] = -log N(y[b,t]; mean, var).
 * Position 0 is the prior.  x [B,S,nf], y [B,S]; K_ws / K_ws_bytes as in pfn_gp_prior_sample; resid_ws / w_ws [B,S] scratch; nll / mean / var may be
 * null.  S % 4 == 0.  info as in pfn_gp_prior_sample. */
int pfn_gp_posterior(const float* x, const float* y, float* K_ws, int64_t K_ws_bytes, float* resid_ws, float* w_ws,
                     const float* lengthscale, const float* outputscale, const float* noise,
                     int B, int S, int nf, int kernel, float* nll, float* mean, float* var,
                     int32_t* info, void* stream);

/* ---- BNN prior sampler: replaces the per-dataset module forwards of priors.mlp.get_batch (priors/mlp.py:116-124
 * network, :150-157 forward of the non-causal branch, :195-197 Python loop over datasets).  For dataset b with model
 * m = model_of[b]:  h_0 = causes W_0^T + b_0;  h_l = act(h_{l-1}) W_l^T + b_l + noise_std[m] * eps_l  (1 <= l < L_m);
 * y[b,t] = h_{L-1}[t, 0].
 * weights [num_models][Lmax][HP][HP] f32: layer l stored TRANSPOSED ([in][out]) and zero padded to HP (a multiple of 4,
 * <= 152); biases [num_models][Lmax][HP]; dims [num_models][3] = (num_causes, hidden, num_layers); noise_std [num_models].
 * causes [B][T][HP]: filled with N(0,1) in the first num_causes columns when gen_causes != 0 (the x of the dataset),
 * else taken as input.  noise: NULL (generated, Philox) or [B][Lmax-1][T][HP] standard normals.
 * hidden: NULL, or [B][Lmax-1][T][HP] receiving the outputs (noise included) of layers 1 .. L-1 -- the node pool from which the
 * causal variant (priors/mlp.py:158-166, `outputs[2:]`) picks its features and target.
 * activation: 0 identity, 1 relu, 2 tanh, 3 sigmoid. */
int pfn_mlp_prior_forward(const float* weights, const float* biases, const int32_t* model_of, const int32_t* dims,
                          const float* noise_std, float* causes, const float* noise, float* y, float* hidden,
                          int B, int T, int HP, int Lmax, int activation, int gen_causes,
                          uint64_t seed, uint64_t offset, void* stream);

/* ---- single-op entry points (unit tests / profiling of individual kernels) --------------------
 * prec selects operand element type T: PFN_PREC_BF16 / PFN_PREC_FP16 (2 bytes) or PFN_PREC_F32; the LDS-DMA kernels (grouped weight gradients,
 * LayerNorm-fused GEMMs) take the two 16-bit formats only. */
int pfn_op_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K,
                   int flags, const float* bias, const void* aux, int64_t ld_aux,
                   const float* resid, int64_t ld_resid, float* out_f32, int64_t ld_out_f32,
                   void* out_t, int64_t ld_out_t, void* out2_t, int64_t ld_out2, int prec, void* stream);
int pfn_op_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                   int M, int P, int Q, int atomic, int prec, void* stream);
/* grouped weight-gradient GEMMs (16-bit operands only): for i < n, C[i][P[i],Q[i]] += A[i][M,P[i]]^T . B[i][M,Q[i]] and, when
 * colsum && colsum[i], colsum[i][P[i]] += column sums of A[i]; one launch of 256x256 tiles.  The pointer / size
 * tables are HOST arrays.  splits: 0 automatic, 1 no split (deterministic, no atomics), > 1 token-axis splits.
 * P[i] and Q[i] must be multiples of 256 (PFN_ERR_UNSUPPORTED otherwise). */
int pfn_op_gemm_tn_group(int n, const void* const* A, const int64_t* lda, const void* const* B, const int64_t* ldb,
                         float* const* C, const int64_t* ldc, const int32_t* P, const int32_t* Q,
                         float* const* colsum, int M, int splits, int prec, void* stream);
/* fused linear + bias + residual + LayerNorm (16-bit operands, N in {128, 256, 512}, K % 32 == 0):
 *   v = A[M,K] . B[N,K]^T + bias + r;  y = v (f32);  mean / rstd of v per row;  x_t = bf16((v - mean) rstd gamma + beta)
 * r = resid[M,N] (f32) when resid != NULL, else the previous LayerNorm's output recomputed as
 * (ry - rmean) rrstd rgamma + rbeta.  Replaces `x = norm(x + dropout(sublayer(x)))` of torch's TransformerEncoderLayer
 * (nn/modules/transformer.py:952-957) for the out_proj and linear2 sublayers.
 * prec | PFN_OP_SUMS_16BIT (PFN_PREC_FP16 only): y is WRITTEN, and ry READ, in operand precision (the pointers then address fp16 rows) -- what the stack does for
 * fp16 models unless PFN_SCHED_F32_RESIDUAL is set; pfn_op_gemm_lnbwd takes the same flag for its y. */
#define PFN_OP_SUMS_16BIT 256
int pfn_op_gemm_ln(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, const float* bias,
                   const float* resid, const float* ry, const float* rmean, const float* rrstd,
                   const float* rgamma, const float* rbeta, const float* gamma, const float* beta, float eps,
                   float* y, float* mean, float* rstd, void* x_t, int prec, void* stream);
/* data-gradient GEMM + residual-branch gradient + the backward of the LayerNorm whose output gradient the sum is (16-bit
 * operands, N in {128, 256, 512}, K % 32 == 0):
 *   v = A[M,K] . B[N,K]^T + aux[M,N];   xhat = (y - mean) rstd;
 *   dx_t = bf16(rstd (gamma v - mean_n(gamma v) - xhat mean_n(gamma v xhat)));   dgamma += sum_m v xhat;   dbeta += sum_m v
 * i.e. autograd of `norm(x + sublayer(x))` (torch nn/modules/transformer.py:952-957) w.r.t. the norm's input, taking the
 * place of pfn_op_gemm_nt(EPI_RESID_T | EPI_OUT_T) followed by pfn_op_layernorm_bwd.  dgamma / dbeta accumulate atomically. */
int pfn_op_gemm_lnbwd(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, const void* aux,
                      const float* y, const float* mean, const float* rstd, const float* gamma,
                      void* dx_t, float* dgamma, float* dbeta, int prec, void* stream);
int pfn_op_attention_fwd(const void* qkv, void* ctx, float* lse, int B, int S, int E, int H, int sep,
                         int prec, void* stream);
/* Attention backward = three launches: delta = rowsum(dO * O); the key-block pass (dK, dV and dS^T into ds_ws); the
 * query-block pass (dQ from ds_ws, plus the self-key terms of the test rows).  ds_ws: pfn_op_attention_bwd_ws_bytes(B, S, H,
 * prec) bytes of scratch; delta_ws: 2 * B * H * S floats (the first launch leaves [delta | lse in log2 units] there).  parts: 0 = all, else a bit mask (1 delta, 2 key-block pass, 4 query-block pass) that lets bench.py
 * time every launch on its own; a partial run leaves the outputs of the skipped launches untouched. */
int64_t pfn_op_attention_bwd_ws_bytes(int B, int S, int H, int prec);
int pfn_op_attention_bwd(const void* qkv, const void* ctx, const float* lse, const void* dctx,
                         void* dqkv, float* delta_ws, void* ds_ws, int B, int S, int E, int H, int sep,
                         int prec, int parts, void* stream);
/* The same two with the queries below q_begin skipped (q_begin is rounded DOWN to a multiple of 256 = whole query blocks of every kernel): their
 * ctx / lse rows are not written; in the backward they add nothing to dK / dV and their dQ rows are zeros.  dctx must be zero in
 * [q_begin rounded down, first row that carries a gradient).  This is how the stack runs its TOP layer, whose train rows feed nothing
 * (the reference returns output[single_eval_pos:], transformer.py:91): q_begin = sep. */
int pfn_op_attention_fwd_from(const void* qkv, void* ctx, float* lse, int B, int S, int E, int H, int sep, int q_begin,
                              int prec, void* stream);
int pfn_op_attention_bwd_from(const void* qkv, const void* ctx, const float* lse, const void* dctx,
                              void* dqkv, float* delta_ws, void* ds_ws, int B, int S, int E, int H, int sep, int q_begin,
                              int prec, int parts, void* stream);
/* Rows of [B, S] token order <-> the decoder's compact test-row order [(S - sep), B] (row bytes a multiple of 4); the scatter
 * writes zeros into rows [zero_from, sep) and leaves rows below zero_from alone. */
int pfn_op_gather_rows(const void* src_bs, void* dst_tb, int B, int S, int64_t row_bytes, int sep, void* stream);
int pfn_op_scatter_rows(const void* src_tb, void* dst_bs, int B, int S, int64_t row_bytes, int sep, int zero_from, void* stream);
int pfn_op_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y_f32, void* y_t,
                         float* mean, float* rstd, int64_t rows, int E, float eps, int prec, void* stream);
/* dy: f32 when dy_is_t == 0, operand precision (prec) otherwise -- the form the backward schedule feeds it;
 * dx_f32 / dx_t may each be null. */
int pfn_op_layernorm_bwd(const void* dy, int dy_is_t, const float* x, const float* gamma, const float* mean,
                         const float* rstd, float* dx_f32, void* dx_t, float* dgamma, float* dbeta,
                         float* dbias_extra, int64_t rows, int E, int prec, void* stream);
int pfn_op_cast(const float* src, void* dst, int64_t n, int prec, void* stream);
/* The packed q|k|v projection of one encoder layer as the stack runs it (torch multi_head_attention_forward's in_proj: reference transformer.py:84):
 *   qkv[b, t, :] = x[b, t, :] . w_in^T + b_in,   x [B, S, E] T,  w_in [3E, E] T,  b_in [3E] f32,  qkv [B, S, 3E] T
 * and, when center != 0 (16-bit operands, E % 64 == 0), with the keys of every dataset centred before they are rounded (ABI 8, PFN_SCHED_NO_KEY_CENTERING clear):
 *   k'[b, t, :] = k[b, t, :] - W_k . xbar[b],   xbar[b] = mean of <= 64 evenly spaced rows t' = i * (sep_b / ns) (i < ns = min(64, sep_b)) of x[b, 0 : sep_b, :]
 * (sep_b = sep_of[b] when sep_of != NULL -- a device array -- else sep).  kshift_ws: B * E floats of scratch (receives W_k . xbar). */
int pfn_op_qkv_projection(const void* x_t, const void* w_in_t, const float* b_in, void* qkv_t, float* kshift_ws,
                          int B, int S, int E, int sep, const int32_t* sep_of, int center, int prec, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PFN_HIP_H */
