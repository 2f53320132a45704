// Internal (C++) launcher interface between the kernels (*.hip) and the C-ABI layer (pfn_api.hip).
// Nothing here crosses the shared-library boundary; see include/pfn_hip.h for the exported ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/pfn_hip.h"

namespace pfn {

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute and setting it costs tens of microseconds of host time, so
// every launcher keeps one of these per kernel: the largest dynamic-LDS size granted so far on each device.  Lock-free; two host
// threads racing on a first use at worst both set the attribute (idempotent) -- neither can launch before it is set.
struct LdsAllowance {
  std::atomic<size_t> granted[16];   // by device ordinal (a node has 8)
  template <typename K> void ensure(K kernel, size_t bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<size_t>& g = granted[dev & 15];
    if (bytes <= g.load(std::memory_order_acquire)) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    size_t cur = g.load(std::memory_order_relaxed);
    while (cur < bytes && !g.compare_exchange_weak(cur, bytes, std::memory_order_release)) {}
  }
};

// Operand element types (include/pfn_hip.h PFN_PREC_*): two 16-bit formats on the matrix cores at the same rate and bytes, and the exact-f32 parity mode.
inline bool prec_is16(int p) { return p == PFN_PREC_BF16 || p == PFN_PREC_FP16; }
inline int prec_esize(int p) { return prec_is16(p) ? 2 : 4; }
// host-side dispatch of a statement over the operand type T (device code only sees the instantiations)
#define PFN_DISPATCH_OP(prec, ...)                                                          \
  do {                                                                                      \
    if ((prec) == PFN_PREC_BF16) { using T = ::pfn::bf16_t; __VA_ARGS__; }                  \
    else if ((prec) == PFN_PREC_FP16) { using T = ::pfn::f16_t; __VA_ARGS__; }              \
    else { using T = float; __VA_ARGS__; }                                                  \
  } while (0)
typedef __bf16 bf16_t;
typedef _Float16 f16_t;

// In-step kernel timing (test / profiling hook pfn_profile_*, pfn_api.hip): an event pair on the launch stream around the launches inside the scope;
// a no-op unless profiling is enabled.
bool prof_enabled();
void* prof_begin(int slot, hipStream_t s);
void prof_end(void* token, int slot, hipStream_t s);
struct ProfScope {
  void* token; int slot; hipStream_t s;
  ProfScope(int slot_, hipStream_t s_) : token(prof_begin(slot_, s_)), slot(slot_), s(s_) {}
  ~ProfScope() { prof_end(token, slot, s); }
  ProfScope(const ProfScope&) = delete;
  ProfScope& operator=(const ProfScope&) = delete;
};

// ---------------------------------------------------------------------------------------------
// Dropout masks (training with dropout > 0: TransformerEncoderLayer's four sites, reference transformer.py:17 / train.py:22).
// A mask is a pure function of (site seed, i, j) so the backward regenerates what the forward applied, in whatever register
// layout a kernel holds the elements, with no mask tensor in HBM:
//     keep(s, i, j)  <=>  mix32(s ^ i * 0x9E3779B1 ^ j * 0x85EBCA77) >= thr,      thr = p * 2^32
// (mix32 = the "lowbias32" integer finaliser).  site seed = dropout_site_seed(call seed, layer, site); the attention site mixes the
// (dataset, head) index in as well (dropout_pair_seed).  i, j = (query, key) for the attention probabilities, (token row b * S + t,
// column) for the three element-wise sites.  oracle/pfn_oracle.py restates the same integers in numpy.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline unsigned mix32(unsigned h) {
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  return h;
}
__host__ __device__ inline unsigned dropout_site_seed(unsigned long long seed, int layer, int site) {
  return mix32(mix32((unsigned)seed ^ (0x9E3779B1u * (unsigned)(layer * 4 + site + 1))) ^ (unsigned)(seed >> 32));
}
__host__ __device__ inline unsigned dropout_pair_seed(unsigned site_seed, int pair) { return mix32(site_seed ^ (0xC2B2AE35u * (unsigned)(pair + 1))); }
__host__ __device__ inline unsigned dropout_threshold(float p) {
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xffffffffu : (unsigned)t;
}
__host__ __device__ inline bool dropout_keep(unsigned s, unsigned i, unsigned j, unsigned thr) {
  return mix32(s ^ (i * 0x9E3779B1u) ^ (j * 0x85EBCA77u)) >= thr;
}

// ---- GEMM -----------------------------------------------------------------------------------
enum : int {
  EPI_BIAS = 1,      // + bias[n] (f32)
  EPI_GELU = 2,      // out = gelu(v)   (exact erf GELU, torch nn.GELU default)
  EPI_GELU_BWD = 4,  // v *= aux[m,n] (T): the GELU derivative the forward GEMM stored (EPI_GELU | EPI_OUT2_T)
  EPI_RESID = 8,     // + resid[m,n] (f32)
  EPI_OUT_F32 = 16,  // store f32
  EPI_OUT_T = 32,    // store operand type T
  EPI_OUT2_T = 64,   // second T output: with EPI_GELU the GELU DERIVATIVE at the pre-activation, else the value before the activation
  EPI_ACCUM = 128,   // out_f32 += v
  EPI_RESID_T = 256, // + aux[m,n] (T): residual taken from an operand-precision tensor (not with EPI_GELU_BWD)
  // (LDS-DMA 256 / 128 x 256 kernels only; gemm_nt_rowdot_fused tells the caller beforehand)  rowdot[b, head, i] += sum over the tile's columns of
  // out_t[m, n] * aux[m, n], m = b * rd_S + i, head = n / rd_D -- the attention backward's delta = rowsum(dO . O) taken while d(ctx) leaves the GEMM that
  // produces it (aux = the forward's context rows), with f32 atomics (two 64-column waves per head of 128: two addends, order-free; the caller zeroes rowdot);
  // the wave that owns a head's first columns also writes lse2 = lse * log2(e) behind it (what attn_delta_kernel did)
  EPI_ROWDOT = 512,
  // out[m, n] -= rowshift[(m / rs_S) * rs_ld + (n - rs_n0)] for rs_n0 <= n < rs_n1: a per-DATASET shift of a column range, applied in f32 before the value is
  // rounded to operand precision -- the key centring of the q|k|v projection (launch_key_shift below; pfn_api.hip).  rs_n0, rs_n1 multiples of 64.
  EPI_ROWSHIFT = 1024,
};

struct GemmNT {
  const void* A; long lda;  // [M,K] T
  const void* B; long ldb;  // [N,K] T
  int M, N, K;
  int flags;
  const float* bias;
  const void* aux; long ld_aux;
  const float* resid; long ld_resid;
  float* out_f32; long ld_out_f32;
  void* out_t; long ld_out_t;
  void* out2_t; long ld_out2;
  int vec_ok;  // filled by the launcher
  int wide_t;  // filled by the launcher: operand-precision outputs may be stored 16 bytes at a time
  // EPI_ROWDOT: rowdot [B, H, S] f32 (+ lse2 [B, H, S] behind it at rd_lse2_off), lse [B, H, S]; rows m = b * rd_S + i, heads of rd_D columns
  float* rowdot; const float* rd_lse; long rd_lse2_off; int rd_S, rd_H, rd_D;
  // EPI_ROWSHIFT: rowshift [M / rs_S, rs_ld] f32
  const float* rowshift; long rs_ld; int rs_S, rs_n0, rs_n1;
};
// Key centring (round 6).  softmax_j(q_i . k_j) does not change when ONE vector c is subtracted from every key of a (dataset, head): q_i . (k_j - c) = q_i . k_j - q_i . c,
// a constant along j (the self key of a test row is shifted like the others).  In a trained PFN the keys of a dataset share a large common component (measured on
// tests/golden/trained_config1.pt: |k| rms 6.6, |k - mean_j k| rms 0.73) and the 16-bit rounding of k is relative to |k|, not to the part that matters: storing
// k' = k - c with c ~ the dataset's mean key makes the attention 9 x less sensitive to the operand rounding of K (profiles/r06_operand_format_simulation.json: bf16
// forward 5e-2 -> 2e-2, fp16 7e-3 -> 2.5e-3 on that checkpoint).  The backward needs nothing: sum_j dS_ij = 0, so dK' = dK and d(c) = 0 (the reference's gradient).
// c = W_k xbar with xbar the mean of a strided SAMPLE of the dataset's TRAIN rows of the layer input (any c is exact; this one removes the common part to ~ 1 / sqrt(samples);
// train rows only, so that no test row reaches another row's output even at rounding level):
//   kshift[b, n] = sum_e w_k[n, e] * mean_{t in sample} x[b, t, e]        x [B, S, E] T,  w_k [E, E] T (rows = the K block of in_proj_weight),  kshift [B, E] f32
//   sep / sep_of: the eval position (sep_of [B] on the device for a ragged batch, else nullptr); a dataset without train rows gets a zero shift
int launch_key_shift(const void* x_t, const void* w_k_t, float* kshift, int B, int S, int E, int sep, const int* sep_of, int precision, hipStream_t s);
int launch_gemm_nt(const GemmNT& g, int precision, hipStream_t stream);
bool gemm_nt_rowdot_fused(const GemmNT& g, int precision);   // true: launch_gemm_nt(g) will run a kernel that implements EPI_ROWDOT (else the caller keeps attn_delta_kernel)
// kernel selection knob (tests / profiling): 0 automatic, 1 only the 128x128 kernel, 2 the 256x256 kernel whenever legal
void set_gemm_nt_big_mode(int mode);
void set_gemm_ln_rows64(int on);

struct GemmTN {
  const void* A; long lda;  // [M,P] T
  const void* B; long ldb;  // [M,Q] T
  float* C; long ldc;       // [P,Q] f32
  int M, P, Q;
  int atomic;               // 1: C += (split over M, f32 atomics); 0: C = (single split)
  int m_chunk;              // filled by the launcher
  float* colsum;            // optional [P]: += column sums of A (bias gradient), atomically
  int max_splits;           // 0 = automatic; else an upper bound on the token-axis splits (small C: fewer atomic partial sums)
  const float* scale_amax;  // fp16 backward: C and colsum receive the product times 2^-k (nullptr = 1)
};
int launch_gemm_tn(GemmTN g, int precision, hipStream_t stream);

// Grouped weight-gradient GEMMs (bf16): C_i[P_i,Q_i] (+)= A_i[M,P_i]^T . B_i[M,Q_i] for every problem of the
// group in ONE launch of 256x256 tiles (gemm_tn_big_kernel).  splits: 0 = automatic (1 when the group has
// enough tiles: no atomics, deterministic), > 1 = split the token axis with f32 atomics.
constexpr int TN_GROUP_MAX = 26;
struct TnProblem {
  const void* A; const void* B; float* C; float* colsum;  // colsum: optional [P] += column sums of A
  long lda, ldb, ldc;
  int P, Q;
  int Pv;   // rows of C (and entries of colsum) that exist: 0 = all P; < P when A's last columns are zero padding (P still a multiple of 256)
};
struct GemmTNGroup {
  TnProblem p[TN_GROUP_MAX];
  int tile_start[TN_GROUP_MAX + 1];  // filled by the launcher
  int n, M, splits, m_chunk;
  int debug_mask;                    // all ones; profiling only (pfn_set_tuning key 1): 2^k - 1 wraps the token index
  const float* scale_amax;           // fp16 backward: every C and colsum of the group receives its product times 2^-k (nullptr = 1)
};
void set_gemm_tn_debug_wrap(int rows);
void set_gemm_tn_group_waves(int waves);       // PFN_TUNE_WGRAD_WAVES: 8 = gemm_tn_big_kernel (eight waves of 128 x 64), 4 = gemm_tn_wide_kernel (four of 128 x 128)
void set_gemm_tn_group_splits(int splits);      // PFN_TUNE_WGRAD_SPLITS: token-axis splits of the grouped weight-gradient launch when the caller leaves them automatic (0 = the occupancy rule)
// GEMM + bias + residual + LayerNorm in one kernel (gemm_nt_ln_kernel): one workgroup owns 128 full rows of the
// N = emsize columns, so the row statistics are taken straight from the accumulators:
//     v = A . B^T + bias + r,   y = v (f32, kept for the LayerNorm backward),   x_t = (T) ((v - mean) rstd gamma + beta)
// The residual r is either a plain f32 tensor (resid) or the PREVIOUS LayerNorm's output recomputed on the fly from
// its pre-LN sums and statistics (r = (ry - rmean) rrstd rgamma + rbeta): the f32 LayerNorm output is never stored.
struct GemmLN {
  const void* A; long lda;      // [M,K] bf16
  const void* B; long ldb;      // [N,K] bf16
  int M, N, K;
  const float* bias;            // [N]
  const float* resid;           // [M,N] f32, or nullptr to use the recomputed form below
  const void* ry; const float* rmean; const float* rrstd; const float* rgamma; const float* rbeta;      // ry: f32, or operand precision when y16
  const float* gamma; const float* beta; float eps;
  void* y; float* mean; float* rstd; void* x_t;    // outputs (y: f32, or operand precision when y16)
  float* x_f32;                                    // optional: the LayerNorm output in f32 as well (last layer -> decoder)
  int y16;                                         // fp16 only: y is written, and ry read, in operand precision (half the bytes of this HBM-bound epilogue's two f32 streams)
};
bool gemm_ln_supported(const GemmLN& g);
int launch_gemm_ln(const GemmLN& g, int precision, hipStream_t stream);      // precision: one of the 16-bit formats

// A data-gradient GEMM with the backward of the LayerNorm whose OUTPUT gradient it produces fused in (gemm_nt_lnbwd_kernel):
//   v = A . B^T + aux;  dx_t = LayerNorm backward of v through (y, mean, rstd, gamma);  dgamma += sum_rows v xhat;  dbeta += sum_rows v
struct GemmLNB {
  const void* A; long lda;      // [M,K] bf16
  const void* B; long ldb;      // [N,K] bf16
  int M, N, K;
  const void* aux;              // [M,N] bf16 (row pitch N): the residual-branch gradient added to the product
  const void* y; const float* mean; const float* rstd; const float* gamma;    // the LayerNorm's input [M,N] (f32, or operand precision when y16), row statistics [M], weight [N]
  void* dx_t;                   // [M,N] bf16: gradient w.r.t. the LayerNorm input
  float* dgamma; float* dbeta;  // [N] f32, accumulated with atomics
  const float* scale_amax;      // fp16 backward: dgamma / dbeta leave times 2^-k (dx_t stays in the scaled chain); nullptr = 1
  int y16;                      // y as GemmLN::y16 stored it
};
bool gemm_lnbwd_supported(const GemmLNB& g);
int launch_gemm_lnbwd(const GemmLNB& g, int precision, hipStream_t stream);

bool gemm_tn_group_supported(const TnProblem& p);
int launch_gemm_tn_group(GemmTNGroup g, int precision, hipStream_t stream);

// ---- attention (attention.hip) ---------------------------------------------------------------
// qkv: [B, S, 3E] T (q | k | v, heads contiguous inside each E), ctx: [B, S, E] T,
// lse: [B, H, S] f32 (natural-log LSE of the scaled scores over the allowed keys).
// Allowed keys for query i: j < sep, plus j == i when i >= sep (reference transformer.py:34-41).
struct AttnArgs {
  const void* qkv; void* ctx; float* lse;
  int B, S, E, H, sep;
  // ragged batch (round 5): per-dataset eval positions [B] on the device; then `sep` is their maximum (grid of the key-block pass, dS^T scratch dims) and
  // every workgroup reads its own dataset's position.  nullptr = every dataset at `sep`.
  const int* sep_of;
  int q_from_sep;      // ragged batch with the top layer on the test rows: dataset b skips the queries below sep_of[b] / 256 * 256 (q_begin then = the smallest of them: the grids)
  // backward
  const void* dctx; void* dqkv; float* delta;  // delta: [B,H,S] f32 scratch
  void* ds;   // dS^T scratch [B, H, ds_rows, ds_ld] T: written by the key-block pass, read by the query-block pass (attn_bwd_ds_bytes)
  int ds_rows, ds_ld;  // filled by the launcher
  int parts;  // backward launches to run, bit mask over ATTN_BWD_*; 0 = all (profiling entry point pfn_op_attention_bwd_parts)
  int zero_delta;  // the query-block pass (the last reader's successor) leaves delta[b, head, its queries] = 0: the next layer's GEMM epilogue ADDS into it (EPI_ROWDOT)
  int pingpong;  // filled by the launcher (PFN_TUNE_ATTN_PINGPONG): bit 0 forward, bit 1 key-block pass
  // filled by the launcher: the backward's key-block / query-block pair may run for a GROUP of datasets at a time (PFN_TUNE_ATTN_BWD_GROUP) -- datasets [b0, b0 + bg) of the
  // B in this call; every group re-uses the front of the dS^T scratch, so what the key-block pass wrote is still in the 256 MB memory-side cache when the query-block pass reads it
  int b0, bg;
  // forward with the Q PROJECTION FUSED IN (north_star's "QKV projection + attention as one kernel", the half that can exist: K and V are shared by every query block
  // of a head and stay a GEMM).  xq != nullptr (16-bit operands, head dim <= 128, E % 128 == 0): the workgroup forms its 256 queries' head slice q = x W_q[h]^T + b_q[h]
  // on the matrix cores in the prologue -- W_q[h] through LDS in 128-column chunks, x rows straight from global -- instead of reading them from qkv;
  // q_store != 0 also writes them to qkv's Q columns (training: the backward reads them there).
  const void* xq;      // [B, S, E] T: the layer input (operand copy)
  const void* wq;      // [E, E] T: the Q block of in_proj_weight (rows h * D .. of head h)
  const float* bq;     // [E] f32: the Q block of in_proj_bias
  int q_store;
  // Queries below q_begin are skipped: their context rows / dQ rows are not written and they add nothing to dK and dV (the top encoder
  // layer, whose train rows feed nothing: pfn_api.hip).  The launcher rounds it down to a multiple of 256 (whole query blocks / tiles of every
  // kernel); the caller hands in dctx rows that are ZERO in [that multiple, its own first live row).
  int q_begin;
  // dropout on the attention probabilities (training with dropout > 0): P' = P * keep / (1 - p), keep = dropout_keep(pair seed, query, key)
  float p_drop;            // 0 = off (the kernels without the mask arithmetic run)
  unsigned drop_seed;      // site seed of this layer's attention (dropout_site_seed(call seed, layer, 0)); the kernels mix (dataset, head) in
};
void set_attn_pingpong(int mask);
void set_attn_bwd_group(int datasets);      // 0 = all datasets of a call in one launch pair
enum : int { ATTN_BWD_DELTA = 1, ATTN_BWD_KV = 2, ATTN_BWD_DQ = 4 };
int64_t attn_bwd_ds_bytes(int B, int S, int H, int precision);   // size of AttnArgs::ds for any sep <= S
int launch_attn_fwd(const AttnArgs& a, int precision, hipStream_t stream);
bool attn_fwd_can_fuse_q(int E, int H, int precision);      // shapes for which AttnArgs::xq is implemented (16-bit operands, head dim <= 128, E % 128 == 0, no dropout)
int launch_attn_bwd(const AttnArgs& a, int precision, hipStream_t stream);

// ---- row-wise / element-wise kernels (rowwise.hip) --------------------------------------------
int launch_cast_params(const float* src, void* dst_t, long n, int precision, hipStream_t s);
constexpr int TRANSPOSE_GROUP_MAX = 64;
struct TransposeGroup {     // by value in the kernel arguments: offsets in elements from one source / one destination base
  int n = 0, blocks = 0;
  long src_off[TRANSPOSE_GROUP_MAX], dst_off[TRANSPOSE_GROUP_MAX];
  int rows[TRANSPOSE_GROUP_MAX], cols[TRANSPOSE_GROUP_MAX], ld_dst[TRANSPOSE_GROUP_MAX], first_block[TRANSPOSE_GROUP_MAX];
};
void transpose_group_add(TransposeGroup& g, long src_off, long dst_off, int rows, int cols, int ld_dst);
int launch_transpose_cast_group(const float* src, void* dst, const TransposeGroup& g, int precision, hipStream_t s);
int launch_transpose_cast(const float* src, void* dst_t, int rows, int cols, long ld_dst, int precision, hipStream_t s);
// dst[r, 0:C] = (T) src[r, 0:C], dst[r, C:ld_dst] = 0
int launch_cast_rows(const float* src, long ld_src, void* dst_t, long ld_dst, long R, int C, int precision, hipStream_t s, const float* scale_amax = nullptr);
// ---- loss scale of the fp16 backward (pfn_device.h loss_scale_up / loss_scale_down): `scale_amax` arguments below point at ONE device float holding
// max|incoming gradient| (launch_absmax), from which every kernel derives the same power of two; nullptr = no scaling (bf16 / f32) ----
int launch_absmax(const float* x, long n, float* amax, hipStream_t s);
int set_loss_scale_target(int log2_target);      // PFN_TUNE_LOSS_SCALE_TARGET                                   // amax[0] = max |x[i]| (x 16-byte aligned)
int launch_scale_copy(const float* src, float* dst, long n, const float* scale_amax, hipStream_t s);    // dst = src * 2^k

// x:[T,B,nf] (strides given in elements), y:[T,B]; out f32 + T in [B,S,E]
struct EmbedArgs {
  const float* x; long x_st, x_sb;  // x[t,b,f] = x[t*x_st + b*x_sb + f]
  const float* y; long y_st, y_sb;
  const float* wx; const float* bx;  // [E,nf], [E]
  const float* wy; const float* by;  // [E,1],  [E]
  float* out_f32; void* out_t;
  void* xaug_t;   // optional [B*S, xaug_ld] T: the token's features, masked y, train flag, zero padding -- the B operand of the backward's GEMM
  int xaug_ld;    // columns of xaug_t: emb_aug_width(nf)
  int S, B, nf, E, sep;
  const int* sep_of;   // ragged batch: per-dataset eval positions [B] on the device (then `sep` is unused); nullptr = every dataset at `sep`
};
// columns of xaug_t / of the [E, aug] gradient accumulator: num_features + 2 rounded up to 32, 64 or 128 (the GEMM form of the embedding backward; 0 = wider
// encoders keep the register kernel).  Round 6: 64 and 128 added -- BASELINE configs[3] has 60 features and spent 9 % of its kernel time in embed_bwd_wide_kernel
__host__ __device__ inline int emb_aug_width(int nf) { return nf + 2 <= 32 ? 32 : nf + 2 <= 64 ? 64 : nf + 2 <= 128 ? 128 : 0; }
int launch_embed_fwd(const EmbedArgs& a, int precision, hipStream_t s);
// backward: dsrc [B,S,E] f32 -> dwx, dbx, dwy, dby (accumulated, f32)
struct EmbedBwdArgs {
  const float* dsrc; const float* x; long x_st, x_sb; const float* y; long y_st, y_sb;
  float* dwx; float* dbx; float* dwy; float* dby;
  int S, B, nf, E, sep;
  const int* sep_of;    // ragged batch: per-dataset eval positions [B] (nullptr = every dataset at `sep`)
  int single_block;     // 1: one workgroup per column block walks every token (one writer per gradient element: PFN_SCHED_DETERMINISTIC)
  const float* scale_amax;   // fp16 backward: dsrc carries the loss scale, the gradients leave without it (nullptr = none)
};
int launch_embed_bwd(const EmbedBwdArgs& a, hipStream_t s);
// the GEMM form: acc[E, aug] = d(src)^T . xaug (launch_gemm_tn), aug = emb_aug_width(nf) -> dwx += acc[:, :nf], dwy += acc[:, nf], dby += acc[:, nf + 1]
int launch_embed_grad_scatter(const float* acc, float* dwx, float* dwy, float* dby, int E, int nf, hipStream_t s);

// src given in the reference layout [S,B,E] f32 (custom encoders): copy into [B,S,E] f32 + T
int launch_sbe_to_bse(const float* src, float* out_f32, void* out_t, int S, int B, int E, int precision, hipStream_t s);
int launch_bse_to_sbe(const float* src_bse, float* dst_sbe, int S, int B, int E, hipStream_t s, const float* scale_amax = nullptr);      // (* 2^-k)

// y = LN(x) * gamma + beta over E; writes f32 and T copies, mean/rstd per row
int launch_layernorm_fwd(const void* x, const float* gamma, const float* beta, float* y_f32, void* y_t,
                         float* mean, float* rstd, long rows, int E, float eps, int precision, hipStream_t s, int x_is_t = 0);      // x_is_t: x in operand precision
// dx = LN'(dy) (dy f32, or T when dy_is_t); dx written f32 + T; dgamma/dbeta accumulated with atomics; optional dbias_extra
// accumulates colsum(dx) (the bias gradient of the linear that produced x's pre-LN sum).
int launch_layernorm_bwd(const void* dy, int dy_is_t, const void* x, const float* gamma, const float* mean, const float* rstd,
                         float* dx_f32, void* dx_t, float* dgamma, float* dbeta, float* dbias_extra,
                         long rows, int E, int precision, hipStream_t s, float* partials = nullptr, const float* scale_amax = nullptr, int x_is_t = 0);      // x_is_t: x in operand precision (GemmLN::y16);      // scale_amax: dgamma / dbeta / dbias_extra leave times 2^-k
// `partials` (PFN_SCHED_DETERMINISTIC): scratch of LNB_MAX_BLOCKS * 3 * E floats -- every workgroup leaves its column sums there and a second tiny launch adds
// them to dgamma / dbeta / dbias_extra in block order (one writer per element, a fixed summation order) instead of the f32 atomics
constexpr int LNB_MAX_BLOCKS = 512;
// out[n] += sum_m a[m,n]
int launch_colsum(const void* a_t, long lda, long rows, int cols, float* out, int precision, hipStream_t s);

// gather test rows: dst[(s-sep)*B + b, :] = src[b, s, :]  (f32 in, T out)  and its transpose
// ---- element-wise dropout (training with dropout > 0 only; masks: pfn_device.h dropout_keep with i = row, j = column) ----
// y[r, c] = resid[r, c] + keep * y[r, c] / (1 - p)            (dropout1 / dropout2 of TransformerEncoderLayer, in place on the f32 sum)
int launch_dropout_add(float* y, const float* resid, long rows, int cols, unsigned site_seed, float p, hipStream_t s);
// dst[r, c] = keep * src[r, c] / (1 - p) in operand precision; src == dst allowed; a second pair shares the mask (h and gelu'(hpre))
int launch_dropout_scale(const void* src, void* dst, const void* src2, void* dst2, long rows, int cols, unsigned site_seed, float p, int precision, hipStream_t s);
int launch_gather_test_rows(const float* src_bse, void* dst_t, int S, int B, int E, int sep, int precision, hipStream_t s);
// ragged batch (per-dataset eval positions sep_of[B], compact rows dataset-major: row_off[b] + (s - sep_of[b]), row_off[B] = total):
// dst[row_off[b] + s - sep_of[b], :] = (T) src[b, s, :] for s >= sep_of[b]   /   dst[b, s, :] = s >= sep_of[b] ? (T) src[row_off[b] + s - sep_of[b], :] : 0
// the top layer's row moves for a ragged batch (compact rows dataset-major): dst[row_off[b] + s - sep_of[b]] = src[b, s] (s >= sep_of[b]);
// dst[b, s] = s >= sep_of[b] ? src[row_off[b] + s - sep_of[b]] : 0 for s >= zero_from(b) = zero_from_block ? sep_of[b] / 256 * 256 : 0; base[b, s, 0 : width] = 0 for s < sep_of[b] / 256 * 256
int launch_gather_rows_ragged(const void* src_bs, void* dst, int S, int B, long row_bytes, const int* sep_of, const long* row_off, hipStream_t s);
int launch_scatter_rows_ragged(const void* src, void* dst_bs, int S, int B, long row_bytes, const int* sep_of, const long* row_off, int zero_from_block, hipStream_t s);
int launch_zero_row_prefix_ragged(void* base, int S, int B, const int* sep_of, long row_bytes, long width_bytes, hipStream_t s);
int launch_gather_test_rows_ragged(const float* src_bse, void* dst_t, int S, int B, int E, const int* sep_of, const long* row_off, int precision, hipStream_t s);
int launch_scatter_test_rows_ragged(const float* src, void* dst_bse_t, int S, int B, int E, const int* sep_of, const long* row_off, int precision, hipStream_t s);
// dst[b, s, :] = (s >= sep) ? src[(s-sep)*B + b, :] : 0
int launch_scatter_test_rows(const float* src, void* dst_bse_t, int S, int B, int E, int sep, int precision, hipStream_t s);
// the same row moves for any operand, as bytes (the top encoder layer runs on the test rows only, pfn_api.hip): compact order [S - sep, B]
int launch_gather_rows(const void* src_bs, void* dst_tb, int S, int B, long row_bytes, int sep, hipStream_t s);
// rows >= sep take the compact rows, rows in [zero_from, sep) zeros, rows below zero_from are not touched
int launch_scatter_rows(const void* src_tb, void* dst_bs, int S, int B, long row_bytes, int sep, int zero_from, hipStream_t s);
// base[b, s, 0 : width_bytes] = 0 for s < nrows
int launch_zero_row_prefix(void* base, int S, int B, int nrows, long row_bytes, long width_bytes, hipStream_t s);

// ---- bar distribution (bar.hip) ----------------------------------------------------------------
struct BarArgs {
  const float* logits; long ld;  // [R, nbars]
  const float* y;                // [R]
  const float* borders;          // [nbars+1]
  float* nll;                    // [R]
  float* lse;                    // [R] saved for backward
  int* bucket;                   // [R] saved for backward
  long R; int nbars; int full_support;
  // backward
  const float* gout; float* dlogits;
  // mean
  float* mean_out;
};
int launch_bar_nll_fwd(const BarArgs& a, hipStream_t s);
int launch_bar_nll_bwd(const BarArgs& a, hipStream_t s);
int launch_bar_mean(const BarArgs& a, hipStream_t s);

// ---- optimizer (optim.hip) ----------------------------------------------------------------------
struct AdamArgs {
  float* p; float* g; float* m; float* v; long n;
  float lr, beta1, beta2, eps, max_norm, grad_scale;
  int step;
  int zero_grad;
  float* scratch;  // >= 1025 floats; scratch[0] receives the pre-clip global grad norm
};
int launch_clip_adam(const AdamArgs& a, hipStream_t s);

// ---- GP prior sampler (gp_prior.hip) ------------------------------------------------------------
struct GpArgs {
  float* x;        // [B,S,nf] uniform(0,1) features: generated when seed_x != 0, else taken as input
  const float* z;  // [B,S] base normals: taken as input when non-null, else generated
  float* y;        // [B,S] output sample
  float* K;        // [B,S,S] workspace (f32)
  const float* lengthscale;  // [B,nf] per dataset / per feature
  const float* outputscale;  // [B]
  const float* noise;        // [B]
  int B, S, nf, kernel;      // kernel: 0 RBF, 1 Matern-5/2
  unsigned long long seed, offset;
  int gen_x, gen_z;
  int* info;                 // [B] 0 ok, else index+1 of the first non-positive pivot
  // posterior mode (launch_gp_posterior): the same factorisation run as a forward solve  w = L^-1 y_data  -- y holds the
  // running residual (y_data on entry), w the solution, and every panel subtracts L[rows, panel] w[panel] from the
  // residual below it where the sampler adds L[rows, panel] z[panel] to the draw
  float* w;                  // [B,S]; null = sampler mode
  // scratch behind K (gp_workspace_bytes): the two scaled fp16 planes (hi, lo) of the current outer block's solved panel, written by gp_trsm_wide_kernel and
  // read by gp_syrk_planes_kernel -- [B][2 planes][16 k-chunks][plane_rows][16] fp16.  null = the trailing update splits the f32 panel itself (gp_syrk_kernel)
  void* planes;
  long plane_rows;           // rows allocated per slab (>= S - 256, a multiple of 128)
};
int64_t gp_workspace_bytes(int B, int S);      // K [B,S,S] f32 + the plane scratch
void gp_attach_planes(GpArgs& a);               // points a.planes behind a.K (a workspace of gp_workspace_bytes)
int launch_gp_sample(const GpArgs& a, hipStream_t s);
// Sequential exact-GP predictions from ONE factorisation: for every t, the posterior at x_t given points 0..t-1.
//   mean[b,t], var[b,t] (with observation noise), nll[b,t] = -log N(y_t; mean, var);  resid_ws / w_ws: [B,S] scratch
int launch_gp_posterior(const GpArgs& a, const float* y_data, float* nll, float* mean, float* var, hipStream_t s);

// ---- BNN prior sampler (mlp_prior.hip) -----------------------------------------------------------
struct MlpPriorArgs {
  const float* weights;   // [num_models][Lmax][HP][HP]: layer l transposed ([in][out]), zero padded
  const float* biases;    // [num_models][Lmax][HP], zero padded
  const int* model_of;    // [B] model index of each dataset
  const int* dims;        // [num_models][3] = (num_causes, hidden, num_layers)
  const float* noise_std; // [num_models]
  float* causes;          // [B][T][HP]: N(0,1) in the first num_causes columns, written when gen_causes != 0, else input
  const float* noise;     // optional [B][Lmax-1][T][HP] injected standard normals (else generated)
  float* y;               // [B][T] last layer, column 0
  float* hidden;          // optional [B][Lmax-1][T][HP]: the outputs (noise included) of layers 1 .. L-1, the node pool of the causal variant
  int B, T, HP, Lmax, activation, gen_causes;
  unsigned long long seed, offset;
};
int launch_mlp_prior(const MlpPriorArgs& a, hipStream_t s);

}  // namespace pfn
