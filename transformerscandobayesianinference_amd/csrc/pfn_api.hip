// C-ABI layer of libpfn_hip.so: parameter packing, workspace carving, and the forward/backward
// schedules of the PFN encoder stack (see include/pfn_hip.h for the contract).
//
// Forward schedule  == TransformerModel.forward (reference transformer.py:55-91) with
// nn.TransformerEncoderLayer in post-norm form (torch nn/modules/transformer.py:952-957):
//     x = LN1(x + out_proj(attn(in_proj(x))));  x = LN2(x + linear2(gelu(linear1(x))))
// Backward schedule == autograd of the same graph (train.py:93), written out by hand.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>
#include "pfn_kernels.h"

using namespace pfn;

// ---- in-step kernel timing (test / profiling hook, include/pfn_hip.h pfn_profile_*) -------------------------------------------------
// Event pairs on the launch stream around the launches of a kernel class; nothing is recorded unless enabled.
namespace pfn {
namespace {
struct ProfPair { hipEvent_t a, b; };
std::mutex g_prof_mu;
std::vector<ProfPair> g_prof[PFN_PROF_SLOTS];
std::atomic<int> g_prof_on{0};
}  // namespace
bool prof_enabled() { return g_prof_on.load(std::memory_order_relaxed) != 0; }
void* prof_begin(int slot, hipStream_t s) {
  const int on = g_prof_on.load(std::memory_order_relaxed);
  if (!on || slot < 0 || slot >= PFN_PROF_SLOTS) return nullptr;
  if (on >= 2 && (slot & ~1) != on - 2) return nullptr;      // pfn_profile_enable(2 + class): that kernel class alone (fewer events in the queues: less perturbation)
  ProfPair* p = new ProfPair;
  if (hipEventCreate(&p->a) != hipSuccess || hipEventCreate(&p->b) != hipSuccess) { delete p; return nullptr; }
  (void)hipEventRecord(p->a, s);
  return p;
}
void prof_end(void* token, int slot, hipStream_t s) {
  if (!token) return;
  ProfPair* p = (ProfPair*)token;
  (void)hipEventRecord(p->b, s);
  std::lock_guard<std::mutex> lock(g_prof_mu);
  g_prof[slot].push_back(*p);
  delete p;
}
}  // namespace pfn

namespace {

thread_local char g_err[512] = "";
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define PFN_TRY(expr)                                                                   \
  do {                                                                                  \
    int rc_ = (expr);                                                                   \
    if (rc_ != PFN_OK) return fail(rc_, "%s failed with %d at %s:%d", #expr, rc_, __FILE__, __LINE__); \
  } while (0)

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
inline int esize(int prec) { return prec_esize(prec); }

int check_desc(const pfn_model_desc* d) {
  if (!d) return fail(PFN_ERR_ARGUMENT, "null model descriptor");
  if (d->precision != PFN_PREC_BF16 && d->precision != PFN_PREC_F32 && d->precision != PFN_PREC_FP16) return fail(PFN_ERR_ARGUMENT, "bad precision %d", d->precision);
  if (d->num_features < 1 || d->emsize < 8 || d->nhead < 1 || d->nhid < 8 || d->nlayers < 0 || d->n_out < 0)
    return fail(PFN_ERR_ARGUMENT, "bad model dimensions");   // n_out == 0: no decoder -- the stack returns the encoder's test rows
  if (d->emsize % d->nhead) return fail(PFN_ERR_ARGUMENT, "emsize %d not divisible by nhead %d", d->emsize, d->nhead);
  if (d->emsize % 8 || d->nhid % 8) return fail(PFN_ERR_UNSUPPORTED, "emsize and nhid must be multiples of 8 (16-byte operand rows)");
  const int dh = d->emsize / d->nhead;
  // (head dim 256 in the exact-f32 mode: forward only -- pfn_stack_backward refuses it; inference passes of a bf16-trained model run there)
  const bool ok = dh == 32 || dh == 64 || dh == 128 || dh == 256;
  if (!ok) return fail(PFN_ERR_UNSUPPORTED, "head dim %d unsupported (32/64/128/256)", dh);
  if (d->emsize > 2048) return fail(PFN_ERR_UNSUPPORTED, "emsize > 2048 unsupported by the LayerNorm kernels");
  if (d->schedule & ~(PFN_SCHED_TOP_LAYER_ALL_ROWS | PFN_SCHED_FUSE_LN_WIDE | PFN_SCHED_SEPARATE_LNBWD | PFN_SCHED_DETERMINISTIC | PFN_SCHED_NO_KEY_CENTERING | PFN_SCHED_FUSE_Q_PROJECTION | PFN_SCHED_KEY_CENTERING | PFN_SCHED_F32_RESIDUAL)) return fail(PFN_ERR_ARGUMENT, "unknown schedule bits 0x%x", d->schedule);
  if (!(d->dropout >= 0.f && d->dropout < 1.f)) return fail(PFN_ERR_ARGUMENT, "dropout %g outside [0, 1)", (double)d->dropout);
  return PFN_OK;
}

// ---- parameter layout ---------------------------------------------------------------------------
struct LayerP { int64_t w_in, b_in, w_o, b_o, w1, b1, w2, b2, g1, be1, g2, be2; };
struct Layout {
  int64_t enc_w, enc_b, yenc_w, yenc_b;
  std::vector<LayerP> layer;
  int64_t dec0_w, dec0_b, dec2_w, dec2_b;
  int64_t total;                       // elements of the flat f32 buffer
  // transposed operand-precision copies (element offsets into the shadow's second region)
  std::vector<LayerP> layer_t;         // only w_in, w_o, w1, w2 used
  int64_t dec0_wt, dec2_wt, total_t;
  int n_out_pad;
  std::vector<int64_t> offsets, numels;  // state-dict order
};

Layout make_layout(const pfn_model_desc& d) {
  Layout L;
  const int64_t E = d.emsize, F = d.nhid, nf = d.num_features, O = d.n_out;
  int64_t cur = 0;
  auto take = [&](int64_t n) { int64_t o = cur; L.offsets.push_back(o); L.numels.push_back(n); cur = align_up(cur + n, 64); return o; };
  L.enc_w = take(E * nf); L.enc_b = take(E); L.yenc_w = take(E); L.yenc_b = take(E);
  L.layer.resize(d.nlayers);
  for (auto& p : L.layer) {
    p.w_in = take(3 * E * E); p.b_in = take(3 * E); p.w_o = take(E * E); p.b_o = take(E);
    p.w1 = take(F * E); p.b1 = take(F); p.w2 = take(E * F); p.b2 = take(E);
    p.g1 = take(E); p.be1 = take(E); p.g2 = take(E); p.be2 = take(E);
  }
  if (O > 0) { L.dec0_w = take(F * E); L.dec0_b = take(F); L.dec2_w = take(O * F); L.dec2_b = take(O); }
  else L.dec0_w = L.dec0_b = L.dec2_w = L.dec2_b = 0;
  L.total = cur;
  L.n_out_pad = (int)align_up(O, O >= 64 ? 64 : 8);   // contraction length of the decoder's backward GEMM: whole 64-deep stages when it is long (zero padded on both operands)
  int64_t ct = 0;
  auto take_t = [&](int64_t n) { int64_t o = ct; ct = align_up(ct + n, 64); return o; };
  L.layer_t.resize(d.nlayers);
  for (auto& p : L.layer_t) { p.w_in = take_t(3 * E * E); p.w_o = take_t(E * E); p.w1 = take_t(F * E); p.w2 = take_t(E * F); }
  if (O > 0) { L.dec0_wt = take_t(F * E); L.dec2_wt = take_t(F * (int64_t)L.n_out_pad); }
  else L.dec0_wt = L.dec2_wt = 0;
  L.total_t = ct;
  return L;
}

// ---- workspace ------------------------------------------------------------------------------------
struct LayerWs {
  // hpre: gelu'(pre-activation of linear1) -- what the backward multiplies by
  char *qkv, *ctx, *x1_t, *hpre, *h, *x2_t; float *lse, *y1, *mean1, *rstd1, *x1, *y2, *mean2, *rstd2, *x2;
  // backward: output-gradient operands of this layer's four weight gradients, kept until the grouped launch
  char *dy2_t, *dh_t, *dy1_t, *dqkv_t;
  char *dy2m_t, *dy1m_t;   // dropout > 0 only: the LayerNorm-input gradients times the dropout2 / dropout1 masks (the GEMM operands; the residual path keeps the unmasked ones)
};
struct Ws {
  float* x0; char* x0_t;
  char* xaug_t;              // [M, emb_aug_width(nf)] operand precision: the embedding's augmented inputs (embed_fwd -> the backward's GEMM)
  float* embacc;             // [E, emb_aug_width(nf)] f32: that GEMM's result before it is scattered into the encoder gradients
  std::vector<LayerWs> layer;
  char *xt_t, *dpre, *dt;
  // backward scratch
  char *dlog_t, *dd_t, *dctx_t;
  float *dxt, *gA, *delta;   // gA: f32 gradient of the embedding output (the last dx of the backward)
  char* ds;                  // dS^T of the attention backward (key-block pass -> query-block pass), one layer at a time
  char* gA_t;                // gradient w.r.t. a layer's output between layers, operand precision
  // the top layer on the test rows only (top_layer_on_test_rows below): compact [S - sep, B] row order
  char *top_ctx_t, *top_dy1_t, *top_dctx_t;   // attention output / LN1-input gradient / d(attention output) of the test rows
  float *top_ry, *top_rmean, *top_rrstd;      // the layer input (the residual of its first LayerNorm) of the test rows
  float* kshift;                              // [B, E] f32: the per-dataset key shift of the layer whose q|k|v projection runs next (16-bit operands; launch_key_shift)
  float* lscale;                              // PFN_PREC_FP16: max|dlogits| of the running backward call, from which every kernel derives the loss scale (pfn_device.h)
  float* ln_part;                             // PFN_SCHED_DETERMINISTIC: per-workgroup column sums of the LayerNorm backward (launch_layernorm_bwd `partials`)
  int64_t bytes;
};

// A ragged batch (round 5; pfn_stack_forward_ragged / pfn_stack_backward_ragged): the micro-batches of one optimizer step as ONE launch set, each dataset with its
// own eval position.  sep_of [B] int32 and row_off [B + 1] int64 live on the device; row_off[b] = first compact test row of dataset b, row_off[B] = test_rows.
struct Ragged { const int32_t* sep_of; const int64_t* row_off; int64_t test_rows; int sep_min; };

// Key centring of the q|k|v projection (pfn_kernels.h launch_key_shift): on by default with fp16 operands (the format chosen for its accuracy: 0.5 % of the step buys
// 1.3 - 1.7 x on a sharply trained model), opt-in with bf16 (whose error is dominated by its 8-bit significand everywhere else; PFN_SCHED_KEY_CENTERING keeps rounds 1-5's
// arithmetic the default there)
static bool key_centering(const pfn_model_desc& d) {
  return (d.precision == PFN_PREC_FP16 && !(d.schedule & PFN_SCHED_NO_KEY_CENTERING)) || (d.precision == PFN_PREC_BF16 && (d.schedule & PFN_SCHED_KEY_CENTERING));
}

// The pre-LayerNorm sums (y1 / y2 of every layer: the residual the NEXT block adds, and the LayerNorm backward's input) in operand precision instead of f32
// (GemmLN::y16, PFN_SCHED_F32_RESIDUAL clear): the LayerNorm-fused GEMMs are bound by their epilogue's HBM streams -- A in, residual in, sums out, operand copy
// out -- and this halves the two f32 ones (392 -> 260 MB per out_proj launch at configs[1]).  fp16 only (11 bits; emulated on trained and untrained weights before
// it was built: tools/sim_operand_formats.py class Y), and only where the fused kernels run on every layer: no dropout in the descriptor, widths they cover.
// (The buffers keep their f32 size: a descriptor-only rule must not decide a layout the pointer-alignment probe in the forward can still overrule.)
static bool residual16(const pfn_model_desc& d) {
  const int E = d.emsize;
  return d.precision == PFN_PREC_FP16 && !(d.schedule & PFN_SCHED_F32_RESIDUAL) && d.nlayers > 0 && d.dropout == 0.f && d.nhid % 32 == 0 &&
         (E == 128 || E == 256 || E == 512 || (E == 1024 && (d.schedule & PFN_SCHED_FUSE_LN_WIDE)));
}
// ... and where the LayerNorm stays its own kernel (emsize 1024 by default: the wide fused kernels lose there) an fp16 model runs the same streams in fp16: the GEMM in
// front takes its residual from the operand-precision copy of the LayerNorm output and stores the sum in fp16, layernorm_fwd reads that and writes the operand copy
// only (the f32 LayerNorm output survives for the last layer, which feeds the decoder gather) -- 8 instead of 18 bytes per element around every LayerNorm.
static bool residual16_separate_ln(const pfn_model_desc& d, bool ln_fusable) {
  // (from emsize 1024 on: there the streams are the kernels' bound.  Narrow unfusable widths keep f32 -- nothing to gain, and here the ROUNDED sum is what the
  // LayerNorm normalises, so the rounding reaches the operands and not only the residual path: measured 1.3 x on the logits at emsize 1024, 4 x at 64)
  return d.precision == PFN_PREC_FP16 && !(d.schedule & PFN_SCHED_F32_RESIDUAL) && d.nlayers > 0 && d.dropout == 0.f && d.emsize >= 1024 && d.emsize % 8 == 0 && !ln_fusable;
}
// can out_proj / linear2 run as gemm_nt_ln_kernel at all (shapes, 16-byte alignment of every stream)?  The forward and the backward ask the same question.
static bool ln_gemm_probe(const pfn_model_desc& d, const Ws& w, const float* params, const void* sh, int M) {
  GemmLN probe; memset(&probe, 0, sizeof(probe));
  probe.A = w.x0_t; probe.lda = d.emsize; probe.B = sh; probe.ldb = d.emsize; probe.M = M; probe.N = d.emsize; probe.K = d.emsize;
  probe.bias = params; probe.gamma = params; probe.beta = params; probe.resid = w.x0; probe.y = w.x0; probe.x_t = w.x0_t;
  return gemm_ln_supported(probe);
}

Ws carve(const pfn_model_desc& d, int B, int S, char* base) {
  Ws w;
  const int64_t M = (int64_t)B * S, E = d.emsize, F = d.nhid, es = esize(d.precision);
  const int64_t npad = align_up(d.n_out, 8);
  int64_t cur = 0;
  auto take = [&](int64_t nbytes) { char* p = base ? base + cur : nullptr; cur = align_up(cur + nbytes, 256); return p; };
  w.x0 = (float*)take(M * E * 4); w.x0_t = take(M * E * es);
  const int64_t aug = emb_aug_width(d.num_features) > 0 ? emb_aug_width(d.num_features) : 32;
  w.xaug_t = take(M * aug * es); w.embacc = (float*)take(E * aug * 4);
  w.layer.resize(d.nlayers);
  for (auto& l : w.layer) {
    l.qkv = take(M * 3 * E * es); l.ctx = take(M * E * es); l.lse = (float*)take((int64_t)B * d.nhead * S * 4);
    l.y1 = (float*)take(M * E * 4); l.mean1 = (float*)take(M * 4); l.rstd1 = (float*)take(M * 4);
    l.x1 = (float*)take(M * E * 4); l.x1_t = take(M * E * es);
    l.hpre = take(M * F * es); l.h = take(M * F * es);
    l.y2 = (float*)take(M * E * 4); l.mean2 = (float*)take(M * 4); l.rstd2 = (float*)take(M * 4);
    l.x2 = (float*)take(M * E * 4); l.x2_t = take(M * E * es);
    l.dy2_t = take(M * E * es); l.dh_t = take(M * F * es); l.dy1_t = take(M * E * es); l.dqkv_t = take(M * 3 * E * es);
    l.dy2m_t = l.dy1m_t = nullptr;
    if (d.dropout > 0.f) { l.dy2m_t = take(M * E * es); l.dy1m_t = take(M * E * es); }
  }
  w.xt_t = take(M * E * es); w.dpre = take(M * F * es); w.dt = take(M * F * es);
  w.dlog_t = take(M * npad * es); w.dd_t = take(M * F * es); w.dxt = (float*)take(M * E * 4);
  w.gA = (float*)take(M * E * 4); w.gA_t = take(M * E * es);
  w.dctx_t = take(M * E * es);
  w.delta = (float*)take(2 * (int64_t)B * d.nhead * S * 4);      // [delta | lse in log2 units], both [B,H,S] (attn_delta_kernel)
  w.ds = take(attn_bwd_ds_bytes(B, S, d.nhead, d.precision));
  // the top layer on the test rows only: used when the schedule allows it (no dropout, PFN_SCHED_TOP_LAYER_ALL_ROWS clear) and then for at most the
  // (S - sep) B <= 3/4 S B rows that 4 sep >= S leaves (top_layer_on_test_rows below) -- not carved otherwise
  const int64_t Mtop = (d.nlayers > 0 && d.dropout == 0.f && !(d.schedule & PFN_SCHED_TOP_LAYER_ALL_ROWS)) ? (int64_t)B * (S - (S + 3) / 4) : 0;
  w.top_ctx_t = take(Mtop * E * es); w.top_dy1_t = take(Mtop * E * es); w.top_dctx_t = take(Mtop * E * es);
  w.top_ry = (float*)take(Mtop * E * 4); w.top_rmean = (float*)take(Mtop * 4); w.top_rrstd = (float*)take(Mtop * 4);
  w.kshift = key_centering(d) ? (float*)take((int64_t)B * E * 4) : nullptr;
  w.lscale = (float*)take(256);
  w.ln_part = (d.schedule & PFN_SCHED_DETERMINISTIC) ? (float*)take((int64_t)LNB_MAX_BLOCKS * 3 * E * 4) : nullptr;
  w.bytes = cur;
  return w;
}

GemmNT nt(const void* A, long lda, const void* B, long ldb, int M, int N, int K, int flags) {
  GemmNT g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.M = M; g.N = N; g.K = K; g.flags = flags;
  return g;
}
GemmTN tn(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int P, int Q, float* colsum = nullptr) {
  GemmTN g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.P = P; g.Q = Q; g.atomic = 1; g.colsum = colsum;
  return g;
}

}  // namespace

extern "C" {

static bool g_gp_planes = true;      // PFN_TUNE_GP_PLANES (see pfn_gp_prior_sample)
static bool g_fuse_delta = true;     // PFN_TUNE_FUSE_DELTA: the attention backward's delta from the d(ctx) GEMM's epilogue (stack_backward_impl)
// Defaults handed to NEW descriptors by pfn_default_schedule() (test / profiling hook); the entry points read pfn_model_desc::schedule only.
static int g_default_schedule = 0;
// The reference returns output[single_eval_pos:] (transformer.py:91): the TOP encoder layer's train rows feed nothing -- no later layer reads
// them as keys, the decoder and the loss never see them, their gradient is zero.  So everything of that layer behind its K / V projection runs on
// the test rows only, in the decoder's compact row order: the attention for the queries >= sep, out_proj / LayerNorm / FFN on (S - sep) B rows,
// and the same in the backward (zero rows dropped from every product).  Same results row for row; at the north star (sep ~ 0.8 S) it removes
// ~80 % of one layer in six.  Off with dropout (the masks are indexed by the full-layout row) and when fewer than a quarter of the rows are train rows.
// (d.dropout, not the live probability: with dropout configured the top_* buffers are not carved -- an inference pass of such a model keeps every row)
static bool top_layer_on_test_rows(const pfn_model_desc& d, int S, int sep, float pdrop) {
  return !(d.schedule & PFN_SCHED_TOP_LAYER_ALL_ROWS) && d.nlayers > 0 && pdrop == 0.f && d.dropout == 0.f && sep < S && 4L * sep >= S;
}
// ... for a ragged batch: every dataset needs 4 sep_b >= S (the compact buffers hold 3/4 of the rows), i.e. the smallest position decides
static bool top_layer_on_test_rows_ragged(const pfn_model_desc& d, int S, const Ragged& rg, float pdrop) {
  return !(d.schedule & PFN_SCHED_TOP_LAYER_ALL_ROWS) && d.nlayers > 0 && pdrop == 0.f && d.dropout == 0.f && rg.test_rows > 0 && 4L * rg.sep_min >= S;
}
// emsize 1024: the 64-row fused kernels exist and are correct, but lose to GEMM + LayerNorm kernels (PFN_SCHED_FUSE_LN_WIDE)
int pfn_abi_version(void) { return PFN_ABI_VERSION; }
int pfn_default_schedule(void) { return g_default_schedule; }
int pfn_profile_enable(int on) { g_prof_on.store(on < 0 ? 0 : on); return PFN_OK; }
int pfn_profile_read(int slot, double* total_ms, int64_t* launches) {
  if (slot < 0 || slot >= PFN_PROF_SLOTS) return fail(PFN_ERR_ARGUMENT, "bad profile slot %d", slot);
  std::vector<ProfPair> pairs;
  {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    pairs.swap(g_prof[slot]);
  }
  double ms = 0.0;
  for (ProfPair& p : pairs) {
    float t = 0.f;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) ms += t;
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = (int64_t)pairs.size();
  return PFN_OK;
}
int pfn_set_tuning(int key, int value) {
  switch (key) {
    case PFN_TUNE_GEMM_NT_KERNEL: set_gemm_nt_big_mode(value); return PFN_OK;
    case PFN_TUNE_GEMM_TN_WRAP: set_gemm_tn_debug_wrap(value); return PFN_OK;
    case PFN_TUNE_FUSE_LNBWD: g_default_schedule = value ? (g_default_schedule & ~PFN_SCHED_SEPARATE_LNBWD) : (g_default_schedule | PFN_SCHED_SEPARATE_LNBWD); return PFN_OK;
    case PFN_TUNE_ATTN_PINGPONG: set_attn_pingpong(value); return PFN_OK;
    case PFN_TUNE_ATTN_BWD_GROUP: set_attn_bwd_group(value); return PFN_OK;
    case PFN_TUNE_WGRAD_SPLITS: set_gemm_tn_group_splits(value); return PFN_OK;
    case PFN_TUNE_WGRAD_WAVES: set_gemm_tn_group_waves(value); return PFN_OK;
    case PFN_TUNE_LOSS_SCALE_TARGET: return set_loss_scale_target(value);
    case PFN_TUNE_RESIDUAL16: g_default_schedule = value ? (g_default_schedule & ~PFN_SCHED_F32_RESIDUAL) : (g_default_schedule | PFN_SCHED_F32_RESIDUAL); return PFN_OK;
    case PFN_TUNE_FUSE_Q_PROJECTION: g_default_schedule = value ? (g_default_schedule | PFN_SCHED_FUSE_Q_PROJECTION) : (g_default_schedule & ~PFN_SCHED_FUSE_Q_PROJECTION); return PFN_OK;
    case PFN_TUNE_KEY_CENTERING:      // 1: on for both 16-bit formats, 0: off for both, -1: the defaults (on with fp16, off with bf16)
      g_default_schedule &= ~(PFN_SCHED_NO_KEY_CENTERING | PFN_SCHED_KEY_CENTERING);
      if (value > 0) g_default_schedule |= PFN_SCHED_KEY_CENTERING; else if (value == 0) g_default_schedule |= PFN_SCHED_NO_KEY_CENTERING;
      return PFN_OK;
    case PFN_TUNE_GEMM_LN_ROWS: set_gemm_ln_rows64(value); return PFN_OK;
    case PFN_TUNE_GP_PLANES: g_gp_planes = value != 0; return PFN_OK;
    case PFN_TUNE_FUSE_DELTA: g_fuse_delta = value != 0; return PFN_OK;
    case PFN_TUNE_FUSE_LN_WIDE: g_default_schedule = value ? (g_default_schedule | PFN_SCHED_FUSE_LN_WIDE) : (g_default_schedule & ~PFN_SCHED_FUSE_LN_WIDE); return PFN_OK;
    case PFN_TUNE_TOP_LAYER_TEST_ROWS: g_default_schedule = value ? (g_default_schedule & ~PFN_SCHED_TOP_LAYER_ALL_ROWS) : (g_default_schedule | PFN_SCHED_TOP_LAYER_ALL_ROWS); return PFN_OK;
    default: return fail(PFN_ERR_ARGUMENT, "unknown tuning key %d", key);
  }
}
const char* pfn_last_error_string(void) { return g_err; }

int pfn_param_layout(const pfn_model_desc* d, int64_t* offsets, int64_t* numels, int max_tensors) {
  int rc = check_desc(d);
  if (rc != PFN_OK) return rc;
  Layout L = make_layout(*d);
  const int n = (int)L.offsets.size();
  if (offsets || numels) {
    if (max_tensors < n) return fail(PFN_ERR_ARGUMENT, "need room for %d tensors, got %d", n, max_tensors);
    for (int i = 0; i < n; ++i) { if (offsets) offsets[i] = L.offsets[i]; if (numels) numels[i] = L.numels[i]; }
  }
  return n;
}
int64_t pfn_param_count(const pfn_model_desc* d) {
  if (check_desc(d) != PFN_OK) return -1;
  return make_layout(*d).total;
}
int64_t pfn_shadow_bytes(const pfn_model_desc* d) {
  if (check_desc(d) != PFN_OK) return -1;
  Layout L = make_layout(*d);
  return (L.total + L.total_t) * esize(d->precision);
}
int64_t pfn_workspace_bytes(const pfn_model_desc* d, int B, int S) {
  if (check_desc(d) != PFN_OK || B < 1 || S < 1) return -1;
  return carve(*d, B, S, nullptr).bytes;
}

int64_t pfn_top_layer_rows(const pfn_model_desc* d, int B, int S, int sep, int use_dropout) {
  if (check_desc(d) != PFN_OK || B < 1 || S < 1 || sep < 0 || sep > S) return -1;
  return top_layer_on_test_rows(*d, S, sep, use_dropout ? d->dropout : 0.f) ? (int64_t)(S - sep) * B : (int64_t)B * S;
}

int pfn_prepare_params(const pfn_model_desc* d, const float* params, void* shadow, void* stream) {
  PFN_TRY(check_desc(d));
  if (!params || !shadow) return fail(PFN_ERR_ARGUMENT, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  Layout L = make_layout(*d);
  const int prec = d->precision, es = esize(prec);
  const int E = d->emsize, F = d->nhid;
  char* sh = (char*)shadow;
  PFN_TRY(launch_cast_params(params, sh, L.total, prec, s));
  char* tr = sh + L.total * es;
  TransposeGroup g;      // all transposed copies in one launch per TRANSPOSE_GROUP_MAX matrices
  auto add = [&](long src_off, long dst_off, int rows, int cols, int ld_dst) -> int {
    if (g.n == TRANSPOSE_GROUP_MAX) {
      PFN_TRY(launch_transpose_cast_group(params, tr, g, prec, s));
      g = TransposeGroup();
    }
    transpose_group_add(g, src_off, dst_off, rows, cols, ld_dst);
    return PFN_OK;
  };
  for (int l = 0; l < d->nlayers; ++l) {
    const LayerP &p = L.layer[l], &t = L.layer_t[l];
    PFN_TRY(add(p.w_in, t.w_in, 3 * E, E, 3 * E));  // [3E,E] -> [E,3E]
    PFN_TRY(add(p.w_o, t.w_o, E, E, E));
    PFN_TRY(add(p.w1, t.w1, F, E, F));               // [F,E] -> [E,F]
    PFN_TRY(add(p.w2, t.w2, E, F, E));               // [E,F] -> [F,E]
  }
  if (d->n_out > 0) {
    PFN_TRY(add(L.dec0_w, L.dec0_wt, F, E, F));
    PFN_TRY(add(L.dec2_w, L.dec2_wt, d->n_out, F, L.n_out_pad));  // [O,F] -> [F,Opad]
  }
  return launch_transpose_cast_group(params, tr, g, prec, s);
}

static int stack_forward_impl(const pfn_model_desc* d, const float* params, const void* shadow,
                              const float* x, int64_t x_st, int64_t x_sb, const float* y, int64_t y_st, int64_t y_sb,
                              const float* src_sbe, int B, int S, int sep, void* workspace, int64_t workspace_bytes,
                              float* logits, void* stream, bool use_dropout, uint64_t dropout_seed, const Ragged* rg = nullptr);
int pfn_stack_forward(const pfn_model_desc* d, const float* params, const void* shadow,
                      const float* x, int64_t x_st, int64_t x_sb, const float* y, int64_t y_st, int64_t y_sb,
                      const float* src_sbe, int B, int S, int sep, void* workspace, int64_t workspace_bytes,
                      float* logits, void* stream) {
  return stack_forward_impl(d, params, shadow, x, x_st, x_sb, y, y_st, y_sb, src_sbe, B, S, sep, workspace, workspace_bytes, logits, stream, false, 0);
}
int pfn_stack_forward_dropout(const pfn_model_desc* d, const float* params, const void* shadow,
                              const float* x, int64_t x_st, int64_t x_sb, const float* y, int64_t y_st, int64_t y_sb,
                              const float* src_sbe, int B, int S, int sep, void* workspace, int64_t workspace_bytes,
                              float* logits, void* stream, uint64_t dropout_seed) {
  return stack_forward_impl(d, params, shadow, x, x_st, x_sb, y, y_st, y_sb, src_sbe, B, S, sep, workspace, workspace_bytes, logits, stream, true, dropout_seed);
}
static int stack_forward_impl(const pfn_model_desc* d, const float* params, const void* shadow,
                              const float* x, int64_t x_st, int64_t x_sb, const float* y, int64_t y_st, int64_t y_sb,
                              const float* src_sbe, int B, int S, int sep, void* workspace, int64_t workspace_bytes,
                              float* logits, void* stream, bool use_dropout, uint64_t dropout_seed, const Ragged* rg) {
  PFN_TRY(check_desc(d));
  const float pdrop = use_dropout ? d->dropout : 0.f;     // > 0: TransformerEncoderLayer's four dropout sites are live (training)
  auto dseed = [&](int layer, int site) { return dropout_site_seed(dropout_seed, layer, site); };
  if (!params || !shadow || !workspace) return fail(PFN_ERR_ARGUMENT, "null pointer");
  if (!src_sbe && (!x || !y)) return fail(PFN_ERR_ARGUMENT, "need x and y (or src_sbe)");
  if (B < 1 || S < 1 || sep < 0 || sep > S) return fail(PFN_ERR_ARGUMENT, "bad B=%d S=%d sep=%d", B, S, sep);
  if (!logits && sep < S) return fail(PFN_ERR_ARGUMENT, "null logits");
  hipStream_t s = (hipStream_t)stream;
  const int prec = d->precision, es = esize(prec);
  const int E = d->emsize, F = d->nhid, H = d->nhead, O = d->n_out;
  Layout L = make_layout(*d);
  Ws w = carve(*d, B, S, (char*)workspace);
  if (workspace_bytes < w.bytes) return fail(PFN_ERR_ARGUMENT, "workspace too small: %lld < %lld", (long long)workspace_bytes, (long long)w.bytes);
  // ragged batch (pfn_stack_forward_ragged): every dataset has its own eval position (rg->sep_of, device), `sep` is their maximum, the decoder's compact
  // rows are dataset-major (rg->row_off) and there are rg->test_rows of them; the top layer then runs on every row
  const int* sep_of = rg ? rg->sep_of : nullptr;
  if (rg && (src_sbe || rg->test_rows < 0 || rg->test_rows > (int64_t)B * S || !rg->sep_of || !rg->row_off)) return fail(PFN_ERR_ARGUMENT, "bad ragged-batch arguments");
  const int M = B * S, Mt = rg ? (int)rg->test_rows : (S - sep) * B;
  const char* sh = (const char*)shadow;
  auto W = [&](int64_t off) { return (const void*)(sh + off * es); };

  if (src_sbe) {
    PFN_TRY(launch_sbe_to_bse(src_sbe, w.x0, w.x0_t, S, B, E, prec, s));
  } else {
    EmbedArgs e;
    e.x = x; e.x_st = x_st; e.x_sb = x_sb; e.y = y; e.y_st = y_st; e.y_sb = y_sb;
    e.wx = params + L.enc_w; e.bx = params + L.enc_b; e.wy = params + L.yenc_w; e.by = params + L.yenc_b;
    e.out_f32 = w.x0; e.out_t = w.x0_t; e.S = S; e.B = B; e.nf = d->num_features; e.E = E; e.sep = sep; e.sep_of = sep_of;
    e.xaug_ld = emb_aug_width(d->num_features);
    e.xaug_t = e.xaug_ld > 0 ? w.xaug_t : nullptr;
    PFN_TRY(launch_embed_fwd(e, prec, s));
  }
  const float* xin = w.x0;
  const char* xin_t = w.x0_t;
  // out_proj and linear2 run fused with their residual add and LayerNorm (gemm_nt_ln_kernel) when the shape allows:
  // the f32 LayerNorm output is then never stored -- the next residual add recomputes it from the pre-LN sum and the
  // row statistics -- except after the last layer, whose f32 output feeds the decoder gather.
  // (dropout sits between the bias and the residual add: it takes the unfused GEMM / LayerNorm kernels with an element-wise pass between)
  const bool ln_ok = ln_gemm_probe(*d, w, params, sh, M);
  const bool fuse_ln = prec_is16(prec) && ln_ok && F % 32 == 0 && pdrop == 0.f && (E <= 512 || (d->schedule & PFN_SCHED_FUSE_LN_WIDE));
  const bool y16 = residual16(*d) && ln_ok;      // (implies fuse_ln: the descriptor's dropout is 0, so is pdrop)
  const bool y16u = residual16_separate_ln(*d, ln_ok && F % 32 == 0 && (E <= 512 || (d->schedule & PFN_SCHED_FUSE_LN_WIDE)));      // (implies !fuse_ln)
  struct Resid { const float* plain; const void* y; const float* mean; const float* rstd; const float* gamma; const float* beta; };
  Resid res = {w.x0, nullptr, nullptr, nullptr, nullptr, nullptr};   // where the layer input lives in f32
  auto set_resid = [](GemmLN& g, const Resid& r) {
    g.resid = r.plain; g.ry = r.y; g.rmean = r.mean; g.rrstd = r.rstd; g.rgamma = r.gamma; g.rbeta = r.beta;
  };
  const bool top_mode = rg ? top_layer_on_test_rows_ragged(*d, S, *rg, pdrop) : top_layer_on_test_rows(*d, S, sep, pdrop);
  // north_star's "QKV projection + attention as one kernel", the half that can exist (PFN_SCHED_FUSE_Q_PROJECTION; measured, not the default: DESIGN.md section 3)
  const bool fuse_q = (d->schedule & PFN_SCHED_FUSE_Q_PROJECTION) && pdrop == 0.f && attn_fwd_can_fuse_q(E, H, prec);
  // the top layer's row moves (token order -> the decoder's compact rows): (t - sep) B + b, or dataset-major for a ragged batch
  auto gather_top = [&](const void* src, void* dst, long row_bytes) {
    return rg ? launch_gather_rows_ragged(src, dst, S, B, row_bytes, rg->sep_of, (const long*)rg->row_off, s) : launch_gather_rows(src, dst, S, B, row_bytes, sep, s);
  };
  for (int l = 0; l < d->nlayers; ++l) {
    const LayerP& p = L.layer[l];
    LayerWs& a = w.layer[l];
    const bool top = top_mode && l == d->nlayers - 1;     // this layer's rows behind the K / V projection: the test rows only (compact order)
    const int Ml = top ? Mt : M;
    {  // packed q/k/v projection
      ProfScope ps(PFN_PROF_GEMM_QKV, s);
      GemmNT g = nt(xin_t, E, W(p.w_in), E, M, 3 * E, E, EPI_BIAS | EPI_OUT_T);
      g.bias = params + p.b_in; g.out_t = a.qkv; g.ld_out_t = 3 * E;
      if (fuse_q) {      // PFN_SCHED_FUSE_Q_PROJECTION: only k | v here, q inside the attention kernel (below)
        g.B = W(p.w_in + (int64_t)E * E); g.N = 2 * E; g.bias = params + p.b_in + E; g.out_t = a.qkv + (int64_t)E * es;
      }
      // 16-bit operands: the keys leave centred per dataset, k' = k - W_k xbar (pfn_kernels.h launch_key_shift: the attention output and every gradient are those of
      // the uncentred keys, the operand rounding of K is 9 x smaller on a trained model).  The shift is taken in f32 inside this GEMM's epilogue.
      if (w.kshift && E % 64 == 0) {
        PFN_TRY(launch_key_shift(xin_t, W(p.w_in + (int64_t)E * E), w.kshift, B, S, E, sep, sep_of, prec, s));
        g.flags |= EPI_ROWSHIFT; g.rowshift = w.kshift; g.rs_ld = E; g.rs_S = S; g.rs_n0 = fuse_q ? 0 : E; g.rs_n1 = g.rs_n0 + E;
      }
      PFN_TRY(launch_gemm_nt(g, prec, s));
    }
    {
      AttnArgs at; memset(&at, 0, sizeof(at));
      at.qkv = a.qkv; at.ctx = a.ctx; at.lse = a.lse; at.B = B; at.S = S; at.E = E; at.H = H; at.sep = sep; at.sep_of = sep_of;
      at.p_drop = pdrop; at.drop_seed = dseed(l, 0);
      at.q_begin = top ? (rg ? rg->sep_min : sep) : 0;
      at.q_from_sep = (top && rg) ? 1 : 0;
      if (fuse_q) { at.xq = xin_t; at.wq = W(p.w_in); at.bq = params + p.b_in; at.q_store = 1; }      // (q_store: the backward reads Q from qkv)
      PFN_TRY(launch_attn_fwd(at, prec, s));
    }
    const char* ctx_in = a.ctx;
    if (top) {      // the test rows of the attention output and of the layer input, gathered
      PFN_TRY(gather_top(a.ctx, w.top_ctx_t, (long)E * es));
      ctx_in = w.top_ctx_t;
      if (fuse_ln && !res.plain) {
        PFN_TRY(gather_top(res.y, w.top_ry, (long)E * (y16 ? es : 4)));
        PFN_TRY(gather_top(res.mean, w.top_rmean, 4));
        PFN_TRY(gather_top(res.rstd, w.top_rrstd, 4));
        res = Resid{nullptr, w.top_ry, w.top_rmean, w.top_rrstd, res.gamma, res.beta};
      } else {
        if (y16u) PFN_TRY(gather_top(xin_t, w.top_ry, (long)E * es));      // (the residual of this layer's out_proj is read in operand precision)
        else PFN_TRY(gather_top(fuse_ln ? res.plain : xin, w.top_ry, (long)E * 4));
        res = Resid{w.top_ry, nullptr, nullptr, nullptr, nullptr, nullptr};
        xin = w.top_ry;
      }
    }
    float* x2_f32 = top ? (O == 0 ? logits : nullptr) : a.x2;     // the stack's f32 output rows: only what the decoder gather (or the caller) reads
    if (fuse_ln) {  // x1 = LN1(x + out_proj(ctx))
      ProfScope ps(PFN_PROF_GEMM_OUT_LN + (top ? 1 : 0), s);
      GemmLN g; memset(&g, 0, sizeof(g));
      g.A = ctx_in; g.lda = E; g.B = W(p.w_o); g.ldb = E; g.M = Ml; g.N = E; g.K = E; g.bias = params + p.b_o;
      set_resid(g, res);
      g.gamma = params + p.g1; g.beta = params + p.be1; g.eps = d->ln_eps;
      g.y = a.y1; g.mean = a.mean1; g.rstd = a.rstd1; g.x_t = a.x1_t; g.y16 = y16;
      PFN_TRY(launch_gemm_ln(g, prec, s));
      res = Resid{nullptr, a.y1, a.mean1, a.rstd1, params + p.g1, params + p.be1};
    } else {
      {  // out_proj + residual  (dropout1: the product leaves alone and the element-wise pass adds the residual)
        GemmNT g = nt(ctx_in, E, W(p.w_o), E, Ml, E, E, EPI_BIAS | (pdrop > 0.f ? 0 : EPI_RESID) | EPI_OUT_F32);
        g.bias = params + p.b_o; g.resid = xin; g.ld_resid = E; g.out_f32 = a.y1; g.ld_out_f32 = E;
        if (y16u) {      // residual from the operand-precision layer input (the gathered test rows in top mode), sum out in operand precision
          g.flags = EPI_BIAS | EPI_RESID_T | EPI_OUT_T; g.aux = top ? (const void*)w.top_ry : (const void*)xin_t; g.ld_aux = E; g.out_t = a.y1; g.ld_out_t = E;
        }
        PFN_TRY(launch_gemm_nt(g, prec, s));
        if (pdrop > 0.f) PFN_TRY(launch_dropout_add(a.y1, xin, M, E, dseed(l, 1), pdrop, s));
      }
      PFN_TRY(launch_layernorm_fwd(a.y1, params + p.g1, params + p.be1, y16u ? nullptr : a.x1, a.x1_t, a.mean1, a.rstd1, Ml, E, d->ln_eps, prec, s, y16u));
    }
    {  // linear1 + GELU (pre-activation kept for the backward)
      ProfScope ps(PFN_PROF_GEMM_LIN1 + (top ? 1 : 0), s);
      GemmNT g = nt(a.x1_t, E, W(p.w1), E, Ml, F, E, EPI_BIAS | EPI_GELU | EPI_OUT_T | EPI_OUT2_T);
      g.bias = params + p.b1; g.out_t = a.h; g.ld_out_t = F; g.out2_t = a.hpre; g.ld_out2 = F;
      PFN_TRY(launch_gemm_nt(g, prec, s));
      // FFN dropout: h and the stored GELU derivative take the same mask, so linear2, its weight gradient and d(hpre) need nothing more
      if (pdrop > 0.f) PFN_TRY(launch_dropout_scale(a.h, a.h, a.hpre, a.hpre, M, F, dseed(l, 2), pdrop, prec, s));
    }
    if (fuse_ln) {  // x2 = LN2(x1 + linear2(h))
      ProfScope ps(PFN_PROF_GEMM_LIN2_LN + (top ? 1 : 0), s);
      GemmLN g; memset(&g, 0, sizeof(g));
      g.A = a.h; g.lda = F; g.B = W(p.w2); g.ldb = F; g.M = Ml; g.N = E; g.K = F; g.bias = params + p.b2;
      set_resid(g, res);
      g.gamma = params + p.g2; g.beta = params + p.be2; g.eps = d->ln_eps;
      g.y = a.y2; g.mean = a.mean2; g.rstd = a.rstd2; g.x_t = a.x2_t; g.y16 = y16;
      g.x_f32 = (l == d->nlayers - 1) ? x2_f32 : nullptr;
      PFN_TRY(launch_gemm_ln(g, prec, s));
      res = Resid{nullptr, a.y2, a.mean2, a.rstd2, params + p.g2, params + p.be2};
    } else {
      {  // linear2 + residual  (dropout2 as above)
        GemmNT g = nt(a.h, F, W(p.w2), F, Ml, E, F, EPI_BIAS | (pdrop > 0.f ? 0 : EPI_RESID) | EPI_OUT_F32);
        g.bias = params + p.b2; g.resid = a.x1; g.ld_resid = E; g.out_f32 = a.y2; g.ld_out_f32 = E;
        if (y16u) { g.flags = EPI_BIAS | EPI_RESID_T | EPI_OUT_T; g.aux = a.x1_t; g.ld_aux = E; g.out_t = a.y2; g.ld_out_t = E; }
        PFN_TRY(launch_gemm_nt(g, prec, s));
        if (pdrop > 0.f) PFN_TRY(launch_dropout_add(a.y2, a.x1, M, E, dseed(l, 3), pdrop, s));
      }
      // (fp16 sums: only the last layer's f32 output has a reader -- the decoder gather, or the caller when there is no decoder)
      float* x2_out = top ? x2_f32 : ((y16u && l < d->nlayers - 1) ? nullptr : a.x2);
      PFN_TRY(launch_layernorm_fwd(a.y2, params + p.g2, params + p.be2, x2_out, a.x2_t, a.mean2, a.rstd2, Ml, E, d->ln_eps, prec, s, y16u));
    }
    xin = a.x2; xin_t = a.x2_t;
  }
  // decoder on the test rows only (the reference decodes all rows, then slices: transformer.py:85,91)
  // (top_mode: the top layer already ran on exactly these rows, in this order -- its operand-precision output IS the decoder's input, its f32
  // output went straight to the caller when there is no decoder)
  if (Mt > 0 && O == 0) {
    // no decoder (a custom decoder module runs in PyTorch, reference transformer.py:23): hand out the test rows [Mt, E] in f32
    if (top_mode) {}
    else if (rg) PFN_TRY(launch_gather_test_rows_ragged(xin, logits, S, B, E, rg->sep_of, (const long*)rg->row_off, PFN_PREC_F32, s));
    else PFN_TRY(launch_gather_test_rows(xin, logits, S, B, E, sep, PFN_PREC_F32, s));
  } else if (Mt > 0) {
    if (top_mode) {}
    else if (rg) PFN_TRY(launch_gather_test_rows_ragged(xin, w.xt_t, S, B, E, rg->sep_of, (const long*)rg->row_off, prec, s));
    else PFN_TRY(launch_gather_test_rows(xin, w.xt_t, S, B, E, sep, prec, s));
    const char* xt_t = top_mode ? w.layer[d->nlayers - 1].x2_t : w.xt_t;
    {
      GemmNT g = nt(xt_t, E, W(L.dec0_w), E, Mt, F, E, EPI_BIAS | EPI_GELU | EPI_OUT_T | EPI_OUT2_T);
      g.bias = params + L.dec0_b; g.out_t = w.dt; g.ld_out_t = F; g.out2_t = w.dpre; g.ld_out2 = F;
      PFN_TRY(launch_gemm_nt(g, prec, s));
    }
    {
      GemmNT g = nt(w.dt, F, W(L.dec2_w), F, Mt, O, F, EPI_BIAS | EPI_OUT_F32);
      g.bias = params + L.dec2_b; g.out_f32 = logits; g.ld_out_f32 = O;
      PFN_TRY(launch_gemm_nt(g, prec, s));
    }
  }
  return PFN_OK;
}

int pfn_stack_backward(const pfn_model_desc* d, const float* params, const void* shadow,
                       const float* x, int64_t x_st, int64_t x_sb, const float* y, int64_t y_st, int64_t y_sb,
                       int B, int S, int sep, void* workspace, int64_t workspace_bytes,
                       const float* dlogits, float* grads, float* dsrc_sbe, void* stream) {
  return pfn_stack_backward_split(d, params, shadow, x, x_st, x_sb, y, y_st, y_sb, B, S, sep, workspace, workspace_bytes, dlogits, grads, dsrc_sbe,
                                  stream, 0, nullptr, nullptr, 0, 0);
}

static int stack_backward_impl(const pfn_model_desc* d, const float* params, const void* shadow,
                               const float* x, int64_t x_st, int64_t x_sb, const float* y, int64_t y_st, int64_t y_sb,
                               int B, int S, int sep, void* workspace, int64_t workspace_bytes,
                               const float* dlogits, float* grads, float* dsrc_sbe, void* stream,
                               int first_group_layers, pfn_host_callback on_first_group, void* user, int use_dropout, uint64_t dropout_seed, const Ragged* rg);
int pfn_stack_backward_split(const pfn_model_desc* d, const float* params, const void* shadow,
                             const float* x, int64_t x_st, int64_t x_sb, const float* y, int64_t y_st, int64_t y_sb,
                             int B, int S, int sep, void* workspace, int64_t workspace_bytes,
                             const float* dlogits, float* grads, float* dsrc_sbe, void* stream,
                             int first_group_layers, pfn_host_callback on_first_group, void* user, int use_dropout, uint64_t dropout_seed) {
  return stack_backward_impl(d, params, shadow, x, x_st, x_sb, y, y_st, y_sb, B, S, sep, workspace, workspace_bytes, dlogits, grads, dsrc_sbe, stream,
                             first_group_layers, on_first_group, user, use_dropout, dropout_seed, nullptr);
}
// ---- the micro-batches of one optimizer step as ONE launch set (round 5): reference train.py:66-97 runs the k batches of an optimizer step one after the other, each
// with its own single_eval_pos; a batch of 4 datasets fills a fraction of the chip, so the datasets of several batches are stacked along B here and every kernel that
// looks at the eval position reads its dataset's own (sep_of [B], device).  Compact test rows (decoder input, logits, dlogits): dataset-major, row_off[b] + (t - sep_of[b]).
// sep_min / sep_max = min / max(sep_of) (grids, scratch dims, whether the top layer can run on the test rows only: 4 sep_min >= S); test_rows = row_off[B] =
// sum(S - sep_of[b]).  Fused embedding only.
int pfn_stack_forward_ragged(const pfn_model_desc* d, const float* params, const void* shadow,
                             const float* x, int64_t x_st, int64_t x_sb, const float* y, int64_t y_st, int64_t y_sb,
                             int B, int S, const int32_t* sep_of, const int64_t* row_off, int sep_min, int sep_max, int64_t test_rows,
                             void* workspace, int64_t workspace_bytes, float* logits, void* stream, int use_dropout, uint64_t dropout_seed) {
  if (sep_min < 0 || sep_min > sep_max) return fail(PFN_ERR_ARGUMENT, "bad sep_min=%d sep_max=%d", sep_min, sep_max);
  Ragged rg = {sep_of, row_off, test_rows, sep_min};
  return stack_forward_impl(d, params, shadow, x, x_st, x_sb, y, y_st, y_sb, nullptr, B, S, sep_max, workspace, workspace_bytes, logits, stream, use_dropout != 0, dropout_seed, &rg);
}
int pfn_stack_backward_ragged(const pfn_model_desc* d, const float* params, const void* shadow,
                              const float* x, int64_t x_st, int64_t x_sb, const float* y, int64_t y_st, int64_t y_sb,
                              int B, int S, const int32_t* sep_of, const int64_t* row_off, int sep_min, int sep_max, int64_t test_rows,
                              void* workspace, int64_t workspace_bytes, const float* dlogits, float* grads, void* stream,
                              int first_group_layers, pfn_host_callback on_first_group, void* user, int use_dropout, uint64_t dropout_seed) {
  if (sep_min < 0 || sep_min > sep_max) return fail(PFN_ERR_ARGUMENT, "bad sep_min=%d sep_max=%d", sep_min, sep_max);
  Ragged rg = {sep_of, row_off, test_rows, sep_min};
  return stack_backward_impl(d, params, shadow, x, x_st, x_sb, y, y_st, y_sb, B, S, sep_max, workspace, workspace_bytes, dlogits, grads, nullptr, stream,
                             first_group_layers, on_first_group, user, use_dropout, dropout_seed, &rg);
}
static int stack_backward_impl(const pfn_model_desc* d, const float* params, const void* shadow,
                               const float* x, int64_t x_st, int64_t x_sb, const float* y, int64_t y_st, int64_t y_sb,
                               int B, int S, int sep, void* workspace, int64_t workspace_bytes,
                               const float* dlogits, float* grads, float* dsrc_sbe, void* stream,
                               int first_group_layers, pfn_host_callback on_first_group, void* user, int use_dropout, uint64_t dropout_seed, const Ragged* rg) {
  PFN_TRY(check_desc(d));
  const float pdrop = use_dropout ? d->dropout : 0.f;
  const bool top_mode = rg ? top_layer_on_test_rows_ragged(*d, S, *rg, pdrop) : top_layer_on_test_rows(*d, S, sep, pdrop);       // (the forward took the same decision: same descriptor, shape, dropout)
  const int* sep_of = rg ? rg->sep_of : nullptr;
  if (rg && (dsrc_sbe || rg->test_rows < 0 || rg->test_rows > (int64_t)B * S || !rg->sep_of || !rg->row_off)) return fail(PFN_ERR_ARGUMENT, "bad ragged-batch arguments");
  // PFN_SCHED_DETERMINISTIC: one writer per gradient element and launch -- no token splits in the weight-gradient GEMMs, the LayerNorm backward as its own
  // kernel with ordered partial sums, the embedding gradient from one workgroup per column block
  const bool det = (d->schedule & PFN_SCHED_DETERMINISTIC) != 0;
  const float* lsc = nullptr;      // fp16 operands: the device float the loss scale is derived from (set below, once the workspace is carved); else no scaling
  auto tn_det = [&](GemmTN g) { if (det) g.max_splits = 1; g.scale_amax = lsc; return g; };
  auto dseed = [&](int layer, int site) { return dropout_site_seed(dropout_seed, layer, site); };
  if (!params || !shadow || !workspace || !grads) return fail(PFN_ERR_ARGUMENT, "null pointer");
  if (!dsrc_sbe && (!x || !y)) return fail(PFN_ERR_ARGUMENT, "need x and y (or dsrc_sbe)");
  if (B < 1 || S < 1 || sep < 0 || sep > S) return fail(PFN_ERR_ARGUMENT, "bad B=%d S=%d sep=%d", B, S, sep);
  hipStream_t s = (hipStream_t)stream;
  const int prec = d->precision, es = esize(prec);
  const int E = d->emsize, F = d->nhid, H = d->nhead, O = d->n_out;
  Layout L = make_layout(*d);
  Ws w = carve(*d, B, S, (char*)workspace);
  if (workspace_bytes < w.bytes) return fail(PFN_ERR_ARGUMENT, "workspace too small");
  const int M = B * S, Mt = rg ? (int)rg->test_rows : (S - sep) * B, npad = L.n_out_pad;
  const char* sh = (const char*)shadow;
  auto W = [&](int64_t off) { return (const void*)(sh + off * es); };
  auto WT = [&](int64_t off) { return (const void*)(sh + (L.total + off) * es); };
  const bool ln_ok = ln_gemm_probe(*d, w, params, sh, M);
  // the forward stored the pre-LayerNorm sums in operand precision (same rules, same pointers): inside the LayerNorm-fused GEMMs, or from a GEMM ahead of layernorm_fwd
  const bool y16 = (residual16(*d) && ln_ok) || residual16_separate_ln(*d, ln_ok && F % 32 == 0 && (E <= 512 || (d->schedule & PFN_SCHED_FUSE_LN_WIDE)));

  // fp16 operands: the backward chain runs on dlogits * 2^k, k from max|dlogits| on the device; every kernel that writes a parameter gradient takes 2^k out again
  if (prec == PFN_PREC_FP16 && Mt > 0) {
    if (!dlogits) return fail(PFN_ERR_ARGUMENT, "null dlogits");
    PFN_TRY(launch_absmax(dlogits, (long)Mt * (O > 0 ? O : E), w.lscale, s));
    lsc = w.lscale;
  }
  // ---- decoder ----
  const float* dxt = w.dxt;
  const char* xt_t = top_mode ? w.layer[d->nlayers - 1].x2_t : w.xt_t;      // the decoder's input rows (forward)
  if (Mt > 0 && O == 0) {
    if (!dlogits) return fail(PFN_ERR_ARGUMENT, "null dlogits");
    dxt = dlogits;   // no decoder: the incoming gradient already is d(test rows) [Mt, E]
    if (lsc) { PFN_TRY(launch_scale_copy(dlogits, w.dxt, (long)Mt * E, lsc, s)); dxt = w.dxt; }
  } else if (Mt > 0) {
    if (!dlogits) return fail(PFN_ERR_ARGUMENT, "null dlogits");
    PFN_TRY(launch_cast_rows(dlogits, O, w.dlog_t, npad, Mt, O, prec, s, lsc));
    // the decoder's two weight gradients: one grouped launch of 256 x 256 tiles when the shapes allow (bars padded to a multiple of
    // 256 by the zero columns of dlog_t: Pv bounds the rows that exist), else a split-K launch each
    TnProblem dp2; memset(&dp2, 0, sizeof(dp2));
    dp2.A = w.dlog_t; dp2.lda = npad; dp2.B = w.dt; dp2.ldb = F; dp2.C = grads + L.dec2_w; dp2.ldc = F; dp2.P = npad; dp2.Q = F; dp2.Pv = O;
    dp2.colsum = grads + L.dec2_b;
    TnProblem dp0; memset(&dp0, 0, sizeof(dp0));
    dp0.A = w.dd_t; dp0.lda = F; dp0.B = xt_t; dp0.ldb = E; dp0.C = grads + L.dec0_w; dp0.ldc = E; dp0.P = F; dp0.Q = E; dp0.colsum = grads + L.dec0_b;
    const bool dec_grouped = prec_is16(prec) && gemm_tn_group_supported(dp2) && gemm_tn_group_supported(dp0);
    if (!dec_grouped) PFN_TRY(launch_gemm_tn(tn_det(tn(w.dlog_t, npad, w.dt, F, grads + L.dec2_w, F, Mt, O, F, grads + L.dec2_b)), prec, s));
    {
      // (contraction over the zero-padded width when that is whole 64-deep stages: the LDS-DMA kernel then takes it)
      GemmNT g = nt(w.dlog_t, npad, WT(L.dec2_wt), npad, Mt, F, npad % 64 == 0 ? npad : O, EPI_GELU_BWD | EPI_OUT_T);
      g.aux = w.dpre; g.ld_aux = F; g.out_t = w.dd_t; g.ld_out_t = F;
      PFN_TRY(launch_gemm_nt(g, prec, s));
    }
    if (dec_grouped) {
      GemmTNGroup g; memset(&g, 0, sizeof(g));
      g.n = 2; g.M = Mt; g.p[0] = dp2; g.p[1] = dp0; g.splits = det ? 1 : 0; g.scale_amax = lsc;
      PFN_TRY(launch_gemm_tn_group(g, prec, s));
    } else {
      PFN_TRY(launch_gemm_tn(tn_det(tn(w.dd_t, F, xt_t, E, grads + L.dec0_w, E, Mt, F, E, grads + L.dec0_b)), prec, s));
    }
    {
      GemmNT g = nt(w.dd_t, F, WT(L.dec0_wt), F, Mt, E, F, EPI_OUT_F32);
      g.out_f32 = w.dxt; g.ld_out_f32 = E;
      PFN_TRY(launch_gemm_nt(g, prec, s));
    }
  }
  // (top_mode: the top layer's backward runs on the compact test rows and takes dxt as it is)
  if (top_mode) {}
  else if (rg) PFN_TRY(launch_scatter_test_rows_ragged(dxt, d->nlayers > 0 ? (void*)w.gA_t : (void*)w.gA, S, B, E, rg->sep_of, (const long*)rg->row_off, d->nlayers > 0 ? prec : PFN_PREC_F32, s));
  else PFN_TRY(launch_scatter_test_rows(dxt, d->nlayers > 0 ? (void*)w.gA_t : (void*)w.gA, S, B, E, sep, d->nlayers > 0 ? prec : PFN_PREC_F32, s));

  // ---- encoder layers, last to first; gA holds d(loss)/d(layer output) ----
  // Only the data-gradient chain runs here.  Each layer leaves the output-gradient operands of its four
  // weight gradients (dy2, dh, dy1, dqkv) in its own buffers; all 4*nlayers weight (and fused bias)
  // gradients are then computed by ONE grouped launch of 256x256 tiles (gemm_tn_big_kernel) -- enough
  // tiles to fill the chip without splitting the token axis into hundreds of atomic partial sums.
  // The two GEMMs whose output is the gradient w.r.t. a LayerNorm output (dx1 -> LN1, dx -> the previous layer's LN2) run that
  // LayerNorm's backward in their epilogue (gemm_nt_lnbwd_kernel) when the shape allows: the sum never reaches HBM, and the
  // bias gradient of the Linear in front of the LayerNorm moves to the weight-gradient GEMM that reads the same operand.
  auto lnb = [&](const void* A, long lda, const void* Bw, long ldb, int K, const void* aux, const void* y, const float* mean, const float* rstd,
                 const float* gamma, void* dx_t, float* dgamma, float* dbeta, int rows) {
    GemmLNB g; memset(&g, 0, sizeof(g));
    g.A = A; g.lda = lda; g.B = Bw; g.ldb = ldb; g.M = rows; g.N = E; g.K = K; g.aux = aux;
    g.y = y; g.mean = mean; g.rstd = rstd; g.gamma = gamma; g.dx_t = dx_t; g.dgamma = dgamma; g.dbeta = dbeta; g.scale_amax = lsc; g.y16 = y16;
    return g;
  };
  // The embedding's weight gradients d(src)^T . [x | masked y | train flag] are a (skinny) weight-gradient GEMM like the others:
  // embed_fwd left the augmented inputs in operand precision, the first layer's dx leaves in operand precision, and the
  // split-K TN kernel does the rest (the register kernel it replaces streamed d(src) at 0.7 TB/s).  Custom encoders
  // (dsrc_sbe), the exact-f32 mode and wide encoders keep the f32 path.
  const int aug = emb_aug_width(d->num_features);
  const bool emb_gemm = !dsrc_sbe && prec_is16(prec) && d->nlayers > 0 && aug > 0 && E % 8 == 0;
  bool fuse_lnb = !det && !(d->schedule & PFN_SCHED_SEPARATE_LNBWD) && prec_is16(prec) && d->nlayers > 0 && pdrop == 0.f && (E <= 512 || (d->schedule & PFN_SCHED_FUSE_LN_WIDE));   // (dropout: masked and unmasked LayerNorm-input gradients both exist)
  if (fuse_lnb) {
    const LayerP& p = L.layer[0]; const LayerP& t = L.layer_t[0]; LayerWs& a = w.layer[0];
    fuse_lnb = gemm_lnbwd_supported(lnb(a.dh_t, F, WT(t.w1), F, F, a.dy2_t, a.y1, a.mean1, a.rstd1, params + p.g1, a.dy1_t, grads + p.g1, grads + p.be1, M)) &&
               gemm_lnbwd_supported(lnb(a.dqkv_t, 3 * E, WT(t.w_in), 3 * E, 3 * E, a.dy1_t, a.y2, a.mean2, a.rstd2, params + p.g2, a.dy2_t, grads + p.g2, grads + p.be2, M));
  }
  // ---- weight gradients of the layers [l_lo, l_hi] as one grouped launch (called once after the chain, or -- data-parallel runs,
  // pfn_stack_backward_split -- once for the top layers in the middle of the chain and once for the rest) ----
  auto launch_weight_gradients = [&](int l_hi, int l_lo) -> int {
    std::vector<TnProblem> probs, probs_top;      // contraction over all B S tokens / over the test rows (a top layer on the test rows)
    auto add = [&](const void* A, long lda, const void* Bm, long ldb, float* C, long ldc, int P, int Q, float* colsum, bool compact = false) {
      TnProblem t; memset(&t, 0, sizeof(t)); t.A = A; t.lda = lda; t.B = Bm; t.ldb = ldb; t.C = C; t.ldc = ldc; t.P = P; t.Q = Q; t.colsum = colsum;
      (compact ? probs_top : probs).push_back(t);
    };
    for (int l = l_hi; l >= l_lo; --l) {
      const LayerP& p = L.layer[l];
      LayerWs& a = w.layer[l];
      const char* xin_t = (l == 0) ? w.x0_t : w.layer[l - 1].x2_t;
      // (b2 / b_o: column sums of dy2 / dy1 -- from the LayerNorm-backward kernel when that ran on its own)
      const bool drop = pdrop > 0.f;
      const bool top = top_mode && l == d->nlayers - 1;      // its FFN / out_proj operands hold the test rows only
      add(drop ? a.dy2m_t : a.dy2_t, E, a.h, F, grads + p.w2, F, E, F, (drop || (fuse_lnb && l < d->nlayers - 1)) ? grads + p.b2 : nullptr, top);
      add(a.dh_t, F, a.x1_t, E, grads + p.w1, E, F, E, grads + p.b1, top);
      add(drop ? a.dy1m_t : (top ? w.top_dy1_t : a.dy1_t), E, top ? w.top_ctx_t : a.ctx, E, grads + p.w_o, E, E, E, (drop || fuse_lnb) ? grads + p.b_o : nullptr, top);
      add(a.dqkv_t, 3 * E, xin_t, E, grads + p.w_in, E, 3 * E, E, grads + p.b_in);
    }
    if (!probs_top.empty()) {
      bool grouped_top = prec_is16(prec);
      for (const TnProblem& t : probs_top) grouped_top = grouped_top && gemm_tn_group_supported(t);
      if (grouped_top) {
        ProfScope ps(PFN_PROF_WGRAD + 1, s);
        GemmTNGroup g;
        memset(&g, 0, sizeof(g));
        g.n = (int)probs_top.size();
        g.M = Mt;
        g.splits = det ? 1 : 0;
        g.scale_amax = lsc;
        for (int i = 0; i < g.n; ++i) g.p[i] = probs_top[i];
        PFN_TRY(launch_gemm_tn_group(g, prec, s));
      } else {
        for (const TnProblem& t : probs_top)
          PFN_TRY(launch_gemm_tn(tn_det(tn(t.A, t.lda, t.B, t.ldb, t.C, t.ldc, Mt, t.P, t.Q, t.colsum)), prec, s));
      }
    }
    bool grouped = prec_is16(prec);
    for (const TnProblem& t : probs) grouped = grouped && gemm_tn_group_supported(t);
    if (grouped) {
      for (size_t i0 = 0; i0 < probs.size(); i0 += TN_GROUP_MAX) {
        ProfScope ps(PFN_PROF_WGRAD, s);
        GemmTNGroup g;
        memset(&g, 0, sizeof(g));
        g.n = (int)std::min<size_t>(TN_GROUP_MAX, probs.size() - i0);
        g.M = M;
        g.splits = det ? 1 : 0;
        g.scale_amax = lsc;
        for (int i = 0; i < g.n; ++i) g.p[i] = probs[i0 + i];
        PFN_TRY(launch_gemm_tn_group(g, prec, s));
      }
    } else {  // exact-f32 parity mode and shapes outside the 256-tile kernel: one split-K launch per gradient
      for (const TnProblem& t : probs)
        PFN_TRY(launch_gemm_tn(tn_det(tn(t.A, t.lda, t.B, t.ldb, t.C, t.ldc, M, t.P, t.Q, t.colsum)), prec, s));
    }
    return PFN_OK;
  };
  const int split_at = (on_first_group && first_group_layers > 0 && first_group_layers < d->nlayers) ? d->nlayers - first_group_layers : -1;
  bool delta_zeroed = false;      // the delta scratch holds zeros for the next EPI_ROWDOT epilogue (d(ctx) below)
  for (int l = d->nlayers - 1; l >= 0; --l) {
    const LayerP &p = L.layer[l], &t = L.layer_t[l];
    LayerWs& a = w.layer[l];
    const bool top = top_mode && l == d->nlayers - 1;      // the chain of this layer down to d(attention output) runs on the test rows (compact order)
    const int Ml = top ? Mt : M;
    char* dy1_t = top ? w.top_dy1_t : a.dy1_t;
    // LN2: the input gradient leaves only in operand precision (dy2_t); it is both the GEMM operand below and the
    // residual-branch gradient that the dx1 GEMM adds back, so no f32 copy is written or re-read.  (Fused: the layer above
    // already left dy2_t.)
    // dropout: the gradient entering linear2 (dropout2) / out_proj (dropout1) is the LayerNorm-input gradient times that site's mask;
    // the residual path keeps the unmasked one, and the two bias gradients become column sums of the masked operands (weight-gradient launch)
    if (!fuse_lnb || l == d->nlayers - 1)
      PFN_TRY(launch_layernorm_bwd(top ? (const void*)dxt : (const void*)w.gA_t, top ? 0 : 1, a.y2, params + p.g2, a.mean2, a.rstd2, nullptr, a.dy2_t,
                                   grads + p.g2, grads + p.be2, pdrop > 0.f ? nullptr : grads + p.b2, Ml, E, prec, s, w.ln_part, lsc, y16));
    const char* dy2_op = a.dy2_t;
    if (pdrop > 0.f) { PFN_TRY(launch_dropout_scale(a.dy2_t, a.dy2m_t, nullptr, nullptr, M, E, dseed(l, 3), pdrop, prec, s)); dy2_op = a.dy2m_t; }
    {  // d(hpre) = (dy2 . W2) * gelu'(hpre)
      ProfScope ps(PFN_PROF_GEMM_DHPRE + (top ? 1 : 0), s);
      GemmNT g = nt(dy2_op, E, WT(t.w2), E, Ml, F, E, EPI_GELU_BWD | EPI_OUT_T);
      g.aux = a.hpre; g.ld_aux = F; g.out_t = a.dh_t; g.ld_out_t = F;
      PFN_TRY(launch_gemm_nt(g, prec, s));
    }
    if (fuse_lnb) {  // dy1 = LN1 backward of (dh . W1 + dy2)
      ProfScope ps(PFN_PROF_GEMM_DY1 + (top ? 1 : 0), s);
      PFN_TRY(launch_gemm_lnbwd(lnb(a.dh_t, F, WT(t.w1), F, F, a.dy2_t, a.y1, a.mean1, a.rstd1, params + p.g1, dy1_t, grads + p.g1, grads + p.be1, Ml), prec, s));
    } else {
      {  // dx1 = dh . W1 + dy2
        GemmNT g = nt(a.dh_t, F, WT(t.w1), F, Ml, E, F, EPI_RESID_T | EPI_OUT_T);
        g.aux = a.dy2_t; g.ld_aux = E; g.out_t = w.gA_t; g.ld_out_t = E;
        PFN_TRY(launch_gemm_nt(g, prec, s));
      }
      PFN_TRY(launch_layernorm_bwd(w.gA_t, 1, a.y1, params + p.g1, a.mean1, a.rstd1, nullptr, dy1_t, grads + p.g1, grads + p.be1,
                                   pdrop > 0.f ? nullptr : grads + p.b_o, Ml, E, prec, s, w.ln_part, lsc, y16));
    }
    const char* dy1_op = dy1_t;
    bool delta_fused = false;
    if (pdrop > 0.f) { PFN_TRY(launch_dropout_scale(a.dy1_t, a.dy1m_t, nullptr, nullptr, M, E, dseed(l, 1), pdrop, prec, s)); dy1_op = a.dy1m_t; }
    {  // d(ctx) = dy1 . Wo
      ProfScope ps(PFN_PROF_GEMM_DCTX + (top ? 1 : 0), s);
      GemmNT g = nt(dy1_op, E, WT(t.w_o), E, Ml, E, E, EPI_OUT_T);
      g.out_t = top ? w.top_dctx_t : w.dctx_t; g.ld_out_t = E;
      // The attention backward's delta = rowsum(dO . O) leaves with d(ctx) from this GEMM's epilogue (EPI_ROWDOT) instead of a pass of its own over dO and O
      // (attn_delta_kernel: 1.4 % of the step's kernel time) -- on the full-sequence layers of the default schedule; the test-row top layer (compact rows), the
      // deterministic schedule (head dim 256: four atomic addends per element) and the shapes the LDS-DMA kernels do not take keep the kernel.
      if (!top && !det && g_fuse_delta) {
        GemmNT gd = g;
        gd.flags |= EPI_ROWDOT; gd.aux = a.ctx; gd.ld_aux = E;
        gd.rowdot = w.delta; gd.rd_lse = a.lse; gd.rd_lse2_off = (long)B * H * S; gd.rd_S = S; gd.rd_H = H; gd.rd_D = E / H;
        if (gemm_nt_rowdot_fused(gd, prec)) {
          // the scratch is zero: once per backward pass by a memset, afterwards by every layer's query-block pass (AttnArgs::zero_delta), which runs after the
          // last reader of the values it clears
          if (!delta_zeroed && hipMemsetAsync(w.delta, 0, sizeof(float) * B * H * S, s) != hipSuccess) return fail(PFN_ERR_LAUNCH, "hipMemsetAsync(delta)");
          delta_zeroed = true;
          g = gd; delta_fused = true;
        }
      }
      PFN_TRY(launch_gemm_nt(g, prec, s));
    }
    if (top) {
      // back to the token order for the attention backward and for the dx product of the K / V projection: d(attention output) is needed from the first
      // query block the attention kernels touch (AttnArgs::q_begin: zeros up to sep), the LayerNorm-input gradient as the residual term of every row
      if (rg) {
        PFN_TRY(launch_scatter_rows_ragged(w.top_dctx_t, w.dctx_t, S, B, (long)E * es, rg->sep_of, (const long*)rg->row_off, 1, s));
        PFN_TRY(launch_scatter_rows_ragged(w.top_dy1_t, a.dy1_t, S, B, (long)E * es, rg->sep_of, (const long*)rg->row_off, 0, s));
      } else {
        PFN_TRY(launch_scatter_rows(w.top_dctx_t, w.dctx_t, S, B, (long)E * es, sep, sep / 256 * 256, s));
        PFN_TRY(launch_scatter_rows(w.top_dy1_t, a.dy1_t, S, B, (long)E * es, sep, 0, s));
      }
    }
    {
      AttnArgs at; memset(&at, 0, sizeof(at));
      at.qkv = a.qkv; at.ctx = a.ctx; at.lse = a.lse; at.B = B; at.S = S; at.E = E; at.H = H; at.sep = sep; at.sep_of = sep_of;
      at.dctx = w.dctx_t; at.dqkv = a.dqkv_t; at.delta = w.delta; at.ds = w.ds;
      at.p_drop = pdrop; at.drop_seed = dseed(l, 0);
      at.q_begin = top ? (rg ? rg->sep_min : sep) : 0;
      at.q_from_sep = (top && rg) ? 1 : 0;
      if (delta_fused) at.parts = ATTN_BWD_KV | ATTN_BWD_DQ;      // delta and lse2 are in place (EPI_ROWDOT above)
      at.zero_delta = delta_zeroed ? 1 : 0;
      PFN_TRY(launch_attn_bwd(at, prec, s));
    }
    if (fuse_lnb && l > 0) {  // dy2 of the layer below = its LN2 backward of (dqkv . Win + dy1)
      ProfScope ps(PFN_PROF_GEMM_DX, s);
      const LayerP& pb = L.layer[l - 1];
      LayerWs& ab = w.layer[l - 1];
      PFN_TRY(launch_gemm_lnbwd(lnb(a.dqkv_t, 3 * E, WT(t.w_in), 3 * E, 3 * E, a.dy1_t, ab.y2, ab.mean2, ab.rstd2, params + pb.g2, ab.dy2_t,
                                    grads + pb.g2, grads + pb.be2, M), prec, s));
    } else {  // dx = dqkv . Win + dy1
      // the gradient stays in operand precision between layers; the embedding's gradient (layer 0) too when its weight
      // gradients are computed as a GEMM (emb_gemm below), else it leaves in f32
      GemmNT g = nt(a.dqkv_t, 3 * E, WT(t.w_in), 3 * E, M, E, 3 * E, EPI_RESID_T | (l == 0 && !emb_gemm ? EPI_OUT_F32 : EPI_OUT_T));
      g.aux = a.dy1_t; g.ld_aux = E; g.out_f32 = w.gA; g.ld_out_f32 = E; g.out_t = w.gA_t; g.ld_out_t = E;
      PFN_TRY(launch_gemm_nt(g, prec, s));
    }
    if (l == split_at) {
      // the chain has left the top layers: their four operand sets are complete, and so is every LayerNorm / bias gradient of theirs
      PFN_TRY(launch_weight_gradients(d->nlayers - 1, split_at));
      on_first_group(user);
    }
  }
  // ---- weight gradients of every layer not launched yet ----
  if (d->nlayers > 0) PFN_TRY(launch_weight_gradients(split_at >= 0 ? split_at - 1 : d->nlayers - 1, 0));
  // ---- embedding ----
  if (dsrc_sbe) {
    PFN_TRY(launch_bse_to_sbe(w.gA, dsrc_sbe, S, B, E, s, lsc));
  } else if (emb_gemm) {
    if (hipMemsetAsync(w.embacc, 0, sizeof(float) * E * aug, s) != hipSuccess) return fail(PFN_ERR_LAUNCH, "memset");
    GemmTN g = tn(w.gA_t, E, w.xaug_t, aug, w.embacc, aug, M, E, aug, grads + L.enc_b);
    g.scale_amax = lsc;
    g.max_splits = det ? 1 : 128;      // a 512 x 32 result: measured 67 / 62 / 88 us with 64 / 128 / 256 splits (the partial sums are added atomically)
    PFN_TRY(launch_gemm_tn(g, prec, s));
    PFN_TRY(launch_embed_grad_scatter(w.embacc, grads + L.enc_w, grads + L.yenc_w, grads + L.yenc_b, E, d->num_features, s));
  } else {
    EmbedBwdArgs e;
    e.dsrc = w.gA; e.x = x; e.x_st = x_st; e.x_sb = x_sb; e.y = y; e.y_st = y_st; e.y_sb = y_sb;
    e.dwx = grads + L.enc_w; e.dbx = grads + L.enc_b; e.dwy = grads + L.yenc_w; e.dby = grads + L.yenc_b;
    e.S = S; e.B = B; e.nf = d->num_features; e.E = E; e.sep = sep; e.sep_of = sep_of; e.single_block = det ? 1 : 0; e.scale_amax = lsc;
    PFN_TRY(launch_embed_bwd(e, s));
  }
  return PFN_OK;
}

int pfn_bar_nll_forward(const float* logits, int64_t ld, const float* y, const float* borders, int64_t R, int nbars,
                        int full_support, float* nll, float* lse, int32_t* bucket, void* stream) {
  if (R < 0 || nbars < 1 || (R > 0 && (!logits || !y || !borders || !nll || !lse || !bucket))) return fail(PFN_ERR_ARGUMENT, "bad bar_nll_forward arguments");
  BarArgs a; memset(&a, 0, sizeof(a));
  a.logits = logits; a.ld = ld; a.y = y; a.borders = borders; a.R = R; a.nbars = nbars; a.full_support = full_support;
  a.nll = nll; a.lse = lse; a.bucket = bucket;
  PFN_TRY(launch_bar_nll_fwd(a, (hipStream_t)stream));
  return PFN_OK;
}
int pfn_bar_nll_backward(const float* logits, int64_t ld, const float* lse, const int32_t* bucket, const float* gout,
                         int64_t R, int nbars, float* dlogits, void* stream) {
  if (R < 0 || nbars < 1 || (R > 0 && (!logits || !lse || !bucket || !gout || !dlogits))) return fail(PFN_ERR_ARGUMENT, "bad bar_nll_backward arguments");
  BarArgs a; memset(&a, 0, sizeof(a));
  a.logits = logits; a.ld = ld; a.R = R; a.nbars = nbars; a.lse = const_cast<float*>(lse); a.bucket = const_cast<int*>(bucket);
  a.gout = gout; a.dlogits = dlogits;
  PFN_TRY(launch_bar_nll_bwd(a, (hipStream_t)stream));
  return PFN_OK;
}
int pfn_bar_mean(const float* logits, int64_t ld, const float* borders, int64_t R, int nbars, int full_support, float* mean, void* stream) {
  if (R < 0 || nbars < 1 || (R > 0 && (!logits || !borders || !mean))) return fail(PFN_ERR_ARGUMENT, "bad bar_mean arguments");
  BarArgs a; memset(&a, 0, sizeof(a));
  a.logits = logits; a.ld = ld; a.borders = borders; a.R = R; a.nbars = nbars; a.full_support = full_support; a.mean_out = mean;
  PFN_TRY(launch_bar_mean(a, (hipStream_t)stream));
  return PFN_OK;
}

int pfn_clip_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                       float beta2, float eps, float max_norm, float grad_scale, int step, int zero_grad, float* scratch, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !scratch || n < 0 || step < 1) return fail(PFN_ERR_ARGUMENT, "bad clip_adam arguments");
  AdamArgs a;
  a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.n = n; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.max_norm = max_norm; a.grad_scale = grad_scale; a.step = step; a.zero_grad = zero_grad; a.scratch = scratch;
  PFN_TRY(launch_clip_adam(a, (hipStream_t)stream));
  return PFN_OK;
}

// PFN_TUNE_GP_PLANES (test / profiling knob): 1 (default) = the rank-256 trailing update of the blocked Cholesky multiplies the pre-split fp16 planes the wide
// triangular solve leaves behind K (gp_syrk_planes_kernel: two terms on a power-of-two scale, three products); 0 = it re-reads the f32 panel per tile and splits it
// into three bf16 terms, six products (gp_syrk_kernel, rounds 2-3).  Both are f32-accurate (tools/sim_gp_split.py, tools/exp_gp_accuracy.py).
int64_t pfn_gp_workspace_bytes(int B, int S) {
  if (B < 1 || S < 1) return -1;
  return gp_workspace_bytes(B, S);
}
static int gp_check_workspace(GpArgs& a, int64_t K_ws_bytes) {      // ABI 7: the planes only when the caller's allocation holds them
  a.planes = nullptr; a.plane_rows = 0;
  if (K_ws_bytes < (int64_t)a.B * a.S * a.S * 4) return fail(PFN_ERR_ARGUMENT, "K_ws_bytes %lld < B*S*S*4 = %lld", (long long)K_ws_bytes, (long long)a.B * a.S * a.S * 4);
  if (g_gp_planes && K_ws_bytes >= gp_workspace_bytes(a.B, a.S)) gp_attach_planes(a);
  return PFN_OK;
}
int pfn_gp_prior_sample(float* x, float* z, float* y, float* K_ws, int64_t K_ws_bytes, const float* lengthscale, const float* outputscale,
                        const float* noise, int B, int S, int nf, int kernel, int gen_x, int gen_z, uint64_t seed, uint64_t offset,
                        int32_t* info, void* stream) {
  if (!x || !z || !y || !K_ws || !lengthscale || !outputscale || !noise || !info || B < 1 || S < 1 || nf < 1) return fail(PFN_ERR_ARGUMENT, "bad gp_prior_sample arguments");
  if (kernel < 0 || kernel > 3) return fail(PFN_ERR_UNSUPPORTED, "kernel %d (0 = RBF, 1 / 2 / 3 = Matern nu 5/2, 3/2, 1/2)", kernel);
  GpArgs a;
  a.x = x; a.z = z; a.y = y; a.K = K_ws; a.lengthscale = lengthscale; a.outputscale = outputscale; a.noise = noise;
  a.B = B; a.S = S; a.nf = nf; a.kernel = kernel; a.seed = seed; a.offset = offset; a.gen_x = gen_x; a.gen_z = gen_z; a.info = info;
  a.w = nullptr;
  if (int rc = gp_check_workspace(a, K_ws_bytes)) return rc;
  PFN_TRY(launch_gp_sample(a, (hipStream_t)stream));
  return PFN_OK;
}

int pfn_gp_posterior(const float* x, const float* y, float* K_ws, int64_t K_ws_bytes, float* resid_ws, float* w_ws, const float* lengthscale,
                     const float* outputscale, const float* noise, int B, int S, int nf, int kernel, float* nll, float* mean,
                     float* var, int32_t* info, void* stream) {
  if (!x || !y || !K_ws || !resid_ws || !w_ws || !lengthscale || !outputscale || !noise || !info || B < 1 || S < 1 || nf < 1)
    return fail(PFN_ERR_ARGUMENT, "bad gp_posterior arguments");
  if (kernel < 0 || kernel > 3) return fail(PFN_ERR_UNSUPPORTED, "kernel %d (0 = RBF, 1 / 2 / 3 = Matern nu 5/2, 3/2, 1/2)", kernel);
  GpArgs a;
  a.x = const_cast<float*>(x); a.z = nullptr; a.y = resid_ws; a.K = K_ws; a.lengthscale = lengthscale; a.outputscale = outputscale;
  a.noise = noise; a.B = B; a.S = S; a.nf = nf; a.kernel = kernel; a.seed = 0; a.offset = 0; a.gen_x = 0; a.gen_z = 0; a.info = info;
  a.w = w_ws;
  if (int rc = gp_check_workspace(a, K_ws_bytes)) return rc;
  PFN_TRY(launch_gp_posterior(a, y, nll, mean, var, (hipStream_t)stream));
  return PFN_OK;
}

int pfn_mlp_prior_forward(const float* weights, const float* biases, const int32_t* model_of, const int32_t* dims, const float* noise_std,
                          float* causes, const float* noise, float* y, float* hidden, int B, int T, int HP, int Lmax, int activation, int gen_causes,
                          uint64_t seed, uint64_t offset, void* stream) {
  if (!weights || !biases || !model_of || !dims || !noise_std || !causes || !y) return fail(PFN_ERR_ARGUMENT, "bad mlp_prior_forward arguments");
  if (activation < 0 || activation > 3) return fail(PFN_ERR_UNSUPPORTED, "activation %d (0 identity, 1 relu, 2 tanh, 3 sigmoid)", activation);
  MlpPriorArgs a;
  a.weights = weights; a.biases = biases; a.model_of = model_of; a.dims = dims; a.noise_std = noise_std; a.causes = causes; a.noise = noise; a.y = y; a.hidden = hidden;
  a.B = B; a.T = T; a.HP = HP; a.Lmax = Lmax; a.activation = activation; a.gen_causes = gen_causes; a.seed = seed; a.offset = offset;
  PFN_TRY(launch_mlp_prior(a, (hipStream_t)stream));
  return PFN_OK;
}

// ---- single-op entry points -------------------------------------------------------------------------
int pfn_op_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, int flags, const float* bias,
                   const void* aux, int64_t ld_aux, const float* resid, int64_t ld_resid, float* out_f32, int64_t ld_out_f32,
                   void* out_t, int64_t ld_out_t, void* out2_t, int64_t ld_out2, int prec, void* stream) {
  GemmNT g = nt(A, lda, B, ldb, M, N, K, flags);
  g.bias = bias; g.aux = aux; g.ld_aux = ld_aux; g.resid = resid; g.ld_resid = ld_resid; g.out_f32 = out_f32; g.ld_out_f32 = ld_out_f32;
  g.out_t = out_t; g.ld_out_t = ld_out_t; g.out2_t = out2_t; g.ld_out2 = ld_out2;
  PFN_TRY(launch_gemm_nt(g, prec, (hipStream_t)stream));
  return PFN_OK;
}
int pfn_op_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int M, int P, int Q, int atomic, int prec, void* stream) {
  GemmTN g = tn(A, lda, B, ldb, C, ldc, M, P, Q);
  g.atomic = atomic;
  PFN_TRY(launch_gemm_tn(g, prec, (hipStream_t)stream));
  return PFN_OK;
}
int pfn_op_gemm_tn_group(int n, const void* const* A, const int64_t* lda, const void* const* B, const int64_t* ldb, float* const* C,
                         const int64_t* ldc, const int32_t* P, const int32_t* Q, float* const* colsum, int M, int splits, int prec, void* stream) {
  if (n < 1 || n > TN_GROUP_MAX || !A || !lda || !B || !ldb || !C || !ldc || !P || !Q) return fail(PFN_ERR_ARGUMENT, "bad gemm_tn_group arguments");
  GemmTNGroup g;
  memset(&g, 0, sizeof(g));
  g.n = n; g.M = M; g.splits = splits;
  for (int i = 0; i < n; ++i) {
    TnProblem& t = g.p[i];
    t.A = A[i]; t.lda = lda[i]; t.B = B[i]; t.ldb = ldb[i]; t.C = C[i]; t.ldc = ldc[i]; t.P = P[i]; t.Q = Q[i];
    t.colsum = colsum ? colsum[i] : nullptr;
  }
  PFN_TRY(launch_gemm_tn_group(g, prec, (hipStream_t)stream));
  return PFN_OK;
}
int pfn_op_gemm_ln(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, const float* bias,
                   const float* resid, const float* ry, const float* rmean, const float* rrstd, const float* rgamma, const float* rbeta,
                   const float* gamma, const float* beta, float eps, float* y, float* mean, float* rstd, void* x_t, int prec, void* stream) {
  GemmLN g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.M = M; g.N = N; g.K = K; g.bias = bias; g.resid = resid;
  g.ry = ry; g.rmean = rmean; g.rrstd = rrstd; g.rgamma = rgamma; g.rbeta = rbeta; g.gamma = gamma; g.beta = beta; g.eps = eps;
  g.y = y; g.mean = mean; g.rstd = rstd; g.x_t = x_t; g.y16 = (prec & PFN_OP_SUMS_16BIT) ? 1 : 0;
  PFN_TRY(launch_gemm_ln(g, prec & ~PFN_OP_SUMS_16BIT, (hipStream_t)stream));
  return PFN_OK;
}
int pfn_op_gemm_lnbwd(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, const void* aux,
                      const float* y, const float* mean, const float* rstd, const float* gamma,
                      void* dx_t, float* dgamma, float* dbeta, int prec, void* stream) {
  GemmLNB g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.M = M; g.N = N; g.K = K; g.aux = aux;
  g.y = y; g.mean = mean; g.rstd = rstd; g.gamma = gamma; g.dx_t = dx_t; g.dgamma = dgamma; g.dbeta = dbeta; g.y16 = (prec & PFN_OP_SUMS_16BIT) ? 1 : 0;
  PFN_TRY(launch_gemm_lnbwd(g, prec & ~PFN_OP_SUMS_16BIT, (hipStream_t)stream));
  return PFN_OK;
}
int pfn_op_attention_fwd(const void* qkv, void* ctx, float* lse, int B, int S, int E, int H, int sep, int prec, void* stream) {
  AttnArgs at; memset(&at, 0, sizeof(at));
  at.qkv = qkv; at.ctx = ctx; at.lse = lse; at.B = B; at.S = S; at.E = E; at.H = H; at.sep = sep;
  PFN_TRY(launch_attn_fwd(at, prec, (hipStream_t)stream));
  return PFN_OK;
}
int pfn_op_attention_fwd_from(const void* qkv, void* ctx, float* lse, int B, int S, int E, int H, int sep, int q_begin, int prec, void* stream) {
  AttnArgs at; memset(&at, 0, sizeof(at));
  at.qkv = qkv; at.ctx = ctx; at.lse = lse; at.B = B; at.S = S; at.E = E; at.H = H; at.sep = sep; at.q_begin = q_begin;
  PFN_TRY(launch_attn_fwd(at, prec, (hipStream_t)stream));
  return PFN_OK;
}
int pfn_op_attention_bwd_from(const void* qkv, const void* ctx, const float* lse, const void* dctx, void* dqkv, float* delta_ws, void* ds_ws,
                              int B, int S, int E, int H, int sep, int q_begin, int prec, int parts, void* stream) {
  AttnArgs at; memset(&at, 0, sizeof(at));
  at.qkv = qkv; at.ctx = const_cast<void*>(ctx); at.lse = const_cast<float*>(lse); at.B = B; at.S = S; at.E = E; at.H = H; at.sep = sep;
  at.dctx = dctx; at.dqkv = dqkv; at.delta = delta_ws; at.ds = ds_ws; at.parts = parts; at.q_begin = q_begin;
  PFN_TRY(launch_attn_bwd(at, prec, (hipStream_t)stream));
  return PFN_OK;
}
int pfn_op_gather_rows(const void* src_bs, void* dst_tb, int B, int S, int64_t row_bytes, int sep, void* stream) {
  if (!src_bs || !dst_tb || B < 1 || S < 1 || sep < 0 || sep > S) return fail(PFN_ERR_ARGUMENT, "bad gather_rows arguments");
  PFN_TRY(launch_gather_rows(src_bs, dst_tb, S, B, row_bytes, sep, (hipStream_t)stream));
  return PFN_OK;
}
int pfn_op_scatter_rows(const void* src_tb, void* dst_bs, int B, int S, int64_t row_bytes, int sep, int zero_from, void* stream) {
  if (!src_tb || !dst_bs || B < 1 || S < 1 || sep < 0 || sep > S) return fail(PFN_ERR_ARGUMENT, "bad scatter_rows arguments");
  PFN_TRY(launch_scatter_rows(src_tb, dst_bs, S, B, row_bytes, sep, zero_from, (hipStream_t)stream));
  return PFN_OK;
}
int64_t pfn_op_attention_bwd_ws_bytes(int B, int S, int H, int prec) {
  if (B < 1 || S < 1 || H < 1) return -1;
  return attn_bwd_ds_bytes(B, S, H, prec);
}
int pfn_op_attention_bwd(const void* qkv, const void* ctx, const float* lse, const void* dctx, void* dqkv, float* delta_ws, void* ds_ws,
                         int B, int S, int E, int H, int sep, int prec, int parts, void* stream) {
  AttnArgs at; memset(&at, 0, sizeof(at));
  at.qkv = qkv; at.ctx = const_cast<void*>(ctx); at.lse = const_cast<float*>(lse); at.B = B; at.S = S; at.E = E; at.H = H; at.sep = sep;
  at.dctx = dctx; at.dqkv = dqkv; at.delta = delta_ws; at.ds = ds_ws; at.parts = parts;
  PFN_TRY(launch_attn_bwd(at, prec, (hipStream_t)stream));
  return PFN_OK;
}
int pfn_op_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y_f32, void* y_t, float* mean, float* rstd,
                         int64_t rows, int E, float eps, int prec, void* stream) {
  PFN_TRY(launch_layernorm_fwd(x, gamma, beta, y_f32, y_t, mean, rstd, rows, E, eps, prec, (hipStream_t)stream));
  return PFN_OK;
}
int pfn_op_layernorm_bwd(const void* dy, int dy_is_t, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx_f32,
                         void* dx_t, float* dgamma, float* dbeta, float* dbias_extra, int64_t rows, int E, int prec, void* stream) {
  PFN_TRY(launch_layernorm_bwd(dy, dy_is_t, x, gamma, mean, rstd, dx_f32, dx_t, dgamma, dbeta, dbias_extra, rows, E, prec, (hipStream_t)stream));
  return PFN_OK;
}
int pfn_op_qkv_projection(const void* x_t, const void* w_in_t, const float* b_in, void* qkv_t, float* kshift_ws,
                          int B, int S, int E, int sep, const int32_t* sep_of, int center, int prec, void* stream) {
  if (!x_t || !w_in_t || !b_in || !qkv_t || B < 1 || S < 1 || E < 8 || sep < 0 || sep > S) return fail(PFN_ERR_ARGUMENT, "bad qkv_projection arguments");
  GemmNT g = nt(x_t, E, w_in_t, E, B * S, 3 * E, E, EPI_BIAS | EPI_OUT_T);
  g.bias = b_in; g.out_t = qkv_t; g.ld_out_t = 3 * E;
  if (center) {
    if (!prec_is16(prec) || E % 64 || !kshift_ws) return fail(PFN_ERR_UNSUPPORTED, "key centring: 16-bit operands, emsize a multiple of 64, scratch of B * E floats");
    PFN_TRY(launch_key_shift(x_t, (const char*)w_in_t + (int64_t)E * E * esize(prec), kshift_ws, B, S, E, sep, sep_of, prec, (hipStream_t)stream));
    g.flags |= EPI_ROWSHIFT; g.rowshift = kshift_ws; g.rs_ld = E; g.rs_S = S; g.rs_n0 = E; g.rs_n1 = 2 * E;
  }
  PFN_TRY(launch_gemm_nt(g, prec, (hipStream_t)stream));
  return PFN_OK;
}
int pfn_op_cast(const float* src, void* dst, int64_t n, int prec, void* stream) {
  PFN_TRY(launch_cast_params(src, dst, n, prec, (hipStream_t)stream));
  return PFN_OK;
}

}  // extern "C"
