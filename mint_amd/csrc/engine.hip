// FACT engine: C-ABI + model orchestration (see include/fact_hip.h).
//
// Replaces, for the hot path only, mint/core/{fact_model,base_models}.py and the tape/optimizer
// section of mint/ctl/single_task_trainer.py:141-196.  Host logic here only sequences kernels on
// the caller's stream; every FLOP runs in the HIP kernels of gemm.hip / attention.hip / rowops.hip.
//
// HBM layout
//   fp32 arenas (param / grad / adam m / adam v): tensors in Keras trainable_variables order,
//     Dense kernels [in][out], each tensor offset aligned to 64 floats.
//   bf16 weight shadows, two per Dense kernel:  s = [in][out_pad]  (dgrad B operand, k = out)
//                                               t = [out][in_pad]  (fwd   B operand, k = in)
//   activations: residual stream fp32 [B*n][d]; GEMM operands bf16; per-head q/k/v/dO in both
//     token-major [B*H][NP][DHP] and head-dim-major [B*H][DH][NP] forms (attention.h).
#include "../../include/fact_hip.h"
#include "../../include/fact_hip_debug.h"
#include "attention.h"
#include "gemm.h"
#include "rowops.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cxxabi.h>
#include <string>
#include <vector>

namespace {

// bf16 row pitches are multiples of 64 elements (128 B): a 64-deep K-step chunk of a row is then
// exactly one cache line (an 800-wide row at pitch 800 straddles two lines on every odd row, which
// costs ~10 % of the K = 800 GEMMs - tools/attic/gemm_bench.py).
constexpr int kPitch = 64;

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define CHK(expr)                                                                      \
  do {                                                                                 \
    int _rc = (expr);                                                                  \
    if (_rc != 0) return fail(_rc, std::string(#expr) + " failed rc=" + std::to_string(_rc)); \
  } while (0)
#define HIPCHK(expr)                                                                        \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) return fail(-100, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

inline int rup(int x, int m) { return (x + m - 1) / m * m; }
inline size_t rups(size_t x, size_t m) { return (x + m - 1) / m * m; }

struct Tensor {
  size_t off = 0;
  int rows = 0, cols = 0;
  size_t numel() const { return (size_t)rows * cols; }
};

struct DenseW {
  Tensor w;           // [in][out]
  bf16_t* s = nullptr;  // [in][lds]
  bf16_t* t = nullptr;  // [out][ldt]
  int lds = 0, ldt = 0;
};

struct LayerP {
  Tensor ln1_g, ln1_b, bo, ln2_g, ln2_b, b1, b2;
  DenseW wqkv, wo, w1, w2;
};

struct LayerA {
  float *x_in = nullptr, *x_mid = nullptr, *x_out = nullptr;
  float *mean1 = nullptr, *rstd1 = nullptr, *mean2 = nullptr, *rstd2 = nullptr;
  bf16_t *h1 = nullptr, *a = nullptr, *h2 = nullptr, *pre = nullptr, *g = nullptr;
  bf16_t* row[3] = {nullptr, nullptr, nullptr};
  float* lse2 = nullptr;
  bool h1_ready = false;  // transient: the previous layer's FFN2 epilogue pass already wrote h1 / mean1 / rstd1 (ln_fuse)
};

struct Stack {
  const char* name = "";
  int n = 0, feat = 0, featp = 0, d = 0, H = 0, dh = 0, dhp = 0, ff = 0, L = 0, NP = 0;
  int dp = 0, qp = 0, fp = 0;  // bf16 row pitches of [.][d], [.][3d], [.][ff] activations (multiples of 64)
  std::vector<LayerP> lp;
  std::vector<LayerA> la;
  // embedding (modal stacks only)
  DenseW emb;
  Tensor emb_b, pos;
  bf16_t* xin16 = nullptr;  // [B*n][featp]
  float* x0 = nullptr;      // stack input  [B*n][d]
  float* out() { return L ? la[L - 1].x_out : x0; }
};

struct Bump {
  char* base = nullptr;
  size_t off = 0;
  template <typename T>
  T* take(size_t count) {
    off = rups(off, 256);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

struct Bucket {
  size_t off = 0, cnt = 0;  // float range in the arenas
  int desc_first = 0, desc_n = 0, tiles = 0;
  int blk_first = 0, blk_n = 0;  // AdamBlock range of the fused optimizer + shadow-refresh kernel
  // the same range WITHOUT the Dense kernels of transformer layers (their update runs in the epilogue of the grouped wgrad
  // launch when FactHandle::wg_adam is set): biases, LayerNorm, position tables, embedding / head kernels
  int lite_first = 0, lite_n = 0;
  size_t lite_floats = 0;
};
struct AdamArgs {
  float lr_t = 0.f, b1 = 0.9f, b2 = 0.999f, eps = 1e-7f, gscale = 1.f;
};

// Scratch of ONE backward chain (dgrad / attention / LayerNorm sequence of a stack).  The cross-modal and
// audio stacks use chain 0 on the caller's stream; the motion stack runs at the same time on the handle's
// third stream with chain 1 (its GEMMs at 1920 tokens fill a fifth of the workgroup slots on their own).
// Buffers that the wgrad stream reads are double-buffered by layer parity (see layer_backward).
constexpr int kBwBuf = 3;  // layer-parity depth of every buffer the wgrad stream reads (see layer_backward)

// Work of one layer that only the optimizer waits for (weight / bias / LayerNorm gradients), recorded at the
// end of the layer's dgrad chain and launched on the wgrad stream a little later (flush_batch).
struct PendingBatch {
  bool valid = false;
  int sr_rows = 0, sr_pad = 0;  // > 0: supervised-rows layer - the compact buffers of SrBuf hold dpre / xmid16 / xin16 / dh2
  int sr_B = 0, sr_T = 0;
  Stack* st = nullptr;
  int l = 0, M = 0, q = 0;
  const bf16_t *xin16 = nullptr, *dpre = nullptr, *xmid16 = nullptr, *dqkv = nullptr, *dh2 = nullptr, *dh1 = nullptr;
  const float *part2 = nullptr, *part1 = nullptr;  // ln_cs: column-sum partials of LayerNorm 2 / 1 (null: col_tasks path)
  int part_blocks = 0;
  hipStream_t s = nullptr, w = nullptr;
  float* slab = nullptr;
};

// Supervised-rows shortcut (training only).  The loss reads the first T output tokens of every sequence
// (fact_model.py:143-148), so in the LAST cross-modal layer only those B*T rows are needed after the attention's
// K/V projections: attention queries, to_out, LayerNorm 2, the MLP and the head run on a compact (B*T)-row copy,
// and in backward the same rows carry all of the gradient (everything else of dL/dx_out is exactly zero), so the
// MLP / to_out dgrads and the weight gradients of W2, W1, Wo contract over B*T instead of B*n tokens.  Loss and
// every gradient are those of the full computation up to fp32 summation order; fact_forward (inference, all n
// rows returned) never takes the shortcut.  Rows are padded to a multiple of 64 with zero rows (wgrad K).
struct SrBuf {
  bf16_t *a_c = nullptr, *h2_c = nullptr, *pre_c = nullptr, *g_c = nullptr, *xf16_c = nullptr;
  bf16_t *dx16_c = nullptr, *dpre_c = nullptr, *dh2_c = nullptr, *xmid16_c = nullptr, *dpred_c = nullptr;
  float *x_in_c = nullptr, *x_mid_c = nullptr, *x_out_c = nullptr, *mean2_c = nullptr, *rstd2_c = nullptr;
  float *pred_c = nullptr, *dx_c = nullptr;
  int rows_max = 0;
};

struct BwScratch {
  bf16_t* dh = nullptr;      // [M][d pitch] dgrad output feeding the LayerNorm backward (fused-kernel path)
  bf16_t* dorow = nullptr;   // per-head dO rows
  float* dsum = nullptr;     // rowsum(dO o O)
  float* ln_ws = nullptr;    // LayerNorm-backward per-block partial column sums (fused-kernel path)
  // buffers both the dgrad chain and the wgrad stream touch, indexed by layer parity q = layer count % 3
  bf16_t* dpre_pp[kBwBuf] = {};   // [M][ff pitch]
  bf16_t* dqkv_pp[kBwBuf] = {};   // [M][3d pitch]
  bf16_t* xmid_pp[kBwBuf] = {};   // bf16 gradient at x_mid
  bf16_t* xb_pp[kBwBuf] = {};     // bf16 gradient at the layer boundary (output of the layer of parity q)
  bf16_t* dh2_pp[kBwBuf] = {};    // dgrad outputs feeding the two LayerNorm backward kernels (also read by the
  bf16_t* dh1_pp[kBwBuf] = {};    //   parameter-gradient column sums on the wgrad stream)
  float* lnpart_pp[kBwBuf][2] = {};  // ln_cs: per-workgroup column-sum partials of the two LayerNorm backward kernels (LN2, LN1)
  hipEvent_t ev_batch[kBwBuf] = {};  // wgrad batch of the last layer of parity q finished
  hipEvent_t ev_waited = nullptr;    // last batch event the chain has already waited for
  PendingBatch pend;
  unsigned bw_i = 0;
  float* slab = nullptr;      // split-K slabs of this chain's wgrad GEMMs (null = the handle's)
  bool inline_wgrad = false;  // run the wgrad batch on the chain's own stream instead of the wgrad stream
};

}  // namespace

// In-step kernel-class timing (fact_kprof): HIP events recorded on the launch stream around each instrumented
// launch site while the train step runs with all its stream overlap; read back per class after a sync.
enum KClass {
  KP_LN_FWD, KP_QKV, KP_ATTN_FWD, KP_OUTPROJ, KP_FFN1, KP_FFN2, KP_GELU_DGRAD, KP_DFFN1, KP_LN_BWD, KP_OUT_DGRAD,
  KP_ATTN_BWD, KP_DQKV, KP_WGRAD, KP_PARAM_GRADS, KP_ADAM, KP_N
};
const char* const kKClassName[KP_N] = {
    "ln_fwd", "qkv_gemm+heads", "attention_fwd", "out_proj+resid", "ffn1+gelu", "ffn2+resid", "gelu'_dgrad",
    "ffn1_dgrad", "ln_bwd_dx", "out_proj_dgrad+heads", "attention_bwd", "qkv_dgrad", "wgrad_group",
    "bias/ln_param_grads", "adam+shadows"};
struct KNote {  // one kernel launch seen by FACT_LAUNCH while the recorder is armed
  const void* fn;
  unsigned grid, block;
  size_t lds;
};
struct KProf {
  bool on = false;
  struct Rec { int cls; int launches; int sid; double flops; double bytes; hipEvent_t a, b; std::vector<KNote> notes; };
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  hipEvent_t get() {
    if (used == pool.size()) {
      hipEvent_t e = nullptr;
      (void)hipEventCreate(&e);
      pool.push_back(e);
    }
    return pool[used++];
  }
};

struct FactHandle {
  FactConfig cfg;
  KProf kp;
  int max_batch = 0;
  bool training = false;
  bool own_arenas = false;
  size_t arena_floats = 0;
  float *params = nullptr, *grads = nullptr, *adam_m = nullptr, *adam_v = nullptr;
  std::vector<FactParamDesc> table;
  Stack motion, audio, cross;
  DenseW head;
  Tensor head_b;
  int outp = 0;  // out_dim padded to 32
  char* shadow = nullptr;
  size_t shadow_bytes = 0;
  char* work = nullptr;
  size_t work_bytes = 0;
  // shared scratch
  bf16_t* xf16 = nullptr;    // final hidden states bf16 [Mc][d]
  float* pred = nullptr;     // [Mc][out_dim]
  bf16_t* dpred = nullptr;   // [Mc][outp]
  float* dx = nullptr;       // running residual gradient [Mc][d]
  bf16_t* dx16 = nullptr;
  float *dxm = nullptr, *dxa = nullptr;  // encoder gradients after the split
  bf16_t *dxm16 = nullptr, *dxa16 = nullptr;
  bf16_t *tA = nullptr, *tB = nullptr;  // transposed operands for the non-tr wgrad path
  CastDesc* cast_table = nullptr;       // device tables for the per-bucket weight-shadow refresh
  AdamBlock* adam_blocks = nullptr;     // device table of the fused Adam + shadow-refresh kernel
  int fuse_adam_cast = 1;
  std::vector<Bucket> buckets;          // gradient / optimizer buckets in backward-completion order
  AdamArgs adam;                        // hyper-parameters of the optimizer step in flight
  bool adam_pending = false;            // fact_adam_begin called: buckets are updated inside backward
  hipStream_t opt = nullptr;            // stream of the overlapped Adam + shadow refresh (HBM-bound)
  float* scalars = nullptr;             // [16] device scalars (loss, sumsq)
  bf16_t* ar_x16 = nullptr;             // AR: bf16 hidden rows of token 0 [B][d]
  float* ar_motion = nullptr;           // AR: extended motion track (B, n_m + steps, F_m)
  size_t ar_motion_floats = 0;
  int64_t step = 0;
  int wgrad_tr = 1;
  int wgrad_slab = 1;
  int wgrad_big = 1;  // whole-K grouped big-tile wgrad launch per layer (gemm_big.hip)
  int ln_split = 1;   // LayerNorm backward: row-wise dx kernel on the chain, parameter gradients on the wgrad stream
  int ln_cs = 0;      // (with ln_split) 1 / 2: the dx kernel leaves the column sums as per-workgroup partials (4 / 2 rows per
                      // wave), a small reduce on the wgrad stream replaces the column-sum pass over dh / x / dy.  Round 4:
                      // parity green, step unchanged (7.65 vs 7.66 ms: the dx kernel pays 22 -> 33 us on the dgrad chain for the
                      // 50 -> 3 x 15 us it takes off the wgrad stream) - off by default, DESIGN 6
  int bias_in_wgrad = 1;  // round 6: dense_1 / dense_2 / to_out bias gradients as operand column sums inside the grouped wgrad launch
                          // (TnProblem::csum) instead of a re-read of dpre / the residual gradients by col_tasks_kernel
  int wgrad_parts = 2;   // launches per layer of the grouped wgrad kernel (each ~190/parts workgroups wide)
  int wgrad_defer = 1;   // release a layer's wgrad batch behind the NEXT layer's GELU' dgrad (240 workgroups)
  int bwd_splitk = 0;    // in-kernel split-K for the N = 800 dgrad GEMMs: 1 = always, 2 = stacks of <= 4096 rows (encoders)
  int sr_rows = 1;       // supervised-rows shortcut of the last cross-modal layer in fact_forward_backward (SrBuf)
  SrBuf sr;
  float* skinny_acc = nullptr;  // zero-filled fp32 accumulator of the skinny-M GEMMs of that layer (caller's stream)
  float* skinny_acc2 = nullptr; // the same for GEMMs on the side stream (AR sampler: the audio encoder runs there)
  bool skinny_fwd = false;      // layer_forward hands its GEMMs the accumulator when M <= 512 (fact_infer_ar at small B)
  size_t skinny_floats = 0;
  float* slab = nullptr;  // split-K partial slabs of the wgrad GEMM running on the side stream
  // second stream: wgrad GEMMs run beside the dgrad chain, the audio encoder beside the motion encoder
  int use_side = 1;
  hipStream_t side = nullptr;
  std::vector<hipEvent_t> ev;
  size_t ev_i = 0;
  // Backward scratch that the wgrad stream reads is double-buffered (layer parity) so the dgrad chain
  // only ever waits for the wgrad GEMMs of TWO layers ago, never for the ones just enqueued.
  BwScratch bw[2];  // backward chains: 0 = cross-modal + audio stacks, 1 = motion stack (concurrent)
  // in-kernel split-K workspaces of the N = 800 GEMMs, one per stream that launches them (caller's / side / aux)
  float* sk_slab[3] = {nullptr, nullptr, nullptr};
  unsigned* sk_cnt[3] = {nullptr, nullptr, nullptr};
  hipStream_t aux = nullptr;  // stream of backward chain 1
  hipStream_t lite = nullptr; // column-sum kernels of the optimizer-only batches (beside the wgrad launches)
  int use_lite = 0;  // measured: no gain (round 2), off by default
  int use_aux = 1;   // third stream for the motion stack's backward chain (0: on the caller's stream - the data-parallel
                     // trainer sets it: with its communication stream and RCCL's own the process would have more
                     // streams than the 4 hardware queues HIP gives it, and streams that share a queue serialise)
  int grad_overwrite = 0;     // 1: a train step WRITES the gradient of every Dense kernel of a transformer layer (plain stores in the
                              // grouped wgrad launch) instead of accumulating into it, and the optimizer pass does not zero those
                              // ranges: 32 instead of 36 bytes per parameter and no read-modify-write in the wgrad epilogue.
                              // The host trainer sets it (one forward_backward per optimizer step); the C-ABI default keeps the
                              // accumulate semantics of fact_forward_backward.
  int skip = 0;               // TIMING-ONLY ablation mask (results are wrong): 1 wgrad 2 col sums 4 attn bwd 8 ln bwd 16 gelu' dgrad
                              // 32 ffn1 dgrad 64 qkv dgrad 128 out-proj dgrad 256 attn fwd 512 ln fwd
  int ln_fuse = 1;            // skinny-M forward GEMMs with a residual add: the LayerNorm that follows runs inside their
                              // epilogue pass (one launch less per sub-block; batch-1 AR sampler, supervised-rows layer)
  bool keep_pre = true;       // forward stores the dense_1 pre-activations (backward's GELU' reads them); inference entry
                              // points clear it for their call: 2 bytes x ff per token and layer less to write.  Backward
                              // never consumes activations of fact_forward / fact_infer_ar: fact_forward_backward always
                              // re-runs its own forward (there is no split forward / backward entry point)
  bool keep_pre_infer = false;  // what the inference entry points set keep_pre to (read FACT_KEEP_PRE once, at fact_create)
  int adam_hold = 1;          // in-backward optimizer: hold the head + cross buckets until the last is final
  int adam_in_wgrad = 0;      // (round 5; off by default - measured slower at fact_v5 / B = 16) with fact_adam_begin pending, no gradient callback and grad_overwrite on, the Adam
                              // update + shadow refresh of every transformer-layer Dense kernel runs in the epilogue of the
                              // grouped whole-K wgrad launch that produced its gradient (gemm.h TnGroup::adam): 28 instead of
                              // 4 + 32 bytes per parameter, and the optimizer pass shrinks to the ~0.7 M parameters that are
                              // not layer kernels.  The gradient arena ranges of those kernels are then not written.
  bool wg_adam = false;       // this fact_forward_backward call runs that way (decided per call: wg_adam_ok)
  // gradient-bucket-ready callback (data-parallel overlap of the RCCL all-reduce with backward)
  fact_grad_cb cb = nullptr;
  void* cb_user = nullptr;
  hipStream_t cb_stream = nullptr;
  int cb_bucket = 0;
};

namespace {

}  // namespace
// Launch notes of the recorder.  Both are PER THREAD: a launch is noted only by the thread that has a KScope of an armed
// handle open (a data-parallel communication thread, or a second model driven from another thread, never writes into
// this thread's buffer), and only while that scope is open - launches outside any scope are not collected, so the buffer
// is bounded by the launches of one scope (hard cap below).
thread_local bool g_fact_note_on = false;
static thread_local std::vector<KNote> g_fact_notes;
static constexpr size_t kMaxNotesPerScope = 256;
void fact_note_launch(const void* host_fn, dim3 grid, dim3 block, size_t lds_bytes) {
  if (g_fact_notes.size() < kMaxNotesPerScope)
    g_fact_notes.push_back(KNote{host_fn, grid.x * grid.y * grid.z, block.x * block.y * block.z, lds_bytes});
}
namespace {

struct KScope {
  FactHandle* h;
  hipStream_t s;
  size_t idx = (size_t)-1;
  bool outer_on = false;              // a scope opened inside another one (none today) hands the notes collected so far
  std::vector<KNote> outer_notes;     //   back to the outer scope when it closes
  KScope(FactHandle* h_, int cls, hipStream_t s_, double flops, double bytes = 0, int launches = 1) : h(h_), s(s_) {
    if (!h->kp.on) return;
    outer_on = g_fact_note_on;
    if (outer_on) outer_notes.swap(g_fact_notes);
    KProf::Rec r;
    r.cls = cls; r.launches = launches; r.flops = flops; r.bytes = bytes;
    r.sid = (s == h->side) ? 1 : (s == h->aux) ? 2 : (s == h->opt) ? 3 : (s == h->lite) ? 4 : 0;
    r.a = h->kp.get();
    r.b = h->kp.get();
    (void)hipEventRecord(r.a, s);
    idx = h->kp.recs.size();
    h->kp.recs.push_back(r);
    g_fact_notes.clear();
    g_fact_note_on = true;
  }
  ~KScope() {
    if (idx == (size_t)-1) return;
    (void)hipEventRecord(h->kp.recs[idx].b, s);
    h->kp.recs[idx].notes.swap(g_fact_notes);
    g_fact_notes.clear();
    g_fact_note_on = outer_on;
    if (outer_on) g_fact_notes.swap(outer_notes);
  }
};


// ---------------------------------------------------------------------------------------------
// parameter table (Keras trainable_variables order: cross_modal_layer, motion_transformer,
// motion_pos_embedding, motion_linear_embedding, audio_transformer, audio_pos_embedding,
// audio_linear_embedding — attribute assignment order in fact_model.py:43-70)
// ---------------------------------------------------------------------------------------------
struct TableBuilder {
  std::vector<FactParamDesc>* out;
  size_t off = 0;
  Tensor add(const std::string& name, int rows, int cols, int kind) {
    off = rups(off, 64);
    Tensor t;
    t.off = off;
    t.rows = rows;
    t.cols = cols;
    if (out) {
      FactParamDesc d;
      memset(&d, 0, sizeof(d));
      snprintf(d.name, sizeof(d.name), "%s", name.c_str());
      d.offset = off;
      d.rows = rows;
      d.cols = cols;
      d.kind = kind;
      out->push_back(d);
    }
    off += (size_t)rows * cols;
    return t;
  }
};

void build_stack_params(TableBuilder& tb, Stack& st, const std::string& prefix) {
  st.lp.resize(st.L);
  for (int l = 0; l < st.L; ++l) {
    LayerP& p = st.lp[l];
    const std::string b = prefix + "/layer_" + std::to_string(l);
    p.ln1_g = tb.add(b + "/attn_norm/gamma", 1, st.d, 2);
    p.ln1_b = tb.add(b + "/attn_norm/beta", 1, st.d, 3);
    p.wqkv.w = tb.add(b + "/attn/to_qkv/kernel", st.d, 3 * st.d, 0);
    p.wo.w = tb.add(b + "/attn/to_out/kernel", st.d, st.d, 0);
    p.bo = tb.add(b + "/attn/to_out/bias", 1, st.d, 1);
    p.ln2_g = tb.add(b + "/mlp_norm/gamma", 1, st.d, 2);
    p.ln2_b = tb.add(b + "/mlp_norm/beta", 1, st.d, 3);
    p.w1.w = tb.add(b + "/mlp/dense_1/kernel", st.d, st.ff, 0);
    p.b1 = tb.add(b + "/mlp/dense_1/bias", 1, st.ff, 1);
    p.w2.w = tb.add(b + "/mlp/dense_2/kernel", st.ff, st.d, 0);
    p.b2 = tb.add(b + "/mlp/dense_2/bias", 1, st.d, 1);
  }
}

void build_table(FactHandle* h, std::vector<FactParamDesc>* out, size_t* total) {
  TableBuilder tb;
  tb.out = out;
  build_stack_params(tb, h->cross, "cross_modal_layer/transformer");
  h->head.w = tb.add("cross_modal_layer/output/kernel", h->cross.d, h->cfg.out_dim, 0);
  h->head_b = tb.add("cross_modal_layer/output/bias", 1, h->cfg.out_dim, 1);
  build_stack_params(tb, h->motion, "motion_transformer");
  h->motion.pos = tb.add("motion_pos_embedding/position_embedding", h->motion.n, h->motion.d, 4);
  h->motion.emb.w = tb.add("motion_linear_embedding/kernel", h->motion.feat, h->motion.d, 0);
  h->motion.emb_b = tb.add("motion_linear_embedding/bias", 1, h->motion.d, 1);
  build_stack_params(tb, h->audio, "audio_transformer");
  h->audio.pos = tb.add("audio_pos_embedding/position_embedding", h->audio.n, h->audio.d, 4);
  h->audio.emb.w = tb.add("audio_linear_embedding/kernel", h->audio.feat, h->audio.d, 0);
  h->audio.emb_b = tb.add("audio_linear_embedding/bias", 1, h->audio.d, 1);
  *total = rups(tb.off, 64);
}

void init_stack_geo(Stack& st, const char* name, const FactStackCfg& c, int n, int feat) {
  st.name = name;
  st.n = n;
  st.feat = feat;
  st.featp = feat > 0 ? rup(feat, 32) : 0;
  st.d = c.hidden;
  st.H = c.heads;
  st.dh = c.heads > 0 ? c.hidden / c.heads : 0;
  st.dhp = rup(st.dh, 32);
  st.ff = c.ff;
  st.dp = rup(st.d, kPitch);
  st.qp = rup(3 * st.d, kPitch);
  st.fp = rup(st.ff, kPitch);
  st.L = c.layers;
  st.NP = rup(n, 128);
}

int validate(const FactConfig& c) {
  const FactStackCfg* s[3] = {&c.motion, &c.audio, &c.cross};
  for (int i = 0; i < 3; ++i) {
    if (s[i]->hidden <= 0 || s[i]->heads <= 0 || s[i]->layers < 0 || s[i]->ff <= 0)
      return fail(-1, "invalid stack hyper-parameters");
    if (s[i]->hidden % s[i]->heads) return fail(-1, "hidden_size must be divisible by num_attention_heads");
    const int dh = s[i]->hidden / s[i]->heads;
    if (dh != 32 && dh != 64 && dh != 80 && dh != 128)
      return fail(-3, "unsupported head dim " + std::to_string(dh) + " (supported: 32, 64, 80, 128)");
    if (s[i]->hidden % 8 || s[i]->ff % 8) return fail(-1, "hidden/intermediate size must be multiples of 8");
    if (s[i]->hidden > 2048) return fail(-1, "hidden_size > 2048 unsupported");
  }
  // base_models.py:184-189: ValueError when the two modal widths differ
  if (c.motion.hidden != c.audio.hidden)
    return fail(-2, "The modal_a hidden size (" + std::to_string(c.motion.hidden) +
                        ") should be the same with the modal_b hidden size (" +
                        std::to_string(c.audio.hidden) + ")");
  if (c.cross.hidden != c.motion.hidden) return fail(-2, "cross-modal hidden size must equal the modal hidden size");
  if (c.motion.seq_len <= 0 || c.audio.seq_len <= 0) return fail(-1, "sequence_length must be positive");
  if (c.motion.seq_len % 8 || c.audio.seq_len % 8) return fail(-1, "sequence lengths must be multiples of 8");
  if (c.motion.feature_dim <= 0 || c.audio.feature_dim <= 0) return fail(-1, "feature_dim must be positive");
  if (c.out_dim <= 0) return fail(-1, "out_dim must be positive");
  return 0;
}

void layout_dense(Bump& b, DenseW& w) {
  w.lds = rup(w.w.cols, kPitch);
  w.ldt = rup(w.w.rows, kPitch);
  w.s = b.take<bf16_t>((size_t)w.w.rows * w.lds);
  w.t = b.take<bf16_t>((size_t)w.w.cols * w.ldt);
}

void layout_shadow(FactHandle* h, Bump& b) {
  Stack* sts[3] = {&h->cross, &h->motion, &h->audio};
  for (Stack* st : sts) {
    for (LayerP& p : st->lp) {
      layout_dense(b, p.wqkv);
      layout_dense(b, p.wo);
      layout_dense(b, p.w1);
      layout_dense(b, p.w2);
    }
  }
  layout_dense(b, h->head);
  layout_dense(b, h->motion.emb);
  layout_dense(b, h->audio.emb);
}

void layout_stack_acts(FactHandle* h, Bump& b, Stack& st, int B) {
  const size_t M = (size_t)B * st.n;
  const size_t BH = (size_t)B * st.H;
  const int nsets = h->training ? st.L : (st.L ? 1 : 0);
  st.la.assign(st.L, LayerA());
  if (st.feat > 0) st.xin16 = b.take<bf16_t>(M * st.featp);
  st.x0 = b.take<float>(M * st.d);
  std::vector<LayerA> sets(nsets);
  for (int i = 0; i < nsets; ++i) {
    LayerA& a = sets[i];
    a.x_mid = b.take<float>(M * st.d);
    a.mean1 = b.take<float>(M);
    a.rstd1 = b.take<float>(M);
    a.mean2 = b.take<float>(M);
    a.rstd2 = b.take<float>(M);
    a.h1 = b.take<bf16_t>(M * st.dp);
    a.a = b.take<bf16_t>(M * st.dp);
    a.h2 = b.take<bf16_t>(M * st.dp);
    a.pre = b.take<bf16_t>(M * st.fp);
    a.g = b.take<bf16_t>(M * st.fp);
    for (int w = 0; w < 3; ++w) {
      a.row[w] = b.take<bf16_t>(BH * st.NP * st.dhp);
    }
    a.lse2 = b.take<float>(BH * st.NP);
  }
  float* pp[2] = {nullptr, nullptr};
  if (!h->training && st.L) {
    pp[0] = b.take<float>(M * st.d);
    pp[1] = b.take<float>(M * st.d);
  }
  for (int l = 0; l < st.L; ++l) {
    LayerA a = sets[h->training ? l : 0];
    a.x_in = (l == 0) ? st.x0 : st.la[l - 1].x_out;
    a.x_out = h->training ? b.take<float>(M * st.d) : pp[l & 1];
    st.la[l] = a;
  }
}

void layout_work(FactHandle* h, Bump& b) {
  const int B = h->max_batch;
  layout_stack_acts(h, b, h->motion, B);
  layout_stack_acts(h, b, h->audio, B);
  layout_stack_acts(h, b, h->cross, B);
  const int d = h->cross.d, dp = h->cross.dp;
  const size_t Mc = (size_t)B * h->cross.n;
  const size_t Mm = (size_t)B * h->motion.n, Ma = (size_t)B * h->audio.n;
  h->xf16 = b.take<bf16_t>(Mc * dp);
  h->pred = b.take<float>(Mc * h->cfg.out_dim);
  h->scalars = b.take<float>(16);
  h->ar_x16 = b.take<bf16_t>((size_t)B * d);
  {  // supervised-rows compact buffers, forward half (training: B*T <= B*32 rows; the AR sampler: B rows)
    SrBuf& r = h->sr;
    const size_t R = rups((size_t)B * (h->training ? 32 : 1), 64);
    const int fpc = h->cross.fp;
    r.rows_max = (int)R;
    r.a_c = b.take<bf16_t>(R * dp); r.h2_c = b.take<bf16_t>(R * dp); r.pre_c = b.take<bf16_t>(R * fpc);
    r.g_c = b.take<bf16_t>(R * fpc); r.xf16_c = b.take<bf16_t>(R * dp);
    r.x_in_c = b.take<float>(R * d); r.x_mid_c = b.take<float>(R * d); r.x_out_c = b.take<float>(R * d);
    r.mean2_c = b.take<float>(R); r.rstd2_c = b.take<float>(R);
  }
  if (h->training) {
    int ffmax = h->cross.fp;
    if (h->motion.fp > ffmax) ffmax = h->motion.fp;
    if (h->audio.fp > ffmax) ffmax = h->audio.fp;
    size_t rowmax = 0, lsemax = 0;
    Stack* sts[3] = {&h->cross, &h->motion, &h->audio};
    for (Stack* st : sts) {
      const size_t BH = (size_t)B * st->H;
      if (BH * st->NP * st->dhp > rowmax) rowmax = BH * st->NP * st->dhp;
      if (BH * st->NP > lsemax) lsemax = BH * st->NP;
    }
    {  // supervised-rows compact buffers, backward half: up to 32 target rows per sequence
      SrBuf& r = h->sr;
      const size_t R = (size_t)r.rows_max;
      const int fpc = h->cross.fp;
      r.dx16_c = b.take<bf16_t>(R * dp);
      r.dpre_c = b.take<bf16_t>(R * fpc); r.dh2_c = b.take<bf16_t>(R * dp); r.xmid16_c = b.take<bf16_t>(R * dp);
      r.dpred_c = b.take<bf16_t>(R * h->outp);
      r.pred_c = b.take<float>(R * h->cfg.out_dim); r.dx_c = b.take<float>(R * d);
    }
    h->dpred = b.take<bf16_t>(Mc * h->outp);
    h->dx = b.take<float>(Mc * d);
    h->dx16 = b.take<bf16_t>(Mc * dp);
    h->dxm = b.take<float>(Mm * d);
    h->dxm16 = b.take<bf16_t>(Mm * dp);
    h->dxa = b.take<float>(Ma * d);
    h->dxa16 = b.take<bf16_t>(Ma * dp);
    // chain 0 (cross-modal / audio stacks) is sized for the cross-modal token count, chain 1 for the motion stack
    for (int c = 0; c < 2; ++c) {
      BwScratch& sc = h->bw[c];
      const Stack& big = c ? h->motion : h->cross;
      const size_t Mx = c ? Mm : Mc;
      size_t rows = (size_t)B * big.H * big.NP * big.dhp, lse = (size_t)B * big.H * big.NP;
      if (!c) {
        rows = rowmax;
        lse = lsemax;
      }
      sc.dh = b.take<bf16_t>(Mx * dp);
      for (int q = 0; q < kBwBuf; ++q) {
        sc.dpre_pp[q] = b.take<bf16_t>(Mx * ffmax);
        sc.dqkv_pp[q] = b.take<bf16_t>(Mx * h->cross.qp);
        sc.xmid_pp[q] = b.take<bf16_t>(Mx * dp);
        sc.xb_pp[q] = b.take<bf16_t>(Mx * dp);
        sc.dh2_pp[q] = b.take<bf16_t>(Mx * dp);
        sc.dh1_pp[q] = b.take<bf16_t>(Mx * dp);
        // (lnpart_pp: the ln_cs partial buffers are allocated when that debug option is first switched on - alloc_ln_cs)
      }
      sc.dorow = b.take<bf16_t>(rows);
      sc.dsum = b.take<float>(lse);
      sc.ln_ws = b.take<float>(ln_bwd_ws_floats((int)Mx, d));
      if (c) {  // chain 1 queues its weight gradients behind its own dgrad chain: own slabs
        sc.slab = b.take<float>((size_t)6 * rups((size_t)d * (ffmax > 3 * d ? ffmax : 3 * d), 4));
        sc.inline_wgrad = true;
      }
    }
    h->slab = b.take<float>((size_t)6 * rups((size_t)d * (ffmax > 3 * d ? ffmax : 3 * d), 4));
    const size_t wide = (size_t)(ffmax > 3 * d ? ffmax : 3 * d);
    h->tA = b.take<bf16_t>(wide * rups(Mc, 8));
    h->tB = b.take<bf16_t>(wide * rups(Mc, 8));
  }
}

float* P(FactHandle* h, const Tensor& t) { return h->params + t.off; }
float* G(FactHandle* h, const Tensor& t) { return h->grads + t.off; }

size_t tensor_end(const Tensor& t) { return rups(t.off + t.numel(), 64); }

// Gradient / optimizer buckets: contiguous arena ranges in the order their gradients become final
// during backward - head, cross layers L-1..0, audio stack, motion stack.  Each bucket carries the
// device table of its Dense kernels so its bf16 shadows can be refreshed with one launch.
int build_buckets(FactHandle* h) {
  std::vector<CastDesc> t;
  h->buckets.clear();
  auto open = [&](size_t beg, size_t end) {
    Bucket b;
    b.off = beg;
    b.cnt = end - beg;
    b.desc_first = (int)t.size();
    b.desc_n = 0;
    b.tiles = 0;
    h->buckets.push_back(b);
  };
  auto add = [&](DenseW& w) {
    Bucket& b = h->buckets.back();
    CastDesc d;
    d.src = P(h, w.w);
    d.s = w.s;
    d.t = w.t;
    d.R = w.w.rows;
    d.C = w.w.cols;
    d.lds = w.lds;
    d.ldt = w.ldt;
    d.tiles_x = (d.C + 63) / 64;
    d.tile_begin = b.tiles;
    b.tiles += d.tiles_x * ((d.R + 63) / 64);
    b.desc_n += 1;
    t.push_back(d);
  };
  auto add_layer = [&](LayerP& p) {
    add(p.wqkv);
    add(p.wo);
    add(p.w1);
    add(p.w2);
  };
  Stack &cr = h->cross, &mo = h->motion, &au = h->audio;
  open(h->head.w.off, tensor_end(h->head_b));
  add(h->head);
  for (int l = cr.L - 1; l >= 0; --l) {
    open(cr.lp[l].ln1_g.off, tensor_end(cr.lp[l].b2));
    add_layer(cr.lp[l]);
  }
  open(au.L ? au.lp[0].ln1_g.off : au.pos.off, tensor_end(au.emb_b));
  for (LayerP& p : au.lp) add_layer(p);
  add(au.emb);
  open(mo.L ? mo.lp[0].ln1_g.off : mo.pos.off, tensor_end(mo.emb_b));
  for (LayerP& p : mo.lp) add_layer(p);
  add(mo.emb);
  HIPCHK(hipMalloc((void**)&h->cast_table, t.size() * sizeof(CastDesc)));
  HIPCHK(hipMemcpy(h->cast_table, t.data(), t.size() * sizeof(CastDesc), hipMemcpyHostToDevice));
  // Fused Adam + shadow refresh: walk every bucket's arena range; Dense kernels become 64x64 tile
  // blocks, everything between them (biases, LayerNorm, position tables, alignment padding) flat blocks.
  std::vector<AdamBlock> ab, lite;
  // Dense kernels of transformer layers: their gradient is written (not accumulated) in overwrite mode
  std::vector<size_t> layer_w;
  for (Stack* st : {&h->cross, &h->audio, &h->motion})
    for (LayerP& p : st->lp)
      for (DenseW* w : {&p.wqkv, &p.wo, &p.w1, &p.w2}) layer_w.push_back(w->w.off);
  for (Bucket& b : h->buckets) {
    b.blk_first = (int)ab.size();
    b.lite_first = (int)lite.size();
    b.lite_floats = 0;
    std::vector<CastDesc> ds(t.begin() + b.desc_first, t.begin() + b.desc_first + b.desc_n);
    std::sort(ds.begin(), ds.end(), [](const CastDesc& x, const CastDesc& y) { return x.src < y.src; });
    size_t cur = b.off;
    auto flat = [&](size_t beg, size_t end) {
      for (size_t o = beg; o < end; o += 4096) {
        AdamBlock k;
        memset(&k, 0, sizeof(k));
        k.off = o;
        k.C = (int)std::min<size_t>(4096, end - o);
        ab.push_back(k);
        lite.push_back(k);
        b.lite_floats += (size_t)k.C;
      }
    };
    for (const CastDesc& d : ds) {
      const size_t toff = (size_t)(d.src - h->params);
      if (toff > cur) flat(cur, toff);
      const int tw = adam_tile_width();
      for (int r0 = 0; r0 < d.R; r0 += 64)
        for (int c0 = 0; c0 < d.C; c0 += tw) {
          AdamBlock k;
          memset(&k, 0, sizeof(k));
          k.off = toff; k.s = d.s; k.t = d.t; k.R = d.R; k.C = d.C; k.lds = d.lds; k.ldt = d.ldt;
          k.r0 = r0; k.c0 = c0;
          k.pad[0] = std::find(layer_w.begin(), layer_w.end(), toff) != layer_w.end() ? 1 : 0;
          ab.push_back(k);
          if (!k.pad[0]) {
            lite.push_back(k);
            b.lite_floats += (size_t)std::min(64, d.R - r0) * (size_t)std::min(tw, d.C - c0);
          }
        }
      // the tail of a Dense tensor whose size is not a multiple of 4 floats is covered by its tile
      // blocks; flat segments resume at the next 64-float boundary (tensor offsets are 64-aligned)
      cur = rups(toff + (size_t)d.R * d.C, 64);
    }
    if (b.off + b.cnt > cur) flat(cur, b.off + b.cnt);
    b.blk_n = (int)ab.size() - b.blk_first;
    b.lite_n = (int)lite.size() - b.lite_first;
  }
  // one device table: the full block list, then the lite list
  const size_t n_full = ab.size();
  for (Bucket& b : h->buckets) b.lite_first += (int)n_full;
  ab.insert(ab.end(), lite.begin(), lite.end());
  HIPCHK(hipMalloc((void**)&h->adam_blocks, ab.size() * sizeof(AdamBlock)));
  HIPCHK(hipMemcpy(h->adam_blocks, ab.data(), ab.size() * sizeof(AdamBlock), hipMemcpyHostToDevice));
  return 0;
}

// ln_cs (debug option, off by default): per-workgroup column-sum partials of the two LayerNorm backward kernels, three
// layer parities per backward chain.  ~6 x 7 MB per chain at fact_v5 / B = 16 and ~6 x 26 MB at the scaled configuration:
// allocated the first time the option is switched on instead of being carved out of every training handle's work arena.
int alloc_ln_cs(FactHandle* h) {
  if (!h->training) return 0;
  const int B = h->max_batch, d = h->cross.d;
  for (int c = 0; c < 2; ++c) {
    const int Mx = B * (c ? h->motion.n : h->cross.n);
    for (int q = 0; q < kBwBuf; ++q)
      for (int i = 0; i < 2; ++i)
        if (!h->bw[c].lnpart_pp[q][i])
          HIPCHK(hipMalloc((void**)&h->bw[c].lnpart_pp[q][i], ln_cs_part_floats(Mx, d) * sizeof(float)));
  }
  return 0;
}

int refresh_bucket(FactHandle* h, int b, hipStream_t s) {
  const Bucket& k = h->buckets[b];
  return launch_multi_cast_transpose(h->cast_table + k.desc_first, k.desc_n, k.tiles, s);
}

int refresh_all(FactHandle* h, hipStream_t s) {
  for (int b = 0; b < (int)h->buckets.size(); ++b) CHK(refresh_bucket(h, b, s));
  return 0;
}

// Keras-Adam on one bucket (params, m, v updated; grads zeroed) + its shadow refresh
int adam_bucket(FactHandle* h, int b, hipStream_t s, const bf16_t* g16 = nullptr) {
  const Bucket& k = h->buckets[b];
  const AdamArgs& a = h->adam;
  if (h->wg_adam) {  // the layer kernels of this bucket were updated by their wgrad launches: what is left of it
    if (g16 || !h->fuse_adam_cast) return fail(-1, "optimizer fused into the wgrad launches: fp32 buckets, fused Adam kernel");
    if (!k.lite_n) return 0;
    KScope ks(h, KP_ADAM, s, 0, (double)k.lite_floats * 36.0);
    return launch_adam_fused(h->adam_blocks + k.lite_first, k.lite_n, h->params, h->adam_m, h->adam_v, h->grads,
                             a.lr_t, a.b1, a.b2, a.eps, a.gscale, s, nullptr, h->grad_overwrite);
  }
  KScope ks(h, KP_ADAM, s, 0, (double)k.cnt * (g16 ? 34.0 : 36.0));
  if (h->fuse_adam_cast)
    return launch_adam_fused(h->adam_blocks + k.blk_first, k.blk_n, h->params, h->adam_m, h->adam_v, h->grads,
                             a.lr_t, a.b1, a.b2, a.eps, a.gscale, s, g16, h->grad_overwrite);
  if (g16) return fail(-1, "bf16 gradient source needs the fused Adam + shadow kernel");
  CHK(launch_adam(h->params + k.off, h->adam_m + k.off, h->adam_v + k.off, h->grads + k.off, k.cnt, a.lr_t,
                  a.b1, a.b2, a.eps, a.gscale, s));
  return refresh_bucket(h, b, s);
}

int g_op_ln_ws = 0;            // bench knob (fact_debug_ln_bwd)
int g_op_tn_parts = 1;         // bench knob (fact_debug_gemm_tn_cfg): launches per group of fact_op_gemm_tn_group
int g_force_generic_gemm = 0;  // test knob (fact_debug_force_generic_gemm)

GemmParams gp(const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K) {
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.M = M; p.N = N; p.K = K; p.splitk = 1;
  p.force_generic = g_force_generic_gemm;
  p.ep.alpha = 1.0f;
  return p;
}

// hand the split-K workspace of stream `s` to a GEMM (the dispatcher decides whether to use it)
void with_ws(FactHandle* h, GemmParams& g, hipStream_t s) {
  const int i = (s == h->side) ? 1 : (s == h->aux) ? 2 : 0;
  g.sk_slab = h->sk_slab[i];
  g.sk_cnt = h->sk_cnt[i];
}

// hand the zero-filled skinny-M accumulator to a GEMM of the supervised-rows layer (caller's stream only)
void with_skinny(FactHandle* h, GemmParams& g) {
  g.skinny_acc = h->skinny_acc;
  g.skinny_floats = h->skinny_floats;
}
// ... and to a GEMM of an ordinary forward layer on stream `s` while the AR sampler runs at small batch: with
// M = B * n <= 512 rows a GEMM has 3-57 tiles of 128x128 and one workgroup walks all of K (FFN2 at M = 360: 59 us);
// cut along K over ~256 workgroups it is a few microseconds plus the epilogue pass.
void with_skinny_fwd(FactHandle* h, GemmParams& g, hipStream_t s) {
  if (!h->skinny_fwd || g.M > 512 || s == h->aux) return;
  g.skinny_acc = (s == h->side) ? h->skinny_acc2 : h->skinny_acc;
  g.skinny_floats = h->skinny_floats;
}

// Ask the residual GEMM `g` (already carrying its skinny accumulator) to run the LayerNorm that follows in its epilogue
// pass; false = it does not take the skinny path and the caller launches the LayerNorm itself.
bool fuse_ln(FactHandle* h, GemmParams& g, const float* gamma, const float* beta, bf16_t* out, int ld, float* mean,
             float* rstd) {
  if (!h->ln_fuse || g.N > 1024 || (g.N & 3) || !gemm_nt_takes_skinny(EPI_F32_BIAS_RESID, g)) return false;
  g.ep.ln_g = gamma; g.ep.ln_b = beta; g.ep.ln_h = out; g.ep.ln_ldh = ld; g.ep.ln_mean = mean; g.ep.ln_rstd = rstd;
  g.ep.ln_eps = h->cfg.ln_eps;
  return true;
}

// dW[Mo][No] += A^T B ; A [K][Mo(lda)], B [K][No(ldb)]
int wgrad(FactHandle* h, const bf16_t* A, int lda, int Mo, const bf16_t* B, int ldb, int No, int K,
          float* out, int ldo, hipStream_t s, float* slab = nullptr) {
  if (!slab) slab = h->slab;
  const int tiles = ((Mo + 127) / 128) * ((No + 127) / 128);
  const int ktiles = (K + 63) / 64;
  // ~2 blocks per CU: measured optimum on MI355X (tools/attic/gemm_bench.py: 168 tiles -> 3, 49 -> 5..8)
  int splitk = 560 / tiles;
  if (splitk > 6) splitk = 6;
  if (splitk > ktiles / 4) splitk = ktiles / 4;
  if (splitk < 1) splitk = 1;
  if (h->wgrad_tr) {
    GemmParams p = gp(A, lda, B, ldb, Mo, No, K);
    p.splitk = splitk;
    if (h->wgrad_slab && ldo == No) {
      // split-K partials as plain float4 stores into per-split slabs + one streaming reduce
      // (fp32 atomics to the fabric cost more than the whole K loop at these sizes)
      const size_t stride = rups((size_t)Mo * No, 4);
      p.ep.out0 = slab;
      p.ep.ldo0 = No;
      p.ep.slab_stride = stride;
      CHK(launch_gemm_tn(EPI_F32_SLAB, p, s));
      return launch_slab_reduce(slab, stride, splitk, out, (size_t)Mo * No, s);
    }
    p.ep.out0 = out;
    p.ep.ldo0 = ldo;
    return launch_gemm_tn(EPI_ATOMIC_F32, p, s);
  }
  const int ldk = rup(K, 8);
  CHK(launch_transpose_bf16(A, lda, K, Mo, h->tA, ldk, s));
  CHK(launch_transpose_bf16(B, ldb, K, No, h->tB, ldk, s));
  GemmParams p = gp(h->tA, ldk, h->tB, ldk, Mo, No, ldk);
  p.splitk = splitk;
  p.ep.out0 = out;
  p.ep.ldo0 = ldo;
  return launch_gemm_nt(EPI_ATOMIC_F32, p, s);
}

// Per-GEMM fallback for a layer's Dense kernel: these kernels accumulate, so in overwrite mode (grad_overwrite: the
// optimizer no longer zeroes the range) the target is cleared first.
int wgrad_layer_tensor(FactHandle* h, const bf16_t* A, int lda, int Mo, const bf16_t* B, int ldb, int No, int K,
                       float* out, int ldo, hipStream_t s, float* slab = nullptr) {
  if (h->grad_overwrite) HIPCHK(hipMemsetAsync(out, 0, (size_t)Mo * No * sizeof(float), s));
  return wgrad(h, A, lda, Mo, B, ldb, No, K, out, ldo, s, slab);
}

// Shape test of the grouped whole-K wgrad launch for a stack at M tokens (wgrad_layer_group / the supervised-rows batch).
bool wgrad_group_ok(const FactHandle* h, const Stack& st, int M) {
  return h->wgrad_big && h->wgrad_tr && !(M & 31) && M >= 512 && !(st.d & 3) && !(st.ff & 3) && st.d >= 160;
}

// Do the grouped wgrad launches of this stack also produce the three Dense bias gradients (TnProblem::csum)?
bool bias_in_wgrad_ok(const FactHandle* h, const Stack& st, int M) {
  return h->bias_in_wgrad && wgrad_group_ok(h, st, M) && big_tn_group_has_colsum() && !(h->skip & 1);
}

// May this fact_forward_backward call run the optimizer inside the wgrad launches (FactHandle::adam_in_wgrad)?  Needs the
// engine-owned fused step (fact_adam_begin, no gradient callback, no clipping), overwrite semantics for the layer
// gradients, and EVERY layer of every stack on the grouped launch (16-wide units on both sides of each Dense kernel).
bool wg_adam_ok(const FactHandle* h, int B) {
  if (!h->adam_in_wgrad || !h->adam_pending || h->cb || !h->grad_overwrite || !h->fuse_adam_cast || !h->ln_split ||
      (h->skip & 1) || h->adam.gscale != 1.0f)
    return false;
  for (const Stack* st : {&h->cross, &h->motion, &h->audio}) {
    if (!st->L) continue;
    if (!wgrad_group_ok(h, *st, B * st->n) || (st->d & 15) || (st->ff & 15)) return false;
  }
  return true;
}

// Hand a problem of a grouped wgrad launch the optimizer state of its Dense kernel (TnGroup::adam).
void arm_adam(FactHandle* h, TnGroup& g, int i, const DenseW& w) {
  TnProblem& q = g.p[i];
  q.p = h->params + w.w.off; q.m1 = h->adam_m + w.w.off; q.v = h->adam_v + w.w.off;
  q.sd = w.s; q.ldsd = w.lds; q.st = w.t; q.ldst = w.ldt;
  g.adam = 1; g.lr_t = h->adam.lr_t; g.b1 = h->adam.b1; g.b2 = h->adam.b2; g.eps = h->adam.eps;
}

// The four weight gradients of one transformer layer as ONE grouped whole-K launch (gemm_big.hip): 160x256
// tiles, the d-wide operand on the 160-tiled side (d = 800 -> 5 tiles exactly); dW2 = g^T dY is computed as
// (dY^T g) and stored transposed.  Returns 1 when the shape is not eligible (caller takes the per-GEMM path).
int wgrad_layer_group(FactHandle* h, const Stack& st, const LayerP& p, const LayerA& a, const bf16_t* xin16,
                      const bf16_t* dpre, const bf16_t* xmid16, const bf16_t* dqkv, int M, hipStream_t s) {
  const int d = st.d, ff = st.ff;
  if (!wgrad_group_ok(h, st, M)) return 1;
  TnGroup g;
  memset(&g, 0, sizeof(g));
  g.n = 4;
  g.K = M;
  auto set = [&](int i, const bf16_t* A, int lda, int Mo, const bf16_t* B, int ldb, int No, float* out, int ldo,
                 int trans) {
    TnProblem& q = g.p[i];
    q.A = A; q.lda = lda; q.M = Mo; q.B = B; q.ldb = ldb; q.N = No; q.out = out; q.ldo = ldo; q.trans_out = trans;
  };
  set(0, xin16, st.dp, d, a.g, st.fp, ff, G(h, p.w2.w), d, 1);          // dW2[ff][d]  = g^T dY
  set(1, a.h2, st.dp, d, dpre, st.fp, ff, G(h, p.w1.w), ff, 0);         // dW1[d][ff]  = LN2(x)^T dpre
  set(2, a.a, st.dp, d, xmid16, st.dp, d, G(h, p.wo.w), d, 0);          // dWo[d][d]   = attn^T dx_mid
  set(3, a.h1, st.dp, d, dqkv, st.qp, 3 * d, G(h, p.wqkv.w), 3 * d, 0); // dWqkv[d][3d] = LN1(x)^T dqkv
  g.overwrite = h->grad_overwrite;
  if (bias_in_wgrad_ok(h, st, M)) {  // column sums of operands the launch streams anyway (atomics: the bias ranges are zeroed)
    g.p[0].csum = G(h, p.b2);   // sum_rows dL/dx_out   (A operand of dW2)
    g.p[1].csum = G(h, p.b1);   // sum_rows dpre        (B operand of dW1)
    g.p[2].csum = G(h, p.bo);   // sum_rows dL/dx_mid   (B operand of dWo)
  }
  if (h->wg_adam) {
    arm_adam(h, g, 0, p.w2); arm_adam(h, g, 1, p.w1); arm_adam(h, g, 2, p.wo); arm_adam(h, g, 3, p.wqkv);
  }
  CHK(launch_big_tn_group(g, s, h->wgrad_parts));
  return 0;
}

void heads_ep(EpiParams& ep, const Stack& st, bf16_t* const* row, int nwhich) {
  for (int w = 0; w < 3; ++w) ep.hrow[w] = (w < nwhich) ? row[w] : nullptr;
  ep.n_tok = st.n;
  ep.n_pad = st.NP;
  ep.heads = st.H;
  ep.dh = st.dh;
  ep.dhp = st.dhp;
  ep.hid = st.d;
}

AttnParams attn_params(const Stack& st, const LayerA& a, int B) {
  AttnParams ap;
  memset(&ap, 0, sizeof(ap));
  ap.qrow = a.row[0]; ap.krow = a.row[1]; ap.vrow = a.row[2];
  ap.out = a.a; ap.o = a.a; ap.lse2 = a.lse2;
  ap.B = B; ap.H = st.H; ap.n = st.n; ap.NP = st.NP; ap.hid = st.d; ap.dh = st.dh;
  ap.ldo = st.dp; ap.ldq = st.qp;
  ap.scale = 1.0f / sqrtf((float)st.d);  // dim**-0.5 with dim = hidden_size (base_models.py:66,104)
  return ap;
}

// `to` waits for everything enqueued so far on `from` (no host sync). Returns the event used.
// NOTE: the caller's stream may legitimately be the null (legacy default) stream, so "no waiter" is
// a separate entry point, never a null `to`.
hipEvent_t stream_mark(FactHandle* h, hipStream_t from) {
  hipEvent_t e = h->ev[h->ev_i++ % h->ev.size()];
  (void)hipEventRecord(e, from);
  return e;
}
hipEvent_t stream_after(FactHandle* h, hipStream_t from, hipStream_t to) {
  hipEvent_t e = stream_mark(h, from);
  (void)hipStreamWaitEvent(to, e, 0);
  return e;
}
hipStream_t side_of(FactHandle* h, hipStream_t s) { return (h->use_side && h->side) ? h->side : s; }

int layer_forward(FactHandle* h, Stack& st, int l, int B, hipStream_t s) {
  const int M = B * st.n, d = st.d, dp = st.dp, fp = st.fp;
  LayerP& p = st.lp[l];
  LayerA& a = st.la[l];
  const double Md = (double)M, fl_attn = 4.0 * (double)B * st.H * (double)st.n * st.n * st.dh;
  bool ln2_fused = false;
  if (a.h1_ready) {
    a.h1_ready = false;  // LayerNorm 1 ran inside the previous layer's FFN2 epilogue pass
  } else {
    KScope k(h, KP_LN_FWD, s, 0, Md * d * 6.0);
    CHK(launch_ln_fwd(a.x_in, P(h, p.ln1_g), P(h, p.ln1_b), a.h1, dp, a.mean1, a.rstd1, M, d, h->cfg.ln_eps, s));
  }
  {
    KScope k(h, KP_QKV, s, 2.0 * Md * 3 * d * d);
    GemmParams g = gp(a.h1, dp, p.wqkv.t, p.wqkv.ldt, M, 3 * d, d);
    heads_ep(g.ep, st, a.row, 3);
    with_ws(h, g, s);
    with_skinny_fwd(h, g, s);
    CHK(launch_gemm_nt(EPI_HEADS, g, s));
  }
  {
    KScope k(h, KP_ATTN_FWD, s, fl_attn);
    CHK(launch_attn_fwd(attn_params(st, a, B), s));
  }
  {
    KScope k(h, KP_OUTPROJ, s, 2.0 * Md * d * d);
    GemmParams g = gp(a.a, dp, p.wo.t, p.wo.ldt, M, d, d);
    g.ep.out0 = a.x_mid; g.ep.ldo0 = d; g.ep.bias = P(h, p.bo); g.ep.resid = a.x_in; g.ep.ldr = d;
    with_ws(h, g, s);
    with_skinny_fwd(h, g, s);
    ln2_fused = fuse_ln(h, g, P(h, p.ln2_g), P(h, p.ln2_b), a.h2, dp, a.mean2, a.rstd2);
    CHK(launch_gemm_nt(EPI_F32_BIAS_RESID, g, s));
  }
  if (!ln2_fused) {
    KScope k(h, KP_LN_FWD, s, 0, Md * d * 6.0);
    CHK(launch_ln_fwd(a.x_mid, P(h, p.ln2_g), P(h, p.ln2_b), a.h2, dp, a.mean2, a.rstd2, M, d, h->cfg.ln_eps, s));
  }
  {
    KScope k(h, KP_FFN1, s, 2.0 * Md * st.ff * d);
    GemmParams g = gp(a.h2, dp, p.w1.t, p.w1.ldt, M, st.ff, d);
    g.ep.out0 = h->keep_pre ? a.pre : nullptr; g.ep.ldo0 = fp; g.ep.out1 = a.g; g.ep.ldo1 = fp; g.ep.bias = P(h, p.b1);
    with_ws(h, g, s);
    with_skinny_fwd(h, g, s);
    CHK(launch_gemm_nt(EPI_BIAS_GELU, g, s));
  }
  {
    KScope k(h, KP_FFN2, s, 2.0 * Md * st.ff * d);
    GemmParams g = gp(a.g, fp, p.w2.t, p.w2.ldt, M, d, st.ff);
    g.ep.out0 = a.x_out; g.ep.ldo0 = d; g.ep.bias = P(h, p.b2); g.ep.resid = a.x_mid; g.ep.ldr = d;
    with_ws(h, g, s);
    with_skinny_fwd(h, g, s);
    if (l + 1 < st.L && st.la[l + 1].x_in == a.x_out) {  // LayerNorm 1 of the next layer of this stack
      LayerP& pn = st.lp[l + 1];
      LayerA& an = st.la[l + 1];
      an.h1_ready = fuse_ln(h, g, P(h, pn.ln1_g), P(h, pn.ln1_b), an.h1, dp, an.mean1, an.rstd1);
    }
    CHK(launch_gemm_nt(EPI_F32_BIAS_RESID, g, s));
  }
  return 0;
}

// Forward of the LAST cross-modal layer with the supervised-rows shortcut (SrBuf): LayerNorm 1 and the QKV
// projection on all rows (every key / value is needed), attention for the first T queries of each sequence, then
// to_out, LayerNorm 2, the MLP on the compact B*T rows.  The layer output exists only as sr.x_out_c.
int layer_forward_sr(FactHandle* h, Stack& st, int l, int B, int T, hipStream_t s) {
  const int M = B * st.n, Mr = B * T, d = st.d, dp = st.dp, fp = st.fp;
  LayerP& p = st.lp[l];
  LayerA& a = st.la[l];
  SrBuf& r = h->sr;
  const double Md = (double)M, Mrd = (double)Mr;
  bool ln2_fused = false;
  if (a.h1_ready) {
    a.h1_ready = false;
  } else {
    KScope k(h, KP_LN_FWD, s, 0, Md * d * 6.0);
    CHK(launch_ln_fwd(a.x_in, P(h, p.ln1_g), P(h, p.ln1_b), a.h1, dp, a.mean1, a.rstd1, M, d, h->cfg.ln_eps, s));
  }
  {
    KScope k(h, KP_QKV, s, 2.0 * Md * 3 * d * d);
    GemmParams g = gp(a.h1, dp, p.wqkv.t, p.wqkv.ldt, M, 3 * d, d);
    heads_ep(g.ep, st, a.row, 3);
    with_ws(h, g, s);
    CHK(launch_gemm_nt(EPI_HEADS, g, s));
  }
  {
    KScope k(h, KP_ATTN_FWD, s, 4.0 * (double)B * st.H * (double)T * st.n * st.dh);
    AttnParams ap = attn_params(st, a, B);
    ap.nq = T;
    CHK(launch_attn_fwd(ap, s));
  }
  CHK(launch_gather_rows(a.x_in, a.a, B, st.n, T, d, dp, r.x_in_c, r.a_c, s));
  {
    KScope k(h, KP_OUTPROJ, s, 2.0 * Mrd * d * d);
    GemmParams g = gp(r.a_c, dp, p.wo.t, p.wo.ldt, Mr, d, d);
    with_skinny(h, g);
    g.ep.out0 = r.x_mid_c; g.ep.ldo0 = d; g.ep.bias = P(h, p.bo); g.ep.resid = r.x_in_c; g.ep.ldr = d;
    with_ws(h, g, s);
    ln2_fused = fuse_ln(h, g, P(h, p.ln2_g), P(h, p.ln2_b), r.h2_c, dp, r.mean2_c, r.rstd2_c);
    CHK(launch_gemm_nt(EPI_F32_BIAS_RESID, g, s));
  }
  if (!ln2_fused) {
    KScope k(h, KP_LN_FWD, s, 0, Mrd * d * 6.0);
    CHK(launch_ln_fwd(r.x_mid_c, P(h, p.ln2_g), P(h, p.ln2_b), r.h2_c, dp, r.mean2_c, r.rstd2_c, Mr, d, h->cfg.ln_eps, s));
  }
  {
    KScope k(h, KP_FFN1, s, 2.0 * Mrd * st.ff * d);
    GemmParams g = gp(r.h2_c, dp, p.w1.t, p.w1.ldt, Mr, st.ff, d);
    with_skinny(h, g);
    g.ep.out0 = h->keep_pre ? r.pre_c : nullptr; g.ep.ldo0 = fp; g.ep.out1 = r.g_c; g.ep.ldo1 = fp; g.ep.bias = P(h, p.b1);
    with_ws(h, g, s);
    CHK(launch_gemm_nt(EPI_BIAS_GELU, g, s));
  }
  {
    KScope k(h, KP_FFN2, s, 2.0 * Mrd * st.ff * d);
    GemmParams g = gp(r.g_c, fp, p.w2.t, p.w2.ldt, Mr, d, st.ff);
    with_skinny(h, g);
    g.ep.out0 = r.x_out_c; g.ep.ldo0 = d; g.ep.bias = P(h, p.b2); g.ep.resid = r.x_mid_c; g.ep.ldr = d;
    with_ws(h, g, s);
    CHK(launch_gemm_nt(EPI_F32_BIAS_RESID, g, s));
  }
  return 0;
}

// Launch the pending optimizer-only batch of a backward chain on its wgrad stream: the four weight gradients
// (grouped whole-K launch in `wgrad_parts` pieces), the dense_1 bias and, with the split LayerNorm backward,
// the LayerNorm / output-bias gradients of both sub-blocks - all behind ONE event recorded on the chain.
int flush_batch(FactHandle* h, BwScratch& sc) {
  PendingBatch& b = sc.pend;
  if (!b.valid) return 0;
  b.valid = false;
  Stack& st = *b.st;
  LayerP& p = st.lp[b.l];
  LayerA& a = st.la[b.l];
  const int M = b.M, d = st.d, ff = st.ff, dp = st.dp, fp = st.fp, qp = st.qp;
  hipStream_t s = b.s, w = b.w;
  const bool two = (w != s);
  hipEvent_t rel = two ? stream_after(h, s, w) : nullptr;
  if (b.sr_rows > 0) {
    // supervised-rows layer: dW2, dW1, dWo contract over the compact (zero-padded) rows, dWqkv over all tokens
    SrBuf& r = h->sr;
    const int Mr = b.sr_rows, Kc = b.sr_pad;
    {
      KScope k(h, KP_WGRAD, w, 2.0 * ((double)Mr * ((double)d * ff * 2 + (double)d * d) + (double)M * d * d * 3),
               h->wg_adam ? 28.0 * ((double)d * ff * 2 + (double)d * d * 4) : 0, 2);
      bool grouped = wgrad_group_ok(h, st, M);
      if (!grouped && h->wg_adam) return fail(-1, "optimizer-in-wgrad step without the grouped wgrad launch");
      if (grouped) {
        TnGroup g;
        memset(&g, 0, sizeof(g));
        auto set = [&](int i, const bf16_t* A, int lda, int Mo, const bf16_t* Bm, int ldb, int No, float* out, int ldo,
                       int trans) {
          TnProblem& q = g.p[i];
          q.A = A; q.lda = lda; q.M = Mo; q.B = Bm; q.ldb = ldb; q.N = No; q.out = out; q.ldo = ldo; q.trans_out = trans;
        };
        g.n = 3;
        g.K = Kc;
        set(0, r.dx16_c, dp, d, r.g_c, fp, ff, G(h, p.w2.w), d, 1);
        set(1, r.h2_c, dp, d, r.dpre_c, fp, ff, G(h, p.w1.w), ff, 0);
        set(2, r.a_c, dp, d, r.xmid16_c, dp, d, G(h, p.wo.w), d, 0);
        g.overwrite = h->grad_overwrite;
        if (bias_in_wgrad_ok(h, st, M)) {
          g.p[0].csum = G(h, p.b2); g.p[1].csum = G(h, p.b1); g.p[2].csum = G(h, p.bo);
        }
        if (h->wg_adam) {
          arm_adam(h, g, 0, p.w2); arm_adam(h, g, 1, p.w1); arm_adam(h, g, 2, p.wo);
        }
        CHK(launch_big_tn_group(g, w, 1));
        memset(&g, 0, sizeof(g));
        g.n = 1;
        g.K = M;
        set(0, a.h1, dp, d, b.dqkv, qp, 3 * d, G(h, p.wqkv.w), 3 * d, 0);
        g.overwrite = h->grad_overwrite;
        if (h->wg_adam) arm_adam(h, g, 0, p.wqkv);
        CHK(launch_big_tn_group(g, w, 1));
      } else {
        CHK(wgrad_layer_tensor(h, r.g_c, fp, ff, r.dx16_c, dp, d, Kc, G(h, p.w2.w), d, w, b.slab));
        CHK(wgrad_layer_tensor(h, r.h2_c, dp, d, r.dpre_c, fp, ff, Kc, G(h, p.w1.w), ff, w, b.slab));
        CHK(wgrad_layer_tensor(h, r.a_c, dp, d, r.xmid16_c, dp, d, Kc, G(h, p.wo.w), d, w, b.slab));
        CHK(wgrad_layer_tensor(h, a.h1, dp, d, b.dqkv, qp, 3 * d, M, G(h, p.wqkv.w), 3 * d, w, b.slab));
      }
    }
    {
      KScope kpg(h, KP_PARAM_GRADS, w, 0, (double)Mr * (ff * 2.0 + d * 12.0) + (double)M * d * 6.0, 2);
      ColTasks ts;
      memset(&ts, 0, sizeof(ts));
      ts.M = Mr;  // compact rows: LayerNorm 2 - and, unless the wgrad launch above took them (bias_in_wgrad), the three biases
      ts.t[0].dh = r.dh2_c; ts.t[0].ld16 = dp; ts.t[0].x = r.x_mid_c; ts.t[0].mean = r.mean2_c; ts.t[0].rstd = r.rstd2_c;
      ts.t[0].dgamma = G(h, p.ln2_g); ts.t[0].dbeta = G(h, p.ln2_b); ts.t[0].C = d;
      ts.n = 1;
      if (!(wgrad_group_ok(h, st, M) && bias_in_wgrad_ok(h, st, M))) {
        ts.t[0].dy = r.dx16_c; ts.t[0].ldy = dp; ts.t[0].dbias = G(h, p.b2);
        ts.t[1].dy = r.dpre_c; ts.t[1].ldy = fp; ts.t[1].dbias = G(h, p.b1); ts.t[1].C = ff;
        ts.t[2].dy = r.xmid16_c; ts.t[2].ldy = dp; ts.t[2].dbias = G(h, p.bo); ts.t[2].C = d;
        ts.n = 3;
      }
      CHK(launch_col_tasks(ts, w));
      memset(&ts, 0, sizeof(ts));
      ts.n = 1;
      ts.M = M;  // all rows: LayerNorm 1
      ts.t[0].dh = b.dh1; ts.t[0].ld16 = dp; ts.t[0].x = a.x_in; ts.t[0].mean = a.mean1; ts.t[0].rstd = a.rstd1;
      ts.t[0].dgamma = G(h, p.ln1_g); ts.t[0].dbeta = G(h, p.ln1_b); ts.t[0].C = d;
      CHK(launch_col_tasks(ts, w));
    }
    if (two) sc.ev_batch[b.q] = stream_mark(h, w);
    return 0;
  }
  bool bias_done = false;  // the three Dense bias gradients came out of the grouped wgrad launch (TnProblem::csum)
  {
    // (with the optimizer in the epilogue the class also moves 28 bytes per layer-kernel parameter: reported beside the FLOPs)
    KScope k(h, KP_WGRAD, w, 2.0 * (double)M * ((double)d * ff * 2 + (double)d * d * 4),
             h->wg_adam ? 28.0 * ((double)d * ff * 2 + (double)d * d * 4) : 0, h->wgrad_big ? h->wgrad_parts : 8);
    const int rc = (h->skip & 1) ? 0 : wgrad_layer_group(h, st, p, a, b.xin16, b.dpre, b.xmid16, b.dqkv, M, w);
    if (rc < 0) return rc;
    bias_done = (rc == 0) && bias_in_wgrad_ok(h, st, M);
    if (rc > 0 && h->wg_adam) return fail(-1, "optimizer-in-wgrad step without the grouped wgrad launch");
    if (rc > 0) {
      CHK(wgrad_layer_tensor(h, a.g, fp, ff, b.xin16, dp, d, M, G(h, p.w2.w), d, w, b.slab));
      CHK(wgrad_layer_tensor(h, a.h2, dp, d, b.dpre, fp, ff, M, G(h, p.w1.w), ff, w, b.slab));
      CHK(wgrad_layer_tensor(h, a.a, dp, d, b.xmid16, dp, d, M, G(h, p.wo.w), d, w, b.slab));
      CHK(wgrad_layer_tensor(h, a.h1, dp, d, b.dqkv, qp, 3 * d, M, G(h, p.wqkv.w), 3 * d, w, b.slab));
    }
  }
  // the HBM-bound column sums share CUs with anything: their own stream, so they run beside the wgrad launches
  // instead of behind them (the wgrad stream was as long as the dgrad chain with them in line)
  // use_lite: 1 = the handle's own extra stream, 2 = the third (aux) stream, which idles while the cross-modal stack runs
  hipStream_t c = w;
  if (two && w == h->side) {
    if (h->use_lite == 1 && h->lite) c = h->lite;
    else if (h->use_lite == 2 && h->aux && h->use_aux) c = h->aux;
  }
  if (c != w) (void)hipStreamWaitEvent(c, rel, 0);  // same release event as the wgrad launches
  if (h->ln_split && b.part2) {
    // ln_cs: the two LayerNorm backward kernels of the chain already left their column sums as per-workgroup partials;
    // what remains is the dense_1 bias (column sum of dpre) and two small reduces over those partial rows
    KScope kpg(h, KP_PARAM_GRADS, c, 0, (double)M * ff * 2.0 + 2.0 * b.part_blocks * 3.0 * d * 4.0, 3);
    if (!(h->skip & 2)) {
      if (!bias_done) CHK(launch_colsum_bf16(b.dpre, fp, G(h, p.b1), M, ff, ff, c));
      CHK(launch_colreduce(b.part2, b.part_blocks, d, G(h, p.ln2_g), G(h, p.ln2_b), bias_done ? nullptr : G(h, p.b2), c));
      CHK(launch_colreduce(b.part1, b.part_blocks, d, G(h, p.ln1_g), G(h, p.ln1_b), bias_done ? nullptr : G(h, p.bo), c));
    }
  } else if (h->ln_split) {
    // ONE launch: the dense_1 bias (column sum of dpre) and, per LayerNorm, gamma / beta from the dgrad output
    // plus the bias gradient of the GEMM that fed the residual add (dense_2 / to_out) = column sum of the
    // gradient that entered it, taken from its bf16 copy (xin16 / xmid16)
    // bytes: the two LayerNorms read dh (bf16) + x (fp32) = 6 B per element; the bias sums - when the wgrad launch did not
    // take them (bias_in_wgrad) - add dpre and the two residual gradients (bf16)
    KScope kpg(h, KP_PARAM_GRADS, c, 0, (double)M * (bias_done ? d * 12.0 : ff * 2.0 + d * 16.0), 1);
    ColTasks ts;
    memset(&ts, 0, sizeof(ts));
    ts.M = M;
    ts.t[0].dh = b.dh2; ts.t[0].ld16 = dp; ts.t[0].x = a.x_mid; ts.t[0].mean = a.mean2; ts.t[0].rstd = a.rstd2;
    ts.t[0].dgamma = G(h, p.ln2_g); ts.t[0].dbeta = G(h, p.ln2_b); ts.t[0].C = d;
    ts.t[1].dh = b.dh1; ts.t[1].ld16 = dp; ts.t[1].x = a.x_in; ts.t[1].mean = a.mean1; ts.t[1].rstd = a.rstd1;
    ts.t[1].dgamma = G(h, p.ln1_g); ts.t[1].dbeta = G(h, p.ln1_b); ts.t[1].C = d;
    ts.n = 2;
    if (!bias_done) {
      // the bias gradient of the GEMM that fed each residual add (dense_2 / to_out) = column sum of the gradient that
      // entered it, from its bf16 copy, and the dense_1 bias = column sum of dpre
      ts.t[0].dy = b.xin16; ts.t[0].ldy = dp; ts.t[0].dbias = G(h, p.b2);
      ts.t[1].dy = b.xmid16; ts.t[1].ldy = dp; ts.t[1].dbias = G(h, p.bo);
      ts.t[2].dy = b.dpre; ts.t[2].ldy = fp; ts.t[2].dbias = G(h, p.b1); ts.t[2].C = ff;
      ts.n = 3;
    }
    if (!(h->skip & 2)) CHK(launch_col_tasks(ts, c));
  } else if (!bias_done) {
    KScope kpg(h, KP_PARAM_GRADS, c, 0, (double)M * ff * 2.0, 1);
    CHK(launch_colsum_bf16(b.dpre, fp, G(h, p.b1), M, ff, ff, c));
  }
  if (c != w) stream_after(h, c, w);  // the batch marker on the wgrad stream covers both
  if (two) sc.ev_batch[b.q] = stream_mark(h, w);
  return 0;
}

// On entry dx / dx16 hold dL/dx_out of layer l; on exit dL/dx_in (dx in place, dx16 re-pointed to the
// boundary buffer this layer wrote).
//
// Stream plan (round 2).  The dgrad / attention / LayerNorm chain stays on `s` and is the critical path; what
// only the optimizer reads (wgrads, bias and LayerNorm gradients) is ONE batch per layer on the wgrad stream.
// Every big-tile kernel owns its CU (104-136 KiB of LDS), so two of them on different streams exclude each
// other: a 190-workgroup, 105 us wgrad launch next to the 240-workgroup GELU' dgrad made the latter wait for it
// (122 us instead of 42, round-2 timeline).  Therefore
//   * the batch of layer l is released behind the GELU' dgrad of the NEXT layer of the chain (that kernel then
//     has the chip to itself), and the wgrad launch is cut into `wgrad_parts` = 2 pieces of ~95 workgroups: it
//     runs beside the narrow rest of the chain (115-160 workgroups: N = 800 dgrads, attention) without
//     taking CUs from it and is over before the following GELU' dgrad starts;
//   * every buffer both streams touch exists kBwBuf = 3 times, indexed by layer parity q: the chain only ever
//     waits for a batch released two to three layers earlier (ev_batch), never for one just enqueued:
//       dpre[q], dqkv[q], xmid[q], dh2[q], dh1[q] : written by this layer, read by this layer's batch; previous
//                                  readers = batch(layer+3) -> waited for at layer entry;
//       xb[q]                    : this layer's output gradient; its previous contents were the INPUT of
//                                  layer+2, read by batch(layer+2) -> waited for before the last LayerNorm bwd.
int layer_backward(FactHandle* h, Stack& st, int l, int B, float* dx, bf16_t*& dx16, hipStream_t s,
                   BwScratch& sc) {
  const int M = B * st.n, d = st.d, ff = st.ff, dp = st.dp, fp = st.fp, qp = st.qp;
  LayerP& p = st.lp[l];
  LayerA& a = st.la[l];
  // chain 1 keeps its wgrads on its own stream when it really runs beside chain 0 (h->wgrad_tr: the
  // transpose fallback shares one scratch pair, so it stays on the single wgrad stream)
  const bool inl = sc.inline_wgrad && h->wgrad_tr && s == h->aux;
  hipStream_t w = inl ? s : side_of(h, s);
  const bool two = (w != s);
  const int q = (int)(sc.bw_i++ % kBwBuf);
  const bool split = h->ln_split != 0;
  bf16_t* dpre = sc.dpre_pp[q];
  bf16_t* dqkv = sc.dqkv_pp[q];
  bf16_t* xmid16 = sc.xmid_pp[q];
  bf16_t* xout16 = sc.xb_pp[q];
  bf16_t* dh2 = split ? sc.dh2_pp[q] : sc.dh;
  bf16_t* dh1 = split ? sc.dh1_pp[q] : sc.dh;
  const bf16_t* xin16 = dx16;
  if (two && sc.ev_batch[q] && sc.ev_batch[q] != sc.ev_waited) (void)hipStreamWaitEvent(s, sc.ev_batch[q], 0);
  // ---- MLP block: x_out = x_mid + W2 gelu(W1 LN2(x_mid) + b1) + b2
  const double Md = (double)M, fl_attn = 4.0 * (double)B * st.H * (double)st.n * st.n * st.dh;
  {
    KScope k(h, KP_GELU_DGRAD, s, 2.0 * Md * ff * d);
    GemmParams g = gp(xin16, dp, p.w2.s, p.w2.lds, M, ff, d);
    g.ep.out0 = dpre; g.ep.ldo0 = fp; g.ep.pre = a.pre; g.ep.ldp = fp;
    if (!(h->skip & 16)) CHK(launch_gemm_nt(EPI_GELU_BWD, g, s));
  }
  CHK(flush_batch(h, sc));  // the previous layer's batch: released behind the kernel just enqueued
  {
    KScope k(h, KP_DFFN1, s, 2.0 * Md * ff * d);
    GemmParams g = gp(dpre, fp, p.w1.s, p.w1.lds, M, d, ff);
    g.ep.out0 = dh2; g.ep.ldo0 = dp;
    if (h->bwd_splitk == 1 || (h->bwd_splitk == 2 && M <= 4096)) with_ws(h, g, s);
    if (!(h->skip & 32)) CHK(launch_gemm_nt(EPI_BF16, g, s));
  }
  KScope* kln = new KScope(h, KP_LN_BWD, s, 0, Md * d * 16.0);
  // (the batch must be flushed AFTER this layer's LayerNorm 1 backward has produced its partials: wgrad_defer)
  const bool cs = split && h->ln_cs && h->wgrad_defer && d <= 1024 && sc.lnpart_pp[q][0] && sc.lnpart_pp[q][1];
  float* part2 = cs ? sc.lnpart_pp[q][0] : nullptr;
  float* part1 = cs ? sc.lnpart_pp[q][1] : nullptr;
  if (h->skip & 8) {
  } else if (cs)
    CHK(launch_ln_bwd_dx_cs(dh2, a.x_mid, a.mean2, a.rstd2, P(h, p.ln2_g), dx, dx, xmid16, part2, M, d, dp, s));
  else if (split)
    CHK(launch_ln_bwd_dx(dh2, a.x_mid, a.mean2, a.rstd2, P(h, p.ln2_g), dx, dx, xmid16, M, d, dp, s));
  else
    CHK(launch_ln_bwd(dh2, a.x_mid, a.mean2, a.rstd2, P(h, p.ln2_g), dx, dx, xmid16, G(h, p.ln2_g),
                      G(h, p.ln2_b), G(h, p.b2), sc.ln_ws, M, d, dp, s));
  delete kln;
  // ---- attention block: x_mid = x_in + Wo attn(Wqkv LN1(x_in)) + bo
  {
    KScope k(h, KP_OUT_DGRAD, s, 2.0 * Md * d * d);
    GemmParams g = gp(xmid16, dp, p.wo.s, p.wo.lds, M, d, d);
    bf16_t* row[1] = {sc.dorow};
    heads_ep(g.ep, st, row, 1);
    if (!(h->skip & 128)) CHK(launch_gemm_nt(EPI_HEADS, g, s));
  }
  {
    KScope k(h, KP_ATTN_BWD, s, 2.5 * fl_attn, 0, 2);
    AttnParams ap = attn_params(st, a, B);
    ap.dorow = sc.dorow; ap.dsum = sc.dsum; ap.dqkv = dqkv;
    if (!(h->skip & 4)) CHK(launch_attn_bwd(ap, s));
  }
  {
    KScope k(h, KP_DQKV, s, 2.0 * Md * 3 * d * d);
    GemmParams g = gp(dqkv, qp, p.wqkv.s, p.wqkv.lds, M, d, 3 * d);
    g.ep.out0 = dh1; g.ep.ldo0 = dp;
    if (h->bwd_splitk == 1 || (h->bwd_splitk == 2 && M <= 4096)) with_ws(h, g, s);
    if (!(h->skip & 64)) CHK(launch_gemm_nt(EPI_BF16, g, s));
  }
  // ---- everything only the optimizer reads: recorded now, released by the next layer (or the caller)
  {
    PendingBatch& b = sc.pend;
    b.valid = true; b.st = &st; b.l = l; b.M = M; b.q = q;
    b.sr_rows = 0; b.sr_pad = 0;
    b.xin16 = xin16; b.dpre = dpre; b.xmid16 = xmid16; b.dqkv = dqkv; b.dh2 = dh2; b.dh1 = dh1;
    b.part2 = (h->skip & 8) ? nullptr : part2; b.part1 = (h->skip & 8) ? nullptr : part1;
    b.part_blocks = ln_cs_blocks(M);
    b.s = s; b.w = w; b.slab = inl ? sc.slab : nullptr;
    if (!h->wgrad_defer) CHK(flush_batch(h, sc));
  }
  // readers of the previous contents of xb[q]: the batch of the layer two steps back
  const int qr = (q + 1) % kBwBuf;  // parity of layer+2 == parity of layer-1
  if (two && sc.ev_batch[qr]) {
    (void)hipStreamWaitEvent(s, sc.ev_batch[qr], 0);
    sc.ev_waited = sc.ev_batch[qr];  // the next layer's entry wait is for this same event
  }
  KScope kln1(h, KP_LN_BWD, s, 0, Md * d * 16.0);
  if (h->skip & 8) {
  } else if (cs)
    CHK(launch_ln_bwd_dx_cs(dh1, a.x_in, a.mean1, a.rstd1, P(h, p.ln1_g), dx, dx, xout16, part1, M, d, dp, s));
  else if (split)
    CHK(launch_ln_bwd_dx(dh1, a.x_in, a.mean1, a.rstd1, P(h, p.ln1_g), dx, dx, xout16, M, d, dp, s));
  else
    CHK(launch_ln_bwd(dh1, a.x_in, a.mean1, a.rstd1, P(h, p.ln1_g), dx, dx, xout16, G(h, p.ln1_g),
                      G(h, p.ln1_b), G(h, p.bo), sc.ln_ws, M, d, dp, s));
  dx16 = xout16;
  return 0;
}

// Backward of the LAST cross-modal layer with the supervised-rows shortcut: on entry sr.dx_c / sr.dx16_c hold
// dL/dx_out of the B*T supervised rows (all other rows are exactly zero); on exit h->dx / dx16 hold dL/dx_in of
// all rows, as after layer_backward.
int layer_backward_sr(FactHandle* h, Stack& st, int l, int B, int T, float* dx, bf16_t*& dx16, hipStream_t s,
                      BwScratch& sc) {
  const int M = B * st.n, Mr = B * T, Kc = (int)rups((size_t)Mr, 64), d = st.d, ff = st.ff, dp = st.dp, fp = st.fp,
            qp = st.qp;
  LayerP& p = st.lp[l];
  LayerA& a = st.la[l];
  SrBuf& r = h->sr;
  hipStream_t w = side_of(h, s);
  const bool two = (w != s);
  const int q = (int)(sc.bw_i++ % kBwBuf);
  bf16_t* dqkv = sc.dqkv_pp[q];
  bf16_t* xout16 = sc.xb_pp[q];
  bf16_t* dh1 = h->ln_split ? sc.dh1_pp[q] : sc.dh;
  if (two && sc.ev_batch[q]) (void)hipStreamWaitEvent(s, sc.ev_batch[q], 0);
  const double Md = (double)M, Mrd = (double)Mr;
  if (Kc > Mr) {  // zero pad rows of every wgrad operand (a smaller batch may have left data there)
    const size_t nr = (size_t)(Kc - Mr);
    HIPCHK(hipMemsetAsync(r.dx16_c + (size_t)Mr * dp, 0, nr * dp * 2, s));
    HIPCHK(hipMemsetAsync(r.g_c + (size_t)Mr * fp, 0, nr * fp * 2, s));
    HIPCHK(hipMemsetAsync(r.h2_c + (size_t)Mr * dp, 0, nr * dp * 2, s));
    HIPCHK(hipMemsetAsync(r.dpre_c + (size_t)Mr * fp, 0, nr * fp * 2, s));
    HIPCHK(hipMemsetAsync(r.a_c + (size_t)Mr * dp, 0, nr * dp * 2, s));
    HIPCHK(hipMemsetAsync(r.xmid16_c + (size_t)Mr * dp, 0, nr * dp * 2, s));
  }
  {
    KScope k(h, KP_GELU_DGRAD, s, 2.0 * Mrd * ff * d);
    GemmParams g = gp(r.dx16_c, dp, p.w2.s, p.w2.lds, Mr, ff, d);
    with_skinny(h, g);
    g.ep.out0 = r.dpre_c; g.ep.ldo0 = fp; g.ep.pre = r.pre_c; g.ep.ldp = fp;
    CHK(launch_gemm_nt(EPI_GELU_BWD, g, s));
  }
  CHK(flush_batch(h, sc));
  {
    KScope k(h, KP_DFFN1, s, 2.0 * Mrd * ff * d);
    GemmParams g = gp(r.dpre_c, fp, p.w1.s, p.w1.lds, Mr, d, ff);
    with_skinny(h, g);
    g.ep.out0 = r.dh2_c; g.ep.ldo0 = dp;
    CHK(launch_gemm_nt(EPI_BF16, g, s));
  }
  {
    KScope k(h, KP_LN_BWD, s, 0, Mrd * d * 16.0);
    if (h->ln_split)
      CHK(launch_ln_bwd_dx(r.dh2_c, r.x_mid_c, r.mean2_c, r.rstd2_c, P(h, p.ln2_g), r.dx_c, r.dx_c, r.xmid16_c, Mr, d, dp, s));
    else
      return fail(-1, "supervised-rows shortcut needs ln_split");
  }
  // per-head dO rows: everything beyond the supervised rows must read as zero (all kernel families)
  HIPCHK(hipMemsetAsync(sc.dorow, 0, (size_t)B * st.H * st.NP * st.dhp * sizeof(bf16_t), s));
  {
    KScope k(h, KP_OUT_DGRAD, s, 2.0 * Mrd * d * d);
    GemmParams g = gp(r.xmid16_c, dp, p.wo.s, p.wo.lds, Mr, d, d);
    with_skinny(h, g);
    bf16_t* row[1] = {sc.dorow};
    heads_ep(g.ep, st, row, 1);
    g.ep.n_tok = T;  // compact row -> (b, t) with T rows per sequence
    CHK(launch_gemm_nt(EPI_HEADS, g, s));
  }
  {
    KScope k(h, KP_ATTN_BWD, s, 2.5 * 4.0 * (double)B * st.H * (double)T * st.n * st.dh, 0, 2);
    AttnParams ap = attn_params(st, a, B);
    ap.dorow = sc.dorow; ap.dsum = sc.dsum; ap.dqkv = dqkv;
    ap.nq = T;
    CHK(launch_attn_bwd(ap, s));
  }
  {
    KScope k(h, KP_DQKV, s, 2.0 * Md * 3 * d * d);
    GemmParams g = gp(dqkv, qp, p.wqkv.s, p.wqkv.lds, M, d, 3 * d);
    g.ep.out0 = dh1; g.ep.ldo0 = dp;
    if (h->bwd_splitk == 1 || (h->bwd_splitk == 2 && M <= 4096)) with_ws(h, g, s);
    CHK(launch_gemm_nt(EPI_BF16, g, s));
  }
  // residual gradient of all rows: the compact rows scattered, zero elsewhere
  CHK(launch_scatter_rows_zero(r.dx_c, B, st.n, T, d, dx, s));
  {
    PendingBatch& b = sc.pend;
    b = PendingBatch();
    b.valid = true; b.st = &st; b.l = l; b.M = M; b.q = q;
    b.dqkv = dqkv; b.dh1 = dh1; b.part2 = nullptr; b.part1 = nullptr;
    b.s = s; b.w = w; b.slab = nullptr;
    b.sr_rows = Mr; b.sr_pad = Kc; b.sr_B = B; b.sr_T = T;
    if (!h->wgrad_defer) CHK(flush_batch(h, sc));
  }
  const int qr = (q + 1) % kBwBuf;
  if (two && sc.ev_batch[qr]) (void)hipStreamWaitEvent(s, sc.ev_batch[qr], 0);
  KScope kln1(h, KP_LN_BWD, s, 0, Md * d * 16.0);
  CHK(launch_ln_bwd_dx(dh1, a.x_in, a.mean1, a.rstd1, P(h, p.ln1_g), dx, dx, xout16, M, d, dp, s));
  dx16 = xout16;
  return 0;
}

// embed: x0 = in W + b + pos   (base_models.py:130-156)
int embed_forward(FactHandle* h, Stack& st, const float* in, size_t batch_stride, int B, hipStream_t s) {
  const int M = B * st.n;
  CHK(launch_pad_cast(in, st.n, batch_stride, M, st.feat, st.xin16, st.featp, s));
  GemmParams g = gp(st.xin16, st.featp, st.emb.t, st.emb.ldt, M, st.d, st.featp);
  g.ep.out0 = st.x0; g.ep.ldo0 = st.d; g.ep.bias = P(h, st.emb_b); g.ep.pos = P(h, st.pos); g.ep.seq = st.n;
  CHK(launch_gemm_nt(EPI_F32_BIAS_POS, g, s));
  return 0;
}

int embed_backward(FactHandle* h, Stack& st, int B, float* dx, bf16_t* dx16, hipStream_t s, BwScratch& sc) {
  const int M = B * st.n;
  const bool inl = sc.inline_wgrad && h->wgrad_tr && s == h->aux;
  hipStream_t w = inl ? s : side_of(h, s);  // wgrads share the wgrad stream unless the chain keeps its own
  if (w != s) stream_after(h, s, w);
  CHK(wgrad(h, st.xin16, st.featp, st.feat, dx16, st.dp, st.d, M, G(h, st.emb.w), st.d, w, inl ? sc.slab : nullptr));
  if (w != s) {  // it read the last boundary buffer
    hipEvent_t e = stream_mark(h, w);
    for (int q = 0; q < kBwBuf; ++q) sc.ev_batch[q] = e;
  }
  CHK(launch_colsum_f32(dx, st.d, G(h, st.emb_b), M, st.d, st.d, s));
  CHK(launch_possum(dx, G(h, st.pos), B, st.n, st.d, s));
  return 0;
}

int copy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
           hipStream_t s) {
  hipError_t e = hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToDevice, s);
  return e == hipSuccess ? 0 : -100;
}

// forward through both encoders and the cross-modal stack; final hidden states in cross.out()
int model_forward_hidden(FactHandle* h, const float* motion, size_t m_stride, const float* audio,
                         size_t a_stride, int B, hipStream_t s, int sr_T = 0) {
  Stack &mo = h->motion, &au = h->audio, &cr = h->cross;
  for (Stack* st : {&mo, &au, &cr})  // (a forward that failed half-way must not leave a "LayerNorm 1 already done" mark behind)
    for (LayerA& la : st->la) la.h1_ready = false;
  hipStream_t w = side_of(h, s);
  if (w != s) stream_after(h, s, w);
  CHK(embed_forward(h, au, audio, a_stride, B, w));
  for (int l = 0; l < au.L; ++l) CHK(layer_forward(h, au, l, B, w));
  CHK(embed_forward(h, mo, motion, m_stride, B, s));
  for (int l = 0; l < mo.L; ++l) CHK(layer_forward(h, mo, l, B, s));
  if (w != s) stream_after(h, w, s);
  // tf.concat([motion, audio], axis=1)  (base_models.py:192-193)
  CHK(launch_concat_seq(mo.out(), au.out(), B, mo.n, au.n, cr.d, cr.x0, s));
  for (int l = 0; l < cr.L; ++l) {
    if (sr_T > 0 && l == cr.L - 1) CHK(layer_forward_sr(h, cr, l, B, sr_T, s));  // supervised rows only (training)
    else CHK(layer_forward(h, cr, l, B, s));
  }
  return 0;
}

// Every kernel that contributes to the next bucket's gradients has been enqueued (main + side): make the
// caller's communication stream wait for them and tell the host, which enqueues the all-reduce of
// that range there (and, with a fused optimizer step, the bucket's Adam right behind it).  Without
// a callback and with fact_adam_begin pending, the bucket's Adam + shadow refresh is enqueued here
// on the optimizer stream: it is HBM-bound while the remaining backward GEMMs are LDS-DMA/MFMA-bound,
// so it hides behind them.  Nothing blocks the host.
int notify_grads(FactHandle* h, hipStream_t s) {
  const int b = h->cb_bucket++;
  const Bucket& k = h->buckets[b];
  if (h->cb) {
    stream_after(h, s, h->cb_stream);
    if (side_of(h, s) != s) stream_after(h, h->side, h->cb_stream);
    h->cb(h->cb_user, b, k.off, k.cnt);
  } else if (h->adam_pending) {
    // In-backward optimizer.  Updating every bucket the moment it is final slows the dense cross-modal
    // backward by more than it hides, so the head + cross-modal buckets are held back and
    // updated together when the LAST cross-modal bucket is final: the HBM-bound update then runs beside
    // the two small encoder stacks' backward, which leaves most of the chip idle.  The encoder buckets
    // follow as they complete.
    // adam_hold = k >= 1: the release point is the bucket of cross layer k - 1 (k = 1: the last one, layer 0 - whose wgrad
    // batch is only flushed when the chain has ended, i.e. the update starts ~0.3 ms into the tail; k = 2: layer 1, final
    // about when the chain ends - the last cross bucket then follows on its own like the encoder buckets)
    const int last_cross = h->cross.L;  // bucket 0 = head, 1..L = cross layers L-1..0
    const int release = h->adam_hold ? std::max(0, last_cross - (h->adam_hold - 1)) : 0;
    if (h->adam_hold && b < release) return 0;
    const int first = (h->adam_hold && b == release) ? 0 : b;
    stream_after(h, s, h->opt);
    if (side_of(h, s) != s) stream_after(h, h->side, h->opt);
    for (int i = first; i <= b; ++i) CHK(adam_bucket(h, i, h->opt));
  }
  return 0;
}

int check_batch(FactHandle* h, int B) {
  if (!h) return fail(-1, "null handle");
  if (B <= 0 || B > h->max_batch)
    return fail(-1, "batch " + std::to_string(B) + " outside (0, max_batch=" + std::to_string(h->max_batch) + "]");
  return 0;
}

}  // namespace

// =============================================================================================
extern "C" {

int fact_abi_version(void) { return FACT_ABI_VERSION; }

unsigned int fact_crc32c(const void* data, size_t n, unsigned int crc) {
  // slicing-by-8, tables built on first use (reflected polynomial 0x82F63B78)
  // (function-local static initialised by a lambda: thread-safe - the input prefetch thread and checkpoint code both call in)
  struct Tables { uint32_t t[8][256]; };
  static const Tables tables = [] {
    Tables r;
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      r.t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) r.t[t][i] = (r.t[t - 1][i] >> 8) ^ r.t[0][r.t[t - 1][i] & 0xff];
    return r;
  }();
  const uint32_t (*T)[256] = tables.t;
  const unsigned char* p = (const unsigned char*)data;
  uint32_t c = ~crc;
  while (n >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = T[7][lo & 0xff] ^ T[6][(lo >> 8) & 0xff] ^ T[5][(lo >> 16) & 0xff] ^ T[4][lo >> 24] ^ T[3][hi & 0xff] ^
        T[2][(hi >> 8) & 0xff] ^ T[1][(hi >> 16) & 0xff] ^ T[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = T[0][(c ^ *p++) & 0xff] ^ (c >> 8);
  return ~c;
}
const char* fact_last_error(void) { return g_err.c_str(); }

// Engine streams; an environment variable (A/B knob) may ask for a dispatch priority: -1 = highest, 1 = lowest the
// device offers, unset / 0 = default.
static hipError_t make_stream(hipStream_t* s, const char* env) {
  const char* v = getenv(env);
  const int want = v ? atoi(v) : 0;
  if (!want) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  return hipStreamCreateWithPriority(s, hipStreamNonBlocking, want > 0 ? least : greatest);
}

static void init_geo(FactHandle* h, const FactConfig* cfg) {
  h->cfg = *cfg;
  if (h->cfg.ln_eps <= 0.f) h->cfg.ln_eps = 1e-5f;
  init_stack_geo(h->motion, "motion", cfg->motion, cfg->motion.seq_len, cfg->motion.feature_dim);
  init_stack_geo(h->audio, "audio", cfg->audio, cfg->audio.seq_len, cfg->audio.feature_dim);
  init_stack_geo(h->cross, "cross", cfg->cross, cfg->motion.seq_len + cfg->audio.seq_len, 0);
  h->outp = rup(cfg->out_dim, 32);
}

int fact_arena_size(const FactConfig* cfg, size_t* arena_floats, int* n_tensors) {
  if (!cfg) return fail(-1, "null config");
  int rc = validate(*cfg);
  if (rc) return rc;
  FactHandle tmp;
  init_geo(&tmp, cfg);
  std::vector<FactParamDesc> t;
  size_t total = 0;
  build_table(&tmp, &t, &total);
  if (arena_floats) *arena_floats = total;
  if (n_tensors) *n_tensors = (int)t.size();
  return 0;
}

int fact_create(const FactConfig* cfg, int max_batch, int training, const FactArenas* arenas,
                FactHandle** out) {
  if (!cfg || !out) return fail(-1, "null argument");
  if (max_batch <= 0) return fail(-1, "max_batch must be positive");
  int rc = validate(*cfg);
  if (rc) return rc;
  FactHandle* h = new FactHandle();
  init_geo(h, cfg);
  h->max_batch = max_batch;
  h->training = training != 0;
  h->keep_pre_infer = getenv("FACT_KEEP_PRE") != nullptr;
  build_table(h, &h->table, &h->arena_floats);
  const size_t abytes = h->arena_floats * sizeof(float);
  if (arenas && arenas->params) {
    h->params = arenas->params;
    h->grads = arenas->grads;
    h->adam_m = arenas->adam_m;
    h->adam_v = arenas->adam_v;
    if (h->training && (!h->grads || !h->adam_m || !h->adam_v)) {
      delete h;
      return fail(-1, "training handle needs grads/adam_m/adam_v arenas");
    }
    // a new handle starts with accumulate semantics (grad_overwrite = 0): whatever an earlier handle left in the
    // caller's gradient arena - ranges that only ever got plain stores under grad_overwrite included - must not be
    // added to the first gradient of this one
    if (h->training) HIPCHK(hipMemset(h->grads, 0, abytes));
  } else {
    h->own_arenas = true;
    HIPCHK(hipMalloc((void**)&h->params, abytes));
    HIPCHK(hipMemset(h->params, 0, abytes));
    if (h->training) {
      HIPCHK(hipMalloc((void**)&h->grads, abytes));
      HIPCHK(hipMalloc((void**)&h->adam_m, abytes));
      HIPCHK(hipMalloc((void**)&h->adam_v, abytes));
      HIPCHK(hipMemset(h->grads, 0, abytes));
      HIPCHK(hipMemset(h->adam_m, 0, abytes));
      HIPCHK(hipMemset(h->adam_v, 0, abytes));
    }
  }
  {
    Bump b;
    layout_shadow(h, b);
    h->shadow_bytes = rups(b.off, 256) + 256;
    HIPCHK(hipMalloc((void**)&h->shadow, h->shadow_bytes));
    HIPCHK(hipMemset(h->shadow, 0, h->shadow_bytes));
    Bump b2;
    b2.base = h->shadow;
    layout_shadow(h, b2);
  }
  {
    Bump b;
    layout_work(h, b);
    h->work_bytes = rups(b.off, 256) + 256;
    HIPCHK(hipMalloc((void**)&h->work, h->work_bytes));
    HIPCHK(hipMemset(h->work, 0, h->work_bytes));
    Bump b2;
    b2.base = h->work;
    layout_work(h, b2);
  }
  {
    int rc2 = build_buckets(h);
    if (rc2) return rc2;
  }
  // (Stream priorities were tried for the wgrad / column-sum streams in round 2: no gain for the dgrad chain,
  //  and with a second low-priority stream the step time doubled on this runtime - all streams stay default.)
  HIPCHK(make_stream(&h->side, "FACT_PRIO_SIDE"));
  HIPCHK(make_stream(&h->aux, "FACT_PRIO_AUX"));
  for (int i = 0; i < 3; ++i) {
    HIPCHK(hipMalloc((void**)&h->sk_slab[i], kSplitKSlabBytes));
    HIPCHK(hipMalloc((void**)&h->sk_cnt[i], kSplitKCounters * sizeof(unsigned)));
    HIPCHK(hipMemset(h->sk_cnt[i], 0, kSplitKCounters * sizeof(unsigned)));
  }
  {
    size_t wide = (size_t)h->cross.ff;
    if ((size_t)3 * h->cross.d > wide) wide = (size_t)3 * h->cross.d;
    h->skinny_floats = (size_t)512 * rups(wide, 4);
    HIPCHK(hipMalloc((void**)&h->skinny_acc, h->skinny_floats * sizeof(float)));
    HIPCHK(hipMemset(h->skinny_acc, 0, h->skinny_floats * sizeof(float)));
    HIPCHK(hipMalloc((void**)&h->skinny_acc2, h->skinny_floats * sizeof(float)));
    HIPCHK(hipMemset(h->skinny_acc2, 0, h->skinny_floats * sizeof(float)));
  }
  h->ev.resize(1024);  // ~170 records per train step: a stored handle is never re-recorded before its use
  for (hipEvent_t& e : h->ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *out = h;
  return 0;
}

int fact_destroy(FactHandle* h) {
  if (!h) return 0;
  if (h->own_arenas) {
    (void)hipFree(h->params);
    (void)hipFree(h->grads);
    (void)hipFree(h->adam_m);
    (void)hipFree(h->adam_v);
  }
  (void)hipFree(h->shadow);
  (void)hipFree(h->work);
  (void)hipFree(h->cast_table);
  (void)hipFree(h->adam_blocks);
  for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
  if (h->side) (void)hipStreamDestroy(h->side);
  if (h->opt) (void)hipStreamDestroy(h->opt);
  if (h->aux) (void)hipStreamDestroy(h->aux);
  if (h->lite) (void)hipStreamDestroy(h->lite);
  for (BwScratch& sc : h->bw)
    for (int q = 0; q < kBwBuf; ++q)
      for (int i = 0; i < 2; ++i) (void)hipFree(sc.lnpart_pp[q][i]);
  (void)hipFree(h->skinny_acc);
  (void)hipFree(h->skinny_acc2);
  (void)hipFree(h->ar_motion);
  for (int i = 0; i < 3; ++i) {
    (void)hipFree(h->sk_slab[i]);
    (void)hipFree(h->sk_cnt[i]);
  }
  delete h;
  return 0;
}

int fact_param_table(FactHandle* h, const FactParamDesc** table, int* n) {
  if (!h) return fail(-1, "null handle");
  if (table) *table = h->table.data();
  if (n) *n = (int)h->table.size();
  return 0;
}

int fact_arenas(FactHandle* h, FactArenas* out, size_t* arena_floats) {
  if (!h) return fail(-1, "null handle");
  if (out) {
    out->params = h->params;
    out->grads = h->grads;
    out->adam_m = h->adam_m;
    out->adam_v = h->adam_v;
  }
  if (arena_floats) *arena_floats = h->arena_floats;
  return 0;
}

int fact_refresh_weights(FactHandle* h, void* stream) {
  if (!h) return fail(-1, "null handle");
  return refresh_all(h, (hipStream_t)stream);
}

int fact_set_grad_callback(FactHandle* h, fact_grad_cb cb, void* user, void* comm_stream) {
  if (!h) return fail(-1, "null handle");
  h->cb = cb;
  h->cb_user = user;
  h->cb_stream = (hipStream_t)comm_stream;
  return 0;
}

// Gradient ranges the grouped wgrad launch owns under grad_overwrite (every Dense kernel of a transformer layer): with
// the option on nobody zeroes them - the next step's plain stores replace them.  Whenever accumulate semantics
// (re)start on an arena that may hold such values - the option going 1 -> 0, a training handle created on caller
// arenas - they are cleared, otherwise the next fact_forward_backward would add a fresh gradient to a stale one.
static int zero_overwritten_grads(FactHandle* h) {
  if (!h->training || !h->grads) return 0;
  HIPCHK(hipDeviceSynchronize());  // rare host-side switch: nothing of an earlier step may still be writing the arena
  for (Stack* st : {&h->cross, &h->motion, &h->audio})
    for (const LayerP& p : st->lp)
      for (const DenseW* w : {&p.wqkv, &p.wo, &p.w1, &p.w2})
        HIPCHK(hipMemset(h->grads + w->w.off, 0, w->w.numel() * sizeof(float)));
  return 0;
}

/* Production options (include/fact_hip.h). */
int fact_set_option(FactHandle* h, const char* key, int value) {
  if (!h || !key) return fail(-1, "null argument");
  if (!strcmp(key, "grad_overwrite")) {
    if (h->grad_overwrite && !value) CHK(zero_overwritten_grads(h));
    h->grad_overwrite = value != 0;
    return 0;
  }
  if (!strcmp(key, "sr_rows")) {  // supervised-rows shortcut of the last cross-modal layer (training step)
    h->sr_rows = value;
    return 0;
  }
  if (!strcmp(key, "aux_stream")) {
    h->use_aux = value;
    return 0;
  }
  if (!strcmp(key, "side_stream")) {
    h->use_side = value;
    return 0;
  }
  if (!strcmp(key, "adam_in_wgrad")) {  // optimizer step of the layer kernels inside their wgrad launches (FactHandle)
    h->adam_in_wgrad = value != 0;
    return 0;
  }
  return fail(-1, std::string("unknown option ") + key + " (A/B and ablation knobs: fact_debug_set_option, fact_hip_debug.h)");
}

/* Test / bench knobs (include/fact_hip_debug.h): kernel-selection and scheduling A/B switches, the timing-only ablation
 * mask.  Not part of the drop-in surface; several are process-wide. */
int fact_debug_set_option(FactHandle* h, const char* key, int value) {
  if (!h || !key) return fail(-1, "null argument");
  if (!strcmp(key, "fuse_adam_cast")) {
    h->fuse_adam_cast = value;
    return 0;
  }
  if (!strcmp(key, "wgrad_tr")) {
    h->wgrad_tr = value;
    return 0;
  }
  if (!strcmp(key, "wgrad_slab")) {
    h->wgrad_slab = value;
    return 0;
  }
  if (!strcmp(key, "lite_stream")) {
    if (value == 1 && !h->lite) HIPCHK(hipStreamCreateWithFlags(&h->lite, hipStreamNonBlocking));
    h->use_lite = value;
    return 0;
  }
  if (!strcmp(key, "adam_variant")) {  // process-wide: optimizer kernel (rowops.hip launch_adam_fused)
    adam_set_variant(value);
    return 0;
  }
  if (!strcmp(key, "k64")) {  // process-wide: 256x160 NT GEMMs on 64-deep ring slots
    gemm_set_k64(value);
    return 0;
  }
  if (!strcmp(key, "tile128x160")) {  // process-wide: short-K N = 800 GEMMs on 128x160 tiles (225 workgroups at M = 5760)
    gemm_set_tile128x160(value);
    return 0;
  }
  if (!strcmp(key, "tile192")) {  // process-wide: 192x160 tiles for the whole-K N = 800 dgrads
    gemm_set_tile192(value);
    return 0;
  }
  if (!strcmp(key, "skip")) {  // TIMING ONLY: results are wrong while it is set
    h->skip = value;
    return 0;
  }
  if (!strcmp(key, "ln_fuse")) {
    h->ln_fuse = value;
    return 0;
  }
  if (!strcmp(key, "adam_hold")) {
    h->adam_hold = value;
    return 0;
  }
  if (!strcmp(key, "big_impl")) {  // process-wide (gemm_set_big_impl): 1 default, 2 no 256x128 pairs, 0 = 128x128 kernel only
    gemm_set_big_impl(value);
    return 0;
  }
  if (!strcmp(key, "attn_variant")) {  // process-wide: attention kernel family (attention.h)
    attn_set_variant(value);
    return 0;
  }
  if (!strcmp(key, "tn_loop")) {  // process-wide: main loop of the grouped wgrad kernel (gemm_set_tn_cfg)
    gemm_set_tn_cfg(value);
    return 0;
  }
  if (!strcmp(key, "bias_in_wgrad")) {  // 0 = bias gradients by col_tasks_kernel (round-2..5 path), 1 = inside the grouped wgrad launch
    h->bias_in_wgrad = value ? 1 : 0;
    return 0;
  }
  if (!strcmp(key, "wgrad_parts")) {
    h->wgrad_parts = value < 1 ? 1 : value;
    return 0;
  }
  if (!strcmp(key, "wgrad_defer")) {
    h->wgrad_defer = value;
    return 0;
  }
  if (!strcmp(key, "bwd_splitk")) {
    h->bwd_splitk = value;
    return 0;
  }
  if (!strcmp(key, "ln_split")) {
    h->ln_split = value;
    return 0;
  }
  if (!strcmp(key, "ln_cs")) {  // 0 = col_tasks pass; 1 = partials from the dx kernel, 4 rows per wave; 2 = 2 rows per wave
    if (value) CHK(alloc_ln_cs(h));
    h->ln_cs = value;
    if (value) ln_set_cs_rows(value == 2 ? 2 : 4);  // process-wide kernel shape
    return 0;
  }
  if (!strcmp(key, "wgrad_big")) {
    h->wgrad_big = value;
    return 0;
  }
  if (!strcmp(key, "keep_pre")) {  // 1 = inference entry points store the dense_1 pre-activations too (A/B of that store)
    h->keep_pre_infer = value != 0;
    return 0;
  }
  return fact_set_option(h, key, value);  // the production keys are accepted here too
}

static int head_forward(FactHandle* h, int B, float* out, hipStream_t s) {
  Stack& cr = h->cross;
  const int Mc = B * cr.n;
  CHK(launch_pad_cast(cr.out(), Mc, 0, Mc, cr.d, h->xf16, cr.dp, s));
  GemmParams g = gp(h->xf16, cr.dp, h->head.t, h->head.ldt, Mc, h->cfg.out_dim, cr.d);
  g.ep.out0 = out; g.ep.ldo0 = h->cfg.out_dim; g.ep.bias = P(h, h->head_b);
  CHK(launch_gemm_nt(EPI_F32_BIAS, g, s));
  return 0;
}

int fact_forward(FactHandle* h, const float* motion, const float* audio, int B, float* out,
                 void* stream) {
  int rc = check_batch(h, B);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  struct InferScope {  // forward only: nobody reads the pre-activations
    FactHandle* h;
    explicit InferScope(FactHandle* hh) : h(hh) { h->keep_pre = h->keep_pre_infer; }  // debug option "keep_pre"
    ~InferScope() { h->keep_pre = true; }
  } infer_scope(h);
  CHK(model_forward_hidden(h, motion, (size_t)h->motion.n * h->motion.feat, audio,
                           (size_t)h->audio.n * h->audio.feat, B, s));
  return head_forward(h, B, out, s);
}

int fact_forward_backward(FactHandle* h, const float* motion, const float* audio,
                          const float* target, int B, int T, float loss_scale, float* loss_out,
                          void* stream) {
  int rc = check_batch(h, B);
  if (rc) return rc;
  if (!h->training) return fail(-1, "handle was created with training=0");
  Stack &mo = h->motion, &au = h->audio, &cr = h->cross;
  if (T <= 0 || T > cr.n) return fail(-1, "target length outside (0, n_motion+n_audio]");
  hipStream_t s = (hipStream_t)stream;
  const int Mc = B * cr.n, d = cr.d, D = h->cfg.out_dim;
  // supervised-rows shortcut (SrBuf): the last cross-modal layer and the head on the B*T rows the loss reads
  const bool sr = h->sr_rows && h->ln_split && cr.L >= 1 && T <= 32 && 4 * T <= cr.n &&
                  rups((size_t)B * T, 64) <= (size_t)h->sr.rows_max;
  h->wg_adam = wg_adam_ok(h, B);
  CHK(model_forward_hidden(h, motion, (size_t)mo.n * mo.feat, audio, (size_t)au.n * au.feat, B, s, sr ? T : 0));
  HIPCHK(hipMemsetAsync(h->scalars, 0, 16 * sizeof(float), s));
  if (sr) {
    SrBuf& r = h->sr;
    const int Mr = B * T, Kc = (int)rups((size_t)Mr, 64);
    CHK(launch_pad_cast(r.x_out_c, Mr, 0, Mr, d, r.xf16_c, cr.dp, s));
    {
      GemmParams g = gp(r.xf16_c, cr.dp, h->head.t, h->head.ldt, Mr, D, d);
      with_skinny(h, g);
      g.ep.out0 = r.pred_c; g.ep.ldo0 = D; g.ep.bias = P(h, h->head_b);
      CHK(launch_gemm_nt(EPI_F32_BIAS, g, s));
    }
    CHK(launch_mse_loss(r.pred_c, target, h->scalars, r.dpred_c, B, T, T, D, h->outp, loss_scale, s));
    if (loss_out) HIPCHK(hipMemcpyAsync(loss_out, h->scalars, sizeof(float), hipMemcpyDeviceToDevice, s));
    if (Kc > Mr) {
      HIPCHK(hipMemsetAsync(r.dpred_c + (size_t)Mr * h->outp, 0, (size_t)(Kc - Mr) * h->outp * 2, s));
      HIPCHK(hipMemsetAsync(r.xf16_c + (size_t)Mr * cr.dp, 0, (size_t)(Kc - Mr) * cr.dp * 2, s));
    }
    {
      hipStream_t w = side_of(h, s);
      if (w != s) stream_after(h, s, w);
      CHK(wgrad(h, r.xf16_c, cr.dp, d, r.dpred_c, h->outp, D, Kc, G(h, h->head.w), D, w));
    }
    CHK(launch_colsum_bf16(r.dpred_c, h->outp, G(h, h->head_b), Mr, h->outp, D, s));
    {
      GemmParams g = gp(r.dpred_c, h->outp, h->head.s, h->head.lds, Mr, d, h->outp);
      with_skinny(h, g);
      g.ep.out0 = r.dx_c; g.ep.ldo0 = d; g.ep.out1 = r.dx16_c; g.ep.ldo1 = cr.dp;
      CHK(launch_gemm_nt(EPI_F32_BF16, g, s));
    }
  } else {
    CHK(head_forward(h, B, h->pred, s));
    // loss + dL/dpred
    CHK(launch_mse_loss(h->pred, target, h->scalars, h->dpred, B, cr.n, T, D, h->outp, loss_scale, s));
    if (loss_out) HIPCHK(hipMemcpyAsync(loss_out, h->scalars, sizeof(float), hipMemcpyDeviceToDevice, s));
    // head backward
    {
      hipStream_t w = side_of(h, s);
      if (w != s) stream_after(h, s, w);
      CHK(wgrad(h, h->xf16, cr.dp, d, h->dpred, h->outp, D, Mc, G(h, h->head.w), D, w));
    }
    CHK(launch_colsum_bf16(h->dpred, h->outp, G(h, h->head_b), Mc, h->outp, D, s));
    {
      GemmParams g = gp(h->dpred, h->outp, h->head.s, h->head.lds, Mc, d, h->outp);
      g.ep.out0 = h->dx; g.ep.ldo0 = d; g.ep.out1 = h->dx16; g.ep.ldo1 = cr.dp;
      CHK(launch_gemm_nt(EPI_F32_BF16, g, s));
    }
  }
  // Gradient buckets are contiguous arena ranges reported in the order they become final:
  // head, cross layers L-1..0, audio stack, motion stack (fact_set_grad_callback).
  h->cb_bucket = 0;
  CHK(notify_grads(h, s));  // head
  bf16_t* g16 = h->dx16;  // bf16 gradient at the current layer boundary (re-pointed by every layer)
  for (int l = cr.L - 1; l >= 0; --l) {
    if (sr && l == cr.L - 1) CHK(layer_backward_sr(h, cr, l, B, T, h->dx, g16, s, h->bw[0]));
    else CHK(layer_backward(h, cr, l, B, h->dx, g16, s, h->bw[0]));
    // layer l+1's optimizer-only batch was released inside the call above: its bucket is complete now
    if (l < cr.L - 1) CHK(notify_grads(h, s));
  }
  CHK(flush_batch(h, h->bw[0]));
  if (cr.L > 0) CHK(notify_grads(h, s));  // cross layer 0
  CHK(launch_split_grad(h->dx, B, mo.n, au.n, d, h->dxm, h->dxm16, h->dxa, h->dxa16, cr.dp, s));
  // The two encoder stacks are independent from here on: the motion stack's backward chain runs on the
  // handle's third stream (own scratch, chain 1) beside the audio stack's chain on the caller's stream;
  // both feed the one wgrad stream.  Alone, the 1920- and 3840-token GEMMs leave most of the chip idle
  // (the four encoder layers used to take 1.5 ms of a 10.4 ms step).
  hipStream_t ms = (h->use_side && h->use_aux && h->aux) ? h->aux : s;
  if (ms != s) stream_after(h, s, ms);
  // (audio first: its big-tile GEMMs need whole CUs; enqueued second they wait ~0.5 ms behind the motion
  //  chain's one-workgroup-per-CU kernels - rocprofv3 timeline, tools/attic/tail_view.py)
  g16 = h->dxa16;
  for (int l = au.L - 1; l >= 0; --l) CHK(layer_backward(h, au, l, B, h->dxa, g16, s, h->bw[0]));
  CHK(flush_batch(h, h->bw[0]));
  CHK(embed_backward(h, au, B, h->dxa, g16, s, h->bw[0]));
  bf16_t* m16 = h->dxm16;
  for (int l = mo.L - 1; l >= 0; --l) CHK(layer_backward(h, mo, l, B, h->dxm, m16, ms, h->bw[1]));
  CHK(flush_batch(h, h->bw[1]));
  CHK(embed_backward(h, mo, B, h->dxm, m16, ms, h->bw[1]));
  CHK(notify_grads(h, s));   // audio stack
  CHK(notify_grads(h, ms));  // motion stack
  if (ms != s) stream_after(h, ms, s);
  if (side_of(h, s) != s) stream_after(h, h->side, s);  // join: the caller's stream sees all gradients
  if (h->adam_pending && !h->cb) {  // fused optimizer step: join the optimizer stream, step is complete
    stream_after(h, h->opt, s);
    h->adam_pending = false;
  }
  h->wg_adam = false;
  // the caller's stream has joined the side stream: no reader of the backward scratch is left
  for (BwScratch& sc : h->bw) {
    for (int q = 0; q < kBwBuf; ++q) sc.ev_batch[q] = nullptr;
    sc.ev_waited = nullptr;
  }
  return 0;
}

int fact_adam_step(FactHandle* h, float lr, float beta1, float beta2, float eps, float clip_norm,
                   void* stream) {
  if (!h) return fail(-1, "null handle");
  if (!h->training) return fail(-1, "handle was created with training=0");
  hipStream_t s = (hipStream_t)stream;
  float gscale = 1.0f;
  if (clip_norm > 0.f) {
    // tf.clip_by_global_norm (single_task_trainer.py:180-183)
    HIPCHK(hipMemsetAsync(h->scalars + 8, 0, sizeof(float), s));
    CHK(launch_sumsq(h->grads, h->arena_floats, h->scalars + 8, s));
    float ss = 0.f;
    HIPCHK(hipMemcpyAsync(&ss, h->scalars + 8, sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const float norm = sqrtf(ss);
    gscale = clip_norm / fmaxf(norm, clip_norm);
  }
  h->step += 1;
  const double t = (double)h->step;
  const double lr_t = (double)lr * std::sqrt(1.0 - std::pow((double)beta2, t)) / (1.0 - std::pow((double)beta1, t));
  if (h->fuse_adam_cast && !h->buckets.empty()) {
    const Bucket& last = h->buckets.back();
    return launch_adam_fused(h->adam_blocks, last.blk_first + last.blk_n, h->params, h->adam_m, h->adam_v,
                             h->grads, (float)lr_t, beta1, beta2, eps, gscale, s, nullptr, h->grad_overwrite);
  }
  CHK(launch_adam(h->params, h->adam_m, h->adam_v, h->grads, h->arena_floats, (float)lr_t, beta1, beta2,
                  eps, gscale, s));
  return refresh_all(h, s);
}

int fact_clip_gradients(FactHandle* h, float clip_norm, float* norm_out, void* stream) {
  if (!h) return fail(-1, "null handle");
  if (!h->training) return fail(-1, "handle was created with training=0");
  if (!(clip_norm > 0.f)) return fail(-1, "fact_clip_gradients: clip_norm must be positive");
  hipStream_t s = (hipStream_t)stream;
  // tf.clip_by_global_norm on this replica's own gradient (single_task_trainer.py:180-183), both halves on the device
  HIPCHK(hipMemsetAsync(h->scalars + 9, 0, sizeof(float), s));
  CHK(launch_sumsq(h->grads, h->arena_floats, h->scalars + 9, s));
  return launch_clip_scale(h->grads, h->arena_floats, h->scalars + 9, clip_norm, norm_out, s);
}

int fact_adam_begin(FactHandle* h, float lr, float beta1, float beta2, float eps) {
  if (!h) return fail(-1, "null handle");
  if (!h->training) return fail(-1, "handle was created with training=0");
  h->step += 1;
  const double t = (double)h->step;
  h->adam.lr_t = (float)((double)lr * std::sqrt(1.0 - std::pow((double)beta2, t)) / (1.0 - std::pow((double)beta1, t)));
  h->adam.b1 = beta1;
  h->adam.b2 = beta2;
  h->adam.eps = eps;
  h->adam.gscale = 1.0f;
  h->wg_adam = false;  // decided per fact_forward_backward call (a call that failed half-way must not leave it set)
  // created on first use, and only when the engine itself will enqueue the updates (with a gradient callback the
  // host calls fact_adam_bucket on its communication stream: one stream less competing for the hardware queues)
  if (!h->opt && !h->cb) HIPCHK(make_stream(&h->opt, "FACT_PRIO_OPT"));
  h->adam_pending = true;
  return 0;
}

int fact_adam_bucket(FactHandle* h, int bucket, void* stream) { return fact_adam_bucket_bf16(h, bucket, nullptr, stream); }

int fact_adam_bucket_bf16(FactHandle* h, int bucket, const void* grads_bf16, void* stream) {
  if (!h) return fail(-1, "null handle");
  if (!h->adam_pending) return fail(-1, "fact_adam_bucket without fact_adam_begin");
  if (bucket < 0 || bucket >= (int)h->buckets.size()) return fail(-1, "bucket index out of range");
  if (grads_bf16 && ((uintptr_t)grads_bf16 & 7)) return fail(-1, "bf16 gradient buffer must be 8-byte aligned");
  CHK(adam_bucket(h, bucket, (hipStream_t)stream, (const bf16_t*)grads_bf16));
  if (bucket == (int)h->buckets.size() - 1) h->adam_pending = false;
  return 0;
}

int fact_adam_cancel(FactHandle* h) {
  if (!h) return fail(-1, "null handle");
  if (h->adam_pending) {
    h->adam_pending = false;
    h->step -= 1;
  }
  h->wg_adam = false;
  return 0;
}

int fact_cast_f32_bf16(const float* src, void* dst_bf16, size_t n, void* stream) {
  if (!src || !dst_bf16) return fail(-1, "null argument");
  CHK(launch_cast_bf16(src, (bf16_t*)dst_bf16, n, (hipStream_t)stream));
  return 0;
}
int fact_cast_bf16_f32(const void* src_bf16, float* dst, size_t n, void* stream) {
  if (!src_bf16 || !dst) return fail(-1, "null argument");
  CHK(launch_cast_f32(( const bf16_t*)src_bf16, dst, n, (hipStream_t)stream));
  return 0;
}

int fact_kprof(FactHandle* h, int on) {
  if (!h) return fail(-1, "null handle");
  h->kp.on = on != 0;
  h->kp.recs.clear();
  h->kp.used = 0;
  return 0;
}

/* The kernels behind one class of the table, as the recorder saw them launched (FACT_LAUNCH): one text line per distinct
 * (kernel, grid, block, dynamic LDS), most frequent first:
 *   count \t grid \t block \t lds_bytes \t workgroups_per_cu \t demangled kernel name
 * workgroups_per_cu is the runtime's occupancy answer for that launch shape (how many such workgroups fit one CU); the
 * dispatcher spreads a launch over min(256, grid) CUs.  Names are what rocprofv3 prints for the same dispatches. */
int fact_kprof_kernels(FactHandle* h, int cls, char* buf, int cap) {
  if (!h || !buf || cap <= 0) return fail(-1, "null argument");
  if (cls < 0 || cls >= KP_N) return fail(-1, "no such kernel class");
  struct Agg { KNote n; long count; };
  std::vector<Agg> agg;
  for (const KProf::Rec& r : h->kp.recs) {
    if (r.cls != cls) continue;
    for (const KNote& n : r.notes) {
      bool found = false;
      for (Agg& a : agg)
        if (a.n.fn == n.fn && a.n.grid == n.grid && a.n.block == n.block && a.n.lds == n.lds) { a.count++; found = true; break; }
      if (!found) agg.push_back(Agg{n, 1});
    }
  }
  std::sort(agg.begin(), agg.end(), [](const Agg& a, const Agg& b) { return a.count > b.count; });
  std::string out;
  for (const Agg& a : agg) {
    const char* mangled = hipKernelNameRefByPtr(a.n.fn, nullptr);
    std::string name = mangled ? mangled : "?";
    if (mangled) {
      int st = 0;
      char* d = abi::__cxa_demangle(mangled, nullptr, nullptr, &st);
      if (st == 0 && d) name = d;
      free(d);
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, a.n.fn, (int)a.n.block, a.n.lds) != hipSuccess) per_cu = 0;
    char line[64];
    snprintf(line, sizeof(line), "%ld\t%u\t%u\t%zu\t%d\t", a.count, a.n.grid, a.n.block, a.n.lds, per_cu);
    out += line;
    out += name;
    out += "\n";
  }
  snprintf(buf, (size_t)cap, "%s", out.c_str());
  return 0;
}

int fact_kprof_read(FactHandle* h, int max_classes, int* n_classes, const char** names, double* launches,
                    double* total_ms, double* flops, double* bytes) {
  if (!h || !n_classes) return fail(-1, "null argument");
  HIPCHK(hipDeviceSynchronize());
  double L[KP_N] = {}, T[KP_N] = {}, F[KP_N] = {}, Bt[KP_N] = {};
  for (const KProf::Rec& r : h->kp.recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    L[r.cls] += r.launches; T[r.cls] += ms; F[r.cls] += r.flops; Bt[r.cls] += r.bytes;
  }
  const int n = KP_N < max_classes ? KP_N : max_classes;
  for (int i = 0; i < n; ++i) {
    if (names) names[i] = kKClassName[i];
    if (launches) launches[i] = L[i];
    if (total_ms) total_ms[i] = T[i];
    if (flops) flops[i] = F[i];
    if (bytes) bytes[i] = Bt[i];
  }
  *n_classes = n;
  return 0;
}

/* Timeline dump of the records (CSV: class, stream, start_us, end_us relative to the first record). */
int fact_kprof_dump(FactHandle* h, const char* path) {
  if (!h || !path) return fail(-1, "null argument");
  HIPCHK(hipDeviceSynchronize());
  FILE* f = fopen(path, "w");
  if (!f) return fail(-1, "cannot open dump file");
  if (!h->kp.recs.empty()) {
    hipEvent_t ref = h->kp.recs[0].a;
    for (const KProf::Rec& r : h->kp.recs) {
      float a = 0.f, b = 0.f;
      if (hipEventElapsedTime(&a, ref, r.a) != hipSuccess || hipEventElapsedTime(&b, ref, r.b) != hipSuccess) continue;
      fprintf(f, "%s,%d,%.1f,%.1f\n", kKClassName[r.cls], r.sid, a * 1e3, b * 1e3);
    }
  }
  fclose(f);
  return 0;
}

int fact_num_buckets(FactHandle* h, int* n) {
  if (!h || !n) return fail(-1, "null argument");
  *n = (int)h->buckets.size();
  return 0;
}

int fact_get_step(FactHandle* h, int64_t* step) {
  if (!h || !step) return fail(-1, "null argument");
  *step = h->step;
  return 0;
}
int fact_set_step(FactHandle* h, int64_t step) {
  if (!h) return fail(-1, "null handle");
  h->step = step;
  return 0;
}

int fact_infer_ar(FactHandle* h, const float* motion_seed, const float* audio, int B, int audio_len,
                  int steps, float* out, int* steps_done, void* stream) {
  int rc = check_batch(h, B);
  if (rc) return rc;
  Stack &mo = h->motion, &au = h->audio, &cr = h->cross;
  if (h->cfg.out_dim != mo.feat) return fail(-1, "auto-regressive inference needs out_dim == motion feature_dim");
  if (steps < 0) return fail(-1, "steps must be >= 0");
  hipStream_t s = (hipStream_t)stream;
  // fact_model.py:124-126: stop when the audio window runs short
  int nsteps = audio_len - au.n + 1;
  if (nsteps > steps) nsteps = steps;
  if (nsteps < 0) nsteps = 0;
  if (steps_done) *steps_done = nsteps;
  if (nsteps == 0) return 0;
  const int F = mo.feat;
  const size_t ext = (size_t)(mo.n + nsteps) * F;  // per-sample floats of the extended motion track
  if (h->ar_motion_floats < ext * B) {
    (void)hipFree(h->ar_motion);
    h->ar_motion = nullptr;
    HIPCHK(hipMalloc((void**)&h->ar_motion, ext * B * sizeof(float)));
    h->ar_motion_floats = ext * B;
  }
  CHK(copy2d(h->ar_motion, ext * sizeof(float), motion_seed, (size_t)mo.n * F * sizeof(float),
             (size_t)mo.n * F * sizeof(float), B, s));
  // output[:, 0:1, :] is all that survives a step (fact_model.py:128): like the supervised rows of a train step, the
  // last cross-modal layer runs its attention queries, to_out, LayerNorm 2 and the MLP on that ONE row per sequence
  // (keys / values still come from all rows), and the head reads the compact rows.
  const bool sr = h->sr_rows && cr.L >= 1 && cr.n >= 4 && rups((size_t)B, 64) <= (size_t)h->sr.rows_max;
  struct SkinnyScope {  // split-K GEMMs for the whole model while this call runs at a batch of <= 512 rows per stack
    FactHandle* h;
    explicit SkinnyScope(FactHandle* hh, bool on) : h(hh) { h->skinny_fwd = on; }
    ~SkinnyScope() { h->skinny_fwd = false; }
  } skinny_scope(h, h->sr_rows != 0);
  struct InferScope {
    FactHandle* h;
    explicit InferScope(FactHandle* hh) : h(hh) { h->keep_pre = h->keep_pre_infer; }  // debug option "keep_pre"
    ~InferScope() { h->keep_pre = true; }
  } infer_scope(h);
  for (int i = 0; i < nsteps; ++i) {
    // motion window = frames [i, i+n_m) of the extended track; audio window = frames [i, i+n_a)
    CHK(model_forward_hidden(h, h->ar_motion + (size_t)i * F, ext, audio + (size_t)i * au.feat,
                             (size_t)audio_len * au.feat, B, s, sr ? 1 : 0));
    const bf16_t* x16 = h->ar_x16;
    int ldx = cr.d;
    if (sr) {
      CHK(launch_pad_cast(h->sr.x_out_c, B, 0, B, cr.d, h->sr.xf16_c, cr.dp, s));
      x16 = h->sr.xf16_c;
      ldx = cr.dp;
    } else {  // head on token 0 of every sample
      CHK(launch_pad_cast(cr.out(), 1, (size_t)cr.n * cr.d, B, cr.d, h->ar_x16, cr.d, s));
    }
    GemmParams g = gp(x16, ldx, h->head.t, h->head.ldt, B, h->cfg.out_dim, cr.d);
    with_skinny_fwd(h, g, s);
    g.ep.out0 = h->ar_motion + (size_t)(mo.n + i) * F; g.ep.ldo0 = (int)ext; g.ep.bias = P(h, h->head_b);
    CHK(launch_gemm_nt(EPI_F32_BIAS, g, s));
  }
  CHK(copy2d(out, (size_t)steps * F * sizeof(float), h->ar_motion + (size_t)mo.n * F, ext * sizeof(float),
             (size_t)nsteps * F * sizeof(float), B, s));
  return 0;
}

// ---------------------------------------------------------------------------------------------
// single-op entry points
// ---------------------------------------------------------------------------------------------
int fact_op_gemm_nt(int epi, const void* A, int lda, const void* B, int ldb, int M, int N, int K,
                    void* out0, int ldo0, void* out1, int ldo1, const float* bias, const float* pos,
                    int seq, const float* resid, int ldr, const void* pre, int ldp, void* stream) {
  if (epi == EPI_HEADS) return fail(-1, "use fact_op_attention for the heads epilogue");
  GemmParams g = gp((const bf16_t*)A, lda, (const bf16_t*)B, ldb, M, N, K);
  g.ep.out0 = out0; g.ep.ldo0 = ldo0; g.ep.out1 = out1; g.ep.ldo1 = ldo1;
  g.ep.bias = bias; g.ep.pos = pos; g.ep.seq = seq; g.ep.resid = resid; g.ep.ldr = ldr;
  g.ep.pre = (const bf16_t*)pre; g.ep.ldp = ldp;
  if (epi == EPI_ATOMIC_F32 && seq > 1) g.splitk = seq;  // bench hook: `seq` carries the K split
  {  // split-K workspace of the op-level entry (one stream at a time)
    static float* slab = nullptr;
    static unsigned* cnt = nullptr;
    if (!slab) {
      if (hipMalloc((void**)&slab, kSplitKSlabBytes) != hipSuccess ||
          hipMalloc((void**)&cnt, kSplitKCounters * sizeof(unsigned)) != hipSuccess)
        return fail(-20, "split-K workspace alloc");
      (void)hipMemset(cnt, 0, kSplitKCounters * sizeof(unsigned));
    }
    g.sk_slab = slab;
    g.sk_cnt = cnt;
  }
  CHK(launch_gemm_nt(epi, g, (hipStream_t)stream));
  return 0;
}

int fact_op_gemm_tn(const void* A, int lda, const void* B, int ldb, int Mo, int No, int K,
                    float* out, int ldo, int splitk, int use_tr, void* scratch, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (use_tr == 2) {  // slab mode: scratch >= splitk * round4(Mo*No) floats
    if (!scratch || ldo != No) return fail(-1, "slab mode needs scratch and ldo == No");
    GemmParams p = gp((const bf16_t*)A, lda, (const bf16_t*)B, ldb, Mo, No, K);
    const size_t stride = rups((size_t)Mo * No, 4);
    p.splitk = splitk; p.ep.out0 = scratch; p.ep.ldo0 = No; p.ep.slab_stride = stride;
    CHK(launch_gemm_tn(EPI_F32_SLAB, p, s));
    CHK(launch_slab_reduce((const float*)scratch, stride, splitk, out, (size_t)Mo * No, s));
    return 0;
  }
  if (use_tr) {
    GemmParams p = gp((const bf16_t*)A, lda, (const bf16_t*)B, ldb, Mo, No, K);
    p.splitk = splitk; p.ep.out0 = out; p.ep.ldo0 = ldo;
    CHK(launch_gemm_tn(EPI_ATOMIC_F32, p, s));
    return 0;
  }
  if (!scratch) return fail(-1, "scratch required for the transpose path");
  const int ldk = rup(K, 8);
  bf16_t* tA = (bf16_t*)scratch;
  bf16_t* tB = tA + (size_t)rup(Mo, 8) * ldk;
  HIPCHK(hipMemsetAsync(scratch, 0, ((size_t)rup(Mo, 8) + rup(No, 8)) * ldk * sizeof(bf16_t), s));
  CHK(launch_transpose_bf16((const bf16_t*)A, lda, K, Mo, tA, ldk, s));
  CHK(launch_transpose_bf16((const bf16_t*)B, ldb, K, No, tB, ldk, s));
  GemmParams p = gp(tA, ldk, tB, ldk, Mo, No, ldk);
  p.splitk = splitk; p.ep.out0 = out; p.ep.ldo0 = ldo;
  CHK(launch_gemm_nt(EPI_ATOMIC_F32, p, s));
  return 0;
}

/* Test/bench driver of the grouped whole-K TN kernel (gemm_big.hip): n problems out_i (+)= A_i^T B_i over a
 * shared K; arrays of n entries; trans[i] = 1 stores out_i as [N][M]. */
int fact_op_gemm_tn_group(int n, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                          float* const* out, const int* ldo, const int* Mo, const int* No, const int* trans, int K,
                          void* stream) {
  if (n < 1 || n > TN_GROUP_MAX) return fail(-1, "1..4 problems");
  TnGroup g;
  memset(&g, 0, sizeof(g));
  g.n = n;
  g.K = K;
  for (int i = 0; i < n; ++i) {
    TnProblem& q = g.p[i];
    q.A = (const bf16_t*)A[i]; q.lda = lda[i]; q.B = (const bf16_t*)B[i]; q.ldb = ldb[i];
    q.out = out[i]; q.ldo = ldo[i]; q.M = Mo[i]; q.N = No[i]; q.trans_out = trans[i];
  }
  CHK(launch_big_tn_group(g, (hipStream_t)stream, g_op_tn_parts));
  return 0;
}
int fact_op_gemm_tn_group_cs(int n, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                             float* const* out, const int* ldo, const int* Mo, const int* No, const int* trans,
                             float* const* csum, int K, void* stream) {
  if (n < 1 || n > TN_GROUP_MAX) return fail(-1, "1..4 problems");
  if (!big_tn_group_has_colsum()) return fail(-1, "the selected wgrad main loop has no operand column sums (tn_loop = 0 only)");
  TnGroup g;
  memset(&g, 0, sizeof(g));
  g.n = n;
  g.K = K;
  for (int i = 0; i < n; ++i) {
    TnProblem& q = g.p[i];
    q.A = (const bf16_t*)A[i]; q.lda = lda[i]; q.B = (const bf16_t*)B[i]; q.ldb = ldb[i];
    q.out = out[i]; q.ldo = ldo[i]; q.M = Mo[i]; q.N = No[i]; q.trans_out = trans[i]; q.csum = csum[i];
  }
  CHK(launch_big_tn_group(g, (hipStream_t)stream, g_op_tn_parts));
  return 0;
}
int fact_op_gemm_tn_group_adam(int n, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                               float* const* p, float* const* m, float* const* v, void* const* sd, const int* lds,
                               void* const* st, const int* ldt, const int* Mo, const int* No, const int* trans, int K,
                               float lr_t, float beta1, float beta2, float eps, void* stream) {
  if (n < 1 || n > TN_GROUP_MAX) return fail(-1, "1..4 problems");
  TnGroup g;
  memset(&g, 0, sizeof(g));
  g.n = n;
  g.K = K;
  g.overwrite = 1;
  g.adam = 1; g.lr_t = lr_t; g.b1 = beta1; g.b2 = beta2; g.eps = eps;
  for (int i = 0; i < n; ++i) {
    TnProblem& q = g.p[i];
    q.A = (const bf16_t*)A[i]; q.lda = lda[i]; q.B = (const bf16_t*)B[i]; q.ldb = ldb[i];
    q.M = Mo[i]; q.N = No[i]; q.trans_out = trans[i];
    q.out = p[i]; q.ldo = trans[i] ? Mo[i] : No[i];  // (out itself is not written; its pitch indexes p / m / v)
    q.p = p[i]; q.m1 = m[i]; q.v = v[i]; q.sd = (bf16_t*)sd[i]; q.ldsd = lds[i]; q.st = (bf16_t*)st[i]; q.ldst = ldt[i];
  }
  CHK(launch_big_tn_group(g, (hipStream_t)stream, g_op_tn_parts));
  return 0;
}
int fact_debug_gemm_tn_cfg(int v) {  // low byte: main loop (0 staggered, 1 / 2 fragment prefetch); bits 8..: launches per group of the op
  gemm_set_tn_cfg(v & 0xff);
  g_op_tn_parts = (v >> 8) > 0 ? (v >> 8) : 1;
  return 0;
}
int fact_debug_gemm_splitk_max(int v) {
  gemm_set_splitk_max(v);
  return 0;
}
int fact_debug_gemm_big_impl(int v) {
  gemm_set_big_impl(v);
  return 0;
}

int fact_op_ln_fwd(const float* x, const float* gamma, const float* beta, void* hh, float* mean,
                   float* rstd, int M, int C, float eps, void* stream) {
  CHK(launch_ln_fwd(x, gamma, beta, (bf16_t*)hh, C, mean, rstd, M, C, eps, (hipStream_t)stream));
  return 0;
}

int fact_op_ln_bwd(const void* dh, const float* x, const float* mean, const float* rstd,
                   const float* gamma, const float* dres, float* dx, void* dx_bf16, float* dgamma,
                   float* dbeta, float* dbias_prev, int M, int C, void* stream) {
  float* ws = nullptr;
  if (g_op_ln_ws == 1) {  // bench knob: the round-1 fused kernel with the partial-sum path (workspace owned by the op)
    static float* buf = nullptr;
    static size_t cap = 0;
    const size_t need = ln_bwd_ws_floats(M, C);
    if (need > cap) {
      if (buf) (void)hipFree(buf);
      if (hipMalloc(&buf, need * sizeof(float)) != hipSuccess) return fail(-20, "ln ws alloc");
      cap = need;
    }
    ws = buf;
  }
  if ((g_op_ln_ws == 3 || g_op_ln_ws == 4) && C <= 1024 && dgamma && dbeta) {
    // round-4 engine form: the dx kernel leaves per-workgroup column-sum partials, a small reduce adds them up
    static float* pbuf = nullptr;
    static size_t pcap = 0;
    const size_t need = ln_cs_part_floats(M, C);
    if (need > pcap) {
      if (pbuf) (void)hipFree(pbuf);
      if (hipMalloc(&pbuf, need * sizeof(float)) != hipSuccess) return fail(-20, "ln partials alloc");
      pcap = need;
    }
    ln_set_cs_rows(g_op_ln_ws == 4 ? 2 : 4);
    CHK(launch_ln_bwd_dx_cs((const bf16_t*)dh, x, mean, rstd, gamma, dres, dx, (bf16_t*)dx_bf16, pbuf, M, C, C,
                            (hipStream_t)stream));
    CHK(launch_colreduce(pbuf, ln_cs_blocks(M), C, dgamma, dbeta, (dres ? dbias_prev : nullptr), (hipStream_t)stream));
    ln_set_cs_rows(4);
    return 0;
  }
  if (g_op_ln_ws == 5) {  // bench: the row-wise dx kernel alone
    CHK(launch_ln_bwd_dx((const bf16_t*)dh, x, mean, rstd, gamma, dres, dx, (bf16_t*)dx_bf16, M, C, C, (hipStream_t)stream));
    return 0;
  }
  if ((g_op_ln_ws == 0 || g_op_ln_ws >= 3) && dgamma && dbeta) {
    // the round-2/3 engine form: parameter gradients (here from the fp32 residual gradient, before dx may overwrite it
    // in place), then the row-wise dx kernel
    CHK(launch_ln_param_grads((const bf16_t*)dh, C, x, mean, rstd, (dbias_prev ? dres : nullptr), C, 1, dgamma, dbeta,
                              dbias_prev, M, C, (hipStream_t)stream));
    CHK(launch_ln_bwd_dx((const bf16_t*)dh, x, mean, rstd, gamma, dres, dx, (bf16_t*)dx_bf16, M, C, C,
                         (hipStream_t)stream));
    return 0;
  }
  CHK(launch_ln_bwd((const bf16_t*)dh, x, mean, rstd, gamma, dres, dx, (bf16_t*)dx_bf16, dgamma, dbeta,
                    dbias_prev, ws, M, C, C, (hipStream_t)stream));
  return 0;
}
int fact_debug_ln_bwd(int rows_per_block, int use_ws) {
  if (rows_per_block >= 4 && (rows_per_block & 3) == 0) ln_set_bwd_rows(rows_per_block);
  g_op_ln_ws = use_ws;
  return 0;
}

namespace {
struct AttnScratch {
  size_t row[4], lse, dsum, total;
};
AttnScratch attn_scratch_layout(int B, int H, int n, int dh) {
  AttnScratch a;
  const size_t BH = (size_t)B * H, NP = rup(n, 128), dhp = rup(dh, 32);
  size_t off = 0;
  for (int i = 0; i < 4; ++i) { a.row[i] = off; off += rups(BH * NP * dhp * 2, 256); }
  a.lse = off; off += rups(BH * NP * 4, 256);
  a.dsum = off; off += rups(BH * NP * 4, 256);
  a.total = off;
  return a;
}
}  // namespace

size_t fact_op_attention_scratch(int B, int H, int n, int dh) {
  return attn_scratch_layout(B, H, n, dh).total;
}

// Test driver for the attention kernels: splits a packed (qkv h d) bf16 tensor into the per-head
// operand buffers with the same EPI_HEADS epilogue the model uses (GEMM against an identity).
int fact_op_attention(const void* qkv, int B, int H, int n, int dh, float scale, void* out,
                      const void* dout, void* dqkv, void* scratch, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int hid = H * dh, M = B * n;
  if (n % 8) return fail(-1, "n must be a multiple of 8");
  AttnScratch L = attn_scratch_layout(B, H, n, dh);
  char* sc = (char*)scratch;
  HIPCHK(hipMemsetAsync(sc, 0, L.total, s));
  Stack st;
  st.n = n; st.d = hid; st.H = H; st.dh = dh; st.dhp = rup(dh, 32); st.NP = rup(n, 128);
  // identity [3*hid][3*hid] bf16 as the B operand
  bf16_t* eye = nullptr;
  const int W = 3 * hid;
  HIPCHK(hipMalloc((void**)&eye, (size_t)W * W * sizeof(bf16_t)));
  {
    std::vector<uint16_t> he((size_t)W * W, 0);
    for (int i = 0; i < W; ++i) he[(size_t)i * W + i] = 0x3F80;  // bf16 1.0
    HIPCHK(hipMemcpyAsync(eye, he.data(), he.size() * 2, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  bf16_t* row[4];
  for (int i = 0; i < 4; ++i) row[i] = (bf16_t*)(sc + L.row[i]);
  {
    GemmParams g = gp((const bf16_t*)qkv, W, eye, W, M, W, W);
    heads_ep(g.ep, st, row, 3);
    int rc = launch_gemm_nt(EPI_HEADS, g, s);
    if (rc) { (void)hipFree(eye); return fail(rc, "heads gemm failed"); }
  }
  AttnParams ap;
  memset(&ap, 0, sizeof(ap));
  ap.qrow = row[0]; ap.krow = row[1]; ap.vrow = row[2];
  ap.out = (bf16_t*)out; ap.o = (const bf16_t*)out; ap.lse2 = (float*)(sc + L.lse);
  ap.B = B; ap.H = H; ap.n = n; ap.NP = st.NP; ap.hid = hid; ap.dh = dh; ap.scale = scale;
  ap.ldo = hid; ap.ldq = 3 * hid;
  int rc = launch_attn_fwd(ap, s);
  if (rc == 0 && dout) {
    GemmParams g = gp((const bf16_t*)dout, hid, eye, W, M, hid, hid);
    bf16_t* r1[1] = {row[3]};
    heads_ep(g.ep, st, r1, 1);
    rc = launch_gemm_nt(EPI_HEADS, g, s);
    if (rc == 0) {
      ap.dorow = row[3]; ap.dsum = (float*)(sc + L.dsum); ap.dqkv = (bf16_t*)dqkv;
      rc = launch_attn_bwd(ap, s);
    }
  }
  (void)hipStreamSynchronize(s);
  (void)hipFree(eye);
  if (rc) return fail(rc, "attention launch failed");
  return 0;
}

int fact_debug_force_generic_gemm(int on) {
  g_force_generic_gemm = on;
  return 0;
}
int fact_debug_attn_force_tiled(int on) {
  attn_set_force_tiled(on);
  return 0;
}
// Occupies `nwg` CUs for ~`micros` microseconds: one 256-thread workgroup per CU (96 KiB of LDS each keeps a second one
// off the CU), spinning on the shader clock.  Stand-in for a communication kernel that holds CUs while the step runs
// (tools/attic/cu_hog_probe.py: what does the train step lose when N CUs are not available to it?).
static __global__ __launch_bounds__(256) void cu_hog_kernel(long long cycles, unsigned* sink, int mode) {
  extern __shared__ unsigned char hog_lds[];
  unsigned acc = 0;
  if (mode & 2) {  // no clock polling: a counted sleep loop (~64 * 64 cycles per trip at the shader clock)
    for (long long i = 0; i < cycles / 2; ++i) {
      __builtin_amdgcn_s_sleep(64);
      asm volatile("" : "+v"(acc));
    }
  } else {
    const long long t0 = (long long)wall_clock64();  // constant 100 MHz counter (s_memrealtime)
    while ((long long)wall_clock64() - t0 < cycles) {
      if (!(mode & 1)) acc += hog_lds[(threadIdx.x * 64) & 1023];
      __builtin_amdgcn_s_sleep(32);
    }
  }
  if (acc == 0xFFFFFFFFu) *sink = acc;
}
// nwg: low 16 bits = workgroups; bit 16 = no LDS allocation (co-resident with anything), bit 17 = no clock polling
int fact_debug_cu_hog(int nwg, int micros, void* stream) {
  static bool once = false;
  if (!once) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(cu_hog_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               96 * 1024));
    once = true;
  }
  static unsigned* sink = nullptr;
  if (!sink) HIPCHK(hipMalloc((void**)&sink, 4));
  const int mode = nwg >> 16, n = nwg & 0xFFFF;
  if (n <= 0) return 0;
  FACT_LAUNCH(cu_hog_kernel, dim3(n), dim3(256), (mode & 1) ? 1024 : 96 * 1024, (hipStream_t)stream,
                     (long long)micros * 100, sink, mode);  // wall_clock64() ticks at 100 MHz
  return 0;
}
int fact_debug_attn_variant(int v) {
  attn_set_variant(v);
  return 0;
}
int fact_debug_attn_variant_get(void) { return attn_get_variant(); }
int fact_debug_gemm_nt_variant(int v) {
  gemm_set_nt_variant(v);
  return 0;
}
int fact_debug_gemm_nt_band(int band) {
  gemm_set_nt_band(band);
  return 0;
}

int fact_op_adam(float* p, float* m, float* v, float* g, size_t n, float lr_t, float b1, float b2,
                 float eps, void* stream) {
  CHK(launch_adam(p, m, v, g, n, lr_t, b1, b2, eps, 1.0f, (hipStream_t)stream));
  return 0;
}

int fact_loss(const float* target, const float* pred, int B, int n, int T, int D, float* loss_out, void* stream) {
  if (!target || !pred || !loss_out) return fail(-1, "null argument");
  if (B <= 0 || n <= 0 || T <= 0 || T > n || D <= 0) return fail(-1, "loss: need 0 < T <= n");
  HIPCHK(hipMemsetAsync(loss_out, 0, sizeof(float), (hipStream_t)stream));
  CHK(launch_mse_loss(pred, target, loss_out, nullptr, B, n, T, D, rup(D, 32), 1.0f, (hipStream_t)stream));
  return 0;
}

int fact_op_mse(const float* pred, const float* target, float* loss, void* dpred, int B, int n, int T,
                int D, int ldp, float gscale, void* stream) {
  CHK(launch_mse_loss(pred, target, loss, (bf16_t*)dpred, B, n, T, D, ldp, gscale, (hipStream_t)stream));
  return 0;
}

}  // extern "C"
