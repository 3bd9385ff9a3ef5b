// fused.hip -- the fused Llama decode step (crabml_hip_llama_*): the hot path as 5 kernels per layer, replayed from one
// hipGraph with token id / position resident in device memory, plus the batched prefill.  This file is the HOST side
// (context, segment enqueue, graph capture, C entry points); the kernels live in fused_common.hpp (quantizer lanes,
// norm, q/k/v), fused_attention.hpp and fused_ffn.hpp (wo / ffn_down with the norm epilogue, gate/up, sampler).
//
// It serves exactly the op sequence Llama2Runner<T> issues for one token (crabml-llama2/src/llama2.rs):
//   forward_llama :213-281, forward_multi_query_attention :527-603, forward_ffn :605-638, classifier :184-211
// with the reference's arithmetic (each fused stage cites the primitive it folds in).  Why fuse: the per-op
// trait path is launch-bound (31 launches/layer, GPU busy 1/3 of the time; profiles/r01_trait_path_kernel_
// trace.md).  The GEMV stages use the same lane-per-block / R-rows-per-wave mapping as gemv.hip and stay
// HBM-bound; the small stages are folded into their producers/consumers so activations never round-trip
// through extra launches:
//   k_norm_quant   rms_norm_inplace + mul_inplace(weight) + quantize_f32_q8_0        (1 workgroup)
//   k_qkv          wq/wk/wv matmul_vec + rope_inplace(q,k) + scale_inplace(q) + concatenate(k,v -> KV cache)
//   k_attn         batch_matmul(q,K^T) + softmax_inplace + batch_matmul(p,V) [+ quantize for wo]
//   k_gemv_res     wo / ffn_down matmul_vec + add_inplace(residual)
//   k_gateup       ffn_gate/ffn_up matmul_vec + silu_inplace + mul_inplace
//   k_argmax_step  greedy sampler (last maximum) + token/position advance
#include <chrono>
#include <thread>
#include <cmath>

#include "fused_common.hpp"
#include "fused_attention.hpp"
#include "fused_ffn.hpp"
#include "prefill_rows.hpp"
#include "lazy.hpp"


// ==============================================================================================================
// Host side: the decode step as a list of segments.  With tensor parallelism (tp_size > 1) every segment ends
// in a partial-sum vector that is all-reduced across ranks (RCCL over xGMI; 2 x dim f32 per layer):
//   segment 2l   : [embed] attn-norm(+ pending residual) -> qkv(local heads) -> attention -> wo(local k-slice)
//   segment 2l+1 : ffn-norm(+ pending residual) -> gate/up(local rows) -> down(local k-slice)
//   segment 2L   : final norm(+ pending residual) -> classifier -> greedy argmax / advance
// Column-parallel: wq/wk/wv by heads, gate/up by rows.  Row-parallel: wo, ffn_down by k.
// ==============================================================================================================
#include <dlfcn.h>

using namespace crabml_hip;

// ---- RCCL, bound at run time (the single-GPU product path never needs it) ------------------------------------
struct crabml_hip_tp_comm {
  crabml_hip_device* dev = nullptr;
  void* nccl = nullptr;  // ncclComm_t (RCCL kind)
  int nranks = 1, rank = 0;
  // P2P kind (crabml_hip_tp_p2p_*): the one-shot all-reduce over peer-mapped inboxes (fused_ffn.hpp, TpP2P)
  bool p2p = false;
  bool connected = false;
  unsigned long long* inbox = nullptr;   // this rank's inbox: TP_SLOTS slots x nranks rows x cap granules
  bool via_ipc[8] = {false};             // peer[r] was mapped with hipIpcOpenMemHandle (closed at destroy; plain pointers are not)
  size_t inbox_bytes = 0;
  unsigned cap = 0;                      // granules per row
  bool finegrained = false;
  void* peer[8] = {nullptr};             // peers' inboxes as mapped here (peer[rank] = inbox)
  int* fault = nullptr;                  // device word raised by a poll that timed out
  unsigned host_epoch = 0;               // crabml_hip_tp_all_reduce outside a decode step
  unsigned sessions = 0;                 // decode contexts created on this group so far (every rank creates them in the same order)
};
namespace {
struct NcclId {  // ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
  char b[128];
};
struct Rccl {
  typedef NcclId IdT;
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (r.lib) {
      r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
      r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
      r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
      r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
      r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    }
  }
  return (r.lib && r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce) ? &r : nullptr;
}
}  // namespace

struct crabml_hip_llama {
  crabml_hip_device* dev = nullptr;
  crabml_hip_llama_config_t cfg{};
  uint32_t wtype = 0;
  int tp = 1, tp_rank = 0;
  crabml_hip_tp_comm* comm = nullptr;
  // local (per-rank) geometry
  int hd = 0, npairs = 0, n_heads_l = 0, n_kv_l = 0, dim_l = 0, kv_dim_l = 0, hidden_l = 0;
  std::vector<crabml_hip_buf*> held;  // retained weight buffers
  crabml_hip_buf* token_embed = nullptr;
  crabml_hip_buf* rms_final = nullptr;
  crabml_hip_buf* output = nullptr;
  std::vector<crabml_hip_buf*> rms_att, rms_ffn, wq, wk, wv, wo, gate, down, up;
  // device state
  std::vector<void*> kc, vc;
  size_t kv_bytes = 0;
  float* x = nullptr;        // residual stream (dim), replicated on every rank
  float* partial = nullptr;  // tp > 1: this rank's wo / ffn_down partial sums (dim), all-reduced in place
  float* qbuf = nullptr;     // roped, scaled q (dim_l)
  float* attn = nullptr;     // attention output (dim_l)
  float* h = nullptr;        // ffn hidden (hidden_l), strict mode only
  float* logits = nullptr;   // vocab
  float* host_logits = nullptr; // lazy.hip: pinned host copy of the logits, written by a kernel behind the classifier; the two words
                                // behind the vocab_size floats are {sequence number of the step that wrote them, its fault word}
  unsigned out_seq = 0;         // sequence number of the last step whose logits were sent to host_logits
  unsigned lazy_serial = 0;     // lazy.hip: the step serial is set by the host at every begin (see lazy_ctx_begin)
  bool ext_kv = false;          // lazy.hip: kc / vc are the runner's own cache buffers (retained in `held`), not allocations of ours
  const crabml_hip_buf* ext_kc0 = nullptr;  // ... the first layer's K cache handle (lazy_ctx_orphaned)
  float* tmp = nullptr;      // strict-mode GEMV outputs
  char* act_dim = nullptr;   // Q8_0 planes of the normalized residual (dim)
  char* act_attn = nullptr;  // Q8_0 planes of the attention output (dim_l)
  char* act_hid = nullptr;   // Q8_0 planes of the ffn hidden vector (hidden_l)
  float* rope = nullptr;     // [seq_len][npairs][2]
  int* state = nullptr;      // token, pos, step, sink, serial (never reset), fault
  unsigned long long* slots = nullptr;  // dim/32 {chunk sum, epoch} granules of the norm epilogue
  unsigned long long* a8gran = nullptr;  // Q4_K layers: granules of the attention output (dim_l) and of h (hidden_l), through which
  unsigned long long* h8gran = nullptr;  // the producing kernels assemble Q8_K super-blocks (q8k_exchange_store)
  bool q8k_producers = false;            // attention / gate-up emit the Q8_K planes of wo's / ffn_down's rhs themselves
  unsigned tp_salt = 0;      // P2P group: epoch salt of this context (see TpP2P::salt)
  bool tp_dry = false;       // CRABML_HIP_LLAMA_TP_DRY_RUN: a lone rank that skips the all-reduces (timing only)
  bool k_norm_in = false;    // fast Q4_K step: gate | up normalizes and quantizes wo's f32 row itself (k_gateup_k_lds<.., NORMIN>)
  bool kfused = false;       // Q4_K layers, fast mode: fused GEMV kernels with the Q4_K inner loop (enqueue_segment_k)
  bool generic = false;      // per-op launches (strict-order device, or a weight format without fused kernels)
  bool ord = false;          // strict-order device, Q4_0 / Q8_0 / Q4_1 layers: the fused launches with block-ordered sums (k_*_ord)
  uint32_t qt = 0, out_qt = 0;  // vec_dot_rhs_dtype of the layer weights / of the classifier
  float* xn = nullptr;       // generic path: normalized residual (f32, dim)
  bool norm_epi = false;     // fast mode, tp == 1: RMSNorm + quantize run in the wo / ffn_down epilogue
  // the hop-free norm of the fast step (Q4_0 / Q8_0 layers, one GPU): wo quantizes x * w_norm block by block and leaves 1 / rms to
  // the gate/up launch (RmsTail, gemv_core.hpp) -- no in-launch gather.  Off: CRABML_HIP_LLAMA_EXACT_NORM, strict order, tp.
  bool defer_norm = false;
  float* rsums = nullptr;    // [dim / 16] chunk sums of squares of the residual stream
  bool norm_epi_k = false;   // the same for Q4_K layers (Q8_K planes out of the epilogue)
  unsigned* out_tokens = nullptr;
  int out_cap = 0;
  float* am_val = nullptr;  // argmax partials
  int* am_idx = nullptr;
  // CRABML_HIP_LLAMA_TP_SPLIT_VOCAB: this rank's classifier rows [vocab_off, vocab_off + vocab_l) (otherwise 0 / vocab_size)
  bool split_vocab = false;
  int vocab_l = 0, vocab_off = 0;
  int* am_best = nullptr;   // {max bits, index} of this rank's shard (single-device simulation: combined by the driver)
  size_t kv_len = 0;
  // [0]: one attention workgroup per head; [1]: the long-context attention kernels (from attn_long_from positions)
  hipGraph_t graph[3] = {nullptr, nullptr, nullptr};
  hipGraphExec_t exec[3] = {nullptr, nullptr, nullptr};
  bool use_graph = false;
  bool capturing = false;
  int attn_variant = 0;         // which of them the next enqueue emits: 0 = one workgroup per head, 1 = the long-context kernels,
                                // 2 = split-KV attention with the merge inside the launch (ticket form: the mid range)
  size_t flash_ticket_until = 0;  // > 0: positions [attn_long_from, this) run variant 2
  bool attn_long_ok = false;    // f16 cache, head_dim % 32 == 0, group size in {1, 2, 4, 8}; and seq_len % 8 == 0 (exact kernels) or k_attn_flash
  bool exact_long_ok = false;   // the exact long-context kernels (score / probability rows of seq_len elements read as 16-byte vectors)
  size_t attn_long_from = 0;    // cached positions (pos + 1) from which variant 1 is used
  bool pv_split = false;        // variant 1: k_attn_pv_split (products by producer waves) instead of k_attn_pv
  // variant 1 of the FAST step: k_attn_flash (split-KV, f32 accumulation) instead of the three exact kernels
  bool attn_flash = false;
  bool flash_ticket = false;    // A/B: the merge by the last-arriving workgroup inside k_attn_flash instead of its own launch
  bool attn_flash_rows = false; // the batched prefill's attention runs k_attn_flash_rows (fast step, f16 cache, head_dim 64 / 128)
  int flash_S = 0;              // position slices (workgroups) per kv head
  int flash_min_rows = FLASH_MIN_ROWS;  // cached rows per active slice, at least
  float* flash_part = nullptr;  // [n_kv_l][flash_S][G][hd + 2] partial {O, m, l}
  unsigned* flash_tick = nullptr;  // [n_kv_l] arrival counters (monotonic)
  int gu_rows = 0;              // > 0 (tensor-parallel ranks): gate/up leaves h as f32 from workgroups of this many rows, ffn_down quantizes it
  int attn_s_rows = 0;          // > 0: variant 0 runs k_attn_s (K / V staged through LDS) with room for this many cached rows
  size_t attn_s_lds = 0;
  float* scores_g = nullptr;    // [n_heads_l][seq_len] f32
  unsigned short* p16 = nullptr;  // [n_heads_l][seq_len] f16 probabilities
  // batched prefill (crabml_hip_llama_prefill): row buffers for pf_cap prompt rows, allocated on first use
  size_t pf_cap = 0;
  int* pf_tokens = nullptr;
  float *pf_x = nullptr, *pf_xn = nullptr, *pf_q = nullptr, *pf_k = nullptr, *pf_v = nullptr, *pf_qr = nullptr, *pf_attn = nullptr,
        *pf_tmp = nullptr, *pf_g = nullptr, *pf_u = nullptr;
  char *pf_act_dim = nullptr, *pf_act_hid = nullptr;
  float* pf_split = nullptr;  // ... and pf_split_floats of scratch for the partial tiles of its k pieces
  size_t pf_split_floats = 0;
  void* pf_xh2 = nullptr;  // a second one: the gate | up launch reads pf_xh while its epilogue writes ffn_down's
  void* pf_xh = nullptr;  // the fast pass's f16 GEMMs: the current rhs rows as pre-scaled f16 (gemm_f16w.hip), gemm_f16w_xh_bytes(cap, max(dim, hidden))
  float* pf_scores = nullptr;          // long prompts: [PF_LONG_ROWS][n_heads][seq_len] f32 scores
  unsigned short* pf_p16 = nullptr;    //               and f16 probabilities, allocated on first use
  std::vector<std::pair<void*, size_t>> allocs;
  // token / pos / step of the next step are staged in pinned host memory owned by the context (a ring, one slot per
  // set_state): the async copy reads it when the stream gets there, long after the caller's stack frame is gone
  int* h_state = nullptr;
  unsigned h_state_next = 0;
  static constexpr unsigned H_STATE_SLOTS = 256;
};

// ---- lazy.hip's context: token / position / serial of a step straight from kernel arguments (a launch on the stream's own queue:
// no copy-engine hand-off in front of the step's first kernel), and the logits to pinned host memory by a kernel behind the
// classifier, followed by a flag the host can spin on (no copy-engine hand-off, no interrupt-driven wait behind the step's last)
__global__ void k_set_state5(int* __restrict__ st, int token, int pos, int step, int serial, int out_seq) {
  st[0] = token;
  st[1] = pos;
  st[2] = step;
  st[4] = serial;
  st[7] = out_seq;  // what k_host_flag raises when this step's logits have reached the host
}
__global__ __launch_bounds__(256) void k_logits_to_host(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int n4, const float* __restrict__ src1,
                                                        float* __restrict__ dst1, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) __builtin_nontemporal_store(src[i], dst + i);
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst1[n4 * 4 + threadIdx.x] = src1[n4 * 4 + threadIdx.x];
}
__global__ void k_host_flag(unsigned* __restrict__ flag, const int* __restrict__ seq_d, const int* __restrict__ fault) {
  const unsigned seq = (unsigned)*seq_d;  // (from device memory: the launch may be a node of the step's replayed graph)
  flag[1] = (unsigned)*fault;
  __threadfence_system();  // (the copy kernel has completed: stream order; this orders the fault word before the flag)
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

namespace {

TpP2P p2p_view(const crabml_hip_tp_comm* m);
// the inbox view a decode-step kernel gets: empty (n = 0) unless this context runs the fused collective; timeouts raise the
// context's own fault word, which forward / decode_greedy check at their sync
TpP2P tp_view(const crabml_hip_llama* c, bool fused_collective) {
  TpP2P t = p2p_view(fused_collective && c->comm && c->comm->p2p ? c->comm : nullptr);  // dry run: no comm, n = 0
  t.fault = c->state + 5;
  t.salt = c->tp_salt;
  return t;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the kernel in this PROCESS, not of a context: a second context
// with a shorter sequence must not lower the limit an earlier context's launches (and captured graphs) were sized for.
// Only ever raise it, per device.
hipError_t raise_dyn_lds(const crabml_hip_device* dev, const void* fn, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> have;
  if (dev->dry) return hipErrorNoDevice;  // record-only test device
  std::lock_guard<std::mutex> g(mu);
  int& cur = have[{dev->ordinal, fn}];
  if (bytes <= cur) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) cur = bytes;
  return e;
}

int dalloc(crabml_hip_llama* c, size_t bytes, void** out) {
  size_t cap = 0;
  CH_TRY(pool_alloc(c->dev, bytes, out, &cap));
  c->allocs.push_back({*out, cap});
  return 0;
}

Planes planes_of(const crabml_hip_buf* b) {
  return Planes{(const i32x4*)b->ptr, (const unsigned short*)((const char*)b->ptr + b->wl.off_scale)};
}

// the planes of a 32-block activation (Q8_0: q | d | isum i32;  Q8_1: q | d | s f16) as the kernels write them
struct ActPtrs {
  signed char* q;
  unsigned short* d;
  void* isum;  // the format's third plane
};
ActPtrs act_ptrs(char* p, size_t n, uint32_t qt) {
  ActLayout al = act_layout(qt, n);
  return ActPtrs{(signed char*)p, (unsigned short*)(p + al.off_d), (void*)(p + al.off_aux)};
}
template <int FMT>
typename ActOf<FMT>::type act_view(const ActPtrs& a) {
  if constexpr (FMT == CRABML_HIP_Q4_1)
    return ActQ8_1{(const i32x4*)a.q, a.d, (const unsigned short*)a.isum};
  else
    return ActQ8_0{(const i32x4*)a.q, a.d, (const int*)a.isum};
}

int n_segments(const crabml_hip_llama* c) { return 2 * (int)c->cfg.n_layers + 1; }

// attention of layer l (llama2.rs:571-590): qbuf x KV cache -> attn (f32), plus its Q8_0 planes for wo when xq != NULL.
// Emits the variant selected in c->attn_variant (0: one workgroup per head, 1: the long-context kernels).
template <int G>
void launch_attn_long(crabml_hip_llama* c, int l, signed char* xq, unsigned short* xd, void* xisum, bool prof) {
  crabml_hip_device* dev = c->dev;
  hipStream_t st = dev->stream;
  const int hd = c->hd, seq_cap = (int)c->cfg.seq_len, n_kv = c->n_kv_l;
  const int* pos_d = c->state + 1;
  const int ts = 256 / G, nsplit = (seq_cap + ts - 1) / ts;
  crabml_hip_device::ProfRec r[3];
  for (int i = 0; i < 3; i++)
    if (prof) prof_begin(dev, &r[i], CRABML_HIP_F32, 7 + i, 0.0);  // stages 7 / 8 / 9: scores / softmax / pv
  launch_k(st, prof ? &r[0] : nullptr, k_attn_scores<G>, dim3(n_kv * nsplit), dim3(256), (size_t)G * hd * sizeof(float),
           (const float*)c->qbuf, (const unsigned short*)c->kc[l], pos_d, c->scores_g, n_kv, hd, seq_cap, nsplit, 0);
  launch_k(st, prof ? &r[1] : nullptr, k_attn_softmax<16>, dim3(c->n_heads_l), dim3(1024), (size_t)seq_cap * sizeof(float),
           (const float*)c->scores_g, pos_d, (const unsigned short*)dev->exp_table, c->p16, seq_cap, 0, dev->strict_order ? 1 : 0);
  if (c->pv_split)
    launch_k(st, prof ? &r[2] : nullptr, k_attn_pv_split<G>, dim3(n_kv * (hd / 32) * PvSplit<G>::NSUB), dim3(PvSplit<G>::THREADS), PvSplit<G>::LDS,
             (const unsigned short*)c->p16, (const unsigned short*)c->vc[l], pos_d, c->attn, xq, xd, xisum, hd, seq_cap,
             c->qt == CRABML_HIP_Q8_1 ? 1 : 0, 0);
  else
    launch_k(st, prof ? &r[2] : nullptr, k_attn_pv<G>, dim3(n_kv * (hd / 32)), dim3(256), 0, (const unsigned short*)c->p16,
             (const unsigned short*)c->vc[l], pos_d, c->attn, xq, xd, xisum, hd, seq_cap, c->qt == CRABML_HIP_Q8_1 ? 1 : 0, 0);
  for (int i = 0; i < 3; i++)
    if (prof) prof_end(dev, &r[i]);
}

// the k_attn_flash instantiation for (G, hd, rhs type of wo); nullptr = no such kernel
typedef void (*FlashFn)(const float*, const unsigned short*, const unsigned short*, const int*, float*, unsigned*, float*, signed char*,
                        unsigned short*, void*, int, int, int);
template <int G>
FlashFn flash_kernel_g(int hd, bool q81, bool ticket) {
  if (ticket) {
    if (hd == 128) return q81 ? (FlashFn)k_attn_flash<G, 128, true, true> : (FlashFn)k_attn_flash<G, 128, false, true>;
    if (hd == 64) return q81 ? (FlashFn)k_attn_flash<G, 64, true, true> : (FlashFn)k_attn_flash<G, 64, false, true>;
    return nullptr;
  }
  if (hd == 128) return (FlashFn)k_attn_flash<G, 128, false, false>;
  if (hd == 64) return (FlashFn)k_attn_flash<G, 64, false, false>;
  return nullptr;
}
FlashFn flash_kernel(int grp, int hd, bool q81, bool ticket) {
  switch (grp) {
    case 1: return flash_kernel_g<1>(hd, q81, ticket);
    case 2: return flash_kernel_g<2>(hd, q81, ticket);
    case 4: return flash_kernel_g<4>(hd, q81, ticket);
    case 8: return flash_kernel_g<8>(hd, q81, ticket);
    default: return nullptr;
  }
}
// which attention form serves cache position `pos` (see attn_variant)
int variant_of(const crabml_hip_llama* c, size_t pos) {
  if (!(c->attn_long_ok && pos + 1 >= c->attn_long_from)) return 0;
  return c->flash_ticket_until > 0 && pos + 1 < c->flash_ticket_until ? 2 : 1;
}
void launch_attn_flash(crabml_hip_llama* c, int l, signed char* xq, unsigned short* xd, void* xisum, bool prof) {
  crabml_hip_device* dev = c->dev;
  hipStream_t st = dev->stream;
  const int hd = c->hd, grp = c->n_heads_l / c->n_kv_l;
  const bool q81 = c->qt == CRABML_HIP_Q8_1;
  const bool ticket = c->flash_ticket || c->attn_variant == 2;
  const FlashFn fn = flash_kernel(grp, hd, q81, ticket);
  crabml_hip_device::ProfRec r[2];
  if (prof) prof_begin(dev, &r[0], CRABML_HIP_F32, 7, 0.0);
  launch_k(st, prof ? &r[0] : nullptr, fn, dim3(c->n_kv_l * c->flash_S), dim3((grp == 8 ? 4 : 8) * 64), flash_lds_bytes(grp, hd),
           (const float*)c->qbuf, (const unsigned short*)c->kc[l], (const unsigned short*)c->vc[l], (const int*)(c->state + 1), c->flash_part,
           c->flash_tick, c->attn, xq, xd, xisum, (int)c->cfg.seq_len, c->flash_S, c->flash_min_rows);
  if (prof) prof_end(dev, &r[0]);
  if (ticket) return;
  if (prof) prof_begin(dev, &r[1], CRABML_HIP_F32, 8, 0.0);
  crabml_hip_device::ProfRec* R1 = prof ? &r[1] : nullptr;
  const int* pos_d = c->state + 1;
  if (hd == 128 && q81)
    launch_k(st, R1, k_attn_flash_merge<128, true>, dim3(c->n_heads_l), dim3(128), 0, (const float*)c->flash_part, pos_d, c->attn, xq, xd, xisum, grp, c->flash_S, c->flash_min_rows);
  else if (hd == 128)
    launch_k(st, R1, k_attn_flash_merge<128, false>, dim3(c->n_heads_l), dim3(128), 0, (const float*)c->flash_part, pos_d, c->attn, xq, xd, xisum, grp, c->flash_S, c->flash_min_rows);
  else if (q81)
    launch_k(st, R1, k_attn_flash_merge<64, true>, dim3(c->n_heads_l), dim3(64), 0, (const float*)c->flash_part, pos_d, c->attn, xq, xd, xisum, grp, c->flash_S, c->flash_min_rows);
  else
    launch_k(st, R1, k_attn_flash_merge<64, false>, dim3(c->n_heads_l), dim3(64), 0, (const float*)c->flash_part, pos_d, c->attn, xq, xd, xisum, grp, c->flash_S, c->flash_min_rows);
  if (prof) prof_end(dev, &r[1]);
}

void enqueue_attention(crabml_hip_llama* c, int l, signed char* xq, unsigned short* xd, void* xisum, const PrefetchPlan& pf,
                       int spare, bool prof, const AttnQ8K* k8 = nullptr) {
  crabml_hip_device* dev = c->dev;
  hipStream_t st = dev->stream;
  const int hd = c->hd, seq_cap = (int)c->cfg.seq_len, n_heads = c->n_heads_l, n_kv = c->n_kv_l;
  const int* pos_d = c->state + 1;
  if (c->attn_variant >= 1 && c->attn_flash) {
    launch_attn_flash(c, l, xq, xd, xisum, prof);
    return;
  }
  if (c->attn_variant >= 1) {
    switch (n_heads / n_kv) {
      case 1: launch_attn_long<1>(c, l, xq, xd, xisum, prof); break;
      case 2: launch_attn_long<2>(c, l, xq, xd, xisum, prof); break;
      case 4: launch_attn_long<4>(c, l, xq, xd, xisum, prof); break;
      default: launch_attn_long<8>(c, l, xq, xd, xisum, prof); break;
    }
    return;
  }
  const size_t attn_lds = (size_t)(seq_cap + hd) * sizeof(float);
  const int sbit = dev->strict_order ? 256 : 0;  // strict device: sequential softmax row sums at any length (softmax.rs:43-48)
  crabml_hip_device::ProfRec ar{};
  crabml_hip_device::ProfRec* AR = prof ? &ar : nullptr;
  if (prof) prof_begin(dev, &ar, CRABML_HIP_F32, 7, 0.0);
  if (c->attn_s_rows > 0 && hd == 128)
    launch_k(st, AR, k_attn_s<128>, dim3(n_heads + spare), dim3(256), c->attn_s_lds, (const float*)c->qbuf, (const unsigned short*)c->kc[l],
             (const unsigned short*)c->vc[l], pos_d, (const unsigned short*)dev->exp_table, c->attn, xq, xd, xisum, n_heads, n_kv, hd, seq_cap,
             c->attn_s_rows, pf, (k8 ? 2 : c->qt == CRABML_HIP_Q8_1 ? 1 : 0) | sbit, (long long*)nullptr, k8 ? *k8 : AttnQ8K{});
  else if (c->attn_s_rows > 0)
    launch_k(st, AR, k_attn_s<0>, dim3(n_heads + spare), dim3(256), c->attn_s_lds, (const float*)c->qbuf, (const unsigned short*)c->kc[l],
             (const unsigned short*)c->vc[l], pos_d, (const unsigned short*)dev->exp_table, c->attn, xq, xd, xisum, n_heads, n_kv, hd, seq_cap,
             c->attn_s_rows, pf, (k8 ? 2 : c->qt == CRABML_HIP_Q8_1 ? 1 : 0) | sbit, (long long*)nullptr, k8 ? *k8 : AttnQ8K{});
  else if (c->cfg.use_f16_kv_cache)
    launch_k(st, AR, k_attn<true>, dim3(n_heads + spare), dim3(256), attn_lds, (const float*)c->qbuf, (const void*)c->kc[l],
             (const void*)c->vc[l], pos_d, (const unsigned short*)dev->exp_table, c->attn, xq, xd, xisum, n_heads, n_kv, hd,
             seq_cap, pf, (c->qt == CRABML_HIP_Q8_1 ? 1 : 0) | sbit, (long long*)nullptr);
  else
    launch_k(st, AR, k_attn<false>, dim3(n_heads + spare), dim3(256), attn_lds, (const float*)c->qbuf, (const void*)c->kc[l],
             (const void*)c->vc[l], pos_d, (const unsigned short*)dev->exp_table, c->attn, xq, xd, xisum, n_heads, n_kv, hd,
             seq_cap, pf, (c->qt == CRABML_HIP_Q8_1 ? 1 : 0) | sbit, (long long*)nullptr);
  if (prof) prof_end(dev, &ar);
}

// the tail of the final segment: classifier GEMV over this rank's rows (all of them unless the vocabulary is split,
// llama2.rs:199-208) + greedy sampler + advance
int enqueue_classifier_and_sampler(crabml_hip_llama* c, const void* cls_act, crabml_hip_device::ProfRec* R) {
  crabml_hip_device* dev = c->dev;
  hipStream_t st = dev->stream;
  const auto& g = c->cfg;
  const int dim = (int)g.embedding_dim;
  int *token_d = c->state, *pos_d = c->state + 1, *step_d = c->state + 2;
  float* out = c->logits + c->vocab_off;
  if (dev->strict_order)
    CH_TRY(launch_gemv_strict(dev, c->output, (size_t)c->vocab_l, dim, cls_act, 1, out));
  else
    CH_TRY(launch_gemv(dev, c->output, (size_t)c->vocab_l, dim, cls_act, 1, out, R));
  if (c->ext_kv) {
    // a context driven by the recorded-op queue (lazy.hip): the HOST samples (Llama2Runner exports the logits and runs its own
    // sampler, llama2.rs:208), token / position / serial of the next step come from the host (lazy_ctx_begin) -- no sampler launch
    if (c->host_logits != nullptr) {
      const int n = c->vocab_l;
      k_logits_to_host<<<64, 256, 0, st>>>((const f32x4*)out, (f32x4*)c->host_logits, n / 4, out, c->host_logits, n);
      k_host_flag<<<1, 1, 0, st>>>((unsigned*)(c->host_logits + c->cfg.vocab_size), (const int*)(c->state + 7), c->state + 5);
    }
    CH_HIP(dev, hipGetLastError());
    return 0;
  }
  k_argmax_partial<<<ARGMAX_BLOCKS, 256, 0, st>>>(out, c->vocab_l, c->am_val, c->am_idx, c->vocab_off);
  if (c->split_vocab && c->comm && c->comm->p2p)
    k_argmax_step_tp<<<1, 64, 0, st>>>(c->am_val, c->am_idx, ARGMAX_BLOCKS, token_d, pos_d, step_d, c->out_tokens, c->out_cap, c->state + 4,
                                       tp_view(c, true), n_segments(c));
  else
    k_argmax_step<<<1, 64, 0, st>>>(c->am_val, c->am_idx, ARGMAX_BLOCKS, token_d, pos_d, step_d, c->out_tokens, c->out_cap, c->state + 4,
                                    c->split_vocab ? c->am_best : (int*)nullptr);
  CH_HIP(dev, hipGetLastError());
  return 0;
}

// enqueue segment `seg` of one decode step on the device stream (see the banner above): the fused kernels
// (fast mode, Q4_0 / Q8_0 weights)
template <int FMT>
int enqueue_segment_t(crabml_hip_llama* c, int seg) {
  crabml_hip_device* dev = c->dev;
  hipStream_t st = dev->stream;
  const auto& g = c->cfg;
  const int dim = (int)g.embedding_dim, hd = c->hd, seq_cap = (int)g.seq_len;
  const int dim_l = c->dim_l, kv_dim_l = c->kv_dim_l, hidden_l = c->hidden_l;
  const int n_heads_l = c->n_heads_l;
  const bool kv16 = g.use_f16_kv_cache != 0;
  const bool tp = c->tp > 1;
  const int L = (int)g.n_layers;
  int* token_d = c->state;
  int* pos_d = c->state + 1;
  constexpr bool Q81 = FMT == CRABML_HIP_Q4_1;
  const uint32_t qt = c->qt;
  ActPtrs ad = act_ptrs(c->act_dim, dim, qt), aa = act_ptrs(c->act_attn, dim_l, qt), ah = act_ptrs(c->act_hid, hidden_l, qt);
  // measurement hook: only meaningful for eager launches (events cannot live inside the captured graph)
  const bool prof = dev->prof_on && !c->use_graph && !c->capturing;
  const double blk_b = (double)block_bytes(c->wtype) / 32.0;  // weight bytes per element
  auto P0 = [&](crabml_hip_device::ProfRec* r, uint32_t stage, double rows, double k) {
    return prof ? prof_begin(dev, r, c->wtype, stage, rows * k * blk_b + 4.0 * k + 4.0 * rows) : 0;
  };
  auto P1 = [&](crabml_hip_device::ProfRec* r) { return prof ? prof_end(dev, r) : 0; };
  crabml_hip_device::ProfRec pr{};
  crabml_hip_device::ProfRec* R = prof ? &pr : nullptr;
  const bool do_pf = !(g.flags & CRABML_HIP_LLAMA_NO_PREFETCH);
  auto plan = [&](const crabml_hip_buf* a, const crabml_hip_buf* b, const crabml_hip_buf* cc) {
    PrefetchPlan pf{};
    const crabml_hip_buf* v[3] = {a, b, cc};
    for (int i = 0; i < 3; i++) {
      pf.p[i] = do_pf && v[i] ? v[i]->ptr : nullptr;
      pf.n[i] = do_pf && v[i] ? (v[i]->wl.total / 16) * 16 : 0;
    }
    pf.sink = c->state + 3;
    return pf;
  };
  const int spare = do_pf ? (dev->n_cu > 1 ? dev->n_cu - 1 : 0) : 0;
  const size_t norm_lds = norm_lds_bytes(dim);
  // rmsnorm * weight -> act_dim; with tp the previous segment's all-reduced output is folded into x first
  auto norm_quant = [&](const float* wn, float eps, bool add_pending, const PrefetchPlan& pf) {
    crabml_hip_device::ProfRec nr{};
    if (prof) prof_begin(dev, &nr, CRABML_HIP_F32, 6, 8.0 * dim);
    const float* addv = add_pending ? c->partial : nullptr;
    if (dim <= 4096)
      launch_k(st, prof ? &nr : nullptr, k_norm_quant<4, Q81>, dim3(1 + spare), dim3(1024), norm_lds, c->x, addv, wn, dim, eps, ad.q, ad.d, ad.isum, pf, c->ord ? 0 : 1);
    else
      launch_k(st, prof ? &nr : nullptr, k_norm_quant<12, Q81>, dim3(1 + spare), dim3(1024), norm_lds, c->x, addv, wn, dim, eps, ad.q, ad.d, ad.isum, pf, c->ord ? 0 : 1);
    if (prof) prof_end(dev, &nr);
  };
  // W(dim x k_local) . act -> x (+= residual) or partial (tp)
  const bool norm_epi = c->norm_epi;
  // workgroups per 32-row chunk of a wo / ffn_down launch: two for long rows (ffn_down), so that every CU streams
  auto split_of = [&](int k) {
    return (g.flags & CRABML_HIP_LLAMA_SPLIT_CHUNKS_ALWAYS)  ? 2
           : (g.flags & CRABML_HIP_LLAMA_SPLIT_CHUNKS_NEVER) ? 1
           : (k / 32 >= 256 && dim / 32 <= dev->n_cu)        ? 2
                                                             : 1;
  };
  // q / k / v rows of exactly 128 units: both 64-unit steps requested up front (5.29 -> 4.66 us per launch on the 8B shape,
  // profiles/r06_small_stage_ab.md; bit-identical).  A/B hook: CRABML_HIP_TEST_HOOKS=1 CRABML_HIP_QKV_UPFRONT=0 keeps the two rounds.
  static const int qkv_upfront = [] {
    const char* h = getenv("CRABML_HIP_TEST_HOOKS");
    const char* e = getenv("CRABML_HIP_QKV_UPFRONT");
    return h && h[0] == '1' && e && e[0] == '0' ? 0 : 1;
  }();
  // the hop-free norm between wo and gate/up of a layer: decided once, for the producer and the consumer alike
  const bool defer_wo = c->defer_norm && !Q81;
  // ... and between ffn_down of layer l and q/k/v of layer l + 1 (the last ffn_down feeds the classifier launch: exact planes)
  const bool defer_down = c->defer_norm && !Q81;
  // wnext / eps_next: the RMSNorm that consumes this GEMV's output (norm epilogue only)
  auto gemv_out = [&](const crabml_hip_buf* w, const ActPtrs& a, int k, uint32_t stage, const float* wnext, float eps_next, bool defer = false,
                      const float* xin = nullptr) -> int {
    CH_TRY(P0(&pr, stage, dim, k));
    float* dst = tp ? c->partial : c->x;
    if (norm_epi) {
      NormGather ng{c->slots, c->slots + dim / 16, c->state + 4, c->state + 5, n_segments(c), seg, c->rsums};
      // long rows (ffn_down): two workgroups per chunk, so that every CU streams (a CU sustains ~26 GB/s here)
      const int split = split_of(k);
      const TpP2P tpv = tp_view(c, tp);
      if (xin != nullptr && !Q81) {  // (tensor-parallel ranks) the rhs arrives as f32 -- h from k_gateup_h -- and is quantized in the prologue
        if constexpr (!Q81) {
          const size_t qlds = q8_0_lds_bytes(k / 32);
          if (tpv.n > 1 && split == 2)
            launch_k(st, R, k_gemv_res_nq<FMT, 2, 1, true>, dim3(dim / 16), dim3(1024), qlds, planes_of(w), act_view<FMT>(a), xin, c->x, wnext, eps_next, ad.q,
                     ad.d, ad.isum, ng, k / 32, Planes6{nullptr, 0}, tpv);
          else if (tpv.n > 1)
            launch_k(st, R, k_gemv_res_nq<FMT, 1, 1, true>, dim3(dim / 32), dim3(1024), qlds, planes_of(w), act_view<FMT>(a), xin, c->x, wnext, eps_next, ad.q,
                     ad.d, ad.isum, ng, k / 32, Planes6{nullptr, 0}, tpv);
          else if (split == 2)
            launch_k(st, R, k_gemv_res_nq<FMT, 2, 1>, dim3(dim / 16), dim3(1024), qlds, planes_of(w), act_view<FMT>(a), xin, c->x, wnext, eps_next, ad.q,
                     ad.d, ad.isum, ng, k / 32, Planes6{nullptr, 0}, NoTp{});
          else
            launch_k(st, R, k_gemv_res_nq<FMT, 1, 1>, dim3(dim / 32), dim3(1024), qlds, planes_of(w), act_view<FMT>(a), xin, c->x, wnext, eps_next, ad.q,
                     ad.d, ad.isum, ng, k / 32, Planes6{nullptr, 0}, NoTp{});
        }
      } else if (tpv.n > 1) {  // tensor parallel over a P2P group: the collective runs inside this launch
        if (split == 2)
          launch_k(st, R, k_gemv_res_nq<FMT, 2, false, true>, dim3(dim / 16), dim3(1024), 0, planes_of(w), act_view<FMT>(a), (const float*)nullptr,
                   c->x, wnext, eps_next, ad.q, ad.d, ad.isum, ng, k / 32, Planes6{nullptr, 0}, tpv);
        else
          launch_k(st, R, k_gemv_res_nq<FMT, 1, false, true>, dim3(dim / 32), dim3(1024), 0, planes_of(w), act_view<FMT>(a), (const float*)nullptr,
                   c->x, wnext, eps_next, ad.q, ad.d, ad.isum, ng, k / 32, Planes6{nullptr, 0}, tpv);
      } else if (c->ord) {  // strict order: the same launch with block-ordered GEMV sums and the reference's norm order
        const size_t lds = (size_t)(32 / split) * (((k / 32 + 3) & ~3) + 4) * sizeof(float);
        static const int pipe_mode = [] {  // tuning hook: 0 = never, 1 = ffn_down only, 2 = wo too (-1 / unset: the default below)
          const char* h = getenv("CRABML_HIP_TEST_HOOKS");
          const char* e = getenv("CRABML_HIP_ORD_PIPE");
          return h && h[0] == '1' && e ? atoi(e) : -1;
        }();
        // measured (profiles/r04_strict_order_decode.md): the pipelined chain pays in ffn_down for every format, in wo for Q8_0 only
        const int pipe = pipe_mode >= 0 ? pipe_mode : FMT == CRABML_HIP_Q8_0 ? 2 : 1;
        if (split == 2) {
          if (pipe >= 1)
            launch_k(st, R, k_gemv_res_nq_ord<FMT, 2, true>, dim3(dim / 16), dim3(1024), lds, planes_of(w), act_view<FMT>(a), c->x, wnext, eps_next, ad.q,
                     ad.d, ad.isum, ng, k / 32);
          else
            launch_k(st, R, k_gemv_res_nq_ord<FMT, 2, false>, dim3(dim / 16), dim3(1024), lds, planes_of(w), act_view<FMT>(a), c->x, wnext, eps_next, ad.q,
                     ad.d, ad.isum, ng, k / 32);
        } else {
          if (pipe >= 2)
            launch_k(st, R, k_gemv_res_nq_ord<FMT, 1, true>, dim3(dim / 32), dim3(1024), lds, planes_of(w), act_view<FMT>(a), c->x, wnext, eps_next, ad.q,
                     ad.d, ad.isum, ng, k / 32);
          else
            launch_k(st, R, k_gemv_res_nq_ord<FMT, 1, false>, dim3(dim / 32), dim3(1024), lds, planes_of(w), act_view<FMT>(a), c->x, wnext, eps_next, ad.q,
                     ad.d, ad.isum, ng, k / 32);
        }
      } else if (defer && !Q81) {  // hop-free: the consumer applies 1 / rms
        if constexpr (!Q81) {
          if (split == 2)
            launch_k(st, R, k_gemv_res_nq<FMT, 2, 0, false, true>, dim3(dim / 16), dim3(1024), 0, planes_of(w), act_view<FMT>(a), (const float*)nullptr,
                     c->x, wnext, eps_next, ad.q, ad.d, ad.isum, ng, k / 32, Planes6{nullptr, 0}, NoTp{});
          else
            launch_k(st, R, k_gemv_res_nq<FMT, 1, 0, false, true>, dim3(dim / 32), dim3(1024), 0, planes_of(w), act_view<FMT>(a), (const float*)nullptr,
                     c->x, wnext, eps_next, ad.q, ad.d, ad.isum, ng, k / 32, Planes6{nullptr, 0}, NoTp{});
        }
      } else if (split == 2)
        launch_k(st, R, k_gemv_res_nq<FMT, 2>, dim3(dim / 16), dim3(1024), 0, planes_of(w), act_view<FMT>(a), (const float*)nullptr, c->x, wnext,
                 eps_next,
                 ad.q, ad.d, ad.isum, ng, k / 32, Planes6{nullptr, 0}, NoTp{});
      else
        launch_k(st, R, k_gemv_res_nq<FMT, 1>, dim3(dim / 32), dim3(1024), 0, planes_of(w), act_view<FMT>(a), (const float*)nullptr, c->x, wnext,
                 eps_next,
                 ad.q, ad.d, ad.isum, ng, k / 32, Planes6{nullptr, 0}, NoTp{});
    } else if (c->ord) {
      launch_k(st, R, k_gemv_res_ord<FMT>, dim3((dim + 7) / 8), dim3(256), (size_t)8 * ((k / 32 + 3) & ~3) * sizeof(float), planes_of(w), act_view<FMT>(a),
               c->x, dim, k / 32);
    } else if (tp) {
      launch_k(st, R, k_gemv_res<FMT, 1, false>, dim3((dim + 1) / 2), dim3(128), 0, planes_of(w), act_view<FMT>(a), dst, dim, k / 32);
    } else {
      launch_k(st, R, k_gemv_res<FMT, 1, true>, dim3((dim + 1) / 2), dim3(128), 0, planes_of(w), act_view<FMT>(a), dst, dim, k / 32);
    }
    CH_TRY(P1(&pr));
    return 0;
  };

  if (seg == 2 * L) {  // final rmsnorm + classifier (llama2.rs:274-278, 199-208) + greedy sampler
    const void* cls_act = c->act_dim;
    if (c->out_qt != qt) {
      // the classifier has its own rhs type (e.g. Q6_K -> Q8_K): normalize the final x to f32 and quantize for it (the
      // planes the last ffn_down epilogue wrote are in the layers' type and stay unused)
      const size_t nlds = norm_lds_bytes(dim);
      const float* addv = tp && !norm_epi ? c->partial : nullptr;  // (fused collective: x is already final)
      if (dim <= 4096)
        k_norm_f32<4><<<1, 1024, nlds, st>>>(c->x, addv, (const float*)c->rms_final->ptr, dim, g.rms_norm_eps, c->xn, c->ord ? 0 : 1);
      else
        k_norm_f32<12><<<1, 1024, nlds, st>>>(c->x, addv, (const float*)c->rms_final->ptr, dim, g.rms_norm_eps, c->xn, c->ord ? 0 : 1);
      if (c->out_qt == CRABML_HIP_F32) {
        cls_act = c->xn;
      } else {
        launch_quantize_act(st, c->out_qt, c->xn, (size_t)dim, c->act_dim);
      }
    } else if (!norm_epi) {
      norm_quant((const float*)c->rms_final->ptr, g.rms_norm_eps, tp, plan(nullptr, nullptr, nullptr));
    }
    if (prof)
      CH_TRY(prof_begin(dev, &pr, c->output->dtype, 5,
                        (double)c->vocab_l * (double)(dim / block_elems(c->output->dtype)) * (double)block_bytes(c->output->dtype) +
                            4.0 * dim + 4.0 * c->vocab_l));
    CH_TRY(enqueue_classifier_and_sampler(c, cls_act, R));
    CH_TRY(P1(&pr));
    return 0;
  }
  const int l = seg / 2;
  if ((seg & 1) == 0) {
    if (l == 0)
      k_embed<<<(dim + 255) / 256, 256, 0, st>>>((const char*)c->token_embed->ptr, (int)c->token_embed->dtype,
                                                  c->token_embed->wl.off_scale, token_d, dim, c->x);
    // attention rmsnorm (llama2.rs:230-234)
    if (!norm_epi || l == 0)
      norm_quant((const float*)c->rms_att[l]->ptr, g.rms_norm_eps, tp && l > 0, plan(c->wq[l], c->wk[l], c->wv[l]));
    // q, k, v + rope + scale + KV append (llama2.rs:244-256, 542-554, 561-565), local heads only
    QkvEpi e{c->qbuf, c->kc[l], c->vc[l], c->rope, pos_d, 1.0f / std::sqrt((float)hd), dim_l, kv_dim_l, hd,
             (int)g.rope_dim, c->npairs, seq_cap, kv16 ? 1 : 0};
    const int total_rows = dim_l + 2 * kv_dim_l;
    CH_TRY(P0(&pr, 1, total_rows, dim));
    // (the planes of layer l > 0 come from the previous layer's ffn_down launch, with its dim / 32 chunk sums)
    const RmsTail rtq{c->rsums, dim / 32, 1.0f / (float)dim, g.rms_norm_eps};
    if (c->ord)
      launch_k(st, R, k_qkv_ord<FMT>, dim3((total_rows / 2 + 3) / 4), dim3(256), (size_t)8 * ((dim / 32 + 3) & ~3) * sizeof(float), planes_of(c->wq[l]),
               planes_of(c->wk[l]), planes_of(c->wv[l]), act_view<FMT>(ad), dim / 32, e, Planes6{nullptr, 0});
    else if (defer_down && l > 0) {
      if constexpr (!Q81)
        launch_k(st, R, k_qkv<FMT, true>, dim3((total_rows / 2 + 1) / 2), dim3(128), 0, planes_of(c->wq[l]), planes_of(c->wk[l]),
                 planes_of(c->wv[l]), act_view<FMT>(ad), dim / 32, e, Planes6{nullptr, 0}, rtq, qkv_upfront);
    } else
      launch_k(st, R, k_qkv<FMT>, dim3((total_rows / 2 + 1) / 2), dim3(128), 0, planes_of(c->wq[l]), planes_of(c->wk[l]),
               planes_of(c->wv[l]), act_view<FMT>(ad), dim / 32, e, Planes6{nullptr, 0}, rtq,
               // few, short waves (a tensor-parallel rank's rows; small models): two steps per request round
               (qkv_upfront && total_rows / 2 <= 4 * dev->n_cu) ? 1 : 0);
    CH_TRY(P1(&pr));
    // attention (llama2.rs:571-590) -> attn (f32) [+ Q8_0 planes for wo]; spare CUs prefetch wo
    const bool attn_quant = (hd % 32) == 0;
    const int attn_spare = do_pf && dev->n_cu > n_heads_l ? dev->n_cu - n_heads_l : 0;
    enqueue_attention(c, l, attn_quant ? aa.q : (signed char*)nullptr, aa.d, aa.isum, plan(c->wo[l], nullptr, nullptr), attn_spare, prof);
    if (!attn_quant) launch_quantize_act(st, qt, c->attn, (size_t)dim_l, c->act_attn);
    // wo (+ residual, llama2.rs:600, 266): k = the local heads' slice
    CH_TRY(gemv_out(c->wo[l], aa, dim_l, 2, (const float*)c->rms_ffn[l]->ptr, 1e-5f, defer_wo));
  } else {
    // ffn rmsnorm, eps = the literal 1e-5 (llama2.rs:611)
    if (!norm_epi) norm_quant((const float*)c->rms_ffn[l]->ptr, 1e-5f, tp, plan(nullptr, nullptr, nullptr));
    const float* wnext_down = (const float*)(l + 1 < L ? c->rms_att[l + 1] : c->rms_final)->ptr;
    // gate / up + silu * mul (llama2.rs:620-630), local rows
    CH_TRY(P0(&pr, 3, 2.0 * hidden_l, dim));
    const RmsTail rt{c->rsums, dim / 32, 1.0f / (float)dim, 1e-5f};  // eps: the literal 1e-5 (llama2.rs:611)
    const bool hq = c->gu_rows > 0 && !c->ord && norm_epi && !Q81 && !defer_wo;
    if (hq) {
      if constexpr (!Q81)
        launch_k(st, R, k_gateup_h<FMT>, dim3(hidden_l / c->gu_rows), dim3(c->gu_rows / 2 * 64), 0, planes_of(c->gate[l]), planes_of(c->up[l]),
                 act_view<FMT>(ad), (const unsigned short*)dev->exp_table, c->h, hidden_l, dim / 32);
    } else if (c->ord)
      launch_k(st, R, k_gateup_q_ord<FMT>, dim3(hidden_l / 32), dim3(1024), (size_t)64 * (((dim / 32 + 3) & ~3) + 4) * sizeof(float), planes_of(c->gate[l]),
               planes_of(c->up[l]), act_view<FMT>(ad), dev->exp_table, ah.q, ah.d, ah.isum, dim / 32);
    else if (defer_wo) {
      if constexpr (!Q81)
        launch_k(st, R, k_gateup_q<FMT, true>, dim3(hidden_l / 32), dim3(1024), 0, planes_of(c->gate[l]), planes_of(c->up[l]), act_view<FMT>(ad),
                 dev->exp_table, ah.q, ah.d, ah.isum, dim / 32, rt);
    } else
      launch_k(st, R, k_gateup_q<FMT>, dim3(hidden_l / 32), dim3(1024), 0, planes_of(c->gate[l]), planes_of(c->up[l]),
               act_view<FMT>(ad), dev->exp_table, ah.q, ah.d, ah.isum, dim / 32, rt);
    CH_TRY(P1(&pr));
    // down (+ residual, llama2.rs:633-636): k = the local hidden slice
    CH_TRY(gemv_out(c->down[l], ah, hidden_l, 4, wnext_down, g.rms_norm_eps, defer_down && l + 1 < L, hq ? c->h : nullptr));
  }
  CH_HIP(dev, hipGetLastError());
  return 0;
}

// The same segment out of per-op launches: one GEMV launch per weight matrix (any format matmul_vec supports;
// the rhs is quantized to vec_dot_rhs_dtype(weight), buf/api.rs:142-159) plus small epilogue kernels.  Used by
// the strict-order device (scalar summation order, bit-exact against the oracle for every format) and, in fast
// mode, by the formats without fused kernels (Q4_1, Q4_K, Q8_K, F16, F32).  Still one hipGraph per step.
int enqueue_segment_generic(crabml_hip_llama* c, int seg) {
  crabml_hip_device* dev = c->dev;
  hipStream_t st = dev->stream;
  const auto& g = c->cfg;
  const int dim = (int)g.embedding_dim, hd = c->hd, seq_cap = (int)g.seq_len;
  const int dim_l = c->dim_l, kv_dim_l = c->kv_dim_l, hidden_l = c->hidden_l;
  const bool kv16 = g.use_f16_kv_cache != 0;
  const bool strict = dev->strict_order;
  const bool tp = c->tp > 1;
  const int L = (int)g.n_layers;
  int* token_d = c->state;
  int* pos_d = c->state + 1;
  const bool prof = dev->prof_on && !c->use_graph && !c->capturing && !strict;
  crabml_hip_device::ProfRec pr{};
  auto gemv = [&](const crabml_hip_buf* w, int m, int k, const void* act, float* out, uint32_t stage) -> int {
    if (strict) return launch_gemv_strict(dev, w, m, k, act, 1, out);
    if (prof)
      CH_TRY(prof_begin(dev, &pr, w->dtype, stage,
                        (double)m * (double)(k / block_elems(w->dtype)) * (double)block_bytes(w->dtype) + 4.0 * k + 4.0 * m));
    CH_TRY(launch_gemv(dev, w, m, k, act, 1, out, prof ? &pr : nullptr));
    if (prof) CH_TRY(prof_end(dev, &pr));
    return 0;
  };
  // CpuTensorBuf::quantize for the rhs of matmul_vec: F32 is the vector itself
  auto quant = [&](const float* src, int n, uint32_t qt, char* planes) -> const void* {
    if (qt == CRABML_HIP_F32) return src;
    launch_quantize_act(st, qt, src, (size_t)n, planes);
    return planes;
  };
  const size_t norm_lds = norm_lds_bytes(dim);
  auto norm = [&](const float* wn, float eps, bool add_pending) {
    const float* addv = add_pending ? c->partial : nullptr;
    if (dim <= 4096)
      k_norm_f32<4><<<1, 1024, norm_lds, st>>>(c->x, addv, wn, dim, eps, c->xn, strict ? 0 : 1);
    else
      k_norm_f32<12><<<1, 1024, norm_lds, st>>>(c->x, addv, wn, dim, eps, c->xn, strict ? 0 : 1);
  };
  float* dst = tp ? c->partial : c->x;

  if (seg == 2 * L) {
    norm((const float*)c->rms_final->ptr, g.rms_norm_eps, tp);
    const void* act = quant(c->xn, dim, c->out_qt, c->act_dim);
    if (prof)
      CH_TRY(prof_begin(dev, &pr, c->output->dtype, 5,
                        (double)c->vocab_l * (double)(dim / block_elems(c->output->dtype)) * (double)block_bytes(c->output->dtype) + 4.0 * dim +
                            4.0 * c->vocab_l));
    CH_TRY(enqueue_classifier_and_sampler(c, act, prof ? &pr : nullptr));
    if (prof) CH_TRY(prof_end(dev, &pr));
    return 0;
  }
  const int l = seg / 2;
  if ((seg & 1) == 0) {
    if (l == 0)
      k_embed<<<(dim + 255) / 256, 256, 0, st>>>((const char*)c->token_embed->ptr, (int)c->token_embed->dtype,
                                                  c->token_embed->wl.off_scale, token_d, dim, c->x);
    norm((const float*)c->rms_att[l]->ptr, g.rms_norm_eps, tp && l > 0);
    const void* act = quant(c->xn, dim, c->qt, c->act_dim);
    QkvEpi e{c->qbuf, c->kc[l], c->vc[l], c->rope, pos_d, 1.0f / std::sqrt((float)hd), dim_l, kv_dim_l, hd,
             (int)g.rope_dim, c->npairs, seq_cap, kv16 ? 1 : 0};
    const int total_rows = dim_l + 2 * kv_dim_l;
    CH_TRY(gemv(c->wq[l], dim_l, dim, act, c->tmp, 1));
    CH_TRY(gemv(c->wk[l], kv_dim_l, dim, act, c->tmp + dim_l, 1));
    CH_TRY(gemv(c->wv[l], kv_dim_l, dim, act, c->tmp + dim_l + kv_dim_l, 1));
    k_qkv_epi<<<(total_rows / 2 + 255) / 256, 256, 0, st>>>(c->tmp, e);
    enqueue_attention(c, l, nullptr, nullptr, nullptr, PrefetchPlan{}, 0, prof);
    const void* aact = quant(c->attn, dim_l, c->qt, c->act_attn);
    if (strict && !tp) {  // the residual inside the GEMV's own store: x = matmul_out + x (llama2.rs:266)
      CH_TRY(launch_gemv_strict(dev, c->wo[l], dim, dim_l, aact, 1, c->x, c->x));
    } else {
      CH_TRY(gemv(c->wo[l], dim, dim_l, aact, c->tmp, 2));
      k_res_epi<<<(dim + 255) / 256, 256, 0, st>>>(c->tmp, dst, dim, tp ? 0 : 1);
    }
  } else {
    norm((const float*)c->rms_ffn[l]->ptr, 1e-5f, tp);  // llama2.rs:611
    const void* act = quant(c->xn, dim, c->qt, c->act_dim);
    CH_TRY(gemv(c->gate[l], hidden_l, dim, act, c->tmp, 3));
    CH_TRY(gemv(c->up[l], hidden_l, dim, act, c->tmp + hidden_l, 3));
    k_gateup_epi<<<(hidden_l + 255) / 256, 256, 0, st>>>(c->tmp, c->tmp + hidden_l, dev->exp_table, c->h, hidden_l);
    const void* hact = quant(c->h, hidden_l, c->qt, c->act_hid);
    if (strict && !tp) {
      CH_TRY(launch_gemv_strict(dev, c->down[l], dim, hidden_l, hact, 1, c->x, c->x));
    } else {
      CH_TRY(gemv(c->down[l], dim, hidden_l, hact, c->tmp, 4));
      k_res_epi<<<(dim + 255) / 256, 256, 0, st>>>(c->tmp, dst, dim, tp ? 0 : 1);
    }
  }
  CH_HIP(dev, hipGetLastError());
  return 0;
}

// Q4_K and Q4_1 layers (fast mode): the fused GEMV kernels with the format's inner loop against Q8_K / Q8_1
// activation planes.  The rhs quantizer is its own launch here (a Q8_K super-block spans 256 rows: eight 32-row
// workgroups; Q8_1 keeps the same structure), so a layer is 11 launches instead of the per-op path's 18.
template <int FMT>
int enqueue_segment_k(crabml_hip_llama* c, int seg) {
  constexpr uint32_t QT = FMT == CRABML_HIP_Q4_K ? CRABML_HIP_Q8_K : CRABML_HIP_Q8_1;
  constexpr int BE = FMT == CRABML_HIP_Q4_K ? 256 : 32;  // elements per weight block
  typedef typename ActOf<FMT>::type Act;
  crabml_hip_device* dev = c->dev;
  hipStream_t st = dev->stream;
  const auto& g = c->cfg;
  const int dim = (int)g.embedding_dim, hd = c->hd, seq_cap = (int)g.seq_len;
  const int dim_l = c->dim_l, kv_dim_l = c->kv_dim_l, hidden_l = c->hidden_l;
  const bool kv16 = g.use_f16_kv_cache != 0;
  const bool tp = c->tp > 1;
  const int L = (int)g.n_layers;
  int* token_d = c->state;
  int* pos_d = c->state + 1;
  const bool prof = dev->prof_on && !c->use_graph && !c->capturing;
  crabml_hip_device::ProfRec pr{};
  crabml_hip_device::ProfRec* R = prof ? &pr : nullptr;
  const double blk_b = (double)block_bytes(FMT) / (double)BE;
  auto P0 = [&](uint32_t stage, double rows, double k) {
    return prof ? prof_begin(dev, &pr, FMT, stage, rows * k * blk_b + 4.0 * k + 4.0 * rows) : 0;
  };
  auto P1 = [&]() { return prof ? prof_end(dev, &pr) : 0; };
  auto act_k = [&](char* planes, int n) {
    ActLayout al = act_layout(QT, (size_t)n);
    if constexpr (FMT == CRABML_HIP_Q4_K)
      return act_q8k_at(planes, al.off_d, al.off_aux, al.off_p);
    else
      return ActQ8_1{(const i32x4*)planes, (const unsigned short*)(planes + al.off_d), (const unsigned short*)(planes + al.off_aux)};
  };
  auto planes_k = [&](const crabml_hip_buf* b) {
    return Planes{(const i32x4*)b->ptr, (const unsigned short*)((const char*)b->ptr + b->wl.off_scale)};
  };
  // a Q6_K tensor inside a Q4_K layer (attn_v / ffn_down of the *_K_M mixes): handed to the kernel beside the planes
  auto six = [&](const crabml_hip_buf* b) {
    return b->dtype == CRABML_HIP_Q6_K ? Planes6{(const char*)b->ptr, b->wl.off_scale} : Planes6{nullptr, 0};
  };
  const size_t norm_lds = norm_lds_bytes(dim);
  // rmsnorm * weight -> xn -> Q8_K planes (buf_q8_k.rs:84-131)
  auto norm_quant = [&](const float* wn, float eps, bool add_pending, uint32_t qt) -> const void* {
    const float* addv = add_pending ? c->partial : nullptr;
    if (dim <= 4096)
      k_norm_f32<4><<<1, 1024, norm_lds, st>>>(c->x, addv, wn, dim, eps, c->xn, c->ord ? 0 : 1);
    else
      k_norm_f32<12><<<1, 1024, norm_lds, st>>>(c->x, addv, wn, dim, eps, c->xn, c->ord ? 0 : 1);
    if (qt == CRABML_HIP_F32) return c->xn;
    launch_quantize_act(st, qt, c->xn, (size_t)dim, c->act_dim);
    return c->act_dim;
  };
  float* dst = tp ? c->partial : c->x;
  const bool nepi = FMT == CRABML_HIP_Q4_K && c->norm_epi_k;
  // strict-order device, Q4_K layers (c->ord): the same five launches with every sum in the reference's order -- nine-term records per
  // super-block added in order (q4k_class_terms / q4k_ordered_sum, gemv_core.hpp), the reference's norm order in the epilogue
  const bool ordk = FMT == CRABML_HIP_Q4_K && c->ord;
  // wnext / eps_next: the RMSNorm that consumes this GEMV's output (norm epilogue only)
  // the rhs of wo / ffn_down quantized by the consuming kernel itself (no quantizer launch)
  const bool qin = nepi && !(g.flags & CRABML_HIP_LLAMA_NO_RHS_PROLOGUE) && dim_l % 256 == 0 && hidden_l % 256 == 0;
  const bool qout = qin && c->q8k_producers;  // attention / gate-up write the planes, wo / ffn_down copy them
  // wnext / eps_next: the RMSNorm that consumes this GEMV's output (norm epilogue only); xin: the f32 rhs
  // qmode: 0 = rhs planes from global memory, 1 = quantize the f32 rhs in the kernel's prologue, 2 = copy finished planes
  auto gemv_out = [&](const crabml_hip_buf* w, const Act& a, const float* xin, int k, uint32_t stage, const float* wnext,
                      float eps_next, int qmode, bool x_only = false) -> int {
    CH_TRY(P0(stage, dim, k));
    if constexpr (FMT == CRABML_HIP_Q4_K) {
      if (nepi) {
        NormGather ng{c->slots, c->slots + dim / 16, c->state + 4, c->state + 5, n_segments(c), seg, c->rsums};
        ActLayout al = act_layout(QT, (size_t)dim);
        ng.qp = (signed char*)(c->act_dim + al.off_p);
        signed char* oq = x_only ? nullptr : (signed char*)c->act_dim;  // (x_only: the consumer quantizes, nq_epilogue)
        void* od = (void*)(c->act_dim + al.off_d);
        void* ob = (void*)(c->act_dim + al.off_aux);
        // (x_only: no hop pairs the halves of a chunk -- two 16-row workgroups per chunk whenever that still fits the chip twice)
        const int split = (g.flags & CRABML_HIP_LLAMA_SPLIT_CHUNKS_ALWAYS)  ? 2
                          : (g.flags & CRABML_HIP_LLAMA_SPLIT_CHUNKS_NEVER) ? 1
                          : (k / 32 >= 256 && dim / 32 <= dev->n_cu)        ? 2
                          : (x_only && dim / 32 <= dev->n_cu)               ? 2
                                                                            : 1;
        const size_t lds = (size_t)k + (size_t)(k / 256) * 4 + (size_t)(k / 16) * 2;
        if (ordk) {  // (qmode is 1 or 2 here: `qin` holds on every ordered context)
          const size_t ldso = ((lds + 15) & ~(size_t)15) + (size_t)(32 / split) * (size_t)q4k_rec_stride(k / 256) * sizeof(float);
#define CRABML_NQ_KO(SPLIT_, QIN_, GRID_)                                                                                                \
  launch_k(st, R, k_gemv_res_nq<FMT, SPLIT_, QIN_, false, false, true>, dim3(GRID_), dim3(1024), ldso, planes_k(w), a, xin, c->x, wnext, \
           eps_next, oq, od, ob, ng, k / BE, six(w), NoTp{})
          if (split == 2 && qmode == 2)
            CRABML_NQ_KO(2, 2, dim / 16);
          else if (split == 2)
            CRABML_NQ_KO(2, 1, dim / 16);
          else if (qmode == 2)
            CRABML_NQ_KO(1, 2, dim / 32);
          else
            CRABML_NQ_KO(1, 1, dim / 32);
#undef CRABML_NQ_KO
          return P1();
        }
#define CRABML_NQ_K(SPLIT_, QIN_, GRID_, LDS_)                                                                                          \
  launch_k(st, R, k_gemv_res_nq<FMT, SPLIT_, QIN_>, dim3(GRID_), dim3(1024), LDS_, planes_k(w), a, xin, c->x, wnext, eps_next, oq, od, ob, \
           ng, k / BE, six(w), NoTp{})
        if (split == 2 && qmode == 2)
          CRABML_NQ_K(2, 2, dim / 16, lds);
        else if (split == 2 && qmode == 1)
          CRABML_NQ_K(2, 1, dim / 16, lds);
        else if (split == 2)
          CRABML_NQ_K(2, 0, dim / 16, 0);
        else if (qmode == 2)
          CRABML_NQ_K(1, 2, dim / 32, lds);
        else if (qmode == 1)
          CRABML_NQ_K(1, 1, dim / 32, lds);
        else
          CRABML_NQ_K(1, 0, dim / 32, 0);
#undef CRABML_NQ_K
        return P1();
      }
    }
    if (tp)
      launch_k(st, R, k_gemv_res<FMT, 1, false>, dim3((dim + 1) / 2), dim3(128), 0, planes_k(w), a, dst, dim, k / BE);
    else
      launch_k(st, R, k_gemv_res<FMT, 1, true>, dim3((dim + 1) / 2), dim3(128), 0, planes_k(w), a, dst, dim, k / BE);
    return P1();
  };

  if (seg == 2 * L) {
    const void* act = nepi ? (const void*)c->act_dim : norm_quant((const float*)c->rms_final->ptr, g.rms_norm_eps, tp, c->out_qt);
    if (prof)
      CH_TRY(prof_begin(dev, &pr, c->output->dtype, 5,
                        (double)c->vocab_l * (double)(dim / block_elems(c->output->dtype)) * (double)block_bytes(c->output->dtype) +
                            4.0 * dim + 4.0 * c->vocab_l));
    CH_TRY(enqueue_classifier_and_sampler(c, act, R));
    CH_TRY(P1());
    return 0;
  }
  const int l = seg / 2;
  if ((seg & 1) == 0) {
    if (l == 0)
      k_embed<<<(dim + 255) / 256, 256, 0, st>>>((const char*)c->token_embed->ptr, (int)c->token_embed->dtype,
                                                  c->token_embed->wl.off_scale, token_d, dim, c->x);
    if (!nepi || l == 0) norm_quant((const float*)c->rms_att[l]->ptr, g.rms_norm_eps, tp && l > 0, QT);
    QkvEpi e{c->qbuf, c->kc[l], c->vc[l], c->rope, pos_d, 1.0f / std::sqrt((float)hd), dim_l, kv_dim_l, hd,
             (int)g.rope_dim, c->npairs, seq_cap, kv16 ? 1 : 0};
    const int total_rows = dim_l + 2 * kv_dim_l;
    CH_TRY(P0(1, total_rows, dim));
    if (ordk)
      launch_k(st, R, k_qkv_ord<FMT>, dim3((total_rows / 2 + 3) / 4), dim3(256), (size_t)8 * (size_t)q4k_rec_stride(dim / BE) * sizeof(float),
               planes_k(c->wq[l]), planes_k(c->wk[l]), planes_k(c->wv[l]), act_k(c->act_dim, dim), dim / BE, e, six(c->wv[l]));
    else
      launch_k(st, R, k_qkv<FMT>, dim3((total_rows / 2 + 1) / 2), dim3(128), 0, planes_k(c->wq[l]), planes_k(c->wk[l]),
               planes_k(c->wv[l]), act_k(c->act_dim, dim), dim / BE, e, six(c->wv[l]), RmsTail{nullptr, 0, 0.f, 0.f}, 0);
    CH_TRY(P1());
    // Q8_K producers: the (short-context) attention kernel assembles the planes of wo's rhs itself; wo copies them
    const bool aq8 = qout && (g.flags & CRABML_HIP_LLAMA_Q8K_ATTN_PRODUCER) && c->attn_variant == 0 && c->attn_s_rows > 0;
    if constexpr (FMT == CRABML_HIP_Q4_K) {
      if (aq8) {
        const ActLayout ala = act_layout(QT, (size_t)dim_l);
        const AttnQ8K k8{Q8KExchange{c->a8gran, c->state + 4, c->state + 5, n_segments(c), seg}, (float*)(c->act_attn + ala.off_d),
                         (short*)(c->act_attn + ala.off_aux), (signed char*)(c->act_attn + ala.off_p)};
        enqueue_attention(c, l, (signed char*)c->act_attn, nullptr, nullptr, PrefetchPlan{}, 0, prof, &k8);
      } else {
        enqueue_attention(c, l, nullptr, nullptr, nullptr, PrefetchPlan{}, 0, prof);
      }
    } else {
      enqueue_attention(c, l, nullptr, nullptr, nullptr, PrefetchPlan{}, 0, prof);
    }
    if (!qin) launch_quantize_act(st, QT, c->attn, (size_t)dim_l, c->act_attn);
    CH_TRY(gemv_out(c->wo[l], act_k(c->act_attn, dim_l), c->attn, dim_l, 2, (const float*)c->rms_ffn[l]->ptr, 1e-5f, aq8 ? 2 : qin ? 1 : 0,
                    qout && c->k_norm_in));
  } else {
    if (!nepi) norm_quant((const float*)c->rms_ffn[l]->ptr, 1e-5f, tp, QT);  // llama2.rs:611
    CH_TRY(P0(3, 2.0 * hidden_l, dim));
    if constexpr (FMT == CRABML_HIP_Q4_K) {
      const size_t lds = (size_t)dim + (size_t)(dim / 256) * 4 + (size_t)(dim / 16) * 2;
      const ActLayout alh = act_layout(QT, (size_t)hidden_l);
      const Q8KExchange hx{c->h8gran, c->state + 4, c->state + 5, n_segments(c), seg};
      const size_t ldso = ((lds + 15) & ~(size_t)15) + (size_t)64 * (size_t)q4k_rec_stride(dim / 256) * sizeof(float);
      if (ordk && qout)
        launch_k(st, R, k_gateup_k_lds<true, true>, dim3(hidden_l / 32), dim3(1024), ldso, planes_k(c->gate[l]), planes_k(c->up[l]),
                 act_k(c->act_dim, dim), (const unsigned short*)dev->exp_table, c->h, hidden_l, dim / 256, hx, (signed char*)c->act_hid,
                 (float*)(c->act_hid + alh.off_d), (short*)(c->act_hid + alh.off_aux), (signed char*)(c->act_hid + alh.off_p), (const float*)nullptr, (const float*)nullptr, 0.f, (const float*)nullptr, 1);
      else if (ordk)
        launch_k(st, R, k_gateup_k_lds<false, true>, dim3(hidden_l / 32), dim3(1024), ldso, planes_k(c->gate[l]), planes_k(c->up[l]),
                 act_k(c->act_dim, dim), (const unsigned short*)dev->exp_table, c->h, hidden_l, dim / 256, hx, (signed char*)nullptr,
                 (float*)nullptr, (short*)nullptr, (signed char*)nullptr, (const float*)nullptr, (const float*)nullptr, 0.f, (const float*)nullptr, 1);
      else if (qout && c->k_norm_in)  // wo left x only (below): this launch normalizes and quantizes the row itself
        launch_k(st, R, k_gateup_k_lds<true, false, true>, dim3(hidden_l / 32), dim3(1024), lds, planes_k(c->gate[l]), planes_k(c->up[l]),
                 act_k(c->act_dim, dim), (const unsigned short*)dev->exp_table, c->h, hidden_l, dim / 256, hx, (signed char*)c->act_hid,
                 (float*)(c->act_hid + alh.off_d), (short*)(c->act_hid + alh.off_aux), (signed char*)(c->act_hid + alh.off_p), (const float*)c->x,
                 (const float*)c->rms_ffn[l]->ptr, 1e-5f, (const float*)c->rsums, dim / 32 <= dev->n_cu ? 2 : 1);
      else if (qout)
        launch_k(st, R, k_gateup_k_lds<true>, dim3(hidden_l / 32), dim3(1024), lds, planes_k(c->gate[l]), planes_k(c->up[l]),
                 act_k(c->act_dim, dim), (const unsigned short*)dev->exp_table, c->h, hidden_l, dim / 256, hx, (signed char*)c->act_hid,
                 (float*)(c->act_hid + alh.off_d), (short*)(c->act_hid + alh.off_aux), (signed char*)(c->act_hid + alh.off_p), (const float*)nullptr, (const float*)nullptr, 0.f, (const float*)nullptr, 1);
      else
        launch_k(st, R, k_gateup_k_lds<false>, dim3((hidden_l + 31) / 32), dim3(1024), lds, planes_k(c->gate[l]), planes_k(c->up[l]),
                 act_k(c->act_dim, dim), (const unsigned short*)dev->exp_table, c->h, hidden_l, dim / 256, hx, (signed char*)nullptr,
                 (float*)nullptr, (short*)nullptr, (signed char*)nullptr, (const float*)nullptr, (const float*)nullptr, 0.f, (const float*)nullptr, 1);
    } else {
      launch_k(st, R, k_gateup<FMT>, dim3((hidden_l + 1) / 2), dim3(128), 0, planes_k(c->gate[l]), planes_k(c->up[l]),
               act_k(c->act_dim, dim), (const unsigned short*)dev->exp_table, c->h, hidden_l, dim / BE);
    }
    CH_TRY(P1());
    if (!qin) launch_quantize_act(st, QT, c->h, (size_t)hidden_l, c->act_hid);
    CH_TRY(gemv_out(c->down[l], act_k(c->act_hid, hidden_l), c->h, hidden_l, 4,
                    (const float*)(l + 1 < L ? c->rms_att[l + 1] : c->rms_final)->ptr, g.rms_norm_eps, qout ? 2 : qin ? 1 : 0));
  }
  CH_HIP(dev, hipGetLastError());
  return 0;
}

int enqueue_segment(crabml_hip_llama* c, int seg) {
  if (c->kfused)
    return c->wtype == CRABML_HIP_Q4_K ? enqueue_segment_k<CRABML_HIP_Q4_K>(c, seg) : enqueue_segment_k<CRABML_HIP_Q4_1>(c, seg);
  if (c->generic) return enqueue_segment_generic(c, seg);
  return c->wtype == CRABML_HIP_Q4_0   ? enqueue_segment_t<CRABML_HIP_Q4_0>(c, seg)
         : c->wtype == CRABML_HIP_Q8_0 ? enqueue_segment_t<CRABML_HIP_Q8_0>(c, seg)
                                       : enqueue_segment_t<CRABML_HIP_Q4_1>(c, seg);
}

TpP2P p2p_view(const crabml_hip_tp_comm* m) {
  TpP2P t{};
  if (m && m->p2p) {
    for (int i = 0; i < 8; i++) t.peer[i] = (unsigned long long*)m->peer[i];
    t.n = m->nranks;
    t.me = m->rank;
    t.cap = m->cap;
    t.fault = m->fault;
  }
  return t;
}

int allreduce(crabml_hip_llama* c, int seg) {
  crabml_hip_device* dev = c->dev;
  if (c->tp_dry) return 0;  // timing-only rank: the partial sums are left as they are
  if (c->comm && c->comm->p2p) {  // one-shot P2P all-reduce as its own launch (per-op segment path)
    if (c->norm_epi) return 0;    // fast path: the collective is fused into the wo / ffn_down epilogue
    const int n = (int)c->cfg.embedding_dim;
    k_tp_allreduce<<<(n + 255) / 256, 256, 0, dev->stream>>>(c->partial, n, tp_view(c, true), c->state + 4, n_segments(c), seg, 0u, 0);
    CH_HIP(dev, hipGetLastError());
    return 0;
  }
  Rccl* r = rccl();
  if (!r || !c->comm || !c->comm->nccl) CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "llama tp: no RCCL communicator");
  int rc = r->AllReduce(c->partial, c->partial, c->cfg.embedding_dim, /*ncclFloat32*/ 7, /*ncclSum*/ 0, c->comm->nccl, dev->stream);
  if (rc != 0) CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "ncclAllReduce failed: %s", r->GetErrorString ? r->GetErrorString(rc) : "?");
  return 0;
}

int enqueue_step(crabml_hip_llama* c) {
  const int n = n_segments(c);
  for (int s = 0; s < n; s++) {
    CH_TRY(enqueue_segment(c, s));
    if (c->tp > 1 && s + 1 < n) CH_TRY(allreduce(c, s));
  }
  return 0;
}

// one decode step at cache position `pos` (the host tracks it; the kernels read their own copy from device memory)
int run_step(crabml_hip_llama* c, size_t pos) {
  const int variant = variant_of(c, pos);
  if (c->use_graph && c->exec[variant]) {
    CH_HIP(c->dev, hipGraphLaunch(c->exec[variant], c->dev->stream));
    return 0;
  }
  c->attn_variant = variant;
  return enqueue_step(c);
}


// ---- batched prefill ---------------------------------------------------------------------------------------
// B prompt rows at positions pos0 .. pos0 + B - 1 through every layer as (B, k) matmul_vec calls (launch_gemv: MFMA
// GEMM for Q4_0 / Q8_0 and B >= 16), row-wise rmsnorm / quantize / rope / append, and causal attention (row r sees
// pos0 + r + 1 cached positions).  Per row this is the arithmetic of the per-op segment path.
int prefill_alloc(crabml_hip_llama* c, size_t cap) {
  if (c->pf_cap >= cap) return 0;
  if (c->pf_cap != 0) CH_BAIL(c->dev, CRABML_HIP_UNEXPECTED, "llama prefill: row buffers already sized for %zu rows", c->pf_cap);
  const auto& g = c->cfg;
  const size_t dim = g.embedding_dim, kv_dim = (size_t)c->kv_dim_l, hidden = g.hidden_dim;
  auto A = [&](size_t bytes, void** out) { return dalloc(c, bytes ? bytes : 16, out); };
  auto act_bytes = [](uint32_t t, size_t n) { return t == CRABML_HIP_F32 ? (size_t)16 : act_layout(t, n).total; };
  CH_TRY(A(cap * 4, (void**)&c->pf_tokens));
  CH_TRY(A(cap * dim * 4, (void**)&c->pf_x));
  CH_TRY(A(cap * dim * 4, (void**)&c->pf_xn));
  CH_TRY(A(cap * dim * 4, (void**)&c->pf_q));
  CH_TRY(A(cap * kv_dim * 4, (void**)&c->pf_k));
  CH_TRY(A(cap * kv_dim * 4, (void**)&c->pf_v));
  CH_TRY(A(cap * dim * 4, (void**)&c->pf_qr));
  CH_TRY(A(cap * dim * 4, (void**)&c->pf_attn));
  CH_TRY(A(cap * dim * 4, (void**)&c->pf_tmp));
  CH_TRY(A(cap * hidden * 4, (void**)&c->pf_g));
  CH_TRY(A(cap * hidden * 4, (void**)&c->pf_u));
  CH_TRY(A(cap * act_bytes(c->qt, dim), (void**)&c->pf_act_dim));
  CH_TRY(A(cap * act_bytes(c->qt, hidden), (void**)&c->pf_act_hid));
  if ((c->qt == CRABML_HIP_Q8_0 || c->qt == CRABML_HIP_Q8_1 || c->qt == CRABML_HIP_Q8_K) && !c->dev->strict_order) {  // (whole column tiles + the look-ahead's slack)
    const size_t xb = gemm_f16w_xh_bytes(cap, dim > hidden ? dim : hidden);
    CH_TRY(A(xb, &c->pf_xh));
    CH_TRY(A(xb, &c->pf_xh2));
    CH_HIP(c->dev, hipMemsetAsync(c->pf_xh2, 0, xb, c->dev->stream));
    // (the widest split launch is q | k | v; up to 7 partial buffers of a short pass, 3 of a full one)
    c->pf_split_floats = (cap + 1024) * (dim + 2 * kv_dim > hidden ? dim + 2 * kv_dim : hidden);
    CH_TRY(A(c->pf_split_floats * 4, (void**)&c->pf_split));
    CH_HIP(c->dev, hipMemsetAsync(c->pf_xh, 0, xb, c->dev->stream));
  }
  c->pf_cap = cap;
  return 0;
}


// the row-tiled causal attention of a prefill pass; false = not covered (the caller launches k_attn per (head, row))
template <bool KV16, int G, int R>
bool launch_attn_tile_t(crabml_hip_llama* c, int l, int B, int pos0) {
  const int hd = c->hd, n_heads = c->n_heads_l, n_kv = c->n_kv_l, seq_cap = (int)c->cfg.seq_len;
  const int sstride = (pos0 + B + 3) & ~3;
  const size_t lds = (size_t)(G * R) * (size_t)(hd + sstride) * sizeof(float);
  if (lds > 64 * 1024) return false;
  k_attn_tile<KV16, G, R><<<dim3(n_kv, (B + R - 1) / R), 256, lds, c->dev->stream>>>(
      c->pf_qr, c->kc[l], c->vc[l], c->state + 6, (const unsigned short*)c->dev->exp_table, c->pf_attn, n_heads, n_kv, hd, seq_cap, B,
      sstride);
  return true;
}
bool launch_attn_tile(crabml_hip_llama* c, int l, int B, int pos0) {
  const int hd = c->hd, g = c->n_heads_l / c->n_kv_l;
  const bool kv16 = c->cfg.use_f16_kv_cache != 0;
  if (c->cfg.flags & CRABML_HIP_LLAMA_NO_TILE_ATTENTION) return false;
  if (pos0 + B > 1024 || hd > (kv16 ? 256 : 128) || hd % (kv16 ? 16 : 4) != 0 || c->n_heads_l % c->n_kv_l != 0) return false;
  if (kv16) {
    switch (g) {
      case 1: return launch_attn_tile_t<true, 1, 4>(c, l, B, pos0);
      case 2: return launch_attn_tile_t<true, 2, 4>(c, l, B, pos0);
      case 4: return launch_attn_tile_t<true, 4, 4>(c, l, B, pos0);
      case 8: return launch_attn_tile_t<true, 8, 2>(c, l, B, pos0);
      default: return false;
    }
  }
  switch (g) {
    case 1: return launch_attn_tile_t<false, 1, 4>(c, l, B, pos0);
    case 2: return launch_attn_tile_t<false, 2, 4>(c, l, B, pos0);
    case 4: return launch_attn_tile_t<false, 4, 4>(c, l, B, pos0);
    case 8: return launch_attn_tile_t<false, 8, 2>(c, l, B, pos0);
    default: return false;
  }
}

// prompts past 1024 positions: the three long-context kernels with a row dimension (grid.y), PF_LONG_ROWS rows at a time
// (score / probability scratch: rows x n_heads x seq_len x 6 bytes)
constexpr int PF_LONG_ROWS = 64;
template <int G>
int launch_attn_long_rows_t(crabml_hip_llama* c, int l, int B) {
  crabml_hip_device* dev = c->dev;
  hipStream_t st = dev->stream;
  const int hd = c->hd, seq_cap = (int)c->cfg.seq_len, n_kv = c->n_kv_l, n_heads = c->n_heads_l;
  const int* pos_d = c->state + 6;
  const int ts = 256 / G, nsplit = (seq_cap + ts - 1) / ts;
  if (!c->pf_scores) {
    CH_TRY(dalloc(c, (size_t)PF_LONG_ROWS * n_heads * seq_cap * 4, (void**)&c->pf_scores));
    CH_TRY(dalloc(c, (size_t)PF_LONG_ROWS * n_heads * seq_cap * 2, (void**)&c->pf_p16));
  }
  for (int r0 = 0; r0 < B; r0 += PF_LONG_ROWS) {
    const unsigned rows = (unsigned)(B - r0 < PF_LONG_ROWS ? B - r0 : PF_LONG_ROWS);
    k_attn_scores<G><<<dim3(n_kv * nsplit, rows), 256, (size_t)G * hd * sizeof(float), st>>>(
        (const float*)c->pf_qr, (const unsigned short*)c->kc[l], pos_d, c->pf_scores, n_kv, hd, seq_cap, nsplit, r0);
    k_attn_softmax<4><<<dim3(n_heads, rows), 256, (size_t)seq_cap * sizeof(float), st>>>(
        (const float*)c->pf_scores, pos_d, (const unsigned short*)dev->exp_table, c->pf_p16, seq_cap, r0, dev->strict_order ? 1 : 0);
    // PV for R prompt rows per workgroup (one V fetch for R x G chains); G = 8 fills the lanes with two rows
    constexpr int PR = G == 8 ? 2 : 4;
    if (c->cfg.flags & CRABML_HIP_LLAMA_NO_PV_ROW_TILES)
      k_attn_pv<G><<<dim3(n_kv * (hd / 32), rows), 256, 0, st>>>((const unsigned short*)c->pf_p16, (const unsigned short*)c->vc[l], pos_d,
                                                                 c->pf_attn, nullptr, nullptr, nullptr, hd, seq_cap, 0, r0);
    else
      k_attn_pv_rows<G, PR><<<dim3(n_kv * (hd / 32), (rows + PR - 1) / PR), 256, 0, st>>>(
          (const unsigned short*)c->pf_p16, (const unsigned short*)c->vc[l], pos_d, c->pf_attn, hd, seq_cap, r0, (int)rows);
  }
  return 0;
}
// 1 = launched, 0 = not covered, < 0 = error
int launch_attn_long_rows(crabml_hip_llama* c, int l, int B) {
  if (!c->exact_long_ok || (c->cfg.flags & CRABML_HIP_LLAMA_NO_TILE_ATTENTION)) return 0;
  int rc;
  switch (c->n_heads_l / c->n_kv_l) {
    case 1: rc = launch_attn_long_rows_t<1>(c, l, B); break;
    case 2: rc = launch_attn_long_rows_t<2>(c, l, B); break;
    case 4: rc = launch_attn_long_rows_t<4>(c, l, B); break;
    case 8: rc = launch_attn_long_rows_t<8>(c, l, B); break;
    default: return 0;
  }
  return rc == 0 ? 1 : -1;
}

int prefill_chunk(crabml_hip_llama* c, const uint32_t* tokens, size_t B, size_t pos0, bool want_logits) {
  crabml_hip_device* dev = c->dev;
  hipStream_t st = dev->stream;
  const auto& g = c->cfg;
  const int dim = (int)g.embedding_dim, kv_dim = c->kv_dim_l, hidden = (int)g.hidden_dim, hd = c->hd, seq_cap = (int)g.seq_len;
  const int n_heads = c->n_heads_l, n_kv = c->n_kv_l, L = (int)g.n_layers;
  const bool kv16 = g.use_f16_kv_cache != 0, strict = dev->strict_order;
  const int half = strict ? 0 : 1;
  const unsigned rows = (unsigned)B;
  {
    std::vector<int> h(B + 1);
    for (size_t i = 0; i < B; i++) h[i] = (int)tokens[i];
    CH_HIP(dev, hipMemcpyAsync(c->pf_tokens, h.data(), B * sizeof(int), hipMemcpyHostToDevice, st));
    const int p0 = (int)pos0;
    CH_HIP(dev, hipMemcpyAsync(c->state + 6, &p0, sizeof(int), hipMemcpyHostToDevice, st));
    CH_HIP(dev, hipStreamSynchronize(st));  // the staging vectors go out of scope
  }
  const int* pos_d = c->state + 6;
  const size_t norm_lds = norm_lds_bytes(dim);
  auto norm_rows = [&](const float* wn, float eps) {
    if (dim <= 4096)
      k_norm_f32_rows<4><<<rows, 1024, norm_lds, st>>>(c->pf_x, wn, dim, eps, c->pf_xn, half);
    else
      k_norm_f32_rows<12><<<rows, 1024, norm_lds, st>>>(c->pf_x, wn, dim, eps, c->pf_xn, half);
  };
  static const bool gemm_exact_hook = [] {  // A/B hook (CRABML_HIP_TEST_HOOKS=1 CRABML_HIP_GEMM_EXACT=1): the fast pass with matmul_vec's own scaling
    const char* h = getenv("CRABML_HIP_TEST_HOOKS");
    const char* e = getenv("CRABML_HIP_GEMM_EXACT");
    return h && h[0] == '1' && e && e[0] == '1';
  }();
  // The fast pass, Q4_0 / Q8_0 weights x Q8_0 rows, Q4_1 x Q8_1, Q4_K / Q6_K x Q8_K, >= 32 rows: the weight-stationary f16 GEMM
  // (gemm_f16w.hip; block scales folded into f16 operands, f32 accumulation inside the matrix core -- a stated deviation of the fast
  // tier).  The rows' pre-scaled f16 planes are made once per rhs and k-slot order (q / k / v and gate / up share theirs): xh_of /
  // xh_order remember what pf_xh currently holds.
  static const bool f16w_off = [] {  // A/B hook (CRABML_HIP_TEST_HOOKS=1 CRABML_HIP_GEMM_INT8=1): the int8 kernels in the fast pass too
    const char* h = getenv("CRABML_HIP_TEST_HOOKS");
    const char* e = getenv("CRABML_HIP_GEMM_INT8");
    return h && h[0] == '1' && e && e[0] == '1';
  }();
  static const int f16w_min = [] {  // lab hook (CRABML_HIP_TEST_HOOKS=1 CRABML_HIP_F16W_MIN=rows): the smallest pass that takes it
    const char* h = getenv("CRABML_HIP_TEST_HOOKS");
    const char* e = getenv("CRABML_HIP_F16W_MIN");
    return h && h[0] == '1' && e ? atoi(e) : 32;
  }();
  const bool f16w = !strict && !gemm_exact_hook && !f16w_off && !(g.flags & CRABML_HIP_LLAMA_PREFILL_INT8_GEMM) &&
                    (c->qt == CRABML_HIP_Q8_0 || c->qt == CRABML_HIP_Q8_1 || c->qt == CRABML_HIP_Q8_K) && c->pf_xh != nullptr && B >= f16w_min;  // (shorter passes: the int8 kernels / the GEMV)
  // CpuTensorBuf::quantize for the rhs of matmul_vec (buf/api.rs:142-159): F32 weights take the rows as they are
  const void* xh_of = nullptr;  // the planes c->pf_xh was made from (reset whenever planes are rewritten) ...
  int xh_order = -1;            // ... and the k-slot order it is in (gemm_f16w_order of the weight format)
  auto rows_to_f16 = [&](const crabml_hip_buf* w, const void* act, int k) {
    const int order = gemm_f16w_order(w->dtype);
    if (xh_of == act && xh_order == order) return;
    launch_rows_to_f16(st, c->qt, w->dtype, act, B, (size_t)k, c->pf_xh);
    xh_of = act;
    xh_order = order;
  };
  // ... written by the kernel that quantizes the rows when the GEMM that reads them next is the f16 one (f16w_rows.hpp: the same
  // bits as k_rows_to_f16 from the finished planes, one launch fewer per GEMM; A/B: CRABML_HIP_LLAMA_PREFILL_SEPARATE_F16_ROWS)
  auto xh_target = [&](const crabml_hip_buf* next, int k, int* order) -> void* {
    if (!f16w || next == nullptr || (g.flags & CRABML_HIP_LLAMA_PREFILL_SEPARATE_F16_ROWS) || !gemm_f16w_covers(next->dtype, c->qt) ||
        (c->qt == CRABML_HIP_Q8_K && k % 256 != 0))
      return nullptr;
    *order = gemm_f16w_order(next->dtype);
    return c->pf_xh;
  };
  // next: the weight matrix whose GEMM reads these planes first
  auto quant_rows = [&](const float* src, int n, char* planes, const crabml_hip_buf* next) -> const void* {
    if (c->qt == CRABML_HIP_F32) return src;
    int order = 0;
    void* xh = xh_target(next, n, &order);
    launch_quantize_act_rows(st, c->qt, src, B, (size_t)n, planes, xh, order);
    xh_of = xh ? planes : nullptr;
    xh_order = order;
    return planes;
  };
  // defer (nullable): a GEMM cut into k pieces may leave the sum of its pieces to the row kernel that consumes `out` (pf_split holds them)
  auto gemm = [&](const crabml_hip_buf* w, int m, int k, const void* act, float* out, int* defer = nullptr) -> int {
    if (defer) *defer = 0;
    if (g.flags & CRABML_HIP_LLAMA_PREFILL_SEPARATE_F16_ROWS) defer = nullptr;  // (A/B: every reduce its own launch)
    if (f16w && gemm_f16w_covers(w->dtype, c->qt) && (c->qt != CRABML_HIP_Q8_K || k % 256 == 0)) {
      rows_to_f16(w, act, k);
      const size_t mm = (size_t)m;
      if (launch_gemm_f16w(dev, &w, &mm, 1, (size_t)k, c->pf_xh, B, &out, c->pf_split, c->pf_split_floats, nullptr, nullptr, defer)) return 0;
    }
    if (!strict) {
      return launch_gemv(dev, w, m, k, act, B, out, nullptr, !gemm_exact_hook);
    }
    // strict order: the Q4_0 / Q8_0 / Q4_1 MFMA GEMM scales its exact integer tiles block by block in the reference's scalar
    // order, i.e. it IS the strict result (bit for bit) -- the other formats take the scalar-order GEMV row by row
    if ((w->dtype == CRABML_HIP_Q4_0 || w->dtype == CRABML_HIP_Q8_0 || w->dtype == CRABML_HIP_Q4_1 || w->dtype == CRABML_HIP_Q8_K) && B >= 16 &&
        launch_gemm_mfma(dev, w, m, k, act, B, out, nullptr))
      return 0;
    return launch_gemv_strict(dev, w, m, k, act, B, out);
  };
  k_embed<<<dim3((dim + 255) / 256, rows), 256, 0, st>>>((const char*)c->token_embed->ptr, (int)c->token_embed->dtype,
                                                         c->token_embed->wl.off_scale, c->pf_tokens, dim, c->pf_x);
  // Q8_0 / Q8_1 rhs: residual add + RMSNorm + quantize as one launch per row (k_norm_quant_rows), SiLU * mul + quantize as one
  // (k_gateup_epi_quant): the (rows, dim) / (rows, hidden) f32 intermediates make one trip through memory instead of three
  const bool fuse_rows = (c->qt == CRABML_HIP_Q8_0 || c->qt == CRABML_HIP_Q8_1) && !(g.flags & CRABML_HIP_LLAMA_NO_PREFILL_ROW_FUSION);
  // Q8_K rhs (K-quant layers): the same for residual add + RMSNorm + quantize (k_norm_quant_rows_k); SiLU * mul keeps its own launch
  // (from 192 rows: one 1024-thread workgroup per row is a chain of four barriers -- below, the four small launches run 1-2 % faster)
  const bool fuse_k = c->qt == CRABML_HIP_Q8_K && dim % 256 == 0 && B >= 192 && !(g.flags & CRABML_HIP_LLAMA_NO_PREFILL_ROW_FUSION);
  const bool fuse_norm = fuse_rows || fuse_k;
  const ActLayout ald = act_layout(c->qt == CRABML_HIP_F32 ? CRABML_HIP_Q8_0 : c->qt, (size_t)dim);
  const ActLayout alh = act_layout(c->qt == CRABML_HIP_F32 ? CRABML_HIP_Q8_0 : c->qt, (size_t)hidden);
  // pending = the wo / ffn_down output that has not been added to x yet (folded into the next norm)
  // nparts: `pending` is piece 0 of a GEMM cut into k pieces, the others wait in pf_split (gemm's defer)
  auto norm_quant_rows = [&](const float* wn, float eps, float* pending, const crabml_hip_buf* next, int nparts = 0) -> const void* {
    const bool q81 = c->qt == CRABML_HIP_Q8_1;
    int order = 0;
    unsigned short* xh = (unsigned short*)xh_target(next, dim, &order);
    const size_t pstride = B * (size_t)dim;
    // Q8_0 / Q8_1 rows of 4096 / 8192 elements: the 256-thread form (a thread owns half a quant block / a whole one; prefill_rows.hpp)
    static const bool rows_1024 = [] {  // lab hook (CRABML_HIP_TEST_HOOKS=1 CRABML_HIP_NORM_ROWS_1024=1): the 1024-thread kernel
      const char* h = getenv("CRABML_HIP_TEST_HOOKS");
      const char* e = getenv("CRABML_HIP_NORM_ROWS_1024");
      return h && h[0] == '1' && e && e[0] == '1';
    }();
    if (!fuse_k && !rows_1024 && (dim == 4096 || dim == 8192) && !(g.flags & CRABML_HIP_LLAMA_PREFILL_SEPARATE_F16_ROWS)) {
#define CRABML_NQW(E_, Q_)                                                                                                            \
  k_norm_quant_rows_w<E_, Q_><<<rows, 256, 0, st>>>(c->pf_x, pending, wn, dim, eps, c->pf_act_dim, ald.total, ald.off_d, ald.off_aux, \
                                                    half, xh, c->pf_split, pstride, nparts)
      if (dim == 4096) {
        if (q81)
          CRABML_NQW(16, true);
        else
          CRABML_NQW(16, false);
      } else {
        if (q81)
          CRABML_NQW(32, true);
        else
          CRABML_NQW(32, false);
      }
#undef CRABML_NQW
      xh_of = xh ? c->pf_act_dim : nullptr;
      xh_order = order;
      return c->pf_act_dim;
    }
    if (fuse_k) {
      if (dim <= 4096)
        k_norm_quant_rows_k<4><<<rows, 1024, norm_lds, st>>>(c->pf_x, pending, wn, dim, eps, c->pf_xn, c->pf_act_dim, ald.total, ald.off_d,
                                                            ald.off_aux, ald.off_p, half, xh, order, c->pf_split, pstride, nparts);
      else
        k_norm_quant_rows_k<12><<<rows, 1024, norm_lds, st>>>(c->pf_x, pending, wn, dim, eps, c->pf_xn, c->pf_act_dim, ald.total, ald.off_d,
                                                             ald.off_aux, ald.off_p, half, xh, order, c->pf_split, pstride, nparts);
      xh_of = xh ? c->pf_act_dim : nullptr;
      xh_order = order;
      return c->pf_act_dim;
    }
#define CRABML_NQR(NIT_, Q_)                                                                                                         \
  if (xh || nparts > 0)                                                                                                              \
    k_norm_quant_rows_h<NIT_, Q_><<<rows, 1024, norm_lds, st>>>(c->pf_x, pending, wn, dim, eps, c->pf_act_dim, ald.total, ald.off_d,  \
                                                                ald.off_aux, half, xh, c->pf_split, pstride, nparts);                \
  else                                                                                                                               \
    k_norm_quant_rows<NIT_, Q_><<<rows, 1024, norm_lds, st>>>(c->pf_x, pending, wn, dim, eps, c->pf_act_dim, ald.total, ald.off_d,    \
                                                              ald.off_aux, half)
    if (dim <= 4096) {
      if (q81)
        CRABML_NQR(4, true);
      else
        CRABML_NQR(4, false);
    } else {
      if (q81)
        CRABML_NQR(12, true);
      else
        CRABML_NQR(12, false);
    }
#undef CRABML_NQR
    xh_of = xh ? c->pf_act_dim : nullptr;
    xh_order = order;
    return c->pf_act_dim;
  };
  bool pending_down = false;  // (fuse_rows) the previous layer's ffn_down output sits in pf_tmp, not yet added to pf_x
  int down_parts = 0;         // ... as piece 0 of this many + 1 k pieces
  for (int l = 0; l < L; l++) {
    const void* a;
    if (fuse_norm) {
      a = norm_quant_rows((const float*)c->rms_att[l]->ptr, g.rms_norm_eps, pending_down ? c->pf_tmp : nullptr, c->wq[l],
                          pending_down ? down_parts : 0);
      pending_down = false;
    } else {
      norm_rows((const float*)c->rms_att[l]->ptr, g.rms_norm_eps);  // llama2.rs:230-234
      a = quant_rows(c->pf_xn, dim, c->pf_act_dim, c->wq[l]);
    }
    bool qkv_done = false;  // llama2.rs:244-246
    if (f16w && gemm_f16w_covers(c->wq[l]->dtype, c->qt) && c->wk[l]->dtype == c->wq[l]->dtype && c->wv[l]->dtype == c->wq[l]->dtype) {
      // the three GEMMs of the same rhs as ONE launch (the 1024-row k / v matrices alone leave most of the chip idle)
      rows_to_f16(c->wq[l], a, dim);
      const crabml_hip_buf* ws[3] = {c->wq[l], c->wk[l], c->wv[l]};
      const size_t ms[3] = {(size_t)dim, (size_t)kv_dim, (size_t)kv_dim};
      float* outs[3] = {c->pf_q, c->pf_k, c->pf_v};
      qkv_done = launch_gemm_f16w(dev, ws, ms, 3, (size_t)dim, c->pf_xh, B, outs, c->pf_split, c->pf_split_floats);
    }
    if (!qkv_done) {
      CH_TRY(gemm(c->wq[l], dim, dim, a, c->pf_q));
      CH_TRY(gemm(c->wk[l], kv_dim, dim, a, c->pf_k));
      CH_TRY(gemm(c->wv[l], kv_dim, dim, a, c->pf_v));
    }
    QkvEpi e{c->pf_qr, c->kc[l], c->vc[l], c->rope, pos_d, 1.0f / std::sqrt((float)hd), dim, kv_dim, hd,
             (int)g.rope_dim, c->npairs, seq_cap, kv16 ? 1 : 0};
    const int pairs = (dim + 2 * kv_dim) / 2;
    k_qkv_epi_rows<<<dim3((pairs + 255) / 256, rows), 256, 0, st>>>(c->pf_q, c->pf_k, c->pf_v, e);
    int along = 0;
    // (a pass whose every row sees fewer cached positions than the decode step's switch to the f32 kernels -- attn_long_from --
    // keeps the exact tile kernel, so that prefill(prompt) and a token loop over the same short prompt agree bit for bit)
    if (c->attn_flash_rows && kv16 && pos0 + B >= c->attn_long_from) {
      // fast step: causal flash attention on the f16 matrix cores (k_attn_flash_rows; the deviation stated for k_attn_flash)
      const dim3 fg((unsigned)((B + 63) / 64), (unsigned)n_heads);
      if (hd == 128)
        k_attn_flash_rows<128><<<fg, 512, flash_rows_lds_bytes(128), st>>>((const float*)c->pf_qr, (const unsigned short*)c->kc[l], (const unsigned short*)c->vc[l], pos_d,
                                                   c->pf_attn, n_heads, n_kv, seq_cap, (int)B);
      else
        k_attn_flash_rows<64><<<fg, 512, flash_rows_lds_bytes(64), st>>>((const float*)c->pf_qr, (const unsigned short*)c->kc[l], (const unsigned short*)c->vc[l], pos_d,
                                                  c->pf_attn, n_heads, n_kv, seq_cap, (int)B);
      along = 1;
    } else if (!launch_attn_tile(c, l, (int)B, (int)pos0)) {
      along = launch_attn_long_rows(c, l, (int)B);  // past 1024 positions: the long-context kernels, rows in grid.y
      if (along < 0) return CRABML_HIP_UNEXPECTED;
    } else {
      along = 1;
    }
    if (!along) {  // unusual shapes (f32 cache past 1024 positions, odd group sizes): one workgroup per (head, row)
      const size_t attn_lds = (size_t)(seq_cap + hd) * sizeof(float);
      if (kv16)
        k_attn<true><<<dim3(n_heads, rows), 256, attn_lds, st>>>(c->pf_qr, c->kc[l], c->vc[l], pos_d, (const unsigned short*)dev->exp_table,
                                                                  c->pf_attn, nullptr, nullptr, nullptr, n_heads, n_kv, hd, seq_cap,
                                                                  PrefetchPlan{}, dev->strict_order ? 256 : 0);
      else
        k_attn<false><<<dim3(n_heads, rows), 256, attn_lds, st>>>(c->pf_qr, c->kc[l], c->vc[l], pos_d, (const unsigned short*)dev->exp_table,
                                                                   c->pf_attn, nullptr, nullptr, nullptr, n_heads, n_kv, hd, seq_cap,
                                                                   PrefetchPlan{}, dev->strict_order ? 256 : 0);
    }
    a = quant_rows(c->pf_attn, dim, c->pf_act_dim, c->wo[l]);
    int wo_parts = 0;
    CH_TRY(gemm(c->wo[l], dim, dim, a, c->pf_tmp, fuse_norm ? &wo_parts : nullptr));  // llama2.rs:600
    if (fuse_norm) {
      a = norm_quant_rows((const float*)c->rms_ffn[l]->ptr, 1e-5f, c->pf_tmp, c->gate[l], wo_parts);  // x += wo out (:266), FFN norm (:611), quantize
    } else {
      k_res_epi<<<(unsigned)(((size_t)B * dim + 255) / 256), 256, 0, st>>>(c->pf_tmp, c->pf_x, (int)(B * dim), 1);  // :266
      norm_rows((const float*)c->rms_ffn[l]->ptr, 1e-5f);  // llama2.rs:611
      a = quant_rows(c->pf_xn, dim, c->pf_act_dim, c->gate[l]);
    }
    bool gu_done = false;  // llama2.rs:620-630
    int h_done = 0;        // the launch stored h = silu(g) * u (pf_g) instead of g and u
    if (f16w && gemm_f16w_covers(c->gate[l]->dtype, c->qt) && c->up[l]->dtype == c->gate[l]->dtype) {
      // gate and up as ONE launch: 2 x 448 workgroups fill the last round of the chip better than 448 twice -- and, where 64-row tiles
      // of both cover the chip, with SiLU * mul as the epilogue (a wave holds the same 16 rows of both matrices)
      rows_to_f16(c->gate[l], a, dim);
      const crabml_hip_buf* ws[2] = {c->gate[l], c->up[l]};
      const size_t ms[2] = {(size_t)hidden, (size_t)hidden};
      float* outs[2] = {c->pf_g, c->pf_u};
      const bool epi = !(g.flags & CRABML_HIP_LLAMA_PREFILL_NO_GU_EPILOGUE);
      // ... and, for Q8_0 / Q8_1 rows, with the row quantizer behind it (h leaves as ffn_down's planes; its f16 planes go to pf_xh2:
      // pf_xh is this launch's own rhs)
      F16wHQuant hq{};
      int hq_order = 0;
      if (epi && fuse_rows && !(g.flags & CRABML_HIP_LLAMA_PREFILL_SEPARATE_F16_ROWS)) {
        hq.planes = c->pf_act_hid;
        hq.stride = alh.total;
        hq.off_d = alh.off_d;
        hq.off_aux = alh.off_aux;
        hq.q81 = c->qt == CRABML_HIP_Q8_1;
        hq.xh = xh_target(c->down[l], hidden, &hq_order) ? (unsigned short*)c->pf_xh2 : nullptr;
      }
      gu_done = launch_gemm_f16w(dev, ws, ms, 2, (size_t)dim, c->pf_xh, B, outs, c->pf_split, c->pf_split_floats,
                                 epi ? (const unsigned short*)dev->exp_table : nullptr, epi ? &h_done : nullptr, nullptr, &hq);
      if (h_done == 2) {
        a = c->pf_act_hid;
        if (hq.xh) std::swap(c->pf_xh, c->pf_xh2);  // (ffn_down's GEMM reads what this launch wrote)
        xh_of = hq.xh ? (const void*)c->pf_act_hid : nullptr;
        xh_order = hq_order;
      }
    }
    if (!gu_done) {
      CH_TRY(gemm(c->gate[l], hidden, dim, a, c->pf_g));
      CH_TRY(gemm(c->up[l], hidden, dim, a, c->pf_u));
    }
    if (h_done == 2) {
      // (quantized by the launch itself)
    } else if (h_done) {  // h sits in pf_g: quantize it (the quantizer launch's arithmetic is quant_lane32's, bit for bit)
      a = quant_rows(c->pf_g, hidden, c->pf_act_hid, c->down[l]);
    } else if (fuse_rows) {
      const dim3 gq((unsigned)((hidden + 255) / 256), rows);
      int order = 0;
      unsigned short* xh = (unsigned short*)xh_target(c->down[l], hidden, &order);
      if (c->qt == CRABML_HIP_Q8_1 && xh)
        k_gateup_epi_quant_h<true><<<gq, 256, 0, st>>>(c->pf_g, c->pf_u, (const unsigned short*)dev->exp_table, hidden, c->pf_act_hid, alh.total,
                                                       alh.off_d, alh.off_aux, xh);
      else if (xh)
        k_gateup_epi_quant_h<false><<<gq, 256, 0, st>>>(c->pf_g, c->pf_u, (const unsigned short*)dev->exp_table, hidden, c->pf_act_hid, alh.total,
                                                        alh.off_d, alh.off_aux, xh);
      else if (c->qt == CRABML_HIP_Q8_1)
        k_gateup_epi_quant<true><<<gq, 256, 0, st>>>(c->pf_g, c->pf_u, (const unsigned short*)dev->exp_table, hidden, c->pf_act_hid, alh.total,
                                                     alh.off_d, alh.off_aux);
      else
        k_gateup_epi_quant<false><<<gq, 256, 0, st>>>(c->pf_g, c->pf_u, (const unsigned short*)dev->exp_table, hidden, c->pf_act_hid, alh.total,
                                                      alh.off_d, alh.off_aux);
      xh_of = xh ? c->pf_act_hid : nullptr;
      xh_order = order;
      a = c->pf_act_hid;
    } else {
      k_gateup_epi<<<(unsigned)(((size_t)B * hidden + 255) / 256), 256, 0, st>>>(c->pf_g, c->pf_u, (const unsigned short*)dev->exp_table,
                                                                                 c->pf_g, (int)(B * hidden));
      a = quant_rows(c->pf_g, hidden, c->pf_act_hid, c->down[l]);
    }
    CH_TRY(gemm(c->down[l], dim, hidden, a, c->pf_tmp, fuse_norm && l + 1 < L ? &down_parts : nullptr));  // llama2.rs:633-636
    if (fuse_norm && l + 1 < L)
      pending_down = true;  // added by the next layer's norm launch
    else
      k_res_epi<<<(unsigned)(((size_t)B * dim + 255) / 256), 256, 0, st>>>(c->pf_tmp, c->pf_x, (int)(B * dim), 1);
  }
  if (want_logits) {  // final rmsnorm + classifier of the last row only (llama2.rs:274-278, 199-208)
    CH_HIP(dev, hipMemcpyAsync(c->x, c->pf_x + (B - 1) * (size_t)dim, (size_t)dim * 4, hipMemcpyDeviceToDevice, st));
    if (dim <= 4096)
      k_norm_f32<4><<<1, 1024, norm_lds, st>>>(c->x, nullptr, (const float*)c->rms_final->ptr, dim, g.rms_norm_eps, c->xn, half);
    else
      k_norm_f32<12><<<1, 1024, norm_lds, st>>>(c->x, nullptr, (const float*)c->rms_final->ptr, dim, g.rms_norm_eps, c->xn, half);
    const void* act = c->xn;
    if (c->out_qt != CRABML_HIP_F32) {
      launch_quantize_act(st, c->out_qt, c->xn, (size_t)dim, c->act_dim);
      act = c->act_dim;
    }
    CH_TRY(strict ? launch_gemv_strict(dev, c->output, g.vocab_size, dim, act, 1, c->logits)
                  : launch_gemv(dev, c->output, g.vocab_size, dim, act, 1, c->logits, nullptr));
  }
  CH_HIP(dev, hipGetLastError());
  return 0;
}

int set_state(crabml_hip_llama* c, size_t token, size_t pos, int step, const unsigned* serial = nullptr) {
  if (c->h_state_next == crabml_hip_llama::H_STATE_SLOTS) {  // every slot may still be waiting for its copy: drain, start over
    CH_HIP(c->dev, hipStreamSynchronize(c->dev->stream));
    c->h_state_next = 0;
  }
  int* st = c->h_state + 4 * c->h_state_next++;
  st[0] = (int)token;
  st[1] = (int)pos;
  st[2] = step;
  if (serial != nullptr) {  // token, pos, step, (prefetch sink), serial: the serial set from the host (lazy_ctx_begin)
    int st5[5] = {st[0], st[1], st[2], 0, (int)*serial};
    // the ring slot holds 4 ints: the 5-int form takes two consecutive slots
    if (c->h_state_next == crabml_hip_llama::H_STATE_SLOTS) {
      CH_HIP(c->dev, hipStreamSynchronize(c->dev->stream));
      c->h_state_next = 1;
      st = c->h_state;
    }
    c->h_state_next++;
    memcpy(st, st5, sizeof st5);
    CH_HIP(c->dev, hipMemcpyAsync(c->state, st, 5 * sizeof(int), hipMemcpyHostToDevice, c->dev->stream));
    return 0;
  }
  CH_HIP(c->dev, hipMemcpyAsync(c->state, st, 3 * sizeof(int), hipMemcpyHostToDevice, c->dev->stream));
  return 0;
}

}  // namespace

extern "C" {

// ---- tensor-parallel communicator (RCCL) ------------------------------------------------------------------
int crabml_hip_tp_get_unique_id(void* id128) {
  Rccl* r = rccl();
  if (!r || !id128) return CRABML_HIP_UNEXPECTED;
  return r->GetUniqueId(id128) == 0 ? 0 : CRABML_HIP_UNEXPECTED;
}

int crabml_hip_tp_comm_create(crabml_hip_device_t* dev, const void* id128, int nranks, int rank, crabml_hip_tp_comm_t** out) {
  if (!dev || !id128 || !out || nranks < 1 || rank < 0 || rank >= nranks) return CRABML_HIP_BAD_INPUT;
  *out = nullptr;
  Rccl* r = rccl();
  if (!r) CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "librccl.so could not be loaded");
  CH_LIVE(dev);
  CH_USE(dev);
  CH_FLUSH(dev);
  Rccl::IdT id;
  memcpy(&id, id128, sizeof id);
  void* comm = nullptr;
  int rc = r->CommInitRank(&comm, nranks, id, rank);
  if (rc != 0) CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "ncclCommInitRank failed: %s", r->GetErrorString ? r->GetErrorString(rc) : "?");
  crabml_hip_tp_comm* c = new crabml_hip_tp_comm();
  c->dev = dev;
  c->nccl = comm;
  c->nranks = nranks;
  c->rank = rank;
  *out = c;
  return 0;
}

int crabml_hip_tp_comm_destroy(crabml_hip_tp_comm_t* comm) {
  if (!comm) return 0;
  if (comm->p2p) {
    (void)hipSetDevice(comm->dev->ordinal);
    (void)hipStreamSynchronize(comm->dev->stream);
    for (int r = 0; r < comm->nranks; r++) {
      if (r == comm->rank || !comm->peer[r]) continue;
      // mapped through IPC (another process): close the mapping; same-process peers (connect_local) are plain pointers owned
      // by their rank and are left alone.  Destroy order: contexts before their group, every rank quiesced.
      if (comm->via_ipc[r]) (void)hipIpcCloseMemHandle(comm->peer[r]);
      (void)hipGetLastError();
    }
    if (comm->inbox) (void)hipFree(comm->inbox);
    delete comm;
    return 0;
  }
  Rccl* r = rccl();
  if (r && comm->nccl) r->CommDestroy(comm->nccl);
  delete comm;
  return 0;
}

// in-place sum over ranks of an F32 buffer's first n elements, on the device stream (the collective the decode
// step issues twice per layer); exposed so the RCCL path can be exercised on its own
int crabml_hip_tp_all_reduce(crabml_hip_tp_comm_t* comm, crabml_hip_buf_t* buf, size_t n) {
  if (!comm || !buf) return CRABML_HIP_BAD_INPUT;
  crabml_hip_device* dev = comm->dev;
  CH_LIVE(dev);
  CH_USE(dev);
  CH_FLUSH(dev);
  if (buf->dtype != CRABML_HIP_F32 || n > buf->n_elems) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "tp_all_reduce: needs an f32 buffer of >= n elements");
  lazy_use(dev, buf);
  CH_TRY(ensure_mem(dev, buf));
  if (comm->p2p) {
    if (!comm->connected) CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "tp_all_reduce: the p2p group is not connected");
    if (n > comm->cap) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "tp_all_reduce: %zu elements exceed the inbox rows (%u)", n, comm->cap);
    const unsigned k = comm->host_epoch++;  // every rank issues the same sequence of calls
    k_tp_allreduce<<<(unsigned)((n + 255) / 256), 256, 0, dev->stream>>>((float*)buf->ptr, (int)n, p2p_view(comm), nullptr, 1, (int)(k & 1u),
                                                                       0x40000000u + k, 2);
    CH_HIP(dev, hipGetLastError());
    int fault = 0;
    CH_HIP(dev, hipMemcpyAsync(&fault, comm->fault, sizeof(int), hipMemcpyDeviceToHost, dev->stream));
    CH_HIP(dev, hipStreamSynchronize(dev->stream));
    if (fault) {
      (void)hipMemsetAsync(comm->fault, 0, sizeof(int), dev->stream);  // the group stays usable once the peer shows up
      CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "tp_all_reduce: a peer's partial never arrived (poll timed out)");
    }
    touch(buf);
    return 0;
  }
  Rccl* r = rccl();
  if (!r) CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "librccl.so could not be loaded");
  int rc = r->AllReduce(buf->ptr, buf->ptr, n, 7, 0, comm->nccl, dev->stream);
  if (rc != 0) CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "ncclAllReduce failed: %s", r->GetErrorString ? r->GetErrorString(rc) : "?");
  touch(buf);
  return 0;
}

// ---- one-shot P2P all-reduce group (the production collective of SURVEY.md 8e; device side: fused_ffn.hpp, TpP2P) --------
int crabml_hip_tp_p2p_create(crabml_hip_device_t* dev, int nranks, int rank, size_t max_elems, crabml_hip_tp_comm_t** out) {
  if (!dev || !out || nranks < 1 || nranks > 8 || rank < 0 || rank >= nranks || max_elems == 0 || max_elems > (1u << 24))
    return CRABML_HIP_BAD_INPUT;
  *out = nullptr;
  CH_LIVE(dev);
  CH_USE(dev);
  CH_FLUSH(dev);
  crabml_hip_tp_comm* c = new crabml_hip_tp_comm();
  c->dev = dev;
  c->nranks = nranks;
  c->rank = rank;
  c->p2p = true;
  c->cap = (unsigned)((max_elems + 2 + 31) / 32 * 32);  // + the two granules of the vocabulary-split sampler (k_argmax_step_tp)
  c->inbox_bytes = (size_t)TP_SLOTS * nranks * c->cap * 8 + 256;  // + the fault word
  // fine-grained device memory: stores from a peer GPU become visible to a kernel that is already running (the coarse-
  // grained default only promises that at kernel boundaries); plain hipMalloc is the fallback where the flag is refused
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, c->inbox_bytes, hipDeviceMallocFinegrained);
  c->finegrained = e == hipSuccess;
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipMalloc(&p, c->inbox_bytes);
  }
  if (e == hipSuccess) e = hipMemsetAsync(p, 0, c->inbox_bytes, dev->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
  if (e != hipSuccess) {
    if (p) (void)hipFree(p);
    delete c;
    return hip_fail(dev, e, "tp_p2p_create", __FILE__, __LINE__);
  }
  c->inbox = (unsigned long long*)p;
  c->fault = (int*)((char*)p + (size_t)TP_SLOTS * nranks * c->cap * 8);
  c->peer[rank] = p;
  c->connected = nranks == 1;
  *out = c;
  return 0;
}

// 64 bytes = hipIpcMemHandle_t of this rank's inbox; ship it to every peer (any side channel), then connect
int crabml_hip_tp_p2p_export(crabml_hip_tp_comm_t* comm, void* handle64) {
  if (!comm || !comm->p2p || !handle64) return CRABML_HIP_BAD_INPUT;
  crabml_hip_device* dev = comm->dev;
  CH_LIVE(dev);
  CH_USE(dev);
  CH_FLUSH(dev);
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  hipIpcMemHandle_t h;
  CH_HIP(dev, hipIpcGetMemHandle(&h, comm->inbox));
  memcpy(handle64, &h, 64);
  return 0;
}

// handles: nranks x 64 bytes in rank order (this rank's own entry is ignored)
int crabml_hip_tp_p2p_connect(crabml_hip_tp_comm_t* comm, const void* handles) {
  if (!comm || !comm->p2p || !handles) return CRABML_HIP_BAD_INPUT;
  crabml_hip_device* dev = comm->dev;
  CH_LIVE(dev);
  CH_USE(dev);
  CH_FLUSH(dev);
  if (comm->connected && comm->nranks > 1) CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "tp_p2p_connect: already connected");
  for (int r = 0; r < comm->nranks; r++) {
    if (r == comm->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)handles + (size_t)r * 64, 64);
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return hip_fail(dev, e, "hipIpcOpenMemHandle (peer inbox)", __FILE__, __LINE__);
    comm->peer[r] = p;
    comm->via_ipc[r] = true;
  }
  comm->connected = true;
  return 0;
}

// the same wiring for ranks that live in ONE process (one HipTensorDevice / stream per rank, possibly on the same GPU):
// the inboxes are plain device pointers, no IPC handle is involved
int crabml_hip_tp_p2p_connect_local(crabml_hip_tp_comm_t* const* comms, int n) {
  if (!comms || n < 1 || n > 8) return CRABML_HIP_BAD_INPUT;
  for (int r = 0; r < n; r++)
    if (!comms[r] || !comms[r]->p2p || comms[r]->nranks != n || comms[r]->rank != r || comms[r]->cap != comms[0]->cap)
      return CRABML_HIP_BAD_INPUT;
  for (int r = 0; r < n; r++) {
    for (int p = 0; p < n; p++) {
      if (p == r) continue;
      if (comms[p]->dev->ordinal != comms[r]->dev->ordinal) {  // another GPU of this process: map it
        (void)hipSetDevice(comms[r]->dev->ordinal);
        hipError_t e = hipDeviceEnablePeerAccess(comms[p]->dev->ordinal, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return hip_fail(comms[r]->dev, e, "hipDeviceEnablePeerAccess", __FILE__, __LINE__);
        (void)hipGetLastError();
      }
      comms[r]->peer[p] = comms[p]->inbox;
    }
    comms[r]->connected = true;
  }
  return 0;
}

}  // extern "C"

// ext_kc / ext_vc (lazy.hip): the caller's KV caches, [n_kv_heads][seq_len][head_dim] in the configured element type -- the layout
// of Llama2Runner's own cache tensors (llama2.rs:65-86) -- used in place
static int llama_create_impl(crabml_hip_device_t* dev, const crabml_hip_llama_config_t* cfg, const crabml_hip_llama_weights_t* w,
                             crabml_hip_buf* const* ext_kc, crabml_hip_buf* const* ext_vc, crabml_hip_llama_t** out) {
  if (!dev || !cfg || !w || !out) return CRABML_HIP_BAD_INPUT;
  *out = nullptr;
  const bool dry = dev->dry;
  const auto& g = *cfg;
  const int tp = g.tp_size > 1 ? g.tp_size : 1;
  if (!g.n_heads || !g.n_kv_heads || !g.n_layers || g.embedding_dim % g.n_heads || g.n_heads % g.n_kv_heads)
    CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "llama: inconsistent head configuration");
  if (tp > 8 || g.tp_rank < 0 || g.tp_rank >= tp || g.n_kv_heads % tp || g.hidden_dim % tp)
    CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "llama: tp_size %d must divide n_kv_heads and hidden_dim (and be <= 8)", tp);
  if (g.tp_comm) {
    const crabml_hip_tp_comm* m = (const crabml_hip_tp_comm*)g.tp_comm;
    if (m->p2p && (!m->connected || m->nranks != tp || m->rank != g.tp_rank || m->cap < g.embedding_dim || m->dev != dev))
      CH_BAIL(dev, CRABML_HIP_BAD_INPUT,
              "llama: the p2p group must be connected, match tp_size / tp_rank, live on this device and hold rows of >= embedding_dim");
  }
  // the F32 KV cache pairs head h with kv head h % n_kv (the batch_matmul broadcast quirk): those sets are not
  // contiguous head slices, so a GQA model shards by heads only with the F16 cache (h / (n_heads / n_kv))
  if (tp > 1 && !g.use_f16_kv_cache && g.n_heads != g.n_kv_heads)
    CH_BAIL(dev, CRABML_HIP_NOT_IMPLEMENTED, "llama: tensor-parallel GQA needs the f16 kv cache");
  const size_t hd = g.embedding_dim / g.n_heads;
  const size_t n_heads_l = g.n_heads / tp, n_kv_l = g.n_kv_heads / tp;
  const size_t dim_l = n_heads_l * hd, kv_dim_l = n_kv_l * hd, hidden_l = g.hidden_dim / tp;
  if (g.embedding_dim % 32 || hidden_l % 32 || dim_l % 32 || (hd & 1) || hd > 256 || (g.rope_dim & 1) || g.rope_dim > hd || !g.seq_len)
    CH_BAIL(dev, CRABML_HIP_NOT_IMPLEMENTED, "llama fused path: needs dim, local dims % 32 == 0, even head_dim <= 256, even rope_dim");
  if (g.embedding_dim > 12288)  // k_norm_quant keeps the row in 64 KiB of LDS
    CH_BAIL(dev, CRABML_HIP_NOT_IMPLEMENTED, "llama fused path: embedding_dim %zu > 12288", g.embedding_dim);
  if ((g.seq_len + hd) * sizeof(float) > 64 * 1024)
    CH_BAIL(dev, CRABML_HIP_NOT_IMPLEMENTED, "llama fused path: seq_len %zu needs more than 64 KiB of LDS for the score row", g.seq_len);
  if (!w->token_embed || !w->rms_final_weight || !w->wq || !w->wk || !w->wv || !w->wo || !w->ffn_gate_weight ||
      !w->ffn_down_weight || !w->ffn_up_weight || !w->rms_att_weight || !w->rms_ffn_weight)
    CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "llama: missing weights");
  const crabml_hip_buf* outw = w->output_weight ? w->output_weight : w->token_embed;
  const uint32_t wt = w->wq[0]->dtype, out_wt = outw->dtype;
  const uint32_t qt = vec_dot_rhs_dtype(wt), out_qt = vec_dot_rhs_dtype(out_wt);
  if (qt == 0xffffffffu || out_qt == 0xffffffffu)
    CH_BAIL(dev, CRABML_HIP_NOT_IMPLEMENTED, "llama: weight dtype %u / classifier dtype %u has no matmul_vec", wt, out_wt);
  // fused kernels exist for Q4_0 / Q8_0 layers (fast mode); everything else runs the per-op segment path
  // (a classifier of another format -- llama.cpp's "Q4_0" files keep output.weight in Q6_K -- does not take the layers
  // off the fused kernels: the final segment quantizes the normalized row for the classifier's own rhs type)
  const bool fused_fmt = wt == CRABML_HIP_Q4_0 || wt == CRABML_HIP_Q8_0 || wt == CRABML_HIP_Q4_1;
  // strict order, one device, a format whose dot is one term per block: the fused launches in their block-ordered form (7 per
  // layer: norm + quantize stay their own launches); everything else strict runs the per-op segments
  bool ord = dev->strict_order && fused_fmt && tp == 1;
  if (ord) {  // the term tables must fit LDS: 64 rows of gate|up (k_gateup_q_ord), a workgroup's 16 / 32 rows of the longest k (ffn_down)
    const size_t gu = (size_t)64 * (((g.embedding_dim / 32 + 3) & ~(size_t)3) + 4) * 4, 
                 dn = (size_t)((hidden_l / 32 >= 256 && (int)(g.embedding_dim / 32) <= dev->n_cu) ? 16 : 32) * (((hidden_l / 32 + 3) & ~(size_t)3) + 4) * 4;
    const void* fn = wt == CRABML_HIP_Q4_0   ? (const void*)k_gateup_q_ord<CRABML_HIP_Q4_0>
                     : wt == CRABML_HIP_Q8_0 ? (const void*)k_gateup_q_ord<CRABML_HIP_Q8_0>
                                             : (const void*)k_gateup_q_ord<CRABML_HIP_Q4_1>;
    if (gu > 150 * 1024 || dn > 60 * 1024 || (gu > 60 * 1024 && raise_dyn_lds(dev, fn, (int)gu) != hipSuccess)) ord = false;
    (void)hipGetLastError();
  }
  bool generic = (dev->strict_order && !ord) || !fused_fmt;
  const bool out_differs = out_wt != wt;
  {
    const size_t be = block_elems(wt) > block_elems(qt) ? block_elems(wt) : block_elems(qt);
    const size_t obe = block_elems(out_wt) > block_elems(out_qt) ? block_elems(out_wt) : block_elems(out_qt);
    if (g.embedding_dim % be || dim_l % be || hidden_l % be || g.embedding_dim % obe)
      CH_BAIL(dev, CRABML_HIP_NOT_IMPLEMENTED, "llama: dim / local dims are not multiples of the %zu-element blocks of dtype %u", be, wt);
  }
  auto check = [&](const crabml_hip_buf* b, size_t m, size_t k, uint32_t t) {
    return b && b->dtype == t && b->n_elems == m * k && (block_elems(t) == 1 || b->k == k);
  };
  // a layer's matrices may differ in GGML type (llama.cpp's *_K_M files: attn_v / ffn_down in Q6_K on some layers) as
  // long as they share the rhs type (buf/api.rs:142-159: every K-quant takes Q8_K): such a model runs the per-op
  // segments, each GEMV picking its kernel by the tensor's own dtype
  bool mixed = false;
  bool mix_v_down_q6k = true;  // every deviating tensor is an attn_v / ffn_down in Q6_K inside a Q4_K layer (the *_K_M recipe)
  auto check_w = [&](const crabml_hip_buf* b, size_t m, size_t k, bool v_or_down) {
    if (!b) return false;
    if (b->dtype != wt) {
      if (vec_dot_rhs_dtype(b->dtype) != qt || k % block_elems(b->dtype)) return false;
      mixed = true;
      if (!(v_or_down && wt == CRABML_HIP_Q4_K && b->dtype == CRABML_HIP_Q6_K)) mix_v_down_q6k = false;
    }
    return check(b, m, k, b->dtype);
  };
  for (size_t l = 0; l < g.n_layers; l++) {
    if (!check_w(w->wq[l], dim_l, g.embedding_dim, false) || !check_w(w->wk[l], kv_dim_l, g.embedding_dim, false) ||
        !check_w(w->wv[l], kv_dim_l, g.embedding_dim, true) || !check_w(w->wo[l], g.embedding_dim, dim_l, false) ||
        !check_w(w->ffn_gate_weight[l], hidden_l, g.embedding_dim, false) ||
        !check_w(w->ffn_up_weight[l], hidden_l, g.embedding_dim, false) ||
        !check_w(w->ffn_down_weight[l], g.embedding_dim, hidden_l, true) ||
        !check(w->rms_att_weight[l], 1, g.embedding_dim, CRABML_HIP_F32) ||
        !check(w->rms_ffn_weight[l], 1, g.embedding_dim, CRABML_HIP_F32))
      CH_BAIL(dev, CRABML_HIP_NOT_IMPLEMENTED,
              "llama fused path: layer %zu weights have an unexpected shape, or dtypes that do not share one rhs dtype (tp=%d)", l, tp);
  }
  // the classifier split by vocabulary: this rank holds rows [tp_rank V / tp, (tp_rank + 1) V / tp)
  const bool split_vocab = tp > 1 && (g.flags & CRABML_HIP_LLAMA_TP_SPLIT_VOCAB) != 0;
  if (split_vocab) {
    if (!w->output_weight || g.vocab_size % tp)
      CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "llama: the vocabulary split needs an untied output.weight and vocab_size %% tp_size == 0");
    if (g.tp_comm && !((const crabml_hip_tp_comm*)g.tp_comm)->p2p)
      CH_BAIL(dev, CRABML_HIP_NOT_IMPLEMENTED, "llama: the vocabulary split exchanges its arg-max pairs through a P2P group (crabml_hip_tp_p2p_*)");
  }
  const size_t vocab_l = split_vocab ? g.vocab_size / tp : g.vocab_size;
  if (!check(outw, vocab_l, g.embedding_dim, out_wt) || !check(w->rms_final_weight, 1, g.embedding_dim, CRABML_HIP_F32) ||
      w->token_embed->n_elems != g.vocab_size * g.embedding_dim)
    CH_BAIL(dev, CRABML_HIP_NOT_IMPLEMENTED, "llama fused path: classifier / final norm / embedding dtype or shape");

  if (!dry) CH_USE(dev);
  crabml_hip_llama* c = new crabml_hip_llama();
  c->dev = dev;
  if (!dry && hipHostMalloc((void**)&c->h_state, (crabml_hip_llama::H_STATE_SLOTS * 4 + 4) * sizeof(int), hipHostMallocDefault) != hipSuccess) {
    delete c;
    CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "llama: hipHostMalloc of the state staging ring failed");
  }
  c->cfg = g;
  c->wtype = wt;
  if (ext_kc != nullptr && !dry) {
    static const bool on = [] { const char* e = getenv("CRABML_HIP_LAZY_NO_HOST_LOGITS"); return !(e && e[0] == '1'); }();
    if (on && hipHostMalloc((void**)&c->host_logits, g.vocab_size * 4 + 64, hipHostMallocDefault) == hipSuccess)
      memset(c->host_logits + g.vocab_size, 0, 64);
    else
      c->host_logits = nullptr;
    (void)hipGetLastError();
  }
  // the Q4_K fused kernels take a Q6_K attn_v / ffn_down beside the Q4_K planes, but only in the norm-epilogue form
  const bool nepi_k_possible = !dev->strict_order && wt == CRABML_HIP_Q4_K && out_qt == CRABML_HIP_Q8_K && tp == 1 &&
                               !(g.flags & (CRABML_HIP_LLAMA_NO_NORM_EPILOGUE | CRABML_HIP_LLAMA_NO_KQUANT_FUSION)) &&
                               g.embedding_dim % 256 == 0 && (int)(g.embedding_dim / 32) <= dev->n_cu;
  const bool mix_fused = mixed && mix_v_down_q6k && nepi_k_possible;
  if (mixed && ord) {  // (a mixed file: per-op segments)
    ord = false;
    generic = true;
  }
  generic = generic || (mixed && !mix_fused);
  // strict order, pure Q4_K layers on one device (round 6): the five fused launches of the fast Q4_K step in their ORDERED form -- nine
  // f32 terms per super-block (eight exact class sums x d, and dmin x sumi) parked in LDS and added in super-block order, the
  // reference's norm order in the wo / ffn_down epilogue (k_qkv_ord, k_gemv_res_nq<.., ORD>, k_gateup_k_lds<.., ORD>); bit-identical
  // to the per-op segments they replace (16 launches per layer).  The term tables must fit LDS.
  // (a *_K_M mix too: its Q6_K attn_v / ffn_down rows leave the same records, rows_terms_q6k)
  bool ordk = dev->strict_order && wt == CRABML_HIP_Q4_K && (!mixed || mix_v_down_q6k) && out_qt == CRABML_HIP_Q8_K && tp == 1 &&
              !(g.flags & (CRABML_HIP_LLAMA_NO_NORM_EPILOGUE | CRABML_HIP_LLAMA_NO_KQUANT_FUSION | CRABML_HIP_LLAMA_NO_RHS_PROLOGUE)) &&
              g.embedding_dim % 256 == 0 && dim_l % 256 == 0 && hidden_l % 256 == 0 && (int)(g.embedding_dim / 32) <= dev->n_cu;
  if (ordk && !dry) {
    auto planes_b = [](size_t k) { return ((k + k / 256 * 4 + k / 16 * 2) + 15) & ~(size_t)15; };
    const int split_dn = (g.flags & CRABML_HIP_LLAMA_SPLIT_CHUNKS_ALWAYS) ? 2 : (g.flags & CRABML_HIP_LLAMA_SPLIT_CHUNKS_NEVER) ? 1 : hidden_l / 32 >= 256 ? 2 : 1;
    const int split_wo = (g.flags & CRABML_HIP_LLAMA_SPLIT_CHUNKS_ALWAYS) ? 2 : (g.flags & CRABML_HIP_LLAMA_SPLIT_CHUNKS_NEVER) ? 1 : dim_l / 32 >= 256 ? 2 : 1;
    const size_t gu = planes_b(g.embedding_dim) + (size_t)64 * (size_t)q4k_rec_stride((int)(g.embedding_dim / 256)) * 4;
    const size_t dn = planes_b(hidden_l) + (size_t)(32 / split_dn) * (size_t)q4k_rec_stride((int)(hidden_l / 256)) * 4;
    const size_t wo = planes_b(dim_l) + (size_t)(32 / split_wo) * (size_t)q4k_rec_stride((int)(dim_l / 256)) * 4;
    const size_t nq = dn > wo ? dn : wo;
    bool fits = gu <= 150 * 1024 && nq <= 150 * 1024;
    if (fits && gu > 48 * 1024)
      fits = raise_dyn_lds(dev, (const void*)k_gateup_k_lds<true, true>, (int)gu) == hipSuccess &&
             raise_dyn_lds(dev, (const void*)k_gateup_k_lds<false, true>, (int)gu) == hipSuccess;
    if (fits && nq > 48 * 1024)
      fits = raise_dyn_lds(dev, (const void*)k_gemv_res_nq<CRABML_HIP_Q4_K, 1, 1, false, false, true>, (int)nq) == hipSuccess &&
             raise_dyn_lds(dev, (const void*)k_gemv_res_nq<CRABML_HIP_Q4_K, 1, 2, false, false, true>, (int)nq) == hipSuccess &&
             raise_dyn_lds(dev, (const void*)k_gemv_res_nq<CRABML_HIP_Q4_K, 2, 1, false, false, true>, (int)nq) == hipSuccess &&
             raise_dyn_lds(dev, (const void*)k_gemv_res_nq<CRABML_HIP_Q4_K, 2, 2, false, false, true>, (int)nq) == hipSuccess;
    (void)hipGetLastError();
    if (!fits) ordk = false;
  }
  if (ordk) ord = true;
  if (ordk && mixed) generic = false;
  c->generic = generic;
  c->ord = ord;
  // Q4_K always; Q4_1 when it cannot take the 5-kernel path (mixed classifier format) or for the A/B flag
  c->kfused = ordk ||
              (!dev->strict_order && (!mixed || mix_fused) && !(g.flags & CRABML_HIP_LLAMA_NO_KQUANT_FUSION) &&
               (wt == CRABML_HIP_Q4_K || (wt == CRABML_HIP_Q4_1 && (generic || out_differs || (g.flags & CRABML_HIP_LLAMA_Q4_1_SEGMENTS)))));
  c->qt = qt;
  c->out_qt = out_qt;
  c->tp = tp;
  c->tp_rank = g.tp_rank;
  c->comm = (crabml_hip_tp_comm*)g.tp_comm;
  c->tp_dry = tp > 1 && !g.tp_comm && (g.flags & CRABML_HIP_LLAMA_TP_DRY_RUN);
  if (c->comm && c->comm->p2p) c->tp_salt = (++c->comm->sessions) * 0x9E3779B1u;
  c->split_vocab = split_vocab;
  c->vocab_l = (int)vocab_l;
  c->vocab_off = split_vocab ? (int)(vocab_l * (size_t)g.tp_rank) : 0;
  c->hd = (int)hd;
  c->npairs = (int)(g.rope_dim / 2);
  c->n_heads_l = (int)n_heads_l;
  c->n_kv_l = (int)n_kv_l;
  c->dim_l = (int)dim_l;
  c->kv_dim_l = (int)kv_dim_l;
  c->hidden_l = (int)hidden_l;
  int rc = 0;
  auto hold = [&](const crabml_hip_buf* b) {
    crabml_hip_buf* m = const_cast<crabml_hip_buf*>(b);
    crabml_hip_buf_retain(m);
    c->held.push_back(m);
    if (rc == 0) rc = ensure_mem(dev, m);
    return m;
  };
  c->token_embed = hold(w->token_embed);
  c->rms_final = hold(w->rms_final_weight);
  c->output = hold(outw);
  for (size_t l = 0; l < g.n_layers; l++) {
    c->rms_att.push_back(hold(w->rms_att_weight[l]));
    c->rms_ffn.push_back(hold(w->rms_ffn_weight[l]));
    c->wq.push_back(hold(w->wq[l]));
    c->wk.push_back(hold(w->wk[l]));
    c->wv.push_back(hold(w->wv[l]));
    c->wo.push_back(hold(w->wo[l]));
    c->gate.push_back(hold(w->ffn_gate_weight[l]));
    c->down.push_back(hold(w->ffn_down_weight[l]));
    c->up.push_back(hold(w->ffn_up_weight[l]));
  }
  auto A = [&](size_t bytes, void** p) {
    if (rc == 0) rc = dalloc(c, bytes, p);
  };
  const size_t es = g.use_f16_kv_cache ? 2 : 4;
  c->kv_bytes = n_kv_l * g.seq_len * hd * es;
  c->kc.resize(g.n_layers);
  c->vc.resize(g.n_layers);
  c->ext_kv = ext_kc != nullptr;
  for (size_t l = 0; l < g.n_layers; l++) {
    if (c->ext_kv) {
      const uint32_t kvt = g.use_f16_kv_cache ? CRABML_HIP_F16 : CRABML_HIP_F32;
      if (!ext_kc[l] || !ext_vc[l] || ext_kc[l]->dtype != kvt || ext_vc[l]->dtype != kvt || ext_kc[l]->n_elems * es != c->kv_bytes ||
          ext_vc[l]->n_elems * es != c->kv_bytes) {
        if (rc == 0) rc = set_error(dev, CRABML_HIP_BAD_INPUT, "llama: external kv cache of layer %zu has the wrong type or size", l);
        continue;
      }
      if (l == 0) c->ext_kc0 = ext_kc[l];
      c->kc[l] = hold(ext_kc[l])->ptr;
      c->vc[l] = hold(ext_vc[l])->ptr;
    } else {
      A(c->kv_bytes, &c->kc[l]);
      A(c->kv_bytes, &c->vc[l]);
    }
  }
  A(g.embedding_dim * 4, (void**)&c->x);
  A(g.embedding_dim * 4, (void**)&c->partial);
  A(dim_l * 4, (void**)&c->qbuf);
  A(dim_l * 4, (void**)&c->attn);
  A(hidden_l * 4, (void**)&c->h);
  A(g.vocab_size * 4, (void**)&c->logits);
  size_t tmp_n = dim_l + 2 * kv_dim_l;
  if (2 * hidden_l > tmp_n) tmp_n = 2 * hidden_l;
  if (g.embedding_dim > tmp_n) tmp_n = g.embedding_dim;
  A(tmp_n * 4, (void**)&c->tmp);
  {
    auto act_bytes = [](uint32_t t, size_t n) { return t == CRABML_HIP_F32 ? (size_t)16 : act_layout(t, n).total; };
    size_t a_dim = act_bytes(qt, g.embedding_dim), a_out = act_bytes(out_qt, g.embedding_dim);
    A(a_dim > a_out ? a_dim : a_out, (void**)&c->act_dim);
    A(act_bytes(qt, dim_l), (void**)&c->act_attn);
    A(act_bytes(qt, hidden_l), (void**)&c->act_hid);
    A(g.embedding_dim * 4, (void**)&c->xn);
  }
  A(g.seq_len * (size_t)(c->npairs ? c->npairs : 1) * 2 * 4, (void**)&c->rope);
  {
    const size_t grp = n_heads_l / n_kv_l;
    const bool long_geom = g.use_f16_kv_cache && hd % 32 == 0 && (grp == 1 || grp == 2 || grp == 4 || grp == 8) &&
                           !(g.flags & CRABML_HIP_LLAMA_NO_LONG_ATTENTION);
    c->attn_long_ok = long_geom && g.seq_len % 8 == 0;
    c->attn_long_from = g.attn_long_from ? g.attn_long_from : 224;  // exact kernels: measured crossover on MI355X (Llama-3-8B shape) ~200-220
    if (c->attn_long_ok && g.seq_len * 4 > 64 * 1024) {
      // the softmax kernels keep a head's score row in LDS: rows past 16384 positions need the raised dynamic-LDS limit,
      // rows past ~38000 do not fit at all (the step then stays on the one-workgroup-per-head kernel)
      const int lds = (int)(g.seq_len * 4);
      if (lds > 150 * 1024 ||
          raise_dyn_lds(dev, (const void*)k_attn_softmax<16>, lds) != hipSuccess ||
          raise_dyn_lds(dev, (const void*)k_attn_softmax<4>, lds) != hipSuccess)
        c->attn_long_ok = false;
      (void)hipGetLastError();
    }
    c->exact_long_ok = c->attn_long_ok;
    if (c->attn_long_ok) {
      A(n_heads_l * g.seq_len * 4, (void**)&c->scores_g);
      A(n_heads_l * g.seq_len * 2, (void**)&c->p16);
      if (!(g.flags & CRABML_HIP_LLAMA_NO_PV_PRODUCER_WAVES) && g.seq_len % 4 == 0) {
        hipError_t e = hipErrorInvalidValue;
        switch (grp) {
          case 1: e = raise_dyn_lds(dev, (const void*)k_attn_pv_split<1>, (int)PvSplit<1>::LDS); break;
          case 2: e = raise_dyn_lds(dev, (const void*)k_attn_pv_split<2>, (int)PvSplit<2>::LDS); break;
          case 4: e = raise_dyn_lds(dev, (const void*)k_attn_pv_split<4>, (int)PvSplit<4>::LDS); break;
          default: e = raise_dyn_lds(dev, (const void*)k_attn_pv_split<8>, (int)PvSplit<8>::LDS); break;
        }
        c->pv_split = e == hipSuccess;
        (void)hipGetLastError();
      }
    }
    // the fast step's long-context attention: split-KV with f32 accumulation (k_attn_flash) unless the exact chain is asked for
    // (k_attn_flash reads the cache rows only -- head_dim halves each --, so any seq_len will do: a cache of 1001 positions must not
    // fall back to one workgroup per head, 45 us per layer at 900 positions)
    if (long_geom && !dev->strict_order && !(g.flags & CRABML_HIP_LLAMA_EXACT_ATTENTION)) {
      c->flash_ticket = (g.flags & CRABML_HIP_LLAMA_FLASH_TICKET) != 0;
      const FlashFn fn = flash_kernel((int)grp, (int)hd, qt == CRABML_HIP_Q8_1, c->flash_ticket);
      if (fn != nullptr && raise_dyn_lds(dev, (const void*)fn, (int)flash_lds_bytes((int)grp, (int)hd)) == hipSuccess) {
        int S = dev->n_cu / (int)n_kv_l;
        S = S < 1 ? 1 : S > FLASH_MAX_SLICES ? FLASH_MAX_SLICES : S;
        if (const char* hooks = getenv("CRABML_HIP_TEST_HOOKS"))  // tuning hook (tools/flash_sweep.py): slices per kv head in the grid
          if (hooks[0] == '1')
            if (const char* e = getenv("CRABML_HIP_FLASH_SLICES")) {
              const int v = atoi(e);
              if (v >= 1 && v <= FLASH_MAX_SLICES) S = v;
            }
        c->flash_S = S;
        if (const char* hooks = getenv("CRABML_HIP_TEST_HOOKS"))  // tuning hook (tools/flash_sweep.py); armed like ASSUME_CUS
          if (hooks[0] == '1')
            if (const char* e = getenv("CRABML_HIP_FLASH_MIN_ROWS")) {
              const int v = atoi(e);
              if (v >= 8 && v <= 65536) c->flash_min_rows = v;
            }
        A(n_kv_l * (size_t)S * flash_part_floats((int)grp, (int)hd) * 4, (void**)&c->flash_part);
        A(n_kv_l * 4, (void**)&c->flash_tick);
        if (rc == 0 && !dry && hipMemsetAsync(c->flash_tick, 0, n_kv_l * 4, dev->stream) != hipSuccess) rc = CRABML_HIP_UNEXPECTED;
        c->attn_flash = rc == 0;
        if (c->attn_flash) c->attn_long_ok = true;
        // Below ~768 cached positions the merge inside the launch (last-arriving workgroup of a kv head, ticket word) beats the
        // second launch -- 7.4 vs 5.1 + 4.1 us per layer at 128 positions, 8.7 vs 10.0 at 512, 10.8 vs 9.95 at 1024
        // (profiles/r05_flash_ticket_sweep.md; same partials, same merge order: bit-identical): a third graph variant serves that range.
        if (c->attn_flash && !c->flash_ticket) {
          size_t until = 768;
          if (const char* hooks = getenv("CRABML_HIP_TEST_HOOKS"))  // tuning hook: 0 = never
            if (hooks[0] == '1')
              if (const char* e = getenv("CRABML_HIP_FLASH_TICKET_UNTIL")) until = (size_t)atol(e);
          const FlashFn tfn = flash_kernel((int)grp, (int)hd, qt == CRABML_HIP_Q8_1, true);
          if (until > 0 && tfn != nullptr && raise_dyn_lds(dev, (const void*)tfn, (int)flash_lds_bytes((int)grp, (int)hd)) == hipSuccess)
            c->flash_ticket_until = until;
          (void)hipGetLastError();
        }
        // the prompt pass's causal attention of the fast step (k_attn_flash_rows): 70 KB of LDS at head_dim 128
        if (c->attn_flash && (hd == 128 || hd == 64) &&
            raise_dyn_lds(dev, hd == 128 ? (const void*)k_attn_flash_rows<128> : (const void*)k_attn_flash_rows<64>,
                          (int)flash_rows_lds_bytes((int)hd)) == hipSuccess)
          c->attn_flash_rows = true;
        (void)hipGetLastError();
        // k_attn_flash + merge overtake the staged one-workgroup kernel between 64 and 96 cached positions (8B shape, per layer:
        // 51.0 vs 51.6 us at 64, 52.2 vs 51.4 at 96, 59.0 vs 51.8 at 224; profiles/r04_flash_sweep.log)
        if (c->attn_flash && g.attn_long_from == 0) c->attn_long_from = 96;
      }
      (void)hipGetLastError();
    }
    // short-context attention with K / V staged through LDS (f16 cache): variant 0 serves positions < S
    if (g.use_f16_kv_cache && hd % 8 == 0 && !(g.flags & CRABML_HIP_LLAMA_NO_STAGED_ATTENTION)) {
      const size_t S = c->attn_long_ok && c->attn_long_from < g.seq_len ? c->attn_long_from : g.seq_len;
      const size_t lds = attn_s_lds_bytes((int)S, (int)hd);
      if (lds <= 150 * 1024 &&
          raise_dyn_lds(dev, (const void*)k_attn_s<128>, (int)lds) == hipSuccess &&
          raise_dyn_lds(dev, (const void*)k_attn_s<0>, (int)lds) == hipSuccess) {
        c->attn_s_rows = (int)S;
        c->attn_s_lds = lds;
      }
      (void)hipGetLastError();
    }
  }
  A(8 * sizeof(int), (void**)&c->state);
  A((g.embedding_dim / 16 + g.embedding_dim) * 8, (void**)&c->slots);
  // tp > 1: the epilogue also hosts the collective when the group is the P2P kind (or in the collective-free dry run)
  const bool p2p_comm = c->comm != nullptr && c->comm->p2p;
  // (not for the K-quant segment path -- a Q4_1 body with a classifier of another format runs it: its wo / ffn_down launches
  // host neither the norm epilogue nor the collective, so over a P2P group the stand-alone all-reduce launch must run)
  c->norm_epi = !generic && !c->kfused && (tp == 1 || p2p_comm || c->tp_dry) && !(g.flags & CRABML_HIP_LLAMA_NO_NORM_EPILOGUE) &&
                (int)(g.embedding_dim / 32) <= dev->n_cu;  // every workgroup of the gather must be resident
  c->norm_epi_k = c->kfused && wt == CRABML_HIP_Q4_K && out_qt == CRABML_HIP_Q8_K && tp == 1 &&
                  !(g.flags & CRABML_HIP_LLAMA_NO_NORM_EPILOGUE) && g.embedding_dim % 256 == 0 && (int)(g.embedding_dim / 32) <= dev->n_cu;
  c->defer_norm = c->norm_epi && tp == 1 && !ord && !dev->strict_order && (wt == CRABML_HIP_Q4_0 || wt == CRABML_HIP_Q8_0) &&
                  !(g.flags & CRABML_HIP_LLAMA_EXACT_NORM) && (int)(g.embedding_dim / 32) <= dev->n_cu;
  if (c->defer_norm) A(g.embedding_dim / 16 * 4, (void**)&c->rsums);
  // A tensor-parallel rank's gate/up: hidden / tp / 32 workgroups of 32 rows would leave most CUs idle.  h stays f32 from workgroups
  // of `gu_rows` rows (the largest even divisor of the rank's rows, at most 32, that gives at least one workgroup per CU) and
  // ffn_down quantizes it in its prologue.
  if (c->norm_epi && tp > 1 && !ord && !dev->strict_order && (wt == CRABML_HIP_Q4_0 || wt == CRABML_HIP_Q8_0) &&
      !(g.flags & CRABML_HIP_LLAMA_NO_H_CONSUMER_QUANT) && hidden_l % 32 == 0 && (int)(hidden_l / 32) * 2 <= dev->n_cu && !dry) {
    int pick = 0;
    for (int r = 30; r >= 2 && !pick; r -= 2)
      if (hidden_l % r == 0 && (int)(hidden_l / r) >= dev->n_cu && (int)(hidden_l / r) <= 2 * dev->n_cu) pick = r;
    if (pick && q8_0_lds_bytes((int)(hidden_l / 32)) <= 60 * 1024) c->gu_rows = pick;
  }
  c->q8k_producers = c->norm_epi_k && !(g.flags & (CRABML_HIP_LLAMA_NO_RHS_PROLOGUE | CRABML_HIP_LLAMA_NO_Q8K_PRODUCERS)) && dim_l % 256 == 0 &&
                     hidden_l % 256 == 0 && (hd == 64 || hd == 128 || hd == 256) && (int)(hidden_l / 32) <= 2 * dev->n_cu;
  // the fast Q4_K step: wo leaves x only, gate | up normalizes and quantizes the row itself (k_gateup_k_lds<.., NORMIN>; the same bits)
  c->k_norm_in = c->q8k_producers && wt == CRABML_HIP_Q4_K && tp == 1 && !ord && !dev->strict_order && g.embedding_dim % 256 == 0 && g.embedding_dim / 256 <= 32 &&
                 !(g.flags & (CRABML_HIP_LLAMA_NO_K_NORM_IN | CRABML_HIP_LLAMA_SPLIT_CHUNKS_ALWAYS | CRABML_HIP_LLAMA_SPLIT_CHUNKS_NEVER));
  if (c->k_norm_in && !c->rsums) A(g.embedding_dim / 16 * 4, (void**)&c->rsums);
  if (c->q8k_producers) {
    A(dim_l * 8, (void**)&c->a8gran);
    A(hidden_l * 8, (void**)&c->h8gran);
  }
  c->out_cap = (int)g.seq_len;
  A((size_t)c->out_cap * 4, (void**)&c->out_tokens);
  A(ARGMAX_BLOCKS * 4, (void**)&c->am_val);
  A(ARGMAX_BLOCKS * 4, (void**)&c->am_idx);
  A(2 * sizeof(int), (void**)&c->am_best);
  if (rc != 0) {
    crabml_hip_llama_destroy(c);
    return rc;
  }
  // RoPE table with the reference's own recurrence (rope.rs:47-54: theta_scale = 10000^(-2/hd), theta = pos,
  // theta *= theta_scale per pair; base hard-coded) evaluated with the host libm, as the trait op does.
  {
    std::vector<float> tab(g.seq_len * (size_t)(c->npairs ? c->npairs : 1) * 2, 0.f);
    const float theta_scale = powf(10000.0f, -2.0f / (float)hd);
    for (size_t p = 0; p < g.seq_len; p++) {
      float theta = (float)p;
      for (int i = 0; i < c->npairs; i++) {
        tab[(p * c->npairs + i) * 2] = cosf(theta);
        tab[(p * c->npairs + i) * 2 + 1] = sinf(theta);
        theta *= theta_scale;
      }
    }
    hipError_t e = dry ? hipErrorUnknown : hipMemcpyAsync(c->rope, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, dev->stream);
    if (dry) {  // record-only test device: nothing to initialize
      *out = c;
      return 0;
    }
    if (e == hipSuccess) e = hipMemsetAsync(c->state, 0, 8 * sizeof(int), dev->stream);
    // vocabulary split: the entries of the other ranks' shards read -inf (an element-wise max over the ranks is the all-gather)
    if (e == hipSuccess && c->split_vocab) e = hipMemsetD32Async((hipDeviceptr_t)c->logits, (int)0xff800000u, g.vocab_size, dev->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->slots, 0, (g.embedding_dim / 16 + g.embedding_dim) * 8, dev->stream);
    if (e == hipSuccess && c->a8gran) e = hipMemsetAsync(c->a8gran, 0, dim_l * 8, dev->stream);
    if (e == hipSuccess && c->h8gran) e = hipMemsetAsync(c->h8gran, 0, hidden_l * 8, dev->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
    if (e != hipSuccess) {
      crabml_hip_llama_destroy(c);
      return hip_fail(dev, e, "llama init", __FILE__, __LINE__);
    }
  }
  // capture one decode step into a graph (token / pos / step are read from device memory by the kernels).
  // tp > 1 without a communicator = a rank of the single-device simulation: driven segment by segment, no graph.
  // tp > 1 over RCCL launches eagerly unless CRABML_HIP_LLAMA_TP_GRAPH asks for the collectives to be captured too.
  const bool want_graph = !(g.flags & CRABML_HIP_LLAMA_NO_GRAPH) &&
                          (tp == 1 || c->tp_dry || p2p_comm || (c->comm != nullptr && (g.flags & CRABML_HIP_LLAMA_TP_GRAPH)));
  if (want_graph) {
    const int nvar = c->attn_long_ok ? (c->flash_ticket_until > 0 ? 3 : 2) : 1;
    bool ok = true;
    for (int v = 0; v < nvar && ok; v++) {
      ok = false;
      hipError_t e = hipStreamBeginCapture(dev->stream, hipStreamCaptureModeThreadLocal);
      if (e != hipSuccess) break;
      c->capturing = true;
      c->attn_variant = v;
      int erc = enqueue_step(c);
      c->capturing = false;
      hipGraph_t graph = nullptr;
      hipError_t e2 = hipStreamEndCapture(dev->stream, &graph);
      if (erc == 0 && e2 == hipSuccess && graph) {
        hipGraphExec_t exec = nullptr;
        if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
          c->graph[v] = graph;
          c->exec[v] = exec;
          ok = true;
        } else {
          (void)hipGraphDestroy(graph);
        }
      } else if (graph) {
        (void)hipGraphDestroy(graph);
      }
    }
    (void)hipGetLastError();
    c->use_graph = ok;
    c->attn_variant = 0;
    if (!ok && tp == 1) {  // fail loudly: the caller asked for the graph path
      crabml_hip_llama_destroy(c);
      CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "llama: hipGraph capture/instantiate failed");
    }
    // tp > 1: if RCCL could not be captured the step simply runs eagerly
  }
  *out = c;
  return 0;
}

extern "C" {

int crabml_hip_llama_create(crabml_hip_device_t* dev, const crabml_hip_llama_config_t* cfg,
                            const crabml_hip_llama_weights_t* w, crabml_hip_llama_t** out) {
  if (!dev || !cfg || !w || !out) return CRABML_HIP_BAD_INPUT;
  *out = nullptr;
  CH_LIVE(dev);
  CH_USE(dev);
  CH_FLUSH(dev);
  return llama_create_impl(dev, cfg, w, nullptr, nullptr, out);
}

int crabml_hip_llama_destroy(crabml_hip_llama_t* c) {
  if (!c) return 0;
  if (!c->dev->dry) {
    (void)hipSetDevice(c->dev->ordinal);
    (void)hipStreamSynchronize(c->dev->stream);
  }
  for (int v = 0; v < 3; v++) {
    if (c->exec[v]) (void)hipGraphExecDestroy(c->exec[v]);
    if (c->graph[v]) (void)hipGraphDestroy(c->graph[v]);
  }
  for (auto& a : c->allocs) pool_free(c->dev, a.first, a.second);
  for (auto* b : c->held) crabml_hip_buf_release(b);
  if (c->h_state) (void)hipHostFree(c->h_state);
  if (c->host_logits) (void)hipHostFree(c->host_logits);
  delete c;
  return 0;
}

static int check_step(crabml_hip_llama* c, size_t token, size_t pos) {
  crabml_hip_device* dev = c->dev;
  if (token >= c->cfg.vocab_size) CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "llama: token %zu out of range", token);
  if (pos != c->kv_len) CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "llama: pos %zu != kv cache length %zu", pos, c->kv_len);
  if (pos >= c->cfg.seq_len) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "llama: kv cache is full (%zu)", c->cfg.seq_len);
  return 0;
}

int crabml_hip_llama_forward(crabml_hip_llama_t* c, size_t token, size_t pos, float* logits) {
  if (!c) return CRABML_HIP_BAD_INPUT;
  crabml_hip_device* dev = c->dev;
  CH_LIVE(dev);
  CH_USE(dev);
  CH_FLUSH(dev);
  if (c->tp > 1 && !c->comm && !c->tp_dry)
    CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "llama: a tp rank without a communicator is driven by crabml_hip_llama_tp_sim_*");
  CH_TRY(check_step(c, token, pos));
  CH_TRY(set_state(c, token, pos, 0));
  CH_TRY(run_step(c, pos));
  c->kv_len++;
  if (logits) {
    int fault = 0;
    CH_HIP(dev, hipMemcpyAsync(logits, c->logits, c->cfg.vocab_size * 4, hipMemcpyDeviceToHost, dev->stream));
    CH_HIP(dev, hipMemcpyAsync(&fault, c->state + 5, sizeof(int), hipMemcpyDeviceToHost, dev->stream));
    CH_HIP(dev, hipStreamSynchronize(dev->stream));
    if (fault == 2) CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "llama: a tensor-parallel peer's partial sums never arrived (poll timed out)");
    if (fault) CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "llama: a norm-epilogue gather timed out (workgroups not co-resident?)");
  }
  return 0;
}

int crabml_hip_llama_decode_greedy(crabml_hip_llama_t* c, size_t token, size_t n_steps, uint32_t* out_tokens) {
  if (!c || (!out_tokens && n_steps)) return CRABML_HIP_BAD_INPUT;
  crabml_hip_device* dev = c->dev;
  CH_LIVE(dev);
  CH_USE(dev);
  CH_FLUSH(dev);
  if (c->tp > 1 && !c->comm && !c->tp_dry)
    CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "llama: a tp rank without a communicator is driven by crabml_hip_llama_tp_sim_*");
  if (token >= c->cfg.vocab_size) CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "llama: token %zu out of range", token);
  if (c->kv_len + n_steps > c->cfg.seq_len || n_steps > (size_t)c->out_cap)
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "llama: %zu steps do not fit the kv cache (%zu of %zu used)", n_steps, c->kv_len, c->cfg.seq_len);
  if (n_steps == 0) return 0;
  CH_TRY(set_state(c, token, c->kv_len, 0));
  for (size_t s = 0; s < n_steps; s++) CH_TRY(run_step(c, c->kv_len + s));
  c->kv_len += n_steps;
  int fault = 0;
  CH_HIP(dev, hipMemcpyAsync(out_tokens, c->out_tokens, n_steps * 4, hipMemcpyDeviceToHost, dev->stream));
  CH_HIP(dev, hipMemcpyAsync(&fault, c->state + 5, sizeof(int), hipMemcpyDeviceToHost, dev->stream));
  CH_HIP(dev, hipStreamSynchronize(dev->stream));
  if (fault == 2) CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "llama: a tensor-parallel peer's partial sums never arrived (poll timed out)");
  if (fault) CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "llama: a norm-epilogue gather timed out (workgroups not co-resident?)");
  return 0;
}


int crabml_hip_llama_prefill(crabml_hip_llama_t* c, const uint32_t* tokens, size_t n, float* logits) {
  if (!c || (!tokens && n)) return CRABML_HIP_BAD_INPUT;
  crabml_hip_device* dev = c->dev;
  CH_LIVE(dev);
  CH_USE(dev);
  CH_FLUSH(dev);
  if (n == 0) CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "llama prefill: expected at least 1 prompt token");  // llama2.rs:117-122
  for (size_t i = 0; i < n; i++)
    if (tokens[i] >= c->cfg.vocab_size) CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "llama: token %u out of range", tokens[i]);
  if (c->kv_len + n > c->cfg.seq_len)
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "llama: %zu prompt tokens do not fit the kv cache (%zu of %zu used)", n, c->kv_len, c->cfg.seq_len);
  if (c->tp > 1) {  // token loop
    for (size_t i = 0; i < n; i++) CH_TRY(crabml_hip_llama_forward(c, tokens[i], c->kv_len, i + 1 == n ? logits : nullptr));
    return 0;
  }
  // rows per pass: 1024 on the fast device (8B shape Q4_0: 31.5k / 38k / 45k / 46k prompt tok/s at 256 / 512 / 1024 / 2048 rows -- the
  // narrow GEMMs get their column tiles; the row buffers are ~0.5 GB), 512 on the strict one (its exact attention tiles hold 1024
  // positions in LDS), never more than the cache holds
  const size_t chunk0 = c->cfg.prefill_chunk ? c->cfg.prefill_chunk : dev->strict_order ? 512 : 1024;
  const size_t chunk = chunk0 < c->cfg.seq_len ? chunk0 : c->cfg.seq_len;
  CH_TRY(prefill_alloc(c, chunk));
  for (size_t i = 0; i < n; i += chunk) {
    const size_t B = n - i < chunk ? n - i : chunk;
    CH_TRY(prefill_chunk(c, tokens + i, B, c->kv_len, logits != nullptr && i + B == n));
    c->kv_len += B;
  }
  if (logits) {
    CH_HIP(dev, hipMemcpyAsync(logits, c->logits, c->cfg.vocab_size * 4, hipMemcpyDeviceToHost, dev->stream));
    CH_HIP(dev, hipStreamSynchronize(dev->stream));
  }
  return 0;
}

// Single-device simulation of a tensor-parallel group: `ranks[r]` was created with tp_size = n, tp_rank = r,
// tp_comm = NULL on the SAME device.  Segments are enqueued rank by rank and the all-reduce is a local kernel
// (sum in rank order).  Validates the sharding, the partial-sum plumbing and the residual hand-off on one GPU.
int crabml_hip_llama_tp_sim_forward(crabml_hip_llama_t* const* ranks, int n, size_t token, size_t pos, float* logits) {
  if (!ranks || n < 1 || n > 8 || !ranks[0]) return CRABML_HIP_BAD_INPUT;
  crabml_hip_device* dev = ranks[0]->dev;
  CH_LIVE(dev);
  CH_USE(dev);
  CH_FLUSH(dev);
  for (int r = 0; r < n; r++) {
    if (!ranks[r] || ranks[r]->dev != dev || ranks[r]->tp != n || ranks[r]->tp_rank != r || ranks[r]->comm)
      CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "tp_sim: rank %d is not a communicator-less rank %d of %d on this device", r, r, n);
    CH_TRY(check_step(ranks[r], token, pos));
    CH_TRY(set_state(ranks[r], token, pos, 0));
  }
  const int nseg = n_segments(ranks[0]);
  SimPtrs ptrs{};
  for (int r = 0; r < n; r++) ptrs.p[r] = ranks[r]->partial;
  const int dim = (int)ranks[0]->cfg.embedding_dim;
  for (int r = 0; r < n; r++) ranks[r]->attn_variant = variant_of(ranks[r], pos);
  for (int s = 0; s < nseg; s++) {
    for (int r = 0; r < n; r++) CH_TRY(enqueue_segment(ranks[r], s));
    if (n > 1 && s + 1 < nseg) k_sim_allreduce<<<(dim + 255) / 256, 256, 0, dev->stream>>>(ptrs, n, dim);
  }
  for (int r = 0; r < n; r++) ranks[r]->kv_len++;
  if (ranks[0]->split_vocab) {
    // every rank sampled from its own shard: the group's token is the rank-order combination of the pairs
    SimBest sb{};
    for (int r = 0; r < n; r++) {
      if (!ranks[r]->split_vocab) CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "tp_sim: rank %d does not split the vocabulary like rank 0", r);
      sb.best[r] = ranks[r]->am_best;
      sb.token[r] = ranks[r]->state;
      sb.step[r] = ranks[r]->state + 2;
      sb.out_tokens[r] = ranks[r]->out_tokens;
    }
    k_sim_argmax_combine<<<1, 64, 0, dev->stream>>>(sb, n, ranks[0]->out_cap);
    CH_HIP(dev, hipGetLastError());
  }
  if (logits) {
    if (ranks[0]->split_vocab) {
      for (int r = 0; r < n; r++)
        CH_HIP(dev, hipMemcpyAsync(logits + ranks[r]->vocab_off, ranks[r]->logits + ranks[r]->vocab_off, (size_t)ranks[r]->vocab_l * 4,
                                   hipMemcpyDeviceToHost, dev->stream));
    } else {
      CH_HIP(dev, hipMemcpyAsync(logits, ranks[0]->logits, ranks[0]->cfg.vocab_size * 4, hipMemcpyDeviceToHost, dev->stream));
    }
    CH_HIP(dev, hipStreamSynchronize(dev->stream));
  }
  return 0;
}

size_t crabml_hip_llama_kv_len(const crabml_hip_llama_t* c) { return c ? c->kv_len : 0; }

int crabml_hip_llama_reset(crabml_hip_llama_t* c) {
  if (!c) return CRABML_HIP_BAD_INPUT;
  c->kv_len = 0;
  if (c->flash_tick) {  // a step that faulted half-way may have left arrivals behind
    CH_USE(c->dev);
    CH_FLUSH(c->dev);
    CH_HIP(c->dev, hipMemsetAsync(c->flash_tick, 0, (size_t)c->n_kv_l * 4, c->dev->stream));
  }
  return 0;
}

int crabml_hip_llama_debug_kv(crabml_hip_llama_t* c, size_t layer, int32_t which_v, void* dst, size_t nbytes) {
  if (!c || !dst) return CRABML_HIP_BAD_INPUT;
  CH_USE(c->dev);
  CH_FLUSH(c->dev);
  if (layer >= c->cfg.n_layers || nbytes > c->kv_bytes) CH_BAIL(c->dev, CRABML_HIP_BAD_INPUT, "llama debug_kv: bad layer/size");
  CH_HIP(c->dev, hipMemcpyAsync(dst, which_v ? c->vc[layer] : c->kc[layer], nbytes, hipMemcpyDeviceToHost, c->dev->stream));
  CH_HIP(c->dev, hipStreamSynchronize(c->dev->stream));
  return 0;
}

// parity hook (crabml_hip_debug.h): k_attn_flash by itself, on caller-supplied q / K / V
int crabml_hip_debug_flash_attention(crabml_hip_device_t* dev, const float* q, const uint16_t* k, const uint16_t* v, size_t n_heads,
                                     size_t n_kv, size_t head_dim, size_t seq, size_t slices, float* out, float* out2) {
  if (!dev || !q || !k || !v || !out || seq == 0 || n_kv == 0 || n_heads % n_kv != 0) return CRABML_HIP_BAD_INPUT;
  CH_LIVE(dev);
  CH_USE(dev);
  CH_FLUSH(dev);
  const int grp = (int)(n_heads / n_kv), hd = (int)head_dim;
  const FlashFn fn = flash_kernel(grp, hd, false, false), fnt = flash_kernel(grp, hd, false, true);
  const bool ticket_form = out2 != nullptr;
  if (fn == nullptr || slices < 1 || slices > FLASH_MAX_SLICES) CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "debug_flash_attention: unsupported group / head_dim / slices");
  if (raise_dyn_lds(dev, (const void*)fn, (int)flash_lds_bytes(grp, hd)) != hipSuccess ||
      raise_dyn_lds(dev, (const void*)fnt, (int)flash_lds_bytes(grp, hd)) != hipSuccess)
    CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "debug_flash_attention: LDS");
  const size_t nq = n_heads * head_dim * 4, nkv = n_kv * seq * head_dim * 2, npart = n_kv * slices * flash_part_floats(grp, hd) * 4;
  char* base = nullptr;
  const size_t o_q = 0, o_k = align_up(o_q + nq, 256), o_v = align_up(o_k + nkv, 256), o_out = align_up(o_v + nkv, 256),
               o_part = align_up(o_out + nq, 256), o_tick = align_up(o_part + npart, 256), o_pos = o_tick + align_up(n_kv * 4, 256),
               total = o_pos + 256;
  CH_HIP(dev, hipMalloc((void**)&base, total));
  hipStream_t st = dev->stream;
  const int pos = (int)seq - 1;
  hipError_t e = hipMemsetAsync(base + o_tick, 0, n_kv * 4, st);
  if (e == hipSuccess) e = hipMemcpyAsync(base + o_q, q, nq, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(base + o_k, k, nkv, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(base + o_v, v, nkv, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(base + o_pos, &pos, 4, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);  // (`pos` leaves scope)
  if (e == hipSuccess) {
    // the shipped form first (partials, then the merge launch), then the single-launch form (last arriver merges) twice on the
    // same ticket words -- the second launch finds them re-armed by the first; all three write the same `out`
    hipLaunchKernelGGL(fn, dim3((unsigned)(n_kv * slices)), dim3((grp == 8 ? 4 : 8) * 64), (uint32_t)flash_lds_bytes(grp, hd), st,
                       (const float*)(base + o_q), (const unsigned short*)(base + o_k), (const unsigned short*)(base + o_v),
                       (const int*)(base + o_pos), (float*)(base + o_part), (unsigned*)(base + o_tick), (float*)(base + o_out),
                       (signed char*)nullptr, (unsigned short*)nullptr, (void*)nullptr, (int)seq, (int)slices, FLASH_MIN_ROWS);
    if (hd == 128)
      hipLaunchKernelGGL((k_attn_flash_merge<128, false>), dim3((unsigned)n_heads), dim3(128), 0, st, (const float*)(base + o_part),
                         (const int*)(base + o_pos), (float*)(base + o_out), (signed char*)nullptr, (unsigned short*)nullptr, (void*)nullptr, grp,
                         (int)slices, FLASH_MIN_ROWS);
    else
      hipLaunchKernelGGL((k_attn_flash_merge<64, false>), dim3((unsigned)n_heads), dim3(64), 0, st, (const float*)(base + o_part),
                         (const int*)(base + o_pos), (float*)(base + o_out), (signed char*)nullptr, (unsigned short*)nullptr, (void*)nullptr, grp,
                         (int)slices, FLASH_MIN_ROWS);
    if (ticket_form) {
      e = hipMemcpyAsync(out2, base + o_out, nq, hipMemcpyDeviceToHost, st);  // (stream order: before the next launches overwrite it)
      for (int rep = 0; rep < 2 && e == hipSuccess; rep++)
        hipLaunchKernelGGL(fnt, dim3((unsigned)(n_kv * slices)), dim3((grp == 8 ? 4 : 8) * 64), (uint32_t)flash_lds_bytes(grp, hd), st,
                           (const float*)(base + o_q), (const unsigned short*)(base + o_k), (const unsigned short*)(base + o_v),
                           (const int*)(base + o_pos), (float*)(base + o_part), (unsigned*)(base + o_tick), (float*)(base + o_out),
                           (signed char*)nullptr, (unsigned short*)nullptr, (void*)nullptr, (int)seq, (int)slices, FLASH_MIN_ROWS);
    }
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, base + o_out, nq, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(base);
  if (e != hipSuccess) return hip_fail(dev, e, "debug_flash_attention", __FILE__, __LINE__);
  return 0;
}

// parity hook (crabml_hip_debug.h): the fast prompt pass's causal attention kernel by itself
int crabml_hip_debug_flash_attention_rows(crabml_hip_device_t* dev, const float* q, const uint16_t* k, const uint16_t* v, size_t n_heads,
                                          size_t n_kv, size_t head_dim, size_t pos0, size_t rows, size_t seq_cap, float* out) {
  if (!dev || !q || !k || !v || !out || rows == 0 || n_kv == 0 || n_heads % n_kv != 0 || seq_cap < pos0 + rows) return CRABML_HIP_BAD_INPUT;
  if (head_dim != 128 && head_dim != 64) CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "debug_flash_attention_rows: head_dim 64 / 128");
  CH_LIVE(dev);
  CH_USE(dev);
  CH_FLUSH(dev);
  if (raise_dyn_lds(dev, head_dim == 128 ? (const void*)k_attn_flash_rows<128> : (const void*)k_attn_flash_rows<64>,
                    (int)flash_rows_lds_bytes((int)head_dim)) != hipSuccess)
    CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "debug_flash_attention_rows: LDS");
  const size_t seq = seq_cap;
  const size_t nq = rows * n_heads * head_dim * 4, nkv = n_kv * seq * head_dim * 2;
  const size_t o_q = 0, o_k = align_up(o_q + nq, 256), o_v = align_up(o_k + nkv, 256), o_out = align_up(o_v + nkv, 256),
               o_pos = align_up(o_out + nq, 256), total = o_pos + 256;
  char* base = nullptr;
  CH_HIP(dev, hipMalloc((void**)&base, total));
  hipStream_t st = dev->stream;
  const int p0 = (int)pos0;
  hipError_t e = hipMemcpyAsync(base + o_q, q, nq, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(base + o_k, k, nkv, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(base + o_v, v, nkv, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(base + o_pos, &p0, 4, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e == hipSuccess) {
    const dim3 fg((unsigned)((rows + 63) / 64), (unsigned)n_heads);
    if (head_dim == 128)
      k_attn_flash_rows<128><<<fg, 512, flash_rows_lds_bytes(128), st>>>((const float*)(base + o_q), (const unsigned short*)(base + o_k), (const unsigned short*)(base + o_v),
                                                 (const int*)(base + o_pos), (float*)(base + o_out), (int)n_heads, (int)n_kv, (int)seq, (int)rows);
    else
      k_attn_flash_rows<64><<<fg, 512, flash_rows_lds_bytes(64), st>>>((const float*)(base + o_q), (const unsigned short*)(base + o_k), (const unsigned short*)(base + o_v),
                                                (const int*)(base + o_pos), (float*)(base + o_out), (int)n_heads, (int)n_kv, (int)seq, (int)rows);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, base + o_out, nq, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(base);
  if (e != hipSuccess) return hip_fail(dev, e, "debug_flash_attention_rows", __FILE__, __LINE__);
  return 0;
}

}  // extern "C"

// ==============================================================================================================
// The decode context as lazy.hip drives it: built from the buffers a recorded token of the reference's unchanged runner
// names (its weight handles, its own KV caches), stepped one SEGMENT at a time while the host is still recording the
// next one.  Everything below runs the same enqueue_segment as crabml_hip_llama_forward.
// ==============================================================================================================
namespace crabml_hip {

int lazy_ctx_create(crabml_hip_device* dev, const LazyModel& m, crabml_hip_llama** out) {
  *out = nullptr;
  crabml_hip_llama_weights_t w{};
  w.token_embed = m.token_embed;
  w.rms_att_weight = m.rms_att.data();
  w.rms_ffn_weight = m.rms_ffn.data();
  w.wq = m.wq.data();
  w.wk = m.wk.data();
  w.wv = m.wv.data();
  w.wo = m.wo.data();
  w.ffn_gate_weight = m.gate.data();
  w.ffn_down_weight = m.down.data();
  w.ffn_up_weight = m.up.data();
  w.rms_final_weight = m.rms_final;
  w.output_weight = m.output;
  const size_t L = m.cfg.n_layers;
  if (m.rms_att.size() != L || m.rms_ffn.size() != L || m.wq.size() != L || m.wk.size() != L || m.wv.size() != L || m.wo.size() != L ||
      m.gate.size() != L || m.down.size() != L || m.up.size() != L || m.kc.size() != L || m.vc.size() != L)
    return CRABML_HIP_BAD_INPUT;
  return llama_create_impl(dev, &m.cfg, &w, m.kc.data(), m.vc.data(), out);
}

void lazy_ctx_destroy(crabml_hip_llama* c) { (void)crabml_hip_llama_destroy(c); }

bool lazy_ctx_orphaned(const crabml_hip_llama* c) {
  // one representative of the model (the first layer's wq) and one of the runner (its first K cache): a handle whose references are
  // all the context's own holds has been released by the host
  auto sole = [&](const crabml_hip_buf* b) {
    if (!b) return false;
    int holds = 0;
    for (const crabml_hip_buf* h : c->held)
      if (h == b) holds++;
    return holds > 0 && b->refcnt.load() <= holds;
  };
  return sole(c->wq.empty() ? nullptr : c->wq[0]) || (c->ext_kv && sole(c->ext_kc0));
}

int lazy_ctx_n_segments(const crabml_hip_llama* c) { return n_segments(c); }

int lazy_ctx_begin(crabml_hip_llama* c, size_t token, size_t pos) {
  if (c->dev->dry) return 0;
  // A token whose shadow is dropped half-way (lazy.hip: the op stream left the template) never reaches the sampler launch that
  // advances the step serial on the device -- and the epochs of the in-launch hand-offs (norm gathers, Q8_K exchanges) are derived
  // from it: the NEXT token's first segments would match the dropped token's granules.  So here the host owns the serial: every
  // begin sets a fresh even value, the sampler's own + 1 lands on the odd one in between.
  c->lazy_serial += 2;
  c->out_seq++;
  k_set_state5<<<1, 1, 0, c->dev->stream>>>(c->state, (int)token, (int)pos, 0, (int)c->lazy_serial, (int)c->out_seq);
  CH_HIP(c->dev, hipGetLastError());
  c->attn_variant = variant_of(c, pos);
  c->kv_len = pos + 1;
  return 0;
}

int lazy_ctx_segment(crabml_hip_llama* c, int seg) {
  if (c->dev->dry) return 0;
  return enqueue_segment(c, seg);
}

// the whole step at once: the context's captured graph (false: this context has none -- the caller enqueues segment by segment)
bool lazy_ctx_has_graph(const crabml_hip_llama* c) { return !c->dev->dry && c->use_graph && c->exec[0] != nullptr; }
int lazy_ctx_step(crabml_hip_llama* c, size_t pos) {
  if (c->dev->dry) return 0;
  return run_step(c, pos);
}

// dst = the logits of the last step (device to device; for a handle the host kept and uses as an operand)
int lazy_ctx_copy_logits(crabml_hip_llama* c, float* dst) {
  if (c->dev->dry) return 0;
  CH_HIP(c->dev, hipMemcpyAsync(dst, c->logits, c->cfg.vocab_size * 4, hipMemcpyDeviceToDevice, c->dev->stream));
  return 0;
}

int lazy_ctx_final_norm(crabml_hip_llama* c, float* dst) {
  crabml_hip_device* dev = c->dev;
  if (dev->dry) return 0;
  const int dim = (int)c->cfg.embedding_dim;
  const size_t lds = norm_lds_bytes(dim);
  // the order of the step's own final norm: the reference's scan on a strict-order device, the fast split otherwise
  const int half = dev->strict_order ? 0 : 1;
  if (dim <= 4096)
    k_norm_f32<4><<<1, 1024, lds, dev->stream>>>(c->x, nullptr, (const float*)c->rms_final->ptr, dim, c->cfg.rms_norm_eps, dst, half);
  else
    k_norm_f32<12><<<1, 1024, lds, dev->stream>>>(c->x, nullptr, (const float*)c->rms_final->ptr, dim, c->cfg.rms_norm_eps, dst, half);
  CH_HIP(dev, hipGetLastError());
  return 0;
}

// the fault word of the in-launch gathers travels with the sync the caller performs anyway: request it before, read it after
int lazy_ctx_fault_request(crabml_hip_llama* c) {
  crabml_hip_device* dev = c->dev;
  if (dev->dry) return 0;
  int* h = c->h_state + crabml_hip_llama::H_STATE_SLOTS * 4;  // pinned, behind the state ring
  CH_HIP(dev, hipMemcpyAsync(h, c->state + 5, sizeof(int), hipMemcpyDeviceToHost, dev->stream));
  return 0;
}
// the logits of the last final segment in pinned host memory: spins on the flag the step's last kernel raises (bounded; then the
// stream is drained the ordinary way).  nullptr: this context has no host copy.
const float* lazy_ctx_wait_logits(crabml_hip_llama* c, int* fault) {
  if (c->dev->dry || c->host_logits == nullptr || c->out_seq == 0) return nullptr;
  volatile unsigned* flag = (volatile unsigned*)(c->host_logits + c->cfg.vocab_size);
  const unsigned want = c->out_seq;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0; __atomic_load_n((const unsigned*)flag, __ATOMIC_ACQUIRE) != want; spins++) {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
    if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
      if (hipStreamSynchronize(c->dev->stream) != hipSuccess) return nullptr;
      if (__atomic_load_n((const unsigned*)flag, __ATOMIC_ACQUIRE) != want) return nullptr;
      break;
    }
  }
  *fault = (int)flag[1];
  return c->host_logits;
}
bool lazy_ctx_has_host_logits(const crabml_hip_llama* c) { return c != nullptr && (c->host_logits != nullptr || c->dev->dry); }
int lazy_ctx_fault_value(const crabml_hip_llama* c) { return c->dev->dry ? 0 : c->h_state[crabml_hip_llama::H_STATE_SLOTS * 4]; }

}  // namespace crabml_hip
