// lazy.hpp -- the recorded-op queue behind the Tensor entry points and the matcher that turns the op stream of an UNCHANGED
// `Llama2Runner<T>::forward` (crabml-llama2/src/llama2.rs:184-281, 527-638) into the fused decode step.
//
// Why: the reference's generic runner issues ~31 Tensor calls per layer.  One launch per call makes batch-1 decode
// host-launch-bound (251 tok/s on the 8B shape against 756 for the fused step), and the fused step was only reachable through an
// entry point of its own (crabml_hip_llama_*) that the reference never calls.  With this queue the per-op entry points RECORD
// (op, operand handles, scalars) and return a handle whose memory is not bound yet; the only points where the host can observe
// data -- export / sync / debug hooks, the same contract as the wgpu backend (crabml-wgpu/src/wgpu_tensor.rs:293-333) -- run
// the queue.  Two ways to run it:
//   * op by op, exactly the launches of the eager path (anything the matcher does not know, CRABML_HIP_FLAG_LAZY_NO_FUSION);
//   * as the fused step: the first complete token of a Llama model is parsed against the op sequence of forward_llama /
//     forward_multi_query_attention / forward_ffn (every operand identity, shape and scalar is checked: see learn_token), a
//     decode context is built over the runner's OWN weight and KV-cache buffers, and a template of the token is kept.  From the
//     next token on every recorded op is compared with the template as it arrives, and as soon as the ops of a SEGMENT are in
//     (attention half / FFN half of a layer / classifier) its fused launches are enqueued -- the GPU works on layer l while the
//     host is still recording layer l + 1.  The fused launches write only the context's private buffers and the KV-cache row
//     the recorded `concatenate` ops name; the recorded ops stay in the queue until the token is complete.  Then the token is
//     COMMITTED (its ops are dropped; outputs the host still holds are bound: the logits were written in place, the final
//     normalized row is produced on demand).  An op that deviates from the template before that ABORTS the shadow: the queue is
//     simply replayed op by op at the next flush, which also overwrites the KV rows.  So the host sees per-op SEMANTICS always --
//     and, on a strict-order device, per-op BITS always.  On the fast device a committed token carries the fused step's
//     re-associated sums and a replayed one the per-op kernels': both inside FAST_TOL of the reference, not identical to each other
//     (a token's last bits depend on whether the host looked in the middle of it; include/crabml_hip.h says the same).
//   * The whole-step launch at the `go` op trusts layer 0: token, position and the FIRST layer's cache handles are verified when
//     the graph goes out; the deeper layers' concatenate ops are only compared afterwards.  A host that swapped a deeper layer's
//     cache handle while keeping layer 0's makes the token deviate there -- the shadow is dropped and the queue replayed as always
//     -- but the graph has by then written row `pos` of the LEARNED context's caches for those layers (buffers the context
//     retains: never a use-after-free, and a row at or past their live length).
//   * Contexts: the one being served plus up to two parked ones (runners taking turns on one device keep theirs); a context whose
//     model or caches the host has released is destroyed at the next flush, every idle one when a device allocation fails.
#pragma once
#include <unordered_map>

#include "common.hpp"

struct crabml_hip_llama;

namespace crabml_hip {

enum LazyKind : uint8_t {
  LZ_DUP = 1,
  LZ_CONTIGUOUS,
  LZ_CONCAT,
  LZ_COPY_ROW,  // copy_rows_from with exactly one row (anything else runs eagerly)
  LZ_ROPE,
  LZ_RMS_NORM,
  LZ_SOFTMAX,
  LZ_SILU,
  LZ_GELU,
  LZ_MUL,
  LZ_ADD,
  LZ_SCALE,
  LZ_MATMUL_VEC,
  LZ_BATCH_MATMUL,
};

// One recorded Tensor call.  a: the tensor the method was called on (in-place target / lhs / concatenate + copy destination /
// the weight of matmul_vec); b: the argument tensor; out: the fresh result.  Every non-null handle is retained once.
//   DUP            a -> out
//   CONTIGUOUS     a -> out            s[0..2] shape, s[3..5] strides (padded to 3-D)
//   CONCAT         a <- b              s[0..2] rhs shape, s[3..5] dst strides, s[6..8] rhs strides, s[9] dst offset,
//                                      s[10] axis (of the padded 3-D form), s[11] dst length on that axis before the call
//   COPY_ROW       a <- b              s[0] cols, s[1] row
//   ROPE           a                   s[0] n_batch, s[1] bi_stride, s[2] head_dim, s[3] mode, s[4] pos, s[5] rope_dims
//   RMS_NORM       a                   s[0] rows, s[1] cols, f eps
//   SOFTMAX        a                   s[0] rows, s[1] cols
//   SILU / GELU    a                   s[0] n
//   MUL / ADD      a op= b             s[0] na, s[1] nb
//   SCALE          a                   s[0] na, f
//   MATMUL_VEC     a (w) x b -> out    s[0] m, s[1] k, s[2] batch
//   BATCH_MATMUL   a x b -> out        s[0] ba, s[1] m, s[2] k, s[3] bb, s[4] n, s[5..7] strides of b
struct LazyOp {
  uint8_t kind = 0;
  crabml_hip_buf* a = nullptr;
  crabml_hip_buf* b = nullptr;
  crabml_hip_buf* out = nullptr;
  size_t s[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  float f = 0.f;
};

// what learn_token reads off a recorded token: the arguments of a decode context over the runner's own buffers
struct LazyModel {
  crabml_hip_llama_config_t cfg{};
  const crabml_hip_buf* token_embed = nullptr;
  const crabml_hip_buf* rms_final = nullptr;
  const crabml_hip_buf* output = nullptr;
  std::vector<const crabml_hip_buf*> rms_att, rms_ffn, wq, wk, wv, wo, gate, down, up;
  std::vector<crabml_hip_buf*> kc, vc;  // the runner's KV caches: [n_kv_heads][seq_len][head_dim], f16 or f32
  bool same_buffers(const LazyModel& o) const {
    return token_embed == o.token_embed && rms_final == o.rms_final && output == o.output && rms_att == o.rms_att && rms_ffn == o.rms_ffn &&
           wq == o.wq && wk == o.wk && wv == o.wv && wo == o.wo && gate == o.gate && down == o.down && up == o.up && kc == o.kc && vc == o.vc;
  }
};

// scalars of a template op that follow the token instead of being constant
enum LazyRule : uint8_t { LR_NONE = 0, LR_TOKEN, LR_ROPE, LR_CONCAT, LR_BMM_N, LR_SOFTMAX, LR_BMM_K };

struct TmplOp {
  uint8_t kind = 0, rule = LR_NONE;
  bool a_new = false;   // `a` is a buffer the host allocated for this token (the destination of copy_rows_from)
  int16_t seg_end = -1; // the decode segment whose ops are complete with this op
  bool go = false;      // token, position and the first layer's caches are verified with this op: the whole step may be launched
  // operands: slot >= 0 = the token's own buffer number `slot`; -1 = the persistent buffer p*; -2 = none
  int32_t sa = -2, sb = -2, so = -2;
  const crabml_hip_buf *pa = nullptr, *pb = nullptr;
  size_t s[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  float f = 0.f;
};

struct LazyStats {  // crabml_hip_debug_lazy_stats
  uint64_t recorded = 0;       // ops recorded
  uint64_t replayed = 0;       // ops run one launch at a time
  uint64_t fused_tokens = 0;   // tokens committed from the fused step
  uint64_t fused_ops = 0;      // recorded ops those tokens replaced
  uint64_t segments = 0;       // fused segments enqueued
  uint64_t aborts = 0;         // tokens whose shadow was dropped (deviation from the template, flush in mid-token)
  uint64_t learned = 0;        // decode contexts built from a recorded token
  uint64_t deferred_bound = 0; // final-norm rows produced on demand for a handle the host kept
  uint64_t wait_ns = 0;        // host time blocked in export / sync (the GPU still working: the host was ahead)
  uint64_t pinned_exports = 0; // exports served from the logits copy that was requested when the token committed
  uint64_t reactivated = 0;    // parked decode contexts taken back into service (a host whose runners take turns on one device)
  uint64_t reaped = 0;         // decode contexts dropped because the host released the model or the caches they serve
};

// a decode context that is not the one being served: the host switched to another runner / model on this device.  Kept (a small
// LRU) so that runners taking turns do not rebuild their contexts -- allocations and a graph capture -- at every switch.
struct ParkedModel {
  crabml_hip_llama* ctx = nullptr;
  LazyModel model;
  std::vector<TmplOp> tmpl;
  std::vector<int> mentions;
  int slot_xnorm = -1, slot_xfinal = -1, slot_logits = -1;
};
constexpr size_t LAZY_PARKED_MAX = 2;

struct LazyState {
  std::vector<LazyOp> q;
  // the learned model
  crabml_hip_llama* ctx = nullptr;
  LazyModel model;
  std::vector<TmplOp> tmpl;
  std::vector<int> mentions;  // per slot: how often the token's ops name it (= references the queue holds on it)
  int slot_xnorm = -1, slot_xfinal = -1, slot_logits = -1;
  std::vector<ParkedModel> parked;  // least recently used first
  uint64_t unfusable_uid = 0;  // wq[0] (its uid: addresses are re-used) of a model the decode context refused
  // the token being shadowed
  bool tracking = false;
  size_t next = 0;  // template index of the next expected op
  std::vector<crabml_hip_buf*> slots;
  size_t token = 0, pos = 0;
  bool pos_known = false, begun = false, dead = false;
  // handles the host kept whose value is the final norm of the context's residual stream (bound on demand)
  crabml_hip_buf* deferred[3] = {nullptr, nullptr, nullptr};  // (buf->deferred: 1 = the final norm row, 2 = the logits)
  bool whole_step = false;  // the token being shadowed was launched as one graph (at its `go` op)
  // the logits of a committed token are sent to pinned host memory by the step's last kernels (the runner exports them next,
  // llama2.rs:208): export() of that very handle, unmodified, is then a wait on a flag + a host copy
  size_t pin_n = 0;
  crabml_hip_buf* pin_buf = nullptr;  // retained while the copy is valid for it
  uint64_t pin_version = 0;
  int pin_kind = 0;  // 1: the handle's value is (on its way) in the context's pinned host copy
  bool check_fault = false;  // a committed token's gather-fault word has not been looked at yet
  bool fault_requested = false;
  LazyStats stats;
};

inline uint16_t host_f2h(float f) {
  _Float16 h = (_Float16)f;  // IEEE RNE
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}
inline float host_h2f(uint16_t u) {
  _Float16 h;
  memcpy(&h, &u, 2);
  return (float)h;
}

// ---- lazy.hip
// the rhs of matmul_vec quantized to `qt` (cached per buffer version: q/k/v and gate/up share one pass)
int ensure_act(crabml_hip_device* dev, const crabml_hip_buf* x, size_t b, size_t k, uint32_t qt, const void** act);
int lazy_record(crabml_hip_device* dev, const LazyOp& op);  // retains the operands, feeds the matcher
int lazy_exec(crabml_hip_device* dev, LazyOp& op);          // the launches of one op, immediately (the eager path uses it too)
int lazy_resolve(crabml_hip_device* dev);                   // binds deferred handles that are still alive
void lazy_destroy(crabml_hip_device* dev);
// drops every decode context nothing is being served from right now (parked ones; the active one between tokens): called when a
// device allocation fails, before the retry.  Returns the number of contexts dropped.
int lazy_release_contexts(crabml_hip_device* dev);
inline void lazy_use(crabml_hip_device* dev, const crabml_hip_buf* b) {
  if (b && b->deferred) (void)lazy_resolve(dev);
}
// around a sync: did a committed token's in-launch gather time out?  (the decode context's fault word; requested before the
// sync, looked at after it)
// export(): was a host copy of this buffer's first n floats requested when its token committed, and is the buffer unchanged?
// 1: the context's kernels send it to pinned memory and raise a flag (lazy_export_wait copies it out); 0: no
int lazy_pinned_kind(crabml_hip_device* dev, const crabml_hip_buf* b, size_t n);
int lazy_export_wait(crabml_hip_device* dev, float* dst, size_t n);  // kind 1: wait for the flag, copy, report a gather fault
int lazy_fault_request(crabml_hip_device* dev);
int lazy_fault_check(crabml_hip_device* dev);

// ---- fused.hip: the decode context as the matcher drives it
int lazy_ctx_create(crabml_hip_device* dev, const LazyModel& m, crabml_hip_llama** out);
void lazy_ctx_destroy(crabml_hip_llama* c);
// the context is the only remaining owner of the weights or of the KV caches it was built over (the host dropped its model or its
// runner): nothing will ever be served from it again, and it pins their device memory
bool lazy_ctx_orphaned(const crabml_hip_llama* c);
int lazy_ctx_begin(crabml_hip_llama* c, size_t token, size_t pos);            // token id / position of the step -> device state
int lazy_ctx_segment(crabml_hip_llama* c, int seg);                           // enqueue one segment
bool lazy_ctx_has_graph(const crabml_hip_llama* c);
int lazy_ctx_step(crabml_hip_llama* c, size_t pos);                           // the whole step from the context's graph
int lazy_ctx_copy_logits(crabml_hip_llama* c, float* dst);                    // dst = the last step's logits (device to device)
int lazy_ctx_final_norm(crabml_hip_llama* c, float* dst);                     // dst = rms_norm(residual) * rms_final
const float* lazy_ctx_wait_logits(crabml_hip_llama* c, int* fault);  // spins; nullptr = no host copy of the logits
bool lazy_ctx_has_host_logits(const crabml_hip_llama* c);
int lazy_ctx_fault_request(crabml_hip_llama* c);
int lazy_ctx_fault_value(const crabml_hip_llama* c);
int lazy_ctx_n_segments(const crabml_hip_llama* c);

}  // namespace crabml_hip
