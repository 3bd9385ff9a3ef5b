// lazy.hip -- the recorded-op queue of a device, the launches of each recorded op, and the matcher that serves the op stream
// of the reference's unchanged runner from the fused decode step.  Design and invariants: lazy.hpp.
#include <chrono>
#include <cmath>

#include "kernels.hpp"
#include "lazy.hpp"

namespace crabml_hip {

// ---- memory binding ------------------------------------------------------------------------------------------
crabml_hip_buf* buf_new_unbound(crabml_hip_device* dev, uint32_t dtype, size_t n_elems, size_t bytes) {
  crabml_hip_buf* b = new crabml_hip_buf();
  b->dev = dev;
  b->dtype = dtype;
  b->n_elems = n_elems;
  b->bytes = bytes;
  b->wl = weight_layout(dtype, n_elems);
  return b;
}

int ensure_mem(crabml_hip_device* dev, crabml_hip_buf* b) {
  if (b->ptr) return 0;
  CH_TRY(pool_alloc(dev, b->bytes, &b->ptr, &b->cap));
  if (b->zero_init && b->bytes && !dev->dry) CH_HIP(dev, hipMemsetAsync(b->ptr, 0, b->bytes, dev->stream));
  b->zero_init = false;
  return 0;
}

// ---- quantize the rhs of matmul_vec (cached per buffer version) ---------------------------------------------------
int ensure_act(crabml_hip_device* dev, const crabml_hip_buf* x_, size_t b, size_t k, uint32_t qt, const void** act) {
  crabml_hip_buf* x = const_cast<crabml_hip_buf*>(x_);
  if (qt == CRABML_HIP_F32) {
    *act = x->ptr;  // CpuTensorBuf::quantize(F32) is a copy (buf/api.rs:197)
    return 0;
  }
  ActLayout al = act_layout(qt, k);
  size_t need = al.total * b;
  if (x->qc.qtype == qt && x->qc.version == x->version && x->qc.n == b * k && x->qc.k == k && x->qc.ptr) {
    *act = x->qc.ptr;
    return 0;
  }
  if (x->qc.cap < need) {
    if (x->qc.ptr) pool_free(dev, x->qc.ptr, x->qc.cap);
    x->qc.ptr = nullptr;
    x->qc.cap = 0;
    CH_TRY(pool_alloc(dev, need, &x->qc.ptr, &x->qc.cap));
  }
  launch_quantize_act_rows(dev->stream, qt, (const float*)x->ptr, b, k, x->qc.ptr);  // one launch for the b rows
  x->qc.qtype = qt;
  x->qc.version = x->version;
  x->qc.n = b * k;
  x->qc.k = k;
  *act = x->qc.ptr;
  return 0;
}

static float gelu_single(float x) {  // gelu.rs:19-22
  const float COEF_A = 0.044715f;
  const float S = (float)0.7978845608028654;
  return 0.5f * x * (1.0f + tanhf(S * x * (1.0f + COEF_A * x * x)));
}

// ---- the launches of one recorded op (arguments were validated when the op was recorded) ----------------------------------
int lazy_exec(crabml_hip_device* dev, LazyOp& o) {
  if (o.a) CH_TRY(ensure_mem(dev, o.a));
  if (o.b) CH_TRY(ensure_mem(dev, o.b));
  if (o.out) CH_TRY(ensure_mem(dev, o.out));
  if (dev->lz) dev->lz->stats.replayed++;
  if (dev->dry) {
    if (o.kind != LZ_DUP && o.kind != LZ_CONTIGUOUS && o.kind != LZ_MATMUL_VEC && o.kind != LZ_BATCH_MATMUL) touch(o.a);
    return 0;
  }
  hipStream_t st = dev->stream;
  const size_t* s = o.s;
  switch (o.kind) {
    case LZ_DUP:
      if (o.a->n_elems) CH_HIP(dev, hipMemcpyAsync(o.out->ptr, o.a->ptr, o.a->n_elems * 4, hipMemcpyDeviceToDevice, st));
      break;
    case LZ_CONTIGUOUS:
      launch_contiguous(st, o.a->ptr, o.out->ptr, o.a->dtype == CRABML_HIP_F32 ? 4 : 2, s, s + 3);
      break;
    case LZ_CONCAT:
      launch_concatenate(st, o.a->ptr, o.a->dtype == CRABML_HIP_F16, s[9], s + 3, o.b->ptr, o.b->dtype == CRABML_HIP_F16, s, s + 6);
      touch(o.a);
      break;
    case LZ_COPY_ROW:
      launch_dequant_row(st, o.b, s[1] * s[0], s[0], o.a->ptr, o.a->dtype == CRABML_HIP_F16);
      touch(o.a);
      break;
    case LZ_ROPE: {
      const size_t n_batch = s[0], bi_stride = s[1], head_dim = s[2], mode = s[3], pos = s[4], rope_dims = s[5];
      const size_t npairs = mode == 0 ? (rope_dims + 1) / 2 : rope_dims / 2;
      const size_t n_heads = bi_stride / head_dim;
      for (size_t bi = 0; bi < n_batch; bi++) {
        RopeTable tab;
        size_t p = pos + bi;  // rope.rs:35-36
        if (mode == 0) {      // rope.rs:47-63: theta is an iterated f32 product, base 10000 hard-coded
          float theta_scale = powf(10000.0f, -2.0f / (float)head_dim);
          float theta = (float)p;
          for (size_t i = 0; i < npairs; i++) {
            tab.cs[2 * i] = cosf(theta);
            tab.cs[2 * i + 1] = sinf(theta);
            theta *= theta_scale;
          }
        } else {  // rope.rs:65-80
          for (size_t i = 0; i < npairs; i++) {
            float fe = 2.0f * (float)i / (float)head_dim;
            float timescale = powf(10000.0f, fe);
            float theta = (float)p / timescale;
            tab.cs[2 * i] = cosf(theta);
            tab.cs[2 * i + 1] = sinf(theta);
          }
        }
        launch_rope(st, (float*)o.a->ptr + bi * bi_stride, n_heads, head_dim, (int)mode, rope_dims, tab);
      }
      touch(o.a);
      break;
    }
    case LZ_RMS_NORM:
      launch_rms_norm(st, (float*)o.a->ptr, s[0], s[1], o.f);
      touch(o.a);
      break;
    case LZ_SOFTMAX:
      launch_softmax(st, (float*)o.a->ptr, s[0], s[1], dev->exp_table);
      touch(o.a);
      break;
    case LZ_SILU:
      launch_silu(st, (float*)o.a->ptr, s[0], dev->exp_table);
      touch(o.a);
      break;
    case LZ_GELU:
      if (!dev->gelu_table) {  // OnceLock<Vec<f16>> (cpu_device.rs:117-124)
        std::vector<uint16_t> tab(65536);
        for (uint32_t i = 0; i < 65536; i++) tab[i] = host_f2h(gelu_single(host_h2f((uint16_t)i)));
        CH_HIP(dev, hipMalloc((void**)&dev->gelu_table, 65536 * 2));
        CH_HIP(dev, hipMemcpyAsync(dev->gelu_table, tab.data(), 65536 * 2, hipMemcpyHostToDevice, st));
        CH_HIP(dev, hipStreamSynchronize(st));
      }
      launch_gelu(st, (float*)o.a->ptr, s[0], dev->gelu_table);
      touch(o.a);
      break;
    case LZ_MUL:
    case LZ_ADD:
      launch_binary(st, o.kind == LZ_MUL ? 1 : 0, (float*)o.a->ptr, s[0], (const float*)o.b->ptr, s[1]);
      touch(o.a);
      break;
    case LZ_SCALE:
      launch_scale(st, (float*)o.a->ptr, s[0], o.f);
      touch(o.a);
      break;
    case LZ_MATMUL_VEC: {
      const crabml_hip_buf* w = o.a;
      const size_t m = s[0], k = s[1], b = s[2];
      const uint32_t qt = vec_dot_rhs_dtype(w->dtype);
      const void* act = nullptr;
      CH_TRY(ensure_act(dev, o.b, b, k, qt, &act));
      crabml_hip_device::ProfRec rec{};
      const bool prof = dev->prof_on && !dev->strict_order;  // the strict-order kernels carry no events: take none
      if (prof)
        CH_TRY(prof_begin(dev, &rec, w->dtype, 0,
                          (double)b * ((double)m * (double)(k / block_elems(w->dtype)) * (double)block_bytes(w->dtype) + 4.0 * k + 4.0 * m)));
      int rc = dev->strict_order ? launch_gemv_strict(dev, w, m, k, act, b, (float*)o.out->ptr)
                                 : launch_gemv(dev, w, m, k, act, b, (float*)o.out->ptr, prof ? &rec : nullptr);
      if (prof) {
        if (rc == 0) {
          CH_TRY(prof_end(dev, &rec));
        } else {  // nothing was recorded: hand the pair back
          dev->prof_free_events.push_back(rec.e0);
          dev->prof_free_events.push_back(rec.e1);
        }
      }
      if (rc != 0) return rc;
      break;
    }
    case LZ_BATCH_MATMUL:
      launch_batch_matmul(st, (const float*)o.a->ptr, s[0], s[1], s[2], o.b->ptr, o.b->dtype == CRABML_HIP_F16, s[3], s[4], s[5], s[6], s[7],
                          (float*)o.out->ptr);
      break;
    default: CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "lazy queue: unknown op kind %d", (int)o.kind);
  }
  return 0;
}

namespace {

void release_op(LazyOp& o) {
  if (o.a) crabml_hip_buf_release(o.a);
  if (o.b) crabml_hip_buf_release(o.b);
  if (o.out) crabml_hip_buf_release(o.out);
  o.a = o.b = o.out = nullptr;
}

// run the queue one op at a time, in order
int run_queue(crabml_hip_device* dev, LazyState& L) {
  int rc = 0;
  std::vector<LazyOp> q;
  q.swap(L.q);  // (an op's release may re-enter nothing, but keep the member consistent while we iterate)
  for (size_t i = 0; i < q.size(); i++) {
    if (rc == 0) rc = lazy_exec(dev, q[i]);
    release_op(q[i]);
  }
  q.clear();
  q.swap(L.q);  // keep the capacity
  return rc;
}

// ---- learning: parse one complete token of Llama2Runner::forward (llama2.rs:184-281, 527-638; n_batch = 1) ----------------
struct Learner {
  const std::vector<LazyOp>& q;
  size_t i;
  std::vector<TmplOp> T;
  std::unordered_map<const crabml_hip_buf*, int> slot;
  std::vector<int> mentions;
  bool ok = true;

  const LazyOp* take(uint8_t kind) {
    if (i >= q.size() || q[i].kind != kind) return nullptr;
    return &q[i++];
  }
  int ref(const crabml_hip_buf* b, const crabml_hip_buf** pers, bool fresh) {
    if (!b) return -2;
    auto it = slot.find(b);
    if (it != slot.end()) {
      if (fresh) ok = false;  // a "fresh" handle that was seen before
      mentions[it->second]++;
      return it->second;
    }
    if (fresh) {
      const int sidx = (int)mentions.size();
      slot[b] = sidx;
      mentions.push_back(1);
      return sidx;
    }
    if (pers) *pers = b;
    return -1;
  }
  void push(const LazyOp& o, uint8_t rule = LR_NONE, bool a_new = false, int seg_end = -1) {
    TmplOp t;
    t.kind = o.kind;
    t.rule = rule;
    t.a_new = a_new;
    t.seg_end = (int16_t)seg_end;
    t.sa = ref(o.a, &t.pa, a_new);
    t.sb = ref(o.b, &t.pb, false);
    if (o.out) t.so = ref(o.out, nullptr, true);
    memcpy(t.s, o.s, sizeof t.s);
    t.f = o.f;
    T.push_back(t);
  }
};

bool is_f32_vec(const crabml_hip_buf* b, size_t n) { return b && b->dtype == CRABML_HIP_F32 && b->n_elems == n; }
bool s3(const size_t* s, size_t a, size_t b, size_t c) { return s[0] == a && s[1] == b && s[2] == c; }

#define NEED(c) \
  do {          \
    if (!(c)) return false; \
  } while (0)

bool learn_token(Learner& P, LazyModel& M, int* slot_xnorm, int* slot_xfinal, int* slot_logits) {
  const LazyOp* o = P.take(LZ_COPY_ROW);  // x = alloc([1, dim]); x.copy_rows_from(token_embed, [tok])   llama2.rs:222-223
  NEED(o);
  const crabml_hip_buf* x = o->a;
  const crabml_hip_buf* emb = o->b;
  const size_t dim = o->s[0];
  NEED(dim && dim % 32 == 0 && is_f32_vec(x, dim) && emb != x && emb->n_elems % dim == 0);
  const size_t vocab_e = emb->n_elems / dim;
  NEED(o->s[1] < vocab_e);
  P.push(*o, LR_TOKEN, true);
  M.token_embed = emb;
  size_t pos = 0, hd = 0, kv_dim = 0, hidden = 0, n_heads = 0, n_kv = 0, seq = 0, rope_dim = 0;
  uint32_t kv_dtype = 0;
  float eps = 0.f;
  int l = 0;
  while (P.i < P.q.size() && P.q[P.i].kind == LZ_DUP) {
    const bool first = l == 0;
    o = P.take(LZ_DUP);  // x_attn_orig = x.dup()                                        :228
    NEED(o->a == x);
    const crabml_hip_buf* xo = o->out;
    P.push(*o);
    o = P.take(LZ_RMS_NORM);  // x.rms_norm_inplace(eps)                                 :231
    NEED(o && o->a == x && o->s[0] == 1 && o->s[1] == dim);
    if (first) eps = o->f;
    NEED(o->f == eps);
    P.push(*o);
    o = P.take(LZ_MUL);  // x.mul_inplace(rms_att_weight[l])                             :232
    NEED(o && o->a == x && o->s[0] == dim && o->s[1] == dim && is_f32_vec(o->b, dim));
    M.rms_att.push_back(o->b);
    P.push(*o);
    o = P.take(LZ_MATMUL_VEC);  // q = wq.matmul_vec(x)                                  :244
    NEED(o && o->b == x && o->s[0] == dim && o->s[1] == dim && o->s[2] == 1);
    const crabml_hip_buf* qv = o->out;
    M.wq.push_back(o->a);
    P.push(*o);
    o = P.take(LZ_MATMUL_VEC);  // k                                                     :245
    NEED(o && o->b == x && o->s[1] == dim && o->s[2] == 1);
    if (first) kv_dim = o->s[0];
    NEED(o->s[0] == kv_dim && kv_dim);
    const crabml_hip_buf* kv = o->out;
    M.wk.push_back(o->a);
    P.push(*o);
    o = P.take(LZ_MATMUL_VEC);  // v                                                     :246
    NEED(o && o->b == x && o->s[0] == kv_dim && o->s[1] == dim && o->s[2] == 1);
    const crabml_hip_buf* vv = o->out;
    M.wv.push_back(o->a);
    P.push(*o);
    o = P.take(LZ_ROPE);  // q.rope_inplace(Llama, pos, rope_dim)                        :255
    NEED(o && o->a == qv && o->s[0] == 1 && o->s[1] == dim && o->s[3] == CRABML_HIP_ROPE_LLAMA);
    if (first) {
      hd = o->s[2];
      pos = o->s[4];
      rope_dim = o->s[5];
      NEED(hd && dim % hd == 0 && kv_dim % hd == 0);
      n_heads = dim / hd;
      n_kv = kv_dim / hd;
      NEED(n_heads > 1 && n_kv >= 1);  // (one head: the transposed q view is already contiguous -- another op sequence)
    }
    NEED(o->s[2] == hd && o->s[4] == pos && o->s[5] == rope_dim);
    P.push(*o, LR_ROPE);
    o = P.take(LZ_ROPE);  // k.rope_inplace                                               :256
    NEED(o && o->a == kv && o->s[0] == 1 && o->s[1] == kv_dim && o->s[2] == hd && o->s[3] == CRABML_HIP_ROPE_LLAMA && o->s[4] == pos &&
         o->s[5] == rope_dim);
    P.push(*o, LR_ROPE);
    // key_cache[l].concatenate(k.reshape.transpose([1, 0, 2]), 1)                         :542-554
    crabml_hip_buf* caches[2] = {nullptr, nullptr};
    for (int which = 0; which < 2; which++) {
      o = P.take(LZ_CONCAT);
      NEED(o && o->b == (which ? vv : kv) && s3(o->s, n_kv, 1, hd) && s3(o->s + 6, hd, kv_dim, 1) && o->s[4] == hd && o->s[5] == 1 &&
           o->s[10] == 1 && o->s[11] == pos && o->s[9] == pos * hd && o->s[3] % hd == 0);
      if (first && which == 0) {
        seq = o->s[3] / hd;
        kv_dtype = o->a->dtype;
        NEED(kv_dtype == CRABML_HIP_F16 || kv_dtype == CRABML_HIP_F32);
      }
      NEED(o->s[3] == seq * hd && pos < seq && o->a->dtype == kv_dtype && o->a->n_elems == n_kv * seq * hd);
      caches[which] = o->a;
      P.push(*o, LR_CONCAT);
      if (first && which == 1) P.T.back().go = true;  // token id, position and layer 0's caches are known from here on
    }
    NEED(caches[0] != caches[1]);
    M.kc.push_back(caches[0]);
    M.vc.push_back(caches[1]);
    o = P.take(LZ_CONTIGUOUS);  // q.reshape.transpose([1, 0, 2]).contiguous()            :561-565
    NEED(o && o->a == qv && s3(o->s, n_heads, 1, hd) && s3(o->s + 3, hd, dim, 1));
    const crabml_hip_buf* qc = o->out;
    P.push(*o);
    o = P.take(LZ_SCALE);  // .scale_inplace(1 / sqrt(head_dim))
    NEED(o && o->a == qc && o->s[0] == dim && o->f == 1.0f / std::sqrt((float)hd));
    P.push(*o);
    o = P.take(LZ_BATCH_MATMUL);  // attn = q.batch_matmul(k_cache.transpose([0, 2, 1]))   :571-577
    NEED(o && o->a == qc && o->b == caches[0] && o->s[0] == n_heads && o->s[1] == 1 && o->s[2] == hd && o->s[3] == n_kv && o->s[4] == pos + 1 &&
         s3(o->s + 5, seq * hd, 1, hd));
    const crabml_hip_buf* att = o->out;
    P.push(*o, LR_BMM_N);
    o = P.take(LZ_SOFTMAX);  // attn.softmax_inplace(2)                                    :578
    NEED(o && o->a == att && o->s[0] == n_heads && o->s[1] == pos + 1);
    P.push(*o, LR_SOFTMAX);
    o = P.take(LZ_BATCH_MATMUL);  // x_with_attn = attn.batch_matmul(v_cache)              :584-590
    NEED(o && o->a == att && o->b == caches[1] && o->s[0] == n_heads && o->s[1] == 1 && o->s[2] == pos + 1 && o->s[3] == n_kv && o->s[4] == hd &&
         s3(o->s + 5, seq * hd, hd, 1));
    const crabml_hip_buf* xa = o->out;
    P.push(*o, LR_BMM_K);
    o = P.take(LZ_MATMUL_VEC);  // wo.matmul_vec(x_with_attn)                              :600
    NEED(o && o->b == xa && o->s[0] == dim && o->s[1] == dim && o->s[2] == 1);
    const crabml_hip_buf* x2 = o->out;
    M.wo.push_back(o->a);
    P.push(*o);
    o = P.take(LZ_ADD);  // x.add_inplace(x_attn_orig)                                     :266
    NEED(o && o->a == x2 && o->b == xo && o->s[0] == dim && o->s[1] == dim);
    P.push(*o, LR_NONE, false, 2 * l);
    o = P.take(LZ_DUP);  // forward_ffn: x_orig_ffn = x.dup()                              :610
    NEED(o && o->a == x2);
    const crabml_hip_buf* xo2 = o->out;
    P.push(*o);
    o = P.take(LZ_RMS_NORM);  // eps = the literal 1e-5                                    :611
    NEED(o && o->a == x2 && o->s[0] == 1 && o->s[1] == dim && o->f == 1e-5f);
    P.push(*o);
    o = P.take(LZ_MUL);
    NEED(o && o->a == x2 && o->s[0] == dim && o->s[1] == dim && is_f32_vec(o->b, dim));
    M.rms_ffn.push_back(o->b);
    P.push(*o);
    o = P.take(LZ_MATMUL_VEC);  // h1 = ffn_gate.matmul_vec(x)                             :620
    NEED(o && o->b == x2 && o->s[1] == dim && o->s[2] == 1);
    if (first) hidden = o->s[0];
    NEED(o->s[0] == hidden && hidden);
    const crabml_hip_buf* h1 = o->out;
    M.gate.push_back(o->a);
    P.push(*o);
    o = P.take(LZ_MATMUL_VEC);  // h2 = ffn_up.matmul_vec(x)                               :621
    NEED(o && o->b == x2 && o->s[0] == hidden && o->s[1] == dim && o->s[2] == 1);
    const crabml_hip_buf* h2 = o->out;
    M.up.push_back(o->a);
    P.push(*o);
    o = P.take(LZ_SILU);  // h1.silu_inplace()                                             :626
    NEED(o && o->a == h1 && o->s[0] == hidden);
    P.push(*o);
    o = P.take(LZ_MUL);  // h1.mul_inplace(h2)                                             :628
    NEED(o && o->a == h1 && o->b == h2 && o->s[0] == hidden && o->s[1] == hidden);
    P.push(*o);
    o = P.take(LZ_MATMUL_VEC);  // ffn_down.matmul_vec(h1)                                 :633
    NEED(o && o->b == h1 && o->s[0] == dim && o->s[1] == hidden && o->s[2] == 1);
    const crabml_hip_buf* x3 = o->out;
    M.down.push_back(o->a);
    P.push(*o);
    o = P.take(LZ_ADD);  // x.add_inplace(x_orig_ffn)                                      :636
    NEED(o && o->a == x3 && o->b == xo2 && o->s[0] == dim && o->s[1] == dim);
    P.push(*o, LR_NONE, false, 2 * l + 1);
    x = x3;
    l++;
  }
  NEED(l > 0);
  o = P.take(LZ_RMS_NORM);  // final norm                                                  :274-276
  NEED(o && o->a == x && o->s[0] == 1 && o->s[1] == dim && o->f == eps);
  P.push(*o);
  o = P.take(LZ_MUL);
  NEED(o && o->a == x && o->s[0] == dim && o->s[1] == dim && is_f32_vec(o->b, dim));
  M.rms_final = o->b;
  P.push(*o);
  o = P.take(LZ_COPY_ROW);  // x_final.copy_rows_from(x, [n_batch - 1])                    :192-197
  NEED(o && o->b == x && o->s[0] == dim && o->s[1] == 0 && is_f32_vec(o->a, dim) && o->a != x);
  const crabml_hip_buf* xf = o->a;
  P.push(*o, LR_NONE, true);
  o = P.take(LZ_MATMUL_VEC);  // logits = output_weight.matmul_vec(x_final)                :199-208
  NEED(o && o->b == xf && o->s[1] == dim && o->s[2] == 1 && o->s[0] == vocab_e);
  M.output = o->a;
  P.push(*o, LR_NONE, false, 2 * l);
  NEED(P.ok);
  *slot_xnorm = P.slot[x];
  *slot_xfinal = P.slot[xf];
  *slot_logits = P.slot[o->out];
  crabml_hip_llama_config_t& c = M.cfg;
  c = crabml_hip_llama_config_t{};
  c.embedding_dim = dim;
  c.hidden_dim = hidden;
  c.n_layers = (size_t)l;
  c.n_heads = n_heads;
  c.n_kv_heads = n_kv;
  c.vocab_size = vocab_e;
  c.seq_len = seq;
  c.rope_dim = rope_dim;
  c.rms_norm_eps = eps;
  c.use_f16_kv_cache = kv_dtype == CRABML_HIP_F16 ? 1 : 0;
  c.flags = 0;  // (the step's graph is captured: the whole token is launched as soon as its position is verified)
  c.tp_size = 1;
  return true;
}
#undef NEED

void drop_model(crabml_hip_device* dev, LazyState& L) {
  if (L.ctx) {
    (void)lazy_resolve(dev);
    lazy_ctx_destroy(L.ctx);
  }
  L.ctx = nullptr;
  L.tmpl.clear();
  L.mentions.clear();
  L.model = LazyModel{};
  L.tracking = false;
}
// the active context steps aside (the host switched to another runner / model): kept, least recently used first; the oldest goes
// when more than LAZY_PARKED_MAX wait
void park_model(crabml_hip_device* dev, LazyState& L) {
  if (!L.ctx) return;
  (void)lazy_resolve(dev);  // handles the host still holds of this context's last token get their values now
  ParkedModel p;
  p.ctx = L.ctx;
  p.model = std::move(L.model);
  p.tmpl = std::move(L.tmpl);
  p.mentions = std::move(L.mentions);
  p.slot_xnorm = L.slot_xnorm;
  p.slot_xfinal = L.slot_xfinal;
  p.slot_logits = L.slot_logits;
  L.parked.push_back(std::move(p));
  L.ctx = nullptr;
  L.tmpl.clear();
  L.mentions.clear();
  L.model = LazyModel{};
  L.tracking = false;
  L.check_fault = false;
  L.fault_requested = false;
  while (L.parked.size() > LAZY_PARKED_MAX) {
    lazy_ctx_destroy(L.parked.front().ctx);
    L.parked.erase(L.parked.begin());
  }
}
void unpark_model(LazyState& L, size_t idx) {
  ParkedModel p = std::move(L.parked[idx]);
  L.parked.erase(L.parked.begin() + (long)idx);
  L.ctx = p.ctx;
  L.model = std::move(p.model);
  L.tmpl = std::move(p.tmpl);
  L.mentions = std::move(p.mentions);
  L.slot_xnorm = p.slot_xnorm;
  L.slot_xfinal = p.slot_xfinal;
  L.slot_logits = p.slot_logits;
  L.dead = false;
  L.stats.reactivated++;
}
// Contexts whose model or caches the host has released (the context's own holds are all that keeps them alive) are destroyed: they
// pin device memory -- the weights of a dropped model -- and nothing will be served from them again.  Called with the queue empty
// (recorded ops hold references of their own).
void reap_orphans(crabml_hip_device* dev, LazyState& L) {
  if (L.ctx && !L.tracking && L.q.empty() && lazy_ctx_orphaned(L.ctx)) {
    drop_model(dev, L);
    L.stats.reaped++;
  }
  for (size_t i = 0; i < L.parked.size();) {
    if (lazy_ctx_orphaned(L.parked[i].ctx)) {
      lazy_ctx_destroy(L.parked[i].ctx);
      L.parked.erase(L.parked.begin() + (long)i);
      L.stats.reaped++;
    } else {
      i++;
    }
  }
}

// at a flush, before the queue runs: does it hold a complete token of a model we do not serve yet?  true: a decode context
// was built from the token that starts at q[*start]
bool try_learn(crabml_hip_device* dev, LazyState& L, size_t* start) {
  if (L.q.size() < 40) return false;
  for (size_t i0 = 0; i0 < L.q.size(); i0++) {
    if (L.q[i0].kind != LZ_COPY_ROW || L.q.size() - i0 < 40) continue;
    Learner P{L.q, i0};
    LazyModel M;
    int sx = -1, sf = -1, sl = -1;
    if (!learn_token(P, M, &sx, &sf, &sl)) continue;
    if (L.ctx && L.model.same_buffers(M)) return false;  // the model we already serve (its token ran op by op for another reason)
    if (M.wq[0]->uid == L.unfusable_uid) return false;
    for (size_t pi = 0; pi < L.parked.size(); pi++)
      if (L.parked[pi].model.same_buffers(M)) {  // a runner we served before takes its turn again: its context is still there
        ParkedModel again = std::move(L.parked[pi]);
        L.parked.erase(L.parked.begin() + (long)pi);
        park_model(dev, L);  // the one being served steps aside (and the oldest parked one goes if too many wait)
        L.parked.push_back(std::move(again));
        unpark_model(L, L.parked.size() - 1);
        *start = i0;
        return true;
      }
    park_model(dev, L);
    crabml_hip_llama* ctx = nullptr;
    if (lazy_ctx_create(dev, M, &ctx) != 0 || !ctx) {
      L.unfusable_uid = M.wq[0]->uid;  // the decode context refused this model (shape / dtype mix): it stays on the per-op launches
      return false;
    }
    L.ctx = ctx;
    L.model = std::move(M);
    L.tmpl = std::move(P.T);
    L.mentions = std::move(P.mentions);
    L.slot_xnorm = sx;
    L.slot_xfinal = sf;
    L.slot_logits = sl;
    L.dead = false;
    L.stats.learned++;
    *start = i0;
    return true;
  }
  return false;
}

// ---- streaming: compare the op just recorded with the template, enqueue the segment it completes -------------------------
bool verify(LazyState& L, const LazyOp& o) {
  const TmplOp& t = L.tmpl[L.next];
  if (o.kind != t.kind) return false;
  // operands
  if (t.a_new) {  // a buffer the host allocated for this token: whole-buffer destination of copy_rows_from
    if (!o.a || o.a->dtype != CRABML_HIP_F32 || o.a->n_elems != t.s[0]) return false;
    for (crabml_hip_buf* b : L.slots)
      if (b == o.a) return false;
    L.slots[t.sa] = o.a;
  } else if (t.sa >= 0 ? L.slots[t.sa] != o.a : t.pa != o.a) {
    return false;
  }
  if (t.sb == -2 ? o.b != nullptr : t.sb >= 0 ? L.slots[t.sb] != o.b : t.pb != o.b) return false;
  if ((t.so == -2) != (o.out == nullptr)) return false;
  // scalars
  unsigned skip = 0;  // bit i: s[i] follows the token
  switch (t.rule) {
    case LR_TOKEN:
      if (o.s[1] >= L.model.cfg.vocab_size) return false;
      L.token = o.s[1];
      skip = 1u << 1;
      break;
    case LR_ROPE:
      if (!L.pos_known) {
        if (o.s[4] >= L.model.cfg.seq_len) return false;
        L.pos = o.s[4];
        L.pos_known = true;
      }
      if (o.s[4] != L.pos) return false;
      skip = 1u << 4;
      break;
    case LR_CONCAT:
      if (!L.pos_known || o.s[11] != L.pos || o.s[9] != L.pos * o.s[4]) return false;
      skip = (1u << 9) | (1u << 11);
      break;
    case LR_BMM_N:
      if (!L.pos_known || o.s[4] != L.pos + 1) return false;
      skip = 1u << 4;
      break;
    case LR_SOFTMAX:
      if (!L.pos_known || o.s[1] != L.pos + 1) return false;
      skip = 1u << 1;
      break;
    case LR_BMM_K:
      if (!L.pos_known || o.s[2] != L.pos + 1) return false;
      skip = 1u << 2;
      break;
    default: break;
  }
  for (int i = 0; i < 12; i++)
    if (!((skip >> i) & 1u) && o.s[i] != t.s[i]) return false;
  if (memcmp(&o.f, &t.f, sizeof(float)) != 0) return false;
  if (t.so >= 0) L.slots[t.so] = o.out;
  return true;
}

void abort_token(LazyState& L) {
  if (L.tracking && L.next > 0) L.stats.aborts++;
  L.tracking = false;
}

int commit_token(crabml_hip_device* dev, LazyState& L) {
  // a handle the host still holds must end up with its value: only the three the runner keeps have one (llama2.rs:184-211)
  const int n = (int)L.slots.size();
  for (int sidx = 0; sidx < n; sidx++) {
    const bool alive = L.slots[sidx]->refcnt.load() > L.mentions[sidx];
    if (alive && sidx != L.slot_xnorm && sidx != L.slot_xfinal && sidx != L.slot_logits) {
      abort_token(L);  // an intermediate escaped: the queue runs op by op (and overwrites what the shadow wrote)
      L.dead = true;   // ... and it would escape again next token: stop shadowing this model
      return 0;
    }
  }
  int di = 0;
  for (int sidx : {L.slot_xnorm, L.slot_xfinal}) {
    crabml_hip_buf* b = L.slots[sidx];
    if (b->refcnt.load() > L.mentions[sidx]) {  // x / x_final of forward(): alive until it returns, never read -- bound on demand
      b->deferred = 1;
      b->zero_init = false;
      crabml_hip_buf_retain(b);
      L.deferred[di++] = b;
    }
  }
  crabml_hip_buf* lb = L.slots[L.slot_logits];
  touch(lb);
  if (L.pin_buf) crabml_hip_buf_release(L.pin_buf);
  L.pin_buf = nullptr;
  L.pin_kind = 0;
  // the logits live in the context's buffer and -- sent there by the step's last kernels -- in pinned host memory: export() of this
  // handle is served from the host copy; device memory is bound (and filled) only if the handle is used as an operand
  lb->deferred = 2;
  crabml_hip_buf_retain(lb);
  L.deferred[2] = lb;
  if (lazy_ctx_has_host_logits(L.ctx)) {
    crabml_hip_buf_retain(lb);
    L.pin_buf = lb;
    L.pin_version = lb->version;
    L.pin_n = lb->n_elems;
    L.pin_kind = 1;
  }
  L.stats.fused_tokens++;
  L.stats.fused_ops += L.q.size();
  for (LazyOp& o : L.q) release_op(o);
  L.q.clear();
  L.tracking = false;
  L.check_fault = true;
  if (!L.parked.empty()) reap_orphans(dev, L);  // (a host that only ever commits tokens never reaches lazy_flush's tail)
  return 0;
}

int track(crabml_hip_device* dev, LazyState& L) {
  if (!L.tracking) {
    const LazyOp& o = L.q.back();
    const TmplOp& t0 = L.tmpl[0];
    if (o.kind != LZ_COPY_ROW || o.b != t0.pb || !o.a || o.a->n_elems != t0.s[0] || o.a->dtype != CRABML_HIP_F32) return 0;
    if (L.q.size() > 1) {  // whatever was recorded before the token runs first: the shadow's launches are ordered behind it
      LazyOp keep = L.q.back();
      L.q.pop_back();
      int rc = run_queue(dev, L);
      L.q.push_back(keep);
      if (rc != 0) return rc;
    }
    L.tracking = true;
    L.next = 0;
    L.pos_known = false;
    L.begun = false;
    L.whole_step = false;
    L.slots.assign(L.mentions.size(), nullptr);
  }
  const LazyOp& o = L.q.back();
  if (!verify(L, o)) {
    abort_token(L);
    return 0;
  }
  const TmplOp& t = L.tmpl[L.next++];
  const bool last = L.next == L.tmpl.size();
  int rc = 0;
  if (t.go && !dev->prof_on && lazy_ctx_has_graph(L.ctx)) {
    // token id, position and the first layer's caches are verified: the WHOLE step goes out now, as one graph launch (the rest
    // of the token's ~800 calls are only compared with the template; should one deviate, the shadow is dropped as always -- what
    // it wrote beyond its private buffers are cache rows at the position of the very concatenate ops the replay re-runs, or,
    // for layers whose ops never came, rows one past the caches' live length)
    rc = lazy_resolve(dev);  // the previous token's promised handles, if the host still holds them: the context is about to move on
    if (rc == 0) rc = lazy_ctx_begin(L.ctx, L.token, L.pos);
    if (rc == 0) rc = lazy_ctx_step(L.ctx, L.pos);
    L.begun = true;
    L.whole_step = true;
    if (rc == 0) L.stats.segments += (uint64_t)lazy_ctx_n_segments(L.ctx);
  } else if (t.seg_end >= 0 && !L.whole_step) {
    if (!L.begun) {
      rc = lazy_resolve(dev);
      if (rc == 0) rc = lazy_ctx_begin(L.ctx, L.token, L.pos);
      L.begun = true;
    }
    if (rc == 0) rc = lazy_ctx_segment(L.ctx, t.seg_end);
    if (rc == 0) L.stats.segments++;
  }
  if (rc != 0) {  // the decode context failed: this model goes back to the per-op launches for good
    abort_token(L);
    L.dead = true;
    return 0;
  }
  if (last) return commit_token(dev, L);
  return 0;
}

}  // namespace

int lazy_record(crabml_hip_device* dev, const LazyOp& op) {
  LazyState& L = *dev->lz;
  if (op.a) crabml_hip_buf_retain(op.a);
  if (op.b) crabml_hip_buf_retain(op.b);
  if (op.out) crabml_hip_buf_retain(op.out);
  L.q.push_back(op);
  L.stats.recorded++;
  if (dev->fuse && L.ctx && !L.dead) CH_TRY(track(dev, L));
  if (L.q.size() >= (1u << 16)) return lazy_flush(dev);  // a host that never looks at anything: bound the queue
  return 0;
}

int lazy_flush(crabml_hip_device* dev) {
  if (!dev->lz) return 0;
  LazyState& L = *dev->lz;
  if (L.q.empty()) return 0;
  abort_token(L);
  size_t i0 = 0;
  // (not at crabml_hip_device_destroy's flush: a context built there would be torn down again a moment later)
  if (dev->fuse && !dev->destroying && try_learn(dev, L, &i0)) {  // (before the ops run and release their handles)
    // the token the context was learned from is served by it right away: whatever precedes it runs op by op, then its ops
    // are fed to the matcher as if they were being recorded now (they stay queued until the token commits, as always)
    std::vector<LazyOp> tail(L.q.begin() + i0, L.q.end());
    L.q.resize(i0);
    int rc = run_queue(dev, L);
    for (size_t i = 0; i < tail.size(); i++) {
      L.q.push_back(tail[i]);
      if (rc == 0 && !L.dead) rc = track(dev, L);
    }
    abort_token(L);
    const int rc2 = run_queue(dev, L);
    reap_orphans(dev, L);  // (the context that just stepped aside may have been its caches' last owner)
    return rc != 0 ? rc : rc2;
  }
  const int rc = run_queue(dev, L);
  reap_orphans(dev, L);
  return rc;
}

int lazy_release_contexts(crabml_hip_device* dev) {
  if (!dev->lz) return 0;
  LazyState& L = *dev->lz;
  int n = 0;
  for (ParkedModel& p : L.parked) {
    lazy_ctx_destroy(p.ctx);
    n++;
  }
  L.parked.clear();
  if (L.ctx && !L.tracking && L.q.empty()) {  // between tokens: the context is rebuilt from the next token if the host goes on
    drop_model(dev, L);
    n++;
  }
  return n;
}

int lazy_resolve(crabml_hip_device* dev) {
  if (!dev->lz) return 0;
  LazyState& L = *dev->lz;
  int rc = 0;
  // (the host copy of the logits belongs to the step that is being left behind: from here on its handle, if anybody still holds
  // it, is an ordinary bound buffer)
  if (L.pin_buf) crabml_hip_buf_release(L.pin_buf);
  L.pin_buf = nullptr;
  L.pin_kind = 0;
  for (int i = 0; i < 3; i++) {
    crabml_hip_buf* b = L.deferred[i];
    if (!b) continue;
    L.deferred[i] = nullptr;
    const int kind = b->deferred;
    b->deferred = 0;
    if (b->refcnt.load() > 1 && rc == 0) {  // still held by the host: bind it and produce the value
      rc = ensure_mem(dev, b);
      if (rc == 0) rc = kind == 2 ? lazy_ctx_copy_logits(L.ctx, (float*)b->ptr) : lazy_ctx_final_norm(L.ctx, (float*)b->ptr);
      L.stats.deferred_bound++;
    }
    crabml_hip_buf_release(b);
  }
  return rc;
}

int lazy_pinned_kind(crabml_hip_device* dev, const crabml_hip_buf* b, size_t n) {
  if (!dev->lz) return 0;
  LazyState& L = *dev->lz;
  if (L.pin_buf != b || L.pin_version != b->version || n > L.pin_n) return 0;
  return L.pin_kind;
}

int lazy_export_wait(crabml_hip_device* dev, float* dst, size_t n) {
  LazyState& L = *dev->lz;
  if (dev->dry) {  // record-only test device: the path is taken, nothing is there to copy
    memset(dst, 0, n * 4);
    L.stats.pinned_exports++;
    return 0;
  }
  int fault = 0;
  const auto t0 = std::chrono::steady_clock::now();
  const float* src = lazy_ctx_wait_logits(L.ctx, &fault);
  L.stats.wait_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  if (!src) CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "the logits of the fused decode step never reached the host");
  memcpy(dst, src, n * 4);
  L.stats.pinned_exports++;
  L.check_fault = false;  // this step's fault word travelled with the flag
  if (fault) CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "a norm-epilogue gather of the fused decode step timed out (workgroups not co-resident?)");
  return 0;
}

int lazy_fault_request(crabml_hip_device* dev) {
  if (!dev->lz || !dev->lz->check_fault || !dev->lz->ctx) return 0;
  dev->lz->check_fault = false;
  dev->lz->fault_requested = true;
  return lazy_ctx_fault_request(dev->lz->ctx);
}

int lazy_fault_check(crabml_hip_device* dev) {
  if (!dev->lz || !dev->lz->fault_requested || !dev->lz->ctx) return 0;
  dev->lz->fault_requested = false;
  if (lazy_ctx_fault_value(dev->lz->ctx))
    CH_BAIL(dev, CRABML_HIP_UNEXPECTED, "a norm-epilogue gather of the fused decode step timed out (workgroups not co-resident?)");
  return 0;
}

void lazy_destroy(crabml_hip_device* dev) {
  if (!dev->lz) return;
  LazyState& L = *dev->lz;
  for (LazyOp& o : L.q) release_op(o);  // (device_destroy flushed first; whatever is left is dropped)
  L.q.clear();
  drop_model(dev, L);
  for (ParkedModel& p : L.parked) lazy_ctx_destroy(p.ctx);
  L.parked.clear();
  if (L.pin_buf) crabml_hip_buf_release(L.pin_buf);
  delete dev->lz;
  dev->lz = nullptr;
}

}  // namespace crabml_hip

extern "C" int crabml_hip_debug_lazy_stats(crabml_hip_device_t* dev, uint64_t* out, size_t cap) {
  if (!dev || !out) return CRABML_HIP_BAD_INPUT;
  uint64_t v[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (dev->lz) {
    const crabml_hip::LazyStats& s = dev->lz->stats;
    v[0] = s.recorded;
    v[1] = s.replayed;
    v[2] = s.fused_tokens;
    v[3] = s.fused_ops;
    v[4] = s.segments;
    v[5] = s.aborts;
    v[6] = s.learned;
    v[7] = s.deferred_bound;
    v[8] = s.wait_ns;
    v[9] = s.pinned_exports;
    v[10] = s.reactivated;
    v[11] = s.reaped;
  }
  for (size_t i = 0; i < cap && i < 12; i++) out[i] = v[i];
  return 0;
}
