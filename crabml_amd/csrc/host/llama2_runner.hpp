// llama2_runner.hpp -- the CALLER of the hot path, generic over the tensor backend exactly like the
// reference: `Llama2Runner<T: Tensor>` (crabml-llama2/src/llama2.rs:26-43).  It issues the same op
// sequence per decode step (forward :184-211, forward_llama :213-281, forward_multi_query_attention
// :527-603, forward_ffn :605-638), so running it over HipTensor is what "crabml-llama2 runs unchanged"
// means on this side of the boundary.  Llama architecture only (gemma/qwen2/phi2 are out of scope).
#pragma once
#include <chrono>
#include <cmath>
#include <optional>
#include <string>
#include <vector>

#include "strider.hpp"

namespace crabml_host {

// crabml-llama2/src/model.rs:30-53
struct LlamaConfig {
  size_t embedding_dim = 0, hidden_dim = 0, n_layers = 0, n_heads = 0, n_kv_heads = 0, vocab_size = 0, seq_len = 0;
  float rms_norm_eps = 1e-5f;
  std::optional<size_t> rope_dim;
  size_t kv_dim() const { return embedding_dim * n_kv_heads / n_heads; }
  size_t head_size() const { return embedding_dim / n_heads; }
};

// crabml-llama2/src/model.rs:55-84 (Llama subset)
template <class T>
struct LlamaWeights {
  T token_embed;
  std::vector<T> rms_att_weight, rms_ffn_weight;
  std::vector<T> wq, wk, wv, wo;
  std::vector<T> ffn_gate_weight, ffn_down_weight, ffn_up_weight;
  T rms_final_weight;
  std::optional<T> output_weight;
};

// Iterator::max_by keeps the LAST maximum (crabml-llama2/src/sampler.rs:109-116)
#if defined(__x86_64__)
// the maximum of NaN-free data; *clean = false when a NaN was seen (the caller then takes the defining loop)
__attribute__((target("avx2"))) inline float max_avx2(const float* p, size_t n, bool* clean) {
  typedef float v8 __attribute__((vector_size(32), aligned(4)));
  v8 m0 = *(const v8*)p, m1 = *(const v8*)(p + 8), m2 = *(const v8*)(p + 16), m3 = *(const v8*)(p + 24);
  typedef int i8 __attribute__((vector_size(32)));
  i8 ok = (m0 == m0) & (m1 == m1) & (m2 == m2) & (m3 == m3);
  size_t i = 32;
  for (; i + 32 <= n; i += 32) {
    const v8 a = *(const v8*)(p + i), b = *(const v8*)(p + i + 8), c = *(const v8*)(p + i + 16), d = *(const v8*)(p + i + 24);
    ok &= (a == a) & (b == b) & (c == c) & (d == d);
    m0 = a > m0 ? a : m0;
    m1 = b > m1 ? b : m1;
    m2 = c > m2 ? c : m2;
    m3 = d > m3 ? d : m3;
  }
  m0 = m1 > m0 ? m1 : m0;
  m2 = m3 > m2 ? m3 : m2;
  m0 = m2 > m0 ? m2 : m0;
  float mx = m0[0];
  bool c = true;
  for (int j = 0; j < 8; j++) {
    c &= ok[j] != 0;
    mx = m0[j] > mx ? m0[j] : mx;
  }
  for (; i < n; i++) {
    c &= p[i] == p[i];
    mx = p[i] > mx ? p[i] : mx;
  }
  *clean = c;
  return mx;
}
#endif

inline size_t sample_argmax(const std::vector<float>& p) {
  const size_t n = p.size();
#if defined(__x86_64__)
  if (n >= 64 && __builtin_cpu_supports("avx2")) {
    bool clean = false;
    const float mx = max_avx2(p.data(), n, &clean);
    if (clean)
      for (size_t k = n; k-- > 0;)
        if (p[k] == mx) return k;
  } else
#endif
  if (n >= 64) {
    // two passes over NaN-free logits: the maximum (eight independent compare chains), then its last position -- the element
    // max_by returns.  Anything with a NaN in it takes the loop below, whose tie / NaN behaviour is the definition.
    float m[8];
    bool clean = true;
    for (int j = 0; j < 8; j++) m[j] = p[j];
    size_t i = 8;
    for (; i + 8 <= n; i += 8)
      for (int j = 0; j < 8; j++) {
        const float v = p[i + j];
        clean &= v == v;
        m[j] = v > m[j] ? v : m[j];
      }
    for (int j = 0; j < 8; j++) clean &= m[j] == m[j];
    for (; i < n; i++) {
      clean &= p[i] == p[i];
      m[0] = p[i] > m[0] ? p[i] : m[0];
    }
    if (clean) {
      float mx = m[0];
      for (int j = 1; j < 8; j++) mx = m[j] > mx ? m[j] : mx;
      for (size_t k = n; k-- > 0;)
        if (p[k] == mx) return k;
    }
  }
  size_t best = 0;
  for (size_t i = 1; i < n; i++)
    if (!(p[best] > p[i])) best = i;
  return best;
}

template <class T>
class Llama2Runner {
 public:
  using DeviceRef = typename T::DeviceRef;
  using DType = decltype(std::declval<T>().dtype());

  Llama2Runner(const LlamaConfig& conf, std::shared_ptr<LlamaWeights<T>> weights, DeviceRef device, size_t seq_len,
               bool use_f16_kv_cache, DType f32, DType f16)
      : conf_(conf), weights_(std::move(weights)), device_(std::move(device)), f32_(f32) {
    DType kvt = use_f16_kv_cache ? f16 : f32;
    logits_.assign(conf_.vocab_size, 0.0f);
    for (size_t l = 0; l < conf_.n_layers; l++) {  // llama2.rs:65-86
      key_cache_.push_back(T::alloc({conf_.n_kv_heads, seq_len, conf_.head_size()}, kvt, device_).resize(1, 0));
      value_cache_.push_back(T::alloc({conf_.n_kv_heads, seq_len, conf_.head_size()}, kvt, device_).resize(1, 0));
    }
  }

  const LlamaConfig& conf() const { return conf_; }
  size_t kv_cache_len() const { return key_cache_[0].shape()[1]; }
  const std::vector<float>& logits() const { return logits_; }

  // llama2.rs:184-211
  void forward(const std::vector<size_t>& tokens, size_t pos) {
    T x = forward_llama(tokens, pos);
    T x_final = T::alloc({conf_.embedding_dim}, f32_, device_);
    x_final.copy_rows_from(x, {tokens.size() - 1});
    const T& ow = weights_->output_weight ? *weights_->output_weight : weights_->token_embed;
    T logits = ow.matmul_vec(x_final);
    logits.export_into(logits_);  // logits.export(&mut self.logits), llama2.rs:208
  }

  // prefill token by token (llama2.rs:127-129), then greedy generation; returns the sampled ids
  std::vector<size_t> generate_greedy(const std::vector<size_t>& prompt, size_t steps) {
    size_t base = kv_cache_len();
    for (size_t i = 0; i < prompt.size(); i++) forward({prompt[i]}, base + i);
    std::vector<size_t> out;
    size_t tok = sample_argmax(logits_);
    out.push_back(tok);
    size_t pos = kv_cache_len();
    for (size_t s = 1; s < steps; s++) {
      forward({tok}, pos);
      tok = sample_argmax(logits_);
      out.push_back(tok);
      pos++;
    }
    return out;
  }

 private:
  // llama2.rs:213-281
  T forward_llama(const std::vector<size_t>& tokens, size_t pos) {
    const size_t embed_dim = conf_.embedding_dim, n_heads = conf_.n_heads, n_kv_heads = conf_.n_kv_heads;
    const size_t head_dim = conf_.head_size();
    const size_t rope_dim = conf_.rope_dim.value_or(head_dim);
    const size_t n_batch = tokens.size();
    const LlamaWeights<T>& w = *weights_;
    T x = T::alloc({n_batch, embed_dim}, f32_, device_);
    x.copy_rows_from(w.token_embed, tokens);
    for (size_t l = 0; l < conf_.n_layers; l++) {
      T x_attn_orig = x.dup();
      x = x.rms_norm_inplace(conf_.rms_norm_eps);
      x = x.mul_inplace(w.rms_att_weight[l]);
      x = x.with_name("attn_rmsnorm:" + std::to_string(l) + ":" + std::to_string(pos));
      x = x.with_name("x_debug:" + std::to_string(l) + ":" + std::to_string(pos));
      T q = w.wq[l].matmul_vec(x);
      T k = w.wk[l].matmul_vec(x);
      T v = w.wv[l].matmul_vec(x);
      q = q.reshape({n_batch, n_heads, head_dim});
      k = k.reshape({n_batch, n_kv_heads, head_dim});
      q = q.rope_inplace(rope_llama(), pos, rope_dim);
      k = k.rope_inplace(rope_llama(), pos, rope_dim);
      x = forward_multi_query_attention(q, k, v, l, n_kv_heads, n_heads, embed_dim, head_dim, n_batch);
      x = x.with_name("attn_out:" + std::to_string(l) + ":" + std::to_string(pos));
      x = x.add_inplace(x_attn_orig);
      x = forward_ffn(x, l);
      x = x.with_name("ffn_out:" + std::to_string(l) + ":" + std::to_string(pos));
    }
    x = x.rms_norm_inplace(conf_.rms_norm_eps);
    x = x.mul_inplace(w.rms_final_weight);
    return x.with_name("final_rmsnorm:" + std::to_string(pos));
  }

  // llama2.rs:527-603
  T forward_multi_query_attention(T q, T k, T v, size_t l, size_t n_kv_heads, size_t n_heads, size_t embed_dim,
                                  size_t head_dim, size_t n_batch) {
    {
      T kt = k.reshape({n_batch, n_kv_heads, head_dim}).transpose({1, 0, 2});
      T vt = v.reshape({n_batch, n_kv_heads, head_dim}).transpose({1, 0, 2});
      key_cache_[l].concatenate(kt, 1);
      value_cache_[l].concatenate(vt, 1);
    }
    q = q.reshape({n_batch, n_heads, head_dim}).transpose({1, 0, 2}).contiguous().scale_inplace(
        1.0f / std::sqrt((float)head_dim));
    T& k_cache = key_cache_[l];
    TensorStrider k_orig = k_cache.strider();
    T k_cache_t = k_cache.transpose({0, 2, 1});
    T attn = q.batch_matmul(k_cache_t);
    attn = attn.softmax_inplace(2);
    key_cache_[l] = k_cache_t.with_strider(k_orig);
    T& v_cache = value_cache_[l];
    T x_with_attn = attn.batch_matmul(v_cache);
    if (n_batch == 1)
      x_with_attn = x_with_attn.reshape({n_batch, embed_dim});
    else
      x_with_attn = x_with_attn.transpose({1, 0, 2}).contiguous().reshape({n_batch, embed_dim});
    return weights_->wo[l].matmul_vec(x_with_attn);
  }

  // llama2.rs:605-638 -- the FFN norm's eps is the literal 1e-5
  T forward_ffn(T x, size_t l) {
    const LlamaWeights<T>& w = *weights_;
    T x_orig_ffn = x.dup();
    x = x.rms_norm_inplace(1e-5f);
    x = x.mul_inplace(w.rms_ffn_weight[l]);
    T h1 = w.ffn_gate_weight[l].matmul_vec(x);
    T h2 = w.ffn_up_weight[l].matmul_vec(x);
    h1 = h1.silu_inplace();
    h1 = h1.mul_inplace(h2);
    x = w.ffn_down_weight[l].matmul_vec(h1);
    return x.add_inplace(x_orig_ffn);
  }

  static auto rope_llama() { return T::rope_mode_llama(); }

  LlamaConfig conf_;
  std::shared_ptr<LlamaWeights<T>> weights_;
  DeviceRef device_;
  DType f32_;
  std::vector<float> logits_;
  std::vector<T> key_cache_, value_cache_;
};

}  // namespace crabml_host
