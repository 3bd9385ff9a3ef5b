// hip_llama.hpp -- host-side handle of the fused decode step (crabml_hip_llama_*).
// In the Rust crate this is `impl HipLlamaRunner` next to `impl Tensor for HipTensor`: crabml-llama2's
// `Llama2Runner::forward_llama` (llama2.rs:213-281) stays generic; a 10-line specialisation hands a whole
// decode step to the backend when T = HipTensor.
#pragma once
#include <memory>
#include <vector>

#include "hip_tensor.hpp"
#include "llama2_runner.hpp"

namespace crabml_host {

// RAII handle of a tensor-parallel RCCL communicator (crabml_hip_tp_comm_*): one per process / GPU
class TpComm {
 public:
  static std::vector<uint8_t> unique_id() {
    std::vector<uint8_t> id(128);
    if (crabml_hip_tp_get_unique_id(id.data()) != 0) throw Error(ErrorKind::Unexpected, "ncclGetUniqueId failed (is librccl.so loadable?)");
    return id;
  }
  TpComm(HipTensorDeviceRef device, const std::vector<uint8_t>& id, int nranks, int rank) : device_(std::move(device)), nranks_(nranks), rank_(rank) {
    if (id.size() != 128) throw Error(ErrorKind::BadInput, "a ncclUniqueId is 128 bytes");
    device_->check(crabml_hip_tp_comm_create(device_->raw(), id.data(), nranks, rank, &comm_));
  }
  // the one-shot P2P group (crabml_hip_tp_p2p_*): create -> export_handle -> (ship handles) -> connect
  struct P2P {};
  TpComm(P2P, HipTensorDeviceRef device, int nranks, int rank, size_t max_elems) : device_(std::move(device)), nranks_(nranks), rank_(rank) {
    device_->check(crabml_hip_tp_p2p_create(device_->raw(), nranks, rank, max_elems, &comm_));
  }
  std::vector<uint8_t> export_handle() const {
    std::vector<uint8_t> h(64);
    device_->check(crabml_hip_tp_p2p_export(comm_, h.data()));
    return h;
  }
  void connect(const std::vector<uint8_t>& handles) {
    if (handles.size() != (size_t)nranks_ * 64) throw Error(ErrorKind::BadInput, "connect: expected nranks x 64 bytes of handles");
    device_->check(crabml_hip_tp_p2p_connect(comm_, handles.data()));
  }
  static void connect_local(const std::vector<std::shared_ptr<TpComm>>& comms) {
    std::vector<crabml_hip_tp_comm_t*> raw;
    for (const auto& c : comms) raw.push_back(c->raw());
    if (raw.empty()) throw Error(ErrorKind::BadInput, "connect_local: no ranks");
    comms[0]->device_->check(crabml_hip_tp_p2p_connect_local(raw.data(), (int)raw.size()));
  }
  ~TpComm() {
    if (comm_) crabml_hip_tp_comm_destroy(comm_);
  }
  TpComm(const TpComm&) = delete;
  TpComm& operator=(const TpComm&) = delete;
  crabml_hip_tp_comm_t* raw() const { return comm_; }
  int nranks() const { return nranks_; }
  int rank() const { return rank_; }
  void all_reduce(HipTensor& t) { device_->check(crabml_hip_tp_all_reduce(comm_, t.raw(), t.strider().len())); }

 private:
  HipTensorDeviceRef device_;
  crabml_hip_tp_comm_t* comm_ = nullptr;
  int nranks_, rank_;
};

class HipLlamaRunner {
 public:
  HipLlamaRunner(const LlamaConfig& conf, std::shared_ptr<LlamaWeights<HipTensor>> w, HipTensorDeviceRef device,
                 size_t seq_len, bool use_f16_kv_cache, bool use_graph = true, bool prefetch = true, int tp_size = 1,
                 int tp_rank = 0, std::shared_ptr<TpComm> comm = nullptr, bool norm_epilogue = true, int extra_flags = 0,
                 size_t attn_long_from = 0, size_t prefill_chunk = 0)
      : conf_(conf), weights_(std::move(w)), device_(std::move(device)), comm_(std::move(comm)), tp_size_(tp_size > 1 ? tp_size : 1) {
    crabml_hip_llama_config_t c{};
    c.embedding_dim = conf.embedding_dim;
    c.hidden_dim = conf.hidden_dim;
    c.n_layers = conf.n_layers;
    c.n_heads = conf.n_heads;
    c.n_kv_heads = conf.n_kv_heads;
    c.vocab_size = conf.vocab_size;
    c.seq_len = seq_len;
    c.rope_dim = conf.rope_dim.value_or(conf.head_size());
    c.rms_norm_eps = conf.rms_norm_eps;
    c.use_f16_kv_cache = use_f16_kv_cache ? 1 : 0;
    c.flags = (use_graph ? 0 : CRABML_HIP_LLAMA_NO_GRAPH) | (prefetch ? 0 : CRABML_HIP_LLAMA_NO_PREFETCH) |
              (norm_epilogue ? 0 : CRABML_HIP_LLAMA_NO_NORM_EPILOGUE) | extra_flags;
    c.tp_size = tp_size;
    c.tp_rank = tp_rank;
    c.tp_comm = comm_ ? comm_->raw() : nullptr;
    c.attn_long_from = attn_long_from;
    c.prefill_chunk = prefill_chunk;
    auto raws = [](const std::vector<HipTensor>& v) {
      std::vector<const crabml_hip_buf_t*> r;
      for (const auto& t : v) r.push_back(t.raw());
      return r;
    };
    const auto& W = *weights_;
    auto att = raws(W.rms_att_weight), ffn = raws(W.rms_ffn_weight), wq = raws(W.wq), wk = raws(W.wk), wv = raws(W.wv),
         wo = raws(W.wo), gate = raws(W.ffn_gate_weight), down = raws(W.ffn_down_weight), up = raws(W.ffn_up_weight);
    for (auto* v : {&att, &ffn, &wq, &wk, &wv, &wo, &gate, &down, &up})
      if (v->size() != conf.n_layers) throw Error(ErrorKind::ModelError, "weights do not have n_layers entries");
    crabml_hip_llama_weights_t cw{};
    cw.token_embed = W.token_embed.raw();
    cw.rms_att_weight = att.data();
    cw.rms_ffn_weight = ffn.data();
    cw.wq = wq.data();
    cw.wk = wk.data();
    cw.wv = wv.data();
    cw.wo = wo.data();
    cw.ffn_gate_weight = gate.data();
    cw.ffn_down_weight = down.data();
    cw.ffn_up_weight = up.data();
    cw.rms_final_weight = W.rms_final_weight.raw();
    cw.output_weight = W.output_weight ? W.output_weight->raw() : nullptr;
    device_->check(crabml_hip_llama_create(device_->raw(), &c, &cw, &ctx_));
  }
  ~HipLlamaRunner() {
    if (ctx_) crabml_hip_llama_destroy(ctx_);
  }
  HipLlamaRunner(const HipLlamaRunner&) = delete;
  HipLlamaRunner& operator=(const HipLlamaRunner&) = delete;

  size_t kv_cache_len() const { return crabml_hip_llama_kv_len(ctx_); }
  crabml_hip_llama_t* raw() const { return ctx_; }
  const LlamaConfig& conf() const { return conf_; }
  const HipTensorDeviceRef& device() const { return device_; }
  void reset() { device_->check(crabml_hip_llama_reset(ctx_)); }
  // Llama2Runner::forward (llama2.rs:184-211) for one token; returns the logits
  std::vector<float> forward(size_t token, size_t pos) {
    std::vector<float> logits(conf_.vocab_size);
    device_->check(crabml_hip_llama_forward(ctx_, token, pos, logits.data()));
    return logits;
  }
  // the token loop of Llama2Runner::prefill (llama2.rs:111-129) as batched passes; returns the last token's logits
  std::vector<float> prefill(const std::vector<uint32_t>& tokens) {
    std::vector<float> logits(conf_.vocab_size);
    device_->check(crabml_hip_llama_prefill(ctx_, tokens.data(), tokens.size(), logits.data()));
    return logits;
  }
  void forward_async(size_t token, size_t pos) { device_->check(crabml_hip_llama_forward(ctx_, token, pos, nullptr)); }
  std::vector<uint32_t> decode_greedy(size_t token, size_t steps) {
    std::vector<uint32_t> ids(steps);
    device_->check(crabml_hip_llama_decode_greedy(ctx_, token, steps, ids.data()));
    return ids;
  }
  std::vector<uint8_t> debug_kv(size_t layer, bool v, bool f16) {
    size_t n = conf_.n_kv_heads / tp_size_ * seq_cap() * conf_.head_size() * (f16 ? 2 : 4);
    std::vector<uint8_t> out(n);
    device_->check(crabml_hip_llama_debug_kv(ctx_, layer, v ? 1 : 0, out.data(), n));
    return out;
  }
  // one decode step of a single-device simulated tp group (crabml_hip_llama_tp_sim_forward); logits from rank 0
  static std::vector<float> tp_sim_forward(const std::vector<HipLlamaRunner*>& ranks, size_t token, size_t pos) {
    if (ranks.empty()) throw Error(ErrorKind::BadInput, "tp_sim_forward: no ranks");
    std::vector<crabml_hip_llama_t*> ctxs;
    for (auto* r : ranks) ctxs.push_back(r->ctx_);
    std::vector<float> logits(ranks[0]->conf_.vocab_size);
    ranks[0]->device_->check(crabml_hip_llama_tp_sim_forward(ctxs.data(), (int)ctxs.size(), token, pos, logits.data()));
    return logits;
  }
  void set_seq_cap(size_t s) { seq_cap_ = s; }
  size_t seq_cap() const { return seq_cap_; }

 private:
  LlamaConfig conf_;
  std::shared_ptr<LlamaWeights<HipTensor>> weights_;
  HipTensorDeviceRef device_;  // declared before ctx_ is destroyed in ~HipLlamaRunner
  std::shared_ptr<TpComm> comm_;
  size_t tp_size_ = 1;
  crabml_hip_llama_t* ctx_ = nullptr;
  size_t seq_cap_ = 0;
};

}  // namespace crabml_host
