// gguf.hpp -- GGUF container reader + the llama loader on top of it, the data format in front of the hot path
// (SURVEY.md 8f-2).  Mirrors crabml-core/src/gguf.rs and crabml-llama2/src/model.rs:
//   * GGUFFileLoader::new mmaps the file (gguf.rs:795-827), GGUFFile::decode walks header -> metadata -> tensor
//     infos -> aligned tensor data (gguf.rs:522-566, 632-646, 710-735);
//   * string / array lengths and tensor dimensions are u32 in v1 and u64 in v2 / v3 (gguf.rs:399-427);
//   * the data of tensor i runs from its offset to the NEXT tensor's offset (the last one to the end of the file), so it
//     includes the alignment padding (gguf.rs:737-759);
//   * `general.alignment` may be any integer type, default 32 (gguf.rs:575-587); the data section starts at
//     position - position % alignment + alignment in the reference (gguf.rs:722-724: a full extra block when the infos
//     already end aligned -- which disagrees with the GGUF spec for 1 file in `alignment`; decode() resolves that case
//     from the file itself and falls back to the spec, see there);
//   * load_config reads `<arch>.*` keys and takes vocab_size from tokenizer.ggml.tokens (model.rs:565-625);
//   * load_weights reverses the on-disk dimensions (model.rs:473-475) and uploads every tensor in its stored type
//     (the CPU-side F32-only gate of GpuLlamaModel::from_cpu is what the hip backend lifts).
// Llama architecture only, like the rest of this host layer.  The tensor bytes go to the device straight from the
// mapping (Tensor::from_cpu -> crabml_hip_buf_from_cpu: chunked H2D + plane repack on the device).
#pragma once
#include <cstdio>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <variant>
#include <vector>

#include "hip_tensor.hpp"
#include "llama2_runner.hpp"

namespace crabml_host {

enum class GGUFValueType : uint32_t {  // gguf.rs:83-108
  U8 = 0, I8 = 1, U16 = 2, I16 = 3, U32 = 4, I32 = 5, F32 = 6, Bool = 7, String = 8, Array = 9, U64 = 10, I64 = 11, F64 = 12
};

struct GGUFValue;
struct GGUFArray {
  GGUFValueType elem_type = GGUFValueType::U8;
  std::vector<GGUFValue> items;
};
struct GGUFValue {
  GGUFValueType type = GGUFValueType::U8;
  std::variant<uint64_t, int64_t, double, std::string, GGUFArray> v;
};

struct GGUFTensorInfo {  // gguf.rs:648-689
  std::string name;
  std::vector<size_t> dimensions;  // as stored: innermost first
  uint32_t ggml_type = 0;
  uint64_t offset = 0;
  const uint8_t* data = nullptr;
  size_t data_len = 0;
};

class GGUFFile {
 public:
  // GGUFFileLoader::new(path, mlock) (gguf.rs:793-826): mmap + madvise(WILLNEED) + optional mlock
  // data_start: -1 = decide from the file (see decode()), 0 = the GGUF spec's data start, 1 = the reference's (gguf.rs:722-724)
  explicit GGUFFile(const std::string& path, bool mlock = false, int data_start = -1) : data_start_override_(data_start) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) throw Error(ErrorKind::IOError, "failed to open the file: " + path);
    struct stat st {};
    if (fstat(fd_, &st) != 0 || st.st_size <= 0) {
      ::close(fd_);
      throw Error(ErrorKind::IOError, "failed to stat the file: " + path);
    }
    len_ = (size_t)st.st_size;
    void* p = mmap(nullptr, len_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (p == MAP_FAILED) {
      ::close(fd_);
      throw Error(ErrorKind::IOError, "failed to mmap the file: " + path);
    }
    base_ = (const uint8_t*)p;
    if (madvise(p, len_, MADV_WILLNEED) != 0) {  // gguf.rs:811-817
      munmap(p, len_);
      ::close(fd_);
      throw Error(ErrorKind::IOError, "failed to advise the mmap: " + path);
    }
    if (mlock && ::mlock(p, len_) != 0) {  // gguf.rs:819-825
      munmap(p, len_);
      ::close(fd_);
      throw Error(ErrorKind::IOError, "failed to lock the mmap: " + path);
    }
    try {
      decode();
    } catch (...) {
      munmap((void*)base_, len_);
      ::close(fd_);
      throw;
    }
  }
  ~GGUFFile() {
    if (base_) munmap((void*)base_, len_);
    if (fd_ >= 0) ::close(fd_);
  }
  GGUFFile(const GGUFFile&) = delete;
  GGUFFile& operator=(const GGUFFile&) = delete;

  uint32_t version() const { return version_; }
  const std::string& architecture() const { return architecture_; }
  const std::map<std::string, GGUFValue>& metadata() const { return kv_; }
  const std::vector<GGUFTensorInfo>& tensor_infos() const { return tensors_; }
  size_t tensor_data_offset() const { return data_off_; }
  // 0: data starts where the GGUF spec puts it; 1: the reference's always-skip convention was detected (see decode())
  int data_start_convention() const { return data_start_convention_; }

  uint64_t alignment() const {  // gguf.rs:575-587
    auto it = kv_.find("general.alignment");
    if (it == kv_.end()) return 32;
    const GGUFValue& x = it->second;
    switch (x.type) {
      case GGUFValueType::U8: case GGUFValueType::U16: case GGUFValueType::U32: case GGUFValueType::U64:
        return std::get<uint64_t>(x.v);
      case GGUFValueType::I8: case GGUFValueType::I16: case GGUFValueType::I32: case GGUFValueType::I64: {
        int64_t s = std::get<int64_t>(x.v);
        return s > 0 ? (uint64_t)s : 32;
      }
      default: return 32;
    }
  }
  const GGUFTensorInfo* get_tensor_info(const std::string& name) const {  // gguf.rs:781-788
    for (const auto& t : tensors_)
      if (t.name == name) return &t;
    return nullptr;
  }
  // typed getters: None when the key is absent OR holds another type (gguf.rs:431-497)
  std::optional<uint32_t> get_u32(const std::string& key) const {
    auto it = kv_.find(key);
    if (it == kv_.end() || it->second.type != GGUFValueType::U32) return std::nullopt;
    return (uint32_t)std::get<uint64_t>(it->second.v);
  }
  std::optional<float> get_f32(const std::string& key) const {
    auto it = kv_.find(key);
    if (it == kv_.end() || it->second.type != GGUFValueType::F32) return std::nullopt;
    return (float)std::get<double>(it->second.v);
  }
  std::optional<std::string> get_string(const std::string& key) const {
    auto it = kv_.find(key);
    if (it == kv_.end() || it->second.type != GGUFValueType::String) return std::nullopt;
    return std::get<std::string>(it->second.v);
  }
  std::optional<size_t> get_string_array_len(const std::string& key) const {
    auto it = kv_.find(key);
    if (it == kv_.end() || it->second.type != GGUFValueType::Array) return std::nullopt;
    const GGUFArray& a = std::get<GGUFArray>(it->second.v);
    if (a.elem_type != GGUFValueType::String) return std::nullopt;
    return a.items.size();
  }

 private:
  // GGUFBufReader (gguf.rs:247-285)
  const uint8_t* take(size_t n) {
    if (n > len_ - pos_) throw Error(ErrorKind::FormatError, "failed to read " + std::to_string(n) + " bytes from the buffer, only " + std::to_string(len_ - pos_) + " bytes left");
    const uint8_t* p = base_ + pos_;
    pos_ += n;
    return p;
  }
  template <class T>
  T rd() {
    T v;
    std::memcpy(&v, take(sizeof(T)), sizeof(T));
    return v;
  }
  size_t rd_len() { return version_ == 1 ? (size_t)rd<uint32_t>() : (size_t)rd<uint64_t>(); }  // gguf.rs:399-407
  std::string rd_string() {
    size_t n = rd_len();
    const uint8_t* p = take(n);
    return std::string((const char*)p, n);
  }
  static GGUFValueType value_type(uint32_t t) {
    if (t > 12) throw Error(ErrorKind::FormatError, "failed to decode the value type for " + std::to_string(t));
    return (GGUFValueType)t;
  }
  // nesting depth of array-of-array values: a crafted file with 200k levels (12 bytes each) would otherwise overflow
  // the stack instead of raising FormatError; real files nest at most once
  static constexpr int MAX_ARRAY_DEPTH = 4;
  static size_t min_value_bytes(GGUFValueType t, uint32_t version) {
    switch (t) {
      case GGUFValueType::U8: case GGUFValueType::I8: case GGUFValueType::Bool: return 1;
      case GGUFValueType::U16: case GGUFValueType::I16: return 2;
      case GGUFValueType::U32: case GGUFValueType::I32: case GGUFValueType::F32: return 4;
      case GGUFValueType::U64: case GGUFValueType::I64: case GGUFValueType::F64: return 8;
      case GGUFValueType::String: return version == 1 ? 4 : 8;             // the length prefix
      case GGUFValueType::Array: return 4 + (version == 1 ? 4 : 8);        // element type + length
    }
    return 1;
  }
  GGUFValue rd_scalar(GGUFValueType t, int depth = 0) {
    GGUFValue x;
    x.type = t;
    switch (t) {
      case GGUFValueType::U8: x.v = (uint64_t)rd<uint8_t>(); break;
      case GGUFValueType::I8: x.v = (int64_t)rd<int8_t>(); break;
      case GGUFValueType::U16: x.v = (uint64_t)rd<uint16_t>(); break;
      case GGUFValueType::I16: x.v = (int64_t)rd<int16_t>(); break;
      case GGUFValueType::U32: x.v = (uint64_t)rd<uint32_t>(); break;
      case GGUFValueType::I32: x.v = (int64_t)rd<int32_t>(); break;
      case GGUFValueType::F32: x.v = (double)rd<float>(); break;
      case GGUFValueType::Bool: x.v = (uint64_t)rd<uint8_t>(); break;
      case GGUFValueType::String: x.v = rd_string(); break;
      case GGUFValueType::U64: x.v = rd<uint64_t>(); break;
      case GGUFValueType::I64: x.v = rd<int64_t>(); break;
      case GGUFValueType::F64: x.v = rd<double>(); break;
      case GGUFValueType::Array: x.v = rd_array(depth + 1); break;
    }
    return x;
  }
  GGUFArray rd_array(int depth = 1) {  // gguf.rs:341-371 (nested arrays included)
    if (depth > MAX_ARRAY_DEPTH)
      throw Error(ErrorKind::FormatError, "metadata arrays nest deeper than " + std::to_string(MAX_ARRAY_DEPTH) + " levels");
    GGUFArray a;
    a.elem_type = value_type(rd<uint32_t>());
    size_t n = rd_len();
    // every element occupies at least min_value_bytes in the file: bounds the reserve() below by the bytes that are left
    if (n > (len_ - pos_) / min_value_bytes(a.elem_type, version_))
      throw Error(ErrorKind::FormatError, "array length " + std::to_string(n) + " exceeds the file");
    a.items.reserve(n);
    for (size_t i = 0; i < n; i++) a.items.push_back(rd_scalar(a.elem_type, depth));
    return a;
  }
  void decode() {
    if (rd<uint32_t>() != 0x46554747u) throw Error(ErrorKind::FormatError, "Invalid magic number");  // "GGUF"
    uint32_t ver = rd<uint32_t>();
    if (ver < 1 || ver > 3) throw Error(ErrorKind::FormatError, "Unsupported version number: " + std::to_string(ver));
    version_ = ver;
    size_t tensor_count = rd_len(), kv_count = rd_len();
    if (tensor_count > len_ || kv_count > len_) throw Error(ErrorKind::FormatError, "implausible header counts");
    for (size_t i = 0; i < kv_count; i++) {
      std::string key = rd_string();
      GGUFValueType t = value_type(rd<uint32_t>());
      kv_[key] = rd_scalar(t);  // a repeated key keeps the last value (HashMap::insert)
    }
    auto arch = get_string("general.architecture");
    if (!arch) throw Error(ErrorKind::FormatError, "Missing string metadata general.architecture");
    architecture_ = *arch;
    tensors_.resize(tensor_count);
    for (auto& t : tensors_) {  // gguf.rs:632-646
      t.name = rd_string();
      uint32_t nd = rd<uint32_t>();
      if (nd > 8) throw Error(ErrorKind::FormatError, "tensor " + t.name + " has " + std::to_string(nd) + " dimensions");
      for (uint32_t d = 0; d < nd; d++) t.dimensions.push_back(version_ == 1 ? (size_t)rd<uint32_t>() : (size_t)rd<uint64_t>());
      t.ggml_type = rd<uint32_t>();
      t.offset = rd<uint64_t>();
    }
    const size_t al = (size_t)alignment();
    if (al == 0) throw Error(ErrorKind::FormatError, "general.alignment is 0");
    // Where the tensor data starts.  The reference always skips to the NEXT multiple of the alignment
    // (`position - position % alignment + alignment`, gguf.rs:722-724), i.e. a whole extra block when the tensor infos
    // already end on a boundary; the GGUF spec (and every llama.cpp / gguf-py writer) pads only when misaligned.  The two
    // agree unless pos % al == 0 (1 file in `al`).  In that case the file itself decides (below); a caller who knows better
    // passes data_start.
    size_t next = pos_ - (pos_ % al) + al;
    if (pos_ % al == 0) {
      size_t max_end = 0;
      bool sized = true;
      for (const auto& t : tensors_) {
        size_t bb = 0, be = 0, n = 1;
        if (!ggml_block_geometry_(t.ggml_type, &bb, &be)) { sized = false; break; }
        for (size_t d : t.dimensions)
          if (__builtin_mul_overflow(n, d, &n)) { sized = false; break; }
        if (!sized) break;
        const size_t end = (size_t)t.offset + n / be * bb;
        if (end > max_end) max_end = end;
      }
      // A writer pads the data section to the alignment at most: under the RIGHT convention the file ends less than one
      // alignment block after the last tensor (exactly at its end, at its padded end, or a few trailing bytes later); under the
      // wrong one the slack is off by a whole block.  slack(spec) = slack(reference) + al, so at most one of them lies in [0, al).
      const bool known = sized && pos_ <= len_ && len_ - pos_ >= max_end;
      const size_t slack_spec = known ? len_ - pos_ - max_end : 0;
      const bool spec_ok = known && slack_spec < al, ref_ok = known && slack_spec >= al && slack_spec - al < al;
      int conv;
      if (data_start_override_ == 0 || data_start_override_ == 1)
        conv = data_start_override_;
      else if (spec_ok || !sized || tensors_.empty())
        conv = 0;  // (tensor sizes unknown, or nothing to place: the spec's start)
      else if (ref_ok)
        conv = 1;
      else
        // a whole alignment block or more of trailing bytes (or a truncated file): the file does not say where its data starts,
        // and guessing wrong shifts EVERY tensor by one block without any error downstream (round-2 review finding).  Refused
        // unless the caller says which convention wrote it (round-3 advisor: a spec file with trailing bytes must stay loadable).
        throw Error(ErrorKind::FormatError,
                    "the tensor infos end on an alignment boundary and the file ends a whole alignment block or more after the last "
                    "tensor under the GGUF spec's data start and under the reference's (gguf.rs:722-724) alike: ambiguous file -- "
                    "pass data_start = 0 (spec) or 1 (reference)");
      if (conv == 0) next = pos_;
      data_start_convention_ = conv;
    }
    (void)take(next - pos_);
    data_off_ = pos_;
    const size_t data_len = len_ - data_off_;
    for (size_t i = 0; i < tensors_.size(); i++) {  // gguf.rs:737-759
      const size_t lo = (size_t)tensors_[i].offset;
      const size_t hi = i + 1 < tensors_.size() ? (size_t)tensors_[i + 1].offset : data_len;
      if (lo > hi || hi > data_len) throw Error(ErrorKind::FormatError, "tensor " + tensors_[i].name + " lies outside the data section");
      tensors_[i].data = base_ + data_off_ + lo;
      tensors_[i].data_len = hi - lo;
    }
  }

  static size_t align_up_(size_t v, size_t a) { return (v + a - 1) / a * a; }
  static bool ggml_block_geometry_(uint32_t t, size_t* bb, size_t* be);
  int data_start_convention_ = 0;  // 0 = GGUF spec (pad only when misaligned), 1 = the reference's always-skip (detected)
  int data_start_override_ = -1;
  int fd_ = -1;
  const uint8_t* base_ = nullptr;
  size_t len_ = 0, pos_ = 0, data_off_ = 0;
  uint32_t version_ = 0;
  std::string architecture_;
  std::map<std::string, GGUFValue> kv_;
  std::vector<GGUFTensorInfo> tensors_;
};

// bytes per block / elements per block of the GGML types this backend stores (buf/api.rs, Appendix A of SURVEY.md)
inline bool ggml_block_geometry(uint32_t t, size_t* block_bytes, size_t* block_elems) {
  switch (t) {
    case 0: *block_bytes = 4; *block_elems = 1; return true;       // F32
    case 1: *block_bytes = 2; *block_elems = 1; return true;       // F16
    case 2: *block_bytes = 18; *block_elems = 32; return true;     // Q4_0
    case 3: *block_bytes = 20; *block_elems = 32; return true;     // Q4_1
    case 6: *block_bytes = 22; *block_elems = 32; return true;     // Q5_0
    case 7: *block_bytes = 24; *block_elems = 32; return true;     // Q5_1
    case 8: *block_bytes = 34; *block_elems = 32; return true;     // Q8_0
    case 9: *block_bytes = 36; *block_elems = 32; return true;     // Q8_1
    case 10: *block_bytes = 84; *block_elems = 256; return true;   // Q2_K
    case 11: *block_bytes = 110; *block_elems = 256; return true;  // Q3_K
    case 12: *block_bytes = 144; *block_elems = 256; return true;  // Q4_K
    case 13: *block_bytes = 176; *block_elems = 256; return true;  // Q5_K (the reference's field order, buf_q5_k.rs:13-21)
    case 14: *block_bytes = 210; *block_elems = 256; return true;  // Q6_K
    case 15: *block_bytes = 292; *block_elems = 256; return true;  // Q8_K
    default: return false;
  }
}
inline bool GGUFFile::ggml_block_geometry_(uint32_t t, size_t* bb, size_t* be) { return ggml_block_geometry(t, bb, be); }

// CpuLlamaModelLoader::load_config (model.rs:545-625), llama architecture
inline LlamaConfig load_llama_config(const GGUFFile& gf) {
  if (gf.architecture() != "llama") throw Error(ErrorKind::ModelError, "unsupported architecture " + gf.architecture());
  const std::string p = "llama";
  auto need_u32 = [&](const std::string& k) -> size_t {
    auto v = gf.get_u32(k);
    if (!v) throw Error(ErrorKind::ModelError, "missing u32 metadata " + k);  // the reference unwrap()s here
    return *v;
  };
  LlamaConfig c;
  c.n_heads = need_u32(p + ".attention.head_count");
  c.n_layers = need_u32(p + ".block_count");
  c.hidden_dim = need_u32(p + ".feed_forward_length");
  c.n_kv_heads = need_u32(p + ".attention.head_count_kv");
  c.seq_len = need_u32(p + ".context_length");
  auto vocab = gf.get_string_array_len("tokenizer.ggml.tokens");
  if (!vocab) throw Error(ErrorKind::ModelError, "missing string array metadata tokenizer.ggml.tokens");
  c.vocab_size = *vocab;
  c.embedding_dim = need_u32(p + ".embedding_length");
  auto eps = gf.get_f32(p + ".attention.layer_norm_rms_epsilon");
  if (!eps) throw Error(ErrorKind::ModelError, "missing f32 metadata " + p + ".attention.layer_norm_rms_epsilon");
  c.rms_norm_eps = *eps;
  if (auto rot = gf.get_u32(p + ".rope.dimension_count")) c.rope_dim = (size_t)*rot;
  return c;
}

// load_tensor (model.rs:462-495) onto the hip device: dims reversed, bytes = the info's data slice, trimmed to the
// tensor's own size (the slice carries the padding up to the next tensor; from_bytes only needs whole blocks)
inline HipTensor load_gguf_tensor(const GGUFFile& gf, const std::string& name, const HipTensorDeviceRef& device) {
  const GGUFTensorInfo* info = gf.get_tensor_info(name);
  if (!info) throw Error(ErrorKind::TensorNotFound, "failed to find tensor " + name);
  std::vector<size_t> dims(info->dimensions.rbegin(), info->dimensions.rend());
  size_t bb = 0, be = 0;
  if (!ggml_block_geometry(info->ggml_type, &bb, &be))
    throw Error(ErrorKind::NotImplemented, "tensor " + name + ": ggml type " + std::to_string(info->ggml_type) + " is not supported by the hip backend");
  size_t n = 1;
  for (size_t d : dims)
    if (__builtin_mul_overflow(n, d, &n))
      throw Error(ErrorKind::FormatError, "tensor " + name + " " + fmt_dims(dims) + ": element count overflows");
  if (dims.empty() || n % be != 0 || dims.back() % be != 0)
    throw Error(ErrorKind::TensorError, "tensor " + name + " " + fmt_dims(dims) + " is not a whole number of blocks per row");
  if (info->ggml_type == 13) {
    // Q5_K is read in the REFERENCE's field order (qs | qh | scales | d | dmin, buf_q5_k.rs:13-21), which is not ggml's
    // (d | dmin | scales | qh | qs): a llama.cpp Q5_K / *_K_M file decodes to garbage in crabml, and identically here.  Say so, once.
    static bool warned = false;
    if (!warned) {
      warned = true;
      fprintf(stderr, "crabml_hip: WARNING: %s is Q5_K; blocks are read in crabml's own field order (buf_q5_k.rs:13-21), NOT ggml's -- a file "
                      "written by llama.cpp decodes wrongly (as it does in the reference)\n", name.c_str());
    }
  }
  const size_t nbytes = n / be * bb;
  if (nbytes > info->data_len) throw Error(ErrorKind::FormatError, "tensor " + name + " needs " + std::to_string(nbytes) + " bytes, the file holds " + std::to_string(info->data_len));
  return HipTensor::from_cpu(info->data, nbytes, dims, (GGMLType)info->ggml_type, device);
}

// CpuLlamaModelLoader::load_weights (model.rs:140-300, "llama" arm) + output.weight optional (model.rs:437)
inline std::shared_ptr<LlamaWeights<HipTensor>> load_llama_weights(const GGUFFile& gf, const LlamaConfig& conf,
                                                                  const HipTensorDeviceRef& device) {
  auto w = std::make_shared<LlamaWeights<HipTensor>>();
  auto f32 = [&](const std::string& name) {  // `.dequantize(GGMLType::F32)` of the norm weights (model.rs:267-281)
    HipTensor t = load_gguf_tensor(gf, name, device);
    return t.dtype() == GGMLType::F32 ? t : t.dequantize(GGMLType::F32);
  };
  w->token_embed = load_gguf_tensor(gf, "token_embd.weight", device);
  for (size_t l = 0; l < conf.n_layers; l++) {
    const std::string b = "blk." + std::to_string(l) + ".";
    w->wq.push_back(load_gguf_tensor(gf, b + "attn_q.weight", device));
    w->wk.push_back(load_gguf_tensor(gf, b + "attn_k.weight", device));
    w->wv.push_back(load_gguf_tensor(gf, b + "attn_v.weight", device));
    w->wo.push_back(load_gguf_tensor(gf, b + "attn_output.weight", device));
    w->ffn_gate_weight.push_back(load_gguf_tensor(gf, b + "ffn_gate.weight", device));
    w->ffn_down_weight.push_back(load_gguf_tensor(gf, b + "ffn_down.weight", device));
    w->ffn_up_weight.push_back(load_gguf_tensor(gf, b + "ffn_up.weight", device));
    w->rms_att_weight.push_back(f32(b + "attn_norm.weight"));
    w->rms_ffn_weight.push_back(f32(b + "ffn_norm.weight"));
  }
  w->rms_final_weight = f32("output_norm.weight");
  if (gf.get_tensor_info("output.weight")) w->output_weight = load_gguf_tensor(gf, "output.weight", device);
  return w;
}

}  // namespace crabml_host
