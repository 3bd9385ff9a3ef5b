// hip_tensor.hpp -- HipTensor / HipTensorDevice: the host-side mirror of crabml's `Tensor` trait
// (crabml-core/src/tensor/api.rs:11-79) over the C ABI of libcrabml_hip.so.
//
// This is what the `crabml-hip` Rust crate would contain; there is no Rust
// toolchain in this image, so the same thin layer is written in C++: it owns the TensorStrider, does the
// validation CpuTensor does (crabml-core/src/cpu/cpu_tensor.rs:126-446) with the same error kind
// (ErrorKind::TensorError) and messages, and forwards to one C-ABI call per trait method.  In-place
// methods consume-and-return in Rust; here they mutate and return a handle that shares the buffer.
#pragma once
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "crabml_hip.h"
#include "crabml_hip_debug.h"  // the parity hooks the tests reach through this mirror
#include "strider.hpp"

namespace crabml_host {

enum class GGMLType : uint32_t { F32 = 0, F16 = 1, Q4_0 = 2, Q4_1 = 3, Q5_0 = 6, Q5_1 = 7, Q8_0 = 8, Q8_1 = 9, Q2K = 10, Q3K = 11, Q4K = 12, Q5K = 13, Q6K = 14, Q8K = 15 };
enum class RopeMode : uint32_t { Llama = 0, Neox = 1 };

// replaces WgpuTensorDeviceOptions (crabml-wgpu/src/wgpu_device.rs:9-38)
struct HipTensorDeviceOptions {
  int device_ordinal = 0;
  bool debug_named_tensor = false;
  bool strict_order = false;  // CRABML_HIP_FLAG_STRICT_ORDER: every sum in the reference's scalar order (bit-identical)
  // how the Tensor calls reach the GPU (crabml_hip.h, ABI version 2): "lazy" = recorded, run at export / sync, a Llama decode
  // token through the fused step (the default); "lazy-per-op" = recorded, always run op by op; "per-op" = one launch per
  // call, immediately (ABI version 1); "dry" = the record-only test device (needs CRABML_HIP_TEST_HOOKS=1)
  std::string mode = "lazy";
  void* stream = nullptr;
};

class HipTensorDevice {
 public:
  explicit HipTensorDevice(const HipTensorDeviceOptions& o) : opts(o) {
    crabml_hip_device_options_t c{};
    c.device_ordinal = o.device_ordinal;
    c.stream = o.stream;
    c.flags = o.strict_order ? CRABML_HIP_FLAG_STRICT_ORDER : 0;
    if (o.mode == "per-op")
      c.flags |= CRABML_HIP_FLAG_PER_OP;
    else if (o.mode == "lazy-per-op")
      c.flags |= CRABML_HIP_FLAG_LAZY_NO_FUSION;
    else if (o.mode == "dry" || o.mode == "dry-per-op")
      c.flags |= CRABML_HIP_FLAG_DRY | (o.mode == "dry-per-op" ? CRABML_HIP_FLAG_LAZY_NO_FUSION : 0);
    else if (o.mode != "lazy")
      throw Error(ErrorKind::BadInput, "HipTensorDevice: unknown mode `" + o.mode + "`");
    int rc = crabml_hip_device_create(&c, &dev_);
    if (rc != 0 || !dev_)
      throw Error(ErrorKind::Unexpected,
                  "crabml_hip_device_create failed: no usable HIP device (the hip backend has no CPU fallback)");
  }
  ~HipTensorDevice() {
    if (dev_) crabml_hip_device_destroy(dev_);
  }
  HipTensorDevice(const HipTensorDevice&) = delete;
  HipTensorDevice& operator=(const HipTensorDevice&) = delete;

  crabml_hip_device_t* raw() const { return dev_; }
  void sync() { check(crabml_hip_device_sync(dev_)); }
  size_t mem_in_use() const { return crabml_hip_device_mem_in_use(dev_); }
  // counters of the recorded-op queue (crabml_hip_debug.h)
  std::vector<uint64_t> lazy_stats() const {
    std::vector<uint64_t> v(12, 0);
    check(crabml_hip_debug_lazy_stats(dev_, v.data(), v.size()));
    return v;
  }

  // maps a C-ABI status back onto crabml::error::Error
  void check(int rc) const {
    if (rc == 0) return;
    char msg[512];
    crabml_hip_last_error(dev_, msg, sizeof msg);
    throw Error(static_cast<ErrorKind>(rc), msg);
  }

  // debug hook of the parity tests: with_name() snapshots (wgpu_device.rs:167-175, cpu_device.rs:126-132)
  void add_debug_tensor(const std::string& name, std::vector<float> v) {
    std::lock_guard<std::mutex> g(mu_);
    debug_[name] = std::move(v);
  }
  bool dump_debug_tensor(const std::string& name, std::vector<float>* out) {
    std::lock_guard<std::mutex> g(mu_);
    auto it = debug_.find(name);
    if (it == debug_.end()) return false;
    *out = it->second;
    return true;
  }

  HipTensorDeviceOptions opts;

 private:
  crabml_hip_device_t* dev_ = nullptr;
  std::mutex mu_;
  std::map<std::string, std::vector<float>> debug_;
};
using HipTensorDeviceRef = std::shared_ptr<HipTensorDevice>;

// RAII over crabml_hip_buf_t (Arc<buffer> in the Rust crate)
class BufRef {
 public:
  BufRef() = default;
  explicit BufRef(crabml_hip_buf_t* b) : b_(b) {}  // adopts one reference
  BufRef(const BufRef& o) : b_(o.b_) {
    if (b_) crabml_hip_buf_retain(b_);
  }
  BufRef(BufRef&& o) noexcept : b_(o.b_) { o.b_ = nullptr; }
  BufRef& operator=(BufRef o) {
    std::swap(b_, o.b_);
    return *this;
  }
  ~BufRef() {
    if (b_) crabml_hip_buf_release(b_);
  }
  crabml_hip_buf_t* get() const { return b_; }

 private:
  crabml_hip_buf_t* b_ = nullptr;
};

class HipTensor {
 public:
  using DeviceRef = HipTensorDeviceRef;

  HipTensor() = default;
  static RopeMode rope_mode_llama() { return RopeMode::Llama; }

  // ---- constructors ------------------------------------------------------------------------------
  // Tensor::from_cpu (api.rs:14-19)
  static HipTensor from_cpu(const void* buf, size_t nbytes, const std::vector<size_t>& shape, GGMLType dtype,
                            DeviceRef device) {
    crabml_hip_buf_t* b = nullptr;
    device->check(crabml_hip_buf_from_cpu(device->raw(), buf, nbytes, shape.data(), (int)shape.size(), (uint32_t)dtype, &b));
    return HipTensor(BufRef(b), dtype, TensorStrider(shape), std::move(device));
  }
  // WgpuTensor::new analogue used by the tests (wgpu_tensor.rs:31-63)
  static HipTensor from_f32(const std::vector<float>& v, const std::vector<size_t>& shape, DeviceRef device) {
    size_t n = 1;
    for (size_t s : shape) n *= s;
    if (v.size() != n)
      throw Error(ErrorKind::TensorError, "invalid shape " + fmt_dims(shape) + " for data of length " + std::to_string(v.size()));
    return from_cpu(v.data(), v.size() * 4, shape, GGMLType::F32, std::move(device));
  }
  // Tensor::alloc (api.rs:21-23; cpu_tensor.rs:138-165)
  static HipTensor alloc(const std::vector<size_t>& shape, GGMLType dtype, DeviceRef device) {
    if (dtype != GGMLType::F32 && dtype != GGMLType::F16) throw Error(ErrorKind::TensorError, "only f32/f16 is supported");
    size_t n = 1;
    for (size_t s : shape) n *= s;
    crabml_hip_buf_t* b = nullptr;
    device->check(crabml_hip_buf_alloc(device->raw(), n, (uint32_t)dtype, &b));
    return HipTensor(BufRef(b), dtype, TensorStrider(shape), std::move(device));
  }

  // ---- metadata (host only) ----------------------------------------------------------------------
  GGMLType dtype() const { return dtype_; }
  const std::vector<size_t>& shape() const { return strider_.shape(); }
  const TensorStrider& strider() const { return strider_; }
  bool is_contiguous() const { return strider_.is_contiguous(); }
  size_t buf_len() const { return crabml_hip_buf_len(buf_.get()); }
  const DeviceRef& device() const { return device_; }
  crabml_hip_buf_t* raw() const { return buf_.get(); }
  const std::string& name() const { return name_; }

  HipTensor resize(size_t axis, size_t n) const {  // cpu_tensor.rs:167-195
    if (axis >= shape().size())
      throw Error(ErrorKind::TensorError, "resize: axis " + std::to_string(axis) + " is larger than the current shape " + fmt_dims(shape()));
    std::vector<size_t> ns = shape();
    ns[axis] = n;
    size_t new_len = 1;
    for (size_t s : ns) new_len *= s;
    if (new_len > buf_len())
      throw Error(ErrorKind::TensorError, "resize: new shape " + fmt_dims(ns) + " is larger than the current shape " + fmt_dims(shape()));
    return view(strider_.resize(ns));
  }
  HipTensor reshape(const std::vector<size_t>& s) const { return view(strider_.reshape(s)); }
  HipTensor transpose(const std::vector<size_t>& d) const { return view(strider_.transpose(d)); }
  HipTensor with_strider(const TensorStrider& s) const { return view(s); }
  HipTensor with_name(const std::string& name) const {  // cpu_tensor.rs:232-241
    HipTensor t = *this;
    t.name_ = name;
    if (device_->opts.debug_named_tensor && dtype_ == GGMLType::F32) {
      std::vector<float> v(buf_len());
      device_->check(crabml_hip_export(device_->raw(), buf_.get(), v.data(), v.size()));
      device_->add_debug_tensor(name, std::move(v));
    }
    return t;
  }

  // ---- data movement -----------------------------------------------------------------------------
  // Tensor::export (api.rs:52): blocks until the stream drains
  std::vector<float> export_() const {
    if (!is_contiguous()) throw Error(ErrorKind::TensorError, "export: tensor is not contiguous");
    std::vector<float> v(strider_.len());
    device_->check(crabml_hip_export(device_->raw(), buf_.get(), v.data(), v.size()));
    return v;
  }
  // Tensor::export (api.rs:52: `fn export(&self, buf: &mut [f32])`) -- into the caller's buffer, as the reference's runner
  // does with its preallocated logits (llama2.rs:208)
  void export_into(std::vector<float>& v) const {
    if (!is_contiguous()) throw Error(ErrorKind::TensorError, "export: tensor is not contiguous");
    v.resize(strider_.len());
    device_->check(crabml_hip_export(device_->raw(), buf_.get(), v.data(), v.size()));
  }
  std::vector<uint8_t> export_raw() const {
    size_t nbytes = buf_len() * (dtype_ == GGMLType::F32 ? 4 : 2);
    std::vector<uint8_t> v(nbytes);
    device_->check(crabml_hip_export_raw(device_->raw(), buf_.get(), v.data(), nbytes));
    return v;
  }
  HipTensor dup() const {  // cpu_tensor.rs:333-337: copies the whole storage, then validates len == shape
    crabml_hip_buf_t* b = nullptr;
    device_->check(crabml_hip_dup(device_->raw(), buf_.get(), &b));
    BufRef nb(b);
    if (buf_len() != strider_.len())
      throw Error(ErrorKind::TensorError, "invalid shape " + fmt_dims(shape()) + " for data of length " + std::to_string(buf_len()));
    return HipTensor(nb, GGMLType::F32, TensorStrider(shape()), device_);
  }
  HipTensor contiguous() const {  // cpu_tensor.rs:294-304
    if (is_contiguous()) return *this;
    if (dtype_ != GGMLType::F32 && dtype_ != GGMLType::F16) throw Error(ErrorKind::TensorError, "contiguous: only f32/f16");
    crabml_hip_buf_t* b = nullptr;
    device_->check(crabml_hip_contiguous(device_->raw(), buf_.get(), shape().data(), strider_.strides().data(), (int)strider_.dims(), &b));
    return HipTensor(BufRef(b), dtype_, TensorStrider(shape()), device_);
  }
  void concatenate(const HipTensor& rhs, size_t axis) {  // cpu_tensor.rs:251-292
    if (dtype_ != GGMLType::F32 && dtype_ != GGMLType::F16)
      throw Error(ErrorKind::TensorError, "only f32/f16 is supported on concatenate");
    if (rhs.dtype_ != GGMLType::F32 && rhs.dtype_ != GGMLType::F16)
      throw Error(ErrorKind::TensorError, "only f32/f16 is supported on concatenate rhs");
    if (rhs.strider_.dims() != strider_.dims() || axis >= strider_.dims())
      throw Error(ErrorKind::TensorError, "shape mismatch on concatenate, want " + fmt_dims(shape()) + " but got " + fmt_dims(rhs.shape()));
    for (size_t i = 0; i < shape().size(); i++)
      if (i != axis && shape()[i] != rhs.shape()[i])
        throw Error(ErrorKind::TensorError, "shape mismatch on concatenate, want " + fmt_dims(shape()) + " but got " + fmt_dims(rhs.shape()));
    device_->check(crabml_hip_concatenate(device_->raw(), buf_.get(), shape().data(), strider_.strides().data(), rhs.buf_.get(),
                                          rhs.shape().data(), rhs.strider_.strides().data(), (int)strider_.dims(), (int)axis));
    std::vector<size_t> ns = shape();
    ns[axis] += rhs.shape()[axis];
    strider_ = strider_.resize(ns);
  }
  void copy_rows_from(const HipTensor& src, const std::vector<size_t>& rows) {  // cpu_tensor.rs:306-331
    if (!is_contiguous()) throw Error(ErrorKind::TensorError, "dst tensor is not contiguous");
    if (!src.is_contiguous()) throw Error(ErrorKind::TensorError, "src tensor is not contiguous");
    if (src.strider_.dims() != 2 && src.strider_.dims() != 1)
      throw Error(ErrorKind::TensorError, "copy_rows_from: src tensor is not 2d or 1d");
    size_t cols = shape().back();
    device_->check(crabml_hip_copy_rows_from(device_->raw(), buf_.get(), src.buf_.get(), cols, rows.data(), rows.size()));
  }

  // Tensor::dequantize (api.rs:26; cpu_tensor.rs:135-151): an f32 copy of a stored-format tensor, rows dequantized
  // exactly as copy_rows_from does
  HipTensor dequantize(GGMLType dtype) const {
    if (dtype != GGMLType::F32) throw Error(ErrorKind::NotImplemented, "dequantize: only to F32");
    if (strider_.dims() != 1 && strider_.dims() != 2) throw Error(ErrorKind::TensorError, "dequantize: tensor is not 2d or 1d");
    HipTensor out = alloc(shape(), GGMLType::F32, device_);
    std::vector<size_t> rows(strider_.dims() == 2 ? shape()[0] : 1);
    for (size_t i = 0; i < rows.size(); i++) rows[i] = i;
    out.copy_rows_from(*this, rows);
    return out;
  }

  // ---- compute -----------------------------------------------------------------------------------
  HipTensor rope_inplace(RopeMode mode, size_t pos, size_t rope_dims) const {  // rope.rs:10-45
    need_contig("rope");
    if (strider_.dims() != 2 && strider_.dims() != 3) throw Error(ErrorKind::TensorError, "rope: tensor must be 2-d or 3-d");
    size_t n_batch, bi_stride, head_dim;
    if (strider_.dims() == 2) {
      n_batch = 1;
      bi_stride = strider_.len();
      head_dim = shape()[1];
    } else {
      n_batch = shape()[0];
      bi_stride = strider_.strides()[0];
      head_dim = shape()[2];
    }
    device_->check(crabml_hip_rope_inplace(device_->raw(), buf_.get(), n_batch, bi_stride, head_dim, (uint32_t)mode, pos, rope_dims));
    return *this;
  }
  HipTensor rms_norm_inplace(float eps) const {  // rms_norm.rs:9-31
    need_contig("rms_norm");
    if (shape().size() != 1 && shape().size() != 2) throw Error(ErrorKind::TensorError, "rms_norm: tensor must be 1-d or 2-d");
    size_t rows = shape().size() == 1 ? 1 : shape()[0];
    size_t cols = shape().size() == 1 ? shape()[0] : shape()[1];
    device_->check(crabml_hip_rms_norm_inplace(device_->raw(), buf_.get(), rows, cols, eps));
    return *this;
  }
  HipTensor softmax_inplace(size_t axis) const {  // softmax.rs:11-57
    if (strider_.dims() != 2 && strider_.dims() != 3) throw Error(ErrorKind::TensorError, "softmax: tensor must be 2-d or 3-d");
    need_contig("softmax");
    if (axis != strider_.dims() - 1)
      throw Error(ErrorKind::TensorError, "only axis=" + std::to_string(strider_.dims() - 1) + " is supported on a " +
                                              std::to_string(strider_.dims()) + " dimensions tensor");
    size_t cols = shape().back();
    size_t rows = cols ? strider_.len() / cols : 0;
    device_->check(crabml_hip_softmax_inplace(device_->raw(), buf_.get(), rows, cols));
    return *this;
  }
  // silu/gelu run over the whole storage, like buf.as_f32_mut().iter_mut() (silu.rs:8)
  HipTensor silu_inplace() const {
    device_->check(crabml_hip_silu_inplace(device_->raw(), buf_.get(), buf_len()));
    return *this;
  }
  HipTensor gelu_inplace() const {
    device_->check(crabml_hip_gelu_inplace(device_->raw(), buf_.get(), buf_len()));
    return *this;
  }
  HipTensor mul_inplace(const HipTensor& rhs) const { return binary(rhs, true); }
  HipTensor add_inplace(const HipTensor& rhs) const { return binary(rhs, false); }
  HipTensor scale_inplace(float f) const {  // cpu_tensor.rs:404-410
    need_contig("scale");
    device_->check(crabml_hip_scale_inplace(device_->raw(), buf_.get(), buf_len(), f));
    return *this;
  }
  // Tensor::matmul_vec (api.rs:76; cpu_tensor.rs:371-386): (m,k) @ (k,) -> (m,) ; (m,k) @ (b,k) -> (b,m)
  HipTensor matmul_vec(const HipTensor& x) const {
    if (!is_contiguous() || !x.is_contiguous()) throw Error(ErrorKind::TensorError, "matmul_vec: tensors must be contiguous");
    if (strider_.dims() != 2 || (x.strider_.dims() != 1 && x.strider_.dims() != 2))
      throw Error(ErrorKind::TensorError, "matmul_vec: expect (m,k) @ (k,) or (m,k) @ (b,k)");
    if (shape().back() != x.shape().back())
      throw Error(ErrorKind::TensorError, "matmul_vec: inner dims differ: " + fmt_dims(shape()) + " vs " + fmt_dims(x.shape()));
    size_t m = shape()[0], k = shape()[1];
    size_t b = x.shape().size() == 1 ? 1 : x.shape()[0];
    std::vector<size_t> shape_c = x.shape().size() == 1 ? std::vector<size_t>{m} : std::vector<size_t>{b, m};
    crabml_hip_buf_t* o = nullptr;
    device_->check(crabml_hip_matmul_vec(device_->raw(), buf_.get(), m, k, x.buf_.get(), b, &o));
    return HipTensor(BufRef(o), GGMLType::F32, TensorStrider(shape_c), x.device_);
  }
  // Tensor::batch_matmul (api.rs:78; cpu_tensor.rs:352-367)
  HipTensor batch_matmul(const HipTensor& b) const {
    if (strider_.dims() != 3 || b.strider_.dims() != 3) throw Error(ErrorKind::TensorError, "batch_matmul: both tensors must be 3-d");
    if (!is_contiguous()) throw Error(ErrorKind::TensorError, "batch_matmul: lhs must be contiguous");
    const auto& bs = b.strider_.strides();
    if (!(bs[1] == 1 || bs[2] == 1)) throw Error(ErrorKind::TensorError, "batch_matmul: rhs must be contiguous on k or n");
    if (shape()[2] != b.shape()[1]) throw Error(ErrorKind::TensorError, "batch_matmul: inner dims differ");
    size_t ba = shape()[0], m = shape()[1], k = shape()[2], bb = b.shape()[0], n = b.shape()[2];
    crabml_hip_buf_t* o = nullptr;
    device_->check(crabml_hip_batch_matmul(device_->raw(), buf_.get(), ba, m, k, b.buf_.get(), bb, n, bs[0], bs[1], bs[2], &o));
    return HipTensor(BufRef(o), GGMLType::F32, TensorStrider({ba, m, n}), device_);
  }

  // ---- parity hooks --------------------------------------------------------------------------------
  std::vector<uint8_t> debug_quantize(GGMLType qtype) const {
    size_t n = strider_.len();
    size_t be = (qtype == GGMLType::Q8K) ? 256 : 32;
    size_t bb = qtype == GGMLType::Q8_0 ? 34 : qtype == GGMLType::Q8_1 ? 36 : 292;
    std::vector<uint8_t> out(n / be * bb);
    device_->check(crabml_hip_debug_quantize(device_->raw(), buf_.get(), n, (uint32_t)qtype, out.data(), out.size()));
    return out;
  }
  std::vector<int32_t> debug_block_dots(size_t row, const HipTensor& x) const {
    size_t m = shape()[0], k = shape()[1];
    // Q6_K / Q2_K / Q3_K: one per 16-element scale group
    std::vector<int32_t> out(dtype_ == GGMLType::Q6K || dtype_ == GGMLType::Q2K || dtype_ == GGMLType::Q3K ? k / 16 : k / 32);
    device_->check(crabml_hip_debug_block_dots(device_->raw(), buf_.get(), m, k, row, x.buf_.get(), out.data()));
    return out;
  }
  // the production K-quant loops' integers per super-block: (isum, msum) pairs + the kernel's own f32 value
  std::pair<std::vector<int32_t>, float> debug_superblock_ints(size_t row, const HipTensor& x, int variant) const {
    size_t m = shape()[0], k = shape()[1];
    std::vector<int32_t> out(k / 256 * 2);
    float v = 0.f;
    device_->check(crabml_hip_debug_superblock_ints(device_->raw(), buf_.get(), m, k, row, x.buf_.get(), variant, out.data(), &v));
    return {out, v};
  }
  // the same out of the matrix-core GEMM: x = (b, k) rows; returns ints[b][m][k/256][2] and the GEMM's f32 (b, m)
  std::pair<std::vector<int32_t>, std::vector<float>> debug_gemm_ints(const HipTensor& x, size_t b) const {
    size_t m = shape()[0], k = shape()[1];
    std::vector<int32_t> ints(b * m * (k / 256) * 2);
    std::vector<float> out(b * m);
    device_->check(crabml_hip_debug_gemm_ints(device_->raw(), buf_.get(), m, k, x.buf_.get(), b, ints.data(), out.data()));
    return {ints, out};
  }

 private:
  HipTensor(BufRef buf, GGMLType dtype, TensorStrider strider, DeviceRef device)
      : device_(std::move(device)), buf_(std::move(buf)), dtype_(dtype), strider_(std::move(strider)) {}
  HipTensor view(TensorStrider s) const { return HipTensor(buf_, dtype_, std::move(s), device_); }  // name: None
  void need_contig(const char* op) const {
    if (!is_contiguous()) throw Error(ErrorKind::TensorError, std::string(op) + ": tensor is not contiguous");
  }
  HipTensor binary(const HipTensor& rhs, bool mul) const {  // arithmetic.rs:11-14
    if (rhs.buf_len() == 0 || buf_len() % rhs.buf_len() != 0)
      throw Error(ErrorKind::TensorError, "lhs length is not a multiple of rhs length");
    if (!(shape().back() == rhs.shape().back() || rhs.buf_len() == 1))
      throw Error(ErrorKind::TensorError, "last dims differ: " + fmt_dims(shape()) + " vs " + fmt_dims(rhs.shape()));
    if (!is_contiguous() || !rhs.is_contiguous()) throw Error(ErrorKind::TensorError, "tensors must be contiguous");
    if (mul)
      device_->check(crabml_hip_mul_inplace(device_->raw(), buf_.get(), buf_len(), rhs.buf_.get(), rhs.buf_len()));
    else
      device_->check(crabml_hip_add_inplace(device_->raw(), buf_.get(), buf_len(), rhs.buf_.get(), rhs.buf_len()));
    return *this;
  }

  // NOTE: device_ is declared before buf_ so the buffer is released while its device is still alive.
  DeviceRef device_;
  BufRef buf_;
  GGMLType dtype_ = GGMLType::F32;
  TensorStrider strider_;
  std::string name_;
};

}  // namespace crabml_host
