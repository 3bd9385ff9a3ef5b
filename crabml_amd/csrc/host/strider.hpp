// strider.hpp -- host-side shape/stride algebra, mirroring crabml-core/src/tensor/strider.rs.
// Stays on the host side of the C ABI: reshape / transpose / resize / with_strider never touch the GPU.
#pragma once
#include <cstddef>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace crabml_host {

// crabml::error::ErrorKind (crabml-core/src/error.rs:5-33); values equal the C ABI status codes.
enum class ErrorKind : int {
  Unexpected = 1, IOError = 2, TensorNotFound = 3, ModelError = 4, BadInput = 5, FormatError = 6,
  TensorError = 7, ChatTemplateNotFound = 8, NotImplemented = 9
};

// crabml::error::Error -- Rust's `Result<T>` becomes a C++ exception on this side of the boundary.
struct Error : std::runtime_error {
  ErrorKind kind;
  Error(ErrorKind k, const std::string& msg) : std::runtime_error(msg), kind(k) {}
};

inline std::string fmt_dims(const std::vector<size_t>& v) {
  std::ostringstream os;
  os << "[";
  for (size_t i = 0; i < v.size(); i++) os << (i ? ", " : "") << v[i];
  os << "]";
  return os.str();
}

class TensorStrider {
 public:
  TensorStrider() = default;
  explicit TensorStrider(std::vector<size_t> shape) : shape_(std::move(shape)) { strides_ = compute_strides(shape_); }
  TensorStrider(std::vector<size_t> shape, std::vector<size_t> strides)
      : shape_(std::move(shape)), strides_(std::move(strides)) {}

  const std::vector<size_t>& shape() const { return shape_; }
  const std::vector<size_t>& strides() const { return strides_; }
  size_t dims() const { return shape_.size(); }
  size_t len() const {
    size_t n = 1;
    for (size_t s : shape_) n *= s;
    return n;
  }
  bool is_empty() const { return len() == 0; }

  // strider.rs:36-51 -- storage and strides are unchanged
  TensorStrider resize(const std::vector<size_t>& new_shape) const {
    if (new_shape.size() != shape_.size())
      throw Error(ErrorKind::TensorError,
                  "invalid new shape " + fmt_dims(new_shape) + " for a tensor of shape " + fmt_dims(shape_));
    return TensorStrider(new_shape, strides_);
  }

  size_t at(const std::vector<size_t>& idx) const {  // strider.rs:65-86
    if (idx.size() != shape_.size())
      throw Error(ErrorKind::TensorError, "invalid index " + fmt_dims(idx) + " for tensor of shape " + fmt_dims(shape_));
    for (size_t i = 0; i < idx.size(); i++)
      if (idx[i] >= shape_[i])
        throw Error(ErrorKind::TensorError,
                    "invalid index " + fmt_dims(idx) + " for tensor of shape " + fmt_dims(shape_));
    return at_unchecked(idx);
  }
  size_t at_unchecked(const std::vector<size_t>& idx) const {
    size_t off = 0;
    for (size_t i = 0; i < idx.size() && i < strides_.size(); i++) off += idx[i] * strides_[i];
    return off;
  }

  std::vector<size_t> iter() const {  // strider.rs:96-104
    std::vector<size_t> pos(shape_.size(), 0), out;
    size_t n = len();
    out.reserve(n);
    for (size_t c = 0; c < n; c++) {
      out.push_back(at_unchecked(pos));
      for (size_t i = pos.size(); i-- > 0;) {
        if (pos[i] + 1 < shape_[i]) {
          pos[i]++;
          break;
        }
        pos[i] = 0;
      }
    }
    return out;
  }

  TensorStrider reshape(const std::vector<size_t>& shape) const {  // strider.rs:143-160
    if (!is_contiguous()) throw Error(ErrorKind::TensorError, "not contiguous");
    size_t n = 1;
    for (size_t s : shape) n *= s;
    if (n != len())
      throw Error(ErrorKind::TensorError,
                  "invalid shape " + fmt_dims(shape) + " for a tensor's origin shape " + fmt_dims(shape_));
    return TensorStrider(shape);
  }

  TensorStrider transpose(const std::vector<size_t>& dims) const {  // strider.rs:162-180
    if (dims.size() != shape_.size())
      throw Error(ErrorKind::TensorError, "invalid dims " + fmt_dims(dims) + " for a tensor of shape " + fmt_dims(shape_));
    std::vector<size_t> ns, nst;
    for (size_t d : dims) {
      if (d >= shape_.size()) throw Error(ErrorKind::TensorError, "invalid dims " + fmt_dims(dims));
      ns.push_back(shape_[d]);
      nst.push_back(strides_[d]);
    }
    return TensorStrider(ns, nst);
  }

  bool is_contiguous() const { return is_contiguous_on_axis(0); }
  bool is_contiguous_on_axis(size_t axis) const {  // strider.rs:188-206
    if (strides_.empty()) return true;
    if (strides_.back() != 1) return false;
    size_t last = 1;
    for (size_t i = shape_.size(); i-- > axis;) {
      if (last != strides_[i]) return false;
      last *= shape_[i];
    }
    return true;
  }

 private:
  static std::vector<size_t> compute_strides(const std::vector<size_t>& shape) {  // strider.rs:213-221
    std::vector<size_t> st;
    if (shape.empty()) return st;
    st.push_back(1);
    for (size_t i = 0; i + 1 < shape.size(); i++) st.push_back(st.back() * shape[shape.size() - i - 1]);
    return std::vector<size_t>(st.rbegin(), st.rend());
  }
  std::vector<size_t> shape_, strides_;
};

}  // namespace crabml_host
