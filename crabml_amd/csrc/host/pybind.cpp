// pybind.cpp -- python face of the host-side mirror (HipTensor / HipTensorDevice / TensorStrider /
// Llama2Runner<HipTensor>), so the pytest suite can be written like the reference's own Rust tests
// (crabml-core/src/cpu/cpu_tensor.rs:455-606, crabml-wgpu/src/wgpu_tensor.rs:742-1099).
// Everything here calls straight through to the C ABI of libcrabml_hip.so; there is no CPU fallback.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <chrono>

#include "gguf.hpp"
#include "hip_llama.hpp"
#include "hip_tensor.hpp"
#include "llama2_runner.hpp"

namespace py = pybind11;
using namespace crabml_host;

static py::object g_tensor_error;  // python exception class crabml_amd.TensorError

static std::vector<size_t> to_shape(const py::sequence& s) {
  std::vector<size_t> v;
  for (auto it : s) v.push_back(it.cast<size_t>());
  return v;
}

using Runner = Llama2Runner<HipTensor>;
using Weights = LlamaWeights<HipTensor>;

PYBIND11_MODULE(_host, m) {
  m.doc() = "crabml-hip host mirror over the C ABI of libcrabml_hip.so";

  static py::exception<Error> exc(m, "CrabmlError");
  py::register_exception_translator([](std::exception_ptr p) {
    try {
      if (p) std::rethrow_exception(p);
    } catch (const Error& e) {
      std::string msg = std::string("ErrorKind(") + std::to_string((int)e.kind) + "): " + e.what();
      PyErr_SetString(exc.ptr(), msg.c_str());
    }
  });
  m.def("abi_version", []() { return crabml_hip_abi_version(); });
  // the runner's greedy sampler (sampler.rs:109-116: Iterator::max_by keeps the LAST maximum)
  m.def("sample_argmax", [](py::array_t<float, py::array::c_style | py::array::forcecast> a) {
    std::vector<float> v(a.data(), a.data() + a.size());
    return sample_argmax(v);
  });

  py::enum_<GGMLType>(m, "GGMLType")
      .value("F32", GGMLType::F32).value("F16", GGMLType::F16).value("Q4_0", GGMLType::Q4_0)
      .value("Q4_1", GGMLType::Q4_1).value("Q8_0", GGMLType::Q8_0).value("Q8_1", GGMLType::Q8_1)
      .value("Q5_0", GGMLType::Q5_0).value("Q5_1", GGMLType::Q5_1).value("Q2K", GGMLType::Q2K).value("Q3K", GGMLType::Q3K)
      .value("Q4K", GGMLType::Q4K).value("Q5K", GGMLType::Q5K).value("Q6K", GGMLType::Q6K).value("Q8K", GGMLType::Q8K);
  py::enum_<RopeMode>(m, "RopeMode").value("Llama", RopeMode::Llama).value("Neox", RopeMode::Neox);

  py::class_<TensorStrider>(m, "TensorStrider")
      .def(py::init([](const py::sequence& s) { return TensorStrider(to_shape(s)); }))
      .def(py::init([](const py::sequence& s, const py::sequence& st) { return TensorStrider(to_shape(s), to_shape(st)); }))
      .def("shape", [](const TensorStrider& s) { return s.shape(); })
      .def("strides", [](const TensorStrider& s) { return s.strides(); })
      .def("dims", &TensorStrider::dims)
      .def("len", &TensorStrider::len)
      .def("resize", [](const TensorStrider& s, const py::sequence& ns) { return s.resize(to_shape(ns)); })
      .def("at", [](const TensorStrider& s, const py::sequence& i) { return s.at(to_shape(i)); })
      .def("iter", &TensorStrider::iter)
      .def("reshape", [](const TensorStrider& s, const py::sequence& ns) { return s.reshape(to_shape(ns)); })
      .def("transpose", [](const TensorStrider& s, const py::sequence& d) { return s.transpose(to_shape(d)); })
      .def("is_contiguous", &TensorStrider::is_contiguous);

  py::class_<HipTensorDevice, std::shared_ptr<HipTensorDevice>>(m, "HipTensorDevice")
      .def(py::init([](int ordinal, bool debug_named_tensor, size_t stream, bool strict_order, const std::string& mode) {
             HipTensorDeviceOptions o;
             o.device_ordinal = ordinal;
             o.debug_named_tensor = debug_named_tensor;
             o.stream = reinterpret_cast<void*>(stream);
             o.strict_order = strict_order;
             o.mode = mode;
             return std::make_shared<HipTensorDevice>(o);
           }),
           py::arg("device_ordinal") = 0, py::arg("debug_named_tensor") = false, py::arg("stream") = 0,
           py::arg("strict_order") = false, py::arg("mode") = "lazy")
      .def("numa_node",
           [](HipTensorDevice& d) {
             int32_t node = -1;
             d.check(crabml_hip_debug_device_numa_node(d.raw(), &node));
             return (int)node;
           })
      .def("lazy_stats",
           [](HipTensorDevice& d) {
             const std::vector<uint64_t> v = d.lazy_stats();
             py::dict r;
             const char* names[12] = {"recorded", "replayed", "fused_tokens", "fused_ops", "segments", "aborts", "learned", "deferred_bound",
                                      "wait_ns", "pinned_exports", "reactivated", "reaped"};
             for (int i = 0; i < 12; i++) r[names[i]] = v[i];
             return r;
           })
      .def("sync", &HipTensorDevice::sync, py::call_guard<py::gil_scoped_release>())
      .def("mem_in_use", &HipTensorDevice::mem_in_use)
      .def("stream", [](HipTensorDevice& d) { return reinterpret_cast<size_t>(crabml_hip_device_stream(d.raw())); })
      .def("raw_handle", [](HipTensorDevice& d) { return reinterpret_cast<size_t>(d.raw()); })
      .def("prof_enable", [](HipTensorDevice& d, bool on) { d.check(crabml_hip_prof_enable(d.raw(), on ? 1 : 0)); })
      .def("read_ceiling_gbps",
           [](HipTensorDevice& d, size_t bytes, int reps) {
             double v = 0.0;
             py::gil_scoped_release rel;
             d.check(crabml_hip_debug_read_ceiling(d.raw(), bytes, reps, &v));
             return v;
           },
           py::arg("bytes") = (size_t)1 << 30, py::arg("reps") = 5)
      .def("debug_flash_attention",
           [](HipTensorDevice& d, py::array_t<float, py::array::c_style | py::array::forcecast> q,
              py::array_t<uint16_t, py::array::c_style | py::array::forcecast> k,
              py::array_t<uint16_t, py::array::c_style | py::array::forcecast> v, size_t n_heads, size_t n_kv, size_t head_dim,
              size_t seq, size_t slices) {
             if ((size_t)q.size() != n_heads * head_dim || (size_t)k.size() != n_kv * seq * head_dim || k.size() != v.size())
               throw Error(ErrorKind::BadInput, "debug_flash_attention: array sizes");
             py::array_t<float> out(n_heads * head_dim), out2(n_heads * head_dim);
             d.check(crabml_hip_debug_flash_attention(d.raw(), q.data(), k.data(), v.data(), n_heads, n_kv, head_dim, seq, slices,
                                                      out.mutable_data(), out2.mutable_data()));
             return py::make_tuple(out, out2);
           })
      .def("debug_flash_attention_rows",
           [](HipTensorDevice& d, py::array_t<float, py::array::c_style | py::array::forcecast> q,
              py::array_t<uint16_t, py::array::c_style | py::array::forcecast> k,
              py::array_t<uint16_t, py::array::c_style | py::array::forcecast> v, size_t n_heads, size_t n_kv, size_t head_dim,
              size_t pos0, size_t rows) {
             if (n_kv == 0 || head_dim == 0 || (size_t)q.size() != rows * n_heads * head_dim || (size_t)k.size() % (n_kv * head_dim) != 0 ||
                 k.size() != v.size())
               throw Error(ErrorKind::BadInput, "debug_flash_attention_rows: array sizes");
             const size_t seq_cap = (size_t)k.size() / (n_kv * head_dim);  // k, v: [n_kv][seq_cap][head_dim]
             py::array_t<float> out(rows * n_heads * head_dim);
             d.check(crabml_hip_debug_flash_attention_rows(d.raw(), q.data(), k.data(), v.data(), n_heads, n_kv, head_dim, pos0, rows, seq_cap,
                                                           out.mutable_data()));
             return out;
           })
      .def("prof_read_launches",
           [](HipTensorDevice& d, size_t cap) {
             std::vector<float> ms(cap);
             size_t n = 0;
             d.check(crabml_hip_prof_read_launches(d.raw(), ms.data(), cap, &n));
             return py::array_t<float>(n, ms.data());
           },
           py::arg("cap") = 65536)
      .def("prof_read",
           [](HipTensorDevice& d) {
             crabml_hip_prof_entry_t e[16];
             size_t n = 0;
             d.check(crabml_hip_prof_read(d.raw(), e, 16, &n));
             py::list out;
             for (size_t i = 0; i < n; i++) {
               py::dict r;
               r["dtype"] = e[i].dtype;
               r["stage"] = e[i].reserved;
               r["launches"] = e[i].launches;
               r["kernel_ms"] = e[i].kernel_ms;
               r["algo_bytes"] = e[i].algo_bytes;
               out.append(r);
             }
             return out;
           })
      .def("dump_debug_tensor", [](HipTensorDevice& d, const std::string& name) -> py::object {
        std::vector<float> v;
        if (!d.dump_debug_tensor(name, &v)) return py::none();
        return py::array_t<float>(v.size(), v.data());
      });

  py::class_<HipTensor>(m, "HipTensor")
      .def_static("from_cpu",
                  [](py::buffer buf, const py::sequence& shape, GGMLType dtype, std::shared_ptr<HipTensorDevice> dev) {
                    py::buffer_info info = buf.request();
                    size_t nbytes = (size_t)info.size * (size_t)info.itemsize;
                    return HipTensor::from_cpu(info.ptr, nbytes, to_shape(shape), dtype, std::move(dev));
                  })
      .def_static("new",
                  [](py::array_t<float, py::array::c_style | py::array::forcecast> a, const py::sequence& shape,
                     std::shared_ptr<HipTensorDevice> dev) {
                    std::vector<float> v(a.data(), a.data() + a.size());
                    return HipTensor::from_f32(v, to_shape(shape), std::move(dev));
                  })
      .def_static("alloc", [](const py::sequence& shape, GGMLType dtype, std::shared_ptr<HipTensorDevice> dev) {
        return HipTensor::alloc(to_shape(shape), dtype, std::move(dev));
      })
      .def("dtype", &HipTensor::dtype)
      .def("shape", [](const HipTensor& t) { return t.shape(); })
      .def("strider", [](const HipTensor& t) { return t.strider(); })
      .def("is_contiguous", &HipTensor::is_contiguous)
      .def("buf_len", &HipTensor::buf_len)
      .def("name", &HipTensor::name)
      .def("resize", &HipTensor::resize)
      .def("reshape", [](const HipTensor& t, const py::sequence& s) { return t.reshape(to_shape(s)); })
      .def("transpose", [](const HipTensor& t, const py::sequence& s) { return t.transpose(to_shape(s)); })
      .def("with_strider", &HipTensor::with_strider)
      .def("with_name", &HipTensor::with_name)
      .def("contiguous", &HipTensor::contiguous)
      .def("concatenate", &HipTensor::concatenate)
      .def("copy_rows_from", [](HipTensor& t, const HipTensor& src, const py::sequence& rows) { t.copy_rows_from(src, to_shape(rows)); })
      .def("export",
           [](const HipTensor& t) {
             std::vector<float> v = t.export_();
             return py::array_t<float>(v.size(), v.data());
           })
      .def("export_raw",
           [](const HipTensor& t) {
             std::vector<uint8_t> v = t.export_raw();
             return py::array_t<uint8_t>(v.size(), v.data());
           })
      .def("to_vec",  // test helper like CpuTensor::to_vec (cpu_tensor.rs:100-109): gathers through the strider
           [](const HipTensor& t) {
             HipTensor c = t.contiguous();
             std::vector<float> v = c.export_();
             return py::array_t<float>(v.size(), v.data());
           })
      .def("dup", &HipTensor::dup)
      .def("rope_inplace", &HipTensor::rope_inplace)
      .def("rms_norm_inplace", &HipTensor::rms_norm_inplace)
      .def("softmax_inplace", &HipTensor::softmax_inplace)
      .def("silu_inplace", &HipTensor::silu_inplace)
      .def("gelu_inplace", &HipTensor::gelu_inplace)
      .def("mul_inplace", &HipTensor::mul_inplace)
      .def("add_inplace", &HipTensor::add_inplace)
      .def("scale_inplace", &HipTensor::scale_inplace)
      .def("matmul_vec", &HipTensor::matmul_vec)
      .def("batch_matmul", &HipTensor::batch_matmul)
      .def("debug_quantize",
           [](const HipTensor& t, GGMLType q) {
             std::vector<uint8_t> v = t.debug_quantize(q);
             return py::array_t<uint8_t>(v.size(), v.data());
           })
      .def("debug_superblock_ints",
           [](const HipTensor& w, size_t row, const HipTensor& x, int variant) {
             auto r = w.debug_superblock_ints(row, x, variant);
             return py::make_tuple(py::array_t<int32_t>(r.first.size(), r.first.data()), r.second);
           },
           py::arg("row"), py::arg("x"), py::arg("variant") = 0)
      .def("debug_gemm_ints",
           [](const HipTensor& w, const HipTensor& x, size_t b) {
             auto r = w.debug_gemm_ints(x, b);
             return py::make_tuple(py::array_t<int32_t>(r.first.size(), r.first.data()), py::array_t<float>(r.second.size(), r.second.data()));
           })
      .def("debug_block_dots", [](const HipTensor& w, size_t row, const HipTensor& x) {
        std::vector<int32_t> v = w.debug_block_dots(row, x);
        return py::array_t<int32_t>(v.size(), v.data());
      });

  py::class_<LlamaConfig>(m, "LlamaConfig")
      .def(py::init([](size_t embedding_dim, size_t hidden_dim, size_t n_layers, size_t n_heads, size_t n_kv_heads,
                       size_t vocab_size, size_t seq_len, float rms_norm_eps, py::object rope_dim) {
             LlamaConfig c;
             c.embedding_dim = embedding_dim;
             c.hidden_dim = hidden_dim;
             c.n_layers = n_layers;
             c.n_heads = n_heads;
             c.n_kv_heads = n_kv_heads;
             c.vocab_size = vocab_size;
             c.seq_len = seq_len;
             c.rms_norm_eps = rms_norm_eps;
             if (!rope_dim.is_none()) c.rope_dim = rope_dim.cast<size_t>();
             return c;
           }),
           py::arg("embedding_dim"), py::arg("hidden_dim"), py::arg("n_layers"), py::arg("n_heads"),
           py::arg("n_kv_heads"), py::arg("vocab_size"), py::arg("seq_len"), py::arg("rms_norm_eps") = 1e-5f,
           py::arg("rope_dim") = py::none())
      .def_readonly("embedding_dim", &LlamaConfig::embedding_dim)
      .def_readonly("hidden_dim", &LlamaConfig::hidden_dim)
      .def_readonly("n_layers", &LlamaConfig::n_layers)
      .def_readonly("n_heads", &LlamaConfig::n_heads)
      .def_readonly("n_kv_heads", &LlamaConfig::n_kv_heads)
      .def_readonly("vocab_size", &LlamaConfig::vocab_size)
      .def_readonly("seq_len", &LlamaConfig::seq_len)
      .def_readonly("rms_norm_eps", &LlamaConfig::rms_norm_eps)
      .def_property_readonly("rope_dim", [](const LlamaConfig& c) -> py::object {
        return c.rope_dim ? py::object(py::int_(*c.rope_dim)) : py::object(py::none());
      })
      .def("head_size", &LlamaConfig::head_size)
      .def("kv_dim", &LlamaConfig::kv_dim);

  py::class_<Weights, std::shared_ptr<Weights>>(m, "LlamaWeights")
      .def(py::init([]() { return std::make_shared<Weights>(); }))
      .def_readwrite("token_embed", &Weights::token_embed)
      .def_readwrite("rms_att_weight", &Weights::rms_att_weight)
      .def_readwrite("rms_ffn_weight", &Weights::rms_ffn_weight)
      .def_readwrite("wq", &Weights::wq)
      .def_readwrite("wk", &Weights::wk)
      .def_readwrite("wv", &Weights::wv)
      .def_readwrite("wo", &Weights::wo)
      .def_readwrite("ffn_gate_weight", &Weights::ffn_gate_weight)
      .def_readwrite("ffn_down_weight", &Weights::ffn_down_weight)
      .def_readwrite("ffn_up_weight", &Weights::ffn_up_weight)
      .def_readwrite("rms_final_weight", &Weights::rms_final_weight)
      .def_property(
          "output_weight", [](Weights& w) -> py::object { return w.output_weight ? py::cast(*w.output_weight) : py::none(); },
          [](Weights& w, py::object o) {
            if (o.is_none())
              w.output_weight.reset();
            else
              w.output_weight = o.cast<HipTensor>();
          });


  // ---- GGUF (gguf.hpp: crabml-core/src/gguf.rs + the llama loader of crabml-llama2/src/model.rs) ---------------
  struct PyValue {
    static py::object conv(const GGUFValue& x) {
      switch (x.type) {
        case GGUFValueType::U8: case GGUFValueType::U16: case GGUFValueType::U32: case GGUFValueType::U64:
          return py::int_(std::get<uint64_t>(x.v));
        case GGUFValueType::Bool: return py::bool_(std::get<uint64_t>(x.v) != 0);
        case GGUFValueType::I8: case GGUFValueType::I16: case GGUFValueType::I32: case GGUFValueType::I64:
          return py::int_(std::get<int64_t>(x.v));
        case GGUFValueType::F32: case GGUFValueType::F64: return py::float_(std::get<double>(x.v));
        case GGUFValueType::String: return py::str(std::get<std::string>(x.v));
        case GGUFValueType::Array: {
          py::list l;
          for (const auto& it : std::get<GGUFArray>(x.v).items) l.append(conv(it));
          return l;
        }
      }
      return py::none();
    }
  };
  py::class_<GGUFFile, std::shared_ptr<GGUFFile>>(m, "GGUFFile")
      .def(py::init([](const std::string& path, bool mlock, int data_start) { return std::make_shared<GGUFFile>(path, mlock, data_start); }),
           py::arg("path"), py::arg("mlock") = false, py::arg("data_start") = -1)
      .def_property_readonly("data_start_convention", &GGUFFile::data_start_convention)
      .def_property_readonly("version", &GGUFFile::version)
      .def_property_readonly("architecture", &GGUFFile::architecture)
      .def_property_readonly("alignment", &GGUFFile::alignment)
      .def_property_readonly("tensor_data_offset", &GGUFFile::tensor_data_offset)
      .def("metadata",
           [](const GGUFFile& g) {
             py::dict d;
             for (const auto& kv : g.metadata()) d[py::str(kv.first)] = PyValue::conv(kv.second);
             return d;
           })
      .def("metadata_type",
           [](const GGUFFile& g, const std::string& key) -> py::object {
             auto it = g.metadata().find(key);
             if (it == g.metadata().end()) return py::none();
             return py::int_((uint32_t)it->second.type);
           })
      .def("tensor_infos",
           [](const GGUFFile& g) {
             py::list l;
             for (const auto& t : g.tensor_infos())
               l.append(py::make_tuple(t.name, t.dimensions, t.ggml_type, t.offset, t.data_len));
             return l;
           })
      .def("tensor_data",
           [](const GGUFFile& g, const std::string& name) {
             const GGUFTensorInfo* t = g.get_tensor_info(name);
             if (!t) throw Error(ErrorKind::TensorNotFound, "failed to find tensor " + name);
             return py::bytes((const char*)t->data, t->data_len);
           })
      .def("load_tensor", [](const GGUFFile& g, const std::string& name, std::shared_ptr<HipTensorDevice> dev) {
        return load_gguf_tensor(g, name, dev);  // model.rs:462-495 for one tensor
      })
      .def("load_config", [](const GGUFFile& g) { return load_llama_config(g); })
      .def("load_weights", [](const GGUFFile& g, const LlamaConfig& conf, std::shared_ptr<HipTensorDevice> dev) {
        py::gil_scoped_release rel;
        return load_llama_weights(g, conf, dev);
      });

  py::class_<TpComm, std::shared_ptr<TpComm>>(m, "TpComm")
      .def_static("unique_id",
                  []() {
                    std::vector<uint8_t> id = TpComm::unique_id();
                    return py::bytes((const char*)id.data(), id.size());
                  })
      .def(py::init([](std::shared_ptr<HipTensorDevice> dev, py::bytes id, int nranks, int rank) {
             std::string s = id;
             py::gil_scoped_release rel;  // ncclCommInitRank blocks until every rank has joined
             return std::make_shared<TpComm>(std::move(dev), std::vector<uint8_t>(s.begin(), s.end()), nranks, rank);
           }),
           py::arg("device"), py::arg("unique_id"), py::arg("nranks"), py::arg("rank"))
      .def_static("p2p",
                  [](std::shared_ptr<HipTensorDevice> dev, int nranks, int rank, size_t max_elems) {
                    return std::make_shared<TpComm>(TpComm::P2P{}, std::move(dev), nranks, rank, max_elems);
                  },
                  py::arg("device"), py::arg("nranks"), py::arg("rank"), py::arg("max_elems"))
      .def("export_handle",
           [](const TpComm& c) {
             std::vector<uint8_t> h = c.export_handle();
             return py::bytes((const char*)h.data(), h.size());
           })
      .def("connect",
           [](TpComm& c, py::bytes handles) {
             std::string s = handles;
             c.connect(std::vector<uint8_t>(s.begin(), s.end()));
           })
      .def_static("connect_local", &TpComm::connect_local)
      .def_property_readonly("nranks", &TpComm::nranks)
      .def_property_readonly("rank", &TpComm::rank)
      .def("all_reduce", &TpComm::all_reduce, py::call_guard<py::gil_scoped_release>());

  py::class_<HipLlamaRunner>(m, "HipLlamaRunner")
      .def(py::init([](const LlamaConfig& conf, std::shared_ptr<Weights> w, std::shared_ptr<HipTensorDevice> dev,
                       size_t seq_len, bool use_f16_kv_cache, bool use_graph, bool prefetch, int tp_size, int tp_rank,
                       std::shared_ptr<TpComm> comm, bool norm_epilogue, int extra_flags, size_t attn_long_from,
                       size_t prefill_chunk) {
             auto* r = new HipLlamaRunner(conf, std::move(w), std::move(dev), seq_len, use_f16_kv_cache, use_graph, prefetch,
                                          tp_size, tp_rank, std::move(comm), norm_epilogue, extra_flags, attn_long_from,
                                          prefill_chunk);
             r->set_seq_cap(seq_len);
             return r;
           }),
           py::arg("conf"), py::arg("weights"), py::arg("device"), py::arg("seq_len"), py::arg("use_f16_kv_cache"),
           py::arg("use_graph") = true, py::arg("prefetch") = true, py::arg("tp_size") = 1, py::arg("tp_rank") = 0,
           py::arg("comm") = std::shared_ptr<TpComm>(), py::arg("norm_epilogue") = true,
           py::arg("extra_flags") = 0, py::arg("attn_long_from") = 0, py::arg("prefill_chunk") = 0)
      .def("prefill",
           [](HipLlamaRunner& r, const std::vector<uint32_t>& tokens) {
             std::vector<float> lg;
             {
               py::gil_scoped_release rel;
               lg = r.prefill(tokens);
             }
             return py::array_t<float>(lg.size(), lg.data());
           })
      .def_static("tp_sim_forward",
                  [](const std::vector<HipLlamaRunner*>& ranks, size_t token, size_t pos) {
                    std::vector<float> lg;
                    {
                      py::gil_scoped_release rel;
                      lg = HipLlamaRunner::tp_sim_forward(ranks, token, pos);
                    }
                    return py::array_t<float>(lg.size(), lg.data());
                  })
      .def("kv_cache_len", &HipLlamaRunner::kv_cache_len)
      .def("reset", &HipLlamaRunner::reset)
      .def("forward",
           [](HipLlamaRunner& r, size_t token, size_t pos) {
             std::vector<float> lg;
             {
               py::gil_scoped_release rel;
               lg = r.forward(token, pos);
             }
             return py::array_t<float>(lg.size(), lg.data());
           })
      .def("forward_async", &HipLlamaRunner::forward_async)
      .def("decode_greedy",
           [](HipLlamaRunner& r, size_t token, size_t steps) {
             py::gil_scoped_release rel;
             return r.decode_greedy(token, steps);
           })
      .def("debug_kv", [](HipLlamaRunner& r, size_t layer, bool v, bool f16) {
        std::vector<uint8_t> b = r.debug_kv(layer, v, f16);
        return py::array_t<uint8_t>(b.size(), b.data());
      });

  py::class_<Runner>(m, "Llama2Runner")
      .def(py::init([](const LlamaConfig& conf, std::shared_ptr<Weights> w, std::shared_ptr<HipTensorDevice> dev,
                       size_t seq_len, bool use_f16_kv_cache) {
             return new Runner(conf, std::move(w), std::move(dev), seq_len, use_f16_kv_cache, GGMLType::F32, GGMLType::F16);
           }),
           py::arg("conf"), py::arg("weights"), py::arg("device"), py::arg("seq_len"), py::arg("use_f16_kv_cache"))
      .def("kv_cache_len", &Runner::kv_cache_len)
      .def("forward",
           [](Runner& r, const std::vector<size_t>& tokens, size_t pos) {
             {
               py::gil_scoped_release rel;
               r.forward(tokens, pos);
             }
             return py::array_t<float>(r.logits().size(), r.logits().data());
           })
      .def("logits", [](Runner& r) { return py::array_t<float>(r.logits().size(), r.logits().data()); })
      .def("generate_greedy",
           [](Runner& r, const std::vector<size_t>& prompt, size_t steps) {
             py::gil_scoped_release rel;
             return r.generate_greedy(prompt, steps);
           })
      // decode `steps` tokens starting from `token` at the current kv length; returns (ids, seconds)
      .def("timed_decode", [](Runner& r, size_t token, size_t steps) {
        py::gil_scoped_release rel;
        std::vector<size_t> ids;
        size_t pos = r.kv_cache_len();
        double sample_sec = 0.0;
        auto t0 = std::chrono::steady_clock::now();
        for (size_t s = 0; s < steps; s++) {
          r.forward({token}, pos + s);
          auto ta = std::chrono::steady_clock::now();
          token = sample_argmax(r.logits());
          sample_sec += std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count();
          ids.push_back(token);
        }
        double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return std::make_tuple(ids, sec, sample_sec);  // (ids, seconds, of which in the host arg-max)
      });
}
