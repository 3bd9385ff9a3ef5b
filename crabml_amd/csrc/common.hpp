// common.hpp -- internal types of libcrabml_hip.so (device, buffers, pool, error plumbing).
// Product code: nothing here may include or link anything under oracle/.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/crabml_hip.h"
#include "../../include/crabml_hip_debug.h"  // parity / measurement hooks and A/B flag bits (not the drop-in surface)

namespace crabml_hip {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- GGML block geometry (crabml-core/src/cpu/buf/buf_q*.rs) ---------------------------------
inline size_t block_elems(uint32_t t) {
  switch (t) {
    case CRABML_HIP_F32: case CRABML_HIP_F16: return 1;
    case CRABML_HIP_Q4_0: case CRABML_HIP_Q4_1: case CRABML_HIP_Q5_0: case CRABML_HIP_Q5_1: case CRABML_HIP_Q8_0: case CRABML_HIP_Q8_1: return 32;
    case CRABML_HIP_Q2_K: case CRABML_HIP_Q3_K: case CRABML_HIP_Q4_K: case CRABML_HIP_Q5_K: case CRABML_HIP_Q6_K: case CRABML_HIP_Q8_K: return 256;
    default: return 0;
  }
}
inline size_t block_bytes(uint32_t t) {
  switch (t) {
    case CRABML_HIP_F32: return 4;
    case CRABML_HIP_F16: return 2;
    case CRABML_HIP_Q4_0: return 18;
    case CRABML_HIP_Q4_1: return 20;
    case CRABML_HIP_Q5_0: return 22;
    case CRABML_HIP_Q5_1: return 24;
    case CRABML_HIP_Q2_K: return 84;
    case CRABML_HIP_Q3_K: return 110;
    case CRABML_HIP_Q8_0: return 34;
    case CRABML_HIP_Q8_1: return 36;
    case CRABML_HIP_Q4_K: return 144;
    case CRABML_HIP_Q5_K: return 176;
    case CRABML_HIP_Q6_K: return 210;
    case CRABML_HIP_Q8_K: return 292;
    default: return 0;
  }
}
// CpuTensorBuf::vec_dot_rhs_dtype, crabml-core/src/cpu/buf/api.rs:142-159
inline uint32_t vec_dot_rhs_dtype(uint32_t t) {
  switch (t) {
    case CRABML_HIP_F32: return CRABML_HIP_F32;
    case CRABML_HIP_F16: return CRABML_HIP_F16;
    case CRABML_HIP_Q8_0: case CRABML_HIP_Q4_0: case CRABML_HIP_Q5_0: return CRABML_HIP_Q8_0;
    case CRABML_HIP_Q8_1: case CRABML_HIP_Q4_1: case CRABML_HIP_Q5_1: return CRABML_HIP_Q8_1;
    case CRABML_HIP_Q8_K: case CRABML_HIP_Q2_K: case CRABML_HIP_Q3_K: case CRABML_HIP_Q4_K: case CRABML_HIP_Q5_K: case CRABML_HIP_Q6_K:
      return CRABML_HIP_Q8_K;
    default: return 0xffffffffu;
  }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- device-resident layouts -----------------------------------------------------------------
// Quantized WEIGHTS are re-laid-out once at upload ("planes"): all quants of the tensor first
// (16-byte aligned, block-major, same block order as GGUF), then the per-block scales.  Measured
// on MI355X (profiles/r01_gemv_lab_layout_sweep.log): 6.9 TB/s vs 3.5 TB/s for the raw 18-byte
// AoS blocks, because a wave's 64 lanes then issue one aligned 1 KiB dwordx4 request.
//   Q4_0: qs[n][16]            | d[n] f16
//   Q8_0: qs[n][32]            | d[n] f16
//   Q4_1: qs[n][16]            | dm[n] (d f16, m f16)
//   Q4_K: qs[n][128]           | hdr[n][16]: d f16, dmin f16, then the 8 (scale, min) 6-bit pairs of scales[12]
//                                (qs CLASS-MAJOR since round 6: inside every 32-byte chunk -- the low / high nibbles of one 64-element
//                                pair -- byte 4 l + k holds what the file's byte 8 k + l held, l = 0..7, k = 0..3, so that dword l
//                                carries the four elements e of the 32-group with e % 8 == l: the reference keeps eight f32 lanes per
//                                row fed by exactly those classes (buf_q4_k.rs:243-263), and one v_dot4 against the equally permuted
//                                activation plane (`qp` below) is a whole class sum; q4k_perm_index)
//                                re-packed pair-major -- 24 bits per 64-element pair p, little endian at bit 24 p:
//                                scale[2p] | scale[2p+1] << 6 | min[2p] << 12 | min[2p+1] << 18 (get_scale_min_k4,
//                                util.rs:19-27, is a bit permutation of the same 96 bits; done once at upload) --
//                                so a lane extracts its pair's four fields with 7 VALU ops instead of 23
//   Q5_K: qs[n][128]           | qh[n][32] | hdr[n][16] = Q4_K's header (d, dmin, pair-major 6-bit fields)   (off_scale = n * 128:
//                                the qh plane; the header plane follows at off_scale + n * 32)
//   Q6_K: ql[n][128]           | qh[n][64] | scales[n][16] | d[n] f16   (off_scale = n * 128 exactly, so the
//                                kernels derive the other three plane offsets from it)
//   Q8_K: qs[n][256]           | d[n] f32            (bsums are derived data; not kept for weights)
//   Q5_0: qs[n][16]            | qh[n] u32 | d[n] f16                      (off_scale = n * 16: the qh plane)
//   Q5_1: qs[n][16]            | (d f16, m f16, qh u32)[n]                 (off_scale = n * 16)
//   Q2_K: qs[n][64]            | scales[n][16] | (d f16, dmin f16)[n]      (off_scale = n * 64)
//   Q3_K: qs[n][64]            | hmask[n][32] | (scales[12], d f16, 2 unused bytes)[n]   (off_scale = n * 64)
// Quantized ACTIVATIONS (the rhs of matmul_vec) live in a per-buffer scratch, also as planes:
//   Q8_0: qs[n] i8 | d[n/32] f16 | isum[n/32] i32 (sum of the 32 quants; exact, derived)
//   Q8_1: qs[n] i8 | d[n/32] f16 | s[n/32] f16
//   Q8_K: qs[n] i8 | d[n/256] f32 | bsums[n/16] i16 | qp[n] i8 -- the same quants once more, class-major inside every 32-element
//         group (byte 4 l + k = element 8 k + l): the plane the Q4_K kernels read; every other K-quant reads qs
//   F16 : h[n] f16 (buf/api.rs:198)
struct WeightLayout {
  size_t n_blocks = 0;
  size_t off_scale = 0;  // byte offset of the scale plane
  size_t total = 0;      // bytes
};
WeightLayout weight_layout(uint32_t dtype, size_t n_elems);

struct ActLayout {
  size_t off_d = 0, off_aux = 0, total = 0;
  size_t off_p = 0;  // Q8_K: the class-major copy of the quants (read by the Q4_K kernels)
};
// position of element e (0..31) of a 32-element group inside its class-major 32 bytes, and back (an involution it is not:
// element 8 k + l sits at byte 4 l + k)
__host__ __device__ inline int q4k_perm_index(int e) { return 4 * (e & 7) + (e >> 3); }
ActLayout act_layout(uint32_t qtype, size_t n_elems);

}  // namespace crabml_hip

namespace crabml_hip {
struct LazyState;  // lazy.hpp: the recorded-op queue + the fused-step matcher of this device
}

// ---- the opaque C types ------------------------------------------------------------------------
struct crabml_hip_device {
  int ordinal = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int n_cu = 256;
  int numa_node = -1;         // the host NUMA node the GPU hangs off (sysfs), -1 = unknown
  bool strict_order = false;  // CRABML_HIP_FLAG_STRICT_ORDER
  std::mutex mu;
  std::string last_error;
  // caching allocator: size class -> free blocks (single stream => stream-ordered reuse is safe)
  std::map<size_t, std::vector<void*>> pool;
  size_t bytes_reserved = 0;
  // f16 lookup tables (cpu_device.rs:108-125)
  uint16_t* exp_table = nullptr;
  uint16_t* gelu_table = nullptr;
  // measurement hook (crabml_hip_prof_*): event pairs around GEMV launches
  bool prof_on = false;
  struct ProfRec {
    hipEvent_t e0, e1;
    uint32_t dtype;
    uint32_t stage;  // 0 = matmul_vec (per-op path); fused stages: 1 qkv, 2 wo+res, 3 gate/up, 4 down+res, 5 classifier
    double bytes;
  };
  std::vector<ProfRec> prof_recs;
  std::vector<hipEvent_t> prof_free_events;
  // Tensor ops are RECORDED and run at the next point the host can observe data (export / sync): lazy.hip.  `lazy` off
  // (CRABML_HIP_FLAG_PER_OP): every call launches immediately.  `fuse` off: the queue is replayed op by op, never matched.
  bool lazy = true, fuse = true;
  bool destroying = false;  // crabml_hip_device_destroy's flush: run what is queued, learn nothing
  bool dry = false;  // test hook (CRABML_HIP_FLAG_DRY + CRABML_HIP_TEST_HOOKS=1): no HIP device behind this object -- ops are
                     // recorded, matched and counted, nothing is computed (the CPU suite drives the recorder / matcher with it)
  crabml_hip::LazyState* lz = nullptr;
};

namespace crabml_hip {
inline std::atomic<uint64_t> g_buf_uid{0};
}
struct crabml_hip_buf {
  crabml_hip_device* dev = nullptr;
  std::atomic<int> refcnt{1};
  uint32_t dtype = 0;
  size_t n_elems = 0;
  void* ptr = nullptr;  // device memory; bound on first use (ensure_mem) for buffers created by recorded ops / alloc
  size_t bytes = 0;     // size to bind
  bool zero_init = false;  // Tensor::alloc(F32): zero-filled when the memory is bound (vec![0.0; n], cpu_tensor.rs:146-149)
  uint8_t deferred = 0;    // lazy.hip: the value still lives in the fused context (1 = the final RMSNorm of its residual stream)
  uint64_t uid = ++crabml_hip::g_buf_uid;  // unique per handle for the life of the process (an address is not: the allocator re-uses them)
  size_t cap = 0;       // pool capacity in bytes
  size_t m = 0, k = 0;  // logical 2-D shape of quantized weights
  crabml_hip::WeightLayout wl;
  // activation-quantization cache: matmul_vec re-quantizes its rhs on every call in the reference
  // (matmul_vec.rs:37-40); here q/k/v and gate/up share one pass.  `version` is bumped by every
  // op that writes the buffer, which invalidates the cache.
  uint64_t version = 1;
  struct {
    uint32_t qtype = 0xffffffffu;
    uint64_t version = 0;
    size_t n = 0;
    size_t k = 0;  // row length the planes were laid out for (act_layout depends on k, not only on b * k)
    void* ptr = nullptr;
    size_t cap = 0;
  } qc;
};

namespace crabml_hip {

int set_error(crabml_hip_device* dev, int status, const char* fmt, ...);
int hip_fail(crabml_hip_device* dev, hipError_t e, const char* what, const char* file, int line);

#define CH_HIP(dev, expr)                                                                        \
  do {                                                                                           \
    hipError_t e__ = (expr);                                                                     \
    if (e__ != hipSuccess) return ::crabml_hip::hip_fail((dev), e__, #expr, __FILE__, __LINE__); \
  } while (0)

// Every entry point that allocates or launches selects its device first: hipMalloc / kernel launches / hipGraphLaunch go
// to the CALLING THREAD's current device, which is device 0 on a fresh thread and whatever torch (or a second
// HipTensorDevice) left behind otherwise.  hipSetDevice on the already-current device is a thread-local compare.
#define CH_USE(dev)                                                                                        \
  do {                                                                                                     \
    hipError_t e__ = hipSetDevice((dev)->ordinal);                                                         \
    if (e__ != hipSuccess) return ::crabml_hip::hip_fail((dev), e__, "hipSetDevice", __FILE__, __LINE__); \
  } while (0)

#define CH_BAIL(dev, status, ...) return ::crabml_hip::set_error((dev), (status), __VA_ARGS__)

#define CH_TRY(expr)            \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ != 0) return rc__; \
  } while (0)

int pool_alloc(crabml_hip_device* dev, size_t bytes, void** out, size_t* cap);
void pool_free(crabml_hip_device* dev, void* ptr, size_t cap);
int buf_new(crabml_hip_device* dev, uint32_t dtype, size_t n_elems, size_t bytes, crabml_hip_buf** out);
// a handle without memory: bound by ensure_mem when an executed op first touches it (outputs of recorded ops that a fused
// launch makes redundant never get any)
crabml_hip_buf* buf_new_unbound(crabml_hip_device* dev, uint32_t dtype, size_t n_elems, size_t bytes);
int ensure_mem(crabml_hip_device* dev, crabml_hip_buf* b);
// runs every recorded op (lazy.hip); every entry point that touches the stream or reads device memory calls it first
int lazy_flush(crabml_hip_device* dev);
#define CH_FLUSH(dev) CH_TRY(::crabml_hip::lazy_flush(dev))
// entry points that compute or measure have nothing to offer on the record-only test device
#define CH_LIVE(dev)                                                                                                      \
  do {                                                                                                                    \
    if ((dev)->dry) return ::crabml_hip::set_error((dev), CRABML_HIP_NOT_IMPLEMENTED, "record-only test device: no HIP device"); \
  } while (0)
inline void touch(crabml_hip_buf* b) { b->version++; }
// measurement hook helpers (runtime.hip)
// prof_begin hands out an event pair; the kernel is then launched with hipExtLaunchKernelGGL(start, stop), which
// stamps the events with the dispatch packet's own begin/end timestamps (the same clock rocprofv3 reads), and
// prof_end files the record.
int prof_begin(crabml_hip_device* dev, crabml_hip_device::ProfRec* rec, uint32_t dtype, uint32_t stage, double bytes);
int prof_end(crabml_hip_device* dev, crabml_hip_device::ProfRec* rec);

}  // namespace crabml_hip
