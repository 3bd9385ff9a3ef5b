// runtime.hip -- device / buffer management and the C ABI glue of libcrabml_hip.so.
//
// What the reference does with `Arc<CpuTensorDevice>` + `Cow<[f32]>` (crabml-core/src/cpu/cpu_device.rs,
// cpu_tensor.rs) and the wgpu backend does with `Arc<wgpu::Buffer>` + one queue (crabml-wgpu/src/
// wgpu_tensor.rs:20-28, wgpu_device.rs) is done here with: one HIP stream per device (all work is
// stream-ordered, the host only blocks in export), a size-class caching allocator over hipMalloc
// (activations are allocated and dropped ~30x per layer by the runner), and reference-counted buffers.
#include <chrono>
#include <cmath>

#include "kernels.hpp"
#include "lazy.hpp"

using namespace crabml_hip;

namespace crabml_hip {

// ---- errors ------------------------------------------------------------------------------------------
int set_error(crabml_hip_device* dev, int status, const char* fmt, ...) {
  char msg[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(msg, sizeof msg, fmt, ap);
  va_end(ap);
  if (dev) {
    std::lock_guard<std::mutex> g(dev->mu);
    dev->last_error = msg;
  }
  return status;
}
int hip_fail(crabml_hip_device* dev, hipError_t e, const char* what, const char* file, int line) {
  return set_error(dev, CRABML_HIP_UNEXPECTED, "HIP error %d (%s) in `%s` at %s:%d", (int)e, hipGetErrorString(e), what,
                   file, line);
}

// ---- layouts -----------------------------------------------------------------------------------------
WeightLayout weight_layout(uint32_t dtype, size_t n_elems) {
  WeightLayout wl;
  size_t be = block_elems(dtype);
  wl.n_blocks = be ? n_elems / be : 0;
  size_t n = wl.n_blocks;
  switch (dtype) {
    case CRABML_HIP_F32: wl.off_scale = 0; wl.total = n_elems * 4; break;
    case CRABML_HIP_F16: wl.off_scale = 0; wl.total = n_elems * 2; break;
    case CRABML_HIP_Q4_0: wl.off_scale = align_up(n * 16, 256); wl.total = wl.off_scale + n * 2; break;
    case CRABML_HIP_Q8_0: wl.off_scale = align_up(n * 32, 256); wl.total = wl.off_scale + n * 2; break;
    case CRABML_HIP_Q4_1: wl.off_scale = align_up(n * 16, 256); wl.total = wl.off_scale + n * 4; break;
    case CRABML_HIP_Q4_K: wl.off_scale = align_up(n * 128, 256); wl.total = wl.off_scale + n * 16; break;
    case CRABML_HIP_Q5_K: wl.off_scale = n * 128; wl.total = align_up(n * 176, 256); break;
    case CRABML_HIP_Q5_0: wl.off_scale = n * 16; wl.total = align_up(n * 22, 256); break;
    case CRABML_HIP_Q5_1: wl.off_scale = n * 16; wl.total = align_up(n * 24, 256); break;
    case CRABML_HIP_Q2_K: wl.off_scale = n * 64; wl.total = align_up(n * 84, 256); break;
    case CRABML_HIP_Q3_K: wl.off_scale = n * 64; wl.total = align_up(n * 112, 256); break;
    case CRABML_HIP_Q6_K: wl.off_scale = n * 128; wl.total = align_up(n * 210, 256); break;
    case CRABML_HIP_Q8_K: wl.off_scale = align_up(n * 256, 256); wl.total = wl.off_scale + n * 4; break;
    default: wl.total = 0;
  }
  return wl;
}

ActLayout act_layout(uint32_t qtype, size_t n) {
  ActLayout al;
  switch (qtype) {
    case CRABML_HIP_F32: al.total = align_up(n * 4, 256); break;
    case CRABML_HIP_F16: al.total = align_up(n * 2, 256); break;
    case CRABML_HIP_Q8_0:
      al.off_d = align_up(n, 256);
      al.off_aux = al.off_d + align_up(n / 32 * 2, 256);
      al.total = al.off_aux + align_up(n / 32 * 4, 256);
      break;
    case CRABML_HIP_Q8_1:
      al.off_d = align_up(n, 256);
      al.off_aux = al.off_d + align_up(n / 32 * 2, 256);
      al.total = al.off_aux + align_up(n / 32 * 2, 256);
      break;
    case CRABML_HIP_Q8_K:
      al.off_d = align_up(n, 256);
      al.off_aux = al.off_d + align_up(n / 256 * 4, 256);
      al.off_p = al.off_aux + align_up(n / 16 * 2, 256);
      al.total = al.off_p + align_up(n, 256);
      break;
    default: break;
  }
  return al;
}

// ---- pool --------------------------------------------------------------------------------------------
static size_t size_class(size_t bytes) {
  if (bytes < 256) bytes = 256;
  if (bytes <= ((size_t)1 << 20)) {
    size_t c = 256;
    while (c < bytes) c <<= 1;
    return c;
  }
  return align_up(bytes, (size_t)1 << 20);
}

int pool_alloc(crabml_hip_device* dev, size_t bytes, void** out, size_t* cap) {
  size_t cls = size_class(bytes);
  {
    std::lock_guard<std::mutex> g(dev->mu);
    auto it = dev->pool.find(cls);
    if (it != dev->pool.end() && !it->second.empty()) {
      *out = it->second.back();
      it->second.pop_back();
      *cap = cls;
      return 0;
    }
  }
  void* p = nullptr;
  if (dev->dry) {  // record-only test device: the pointer is never dereferenced
    p = malloc(cls);
    if (!p) return set_error(dev, CRABML_HIP_UNEXPECTED, "malloc(%zu) failed", cls);
    std::lock_guard<std::mutex> g(dev->mu);
    dev->bytes_reserved += cls;
    *out = p;
    *cap = cls;
    return 0;
  }
  (void)hipSetDevice(dev->ordinal);  // hipMalloc allocates on the calling thread's current device
  hipError_t e = hipMalloc(&p, cls);
  if (e != hipSuccess) {
    // give pooled blocks back to the driver once, then retry
    {
      std::lock_guard<std::mutex> g(dev->mu);
      for (auto& kv : dev->pool) {
        for (void* q : kv.second) {
          (void)hipFree(q);
          dev->bytes_reserved -= kv.first;
        }
        kv.second.clear();
      }
    }
    e = hipMalloc(&p, cls);
    if (e != hipSuccess && lazy_release_contexts(dev) > 0) {
      // ... and the decode contexts nothing is being served from (a parked runner's, the last model's between tokens): they retain
      // their weights and private buffers; the queue rebuilds what the host still uses
      std::lock_guard<std::mutex> g(dev->mu);
      for (auto& kv : dev->pool) {
        for (void* q : kv.second) {
          (void)hipFree(q);
          dev->bytes_reserved -= kv.first;
        }
        kv.second.clear();
      }
      e = hipMalloc(&p, cls);
    }
    if (e != hipSuccess) return hip_fail(dev, e, "hipMalloc", __FILE__, __LINE__);
  }
  {
    std::lock_guard<std::mutex> g(dev->mu);
    dev->bytes_reserved += cls;
  }
  *out = p;
  *cap = cls;
  return 0;
}

void pool_free(crabml_hip_device* dev, void* ptr, size_t cap) {
  if (!ptr) return;
  std::lock_guard<std::mutex> g(dev->mu);
  dev->pool[cap].push_back(ptr);
}

int buf_new(crabml_hip_device* dev, uint32_t dtype, size_t n_elems, size_t bytes, crabml_hip_buf** out) {
  crabml_hip_buf* b = new crabml_hip_buf();
  b->dev = dev;
  b->dtype = dtype;
  b->n_elems = n_elems;
  b->bytes = bytes;
  int rc = pool_alloc(dev, bytes, &b->ptr, &b->cap);
  if (rc != 0) {
    delete b;
    return rc;
  }
  *out = b;
  return 0;
}

int prof_begin(crabml_hip_device* dev, crabml_hip_device::ProfRec* rec, uint32_t dtype, uint32_t stage, double bytes) {
  auto get_ev = [&](hipEvent_t* e) -> hipError_t {
    if (!dev->prof_free_events.empty()) {
      *e = dev->prof_free_events.back();
      dev->prof_free_events.pop_back();
      return hipSuccess;
    }
    return hipEventCreate(e);
  };
  CH_HIP(dev, get_ev(&rec->e0));
  CH_HIP(dev, get_ev(&rec->e1));
  rec->dtype = dtype;
  rec->stage = stage;
  rec->bytes = bytes;
  return 0;
}
int prof_end(crabml_hip_device* dev, crabml_hip_device::ProfRec* rec) {
  dev->prof_recs.push_back(*rec);
  return 0;
}

// record the op (default) or run its launches right away (CRABML_HIP_FLAG_PER_OP)
static int submit(crabml_hip_device* dev, LazyOp& op) { return dev->lazy ? lazy_record(dev, op) : lazy_exec(dev, op); }

}  // namespace crabml_hip

// ===========================================================================================================
// C ABI
// ===========================================================================================================
extern "C" {

int crabml_hip_abi_version(void) { return CRABML_HIP_ABI_VERSION; }

int crabml_hip_device_create(const crabml_hip_device_options_t* opts, crabml_hip_device_t** out) {
  if (!out) return CRABML_HIP_BAD_INPUT;
  *out = nullptr;
  crabml_hip_device* dev = new crabml_hip_device();
  dev->ordinal = opts ? opts->device_ordinal : 0;
  dev->strict_order = opts && (opts->flags & CRABML_HIP_FLAG_STRICT_ORDER);
  dev->lazy = !(opts && (opts->flags & CRABML_HIP_FLAG_PER_OP));
  dev->fuse = dev->lazy && !(opts && (opts->flags & CRABML_HIP_FLAG_LAZY_NO_FUSION));
  dev->lz = new LazyState();
  const char* hooks = getenv("CRABML_HIP_TEST_HOOKS");
  const bool hooks_on = hooks != nullptr && hooks[0] == '1';
  if (opts && (opts->flags & CRABML_HIP_FLAG_DRY)) {
    // test hook (tests/test_lazy_queue.py): a device object with NO HIP device behind it -- Tensor calls are recorded, matched
    // against the decode template and counted; nothing is computed and export() hands out zeros.  Armed only together with
    // CRABML_HIP_TEST_HOOKS=1, and it says so: this is not a CPU fallback, it cannot produce a single logit.
    if (!hooks_on) {
      lazy_destroy(dev);
      delete dev;
      return CRABML_HIP_UNEXPECTED;
    }
    fprintf(stderr, "crabml_hip: TEST HOOK active: record-only device (no HIP device, nothing is computed)\n");
    dev->dry = true;
    dev->own_stream = false;
    *out = dev;
    return 0;
  }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  auto fail = [&]() {
    lazy_destroy(dev);
    delete dev;
    return CRABML_HIP_UNEXPECTED;  // no usable MI355X: the backend fails loudly, there is no CPU fallback
  };
  if (e != hipSuccess || n <= 0 || dev->ordinal >= n) return fail();
  if (hipSetDevice(dev->ordinal) != hipSuccess) return fail();
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev->ordinal) == hipSuccess) dev->n_cu = prop.multiProcessorCount;
  {  // which host NUMA node is the GPU attached to?  (a host thread that drives it token by token -- the reference's runner -- is a
     // few percent faster from that node: launches, the logits in pinned memory, the completion flag all cross the socket otherwise)
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, dev->ordinal) == hipSuccess) {
      for (char* p = bus; *p; p++)
        if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');
      char path[160];
      snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
      if (FILE* f = fopen(path, "r")) {
        int node = -1;
        if (fscanf(f, "%d", &node) == 1) dev->numa_node = node;
        fclose(f);
      }
    }
    (void)hipGetLastError();
  }
  // test hook (tests/test_hip_fault_paths.py): claim this many CUs whatever the device reports, so that a CU-masked process
  // (HSA_CU_MASK) loses the co-residency the in-launch hand-offs rely on -- their bounded polls must raise, not hang
  // Armed only when CRABML_HIP_TEST_HOOKS=1 is set as well (a stray CRABML_HIP_ASSUME_CUS alone is ignored), and it says so.
  if (hooks_on) {
    if (const char* e = getenv("CRABML_HIP_ASSUME_CUS")) {
      const int v = atoi(e);
      if (v > 0 && v <= 1024) {
        fprintf(stderr, "crabml_hip: TEST HOOK active: assuming %d CUs (device reports %d)\n", v, dev->n_cu);
        dev->n_cu = v;
      }
    }
  }
  if (opts && opts->stream) {
    dev->stream = (hipStream_t)opts->stream;
    dev->own_stream = false;
  } else {
    if (hipStreamCreateWithFlags(&dev->stream, hipStreamNonBlocking) != hipSuccess) return fail();
    dev->own_stream = true;
  }
  // exp table: f16 bits -> f16(exp(f32(x)))   (cpu_device.rs:108-115)
  std::vector<uint16_t> tab(65536);
  for (uint32_t x = 0; x < 65536; x++) tab[x] = host_f2h(expf(host_h2f((uint16_t)x)));
  if (hipMalloc((void**)&dev->exp_table, 65536 * 2) != hipSuccess ||
      hipMemcpy(dev->exp_table, tab.data(), 65536 * 2, hipMemcpyHostToDevice) != hipSuccess)
    return fail();
  *out = dev;
  return 0;
}

int crabml_hip_device_destroy(crabml_hip_device_t* dev) {
  if (!dev) return 0;
  if (!dev->dry) (void)hipSetDevice(dev->ordinal);
  dev->destroying = true;  // (the flush runs what is queued and learns nothing)
  (void)lazy_flush(dev);
  if (!dev->dry) (void)hipStreamSynchronize(dev->stream);
  lazy_destroy(dev);  // the decode context the queue built, its retained buffers
  for (auto& kv : dev->pool)
    for (void* p : kv.second) {
      if (dev->dry)
        free(p);
      else
        (void)hipFree(p);
    }
  dev->pool.clear();
  if (dev->exp_table) (void)hipFree(dev->exp_table);
  if (dev->gelu_table) (void)hipFree(dev->gelu_table);
  if (dev->own_stream) (void)hipStreamDestroy(dev->stream);
  delete dev;
  return 0;
}

int crabml_hip_device_sync(crabml_hip_device_t* dev) {
  if (!dev) return CRABML_HIP_BAD_INPUT;
  if (dev->dry) return lazy_flush(dev);
  CH_USE(dev);
  CH_FLUSH(dev);
  CH_TRY(lazy_fault_request(dev));
  CH_HIP(dev, hipStreamSynchronize(dev->stream));
  return lazy_fault_check(dev);
}

size_t crabml_hip_last_error(crabml_hip_device_t* dev, char* buf, size_t cap) {
  if (!dev) return 0;
  std::lock_guard<std::mutex> g(dev->mu);
  if (buf && cap) {
    size_t n = dev->last_error.size() < cap - 1 ? dev->last_error.size() : cap - 1;
    memcpy(buf, dev->last_error.data(), n);
    buf[n] = 0;
  }
  return dev->last_error.size();
}

void* crabml_hip_device_stream(crabml_hip_device_t* dev) { return dev ? (void*)dev->stream : nullptr; }

size_t crabml_hip_device_mem_in_use(crabml_hip_device_t* dev) {
  if (!dev) return 0;
  std::lock_guard<std::mutex> g(dev->mu);
  return dev->bytes_reserved;
}

// ---- buffers -----------------------------------------------------------------------------------------
int crabml_hip_buf_from_cpu(crabml_hip_device_t* dev, const void* bytes, size_t nbytes, const size_t* shape, int ndim,
                            uint32_t t, crabml_hip_buf_t** out) {
  if (!dev || !out || (!bytes && nbytes) || ndim < 1 || ndim > 4 || !shape) return CRABML_HIP_BAD_INPUT;
  *out = nullptr;
  size_t be = block_elems(t), bb = block_bytes(t);
  if (be == 0 || t == CRABML_HIP_Q8_1)
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "unsupported tensor type on hip %u", t);
  size_t n_elems = 1;
  for (int i = 0; i < ndim; i++)
    if (__builtin_mul_overflow(n_elems, shape[i], &n_elems))
      CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "shape overflows the element count");
  size_t k = shape[ndim - 1];
  size_t m = k ? n_elems / k : 0;
  if (be > 1 && k % be != 0)
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "last dim %zu is not a multiple of the block size %zu", k, be);
  size_t expect = n_elems / be * bb;
  if (nbytes < expect)  // GGUF slices may carry trailing alignment padding (gguf.rs:743-748): >= is accepted
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "data length %zu too small for shape (need %zu)", nbytes, expect);
  WeightLayout wl = weight_layout(t, n_elems);
  crabml_hip_buf* b = nullptr;
  if (dev->dry) {  // record-only device: the handle and its geometry, no bytes
    CH_TRY(buf_new(dev, t, n_elems, 256, &b));
    b->wl = wl;
    b->m = m;
    b->k = k;
    *out = b;
    return 0;
  }
  CH_USE(dev);
  CH_TRY(buf_new(dev, t, n_elems, wl.total, &b));
  b->wl = wl;
  b->m = m;
  b->k = k;
  const uint8_t* src = (const uint8_t*)bytes;
  const size_t nblk = wl.n_blocks;
  hipError_t e = hipSuccess;
  if (t == CRABML_HIP_F32 || t == CRABML_HIP_F16) {
    e = hipMemcpyAsync(b->ptr, src, wl.total, hipMemcpyHostToDevice, dev->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
  } else {
    // One-time re-layout into planes (see common.hpp); pure byte moves, no arithmetic.  The raw GGUF bytes are
    // DMA'd in chunks into a device staging buffer (the source is the caller's mmap'd file: pageable memory, which
    // the runtime streams through its pinned bounce buffers) and split into planes ON THE DEVICE -- the host never
    // touches the bytes (tools/upload_lab.hip: pageable hipMemcpyAsync in 128 MiB chunks runs at 56 GB/s on this box,
    // the PCIe rate; registering the range first or an own pinned ring are slower), and the two staging buffers
    // ping-pong so the re-layout of chunk i overlaps the copy of chunk i + 1.
    // per format: where each piece of a GGUF block goes (source offset, length, destination plane offset)
    RepackPlan plan{};
    auto seg = [&](int src_off, int len, size_t dst_off, int stride = 0) {
      plan.src_off2[plan.nseg] = src_off / 2;
      plan.len2[plan.nseg] = len / 2;
      plan.stride2[plan.nseg] = (stride ? stride : len) / 2;  // bytes per block in the destination plane
      plan.dst_off[plan.nseg] = dst_off;
      plan.nseg++;
    };
    switch (t) {
      case CRABML_HIP_Q4_0: seg(0, 2, wl.off_scale); seg(2, 16, 0); break;
      case CRABML_HIP_Q8_0: seg(0, 2, wl.off_scale); seg(2, 32, 0); break;
      case CRABML_HIP_Q4_1: seg(0, 4, wl.off_scale); seg(4, 16, 0); break;
      case CRABML_HIP_Q4_K: seg(0, 16, wl.off_scale); seg(16, 128, 0); break;
      case CRABML_HIP_Q5_K:  // qs | qh | scales | d | dmin (buf_q5_k.rs:13-21) -> qs | qh | (d, dmin, scales)
        seg(0, 128, 0); seg(128, 32, wl.off_scale); seg(172, 4, wl.off_scale + nblk * 32, 16); seg(160, 12, wl.off_scale + nblk * 32 + 4, 16);
        break;
      case CRABML_HIP_Q5_0: seg(0, 2, wl.off_scale + nblk * 4); seg(2, 4, wl.off_scale); seg(6, 16, 0); break;  // d | qh | qs
      case CRABML_HIP_Q5_1: seg(0, 8, wl.off_scale); seg(8, 16, 0); break;                                     // d, m, qh | qs
      case CRABML_HIP_Q2_K: seg(0, 16, wl.off_scale); seg(16, 64, 0); seg(80, 4, wl.off_scale + nblk * 16); break;  // scales | qs | d, dmin
      case CRABML_HIP_Q3_K:  // hmask | qs | scales | d (buf_q3_k.rs:19-30)
        seg(0, 32, wl.off_scale); seg(32, 64, 0); seg(96, 14, wl.off_scale + nblk * 32, 16);
        break;
      case CRABML_HIP_Q6_K:  // ql | qh | scales | d (buf_q6_k.rs:11-18)
        seg(0, 128, 0); seg(128, 64, wl.off_scale); seg(192, 16, wl.off_scale + nblk * 64); seg(208, 2, wl.off_scale + nblk * 80);
        break;
      case CRABML_HIP_Q8_K: seg(0, 4, wl.off_scale); seg(4, 256, 0); break;  // the trailing bsums are derived data: dropped
      default: break;
    }
    const size_t chunk_blocks = nblk < ((size_t)128 << 20) / bb ? nblk : ((size_t)128 << 20) / bb;
    void* stage[2] = {nullptr, nullptr};
    size_t stage_cap[2] = {0, 0};
    hipEvent_t done[2] = {nullptr, nullptr};
    int rc = 0;
    for (int i = 0; i < 2 && rc == 0; i++) {
      rc = pool_alloc(dev, chunk_blocks * bb, &stage[i], &stage_cap[i]);
      if (rc == 0 && hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess) rc = CRABML_HIP_UNEXPECTED;
    }
    if (rc == 0) {
      int slot = 0;
      for (size_t b0 = 0; b0 < nblk && e == hipSuccess; b0 += chunk_blocks, slot ^= 1) {
        const size_t nb = nblk - b0 < chunk_blocks ? nblk - b0 : chunk_blocks;
        e = hipEventSynchronize(done[slot]);  // the re-layout that last read this staging buffer has finished
        if (e == hipSuccess) e = hipMemcpyAsync(stage[slot], src + b0 * bb, nb * bb, hipMemcpyHostToDevice, dev->stream);
        if (e != hipSuccess) break;
        launch_repack(dev->stream, stage[slot], b->ptr, b0, nb, (int)bb, plan);
        if (t == CRABML_HIP_Q4_K) launch_q4k_pack_scales(dev->stream, (char*)b->ptr + wl.off_scale, b0, nb);
        if (t == CRABML_HIP_Q4_K) launch_q4k_class_major(dev->stream, b->ptr, b0, nb);
        if (t == CRABML_HIP_Q5_K) launch_q4k_pack_scales(dev->stream, (char*)b->ptr + wl.off_scale + nblk * 32, b0, nb);
        e = hipEventRecord(done[slot], dev->stream);
      }
      if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
    }
    for (int i = 0; i < 2; i++) {
      if (done[i]) (void)hipEventDestroy(done[i]);
      if (stage[i]) pool_free(dev, stage[i], stage_cap[i]);
    }
    if (rc != 0) {
      crabml_hip_buf_release(b);
      return rc;
    }
  }
  if (e != hipSuccess) {
    crabml_hip_buf_release(b);
    return hip_fail(dev, e, "upload", __FILE__, __LINE__);
  }
  *out = b;
  return 0;
}

int crabml_hip_buf_alloc(crabml_hip_device_t* dev, size_t n_elems, uint32_t t, crabml_hip_buf_t** out) {
  if (!dev || !out) return CRABML_HIP_BAD_INPUT;
  *out = nullptr;
  if (t != CRABML_HIP_F32 && t != CRABML_HIP_F16) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "only f32/f16 is supported");
  size_t bytes = n_elems * (t == CRABML_HIP_F32 ? 4 : 2);
  // the memory is bound -- and, F32, zero-filled: vec![0.0; n] (cpu_tensor.rs:146-149); F16 contents are unspecified -- when an
  // executed op first touches the buffer (a destination a fused launch makes redundant never costs a memset)
  crabml_hip_buf* b = buf_new_unbound(dev, t, n_elems, bytes);
  b->zero_init = t == CRABML_HIP_F32;
  if (!dev->lazy) {
    if (!dev->dry) CH_USE(dev);
    int rc = ensure_mem(dev, b);
    if (rc != 0) {
      crabml_hip_buf_release(b);
      return rc;
    }
  }
  *out = b;
  return 0;
}

int crabml_hip_buf_retain(crabml_hip_buf_t* b) {
  if (!b) return CRABML_HIP_BAD_INPUT;
  b->refcnt.fetch_add(1);
  return 0;
}

int crabml_hip_buf_release(crabml_hip_buf_t* b) {
  if (!b) return 0;
  if (b->refcnt.fetch_sub(1) == 1) {
    if (b->ptr) pool_free(b->dev, b->ptr, b->cap);
    if (b->qc.ptr) pool_free(b->dev, b->qc.ptr, b->qc.cap);
    delete b;
  }
  return 0;
}

uint32_t crabml_hip_buf_dtype(const crabml_hip_buf_t* b) { return b ? b->dtype : 0xffffffffu; }
size_t crabml_hip_buf_len(const crabml_hip_buf_t* b) { return b ? b->n_elems : 0; }

// ---- data movement -----------------------------------------------------------------------------------
// the host looks at a buffer: run what was recorded, bind the buffer (a handle nothing has written yet reads as Tensor::alloc
// left it)
static int observe(crabml_hip_device* dev, const crabml_hip_buf* b) {
  CH_FLUSH(dev);
  lazy_use(dev, b);
  return ensure_mem(dev, const_cast<crabml_hip_buf*>(b));
}

int crabml_hip_export(crabml_hip_device_t* dev, const crabml_hip_buf_t* b, float* dst, size_t n) {
  if (!dev || !b || (!dst && n)) return CRABML_HIP_BAD_INPUT;
  if (b->dtype != CRABML_HIP_F32) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "export: not f32, but got %u", b->dtype);
  if (n > b->n_elems) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "export: %zu elements requested, buffer holds %zu", n, b->n_elems);
  if (!dev->dry) CH_USE(dev);
  CH_FLUSH(dev);
  // the logits of a token the fused step served are already on their way to pinned host memory (sent by the step's last kernels):
  // wait for THAT data -- a flag in host memory -- and copy; the handle never needs device memory for this
  if (n && lazy_pinned_kind(dev, b, n) == 1) return lazy_export_wait(dev, dst, n);
  if (dev->dry) {
    CH_TRY(observe(dev, b));
    if (n) memset(dst, 0, n * 4);
    return 0;
  }
  lazy_use(dev, b);
  CH_TRY(ensure_mem(dev, const_cast<crabml_hip_buf*>(b)));
  if (n) CH_HIP(dev, hipMemcpyAsync(dst, b->ptr, n * 4, hipMemcpyDeviceToHost, dev->stream));
  CH_TRY(lazy_fault_request(dev));
  const auto t0 = std::chrono::steady_clock::now();
  CH_HIP(dev, hipStreamSynchronize(dev->stream));
  dev->lz->stats.wait_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  return lazy_fault_check(dev);
}

int crabml_hip_export_raw(crabml_hip_device_t* dev, const crabml_hip_buf_t* b, void* dst, size_t nbytes) {
  if (!dev || !b || (!dst && nbytes)) return CRABML_HIP_BAD_INPUT;
  if (b->dtype != CRABML_HIP_F32 && b->dtype != CRABML_HIP_F16)
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "export_raw: only f32/f16 buffers");
  size_t have = b->n_elems * (b->dtype == CRABML_HIP_F32 ? 4 : 2);
  if (nbytes > have) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "export_raw: %zu bytes requested, buffer holds %zu", nbytes, have);
  if (dev->dry) {
    CH_TRY(observe(dev, b));
    if (nbytes) memset(dst, 0, nbytes);
    return 0;
  }
  CH_USE(dev);
  CH_TRY(observe(dev, b));
  if (nbytes) CH_HIP(dev, hipMemcpyAsync(dst, b->ptr, nbytes, hipMemcpyDeviceToHost, dev->stream));
  CH_TRY(lazy_fault_request(dev));
  CH_HIP(dev, hipStreamSynchronize(dev->stream));
  return lazy_fault_check(dev);
}

#define CH_USE_LIVE(dev)          \
  do {                            \
    if (!(dev)->dry) CH_USE(dev); \
  } while (0)

// hands a recorded (or, per-op mode, executed) op's fresh output to the caller
static int submit_out(crabml_hip_device* dev, LazyOp& op, crabml_hip_buf_t** out) {
  int rc = submit(dev, op);
  if (rc != 0) {
    crabml_hip_buf_release(op.out);
    return rc;
  }
  *out = op.out;
  return 0;
}

int crabml_hip_dup(crabml_hip_device_t* dev, const crabml_hip_buf_t* src, crabml_hip_buf_t** out) {
  if (!dev || !src || !out) return CRABML_HIP_BAD_INPUT;
  CH_USE_LIVE(dev);
  *out = nullptr;
  if (src->dtype != CRABML_HIP_F32) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "dup: not f32, but got %u", src->dtype);
  lazy_use(dev, src);
  LazyOp op;
  op.kind = LZ_DUP;
  op.a = const_cast<crabml_hip_buf*>(src);
  op.out = buf_new_unbound(dev, CRABML_HIP_F32, src->n_elems, src->n_elems * 4);
  return submit_out(dev, op, out);
}

static void pad3(const size_t* v, int ndim, size_t fill, size_t out[3]) {
  int off = 3 - ndim;
  for (int i = 0; i < 3; i++) out[i] = fill;
  for (int i = 0; i < ndim; i++) out[off + i] = v[i];
}

int crabml_hip_contiguous(crabml_hip_device_t* dev, const crabml_hip_buf_t* src, const size_t* shape,
                          const size_t* strides, int ndim, crabml_hip_buf_t** out) {
  if (!dev || !src || !out || !shape || !strides) return CRABML_HIP_BAD_INPUT;
  CH_USE_LIVE(dev);
  *out = nullptr;
  if (ndim != 2 && ndim != 3) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "contiguous: only 2-d / 3-d tensors");
  if (src->dtype != CRABML_HIP_F32 && src->dtype != CRABML_HIP_F16)
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "contiguous: only f32/f16");
  LazyOp op;
  op.kind = LZ_CONTIGUOUS;
  size_t *sh = op.s, *st = op.s + 3;
  pad3(shape, ndim, 1, sh);
  pad3(strides, ndim, 0, st);
  size_t n = sh[0] * sh[1] * sh[2];
  size_t max_off = 0;
  for (int i = 0; i < 3; i++)
    if (sh[i]) max_off += (sh[i] - 1) * st[i];
  if (n && max_off >= src->n_elems) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "contiguous: view exceeds the buffer");
  int es = src->dtype == CRABML_HIP_F32 ? 4 : 2;
  lazy_use(dev, src);
  op.a = const_cast<crabml_hip_buf*>(src);
  op.out = buf_new_unbound(dev, src->dtype, n, n * es);
  return submit_out(dev, op, out);
}

int crabml_hip_concatenate(crabml_hip_device_t* dev, crabml_hip_buf_t* dst, const size_t* dshape,
                           const size_t* dstrides, const crabml_hip_buf_t* rhs, const size_t* rshape,
                           const size_t* rstrides, int ndim, int axis) {
  if (!dev || !dst || !rhs || !dshape || !dstrides || !rshape || !rstrides) return CRABML_HIP_BAD_INPUT;
  CH_USE_LIVE(dev);
  if (ndim < 1 || ndim > 3 || axis < 0 || axis >= ndim) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "concatenate: bad ndim/axis");
  if (dst->dtype != CRABML_HIP_F32 && dst->dtype != CRABML_HIP_F16)
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "only f32/f16 is supported on concatenate");
  if (rhs->dtype != CRABML_HIP_F32 && rhs->dtype != CRABML_HIP_F16)
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "only f32/f16 is supported on concatenate rhs");
  if (dst->dtype == CRABML_HIP_F32 && rhs->dtype == CRABML_HIP_F16)
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "can not concatenate F32 and F16");
  for (int i = 0; i < ndim; i++)
    if (i != axis && dshape[i] != rshape[i])
      CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "shape mismatch on concatenate");
  LazyOp op;
  op.kind = LZ_CONCAT;
  size_t *sh = op.s, *ds = op.s + 3, *ss = op.s + 6;
  pad3(rshape, ndim, 1, sh);
  pad3(dstrides, ndim, 0, ds);
  pad3(rstrides, ndim, 0, ss);
  size_t off = dshape[axis] * dstrides[axis];
  size_t n = sh[0] * sh[1] * sh[2];
  if (n) {
    size_t dmax = off, smax = 0;
    for (int i = 0; i < 3; i++) {
      dmax += (sh[i] - 1) * ds[i];
      smax += (sh[i] - 1) * ss[i];
    }
    if (dmax >= dst->n_elems) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "concatenate: destination is full");
    if (smax >= rhs->n_elems) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "concatenate: rhs view exceeds its buffer");
  }
  op.s[9] = off;
  op.s[10] = (size_t)(axis + 3 - ndim);
  op.s[11] = dshape[axis];
  lazy_use(dev, dst);
  lazy_use(dev, rhs);
  op.a = dst;
  op.b = const_cast<crabml_hip_buf*>(rhs);
  return submit(dev, op);
}

int crabml_hip_copy_rows_from(crabml_hip_device_t* dev, crabml_hip_buf_t* dst, const crabml_hip_buf_t* src, size_t cols,
                              const size_t* rows, size_t n_rows) {
  if (!dev || !dst || !src || (!rows && n_rows)) return CRABML_HIP_BAD_INPUT;
  CH_USE_LIVE(dev);
  if (dst->dtype != CRABML_HIP_F32 && dst->dtype != CRABML_HIP_F16)
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "only f32/f16 can be copied to");
  size_t be = block_elems(src->dtype);
  if (be == 0) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "copy_rows_from: unsupported source dtype %u", src->dtype);
  if (n_rows * cols > dst->n_elems) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "copy_rows_from: destination too small");
  for (size_t i = 0; i < n_rows; i++) {
    size_t start = rows[i] * cols;
    if (start + cols > src->n_elems) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "copy_rows_from: row %zu out of range", rows[i]);
    if (be > 1 && start % be != 0)  // QuantBuf*::dequantize asserts start % block == 0
      CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "copy_rows_from: row start %zu is not block aligned", start);
  }
  lazy_use(dev, dst);
  lazy_use(dev, src);
  if (n_rows == 1) {  // the embedding lookup / the last row of a batch (llama2.rs:222-223, 192-197)
    LazyOp op;
    op.kind = LZ_COPY_ROW;
    op.a = dst;
    op.b = const_cast<crabml_hip_buf*>(src);
    op.s[0] = cols;
    op.s[1] = rows[0];
    return submit(dev, op);
  }
  // several rows (a prompt batch, dequantize()): run what is queued, then one launch per row right away
  CH_FLUSH(dev);
  CH_TRY(ensure_mem(dev, dst));
  CH_TRY(ensure_mem(dev, const_cast<crabml_hip_buf*>(src)));
  if (dev->dry) return 0;
  for (size_t i = 0; i < n_rows; i++) {
    char* d = (char*)dst->ptr + i * cols * (dst->dtype == CRABML_HIP_F32 ? 4 : 2);
    launch_dequant_row(dev->stream, src, rows[i] * cols, cols, d, dst->dtype == CRABML_HIP_F16);
  }
  touch(dst);
  return 0;
}

// ---- compute -----------------------------------------------------------------------------------------
static int need_f32(crabml_hip_device* dev, const crabml_hip_buf* b, size_t n, const char* op) {
  if (b->dtype != CRABML_HIP_F32) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "%s: not f32, but got %u", op, b->dtype);
  if (n > b->n_elems) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "%s: %zu elements exceed the buffer (%zu)", op, n, b->n_elems);
  return 0;
}

int crabml_hip_rope_inplace(crabml_hip_device_t* dev, crabml_hip_buf_t* x, size_t n_batch, size_t bi_stride,
                            size_t head_dim, uint32_t mode, size_t pos, size_t rope_dims) {
  if (!dev || !x) return CRABML_HIP_BAD_INPUT;
  CH_USE_LIVE(dev);
  CH_TRY(need_f32(dev, x, n_batch * bi_stride, "rope"));
  if (head_dim == 0 || rope_dims > head_dim || (mode != 0 && mode != 1))
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "rope: bad head_dim/rope_dims/mode");
  size_t npairs = mode == 0 ? (rope_dims + 1) / 2 : rope_dims / 2;
  if (npairs > 256) CH_BAIL(dev, CRABML_HIP_NOT_IMPLEMENTED, "rope: more than 256 rotary pairs");
  if (mode == 0 && (rope_dims & 1) && rope_dims + 1 > head_dim)
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "rope: odd rope_dims reaches past the head");
  lazy_use(dev, x);
  LazyOp op;
  op.kind = LZ_ROPE;
  op.a = x;
  op.s[0] = n_batch;
  op.s[1] = bi_stride;
  op.s[2] = head_dim;
  op.s[3] = mode;
  op.s[4] = pos;
  op.s[5] = rope_dims;
  return submit(dev, op);
}

int crabml_hip_rms_norm_inplace(crabml_hip_device_t* dev, crabml_hip_buf_t* x, size_t rows, size_t cols, float eps) {
  if (!dev || !x) return CRABML_HIP_BAD_INPUT;
  CH_USE_LIVE(dev);
  CH_TRY(need_f32(dev, x, rows * cols, "rms_norm"));
  if (cols % 32 != 0) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "rms_norm: row length %zu is not a multiple of 32", cols);
  if (cols / 32 * 4 > 64 * 1024) CH_BAIL(dev, CRABML_HIP_NOT_IMPLEMENTED, "rms_norm: row too long");
  lazy_use(dev, x);
  LazyOp op;
  op.kind = LZ_RMS_NORM;
  op.a = x;
  op.s[0] = rows;
  op.s[1] = cols;
  op.f = eps;
  return submit(dev, op);
}

int crabml_hip_softmax_inplace(crabml_hip_device_t* dev, crabml_hip_buf_t* x, size_t rows, size_t cols) {
  if (!dev || !x) return CRABML_HIP_BAD_INPUT;
  CH_USE_LIVE(dev);
  CH_TRY(need_f32(dev, x, rows * cols, "softmax"));
  lazy_use(dev, x);
  LazyOp op;
  op.kind = LZ_SOFTMAX;
  op.a = x;
  op.s[0] = rows;
  op.s[1] = cols;
  return submit(dev, op);
}

static int unary(crabml_hip_device* dev, uint8_t kind, crabml_hip_buf* x, size_t n, const char* name) {
  if (!dev || !x) return CRABML_HIP_BAD_INPUT;
  CH_USE_LIVE(dev);
  CH_TRY(need_f32(dev, x, n, name));
  lazy_use(dev, x);
  LazyOp op;
  op.kind = kind;
  op.a = x;
  op.s[0] = n;
  return submit(dev, op);
}
int crabml_hip_silu_inplace(crabml_hip_device_t* dev, crabml_hip_buf_t* x, size_t n) { return unary(dev, LZ_SILU, x, n, "silu"); }
int crabml_hip_gelu_inplace(crabml_hip_device_t* dev, crabml_hip_buf_t* x, size_t n) { return unary(dev, LZ_GELU, x, n, "gelu"); }

static int binary(crabml_hip_device* dev, int op_, crabml_hip_buf* a, size_t na, const crabml_hip_buf* b, size_t nb) {
  if (!dev || !a || !b) return CRABML_HIP_BAD_INPUT;
  CH_USE_LIVE(dev);
  CH_TRY(need_f32(dev, a, na, op_ ? "mul" : "add"));
  CH_TRY(need_f32(dev, b, nb, op_ ? "mul" : "add"));
  if (nb == 0 || na % nb != 0) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "%s: lhs len %zu is not a multiple of rhs len %zu", op_ ? "mul" : "add", na, nb);
  lazy_use(dev, a);
  lazy_use(dev, b);
  LazyOp op;
  op.kind = op_ ? LZ_MUL : LZ_ADD;
  op.a = a;
  op.b = const_cast<crabml_hip_buf*>(b);
  op.s[0] = na;
  op.s[1] = nb;
  return submit(dev, op);
}
int crabml_hip_mul_inplace(crabml_hip_device_t* dev, crabml_hip_buf_t* a, size_t na, const crabml_hip_buf_t* b, size_t nb) {
  return binary(dev, 1, a, na, b, nb);
}
int crabml_hip_add_inplace(crabml_hip_device_t* dev, crabml_hip_buf_t* a, size_t na, const crabml_hip_buf_t* b, size_t nb) {
  return binary(dev, 0, a, na, b, nb);
}
int crabml_hip_scale_inplace(crabml_hip_device_t* dev, crabml_hip_buf_t* a, size_t na, float f) {
  if (!dev || !a) return CRABML_HIP_BAD_INPUT;
  CH_USE_LIVE(dev);
  CH_TRY(need_f32(dev, a, na, "scale"));
  lazy_use(dev, a);
  LazyOp op;
  op.kind = LZ_SCALE;
  op.a = a;
  op.s[0] = na;
  op.f = f;
  return submit(dev, op);
}

int crabml_hip_matmul_vec(crabml_hip_device_t* dev, const crabml_hip_buf_t* w, size_t m, size_t k,
                          const crabml_hip_buf_t* x, size_t b, crabml_hip_buf_t** out) {
  if (!dev || !w || !x || !out) return CRABML_HIP_BAD_INPUT;
  CH_USE_LIVE(dev);
  *out = nullptr;
  uint32_t qt = vec_dot_rhs_dtype(w->dtype);
  if (qt == 0xffffffffu) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "matmul_vec: unsupported weight dtype %u", w->dtype);
  CH_TRY(need_f32(dev, x, b * k, "matmul_vec rhs"));
  if (m * k > w->n_elems) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "matmul_vec: (%zu,%zu) exceeds the weight buffer", m, k);
  if (block_elems(w->dtype) > 1 && w->k != k)
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "matmul_vec: quantized weight has k=%zu, called with k=%zu", w->k, k);
  if (k % block_elems(qt) != 0 || k % block_elems(w->dtype) != 0)
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "matmul_vec: k=%zu is not a multiple of the block size", k);
  if (m > 0x7fffffff || k > 0x7fffffff) CH_BAIL(dev, CRABML_HIP_NOT_IMPLEMENTED, "matmul_vec: dimension too large");
  lazy_use(dev, w);
  lazy_use(dev, x);
  LazyOp op;
  op.kind = LZ_MATMUL_VEC;
  op.a = const_cast<crabml_hip_buf*>(w);
  op.b = const_cast<crabml_hip_buf*>(x);
  op.s[0] = m;
  op.s[1] = k;
  op.s[2] = b;
  op.out = buf_new_unbound(dev, CRABML_HIP_F32, b * m, b * m * 4);
  return submit_out(dev, op, out);
}

int crabml_hip_batch_matmul(crabml_hip_device_t* dev, const crabml_hip_buf_t* a, size_t ba, size_t m, size_t k,
                            const crabml_hip_buf_t* b, size_t bb, size_t n, size_t sb0, size_t sb1, size_t sb2,
                            crabml_hip_buf_t** out) {
  if (!dev || !a || !b || !out) return CRABML_HIP_BAD_INPUT;
  CH_USE_LIVE(dev);
  *out = nullptr;
  CH_TRY(need_f32(dev, a, ba * m * k, "batch_matmul lhs"));
  if (b->dtype != CRABML_HIP_F32 && b->dtype != CRABML_HIP_F16)
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "batch_matmul: rhs must be f32/f16");
  if (!(sb1 == 1 || sb2 == 1)) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "batch_matmul: rhs must be contiguous on k or n");
  if (bb == 0 || ba < bb || ba % bb != 0) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "batch_matmul: lhs batch %zu vs rhs batch %zu", ba, bb);
  if (bb && k && n) {
    size_t max_off = (bb - 1) * sb0 + (k - 1) * sb1 + (n - 1) * sb2;
    if (max_off >= b->n_elems) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "batch_matmul: rhs view exceeds its buffer");
  }
  lazy_use(dev, a);
  lazy_use(dev, b);
  LazyOp op;
  op.kind = LZ_BATCH_MATMUL;
  op.a = const_cast<crabml_hip_buf*>(a);
  op.b = const_cast<crabml_hip_buf*>(b);
  op.s[0] = ba;
  op.s[1] = m;
  op.s[2] = k;
  op.s[3] = bb;
  op.s[4] = n;
  op.s[5] = sb0;
  op.s[6] = sb1;
  op.s[7] = sb2;
  op.out = buf_new_unbound(dev, CRABML_HIP_F32, ba * m * n, ba * m * n * 4);
  return submit_out(dev, op, out);
}

// ---- parity / debug hooks ----------------------------------------------------------------------------
int crabml_hip_debug_quantize(crabml_hip_device_t* dev, const crabml_hip_buf_t* x, size_t n, uint32_t qt, void* dst,
                              size_t dst_bytes) {
  if (!dev || !x || !dst) return CRABML_HIP_BAD_INPUT;
  CH_LIVE(dev);
  CH_USE(dev);
  CH_TRY(observe(dev, x));
  CH_TRY(need_f32(dev, x, n, "debug_quantize"));
  if (qt != CRABML_HIP_Q8_0 && qt != CRABML_HIP_Q8_1 && qt != CRABML_HIP_Q8_K)
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "debug_quantize: unsupported target %u", qt);
  size_t be = block_elems(qt), bb = block_bytes(qt);
  if (n % be != 0) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "debug_quantize: n is not a multiple of the block size");
  size_t nb = n / be;
  if (dst_bytes < nb * bb) CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "debug_quantize: dst too small");
  ActLayout al = act_layout(qt, n);
  void* planes = nullptr;
  size_t cap = 0;
  CH_TRY(pool_alloc(dev, al.total, &planes, &cap));
  launch_quantize_act(dev->stream, qt, (const float*)x->ptr, n, planes);
  std::vector<uint8_t> h(al.total);
  hipError_t e = hipMemcpyAsync(h.data(), planes, al.total, hipMemcpyDeviceToHost, dev->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
  pool_free(dev, planes, cap);
  if (e != hipSuccess) return hip_fail(dev, e, "debug_quantize", __FILE__, __LINE__);
  uint8_t* o = (uint8_t*)dst;
  for (size_t i = 0; i < nb; i++) {
    if (qt == CRABML_HIP_Q8_0) {
      memcpy(o + i * 34, h.data() + al.off_d + i * 2, 2);
      memcpy(o + i * 34 + 2, h.data() + i * 32, 32);
    } else if (qt == CRABML_HIP_Q8_1) {
      memcpy(o + i * 36, h.data() + al.off_d + i * 2, 2);
      memcpy(o + i * 36 + 2, h.data() + al.off_aux + i * 2, 2);
      memcpy(o + i * 36 + 4, h.data() + i * 32, 32);
    } else {
      memcpy(o + i * 292, h.data() + al.off_d + i * 4, 4);
      memcpy(o + i * 292 + 4, h.data() + i * 256, 256);
      memcpy(o + i * 292 + 260, h.data() + al.off_aux + i * 32, 32);
    }
  }
  return 0;
}

int crabml_hip_debug_block_dots(crabml_hip_device_t* dev, const crabml_hip_buf_t* w, size_t m, size_t k, size_t row,
                                const crabml_hip_buf_t* x, int32_t* dst) {
  if (!dev || !w || !x || !dst) return CRABML_HIP_BAD_INPUT;
  CH_LIVE(dev);
  CH_USE(dev);
  CH_TRY(observe(dev, w));
  CH_TRY(observe(dev, x));
  uint32_t qt = vec_dot_rhs_dtype(w->dtype);
  if (block_elems(w->dtype) <= 1 || qt == 0xffffffffu)
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "debug_block_dots: quantized weights only");
  if (row >= m || w->k != k || m * k > w->n_elems) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "debug_block_dots: bad row/shape");
  CH_TRY(need_f32(dev, x, k, "debug_block_dots rhs"));
  const void* act = nullptr;
  CH_TRY(ensure_act(dev, x, 1, k, qt, &act));
  // Q6_K / Q2_K / Q3_K: one value per 16-element scale group
  size_t n = w->dtype == CRABML_HIP_Q6_K || w->dtype == CRABML_HIP_Q2_K || w->dtype == CRABML_HIP_Q3_K ? k / 16 : k / 32;
  void* d = nullptr;
  size_t cap = 0;
  CH_TRY(pool_alloc(dev, n * 4, &d, &cap));
  launch_block_dots(dev->stream, w, k, row, act, (int32_t*)d);
  hipError_t e = hipMemcpyAsync(dst, d, n * 4, hipMemcpyDeviceToHost, dev->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
  pool_free(dev, d, cap);
  if (e != hipSuccess) return hip_fail(dev, e, "debug_block_dots", __FILE__, __LINE__);
  return 0;
}

int crabml_hip_debug_superblock_ints(crabml_hip_device_t* dev, const crabml_hip_buf_t* w, size_t m, size_t k, size_t row,
                                     const crabml_hip_buf_t* x, int32_t variant, int32_t* dst, float* value) {
  if (!dev || !w || !x || !dst) return CRABML_HIP_BAD_INPUT;
  CH_LIVE(dev);
  CH_USE(dev);
  CH_TRY(observe(dev, w));
  CH_TRY(observe(dev, x));
  if (w->dtype != CRABML_HIP_Q4_K && w->dtype != CRABML_HIP_Q6_K)
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "debug_superblock_ints: Q4_K / Q6_K weights only");
  if (row >= m || w->k != k || m * k > w->n_elems || k % 256 != 0)
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "debug_superblock_ints: bad row/shape");
  CH_TRY(need_f32(dev, x, k, "debug_superblock_ints rhs"));
  const void* act = nullptr;
  CH_TRY(ensure_act(dev, x, 1, k, CRABML_HIP_Q8_K, &act));
  const size_t nsb = k / 256, npieces = nsb * 8;
  void* d = nullptr;
  size_t cap = 0;
  CH_TRY(pool_alloc(dev, npieces * 8 + 16, &d, &cap));
  float* fout = (float*)((char*)d + npieces * 8);
  int rc = launch_piece_ints(dev, w, m, k, row, act, variant, (int32_t*)d, fout);
  std::vector<int32_t> h(npieces * 2);
  float fv = 0.f;
  hipError_t e = rc == 0 ? hipMemcpyAsync(h.data(), d, npieces * 8, hipMemcpyDeviceToHost, dev->stream) : hipSuccess;
  if (rc == 0 && e == hipSuccess) e = hipMemcpyAsync(&fv, fout, 4, hipMemcpyDeviceToHost, dev->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
  pool_free(dev, d, cap);
  if (rc != 0) return rc;
  if (e != hipSuccess) return hip_fail(dev, e, "debug_superblock_ints", __FILE__, __LINE__);
  for (size_t sb = 0; sb < nsb; sb++) {  // the 8 pieces of a super-block: plain integer sums
    long long a = 0, b = 0;
    for (size_t j = 0; j < 8; j++) {
      a += h[(sb * 8 + j) * 2];
      b += h[(sb * 8 + j) * 2 + 1];
    }
    if (w->dtype == CRABML_HIP_Q6_K) {  // both halves are scaled group sums
      a += b;
      b = 0;
    }
    dst[2 * sb] = (int32_t)a;
    dst[2 * sb + 1] = (int32_t)b;
  }
  if (value) *value = fv;
  return 0;
}

int crabml_hip_debug_gemm_ints(crabml_hip_device_t* dev, const crabml_hip_buf_t* w, size_t m, size_t k, const crabml_hip_buf_t* x,
                               size_t b, int32_t* dst, float* out) {
  if (!dev || !w || !x || !dst) return CRABML_HIP_BAD_INPUT;
  CH_LIVE(dev);
  CH_USE(dev);
  CH_TRY(observe(dev, w));
  CH_TRY(observe(dev, x));
  if (w->dtype != CRABML_HIP_Q4_K && w->dtype != CRABML_HIP_Q6_K)
    CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "debug_gemm_ints: Q4_K / Q6_K weights only");
  if (b < 16 || w->k != k || m * k > w->n_elems || k % 256 != 0) CH_BAIL(dev, CRABML_HIP_TENSOR_ERROR, "debug_gemm_ints: needs b >= 16 rows and a matching shape");
  CH_TRY(need_f32(dev, x, b * k, "debug_gemm_ints rhs"));
  const void* act = nullptr;
  CH_TRY(ensure_act(dev, x, b, k, CRABML_HIP_Q8_K, &act));
  const size_t n = b * m * (k / 256) * 2;
  void *d = nullptr, *o = nullptr;
  size_t cap = 0, ocap = 0;
  CH_TRY(pool_alloc(dev, n * 4, &d, &cap));
  int rc = pool_alloc(dev, b * m * 4, &o, &ocap);
  if (rc != 0) {
    pool_free(dev, d, cap);
    return rc;
  }
  hipError_t e = hipMemsetAsync(d, 0xff, n * 4, dev->stream);
  const bool ran = e == hipSuccess && launch_gemm_mfma(dev, w, m, k, act, b, (float*)o, nullptr, (int*)d);
  if (ran) e = hipMemcpyAsync(dst, d, n * 4, hipMemcpyDeviceToHost, dev->stream);
  if (ran && e == hipSuccess && out) e = hipMemcpyAsync(out, o, b * m * 4, hipMemcpyDeviceToHost, dev->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
  pool_free(dev, d, cap);
  pool_free(dev, o, ocap);
  if (e != hipSuccess) return hip_fail(dev, e, "debug_gemm_ints", __FILE__, __LINE__);
  if (!ran) CH_BAIL(dev, CRABML_HIP_NOT_IMPLEMENTED, "debug_gemm_ints: the matrix-core GEMM did not take this shape");
  return 0;
}

int crabml_hip_debug_read_ceiling(crabml_hip_device_t* dev, size_t bytes, int32_t reps, double* gbytes_per_s) {
  if (!dev || !gbytes_per_s || reps < 1) return CRABML_HIP_BAD_INPUT;
  CH_LIVE(dev);
  CH_USE(dev);
  CH_FLUSH(dev);
  bytes = bytes / 4096 * 4096;
  if (bytes < ((size_t)1 << 20)) CH_BAIL(dev, CRABML_HIP_BAD_INPUT, "debug_read_ceiling: at least 1 MiB");
  void *buf = nullptr, *sink = nullptr;
  size_t cap = 0, scap = 0;
  CH_TRY(pool_alloc(dev, bytes, &buf, &cap));
  int rc = pool_alloc(dev, 256, &sink, &scap);
  if (rc != 0) {
    pool_free(dev, buf, cap);
    return rc;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  if (e == hipSuccess) e = hipMemsetAsync(buf, 1, bytes, dev->stream);
  float best = 0.f;
  for (int pattern = 0; pattern < 3; pattern++)  // the ceiling is the best of the three access patterns (elementwise.hip)
    for (int i = 0; i < reps + 1 && e == hipSuccess; i++) {  // first launch = warm-up
      launch_stream_read(dev->stream, buf, bytes, (int*)sink, e0, e1, pattern);
      e = hipEventSynchronize(e1);
      float ms = 0.f;
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
      if (i > 0 && e == hipSuccess && (best == 0.f || ms < best)) best = ms;
    }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  pool_free(dev, buf, cap);
  pool_free(dev, sink, scap);
  if (e != hipSuccess) return hip_fail(dev, e, "debug_read_ceiling", __FILE__, __LINE__);
  *gbytes_per_s = best > 0.f ? (double)bytes / ((double)best * 1e-3) / 1e9 : 0.0;
  return 0;
}

int crabml_hip_debug_device_numa_node(crabml_hip_device_t* dev, int32_t* node) {
  if (!dev || !node) return CRABML_HIP_BAD_INPUT;
  *node = dev->numa_node;
  return 0;
}

int crabml_hip_prof_enable(crabml_hip_device_t* dev, int on) {
  if (!dev) return CRABML_HIP_BAD_INPUT;
  dev->prof_on = on != 0;
  return 0;
}

int crabml_hip_prof_read_launches(crabml_hip_device_t* dev, float* ms_out, size_t cap, size_t* n) {
  if (!dev || !n || (!ms_out && cap)) return CRABML_HIP_BAD_INPUT;
  CH_LIVE(dev);
  CH_USE(dev);
  CH_FLUSH(dev);
  CH_HIP(dev, hipStreamSynchronize(dev->stream));
  size_t i = 0;
  for (auto& r : dev->prof_recs) {
    float ms = 0.f;
    CH_HIP(dev, hipEventElapsedTime(&ms, r.e0, r.e1));
    if (i < cap) ms_out[i++] = ms;
    dev->prof_free_events.push_back(r.e0);
    dev->prof_free_events.push_back(r.e1);
  }
  dev->prof_recs.clear();
  *n = i;
  return 0;
}

int crabml_hip_prof_read(crabml_hip_device_t* dev, crabml_hip_prof_entry_t* out, size_t cap, size_t* n) {
  if (!dev || !n || (!out && cap)) return CRABML_HIP_BAD_INPUT;
  CH_LIVE(dev);
  CH_USE(dev);
  CH_FLUSH(dev);
  CH_HIP(dev, hipStreamSynchronize(dev->stream));
  std::map<uint64_t, crabml_hip_prof_entry_t> agg;
  for (auto& r : dev->prof_recs) {
    float ms = 0.f;
    CH_HIP(dev, hipEventElapsedTime(&ms, r.e0, r.e1));
    auto& e = agg[((uint64_t)r.stage << 32) | r.dtype];
    e.dtype = r.dtype;
    e.reserved = r.stage;
    e.launches++;
    e.kernel_ms += ms;
    e.algo_bytes += r.bytes;
    dev->prof_free_events.push_back(r.e0);
    dev->prof_free_events.push_back(r.e1);
  }
  dev->prof_recs.clear();
  size_t i = 0;
  for (auto& kv : agg)
    if (i < cap) out[i++] = kv.second;
  *n = i;
  return 0;
}

}  // extern "C"
