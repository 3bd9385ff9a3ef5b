// kernels.hpp -- host-side launchers of the gfx950 kernels (one per reference primitive).
#pragma once
#include <hip/hip_ext.h>

#include "common.hpp"

namespace crabml_hip {

// kernel launch that optionally carries a profiling event pair (hipExtLaunchKernelGGL start/stop events)
template <typename K, typename... A>
inline void launch_k(hipStream_t st, crabml_hip_device::ProfRec* rec, K kernel, dim3 grid, dim3 block, size_t lds, A... args) {
  if (rec)
    hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, st, rec->e0, rec->e1, 0, args...);
  else
    hipLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, st, args...);
}

// ---- quantize.hip: activation quantizers (buf_q8_0.rs:87-134, buf_q8_1.rs:90-129, buf_q8_k.rs:84-131)
void launch_quantize_act(hipStream_t st, uint32_t qtype, const float* x, size_t n, void* planes);
void launch_quantize_act_rows(hipStream_t st, uint32_t qtype, const float* x, size_t rows, size_t n, void* planes, void* xh = nullptr,
                              int xh_order = 0);  // xh: the rows' f16 planes for gemm_f16w.hip, written alongside (f16w_rows.hpp)

// ---- gemv.hip: W(m,k) x quantized activations (b,k) -> out (b,m)
// wq: weight planes; aq: activation planes (one set per batch row, stride act_layout(qtype,k).total)
// rec != nullptr: the (first) kernel is launched with the record's event pair (measurement hook)
// fused_add (the FAST prompt pass only): a batched rhs's block term takes its second product and the add as one fma -- an explicit
// argument, never device state: matmul_vec itself is bit-exact and must not pick it up by accident
int launch_gemv(crabml_hip_device* dev, const crabml_hip_buf* w, size_t m, size_t k, const void* act, size_t b,
                float* out, crabml_hip_device::ProfRec* rec = nullptr, bool fused_add = false);
// gemv_strict.hip: same contract, block terms added in the reference's scalar order (bit-exact; slow)
int launch_gemv_strict(crabml_hip_device* dev, const crabml_hip_buf* w, size_t m, size_t k, const void* act, size_t b,
                       float* out, const float* add = nullptr);  // add: out[i] = sum + add[i] (the residual; may alias out)
// weight upload: scatter the pieces of `n_blocks` GGUF blocks (`bb` bytes each) starting at block `blk0` into their
// planes (byte moves only).  Piece s = bytes [2 src_off2, 2 src_off2 + 2 len2) of a block -> base + dst_off[s],
// one `stride2`-unit record per block (stride2 = len2 unless two pieces share a record: Q5_K's header); bytes covered by no
// piece are dropped.
struct RepackPlan {
  int nseg;
  int src_off2[4], len2[4], stride2[4];
  size_t dst_off[4];
};
void launch_repack(hipStream_t st, const void* raw, void* base, size_t blk0, size_t n_blocks, int bb, const RepackPlan& plan);
void launch_q4k_pack_scales(hipStream_t st, void* hdr_plane, size_t blk0, size_t n_blocks);
void launch_q4k_class_major(hipStream_t st, void* qs_plane, size_t blk0, size_t n_blocks);
// gemm_f16w.hip: the fast prompt pass's weight-stationary f16 GEMM (Q4_0 / Q8_0 weights x Q8_0 rows, Q4_1 x Q8_1, Q4_K / Q6_K x Q8_K)
// and the rows' pre-scaled f16 planes it reads, in the k-slot order of the weight format (gemm_f16w_order: Q4_K and Q6_K have their own)
struct F16wHQuant {  // gate | up launches: where h goes as Q8_0 / Q8_1 row planes (act_layout) and, optionally, as ffn_down's f16 planes
  char* planes = nullptr;
  size_t stride = 0, off_d = 0, off_aux = 0;
  unsigned short* xh = nullptr;
  int q81 = 0;
};
bool gemm_f16w_covers(uint32_t w_dtype, uint32_t act_qtype);
int gemm_f16w_order(uint32_t w_dtype);
size_t gemm_f16w_xh_bytes(size_t rows, size_t k);  // the allocation behind xh: whole 128-row tiles + the look-ahead's slack
bool launch_rows_to_f16(hipStream_t st, uint32_t act_qtype, uint32_t w_dtype, const void* planes, size_t rows, size_t k, void* xh);
bool launch_gemm_f16w(crabml_hip_device* dev, const crabml_hip_buf* const* w, const size_t* m, int nw, size_t k, const void* xh, size_t b,
                      float* const* out, float* ws = nullptr, size_t ws_floats = 0,  // ws: scratch for the partial tiles of k pieces
                      const unsigned short* gu_exp_tab = nullptr, int* gu_done = nullptr, int* defer_parts = nullptr,
                      const F16wHQuant* hq = nullptr);  // hq: *gu_done = 2 -- h left as quantized row planes, no f32 h
// defer_parts (one matrix): a launch cut into k pieces leaves piece 0 in out and pieces 1.. in ws (b * m floats apart) and returns
// their number instead of launching the reduce: the row kernel that consumes out adds them first, in piece order (prefill_rows.hpp)
// gu_exp_tab / gu_done (two matrices = ffn_gate, ffn_up): the launch may store h = silu(g) * u to out[0] instead of g and u (*gu_done = 1)
// batched rhs on the matrix cores (gemm_mfma.hip); false = not covered
bool launch_gemm_mfma(crabml_hip_device* dev, const crabml_hip_buf* w, size_t m, size_t k, const void* act, size_t b, float* out,
                      crabml_hip_device::ProfRec* rec, int* dbg = nullptr, bool fused_add = false);
int launch_piece_ints(crabml_hip_device* dev, const crabml_hip_buf* w, size_t m, size_t k, size_t row, const void* act, int variant,
                      int32_t* out, float* fout);
// elementwise.hip: plain streaming read of `bytes` bytes (crabml_hip_debug_read_ceiling), timed by the event pair
void launch_stream_read(hipStream_t st, const void* buf, size_t bytes, int* sink, hipEvent_t e0, hipEvent_t e1, int pattern);
void launch_block_dots(hipStream_t st, const crabml_hip_buf* w, size_t k, size_t row, const void* act, int32_t* out);

// ---- elementwise.hip
void launch_binary(hipStream_t st, int op /*0 add, 1 mul*/, float* a, size_t na, const float* b, size_t nb);
void launch_scale(hipStream_t st, float* a, size_t n, float f);
void launch_silu(hipStream_t st, float* x, size_t n, const uint16_t* exp_table);
void launch_gelu(hipStream_t st, float* x, size_t n, const uint16_t* gelu_table);
void launch_rms_norm(hipStream_t st, float* x, size_t rows, size_t cols, float eps);
void launch_softmax(hipStream_t st, float* x, size_t rows, size_t cols, const uint16_t* exp_table);
// rope: cos/sin of the reference's iterated-theta recurrence are evaluated on the host (same libm as
// the reference CPU path) and passed by value; see rope.rs:47-80.
struct RopeTable {
  float cs[512];  // (cos, sin) pairs, up to 256 rotary pairs
};
void launch_rope(hipStream_t st, float* x, size_t n_heads, size_t head_dim, int mode, size_t rope_dims,
                 const RopeTable& tab);
void launch_contiguous(hipStream_t st, const void* src, void* dst, int elem_size, const size_t shape[3],
                       const size_t strides[3]);
// dst_kind/src_kind: 0 = f32, 1 = f16
void launch_concatenate(hipStream_t st, void* dst, int dst_f16, size_t dst_off, const size_t dstrides[3],
                        const void* src, int src_f16, const size_t shape[3], const size_t sstrides[3]);
void launch_dequant_row(hipStream_t st, const crabml_hip_buf* src, size_t start_elem, size_t n, void* dst,
                        int dst_f16);

// ---- attention.hip: batch_matmul (batch_matmul.rs:15-131)
void launch_batch_matmul(hipStream_t st, const float* a, size_t ba, size_t m, size_t k, const void* b, int b_f16,
                         size_t bb, size_t n, size_t sb0, size_t sb1, size_t sb2, float* c);

}  // namespace crabml_hip
