// Gather / reduce operators around the GEMMs: token embedding, greedy argmax (+ next-step
// embedding gather), LayerNorm, row gather.  All are HBM/L2-bound byte movers: one workgroup per
// row, 16-byte (fp32) / 8-byte (bf16x4) accesses per lane, fixed-order reductions.
#include "argmax_dev.h"
#include "embed_dev.h"

namespace smoltts {

// ------------------------------------------------------------------------------------ embed
// (row body: embed_row in embed_dev.h, shared with the frame loop's commit kernel)
__global__ __launch_bounds__(256) void embed_kernel(const int* cols, int n_code_rows, const uint16_t* text_emb,
                                                    const uint16_t* cb_emb, int dim, int codebook_size,
                                                    int cb_first_offset, int mask_mode, int sem_start, int sem_end,
                                                    int text_rows, int cb_rows, float* x, EmitDev emit) {
  __shared__ float sh4[4];
  const int r = blockIdx.x;
  const EmbedTables t{text_emb, cb_emb, dim, codebook_size, cb_first_offset, mask_mode, sem_start, sem_end, text_rows, cb_rows, n_code_rows};
  embed_row(t, cols + (long)r * (1 + n_code_rows), r, x, emit, sh4);
}

int launch_embed(const int32_t* cols, int n_rows, int n_code_rows, const void* text_emb, const void* cb_emb, int dim,
                 int codebook_size, int cb_first_offset, int mask_mode, int sem_start, int sem_end, int text_rows,
                 int cb_rows, float* x, const EmitArgs* emit, hipStream_t stream) {
  EmitDev e{nullptr, nullptr, nullptr, nullptr, nullptr};
  if (emit) e = EmitDev{(char*)emit->x3a, emit->gamma_a, (char*)emit->x3b, emit->gamma_b, emit->ssq};
  ST_REQUIRE(emit == nullptr || dim % 64 == 0, SMOLTTS_E_INVALID, "embed: X3 emission needs dim %% 64 == 0");
  ST_REQUIRE(cols && text_emb && cb_emb && x && n_rows > 0 && dim % 4 == 0 && n_code_rows >= 1, SMOLTTS_E_INVALID,
             "embed: bad arguments");
  hipLaunchKernelGGL(embed_kernel, dim3(n_rows), dim3(256), 0, stream, cols, n_code_rows, (const uint16_t*)text_emb,
                     (const uint16_t*)cb_emb, dim, codebook_size, cb_first_offset, mask_mode, sem_start, sem_end, text_rows, cb_rows, x, e);
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

// ------------------------------------------------------------------------------------ argmax
// Row pick (argmax_dev.h) + optionally the next depth step's input: the embedding row of the picked code
// (lm/generate.py:134-140: fast_embeddings(code + i*codebook_size)) as fp32 residual row (+ X3 operand / sums of squares when a
// GEMM consumes it), and / or the precomputed layer-0 q | k | v of that row (QkvGather: no wqkv launch in the next step).
__global__ __launch_bounds__(256) void argmax_kernel(const float* logits, int n_cols, long ld, int* ids, int ids_stride,
                                                     float* margin, const int* margin_mask, const uint16_t* emb,
                                                     int emb_row_offset, int dim, float* xnext, EmitDev emit, SampleArgs sa, QkvGather qg) {
  __shared__ ArgmaxScratch S;
  __shared__ float sh4[4];
  const int r = blockIdx.x, tid = threadIdx.x;
  const int id = argmax_row(logits + (long)r * ld, n_cols, (ld & 3) == 0, r, margin, margin_mask, sa, S);
  if (tid == 0) ids[(long)r * ids_stride] = id;
  if (emb == nullptr) return;
  const long erow = (long)id + emb_row_offset;
  // everything that depends on the id is requested in one go: the embedding row and (where the engine has the table) the row's
  // layer-0 q | k | v with their RoPE rows -- one memory round trip behind the pick instead of one per loop iteration
  const int d0 = tid * 4;
  uint2 e0 = make_uint2(0, 0);
  if (d0 < dim) e0 = *reinterpret_cast<const uint2*>(emb + erow * dim + d0);
  QkvRegs qr;
  const int nqkv = qg.table ? (qg.n_q_heads + 2 * qg.n_kv_heads) * 64 : 0;
  if (qg.table) qkv_gather_load(qg, erow, 0, qr);
  float ss = 0.f;
  for (int d = d0; d < dim; d += 256 * 4) {
    const uint2 e = d == d0 ? e0 : *reinterpret_cast<const uint2*>(emb + erow * dim + d);
    const float4 o = make_float4(bf16_lo(e.x), bf16_hi(e.x), bf16_lo(e.y), bf16_hi(e.y));
    *reinterpret_cast<float4*>(xnext + (long)r * dim + d) = o;
    ss += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
    emit_x4(emit, r, d, dim >> 5, o.x, o.y, o.z, o.w);
  }
  if (qg.table) {
    qkv_gather_store(qg, r, 0, qr);
    for (int base = QG_MAX * 1024; base < nqkv; base += QG_MAX * 1024) {  // (rows beyond 2048 values: further rounds)
      qkv_gather_load(qg, erow, base, qr);
      qkv_gather_store(qg, r, base, qr);
    }
  }
  emit_row_ssq(emit, r, dim, ss, sh4);
}

int launch_argmax(const float* logits, int n_rows, int n_cols, int64_t ld, int32_t* ids, int ids_stride, float* margin,
                  const int32_t* margin_mask, const void* emb, int emb_row_offset, int dim, float* xnext,
                  const EmitArgs* emit, const SampleArgs* sample, hipStream_t stream, const QkvGather* qkv) {
  SampleArgs sa{0.f, 0.f, 0, 0, 0, nullptr, nullptr, nullptr};
  if (sample) sa = *sample;
  EmitDev e{nullptr, nullptr, nullptr, nullptr, nullptr};
  if (emit && emb) e = EmitDev{(char*)emit->x3a, emit->gamma_a, (char*)emit->x3b, emit->gamma_b, emit->ssq};
  QkvGather qg;
  memset(&qg, 0, sizeof(qg));
  if (qkv && emb) qg = *qkv;
  ST_REQUIRE(logits && ids && n_rows > 0 && n_cols > 0, SMOLTTS_E_INVALID, "argmax: bad arguments");
  ST_REQUIRE(emb == nullptr || (xnext && dim % 4 == 0), SMOLTTS_E_INVALID, "argmax: bad gather arguments");
  ST_REQUIRE(qg.table == nullptr || (qg.rope && qg.q_out && qg.kc && qg.vc && qg.pos >= 0 && qg.pos < qg.cache_len), SMOLTTS_E_INVALID,
             "argmax: bad q|k|v gather arguments");
  hipLaunchKernelGGL(argmax_kernel, dim3(n_rows), dim3(256), 0, stream, logits, n_cols, (long)ld, ids, ids_stride, margin,
                     margin_mask, (const uint16_t*)emb, emb_row_offset, dim, xnext, e, sa, qg);
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

// One fast-embedding row per workgroup as a depth-step input: fp32 row -> X3 operand (x gamma) + sums of squares, exactly what
// argmax_kernel emits for a picked code (smoltts_engine_build_fast_qkv feeds the layer-0 wqkv GEMM with whole tables of them).
__global__ __launch_bounds__(256) void emb_rows_pack_kernel(const uint16_t* emb, long row0, int dim, EmitDev emit) {
  __shared__ float sh4[4];
  const int r = blockIdx.x, tid = threadIdx.x;
  const long erow = row0 + r;
  float ss = 0.f;
  for (int d = tid * 4; d < dim; d += 256 * 4) {
    const uint2 e = *reinterpret_cast<const uint2*>(emb + erow * dim + d);
    const float4 o = make_float4(bf16_lo(e.x), bf16_hi(e.x), bf16_lo(e.y), bf16_hi(e.y));
    ss += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
    emit_x4(emit, r, d, dim >> 5, o.x, o.y, o.z, o.w);
  }
  emit_row_ssq(emit, r, dim, ss, sh4);
}

int launch_emb_rows_pack(const void* emb, int64_t row0, int n_rows, int dim, const EmitArgs& emit, hipStream_t stream) {
  ST_REQUIRE(emb && n_rows > 0 && dim % 64 == 0 && emit.x3a && emit.ssq, SMOLTTS_E_INVALID, "emb_rows_pack: bad arguments");
  const EmitDev e{(char*)emit.x3a, emit.gamma_a, (char*)emit.x3b, emit.gamma_b, emit.ssq};
  hipLaunchKernelGGL(emb_rows_pack_kernel, dim3(n_rows), dim3(256), 0, stream, (const uint16_t*)emb, (long)row0, dim, e);
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

// ------------------------------------------------------------------------------------ layernorm
// nn.LayerNorm(d_model) with bias, eps 1e-5 (mlx_inference/.../codec/transformer.py:113-114).
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, const float* w, const float* b, int dim, float eps,
                                                        float* out) {
  __shared__ float sh[8];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xr = x + (long)r * dim;
  float s = 0.f;
  for (int d = tid; d < dim; d += 256) s += xr[d];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) sh[wave] = s;
  __syncthreads();
  const float mean = (((sh[0] + sh[1]) + sh[2]) + sh[3]) / (float)dim;
  float v = 0.f;
  for (int d = tid; d < dim; d += 256) {
    const float c = xr[d] - mean;
    v = fmaf(c, c, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (lane == 0) sh[4 + wave] = v;
  __syncthreads();
  const float var = (((sh[4] + sh[5]) + sh[6]) + sh[7]) / (float)dim;
  const float rstd = 1.0f / sqrtf(var + eps);
  for (int d = tid; d < dim; d += 256) out[(long)r * dim + d] = (xr[d] - mean) * rstd * w[d] + b[d];
}

int launch_layernorm(const float* x, const float* w, const float* b, int n_rows, int dim, float eps, float* out,
                     hipStream_t stream) {
  ST_REQUIRE(x && w && b && out && n_rows > 0 && dim > 0, SMOLTTS_E_INVALID, "layernorm: bad arguments");
  hipLaunchKernelGGL(layernorm_kernel, dim3(n_rows), dim3(256), 0, stream, x, w, b, dim, eps, out);
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

// ------------------------------------------------------------------------------------ row gather
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* src, const int* idx, int dim, float* dst) {
  const int r = blockIdx.x;
  const long s = idx[r];
  for (int d = threadIdx.x * 4; d < dim; d += 256 * 4)
    *reinterpret_cast<float4*>(dst + (long)r * dim + d) = *reinterpret_cast<const float4*>(src + s * dim + d);
}

int launch_gather_rows(const float* src, const int32_t* idx, int n, int dim, float* dst, hipStream_t stream) {
  ST_REQUIRE(src && idx && dst && n > 0 && dim % 4 == 0, SMOLTTS_E_INVALID, "gather_rows: bad arguments");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(256), 0, stream, src, idx, dim, dst);
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

}  // namespace smoltts
