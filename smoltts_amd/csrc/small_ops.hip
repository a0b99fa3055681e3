// Gather / reduce operators around the GEMMs: token embedding, greedy argmax (+ next-step
// embedding gather), LayerNorm, row gather.  All are HBM/L2-bound byte movers: one workgroup per
// row, 16-byte (fp32) / 8-byte (bf16x4) accesses per lane, fixed-order reductions.
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
// torch.argmax / mx.argmax semantics: index of the first maximal element.  Also tracks the
// top-1/top-2 gap (parity diagnostics) and optionally gathers the next fast-step embedding
// (lm/generate.py:134-140: fast_embeddings(code + i*codebook_size)).
struct Top2 {
  float v1;
  int i1;
  float v2;
};
__device__ __forceinline__ Top2 top2_merge(Top2 a, Top2 b) {
  Top2 o;
  const bool a_first = (a.v1 > b.v1) || (a.v1 == b.v1 && a.i1 < b.i1);
  if (a_first) {
    o.v1 = a.v1; o.i1 = a.i1; o.v2 = fmaxf(a.v2, b.v1);
  } else {
    o.v1 = b.v1; o.i1 = b.i1; o.v2 = fmaxf(b.v2, a.v1);
  }
  return o;
}

// Counter-based uniform in (0, 1): a function of (seed, slot, frame, step, column) only, so sampling is
// reproducible under graph replay and independent of launch geometry.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float uniform01(uint64_t seed, int slot, int frame, int step, int col) {
  uint32_t h = mix32((uint32_t)seed ^ 0x9E3779B9U * (uint32_t)(slot + 1));
  h = mix32(h ^ (uint32_t)(seed >> 32) ^ 0x85EBCA6BU * (uint32_t)(frame + 1));
  h = mix32(h ^ 0xC2B2AE35U * (uint32_t)(step + 1));
  h = mix32(h ^ 0x27D4EB2FU * (uint32_t)(col + 1));
  return ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

// Greedy: first index of the row maximum (torch.argmax / mx.argmax).  Sampling (temp > 0): exact
// categorical sampling from softmax(logits / temp) by the Gumbel-max trick, optionally restricted to
// tokens with p >= min_p * p_max (the intent of lm/utils/samplers.py:8-34; as written there the
// threshold is compared with the token's own value and never removes anything).
__global__ __launch_bounds__(256) void argmax_kernel(const float* logits, int n_cols, long ld, int* ids, int ids_stride,
                                                     float* margin, const int* margin_mask, const uint16_t* emb,
                                                     int emb_row_offset, int dim, float* xnext, EmitDev emit, SampleArgs sa) {
  __shared__ Top2 sh[4];
  __shared__ Top2 sh2[4];
  __shared__ float sh4[4];
  __shared__ int s_id;
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + (long)r * ld;
  Top2 t{-INFINITY, 0x7fffffff, -INFINITY};
#define ST_TAKE(V, J)                                                                               \
  {                                                                                                 \
    const float v_ = (V);                                                                           \
    if (v_ > t.v1) { /* strictly greater keeps the earliest index inside a thread (j ascending) */ \
      t.v2 = t.v1; t.v1 = v_; t.i1 = (J);                                                           \
    } else if (v_ > t.v2) {                                                                         \
      t.v2 = v_;                                                                                    \
    }                                                                                               \
  }
  // the whole row in one round trip: up to 8 float4 per thread, all requested before the first compare (a scalar loop is a
  // chain of n_cols / 256 dependent L2 latencies: 8 for a 2048-entry codebook); kept for the sampling pass
  const bool vec = (n_cols & 3) == 0 && (ld & 3) == 0 && n_cols <= 8 * 1024;
  float4 v[8];
  if (vec) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = tid * 4 + i * 1024;
      v[i] = j < n_cols ? *reinterpret_cast<const float4*>(row + j) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = tid * 4 + i * 1024;
      if (j < n_cols) { ST_TAKE(v[i].x, j) ST_TAKE(v[i].y, j + 1) ST_TAKE(v[i].z, j + 2) ST_TAKE(v[i].w, j + 3) }
    }
  } else {
    for (int j = tid; j < n_cols; j += 256) ST_TAKE(row[j], j)
  }
#undef ST_TAKE
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Top2 b;
    b.v1 = __shfl_xor(t.v1, o); b.i1 = __shfl_xor(t.i1, o); b.v2 = __shfl_xor(t.v2, o);
    t = top2_merge(t, b);
  }
  if (lane == 0) sh[wave] = t;
  __syncthreads();
  Top2 a = top2_merge(top2_merge(sh[0], sh[1]), top2_merge(sh[2], sh[3]));
  if (sa.temp > 0.f) {  // uniform: second pass over the row with perturbed keys
    const int frame = sa.frames ? sa.frames[r] : sa.frame_base + r;
    const uint64_t seed = sa.seed + (sa.salt ? 0x9E3779B97F4A7C15ULL * (uint64_t)sa.salt[r] : 0ULL);
    const float inv_t = 1.0f / sa.temp;
    const float cut = sa.min_p > 0.f ? logf(sa.min_p) : -INFINITY;
    Top2 k{-INFINITY, 0x7fffffff, -INFINITY};
#define ST_KEY(V, J)                                                  \
  {                                                                   \
    const float z = ((V) - a.v1) * inv_t; /* <= 0 */                  \
    if (z >= cut) {                                                   \
      const float u = uniform01(seed, r, frame, sa.step, (J));        \
      const float key = z - logf(-logf(u));                           \
      if (key > k.v1) { k.v1 = key; k.i1 = (J); }                     \
    }                                                                 \
  }
    if (vec) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int j = tid * 4 + i * 1024;
        if (j < n_cols) { ST_KEY(v[i].x, j) ST_KEY(v[i].y, j + 1) ST_KEY(v[i].z, j + 2) ST_KEY(v[i].w, j + 3) }
      }
    } else {
      for (int j = tid; j < n_cols; j += 256) ST_KEY(row[j], j)
    }
#undef ST_KEY
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      Top2 b;
      b.v1 = __shfl_xor(k.v1, o); b.i1 = __shfl_xor(k.i1, o); b.v2 = -INFINITY;
      k = top2_merge(k, b);
    }
    if (lane == 0) sh2[wave] = k;
    __syncthreads();
    const Top2 w = top2_merge(top2_merge(sh2[0], sh2[1]), top2_merge(sh2[2], sh2[3]));
    a.i1 = w.i1;
  }
  if (tid == 0) {
    if (a.i1 < 0 || a.i1 >= n_cols) a.i1 = 0;  // all-NaN row: stay inside the tables
    ids[(long)r * ids_stride] = a.i1;
    if (sa.temp <= 0.f && margin && (margin_mask == nullptr || margin_mask[r])) {
      const float gap = a.v1 - a.v2;
      if (gap < margin[r]) {  // also remember where the slot's smallest gap occurred: frame * 64 + step (0 = slow id)
        margin[r] = gap;
        if (sa.margin_at) sa.margin_at[r] = (sa.frames ? sa.frames[r] : sa.frame_base + r) * 64 + sa.step;
      }
    }
    s_id = a.i1;
  }
  if (emb == nullptr) return;
  __syncthreads();
  const long erow = (long)s_id + emb_row_offset;
  float ss = 0.f;
  for (int d = tid * 4; d < dim; d += 256 * 4) {
    const uint2 e = *reinterpret_cast<const uint2*>(emb + erow * dim + d);
    const float4 o = make_float4(bf16_lo(e.x), bf16_hi(e.x), bf16_lo(e.y), bf16_hi(e.y));
    *reinterpret_cast<float4*>(xnext + (long)r * dim + d) = o;
    ss += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
    emit_x4(emit, r, d, dim >> 5, o.x, o.y, o.z, o.w);
  }
  emit_row_ssq(emit, r, dim, ss, sh4);
}

int launch_argmax(const float* logits, int n_rows, int n_cols, int64_t ld, int32_t* ids, int ids_stride, float* margin,
                  const int32_t* margin_mask, const void* emb, int emb_row_offset, int dim, float* xnext,
                  const EmitArgs* emit, const SampleArgs* sample, hipStream_t stream) {
  SampleArgs sa{0.f, 0.f, 0, 0, 0, nullptr, nullptr, nullptr};
  if (sample) sa = *sample;
  EmitDev e{nullptr, nullptr, nullptr, nullptr, nullptr};
  if (emit && emb) e = EmitDev{(char*)emit->x3a, emit->gamma_a, (char*)emit->x3b, emit->gamma_b, emit->ssq};
  ST_REQUIRE(logits && ids && n_rows > 0 && n_cols > 0, SMOLTTS_E_INVALID, "argmax: bad arguments");
  ST_REQUIRE(emb == nullptr || (xnext && dim % 4 == 0), SMOLTTS_E_INVALID, "argmax: bad gather arguments");
  hipLaunchKernelGGL(argmax_kernel, dim3(n_rows), dim3(256), 0, stream, logits, n_cols, (long)ld, ids, ids_stride, margin,
                     margin_mask, (const uint16_t*)emb, emb_row_offset, dim, xnext, e, sa);
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
