// Single-query GQA attention over a slot's KV-cache prefix (decode step, and prefill row by row).
//
// Reference: F.scaled_dot_product_attention(q, k, v, is_causal=True) with repeat_interleave GQA
// (modeling/model/rq_transformer.py:554-568); at decode time
// mx.fast.scaled_dot_product_attention over KVCache.update_and_fetch
// (mlx_inference/.../lm/rq_transformer.py:281-295, lm/cache.py:12-22); Mimi:
// codec/transformer.py:75-96.  softmax(q.K^T / 8) . V in fp32, head_dim 64.
//
// One workgroup (4 waves) per (query row, kv head); the G query heads of the group share every
// K/V row read.  The cache is fp32 [slot][kv head][cache_len][64], so a key is one 256-byte line
// pair: 16 lanes x float4 read one key, a wave reads 4 keys per instruction (1 KiB, coalesced).
// Two passes with the scores parked in LDS (exact softmax: max, exp, sum in a fixed reduction
// order => deterministic), then P.V with the same lane mapping and a fixed-order cross-wave sum.
#include "common.h"

namespace smoltts {

struct AttnDev {
  const float* q;
  const float* kc;
  const float* vc;
  const int* row_pos;
  const int* row_slot;
  float* out;
  int n_q_heads, n_kv_heads, cache_len, window, score_cap;
};

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int G>
__global__ __launch_bounds__(256) void attn_kernel(AttnDev p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* scores = smem;                      // [G][score_cap]
  float* part = smem + G * p.score_cap;      // [4 waves][G][64]
  float* stat = part + 4 * G * 64;           // [4][G] scratch, then [G] max, [G] sum
  const int row = blockIdx.x, h = blockIdx.y;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int kk = lane >> 4, dl = lane & 15;
  const int pos = p.row_pos[row], slot = p.row_slot[row];
  const int HD = p.n_q_heads * 64;
  float* orow = p.out + (long)row * HD + (h * G) * 64;
  if (pos < 0 || pos >= p.cache_len) {  // nothing cached for this row: defined output, no OOB
    for (int i = tid; i < G * 64; i += 256) orow[i] = 0.f;
    return;
  }
  const int j_lo = (p.window > 0 && pos + 1 > p.window) ? pos + 1 - p.window : 0;
  const int L = pos + 1 - j_lo;
  const long cbase = (((long)slot * p.n_kv_heads + h) * p.cache_len + j_lo) * 64;
  const float* K = p.kc + cbase;
  const float* V = p.vc + cbase;

  float4 qv[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float4 t = *reinterpret_cast<const float4*>(p.q + (long)row * HD + (h * G + g) * 64 + dl * 4);
    qv[g] = make_float4(t.x * 0.125f, t.y * 0.125f, t.z * 0.125f, t.w * 0.125f);
  }

  // ---- pass 1: scores
  for (int j0 = wave * 4; j0 < L; j0 += 16) {
    const int j = j0 + kk;
    const bool valid = j < L;
    float4 kv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) kv = *reinterpret_cast<const float4*>(K + (long)j * 64 + dl * 4);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float s = qv[g].x * kv.x;
      s = fmaf(qv[g].y, kv.y, s);
      s = fmaf(qv[g].z, kv.z, s);
      s = fmaf(qv[g].w, kv.w, s);
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      s += __shfl_xor(s, 4);
      s += __shfl_xor(s, 8);
      if (valid && dl == 0) scores[g * p.score_cap + j] = s;
    }
  }
  __syncthreads();
  // ---- max
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float m = -INFINITY;
    for (int j = tid; j < L; j += 256) m = fmaxf(m, scores[g * p.score_cap + j]);
    m = wave_max(m);
    if (lane == 0) stat[wave * G + g] = m;
  }
  __syncthreads();
  float gmax[G];
#pragma unroll
  for (int g = 0; g < G; ++g)
    gmax[g] = fmaxf(fmaxf(stat[0 * G + g], stat[1 * G + g]), fmaxf(stat[2 * G + g], stat[3 * G + g]));
  __syncthreads();
  // ---- exp + sum
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float s = 0.f;
    for (int j = tid; j < L; j += 256) {
      const float e = expf(scores[g * p.score_cap + j] - gmax[g]);
      scores[g * p.score_cap + j] = e;
      s += e;
    }
    s = wave_sum(s);
    if (lane == 0) stat[wave * G + g] = s;
  }
  __syncthreads();
  float inv[G];
#pragma unroll
  for (int g = 0; g < G; ++g)
    inv[g] = 1.0f / (((stat[0 * G + g] + stat[1 * G + g]) + stat[2 * G + g]) + stat[3 * G + g]);

  // ---- pass 2: P.V
  float4 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j0 = wave * 4; j0 < L; j0 += 16) {
    const int j = j0 + kk;
    if (j < L) {
      const float4 vv = *reinterpret_cast<const float4*>(V + (long)j * 64 + dl * 4);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float pj = scores[g * p.score_cap + j];
        acc[g].x = fmaf(pj, vv.x, acc[g].x);
        acc[g].y = fmaf(pj, vv.y, acc[g].y);
        acc[g].z = fmaf(pj, vv.z, acc[g].z);
        acc[g].w = fmaf(pj, vv.w, acc[g].w);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float4 a = acc[g];
    a.x += __shfl_xor(a.x, 16); a.y += __shfl_xor(a.y, 16); a.z += __shfl_xor(a.z, 16); a.w += __shfl_xor(a.w, 16);
    a.x += __shfl_xor(a.x, 32); a.y += __shfl_xor(a.y, 32); a.z += __shfl_xor(a.z, 32); a.w += __shfl_xor(a.w, 32);
    if (kk == 0) *reinterpret_cast<float4*>(part + (wave * G + g) * 64 + dl * 4) = a;
  }
  __syncthreads();
  for (int i = tid; i < G * 64; i += 256) {
    const int g = i >> 6, d = i & 63;
    float s = part[(0 * G + g) * 64 + d];
    s += part[(1 * G + g) * 64 + d];
    s += part[(2 * G + g) * 64 + d];
    s += part[(3 * G + g) * 64 + d];
    float ig = inv[0];
#pragma unroll
    for (int gg = 1; gg < G; ++gg) ig = (g == gg) ? inv[gg] : ig;
    orow[g * 64 + d] = s * ig;
  }
}

int launch_attention(const float* q, const float* kc, const float* vc, const int32_t* row_pos,
                     const int32_t* row_slot, int n_rows, int n_q_heads, int n_kv_heads,
                     int cache_len, int window, float* out, hipStream_t stream) {
  ST_REQUIRE(q && kc && vc && row_pos && row_slot && out, SMOLTTS_E_INVALID, "attention: null pointer");
  ST_REQUIRE(n_rows > 0 && n_kv_heads > 0 && n_q_heads % n_kv_heads == 0 && cache_len > 0, SMOLTTS_E_INVALID,
             "attention: bad shape rows=%d q_heads=%d kv_heads=%d cache_len=%d", n_rows, n_q_heads, n_kv_heads, cache_len);
  const int G = n_q_heads / n_kv_heads;
  AttnDev d{q, kc, vc, row_pos, row_slot, out, n_q_heads, n_kv_heads, cache_len, window, 0};
  d.score_cap = (window > 0 && window < cache_len) ? window : cache_len;
  d.score_cap = (d.score_cap + 3) & ~3;
  const size_t lds = ((size_t)G * d.score_cap + 4 * G * 64 + 4 * G + 16) * sizeof(float);
  ST_REQUIRE(lds <= 150 * 1024, SMOLTTS_E_CAPACITY, "attention: %zu bytes of LDS needed for %d keys x %d heads", lds,
             d.score_cap, G);
  ST_REQUIRE(n_rows <= 0x7fffffff && n_kv_heads <= 65535, SMOLTTS_E_INVALID, "attention: grid too large");
  const dim3 grid(n_rows, n_kv_heads);
#define ST_ATTN(GG)                                                                           \
  case GG:                                                                                    \
    if (lds > 64 * 1024)                                                                      \
      ST_CHECK_HIP(hipFuncSetAttribute((const void*)attn_kernel<GG>,                          \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(attn_kernel<GG>, grid, dim3(256), lds, stream, d);                     \
    break;
  switch (G) {
    ST_ATTN(1)
    ST_ATTN(2)
    ST_ATTN(3)
    ST_ATTN(4)
    default:
      set_error("attention: GQA group size %d not instantiated (1..4)", G);
      return SMOLTTS_E_INVALID;
  }
#undef ST_ATTN
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

}  // namespace smoltts
