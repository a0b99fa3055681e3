// Row argmax / sampling (device side), shared by argmax_kernel (small_ops.hip) and the frame loop's commit kernel
// (lm_engine.hip), which picks the slow token and the last depth code itself (two launches fewer per frame).
//
// torch.argmax / mx.argmax semantics: index of the first maximal element (lm/generate.py:88-99,118-132).  Also tracks the
// top-1 / top-2 gap (parity diagnostics).  Sampling (temp > 0): exact categorical sampling from softmax(logits / temp) by
// the Gumbel-max trick with a counter-based generator, optionally restricted to tokens with p >= min_p * p_max.
#pragma once
#include "x3.h"

namespace smoltts {

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

struct ArgmaxScratch {  // LDS of one call; a workgroup that makes several calls gives each its own (no barrier in between needed)
  Top2 sh[4];
  Top2 sh2[4];
  int id;
};

// One workgroup of exactly 256 threads picks the id of logits row `row` (n_cols entries); every thread returns it.
// `r`: the row's index into margin / margin_mask / the sampling arrays.  Greedy rows update margin[r] / margin_at[r].
__device__ __forceinline__ int argmax_row(const float* row, int n_cols, bool ld_vec, int r, float* margin, const int* margin_mask,
                                          const SampleArgs& sa, ArgmaxScratch& S) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  const bool vec = (n_cols & 3) == 0 && ld_vec && n_cols <= 8 * 1024;
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
  if (lane == 0) S.sh[wave] = t;
  __syncthreads();
  Top2 a = top2_merge(top2_merge(S.sh[0], S.sh[1]), top2_merge(S.sh[2], S.sh[3]));
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
    if (lane == 0) S.sh2[wave] = k;
    __syncthreads();
    const Top2 w = top2_merge(top2_merge(S.sh2[0], S.sh2[1]), top2_merge(S.sh2[2], S.sh2[3]));
    a.i1 = w.i1;
  }
  if (a.i1 < 0 || a.i1 >= n_cols) a.i1 = 0;  // all-NaN row: stay inside the tables
  if (tid == 0) {
    if (sa.temp <= 0.f && margin && (margin_mask == nullptr || margin_mask[r])) {
      const float gap = a.v1 - a.v2;
      if (gap < margin[r]) {  // also remember where the slot's smallest gap occurred: frame * 64 + step (0 = slow id)
        margin[r] = gap;
        if (sa.margin_at) sa.margin_at[r] = (sa.frames ? sa.frames[r] : sa.frame_base + r) * 64 + sa.step;
      }
    }
  }
  return a.i1;  // the same value in every thread (merged from LDS)
}

// The picked id of one row from the head GEMM's tile candidates (SmolttsGemm3Args.cand_out_dev layout), by a whole wave: every lane returns it.
// The merge across lanes stays on the VALU (DPP inside 16-lane rows, v_permlane16/32_swap across them): six dependent
// ds_bpermute round trips per row -- what __shfl_xor compiles to -- were most of what the pick added to the launch.
template <int CTRL>
__device__ __forceinline__ Top2 top2_dpp(Top2 t) {
  Top2 b;
  b.v1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t.v1), CTRL, 0xF, 0xF, true));
  b.i1 = __builtin_amdgcn_update_dpp(0, t.i1, CTRL, 0xF, 0xF, true);
  b.v2 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t.v2), CTRL, 0xF, 0xF, true));
  return top2_merge(t, b);
}
template <bool ROWS32>
__device__ __forceinline__ Top2 top2_swap(Top2 t, int lane) {  // partner = lane ^ 16 (ROWS32: lane ^ 32)
  const unsigned a[3] = {__float_as_uint(t.v1), (unsigned)t.i1, __float_as_uint(t.v2)};
  unsigned o[3];
  const bool upper = (lane & (ROWS32 ? 32 : 16)) != 0;  // after swap(x, x): the lower lane's partner value is result 1, the upper's result 0
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (ROWS32) {
      const auto r = __builtin_amdgcn_permlane32_swap(a[i], a[i], false, false);
      o[i] = upper ? r[0] : r[1];
    } else {
      const auto r = __builtin_amdgcn_permlane16_swap(a[i], a[i], false, false);
      o[i] = upper ? r[0] : r[1];
    }
  }
  return top2_merge(t, Top2{__uint_as_float(o[0]), (int)o[1], __uint_as_float(o[2])});
}
__device__ __forceinline__ Top2 cand_pick_wave(const float* cand, int tiles, int lane) {
  Top2 t{-INFINITY, 0x7fffffff, -INFINITY};
  for (int j = lane; j < tiles; j += 64) {
    const float4 c = *reinterpret_cast<const float4*>(cand + (size_t)j * 4);
    t = top2_merge(t, Top2{c.x, __float_as_int(c.y), c.z});
  }
  t = top2_dpp<0xB1>(t);   // lane ^ 1
  t = top2_dpp<0x4E>(t);   // lane ^ 2
  t = top2_dpp<0x141>(t);  // the other quad of the half row
  t = top2_dpp<0x140>(t);  // the other half row
  t = top2_swap<false>(t, lane);
  t = top2_swap<true>(t, lane);
  return t;
}

// Projections of a depth-transformer input row that are known before the row is: row e of the fast embedding table always
// enters layer 0 as RMSNorm(E[e]) -> wqkv, so q | k | v (before RoPE) are a table lookup [rows][nqkv] instead of a GEMM launch
// (smoltts_engine_build_fast_qkv).  The kernel that picks the code gathers them, applies RoPE for the row's position and writes
// q and the depth cache rows exactly as the wqkv GEMM's epilogue would (gemm3.hip EPI_QKV_ROPE).
struct QkvGather {
  const float* table;   // [emb rows][nqkv] fp32, pre-RoPE; nullptr = off
  const float* rope;    // fp32 [pos][32][2]
  float* q_out;         // [rows][n_q_heads * 64]
  float* kc;            // depth cache of layer 0: [slot][kv head][cache_len][64]
  float* vc;
  int n_q_heads, n_kv_heads, cache_len, pos;
};

// The row's q | k | v: loaded by `qkv_gather_load` (up to QG_MAX float4 per thread, all requested at once, together with the RoPE
// rows they need -- a loop over the row would be one dependent L2 round trip per iteration), finished by `qkv_gather_store`.
// RoPE of the two (even, odd) pairs of a float4 of table values: rounded products, rounded sums (no contraction) -- the one
// place that defines these bits, shared by the picking kernel's gather (small_ops.hip) and by the pick inside the attention + wo
// launch (gemm3.hip attn_wo_kernel), which must publish the same q / K rows.
__device__ __forceinline__ float4 rope_gathered4(float4 v, float4 cs) {
#pragma clang fp contract(off)
  const float o0 = v.x * cs.x - v.y * cs.y, o1 = v.y * cs.x + v.x * cs.y;
  const float o2 = v.z * cs.z - v.w * cs.w, o3 = v.w * cs.z + v.z * cs.w;
  return make_float4(o0, o1, o2, o3);
}

constexpr int QG_MAX = 2;  // 256 threads x 2 x 4 floats = rows of up to 2048 values (150m: 1280); longer rows take more rounds
struct QkvRegs { float4 v[QG_MAX], cs[QG_MAX]; };

__device__ __forceinline__ void qkv_gather_load(const QkvGather& g, long erow, int base, QkvRegs& o) {
  const int qd = g.n_q_heads * 64, kd = g.n_kv_heads * 64, nqkv = qd + 2 * kd;
  const float* trow = g.table + erow * nqkv;
#pragma unroll
  for (int i = 0; i < QG_MAX; ++i) {
    const int n0 = base + (threadIdx.x + i * 256) * 4;
    o.v[i] = n0 < nqkv ? *reinterpret_cast<const float4*>(trow + n0) : make_float4(0.f, 0.f, 0.f, 0.f);
    o.cs[i] = n0 < qd + kd ? *reinterpret_cast<const float4*>(g.rope + ((long)g.pos * 32 + ((n0 & 63) >> 1)) * 2) : make_float4(1.f, 0.f, 1.f, 0.f);
  }
}

__device__ __forceinline__ void qkv_gather_store(const QkvGather& g, int r, int base, const QkvRegs& in) {
  const int qd = g.n_q_heads * 64, kd = g.n_kv_heads * 64, nqkv = qd + 2 * kd;
#pragma unroll
  for (int i = 0; i < QG_MAX; ++i) {
    const int n0 = base + (threadIdx.x + i * 256) * 4;
    if (n0 >= nqkv) continue;
    float4 v = in.v[i];
    if (n0 < qd + kd) v = rope_gathered4(v, in.cs[i]);
    if (n0 < qd) {
      *reinterpret_cast<float4*>(g.q_out + (long)r * qd + n0) = v;
    } else {
      const int nn = n0 - qd;
      float* dst = nn < kd ? g.kc : g.vc;
      const int h = (nn < kd ? nn : nn - kd) >> 6, d = nn & 63;
      *reinterpret_cast<float4*>(dst + (((long)r * g.n_kv_heads + h) * g.cache_len + g.pos) * 64 + d) = v;
    }
  }
}

}  // namespace smoltts
