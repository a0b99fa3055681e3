// Device-side pieces shared by the fp32-MFMA GEMMs (gemm.hip) and the bf16x3-split many-row GEMM (gemm_b3.hip).
#pragma once
#include "x3.h"

namespace smoltts {

struct GemmDev {
  const char* w;
  const char* w3;  // optional: the same matrix as bf16x3 piece tiles ("W3": per 16-row x 32-k tile three 1 KiB A-fragment blocks hi|mid|lo)
  const float* x;
  long ldx, x_bstride;
  int rows_per_batch;
  int M, N, K;
  const float* gamma;
  float eps;
  const float* bias;
  const float* scale;
  const float* resid;
  float* out;
  long ldo, o_bstride;
  float* raw_out;  // optional copy of the un-activated result (same row stride, own batch stride)
  long raw_bstride;
  int elu_out;     // apply ELU to what goes to `out`
  int pro_elu;     // gemm_b3: apply ELU to X as it is staged (the producer stored the raw tensor only)
  const float* ln_w;  // LayerNorm prologue (conv_xs Linear mode, the skinny kernel): weight / bias [K]; eps in `eps`; nullptr = none
  const float* ln_b;
  long ldr, r_bstride;
  const float* rope;
  const int* row_pos;
  const int* row_slot;
  float* kc;
  float* vc;
  char* kc3;  // optional (EPI_QKV_ROPE): the K / V rows also as bf16x3 piece caches (include/smoltts_hip.h, smoltts_k_attention_rows3)
  char* vc3;
  int n_q_heads, n_kv_heads, cache_len;
  unsigned long long* stamps;  // diagnostic build aid (SMOLTTS debug API); nullptr in production
  int grid_rb, grid_cb;        // gemm_b3: row blocks x column blocks of the (1-D, XCD-aware) launch
  float* splitk_ws;            // gemm_b3, optional: workspace of splitk_cap floats for split-K partial sums [split][M][N]
  long splitk_cap;
  int ksplit;                  // set by the launcher: K split over this many workgroups per tile (1 = no split)
  int taps, cpt;               // set by the launcher: conv windows (K = taps * ldx, overlapping rows): 32-k chunks are visited channel
                               // slice by channel slice, all taps of a slice back to back; cpt = chunks per tap.  taps = 1: in order
  int b3_products;             // 3 = the bf16x3 kernels form three products per operand pair instead of six (mfma_b3<3>); anything else: six
  int xcd_cols;                // set by the launcher: 1 = the XCDs are dealt column blocks (W leaves memory once), 0 = row blocks (X does)
};

__device__ __forceinline__ long row_off(int m, int rpb, long ld, long bstride) {
  if (rpb <= 0) return (long)m * ld;
  int b = m / rpb;
  return (long)b * bstride + (long)(m - b * rpb) * ld;
}

// Workgroup barrier that orders LDS traffic only: global loads already in flight (register prefetches of the next weight group
// / tile) stay in flight across it -- __syncthreads() is a full fence and drains them (s_waitcnt vmcnt(0)).
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }
// ELU of the many-row epilogues, with the hardware exponential: exp(x) - 1 carries an absolute error of ~1 ulp of 1 (6e-8)
// where expm1f is relatively exact -- the size of the fp32 rounding of the O(1) activations around it, at a fifth of the
// instructions (a k1 conv's workgroup applies it to 64 values per lane; seanet.hip and the ELU prologues do the same).
__device__ __forceinline__ float elu_rows(float x) { return x > 0.f ? x : __expf(x) - 1.0f; }
__device__ __forceinline__ float silu1(float x) { return x / (1.f + expf(-x)); }
__device__ __forceinline__ float gelu1(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

// K / V values [pos][d .. d + 4) of (slot, kv head h) into the bf16x3 piece caches (layouts: include/smoltts_hip.h at
// smoltts_k_attention_rows3).  K: the X3 row format over positions; V: per 32-position block and 16-dim tile, lane 16 q + (dim % 16)
// holds 8 positions in the order the score MFMA leaves a lane its probabilities.
__device__ __forceinline__ void kv3_store(char* kc3, char* vc3, int n_kv_heads, int cache_len, int slot, int h, int pos, int d, bool is_v,
                                          const float4 o) {
  const long per_head = (long)((cache_len + 31) & ~31) * 384;
  const long head = ((long)slot * n_kv_heads + h) * per_head;
  if (!is_v) {
    x3_emit4(kc3 + head, pos, d, 2, o.x, o.y, o.z, o.w);
    return;
  }
  uint32_t h0, m0, l0, h1, m1, l1;
  split3_pair(o.x, o.y, h0, m0, l0);
  split3_pair(o.z, o.w, h1, m1, l1);
  const int kk = pos & 31, key = kk & 15, j = (key & 3) + 4 * (kk >> 4);
  unsigned short* dst = reinterpret_cast<unsigned short*>(vc3 + head + (long)((pos >> 5) * 4 + (d >> 4)) * 3072 + ((key >> 2) * 16 + (d & 15)) * 16 + j * 2);
  // dims d .. d + 3 are the lanes r .. r + 3 of the fragment: 16 bytes (8 shorts) apart; pieces 1 KiB (512 shorts) apart
  dst[0] = (unsigned short)h0; dst[8] = (unsigned short)(h0 >> 16); dst[16] = (unsigned short)h1; dst[24] = (unsigned short)(h1 >> 16);
  dst[512] = (unsigned short)m0; dst[520] = (unsigned short)(m0 >> 16); dst[528] = (unsigned short)m1; dst[536] = (unsigned short)(m1 >> 16);
  dst[1024] = (unsigned short)l0; dst[1032] = (unsigned short)(l0 >> 16); dst[1040] = (unsigned short)l1; dst[1048] = (unsigned short)(l1 >> 16);
}

// Epilogue of the many-row kernels, straight from the accumulators: the lane holds acc = X.W^T for out[m][n0 .. n0+4)
// (`orow` = offset of row m in `out`).  Bias, then by EPI: GELU | layer-scale + residual | RoPE + q / KV-cache scatter |
// residual and/or raw copy + optional ELU.
template <int EPI>
__device__ __forceinline__ void rows_epilogue(const GemmDev& p, int m, long orow, int n0, float v[4]) {
  if (n0 >= p.N) return;
  if (p.bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (n0 + i < p.N) v[i] += p.bias[n0 + i];
  }
  if (n0 + 4 > p.N) {  // N < 4: scalar tail (the 1-channel output conv)
    for (int i = 0; i < 4 && n0 + i < p.N; ++i) p.out[orow + n0 + i] = p.elu_out ? elu_rows(v[i]) : v[i];
    return;
  }
  if (EPI == SMOLTTS_EPI_GELU) {
    *reinterpret_cast<float4*>(p.out + orow + n0) = make_float4(gelu1(v[0]), gelu1(v[1]), gelu1(v[2]), gelu1(v[3]));
    return;
  }
  if (EPI == SMOLTTS_EPI_SCALE_RESID) {
    const float4 rr = *reinterpret_cast<const float4*>(p.resid + row_off(m, p.rows_per_batch, p.ldr, p.r_bstride) + n0);
    const float4 sc = *reinterpret_cast<const float4*>(p.scale + n0);
    *reinterpret_cast<float4*>(p.out + orow + n0) =
        make_float4(rr.x + sc.x * v[0], rr.y + sc.y * v[1], rr.z + sc.z * v[2], rr.w + sc.w * v[3]);
    return;
  }
  if (EPI == SMOLTTS_EPI_QKV_ROPE) {
    const int pos = p.row_pos[m], slot = p.row_slot[m];
    const int qd = p.n_q_heads * 64, kd = p.n_kv_heads * 64;
    if (n0 < qd + kd) {
      const float4 cs = *reinterpret_cast<const float4*>(p.rope + ((long)pos * 32 + ((n0 & 63) >> 1)) * 2);
      const float o0 = v[0] * cs.x - v[1] * cs.y, o1 = v[1] * cs.x + v[0] * cs.y;
      const float o2 = v[2] * cs.z - v[3] * cs.w, o3 = v[3] * cs.z + v[2] * cs.w;
      v[0] = o0; v[1] = o1; v[2] = o2; v[3] = o3;
    }
    const float4 o = make_float4(v[0], v[1], v[2], v[3]);
    if (n0 < qd) {
      *reinterpret_cast<float4*>(p.out + orow + n0) = o;
    } else if (pos >= 0 && pos < p.cache_len) {
      const int nn = n0 - qd;
      float* base = nn < kd ? p.kc : p.vc;
      const int h = (nn < kd ? nn : nn - kd) >> 6, d = nn & 63;
      *reinterpret_cast<float4*>(base + (((long)slot * p.n_kv_heads + h) * p.cache_len + pos) * 64 + d) = o;
      if (p.kc3) kv3_store(p.kc3, p.vc3, p.n_kv_heads, p.cache_len, slot, h, pos, d, nn >= kd, o);
    }
    return;
  }
  if (EPI == SMOLTTS_EPI_RESID) {
    const float4 rr = *reinterpret_cast<const float4*>(p.resid + row_off(m, p.rows_per_batch, p.ldr, p.r_bstride) + n0);
    v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
  } else if (p.raw_out) {
    *reinterpret_cast<float4*>(p.raw_out + row_off(m, p.rows_per_batch, p.ldo, p.raw_bstride) + n0) = make_float4(v[0], v[1], v[2], v[3]);
  }
  if (p.elu_out) { v[0] = elu_rows(v[0]); v[1] = elu_rows(v[1]); v[2] = elu_rows(v[2]); v[3] = elu_rows(v[3]); }
  *reinterpret_cast<float4*>(p.out + orow + n0) = make_float4(v[0], v[1], v[2], v[3]);
}

// The residual epilogue with the residual values already in registers (a workgroup with many output tiles per lane loads
// them all before the first store: one memory round trip instead of one per tile -- `out` may alias nothing the loads read,
// but the compiler cannot know): same arithmetic as rows_epilogue<SMOLTTS_EPI_RESID>.
__device__ __forceinline__ void rows_epilogue_resid(const GemmDev& p, long orow, int n0, float v[4], const float4 rr) {
  if (p.bias) {
    const float4 bb = *reinterpret_cast<const float4*>(p.bias + n0);
    v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
  }
  v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
  if (p.elu_out) { v[0] = elu_rows(v[0]); v[1] = elu_rows(v[1]); v[2] = elu_rows(v[2]); v[3] = elu_rows(v[3]); }
  *reinterpret_cast<float4*>(p.out + orow + n0) = make_float4(v[0], v[1], v[2], v[3]);
}

// 8 consecutive fp32 values -> the three bf16x8 pieces (hi + mid + lo == the value, exactly) as packed 16-byte fragments
__device__ __forceinline__ void split3x8(const float4 a, const float4 b, uint4& h, uint4& m, uint4& l) {
  split3_pair(a.x, a.y, h.x, m.x, l.x);
  split3_pair(a.z, a.w, h.y, m.y, l.y);
  split3_pair(b.x, b.y, h.z, m.z, l.z);
  split3_pair(b.z, b.w, h.w, m.w, l.w);
}

// acc += W . x over one 32-k chunk with both operands split in three bf16 pieces: the six products whose weight is
// >= 2^-16 of the leading one (dropped: mid*lo, lo*mid, lo*lo, each <= 2^-24 relative: below the rounding of the fp32
// accumulation itself), smallest first.  Products of bf16 pieces are exact in the fp32 accumulator.
// NP = 6 (default): fp32-grade.  NP = 3 (a session option, SMOLTTS_MIMI_OPT_PRODUCTS: DESIGN.md 4.7): only the products of weight
// >= 2^-8 of the leading one (hi*hi, mid*hi, hi*mid); the dropped ones are each <= 2^-16 relative, the lo pieces are never loaded.
template <int NP = 6>
__device__ __forceinline__ f32x4 mfma_b3(const uint4 w[3], const uint4 x[3], f32x4 acc) {
  static_assert(NP == 3 || NP == 6, "bf16x3 products: 3 or 6");
#define ST_MF(WP, XP) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w[WP]), __builtin_bit_cast(bf16x8_t, x[XP]), acc, 0, 0, 0)
  if constexpr (NP == 6) { ST_MF(1, 1); ST_MF(2, 0); ST_MF(0, 2); }
  ST_MF(1, 0); ST_MF(0, 1); ST_MF(0, 0);
#undef ST_MF
  return acc;
}

// out[m][n0..n0+4) = epilogue( sum over the splits, in split order )
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmDev p) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;  // one float4 of the [M][N] result
  const int n4 = p.N >> 2;
  if (i >= (long)p.M * n4) return;
  const int m = (int)(i / n4), n0 = (int)(i - (long)m * n4) * 4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < p.ksplit; ++s) {
    const float4 t = *reinterpret_cast<const float4*>(p.splitk_ws + ((long)s * p.M + m) * p.N + n0);
    v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
  }
  rows_epilogue<EPI>(p, m, row_off(m, p.rows_per_batch, p.ldo, p.o_bstride), n0, v);
}

int launch_gemm_b3(const GemmDev& d, int epilogue, hipStream_t stream);  // gemm_b3.hip
bool gemm_b3_applies(int M, int N, int K, int epilogue);
int launch_conv_xs(const GemmDev& d, int epilogue, hipStream_t stream);  // conv_xs.hip: conv windows of <= 256 channels / Linears of K <= 512, X stationary in LDS
bool conv_xs_applies(const GemmDev& d, int epilogue);
int launch_conv_ks(const GemmDev& d, int epilogue, hipStream_t stream);  // conv_ks.hip: conv windows over > 256 channels, X through LDS in 128-channel slices
bool conv_ks_applies(const GemmDev& d, int epilogue);

}  // namespace smoltts
