// Causal conv / ConvTranspose GEMM with the activation window stationary in LDS (the SEANet layers with <= 256 input channels):
//   out[m][n] = bias[n] + sum_k X[m][k] W[n][k],  row m's K = taps * C_in values are the `taps` consecutive channel-last rows
//   starting at row m of its slot's halo-prefixed buffer (conv k: taps = k; ConvTranspose stride s, kernel 2s: taps = 2).
//
// Reference ops: Conv1d / ConvTranspose1d of mlx_inference/src/smoltts_mlx/codec/conv.py:68-220 as used by the SEANet decoder
// (codec/seanet.py:8-49,99-139), fp32.
//
// Why a second many-row kernel beside gemm_b3.hip: that one stages a 32-k chunk of X *and* of W in LDS per step, two barriers
// per chunk, and reaches 30-37 % of the bf16x3 matrix-core rate on these shapes.  Here (the layout of seanet_last.hip's
// ConvTranspose phase, which runs at ~80 %): the 64 + taps - 1 input rows a tile of 64 output rows touches are split into
// bf16x3 pieces ONCE (a window row is shared by `taps` output rows: a tap is a row shift in LDS), every wave owns NTW of the
// N / 16 column tiles for all four row tiles and streams its W3 fragments from L2 straight into registers, one chunk ahead
// (each fragment is used by exactly one wave of the workgroup), and the K loop has no barrier at all.
// LDS: 3 pieces x C_in / 8 groups x 80 rows x 16 B = 120 KB at C_in = 256: one 8-wave workgroup per CU.
//
// The same kernel with one "tap" is a plain Linear over rows of K <= 512 values (the decoder transformer's wqkv, wo and fc1 at
// M = 2048: transformer.py:34-96): tiles of 32 rows (96 KB of pieces), the N / 16 column tiles split over 4 workgroups per row
// tile so that 256 workgroups fill the chip; epilogues of gemm_dev.h (RoPE + q / KV scatter, layer scale + residual, GELU).
#include <stdlib.h>

#include "gemm_dev.h"

namespace smoltts {

namespace {

constexpr int XS_RA_CONV = 80;  // conv: 64 output rows per tile; LDS rows per plane >= 64 + taps - 1, % 16 == 0
constexpr int XS_MAX_CIN = 256, XS_MAX_K_LINEAR = 512;

__device__ __forceinline__ float elu_hw_xs(float x) { return x > 0.f ? x : __expf(x) - 1.0f; }

#ifdef SMOLTTS_DBG_XS_STAMPS  // (diagnostic variant, tools/stamps_xs.py: cycles per phase as seen by thread 0 of every workgroup)
__device__ unsigned long long g_xs_stamps[8 * 8];  // [epilogue][phase 0..5, 6 = workgroups, 7 unused]
#define XS_STAMP(I) { const long long now_ = clock64(); st_sum[I] += now_ - st_last; st_last = now_; }
#else
#define XS_STAMP(I)
#endif

// MT 16-row tiles of output rows per workgroup, NTW column tiles per wave, RA LDS rows per plane, EPI the epilogue (gemm_dev.h)
template <int NTW, int MT, int RA, int EPI, int NP>
__global__ __launch_bounds__(512) void conv_xs_kernel(GemmDev p) {
  constexpr int XS_S = 16 * MT, XS_MT = MT, XS_RA = RA;
  extern __shared__ __attribute__((aligned(16))) uint4 xp[];  // [piece][C_in / 8 groups = (chunk, q)][XS_RA rows]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int b = blockIdx.y, row0 = blockIdx.x * XS_S, T = p.rows_per_batch > 0 ? p.rows_per_batch : p.M;
  const int G = p.cpt * 4, nrows = XS_S + p.taps - 1, nchunks = p.K >> 5;  // G 8-value groups per window row
  // blockIdx.z: the part of the column tiles (N split over workgroups), or -- ksplit > 1 -- the part of K (partial sums out)
  const int kz = p.ksplit > 1 ? (int)blockIdx.z : 0;
  const int nt0 = (p.ksplit > 1 ? 0 : (int)blockIdx.z * 8 * NTW) + wave * NTW;  // the wave's first column tile
  const int nloc = p.ksplit > 1 ? p.cpt : nchunks;                               // chunks this workgroup runs over
  const float* xb = p.x + (long)b * p.x_bstride + (long)row0 * p.ldx + (long)kz * p.cpt * 32;
#ifdef SMOLTTS_DBG_XS_STAMPS
  long long st_sum[6] = {0, 0, 0, 0, 0, 0}, st_last = clock64();
#endif

  // ---- the tile's window rows -> pieces; 8 consecutive lanes take 8 consecutive rows of one 8-channel group (128 contiguous
  //      LDS bytes per ds_write_b128 lane group)
  if (p.ln_w != nullptr) {
    // LayerNorm prologue (nn.LayerNorm over the K values of a row, codec/transformer.py:113-114; Linear mode, 32 rows x 512 values =
    // 4 items per thread): the tile is loaded once into registers; mean, then the variance about it (two passes, as the
    // stand-alone kernel does), through per-(row, group) partial sums in LDS; (x - mean) * rstd * w + b goes into the pieces.
    float4 va[4][2];
    float* psum = reinterpret_cast<float*>(xp + 3 * G * XS_RA);  // [32 rows][64 groups], then mean[32], rstd[32]
    float* stat = psum + 32 * 64;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + 512 * it, i = idx / (8 * G) * 8 + (idx & 7), g8 = (idx >> 3) % G;
      va[it][0] = va[it][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + i < T) {
        const float* src = xb + (long)i * p.ldx + g8 * 8;
        va[it][0] = *reinterpret_cast<const float4*>(src);
        va[it][1] = *reinterpret_cast<const float4*>(src + 4);
      }
      const float4 a = va[it][0], c = va[it][1];
      psum[i * 64 + g8] = ((a.x + a.y) + (a.z + a.w)) + ((c.x + c.y) + (c.z + c.w));
    }
    lds_barrier();
    const int srow = tid >> 4, l16 = tid & 15;  // 16 lanes per row
    {
      const float4 t = *reinterpret_cast<const float4*>(psum + srow * 64 + l16 * 4);
      float sm = (t.x + t.y) + (t.z + t.w);
      sm += __shfl_xor(sm, 8); sm += __shfl_xor(sm, 4); sm += __shfl_xor(sm, 2); sm += __shfl_xor(sm, 1);
      if (l16 == 0) stat[srow] = sm / (float)p.K;
    }
    lds_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + 512 * it, i = idx / (8 * G) * 8 + (idx & 7), g8 = (idx >> 3) % G;
      const float mu = stat[i];
      const float4 a = va[it][0], c = va[it][1];
      float v = 0.f;
      v = fmaf(a.x - mu, a.x - mu, v); v = fmaf(a.y - mu, a.y - mu, v); v = fmaf(a.z - mu, a.z - mu, v); v = fmaf(a.w - mu, a.w - mu, v);
      v = fmaf(c.x - mu, c.x - mu, v); v = fmaf(c.y - mu, c.y - mu, v); v = fmaf(c.z - mu, c.z - mu, v); v = fmaf(c.w - mu, c.w - mu, v);
      psum[i * 64 + g8] = v;
    }
    lds_barrier();
    {
      const float4 t = *reinterpret_cast<const float4*>(psum + srow * 64 + l16 * 4);
      float sm = (t.x + t.y) + (t.z + t.w);
      sm += __shfl_xor(sm, 8); sm += __shfl_xor(sm, 4); sm += __shfl_xor(sm, 2); sm += __shfl_xor(sm, 1);
      if (l16 == 0) stat[32 + srow] = 1.0f / sqrtf(sm / (float)p.K + p.eps);
    }
    lds_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + 512 * it, i = idx / (8 * G) * 8 + (idx & 7), g8 = (idx >> 3) % G;
      const float mu = stat[i], rs = stat[32 + i];
      const float4 w0 = *reinterpret_cast<const float4*>(p.ln_w + g8 * 8), w1 = *reinterpret_cast<const float4*>(p.ln_w + g8 * 8 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(p.ln_b + g8 * 8), b1 = *reinterpret_cast<const float4*>(p.ln_b + g8 * 8 + 4);
      float4 a = va[it][0], c = va[it][1];
      a = make_float4((a.x - mu) * rs * w0.x + b0.x, (a.y - mu) * rs * w0.y + b0.y, (a.z - mu) * rs * w0.z + b0.z, (a.w - mu) * rs * w0.w + b0.w);
      c = make_float4((c.x - mu) * rs * w1.x + b1.x, (c.y - mu) * rs * w1.y + b1.y, (c.z - mu) * rs * w1.z + b1.z, (c.w - mu) * rs * w1.w + b1.w);
      uint4 h, m, l;
      split3x8(a, c, h, m, l);
      const int slot = g8 * XS_RA + i;
      xp[slot] = h;
      xp[slot + G * XS_RA] = m;
      xp[slot + 2 * G * XS_RA] = l;
    }
  } else
  for (int idx = tid; idx < ((nrows + 7) & ~7) * G; idx += 512) {
    const int i = idx / (8 * G) * 8 + (idx & 7), g8 = (idx >> 3) % G;
    if (i >= nrows) continue;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
    if (row0 + i < T + p.taps - 1) {  // the slot's buffer has T + taps - 1 rows from its x pointer
      const float* src = xb + (long)i * p.ldx + g8 * 8;
      a = *reinterpret_cast<const float4*>(src);
      c = *reinterpret_cast<const float4*>(src + 4);
    }
    if (p.pro_elu) {
      a = make_float4(elu_hw_xs(a.x), elu_hw_xs(a.y), elu_hw_xs(a.z), elu_hw_xs(a.w));
      c = make_float4(elu_hw_xs(c.x), elu_hw_xs(c.y), elu_hw_xs(c.z), elu_hw_xs(c.w));
    }
    uint4 h, m, l;
    split3x8(a, c, h, m, l);
    const int slot = g8 * XS_RA + i;
    xp[slot] = h;
    xp[slot + G * XS_RA] = m;
    xp[slot + 2 * G * XS_RA] = l;
  }

  XS_STAMP(0)
  f32x4 acc[XS_MT][NTW];
#pragma unroll
  for (int mt = 0; mt < XS_MT; ++mt)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // W3 tile (column tile nt, chunk kc) at (nt * nchunks + kc) * 3072: the wave's tiles are NTW * nchunks consecutive tiles
  const char* wtile = p.w3 + ((size_t)nt0 * nchunks + (size_t)kz * p.cpt) * 3072;
  const int wlane = lane * 16;
  uint4 wq[2][NTW][3];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) wq[0][t][pc] = *reinterpret_cast<const uint4*>(wtile + (size_t)(t * nchunks) * 3072 + pc * 1024 + wlane);
  lds_barrier();  // (the W3 loads above stay in flight)
  XS_STAMP(1)

  // (every workgroup streams the same W3 tiles in the same order; starting each tile's K loop at a different chunk, so that
  // they do not ask the same L2 lines at the same time, was measured: no difference)
  int tap = 0, xc = 0;  // chunk kc = tap * cpt + xc: k = tap * C_in + 32 xc ..
#define XS_CHUNK(BUF, KC)                                                                                               \
  {                                                                                                                     \
    if ((KC) + 1 < nloc) {                                                                                              \
      _Pragma("unroll") for (int t = 0; t < NTW; ++t)                                                                   \
        _Pragma("unroll") for (int pc = 0; pc < 3; ++pc)                                                                \
          wq[(BUF) ^ 1][t][pc] = *reinterpret_cast<const uint4*>(wtile + (size_t)(t * nchunks + (KC) + 1) * 3072 + pc * 1024 + wlane); \
    }                                                                                                                   \
    __builtin_amdgcn_sched_barrier(0); /* the loads go out HERE: left alone, the scheduler sinks them to the end of the chunk's MFMAs */ \
    _Pragma("unroll") for (int mt = 0; mt < XS_MT; ++mt) {                                                              \
      uint4 xf[3];                                                                                                      \
      _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) xf[pc] = xp[(pc * G + xc * 4 + q) * XS_RA + mt * 16 + r + tap];  \
      _Pragma("unroll") for (int t = 0; t < NTW; ++t) acc[mt][t] = mfma_b3<NP>(wq[BUF][t], xf, acc[mt][t]);                \
    }                                                                                                                   \
    if (++xc == p.cpt) { xc = 0; ++tap; }                                                                               \
  }
  int kc = 0;
  for (; kc + 1 < nloc; kc += 2) {
    XS_CHUNK(0, kc)
    XS_CHUNK(1, kc + 1)
  }
  if (kc < nloc) XS_CHUNK(0, kc)
#undef XS_CHUNK
  XS_STAMP(2)

  // ---- epilogue from the accumulators: the lane holds out[row mt * 16 + r][n0 .. n0 + 4)
  if (EPI == SMOLTTS_EPI_RESID && p.ksplit <= 1 && (p.N & 15) == 0) {  // all residual values first (one round trip), then the stores
    float4 rr[XS_MT][NTW];
#pragma unroll
    for (int mt = 0; mt < XS_MT; ++mt) {
      const int rl = row0 + mt * 16 + r;
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        rr[mt][t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rl < T) rr[mt][t] = *reinterpret_cast<const float4*>(p.resid + row_off(b * T + rl, p.rows_per_batch, p.ldr, p.r_bstride) + (nt0 + t) * 16 + q * 4);
      }
    }
#pragma unroll
    for (int mt = 0; mt < XS_MT; ++mt) {
      const int rl = row0 + mt * 16 + r;
      if (rl >= T) continue;
      const long orow = row_off(b * T + rl, p.rows_per_batch, p.ldo, p.o_bstride);
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        float v[4] = {acc[mt][t][0], acc[mt][t][1], acc[mt][t][2], acc[mt][t][3]};
        rows_epilogue_resid(p, orow, (nt0 + t) * 16 + q * 4, v, rr[mt][t]);
      }
    }
  } else {
#pragma unroll
  for (int mt = 0; mt < XS_MT; ++mt) {
    const int rl = row0 + mt * 16 + r;
    if (rl >= T) continue;
    const int m = b * T + rl;
    const long orow = row_off(m, p.rows_per_batch, p.ldo, p.o_bstride);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int n0 = (nt0 + t) * 16 + q * 4;
      if (p.ksplit > 1) {  // partial sums [part][M][N]; splitk_reduce_kernel adds them in fixed order and applies the epilogue
        *reinterpret_cast<float4*>(p.splitk_ws + ((long)kz * p.M + m) * p.N + n0) = make_float4(acc[mt][t][0], acc[mt][t][1], acc[mt][t][2], acc[mt][t][3]);
        continue;
      }
      float v[4] = {acc[mt][t][0], acc[mt][t][1], acc[mt][t][2], acc[mt][t][3]};
      rows_epilogue<EPI>(p, m, orow, n0, v);
    }
  }
  }
#ifdef SMOLTTS_DBG_XS_STAMPS
  XS_STAMP(3)
  if (tid == 0) {
    const int st_row = p.ksplit > 1 ? 6 : (EPI == 0 && NTW == 1) ? 2 : EPI;  // (fc2's K parts and the N = 128 conv get rows of their own)
    for (int i = 0; i < 4; ++i) atomicAdd(&g_xs_stamps[st_row * 8 + i], (unsigned long long)st_sum[i]);
    atomicAdd(&g_xs_stamps[st_row * 8 + 6], 1ull);
  }
#endif
}

template <int NTW, int MT, int RA, int EPI, int NP>
int launch_xs_np(const GemmDev& g, int nsplit, hipStream_t stream) {
  const size_t lds = (size_t)3 * (g.cpt * 4) * RA * 16 + (g.ln_w ? (32 * 64 + 64) * sizeof(float) : 0);
  static PerDevice attr;  // value[d] = the largest size requested on device d so far
  const int dev = PerDevice::current();
  if (!attr.done(dev) || (int)lds > attr.value[dev]) {  // > 64 KB of dynamic LDS must be requested per kernel and device
    ST_CHECK_HIP(hipFuncSetAttribute((const void*)conv_xs_kernel<NTW, MT, RA, EPI, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if ((int)lds > attr.value[dev]) attr.value[dev] = (int)lds;
    attr.mark_done(dev);
  }
  const int T = g.rows_per_batch > 0 ? g.rows_per_batch : g.M;
  const dim3 grid((T + 16 * MT - 1) / (16 * MT), g.M / T, nsplit);
  hipLaunchKernelGGL((conv_xs_kernel<NTW, MT, RA, EPI, NP>), grid, dim3(512), lds, stream, g);
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

template <int NTW, int MT, int RA, int EPI>
int launch_xs(const GemmDev& g, int nsplit, hipStream_t stream) {
  return g.b3_products == 3 ? launch_xs_np<NTW, MT, RA, EPI, 3>(g, nsplit, stream) : launch_xs_np<NTW, MT, RA, EPI, 6>(g, nsplit, stream);
}

// fc2 of the decoder transformer (K = 2048, N = 512, layer scale + residual): K in four parts of 512, one workgroup each
// (32 rows x all 32 column tiles), partial sums to the split-K workspace, then the fixed-order reduce + epilogue pass.
bool linear_ksplit(const GemmDev& d, int epilogue) {
  return epilogue == SMOLTTS_EPI_SCALE_RESID && d.N == 512 && d.K == 2048 && d.splitk_ws && 4L * d.M * d.N <= d.splitk_cap;
}

// Linear shapes: (N, epilogue) -> column tiles per wave.  The transformer's Linears (one flat row range): 32-row tiles, the
// column tiles split over 4 workgroups; the k1 convs that end a resnet block (per-slot rows, + residual): 64-row tiles, all columns.
int linear_ntw(const GemmDev& d, int epilogue) {
  if (d.rows_per_batch > 0) {
    if (epilogue == SMOLTTS_EPI_RESID && d.K <= 256 && (d.N == 256 || d.N == 512)) return d.N / 128;
    return 0;
  }
  if (epilogue == SMOLTTS_EPI_QKV_ROPE && d.N == 1536) return 3;
  if (epilogue == SMOLTTS_EPI_SCALE_RESID && d.N == 512) return 1;
  if (epilogue == SMOLTTS_EPI_GELU && d.N == 2048) return 4;
  return 0;
}

}  // namespace

// Shapes the kernel takes.  Conv windows (K = taps * ldx over overlapping rows) of at most 256 channels: plain store epilogue,
// N = 128 or 640 (one or five column tiles per wave), whole slots, enough tiles for the chip.  Linears of K <= 512 over many
// rows: the decoder transformer's wqkv (N = 1536, RoPE / cache scatter: 32 -> 26 us at M = 2048), wo (N = 512, layer scale +
// residual: 20.6 -> 13.4 us) and fc1 (N = 2048, GELU: 38 -> 33 us); the k1 convs that end the resnet blocks of stages 1 and 2
// (K = 256 / 128 -> N = 512 / 256, + residual, per-slot rows).
bool conv_xs_applies(const GemmDev& d, int epilogue) {
  static const bool off = ST_KNOB_INT("SMOLTTS_CONV_XS", 1) == 0;  // experiments (knobs builds only)
  static const bool lin_off = ST_KNOB_INT("SMOLTTS_LINEAR_XS", 1) == 0;
  if (off || !d.w3) return false;
  if (d.x_bstride % 4 != 0 || d.ldo % 4 != 0 || d.o_bstride % 4 != 0 || d.ldx % 4 != 0) return false;
  if (d.ldx >= d.K && linear_ksplit(d, epilogue)) {
    if (lin_off || d.pro_elu || d.ln_w) return false;
    const int T = d.rows_per_batch > 0 ? d.rows_per_batch : d.M;
    return d.M % T == 0 && d.M / T <= 65535 && (long)((T + 31) / 32) * (d.M / T) * 4 >= 256;
  }
  if (d.ldx >= d.K) {  // Linear
    if (lin_off || d.pro_elu || d.K > XS_MAX_K_LINEAR || d.K % 32 != 0 || linear_ntw(d, epilogue) == 0) return false;
    if (d.ln_w && (d.rows_per_batch > 0 || d.K != 512 || !d.ln_b)) return false;  // the LayerNorm prologue: 32-row tiles of 512 values
    if (d.rows_per_batch > 0)
      return d.M % d.rows_per_batch == 0 && d.M / d.rows_per_batch <= 65535 && (long)((d.rows_per_batch + 63) / 64) * (d.M / d.rows_per_batch) >= 256;
    return (long)((d.M + 31) / 32) * 4 >= 256;
  }
  if (d.ln_w || epilogue != SMOLTTS_EPI_STORE || d.rows_per_batch <= 0 || d.M % d.rows_per_batch != 0) return false;
  if (d.ldx % 32 != 0 || d.ldx > XS_MAX_CIN || d.K % d.ldx != 0) return false;
  const int taps = (int)(d.K / d.ldx);
  if (taps < 2 || 64 + taps - 1 > XS_RA_CONV || (d.N != 128 && d.N != 640)) return false;
  const long tiles = (long)((d.rows_per_batch + 63) / 64) * (d.M / d.rows_per_batch);
  return tiles >= 256 && d.M / d.rows_per_batch <= 65535;
}

int launch_conv_xs(const GemmDev& d, int epilogue, hipStream_t stream) {
  GemmDev g = d;
  if (d.ldx >= d.K && linear_ksplit(d, epilogue)) {
    g.taps = 1;
    g.cpt = 16;  // chunks per K part
    g.ksplit = 4;
    ST_TRY((launch_xs<4, 2, 32, SMOLTTS_EPI_SCALE_RESID>(g, 4, stream)));
    hipLaunchKernelGGL((splitk_reduce_kernel<SMOLTTS_EPI_SCALE_RESID>), dim3((unsigned)(((long)d.M * (d.N >> 2) + 255) / 256)), dim3(256), 0, stream, g);
    ST_CHECK_HIP(hipGetLastError());
    return SMOLTTS_OK;
  }
  if (d.ldx >= d.K) {  // Linear: one "tap" of K values
    g.taps = 1;
    g.cpt = d.K >> 5;
    if (d.rows_per_batch > 0)
      return d.N == 256 ? launch_xs<2, 4, 64, SMOLTTS_EPI_RESID>(g, 1, stream) : launch_xs<4, 4, 64, SMOLTTS_EPI_RESID>(g, 1, stream);
    switch (linear_ntw(d, epilogue)) {
      case 3: return launch_xs<3, 2, 32, SMOLTTS_EPI_QKV_ROPE>(g, 4, stream);
      case 1: return launch_xs<1, 2, 32, SMOLTTS_EPI_SCALE_RESID>(g, 4, stream);
      default: return launch_xs<4, 2, 32, SMOLTTS_EPI_GELU>(g, 4, stream);
    }
  }
  g.taps = (int)(d.K / d.ldx);
  g.cpt = (int)(d.ldx >> 5);
  return d.N == 640 ? launch_xs<5, 4, XS_RA_CONV, SMOLTTS_EPI_STORE>(g, 1, stream) : launch_xs<1, 4, XS_RA_CONV, SMOLTTS_EPI_STORE>(g, 1, stream);
}

}  // namespace smoltts

#ifdef SMOLTTS_DBG_XS_STAMPS
extern "C" int smoltts_debug_xs_stamps(unsigned long long* out64, int reset) {
  if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(smoltts::g_xs_stamps), 64 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[64] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(smoltts::g_xs_stamps), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
