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
#include <stdlib.h>

#include "gemm_dev.h"

namespace smoltts {

namespace {

constexpr int XS_S = 64, XS_MT = 4, XS_RA = 80;  // output rows per tile, 16-row tiles, LDS rows per plane (>= S + taps - 1; % 16 == 0)
constexpr int XS_MAX_CIN = 256;

__device__ __forceinline__ float elu_hw_xs(float x) { return x > 0.f ? x : __expf(x) - 1.0f; }

template <int NTW>
__global__ __launch_bounds__(512) void conv_xs_kernel(GemmDev p) {
  extern __shared__ __attribute__((aligned(16))) uint4 xp[];  // [piece][C_in / 8 groups = (chunk, q)][XS_RA rows]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int b = blockIdx.y, row0 = blockIdx.x * XS_S, T = p.rows_per_batch;
  const int G = (int)(p.ldx >> 3), nrows = XS_S + p.taps - 1, nchunks = p.K >> 5;
  const float* xb = p.x + (long)b * p.x_bstride + (long)row0 * p.ldx;

  // ---- the tile's window rows -> pieces; 8 consecutive lanes take 8 consecutive rows of one 8-channel group (128 contiguous
  //      LDS bytes per ds_write_b128 lane group)
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

  f32x4 acc[XS_MT][NTW];
#pragma unroll
  for (int mt = 0; mt < XS_MT; ++mt)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // W3 tile (column tile nt, chunk kc) at (nt * nchunks + kc) * 3072: the wave's tiles are NTW * nchunks consecutive tiles
  const char* wtile = p.w3 + (size_t)(wave * NTW) * nchunks * 3072;
  const int wlane = lane * 16;
  uint4 wq[2][NTW][3];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) wq[0][t][pc] = *reinterpret_cast<const uint4*>(wtile + (size_t)(t * nchunks) * 3072 + pc * 1024 + wlane);
  lds_barrier();  // (the W3 loads above stay in flight)

  // (every workgroup streams the same W3 tiles in the same order; starting each tile's K loop at a different chunk, so that
  // they do not ask the same L2 lines at the same time, was measured: no difference)
  int tap = 0, xc = 0;  // chunk kc = tap * cpt + xc: k = tap * C_in + 32 xc ..
#define XS_CHUNK(BUF, KC)                                                                                               \
  {                                                                                                                     \
    if ((KC) + 1 < nchunks) {                                                                                           \
      _Pragma("unroll") for (int t = 0; t < NTW; ++t)                                                                   \
        _Pragma("unroll") for (int pc = 0; pc < 3; ++pc)                                                                \
          wq[(BUF) ^ 1][t][pc] = *reinterpret_cast<const uint4*>(wtile + (size_t)(t * nchunks + (KC) + 1) * 3072 + pc * 1024 + wlane); \
    }                                                                                                                   \
    _Pragma("unroll") for (int mt = 0; mt < XS_MT; ++mt) {                                                              \
      uint4 xf[3];                                                                                                      \
      _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) xf[pc] = xp[(pc * G + xc * 4 + q) * XS_RA + mt * 16 + r + tap];  \
      _Pragma("unroll") for (int t = 0; t < NTW; ++t) acc[mt][t] = mfma_b3(wq[BUF][t], xf, acc[mt][t]);                \
    }                                                                                                                   \
    if (++xc == p.cpt) { xc = 0; ++tap; }                                                                               \
  }
  int kc = 0;
  for (; kc + 1 < nchunks; kc += 2) {
    XS_CHUNK(0, kc)
    XS_CHUNK(1, kc + 1)
  }
  if (kc < nchunks) XS_CHUNK(0, kc)
#undef XS_CHUNK

  // ---- epilogue from the accumulators: the lane holds out[row mt * 16 + r][n0 .. n0 + 4)
#pragma unroll
  for (int mt = 0; mt < XS_MT; ++mt) {
    const int rl = row0 + mt * 16 + r;
    if (rl >= T) continue;
    const int m = b * T + rl;
    const long orow = row_off(m, T, p.ldo, p.o_bstride);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      float v[4] = {acc[mt][t][0], acc[mt][t][1], acc[mt][t][2], acc[mt][t][3]};
      rows_epilogue<SMOLTTS_EPI_STORE>(p, m, orow, (wave * NTW + t) * 16 + q * 4, v);
    }
  }
}

template <int NTW>
int launch_xs(const GemmDev& g, hipStream_t stream) {
  const size_t lds = (size_t)3 * (g.ldx >> 3) * XS_RA * 16;
  static size_t attr = 0;
  if (lds > attr) {  // > 64 KB of dynamic LDS must be requested per kernel
    ST_CHECK_HIP(hipFuncSetAttribute((const void*)conv_xs_kernel<NTW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = lds;
  }
  const dim3 grid((g.rows_per_batch + XS_S - 1) / XS_S, g.M / g.rows_per_batch);
  hipLaunchKernelGGL((conv_xs_kernel<NTW>), grid, dim3(512), lds, stream, g);
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

}  // namespace

// Shapes the kernel takes: plain store epilogue, a conv window (K = taps * ldx over overlapping rows) of at most 256 channels,
// N = 128 or 640 (one or five column tiles per wave), whole slots, and enough tiles for the chip.
bool conv_xs_applies(const GemmDev& d, int epilogue) {
  static const bool off = [] { const char* e = getenv("SMOLTTS_CONV_XS"); return e && atoi(e) == 0; }();  // experiments
  if (off || epilogue != SMOLTTS_EPI_STORE || !d.w3 || d.rows_per_batch <= 0 || d.M % d.rows_per_batch != 0) return false;
  if (d.ldx >= d.K || d.ldx % 32 != 0 || d.ldx > XS_MAX_CIN || d.K % d.ldx != 0) return false;
  const int taps = (int)(d.K / d.ldx);
  if (taps < 2 || XS_S + taps - 1 > XS_RA || (d.N != 128 && d.N != 640)) return false;
  if (d.x_bstride % 4 != 0 || d.ldo % 4 != 0 || d.o_bstride % 4 != 0) return false;
  const long tiles = (long)((d.rows_per_batch + XS_S - 1) / XS_S) * (d.M / d.rows_per_batch);
  return tiles >= 256 && d.M / d.rows_per_batch <= 65535;
}

int launch_conv_xs(const GemmDev& d, hipStream_t stream) {
  GemmDev g = d;
  g.taps = (int)(d.K / d.ldx);
  g.cpt = (int)(d.ldx >> 5);
  return d.N == 640 ? launch_xs<5>(g, stream) : launch_xs<1>(g, stream);
}

}  // namespace smoltts
