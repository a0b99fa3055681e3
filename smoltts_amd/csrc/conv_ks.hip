// Causal conv / ConvTranspose GEMM over MORE than 256 input channels (the first SEANet layers: conv0 512 -> 1024 k7, the
// ConvTranspose 1024 -> 512 s8, the k3 conv of the first resnet block at 512 channels, the ConvTranspose 512 -> 256 s6):
//   out[m][n] = bias[n] + sum_k X[m][k] W[n][k],  row m's K = taps * C_in values are the `taps` consecutive channel-last rows
//   starting at row m of its slot's halo-prefixed buffer.
//
// Reference ops: Conv1d / ConvTranspose1d of mlx_inference/src/smoltts_mlx/codec/conv.py:68-220 as used by the SEANet decoder
// (codec/seanet.py:99-161), fp32.
//
// conv_xs.hip keeps a tile's whole window (all channels) as bf16x3 pieces in LDS, which stops at 256 channels; gemm_b3.hip
// (what these four layers ran on until round 3) re-splits its X chunk in every column block and pays two barriers per 32-k
// chunk: 26-40 % of the bf16x3 matrix-core rate.  Here the channels go through LDS in SLICES of 128: the 64 + taps - 1 window
// rows of a slice are split into pieces once per workgroup (a tap is a row shift in LDS, so all `taps` x 4 chunks of the
// slice run off it), two slice buffers (one barrier per slice, the next slice's global loads fly under this slice's MFMAs),
// every wave owns NTW column tiles for all four 16-row tiles and streams its W3 fragments from L2 straight into registers
// one chunk ahead, as conv_xs does.  The K order is therefore (slice, tap, chunk) -- fixed, so results are deterministic.
//
// Workgroup -> (row tile, column part): with 8 column parts the part is id % 8 = the XCD the workgroup lands on, so each
// XCD's L2 holds ONE part of W3 (conv0: 2.75 of 22 MB) and X (4-8 MB) is what every XCD reads; with fewer parts the parts
// run one after the other over the whole chip (the ConvTranspose 512 -> 256: 3 parts of 3.1 MB).
#include <stdlib.h>

#include "gemm_dev.h"

namespace smoltts {

namespace {

constexpr int KS_CS = 128, KS_GS = KS_CS / 8, KS_CPS = KS_CS / 32;  // channels, 8-channel groups, 32-k chunks per slice (and tap)
constexpr int KS_RA = 80;                                            // LDS rows per plane >= 64 + taps - 1, % 16 == 0
constexpr int KS_BUF_U4 = 3 * KS_GS * KS_RA;                         // one slice buffer: [piece][group][row] 16-byte slots
constexpr int KS_IT = (KS_RA * KS_GS + 511) / 512;                   // (row, group) items of a slice per thread
constexpr size_t KS_LDS_BYTES = (size_t)2 * KS_BUF_U4 * 16;

__device__ __forceinline__ float elu_hw_ks(float x) { return x > 0.f ? x : __expf(x) - 1.0f; }

#ifdef SMOLTTS_DBG_XS_STAMPS  // (diagnostic variant, tools/stamps_xs.py: cycles per phase as seen by thread 0 of every workgroup)
__device__ unsigned long long g_ks_stamps[8 * 8];  // [NTW, 3 = NTW 4 in 3 column parts][phase 0..4, 6 = workgroups]
#define KS_STAMP(I) { const long long now_ = clock64(); st_sum[I] += now_ - st_last; st_last = now_; }
#else
#define KS_STAMP(I)
#endif

template <int NTW, int EPI, int NP>
__global__ __launch_bounds__(512) void conv_ks_kernel(GemmDev p, int tiles_per_slot, int nparts) {
  constexpr int MT = 4;
  extern __shared__ __attribute__((aligned(16))) uint4 xp[];  // two slice buffers
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int T = p.rows_per_batch, ntiles = tiles_per_slot * (p.M / T);
  const int id = blockIdx.x;
  int part, tile;
  if ((nparts & 7) == 0) { part = id % nparts; tile = id / nparts; } else { part = id / ntiles; tile = id - part * ntiles; }
  const int b = tile / tiles_per_slot, row0 = (tile - b * tiles_per_slot) * 64;
  const int C = (int)p.ldx, taps = p.taps, cpt = C >> 5, nchunks = p.K >> 5, nslices = C / KS_CS;
  const int nrows = 64 + taps - 1;
  const int nt0 = (part * 8 + wave) * NTW;  // the wave's first column tile
  const float* xb = p.x + (long)b * p.x_bstride + (long)row0 * p.ldx;
#ifdef SMOLTTS_DBG_XS_STAMPS
  long long st_sum[6] = {0, 0, 0, 0, 0, 0}, st_last = clock64();
#endif

  // the thread's (row, group) items of a slice: 8 consecutive lanes take 8 consecutive rows of one 8-channel group
  // (128 contiguous LDS bytes per ds_write_b128 lane group)
  int item_off[KS_IT];   // slot within a buffer (piece 0), -1: none
  long item_src[KS_IT];  // float offset from xb (slice 0), -1: zeros (row past the slot's buffer)
#pragma unroll
  for (int it = 0; it < KS_IT; ++it) {
    const int idx = tid + 512 * it, i = idx / (8 * KS_GS) * 8 + (idx & 7), g8 = (idx >> 3) % KS_GS;
    const bool have = idx < KS_RA * KS_GS && i < nrows;
    item_off[it] = have ? g8 * KS_RA + i : -1;
    item_src[it] = have && row0 + i < T + taps - 1 ? (long)i * p.ldx + g8 * 8 : -1;
  }
  float4 xr[KS_IT][2];
#define KS_FETCH(S)                                                                        \
  _Pragma("unroll") for (int it = 0; it < KS_IT; ++it) {                                   \
    xr[it][0] = xr[it][1] = make_float4(0.f, 0.f, 0.f, 0.f);                               \
    if (item_src[it] >= 0) {                                                               \
      const float* src = xb + item_src[it] + (S) * KS_CS;                                  \
      xr[it][0] = *reinterpret_cast<const float4*>(src);                                   \
      xr[it][1] = *reinterpret_cast<const float4*>(src + 4);                               \
    }                                                                                      \
  }
#define KS_STAGE(BUF)                                                                      \
  _Pragma("unroll") for (int it = 0; it < KS_IT; ++it) {                                   \
    if (item_off[it] >= 0) {                                                               \
      float4 a = xr[it][0], c = xr[it][1];                                                 \
      if (p.pro_elu) {                                                                     \
        a = make_float4(elu_hw_ks(a.x), elu_hw_ks(a.y), elu_hw_ks(a.z), elu_hw_ks(a.w));   \
        c = make_float4(elu_hw_ks(c.x), elu_hw_ks(c.y), elu_hw_ks(c.z), elu_hw_ks(c.w));   \
      }                                                                                    \
      uint4 h, m, l;                                                                       \
      split3x8(a, c, h, m, l);                                                             \
      uint4* dst = xp + (BUF) * KS_BUF_U4 + item_off[it];                                  \
      dst[0] = h;                                                                          \
      dst[KS_GS * KS_RA] = m;                                                              \
      dst[2 * KS_GS * KS_RA] = l;                                                          \
    }                                                                                      \
  }

  f32x4 acc[MT][NTW];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // W3 tile (column tile nt, chunk kc) at (nt * nchunks + kc) * 3072
  const char* wtile = p.w3 + (size_t)nt0 * nchunks * 3072 + lane * 16;
  uint4 wq[2][NTW][3];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) wq[0][t][pc] = *reinterpret_cast<const uint4*>(wtile + (size_t)(t * nchunks) * 3072 + pc * 1024);
  KS_FETCH(0)

  // chunk (slice s, tap, XC): k = tap * C_in + 128 s + 32 XC ..; the next chunk's W3 fragments go out before this chunk's MFMAs
#define KS_CHUNK(BUF, XC)                                                                                               \
  {                                                                                                                     \
    int nkc = tap * cpt + s * KS_CPS + (XC) + 1;                                                                        \
    if ((XC) == KS_CPS - 1) nkc = tap + 1 < taps ? (tap + 1) * cpt + s * KS_CPS : (s + 1 < nslices ? (s + 1) * KS_CPS : -1); \
    if (nkc >= 0) {                                                                                                     \
      _Pragma("unroll") for (int t = 0; t < NTW; ++t)                                                                   \
        _Pragma("unroll") for (int pc = 0; pc < 3; ++pc)                                                                \
          wq[(BUF) ^ 1][t][pc] = *reinterpret_cast<const uint4*>(wtile + (size_t)(t * nchunks + nkc) * 3072 + pc * 1024); \
    }                                                                                                                   \
    __builtin_amdgcn_sched_barrier(0); /* the loads go out HERE, not after the chunk's MFMAs */                         \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                                 \
      uint4 xf[3];                                                                                                      \
      _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) xf[pc] = xs[(pc * KS_GS + (XC) * 4 + q) * KS_RA + mt * 16 + r + tap]; \
      _Pragma("unroll") for (int t = 0; t < NTW; ++t) acc[mt][t] = mfma_b3<NP>(wq[BUF][t], xf, acc[mt][t]);                \
    }                                                                                                                   \
  }
  for (int s = 0; s < nslices; ++s) {
    // slice s -> buffer s & 1 (last read in slice s - 2: every wave is past that since the barrier of slice s - 1)
    if (s & 1) { KS_STAGE(1) } else { KS_STAGE(0) }
    KS_STAMP(0)
    lds_barrier();
    KS_STAMP(1)
    if (s + 1 < nslices) { KS_FETCH(s + 1) }
    const uint4* xs = xp + (s & 1) * KS_BUF_U4;
    for (int tap = 0; tap < taps; ++tap) {
      KS_CHUNK(0, 0)
      KS_CHUNK(1, 1)
      KS_CHUNK(0, 2)
      KS_CHUNK(1, 3)
    }
    KS_STAMP(2)
  }
#undef KS_CHUNK
#undef KS_STAGE
#undef KS_FETCH

  // ---- epilogue from the accumulators: the lane holds out[row mt * 16 + r][n0 .. n0 + 4)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int rl = row0 + mt * 16 + r;
    if (rl >= T) continue;
    const int m = b * T + rl;
    const long orow = row_off(m, p.rows_per_batch, p.ldo, p.o_bstride);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      float v[4] = {acc[mt][t][0], acc[mt][t][1], acc[mt][t][2], acc[mt][t][3]};
      rows_epilogue<EPI>(p, m, orow, (nt0 + t) * 16 + q * 4, v);
    }
  }
#ifdef SMOLTTS_DBG_XS_STAMPS
  KS_STAMP(3)
  if (tid == 0) {
    const int st_row = NTW == 4 && nparts == 3 ? 3 : NTW;
    for (int i = 0; i < 4; ++i) atomicAdd(&g_ks_stamps[st_row * 8 + i], (unsigned long long)st_sum[i]);
    atomicAdd(&g_ks_stamps[st_row * 8 + 6], 1ull);
  }
#endif
}

template <int NTW, int NP>
int launch_ks_np(const GemmDev& g, int tiles_per_slot, int nparts, hipStream_t stream) {
  static PerDevice attr;
  const int dev = PerDevice::current();
  if (!attr.done(dev)) {  // > 64 KB of dynamic LDS must be requested per kernel and device
    ST_CHECK_HIP(hipFuncSetAttribute((const void*)conv_ks_kernel<NTW, SMOLTTS_EPI_STORE, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KS_LDS_BYTES));
    attr.mark_done(dev);
  }
  const long blocks = (long)tiles_per_slot * (g.M / g.rows_per_batch) * nparts;
  hipLaunchKernelGGL((conv_ks_kernel<NTW, SMOLTTS_EPI_STORE, NP>), dim3((unsigned)blocks), dim3(512), KS_LDS_BYTES, stream, g, tiles_per_slot, nparts);
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

template <int NTW>
int launch_ks(const GemmDev& g, int tiles_per_slot, int nparts, hipStream_t stream) {
  return g.b3_products == 3 ? launch_ks_np<NTW, 3>(g, tiles_per_slot, nparts, stream) : launch_ks_np<NTW, 6>(g, tiles_per_slot, nparts, stream);
}

// column tiles per wave: the largest of 4 / 2 / 1 that still gives the chip a workgroup per CU
int ks_ntw(const GemmDev& d) {
  const int ct = d.N / 16;
  const long tiles = (long)((d.rows_per_batch + 63) / 64) * (d.M / d.rows_per_batch);
  for (int ntw = 4; ntw >= 1; ntw >>= 1)
    if (ct % (8 * ntw) == 0 && tiles * (ct / (8 * ntw)) >= 256) return ntw;
  return 0;
}

}  // namespace

bool conv_ks_applies(const GemmDev& d, int epilogue) {
  static const bool off = ST_KNOB_INT("SMOLTTS_CONV_KS", 1) == 0;  // experiments (knobs builds only)
  if (off || !d.w3 || epilogue != SMOLTTS_EPI_STORE || d.ln_w) return false;
  if (d.rows_per_batch <= 0 || d.M % d.rows_per_batch != 0 || d.M / d.rows_per_batch > (1 << 20)) return false;
  if (d.ldx >= d.K || d.ldx <= 256 || d.ldx % KS_CS != 0 || d.K % d.ldx != 0) return false;  // conv windows over > 256 channels
  if (d.x_bstride % 4 != 0 || d.ldo % 4 != 0 || d.o_bstride % 4 != 0 || (d.raw_out && d.raw_bstride % 4 != 0)) return false;
  const int taps = (int)(d.K / d.ldx);
  if (64 + taps - 1 > KS_RA || d.N % 128 != 0) return false;
  return ks_ntw(d) != 0;
}

int launch_conv_ks(const GemmDev& d, int epilogue, hipStream_t stream) {
  ST_REQUIRE(conv_ks_applies(d, epilogue), SMOLTTS_E_INVALID, "conv_ks: shape or epilogue not supported");
  GemmDev g = d;
  g.taps = (int)(d.K / d.ldx);
  g.cpt = (int)(d.ldx >> 5);
  g.ksplit = 1;
  const int ntw = ks_ntw(d), tiles_per_slot = (d.rows_per_batch + 63) / 64, nparts = d.N / 16 / (8 * ntw);
  switch (ntw) {
    case 4: return launch_ks<4>(g, tiles_per_slot, nparts, stream);
    case 2: return launch_ks<2>(g, tiles_per_slot, nparts, stream);
    default: return launch_ks<1>(g, tiles_per_slot, nparts, stream);
  }
}

}  // namespace smoltts

#ifdef SMOLTTS_DBG_XS_STAMPS
extern "C" int smoltts_debug_ks_stamps(unsigned long long* out64, int reset) {
  if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(smoltts::g_ks_stamps), 64 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[64] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(smoltts::g_ks_stamps), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
