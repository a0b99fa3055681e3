// Fused SEANet resnet block of the Mimi decoder (stage 3; the last stage is seanet_last.hip).
//
// Reference: MimiResnetBlock (mlx_inference/src/smoltts_mlx/codec/seanet.py:8-49, dilation 1):
//   y = x + Conv1d_k1( ELU( Conv1d_k3( ELU(x) ) ) ),   C -> C/2 -> C channels, causal (left zero padding of ELU(x)),
// followed in MimiDecoder (seanet.py:99-139) by ELU and the next ConvTranspose1d.
//
// Unfused, the block's tensors move through HBM five times per 1024 frames (raw + ELU copy of x, the hidden, y, the
// residual re-read: ~1.8 GB at 128 channels x 0.49 M rows) for 10 % of the FLOPs.
// Here one workgroup takes RC consecutive rows of one slot: x (raw ConvTranspose output) is read once, ELU(x) is split
// into bf16x3 pieces in LDS (gemm_dev.h: fp32-grade products on the bf16 matrix cores), both convolutions run as
// 16x16x32 MFMAs over LDS-resident operands with the weights staged through LDS in groups of 32-k chunks, the hidden
// never leaves LDS, and only ELU(y) is written.  The causal k3 windows of a tile need 2 rows of the previous tile: read
// again from x, whose halo rows in front of the buffer carry the previous call's last rows (streaming), so chunked
// decoding equals decoding in one call.
#include "gemm_dev.h"
#include "mimi_common.h"

namespace smoltts {

namespace {

struct ResDev {
  const float* x;   // raw block input, row 0 of slot 0; 2 halo rows sit in front of every slot's rows
  long x_bstride;   // floats per slot
  int T;            // rows per slot in this call
  const char* w2;   // conv k3, C -> C/2, as W3 tiles of the GEMM matrix [C/2][3C] (k = tap * C + channel)
  const float* b2;
  const char* w3;   // conv k1, C/2 -> C, as W3 tiles of [C][C/2]
  const float* b3;
  float* out;       // ELU(y), row 0 of slot 0 of the next stage's input buffer
  long o_bstride;
};

// Tiles are kept small on purpose: a tile is a chain of dependent memory round trips (x, weight groups, residual) with
// barriers in between, so what hides the latency is several workgroups per CU (LDS is the limit: 37-74 KB per workgroup),
// at the price of re-reading the (L2-resident) weights once per tile.
// ELU with the hardware exponential: exp(x) - 1 carries an absolute error of ~1 ulp of 1 (6e-8) where expm1f is relatively
// exact -- the size of the fp32 rounding of the O(1) activations around it, at a fifth of the instructions (the block applies
// ELU to every element of x, of the hidden and of y: with expm1f the vector units, not the matrix cores, bound the kernel).
__device__ __forceinline__ float elu_fast(float x) { return x > 0.f ? x : __expf(x) - 1.0f; }

template <int C, int RC, int NW>
struct ResCfg {
  static constexpr int H = C / 2, RT = RC, RL = RC + 2, RLP = (RL + 15) / 16 * 16;
  static constexpr int CC = C / 32, HC = H / 32, K2C = 3 * CC;
  static constexpr int NT2 = H / 16, NT3 = C / 16, MT = RC / 16;
  static constexpr int WR = MT < NW ? MT : NW, WC = NW / WR;  // wave arrangement: row tiles x column groups
  static constexpr int HP_U4 = 3 * CC * 4 * RLP, VP_U4 = 3 * HC * 4 * RC;
  static constexpr int GK3 = 1;                                   // conv k1: one 32-k chunk of all its column tiles at a time
  static constexpr int WB_BYTES = NT3 * 3072 > 12288 ? NT3 * 3072 : 12288;
  static constexpr int GK2 = (WB_BYTES / (NT2 * 3072)) >= K2C ? K2C : ((WB_BYTES / (NT2 * 3072)) >= 2 ? 2 : 1);
  static constexpr size_t LDS = (size_t)(HP_U4 + VP_U4) * 16 + WB_BYTES;
  static_assert(MT % WR == 0 && NW % WR == 0 && NT2 % WC == 0 && NT3 % WC == 0, "wave arrangement");
  static_assert(K2C % GK2 == 0 && NT2 * GK2 * 3072 <= WB_BYTES && NT3 * GK3 * 3072 <= WB_BYTES, "weight groups");
};

template <int C, int RC, int NW, int NP>
__global__ __launch_bounds__(NW * 64) void resblock_kernel(ResDev p) {
  using K = ResCfg<C, RC, NW>;
  constexpr int NTHR = NW * 64;
  constexpr int H = K::H, RT = K::RT, RL = K::RL, RLP = K::RLP, CC = K::CC, HC = K::HC, K2C = K::K2C;
  constexpr int NT2 = K::NT2, NT3 = K::NT3, MT = K::MT, WR = K::WR, WC = K::WC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* hp = reinterpret_cast<uint4*>(smem);                  // ELU(x) pieces [piece][x chunk][q][row]
  uint4* vp = hp + K::HP_U4;                                   // ELU(hidden) pieces [piece][h chunk][q][row]
  uint4* wb = vp + K::VP_U4;                                   // weight group [col tile][chunk in group][piece][lane]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const int b = blockIdx.y;
  const int u0 = blockIdx.x * RT;   // first output row of the tile
  const int cu0 = u0;               // first computed row (y); x rows cu0 - 2 .. cu0 + RC - 1 are loaded
  const float* xb = p.x + (long)b * p.x_bstride;
  const int wr = wave % WR, wc = wave / WR;

  // Weight groups go global -> registers -> LDS so that the L2 round trip of the next group hides behind the MFMAs of the
  // current one: GK chunks of every column tile = NT * GK * 192 16-byte pieces, WPT per thread (compile-time shapes: the
  // register array must never become addressable).
  constexpr int WPT = (K::WB_BYTES / 16 + NTHR - 1) / NTHR;
  uint4 wreg[WPT];
#define ST_FETCH_W(WSRC, NT, KC, K0, GK)                                                                          \
  _Pragma("unroll") for (int t_ = 0; t_ < WPT; ++t_) {                                                            \
    const int j_ = tid + t_ * NTHR;                                                                               \
    const int blk_ = j_ / 192, rem_ = j_ - blk_ * 192, nt_ = blk_ / (GK), kcl_ = blk_ - nt_ * (GK);              \
    wreg[t_] = j_ < (NT) * (GK) * 192                                                                             \
                   ? *reinterpret_cast<const uint4*>((WSRC) + (size_t)(nt_ * (KC) + (K0) + kcl_) * 3072 + rem_ * 16) \
                   : make_uint4(0, 0, 0, 0);                                                                      \
  }
#define ST_STORE_W(NT, GK)                                                                                        \
  _Pragma("unroll") for (int t_ = 0; t_ < WPT; ++t_) {                                                            \
    const int j_ = tid + t_ * NTHR;                                                                               \
    if (j_ < (NT) * (GK) * 192) wb[j_] = wreg[t_];                                                                \
  }

  // ---- phase 0: x -> ELU -> bf16x3 pieces in LDS (the first weight group rides along)
  ST_FETCH_W(p.w2, NT2, K2C, 0, K::GK2)
  // 8 consecutive lanes take 8 consecutive rows of one 8-channel group: 128 contiguous LDS bytes per ds_write_b128 lane group
  // (lanes along the channel groups would put all 8 on one bank: the planes are a multiple of 256 B apart)
  for (int idx = tid; idx < (RL + 7) / 8 * 8 * (C / 8); idx += NTHR) {
    const int lr = idx / (C / 8 * 8) * 8 + (idx & 7), g8 = (idx >> 3) % (C / 8);
    if (lr >= RL) continue;
    const int row = cu0 - 2 + lr;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
    if (row < p.T) {
      const float* src = xb + (long)row * C + g8 * 8;
      a = *reinterpret_cast<const float4*>(src);
      c = *reinterpret_cast<const float4*>(src + 4);
    }
    a = make_float4(elu_fast(a.x), elu_fast(a.y), elu_fast(a.z), elu_fast(a.w));
    c = make_float4(elu_fast(c.x), elu_fast(c.y), elu_fast(c.z), elu_fast(c.w));
    uint4 h, m, l;
    split3x8(a, c, h, m, l);
    const int slot = ((g8 >> 2) * 4 + (g8 & 3)) * RLP + lr;  // [x chunk][q][row] inside a piece plane
    hp[slot] = h;
    hp[slot + CC * 4 * RLP] = m;
    hp[slot + 2 * CC * 4 * RLP] = l;
  }
  ST_STORE_W(NT2, K::GK2)
  __syncthreads();

  // ---- phase B: hidden = conv k3 over ELU(x): out row i reads ELU(x) local rows i, i+1, i+2 (taps 0..2)
  {
    constexpr int RW = MT / WR, CW = NT2 / WC, G2 = K2C / K::GK2;
    f32x4 acc[RW][CW];
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
      for (int j = 0; j < CW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < G2; ++g) {
      if (g + 1 < G2) {  // next group (or conv k1's first chunk) in flight
        ST_FETCH_W(p.w2, NT2, K2C, (g + 1) * K::GK2, K::GK2)
      } else {
        ST_FETCH_W(p.w3, NT3, HC, 0, K::GK3)
      }
#pragma unroll
      for (int kcl = 0; kcl < K::GK2; ++kcl) {
        const int kc = g * K::GK2 + kcl, tap = kc / CC, xc = kc - tap * CC;
#pragma unroll
        for (int i = 0; i < RW; ++i) {
          const int mt = wr + i * WR;
          uint4 xf[3];
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) xf[pc] = hp[((pc * CC + xc) * 4 + q) * RLP + mt * 16 + r + tap];
#pragma unroll
          for (int j = 0; j < CW; ++j) {
            uint4 wf[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) wf[pc] = wb[(((wc * CW + j) * K::GK2 + kcl) * 3 + pc) * 64 + lane];
            acc[i][j] = mfma_b3<NP>(wf, xf, acc[i][j]);
          }
        }
      }
      __syncthreads();  // everyone is done with this group
      if (g + 1 < G2) {
        ST_STORE_W(NT2, K::GK2)
        __syncthreads();
      } else {
        ST_STORE_W(NT3, K::GK3)
      }
    }
    // hidden + bias -> ELU -> pieces: the lane holds hidden[row mt*16 + r][n = ct*16 + 4q .. +4)
    uint2* vp2 = reinterpret_cast<uint2*>(vp);
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
      for (int j = 0; j < CW; ++j) {
        const int row = (wr + i * WR) * 16 + r, ct = wc * CW + j, n = ct * 16 + 4 * q;
        const float4 bb = *reinterpret_cast<const float4*>(p.b2 + n);
        const float v0 = elu_fast(acc[i][j][0] + bb.x), v1 = elu_fast(acc[i][j][1] + bb.y), v2 = elu_fast(acc[i][j][2] + bb.z),
                    v3 = elu_fast(acc[i][j][3] + bb.w);
        uint32_t h0, m0, l0, h1, m1, l1;
        split3_pair(v0, v1, h0, m0, l0);
        split3_pair(v2, v3, h1, m1, l1);
        const int hc = ct >> 1, qv = (ct & 1) * 2 + (q >> 1);
        const int s2 = ((hc * 4 + qv) * RC + row) * 2 + (q & 1);  // 8-byte half of the row's 16-byte slot
        vp2[s2] = make_uint2(h0, h1);
        vp2[s2 + HC * 4 * RC * 2] = make_uint2(m0, m1);
        vp2[s2 + 2 * HC * 4 * RC * 2] = make_uint2(l0, l1);
      }
    __syncthreads();  // hidden pieces and conv k1's first chunk are in LDS
  }

  // ---- phase C: y = conv k1 over ELU(hidden) + x
  {
    constexpr int RW = MT / WR, CW = NT3 / WC, G3 = HC / K::GK3;
    f32x4 acc[RW][CW];
    float4 xres[RW][CW];  // the residual: x again (L2), requested before the MFMAs
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int grow = cu0 + (wr + i * WR) * 16 + r;
#pragma unroll
      for (int j = 0; j < CW; ++j) {
        acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        xres[i][j] = grow < p.T ? *reinterpret_cast<const float4*>(xb + (long)grow * C + (wc * CW + j) * 16 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    for (int g = 0; g < G3; ++g) {
      if (g + 1 < G3) {
        ST_FETCH_W(p.w3, NT3, HC, (g + 1) * K::GK3, K::GK3)
      }
#pragma unroll
      for (int kcl = 0; kcl < K::GK3; ++kcl) {
        const int kc = g * K::GK3 + kcl;
#pragma unroll
        for (int i = 0; i < RW; ++i) {
          const int mt = wr + i * WR;
          uint4 xf[3];
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) xf[pc] = vp[((pc * HC + kc) * 4 + q) * RC + mt * 16 + r];
#pragma unroll
          for (int j = 0; j < CW; ++j) {
            uint4 wf[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) wf[pc] = wb[(((wc * CW + j) * K::GK3 + kcl) * 3 + pc) * 64 + lane];
            acc[i][j] = mfma_b3<NP>(wf, xf, acc[i][j]);
          }
        }
      }
      if (g + 1 < G3) {
        __syncthreads();
        ST_STORE_W(NT3, K::GK3)
        __syncthreads();
      }
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int row = (wr + i * WR) * 16 + r, grow = cu0 + row;
#pragma unroll
      for (int j = 0; j < CW; ++j) {
        const int n = (wc * CW + j) * 16 + 4 * q;
        float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
        if (grow < p.T) {
          const float4 bb = *reinterpret_cast<const float4*>(p.b3 + n), xr = xres[i][j];
          e = make_float4(elu_fast(acc[i][j][0] + bb.x + xr.x), elu_fast(acc[i][j][1] + bb.y + xr.y), elu_fast(acc[i][j][2] + bb.z + xr.z),
                          elu_fast(acc[i][j][3] + bb.w + xr.w));
        }
        if (grow < p.T) *reinterpret_cast<float4*>(p.out + (long)b * p.o_bstride + (long)grow * C + n) = e;
      }
    }
  }

#undef ST_FETCH_W
#undef ST_STORE_W

}

template <int C, int RC, int NW, int NP>
int launch_res(const ResDev& d, int batch, hipStream_t st) {
  using K = ResCfg<C, RC, NW>;
  static PerDevice attr;
  const int dev = PerDevice::current();
  if (!attr.done(dev)) {  // > 64 KB of dynamic LDS must be requested once per kernel and device
    ST_CHECK_HIP(hipFuncSetAttribute((const void*)resblock_kernel<C, RC, NW, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS));
    attr.mark_done(dev);
  }
  const dim3 grid((d.T + K::RT - 1) / K::RT, batch);
  ST_REQUIRE(grid.y <= 65535, SMOLTTS_E_INVALID, "resblock: batch too large");
  hipLaunchKernelGGL((resblock_kernel<C, RC, NW, NP>), grid, dim3(NW * 64), K::LDS, st, d);
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

}  // namespace

int launch_seanet_resblock(const MimiResblockArgs& a, hipStream_t st) {
  ST_REQUIRE(a.x && a.w2 && a.b2 && a.w3 && a.b3 && a.out && a.T > 0 && a.batch > 0, SMOLTTS_E_INVALID, "resblock: null or empty argument");
  ResDev d{a.x, (long)a.x_bstride, a.T, (const char*)a.w2, a.b2, (const char*)a.w3, a.b3, a.out, (long)a.o_bstride};
  // (48- and 64-row tiles were measured: 460 / 367 us against 341 us -- one workgroup per CU hides less than two or three)
  if (a.channels == 128) return a.b3_products == 3 ? launch_res<128, 32, 8, 3>(d, a.batch, st) : launch_res<128, 32, 8, 6>(d, a.batch, st);
  set_error("resblock: no instance for %d channels", a.channels);
  return SMOLTTS_E_INVALID;
}

}  // namespace smoltts
