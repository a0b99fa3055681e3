// Last SEANet decoder stage in one kernel: ConvTranspose1d (128 -> 64 channels, stride 4, kernel 8) -> resnet block
// (64 -> 32 -> 64) -> ELU -> output Conv1d k3 (64 -> 1) -> PCM.
//
// Reference: MimiDecoder (mlx_inference/src/smoltts_mlx/codec/seanet.py:99-139, last ratio) = ConvTranspose1d
// (codec/conv.py:143-220, causal: the stride-4 output rows 4s + j, j < 4, read input rows s-1 and s), MimiResnetBlock
// (seanet.py:8-49: y = x + Conv1d_k1(ELU(Conv1d_k3(ELU(x)))), causal), ELU, Conv1d k3 to one channel.
//
// Why one kernel: the stage's tensors are the largest of the decoder -- per 1024 frames the ConvTranspose output is 503 MB
// (1.97 M rows x 64 channels, fp32).  As two kernels it is written once and read once (rocprofv3: 0.78 + 0.58 GB of the
// 3.7 GB the whole SEANet decoder moves); here it only ever exists as accumulators and LDS pieces: the stage reads its 252 MB
// input and writes 7.9 MB of samples.
//
// One workgroup (8 waves) takes S = 64 consecutive input rows of one slot = 256 rows of the 64-channel tensor x:
//   A. x = ConvTranspose as a [64 x 256] . [256 x 256] GEMM on the bf16 matrix cores with both operands split in three bf16
//      pieces (gemm_dev.h): the input rows are split once into LDS, every wave owns two of the 16 column tiles (= one output
//      phase j and 32 channels) for all four row tiles and streams its W3 fragments straight from L2 into registers, one chunk
//      ahead (each fragment is used by exactly one wave: LDS staging would only add a copy).  The accumulators (+ bias) ARE x.
//   then, for each half of 32 input rows (128 rows of x):
//   0. ELU(x) -> pieces in LDS, rows kept in PHASE-MAJOR order (plane j holds the rows 4s + j): that is the order the
//      accumulators have, and a causal tap (row t - d) is a shift inside / between planes, so all fragment reads stay contiguous;
//   B. hidden = conv k3 over ELU(x) (weights resident in LDS for the whole tile), ELU, pieces;
//   C. y = conv k1 over the hidden + x: the wave assignment matches phase A, so the residual x is the wave's own accumulator;
//      ELU(y) -> LDS (fp32, phase-major planes);
//   D. sample t = b + sum_{tap, c} w[tap][c] ELU(y)[t - 2 + tap][c] on the vector units, 4 threads per sample.
// Tile row 0 is a halo (its samples belong to the previous tile; its x / y rows are what the causal convolutions of row 1
// need), so tiles advance by 63 input rows; the input buffer carries the previous call's last two rows in front of every
// slot's rows (streaming), which makes chunked decoding equal to decoding in one call.
#include "gemm_dev.h"
#include "mimi_common.h"

namespace smoltts {

namespace {

struct LastDev {
  const float* in;     // ELU(stage-3 output), row 0 of slot 0; 2 halo rows sit in front of every slot's rows
  long in_bstride;     // floats per slot
  int T;               // input rows per slot in this call (4 T samples come out)
  const char* wt;      // ConvTranspose as GEMM matrix [256 = j * 64 + channel][256 = (row s-1 | row s) x 128], W3 tiles
  const float* bt;     // [256]
  const char* w2;      // conv k3, 64 -> 32: W3 tiles of [32][192] (k = tap * 64 + channel)
  const float* b2;
  const char* w3;      // conv k1, 32 -> 64: W3 tiles of [64][32]
  const float* b3;
  const float* wf;     // output conv k3, 64 -> 1: fp32 [3][64] (tap-major)
  float bf;
  float* pcm;
  long pcm_stride;
  const int* slot_pos;  // [slots] stream position before this call (0: rows before this call's first are padding)
};

// exp(x) - 1 with the hardware exponential: see seanet.hip (absolute error ~6e-8 on O(1) activations)
__device__ __forceinline__ float elu_hw(float x) { return x > 0.f ? x : __expf(x) - 1.0f; }

constexpr int CI = 128, C = 64, NHALF = 2, S = 32 * NHALF, MT = 2 * NHALF;
constexpr int RA = (S + 1 + 15) / 16 * 16;  // input rows s0-1 .. s0+S-1 of a tile, plane stride a multiple of 256 B (ds_read_b128 banking)
constexpr int PL = 36, PR = 4 * PL;         // ELU(x) pieces: 4 phase planes of 33 (+3) rows: rows 32h-1 .. 32h+31 of the half
constexpr int RC = 128;                     // hidden rows of a half
constexpr int AP_U4 = 3 * 4 * 4 * RA, HP_U4 = 3 * 2 * 4 * PR, VP_U4 = 3 * 4 * RC;
constexpr int REGION_U4 = AP_U4 > HP_U4 + VP_U4 ? AP_U4 : HP_U4 + VP_U4;
constexpr int W2_U4 = 2 * 6 * 192, W3_U4 = 4 * 192;
constexpr int EY_ROW = 68, EY_PL = 33;      // ELU(y): [phase j][row 0 = last row of the previous half, 1..32][64 (+4: bank spread)]
constexpr size_t LDS_BYTES = (size_t)(REGION_U4 + W2_U4 + W3_U4) * 16;
static_assert(4 * EY_PL * EY_ROW * 4 <= HP_U4 * 16, "ELU(y) lives over the (dead) ELU(x) pieces");

__global__ __launch_bounds__(512) void seanet_last_kernel(LastDev p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* ap = reinterpret_cast<uint4*>(smem);     // phase A: input pieces [piece][in chunk 4][q][RA]
  uint4* hp = reinterpret_cast<uint4*>(smem);     // per half: ELU(x) pieces [piece][x chunk 2][q][PR]
  uint4* vp = hp + HP_U4;                         // ELU(hidden) pieces [piece][q][RC]
  uint4* w2s = reinterpret_cast<uint4*>(smem) + REGION_U4;  // conv k3 tiles [col tile 2][chunk 6][piece][lane]
  uint4* w3s = w2s + W2_U4;                                 // conv k1 tiles [col tile 4][piece][lane]
  float* ey = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const int b = blockIdx.y;
  const int s0 = (int)blockIdx.x * (S - 1) - 1;   // input row of tile row 0
  const float* inb = p.in + (long)b * p.in_bstride;
  const bool stream_start = blockIdx.x == 0 && p.slot_pos[b] == 0;  // tile row 0 lies before the stream: zero padding, not data
  const int j = wave >> 1, wh = wave & 1;         // the wave's output phase and channel half, phases A, 0 and C

  // ---- load: the block's weights -> LDS (resident), the tile's input rows -> pieces
  {
    uint4 wreg[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      const int i = tid + 512 * t;
      wreg[t] = i < W2_U4 ? reinterpret_cast<const uint4*>(p.w2)[i] : reinterpret_cast<const uint4*>(p.w3)[i - W2_U4];
    }
    // 8 consecutive lanes take 8 consecutive rows of one 8-channel group: 128 contiguous LDS bytes per ds_write_b128 group
    for (int idx = tid; idx < (S + 8) * 16; idx += 512) {
      const int i = (idx >> 7) * 8 + (idx & 7), g8 = (idx >> 3) & 15;
      const int row = s0 - 1 + i;
      if (i > S) continue;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
      if (row < p.T) {
        const float* src = inb + (long)row * CI + g8 * 8;
        a = *reinterpret_cast<const float4*>(src);
        c = *reinterpret_cast<const float4*>(src + 4);
      }
      uint4 h, m, l;
      split3x8(a, c, h, m, l);
      const int slot = g8 * RA + i;  // [in chunk g8 >> 2][q = g8 & 3][row] inside a piece plane
      ap[slot] = h;
      ap[slot + 16 * RA] = m;
      ap[slot + 32 * RA] = l;
    }
#pragma unroll
    for (int t = 0; t < 6; ++t) w2s[tid + 512 * t] = wreg[t];  // (w3s follows w2s)
  }
  __syncthreads();

  // ---- phase A: x[s][n] for the wave's column tiles 2 wave, 2 wave + 1 (n = j * 64 + channel), all MT row tiles
  f32x4 acc[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const char* wsrc = p.wt + (size_t)(2 * wave) * 8 * 3072 + lane * 16;  // tile (2 wave + t, chunk kc) at + (t * 8 + kc) * 3072
    uint4 wq[2][2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) wq[0][t][pc] = *reinterpret_cast<const uint4*>(wsrc + (size_t)(t * 8) * 3072 + pc * 1024);
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) {
      if (kc + 1 < 8) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int pc = 0; pc < 3; ++pc)
            wq[(kc + 1) & 1][t][pc] = *reinterpret_cast<const uint4*>(wsrc + (size_t)(t * 8 + kc + 1) * 3072 + pc * 1024);
      }
      const int tap = kc >> 2, xc = kc & 3;  // k < 128: input row s - 1 (LDS row s_local), else row s (LDS row s_local + 1)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        uint4 xf[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) xf[pc] = ap[((pc * 4 + xc) * 4 + q) * RA + mt * 16 + r + tap];
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[mt][t] = mfma_b3(wq[kc & 1][t], xf, acc[mt][t]);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float4 bb = *reinterpret_cast<const float4*>(p.bt + (2 * wave + t) * 16 + 4 * q);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        acc[mt][t][0] += bb.x; acc[mt][t][1] += bb.y; acc[mt][t][2] += bb.z; acc[mt][t][3] += bb.w;
      }
    }
  }
  __syncthreads();  // the input pieces are dead: their LDS becomes the halves' working set

  float4 keep = make_float4(0.f, 0.f, 0.f, 0.f);  // threads 0..31: ELU(y) of the half's last input row, phases 2 and 3, for the next half
  uint2* hp2 = reinterpret_cast<uint2*>(hp);
  uint2* vp2 = reinterpret_cast<uint2*>(vp);
#pragma unroll
  for (int h = 0; h < NHALF; ++h) {
    // ---- phase 0: ELU(x) of tile rows 32h - 1 .. 32h + 31 -> pieces; the lane holds x[row mt * 16 + r][channels wh*32 + t*16 + 4q .. +4)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int k = -1; k < 2; ++k) {  // row tile 2h - 1 contributes its last row only (plane row 0)
        const int mt = 2 * h + k;
        if (k < 0 && r != 15) continue;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (mt >= 0) v = acc[mt < 0 ? 0 : mt][t];
        const bool pad = stream_start && mt == 0 && r == 0;
        const float e0 = pad ? 0.f : elu_hw(v[0]), e1 = pad ? 0.f : elu_hw(v[1]), e2 = pad ? 0.f : elu_hw(v[2]), e3 = pad ? 0.f : elu_hw(v[3]);
        uint32_t h0, m0, l0, h1, m1, l1;
        split3_pair(e0, e1, h0, m0, l0);
        split3_pair(e2, e3, h1, m1, l1);
        const int prow = j * PL + (k < 0 ? 0 : k * 16 + r + 1);
        const int s2 = (((wh * 4) + t * 2 + (q >> 1)) * PR + prow) * 2 + (q & 1);  // 8-byte half of the row's 16-byte slot
        hp2[s2] = make_uint2(h0, h1);
        hp2[s2 + 2 * 4 * PR * 2] = make_uint2(m0, m1);
        hp2[s2 + 2 * 2 * 4 * PR * 2] = make_uint2(l0, l1);
      }
    }
    __syncthreads();

    // ---- phase B: hidden rows 4s + j of the 16 input rows s = 32h + mtl * 16 + r: taps d = 0..2 read x rows 4s + j - 2 + d
    {
      const int mtl = wh;
      f32x4 hb[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int kc = 0; kc < 6; ++kc) {
        const int d = kc >> 1, xc = kc & 1;
        const int jj = j - 2 + d;  // < 0: the row lives in the previous input row's planes
        const int prow = (jj & 3) * PL + mtl * 16 + r + (jj < 0 ? 0 : 1);
        uint4 xf[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) xf[pc] = hp[((pc * 2 + xc) * 4 + q) * PR + prow];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          uint4 wf[3];
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) wf[pc] = w2s[((nt * 6 + kc) * 3 + pc) * 64 + lane];
          hb[nt] = mfma_b3(wf, xf, hb[nt]);
        }
      }
      const int row = (j * 2 + mtl) * 16 + r;  // hidden rows stay in (phase, row tile) order: conv k1 has no taps
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const float4 bb = *reinterpret_cast<const float4*>(p.b2 + nt * 16 + 4 * q);
        const float v0 = elu_hw(hb[nt][0] + bb.x), v1 = elu_hw(hb[nt][1] + bb.y), v2 = elu_hw(hb[nt][2] + bb.z), v3 = elu_hw(hb[nt][3] + bb.w);
        uint32_t h0, m0, l0, h1, m1, l1;
        split3_pair(v0, v1, h0, m0, l0);
        split3_pair(v2, v3, h1, m1, l1);
        const int s2 = ((nt * 2 + (q >> 1)) * RC + row) * 2 + (q & 1);
        vp2[s2] = make_uint2(h0, h1);
        vp2[s2 + 4 * RC * 2] = make_uint2(m0, m1);
        vp2[s2 + 2 * 4 * RC * 2] = make_uint2(l0, l1);
      }
    }
    __syncthreads();  // hidden pieces complete; nobody reads the ELU(x) pieces any more

    // ---- phase C: y = conv k1 (hidden) + x for the wave's own (phase, channel half): column tiles 2 wh, 2 wh + 1, both row tiles
    {
#pragma unroll
      for (int mtl = 0; mtl < 2; ++mtl) {
        uint4 xf[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) xf[pc] = vp[(pc * 4 + q) * RC + (j * 2 + mtl) * 16 + r];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int ct = 2 * wh + t;
          uint4 wf[3];
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) wf[pc] = w3s[(ct * 3 + pc) * 64 + lane];
          f32x4 y = mfma_b3(wf, xf, (f32x4){0.f, 0.f, 0.f, 0.f});
          const float4 bb = *reinterpret_cast<const float4*>(p.b3 + ct * 16 + 4 * q);
          const f32x4 x = acc[2 * h + mtl][t];
          float4 e = make_float4(elu_hw(y[0] + bb.x + x[0]), elu_hw(y[1] + bb.y + x[1]), elu_hw(y[2] + bb.z + x[2]), elu_hw(y[3] + bb.w + x[3]));
          if (stream_start && h == 0 && mtl == 0 && r == 0) e = make_float4(0.f, 0.f, 0.f, 0.f);  // before the stream: the output conv's zero padding
          *reinterpret_cast<float4*>(ey + (j * EY_PL + 1 + mtl * 16 + r) * EY_ROW + ct * 16 + 4 * q) = e;
        }
      }
      if (tid < 32) {  // row 0 of planes 2 and 3 = the previous half's last input row (tile row 32h - 1); unused for h = 0
        const int pj = 2 + (tid >> 4), c4 = (tid & 15) * 4;
        *reinterpret_cast<float4*>(ey + (pj * EY_PL) * EY_ROW + c4) = keep;
      }
    }
    __syncthreads();

    // ---- phase D: sample of x row 4s + jo = out bias + sum over taps of ELU(y) rows 4s + jo - 2 .. 4s + jo; 4 threads per sample
    {
      const int i = tid >> 2, part = tid & 3, sl = i >> 2, jo = i & 3;
      float sum = 0.f;
#pragma unroll
      for (int tap = 0; tap < 3; ++tap) {
        const int jj = jo - 2 + tap;
        const float* er = ey + ((jj & 3) * EY_PL + sl + (jj < 0 ? 0 : 1)) * EY_ROW + part * 16;
        const float* wr_ = p.wf + tap * C + part * 16;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
          const float4 ev = *reinterpret_cast<const float4*>(er + c4 * 4);
          const float4 wv = *reinterpret_cast<const float4*>(wr_ + c4 * 4);
          sum = fmaf(ev.x, wv.x, sum); sum = fmaf(ev.y, wv.y, sum); sum = fmaf(ev.z, wv.z, sum); sum = fmaf(ev.w, wv.w, sum);
        }
      }
      sum += __shfl_xor(sum, 1);
      sum += __shfl_xor(sum, 2);
      const int srow = s0 + 32 * h + sl;  // input row; tile row 0 belongs to the previous tile
      if (part == 0 && (h > 0 || sl > 0) && srow < p.T) p.pcm[(long)b * p.pcm_stride + 4L * srow + jo] = sum + p.bf;
      if (h + 1 < NHALF && tid < 32) keep = *reinterpret_cast<const float4*>(ey + ((2 + (tid >> 4)) * EY_PL + 32) * EY_ROW + (tid & 15) * 4);
    }
    if (h + 1 < NHALF) __syncthreads();  // the next half overwrites ELU(y) with its ELU(x) pieces
  }
}

}  // namespace

int launch_seanet_last(const MimiLastStageArgs& a, hipStream_t st) {
  ST_REQUIRE(a.in && a.wt && a.bt && a.w2 && a.b2 && a.w3 && a.b3 && a.final_w && a.pcm && a.slot_pos && a.T > 0 && a.batch > 0,
             SMOLTTS_E_INVALID, "seanet last stage: null or empty argument");
  ST_REQUIRE(a.batch <= 65535, SMOLTTS_E_INVALID, "seanet last stage: batch too large");
  static bool attr_set = false;
  if (!attr_set) {  // > 64 KB of dynamic LDS must be requested once per kernel
    ST_CHECK_HIP(hipFuncSetAttribute((const void*)seanet_last_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    attr_set = true;
  }
  LastDev d{a.in, (long)a.in_bstride, a.T, (const char*)a.wt, a.bt, (const char*)a.w2, a.b2, (const char*)a.w3, a.b3,
            a.final_w, a.final_b, a.pcm, (long)a.pcm_stride, a.slot_pos};
  const dim3 grid((a.T + S - 2) / (S - 1), a.batch);
  hipLaunchKernelGGL(seanet_last_kernel, grid, dim3(512), LDS_BYTES, st, d);
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

}  // namespace smoltts
