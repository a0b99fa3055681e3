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
// The working set (126 KB of LDS) allows one workgroup per CU, so nothing else hides a tile's dependent phases: workgroups
// are persistent (one per CU, tiles dealt round-robin), keep the block's weights, the output conv and the biases in LDS /
// registers for all their tiles, and fetch the next tile's input rows into registers under the halves of the current one.
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
  int tiles_per_slot, n_tiles;
};

// exp(x) - 1 with the hardware exponential: see seanet.hip (absolute error ~6e-8 on O(1) activations)
#ifdef SMOLTTS_DBG_LAST_NO_ELU  // (timing-only variants, tools/ab_last.sh: wrong numerics)
__device__ __forceinline__ float elu_hw(float x) { return x; }
#else
__device__ __forceinline__ float elu_hw(float x) { return x > 0.f ? x : __expf(x) - 1.0f; }
#endif

constexpr int CI = 128, C = 64, NHALF = 2, S = 32 * NHALF, MT = 2 * NHALF;
constexpr int RA = (S + 1 + 15) / 16 * 16;  // input rows s0-1 .. s0+S-1 of a tile, plane stride a multiple of 256 B (ds_read_b128 banking)
constexpr int PL = 36, PR = 4 * PL;         // ELU(x) pieces: 4 phase planes of 33 (+3) rows: rows 32h-1 .. 32h+31 of the half
constexpr int RC = 128;                     // hidden rows of a half
constexpr int AP_U4 = 3 * 4 * 4 * RA, HP_U4 = 3 * 2 * 4 * PR, VP_U4 = 3 * 4 * RC;
constexpr int REGION_U4 = AP_U4 > HP_U4 + VP_U4 ? AP_U4 : HP_U4 + VP_U4;
constexpr int W2_U4 = 2 * 6 * 192, W3_U4 = 4 * 192;
// partial sums of the output conv: [half parity][channel half][tap][phase j][row 1..32 (0 unused)] floats
constexpr int PART_PL = 33, PART_HALF = 2 * 3 * 4 * PART_PL, PART_U4 = (2 * PART_HALF + 3) / 4;
constexpr size_t LDS_BYTES = (size_t)(REGION_U4 + W2_U4 + W3_U4 + PART_U4) * 16;
constexpr int IN_IT = ((S + 8) * 16 + 511) / 512;  // (row, 8-channel group) items of a tile's input per thread

#ifdef SMOLTTS_DBG_LAST_STAMPS  // (diagnostic variant, tools/stamps_last.py: cycles per phase as seen by wave 0 of every workgroup)
__device__ unsigned long long g_last_stamps[16];
#define ST_STAMP(I) { const long long now_ = clock64(); st_sum[I] += now_ - st_last; st_last = now_; }
#else
#define ST_STAMP(I)
#endif

template <int NP>
__global__ __launch_bounds__(512) void seanet_last_kernel(LastDev p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* ap = reinterpret_cast<uint4*>(smem);     // phase A: input pieces [piece][in chunk 4][q][RA]
  uint4* hp = reinterpret_cast<uint4*>(smem);     // per half: ELU(x) pieces [piece][x chunk 2][q][PR]
  uint4* vp = hp + HP_U4;                         // ELU(hidden) pieces [piece][q][RC]
  uint4* w2s = reinterpret_cast<uint4*>(smem) + REGION_U4;  // conv k3 tiles [col tile 2][chunk 6][piece][lane]
  uint4* w3s = w2s + W2_U4;                                 // conv k1 tiles [col tile 4][piece][lane]
  float* part = reinterpret_cast<float*>(w3s + W3_U4);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform for the compiler too: scalar bases for the W3 stream
  const int r = lane & 15, q = lane >> 4;
  const int j = wave >> 1, wh = wave & 1;         // the wave's output phase and channel half, phases A, 0 and C

  // ---- once per workgroup: the block's weights and the output conv -> LDS, the biases -> registers
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    const int i = tid + 512 * t;
    w2s[i] = i < W2_U4 ? reinterpret_cast<const uint4*>(p.w2)[i] : reinterpret_cast<const uint4*>(p.w3)[i - W2_U4];  // (w3s follows w2s)
  }
  float4 bt_r[2], b2_r[2], b3_r[2], wf_r[3][2];  // wf_r: output conv weights of the lane's channels, per tap
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    bt_r[t] = *reinterpret_cast<const float4*>(p.bt + (2 * wave + t) * 16 + 4 * q);
    b2_r[t] = *reinterpret_cast<const float4*>(p.b2 + t * 16 + 4 * q);
    b3_r[t] = *reinterpret_cast<const float4*>(p.b3 + (2 * wh + t) * 16 + 4 * q);
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) wf_r[tap][t] = *reinterpret_cast<const float4*>(p.wf + tap * C + (2 * wh + t) * 16 + 4 * q);
  }

  // a tile's input rows s0 - 1 .. s0 + S - 1, in registers: 8 consecutive lanes take 8 consecutive rows of one 8-channel
  // group (128 contiguous LDS bytes per ds_write_b128 lane group when they are stored)
  float4 pre[IN_IT][2];
#define ST_FETCH_ROWS(TILE)                                                                                              \
  {                                                                                                                     \
    const int fb_ = (TILE) / p.tiles_per_slot, fs0_ = ((TILE) - fb_ * p.tiles_per_slot) * (S - 1) - 1;                  \
    const float* src_ = p.in + (long)fb_ * p.in_bstride;                                                                \
    _Pragma("unroll") for (int it_ = 0; it_ < IN_IT; ++it_) {                                                           \
      const int idx_ = tid + 512 * it_, i_ = (idx_ >> 7) * 8 + (idx_ & 7), g8_ = (idx_ >> 3) & 15, row_ = fs0_ - 1 + i_; \
      const bool on_ = i_ <= S && row_ < p.T;                                                                           \
      pre[it_][0] = on_ ? *reinterpret_cast<const float4*>(src_ + (long)row_ * CI + g8_ * 8) : make_float4(0.f, 0.f, 0.f, 0.f);     \
      pre[it_][1] = on_ ? *reinterpret_cast<const float4*>(src_ + (long)row_ * CI + g8_ * 8 + 4) : make_float4(0.f, 0.f, 0.f, 0.f); \
    }                                                                                                                   \
  }
  if ((int)blockIdx.x < p.n_tiles) ST_FETCH_ROWS((int)blockIdx.x)

  // phase 0 of half H: ELU(x) of tile rows 32H - 1 .. 32H + 31 -> pieces; the lane holds x[row mt * 16 + r][channels wh*32 + t*16 + 4q .. +4)
#define ST_PHASE0(H)                                                                                                    \
  _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                                       \
    _Pragma("unroll") for (int k = -1; k < 2; ++k) { /* row tile 2H - 1 contributes its last row only (plane row 0) */  \
      const int mt = 2 * (H) + k;                                                                                       \
      if (k < 0 && r != 15) continue;                                                                                   \
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};                                                                            \
      if (mt >= 0) v = acc[mt < 0 ? 0 : mt][t];                                                                         \
      const bool pad = stream_start && mt == 0 && r == 0;                                                               \
      const float e0 = pad ? 0.f : elu_hw(v[0]), e1 = pad ? 0.f : elu_hw(v[1]), e2 = pad ? 0.f : elu_hw(v[2]),          \
                  e3 = pad ? 0.f : elu_hw(v[3]);                                                                        \
      uint32_t h0, m0, l0, h1, m1, l1;                                                                                  \
      split3_pair(e0, e1, h0, m0, l0);                                                                                  \
      split3_pair(e2, e3, h1, m1, l1);                                                                                  \
      const int prow = j * PL + (k < 0 ? 0 : k * 16 + r + 1);                                                           \
      const int s2 = (((wh * 4) + t * 2 + (q >> 1)) * PR + prow) * 2 + (q & 1); /* 8-byte half of the row's slot */     \
      hp2[s2] = make_uint2(h0, h1);                                                                                     \
      hp2[s2 + 2 * 4 * PR * 2] = make_uint2(m0, m1);                                                                    \
      hp2[s2 + 2 * 2 * 4 * PR * 2] = make_uint2(l0, l1);                                                                \
    }                                                                                                                   \
  }

  // phase D of half H (threads 0..127, one sample each): sample of x row 4s + jo = bias + sum over the taps of the partial dot
  // products of ELU(y) rows 4s + jo - 2 .. 4s + jo (both channel halves); plane row 0 = the previous half's row 32
#define ST_PHASE_D(H, S0, B_)                                                                                           \
  if (tid < 128) {                                                                                                      \
    const int sl = tid >> 2, jo = tid & 3;                                                                              \
    float sum = p.bf;                                                                                                   \
    _Pragma("unroll") for (int tap = 0; tap < 3; ++tap) {                                                               \
      const int jj = jo - 2 + tap, row = sl + (jj < 0 ? 0 : 1);                                                         \
      const float* pp = row == 0 ? part + (((H) & 1) ^ 1) * PART_HALF + ((tap * 4 + (jj & 3)) * PART_PL + 32)           \
                                 : part + ((H) & 1) * PART_HALF + ((tap * 4 + (jj & 3)) * PART_PL + row);               \
      sum += pp[0] + pp[3 * 4 * PART_PL];                                                                               \
    }                                                                                                                   \
    const int srow = (S0) + 32 * (H) + sl; /* input row; tile row 0 belongs to the previous tile */                     \
    if (((H) > 0 || sl > 0) && srow < p.T) p.pcm[(long)(B_) * p.pcm_stride + 4L * srow + jo] = sum;                     \
  }

#ifdef SMOLTTS_DBG_LAST_STAMPS
  long long st_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = clock64();
#endif
  int prev_s0 = 0, prev_b = -1;
  for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
  const int b = tile / p.tiles_per_slot;
  const int ti = tile - b * p.tiles_per_slot;
  const int s0 = ti * (S - 1) - 1;                // input row of tile row 0
  const bool stream_start = ti == 0 && p.slot_pos[b] == 0;  // tile row 0 lies before the stream: zero padding, not data

  // ---- the previous tile's last samples (their partial sums are complete since the barrier that ended its loop body)
#ifndef SMOLTTS_DBG_LAST_NO_HALVES
  if (prev_b >= 0) ST_PHASE_D(NHALF - 1, prev_s0, prev_b)
#endif
  // ---- the tile's input rows -> pieces (the previous tile is done with the LDS)
#pragma unroll
  for (int it = 0; it < IN_IT; ++it) {
    const int idx = tid + 512 * it, i = (idx >> 7) * 8 + (idx & 7), g8 = (idx >> 3) & 15;
    if (i > S) continue;
    uint4 h, m, l;
    split3x8(pre[it][0], pre[it][1], h, m, l);
    const int slot = g8 * RA + i;  // [in chunk g8 >> 2][q = g8 & 3][row] inside a piece plane
    ap[slot] = h;
    ap[slot + 16 * RA] = m;
    ap[slot + 32 * RA] = l;
  }
  ST_STAMP(0)
  lds_barrier();
  ST_STAMP(1)

  // ---- phase A: x[s][n] for the wave's column tiles 2 wave, 2 wave + 1 (n = j * 64 + channel), all MT row tiles
  f32x4 acc[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const char* wtile = p.wt + (size_t)(2 * wave) * 8 * 3072;  // (uniform) tile (2 wave + t, chunk kc) at + (t * 8 + kc) * 3072
    const int wlane = lane * 16;
    uint4 wq[2][2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) wq[0][t][pc] = *reinterpret_cast<const uint4*>(wtile + (size_t)(t * 8) * 3072 + pc * 1024 + wlane);
#ifndef SMOLTTS_DBG_LAST_NO_A  // (timing-only variants, tools/ab_last.sh: wrong numerics)
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) {
      if (kc + 1 < 8) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int pc = 0; pc < 3; ++pc)
            wq[(kc + 1) & 1][t][pc] = *reinterpret_cast<const uint4*>(wtile + (size_t)(t * 8 + kc + 1) * 3072 + pc * 1024 + wlane);
      }
      // the two waves of a SIMD (w, w + 4) take turns at the higher issue priority: with a fixed order the older wave runs
      // ahead and the younger one finishes the phase alone, without a partner to cover its LDS / L2 waits
      if (((kc ^ (wave >> 2)) & 1) != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
      const int tap = kc >> 2, xc = kc & 3;  // k < 128: input row s - 1 (LDS row s_local), else row s (LDS row s_local + 1)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        uint4 xf[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) xf[pc] = ap[((pc * 4 + xc) * 4 + q) * RA + mt * 16 + r + tap];
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[mt][t] = mfma_b3<NP>(wq[kc & 1][t], xf, acc[mt][t]);
      }
    }
#endif
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float4 bb = bt_r[t];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        acc[mt][t][0] += bb.x; acc[mt][t][1] += bb.y; acc[mt][t][2] += bb.z; acc[mt][t][3] += bb.w;
      }
    }
  }
  __builtin_amdgcn_s_setprio(0);
  ST_STAMP(2)
  lds_barrier();  // the input pieces are dead: their LDS becomes the halves' working set
  ST_STAMP(3)
  // the next tile's input rows: requested after phase A's last W3 load (loads return in order: in front of them they would
  // hold up every fragment wait), consumed at the top of the next tile -- in flight across the halves' (LDS-only) barriers
  if (tile + (int)gridDim.x < p.n_tiles) ST_FETCH_ROWS(tile + (int)gridDim.x)

  uint2* hp2 = reinterpret_cast<uint2*>(hp);
  uint2* vp2 = reinterpret_cast<uint2*>(vp);

#ifdef SMOLTTS_DBG_LAST_NO_HALVES
  if (acc[0][0][0] == 12345.f) p.pcm[tid] = acc[1][1][1] + acc[2][0][2] + acc[3][1][3];
  lds_barrier();
#else
  ST_PHASE0(0)
  ST_STAMP(4)
  lds_barrier();
  ST_STAMP(5)
#pragma unroll
  for (int h = 0; h < NHALF; ++h) {
    // ---- phase B: hidden rows 4s + j of the 16 input rows s = 32h + mtl * 16 + r: taps d = 0..2 read x rows 4s + j - 2 + d
    if (h > 0) ST_PHASE_D(h - 1, s0, b)  // (the previous half's samples: its partial sums are complete since the last barrier)
    {
      const int mtl = wh;
      f32x4 hb[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#ifndef SMOLTTS_DBG_LAST_NO_B
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
          hb[nt] = mfma_b3<NP>(wf, xf, hb[nt]);
        }
      }
#endif
      const int row = (j * 2 + mtl) * 16 + r;  // hidden rows stay in (phase, row tile) order: conv k1 has no taps
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const float4 bb = b2_r[nt];
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
    ST_STAMP(6)
    lds_barrier();  // hidden pieces complete; nobody reads the ELU(x) pieces any more
    ST_STAMP(7)

    // ---- phase C: y = conv k1 (hidden) + x for the wave's own (phase, channel half): column tiles 2 wh, 2 wh + 1, both row tiles;
    //      the next half's phase 0 (vector units only, writes the now idle ELU(x) planes) sits between the MFMAs and their use
    {
      f32x4 y[2][2];
#pragma unroll
      for (int mtl = 0; mtl < 2; ++mtl) {
        uint4 xf[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) xf[pc] = vp[(pc * 4 + q) * RC + (j * 2 + mtl) * 16 + r];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          uint4 wf[3];
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) wf[pc] = w3s[((2 * wh + t) * 3 + pc) * 64 + lane];
          y[mtl][t] = mfma_b3<NP>(wf, xf, (f32x4){0.f, 0.f, 0.f, 0.f});
        }
      }
      if (h + 1 < NHALF) ST_PHASE0(h + 1)
      // ELU(y) never leaves the registers: the output conv's three taps as partial dot products over the lane's 4 channels,
      // summed over the wave's 32 channels (2 column tiles in registers, 4 lane groups by shuffle) -> part[half][tap][phase][row]
#pragma unroll
      for (int mtl = 0; mtl < 2; ++mtl) {
        float pk[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float4 bb = b3_r[t];
          const f32x4 x = acc[2 * h + mtl][t];
          float e[4] = {elu_hw(y[mtl][t][0] + bb.x + x[0]), elu_hw(y[mtl][t][1] + bb.y + x[1]), elu_hw(y[mtl][t][2] + bb.z + x[2]),
                        elu_hw(y[mtl][t][3] + bb.w + x[3])};
#pragma unroll
          for (int tap = 0; tap < 3; ++tap) {
            const float4 w = wf_r[tap][t];
            pk[tap] = fmaf(e[0], w.x, pk[tap]); pk[tap] = fmaf(e[1], w.y, pk[tap]);
            pk[tap] = fmaf(e[2], w.z, pk[tap]); pk[tap] = fmaf(e[3], w.w, pk[tap]);
          }
        }
        const bool pad = stream_start && h == 0 && mtl == 0 && r == 0;  // before the stream: the output conv's zero padding
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
          float v = pk[tap];
          v += __shfl_xor(v, 16);
          v += __shfl_xor(v, 32);
          if (q == 0) part[(h & 1) * PART_HALF + (wh * 3 + tap) * 4 * PART_PL + j * PART_PL + 1 + mtl * 16 + r] = pad ? 0.f : v;
        }
      }
    }
    ST_STAMP(8)
    lds_barrier();
    ST_STAMP(9)
  }
#endif
  prev_s0 = s0; prev_b = b;
  }  // tiles
#ifndef SMOLTTS_DBG_LAST_NO_HALVES
  if (prev_b >= 0) ST_PHASE_D(NHALF - 1, prev_s0, prev_b)
#endif
#ifdef SMOLTTS_DBG_LAST_STAMPS
  if (tid == 0)
    for (int i = 0; i < 12; ++i) atomicAdd(&g_last_stamps[i], (unsigned long long)st_sum[i]);
#endif
#undef ST_FETCH_ROWS
#undef ST_PHASE0
#undef ST_PHASE_D
}

}  // namespace

#ifdef SMOLTTS_DBG_LAST_STAMPS
extern "C" int smoltts_debug_last_stamps(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_last_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_last_stamps), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

int launch_seanet_last(const MimiLastStageArgs& a, hipStream_t st) {
  ST_REQUIRE(a.in && a.wt && a.bt && a.w2 && a.b2 && a.w3 && a.b3 && a.final_w && a.pcm && a.slot_pos && a.T > 0 && a.batch > 0,
             SMOLTTS_E_INVALID, "seanet last stage: null or empty argument");
  ST_REQUIRE(a.batch <= 65535, SMOLTTS_E_INVALID, "seanet last stage: batch too large");
  static PerDevice attr;
  const int dev = PerDevice::current();
  if (!attr.done(dev)) {  // > 64 KB of dynamic LDS must be requested once per kernel and device
    ST_CHECK_HIP(hipFuncSetAttribute((const void*)seanet_last_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    ST_CHECK_HIP(hipFuncSetAttribute((const void*)seanet_last_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    attr.mark_done(dev);
  }
  const int tiles_per_slot = (a.T + S - 2) / (S - 1);
  ST_REQUIRE((long)tiles_per_slot * a.batch < (1L << 30), SMOLTTS_E_INVALID, "seanet last stage: too many rows for one launch");
  LastDev d{a.in, (long)a.in_bstride, a.T, (const char*)a.wt, a.bt, (const char*)a.w2, a.b2, (const char*)a.w3, a.b3,
            a.final_w, a.final_b, a.pcm, (long)a.pcm_stride, a.slot_pos, tiles_per_slot, tiles_per_slot * a.batch};
  const int n_cu = device_cu_count();
  const dim3 grid(d.n_tiles < n_cu ? d.n_tiles : n_cu);  // persistent: the LDS footprint admits exactly one workgroup per CU
  if (a.b3_products == 3) hipLaunchKernelGGL(seanet_last_kernel<3>, grid, dim3(512), LDS_BYTES, st, d);
  else hipLaunchKernelGGL(seanet_last_kernel<6>, grid, dim3(512), LDS_BYTES, st, d);
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

}  // namespace smoltts
