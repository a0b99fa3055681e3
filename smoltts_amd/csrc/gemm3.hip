// bf16-MFMA skinny GEMM of the DualAR transformer:  out[M,N] = epi( X[M,K] . W[N,K]^T ), where the
// activation operand arrives pre-split in the X3 format (x3.h: hi+mid+lo bf16 pieces == the fp32
// value, already multiplied by the RMSNorm weight) and the bf16 weights in T16x32 tiles.
//
// Reference ops: the nn.Linear calls of modeling/model/rq_transformer.py (wqkv :545, wo :570,
// w1/w3/w2 :582, tied head :253, depthwise head :598) with their RMSNorm (:607-613), RoPE (:627-640),
// SiLU-gate and residual adds fused around them.
//
// Why this shape (measured on MI355X, tools/microbench_gemm.py): at 32 rows the fp32-input MFMA
// (gemm.hip) is matrix-pipe bound (8 x 32-cycle MFMAs per 32-k chunk and 16-row tile) and every
// workgroup re-converts the same activations.  Here one chunk costs 3 x 16-cycle bf16 MFMAs with no
// VALU work at all on the operand path: both fragments are loaded with one coalesced 16-byte-per-
// lane load each and fed straight to v_mfma_f32_16x16x32_bf16.  Products are exact (8-bit x 8-bit
// significands), accumulation is fp32, so results stay within fp32 rounding of the CPU oracle.
//
// Work split: workgroup = T adjacent 16-column weight tiles x MT 16-row tiles x all of K; its
// waves split K chunk-wise, keep the activation fragments in registers across the T tiles, and
// their partial tiles are summed through LDS in fixed wave order (deterministic, no atomics).
// RMSNorm: the producer of x also publishes per-(row, 16-column) partial sums of squares; the
// consumer adds them in fixed order and scales its result rows by rsqrt(mean + eps).
// Epilogues write what the *next* kernel consumes: fp32 residual stream + X3 operand(s) + partial
// sums of squares (wo / w2), the SwiGLU product as X3 (w1|w3), RoPE'd q + KV-cache rows (wqkv).
#include <stdlib.h>

#include "argmax_dev.h"  // Top2 / top2_merge / rope_gathered4 (and x3.h)
#ifndef SMOLTTS_DBG_PIECES
#define SMOLTTS_DBG_PIECES 3
#endif
#ifndef SMOLTTS_DBG_PIECES_RESID  // timing-only variants (tools/ab_pieces.sh): activation pieces read by the residual GEMMs (wo, w2) alone
#define SMOLTTS_DBG_PIECES_RESID SMOLTTS_DBG_PIECES
#endif

namespace smoltts {

struct Gemm3Dev {
  const char* w;
  const float* wscale;  // fp8 weights: per-output-row scale [N]; nullptr = bf16 weights
  const char* x3;
  int M, N, K;
  int w_nt;       // weight loads carry the non-temporal hint (SmolttsGemm3Args.w_stream)
  int f8_act;     // many-row calls with fp8 weights: fp8 x fp8 MFMA on the activation's hi piece rounded to e4m3 (fp8_activations)
  int half_rows;  // MT == 1 only: a workgroup owns 8 of the tile's 16 rows (twice the workgroups, half the X3 bytes each)
  const float* ssq_in;
  float eps;
  const float* bias;
  const float* resid;
  float* out;
  long ldo;
  char* x3_out;  // SWIGLU
  EmitDev emit;
  const float* rope;
  const int* row_pos;
  const int* row_slot;
  float* kc;   // fp32, or bf16 when kv_bf16 (element addressing is the same, the element is 2 bytes)
  float* vc;
  int kv_bf16;
  char* v_x3;  // QKV_ROPE, rows at position 0: attention over one key is V itself -> V published as wo's X3 operand
  int n_q_heads, n_kv_heads, cache_len;
  float* cand;      // EPI_STORE: per (row, 16-column tile) (max, first column of it, runner-up, -) for a later pick; nullptr = off
  // attn_wo_kernel<.., PICK>: the previous depth step's code picked in front of the attention (SmolttsPickArgs)
  const float* pk_cand; int pk_tiles;
  const float* pk_table; const float* pk_rope;
  const uint16_t* pk_emb; int pk_off;
  int* pk_ids; int pk_ids_stride;
  float* pk_margin; const int* pk_mask; int* pk_margin_at; const int* pk_frames; int pk_step;
  const float* aq;  // attn_wo_kernel: q rows [M][n_q_heads * 64] of the attention recomputed in front of the K loop ...
  int a_pos;        // ... every row r being slot r at this position of the <= 8-entry caches kc / vc (a depth step)
  unsigned long long* stamps;  // diagnostics only (smoltts_debug_set_stamps); nullptr in production
};

// 64-byte lines of kernel arguments behind the first: 7 leading scalars (0x28 bytes) + Gemm3Dev
constexpr int G3_KERNARG_LINES = (0x28 + (int)sizeof(Gemm3Dev) - 1) / 64;
constexpr int AWO_KERNARG_LINES = (0x38 + (int)sizeof(Gemm3Dev) - 1) / 64;  // attn_wo_kernel: 14 dwords in front

#ifdef SMOLTTS_DEBUG_HOOKS  // diagnostic builds only (python -m smoltts_amd.build --variant hooks): never in the product library
#define STAMP3(k)                                                                             \
  do {                                                                                        \
    if (p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {                        \
      p.stamps[(wave * 8 + (k)) * 2] = clock64();                                             \
      p.stamps[(wave * 8 + (k)) * 2 + 1] = wall_clock64();                                    \
    }                                                                                         \
  } while (0)
#else
#define STAMP3(k) do { } while (0)
#endif

// Weight tiles are read once per launch by the one or two workgroups that own them: with
// SMOLTTS_NT_W the loads carry the non-temporal hint (MI355X_MICROARCH.md 'nt-weights').
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ uint4 load_w16(const char* ptr) {
  if (NT) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(ptr));
    return make_uint4(v[0], v[1], v[2], v[3]);
  }
  return *reinterpret_cast<const uint4*>(ptr);
}

// fp8 weights (e4m3, per-row scale applied in the epilogue): a lane's 8 bytes -> the bf16x8 A fragment.
// gfx950 converts two e4m3 bytes straight to a bf16 pair (v_cvt_scalef32_pk_bf16_fp8, scale 1.0: e4m3's 3 mantissa bits fit
// bf16's 7, so the conversion is exact): 4 VALU instructions per fragment on the operand path (round 3 went through fp32:
// 4 x v_cvt_pk_f32_fp8 + 8 shift / and-or instructions).
__device__ __forceinline__ uint32_t fp8x2_to_bf16x2(uint32_t v, bool upper) {
  return upper ? __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v, 1.0f, true))
               : __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v, 1.0f, false));
}
__device__ __forceinline__ bf16x8_t fp8x8_to_bf16x8(uint32_t lo, uint32_t hi) {
  uint4 o;
  o.x = fp8x2_to_bf16x2(lo, false);
  o.y = fp8x2_to_bf16x2(lo, true);
  o.z = fp8x2_to_bf16x2(hi, false);
  o.w = fp8x2_to_bf16x2(hi, true);
  return __builtin_bit_cast(bf16x8_t, o);
}
template <bool W8, bool NT = false>
__device__ __forceinline__ uint4 load_wfrag(const char* ptr) {  // ptr already includes the lane offset
  if (W8) {
    if (NT) {
      const u32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(ptr));
      return make_uint4(v[0], v[1], 0, 0);
    }
    const uint2 v = *reinterpret_cast<const uint2*>(ptr);
    return make_uint4(v.x, v.y, 0, 0);
  }
  return load_w16<NT>(ptr);
}
template <bool W8>
__device__ __forceinline__ bf16x8_t wfrag_bf16(uint4 v) {
  return W8 ? fp8x8_to_bf16x8(v.x, v.y) : __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ float silu3(float x) { return x / (1.f + expf(-x)); }

// NW != 0 ("FULL", MT == 1): NW = 8 or 6 waves with exactly U chunks each and whole groups of T column tiles -- every operand load of the kernel is
// valid, so the load phase carries no predicates (none of the zero fills, exec masks and branches of the general form: a third of
// the instructions in front of the last load) and the 8 partial tiles are summed without selects.  Same loads, same sums.
template <int MT, int T, int U, int EPI, bool W8, bool NT = false, int NW = 0>  // NT: weight loads with the non-temporal hint (Gemm3Dev.w_nt)
__global__ __launch_bounds__(512) void gemm3_kernel(const char* w_, const char* x3_, int M_, int N_, int K_, int half_rows_, int nwaves, Gemm3Dev p) {
  // what the first loads need comes as leading scalar arguments (preloaded into SGPRs at wave launch: build.py's
  // -amdgpu-kernarg-preload-count), the rest is read from the kernarg segment behind them (common.h: kernarg_touch)
  constexpr bool FULL = NW != 0;
  KernargTouch<G3_KERNARG_LINES> kt;
  kernarg_touch(kt);
  p.w = w_; p.x3 = x3_; p.M = M_; p.N = N_; p.K = K_; p.half_rows = half_rows_;
  constexpr int WTILE = W8 ? 512 : 1024, WLANE = W8 ? 8 : 16;  // bytes per weight tile-chunk / per lane
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const int ng = blockIdx.x;
  const int mg = p.half_rows ? (blockIdx.y >> 1) : blockIdx.y;
  const bool row_on = !p.half_rows || ((r >> 3) == (int)(blockIdx.y & 1));  // this lane's row belongs to the workgroup
  const int nchunks = p.K >> 5;
  constexpr bool kResid = EPI == SMOLTTS_EPI_RESID;
  constexpr bool kEmits = EPI == SMOLTTS_EPI_RESID || EPI == SMOLTTS_EPI_STORE;
  constexpr bool kRope = EPI == SMOLTTS_EPI_QKV_ROPE;
  constexpr int kPieces = kResid ? SMOLTTS_DBG_PIECES_RESID : SMOLTTS_DBG_PIECES;  // 3 in the product
  STAMP3(0);

  // Finishing waves: one per (column tile, row tile) of the workgroup -- wave f finishes tile tf = f / MT of row tile f % MT, so
  // the T tile epilogues of a workgroup (w1|w3: three SwiGLU tiles) run side by side on different SIMDs instead of one after
  // the other in wave 0.  Each loads the epilogue inputs of ITS tile only (the same loads in total); the RMSNorm row scale is
  // worked out once per row tile, by tile 0's finisher, and handed over through LDS (no second read of the sums of squares).
  constexpr int NF = MT * T;
  const bool fin = wave < NF;
  const int fmt = wave % MT, tf = fin ? wave / MT : 0;
  const int m = (mg * MT + fmt) * 16 + r;  // meaningful for fin waves only
  const bool mvalid = fin && m < p.M && row_on;
  f32x4 acc[T][MT];
  const char* xb[MT];
  bool xv[MT];
  const char* wb[T];
  bool wv[T];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int mtile = mg * MT + mt;
    xv[mt] = mtile * 16 < p.M && row_on;
    // FULL: a lane whose row belongs to the tile's other half reads the row of its partner lane (r ^ 8: the same lines, nothing
    // more leaves L2) instead of being masked off; the column of the product it gets is never stored (row_on)
    const int xlane = FULL ? ((lane & ~8) | (p.half_rows ? (int)(blockIdx.y & 1) << 3 : (lane & 8))) : lane;
    xb[mt] = p.x3 + (size_t)mtile * nchunks * 3072 + xlane * 16;
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int ntile = ng * T + t;
    wv[t] = ntile * 16 < p.N;
    wb[t] = p.w + (size_t)ntile * nchunks * WTILE + lane * WLANE;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  uint4 xf[U][MT][3];
  uint4 wf[U][T];
  auto load_group = [&](int c0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * (FULL ? NW : nwaves);
      const bool cv = FULL || c < nchunks;
#pragma unroll
      for (int t = 0; t < T; ++t)
        wf[u][t] = (FULL || (cv && wv[t])) ? load_wfrag<W8, NT>(wb[t] + (size_t)c * WTILE) : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int pc = 0; pc < kPieces; ++pc)
          xf[u][mt][pc] = (FULL || (cv && xv[mt])) ? *reinterpret_cast<const uint4*>(xb[mt] + (size_t)c * 3072 + pc * 1024)
                                                   : make_uint4(0, 0, 0, 0);
      }
    }
  };
  // the operand stream goes out first ...
  load_group(wave);
  kernarg_touched(kt);

  // ... then the epilogue inputs of the finishing waves, all issued here so that the tail of the
  // kernel waits on nothing
  int pos = 0, slot = 0;
  float ssv[16];
  float4 rr, bb, ga, gb, ws;
  rr = bb = make_float4(0.f, 0.f, 0.f, 0.f);
  ga = gb = ws = make_float4(1.f, 1.f, 1.f, 1.f);
  const bool norm_wave = fin && tf == 0;  // reads the sums of squares of its row tile
  if (fin) {  // wave-uniform branches, clamped (always valid) addresses: no per-lane control flow
    const int mc = m < p.M ? m : p.M - 1;
    if (kRope) { pos = p.row_pos[mc]; slot = p.row_slot[mc]; }
    const int nt_in = p.K >> 4;
#pragma unroll
    for (int j = 0; j < 16; ++j) ssv[j] = 0.f;
    if (p.ssq_in != nullptr && norm_wave) {
      const float* sp = p.ssq_in + (size_t)mc * nt_in;
      // lane q sums entries q, q + 4, ..: for the usual widths (K = 768: 48 entries, K >= 1024: the first 64) every lane has the
      // same number of them -- one address and constant offsets, no clamps or selects; anything else takes the general form
      if (nt_in == 48) {
#pragma unroll
        for (int j = 0; j < 12; ++j) ssv[j] = sp[q + 4 * j];
      } else if (nt_in >= 64) {
#pragma unroll
        for (int j = 0; j < 16; ++j) ssv[j] = sp[q + 4 * j];
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int i = q + 4 * j;
          const float vld = sp[i < nt_in ? i : nt_in - 1];
          ssv[j] = i < nt_in ? vld : 0.f;
        }
      }
    }
    {
      const int n0r = (ng * T + tf) * 16 + q * 4;
      const int n0 = n0r < p.N ? n0r : p.N - 4;  // N % 4 == 0
      if (W8) ws = *reinterpret_cast<const float4*>(p.wscale + n0);
      if (kResid) rr = *reinterpret_cast<const float4*>(p.resid + (long)mc * p.ldo + n0);
      if (p.bias != nullptr) bb = *reinterpret_cast<const float4*>(p.bias + n0);
      if (kEmits && p.emit.x3a && p.emit.gamma_a) ga = *reinterpret_cast<const float4*>(p.emit.gamma_a + n0);
      if (kEmits && p.emit.x3b && p.emit.gamma_b) gb = *reinterpret_cast<const float4*>(p.emit.gamma_b + n0);
    }
  }

  STAMP3(5);  // every load of the kernel has been issued
  for (int c0 = wave; c0 < nchunks;) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const bf16x8_t a = wfrag_bf16<W8>(wf[u][t]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
          for (int pc = 0; pc < kPieces; ++pc)
            acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8_t, xf[u][mt][pc]), acc[t][mt], 0, 0, 0);
        }
      }
    }
    if (FULL) break;  // nchunks == NW U: the one group was all
    c0 += nwaves * U;
    if (c0 < nchunks) load_group(c0);
  }

  STAMP3(1);
  // RoPE rows of the finishing waves (needs pos, which arrived long ago); in flight across the barrier
  float4 cs = make_float4(1.f, 0.f, 1.f, 0.f);
  if (kRope && fin) {
    const int n0 = (ng * T + tf) * 16 + q * 4;
    const bool rot = mvalid && n0 < (p.n_q_heads + p.n_kv_heads) * 64 && pos >= 0 && pos < p.cache_len;  // the table has >= cache_len rows
    if (rot) cs = *reinterpret_cast<const float4*>(p.rope + ((long)pos * 32 + ((n0 & 63) >> 1)) * 2);
  }

  // ---- cross-wave reduction: red4[((wave*T + t)*MT + mt)*64 + lane]
  float4* red4 = reinterpret_cast<float4*>(smem);
#pragma unroll
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      red4[((wave * T + t) * MT + mt) * 64 + lane] = make_float4(acc[t][mt][0], acc[t][mt][1], acc[t][mt][2], acc[t][mt][3]);
  }
  // RMSNorm row scale from the producer's partial sums of squares (fixed order), by the row tile's first finisher, before the
  // barrier: the other finishers of the row tile pick it up from LDS behind it
  float* rstd_lds = smem + (size_t)nwaves * T * MT * 256;  // [MT][16] behind the partial tiles
  float rstd = 1.f;
  if (p.ssq_in != nullptr && norm_wave) {
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) sq += ssv[j];
    const int nt_in = p.K >> 4;
    if (nt_in > 64 && mvalid)
      for (int i = 64 + q; i < nt_in; i += 4) sq += p.ssq_in[(size_t)m * nt_in + i];
    sq = sum_xor16_32(sq);
    rstd = 1.0f / sqrtf(sq / (float)p.K + p.eps);
    if (NF > MT && q == 0) rstd_lds[fmt * 16 + r] = rstd;
  }
  STAMP3(2);
  __syncthreads();
  STAMP3(3);
  if (!fin) return;
  const int mt = fmt;  // lane, r, q keep their meaning
  if (NF > MT && p.ssq_in != nullptr && !norm_wave) rstd = rstd_lds[fmt * 16 + r];
  if (kRope && !mvalid) pos = -1;

  {
    const int t = tf;
    float4 part[8];  // workgroups have at most 8 waves (launch3_epi)
#pragma unroll
    for (int w = 0; w < 8; ++w) part[w] = red4[((((FULL ? w < NW : w < nwaves) ? w : 0) * T + t) * MT + mt) * 64 + lane];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const bool on = FULL ? w < NW : w < nwaves;  // (a wave that is not there adds + 0.0: the sum of the NW partial tiles, in order, either way)
      v[0] += on ? part[w].x : 0.f;
      v[1] += on ? part[w].y : 0.f;
      v[2] += on ? part[w].z : 0.f;
      v[3] += on ? part[w].w : 0.f;
    }
    const int ntile = ng * T + t;
    const int n0 = ntile * 16 + q * 4;
    const bool valid = mvalid && n0 < p.N;  // N % 4 == 0
    if (W8) { v[0] *= ws.x; v[1] *= ws.y; v[2] *= ws.z; v[3] *= ws.w; }
    v[0] = v[0] * rstd + bb.x; v[1] = v[1] * rstd + bb.y; v[2] = v[2] * rstd + bb.z; v[3] = v[3] * rstd + bb.w;

    if (EPI == SMOLTTS_EPI_STORE || EPI == SMOLTTS_EPI_RESID) {
      if (kResid) { v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w; }
      if (valid) {
        *reinterpret_cast<float4*>(p.out + (long)m * p.ldo + n0) = make_float4(v[0], v[1], v[2], v[3]);
        if (p.emit.x3a) x3_emit4(p.emit.x3a, m, n0, p.N >> 5, v[0] * ga.x, v[1] * ga.y, v[2] * ga.z, v[3] * ga.w);
        if (p.emit.x3b) x3_emit4(p.emit.x3b, m, n0, p.N >> 5, v[0] * gb.x, v[1] * gb.y, v[2] * gb.z, v[3] * gb.w);
      }
      if (p.emit.ssq) {  // all 64 lanes take part in the shuffles
        float sq = valid ? ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) : 0.f;
        sq = sum_xor16_32(sq);
        if (q == 0 && mvalid && ntile * 16 < p.N) p.emit.ssq[(size_t)m * (p.N >> 4) + ntile] = sq;
      }
      if (EPI == SMOLTTS_EPI_STORE && p.cand) {  // (wave-uniform) the tile's top-2 per row: what a greedy pick needs of these logits
        Top2 t{-INFINITY, 0x7fffffff, -INFINITY};
        if (valid) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (v[i] > t.v1) { t.v2 = t.v1; t.v1 = v[i]; t.i1 = n0 + i; }  // strictly greater: the first of equal columns stays
            else if (v[i] > t.v2) t.v2 = v[i];
          }
        }
        t = top2_swap<false>(t, lane);  // the 4 lanes (q) that hold the row's 16 columns: lane ^ 16, then lane ^ 32
        t = top2_swap<true>(t, lane);
        if (q == 0 && mvalid && ntile * 16 < p.N)
          *reinterpret_cast<float4*>(p.cand + ((size_t)m * ((p.N + 15) >> 4) + ntile) * 4) = make_float4(t.v1, __int_as_float(t.i1), t.v2, 0.f);
      }
    } else if (EPI == SMOLTTS_EPI_SWIGLU) {
      if (valid) x3_emit2(p.x3_out, m, n0 >> 1, p.N >> 6, silu3(v[0]) * v[1], silu3(v[2]) * v[3]);
    } else if (EPI == SMOLTTS_EPI_QKV_ROPE) {
      if (valid) {
        const int qd = p.n_q_heads * 64, kd = p.n_kv_heads * 64;
        if (n0 < qd + kd) {
          const float o0 = v[0] * cs.x - v[1] * cs.y, o1 = v[1] * cs.x + v[0] * cs.y;
          const float o2 = v[2] * cs.z - v[3] * cs.w, o3 = v[3] * cs.z + v[2] * cs.w;
          v[0] = o0; v[1] = o1; v[2] = o2; v[3] = o3;
        }
        const float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (n0 < qd) {
          *reinterpret_cast<float4*>(p.out + (long)m * p.ldo + n0) = o;
        } else if (pos >= 0 && pos < p.cache_len) {
          const int nn = n0 - qd;
          float* base = nn < kd ? p.kc : p.vc;
          const int h = (nn < kd ? nn : nn - kd) >> 6, d = nn & 63;
          const long ce = (((long)slot * p.n_kv_heads + h) * p.cache_len + pos) * 64 + d;
          if (p.kv_bf16) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + ce) = make_uint2(pack_bf16(o.x, o.y), pack_bf16(o.z, o.w));
          else *reinterpret_cast<float4*>(base + ce) = o;
          if (p.v_x3 != nullptr && nn >= kd) {
            const int G = p.n_q_heads / p.n_kv_heads;
            for (int g = 0; g < G; ++g) x3_emit4(p.v_x3, m, (h * G + g) * 64 + d, (p.n_q_heads * 64) >> 5, v[0], v[1], v[2], v[3]);
          }
        }
      }
    }
  }
  STAMP3(4);
}

// ---------------------------------------------------------------------------------------------
// Many-row variant (prompt prefill: M = all prompt tokens of all utterances): a workgroup owns 64
// rows x 16*NTW*4 columns.  The X3 blocks of a 32-k chunk (4 row tiles x 3 pieces = 12 KiB, already in
// B-fragment order) are copied into LDS once per chunk and shared by the 4 waves, each of which holds
// NTW column tiles and accumulates the whole K itself: no split-K, no cross-wave reduction, the
// activation operand leaves L2 once per 256 output columns instead of once per 16.  (Kernel below its epilogue.)

// Epilogue of the many-row kernels, straight from the accumulators (inputs loaded here: amortised over the long K loop).
template <int NTW, int EPI, bool W8>
__device__ __forceinline__ void rows_epilogue(const Gemm3Dev& p, f32x4 (&acc)[NTW][4], int mtile0, int tile0, int r, int q) {
  constexpr bool kResid = EPI == SMOLTTS_EPI_RESID;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = (mtile0 + mt) * 16 + r;
    const bool mvalid = m < p.M;
    const int mc = mvalid ? m : p.M - 1;
    float rstd = 1.f;
    if (p.ssq_in != nullptr) {
      const int nt_in = p.K >> 4;
      const float* ssp = p.ssq_in + (size_t)mc * nt_in;
      float s = 0.f;
      for (int i = q; i < nt_in; i += 4) s += ssp[i];
      s = sum_xor16_32(s);
      rstd = 1.0f / sqrtf(s / (float)p.K + p.eps);
    }
    int pos = 0, slot = 0;
    if (EPI == SMOLTTS_EPI_QKV_ROPE) { pos = p.row_pos[mc]; slot = p.row_slot[mc]; }
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int ntile = tile0 + t;
      const int n0 = ntile * 16 + q * 4;
      const bool valid = mvalid && n0 < p.N;
      const int n0c = n0 < p.N ? n0 : p.N - 4;
      float v[4] = {acc[t][mt][0], acc[t][mt][1], acc[t][mt][2], acc[t][mt][3]};
      if (W8) {
        const float4 sc = *reinterpret_cast<const float4*>(p.wscale + n0c);
        v[0] *= sc.x; v[1] *= sc.y; v[2] *= sc.z; v[3] *= sc.w;
      }
      v[0] *= rstd; v[1] *= rstd; v[2] *= rstd; v[3] *= rstd;
      if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n0c);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      }
      if (EPI == SMOLTTS_EPI_STORE || EPI == SMOLTTS_EPI_RESID) {
        if (kResid) {
          const float4 rr = *reinterpret_cast<const float4*>(p.resid + (long)mc * p.ldo + n0c);
          v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
        }
        if (valid) {
          *reinterpret_cast<float4*>(p.out + (long)m * p.ldo + n0) = make_float4(v[0], v[1], v[2], v[3]);
          emit_x4(p.emit, m, n0, p.N >> 5, v[0], v[1], v[2], v[3]);
        }
        if (p.emit.ssq) {  // all 64 lanes take part in the shuffles
          float s = valid ? ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) : 0.f;
          s = sum_xor16_32(s);
          if (q == 0 && mvalid && ntile * 16 < p.N) p.emit.ssq[(size_t)m * (p.N >> 4) + ntile] = s;
        }
      } else if (EPI == SMOLTTS_EPI_SWIGLU) {
        if (valid) x3_emit2(p.x3_out, m, n0 >> 1, p.N >> 6, silu3(v[0]) * v[1], silu3(v[2]) * v[3]);
      } else if (EPI == SMOLTTS_EPI_QKV_ROPE) {
        if (valid) {
          const int qd = p.n_q_heads * 64, kd = p.n_kv_heads * 64;
          if (n0 < qd + kd && pos >= 0 && pos < p.cache_len) {  // the table has >= cache_len rows
            const float4 cs = *reinterpret_cast<const float4*>(p.rope + ((long)pos * 32 + ((n0 & 63) >> 1)) * 2);
            const float o0 = v[0] * cs.x - v[1] * cs.y, o1 = v[1] * cs.x + v[0] * cs.y;
            const float o2 = v[2] * cs.z - v[3] * cs.w, o3 = v[3] * cs.z + v[2] * cs.w;
            v[0] = o0; v[1] = o1; v[2] = o2; v[3] = o3;
          }
          const float4 o = make_float4(v[0], v[1], v[2], v[3]);
          if (n0 < qd) {
            *reinterpret_cast<float4*>(p.out + (long)m * p.ldo + n0) = o;
          } else if (pos >= 0 && pos < p.cache_len) {
            const int nn = n0 - qd;
            float* base = nn < kd ? p.kc : p.vc;
            const int h = (nn < kd ? nn : nn - kd) >> 6, d = nn & 63;
            const long ce = (((long)slot * p.n_kv_heads + h) * p.cache_len + pos) * 64 + d;
          if (p.kv_bf16) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + ce) = make_uint2(pack_bf16(o.x, o.y), pack_bf16(o.z, o.w));
          else *reinterpret_cast<float4*>(base + ce) = o;
            if (p.v_x3 != nullptr && nn >= kd) {
              const int G = p.n_q_heads / p.n_kv_heads;
              for (int g = 0; g < G; ++g) x3_emit4(p.v_x3, m, (h * G + g) * 64 + d, (p.n_q_heads * 64) >> 5, v[0], v[1], v[2], v[3]);
            }
          }
        }
      }
    }
  }
}

template <int NTW, int EPI, bool W8>
__global__ __launch_bounds__(256) void gemm3_rows_kernel(Gemm3Dev p) {
  constexpr int WTILE = W8 ? 512 : 1024, WLANE = W8 ? 8 : 16;
  __shared__ uint4 xs[12 * 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const int nchunks = p.K >> 5;
  const int mtile0 = blockIdx.y * 4;
  const int tile0 = (blockIdx.x * 4 + wave) * NTW;

  const char* sp[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int idx = tid + 256 * i, blk = idx >> 6, ln = idx & 63;
    const int mt = blk / 3, pc = blk - mt * 3;
    sp[i] = ((mtile0 + mt) * 16 < p.M) ? p.x3 + (size_t)(mtile0 + mt) * nchunks * 3072 + pc * 1024 + ln * 16 : nullptr;
  }
  const char* wb[NTW];
  bool wv[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    wv[t] = (tile0 + t) * 16 < p.N;
    wb[t] = p.w + (size_t)(tile0 + t) * nchunks * WTILE + lane * WLANE;
  }
  f32x4 acc[NTW][4];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 stage[3], wnext[NTW];
  auto fetch = [&](int c) {
#pragma unroll
    for (int i = 0; i < 3; ++i) stage[i] = sp[i] ? *reinterpret_cast<const uint4*>(sp[i] + (size_t)c * 3072) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NTW; ++t) wnext[t] = wv[t] ? load_wfrag<W8>(wb[t] + (size_t)c * WTILE) : make_uint4(0, 0, 0, 0);
  };
  fetch(0);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();  // previous chunk fully consumed
#pragma unroll
    for (int i = 0; i < 3; ++i) xs[tid + 256 * i] = stage[i];
    bf16x8_t a[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) a[t] = wfrag_bf16<W8>(wnext[t]);
    __syncthreads();
    if (c + 1 < nchunks) fetch(c + 1);  // next chunk's loads fly under this chunk's MFMAs
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
        const bf16x8_t b = __builtin_bit_cast(bf16x8_t, xs[(mt * 3 + pc) * 64 + lane]);
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], b, acc[t][mt], 0, 0, 0);
      }
    }
  }

  rows_epilogue<NTW, EPI, W8>(p, acc, mtile0, tile0, r, q);
}

// fp8 x fp8 MFMA variant of the many-row kernel (BASELINE configs[4]: "CDNA4 fp8 MFMA prefill"; SMOLTTS_OPT_FP8_PREFILL): the e4m3
// weight tiles go to the matrix cores as they are (no conversion on the operand path) and the activation operand is the X3
// operand's hi piece rounded to e4m3 while it is staged into LDS (v_cvt_scalef32_pk_fp8_bf16, scale 1: RMSNorm-ed activations are
// O(1)) -- one v_mfma_f32_16x16x32_fp8_fp8 per chunk and tile pair where the exact path runs three bf16 ones, a third of the
// activation bytes.  Products of two e4m3 values are exact in fp32 and the accumulation is fp32; what is lost is the activation's
// precision (3 mantissa bits): the prompt's KV rows are then an approximation, so ids are NOT guaranteed to equal the reference
// greedy decode -- an opt-in mode, never the default.  Same epilogue (it still publishes exact X3 operands of ITS result).
typedef short s16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 bf16x8_to_fp8x8(uint4 v) {
  s16x2_t a = {0, 0}, b = {0, 0};
  a = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(a, __builtin_bit_cast(bf16x2_t, v.x), 1.0f, false);
  a = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(a, __builtin_bit_cast(bf16x2_t, v.y), 1.0f, true);
  b = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(b, __builtin_bit_cast(bf16x2_t, v.z), 1.0f, false);
  b = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(b, __builtin_bit_cast(bf16x2_t, v.w), 1.0f, true);
  return make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
}

template <int NTW, int EPI>
__global__ __launch_bounds__(256) void gemm3_rows_f8_kernel(Gemm3Dev p) {
  __shared__ uint2 xs[4 * 64];  // one 32-k chunk of 4 row tiles as fp8 B fragments
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const int nchunks = p.K >> 5;
  const int mtile0 = blockIdx.y * 4;
  const int tile0 = (blockIdx.x * 4 + wave) * NTW;
  // thread (mt = wave, lane) stages the hi piece of row tile mtile0 + mt
  const char* sp = ((mtile0 + wave) * 16 < p.M) ? p.x3 + (size_t)(mtile0 + wave) * nchunks * 3072 + lane * 16 : nullptr;
  const char* wb[NTW];
  bool wv[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    wv[t] = (tile0 + t) * 16 < p.N;
    wb[t] = p.w + (size_t)(tile0 + t) * nchunks * 512 + lane * 8;
  }
  f32x4 acc[NTW][4];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  uint4 stage;
  uint2 wnext[NTW];
  auto fetch = [&](int c) {
    stage = sp ? *reinterpret_cast<const uint4*>(sp + (size_t)c * 3072) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NTW; ++t) wnext[t] = wv[t] ? *reinterpret_cast<const uint2*>(wb[t] + (size_t)c * 512) : make_uint2(0, 0);
  };
  fetch(0);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();  // previous chunk fully consumed
    xs[tid] = bf16x8_to_fp8x8(stage);
    long a[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) a[t] = __builtin_bit_cast(long, wnext[t]);
    __syncthreads();
    if (c + 1 < nchunks) fetch(c + 1);  // next chunk's loads fly under this chunk's MFMAs
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const long b = __builtin_bit_cast(long, xs[mt * 64 + lane]);
#pragma unroll
      for (int t = 0; t < NTW; ++t) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a[t], b, acc[t][mt], 0, 0, 0);
    }
  }
  rows_epilogue<NTW, EPI, true>(p, acc, mtile0, tile0, r, q);
}

// ---------------------------------------------------------------------------------------------
// Depth-step attention recomputed in front of the output projection (round 4): x += Wo . attention(q, K, V) in ONE launch.
//
// The depth transformer's cache holds <= 8 keys (lm/generate.py:112: one per codebook step), so the attention of a row is
// 12 heads x 8 keys x 64 dims -- small enough for every workgroup of the wo GEMM to work out the rows IT multiplies instead of
// waiting for a launch of its own to publish them (28 launches per frame at 150m: every depth layer of steps 1..7).  No hand-off
// between workgroups, no flag: the price is the q / K / V intake of the workgroup's rows and the redundant arithmetic, so the
// decomposition keeps both small: a workgroup owns only R = 2 rows and, to keep the grid at one workgroup per CU, T = 3
// adjacent 16-column weight tiles (150m, 32 rows: 16 x 16 = 256 workgroups).
//
//   phase A  one wave per (row, pair of kv heads): lane = 32 kq + 8 jj + m  (kq: kv head of the pair, jj = 0..3: key group,
//            m = 0..7) holds dims [4m, 4m+4) and [32+4m, 32+4m+4) of its kv head's keys jj and jj + 4; the G query heads of the
//            kv head share those registers.  Score = 8 FMAs + a 3-step DPP sum over the 8 lanes of a key; softmax per lane over
//            its <= 2 keys, merged over the 4 key groups (row_ror:8, then lane ^ 16); the weighted V sums travel the same two
//            steps, each lane handing over the half it does not publish.  Each lane ends up with 2 output values, splits them
//            into bf16x3 and writes them to LDS in the MFMA B-fragment order of x3.h (R rows instead of 16 per chunk).
//   phase B  the other eight waves, which have had the whole weight stream in flight meanwhile (chunk c belongs to GEMM wave c % nb, nb = gemm3_kernel's wave count for this K: 8 at K = 768), take the B fragments from LDS, run all the MFMAs and leave their partial tiles in LDS; the
//            epilogue of gemm3_kernel's RESID case (fixed-order sum of the four partials, residual, fp32 store, X3 emission,
//            partial sums of squares) runs on waves 0 .. T-1, whose inputs were requested before phase A.
// Every sum is formed in the order attn_short_kernel + gemm3_kernel<1, 1, 3, RESID> form it (the same 4-term FMA chains and
// butterfly over the 16 partial dot products, keys j and j + 4 per lane group, the key groups merged (0 + 1) + (2 + 3), IEEE
// division, chunk c on K part c % nb, partial tiles added in part order): the launch is BIT-IDENTICAL to the two it replaces.
// Reference: Attention.forward (modeling/model/rq_transformer.py:535-570) at decode time inside forward_generate_fast
// (mlx_inference/src/smoltts_mlx/lm/rq_transformer.py:194-220, 281-295).
__device__ __forceinline__ float dpp_ror8(float x) {  // lane i of a 16-lane row <- lane i ^ 8
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xF, 0xF, true));
}
__device__ __forceinline__ float row8_sum(float x) {  // sum over the 8 lanes of a half row, result in all of them
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true));  // row_half_mirror
  return x;
}
// Exchange between lane i and lane i ^ 16 on the VALU (v_permlane16_swap_b32, gfx950): after swap16(a, b) an even 16-lane row holds
// (its own a, the partner's a) and an odd row (the partner's b, its own b) -- so with a == b both lanes of a pair hold (own, partner's)
// in some order, and a + b / max(a, b) is the pair's sum / maximum in every lane; with a != b the even row's b and the odd row's a
// have changed places.  No LDS crossbar round trip (ds_bpermute) on the critical path of a lone wave.
__device__ __forceinline__ void swap16(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
// q . k over the lane's two float4: the same two 4-term FMA chains attn_short_kernel's lanes dl = m and dl = 8 + m run
__device__ __forceinline__ void dot4x2(float4 a, float4 b, float4 k0, float4 k1, float& lo, float& hi) {
  lo = a.x * k0.x; lo = fmaf(a.y, k0.y, lo); lo = fmaf(a.z, k0.z, lo); lo = fmaf(a.w, k0.w, lo);
  hi = b.x * k1.x; hi = fmaf(b.y, k1.y, hi); hi = fmaf(b.z, k1.z, hi); hi = fmaf(b.w, k1.w, hi);
}
// ... and the same tree over the 16 partial sums: row16_sum = (dims 0..31 over lanes 0..7) + (dims 32..63 over lanes 8..15)
__device__ __forceinline__ float score16(float4 a, float4 b, float4 k0, float4 k1) {
  float lo, hi;
  dot4x2(a, b, k0, k1, lo, hi);
  return row8_sum(lo) + row8_sum(hi);
}

#ifndef SMOLTTS_DBG_AWO_ORDER  // 0 (timing experiments): the GEMM waves start their weight stream at once
#define SMOLTTS_DBG_AWO_ORDER 1
#endif
constexpr int AWO_R = 2;  // rows per workgroup

// One (row, pair of kv heads) unit of phase A: what a lane holds between its loads and its arithmetic.
template <int G, bool TWO>
struct AwoUnit {
  float4 ka0, kb0, va0, vb0, ka1, kb1, va1, vb1, qa[G], qb[G];
  int rl, kvc;
  bool hv, ok0, ok1;
};

// PICK (erow is then the row of the engine's q | k | v table picked for this row, SmolttsPickArgs): q and the NEWEST key's K / V
// (key a_pos) come out of that table row instead of the q buffer / the cache, RoPE for position a_pos applied here with the
// picking kernel's own arithmetic (rope_gathered4); `write_kv`: this workgroup also puts the new K / V rows into the cache.
template <int G, bool TWO, bool PICK>
__device__ __forceinline__ void awo_load(AwoUnit<G, TWO>& u, const Gemm3Dev& p, int unit, int kv_pairs, int row0, int lane, long erow = 0,
                                         bool write_kv = false) {
  const int kq = lane >> 5, jj = (lane >> 3) & 3, hb = jj & 1, m8 = lane & 7;
  const int L = p.a_pos + 1;  // keys 0 .. a_pos
  u.rl = unit / kv_pairs;
  const int kvh = (unit - u.rl * kv_pairs) * 2 + kq, row = row0 + u.rl;  // row == slot
  u.hv = kvh < p.n_kv_heads;
  u.kvc = u.hv ? kvh : p.n_kv_heads - 1;
  const long cb = (((long)row * p.n_kv_heads + u.kvc) * p.cache_len) * 64 + 4 * m8;
  u.ok0 = jj < L;
  u.ok1 = TWO && jj + 4 < L;
  const long o0 = cb + (long)(u.ok0 ? jj : 0) * 64, o1 = cb + (long)(u.ok1 ? jj + 4 : 0) * 64;
  const float *k0p = p.kc + o0, *v0p = p.vc + o0, *k1p = p.kc + o1, *v1p = p.vc + o1;
  const int qd = p.n_q_heads * 64, kd = p.n_kv_heads * 64;
  const float* trow = PICK ? p.pk_table + erow * (qd + 2 * kd) : nullptr;
  const bool new0 = PICK && jj == p.a_pos, new1 = PICK && TWO && jj + 4 == p.a_pos;  // this lane's key is the row's own (newest) one
  if (PICK) {
    const float *tk = trow + qd + u.kvc * 64 + 4 * m8, *tv = trow + qd + kd + u.kvc * 64 + 4 * m8;
    if (new0) { k0p = tk; v0p = tv; }
    if (new1) { k1p = tk; v1p = tv; }
  }
  // K: dims [4m, +4) and [32 + 4m, +4); V: the half this lane publishes (hb) first, the other half second
  u.ka0 = *reinterpret_cast<const float4*>(k0p); u.kb0 = *reinterpret_cast<const float4*>(k0p + 32);
  u.va0 = *reinterpret_cast<const float4*>(v0p + 32 * hb); u.vb0 = *reinterpret_cast<const float4*>(v0p + 32 * (1 - hb));
  if (TWO) {
    u.ka1 = *reinterpret_cast<const float4*>(k1p); u.kb1 = *reinterpret_cast<const float4*>(k1p + 32);
    u.va1 = *reinterpret_cast<const float4*>(v1p + 32 * hb); u.vb1 = *reinterpret_cast<const float4*>(v1p + 32 * (1 - hb));
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float* qp = PICK ? trow + (u.kvc * G + g) * 64 + 4 * m8 : p.aq + (long)row * qd + (u.kvc * G + g) * 64 + 4 * m8;
    u.qa[g] = *reinterpret_cast<const float4*>(qp);
    u.qb[g] = *reinterpret_cast<const float4*>(qp + 32);
  }
  if (PICK) {
    // RoPE rows of position a_pos for the lane's dims: (4m, 4m+1), (4m+2, 4m+3) and the same 32 dims on
    const float4 csa = *reinterpret_cast<const float4*>(p.pk_rope + ((long)p.a_pos * 32 + 2 * m8) * 2);
    const float4 csb = *reinterpret_cast<const float4*>(p.pk_rope + ((long)p.a_pos * 32 + 16 + 2 * m8) * 2);
#pragma unroll
    for (int g = 0; g < G; ++g) { u.qa[g] = rope_gathered4(u.qa[g], csa); u.qb[g] = rope_gathered4(u.qb[g], csb); }
    const float4 ra0 = rope_gathered4(u.ka0, csa), rb0 = rope_gathered4(u.kb0, csb);
    if (new0) { u.ka0 = ra0; u.kb0 = rb0; }
    if (TWO) {
      const float4 ra1 = rope_gathered4(u.ka1, csa), rb1 = rope_gathered4(u.kb1, csb);
      if (new1) { u.ka1 = ra1; u.kb1 = rb1; }
    }
    if (write_kv && u.hv && (new0 || new1)) {  // the cache rows later steps attend to (what the picking kernel's gather writes)
      float* kw = p.kc + cb + (long)p.a_pos * 64;
      float* vw = p.vc + cb + (long)p.a_pos * 64;
      const float4 ka = new0 ? u.ka0 : u.ka1, kb = new0 ? u.kb0 : u.kb1, va = new0 ? u.va0 : u.va1, vb = new0 ? u.vb0 : u.vb1;
      *reinterpret_cast<float4*>(kw) = ka; *reinterpret_cast<float4*>(kw + 32) = kb;
      *reinterpret_cast<float4*>(vw + 32 * hb) = va; *reinterpret_cast<float4*>(vw + 32 * (1 - hb)) = vb;
    }
  }
}

template <int G, bool TWO>
__device__ __forceinline__ void awo_compute(const AwoUnit<G, TWO>& u, char* frag, int lane) {
  // No contraction of this function's products into later additions (written as plain `*` HERE: `__fmul_rn` is a header function
  // whose product carries the header's contraction flag, and the compiler did re-use the unrounded product inside the bf16x3
  // split's subtractions -- the pieces then summed to a neighbour of the fp32 value attn_short_kernel rounds to, one ulp off in
  // ~10 % of the outputs).  Explicit fmaf calls stay fused multiply-adds.
#pragma clang fp contract(off)
  constexpr int R = AWO_R;
  const int hb = (lane >> 3) & 1, m8 = lane & 7;
  const bool up = (lane & 16) != 0;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float4 a = make_float4(u.qa[g].x * 0.125f, u.qa[g].y * 0.125f, u.qa[g].z * 0.125f, u.qa[g].w * 0.125f);
    const float4 b = make_float4(u.qb[g].x * 0.125f, u.qb[g].y * 0.125f, u.qb[g].z * 0.125f, u.qb[g].w * 0.125f);
    float s0 = score16(a, b, u.ka0, u.kb0), s1 = -INFINITY;
    s0 = u.ok0 ? s0 : -INFINITY;
    if (TWO) {
      s1 = score16(a, b, u.ka1, u.kb1);
      s1 = u.ok1 ? s1 : -INFINITY;
    }
    float mx = TWO ? fmaxf(s0, s1) : s0;
    mx = fmaxf(mx, dpp_ror8(mx));
    {
      float t = mx;
      swap16(mx, t);
      mx = fmaxf(mx, t);  // key 0 is always there: finite
    }
    const float e0 = __expf(s0 - mx);    // exp(-inf) = 0: keys behind the row's position drop out by themselves
    float den = e0;
    // (rounded products, as attn_short_kernel's fmaf(e, v, 0): this scope does not contract them into the adds below)
    float4 oa = make_float4(e0 * u.va0.x, e0 * u.va0.y, e0 * u.va0.z, e0 * u.va0.w);
    float4 ob = make_float4(e0 * u.vb0.x, e0 * u.vb0.y, e0 * u.vb0.z, e0 * u.vb0.w);
    if (TWO) {
      const float e1 = __expf(s1 - mx);
      den += e1;
      oa.x = fmaf(e1, u.va1.x, oa.x); oa.y = fmaf(e1, u.va1.y, oa.y); oa.z = fmaf(e1, u.va1.z, oa.z); oa.w = fmaf(e1, u.va1.w, oa.w);
      ob.x = fmaf(e1, u.vb1.x, ob.x); ob.y = fmaf(e1, u.vb1.y, ob.y); ob.z = fmaf(e1, u.vb1.z, ob.z); ob.w = fmaf(e1, u.vb1.w, ob.w);
    }
    den += dpp_ror8(den);
    {
      float t = den;
      swap16(den, t);
      den += t;
    }
    // step 1 (lane ^ 8: the other half's publisher): hand over the sums of the half this lane does not publish
    oa.x += dpp_ror8(ob.x); oa.y += dpp_ror8(ob.y); oa.z += dpp_ror8(ob.z); oa.w += dpp_ror8(ob.w);
    // step 2 (lane ^ 16: same four values over the other two key groups): the lower lane publishes (x, y), the upper (z, w); one swap
    // moves the lower lane's z to the upper and the upper lane's x to the lower, after which x + z is what the lane publishes
    swap16(oa.x, oa.z);
    swap16(oa.y, oa.w);
    const float inv = 1.0f / den;
    const float ox = (oa.x + oa.z) * inv, oy = (oa.y + oa.w) * inv;  // (plain products: this scope does not contract; __fmul_rn's would)
    uint32_t h, mm, l;
    split3_pair(ox, oy, h, mm, l);
    const int c = 2 * (u.kvc * G + g) + hb;  // chunk: head, half of its 64 dims
    char* fp = frag + ((size_t)((c * 3) * 4 + (m8 >> 1)) * R + u.rl) * 16 + (m8 & 1) * 8 + (up ? 4 : 0);
    if (u.hv) {
      *reinterpret_cast<uint32_t*>(fp) = h;
      *reinterpret_cast<uint32_t*>(fp + 4 * R * 16) = mm;
      *reinterpret_cast<uint32_t*>(fp + 8 * R * 16) = l;
    }
  }
}

constexpr int AWO_NA = 4, AWO_NB = 8, AWO_U = 3;  // attention waves, GEMM waves, 32-k chunks per GEMM wave (K <= 768)

template <int G, int T, bool TWO, bool W8, bool PICK, int NBF>  // TWO: more than 4 keys (a second key per lane); PICK: SmolttsPickArgs;
// NBF != 0 ("FULL"): NBF = 8 or 6 K parts of exactly AWO_U chunks and whole groups of T column tiles -- no predicates on the weight stream (see gemm3_kernel)
__global__ __launch_bounds__(768) void attn_wo_kernel(const char* w_, const float* kc_, const float* vc_, const float* q_or_cand, const float* pk_table_,
                                                      int m_tiles, int N_, int K_, int heads_pos, Gemm3Dev p) {
  // 14 dwords of leading scalar arguments = what the first loads of both kinds of waves need, in SGPRs at wave launch (see gemm3_kernel)
  KernargTouch<AWO_KERNARG_LINES> kt;
  kernarg_touch(kt);
  p.w = w_; p.kc = const_cast<float*>(kc_); p.vc = const_cast<float*>(vc_); p.pk_table = pk_table_;
  if (PICK) p.pk_cand = q_or_cand; else p.aq = q_or_cand;
  p.M = m_tiles & 0xffff; p.pk_tiles = m_tiles >> 16; p.N = N_; p.K = K_;
  p.n_q_heads = heads_pos & 255; p.n_kv_heads = (heads_pos >> 8) & 255; p.cache_len = (heads_pos >> 16) & 255; p.a_pos = heads_pos >> 24;
  constexpr int WTILE = W8 ? 512 : 1024, WLANE = W8 ? 8 : 16;
  constexpr int R = AWO_R, NA = AWO_NA, NB = AWO_NB, U = AWO_U;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const int ng = blockIdx.x, row0 = blockIdx.y * R;
  const int nchunks = p.K >> 5;
  // K parts: as many as gemm3_kernel has waves for this K (launch3_fmt), so that the partial sums are the same numbers
  constexpr bool FULL = NBF != 0;
  const int nb = FULL ? NBF : ((nchunks + 2) / 3 < 1 ? 1 : ((nchunks + 2) / 3 > NB ? NB : (nchunks + 2) / 3));
  char* frag = reinterpret_cast<char*>(smem);  // [chunk][piece][q][row] x 16 B
  float4* red4 = reinterpret_cast<float4*>(frag + (size_t)nchunks * 3 * 4 * R * 16);  // [GEMM wave][tile][lane]
  STAMP3(0);

  // Two kinds of waves, because a CU takes its operands in at ~64 B per clock whatever the waves do: the
  // attention waves must not queue their few q / K / V loads behind the weight stream, and the weight stream must not wait for
  // the attention arithmetic.  Waves 0 .. NA-1: phase A, then (waves 0 .. T-1) the epilogue.  Waves NA .. NA+NB-1: the whole
  // weight tile stream (chunk c belongs to GEMM wave c % NB) and, behind the barrier, all the MFMAs.
  if (wave < NA) {
    const int kv_pairs = (p.n_kv_heads + 1) >> 1;
    const int n_units = (p.M - row0 < R ? p.M - row0 : R) * kv_pairs;  // (row, kv pair) units of this workgroup: wave, wave + NA, ..
    // PICK: the previous step's code of both rows, by every attention wave (the candidates are 2 KB per row, and the finishing
    // waves need both rows' embedding rows): id -> row of the embedding / q | k | v tables
    long erow0 = 0, erow1 = 0;
    Top2 t0{0.f, 0, 0.f}, t1 = t0;
#if SMOLTTS_DBG_AWO_ORDER
    if (PICK) __syncthreads();  // (the GEMM waves' go: with a pick in front, this wave's own loads come too late to wait for)
#endif
    if (PICK) {
      const int n_cols = p.pk_tiles * 16;
      t0 = cand_pick_wave(p.pk_cand + (size_t)row0 * p.pk_tiles * 4, p.pk_tiles, lane);
      t1 = t0;
      if (row0 + 1 < p.M) t1 = cand_pick_wave(p.pk_cand + (size_t)(row0 + 1) * p.pk_tiles * 4, p.pk_tiles, lane);
      if (t0.i1 < 0 || t0.i1 >= n_cols) t0.i1 = 0;  // all-NaN row: stay inside the tables (argmax_row does the same)
      if (t1.i1 < 0 || t1.i1 >= n_cols) t1.i1 = 0;
      erow0 = (long)t0.i1 + p.pk_off;
      erow1 = (long)t1.i1 + p.pk_off;
    }
    AwoUnit<G, TWO> un;
    if (wave < n_units) awo_load<G, TWO, PICK>(un, p, wave, kv_pairs, row0, lane, wave / kv_pairs ? erow1 : erow0, blockIdx.x == 0);
    kernarg_touched(kt);
    // epilogue inputs of the finishing waves (wave f: column tile f): in flight across phase A, the barriers and the K loop
    const bool fin = wave < T;
    const int tf = fin ? wave : 0;
    const int m = row0 + r;
    const bool row_on = r < R;
    const bool mvalid = fin && row_on && m < p.M;
    float4 rr, bb, ga, gb, ws;
    rr = bb = make_float4(0.f, 0.f, 0.f, 0.f);
    ga = gb = ws = make_float4(1.f, 1.f, 1.f, 1.f);
    if (fin) {
      const int mc = m < p.M ? m : p.M - 1;
      const int n0r = (ng * T + tf) * 16 + q * 4;
      const int n0 = n0r < p.N ? n0r : p.N - 4;  // N % 4 == 0
      if (W8) ws = *reinterpret_cast<const float4*>(p.wscale + n0);
      if (PICK) {  // the residual row is the picked code's embedding row (bf16, widened exactly: what the picking kernel stores as x)
        const uint2 e = *reinterpret_cast<const uint2*>(p.pk_emb + ((r & 1) ? erow1 : erow0) * p.K + n0);
        rr = make_float4(bf16_lo(e.x), bf16_hi(e.x), bf16_lo(e.y), bf16_hi(e.y));
      } else {
        rr = *reinterpret_cast<const float4*>(p.resid + (long)mc * p.ldo + n0);
      }
      if (p.bias != nullptr) bb = *reinterpret_cast<const float4*>(p.bias + n0);
      if (p.emit.x3a && p.emit.gamma_a) ga = *reinterpret_cast<const float4*>(p.emit.gamma_a + n0);
      if (p.emit.x3b && p.emit.gamma_b) gb = *reinterpret_cast<const float4*>(p.emit.gamma_b + n0);
    }
    STAMP3(5);
#if SMOLTTS_DBG_AWO_ORDER
    if (!PICK) __syncthreads();  // (lets the GEMM waves start their weight stream: see there)
#endif
    if (wave < n_units) awo_compute(un, frag, lane);
    for (int unit = wave + NA; unit < n_units; unit += NA) {  // (more than 2 x NA / R kv heads: no shipped config)
      awo_load<G, TWO, PICK>(un, p, unit, kv_pairs, row0, lane, unit / kv_pairs ? erow1 : erow0, blockIdx.x == 0);
      awo_compute(un, frag, lane);
    }
    STAMP3(6);
    __syncthreads();  // the fragments are in LDS
    __syncthreads();  // the GEMM waves' partial tiles are in LDS
    STAMP3(3);
    if (PICK && blockIdx.x == 0 && wave == NA - 1 && lane == 0) {
      // the records of the picking kernel (argmax_row), once per row, by a wave that has nothing left to do (NA - 1 >= T: not a
      // finishing wave): its dependent loads (mask, smallest gap so far, frame number) are on nobody's critical path here
#pragma unroll
      for (int rl = 0; rl < R; ++rl) {
        const int row = row0 + rl;
        if (row >= p.M) break;
        const Top2 t = rl ? t1 : t0;
        p.pk_ids[(long)row * p.pk_ids_stride] = t.i1;
        if (p.pk_margin && (p.pk_mask == nullptr || p.pk_mask[row])) {
          const float gap = t.v1 - t.v2;
          if (gap < p.pk_margin[row]) {
            p.pk_margin[row] = gap;
            if (p.pk_margin_at) p.pk_margin_at[row] = p.pk_frames[row] * 64 + p.pk_step;
          }
        }
      }
    }
    if (!fin) return;
    float4 part[NB];  // (gemm3_kernel's reduction, term for term)
#pragma unroll
    for (int w = 0; w < NB; ++w) part[w] = red4[((w < nb ? w : 0) * T + tf) * 64 + lane];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NB; ++w) {
      const bool on = w < nb;  // (nb is a constant in the predicate-free form)
      v[0] += on ? part[w].x : 0.f;
      v[1] += on ? part[w].y : 0.f;
      v[2] += on ? part[w].z : 0.f;
      v[3] += on ? part[w].w : 0.f;
    }
    const int ntile = ng * T + tf;
    const int n0 = ntile * 16 + q * 4;
    const bool valid = mvalid && n0 < p.N;
    if (W8) { v[0] *= ws.x; v[1] *= ws.y; v[2] *= ws.z; v[3] *= ws.w; }
    v[0] = v[0] + bb.x + rr.x; v[1] = v[1] + bb.y + rr.y; v[2] = v[2] + bb.z + rr.z; v[3] = v[3] + bb.w + rr.w;
    if (valid) {
      *reinterpret_cast<float4*>(p.out + (long)m * p.ldo + n0) = make_float4(v[0], v[1], v[2], v[3]);
      if (p.emit.x3a) x3_emit4(p.emit.x3a, m, n0, p.N >> 5, v[0] * ga.x, v[1] * ga.y, v[2] * ga.z, v[3] * ga.w);
      if (p.emit.x3b) x3_emit4(p.emit.x3b, m, n0, p.N >> 5, v[0] * gb.x, v[1] * gb.y, v[2] * gb.z, v[3] * gb.w);
    }
    if (p.emit.ssq) {  // all 64 lanes take part in the shuffles
      float sq = valid ? ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) : 0.f;
      sq = sum_xor16_32(sq);
      if (q == 0 && mvalid && ntile * 16 < p.N) p.emit.ssq[(size_t)m * (p.N >> 4) + ntile] = sq;
    }
    STAMP3(4);
    return;
  }

  // ---- GEMM waves.  They hold their loads back until the attention waves have issued theirs: the CU's vector memory pipeline
  // serves requests in arrival order, and phase A -- the critical path -- needs its 26 KB before the K loop needs its 72 KB.
  const int gw = wave - NA;
#if SMOLTTS_DBG_AWO_ORDER
  __syncthreads();
#endif
  uint4 wf[U][T];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int c = gw + u * nb;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int ntile = ng * T + t;
      wf[u][t] = (FULL ? (NBF == NB || gw < nb) : (gw < nb && c < nchunks && ntile * 16 < p.N)) ? load_wfrag<W8, false>(p.w + ((size_t)ntile * nchunks + c) * WTILE + lane * WLANE)
                                                             : make_uint4(0, 0, 0, 0);
    }
  }
  kernarg_touched(kt);
  STAMP3(5);
  __syncthreads();  // the fragments are in LDS
  f32x4 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int c = gw + u * nb;
    if (FULL ? (NBF == NB || gw < nb) : (gw < nb && c < nchunks)) {  // wave-uniform
      const char* fp = frag + ((size_t)((c * 3) * 4 + q) * R + (r & (R - 1))) * 16;
      uint4 xb[3];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) xb[pc] = *reinterpret_cast<const uint4*>(fp + pc * 4 * R * 16);  // lanes r >= R: row r mod R again (columns of the product nobody reads)
      bf16x8_t a[T];
#pragma unroll
      for (int t = 0; t < T; ++t) a[t] = wfrag_bf16<W8>(wf[u][t]);
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)  // tiles innermost: consecutive MFMAs go to different accumulators
#pragma unroll
        for (int t = 0; t < T; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], __builtin_bit_cast(bf16x8_t, xb[pc]), acc[t], 0, 0, 0);
    }
  }
  STAMP3(1);
#pragma unroll
  for (int t = 0; t < T; ++t) red4[(gw * T + t) * 64 + lane] = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
  STAMP3(2);
  __syncthreads();  // the partial tiles are in LDS
}

template <int G, bool W8>
static int launch_attn_wo_g(const Gemm3Dev& d, hipStream_t stream) {
  const int nchunks = d.K / 32, ntiles = (d.N + 15) / 16;
  // few rows: one column tile per workgroup so that the weight stream is spread over the chip; many: three (<= one workgroup per
  // CU at 32 rows x 48 tiles)
  const int T = d.M > 16 ? 3 : 1;
  const dim3 grid((ntiles + T - 1) / T, (d.M + AWO_R - 1) / AWO_R);
  ST_REQUIRE(grid.y <= 65535, SMOLTTS_E_INVALID, "gemm3: M=%d too large for one launch", d.M);
  const size_t lds = (size_t)nchunks * 3 * 4 * AWO_R * 16 + (size_t)AWO_NB * T * 1024;
  const bool two = d.a_pos + 1 > 4;
  const dim3 block((AWO_NA + AWO_NB) * 64);
  ST_REQUIRE(d.M < 65536 && d.pk_tiles < 32768 && d.n_q_heads < 256 && d.cache_len < 256 && d.a_pos < 128, SMOLTTS_E_INVALID,
             "gemm3: attention + wo: sizes out of the packed arguments' range");
  const int nb_full = ((nchunks == 8 * AWO_U || nchunks == 6 * AWO_U) && ntiles % T == 0) ? nchunks / AWO_U : 0;
  const int m_tiles = d.M | (d.pk_tiles << 16), heads_pos = d.n_q_heads | (d.n_kv_heads << 8) | (d.cache_len << 16) | (d.a_pos << 24);
#define ST_AWO(TT, TWO_, PK_) \
  do { if (nb_full == 8) ST_AWO_(TT, TWO_, PK_, 8); else if (nb_full == 6) ST_AWO_(TT, TWO_, PK_, 6); else ST_AWO_(TT, TWO_, PK_, 0); } while (0)
#define ST_AWO_(TT, TWO_, PK_, FULL_)                                                                                                   \
  hipLaunchKernelGGL((attn_wo_kernel<G, TT, TWO_, W8, PK_, FULL_>), grid, block, lds, stream, d.w, (const float*)d.kc, (const float*)d.vc, \
                     PK_ ? d.pk_cand : d.aq, d.pk_table, m_tiles, d.N, d.K, heads_pos, d)
  if (d.pk_cand) {
    if (T == 3) { if (two) ST_AWO(3, true, true); else ST_AWO(3, false, true); }
    else { if (two) ST_AWO(1, true, true); else ST_AWO(1, false, true); }
  } else {
    if (T == 3) { if (two) ST_AWO(3, true, false); else ST_AWO(3, false, false); }
    else { if (two) ST_AWO(1, true, false); else ST_AWO(1, false, false); }
  }
#undef ST_AWO
#undef ST_AWO_
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

template <bool W8>
static int launch_attn_wo(const Gemm3Dev& d, hipStream_t stream) {
  switch (d.n_q_heads / d.n_kv_heads) {
    case 1: return launch_attn_wo_g<1, W8>(d, stream);
    case 2: return launch_attn_wo_g<2, W8>(d, stream);
    case 3: return launch_attn_wo_g<3, W8>(d, stream);
    default: return launch_attn_wo_g<4, W8>(d, stream);
  }
}

template <int EPI, bool W8>
static int launch3_rows(const Gemm3Dev& d, hipStream_t stream) {
  const int ntiles = (d.N + 15) / 16;
  int NTW = ntiles >= 16 ? 4 : (ntiles >= 8 ? 2 : 1);
  // narrow outputs over few row blocks (wo / w2 of a 3.5k-row prompt batch: 3 x 56 workgroups) leave CUs idle: halve the
  // column tile until the grid covers the chip
  const int row_blocks = (d.M + 63) / 64;
  while (NTW > 1 && ((ntiles + 4 * NTW - 1) / (4 * NTW)) * row_blocks < 256) NTW >>= 1;
  const dim3 grid((ntiles + 4 * NTW - 1) / (4 * NTW), row_blocks);
  ST_REQUIRE(grid.y <= 65535, SMOLTTS_E_INVALID, "gemm3: M=%d too large for one launch", d.M);
  if (W8 && d.f8_act) {  // fp8 x fp8 MFMA (opt-in: SmolttsGemm3Args.fp8_activations)
    if (NTW == 4) hipLaunchKernelGGL((gemm3_rows_f8_kernel<4, EPI>), grid, dim3(256), 0, stream, d);
    else if (NTW == 2) hipLaunchKernelGGL((gemm3_rows_f8_kernel<2, EPI>), grid, dim3(256), 0, stream, d);
    else hipLaunchKernelGGL((gemm3_rows_f8_kernel<1, EPI>), grid, dim3(256), 0, stream, d);
  } else if (NTW == 4) hipLaunchKernelGGL((gemm3_rows_kernel<4, EPI, W8>), grid, dim3(256), 0, stream, d);
  else if (NTW == 2) hipLaunchKernelGGL((gemm3_rows_kernel<2, EPI, W8>), grid, dim3(256), 0, stream, d);
  else hipLaunchKernelGGL((gemm3_rows_kernel<1, EPI, W8>), grid, dim3(256), 0, stream, d);
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

template <int MT, int T, int U, int EPI, bool W8>
static int launch3_one(const Gemm3Dev& d, int nwaves, hipStream_t stream) {
  const int ntiles = (d.N + 15) / 16;
  if (nwaves < MT * T) nwaves = MT * T;  // one finishing wave per (column tile, row tile) of the workgroup
  const dim3 grid((ntiles + T - 1) / T, ((d.M + 16 * MT - 1) / (16 * MT)) * (d.half_rows ? 2 : 1));
  const size_t lds = (size_t)nwaves * T * MT * 1024 + (size_t)MT * 16 * sizeof(float);  // partial tiles + the row scales
  // the predicate-free form: 8 or 6 waves (K = 768 / 3072 / 1536; K = 576), exactly U chunks each, whole groups of T tiles
  const int nw_full = (MT == 1 && (nwaves == 8 || nwaves == 6) && d.K / 32 == nwaves * U && ntiles % T == 0) ? nwaves : 0;  // (MT == 1: every launched row tile starts below M)
#define ST_G3(NT_, NW_)                                                                                                          \
  hipLaunchKernelGGL((gemm3_kernel<MT, T, U, EPI, W8, NT_, NW_>), grid, dim3(nwaves * 64), lds, stream, d.w, d.x3, d.M, d.N, d.K, \
                     d.half_rows, nwaves, d)
#define ST_G3N(NT_) do { if (nw_full == 8) ST_G3(NT_, (MT == 1 ? 8 : 0)); else if (nw_full == 6) ST_G3(NT_, (MT == 1 ? 6 : 0)); else ST_G3(NT_, 0); } while (0)
  if (MT == 1 && d.w_nt) ST_G3N(MT == 1);
  else ST_G3N(false);
#undef ST_G3N
#undef ST_G3
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

// Decomposition: enough workgroups to spread the weight stream over the chip (16-row tiles go to
// separate workgroups when there are few column tiles), all of a wave's loads in flight at once.
template <int EPI, bool W8>
static int launch3_fmt(const Gemm3Dev& d, hipStream_t stream) {
  const int nchunks = d.K / 32, ntiles = (d.N + 15) / 16;
  int nwaves = (nchunks + 2) / 3;
  nwaves = nwaves < 1 ? 1 : (nwaves > 8 ? 8 : nwaves);  // <= 512 threads: 256 VGPRs per lane
  const int cpw = (nchunks + nwaves - 1) / nwaves;  // chunks per wave
  if (d.M >= 256) return launch3_rows<EPI, W8>(d, stream);  // prefill: LDS-shared activation chunks, no split-K
  if (d.M > 128)  // 129..255 rows: 64 rows per workgroup, weights re-used from registers
    return launch3_one<4, 1, 2, EPI, W8>(d, nwaves < 4 ? 4 : nwaves, stream);
  // 16-row tiles go to separate workgroups (MT = 1); T = the smallest number of column tiles per
  // workgroup that keeps the grid within one workgroup per CU and 32 rows (measured at 64 and 128 rows:
  // interleaved 32-row chains run 1.3-1.45x faster than one chain of 64-row workgroups), so every CU
  // takes the activation operand in once and the per-CU byte load (the bound of these kernels) stays even.
  const int row_tiles = (d.M + 15) / 16;
  const int wg_limit = row_tiles > 2 ? 128 * row_tiles : 256;
  int T = 1;
  while (T < 4 && ((ntiles + T - 1) / T) * row_tiles > wg_limit) ++T;
  if (T == 4) return launch3_one<1, 4, 3, EPI, W8>(d, nwaves, stream);
  if (T == 3) return launch3_one<1, 3, 3, EPI, W8>(d, nwaves, stream);
  if (T == 2) return launch3_one<1, 2, 3, EPI, W8>(d, nwaves, stream);
  // few column tiles: split the 16-row tiles in two so that ~2x the CUs share the activation bytes
  Gemm3Dev dd = d;
  dd.half_rows = (d.M > 8 && ntiles * row_tiles * 2 <= wg_limit) ? 1 : 0;
  if (cpw > 6) return launch3_one<1, 1, 12, EPI, W8>(dd, nwaves, stream);
  if (cpw > 3) return launch3_one<1, 1, 6, EPI, W8>(dd, nwaves, stream);
  return launch3_one<1, 1, 3, EPI, W8>(dd, nwaves, stream);
}

template <int EPI>
static int launch3_epi(const Gemm3Dev& d, hipStream_t stream) {
  return d.wscale ? launch3_fmt<EPI, true>(d, stream) : launch3_fmt<EPI, false>(d, stream);
}

#ifdef SMOLTTS_DEBUG_HOOKS
unsigned long long* debug_stamp_buffer();  // gemm.hip
int profile_hook_begin(int prologue, int epilogue, int N, hipStream_t stream);  // gemm.hip
void profile_hook_end(int i, hipStream_t stream);
#endif
static int launch_gemm3_impl(const SmolttsGemm3Args& a, hipStream_t stream);

int launch_gemm3(const SmolttsGemm3Args& a, hipStream_t stream) {
#ifdef SMOLTTS_DEBUG_HOOKS
  // the event hook sees a normed input as the RMSNorm prologue
  const int i = profile_hook_begin(a.ssq_in_dev ? SMOLTTS_PRO_RMSNORM : SMOLTTS_PRO_NONE, a.epilogue, a.N, stream);
  const int rc = launch_gemm3_impl(a, stream);
  profile_hook_end(i, stream);
  return rc;
#else
  return launch_gemm3_impl(a, stream);
#endif
}

static int launch_gemm3_impl(const SmolttsGemm3Args& a, hipStream_t stream) {
  ST_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.K % 32 == 0 && a.N % 4 == 0, SMOLTTS_E_INVALID,
             "gemm3: bad shape M=%d N=%d K=%d (K %% 32, N %% 4)", a.M, a.N, a.K);
  ST_REQUIRE(a.w_dev && (a.x3_dev || ((a.attn_q_dev || a.pick) && a.epilogue == SMOLTTS_EPI_RESID)), SMOLTTS_E_INVALID, "gemm3: null operand");
  ST_REQUIRE(a.ssq_in_dev == nullptr || a.K % 64 == 0, SMOLTTS_E_INVALID, "gemm3: normed input needs K %% 64 == 0");
  Gemm3Dev d;
  memset(&d, 0, sizeof(d));
  ST_REQUIRE((a.w_format == SMOLTTS_W_BF16 && !a.w_scale_dev) || (a.w_format == SMOLTTS_W_FP8 && a.w_scale_dev && a.N % 16 == 0),
             SMOLTTS_E_INVALID, "gemm3: w_format %d / w_scale_dev inconsistent (fp8 needs scales and N %% 16 == 0)", a.w_format);
  d.w = (const char*)a.w_dev; d.wscale = a.w_scale_dev; d.x3 = (const char*)a.x3_dev; d.M = a.M; d.N = a.N; d.K = a.K;
  d.ssq_in = a.ssq_in_dev; d.eps = a.eps; d.bias = a.bias_dev; d.resid = a.resid_dev; d.out = a.out_dev; d.ldo = a.ldo;
  d.x3_out = (char*)a.x3_out_dev;
  d.emit.x3a = (char*)a.emit_a_dev; d.emit.gamma_a = a.gamma_a_dev; d.emit.x3b = (char*)a.emit_b_dev;
  d.emit.gamma_b = a.gamma_b_dev; d.emit.ssq = a.ssq_out_dev;
  d.rope = a.rope_dev; d.row_pos = a.row_pos_dev; d.row_slot = a.row_slot_dev; d.kc = a.k_cache_dev; d.vc = a.v_cache_dev;
  d.n_q_heads = a.n_q_heads; d.n_kv_heads = a.n_kv_heads; d.cache_len = a.cache_len;
  d.kv_bf16 = a.kv_format == SMOLTTS_KV_BF16;
  d.w_nt = a.w_stream != 0;
  d.f8_act = a.fp8_activations != 0 && a.w_format == SMOLTTS_W_FP8;
  d.v_x3 = a.epilogue == SMOLTTS_EPI_QKV_ROPE ? (char*)a.v_x3_dev : nullptr;
  d.cand = a.epilogue == SMOLTTS_EPI_STORE ? a.cand_out_dev : nullptr;
#ifdef SMOLTTS_DEBUG_HOOKS
  d.stamps = debug_stamp_buffer();
#endif
  switch (a.epilogue) {
    case SMOLTTS_EPI_STORE:
      ST_REQUIRE(a.out_dev && a.ldo % 4 == 0, SMOLTTS_E_INVALID, "gemm3: STORE needs out/ldo");
      ST_REQUIRE(a.cand_out_dev == nullptr || a.M < 256, SMOLTTS_E_INVALID, "gemm3: cand_out_dev needs M < 256 (the many-row kernel leaves no candidates)");
      ST_REQUIRE(!(a.emit_a_dev || a.emit_b_dev || a.ssq_out_dev) || a.N % 64 == 0, SMOLTTS_E_INVALID, "gemm3: emission needs N %% 64 == 0");
      return launch3_epi<SMOLTTS_EPI_STORE>(d, stream);
    case SMOLTTS_EPI_RESID:
      ST_REQUIRE(a.out_dev && (a.resid_dev || a.pick) && a.ldo % 4 == 0, SMOLTTS_E_INVALID, "gemm3: RESID needs out/resid/ldo");
      ST_REQUIRE(!(a.emit_a_dev || a.emit_b_dev || a.ssq_out_dev) || a.N % 64 == 0, SMOLTTS_E_INVALID, "gemm3: emission needs N %% 64 == 0");
      if (a.pick) {  // ... with the previous step's code picked in front of it
        const SmolttsPickArgs& k = *a.pick;
        ST_REQUIRE(k.cand_dev && k.cand_tiles > 0 && k.qkv_table_dev && k.rope_dev && k.emb_dev && k.ids_dev && a.attn_pos >= 1 &&
                       (!k.margin_at_dev || k.frames_dev) && a.M > 0,
                   SMOLTTS_E_INVALID, "gemm3: pick arguments incomplete (or attn_pos < 1: step 0 has no pick in front of it)");
        d.pk_cand = k.cand_dev; d.pk_tiles = k.cand_tiles; d.pk_table = k.qkv_table_dev; d.pk_rope = k.rope_dev;
        d.pk_emb = (const uint16_t*)k.emb_dev; d.pk_off = k.emb_row_offset; d.pk_ids = k.ids_dev; d.pk_ids_stride = k.ids_stride;
        d.pk_margin = k.margin_dev; d.pk_mask = k.margin_mask_dev; d.pk_margin_at = k.margin_at_dev; d.pk_frames = k.frames_dev; d.pk_step = k.step;
      }
      if (a.attn_q_dev || a.pick) {  // the activation operand is the depth-step attention of these rows, worked out inside the launch
        ST_REQUIRE(smoltts_gemm3_attn_fusable(a.n_q_heads, a.n_kv_heads, a.cache_len) && a.K == a.n_q_heads * 64 && a.k_cache_dev && a.v_cache_dev &&
                       a.kv_format == SMOLTTS_KV_F32 && a.attn_pos >= 0 && a.attn_pos < a.cache_len && !a.ssq_in_dev,
                   SMOLTTS_E_INVALID, "gemm3: attention prologue needs <= 12 query heads in groups of <= 4, <= 8 cache entries (fp32), "
                   "K = heads * 64 and 0 <= attn_pos < cache_len (heads %d/%d, cache_len %d, K %d, attn_pos %d)", a.n_q_heads, a.n_kv_heads,
                   a.cache_len, a.K, a.attn_pos);
        d.aq = a.attn_q_dev; d.a_pos = a.attn_pos;
        return d.wscale ? launch_attn_wo<true>(d, stream) : launch_attn_wo<false>(d, stream);
      }
      return launch3_epi<SMOLTTS_EPI_RESID>(d, stream);
    case SMOLTTS_EPI_SWIGLU:
      ST_REQUIRE(a.x3_out_dev && a.N % 64 == 0, SMOLTTS_E_INVALID, "gemm3: SWIGLU needs x3_out and N %% 64 == 0");
      return launch3_epi<SMOLTTS_EPI_SWIGLU>(d, stream);
    case SMOLTTS_EPI_QKV_ROPE:
      ST_REQUIRE(a.rope_dev && a.row_pos_dev && a.row_slot_dev && a.k_cache_dev && a.v_cache_dev && a.out_dev &&
                     a.N == (a.n_q_heads + 2 * a.n_kv_heads) * 64 && a.cache_len > 0 && a.ldo % 4 == 0 &&
                     (a.kv_format == SMOLTTS_KV_F32 || a.kv_format == SMOLTTS_KV_BF16),
                 SMOLTTS_E_INVALID, "gemm3: QKV_ROPE arguments inconsistent");
      return launch3_epi<SMOLTTS_EPI_QKV_ROPE>(d, stream);
    default:
      set_error("gemm3: unsupported epilogue %d", a.epilogue);
      return SMOLTTS_E_INVALID;
  }
}

// ---- fp32 rows -> X3 (+ ssq): test / bring-up entry for operands that no fused producer writes
__global__ __launch_bounds__(256) void x3_pack_kernel(const float* x, long ldx, int dim, EmitDev e) {
  __shared__ float sh4[4];
  const int m = blockIdx.x;
  float ss = 0.f;
  for (int k = threadIdx.x * 4; k < dim; k += 256 * 4) {
    const float4 v = *reinterpret_cast<const float4*>(x + (long)m * ldx + k);
    ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    emit_x4(e, m, k, dim >> 5, v.x, v.y, v.z, v.w);
  }
  emit_row_ssq(e, m, dim, ss, sh4);
}

int launch_x3_pack(const float* x, int64_t ldx, int n_rows, int dim, void* x3a, const float* gamma_a, void* x3b,
                   const float* gamma_b, float* ssq, hipStream_t stream) {
  ST_REQUIRE(x && n_rows > 0 && dim % 32 == 0 && ldx % 4 == 0, SMOLTTS_E_INVALID, "x3_pack: bad arguments");
  EmitDev e{(char*)x3a, gamma_a, (char*)x3b, gamma_b, ssq};
  hipLaunchKernelGGL(x3_pack_kernel, dim3(n_rows), dim3(256), 0, stream, x, (long)ldx, dim, e);
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

}  // namespace smoltts

extern "C" {
int smoltts_gemm3_attn_fusable(int32_t n_q_heads, int32_t n_kv_heads, int32_t cache_len) {
  return n_q_heads > 0 && n_kv_heads > 0 && n_q_heads % n_kv_heads == 0 && n_q_heads / n_kv_heads <= 4 && n_q_heads <= 12 &&
         cache_len > 0 && cache_len <= 8;
}
int smoltts_k_gemm3(const SmolttsGemm3Args* a, void* stream) {
  using namespace smoltts;
  ST_REQUIRE(a, SMOLTTS_E_INVALID, "k_gemm3: null args");
  return launch_gemm3(*a, (hipStream_t)stream);
}
int smoltts_k_x3_pack(const float* x_dev, int64_t ldx, int32_t n_rows, int32_t dim, void* x3a_dev, const float* gamma_a_dev,
                      void* x3b_dev, const float* gamma_b_dev, float* ssq_dev, void* stream) {
  return smoltts::launch_x3_pack(x_dev, ldx, n_rows, dim, x3a_dev, gamma_a_dev, x3b_dev, gamma_b_dev, ssq_dev, (hipStream_t)stream);
}
}
