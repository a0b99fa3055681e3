// Skinny GEMM on the fp32-input matrix cores:  out[M,N] = epi( pro(X)[M,K] . W[N,K]^T ).
//
// This one kernel family carries every dense contraction of the hot path (reference:
// nn.Linear / F.linear in modeling/model/rq_transformer.py:253,545,570,582,598 and the Mimi
// Linear/Conv1d/ConvTranspose1d layers of mlx_inference/.../codec/{transformer,conv,seanet}.py).
//
// Design (MI355X):
//  * v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate, a k-ordered fma chain, so
//    results are reproducible and within fp32 rounding of the CPU oracle (token-id parity).
//    The weight tile is the A operand (16 output columns x 4 k), the activations are B
//    (4 k x 16 rows): one MFMA retires 64 weights for 16 rows, so for batch <= 32 the kernel is
//    bound by streaming the weights, not by the matrix pipe.
//  * Weights are pre-tiled on the host in fragment order ("T16x32", include/smoltts_hip.h): one
//    wave-wide 16-byte load = one contiguous 1 KiB piece; each weight byte is read from HBM once.
//  * Workgroup = one 16-column tile x MT 16-row tiles x all of K; the waves of the workgroup split
//    K chunk-wise (chunk = 32 k) and keep U chunks of loads in flight each; partial tiles are
//    summed through LDS in fixed wave order (deterministic, no atomics).
//  * Prologues/epilogues fuse RMSNorm (the row rsqrt is applied after the contraction), ELU,
//    bias, residual, SwiGLU, GELU, layer scale and RoPE + q/KV-cache scatter into the same launch.
#include <stdlib.h>

#include "gemm_dev.h"

namespace smoltts {

// cycle stamps of workgroup (0,0): [wave][slot] = {s_memtime, s_memrealtime}
#ifdef SMOLTTS_DEBUG_HOOKS
#define STAMP(k)                                                                              \
  do {                                                                                        \
    if (p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {                        \
      p.stamps[(wave * 8 + (k)) * 2] = clock64();                                             \
      p.stamps[(wave * 8 + (k)) * 2 + 1] = wall_clock64();                                    \
    }                                                                                         \
  } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

template <bool WF32, int MT, int U, int PRO, int EPI>
__global__ __launch_bounds__(MT >= 4 ? 512 : 1024) void gemm_kernel(GemmDev p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nwaves = blockDim.x >> 6;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const int nt = blockIdx.x, mg = blockIdx.y;
  const int nchunks = p.K >> 5;
  constexpr int WBYTES = WF32 ? 2048 : 1024;

  f32x4 acc[MT];
  float ss[MT];
  const float* xrow[MT];
  bool xv[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ss[mt] = 0.f;
    const int m = (mg * MT + mt) * 16 + r;
    xv[mt] = m < p.M;
    xrow[mt] = p.x + (xv[mt] ? row_off(m, p.rows_per_batch, p.ldx, p.x_bstride) : 0);
  }
  const char* wt = p.w + (size_t)nt * nchunks * WBYTES + lane * 16;
  STAMP(0);

  // LayerNorm prologue: mean and rstd of the workgroup's rows first (two passes over each row, a wave per row; the second pass
  // re-reads the row from L1 / L2), kept in LDS behind the reduction buffers; x is normalised as it is loaded below.
  float ln_mu[MT], ln_rs[MT];
  if (PRO == SMOLTTS_PRO_LAYERNORM) {
    float* stat = smem + nwaves * MT * (256 + 16);  // [MT * 16][2]
    // (the dispatcher sends only calls of at most 16 rows and K <= 512 here: a wave has at most 4 rows, whose values it keeps in
    // registers between the two passes -- all its loads are in flight together)
    constexpr int RPW = 4;
    float4 xv4[RPW][2];
    bool on[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int rr = wave + i * nwaves, m = mg * MT * 16 + rr;
      on[i] = rr < MT * 16 && m < p.M;
      xv4[i][0] = xv4[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (on[i]) {
        const float* xr = p.x + row_off(m, p.rows_per_batch, p.ldx, p.x_bstride);
        if (lane * 4 < p.K) xv4[i][0] = *reinterpret_cast<const float4*>(xr + lane * 4);
        if (256 + lane * 4 < p.K) xv4[i][1] = *reinterpret_cast<const float4*>(xr + 256 + lane * 4);
      }
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int rr = wave + i * nwaves;
      const float4 a = xv4[i][0], c = xv4[i][1];
      float sm = ((a.x + a.y) + (a.z + a.w)) + ((c.x + c.y) + (c.z + c.w));
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
      const float mu = sm / (float)p.K;
      float v = 0.f;
      if (lane * 4 < p.K) { v = fmaf(a.x - mu, a.x - mu, v); v = fmaf(a.y - mu, a.y - mu, v); v = fmaf(a.z - mu, a.z - mu, v); v = fmaf(a.w - mu, a.w - mu, v); }
      if (256 + lane * 4 < p.K) { v = fmaf(c.x - mu, c.x - mu, v); v = fmaf(c.y - mu, c.y - mu, v); v = fmaf(c.z - mu, c.z - mu, v); v = fmaf(c.w - mu, c.w - mu, v); }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0 && rr < MT * 16) { stat[rr * 2] = on[i] ? mu : 0.f; stat[rr * 2 + 1] = on[i] ? 1.0f / sqrtf(v / (float)p.K + p.eps) : 0.f; }
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { ln_mu[mt] = stat[(mt * 16 + r) * 2]; ln_rs[mt] = stat[(mt * 16 + r) * 2 + 1]; }
  }

  for (int c0 = wave; c0 < nchunks; c0 += nwaves * U) {
    uint4 wraw[U][WF32 ? 2 : 1];
    float4 xa[U][MT][2];
    float4 ga[U][2], gb[U][2];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * nwaves;
      if (c < nchunks) {
        const char* wp = wt + (size_t)c * WBYTES;
        wraw[u][0] = *reinterpret_cast<const uint4*>(wp);
        if (WF32) wraw[u][WF32 ? 1 : 0] = *reinterpret_cast<const uint4*>(wp + 1024);
        const int k0 = c * 32 + q * 8;
        if (PRO == SMOLTTS_PRO_RMSNORM || PRO == SMOLTTS_PRO_LAYERNORM) {
          ga[u][0] = *reinterpret_cast<const float4*>(p.gamma + k0);
          ga[u][1] = *reinterpret_cast<const float4*>(p.gamma + k0 + 4);
        }
        if (PRO == SMOLTTS_PRO_LAYERNORM) {
          gb[u][0] = *reinterpret_cast<const float4*>(p.ln_b + k0);
          gb[u][1] = *reinterpret_cast<const float4*>(p.ln_b + k0 + 4);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if (xv[mt]) {
            xa[u][mt][0] = *reinterpret_cast<const float4*>(xrow[mt] + k0);
            xa[u][mt][1] = *reinterpret_cast<const float4*>(xrow[mt] + k0 + 4);
          } else {
            xa[u][mt][0] = make_float4(0.f, 0.f, 0.f, 0.f);
            xa[u][mt][1] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      } else {
        wraw[u][0] = make_uint4(0, 0, 0, 0);
        if (WF32) wraw[u][WF32 ? 1 : 0] = make_uint4(0, 0, 0, 0);
        ga[u][0] = ga[u][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        gb[u][0] = gb[u][1] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          xa[u][mt][0] = make_float4(0.f, 0.f, 0.f, 0.f);
          xa[u][mt][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float wv[8];
      if (WF32) {
        wv[0] = __uint_as_float(wraw[u][0].x); wv[1] = __uint_as_float(wraw[u][0].y);
        wv[2] = __uint_as_float(wraw[u][0].z); wv[3] = __uint_as_float(wraw[u][0].w);
        wv[4] = __uint_as_float(wraw[u][WF32 ? 1 : 0].x); wv[5] = __uint_as_float(wraw[u][WF32 ? 1 : 0].y);
        wv[6] = __uint_as_float(wraw[u][WF32 ? 1 : 0].z); wv[7] = __uint_as_float(wraw[u][WF32 ? 1 : 0].w);
      } else {
        wv[0] = bf16_lo(wraw[u][0].x); wv[1] = bf16_hi(wraw[u][0].x);
        wv[2] = bf16_lo(wraw[u][0].y); wv[3] = bf16_hi(wraw[u][0].y);
        wv[4] = bf16_lo(wraw[u][0].z); wv[5] = bf16_hi(wraw[u][0].z);
        wv[6] = bf16_lo(wraw[u][0].w); wv[7] = bf16_hi(wraw[u][0].w);
      }
      float gv[8], bv[8];
      if (PRO == SMOLTTS_PRO_RMSNORM || PRO == SMOLTTS_PRO_LAYERNORM) {
        gv[0] = ga[u][0].x; gv[1] = ga[u][0].y; gv[2] = ga[u][0].z; gv[3] = ga[u][0].w;
        gv[4] = ga[u][1].x; gv[5] = ga[u][1].y; gv[6] = ga[u][1].z; gv[7] = ga[u][1].w;
      }
      if (PRO == SMOLTTS_PRO_LAYERNORM) {
        bv[0] = gb[u][0].x; bv[1] = gb[u][0].y; bv[2] = gb[u][0].z; bv[3] = gb[u][0].w;
        bv[4] = gb[u][1].x; bv[5] = gb[u][1].y; bv[6] = gb[u][1].z; bv[7] = gb[u][1].w;
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float xvv[8] = {xa[u][mt][0].x, xa[u][mt][0].y, xa[u][mt][0].z, xa[u][mt][0].w,
                        xa[u][mt][1].x, xa[u][mt][1].y, xa[u][mt][1].z, xa[u][mt][1].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float xj = xvv[j];
          if (PRO == SMOLTTS_PRO_RMSNORM) {
            ss[mt] = fmaf(xj, xj, ss[mt]);
            xj *= gv[j];
          } else if (PRO == SMOLTTS_PRO_ELU) {
            xj = elu1(xj);
          } else if (PRO == SMOLTTS_PRO_LAYERNORM) {
            xj = (xj - ln_mu[mt]) * ln_rs[mt] * gv[j] + bv[j];  // (a chunk beyond K carries zero weights: its value is irrelevant)
          }
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], xj, acc[mt], 0, 0, 0);
        }
      }
    }
  }

  STAMP(1);
  // ---- cross-wave reduction (fixed order): red4[wave][mt][lane] holds the lane's 4 accumulators
  STAMP(2);
  float4* red4 = reinterpret_cast<float4*>(smem);
  float* ssred = smem + nwaves * MT * 256;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    red4[(wave * MT + mt) * 64 + lane] = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
    if (PRO == SMOLTTS_PRO_RMSNORM) {
      float s = ss[mt];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (lane < 16) ssred[(wave * MT + mt) * 16 + lane] = s;
    }
  }
  STAMP(3);
  __syncthreads();
  STAMP(4);
  if (tid >= 64 * MT) return;
  const int mt = tid >> 6;  // lane, r, q keep their meaning
  const int m = (mg * MT + mt) * 16 + r;
  if (m >= p.M) return;
  // all partials are fetched before the first add (one LDS round trip); waves beyond nwaves add +0
  float4 part[16];
  float sp[16];
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const int ww = w < nwaves ? w : 0;
    part[w] = red4[(ww * MT + mt) * 64 + lane];
    if (PRO == SMOLTTS_PRO_RMSNORM) sp[w] = ssred[(ww * MT + mt) * 16 + r];
  }
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const bool on = w < nwaves;
    v[0] += on ? part[w].x : 0.f;
    v[1] += on ? part[w].y : 0.f;
    v[2] += on ? part[w].z : 0.f;
    v[3] += on ? part[w].w : 0.f;
    if (PRO == SMOLTTS_PRO_RMSNORM) tot += on ? sp[w] : 0.f;
  }
  if (PRO == SMOLTTS_PRO_RMSNORM) {
    const float rstd = 1.0f / sqrtf(tot / (float)p.K + p.eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] *= rstd;
  }
  const int n0 = nt * 16 + q * 4;
  if (n0 >= p.N) return;  // N is padded to 16 in the tiles only; N % 4 == 0 is required
  if (p.bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (n0 + i < p.N) v[i] += p.bias[n0 + i];
  }
  const long orow = row_off(m, p.rows_per_batch, p.ldo, p.o_bstride);
  const long rrow = (EPI == SMOLTTS_EPI_RESID || EPI == SMOLTTS_EPI_SCALE_RESID)
                        ? row_off(m, p.rows_per_batch, p.ldr, p.r_bstride) : 0;

  if (EPI == SMOLTTS_EPI_STORE) {
    if (n0 + 4 <= p.N) {
      if (p.raw_out)
        *reinterpret_cast<float4*>(p.raw_out + row_off(m, p.rows_per_batch, p.ldo, p.raw_bstride) + n0) = make_float4(v[0], v[1], v[2], v[3]);
      if (p.elu_out) { v[0] = elu1(v[0]); v[1] = elu1(v[1]); v[2] = elu1(v[2]); v[3] = elu1(v[3]); }
      *reinterpret_cast<float4*>(p.out + orow + n0) = make_float4(v[0], v[1], v[2], v[3]);
    } else {  // N < 4 (the final 1-channel conv): scalar tail
      for (int i = 0; i < 4 && n0 + i < p.N; ++i) p.out[orow + n0 + i] = v[i];
    }
  } else if (EPI == SMOLTTS_EPI_RESID) {
    const float4 rr = *reinterpret_cast<const float4*>(p.resid + rrow + n0);
    float4 o = make_float4(rr.x + v[0], rr.y + v[1], rr.z + v[2], rr.w + v[3]);
    if (p.elu_out) o = make_float4(elu1(o.x), elu1(o.y), elu1(o.z), elu1(o.w));
    *reinterpret_cast<float4*>(p.out + orow + n0) = o;
  } else if (EPI == SMOLTTS_EPI_SCALE_RESID) {
    const float4 rr = *reinterpret_cast<const float4*>(p.resid + rrow + n0);
    const float4 sc = *reinterpret_cast<const float4*>(p.scale + n0);
    *reinterpret_cast<float4*>(p.out + orow + n0) =
        make_float4(rr.x + sc.x * v[0], rr.y + sc.y * v[1], rr.z + sc.z * v[2], rr.w + sc.w * v[3]);
  } else if (EPI == SMOLTTS_EPI_GELU) {
    *reinterpret_cast<float4*>(p.out + orow + n0) = make_float4(gelu1(v[0]), gelu1(v[1]), gelu1(v[2]), gelu1(v[3]));
  } else if (EPI == SMOLTTS_EPI_SWIGLU) {
    *reinterpret_cast<float2*>(p.out + orow + (n0 >> 1)) = make_float2(silu1(v[0]) * v[1], silu1(v[2]) * v[3]);
  } else if (EPI == SMOLTTS_EPI_QKV_ROPE) {
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
  }
  STAMP(5);
}

// ---------------------------------------------------------------------------------------------
// Many-row variant for the Mimi decoder (M = slots x time in the thousands to millions, N <= a few
// hundred, fp32 weights): the skinny kernel above re-reads the activation rows from L2 for every
// 16-column tile and needs ~64 B/clk/CU of operands, which the vector-memory path does not deliver.
// Here a workgroup owns 64*WM rows x 16*NTW*WN columns: each 32-k activation chunk goes through LDS
// once (in B-fragment order, read back conflict-free with two ds_read_b128 per fragment) and is shared
// by the WN waves that hold different column tiles; every wave accumulates the whole K itself, so
// there is no cross-wave reduction and the epilogue stores straight from the accumulators.
template <int NTW, int EPI>
__global__ __launch_bounds__(256) void gemm_rows_kernel(GemmDev p, int WN) {
  // [row group][row][8 float4], the float4 column xor-swizzled with (row >> 1) & 7: the 16-lane phases of both the
  // staging writes (2 rows x 8 columns) and the fragment reads (16 rows x 1 column) then hit 16 distinct bank groups
  __shared__ __attribute__((aligned(16))) float xs[4 * 64 * 8 * 4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const int WM = 4 / WN;
  const int wn = wave % WN, wm = wave / WN;
  const int nchunks = p.K >> 5;
  const int row0 = blockIdx.y * 64 * WM;
  const int tile0 = (blockIdx.x * WN + wn) * NTW;

  // staging assignment: 2*WM float4 per thread and chunk
  const float* sp[8];
  int sdst[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = tid + 256 * i;
    const int rg = idx >> 9, row = (idx & 511) >> 3, kq = idx & 7;
    const int m = row0 + rg * 64 + row;
    const bool v = i < 2 * WM && m < p.M;
    sp[i] = v ? p.x + row_off(m, p.rows_per_batch, p.ldx, p.x_bstride) + kq * 4 : nullptr;
    sdst[i] = ((rg * 64 + row) * 8 + (kq ^ ((row >> 1) & 7))) * 4;
  }
  const char* wb[NTW];
  bool wv[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    wv[t] = (tile0 + t) * 16 < p.N;
    wb[t] = p.w + (size_t)(tile0 + t) * nchunks * 2048 + lane * 16;
  }
  f32x4 acc[NTW][4];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 stage[8];
  uint4 wnext[NTW][2];
  auto fetch = [&](int c) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      stage[i] = (i < 2 * WM && sp[i]) ? *reinterpret_cast<const float4*>(sp[i] + c * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      wnext[t][0] = wv[t] ? *reinterpret_cast<const uint4*>(wb[t] + (size_t)c * 2048) : make_uint4(0, 0, 0, 0);
      wnext[t][1] = wv[t] ? *reinterpret_cast<const uint4*>(wb[t] + (size_t)c * 2048 + 1024) : make_uint4(0, 0, 0, 0);
    }
  };
  fetch(0);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();  // previous chunk fully consumed
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < 2 * WM) *reinterpret_cast<float4*>(xs + sdst[i]) = stage[i];
    float wcur[NTW][8];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      wcur[t][0] = __uint_as_float(wnext[t][0].x); wcur[t][1] = __uint_as_float(wnext[t][0].y);
      wcur[t][2] = __uint_as_float(wnext[t][0].z); wcur[t][3] = __uint_as_float(wnext[t][0].w);
      wcur[t][4] = __uint_as_float(wnext[t][1].x); wcur[t][5] = __uint_as_float(wnext[t][1].y);
      wcur[t][6] = __uint_as_float(wnext[t][1].z); wcur[t][7] = __uint_as_float(wnext[t][1].w);
    }
    __syncthreads();
    if (c + 1 < nchunks) fetch(c + 1);  // next chunk's global loads fly under this chunk's MFMAs
    float xv[4][8];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int row = mt * 16 + r, sw = (row >> 1) & 7;
      const float* xrow = xs + (wm * 64 + row) * 32;
      const float4 x0 = *reinterpret_cast<const float4*>(xrow + ((2 * q) ^ sw) * 4);
      const float4 x1 = *reinterpret_cast<const float4*>(xrow + ((2 * q + 1) ^ sw) * 4);
      xv[mt][0] = x0.x; xv[mt][1] = x0.y; xv[mt][2] = x0.z; xv[mt][3] = x0.w;
      xv[mt][4] = x1.x; xv[mt][5] = x1.y; xv[mt][6] = x1.z; xv[mt][7] = x1.w;
    }
    // consecutive MFMAs go to different accumulators (4 x NTW of them): no wait on the 40-cycle dependent latency
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
#ifndef SMOLTTS_DBG_NO_MFMA
          acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcur[t][j], xv[mt][j], acc[t][mt], 0, 0, 0);
#else
          acc[t][mt][0] += wcur[t][j] * xv[mt][j];  // bottleneck experiment: one scalar fma instead of the MFMA
#endif
        }
  }

  // ---- epilogue straight from the accumulators: lane holds out[m = rows + mt*16 + r][n0 .. n0+4)
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = row0 + wm * 64 + mt * 16 + r;
#ifdef SMOLTTS_DBG_NO_STORE
    if (acc[0][mt][0] != 12345.678f) continue;  // bottleneck experiment: (almost) never store
#endif
    if (m >= p.M) continue;
    const long orow = row_off(m, p.rows_per_batch, p.ldo, p.o_bstride);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int n0 = (tile0 + t) * 16 + q * 4;
      if (n0 >= p.N) continue;
      float v[4] = {acc[t][mt][0], acc[t][mt][1], acc[t][mt][2], acc[t][mt][3]};
      rows_epilogue<EPI>(p, m, orow, n0, v);
    }
  }
}

// Column tiles per wave of the many-row kernel.  Two, not four, even for wide outputs: 152 instead of 220 registers
// keep three workgroups per CU resident, which measured 0-11 % faster on the SEANet shapes (tools/microbench_rows.py).
static int rows_ntw(int ntiles) { return ntiles >= 8 ? 2 : 1; }

template <int EPI>
static int launch_rows(const GemmDev& d, hipStream_t stream) {
  const int ntiles = (d.N + 15) / 16;
  const int NTW = rows_ntw(ntiles);
  const int per = (ntiles + NTW - 1) / NTW;       // column-tile groups a workgroup could hold
  const int WN = per >= 4 ? 4 : (per >= 2 ? 2 : 1);
  const int WM = 4 / WN;
  const dim3 grid((per + WN - 1) / WN, (d.M + 64 * WM - 1) / (64 * WM));
  ST_REQUIRE(grid.y <= 65535, SMOLTTS_E_INVALID, "gemm: M=%d too large for one launch", d.M);
  if (NTW == 2) hipLaunchKernelGGL((gemm_rows_kernel<2, EPI>), grid, dim3(256), 0, stream, d, WN);
  else hipLaunchKernelGGL((gemm_rows_kernel<1, EPI>), grid, dim3(256), 0, stream, d, WN);
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

template <bool WF32, int MT, int U, int PRO, int EPI>
static int launch_one(const GemmDev& d, int nwaves, hipStream_t stream) {
  const dim3 grid((d.N + 15) / 16, (d.M + 16 * MT - 1) / (16 * MT));
  const size_t lds = ((size_t)nwaves * MT * (256 + 16) + (PRO == SMOLTTS_PRO_LAYERNORM ? MT * 32 : 0)) * sizeof(float);
  hipLaunchKernelGGL((gemm_kernel<WF32, MT, U, PRO, EPI>), grid, dim3(nwaves * 64), lds, stream, d);
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

template <bool WF32, int PRO, int EPI>
static int launch_mt(const GemmDev& d, int nwaves, hipStream_t stream) {
  if (d.M <= 16) return launch_one<WF32, 1, 3, PRO, EPI>(d, nwaves, stream);
  if (d.M <= 32) return launch_one<WF32, 2, 3, PRO, EPI>(d, nwaves, stream);
  return launch_one<WF32, 4, 2, PRO, EPI>(d, nwaves, stream);
}

static int launch_gemm_impl(const SmolttsGemmArgs& a, hipStream_t stream);

#ifdef SMOLTTS_DEBUG_HOOKS
// ---- diagnostic builds only (python -m smoltts_amd.build --variant hooks; never in the product library):
// hipEvent pairs around matching launches (smoltts_profile_begin/end) and the in-kernel cycle stamp buffer
namespace {
struct ProfileState {
  bool on = false;
  int pro = 0, epi = 0, n = 0, cap = 0, used = 0;
  hipEvent_t* ev = nullptr;
} g_prof;
}  // namespace

static unsigned long long* g_stamps = nullptr;

unsigned long long* debug_stamp_buffer() { return g_stamps; }

// shared with gemm3.hip: returns the event-pair index to close with profile_hook_end, or -1
int profile_hook_begin(int prologue, int epilogue, int N, hipStream_t stream) {
  const bool hit = g_prof.on && prologue == g_prof.pro && epilogue == g_prof.epi && (g_prof.n <= 0 || N == g_prof.n) &&
                   g_prof.used < g_prof.cap;
  if (!hit) return -1;
  const int i = g_prof.used++;
  (void)hipEventRecord(g_prof.ev[2 * i], stream);
  return i;
}
void profile_hook_end(int i, hipStream_t stream) {
  if (i >= 0) (void)hipEventRecord(g_prof.ev[2 * i + 1], stream);
}

int launch_gemm(const SmolttsGemmArgs& a, hipStream_t stream) {
  const int i = profile_hook_begin(a.prologue, a.epilogue, a.N, stream);
  const int rc = launch_gemm_impl(a, stream);
  profile_hook_end(i, stream);
  return rc;
}
#else
int launch_gemm(const SmolttsGemmArgs& a, hipStream_t stream) { return launch_gemm_impl(a, stream); }
#endif

static int launch_gemm_impl(const SmolttsGemmArgs& a, hipStream_t stream) {
  ST_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, SMOLTTS_E_INVALID, "gemm: empty shape M=%d N=%d K=%d", a.M, a.N, a.K);
  ST_REQUIRE(a.K % 32 == 0, SMOLTTS_E_INVALID, "gemm: K=%d must be a multiple of 32", a.K);
  ST_REQUIRE(a.N % 4 == 0 || a.N < 4, SMOLTTS_E_INVALID, "gemm: N=%d must be a multiple of 4", a.N);
  ST_REQUIRE(a.w_dev && a.x_dev, SMOLTTS_E_INVALID, "gemm: null operand");
  ST_REQUIRE(a.ldx % 4 == 0 && a.x_bstride % 4 == 0, SMOLTTS_E_INVALID,
             "gemm: x strides must keep 16-byte alignment (ldx=%ld)", (long)a.ldx);
  ST_REQUIRE(a.N < 4 || (a.ldo % 4 == 0 && a.o_bstride % 4 == 0) ||
                 (a.epilogue == SMOLTTS_EPI_SWIGLU && a.ldo % 2 == 0 && a.o_bstride % 2 == 0),
             SMOLTTS_E_INVALID, "gemm: out strides must keep 16-byte alignment (ldo=%ld)", (long)a.ldo);
  ST_REQUIRE(a.ldr % 4 == 0 && a.r_bstride % 4 == 0, SMOLTTS_E_INVALID, "gemm: resid strides must keep 16-byte alignment");
  ST_REQUIRE(a.N >= 4 || a.epilogue == SMOLTTS_EPI_STORE, SMOLTTS_E_INVALID, "gemm: N < 4 only with EPI_STORE");
  ST_REQUIRE((long)((a.M + 63) / 64) <= 65535, SMOLTTS_E_INVALID, "gemm: M=%d too large for one launch", a.M);
  GemmDev d;
  d.w = (const char*)a.w_dev; d.w3 = (const char*)a.w3_dev; d.splitk_ws = a.splitk_ws_dev; d.splitk_cap = a.splitk_ws_floats; d.ksplit = 1;
  d.x = a.x_dev; d.ldx = a.ldx; d.x_bstride = a.x_bstride;
  d.rows_per_batch = a.rows_per_batch; d.M = a.M; d.N = a.N; d.K = a.K; d.gamma = a.gamma_dev; d.eps = a.eps;
  d.bias = a.bias_dev; d.scale = a.scale_dev; d.resid = a.resid_dev; d.out = a.out_dev; d.ldo = a.ldo;
  d.raw_out = a.raw_out_dev; d.raw_bstride = a.raw_bstride; d.elu_out = a.elu_out; d.pro_elu = a.prologue == SMOLTTS_PRO_ELU;
  d.ln_w = nullptr; d.ln_b = nullptr; d.b3_products = a.b3_products;
  d.o_bstride = a.o_bstride; d.ldr = a.ldr ? a.ldr : a.ldo; d.r_bstride = a.ldr ? a.r_bstride : a.o_bstride;
  d.rope = a.rope_dev; d.row_pos = a.row_pos_dev; d.row_slot = a.row_slot_dev;
  d.kc = a.k_cache_dev; d.vc = a.v_cache_dev; d.kc3 = (char*)a.k_cache3_dev; d.vc3 = (char*)a.v_cache3_dev; d.n_q_heads = a.n_q_heads; d.n_kv_heads = a.n_kv_heads;
  d.cache_len = a.cache_len;
#ifdef SMOLTTS_DEBUG_HOOKS
  d.stamps = g_stamps;
#endif
  const int nchunks = a.K / 32;
  // waves split K: keep ~3 chunks per wave, at most 16 waves
  int nwaves = (nchunks + 2) / 3;
  nwaves = nwaves < 1 ? 1 : (nwaves > 16 ? 16 : nwaves);
  if (a.M > 32 && nwaves > 8) nwaves = 8;  // MT=4 instantiation: stay inside the register budget
  const int mt_used = a.M <= 16 ? 1 : (a.M <= 32 ? 2 : 4);
  if (nwaves < mt_used) nwaves = mt_used;  // the epilogue needs one wave per 16-row tile
  const int P = a.prologue, E = a.epilogue;
  if (P == SMOLTTS_PRO_RMSNORM) ST_REQUIRE(a.gamma_dev, SMOLTTS_E_INVALID, "gemm: RMSNorm prologue needs gamma");
  if (E == SMOLTTS_EPI_RESID || E == SMOLTTS_EPI_SCALE_RESID)
    ST_REQUIRE(a.resid_dev && a.out_dev, SMOLTTS_E_INVALID, "gemm: residual epilogue needs resid/out");
  if (E == SMOLTTS_EPI_SCALE_RESID) ST_REQUIRE(a.scale_dev, SMOLTTS_E_INVALID, "gemm: scale missing");
  if (E == SMOLTTS_EPI_QKV_ROPE)
    ST_REQUIRE(a.rope_dev && a.row_pos_dev && a.row_slot_dev && a.k_cache_dev && a.v_cache_dev && a.out_dev &&
                   a.N == (a.n_q_heads + 2 * a.n_kv_heads) * 64 && a.cache_len > 0,
               SMOLTTS_E_INVALID, "gemm: QKV_ROPE epilogue arguments inconsistent");
  else
    ST_REQUIRE(a.out_dev, SMOLTTS_E_INVALID, "gemm: null output");
  ST_REQUIRE((a.k_cache3_dev == nullptr) == (a.v_cache3_dev == nullptr), SMOLTTS_E_INVALID, "gemm: the piece caches come in pairs");

  // many rows: the LDS-staged kernel, provided its grid (64*WM rows x 16*NTW*WN columns per workgroup,
  // K not split) still fills the chip; otherwise the K-split skinny kernel has the shorter critical path
  long rows_grid = 0;
  {
    const int ntiles = (a.N + 15) / 16;
    const int NTW = rows_ntw(ntiles);
    const int per = (ntiles + NTW - 1) / NTW;
    const int WN = per >= 4 ? 4 : (per >= 2 ? 2 : 1);
    rows_grid = (long)((per + WN - 1) / WN) * ((a.M + 64 * (4 / WN) - 1) / (64 * (4 / WN)));
  }
  if (P == SMOLTTS_PRO_LAYERNORM) {
    ST_REQUIRE(a.w_is_fp32 && a.gamma_dev && a.beta_dev && a.rows_per_batch == 0 && a.K % 4 == 0, SMOLTTS_E_INVALID,
               "gemm: the LayerNorm prologue needs fp32 weights, weight and bias, and one flat row range");
    d.ln_w = a.gamma_dev; d.ln_b = a.beta_dev;
    const bool many_b3 = a.w3_dev && gemm_b3_applies(a.M, a.N, a.K, E);
    if (many_b3 && conv_xs_applies(d, E)) return launch_conv_xs(d, E, stream);  // fused (row-stationary kernel, K = 512)
    const bool many_rows = a.M >= 1024 && rows_grid >= 192;
    // (in the skinny kernel every workgroup works out the statistics of all its rows itself: worth it for a handful of rows only --
    // measured: 32 slots x 1 frame = 64 rows 0.74 -> 0.95 ms with it, 1 slot x 1 frame 0.39 -> 0.36 ms)
    if (many_b3 || many_rows || a.M > 16 || a.K > 512 || !(E == SMOLTTS_EPI_QKV_ROPE || E == SMOLTTS_EPI_GELU)) {
      // no such prologue in the kernel this shape goes to: the stand-alone LayerNorm first
      ST_REQUIRE(a.ln_scratch_dev && a.ldx == a.K, SMOLTTS_E_INVALID, "gemm: LayerNorm prologue: scratch missing or strided rows");
      ST_TRY(launch_layernorm(a.x_dev, a.gamma_dev, a.beta_dev, a.M, a.K, a.eps, a.ln_scratch_dev, stream));
      SmolttsGemmArgs b = a;
      b.prologue = SMOLTTS_PRO_NONE; b.x_dev = a.ln_scratch_dev; b.gamma_dev = nullptr;
      return launch_gemm_impl(b, stream);
    }
  }
  if (a.w_is_fp32 && a.w3_dev && (P == SMOLTTS_PRO_NONE || P == SMOLTTS_PRO_ELU) && gemm_b3_applies(a.M, a.N, a.K, E) && !(E == SMOLTTS_EPI_STORE && a.N < 4))
    return launch_gemm_b3(d, E, stream);
  if (a.w_is_fp32 && P == SMOLTTS_PRO_NONE && a.M >= 1024 && rows_grid >= 192) {
    ST_REQUIRE((long)((a.M + 63) / 64) <= 65535 * 4L, SMOLTTS_E_INVALID, "gemm: M=%d too large for one launch", a.M);
    switch (E) {
      case SMOLTTS_EPI_STORE: return launch_rows<SMOLTTS_EPI_STORE>(d, stream);
      case SMOLTTS_EPI_RESID: return launch_rows<SMOLTTS_EPI_RESID>(d, stream);
      case SMOLTTS_EPI_GELU: return launch_rows<SMOLTTS_EPI_GELU>(d, stream);
      case SMOLTTS_EPI_SCALE_RESID: return launch_rows<SMOLTTS_EPI_SCALE_RESID>(d, stream);
      case SMOLTTS_EPI_QKV_ROPE: return launch_rows<SMOLTTS_EPI_QKV_ROPE>(d, stream);
      default: break;
    }
  }
#define ST_CASE(WF, PP, EE)                                                       \
  if ((a.w_is_fp32 != 0) == WF && P == PP && E == EE) return launch_mt<WF, PP, EE>(d, nwaves, stream);
  // bf16 weights: the DualAR transformer
  ST_CASE(false, SMOLTTS_PRO_RMSNORM, SMOLTTS_EPI_QKV_ROPE)
  ST_CASE(false, SMOLTTS_PRO_NONE, SMOLTTS_EPI_RESID)
  ST_CASE(false, SMOLTTS_PRO_RMSNORM, SMOLTTS_EPI_SWIGLU)
  ST_CASE(false, SMOLTTS_PRO_RMSNORM, SMOLTTS_EPI_STORE)
  ST_CASE(false, SMOLTTS_PRO_NONE, SMOLTTS_EPI_STORE)
  // fp32 weights: the Mimi decoder
  ST_CASE(true, SMOLTTS_PRO_NONE, SMOLTTS_EPI_QKV_ROPE)
  ST_CASE(true, SMOLTTS_PRO_NONE, SMOLTTS_EPI_SCALE_RESID)
  ST_CASE(true, SMOLTTS_PRO_NONE, SMOLTTS_EPI_GELU)
  ST_CASE(true, SMOLTTS_PRO_NONE, SMOLTTS_EPI_STORE)
  ST_CASE(true, SMOLTTS_PRO_NONE, SMOLTTS_EPI_RESID)
  ST_CASE(true, SMOLTTS_PRO_ELU, SMOLTTS_EPI_STORE)
  ST_CASE(true, SMOLTTS_PRO_ELU, SMOLTTS_EPI_RESID)
  ST_CASE(true, SMOLTTS_PRO_LAYERNORM, SMOLTTS_EPI_QKV_ROPE)
  ST_CASE(true, SMOLTTS_PRO_LAYERNORM, SMOLTTS_EPI_GELU)
#undef ST_CASE
  set_error("gemm: unsupported combination w_is_fp32=%d prologue=%d epilogue=%d", a.w_is_fp32, P, E);
  return SMOLTTS_E_INVALID;
}

}  // namespace smoltts

#ifdef SMOLTTS_DEBUG_HOOKS
extern "C" {

// undocumented diagnostic: device buffer of 16 waves x 8 slots x 2 u64 receiving cycle stamps
int smoltts_debug_set_stamps(void* buf_dev) {
  smoltts::g_stamps = (unsigned long long*)buf_dev;
  return SMOLTTS_OK;
}

int smoltts_profile_begin(int32_t prologue, int32_t epilogue, int32_t n_filter, int32_t max_launches) {
  using namespace smoltts;
  ST_REQUIRE(!g_prof.on && max_launches > 0 && max_launches <= 1 << 20, SMOLTTS_E_INVALID, "profile_begin: bad state or size");
  g_prof.ev = (hipEvent_t*)calloc(2 * (size_t)max_launches, sizeof(hipEvent_t));
  ST_REQUIRE(g_prof.ev, SMOLTTS_E_INVALID, "profile_begin: out of host memory");
  for (int i = 0; i < 2 * max_launches; ++i) ST_CHECK_HIP(hipEventCreate(&g_prof.ev[i]));
  g_prof.pro = prologue; g_prof.epi = epilogue; g_prof.n = n_filter; g_prof.cap = max_launches; g_prof.used = 0;
  g_prof.on = true;
  return SMOLTTS_OK;
}

int smoltts_profile_end(float* total_ms, int32_t* n_launches) {
  using namespace smoltts;
  ST_REQUIRE(g_prof.on && total_ms && n_launches, SMOLTTS_E_INVALID, "profile_end: not profiling");
  g_prof.on = false;
  double tot = 0.0;
  int rc = SMOLTTS_OK;
  for (int i = 0; i < g_prof.used; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess ||
        hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) {
      set_error("profile_end: event %d failed", i);
      rc = SMOLTTS_E_HIP;
      break;
    }
    tot += ms;
  }
  for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[i]);
  free(g_prof.ev);
  g_prof.ev = nullptr;
  *total_ms = (float)tot;
  *n_launches = g_prof.used;
  return rc;
}

}  // extern "C"
#endif  // SMOLTTS_DEBUG_HOOKS
