// XCD-local depth step: what if every XCD ran the whole depth transformer for its own 4 of the 32 utterances?
//
// Utterances are independent, so the 32 rows of a depth step can be dealt over the 8 XCDs (4 rows each, padded into one 16-row
// MFMA tile).  Every seam of the chain (wqkv -> attention -> wo -> w1|w3 -> w2, x 4 layers) then stays inside one XCD, whose
// 32 CUs share one L2: a hand-off is plain stores into that L2, a flag per workgroup, sc1 polls / loads served by the same L2 --
// no write-through to the fabric, no cross-XCD visibility protocol.  The price: each XCD streams ALL depth weights itself
// (69 MB per step, 8 x the traffic of today's launches: out of the Infinity Cache when the groups run in step).
//
// Grouping is by the XCD a workgroup actually runs on (s_getreg XCC_ID) plus a ticket, never by blockIdx: HIP promises no
// placement.  256 workgroups of 512 threads, one per CU.  Activations travel as fp32 rows [4][K] (a consumer loads its K slice,
// splits it into the three bf16 pieces itself: 4 rows make that cheap), weights are the T16x32 tiles of csrc/gemm3.hip.
// Modes: stream = real weight addresses; hot = every tile reads the same 24 KB (isolates the seam latency from streaming).
// `check` verifies every handed-off value (tagged constants).  Every spin is bounded and watches an abort word.
//
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/xcd_local.hip -o /tmp/xcd_local && /tmp/xcd_local
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int D = 768, INTER = 3072, NQH = 12, NKVH = 4, NLAYER = 4, NSTEP = 8, GROWS = 4, NGROUP = 8, GSIZE = 32;
constexpr int NQKV = (NQH + 2 * NKVH) * 64;  // 1280
constexpr int NWAVES = 8;
constexpr int SPIN_LIMIT = 4000000;
constexpr size_t W_QKV = (size_t)NQKV * D * 2, W_WO = (size_t)D * D * 2, W_W13 = (size_t)2 * INTER * D * 2, W_W2 = (size_t)D * INTER * 2;
constexpr size_t W_LAYER = W_QKV + W_WO + W_W13 + W_W2;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
// sc1 (aux 16): bypasses this CU's L1, served by the XCD's L2 -- the loads that read another workgroup's bytes
__device__ __forceinline__ uint4 ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 16);
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint4 ld16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  bf16x2_t h = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ float bf16_lo(uint32_t dw) { return __uint_as_float(dw << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t dw) { return __uint_as_float(dw & 0xffff0000u); }
__device__ __forceinline__ void split3_pair(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = pack_bf16(a, b);
  const float ra = a - bf16_lo(h), rb = b - bf16_hi(h);
  m = pack_bf16(ra, rb);
  l = pack_bf16(ra - bf16_lo(m), rb - bf16_hi(m));
}

struct Bufs {
  const char* w;       // all layers' weights (T16x32 tiles): per layer wqkv | wo | w13 | w2
  float *x, *qkv, *att, *h;   // per group: [NGROUP][4][768], [..][1280], [..][768], [..][3072]
  unsigned* flags;     // [NGROUP][GSIZE] last phase completed by each member
  unsigned* ctl;       // [0] abort, [1] timeouts, [2] data errors, [8 + g] members registered on XCD g
  int check, hot, flag_sc1;   // hot: 1 = every tile reads the same 24 KB, 2 = no work at all between the seams
};

// wave 0: wait until every member of the group has completed phase `seq`
__device__ __forceinline__ bool wait_group(const Bufs& b, int g, unsigned seq, int lane) {
  for (int spins = 0;; ++spins) {
    unsigned v = seq;
    if (lane < GSIZE) v = __hip_atomic_load((gu32*)(b.flags + g * GSIZE + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__all(v >= seq)) return true;
    if ((spins & 63) == 63 && __hip_atomic_load((gu32*)b.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    if (spins > SPIN_LIMIT) {
      if (lane == 0) { atomicAdd(b.ctl + 1, 1u); __hip_atomic_store((gu32*)b.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

__device__ __forceinline__ float tag_of(unsigned seq) { return 1.0f + (float)(seq & 63); }

// Activation K slice of this wave as B fragments: chunks c = wave + 8u; lane (q, r16) holds row r16 & 3, k = 32c + 8q .. +8
template <int U>
struct XFrag { uint4 p[U][3]; };

template <int U>
__device__ __forceinline__ void load_x(XFrag<U>& f, const float* x, int ld, int u0, int wave, int lane, const Bufs& b, float tag_in, unsigned& bad) {
  const int r = lane & 15, q = lane >> 4;  // ld: row stride of x in floats
  const __amdgpu_buffer_rsrc_t rx = rsrc_of(x, GROWS * ld * 4);
  const unsigned wu = __builtin_amdgcn_readfirstlane(wave);
  const unsigned voff = (unsigned)(((r & 3) * ld + q * 8) * 4);
  uint4 raw[U][2];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const unsigned soff = (wu + (u0 + u) * NWAVES) * 128u;
    raw[u][0] = ld16_sc1(rx, voff, soff);
    raw[u][1] = ld16_sc1(rx, voff + 16, soff);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const float a[8] = {__uint_as_float(raw[u][0].x), __uint_as_float(raw[u][0].y), __uint_as_float(raw[u][0].z), __uint_as_float(raw[u][0].w),
                        __uint_as_float(raw[u][1].x), __uint_as_float(raw[u][1].y), __uint_as_float(raw[u][1].z), __uint_as_float(raw[u][1].w)};
    if (b.check) {
#pragma unroll
      for (int i = 0; i < 8; ++i) bad += a[i] != tag_in;
    }
    uint32_t h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split3_pair(a[2 * i], a[2 * i + 1], h[i], m[i], l[i]);
    f.p[u][0] = make_uint4(h[0], h[1], h[2], h[3]);
    f.p[u][1] = make_uint4(m[0], m[1], m[2], m[3]);
    f.p[u][2] = make_uint4(l[0], l[1], l[2], l[3]);
  }
}

// Weight fragments of NT tiles, chunks u0 .. u0 + U of this wave
template <int NT, int U>
struct WFrag { uint4 v[NT][U]; };

template <int NT, int U>
__device__ __forceinline__ void load_w(WFrag<NT, U>& f, __amdgpu_buffer_rsrc_t rw, unsigned woff, int nchunks, const int* tiles, int u0, int wave, int lane, int hot) {
  const unsigned wu = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned tile = hot ? 0u : (unsigned)tiles[t];
      f.v[t][u] = ld16(rw, lane * 16, woff + (tile * nchunks + wu + (u0 + u) * NWAVES) * 1024u);
    }
}

// NT tiles x all of K (K / 32 / 8 = UK chunks per wave): MFMAs, cross-wave reduction, tagged fp32 rows [4][N] out
template <int NT, int UK>
__device__ __forceinline__ void mfma_tiles(f32x4 (&acc)[NT], const WFrag<NT, UK>& w, const XFrag<UK>& x) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int u = 0; u < UK; ++u) {
      const bf16x8_t a = __builtin_bit_cast(bf16x8_t, w.v[t][u]);
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8_t, x.p[u][pc]), acc[t], 0, 0, 0);
    }
}

template <int NT>
__device__ __forceinline__ void reduce_store(f32x4 (&acc)[NT], const int* tiles, int ntiles_valid, float* out, int N, float tag_out, float* red, int wave, int lane) {
  float4* red4 = reinterpret_cast<float4*>(red);
#pragma unroll
  for (int t = 0; t < NT; ++t) red4[(wave * NT + t) * 64 + lane] = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
  __syncthreads();
  if (wave == 0) {
    const int r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) {
        const float4 p = red4[(w * NT + t) * 64 + lane];
        v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
      }
      if (r < GROWS && t < ntiles_valid)
        *reinterpret_cast<float4*>(out + (size_t)r * N + tiles[t] * 16 + q * 4) =
            make_float4(v[0] * 0.f + tag_out, v[1] * 0.f + tag_out, v[2] * 0.f + tag_out, v[3] * 0.f + tag_out);
    }
  }
  __syncthreads();  // red is reused by the next group of tiles
}

__device__ __forceinline__ void publish(const Bufs& b, int g, int rank, unsigned seq) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this workgroup's stores have reached the L2
  __syncthreads();
  // a PLAIN store: the flag line stays in this XCD's L2, where the group's sc1 polls find it (an sc1 / agent-scope store would
  // write through and drop the line: every poll would then cross the fabric)
  if (threadIdx.x == 0) {
    if (b.flag_sc1) __hip_atomic_store((gu32*)(b.flags + g * GSIZE + rank), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *(volatile unsigned*)(b.flags + g * GSIZE + rank) = seq;
  }
}

// short attention of one (row, kv head) of the group; `nkeys` emulated cache rows
__device__ __forceinline__ void attn_unit(const Bufs& b, const float* qkv, float* att, int pair, int nkeys, float tag_in, float tag_out, int lane) {
  const int row = pair >> 2, h = pair & 3;
  const __amdgpu_buffer_rsrc_t rq = rsrc_of(qkv, GROWS * NQKV * 4);
  const int dl = lane & 15, kk = lane >> 4;
  float4 qv[3], kv[2], vv[2];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const uint4 t = ld16_sc1(rq, (unsigned)((row * NQKV + (h * 3 + g) * 64 + dl * 4) * 4), 0);
    qv[g] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {  // (the real cache is [slot][kv head][8][64]; the k / v columns of the group's rows stand in for it)
    const int j = (u * 4 + kk) & 3;
    const uint4 tk = ld16_sc1(rq, (unsigned)((j * NQKV + NQH * 64 + h * 64 + dl * 4) * 4), 0);
    const uint4 tv = ld16_sc1(rq, (unsigned)((j * NQKV + (NQH + NKVH) * 64 + h * 64 + dl * 4) * 4), 0);
    kv[u] = make_float4(__uint_as_float(tk.x), __uint_as_float(tk.y), __uint_as_float(tk.z), __uint_as_float(tk.w));
    vv[u] = make_float4(__uint_as_float(tv.x), __uint_as_float(tv.y), __uint_as_float(tv.z), __uint_as_float(tv.w));
  }
  if (b.check) {
    unsigned bad = 0;
#pragma unroll
    for (int g = 0; g < 3; ++g) bad += (qv[g].x != tag_in) + (qv[g].w != tag_in);
#pragma unroll
    for (int u = 0; u < 2; ++u) bad += (kv[u].x != tag_in) + (vv[u].w != tag_in);
    if (bad) atomicAdd(b.ctl + 2, bad);
  }
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    float s[2], mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float t = qv[g].x * kv[u].x + qv[g].y * kv[u].y + qv[g].z * kv[u].z + qv[g].w * kv[u].w;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) t += __shfl_xor(t, o);
      s[u] = (u * 4 + kk < nkeys) ? t * 0.125f : -INFINITY;
      mx = fmaxf(mx, s[u]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float den = 0.f, a = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float e = __expf(s[u] - mx);
      den += e;
      a += e * vv[u].x;
    }
    den += __shfl_xor(den, 16); den += __shfl_xor(den, 32);
    a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
    if (kk == 0) {
      const float o = (a / den) * 0.0f + tag_out;
      *reinterpret_cast<float4*>(att + (size_t)row * D + (h * 3 + g) * 64 + dl * 4) = make_float4(o, o, o, o);
    }
  }
}

template <bool PREFETCH>
__global__ __launch_bounds__(512) void xcd_local_chain(Bufs b, int nsteps) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // NWAVES * 4 tiles * 256 floats = 32 KB
  __shared__ int s_info[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) {
    const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;  // HW_REG_XCC_ID[3:0]: the XCD this workgroup runs on
    s_info[0] = xcc;
    s_info[1] = (int)atomicAdd(b.ctl + 8 + xcc, 1u);               // ticket inside the XCD
  }
  __syncthreads();
  const int g = s_info[0], rank = s_info[1];
  if (rank >= GSIZE) return;  // (cannot happen with one workgroup per CU on 32-CU XCDs; the host checks the census)
  float* x = b.x + (size_t)g * GROWS * D;
  float* qkv = b.qkv + (size_t)g * GROWS * NQKV;
  float* att = b.att + (size_t)g * GROWS * D;
  float* hh = b.h + (size_t)g * GROWS * 2 * INTER;
  const __amdgpu_buffer_rsrc_t rw = rsrc_of(b.w, (unsigned)(W_LAYER * NLAYER));

  // tiles of this member per GEMM (column tile indices; -1 padded -> tile 0 computed and dropped)
  int tq[3], to[2], t2[2], nq = 0, no = 0;
  for (int j = 0; j < 3; ++j) { const int t = rank + GSIZE * j; tq[j] = t < 80 ? t : 0; nq += t < 80; }
  for (int j = 0; j < 2; ++j) { const int t = rank + GSIZE * j; to[j] = t2[j] = t < 48 ? t : 0; no += t < 48; }

  unsigned seq = 0;
  auto wait_all = [&](unsigned s) -> bool {
    if (s == 0) return true;
    if (wave == 0) { const bool ok = wait_group(b, g, s, lane); if (lane == 0) s_info[2] = ok; }
    __syncthreads();
    return s_info[2] != 0;
  };
  WFrag<3, 3> wq;
  WFrag<2, 3> wo;
  WFrag<4, 3> w13a, w13b;
  WFrag<2, 4> w2a, w2b;
  if (PREFETCH) load_w(wq, rw, 0, 24, tq, 0, wave, lane, b.hot);

  for (int step = 0; step < nsteps; ++step) {
    for (int l = 0; l < NLAYER; ++l) {
      const unsigned wl = (unsigned)(l * W_LAYER);
      const unsigned wnext = (unsigned)(((l + 1) % NLAYER) * W_LAYER);
      unsigned bad = 0;
      // ---- QKV: x [4][768] -> qkv [4][1280]
      ++seq;
      {
        if (!PREFETCH) load_w(wq, rw, wl, 24, tq, 0, wave, lane, b.hot);
        if (!wait_all(seq - 1)) return;
        if (b.hot != 2) {
          XFrag<3> xf;
          load_x(xf, x, D, 0, wave, lane, b, tag_of(seq - 1), bad);
          f32x4 acc[3] = {};
          mfma_tiles(acc, wq, xf);
          if (PREFETCH) load_w(wo, rw, wl + (unsigned)W_QKV, 24, to, 0, wave, lane, b.hot);
          reduce_store(acc, tq, nq, qkv, NQKV, tag_of(seq), red, wave, lane);
        }
        publish(b, g, rank, seq);
      }
      // ---- attention: 16 (row, kv head) pairs of the group
      ++seq;
      {
        if (rank < 16) {
          if (!wait_all(seq - 1)) return;
          if (wave == 0 && b.hot != 2) attn_unit(b, qkv, att, rank, step + 1, tag_of(seq - 1), tag_of(seq), lane);
        }
        publish(b, g, rank, seq);
      }
      // ---- WO: att [4][768] -> x
      ++seq;
      {
        if (!PREFETCH) load_w(wo, rw, wl + (unsigned)W_QKV, 24, to, 0, wave, lane, b.hot);
        if (!wait_all(seq - 1)) return;
        if (b.hot != 2) {
          XFrag<3> xf;
          load_x(xf, att, D, 0, wave, lane, b, tag_of(seq - 1), bad);
          f32x4 acc[2] = {};
          mfma_tiles(acc, wo, xf);
          int t13[4];
          for (int j = 0; j < 4; ++j) t13[j] = rank * 12 + j;
          if (PREFETCH) load_w(w13a, rw, wl + (unsigned)(W_QKV + W_WO), 24, t13, 0, wave, lane, b.hot);
          reduce_store(acc, to, no, x, D, tag_of(seq), red, wave, lane);
        }
        publish(b, g, rank, seq);
      }
      // ---- W1|W3: x -> h [4][3072] (12 tiles per member, in three groups of four)
      ++seq;
      {
        int t13[3][4];
        for (int gi = 0; gi < 3; ++gi)
          for (int j = 0; j < 4; ++j) t13[gi][j] = rank * 12 + gi * 4 + j;
        if (!PREFETCH) load_w(w13a, rw, wl + (unsigned)(W_QKV + W_WO), 24, t13[0], 0, wave, lane, b.hot);
        if (!wait_all(seq - 1)) return;
        if (b.hot == 2) { publish(b, g, rank, seq); goto w2_phase; }
        XFrag<3> xf;
        load_x(xf, x, D, 0, wave, lane, b, tag_of(seq - 1), bad);
        load_w(w13b, rw, wl + (unsigned)(W_QKV + W_WO), 24, t13[1], 0, wave, lane, b.hot);
        {
          f32x4 acc[4] = {};
          mfma_tiles(acc, w13a, xf);
          load_w(w13a, rw, wl + (unsigned)(W_QKV + W_WO), 24, t13[2], 0, wave, lane, b.hot);
          reduce_store(acc, t13[0], 4, hh, 2 * INTER, tag_of(seq), red, wave, lane);
        }
        {
          f32x4 acc[4] = {};
          mfma_tiles(acc, w13b, xf);
          if (PREFETCH) load_w(w2a, rw, wl + (unsigned)(W_QKV + W_WO + W_W13), 96, t2, 0, wave, lane, b.hot);
          reduce_store(acc, t13[1], 4, hh, 2 * INTER, tag_of(seq), red, wave, lane);
        }
        {
          f32x4 acc[4] = {};
          mfma_tiles(acc, w13a, xf);
          reduce_store(acc, t13[2], 4, hh, 2 * INTER, tag_of(seq), red, wave, lane);
        }
        publish(b, g, rank, seq);
      }
      // ---- W2: h [4][3072] -> x (K = 3072: 12 chunks per wave in three batches of four)
    w2_phase:
      ++seq;
      {
        const unsigned w2off = wl + (unsigned)(W_QKV + W_WO + W_W13);
        if (!PREFETCH) load_w(w2a, rw, w2off, 96, t2, 0, wave, lane, b.hot);
        if (!wait_all(seq - 1)) return;
        if (b.hot == 2) { publish(b, g, rank, seq); continue; }
        f32x4 acc[2] = {};
        XFrag<4> xa, xb;
        load_x(xa, hh, 2 * INTER, 0, wave, lane, b, tag_of(seq - 1), bad);
        load_w(w2b, rw, w2off, 96, t2, 4, wave, lane, b.hot);
        load_x(xb, hh, 2 * INTER, 4, wave, lane, b, tag_of(seq - 1), bad);
        mfma_tiles(acc, w2a, xa);
        load_w(w2a, rw, w2off, 96, t2, 8, wave, lane, b.hot);
        load_x(xa, hh, 2 * INTER, 8, wave, lane, b, tag_of(seq - 1), bad);
        mfma_tiles(acc, w2b, xb);
        mfma_tiles(acc, w2a, xa);
        if (PREFETCH) load_w(wq, rw, wnext, 24, tq, 0, wave, lane, b.hot);
        reduce_store(acc, t2, no, x, D, tag_of(seq), red, wave, lane);
        publish(b, g, rank, seq);
      }
      if (b.check && bad) atomicAdd(b.ctl + 2, bad);
    }
  }
}

// (h is kept as [4][6144]: the W1|W3 phase writes its 384 column tiles unfolded -- the real phase halves them through SwiGLU --
// and W2 reads the first 3072 columns as its K: same traffic and arithmetic as the real phases.)

__global__ void init_bufs(Bufs b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, n = gridDim.x * blockDim.x;
  for (int k = i; k < NGROUP * GROWS * D; k += n) { b.x[k] = 1.0f; b.att[k] = 1.0f; }
  for (int k = i; k < NGROUP * GROWS * NQKV; k += n) b.qkv[k] = 1.0f;
  for (int k = i; k < NGROUP * GROWS * 2 * INTER; k += n) b.h[k] = 1.0f;
  for (int k = i; k < NGROUP * GSIZE; k += n) b.flags[k] = 0;
  if (i < 16) b.ctl[i] = 0;
}

int main(int argc, char** argv) {
  int reps = argc > 1 ? atoi(argv[1]) : 20;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %d CUs\n", prop.multiProcessorCount);
  if (prop.multiProcessorCount < 256) { printf("needs 256 CUs\n"); return 1; }
  Bufs b;
  memset(&b, 0, sizeof(b));
  char* w;
  CK(hipMalloc(&w, W_LAYER * NLAYER));
  {
    std::vector<uint16_t> h(W_LAYER * NLAYER / 2, 0x3F80);
    CK(hipMemcpy(w, h.data(), W_LAYER * NLAYER, hipMemcpyHostToDevice));
  }
  b.w = w;
  CK(hipMalloc(&b.x, NGROUP * GROWS * D * 4)); CK(hipMalloc(&b.att, NGROUP * GROWS * D * 4));
  CK(hipMalloc(&b.qkv, NGROUP * GROWS * NQKV * 4)); CK(hipMalloc(&b.h, NGROUP * GROWS * 2 * INTER * 4));
  CK(hipMalloc(&b.flags, NGROUP * GSIZE * 4)); CK(hipMalloc(&b.ctl, 64));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int nphases = NSTEP * NLAYER * 5;
  const size_t lds = NWAVES * 4 * 256 * sizeof(float);
  unsigned h_ctl[16];
  for (int mode = 0; mode < 5; ++mode) {  // streamed | hot | hot with sc1 flags | no work | no work with sc1 flags
    const int hot = mode == 0 ? 0 : (mode <= 2 ? 1 : 2);
    b.flag_sc1 = mode == 2 || mode == 4;
    for (int prefetch = 0; prefetch < 2; ++prefetch)
      for (int check = 1; check >= 0; --check) {
        if (hot == 2 && (check || prefetch)) continue;
        b.check = check; b.hot = hot;
        float best = 1e30f;
        unsigned tmo = 0, errs = 0;
        bool census_ok = true;
        const int n = check ? 2 : reps;
        for (int r = 0; r < n; ++r) {
          hipLaunchKernelGGL(init_bufs, dim3(64), dim3(256), 0, st, b);
          CK(hipEventRecord(e0, st));
          if (prefetch) hipLaunchKernelGGL(xcd_local_chain<true>, dim3(256), dim3(512), lds, st, b, NSTEP);
          else hipLaunchKernelGGL(xcd_local_chain<false>, dim3(256), dim3(512), lds, st, b, NSTEP);
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
          CK(hipMemcpy(h_ctl, b.ctl, 64, hipMemcpyDeviceToHost));
          tmo += h_ctl[1]; errs += h_ctl[2];
          for (int g = 0; g < NGROUP; ++g) census_ok = census_ok && h_ctl[8 + g] == GSIZE;
          if (h_ctl[0]) { printf("a spin gave up (timeouts %u): stopping\n", h_ctl[1]); return 2; }
        }
        const char* what = hot == 0 ? "streamed weights" : (hot == 1 ? "hot weights" : "no work between seams");
        if (check) printf("xcd-local (%s, %s flags, prefetch %d) check: data errors %u, timeouts %u, 32 workgroups on every XCD: %s\n",
                          what, b.flag_sc1 ? "sc1" : "plain", prefetch, errs, tmo, census_ok ? "yes" : "NO");
        else printf("xcd-local (%s, %s flags, prefetch %d): %.1f us total, %.2f us per phase, %.1f us per depth step of 20 phases (all 32 rows)\n",
                    what, b.flag_sc1 ? "sc1" : "plain", prefetch, best * 1e3, best * 1e3 / nphases, best * 1e3 / NSTEP);
      }
  }
  return 0;
}
