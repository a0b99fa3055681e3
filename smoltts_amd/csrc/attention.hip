// Single-query GQA attention over a slot's KV-cache prefix (decode step, and prefill row by row).
//
// Reference: F.scaled_dot_product_attention(q, k, v, is_causal=True) with repeat_interleave GQA
// (modeling/model/rq_transformer.py:554-568); at decode time
// mx.fast.scaled_dot_product_attention over KVCache.update_and_fetch
// (mlx_inference/.../lm/rq_transformer.py:281-295, lm/cache.py:12-22); Mimi:
// codec/transformer.py:75-96.  softmax(q.K^T / 8) . V in fp32, head_dim 64.
//
// One workgroup per (query row, kv head); the G query heads of the group share every K/V row read.
// The cache is fp32 [slot][kv head][cache_len][64], so a key is 256 contiguous bytes: 16 lanes x
// float4 read one key and a wave reads 4 keys per instruction (1 KiB, coalesced), four such loads
// in flight per wave.  Short caches (the 8-entry depth-transformer cache) run 4 waves, long ones 16.
// Softmax is computed online per lane group and merged in a fixed order (deterministic).  exp() is the
// hardware exp2 path (__expf): its error is |x| * 2^-24 relative on e^x with x <= 0, i.e. at most 2.2e-8
// absolute on a softmax weight -- below the fp32 rounding of the weights themselves.
// The result is written as fp32 rows and/or directly as the X3 operand of the following wo GEMM.
#include "gemm_dev.h"  // split3x8, mfma_b3 (the bf16x3 product scheme of the many-row GEMMs)
#include "x3.h"

namespace smoltts {

struct AttnDev {
  const float* q;
  const void* kc;   // fp32 or bf16 (template parameter KB of the long-cache kernels)
  const void* vc;
  const int* row_pos;
  const int* row_slot;
  float* out;
  char* out_x3;
  int n_q_heads, n_kv_heads, cache_len, window, score_cap;
  int iota_pos;  // >= 0 (short-cache kernel): every row r is at this position of slot r -- the depth transformer's steps; no
                 // row_pos / row_slot round trip in front of the cache loads
  float* split_part;   // attn_split_kernel with NS > 1: [pairs][NS][G * 64 + 8] partial (sums | max, denominator) records
  int* split_ticket;   // [pairs] arrival tickets, counting up across launches (NS arrivals per pair and launch)
};

// The slow cache is read once per frame and is far larger than anything that could stay cached: its loads carry the
// non-temporal hint, so that a frame's 100-400 MB of K / V rows do not sweep the weights out of the Infinity Cache
// (measured +1..2 % frames/s, profiles/r03_ab_nt_kv.txt; -DSMOLTTS_NT_KV=0 builds the variant without it).
#ifndef SMOLTTS_NT_KV
#define SMOLTTS_NT_KV 1
#endif
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_f4_stream(const float* p) {
#if SMOLTTS_NT_KV
  const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
#else
  return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ uint4 ld_u4_stream(const uint16_t* p) {
#if SMOLTTS_NT_KV
  const u32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(p));
  return make_uint4(v[0], v[1], v[2], v[3]);
#else
  return *reinterpret_cast<const uint4*>(p);
#endif
}

// 4 consecutive cache elements starting at element index e: fp32 (16 B) or bf16 (8 B, widened exactly)
template <bool KB>
__device__ __forceinline__ float4 load_kv4(const void* base, long e) {
  if (KB) {
    const uint2 v = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(base) + e);
    return make_float4(bf16_lo(v.x), bf16_hi(v.x), bf16_lo(v.y), bf16_hi(v.y));
  }
  return *reinterpret_cast<const float4*>(static_cast<const float*>(base) + e);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Sum over the 16 lanes of a DPP row, result in every lane: four v_add_f32 with DPP operands
// (quad xor 1, quad xor 2, half-row mirror, row mirror) instead of four LDS-crossbar shuffles.
__device__ __forceinline__ float row16_sum(float x) {
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true));  // row_half_mirror
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xF, 0xF, true));  // row_mirror
  return x;
}

constexpr int ATT_UN = 8;  // keys per lane group and pass: 8 K + 8 V rows in flight per wave

// Long caches: single pass with an online softmax per lane group (4 groups x nwaves per workgroup, each
// owning every (4*nwaves)-th key), K and V rows of a pass all in flight together, one rescale per
// 8-key block (9 exps per 8 keys and head); the groups' (max, sum, acc) triples are merged in fixed
// order: the 4 groups of a wave with shuffles, the waves through LDS.  Two barriers in total.
template <int G, bool KB>
__global__ __launch_bounds__(1024) void attn_kernel(AttnDev p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nwaves = blockDim.x >> 6;
  float* part = smem;                  // [16 waves][G][64]
  float* stat = smem + 16 * G * 64;    // [16 waves][G][2] (max, sum)
  const int row = blockIdx.x, h = blockIdx.y;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int kk = lane >> 4, dl = lane & 15;
  const int pos = p.row_pos[row], slot = p.row_slot[row];
  const int HD = p.n_q_heads * 64;
  if (pos < 0 || pos >= p.cache_len) {  // nothing cached for this row: defined output, no OOB
    for (int i = tid; i < G * 16; i += blockDim.x) {
      const int k = (h * G + (i >> 4)) * 64 + (i & 15) * 4;
      if (p.out) *reinterpret_cast<float4*>(p.out + (long)row * HD + k) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.out_x3) x3_emit4(p.out_x3, row, k, HD >> 5, 0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  const int j_lo = (p.window > 0 && pos + 1 > p.window) ? pos + 1 - p.window : 0;
  const int L = pos + 1 - j_lo;
  const long cbase = (((long)slot * p.n_kv_heads + h) * p.cache_len + j_lo) * 64;
  const long ebase = cbase + dl * 4;
  const int step = 4 * nwaves;

  float4 qv[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float4 t = *reinterpret_cast<const float4*>(p.q + (long)row * HD + (h * G + g) * 64 + dl * 4);
    qv[g] = make_float4(t.x * 0.125f, t.y * 0.125f, t.z * 0.125f, t.w * 0.125f);
  }
  float mx[G], den[G];
  float4 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) { mx[g] = -INFINITY; den[g] = 0.f; acc[g] = make_float4(0.f, 0.f, 0.f, 0.f); }

  for (int j0 = wave * 4 + kk; j0 < L + kk; j0 += step * ATT_UN) {  // (+kk keeps the trip count wave-uniform)
    float4 kv[ATT_UN], vv[ATT_UN];
#pragma unroll
    for (int u = 0; u < ATT_UN; ++u) {
      const int j = j0 + u * step;
      const bool ok = j < L;
      kv[u] = ok ? load_kv4<KB>(p.kc, ebase + (long)j * 64) : make_float4(0.f, 0.f, 0.f, 0.f);
      vv[u] = ok ? load_kv4<KB>(p.vc, ebase + (long)j * 64) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float s[ATT_UN];
      float bm = -INFINITY;
#pragma unroll
      for (int u = 0; u < ATT_UN; ++u) {
        float t = qv[g].x * kv[u].x;
        t = fmaf(qv[g].y, kv[u].y, t);
        t = fmaf(qv[g].z, kv[u].z, t);
        t = fmaf(qv[g].w, kv[u].w, t);
        t = row16_sum(t);
        s[u] = (j0 + u * step < L) ? t : -INFINITY;
        bm = fmaxf(bm, s[u]);
      }
      if (bm > -INFINITY) {  // uniform inside the 16-lane group
        const float mn = fmaxf(mx[g], bm);
        const float sc = __expf(mx[g] - mn);  // exp(-inf) = 0 on the first block
        float d = den[g] * sc;
        float4 a = make_float4(acc[g].x * sc, acc[g].y * sc, acc[g].z * sc, acc[g].w * sc);
#pragma unroll
        for (int u = 0; u < ATT_UN; ++u) {
          const float e = __expf(s[u] - mn);  // masked keys: exp(-inf) = 0
          d += e;
          a.x = fmaf(e, vv[u].x, a.x); a.y = fmaf(e, vv[u].y, a.y); a.z = fmaf(e, vv[u].z, a.z); a.w = fmaf(e, vv[u].w, a.w);
        }
        mx[g] = mn; den[g] = d; acc[g] = a;
      }
    }
  }

  // ---- merge the 4 lane groups of the wave (fixed order: xor 16, then xor 32)
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
      // (__shfl_xor here: the VALU exchanges of attn_split_kernel cost this kernel's G = 4 form two registers it does not have -- scratch)
      const float om = __shfl_xor(mx[g], o), od = __shfl_xor(den[g], o);
      const float ox = __shfl_xor(acc[g].x, o), oy = __shfl_xor(acc[g].y, o), oz = __shfl_xor(acc[g].z, o), ow = __shfl_xor(acc[g].w, o);
      const float mn = fmaxf(mx[g], om);
      const float sa = mn > -INFINITY ? __expf(mx[g] - mn) : 0.f, sb = mn > -INFINITY ? __expf(om - mn) : 0.f;
      den[g] = den[g] * sa + od * sb;
      acc[g] = make_float4(acc[g].x * sa + ox * sb, acc[g].y * sa + oy * sb, acc[g].z * sa + oz * sb, acc[g].w * sa + ow * sb);
      mx[g] = mn;
    }
    if (kk == 0) *reinterpret_cast<float4*>(part + (wave * G + g) * 64 + dl * 4) = acc[g];
    if (lane == 0) { stat[(wave * G + g) * 2] = mx[g]; stat[(wave * G + g) * 2 + 1] = den[g]; }
  }
  __syncthreads();
  // ---- merge the waves
  if (tid < G * 16) {
    const int g = tid >> 4, d4 = (tid & 15) * 4;
    // every wave's (max, denominator, sums) requested from LDS at once: a loop over a run-time wave count is one LDS round trip
    // per iteration on the critical path of a latency-bound launch (at most 16 waves; absent ones contribute nothing)
    float wm[16], wd[16];
    float4 wt[16];
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int ww = w < nwaves ? w : 0;
      wm[w] = w < nwaves ? stat[(ww * G + g) * 2] : -INFINITY;
      wd[w] = stat[(ww * G + g) * 2 + 1];
      wt[w] = *reinterpret_cast<const float4*>(part + (ww * G + g) * 64 + d4);
    }
    float gm = -INFINITY;
#pragma unroll
    for (int w = 0; w < 16; ++w) gm = fmaxf(gm, wm[w]);
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    float dsum = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const float sc = wm[w] > -INFINITY ? __expf(wm[w] - gm) : 0.f;
      sum.x = fmaf(wt[w].x, sc, sum.x); sum.y = fmaf(wt[w].y, sc, sum.y); sum.z = fmaf(wt[w].z, sc, sum.z); sum.w = fmaf(wt[w].w, sc, sum.w);
      dsum = fmaf(wd[w], sc, dsum);
    }
    const float inv = 1.0f / dsum;
    const int k = (h * G + g) * 64 + d4;
    // rounded once (no contraction into the X3 split): the fp32 and the X3 outputs are the same numbers
    const float ox = __fmul_rn(sum.x, inv), oy = __fmul_rn(sum.y, inv), oz = __fmul_rn(sum.z, inv), ow = __fmul_rn(sum.w, inv);
    if (p.out) *reinterpret_cast<float4*>(p.out + (long)row * HD + k) = make_float4(ox, oy, oz, ow);
    if (p.out_x3) x3_emit4(p.out_x3, row, k, HD >> 5, ox, oy, oz, ow);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Decode-time attention over a long cache, second form: (a) the keys of one (row, kv head) pair can be dealt over NS
// workgroups -- with B = 32 slots x 4 kv heads the plain kernel keeps only 128 of the 256 CUs streaming, and a CU cannot keep
// more than a few tens of KB of HBM reads in flight, so the cache arrives at about half of what the chip can stream; (b) a bf16
// cache is read with 8 lanes per key and 16-byte loads (8 elements per lane), so that it halves the load instructions as well
// as the bytes (the plain kernel keeps 16 lanes per key: 8-byte loads, same instruction count as fp32).
// The NS partial results (unnormalised sums, running maximum, denominator) meet inside the launch: every workgroup stores its
// record write-through (sc1), drains, and takes a ticket; the one whose ticket completes the pair loads all NS records (sc1)
// and merges them IN PART ORDER -- deterministic whoever arrives last -- then normalises and writes the row
// (MI355X_MICROARCH.md "Valid forms", first table row: one lane adds to one counter behind the storing wave's vmcnt(0); the
// workgroup whose add returned last loads, all loads sc1).  Tickets count up across launches (NS arrivals per pair and launch).
constexpr int ATT_PS_PAD = 8;  // floats behind the G * 64 sums of a record: (max, denominator) per head
#ifndef SMOLTTS_ATT_SPLIT_MIN_KEYS  // (timing variants: tools/ab_lib.sh; round 4 re-checked 256 / 384 / 768 against 512)
#define SMOLTTS_ATT_SPLIT_MIN_KEYS 512
#endif
constexpr int ATT_SPLIT_MIN_KEYS = SMOLTTS_ATT_SPLIT_MIN_KEYS;  // rows with fewer cached keys run on one workgroup

template <int LPK>
__device__ __forceinline__ float group_sum(float x) {  // sum over the LPK (8 or 16) lanes of a key's lane group, result in every lane
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true));  // row_half_mirror
  if (LPK == 16) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xF, 0xF, true));  // row_mirror
  return x;
}

template <bool KB> struct KVLane { static constexpr int LPK = KB ? 8 : 16, DPL = 64 / LPK; float v[DPL]; };

template <bool KB>
__device__ __forceinline__ void load_kv_lane(const void* base, long e, bool ok, KVLane<KB>& o) {
  if constexpr (KB) {  // 8 bf16 elements, widened exactly
    const uint4 t = ok ? ld_u4_stream(static_cast<const uint16_t*>(base) + e) : make_uint4(0, 0, 0, 0);
    o.v[0] = bf16_lo(t.x); o.v[1] = bf16_hi(t.x); o.v[2] = bf16_lo(t.y); o.v[3] = bf16_hi(t.y);
    o.v[4] = bf16_lo(t.z); o.v[5] = bf16_hi(t.z); o.v[6] = bf16_lo(t.w); o.v[7] = bf16_hi(t.w);
  } else {
    const float4 t = ok ? ld_f4_stream(static_cast<const float*>(base) + e) : make_float4(0.f, 0.f, 0.f, 0.f);
    o.v[0] = t.x; o.v[1] = t.y; o.v[2] = t.z; o.v[3] = t.w;
  }
}

template <int G, bool KB, int NS>
__global__ __launch_bounds__(1024) void attn_split_kernel(const int* row_pos_, const int* row_slot_, const float* q_, const void* kc_, const void* vc_,
                                                         int heads_, int cache_len_, int nwaves, AttnDev p) {
  // the head of the dependent chain (row -> position -> cache rows) in SGPRs at wave launch: common.h, kernarg_touch
  KernargTouch<2> kt;
  kernarg_touch(kt);
  p.row_pos = row_pos_; p.row_slot = row_slot_; p.q = q_; p.kc = kc_; p.vc = vc_;
  p.n_q_heads = heads_ & 0xffff; p.n_kv_heads = heads_ >> 16; p.cache_len = cache_len_;
  constexpr int LPK = KVLane<KB>::LPK, DPL = KVLane<KB>::DPL, KPI = 64 / LPK;  // lanes per key, dims per lane, keys per wave-instruction
  constexpr int PS = G * 64 + ATT_PS_PAD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* part = smem;                  // [16 waves][G][64]
  float* stat = smem + 16 * G * 64;    // [16 waves][G][2] (max, sum)
  const int row = blockIdx.x, h = blockIdx.y, prt = NS > 1 ? (int)blockIdx.z : 0;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int kk = lane / LPK, dl = lane % LPK;
  const int pos = p.row_pos[row], slot = p.row_slot[row];
  kernarg_touched(kt);
  const int HD = p.n_q_heads * 64;
  // A row whose position lies outside the cache has nothing to attend to.  This guard stands in front of every use of `pos`: the
  // cache reads below are the only accesses of this kernel whose address depends on an unvalidated input, and for pos >= cache_len
  // they run past the (slot, head)'s rows -- past the allocation for the last slot, which the ROCm runtime reports by aborting the
  // process (SIGABRT, exit 134: DESIGN.md 4.5).  Both parts of the pair leave here, so no ticket is taken and the pair's count stays even.
  // tests/test_attn_split_gpu.py covers pos = -1, cache_len, cache_len + 7.
  if (pos < 0 || pos >= p.cache_len) {
    if (prt == 0)
      for (int i = tid; i < G * 16; i += nwaves * 64) {
        const int k = (h * G + (i >> 4)) * 64 + (i & 15) * 4;
        if (p.out) *reinterpret_cast<float4*>(p.out + (long)row * HD + k) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.out_x3) x3_emit4(p.out_x3, row, k, HD >> 5, 0.f, 0.f, 0.f, 0.f);
      }
    return;
  }
  const int j_lo = (p.window > 0 && pos + 1 > p.window) ? pos + 1 - p.window : 0;
  const int L = pos + 1 - j_lo;
  // Short rows are not worth a hand-off (measured: the merge costs ~1 us per launch, the second workgroup buys nothing below a
  // few hundred keys): part 0 then takes all the keys and writes the row itself, the other parts leave; nobody takes a ticket.
  const bool shared = NS > 1 && L >= ATT_SPLIT_MIN_KEYS;
  if (NS > 1 && !shared && prt != 0) return;
  const int ns = shared ? NS : 1, my = shared ? prt : 0;
  const long cbase = (((long)slot * p.n_kv_heads + h) * p.cache_len + j_lo) * 64;
  const long ebase = cbase + dl * DPL;
  const int step = KPI * nwaves * ns;

  float qv[G][DPL];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float* qp = p.q + (long)row * HD + (h * G + g) * 64 + dl * DPL;
#pragma unroll
    for (int i = 0; i < DPL; i += 4) {
      const float4 t = *reinterpret_cast<const float4*>(qp + i);
      qv[g][i] = t.x * 0.125f; qv[g][i + 1] = t.y * 0.125f; qv[g][i + 2] = t.z * 0.125f; qv[g][i + 3] = t.w * 0.125f;
    }
  }
  float mx[G], den[G], acc[G][DPL];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    mx[g] = -INFINITY; den[g] = 0.f;
#pragma unroll
    for (int i = 0; i < DPL; ++i) acc[g][i] = 0.f;
  }

  if constexpr (KB) {
    // bf16 cache: the loaded rows stay packed (4 registers per key) and are widened where they are used -- K once per key for
    // all G heads, V once per key for all G heads -- so that 1024 threads fit their 128 registers; 4 K + 4 V rows in flight per lane
    constexpr int UNB = 4;
    for (int j0 = (my * nwaves + wave) * KPI + kk; j0 < L + kk; j0 += step * UNB) {  // (+kk keeps the trip count wave-uniform)
      uint4 kr[UNB], vr[UNB];
#pragma unroll
      for (int u = 0; u < UNB; ++u) {
        const int j = j0 + u * step;
        const bool ok = j < L;
        kr[u] = ok ? ld_u4_stream(static_cast<const uint16_t*>(p.kc) + ebase + (long)j * 64) : make_uint4(0, 0, 0, 0);
        vr[u] = ok ? ld_u4_stream(static_cast<const uint16_t*>(p.vc) + ebase + (long)j * 64) : make_uint4(0, 0, 0, 0);
      }
      float sc[G][UNB];
#pragma unroll
      for (int u = 0; u < UNB; ++u) {
        const float kf[8] = {bf16_lo(kr[u].x), bf16_hi(kr[u].x), bf16_lo(kr[u].y), bf16_hi(kr[u].y),
                             bf16_lo(kr[u].z), bf16_hi(kr[u].z), bf16_lo(kr[u].w), bf16_hi(kr[u].w)};
#pragma unroll
        for (int g = 0; g < G; ++g) {
          float t = qv[g][0] * kf[0];
#pragma unroll
          for (int i = 1; i < 8; ++i) t = fmaf(qv[g][i], kf[i], t);
          t = group_sum<LPK>(t);
          sc[g][u] = (j0 + u * step < L) ? t : -INFINITY;
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {  // one rescale per block of UNB keys
        float bm = sc[g][0];
#pragma unroll
        for (int u = 1; u < UNB; ++u) bm = fmaxf(bm, sc[g][u]);
        const float mn = fmaxf(mx[g], bm);
        const float rs = mn > -INFINITY ? __expf(mx[g] - mn) : 1.f;  // exp(-inf - finite) = 0 on the first block
        den[g] *= rs;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[g][i] *= rs;
        mx[g] = mn;
      }
#pragma unroll
      for (int u = 0; u < UNB; ++u) {
        const float vf[8] = {bf16_lo(vr[u].x), bf16_hi(vr[u].x), bf16_lo(vr[u].y), bf16_hi(vr[u].y),
                             bf16_lo(vr[u].z), bf16_hi(vr[u].z), bf16_lo(vr[u].w), bf16_hi(vr[u].w)};
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float e = mx[g] > -INFINITY ? __expf(sc[g][u] - mx[g]) : 0.f;  // masked keys: exp(-inf) = 0
          den[g] += e;
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[g][i] = fmaf(e, vf[i], acc[g][i]);
        }
      }
    }
  } else {
  for (int j0 = (my * nwaves + wave) * KPI + kk; j0 < L + kk; j0 += step * ATT_UN) {  // (+kk keeps the trip count wave-uniform)
    KVLane<KB> kv[ATT_UN], vv[ATT_UN];
#pragma unroll
    for (int u = 0; u < ATT_UN; ++u) {
      const int j = j0 + u * step;
      load_kv_lane<KB>(p.kc, ebase + (long)j * 64, j < L, kv[u]);
      load_kv_lane<KB>(p.vc, ebase + (long)j * 64, j < L, vv[u]);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float sc[ATT_UN];
      float bm = -INFINITY;
#pragma unroll
      for (int u = 0; u < ATT_UN; ++u) {
        float t = qv[g][0] * kv[u].v[0];
#pragma unroll
        for (int i = 1; i < DPL; ++i) t = fmaf(qv[g][i], kv[u].v[i], t);
        t = group_sum<LPK>(t);
        sc[u] = (j0 + u * step < L) ? t : -INFINITY;
        bm = fmaxf(bm, sc[u]);
      }
      if (bm > -INFINITY) {  // uniform inside the lane group
        const float mn = fmaxf(mx[g], bm);
        const float rs = __expf(mx[g] - mn);  // exp(-inf) = 0 on the first block
        float d = den[g] * rs;
        float a[DPL];
#pragma unroll
        for (int i = 0; i < DPL; ++i) a[i] = acc[g][i] * rs;
#pragma unroll
        for (int u = 0; u < ATT_UN; ++u) {
          const float e = __expf(sc[u] - mn);  // masked keys: exp(-inf) = 0
          d += e;
#pragma unroll
          for (int i = 0; i < DPL; ++i) a[i] = fmaf(e, vv[u].v[i], a[i]);
        }
        mx[g] = mn; den[g] = d;
#pragma unroll
        for (int i = 0; i < DPL; ++i) acc[g][i] = a[i];
      }
    }
  }

  }

  // ---- merge the KPI lane groups of the wave (fixed order: xor LPK, 2 LPK, .. 32)
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int o = LPK; o <= 32; o <<= 1) {
      const float om = lane_xor_c(mx[g], o, lane), od = lane_xor_c(den[g], o, lane);
      const float mn = fmaxf(mx[g], om);
      const float sa = mn > -INFINITY ? __expf(mx[g] - mn) : 0.f, sb = mn > -INFINITY ? __expf(om - mn) : 0.f;
      den[g] = den[g] * sa + od * sb;
#pragma unroll
      for (int i = 0; i < DPL; ++i) acc[g][i] = acc[g][i] * sa + lane_xor_c(acc[g][i], o, lane) * sb;
      mx[g] = mn;
    }
    if (kk == 0) {
#pragma unroll
      for (int i = 0; i < DPL; i += 4)
        *reinterpret_cast<float4*>(part + (wave * G + g) * 64 + dl * DPL + i) = make_float4(acc[g][i], acc[g][i + 1], acc[g][i + 2], acc[g][i + 3]);
    }
    if (lane == 0) { stat[(wave * G + g) * 2] = mx[g]; stat[(wave * G + g) * 2 + 1] = den[g]; }
  }
  __syncthreads();
  if (tid >= 64) return;  // everything below happens inside wave 0 (G * 16 <= 64 lanes carry data)
  // ---- merge the waves
  const bool on = tid < G * 16;
  const int g = on ? tid >> 4 : 0, d4 = (tid & 15) * 4;
  // (all waves' partials requested from LDS at once: see attn_kernel)
  float wm[16], wd[16];
  float4 wt[16];
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const int ww = w < nwaves ? w : 0;
    wm[w] = w < nwaves ? stat[(ww * G + g) * 2] : -INFINITY;
    wd[w] = stat[(ww * G + g) * 2 + 1];
    wt[w] = *reinterpret_cast<const float4*>(part + (ww * G + g) * 64 + d4);
  }
  float gm = -INFINITY;
#pragma unroll
  for (int w = 0; w < 16; ++w) gm = fmaxf(gm, wm[w]);
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  float dsum = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const float sc = wm[w] > -INFINITY ? __expf(wm[w] - gm) : 0.f;
    sum.x = fmaf(wt[w].x, sc, sum.x); sum.y = fmaf(wt[w].y, sc, sum.y); sum.z = fmaf(wt[w].z, sc, sum.z); sum.w = fmaf(wt[w].w, sc, sum.w);
    dsum = fmaf(wd[w], sc, dsum);
  }
  if (shared) {
    // ---- publish this part's record, take a ticket; the workgroup that completes the pair merges all parts in part order
    const int pair = row * p.n_kv_heads + h;
    const unsigned rec_bytes = (unsigned)((size_t)gridDim.x * p.n_kv_heads * NS * PS * sizeof(float));
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)p.split_part, 0, rec_bytes, 0x00020000);
    const unsigned rec0 = (unsigned)(((size_t)pair * NS) * PS * sizeof(float));
    if (on) {
      typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
      typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
      u32x4s sv; sv[0] = __float_as_uint(sum.x); sv[1] = __float_as_uint(sum.y); sv[2] = __float_as_uint(sum.z); sv[3] = __float_as_uint(sum.w);
      __builtin_amdgcn_raw_buffer_store_b128(sv, rr, rec0 + (unsigned)((prt * PS + g * 64 + d4) * 4), 0, 16);  // aux 16 = sc1 (write-through)
      if ((tid & 15) == 0) {
        u32x2s mv; mv[0] = __float_as_uint(gm); mv[1] = __float_as_uint(dsum);
        __builtin_amdgcn_raw_buffer_store_b64(mv, rr, rec0 + (unsigned)((prt * PS + G * 64 + g * 2) * 4), 0, 16);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the one storing wave has drained before its lane 0 signals
    unsigned old = 0;  // (unsigned: the count may wrap, the remainder below stays defined; sessions zero it at every end of frame)
    if (lane == 0) old = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(p.split_ticket) + pair, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
    if (((old + 1u) % (unsigned)NS) != 0u) return;  // not the last arriver of this launch
    typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
    float pm[NS], pd[NS];
    float4 ps[NS];
    float gm2 = -INFINITY;
#pragma unroll
    for (int q2 = 0; q2 < NS; ++q2) {  // every record through sc1 loads, this workgroup's own included: same bits whoever merges
      const u32x2s mv = __builtin_amdgcn_raw_buffer_load_b64(rr, rec0 + (unsigned)((q2 * PS + G * 64 + g * 2) * 4), 0, 16);
      const u32x4s sv = __builtin_amdgcn_raw_buffer_load_b128(rr, rec0 + (unsigned)((q2 * PS + g * 64 + d4) * 4), 0, 16);
      pm[q2] = __uint_as_float(mv[0]); pd[q2] = __uint_as_float(mv[1]);
      ps[q2] = make_float4(__uint_as_float(sv[0]), __uint_as_float(sv[1]), __uint_as_float(sv[2]), __uint_as_float(sv[3]));
      gm2 = fmaxf(gm2, pm[q2]);
    }
    sum = make_float4(0.f, 0.f, 0.f, 0.f);
    dsum = 0.f;
#pragma unroll
    for (int q2 = 0; q2 < NS; ++q2) {
      const float sc = pm[q2] > -INFINITY ? __expf(pm[q2] - gm2) : 0.f;
      sum.x = fmaf(ps[q2].x, sc, sum.x); sum.y = fmaf(ps[q2].y, sc, sum.y); sum.z = fmaf(ps[q2].z, sc, sum.z); sum.w = fmaf(ps[q2].w, sc, sum.w);
      dsum = fmaf(pd[q2], sc, dsum);
    }
  }
  if (on) {
    const float inv = 1.0f / dsum;
    const int k = (h * G + g) * 64 + d4;
    // rounded once (no contraction into the X3 split): the fp32 and the X3 outputs are the same numbers
    const float ox = __fmul_rn(sum.x, inv), oy = __fmul_rn(sum.y, inv), oz = __fmul_rn(sum.z, inv), ow = __fmul_rn(sum.w, inv);
    if (p.out) *reinterpret_cast<float4*>(p.out + (long)row * HD + k) = make_float4(ox, oy, oz, ow);
    if (p.out_x3) x3_emit4(p.out_x3, row, k, HD >> 5, ox, oy, oz, ow);
  }
}

// Many rows (prompt prefill, the codec transformer): flash-style attention on the fp32 matrix cores.  One workgroup per
// (16 consecutive rows, G query heads); keys in tiles of 16 dealt round-robin to its 4 waves; per tile and query head 16 v_mfma_f32_16x16x4_f32 for
// S = K Q^T and 16 for O += V^T P, every K / V tile loaded once for 16 rows x G heads (exact fp32 products, fp32
// accumulation: the same arithmetic class as the per-row kernel).  Fragment bookkeeping, lane l = 16 q + r:
//   S:  A = K  (lane supplies K[key r][16q + c] in chunk c), B = Q (Q[row r][16q + c]); D[i] = S[key 4q + i][row r]
//   O:  A = V  (V[key 4q + c][4r + t] for dim tile t), B = P (P[row r][key 4q + c] = the lane's own D[c] of S);
//       D_t[i] = O[dim 16q + 4i + t][row r]  ->  the lane owns out[row r][16q .. 16q + 16)
// so no value ever crosses lanes except the row maximum (two shuffles per tile and head).  Rows of a tile may belong to
// different slots (utterance boundaries in the packed prompt): one pass per distinct slot, the other rows masked.
template <int G, bool KB>
__global__ __launch_bounds__(256) void attn_prefill_kernel(AttnDev p, int n_rows, int nx) {  // nx = row tiles
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  // G = query heads per wave (consecutive heads of one kv group): blockIdx.y counts groups of G query heads
  // The workgroup's 4 waves share the 16 rows and take every 4th key tile each (a causal row tile late in a long prompt
  // has many: the critical path of the launch); their (max, sum, O) partials are merged through LDS in fixed order.
  __shared__ float merge[4][64][G * 18];
  // XCD-aware order (1-D grid; ids are dealt round-robin over the 8 XCDs, each with its own L2): the workgroups that read the
  // same K / V -- 4 neighbouring row tiles (one slot's rows of a codec chunk, a stretch of one prompt) x the query-head groups
  // of one kv head -- form a unit, and a unit's members get consecutive slots on ONE XCD.  With the plain 2-D grid the 4 row
  // tiles of a slot landed on 4 XCDs and each fetched the slot's K / V for itself (rocprofv3 FETCH_SIZE of the codec
  // transformer's attention: 1.0 GB per 1024-frame chunk against 0.3 GB of K / V).
  const int gq = p.n_q_heads / (p.n_kv_heads * G);          // head groups per kv head
  const int members = 4 * gq, nxg = (nx + 3) >> 2, units = nxg * p.n_kv_heads;
  const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
  const int mem = k % members, u = (k / members) * 8 + xcd;
  if (u >= units) return;  // (the grid is padded to whole groups of 8 units; uniform over the workgroup)
  const int kvh = u % p.n_kv_heads, bx = (u / p.n_kv_heads) * 4 + mem / gq;
  if (bx >= nx) return;
  const int hq0 = (kvh * gq + mem % gq) * G, row0 = bx * 16;
  const int HD = p.n_q_heads * 64, row = row0 + r;
  const bool in_range = row < n_rows;
  const int pos = in_range ? p.row_pos[row] : -1;
  const int slot = in_range ? p.row_slot[row] : -1;
  const bool valid = in_range && pos >= 0 && pos < p.cache_len;
  const int jlo = (p.window > 0 && pos + 1 > p.window) ? pos + 1 - p.window : 0;

  float qf[G][16];  // Q[row r][head g][16q .. 16q + 16), pre-scaled by 1/8
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float* qp = p.q + (long)(valid ? row : 0) * HD + (hq0 + g) * 64 + q * 16;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      const float4 t = valid ? *reinterpret_cast<const float4*>(qp + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      qf[g][c4 * 4 + 0] = t.x * 0.125f; qf[g][c4 * 4 + 1] = t.y * 0.125f; qf[g][c4 * 4 + 2] = t.z * 0.125f; qf[g][c4 * 4 + 3] = t.w * 0.125f;
    }
  }
  f32x4 o[G][4];  // o[g][t][i] = O[dim 16q + 4i + t][row r]
  float mrow[G], lpart[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    mrow[g] = -INFINITY; lpart[g] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) o[g][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  unsigned long long todo = __ballot(valid);  // rows whose slot has not been processed yet
  while (todo != 0ull) {
    const int leader = __ffsll((long long)todo) - 1;  // lanes 0..15 carry the rows: the lowest pending one picks the slot
    const int sl = __shfl(slot, leader);
    const bool mine = valid && slot == sl;
    todo &= ~__ballot(mine);
    int lo = mine ? jlo : 0x7fffffff, hi = mine ? pos : -1;
#pragma unroll
    for (int ofs = 8; ofs > 0; ofs >>= 1) {  // over the 16 rows (all four q copies hold the same values)
      lo = min(lo, __shfl_xor(lo, ofs));
      hi = max(hi, __shfl_xor(hi, ofs));
    }
    const long cbase = (((long)sl * p.n_kv_heads + kvh) * p.cache_len) * 64;
    // K[key j0 + r][16q .. +16) and V[key j0 + 4q + c][4r .. +4), c = 0..3; the next tile's rows are requested before
    // the current tile is consumed (one wave per SIMD at most: nothing else would hide the load latency)
    float4 kn[4], vn[4];
    auto fetch_tile = [&](int j0) {
      const int jk = j0 + r;
      const long ke = cbase + (long)jk * 64 + q * 16;
      const bool okk = jk <= hi;
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) kn[c4] = okk ? load_kv4<KB>(p.kc, ke + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int jv = j0 + 4 * q + c;
        vn[c] = jv <= hi ? load_kv4<KB>(p.vc, cbase + (long)jv * 64 + r * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    fetch_tile((lo & ~15) + 16 * wave);
    for (int j0 = (lo & ~15) + 16 * wave; j0 <= hi; j0 += 64) {
      float kf[16];
      float4 vf[4];
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        kf[c4 * 4 + 0] = kn[c4].x; kf[c4 * 4 + 1] = kn[c4].y; kf[c4 * 4 + 2] = kn[c4].z; kf[c4 * 4 + 3] = kn[c4].w;
        vf[c4] = vn[c4];
      }
      if (j0 + 64 <= hi) fetch_tile(j0 + 64);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 16; ++c) s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c], qf[g][c], s, 0, 0, 0);
        // s[i] = score of row r against key j0 + 4q + i; mask by this row's slot / causal range
        float sc[4];
        float bm = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int j = j0 + 4 * q + i;
          sc[i] = (mine && j >= jlo && j <= pos) ? s[i] : -INFINITY;
          bm = fmaxf(bm, sc[i]);
        }
        bm = fmaxf(bm, __shfl_xor(bm, 16));
        bm = fmaxf(bm, __shfl_xor(bm, 32));
        const float mn = fmaxf(mrow[g], bm);
        const float rs = mn > -INFINITY ? __expf(mrow[g] - mn) : 1.f;  // exp(-inf - finite) = 0 on a row's first tile
        float pr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pr[i] = mn > -INFINITY ? __expf(sc[i] - mn) : 0.f;
        lpart[g] = lpart[g] * rs + ((pr[0] + pr[1]) + (pr[2] + pr[3]));
        mrow[g] = mn;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          f32x4 acc = o[g][t];
          acc[0] *= rs; acc[1] *= rs; acc[2] *= rs; acc[3] *= rs;
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t == 0 ? vf[0].x : t == 1 ? vf[0].y : t == 2 ? vf[0].z : vf[0].w, pr[0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t == 0 ? vf[1].x : t == 1 ? vf[1].y : t == 2 ? vf[1].z : vf[1].w, pr[1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t == 0 ? vf[2].x : t == 1 ? vf[2].y : t == 2 ? vf[2].z : vf[2].w, pr[2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t == 0 ? vf[3].x : t == 1 ? vf[3].y : t == 2 ? vf[3].z : vf[3].w, pr[3], acc, 0, 0, 0);
          o[g][t] = acc;
        }
      }
    }
  }

  // merge the 4 waves' partials (same lane = same row and columns in every wave), then normalise and store:
  // the lane owns out[row r][head g][16q .. 16q + 16)
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float* mp = &merge[wave][lane][g * 18];
    mp[0] = mrow[g]; mp[1] = lpart[g];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) mp[2 + t * 4 + i] = o[g][t][i];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float gm = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) gm = fmaxf(gm, merge[w][lane][g * 18]);
    float l = 0.f;
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float* mp = &merge[w][lane][g * 18];
      const float f = mp[0] > -INFINITY ? __expf(mp[0] - gm) : 0.f;
      l = fmaf(mp[1], f, l);
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = fmaf(mp[2 + j], f, acc[j]);
    }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = valid ? 1.0f / l : 0.f;
    if (!in_range) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = (hq0 + g) * 64 + 16 * q + 4 * i;
      const float ox = valid ? __fmul_rn(acc[i], inv) : 0.f, oy = valid ? __fmul_rn(acc[4 + i], inv) : 0.f;
      const float oz = valid ? __fmul_rn(acc[8 + i], inv) : 0.f, ow = valid ? __fmul_rn(acc[12 + i], inv) : 0.f;
      if (p.out) *reinterpret_cast<float4*>(p.out + (long)row * HD + k) = make_float4(ox, oy, oz, ow);
      if (p.out_x3) x3_emit4(p.out_x3, row, k, HD >> 5, ox, oy, oz, ow);
    }
  }
}

// Many rows per slot over bf16x3 PIECE caches (the codec transformer over a chunk; layouts: include/smoltts_hip.h at
// smoltts_k_attention_rows3): the same flash-style scheme on v_mfma_f32_16x16x32_bf16 with both operands in three bf16 pieces
// (six products per pair: exact products, fp32 accumulation, dropped terms < 2^-24 relative -- gemm_dev.h), 384 instead of 1024
// matrix-pipe cycles per 16 keys and row tile, and no conversion work on K / V: their pieces were written once, by the QKV
// GEMM's epilogue.  One workgroup per (slot, head, RT consecutive 16-row tiles); its 4 waves take every 4th block of 32 keys;
// a block's K3 / V3 fragments are loaded once for all RT row tiles.  Fragment bookkeeping, lane l = 16 q + r:
//   S:  A = K3 (lane: K[key r of a 16-key tile][32 c + 8q .. + 8)), B = Q3 (Q[row r][32 c + 8q .. + 8)); D[i] = S[key 4q + i][row r]
//   O:  A = V3 (lane: V[the 8 keys of slot q][dim 16 t + r]), B = P3 (the lane's own 8 probabilities: keys 4q .. 4q + 3 of the
//       block's two 16-key tiles); D[i] = O[dim 16 t + 4q + i][row r]  ->  the lane owns out[row r][16 t + 4q .. + 4), t = 0..3
// so again no value crosses lanes except the row maximum.
#ifndef SMOLTTS_NT_KV3
#define SMOLTTS_NT_KV3 0  // (1 = the piece caches are read with the non-temporal hint: measured)
#endif
__device__ __forceinline__ uint4 ld_u4_kv3(const char* p) {
#if SMOLTTS_NT_KV3
  const u32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(p));
  return make_uint4(v[0], v[1], v[2], v[3]);
#else
  return *reinterpret_cast<const uint4*>(p);
#endif
}

struct Rows3Dev {
  const float* q;
  const char* kc3;
  const char* vc3;
  const int* row_pos;
  const int* row_slot;
  float* out;
  int n_heads, cache_len, window, rows_per_slot, n_units, members;  // units = (slot group, head); members = row groups per slot
};

template <int RT, int NP>
__global__ __launch_bounds__(256) void attn_rows3_kernel(Rows3Dev p) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  __shared__ float merge[4][64][RT * 18];
  // XCD-aware order: the row groups of one (slot, head) read the same K3 / V3: consecutive slots on ONE XCD (ids are dealt
  // round-robin over the 8 XCDs)
  const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
  const int mem = kk % p.members, unit = (kk / p.members) * 8 + xcd;
  if (unit >= p.n_units) return;  // (grid padded to whole groups of 8 units; uniform over the workgroup)
  const int sg = unit / p.n_heads, head = unit - sg * p.n_heads;
  const int row0 = sg * p.rows_per_slot + mem * 16 * RT;
  const int HD = p.n_heads * 64;
  const int slot = p.row_slot[row0];

  int pos[RT], jlo[RT];
  bool valid[RT];
  int lo = 0x7fffffff, hi = -1;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    pos[rt] = p.row_pos[row0 + rt * 16 + r];
    valid[rt] = pos[rt] >= 0 && pos[rt] < p.cache_len;
    jlo[rt] = (p.window > 0 && pos[rt] + 1 > p.window) ? pos[rt] + 1 - p.window : 0;
    if (valid[rt]) { lo = min(lo, jlo[rt]); hi = max(hi, pos[rt]); }
  }
#pragma unroll
  for (int ofs = 8; ofs > 0; ofs >>= 1) {  // over the 16 rows (all four q copies hold the same values)
    lo = min(lo, __shfl_xor(lo, ofs));
    hi = max(hi, __shfl_xor(hi, ofs));
  }

  // Q[row r][32 c + 8q .. + 8) / 8 in pieces, through LDS (the four waves work on the same rows; in registers the 12 fragments
  // would push the kernel past 256 registers = one wave per SIMD): wave w splits (row tile, chunk) pair w, w + 4, ..
  __shared__ uint4 qs[RT * 2 * 3 * 64];
  for (int pr2 = wave; pr2 < RT * 2; pr2 += 4) {
    const int rt = pr2 >> 1, c = pr2 & 1;
    const float* qp = p.q + (long)(row0 + rt * 16 + r) * HD + head * 64 + c * 32 + q * 8;
    float4 a = *reinterpret_cast<const float4*>(qp), b = *reinterpret_cast<const float4*>(qp + 4);
    a = make_float4(a.x * 0.125f, a.y * 0.125f, a.z * 0.125f, a.w * 0.125f);
    b = make_float4(b.x * 0.125f, b.y * 0.125f, b.z * 0.125f, b.w * 0.125f);
    uint4 h, m, l;
    split3x8(a, b, h, m, l);
    qs[(pr2 * 3 + 0) * 64 + lane] = h;
    qs[(pr2 * 3 + 1) * 64 + lane] = m;
    qs[(pr2 * 3 + 2) * 64 + lane] = l;
  }
  __syncthreads();
  f32x4 o[RT][4];
  float mrow[RT], lpart[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    mrow[rt] = -INFINITY; lpart[rt] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) o[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const long per_head = (long)((p.cache_len + 31) & ~31) * 384;
  const char* kb = p.kc3 + ((long)slot * p.n_heads + head) * per_head + lane * 16;
  const char* vb = p.vc3 + ((long)slot * p.n_heads + head) * per_head + lane * 16;
  uint4 kf[2][2][3], vf[4][3];  // K3 [16-key tile][chunk][piece], V3 [dim tile][piece] of the current 32-key block
#define R3_LOAD_K(PB)                                                                                   \
  _Pragma("unroll") for (int tl = 0; tl < 2; ++tl)                                                      \
    _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                       \
      _Pragma("unroll") for (int pc = 0; pc < 3; ++pc)                                                  \
        kf[tl][c][pc] = ld_u4_kv3(kb + ((long)(((PB) * 2 + tl) * 2 + c) * 3 + pc) * 1024);
#define R3_LOAD_V(PB)                                                                                   \
  _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                         \
    _Pragma("unroll") for (int pc = 0; pc < 3; ++pc)                                                    \
      vf[t][pc] = ld_u4_kv3(vb + ((long)((PB) * 4 + t) * 3 + pc) * 1024);
  const int b_hi = hi >> 5;  // hi < 0 (no valid row): no block
  int pb = (lo >> 5) + wave;
  if (hi >= 0 && pb <= b_hi) { R3_LOAD_K(pb) R3_LOAD_V(pb) }
  for (; hi >= 0 && pb <= b_hi; pb += 4) {
    f32x4 sc[RT][2];
    int fresh = 0;  // (an offset the compiler cannot see through: the Q3 reads stay in the loop instead of living in 48 registers)
    asm volatile("" : "+v"(fresh));
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) {
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          uint4 qf[3];
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) qf[pc] = qs[((rt * 2 + c) * 3 + pc) * 64 + lane + fresh];
          a = mfma_b3<NP>(kf[tl][c], qf, a);
        }
        sc[rt][tl] = a;
      }
    if (pb + 4 <= b_hi) { R3_LOAD_K(pb + 4) }  // (the K3 registers are free: the next block's fragments fly under the softmax and O)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float sv[8];
      float bm = -INFINITY;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int j = pb * 32 + (i >> 2) * 16 + 4 * q + (i & 3);
        sv[i] = (valid[rt] && j >= jlo[rt] && j <= pos[rt]) ? sc[rt][i >> 2][i & 3] : -INFINITY;
        bm = fmaxf(bm, sv[i]);
      }
      bm = fmaxf(bm, __shfl_xor(bm, 16));
      bm = fmaxf(bm, __shfl_xor(bm, 32));
      const float mn = fmaxf(mrow[rt], bm);
      const float rs = mn > -INFINITY ? __expf(mrow[rt] - mn) : 1.f;  // exp(-inf - finite) = 0 on a row's first block
#pragma unroll
      for (int i = 0; i < 8; ++i) sv[i] = mn > -INFINITY ? __expf(sv[i] - mn) : 0.f;
      lpart[rt] = lpart[rt] * rs + (((sv[0] + sv[1]) + (sv[2] + sv[3])) + ((sv[4] + sv[5]) + (sv[6] + sv[7])));
      mrow[rt] = mn;
      uint4 pf[3];
      split3x8(make_float4(sv[0], sv[1], sv[2], sv[3]), make_float4(sv[4], sv[5], sv[6], sv[7]), pf[0], pf[1], pf[2]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 a = o[rt][t];
        a[0] *= rs; a[1] *= rs; a[2] *= rs; a[3] *= rs;
        o[rt][t] = mfma_b3<NP>(vf[t], pf, a);
      }
    }
    if (pb + 4 <= b_hi) { R3_LOAD_V(pb + 4) }
  }
#undef R3_LOAD_K
#undef R3_LOAD_V

  // merge the 4 waves' partials (same lane = same rows and columns in every wave), then normalise and store
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    float* mp = &merge[wave][lane][rt * 18];
    mp[0] = mrow[rt]; mp[1] = lpart[rt];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) mp[2 + t * 4 + i] = o[rt][t][i];
  }
  __syncthreads();
  if (wave >= RT) return;  // wave w finishes row tile w
  {
    const int rt = wave;
    float gm = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) gm = fmaxf(gm, merge[w][lane][rt * 18]);
    float l = 0.f;
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float* mp = &merge[w][lane][rt * 18];
      const float f = mp[0] > -INFINITY ? __expf(mp[0] - gm) : 0.f;
      l = fmaf(mp[1], f, l);
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = fmaf(mp[2 + j], f, acc[j]);
    }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const bool ok = l > 0.f;
    const float inv = ok ? 1.0f / l : 0.f;
    float* orow = p.out + (long)(row0 + rt * 16 + r) * HD + head * 64 + 4 * q;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      *reinterpret_cast<float4*>(orow + 16 * t) = make_float4(ok ? __fmul_rn(acc[t * 4 + 0], inv) : 0.f, ok ? __fmul_rn(acc[t * 4 + 1], inv) : 0.f,
                                                              ok ? __fmul_rn(acc[t * 4 + 2], inv) : 0.f, ok ? __fmul_rn(acc[t * 4 + 3], inv) : 0.f);
  }
}

// Caches of at most 16 entries (the depth transformer's per-frame cache, lm/generate.py:112): one wave
// per (row, kv head), all K/V rows loaded up front, no LDS and no barrier -- the kernel is a single
// memory round trip plus wave shuffles.
template <int G, int UN>
__global__ __launch_bounds__(256) void attn_short_kernel(AttnDev p, int n_pairs) {
  // one wave per (row, QUERY head): the G heads of a kv group used to be worked off one after the other inside one wave -- three
  // softmaxes in a row on the critical path of a launch that is nothing but latency; each wave now loads its kv head's <= 16
  // K / V rows itself (L2 hits: the G waves of a group sit in the same workgroup or the next) and does one head
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int kk = lane >> 4, dl = lane & 15;
  const int pair = blockIdx.x * 4 + wave;  // (row, query head)
  if (pair >= n_pairs) return;
  const int row = pair / p.n_q_heads, hq = pair - row * p.n_q_heads, h = hq / G;
  const int pos = p.iota_pos >= 0 ? p.iota_pos : p.row_pos[row], slot = p.iota_pos >= 0 ? row : p.row_slot[row];
  const int HD = p.n_q_heads * 64;
  const bool ok = pos >= 0 && pos < p.cache_len;
  const int j_lo = (ok && p.window > 0 && pos + 1 > p.window) ? pos + 1 - p.window : 0;
  const int L = ok ? pos + 1 - j_lo : 0;
  const long cbase = (((long)slot * p.n_kv_heads + h) * p.cache_len + j_lo) * 64;
  float4 kv[UN], vv[UN];
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    const int j = u * 4 + kk;
    const bool v = j < L;
    kv[u] = v ? load_kv4<false>(p.kc, cbase + (long)j * 64 + dl * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    vv[u] = v ? load_kv4<false>(p.vc, cbase + (long)j * 64 + dl * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float4 t = *reinterpret_cast<const float4*>(p.q + (long)row * HD + hq * 64 + dl * 4);
  const float4 qv = make_float4(t.x * 0.125f, t.y * 0.125f, t.z * 0.125f, t.w * 0.125f);
  float s[UN];
  float mx = -INFINITY;
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    float d = qv.x * kv[u].x;
    d = fmaf(qv.y, kv[u].y, d);
    d = fmaf(qv.z, kv[u].z, d);
    d = fmaf(qv.w, kv[u].w, d);
    d = row16_sum(d);
    s[u] = d;
    if (u * 4 + kk < L) mx = fmaxf(mx, d);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float den = 0.f;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    const float e = (u * 4 + kk < L) ? __expf(s[u] - mx) : 0.f;
    den += e;
    a.x = fmaf(e, vv[u].x, a.x); a.y = fmaf(e, vv[u].y, a.y); a.z = fmaf(e, vv[u].z, a.z); a.w = fmaf(e, vv[u].w, a.w);
  }
  den += __shfl_xor(den, 16);
  den += __shfl_xor(den, 32);
  a.x += __shfl_xor(a.x, 16); a.y += __shfl_xor(a.y, 16); a.z += __shfl_xor(a.z, 16); a.w += __shfl_xor(a.w, 16);
  a.x += __shfl_xor(a.x, 32); a.y += __shfl_xor(a.y, 32); a.z += __shfl_xor(a.z, 32); a.w += __shfl_xor(a.w, 32);
  if (kk == 0) {
    const float inv = L > 0 ? 1.0f / den : 0.f;
    const int k = hq * 64 + dl * 4;
    const float ox = __fmul_rn(a.x, inv), oy = __fmul_rn(a.y, inv), oz = __fmul_rn(a.z, inv), ow = __fmul_rn(a.w, inv);
    if (p.out) *reinterpret_cast<float4*>(p.out + (long)row * HD + k) = make_float4(ox, oy, oz, ow);
    if (p.out_x3) x3_emit4(p.out_x3, row, k, HD >> 5, ox, oy, oz, ow);
  }
}

int launch_attention(const float* q, const void* kc, const void* vc, const int32_t* row_pos, const int32_t* row_slot,
                     int n_rows, int n_q_heads, int n_kv_heads, int cache_len, int window, float* out, void* out_x3,
                     hipStream_t stream, int kv_format, int iota_pos, float* split_part, int32_t* split_ticket) {
  ST_REQUIRE(kv_format == SMOLTTS_KV_F32 || (kv_format == SMOLTTS_KV_BF16 && cache_len > 16), SMOLTTS_E_INVALID,
             "attention: kv_format %d unsupported here (bf16 caches need more than 16 entries)", kv_format);
  const bool kb = kv_format == SMOLTTS_KV_BF16;
  ST_REQUIRE(q && kc && vc && row_pos && row_slot && (out || out_x3), SMOLTTS_E_INVALID, "attention: null pointer");
  ST_REQUIRE(n_rows > 0 && n_kv_heads > 0 && n_q_heads % n_kv_heads == 0 && cache_len > 0, SMOLTTS_E_INVALID,
             "attention: bad shape rows=%d q_heads=%d kv_heads=%d cache_len=%d", n_rows, n_q_heads, n_kv_heads, cache_len);
  const int G = n_q_heads / n_kv_heads;
  AttnDev d{q, kc, vc, row_pos, row_slot, out, (char*)out_x3, n_q_heads, n_kv_heads, cache_len, window, 0, -1, split_part, split_ticket};
  if (iota_pos >= 0 && cache_len <= 16) d.iota_pos = iota_pos;  // (only the short-cache kernel takes it)
  d.score_cap = (window > 0 && window < cache_len) ? window : cache_len;
  d.score_cap = (d.score_cap + 3) & ~3;
  // few workgroups (decode: rows x kv heads ~ 128): spread each cache over 16 waves; many rows
  // (prefill, Mimi): 4 waves per workgroup and let the grid fill the chip
  const int nwaves = ((long)n_rows * n_kv_heads >= 1024 || d.score_cap <= 64) ? 4 : 16;
  const size_t lds = ((size_t)16 * G * 64 + 16 * G * 2 + 16) * sizeof(float);
  ST_REQUIRE(lds <= 150 * 1024, SMOLTTS_E_CAPACITY, "attention: %zu bytes of LDS needed for %d keys x %d heads", lds,
             d.score_cap, G);
  ST_REQUIRE(n_kv_heads <= 65535, SMOLTTS_E_INVALID, "attention: grid too large");
  if (cache_len <= 16) {
    const int n_pairs = n_rows * n_q_heads;  // one wave per (row, query head)
    const dim3 sgrid((n_pairs + 3) / 4);
#define ST_SHORT(GG)                                                                                       \
  case GG:                                                                                                 \
    if (cache_len <= 8) hipLaunchKernelGGL((attn_short_kernel<GG, 2>), sgrid, dim3(256), 0, stream, d, n_pairs); \
    else hipLaunchKernelGGL((attn_short_kernel<GG, 4>), sgrid, dim3(256), 0, stream, d, n_pairs);           \
    break;
    switch (G) {
      ST_SHORT(1)
      ST_SHORT(2)
      ST_SHORT(3)
      ST_SHORT(4)
#undef ST_SHORT
      default:
        set_error("attention: GQA group size %d not instantiated (1..4)", G);
        return SMOLTTS_E_INVALID;
    }
    ST_CHECK_HIP(hipGetLastError());
    return SMOLTTS_OK;
  }
  static const bool no_mfma_attn = ST_KNOB_INT("SMOLTTS_NO_MFMA_ATTN", 0) != 0;  // experiments (knobs builds only)
  if ((long)n_rows * n_kv_heads >= 1024 && !no_mfma_attn) {  // prompt prefill, codec transformer
    // one query head per wave: the longest row tile (the critical path of the launch) is G times shorter, and the K / V
    // tiles re-read by the G waves of a kv group come from L2
    const int nx = (n_rows + 15) / 16, units = ((nx + 3) / 4) * n_kv_heads, members = 4 * (n_q_heads / n_kv_heads);
    const dim3 pgrid((unsigned)(((units + 7) / 8) * 8 * members));
    if (kb) hipLaunchKernelGGL((attn_prefill_kernel<1, true>), pgrid, dim3(256), 0, stream, d, n_rows, nx);
    else hipLaunchKernelGGL((attn_prefill_kernel<1, false>), pgrid, dim3(256), 0, stream, d, n_rows, nx);
    ST_CHECK_HIP(hipGetLastError());
    return SMOLTTS_OK;
  }
  // Few (row, kv head) pairs over a long cache (the decode frame's slow attention at B <= 32): the keys of a pair go to two
  // workgroups so that all CUs stream the cache, merged inside the launch (attn_split_kernel); a bf16 cache always takes that
  // kernel (8 lanes per key).  The caller provides the records / tickets (sized for ATT_SPLIT_MAX_PAIRS pairs).
  const bool can_split = split_part && split_ticket && (long)n_rows * n_kv_heads <= ATT_SPLIT_MAX_PAIRS && d.score_cap > 128 && nwaves == 16;
  if (kb || can_split) {
    const int NS = can_split ? 2 : 1;
    const dim3 sgrid(n_rows, n_kv_heads, NS);
#define SPLIT_ARGS d.row_pos, d.row_slot, d.q, d.kc, d.vc, d.n_q_heads | (d.n_kv_heads << 16), d.cache_len, nwaves, d
#define ST_SPLIT(GG)                                                                                                              \
  case GG:                                                                                                                          \
    if (kb && NS == 2) hipLaunchKernelGGL((attn_split_kernel<GG, true, 2>), sgrid, dim3(nwaves * 64), lds, stream, SPLIT_ARGS);              \
    else if (kb) hipLaunchKernelGGL((attn_split_kernel<GG, true, 1>), sgrid, dim3(nwaves * 64), lds, stream, SPLIT_ARGS);                    \
    else hipLaunchKernelGGL((attn_split_kernel<GG, false, 2>), sgrid, dim3(nwaves * 64), lds, stream, SPLIT_ARGS);                           \
    break;
    ST_REQUIRE(lds <= 64 * 1024, SMOLTTS_E_CAPACITY, "attention: %zu bytes of LDS", lds);
    switch (G) {
      ST_SPLIT(1)
      ST_SPLIT(2)
      ST_SPLIT(3)
      ST_SPLIT(4)
      default:
        set_error("attention: GQA group size %d not instantiated (1..4)", G);
        return SMOLTTS_E_INVALID;
    }
#undef ST_SPLIT
#undef SPLIT_ARGS
    ST_CHECK_HIP(hipGetLastError());
    return SMOLTTS_OK;
  }
  const dim3 grid(n_rows, n_kv_heads);
#define ST_ATTN(GG)                                                                          \
  case GG:                                                                                    \
    if (lds > 64 * 1024) {                                                                    \
      ST_CHECK_HIP(hipFuncSetAttribute((const void*)attn_kernel<GG, false>,                   \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      ST_CHECK_HIP(hipFuncSetAttribute((const void*)attn_kernel<GG, true>,                    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    }                                                                                         \
    if (kb) hipLaunchKernelGGL((attn_kernel<GG, true>), grid, dim3(nwaves * 64), lds, stream, d); \
    else hipLaunchKernelGGL((attn_kernel<GG, false>), grid, dim3(nwaves * 64), lds, stream, d);    \
    break;
  switch (G) {
    ST_ATTN(1)
    ST_ATTN(2)
    ST_ATTN(3)
    ST_ATTN(4)
    default:
      set_error("attention: GQA group size %d not instantiated (1..4)", G);
      return SMOLTTS_E_INVALID;
  }
#undef ST_ATTN
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

int launch_attention_rows3(const float* q, const void* kc3, const void* vc3, const int32_t* row_pos, const int32_t* row_slot, int n_rows,
                           int rows_per_slot, int n_heads, int cache_len, int window, float* out, hipStream_t stream, int b3_products) {
  ST_REQUIRE(q && kc3 && vc3 && row_pos && row_slot && out, SMOLTTS_E_INVALID, "attention_rows3: null pointer");
  ST_REQUIRE(n_rows > 0 && n_heads > 0 && cache_len > 0 && rows_per_slot > 0 && rows_per_slot % 32 == 0 && n_rows % rows_per_slot == 0,
             SMOLTTS_E_INVALID, "attention_rows3: bad shape rows=%d rows_per_slot=%d heads=%d cache_len=%d", n_rows, rows_per_slot, n_heads, cache_len);
  Rows3Dev d{q, (const char*)kc3, (const char*)vc3, row_pos, row_slot, out, n_heads, cache_len, window, rows_per_slot,
             (n_rows / rows_per_slot) * n_heads, rows_per_slot / 32};
  const long blocks = (long)((d.n_units + 7) / 8) * 8 * d.members;
  ST_REQUIRE(blocks < (1L << 30), SMOLTTS_E_INVALID, "attention_rows3: grid too large");
  if (b3_products == 3) hipLaunchKernelGGL((attn_rows3_kernel<2, 3>), dim3((unsigned)blocks), dim3(256), 0, stream, d);
  else hipLaunchKernelGGL((attn_rows3_kernel<2, 6>), dim3((unsigned)blocks), dim3(256), 0, stream, d);
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

}  // namespace smoltts
