// "X3" activation operand format of the bf16-MFMA GEMM (gemm3.hip).
//
// An fp32 activation a is carried as three bf16 pieces hi + mid + lo == a (exactly: 3 x 8
// significand bits cover fp32's 24), so W(bf16) . a is computed on the bf16 matrix cores with exact
// products and fp32 accumulation -- fp32-grade results at 1/5 of the fp32-MFMA cost and with no
// conversion work in the consumer.  The producer of an activation (GEMM epilogue, attention,
// embedding) writes it once in this format, already multiplied by the consumer's RMSNorm weight.
//
// Layout: rows in tiles of 16, K in chunks of 32.  Block (mtile, chunk, piece) is 1 KiB at byte
// offset ((mtile * K/32 + chunk) * 3 + piece) * 1024 and holds, for lane l = 16*q + r, the 8 bf16
// a[16*mtile + r][32*chunk + 8q .. 8q+8) at byte l*16: exactly the B-operand fragment of
// v_mfma_f32_16x16x32_bf16, so a wave loads one block with one 16-byte-per-lane instruction.
#pragma once
#include "common.h"

namespace smoltts {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  bf16x2_t h = {(__bf16)a, (__bf16)b};  // v_cvt_pk_bf16_f32, round to nearest even
  return __builtin_bit_cast(uint32_t, h);
}

// (a, b) -> packed pairs of the three pieces
__device__ __forceinline__ void split3_pair(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = pack_bf16(a, b);
  const float ra = a - bf16_lo(h), rb = b - bf16_hi(h);
  m = pack_bf16(ra, rb);
  l = pack_bf16(ra - bf16_lo(m), rb - bf16_hi(m));
}

__device__ __forceinline__ size_t x3_offset(int m, int k, int nchunks) {  // piece 0; pieces are +1024 apart
  const int mtile = m >> 4, r = m & 15, c = k >> 5, q = (k & 31) >> 3, j0 = k & 7;
  return ((size_t)mtile * nchunks + c) * 3072 + (size_t)(q * 16 + r) * 16 + j0 * 2;
}

// write a[m][k..k+4), k % 4 == 0
__device__ __forceinline__ void x3_emit4(char* x3, int m, int k, int nchunks, float a0, float a1, float a2, float a3) {
  uint32_t h0, m0, l0, h1, m1, l1;
  split3_pair(a0, a1, h0, m0, l0);
  split3_pair(a2, a3, h1, m1, l1);
  char* p = x3 + x3_offset(m, k, nchunks);
  *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(p + 1024) = make_uint2(m0, m1);
  *reinterpret_cast<uint2*>(p + 2048) = make_uint2(l0, l1);
}

// write a[m][k..k+2), k % 2 == 0
__device__ __forceinline__ void x3_emit2(char* x3, int m, int k, int nchunks, float a0, float a1) {
  uint32_t h, mm, l;
  split3_pair(a0, a1, h, mm, l);
  char* p = x3 + x3_offset(m, k, nchunks);
  *reinterpret_cast<uint32_t*>(p) = h;
  *reinterpret_cast<uint32_t*>(p + 1024) = mm;
  *reinterpret_cast<uint32_t*>(p + 2048) = l;
}

// Emission targets of a producer of the residual stream x: up to two consumers with their own
// RMSNorm weights (nullptr gamma = no norm), plus the per-(row, 16-column tile) partial sums of
// squares from which the consumer derives rsqrt(mean(x^2) + eps).
struct EmitDev {
  char* x3a;
  const float* gamma_a;
  char* x3b;
  const float* gamma_b;
  float* ssq;   // [rows][K/16]
};

__device__ __forceinline__ void emit_x4(const EmitDev& e, int m, int k, int nchunks, float x0, float x1, float x2, float x3v) {
  if (e.x3a) {
    if (e.gamma_a) {
      const float4 g = *reinterpret_cast<const float4*>(e.gamma_a + k);
      x3_emit4(e.x3a, m, k, nchunks, x0 * g.x, x1 * g.y, x2 * g.z, x3v * g.w);
    } else {
      x3_emit4(e.x3a, m, k, nchunks, x0, x1, x2, x3v);
    }
  }
  if (e.x3b) {
    if (e.gamma_b) {
      const float4 g = *reinterpret_cast<const float4*>(e.gamma_b + k);
      x3_emit4(e.x3b, m, k, nchunks, x0 * g.x, x1 * g.y, x2 * g.z, x3v * g.w);
    } else {
      x3_emit4(e.x3b, m, k, nchunks, x0, x1, x2, x3v);
    }
  }
}

// Row-wise producers (one workgroup per row, blockDim 256) publish the whole-row sum of squares in
// entry 0 and zero the other K/16 - 1 entries.
__device__ __forceinline__ void emit_row_ssq(const EmitDev& e, int m, int dim, float local_ss, float* sh4) {
  if (!e.ssq) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s = local_ss;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) sh4[wave] = s;
  __syncthreads();
  const int nt = dim >> 4;
  for (int i = threadIdx.x; i < nt; i += blockDim.x)
    e.ssq[(size_t)m * nt + i] = i == 0 ? ((sh4[0] + sh4[1]) + sh4[2]) + sh4[3] : 0.f;
}

}  // namespace smoltts
