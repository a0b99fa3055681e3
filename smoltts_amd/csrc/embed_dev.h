// Token embedding of one grid column (device side), shared by embed_kernel (small_ops.hip) and the frame loop's
// commit kernel (lm_engine.hip), which embeds the column it has just committed for the next frame's slow step.
//
// x[r] = E_text[col[0]] + keep * sum_k E_cb[col[1+k] + cb_first_offset + k*codebook_size]
// reference: BaseTransformer.embed (modeling/model/rq_transformer.py:205-221); MLX twin
// lm/rq_transformer.py:150-170 (mask rule differs: mask_mode 1).
#pragma once
#include "x3.h"

namespace smoltts {

struct EmbedTables {
  const uint16_t* text_emb;  // bf16 row-major [text_rows][dim]
  const uint16_t* cb_emb;    // bf16 row-major [cb_rows][dim]
  int dim, codebook_size, cb_first_offset, mask_mode, sem_start, sem_end, text_rows, cb_rows, n_code_rows;
};

// One workgroup of 256 threads per row; `c` = the row's 1 + n_code_rows ids (global or LDS); sh4: 4 floats of LDS.
__device__ __forceinline__ void embed_row(const EmbedTables& t, const int* c, int r, float* x, const EmitDev& emit, float* sh4) {
  float ss = 0.f;
  int tok = c[0];
  tok = tok < 0 ? 0 : (tok >= t.text_rows ? t.text_rows - 1 : tok);  // never read outside the table
  const bool keep = t.mask_mode == 0 ? (c[1] != 0) : (tok >= t.sem_start && tok <= t.sem_end);
  const int dim = t.dim;
  for (int d = threadIdx.x * 4; d < dim; d += blockDim.x * 4) {
    const uint2 tv = *reinterpret_cast<const uint2*>(t.text_emb + (long)tok * dim + d);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (keep) {
      // the code rows in batches of 8, all of a batch requested before the first add (a loop of load-then-add is one L2
      // round trip per codebook); added in codebook order: the summation order of vq_embeds.sum(dim=1)
      for (int k0 = 0; k0 < t.n_code_rows; k0 += 8) {
        uint2 e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = k0 + j;
          long row = (long)c[1 + (k < t.n_code_rows ? k : 0)] + t.cb_first_offset + (long)k * t.codebook_size;
          row = row < 0 ? 0 : (row >= t.cb_rows ? t.cb_rows - 1 : row);
          e[j] = k < t.n_code_rows ? *reinterpret_cast<const uint2*>(t.cb_emb + row * dim + d) : make_uint2(0, 0);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (k0 + j < t.n_code_rows) { acc.x += bf16_lo(e[j].x); acc.y += bf16_hi(e[j].x); acc.z += bf16_lo(e[j].y); acc.w += bf16_hi(e[j].y); }
        }
      }
    }
    const float4 o = make_float4(bf16_lo(tv.x) + acc.x, bf16_hi(tv.x) + acc.y, bf16_lo(tv.y) + acc.z, bf16_hi(tv.y) + acc.w);
    *reinterpret_cast<float4*>(x + (long)r * dim + d) = o;
    ss += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
    emit_x4(emit, r, d, dim >> 5, o.x, o.y, o.z, o.w);
  }
  emit_row_ssq(emit, r, dim, ss, sh4);
}

}  // namespace smoltts
