// Pieces shared by the Mimi decoder engine (mimi_engine.hip) and encoder (mimi_encoder.hip).
#pragma once
#include "common.h"

namespace smoltts {

constexpr int MIMI_D = 512, MIMI_HEADS = 8, MIMI_FF = 2048;

struct MimiTransformerBufs {
  float *tx, *tn, *tq, *ta, *th;  // rows x [512 | 512 | 512 | 512 | 2048]
  float *kc, *vc;                 // [n_layers][slots][8][cache_len][64]
  size_t layer_stride;            // floats between two layers' caches
  const int *row_pos, *row_slot;  // [rows]
  float* tws = nullptr;           // optional split-K workspace (4 x rows x 512 floats) for the fc2 GEMM of many-row calls
  char *kc3 = nullptr, *vc3 = nullptr;  // optional bf16x3 piece caches [n_layers][slots][8][ceil32(cache_len)] x 384 bytes (zero-filled once):
  size_t layer_stride3 = 0;             // the QKV GEMM writes them beside kc / vc, chunks of a multiple of 32 rows per slot attend over them
  int b3_products = 6;                  // 3: the many-row kernels form three bf16x3 products per operand pair (SMOLTTS_MIMI_OPT_PRODUCTS)
};

// The 8-layer pre-LayerNorm block stack of codec/transformer.py:109-150 over `rows` rows (`rows_per_slot`
// consecutive rows per slot).  The last layer's MLP output (+ residual) goes to `last_out` (row stride 512,
// slot stride `last_bstride` floats) instead of tx.  Implemented in mimi_engine.hip.
int run_mimi_transformer(const char* arena, const SmolttsMimiLayerWeights* layers, int n_layers, const float* rope,
                         int cache_len, int window, const MimiTransformerBufs& b, int rows, int rows_per_slot,
                         float* last_out, int64_t last_bstride, hipStream_t st);

// row_slot[m] = m / rows_per_slot, row_pos[m] = (slot_pos ? slot_pos[slot] : pos0) + m % rows_per_slot
int launch_mimi_rows(int n_rows, int rows_per_slot, int pos0, int* row_pos, int* row_slot, hipStream_t st,
                     const int* slot_pos = nullptr);

// Fused resnet block of a SEANet decoder stage (seanet.hip): x = raw ConvTranspose output, channel-last, with 2 halo rows in
// front of every slot's rows; writes ELU(x + conv1(ELU(conv3(ELU(x))))) to `out`.
struct MimiResblockArgs {
  int channels, batch, T;
  const float* x; int64_t x_bstride;
  const void* w2; const float* b2;   // conv k3: W3 tiles of [C/2][3C], bias
  const void* w3; const float* b3;   // conv k1: W3 tiles of [C][C/2], bias
  float* out; int64_t o_bstride;
  int b3_products;                   // 3 = three bf16x3 products per operand pair (SMOLTTS_MIMI_OPT_PRODUCTS); otherwise six
};
int launch_seanet_resblock(const MimiResblockArgs& a, hipStream_t st);

// The whole last SEANet stage (seanet_last.hip): ConvTranspose 128 -> 64 (stride 4) + resnet block + ELU + output conv -> PCM.
// `in` = ELU(stage-3 output), channel-last, with 2 halo rows (the previous call's last rows) in front of every slot's rows.
struct MimiLastStageArgs {
  int batch, T;                        // slots, input rows per slot (4 T samples each)
  const float* in; int64_t in_bstride;
  const void* wt; const float* bt;     // ConvTranspose as GEMM [256][256]: W3 tiles, bias [256]
  const void* w2; const float* b2;     // conv k3 64 -> 32: W3 tiles of [32][192], bias
  const void* w3; const float* b3;     // conv k1 32 -> 64: W3 tiles of [64][32], bias
  const float* final_w; float final_b; // output conv k3 64 -> 1: fp32 [3][64]
  float* pcm; int64_t pcm_stride;
  const int* slot_pos;
  int b3_products;                     // as MimiResblockArgs
};
int launch_seanet_last(const MimiLastStageArgs& a, hipStream_t st);

inline SmolttsGemmArgs mimi_gemm_f32(const void* w, const float* x, long ldx, int M, int N, int K, const void* w3 = nullptr) {
  SmolttsGemmArgs a;
  memset(&a, 0, sizeof(a));
  a.w_dev = w; a.w_is_fp32 = 1; a.x_dev = x; a.ldx = ldx; a.M = M; a.N = N; a.K = K;
  a.w3_dev = w3;  // bf16x3 piece tiles of the same matrix (many-row calls), or null
  return a;
}

}  // namespace smoltts
