// C-ABI glue: error reporting and the operator-level test entry points (include/smoltts_hip.h).
#include <stdarg.h>

#include "common.h"

namespace smoltts {
static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace smoltts

using namespace smoltts;

extern "C" {

const char* smoltts_last_error(void) { return g_err; }
int smoltts_abi_version(void) { return SMOLTTS_ABI_VERSION; }

int smoltts_k_gemm(const SmolttsGemmArgs* a, void* stream) {
  ST_REQUIRE(a, SMOLTTS_E_INVALID, "k_gemm: null args");
  return launch_gemm(*a, (hipStream_t)stream);
}

int smoltts_k_attention(const float* q_dev, const float* k_cache_dev, const float* v_cache_dev, const int32_t* row_pos_dev,
                        const int32_t* row_slot_dev, int32_t n_rows, int32_t n_q_heads, int32_t n_kv_heads,
                        int32_t cache_len, int32_t window, float* out_dev, void* out_x3_dev, void* stream) {
  return launch_attention(q_dev, k_cache_dev, v_cache_dev, row_pos_dev, row_slot_dev, n_rows, n_q_heads, n_kv_heads, cache_len,
                          window, out_dev, out_x3_dev, (hipStream_t)stream);
}

int smoltts_k_attention_kv(const float* q_dev, const void* k_cache_dev, const void* v_cache_dev, const int32_t* row_pos_dev,
                           const int32_t* row_slot_dev, int32_t n_rows, int32_t n_q_heads, int32_t n_kv_heads,
                           int32_t cache_len, int32_t window, float* out_dev, void* out_x3_dev, int32_t kv_format, void* stream) {
  return launch_attention(q_dev, k_cache_dev, v_cache_dev, row_pos_dev, row_slot_dev, n_rows, n_q_heads, n_kv_heads, cache_len,
                          window, out_dev, out_x3_dev, (hipStream_t)stream, kv_format);
}

int smoltts_k_attention_rows3(const float* q_dev, const void* k_cache3_dev, const void* v_cache3_dev, const int32_t* row_pos_dev,
                              const int32_t* row_slot_dev, int32_t n_rows, int32_t rows_per_slot, int32_t n_heads, int32_t cache_len,
                              int32_t window, float* out_dev, void* stream) {
  return launch_attention_rows3(q_dev, k_cache3_dev, v_cache3_dev, row_pos_dev, row_slot_dev, n_rows, rows_per_slot, n_heads, cache_len,
                                window, out_dev, (hipStream_t)stream);
}

int smoltts_k_attention_split(const float* q_dev, const void* k_cache_dev, const void* v_cache_dev, const int32_t* row_pos_dev,
                              const int32_t* row_slot_dev, int32_t n_rows, int32_t n_q_heads, int32_t n_kv_heads, int32_t cache_len,
                              int32_t window, float* out_dev, void* out_x3_dev, int32_t kv_format, float* split_part_dev,
                              int32_t* split_ticket_dev, void* stream) {
  ST_REQUIRE(split_part_dev && split_ticket_dev, SMOLTTS_E_INVALID, "k_attention_split: null scratch");
  return launch_attention(q_dev, k_cache_dev, v_cache_dev, row_pos_dev, row_slot_dev, n_rows, n_q_heads, n_kv_heads, cache_len,
                          window, out_dev, out_x3_dev, (hipStream_t)stream, kv_format, -1, split_part_dev, split_ticket_dev);
}

int smoltts_k_embed(const int32_t* cols_dev, int32_t n_rows, int32_t n_code_rows, const void* text_emb_dev,
                    const void* cb_emb_dev, int32_t dim, int32_t codebook_size, int32_t cb_first_offset, int32_t mask_mode,
                    int32_t sem_start, int32_t sem_end, float* x_dev, void* stream) {
  // table heights are unknown here: the test entry point trusts its caller's indices
  return launch_embed(cols_dev, n_rows, n_code_rows, text_emb_dev, cb_emb_dev, dim, codebook_size, cb_first_offset, mask_mode,
                      sem_start, sem_end, 0x7fffffff, 0x7fffffff, x_dev, nullptr, (hipStream_t)stream);
}

int smoltts_k_argmax(const float* logits_dev, int32_t n_rows, int32_t n_cols, int64_t ld, int32_t* ids_dev, int32_t ids_stride,
                     float* margin_dev, void* stream) {
  return launch_argmax(logits_dev, n_rows, n_cols, ld, ids_dev, ids_stride, margin_dev, nullptr, nullptr, 0, 0, nullptr, nullptr,
                       nullptr, (hipStream_t)stream);
}

int smoltts_k_sample(const float* logits_dev, int32_t n_rows, int32_t n_cols, int64_t ld, float temp, float min_p, uint64_t seed,
                     int32_t frame_base, int32_t step, int32_t* ids_dev, void* stream) {
  const SampleArgs sa{temp, min_p, seed, step, frame_base, nullptr, nullptr, nullptr};
  return launch_argmax(logits_dev, n_rows, n_cols, ld, ids_dev, 1, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, &sa,
                       (hipStream_t)stream);
}

int smoltts_k_layernorm(const float* x_dev, const float* w_dev, const float* b_dev, int32_t n_rows, int32_t dim, float eps,
                        float* out_dev, void* stream) {
  return launch_layernorm(x_dev, w_dev, b_dev, n_rows, dim, eps, out_dev, (hipStream_t)stream);
}

}  // extern "C"
