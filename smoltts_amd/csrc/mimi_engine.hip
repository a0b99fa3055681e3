// Mimi decoder engine: codes -> 24 kHz PCM, chunk-streaming (state carried across calls).
//
// Reference chain MimiModel._decode_frame (mlx_inference/src/smoltts_mlx/codec/mimi.py:73-104):
//   quantizer.decode (codec/rvq.py:179-186) -> upsample (codec/conv.py:232-282) ->
//   decoder_transformer (codec/transformer.py:134-150) -> SEANet decoder (codec/seanet.py:99-161).
// All activations are channel-last fp32 [slot][time][channels]; every Linear / Conv1d /
// ConvTranspose1d runs on the fp32 matrix-core GEMM (gemm.hip) with the causal padding expressed as
// halo rows in front of each conv input buffer: a stride-1 conv of k taps reads the k consecutive
// rows ending at t as one K = k*Cin contraction, a ConvTranspose1d (k = 2*stride) reads rows
// (t-1, t) and produces its `stride` output rows at once (N = stride*Cout), which lands directly
// in channel-last order.  After a chunk the last halo rows are shifted to the front, so decoding
// F frames in chunks == decoding them at once == MimiModel.decode.  (The reference's own
// decode_step re-runs the upsample statelessly, mimi.py:77, and therefore differs from its batch
// decode; this engine carries the upsample tap overlap instead.)
#include <stdlib.h>
#include <string.h>

#include <new>

#include "mimi_common.h"

using namespace smoltts;

namespace {
constexpr int D = MIMI_D, HEADS = MIMI_HEADS, FF = MIMI_FF, NCONV = 14, NBUF = 14;
constexpr int SAMPLES = 1920;
// buffer i feeds conv i (i < 13); channels, halo rows and rows per frame of each conv input
// conv order: conv0 | convT1 res1.c3 res1.c1 | convT2 ... | convT4 res4.c3 res4.c1 | final
constexpr int BUF_C[NBUF] = {512, 1024, 512, 256, 512, 256, 128, 256, 128, 64, 128, 64, 32, 64};
// The resnet blocks take the RAW ConvTranspose output (buffers 2, 5, 8): the residual needs it and the block's first conv applies
// ELU on the way in.  Stage 3 runs its block fused (seanet.hip: the hidden, buffer 9, never exists).  Stage 4 is one kernel from
// its ConvTranspose to the PCM samples (seanet_last.hip, convs 10 .. 13): it reads buffer 10 with 2 halo rows (one for the
// ConvTranspose's second tap + one because a tile recomputes the 4 rows in front of it: 2 for the block's k3 conv + 2 for the
// output conv); buffers 11 .. 13 never exist.
constexpr int BUF_HALO[NBUF] = {6, 1, 2, 0, 1, 2, 0, 1, 2, 0, 2, 0, 0, 0};
constexpr bool FUSED_BLOCK[NCONV] = {false, false, false, false, false, false, false, false, true, false, false, false, false, false};
constexpr int LAST_STAGE = 10;  // first conv of the stage that runs as one kernel
constexpr bool buf_used(int i) { return i <= LAST_STAGE && !(i >= 1 && FUSED_BLOCK[i - 1]); }
constexpr int BUF_RPF[NBUF] = {2, 2, 16, 16, 16, 96, 96, 96, 480, 480, 480, 1920, 1920, 1920};
// buffer index feeding each conv (conv c reads BUF[c]; writes BUF[c+1], the last writes pcm)
}  // namespace

struct SmolttsMimi {
  SmolttsMimiConfig cfg;
  SmolttsMimiWeights w;
  const char* arena;
  size_t arena_bytes;
};

struct SmolttsMimiSession {
  SmolttsMimi* m;
  int B, chunk;
  float* carry[2];   // [B][512] upsample carry (previous frame's RVQ embedding), double-buffered
  float *tx, *tn, *tq, *ta, *th;  // transformer rows [B*2*chunk][512|512|512|512|2048]
  float* tws;        // split-K partial sums of the fc2 GEMM: [4][B*2*chunk][512]
  float *kc, *vc;    // [n_layers][B][8][max_positions][64]
  char *kc3, *vc3;   // the same as bf16x3 pieces (sessions whose chunks reach 16 frames = 32 rows per slot; null otherwise), zeroed at creation
  size_t kv3_layer;  // bytes per layer
  int *row_pos, *row_slot;  // [B*2*chunk]
  float* buf[NBUF];
  size_t buf_bstride[NBUF];  // floats per slot
  size_t halo_total;         // bytes of everything that reset must zero: tracked via pointers below
  char* zero_begin;
  size_t zero_bytes;
  int* pos_dev;      // [B] transformer positions consumed so far, per slot (slots are reset independently when streams
                     // of different utterances share the session)
  float final_bias;  // bias of the output conv (read from the arena once, at session creation)
  int* pos_host;     // host mirror (deterministic: += 2 * frames per call for the slots decoded, 0 on reset)
  int parity;
  int products;            // SMOLTTS_MIMI_OPT_PRODUCTS: 3 or 6 (0 = 6) bf16x3 products per operand pair in the matrix-core kernels
  int stateless_upsample;  // SMOLTTS_MIMI_OPT_STATELESS_UPSAMPLE: every call up-samples its frames with no carry (mimi.py:77)
};

namespace {

size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct Carver {
  char* base;
  size_t off;
  template <typename T>
  T* take(size_t n) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off = align_up(off + n * sizeof(T));
    return p;
  }
};

void carve(SmolttsMimiSession* s, char* base, size_t* total) {
  Carver cv{base, 0};
  const size_t B = s->B, F = s->chunk, R = B * 2 * F;
  const SmolttsMimiConfig& c = s->m->cfg;
  // --- state that smoltts_mimi_reset zeroes (contiguous)
  const size_t z0 = cv.off;
  s->carry[0] = cv.take<float>(B * D);
  s->carry[1] = cv.take<float>(B * D);
  s->pos_dev = cv.take<int>(B);
  for (int i = 0; i < NBUF; ++i) {
    s->buf_bstride[i] = buf_used(i) ? (size_t)(BUF_HALO[i] + BUF_RPF[i] * F) * BUF_C[i] : 0;
    s->buf[i] = cv.take<float>(B * s->buf_bstride[i]);
  }
  s->zero_begin = base ? base + z0 : nullptr;
  s->zero_bytes = cv.off - z0;
  // --- scratch / caches (need no zeroing: only positions < `positions` are ever read)
  s->tx = cv.take<float>(R * D);
  s->tn = cv.take<float>(R * D);
  s->tq = cv.take<float>(R * D);
  s->ta = cv.take<float>(R * D);
  s->th = cv.take<float>(R * FF);
  s->tws = cv.take<float>(4 * R * D);
  const size_t kv = (size_t)c.n_layers * B * HEADS * c.max_positions * 64;
  s->kc = cv.take<float>(kv);
  s->vc = cv.take<float>(kv);
  s->row_pos = cv.take<int>(R);
  s->row_slot = cv.take<int>(R);
  // piece caches for the many-row attention (attn_rows3_kernel): only where a chunk can have 32 rows per slot
  s->kv3_layer = F >= 16 ? SMOLTTS_KV3_BYTES(B, HEADS, c.max_positions) : 0;
  s->kc3 = s->kv3_layer ? cv.take<char>(s->kv3_layer * c.n_layers) : nullptr;
  s->vc3 = s->kv3_layer ? cv.take<char>(s->kv3_layer * c.n_layers) : nullptr;
  *total = cv.off;
}

// e[b][f] = sum_q table[q][code[b][f][q]]  (rvq.py:118-131,179-186 with output_proj folded in),
// then the depthwise ConvTranspose1d k=4 s=2 (conv.py:271-282):
// out[2f + r][c] = e[f][c] * w[r][c] + e[f-1][c] * w[r+2][c], e[-1] = carry.
__global__ __launch_bounds__(128) void rvq_upsample_kernel(const int* codes, long codes_stride, int frame_stride, int code_offset,
                                                           int nq, int n_frames, const float* table, const float* upw,
                                                           const float* carry_in, float* carry_out, float* tx) {
  const int f = blockIdx.x, b = blockIdx.y, c = threadIdx.x * 4;
  auto embed = [&](int ff) {
    const int* cd = codes + (long)b * codes_stride + (long)ff * frame_stride + code_offset;
    float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = 0; q < nq; ++q) {
      int code = cd[q];
      code = code < 0 ? 0 : (code > 2047 ? 2047 : code);
      const float4 t = *reinterpret_cast<const float4*>(table + ((long)q * 2048 + code) * D + c);
      e.x += t.x; e.y += t.y; e.z += t.z; e.w += t.w;
    }
    return e;
  };
  const float4 cur = embed(f);
  // carry_in == nullptr: the call's first frame has no predecessor (the reference's per-call up-sampling, mimi.py:77)
  const float4 prev = f > 0 ? embed(f - 1) : carry_in ? *reinterpret_cast<const float4*>(carry_in + (long)b * D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 w0 = *reinterpret_cast<const float4*>(upw + 0 * D + c), w1 = *reinterpret_cast<const float4*>(upw + 1 * D + c);
  const float4 w2 = *reinterpret_cast<const float4*>(upw + 2 * D + c), w3 = *reinterpret_cast<const float4*>(upw + 3 * D + c);
  float* o = tx + ((long)b * 2 * n_frames + 2 * f) * D + c;
  *reinterpret_cast<float4*>(o) = make_float4(cur.x * w0.x + prev.x * w2.x, cur.y * w0.y + prev.y * w2.y,
                                              cur.z * w0.z + prev.z * w2.z, cur.w * w0.w + prev.w * w2.w);
  *reinterpret_cast<float4*>(o + D) = make_float4(cur.x * w1.x + prev.x * w3.x, cur.y * w1.y + prev.y * w3.y,
                                                  cur.z * w1.z + prev.z * w3.z, cur.w * w1.w + prev.w * w3.w);
  if (f == n_frames - 1) *reinterpret_cast<float4*>(carry_out + (long)b * D + c) = cur;
}

__global__ void mimi_rows_kernel(int n_rows, int rows_per_slot, int pos0, const int* slot_pos, int* row_pos, int* row_slot) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n_rows) return;
  const int b = m / rows_per_slot;
  row_slot[m] = b;
  row_pos[m] = (slot_pos ? slot_pos[b] : pos0) + (m - b * rows_per_slot);
}

__global__ void mimi_advance_kernel(int batch, int by, int* slot_pos) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < batch) slot_pos[b] += by;
}

// Shift the last `halo` rows of every conv input buffer to its front (streaming carry).
struct HaloDesc {
  float* buf[NBUF];
  long bstride[NBUF];
  int C[NBUF], halo[NBUF], T[NBUF];
};
__global__ __launch_bounds__(256) void halo_shift_kernel(HaloDesc d) {
  extern __shared__ float sh[];
  const int b = blockIdx.x, i = blockIdx.y;
  const int n = d.halo[i] * d.C[i];
  if (n == 0) return;
  float* base = d.buf[i] + (long)b * d.bstride[i];
  const float* src = base + (long)d.T[i] * d.C[i];  // rows [T, T+halo) == the last halo rows of halo+data
  for (int j = threadIdx.x; j < n; j += 256) sh[j] = src[j];
  __syncthreads();
  for (int j = threadIdx.x; j < n; j += 256) base[j] = sh[j];
}

}  // namespace

namespace smoltts {

int launch_mimi_rows(int n_rows, int rows_per_slot, int pos0, int* row_pos, int* row_slot, hipStream_t st, const int* slot_pos) {
  hipLaunchKernelGGL(mimi_rows_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, st, n_rows, rows_per_slot, pos0, slot_pos, row_pos, row_slot);
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

int run_mimi_transformer(const char* A, const SmolttsMimiLayerWeights* layers, int n_layers, const float* rope, int cache_len,
                         int window, const MimiTransformerBufs& b, int R, int Tt, float* last_out, int64_t last_bstride,
                         hipStream_t st) {
  for (int l = 0; l < n_layers; ++l) {
    const SmolttsMimiLayerWeights& lw = layers[l];
    float* kc = b.kc + l * b.layer_stride;
    float* vc = b.vc + l * b.layer_stride;
    {  // LayerNorm (as the GEMM's prologue where its kernel has one, else through b.tn) + QKV + RoPE + cache write
      SmolttsGemmArgs a = mimi_gemm_f32(A + lw.wqkv, b.tx, D, R, 3 * D, D, lw.wqkv3 ? A + lw.wqkv3 : nullptr);
      a.prologue = SMOLTTS_PRO_LAYERNORM; a.gamma_dev = (const float*)(A + lw.ln1_w); a.beta_dev = (const float*)(A + lw.ln1_b);
      a.eps = 1e-5f; a.ln_scratch_dev = b.tn;
      a.epilogue = SMOLTTS_EPI_QKV_ROPE; a.out_dev = b.tq; a.ldo = D;
      a.rope_dev = rope; a.row_pos_dev = b.row_pos; a.row_slot_dev = b.row_slot;
      a.k_cache_dev = kc; a.v_cache_dev = vc;
      if (b.kc3) { a.k_cache3_dev = b.kc3 + l * b.layer_stride3; a.v_cache3_dev = b.vc3 + l * b.layer_stride3; }
      a.n_q_heads = HEADS; a.n_kv_heads = HEADS; a.cache_len = cache_len;
      a.b3_products = b.b3_products;
      ST_TRY(launch_gemm(a, st));
    }
    static const bool rows3_off = ST_KNOB_INT("SMOLTTS_ROWS3", 1) == 0;  // experiments (knobs builds only)
    if (b.kc3 && Tt % 32 == 0 && !rows3_off)  // whole 32-row groups per slot: the bf16x3 kernel over the piece caches
      ST_TRY(launch_attention_rows3(b.tq, b.kc3 + l * b.layer_stride3, b.vc3 + l * b.layer_stride3, b.row_pos, b.row_slot, R, Tt, HEADS,
                                    cache_len, window, b.ta, st, b.b3_products));
    else
      ST_TRY(launch_attention(b.tq, kc, vc, b.row_pos, b.row_slot, R, HEADS, HEADS, cache_len, window, b.ta, nullptr, st));
    {
      SmolttsGemmArgs a = mimi_gemm_f32(A + lw.wo, b.ta, D, R, D, D, lw.wo3 ? A + lw.wo3 : nullptr);
      a.epilogue = SMOLTTS_EPI_SCALE_RESID; a.scale_dev = (const float*)(A + lw.ls1);
      a.resid_dev = b.tx; a.out_dev = b.tx; a.ldo = D;
      a.b3_products = b.b3_products;
      ST_TRY(launch_gemm(a, st));
    }
    {
      SmolttsGemmArgs a = mimi_gemm_f32(A + lw.fc1, b.tx, D, R, FF, D, lw.fc13 ? A + lw.fc13 : nullptr);
      a.prologue = SMOLTTS_PRO_LAYERNORM; a.gamma_dev = (const float*)(A + lw.ln2_w); a.beta_dev = (const float*)(A + lw.ln2_b);
      a.eps = 1e-5f; a.ln_scratch_dev = b.tn;
      a.epilogue = SMOLTTS_EPI_GELU; a.out_dev = b.th; a.ldo = FF;
      a.b3_products = b.b3_products;
      ST_TRY(launch_gemm(a, st));
    }
    {
      SmolttsGemmArgs a = mimi_gemm_f32(A + lw.fc2, b.th, FF, R, D, FF, lw.fc23 ? A + lw.fc23 : nullptr);
      a.splitk_ws_dev = b.tws; a.splitk_ws_floats = b.tws ? (int64_t)4 * R * D : 0;
      a.epilogue = SMOLTTS_EPI_SCALE_RESID; a.scale_dev = (const float*)(A + lw.ls2);
      a.resid_dev = b.tx; a.ldr = D; a.r_bstride = (int64_t)Tt * D; a.rows_per_batch = Tt;
      a.x_bstride = (int64_t)Tt * FF;
      if (l + 1 < n_layers) {
        a.out_dev = b.tx; a.ldo = D; a.o_bstride = (int64_t)Tt * D;
      } else {
        a.out_dev = last_out; a.ldo = D; a.o_bstride = last_bstride;
      }
      a.b3_products = b.b3_products;
      ST_TRY(launch_gemm(a, st));
    }
  }
  return SMOLTTS_OK;
}

}  // namespace smoltts

extern "C" {

int smoltts_mimi_create(const SmolttsMimiConfig* cfg, const SmolttsMimiWeights* offsets, const void* arena_dev,
                        size_t arena_bytes, SmolttsMimi** out) {
  ST_REQUIRE(cfg && offsets && arena_dev && out, SMOLTTS_E_INVALID, "mimi_create: null argument");
  ST_REQUIRE(cfg->num_codebooks >= 1 && cfg->num_codebooks <= 32 && cfg->n_layers >= 1 && cfg->n_layers <= SMOLTTS_MIMI_MAX_LAYERS &&
                 cfg->max_positions >= 2 && cfg->window >= 0,
             SMOLTTS_E_INVALID, "mimi_create: bad config");
  for (int i = 0; i < NCONV; ++i) {
    const SmolttsMimiConv& cv = offsets->convs[i];
    ST_REQUIRE(cv.cin == BUF_C[i] && (i + 1 == NCONV ? cv.cout == 1 : (cv.transposed ? cv.cout : cv.cout) == BUF_C[i + 1]) &&
                   (cv.transposed ? (cv.k == 2 * cv.stride && BUF_HALO[i] >= 1) : (cv.stride == 1 && (i > LAST_STAGE || BUF_HALO[i] == cv.k - 1))) && cv.w % 16 == 0 && cv.b % 16 == 0 && cv.w < arena_bytes && cv.b < arena_bytes,
               SMOLTTS_E_INVALID, "mimi_create: conv %d descriptor inconsistent (cin=%d)", i, cv.cin);
  }
  ST_REQUIRE(offsets->rvq_table + (size_t)cfg->num_codebooks * 2048 * D * 4 <= arena_bytes &&
                 offsets->rope + (size_t)cfg->max_positions * 64 * 4 <= arena_bytes,
             SMOLTTS_E_INVALID, "mimi_create: table offsets outside the arena");
  SmolttsMimi* m = new (std::nothrow) SmolttsMimi;
  ST_REQUIRE(m, SMOLTTS_E_INVALID, "mimi_create: out of host memory");
  m->cfg = *cfg; m->w = *offsets; m->arena = (const char*)arena_dev; m->arena_bytes = arena_bytes;
  *out = m;
  return SMOLTTS_OK;
}

void smoltts_mimi_destroy(SmolttsMimi* m) { delete m; }

size_t smoltts_mimi_slab_bytes(const SmolttsMimi* m, int32_t max_batch, int32_t max_chunk_frames) {
  if (!m || max_batch <= 0 || max_chunk_frames <= 0) return 0;
  SmolttsMimiSession tmp;
  memset(&tmp, 0, sizeof(tmp));
  tmp.m = const_cast<SmolttsMimi*>(m);
  tmp.B = max_batch; tmp.chunk = max_chunk_frames;
  size_t total = 0;
  carve(&tmp, nullptr, &total);
  return total;
}

int smoltts_mimi_session_create(SmolttsMimi* m, void* slab_dev, size_t slab_bytes, int32_t max_batch,
                                int32_t max_chunk_frames, SmolttsMimiSession** out) {
  ST_REQUIRE(m && slab_dev && out && max_batch > 0 && max_chunk_frames > 0, SMOLTTS_E_INVALID, "mimi_session_create: bad argument");
  ST_REQUIRE(((uintptr_t)slab_dev & 255) == 0, SMOLTTS_E_INVALID, "mimi_session_create: slab must be 256-byte aligned");
  ST_REQUIRE(2 * max_chunk_frames <= m->cfg.max_positions, SMOLTTS_E_CAPACITY, "mimi_session_create: chunk exceeds max_positions");
  const size_t need = smoltts_mimi_slab_bytes(m, max_batch, max_chunk_frames);
  ST_REQUIRE(slab_bytes >= need, SMOLTTS_E_CAPACITY, "mimi_session_create: slab has %zu bytes, %zu needed", slab_bytes, need);
  SmolttsMimiSession* s = new (std::nothrow) SmolttsMimiSession;
  ST_REQUIRE(s, SMOLTTS_E_INVALID, "mimi_session_create: out of host memory");
  memset(s, 0, sizeof(*s));
  s->m = m; s->B = max_batch; s->chunk = max_chunk_frames;
  size_t total = 0;
  carve(s, (char*)slab_dev, &total);
  s->pos_host = static_cast<int*>(calloc((size_t)max_batch, sizeof(int)));
  // (the piece caches are read in whole 32-position blocks: positions not written yet get zero weights and must hold finite values)
  if (s->pos_host == nullptr || hipMemset(s->zero_begin, 0, s->zero_bytes) != hipSuccess ||
      (s->kc3 && (hipMemset(s->kc3, 0, s->kv3_layer * m->cfg.n_layers) != hipSuccess || hipMemset(s->vc3, 0, s->kv3_layer * m->cfg.n_layers) != hipSuccess)) ||
      hipMemcpy(&s->final_bias, m->arena + m->w.convs[NCONV - 1].b, sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
    free(s->pos_host);
    delete s;
    set_error("mimi_session_create: allocation or hipMemset failed");
    return SMOLTTS_E_HIP;
  }
  *out = s;
  return SMOLTTS_OK;
}

void smoltts_mimi_session_destroy(SmolttsMimiSession* s) {
  if (!s) return;
  free(s->pos_host);
  delete s;
}

int smoltts_mimi_reset(SmolttsMimiSession* s, void* stream) {
  ST_REQUIRE(s, SMOLTTS_E_INVALID, "mimi_reset: null session");
  ST_CHECK_HIP(hipMemsetAsync(s->zero_begin, 0, s->zero_bytes, (hipStream_t)stream));
  for (int b = 0; b < s->B; ++b) s->pos_host[b] = 0;
  s->parity = 0;
  return SMOLTTS_OK;
}

int smoltts_mimi_session_set_option(SmolttsMimiSession* s, int32_t option, int32_t value) {
  ST_REQUIRE(s, SMOLTTS_E_INVALID, "mimi_session_set_option: null session");
  if (option == SMOLTTS_MIMI_OPT_PRODUCTS) {
    ST_REQUIRE(value == 3 || value == 6, SMOLTTS_E_INVALID, "mimi_session_set_option: products %d (3 or 6)", value);
    s->products = value;
    return SMOLTTS_OK;
  }
  ST_REQUIRE(option == SMOLTTS_MIMI_OPT_STATELESS_UPSAMPLE, SMOLTTS_E_INVALID, "mimi_session_set_option: unknown option %d", option);
  ST_REQUIRE(value == 0 || value == 1, SMOLTTS_E_INVALID, "mimi_session_set_option: value %d (0 or 1)", value);
  s->stateless_upsample = value;
  return SMOLTTS_OK;
}

int smoltts_mimi_reset_slots(SmolttsMimiSession* s, const int32_t* slots_host, int32_t n_slots, void* stream) {
  ST_REQUIRE(s && slots_host && n_slots > 0, SMOLTTS_E_INVALID, "mimi_reset_slots: bad argument");
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < n_slots; ++i) {
    const int b = slots_host[i];
    ST_REQUIRE(b >= 0 && b < s->B, SMOLTTS_E_INVALID, "mimi_reset_slots: slot %d out of range", b);
    ST_CHECK_HIP(hipMemsetAsync(s->carry[0] + (size_t)b * D, 0, D * sizeof(float), st));
    ST_CHECK_HIP(hipMemsetAsync(s->carry[1] + (size_t)b * D, 0, D * sizeof(float), st));
    ST_CHECK_HIP(hipMemsetAsync(s->pos_dev + b, 0, sizeof(int), st));
    for (int j = 0; j < NBUF; ++j)  // only the halo rows carry state from chunk to chunk
      if (BUF_HALO[j] > 0 && buf_used(j))
        ST_CHECK_HIP(hipMemsetAsync(s->buf[j] + (size_t)b * s->buf_bstride[j], 0, (size_t)BUF_HALO[j] * BUF_C[j] * sizeof(float), st));
    s->pos_host[b] = 0;
  }
  return SMOLTTS_OK;
}

static int decode_chunk_impl(SmolttsMimiSession* s, const int32_t* codes_dev, int64_t codes_stride, int32_t frame_stride,
                             int32_t code_offset, int32_t batch, int32_t n_frames, float* pcm_dev, int64_t pcm_stride,
                             void* stream);

int smoltts_mimi_decode_chunk(SmolttsMimiSession* s, const int32_t* codes_dev, int64_t codes_stride, int32_t frame_stride,
                              int32_t code_offset, int32_t batch, int32_t n_frames, float* pcm_dev, int64_t pcm_stride,
                              void* stream) {
  ST_REQUIRE(s && codes_dev && pcm_dev, SMOLTTS_E_INVALID, "mimi_decode_chunk: null argument");
  ST_REQUIRE(batch > 0 && n_frames > 0, SMOLTTS_E_INVALID, "mimi_decode_chunk: empty batch or chunk");
  // the last convolutions run batch * frames * 1920 rows in one launch (<= 65535 * 64): longer chunks are
  // decoded in pieces, which the streaming state makes equivalent
  const long per_frame = (long)batch * SAMPLES;
  const int piece = (int)(4000000L / per_frame) > 0 ? (int)(4000000L / per_frame) : 1;
  for (int f0 = 0; f0 < n_frames; f0 += piece) {
    const int n = n_frames - f0 < piece ? n_frames - f0 : piece;
    ST_TRY(decode_chunk_impl(s, codes_dev + (int64_t)f0 * frame_stride, codes_stride, frame_stride, code_offset, batch, n,
                             pcm_dev + (int64_t)f0 * SAMPLES, pcm_stride, stream));
  }
  return SMOLTTS_OK;
}

}  // extern "C"

static int decode_chunk_impl(SmolttsMimiSession* s, const int32_t* codes_dev, int64_t codes_stride, int32_t frame_stride,
                             int32_t code_offset, int32_t batch, int32_t n_frames, float* pcm_dev, int64_t pcm_stride,
                             void* stream) {
  ST_REQUIRE(batch > 0 && batch <= s->B && n_frames > 0 && n_frames <= s->chunk, SMOLTTS_E_CAPACITY,
             "mimi_decode_chunk: batch=%d frames=%d exceed the session (%d, %d)", batch, n_frames, s->B, s->chunk);
  const SmolttsMimi* m = s->m;
  const SmolttsMimiConfig& c = m->cfg;
  for (int b = 0; b < batch; ++b)
    ST_REQUIRE(s->pos_host[b] + 2 * n_frames <= c.max_positions, SMOLTTS_E_CAPACITY,
               "mimi_decode_chunk: slot %d: %d positions exceed max_positions=%d", b, s->pos_host[b] + 2 * n_frames, c.max_positions);
  ST_REQUIRE(frame_stride >= c.num_codebooks + code_offset && code_offset >= 0 && pcm_stride >= (int64_t)SAMPLES * n_frames &&
                 pcm_stride % 4 == 0 && ((uintptr_t)pcm_dev & 15) == 0,
             SMOLTTS_E_INVALID, "mimi_decode_chunk: bad strides");
  hipStream_t st = (hipStream_t)stream;
  const char* A = m->arena;
  const int F = n_frames, Tt = 2 * F, R = batch * Tt;

  // 1. RVQ gather (+ folded output_proj) and depthwise upsample -> transformer rows tx[b][2F][512]
  hipLaunchKernelGGL(rvq_upsample_kernel, dim3(F, batch), dim3(128), 0, st, codes_dev, (long)codes_stride, frame_stride, code_offset,
                     c.num_codebooks, F, (const float*)(A + m->w.rvq_table), (const float*)(A + m->w.upsample_w),
                     s->stateless_upsample ? (const float*)nullptr : (const float*)s->carry[s->parity], s->carry[s->parity ^ 1], s->tx);
  ST_CHECK_HIP(hipGetLastError());
  ST_TRY(launch_mimi_rows(R, Tt, 0, s->row_pos, s->row_slot, st, s->pos_dev));

  // 2. decoder transformer (transformer.py:109-131); the last layer writes straight into conv0's input
  //    buffer, behind its halo
  {
    MimiTransformerBufs tb{s->tx, s->tn, s->tq, s->ta, s->th, s->kc, s->vc, (size_t)s->B * HEADS * c.max_positions * 64,
                           s->row_pos, s->row_slot};
    tb.tws = s->tws;
    tb.kc3 = s->kc3; tb.vc3 = s->vc3; tb.layer_stride3 = s->kv3_layer; tb.b3_products = s->products == 3 ? 3 : 6;
    ST_TRY(run_mimi_transformer(A, m->w.layers, c.n_layers, (const float*)(A + m->w.rope), c.max_positions, c.window, tb, R, Tt,
                                s->buf[0] + (size_t)BUF_HALO[0] * BUF_C[0], (int64_t)s->buf_bstride[0], st));
  }

  // 3. SEANet decoder (seanet.py:105-139): GEMMs over halo-prefixed channel-last buffers + the fused kernels of stages 3, 4.
  //    A buffer holds ELU(activation) (each consumer applies ELU first, seanet.py:16-20,117-137: computed once, by the
  //    producer) except the ConvTranspose outputs, which stay raw for their resnet block.
  for (int i = 0; i < NCONV; ++i) {
    const SmolttsMimiConv& cv = m->w.convs[i];
    if (i == LAST_STAGE) {  // ConvTranspose i, resnet block i+1, i+2 and the output conv i+3 in one kernel
      const SmolttsMimiConv &c2 = m->w.convs[i + 1], &c3 = m->w.convs[i + 2];
      ST_REQUIRE(cv.w3 && c2.w3 && c3.w3 && m->w.final_w && cv.cin == 128 && cv.cout == 64 && cv.stride == 4, SMOLTTS_E_INVALID,
                 "mimi: the arena lacks the W3 tiles of the last stage (or its shape is not 128 -> 64, stride 4)");
      MimiLastStageArgs a;
      memset(&a, 0, sizeof(a));
      a.batch = batch; a.T = BUF_RPF[i] * F;
      a.in = s->buf[i] + (size_t)BUF_HALO[i] * BUF_C[i]; a.in_bstride = (int64_t)s->buf_bstride[i];
      a.wt = A + cv.w3; a.bt = (const float*)(A + cv.b);
      a.w2 = A + c2.w3; a.b2 = (const float*)(A + c2.b); a.w3 = A + c3.w3; a.b3 = (const float*)(A + c3.b);
      a.final_w = (const float*)(A + m->w.final_w); a.final_b = s->final_bias; a.pcm = pcm_dev; a.pcm_stride = pcm_stride;
      a.slot_pos = s->pos_dev; a.b3_products = s->products;
      ST_TRY(launch_seanet_last(a, st));
      break;
    }
    if (FUSED_BLOCK[i]) {  // resnet block i, i+1 in one kernel
      const SmolttsMimiConv& c1 = m->w.convs[i + 1];
      ST_REQUIRE(cv.w3 && c1.w3, SMOLTTS_E_INVALID, "mimi: the arena lacks the W3 tiles of conv %d", i);
      MimiResblockArgs a;
      memset(&a, 0, sizeof(a));
      a.channels = cv.cin; a.batch = batch; a.T = BUF_RPF[i] * F;
      a.x = s->buf[i] + (size_t)BUF_HALO[i] * BUF_C[i]; a.x_bstride = (int64_t)s->buf_bstride[i];
      a.w2 = A + cv.w3; a.b2 = (const float*)(A + cv.b); a.w3 = A + c1.w3; a.b3 = (const float*)(A + c1.b);
      a.out = s->buf[i + 2] + (size_t)BUF_HALO[i + 2] * BUF_C[i + 2]; a.o_bstride = (int64_t)s->buf_bstride[i + 2];
      a.b3_products = s->products;
      ST_TRY(launch_seanet_resblock(a, st));
      i += 1;
      continue;
    }
    const int Tin = BUF_RPF[i] * F;
    const int K = cv.transposed ? 2 * cv.cin : cv.k * cv.cin;
    const int N = cv.transposed ? cv.stride * cv.cout : cv.cout;
    // the GEMM window of row t starts k - 1 (ConvTranspose: 1) rows before it; a buffer may carry more halo rows than that
    const int lead = BUF_HALO[i] - (cv.transposed ? 1 : cv.k - 1);
    SmolttsGemmArgs a = mimi_gemm_f32(A + cv.w, s->buf[i] + (size_t)lead * cv.cin, cv.cin, batch * Tin, N, K, cv.w3 ? A + cv.w3 : nullptr);
    a.rows_per_batch = Tin; a.x_bstride = (int64_t)s->buf_bstride[i];
    a.bias_dev = (const float*)(A + cv.b);
    a.prologue = SMOLTTS_PRO_NONE;
    a.epilogue = SMOLTTS_EPI_STORE;
    if (i + 1 < NCONV) {
      a.out_dev = s->buf[i + 1] + (size_t)BUF_HALO[i + 1] * BUF_C[i + 1];
      a.ldo = N; a.o_bstride = (int64_t)s->buf_bstride[i + 1];
      a.elu_out = cv.transposed ? 0 : 1;  // a resnet block takes the raw tensor: its residual needs it, its first conv applies ELU
    } else {
      a.out_dev = pcm_dev; a.ldo = 1; a.o_bstride = pcm_stride;
    }
    const bool res_c3 = i >= 2 && i <= 11 && (i % 3) == 2;  // first conv of a residual block: ELU on the way in
    if (res_c3) a.prologue = SMOLTTS_PRO_ELU;
    const bool res_c1 = i >= 3 && i <= 12 && (i % 3) == 0;  // second conv of a residual block: + block input (two buffers back)
    if (res_c1) {
      a.epilogue = SMOLTTS_EPI_RESID;
      a.resid_dev = s->buf[i - 1] + (size_t)BUF_HALO[i - 1] * BUF_C[i - 1];
      a.ldr = BUF_C[i - 1]; a.r_bstride = (int64_t)s->buf_bstride[i - 1];
    }
    a.b3_products = s->products;
    ST_TRY(launch_gemm(a, st));
  }

  // 4. streaming carry: last halo rows of every conv input buffer move to the front
  {
    HaloDesc d;
    for (int i = 0; i < NBUF; ++i) {
      d.buf[i] = s->buf[i]; d.bstride[i] = (long)s->buf_bstride[i];
      d.C[i] = BUF_C[i]; d.halo[i] = buf_used(i) ? BUF_HALO[i] : 0; d.T[i] = BUF_RPF[i] * F;
    }
    hipLaunchKernelGGL(halo_shift_kernel, dim3(batch, NBUF), dim3(256), 6 * 512 * sizeof(float), st, d);
    ST_CHECK_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(mimi_advance_kernel, dim3((batch + 63) / 64), dim3(64), 0, st, batch, Tt, s->pos_dev);
  ST_CHECK_HIP(hipGetLastError());
  for (int b = 0; b < batch; ++b) s->pos_host[b] += Tt;
  s->parity ^= 1;
  return SMOLTTS_OK;
}
