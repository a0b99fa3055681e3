// DualAR engine: orchestrates the kernels into prefill / frame-decode launch sequences.
//
// One frame == one SingleBatchGenerator.__next__ of the reference (lm/generate.py:59-171) for every
// slot of the session at once:  slow transformer step (forward_generate, lm/rq_transformer.py:173-192)
// -> argmax -> n_fast depth-transformer steps with a per-frame KV cache (forward_generate_fast :194-220,
// generate.py:110-141) -> commit of the (1 + n_fast)-high column.  Sampling, the column feedback, the
// stop rule and the position counters all live on the device, so a frame is a fixed launch sequence that
// is captured once into a hipGraph and replayed (the reference synchronises with the host 9x per frame).
//
// Dataflow between kernels (x3.h): the residual stream x stays fp32; whoever produces x also publishes
// it as the bf16x3 operand of the next GEMM (times that GEMM's RMSNorm weight) plus partial sums of
// squares, so each transformer block is exactly five launches:
//   wqkv GEMM (+RMSNorm scale, RoPE, KV write) -> attention (writes wo's operand) -> wo GEMM (+residual,
//   publishes w1|w3's operand) -> w1|w3 GEMM (+RMSNorm scale, SwiGLU, writes w2's operand) -> w2 GEMM
//   (+residual, publishes the next block's / head's operand).
#include <stdlib.h>
#include <string.h>

#include <new>

#include "argmax_dev.h"
#include "embed_dev.h"

using namespace smoltts;

struct SmolttsEngine {
  SmolttsLMConfig cfg;
  SmolttsLMWeights w;
  const char* arena;
  size_t arena_bytes;
  // derived table (smoltts_engine_build_fast_qkv; caller-owned memory): q | k | v of depth layer 0 for every row of the fast
  // embedding table, before RoPE -- [fast_emb_rows][(fast heads + 2 fast kv heads) * 64] fp32; nullptr = not built
  const float* fast_qkv;
};

constexpr int STAGE_RING = 8;

struct SmolttsSession {
  SmolttsEngine* e;
  int B, max_seq, max_rows, max_frames;
  int stop_on_eos;
  float temp, fast_temp, min_p;  // <= 0: greedy (lm/generate.py:88-99,118-132)
  uint64_t seed;
  // activations
  float *xr, *qr;            // prefill rows: fp32 residual stream [max_rows][dim], q [max_rows][Hq*64]
  float *xt, *xf;            // decode/tail rows: [B][dim], [B][fast_dim]
  float* qt;                 // [B][max(Hq, fast Hq)*64]
  char *x3n, *x3n2, *x3a, *x3h;  // X3 operands: normed stream (2 consumers), attention out, SwiGLU out
  float* ssq;                // [rows][dim/16] partial sums of squares of the stream
  // side workspace of smoltts_lm_prefill_side (prompt rows of NEW tenants beside the decode frames of the others): a second set
  // of everything the slow layers write that is not a KV-cache row
  float *sd_xr, *sd_qr;
  char *sd_x3n, *sd_x3a, *sd_x3h;
  float* sd_ssq;
  float* logits;             // [B][max(vocab, codebook)]: the depth heads' rows
  float* logits_slow;        // [B][vocab]: the slow head's rows (picked by the commit kernel at the end of the frame)
  // caches
  char *kc, *vc;             // [n_layer][B][KV][max_seq][64] in fp32 or bf16 (kv_format)
  int kv_format;
  float *fkc, *fvc;          // [n_fast_layer][B][FKV][n_fast][64]
  // integer state
  int *cur_col, *new_col;    // [B][1+n_fast]
  int *pos, *frames, *done, *mask, *iota, *fastpos;  // [B] each; fastpos [n_fast][B]
  int* salt;                   // [B] tenants a slot has had: mixed into the sampling seed
  int *stage_slots, *stage_last;                     // [B]
  float* margin;             // [B]
  float* attn_part;          // key-split slow attention: partial records (ATT_SPLIT_PART_FLOATS)
  int* attn_ticket;          // ... and arrival tickets [ATT_SPLIT_MAX_PAIRS]: zeroed at creation and again by every commit kernel (end of frame)
  int* margin_at;            // [B] frame * 64 + step of the slot's smallest top-2 gap
  int* codes;                // [B][max_frames][1+n_fast]
  // host staging (pinned)
  int* h_stage;                // STAGE_RING pinned areas of 2*B ints, used in turn
  hipEvent_t stage_ev[8];      // recorded behind the async copies out of each area
  bool stage_ev_live[8];
  int stage_next;
  hipGraphExec_t graph_exec;   // one decode frame (slow step + tail)
  bool graph_ready;
  hipGraphExec_t tail_exec;    // the tail alone (slow head + depth steps + commit) as run after a prefill
  bool tail_ready;
  hipGraphExec_t multi_exec;   // `multi_frames` consecutive decode frames in one graph (fewer graph launches per tick / chunk)
  bool multi_ready;
  int flight_limit;            // SMOLTTS_MAX_FRAMES_IN_FLIGHT as read at creation (0 = unbounded)
  int multi_frames;            // frames per multi-frame graph (1 = single-frame graphs only): SMOLTTS_FRAMES_PER_GRAPH, else
                               // smoltts_session_set_frames_per_graph, else min(n_frames, 4) of the largest decode call so far
  bool multi_fixed;            // set by the environment or the API: decode calls do not change it
  bool multi_env;              // set by SMOLTTS_FRAMES_PER_GRAPH: the API does not change it either
  bool prefilled;
  // bounded run-ahead of the host over the GPU (smoltts_lm_decode): an event every `flight_group` frame graphs, the host
  // waits for the one recorded two groups ago before it launches further
  hipEvent_t flight_ev[2];
  bool flight_live[2];
  int flight_cur, flight_count, flight_group;
  // measurement aid of THIS session (smoltts_session_measure_duplicate): launches of one kernel class are issued twice
  int dup_code, dup_n;
  bool use_qkv_table;          // depth layer-0 q | k | v from the engine's table where it exists (SMOLTTS_OPT_QKV_TABLE)
  bool commit_picks;           // slow token and last depth code picked inside the commit kernel (SMOLTTS_OPT_COMMIT_PICKS)
  bool split_attn;             // slow attention of few rows over two workgroups per (row, kv head) (SMOLTTS_OPT_SPLIT_ATTN)
  int stream_w;                // which weights of a decode frame are loaded with the non-temporal hint (SMOLTTS_OPT_STREAM_W, bit mask)
  bool fuse_depth_attn;        // depth attention worked out inside the wo launch (SMOLTTS_OPT_FUSE_DEPTH_ATTN)
  bool fp8_prefill;            // prompt rows (M >= 256) of an fp8 engine on the fp8 x fp8 MFMA (SMOLTTS_OPT_FP8_PREFILL; not the parity path)
  bool fuse_pick;              // greedy depth codes picked inside the next step's layer-0 attention + wo launch (SMOLTTS_OPT_FUSE_PICK)
  float* cand;                 // [B][codebook_size / 16][4]: the depth head GEMM's tile candidates for that pick
  float* cand_slow;            // [B][vocab_size / 16][4]: the slow head GEMM's, for the commit kernel's greedy pick
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

int imax(int a, int b) { return a > b ? a : b; }

void carve(SmolttsSession* s, char* base, size_t* total) {
  const SmolttsLMConfig& c = s->e->cfg;
  Carver cv{base, 0};
  const size_t B = s->B, R = s->max_rows, H = 1 + c.n_fast;
  const size_t R16 = (R + 15) / 16 * 16, B16 = (B + 15) / 16 * 16;
  const size_t dq = (size_t)imax(c.n_head, c.fast_n_head) * 64;
  const size_t dmax = imax(c.dim, c.fast_dim), imx = imax(c.inter, c.fast_inter);
  s->xr = cv.take<float>(R * c.dim);
  s->qr = cv.take<float>(R * c.n_head * 64);
  s->xt = cv.take<float>(B * c.dim);
  s->xf = cv.take<float>(B * c.fast_dim);
  s->qt = cv.take<float>(B * dq);
  s->x3n = cv.take<char>(R16 * dmax * 6);
  s->x3n2 = cv.take<char>(B16 * dmax * 6);
  s->x3a = cv.take<char>(R16 * dmax * 6);
  s->x3h = cv.take<char>(R16 * imx * 6);
  s->ssq = cv.take<float>(R16 * (dmax / 16));
  s->sd_xr = cv.take<float>(R * c.dim);
  s->sd_qr = cv.take<float>(R * c.n_head * 64);
  s->sd_x3n = cv.take<char>(R16 * (size_t)c.dim * 6);
  s->sd_x3a = cv.take<char>(R16 * (size_t)c.dim * 6);
  s->sd_x3h = cv.take<char>(R16 * (size_t)c.inter * 6);
  s->sd_ssq = cv.take<float>(R16 * (c.dim / 16));
  s->logits = cv.take<float>(B * (size_t)imax(c.vocab_size, c.codebook_size));
  s->logits_slow = cv.take<float>(B * (size_t)c.vocab_size);
  s->cand = cv.take<float>(B * (size_t)((c.codebook_size + 15) / 16) * 4);
  s->cand_slow = cv.take<float>(B * (size_t)((c.vocab_size + 15) / 16) * 4);
  const size_t kv = (size_t)c.n_layer * B * c.n_kv_head * s->max_seq * 64 * (s->kv_format == SMOLTTS_KV_BF16 ? 2 : 4);
  s->kc = cv.take<char>(kv);
  s->vc = cv.take<char>(kv);
  const size_t fkv = (size_t)c.n_fast_layer * B * c.fast_n_kv_head * c.n_fast * 64;
  s->fkc = cv.take<float>(fkv);
  s->fvc = cv.take<float>(fkv);
  s->cur_col = cv.take<int>(B * H);
  s->new_col = cv.take<int>(B * H);
  s->pos = cv.take<int>(B);
  s->frames = cv.take<int>(B);
  s->salt = cv.take<int>(B);
  s->done = cv.take<int>(B);
  s->mask = cv.take<int>(B);
  s->iota = cv.take<int>(B);
  s->fastpos = cv.take<int>((size_t)c.n_fast * B);
  s->stage_slots = cv.take<int>(B);
  s->stage_last = cv.take<int>(B);
  s->margin = cv.take<float>(B);
  s->attn_part = cv.take<float>(ATT_SPLIT_PART_FLOATS);
  s->attn_ticket = cv.take<int>(ATT_SPLIT_MAX_PAIRS);
  s->margin_at = cv.take<int>(B);
  s->codes = cv.take<int>(B * (size_t)s->max_frames * H);
  *total = cv.off;
}

__global__ void init_state_kernel(int B, int n_fast, int* iota, int* fastpos, int* pos, int* frames, int* done, int* mask,
                                  float* margin, int* margin_at, int* cur_col, int* new_col, int* salt, int* attn_ticket) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = b; i < ATT_SPLIT_MAX_PAIRS; i += gridDim.x * blockDim.x) attn_ticket[i] = 0;
  if (b >= B) return;
  margin_at[b] = 0;
  iota[b] = b;
  for (int i = 0; i < n_fast; ++i) fastpos[i * B + b] = i;
  pos[b] = 0; frames[b] = 0; done[b] = 1; mask[b] = 0; salt[b] = 0;
  margin[b] = INFINITY;
  for (int i = 0; i <= n_fast; ++i) { cur_col[b * (1 + n_fast) + i] = 0; new_col[b * (1 + n_fast) + i] = 0; }
}

// Restart the listed slots: position = prompt length, counters cleared; only they commit this frame.
__global__ void slot_reset_kernel(int B, int n_slots, const int* slots, const int* last_row, const int* row_pos, int* pos,
                                  int* frames, int* done, int* mask, float* margin, int* salt) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int hit = -1;
  for (int i = 0; i < n_slots; ++i)
    if (slots[i] == b) hit = i;
  mask[b] = hit >= 0;
  if (hit >= 0) {
    pos[b] = row_pos[last_row[hit]] + 1;
    frames[b] = 0; done[b] = 0; margin[b] = INFINITY;
    salt[b] += 1;  // a new tenant: its samples must not repeat the previous tenant's (same seed, slot and frame numbers)
  }
}

// Deferred start: the prompt's KV rows are in place; the slot is armed so that the next decode frame takes the last prompt
// column as its slow-step input at that column's position and emits frame 0 in its own tail (no separate tail launch).
__global__ void slot_start_kernel(int B, int H, int n_slots, const int* slots, const int* last_row, const int* row_pos,
                                  const int* grid, int* pos, int* frames, int* done, float* margin, int* cur_col, int* salt) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  for (int i = 0; i < n_slots; ++i)
    if (slots[i] == b) {
      const int lr = last_row[i];
      pos[b] = row_pos[lr];
      frames[b] = 0; done[b] = 0; margin[b] = INFINITY;
      salt[b] += 1;
      for (int k = 0; k < H; ++k) cur_col[b * H + k] = grid[(long)lr * H + k];
    }
}

// Chunked prefill: the listed slots stay idle (done, masked) and their position counter is parked just
// behind the chunk, where the idle slot's decode rows may scribble without harm (the next chunk rewrites
// that cache row before any attention reads it).
__global__ void slot_park_kernel(int B, int n_slots, const int* slots, const int* last_row, const int* row_pos, int* pos,
                                 int* done, int* mask) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  for (int i = 0; i < n_slots; ++i)
    if (slots[i] == b) { pos[b] = row_pos[last_row[i]] + 1; done[b] = 1; mask[b] = 0; }
}

// The same with the parking positions given by the host (smoltts_lm_park_slots): the slots of tenants whose prompts are about to
// be prefilled beside the running frames.
__global__ void slot_parkpos_kernel(int B, int n_slots, const int* slots, const int* where, int* pos, int* done, int* mask) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  for (int i = 0; i < n_slots; ++i)
    if (slots[i] == b) { pos[b] = where[i]; done[b] = 1; mask[b] = 0; }
}

// End of frame (lm/generate.py:143-171) and start of the next one, one workgroup per slot:
//  (1) commit (when `do_commit` and the slot's mask is set): publish the column, advance counters, apply the stop rule.
//      `advance_pos`: the slow step of this frame consumed one new token (decode) vs. the prompt (prefill).  A slot also
//      stops when its context is full (the next token would land at position max_seq: no KV row, no RoPE row for it) --
//      the reference's torch model has the same hard limit (max_seq_len buffers, modeling/...:180-199).
//  (2) mask = !done for the next frame (only live slots commit there);
//  (3) embed the slot's current column as the next slow step's input row (residual stream + X3 operand + sum of squares
//      for layer 0), so a decode frame starts with the first QKV GEMM: no separate mask / embed launches.
// Invariant between API calls: (xt, x3n, ssq) hold the embedded current columns of all slots and mask == !done.
struct CommitArgs {
  int B, H, max_frames, max_seq, im_end, stop_on_eos, advance_pos, do_commit;
  int* new_col;
  int *mask, *cur_col, *codes, *pos, *frames, *done;
  // the two picks nothing else in the frame waits for happen here instead of in launches of their own: the slow token
  // (lm/generate.py:88-99: only the commit needs it) and the last depth code (no further step consumes it); null = already in new_col
  const float* slow_logits; int slow_cols; SampleArgs slow_sa;
  const float* last_logits; int last_cols; SampleArgs last_sa;
  float* margin;
  // greedy picks from the head GEMMs' tile candidates (SmolttsGemm3Args.cand_out_dev) instead of the rows of logits: one wave
  // each, side by side, 2.4 KB instead of 9.5 + 8 KB per slot to look at; null = from the logits (argmax_row: sampling)
  const float* slow_cand; int slow_tiles;
  const float* last_cand; int last_tiles;
  int* attn_ticket;  // the key-split attention's arrival tickets: back to zero at the end of every frame, so that no history (an
                     // interrupted launch, a caller's odd count) can invert a later launch's "who arrived last" decision
};

__global__ __launch_bounds__(256) void commit_embed_kernel(CommitArgs a, EmbedTables t, float* xt, EmitDev emit) {
  __shared__ int s_col[64];
  __shared__ float sh4[4];
  __shared__ ArgmaxScratch s_pick[2];
  const int b = blockIdx.x, tid = threadIdx.x, H = a.H;  // H <= 64: the whole commit happens inside wave 0, in program order
  if (b == 0 && a.attn_ticket)
    for (int i = tid; i < ATT_SPLIT_MAX_PAIRS; i += blockDim.x) a.attn_ticket[i] = 0;
  int id_slow = 0, id_last = 0;
  const bool c_slow = a.slow_logits && a.slow_cand, c_last = a.last_logits && a.last_cand;
  if (c_slow || c_last) {  // (uniform) greedy picks from tile candidates: wave 0 the slow token, wave 1 the last depth code
    __shared__ int s_id[2];
    __shared__ float s_gap[2];
    const int wave = tid >> 6, lane = tid & 63;
    if ((wave == 0 && c_slow) || (wave == 1 && c_last)) {
      const float* cand = wave == 0 ? a.slow_cand + (size_t)b * a.slow_tiles * 4 : a.last_cand + (size_t)b * a.last_tiles * 4;
      Top2 t = cand_pick_wave(cand, wave == 0 ? a.slow_tiles : a.last_tiles, lane);
      if (t.i1 < 0 || t.i1 >= (wave == 0 ? a.slow_cols : a.last_cols)) t.i1 = 0;  // all-NaN row: stay inside the tables
      if (lane == 0) { s_id[wave] = t.i1; s_gap[wave] = t.v1 - t.v2; }
    }
    __syncthreads();
    if (tid == 0 && a.margin && a.mask[b]) {  // argmax_row's gap records, in its order: the slow token's, then the last code's
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (!(k == 0 ? c_slow : c_last)) continue;
        const SampleArgs& sa = k == 0 ? a.slow_sa : a.last_sa;
        if (s_gap[k] < a.margin[b]) {
          a.margin[b] = s_gap[k];
          if (sa.margin_at) sa.margin_at[b] = (sa.frames ? sa.frames[b] : sa.frame_base + b) * 64 + sa.step;
        }
      }
    }
    if (c_slow) id_slow = s_id[0];
    if (c_last) id_last = s_id[1];
  }
  if (a.slow_logits && !c_slow) id_slow = argmax_row(a.slow_logits + (long)b * a.slow_cols, a.slow_cols, (a.slow_cols & 3) == 0, b, a.margin, a.mask, a.slow_sa, s_pick[0]);
  if (a.last_logits && !c_last) id_last = argmax_row(a.last_logits + (long)b * a.last_cols, a.last_cols, (a.last_cols & 3) == 0, b, a.margin, a.mask, a.last_sa, s_pick[1]);
  int f = 0;
  bool live = false;
  if (tid < 64) { f = a.frames[b]; live = a.do_commit && a.mask[b]; }
  const bool publish = live && f < a.max_frames;
  int slow_now = 0;
  if (tid < H) {
    int nv = a.new_col[b * H + tid];
    if (a.slow_logits && tid == 0) nv = id_slow;
    if (a.last_logits && tid == H - 1) nv = id_last;
    if ((a.slow_logits && tid == 0) || (a.last_logits && tid == H - 1)) a.new_col[b * H + tid] = nv;  // (the host-visible column stays whole)
    const int v = publish ? nv : a.cur_col[b * H + tid];
    s_col[tid] = v;
    if (publish) {
      a.cur_col[b * H + tid] = v;
      a.codes[((long)b * a.max_frames + f) * H + tid] = v;
    }
  }
  slow_now = a.slow_logits ? id_slow : a.new_col[b * H];  // (tid 0 reads what it may just have written itself)
  if (tid == 0) {
    int d = a.done[b];
    if (live) {
      int p = a.pos[b];
      if (a.advance_pos) a.pos[b] = ++p;
      if (publish) a.frames[b] = f + 1;
      if ((a.stop_on_eos && slow_now == a.im_end) || f + 1 >= a.max_frames || p >= a.max_seq) a.done[b] = d = 1;
    }
    a.mask[b] = !d;
  }
  __syncthreads();
  embed_row(t, s_col, b, xt, emit, sh4);
}

// `fmt` = the engine's weight_format; an fp8 matrix is its tiles (N*K bytes, N % 16 == 0) followed by N row scales
SmolttsGemm3Args base3(int fmt, const void* w, const void* x3, int M, int N, int K, int epilogue) {
  SmolttsGemm3Args a;
  memset(&a, 0, sizeof(a));
  a.w_dev = w; a.x3_dev = x3; a.M = M; a.N = N; a.K = K; a.epilogue = epilogue;
  a.w_format = fmt;
  if (fmt == SMOLTTS_W_FP8) a.w_scale_dev = reinterpret_cast<const float*>(static_cast<const char*>(w) + (size_t)N * K);
  return a;
}

// Measurement aid (smoltts_session_measure_duplicate): kernel classes are named by their GEMM epilogue (+ N), or
// 100 = short-cache (depth) attention, 101 = long-cache (slow) attention.  Only idempotent launches can be doubled
// (EPI_RESID accumulates into its own input).
bool dup_hit(const SmolttsSession* s, int code, int N) { return s->dup_code == code && (s->dup_n <= 0 || s->dup_n == N); }

int launch_gemm3_m(const SmolttsSession* s, const SmolttsGemm3Args& a, hipStream_t st) {
  ST_TRY(launch_gemm3(a, st));
  if (a.epilogue != SMOLTTS_EPI_RESID && dup_hit(s, a.epilogue, a.N)) ST_TRY(launch_gemm3(a, st));
  return SMOLTTS_OK;
}

// One pre-norm block (modeling/model/rq_transformer.py:492-501) over `M` rows of the fp32 stream x (in place).
// `in_x3`/ssq: x published for the wqkv GEMM by whoever produced x; `next`: where the block publishes its output.
int run_block(const SmolttsSession* s, const SmolttsBlockWeights& bw, int dim, int n_head, int n_kv, int inter, float* x,
              float* q, int M, const int* row_pos, const int* row_slot, const float* rope, void* kc, void* vc,
              int cache_len, const char* in_x3, const EmitArgs& next, hipStream_t st, bool first_pos = false,
              int kv_format = SMOLTTS_KV_F32, int iota_pos = -1, bool qkv_done = false, int w_stream = 0,
              const SmolttsPickArgs* pick = nullptr) {
  // pick: the rows' codes of the previous depth step have not been picked yet -- this block's attention + wo launch does it
  // (q | k | v out of the table, residual = the picked embedding row: SmolttsPickArgs); implies qkv_done and the fused attention
  // w_stream: 1 = every matrix of the block, or a mask of SMOLTTS_STREAM_W_DEPTH_* bits (which of a depth block's matrices)
  // qkv_done: q and the cache rows of this block are in place already (gathered from the engine's fast_qkv table by the
  // kernel that picked the row's code): no wqkv launch
  // iota_pos >= 0: row r is slot r at that position (the depth steps): the attention kernel needs no row_pos / row_slot loads
  // first_pos: every row is at position 0 (depth step 0), so attention over its single key is the row's own V: the QKV
  // epilogue publishes V as wo's operand and the attention launch is skipped.
  const SmolttsEngine* e = s->e;
  const char* A = e->arena;
  const float eps = e->cfg.norm_eps;
  if (!qkv_done) {  // RMSNorm scale + QKV + RoPE + cache write
    SmolttsGemm3Args a = base3(e->cfg.weight_format, A + bw.wqkv, in_x3, M, (n_head + 2 * n_kv) * 64, dim, SMOLTTS_EPI_QKV_ROPE);
    a.fp8_activations = s->fp8_prefill;
    a.ssq_in_dev = s->ssq; a.eps = eps; a.out_dev = q; a.ldo = n_head * 64;
    a.rope_dev = rope; a.row_pos_dev = row_pos; a.row_slot_dev = row_slot;
    a.k_cache_dev = (float*)kc; a.v_cache_dev = (float*)vc; a.n_q_heads = n_head; a.n_kv_heads = n_kv; a.cache_len = cache_len;
    a.kv_format = kv_format; a.w_stream = (w_stream & (1 | SMOLTTS_STREAM_W_DEPTH_QKVO)) != 0;
    if (first_pos) a.v_x3_dev = s->x3a;
    ST_TRY(launch_gemm3_m(s, a, st));
  }
  // depth steps 1..: the <= 8-key attention is recomputed by every workgroup of the wo launch for its own rows (gemm3.hip
  // attn_wo_kernel): one launch less per layer
  const bool fused_attn = !first_pos && iota_pos > 0 && s->fuse_depth_attn && kv_format == SMOLTTS_KV_F32 &&
                          smoltts_gemm3_attn_fusable(n_head, n_kv, cache_len) && dim == n_head * 64;
  if (!first_pos && !fused_attn) {
    const int reps = dup_hit(s, cache_len <= 16 ? 100 : 101, 0) ? 2 : 1;
    for (int i = 0; i < reps; ++i)
      ST_TRY(launch_attention(q, kc, vc, row_pos, row_slot, M, n_head, n_kv, cache_len, 0, nullptr, s->x3a, st, kv_format, iota_pos,
                              s->split_attn ? s->attn_part : nullptr, s->attn_ticket));
  }
  {  // x += attn . Wo^T ; publish x * ffn_norm for w1|w3
    SmolttsGemm3Args a = base3(e->cfg.weight_format, A + bw.wo, s->x3a, M, dim, dim, SMOLTTS_EPI_RESID);
    a.fp8_activations = s->fp8_prefill;
    a.resid_dev = x; a.out_dev = x; a.ldo = dim;
    a.emit_a_dev = s->x3n; a.gamma_a_dev = (const float*)(A + bw.ffn_norm); a.ssq_out_dev = s->ssq; a.w_stream = (w_stream & (1 | SMOLTTS_STREAM_W_DEPTH_QKVO)) != 0;
    if (fused_attn) {
      a.attn_q_dev = q; a.attn_pos = iota_pos; a.k_cache_dev = (float*)kc; a.v_cache_dev = (float*)vc;
      a.n_q_heads = n_head; a.n_kv_heads = n_kv; a.cache_len = cache_len; a.kv_format = SMOLTTS_KV_F32;
      a.pick = pick;
    }
    ST_REQUIRE(pick == nullptr || fused_attn, SMOLTTS_E_STATE, "run_block: a pick needs the fused depth attention");
    ST_TRY(launch_gemm3(a, st));
  }
  {  // h = silu(w1 n) * (w3 n), n = RMSNorm(x); written as w2's operand
    SmolttsGemm3Args a = base3(e->cfg.weight_format, A + bw.w13, s->x3n, M, 2 * inter, dim, SMOLTTS_EPI_SWIGLU);
    a.fp8_activations = s->fp8_prefill;
    a.ssq_in_dev = s->ssq; a.eps = eps; a.x3_out_dev = s->x3h; a.w_stream = (w_stream & (1 | SMOLTTS_STREAM_W_DEPTH_W13)) != 0;
    ST_TRY(launch_gemm3_m(s, a, st));
  }
  {  // x += h . W2^T ; publish for the next consumer(s)
    SmolttsGemm3Args a = base3(e->cfg.weight_format, A + bw.w2, s->x3h, M, dim, inter, SMOLTTS_EPI_RESID);
    a.fp8_activations = s->fp8_prefill;
    a.resid_dev = x; a.out_dev = x; a.ldo = dim;
    a.emit_a_dev = next.x3a; a.gamma_a_dev = next.gamma_a; a.emit_b_dev = next.x3b; a.gamma_b_dev = next.gamma_b;
    a.ssq_out_dev = next.ssq; a.w_stream = (w_stream & (1 | SMOLTTS_STREAM_W_DEPTH_W2)) != 0;
    ST_TRY(launch_gemm3(a, st));
  }
  return SMOLTTS_OK;
}

const float* gamma_at(const SmolttsEngine* e, uint64_t off) { return (const float*)(e->arena + off); }

// What the producer of the slow hidden state publishes: head operand (x * norm) and the depth
// transformer's first operand (x * fast_layers[0].attention_norm, or raw x for fast_project_in).
EmitArgs slow_hidden_emit(const SmolttsSession* s) {
  const SmolttsEngine* e = s->e;
  EmitArgs em{s->x3n, gamma_at(e, e->w.norm), s->x3n2,
              e->cfg.has_fast_project_in ? nullptr : gamma_at(e, e->w.fast_layers[0].attn_norm), s->ssq};
  return em;
}

// xt[slots[i]] = xr[last_row[i]]
__global__ __launch_bounds__(256) void scatter_last_kernel(const float* xr, const int* slots, const int* last_row, int dim,
                                                           float* xt) {
  const int i = blockIdx.x;
  const long src = last_row[i], dst = slots[i];
  for (int d = threadIdx.x * 4; d < dim; d += 256 * 4)
    *reinterpret_cast<float4*>(xt + dst * dim + d) = *reinterpret_cast<const float4*>(xr + src * dim + d);
}

EmbedTables embed_tables(const SmolttsEngine* e) {
  const SmolttsLMConfig& c = e->cfg;
  const char* A = e->arena;
  return EmbedTables{(const uint16_t*)(A + e->w.text_emb), (const uint16_t*)(A + e->w.codebook_emb), c.dim, c.codebook_size,
                     c.duplicate_code_0 ? 0 : c.codebook_size, c.embed_mask_mode, c.semantic_start_id, c.semantic_end_id,
                     c.vocab_size, c.codebook_size * c.num_codebooks, c.n_fast};
}

// The commit kernel's greedy picks read the head GEMMs' tile candidates where run_tail has had them written (same conditions there)
bool commit_cand_slow(const SmolttsSession* s) { return s->commit_picks && s->fuse_pick && s->temp <= 0.f && s->B < 256; }
bool commit_cand_last(const SmolttsSession* s) { return s->commit_picks && s->fuse_pick && s->fast_temp <= 0.f && s->B < 256; }

// commit (optional) + next-frame mask + embedding of every slot's current column (see commit_embed_kernel); `picks`: the
// frame's slow token and last depth code are still logits (run_tail) and are picked here
int launch_commit_embed(SmolttsSession* s, int do_commit, int advance_pos, hipStream_t st, bool picks = false) {
  const SmolttsEngine* e = s->e;
  const SmolttsLMConfig& c = e->cfg;
  ST_REQUIRE(1 + c.n_fast <= 64 && c.dim % 64 == 0, SMOLTTS_E_INVALID, "commit: grid height > 64 or dim %% 64 != 0");
  CommitArgs a;
  memset(&a, 0, sizeof(a));
  a.B = s->B; a.H = 1 + c.n_fast; a.max_frames = s->max_frames; a.max_seq = s->max_seq; a.im_end = c.im_end_id;
  a.stop_on_eos = s->stop_on_eos; a.advance_pos = advance_pos; a.do_commit = do_commit;
  a.new_col = s->new_col; a.mask = s->mask; a.cur_col = s->cur_col; a.codes = s->codes; a.pos = s->pos; a.frames = s->frames; a.done = s->done;
  a.margin = s->margin;
  a.attn_ticket = s->attn_ticket;
  if (picks) {
    a.slow_logits = s->logits_slow; a.slow_cols = c.vocab_size;
    a.slow_sa = SampleArgs{s->temp, s->min_p, s->seed, 0, 0, s->frames, s->salt, s->margin_at};
    a.last_logits = s->logits; a.last_cols = c.codebook_size;
    a.last_sa = SampleArgs{s->fast_temp, s->fast_temp > 0.f ? s->min_p : 0.f, s->seed, c.n_fast, 0, s->frames, s->salt, s->margin_at};
    if (commit_cand_slow(s)) { a.slow_cand = s->cand_slow; a.slow_tiles = (c.vocab_size + 15) / 16; }
    if (commit_cand_last(s)) { a.last_cand = s->cand; a.last_tiles = (c.codebook_size + 15) / 16; }
  }
  const EmitDev em{s->x3n, gamma_at(e, e->w.layers[0].attn_norm), nullptr, nullptr, s->ssq};
  hipLaunchKernelGGL(commit_embed_kernel, dim3(s->B), dim3(256), 0, st, a, embed_tables(e), s->xt, em);
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

// Slow head + the depth transformer for all B slots; columns land in new_col, commit applies `mask`.
// Precondition: xt holds the pre-norm slow hidden and has been published via slow_hidden_emit().
int run_tail(SmolttsSession* s, int advance_pos, hipStream_t st) {
  const SmolttsEngine* e = s->e;
  const SmolttsLMConfig& c = e->cfg;
  const char* A = e->arena;
  const int B = s->B, H = 1 + c.n_fast;
  {  // logits = RMSNorm(x) . E^T   (lm/rq_transformer.py:184-189); the token is picked by the commit kernel: nothing before it needs it
    SmolttsGemm3Args a = base3(c.weight_format, A + e->w.head, s->x3n, B, c.vocab_size, c.dim, SMOLTTS_EPI_STORE);
    a.ssq_in_dev = s->ssq; a.eps = c.norm_eps; a.out_dev = s->logits_slow; a.ldo = c.vocab_size; a.w_stream = (s->stream_w & SMOLTTS_STREAM_W_SLOW_HEAD) != 0;
    if (commit_cand_slow(s)) a.cand_out_dev = s->cand_slow;
    ST_TRY(launch_gemm3_m(s, a, st));
  }
  const bool picks = s->commit_picks;
  if (!picks) {  // (A/B: the pick as a launch of its own)
    const SampleArgs slow_sa{s->temp, s->min_p, s->seed, 0, 0, s->frames, s->salt, s->margin_at};
    ST_TRY(launch_argmax(s->logits_slow, B, c.vocab_size, c.vocab_size, s->new_col, H, s->margin, s->mask, nullptr, 0, 0, nullptr,
                         nullptr, &slow_sa, st));
  }
  float* xf = s->xt;  // fast input = pre-norm slow hidden (lm/rq_transformer.py:191)
  const char* first_x3 = s->x3n2;
  const EmitArgs to_fast0{s->x3n, gamma_at(e, e->w.fast_layers[0].attn_norm), nullptr, nullptr, s->ssq};
  if (c.has_fast_project_in) {  // Linear(dim, fast_dim) with bias (modeling :339-342)
    SmolttsGemm3Args a = base3(c.weight_format, A + e->w.fast_proj_w, s->x3n2, B, c.fast_dim, c.dim, SMOLTTS_EPI_STORE);
    a.bias_dev = (const float*)(A + e->w.fast_proj_b); a.out_dev = s->xf; a.ldo = c.fast_dim;
    a.emit_a_dev = to_fast0.x3a; a.gamma_a_dev = to_fast0.gamma_a; a.ssq_out_dev = s->ssq;
    ST_TRY(launch_gemm3(a, st));
    xf = s->xf;
    first_x3 = s->x3n;
  }
  const size_t fl_stride = (size_t)B * c.fast_n_kv_head * c.n_fast * 64;
  const bool table = e->fast_qkv != nullptr && s->use_qkv_table;
  // greedy depth codes are picked by the launch that consumes them (the next step's layer-0 attention + wo: SmolttsPickArgs)
  const bool pick_fused = s->fuse_pick && table && s->fuse_depth_attn && s->fast_temp <= 0.f && B < 256 && c.codebook_size % 16 == 0 &&
                          c.fast_dim == c.fast_n_head * 64 && smoltts_gemm3_attn_fusable(c.fast_n_head, c.fast_n_kv_head, c.n_fast);
  SmolttsPickArgs pk;
  memset(&pk, 0, sizeof(pk));
  bool pk_pending = false;  // step i - 1's code is still candidates in s->cand
  for (int i = 0; i < c.n_fast; ++i) {
    for (int l = 0; l < c.n_fast_layer; ++l) {
      const EmitArgs next{s->x3n,
                          gamma_at(e, l + 1 < c.n_fast_layer ? e->w.fast_layers[l + 1].attn_norm : e->w.fast_norm),
                          nullptr, nullptr, s->ssq};
      ST_TRY(run_block(s, e->w.fast_layers[l], c.fast_dim, c.fast_n_head, c.fast_n_kv_head, c.fast_inter, xf, s->qt, B,
                       s->fastpos + (size_t)i * B, s->iota, (const float*)(A + e->w.fast_rope), s->fkc + l * fl_stride,
                       s->fvc + l * fl_stride, c.n_fast, (i == 0 && l == 0) ? first_x3 : s->x3n, next, st, /*first_pos=*/i == 0,
                       SMOLTTS_KV_F32, /*iota_pos=*/i, /*qkv_done=*/table && i > 0 && l == 0,
                       /*w_stream=*/s->stream_w & (SMOLTTS_STREAM_W_DEPTH_QKVO | SMOLTTS_STREAM_W_DEPTH_W13 | SMOLTTS_STREAM_W_DEPTH_W2),
                       (pk_pending && l == 0) ? &pk : nullptr));
      if (l == 0) pk_pending = false;
    }
    {  // fast_norm + depthwise head slice i  (lm/rq_transformer.py:209-217)
      const size_t wrow = (size_t)i * e->w.fast_head_step_stride;  // rows; a row tile is 16 rows
      const size_t wbytes = c.weight_format == SMOLTTS_W_FP8 ? wrow * (c.fast_dim + 4) : wrow * c.fast_dim * 2;
      SmolttsGemm3Args a = base3(c.weight_format, A + e->w.fast_head + wbytes, s->x3n, B, c.codebook_size, c.fast_dim, SMOLTTS_EPI_STORE);
      a.ssq_in_dev = s->ssq; a.eps = c.norm_eps; a.out_dev = s->logits; a.ldo = c.codebook_size;
      a.w_stream = (s->stream_w & SMOLTTS_STREAM_W_DEPTH_HEAD) != 0;  // each head slice is read once per frame
      if (pick_fused && i + 1 < c.n_fast) a.cand_out_dev = s->cand;
      if (i + 1 == c.n_fast && commit_cand_last(s)) a.cand_out_dev = s->cand;  // (no pick is pending on the buffer at the last step)
      ST_TRY(launch_gemm3_m(s, a, st));
    }
    const SampleArgs fast_sa{s->fast_temp, s->fast_temp > 0.f ? s->min_p : 0.f, s->seed, 1 + i, 0, s->frames, s->salt, s->margin_at};
    if (i + 1 == c.n_fast) {  // the last code has no consumer inside the frame: picked by the commit kernel
      if (!picks)
        ST_TRY(launch_argmax(s->logits, B, c.codebook_size, c.codebook_size, s->new_col + 1 + i, H, s->margin, s->mask, nullptr, 0, 0,
                             nullptr, nullptr, &fast_sa, st));
      break;
    }
    const int off = c.depthwise_wte ? (c.duplicate_code_0 ? i : i + 1) * c.codebook_size : 0;  // generate.py:136-140
    float* xnext = c.has_fast_project_in ? s->xf : s->xt;
    if (pick_fused) {  // no launch here: layer 0 of step i + 1 picks (the residual row it starts from is the embedding row itself)
      pk.cand_dev = s->cand; pk.cand_tiles = c.codebook_size / 16; pk.qkv_table_dev = e->fast_qkv;
      pk.rope_dev = (const float*)(A + e->w.fast_rope); pk.emb_dev = A + e->w.fast_emb; pk.emb_row_offset = off;
      pk.ids_dev = s->new_col + 1 + i; pk.ids_stride = H; pk.margin_dev = s->margin; pk.margin_mask_dev = s->mask;
      pk.margin_at_dev = s->margin_at; pk.frames_dev = s->frames; pk.step = 1 + i;
      pk_pending = true;
      xf = xnext;
      continue;
    }
    // the next step's layer-0 q | k | v come out of the engine's table (position i + 1) where it has been built: the row then
    // needs no X3 operand of its own, only the fp32 residual
    QkvGather qg;
    memset(&qg, 0, sizeof(qg));
    if (table)
      qg = QkvGather{e->fast_qkv, (const float*)(A + e->w.fast_rope), s->qt, s->fkc, s->fvc, c.fast_n_head, c.fast_n_kv_head, c.n_fast, i + 1};
    ST_TRY(launch_argmax(s->logits, B, c.codebook_size, c.codebook_size, s->new_col + 1 + i, H, s->margin, s->mask,
                         (const void*)(A + e->w.fast_emb), off, c.fast_dim, xnext, table ? nullptr : &to_fast0, &fast_sa, st,
                         table ? &qg : nullptr));
    xf = xnext;
  }
  return launch_commit_embed(s, /*do_commit=*/1, advance_pos, st, picks);
}

// Slow transformer over M rows of x (already embedded and published for layer 0).
int run_slow_layers(SmolttsSession* s, float* x, float* q, int M, const int* row_pos, const int* row_slot, bool publish_hidden,
                    hipStream_t st) {
  const SmolttsEngine* e = s->e;
  const SmolttsLMConfig& c = e->cfg;
  const size_t l_stride = (size_t)s->B * c.n_kv_head * s->max_seq * 64 * (s->kv_format == SMOLTTS_KV_BF16 ? 2 : 4);  // bytes
  for (int l = 0; l < c.n_layer; ++l) {
    EmitArgs next{nullptr, nullptr, nullptr, nullptr, nullptr};
    if (l + 1 < c.n_layer) next = EmitArgs{s->x3n, gamma_at(e, e->w.layers[l + 1].attn_norm), nullptr, nullptr, s->ssq};
    else if (publish_hidden) next = slow_hidden_emit(s);
    ST_TRY(run_block(s, e->w.layers[l], c.dim, c.n_head, c.n_kv_head, c.inter, x, q, M, row_pos, row_slot,
                     (const float*)(e->arena + e->w.rope), s->kc + l * l_stride, s->vc + l * l_stride, s->max_seq, s->x3n, next, st,
                     /*first_pos=*/false, s->kv_format, -1, false, /*w_stream=*/((s->stream_w & SMOLTTS_STREAM_W_SLOW) && M <= 128)
                         ? ((s->stream_w & SMOLTTS_STREAM_W_SLOW_ONLY_W13) ? SMOLTTS_STREAM_W_DEPTH_W13 : ((s->stream_w & SMOLTTS_STREAM_W_SLOW_NOT_W13) ? (SMOLTTS_STREAM_W_DEPTH_W2 | SMOLTTS_STREAM_W_DEPTH_QKVO) : 1)) : 0));
  }
  return SMOLTTS_OK;
}

int embed_rows(SmolttsSession* s, const int* cols, int M, float* x, hipStream_t st) {
  const SmolttsEngine* e = s->e;
  const SmolttsLMConfig& c = e->cfg;
  const char* A = e->arena;
  const int cb_first = c.duplicate_code_0 ? 0 : c.codebook_size;
  const EmitArgs em{s->x3n, gamma_at(e, e->w.layers[0].attn_norm), nullptr, nullptr, s->ssq};
  return launch_embed(cols, M, c.n_fast, A + e->w.text_emb, A + e->w.codebook_emb, c.dim, c.codebook_size, cb_first,
                      c.embed_mask_mode, c.semantic_start_id, c.semantic_end_id, c.vocab_size, c.codebook_size * c.num_codebooks,
                      x, &em, st);
}

// One decode frame.  Precondition (the session invariant): xt / x3n / ssq hold the embedded current columns, mask == !done.
int run_decode_frame(SmolttsSession* s, hipStream_t st) {
  ST_TRY(run_slow_layers(s, s->xt, s->qt, s->B, s->pos, s->iota, /*publish_hidden=*/true, st));
  return run_tail(s, /*advance_pos=*/1, st);
}

// Slot / last-row lists of a prefill call -> device, through the session's pinned staging buffer.  Only the previous
// upload is waited for (an event), never the stream: a serving loop calls this while earlier frames are still running.
int stage_upload(SmolttsSession* s, const int32_t* slots_host, const int32_t* last_row_host, int n_slots, hipStream_t st) {
  // a ring of pinned areas: the host only waits when STAGE_RING uploads are still queued behind other work on `st`
  // (a serving loop uploads while a tick of frame graphs is pending; waiting for the previous upload would stall it)
  const int k = s->stage_next;
  s->stage_next = (k + 1) % STAGE_RING;
  if (s->stage_ev_live[k]) ST_CHECK_HIP(hipEventSynchronize(s->stage_ev[k]));
  int* h = s->h_stage + (size_t)k * 2 * s->B;
  for (int i = 0; i < n_slots; ++i) { h[i] = slots_host[i]; h[s->B + i] = last_row_host[i]; }
  ST_CHECK_HIP(hipMemcpyAsync(s->stage_slots, h, sizeof(int) * n_slots, hipMemcpyHostToDevice, st));
  ST_CHECK_HIP(hipMemcpyAsync(s->stage_last, h + s->B, sizeof(int) * n_slots, hipMemcpyHostToDevice, st));
  ST_CHECK_HIP(hipEventRecord(s->stage_ev[k], st));
  s->stage_ev_live[k] = true;
  return SMOLTTS_OK;
}

void drop_graphs(SmolttsSession* s) {
  if (s->graph_ready) { (void)hipGraphExecDestroy(s->graph_exec); s->graph_ready = false; }
  if (s->tail_ready) { (void)hipGraphExecDestroy(s->tail_exec); s->tail_ready = false; }
  if (s->multi_ready) { (void)hipGraphExecDestroy(s->multi_exec); s->multi_ready = false; }
}

// Record `body` (a fixed launch sequence on the given stream) into an executable graph.
template <typename F>
int capture_graph(hipStream_t st, hipGraphExec_t* exec, F body) {
  hipStream_t cap = st;
  bool own = false;
  if (cap == nullptr) {  // the legacy default stream cannot be captured
    ST_CHECK_HIP(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    own = true;
  }
  ST_CHECK_HIP(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
  const int rc = body(cap);
  hipGraph_t graph = nullptr;
  const hipError_t ee = hipStreamEndCapture(cap, &graph);
  if (own) (void)hipStreamDestroy(cap);
  if (rc != SMOLTTS_OK) {
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
  }
  ST_CHECK_HIP(ee);
  const hipError_t ei = hipGraphInstantiate(exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  ST_CHECK_HIP(ei);
  return SMOLTTS_OK;
}

bool graphs_enabled() {
  const char* no_graph = getenv("SMOLTTS_NO_GRAPH");
  return !(no_graph && no_graph[0] == '1');
}

int run_decode_frame(SmolttsSession* s, hipStream_t st);

// The graph of `multi_frames` consecutive frames (every graph launch costs the GPU a gap between the last node of one graph and
// the first of the next); a graph of several frames is that many frames queued at once, so it stays inside the run-ahead bound.
int ensure_multi_graph(SmolttsSession* s, int n_frames, hipStream_t st) {
  if (s->flight_limit > 0 && s->multi_frames > s->flight_limit) s->multi_frames = s->flight_limit;
  const int mf = s->multi_frames;
  if (mf > 1 && n_frames >= mf && !s->multi_ready) {
    ST_TRY(capture_graph(st, &s->multi_exec, [&](hipStream_t cap) {
      for (int i = 0; i < mf; ++i) ST_TRY(run_decode_frame(s, cap));
      return (int)SMOLTTS_OK;
    }));
    s->multi_ready = true;
  }
  return SMOLTTS_OK;
}

int check_offsets(const SmolttsLMConfig& c, const SmolttsLMWeights& w, size_t bytes) {
  auto ok = [&](uint64_t off, size_t need) { return off % 16 == 0 && off + need <= bytes; };
  ST_REQUIRE(c.weight_format == SMOLTTS_W_BF16 || c.weight_format == SMOLTTS_W_FP8, SMOLTTS_E_INVALID,
             "engine: unknown weight_format %d", c.weight_format);
  const bool f8 = c.weight_format == SMOLTTS_W_FP8;
  // bytes of one Linear [rows][k]: bf16 tiles, or e4m3 tiles + fp32 row scales
  auto mat = [&](size_t rows, size_t k) { return f8 ? rows * (k + 4) : rows * k * 2; };
  const size_t d = c.dim, fd = c.fast_dim;
  if (f8)
    ST_REQUIRE(c.vocab_size % 16 == 0 && c.codebook_size % 16 == 0, SMOLTTS_E_INVALID, "engine: fp8 weights need vocab and codebook sizes %% 16 == 0");
  ST_REQUIRE(ok(w.text_emb, (size_t)c.vocab_size * d * 2) && ok(w.head, mat(c.vocab_size, d)) &&
                 ok(w.codebook_emb, (size_t)c.codebook_size * c.num_codebooks * d * 2) && ok(w.norm, d * 4) &&
                 ok(w.fast_norm, fd * 4) && ok(w.rope, (size_t)c.max_seq_len * 64 * 4) && ok(w.fast_rope, (size_t)c.n_fast * 64 * 4),
             SMOLTTS_E_INVALID, "engine: weight offsets outside the arena or misaligned");
  const size_t head_rows = (size_t)(c.n_fast - 1) * w.fast_head_step_stride + c.codebook_size;
  ST_REQUIRE(ok(w.fast_head, mat(head_rows, fd)), SMOLTTS_E_INVALID, "engine: fast_head outside the arena");
  ST_REQUIRE(w.fast_head_step_stride % 16 == 0, SMOLTTS_E_INVALID, "engine: fast_head_step_stride must be a multiple of 16");
  if (c.has_fast_project_in)
    ST_REQUIRE(ok(w.fast_proj_w, mat(fd, d)) && ok(w.fast_proj_b, fd * 4), SMOLTTS_E_INVALID, "engine: fast_project_in outside the arena");
  for (int l = 0; l < c.n_layer + c.n_fast_layer; ++l) {
    const bool fast = l >= c.n_layer;
    const SmolttsBlockWeights& b = fast ? w.fast_layers[l - c.n_layer] : w.layers[l];
    const size_t dd = fast ? fd : d, hh = fast ? c.fast_n_head : c.n_head, kk = fast ? c.fast_n_kv_head : c.n_kv_head;
    const size_t ii = fast ? c.fast_inter : c.inter;
    ST_REQUIRE(ok(b.attn_norm, dd * 4) && ok(b.ffn_norm, dd * 4) && ok(b.wqkv, mat((hh + 2 * kk) * 64, dd)) &&
                   ok(b.wo, mat(dd, dd)) && ok(b.w13, mat(2 * ii, dd)) && ok(b.w2, mat(dd, ii)),
               SMOLTTS_E_INVALID, "engine: layer %d weight offsets outside the arena or misaligned", l);
  }
  return SMOLTTS_OK;
}

}  // namespace

extern "C" {

int smoltts_engine_create(const SmolttsLMConfig* cfg, const SmolttsLMWeights* offsets, const void* arena_dev,
                          size_t arena_bytes, SmolttsEngine** out) {
  ST_REQUIRE(cfg && offsets && arena_dev && out, SMOLTTS_E_INVALID, "engine_create: null argument");
  const SmolttsLMConfig& c = *cfg;
  ST_REQUIRE(c.n_layer > 0 && c.n_layer <= SMOLTTS_MAX_LAYERS && c.n_fast_layer > 0 && c.n_fast_layer <= SMOLTTS_MAX_FAST_LAYERS,
             SMOLTTS_E_INVALID, "engine_create: layer counts out of range");
  ST_REQUIRE(c.dim == c.n_head * 64 && c.fast_dim == c.fast_n_head * 64, SMOLTTS_E_INVALID,
             "engine_create: dim must equal n_head * 64 (head_dim 64 only)");
  ST_REQUIRE(c.dim % 32 == 0 && c.fast_dim % 32 == 0 && c.inter % 32 == 0 && c.fast_inter % 32 == 0, SMOLTTS_E_INVALID,
             "engine_create: dims must be multiples of 32");
  ST_REQUIRE(c.n_head % c.n_kv_head == 0 && c.fast_n_head % c.fast_n_kv_head == 0 && c.n_head / c.n_kv_head <= 4 &&
                 c.fast_n_head / c.fast_n_kv_head <= 4,
             SMOLTTS_E_INVALID, "engine_create: unsupported GQA grouping");
  ST_REQUIRE(c.vocab_size % 16 == 0 && c.codebook_size % 16 == 0, SMOLTTS_E_INVALID,
             "engine_create: vocab_size and codebook_size must be multiples of 16");
  ST_REQUIRE(c.n_fast == c.num_codebooks - (c.duplicate_code_0 ? 0 : 1) && c.n_fast >= 1 && c.n_fast <= 32, SMOLTTS_E_INVALID,
             "engine_create: n_fast inconsistent with num_codebooks/duplicate_code_0");
  ST_REQUIRE(c.has_fast_project_in == (c.fast_dim != c.dim), SMOLTTS_E_INVALID,
             "engine_create: has_fast_project_in must be set exactly when fast_dim != dim");
  ST_REQUIRE(c.max_seq_len > 0, SMOLTTS_E_INVALID, "engine_create: max_seq_len");
  ST_TRY(check_offsets(c, *offsets, arena_bytes));
  SmolttsEngine* e = new (std::nothrow) SmolttsEngine;
  ST_REQUIRE(e, SMOLTTS_E_INVALID, "engine_create: out of host memory");
  e->cfg = c; e->w = *offsets; e->arena = (const char*)arena_dev; e->arena_bytes = arena_bytes;
  e->fast_qkv = nullptr;
  *out = e;
  return SMOLTTS_OK;
}

void smoltts_engine_destroy(SmolttsEngine* e) { delete e; }

// ---- derived table: depth layer-0 q | k | v of every fast-embedding row (see QkvGather in argmax_dev.h)
// Rows a depth step can be fed with: code + offset of the steps that have a successor (lm/generate.py:134-140).
static long fast_emb_rows_used(const SmolttsLMConfig& c) {
  if (c.n_fast < 2) return 0;
  return c.depthwise_wte ? (long)(c.duplicate_code_0 ? c.n_fast - 1 : c.n_fast) * c.codebook_size : (long)c.codebook_size;
}
constexpr int QKV_BUILD_ROWS = 128;  // rows per GEMM launch of the build: the decode kernels' own variant (M <= 128)

size_t smoltts_engine_fast_qkv_bytes(const SmolttsEngine* e) {
  if (!e) return 0;
  const SmolttsLMConfig& c = e->cfg;
  const long rows = fast_emb_rows_used(c);
  if (rows <= 0) return 0;
  const size_t nqkv = (size_t)(c.fast_n_head + 2 * c.fast_n_kv_head) * 64;
  const size_t table = align_up((size_t)rows * nqkv * sizeof(float));
  const size_t x3 = align_up((size_t)QKV_BUILD_ROWS * c.fast_dim * 6), ssq = align_up((size_t)QKV_BUILD_ROWS * (c.fast_dim / 16) * sizeof(float));
  return table + x3 + ssq;
}

int smoltts_engine_build_fast_qkv(SmolttsEngine* e, void* slab_dev, size_t slab_bytes, void* stream) {
  ST_REQUIRE(e && slab_dev, SMOLTTS_E_INVALID, "engine_build_fast_qkv: null argument");
  const SmolttsLMConfig& c = e->cfg;
  const size_t need = smoltts_engine_fast_qkv_bytes(e);
  ST_REQUIRE(need > 0, SMOLTTS_E_INVALID, "engine_build_fast_qkv: this model has no depth step with a successor");
  ST_REQUIRE(slab_bytes >= need && ((uintptr_t)slab_dev & 255) == 0, SMOLTTS_E_CAPACITY,
             "engine_build_fast_qkv: slab has %zu bytes (256-byte aligned), %zu needed", slab_bytes, need);
  const long rows = fast_emb_rows_used(c);
  ST_REQUIRE(e->w.fast_emb % 16 == 0 && e->w.fast_emb + (size_t)rows * c.fast_dim * 2 <= e->arena_bytes, SMOLTTS_E_INVALID,
             "engine_build_fast_qkv: the fast embedding table (%ld rows) lies outside the arena", rows);
  hipStream_t st = (hipStream_t)stream;
  const char* A = e->arena;
  const int nqkv = (c.fast_n_head + 2 * c.fast_n_kv_head) * 64;
  Carver cv{(char*)slab_dev, 0};
  float* table = cv.take<float>((size_t)rows * nqkv);
  char* x3 = cv.take<char>((size_t)QKV_BUILD_ROWS * c.fast_dim * 6);
  float* ssq = cv.take<float>((size_t)QKV_BUILD_ROWS * (c.fast_dim / 16));
  e->fast_qkv = nullptr;
  const SmolttsBlockWeights& bw = e->w.fast_layers[0];
  const EmitArgs em{x3, gamma_at(e, bw.attn_norm), nullptr, nullptr, ssq};
  for (long r0 = 0; r0 < rows; r0 += QKV_BUILD_ROWS) {
    const int n = (int)(rows - r0 < QKV_BUILD_ROWS ? rows - r0 : QKV_BUILD_ROWS);
    // the row exactly as the picking kernel would publish it (X3 of E[row] * attention_norm + its sum of squares) ...
    ST_TRY(launch_emb_rows_pack(A + e->w.fast_emb, r0, n, c.fast_dim, em, st));
    // ... through the decode path's own wqkv GEMM (same accumulation order), the RMSNorm scale applied, RoPE left to the gather
    SmolttsGemm3Args a = base3(c.weight_format, A + bw.wqkv, x3, n, nqkv, c.fast_dim, SMOLTTS_EPI_STORE);
    a.ssq_in_dev = ssq; a.eps = c.norm_eps; a.out_dev = table + (size_t)r0 * nqkv; a.ldo = nqkv;
    ST_TRY(launch_gemm3(a, st));
  }
  ST_CHECK_HIP(hipStreamSynchronize(st));  // the scratch part of the slab is dead from here; the table part must outlive the engine's sessions
  e->fast_qkv = table;
  return SMOLTTS_OK;
}

size_t smoltts_session_slab_bytes(const SmolttsEngine* e, int32_t max_batch, int32_t max_seq, int32_t max_rows,
                                  int32_t max_frames) {
  return smoltts_session_slab_bytes_kv(e, max_batch, max_seq, max_rows, max_frames, SMOLTTS_KV_F32);
}

size_t smoltts_session_slab_bytes_kv(const SmolttsEngine* e, int32_t max_batch, int32_t max_seq, int32_t max_rows,
                                     int32_t max_frames, int32_t kv_format) {
  if (!e || max_batch <= 0 || max_seq <= 0 || max_rows <= 0 || max_frames <= 0) return 0;
  if (kv_format != SMOLTTS_KV_F32 && kv_format != SMOLTTS_KV_BF16) return 0;
  SmolttsSession tmp;
  memset(&tmp, 0, sizeof(tmp));
  tmp.e = const_cast<SmolttsEngine*>(e);
  tmp.kv_format = kv_format;
  tmp.B = max_batch; tmp.max_seq = max_seq; tmp.max_rows = max_rows; tmp.max_frames = max_frames;
  size_t total = 0;
  carve(&tmp, nullptr, &total);
  return total;
}

int smoltts_session_create(SmolttsEngine* e, void* slab_dev, size_t slab_bytes, int32_t max_batch, int32_t max_seq,
                           int32_t max_rows, int32_t max_frames, SmolttsSession** out) {
  return smoltts_session_create_kv(e, slab_dev, slab_bytes, max_batch, max_seq, max_rows, max_frames, SMOLTTS_KV_F32, out);
}

int smoltts_session_create_kv(SmolttsEngine* e, void* slab_dev, size_t slab_bytes, int32_t max_batch, int32_t max_seq,
                              int32_t max_rows, int32_t max_frames, int32_t kv_format, SmolttsSession** out) {
  ST_REQUIRE(e && slab_dev && out, SMOLTTS_E_INVALID, "session_create: null argument");
  ST_REQUIRE(kv_format == SMOLTTS_KV_F32 || kv_format == SMOLTTS_KV_BF16, SMOLTTS_E_INVALID, "session_create: unknown kv_format %d", kv_format);
  ST_REQUIRE(kv_format == SMOLTTS_KV_F32 || max_seq > 16, SMOLTTS_E_INVALID, "session_create: a bf16 KV cache needs max_seq > 16");
  ST_REQUIRE(max_batch > 0 && max_batch <= 4096 && max_rows > 0 && max_frames > 0, SMOLTTS_E_INVALID, "session_create: bad sizes");
  ST_REQUIRE(max_seq > 0 && max_seq <= e->cfg.max_seq_len, SMOLTTS_E_CAPACITY,
             "session_create: max_seq %d exceeds the RoPE table (%d)", max_seq, e->cfg.max_seq_len);
  ST_REQUIRE(((uintptr_t)slab_dev & 255) == 0, SMOLTTS_E_INVALID, "session_create: slab must be 256-byte aligned");
  const size_t need = smoltts_session_slab_bytes_kv(e, max_batch, max_seq, max_rows, max_frames, kv_format);
  ST_REQUIRE(slab_bytes >= need, SMOLTTS_E_CAPACITY, "session_create: slab has %zu bytes, %zu needed", slab_bytes, need);
  SmolttsSession* s = new (std::nothrow) SmolttsSession;
  ST_REQUIRE(s, SMOLTTS_E_INVALID, "session_create: out of host memory");
  memset(s, 0, sizeof(*s));
  s->e = e; s->B = max_batch; s->max_seq = max_seq; s->max_rows = max_rows; s->max_frames = max_frames;
  s->dup_code = -1;
  s->use_qkv_table = true;
  s->commit_picks = true;
  s->split_attn = true;
  s->stream_w = SMOLTTS_STREAM_W_DEFAULT;
  s->fuse_depth_attn = true;
  s->fuse_pick = true;
  s->kv_format = kv_format;
  size_t total = 0;
  carve(s, (char*)slab_dev, &total);
  if (hipHostMalloc((void**)&s->h_stage, sizeof(int) * 2 * max_batch * STAGE_RING, hipHostMallocDefault) != hipSuccess) {
    delete s;
    set_error("session_create: hipHostMalloc failed");
    return SMOLTTS_E_HIP;
  }
  for (int k = 0; k < STAGE_RING; ++k) {
    if (hipEventCreateWithFlags(&s->stage_ev[k], hipEventDisableTiming) != hipSuccess) {
      for (int j = 0; j < k; ++j) (void)hipEventDestroy(s->stage_ev[j]);
      (void)hipHostFree(s->h_stage);
      delete s;
      set_error("session_create: hipEventCreate failed");
      return SMOLTTS_E_HIP;
    }
  }
  {  // frames the host may run ahead of the GPU: SMOLTTS_MAX_FRAMES_IN_FLIGHT (default 64; 0 = unbounded).  A profiler that
     // rewrites every dispatch into several packets (rocprofv3 --pmc) needs a small bound and one frame per graph: its
     // command line sets both variables explicitly (tools/collect_pmc.sh) -- the library itself never looks for a profiler.
    const char* lim = getenv("SMOLTTS_MAX_FRAMES_IN_FLIGHT");
    const int limit = lim ? atoi(lim) : 64;
    s->flight_group = limit > 0 ? (limit + 1) / 2 : 0;
    for (int k = 0; k < 2; ++k)
      if (hipEventCreateWithFlags(&s->flight_ev[k], hipEventDisableTiming) != hipSuccess) s->flight_group = 0;
    const char* fpg = getenv("SMOLTTS_FRAMES_PER_GRAPH");  // frames captured into one graph where that many remain to be launched
    s->multi_frames = fpg ? atoi(fpg) : 0;  // 0: follows the calls (smoltts_session_set_frames_per_graph, or min(n_frames, 4) of the largest call so far)
    if (s->multi_frames < 0 || s->multi_frames > 16) s->multi_frames = 1;
    s->multi_fixed = s->multi_env = fpg != nullptr;
    s->flight_limit = limit;
  }
  hipLaunchKernelGGL(init_state_kernel, dim3((max_batch + 63) / 64), dim3(64), 0, 0, max_batch, e->cfg.n_fast, s->iota,
                     s->fastpos, s->pos, s->frames, s->done, s->mask, s->margin, s->margin_at, s->cur_col, s->new_col, s->salt, s->attn_ticket);
  hipError_t err = hipGetLastError();
  if (err == hipSuccess) err = hipStreamSynchronize(0);
  if (err != hipSuccess) {
    for (int k = 0; k < STAGE_RING; ++k) (void)hipEventDestroy(s->stage_ev[k]);
    (void)hipHostFree(s->h_stage);
    delete s;
    set_error("session_create: init kernel failed: %s", hipGetErrorString(err));
    return SMOLTTS_E_HIP;
  }
  *out = s;
  return SMOLTTS_OK;
}

void smoltts_session_destroy(SmolttsSession* s) {
  if (!s) return;
  drop_graphs(s);
  for (int k = 0; k < STAGE_RING; ++k)
    if (s->stage_ev[k]) (void)hipEventDestroy(s->stage_ev[k]);
  if (s->h_stage) (void)hipHostFree(s->h_stage);
  for (int k = 0; k < 2; ++k)
    if (s->flight_ev[k]) (void)hipEventDestroy(s->flight_ev[k]);
  delete s;
}

int smoltts_lm_prefill(SmolttsSession* s, const int32_t* grid_dev, const int32_t* row_slot_dev, const int32_t* row_pos_dev,
                       int32_t n_rows, const int32_t* slots_host, const int32_t* last_row_host, int32_t n_slots,
                       int32_t stop_on_eos, void* stream) {
  ST_REQUIRE(s && grid_dev && row_slot_dev && row_pos_dev && slots_host && last_row_host, SMOLTTS_E_INVALID,
             "lm_prefill: null argument");
  ST_REQUIRE(n_rows > 0 && n_rows <= s->max_rows, SMOLTTS_E_CAPACITY, "lm_prefill: %d rows, session holds %d", n_rows, s->max_rows);
  ST_REQUIRE(n_slots > 0 && n_slots <= s->B, SMOLTTS_E_CAPACITY, "lm_prefill: %d slots, session holds %d", n_slots, s->B);
  for (int i = 0; i < n_slots; ++i) {
    ST_REQUIRE(slots_host[i] >= 0 && slots_host[i] < s->B, SMOLTTS_E_INVALID, "lm_prefill: slot %d out of range", slots_host[i]);
    ST_REQUIRE(last_row_host[i] >= 0 && last_row_host[i] < n_rows, SMOLTTS_E_INVALID, "lm_prefill: last_row %d out of range", last_row_host[i]);
  }
  hipStream_t st = (hipStream_t)stream;
  const SmolttsEngine* e = s->e;
  const SmolttsLMConfig& c = e->cfg;
  if (s->stop_on_eos != stop_on_eos) drop_graphs(s);  // the flag is baked into the captured commit nodes
  s->stop_on_eos = stop_on_eos;
  ST_TRY(stage_upload(s, slots_host, last_row_host, n_slots, st));
  hipLaunchKernelGGL(slot_reset_kernel, dim3((s->B + 63) / 64), dim3(64), 0, st, s->B, n_slots, s->stage_slots, s->stage_last,
                     row_pos_dev, s->pos, s->frames, s->done, s->mask, s->margin, s->salt);
  ST_CHECK_HIP(hipGetLastError());
  ST_TRY(embed_rows(s, grid_dev, n_rows, s->xr, st));
  ST_TRY(run_slow_layers(s, s->xr, s->qr, n_rows, row_pos_dev, row_slot_dev, /*publish_hidden=*/false, st));
  hipLaunchKernelGGL(scatter_last_kernel, dim3(n_slots), dim3(256), 0, st, s->xr, s->stage_slots, s->stage_last, c.dim, s->xt);
  ST_CHECK_HIP(hipGetLastError());
  {  // publish the slow hidden rows of all B slots (rows of slots not being started are masked at commit)
    const EmitArgs em = slow_hidden_emit(s);
    ST_TRY(launch_x3_pack(s->xt, c.dim, s->B, c.dim, em.x3a, em.gamma_a, em.x3b, em.gamma_b, em.ssq, st));
  }
  // frame 0: the tail is the same ~180 launches after every prefill -> replayed from a graph (host launch time is what
  // a single short prompt waits for)
  if (graphs_enabled()) {
    if (!s->tail_ready) {
      ST_TRY(capture_graph(st, &s->tail_exec, [&](hipStream_t cap) { return run_tail(s, /*advance_pos=*/0, cap); }));
      s->tail_ready = true;
    }
    ST_CHECK_HIP(hipGraphLaunch(s->tail_exec, st));
  } else {
    ST_TRY(run_tail(s, /*advance_pos=*/0, st));
  }
  s->prefilled = true;
  return SMOLTTS_OK;
}

int smoltts_lm_prefill_chunk(SmolttsSession* s, const int32_t* grid_dev, const int32_t* row_slot_dev, const int32_t* row_pos_dev,
                             int32_t n_rows, const int32_t* slots_host, const int32_t* last_row_host, int32_t n_slots, void* stream) {
  ST_REQUIRE(s && grid_dev && row_slot_dev && row_pos_dev && slots_host && last_row_host, SMOLTTS_E_INVALID,
             "lm_prefill_chunk: null argument");
  ST_REQUIRE(n_rows > 0 && n_rows <= s->max_rows, SMOLTTS_E_CAPACITY, "lm_prefill_chunk: %d rows, session holds %d", n_rows, s->max_rows);
  ST_REQUIRE(n_slots > 0 && n_slots <= s->B, SMOLTTS_E_CAPACITY, "lm_prefill_chunk: %d slots, session holds %d", n_slots, s->B);
  for (int i = 0; i < n_slots; ++i) {
    ST_REQUIRE(slots_host[i] >= 0 && slots_host[i] < s->B, SMOLTTS_E_INVALID, "lm_prefill_chunk: slot %d out of range", slots_host[i]);
    ST_REQUIRE(last_row_host[i] >= 0 && last_row_host[i] < n_rows, SMOLTTS_E_INVALID, "lm_prefill_chunk: last_row %d out of range", last_row_host[i]);
  }
  hipStream_t st = (hipStream_t)stream;
  ST_TRY(stage_upload(s, slots_host, last_row_host, n_slots, st));
  hipLaunchKernelGGL(slot_park_kernel, dim3((s->B + 63) / 64), dim3(64), 0, st, s->B, n_slots, s->stage_slots, s->stage_last, row_pos_dev,
                     s->pos, s->done, s->mask);
  ST_CHECK_HIP(hipGetLastError());
  ST_TRY(embed_rows(s, grid_dev, n_rows, s->xr, st));
  ST_TRY(run_slow_layers(s, s->xr, s->qr, n_rows, row_pos_dev, row_slot_dev, /*publish_hidden=*/false, st));
  return launch_commit_embed(s, /*do_commit=*/0, 0, st);  // the prompt rows went through x3n / ssq: re-publish the decode rows
}

int smoltts_lm_prefill_deferred(SmolttsSession* s, const int32_t* grid_dev, const int32_t* row_slot_dev, const int32_t* row_pos_dev,
                                int32_t n_rows, const int32_t* slots_host, const int32_t* last_row_host, int32_t n_slots,
                                int32_t stop_on_eos, void* stream) {
  ST_REQUIRE(s && grid_dev && row_slot_dev && row_pos_dev && slots_host && last_row_host, SMOLTTS_E_INVALID,
             "lm_prefill_deferred: null argument");
  ST_REQUIRE(n_rows > 0 && n_rows <= s->max_rows, SMOLTTS_E_CAPACITY, "lm_prefill_deferred: %d rows, session holds %d", n_rows, s->max_rows);
  ST_REQUIRE(n_slots > 0 && n_slots <= s->B, SMOLTTS_E_CAPACITY, "lm_prefill_deferred: %d slots, session holds %d", n_slots, s->B);
  for (int i = 0; i < n_slots; ++i) {
    ST_REQUIRE(slots_host[i] >= 0 && slots_host[i] < s->B, SMOLTTS_E_INVALID, "lm_prefill_deferred: slot %d out of range", slots_host[i]);
    ST_REQUIRE(last_row_host[i] >= 0 && last_row_host[i] < n_rows, SMOLTTS_E_INVALID, "lm_prefill_deferred: last_row %d out of range", last_row_host[i]);
  }
  hipStream_t st = (hipStream_t)stream;
  if (s->stop_on_eos != stop_on_eos) drop_graphs(s);  // the flag is baked into the captured commit nodes
  s->stop_on_eos = stop_on_eos;
  ST_TRY(stage_upload(s, slots_host, last_row_host, n_slots, st));
  // KV rows of the whole prompt (the last column's row is recomputed by the first decode frame, identically)
  ST_TRY(embed_rows(s, grid_dev, n_rows, s->xr, st));
  ST_TRY(run_slow_layers(s, s->xr, s->qr, n_rows, row_pos_dev, row_slot_dev, /*publish_hidden=*/false, st));
  hipLaunchKernelGGL(slot_start_kernel, dim3((s->B + 63) / 64), dim3(64), 0, st, s->B, 1 + s->e->cfg.n_fast, n_slots, s->stage_slots,
                     s->stage_last, row_pos_dev, grid_dev, s->pos, s->frames, s->done, s->margin, s->cur_col, s->salt);
  ST_CHECK_HIP(hipGetLastError());
  ST_TRY(launch_commit_embed(s, /*do_commit=*/0, 0, st));  // mask = !done; embed every slot's (new) current column
  s->prefilled = true;
  return SMOLTTS_OK;
}

// ---- prompt prefill beside the decode frames (serving: the refill of a slot no longer stops the other slots' ticks)
//   tick stream:   ... tick k-1 | park(new slots) | tick k ............ | start(new slots) | tick k+1 (their frame 0) ...
//   side stream:                     (host saw the park done) prefill_side(new prompts)  ^ host saw it done
// The side call shares nothing with the frames but the KV cache, and of that only rows 0 .. T-2 of the parked slots: their idle
// decode rows write at the parking position T-1 (rewritten by the tenant's first frame) and read garbage nobody keeps.
int smoltts_lm_park_slots(SmolttsSession* s, const int32_t* slots_host, const int32_t* pos_host, int32_t n_slots, void* stream) {
  ST_REQUIRE(s && slots_host && pos_host, SMOLTTS_E_INVALID, "lm_park_slots: null argument");
  ST_REQUIRE(n_slots > 0 && n_slots <= s->B, SMOLTTS_E_CAPACITY, "lm_park_slots: %d slots, session holds %d", n_slots, s->B);
  for (int i = 0; i < n_slots; ++i) {
    ST_REQUIRE(slots_host[i] >= 0 && slots_host[i] < s->B, SMOLTTS_E_INVALID, "lm_park_slots: slot %d out of range", slots_host[i]);
    ST_REQUIRE(pos_host[i] >= 0 && pos_host[i] < s->max_seq, SMOLTTS_E_INVALID, "lm_park_slots: position %d outside the cache (%d)", pos_host[i], s->max_seq);
  }
  hipStream_t st = (hipStream_t)stream;
  ST_TRY(stage_upload(s, slots_host, pos_host, n_slots, st));
  hipLaunchKernelGGL(slot_parkpos_kernel, dim3((s->B + 63) / 64), dim3(64), 0, st, s->B, n_slots, s->stage_slots, s->stage_last, s->pos, s->done, s->mask);
  ST_CHECK_HIP(hipGetLastError());
  return launch_commit_embed(s, /*do_commit=*/0, 0, st);  // mask = !done for the next frame
}

int smoltts_lm_prefill_side(SmolttsSession* s, const int32_t* grid_dev, const int32_t* row_slot_dev, const int32_t* row_pos_dev,
                            int32_t n_rows, void* stream) {
  ST_REQUIRE(s && grid_dev && row_slot_dev && row_pos_dev, SMOLTTS_E_INVALID, "lm_prefill_side: null argument");
  ST_REQUIRE(n_rows > 0 && n_rows <= s->max_rows, SMOLTTS_E_CAPACITY, "lm_prefill_side: %d rows, session holds %d", n_rows, s->max_rows);
  SmolttsSession side = *s;  // the same engine, caches and options; every activation buffer of the slow layers swapped for the side set
  side.xr = s->sd_xr; side.qr = s->sd_qr; side.x3n = s->sd_x3n; side.x3a = s->sd_x3a; side.x3h = s->sd_x3h; side.ssq = s->sd_ssq;
  side.split_attn = false;  // (the key-split attention's records and tickets belong to the frames)
  side.dup_code = -1;
  hipStream_t st = (hipStream_t)stream;
  ST_TRY(embed_rows(&side, grid_dev, n_rows, side.xr, st));
  return run_slow_layers(&side, side.xr, side.qr, n_rows, row_pos_dev, row_slot_dev, /*publish_hidden=*/false, st);
}

int smoltts_lm_start_slots(SmolttsSession* s, const int32_t* grid_dev, const int32_t* row_pos_dev, const int32_t* slots_host,
                           const int32_t* last_row_host, int32_t n_slots, int32_t stop_on_eos, void* stream) {
  ST_REQUIRE(s && grid_dev && row_pos_dev && slots_host && last_row_host, SMOLTTS_E_INVALID, "lm_start_slots: null argument");
  ST_REQUIRE(n_slots > 0 && n_slots <= s->B, SMOLTTS_E_CAPACITY, "lm_start_slots: %d slots, session holds %d", n_slots, s->B);
  for (int i = 0; i < n_slots; ++i) {
    ST_REQUIRE(slots_host[i] >= 0 && slots_host[i] < s->B, SMOLTTS_E_INVALID, "lm_start_slots: slot %d out of range", slots_host[i]);
    ST_REQUIRE(last_row_host[i] >= 0 && last_row_host[i] < s->max_rows, SMOLTTS_E_INVALID, "lm_start_slots: last_row %d out of range", last_row_host[i]);
  }
  hipStream_t st = (hipStream_t)stream;
  if (s->stop_on_eos != stop_on_eos) drop_graphs(s);  // the flag is baked into the captured commit nodes
  s->stop_on_eos = stop_on_eos;
  ST_TRY(stage_upload(s, slots_host, last_row_host, n_slots, st));
  hipLaunchKernelGGL(slot_start_kernel, dim3((s->B + 63) / 64), dim3(64), 0, st, s->B, 1 + s->e->cfg.n_fast, n_slots, s->stage_slots,
                     s->stage_last, row_pos_dev, grid_dev, s->pos, s->frames, s->done, s->margin, s->cur_col, s->salt);
  ST_CHECK_HIP(hipGetLastError());
  ST_TRY(launch_commit_embed(s, /*do_commit=*/0, 0, st));  // mask = !done; embed every slot's (new) current column
  s->prefilled = true;
  return SMOLTTS_OK;
}

int smoltts_lm_decode(SmolttsSession* s, int32_t n_frames, void* stream) {
  ST_REQUIRE(s, SMOLTTS_E_INVALID, "lm_decode: null session");
  ST_REQUIRE(s->prefilled, SMOLTTS_E_STATE, "lm_decode: call smoltts_lm_prefill first");
  ST_REQUIRE(n_frames >= 0, SMOLTTS_E_INVALID, "lm_decode: n_frames < 0");
  hipStream_t st = (hipStream_t)stream;
  if (!graphs_enabled()) {
    for (int f = 0; f < n_frames; ++f) ST_TRY(run_decode_frame(s, st));
    return SMOLTTS_OK;
  }
  if (!s->graph_ready) {
    ST_TRY(capture_graph(st, &s->graph_exec, [&](hipStream_t cap) { return run_decode_frame(s, cap); }));
    s->graph_ready = true;
  }
  // Bounded run-ahead: a frame is ~230 AQL packets, and nothing stops a caller from queueing hundreds of frames.  The
  // hardware queue holds 16K packets; the runtime copes with a full ring, but a profiler that rewrites every dispatch into
  // several packets (rocprofv3 --pmc: counter start / stop + serialisation barriers around each kernel) overflows its
  // intercept queue and the run hangs behind the queued graph launches (round 1: gpurun_out/probe_d.log stops at the
  // first un-synchronised run of 63 frame graphs, while 3 frames per synchronisation completed).  So the host never
  // runs more than SMOLTTS_MAX_FRAMES_IN_FLIGHT frames ahead: it records an event every half limit and, before queueing
  // more, waits for the event of two groups ago -- the GPU queue never drains, the host merely stops piling up packets.
  // Frames are launched `multi_frames` at a time where that many remain: every graph launch costs the GPU a gap between the
  // last node of one graph and the first of the next (measured: the frame rate of a 32-frame chunk rises by the gaps saved).
  if (!s->multi_fixed && n_frames >= 2) {  // follows the calls: a later, longer call (a tick after a 2-frame warm-up) re-captures
    // (4 since round 4: with 181 launches per frame graphs of 1 / 2 / 4 / 8 / 16 frames give 27.5 / 27.58 / 27.61 / 27.47 / 27.25k
    // frames/s in the bench, profiles/r04_ab_frames_per_graph.txt; round 3's 216-launch frame was best at 8)
    const int want = n_frames < 4 ? n_frames : 4;
    if (want > s->multi_frames) {
      s->multi_frames = want;
      if (s->multi_ready) { (void)hipGraphExecDestroy(s->multi_exec); s->multi_ready = false; }
    }
  }
  ST_TRY(ensure_multi_graph(s, n_frames, st));
  const int mf = s->multi_ready ? s->multi_frames : 1;
  for (int f = 0; f < n_frames;) {
    if (s->flight_group > 0 && s->flight_count >= s->flight_group) {
      const int k = s->flight_cur;
      ST_CHECK_HIP(hipEventRecord(s->flight_ev[k], st));
      s->flight_live[k] = true;
      if (s->flight_live[k ^ 1]) ST_CHECK_HIP(hipEventSynchronize(s->flight_ev[k ^ 1]));
      s->flight_cur = k ^ 1;
      s->flight_count = 0;
    }
    if (mf > 1 && n_frames - f >= mf) {
      ST_CHECK_HIP(hipGraphLaunch(s->multi_exec, st));
      s->flight_count += mf;
      f += mf;
    } else {
      ST_CHECK_HIP(hipGraphLaunch(s->graph_exec, st));
      s->flight_count++;
      f++;
    }
  }
  return SMOLTTS_OK;
}

// Sampling mode of the session (reference GenerationSettings, lm/generate.py:12-16): temp / fast_temp <= 0
// select greedy argmax for the slow / depth tokens; min_p <= 0 disables the filter.  Takes effect from
// the next frame (the captured graph is dropped).
int smoltts_session_set_sampling(SmolttsSession* s, float temp, float fast_temp, float min_p, uint64_t seed) {
  ST_REQUIRE(s, SMOLTTS_E_INVALID, "session_set_sampling: null session");
  ST_REQUIRE(min_p < 1.0f, SMOLTTS_E_INVALID, "session_set_sampling: min_p must be < 1");
  s->temp = temp; s->fast_temp = fast_temp; s->min_p = min_p; s->seed = seed;
  drop_graphs(s);
  return SMOLTTS_OK;
}

// Measurement aid of this session only: while code >= 0, every launch of that kernel class inside this session's frames
// (GEMM epilogue code, N == n_filter when n_filter > 0; 100 / 101 = depth / slow attention) is issued twice; the captured
// graphs are dropped.  (t_dup - t_base) / extra launches is the kernel's duration in situ.
int smoltts_session_measure_duplicate(SmolttsSession* s, int32_t code, int32_t n_filter) {
  ST_REQUIRE(s, SMOLTTS_E_INVALID, "session_measure_duplicate: null session");
  ST_REQUIRE(code != SMOLTTS_EPI_RESID, SMOLTTS_E_INVALID, "session_measure_duplicate: EPI_RESID launches are not idempotent");
  s->dup_code = code; s->dup_n = n_filter;
  drop_graphs(s);
  return SMOLTTS_OK;
}

// Launch-structure options of a session (both default on; switching one drops the captured graphs): the ids are the same
// either way -- the switches exist for A/B runs and tests.
int smoltts_session_set_option(SmolttsSession* s, int32_t option, int32_t value) {
  ST_REQUIRE(s, SMOLTTS_E_INVALID, "session_set_option: null session");
  switch (option) {
    case SMOLTTS_OPT_QKV_TABLE: s->use_qkv_table = value != 0; break;
    case SMOLTTS_OPT_COMMIT_PICKS: s->commit_picks = value != 0; break;
    case SMOLTTS_OPT_SPLIT_ATTN: s->split_attn = value != 0; break;
    case SMOLTTS_OPT_STREAM_W: s->stream_w = value; break;
    case SMOLTTS_OPT_FUSE_DEPTH_ATTN: s->fuse_depth_attn = value != 0; break;
    case SMOLTTS_OPT_FUSE_PICK: s->fuse_pick = value != 0; break;
    case SMOLTTS_OPT_FP8_PREFILL: s->fp8_prefill = value != 0; break;
    default:
      set_error("session_set_option: unknown option %d", option);
      return SMOLTTS_E_INVALID;
  }
  drop_graphs(s);
  return SMOLTTS_OK;
}

// Forget the captured frame graph (the next smoltts_lm_decode captures it again).
int smoltts_session_drop_graph(SmolttsSession* s) {
  ST_REQUIRE(s, SMOLTTS_E_INVALID, "session_drop_graph: null session");
  drop_graphs(s);
  return SMOLTTS_OK;
}

// Frames per multi-frame graph of this session (1 = single-frame graphs only, 0 = follow the decode calls again); with a
// stream and after a prefill, the graphs are captured here (a warm-up step) instead of inside the first decode call.
int smoltts_session_set_frames_per_graph(SmolttsSession* s, int32_t n, void* stream) {
  ST_REQUIRE(s, SMOLTTS_E_INVALID, "session_set_frames_per_graph: null session");
  ST_REQUIRE(n >= 0 && n <= 16, SMOLTTS_E_INVALID, "session_set_frames_per_graph: %d frames per graph (0..16)", n);
  if (s->multi_env) n = s->multi_frames;  // SMOLTTS_FRAMES_PER_GRAPH wins (a profiler run pins one frame per graph)
  if (n != s->multi_frames && s->multi_ready) { (void)hipGraphExecDestroy(s->multi_exec); s->multi_ready = false; }
  s->multi_frames = n;
  s->multi_fixed = n > 0;
  if (n > 0 && s->prefilled && graphs_enabled()) {
    hipStream_t st = (hipStream_t)stream;
    if (!s->graph_ready) {
      ST_TRY(capture_graph(st, &s->graph_exec, [&](hipStream_t cap) { return run_decode_frame(s, cap); }));
      s->graph_ready = true;
    }
    ST_TRY(ensure_multi_graph(s, n, st));
  }
  return SMOLTTS_OK;
}

int smoltts_session_outputs(SmolttsSession* s, int32_t** codes_dev, int32_t** n_frames_dev, int32_t** done_dev,
                            float** margin_dev) {
  ST_REQUIRE(s, SMOLTTS_E_INVALID, "session_outputs: null session");
  if (codes_dev) *codes_dev = s->codes;
  if (n_frames_dev) *n_frames_dev = s->frames;
  if (done_dev) *done_dev = s->done;
  if (margin_dev) *margin_dev = s->margin;
  return SMOLTTS_OK;
}

int smoltts_session_kv_cache(SmolttsSession* s, void** k_dev, void** v_dev, uint64_t* layer_bytes) {
  ST_REQUIRE(s && k_dev && v_dev && layer_bytes, SMOLTTS_E_INVALID, "session_kv_cache: null argument");
  const SmolttsLMConfig& c = s->e->cfg;
  *k_dev = s->kc; *v_dev = s->vc;
  *layer_bytes = (uint64_t)s->B * c.n_kv_head * s->max_seq * 64 * (s->kv_format == SMOLTTS_KV_BF16 ? 2 : 4);
  return SMOLTTS_OK;
}

int smoltts_session_margin_at(SmolttsSession* s, int32_t** margin_at_dev) {
  ST_REQUIRE(s && margin_at_dev, SMOLTTS_E_INVALID, "session_margin_at: null argument");
  *margin_at_dev = s->margin_at;
  return SMOLTTS_OK;
}

}  // extern "C"
