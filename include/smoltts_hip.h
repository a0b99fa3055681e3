/*
 * libsmoltts_hip.so — C ABI of the MI355X (gfx950) DualAR + Mimi decode hot path.
 *
 * The reference (EndlessReform/smoltts) has no FFI: its hot path is Python objects calling MLX
 * ops.  The seams this ABI replaces are (paths relative to the reference tree):
 *
 *   smoltts_lm_prefill / smoltts_lm_decode     <- RQTransformer.forward_generate + the 8x
 *       forward_generate_fast loop + argmax, i.e. one SingleBatchGenerator.__next__
 *       (mlx_inference/src/smoltts_mlx/lm/rq_transformer.py:173-220, lm/generate.py:59-171),
 *       generalised from batch 1 to B independent utterance slots
 *   smoltts_mimi_decode / smoltts_mimi_decode_step <- MimiModel.decode / decode_step
 *       (mlx_inference/src/smoltts_mlx/codec/mimi.py:88-104)
 *   smoltts_k_*                                 <- single operators (RMSNorm+Linear, RoPE, SDPA,
 *       SwiGLU, embed, argmax, conv-as-GEMM, LayerNorm) exported for unit parity tests
 *
 * Conventions: plain C types only; every pointer named *_dev is a device (HBM) address owned by
 * the caller (e.g. torch.Tensor.data_ptr()); the library never allocates or frees caller memory
 * and keeps all per-session state inside the caller-allocated slab, so one engine can serve many
 * sessions and one host thread per stream can drive it.  `stream` is a hipStream_t passed as
 * void* (NULL = default stream).  Every function returns 0 on success or a negative SMOLTTS_E_*
 * code; smoltts_last_error() returns a thread-local message.  No exceptions cross the boundary.
 */
#ifndef SMOLTTS_HIP_H
#define SMOLTTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMOLTTS_ABI_VERSION 6

enum {
  SMOLTTS_OK = 0,
  SMOLTTS_E_INVALID = -1,   /* bad argument / unsupported shape */
  SMOLTTS_E_HIP = -2,       /* a HIP runtime call failed */
  SMOLTTS_E_STATE = -3,     /* call sequence error (e.g. decode before prefill) */
  SMOLTTS_E_CAPACITY = -4   /* slab / sequence / batch capacity exceeded */
};

const char* smoltts_last_error(void);
int smoltts_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Weight tile format ("T16x32"): a row-major [N][K] matrix is stored as tiles of 16 rows x 32
 * columns, tile (nt, kc) at element offset (nt * K/32 + kc) * 512, inside a tile in fp32-MFMA
 * fragment order: lane l = 16*q + r (r = row in tile, q = 0..3) owns the 8 consecutive
 * elements [r][8q .. 8q+8).  bf16 tiles: 16 bytes per lane, lane-linear (one 1 KiB wave load).
 * fp32 tiles: two 1 KiB halves, half h holds elements [r][8q+4h .. 8q+4h+4) of every lane.
 * N is padded to a multiple of 16 with zero rows; K must be a multiple of 32.
 * smoltts_amd/packing.py produces it.
 * ------------------------------------------------------------------------------------------- */

/* "W3" tiles (Mimi many-row GEMMs, smoltts_amd/csrc/gemm_b3.hip): an fp32 [N][K] matrix as three bf16 pieces hi + mid + lo
 * (== the fp32 value exactly); tile (nt, kc) is 3 KiB at (nt * K/32 + kc) * 3072: piece p at + p * 1024 in the bf16 T16x32
 * lane order above.  smoltts_amd/packing.py (tile_w3) produces it. */

/* Byte offsets into the LM weight arena. */
typedef struct SmolttsBlockWeights {
  uint64_t attn_norm;   /* fp32 [dim] */
  uint64_t wqkv;        /* bf16 T16x32 [(H+2KV)*64][dim], rows q|k|v */
  uint64_t wo;          /* bf16 T16x32 [dim][dim] */
  uint64_t ffn_norm;    /* fp32 [dim] */
  uint64_t w13;         /* bf16 T16x32 [2*inter][dim], row 2i = w1 row i, row 2i+1 = w3 row i */
  uint64_t w2;          /* bf16 T16x32 [dim][inter] */
} SmolttsBlockWeights;

#define SMOLTTS_MAX_LAYERS 64
#define SMOLTTS_MAX_FAST_LAYERS 16

typedef struct SmolttsLMConfig {
  int32_t dim, n_layer, n_head, n_kv_head, inter;
  int32_t fast_dim, n_fast_layer, fast_n_head, fast_n_kv_head, fast_inter;
  int32_t vocab_size, codebook_size, num_codebooks;
  int32_t n_fast;             /* fast steps per frame = num_codebooks - (duplicate_code_0 ? 0 : 1) */
  int32_t duplicate_code_0;
  int32_t depthwise_wte;
  int32_t has_fast_project_in;/* Linear(dim, fast_dim) with bias between slow hidden and fast input */
  int32_t embed_mask_mode;    /* 0: zero codebook-embedding sum where code0 == 0 (torch);
                                 1: zero it where the text token is outside [semantic_start, semantic_end] (MLX) */
  int32_t semantic_start_id, semantic_end_id, im_end_id;
  int32_t max_seq_len;        /* rows of the slow RoPE table */
  float norm_eps;
  int32_t weight_format;      /* SMOLTTS_W_BF16 (0) | SMOLTTS_W_FP8 (1): format of every "T16x32" Linear below.
                                 fp8: each matrix is its e4m3 tiles (rows * K bytes) followed by fp32 row scales
                                 [rows]; the depthwise head is one such block per step (stride rows * (K + 4) bytes) */
} SmolttsLMConfig;

typedef struct SmolttsLMWeights {
  uint64_t text_emb;      /* bf16 row-major [vocab][dim] */
  uint64_t codebook_emb;  /* bf16 row-major [codebook_size*num_codebooks][dim] */
  uint64_t fast_emb;      /* bf16 row-major [fast rows][fast_dim] */
  uint64_t norm;          /* fp32 [dim] */
  uint64_t head;          /* bf16 T16x32 [vocab][dim] (tied: same values as text_emb) */
  uint64_t fast_norm;     /* fp32 [fast_dim] */
  uint64_t fast_head;     /* bf16 T16x32 [n_fast*codebook_size][fast_dim]; step i uses rows
                             [i*codebook_size, (i+1)*codebook_size) (all steps share rows [0, cs)
                             when the checkpoint has no depthwise output) */
  uint64_t fast_head_step_stride; /* rows between consecutive steps: codebook_size or 0 */
  uint64_t fast_proj_w;   /* bf16 T16x32 [fast_dim][dim], only if has_fast_project_in */
  uint64_t fast_proj_b;   /* fp32 [fast_dim] */
  uint64_t rope;          /* fp32 [max_seq_len][32][2] (cos, sin) */
  uint64_t fast_rope;     /* fp32 [n_fast][32][2] */
  SmolttsBlockWeights layers[SMOLTTS_MAX_LAYERS];
  SmolttsBlockWeights fast_layers[SMOLTTS_MAX_FAST_LAYERS];
} SmolttsLMWeights;

enum {  /* storage of the slow transformer's KV cache */
  SMOLTTS_KV_F32 = 0,   /* fp32 (default): K/V exactly as computed */
  SMOLTTS_KV_BF16 = 1   /* bf16: K (after RoPE) and V rounded to nearest even once, when written; half the attention stream.
                           Changes the arithmetic (the oracle mirrors it with kv_bf16=True); the reference keeps K/V in its
                           activation dtype (lm/cache.py:6-22), i.e. bf16 for a bf16 checkpoint */
};

typedef struct SmolttsEngine SmolttsEngine;
typedef struct SmolttsSession SmolttsSession;

/* Engine = immutable model (config + weight arena pointers). The arena must outlive the engine. */
int smoltts_engine_create(const SmolttsLMConfig* cfg, const SmolttsLMWeights* offsets,
                          const void* arena_dev, size_t arena_bytes, SmolttsEngine** out);
void smoltts_engine_destroy(SmolttsEngine* e);

/* Derived table of an engine (optional, caller-owned memory like everything else): row e of the depth transformer's
 * embedding table (fast_embeddings, lm/generate.py:134-140) always enters depth layer 0 as RMSNorm(E[e]) -> wqkv, so its
 * q | k | v (before RoPE) can be looked up instead of computed: [rows][(fast heads + 2 fast kv heads) * 64] fp32 (150m: 73 MB).
 * With the table in place a frame has 7 launches fewer: the kernel that picks a depth code also gathers the next step's
 * layer-0 q / k / v (RoPE for its position applied on the way) and that step starts with its attention.  The values are the
 * decode path's own (same GEMM kernel, same accumulation order).
 *   smoltts_engine_fast_qkv_bytes   slab size (table + build scratch); 0 if the model has no depth step with a successor
 *   smoltts_engine_build_fast_qkv   fills the table (synchronises `stream`); the slab must outlive the engine's sessions.
 * Sessions use the table from their next captured frame on; smoltts_session_set_option(s, SMOLTTS_OPT_QKV_TABLE, 0) keeps one on
 * the GEMM path. */
size_t smoltts_engine_fast_qkv_bytes(const SmolttsEngine* e);
int smoltts_engine_build_fast_qkv(SmolttsEngine* e, void* slab_dev, size_t slab_bytes, void* stream);

/* Session = B utterance slots with KV caches, all inside one caller-allocated device slab.
 * max_rows bounds the prompt rows one smoltts_lm_prefill call may carry; max_frames bounds the
 * frames a slot may emit (output ring). */
size_t smoltts_session_slab_bytes(const SmolttsEngine* e, int32_t max_batch, int32_t max_seq,
                                  int32_t max_rows, int32_t max_frames);
int smoltts_session_create(SmolttsEngine* e, void* slab_dev, size_t slab_bytes, int32_t max_batch,
                           int32_t max_seq, int32_t max_rows, int32_t max_frames,
                           SmolttsSession** out);
/* The same with a choice of KV-cache storage (SMOLTTS_KV_*); the plain forms above are kv_format = SMOLTTS_KV_F32. */
size_t smoltts_session_slab_bytes_kv(const SmolttsEngine* e, int32_t max_batch, int32_t max_seq,
                                     int32_t max_rows, int32_t max_frames, int32_t kv_format);
int smoltts_session_create_kv(SmolttsEngine* e, void* slab_dev, size_t slab_bytes, int32_t max_batch,
                              int32_t max_seq, int32_t max_rows, int32_t max_frames, int32_t kv_format,
                              SmolttsSession** out);
void smoltts_session_destroy(SmolttsSession* s);

/* Prefill: run the slow transformer over n_rows prompt tokens (packed over utterances) and emit
 * frame 0 of every listed slot (slow head + n_fast depth steps, greedy).
 *   grid_dev      int32 [n_rows][1 + n_fast]   prompt columns (text id, code rows)
 *   row_slot_dev  int32 [n_rows]               slot of each row
 *   row_pos_dev   int32 [n_rows]               position of each row inside its utterance (0-based)
 *   slots_host    int32 [n_slots]              slots being (re)started (host memory)
 *   last_row_host int32 [n_slots]              index of each slot's last prompt row (host memory)
 * Slot state (position, frame counter, done flag) is reset for the listed slots; other slots are
 * untouched.  Asynchronous on `stream`. */
int smoltts_lm_prefill(SmolttsSession* s, const int32_t* grid_dev, const int32_t* row_slot_dev,
                       const int32_t* row_pos_dev, int32_t n_rows, const int32_t* slots_host,
                       const int32_t* last_row_host, int32_t n_slots, int32_t stop_on_eos,
                       void* stream);

/* Chunked prompt prefill (BASELINE config 5): run the slow transformer over n_rows further prompt rows of
 * idle slots, filling their KV rows only — no frame is emitted and the listed slots stay idle, so decode
 * calls for the other slots may run between chunks.  Rows attend to their slot's earlier positions:
 * chunks of a slot must arrive in position order, and its last chunk goes through smoltts_lm_prefill
 * (row_pos continuing where the chunks stopped), which emits frame 0.  Arguments as smoltts_lm_prefill. */
int smoltts_lm_prefill_chunk(SmolttsSession* s, const int32_t* grid_dev, const int32_t* row_slot_dev,
                             const int32_t* row_pos_dev, int32_t n_rows, const int32_t* slots_host,
                             const int32_t* last_row_host, int32_t n_slots, void* stream);

/* Prefill without the frame-0 tail (continuous batching: a decode call follows anyway): fills the KV rows of the prompt
 * and arms the listed slots so that the *next decode frame* takes the last prompt column as its input and emits the
 * slot's frame 0.  Same arguments as smoltts_lm_prefill; the ids produced are the same. */
int smoltts_lm_prefill_deferred(SmolttsSession* s, const int32_t* grid_dev, const int32_t* row_slot_dev,
                                const int32_t* row_pos_dev, int32_t n_rows, const int32_t* slots_host,
                                const int32_t* last_row_host, int32_t n_slots, int32_t stop_on_eos,
                                void* stream);

/* Prompt prefill BESIDE the decode frames of the same session (ABI 5; serving: refilling a slot no longer stops the other slots).
 * Three calls replace smoltts_lm_prefill_deferred:
 *   smoltts_lm_park_slots   (frame stream) the listed slots -- idle, their previous tenants gone -- are frozen with their position
 *                           counter at pos_host[i] = the new prompt's last position T-1: the rows their idle decode steps write
 *                           land there, where the tenant's first frame writes anyway;
 *   smoltts_lm_prefill_side (any stream, once the park has RUN: wait for it on the host or by event) KV-cache rows of the prompt
 *                           rows, computed on a workspace of its own -- it shares nothing with concurrently running frames but the
 *                           KV rows 0 .. T-2 of the parked slots;
 *   smoltts_lm_start_slots  (frame stream, once the side call has finished) arms the slots exactly as the deferred prefill does:
 *                           the next decode frame takes the last prompt column as its input and emits frame 0.
 * Same arguments as smoltts_lm_prefill_deferred, split over the calls; grid_dev / row_pos_dev must stay valid until the start call
 * has run.  Ids are those of the in-line prefill (same kernels, same order; tests/test_side_prefill_gpu.py). */
int smoltts_lm_park_slots(SmolttsSession* s, const int32_t* slots_host, const int32_t* pos_host, int32_t n_slots, void* stream);
int smoltts_lm_prefill_side(SmolttsSession* s, const int32_t* grid_dev, const int32_t* row_slot_dev, const int32_t* row_pos_dev,
                            int32_t n_rows, void* stream);
int smoltts_lm_start_slots(SmolttsSession* s, const int32_t* grid_dev, const int32_t* row_pos_dev, const int32_t* slots_host,
                           const int32_t* last_row_host, int32_t n_slots, int32_t stop_on_eos, void* stream);

/* Decode n_frames further frames for every slot of the session: each frame feeds the previous
 * column back (slow step at the slot's next position), then slow head + n_fast depth steps, all
 * greedy and on device; the frame loop is a captured hipGraph replayed n_frames times.  Slots that
 * are done (emitted <|im_end|> with stop_on_eos, reached max_frames, or filled their context: the next
 * token would land at position max_seq) are frozen.
 *
 * Host behaviour: the call queues graph launches on `stream` and returns, EXCEPT that the host never runs more than
 * SMOLTTS_MAX_FRAMES_IN_FLIGHT frames (environment, read when the session is created; default 64, 0 = unbounded) ahead of
 * the GPU: every half bound it records an event and, before queueing further, waits (hipEventSynchronize, blocking the
 * calling thread) for the event of two groups ago.  One host thread that drives several sessions therefore overlaps them
 * only up to that many frames each.  Frames go out `frames per graph` at a time where that many remain (see
 * smoltts_session_set_frames_per_graph; SMOLTTS_FRAMES_PER_GRAPH overrides).  The library never inspects a profiler's
 * environment: a counter-collecting profiler run sets SMOLTTS_MAX_FRAMES_IN_FLIGHT=2 SMOLTTS_FRAMES_PER_GRAPH=1 itself
 * (tools/collect_pmc.sh). */
int smoltts_lm_decode(SmolttsSession* s, int32_t n_frames, void* stream);

/* Frames captured into one multi-frame graph of this session: n = 1 single-frame graphs only, 2..16 that many, 0 = follow
 * the decode calls (min(n_frames, 4) of the largest call so far -- the default).  Called after a prefill with n > 0 it also
 * captures the graphs now (on `stream`), so that the first decode call of a request does not pay the capture. */
int smoltts_session_set_frames_per_graph(SmolttsSession* s, int32_t n, void* stream);
/* Launch-structure options of a session (A/B runs, tests): the ids produced are the same either way; the captured graphs are dropped.
 *   SMOLTTS_OPT_QKV_TABLE     depth layer-0 q | k | v out of the engine's fast_qkv table where it has been built (default 1)
 *   SMOLTTS_OPT_COMMIT_PICKS  the frame's slow token and last depth code picked inside the commit kernel instead of in
 *                             launches of their own (default 1)
 *   SMOLTTS_OPT_SPLIT_ATTN    decode attention over the slow cache with the keys of a (row, kv head) pair on two workgroups,
 *                             merged in part order inside the launch (<= 128 pairs, i.e. B <= 32 at 4 kv heads; default 1).
 *                             fp32 sums in a different order than the one-workgroup kernel: same ids except at near-ties
 *   SMOLTTS_OPT_FUSE_DEPTH_ATTN  (ABI 5) depth steps 1..: the attention over the <= 8-entry cache is worked out by the wo launch
 *                             itself (SmolttsGemm3Args.attn_q_dev) instead of by a launch of its own (default 1; applies where
 *                             smoltts_gemm3_attn_fusable); every sum formed in the stand-alone kernels' order: bit-identical to them */
#define SMOLTTS_OPT_QKV_TABLE 1
#define SMOLTTS_OPT_COMMIT_PICKS 2
#define SMOLTTS_OPT_SPLIT_ATTN 3
#define SMOLTTS_OPT_FUSE_DEPTH_ATTN 5
#define SMOLTTS_OPT_FP8_PREFILL 7  /* (ABI 5; default 0) engines with fp8 weights: prompt prefills of >= 256 rows run their GEMMs with fp8
                                      activations on the fp8 MFMA (SmolttsGemm3Args.fp8_activations): faster first chunk, prompt KV rows
                                      approximate -- ids are then no longer guaranteed to equal the reference greedy decode */
#define SMOLTTS_OPT_FUSE_PICK 6  /* (ABI 5) greedy depth codes picked inside the next step's layer-0 attention + wo launch (SmolttsPickArgs)
                                    instead of by a launch of their own (default 1; needs the fast_qkv table, the fused depth
                                    attention and greedy depth tokens); same ids, same gap records */
#define SMOLTTS_OPT_STREAM_W 4  /* value = mask of SMOLTTS_STREAM_W_*: which weights of a decode frame are loaded with the non-temporal
                                  hint (read once per frame: keeping them out of the caches leaves room for the depth layers'
                                  weights, which are re-read for each of the 8 depth steps) */
#define SMOLTTS_STREAM_W_SLOW 1        /* the slow transformer's blocks */
#define SMOLTTS_STREAM_W_SLOW_HEAD 2   /* the slow head */
#define SMOLTTS_STREAM_W_DEPTH_HEAD 4  /* the depth head slices (one per step) */
#define SMOLTTS_STREAM_W_DEPTH_W13 8   /* experiments: parts of the depth blocks */
#define SMOLTTS_STREAM_W_DEPTH_W2 16
#define SMOLTTS_STREAM_W_DEPTH_QKVO 32
#define SMOLTTS_STREAM_W_SLOW_ONLY_W13 64   /* with _SLOW: only the slow blocks' w1|w3 (the rest of the slow weights stays cacheable) */
#define SMOLTTS_STREAM_W_SLOW_NOT_W13 128   /* with _SLOW: everything of the slow blocks but w1|w3 */
/* default: the slow blocks' w1|w3 (94 MB at 150m) and the slow head stream past the caches; what is left -- the depth
 * transformer (94 MB, re-read 8 times per frame) and the rest of the slow blocks (83 MB) -- fits the 256 MB Infinity Cache together */
#define SMOLTTS_STREAM_W_DEFAULT (SMOLTTS_STREAM_W_SLOW | SMOLTTS_STREAM_W_SLOW_HEAD | SMOLTTS_STREAM_W_SLOW_ONLY_W13)
int smoltts_session_set_option(SmolttsSession* s, int32_t option, int32_t value);

/* Sampling mode (reference GenerationSettings, lm/generate.py:12-16): temp / fast_temp <= 0 select
 * greedy argmax for the slow / depth tokens (the default), otherwise exact categorical sampling from
 * softmax(logits / temp) with a counter-based generator keyed by (seed, slot, frame, step); min_p > 0
 * keeps only tokens with p >= min_p * p_max (applied to the depth tokens only when fast_temp > 0). */
int smoltts_session_set_sampling(SmolttsSession* s, float temp, float fast_temp, float min_p, uint64_t seed);

/* Device pointers to the session's results (valid for the session's lifetime):
 *   codes      int32 [max_batch][max_frames][1 + n_fast]   emitted columns (slow id, codes)
 *   n_frames   int32 [max_batch]                            frames emitted so far per slot
 *   done       int32 [max_batch]
 *   margin     float [max_batch]   smallest top-1/top-2 logit gap seen by the slot's argmaxes */
int smoltts_session_outputs(SmolttsSession* s, int32_t** codes_dev, int32_t** n_frames_dev,
                            int32_t** done_dev, float** margin_dev);
/* Diagnostics: the slow transformer's KV cache of the session -- k / v [n_layer][max_batch][n_kv_head][max_seq][64] in the session's
 * kv format (fp32 or bf16), `layer_bytes` apart per layer.  Read-only use (tests compare cache rows between launch structures). */
int smoltts_session_kv_cache(SmolttsSession* s, void** k_dev, void** v_dev, uint64_t* layer_bytes);
/* int32 [max_batch]: where each slot's smallest gap occurred, frame * 64 + step (step 0 = slow id, i = depth code i-1). */
int smoltts_session_margin_at(SmolttsSession* s, int32_t** margin_at_dev);

/* ------------------------------------------------------------------------------ Mimi decoder */
#define SMOLTTS_MIMI_MAX_LAYERS 16

typedef struct SmolttsMimiLayerWeights {
  uint64_t ln1_w, ln1_b;     /* fp32 [512] */
  uint64_t wqkv;             /* fp32 T16x32 [1536][512]; q and k rows permuted per head so that the
                                half-split RoPE pair (j, j+32) sits at rows (2j, 2j+1) */
  uint64_t wo;               /* fp32 T16x32 [512][512] */
  uint64_t ls1;              /* fp32 [512] attention layer scale */
  uint64_t ln2_w, ln2_b;
  uint64_t fc1;              /* fp32 T16x32 [2048][512] */
  uint64_t fc2;              /* fp32 T16x32 [512][2048] */
  uint64_t ls2;
  uint64_t wqkv3, wo3, fc13, fc23;  /* the same four matrices as "W3" tiles (below); 0 = absent (fp32 kernels only) */
} SmolttsMimiLayerWeights;

typedef struct SmolttsMimiConv {
  uint64_t w;                /* fp32 T16x32 [N][K] GEMM form, see DESIGN.md "convolutions as GEMMs" */
  uint64_t b;                /* fp32 [N] (bias repeated per phase for transposed convs) */
  int32_t cin, cout, k, stride, transposed;
  uint64_t w3;               /* the same GEMM matrix as "W3" tiles; 0 = absent */
} SmolttsMimiConv;

typedef struct SmolttsMimiConfig {
  int32_t num_codebooks;     /* codebooks consumed per frame (8) */
  int32_t n_layers;          /* decoder transformer layers (8) */
  int32_t window;            /* attention window in positions; 0 = unbounded (reference MLX behaviour) */
  int32_t max_positions;     /* rows of the RoPE table / KV capacity per slot (2 per frame) */
} SmolttsMimiConfig;

typedef struct SmolttsMimiWeights {
  uint64_t rvq_table;        /* fp32 [num_codebooks][2048][512]: codebook rows already projected by
                                their group's output_proj */
  uint64_t upsample_w;       /* fp32 [4][512] depthwise ConvTranspose taps, tap-major */
  uint64_t rope;             /* fp32 [max_positions][32][2] */
  SmolttsMimiLayerWeights layers[SMOLTTS_MIMI_MAX_LAYERS];
  SmolttsMimiConv convs[14]; /* SEANet decoder in execution order: conv0, then per ratio
                                (convtr, res.conv3, res.conv1) x4, then the final conv (14 in all) */
  uint64_t final_w;          /* fp32 [3][64]: the output conv (64 -> 1, k3) tap-major, for the fused last stage; its bias is
                                convs[13].b */
} SmolttsMimiWeights;

typedef struct SmolttsMimi SmolttsMimi;
typedef struct SmolttsMimiSession SmolttsMimiSession;

int smoltts_mimi_create(const SmolttsMimiConfig* cfg, const SmolttsMimiWeights* offsets,
                        const void* arena_dev, size_t arena_bytes, SmolttsMimi** out);
void smoltts_mimi_destroy(SmolttsMimi* m);

/* Session: per-slot streaming state (upsample carry, transformer KV, conv halos) + scratch for
 * up to max_chunk_frames frames per slot per call. */
size_t smoltts_mimi_slab_bytes(const SmolttsMimi* m, int32_t max_batch, int32_t max_chunk_frames);
int smoltts_mimi_session_create(SmolttsMimi* m, void* slab_dev, size_t slab_bytes, int32_t max_batch,
                                int32_t max_chunk_frames, SmolttsMimiSession** out);
void smoltts_mimi_session_destroy(SmolttsMimiSession* s);
/* Forget all streaming state of the session (start new utterances in every slot). */
int smoltts_mimi_reset(SmolttsMimiSession* s, void* stream);

/* Forget the streaming state of the listed slots only (slots_host: host memory): the other slots' streams continue.
 * Every slot keeps its own transformer position, so streams of different utterances can share one session. */
int smoltts_mimi_reset_slots(SmolttsMimiSession* s, const int32_t* slots_host, int32_t n_slots, void* stream);

/* (ABI 6) Reference-quirk switch, default 0.  SMOLTTS_MIMI_OPT_STATELESS_UPSAMPLE = 1: every decode call up-samples its frames
 * as if they were the whole utterance (no tap overlap carried in from the previous call), which is what the reference's
 * decode_step does (mlx_inference/src/smoltts_mlx/codec/mimi.py:73-77,101-104: self.upsample(embeddings) on the call's frames
 * alone; transformer cache and SEANet state are carried as here).  With one frame per call this reproduces the reference's
 * SmolTTS.stream (__init__.py:83-95) sample for sample; chunked decode then no longer equals the batch decode -- which is the
 * reference's defect, kept switchable like the other two torch / MLX disagreements.  Takes effect from the
 * next decode call; returns SMOLTTS_E_INVALID for an unknown option or a value other than 0 / 1.
 *
 * SMOLTTS_MIMI_OPT_PRODUCTS = 3 | 6 (default 6): how many of the bf16 x bf16 products of the split operands the codec's matrix-core
 * kernels form (every Linear, conv, ConvTranspose and the chunk attention of many-row calls, and the fused SEANet stages at any
 * chunk size; the skinny GEMMs of few-row calls are fp32-MFMA kernels and do not change).  6: every product of weight >= 2^-16 -- fp32-grade, PCM RMS ~1e-7 against the fp32 oracle.  3: the products of
 * weight >= 2^-8 only -- half the matrix-core work, the lo pieces are never loaded: a 1024-frame chunk decodes 23 % faster at a PCM
 * RMS error of 7e-7 against the fp32 oracle (signal RMS 0.2-0.4; the contract's bar is 1e-4).  The reference's own codec runs fp32
 * (codec/mimi.py:107,147: format "fp32" unless "bf16" is asked for), hence the default. */
#define SMOLTTS_MIMI_OPT_STATELESS_UPSAMPLE 1
#define SMOLTTS_MIMI_OPT_PRODUCTS 2
int smoltts_mimi_session_set_option(SmolttsMimiSession* s, int32_t option, int32_t value);

/* Decode n_frames new frames for slots [0, batch): codes_dev int32 [batch][codes_stride] where
 * frame f of slot b starts at codes_dev[b*codes_stride + f*frame_stride + code_offset] and holds
 * num_codebooks ints; pcm_dev float [batch][pcm_stride] receives 1920*n_frames samples per slot.
 * Streaming state is carried between calls, so decoding F frames in chunks equals decoding them
 * at once (== MimiModel.decode of the whole code grid). */
int smoltts_mimi_decode_chunk(SmolttsMimiSession* s, const int32_t* codes_dev, int64_t codes_stride,
                              int32_t frame_stride, int32_t code_offset, int32_t batch,
                              int32_t n_frames, float* pcm_dev, int64_t pcm_stride, void* stream);

/* ------------------------------------------------------------------------------ Mimi encoder
 * Voice-clone prompts: MimiModel.encode (mlx_inference/src/smoltts_mlx/codec/mimi.py:64-71) =
 * SEANet encoder (codec/seanet.py:52-96) -> encoder transformer (codec/transformer.py:134-150) ->
 * downsample conv (mimi.py:37-46) -> split RVQ encode (codec/rvq.py:99-116,157-177).  One utterance
 * per call, whole signal at once, as in the reference (no streaming encode there either). */
typedef struct SmolttsMimiEncConfig {
  int32_t num_codebooks;     /* codebooks produced per frame: 1 semantic + (num_codebooks-1) acoustic */
  int32_t n_layers;          /* encoder transformer layers (8) */
  int32_t window;            /* attention window in positions (25 Hz); 0 = unbounded (reference MLX) */
  int32_t max_positions;     /* rows of the RoPE table = longest signal in 960-sample steps */
  int32_t extra_right;       /* 0: the stride-alignment padding goes on the left with the causal padding
                                (reference, codec/conv.py:25-41); 1: on the right (transformers.MimiConv1d) */
} SmolttsMimiEncConfig;

typedef struct SmolttsMimiEncWeights {
  uint64_t conv0_w;          /* fp32 [64][8]: the 7 taps of encoder.layers.0 (1 -> 64 channels), 8th = 0 */
  uint64_t conv0_b;          /* fp32 [64] */
  SmolttsMimiConv convs[13]; /* per ratio 4,5,6,8: res.conv3, res.conv1, strided conv (k = 2*ratio); then the
                                final conv k3 (1024 -> 512); GEMM form [cout][k*cin] */
  SmolttsMimiLayerWeights layers[SMOLTTS_MIMI_MAX_LAYERS];
  uint64_t rope;             /* fp32 [max_positions][32][2] */
  uint64_t downsample_w;     /* fp32 T16x32 [512][4*512], no bias */
  uint64_t in_proj[2];       /* fp32 T16x32 [256][512]: semantic, acoustic group input_proj */
  uint64_t codebooks_t;      /* fp32 T16x32 [num_codebooks][2048][256]: embed_sum / max(usage, 1e-5) */
  uint64_t codebooks;        /* the same rows, row-major (residual update) */
  uint64_t codebook_sq;      /* fp32 [num_codebooks][2048] squared norms of the rows */
} SmolttsMimiEncWeights;

typedef struct SmolttsMimiEncoder SmolttsMimiEncoder;

int smoltts_mimi_encoder_create(const SmolttsMimiEncConfig* cfg, const SmolttsMimiEncWeights* offsets,
                                const void* arena_dev, size_t arena_bytes, SmolttsMimiEncoder** out);
void smoltts_mimi_encoder_destroy(SmolttsMimiEncoder* e);
/* Frames produced for n_samples of 24 kHz audio: ceil(ceil(n/960)/2). */
int32_t smoltts_mimi_encode_frames(int32_t n_samples);
size_t smoltts_mimi_encode_workspace_bytes(const SmolttsMimiEncoder* e, int32_t n_samples);
/* pcm_dev float [n_samples] -> codes_dev int32 [num_codebooks][frames] (row q = codebook q).
 * Optional outputs (NULL to skip): emb_dev float [frames][512] pre-quantisation latents; gap_dev float
 * [num_codebooks][frames] squared-distance gap between the chosen and the second-nearest entry. */
int smoltts_mimi_encode(SmolttsMimiEncoder* e, const float* pcm_dev, int32_t n_samples, int32_t* codes_dev,
                        float* emb_dev, float* gap_dev, void* workspace_dev, size_t workspace_bytes,
                        void* stream);

/* --------------------------------------------------------------- operator-level test entry points */
enum {  /* prologue applied to the activation operand */
  SMOLTTS_PRO_NONE = 0,
  SMOLTTS_PRO_RMSNORM = 1,  /* x * rsqrt(mean(x^2)+eps) * gamma */
  SMOLTTS_PRO_ELU = 2,
  SMOLTTS_PRO_LAYERNORM = 3 /* fp32 weights: nn.LayerNorm over the K values of a row: gamma_dev = weight, beta_dev = bias, eps;
                               fused into the GEMM where its kernel can (few rows; K = 512 at chunk size), otherwise applied by the
                               stand-alone kernel into ln_scratch_dev [M][K] first */
};
enum {  /* epilogue */
  SMOLTTS_EPI_STORE = 0,        /* out = acc (+ bias) */
  SMOLTTS_EPI_RESID = 1,        /* out = resid + acc (+ bias) */
  SMOLTTS_EPI_SWIGLU = 2,       /* rows interleaved (gate, up): out[n/2] = silu(gate) * up */
  SMOLTTS_EPI_GELU = 3,         /* out = gelu_erf(acc) */
  SMOLTTS_EPI_SCALE_RESID = 4,  /* out = resid + scale[n] * acc */
  SMOLTTS_EPI_QKV_ROPE = 5      /* interleaved RoPE on q,k; q -> out, k/v -> caches */
};

typedef struct SmolttsGemmArgs {
  const void* w_dev;          /* T16x32 tiles */
  int32_t w_is_fp32;          /* 0: bf16, 1: fp32 */
  const float* x_dev;         /* activations; row m at x + (m / rows_per_batch) * x_bstride
                                 + (m % rows_per_batch) * ldx   (rows_per_batch 0 => m * ldx) */
  int64_t ldx, x_bstride;
  int32_t rows_per_batch;
  int32_t M, N, K;
  int32_t prologue, epilogue;
  const float* gamma_dev;     /* RMSNorm weight [K] */
  float eps;
  const float* bias_dev;      /* [N] or NULL */
  const float* scale_dev;     /* [N], EPI_SCALE_RESID */
  const float* resid_dev;     /* row m at resid + (m / rpb) * r_bstride + (m % rpb) * ldr;
                                 ldr == 0 => same addressing as out */
  int64_t ldr, r_bstride;
  float* out_dev;             /* row m at out + (m / rows_per_batch) * o_bstride + (m % rpb) * ldo */
  int64_t ldo, o_bstride;
  int32_t elu_out;            /* STORE / RESID: store ELU(result) (the consumer then needs no ELU prologue) */
  float* raw_out_dev;         /* STORE: optional second copy of the un-activated result, row stride ldo */
  int64_t raw_bstride;
  /* EPI_QKV_ROPE */
  const float* rope_dev;      /* [pos][32][2] */
  const int32_t* row_pos_dev; /* [M] */
  const int32_t* row_slot_dev;/* [M] */
  float* k_cache_dev;         /* [slots][n_kv][cache_len][64] */
  float* v_cache_dev;
  int32_t n_q_heads, n_kv_heads, cache_len;
  const void* w3_dev;         /* optional, fp32 weights only: the same matrix as "W3" tiles; many-row calls (M >= 256, N >= 64,
                                 no prologue or the ELU one) then run on the bf16 matrix cores with split operands (fp32-grade results) */
  float* splitk_ws_dev;       /* optional, with w3_dev: workspace for split-K partial sums (long K over few tiles) ... */
  int64_t splitk_ws_floats;   /* ... and its size in floats (4 * M * N needed; smaller = no split) */
  const float* beta_dev;      /* PRO_LAYERNORM: bias [K] */
  float* ln_scratch_dev;      /* PRO_LAYERNORM: [M][K] floats for the calls whose kernel has no such prologue */
  void* k_cache3_dev;         /* EPI_QKV_ROPE, optional (both or neither): the same K / V rows also as bf16x3 PIECE caches, the operands */
  void* v_cache3_dev;         /* of smoltts_k_attention_rows3 (layouts there); SMOLTTS_KV3_BYTES(slots, n_kv, cache_len) bytes each, */
                              /* zero-filled once by the owner (unwritten positions are multiplied by zero weights: they must be finite) */
  int32_t b3_products;        /* (ABI 6) calls that run on bf16x3 pieces (w3_dev): 3 = three MFMA products per operand pair (hi*hi, mid*hi, */
                              /* hi*mid: 2^-16-grade results, lo pieces never loaded), anything else = six (fp32-grade, the default) */
} SmolttsGemmArgs;

/* bytes of one bf16x3 piece cache: per (slot, kv head) 6 bytes per value over cache_len rounded up to whole 32-position blocks */
#define SMOLTTS_KV3_BYTES(slots, n_kv, cache_len) ((size_t)(slots) * (size_t)(n_kv) * (size_t)(((cache_len) + 31) / 32 * 32) * 384u)

int smoltts_k_gemm(const SmolttsGemmArgs* a, void* stream);

/* bf16-MFMA GEMM of the DualAR transformer (smoltts_amd/csrc/gemm3.hip).  The activation operand is
 * in the "X3" format (smoltts_amd/csrc/x3.h): rows in tiles of 16, K in chunks of 32, block
 * (row tile, chunk, piece p in 0..2) = 1 KiB at ((tile * K/32 + chunk) * 3 + p) * 1024 holding for
 * lane l = 16*q + r the 8 bf16 of piece p of x[16*tile + r][32*chunk + 8q .. +8); the three pieces
 * sum exactly to the fp32 value (already multiplied by the consumer's RMSNorm weight).  Buffer
 * size: ceil(M/16)*16 * K * 6 bytes. */
enum {  /* weight formats of the DualAR Linears */
  SMOLTTS_W_BF16 = 0,  /* bf16 T16x32 tiles (1 KiB per 16 rows x 32 columns) */
  SMOLTTS_W_FP8 = 1    /* fp8 e4m3 (OCP) T16x32 tiles (512 B) + one fp32 scale per output row: w = q * scale */
};

typedef struct SmolttsGemm3Args {
  const void* w_dev;           /* T16x32 [N][K] in w_format */
  const void* x3_dev;          /* X3 operand [M][K] */
  int32_t M, N, K;
  int32_t epilogue;            /* SMOLTTS_EPI_STORE | _RESID | _SWIGLU | _QKV_ROPE */
  const float* ssq_in_dev;     /* [M][K/16] partial sums of squares of the un-normed input rows: the result
                                  rows are scaled by rsqrt(sum/K + eps); NULL = input not normed */
  float eps;
  const float* bias_dev;       /* [N] or NULL */
  const float* resid_dev;      /* EPI_RESID: fp32 [M][ldo] */
  float* out_dev;              /* fp32 [M][ldo]: STORE / RESID result, QKV_ROPE q rows */
  int64_t ldo;
  void* x3_out_dev;            /* EPI_SWIGLU: X3 [M][N/2] of silu(gate) * up */
  /* STORE / RESID: also publish the result for the next GEMM(s): X3 of out * gamma (gamma NULL = 1),
   * up to two consumers, and the per-(row, 16-column tile) sums of squares [M][N/16] */
  void* emit_a_dev; const float* gamma_a_dev;
  void* emit_b_dev; const float* gamma_b_dev;
  float* ssq_out_dev;
  /* EPI_QKV_ROPE */
  const float* rope_dev; const int32_t* row_pos_dev; const int32_t* row_slot_dev;
  float* k_cache_dev; float* v_cache_dev;
  int32_t n_q_heads, n_kv_heads, cache_len;
  int32_t w_format;            /* SMOLTTS_W_BF16 | SMOLTTS_W_FP8 */
  const float* w_scale_dev;    /* SMOLTTS_W_FP8: [N] row scales; NULL otherwise */
  void* v_x3_dev;              /* EPI_QKV_ROPE, optional: every row sits at position 0, so its attention output is its own
                                  V row (softmax over one key): V is also written as the X3 operand [M][n_q_heads*64] of
                                  the output projection (each kv head repeated for its query heads) and no attention
                                  launch is needed */
  int32_t kv_format;           /* EPI_QKV_ROPE: SMOLTTS_KV_F32 | SMOLTTS_KV_BF16 storage of k_cache_dev / v_cache_dev */
  int32_t w_stream;            /* != 0: these weights are read once now and not again before the caches have turned over (the slow
                                  layers of a decode frame): their loads carry the non-temporal hint, so that they do not push the
                                  depth transformer's weights -- re-read 8 times per frame -- out of the Infinity Cache (M <= 128 only) */
  /* EPI_RESID with the attention of a depth step as its activation operand (ABI 5): attn_q_dev != NULL means x3_dev is not read;
   * instead every workgroup works out, for the rows it multiplies, softmax(q K^T / 8) V over keys 0 .. attn_pos of the row's own
   * slot (row r == slot r) in k_cache_dev / v_cache_dev [M][n_kv_heads][cache_len][64] fp32 -- the output of
   * smoltts_k_attention for those rows -- and feeds it to the matrix cores as bf16x3 pieces through LDS.  One launch instead of
   * two per depth layer, with the same bits (every sum in the order smoltts_k_attention + this launch without the prologue form it, at
 * up to 128 rows); needs smoltts_gemm3_attn_fusable(n_q_heads, n_kv_heads, cache_len) and K == n_q_heads * 64. */
  const float* attn_q_dev;     /* fp32 [M][n_q_heads*64] (RoPE applied: the q rows an EPI_QKV_ROPE launch wrote) */
  int32_t attn_pos;
  /* EPI_STORE (ABI 5): also leave, for every (row, 16-column tile) of the result, the tile's largest value, the first column that
   * holds it (int32 bits) and the runner-up: float [M][ceil(N/16)][4] (4th unused) -- what a greedy pick needs of a row of logits
   * (M < 256: the many-row kernel has no such epilogue) */
  float* cand_out_dev;
  /* the attention prologue with the PICK in front of it (ABI 5; NULL = off): the rows are depth step attn_pos of a frame whose
   * previous head GEMM left cand_out_dev; see SmolttsPickArgs.  attn_q_dev and resid_dev are then not read. */
  const struct SmolttsPickArgs* pick;
  /* SMOLTTS_W_FP8 weights, M >= 256 (prompt rows) only (ABI 5): != 0 runs the GEMM on the fp8 matrix-core instruction
   * (v_mfma_f32_16x16x32_fp8_fp8) -- the weight tiles as they are, the activation operand's leading bf16 piece rounded to e4m3
   * -- instead of three exact bf16 MFMAs per chunk: BASELINE configs[4]'s "fp8 MFMA prefill".  The result then carries the
   * activations' fp8 rounding (~2^-4 relative per element, ~1e-3 of the output scale): NOT the parity path. */
  int32_t fp8_activations;
} SmolttsGemm3Args;

/* Greedy pick of the previous depth step's code inside the launch that consumes it (depth layer 0 of the next step, whose q | k | v
 * come out of the engine's fast_qkv table): every workgroup merges its rows' tile candidates (first maximal column, as
 * torch.argmax), gathers q and the new key's K / V rows of the picked embedding row from the table (RoPE for attn_pos applied on the
 * way, bit-identical to the stand-alone picking kernel's), takes the residual row from the embedding table, and runs attention + wo
 * as with attn_q_dev.  The workgroups of column group 0 also record the id, the top-2 gap bookkeeping of smoltts_k_argmax and the
 * new K / V cache rows.  One launch fewer per depth step (lm/generate.py:118-141). */
typedef struct SmolttsPickArgs {
  const float* cand_dev;          /* [M][cand_tiles][4] from the head GEMM's cand_out_dev */
  int32_t cand_tiles;             /* codebook_size / 16 */
  const float* qkv_table_dev;     /* fp32 [emb rows][(n_q_heads + 2 n_kv_heads) * 64], before RoPE */
  const float* rope_dev;          /* fp32 [pos][32][2] */
  const void* emb_dev;            /* bf16 [emb rows][K]: the residual row of a picked id */
  int32_t emb_row_offset;         /* embedding / table row = id + emb_row_offset (depthwise tables: lm/generate.py:136-140) */
  int32_t* ids_dev;               /* row r's id -> ids_dev[r * ids_stride] */
  int32_t ids_stride;
  float* margin_dev;              /* [M] smallest top-2 gap so far (updated where margin_mask_dev[r] != 0), or NULL */
  const int32_t* margin_mask_dev;
  int32_t* margin_at_dev;         /* [M] frames_dev[r] * 64 + step of that gap, or NULL */
  const int32_t* frames_dev;
  int32_t step;
} SmolttsPickArgs;

/* 1 when the attention prologue above exists for this head layout (<= 12 query heads, groups of <= 4 per kv head, <= 8 cache entries) */
int smoltts_gemm3_attn_fusable(int32_t n_q_heads, int32_t n_kv_heads, int32_t cache_len);

int smoltts_k_gemm3(const SmolttsGemm3Args* a, void* stream);
/* fp32 rows [n_rows][ldx] -> X3 operand(s) (+ sums of squares), for tests and operands without a fused producer */
int smoltts_k_x3_pack(const float* x_dev, int64_t ldx, int32_t n_rows, int32_t dim, void* x3a_dev,
                      const float* gamma_a_dev, void* x3b_dev, const float* gamma_b_dev, float* ssq_dev,
                      void* stream);

/* Measurement aid of ONE session (bench.py, tools/marginal_cost.py): while code >= 0, every launch of that kernel class
 * inside this session's frames is issued twice -- code = the GEMM's epilogue (and N == n_filter when n_filter > 0;
 * EPI_RESID is refused: not idempotent), 100 = depth (<= 16 keys) attention, 101 = slow attention.  The session's
 * captured graphs are dropped; time a frame graph with and without it: the difference per extra launch is the kernel's
 * in-situ duration.  Other sessions are unaffected (there is no process-global debug state in this library; the event /
 * cycle-stamp hooks of tools/ exist only in diagnostic builds with -DSMOLTTS_DEBUG_HOOKS, built to a separate path). */
int smoltts_session_measure_duplicate(SmolttsSession* s, int32_t code, int32_t n_filter);
/* Forget the captured frame graphs (the next smoltts_lm_decode captures again). */
int smoltts_session_drop_graph(SmolttsSession* s);

/* GQA attention of one query row per (row, kv head) over the slot's cache prefix:
 * keys [max(0, pos+1-window), pos]; q_dev/out_dev [n_rows][n_q_heads*64]; out_x3_dev (optional) receives
 * the same rows as an X3 operand; either output may be NULL. */
int smoltts_k_attention(const float* q_dev, const float* k_cache_dev, const float* v_cache_dev,
                        const int32_t* row_pos_dev, const int32_t* row_slot_dev, int32_t n_rows,
                        int32_t n_q_heads, int32_t n_kv_heads, int32_t cache_len, int32_t window,
                        float* out_dev, void* out_x3_dev, void* stream);

/* The same over a cache stored in kv_format (SMOLTTS_KV_BF16: caches of more than 16 entries). */
int smoltts_k_attention_kv(const float* q_dev, const void* k_cache_dev, const void* v_cache_dev,
                           const int32_t* row_pos_dev, const int32_t* row_slot_dev, int32_t n_rows,
                           int32_t n_q_heads, int32_t n_kv_heads, int32_t cache_len, int32_t window,
                           float* out_dev, void* out_x3_dev, int32_t kv_format, void* stream);
/* The same with the keys of every (row, kv head) pair dealt over two workgroups and merged inside the launch (decode frames
 * at B <= 32: rows x kv heads <= 128, caches longer than 128 entries; other shapes run the one-workgroup kernels).
 * split_part_dev: 128 * 2 * (4 * 64 + 8) floats of scratch; split_ticket_dev: 128 int32 zeroed ONCE (tickets count up across
 * launches: two arrivals per pair and launch). */
int smoltts_k_attention_split(const float* q_dev, const void* k_cache_dev, const void* v_cache_dev, const int32_t* row_pos_dev,
                              const int32_t* row_slot_dev, int32_t n_rows, int32_t n_q_heads, int32_t n_kv_heads, int32_t cache_len,
                              int32_t window, float* out_dev, void* out_x3_dev, int32_t kv_format, float* split_part_dev,
                              int32_t* split_ticket_dev, void* stream);

/* Attention of MANY rows per slot (the codec transformer over a chunk: rows_per_slot consecutive positions per slot,
 * rows_per_slot % 32 == 0, n_rows % rows_per_slot == 0, one query head per kv head) on the bf16 matrix cores over the PIECE
 * caches an EPI_QKV_ROPE GEMM with k_cache3_dev / v_cache3_dev wrote: every K / V value as three bf16 pieces (hi + mid + lo ==
 * the fp32 value exactly), six products per operand pair as in the many-row GEMMs, fp32 accumulation, flash-style softmax.
 * Layouts, per (slot, head) at ((slot * n_heads + head) * ceil32(cache_len) * 384 bytes:
 *   K3: positions in tiles of 16, the 64 dims in two chunks of 32: block (tile, chunk, piece) = 1 KiB at
 *       ((tile * 2 + chunk) * 3 + piece) * 1024; lane l = 16 q + r holds the 8 bf16 K[16 tile + r][32 chunk + 8q .. + 8)
 *       (the "X3" row format of smoltts_amd/csrc/x3.h with K = 64);
 *   V3: positions in blocks of 32, dims in four tiles of 16: block (pb, dim tile t, piece) = 1 KiB at
 *       ((pb * 4 + t) * 3 + piece) * 1024; lane l = 16 q + r holds, for j = 0..7, V[32 pb + (j < 4 ? 4q + j : 16 + 4q + j - 4)][16 t + r]
 *       (the key order in which a lane of the score MFMA holds its eight probabilities of a 32-key block).
 * Same keys as smoltts_k_attention: [max(0, pos + 1 - window), pos] of the row's slot. */
int smoltts_k_attention_rows3(const float* q_dev, const void* k_cache3_dev, const void* v_cache3_dev, const int32_t* row_pos_dev,
                              const int32_t* row_slot_dev, int32_t n_rows, int32_t rows_per_slot, int32_t n_heads, int32_t cache_len,
                              int32_t window, float* out_dev, void* stream);

/* x[r] = E_text[cols[r][0]] + keep * sum_k E_cb[cols[r][1+k] + k*codebook_size] */
int smoltts_k_embed(const int32_t* cols_dev, int32_t n_rows, int32_t n_code_rows,
                    const void* text_emb_dev, const void* cb_emb_dev, int32_t dim,
                    int32_t codebook_size, int32_t cb_first_offset, int32_t mask_mode,
                    int32_t sem_start, int32_t sem_end, float* x_dev, void* stream);

/* ids[r] = first index of the row maximum; margin[r] = min(margin[r], top1 - top2). */
int smoltts_k_argmax(const float* logits_dev, int32_t n_rows, int32_t n_cols, int64_t ld,
                     int32_t* ids_dev, int32_t ids_stride, float* margin_dev, void* stream);

/* ids[r] ~ softmax(logits[r] / temp) (min_p as above); row r uses the stream (seed, r, frame_base + r, step). */
int smoltts_k_sample(const float* logits_dev, int32_t n_rows, int32_t n_cols, int64_t ld, float temp, float min_p,
                     uint64_t seed, int32_t frame_base, int32_t step, int32_t* ids_dev, void* stream);

int smoltts_k_layernorm(const float* x_dev, const float* w_dev, const float* b_dev, int32_t n_rows,
                        int32_t dim, float eps, float* out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SMOLTTS_HIP_H */
