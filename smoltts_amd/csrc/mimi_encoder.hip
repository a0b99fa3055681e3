// Mimi encoder: 24 kHz PCM -> RVQ codes (voice-clone prompts).
//
// Reference chain MimiModel.encode (mlx_inference/src/smoltts_mlx/codec/mimi.py:64-71):
//   SEANet encoder (codec/seanet.py:52-96) -> encoder transformer (codec/transformer.py:134-150) ->
//   downsample conv k4 s2 with edge padding (mimi.py:37-46) -> split RVQ encode (codec/rvq.py:99-116,
//   157-177: nearest codebook row by Euclidean distance, residual loop per group).
// Same construction as the decoder engine: channel-last fp32 buffers with zero halo rows in front, every
// convolution one GEMM on the fp32 matrix-core kernel (gemm.hip).  A stride-s conv of k = 2s taps reads,
// for output row t, the k consecutive input rows starting at row t*s of its halo-prefixed buffer: the
// GEMM's row stride is s*Cin and its K is k*Cin.  Each producer stores what its consumers read: the raw
// activation for the residual add and ELU(activation) for the next conv (ELU(0) = 0 keeps the halos valid).
// The causal padding k-s and the stride-alignment extra padding are both rows of zeros in front of the
// data (the reference's causal_pad1d pads everything on the left, codec/conv.py:25-41); with
// cfg.extra_right the extra rows sit behind the data instead, as transformers.MimiConv1d does.
#include <new>

#include "mimi_common.h"

using namespace smoltts;

namespace {
constexpr int NSTAGE = 4;
constexpr int RATIO[NSTAGE] = {4, 5, 6, 8};
constexpr int D = MIMI_D, HEADS = MIMI_HEADS, FF = MIMI_FF, CB_DIM = 256, CB_SIZE = 2048;

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct Plan {
  int T[NSTAGE + 1];  // rows entering stage i (T[0] = samples); T[4] = 25 Hz rows
  int extra[NSTAGE];  // stride-alignment rows of the stage's strided conv
  int left[NSTAGE];   // zero rows in front of the strided conv's input
  int F, ds_extra, ds_left;
  float *xraw[NSTAGE], *xelu[NSTAGE], *helu[NSTAGE], *yelu[NSTAGE], *zelu;
  float *tx, *tn, *tq, *ta, *th, *kc, *vc, *ds, *emb, *res, *dots;
  int *row_pos, *row_slot;
  size_t total;
};

void make_plan(Plan& p, int n_samples, int n_layers, bool extra_right, char* base) {
  size_t off = 0;
  auto take = [&](size_t n_floats) {
    float* r = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off = align_up(off + n_floats * sizeof(float));
    return r;
  };
  p.T[0] = n_samples;
  for (int i = 0; i < NSTAGE; ++i) {
    const int r = RATIO[i], C = 64 << i;
    p.T[i + 1] = ceil_div(p.T[i], r);
    p.extra[i] = p.T[i + 1] * r - p.T[i];
    p.left[i] = r + (extra_right ? 0 : p.extra[i]);
    p.xraw[i] = take((size_t)p.T[i] * C);
    p.xelu[i] = take((size_t)(2 + p.T[i]) * C);
    p.helu[i] = take((size_t)p.T[i] * (C / 2));
    p.yelu[i] = take((size_t)(r + p.extra[i] + p.T[i]) * C);
  }
  const size_t T4 = p.T[NSTAGE];
  p.zelu = take((2 + T4) * 1024);
  p.tx = take(T4 * D); p.tn = take(T4 * D); p.tq = take(T4 * D); p.ta = take(T4 * D); p.th = take(T4 * FF);
  p.kc = take((size_t)n_layers * HEADS * T4 * 64);
  p.vc = take((size_t)n_layers * HEADS * T4 * 64);
  p.F = ceil_div((int)T4, 2);
  p.ds_extra = p.F * 2 - (int)T4;
  p.ds_left = 2 + (extra_right ? 0 : p.ds_extra);
  p.ds = take((2 + p.ds_extra + T4) * D);
  p.emb = take((size_t)p.F * D);
  p.res = take((size_t)p.F * CB_DIM);
  p.dots = take((size_t)p.F * CB_SIZE);
  p.row_pos = reinterpret_cast<int*>(take(T4));
  p.row_slot = reinterpret_cast<int*>(take(T4));
  p.total = off;
}

__device__ __forceinline__ float elu1(float v) { return v > 0.f ? v : expm1f(v); }

// encoder.layers.0: Conv1d(1 -> 64, k7), causal.  One thread = 4 channels of one sample; writes the raw
// row (residual input of the first resnet block) and its ELU behind a 2-row halo (that block's conv k3).
__global__ __launch_bounds__(256) void enc_conv0_kernel(const float* pcm, int n, const float* w, const float* b, float* xraw,
                                                        float* xelu) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long t = gid >> 4;
  const int c = (int)(gid & 15) * 4;
  if (t >= n) return;
  float x[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const long s = t - 6 + j;
    x[j] = s >= 0 ? pcm[s] : 0.f;
  }
  float o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 w0 = *reinterpret_cast<const float4*>(w + (c + i) * 8), w1 = *reinterpret_cast<const float4*>(w + (c + i) * 8 + 4);
    o[i] = b[c + i] + x[0] * w0.x + x[1] * w0.y + x[2] * w0.z + x[3] * w0.w + x[4] * w1.x + x[5] * w1.y + x[6] * w1.z;
  }
  *reinterpret_cast<float4*>(xraw + t * 64 + c) = make_float4(o[0], o[1], o[2], o[3]);
  *reinterpret_cast<float4*>(xelu + (t + 2) * 64 + c) = make_float4(elu1(o[0]), elu1(o[1]), elu1(o[2]), elu1(o[3]));
}

// "edge" padding of the downsample conv: rows [0, left) repeat the first data row, rows behind the data the last
__global__ __launch_bounds__(128) void edge_fill_kernel(float* buf, int left, int n_rows, int right) {
  const int c = threadIdx.x * 4, row = blockIdx.x;  // row over left + right pad rows
  const bool front = row < left;
  const float4 v = *reinterpret_cast<const float4*>(buf + (long)(front ? left : left + n_rows - 1) * D + c);
  *reinterpret_cast<float4*>(buf + (long)(front ? row : n_rows + row) * D + c) = v;
}

// One frame per workgroup: nearest codebook row to the residual by squared distance
// |r|^2 + |e_j|^2 - 2 r.e_j (rvq.py:16-22; the sqrt there is monotone), lowest index on ties; then the residual
// update r -= e_code (rvq.py:108-112).
__global__ __launch_bounds__(256) void rvq_pick_kernel(const float* dots, const float* sq, const float* rows, float* res,
                                                       int32_t* codes, float* gap) {
  __shared__ float s_a[256];
  __shared__ float s_b[256];
  __shared__ int s_i[256];
  const int f = blockIdx.x, tid = threadIdx.x;
  float* r = res + (long)f * CB_DIM;
  const float rv = r[tid];
  s_a[tid] = rv * rv;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (tid < h) s_a[tid] += s_a[tid + h];
    __syncthreads();
  }
  const float rsq = s_a[0];
  __syncthreads();
  float best = INFINITY, second = INFINITY;
  int bi = 0;
  const float* dp = dots + (long)f * CB_SIZE;
#pragma unroll
  for (int i = 0; i < CB_SIZE / 256; ++i) {
    const int j = tid + 256 * i;
    const float d2 = (rsq + sq[j]) - 2.0f * dp[j];
    if (d2 < best) { second = best; best = d2; bi = j; }
    else if (d2 < second) second = d2;
  }
  s_a[tid] = best; s_b[tid] = second; s_i[tid] = bi;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (tid < h) {
      const float a1 = s_a[tid], a2 = s_b[tid], b1 = s_a[tid + h], b2 = s_b[tid + h];
      const int ia = s_i[tid], ib = s_i[tid + h];
      const bool take_b = b1 < a1 || (b1 == a1 && ib < ia);
      s_a[tid] = take_b ? b1 : a1;
      s_i[tid] = take_b ? ib : ia;
      s_b[tid] = fminf(take_b ? a1 : b1, fminf(a2, b2));
    }
    __syncthreads();
  }
  const int code = s_i[0];
  if (tid == 0) {
    codes[f] = code;
    if (gap) gap[f] = s_b[0] - s_a[0];
  }
  r[tid] = rv - rows[(long)code * CB_DIM + tid];
}

}  // namespace

struct SmolttsMimiEncoder {
  SmolttsMimiEncConfig cfg;
  SmolttsMimiEncWeights w;
  const char* arena;
  size_t arena_bytes;
};

extern "C" {

int smoltts_mimi_encoder_create(const SmolttsMimiEncConfig* cfg, const SmolttsMimiEncWeights* offsets, const void* arena_dev,
                                size_t arena_bytes, SmolttsMimiEncoder** out) {
  ST_REQUIRE(cfg && offsets && arena_dev && out, SMOLTTS_E_INVALID, "mimi_encoder_create: null argument");
  ST_REQUIRE(cfg->num_codebooks >= 1 && cfg->num_codebooks <= 32 && cfg->n_layers >= 1 && cfg->n_layers <= SMOLTTS_MIMI_MAX_LAYERS &&
                 cfg->max_positions >= 2 && cfg->window >= 0,
             SMOLTTS_E_INVALID, "mimi_encoder_create: bad config");
  for (int i = 0; i < 13; ++i) {
    const SmolttsMimiConv& cv = offsets->convs[i];
    const int st = i / 3, j = i % 3, C = 64 << (st < NSTAGE ? st : NSTAGE);
    bool ok;
    if (i == 12) ok = cv.cin == 1024 && cv.cout == 512 && cv.k == 3 && cv.stride == 1;
    else if (j == 0) ok = cv.cin == C && cv.cout == C / 2 && cv.k == 3 && cv.stride == 1;
    else if (j == 1) ok = cv.cin == C / 2 && cv.cout == C && cv.k == 1 && cv.stride == 1;
    else ok = cv.cin == C && cv.cout == 2 * C && cv.stride == RATIO[st] && cv.k == 2 * RATIO[st];
    ST_REQUIRE(ok && !cv.transposed && cv.w % 16 == 0 && cv.b % 16 == 0 && cv.w < arena_bytes && cv.b < arena_bytes, SMOLTTS_E_INVALID,
               "mimi_encoder_create: conv %d descriptor inconsistent (cin=%d cout=%d k=%d stride=%d)", i, cv.cin, cv.cout, cv.k, cv.stride);
  }
  ST_REQUIRE(offsets->codebooks + (size_t)cfg->num_codebooks * CB_SIZE * CB_DIM * 4 <= arena_bytes &&
                 offsets->codebooks_t + (size_t)cfg->num_codebooks * CB_SIZE * CB_DIM * 4 <= arena_bytes &&
                 offsets->codebook_sq + (size_t)cfg->num_codebooks * CB_SIZE * 4 <= arena_bytes &&
                 offsets->rope + (size_t)cfg->max_positions * 64 * 4 <= arena_bytes,
             SMOLTTS_E_INVALID, "mimi_encoder_create: table offsets outside the arena");
  SmolttsMimiEncoder* e = new (std::nothrow) SmolttsMimiEncoder;
  ST_REQUIRE(e, SMOLTTS_E_INVALID, "mimi_encoder_create: out of host memory");
  e->cfg = *cfg; e->w = *offsets; e->arena = (const char*)arena_dev; e->arena_bytes = arena_bytes;
  *out = e;
  return SMOLTTS_OK;
}

void smoltts_mimi_encoder_destroy(SmolttsMimiEncoder* e) { delete e; }

int32_t smoltts_mimi_encode_frames(int32_t n_samples) {
  if (n_samples <= 0) return 0;
  int t = n_samples;
  for (int i = 0; i < NSTAGE; ++i) t = ceil_div(t, RATIO[i]);
  return ceil_div(t, 2);
}

size_t smoltts_mimi_encode_workspace_bytes(const SmolttsMimiEncoder* e, int32_t n_samples) {
  if (!e || n_samples <= 0) return 0;
  Plan p;
  make_plan(p, n_samples, e->cfg.n_layers, e->cfg.extra_right != 0, nullptr);
  return p.total;
}

int smoltts_mimi_encode(SmolttsMimiEncoder* e, const float* pcm_dev, int32_t n_samples, int32_t* codes_dev, float* emb_dev,
                        float* gap_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
  ST_REQUIRE(e && pcm_dev && codes_dev && workspace_dev, SMOLTTS_E_INVALID, "mimi_encode: null argument");
  ST_REQUIRE(n_samples > 0, SMOLTTS_E_INVALID, "mimi_encode: empty signal");
  ST_REQUIRE(((uintptr_t)workspace_dev & 255) == 0, SMOLTTS_E_INVALID, "mimi_encode: workspace must be 256-byte aligned");
  const SmolttsMimiEncConfig& c = e->cfg;
  Plan p;
  make_plan(p, n_samples, c.n_layers, c.extra_right != 0, (char*)workspace_dev);
  ST_REQUIRE(workspace_bytes >= p.total, SMOLTTS_E_CAPACITY, "mimi_encode: workspace has %zu bytes, %zu needed", workspace_bytes, p.total);
  const int T4 = p.T[NSTAGE];
  ST_REQUIRE(T4 <= c.max_positions, SMOLTTS_E_CAPACITY, "mimi_encode: %d samples = %d positions exceed max_positions=%d", n_samples, T4,
             c.max_positions);
  hipStream_t st = (hipStream_t)stream;
  const char* A = e->arena;
  // every halo / padding row must read as zero
  ST_CHECK_HIP(hipMemsetAsync(workspace_dev, 0, p.total, st));

  // 1. SEANet encoder
  {
    const long threads = (long)n_samples * 16;
    hipLaunchKernelGGL(enc_conv0_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, pcm_dev, n_samples,
                       (const float*)(A + e->w.conv0_w), (const float*)(A + e->w.conv0_b), p.xraw[0], p.xelu[0]);
    ST_CHECK_HIP(hipGetLastError());
  }
  for (int i = 0; i < NSTAGE; ++i) {
    const int C = 64 << i, r = RATIO[i], T = p.T[i];
    const SmolttsMimiConv &c3 = e->w.convs[3 * i], &c1 = e->w.convs[3 * i + 1], &cs = e->w.convs[3 * i + 2];
    {  // resnet block, first conv: ELU(x) (halo 2) -> ELU(h)
      SmolttsGemmArgs a = mimi_gemm_f32(A + c3.w, p.xelu[i], C, T, C / 2, 3 * C);
      a.bias_dev = (const float*)(A + c3.b); a.epilogue = SMOLTTS_EPI_STORE; a.elu_out = 1;
      a.out_dev = p.helu[i]; a.ldo = C / 2;
      ST_TRY(launch_gemm(a, st));
    }
    {  // second conv (k1) + block input -> ELU(y) behind the strided conv's padding rows
      SmolttsGemmArgs a = mimi_gemm_f32(A + c1.w, p.helu[i], C / 2, T, C, C / 2);
      a.bias_dev = (const float*)(A + c1.b); a.epilogue = SMOLTTS_EPI_RESID; a.elu_out = 1;
      a.resid_dev = p.xraw[i]; a.ldr = C;
      a.out_dev = p.yelu[i] + (size_t)p.left[i] * C; a.ldo = C;
      ST_TRY(launch_gemm(a, st));
    }
    {  // strided conv: output row t reads input rows [t*r, t*r + 2r)
      SmolttsGemmArgs a = mimi_gemm_f32(A + cs.w, p.yelu[i], (long)r * C, p.T[i + 1], 2 * C, 2 * r * C);
      a.bias_dev = (const float*)(A + cs.b); a.epilogue = SMOLTTS_EPI_STORE; a.elu_out = 1; a.ldo = 2 * C;
      if (i + 1 < NSTAGE) {
        a.out_dev = p.xelu[i + 1] + (size_t)2 * (2 * C);
        a.raw_out_dev = p.xraw[i + 1];
      } else {
        a.out_dev = p.zelu + (size_t)2 * 1024;
      }
      ST_TRY(launch_gemm(a, st));
    }
  }
  {  // final conv k3: 1024 -> 512, raw output = transformer input
    const SmolttsMimiConv& cf = e->w.convs[12];
    SmolttsGemmArgs a = mimi_gemm_f32(A + cf.w, p.zelu, 1024, T4, D, 3 * 1024);
    a.bias_dev = (const float*)(A + cf.b); a.epilogue = SMOLTTS_EPI_STORE; a.out_dev = p.tx; a.ldo = D;
    ST_TRY(launch_gemm(a, st));
  }

  // 2. encoder transformer; its last layer writes behind the downsample conv's padding rows
  ST_TRY(launch_mimi_rows(T4, T4, 0, p.row_pos, p.row_slot, st));
  {
    MimiTransformerBufs tb{p.tx, p.tn, p.tq, p.ta, p.th, p.kc, p.vc, (size_t)HEADS * T4 * 64, p.row_pos, p.row_slot};
    ST_TRY(run_mimi_transformer(A, e->w.layers, c.n_layers, (const float*)(A + e->w.rope), T4, c.window, tb, T4, T4,
                                p.ds + (size_t)p.ds_left * D, 0, st));
  }

  // 3. downsample: Conv1d(512 -> 512, k4, s2, no bias) with edge padding
  {
    const int right = c.extra_right ? p.ds_extra : 0;
    hipLaunchKernelGGL(edge_fill_kernel, dim3(p.ds_left + right), dim3(128), 0, st, p.ds, p.ds_left, T4, right);
    ST_CHECK_HIP(hipGetLastError());
    float* emb = emb_dev ? emb_dev : p.emb;
    SmolttsGemmArgs a = mimi_gemm_f32(A + e->w.downsample_w, p.ds, 2 * D, p.F, D, 4 * D);
    a.epilogue = SMOLTTS_EPI_STORE; a.out_dev = emb; a.ldo = D;
    ST_TRY(launch_gemm(a, st));

    // 4. split RVQ encode: the semantic group and the acoustic group both start from the latents
    for (int q = 0; q < c.num_codebooks; ++q) {
      if (q <= 1) {
        SmolttsGemmArgs g = mimi_gemm_f32(A + e->w.in_proj[q], emb, D, p.F, CB_DIM, D);
        g.epilogue = SMOLTTS_EPI_STORE; g.out_dev = p.res; g.ldo = CB_DIM;
        ST_TRY(launch_gemm(g, st));
      }
      SmolttsGemmArgs g = mimi_gemm_f32(A + e->w.codebooks_t + (size_t)q * CB_SIZE * CB_DIM * 4, p.res, CB_DIM, p.F, CB_SIZE, CB_DIM);
      g.epilogue = SMOLTTS_EPI_STORE; g.out_dev = p.dots; g.ldo = CB_SIZE;
      ST_TRY(launch_gemm(g, st));
      hipLaunchKernelGGL(rvq_pick_kernel, dim3(p.F), dim3(256), 0, st, p.dots, (const float*)(A + e->w.codebook_sq) + (size_t)q * CB_SIZE,
                         (const float*)(A + e->w.codebooks) + (size_t)q * CB_SIZE * CB_DIM, p.res, codes_dev + (size_t)q * p.F,
                         gap_dev ? gap_dev + (size_t)q * p.F : nullptr);
      ST_CHECK_HIP(hipGetLastError());
    }
  }
  return SMOLTTS_OK;
}

}  // extern "C"
