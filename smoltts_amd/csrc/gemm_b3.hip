// Many-row GEMM of the Mimi decoder on the bf16 matrix cores with fp32-grade results:
//   out[M,N] = epi( X[M,K] . W[N,K]^T ),  X fp32 rows (channel-last conv windows / Linear inputs), W fp32 weights.
//
// Reference ops: the Linear / Conv1d / ConvTranspose1d layers of mlx_inference/src/smoltts_mlx/codec/
// {transformer.py:34-96, conv.py:68-220, seanet.py:99-161}, all fp32 (codec/mimi.py:107,149).
//
// Why: the fp32-input MFMA (v_mfma_f32_16x16x4_f32, gemm.hip) runs at 1/16 of the bf16 rate, and the SEANet
// decoder is 337 GFLOP per 1024 frames -- 2.2 ms at that rate before anything else.  Here both operands are split
// into three bf16 pieces (hi + mid + lo == the fp32 value exactly): the weights once on the host ("W3" tiles), the
// activations when a chunk is staged into LDS, and six bf16 MFMAs per 32-k chunk (v_mfma_f32_16x16x32_bf16) replace
// eight fp32 ones: 96 instead of 256 matrix-pipe cycles for the same 16x16x32 block, exact products, fp32 accumulation;
// the three dropped cross terms are below 2^-24 relative (gemm_dev.h).
//
// Workgroup = 4 waves in a 2 x 2 arrangement, wave tile TM x TN MFMA tiles: 32 TM rows x 32 TN columns per
// workgroup.  Per 32-k chunk the workgroup stages its X rows (split into pieces, B-fragment order) and its W3 tiles
// (already in A-fragment order) in LDS once; every wave reads TM + TN fragments triples and issues 6 TM TN MFMAs;
// the next chunk's global loads fly under them (register prefetch, one LDS buffer, two barriers per chunk).
#include <stdlib.h>

#include "gemm_dev.h"

namespace smoltts {

namespace {
// LDS image of the X chunk: [piece][q][row] 16-byte slots; within a 16-row block, row r of plane q sits at slot
// r ^ XS_MASK[q].  The masks map the ds_read_b128 lane-group row sets {0-3,12-15} / {4-11} onto themselves, so the
// fragment reads (16 rows x one q per lane group half) and the staging writes (2 rows x 4 q per 8 lanes) are both
// bank-conflict free (MI355X_MICROARCH.md, LDS table).
__device__ __forceinline__ int xs_slot(int pc, int q, int row, int BM) {
  const int mask = q == 0 ? 0 : (q == 1 ? 3 : (q == 2 ? 12 : 15));
  return (pc * 4 + q) * BM + (row & ~15) + ((row & 15) ^ mask);
}
__device__ __forceinline__ float elu_hw1(float x) { return x > 0.f ? x : __expf(x) - 1.0f; }
}  // namespace

template <int TM, int TN, int EPI, int NP>
__global__ __launch_bounds__(256) void gemm_b3_kernel(GemmDev p) {
  constexpr int BM = 32 * TM, BN = 32 * TN, NT = BN / 16;
  constexpr int XI = BM * 4 / 256;       // (row, q) pairs of the X chunk per thread
  constexpr int WI = NT * 3 * 64 / 256;  // 16-byte pieces of the W3 chunk per thread
  __shared__ uint4 xs[3 * 4 * BM];
  __shared__ uint4 ws[NT * 3 * 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int nchunks = p.K >> 5;
  // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (id % 8), each with its own L2.  The column
  // blocks of one row block all read the same X rows: they get consecutive slots on ONE XCD (row blocks are dealt over the
  // XCDs instead), so X leaves HBM / the Infinity Cache once, not once per column block (rocprofv3 FETCH_SIZE of the
  // 256 -> 128 ConvTranspose, 5 column blocks: 1.15 GB per 1024 frames with the plain 2-D grid against 0.1 GB of X).
  // Where W is the larger operand (conv0, the first ConvTranspose: 22 / 50 MB of W3 against 4 / 16 MB of X) the roles swap
  // (xcd_cols): column blocks are dealt over the XCDs, so each XCD reads 1/8 of W and all of X instead of all of W.
  const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
  // split-K (long K, few tiles: the transformer's fc2 at M = 2048): `ksplit` workgroups per tile take consecutive chunk ranges
  // and store plain partial sums; splitk_reduce_kernel adds them in fixed order and applies the epilogue
  const int ks = p.ksplit > 1 ? kk % p.ksplit : 0, kt = p.ksplit > 1 ? kk / p.ksplit : kk;
  const int cb = p.xcd_cols ? (kt / p.grid_rb) * 8 + xcd : kt % p.grid_cb;
  const int rb = p.xcd_cols ? kt % p.grid_rb : (kt / p.grid_cb) * 8 + xcd;
  if (rb >= p.grid_rb || cb >= p.grid_cb) return;  // the grid is padded to whole groups of 8 blocks (uniform over the workgroup)
  const int row0 = rb * BM, tile0 = cb * NT;
  const int nch_all = p.K >> 5;
  const int c_lo = p.ksplit > 1 ? (int)((long)nch_all * ks / p.ksplit) : 0, c_hi = p.ksplit > 1 ? (int)((long)nch_all * (ks + 1) / p.ksplit) : nch_all;

  const float* xsrc[XI];
  int xdst[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int idx = tid + 256 * i, row = idx >> 2, qq = idx & 3;
    const int m = row0 + row;
    xsrc[i] = m < p.M ? p.x + row_off(m, p.rows_per_batch, p.ldx, p.x_bstride) + qq * 8 : nullptr;
    xdst[i] = xs_slot(0, qq, row, BM);
  }
  const char* wsrc[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int j = tid + 256 * i, t = j / 192, rem = j - t * 192;
    wsrc[i] = (tile0 + t) * 16 < p.N ? p.w3 + (size_t)(tile0 + t) * nchunks * 3072 + rem * 16 : nullptr;
  }

  f32x4 acc[TN][TM];
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 xr[XI][2];
  uint4 wr[WI];
  // chunk number -> 32-k chunk of K.  Conv windows overlap (row m's tap t + 1 is row m + 1's tap t): taken in K order the
  // same X bytes come back `taps` times, K / taps chunks apart -- long gone from L1 and, with 64 workgroups per XCD, from
  // L2.  Visiting the taps of one channel slice back to back makes the repeats hits.
  auto kchunk = [&](int c) { return p.taps > 1 ? (c % p.taps) * p.cpt + c / p.taps : c; };
  auto fetch = [&](int cn) {
    const int c = kchunk(cn);
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      xr[i][0] = xsrc[i] ? *reinterpret_cast<const float4*>(xsrc[i] + c * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
      xr[i][1] = xsrc[i] ? *reinterpret_cast<const float4*>(xsrc[i] + c * 32 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) wr[i] = wsrc[i] ? *reinterpret_cast<const uint4*>(wsrc[i] + (size_t)c * 3072) : make_uint4(0, 0, 0, 0);
  };
  fetch(c_lo);
  for (int c = c_lo; c < c_hi; ++c) {
    __syncthreads();  // the previous chunk has been consumed
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      uint4 h, m, l;
      if (p.pro_elu) {  // the hardware exponential: see seanet.hip (absolute error ~6e-8 on O(1) activations)
        float4& a = xr[i][0];
        float4& b = xr[i][1];
        a = make_float4(elu_hw1(a.x), elu_hw1(a.y), elu_hw1(a.z), elu_hw1(a.w));
        b = make_float4(elu_hw1(b.x), elu_hw1(b.y), elu_hw1(b.z), elu_hw1(b.w));
      }
      split3x8(xr[i][0], xr[i][1], h, m, l);
      xs[xdst[i]] = h;
      xs[xdst[i] + 4 * BM] = m;
      xs[xdst[i] + 8 * BM] = l;
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) ws[tid + 256 * i] = wr[i];
    __syncthreads();
    if (c + 1 < c_hi) fetch(c + 1);  // the next chunk's global loads fly under this chunk's MFMAs
    uint4 xf[TM][3];
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) xf[mt][pc] = xs[xs_slot(pc, q, (wm * TM + mt) * 16 + r, BM)];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      uint4 wf[3];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) wf[pc] = ws[((wn * TN + t) * 3 + pc) * 64 + lane];
#pragma unroll
      for (int mt = 0; mt < TM; ++mt) acc[t][mt] = mfma_b3<NP>(wf, xf[mt], acc[t][mt]);
    }
  }

  // ---- epilogue straight from the accumulators: the lane holds out[m = tile row r][n0 .. n0 + 4)
#pragma unroll
  for (int mt = 0; mt < TM; ++mt) {
    const int m = row0 + (wm * TM + mt) * 16 + r;
    if (m >= p.M) continue;
    const long orow = row_off(m, p.rows_per_batch, p.ldo, p.o_bstride);
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      const int n0 = (tile0 + wn * TN + t) * 16 + q * 4;
      if (p.ksplit > 1) {  // (N % 4 == 0 on this path)
        if (n0 < p.N) *reinterpret_cast<float4*>(p.splitk_ws + ((long)ks * p.M + m) * p.N + n0) = make_float4(acc[t][mt][0], acc[t][mt][1], acc[t][mt][2], acc[t][mt][3]);
        continue;
      }
      float v[4] = {acc[t][mt][0], acc[t][mt][1], acc[t][mt][2], acc[t][mt][3]};
      rows_epilogue<EPI>(p, m, orow, n0, v);
    }
  }
}

// Tile choice, measured on the 16 GEMM shapes of a 1024-frame chunk (tools/microbench_b3.py with SMOLTTS_B3_TILE forced):
// 128 x 128 tiles win once they make >= 2 workgroups per CU (ConvTranspose stages); below that the 64 x 64 tiles win although
// they re-read more -- at M = 2048 (conv0, the transformer Linears) what counts is workgroups in flight per CU (5 fit), e.g.
// conv0 188 -> 137 us, qkv 41 -> 34 us, fc1 48 -> 40 us; 128 x 64 only serves the narrow outputs (N < 128) of long tensors.
template <int EPI, int NP>
static int launch_b3_epi_np(const GemmDev& d, hipStream_t stream) {
  auto blocks = [&](int bm, int bn) { return (long)((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn); };
  GemmDev g = d;
  static const int env_taps = ST_KNOB_INT("SMOLTTS_B3_TAPS", 1);  // experiments (knobs builds only): 0 = K order
  static const int env_xcd = ST_KNOB_INT("SMOLTTS_B3_XCD", -1);   // experiments: 0 rows | 1 columns
  g.taps = 1; g.cpt = d.K >> 5;
  if (env_taps && d.ldx < d.K && d.ldx % 32 == 0 && d.K % d.ldx == 0) { g.taps = (int)(d.K / d.ldx); g.cpt = (int)(d.ldx >> 5); }
  const double x_bytes = 4.0 * d.M * (d.ldx < d.K ? d.ldx : d.K), w_bytes = 6.0 * d.N * d.K;
  auto grid1d = [&](int bm, int bn) {  // 8 consecutive ids = 8 XCDs = 8 different row (or column) blocks, see the kernel
    g.grid_rb = (d.M + bm - 1) / bm;
    g.grid_cb = (d.N + bn - 1) / bn;
    const double by_rows = x_bytes + (g.grid_rb < 8 ? g.grid_rb : 8) * w_bytes, by_cols = (g.grid_cb < 8 ? g.grid_cb : 8) * x_bytes + w_bytes;
    // (a dimension dealt over the XCDs must fill them evenly: whole groups of 8 column blocks)
    g.xcd_cols = env_xcd >= 0 ? env_xcd : (g.grid_cb % 8 == 0 && by_cols < by_rows);
    const long blocks8 = g.xcd_cols ? (long)((g.grid_cb + 7) / 8) * 8 * g.grid_rb : (long)((g.grid_rb + 7) / 8) * 8 * g.grid_cb;
    return dim3((unsigned)(blocks8 * g.ksplit));
  };
  ST_REQUIRE(blocks(64, 64) < (1L << 30), SMOLTTS_E_INVALID, "gemm_b3: M=%d too large for one launch", d.M);
  g.ksplit = 1;
  // long K over few tiles (fc2 of the decoder transformer at M = 2048: 256 tiles x 64 chunks of 0.2 us MFMA each -- one
  // workgroup per CU, every chunk a full memory round trip): four workgroups per tile, 16 chunks each, + a 4 MB reduce pass
  if (d.splitk_ws && d.N % 4 == 0 && blocks(64, 64) <= 512 && (d.K >> 5) >= 48 && 4L * d.M * d.N <= d.splitk_cap) {
    g.ksplit = 4;
    const dim3 grid = grid1d(64, 64);
    hipLaunchKernelGGL((gemm_b3_kernel<2, 2, EPI, NP>), grid, dim3(256), 0, stream, g);
    ST_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL((splitk_reduce_kernel<EPI>), dim3((unsigned)(((long)d.M * (d.N >> 2) + 255) / 256)), dim3(256), 0, stream, g);
    ST_CHECK_HIP(hipGetLastError());
    return SMOLTTS_OK;
  }
  static const int force = ST_KNOB_INT("SMOLTTS_B3_TILE", 0);  // experiments (knobs builds only): 44 | 42 | 22
  if (force == 44 || (force == 0 && blocks(128, 128) >= 512 && d.N >= 128)) {
    const dim3 grid = grid1d(128, 128);
    hipLaunchKernelGGL((gemm_b3_kernel<4, 4, EPI, NP>), grid, dim3(256), 0, stream, g);
  } else if (force == 42 || (force == 0 && blocks(128, 64) >= 512 && d.N < 128)) {
    const dim3 grid = grid1d(128, 64);
    hipLaunchKernelGGL((gemm_b3_kernel<4, 2, EPI, NP>), grid, dim3(256), 0, stream, g);
  } else {
    const dim3 grid = grid1d(64, 64);
    hipLaunchKernelGGL((gemm_b3_kernel<2, 2, EPI, NP>), grid, dim3(256), 0, stream, g);
  }
  ST_CHECK_HIP(hipGetLastError());
  return SMOLTTS_OK;
}

template <int EPI>
static int launch_b3_epi(const GemmDev& d, hipStream_t stream) {
  return d.b3_products == 3 ? launch_b3_epi_np<EPI, 3>(d, stream) : launch_b3_epi_np<EPI, 6>(d, stream);
}

bool gemm_b3_applies(int M, int N, int K, int epilogue) {
  if (M < 256 || N < 64 || N % 4 != 0 || K % 32 != 0) return false;
  // no split-K here: with fewer than ~3/4 of the CUs busy (streaming passes of a few frames) the K-split skinny kernel of
  // gemm.hip has the shorter critical path
  if ((long)((M + 63) / 64) * ((N + 63) / 64) < 192) return false;
  return epilogue == SMOLTTS_EPI_STORE || epilogue == SMOLTTS_EPI_RESID || epilogue == SMOLTTS_EPI_GELU ||
         epilogue == SMOLTTS_EPI_SCALE_RESID || epilogue == SMOLTTS_EPI_QKV_ROPE;
}

int launch_gemm_b3(const GemmDev& d, int epilogue, hipStream_t stream) {
  ST_REQUIRE(d.w3 && gemm_b3_applies(d.M, d.N, d.K, epilogue), SMOLTTS_E_INVALID, "gemm_b3: shape or epilogue not supported");
  if (conv_xs_applies(d, epilogue)) return launch_conv_xs(d, epilogue, stream);
  if (conv_ks_applies(d, epilogue)) return launch_conv_ks(d, epilogue, stream);
  switch (epilogue) {
    case SMOLTTS_EPI_STORE: return launch_b3_epi<SMOLTTS_EPI_STORE>(d, stream);
    case SMOLTTS_EPI_RESID: return launch_b3_epi<SMOLTTS_EPI_RESID>(d, stream);
    case SMOLTTS_EPI_GELU: return launch_b3_epi<SMOLTTS_EPI_GELU>(d, stream);
    case SMOLTTS_EPI_SCALE_RESID: return launch_b3_epi<SMOLTTS_EPI_SCALE_RESID>(d, stream);
    default: return launch_b3_epi<SMOLTTS_EPI_QKV_ROPE>(d, stream);
  }
}

}  // namespace smoltts
