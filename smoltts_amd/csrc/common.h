// Shared helpers for the gfx950 kernels and the C-ABI glue.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <string>

#include "../../include/smoltts_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace smoltts {

void set_error(const char* fmt, ...);

#define ST_CHECK_HIP(expr)                                                                \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      smoltts::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,               \
                         hipGetErrorString(_e));                                          \
      return SMOLTTS_E_HIP;                                                               \
    }                                                                                     \
  } while (0)

#define ST_REQUIRE(cond, code, ...)                                                       \
  do {                                                                                    \
    if (!(cond)) {                                                                        \
      smoltts::set_error(__VA_ARGS__);                                                    \
      return (code);                                                                      \
    }                                                                                     \
  } while (0)

#define ST_TRY(expr)                                                                      \
  do {                                                                                    \
    int _r = (expr);                                                                      \
    if (_r != SMOLTTS_OK) return _r;                                                      \
  } while (0)

// Per-device "first time" flags of the launchers: a kernel's MaxDynamicSharedMemorySize attribute and the CU count are
// properties of a DEVICE, so a process that moves to a second GPU must set / read them again there.
struct PerDevice {
  static constexpr int kMax = 64;
  std::atomic<unsigned long long> done_mask{0};  // bit d: done on device d.  Set only AFTER the guarded call has succeeded (mark_done):
                                                 // a second host thread (the scheduler's codec thread beside another session) either sees
                                                 // the bit and the finished call, or repeats the call -- which is idempotent -- itself; a
                                                 // failed call is retried by the next launch
  int value[kMax] = {0};
  static int current() { int d = 0; (void)hipGetDevice(&d); return d < 0 ? 0 : (d >= kMax ? kMax - 1 : d); }
  bool done(int dev) const { return (done_mask.load(std::memory_order_acquire) >> dev) & 1ull; }
  void mark_done(int dev) { done_mask.fetch_or(1ull << dev, std::memory_order_release); }
};
inline int device_cu_count() {  // CUs of the current device (cached per device)
  static PerDevice cache;
  const int d = PerDevice::current();
  if (!cache.done(d)) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
    cache.value[d] = n;
    cache.mark_done(d);
  }
  return cache.value[d];
}

// Experiment switches of tools/ (forced tile shapes, kernels switched off) exist only in builds with -DSMOLTTS_DBG_KNOBS
// (python -m smoltts_amd.build --variant knobs): the product library reads no such environment variable.
#ifdef SMOLTTS_DBG_KNOBS
#define ST_KNOB_INT(name, dflt) ([] { const char* e_ = getenv(name); return e_ ? atoi(e_) : (dflt); }())
#else
#define ST_KNOB_INT(name, dflt) (dflt)
#endif

// ---- device helpers
__device__ __forceinline__ float bf16_lo(uint32_t dw) { return __uint_as_float(dw << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t dw) { return __uint_as_float(dw & 0xffff0000u); }

// The value of lane ^ O, on the VALU: DPP inside a 16-lane row (O = 1, 2, 4, 8), v_permlane16_swap / v_permlane32_swap across
// rows (gfx950).  __shfl_xor compiles to ds_bpermute_b32 -- a round trip through the LDS crossbar per step -- and the frame's
// kernels end on two or four dependent ones (sums of squares, top-2 merges).  Exact lane ^ O in every case, so a reduction
// keeps its order of additions.
template <int O>
__device__ __forceinline__ unsigned lane_xor_u(unsigned x, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (O == 32) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);  // lower half: (own, partner's); upper: (partner's, own)
    return (lane & 32) ? r[0] : r[1];
  } else if constexpr (O == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    return (lane & 16) ? r[0] : r[1];
  } else if constexpr (O == 8) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xF, 0xF, true);  // row_ror:8
  } else if constexpr (O == 4) {
    const int y = __builtin_amdgcn_update_dpp(0, (int)x, 0x1B, 0xF, 0xF, true);  // quad_perm [3,2,1,0]: lane ^ 3
    return (unsigned)__builtin_amdgcn_update_dpp(0, y, 0x141, 0xF, 0xF, true);   // row_half_mirror: lane ^ 7
  } else if constexpr (O == 2) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
  } else {
    static_assert(O == 1, "lane_xor: O must be a power of two below 64");
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
  }
#else
  return x;
#endif
}
template <int O>
__device__ __forceinline__ float lane_xor(float x, int lane) { return __uint_as_float(lane_xor_u<O>(__float_as_uint(x), lane)); }
template <int O>
__device__ __forceinline__ int lane_xor(int x, int lane) { return (int)lane_xor_u<O>((unsigned)x, lane); }
__device__ __forceinline__ float lane_xor_c(float x, int o, int lane) {  // o: a constant once the caller's loop is unrolled
  switch (o) {
    case 32: return lane_xor<32>(x, lane);
    case 16: return lane_xor<16>(x, lane);
    case 8: return lane_xor<8>(x, lane);
    case 4: return lane_xor<4>(x, lane);
    case 2: return lane_xor<2>(x, lane);
    default: return lane_xor<1>(x, lane);
  }
}
// x + (lane ^ 16's x) + ... in the order `x += shfl_xor(x, 16); x += shfl_xor(x, 32)` (a + b == b + a: no lane select needed)
__device__ __forceinline__ float sum_xor16_32(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(b[0]) + __uint_as_float(b[1]);
#endif
  return x;
}

// The kernel arguments of a frame's launches sit in memory nothing has touched since the graph's last replay: every 64-byte line
// of them a kernel reads is a miss of its own (to the fabric: ~0.5 us), and the compiler asks for each where it is first used,
// one after the other.  kernarg_touch() asks for one dword of each of the lines 1 .. LINES at the kernel's first instruction, all
// misses in flight at once (line 0 holds the leading scalar arguments, which arrive in SGPRs: build.py's
// -amdgpu-kernarg-preload-count); kernarg_touched() ends the dwords' lifetime behind an s_waitcnt.  The later reads of the
// compiler's own code hit the scalar cache.
template <int LINES>
struct KernargTouch {
  unsigned t[LINES];
};
template <int LINES>
__device__ __forceinline__ void kernarg_touch(KernargTouch<LINES>& k) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
#pragma unroll
  for (int i = 0; i < LINES; ++i) asm volatile("s_load_dword %0, %1, %2" : "=s"(k.t[i]) : "s"(ka), "i"(64 * (i + 1)));
#endif
}
template <int LINES>
__device__ __forceinline__ void kernarg_touched(KernargTouch<LINES>& k) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < LINES; ++i) asm volatile("" : "+s"(k.t[i]));  // the registers stay reserved up to here
#endif
}

// Host-side description of where a producer of the residual stream publishes it for the next GEMM(s)
// (device form: EmitDev in x3.h).
struct EmitArgs {
  void* x3a;
  const float* gamma_a;
  void* x3b;
  const float* gamma_b;
  float* ssq;
};

// Sampling of one row of logits (temp <= 0: greedy). The random stream is a pure function of
// (seed, row, frame, step, column); frame = frames[row] when given, else frame_base + row.
struct SampleArgs {
  float temp;
  float min_p;
  uint64_t seed;
  int step;
  int frame_base;
  const int32_t* frames;
  const int32_t* salt;  // per-row generation counter mixed into the seed (a slot's n-th tenant draws its own numbers), or null
  int32_t* margin_at;   // greedy only, optional: [rows] receives frame * 64 + step where the row's smallest top-2 gap occurred
};

// Launchers implemented across the .hip files (all asynchronous on `stream`).
int launch_gemm(const SmolttsGemmArgs& a, hipStream_t stream);
int launch_gemm3(const SmolttsGemm3Args& a, hipStream_t stream);
int launch_x3_pack(const float* x, int64_t ldx, int n_rows, int dim, void* x3a, const float* gamma_a, void* x3b,
                   const float* gamma_b, float* ssq, hipStream_t stream);
// kc / vc: [slot][kv head][cache_len][64] in fp32 (kv_format SMOLTTS_KV_F32) or bf16 (SMOLTTS_KV_BF16; caches of more than 16 entries only)
int launch_attention(const float* q, const void* kc, const void* vc, const int32_t* row_pos,
                     const int32_t* row_slot, int n_rows, int n_q_heads, int n_kv_heads,
                     int cache_len, int window, float* out, void* out_x3, hipStream_t stream,
                     int kv_format = SMOLTTS_KV_F32, int iota_pos = -1,  // iota_pos >= 0 (caches <= 16): row r = slot r at that position
                     float* split_part = nullptr, int32_t* split_ticket = nullptr);
int launch_attention_rows3(const float* q, const void* kc3, const void* vc3, const int32_t* row_pos, const int32_t* row_slot, int n_rows,
                           int rows_per_slot, int n_heads, int cache_len, int window, float* out, hipStream_t stream,
                           int b3_products = 6);  // attention.hip; b3_products = 3: three products per operand pair (gemm_dev.h, mfma_b3)  // key-split decode attention (attention.hip)
// scratch of the key-split attention: records for this many (row, kv head) pairs x 2 parts x (4 heads x 64 + 8) floats, one ticket per pair
constexpr int ATT_SPLIT_MAX_PAIRS = 128;
constexpr size_t ATT_SPLIT_PART_FLOATS = (size_t)ATT_SPLIT_MAX_PAIRS * 2 * (4 * 64 + 8);
int launch_embed(const int32_t* cols, int n_rows, int n_code_rows, const void* text_emb,
                 const void* cb_emb, int dim, int codebook_size, int cb_first_offset, int mask_mode,
                 int sem_start, int sem_end, int text_rows, int cb_rows, float* x, const EmitArgs* emit,
                 hipStream_t stream);
// argmax over each row; writes ids[r*ids_stride]; optionally (emb != nullptr) gathers
// emb[(id + emb_row_offset)] (bf16 row-major, `dim` wide) into xnext[r].
struct QkvGather;  // argmax_dev.h
int launch_argmax(const float* logits, int n_rows, int n_cols, int64_t ld, int32_t* ids,
                  int ids_stride, float* margin, const int32_t* margin_mask, const void* emb,
                  int emb_row_offset, int dim, float* xnext, const EmitArgs* emit, const SampleArgs* sample,
                  hipStream_t stream, const QkvGather* qkv = nullptr);
// rows [row0, row0 + n_rows) of a bf16 embedding table -> X3 operand (x gamma) + sums of squares
int launch_emb_rows_pack(const void* emb, int64_t row0, int n_rows, int dim, const EmitArgs& emit, hipStream_t stream);
int launch_layernorm(const float* x, const float* w, const float* b, int n_rows, int dim, float eps,
                     float* out, hipStream_t stream);
int launch_gather_rows(const float* src, const int32_t* idx, int n, int dim, float* dst,
                       hipStream_t stream);

}  // namespace smoltts
