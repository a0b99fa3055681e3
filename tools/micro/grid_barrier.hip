// What does a grid-wide barrier cost on MI355X (256 co-resident workgroups, 8 XCDs)?  The frame graph pays ~1.7 us of dependent-launch
// gap per kernel node; a persistent kernel would pay this instead.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/micro/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
// Every spin is bounded (a workgroup that never sees the count gives up and flags it), so the kernel always drains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int FENCE>
__global__ __launch_bounds__(512) void barrier_loop(unsigned* counter, int n_barriers, unsigned* gave_up, float* sink) {
  float acc = threadIdx.x;
  __shared__ int s_dead;
  if (threadIdx.x == 0) s_dead = 0;
  for (int k = 0; k < n_barriers; ++k) {
    acc = acc * 1.0001f + 1.f;  // a token of work between barriers
    __syncthreads();
    if (threadIdx.x == 0) {
      if (FENCE) __threadfence();  // release: this workgroup's global writes are visible device-wide
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(k + 1) * gridDim.x;
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > 2000000) { atomicAdd(gave_up, 1u); s_dead = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (FENCE) __threadfence();  // acquire
    }
    __syncthreads();
    if (s_dead) return;  // (every other workgroup then gives up at this barrier too: at most one bounded spin each)
  }
  if (acc == 12345.678f) sink[0] = acc;
}

int main() {
  unsigned *counter, *gave_up;
  float* sink;
  CK(hipMalloc(&counter, 4)); CK(hipMalloc(&gave_up, 4)); CK(hipMalloc(&sink, 4));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int N = 2000;
  for (int grid : {64, 128, 256}) {
    for (int fence = 0; fence < 2; ++fence) {
      float best = 1e30f;
      unsigned gu = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(counter, 0, 4)); CK(hipMemset(gave_up, 0, 4));
        CK(hipEventRecord(a));
        if (fence) hipLaunchKernelGGL(barrier_loop<1>, dim3(grid), dim3(512), 0, 0, counter, N, gave_up, sink);
        else hipLaunchKernelGGL(barrier_loop<0>, dim3(grid), dim3(512), 0, 0, counter, N, gave_up, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
        CK(hipMemcpy(&gu, gave_up, 4, hipMemcpyDeviceToHost));
      }
      printf("grid %3d x 512 threads, %s: %.3f us per barrier (%d barriers, gave up: %u)\n", grid, fence ? "release/acquire fences" : "no fences", best * 1e3 / N, N, gu);
    }
  }
  return 0;
}
