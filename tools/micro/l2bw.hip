// Micro-probe: L2 -> CU read bandwidth on MI355X.  Every workgroup streams the same `window` bytes (L2 resident when small)
// `reps` times with 16-B buffer loads; prints aggregate TB/s for a few windows / workgroup counts / waves per CU.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/l2bw.hip -o /tmp/l2bw && /tmp/l2bw
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int UNROLL>
__global__ __launch_bounds__(256) void rd(const char* p, unsigned window, int reps, unsigned* out, int mode) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
  u32x4 acc = {0, 0, 0, 0};
  // mode 0: all workgroups read the same window; mode 1: each workgroup reads its own window (offset by blockIdx)
  unsigned base = mode ? blockIdx.x * window : 0;
  for (int it = 0; it < reps; ++it) {
    for (unsigned o = threadIdx.x * 16; o < window; o += 256 * 16 * UNROLL) {
      u32x4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(r, base + o + u * 256 * 16, 0, 0);
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
    }
  }
  if (acc.x == 0x12345 && acc.y == 77) out[0] = acc.z + acc.w;
}
int main() {
  char* p; unsigned* out;
  size_t total = 1ull << 30;
  hipMalloc(&p, total); hipMemset(p, 1, total); hipMalloc(&out, 64);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 2; ++mode)
    for (unsigned window : {64u << 10, 512u << 10, 2u << 20}) {
      for (int blocks : {256, 512, 1024}) {
        if (mode == 1 && (size_t)blocks * window > total) continue;
        int reps = (int)((64u << 20) / window); if (reps < 1) reps = 1;
        rd<8><<<blocks, 256>>>(p, window, 2, out, mode);
        hipEventRecord(a);
        rd<8><<<blocks, 256>>>(p, window, reps, out, mode);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double bytes = (double)blocks * window * reps;
        printf("mode %d window %5u KB blocks %4d: %.2f TB/s (%.1f us)\n", mode, window >> 10, blocks, bytes / ms / 1e9, ms * 1e3);
      }
    }
  return 0;
}
