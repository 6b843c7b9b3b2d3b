// Micro-probe: do the global -> LDS DMA path (buffer_load ... lds) and the global -> VGPR path (buffer_load_dwordx4) of a CU share one
// limit, or do they add up?  One 512-thread workgroup per CU streams an L2-resident window: wavefronts 0-3 by LDS-DMA pieces into a
// scratch ring in LDS, wavefronts 4-7 by 16-byte loads into registers.  Three runs: DMA wavefronts only, VGPR wavefronts only, both.
// Prints bytes per clock per CU for each role (clock: the 2.1 GHz the chip sustains under load is assumed only for the B/clk column;
// the TB/s column is measured).  Motivation: the K loops of the 128 x 160 GEMM / conv kernels are paced by ~40 B/clk/CU of LDS-DMA
// (third session of round 3); if the register path adds bandwidth, one operand could bypass the LDS.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/dma_vs_vgpr.hip -o /tmp/dma_vs_vgpr && /tmp/dma_vs_vgpr
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// roles: bit 0 = DMA wavefronts active, bit 1 = VGPR wavefronts active.  nd / nv = wavefronts per role (1..4)
// out[1] / out[2]: longest DMA / VGPR wavefront in clock64 ticks (100 MHz on this chip: 10 ns)
__global__ __launch_bounds__(512) void probe(const char* p, unsigned window, int reps, unsigned* out, int roles, int nd, int nv) {
  __shared__ __attribute__((aligned(1024))) char ring[64 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(p + (size_t)blockIdx.x * window * 2), 0, 0x7fffffff, 0x00020000);
  if (wave < 4) {
    if (!(roles & 1) || wave >= nd) return;
    const long long t0 = wall_clock64();
    // each DMA wavefront walks its share of the window in 1 KB pieces (64 lanes x 16 B), 16 pieces in flight
    for (int it = 0; it < reps; ++it) {
      for (unsigned o = wave * 1024u; o < window; o += (unsigned)nd * 1024u * 16u) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const unsigned off = o + (unsigned)u * nd * 1024u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(ring + ((wave * 16 + u) << 10)), 16,
                                                   (off < window ? off : 0u) + lane * 16u, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) atomicMax(out + 1, (unsigned)(wall_clock64() - t0));
  } else {
    const int w = wave - 4;
    if (!(roles & 2) || w >= nv) return;
    u32x4 acc = {0, 0, 0, 0};
    const long long t0 = wall_clock64();
    for (int it = 0; it < reps; ++it) {
      for (unsigned o = w * 1024u; o < window; o += (unsigned)nv * 1024u * 8u) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const unsigned off = o + (unsigned)u * nv * 1024u;
          v[u] = __builtin_amdgcn_raw_buffer_load_b128(r, window + (off < window ? off : 0u) + lane * 16u, 0, 0);   // the second half of the block's region
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
      }
    }
    if (acc.x == 0x12345 && acc.y == 77) out[0] = acc.z + acc.w;
    if (lane == 0) atomicMax(out + 2, (unsigned)(wall_clock64() - t0));
  }
}

int main() {
  char* p; unsigned* out;
  const unsigned window = 64u << 10;   // per role and workgroup: 2 x 64 KB x 256 workgroups = 32 MB over 8 L2s of 4 MB
  const int blocks = 256;
  size_t total = (size_t)blocks * window * 2;
  hipMalloc(&p, total); hipMemset(p, 1, total); hipMalloc(&out, 64);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int reps = 400;
  const double clk = 2.1e9;
  for (int nd = 1; nd <= 4; nd *= 2)
    for (int nv = 1; nv <= 4; nv *= 2)
      for (int roles = 1; roles <= 3; ++roles) {
        if ((roles == 1 && nv != 1) || (roles == 2 && nd != 1)) continue;   // single-role runs once per count
        probe<<<blocks, 512>>>(p, window, 4, out, roles, nd, nv);
        hipMemset(out, 0, 64);
        hipEventRecord(a);
        probe<<<blocks, 512>>>(p, window, reps, out, roles, nd, nv);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double per_role = (double)blocks * window * reps;
        const int nroles = (roles & 1) + ((roles >> 1) & 1);
        const double tbs = per_role * nroles / ms / 1e9;
        unsigned h[4];
        hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        const double td = h[1] * 1e-8, tv = h[2] * 1e-8;   // seconds (wall_clock64: 100 MHz)
        const double bd = (roles & 1) ? (double)window * reps / td / clk : 0.0, bv = (roles & 2) ? (double)window * reps / tv / clk : 0.0;
        printf("roles %s  dma waves %d  vgpr waves %d: kernel %7.1f us (%.2f TB/s)  DMA role %5.1f B/clk/CU over %6.1f us   VGPR role %5.1f B/clk/CU over %6.1f us   sum %.1f\n",
               roles == 1 ? "DMA     " : roles == 2 ? "VGPR    " : "DMA+VGPR", (roles & 1) ? nd : 0, (roles & 2) ? nv : 0, ms * 1e3, tbs, bd,
               td * 1e6, bv, tv * 1e6, bd + bv);
        (void)nroles;
      }
  return 0;
}
