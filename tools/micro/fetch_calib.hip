// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of this library: a known number of bytes
// is read ONCE from a buffer far larger than the Infinity Cache (2 GiB against 256 MiB), by
//   mode 0: global_load_dwordx4            (16 B / lane, registers)
//   mode 1: buffer_load_dwordx4 ... lds    (16 B / lane, LDS-DMA: what the GEMM / conv / chain kernels' operand loads are)
//   mode 2: global_load_dwordx2            ( 8 B / lane)
//   mode 3: buffer_load_dwordx4 ... lds with the GEMM loaders' row pattern: 8 rows x 128 B per instruction, row stride 640 B
//           (64-byte halves of a line fetched by different instructions: the K-tile walk over a [M][320] bf16 activation)
// and, mode 4, the same 32 MiB re-read 64 times by mode 1 inside ONE launch (L2 hits), and, mode 5, a 128 MiB buffer read by two
// consecutive launches of mode 1 (the second launch finds it in the 256 MiB Infinity Cache but not in the 32 MiB of L2s: are
// Infinity-Cache hits counted?).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/fetch_calib.hip -o tools/micro/fetch_calib ; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/micro/fetch_calib <mode>          (tools/micro/fetch_calib.sh)
// FETCH_SIZE is reported in KiB; the known byte count is printed by the program.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xffffffffu, 0x00020000);
}

// every workgroup streams `per_wg` contiguous bytes starting at blockIdx.x * per_wg (+ base)
template <int MODE>
__global__ __launch_bounds__(256) void reader(const char* src, long per_wg, uint32_t* sink, int repeat) {
  __shared__ __attribute__((aligned(1024))) char lds[16384];
  const int tid = threadIdx.x, wave = tid >> 6;
  uint32_t acc = 0;
  for (int rep = 0; rep < repeat; ++rep) {
    const char* base = src + (long)blockIdx.x * per_wg;
    if (MODE == 0) {
      for (long off = (long)tid * 16; off < per_wg; off += 256 * 16) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(base + off);
        acc += v.x ^ v.y ^ v.z ^ v.w;
      }
    } else if (MODE == 2) {
      for (long off = (long)tid * 8; off < per_wg; off += 256 * 8) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(base + off);
        acc += v.x ^ v.y;
      }
    } else if (MODE == 1 || MODE == 4) {
      const __amdgpu_buffer_rsrc_t rs = rsrc(base);
      for (long off = 0; off < per_wg; off += 4096) {   // 4 wavefronts x 1 KiB per round
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + wave * 1024), 16,
                                                 (uint32_t)(off + wave * 1024 + (tid & 63) * 16), 0, 0, 0);
        if ((off & 0xffff) == 0xf000) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc += *reinterpret_cast<uint32_t*>(lds + tid * 4);
    } else {   // MODE 3: rows of 640 B; an instruction takes 8 rows x 128 B at K offset kt * 128 (the last K tile is a 64-byte half: masked)
      const __amdgpu_buffer_rsrc_t rs = rsrc(base);
      const long rows = per_wg / 640;
      for (long r0 = 0; r0 < rows; r0 += 32) {          // 4 wavefronts x 8 rows
        for (int kt = 0; kt < 5; ++kt) {
          const long row = r0 + wave * 8 + ((tid & 63) >> 3);
          const uint32_t off = (uint32_t)(row * 640 + kt * 128 + (tid & 7) * 16);
          const bool ok = row < rows && (kt * 128 + (tid & 7) * 16) < 640;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + wave * 1024), 16,
                                                   ok ? off : 0xffffffffu, 0, 0, 0);
        }
        if ((r0 & 255) == 224) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc += *reinterpret_cast<uint32_t*>(lds + tid * 4);
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;   // never true in practice: keeps the loads alive
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const long total = mode == 4 ? (32L << 20) : mode == 5 ? (128L << 20) : (2L << 30);
  const int repeat = mode == 4 ? 64 : 1;
  const int wgs = 2048;
  long per_wg = total / wgs;
  if (mode == 3) per_wg = per_wg / 640 * 640;
  char* buf;
  uint32_t* sink;
  hipMalloc(&buf, total + 4096);
  hipMalloc(&sink, 64);
  hipMemset(buf, 1, total + 4096);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  switch (mode) {
    case 0: hipLaunchKernelGGL(reader<0>, dim3(wgs), dim3(256), 0, 0, buf, per_wg, sink, repeat); break;
    case 1: hipLaunchKernelGGL(reader<1>, dim3(wgs), dim3(256), 0, 0, buf, per_wg, sink, repeat); break;
    case 2: hipLaunchKernelGGL(reader<2>, dim3(wgs), dim3(256), 0, 0, buf, per_wg, sink, repeat); break;
    case 3: hipLaunchKernelGGL(reader<3>, dim3(wgs), dim3(256), 0, 0, buf, per_wg, sink, repeat); break;
    case 4: hipLaunchKernelGGL(reader<4>, dim3(wgs), dim3(256), 0, 0, buf, per_wg, sink, repeat); break;
    default:
      hipLaunchKernelGGL(reader<1>, dim3(wgs), dim3(256), 0, 0, buf, per_wg, sink, repeat);
      hipLaunchKernelGGL(reader<1>, dim3(wgs), dim3(256), 0, 0, buf, per_wg, sink, repeat);
      break;
  }
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)per_wg * wgs * repeat * (mode == 5 ? 2 : 1);
  printf("mode %d: %.0f bytes requested (%.1f MiB%s) in %.3f ms = %.2f TB/s\n", mode, bytes, bytes / 1048576.0,
         mode == 4 ? ", 32 MiB re-read 64 times" : mode == 5 ? ", 128 MiB read by two consecutive launches" : "", ms, bytes / ms / 1e9);
  return 0;
}
