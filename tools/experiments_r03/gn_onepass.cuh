// SHELVED EXPERIMENT (round 3, not compiled into the library): GroupNorm in one pass over the activation and one launch, with an
// in-kernel barrier per sample built from device-scope (sc1) stores / loads and atomic tickets.  Bit-exact against the two-pass
// kernels' statistics to 1e-5, deterministic, counters self-resetting -- and SLOWER than the two-pass forms on every U-Net shape
// (profiles/r03_gn_onepass_experiment.txt): a device-scope round trip costs 1.5-2.5 us on MI355X and the barrier needs four in
// series (store ack, ticket, poll, partial-sum loads), 7-10 us per launch, more than the second read of x it saves; and the
// apply pass is VALU-bound (SiLU), not HBM-bound.  To revive: paste both parts into csrc/aql_norm.hip (kernel inside the
// anonymous namespace, host part before aql_layernorm_fwd) and declare the two entry points in include/aqualora_hip.h.
// ---------------------------------------------------------------------------------------------------
// One-pass GroupNorm (round 3): the activation is read ONCE, coalesced, and written once -- in one launch.
// Workgroup (chunk, sample) owns rp * NR whole pixel rows (thread = one 16-byte chunk column x NR rows, held in registers),
// sums its rows per group (fixed order, LDS), publishes the 64 partial sums with device-scope write-through stores, takes a
// ticket on the sample's arrival counter and WAITS until every chunk of the sample has arrived (an in-kernel barrier per
// sample); then it adds the chunks' sums in chunk order (deterministic, every workgroup the same bits), normalises its rows
// from registers and stores them.  The per-(sample, group) kernels read 20-80 byte pixel segments twice (13-23 us for the
// 16-21 MB of a 32x32x640 map); the split forms at 64x64 need two launches and read x twice.
// The barrier is safe because the host only uses this kernel when the whole grid is co-resident (occupancy x CUs >= grid:
// every workgroup is dispatched without waiting for another to retire) and nothing else of this stream runs beside it; a
// concurrent kernel of another stream (RCCL) can only delay the last arrivals, not block them.  No fences (a device-scope
// release fence writes back the whole L2): partial sums are the only data exchanged, stored sc1 / loaded sc1.  The spin is
// bounded (~seconds) and raises `err` instead of hanging the GPU.  Counters: [B][2] ints, zero before and after the launch
// (the last workgroup to leave the barrier resets them).
// MODE 0: y = act(xhat * gamma + beta), writes stats (mean, rstd).  MODE 1: dx = rstd (dxhat - mean dxhat - xhat mean(dxhat xhat)) (+ dres)
template <int MODE, int NR>
__global__ __launch_bounds__(512) void gn_onepass_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                         const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
                                                         float* __restrict__ stats, int HW, int C, int rp, float eps, int silu,
                                                         const bf16_t* __restrict__ dres, bf16_t* __restrict__ out,
                                                         float* __restrict__ part, int* __restrict__ counters,
                                                         int* __restrict__ err) {
  __shared__ float sp[512 * 4];
  __shared__ float tot[64];
  const int cols = C >> 3, cpg = C / G;
  const int tid = threadIdx.x;
  const int col = tid % cols, rr = tid / cols;
  const bool active = rr < rp;
  const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
  const int c0 = col * 8;
  const int g0 = c0 / cpg, g1 = (c0 + 7) / cpg;
  const int split = (g0 + 1) * cpg - c0;   // channels j < split belong to g0, the rest to g1
  const int r0 = chunk * rp * NR;
  const float inv_count = 1.f / ((float)HW * cpg);
  float ga[8], be[8];
  unpack8(*reinterpret_cast<const uint4*>(gamma + c0), ga);
  unpack8(*reinterpret_cast<const uint4*>(beta + c0), be);
  float mu2[2] = {0.f, 0.f}, rs2[2] = {1.f, 1.f};
  if (MODE == 1) {
    mu2[0] = stats[(b * G + g0) * 2 + 0], rs2[0] = stats[(b * G + g0) * 2 + 1];
    mu2[1] = stats[(b * G + g1) * 2 + 0], rs2[1] = stats[(b * G + g1) * 2 + 1];
  }
  uint4 xw[NR], dw[NR];
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int r = min(r0 + rr + u * rp, HW - 1);   // rows past the end re-read the last row (masked below)
    const long off = ((long)b * HW + r) * C + c0;
    xw[u] = *reinterpret_cast<const uint4*>(x + off);
    if (MODE == 1) dw[u] = *reinterpret_cast<const uint4*>(dy + off);
  }
  float s[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    if (!active || r0 + rr + u * rp >= HW) continue;
    float xv[8];
    unpack8(xw[u], xv);
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = j < split ? 0 : 1;
        s[k][0] += xv[j];
        s[k][1] += xv[j] * xv[j];
      }
    } else {
      float dv[8];
      unpack8(dw[u], dv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = j < split ? 0 : 1;
        const float xh = (xv[j] - mu2[k]) * rs2[k];
        float d = dv[j];
        if (silu) {
          const float z = xh * ga[j] + be[j];
          const float sg = sigmoidf_(z);
          d *= sg * (1.f + z * (1.f - sg));
        }
        d *= ga[j];
        s[k][0] += d;
        s[k][1] += d * xh;
      }
    }
  }
  if (active) {
    sp[tid * 4 + 0] = s[0][0], sp[tid * 4 + 1] = s[0][1], sp[tid * 4 + 2] = s[1][0], sp[tid * 4 + 3] = s[1][1];
  }
  __syncthreads();
  if (tid < 64) {   // (group, which) = (tid >> 1, tid & 1): fixed order over the group's columns and the rp row slots
    const int g = tid >> 1, which = tid & 1;
    const int cf = (g * cpg) >> 3, cl = ((g + 1) * cpg - 1) >> 3;
    float acc = 0.f;
    for (int c = cf; c <= cl; ++c) {
      const int kk = ((c * 8) / cpg == g) ? 0 : 1;
      for (int q = 0; q < rp; ++q) acc += sp[(q * cols + c) * 4 + kk * 2 + which];
    }
    __hip_atomic_store(part + ((long)b * nchunk + chunk) * 64 + tid, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this workgroup's sums are in memory before its ticket is taken
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_fetch_add(counters + b * 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while (__hip_atomic_load(counters + b * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nchunk) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1L << 21)) {   // ~seconds: never reached unless the co-residency contract was broken
        *err = 1;
        break;
      }
    }
  }
  __syncthreads();
  {   // every load is a round trip to memory: all of them in flight at once (thread = (chunk lane q, value), chunks q, q+NQ, ...),
      // then the chunks are added in chunk order -- the same bits in every workgroup
    const int NQ = blockDim.x >> 6;   // chunk lanes (<= 8)
    const int q = tid >> 6, val = tid & 63;
    float v[32];
    const float* src = part + (long)b * nchunk * 64 + val;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int k = q + i * NQ;
      v[i] = (i * NQ < nchunk) ? __hip_atomic_load(src + (long)min(k, nchunk - 1) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    }
    // park: sp[k][val] for k < nchunk (sp has 2048 floats = 32 chunks at a time)
    float acc = 0.f;
    for (int base = 0; base < nchunk; base += 32) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const int k = q + i * NQ;
        if (k >= base && k < base + 32 && k < nchunk) sp[(k - base) * 64 + val] = v[i];
      }
      __syncthreads();
      if (tid < 64) {
        const int n = min(32, nchunk - base);
        for (int k = 0; k < n; ++k) acc += sp[k * 64 + tid];
      }
    }
    if (tid < 64) tot[tid] = acc;
  }
  __syncthreads();
  float m12[2] = {0.f, 0.f}, m22[2] = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int g = k ? g1 : g0;
    const float a0 = tot[2 * g], a1 = tot[2 * g + 1];
    if (MODE == 0) {
      mu2[k] = a0 * inv_count;
      rs2[k] = rsqrtf(fmaxf(a1 * inv_count - mu2[k] * mu2[k], 0.f) + eps);
    } else {
      m12[k] = a0 * inv_count;
      m22[k] = a1 * inv_count;
    }
  }
  if (MODE == 0 && chunk == 0 && tid < G) {
    const float mean = tot[2 * tid] * inv_count;
    stats[(b * G + tid) * 2 + 0] = mean;
    stats[(b * G + tid) * 2 + 1] = rsqrtf(fmaxf(tot[2 * tid + 1] * inv_count - mean * mean, 0.f) + eps);
  }
  if (tid == 0) {   // leave the barrier (its returning atomic is a round trip to memory: issued here, consumed at the very end)
    const int d = __hip_atomic_fetch_add(counters + b * 2 + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (d == nchunk - 1) {   // every workgroup of the sample is past the wait: ready for the next launch
      __hip_atomic_store(counters + b * 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(counters + b * 2 + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (!active) return;
  uint4 rw[NR];
  if (MODE == 1 && dres != nullptr) {
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const int r = min(r0 + rr + u * rp, HW - 1);
      rw[u] = *reinterpret_cast<const uint4*>(dres + ((long)b * HW + r) * C + c0);
    }
  }
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int r = r0 + rr + u * rp;
    if (r >= HW) continue;
    float xv[8], dv[8], o[8];
    unpack8(xw[u], xv);
    if (MODE == 1) unpack8(dw[u], dv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = j < split ? 0 : 1;
      const float mu = mu2[k], rs = rs2[k];
      const float xh = (xv[j] - mu) * rs;
      const float z = xh * ga[j] + be[j];
      if (MODE == 0) {
        o[j] = silu ? z * sigmoidf_(z) : z;
      } else {
        float d = dv[j];
        if (silu) {
          const float sg = sigmoidf_(z);
          d *= sg * (1.f + z * (1.f - sg));
        }
        d *= ga[j];
        o[j] = rs * (d - m12[k] - xh * m22[k]);
      }
    }
    if (MODE == 1 && dres != nullptr) {
      float rsd[8];
      unpack8(rw[u], rsd);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += rsd[j];
    }
    *reinterpret_cast<uint4*>(out + ((long)b * HW + r) * C + c0) = pack8(o);
  }
}


// ---- one-pass form (gn_onepass_kernel): geometry, co-residency check, launch ---------------------------------------
namespace {
struct OnePassPlan {
  int threads, rp, nr, nchunk;
};
template <int MODE, int NR>
int onepass_capacity(int threads) {   // workgroups of this instance that are resident at once on the whole device
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  }
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gn_onepass_kernel<MODE, NR>, threads, 0) != hipSuccess) return 0;
  return per_cu * cus;
}
template <int MODE>
int onepass_capacity_nr(int nr, int threads) {
  switch (nr) {
    case 2: return onepass_capacity<MODE, 2>(threads);
    case 4: return onepass_capacity<MODE, 4>(threads);
    case 8: return onepass_capacity<MODE, 8>(threads);
    default: return onepass_capacity<MODE, MODE == 0 ? 16 : 8>(threads);   // 16 rows per thread: forward only (x AND dy would spill)
  }
}
// rows per workgroup = rp * NR with NR in {16, 8, 4, 2}: the fattest chunks that still give >= ~1 workgroup per CU; the
// grid must be co-resident (the sample barrier) and a sample's chunks must fit the partial-sum buffer (256 chunks)
template <int MODE>
bool onepass_plan(int B, int HW, int C, OnePassPlan* p) {
  static const int en = getenv("AQL_GN_ONEPASS") ? atoi(getenv("AQL_GN_ONEPASS")) : 1;   // A/B hook: 0 = the older forms
  const int cols = C / 8, cpg = C / G;
  if (!en || cpg < 8 || cols > 512 || C % 8 != 0) return false;
  const int rp = cols >= 256 ? 1 : 512 / cols > 12 ? 12 : 512 / cols;
  const int threads = ((cols * rp + 63) / 64) * 64;
  if (threads > 512) return false;
  const int nrs[4] = {MODE == 0 ? 16 : 8, MODE == 0 ? 8 : 4, MODE == 0 ? 4 : 2, 2};
  for (int i = 0; i < 4; ++i) {
    const int nr = nrs[i];
    const int nchunk = (HW + rp * nr - 1) / (rp * nr);
    const long grid = (long)nchunk * B;
    if (nchunk > 256) continue;
    if (grid < 224 && i < 3 && nr > 2) continue;   // too few workgroups: try thinner chunks first
    if (grid > onepass_capacity_nr<MODE>(nr, threads)) return false;   // thinner chunks only make the grid larger
    p->threads = threads, p->rp = rp, p->nr = nr, p->nchunk = nchunk;
    return true;
  }
  return false;
}
template <int MODE>
void onepass_launch(const OnePassPlan& p, int B, const bf16_t* x, const bf16_t* dy, const bf16_t* gamma, const bf16_t* beta,
                    float* stats, int HW, int C, float eps, int silu, const bf16_t* dres, bf16_t* out, float* part,
                    int* counters, int* err, hipStream_t stream) {
  const dim3 grid(p.nchunk, B), block(p.threads);
  switch (p.nr) {
    case 2: hipLaunchKernelGGL((gn_onepass_kernel<MODE, 2>), grid, block, 0, stream, x, dy, gamma, beta, stats, HW, C, p.rp, eps, silu, dres, out, part, counters, err); break;
    case 4: hipLaunchKernelGGL((gn_onepass_kernel<MODE, 4>), grid, block, 0, stream, x, dy, gamma, beta, stats, HW, C, p.rp, eps, silu, dres, out, part, counters, err); break;
    case 8: hipLaunchKernelGGL((gn_onepass_kernel<MODE, 8>), grid, block, 0, stream, x, dy, gamma, beta, stats, HW, C, p.rp, eps, silu, dres, out, part, counters, err); break;
    default: hipLaunchKernelGGL((gn_onepass_kernel<MODE, MODE == 0 ? 16 : 8>), grid, block, 0, stream, x, dy, gamma, beta, stats, HW, C, p.rp, eps, silu, dres, out, part, counters, err); break;
  }
}
}  // namespace

// GroupNorm(32) (+ SiLU) forward / backward-data in ONE pass over the activation (gn_onepass_kernel).  sync: caller-owned
// persistent scratch of aql_groupnorm_onepass_sync_bytes(B) bytes, ZERO before the first call and never written by anyone
// else ([B][256][64] fp32 partial sums, then [B][2] int counters and one int error flag, which the kernel raises if its
// sample barrier timed out).  Returns 100 when the shape is not served (grid not co-resident, < 8 channels per group):
// the caller then uses aql_groupnorm_silu_fwd / _bwd.
extern "C" long aql_groupnorm_onepass_sync_bytes(int B) { return ((long)B * 256 * 64 + (long)B * 2 + 1) * 4; }

extern "C" int aql_groupnorm_silu_fwd_onepass(const bf16_t* x, int B, int HW, int C, const bf16_t* gamma, const bf16_t* beta,
                                              float eps, int silu, bf16_t* y, float* stats, void* sync, hipStream_t stream) {
  AQL_CHECK_ARG(x && gamma && beta && y && stats && sync, "aql_groupnorm_silu_fwd_onepass: null operand");
  AQL_CHECK_ARG(C % 8 == 0 && C % G == 0 && B > 0 && HW > 0, "aql_groupnorm_silu_fwd_onepass: bad shape C=%d", C);
  OnePassPlan p;
  if (!onepass_plan<0>(B, HW, C, &p)) return 100;
  float* part = static_cast<float*>(sync);
  int* counters = reinterpret_cast<int*>(part + (long)B * 256 * 64);
  onepass_launch<0>(p, B, x, nullptr, gamma, beta, stats, HW, C, eps, silu, nullptr, y, part, counters, counters + 2 * B, stream);
  AQL_CHECK_LAUNCH("aql_groupnorm_silu_fwd_onepass");
  return AQL_OK;
}

extern "C" int aql_groupnorm_silu_bwd_onepass(const bf16_t* x, const bf16_t* dy, int B, int HW, int C, const bf16_t* gamma,
                                              const bf16_t* beta, int silu, const float* stats, const bf16_t* dres, bf16_t* dx,
                                              void* sync, hipStream_t stream) {
  AQL_CHECK_ARG(x && dy && gamma && beta && dx && stats && sync, "aql_groupnorm_silu_bwd_onepass: null operand");
  AQL_CHECK_ARG(C % 8 == 0 && C % G == 0 && B > 0 && HW > 0, "aql_groupnorm_silu_bwd_onepass: bad shape C=%d", C);
  OnePassPlan p;
  if (!onepass_plan<1>(B, HW, C, &p)) return 100;
  float* part = static_cast<float*>(sync);
  int* counters = reinterpret_cast<int*>(part + (long)B * 256 * 64);
  onepass_launch<1>(p, B, x, dy, gamma, beta, const_cast<float*>(stats), HW, C, 0.f, silu, dres, dx, part, counters,
                    counters + 2 * B, stream);
  AQL_CHECK_LAUNCH("aql_groupnorm_silu_bwd_onepass");
  return AQL_OK;
}

