// Measurement helpers for profiles/ (not part of the product library).
//
//   microbench issue    -- VALU issue / latency figures on this chip: cycles per wave64 VALU
//                          instruction per SIMD with 1, 2, 4, 8 waves resident per SIMD
//                          (independent v_add_f32 streams), the dependent-chain latency of the
//                          instruction kinds the CTC consumer is made of (v_add_f32, v_max_u32
//                          with a DPP operand, v_exp_f32, ds_bpermute_b32, v_readlane + use).
//   microbench traffic  -- streaming kernels of KNOWN byte counts (1 GiB each) in the access
//                          widths the product kernels use, to calibrate rocprofv3's FETCH_SIZE /
//                          WRITE_SIZE on this image: read4 / read16 (4 / 16 B per lane coalesced
//                          loads), write4 / write8 / write16.
// Build: profiles/microbench/build.sh.  Output: one JSON object on stdout.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIters = 2048;

// 16 independent accumulators, 16 v_add_f32 per iteration
__global__ void __launch_bounds__(256) k_issue_indep(float *out, unsigned long long *cyc, unsigned long long *real) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.5f + i;
  const float inc = 1.0f + blockIdx.x * 1e-9f;
  __builtin_amdgcn_s_barrier();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(inc));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) real[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = r1 - r0;
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// one dependent chain; KIND selects the instruction
template <int KIND>
__global__ void __launch_bounds__(256) k_chain(float *out, unsigned long long *cyc) {
  float a = threadIdx.x * 0.5f + 1.0f;
  unsigned u = threadIdx.x * 2654435761u + 12345u;
  const float inc = 1.0f + blockIdx.x * 1e-9f;
  const int lane4 = ((threadIdx.x & 63) ^ 4) << 2;
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if constexpr (KIND == 0) {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(inc));
      } else if constexpr (KIND == 1) {  // max with the quad neighbour: one DPP-folded instruction
        asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u));
      } else if constexpr (KIND == 2) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(a));
      } else if constexpr (KIND == 3) {  // LDS crossbar shuffle + wait
        u = (unsigned)__builtin_amdgcn_ds_bpermute(lane4, (int)u) + 1u;
      } else if constexpr (KIND == 4) {  // VALU -> SALU -> VALU round trip
        const int s = __builtin_amdgcn_readlane((int)u, 7);
        u = u + (unsigned)s;
      } else if constexpr (KIND == 5) {  // compare-exchange as the sort network writes it: 3 VALU
        unsigned o;
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(u));
        u = (threadIdx.x & 1) ? min(u, o) + 1u : max(u, o) + 1u;
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + (float)u;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

static double median(std::vector<unsigned long long> v) {
  std::sort(v.begin(), v.end());
  return (double)v[v.size() / 2];
}

template <typename K>
static double run_cycles(K kernel, int blocks, int threads, float *out, unsigned long long *cyc) {
  const int waves = blocks * threads / 64;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out, cyc);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(waves);
  CHECK(hipMemcpy(h.data(), cyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return median(h);
}

static void issue() {
  float *out;
  unsigned long long *cyc;
  CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
  CHECK(hipMalloc(&cyc, 256 * 8 * 4 * sizeof(unsigned long long)));
  printf("{\"what\": \"VALU issue and dependent-chain latency, gfx950, s_memtime cycles\", \"insts_per_wave\": %d,\n", kIters * 16);
  // W waves per SIMD: 256-thread blocks (4 waves, one per SIMD), W blocks per CU, 256 CUs
  printf(" \"independent_v_add_f32\": {");
  unsigned long long *real;
  CHECK(hipMalloc(&real, 256 * 8 * 4 * sizeof(unsigned long long)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int W : {1, 2, 4, 8}) {
    const int waves = 256 * W * 4;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_issue_indep, dim3(256 * W), dim3(256), 0, 0, out, cyc, real);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_issue_indep, dim3(256 * W), dim3(256), 0, 0, out, cyc, real);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(waves), hr(waves);
    CHECK(hipMemcpy(h.data(), cyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hr.data(), real, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const double c = median(h), r = median(hr);
    // every wave issues kIters * 16 instructions in c ticks of s_memtime; W waves share the SIMD.
    // s_memrealtime ticks at 100 MHz: ticks_per_us relates s_memtime to wall time; the launch's
    // event time gives the whole grid's rate independently of either counter.
    const double insts_per_simd = (double)kIters * 16 * W;
    printf("%s\"waves_per_simd_%d\": {\"memtime_ticks_per_wave\": %.0f, \"memtime_ticks_per_us\": %.1f, "
           "\"launch_us\": %.1f, \"ns_per_wave_inst_per_simd_by_event\": %.3f, \"ns_per_wave_inst_per_simd_by_memrealtime\": %.3f}",
           W == 1 ? "" : ", ", W, c, c / (r / 100.0), ms * 1e3, ms * 1e6 / insts_per_simd, r * 10.0 / insts_per_simd);
  }
  printf("},\n \"dependent_chain_cycles_per_inst\": {");
  const char *names[6] = {"v_add_f32", "v_max_u32_dpp", "v_exp_f32", "ds_bpermute_b32_plus_add", "v_readlane_plus_add", "sort_cmpx_stage_3valu"};
  for (int W : {1, 8}) {
    double c[6];
    c[0] = run_cycles(k_chain<0>, 256 * W, 256, out, cyc);
    c[1] = run_cycles(k_chain<1>, 256 * W, 256, out, cyc);
    c[2] = run_cycles(k_chain<2>, 256 * W, 256, out, cyc);
    c[3] = run_cycles(k_chain<3>, 256 * W, 256, out, cyc);
    c[4] = run_cycles(k_chain<4>, 256 * W, 256, out, cyc);
    c[5] = run_cycles(k_chain<5>, 256 * W, 256, out, cyc);
    printf("%s\"waves_per_simd_%d\": {", W == 1 ? "" : ", ", W);
    for (int i = 0; i < 6; ++i) printf("%s\"%s\": %.2f", i ? ", " : "", names[i], c[i] / (double)(kIters * 16));
    printf("}");
  }
  printf("}}\n");
}

// ---- streaming kernels of known size -----------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_read(const T *src, size_t n, float *sink) {
  float acc = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const T v = src[i];
    acc += reinterpret_cast<const float *>(&v)[0];
  }
  if (acc == 12345.678f) sink[0] = acc;  // never true: keeps the loads alive
}
template <typename T>
__global__ void __launch_bounds__(256) k_write(T *dst, size_t n) {
  T v;
  memset(&v, 0, sizeof(T));
  reinterpret_cast<int *>(&v)[0] = threadIdx.x;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
}

static void traffic() {
  const size_t bytes = 1ull << 30;
  void *buf;
  float *sink;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(buf, 1, bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("{\"what\": \"streaming kernels of known size\", \"bytes_each\": %zu", bytes);
  auto timed = [&](const char *name, auto launch) {
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf(", \"%s_GBs\": %.0f", name, bytes * 5 / (ms * 1e-3) / 1e9);
  };
  const int grid = 256 * 16;
  timed("read4", [&] { hipLaunchKernelGGL(k_read<float>, dim3(grid), dim3(256), 0, 0, (const float *)buf, bytes / 4, sink); });
  timed("read16", [&] { hipLaunchKernelGGL(k_read<float4>, dim3(grid), dim3(256), 0, 0, (const float4 *)buf, bytes / 16, sink); });
  timed("write4", [&] { hipLaunchKernelGGL(k_write<float>, dim3(grid), dim3(256), 0, 0, (float *)buf, bytes / 4); });
  timed("write8", [&] { hipLaunchKernelGGL(k_write<float2>, dim3(grid), dim3(256), 0, 0, (float2 *)buf, bytes / 8); });
  timed("write16", [&] { hipLaunchKernelGGL(k_write<float4>, dim3(grid), dim3(256), 0, 0, (float4 *)buf, bytes / 16); });
  printf("}\n");
}

// ---- store patterns: what a 16-byte-per-lane store stream reaches, by how a wave's successive
// stores are laid out (1 GiB written per launch) ----------------------------------------------
// mode 0: grid-stride (every wave's next store is gridDim * 4 KiB further on)
// mode 1: every WAVE owns a contiguous region and walks it 1 KiB at a time
// mode 2: every WORKGROUP owns a contiguous region, its four waves interleave 1 KiB pieces
// mode 3: rows of 992 B (62 lanes x 16 B), a workgroup = one column of rows 4096 rows apart,
//         its waves take rows in turn (the row-at-a-time oc_expand pattern)
// mode 4: rows of 992 B, a wave walks consecutive rows (contiguous)
__global__ void __launch_bounds__(256) k_store(float4 *dst, size_t n16, int mode) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 v = {1.0f * threadIdx.x, 2.0f, 3.0f, 4.0f};
  const size_t nblk = gridDim.x;
  if (mode == 0) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += nblk * 256) dst[i] = v;
  } else if (mode == 1) {
    const size_t per_wave = n16 / (nblk * 4), w = (size_t)blockIdx.x * 4 + wave;
    for (size_t i = lane; i < per_wave; i += 64) dst[w * per_wave + i] = v;
  } else if (mode == 2) {
    const size_t per_blk = n16 / nblk;
    for (size_t i = threadIdx.x; i < per_blk; i += 256) dst[blockIdx.x * per_blk + i] = v;
  } else if (mode == 3) {
    // 62 x 16 B rows; matrix of (H rows) x (nblk columns): element (h, c) at (h * nblk + c) * 62
    const size_t H = n16 / (nblk * 62);
    if (lane < 62)
      for (size_t h = wave; h < H; h += 4) dst[(h * nblk + blockIdx.x) * 62 + lane] = v;
  } else if (mode == 4) {
    const size_t rows = n16 / 62, per_wave = rows / (nblk * 4), w = (size_t)blockIdx.x * 4 + wave;
    if (lane < 62)
      for (size_t r = 0; r < per_wave; ++r) dst[(w * per_wave + r) * 62 + lane] = v;
  } else {
    // the tiled oc_expand traversal: output (H = 513) x (N = 4096 rows of 62 x 16 B); a tile = 8
    // consecutive rows of one h (7936 B contiguous, written 1 KiB per instruction).
    // mode 5: workgroup = (n tile, chunk of 32 h), its waves take h, h + 4, ... (tiles of one
    //         workgroup lie 4096 rows apart);  mode 6: workgroup = 32 consecutive n tiles of one h
    //         (one contiguous 254 KB region), waves take tiles in turn;  mode 7: as 6, but the
    //         workgroup's region is walked 1 KiB per wave in turn (all four waves inside one 4 KiB).
    const size_t H = 513, NT = 512, tile16 = 8 * 62;
    if (mode == 5) {
      const size_t tile = blockIdx.x % NT, hc = blockIdx.x / NT;
      for (size_t h = hc * 32 + wave; h < H && h < (hc + 1) * 32; h += 4) {
        float4 *t = dst + (h * NT + tile) * tile16;
        for (size_t i = lane; i < tile16; i += 64) t[i] = v;
      }
    } else if (mode == 6) {
      const size_t h = blockIdx.x / 16, t0 = (blockIdx.x % 16) * 32;
      if (h < H)
        for (size_t tt = wave; tt < 32; tt += 4) {
          float4 *t = dst + (h * NT + t0 + tt) * tile16;
          for (size_t i = lane; i < tile16; i += 64) t[i] = v;
        }
    } else {
      const size_t h = blockIdx.x / 16, t0 = (blockIdx.x % 16) * 32;
      if (h < H) {
        float4 *t = dst + (h * NT + t0) * tile16;
        for (size_t i = threadIdx.x; i < 32 * tile16; i += 256) t[i] = v;
      }
    }
  }
}

// the tiled traversal of mode 5 with the product kernel's other ingredients added one by one:
// bit 0: the data of every store comes out of LDS (written, fenced, read back);
// bit 1: blockIdx -> item through the XCD remap; bit 2: a preamble of 32 KiB of global loads
// into LDS per workgroup; `lds` bytes of dynamic LDS set the occupancy.
__device__ __forceinline__ unsigned xcd_remap_mb(unsigned b, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u, xcd = b & 7u, idx = b >> 3;
  const unsigned base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
  return base + idx;
}
__global__ void __launch_bounds__(256) k_tiles(float4 *dst, const float4 *src, int flags) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t H = 513, NT = 512, tile16 = 8 * 62;
  const unsigned item = (flags & 2) ? xcd_remap_mb(blockIdx.x, gridDim.x) : blockIdx.x;
  const size_t tile = item % NT, hc = item / NT;
  float4 *pre = reinterpret_cast<float4 *>(smem);
  float4 *stage = pre + 2048 + wave * 512;
  if (flags & 4) {
    for (int k0 = threadIdx.x; k0 < 2048; k0 += 256 * 8) {
      float4 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = src[tile * 2048 + k0 + q * 256];
#pragma unroll
      for (int q = 0; q < 8; ++q) pre[k0 + q * 256] = v[q];
    }
    __syncthreads();
  }
  float4 v = {1.0f * threadIdx.x, 2.0f, 3.0f, 4.0f};
  for (size_t h = hc * 32 + wave; h < H && h < (hc + 1) * 32; h += 4) {
    float4 *t = dst + (h * NT + tile) * tile16;
    if (flags & 1) {
      for (size_t i = lane; i < tile16; i += 64) stage[i] = v;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (size_t i = lane; i < tile16; i += 64) t[i] = stage[(i + 64) % tile16];
      __builtin_amdgcn_wave_barrier();
    } else {
      for (size_t i = lane; i < tile16; i += 64) t[i] = v;
    }
  }
}

static void tiles() {
  const size_t out = (size_t)513 * 4096 * 992;
  float4 *buf, *src;
  CHECK(hipMalloc(&buf, out));
  CHECK(hipMalloc(&src, (size_t)512 * 2048 * 16));
  CHECK(hipMemset(src, 0, (size_t)512 * 2048 * 16));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tiles), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  printf("{\"what\": \"tiled oc_expand traversal as a store stream, GB/s; flags: 1 data through LDS, 2 XCD remap, 4 32 KiB preamble\"");
  for (int lds : {65536, 40960, 20480}) {
    for (int flags : {0, 1, 2, 3, 4, 7}) {
      hipLaunchKernelGGL(k_tiles, dim3(512 * 17), dim3(256), lds, 0, buf, src, flags);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_tiles, dim3(512 * 17), dim3(256), lds, 0, buf, src, flags);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf(",\n \"lds_%d_flags_%d\": %.0f", lds, flags, out * 5 / (ms * 1e-3) / 1e9);
    }
  }
  printf("}\n");
}

static void stores() {
  const size_t bytes = 1ull << 30;
  float4 *buf;
  CHECK(hipMalloc(&buf, (size_t)513 * 4096 * 992 + bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("{\"what\": \"16-byte store streams, 1 GiB per launch, GB/s by layout and grid\"");
  const char *names[8] = {"grid_stride", "wave_contiguous", "workgroup_contiguous", "rows992_strided", "rows992_contiguous",
                          "oc_tiles_h_strided", "oc_tiles_contiguous_by_wave", "oc_tiles_contiguous_by_kib"};
  for (int mode = 0; mode < 8; ++mode) {
    printf(",\n \"%s\": {", names[mode]);
    bool first = true;
    for (int grid : {512, 1024, 2048, 4096, 8192}) {
      if (mode >= 5) grid = mode == 5 ? 512 * 17 : 513 * 16;
      hipLaunchKernelGGL(k_store, dim3(grid), dim3(256), 0, 0, buf, bytes / 16, mode);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_store, dim3(grid), dim3(256), 0, 0, buf, bytes / 16, mode);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const size_t wrote = mode >= 5 ? (size_t)513 * 4096 * 992 : bytes;
      printf("%s\"grid_%d\": %.0f", first ? "" : ", ", grid, wrote * 5 / (ms * 1e-3) / 1e9);
      first = false;
      if (mode >= 5) break;
    }
    printf("}");
  }
  printf("}\n");
}

int main(int argc, char **argv) {
  if (argc > 1 && !strcmp(argv[1], "stores")) { stores(); return 0; }
  if (argc > 1 && !strcmp(argv[1], "tiles")) { tiles(); return 0; }
  if (argc > 1 && !strcmp(argv[1], "issue")) issue();
  else if (argc > 1 && !strcmp(argv[1], "traffic")) traffic();
  else { fprintf(stderr, "usage: microbench issue|traffic\n"); return 2; }
  return 0;
}
