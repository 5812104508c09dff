// Micro-benchmarks that establish how the gfx950 matrix pipe, LDS reads and LDS-DMA overlap inside one wave and
// between the two waves of a SIMD -- the facts the GEMM schedules in qutlass_amd/csrc/gemm_mx.hip.h are built on.
//   mode 0  MFMA only                (8 x v_mfma_scale_f32_32x32x64 fp4 per iteration, fixed operands)
//   mode 1  LDS reads only           (6 x ds_read_b128 per iteration)
//   mode 2  MFMA(cur) ; reads -> the OTHER fragment set (double buffered)          = "queue" schedule
//   mode 3  reads ; wait ; MFMA on the same single set                              = serialised
//   mode 4  mode 2 + 2 LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB each) per iteration
//   mode 5  mode 2 with the reads issued BEFORE the MFMAs of the iteration
//   mode 6  LDS-DMA only (2 per iteration)
//   mode 7  mode 4 + s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier every 4 slices (the stage hand-off)
// Reported: shader cycles per iteration (s_memtime, wave 0 of workgroup 0) with every CU busy.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "smi_sampler.h"

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP ERROR %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2);} } while (0)

template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void ub_kernel(const char* g, uint32_t gbytes, float* out, uint32_t* cyc, int iters, int random_fill, const uint32_t* fill = nullptr) {
  __shared__ __attribute__((aligned(16))) char smem[64 * 1024 + 16 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 64 * 1024 / 4; i += THREADS) {
    uint32_t x = (uint32_t)i * 2654435761u + blockIdx.x * 40503u + 12345u;   // operand bits: constant (low toggle power) or pseudo-random
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    ((uint32_t*)smem)[i] = fill ? fill[i] : (random_fill ? x : 0x22222222u);   // fill: 64 KiB operand image prepared by the host (e.g. quantised Gaussian codes)
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g), 0, gbytes, 0x00020000);
  v16f acc[8];
  for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  typedef float v4f_ __attribute__((ext_vector_type(4)));
  v4f_ acc4[32];   // MODE 26 only (16x16x128 MFMAs)
  for (int a = 0; a < 32; ++a) acc4[a] = v4f_{0.f, 0.f, 0.f, 0.f};
  v4i fa[2][4], fb[2][2];
  // conflict-free fragment addressing of the GEMM: row = lane&31, chunk c = 4*(lane>>5) + j, phys = c ^ ((row>>1)&7)
  const int sw = ((lane & 31) >> 1) & 7;
  auto addr = [&](int j) __attribute__((always_inline)) { return (lane & 31) * 128 + (((4 * (lane >> 5) + j) ^ sw) << 4); };
  const int base = addr(0);
  for (int s = 0; s < 2; ++s) {
    for (int t = 0; t < 4; ++t) fa[s][t] = *(const v4i*)(smem + base + t * 4096);
    for (int t = 0; t < 2; ++t) fb[s][t] = *(const v4i*)(smem + 32768 + base + t * 4096);
  }
  const int scale = 0x7f7f7f7f;
  auto mfma8 = [&](int set) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const v4i a = fa[set][m], b = fb[set][n];
        acc[m * 2 + n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v8i{b[0], b[1], b[2], b[3], 0, 0, 0, 0}, v8i{a[0], a[1], a[2], a[3], 0, 0, 0, 0},
                                                                      acc[m * 2 + n], 4, 4, 0, scale, 0, scale);
      }
  };
  auto reads = [&](int set, int off) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 4; ++t) fa[set][t] = *(const v4i*)(smem + addr(off) + t * 4096);
#pragma unroll
    for (int t = 0; t < 2; ++t) fb[set][t] = *(const v4i*)(smem + 32768 + addr(off) + t * 4096);
  };
  auto keep = [&](int set) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 4; ++t) asm volatile("" ::"v"(fa[set][t]));
#pragma unroll
    for (int t = 0; t < 2; ++t) asm volatile("" ::"v"(fb[set][t]));
  };
  auto dma2 = [&](int it) __attribute__((always_inline)) {
    const int v = lane * 16 + (it & 63) * 2048 + wave * 131072;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem + 65536 + wave * 2048), 16, v, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem + 65536 + wave * 2048 + 1024), 16, v + 1024, 0, 0, 0);
  };
  auto dma1 = [&](int it) __attribute__((always_inline)) {
    const int v = lane * 16 + (it & 63) * 2048 + wave * 131072 + 65536;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem + 65536 + wave * 2048), 16, v, 0, 0, 0);
  };
  auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };

  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it += 2) {
    const int off0 = (it >> 1) & 3, off1 = (off0 + 1) & 3;   // k-slice index j (runtime, like the stage loop)
    if (MODE == 0) { mfma8(0); fence(); mfma8(1); fence(); }
    if (MODE >= 20 && MODE <= 23) {     // accumulator-stationary orders: CH back-to-back MFMAs into the SAME accumulator (8 / 4 / 2), 23 = operand-stationary A
      constexpr int CH = MODE == 20 ? 8 : MODE == 21 ? 4 : 2;
#pragma unroll
      for (int set = 0; set < 2; ++set)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int a_i = (MODE == 23) ? (i >> 1) : (i & 3), b_i = (MODE == 23) ? (i & 1) : ((i >> 2) & 1);
          const int acc_i = (MODE == 23) ? i : (i / CH) * CH % 8;
          const v4i a = fa[set][a_i], b = fb[set][b_i];
          acc[acc_i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v8i{b[0], b[1], b[2], b[3], 0, 0, 0, 0}, v8i{a[0], a[1], a[2], a[3], 0, 0, 0, 0},
                                                                        acc[acc_i], 4, 4, 0, scale, 0, scale);
        }
      fence();
    }
    if (MODE == 24 || MODE == 25) {   // 24: every MFMA the SAME two operands (nothing toggles between MFMAs); 25: both operands change on every MFMA
      auto frag = [&](const int k) __attribute__((always_inline)) { return k < 8 ? fa[k >> 2][k & 3] : fb[(k - 8) >> 1][(k - 8) & 1]; };
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const v4i a = (MODE == 24) ? fa[0][0] : frag((2 * i) % 12);
        const v4i b = (MODE == 24) ? fb[0][0] : frag((2 * i + 1) % 12);
        acc[i & 7] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v8i{b[0], b[1], b[2], b[3], 0, 0, 0, 0}, v8i{a[0], a[1], a[2], a[3], 0, 0, 0, 0},
                                                                     acc[i & 7], 4, 4, 0, scale, 0, scale);
      }
      fence();
    }
    if (MODE == 26) {   // the same FLOPs on v_mfma_scale_f32_16x16x128_f8f6f4: 32 MFMAs of 16x16x128 = 16 of 32x32x64
#pragma unroll
      for (int set = 0; set < 2; ++set)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const v4i a = fa[set][i & 3], b = fb[set][(i >> 2) & 1];
          acc4[set * 16 + i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(v8i{b[0], b[1], b[2], b[3], 0, 0, 0, 0}, v8i{a[0], a[1], a[2], a[3], 0, 0, 0, 0},
                                                                                acc4[set * 16 + i], 4, 4, 0, scale, 0, scale);
        }
      fence();
    }
    if (MODE == 1) { reads(0, off0); keep(0); fence(); reads(1, off1); keep(1); fence(); }
    if (MODE == 2 || MODE == 4) {
      mfma8(0); fence(); reads(1, off0); keep(0); fence();
      if (MODE == 4) { dma2(it); fence(); }
      mfma8(1); fence(); reads(0, off1); keep(1); fence();
      if (MODE == 4) { dma2(it + 1); fence(); }
    }
    if (MODE == 3) { reads(0, off0); fence(); mfma8(0); fence(); reads(0, off1); fence(); mfma8(0); fence(); }
    if (MODE == 5) {
      reads(1, off0); fence(); mfma8(0); fence();
      reads(0, off1); fence(); mfma8(1); fence();
    }
    if (MODE == 6) { dma2(it); fence(); dma2(it + 1); fence(); }
    if (MODE == 8 || MODE == 9 || MODE == 10) {   // mode 2 + stage hand-off every 4 slices: 8 = lgkmcnt(0)+barrier, 9 = lgkmcnt(0) only, 10 = barrier only
      mfma8(0); fence(); reads(1, off0); keep(0); fence();
      mfma8(1); fence();
      if ((it & 2) == 2) {
        if (MODE != 10) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (MODE != 9) __builtin_amdgcn_s_barrier();
        fence();
      }
      reads(0, off1); keep(1); fence();
    }
    if (MODE == 11 || MODE == 12) {   // single fragment set, R -> M per slice; DMA 3,3,2,0 over the 4 slices of a "stage"; hand-off after slice 3
      const int ph = it & 2;          // it advances by 2 slices: ph == 0 -> slices 0,1 ; ph == 2 -> slices 2,3
      reads(0, off0); fence(); mfma8(0); fence();
      if (MODE == 11) { if (ph == 0) { dma2(it); dma1(it); } else { dma2(it); } fence(); }
      reads(0, off1); fence(); mfma8(0); fence();
      if (MODE == 11 && ph == 0) { dma2(it + 1); dma1(it + 1); fence(); }
      if (ph == 2) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); fence(); }
    }
    if (MODE == 13 || MODE == 14) {   // [r3] the same work as modes 2 / 4, but ONE fragment read (and, 14, one LDS-DMA) behind each MFMA instead of bursts of 8 + 6 (+ 2)
      auto half = [&](const int cur, const int oth, const int off, const int itx) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const v4i a = fa[cur][i >> 1], b = fb[cur][i & 1];
          acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v8i{b[0], b[1], b[2], b[3], 0, 0, 0, 0}, v8i{a[0], a[1], a[2], a[3], 0, 0, 0, 0}, acc[i], 4, 4, 0, scale, 0, scale);
          if (i < 4) fa[oth][i] = *(const v4i*)(smem + addr(off) + i * 4096);
          else if (i < 6) fb[oth][i - 4] = *(const v4i*)(smem + 32768 + addr(off) + (i - 4) * 4096);
          else if (MODE == 14) {
            const int v = lane * 16 + (itx & 63) * 2048 + wave * 131072 + (i - 6) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem + 65536 + wave * 2048 + (i - 6) * 1024), 16, v, 0, 0, 0);
          }
          fence();
        }
      };
      half(0, 1, off0, it);
      half(1, 0, off1, it + 1);
    }
    if (MODE == 7) {
      mfma8(0); fence(); reads(1, off0); keep(0); fence(); dma2(it); fence();
      mfma8(1); fence(); reads(0, off1); keep(1); fence(); dma2(it + 1); fence();
      if ((it & 2) == 2) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); fence(); }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  if (MODE == 26) for (int a = 0; a < 32; ++a) s += acc4[a][0] + acc4[a][1] + acc4[a][2] + acc4[a][3];
  for (int st = 0; st < 2; ++st) { for (int t = 0; t < 4; ++t) s += (float)fa[st][t][0]; for (int t = 0; t < 2; ++t) s += (float)fb[st][t][1]; }
  out[blockIdx.x * THREADS + tid] = s + ((float*)smem)[16384 + tid];
  if (blockIdx.x == 0 && lane == 0) cyc[wave] = (uint32_t)(t1 - t0);
}

static int g_random_fill = 0;
static int g_steady_warm = 0;
static const uint32_t* g_fill = nullptr;   // device pointer: 64 KiB operand image (overrides g_random_fill)
static const char* g_fill_name = nullptr;
template <int MODE, int THREADS>
static void run_one(const char* name, const char* g, uint32_t gbytes, float* out, uint32_t* cyc, int blocks) {
  const int iters = 20000;
  // g_steady_warm launches first: the part needs ~50 ms under load to leave its clock ramp (tools/clock_ramp.py)
  for (int w = 0; w < (g_steady_warm ? g_steady_warm : 1); ++w) ub_kernel<MODE, THREADS><<<blocks, THREADS>>>(g, gbytes, out, cyc, iters, g_random_fill, g_fill);
  HIP_OK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  HIP_OK(hipEventRecord(e0, 0));
  ub_kernel<MODE, THREADS><<<blocks, THREADS>>>(g, gbytes, out, cyc, iters, g_random_fill, g_fill);
  HIP_OK(hipEventRecord(e1, 0));
  HIP_OK(hipEventSynchronize(e1));
  float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  uint32_t h[8];
  HIP_OK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
  printf("UBENCH %s mode %d %-44s waves/SIMD=%d blocks=%3d : %7.1f cycles/iter (wave0)  %7.1f ns/iter  -> clock %.2f GHz\n", g_fill_name ? g_fill_name : g_random_fill ? "[random operands]" : "[const operands] ", MODE, name, THREADS / 256, blocks,
         (double)h[0] / iters, ms * 1e6 / iters, (double)h[0] / (ms * 1e6));
  hipEventDestroy(e0); hipEventDestroy(e1);
}

// DMA streaming patterns with the GEMM's real addressing (256 CUs, each streaming a 256-row A panel and a 256-row B
// panel of a 4096 x 2048-byte operand, 16 CUs per panel):
//   PAT 0: pieces of 8 rows x 128 B (whole cache lines), 64 KiB per stage = 8 pieces per wave
//   PAT 1: pieces of 16 rows x 64 B (half lines; the other half is fetched one stage later), 32 KiB per stage
template <int PAT>
__global__ __launch_bounds__(512) void dma_pattern_kernel(const char* A, const char* B, uint32_t bytes, float* out, uint32_t* cyc, int sweeps) {
  __shared__ __attribute__((aligned(16))) char smem[144 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rowbytes = 2048, tile_m = blockIdx.x / 16, tile_n = blockIdx.x % 16;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(A) + tile_m * 256 * rowbytes, 0, bytes - tile_m * 256 * rowbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(B) + tile_n * 256 * rowbytes, 0, bytes - tile_n * 256 * rowbytes, 0x00020000);
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int sw = 0; sw < sweeps; ++sw) {
    if (PAT == 0) {
      for (int kt = 0; kt < 16; ++kt) {
        char* st = smem + (kt & 1) * 65536;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int q = wave * 4 + t, v = (8 * q + (lane >> 3)) * rowbytes + (lane & 7) * 16;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(st + q * 1024), 16, v, kt * 128, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr_t)(st + 32768 + q * 1024), 16, v, kt * 128, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      }
    } else {
      for (int kt = 0; kt < 32; ++kt) {
        char* st = smem + (kt & 3) * 32768;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int q = wave * 2 + t, v = (16 * q + (lane >> 2)) * rowbytes + (lane & 3) * 16;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(st + q * 1024), 16, v, kt * 64, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr_t)(st + 16384 + q * 1024), 16, v, kt * 64, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint64_t t1 = __builtin_readcyclecounter();
  __syncthreads();
  out[blockIdx.x * 512 + tid] = ((float*)smem)[tid];
  if (blockIdx.x == 0 && lane == 0) cyc[wave] = (uint32_t)(t1 - t0);
}

template <int PAT>
static void run_dma_pattern(const char* name, const char* A, const char* B, uint32_t bytes, float* out, uint32_t* cyc) {
  const int sweeps = 20;
  dma_pattern_kernel<PAT><<<256, 512>>>(A, B, bytes, out, cyc, sweeps);
  HIP_OK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  HIP_OK(hipEventRecord(e0, 0));
  dma_pattern_kernel<PAT><<<256, 512>>>(A, B, bytes, out, cyc, sweeps);
  HIP_OK(hipEventRecord(e1, 0));
  HIP_OK(hipEventSynchronize(e1));
  float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  const double per_sweep_us = ms * 1e3 / sweeps;   // one sweep = 1 MiB per CU = the whole K=4096 loop of one 256x256 fp4 tile
  printf("UBENCH dma-pattern %-40s : %7.2f us per K-sweep (1 MiB/CU, 256 CUs) = %6.1f GB/s per CU, %5.2f TB/s aggregate L2->LDS\n", name, per_sweep_us,
         1048576.0 / per_sweep_us * 1e-3, 256 * 1048576.0 / per_sweep_us * 1e-6);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

void run_ubench() {
  {
    const uint32_t bytes = 8u << 20;
    char *A, *B; float* o; uint32_t* c;
    HIP_OK(hipMalloc(&A, bytes)); HIP_OK(hipMalloc(&B, bytes)); HIP_OK(hipMemset(A, 1, bytes)); HIP_OK(hipMemset(B, 2, bytes));
    HIP_OK(hipMalloc(&o, 256 * 512 * 4)); HIP_OK(hipMalloc(&c, 64));
    run_dma_pattern<0>("8 rows x 128 B pieces, 2 x 64 KiB ring", A, B, bytes, o, c);
    run_dma_pattern<1>("16 rows x 64 B pieces, 4 x 32 KiB ring", A, B, bytes, o, c);
    run_dma_pattern<0>("8 rows x 128 B pieces, 2 x 64 KiB ring", A, B, bytes, o, c);
    run_dma_pattern<1>("16 rows x 64 B pieces, 4 x 32 KiB ring", A, B, bytes, o, c);
    hipFree(A); hipFree(B); hipFree(o); hipFree(c);
  }
  const uint32_t gbytes = 64u << 20;
  char* g; float* out; uint32_t* cyc;
  HIP_OK(hipMalloc(&g, gbytes)); HIP_OK(hipMemset(g, 0x22, gbytes));
  HIP_OK(hipMalloc(&out, 256 * 512 * 4)); HIP_OK(hipMalloc(&cyc, 64));
  for (int rf : {0, 1}) {
   g_random_fill = rf;
   for (int blocks : {256}) {
    run_one<0, 256>("MFMA x8 only", g, gbytes, out, cyc, blocks);
    run_one<0, 512>("MFMA x8 only", g, gbytes, out, cyc, blocks);
    run_one<1, 256>("ds_read_b128 x6 only", g, gbytes, out, cyc, blocks);
    run_one<1, 512>("ds_read_b128 x6 only", g, gbytes, out, cyc, blocks);
    run_one<2, 256>("MFMA(cur) ; reads->other set", g, gbytes, out, cyc, blocks);
    run_one<2, 512>("MFMA(cur) ; reads->other set", g, gbytes, out, cyc, blocks);
    run_one<5, 256>("reads->other set ; MFMA(cur)", g, gbytes, out, cyc, blocks);
    run_one<5, 512>("reads->other set ; MFMA(cur)", g, gbytes, out, cyc, blocks);
    run_one<3, 256>("reads ; MFMA same set (serialised)", g, gbytes, out, cyc, blocks);
    run_one<3, 512>("reads ; MFMA same set (serialised)", g, gbytes, out, cyc, blocks);
    run_one<6, 256>("LDS-DMA x2 only", g, gbytes, out, cyc, blocks);
    run_one<6, 512>("LDS-DMA x2 only", g, gbytes, out, cyc, blocks);
    run_one<4, 256>("MFMA ; reads->other ; LDS-DMA x2", g, gbytes, out, cyc, blocks);
    run_one<4, 512>("MFMA ; reads->other ; LDS-DMA x2", g, gbytes, out, cyc, blocks);
    run_one<8, 256>("mode 2 + lgkmcnt(0)+barrier every 4 slices", g, gbytes, out, cyc, blocks);
    run_one<8, 512>("mode 2 + lgkmcnt(0)+barrier every 4 slices", g, gbytes, out, cyc, blocks);
    run_one<9, 512>("mode 2 + lgkmcnt(0) only every 4 slices", g, gbytes, out, cyc, blocks);
    run_one<10, 512>("mode 2 + barrier only every 4 slices", g, gbytes, out, cyc, blocks);
    run_one<11, 512>("single set R->M, DMA 3,3,2,0 + hand-off / 4 slices", g, gbytes, out, cyc, blocks);
    run_one<12, 512>("single set R->M, no DMA, hand-off / 4 slices", g, gbytes, out, cyc, blocks);
    run_one<7, 256>("mode 4 + vmcnt(0)+barrier every 4 slices", g, gbytes, out, cyc, blocks);
    run_one<7, 512>("mode 4 + vmcnt(0)+barrier every 4 slices", g, gbytes, out, cyc, blocks);
   }
  }
  hipFree(g); hipFree(out); hipFree(cyc);
}

// ------------------------------------------------------------------------------------------------
// VALU issue rates of the NVFP4 dequant instructions (one wave per SIMD, s_memtime around an unrolled
// run of independent instructions): decides whether the f16 path is cvt-bound.
// ------------------------------------------------------------------------------------------------
typedef _Float16 ub_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 ub_h8 __attribute__((ext_vector_type(8)));
typedef float ub_f16v __attribute__((ext_vector_type(16)));

template <int OP>
__global__ __launch_bounds__(256) void valu_rate_kernel(uint32_t* outv, uint64_t* cyc, int iters, uint32_t seed) {
  uint32_t w[8];
  ub_h2 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { w[i] = seed * (threadIdx.x + 1) * (i + 3); acc[i] = ub_h2{(_Float16)1.0f, (_Float16)1.0f}; }
  ub_f16v c0 = {}, c1 = {};
  ub_h8 fa = {(_Float16)1, (_Float16)2, (_Float16)3, (_Float16)4, (_Float16)1, (_Float16)2, (_Float16)3, (_Float16)4};
  ub_h8 fb = fa;
  if (OP == 6) {   // random finite f16 operands (sign + 5-bit exponent around 1 + random mantissa)
    uint32_t x = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    uint32_t r4[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      x = x * 1664525u + 1013904223u;
      const uint32_t lo = (x & 0x83ffu) | (((x >> 10) % 5 + 13) << 10);
      const uint32_t hi = ((x >> 16) & 0x83ffu) | ((((x >> 26) % 5) + 13) << 10);
      r4[i] = lo | (hi << 16);
    }
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    fa = __builtin_bit_cast(ub_h8, (u4){r4[0], r4[1], r4[2], r4[3]});
    fb = __builtin_bit_cast(ub_h8, (u4){r4[4], r4[5], r4[6], r4[7]});
  }
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (OP == 6) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fa, c1, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) acc[i] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w[i], 1.0f, 0) ;           // 32 cvt
        if (OP == 1) acc[i] = acc[i] * ub_h2{(_Float16)1.0f, (_Float16)1.0009765625f};           // 32 pk_mul (dependent per i, 8 chains)
        if (OP == 2) w[i] = __builtin_amdgcn_perm(w[i], w[(i + 1) & 7], 0x05010400u + r);       // 32 perm
        if (OP == 3) { acc[i] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w[i], 1.0f, 0) * acc[i]; }   // 32 cvt + 32 mul
      }
      if (OP == 4 || OP == 5) {   // 2 MFMAs (+ 32 cvt/mul pairs for OP 5) per r
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fa, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fa, c1, 0, 0, 0);
        if (OP == 5) {
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w[i], 1.0f, 0) * acc[i];
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(w[i]), "+v"(acc[i]));
    }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += w[i] + __builtin_bit_cast(uint32_t, acc[i]);
  outv[blockIdx.x * 256 + threadIdx.x] = s + (uint32_t)(c0[0] + c1[3]);
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
static void run_valu(const char* name, int per_iter, int waves_per_simd) {
  uint32_t* o; uint64_t* c;
  HIP_OK(hipMalloc(&o, 256 * 512 * 4)); HIP_OK(hipMalloc(&c, 8));
  const int iters = 20000;
  valu_rate_kernel<OP><<<256 * waves_per_simd, 256>>>(o, c, iters, 12345u);
  HIP_OK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  HIP_OK(hipEventRecord(e0, 0));
  valu_rate_kernel<OP><<<256 * waves_per_simd, 256>>>(o, c, iters, 12345u);
  HIP_OK(hipEventRecord(e1, 0));
  HIP_OK(hipEventSynchronize(e1));
  float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  uint64_t cy; HIP_OK(hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost));
  printf("UBENCH valu %-44s waves/SIMD=%d : %8.1f ns per iteration-group of %d, s_memtime ticks/group %.1f\n", name, waves_per_simd,
         ms * 1e6 / iters, per_iter, (double)cy / iters);
  hipFree(o); hipFree(c); hipEventDestroy(e0); hipEventDestroy(e1);
}

void run_ubench_steady() {   // MFMA-only ceilings in the steady state (after ~100 ms under load)
  const uint32_t gbytes = 64u << 20;
  char* g; float* out; uint32_t* cyc;
  HIP_OK(hipMalloc(&g, gbytes)); HIP_OK(hipMemset(g, 0x22, gbytes));
  HIP_OK(hipMalloc(&out, 256 * 512 * 4)); HIP_OK(hipMalloc(&cyc, 64));
  g_steady_warm = 40;
  for (int rf : {0, 1, 0, 1}) {
    g_random_fill = rf;
    run_one<0, 256>("MFMA x8 only (steady)", g, gbytes, out, cyc, 256);
    run_one<0, 512>("MFMA x8 only (steady)", g, gbytes, out, cyc, 256);
    run_one<20, 256>("MFMA x8, ONE accumulator (chain 8)", g, gbytes, out, cyc, 256);
    run_one<20, 512>("MFMA x8, ONE accumulator (chain 8)", g, gbytes, out, cyc, 256);
    run_one<21, 256>("MFMA x8, chains of 4 on one accumulator", g, gbytes, out, cyc, 256);
    run_one<21, 512>("MFMA x8, chains of 4 on one accumulator", g, gbytes, out, cyc, 256);
    run_one<22, 256>("MFMA x8, chains of 2 on one accumulator", g, gbytes, out, cyc, 256);
    run_one<22, 512>("MFMA x8, chains of 2 on one accumulator", g, gbytes, out, cyc, 256);
    run_one<2, 512>("MFMA(cur) ; reads->other set (steady)", g, gbytes, out, cyc, 256);
    run_one<4, 512>("MFMA ; reads->other ; LDS-DMA x2 (steady)", g, gbytes, out, cyc, 256);
  }
  g_steady_warm = 0;
  hipFree(g); hipFree(out); hipFree(cyc);
}

std::vector<uint32_t> gaussian_e2m1_image(size_t bytes, uint32_t seed);
void run_ubench_interleave() {   // [r3] bursts vs one-behind-each-MFMA, ONE wave per SIMD (256 threads, the GEMM's occupancy), steady state
  const uint32_t gbytes = 64u << 20;
  char* g; float* out; uint32_t* cyc; uint32_t* fill;
  HIP_OK(hipMalloc(&g, gbytes)); HIP_OK(hipMemset(g, 0x22, gbytes));
  HIP_OK(hipMalloc(&out, 256 * 512 * 4)); HIP_OK(hipMalloc(&cyc, 64)); HIP_OK(hipMalloc(&fill, 65536));
  std::vector<uint32_t> gauss = gaussian_e2m1_image(65536, 7);
  HIP_OK(hipMemcpy(fill, gauss.data(), 65536, hipMemcpyHostToDevice));
  g_steady_warm = 40;
  for (int cls : {0, 2, 0, 2}) {
    g_fill = cls == 2 ? fill : nullptr; g_fill_name = cls == 2 ? "[gaussian codes]  " : nullptr; g_random_fill = 0;
    run_one<0, 256>("MFMA x8 only", g, gbytes, out, cyc, 256);
    run_one<2, 256>("bursts: 8 MFMA ; 6 reads -> other set", g, gbytes, out, cyc, 256);
    run_one<13, 256>("interleaved: MFMA, read, MFMA, read, ...", g, gbytes, out, cyc, 256);
    run_one<4, 256>("bursts: 8 MFMA ; 6 reads ; 2 LDS-DMA", g, gbytes, out, cyc, 256);
    run_one<14, 256>("interleaved: ... + LDS-DMA behind MFMA 7, 8", g, gbytes, out, cyc, 256);
  }
  g_steady_warm = 0; g_fill = nullptr; g_fill_name = nullptr;
  hipFree(g); hipFree(out); hipFree(cyc); hipFree(fill);
}

void run_valu_rates() {
  for (int w : {1, 2}) {
    run_valu<0>("32 x v_cvt_scalef32_pk_f16_fp4", 32, w);
    run_valu<1>("32 x v_pk_mul_f16", 32, w);
    run_valu<2>("32 x v_perm_b32", 32, w);
    run_valu<3>("32 x (cvt + pk_mul)", 64, w);
    run_valu<4>("8 x v_mfma_f32_32x32x16_f16", 8, w);
    run_valu<5>("8 x MFMA f16 + 32 x (cvt + pk_mul)", 72, w);
    run_valu<6>("8 x v_mfma_f32_32x32x16_f16, RANDOM operands", 8, w);
  }
}


// ------------------------------------------------------------------------------------------------
// Power / clock traces: socket power and shader clock sampled through librocm_smi64 from a host
// thread (>= 100 Hz) while one kernel runs back to back for a fixed wall time.  Operand classes of the FP4 MFMA loop:
//   const      every nibble 0x2 (= 1.0): the data class micro-benchmarks that quote ~9 PF use
//   uniform    uniformly random bytes (worst-case toggling)
//   gaussian   e2m1 codes of N(0,1) data quantised per 32 with the abs-max rule of fusedQuantizeMx (what the GEMM sees)
// ------------------------------------------------------------------------------------------------
// 64 KiB image of e2m1 codes: N(0,1) values, groups of 32, scale = 2^floor(log2(amax)), q = RTNE_e2m1(3 x / scale)
std::vector<uint32_t> gaussian_e2m1_image(size_t bytes, uint32_t seed) {
  std::mt19937 rng(seed);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<uint8_t> out(bytes);
  const float grid[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
  for (size_t g0 = 0; g0 < bytes * 2; g0 += 32) {
    float x[32], amax = 0.f;
    for (int i = 0; i < 32; ++i) { x[i] = nd(rng); amax = std::max(amax, std::fabs(x[i])); }
    const float sc = std::exp2(std::floor(std::log2(amax)));
    for (int i = 0; i < 32; ++i) {
      const float v = std::fabs(x[i]) / sc * 3.0f;
      int best = 0;
      for (int c = 1; c < 8; ++c) if (std::fabs(v - grid[c]) < std::fabs(v - grid[best]) || (std::fabs(v - grid[c]) == std::fabs(v - grid[best]) && !(c & 1))) best = c;
      const uint8_t code = (uint8_t)(best | (x[i] < 0 ? 8 : 0));
      const size_t e = g0 + i;
      if (e & 1) out[e >> 1] |= (uint8_t)(code << 4); else out[e >> 1] = code;
    }
  }
  std::vector<uint32_t> w(bytes / 4);
  memcpy(w.data(), out.data(), bytes);
  return w;
}

template <int MODE, int THREADS>
static void power_run(SmiSampler& smi, std::vector<std::pair<double, std::string>>& marks, const char* cls, const char* name, const char* g, uint32_t gbytes, float* out,
                      uint32_t* cyc, double seconds, double flop_per_iter) {
  const int iters = 20000;
  char tag[160];
  snprintf(tag, sizeof tag, "%s mode %d %s", cls, MODE, name);
  marks.push_back({smi.now_ms(), tag});
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  const auto t_begin = std::chrono::steady_clock::now();
  auto el = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
  while (el() < seconds * 0.35) {   // ramp
    for (int i = 0; i < 4; ++i) ub_kernel<MODE, THREADS><<<256, THREADS>>>(g, gbytes, out, cyc, iters, g_random_fill, g_fill);
    HIP_OK(hipDeviceSynchronize());
  }
  const double a_ms = smi.now_ms();
  HIP_OK(hipEventRecord(e0, 0));
  int n = 0;
  while (el() < seconds) {
    for (int i = 0; i < 4; ++i) ub_kernel<MODE, THREADS><<<256, THREADS>>>(g, gbytes, out, cyc, iters, g_random_fill, g_fill);
    n += 4;
    HIP_OK(hipStreamSynchronize(0));
  }
  HIP_OK(hipEventRecord(e1, 0));
  HIP_OK(hipEventSynchronize(e1));
  const double b_ms = smi.now_ms();
  float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  uint32_t h[8];
  HIP_OK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
  double w, mhz; int ns;
  smi.mean(a_ms + 20, b_ms, w, mhz, ns);
  const double ns_iter = ms * 1e6 / ((double)n * iters);
  printf("POWER %-10s mode %2d %-46s %7.1f ns/iter  %6.0f TFLOP/s  cycles/iter %6.1f -> %4.2f GHz (s_memtime)   socket %6.0f W  sclk(smi) %5.0f MHz  (%d samples)\n", cls, MODE, name, ns_iter,
         flop_per_iter * 256 * (THREADS / 64) / ns_iter * 1e-3, (double)h[0] / iters, (double)h[0] / iters / ns_iter, w, mhz, ns);
  fflush(stdout);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

void run_ubench_power(const char* csv_path) {
  const uint32_t gbytes = 64u << 20;
  char* g; float* out; uint32_t* cyc; uint32_t* fill;
  HIP_OK(hipMalloc(&g, gbytes)); HIP_OK(hipMemset(g, 0x22, gbytes));
  HIP_OK(hipMalloc(&out, 256 * 512 * 4)); HIP_OK(hipMalloc(&cyc, 64)); HIP_OK(hipMalloc(&fill, 65536));
  std::vector<uint32_t> gauss = gaussian_e2m1_image(65536, 7);
  SmiSampler smi;
  std::vector<std::pair<double, std::string>> marks;
  smi.start();
  std::this_thread::sleep_for(std::chrono::milliseconds(500));
  marks.push_back({0.0, "idle"});
  const double sec = 0.8;
  const double F = 8.0 * 2 * 32 * 32 * 64;   // flop per wave per "iteration" (8 MFMAs of 32x32x64)
  for (int cls = 0; cls < 3; ++cls) {
    const char* cn = cls == 0 ? "const" : cls == 1 ? "uniform" : "gaussian";
    g_random_fill = cls == 1;
    g_fill = nullptr;
    if (cls == 2) { HIP_OK(hipMemcpy(fill, gauss.data(), 65536, hipMemcpyHostToDevice)); g_fill = fill; }
    g_fill_name = cls == 2 ? "[gaussian codes]  " : nullptr;
    power_run<0, 256>(smi, marks, cn, "MFMA x8, A held for 2, B alternates (GEMM order)", g, gbytes, out, cyc, sec, F);
    power_run<24, 256>(smi, marks, cn, "MFMA, SAME A and B every time", g, gbytes, out, cyc, sec, F);
    power_run<25, 256>(smi, marks, cn, "MFMA, A and B both change every time", g, gbytes, out, cyc, sec, F);
    power_run<20, 256>(smi, marks, cn, "MFMA x8 into ONE accumulator", g, gbytes, out, cyc, sec, F);
    power_run<26, 256>(smi, marks, cn, "32 x v_mfma_scale 16x16x128 (same FLOPs)", g, gbytes, out, cyc, sec, F);
    power_run<2, 256>(smi, marks, cn, "MFMA + fragment reads", g, gbytes, out, cyc, sec, F);
    power_run<4, 256>(smi, marks, cn, "MFMA + fragment reads + LDS-DMA", g, gbytes, out, cyc, sec, F);
  }
  g_fill = nullptr; g_fill_name = nullptr; g_random_fill = 0;
  std::this_thread::sleep_for(std::chrono::milliseconds(300));
  smi.finish();
  if (csv_path) smi.dump(csv_path, marks);
  hipFree(g); hipFree(out); hipFree(cyc); hipFree(fill);
}


// ------------------------------------------------------------------------------------------------
// Output-store burst of the 4096^3 GEMM in isolation: 256 workgroups x 4 waves, every wave writes its 128 x 128 bf16
// quadrant of a 256 x 256 tile (32 MiB in total) from registers, as 32-row x 32-column pieces in the order the GEMM retires them.
//   PAT 0: 16 rows x 64 B per wave instruction (what gemm_mx_deepp does)    PAT 1: 8 rows x 128 B (pairs of pieces, whole lines)
//   PAT 2: 32 rows x 32 B (the register-direct epilogue's pattern)          AUX: cache policy bits (1 sc0, 2 nt, 16 sc1)
// ------------------------------------------------------------------------------------------------
template <int PAT, int AUX>
__global__ __launch_bounds__(256) void store_pattern_kernel(uint16_t* D, int ldd, uint32_t seed) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile_m = blockIdx.x / 16, tile_n = blockIdx.x % 16, wave_m = wave >> 1, wave_n = wave & 1;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(D + ((size_t)(tile_m * 256) * ldd + tile_n * 256), 0, 0x7fffffff, 0x00020000);
  typedef unsigned int v4u_ __attribute__((ext_vector_type(4)));
  v4u_ v = {seed ^ (uint32_t)threadIdx.x, seed * 3u, seed * 5u + blockIdx.x, seed * 7u};
  for (int m = 0; m < 4; ++m)
    for (int n = 0; n < 4; ++n) {
      if (PAT == 0) {
        const int rr = lane >> 2, cc = lane & 3;
        for (int ps = 0; ps < 2; ++ps) {
          const int off = ((wave_m * 128 + 32 * m + 16 * ps + rr) * ldd + wave_n * 128 + 32 * n + 8 * cc) * 2;
          __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, AUX);
          v[0] += 1;
        }
      } else if (PAT == 1) {
        if (n & 1) continue;   // a pair of pieces (n, n + 1) = 32 rows x 128 B: 4 instructions of 8 rows x 128 B
        const int rr = lane >> 3, cc = lane & 7;
        for (int ps = 0; ps < 4; ++ps) {
          const int off = ((wave_m * 128 + 32 * m + 8 * ps + rr) * ldd + wave_n * 128 + 32 * n + 8 * cc) * 2;
          __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, AUX);
          v[0] += 1;
        }
      } else {
        const int rr = lane & 31, cc = lane >> 5;
        for (int ps = 0; ps < 2; ++ps) {
          const int off = ((wave_m * 128 + 32 * m + rr) * ldd + wave_n * 128 + 32 * n + 16 * ps + 8 * cc) * 2;
          __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, AUX);
          v[0] += 1;
        }
      }
    }
}

template <int PAT, int AUX>
static void run_store_pattern(const char* name, uint16_t* D) {
  auto go = [&](int n) { for (int i = 0; i < n; ++i) store_pattern_kernel<PAT, AUX><<<256, 256>>>(D, 4096, 17u + i); };
  go(3000);
  HIP_OK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  HIP_OK(hipEventRecord(e0, 0));
  go(3000);
  HIP_OK(hipEventRecord(e1, 0));
  HIP_OK(hipEventSynchronize(e1));
  float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / 3000;
  printf("UBENCH stores %-58s aux %2d : %6.2f us per 32 MiB launch = %5.2f TB/s (launch-to-launch, incl. dispatch + end-of-kernel write-back)\n", name, AUX, us, 33.554432 / us);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

__global__ void empty_kernel_256(uint32_t* p) { if (p == nullptr && threadIdx.x == 1234567) *p = 1; }

void run_store_patterns() {
  uint16_t* D;
  HIP_OK(hipMalloc(&D, 4096ull * 4096 * 2));
  {
    for (int i = 0; i < 3000; ++i) empty_kernel_256<<<256, 256>>>((uint32_t*)D);
    HIP_OK(hipDeviceSynchronize());
    hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, 0));
    for (int i = 0; i < 3000; ++i) empty_kernel_256<<<256, 256>>>((uint32_t*)D);
    HIP_OK(hipEventRecord(e1, 0)); HIP_OK(hipEventSynchronize(e1));
    float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    printf("UBENCH empty kernel, 256 workgroups x 256 threads: %6.2f us launch-to-launch\n", ms * 1e3 / 3000);
  }
  for (int rep = 0; rep < 2; ++rep) {
    run_store_pattern<0, 0>("16 rows x 64 B per instruction (gemm_mx_deepp)", D);
    run_store_pattern<1, 0>("8 rows x 128 B per instruction (whole lines)", D);
    run_store_pattern<2, 0>("32 rows x 32 B per instruction (register-direct)", D);
    run_store_pattern<0, 2>("16 rows x 64 B", D);
    run_store_pattern<0, 16>("16 rows x 64 B", D);
    run_store_pattern<0, 17>("16 rows x 64 B", D);
    run_store_pattern<0, 19>("16 rows x 64 B", D);
    run_store_pattern<1, 2>("8 rows x 128 B", D);
    run_store_pattern<1, 17>("8 rows x 128 B", D);
  }
  hipFree(D);
}
