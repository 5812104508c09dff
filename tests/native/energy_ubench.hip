// Joules per unit of work of the pieces the MXFP4 GEMM is made of (test / bench infrastructure, own main()).
// The headline launch runs at the socket power cap, so what it costs is energy: time = energy / (cap - idle).  This file prices the
// pieces one at a time with every CU busy (256 workgroups x 4 waves, one wave per SIMD = the GEMM's occupancy), socket power sampled
// through librocm_smi64 (smi_sampler.h) while one kernel runs back to back for ~0.7 s:
//   (a) the scaled FP4 MFMA alone on quantised-Gaussian operands (what fusedQuantizeMx hands the GEMM), 64 MFMAs of a 128x128 wave tile per
//       iteration (four k-slices), in the GEMM's order / serpentine / operand roles swapped / unit scales / accumulator chains;
//       and a 128x64 wave tile (8 accumulator tiles) that a second build (-mllvm -amdgpu-mfma-vgpr-form) keeps in VGPRs instead of AGPRs;
//   (b) the operand fill: LDS fragment reads (ds_read_b128 / 2 x ds_read_b64), and the LDS-DMA (buffer_load_dwordx4 ... lds, 1-KiB pieces, the
//       GEMM's piece shape) from a source that lives in the vector L1 / the XCD's L2 / the memory-side cache / HBM.
// Output: one line per piece -- ns per iteration, socket W, shader clock, energy per iteration above the idle socket power, and pJ per byte
// (fill) or pJ per MFMA flop -- the table under profiles/energy_ubench_r6*.txt.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 energy_ubench.hip -o energy_ubench -lrocm_smi64
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "smi_sampler.h"

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP ERROR %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2);} } while (0)

#ifndef EU_BUILD_NAME
#define EU_BUILD_NAME "default"
#endif

struct Args {
  const uint32_t* codes;   // 64 KiB of e2m1 codes per workgroup class (quantised Gaussian) -- fragment registers are filled from it
  const uint32_t* scales;  // e8m0 bytes, 4 per dword
  const char* src;         // DMA source
  uint32_t src_bytes;
  uint32_t region;         // bytes of the source one workgroup cycles through
  uint32_t region_stride;  // distance between the regions of two workgroups
  float* out;
  int iters;
  int unit_scales;
};

// MODE  0 GEMM order   1 serpentine   2 roles swapped   3 accumulator-stationary (4 k-slices back to back per tile)   4 unit scales (args)
//      10 128x64 wave tile (8 accumulator tiles, 32 MFMAs per iteration): AGPR vs VGPR accumulators by build
//      20 8 x ds_read_b128 per k-slice (32 per iteration)   21 the same bytes as 16 x ds_read_b64
//      30 LDS-DMA: 16 pieces of 1 KiB per wave and iteration (64 KiB per workgroup = one K stage of the GEMM), source by args.region
template <int MODE>
__global__ __launch_bounds__(256) void eu_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 64 * 1024 / 4; i += 256) ((uint32_t*)smem)[i] = a.codes[(i + blockIdx.x * 977) & 16383];
  __syncthreads();
  // conflict-free fragment addressing of the GEMM: row = lane & 31, chunk c = 4 (lane >> 5) + j, stored at c ^ ((row >> 1) & 7)
  const int sw = ((lane & 31) >> 1) & 7;
  auto addr = [&](int j) __attribute__((always_inline)) { return (lane & 31) * 128 + (((4 * (lane >> 5) + j) ^ sw) << 4); };
  auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
  float sink = 0.f;

  if constexpr (MODE <= 4) {
    v16f acc[4][4];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    v4i fa[4][4], fb[4][4];
    for (int j = 0; j < 4; ++j)
      for (int t = 0; t < 4; ++t) {
        fa[j][t] = *(const v4i*)(smem + addr(j) + t * 4096);
        fb[j][t] = *(const v4i*)(smem + 32768 + addr(j) + t * 4096);
      }
    int sa[4], sb[4];
    for (int t = 0; t < 4; ++t) {
      sa[t] = a.unit_scales ? 0x7f7f7f7f : (int)a.scales[(lane & 31) + 32 * t + 128 * wave];
      sb[t] = a.unit_scales ? 0x7f7f7f7f : (int)a.scales[(lane & 31) + 32 * t + 128 * wave + 512];
    }
    auto mf = [&](const int j, const int m, const int n) __attribute__((always_inline)) {
      const v4i x = fa[j][m], y = fb[j][n];
      const v8i A8 = {x[0], x[1], x[2], x[3], 0, 0, 0, 0}, B8 = {y[0], y[1], y[2], y[3], 0, 0, 0, 0};
      if (MODE == 2) {
        if (j == 0) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A8, B8, acc[m][n], 4, 4, 0, sa[m], 0, sb[n]);
        if (j == 1) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A8, B8, acc[m][n], 4, 4, 1, sa[m], 1, sb[n]);
        if (j == 2) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A8, B8, acc[m][n], 4, 4, 2, sa[m], 2, sb[n]);
        if (j == 3) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A8, B8, acc[m][n], 4, 4, 3, sa[m], 3, sb[n]);
      } else {
        if (j == 0) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc[m][n], 4, 4, 0, sb[n], 0, sa[m]);
        if (j == 1) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc[m][n], 4, 4, 1, sb[n], 1, sa[m]);
        if (j == 2) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc[m][n], 4, 4, 2, sb[n], 2, sa[m]);
        if (j == 3) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc[m][n], 4, 4, 3, sb[n], 3, sa[m]);
      }
    };
    for (int it = 0; it < a.iters; ++it) {
      if (MODE == 3) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int j = 0; j < 4; ++j) { mf(j, m, n); fence(); }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) { mf(j, m, (MODE == 1 && (m & 1)) ? 3 - n : n); fence(); }
      }
      // (the accumulators would grow without bound over 10^4 iterations: halve the exponent range now and then -- one v_mul per 64 MFMAs would
      //  distort the picture, so instead the operands carry both signs and the sums random-walk; nothing to do)
    }
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) sink += acc[m][n][r];
  }

  if constexpr (MODE == 10) {
    v16f acc[4][2];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    v4i fa[4][4], fb[4][2];
    for (int j = 0; j < 4; ++j)
      for (int t = 0; t < 4; ++t) {
        fa[j][t] = *(const v4i*)(smem + addr(j) + t * 4096);
        if (t < 2) fb[j][t] = *(const v4i*)(smem + 32768 + addr(j) + t * 4096);
      }
    int sa[4], sb[2];
    for (int t = 0; t < 4; ++t) sa[t] = (int)a.scales[(lane & 31) + 32 * t + 128 * wave];
    for (int t = 0; t < 2; ++t) sb[t] = (int)a.scales[(lane & 31) + 32 * t + 128 * wave + 512];
    for (int it = 0; it < a.iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            const v4i x = fa[j][m], y = fb[j][n];
            const v8i A8 = {x[0], x[1], x[2], x[3], 0, 0, 0, 0}, B8 = {y[0], y[1], y[2], y[3], 0, 0, 0, 0};
            if (j == 0) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc[m][n], 4, 4, 0, sb[n], 0, sa[m]);
            if (j == 1) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc[m][n], 4, 4, 1, sb[n], 1, sa[m]);
            if (j == 2) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc[m][n], 4, 4, 2, sb[n], 2, sa[m]);
            if (j == 3) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B8, A8, acc[m][n], 4, 4, 3, sb[n], 3, sa[m]);
            fence();
          }
    }
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) sink += acc[m][n][r];
  }

  if constexpr (MODE == 20 || MODE == 21) {
    int x = 0;
    for (int it = 0; it < a.iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const char* p = smem + (t >= 4 ? 32768 : 0) + addr(j) + (t & 3) * 4096;
          if (MODE == 20) {
            v4i v = *(const v4i*)p;
            asm volatile("" ::"v"(v));
          } else {
            v2i v0 = *(const v2i*)p, v1 = *(const v2i*)(p + 8);
            asm volatile("" ::"v"(v0), "v"(v1));
          }
        }
        fence();
      }
    }
    sink += (float)x;
  }

  if constexpr (MODE == 30) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.src), 0, a.src_bytes, 0x00020000);
    const uint32_t base = blockIdx.x * a.region_stride;
    const uint32_t mask = a.region - 1;   // region: power of two, >= 64 KiB... or 16 KiB (then every wave re-reads its 4 KiB)
    for (int it = 0; it < a.iters; ++it) {
      const uint32_t off = ((uint32_t)it * 65536u) & mask;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const uint32_t v = base + ((off + (uint32_t)(wave * 16 + q) * 1024u) & mask) + lane * 16;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem + (wave * 16 + q) * 1024), 16, (int)v, 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // two iterations of pieces in flight, like the GEMM's two stages
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    sink += ((float*)smem)[tid];
  }
  if (sink == 12345.678f) a.out[blockIdx.x * 256 + tid] = sink;
}

// 64 KiB image of e2m1 codes: N(0,1) values, groups of 32, scale = 2^floor(log2(amax)), q = RTNE_e2m1(3 x / scale)  (as tests/native/ubench.hip)
static std::vector<uint32_t> gaussian_e2m1_image(size_t bytes, uint32_t seed, std::vector<uint8_t>* scales_out) {
  std::mt19937 rng(seed);
  std::normal_distribution<float> nd(0.f, 25.f);
  std::vector<uint8_t> out(bytes);
  const float grid[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
  for (size_t g0 = 0; g0 < bytes * 2; g0 += 32) {
    float x[32], amax = 0.f;
    for (int i = 0; i < 32; ++i) { x[i] = nd(rng); amax = std::max(amax, std::fabs(x[i])); }
    const float e = std::floor(std::log2(amax));
    const float sc = std::exp2(e);
    if (scales_out) scales_out->push_back((uint8_t)(127 + (int)e - 2));
    for (int i = 0; i < 32; ++i) {
      const float v = std::fabs(x[i]) / sc * 3.0f;
      int best = 0;
      for (int c = 1; c < 8; ++c) if (std::fabs(v - grid[c]) < std::fabs(v - grid[best]) || (std::fabs(v - grid[c]) == std::fabs(v - grid[best]) && !(c & 1))) best = c;
      const uint8_t code = (uint8_t)(best | (x[i] < 0 ? 8 : 0));
      const size_t el = g0 + i;
      if (el & 1) out[el >> 1] |= (uint8_t)(code << 4); else out[el >> 1] = code;
    }
  }
  std::vector<uint32_t> w(bytes / 4);
  memcpy(w.data(), out.data(), bytes);
  return w;
}

static double g_idle_w = 0;

template <int MODE>
static void run(SmiSampler& smi, const char* name, Args a, double seconds, double flop_per_iter_wg, double bytes_per_iter_wg) {
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&eu_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  // calibrate iters so that one launch takes ~2 ms
  a.iters = 200;
  HIP_OK(hipEventRecord(e0, 0));
  eu_kernel<MODE><<<256, 256, 128 * 1024>>>(a);
  HIP_OK(hipEventRecord(e1, 0)); HIP_OK(hipEventSynchronize(e1));
  eu_kernel<MODE><<<256, 256, 128 * 1024>>>(a);
  HIP_OK(hipEventRecord(e0, 0));
  eu_kernel<MODE><<<256, 256, 128 * 1024>>>(a);
  HIP_OK(hipEventRecord(e1, 0)); HIP_OK(hipEventSynchronize(e1));
  float ms0; HIP_OK(hipEventElapsedTime(&ms0, e0, e1));
  a.iters = std::max(200, (int)(200 * 2.0 / std::max(ms0, 1e-3f)));
  const auto t_begin = std::chrono::steady_clock::now();
  auto el = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
  while (el() < seconds * 0.4) {   // ramp: the power controller settles within ~100 ms
    for (int i = 0; i < 8; ++i) eu_kernel<MODE><<<256, 256, 128 * 1024>>>(a);
    HIP_OK(hipDeviceSynchronize());
  }
  const double a_ms = smi.now_ms();
  HIP_OK(hipEventRecord(e0, 0));
  long n = 0;
  while (el() < seconds) {
    for (int i = 0; i < 8; ++i) eu_kernel<MODE><<<256, 256, 128 * 1024>>>(a);
    n += 8;
    HIP_OK(hipStreamSynchronize(0));
  }
  HIP_OK(hipEventRecord(e1, 0)); HIP_OK(hipEventSynchronize(e1));
  const double b_ms = smi.now_ms();
  float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  double w, mhz; int ns;
  smi.mean(a_ms + 20, b_ms, w, mhz, ns);
  const double ns_iter = ms * 1e6 / ((double)n * a.iters);
  const double nj_iter = w * ns_iter, nj_dyn = (w - g_idle_w) * ns_iter;   // W x ns = nJ, whole chip per iteration
  printf("ENERGY %-8s mode %2d %-58s %8.1f ns/iter  %6.0f W  %5.0f MHz  %8.1f nJ/iter  %8.1f nJ/iter above idle", EU_BUILD_NAME, MODE, name, ns_iter, w, mhz, nj_iter, nj_dyn);
  if (flop_per_iter_wg > 0) printf("  %6.0f TFLOP/s  %.4f pJ/flop (%.4f above idle)", flop_per_iter_wg * 256 / ns_iter * 1e-3, nj_iter * 1e3 / (flop_per_iter_wg * 256), nj_dyn * 1e3 / (flop_per_iter_wg * 256));
  if (bytes_per_iter_wg > 0) printf("  %7.2f TB/s  %6.2f pJ/B (%.2f above idle)", bytes_per_iter_wg * 256 / ns_iter * 1e-3, nj_iter * 1e3 / (bytes_per_iter_wg * 256), nj_dyn * 1e3 / (bytes_per_iter_wg * 256));
  printf("  (%d samples)\n", ns);
  fflush(stdout);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

int main(int argc, char** argv) {
  const double sec = argc > 1 ? atof(argv[1]) : 0.7;
  const bool only_tile8 = argc > 2 && !strcmp(argv[2], "tile8");
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, 0));
  printf("DEVICE %s CUs=%d build=%s\n", prop.gcnArchName, prop.multiProcessorCount, EU_BUILD_NAME);
  std::vector<uint8_t> sc;
  std::vector<uint32_t> codes = gaussian_e2m1_image(65536, 7, &sc);
  std::vector<uint32_t> scw(1024);
  for (int i = 0; i < 1024; ++i) scw[i] = sc[(4 * i) % sc.size()] | sc[(4 * i + 1) % sc.size()] << 8 | sc[(4 * i + 2) % sc.size()] << 16 | (uint32_t)sc[(4 * i + 3) % sc.size()] << 24;
  uint32_t *d_codes, *d_sc; float* d_out; char* d_src;
  const size_t SRC = (size_t)1 << 30;
  HIP_OK(hipMalloc(&d_codes, 65536)); HIP_OK(hipMalloc(&d_sc, 4096)); HIP_OK(hipMalloc(&d_out, 256 * 256 * 4)); HIP_OK(hipMalloc(&d_src, SRC));
  HIP_OK(hipMemcpy(d_codes, codes.data(), 65536, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_sc, scw.data(), 4096, hipMemcpyHostToDevice));
  for (size_t o = 0; o < SRC; o += 65536) HIP_OK(hipMemcpyAsync(d_src + o, d_codes, 65536, hipMemcpyDeviceToDevice, 0));   // Gaussian codes everywhere
  HIP_OK(hipDeviceSynchronize());
  SmiSampler smi;
  smi.start();
  std::this_thread::sleep_for(std::chrono::milliseconds(1500));
  { double mhz; int n; smi.mean(300, smi.now_ms(), g_idle_w, mhz, n); printf("IDLE socket %.0f W (%d samples)\n", g_idle_w, n); }
  Args a{d_codes, d_sc, d_src, (uint32_t)0x7fffffff, 65536, 65536, d_out, 0, 0};
  const double F64 = 64.0 * 2 * 32 * 32 * 64 * 4;   // flop per workgroup per iteration: 4 waves x 64 MFMAs
  if (!only_tile8) {
    for (int rep = 0; rep < 2; ++rep) {
      run<0>(smi, "MFMA 128x128 wave tile, GEMM order (k-slice, m, n)", a, sec, F64, 0);
      run<1>(smi, "MFMA serpentine (n reversed on odd m: one operand changes)", a, sec, F64, 0);
      run<2>(smi, "MFMA operand roles swapped (A fragment first)", a, sec, F64, 0);
      run<3>(smi, "MFMA accumulator-stationary (4 k-slices per tile in a row)", a, sec, F64, 0);
      Args u = a; u.unit_scales = 1;
      run<0>(smi, "MFMA GEMM order, every scale byte 127", u, sec, F64, 0);
    }
  }
  for (int rep = 0; rep < 2; ++rep) run<10>(smi, "MFMA 128x64 wave tile (accumulator file by build)", a, sec, F64 / 2, 0);
  if (!only_tile8) {
    run<20>(smi, "LDS fragment reads: 32 x ds_read_b128 per wave", a, sec, 0, 4.0 * 32 * 1024);
    run<21>(smi, "LDS fragment reads: 64 x ds_read_b64 per wave", a, sec, 0, 4.0 * 32 * 1024);
    struct R { const char* name; uint32_t region, stride; } rs[] = {
        {"LDS-DMA 64 KiB/iter, source 16 KiB per CU (vector L1 hits)", 16384, 65536},
        {"LDS-DMA 64 KiB/iter, source 64 KiB per CU (L2 hits: 2 MiB per XCD)", 65536, 65536},
        {"LDS-DMA 64 KiB/iter, source 512 KiB per CU (128 MiB: memory-side cache)", 524288, 524288},
        {"LDS-DMA 64 KiB/iter, source 4 MiB per CU (1 GiB: HBM)", 4194304, 4194304},
        {"LDS-DMA 64 KiB/iter, ALL CUs the same 64 KiB (L2 hits, one line for all)", 65536, 0}};
    for (auto& r : rs) {
      Args d = a; d.region = r.region; d.region_stride = r.stride;
      run<30>(smi, r.name, d, sec, 0, 65536.0);
    }
  }
  smi.finish();
  return 0;
}
