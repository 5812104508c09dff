// What a transposing fp4 -> fp4 pass (backward_qt_bf16, mxfp4_transpose_mxfp8's input side) can get from HBM as a function of the TILE SHAPE alone: no arithmetic, the
// bytes of an [NT n][MT m] tile of the (N, M/2)-byte input go through LDS and leave as the [MT m][NT n] tile of the (M, N/2)-byte output (same byte count, arbitrary
// content), the e8m0 bytes likewise ((N, M/32) in, (M, N/32) out).  One workgroup per tile, single-buffered (load everything, barrier, store everything); how many
// workgroups a CU holds follows from the tile's LDS.  Cold: 8 buffer sets in rotation (0.57 GB at 8192^2).
//   hipcc --offload-arch=gfx950 -O3 tests/native/xpose_traffic_ubench.hip -o tests/native/xpose_traffic_ubench && tests/native/xpose_traffic_ubench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

struct P {
  const uint8_t* q; const uint8_t* s; uint8_t* oq; uint8_t* os;
  int N, M;   // elements
};

template <int B> struct Piece;
template <> struct Piece<2> { using T = uint16_t; };
template <> struct Piece<4> { using T = uint32_t; };
template <> struct Piece<8> { using T = uint2; };
template <> struct Piece<16> { using T = uint4; };
template <> struct Piece<32> { using T = uint4; };   // two of them

// MFAST: consecutive workgroups walk m (the input row's direction) first; else n first (the output row's direction).
template <int NT, int MT, int THREADS, bool MFAST>
__global__ __launch_bounds__(THREADS) void xpose(const P p) {
  constexpr int TB = NT * MT / 2;            // tile bytes of e2m1
  constexpr int CH = TB / 16;                // 16-byte chunks
  constexpr int PER = CH / THREADS;          // chunks per thread
  static_assert(CH % THREADS == 0, "tile");
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint4* l4 = reinterpret_cast<uint4*>(lds);
  uint8_t* ls = lds + TB;                    // NT * MT / 32 scale bytes
  const int tid = threadIdx.x;
  const int tm = p.M / MT, tn = p.N / NT;
  const int t = blockIdx.x;
  const int n0 = (MFAST ? t / tm : t % tn) * NT, m0 = (MFAST ? t % tm : t / tn) * MT;
  const size_t irow = p.M / 2, orow = p.N / 2;
  uint4 v[PER];
  constexpr int IC = MT / 32;                // chunks per input row piece
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = tid + j * THREADS, r = i / IC, c = i % IC;
    v[j] = *reinterpret_cast<const uint4*>(p.q + (size_t)(n0 + r) * irow + m0 / 2 + c * 16);
  }
  // scales in: NT rows x MT / 32 bytes
  constexpr int SI = MT / 32, SO = NT / 32;
  using TI = typename Piece<(SI > 16 ? 16 : SI)>::T;
  using TO = typename Piece<(SO > 16 ? 16 : SO)>::T;
  constexpr int SIP = SI > 16 ? SI / 16 : 1, SOP = SO > 16 ? SO / 16 : 1;
  for (int i = tid; i < NT * SIP; i += THREADS) {
    const int r = i / SIP, c = i % SIP;
    *reinterpret_cast<TI*>(ls + (size_t)i * sizeof(TI)) = *reinterpret_cast<const TI*>(p.s + (size_t)(n0 + r) * (p.M / 32) + m0 / 32 + c * sizeof(TI));
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) l4[tid + j * THREADS] = v[j];
  __syncthreads();
  constexpr int OC = NT / 32;                // chunks per output row piece
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = tid + j * THREADS, r = i / OC, c = i % OC;
    *reinterpret_cast<uint4*>(p.oq + (size_t)(m0 + r) * orow + n0 / 2 + c * 16) = l4[i ^ 1];
  }
  for (int i = tid; i < MT * SOP; i += THREADS) {
    const int r = i / SOP, c = i % SOP;
    *reinterpret_cast<TO*>(p.os + (size_t)(m0 + r) * (p.N / 32) + n0 / 32 + c * sizeof(TO)) = *reinterpret_cast<const TO*>(ls + (size_t)i * sizeof(TO));
  }
}

__global__ __launch_bounds__(256) void plain_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

struct Set { uint8_t *q, *s, *oq, *os; };

template <int NT, int MT, int THREADS, bool MFAST>
static float run(const std::vector<Set>& sets, int N, int M, int rounds) {
  constexpr int LDSB = NT * MT / 2 + NT * MT / 32;
  auto kern = xpose<NT, MT, THREADS, MFAST>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
  const int grid = (N / NT) * (M / MT);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int r = 0; r < rounds; ++r)
      for (const Set& s : sets) { P p{s.q, s.s, s.oq, s.os, N, M}; hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), LDSB, 0, p); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms * 1000.f / (rounds * sets.size()));
  }
  return best;
}

int main(int argc, char** argv) {
  const int sizes[][2] = {{8192, 8192}, {4096, 4096}, {2048, 16384}};
  for (auto& sz : sizes) {
    const int N = sz[0], M = sz[1];
    const size_t qb = (size_t)N * M / 2, sb = (size_t)N * M / 32;
    const int nset = std::max<int>(2, (int)(600e6 / (2 * (qb + sb))) + 1);
    std::vector<Set> sets(nset);
    for (auto& s : sets) {
      CK(hipMalloc(&s.q, qb)); CK(hipMalloc(&s.s, sb)); CK(hipMalloc(&s.oq, qb)); CK(hipMalloc(&s.os, sb));
      CK(hipMemset(s.q, 0x35, qb)); CK(hipMemset(s.s, 0x7f, sb));
    }
    CK(hipDeviceSynchronize());
    const double bytes = 2.0 * (qb + sb);
    const int rounds = 3;
    printf("N = %d, M = %d: %.1f MB per pass, %d buffer sets (cold)\n", N, M, bytes / 1e6, nset);
    {   // calibration: the same bytes as a plain copy
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int r = 0; r < rounds; ++r) for (auto& s : sets) hipLaunchKernelGGL(plain_copy, dim3(256 * 8), dim3(256), 0, 0, (const uint4*)s.q, (uint4*)s.oq, qb / 16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms * 1000.f / (rounds * nset));
      }
      printf("  %-44s %8.2f us  %6.2f TB/s\n", "plain copy of the e2m1 bytes (2048 x 256)", best, 2.0 * qb / best / 1e6);
    }
#define RUN(NT, MT, TH, MF) do { float us = run<NT, MT, TH, MF>(sets, N, M, rounds); \
    printf("  tile [%4d n][%4d m] %4d thr %s  in %4d B runs, out %4d B runs, LDS %6d: %8.2f us  %6.2f TB/s  (%.3f of 8)\n", NT, MT, TH, MF ? "m-fast" : "n-fast", MT / 2, NT / 2, \
           NT * MT / 2 + NT * MT / 32, us, bytes / us / 1e6, bytes / us / 8e6); } while (0)
    RUN(128, 64, 256, true);     // about the product's unit today (4 groups x 64 rows per wave): 32-byte input pieces, 64-byte output pieces
    RUN(256, 64, 256, true);
    RUN(256, 256, 256, true);
    RUN(256, 256, 512, true);
    RUN(256, 256, 512, false);
    RUN(512, 256, 512, true);
    RUN(512, 256, 512, false);
    RUN(256, 512, 512, true);
    RUN(256, 512, 512, false);
    RUN(512, 512, 1024, true);
    RUN(512, 512, 1024, false);
    RUN(1024, 256, 1024, true);
    RUN(1024, 256, 1024, false);
    RUN(1024, 128, 512, true);
    RUN(1024, 128, 512, false);
    RUN(512, 128, 256, true);
    RUN(512, 128, 256, false);
    for (auto& s : sets) { hipFree(s.q); hipFree(s.s); hipFree(s.oq); hipFree(s.os); }
  }
  return 0;
}
