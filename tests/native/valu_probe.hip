// Issue cost of the VALU instructions the streaming quantisers are made of (gfx950): cycles per wave-instruction with 1, 2 and 4
// waves per SIMD, measured with s_memtime around an unrolled run of independent instructions (8 chains).  Decides what is worth
// removing from bwd_quant_t_kernel / fused_quantize_kernel, which are VALU-issue-bound.
//   hipcc --offload-arch=gfx950 -O3 -o valu_probe valu_probe.hip && ./valu_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP ERROR %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void probe(uint32_t* outv, uint64_t* cyc, int iters, uint32_t seed, float fs) {
  float f[16];
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) f[i] = 1.0f + 0.001f * (float)((threadIdx.x + i) & 31);
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = seed * (threadIdx.x + 1) * (i + 3);
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) f[i] = f[i] * fs;                                                                     // v_mul_f32
        if (OP == 1) { v2f t = v2f{f[2 * i], f[2 * i + 1]} * v2f{fs, fs}; f[2 * i] = t[0]; f[2 * i + 1] = t[1]; }   // v_pk_mul_f32
        if (OP == 2) f[i] = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(f[i]), __builtin_fabsf(f[i + 8])), fs);   // v_max3_f32 with |.|
        if (OP == 3) w[i] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w[i], f[i], f[i + 8], 1.0f, 1);           // f32 x2 -> fp4 x2
        if (OP == 4) w[i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w[(i + 1) & 7], fs, 1));   // fp4 x2 -> bf16 x2
        if (OP == 5) w[i] = w[i] * seed;                                                                   // v_mul_lo_u32
        if (OP == 6) f[i] = __builtin_ldexpf(f[i], (int)seed);                                             // v_ldexp_f32
        if (OP == 7) f[i] = __builtin_amdgcn_rcpf(f[i]);                                                   // v_rcp_f32
        if (OP == 8) w[i] = __builtin_amdgcn_perm(w[i], w[(i + 1) & 7], 0x05010400u + r);                  // v_perm_b32
        if (OP == 9) { auto s = __builtin_amdgcn_permlane32_swap(w[i], w[(i + 1) & 7], false, false); w[i] = s[0] ^ s[1]; }   // v_permlane32_swap (+ xor)
        if (OP == 10) f[i] = __builtin_fmaf(f[i], fs, f[i + 8]);                                           // v_fma_f32
        if (OP == 11) { v2f t = __builtin_elementwise_fma(v2f{f[2 * i], f[2 * i + 1]}, v2f{fs, fs}, v2f{f[i], f[i]}); f[2 * i] = t[0]; f[2 * i + 1] = t[1]; }   // v_pk_fma_f32
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(w[i]), "+v"(f[i]), "+v"(f[i + 8]));
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += w[i] + __builtin_bit_cast(uint32_t, f[i]) + __builtin_bit_cast(uint32_t, f[i + 8]);
  outv[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
static int run(const char* name, int per_group) {
  uint32_t* o; uint64_t* c;
  HIP_OK(hipMalloc(&o, 256 * 4 * 256 * 4)); HIP_OK(hipMalloc(&c, 8));
  const int iters = 20000;
  for (int wps : {1, 2, 4}) {
    probe<OP><<<256 * wps, 256>>>(o, c, iters, 12345u, 1.0009765625f);
    HIP_OK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, 0));
    probe<OP><<<256 * wps, 256>>>(o, c, iters, 12345u, 1.0009765625f);
    HIP_OK(hipEventRecord(e1, 0));
    HIP_OK(hipEventSynchronize(e1));
    float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    uint64_t cy; HIP_OK(hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost));
    // s_memtime ticks at 100 MHz; the event time per group / wave-instruction is the robust figure
    printf("VALU %-44s waves/SIMD=%d : %7.2f ns per %d instructions per wave = %6.3f ns per SIMD-instruction (ticks/group %.2f)\n", name, wps, ms * 1e6 / iters,
           per_group, ms * 1e6 / iters / per_group / wps, (double)cy / iters);
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  hipFree(o); hipFree(c);
  return 0;
}

int main() {
  run<0>("v_mul_f32", 32);
  run<1>("v_pk_mul_f32 (2 floats each)", 32);
  run<10>("v_fma_f32", 32);
  run<11>("v_pk_fma_f32 (2 floats each)", 32);
  run<2>("v_max3_f32 |a| |b| c", 32);
  run<3>("v_cvt_scalef32_pk_fp4_f32", 32);
  run<4>("v_cvt_scalef32_pk_bf16_fp4", 32);
  run<5>("v_mul_lo_u32", 32);
  run<6>("v_ldexp_f32", 32);
  run<7>("v_rcp_f32", 32);
  run<8>("v_perm_b32", 32);
  run<9>("v_permlane32_swap + v_xor", 64);
  return 0;
}
