// Does the scale operand of the gfx950 fp4 converts honour the MANTISSA of its f32 scale, or only the exponent (e8m0 semantics)?
// If cvt_scalef32_pk_f16_fp4(code, 1.5) returned 1.5 * value, the NVFP4 GEMM's dequantisation (convert + packed multiply by the e4m3 group scale)
// would be one instruction instead of two.   hipcc --offload-arch=gfx950 -O3 -o cvt_mant_probe cvt_mant_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
__global__ void probe(float* out) {
  const uint32_t w = 0x00000053u;   // byte 0: nibbles 3 (1.5) and 5 (3.0)
  const float scales[6] = {1.0f, 1.5f, 1.75f, 3.0f, 0.8125f, 2.0f};
  for (int i = 0; i < 6; ++i) {
    const h2 a = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, scales[i], 0);
    const b2 b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scales[i], 0);
    out[i * 6 + 0] = scales[i]; out[i * 6 + 1] = (float)a[0]; out[i * 6 + 2] = (float)a[1]; out[i * 6 + 3] = (float)b[0]; out[i * 6 + 4] = (float)b[1];
    out[i * 6 + 5] = __builtin_bit_cast(float, (uint32_t)__builtin_amdgcn_cvt_scalef32_pk_fp4_f32(0u, 3.0f, 4.5f, scales[i], 0));
  }
}
int main() {
  float* d; hipMalloc(&d, 36 * 4);
  probe<<<1, 1>>>(d);
  float h[36]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int i = 0; i < 6; ++i) {
    uint32_t bits; memcpy(&bits, &h[i * 6 + 5], 4);
    printf("scale %-7g : fp4(1.5, 3.0) -> f16 (%g, %g)  bf16 (%g, %g) ; f32 (3.0, 4.5) -> fp4 byte %02x\n", h[i * 6], h[i * 6 + 1], h[i * 6 + 2], h[i * 6 + 3], h[i * 6 + 4], bits & 0xff);
  }
  return 0;
}
