// What does the f32 scale operand of v_cvt_scalef32_pk_f16_fp4 / _bf16_fp4 do with a non-power-of-two scale?
// (hipcc --offload-arch=gfx950 -O2 tests/native/cvt_probe.hip -o tests/native/cvt_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
__global__ void k(const float* sc, float* out) {
  const float s = sc[threadIdx.x];
  const uint32_t w = 0x00000072u;   // byte 0 = codes {2 (1.0) low nibble, 7 (6.0) high nibble}
  const h2_t r = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w, s, 0);
  out[2 * threadIdx.x] = (float)r[0];
  out[2 * threadIdx.x + 1] = (float)r[1];
}
int main() {
  const float hs[8] = {1.0f, 2.0f, 1.5f, 3.0f, 0.75f, 1.125f, 1.875f, 0.4375f};
  float *d, *o, ho[16];
  (void)hipMalloc(&d, sizeof hs); (void)hipMalloc(&o, sizeof ho);
  (void)hipMemcpy(d, hs, sizeof hs, hipMemcpyHostToDevice);
  k<<<1, 8>>>(d, o);
  (void)hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
  for (int i = 0; i < 8; ++i) printf("scale %-7g : e2m1 1.0 -> %-8g e2m1 6.0 -> %-8g  (true products %g, %g)\n", hs[i], ho[2 * i], ho[2 * i + 1], hs[i], 6 * hs[i]);
  return 0;
}
