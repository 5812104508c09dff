// Does the scale operand of v_cvt_scalef32_pk_fp4_f32 replace an explicit power-of-two scaling bit for bit?
//   A = cvt(ldexp(y, sh), 1.0)      what the quantisers do today (Quest path: v_ldexp_f32 per element, then the convert)
//   B = cvt(y, 2^-sh)               hypothesis 1: the instruction DIVIDES by its scale operand
//   C = cvt(y, 2^sh)                hypothesis 2: it multiplies
// over random values, exact rounding ties of the e2m1 grid, saturating values, zeros, denormal products.  Prints mismatch counts.
//   hipcc --offload-arch=gfx950 -O3 -o cvt_scale_probe cvt_scale_probe.hip && ./cvt_scale_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

__global__ void probe(const float* y, const int* sh, uint32_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = y[2 * i], b = y[2 * i + 1];
  const int s = sh[i];
  const uint32_t A = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(0u, __builtin_ldexpf(a, s), __builtin_ldexpf(b, s), 1.0f, 0) & 0xffu;
  const uint32_t B = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(0u, a, b, __builtin_ldexpf(1.0f, -s), 0) & 0xffu;
  const uint32_t C = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(0u, a, b, __builtin_ldexpf(1.0f, s), 0) & 0xffu;
  out[i] = A | (B << 8) | (C << 16);
}

int main() {
  std::mt19937 rng(1234);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::uniform_int_distribution<int> shd(-40, 40);
  std::vector<float> y;
  std::vector<int> sh;
  const float grid[] = {0.f, 0.25f, 0.5f, 0.75f, 1.f, 1.25f, 1.5f, 1.75f, 2.f, 2.5f, 3.f, 3.5f, 4.f, 5.f, 6.f, 7.f, 8.f, 100.f};
  for (int rep = 0; rep < 200000; ++rep) {
    const int s = shd(rng);
    float a, b;
    const int kind = rep % 4;
    if (kind == 0) { a = nd(rng) * 3.f; b = nd(rng) * 3.f; }
    else if (kind == 1) { a = grid[rng() % 18] * ((rng() & 1) ? -1.f : 1.f); b = grid[rng() % 18] * ((rng() & 1) ? -1.f : 1.f); }   // exact ties / grid points
    else if (kind == 2) { a = std::nextafterf(grid[rng() % 18], (rng() & 1) ? 100.f : -100.f); b = nd(rng) * 1e-3f; }                 // one ulp beside a tie
    else { a = nd(rng) * 8.f; b = (rng() & 1) ? 0.f : -0.f; }
    y.push_back(std::ldexp(a, -s)); y.push_back(std::ldexp(b, -s));   // so that ldexp(y, s) lands on the interesting values
    sh.push_back(s);
  }
  // products in the fp32 denormal range: y tiny, sh very negative
  for (int rep = 0; rep < 20000; ++rep) { y.push_back(nd(rng) * 1e-30f); y.push_back(nd(rng) * 1e-35f); sh.push_back(-(int)(rng() % 40) - 10); }
  const int n = (int)sh.size();
  float* dy; int* ds; uint32_t* dout;
  hipMalloc(&dy, n * 8); hipMalloc(&ds, n * 4); hipMalloc(&dout, n * 4);
  hipMemcpy(dy, y.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(ds, sh.data(), n * 4, hipMemcpyHostToDevice);
  probe<<<(n + 255) / 256, 256>>>(dy, ds, dout, n);
  std::vector<uint32_t> out(n);
  hipMemcpy(out.data(), dout, n * 4, hipMemcpyDeviceToHost);
  int mb = 0, mc = 0, shown = 0;
  for (int i = 0; i < n; ++i) {
    const uint32_t A = out[i] & 0xff, B = (out[i] >> 8) & 0xff, C = (out[i] >> 16) & 0xff;
    mb += A != B; mc += A != C;
    if (A != B && shown < 10) { printf("  A != B at %d: y = (%g, %g) sh = %d  A = %02x B = %02x C = %02x\n", i, y[2 * i], y[2 * i + 1], sh[i], A, B, C); ++shown; }
  }
  printf("CVT-SCALE pairs %d : cvt(ldexp(y, sh), 1) vs cvt(y, 2^-sh): %d mismatches ; vs cvt(y, 2^sh): %d mismatches\n", n, mb, mc);
  return 0;
}
