/*
 * qutlass_oracle.c -- CPU restatement of the qutlass hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the MI355X build.  It is NOT part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product path (qutlass_amd/) never imports, links or calls anything in oracle/.
 *
 * Every function restates, in plain scalar C (fp32 arithmetic exactly where the reference
 * kernel uses fp32, fmaf where nvcc contracts, double where the reference expression is
 * evaluated in double), the algorithm of the reference file:line it cites.  Paths are
 * relative to the reference checkout (IST-DASLab/qutlass v0.2.0).
 *
 * Pinning: oracle is checked against golden vectors produced by the reference's own
 * Python test oracles (tests/mxfp4_test.py, tests/nvfp4_test.py, tests/mxfp8_test.py,
 * qutlass/utils.py::to_blocked) -- see tests/golden/make_golden.py and
 * tests/test_oracle_golden.py.
 *
 * Build: make -C oracle      (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* scalar format helpers                                                                        */
/* ------------------------------------------------------------------------------------------ */

static inline float bf16_to_f32(uint16_t v) {
  uint32_t u = (uint32_t)v << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* fp32 -> bf16, round-to-nearest-even (what the GEMM epilogue's bf16 store does). */
static inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40); /* quiet NaN */
  uint32_t lsb = (u >> 16) & 1u;
  u += 0x7fffu + lsb;
  return (uint16_t)(u >> 16);
}

/* e2m1 code -> value.  Table: tests/mxfp4_test.py:92-113 (grid_dq), bit 3 = sign. */
static const float E2M1_VALUES[8] = {0.0f, 0.5f, 1.0f, 1.5f, 2.0f, 3.0f, 4.0f, 6.0f};

float orc_e2m1_decode(uint8_t code) {
  float v = E2M1_VALUES[code & 7];
  return (code & 8) ? -v : v;
}

/*
 * fp32 -> e2m1 code with the semantics of PTX `cvt.rn.satfinite.e2m1x2.f32`
 * (qutlass/csrc/include/cutlass_extensions/epilogue/threadblock/epilogue_quant.h:77-97):
 * round-to-nearest-even onto {0,.5,1,1.5,2,3,4,6}, saturate to +-6 (also +-inf),
 * sign of zero preserved, NaN -> +6 (0x7).
 * Ties (even code wins): .25->0  .75->1  1.25->1  1.75->2  2.5->2  3.5->4  5->4
 * (KATs in tests/mxfp4_test.py:45-81 `_rtne_fp4`; that Python helper encodes an exact +0.0
 *  as 0x8 because of torch.bucketize -- an artefact; the kernel/PTX semantics give 0x0.)
 */
uint8_t orc_e2m1_encode(float x) {
  if (x != x) return 0x7;
  uint8_t sign = signbit(x) ? 8 : 0;
  float a = fabsf(x);
  uint8_t c = 0;
  c += (a > 0.25f);
  c += (a >= 0.75f);
  c += (a > 1.25f);
  c += (a >= 1.75f);
  c += (a > 2.5f);
  c += (a >= 3.5f);
  c += (a > 5.0f);
  return sign | c;
}

/* e8m0 byte -> 2^(b-127) as double (0xFF is NaN in the format; callers never produce it). */
static inline double e8m0_to_f64(uint8_t b) { return (b == 0xFF) ? NAN : ldexp(1.0, (int)b - 127); }

/* OCP e4m3fn decode. */
float orc_e4m3_decode(uint8_t b) {
  int s = b >> 7, e = (b >> 3) & 0xF, m = b & 7;
  float v;
  if (e == 0xF && m == 7) v = NAN;
  else if (e == 0) v = ldexpf((float)m, -9);
  else v = ldexpf(1.0f + (float)m / 8.0f, e - 7);
  return s ? -v : v;
}

/*
 * fp32 -> OCP e4m3fn, RNE, saturate-to-finite (+-448), NaN -> 0x7F: the semantics of
 * `cvt.rn.satfinite.e4m3x2.f32` / `__nv_fp8_e4m3(float)` used at epilogue_quant.h:99-109
 * and :1631-1680.
 */
uint8_t orc_e4m3_encode(float x) {
  if (x != x) return 0x7F;
  uint8_t sign = signbit(x) ? 0x80 : 0;
  float a = fabsf(x);
  if (a >= 448.0f) return sign | 0x7E; /* satfinite (also 464 tie -> 448) */
  if (a < ldexpf(1.0f, -10)) {         /* below half the smallest subnormal (2^-9): ties-to-even -> 0 */
    return sign;                         /* a == 2^-10 exactly is a tie between 0 and 2^-9 -> even (0)  */
  }
  int e;
  (void)frexpf(a, &e); /* a = f * 2^e, f in [0.5,1) -> unbiased exponent e-1 */
  int ue = e - 1;
  if (ue < -6) ue = -6;                 /* subnormal range shares the quantum 2^-9 */
  float q = ldexpf(1.0f, ue - 3);       /* quantum = 2^(ue-3) */
  float r = nearbyintf(a / q);          /* RNE (default rounding mode); a/q exact (power of two) */
  float v = r * q;
  /* re-encode v exactly */
  if (v >= 448.0f) return sign | 0x7E;
  if (v < ldexpf(1.0f, -6)) return sign | (uint8_t)(int)(v * 512.0f);
  (void)frexpf(v, &e);
  ue = e - 1;
  int m = (int)(ldexpf(v, -ue) * 8.0f) - 8;
  return sign | (uint8_t)(((ue + 7) << 3) | m);
}

/*
 * OCP e5m2 (IEEE-like: 5 exponent bits, bias 15, 2 mantissa bits, max finite 57344, inf 0x7C, NaN 0x7D..0x7F).
 * EXTENSION: the reference's MXFP8 GEMMs reject every element type but e4m3 (bindings.cpp:157-160, 196-199;
 * gemm.cu:339-345, 399-403).  BASELINE.json configs[4] names an e5m2-gradient x e4m3-activation leg, which CDNA4's
 * scaled MFMA supports natively (cbsz / blgp select the format per operand), so the oracle restates the format here.
 * The conversion is pinned by tests/golden/gemm_mxfp8_e5m2.npz, generated with torch's own float8_e5m2 cast
 * (tests/golden/make_golden.py), NOT by anything in the reference.
 */
float orc_e5m2_decode(uint8_t b) {
  int s = b >> 7, e = (b >> 2) & 0x1F, m = b & 3;
  float v;
  if (e == 0x1F) v = m ? NAN : INFINITY;
  else if (e == 0) v = ldexpf((float)m, -16);
  else v = ldexpf(1.0f + (float)m / 4.0f, e - 15);
  return s ? -v : v;
}

/* fp32 -> e5m2, RNE, saturate-to-finite (+-57344), NaN -> 0x7F (the satfinite convention used for e4m3 above). */
uint8_t orc_e5m2_encode(float x) {
  if (x != x) return 0x7F;
  uint8_t sign = signbit(x) ? 0x80 : 0;
  float a = fabsf(x);
  if (a >= 57344.0f) return sign | 0x7B;
  if (a <= ldexpf(1.0f, -17)) return sign;   /* at most half the smallest subnormal (2^-16): tie -> even (0) */
  int e;
  (void)frexpf(a, &e);
  int ue = e - 1;
  if (ue < -14) ue = -14;
  float q = ldexpf(1.0f, ue - 2);
  float v = nearbyintf(a / q) * q;
  if (v >= 57344.0f) return sign | 0x7B;
  if (v < ldexpf(1.0f, -14)) return sign | (uint8_t)(int)(v * 65536.0f);
  (void)frexpf(v, &e);
  ue = e - 1;
  int m = (int)(ldexpf(v, -ue) * 4.0f) - 4;
  return sign | (uint8_t)(((ue + 15) << 2) | m);
}

/* ------------------------------------------------------------------------------------------ */
/* a8: to_blocked  (qutlass/utils.py:160-193 torch path; :16-133 Triton path = same map + 0-pad) */
/* ------------------------------------------------------------------------------------------ */
/*
 * in : (rows, cols) row-major bytes.  out: ceil(rows/128)*128 * ceil(cols/4)*4 bytes,
 * out[(rb*CB + cb)*512 + (r%32)*16 + ((r%128)/32)*4 + c%4] = in[r][c], zero where r>=rows or
 * c>=cols (Triton path, utils.py:52-56 `other=0.0`).
 */
void orc_to_blocked(const uint8_t* in, int64_t rows, int64_t cols, uint8_t* out) {
  int64_t RB = (rows + 127) / 128, CB = (cols + 3) / 4;
  memset(out, 0, (size_t)(RB * CB * 512));
  for (int64_t r = 0; r < rows; ++r) {
    int64_t rb = r / 128, rr = r % 128;
    for (int64_t c = 0; c < cols; ++c) {
      int64_t cb = c / 4;
      out[(rb * CB + cb) * 512 + (rr % 32) * 16 + (rr / 32) * 4 + (c % 4)] = in[r * cols + c];
    }
  }
}

static inline uint8_t blocked_sf(const uint8_t* sf, int64_t CB, int64_t r, int64_t kb) {
  int64_t rb = r / 128, rr = r % 128, cb = kb / 4;
  return sf[(rb * CB + cb) * 512 + (rr % 32) * 16 + (rr / 32) * 4 + (kb % 4)];
}

/* ------------------------------------------------------------------------------------------ */
/* a4/a5/a6/a7: fusedQuantizeMx                                                                 */
/* ------------------------------------------------------------------------------------------ */
/*
 * Rotation y[g*R + j] = sum_k x[g*R + k] * h[k*R + j]   (A row-major ld=K, B row-major ld=N:
 * fused_quantize_mx.cu:70-102,124-139  =>  Y = X_g . h, not h^T), bf16 x bf16 -> fp32.
 * Products of two bf16 are exact in fp32; the accumulation order of the tensor-core
 * instruction is not architecturally defined on either vendor, so the oracle fixes ONE
 * order: a k-ascending fmaf chain.  (acc_model 1 = exact sum rounded once to fp32, used by
 * the tests to bound order effects.)
 */
static void rotate_group(const uint16_t* x, const float* hf, int R, int acc_model, float* y) {
  if (acc_model == 1) {
    for (int j = 0; j < R; ++j) {
      double acc = 0.0; /* 16-bit products, <=128 terms: exact in double for sane ranges */
      for (int k = 0; k < R; ++k) acc += (double)bf16_to_f32(x[k]) * (double)hf[k * R + j];
      y[j] = (float)acc;
    }
    return;
  }
  for (int j = 0; j < R; ++j) {
    float acc = 0.0f;
    for (int k = 0; k < R; ++k) acc = fmaf(bf16_to_f32(x[k]), hf[k * R + j], acc);
    y[j] = acc;
  }
}

/* pack 32 fp32 -> 16 bytes; element 2j -> low nibble, 2j+1 -> high nibble of byte j
 * (epilogue_quant.h:87-90; tests/mxfp4_test.py:80). */
static void pack32(const float* q, uint8_t* out16) {
  for (int j = 0; j < 16; ++j)
    out16[j] = (uint8_t)(orc_e2m1_encode(q[2 * j]) | (orc_e2m1_encode(q[2 * j + 1]) << 4));
}

/*
 * method: 0 = quest, 1 = abs_max.  mask (nullable): one u32 per 32-group, bit i set iff
 * |y_i / s| < 6 (epilogue_quant.h:1180-1196), little-endian bytes = (.., K/8) u8 view.
 *
 * abs_max (epilogue_quant.h:546-571): s = amax|y| + 1e-8f ; s &= 0x7f800000 ; e8m0 = bits>>23 ;
 *                                      q = (y / s) * 3 ; e2m1(q)
 * quest   (epilogue_quant.h:520-545): sum1 += y ; sum2 = fma(y,y,sum2)   (sequential i=0..31)
 *                                      mean = sum1/32 ; var = sum2/32 - mean*mean
 *                                      s = var>=0 ? (float)( (double)sqrtf(var)*(2.92247856/6.) + 1e-8 ) : 1
 *                                      s &= 0x7f800000 ; q = y / s
 * Division by s (a power of two) is exact, so fast-math `/` == IEEE `/` here.
 * nvcc contracts `sum2 += v*v` to fma and `x - mean*mean` to fma(-mean, mean, x).
 */
void orc_fused_quantize_mx(const uint16_t* x, const uint16_t* h, int R, int64_t numel, int method,
                           int acc_model, uint8_t* out_e2m1, uint8_t* out_e8m0, uint32_t* out_mask) {
  float* hf = (float*)malloc(sizeof(float) * R * R);
  for (int i = 0; i < R * R; ++i) hf[i] = bf16_to_f32(h[i]);
  int64_t nrot = numel / R;
#pragma omp parallel for schedule(static)
  for (int64_t g = 0; g < nrot; ++g) {
    float y[128];
    rotate_group(x + g * R, hf, R, acc_model, y);
    for (int sub = 0; sub < R / 32; ++sub) {
      const float* v = y + sub * 32;
      int64_t grp = g * (R / 32) + sub;
      float scale;
      if (method == 0) {
        float s1 = 0.f, s2 = 0.f;
        if (acc_model == 2) {
          /* the HIP kernel's order (qutlass_amd/csrc/quantize.hip.h: lane half h holds j = 8q + 4h + e, q, e = 0..3, summed in
           * register order, then ONE cross-lane add): same terms, different fp32 association than the reference's sequential
           * loop -- tests/test_gpu_round3.py uses it to show that a scale byte which differs from the sequential sum differs
           * by the summation order alone */
          float p1[2] = {0.f, 0.f}, p2[2] = {0.f, 0.f};
          for (int hh = 0; hh < 2; ++hh)
            for (int qq = 0; qq < 4; ++qq)
              for (int e = 0; e < 4; ++e) {
                const float t = v[8 * qq + 4 * hh + e];
                p1[hh] = p1[hh] + t;
                p2[hh] = fmaf(t, t, p2[hh]);
              }
          s1 = p1[0] + p1[1];
          s2 = p2[0] + p2[1];
        } else
        for (int i = 0; i < 32; ++i) {
          s1 = s1 + v[i];
          s2 = fmaf(v[i], v[i], s2);
        }
        float mean = s1 / 32;
        float var = fmaf(-mean, mean, s2 / 32);
        scale = 1.0f;
        if (var >= 0) scale = (float)((double)sqrtf(var) * (2.92247856 / 6.) + 1e-8);
      } else {
        float amax = 0.f;
        for (int i = 0; i < 32; ++i) {
          float a = fabsf(v[i]);
          if (a > amax) amax = a;
        }
        scale = amax + 1e-8f;
      }
      uint32_t sb;
      memcpy(&sb, &scale, 4);
      sb &= 0x7f800000u;
      memcpy(&scale, &sb, 4);
      out_e8m0[grp] = (uint8_t)(sb >> 23);
      float q[32];
      uint32_t m = 0;
      for (int i = 0; i < 32; ++i) {
        float t = v[i] / scale;
        if (fabsf(t) < 6.f) m |= (1u << i);
        if (method == 1) t = t * 3;
        q[i] = t;
      }
      pack32(q, out_e2m1 + grp * 16);
      if (out_mask) out_mask[grp] = m;
    }
  }
  free(hf);
}

/* ------------------------------------------------------------------------------------------ */
/* N1: fusedQuantizeNv  (epilogue_quant.h:1560-1700, op_16; op_32/64/128 identical per 16-group) */
/* ------------------------------------------------------------------------------------------ */
/*
 * abs_max: SF = e4m3_rn_sat( gs * (amax * rcp(6)) ) ; out = SF!=0 ? rcp(SF * rcp(gs)) : 0 ; q = y*out
 * quest  : SF = e4m3( sqrt(sum2*rcp(16) - mean^2) * (2.92247856/6.) + 1e-8 ), mean = sum1*rcp(16)
 *          out = SF>0 ? rcp(SF) : 0 ; q = y*out
 * `rcp` is rcp.approx.ftz.f32 (<=1 ulp) in the reference; the oracle uses the correctly rounded
 * 1.0f/x, so NVFP4 code parity is a tolerance check (the reference's own bound is 1e-1 mismatch
 * fraction, tests/nvfp4_test.py:204-205), scales (e4m3 bytes) are compared exactly up to that rcp.
 */
void orc_fused_quantize_nv(const uint16_t* x, const uint16_t* h, int R, int64_t numel, int method,
                           int acc_model, float global_scale, uint8_t* out_e2m1, uint8_t* out_e4m3) {
  float* hf = (float*)malloc(sizeof(float) * R * R);
  for (int i = 0; i < R * R; ++i) hf[i] = bf16_to_f32(h[i]);
  int64_t nrot = numel / R;
#pragma omp parallel for schedule(static)
  for (int64_t g = 0; g < nrot; ++g) {
    float y[128];
    rotate_group(x + g * R, hf, R, acc_model, y);
    for (int sub = 0; sub < R / 16; ++sub) {
      const float* v = y + sub * 16;
      int64_t grp = g * (R / 16) + sub;
      float out_scale;
      uint8_t sfb;
      if (method == 0) {
        float s1 = 0.f, s2 = 0.f;
        for (int i = 0; i < 16; ++i) {
          s1 = s1 + v[i];
          s2 = fmaf(v[i], v[i], s2);
        }
        float r16 = 1.0f / 16.0f;
        float mean = s1 * r16;
        float var = fmaf(-mean, mean, s2 * r16);
        /* var < 0 (fp32 rounding on a nearly constant group): sqrt gives NaN, the scale byte is 0x7f, `sq > 0` false, the multiplier 0 and
         * every code +-0 -- epilogue_quant.h:1631-1640 has no guard, and neither has this restatement (rounds 1-5 clamped at 0) */
        float scale = (float)((double)sqrtf(var) * (2.92247856 / 6.) + 1e-8);
        sfb = orc_e4m3_encode(scale);
        float sq = orc_e4m3_decode(sfb);
        out_scale = (sq > 0.f) ? 1.0f / sq : 0.0f;
      } else {
        float amax = 0.f;
        for (int i = 0; i < 16; ++i) {
          float a = fabsf(v[i]);
          if (a > amax) amax = a;
        }
        float sf = global_scale * (amax * (1.0f / 6.0f));
        sfb = orc_e4m3_encode(sf);
        sf = orc_e4m3_decode(sfb);
        out_scale = (sf != 0.f) ? 1.0f / (sf * (1.0f / global_scale)) : 0.0f;
      }
      out_e4m3[grp] = sfb;
      uint8_t* o = out_e2m1 + grp * 8;
      for (int j = 0; j < 8; ++j)
        o[j] = (uint8_t)(orc_e2m1_encode(v[2 * j] * out_scale) |
                         (orc_e2m1_encode(v[2 * j + 1] * out_scale) << 4));
    }
  }
  free(hf);
}

/* ------------------------------------------------------------------------------------------ */
/* a12/a13/a14: block-scaled GEMMs  D = bf16_rn(alpha * (A.SFA)(B.SFB)^T)                      */
/* ------------------------------------------------------------------------------------------ */
/*
 * Semantics from qutlass/csrc/gemm.cu:174-434 (type assembly) + the reference's own test oracle
 * tests/mxfp4_test.py:229-237 (dequantise both operands, a_dq @ b_dq.T in fp64, cast to bf16).
 * Scales are read from the to_blocked layout (gemm.cu:113-123 via sm100_blockscaled_layout.hpp:53-104).
 * Accumulation here is in double (every product is exact; the sum is exact for any realistic
 * exponent spread), then -> fp32 -> * alpha (fp32) -> bf16 RNE, which is what an fp32-accumulating
 * kernel produces whenever its partial sums are exact.
 * kind: 0 = MXFP4 (e2m1, e8m0/32)   1 = NVFP4 (e2m1, e4m3/16)   2 = MXFP8 TN (e4m3, e8m0/32)
 *       3 = MXFP8 NN (A stored (K,M) row-major)
 *       4 / 5 = kinds 2 / 3 with an e5m2 A operand (extension, see orc_e5m2_decode; B stays e4m3)
 */
static void dequant_row_fp4(const uint8_t* row, int64_t K, double* out) {
  for (int64_t j = 0; j < K / 2; ++j) {
    out[2 * j] = orc_e2m1_decode(row[j] & 0xF);
    out[2 * j + 1] = orc_e2m1_decode(row[j] >> 4);
  }
}

void orc_gemm_blockscaled(int kind, const uint8_t* A, const uint8_t* B, const uint8_t* SFA,
                          const uint8_t* SFB, float alpha, int64_t M, int64_t N, int64_t K,
                          uint16_t* D) {
  int gs = (kind == 1) ? 16 : 32;
  int64_t KB = K / gs, CB = (KB + 3) / 4;
  double* Ad = (double*)malloc(sizeof(double) * M * K);
  double* Bd = (double*)malloc(sizeof(double) * N * K);
#pragma omp parallel for schedule(static)
  for (int64_t m = 0; m < M; ++m) {
    double* o = Ad + m * K;
    if (kind <= 1) dequant_row_fp4(A + m * (K / 2), K, o);
    else if (kind == 2) for (int64_t k = 0; k < K; ++k) o[k] = orc_e4m3_decode(A[m * K + k]);
    else if (kind == 3) for (int64_t k = 0; k < K; ++k) o[k] = orc_e4m3_decode(A[k * M + m]);
    else if (kind == 4) for (int64_t k = 0; k < K; ++k) o[k] = orc_e5m2_decode(A[m * K + k]);
    else for (int64_t k = 0; k < K; ++k) o[k] = orc_e5m2_decode(A[k * M + m]);
    for (int64_t kb = 0; kb < KB; ++kb) {
      uint8_t sb = blocked_sf(SFA, CB, m, kb);
      double s = (kind == 1) ? (double)orc_e4m3_decode(sb) : e8m0_to_f64(sb);
      for (int i = 0; i < gs; ++i) o[kb * gs + i] *= s;
    }
  }
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < N; ++n) {
    double* o = Bd + n * K;
    if (kind <= 1) dequant_row_fp4(B + n * (K / 2), K, o);
    else for (int64_t k = 0; k < K; ++k) o[k] = orc_e4m3_decode(B[n * K + k]);
    for (int64_t kb = 0; kb < KB; ++kb) {
      uint8_t sb = blocked_sf(SFB, CB, n, kb);
      double s = (kind == 1) ? (double)orc_e4m3_decode(sb) : e8m0_to_f64(sb);
      for (int i = 0; i < gs; ++i) o[kb * gs + i] *= s;
    }
  }
#pragma omp parallel for schedule(static) collapse(2)
  for (int64_t m = 0; m < M; ++m)
    for (int64_t n = 0; n < N; ++n) {
      const double* a = Ad + m * K;
      const double* b = Bd + n * K;
      double acc = 0.0;
      for (int64_t k = 0; k < K; ++k) acc += a[k] * b[k];
      float f = (float)acc;
      D[m * N + n] = f32_to_bf16_rne(f * alpha);
    }
  free(Ad);
  free(Bd);
}

/* ------------------------------------------------------------------------------------------ */
/* MXFP8 pseudo-quantiser used to PRODUCE inputs for a14 (tests/mxfp8_test.py:26-46)            */
/* ------------------------------------------------------------------------------------------ */
/*
 * per 32-group of bf16 x: amax>0 ? e8m0 = floor(bf16(log2(amax))) - 8 + 128 : 128  (uint8 arithmetic)
 *                         q = e4m3_rn( clamp(x / 2^(e8m0-127), +-448) )   (division in bf16 in the
 * reference because `x` is bf16 there: x/2^e is exact in bf16 unless it underflows; the oracle
 * divides in fp32 and rounds the quotient to bf16 first to follow it.)
 */
void orc_pseudoquant_mxfp8(const uint16_t* x, int64_t numel, uint8_t* out_e4m3, uint8_t* out_e8m0) {
#pragma omp parallel for schedule(static)
  for (int64_t g = 0; g < numel / 32; ++g) {
    float amax = 0.f;
    for (int i = 0; i < 32; ++i) {
      float a = fabsf(bf16_to_f32(x[g * 32 + i]));
      if (a > amax) amax = a;
    }
    uint8_t e = 128;
    if (amax > 0) {
      /* torch.log2 on a bf16 tensor: computed in fp32, rounded to bf16, THEN floored -- so an
       * amax just below a power of two (e.g. 63.75) lands on the upper binade.  Follow it. */
      float l = bf16_to_f32(f32_to_bf16_rne(log2f(amax)));
      e = (uint8_t)((uint8_t)(int)floorf(l) - 8 + 128);
    }
    out_e8m0[g] = e;
    float s = (float)e8m0_to_f64(e);
    for (int i = 0; i < 32; ++i) {
      float q = bf16_to_f32(f32_to_bf16_rne(bf16_to_f32(x[g * 32 + i]) / s));
      if (q > 448.f) q = 448.f;
      if (q < -448.f) q = -448.f;
      out_e4m3[g * 32 + i] = orc_e4m3_encode(q);
    }
  }
}

/*
 * The same pseudo-quantiser with an e5m2 payload (extension): the expression of tests/mxfp8_test.py:26-46 with the
 * format constants swapped -- shared exponent floor(log2 amax) - 15 + 128 (e5m2: emax = 15, e4m3: 8), clamp +-57344.
 */
void orc_pseudoquant_mxfp8_e5m2(const uint16_t* x, int64_t numel, uint8_t* out_e5m2, uint8_t* out_e8m0) {
#pragma omp parallel for schedule(static)
  for (int64_t g = 0; g < numel / 32; ++g) {
    float amax = 0.f;
    for (int i = 0; i < 32; ++i) {
      float a = fabsf(bf16_to_f32(x[g * 32 + i]));
      if (a > amax) amax = a;
    }
    uint8_t e = 128;
    if (amax > 0) {
      float l = bf16_to_f32(f32_to_bf16_rne(log2f(amax)));
      e = (uint8_t)((uint8_t)(int)floorf(l) - 15 + 128);
    }
    out_e8m0[g] = e;
    float s = (float)e8m0_to_f64(e);
    for (int i = 0; i < 32; ++i) {
      float q = bf16_to_f32(f32_to_bf16_rne(bf16_to_f32(x[g * 32 + i]) / s));
      if (q > 57344.f) q = 57344.f;
      if (q < -57344.f) q = -57344.f;
      out_e5m2[g * 32 + i] = orc_e5m2_encode(q);
    }
  }
}

/* dequantised value of one packed-fp4 operand element (for tests that compare dequantised
 * tensors like tests/mxfp4_test.py:84-120 `_dq_fp4`). */
void orc_dequant_fp4(const uint8_t* packed, const uint8_t* sf_flat, int gs, int is_e4m3,
                     int64_t numel, float alpha, double* out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < numel; ++i) {
    uint8_t b = packed[i / 2];
    uint8_t code = (i & 1) ? (b >> 4) : (b & 0xF);
    uint8_t sb = sf_flat[i / gs];
    double s = is_e4m3 ? (double)orc_e4m3_decode(sb) : e8m0_to_f64(sb);
    out[i] = (double)orc_e2m1_decode(code) * s / (double)alpha;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* QAT-backward data-prep kernels (qutlass/csrc/quartet_bwd_sm120.cu)       */
/* ------------------------------------------------------------------------------------------ */

/* shared tail of quantize_g32t / quantize_g32qt (quartet_bwd_sm120.cu:304-323, :407-426):
 *   scale = amax|y| [ / alpha ] ; scale &= 0x7f800000 ; e8m0 = bits >> 23 ;
 *   q = y * (3.f / scale)   [ y * (3.f / (scale * alpha)) ] ; e2m1(q)
 * NO epsilon (unlike fusedQuantizeMx abs_max): an all-zero group gives scale 0, 3/0 = inf, 0*inf = NaN,
 * and cvt.rn.satfinite.e2m1x2 maps NaN to +6 (code 7) -- restated as is. */
static void bwd_group_tail(const float* y, int use_alpha, float alpha, uint8_t* out16, uint8_t* out_e8m0) {
  float amax = 0.f;
  for (int i = 0; i < 32; ++i) {
    float a = fabsf(y[i]);
    if (a > amax) amax = a;
  }
  float scale = use_alpha ? amax / alpha : amax;
  uint32_t sb;
  memcpy(&sb, &scale, 4);
  sb &= 0x7f800000u;
  memcpy(&scale, &sb, 4);
  *out_e8m0 = (uint8_t)(sb >> 23);
  const float mult = use_alpha ? 3.f / (scale * alpha) : 3.f / scale;
  float q[32];
  for (int i = 0; i < 32; ++i) q[i] = y[i] * mult;
  pack32(q, out16);
}

/*
 * backward_t_bf16 (quartet_bwd_sm120.cu:237-325; wrapper qutlass/__init__.py:206-243; test oracle
 * tests/quartet_test.py:155-173 applied to x.transpose(-2,-1), :239-245).
 * x: (B, N, M) bf16.  Output row (b, m), group g: y_j = sum_k x[b][32g+k][m] * h[k][j]  (x^T rotated per 32
 * along N), abs-max quantised.  out_e2m1: (B, M, N/2) bytes, out_e8m0: (B, M, N/32).
 */
void orc_backward_t_bf16(const uint16_t* x, const uint16_t* h, int64_t B, int64_t N, int64_t M, int acc_model,
                         uint8_t* out_e2m1, uint8_t* out_e8m0) {
  float hf[32 * 32];
  for (int i = 0; i < 1024; ++i) hf[i] = bf16_to_f32(h[i]);
  const int64_t G = N / 32;
#pragma omp parallel for schedule(static)
  for (int64_t bm = 0; bm < B * M; ++bm) {
    const int64_t b = bm / M, m = bm % M;
    for (int64_t g = 0; g < G; ++g) {
      uint16_t col[32];
      for (int k = 0; k < 32; ++k) col[k] = x[(b * N + 32 * g + k) * M + m];
      float y[32];
      rotate_group(col, hf, 32, acc_model, y);
      bwd_group_tail(y, 0, 1.f, out_e2m1 + (bm * G + g) * 16, out_e8m0 + bm * G + g);
    }
  }
}

/*
 * backward_qt_bf16 (quartet_bwd_sm120.cu:327-428; wrapper __init__.py:246-286; test quartet_test.py:247-260).
 * x_e2m1: (B, N, M/2), x_e8m0: (B, N, M/32).  Operand = bf16( e2m1 value * 2^(e8m0-127) ) (:369-375: bf16 product of
 * the decoded code and a bf16 whose bits are e8m0 << 7 -- exact), NOT divided by alpha; alpha enters the scale only.
 */
void orc_backward_qt_bf16(const uint8_t* x_e2m1, const uint8_t* x_e8m0, const uint16_t* h, float alpha, int64_t B,
                          int64_t N, int64_t M, int acc_model, uint8_t* out_e2m1, uint8_t* out_e8m0) {
  float hf[32 * 32];
  for (int i = 0; i < 1024; ++i) hf[i] = bf16_to_f32(h[i]);
  const int64_t G = N / 32;
#pragma omp parallel for schedule(static)
  for (int64_t bm = 0; bm < B * M; ++bm) {
    const int64_t b = bm / M, m = bm % M;
    for (int64_t g = 0; g < G; ++g) {
      uint16_t col[32];
      for (int k = 0; k < 32; ++k) {
        const int64_t row = b * N + 32 * g + k;
        const uint8_t byte = x_e2m1[row * (M / 2) + m / 2];
        const uint8_t code = (m & 1) ? (byte >> 4) : (byte & 0xF);
        const float sc = bf16_to_f32((uint16_t)((uint16_t)x_e8m0[row * (M / 32) + m / 32] << 7));
        col[k] = f32_to_bf16_rne(orc_e2m1_decode(code) * sc);
      }
      float y[32];
      rotate_group(col, hf, 32, acc_model, y);
      bwd_group_tail(y, 1, alpha, out_e2m1 + (bm * G + g) * 16, out_e8m0 + bm * G + g);
    }
  }
}

/* encode_e8m0_shiftm8 (quartet_bwd_sm120.cu:503-509): 0 -> 127, else (biased exponent of bf16_rn(amax)) - 7
 * (e8m0 conversion with round-toward-zero = the exponent field), in uint8 arithmetic. */
static uint8_t e8m0_shift7(float amax) {
  if (amax == 0.0f) return 127;
  const uint16_t b = f32_to_bf16_rne(amax);
  return (uint8_t)(((b >> 7) & 0xFF) - 7);
}

/* x / 2^(e-127) -> bf16_rn -> e4m3 satfinite RN (quartet_bwd_sm120.cu:580-586, :695-701).  qscale is the bf16 with bits
 * e << 7 (e = 0 gives bits 0x0040 = 2^-127, __nv_cvt_e8m0_to_bf16raw). */
static uint8_t requant_e4m3(float v, uint8_t e) {
  const float qscale = bf16_to_f32(e ? (uint16_t)((uint16_t)e << 7) : (uint16_t)0x0040);
  const float q = bf16_to_f32(f32_to_bf16_rne(v / qscale));
  return orc_e4m3_encode(q);
}

/*
 * backward_bf16_square_double_mxfp8 (quartet_bwd_sm120.cu:511-621; wrapper __init__.py:288-297; test oracle
 * quartet_test.py:284-307).  x: (m, n) bf16, m % 32 == 0, n % 128 == 0.  One shared exponent per 32x32 block.
 * y: (m, n) e4m3; row_scales: (m, n/32); col_scales: (n, m/32).
 */
void orc_backward_bf16_square_double_mxfp8(const uint16_t* x, int64_t m, int64_t n, uint8_t* y, uint8_t* row_scales,
                                           uint8_t* col_scales) {
  const int64_t BM = m / 32, BN = n / 32;
#pragma omp parallel for schedule(static)
  for (int64_t blk = 0; blk < BM * BN; ++blk) {
    const int64_t bi = blk / BN, bj = blk % BN;
    float amax = 0.f;
    for (int r = 0; r < 32; ++r)
      for (int c = 0; c < 32; ++c) amax = fmaxf(amax, fabsf(bf16_to_f32(x[(bi * 32 + r) * n + bj * 32 + c])));
    const uint8_t e = e8m0_shift7(amax);
    for (int r = 0; r < 32; ++r) {
      row_scales[(bi * 32 + r) * BN + bj] = e;
      col_scales[(bj * 32 + r) * BM + bi] = e;
      for (int c = 0; c < 32; ++c)
        y[(bi * 32 + r) * n + bj * 32 + c] = requant_e4m3(bf16_to_f32(x[(bi * 32 + r) * n + bj * 32 + c]), e);
    }
  }
}

/*
 * mxfp4_transpose_mxfp8 (quartet_bwd_sm120.cu:628-733; wrapper __init__.py:299-315; test oracle
 * quartet_test.py:310-366).  x_fp4: (m, n/2) packed e2m1, scales: (m, n/32) e8m0.  Dequantise to bf16 (value * scale,
 * bf16_rn), transpose, one shared exponent per 32 along m.  y: (n, m) e4m3; out_e8m0: (n, m/32).  m % 32 == 0.
 */
void orc_mxfp4_transpose_mxfp8(const uint8_t* x_fp4, const uint8_t* scales, int64_t m, int64_t n, uint8_t* y,
                               uint8_t* out_e8m0) {
  const int64_t GM = m / 32;
#pragma omp parallel for schedule(static)
  for (int64_t col = 0; col < n; ++col) {
    for (int64_t g = 0; g < GM; ++g) {
      float v[32];
      float amax = 0.f;
      for (int k = 0; k < 32; ++k) {
        const int64_t row = g * 32 + k;
        const uint8_t byte = x_fp4[row * (n / 2) + col / 2];
        const uint8_t code = (col & 1) ? (byte >> 4) : (byte & 0xF);
        const uint8_t se = scales[row * (n / 32) + col / 32];
        /* __nv_cvt_e8m0_to_bf16raw (:658-660): 0 -> 2^-127 (bits 0x0040), 0xFF -> NaN (the e8m0 NaN), else 2^(se-127) */
        const float s_in = se == 0xFF ? NAN : bf16_to_f32(se ? (uint16_t)((uint16_t)se << 7) : (uint16_t)0x0040);
        v[k] = bf16_to_f32(f32_to_bf16_rne(orc_e2m1_decode(code) * s_in));
        amax = fmaxf(amax, fabsf(v[k]));
      }
      const uint8_t e = e8m0_shift7(amax);
      out_e8m0[col * GM + g] = e;
      for (int k = 0; k < 32; ++k) y[col * m + g * 32 + k] = requant_e4m3(v[k], e);
    }
  }
}
