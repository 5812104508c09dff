// Issue cost of the retirement's instructions with ONE wave per SIMD (4 waves per workgroup, one workgroup per CU), no MFMA anywhere: fixed registers, straight asm,
// 16 copies of the sequence per loop trip.  Prints shader cycles per sequence (s_memtime, wave 0 of workgroup 0, median of 5 launches).
//   hipcc --offload-arch=gfx950 -O3 tests/native/issue_ubench.hip -o tests/native/issue_ubench && tests/native/issue_ubench [workgroups]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define AR "v_accvgpr_read_b32 v10, a0\n v_accvgpr_read_b32 v11, a1\n v_accvgpr_read_b32 v12, a2\n v_accvgpr_read_b32 v13, a3\n"
#define AR2 "v_accvgpr_read_b32 v10, a16\n v_accvgpr_read_b32 v11, a17\n v_accvgpr_read_b32 v12, a18\n v_accvgpr_read_b32 v13, a19\n"
#define MUL "v_mul_f32 v10, %[al], v10\n v_mul_f32 v11, %[al], v11\n v_mul_f32 v12, %[al], v12\n v_mul_f32 v13, %[al], v13\n"
#define PKMUL "v_pk_mul_f32 v[10:11], v[10:11], v[16:17]\n v_pk_mul_f32 v[12:13], v[12:13], v[16:17]\n"
#define CVT "v_cvt_pk_bf16_f32 v14, v10, v11\n v_cvt_pk_bf16_f32 v15, v12, v13\n"
#define W64 "ds_write_b64 %[la], v[14:15]\n"
#define W128A "ds_write_b128 %[la], a[0:3]\n"
#define R128 "ds_read_b128 v[20:23], %[la]\n"
#define MOV "v_mov_b32 v10, v18\n v_mov_b32 v11, v18\n v_mov_b32 v12, v18\n v_mov_b32 v13, v18\n"
#define MFMA "v_mfma_scale_f32_32x32x64_f8f6f4 a[32:47], v[24:27], v[28:31], a[32:47], v18, v18 op_sel_hi:[0,0,0] cbsz:4 blgp:4\n"
#define MFMB "v_mfma_scale_f32_32x32x64_f8f6f4 a[48:63], v[24:27], v[28:31], a[48:63], v18, v18 op_sel_hi:[0,0,0] cbsz:4 blgp:4\n"
#define X4(s) s s s s
#define X16(s) X4(X4(s))
#define X64(s) X4(X16(s))
#define X256(s) X4(X64(s))

template <int V>
__global__ __launch_bounds__(256) void k(uint32_t* out, float alpha, int trips) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + threadIdx.x * 16);
  asm volatile("v_mov_b32 v16, 2.0\n v_mov_b32 v17, 2.0\n v_mov_b32 v18, 0\n v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n"
               "v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0\n"
               ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < trips; ++i) {
#define BODY(seq) asm volatile(X16(seq) ::[al] "s"(alpha), [la] "v"(la) : "v10", "v11", "v12", "v13", "v14", "v15", "v20", "v21", "v22", "v23", "a0", "a1", "a2", "a3", "a16", "a17", "a18", "a19", "memory")
    if constexpr (V == 0) BODY(AR MUL CVT W64);
    if constexpr (V == 1) BODY(AR);
    if constexpr (V == 2) BODY(MUL);
    if constexpr (V == 3) BODY(CVT);
    if constexpr (V == 4) BODY(W64);
    if constexpr (V == 5) BODY(AR CVT W64);
    if constexpr (V == 6) BODY(MOV MUL CVT W64);
    if constexpr (V == 7) BODY(W128A);
    if constexpr (V == 8) BODY(R128);
    if constexpr (V == 9) BODY(PKMUL);
    if constexpr (V == 10) BODY(AR AR2);
    if constexpr (V == 11) BODY(MOV);
    if constexpr (V == 12) BODY(MFMA MFMB);                      // two independent MFMAs: cycles per PAIR
    if constexpr (V == 13) BODY(MFMA MOV MFMB MOV);              //   + 4 v_mov_b32 behind each
    if constexpr (V == 14) BODY(MFMA AR MFMB AR2);               //   + 4 v_accvgpr_read_b32 (of other registers) behind each
    if constexpr (V == 15) BODY(MFMA AR MUL CVT W64 MFMB AR2 MUL CVT W64);   //   + a piece behind each
    if constexpr (V == 16) BODY(MFMA W128A MFMB W128A);
    if constexpr (V == 17) BODY(MFMA R128 MFMB R128);
    if constexpr (V == 18) BODY(MFMA MOV MOV MFMB MOV MOV);      //   + 8 v_mov_b32 behind each
    // code size: the same piece, 256 copies in a row = 2 816 instructions = 22 KB of straight-line code per trip (the last stage of gemm_mx_deepp is ~1 200 instructions
    // executed once per tile) -- does instruction supply hold the issue rate of the 16-copy loop?
#define BODYN(xn, seq) asm volatile(xn(seq) ::[al] "s"(alpha), [la] "v"(la) : "v10", "v11", "v12", "v13", "v14", "v15", "v20", "v21", "v22", "v23", "a0", "a1", "a2", "a3", "a16", "a17", "a18", "a19", "memory")
    if constexpr (V == 19) BODYN(X256, AR MUL CVT W64);
    if constexpr (V == 20) BODYN(X64, MFMA AR MUL CVT W64 MFMB AR2 MUL CVT W64);
    if constexpr (V == 21) BODYN(X256, MOV);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const uint64_t t1 = __builtin_readcyclecounter();
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (uint32_t)(t1 - t0);
}

static uint32_t* d_out;
template <int V>
static void run(const char* what, int grid, double scale = 1.0) {
  const int trips = 200;
  std::vector<uint32_t> c;
  for (int rep = 0; rep < 5; ++rep) {
    k<V><<<grid, 256>>>(d_out, 1.5f, trips);
    uint32_t h = 0;
    (void)hipMemcpy(&h, d_out, 4, hipMemcpyDeviceToHost);
    c.push_back(h);
  }
  std::sort(c.begin(), c.end());
  printf("%-100s %7.1f cycles\n", what, c[2] / (16.0 * trips) / scale);
}

int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 8;
  (void)hipMalloc(&d_out, 64);
  printf("issue_ubench: %d workgroups x 4 waves (one per SIMD); cycles per sequence\n", grid);
  run<0>("piece: 4 v_accvgpr_read + 4 v_mul_f32 + 2 v_cvt_pk_bf16_f32 + ds_write_b64", grid);
  run<1>("4 v_accvgpr_read_b32", grid);
  run<10>("8 v_accvgpr_read_b32", grid);
  run<11>("4 v_mov_b32", grid);
  run<2>("4 v_mul_f32 (each on its own register)", grid);
  run<9>("2 v_pk_mul_f32", grid);
  run<3>("2 v_cvt_pk_bf16_f32", grid);
  run<4>("1 ds_write_b64", grid);
  run<7>("1 ds_write_b128 from AGPRs", grid);
  run<8>("1 ds_read_b128", grid);
  run<5>("4 v_accvgpr_read + 2 v_cvt_pk + ds_write_b64", grid);
  run<6>("4 v_mov + 4 v_mul + 2 v_cvt_pk + ds_write_b64", grid);
  run<12>("2 independent scaled fp4 MFMAs 32x32x64 (per PAIR)", grid);
  run<13>("2 MFMAs, 4 v_mov_b32 behind each", grid);
  run<18>("2 MFMAs, 8 v_mov_b32 behind each", grid);
  run<14>("2 MFMAs, 4 v_accvgpr_read_b32 behind each", grid);
  run<15>("2 MFMAs, a piece behind each", grid);
  run<16>("2 MFMAs, a ds_write_b128 from AGPRs behind each", grid);
  run<17>("2 MFMAs, a ds_read_b128 behind each", grid);
  run<19>("piece, 256 copies straight-line (22 KB of code per trip)", grid, 16.0);
  run<21>("4 v_mov_b32, 256 copies straight-line (8 KB of code per trip)", grid, 16.0);
  run<20>("2 MFMAs + a piece behind each, 64 copies straight-line (15 KB per trip; per PAIR)", grid, 4.0);
  return 0;
}
