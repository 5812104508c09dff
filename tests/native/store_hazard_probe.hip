// Does gfx950 need wait states between a 16-byte buffer store and an instruction that overwrites the store's DATA registers?
//   LLVM (GCNHazardRecognizer::createsVALUHazard) guards "VMEM store of more than 8 bytes -> VALU write of its data registers" with wait states ONLY for stores
//   that do NOT carry an SGPR in the soffset field; round 4 suspected the hardware of having the hazard for scalar-offset stores too (tools/store_data_hazard.py,
//   the unexplained wrong bytes of the QAMD_DEEPP_RB2 = 1 variant).  This probe settles it on the device: every lane stores a known pattern and overwrites the data
//   registers `dist` wait states later -- by a VALU write or by an LDS read returning into them -- then reads memory back (system-coherent load) and counts words
//   that are not the pattern.  Whole sequence in ONE asm statement, so the compiler's own hazard recognizer adds nothing inside it.
//     (hipcc --offload-arch=gfx950 -O2 tests/native/store_hazard_probe.hip -o tests/native/store_hazard_probe; prints one line per case)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));

// SOFF: 1 = SGPR soffset, 0 = immediate 0.  KILL: 0 = v_mov_b32 x 4, 1 = one ds_read_b128 into the data registers, 2 = v_pk_mul_f32 x 2 (the RB2 schedule's overwrite).
// DIST: s_nop wait states between the store and the overwrite (0 = directly behind it).
template <int SOFF, int KILL, int DIST>
__global__ __launch_bounds__(256) void probe(uint32_t* out, unsigned long long* bad, int iters, int slots) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[256 * 4];
  for (int i = 0; i < 4; ++i) lds[threadIdx.x * 4 + i] = 0xDEAD0000u + i;
  __syncthreads();
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7fffffff, 0x00020000);
  const uint32_t ldsa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(lds + threadIdx.x * 4);
  unsigned long long nbad = 0;
  for (int it = 0; it < iters; ++it) {
    const int slot = it % slots;
    const int voff = gid * 16;
    int soff = slot * (gridDim.x * 256 * 16);
    soff = __builtin_amdgcn_readfirstlane(soff);
    const uint32_t x = 0x10000000u + (uint32_t)it * 64u + (threadIdx.x & 63);
    const uint32_t poison = 0xBAD00000u + it;
    const int vtot = voff + soff;
#define SETUP "v_mov_b32 v20, %[x]\n v_add_u32 v21, 1, %[x]\n v_add_u32 v22, 2, %[x]\n v_add_u32 v23, 3, %[x]\n v_mov_b32 v24, 2.0\n v_mov_b32 v25, 2.0\n s_nop 7\n"
#define STORE_S SETUP "buffer_store_dwordx4 v[20:23], %[vo], %[rs], %[so] offen sc0 sc1\n"
#define STORE_I SETUP "buffer_store_dwordx4 v[20:23], %[vt], %[rs], 0 offen sc0 sc1\n"
#define NOPS(n) ((n) == 0 ? "" : (n) == 1 ? "s_nop 0\n" : (n) == 2 ? "s_nop 1\n" : "s_nop 3\n")
    if constexpr (SOFF) {
      if constexpr (KILL == 0) {
        if constexpr (DIST == 0) asm volatile(STORE_S "v_mov_b32 v20, %[p]\n v_mov_b32 v21, %[p]\n v_mov_b32 v22, %[p]\n v_mov_b32 v23, %[p]\n" ::[vo] "v"(voff), [x] "v"(x), [rs] "s"(rs), [so] "s"(soff), [p] "v"(poison) : "v20", "v21", "v22", "v23", "v24", "v25", "memory");
        if constexpr (DIST == 1) asm volatile(STORE_S "s_nop 0\n v_mov_b32 v20, %[p]\n v_mov_b32 v21, %[p]\n v_mov_b32 v22, %[p]\n v_mov_b32 v23, %[p]\n" ::[vo] "v"(voff), [x] "v"(x), [rs] "s"(rs), [so] "s"(soff), [p] "v"(poison) : "v20", "v21", "v22", "v23", "v24", "v25", "memory");
        if constexpr (DIST == 2) asm volatile(STORE_S "s_nop 1\n v_mov_b32 v20, %[p]\n v_mov_b32 v21, %[p]\n v_mov_b32 v22, %[p]\n v_mov_b32 v23, %[p]\n" ::[vo] "v"(voff), [x] "v"(x), [rs] "s"(rs), [so] "s"(soff), [p] "v"(poison) : "v20", "v21", "v22", "v23", "v24", "v25", "memory");
      } else if constexpr (KILL == 1) {
        asm volatile(STORE_S "ds_read_b128 v[20:23], %[la]\n s_waitcnt lgkmcnt(0)\n" ::[vo] "v"(voff), [x] "v"(x), [rs] "s"(rs), [so] "s"(soff), [la] "v"(ldsa) : "v20", "v21", "v22", "v23", "v24", "v25", "memory");
      } else {
        asm volatile(STORE_S "v_pk_mul_f32 v[20:21], v[20:21], v[24:25]\n v_pk_mul_f32 v[22:23], v[22:23], v[24:25]\n" ::[vo] "v"(voff), [x] "v"(x), [rs] "s"(rs), [so] "s"(soff) : "v20", "v21", "v22", "v23", "v24", "v25", "memory");
      }
    } else {
      if constexpr (KILL == 0) {
        if constexpr (DIST == 0) asm volatile(STORE_I "v_mov_b32 v20, %[p]\n v_mov_b32 v21, %[p]\n v_mov_b32 v22, %[p]\n v_mov_b32 v23, %[p]\n" ::[vt] "v"(vtot), [x] "v"(x), [rs] "s"(rs), [p] "v"(poison) : "v20", "v21", "v22", "v23", "v24", "v25", "memory");
        if constexpr (DIST == 1) asm volatile(STORE_I "s_nop 0\n v_mov_b32 v20, %[p]\n v_mov_b32 v21, %[p]\n v_mov_b32 v22, %[p]\n v_mov_b32 v23, %[p]\n" ::[vt] "v"(vtot), [x] "v"(x), [rs] "s"(rs), [p] "v"(poison) : "v20", "v21", "v22", "v23", "v24", "v25", "memory");
        if constexpr (DIST == 2) asm volatile(STORE_I "s_nop 1\n v_mov_b32 v20, %[p]\n v_mov_b32 v21, %[p]\n v_mov_b32 v22, %[p]\n v_mov_b32 v23, %[p]\n" ::[vt] "v"(vtot), [x] "v"(x), [rs] "s"(rs), [p] "v"(poison) : "v20", "v21", "v22", "v23", "v24", "v25", "memory");
      } else if constexpr (KILL == 1) {
        asm volatile(STORE_I "ds_read_b128 v[20:23], %[la]\n s_waitcnt lgkmcnt(0)\n" ::[vt] "v"(vtot), [x] "v"(x), [rs] "s"(rs), [la] "v"(ldsa) : "v20", "v21", "v22", "v23", "v24", "v25", "memory");
      } else {
        asm volatile(STORE_I "v_pk_mul_f32 v[20:21], v[20:21], v[24:25]\n v_pk_mul_f32 v[22:23], v[22:23], v[24:25]\n" ::[vt] "v"(vtot), [x] "v"(x), [rs] "s"(rs) : "v20", "v21", "v22", "v23", "v24", "v25", "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const v4i back = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 17));
    nbad += (uint32_t)back[0] != x;
    nbad += (uint32_t)back[1] != x + 1;
    nbad += (uint32_t)back[2] != x + 2;
    nbad += (uint32_t)back[3] != x + 3;
  }
  if (nbad) atomicAdd(bad, nbad);
}

template <int SOFF, int KILL, int DIST>
static void run(const char* what, uint32_t* out, unsigned long long* bad, int grid, int iters, int slots) {
  (void)hipMemset(bad, 0, 8);
  probe<SOFF, KILL, DIST><<<grid, 256>>>(out, bad, iters, slots);
  unsigned long long h = 0;
  hipError_t e = hipDeviceSynchronize();
  (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
  printf("%-94s %llu wrong words of %llu%s\n", what, h, 4ull * grid * 256 * iters, e == hipSuccess ? "" : "  (HIP error)");
}

int main() {
  const int grid = 1024, iters = 2000, slots = 8;
  uint32_t* out;
  unsigned long long* bad;
  (void)hipMalloc(&out, (size_t)slots * grid * 256 * 16);
  (void)hipMalloc(&bad, 8);
  printf("store-data hazard probe: %d workgroups x 256 lanes x %d iterations, 16-byte buffer stores (sc0 sc1), data registers overwritten `dist` wait states behind the store\n", grid, iters);
  run<1, 0, 0>("SGPR soffset   | v_mov_b32 x 4 directly behind the store (dist 0)", out, bad, grid, iters, slots);
  run<1, 0, 1>("SGPR soffset   | v_mov_b32 x 4 after s_nop 0 (dist 1)", out, bad, grid, iters, slots);
  run<1, 0, 2>("SGPR soffset   | v_mov_b32 x 4 after s_nop 1 (dist 2)", out, bad, grid, iters, slots);
  run<1, 2, 0>("SGPR soffset   | v_pk_mul_f32 x 2 directly behind the store (the RB2 = 1 pattern)", out, bad, grid, iters, slots);
  run<1, 1, 0>("SGPR soffset   | ds_read_b128 into the data registers directly behind the store", out, bad, grid, iters, slots);
  run<0, 0, 0>("immediate soff | v_mov_b32 x 4 directly behind the store (dist 0; LLVM inserts wait states here)", out, bad, grid, iters, slots);
  run<0, 0, 1>("immediate soff | v_mov_b32 x 4 after s_nop 0 (dist 1)", out, bad, grid, iters, slots);
  run<0, 0, 2>("immediate soff | v_mov_b32 x 4 after s_nop 1 (dist 2)", out, bad, grid, iters, slots);
  run<0, 2, 0>("immediate soff | v_pk_mul_f32 x 2 directly behind the store", out, bad, grid, iters, slots);
  run<0, 1, 0>("immediate soff | ds_read_b128 into the data registers directly behind the store", out, bad, grid, iters, slots);
  return 0;
}
