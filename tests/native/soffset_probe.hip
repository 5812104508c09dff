// Does the range check of a raw buffer descriptor (stride 0) on gfx950 include soffset?
// (hipcc --offload-arch=gfx950 -O2 tests/native/soffset_probe.hip -o tests/native/soffset_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint32_t* base, uint32_t* out, int soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(base), 0, 1024, 0x00020000);   // 1 KiB of records
  out[threadIdx.x] = __builtin_amdgcn_raw_buffer_load_b32(r, threadIdx.x * 4, soff, 0);                // voffset < 1024 always
  out[64 + threadIdx.x] = __builtin_amdgcn_raw_buffer_load_b32(r, threadIdx.x * 4 + soff, 0, 0);       // the same address through voffset
}
int main() {
  uint32_t h[1024], o[128], *d, *od;
  for (int i = 0; i < 1024; ++i) h[i] = 0xA0000000u + i;
  (void)hipMalloc(&d, sizeof h); (void)hipMalloc(&od, sizeof o);
  (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  for (int soff : {0, 512, 1024, 2048}) {
    k<<<1, 64>>>(d, od, soff);
    (void)hipMemcpy(o, od, sizeof o, hipMemcpyDeviceToHost);
    printf("soffset %4d: lane 0 / 63 via soffset -> %08x %08x   via voffset -> %08x %08x   (in-memory words: %08x %08x)\n", soff, o[0], o[63], o[64], o[127],
           h[soff / 4], h[soff / 4 + 63]);
  }
  return 0;
}
