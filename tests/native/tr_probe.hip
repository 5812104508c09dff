#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short v4s __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void k16(const uint16_t* in, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t s[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) s[i] = in[i];
  __syncthreads();
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(s + threadIdx.x * 4));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = r[j];
}
__global__ void k8(const uint8_t* in, uint8_t* out) {
  __shared__ __attribute__((aligned(16))) uint8_t s[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) s[i] = in[i];
  __syncthreads();
  v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(s + threadIdx.x * 8));
  *(v2i*)(out + threadIdx.x * 8) = r;
}
__global__ void k4(const uint8_t* in, uint8_t* out) {
  __shared__ __attribute__((aligned(16))) uint8_t s[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) s[i] = in[i];
  __syncthreads();
  v2i r = __builtin_amdgcn_ds_read_tr4_b64_v2i32((__attribute__((address_space(3))) v2i*)(s + threadIdx.x * 8));
  *(v2i*)(out + threadIdx.x * 8) = r;
}
int main() {
  uint16_t h16[1024], o16[256]; uint8_t h8[1024], o8[512];
  for (int i = 0; i < 1024; ++i) { h16[i] = i; h8[i] = i & 255; }
  void *d, *o; hipMalloc(&d, 2048); hipMalloc(&o, 2048);
  hipMemcpy(d, h16, 2048, hipMemcpyHostToDevice);
  k16<<<1, 64>>>((uint16_t*)d, (uint16_t*)o); hipMemcpy(o16, o, 512, hipMemcpyDeviceToHost);
  printf("tr16: element index (in b16 units) received by lane l elem j, lane addr = 8*l bytes\n");
  for (int l = 0; l < 64; ++l) { printf("l%2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", o16[l * 4 + j]); printf("\n"); }
  hipMemcpy(d, h8, 1024, hipMemcpyHostToDevice);
  k8<<<1, 64>>>((uint8_t*)d, (uint8_t*)o); hipMemcpy(o8, o, 512, hipMemcpyDeviceToHost);
  printf("tr8: byte index (mod 256) received by lane l byte j, lane addr = 8*l bytes\n");
  for (int l = 0; l < 64; ++l) { printf("l%2d:", l); for (int j = 0; j < 8; ++j) printf(" %4d", o8[l * 8 + j] + (l >= 32 ? 256 : 0)); printf("\n"); }
  k4<<<1, 64>>>((uint8_t*)d, (uint8_t*)o); hipMemcpy(o8, o, 512, hipMemcpyDeviceToHost);
  printf("tr4: raw bytes received by lane l (input byte i = i & 255 ; nibble lo = i & 15, hi = (i >> 4) & 15)\n");
  for (int l = 0; l < 64; ++l) { printf("l%2d:", l); for (int j = 0; j < 8; ++j) printf(" %02x", o8[l * 8 + j]); printf("\n"); }
  return 0;
}
