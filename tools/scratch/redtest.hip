#include "../../zonos_amd/csrc/zn_common.h"
#include <cstdio>
__global__ void k(const float* a, float* o) {
  float v = a[threadIdx.x];
  o[threadIdx.x] = dpp_mov<ZN_DPP_XOR1>(v);
  o[64 + threadIdx.x] = dpp_mov<ZN_DPP_XOR2>(v);
  o[128 + threadIdx.x] = dpp_mov<ZN_DPP_HALF_MIRROR>(v);
  o[192 + threadIdx.x] = dpp_mov<ZN_DPP_MIRROR>(v);
  o[256 + threadIdx.x] = group_sum<8>(v);
  o[320 + threadIdx.x] = group_max<4>(v);
  o[384 + threadIdx.x] = wave_sum(v);
  o[448 + threadIdx.x] = group_sum<16>(v);
  o[512 + threadIdx.x] = row_stride4_sum(v);
  o[576 + threadIdx.x] = wave_max(v);
  o[640 + threadIdx.x] = dpp_mov<ZN_DPP_ROR4>(v);
}
int main() {
  float h[64], *d, *o, r[704];
  for (int i = 0; i < 64; ++i) h[i] = (float)(i + 1);
  hipMalloc(&d, 256); hipMalloc(&o, sizeof r);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  const char* names[] = {"xor1", "xor2", "halfmirror", "mirror", "swap16_sum", "swap32_sum", "wave_sum", "group_sum16", "stride4_sum", "wave_max", "ror4"};
  for (int t = 0; t < 11; ++t) { printf("%-12s", names[t]); for (int i = 0; i < 64; i += (t < 4 || t == 10 ? 1 : 4)) printf(" %g", r[t * 64 + i]); printf("\n"); }
  return 0;
}
