// Unit check of stack_attention (zn_chain_kernel.h: the whole-step kernel's attention on 4 waves) against attn_pv_kernel<128, 4, 1>
// (the fused attention launch) on random q / K / V, contexts 1 .. 1024: outputs must be bit-identical.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fhip-fp32-correctly-rounded-divide-sqrt tools/stackattn_test.hip -o build/stackattn_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include <cmath>
#include "../zonos_amd/csrc/zn_chain_kernel.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ __launch_bounds__(256) void att_only(const bf16_t* kv, const bf16_t* q, const int* lengths, int max_len, int n_heads, int n_heads_kv, float scale, bf16_t* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  StackAttnLds& AL = *reinterpret_cast<StackAttnLds*>(dyn);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = blockIdx.x, npairs = n_heads_kv * 2, pair = c % npairs, slice = c / npairs, kvh = pair % n_heads_kv, ar = pair / n_heads_kv;
  const int L = lengths[ar] + 1, nk = n_heads_kv * 128;
  u32x4 kk0[4][4], vv0[4];
  stack_attn_issue_k(kk0, kv, ar, kvh, L, max_len, n_heads_kv, wave, 0, lane);
  stack_attn_issue_v(vv0, kv, ar, kvh, slice, L, max_len, n_heads_kv, wave, 0, lane);
  if (wave == 0) *(u32x4*)&AL.q[lane >> 4][(lane & 15) * 8] = ld16(q + ((size_t)ar * n_heads + kvh * 4 + (lane >> 4)) * 128 + (lane & 15) * 8);
  if (wave == 1 && lane < 32) {
    const bf16_t* rowp = kv + ((size_t)ar * max_len + (L - 1)) * (2 * nk) + (size_t)(lane < 16 ? 0 : nk) + (size_t)kvh * 128 + (lane & 15) * 8;
    const u32x4 v = ld16(rowp);
    if (lane < 16) *(u32x4*)&AL.knew[lane * 8] = v; else *(u32x4*)&AL.vnew[(lane - 16) * 8] = v;
  }
  __syncthreads();
  stack_attention(AL, kk0, vv0, kv, ar, kvh, slice, L, max_len, n_heads_kv, scale, wave, lane);
  __syncthreads();
  if (threadIdx.x < 128) out[((size_t)ar * n_heads + kvh * 4 + (threadIdx.x >> 5)) * 128 + slice * 32 + (threadIdx.x & 31)] = AL.out[threadIdx.x >> 5][threadIdx.x & 31];
}

int main() {
  const int H = 16, Hkv = 4, hd = 128, R = 2, max_len = 1100;
  const size_t kvn = (size_t)R * max_len * 2 * Hkv * hd, qn = (size_t)R * H * hd;
  std::vector<bf16_t> hkv(kvn), hq(qn);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  auto tobf = [](float f) { unsigned u; memcpy(&u, &f, 4); return (bf16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); };
  for (auto& v : hkv) v = tobf(rnd() * 2.0f);
  for (auto& v : hq) v = tobf(rnd() * 3.0f);
  bf16_t *kv, *q, *o1, *o2; int* len;
  CK(hipMalloc(&kv, kvn * 2)); CK(hipMalloc(&q, qn * 2)); CK(hipMalloc(&o1, qn * 2)); CK(hipMalloc(&o2, qn * 2)); CK(hipMalloc(&len, 8));
  CK(hipMemcpy(kv, hkv.data(), kvn * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(q, hq.data(), qn * 2, hipMemcpyHostToDevice));
  std::vector<bf16_t> h1(qn), h2(qn);
  int bad_total = 0;
  for (int L : {1, 2, 15, 16, 17, 26, 31, 32, 33, 38, 39, 40, 41, 47, 48, 49, 64, 65, 100, 127, 128, 129, 400, 511, 512, 513, 600, 893, 1000, 1024}) {
    int hl[2] = {L - 1, (L > 3 ? L - 3 : L) - 1};
    CK(hipMemcpy(len, hl, 8, hipMemcpyHostToDevice));
    AttnArgs a{};
    a.q = q; a.kv = kv; a.lengths = len; a.max_len = max_len; a.n_heads = H; a.n_heads_kv = Hkv; a.lcap = 1536; a.scale = 1.0f / sqrtf(128.f);
    a.out = o1; a.rows = R;
    hipLaunchKernelGGL((attn_pv_kernel<128, 4, 1>), dim3(4 * Hkv * R), dim3(512), 0, 0, a);
    hipLaunchKernelGGL(att_only, dim3(4 * Hkv * R), dim3(256), sizeof(StackAttnLds), 0, kv, q, len, max_len, H, Hkv, a.scale, o2);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h1.data(), o1, qn * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), o2, qn * 2, hipMemcpyDeviceToHost));
    int bad = 0, first = -1;
    for (size_t i = 0; i < qn; ++i) if (h1[i] != h2[i]) { if (first < 0) first = (int)i; ++bad; }
    printf("L = %4d / %4d: %d of %zu outputs differ%s\n", hl[0] + 1, hl[1] + 1, bad, qn, bad ? "" : "  (bit-identical)");
    if (bad) printf("   first at row %d head %d dim %d\n", first / (H * hd), (first / hd) % H, first % hd);
    bad_total += bad;
  }
  printf(bad_total ? "FAILED\n" : "ALL BIT-IDENTICAL\n");
  return bad_total != 0;
}
