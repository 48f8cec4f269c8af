// Probe: do weight bytes touched by one launch stay in the XCDs' L2 (4 MB each) / the Infinity Cache for the NEXT launch on
// MI355X?  The decode step alternates a small attention launch (32 of 256 CUs busy, HBM idle for ~10 us) with a persistent
// chain launch that streams 121.6 MB of weights; if lines survive the kernel boundary, idle workgroups of the attention
// launch could warm the first tiles the chain will ask for.
//
// Kernel `consume` = 256 workgroups, workgroup c streams its own contiguous slice (475 KB, like one block's weights per
// CU) with non-temporal 16-B loads, as chain_kernel does.  Kernel `warm` touches the first P bytes of every slice with
// ordinary loads from a workgroup with the SAME blockIdx.x % 8 (round-robin XCD placement).  Timed: consume alone after a
// 1 GB thrash, and consume right after warm, with in-kernel s_memrealtime stamps (min start .. max end).
//
// Result on MI355X (ROCm 7.2): lines DO survive the boundary (25 MB warmed: the first 25 MB of consume 5.5 -> 2.5 us, the full
// 124.5 MB 21.1 -> 18.0 us; a dependency-free stream of this shape reaches 5.9 TB/s), from any grid with the same residues, with
// plain or non-temporal loads.  Built into the attention launch (224 extra workgroups warming out_proj + the first fc1 tiles of the
// chain that follows) it bought the chain 1.5-2.5 us and cost the attention 0.6-1.5 us (its length / K requests queue behind
// the bulk, delayed start or not) plus a later kernel end: 0.99-1.01 ms per decode step against 0.989 without.  Not adopted.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/l2_probe.hip -o build/l2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ __launch_bounds__(256) void consume(const char* W, size_t slice, size_t bytes, unsigned* sink, unsigned long long* stamps, int nt) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const char* p = W + (size_t)blockIdx.x * slice;
  unsigned acc = 0;
  for (size_t o = (size_t)threadIdx.x * 16; o < bytes; o += 256 * 16 * 16) {
    u32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const size_t oo = o + (size_t)i * 256 * 16;
      const u32x4* q = (const u32x4*)(p + (oo < bytes ? oo : 0));
      v[i] = nt ? __builtin_nontemporal_load(q) : *q;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
}

// grid = any multiple of 8; workgroup b warms slices b, b + grid, ... (same b % 8 as the consumer of those slices)
__global__ __launch_bounds__(256) void warm(const char* W, size_t slice, size_t bytes, int nslices, unsigned* sink, int nt) {
  unsigned acc = 0;
  for (int s = blockIdx.x; s < nslices; s += gridDim.x) {
    const char* p = W + (size_t)s * slice;
    for (size_t o = (size_t)threadIdx.x * 16; o < bytes; o += 256 * 16) {
      const u32x4* q = (const u32x4*)(p + o);
      const u32x4 v = nt ? __builtin_nontemporal_load(q) : *q;
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void thrash(const u32x4* B, size_t n, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const u32x4 v = B[i]; acc += v.x ^ v.w; }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const int G = 256;
  const size_t slice = 475 * 1024, total = slice * G;
  char* W; u32x4* T; unsigned* sink; unsigned long long* stamps;
  const size_t tbytes = (size_t)1 << 30;
  CK(hipMalloc(&W, total * 4)); CK(hipMalloc(&T, tbytes)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&stamps, G * 16));
  CK(hipMemset(W, 1, total * 4)); CK(hipMemset(T, 2, tbytes));
  std::vector<unsigned long long> hs(2 * G);
  auto run = [&](size_t warm_bytes, int warm_grid, int warm_nt, int cons_nt, size_t cons_bytes, int layer) {
    std::vector<double> us;
    for (int rep = 0; rep < 7; ++rep) {
      const char* Wl = W + (size_t)((layer + rep) % 4) * total;
      hipLaunchKernelGGL(thrash, dim3(2048), dim3(256), 0, 0, T, tbytes / 16, sink);
      if (warm_bytes) hipLaunchKernelGGL(warm, dim3(warm_grid), dim3(256), 0, 0, Wl, slice, warm_bytes, G, sink, warm_nt);
      hipLaunchKernelGGL(consume, dim3(G), dim3(256), 0, 0, Wl, slice, cons_bytes, sink, stamps, cons_nt);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(hs.data(), stamps, 2 * G * 8, hipMemcpyDeviceToHost));
      unsigned long long lo = ~0ull, hi = 0;
      for (int i = 0; i < G; ++i) { lo = std::min(lo, hs[2 * i]); hi = std::max(hi, hs[2 * i + 1]); }
      us.push_back((hi - lo) / 100.0);
    }
    std::sort(us.begin(), us.end());
    return us[us.size() / 2];
  };
  printf("consume = 256 workgroups x %zu KB (%.1f MB), median of 7, after a 1 GB thrash\n", slice / 1024, total / 1e6);
  for (size_t cb : {(size_t)32 * 1024, (size_t)96 * 1024, slice}) {
    for (int cnt = 1; cnt >= 0; --cnt) {
      const double base = run(0, 0, 0, cnt, cb, 0);
      printf("consume %4zu KB/wg (%6.1f MB) %s loads: cold %7.2f us (%6.0f GB/s)", cb / 1024, cb * G / 1e6, cnt ? "nt   " : "plain", base, cb * G / base / 1e3);
      for (size_t wb : {(size_t)32 * 1024, (size_t)64 * 1024, (size_t)96 * 1024, (size_t)128 * 1024}) {
        if (wb > cb && cb != slice) continue;
        const double t = run(wb, 256, 0, cnt, cb, 1);
        printf(" | warm %3zu KB/wg: %7.2f", wb / 1024, t);
      }
      printf("\n");
    }
  }
  // warm from a grid of 224 extra workgroups (+ 32 that do something else): same residue classes
  printf("warm grid 224 (plain), consume nt full slice: %7.2f us; warm nt loads 64 KB: %7.2f us\n", run(64 * 1024, 224, 0, 1, slice, 2), run(64 * 1024, 256, 1, 1, slice, 3));
  return 0;
}
