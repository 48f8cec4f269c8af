// Cross-workgroup hand-off latency on MI355X (hipcc --offload-arch=gfx950 -O3 tools/handoff_latency.hip; measurement only): how long does one "phase" of a persistent kernel take when every
// workgroup publishes a few values and every workgroup needs all of them?
//   mode 0: tagged 8-byte words (value | phase tag), consumers spin on the data itself
//   mode 1: plain stores + release, one atomic counter, poll, acquire, plain loads
// All spins are bounded; on overflow the kernel sets an abort flag and every workgroup leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define SPIN_CAP (1 << 22)

__device__ __forceinline__ unsigned long long ld_ll(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_ll(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// buf: [2][G*P] words.  Each phase: workgroup g writes P words; then every workgroup reads all G*P words.
__global__ void __launch_bounds__(256) ll_kernel(unsigned long long* buf, int G, int P, int phases, int* abort_flag,
                                                 unsigned long long* out, long long* cycles) {
  const int g = blockIdx.x, t = threadIdx.x;
  const int total = G * P;
  long long t0 = wall_clock64();
  unsigned acc = 0;
  for (int ph = 0; ph < phases; ++ph) {
    unsigned long long* b = buf + (size_t)(ph & 1) * total;
    const unsigned tag = ph + 1;
    if (t < P) st_ll(b + g * P + t, ((unsigned long long)tag << 32) | (unsigned)(g * P + t + ph));
    for (int i = t; i < total; i += 256) {
      unsigned long long v;
      int spins = 0;
      do {
        v = ld_ll(b + i);
        if (++spins > SPIN_CAP) { atomicExch(abort_flag, 1); break; }
      } while ((unsigned)(v >> 32) != tag);
      acc += (unsigned)v;
    }
    if (*(volatile int*)abort_flag) break;
    __syncthreads();
  }
  long long t1 = wall_clock64();
  if (t == 0) { cycles[g] = t1 - t0; }
  atomicAdd((unsigned*)out, acc);
}

__global__ void __launch_bounds__(256) ctr_kernel(unsigned* data, unsigned* counter, int G, int P, int phases, int* abort_flag,
                                                  unsigned long long* out, long long* cycles) {
  const int g = blockIdx.x, t = threadIdx.x;
  const int total = G * P;
  long long t0 = wall_clock64();
  unsigned acc = 0;
  for (int ph = 0; ph < phases; ++ph) {
    unsigned* b = data + (size_t)(ph & 1) * total;
    if (t < P) b[g * P + t] = g * P + t + ph;
    __syncthreads();
    if (t == 0) {
      __atomic_thread_fence(__ATOMIC_RELEASE);   // agent scope by default for HIP __atomic? use builtin below
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(G * (ph + 1))) {
        if (++spins > SPIN_CAP) { atomicExch(abort_flag, 1); break; }
      }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (*(volatile int*)abort_flag) break;
    for (int i = t; i < total; i += 256) acc += b[i];
  }
  long long t1 = wall_clock64();
  if (t == 0) { cycles[g] = t1 - t0; }
  atomicAdd((unsigned*)out, acc);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  int phases = 2000;
  int G = 256;
  unsigned long long* buf; unsigned* data; unsigned* counter; int* abort_flag; unsigned long long* out; long long* cycles;
  CK(hipMalloc(&buf, 2 * 256 * 256 * 8)); CK(hipMalloc(&data, 2 * 256 * 256 * 4)); CK(hipMalloc(&counter, 4));
  CK(hipMalloc(&abort_flag, 4)); CK(hipMalloc(&out, 8)); CK(hipMalloc(&cycles, 8 * 1024));
  int clk_khz = 0; CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0));
  printf("wall clock rate %d kHz\n", clk_khz);
  for (int mode = 0; mode < 2; ++mode)
    for (int P : {1, 8, 32, 128}) {
      CK(hipMemset(buf, 0, 2 * 256 * 256 * 8)); CK(hipMemset(data, 0, 2 * 256 * 256 * 4)); CK(hipMemset(counter, 0, 4));
      CK(hipMemset(abort_flag, 0, 4)); CK(hipMemset(out, 0, 8));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(ll_kernel, dim3(G), dim3(256), 0, 0, buf, G, P, phases, abort_flag, out, cycles);
      else hipLaunchKernelGGL(ctr_kernel, dim3(G), dim3(256), 0, 0, data, counter, G, P, phases, abort_flag, out, cycles);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      int ab; CK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost));
      std::vector<long long> cy(G); CK(hipMemcpy(cy.data(), cycles, 8 * G, hipMemcpyDeviceToHost));
      printf("mode %d (%s) P=%3d words/wg (%5d total): %.3f us/phase (event), wg0 %.3f us/phase (clock), abort=%d\n", mode,
             mode == 0 ? "tagged LL   " : "counter+fence", P, G * P, ms * 1e3 / phases, cy[0] * 1e3 / clk_khz / phases, ab);
      fflush(stdout);
    }
  return 0;
}
