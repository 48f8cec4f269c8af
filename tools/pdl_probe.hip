// Probe: can the decode chain's dependent GEMVs overlap across their dependency on MI355X?
//
// Each kernel of the chain requests its first weight tiles BEFORE waiting (in-kernel, bounded) for its
// predecessor's arrival counters, so launch gap, ramp and first-byte latency of kernel k+1 hide under kernel k.
// Hand-off form: Guideline 16 R1 (sc1 payload stores, every storing wave drains, one lane per workgroup adds to a
// sharded agent-scope counter; consumer: one wave polls the shards with sc1 loads, workgroup barrier, sc1 loads).
//
// Measured variants (same kernels, same data, outputs compared bit for bit):
//   base   one stream, ordinary dependent launches, no flags                 (what the library does today)
//   s2/s3  kernels dealt round-robin over 2 / 3 streams, flags carry the dependency
//   any    one stream, hipExtAnyOrderLaunch (documented "not supported on gfx9": probed, not assumed)
// each as eager launches and as a captured hipGraph.  Every spin is bounded (s_memrealtime) and reports a timeout.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pdl_probe.hip -o build/pdl_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include "../zonos_amd/csrc/zn_common.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define NSHARD 16            // arrival counter shards, one 128-B line each
#define SHARD_STRIDE 32      // in u32

struct PArgs {
  const bf16_t* W; int N, K;
  const bf16_t* x;           // [2][K]
  bf16_t* out;               // [2][N] (silu: [2][N/2])
  const bf16_t* resid;       // optional [2][N]
  int silu;                  // fc1: out = y * silu(gate), rows u / u + N/2
  unsigned* wait_ctr; unsigned wait_n;   // predecessor's counter block (NSHARD shards), total arrivals expected
  unsigned* sig_ctr;                     // own counter block
  unsigned* tmo;                         // timeout word
  unsigned long long* stamps;            // optional [2]: min start / max end (s_memrealtime)
  int pre;                               // units requested before the wait (0: wait first = ordering only)
};

ZN_DEVINL u32x4 ld_sc1_16(const void* p) {          // two 8-B sc1 loads (relaxed agent atomics): bypass this CU's L1
  const unsigned long long* q = (const unsigned long long*)p;
  const unsigned long long a = __hip_atomic_load(q, RLX_AGENT), b = __hip_atomic_load(q + 1, RLX_AGENT);
  return u32x4{(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
}
ZN_DEVINL void st_sc1_4(void* p, unsigned v) { __hip_atomic_store((unsigned*)p, v, RLX_AGENT); }

// one wave polls the predecessor's shards (lane i < NSHARD reads shard i), bounded by ~2 ms
ZN_DEVINL bool wait_arrivals(const unsigned* ctr, unsigned expect, unsigned* tmo, int lane) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    unsigned v = 0;
    if (lane < NSHARD) v = __hip_atomic_load(ctr + lane * SHARD_STRIDE, RLX_AGENT);
    unsigned s = v;
#pragma unroll
    for (int o = 1; o < NSHARD; o <<= 1) s += __shfl_xor(s, o);
    s = __builtin_amdgcn_readfirstlane(s);
    if (s >= expect) return true;
    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000ull) { if (lane == 0) atomicAdd(tmo, 1u); return false; }
    __builtin_amdgcn_s_sleep(2);
  }
}

// K = 2048: one wave per weight-row pair (NCH = 4), units dealt (block*4 + wave) + i * 4*gridDim
// K = 8192: KS = 4, the four waves of a block split K, units dealt block + i * gridDim
template <int KS>
__global__ __launch_bounds__(256) void pgemv(PArgs a) {
  constexpr int NCH = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K = a.K, kbase = (KS == 1) ? 0 : wave * (K / KS);
  const int units = a.N >> 1, F = a.N >> 1;
  const int ustride = (KS == 1) ? gridDim.x * 4 : gridDim.x;
  const int u0 = (KS == 1) ? blockIdx.x * 4 + wave : blockIdx.x;
  __shared__ float red[4][2][2];
  unsigned long long tstart = 0;
  if (a.stamps && threadIdx.x == 0) tstart = __builtin_amdgcn_s_memrealtime();
  auto rows = [&](int u, int& rA, int& rB) { if (a.silu) { rA = u; rB = u + F; } else { rA = 2 * u; rB = 2 * u + 1; } };
  u32x4 wa[NCH], wb[NCH];
  auto load_unit = [&](int u) {
    int rA, rB; rows(u < units ? u : units - 1, rA, rB);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int k = (c * 64 + lane) * 8;
      wa[c] = ld_nt16(a.W + (size_t)rA * K + kbase + k);
      wb[c] = ld_nt16(a.W + (size_t)rB * K + kbase + k);
    }
  };
  const bool flagged = a.wait_ctr != nullptr;
  if (a.pre) load_unit(u0);
  if (flagged) {
    if (wave == 0) wait_arrivals(a.wait_ctr, a.wait_n, a.tmo, lane);
    __syncthreads();
  }
  u32x4 xr[NCH][2];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const bf16_t* p = a.x + (size_t)r * K + kbase + (c * 64 + lane) * 8;
      xr[c][r] = flagged ? ld_sc1_16(p) : ld16(p);
    }
  if (!a.pre) load_unit(u0);
  for (int u = u0; u < units; u += ustride) {
    u32x4 ca[NCH], cb[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) { ca[c] = wa[c]; cb[c] = wb[c]; }
    int rA, rB; rows(u, rA, rB);
    unsigned resid = 0;
    if (a.resid && lane < 2 && (KS == 1 || wave == 0)) {
      const bf16_t* rp = a.resid + (size_t)lane * a.N + rA;
      resid = flagged ? __hip_atomic_load((const unsigned*)rp, RLX_AGENT) : *(const unsigned*)rp;
    }
    if (u + ustride < units) load_unit(u + ustride);
    float accA[2] = {0.f, 0.f}, accB[2] = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int r = 0; r < 2; ++r) { accA[r] = dot8(ca[c], xr[c][r], accA[r]); accB[r] = dot8(cb[c], xr[c][r], accB[r]); }
#pragma unroll
    for (int r = 0; r < 2; ++r) { accA[r] = wave_sum(accA[r]); accB[r] = wave_sum(accB[r]); }
    if constexpr (KS > 1) {
      __syncthreads();
      if (lane == 0) { for (int r = 0; r < 2; ++r) { red[wave][0][r] = accA[r]; red[wave][1][r] = accB[r]; } }
      __syncthreads();
      for (int r = 0; r < 2; ++r) {
        accA[r] = ((red[0][0][r] + red[1][0][r]) + red[2][0][r]) + red[3][0][r];
        accB[r] = ((red[0][1][r] + red[1][1][r]) + red[2][1][r]) + red[3][1][r];
      }
      if (wave != 0) continue;
    }
    if (lane < 2) {
      const float vA = lane == 0 ? accA[0] : accA[1], vB = lane == 0 ? accB[0] : accB[1];
      if (a.silu) {
        const float y = bfround(vA), g = bfround(vB), s = bfround(g / (1.0f + expf(-g)));
        const bf16_t o = f2bf(y * s);
        // two bf16 of neighbouring units share a dword: write 2-byte (plain store is fine for base; sc1 path uses a short atomic-free store)
        if (flagged) __hip_atomic_store((unsigned short*)(a.out + (size_t)lane * F + u), o, RLX_AGENT);
        else a.out[(size_t)lane * F + u] = o;
      } else {
        unsigned o;
        if (a.resid) o = pack2(lo_f(resid) + bfround(vA), hi_f(resid) + bfround(vB)); else o = pack2(vA, vB);
        if (flagged) st_sc1_4(a.out + (size_t)lane * a.N + rA, o); else *(unsigned*)(a.out + (size_t)lane * a.N + rA) = o;
      }
    }
  }
  if (a.sig_ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(a.sig_ctr + (blockIdx.x % NSHARD) * SHARD_STRIDE, 1u, RLX_AGENT);
  }
  if (a.stamps && threadIdx.x == 0) {
    atomicMin(a.stamps, tstart);
    atomicMax(a.stamps + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
  }
}

__global__ void fill_bf16(bf16_t* p, size_t n, unsigned seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = f2bf(((float)(h & 0xffffff) / 8388608.0f - 1.0f) * scale);
  }
}

// ---------------------------------------------------------------- probe A/B: do two kernels overlap at all?
__global__ void spin_kernel(unsigned long long* stamps, unsigned* flag, int us) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = t0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) { __hip_atomic_store(flag, 1u, RLX_AGENT); stamps[1] = __builtin_amdgcn_s_memrealtime(); }
}
__global__ void observe_kernel(unsigned long long* stamps, unsigned* flag, unsigned* seen) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[2] = __builtin_amdgcn_s_memrealtime(); *seen = __hip_atomic_load(flag, RLX_AGENT); }
}

struct Layer { bf16_t *in_proj, *out_proj, *fc1, *fc2; };

int main(int argc, char** argv) {
  const int NL = argc > 1 ? atoi(argv[1]) : 26, REPS = argc > 2 ? atoi(argv[2]) : 20;
  const int d = 2048, F = 8192, NQKV = 3072;
  CK(hipSetDevice(0));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  hipStream_t st[3];
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));

  // ---------------- A/B: overlap observation
  unsigned long long* stamps; unsigned *flag, *seen;
  CK(hipMalloc(&stamps, 64)); CK(hipMalloc(&flag, 4)); CK(hipMalloc(&seen, 4));
  auto observe = [&](const char* name, int mode) {
    CK(hipMemset(stamps, 0, 64)); CK(hipMemset(flag, 0, 4)); CK(hipMemset(seen, 0xff, 4));
    CK(hipDeviceSynchronize());
    if (mode == 0) {          // same stream, ordinary
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st[0], stamps, flag, 30);
      hipLaunchKernelGGL(observe_kernel, dim3(1), dim3(64), 0, st[0], stamps, flag, seen);
    } else if (mode == 1) {   // same stream, any-order flag on the second
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st[0], stamps, flag, 30);
      hipExtLaunchKernelGGL(observe_kernel, dim3(1), dim3(64), 0, st[0], nullptr, nullptr, hipExtAnyOrderLaunch, stamps, flag, seen);
    } else {                  // two streams
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st[0], stamps, flag, 30);
      hipLaunchKernelGGL(observe_kernel, dim3(1), dim3(64), 0, st[1], stamps, flag, seen);
    }
    hipError_t e = hipDeviceSynchronize();
    unsigned long long hs[3]; unsigned sn;
    CK(hipMemcpy(hs, stamps, 24, hipMemcpyDeviceToHost)); CK(hipMemcpy(&sn, seen, 4, hipMemcpyDeviceToHost));
    printf("overlap %-28s: err=%s second kernel started %+.2f us after the first started (first ran %.2f us), saw flag=%u -> %s\n", name,
           hipGetErrorString(e), ((double)hs[2] - (double)hs[0]) / 100.0, ((double)hs[1] - (double)hs[0]) / 100.0, sn, sn == 0 ? "OVERLAPPED" : "serialised");
    (void)hipGetLastError();
  };
  observe("same stream", 0);
  observe("same stream + AnyOrderLaunch", 1);
  observe("two streams", 2);

  // ---------------- C: the chain
  std::vector<Layer> L(NL);
  for (int i = 0; i < NL; ++i) {
    CK(hipMalloc(&L[i].in_proj, (size_t)NQKV * d * 2)); CK(hipMalloc(&L[i].out_proj, (size_t)d * d * 2));
    CK(hipMalloc(&L[i].fc1, (size_t)2 * F * d * 2)); CK(hipMalloc(&L[i].fc2, (size_t)d * F * 2));
    hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, 0, L[i].in_proj, (size_t)NQKV * d, 11u + i, 0.03f);
    hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, 0, L[i].out_proj, (size_t)d * d, 12u + 7 * i, 0.03f);
    hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, 0, L[i].fc1, (size_t)2 * F * d, 13u + 13 * i, 0.03f);
    hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, 0, L[i].fc2, (size_t)d * F, 14u + 17 * i, 0.015f);
  }
  bf16_t *x, *x0, *q, *o1, *m, *xref;
  CK(hipMalloc(&x, 2 * d * 2)); CK(hipMalloc(&x0, 2 * d * 2)); CK(hipMalloc(&xref, 2 * d * 2));
  CK(hipMalloc(&q, 2 * NQKV * 2)); CK(hipMalloc(&o1, 2 * d * 2)); CK(hipMalloc(&m, 2 * F * 2));
  hipLaunchKernelGGL(fill_bf16, dim3(16), dim3(256), 0, 0, x0, (size_t)2 * d, 99u, 1.0f);
  const int NK = NL * 5;
  unsigned *ctr, *tmo;
  const size_t ctr_bytes = (size_t)(NK + 1) * NSHARD * SHARD_STRIDE * 4;
  CK(hipMalloc(&ctr, ctr_bytes)); CK(hipMalloc(&tmo, 16)); CK(hipMemset(tmo, 0, 16));
  unsigned long long* kst; CK(hipMalloc(&kst, (size_t)NK * 16));
  CK(hipDeviceSynchronize());

  struct Variant { const char* name; int nstreams; bool flags; bool anyorder; int pre; int blocks; };
  auto ctr_of = [&](int k) { return ctr + (size_t)k * NSHARD * SHARD_STRIDE; };
  // enqueue one forward of the chain
  auto enqueue = [&](const Variant& v, hipStream_t* ss, bool stamp) {
    int k = 0;
    auto launch = [&](const bf16_t* W, int N, int K, const bf16_t* xin, bf16_t* out, const bf16_t* resid, int silu, int prev_blocks, int& my_blocks) {
      PArgs a{}; a.W = W; a.N = N; a.K = K; a.x = xin; a.out = out; a.resid = resid; a.silu = silu; a.tmo = tmo; a.pre = v.pre;
      my_blocks = v.blocks;
      if (v.flags) { if (k > 0) { a.wait_ctr = ctr_of(k - 1); a.wait_n = prev_blocks; } else { a.wait_ctr = ctr_of(NK); a.wait_n = 0; } a.sig_ctr = ctr_of(k); }
      a.stamps = stamp ? kst + 2 * k : nullptr;
      hipStream_t s = ss[k % v.nstreams];
      const unsigned fl = (v.anyorder && k > 0) ? hipExtAnyOrderLaunch : 0;
      if (K == 2048) hipExtLaunchKernelGGL((pgemv<1>), dim3(my_blocks), dim3(256), 0, s, nullptr, nullptr, fl, a);
      else hipExtLaunchKernelGGL((pgemv<4>), dim3(my_blocks), dim3(256), 0, s, nullptr, nullptr, fl, a);
      ++k;
    };
    int pb = 0, mb = 0;
    for (int i = 0; i < NL; ++i) {
      launch(L[i].in_proj, NQKV, d, x, q, nullptr, 0, pb, mb); pb = mb;
      launch(L[i].out_proj, d, d, q, o1, nullptr, 0, pb, mb); pb = mb;      // the first 2 x 2048 values of q stand in for the attention output
      launch(L[i].out_proj, d, d, o1, x, x, 0, pb, mb); pb = mb;            // second out_proj + residual, in place
      launch(L[i].fc1, 2 * F, d, x, m, nullptr, 1, pb, mb); pb = mb;
      launch(L[i].fc2, d, F, m, x, x, 0, pb, mb); pb = mb;
    }
  };
  auto run_variant = [&](const Variant& v, bool graph) -> double {
    hipStream_t* ss = st;
    std::string nm = std::string(v.name) + (graph ? " [graph]" : " [eager]");
    // fork/join helpers
    hipEvent_t ef, ej[3]; CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); for (auto& e : ej) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    auto forward = [&](bool stamp) {
      CK(hipMemcpyAsync(x, x0, 2 * d * 2, hipMemcpyDeviceToDevice, ss[0]));
      if (v.flags) CK(hipMemsetAsync(ctr, 0, ctr_bytes, ss[0]));
      if (v.nstreams > 1) { CK(hipEventRecord(ef, ss[0])); for (int j = 1; j < v.nstreams; ++j) CK(hipStreamWaitEvent(ss[j], ef, 0)); }
      enqueue(v, ss, stamp);
      if (v.nstreams > 1) for (int j = 1; j < v.nstreams; ++j) { CK(hipEventRecord(ej[j], ss[j])); CK(hipStreamWaitEvent(ss[0], ej[j], 0)); }
    };
    hipGraphExec_t ge = nullptr; hipGraph_t g = nullptr;
    if (graph) {
      if (hipStreamBeginCapture(ss[0], hipStreamCaptureModeThreadLocal) != hipSuccess) { printf("%-34s capture refused\n", nm.c_str()); (void)hipGetLastError(); return -1; }
      forward(false);
      hipError_t e = hipStreamEndCapture(ss[0], &g);
      if (e != hipSuccess || !g || hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("%-34s capture/instantiate failed: %s\n", nm.c_str(), hipGetErrorString(e)); (void)hipGetLastError(); return -1; }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto once = [&](bool stamp) { if (graph) CK(hipGraphLaunch(ge, ss[0])); else forward(stamp); };
    CK(hipMemset(tmo, 0, 16));
    once(false); once(false);
    CK(hipStreamSynchronize(ss[0]));
    CK(hipEventRecord(e0, ss[0]));
    for (int r = 0; r < REPS; ++r) once(false);
    CK(hipEventRecord(e1, ss[0]));
    CK(hipStreamSynchronize(ss[0]));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned t; CK(hipMemcpy(&t, tmo, 4, hipMemcpyDeviceToHost));
    // result check against the base variant's output
    std::vector<unsigned short> hx(2 * d), hr(2 * d);
    CK(hipMemcpy(hx.data(), x, 2 * d * 2, hipMemcpyDeviceToHost));
    static bool have_ref = false;
    if (!have_ref) { CK(hipMemcpy(xref, x, 2 * d * 2, hipMemcpyDeviceToDevice)); have_ref = true; }
    CK(hipMemcpy(hr.data(), xref, 2 * d * 2, hipMemcpyDeviceToHost));
    int diff = 0; for (int i = 0; i < 2 * d; ++i) diff += hx[i] != hr[i];
    const double us_layer = ms * 1000.0 / REPS / NL;
    printf("%-34s %8.2f us/layer  (%.3f ms per %d-layer forward)  timeouts=%u  mismatched outputs=%d/%d  x[0]=%04x\n", nm.c_str(), us_layer, ms / REPS, NL, t, diff, 2 * d, hx[0]);
    if (!graph && v.nstreams == 1 && !v.flags) {
      // per-kernel spans of one stamped eager forward (start of first block .. end of last block)
      std::vector<unsigned long long> init(2 * NK); for (int i = 0; i < NK; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; }
      CK(hipMemcpy(kst, init.data(), NK * 16, hipMemcpyHostToDevice));
      forward(true); CK(hipStreamSynchronize(ss[0]));
      std::vector<unsigned long long> h(2 * NK); CK(hipMemcpy(h.data(), kst, NK * 16, hipMemcpyDeviceToHost));
      const int l = NL / 2;
      printf("   stamps, layer %d (us): ", l);
      const char* nn[5] = {"in_proj", "out1", "out2", "fc1", "fc2"};
      for (int j = 0; j < 5; ++j) { const int k = l * 5 + j; printf("%s span %.2f gap-before %.2f | ", nn[j], (h[2 * k + 1] - h[2 * k]) / 100.0, (double)((long long)h[2 * k] - (long long)h[2 * k - 1]) / 100.0); }
      printf("\n");
    }
    if (ge) (void)hipGraphExecDestroy(ge);
    if (g) (void)hipGraphDestroy(g);
    return us_layer;
  };
  const Variant vs[] = {
    {"base 1 stream", 1, false, false, 0, 256},
    {"base 1 stream, 1024 blocks", 1, false, false, 0, 1024},
    {"flags 1 stream (order only)", 1, true, false, 1, 256},
    {"flags 2 streams", 2, true, false, 1, 256},
    {"flags 3 streams", 3, true, false, 1, 256},
    {"flags 2 streams, wait first", 2, true, false, 0, 256},
    {"flags any-order 1 stream", 1, true, true, 1, 256},
  };
  for (const auto& v : vs) {
    run_variant(v, false);
    run_variant(v, true);
  }
  return 0;
}
