// Shared device helpers for the gfx950 Zonos kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(2))) __bf16 zn_bf16x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define ZN_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((unsigned)u) << 16); }
// plain cast -> v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN-preserving; MI355X_MICROARCH.md correctness table)
__device__ __forceinline__ bf16_t f2bf(float f) { __bf16 b = (__bf16)f; return __builtin_bit_cast(unsigned short, b); }
__device__ __forceinline__ float bfround(float f) { return bf2f(f2bf(f)); }
__device__ __forceinline__ float lo_f(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_f(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ unsigned pack2(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }

// v_dot2c_f32_bf16: c + a.lo*b.lo + a.hi*b.hi, fp32 accumulate of exact bf16 products
__device__ __forceinline__ float dot2(unsigned a, unsigned b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(zn_bf16x2, a), __builtin_bit_cast(zn_bf16x2, b), c, false);
}
__device__ __forceinline__ float dot8(const u32x4& w, const u32x4& x, float c) {
  c = dot2(w.x, x.x, c); c = dot2(w.y, x.y, c); c = dot2(w.z, x.z, c); return dot2(w.w, x.w, c);
}

// streamed-once weights: non-temporal 16-B load (MI355X_MICROARCH.md "nt-weights")
__device__ __forceinline__ u32x4 ld_nt16(const void* p) { return __builtin_nontemporal_load((const u32x4*)p); }
__device__ __forceinline__ u32x4 ld16(const void* p) { return *(const u32x4*)p; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// reduce over groups of `width` consecutive lanes (width power of two <= 64)
template <int WIDTH> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- fexp_u20: the fast exponential torch's CPU flash-attention applies to full SIMD vectors of the softmax
// numerator when the value dtype is bf16 (ATen/cpu/vec/vec512/vec512_float.h fexp_u20: Malossi et al.,
// degree-3 polynomial, truncating float->int cast).  Mean relative error 5e-5, so the bf16-rounded
// probabilities the reference feeds to P.V differ from an accurate exp in ~0.8 % of entries; the oracle is the
// reference CPU path, so the kernels reproduce it bit for bit (explicit _rn ops: no contraction).
__device__ __forceinline__ float zn_fexp_u20(float x) {
  const float log2e = __uint_as_float(0x3fb8aa3bu);
  const float c0 = 0.00010703434948458272f, c1 = 0.30354260500649682f;
  const float c2 = (float)-0.22433836478672356, c3 = (float)-0.079204240219773236;
  float src = __fmul_rn(x, log2e);
  float frac = __fsub_rn(src, floorf(src));
  float r = __fmaf_rn(frac, c3, c2);
  r = __fmaf_rn(frac, r, c1);
  r = __fmaf_rn(frac, r, c0);
  src = __fsub_rn(src, r);
  float tmp = __fmaf_rn(8388608.0f, src, 1065353216.0f);
  int i = (int)tmp;  // cvtt: truncation
  if (!(x >= __uint_as_float(0xc2aeac50u))) i = 0;  // x < ln(FLT_MIN) (and NaN from -inf inputs) -> 0
  return __int_as_float(i);
}

#define ZN_DEVINL __device__ __forceinline__
