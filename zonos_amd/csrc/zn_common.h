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
// The same through a GLOBAL-address-space pointer, for weights whose address was read from a device table: loaded through a
// generic pointer they become flat_load instructions, which count on lgkmcnt as well as vmcnt - every wait for an LDS read then
// also waits for every weight tile still in flight (found in the whole-step kernels' ISA: 201 flat loads).
__device__ __forceinline__ u32x4 ld_nt16g(const void* p) { return __builtin_nontemporal_load((const __attribute__((address_space(1))) u32x4*)p); }
__device__ __forceinline__ u32x4 ld16g(const void* p) { return *(const __attribute__((address_space(1))) u32x4*)p; }

// ---- cross-lane reductions on the VALU (DPP within a 16-lane row, v_permlane16/32_swap across rows): no LDS
// round trips (ds_bpermute chains made the first version of these kernels latency-bound).
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
#define ZN_DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define ZN_DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define ZN_DPP_HALF_MIRROR 0x141  // lane i <-> 7-i inside each 8 lanes
#define ZN_DPP_MIRROR 0x140       // lane i <-> 15-i inside each row of 16
#define ZN_DPP_ROR4 0x124
#define ZN_DPP_ROR8 0x128
__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// reduce over groups of WIDTH consecutive lanes (WIDTH in {4,8,16}); every lane of the group gets the result
template <int WIDTH> __device__ __forceinline__ float group_sum(float v) {
  static_assert(WIDTH == 4 || WIDTH == 8 || WIDTH == 16, "row-local reduction");
  v += dpp_mov<ZN_DPP_XOR1>(v);
  v += dpp_mov<ZN_DPP_XOR2>(v);
  if (WIDTH >= 8) v += dpp_mov<ZN_DPP_HALF_MIRROR>(v);
  if (WIDTH >= 16) v += dpp_mov<ZN_DPP_MIRROR>(v);
  return v;
}
template <int WIDTH> __device__ __forceinline__ float group_max(float v) {
  static_assert(WIDTH == 4 || WIDTH == 8 || WIDTH == 16, "row-local reduction");
  v = fmaxf(v, dpp_mov<ZN_DPP_XOR1>(v));
  v = fmaxf(v, dpp_mov<ZN_DPP_XOR2>(v));
  if (WIDTH >= 8) v = fmaxf(v, dpp_mov<ZN_DPP_HALF_MIRROR>(v));
  if (WIDTH >= 16) v = fmaxf(v, dpp_mov<ZN_DPP_MIRROR>(v));
  return v;
}
// whole-wave reductions: row totals by DPP, then the four rows through v_readlane (wave-uniform result)
__device__ __forceinline__ float wave_sum(float v) {
  v = group_sum<16>(v);
  return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = group_max<16>(v);
  return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}
// fp64 variants (the exact-accumulation GEMV experiment, ZN_GEMV_F64): a double moves as two DPP / readlane halves
template <int CTRL> __device__ __forceinline__ double dpp_mov_d(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double readlane_d(double v, int lane) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {
  v += dpp_mov_d<ZN_DPP_XOR1>(v);
  v += dpp_mov_d<ZN_DPP_XOR2>(v);
  v += dpp_mov_d<ZN_DPP_HALF_MIRROR>(v);
  v += dpp_mov_d<ZN_DPP_MIRROR>(v);
  return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}
// sum over the lanes of one 16-lane row that share (lane % 4)
__device__ __forceinline__ float row_stride4_sum(float v) {
  v += dpp_mov<ZN_DPP_ROR4>(v);
  v += dpp_mov<ZN_DPP_ROR8>(v);
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
