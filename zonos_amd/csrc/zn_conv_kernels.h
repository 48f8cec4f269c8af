// Implicit-GEMM 1-D convolution on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32), shared by the DAC codec
// (zn_dac.hip) and the speaker-embedding network (zn_spk.hip).  Activations are channels-last [B][T][C]; a workgroup
// owns a 128-time x (NT*32)-channel output tile, stages the input rows it needs (with the tap halo) once per 16-channel
// chunk in LDS (Snake activation applied on the way in when alpha is given) next to all taps of the chunk's weights.
// GEMM row m, tap k reads input row m + off0 + k*offstep (zero outside [0, Tin)); row m of phase p writes output time
// m*ostride + ooff + p.  Batch elements may be strided (in_bs / out_bs elements apart; 0 = densely packed), which lets
// a 2-D convolution run as three row-shifted 1-D passes over image rows.
#pragma once
#include "zn_common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

#define DAC_KC 16
#define DAC_TM 128
#define DAC_MAXTAPS 7
#define DAC_MAXROWS (DAC_TM + 6 * 9)

struct ConvArgs {
  const float* in; int Tin, Cin;          // [B][Tin][Cin]
  const float* w;                          // [phase][tap][Cin][CoutPad]
  const float* bias;                       // [Cout]
  const float* alpha;                      // Snake alpha of the input channels, or NULL
  const float* skip;                       // residual [B][Tout][Cout], or NULL
  float* out; int Tout, Cout, CoutPad;     // [B][Tout][Cout]
  int M;                                   // GEMM rows per phase
  int taps, off0, offstep;                 // input row of GEMM row m, tap k: m + off0 + k*offstep
  int ostride, ooff, phases;               // output time of row m in phase p: m*ostride + ooff + p
  long long in_bs, out_bs;                 // batch strides in elements (0: Tin*Cin / Tout*Cout); skip shares out_bs
  int relu;                                // 1: max(0, .) after bias and skip
};

__device__ __forceinline__ float snake_f(float x, float alpha) {
  // modeling_dac.py:98: x + (alpha + 1e-9)^-1 * sin(alpha x)^2   (accurate sinf, IEEE reciprocal)
  const float s = sinf(alpha * x);
  return x + (1.0f / (alpha + 1e-9f)) * (s * s);
}

template <int NT>
__global__ __launch_bounds__(256) void dac_conv_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TN = NT * 32;
  float* s_in = smem;                                   // [rows][KC+1]
  float* s_w = smem + DAC_MAXROWS * (DAC_KC + 1);       // [taps*KC][TN]
  const int m0 = blockIdx.x * DAC_TM, n0 = blockIdx.y * TN;
  const int b = blockIdx.z / a.phases, phase = blockIdx.z % a.phases;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int offlast = a.off0 + (a.taps - 1) * a.offstep;
  const int offmin = a.off0 < offlast ? a.off0 : offlast, offmax = a.off0 < offlast ? offlast : a.off0;
  const int nrows = DAC_TM + offmax - offmin;
  const float* inb = a.in + (size_t)b * (a.in_bs ? (size_t)a.in_bs : (size_t)a.Tin * a.Cin);
  const float* wp = a.w + (size_t)phase * a.taps * a.Cin * a.CoutPad;
  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

  // Per-thread staging slots: the next chunk's global requests are issued before this chunk's MFMAs and written to LDS
  // (Snake applied on the way) after them, so that the HBM/L2 latency of a chunk hides behind the previous chunk's math.
  constexpr int IN_P = (DAC_MAXROWS * (DAC_KC / 4) + 255) / 256;             // input pieces (f32x4) per thread
  constexpr int W_P = (DAC_MAXTAPS * DAC_KC * (TN / 4) + 255) / 256;         // weight pieces per thread
  f32x4 rin[IN_P], ral[IN_P], rw[W_P];
  const int n_in = nrows * (DAC_KC / 4), n_w = a.taps * DAC_KC * (TN / 4);
  auto fetch = [&](int c0) {
#pragma unroll
    for (int j = 0; j < IN_P; ++j) {
      const int i = tid + j * 256;
      const int row = i / (DAC_KC / 4), c4 = (i % (DAC_KC / 4)) * 4;
      int t = m0 + offmin + row;
      t = t < 0 ? 0 : (t >= a.Tin ? a.Tin - 1 : t);                          // clamped; validity is re-derived when storing
      if (i < n_in) {
        rin[j] = *(const f32x4*)(inb + (size_t)t * a.Cin + c0 + c4);
        if (a.alpha) ral[j] = *(const f32x4*)(a.alpha + c0 + c4);
      }
    }
#pragma unroll
    for (int j = 0; j < W_P; ++j) {
      const int i = tid + j * 256;
      const int rowi = i / (TN / 4), c4 = (i % (TN / 4)) * 4;
      const int tap = rowi / DAC_KC, ci = rowi % DAC_KC;
      if (i < n_w) rw[j] = *(const f32x4*)(wp + ((size_t)tap * a.Cin + c0 + ci) * a.CoutPad + n0 + c4);
    }
  };
  fetch(0);
  for (int c0 = 0; c0 < a.Cin; c0 += DAC_KC) {
    __syncthreads();                                     // the previous chunk's fragment reads are done
    // input rows [m0+offmin, m0+offmin+nrows) x KC channels, Snake on the way in (zero outside [0,Tin))
#pragma unroll
    for (int j = 0; j < IN_P; ++j) {
      const int i = tid + j * 256;
      if (i < n_in) {
        const int row = i / (DAC_KC / 4), c4 = (i % (DAC_KC / 4)) * 4;
        const int t = m0 + offmin + row;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t >= 0 && t < a.Tin) {
          v = rin[j];
          if (a.alpha) { v.x = snake_f(v.x, ral[j].x); v.y = snake_f(v.y, ral[j].y); v.z = snake_f(v.z, ral[j].z); v.w = snake_f(v.w, ral[j].w); }
        }
        float* d = s_in + row * (DAC_KC + 1) + c4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    }
    // weights: taps x KC rows of TN output channels
#pragma unroll
    for (int j = 0; j < W_P; ++j) {
      const int i = tid + j * 256;
      if (i < n_w) { const int rowi = i / (TN / 4), c4 = (i % (TN / 4)) * 4; *(f32x4*)(s_w + (size_t)rowi * TN + c4) = rw[j]; }
    }
    __syncthreads();
    if (c0 + DAC_KC < a.Cin) fetch(c0 + DAC_KC);
    const int ai = lane & 31, ak = lane >> 5;
    for (int tap = 0; tap < a.taps; ++tap) {
      const float* arow = s_in + (wave * 32 + ai + a.off0 + tap * a.offstep - offmin) * (DAC_KC + 1) + ak;
      const float* brow = s_w + (size_t)(tap * DAC_KC + ak) * TN + ai;
#pragma unroll
      for (int k = 0; k < DAC_KC; k += 2) {
        const float av = arow[k];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, brow[(size_t)k * TN + nt * 32], acc[nt], 0, 0, 0);
      }
    }
  }
  // epilogue: bias, residual, store.  C layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int col = lane & 31;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = n0 + nt * 32 + col;
    if (co >= a.Cout) continue;
    const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      const int m = m0 + wave * 32 + row;
      if (m >= a.M) continue;
      const int to = m * a.ostride + a.ooff + phase;
      if (to < 0 || to >= a.Tout) continue;
      const size_t o = (size_t)b * (a.out_bs ? (size_t)a.out_bs : (size_t)a.Tout * a.Cout) + (size_t)to * a.Cout + co;
      float v = acc[nt][reg] + bv;
      if (a.skip) v = a.skip[o] + v;
      if (a.relu) v = fmaxf(v, 0.f);
      a.out[o] = v;
    }
  }
}


// ------------------------------------------------------------------------------------------------ host helpers
// output channels are tiled by 128 (NT = 4), by 96 where that divides evenly (NT = 3) or by 64 (NT = 2)
static inline int zn_conv_pad(int c) { return c % 128 == 0 ? c : (c % 96 == 0 ? c : (c % 64 == 0 ? c : (c + 127) / 128 * 128)); }
static inline int zn_conv_nt(int coutpad) { return coutpad % 128 == 0 ? 4 : (coutpad % 96 == 0 ? 3 : 2); }
static inline hipError_t zn_conv_set_attrs() {
  for (int nt : {2, 3, 4}) {
    const int bytes = (DAC_MAXROWS * (DAC_KC + 1) + DAC_MAXTAPS * DAC_KC * nt * 32) * (int)sizeof(float);
    hipError_t e = nt == 2 ? hipFuncSetAttribute((const void*)dac_conv_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)
                 : nt == 3 ? hipFuncSetAttribute((const void*)dac_conv_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)
                           : hipFuncSetAttribute((const void*)dac_conv_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
static inline void zn_conv_launch(const ConvArgs& a, int coutpad, int B, hipStream_t s) {
  const int nt = zn_conv_nt(coutpad);
  const int TN = nt * 32;
  dim3 grid((a.M + DAC_TM - 1) / DAC_TM, coutpad / TN, B * a.phases);
  const size_t lds = (size_t)(DAC_MAXROWS * (DAC_KC + 1) + DAC_MAXTAPS * DAC_KC * TN) * sizeof(float);
  if (nt == 4) hipLaunchKernelGGL((dac_conv_kernel<4>), grid, dim3(256), lds, s, a);
  else if (nt == 3) hipLaunchKernelGGL((dac_conv_kernel<3>), grid, dim3(256), lds, s, a);
  else hipLaunchKernelGGL((dac_conv_kernel<2>), grid, dim3(256), lds, s, a);
}
