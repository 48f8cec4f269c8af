// DAC decoder convolutions on the bf16 matrix cores with fp32-grade operands (zn_dac.hip's decode path).
//
// An fp32 value v is carried as three bf16 terms h + m + l (h = bf16(v), m = bf16(v - h), l = bf16(v - h - m): 24 significand
// bits, the differences are exact in fp32), and a product a * b as the six partial products whose magnitude is >= 2^-16 |a b|
// (hh, hm, mh, mm, hl, lh; the three dropped ones are <= 2^-24 |a b|, the size of an fp32 rounding), each exact in the matrix
// core and accumulated in fp32.  v_mfma_f32_32x32x16_bf16 is 16 x the rate of v_mfma_f32_32x32x2_f32, so six of them do the
// work of eight fp32 MFMAs in 3/8 of the issue time - and the Snake activation, which zn_conv_kernels.h applies to its
// input rows once per output-channel tile (6 - 12 times per element), moves to the PRODUCER's epilogue: a layer stores the
// next layer's input activated, once.
//
// Layouts.  Activations stay fp32 channels-last [B][T][C] in memory (4 bytes per element; a first version kept the three terms
// there, 6 bytes: the decode wrote 4.4 GB of them per clip and needed an LDS transpose in every epilogue) and are split on the way
// into LDS (8 VALU instructions per element and output-channel tile).  Weights are split once: "W3" [phase][Cin/16][tap][CoutPad][3][16].
// The residual stream a unit adds back at its end (DacResidualUnit, modeling_dac.py:175-209) is a second fp32 tensor.
// Workgroup: 256 threads, RB x 128 output times x 64 output channels; wave = RB x 32 rows x 64 channels; K advances 16 input
// channels per stage: the rows a tile needs (with the dilation halo) and all taps of the stage's weights sit in LDS as three-term
// rows of 96 bytes (the two 16-byte halves of a term swapped on rows with bit 3 set: conflict-free ds_read_b128 fragments without
// padding; 71 KB at RB = 2, so two workgroups share a CU and one stages while the other multiplies).
#pragma once
#include "zn_conv_kernels.h"

typedef __attribute__((ext_vector_type(8))) __bf16 c3_bf16x8;

#define C3_KC 16
#define C3_TN 64
#define C3_ROW 96                      // LDS / S3 bytes per (row, 16-channel chunk): 3 terms x 16 bf16
#ifndef C3_NFAST
#define C3_NFAST 1                     // 1: output-channel tiles vary fastest in the dispatch order (workgroups of one time tile run together)
#endif
#ifndef C3_K1_RB
#define C3_K1_RB 1                     // row blocks per wave of the 1-tap (pointwise) layers
#endif

struct Conv3Args {
  const float* in; int Tin, Cin;           // fp32 [B][Tin][Cin]: already activated (or the raw latent)
  const bf16_t* w;                         // W3 [phase][Cin/16][tap][CoutPad][3][16]
  const float* bias;                       // [Cout]
  const float* alpha;                      // Snake alpha of the NEXT layer (over this layer's output channels), or NULL: out_act holds the plain output
  const float* skip;                       // fp32 residual [B][Tout][Cout], or NULL
  float* out32;                            // fp32 output [B][Tout][Cout] before the next Snake (the residual stream), or NULL; may alias skip
  float* out_act;                          // fp32 output after the next Snake (the next layer's input), or NULL
  int Tout, Cout, CoutPad;
  int M;                                   // GEMM rows per phase
  int taps, off0, offstep;                 // input row of GEMM row m, tap k: m + off0 + k*offstep
  int ostride, ooff, phases;               // output time of row m in phase p: m*ostride + ooff + p
};

ZN_DEVINL void c3_split(float v, bf16_t& h, bf16_t& m, bf16_t& l) {
  h = f2bf(v);
  const float r1 = v - bf2f(h);            // exact
  m = f2bf(r1);
  l = f2bf(r1 - bf2f(m));                  // exact difference, one rounding
}
ZN_DEVINL int c3_swz(int row, int term, int half) { return row * C3_ROW + term * 32 + ((half ^ ((row >> 3) & 1)) << 4); }

// sin(a)^2 for Snake (modeling_dac.py:98) in 13 instructions: a = k pi + r (pi in two fp32 terms, fused), sin(r) = r g(r^2) on
// [-pi/2, pi/2] (degree-5 fit in r^2); the sign of sin drops out of the square.  Against sin^2 in fp64 over |a| <= 50: max error
// 2.9e-7, rms 4.7e-8 (sinf of the device library squared: 1.3e-7, 2.9e-8).  |a| > 2^16 (never seen; the reduction would lose bits)
// takes the library's sinf.
ZN_DEVINL float c3_sin2(float a) {
  if (!(fabsf(a) <= 65536.0f)) { const float s = sinf(a); return s * s; }
  const float k = rintf(a * 0.318309886183790672f);
  float r = fmaf(k, -3.1415927410125732f, a);
  r = fmaf(k, 8.742277657347586e-08f, r);
  const float u = r * r;
  float g = -2.3866771670e-08f;
  g = fmaf(g, u, 2.7524013149e-06f);
  g = fmaf(g, u, -1.9840836467e-04f);
  g = fmaf(g, u, 8.3333309740e-03f);
  g = fmaf(g, u, -1.6666667163e-01f);
  g = fmaf(g, u, 1.0f);
  const float sn = r * g;
  return sn * sn;
}

// RB: 32-row blocks per wave (workgroup = RB * 128 output times); MAXT: taps the launch may have (7: the dilated 7-tap layers, halo
// of up to 54 rows; 2: the two taps of a transposed convolution's phase; 1: the pointwise layers - small staging, several workgroups
// per CU hide one another's per-stage latency).
template <int RB, int MAXT>
__global__ __launch_bounds__(256, 2) void dac_conv3_kernel(Conv3Args a) {
  constexpr int TM = RB * 128, NT = C3_TN / 32, HALO = MAXT == 7 ? 54 : (MAXT == 2 ? 1 : 0), MAXROWS = TM + HALO;
  extern __shared__ __attribute__((aligned(16))) unsigned char c3_smem[];
  unsigned char* s_in = c3_smem;                            // [MAXROWS][96]
  unsigned char* s_w = c3_smem + MAXROWS * C3_ROW;          // [taps][64][96]
  const int m0 = (C3_NFAST ? blockIdx.y : blockIdx.x) * TM, n0 = (C3_NFAST ? blockIdx.x : blockIdx.y) * C3_TN;
  const int b = blockIdx.z / a.phases, phase = blockIdx.z % a.phases;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int offlast = a.off0 + (a.taps - 1) * a.offstep;
  const int offmin = a.off0 < offlast ? a.off0 : offlast, offmax = a.off0 < offlast ? offlast : a.off0;
  const int nrows = TM + offmax - offmin;
  const int nck = a.Cin / C3_KC;
  const float* inb = a.in + (size_t)b * a.Tin * a.Cin;
  const bf16_t* wp = a.w + (size_t)phase * nck * a.taps * a.CoutPad * 48;
  f32x16 acc[RB][NT];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[rb][nt][e] = 0.f;

  // staging slots: the next stage's requests go out before this stage's MFMAs and land in LDS after them.  Input pieces: 16 bytes = 4
  // fp32 channels, 4 per row and stage; weight pieces: 16 bytes of a term, 6 per weight row.  Buffer loads: one 32-bit offset per
  // piece (rows outside [0, Tin) fall outside the batch element's range and read as zeros: the convolution's zero padding), the K stage
  // advances through the scalar offset.
  constexpr int IN_P = (MAXROWS * 4 + 255) / 256, W_P = (MAXT * C3_TN * 6 + 255) / 256;
  u32x4 rin[IN_P], rw[W_P];
  int vin[IN_P], vw[W_P];
  const int n_in = nrows * 4, n_w = a.taps * C3_TN * 6;
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)inb, 0, (int)((size_t)a.Tin * a.Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, (int)((size_t)nck * a.taps * a.CoutPad * C3_ROW), 0x00020000);
#pragma unroll
  for (int j = 0; j < IN_P; ++j) {
    const int i = tid + j * 256, row = i >> 2, q = i & 3;
    vin[j] = i < n_in ? (m0 + offmin + row) * (a.Cin * 4) + q * 16 : (int)0x80000000;
  }
#pragma unroll
  for (int j = 0; j < W_P; ++j) {
    const int i = tid + j * 256, rowi = i / 6, q = i - rowi * 6;         // rowi = tap * 64 + n
    const int tap = rowi / C3_TN, n = rowi - tap * C3_TN;
    vw[j] = i < n_w ? (tap * a.CoutPad + n0 + n) * C3_ROW + q * 16 : (int)0x80000000;
  }
  auto fetch = [&](int ck) {
#pragma unroll
    for (int j = 0; j < IN_P; ++j) rin[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, vin[j], ck * (C3_KC * 4), 0);
#pragma unroll
    for (int j = 0; j < W_P; ++j) rw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, vw[j], ck * a.taps * a.CoutPad * C3_ROW, 0);
  };
  fetch(0);
  const int fr = lane & 31, fh = lane >> 5;                  // fragment row / column, K half
  for (int ck = 0; ck < nck; ++ck) {
    __syncthreads();                                     // the previous stage's fragment reads are done
#pragma unroll
    for (int j = 0; j < IN_P; ++j) {
      const int i = tid + j * 256, row = i >> 2, q = i & 3;
      if (i < n_in) {                                    // channels 4 q .. 4 q + 3 of the stage: split, 8 bytes into each term's row
        const float f[4] = {__uint_as_float(rin[j].x), __uint_as_float(rin[j].y), __uint_as_float(rin[j].z), __uint_as_float(rin[j].w)};
        bf16_t t[3][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) c3_split(f[e], t[0][e], t[1][e], t[2][e]);
        unsigned char* d = s_in + c3_swz(row, 0, q >> 1) + (q & 1) * 8;
#pragma unroll
        for (int p3 = 0; p3 < 3; ++p3)
          *(u32x2*)(d + p3 * 32) = u32x2{(unsigned)t[p3][0] | ((unsigned)t[p3][1] << 16), (unsigned)t[p3][2] | ((unsigned)t[p3][3] << 16)};
      }
    }
#pragma unroll
    for (int j = 0; j < W_P; ++j) {
      const int i = tid + j * 256, rowi = i / 6, q = i - rowi * 6;
      if (i < n_w) *(u32x4*)(s_w + (rowi / C3_TN) * (C3_TN * C3_ROW) + c3_swz(rowi % C3_TN, q >> 1, q & 1)) = rw[j];
    }
    __syncthreads();
    if (ck + 1 < nck) fetch(ck + 1);
#ifndef C3_TAP_UNROLL
#define C3_TAP_UNROLL 1
#endif
#pragma unroll C3_TAP_UNROLL
    for (int tap = 0; tap < MAXT; ++tap) {                  // a.taps == MAXT (zn_conv3_launch)
      const int r0 = wave * (32 * RB) + fr + a.off0 + tap * a.offstep - offmin;
      c3_bf16x8 A[RB][3], Bf[NT][3];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int p = 0; p < 3; ++p) A[rb][p] = *(const c3_bf16x8*)(s_in + c3_swz(r0 + rb * 32, p, fh));
      const unsigned char* wt = s_w + tap * (C3_TN * C3_ROW);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < 3; ++p) Bf[nt][p] = *(const c3_bf16x8*)(wt + c3_swz(nt * 32 + fr, p, fh));
      // the six partial products, smallest first; consecutive MFMAs go to different accumulators
      constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
      for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[rb][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[rb][PA[t6]], Bf[nt][PB[t6]], acc[rb][nt], 0, 0, 0);
    }
  }
  // ---- epilogue.  C layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5): a lane holds 16 times of one channel, a store
  // instruction covers 32 consecutive channels of two rows.  bias, residual, the fp32 residual stream; the next Snake; the next layer's input.
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = n0 + nt * 32 + fr;
      if (co >= a.Cout) continue;
      const float bv = a.bias ? a.bias[co] : 0.f;
      const float al = a.alpha ? a.alpha[co] : 1.f;
      const float ial = 1.0f / (al + 1e-9f);
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int rowi = (reg & 3) + 8 * (reg >> 2) + 4 * fh;
        const int m = m0 + wave * (32 * RB) + rb * 32 + rowi;
        const int to = m * a.ostride + a.ooff + phase;
        if (m >= a.M || to < 0 || to >= a.Tout) continue;
        const size_t o = ((size_t)b * a.Tout + to) * a.Cout + co;
        float v = acc[rb][nt][reg] + bv;
        if (a.skip) v = a.skip[o] + v;
        if (a.out32) a.out32[o] = v;
        if (a.out_act) a.out_act[o] = a.alpha ? v + ial * c3_sin2(al * v) : v;
      }
    }
  }
}

// conv weight [Cout][Cin][K] -> W3 [Cin/16][K][CoutPad][3][16]
__global__ void dac_w3conv_kernel(const float* w, bf16_t* o, int Cout, int Cin, int K, int CoutPad) {
  const size_t n = (size_t)K * Cin * CoutPad;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = i % Cin, co = (i / Cin) % CoutPad, k = i / ((size_t)Cin * CoutPad);
    bf16_t h, m, l;
    c3_split(co < Cout ? w[((size_t)co * Cin + ci) * K + k] : 0.f, h, m, l);
    bf16_t* d = o + ((((size_t)(ci >> 4)) * K + k) * CoutPad + co) * 48 + (ci & 15);
    d[0] = h; d[16] = m; d[32] = l;
  }
}
// conv-transpose weight [Cin][Cout][2s] -> W3 [phase p][Cin/16][tap j][CoutPad][3][16], tap 0 <-> k = p (input q), tap 1 <-> k = p + s (input q-1)
__global__ void dac_w3convt_kernel(const float* w, bf16_t* o, int Cin, int Cout, int s, int CoutPad) {
  const size_t n = (size_t)s * 2 * Cin * CoutPad;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = i % Cin, co = (i / Cin) % CoutPad, j = (i / ((size_t)Cin * CoutPad)) % 2, p = i / ((size_t)Cin * CoutPad * 2);
    bf16_t h, m, l;
    c3_split(co < Cout ? w[((size_t)ci * Cout + co) * (2 * s) + p + j * s] : 0.f, h, m, l);
    bf16_t* d = o + (((((size_t)p * (Cin >> 4)) + (ci >> 4)) * 2 + j) * CoutPad + co) * 48 + (ci & 15);
    d[0] = h; d[16] = m; d[32] = l;
  }
}

template <int RB, int MAXT> static inline size_t zn_conv3_lds() {
  return (size_t)(RB * 128 + (MAXT == 7 ? 54 : (MAXT == 2 ? 1 : 0))) * C3_ROW + (size_t)MAXT * C3_TN * C3_ROW;
}
template <int RB, int MAXT> static inline hipError_t zn_conv3_attr() {
  return hipFuncSetAttribute((const void*)dac_conv3_kernel<RB, MAXT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)zn_conv3_lds<RB, MAXT>());
}
static inline hipError_t zn_conv3_set_attrs() {
  hipError_t e;
  if ((e = zn_conv3_attr<1, 7>()) != hipSuccess || (e = zn_conv3_attr<2, 7>()) != hipSuccess || (e = zn_conv3_attr<1, 2>()) != hipSuccess ||
      (e = zn_conv3_attr<2, 2>()) != hipSuccess || (e = zn_conv3_attr<C3_K1_RB, 1>()) != hipSuccess) return e;
  return hipSuccess;
}
static inline int zn_conv3_pad(int c) { return (c + C3_TN - 1) / C3_TN * C3_TN; }
template <int RB, int MAXT> static inline void zn_conv3_go(const Conv3Args& a, int B, hipStream_t s) {
  const int ntile = a.CoutPad / C3_TN, mt = (a.M + RB * 128 - 1) / (RB * 128);
  const dim3 grid = C3_NFAST ? dim3(ntile, mt, B * a.phases) : dim3(mt, ntile, B * a.phases);
  const size_t lds = zn_conv3_lds<RB, MAXT>();
  hipLaunchKernelGGL((dac_conv3_kernel<RB, MAXT>), grid, dim3(256), lds, s, a);
}
// 256-row tiles once they fill the chip's 512 workgroup slots, 128-row tiles for the short early layers
static inline void zn_conv3_launch(const Conv3Args& a, int B, hipStream_t s) {
  // (the decoder's layers have 7, 2 or 1 taps; the kernels' tap loops are compile-time)
  if (a.taps == 1) return zn_conv3_go<C3_K1_RB, 1>(a, B, s);
  const bool big = (long)((a.M + 255) / 256) * (a.CoutPad / C3_TN) * B * a.phases >= 512;
  if (a.taps == 2) { if (big) zn_conv3_go<2, 2>(a, B, s); else zn_conv3_go<1, 2>(a, B, s); return; }
  if (big) zn_conv3_go<2, 7>(a, B, s); else zn_conv3_go<1, 7>(a, B, s);
}
