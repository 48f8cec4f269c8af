// Decode-step kernels (S = 1, R = 2B rows): weight-streaming bf16 GEMV with fused prologues/epilogues (R <= 4), small-M
// MFMA projections for 5..16 rows (direct-fragment and LDS-staged), KV-cached GQA attention that reproduces the
// reference CPU flash-attention rounding points (two-pass, fused single launch for short contexts, per-block split for
// long ones), embedding sum, sampler and frame bookkeeping.  All HBM-bound: weights are read once per step with 16-B
// non-temporal loads, activations (a few KB) live in registers or LDS, reductions are wavefront shuffles.
#pragma once
#include "zn_common.h"

// ------------------------------------------------------------------------------------------------ GEMV
enum { PRO_NONE = 0, PRO_LN = 1, PRO_GATED = 2 };
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_SILU = 2, EPI_ROPE_KV = 3, EPI_F32 = 4, EPI_MAMBA = 5 };

struct GemvArgs {
  const bf16_t* W;  // [N][K]
  int N, K, nrows;
  int units;        // work units (a unit = two weight rows)
  int upw;          // units per wave (KSPLIT=1) / per block (KSPLIT=4)
  // prologue
  const bf16_t* x;  // bf16 [rows][K]
  const bf16_t *ln_w, *ln_b;
  float eps;
  const float* gv;     // PRO_GATED (Mamba2 RMSNormGated, one group): fp32 [rows][K] gated values y * silu(z) in place of x;
                       // the prologue adds the row's RMS statistic and the weight ln_w
  const int* lengths;  // int32 [rows]: keys already in the cache (position of the new token)
  int hd, n_heads;
  // epilogue
  bf16_t* out;           // bf16 [rows][N] (EPI_SILU: [rows][N/2])
  const bf16_t* resid;   // EPI_RESID
  const bf16_t* bias;    // EPI_STORE: optional nn.Linear bias, added in fp32 before the single bf16 rounding
  float* out_f32;        // EPI_F32 [rows][N]
  // EPI_ROPE_KV
  bf16_t* q_out;         // [rows][Hq*hd]
  bf16_t* kv;            // [rows][max_len][2][Hkv][hd]
  const float* rope;     // [positions][hd/2][2]
  int max_len, n_heads_kv, rope_positions;
  // gemm16s_kernel (K split over workgroups): fp32 partial tiles [ksplit][16][N] and one arrival ticket per row group
  float* part;
  int* tickets;
  int ksplit;
  // EPI_MAMBA (Mamba2 in_proj, out = [z | xBC | dt] bf16 [rows][N]): weight rows [d_inner, d_inner + conv_dim) also go
  // through the causal-conv window update (causal_conv1d_update, width 4) + SiLU
  bf16_t* conv_state;      // [rows][conv_dim][4]
  const bf16_t* conv_w;    // [conv_dim][4]
  const bf16_t* conv_b;    // [conv_dim]
  bf16_t* xbc;             // [rows][conv_dim] activated conv output
  int d_inner, conv_dim;
  // gemm16k_kernel<EPI_ROPE_KV> over the rows of a short prefill (pf_S > 0): activation row m = r * pf_S + s is position pf_base + s of
  // cache row r (lengths unused); q goes to q_out [m][Hq * hd]
  int pf_S, pf_base;
  // LayerNorm statistics handed from the producer of a residual stream to the consumer that normalises it (rows 5..16): gemm16k_kernel<EPI_RESID>
  // leaves, per activation row and 16-column tile of its bf16 output, {sum, centred second moment about the tile's mean} (ln_part_out
  // [rows][N / 16][2] fp32); gemm16s_kernel<EPI_SILU, ., true> (ln_part_in, ln_tiles = K / 16) adds the tiles in a fixed order - the row
  // mean from the sums, then sum_t M2_t + 16 (mean_t - mean)^2, which is the centred second moment of the row exactly - and normalises
  // its activation chunks while staging them, in place of a layernorm_kernel launch in between.
  float* ln_part_out;
  const float* ln_part_in;
};

// EPI_MAMBA operands of activation row r and the weight-row pair starting at rowA (both rows lie in the same segment:
// d_inner and conv_dim are even): window state and taps of the two channels, their biases.  Requested ahead of the epilogue.
ZN_DEVINL void mamba_epi_operands(const GemvArgs& a, int r, int rowA, unsigned& cb, u32x4& cst, u32x4& cw) {
  const int c = rowA - a.d_inner;
  if (c >= 0 && c < a.conv_dim) {
    cst = ld16(a.conv_state + ((size_t)r * a.conv_dim + c) * 4);
    cw = ld16(a.conv_w + (size_t)c * 4);
    cb = *(const unsigned*)(a.conv_b + c);
  }
}
// one channel of causal_conv1d_update: window (e0 e1 | e2 e3) <- (e1 e2 | e3 x); out = silu(bias + sum_i w[i] * win[i]) with
// separately rounded fp32 multiplies and adds in tap order
ZN_DEVINL float mamba_conv_channel(unsigned& s0, unsigned& s1, unsigned w0, unsigned w1, float bias, bf16_t xn) {
  const float e1 = hi_f(s0), e2 = lo_f(s1), e3 = hi_f(s1), e4 = bf2f(xn);
  s0 = (s0 >> 16) | (s1 << 16);
  s1 = (s1 >> 16) | ((unsigned)xn << 16);
  float acc = bias;
  acc = __fadd_rn(acc, __fmul_rn(lo_f(w0), e1));
  acc = __fadd_rn(acc, __fmul_rn(hi_f(w0), e2));
  acc = __fadd_rn(acc, __fmul_rn(lo_f(w1), e3));
  acc = __fadd_rn(acc, __fmul_rn(hi_f(w1), e4));
  return acc / (1.0f + expf(-acc));
}

// Interleaved-pair rotation in fp32 (_torch.py:57-68): four products and two sums, each rounded on its own.  HIP's __fmul_rn /
// __fadd_rn are plain operators, which the compiler may fuse into an fma differently from kernel to kernel (it did: one q value
// in ~10^5 differed by an ulp between two kernels inlining the same expression); contraction is switched off here instead.
ZN_DEVINL void zn_rope_pair(float x0, float x1, float cs, float sn, float& re, float& im) {
#pragma clang fp contract(off)
  const float a = x0 * cs, b = x1 * sn, c = x1 * cs, d = x0 * sn;
  re = a - b;
  im = c + d;
}

// Fused epilogues shared by the GEMV (rows <= 4) and the small-M MFMA kernel (rows <= 16): finishes activation row r
// for the weight-row pair (rowA, rowB) of work unit u.
template <int EPI>
ZN_DEVINL void gemv_epilogue(const GemvArgs& a, int r, int rowA, int rowB, bool b_ok, int u, float vA, float vB, unsigned resid, float cs,
                             float sn, int pos, u32x4 cst = u32x4{0, 0, 0, 0}, u32x4 cw = u32x4{0, 0, 0, 0}) {
  if constexpr (EPI == EPI_STORE) {
    if (a.bias) { vA += bf2f(a.bias[rowA]); if (b_ok) vB += bf2f(a.bias[rowB]); }
    if (b_ok) *(unsigned*)(a.out + (size_t)r * a.N + rowA) = pack2(vA, vB);
    else a.out[(size_t)r * a.N + rowA] = f2bf(vA);
  } else if constexpr (EPI == EPI_F32) {
    a.out_f32[(size_t)r * a.N + rowA] = bfround(vA);
    if (b_ok) a.out_f32[(size_t)r * a.N + rowB] = bfround(vB);
  } else if constexpr (EPI == EPI_RESID) {
    // x + linear(...) with both operands bf16 (_torch.py:326-327)
    const size_t o = (size_t)r * a.N + rowA;
    if (b_ok) *(unsigned*)(a.out + o) = pack2(lo_f(resid) + bfround(vA), hi_f(resid) + bfround(vB));
    else a.out[o] = f2bf(lo_f(resid) + bfround(vA));
  } else if constexpr (EPI == EPI_SILU) {
    // y * silu(gate), fc1(x).chunk(2) (_torch.py:473-474): bf16 roundings after fc1, silu and mul
    const float y = bfround(vA), g = bfround(vB);
    const float s = bfround(g / (1.0f + expf(-g)));
    a.out[(size_t)r * (a.N >> 1) + u] = f2bf(y * s);
  } else if constexpr (EPI == EPI_MAMBA) {
    // Mamba2.step: zxbcdt = in_proj(x) (bf16), then xBC through the conv window; z and dt are consumed as stored
    const bf16_t xa = f2bf(vA), xb = f2bf(vB);
    if (b_ok) *(unsigned*)(a.out + (size_t)r * a.N + rowA) = (unsigned)xa | ((unsigned)xb << 16);
    else a.out[(size_t)r * a.N + rowA] = xa;                 // odd N (odd head count): the last row has no partner
    const int c = rowA - a.d_inner;
    if (c >= 0 && c < a.conv_dim) {
      unsigned s0 = cst.x, s1 = cst.y, s2 = cst.z, s3 = cst.w;
      const float o0 = mamba_conv_channel(s0, s1, cw.x, cw.y, lo_f(resid), xa);
      const float o1 = mamba_conv_channel(s2, s3, cw.z, cw.w, hi_f(resid), xb);
      *(u32x4*)(a.conv_state + ((size_t)r * a.conv_dim + c) * 4) = u32x4{s0, s1, s2, s3};
      *(unsigned*)(a.xbc + (size_t)r * a.conv_dim + c) = pack2(o0, o1);
    }
  } else if constexpr (EPI == EPI_ROPE_KV) {
    // split q|k|v (_torch.py:399-405), interleaved-pair RoPE in fp32 (_torch.py:57-68), KV append (:105-106)
    const int hd = a.hd, nq = a.n_heads * hd, nk = a.n_heads_kv * hd;
    const float x0 = bfround(vA), x1 = bfround(vB);
    if (rowA < nq + nk) {
      float re, im;
      zn_rope_pair(x0, x1, cs, sn, re, im);
      if (rowA < nq) *(unsigned*)(a.q_out + (size_t)r * nq + rowA) = pack2(re, im);
      else if (pos < a.max_len)
        *(unsigned*)(a.kv + (((size_t)r * a.max_len + pos) * 2 + 0) * nk + (rowA - nq)) = pack2(re, im);
    } else if (pos < a.max_len) {
      *(unsigned*)(a.kv + (((size_t)r * a.max_len + pos) * 2 + 1) * nk + (rowA - nq - nk)) = pack2(x0, x1);
    }
  }
}

// nn.LayerNorm pieces with every multiply-add spelled out (fused where written, nowhere else): left to the compiler, the
// same source line is contracted or packed (v_pk_mul + v_pk_add vs v_fmac) differently from one kernel to the next, and
// the persistent chain kernel (zn_chain_kernel.h) must reproduce gemv_kernel's statistics bit for bit.
ZN_DEVINL float ln_sum8(const u32x4 v) {
  return lo_f(v.x) + hi_f(v.x) + lo_f(v.y) + hi_f(v.y) + lo_f(v.z) + hi_f(v.z) + lo_f(v.w) + hi_f(v.w);
}
ZN_DEVINL float ln_sq8(const u32x4 v, float mean, float ss) {
  float d;
  d = __fsub_rn(lo_f(v.x), mean); ss = __fmaf_rn(d, d, ss); d = __fsub_rn(hi_f(v.x), mean); ss = __fmaf_rn(d, d, ss);
  d = __fsub_rn(lo_f(v.y), mean); ss = __fmaf_rn(d, d, ss); d = __fsub_rn(hi_f(v.y), mean); ss = __fmaf_rn(d, d, ss);
  d = __fsub_rn(lo_f(v.z), mean); ss = __fmaf_rn(d, d, ss); d = __fsub_rn(hi_f(v.z), mean); ss = __fmaf_rn(d, d, ss);
  d = __fsub_rn(lo_f(v.w), mean); ss = __fmaf_rn(d, d, ss); d = __fsub_rn(hi_f(v.w), mean); ss = __fmaf_rn(d, d, ss);
  return ss;
}
ZN_DEVINL float ln_one(float x, float m, float q, float g, float b) { return __fmaf_rn(__fmul_rn(__fsub_rn(x, m), q), g, b); }
ZN_DEVINL u32x4 ln_norm8(const u32x4 v, float m, float q, const u32x4 g, const u32x4 b) {
  u32x4 o;
  o.x = pack2(ln_one(lo_f(v.x), m, q, lo_f(g.x), lo_f(b.x)), ln_one(hi_f(v.x), m, q, hi_f(g.x), hi_f(b.x)));
  o.y = pack2(ln_one(lo_f(v.y), m, q, lo_f(g.y), lo_f(b.y)), ln_one(hi_f(v.y), m, q, hi_f(g.y), hi_f(b.y)));
  o.z = pack2(ln_one(lo_f(v.z), m, q, lo_f(g.z), lo_f(b.z)), ln_one(hi_f(v.z), m, q, hi_f(g.z), hi_f(b.z)));
  o.w = pack2(ln_one(lo_f(v.w), m, q, lo_f(g.w), lo_f(b.w)), ln_one(hi_f(v.w), m, q, hi_f(g.w), hi_f(b.w)));
  return o;
}
ZN_DEVINL float ln_rstd(float sumsq, float invK, float eps) { return 1.0f / sqrtf(__fmaf_rn(sumsq, invK, eps)); }

// Weight tile of one work unit (two weight rows) held in registers.
template <int NCH> struct WTile {
  u32x4 a[NCH], b[NCH];
  unsigned resid;   // EPI_RESID: this lane's (row = lane) residual pair, requested with the weights so that the
  float cs, sn;     // EPI_ROPE_KV: cos/sin      epilogue never has to wait behind the next unit's prefetch
  u32x4 cst, cw;    // EPI_MAMBA: conv window state and taps of the pair's two channels (biases in resid)
};

template <int NCH, int KSPLIT, int EPI, bool FULL>
ZN_DEVINL void gemv_load_unit(const GemvArgs& a, int u, int lane, int kw, int kbase, int pos, WTile<NCH>& t) {
  const int F = a.N >> 1;
  const bool u_ok = FULL || u < a.units;
  int rowA, rowB;
  if constexpr (EPI == EPI_SILU) { rowA = u; rowB = u + F; }
  else { rowA = 2 * u; rowB = 2 * u + 1; }
  const bool b_ok = FULL || (u_ok && rowB < a.N);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k = (c * 64 + lane) * 8;
    if constexpr (FULL) {
      t.a[c] = ld_nt16(a.W + (size_t)rowA * a.K + kbase + k);
      t.b[c] = ld_nt16(a.W + (size_t)rowB * a.K + kbase + k);
    } else {
      t.a[c] = u32x4{0, 0, 0, 0}; t.b[c] = u32x4{0, 0, 0, 0};
      if (u_ok && k < kw) {
        t.a[c] = ld_nt16(a.W + (size_t)rowA * a.K + kbase + k);
        if (b_ok) t.b[c] = ld_nt16(a.W + (size_t)rowB * a.K + kbase + k);
      }
    }
  }
  t.resid = 0; t.cs = 1.f; t.sn = 0.f;
  if constexpr (EPI == EPI_MAMBA) { t.cst = u32x4{0, 0, 0, 0}; t.cw = u32x4{0, 0, 0, 0}; }
  if (u_ok && lane < a.nrows) {
    if constexpr (EPI == EPI_MAMBA) mamba_epi_operands(a, lane, rowA, t.resid, t.cst, t.cw);
    if constexpr (EPI == EPI_RESID) {
      const size_t o = (size_t)lane * a.N + rowA;
      t.resid = b_ok ? *(const unsigned*)(a.resid + o) : (unsigned)a.resid[o];
    }
    if constexpr (EPI == EPI_ROPE_KV) {
      const int hd = a.hd;
      if (rowA < (a.n_heads + a.n_heads_kv) * hd) {
        const int i = (rowA % hd) >> 1;
        const int p = pos < a.rope_positions ? pos : a.rope_positions - 1;
        const float2 c2 = *(const float2*)(a.rope + ((size_t)p * (hd >> 1) + i) * 2);
        t.cs = c2.x; t.sn = c2.y;
      }
    }
  }
}

// FULL = no masking anywhere: K == NCH*512*KSPLIT, N even, every wave's units exist, nrows == R (host-checked).  The
// masked variant turns each guarded load into an exec branch with early waits, which serialises the weight stream.
template <int R, int NCH, int KSPLIT, int PRO, int EPI, bool FULL>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int K = a.K;
  const int kw = K / KSPLIT;                      // k-range of this wave
  const int kbase = (KSPLIT == 1) ? 0 : wave * kw;
  __shared__ float red[4][2][R];
  const int u0 = (KSPLIT == 1) ? (blockIdx.x * 4 + wave) * a.upw : blockIdx.x * a.upw;

  int pos = 0;
  if constexpr (EPI == EPI_ROPE_KV) { if (lane < a.nrows) pos = a.lengths[lane]; }
  // Issue order matters (vmcnt retires in order): the small L2-resident loads the prologue needs first, then the first
  // unit's weights (they do not depend on the activations), then pin that order; the prologue then waits only for the
  // former while the HBM stream is already in flight.
  u32x4 xr[NCH][R];
  u32x4 lng[PRO != PRO_NONE ? NCH : 1], lnb[PRO == PRO_LN ? NCH : 1];
  u32x4 zr[PRO == PRO_GATED ? NCH : 1][R];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k = (c * 64 + lane) * 8;
    const bool kv_ok = FULL || k < kw;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if constexpr (PRO != PRO_GATED) {
        if constexpr (FULL) xr[c][r] = ld16(a.x + (size_t)r * K + kbase + k);
        else {
          xr[c][r] = u32x4{0, 0, 0, 0};
          if (kv_ok && r < a.nrows) xr[c][r] = ld16(a.x + (size_t)r * K + kbase + k);
        }
      }
      if constexpr (PRO == PRO_GATED) {                    // eight fp32 values: xr = first four, zr = last four
        if constexpr (FULL) { xr[c][r] = ld16(a.gv + (size_t)r * K + kbase + k); zr[c][r] = ld16(a.gv + (size_t)r * K + kbase + k + 4); }
        else {
          xr[c][r] = u32x4{0, 0, 0, 0}; zr[c][r] = u32x4{0, 0, 0, 0};
          if (kv_ok && r < a.nrows) { xr[c][r] = ld16(a.gv + (size_t)r * K + kbase + k); zr[c][r] = ld16(a.gv + (size_t)r * K + kbase + k + 4); }
        }
      }
    }
    if constexpr (PRO == PRO_LN) {
      lng[c] = u32x4{0, 0, 0, 0}; lnb[c] = u32x4{0, 0, 0, 0};
      if (kv_ok) { lng[c] = ld16(a.ln_w + k); lnb[c] = ld16(a.ln_b + k); }
    }
    if constexpr (PRO == PRO_GATED) {
      lng[c] = u32x4{0, 0, 0, 0};
      if (kv_ok) lng[c] = ld16(a.ln_w + kbase + k);
    }
  }
  WTile<NCH> wt;
  gemv_load_unit<NCH, KSPLIT, EPI, FULL>(a, u0, lane, kw, kbase, pos, wt);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (PRO == PRO_LN) {
    // nn.LayerNorm (_torch.py:278,280,155): fp32 statistics, biased variance, affine, bf16 out.  KSPLIT == 1.
    const float invK = 1.0f / (float)K;
    float s[R], ss[R], mean[R], rstd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      s[r] = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) s[r] += ln_sum8(xr[c][r]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) mean[r] = __fmul_rn(wave_sum(s[r]), invK);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      ss[r] = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (FULL || (c * 64 + lane) * 8 < kw) ss[r] = ln_sq8(xr[c][r], mean[r], ss[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) rstd[r] = ln_rstd(wave_sum(ss[r]), invK, a.eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int k = (c * 64 + lane) * 8;
      if (FULL || k < kw) {
#pragma unroll
        for (int r = 0; r < R; ++r) xr[c][r] = ln_norm8(xr[c][r], mean[r], rstd[r], lng[c], lnb[c]);
      }
    }
  }

  if constexpr (PRO == PRO_GATED) {
    // mamba_ssm RMSNormGated(norm_before_gate=False), one group: v = y * silu(z) arrives in fp32 (mamba_ssm_kernel);
    // out = bf16(v * rstd * w) with rstd from the mean of v^2 over the row.  KSPLIT == 1: every wave holds whole rows;
    // KSPLIT == 4: a quarter each, the four partial sums meet in LDS in wave order.
    const float invK = 1.0f / (float)K;
    float ss[R], rstd[R];
    __shared__ float red_pro[4][R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      ss[r] = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const u32x4 va = xr[c][r], vb = zr[c][r];
        const float v[8] = {__uint_as_float(va.x), __uint_as_float(va.y), __uint_as_float(va.z), __uint_as_float(va.w),
                            __uint_as_float(vb.x), __uint_as_float(vb.y), __uint_as_float(vb.z), __uint_as_float(vb.w)};
#pragma unroll
        for (int e = 0; e < 8; ++e) ss[r] += v[e] * v[e];
      }
    }
    if constexpr (KSPLIT == 1) {
#pragma unroll
      for (int r = 0; r < R; ++r) rstd[r] = 1.0f / sqrtf(wave_sum(ss[r]) * invK + a.eps);
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) { const float t = wave_sum(ss[r]); if (lane == 0) red_pro[wave][r] = t; }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < R; ++r) rstd[r] = 1.0f / sqrtf((((red_pro[0][r] + red_pro[1][r]) + red_pro[2][r]) + red_pro[3][r]) * invK + a.eps);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const u32x4 g = lng[c];
      const float wf[8] = {lo_f(g.x), hi_f(g.x), lo_f(g.y), hi_f(g.y), lo_f(g.z), hi_f(g.z), lo_f(g.w), hi_f(g.w)};
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const u32x4 va = xr[c][r], vb = zr[c][r];
        const float q = rstd[r];
        u32x4 o;
        o.x = pack2(__fmul_rn(__fmul_rn(__uint_as_float(va.x), q), wf[0]), __fmul_rn(__fmul_rn(__uint_as_float(va.y), q), wf[1]));
        o.y = pack2(__fmul_rn(__fmul_rn(__uint_as_float(va.z), q), wf[2]), __fmul_rn(__fmul_rn(__uint_as_float(va.w), q), wf[3]));
        o.z = pack2(__fmul_rn(__fmul_rn(__uint_as_float(vb.x), q), wf[4]), __fmul_rn(__fmul_rn(__uint_as_float(vb.y), q), wf[5]));
        o.w = pack2(__fmul_rn(__fmul_rn(__uint_as_float(vb.z), q), wf[6]), __fmul_rn(__fmul_rn(__uint_as_float(vb.w), q), wf[7]));
        xr[c][r] = o;
      }
    }
  }

  // ZN_GEMV_F64 (experiment): every bf16 x bf16 product is exact in fp64 and so, to 2^-53, is their sum - the
  // dot product is then correctly rounded like the reference's oneDNN result, instead of carrying the ~2e-7 relative error of
  // fp32 lane partials + a reduction tree that flips ~2e-4 of the bf16 outputs by one ulp.
#ifdef ZN_GEMV_F64
  constexpr bool F64 = (R == 2);
#else
  constexpr bool F64 = false;
#endif
  double xd[F64 ? NCH : 1][F64 ? R : 1][8];
  if constexpr (F64) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const u32x4 v = xr[c][r];
        xd[c][r][0] = lo_f(v.x); xd[c][r][1] = hi_f(v.x); xd[c][r][2] = lo_f(v.y); xd[c][r][3] = hi_f(v.y);
        xd[c][r][4] = lo_f(v.z); xd[c][r][5] = hi_f(v.z); xd[c][r][6] = lo_f(v.w); xd[c][r][7] = hi_f(v.w);
      }
  }
  __shared__ double red_d[F64 && KSPLIT > 1 ? 4 : 1][2][R];
  // ---------------- main loop over work units (next unit's weights are requested before this one is reduced)
  const int F = a.N >> 1;  // EPI_SILU: gate rows start at N/2
  for (int it = 0; it < a.upw; ++it) {
    const int u = u0 + it;
    const bool u_ok = FULL || u < a.units;  // wave-uniform (block-uniform for KSPLIT=4)
    int rowA, rowB;
    if constexpr (EPI == EPI_SILU) { rowA = u; rowB = u + F; }
    else { rowA = 2 * u; rowB = 2 * u + 1; }
    const bool b_ok = FULL || (u_ok && rowB < a.N);
    float accA[R], accB[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { accA[r] = 0.f; accB[r] = 0.f; }
    WTile<NCH> cur = wt;
    if (it + 1 < a.upw) gemv_load_unit<NCH, KSPLIT, EPI, FULL>(a, u + 1, lane, kw, kbase, pos, wt);
    if constexpr (F64) {
      double dA[R], dB[R];
#pragma unroll
      for (int r = 0; r < R; ++r) { dA[r] = 0.0; dB[r] = 0.0; }
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const u32x4 wa = cur.a[c], wb = cur.b[c];
        const float fa[8] = {lo_f(wa.x), hi_f(wa.x), lo_f(wa.y), hi_f(wa.y), lo_f(wa.z), hi_f(wa.z), lo_f(wa.w), hi_f(wa.w)};
        const float fb[8] = {lo_f(wb.x), hi_f(wb.x), lo_f(wb.y), hi_f(wb.y), lo_f(wb.z), hi_f(wb.z), lo_f(wb.w), hi_f(wb.w)};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const double da = (double)fa[e], db = (double)fb[e];
#pragma unroll
          for (int r = 0; r < R; ++r) { dA[r] = __fma_rn(da, xd[c][r][e], dA[r]); dB[r] = __fma_rn(db, xd[c][r][e], dB[r]); }
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) { dA[r] = wave_sum_d(dA[r]); dB[r] = wave_sum_d(dB[r]); }
      if constexpr (KSPLIT > 1) {
        __syncthreads();
        if (lane == 0) {
#pragma unroll
          for (int r = 0; r < R; ++r) { red_d[wave][0][r] = dA[r]; red_d[wave][1][r] = dB[r]; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
          dA[r] = ((red_d[0][0][r] + red_d[1][0][r]) + red_d[2][0][r]) + red_d[3][0][r];
          dB[r] = ((red_d[0][1][r] + red_d[1][1][r]) + red_d[2][1][r]) + red_d[3][1][r];
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) { accA[r] = (float)dA[r]; accB[r] = (float)dB[r]; }
      if constexpr (KSPLIT > 1) { if (wave != 0) continue; }
    } else {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        accA[r] = dot8(cur.a[c], xr[c][r], accA[r]);
        accB[r] = dot8(cur.b[c], xr[c][r], accB[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { accA[r] = wave_sum(accA[r]); accB[r] = wave_sum(accB[r]); }
    }
    if constexpr (KSPLIT > 1 && !F64) {
      __syncthreads();
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) { red[wave][0][r] = accA[r]; red[wave][1][r] = accB[r]; }
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < R; ++r) {
        accA[r] = ((red[0][0][r] + red[1][0][r]) + red[2][0][r]) + red[3][0][r];
        accB[r] = ((red[0][1][r] + red[1][1][r]) + red[2][1][r]) + red[3][1][r];
      }
      if (wave != 0) continue;
    }
    if (!u_ok) continue;
    // ---------------- epilogue: lane r finishes row r
    float vA = 0.f, vB = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) if (lane == r) { vA = accA[r]; vB = accB[r]; }
    if (lane >= (FULL ? R : a.nrows)) continue;
    if constexpr (EPI == EPI_MAMBA) gemv_epilogue<EPI>(a, lane, rowA, rowB, b_ok, u, vA, vB, cur.resid, cur.cs, cur.sn, pos, cur.cst, cur.cw);
    else gemv_epilogue<EPI>(a, lane, rowA, rowB, b_ok, u, vA, vB, cur.resid, cur.cs, cur.sn, pos);
  }
}

// ------------------------------------------------------------------------------------------------ small-M MFMA
// 5..16 activation rows (batches of 3..8 utterances): one weight pass serves all rows on the matrix cores.  A workgroup
// owns 16 consecutive weight rows (EPI_SILU: the value tile and its gate tile); its NW waves split K, each streaming its
// 16 x K/NW weight slab straight into MFMA B fragments (v_mfma_f32_16x16x32_bf16: lane = (row n = l&15, k-group l>>4),
// 16 B per lane per 32-deep step) with the activation fragment fetched from L2; the partial 16x16 tiles meet in LDS
// and the shared epilogues finish them.  HBM-bound like the GEMV; MFMA only replaces 16 rows x dot2 on the VALU.
typedef __attribute__((ext_vector_type(8))) __bf16 zn_bf16x8;
// TR = weight rows per workgroup: 16, or 8 (lanes n >= 8 idle) when N / 16 tiles would leave CUs without a workgroup
// (N = d_model = 2048: 128 tiles on 256 CUs) — twice the workgroups at half the rows each.
template <int NW, int EPI, int TR = 16>
__global__ __launch_bounds__(NW * 64) void gemm16_kernel(GemvArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int u = blockIdx.x;                       // tile of TR weight rows
  const int F = a.N >> 1;
  const int rowbaseA = TR * u;
  const int rowbaseB = F + TR * u;                // EPI_SILU only
  const int K = a.K, kw = K / NW, kbase = wave * kw;
  const bool n_ok = n < TR && rowbaseA + n < ((EPI == EPI_SILU) ? F : a.N);
  const bf16_t* wa = a.W + (size_t)(rowbaseA + (n_ok ? n : 0)) * K + kbase + 8 * g;
  const bf16_t* wb = a.W + (size_t)(rowbaseB + (n_ok ? n : 0)) * K + kbase + 8 * g;
  const bool m_ok = n < a.nrows;                  // A fragment: lane's activation row is (lane & 15)
  const bf16_t* xa = a.x + (size_t)(m_ok ? n : 0) * K + kbase + 8 * g;
  f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
  constexpr int UN = 8;                           // k-steps in flight
  for (int k0 = 0; k0 < kw; k0 += 32 * UN) {
    u32x4 fa[UN], fb[UN], fx[UN];
#pragma unroll
    for (int s = 0; s < UN; ++s) {
      const int k = k0 + 32 * s;
      const bool ok = k < kw;
      fa[s] = (ok && n_ok) ? ld_nt16(wa + k) : u32x4{0, 0, 0, 0};
      if constexpr (EPI == EPI_SILU) fb[s] = (ok && n_ok) ? ld_nt16(wb + k) : u32x4{0, 0, 0, 0};
      fx[s] = (ok && m_ok) ? ld16(xa + k) : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int s = 0; s < UN; ++s) {
      const zn_bf16x8 xf = __builtin_bit_cast(zn_bf16x8, fx[s]);
      accA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, __builtin_bit_cast(zn_bf16x8, fa[s]), accA, 0, 0, 0);
      if constexpr (EPI == EPI_SILU) accB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, __builtin_bit_cast(zn_bf16x8, fb[s]), accB, 0, 0, 0);
    }
  }
  // C layout: col (weight row) = lane & 15, row (activation row) = 4*(lane>>4) + reg
  __shared__ float s_t[NW][2][16][17];
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    s_t[wave][0][4 * g + reg][n] = accA[reg];
    if constexpr (EPI == EPI_SILU) s_t[wave][1][4 * g + reg][n] = accB[reg];
  }
  __syncthreads();
  const int t = threadIdx.x;
  if constexpr (EPI == EPI_SILU) {
    if (t >= 256) return;
    const int m = t >> 4, nn = t & 15;
    if (m >= a.nrows || nn >= TR || TR * u + nn >= F) return;
    float vA = 0.f, vB = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { vA += s_t[w][0][m][nn]; vB += s_t[w][1][m][nn]; }
    gemv_epilogue<EPI>(a, m, TR * u + nn, F + TR * u + nn, true, TR * u + nn, vA, vB, 0u, 1.f, 0.f, 0);
  } else {
    if (t >= 128) return;
    const int m = t >> 3, np = t & 7;
    const int rowA = TR * u + 2 * np, rowB = rowA + 1;
    if (m >= a.nrows || 2 * np >= TR || rowA >= a.N) return;
    const bool b_ok = rowB < a.N;
    float vA = 0.f, vB = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { vA += s_t[w][0][m][2 * np]; vB += s_t[w][0][m][2 * np + 1]; }
    unsigned resid = 0; float cs = 1.f, sn = 0.f; int pos = 0;
    if constexpr (EPI == EPI_RESID) {
      const size_t o = (size_t)m * a.N + rowA;
      resid = b_ok ? *(const unsigned*)(a.resid + o) : (unsigned)a.resid[o];
    }
    if constexpr (EPI == EPI_ROPE_KV) {
      pos = a.lengths[m];
      if (rowA < (a.n_heads + a.n_heads_kv) * a.hd) {
        const int i = (rowA % a.hd) >> 1;
        const int p = pos < a.rope_positions ? pos : a.rope_positions - 1;
        const float2 c2 = *(const float2*)(a.rope + ((size_t)p * (a.hd >> 1) + i) * 2);
        cs = c2.x; sn = c2.y;
      }
    }
    if constexpr (EPI == EPI_MAMBA) {
      u32x4 cst = u32x4{0, 0, 0, 0}, cw = u32x4{0, 0, 0, 0};
      mamba_epi_operands(a, m, rowA, resid, cst, cw);
      gemv_epilogue<EPI>(a, m, rowA, rowB, b_ok, rowA >> 1, vA, vB, resid, cs, sn, pos, cst, cw);
    } else gemv_epilogue<EPI>(a, m, rowA, rowB, b_ok, rowA >> 1, vA, vB, resid, cs, sn, pos);
  }
}

// ------------------------------------------------------------------------------------------------ small-M MFMA, LDS-staged
// gemm16_kernel feeds MFMA fragments straight from global memory: every 16-lane phase of its weight loads touches 16
// different rows (16 cache lines for 256 bytes), and the vector-memory pipe, not HBM, bounds it (2-3.7 TB/s at 16 rows).
// Here the weight stream is read the way the GEMV reads it (each 16-lane phase = 256 contiguous bytes of one row), staged
// in LDS (row stride KC + 8 elements: conflict-free 16-B fragment reads) together with the activation chunk, and the
// fragments come from LDS.  A workgroup owns 64 weight rows (EPI_SILU: 32 value rows + their 32 gate rows), one 16-row
// MFMA tile per wave, and walks its K slice in chunks of KC = 256 with the next chunk's global loads in flight during
// the MFMAs.  Small N (d_model rows) splits K over gridDim.y workgroups: fp32 partial tiles go to a scratch buffer and
// the last workgroup to arrive (ticket, no waiting) adds them in slice order and runs the epilogue — deterministic.
#define ZN_G16_KC 256
ZN_DEVINL void st_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // write-through
ZN_DEVINL float ld_wt(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // L2-bypassing
// NWV = waves (16-row tiles) per workgroup: 4 (64 rows) or 2 (32 rows; more, smaller workgroups for mid-size N).
// LNP: the activations are a residual stream whose nn.LayerNorm has not been applied yet; its producer left per-tile statistics
// (GemvArgs::ln_part_in, ZN_G16_LNT tiles of 16 columns per row).  The workgroup adds them in a fixed order while its first weight chunk
// is on the way and normalises every activation chunk as it stages it (the arithmetic of layernorm_kernel on each value; the statistics
// are summed tile-wise instead of lane-wise) - no LayerNorm launch, no normalised copy of the rows in memory.
#define ZN_G16_LNT 128
template <int EPI, int NWV, bool LNP = false>
__global__ __launch_bounds__(NWV * 64) void gemm16s_kernel(GemvArgs a) {
  constexpr int KC = ZN_G16_KC, LDW = KC + 8, NT = NWV * 64, TN = NWV * 16, HALF = TN / 2;
  constexpr int XP = 512 / NT;                        // activation pieces per thread (16 rows x 32 pieces)
  __shared__ __attribute__((aligned(16))) bf16_t Ws[TN * LDW];
  __shared__ __attribute__((aligned(16))) bf16_t Xs[16 * LDW];
  __shared__ float Ct[TN][17];
  __shared__ int s_last;
  constexpr int LPT = LNP ? 16 * ZN_G16_LNT / NT : 1, LSEG = NT / 16, LTS = ZN_G16_LNT / LSEG;   // partials per thread; segments of a row; tiles per segment
  __shared__ float2 s_lnp[LNP ? 16 : 1][LNP ? ZN_G16_LNT + 1 : 1];
  __shared__ float s_lnseg[LNP ? 16 : 1][LNP ? LSEG : 1];
  __shared__ float s_mr[LNP ? 16 : 1][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int grp = blockIdx.x, ys = blockIdx.y;
  const int K = a.K, F = a.N >> 1;
  // LDS row j of this workgroup <-> weight row (clamped: never out of bounds; masked in the epilogue)
  auto wrow = [&](int j) -> int {
    if constexpr (EPI == EPI_SILU) {
      const int r = (j < HALF) ? grp * HALF + j : F + grp * HALF + (j - HALF);
      const int lim = (j < HALF) ? F : a.N;
      return r < lim ? r : lim - 1;
    } else {
      const int r = grp * TN + j;
      return r < a.N ? r : a.N - 1;
    }
  };
  const int kslice = K / a.ksplit, kbeg = ys * kslice, nchunks = kslice / KC;
  u32x4 wr[8], xr[XP];
  const bf16_t* wp[8];
  int wl[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int idx = j * NT + tid, row = idx >> 5, c16 = idx & 31;      // 32 x 16 B per row chunk
    wp[j] = a.W + (size_t)wrow(row) * K + kbeg + c16 * 8;
    wl[j] = row * LDW + c16 * 8;
  }
  const bf16_t* xp[XP];
  int xl[XP];
#pragma unroll
  for (int j = 0; j < XP; ++j) {
    const int idx = j * NT + tid, row = idx >> 5, c16 = idx & 31;
    xp[j] = a.x + (size_t)(row < a.nrows ? row : a.nrows - 1) * K + kbeg + c16 * 8;
    xl[j] = row * LDW + c16 * 8;
  }
  float2 lp[LPT];
  u32x4 lnw = u32x4{0, 0, 0, 0}, lnb = u32x4{0, 0, 0, 0};
  const int lnk = kbeg + (tid & 31) * 8;               // every piece of a thread sits in the same 16-byte column of the chunk (NT % 32 == 0)
  if constexpr (LNP) {
    // requests return in order: statistics first, then the first activation chunk and its affine parameters, then the weights
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
      const int idx = j * NT + tid, row = idx / ZN_G16_LNT, t = idx % ZN_G16_LNT;
      lp[j] = *(const float2*)(a.ln_part_in + ((size_t)(row < a.nrows ? row : a.nrows - 1) * ZN_G16_LNT + t) * 2);
    }
  }
#pragma unroll
  for (int j = 0; j < XP; ++j) xr[j] = ld16(xp[j]);
  if constexpr (LNP) { lnw = ld16(a.ln_w + lnk); lnb = ld16(a.ln_b + lnk); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
  for (int j = 0; j < 8; ++j) wr[j] = ld_nt16(wp[j]);
  // (LNP: the statistics end later than the first weight chunk arrives - their inputs are as far away as the weights.  Requesting the second
  // weight chunk here as well, for the loop's first iteration to take over: 1.765 vs 1.688 ms per batch-8 step - more in flight per thread is
  // slower, as the note at the loop says.)
  if constexpr (LNP) {
    __builtin_amdgcn_sched_barrier(0);                 // the weight requests stay above the statistics' wait
#pragma unroll
    for (int j = 0; j < LPT; ++j) { const int idx = j * NT + tid; s_lnp[idx / ZN_G16_LNT][idx % ZN_G16_LNT] = lp[j]; }
    __syncthreads();
    const int lrow = tid & 15, lseg = tid >> 4;
    float ssum = 0.f;
#pragma unroll
    for (int t = 0; t < LTS; ++t) ssum += s_lnp[lrow][lseg * LTS + t].x;
    s_lnseg[lrow][lseg] = ssum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < LSEG; ++q) tot += s_lnseg[lrow][q];
    const float invK = 1.0f / (float)K, mean = tot * invK;
    __syncthreads();
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < LTS; ++t) {
      const float2 pt = s_lnp[lrow][lseg * LTS + t];
      const float dm = pt.x * 0.0625f - mean;
      sq += pt.y + 16.0f * dm * dm;
    }
    s_lnseg[lrow][lseg] = sq;
    __syncthreads();
    if (tid < 16) {
      float tq = 0.f;
#pragma unroll
      for (int q = 0; q < LSEG; ++q) tq += s_lnseg[tid][q];
      s_mr[tid][0] = mean;
      s_mr[tid][1] = ln_rstd(tq, invK, a.eps);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < XP; ++j) { const int row = (j * NT + tid) >> 5; xr[j] = ln_norm8(xr[j], s_mr[row][0], s_mr[row][1], lnw, lnb); }
  }
  // the residual operands of this thread's two epilogue items are requested with the first chunk: with K split over workgroups only the
  // last arriver uses them, but there they would be one more dependent round trip behind the ticket and the read-back of the slices
  unsigned resid_pre[2] = {0u, 0u};
  if constexpr (EPI == EPI_RESID) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = it * NT + tid, m = item & 15, pj = item >> 4, rowA = grp * TN + 2 * pj;
      if (m < a.nrows && rowA < a.N) {
        const size_t o = (size_t)m * a.N + rowA;
        resid_pre[it] = (rowA + 1 < a.N) ? *(const unsigned*)(a.resid + o) : (unsigned)a.resid[o];
      }
    }
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // one chunk in flight per thread: two or four (register sets requested further ahead) measured 1-3 % slower per step
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();                                   // the previous chunk's fragment reads are done
#pragma unroll
    for (int j = 0; j < XP; ++j) *(u32x4*)&Xs[xl[j]] = xr[j];       // (LNP: normalised one iteration earlier, below)
#pragma unroll
    for (int j = 0; j < 8; ++j) *(u32x4*)&Ws[wl[j]] = wr[j];
    __syncthreads();
    if (c + 1 < nchunks) {
#pragma unroll
      for (int j = 0; j < XP; ++j) xr[j] = ld16(xp[j] + (size_t)(c + 1) * KC);
      if constexpr (LNP) {
        lnw = ld16(a.ln_w + lnk + (c + 1) * KC); lnb = ld16(a.ln_b + lnk + (c + 1) * KC);
        __builtin_amdgcn_sched_barrier(0);             // requests return in order: the activations must not queue behind the weights (the scheduler interleaved them)
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) wr[j] = ld_nt16(wp[j] + (size_t)(c + 1) * KC);
      if constexpr (LNP) __builtin_amdgcn_sched_barrier(0);
    }
    const bf16_t* wf = &Ws[(wave * 16 + n) * LDW + 8 * g];
    const bf16_t* xf = &Xs[n * LDW + 8 * g];
#pragma unroll
    for (int st = 0; st < KC / 32; ++st) {
      const u32x4 bw = *(const u32x4*)(wf + 32 * st);
      const u32x4 ax = *(const u32x4*)(xf + 32 * st);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(zn_bf16x8, ax), __builtin_bit_cast(zn_bf16x8, bw), acc, 0, 0, 0);
    }
    if constexpr (LNP) {
      // the next chunk's activations (L2-resident, requested ahead of its weights) are normalised here, while those weights are still on
      // their way from HBM (every workgroup repeats the normalisation of all 16 rows: 256 values per thread over the launch).  In front
      // of the LDS writes, i.e. behind the wait for the weights: 16.3 vs 13.7 us per launch for the kernel without the normalisation, here
      // 14.5; activations requested two chunks ahead and normalised right behind the weight requests: slower (their late requests end
      // up in the loop head's wait).
      if (c + 1 < nchunks) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < XP; ++j) { const int row = (j * NT + tid) >> 5; xr[j] = ln_norm8(xr[j], s_mr[row][0], s_mr[row][1], lnw, lnb); }
      }
    }
  }
  // D layout: col (LDS row wave*16 + n) = lane & 15, row (activation row) = 4*(lane>>4) + reg
  if (a.ksplit > 1) {
    // partial tiles leave by write-through stores, the ticket follows once they are acknowledged (no cache-wide fence);
    // the last workgroup to arrive reads every slice past L2 and adds them in slice order
    const size_t ld = (size_t)gridDim.x * TN;
    float* pp = a.part + ((size_t)ys * 16) * ld + (size_t)grp * TN + wave * 16 + n;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) st_wt(pp + (size_t)(4 * g + reg) * ld, acc[reg]);
    __builtin_amdgcn_s_waitcnt(0);                     // vmcnt(0): stores acknowledged
    __syncthreads();
    if (tid == 0) {
      const int t = __hip_atomic_fetch_add(a.tickets + grp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (t == a.ksplit - 1);
      if (s_last) __hip_atomic_store(a.tickets + grp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    __syncthreads();
    if (!s_last) return;
    // 16 * TN / NT = 4 tile elements per thread, every slice of all four requested before the first add
    constexpr int IPT = 16 * TN / NT;
    float v[IPT][16];
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
      const int item = i * NT + tid, m = item / TN, col = item % TN;
      const float* q = a.part + (size_t)m * ld + (size_t)grp * TN + col;
#pragma unroll
      for (int y = 0; y < 16; ++y) v[i][y] = (y < a.ksplit) ? ld_wt(q + (size_t)y * 16 * ld) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
      const int item = i * NT + tid, m = item / TN, col = item % TN;
      float sum = 0.f;
#pragma unroll
      for (int y = 0; y < 16; ++y) sum += v[i][y];
      Ct[col][m] = sum;
    }
  } else {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) Ct[wave * 16 + n][4 * g + reg] = acc[reg];
  }
  __syncthreads();
  // ---- epilogue: TN/2 row pairs x 16 activation rows, 2 items per thread
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int item = it * NT + tid, m = item & 15, pj = item >> 4;       // pj in [0, TN/2)
    if (m >= a.nrows) continue;
    if constexpr (EPI == EPI_SILU) {
      const int u = grp * HALF + pj;
      if (u >= F) continue;
      gemv_epilogue<EPI>(a, m, u, F + u, true, u, Ct[pj][m], Ct[HALF + pj][m], 0u, 1.f, 0.f, 0);
    } else {
      const int rowA = grp * TN + 2 * pj, rowB = rowA + 1;
      if (rowA >= a.N) continue;
      const bool b_ok = rowB < a.N;
      unsigned resid = 0; float cs = 1.f, sn = 0.f; int pos = 0;
      if constexpr (EPI == EPI_RESID) resid = resid_pre[it];
      if constexpr (EPI == EPI_ROPE_KV) {
        pos = a.lengths[m];
        if (rowA < (a.n_heads + a.n_heads_kv) * a.hd) {
          const int i = (rowA % a.hd) >> 1;
          const int p = pos < a.rope_positions ? pos : a.rope_positions - 1;
          const float2 c2 = *(const float2*)(a.rope + ((size_t)p * (a.hd >> 1) + i) * 2);
          cs = c2.x; sn = c2.y;
        }
      }
      if constexpr (EPI == EPI_MAMBA) {
        u32x4 cst = u32x4{0, 0, 0, 0}, cw = u32x4{0, 0, 0, 0};
        mamba_epi_operands(a, m, rowA, resid, cst, cw);
        gemv_epilogue<EPI>(a, m, rowA, rowB, b_ok, rowA >> 1, Ct[2 * pj][m], b_ok ? Ct[2 * pj + 1][m] : 0.f, resid, cs, sn, pos, cst, cw);
      } else gemv_epilogue<EPI>(a, m, rowA, rowB, b_ok, rowA >> 1, Ct[2 * pj][m], b_ok ? Ct[2 * pj + 1][m] : 0.f, resid, cs, sn, pos);
    }
  }
}

// ------------------------------------------------------------------------------------------------ small-M MFMA, K split inside the workgroup
// Short contractions (K = 8 x 256 or 8 x 512: in_proj, out_proj) with few weight rows cannot fill the chip with
// gemm16s_kernel's 32/64-row workgroups unless K is split over workgroups, and that combine (write-through partials,
// ticket, read-back) is three dependent memory round trips on a launch-bound chain.  Here a workgroup owns one 16-row
// weight tile over the whole K: its 8 waves take K/8 each, request their entire slice at once (weights as whole
// 256-byte row pieces -> per-wave LDS -> B fragments; activations straight as A fragments, they are L2-resident), and the
// eight partial tiles meet in LDS in wave order - one memory round trip, deterministic.  (K = 8192, fc2, in rounds of four
// chunks per wave was measured 3 % slower per step than the split-K kernel: a 16-row tile re-reads the activations once
// per 16 weight rows, which is as many bytes through the vector-memory pipe as the weights themselves.)
#define ZN_G16K_NKW 8
#define ZN_G16K_KCH 128
// PRO_LN: the workgroup holds whole activation rows (as A fragments, spread over its waves), so nn.LayerNorm runs in
// place on them - row sums through LDS in wave order, two statistics passes like layernorm_kernel - instead of as a
// launch of its own in front (4.8 us at 16 rows).
template <int EPI, int NCH, int PRO>      // NCH = 128-wide chunks per wave: K = 8 * 128 * NCH
__global__ __launch_bounds__(ZN_G16K_NKW * 64) void gemm16k_kernel(GemvArgs a) {
  static_assert(EPI != EPI_SILU, "gemm16k: row-pair epilogues only");
  static_assert(PRO == PRO_NONE || PRO == PRO_LN, "gemm16k: LayerNorm is the only fused prologue");
  constexpr int NKW = ZN_G16K_NKW, KCH = ZN_G16K_KCH, LDW = KCH + 8, KS = KCH * NCH;
  __shared__ __attribute__((aligned(16))) bf16_t Ws[NKW][16 * LDW];
  __shared__ float Ct[NKW][16][17];
  __shared__ float s_red[2][NKW][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x, K = a.K, kbeg = wave * KS;
  // gridDim.y > 1 (short-prompt prefill: up to 64 rows): workgroup (tile, y) serves activation rows 16 y .. 16 y + 15; the tile's
  // weights are read once per row group (the groups of a tile share blockIdx.x % 8, i.e. an XCD and its L2, when gridDim.x % 8 == 0)
  if (blockIdx.y) {
    const size_t r0 = (size_t)blockIdx.y * 16;
    a.x += r0 * K; a.nrows -= (int)r0;
    if (a.out) a.out += r0 * a.N;
    if (a.resid) a.resid += r0 * a.N;
  }
  if (a.nrows > 16) a.nrows = 16;
  int pf_r = 0;                                            // prefill rows: cache row of this thread's activation row
  if constexpr (EPI == EPI_ROPE_KV) {
    if (a.pf_S > 0) {
      const int m = (int)blockIdx.y * 16 + (tid & 15);
      pf_r = m / a.pf_S;
      a.q_out += (size_t)(m - pf_r) * a.n_heads * a.hd;    // gemv_epilogue indexes q_out and the cache with one row number: q row m, cache row pf_r
    }
  }
  // epilogue operands first (item = tid < 128: activation row m, weight-row pair pj): their round trips overlap the stream's
  const int em = tid & 15, epj = (tid >> 4) & 7;
  const int erowA = tile * 16 + 2 * epj, erowB = erowA + 1;
  const bool e_on = tid < 128 && em < a.nrows && erowA < a.N, eb_ok = erowB < a.N;
  unsigned resid = 0; int pos = 0;
  if constexpr (EPI == EPI_RESID) {
    if (e_on) { const size_t o = (size_t)em * a.N + erowA; resid = eb_ok ? *(const unsigned*)(a.resid + o) : (unsigned)a.resid[o]; }
  }
  if constexpr (EPI == EPI_ROPE_KV) { if (e_on) pos = a.pf_S > 0 ? a.pf_base + ((int)blockIdx.y * 16 + em) % a.pf_S : a.lengths[em]; }
  u32x4 cst = u32x4{0, 0, 0, 0}, cw = u32x4{0, 0, 0, 0};
  if constexpr (EPI == EPI_MAMBA) { if (e_on) mamba_epi_operands(a, em, erowA, resid, cst, cw); }
  // activations (and LayerNorm parameters) are requested first: requests return in order, and the statistics passes then
  // run while the weights are still on their way
  u32x4 wr[NCH][4], xr[NCH][KCH / 32];
  u32x4 lw[PRO == PRO_LN ? NCH : 1][KCH / 32], lb[PRO == PRO_LN ? NCH : 1][KCH / 32];
  {
    const bf16_t* xp = a.x + (size_t)min(n, a.nrows - 1) * K + kbeg + 8 * g;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int st = 0; st < KCH / 32; ++st) xr[c][st] = ld16(xp + c * KCH + 32 * st);
    if constexpr (PRO == PRO_LN) {
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int st = 0; st < KCH / 32; ++st) {
          lw[c][st] = ld16(a.ln_w + kbeg + 8 * g + c * KCH + 32 * st);
          lb[c][st] = ld16(a.ln_b + kbeg + 8 * g + c * KCH + 32 * st);
        }
    }
  }
  // weights: load i of chunk c = rows 4i .. 4i+3 of the tile, 16 lanes x 16 B = 256 contiguous bytes of one row each
  const int wsub = lane & 15, wq = lane >> 4;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = min(tile * 16 + 4 * i + wq, a.N - 1);        // clamped: never out of bounds; masked in the epilogue
      wr[c][i] = ld_nt16(a.W + (size_t)row * K + kbeg + c * KCH + 8 * wsub);
    }
  float cs = 1.f, sn = 0.f;
  if constexpr (EPI == EPI_ROPE_KV) {
    if (e_on && erowA < (a.n_heads + a.n_heads_kv) * a.hd) {
      const int i = (erowA % a.hd) >> 1;
      const int p = pos < a.rope_positions ? pos : a.rope_positions - 1;
      const float2 c2 = *(const float2*)(a.rope + ((size_t)p * (a.hd >> 1) + i) * 2);
      cs = c2.x; sn = c2.y;
    }
  }
  if constexpr (PRO == PRO_LN) {
    // lane (n, g) holds 32 * NCH values of activation row n; the row's other values sit in lanes n + 16 q and in the
    // other waves.  fp32 statistics (nn.LayerNorm): mean, then the centred second moment, each summed lane -> row group
    // -> waves in a fixed order.
    float ps = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int st = 0; st < KCH / 32; ++st) {
        const u32x4 v = xr[c][st];
        ps += lo_f(v.x) + hi_f(v.x) + lo_f(v.y) + hi_f(v.y) + lo_f(v.z) + hi_f(v.z) + lo_f(v.w) + hi_f(v.w);
      }
    ps += __shfl_xor(ps, 16); ps += __shfl_xor(ps, 32);
    if (g == 0) s_red[0][wave][n] = ps;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NKW; ++w) tot += s_red[0][w][n];
    const float mean = tot / (float)K;
    float pss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int st = 0; st < KCH / 32; ++st) {
        const u32x4 v = xr[c][st];
        const float f[8] = {lo_f(v.x), hi_f(v.x), lo_f(v.y), hi_f(v.y), lo_f(v.z), hi_f(v.z), lo_f(v.w), hi_f(v.w)};
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float dd = f[e] - mean; pss += dd * dd; }
      }
    pss += __shfl_xor(pss, 16); pss += __shfl_xor(pss, 32);
    if (g == 0) s_red[1][wave][n] = pss;
    __syncthreads();
    float tot2 = 0.f;
#pragma unroll
    for (int w = 0; w < NKW; ++w) tot2 += s_red[1][w][n];
    const float rstd = 1.0f / sqrtf(tot2 / (float)K + a.eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int st = 0; st < KCH / 32; ++st) {
        const u32x4 v = xr[c][st], gw = lw[c][st], gb = lb[c][st];
        u32x4 o;
        o.x = pack2((lo_f(v.x) - mean) * rstd * lo_f(gw.x) + lo_f(gb.x), (hi_f(v.x) - mean) * rstd * hi_f(gw.x) + hi_f(gb.x));
        o.y = pack2((lo_f(v.y) - mean) * rstd * lo_f(gw.y) + lo_f(gb.y), (hi_f(v.y) - mean) * rstd * hi_f(gw.y) + hi_f(gb.y));
        o.z = pack2((lo_f(v.z) - mean) * rstd * lo_f(gw.z) + lo_f(gb.z), (hi_f(v.z) - mean) * rstd * hi_f(gw.z) + hi_f(gb.z));
        o.w = pack2((lo_f(v.w) - mean) * rstd * lo_f(gw.w) + lo_f(gb.w), (hi_f(v.w) - mean) * rstd * hi_f(gw.w) + hi_f(gb.w));
        xr[c][st] = o;
      }
  }
  bf16_t* ws = &Ws[wave][0];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *(u32x4*)(ws + (4 * i + wq) * LDW + 8 * wsub) = wr[c][i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");           // the region is private to the wave: its LDS ops complete in order
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int st = 0; st < KCH / 32; ++st) {
      const u32x4 bw = *(const u32x4*)(ws + n * LDW + 32 * st + 8 * g);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(zn_bf16x8, xr[c][st]), __builtin_bit_cast(zn_bf16x8, bw), acc, 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // D layout: col (weight row of the tile) = lane & 15, row (activation row) = 4*(lane>>4) + reg
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) Ct[wave][n][4 * g + reg] = acc[reg];
  __syncthreads();
  float vA = 0.f, vB = 0.f;
  if (e_on) {
#pragma unroll
    for (int w = 0; w < NKW; ++w) { vA += Ct[w][2 * epj][em]; vB += Ct[w][2 * epj + 1][em]; }
    int er = em;
    if constexpr (EPI == EPI_ROPE_KV) { if (a.pf_S > 0) er = pf_r; }
    gemv_epilogue<EPI>(a, er, erowA, erowB, eb_ok, erowA >> 1, vA, eb_ok ? vB : 0.f, resid, cs, sn, pos, cst, cw);
  }
  if constexpr (EPI == EPI_RESID) {
    // LayerNorm statistics of this tile's 16 output columns per activation row (GemvArgs::ln_part_out; the host asks for them only
    // when N % 16 == 0): the eight column pairs of row em sit in lanes em + 16 q of waves 0 and 1.  Per wave: sum and centred second
    // moment of its 8 values; the two halves meet in LDS (Chan's pairwise update, equal counts).
    if (a.ln_part_out == nullptr) return;                 // uniform
    const float oA = bfround(lo_f(resid) + bfround(vA)), oB = bfround(hi_f(resid) + bfround(vB));      // the values gemv_epilogue stored
    float s8 = oA + oB;
    s8 += __shfl_xor(s8, 16); s8 += __shfl_xor(s8, 32);
    const float m8 = s8 * 0.125f;
    float q8 = (oA - m8) * (oA - m8) + (oB - m8) * (oB - m8);
    q8 += __shfl_xor(q8, 16); q8 += __shfl_xor(q8, 32);
    if (wave == 1 && lane < 16) { s_red[0][0][lane] = s8; s_red[1][0][lane] = q8; }
    __syncthreads();
    if (wave == 0 && lane < 16 && lane < a.nrows) {
      const float sb = s_red[0][0][lane], qb = s_red[1][0][lane], dm = (sb - s8) * 0.125f;
      float2 o;
      o.x = s8 + sb;
      o.y = q8 + qb + 4.0f * dm * dm;
      *(float2*)(a.ln_part_out + ((size_t)((int)blockIdx.y * 16 + lane) * gridDim.x + tile) * 2) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------ attention
// Reference semantics (torch CPU flash attention as called at _torch.py:415, verified numerically):
//   s_t = fl32(q.k_t) * fl32(1/sqrt(hd)); keys are walked in blocks of 512; per block the running max m is
//   updated, e_t = exp(s_t - m) uses fexp_u20 for the first 16*floor(n/16) keys of the block and libm exp for
//   the rest, the row sum uses the unrounded e_t, P.V uses bf16(e_t), and the output is bf16(acc * (1/sum)).
struct AttnArgs {
  const bf16_t* q;      // [rows][Hq*hd] (post-RoPE)
  const bf16_t* kv;     // [rows][max_len][2][Hkv][hd]
  const int* lengths;   // position of the newest key (already appended): L = lengths[r] + 1 keys
  const int* ext;       // optional int32 [rows]: keys the reference's block loop spans (prefill emulation)
  int ext_scalar;       // used when ext == NULL and > 0
  int max_len, n_heads, n_heads_kv, lcap;   // lcap = scores row stride (multiple of 512)
  int rows;             // activation rows (fused variant's 1-D grid decode)
  float* part;          // split variant: per (row, kv head, block) partial [G*HD + G] (P.V sums, e sums)
  int* tickets;         // split variant: arrival counter per (row, kv head)
  int nbcap;            // split variant: blocks the grid covers (lcap / 512)
  float scale;
  float* scores;        // [rows][Hq][lcap]
  float* cmax;          // [rows][Hq][lcap/64]   per-64-key-chunk maxima
  bf16_t* out;          // [rows][Hq*hd]
  unsigned long long* stamps;   // optional [8] diagnostic timeline (s_memrealtime) of workgroup 0 of the fused launch
};
#define ZN_ACHUNK 64

// A single CU sustains only a few tens of GB/s, so the KV read of a (row, kv-head) pair is spread over one workgroup
// per 64-key chunk (grid = chunks x Hkv x rows, ~112 workgroups at 10 s of context).  Pass 1: scores + chunk maxima.
// The scores are formed on the matrix cores exactly as every other attention kernel of this library forms them (the fused launch
// below, the whole-step kernels' attention roles): S[head][key] = Q K^T as 16x16x32 tiles (A = the G query heads padded to 16 rows,
// B = 16 keys, k = 32 consecutive head dims per step, steps in ascending order), fp32 accumulate, then the reference's fp32
// scale - ONE summation order for q.k on every path, so that the paths stay bit-identical to one another at every context length.
// Each wave owns one 16-key tile and requests its keys straight in the B-fragment layout (lane = (key l & 15, k-group l >> 4)).
template <int HD, int G>
__global__ __launch_bounds__(256) void attn_scores_kernel(AttnArgs a) {
  static_assert(HD % 32 == 0 && G <= 16, "score tile shape");
  constexpr int KST = HD / 32;     // MFMA steps over the head dimension
  typedef __attribute__((ext_vector_type(4))) float f32x4_t;
  const int chunk = blockIdx.x, kvh = blockIdx.y, r = blockIdx.z;
  // The length is requested first and every other request is issued before it is needed (addresses clamped to the
  // cache capacity, results masked at use): one memory round trip on the launch-bound chain instead of two.
  const int Lraw = a.lengths[r];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kn = lane & 15, kg = lane >> 4;
  const size_t kvrow = (size_t)2 * a.n_heads_kv * HD;
  const int t = chunk * ZN_ACHUNK + wave * 16 + kn;
  const bf16_t* kp = a.kv + ((size_t)r * a.max_len + (size_t)min(t, a.max_len - 1)) * kvrow + (size_t)kvh * HD + 8 * kg;
  u32x4 kk[KST];
#pragma unroll
  for (int st = 0; st < KST; ++st) kk[st] = ld16(kp + 32 * st);
  zn_bf16x8 qa[KST];
#pragma unroll
  for (int st = 0; st < KST; ++st) {
    u32x4 v = u32x4{0, 0, 0, 0};
    if (kn < G) v = ld16(a.q + ((size_t)r * a.n_heads + kvh * G + kn) * HD + 32 * st + 8 * kg);
    qa[st] = __builtin_bit_cast(zn_bf16x8, v);
  }
  __builtin_amdgcn_sched_barrier(0);
  const int L = Lraw + 1;
  if (chunk * ZN_ACHUNK >= L) return;
  f32x4_t c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int st = 0; st < KST; ++st) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[st], __builtin_bit_cast(zn_bf16x8, kk[st]), c, 0, 0, 0);
  float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};     // D: col = key l & 15, row = head 4 * (l >> 4) + reg
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int head = 4 * kg + reg;
    if (head < G && t < L) {
      const float sv = __fmul_rn(c[reg], a.scale);
      mx[reg] = sv;
      a.scores[((size_t)r * a.n_heads + kvh * G + head) * a.lcap + t] = sv;
    }
  }
  __shared__ float sm[4][G];
#pragma unroll
  for (int g = 0; g < G; ++g) { const float m = wave_max(kg == g / 4 ? mx[g % 4] : -INFINITY); if (lane == 0) sm[wave][g] = m; }
  __syncthreads();
  if (threadIdx.x < G) {
    const int g = threadIdx.x;
    a.cmax[((size_t)r * a.n_heads + kvh * G + g) * (a.lcap / ZN_ACHUNK) + chunk] = fmaxf(fmaxf(sm[0][g], sm[1][g]), fmaxf(sm[2][g], sm[3][g]));
  }
}

// ---- Pass 2 of ONE 512-key block of a (row, kv head) pair: the library's definition of the decode attention, shared by the launches below and
// by the whole-step kernel's attention role (zn_step_kernel.h), so that every path gives the same bits at every context length:
//   e_t = exp(s_t - m) with m = the running maximum through this block (fexp_u20 for the first 16 * floor(n / 16) keys of the block, libm exp
//   for the rest: the reference's SIMD / scalar split), row sum of the unrounded e_t, P = bf16(e_t);
//   P.V on the matrix cores: each of the workgroup's 8 waves contracts its 64 keys as two 16x16x32 steps per 16-column tile (A = P: rows =
//   the G heads, k = 32 consecutive keys; B = V: k = keys, columns = value dims), fp32 accumulate; the 8 per-wave partials are added in wave
//   order.  Blocks of one context are combined by the reference's recurrence acc = acc * f_j + pv_j, lsum = lsum * f_j + l_j in block order.
// (Rounds 1-3 accumulated P.V on the VALU, key by key per lane: 0.8 us per 32-wide value slice and block against 0.2 us for all 128 columns here.)
template <int HD> struct AttnV { unsigned v[2][8][HD / 32]; };    // lane (kn = l & 15, kg = l >> 4): [32-key step][key 8 kg + j of the step] x dims (HD / 16) kn .. + HD / 16

// e, P and the row sums of a block.  sc: the block's scores [G][512] in LDS; mnew: the running maximum through this block for the heads this
// lane evaluates (vsub + 4 q, vsub = lane & 3); p: P [G][512] (zero past nkeys); l_out[wave][g]: per-wave sums of e.
template <int G>
ZN_DEVINL void attn_block_probs(const float (*sc)[512], const float (&mnew)[(G + 3) / 4], int nkeys, int nvec, bf16_t (*p)[512], float (*l_out)[G], int wave, int lane) {
  constexpr int GL = (G + 3) / 4, NW = 8, NR = 4;
  const int vsub = lane & 3, vkey = lane >> 2;           // 16 keys per wave and round, 4 rounds: key = i * 128 + wave * 16 + vkey
  float lsum[GL];
#pragma unroll
  for (int q = 0; q < GL; ++q) lsum[q] = 0.f;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int base = i * (NW * 16), idx = base + wave * 16 + vkey;
    const bool ok = idx < nkeys;
#pragma unroll
    for (int q = 0; q < GL; ++q) {
      const int g = min(vsub + 4 * q, G - 1);
      const float x = __fsub_rn(ok ? sc[g][idx] : 0.f, mnew[q]);
      float ev;
      if (base + NW * 16 <= nvec) ev = ok ? zn_fexp_u20(x) : 0.f;          // whole round inside the SIMD part: fexp only (uniform branch)
      else ev = ok ? ((idx < nvec) ? zn_fexp_u20(x) : expf(x)) : 0.f;
      if (vsub + 4 * q < G) { lsum[q] += ev; p[vsub + 4 * q][idx] = f2bf(ev); }
    }
  }
#pragma unroll
  for (int q = 0; q < GL; ++q) {
    // lsum[q] of lane vsub belongs to head vsub + 4q: sum over the 16 lanes with that vsub, per wave
    float ls = row_stride4_sum(lsum[q]);
    ls = (readlane_f(ls, 0) + readlane_f(ls, 16)) + (readlane_f(ls, 32) + readlane_f(ls, 48));
    float l1 = row_stride4_sum(lsum[q]); l1 = (readlane_f(l1, 1) + readlane_f(l1, 17)) + (readlane_f(l1, 33) + readlane_f(l1, 49));
    float l2 = row_stride4_sum(lsum[q]); l2 = (readlane_f(l2, 2) + readlane_f(l2, 18)) + (readlane_f(l2, 34) + readlane_f(l2, 50));
    float l3 = row_stride4_sum(lsum[q]); l3 = (readlane_f(l3, 3) + readlane_f(l3, 19)) + (readlane_f(l3, 35) + readlane_f(l3, 51));
    if (lane == 0) {
      if (4 * q + 0 < G) l_out[wave][4 * q + 0] = ls;
      if (4 * q + 1 < G) l_out[wave][4 * q + 1] = l1;
      if (4 * q + 2 < G) l_out[wave][4 * q + 2] = l2;
      if (4 * q + 3 < G) l_out[wave][4 * q + 3] = l3;
    }
  }
}

// This wave's 64 keys of P.V on the matrix cores -> accw[wave][head][dim].  Column n of tile t is dim (HD / 16) n + t, so that a lane's row
// pieces (HD / 16 consecutive dims of 8 keys) transpose in registers (v_perm_b32) into the B operands: no LDS transpose of V.
template <int HD, int G>
ZN_DEVINL void attn_block_pv(const bf16_t (*p)[512], const AttnV<HD>& V, float (*accw)[G][HD], int wave, int lane) {
  static_assert(HD % 32 == 0 && G <= 16, "tile shape");
  constexpr int DPL = HD / 16;                           // dims per lane and key = tiles
  const int kn = lane & 15, kg = lane >> 4;
  f32x4 pacc[DPL];
#pragma unroll
  for (int t = 0; t < DPL; ++t) pacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    u32x4 pa = u32x4{0, 0, 0, 0};
    if (kn < G) pa = *(const u32x4*)&p[kn][wave * 64 + 32 * ks + 8 * kg];
#pragma unroll
    for (int t = 0; t < DPL; ++t) {
      const unsigned sel = (t & 1) ? 0x07060302u : 0x05040100u;
      u32x4 bt;
      bt.x = __builtin_amdgcn_perm(V.v[ks][1][t >> 1], V.v[ks][0][t >> 1], sel);
      bt.y = __builtin_amdgcn_perm(V.v[ks][3][t >> 1], V.v[ks][2][t >> 1], sel);
      bt.z = __builtin_amdgcn_perm(V.v[ks][5][t >> 1], V.v[ks][4][t >> 1], sel);
      bt.w = __builtin_amdgcn_perm(V.v[ks][7][t >> 1], V.v[ks][6][t >> 1], sel);
      pacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(zn_bf16x8, pa), __builtin_bit_cast(zn_bf16x8, bt), pacc[t], 0, 0, 0);
    }
  }
  // D: column = dim column kn, row = head 4 kg + reg
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int head = 4 * kg + reg;
    if (head < G) {
#pragma unroll
      for (int t = 0; t < DPL; ++t) accw[wave][head][DPL * kn + t] = pacc[t][reg];
    }
  }
}

// The value rows of this wave's 64 keys in AttnV's layout, by plain loads (rows clamped to the cache: the caller masks keys past the context).
template <int HD>
ZN_DEVINL void attn_block_load_v(AttnV<HD>& V, const bf16_t* vcol0, size_t kvrow, int key0, int row_max, int lane) {
  constexpr int DPL = HD / 16;
  const int kn = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bf16_t* src = vcol0 + (size_t)min(key0 + 32 * ks + 8 * kg + j, row_max) * kvrow + DPL * kn;
      if constexpr (DPL == 8) { const u32x4 x = ld16(src); V.v[ks][j][0] = x.x; V.v[ks][j][1] = x.y; V.v[ks][j][2] = x.z; V.v[ks][j][3] = x.w; }
      else if constexpr (DPL == 4) { const u32x2 x = *(const u32x2*)src; V.v[ks][j][0] = x.x; V.v[ks][j][1] = x.y; }
      else V.v[ks][j][0] = *(const unsigned*)src;
    }
}
// Rows past the context may be uninitialised memory (NaN bit patterns; 0 x NaN = NaN on the matrix cores): zero them.
template <int HD>
ZN_DEVINL void attn_block_mask_v(AttnV<HD>& V, int key0, int L, int lane) {
  const int kg = lane >> 4;
  if (key0 + 64 <= L) return;                             // wave-uniform
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (key0 + 32 * ks + 8 * kg + j >= L) {
#pragma unroll
        for (int q = 0; q < HD / 32; ++q) V.v[ks][j][q] = 0u;
      }
}

// One launch per decode step and layer for contexts of ONE block (FUSED; up to 512 keys, host-checked: attn_fused_for), two beyond:
// FUSED: one workgroup per (row, kv head) computes the block's scores into LDS itself (the arithmetic of attn_scores_kernel; K rows fetched
//   whole and staged per wave in padded LDS), then pass 2, normalises and stores - one launch on the latency-bound chain.
// SPLIT (!FUSED): after attn_scores_kernel, one workgroup per (row, kv head, 512-key block): running maximum from the chunk maxima of all
//   earlier chunks, the block's unnormalised P.V and e sums leave as write-through partials, and the last workgroup of a (row, kv head) to
//   arrive (ticket, no waiting) replays the recurrence over the blocks in order and normalises.
// Pass 1 of a 512-key block inside a one-launch shape: this wave's 16-key tiles (K rows in kk, requested whole - 16 lanes x 16 B = one 256-byte
// key row per 16-lane phase; a fragment-shaped request would touch 16 different rows per phase and bound the pass by the vector-memory pipe -
// and staged per wave in padded LDS, where the MFMA B fragments are read back conflict-free) -> S[head][key] = Q K^T as 16x16x32 tiles
// (attn_scores_kernel's operands and order) into sc, the wave's maxima into bm[wave].  nk: keys of the block inside the context.
template <int HD> struct AttnFusedK { static constexpr int KLD = HD + 8, LPK = HD / 8, KPL = 64 / LPK, NLD = 16 / KPL, TPW = 512 / 16 / 8; };
template <int HD>
ZN_DEVINL void attn_fused_load_k(u32x4 (&kk)[AttnFusedK<HD>::TPW][AttnFusedK<HD>::NLD], const bf16_t* kcol0, size_t kvrow, int tb, int row_max, int wave, int lane) {
  using C = AttnFusedK<HD>;
  const int kq = lane / C::LPK, kd = lane % C::LPK;
#pragma unroll
  for (int tl = 0; tl < C::TPW; ++tl)
#pragma unroll
    for (int i = 0; i < C::NLD; ++i) kk[tl][i] = ld16(kcol0 + kd * 8 + (size_t)min(tb + (tl * 8 + wave) * 16 + C::KPL * i + kq, row_max) * kvrow);
}
template <int HD, int G>
ZN_DEVINL void attn_fused_scores(const zn_bf16x8 (&qa)[HD / 32], const u32x4 (&kk)[AttnFusedK<HD>::TPW][AttnFusedK<HD>::NLD], bf16_t* kw, float (*sc)[512],
                                 float (*bm)[G], int nk, float scale, int wave, int lane) {
  using C = AttnFusedK<HD>;
  constexpr int KST = HD / 32;
  const int kn = lane & 15, kg = lane >> 4, kq = lane / C::LPK, kd = lane % C::LPK;
  float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int tl = 0; tl < C::TPW; ++tl) {
    const int tt = (tl * 8 + wave) * 16;
    if (tt < nk) {                                            // wave-uniform
#pragma unroll
      for (int i = 0; i < C::NLD; ++i) *(u32x4*)(kw + (C::KPL * i + kq) * C::KLD + kd * 8) = kk[tl][i];
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < KST; ++st) {
        const u32x4 bfrag = *(const u32x4*)(kw + kn * C::KLD + 32 * st + 8 * kg);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[st], __builtin_bit_cast(zn_bf16x8, bfrag), c, 0, 0, 0);
      }
      const int t = tt + kn;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int head = 4 * kg + reg;
        if (head < G && t < nk) {
          const float sv = __fmul_rn(c[reg], scale);
          mx[reg] = fmaxf(mx[reg], sv);
          sc[head][t] = sv;
        }
      }
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float m = wave_max(kg == g / 4 ? mx[g % 4] : -INFINITY);
    if (lane == 0) bm[wave][g] = m;
  }
}

// DS > 1 (batches of 3..8 utterances): DS workgroups per (row, kv head[, block]), each with all the scores / P of the pair and HD / DS of the value
//   columns - every output column keeps its arithmetic (a column of P.V depends on P and its own V column only), the K rows are read DS times
//   (the second reader of a pair sits 8 workgroups later, i.e. on the same XCD's L2 as dispatch goes), the V read per CU halves: a CU sustains
//   only a few tens of GB/s and 16 rows x 4 kv heads are 64 workgroups.
template <int HD, int G, bool FUSED, int DS = 1>
__global__ __launch_bounds__(512) void attn_block_kernel(AttnArgs a) {
  constexpr int NW = 8, GL = (G + 3) / 4, HDV = HD / DS;
  static_assert(HD % DS == 0 && HDV % 32 == 0, "value-column split");
  int kvh, r, jb, hv = 0;
  if constexpr (FUSED) {
    int pair = blockIdx.x;
    if constexpr (DS > 1) {
      const int q = blockIdx.x, npairs = gridDim.x / DS;
      if (npairs % 8 == 0) { pair = (q / (8 * DS)) * 8 + (q % 8); hv = (q / 8) % DS; }
      else { pair = q / DS; hv = q % DS; }
    }
    kvh = pair % a.n_heads_kv; r = pair / a.n_heads_kv; jb = 0;
  } else { kvh = blockIdx.x % a.n_heads_kv; hv = blockIdx.x / a.n_heads_kv; r = blockIdx.y / a.nbcap; jb = blockIdx.y % a.nbcap; }
  int nst = 0;
  auto stamp = [&]() { if (FUSED && a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[nst] = __builtin_amdgcn_s_memrealtime(); ++nst; };
  stamp();
  // length (and span) first; the requests below do not wait for them (clamped addresses, masked use)
  const int Lraw = a.lengths[r];
  const int Eraw = a.ext ? a.ext[r] : a.ext_scalar;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vsub = lane & 3;
  const size_t kvrow = (size_t)2 * a.n_heads_kv * HD;
  const bf16_t* kvr = a.kv + (size_t)r * a.max_len * kvrow;
  const int tb = jb * 512;
  __shared__ float s_sc[G][512];
  __shared__ __attribute__((aligned(16))) bf16_t s_p[G][512];
  __shared__ __attribute__((aligned(16))) float s_accw[NW][G][HDV];
  __shared__ float s_l[NW][G];
  __shared__ float s_bm[NW][G];
  AttnV<HDV> V;
  attn_block_load_v<HDV>(V, kvr + (size_t)(a.n_heads_kv + kvh) * HD + hv * HDV, kvrow, tb + wave * 64, a.max_len - 1, lane);
  float mnew[GL];
  int L, nb;
  if constexpr (FUSED) {
    // ---- pass 1 in LDS: S[head][key] = Q K^T as 16x16x32 tiles (attn_scores_kernel's operands and order)
    static_assert(HD % 32 == 0 && G <= 16, "fused attention tile shape");
    constexpr int KST = HD / 32;
    const int kn = lane & 15, kg = lane >> 4;
    zn_bf16x8 qa[KST];
#pragma unroll
    for (int st = 0; st < KST; ++st) {
      u32x4 v = u32x4{0, 0, 0, 0};
      if (kn < G) v = ld16(a.q + ((size_t)r * a.n_heads + kvh * G + kn) * HD + 32 * st + 8 * kg);
      qa[st] = __builtin_bit_cast(zn_bf16x8, v);
    }
    using KC = AttnFusedK<HD>;
    __shared__ __attribute__((aligned(16))) bf16_t s_k[NW][16 * KC::KLD];
    u32x4 kk[KC::TPW][KC::NLD];
    attn_fused_load_k<HD>(kk, kvr + (size_t)kvh * HD, kvrow, 0, a.max_len - 1, wave, lane);
    __builtin_amdgcn_sched_barrier(0);
    L = Lraw + 1; nb = 1;
    stamp();
    attn_fused_scores<HD, G>(qa, kk, &s_k[wave][0], s_sc, s_bm, min(L, 512), a.scale, wave, lane);
    stamp();
    __syncthreads();
    stamp();
#pragma unroll
    for (int q = 0; q < GL; ++q) {
      const int g = min(vsub + 4 * q, G - 1);
      float bm = -INFINITY;
#pragma unroll
      for (int w = 0; w < NW; ++w) bm = fmaxf(bm, s_bm[w][g]);
      mnew[q] = bm;
    }
  } else {
    // the block's scores (attn_scores_kernel) and the chunk maxima: requested before the length is known
    const int cstride = a.lcap / ZN_ACHUNK;
    float scv[G];
#pragma unroll
    for (int g = 0; g < G; ++g) scv[g] = a.scores[((size_t)r * a.n_heads + kvh * G + g) * a.lcap + tb + tid];     // tb + tid < 512 * (jb + 1) <= lcap
    const float* crow[GL];
    float cmn[GL];
#pragma unroll
    for (int q = 0; q < GL; ++q) {
      crow[q] = a.cmax + ((size_t)r * a.n_heads + kvh * G + min(vsub + 4 * q, G - 1)) * cstride;
      cmn[q] = crow[q][8 * jb + (lane >> 3)];              // chunk < 8 * (jb + 1) <= lcap / 64
    }
    __builtin_amdgcn_sched_barrier(0);
    L = Lraw + 1; nb = (L + 511) >> 9;
    if (jb >= nb) return;                                 // the grid covers the capacity; blocks past the context do nothing
#pragma unroll
    for (int g = 0; g < G; ++g) s_sc[g][tid] = scv[g];
    const int nchunks = (L + ZN_ACHUNK - 1) / ZN_ACHUNK;
    // running max before this block = max over every earlier chunk maximum (lanes with equal lane & 3 hold one head); the block's own
    // maximum = max over its <= 8 chunk maxima (lanes 8c .. 8c+7 hold chunk c)
#pragma unroll
    for (int q = 0; q < GL; ++q) {
      float pm = -INFINITY;
      for (int j0 = 0; j0 < jb; j0 += 8) {                 // eight requests in flight (chunks < 8 * jb all exist: jb < nb)
        float t8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t8[u] = crow[q][8 * min(j0 + u, jb - 1) + (lane >> 3)];
#pragma unroll
        for (int u = 0; u < 8; ++u) pm = fmaxf(pm, t8[u]);
      }
      float bm = fmaxf(pm, (8 * jb + (lane >> 3) < nchunks) ? cmn[q] : -INFINITY);
      bm = fmaxf(bm, dpp_mov<ZN_DPP_ROR4>(bm));
      bm = fmaxf(bm, dpp_mov<ZN_DPP_ROR8>(bm));
      bm = fmaxf(bm, __shfl_xor(bm, 16));
      bm = fmaxf(bm, __shfl_xor(bm, 32));
      mnew[q] = bm;
    }
    __syncthreads();                                      // the scores are in LDS
  }
  attn_block_mask_v<HDV>(V, tb + wave * 64, L, lane);
  int E = Eraw > 0 ? Eraw : L;
  if (E < L) E = L;
  const int nkeys = min(512, L - tb), nvec = min(512, E - tb) & ~15;
  attn_block_probs<G>(s_sc, mnew, nkeys, nvec, s_p, s_l, wave, lane);
  __syncthreads();
  stamp();
  attn_block_pv<HDV, G>(s_p, V, s_accw, wave, lane);
  __syncthreads();
  stamp();
  if constexpr (FUSED) {
    for (int o = tid; o < G * HDV; o += 512) {
      const int g = o / HDV, d = o % HDV;
      float v = 0.f, l = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += s_accw[w][g][d];
#pragma unroll
      for (int w = 0; w < NW; ++w) l += s_l[w][g];
      a.out[((size_t)r * a.n_heads + kvh * G + g) * HD + hv * HDV + d] = f2bf(__fmul_rn(v, 1.0f / l));
    }
    stamp();
  } else {
    constexpr int PSZ = G * HDV + G;                      // per (row, kv head, column part, block): its P.V columns and (a copy of) the e sums
    const int group = (r * a.n_heads_kv + kvh) * DS + hv;
    float* pp = a.part + ((size_t)group * a.nbcap + jb) * PSZ;
    for (int o = tid; o < G * HDV; o += 512) {
      const int g = o / HDV, d = o % HDV;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += s_accw[w][g][d];
      st_wt(pp + o, v);
      if (d == 0) {
        float l = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) l += s_l[w][g];
        st_wt(pp + G * HDV + g, l);
      }
    }
    __builtin_amdgcn_s_waitcnt(0);                        // stores acknowledged before the ticket
    __syncthreads();
    __shared__ int s_last;
    if (tid == 0) {
      const int t = __hip_atomic_fetch_add(a.tickets + group, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (t == nb - 1);
      if (s_last) __hip_atomic_store(a.tickets + group, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    // the reference's recurrence over the blocks, in order: acc = acc * f_j + pv_j, lsum = lsum * f_j + sum_j,
    // f_j = exp(m_{j-1} - m_j) with m_j the running maximum through block j (from the chunk maxima of pass 1)
    const int cstride = a.lcap / ZN_ACHUNK, nchunks = (L + ZN_ACHUNK - 1) / ZN_ACHUNK;
    __shared__ float s_bmax[G][32];                       // block maxima per head (nbcap <= 32: 16384 keys)
    if (tid < G * 32 && (tid & 31) < nb) {
      const int g = tid >> 5, d = tid & 31;
      const float* crq = a.cmax + ((size_t)r * a.n_heads + kvh * G + g) * cstride;
      float t8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t8[u] = crq[min(8 * d + u, nchunks - 1)];
      float bm = -INFINITY;
#pragma unroll
      for (int u = 0; u < 8; ++u) if (8 * d + u < nchunks) bm = fmaxf(bm, t8[u]);
      s_bmax[g][d] = bm;
    }
    __syncthreads();
    const float* pb = a.part + (size_t)group * a.nbcap * PSZ;
    for (int o = tid; o < G * HDV; o += 512) {
      const int g = o / HDV;
      float tot = 0.f, lt = 0.f, mprev = -INFINITY;
      for (int j0 = 0; j0 < nb; j0 += 8) {
        // eight blocks' partials requested at once (one memory round trip per eight blocks, not per block)
        float pv[8], pl[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = min(j0 + u, nb - 1);
          pv[u] = ld_wt(pb + (size_t)j * PSZ + o);
          pl[u] = ld_wt(pb + (size_t)j * PSZ + G * HDV + g);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + u;
          if (j < nb) {
            const float mj = fmaxf(mprev, s_bmax[g][j]);
            const float fj = (j == 0) ? 0.f : expf(mprev - mj);
            mprev = mj;
            tot = __fadd_rn(__fmul_rn(tot, fj), pv[u]);
            lt = __fadd_rn(__fmul_rn(lt, fj), pl[u]);
          }
        }
      }
      a.out[((size_t)r * a.n_heads + kvh * G + g) * HD + hv * HDV + (o % HDV)] = f2bf(__fmul_rn(tot, 1.0f / lt));
    }
  }
}

// (Measured and dropped, round 4: a one-launch shape that walks TWO blocks per workgroup - scores, e / P, P.V per block, the recurrence in the
// split shape's arithmetic; bit-identical - for 5..16 rows at 513..1024 keys: 1.699 vs 1.697 ms per batch-8 step.  One CU then pulls the K rows
// of 1024 keys, which costs what the second launch and the ticket do.)

// ------------------------------------------------------------------------------------------------ embedding
struct EmbedArgs {
  const bf16_t* const* tables;  // device array [n_q] of [vocab_embed][d]
  const int* codes;             // code(b,i) = codes[b*sb + i*si + col]
  const int* col_dev;           // optional device scalar added to the index (current column), else col
  int sb, si, col, n_q, d, batch, vocab_embed;
  bf16_t* out;                  // [2*batch or batch][d]
  int dup;                      // 1: also write row b + batch (CFG duplicate, generation_utils.py:192)
};
#define ZN_EMBED_MAXQ 16
// Sum of the n_q codebook embeddings of utterance b (codes clamped into the table) -> out rows b (and b + batch).
ZN_DEVINL void embed_row(const EmbedArgs& a, int b, const int (&code)[ZN_EMBED_MAXQ]) {
  for (int k = threadIdx.x * 8; k < a.d; k += 256 * 8) {
    u32x4 v[ZN_EMBED_MAXQ];
#pragma unroll
    for (int i = 0; i < ZN_EMBED_MAXQ; ++i) v[i] = ld16(a.tables[min(i, a.n_q - 1)] + (size_t)code[i] * a.d + k);
    float acc[8];
#pragma unroll
    for (int i = 0; i < ZN_EMBED_MAXQ; ++i) {
      if (i < a.n_q) {
        const float f[8] = {lo_f(v[i].x), hi_f(v[i].x), lo_f(v[i].y), hi_f(v[i].y), lo_f(v[i].z), hi_f(v[i].z), lo_f(v[i].w), hi_f(v[i].w)};
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = (i == 0) ? f[e] : bfround(acc[e] + f[e]);  // python sum(): bf16 adds
      }
    }
    u32x4 o;
    o.x = pack2(acc[0], acc[1]); o.y = pack2(acc[2], acc[3]); o.z = pack2(acc[4], acc[5]); o.w = pack2(acc[6], acc[7]);
    *(u32x4*)(a.out + (size_t)b * a.d + k) = o;
    if (a.dup) *(u32x4*)(a.out + (size_t)(b + a.batch) * a.d + k) = o;
  }
}
__global__ __launch_bounds__(256) void embed_kernel(EmbedArgs a) {
  const int b = blockIdx.x;
  const int col = a.col_dev ? *a.col_dev : a.col;
  // all codes, then all table rows, are requested before the first add (three memory round trips on the step's
  // launch-bound tail instead of one per codebook); indices past n_q repeat the last codebook and are not added
  int code[ZN_EMBED_MAXQ];
#pragma unroll
  for (int i = 0; i < ZN_EMBED_MAXQ; ++i) {
    const int c = a.codes[(size_t)b * a.sb + (size_t)min(i, a.n_q - 1) * a.si + col];
    code[i] = c < 0 ? 0 : (c >= a.vocab_embed ? a.vocab_embed - 1 : c);
  }
  embed_row(a, b, code);
}

// rows of a [rows][S][d] tensor at position s -> x [rows][d]
__global__ void gather_pos_kernel(const bf16_t* hidden, bf16_t* x, int S, int s, int d) {
  const int r = blockIdx.x;
  for (int k = threadIdx.x * 8; k < d; k += blockDim.x * 8)
    *(u32x4*)(x + (size_t)r * d + k) = *(const u32x4*)(hidden + ((size_t)r * S + s) * d + k);
}
// x [rows][d] -> rows of a [rows][S][d] tensor at position s
__global__ void scatter_pos_kernel(const bf16_t* x, bf16_t* hidden, int S, int s, int d) {
  const int r = blockIdx.x;
  for (int k = threadIdx.x * 8; k < d; k += blockDim.x * 8)
    *(u32x4*)(hidden + ((size_t)r * S + s) * d + k) = *(const u32x4*)(x + (size_t)r * d + k);
}
// standalone nn.LayerNorm: one wave per row; the row (d <= 4096) and the affine parameters are requested once, up
// front, and stay in registers through both statistics passes (one memory round trip instead of three)
__global__ __launch_bounds__(64) void layernorm_kernel(const bf16_t* x, const bf16_t* w, const bf16_t* b, bf16_t* out, int d, float eps) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const bf16_t* xr = x + (size_t)r * d;
  constexpr int MV = 8;
  u32x4 v[MV], g[MV], bb[MV];
#pragma unroll
  for (int j = 0; j < MV; ++j) {
    const int k = min(lane * 8 + j * 512, d - 8);          // clamped, not masked: no exec branch between the requests
    v[j] = ld16(xr + k); g[j] = ld16(w + k); bb[j] = ld16(b + k);
  }
  __builtin_amdgcn_sched_barrier(0);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < MV; ++j)
    if (lane * 8 + j * 512 < d) s += lo_f(v[j].x) + hi_f(v[j].x) + lo_f(v[j].y) + hi_f(v[j].y) + lo_f(v[j].z) + hi_f(v[j].z) + lo_f(v[j].w) + hi_f(v[j].w);
  const float mean = wave_sum(s) / (float)d;
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < MV; ++j)
    if (lane * 8 + j * 512 < d) {
      const float f[8] = {lo_f(v[j].x), hi_f(v[j].x), lo_f(v[j].y), hi_f(v[j].y), lo_f(v[j].z), hi_f(v[j].z), lo_f(v[j].w), hi_f(v[j].w)};
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float dd = f[e] - mean; ss += dd * dd; }
    }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)d + eps);
#pragma unroll
  for (int j = 0; j < MV; ++j) {
    const int k = lane * 8 + j * 512;
    if (k < d) {
      u32x4 o;
      o.x = pack2((lo_f(v[j].x) - mean) * rstd * lo_f(g[j].x) + lo_f(bb[j].x), (hi_f(v[j].x) - mean) * rstd * hi_f(g[j].x) + hi_f(bb[j].x));
      o.y = pack2((lo_f(v[j].y) - mean) * rstd * lo_f(g[j].y) + lo_f(bb[j].y), (hi_f(v[j].y) - mean) * rstd * hi_f(g[j].y) + hi_f(bb[j].y));
      o.z = pack2((lo_f(v[j].z) - mean) * rstd * lo_f(g[j].z) + lo_f(bb[j].z), (hi_f(v[j].z) - mean) * rstd * hi_f(g[j].z) + hi_f(bb[j].z));
      o.w = pack2((lo_f(v[j].w) - mean) * rstd * lo_f(g[j].w) + lo_f(bb[j].w), (hi_f(v[j].w) - mean) * rstd * hi_f(g[j].w) + hi_f(bb[j].w));
      *(u32x4*)(out + (size_t)r * d + k) = o;
    }
  }
}
__global__ void add_lengths_kernel(int* lengths, int rows, int n) {
  if (threadIdx.x < rows) lengths[threadIdx.x] += n;
}

// ------------------------------------------------------------------------------------------------ sampler
struct GenState {      // device-resident loop state (model.py:439-465)
  int offset;          // column of delayed_codes that feeds the next step (model.py:474: offset-1 after +=1)
  int step;            // completed loop steps
  int all_done;        // (remaining_steps <= 0).all()
  int force_eos_step;  // test hook, -1 = off
  float eos_bias;      // test/bench hook added to the codebook-0 EOS logit
  int pad[3];
};

struct FrameArgs {
  GenState* st;
  int* codes; int t_total, batch, n_q, eos_id, mask_id;
  const int* tokens;   // raw sampled [B][n_q]
  int* remaining;      // [B]
  int* stopping;       // [B]
  int* lengths; int rows;
  int first;           // 1: model.py:423-431 (first frame after prefill: plain write-where-unknown)
  const int* override; // test hook: raw tokens [calls][B][n_q] replacing the sampled ones (call 0 = first frame)
  int override_calls;
};
#define ZN_FRAME_MAXQ 16
#define ZN_TAIL_MAXB 64
struct SampleArgs {
  const float* raw;        // [2B][n_q*V] bf16-valued fp32 from the heads GEMV (mix=1) or [B][n_q][V] final logits
  int mix; float cfg_scale;
  int apply_bias;          // loop steps: logit_bias (model.py:433-437,476)
  int batch, n_q, V, eos_id;
  const int* codes; int t_total;   // delayed codes [B][n_q][t_total] (penalty history), or
  const int* recent; int window;   // explicit recent tokens [B][n_q][window]
  int ctx;                         // min(max_new_tokens, 100) (model.py:463)
  int use_penalty; float penalty; int pen_window;
  float temperature, top_p; int top_k; float min_p, linear, conf, quad;
  unsigned long long seed, draw;
  GenState* st;            // may be NULL (op mode)
  float* logits_out;       // [B][n_q][V] logits as consumed by the sampler (after bias), may be NULL
  float* probs_out;        // optional filtered probabilities
  int* tokens;             // [B][n_q]
  // Loop steps: the last workgroup of the launch to finish (arrival ticket, nobody waits) also runs the step's bookkeeping
  // (frame_update_kernel's body) and the NEXT step's embedding (embed_kernel's body, into em.out) - two launches fewer on the
  // step's launch-bound tail.  ticket == NULL: off (single ops, the first frame).
  int* ticket;
  FrameArgs fr;
  EmbedArgs em;            // codes/col fields unused: the codes come from the bookkeeping just done
};

ZN_DEVINL void block_argmax(float v, int i, float* sv, int* si, float& bv, int& bi) {
  // first-max-wins (torch CPU argmax returns the lowest index among ties)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o);
    const int oi = __shfl_xor(i, o);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { sv[wave] = v; si[wave] = i; }
  __syncthreads();
  bv = sv[0]; bi = si[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
    if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
}
ZN_DEVINL float block_sum(float v, float* sv) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sv[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sv[w];
  return t;
}
ZN_DEVINL float block_max(float v, float* sv) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sv[wave] = v;
  __syncthreads();
  float t = sv[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t = fmaxf(t, sv[w]);
  return t;
}
ZN_DEVINL unsigned long long zn_mix64(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// TAIL: called by the last workgroup of sample_kernel (tokens of the other workgroups arrive by sc1 loads); s_code[b][cb] receives
// the codes of the column just written (= what the next step embeds).
template <bool TAIL>
ZN_DEVINL void frame_update_body(const FrameArgs& a, int (*s_code)[ZN_FRAME_MAXQ]) {
  // 16 lanes per utterance, one per codebook: every cell, token and counter is requested at once (the per-codebook loop
  // of the reference becomes one memory round trip); lanes of one utterance share a wave, so all of them have read the
  // counters before lane 0 of the group rewrites them.  n_q <= 16 is checked by zn_create.
  const int o = a.st->offset, stp = a.st->step;
  __shared__ int s_done;
  if (threadIdx.x == 0) s_done = 1;
  __syncthreads();
  const int cb = threadIdx.x & (ZN_FRAME_MAXQ - 1);
  for (int b = threadIdx.x / ZN_FRAME_MAXQ; b < a.batch; b += blockDim.x / ZN_FRAME_MAXQ) {
    const int* tk = a.tokens + b * a.n_q;
    const int call = a.first ? 0 : stp + 1;
    const bool ovr = a.override && call < a.override_calls;
    if (ovr) tk = a.override + ((size_t)call * a.batch + b) * a.n_q;
    const int cbc = min(cb, a.n_q - 1);
    const int col = a.first ? o : o + 1;
    const bool in_range = cb < a.n_q && col < a.t_total;
    int* cell = a.codes + ((size_t)b * a.n_q + cbc) * a.t_total + min(col, a.t_total - 1);
    int tok, tok0;
    if (TAIL && !ovr) {
      tok = __hip_atomic_load(tk + cbc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      tok0 = __hip_atomic_load(tk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else { tok = tk[cbc]; tok0 = tk[0]; }
    const int cur = *cell;
    if (a.first) {                                           // model.py:423-431: plain write-where-unknown
      if (in_range && cur == -1) *cell = tok;
      continue;
    }
    // model.py:483-497 + tensor_ops.py:155-211
    int rem = a.remaining[b];
    int stop = a.stopping[b];
    if (tok0 == a.eos_id) { rem = rem < a.n_q ? rem : a.n_q; stop = 1; }
    int eos_idx = a.n_q - rem; if (eos_idx > a.n_q - 1) eos_idx = a.n_q - 1;
    int t = tok;
    if (stop && cb < eos_idx) t = a.mask_id; else if (stop && cb == eos_idx) t = a.eos_id;
    if (in_range && cur == -1) *cell = t;                    // tensor_ops.py:42-49
    if (TAIL) s_code[b][cb] = (in_range && cur == -1) ? t : cur;
    rem -= 1;                                                // tensor_ops.py:87
    if (cb == 0) {
      a.remaining[b] = rem; a.stopping[b] = stop;
      if (rem > 0) atomicAnd(&s_done, 0);
    }
  }
  __syncthreads();
  if (!a.first) {
    for (int r = threadIdx.x; r < a.rows; r += blockDim.x) a.lengths[r] += 1;   // tensor_ops.py:85-86
    if (threadIdx.x == 0) { a.st->offset = o + 1; a.st->step = stp + 1; a.st->all_done = s_done; }
  }
}
__global__ __launch_bounds__(256) void frame_update_kernel(FrameArgs a) { frame_update_body<false>(a, nullptr); }

#define ZN_SAMPLE_MAXV 2048
__global__ __launch_bounds__(256) void sample_kernel(SampleArgs a) {
  const int cb = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int V = a.V;
  __shared__ float sp[ZN_SAMPLE_MAXV];
  __shared__ int sidx[ZN_SAMPLE_MAXV];
  __shared__ float sv[4];
  __shared__ int si[4];
  // every input is requested before the first use (clamped indices, masked use): the kernel sits on the step's
  // launch-bound tail, where one memory round trip per loop iteration used to cost several microseconds
  const int o = a.st ? a.st->offset : 0;
  const float st_bias = a.st ? a.st->eos_bias : 0.f;
  const bool st_force = a.st ? (a.st->force_eos_step == a.st->step) : false;
  constexpr int IT = ZN_SAMPLE_MAXV / 256;
  float cc[IT], uu[IT];
  const float* rc = a.raw + ((size_t)b * a.n_q + cb) * V;
  const float* ru = a.raw + ((size_t)(b + (a.mix ? a.batch : 0)) * a.n_q + cb) * V;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int i = min(tid + it * 256, V - 1);
    cc[it] = rc[i];
    uu[it] = ru[i];
  }
  // ---- logits: CFG mix (model.py:231-232), logit bias (model.py:433-437,476)
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int i = tid + it * 256;
    if (i < V) {
      float l = a.mix ? __fadd_rn(uu[it], __fmul_rn(__fsub_rn(cc[it], uu[it]), a.cfg_scale)) : cc[it];
      if (a.apply_bias && i == a.eos_id) {
        if (cb == 0) {
          l = __fadd_rn(l, -0.6931471824645996f);  // -log(2) in fp32
          if (a.st) {
            l += st_bias;
            if (st_force) l = 1.0e4f;
          }
        } else l = -INFINITY;
      }
      if (a.logits_out) a.logits_out[((size_t)b * a.n_q + cb) * V + i] = l;
      sp[i] = l;
    }
  }
  __syncthreads();
  // ---- repetition penalty (sampling.py:159-163): factor = penalty^(#occurrences in the last `window` tokens)
  if (a.use_penalty) {
    int nw = 0;
    const int* hist = nullptr;
    if (a.recent) { nw = a.window < a.pen_window ? a.window : a.pen_window; hist = a.recent + ((size_t)b * a.n_q + cb) * a.window + (a.window - nw); }
    else if (a.codes) {
      int avail = o + 1 < a.ctx ? o + 1 : a.ctx;   // columns [max(0,o+1-ctx), o]
      nw = avail < a.pen_window ? avail : a.pen_window;
      hist = a.codes + ((size_t)b * a.n_q + cb) * a.t_total + (o + 1 - nw);
    }
    // the window's tokens are fetched once, together, into LDS (the default window is 2); longer windows re-read memory
    constexpr int HW = 16;
    __shared__ int s_hist[HW];
    if (tid < HW && tid < nw) s_hist[tid] = hist[tid];
    __syncthreads();
    if (tid == 0) {
      auto tokat = [&](int w) -> int { const int g = w < HW ? s_hist[w] : hist[w]; return g > V - 1 ? V - 1 : g; };
      for (int w = 0; w < nw; ++w) {
        const int g = tokat(w);
        if (g < 0) continue;
        // scatter_reduce(prod) builds the factor first; apply once per distinct token
        bool seen = false;
        for (int w2 = 0; w2 < w; ++w2) if (tokat(w2) == g) seen = true;
        if (seen) continue;
        float f = 1.f;
        for (int w2 = w; w2 < nw; ++w2) if (tokat(w2) == g) f = __fmul_rn(f, a.penalty);
        const float l = sp[g];
        sp[g] = (l <= 0.f) ? __fmul_rn(l, f) : __fdiv_rn(l, f);
      }
    }
    __syncthreads();
  }
  int tok;
  if (!(a.temperature > 0.f)) {
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int i = tid; i < V; i += 256) { const float v = sp[i]; if (v > bv) { bv = v; bi = i; } }
    float rv; int ri;
    block_argmax(bv, bi, sv, si, rv, ri);
    tok = ri == 0x7fffffff ? 0 : ri;
  } else {
    // ---- softmax(logits / T) (sampling.py:217)
    float mx = -INFINITY;
    for (int i = tid; i < V; i += 256) { sp[i] = sp[i] / a.temperature; mx = fmaxf(mx, sp[i]); }
    mx = block_max(mx, sv);
    float s = 0.f;
    for (int i = tid; i < V; i += 256) { const float e = expf(sp[i] - mx); sp[i] = e; s += e; }
    s = block_sum(s, sv);
    for (int i = tid; i < V; i += 256) sp[i] = sp[i] / s;
    __syncthreads();
    if (a.linear > 0.f) {  // apply_unified (sampling.py:60-63)
      float ent = 0.f;
      for (int i = tid; i < V; i += 256) { const float lp = logf(fmaxf(sp[i], 1e-20f)); ent += sp[i] * lp; }
      ent = -block_sum(ent, sv);
      float m2 = -INFINITY;
      for (int i = tid; i < V; i += 256) {
        const float lp = logf(fmaxf(sp[i], 1e-20f));
        const float raw = lp * (a.linear + ent * a.conf) - lp * lp * a.quad;
        sp[i] = raw; m2 = fmaxf(m2, raw);
      }
      m2 = block_max(m2, sv);
      float s2 = 0.f;
      for (int i = tid; i < V; i += 256) { const float e = expf(sp[i] - m2); sp[i] = e; s2 += e; }
      s2 = block_sum(s2, sv);
      for (int i = tid; i < V; i += 256) sp[i] = sp[i] / s2;
      __syncthreads();
    }
    if (a.top_p > 0.f || a.top_k > 0) {
      // descending bitonic sort of (prob, index) padded to 2048 (sampling.py:93 torch.sort, :77 topk)
      for (int i = tid; i < ZN_SAMPLE_MAXV; i += 256) { sidx[i] = i; if (i >= V) sp[i] = -1.f; }
      __syncthreads();
      for (int k = 2; k <= ZN_SAMPLE_MAXV; k <<= 1) {
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
          for (int i = tid; i < ZN_SAMPLE_MAXV; i += 256) {
            const int p = i ^ jj;
            if (p > i) {
              const bool desc = (i & k) == 0;
              const float x = sp[i], y = sp[p];
              const int xi = sidx[i], yi = sidx[p];
              const bool x_first = (x > y) || (x == y && xi < yi);  // order wanted for a descending run
              if (desc ? !x_first : x_first) { sp[i] = y; sp[p] = x; sidx[i] = yi; sidx[p] = xi; }
            }
          }
          __syncthreads();
        }
      }
      if (a.top_p > 0.f) {  // apply_top_p (sampling.py:93-99): sequential fp32 cumsum like torch.cumsum on CPU
        if (tid == 0) {
          float cum = 0.f;
          for (int i = 0; i < V; ++i) { const float p = sp[i]; cum += p; if (cum - p > a.top_p) sp[i] = 0.f; }
        }
        __syncthreads();
        float s3 = 0.f;
        for (int i = tid; i < V; i += 256) s3 += sp[i];
        s3 = block_sum(s3, sv);
        for (int i = tid; i < V; i += 256) sp[i] = sp[i] / s3;
        __syncthreads();
      }
      if (a.top_k > 0) {  // apply_top_k (sampling.py:77-81): still sorted descending (zeros sink consistently)
        const int kk = a.top_k < V ? a.top_k : V;
        // after top_p the array is no longer strictly sorted only among zeros; the k-th largest is sp[kk-1]
        const float pivot = sp[kk - 1];
        __syncthreads();
        float s4 = 0.f;
        for (int i = tid; i < V; i += 256) { if (sp[i] < pivot) sp[i] = 0.f; s4 += sp[i]; }
        s4 = block_sum(s4, sv);
        for (int i = tid; i < V; i += 256) sp[i] = sp[i] / s4;
        __syncthreads();
      }
    } else {
      for (int i = tid; i < V; i += 256) sidx[i] = i;
      __syncthreads();
    }
    if (a.min_p > 0.f) {  // apply_min_p (sampling.py:123-127)
      float m3 = 0.f;
      for (int i = tid; i < V; i += 256) m3 = fmaxf(m3, sp[i]);
      m3 = block_max(m3, sv);
      float s5 = 0.f;
      for (int i = tid; i < V; i += 256) { if (sp[i] < a.min_p * m3) sp[i] = 0.f; s5 += sp[i]; }
      s5 = block_sum(s5, sv);
      for (int i = tid; i < V; i += 256) sp[i] = sp[i] / s5;
      __syncthreads();
    }
    if (a.probs_out) for (int i = tid; i < V; i += 256) a.probs_out[((size_t)b * a.n_q + cb) * V + sidx[i]] = sp[i];
    // Gumbel-max: argmax(p / Exp(1)) (sampling.py:28-30); counter-based RNG stream
    float bv = -1.f; int bi = 0x7fffffff;
    for (int i = tid; i < V; i += 256) {
      const int tokid = sidx[i];
      unsigned long long h = zn_mix64(a.seed ^ zn_mix64(a.draw + (a.st ? (unsigned long long)a.st->step : 0ull)));
      h = zn_mix64(h + 0x9E3779B97F4A7C15ull * (unsigned long long)(((size_t)b * a.n_q + cb) * V + tokid + 1));
      const float uu = ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
      const float q = -logf(uu);
      const float v = sp[i] / q;
      if (v > bv || (v == bv && tokid < bi)) { bv = v; bi = tokid; }
    }
    float rv; int ri;
    block_argmax(bv, bi, sv, si, rv, ri);
    tok = ri == 0x7fffffff ? 0 : ri;
  }
  if (!a.ticket) {
    if (tid == 0) a.tokens[b * a.n_q + cb] = tok;
    return;
  }
  // ---- the step's tail in the last workgroup to arrive (Guideline 16 R1: write-through token, drain, barrier, ticket; sc1 loads)
  __shared__ int s_last;
  __shared__ int s_code[ZN_TAIL_MAXB][ZN_FRAME_MAXQ];
  if (tid == 0) __hip_atomic_store(a.tokens + b * a.n_q + cb, tok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_s_waitcnt(0);                          // vmcnt(0): the token store is acknowledged before the ticket
  __syncthreads();
  if (tid == 0) {
    const int n = (int)(gridDim.x * gridDim.y);
    const int t = __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == n - 1);
    if (s_last) __hip_atomic_store(a.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_last) return;
  frame_update_body<true>(a.fr, s_code);
  __syncthreads();
  for (int ub = 0; ub < a.fr.batch; ++ub) {
    int code[ZN_EMBED_MAXQ];
#pragma unroll
    for (int i = 0; i < ZN_EMBED_MAXQ; ++i) {
      const int c = s_code[ub][min(i, a.em.n_q - 1)];
      code[i] = c < 0 ? 0 : (c >= a.em.vocab_embed ? a.em.vocab_embed - 1 : c);
    }
    embed_row(a.em, ub, code);
  }
}

// Batch 1, loop steps: the whole tail of a decode step in ONE workgroup - one wave per codebook samples from registers (17 logits per
// lane), then the bookkeeping and the next step's embedding follow behind a barrier instead of behind sample_kernel's token store,
// arrival ticket and re-read (three memory round trips on the step's launch-bound tail), and the bookkeeping's inputs are requested
// together with the penalty window.  Arithmetic and tie rules are sample_kernel's: its block reductions (per-thread partial in
// element order, wave_sum per wave, waves added in order) are reproduced with the four "virtual waves" j mod 4 of the 64-lane slots, so
// both kernels return the same token and the same probabilities.  Not here (zn_api.hip keeps sample_kernel for them): top-p / top-k
// (sorting), the unified sampler, penalty windows > 16, vocabularies > 1088, batches.
#define ZN_S1_IT 17
template <typename F> ZN_DEVINL float s1_sum(const float (&x)[ZN_S1_IT], int lane, int V, F keep) {
  float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < ZN_S1_IT; ++j) if (lane + 64 * j < V) part[j & 3] += keep(x[j]);
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) t += wave_sum(part[w]);
  return t;
}
ZN_DEVINL void s1_argmax(float& v, int& i) {          // first-max-wins over the wave; every lane gets the result
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o);
    const int oi = __shfl_xor(i, o);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}
__global__ __launch_bounds__(1024) void sample1_kernel(SampleArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, cb = tid >> 6;
  const int V = a.V, nq = a.n_q;
  __shared__ int s_tok[ZN_FRAME_MAXQ];
  __shared__ int s_code[ZN_FRAME_MAXQ];
  const FrameArgs& fr = a.fr;
  const int o = a.st->offset, stp = a.st->step;
  const float st_bias = a.st->eos_bias;
  const bool st_force = a.st->force_eos_step == stp;
  // ---- requests: this wave's logits; the penalty window; the bookkeeping's cell, counters and override token (lanes 0..15 of wave 0)
  float cc[ZN_S1_IT], uu[ZN_S1_IT];
  const bool on = cb < nq;
  const int cbq = on ? cb : nq - 1;
  const float* rc = a.raw + (size_t)cbq * V;
  const float* ru = a.raw + (size_t)((a.mix ? 1 : 0) * nq + cbq) * V;
#pragma unroll
  for (int j = 0; j < ZN_S1_IT; ++j) {
    const int i = min(lane + 64 * j, V - 1);
    cc[j] = rc[i];
    uu[j] = ru[i];
  }
  int nw = 0, hist_tok = -1;                               // (windows of at most 16 tokens: zn_api.hip)
  if (a.use_penalty && a.codes) {
    const int avail = o + 1 < a.ctx ? o + 1 : a.ctx;   // columns [max(0,o+1-ctx), o]
    nw = avail < a.pen_window ? avail : a.pen_window;
    if (lane < nw) hist_tok = a.codes[(size_t)cbq * a.t_total + (o + 1 - nw) + lane];
  }
  const int fcb = tid & (ZN_FRAME_MAXQ - 1), fcbc = min(fcb, nq - 1);
  const int col = o + 1;
  const bool in_range = fcb < nq && col < fr.t_total;
  int* cell = fr.codes + (size_t)fcbc * fr.t_total + min(col, fr.t_total - 1);
  const int call = stp + 1;
  const bool ovr = fr.override && call < fr.override_calls;
  int cur = 0, rem = 0, stop = 0, otok = 0, otok0 = 0;
  if (tid < ZN_FRAME_MAXQ) {
    cur = *cell; rem = fr.remaining[0]; stop = fr.stopping[0];
    if (ovr) { otok = fr.override[(size_t)call * nq + fcbc]; otok0 = fr.override[(size_t)call * nq]; }
  }
  int tok = 0;
  if (on) {
    // ---- logits: CFG mix (model.py:231-232), logit bias (model.py:433-437,476)
    float l[ZN_S1_IT];
#pragma unroll
    for (int j = 0; j < ZN_S1_IT; ++j) {
      const int i = lane + 64 * j;
      float x = a.mix ? __fadd_rn(uu[j], __fmul_rn(__fsub_rn(cc[j], uu[j]), a.cfg_scale)) : cc[j];
      if (a.apply_bias && i == a.eos_id) {
        if (cb == 0) {
          x = __fadd_rn(x, -0.6931471824645996f);
          x += st_bias;
          if (st_force) x = 1.0e4f;
        } else x = -INFINITY;
      }
      if (a.logits_out && i < V) a.logits_out[(size_t)cb * V + i] = x;
      l[j] = x;
    }
    // ---- repetition penalty (sampling.py:159-163): factor = penalty^(#occurrences in the window), once per distinct token
    for (int w = 0; w < nw; ++w) {
      int g = __shfl(hist_tok, w);
      if (g < 0) continue;
      g = g > V - 1 ? V - 1 : g;
      bool seen = false;
      float f = 1.f;
      for (int w2 = 0; w2 < nw; ++w2) {
        int g2 = __shfl(hist_tok, w2);
        g2 = g2 > V - 1 ? V - 1 : g2;
        if (g2 == g) { if (w2 < w) seen = true; else f = __fmul_rn(f, a.penalty); }
      }
      if (seen) continue;
#pragma unroll
      for (int j = 0; j < ZN_S1_IT; ++j)
        if (lane + 64 * j == g) l[j] = (l[j] <= 0.f) ? __fmul_rn(l[j], f) : __fdiv_rn(l[j], f);
    }
    if (!(a.temperature > 0.f)) {
      float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < ZN_S1_IT; ++j) { const int i = lane + 64 * j; if (i < V && l[j] > bv) { bv = l[j]; bi = i; } }
      s1_argmax(bv, bi);
      tok = bi == 0x7fffffff ? 0 : bi;
    } else {
      // ---- softmax(logits / T) (sampling.py:217), min_p (sampling.py:123-127), Gumbel-max (sampling.py:28-30)
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < ZN_S1_IT; ++j) { l[j] = l[j] / a.temperature; if (lane + 64 * j < V) mx = fmaxf(mx, l[j]); }
      mx = wave_max(mx);
#pragma unroll
      for (int j = 0; j < ZN_S1_IT; ++j) l[j] = expf(l[j] - mx);
      const float ssum = s1_sum(l, lane, V, [](float e) { return e; });
#pragma unroll
      for (int j = 0; j < ZN_S1_IT; ++j) l[j] = l[j] / ssum;
      if (a.min_p > 0.f) {
        float m3 = 0.f;
#pragma unroll
        for (int j = 0; j < ZN_S1_IT; ++j) if (lane + 64 * j < V) m3 = fmaxf(m3, l[j]);
        m3 = wave_max(m3);
        const float thr = a.min_p * m3;
#pragma unroll
        for (int j = 0; j < ZN_S1_IT; ++j) if (l[j] < thr) l[j] = 0.f;
        const float s5 = s1_sum(l, lane, V, [](float e) { return e; });
#pragma unroll
        for (int j = 0; j < ZN_S1_IT; ++j) l[j] = l[j] / s5;
      }
      if (a.probs_out) {
#pragma unroll
        for (int j = 0; j < ZN_S1_IT; ++j) { const int i = lane + 64 * j; if (i < V) a.probs_out[(size_t)cb * V + i] = l[j]; }
      }
      const unsigned long long h0 = zn_mix64(a.seed ^ zn_mix64(a.draw + (unsigned long long)stp));
      float bv = -1.f; int bi = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < ZN_S1_IT; ++j) {
        const int i = lane + 64 * j;
        if (i < V) {
          const unsigned long long h = zn_mix64(h0 + 0x9E3779B97F4A7C15ull * (unsigned long long)((size_t)cb * V + i + 1));
          const float u01 = ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
          const float v = l[j] / (-logf(u01));
          if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
        }
      }
      s1_argmax(bv, bi);
      tok = bi == 0x7fffffff ? 0 : bi;
    }
    if (lane == 0) { s_tok[cb] = tok; a.tokens[cb] = tok; }
  }
  __syncthreads();
  // ---- bookkeeping of the step (frame_update_body; model.py:483-497 + tensor_ops.py:155-211) for the one utterance
  if (tid < ZN_FRAME_MAXQ) {
    const int tk = ovr ? otok : s_tok[fcbc], tk0 = ovr ? otok0 : s_tok[0];
    if (tk0 == fr.eos_id) { rem = rem < nq ? rem : nq; stop = 1; }
    int eos_idx = nq - rem; if (eos_idx > nq - 1) eos_idx = nq - 1;
    int t = tk;
    if (stop && fcb < eos_idx) t = fr.mask_id; else if (stop && fcb == eos_idx) t = fr.eos_id;
    if (in_range && cur == -1) *cell = t;
    s_code[fcb] = (in_range && cur == -1) ? t : cur;
    rem -= 1;
    if (fcb == 0) {
      fr.remaining[0] = rem; fr.stopping[0] = stop;
      a.st->offset = o + 1; a.st->step = stp + 1; a.st->all_done = rem > 0 ? 0 : 1;
    }
  }
  if (tid < fr.rows) fr.lengths[tid] += 1;                  // tensor_ops.py:85-86
  __syncthreads();
  // ---- the next step's embedding
  if (tid < 256) {
    int code[ZN_EMBED_MAXQ];
#pragma unroll
    for (int i = 0; i < ZN_EMBED_MAXQ; ++i) {
      const int c = s_code[min(i, a.em.n_q - 1)];
      code[i] = c < 0 ? 0 : (c >= a.em.vocab_embed ? a.em.vocab_embed - 1 : c);
    }
    embed_row(a.em, 0, code);
  }
}

// ------------------------------------------------------------------------------------------------ bookkeeping
