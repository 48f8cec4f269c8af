// Hybrid backbone (zonos/backbone/_mamba_ssm.py:8-119) decode-step kernels: fused residual-add + LayerNorm of the
// mamba_ssm Block, and the single-token Mamba2 mixer pieces (causal conv window update + SiLU, SSD state update,
// gated RMSNorm).  The arithmetic of these pieces lives in third-party mamba_ssm 2.2.5 / causal_conv1d 1.5.2, absent
// from the reference tree: the kernels follow the published recurrence (Mamba-2, arXiv:2405.21060) with one bf16
// rounding per kernel output and bf16 conv/SSM state; the CPU restatement they are tested against is
// oracle/zonos_oracle.py (mamba2_step, add_norm) — parity with the reference itself is unpinned.
// All of it is HBM-bound byte moving: the SSM state (H x P x N bf16 per row and layer) is read and written once per
// step with 16-B accesses, reductions are wavefront shuffles.
#pragma once
#include "zn_common.h"

// ------------------------------------------------------------------------------------------------ add + LayerNorm
struct AddLnArgs {
  const bf16_t* h;      // [rows][d] mixer / MLP output of the previous sub-block
  bf16_t* res;          // [rows][d] residual stream, updated in place (has_res = 0: written only)
  const bf16_t *w, *b;  // [d]
  bf16_t* out;          // [rows][d] normalised
  int d, has_res, write_res;
  float eps;
  int rms;              // BackboneConfig.rms_norm (config.py:82): RMSNorm - no mean, out = s * rsqrt(mean(s^2) + eps) * w (+ b when given)
  int res32;            // BackboneConfig.residual_in_fp32 (config.py:83): `res` is float [rows][d] and keeps the unrounded sum
};
// s = h + res in fp32; res <- bf16(s) (or s itself, res32); out <- bf16(norm_fp32(s)), norm = LayerNorm (biased variance, two
// passes over registers) or RMSNorm: the mamba_ssm Block's fused add + norm (layer_norm_fn, _mamba_ssm.py:45-58,111-119).
// One workgroup per row, 8 elements per thread per sweep (d <= 4096).
__global__ __launch_bounds__(256) void add_ln_kernel(AddLnArgs a) {
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = a.d;
  constexpr int MAXV = 2;                      // sweeps of 256 threads x 8 elements
  float v[MAXV][8];
  int nv = 0;
  float sum = 0.f;
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int k = (it * 256 + tid) * 8;
    if (k < d) {
      const u32x4 hv = ld16(a.h + (size_t)r * d + k);
      float f[8] = {lo_f(hv.x), hi_f(hv.x), lo_f(hv.y), hi_f(hv.y), lo_f(hv.z), hi_f(hv.z), lo_f(hv.w), hi_f(hv.w)};
      float* res32 = (float*)a.res + (size_t)r * d + k;
      if (a.has_res) {
        float g[8];
        if (a.res32) {
          const f32x4 g0 = *(const f32x4*)res32, g1 = *(const f32x4*)(res32 + 4);
          g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
        } else {
          const u32x4 rv = ld16(a.res + (size_t)r * d + k);
          g[0] = lo_f(rv.x); g[1] = hi_f(rv.x); g[2] = lo_f(rv.y); g[3] = hi_f(rv.y); g[4] = lo_f(rv.z); g[5] = hi_f(rv.z); g[6] = lo_f(rv.w); g[7] = hi_f(rv.w);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = __fadd_rn(f[e], g[e]);
      }
      if (a.write_res) {
        if (a.res32) {
          *(f32x4*)res32 = f32x4{f[0], f[1], f[2], f[3]};
          *(f32x4*)(res32 + 4) = f32x4{f[4], f[5], f[6], f[7]};
        } else {
          u32x4 o;
          o.x = pack2(f[0], f[1]); o.y = pack2(f[2], f[3]); o.z = pack2(f[4], f[5]); o.w = pack2(f[6], f[7]);
          *(u32x4*)(a.res + (size_t)r * d + k) = o;
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[it][e] = f[e]; sum += f[e]; }
      nv = it + 1;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[it][e] = 0.f;
    }
  }
  __shared__ float red[2][4];
  sum = wave_sum(sum);
  if (lane == 0) red[0][wave] = sum;
  __syncthreads();
  const float mean = a.rms ? 0.f : ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)d;
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < MAXV; ++it)
    if (it < nv) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float c = v[it][e] - mean; ss += c * c; }
    }
  ss = wave_sum(ss);
  if (lane == 0) red[1][wave] = ss;
  __syncthreads();
  const float rstd = 1.0f / sqrtf(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)d + a.eps);
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int k = (it * 256 + tid) * 8;
    if (it < nv) {
      const u32x4 wv = ld16(a.w + k), bv = a.b ? ld16(a.b + k) : u32x4{0, 0, 0, 0};     // RMSNorm modules carry no bias
      const float wf[8] = {lo_f(wv.x), hi_f(wv.x), lo_f(wv.y), hi_f(wv.y), lo_f(wv.z), hi_f(wv.z), lo_f(wv.w), hi_f(wv.w)};
      const float bf[8] = {lo_f(bv.x), hi_f(bv.x), lo_f(bv.y), hi_f(bv.y), lo_f(bv.z), hi_f(bv.z), lo_f(bv.w), hi_f(bv.w)};
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[it][e] - mean) * rstd * wf[e] + bf[e];
      u32x4 ov;
      ov.x = pack2(o[0], o[1]); ov.y = pack2(o[2], o[3]); ov.z = pack2(o[4], o[5]); ov.w = pack2(o[6], o[7]);
      *(u32x4*)(a.out + (size_t)r * d + k) = ov;
    }
  }
}

// ------------------------------------------------------------------------------------------------ Mamba2 step
struct MambaArgs {
  const bf16_t* zx;       // [rows][d_in_proj] = [z (d_inner) | xBC (conv_dim) | dt (nheads)]  (in_proj output)
  bf16_t* conv_state;     // [rows][conv_dim][4]
  bf16_t* ssm_state;      // [rows][nheads][headdim][d_state]
  const bf16_t *conv_w;   // [conv_dim][4]  (conv1d.weight [conv_dim,1,4])
  const bf16_t *conv_b;   // [conv_dim]
  const bf16_t *dt_bias, *A_log, *D;   // [nheads]
  const bf16_t* norm_w;   // [d_inner]
  bf16_t* xbc;            // [rows][conv_dim] activated conv output
  bf16_t* y;              // [rows][d_inner]
  float* vg;              // [rows][d_inner] fp32 gated value y * silu(z) (RMSNormGated's first step), or NULL
  bf16_t* g;              // [rows][d_inner] gated-normalised
  int d_inner, conv_dim, nheads, d_state, ngroups, d_in_proj, rows;
  float eps;
};

// causal_conv1d_update (width 4): shift the window, append the new sample, out = silu(bias + sum_i w[i] * win[i]) with
// separately rounded fp32 multiplies and adds in tap order, one rounding to bf16.  One thread per channel.
__global__ __launch_bounds__(256) void mamba_conv_kernel(MambaArgs a) {
  const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (c >= a.conv_dim) return;
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
  bf16_t* sp = a.conv_state + ((size_t)r * a.conv_dim + c) * 4;
  const u32x2_t st = *(const u32x2_t*)sp;
  const u32x2_t wv = *(const u32x2_t*)(a.conv_w + (size_t)c * 4);
  const bf16_t xn = a.zx[(size_t)r * a.d_in_proj + a.d_inner + c];
  const float s0 = hi_f(st.x), s1 = lo_f(st.y), s2 = hi_f(st.y), s3 = bf2f(xn);
  u32x2_t ns;
  ns.x = (st.x >> 16) | (st.y << 16);
  ns.y = (st.y >> 16) | ((unsigned)xn << 16);
  *(u32x2_t*)sp = ns;
  float acc = bf2f(a.conv_b[c]);
  acc = __fadd_rn(acc, __fmul_rn(lo_f(wv.x), s0));
  acc = __fadd_rn(acc, __fmul_rn(hi_f(wv.x), s1));
  acc = __fadd_rn(acc, __fmul_rn(lo_f(wv.y), s2));
  acc = __fadd_rn(acc, __fmul_rn(hi_f(wv.y), s3));
  a.xbc[(size_t)r * a.conv_dim + c] = f2bf(acc / (1.0f + expf(-acc)));
}

// selective_state_update for one (row, head): h <- h * exp(dt A) + (B dt) x (each product and the sum rounded in fp32,
// state stored bf16), y = C . h_new + D x from the unrounded new state.  headdim 64; 256 threads: thread = (p = tid / 4,
// quarter of the state row); the state tile (64 x N bf16, 16 KB at N = 128) is read and written once, 16 B per access.
// RW rows per workgroup (both state tiles requested up front: 128 B per lane in flight).  Default cache policy: the
// non-temporal hint on this read-modify-write stream cost 15 % of the batch-8 step (measured), unlike on read-once weights.
template <int N, int RW>
__global__ __launch_bounds__(256) void mamba_ssm_kernel(MambaArgs a) {
  constexpr int P = 64, NT = N / 4, NV = NT / 8;     // state elements per thread, 16-B vectors per thread
  const int h = blockIdx.x, r0 = blockIdx.y * RW, tid = threadIdx.x;
  const int p = tid >> 2, q = tid & 3;
  const int grp = h / (a.nheads / a.ngroups);
  bf16_t* sp[RW];
  u32x4 sv[RW][NV], bv[RW][NV], cv[RW][NV];
  float x[RW], dt[RW], zg[RW];
#pragma unroll
  for (int w = 0; w < RW; ++w) {
    const int r = min(r0 + w, a.rows - 1);            // clamped; a duplicate row recomputes and rewrites identical values
    sp[w] = a.ssm_state + (((size_t)r * a.nheads + h) * P + p) * N + q * NT;
#pragma unroll
    for (int i = 0; i < NV; ++i) sv[w][i] = ld16(sp[w] + i * 8);
  }
#pragma unroll
  for (int w = 0; w < RW; ++w) {
    const int r = min(r0 + w, a.rows - 1);
    const bf16_t* xb = a.xbc + (size_t)r * a.conv_dim;
    const bf16_t* Bp = xb + a.d_inner + grp * N + q * NT;
    const bf16_t* Cp = Bp + a.ngroups * N;
#pragma unroll
    for (int i = 0; i < NV; ++i) { bv[w][i] = ld16(Bp + i * 8); cv[w][i] = ld16(Cp + i * 8); }
    x[w] = bf2f(xb[h * P + p]);
    dt[w] = __fadd_rn(bf2f(a.zx[(size_t)r * a.d_in_proj + a.d_inner + a.conv_dim + h]), bf2f(a.dt_bias[h]));
    zg[w] = bf2f(a.zx[(size_t)r * a.d_in_proj + h * P + p]);
  }
  const float A = -expf(bf2f(a.A_log[h]));
  const float Dh = bf2f(a.D[h]);
#pragma unroll
  for (int w = 0; w < RW; ++w) {
    if (r0 + w >= a.rows && w > 0) break;             // the clamped duplicate of an odd tail (wave-uniform)
    float dtv = dt[w];
    if (dtv <= 20.0f) dtv = log1pf(expf(dtv));
    const float dA = expf(__fmul_rn(dtv, A));
    float y = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const unsigned sw[4] = {sv[w][i].x, sv[w][i].y, sv[w][i].z, sv[w][i].w};
      const unsigned bw[4] = {bv[w][i].x, bv[w][i].y, bv[w][i].z, bv[w][i].w};
      const unsigned cw[4] = {cv[w][i].x, cv[w][i].y, cv[w][i].z, cv[w][i].w};
      unsigned ow[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float n0 = __fadd_rn(__fmul_rn(lo_f(sw[e]), dA), __fmul_rn(__fmul_rn(lo_f(bw[e]), dtv), x[w]));
        const float n1 = __fadd_rn(__fmul_rn(hi_f(sw[e]), dA), __fmul_rn(__fmul_rn(hi_f(bw[e]), dtv), x[w]));
        y = fmaf(n0, lo_f(cw[e]), y);
        y = fmaf(n1, hi_f(cw[e]), y);
        ow[e] = pack2(n0, n1);
      }
      *(u32x4*)(sp[w] + i * 8) = u32x4{ow[0], ow[1], ow[2], ow[3]};
    }
    // the 4 quarter-row partials of p sit in one quad
    y += dpp_mov<ZN_DPP_XOR1>(y);
    y += dpp_mov<ZN_DPP_XOR2>(y);
    if (q == 0) {
      const bf16_t yb = f2bf(__fadd_rn(y, __fmul_rn(x[w], Dh)));
      a.y[(size_t)(r0 + w) * a.d_inner + h * P + p] = yb;
      // first step of RMSNormGated (norm_before_gate=False), once per element here instead of once per consumer: the
      // fp32 product of the bf16 y and silu(z); the consumer (out_proj's prologue) adds the row statistics and the weight
      if (a.vg) a.vg[(size_t)(r0 + w) * a.d_inner + h * P + p] = __fmul_rn(bf2f(yb), __fmul_rn(zg[w], 1.0f / (1.0f + expf(-zg[w]))));
    }
  }
}

// RMSNormGated(norm_before_gate=False): v = y * silu(z); g = bf16(v * rsqrt(mean_group(v^2) + eps) * w).
// One workgroup per (row, group), 8 elements per thread per sweep (group size <= 8192).
__global__ __launch_bounds__(256) void mamba_gated_norm_kernel(MambaArgs a) {
  const int r = blockIdx.y, grp = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gs = a.d_inner / a.ngroups;
  constexpr int MAXV = 4;
  float v[MAXV][8];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int k = (it * 256 + tid) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[it][e] = 0.f;
    if (k < gs) {
      const u32x4 yv = ld16(a.y + (size_t)r * a.d_inner + grp * gs + k);
      const u32x4 zv = ld16(a.zx + (size_t)r * a.d_in_proj + grp * gs + k);
      const float yf[8] = {lo_f(yv.x), hi_f(yv.x), lo_f(yv.y), hi_f(yv.y), lo_f(yv.z), hi_f(yv.z), lo_f(yv.w), hi_f(yv.w)};
      const float zf[8] = {lo_f(zv.x), hi_f(zv.x), lo_f(zv.y), hi_f(zv.y), lo_f(zv.z), hi_f(zv.z), lo_f(zv.w), hi_f(zv.w)};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float sg = 1.0f / (1.0f + expf(-zf[e]));
        v[it][e] = __fmul_rn(yf[e], __fmul_rn(zf[e], sg));
        ss += v[it][e] * v[it][e];
      }
    }
  }
  __shared__ float red[4];
  ss = wave_sum(ss);
  if (lane == 0) red[wave] = ss;
  __syncthreads();
  const float rstd = 1.0f / sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)gs + a.eps);
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int k = (it * 256 + tid) * 8;
    if (k < gs) {
      const u32x4 wv = ld16(a.norm_w + grp * gs + k);
      const float wf[8] = {lo_f(wv.x), hi_f(wv.x), lo_f(wv.y), hi_f(wv.y), lo_f(wv.z), hi_f(wv.z), lo_f(wv.w), hi_f(wv.w)};
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = __fmul_rn(__fmul_rn(v[it][e], rstd), wf[e]);
      u32x4 ov;
      ov.x = pack2(o[0], o[1]); ov.y = pack2(o[2], o[3]); ov.z = pack2(o[4], o[5]); ov.w = pack2(o[6], o[7]);
      *(u32x4*)(a.g + (size_t)r * a.d_inner + grp * gs + k) = ov;
    }
  }
}


// ------------------------------------------------------------------------------------------------ Mamba2 over a sequence
// Prefill of S positions (the reference calls mamba_chunk_scan_combined there, _mamba_ssm.py:106-119 -> Mamba2.forward; its
// chunked SSD form computes the same recurrence): rows of every [.][S] operand are position-major inside an utterance
// row, index r * S + s.  The conv kernel repeats the single-step arithmetic statement for statement (window and outputs bit-identical
// to S single steps); the scan keeps its state in fp32 across the positions, as the reference's prefill does (below).

// causal conv + SiLU along the sequence: one thread per (row, channel) walks the S positions with the window in registers
__global__ __launch_bounds__(256) void mamba_conv_seq_kernel(MambaArgs a, int S) {
  const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (c >= a.conv_dim) return;
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
  bf16_t* sp = a.conv_state + ((size_t)r * a.conv_dim + c) * 4;
  u32x2_t st = *(const u32x2_t*)sp;
  const u32x2_t wv = *(const u32x2_t*)(a.conv_w + (size_t)c * 4);
  const float bias = bf2f(a.conv_b[c]);
  const bf16_t* xp = a.zx + (size_t)r * S * a.d_in_proj + a.d_inner + c;
  bf16_t* op = a.xbc + (size_t)r * S * a.conv_dim + c;
  constexpr int AH = 8;                                   // positions requested ahead
  bf16_t xq[AH];
#pragma unroll
  for (int i = 0; i < AH; ++i) xq[i] = xp[(size_t)min(i, S - 1) * a.d_in_proj];
  for (int s0 = 0; s0 < S; s0 += AH) {
    bf16_t xc[AH];
#pragma unroll
    for (int i = 0; i < AH; ++i) { xc[i] = xq[i]; xq[i] = xp[(size_t)min(s0 + AH + i, S - 1) * a.d_in_proj]; }
#pragma unroll
    for (int i = 0; i < AH; ++i) {
      if (s0 + i < S) {
        const bf16_t xn = xc[i];
        const float e0 = hi_f(st.x), e1 = lo_f(st.y), e2 = hi_f(st.y), e3 = bf2f(xn);
        u32x2_t ns;
        ns.x = (st.x >> 16) | (st.y << 16);
        ns.y = (st.y >> 16) | ((unsigned)xn << 16);
        st = ns;
        float acc = bias;
        acc = __fadd_rn(acc, __fmul_rn(lo_f(wv.x), e0));
        acc = __fadd_rn(acc, __fmul_rn(hi_f(wv.x), e1));
        acc = __fadd_rn(acc, __fmul_rn(lo_f(wv.y), e2));
        acc = __fadd_rn(acc, __fmul_rn(hi_f(wv.y), e3));
        op[(size_t)(s0 + i) * a.conv_dim] = f2bf(acc / (1.0f + expf(-acc)));
      }
    }
  }
  *(u32x2_t*)sp = st;
}

// selective scan: one workgroup per (row, head) keeps its 64 x N state tile in fp32 registers across the whole sequence and rounds it
// to the cache's bf16 ONCE, when it is stored: that is what the reference's S > 1 path does (Mamba2.forward ->
// mamba_chunk_scan_combined carries the state in fp32 and casts the final state into the cache, _mamba_ssm.py:106-119), unlike its
// S == 1 path (Mamba2.step -> selective_state_update on the bf16 cache: one rounding per token, mamba_ssm_kernel above).  A prefill is
// therefore NOT bit-identical to stepping the same tokens one by one (it is more accurate); position s's output uses the
// unrounded state either way.  The next position's B, C, x, dt are requested while the current one is computed.
template <int N>
__global__ __launch_bounds__(256) void mamba_scan_kernel(MambaArgs a, int S) {
  constexpr int P = 64, NT = N / 4, NV = NT / 8;
  const int h = blockIdx.x, r = blockIdx.y, tid = threadIdx.x;
  const int p = tid >> 2, q = tid & 3;
  const int grp = h / (a.nheads / a.ngroups);
  bf16_t* sp = a.ssm_state + (((size_t)r * a.nheads + h) * P + p) * N + q * NT;
  float sf[NT];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const u32x4 v = ld16(sp + i * 8);
    sf[8 * i + 0] = lo_f(v.x); sf[8 * i + 1] = hi_f(v.x); sf[8 * i + 2] = lo_f(v.y); sf[8 * i + 3] = hi_f(v.y);
    sf[8 * i + 4] = lo_f(v.z); sf[8 * i + 5] = hi_f(v.z); sf[8 * i + 6] = lo_f(v.w); sf[8 * i + 7] = hi_f(v.w);
  }
  const float A = -expf(bf2f(a.A_log[h]));
  const float Dh = bf2f(a.D[h]);
  const float dtb = bf2f(a.dt_bias[h]);
  u32x4 bn[NV], cn[NV];
  bf16_t xn, dn;
  auto request = [&](int s) {
    const size_t rs = (size_t)r * S + min(s, S - 1);
    const bf16_t* xb = a.xbc + rs * a.conv_dim;
    const bf16_t* Bp = xb + a.d_inner + grp * N + q * NT;
    const bf16_t* Cp = Bp + a.ngroups * N;
#pragma unroll
    for (int i = 0; i < NV; ++i) { bn[i] = ld16(Bp + i * 8); cn[i] = ld16(Cp + i * 8); }
    xn = xb[h * P + p];
    dn = a.zx[rs * a.d_in_proj + a.d_inner + a.conv_dim + h];
  };
  request(0);
  for (int s = 0; s < S; ++s) {
    u32x4 bv[NV], cv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { bv[i] = bn[i]; cv[i] = cn[i]; }
    const float x = bf2f(xn);
    float dtv = __fadd_rn(bf2f(dn), dtb);
    request(s + 1);
    if (dtv <= 20.0f) dtv = log1pf(expf(dtv));
    const float dA = expf(__fmul_rn(dtv, A));
    float y = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const unsigned bw[4] = {bv[i].x, bv[i].y, bv[i].z, bv[i].w};
      const unsigned cw[4] = {cv[i].x, cv[i].y, cv[i].z, cv[i].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float n0 = __fadd_rn(__fmul_rn(sf[8 * i + 2 * e], dA), __fmul_rn(__fmul_rn(lo_f(bw[e]), dtv), x));
        const float n1 = __fadd_rn(__fmul_rn(sf[8 * i + 2 * e + 1], dA), __fmul_rn(__fmul_rn(hi_f(bw[e]), dtv), x));
        y = fmaf(n0, lo_f(cw[e]), y);
        y = fmaf(n1, hi_f(cw[e]), y);
        sf[8 * i + 2 * e] = n0; sf[8 * i + 2 * e + 1] = n1;
      }
    }
    y += dpp_mov<ZN_DPP_XOR1>(y);
    y += dpp_mov<ZN_DPP_XOR2>(y);
    if (q == 0) a.y[((size_t)r * S + s) * a.d_inner + h * P + p] = f2bf(__fadd_rn(y, __fmul_rn(x, Dh)));
  }
#pragma unroll
  for (int i = 0; i < NV; ++i)
    *(u32x4*)(sp + i * 8) = u32x4{pack2(sf[8 * i], sf[8 * i + 1]), pack2(sf[8 * i + 2], sf[8 * i + 3]), pack2(sf[8 * i + 4], sf[8 * i + 5]), pack2(sf[8 * i + 6], sf[8 * i + 7])};
}
