// Batched prefill (S > 1): zonos/utilities/generation_utils.py:206-244 -> TorchZonosBackbone.forward with S positions.
// Row-wise ops are the decode path's arithmetic applied to M = R*S rows; the dense contractions run on the bf16 matrix
// cores (compute-bound here, unlike decode); attention reproduces the reference CPU flash kernel's per-row rounding
// (512-key blocks, running max, fexp_u20 / libm exp split governed by the query block's key span, bf16 P).
#pragma once
#include "zn_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 zn_bf16x8p;

// ------------------------------------------------------------------------------------------------ GEMM
// out[M][N] = A[M][K] . W[N][K]^T  (bf16 in, fp32 accumulate, bf16 out; EPI 1: out = bf16(resid + bf16(acc))).
// Workgroup tile 128 x 128, 4 waves as 2 x 2, each wave 64 x 64 = 4 x 4 tiles of v_mfma_f32_16x16x32_bf16.  Both
// operands are K-contiguous, so a fragment is one 16-B load per lane (row = lane & 15, k-group = lane >> 4) straight
// from L2: A panels are shared by the N/128 workgroups of a row, W panels by the M/128 of a column.
struct GemmArgs {
  const bf16_t* A; const bf16_t* W; bf16_t* out; const bf16_t* resid;
  int M, N, K, lda, ldo;   // lda = row stride of A (elements), ldo = row stride of out / resid
  const bf16_t* bias;      // optional nn.Linear bias [N], added in fp32 before the single bf16 rounding
};
template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 128 + wm * 64, n0 = blockIdx.x * 128 + wn * 64;
  const int fr = lane & 15, fg = lane >> 4;
  const bf16_t* ap[4];
  const bf16_t* wp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = min(m0 + i * 16 + fr, a.M - 1), n = min(n0 + i * 16 + fr, a.N - 1);
    ap[i] = a.A + (size_t)m * a.lda + 8 * fg;
    wp[i] = a.W + (size_t)n * a.K + 8 * fg;
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 fa[4], fw[4], na[4], nw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { fa[i] = ld16(ap[i]); fw[i] = ld16(wp[i]); }
  for (int k0 = 0; k0 < a.K; k0 += 32) {
    const bool more = k0 + 32 < a.K;
    if (more) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { na[i] = ld16(ap[i] + k0 + 32); nw[i] = ld16(wp[i] + k0 + 32); }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(zn_bf16x8p, fa[i]), __builtin_bit_cast(zn_bf16x8p, fw[j]), acc[i][j], 0, 0, 0);
    if (more) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { fa[i] = na[i]; fw[i] = nw[i]; }
    }
  }
  // C: col (n) = lane & 15, row (m) = 4 * (lane >> 4) + reg
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + j * 16 + fr;
      if (n >= a.N) continue;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int m = m0 + i * 16 + 4 * fg + reg;
        if (m >= a.M) continue;
        const size_t o = (size_t)m * a.ldo + n;
        float v = bfround(a.bias ? acc[i][j][reg] + bf2f(a.bias[n]) : acc[i][j][reg]);
        if constexpr (EPI == 1) v = bf2f(a.resid[o]) + v;
        a.out[o] = f2bf(v);
      }
    }
}

// Short prompts (17..64 rows: the L_c + 1 positions of a text-only prompt x 2 CFG rows): the 128 x 128 tiles above leave
// N/128 = 16..128 workgroups to stream a whole weight matrix through direct-fragment loads (0.3 TB/s: 10.7 ms per
// 25-position prefill).  This is the decode side's gemm16s_kernel with four 16-row blocks per weight fragment: a workgroup owns 64
// weight rows (EPI_SILU: 32 value + their 32 gate rows), walks its K slice in chunks of 256 with whole 512-byte row pieces staged
// in padded LDS next to the 64-row activation chunk (L2-resident), one 16-row weight tile per wave x four activation blocks
// on v_mfma_f32_16x16x32_bf16; K splits over gridDim.y workgroups whose fp32 partial tiles meet through write-through
// stores and an arrival ticket (the last workgroup adds the slices in order: deterministic), then the decode path's epilogues.
template <int EPI>
__global__ __launch_bounds__(256) void gemm64s_kernel(GemvArgs a) {
  constexpr int KC = 256, LDW = KC + 8, NT = 256, TN = 64, HALF = 32, RB = 4, MR = 16 * RB;
  __shared__ __attribute__((aligned(16))) bf16_t Ws[TN * LDW];
  __shared__ __attribute__((aligned(16))) bf16_t Xs[MR * LDW];
  __shared__ int s_last;
  float (*Ct)[MR + 1] = (float (*)[MR + 1])Ws;                 // the output tile reuses the weight staging area after the K loop
  static_assert(TN * (MR + 1) * sizeof(float) <= sizeof(Ws), "Ct fits the staging area");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int grp = blockIdx.x, ys = blockIdx.y;
  const int K = a.K, F = a.N >> 1;
  auto wrow = [&](int j) -> int {                              // LDS row j <-> weight row (clamped; masked in the epilogue)
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
  u32x4 wr[8], xr[8];
  const bf16_t *wp[8], *xp[8];
  int ll[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int idx = j * NT + tid, row = idx >> 5, c16 = idx & 31;          // 64 rows x 32 pieces of 16 B per chunk, both operands
    wp[j] = a.W + (size_t)wrow(row) * K + kbeg + c16 * 8;
    xp[j] = a.x + (size_t)(row < a.nrows ? row : a.nrows - 1) * K + kbeg + c16 * 8;
    ll[j] = row * LDW + c16 * 8;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) xr[j] = ld16(xp[j]);
#pragma unroll
  for (int j = 0; j < 8; ++j) wr[j] = ld_nt16(wp[j]);
  // (the residual operands of this thread's epilogue items travel with the first chunk, as in gemm16s_kernel: one dependent round trip
  // less behind the split-K combine)
  constexpr int NIT = (TN / 2) * MR / NT;
  unsigned resid_pre[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    resid_pre[it] = 0u;
    if constexpr (EPI == EPI_RESID) {
      const int item = it * NT + tid, m = item % MR, pj = item / MR, rowA = grp * TN + 2 * pj;
      if (m < a.nrows && rowA < a.N) {
        const size_t o = (size_t)m * a.N + rowA;
        resid_pre[it] = (rowA + 1 < a.N) ? *(const unsigned*)(a.resid + o) : (unsigned)a.resid[o];
      }
    }
  }
  f32x4 acc[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) { *(u32x4*)&Xs[ll[j]] = xr[j]; *(u32x4*)&Ws[ll[j]] = wr[j]; }
    __syncthreads();
    if (c + 1 < nchunks) {
#pragma unroll
      for (int j = 0; j < 8; ++j) xr[j] = ld16(xp[j] + (size_t)(c + 1) * KC);
#pragma unroll
      for (int j = 0; j < 8; ++j) wr[j] = ld_nt16(wp[j] + (size_t)(c + 1) * KC);
    }
    const bf16_t* wf = &Ws[(wave * 16 + n) * LDW + 8 * g];
#pragma unroll
    for (int st = 0; st < KC / 32; ++st) {
      const u32x4 bw = *(const u32x4*)(wf + 32 * st);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const u32x4 ax = *(const u32x4*)(&Xs[(rb * 16 + n) * LDW + 8 * g] + 32 * st);
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(zn_bf16x8p, ax), __builtin_bit_cast(zn_bf16x8p, bw), acc[rb], 0, 0, 0);
      }
    }
  }
  __syncthreads();                                             // every wave is done with Ws: it becomes Ct
  // D layout: col (LDS row wave*16 + n) = lane & 15, row (activation row of block rb) = 4*(lane>>4) + reg
  if (a.ksplit > 1) {
    const size_t ld = (size_t)gridDim.x * TN;
    float* pp = a.part + ((size_t)ys * MR) * ld + (size_t)grp * TN + wave * 16 + n;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) st_wt(pp + (size_t)(rb * 16 + 4 * g + reg) * ld, acc[rb][reg]);
    __builtin_amdgcn_s_waitcnt(0);                             // vmcnt(0): stores acknowledged
    __syncthreads();
    if (tid == 0) {
      const int t = __hip_atomic_fetch_add(a.tickets + grp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (t == a.ksplit - 1);
      if (s_last) __hip_atomic_store(a.tickets + grp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    // MR * TN / NT = 16 tile elements per thread; every slice of a batch of elements is requested before the first add: eight
    // elements at a time with up to 8 slices (two dependent read batches), four with up to 16 (the sums run over the same slices in
    // the same order either way)
    if (a.ksplit <= 8) {
      for (int i0 = 0; i0 < MR * TN / NT; i0 += 8) {
        float v[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int item = (i0 + i) * NT + tid, m = item / TN, col = item % TN;
          const float* q = a.part + (size_t)m * ld + (size_t)grp * TN + col;
#pragma unroll
          for (int y = 0; y < 8; ++y) v[i][y] = (y < a.ksplit) ? ld_wt(q + (size_t)y * MR * ld) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int item = (i0 + i) * NT + tid, m = item / TN, col = item % TN;
          float sum = 0.f;
#pragma unroll
          for (int y = 0; y < 8; ++y) sum += v[i][y];
          Ct[col][m] = sum;
        }
      }
    } else {
      for (int i0 = 0; i0 < MR * TN / NT; i0 += 4) {
        float v[4][16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int item = (i0 + i) * NT + tid, m = item / TN, col = item % TN;
          const float* q = a.part + (size_t)m * ld + (size_t)grp * TN + col;
#pragma unroll
          for (int y = 0; y < 16; ++y) v[i][y] = (y < a.ksplit) ? ld_wt(q + (size_t)y * MR * ld) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int item = (i0 + i) * NT + tid, m = item / TN, col = item % TN;
          float sum = 0.f;
#pragma unroll
          for (int y = 0; y < 16; ++y) sum += v[i][y];
          Ct[col][m] = sum;
        }
      }
    }
  } else {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) Ct[wave * 16 + n][rb * 16 + 4 * g + reg] = acc[rb][reg];
  }
  __syncthreads();
  // ---- epilogue: TN/2 row pairs x MR activation rows, 8 items per thread
#pragma unroll
  for (int it = 0; it < (TN / 2) * MR / NT; ++it) {
    const int item = it * NT + tid, m = item % MR, pj = item / MR;
    if (m >= a.nrows) continue;
    if constexpr (EPI == EPI_SILU) {
      const int u = grp * HALF + pj;
      if (u >= F) continue;
      gemv_epilogue<EPI>(a, m, u, F + u, true, u, Ct[pj][m], Ct[HALF + pj][m], 0u, 1.f, 0.f, 0);
    } else {
      const int rowA = grp * TN + 2 * pj, rowB = rowA + 1;
      if (rowA >= a.N) continue;
      const bool b_ok = rowB < a.N;
      const unsigned resid = resid_pre[it];
      gemv_epilogue<EPI>(a, m, rowA, rowB, b_ok, rowA >> 1, Ct[2 * pj][m], b_ok ? Ct[2 * pj + 1][m] : 0.f, resid, 1.f, 0.f, 0);
    }
  }
}

// LDS-staged variant for M >= 256 rows (long prompts).  The direct-fragment kernel above asks the vector-memory pipe for
// 16 different rows per 16-lane phase (the access pattern that bounded the decode-side gemm16_kernel); here both operand
// panels of a 128 x 128 x 64 step are fetched as whole 128-byte row pieces (8 lanes x 16 B per row piece), stored to LDS
// with a +8-element row pad (conflict-free 16-B fragment reads) and the fragments come from LDS; the next step's global
// requests are in flight during the MFMAs.  Same tile shape, accumulation type and epilogue as above.
#define ZN_PG_KC 64
template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16s_kernel(GemmArgs a) {
  constexpr int KC = ZN_PG_KC, LDW = KC + 8;
  __shared__ __attribute__((aligned(16))) bf16_t As[128 * LDW];
  __shared__ __attribute__((aligned(16))) bf16_t Bs[128 * LDW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int mb = blockIdx.y * 128, nb = blockIdx.x * 128;
  const int fr = lane & 15, fg = lane >> 4;
  // staging slots: 128 rows x 8 pieces of 16 B per panel = 1024 pieces over 256 threads
  const bf16_t* ap[4];
  const bf16_t* wp[4];
  int sl[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = j * 256 + tid, row = idx >> 3, c16 = idx & 7;
    ap[j] = a.A + (size_t)min(mb + row, a.M - 1) * a.lda + c16 * 8;          // clamped rows: masked in the epilogue
    wp[j] = a.W + (size_t)min(nb + row, a.N - 1) * a.K + c16 * 8;
    sl[j] = row * LDW + c16 * 8;
  }
  u32x4 ra[4], rw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { ra[j] = ld16(ap[j]); rw[j] = ld16(wp[j]); }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < a.K; k0 += KC) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) { *(u32x4*)&As[sl[j]] = ra[j]; *(u32x4*)&Bs[sl[j]] = rw[j]; }
    __syncthreads();
    if (k0 + KC < a.K) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { ra[j] = ld16(ap[j] + k0 + KC); rw[j] = ld16(wp[j] + k0 + KC); }
    }
#pragma unroll
    for (int ks = 0; ks < KC / 32; ++ks) {
      u32x4 fa[4], fw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = *(const u32x4*)&As[(wm * 64 + i * 16 + fr) * LDW + 32 * ks + 8 * fg];
        fw[i] = *(const u32x4*)&Bs[(wn * 64 + i * 16 + fr) * LDW + 32 * ks + 8 * fg];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(zn_bf16x8p, fa[i]), __builtin_bit_cast(zn_bf16x8p, fw[j]), acc[i][j], 0, 0, 0);
    }
  }
  const int m0 = mb + wm * 64, n0 = nb + wn * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + j * 16 + fr;
      if (n >= a.N) continue;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int m = m0 + i * 16 + 4 * fg + reg;
        if (m >= a.M) continue;
        const size_t o = (size_t)m * a.ldo + n;
        float v = bfround(a.bias ? acc[i][j][reg] + bf2f(a.bias[n]) : acc[i][j][reg]);
        if constexpr (EPI == 1) v = bf2f(a.resid[o]) + v;
        a.out[o] = f2bf(v);
      }
    }
}

// ------------------------------------------------------------------------------------------------ row-wise pieces
// q|k|v split, interleaved-pair RoPE (fp32, separate roundings) and KV append for all positions (_torch.py:399-411).
// qkv [R][S][nq + 2*nkv]: q is rotated in place, k (rotated) and v go to the cache at position base + s.
__global__ __launch_bounds__(256) void rope_kv_rows_kernel(bf16_t* qkv, bf16_t* kv, const float* rope, int S, int base, int max_len, int n_heads,
                                                           int n_heads_kv, int hd, int rope_positions) {
  const int s = blockIdx.x, r = blockIdx.y;
  const int nq = n_heads * hd, nk = n_heads_kv * hd, ld = nq + 2 * nk;
  const int pos = base + s;
  const int p = pos < rope_positions ? pos : rope_positions - 1;
  bf16_t* row = qkv + ((size_t)r * S + s) * ld;
  for (int n = threadIdx.x * 2; n < ld; n += 512) {
    const unsigned pr = *(const unsigned*)(row + n);
    const float x0 = lo_f(pr), x1 = hi_f(pr);
    if (n < nq + nk) {
      const int i = (n % hd) >> 1;
      const float cs = rope[((size_t)p * (hd >> 1) + i) * 2], sn = rope[((size_t)p * (hd >> 1) + i) * 2 + 1];
      float re, im;
      zn_rope_pair(x0, x1, cs, sn, re, im);
      const unsigned o = pack2(re, im);
      if (n < nq) *(unsigned*)(row + n) = o;
      else if (pos < max_len) *(unsigned*)(kv + (((size_t)r * max_len + pos) * 2 + 0) * nk + (n - nq)) = o;
    } else if (pos < max_len) {
      *(unsigned*)(kv + (((size_t)r * max_len + pos) * 2 + 1) * nk + (n - nq - nk)) = pr;
    }
  }
}
// General form for the hybrid stack's attention layers (mamba_ssm MHA, attn_cfg): qkv [R][S][ld] (biases already added) ->
// q_out [R][S][ldq] rotated (may alias qkv: every thread reads its two elements before it writes them), k (rotated) and v
// to the cache at position (lengths ? lengths[r] : base) + s.  mode 0: interleaved pairs (2i, 2i+1); mode 1: half-split
// pairs (i, i + hd/2) (rotary_emb_interleaved = False); mode 2: no rotary.  fp32 products and sums rounded separately.
__global__ __launch_bounds__(256) void rope_kv_any_kernel(const bf16_t* qkv, bf16_t* q_out, int ldq, bf16_t* kv, const float* rope, int S, int base,
                                                          const int* lengths, int max_len, int n_heads, int n_heads_kv, int hd, int rope_positions,
                                                          int mode) {
  const int s = blockIdx.x, r = blockIdx.y;
  const int nq = n_heads * hd, nk = n_heads_kv * hd, ld = nq + 2 * nk, hh = hd >> 1;
  const int pos = (lengths ? lengths[r] : base) + s;
  const int p = pos < rope_positions ? pos : rope_positions - 1;
  const bf16_t* row = qkv + ((size_t)r * S + s) * ld;
  bf16_t* qrow = q_out + ((size_t)r * S + s) * ldq;
  bf16_t* krow = kv + (((size_t)r * max_len + min(pos, max_len - 1)) * 2 + 0) * nk;
  bf16_t* vrow = krow + nk;
  const bool in_cache = pos < max_len;
  for (int pi = threadIdx.x; pi < ld / 2; pi += 256) {
    if (pi < (nq + nk) / 2) {
      const int g = pi / hh, i = pi % hh;
      const int e0 = (mode == 1) ? g * hd + i : g * hd + 2 * i, e1 = (mode == 1) ? e0 + hh : e0 + 1;
      const float x0 = bf2f(row[e0]), x1 = bf2f(row[e1]);
      float o0 = x0, o1 = x1;
      if (mode != 2) {
        const float cs = rope[((size_t)p * hh + i) * 2], sn = rope[((size_t)p * hh + i) * 2 + 1];
        zn_rope_pair(x0, x1, cs, sn, o0, o1);
      }
      if (e0 < nq) { qrow[e0] = f2bf(o0); qrow[e1] = f2bf(o1); }
      else if (in_cache) { krow[e0 - nq] = f2bf(o0); krow[e1 - nq] = f2bf(o1); }
    } else if (in_cache) {
      const int n = 2 * pi;
      *(unsigned*)(vrow + (n - nq - nk)) = *(const unsigned*)(row + n);
    }
  }
}
// m = y * silu(gate) with bf16 roundings after silu and mul (_torch.py:473-474); u [M][2F] -> m [M][F]
__global__ __launch_bounds__(256) void silu_mul_rows_kernel(const bf16_t* u, bf16_t* m, int F) {
  const size_t row = blockIdx.x;
  for (int i = threadIdx.x * 2; i < F; i += 512) {
    const unsigned y = *(const unsigned*)(u + row * 2 * F + i), g = *(const unsigned*)(u + row * 2 * F + F + i);
    const float g0 = lo_f(g), g1 = hi_f(g);
    const float s0 = bfround(g0 / (1.0f + expf(-g0))), s1 = bfround(g1 / (1.0f + expf(-g1)));
    *(unsigned*)(m + row * F + i) = pack2(lo_f(y) * s0, hi_f(y) * s1);
  }
}
// row r of dst <- the last of the S positions of row r of src (rows of row_bytes bytes, a multiple of 16)
__global__ void gather_last_bytes_kernel(const void* src, void* dst, int S, int row_bytes) {
  const int r = blockIdx.x;
  const u32x4* sp = (const u32x4*)((const char*)src + ((size_t)r * S + (S - 1)) * row_bytes);
  u32x4* dp = (u32x4*)((char*)dst + (size_t)r * row_bytes);
  for (int i = threadIdx.x; i < row_bytes / 16; i += blockDim.x) dp[i] = sp[i];
}
__global__ void fill_int_kernel(int* p, int n, int v) { if ((int)threadIdx.x < n) p[threadIdx.x] = v; }
__global__ void gather_last_kernel(const bf16_t* xP, bf16_t* x, int S, int d) {
  const int r = blockIdx.x;
  for (int k = threadIdx.x * 8; k < d; k += blockDim.x * 8) *(u32x4*)(x + (size_t)r * d + k) = *(const u32x4*)(xP + ((size_t)r * S + S - 1) * d + k);
}

// ------------------------------------------------------------------------------------------------ causal attention
// One workgroup = 64 (position, head) rows of one (sequence row r, kv head): 64/G consecutive positions x G heads; wave w
// owns rows 16w..16w+15, lane = (row, key quarter): the 4 lanes of a row split every staged 64-key chunk.  K and V chunks
// are staged in LDS (padded rows) and reused by all 64 rows.  Each 512-key block takes two passes over its chunks:
// pass 1 = row max, pass 2 = recompute the scores, e/P with the reference's rounding, P.V — so no score buffer.
struct PrefillAttnArgs {
  const bf16_t* q; int ldq;     // rows (r*S + s), head h at column h*HD
  const bf16_t* kv;             // [R][max_len][2][Hkv][HD]
  bf16_t* out; int ldo;
  int S, base, max_len, n_heads, n_heads_kv, qsplit;
  float scale;
};
template <int HD, int G>
__global__ __launch_bounds__(256) void attn_prefill_kernel(PrefillAttnArgs a) {
  constexpr int TQ = 64 / G;              // positions per workgroup
  constexpr int KP = HD + 8;              // padded LDS row (bf16 elements): 16 B of padding breaks the 256-B bank period
  constexpr int NQ = HD / 8;              // 16-B pieces per head row
  __shared__ __attribute__((aligned(16))) bf16_t s_k[64 * KP];
  __shared__ __attribute__((aligned(16))) bf16_t s_v[64 * KP];
  const int s0 = blockIdx.x * TQ, kvh = blockIdx.y, r = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rw = lane & 15, kq = lane >> 4;
  const int row = wave * 16 + rw;                     // 0..63
  const int s = s0 + row / G, g = row % G;
  const bool row_ok = s < a.S;
  const int L_row = row_ok ? a.base + s + 1 : 0;      // keys this row may see
  const int Lmax = a.base + min(s0 + TQ, a.S);
  const int sq = row_ok ? s : s0;
  const int E = a.base + min((sq / a.qsplit) * a.qsplit + a.qsplit, a.S);   // keys spanned by this row's query block in the reference
  u32x4 qv[NQ];
  {
    const bf16_t* qp = a.q + ((size_t)r * a.S + (row_ok ? s : a.S - 1)) * a.ldq + (size_t)(kvh * G + g) * HD;
#pragma unroll
    for (int i = 0; i < NQ; ++i) qv[i] = ld16(qp + 8 * i);
  }
  float acc[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) acc[d] = 0.f;
  float lsum = 0.f, m_run = -INFINITY;
  const size_t kvrow = (size_t)2 * a.n_heads_kv * HD;
  const bf16_t* kg = a.kv + (size_t)r * a.max_len * kvrow + (size_t)kvh * HD;
  const bf16_t* vg = kg + (size_t)a.n_heads_kv * HD;

  for (int t0 = 0; t0 < Lmax; t0 += 512) {
    const int nch = min(8, (Lmax - t0 + 63) / 64);
    const int nblk = min(512, E - t0), nvec = nblk & ~15;
    // ---------------- pass 1: block max of this row
    float mx = -INFINITY;
    for (int c = 0; c < nch; ++c) {
      __syncthreads();
      for (int i = tid; i < 64 * NQ; i += 256) {
        const int key = i / NQ, pc = i % NQ, t = t0 + c * 64 + key;
        *(u32x4*)(s_k + key * KP + 8 * pc) = (t < Lmax) ? ld16(kg + (size_t)t * kvrow + 8 * pc) : u32x4{0, 0, 0, 0};
      }
      __syncthreads();
#pragma unroll 4
      for (int kk = 0; kk < 16; ++kk) {
        const int key = kq * 16 + kk, t = t0 + c * 64 + key;
        float dsum = 0.f;
#pragma unroll
        for (int i = 0; i < NQ; ++i) dsum = dot8(*(const u32x4*)(s_k + key * KP + 8 * i), qv[i], dsum);
        const float sc = __fmul_rn(dsum, a.scale);
        if (t < L_row) mx = fmaxf(mx, sc);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mnew = fmaxf(m_run, mx);
    const float f = (t0 == 0) ? 0.f : expf(m_run - mnew);
    m_run = mnew;
    lsum = __fmul_rn(lsum, f);
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = __fmul_rn(acc[d], f);
    // ---------------- pass 2: e / P / P.V
    for (int c = 0; c < nch; ++c) {
      __syncthreads();
      for (int i = tid; i < 64 * NQ; i += 256) {
        const int key = i / NQ, pc = i % NQ, t = t0 + c * 64 + key;
        const bool ok = t < Lmax;
        *(u32x4*)(s_k + key * KP + 8 * pc) = ok ? ld16(kg + (size_t)t * kvrow + 8 * pc) : u32x4{0, 0, 0, 0};
        *(u32x4*)(s_v + key * KP + 8 * pc) = ok ? ld16(vg + (size_t)t * kvrow + 8 * pc) : u32x4{0, 0, 0, 0};
      }
      __syncthreads();
      for (int kk = 0; kk < 16; ++kk) {
        const int key = kq * 16 + kk, idx = c * 64 + key, t = t0 + idx;
        if (__all(t >= L_row)) continue;               // wave-uniform skip of fully masked keys
        float dsum = 0.f;
#pragma unroll
        for (int i = 0; i < NQ; ++i) dsum = dot8(*(const u32x4*)(s_k + key * KP + 8 * i), qv[i], dsum);
        float e = 0.f;
        if (t < L_row) { const float x = __fsub_rn(__fmul_rn(dsum, a.scale), mnew); e = (idx < nvec) ? zn_fexp_u20(x) : expf(x); }
        lsum += e;
        const float p = bfround(e);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          const u32x4 v = *(const u32x4*)(s_v + key * KP + 8 * i);
          acc[8 * i + 0] = fmaf(p, lo_f(v.x), acc[8 * i + 0]); acc[8 * i + 1] = fmaf(p, hi_f(v.x), acc[8 * i + 1]);
          acc[8 * i + 2] = fmaf(p, lo_f(v.y), acc[8 * i + 2]); acc[8 * i + 3] = fmaf(p, hi_f(v.y), acc[8 * i + 3]);
          acc[8 * i + 4] = fmaf(p, lo_f(v.z), acc[8 * i + 4]); acc[8 * i + 5] = fmaf(p, hi_f(v.z), acc[8 * i + 5]);
          acc[8 * i + 6] = fmaf(p, lo_f(v.w), acc[8 * i + 6]); acc[8 * i + 7] = fmaf(p, hi_f(v.w), acc[8 * i + 7]);
        }
      }
    }
  }
  // combine the 4 key quarters of each row (lanes l, l+16, l+32, l+48), normalise, store
  lsum += __shfl_xor(lsum, 16);
  lsum += __shfl_xor(lsum, 32);
  const float rl = 1.0f / lsum;
  bf16_t* op = a.out + ((size_t)r * a.S + (row_ok ? s : 0)) * a.ldo + (size_t)(kvh * G + g) * HD;
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = acc[8 * i + e];
      x += __shfl_xor(x, 16);
      x += __shfl_xor(x, 32);
      v[e] = __fmul_rn(x, rl);
    }
    // piece i is stored by key-quarter lane (i & 3): spreads the 16-B stores over the 4 lanes of the row
    if (row_ok && kq == (i & 3)) {
      u32x4 o;
      o.x = pack2(v[0], v[1]); o.y = pack2(v[2], v[3]); o.z = pack2(v[4], v[5]); o.w = pack2(v[6], v[7]);
      *(u32x4*)(op + 8 * i) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------ attention on the matrix cores
// Same row semantics as attn_prefill_kernel (the reference CPU flash kernel: 512-key blocks, running max, fexp_u20 on the
// first 16*floor(n/16) keys of the block's span / libm exp on the tail, row sum of the unrounded e, P = bf16(e), rescale
// by exp(m_old - m_new)), with both contractions on v_mfma_f32_16x16x32_bf16.  A workgroup owns 64 query rows (64/G
// positions x G heads of one kv head), 16 per wave, and walks the keys in 64-key chunks staged row-major in LDS by
// coalesced 16-byte pieces.  Per 512-key block: pass 1 computes S = Q K^T per chunk and keeps only the row maxima;
// pass 2 recomputes S, forms e and P, writes the wave's P tile to LDS in A-operand order and accumulates P.V.
// Neither contraction cares about the order of its k index nor P.V about the order of its output columns, so both are
// permuted to fit the data as it lies: k-slot 4*fn + ct of P holds key ct*16 + fn (a lane's four scores of one row are one
// 8-byte LDS store), and column fn of output tile dt is dim 8*fn + dt, which lets a lane fetch V as eight 16-byte row
// pieces (keys of its k-slots x dims 8*fn..8*fn+7) and transpose them in registers (v_perm_b32) into the eight B operands.
// The accumulators (16 rows x 128 dims per wave) live in registers in the D layout: col = lane & 15, row = 4*(lane>>4)+reg.
template <int G>
__global__ __launch_bounds__(256) void attn_prefill_mfma_kernel(PrefillAttnArgs a) {
  constexpr int HD = 128, TQ = 64 / G, KP = HD + 8, PP = 64 + 8, OP = HD + 4;
  constexpr int K_BYTES = 64 * KP * 2, P_BYTES = 4 * 16 * PP * 2, O_BYTES = 4 * 16 * OP * 4;
  constexpr int SMEM = (2 * K_BYTES + P_BYTES) > O_BYTES ? (2 * K_BYTES + P_BYTES) : O_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  bf16_t* s_k = (bf16_t*)smem;                          // [key][dim]
  bf16_t* s_v = (bf16_t*)(smem + K_BYTES);              // [key][dim]
  bf16_t* s_p = (bf16_t*)(smem + 2 * K_BYTES);          // per wave [row][k-slot]
  float* s_o = (float*)smem;                            // epilogue staging [wave][row][dim] (after the key loop)
  const int s0 = ((int)gridDim.x - 1 - (int)blockIdx.x) * TQ, kvh = blockIdx.y, r = blockIdx.z;     // longest rows first
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fn = lane & 15, fg = lane >> 4;
  u32x4 qa[4];                                          // A operand: row = fn, k = 32*st + 8*fg + j
  {
    const int arow = wave * 16 + fn;
    const int as = min(s0 + arow / G, a.S - 1), ag = arow % G;
    const bf16_t* qp = a.q + ((size_t)r * a.S + as) * a.ldq + (size_t)(kvh * G + ag) * HD + 8 * fg;
#pragma unroll
    for (int st = 0; st < 4; ++st) qa[st] = ld16(qp + 32 * st);
  }
  int Lrow[4], Erow[4];             // keys row 4*fg + reg of this wave may see / keys spanned by its query block in the reference
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = wave * 16 + 4 * fg + reg, s = s0 + row / G;
    const bool ok = s < a.S;
    Lrow[reg] = ok ? a.base + s + 1 : 0;
    const int sq = ok ? s : s0;
    Erow[reg] = a.base + min((sq / a.qsplit) * a.qsplit + a.qsplit, a.S);
  }
  const int Lmax = a.base + min(s0 + TQ, a.S);
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float lsum[4] = {0.f, 0.f, 0.f, 0.f}, m_run[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  const size_t kvrow = (size_t)2 * a.n_heads_kv * HD;
  const bf16_t* kg = a.kv + (size_t)r * a.max_len * kvrow + (size_t)kvh * HD;
  const bf16_t* vg = kg + (size_t)a.n_heads_kv * HD;
  bf16_t* pw = s_p + wave * 16 * PP;

  // scaled scores of one staged chunk: sc[ct][reg] = row 4*fg+reg, key ct*16 + fn
  auto scores = [&](f32x4 (&sc)[4]) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const u32x4 kb = *(const u32x4*)(s_k + (ct * 16 + fn) * KP + 32 * st + 8 * fg);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(zn_bf16x8p, qa[st]), __builtin_bit_cast(zn_bf16x8p, kb), c, 0, 0, 0);
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) sc[ct][reg] = __fmul_rn(c[reg], a.scale);
    }
  };
  // rows at or past Lmax are never visible to this workgroup; their (clamped) loads are zeroed so that 0 * V stays 0
  auto stage = [&](int tb, bool with_v) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = tid + 256 * j, key = i >> 4, pc = i & 15, t = tb + key;
      const size_t off = (size_t)min(t, Lmax - 1) * kvrow + 8 * pc;
      u32x4 kx = ld16(kg + off);
      if (t >= Lmax) kx = u32x4{0, 0, 0, 0};
      *(u32x4*)(s_k + key * KP + 8 * pc) = kx;
      if (with_v) {
        u32x4 vx = ld16(vg + off);
        if (t >= Lmax) vx = u32x4{0, 0, 0, 0};
        *(u32x4*)(s_v + key * KP + 8 * pc) = vx;
      }
    }
  };

  for (int t0 = 0; t0 < Lmax; t0 += 512) {
    const int nch = min(8, (Lmax - t0 + 63) / 64);
    // ---------------- pass 1: block maxima of the rows
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int c = 0; c < nch; ++c) {
      __syncthreads();
      stage(t0 + c * 64, false);
      __syncthreads();
      f32x4 sc[4];
      scores(sc);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int t = t0 + c * 64 + ct * 16 + fn;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) mx[reg] = fmaxf(mx[reg], t < Lrow[reg] ? sc[ct][reg] : -INFINITY);
      }
    }
    float mnew[4];
    int nvec[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const float bm = group_max<16>(mx[reg]);            // over the 16 key lanes of the row group
      mnew[reg] = fmaxf(m_run[reg], bm);
      const float f = (t0 == 0) ? 0.f : expf(m_run[reg] - mnew[reg]);
      m_run[reg] = mnew[reg];
      lsum[reg] = __fmul_rn(lsum[reg], f);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i][reg] = __fmul_rn(acc[i][reg], f);
      nvec[reg] = min(512, Erow[reg] - t0) & ~15;
    }
    // ---------------- pass 2: e / P / P.V
    for (int c = 0; c < nch; ++c) {
      __syncthreads();
      stage(t0 + c * 64, true);
      __syncthreads();
      f32x4 sc[4];
      scores(sc);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        float e[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const int idx = c * 64 + ct * 16 + fn, t = t0 + idx;
          const float x = __fsub_rn(sc[ct][reg], mnew[reg]);
          const float ev = (idx < nvec[reg]) ? zn_fexp_u20(x) : expf(x);
          e[ct] = (t < Lrow[reg]) ? ev : 0.f;
          lsum[reg] += e[ct];
        }
        uint2 pk; pk.x = pack2(e[0], e[1]); pk.y = pack2(e[2], e[3]);
        *(uint2*)(pw + (4 * fg + reg) * PP + 4 * fn) = pk;             // k-slots 4*fn .. 4*fn+3
      }
      // the P tile is private to the wave: LDS ops of one wave complete in order, so a wave-level fence is enough
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const u32x4 pa = *(const u32x4*)(pw + fn * PP + 32 * ks + 8 * fg);       // k-slots 32*ks + 8*fg + j
        u32x4 vr[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {                                             // slot sigma holds key (sigma&3)*16 + (sigma>>2)
          const int key = (j & 3) * 16 + 8 * ks + 2 * fg + (j >> 2);
          vr[j] = *(const u32x4*)(s_v + key * KP + 8 * fn);
        }
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
          const unsigned sel = (dt & 1) ? 0x07060302u : 0x05040100u;
          u32x4 vb;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned lo = (dt >> 1) == 0 ? vr[2 * q].x : (dt >> 1) == 1 ? vr[2 * q].y : (dt >> 1) == 2 ? vr[2 * q].z : vr[2 * q].w;
            const unsigned hi = (dt >> 1) == 0 ? vr[2 * q + 1].x : (dt >> 1) == 1 ? vr[2 * q + 1].y : (dt >> 1) == 2 ? vr[2 * q + 1].z : vr[2 * q + 1].w;
            const unsigned pr = __builtin_amdgcn_perm(hi, lo, sel);
            if (q == 0) vb.x = pr; else if (q == 1) vb.y = pr; else if (q == 2) vb.z = pr; else vb.w = pr;
          }
          acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(zn_bf16x8p, pa), __builtin_bit_cast(zn_bf16x8p, vb), acc[dt], 0, 0, 0);
        }
      }
    }
  }
  // row sums are split over the 16 key lanes of each row group; normalise and stage the tile for coalesced stores
  __syncthreads();
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const float l = group_sum<16>(lsum[reg]);
    const float rl = 1.0f / l;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) s_o[(wave * 16 + 4 * fg + reg) * OP + 8 * fn + dt] = __fmul_rn(acc[dt][reg], rl);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {                          // 16 rows x 16 pieces of 8 dims per wave
    const int i = lane + 64 * j, row = i >> 4, pc = i & 15, grow = wave * 16 + row, s = s0 + grow / G, g = grow % G;
    if (s >= a.S) continue;
    const float* o = s_o + grow * OP + 8 * pc;
    u32x4 ov;
    ov.x = pack2(o[0], o[1]); ov.y = pack2(o[2], o[3]); ov.z = pack2(o[4], o[5]); ov.w = pack2(o[6], o[7]);
    *(u32x4*)(a.out + ((size_t)r * a.S + s) * a.ldo + (size_t)(kvh * G + g) * HD + 8 * pc) = ov;
  }
}
