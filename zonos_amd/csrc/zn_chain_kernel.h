// Persistent post-attention chain of one transformer block at batch 1 (R = 2 rows, CFG pair), one launch per layer:
//
//   op 0  y1 = out_proj(a)                                   (_torch.py:419)
//   op 1  x  = x + out_proj(y1)                              (_torch.py:420, 326)
//   op 2  m  = y * silu(gate), (y, gate) = fc1(LayerNorm2(x))  (_torch.py:327, 473-474)
//   op 3  x  = x + fc2(m)
//   op 4  q | k | v = in_proj(LayerNorm1'(x)) of the NEXT block, RoPE, KV append   (_torch.py:399-408, 105-106)
//
// As five launches these cost ~3.3 us of fixed time each (boundary + ramp until the first weights arrive) on top of
// their streaming time; here one workgroup per CU stays resident for the whole chain: its 4 compute waves walk a STATIC
// list of weight tiles (a tile = two weight rows x 512*NCH columns = one gemv_kernel work unit, same arithmetic order,
// so the results are bit-identical to the launches path) with three tiles requested ahead in registers — requests run
// across the op boundaries, so the HBM stream no longer drains at every dependency —, and a fifth, communication wave
// (it issues no weight loads, so its own waits cover only the hand-off traffic) finishes each op: epilogue of the
// workgroup's rows, write-through (sc1) stores, drain, one arrival on a sharded agent-scope counter, a bounded relaxed
// poll of the shards (sc1), an sc1 gather of the op's whole output vector (8-32 KB), LayerNorm where the next op wants
// it, and the vector into LDS for the compute waves (cdna_hip_programming.md Guideline 16 R1; MI355X_MICROARCH.md
// visibility table row 1).  Every spin is bounded; a timeout sets a word the host turns into an error.
#pragma once
#include <utility>
#include "zn_decode_kernels.h"

#define ZN_CH_NSHARD 8
#define ZN_CH_SSTRIDE 16                                   // u32 between shards: one 64-B line each
#define ZN_CH_CTR_WORDS (ZN_CH_NSHARD * ZN_CH_SSTRIDE)     // words per hand-off counter block
#define ZN_CH_HANDOFFS 4
#define ZN_CH_CWAVES 4                                     // compute waves (a multiple of 4); the next wave = communication wave
#define ZN_CH_NBUF 3                                       // weight tiles requested ahead per compute wave (register buffers)
#define ZN_CH_THREADS ((ZN_CH_CWAVES + 1) * 64)
#define ZN_CH_TIMEOUT_TICKS 2000000ull                     // 20 ms of s_memrealtime (100 MHz)

struct ChainArgs {
  const bf16_t *W_out, *W_fc1, *W_fc2, *W_in;              // W_in = next block's in_proj (NULL: the chain ends after fc2)
  const bf16_t *ln2_w, *ln2_b, *lnn_w, *lnn_b;             // this block's norm2, next block's norm
  float eps;
  int F, nqkv;                                             // d_ff (= 4 d_model), rows of the next in_proj
  const bf16_t* a;                                         // [2][d] attention output
  // No address is written twice inside a launch, nor read before it is written there: a line fetched earlier in the launch
  // (even by an sc1 load) can survive in the XCD's L2 and be hit by a later gather although other XCDs have rewritten it
  // (observed: a stale x element once per ~4000 launches with x updated in place).
  const bf16_t* xin;                                       // [2][d] residual stream entering the block (read only)
  bf16_t* x1;                                              // [2][d] after the attention half (op 1)
  bf16_t* xout;                                            // [2][d] residual stream leaving the block (op 3); != xin
  bf16_t* y1;                                              // [2][d]
  bf16_t* m;                                               // [2][F]
  bf16_t* q_out; bf16_t* kv; const float* rope; const int* lengths;   // op 4 epilogue (next block's cache)
  int max_len, hd, n_heads, n_heads_kv, rope_positions;
  unsigned* ctr;                                           // [ZN_CH_HANDOFFS][ZN_CH_CTR_WORDS], zero at launch
  int* tmo;                                                // sticky timeout word (GenState.pad[0])
  unsigned long long* stamps;                              // optional [32] (diagnostic builds of the timeline: workgroup 0's communication wave)
};

template <int I, int N, class Fn> ZN_DEVINL void zn_static_for(Fn&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); zn_static_for<I + 1, N>(f); }
}
ZN_DEVINL u32x4 ld_sc1_16(__amdgpu_buffer_rsrc_t rs, int byte_off) { return __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16); }   // aux 16 = sc1
ZN_DEVINL __amdgpu_buffer_rsrc_t zn_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000); }
ZN_DEVINL void st_sc1_u32(void* p, unsigned v) { __hip_atomic_store((unsigned*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
ZN_DEVINL unsigned ld_sc1_u32(const void* p) { return __hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
ZN_DEVINL unsigned dpp_movu(unsigned v, int) { return v; }
template <int CTRL> ZN_DEVINL unsigned dpp_u(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }

// The communication wave waits until `expect` workgroups have arrived at the hand-off (lanes 0..7 read one shard each,
// lane 8 the sticky timeout word): relaxed sc1 polls, one load in flight, bounded.
ZN_DEVINL bool chain_wait(const unsigned* ctr, unsigned expect, int* tmo, int lane) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    unsigned v = 0;
    if (lane < ZN_CH_NSHARD) v = ld_sc1_u32(ctr + lane * ZN_CH_SSTRIDE);
    else if (lane == ZN_CH_NSHARD) v = ld_sc1_u32(tmo);
    unsigned s = lane < ZN_CH_NSHARD ? v : 0u;
    s += dpp_u<ZN_DPP_XOR1>(s); s += dpp_u<ZN_DPP_XOR2>(s); s += dpp_u<ZN_DPP_HALF_MIRROR>(s);     // lanes 0..7
    const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)s, 0);
    const unsigned dead = (unsigned)__builtin_amdgcn_readlane((int)v, ZN_CH_NSHARD);
    if (total >= expect) return true;
    if (dead) return false;
    if (__builtin_amdgcn_s_memrealtime() - t0 > ZN_CH_TIMEOUT_TICKS) { if (lane == 0) atomicAdd(tmo, 1); return false; }
  }
}

// nn.LayerNorm on two rows held as gemv_kernel holds them (lane owns elements (c*64 + lane)*8 .. +8 of each row), through the
// helpers gemv_kernel's PRO_LN prologue uses (explicit roundings): identical statistics and outputs.
template <int NCH>
ZN_DEVINL void chain_layernorm(u32x4 (&xr)[NCH][2], const u32x4 (&lng)[NCH], const u32x4 (&lnb)[NCH], float eps) {
  constexpr int R = 2;
  const float invK = 1.0f / (float)(NCH * 512);
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
    for (int c = 0; c < NCH; ++c) ss[r] = ln_sq8(xr[c][r], mean[r], ss[r]);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) rstd[r] = ln_rstd(wave_sum(ss[r]), invK, eps);
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int r = 0; r < R; ++r) xr[c][r] = ln_norm8(xr[c][r], mean[r], rstd[r], lng[c], lnb[c]);
}

// T_* = tiles per compute wave per op (upper bounds; a wave skips the tiles its workgroup does not have).  d_model =
// 512 * NCH, d_ff = 4 * d_model; every op's units divide evenly over the grid (host-checked).
template <int NCH, int T_OUT, int T_FC1, int T_FC2, int T_IN>
__global__ __launch_bounds__(ZN_CH_THREADS) void chain_kernel(ChainArgs a) {
  constexpr int R = 2, D = NCH * 512, CW = ZN_CH_CWAVES;
  constexpr int S1 = T_OUT, S2 = 2 * T_OUT, S3 = S2 + T_FC1, S4 = S3 + T_FC2, NS = S4 + T_IN;     // slot ranges per op
  constexpr int NOPS = T_IN > 0 ? 5 : 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = blockIdx.x, G = gridDim.x;
  const int F = a.F;
  // units (weight-row pairs) per workgroup and op
  const int ppw_out = (D / 2) / G, ppw_fc1 = F / G, ppw_fc2 = (D / 2) / G, ppw_in = T_IN > 0 ? (a.nqkv / 2) / G : 0;
  __shared__ __attribute__((aligned(16))) bf16_t s_act[R * 4 * D];          // the current op's input vector (rows of up to 4 d)
  __shared__ float s_res[64][2][R];                                         // per-unit results (fc2: [unit * 4 + quarter])

  auto op_of = [](int s) constexpr { return s < S1 ? 0 : s < S2 ? 1 : s < S3 ? 2 : s < S4 ? 3 : 4; };
  auto first_of = [](int op) constexpr { return op == 0 ? 0 : op == 1 ? S1 : op == 2 ? S2 : op == 3 ? S3 : S4; };

  if (wave < CW) {
    // ------------------------------------------------------------------------------------ compute waves
    struct WT { u32x4 a[NCH], b[NCH]; };
    WT buf0, buf1, buf2;
    static_assert(ZN_CH_NBUF == 3 && CW % 4 == 0, "three rotating tile buffers; fc2 splits K over groups of four waves");
    // tile of slot s for this wave: exists?, weight pointer of rows A and B (lane's first chunk), result index
    auto tile = [&](int s, bool& ok, const bf16_t*& pa, const bf16_t*& pb, int& ridx) {
      const int op = op_of(s), t = s - first_of(op);
      if (op == 3) {
        const int qt = wave & 3, j = (wave >> 2) + (CW / 4) * t;
        ok = j < ppw_fc2;
        const int u = c * ppw_fc2 + (ok ? j : 0);
        pa = a.W_fc2 + (size_t)(2 * u) * (4 * D) + qt * D + lane * 8;
        pb = pa + 4 * D;
        ridx = j * 4 + qt;
        return;
      }
      const int j = wave + CW * t;
      const int ppw = op <= 1 ? ppw_out : op == 2 ? ppw_fc1 : ppw_in;
      ok = j < ppw;
      const int u = c * ppw + (ok ? j : 0);
      ridx = j;
      if (op <= 1) { pa = a.W_out + (size_t)(2 * u) * D + lane * 8; pb = pa + D; }
      else if (op == 2) { pa = a.W_fc1 + (size_t)u * D + lane * 8; pb = pa + (size_t)F * D; }
      else { pa = a.W_in + (size_t)(2 * u) * D + lane * 8; pb = pa + D; }
    };
    auto load = [&](int s, WT& w) {
      bool ok; const bf16_t *pa, *pb; int ridx;
      tile(s, ok, pa, pb, ridx);
      if (ok) {                                           // wave-uniform
#pragma unroll
        for (int c2 = 0; c2 < NCH; ++c2) { w.a[c2] = ld_nt16(pa + c2 * 512); w.b[c2] = ld_nt16(pb + c2 * 512); }
      }
    };
    u32x4 xr[NCH][R];
    auto process = [&](int s, const WT& w) {
      bool ok; const bf16_t *pa, *pb; int ridx;
      tile(s, ok, pa, pb, ridx);
      if (!ok) return;
      float accA[R] = {0.f, 0.f}, accB[R] = {0.f, 0.f};
#pragma unroll
      for (int c2 = 0; c2 < NCH; ++c2) {
#pragma unroll
        for (int r = 0; r < R; ++r) { accA[r] = dot8(w.a[c2], xr[c2][r], accA[r]); accB[r] = dot8(w.b[c2], xr[c2][r], accB[r]); }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) { accA[r] = wave_sum(accA[r]); accB[r] = wave_sum(accB[r]); }
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) { s_res[ridx][0][r] = accA[r]; s_res[ridx][1][r] = accB[r]; }
      }
    };
    load(0, buf0);
    if constexpr (NS > 1) load(1, buf1);
    if constexpr (NS > 2) load(2, buf2);
    zn_static_for<0, NS>([&](auto SC) {
      constexpr int s = decltype(SC)::value;
      constexpr int op = op_of(s);
      if constexpr (s == first_of(op)) {
        if constexpr (op > 0) __syncthreads();            // A(op-1): this workgroup's results of the previous op are in LDS
        __syncthreads();                                  // B(op): the op's input vector is in LDS
        const int kofs = (op == 3) ? (wave & 3) * D : 0, rstride = (op == 3) ? 4 * D : D;
#pragma unroll
        for (int c2 = 0; c2 < NCH; ++c2)
#pragma unroll
          for (int r = 0; r < R; ++r) xr[c2][r] = *(const u32x4*)&s_act[r * rstride + kofs + (c2 * 64 + lane) * 8];
      }
      if constexpr (s % 3 == 0) { process(s, buf0); if constexpr (s + 3 < NS) load(s + 3, buf0); }
      else if constexpr (s % 3 == 1) { process(s, buf1); if constexpr (s + 3 < NS) load(s + 3, buf1); }
      else { process(s, buf2); if constexpr (s + 3 < NS) load(s + 3, buf2); }
    });
    __syncthreads();                                      // A(last op)
    return;
  }

  // -------------------------------------------------------------------------------------- communication wave
  // operands that do not depend on this launch's hand-offs are requested up front
  u32x4 g[NCH][R];                                         // gathered vector in gemv_kernel's lane layout
  u32x4 l2w[NCH], l2b[NCH], lnw[NCH], lnbb[NCH];
#pragma unroll
  for (int c2 = 0; c2 < NCH; ++c2)
#pragma unroll
    for (int r = 0; r < R; ++r) g[c2][r] = ld16(a.a + (size_t)r * D + (c2 * 64 + lane) * 8);
#pragma unroll
  for (int c2 = 0; c2 < NCH; ++c2) { l2w[c2] = ld16(a.ln2_w + (c2 * 64 + lane) * 8); l2b[c2] = ld16(a.ln2_b + (c2 * 64 + lane) * 8); }
  if constexpr (T_IN > 0) {
#pragma unroll
    for (int c2 = 0; c2 < NCH; ++c2) { lnw[c2] = ld16(a.lnn_w + (c2 * 64 + lane) * 8); lnbb[c2] = ld16(a.lnn_b + (c2 * 64 + lane) * 8); }
  }
  // items of the row-pair ops (0, 1, 3, 4): lane = j * R + r
  const int ij = lane >> 1, ir = lane & 1;
  const bool it_out = ij < ppw_out;                        // ops 0, 1, 3 (same units: d/2 pairs)
  const int u_out = c * ppw_out + (it_out ? ij : 0);
  unsigned resid = 0;
  if (it_out) resid = *(const unsigned*)(a.xin + (size_t)ir * D + 2 * u_out);
  const bool it_in = T_IN > 0 && ij < ppw_in;
  const int u_in = c * ppw_in + (it_in ? ij : 0);
  int pos = 0; float cs = 1.f, sn = 0.f;
  if constexpr (T_IN > 0) {
    if (it_in) {
      pos = a.lengths[ir];
      const int rowA = 2 * u_in;
      if (rowA < (a.n_heads + a.n_heads_kv) * a.hd) {
        const int i = (rowA % a.hd) >> 1;
        const int p = pos < a.rope_positions ? pos : a.rope_positions - 1;
        const float2 c2v = *(const float2*)(a.rope + ((size_t)p * (a.hd >> 1) + i) * 2);
        cs = c2v.x; sn = c2v.y;
      }
    }
  }
#pragma unroll
  for (int c2 = 0; c2 < NCH; ++c2)
#pragma unroll
    for (int r = 0; r < R; ++r) *(u32x4*)&s_act[r * D + (c2 * 64 + lane) * 8] = g[c2][r];
  __syncthreads();                                         // B(0)
  int nst = 0;
  auto stamp = [&]() { if (a.stamps && c == 0 && lane == 0) a.stamps[nst] = __builtin_amdgcn_s_memrealtime(); ++nst; };
  stamp();

  const __amdgpu_buffer_rsrc_t rs_y1 = zn_rsrc(a.y1), rs_x1 = zn_rsrc(a.x1), rs_xo = zn_rsrc(a.xout), rs_m = zn_rsrc(a.m);
  unsigned x1own = 0;
  zn_static_for<0, NOPS>([&](auto OC) {
    constexpr int op = decltype(OC)::value;
    __syncthreads();                                       // A(op): every compute wave's results are in LDS
    stamp();
    // ---- epilogue of this workgroup's units, published write-through
    if constexpr (op == 0) {                               // EPI_STORE
      if (it_out) st_sc1_u32(a.y1 + (size_t)ir * D + 2 * u_out, pack2(s_res[ij][0][ir], s_res[ij][1][ir]));
    } else if constexpr (op == 1) {                        // EPI_RESID
      if (it_out) {
        x1own = pack2(lo_f(resid) + bfround(s_res[ij][0][ir]), hi_f(resid) + bfround(s_res[ij][1][ir]));
        st_sc1_u32(a.x1 + (size_t)ir * D + 2 * u_out, x1own);
      }
    } else if constexpr (op == 2) {                        // EPI_SILU: lane = r * ppw_fc1 + j, neighbours pack a dword
      const int r2 = lane / ppw_fc1, j2 = lane % ppw_fc1;
      const bool on = r2 < R;
      const int jj = on ? j2 : 0, rr = on ? r2 : 0;
      const float y = bfround(s_res[jj][0][rr]), gt = bfround(s_res[jj][1][rr]);
      const float sg = bfround(gt / (1.0f + expf(-gt)));
      const unsigned mine = (unsigned)f2bf(y * sg);
      const unsigned nb = (unsigned)__shfl_down((int)mine, 1);
      if (on && (j2 & 1) == 0) st_sc1_u32(a.m + (size_t)r2 * F + c * ppw_fc1 + j2, mine | (nb << 16));
    } else if constexpr (op == 3) {                        // EPI_RESID over the four K quarters, in gemv_kernel's order
      if (it_out) {
        const float vA = ((s_res[ij * 4 + 0][0][ir] + s_res[ij * 4 + 1][0][ir]) + s_res[ij * 4 + 2][0][ir]) + s_res[ij * 4 + 3][0][ir];
        const float vB = ((s_res[ij * 4 + 0][1][ir] + s_res[ij * 4 + 1][1][ir]) + s_res[ij * 4 + 2][1][ir]) + s_res[ij * 4 + 3][1][ir];
        const unsigned o = pack2(lo_f(x1own) + bfround(vA), hi_f(x1own) + bfround(vB));
        if constexpr (NOPS == 5) st_sc1_u32(a.xout + (size_t)ir * D + 2 * u_out, o);
        else *(unsigned*)(a.xout + (size_t)ir * D + 2 * u_out) = o;         // last op of the launch: the next kernel reads it
      }
    } else {                                               // EPI_ROPE_KV of the next block (read by the next launch)
      if (it_in) {
        GemvArgs ga{};
        ga.hd = a.hd; ga.n_heads = a.n_heads; ga.n_heads_kv = a.n_heads_kv; ga.q_out = a.q_out; ga.kv = a.kv; ga.max_len = a.max_len;
        gemv_epilogue<EPI_ROPE_KV>(ga, ir, 2 * u_in, 2 * u_in + 1, true, u_in, s_res[ij][0][ir], s_res[ij][1][ir], 0u, cs, sn, pos);
      }
    }
    if constexpr (op + 1 < NOPS) {
      // ---- arrive (stores drained first), wait for every workgroup, gather the op's output vector
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned* ctr = a.ctr + op * ZN_CH_CTR_WORDS;
      if (lane == 0) __hip_atomic_fetch_add(ctr + (c % ZN_CH_NSHARD) * ZN_CH_SSTRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      stamp();
      chain_wait(ctr, (unsigned)G, a.tmo, lane);
      stamp();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // no instruction: keeps the gather below the poll
      if constexpr (op == 2) {
        // m [R][4 d]: 8 x NCH wave-wide 16-B loads, staged through registers in groups of 8
#pragma unroll
        for (int grp = 0; grp < NCH; ++grp) {
          u32x4 t8[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) t8[i] = ld_sc1_16(rs_m, ((grp * 8 + i) * 64 + lane) * 16);
#pragma unroll
          for (int i = 0; i < 8; ++i) *(u32x4*)&s_act[((grp * 8 + i) * 64 + lane) * 8] = t8[i];
        }
      } else {
        const __amdgpu_buffer_rsrc_t rs = (op == 0) ? rs_y1 : (op == 1) ? rs_x1 : rs_xo;
#pragma unroll
        for (int c2 = 0; c2 < NCH; ++c2)
#pragma unroll
          for (int r = 0; r < R; ++r) g[c2][r] = ld_sc1_16(rs, (r * D + (c2 * 64 + lane) * 8) * 2);
        if constexpr (op == 1) chain_layernorm<NCH>(g, l2w, l2b, a.eps);
        if constexpr (op == 3) chain_layernorm<NCH>(g, lnw, lnbb, a.eps);
#pragma unroll
        for (int c2 = 0; c2 < NCH; ++c2)
#pragma unroll
          for (int r = 0; r < R; ++r) *(u32x4*)&s_act[r * D + (c2 * 64 + lane) * 8] = g[c2][r];
      }
      __syncthreads();                                     // B(op + 1)
      stamp();
    }
  });
  stamp();
}
