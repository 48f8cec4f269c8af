// Whole decode step of the transformer stack at batch 1 (R = 2 rows) in ONE persistent launch: for every block
//
//   op A  a = attention(q, K, V)  (32 of the 256 workgroups: one per (row, kv head, 32-wide value slice); zn_chain_kernel.h,
//         stack_attention: attn_pv_kernel<128, 4, 1>'s arithmetic on the workgroup's 4 compute waves)
//   ops 0-4 of chain_kernel (out_proj twice, LayerNorm + fc1 + SiLU-gate, fc2, next block's LayerNorm + in_proj + RoPE + KV append;
//         last block: norm_f + fused heads)
//
// with the same static tile schedule, register tile buffers and tagged-granule hand-offs as chain_kernel, plus two more
// hand-offs per block: q | k | v of the next block leave op 4's epilogue as granules (the KV cache row is written as well, for
// later steps), and the attention output reaches every workgroup as granules.  Tag of block li = epoch + li.  What the
// launch-per-block path pays between two chain launches (two kernel boundaries, the attention launch's cold start, the chain's
// ramp until its first tiles arrive) is replaced by two hand-offs, and the next block's first weight tiles are requested while
// the attention runs.  Block 0's q / K / V come from the in_proj launch before this one (plain loads).
// Results are bit-identical to the per-block path (same arithmetic, same order).
#pragma once
#include "zn_chain_kernel.h"

#ifndef ZN_ST_POLL_DELAY
#define ZN_ST_POLL_DELAY 300                               // 10 ns ticks a workgroup without attention work sleeps before it starts polling for the attention output
#endif

template <int NCH, int T_OUT, int T_FC1, int T_FC2, int T_IN>
__global__ __launch_bounds__(ZN_CH_THREADS) void stack_kernel(ChainArgs a) {
  constexpr int R = 2, D = NCH * 512, CW = ZN_CH_CWAVES;
  constexpr int S1 = T_OUT, S2 = 2 * T_OUT, S3 = S2 + T_FC1, S4 = S3 + T_FC2, NS = S4 + T_IN;
  constexpr int NOPS = 5;
  constexpr int MASK = ZN_CH_DEFER_MASK;
  // Two tile buffers, not chain_kernel's three: with three the compute waves spilled 27 VGPRs to scratch around the attention.  A
  // kernel whose hand-offs need every workgroup resident must not depend on scratch-wave slots being granted to all of them at
  // once (one hand-off timeout was observed in ~10 generations with the spilling build; none since).
  constexpr int NB = 2;
  constexpr int NL = NS - T_OUT, INIT = T_OUT + 1 < NB ? T_OUT + 1 : NB;
  static_assert(T_IN > 0 && NCH == 4, "stack_kernel: d_model 2048, head size 128");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = blockIdx.x, G = gridDim.x;
  const int F = a.F;
  const unsigned tag0 = *a.epoch;
  const int ppw_out = (D / 2) / G, ppw_fc1 = F / G, ppw_fc2 = (D / 2) / G;
  __shared__ __attribute__((aligned(16))) bf16_t s_act[R * D];
  __shared__ float s_res[2][64][2][R];
  extern __shared__ __attribute__((aligned(16))) unsigned char zn_dyn_lds[];
  StackAttnLds& AL = *reinterpret_cast<StackAttnLds*>(zn_dyn_lds);
  // attention role of this workgroup (attn_pv_kernel's fused grid: blockIdx.x = pair + npairs * slice)
  const int npairs = a.n_heads_kv * R, natt = npairs * (a.hd / 32);
  const bool att = c < natt;
  const int pair = c % npairs, slice = att ? c / npairs : 0, kvh = pair % a.n_heads_kv, ar = pair / a.n_heads_kv;
  const int Gq = a.n_heads / a.n_heads_kv;                 // 4 (host-checked)
  const int nq = a.n_heads * a.hd, nk = a.n_heads_kv * a.hd;

  auto op_of = [](int s) constexpr { return s < S1 ? 0 : s < S2 ? 1 : s < S3 ? 2 : s < S4 ? 3 : 4; };
  auto first_of = [](int op) constexpr { return op == 0 ? 0 : op == 1 ? S1 : op == 2 ? S2 : op == 3 ? S3 : S4; };

  if (wave < CW) {
    // ------------------------------------------------------------------------------------ compute waves
    struct WT { u32x4 a[NCH], b[NCH]; };
    WT bufs[NB];
    static_assert(NB >= 2 && CW % 4 == 0, "rotating tile buffers; fc2 splits K over groups of four waves");
    const int L = att ? a.lengths[ar] + 1 : 1;
#pragma unroll 1
    for (int li = 0; li < a.n_layer; ++li) {
      const StackLayer& Lr = a.layers[li];
      const bf16_t *W_out = Lr.W_out, *W_fc1 = Lr.W_fc1, *W_fc2 = Lr.W_fc2, *W_in = Lr.W_in;
      const bool last = li + 1 == a.n_layer;
      const int rows_in = last ? a.heads_rows : a.nqkv, units_in = (rows_in + 1) / 2, ppw_in = (units_in + G - 1) / G;
      const unsigned tag = tag0 + (unsigned)li;
      auto tile = [&](int s, bool& ok, const bf16_t*& pa, const bf16_t*& pb, int& ridx) {
        const int op = op_of(s), t = s - first_of(op);
        if (op == 3) {
          const int qt = wave & 3, j = (wave >> 2) + (CW / 4) * t;
          ok = j < ppw_fc2;
          const int u = c * ppw_fc2 + (ok ? j : 0);
          pa = W_fc2 + (size_t)(2 * u) * (4 * D) + qt * D + lane * 8;
          pb = pa + 4 * D;
          ridx = j * 4 + qt;
          return;
        }
        const int j = wave + CW * t;
        const int ppw = op <= 1 ? ppw_out : op == 2 ? ppw_fc1 : ppw_in;
        ok = j < ppw && (op < 4 || c * ppw + j < units_in);
        const int u = c * ppw + (ok ? j : 0);
        ridx = j;
        if (op <= 1) { pa = W_out + (size_t)(2 * u) * D + lane * 8; pb = pa + D; }
        else if (op == 2) { pa = W_fc1 + (size_t)u * D + lane * 8; pb = pa + (size_t)F * D; }
        else { pa = W_in + (size_t)(2 * u) * D + lane * 8; pb = (2 * u + 1 < rows_in) ? pa + D : pa; }
      };
      auto load = [&](int s, WT& w) {
        bool ok; const bf16_t *pa, *pb; int ridx;
        tile(s, ok, pa, pb, ridx);
        if (ok) {
#pragma unroll
          for (int c2 = 0; c2 < NCH; ++c2) { w.a[c2] = ld_nt16(pa + c2 * 512); w.b[c2] = ld_nt16(pb + c2 * 512); }
        }
      };
      auto slot_of_load = [](int l) constexpr { return l < S1 ? l : l + T_OUT; };
      auto load_req = [&](auto LC) { constexpr int l = decltype(LC)::value; load(slot_of_load(l), bufs[l % NB]); };
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
          const int par = op_of(s) & 1;
#pragma unroll
          for (int r = 0; r < R; ++r) { s_res[par][ridx][0][r] = accA[r]; s_res[par][ridx][1][r] = accB[r]; }
        }
      };
      auto raised_by = [](int s) constexpr {
        const int op = s < S1 ? 0 : s < S2 ? 1 : s < S3 ? 2 : s < S4 ? 3 : 4;
        if (op == 0) return -1;
        const int l = (s - T_OUT) + NB;
        return l < NL ? l : -1;
      };
      u32x4 kk0[4][4], vv0[4];
      __syncthreads();                                    // T(li): block 0: the communication waves' own requests are queued; later: op 4 of the block before is published
      // the first K batch does not depend on q: requested now, ahead of the weight tiles in this CU's queue (and behind the publish above)
      if (att) {
        // (no weight tile is held in registers across the attention: with op 0's tiles requested first the kernel spilled 150-170 VGPRs;
        // op 0 cannot start before the attention output has been swept anyway, ~2 us after these workgroups publish it)
        // the first batch's key rows and value pieces do not depend on q: requested now, they arrive during the hand-off
        stack_attn_issue_k(kk0, Lr.kv, ar, kvh, L, a.max_len, a.n_heads_kv, wave, 0, lane);
        stack_attn_issue_v(vv0, Lr.kv, ar, kvh, slice, L, a.max_len, a.n_heads_kv, wave, 0, lane);
        __syncthreads();                                  // A1: q, newest key and value rows are in LDS
        stack_attention(AL, kk0, vv0, Lr.kv, ar, kvh, slice, L, a.max_len, a.n_heads_kv, a.scale, wave, lane,
                        (a.stamps && li == a.stamp_layer && c == 0) ? a.stamps + 32 : nullptr);   // A2, A3 inside
        __syncthreads();                                  // A4: the result slice is in LDS
      }
      // a tile that does not exist leaves its buffer untouched: tell the compiler that nothing of the block before survives
      zn_static_for<0, NB>([&](auto BC) { bufs[decltype(BC)::value] = WT{}; });
      zn_static_for<0, INIT>([&](auto LC) { load_req(LC); });
      zn_static_for<0, NS>([&](auto SC) {
        constexpr int s = decltype(SC)::value;
        constexpr int op = op_of(s);
        if constexpr (s == first_of(op)) {
          if constexpr (op > 0) {
            __syncthreads();                              // A(op-1)
            __syncthreads();                              // P(op-1)
            if constexpr (op == 1) zn_static_for<INIT, (NL < NB ? NL : NB)>([&](auto LC) { load_req(LC); });
            if constexpr (((MASK >> (op - 1)) & 1) != 0) {
              zn_static_for<first_of(op - 1), first_of(op)>([&](auto QC) {
                constexpr int q = decltype(QC)::value, l = raised_by(q);
                if constexpr (l >= 0) { if constexpr (op_of(slot_of_load(l >= 0 ? l : 0)) != op - 1) load_req(std::integral_constant<int, (l >= 0 ? l : 0)>{}); }
              });
            }
          }
          if constexpr (op == 3) {
            const int qt = wave & 3;
            int off[NCH * R];
            u32x4 dat[NCH * R];
#pragma unroll
            for (int c2 = 0; c2 < NCH; ++c2)
#pragma unroll
              for (int r = 0; r < R; ++r) off[c2 * R + r] = (r * (2 * D) + qt * (D / 2) + (c2 * 64 + lane) * 4) * 8;
            sweep_granules<NCH * R>(zn_rsrc(a.g_m), off, tag, dat, a.tmo, lane);
#pragma unroll
            for (int c2 = 0; c2 < NCH; ++c2)
#pragma unroll
              for (int r = 0; r < R; ++r) xr[c2][r] = dat[c2 * R + r];
          } else {
            __syncthreads();                              // B(op)
#pragma unroll
            for (int c2 = 0; c2 < NCH; ++c2)
#pragma unroll
              for (int r = 0; r < R; ++r) xr[c2][r] = *(const u32x4*)&s_act[r * D + (c2 * 64 + lane) * 8];
          }
        }
        constexpr int lb = (op == 0 ? s : s - T_OUT) % NB;
        process(s, bufs[lb]);
        constexpr int l = raised_by(s);
        if constexpr (l >= 0) {
          if constexpr (((MASK >> op) & 1) == 0 || op_of(slot_of_load(l >= 0 ? l : 0)) == op) load_req(std::integral_constant<int, (l >= 0 ? l : 0)>{});
        }
      });
      __syncthreads();                                    // A(4)
    }
    return;
  }

  // -------------------------------------------------------------------------------------- communication waves
  const int myr = wave - CW;
  const bool epi = myr == 0;
  u32x4 g[NCH];
  u32x4 l2w[NCH], l2b[NCH], lnw[NCH], lnbb[NCH];
  const int ij = lane >> 1, ir = lane & 1;
  const bool it_out = epi && ij < ppw_out;
  const int u_out = c * ppw_out + (it_out ? ij : 0);
  unsigned resid = 0;
  if (it_out) resid = *(const unsigned*)(a.xin + (size_t)ir * D + 2 * u_out);
  // op 4 as the next block's in_proj: unit, position and rotation of this lane (the same in every block)
  const int ppw_qkv = ((a.nqkv + 1) / 2 + G - 1) / G;
  const bool it_qkv = epi && ij < ppw_qkv && c * ppw_qkv + ij < (a.nqkv + 1) / 2;
  const int u_qkv = c * ppw_qkv + (it_qkv ? ij : 0);
  int pos = 0; float cs = 1.f, sn = 0.f;
  if (it_qkv) {
    pos = a.lengths[ir];
    const int rowA = 2 * u_qkv;
    if (rowA < nq + nk) {
      const int i = (rowA % a.hd) >> 1;
      const int p = pos < a.rope_positions ? pos : a.rope_positions - 1;
      const float2 c2v = *(const float2*)(a.rope + ((size_t)p * (a.hd >> 1) + i) * 2);
      cs = c2v.x; sn = c2v.y;
    }
  }
  const int L = att ? a.lengths[ar] + 1 : 1;
  int goff[NCH];
#pragma unroll
  for (int c2 = 0; c2 < NCH; ++c2) goff[c2] = (myr * (D / 2) + (c2 * 64 + lane) * 4) * 8;
  unsigned x1own = 0;
#pragma unroll 1
  for (int li = 0; li < a.n_layer; ++li) {
    const StackLayer& Lr = a.layers[li];
    const bool last = li + 1 == a.n_layer;
    const unsigned tag = tag0 + (unsigned)li;
    const bool stamped = a.stamps && li == a.stamp_layer && epi && c == 0 && lane == 0;
    int nst = 0;
    auto stamp = [&]() { if (stamped) a.stamps[nst] = __builtin_amdgcn_s_memrealtime(); ++nst; };
#pragma unroll
    for (int c2 = 0; c2 < NCH; ++c2) { l2w[c2] = ld16(Lr.ln2_w + (c2 * 64 + lane) * 8); l2b[c2] = ld16(Lr.ln2_b + (c2 * 64 + lane) * 8); }
#pragma unroll
    for (int c2 = 0; c2 < NCH; ++c2) { lnw[c2] = ld16(Lr.lnn_w + (c2 * 64 + lane) * 8); lnbb[c2] = ld16(Lr.lnn_b + (c2 * 64 + lane) * 8); }
    __syncthreads();                                       // T(li)
    stamp();                                               // 0: block starts
    // ---- attention inputs into LDS, result out as granules
    if (att) {
      if (li == 0) {
        // from the launch before: plain loads
        if (epi) *(u32x4*)&AL.q[lane >> 4][(lane & 15) * 8] = ld16(a.q0 + ((size_t)ar * a.n_heads + kvh * Gq + (lane >> 4)) * a.hd + (lane & 15) * 8);
        else if (lane < 32) {
          const size_t kvrow = (size_t)2 * nk;
          const bf16_t* rowp = Lr.kv + ((size_t)ar * a.max_len + (L - 1)) * kvrow + (size_t)(lane < 16 ? 0 : nk) + (size_t)kvh * a.hd + (lane & 15) * 8;
          const u32x4 v = ld16(rowp);
          if (lane < 16) *(u32x4*)&AL.knew[lane * 8] = v; else *(u32x4*)&AL.vnew[(lane - 16) * 8] = v;
        }
      } else {
        // granules of the block before (tag - 1): wave `epi` takes the 4 query heads (512 values), the other wave the key and value rows
        const int qoff = (ar * (a.nqkv / 2) + ((kvh * Gq) * a.hd) / 2 + lane * 4) * 8;
        const int kvsel = lane < 16 ? nq + kvh * a.hd : nq + nk + kvh * a.hd;
        const int koff = (ar * (a.nqkv / 2) + kvsel / 2 + (lane & 15) * 4) * 8;
        int off1[1] = {epi ? qoff : koff};
        u32x4 d1[1];
        sweep_granules<1>(zn_rsrc(a.g_qkv), off1, tag - 1u, d1, a.tmo, lane);
        if (epi) *(u32x4*)&AL.q[lane >> 4][(lane & 15) * 8] = d1[0];
        else if (lane < 16) *(u32x4*)&AL.knew[lane * 8] = d1[0];
        else if (lane < 32) *(u32x4*)&AL.vnew[(lane - 16) * 8] = d1[0];
      }
      __syncthreads();                                     // A1
      stamp();                                             // 1: attention inputs in LDS
      __syncthreads();                                     // A2
      stamp();                                             // 2: scores done
      __syncthreads();                                     // A3
      __syncthreads();                                     // A4
      stamp();                                             // 3: attention result in LDS
      if (epi) {
        const int gq = lane >> 4, dp = lane & 15;
        const unsigned v = *(const unsigned*)&AL.out[gq][2 * dp];
        st_granule(a.g_a + (size_t)ar * (D / 2) + ((kvh * Gq + gq) * a.hd + slice * 32) / 2 + dp, tag, v);
        if (a.trace) {
          *(unsigned*)(a.trace + ((size_t)(8 * li + 1) * R + ar) * D + (kvh * Gq + gq) * a.hd + slice * 32 + 2 * dp) = v;
          if (slice == 0) *(u32x4*)(a.trace + ((size_t)(8 * li + 2) * R + ar) * D + (kvh * Gq + (lane >> 4)) * a.hd + (lane & 15) * 8) = *(const u32x4*)&AL.q[lane >> 4][(lane & 15) * 8];
        }
      }
    } else {
      stamp(); stamp(); stamp();
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ZN_ST_POLL_DELAY) __builtin_amdgcn_s_sleep(16);
    }
    // ---- the attention output of all heads -> s_act
    sweep_granules<NCH>(zn_rsrc(a.g_a), goff, tag, g, a.tmo, lane);
#pragma unroll
    for (int c2 = 0; c2 < NCH; ++c2) *(u32x4*)&s_act[myr * D + (c2 * 64 + lane) * 8] = g[c2];
    __syncthreads();                                       // B(0)
    stamp();                                               // 4: op 0's input ready
    zn_static_for<0, NOPS>([&](auto OC) {
      constexpr int op = decltype(OC)::value;
      constexpr int par = op & 1;
      __syncthreads();                                     // A(op)
      stamp();
      if constexpr (op == 0) {
        if (it_out) st_granule(a.g_y1 + (size_t)ir * (D / 2) + u_out, tag, pack2(s_res[par][ij][0][ir], s_res[par][ij][1][ir]));
      } else if constexpr (op == 1) {
        if (it_out) {
          x1own = pack2(lo_f(resid) + bfround(s_res[par][ij][0][ir]), hi_f(resid) + bfround(s_res[par][ij][1][ir]));
          st_granule(a.g_x1 + (size_t)ir * (D / 2) + u_out, tag, x1own);
        }
      } else if constexpr (op == 2) {
        if (epi) {
          const int r2 = lane / ppw_fc1, j2 = lane % ppw_fc1;
          const bool on = r2 < R;
          const int jj = on ? j2 : 0, rr = on ? r2 : 0;
          const float y = bfround(s_res[par][jj][0][rr]), gt = bfround(s_res[par][jj][1][rr]);
          const float sg = bfround(gt / (1.0f + expf(-gt)));
          const unsigned mine = (unsigned)f2bf(y * sg);
          const unsigned nb = (unsigned)__shfl_down((int)mine, 1);
          if (on && (j2 & 1) == 0) st_granule(a.g_m + (size_t)r2 * (F / 2) + ((c * ppw_fc1 + j2) >> 1), tag, mine | (nb << 16));
        }
      } else if constexpr (op == 3) {
        if (it_out) {
          const float vA = ((s_res[par][ij * 4 + 0][0][ir] + s_res[par][ij * 4 + 1][0][ir]) + s_res[par][ij * 4 + 2][0][ir]) + s_res[par][ij * 4 + 3][0][ir];
          const float vB = ((s_res[par][ij * 4 + 0][1][ir] + s_res[par][ij * 4 + 1][1][ir]) + s_res[par][ij * 4 + 2][1][ir]) + s_res[par][ij * 4 + 3][1][ir];
          const unsigned o = pack2(lo_f(x1own) + bfround(vA), hi_f(x1own) + bfround(vB));
          st_granule(a.g_x2 + (size_t)ir * (D / 2) + u_out, tag, o);
          if (last) *(unsigned*)(a.xout + (size_t)ir * D + 2 * u_out) = o;
          if (a.trace) *(unsigned*)(a.trace + ((size_t)(8 * li) * R + ir) * D + 2 * u_out) = o;
          resid = o;                                       // the residual stream entering the next block
        }
      } else {
        if (last) {                                        // EPI_F32 (gemv_epilogue): bf16-valued fp32 logits
          const int units = (a.heads_rows + 1) / 2, ppw = (units + G - 1) / G;
          if (epi && ij < ppw && c * ppw + ij < units) {
            const int u = c * ppw + ij;
            a.heads_out[(size_t)ir * a.heads_rows + 2 * u] = bfround(s_res[par][ij][0][ir]);
            if (2 * u + 1 < a.heads_rows) a.heads_out[(size_t)ir * a.heads_rows + 2 * u + 1] = bfround(s_res[par][ij][1][ir]);
          }
        } else if (it_qkv) {                               // EPI_ROPE_KV (gemv_epilogue) of the next block, q | k | v also as granules
          const int rowA = 2 * u_qkv;
          const float x0 = bfround(s_res[par][ij][0][ir]), x1 = bfround(s_res[par][ij][1][ir]);
          unsigned packed;
          if (rowA < nq + nk) {
            float re, im;
            zn_rope_pair(x0, x1, cs, sn, re, im);
            packed = pack2(re, im);
          } else packed = pack2(x0, x1);
          st_granule(a.g_qkv + (size_t)ir * (a.nqkv / 2) + u_qkv, tag, packed);
          if (rowA >= nq && pos < a.max_len) {
            const int which = rowA < nq + nk ? 0 : 1, colk = rowA - nq - which * nk;
            *(unsigned*)(Lr.kv_next + (((size_t)ir * a.max_len + pos) * 2 + which) * nk + colk) = packed;
          }
        }
      }
      if constexpr (op + 1 < NOPS) {
        __syncthreads();                                   // P(op)
        stamp();
        if constexpr (op == 2) {
          // The next block's attention reads K and V of its (row, kv head) pair through ONE CU per value slice, cold from HBM (a CU
          // sustains ~40 GB/s there).  The communication waves of the workgroups without attention work are idle until fc2's
          // results: those on the pair's XCD (workgroup c runs on XCD c % 8, pair = c % 8) touch the rows now, so that the
          // attention finds them in its L2.  Rows 0 .. L-2 are from earlier launches; speed only, the bytes are discarded.
          if (!att && !last) {
            const int wpair = c % npairs, wkvh = wpair % a.n_heads_kv, wr = wpair / a.n_heads_kv;
            const int nhelp = (G - natt) / npairs;                       // helper workgroups per pair
            const int hidx = (c - natt) / npairs;
            const int Lw = a.lengths[wr];                                 // rows in the cache before this step's append
            const size_t kvrow = (size_t)2 * nk;
            const bf16_t* base = a.layers[li + 1].kv + (size_t)wr * a.max_len * kvrow + (size_t)wkvh * a.hd;
            // unit = 4 key rows of K (even units) or V (odd units): 64 lanes x 16 B
            const int nunits = 2 * ((Lw + 3) / 4);
            for (int u = hidx * 2 + myr; u < nunits; u += 2 * nhelp) {
              const int key = (u >> 1) * 4 + (lane >> 4);
              const bf16_t* p = base + (size_t)min(key, Lw > 0 ? Lw - 1 : 0) * kvrow + (size_t)(u & 1) * nk + (lane & 15) * 8;
              const u32x4 v = ld16(p);
              asm volatile("" : : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
            }
          }
        }
        if constexpr (op != 2) {
          sweep_granules<NCH>(zn_rsrc(op == 0 ? a.g_y1 : op == 1 ? a.g_x1 : a.g_x2), goff, tag, g, a.tmo, lane);
          stamp();
          if constexpr (op == 1 || op == 3) chain_layernorm_row<NCH>(g, op == 1 ? l2w : lnw, op == 1 ? l2b : lnbb, a.eps);
#pragma unroll
          for (int c2 = 0; c2 < NCH; ++c2) *(u32x4*)&s_act[myr * D + (c2 * 64 + lane) * 8] = g[c2];
          __syncthreads();                                 // B(op + 1)
        } else stamp();
        stamp();
      }
    });
    stamp();
  }
  if (epi && c == 0 && lane == 0) *a.epoch = tag0 + (unsigned)a.n_layer;
}
