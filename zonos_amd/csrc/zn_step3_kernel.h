// Whole decode step at batch 1 in ONE persistent launch, an EXPERIMENT kept for the record (opt-in: zn_debug_tune(15, 4)): the
// workgroups specialised by OP, not only by role.  Bit-identical to the other paths; measured SLOWER than the two-role kernel
// (1.076 vs 0.859 ms per decode step at 400 tokens), and the timeline says why (tools/step3sweep.py, profiles/r03_step3_timeline.txt).
//
// Premise.  zn_step_kernel.h (two roles) measured what bounds a block: its time is the weight stream (121.6 MB at ~5.9 TB/s = 20.6 us)
// PLUS the dependent hand-off phases during which nothing streams (~13 us) - a CU's memory queue is in order, so a sweep on a CU that
// streams weights waits for them, and every byte requested in a hand-off's shadow delays that hand-off by about the time it takes to
// stream.  The small ops are what serialises: out_proj twice and the next in_proj are 21 MB of the block's 121.6 but four of its six
// hand-offs.  Here they run on workgroups that never stream in their way:
//
//  * 32 ATTENTION workgroups - step_attention_role of zn_step_kernel.h, unchanged.
//  * ZN_S3_NPROJ = 64 PROJECTION workgroups hold the block's out_proj slice (2 tiles per wave) and the next block's in_proj slice (3 tiles
//    per wave: one in registers, two in LDS) on chip, requested a block (~25 us) before their use.  Ops 0, 1 and 4 compute the moment
//    their input vector is in LDS, every wave publishes its own results, and the hand-offs around them run on CUs whose queues are idle.
//  * the remaining 160 BULK workgroups stream only fc1 and fc2 (and the heads in the last block); all four LDS park slots and the three
//    register buffers hold fc1 tiles (7 of ~13 per wave) by the time x1 arrives.
//
// What the measurement showed (block 13, 400 tokens):
//  * a hand-off between CUs with IDLE queues costs what it costs between busy ones: y1 among the 64 projection workgroups 2.8 us
//    (2.3 us among 224 streaming workgroups in the two-role kernel) - the latency is the write-through + remote-read round trip and
//    the phase of the polling passes, not queueing;
//  * one CU streams ~24-27 GB/s with 3 register buffers x 4 waves (96 KB in flight), so 160 bulk CUs reach ~4 TB/s, not the
//    5.9 TB/s of 224: fc1 7.9 us (5.4-6.3), m hand-off + fc2 8.4 us (6.3): the per-CU rate, not HBM, bounds the bulk role;
//  * requests raised right after a publish by OTHER waves of the workgroup (the next out_proj slice) sat in front of the last waves'
//    granule stores: x1 took 6 us to reach the bulk workgroups (fixable: request after the barrier - it would not close the gap).
// Block 41 us against 33 us.  Kept opt-in with its parity test; not developed further.
//
// Hand-offs, tags (epoch + block), bounded self-describing waits: as zn_step_kernel.h.  Reuse of the granule buffers stays safe: every
// sweep covers its whole vector, so a stage of block b + 1 can only be published after every producer of the stage before it - and
// therefore every consumer of the same stage of block b - has passed that point (g_a(b+1) <- attention swept q|k|v(b) <- projection
// op 4(b) <- swept x2(b) <- ALL bulk op 3(b) <- each swept m(b) <- ALL bulk op 2(b) <- each swept x1(b) <- ALL projection op 1(b) <- each
// swept y1(b) <- ALL projection op 0(b) <- each swept a(b)).
#pragma once
#include "zn_step_kernel.h"

#ifndef ZN_S3_NPROJ
#define ZN_S3_NPROJ 64
#endif
#define ZN_S3_TO 2                                          // out_proj tiles (row pairs) per projection wave: (d / 2) / (NPROJ * 8)
#define ZN_S3_TI 3                                          // in_proj tiles per projection wave: (nqkv / 2) / (NPROJ * 8)
#define ZN_S3_DYN_LDS (ZN_SK_DYN_LDS + 8192)                // bulk: 128 KB of parked tiles; projection: LayerNorm parameters + 128 KB of parked in_proj tiles

// ------------------------------------------------------------------------------------------------ projection role
template <int NCH>
ZN_DEVINL void step3_projection_role(const ChainArgs& a, const unsigned tag0, const int pc, const int wave, const int lane, bf16_t* s_act, bf16_t* s_x, bf16_t* s_ln) {
  constexpr int R = 2, D = NCH * 512, TO = ZN_S3_TO, TI = ZN_S3_TI;
  struct WT { u32x4 a[NCH], b[NCH]; };
  WT wo[TO], wi[TI];
  const int nq = a.n_heads * a.hd, nk = a.n_heads_kv * a.hd;
  const int u_out0 = (pc * 8 + wave) * TO, u_in0 = (pc * 8 + wave) * TI;     // this wave's first row pair of out_proj / in_proj
  auto load_out = [&](int li) {
    const bf16_t* W = a.layers[li].W_out;
#pragma unroll
    for (int t = 0; t < TO; ++t) {
      const bf16_t* pa = W + (size_t)(2 * (u_out0 + t)) * D + lane * 8;
#pragma unroll
      for (int c2 = 0; c2 < NCH; ++c2) { wo[t].a[c2] = ld_nt16g(pa + c2 * 512); wo[t].b[c2] = ld_nt16g(pa + D + c2 * 512); }
    }
  };
  auto load_in = [&](int li) {                                // layers[li].W_in = the in_proj of block li + 1
    const bf16_t* W = a.layers[li].W_in;
#pragma unroll
    for (int t = 0; t < TI; ++t) {
      const bf16_t* pa = W + (size_t)(2 * (u_in0 + t)) * D + lane * 8;
#pragma unroll
      for (int c2 = 0; c2 < NCH; ++c2) { wi[t].a[c2] = ld_nt16g(pa + c2 * 512); wi[t].b[c2] = ld_nt16g(pa + D + c2 * 512); }
    }
  };
  // Only the first in_proj tile stays in registers: with all five tiles (160 VGPRs) beside the sweeps' loads in flight the role
  // spilled ~100 registers.  The other TI - 1 are parked in this workgroup's LDS (2 x 8 x 8 KB) once they have arrived - at a moment
  // the wave is idle anyway (the next attention output is >= 5 us away) - and read back tile by tile in op 4.
  u32x4* ipark = reinterpret_cast<u32x4*>(s_ln + 2 * D) + (size_t)wave * ((TI - 1) * 2 * NCH * 64) + lane;
  auto park_in = [&]() {
#pragma unroll
    for (int t = 1; t < TI; ++t)
#pragma unroll
      for (int c2 = 0; c2 < NCH; ++c2) { ipark[((t - 1) * 2 * NCH + c2) * 64] = wi[t].a[c2]; ipark[((t - 1) * 2 * NCH + NCH + c2) * 64] = wi[t].b[c2]; }
  };
  load_out(0);
  if (a.n_layer > 1) { load_in(0); park_in(); }
  // the rotation of this wave's in_proj pairs is the same in every block: lane r (< 2) keeps (cos, sin) of row r's position
  const int er = lane & 1;
  const int pos = a.lengths[er];
  float cs[TI], sn[TI];
#pragma unroll
  for (int t = 0; t < TI; ++t) {
    cs[t] = 1.f; sn[t] = 0.f;
    const int rowA = 2 * (u_in0 + t);
    if (rowA < nq + nk) {
      const int i = (rowA % a.hd) >> 1;
      const int p = pos < a.rope_positions ? pos : a.rope_positions - 1;
      const float2 c2v = *(const float2*)(a.rope + ((size_t)p * (a.hd >> 1) + i) * 2);
      cs[t] = c2v.x; sn[t] = c2v.y;
    }
  }
  const bool comm = wave < R;                                 // waves 0, 1 also gather (sweep), normalise and stage row `wave` of every input vector
  int goff[NCH];
#pragma unroll
  for (int c2 = 0; c2 < NCH; ++c2) goff[c2] = (wave * (D / 2) + (c2 * 64 + lane) * 4) * 8;
  if (comm) {                                                 // the residual stream entering block 0
#pragma unroll
    for (int c2 = 0; c2 < NCH; ++c2) *(u32x4*)&s_x[wave * D + (c2 * 64 + lane) * 8] = ld16(a.xin + (size_t)wave * D + (c2 * 64 + lane) * 8);
  }
  const bool stamped = a.stamps && pc == 0 && wave == 0 && lane == 0;
  StepPacer pace_a{0ull, 0u}, pace_x{0ull, 0u};
  pace_a.start();
  // (the input vector is read from LDS chunk by chunk for every tile: held in registers beside the 160 weight registers it spilled)
  auto dots = [&](const WT& w, float& vA, float& vB) {         // this wave's pair against both rows; lane r (< 2) gets row r's two sums
    float accA[R] = {0.f, 0.f}, accB[R] = {0.f, 0.f};
#pragma unroll
    for (int c2 = 0; c2 < NCH; ++c2) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const u32x4 xv = *(const u32x4*)&s_act[r * D + (c2 * 64 + lane) * 8];
        accA[r] = dot8(w.a[c2], xv, accA[r]); accB[r] = dot8(w.b[c2], xv, accB[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { accA[r] = wave_sum(accA[r]); accB[r] = wave_sum(accB[r]); }
    vA = er ? accA[1] : accA[0];
    vB = er ? accB[1] : accB[0];
  };
#pragma unroll 1
  for (int li = 0; li < a.n_layer; ++li) {
    const StackLayer& Lr = a.layers[li];
    const bool last = li + 1 == a.n_layer;
    const unsigned tag = tag0 + (unsigned)li;
    const bool st_on = stamped && li == a.stamp_layer;
    auto stamp = [&](int i) { if (st_on) a.stamps[i] = __builtin_amdgcn_s_memrealtime(); };
    stamp(0);
    u32x4 g[NCH];
    // ---- op 0: y1 = out_proj(a)
    if (comm) {
      pace_a.sleep();
      stamp(1);
      sweep_granules<NCH>(zn_rsrc(a.g_a), goff, tag, g, a.tmo, lane, SweepWho{(6u << 8) | (unsigned)li, a.diag});
      pace_a.done();
#pragma unroll
      for (int c2 = 0; c2 < NCH; ++c2) *(u32x4*)&s_act[wave * D + (c2 * 64 + lane) * 8] = g[c2];
    }
    __syncthreads();                                          // the attention output is in LDS
    stamp(2);
#pragma unroll
    for (int t = 0; t < TO; ++t) {
      float vA, vB;
      dots(wo[t], vA, vB);
      if (lane < R) st_granule(a.g_y1 + (size_t)er * (D / 2) + u_out0 + t, tag, pack2(vA, vB));       // EPI_STORE
    }
    stamp(3);
    __syncthreads();                                          // every wave has read the attention output: s_act may take y1
    // ---- op 1: x1 = x + out_proj(y1)
    if (comm) {
      sweep_granules<NCH>(zn_rsrc(a.g_y1), goff, tag, g, a.tmo, lane, SweepWho{(1u << 8) | (unsigned)li, a.diag});
#pragma unroll
      for (int c2 = 0; c2 < NCH; ++c2) *(u32x4*)&s_act[wave * D + (c2 * 64 + lane) * 8] = g[c2];
    }
    __syncthreads();
    stamp(4);
#pragma unroll
    for (int t = 0; t < TO; ++t) {
      float vA, vB;
      dots(wo[t], vA, vB);
      if (lane < R) {                                          // EPI_RESID
        const unsigned resid = *(const unsigned*)&s_x[er * D + 2 * (u_out0 + t)];
        st_granule(a.g_x1 + (size_t)er * (D / 2) + u_out0 + t, tag, pack2(lo_f(resid) + bfround(vA), hi_f(resid) + bfround(vB)));
      }
    }
    stamp(5);
    // the out_proj registers are free: the next block's slice, a block ahead of its use
    if (!last) load_out(li + 1);
    pace_x.start();
    if (!last && (wave == 2 || wave == 3)) {                  // the next LayerNorm's weight / bias -> LDS (a wave that does not sweep: its registers carry no hand-off state)
      const bf16_t* src = wave == 2 ? Lr.lnn_w : Lr.lnn_b;
#pragma unroll
      for (int c2 = 0; c2 < NCH; ++c2) *(u32x4*)&s_ln[(wave - 2) * D + (c2 * 64 + lane) * 8] = ld16g(src + (c2 * 64 + lane) * 8);
    }
    __syncthreads();                                          // every wave has read y1 (s_act) and its residual pairs (s_x); the LayerNorm parameters are in LDS
    // ---- op 4: q | k | v = in_proj(LayerNorm(x2)) of the next block, RoPE, KV append (the last block's heads are the bulk workgroups')
    if (!last) {
      if (comm) {
        pace_x.sleep();
        stamp(6);
        sweep_granules<NCH>(zn_rsrc(a.g_x2), goff, tag, g, a.tmo, lane, SweepWho{(4u << 8) | (unsigned)li, a.diag});
        pace_x.done();
        stamp(7);
#pragma unroll
        for (int c2 = 0; c2 < NCH; ++c2) *(u32x4*)&s_x[wave * D + (c2 * 64 + lane) * 8] = g[c2];        // the residual stream entering the next block
        u32x4 lnw[NCH], lnbb[NCH];
#pragma unroll
        for (int c2 = 0; c2 < NCH; ++c2) { lnw[c2] = *(const u32x4*)&s_ln[(c2 * 64 + lane) * 8]; lnbb[c2] = *(const u32x4*)&s_ln[D + (c2 * 64 + lane) * 8]; }
        chain_layernorm_row<NCH>(g, lnw, lnbb, a.eps);
#pragma unroll
        for (int c2 = 0; c2 < NCH; ++c2) *(u32x4*)&s_act[wave * D + (c2 * 64 + lane) * 8] = g[c2];
      }
      __syncthreads();
      stamp(8);
  #pragma unroll
      for (int t = 0; t < TI; ++t) {
        float vA, vB;
        if (t == 0) dots(wi[0], vA, vB);
        else {
          WT w;
#pragma unroll
          for (int c2 = 0; c2 < NCH; ++c2) { w.a[c2] = ipark[((t - 1) * 2 * NCH + c2) * 64]; w.b[c2] = ipark[((t - 1) * 2 * NCH + NCH + c2) * 64]; }
          dots(w, vA, vB);
        }
        if (lane < R) {                                        // EPI_ROPE_KV (gemv_epilogue), q | k | v also as granules
          const int u = u_in0 + t, rowA = 2 * u;
          const float x0 = bfround(vA), x1 = bfround(vB);
          unsigned packed;
          if (rowA < nq + nk) {
            float re, im;
            zn_rope_pair(x0, x1, cs[t], sn[t], re, im);
            packed = pack2(re, im);
          } else packed = pack2(x0, x1);
          st_granule(a.g_qkv + (size_t)er * (a.nqkv / 2) + u, tag, packed);
          if (rowA >= nq && pos < a.max_len) {
            const int which = rowA < nq + nk ? 0 : 1, colk = rowA - nq - which * nk;
            *(unsigned*)(Lr.kv_next + (((size_t)er * a.max_len + pos) * 2 + which) * nk + colk) = packed;
          }
        }
      }
      stamp(9);
      // the in_proj registers and LDS slots are free: the slice of the block after the next.  (One conditional region around request
      // and parking: split in two, the compiler kept the parked tiles alive across the loop - and spilled them.)
      if (li + 2 < a.n_layer) {
        load_in(li + 1);
        __syncthreads();                                      // every wave has read LayerNorm(x2): s_act may take the next attention output
        park_in();                                            // (waits for the tiles: the next attention output is a q hand-off + an attention away)
      } else __syncthreads();
    }
    pace_a.start();
  }
}

// ------------------------------------------------------------------------------------------------ the launch
// T_* = tiles per bulk compute wave per op (upper bounds over the 160 bulk workgroups).
template <int NCH, int T_FC1, int T_FC2, int T_IN>
__global__ __launch_bounds__(ZN_SK_THREADS) void step3_kernel(ChainArgs a) {
  constexpr int R = 2, D = NCH * 512, CW = ZN_SK_CW;
  constexpr int NB = ZN_SK_NBUF, P = ZN_SK_PARK;
  using SC = StepSched<0, T_FC1, T_FC2, T_IN, NB, P, 0, ZN_CH_DEFER_MASK, 0>;
  constexpr int NS = SC::NS, NL = SC::NL, NREG = SC::NREG;
  static_assert(NCH == 4 && P + NB <= T_FC1, "step3_kernel: d_model 2048; the prefetch holds fc1 tiles only");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = blockIdx.x;
  const int F = a.F;
  const unsigned tag0 = ld_sc1_u32(a.epoch);
  extern __shared__ __attribute__((aligned(16))) unsigned char zn_dyn_lds[];
  __shared__ __attribute__((aligned(16))) bf16_t s_act[R * D];              // the current op's input vector
  __shared__ __attribute__((aligned(16))) bf16_t s_x[R * D];                // projection role: the residual stream entering the block; bulk role: x1 (raw)
  __shared__ float s_res[2][64][2][R];                                      // bulk role: per-unit results of even / odd ops (fc2: [unit * 4 + quarter])
  const int natt = a.n_heads_kv * R * (a.hd / 32);
#ifndef ZN_S3_ONLY
#define ZN_S3_ONLY 0                                        // register-pressure study: 1 / 2 / 3 = compile only the attention / projection / bulk role
#endif
  if (c < natt) {
    if (ZN_S3_ONLY == 0 || ZN_S3_ONLY == 1) step_attention_role(a, *reinterpret_cast<StepAttnLds*>(zn_dyn_lds), tag0, c, wave, lane);
    return;
  }
  if (c < natt + ZN_S3_NPROJ) {
    if (ZN_S3_ONLY == 0 || ZN_S3_ONLY == 2) step3_projection_role<NCH>(a, tag0, c - natt, wave, lane, s_act, s_x, reinterpret_cast<bf16_t*>(zn_dyn_lds));
    return;
  }
  if (ZN_S3_ONLY != 0 && ZN_S3_ONLY != 3) return;
  // ------------------------------------------------------------------------------------------------ bulk role
  const int bc = c - natt - ZN_S3_NPROJ, NBW = (int)gridDim.x - natt - ZN_S3_NPROJ;
  auto split = [&](int U, int& n, int& start) { const int base = U / NBW, extra = U % NBW; n = base + (bc < extra ? 1 : 0); start = bc * base + min(bc, extra); };
  int n_out, s_out, ng1, sg1, n_hd, s_hd;
  split(D / 2, n_out, s_out);                                 // row pairs of fc2
  split(F / 2, ng1, sg1);                                     // fc1: pairs of m elements (one granule each)
  const int n_fc1 = 2 * ng1, s_fc1 = 2 * sg1;
  split((a.heads_rows + 1) / 2, n_hd, s_hd);
  auto op_of = [](int s) constexpr { return SC::op_of(s); };
  auto first_of = [](int op) constexpr { return SC::first_of(op); };
  auto slot_of_load = [](int l) constexpr { return SC::slot_of_load(l); };
  struct WT { u32x4 a[NCH], b[NCH]; };
  struct LW { const bf16_t *fc1, *fc2, *in; };                 // read from the layer table once per block (SGPRs)
  auto tile_of = [&](const LW Lr, int n_in, int s, int w, bool& ok, const bf16_t*& pa, const bf16_t*& pb, int& ridx) {
    const int op = op_of(s), t = s - first_of(op);
    if (op == 3) {
      const int qt = w, j = t;
      ok = j < n_out;
      const int u = s_out + (ok ? j : 0);
      pa = Lr.fc2 + (size_t)(2 * u) * (4 * D) + qt * D + lane * 8;
      pb = pa + 4 * D;
      ridx = j * 4 + qt;
      return;
    }
    const int j = w + CW * t;
    const int n = op == 2 ? n_fc1 : n_in;
    const int st = op == 2 ? s_fc1 : s_hd;
    ok = j < n;
    const int u = st + (ok ? j : 0);
    ridx = j;
    if (op == 2) { pa = Lr.fc1 + (size_t)u * D + lane * 8; pb = pa + (size_t)F * D; }
    else { pa = Lr.in + (size_t)(2 * u) * D + lane * 8; pb = (2 * u + 1 < a.heads_rows) ? pa + D : pa; }   // heads: half pair, row B repeats row A, result dropped
  };

  if (wave >= CW + 2) {                                       // idle waves (the launch's shape is the attention's): the barriers only
#pragma unroll 1
    for (int li = 0; li < a.n_layer; ++li) {
      const int nbar = li + 1 == a.n_layer ? 8 : 5;
      for (int b = 0; b < nbar; ++b) __syncthreads();
    }
    return;
  }

  if (wave < CW) {
    // ------------------------------------------------------------------------------------ compute waves
    WT bufs[NB];
    u32x4* park = reinterpret_cast<u32x4*>(zn_dyn_lds) + (size_t)wave * (P * 2 * NCH * 64) + lane;
    const bool st_on = a.stamps && bc == 0 && wave == 0 && lane == 0;
#pragma unroll 1
    for (int li = 0; li < a.n_layer; ++li) {
      const LW Lr{a.layers[li].W_fc1, a.layers[li].W_fc2, a.layers[li].W_in};
      const bool last = li + 1 == a.n_layer;
      const int n_in = last ? n_hd : 0;                      // op 4 exists in the last block only (norm_f + heads)
      const unsigned tag = tag0 + (unsigned)li;
      const bool stamped = st_on && li == a.stamp_layer;
      auto cstamp = [&](int i) { if (stamped) a.stamps[40 + i] = __builtin_amdgcn_s_memrealtime(); };
      auto tile = [&](int s, bool& ok, const bf16_t*& pa, const bf16_t*& pb, int& ridx) { tile_of(Lr, n_in, s, wave, ok, pa, pb, ridx); };
      auto load_into = [&](auto LC, auto BC) {
        constexpr int l = decltype(LC)::value, b = decltype(BC)::value;
        if constexpr (l >= 0 && l < NL) {
          bool ok; const bf16_t *pa, *pb; int ridx;
          tile(slot_of_load(l), ok, pa, pb, ridx);
          if (ok) {                                         // wave-uniform
            WT& w = bufs[b];
#pragma unroll
            for (int c2 = 0; c2 < NCH; ++c2) { w.a[c2] = ld_nt16g(pa + c2 * 512); w.b[c2] = ld_nt16g(pb + c2 * 512); }
          }
        }
      };
      auto reg_req = [&](auto KC) {
        constexpr int k = decltype(KC)::value;
        if constexpr (k >= 0 && k < NREG) load_into(std::integral_constant<int, SC::nth_reg(k >= 0 && k < NREG ? k : 0)>{}, std::integral_constant<int, ((k >= 0 ? k : 0) + P) % NB>{});
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
          const int par = op_of(s) & 1;
#pragma unroll
          for (int r = 0; r < R; ++r) { s_res[par][ridx][0][r] = accA[r]; s_res[par][ridx][1][r] = accB[r]; }
        }
      };
      // ---- prefetch while the attention and the projections run: fc1 tiles 0 .. P-1 into LDS, the next NB stay in flight in registers
      cstamp(0);
      zn_static_for<0, NB>([&](auto IC) {
        constexpr int i = decltype(IC)::value;
        if constexpr (i < P) load_into(std::integral_constant<int, i>{}, std::integral_constant<int, i % NB>{});
        else reg_req(std::integral_constant<int, i - P>{});
      });
      zn_static_for<0, P>([&](auto LC) {
        constexpr int l = decltype(LC)::value;
        bool ok; const bf16_t *pa, *pb; int ridx;
        tile(slot_of_load(l), ok, pa, pb, ridx);
        if (ok) {
          const WT& w = bufs[l % NB];
#pragma unroll
          for (int c2 = 0; c2 < NCH; ++c2) { park[(l * 2 * NCH + c2) * 64] = w.a[c2]; park[(l * 2 * NCH + NCH + c2) * 64] = w.b[c2]; }
        }
        constexpr int i = l + NB;
        if constexpr (i < P) load_into(std::integral_constant<int, i>{}, std::integral_constant<int, i % NB>{});
        else reg_req(std::integral_constant<int, i - P>{});
      });
      cstamp(1);
      zn_static_for<0, NS>([&](auto SCV) {
        constexpr int s = decltype(SCV)::value;
        constexpr int op = op_of(s);
        bool run = true;
        if constexpr (op == 4) run = last;                  // wave-uniform: the heads' slots (and their barriers) exist in the last block only
        if (run) {
          if constexpr (s == first_of(op)) {
            if constexpr (op > 2) {
              __syncthreads();                              // A(op-1): this workgroup's results of the previous op are in LDS
              __syncthreads();                              // P(op-1): ... and published; the requests held back for that go out now
              zn_static_for<SC::first_of(op - 1), SC::first_of(op)>([&](auto QC) {
                constexpr int q = decltype(QC)::value;
                if constexpr (SC::raise_late(q)) reg_req(std::integral_constant<int, SC::raised_by(q)>{});
              });
            }
            if constexpr (op == 3) {
              // fc2's input m [2][4 d]: this wave's K quarter straight from the granules (no LDS, no barrier)
              int off[NCH * R];
              u32x4 dat[NCH * R];
#pragma unroll
              for (int c2 = 0; c2 < NCH; ++c2)
#pragma unroll
                for (int r = 0; r < R; ++r) off[c2 * R + r] = (r * (2 * D) + wave * (D / 2) + (c2 * 64 + lane) * 4) * 8;
              sweep_granules<NCH * R>(zn_rsrc(a.g_m), off, tag, dat, a.tmo, lane, SweepWho{(3u << 8) | (unsigned)li, a.diag});
#pragma unroll
              for (int c2 = 0; c2 < NCH; ++c2)
#pragma unroll
                for (int r = 0; r < R; ++r) xr[c2][r] = dat[c2 * R + r];
            } else {
              __syncthreads();                              // B(op): the op's input vector is in LDS
#pragma unroll
              for (int c2 = 0; c2 < NCH; ++c2)
#pragma unroll
                for (int r = 0; r < R; ++r) xr[c2][r] = *(const u32x4*)&s_act[r * D + (c2 * 64 + lane) * 8];
            }
            cstamp(2 * op - 2);
          }
          constexpr int l = SC::load_of_slot(s);
          if constexpr (SC::src_of(l) != 0) {               // parked tile: from LDS
            bool ok; const bf16_t *pa, *pb; int ridx;
            tile(s, ok, pa, pb, ridx);
            if (ok) {
              constexpr int sl = SC::slot_of(l);
              WT w;
#pragma unroll
              for (int c2 = 0; c2 < NCH; ++c2) { w.a[c2] = park[(sl * 2 * NCH + c2) * 64]; w.b[c2] = park[(sl * 2 * NCH + NCH + c2) * 64]; }
              process(s, w);
            }
          } else {
            constexpr int k = SC::regk(l);
            process(s, bufs[(k + P) % NB]);
            if constexpr (SC::raise_now(s)) reg_req(std::integral_constant<int, SC::raised_by(s)>{});
          }
          if constexpr (s + 1 == NS || op_of(s + 1 < NS ? s + 1 : s) != op) cstamp(2 * op - 1);
        }
      });
      __syncthreads();                                      // A(3), or A(4) in the last block
      __syncthreads();                                      // P(3) / P(4): published; the next block's prefetch may enter the CU's queue
    }
    return;
  }

  // -------------------------------------------------------------------------------------- communication waves
  const int myr = wave - CW;
  const bool epi = myr == 0;
  u32x4 g[NCH];
  u32x4 l2w[NCH], l2b[NCH];
  const int ij = lane >> 1, ir = lane & 1;
  const bool it_out = epi && ij < n_out;
  const int u_out = s_out + (it_out ? ij : 0);
  int goff[NCH];
#pragma unroll
  for (int c2 = 0; c2 < NCH; ++c2) goff[c2] = (myr * (D / 2) + (c2 * 64 + lane) * 4) * 8;
  StepPacer pace{0ull, 0u};
  pace.start();
#pragma unroll 1
  for (int li = 0; li < a.n_layer; ++li) {
    const StackLayer& Lr = a.layers[li];
    const bool last = li + 1 == a.n_layer;
    const unsigned tag = tag0 + (unsigned)li;
    const bool stamped = a.stamps && li == a.stamp_layer && epi && bc == 0 && lane == 0;
    int nst = 0;
    auto stamp = [&]() { if (stamped) a.stamps[16 + nst] = __builtin_amdgcn_s_memrealtime(); ++nst; };
    stamp();                                               // 0: block starts
#pragma unroll
    for (int c2 = 0; c2 < NCH; ++c2) { l2w[c2] = ld16g(Lr.ln2_w + (c2 * 64 + lane) * 8); l2b[c2] = ld16g(Lr.ln2_b + (c2 * 64 + lane) * 8); }
    // ---- x1 (the projection workgroups' op 1) -> s_x (raw: fc2's residual) and LayerNorm2(x1) -> s_act
    pace.sleep();
    stamp();                                               // 1: polling starts
    sweep_granules<NCH>(zn_rsrc(a.g_x1), goff, tag, g, a.tmo, lane, SweepWho{(2u << 8) | (unsigned)li, a.diag});
    pace.done();
    stamp();                                               // 2: x1 swept
#pragma unroll
    for (int c2 = 0; c2 < NCH; ++c2) *(u32x4*)&s_x[myr * D + (c2 * 64 + lane) * 8] = g[c2];
    chain_layernorm_row<NCH>(g, l2w, l2b, a.eps);
#pragma unroll
    for (int c2 = 0; c2 < NCH; ++c2) *(u32x4*)&s_act[myr * D + (c2 * 64 + lane) * 8] = g[c2];
    __syncthreads();                                       // B(2)
    stamp();                                               // 3: fc1's input ready
    const unsigned x1own = it_out ? *(const unsigned*)&s_x[ir * D + 2 * u_out] : 0u;
    // ---- op 2: m = y * silu(gate)
    __syncthreads();                                       // A(2)
    stamp();                                               // 4
    {
      constexpr int par = 0;                               // op 2
      const bool on = lane < n_fc1;
      const int jj = on ? lane : 0;
      const float y = bfround(s_res[par][jj][0][myr]), gt = bfround(s_res[par][jj][1][myr]);
      const float sg = bfround(gt / (1.0f + expf(-gt)));
      const unsigned mine = (unsigned)f2bf(y * sg);
      const unsigned nbv = (unsigned)__shfl_down((int)mine, 1);
      if (on && (lane & 1) == 0) st_granule(a.g_m + (size_t)myr * (F / 2) + ((s_fc1 + lane) >> 1), tag, mine | (nbv << 16));
    }
    __syncthreads();                                       // P(2)
    stamp();                                               // 5
    // ---- op 3: x2 = x1 + fc2(m) (the compute waves sweep m themselves)
    __syncthreads();                                       // A(3)
    stamp();                                               // 6
    if (it_out) {
      constexpr int par = 1;                               // op 3
      const float vA = ((s_res[par][ij * 4 + 0][0][ir] + s_res[par][ij * 4 + 1][0][ir]) + s_res[par][ij * 4 + 2][0][ir]) + s_res[par][ij * 4 + 3][0][ir];
      const float vB = ((s_res[par][ij * 4 + 0][1][ir] + s_res[par][ij * 4 + 1][1][ir]) + s_res[par][ij * 4 + 2][1][ir]) + s_res[par][ij * 4 + 3][1][ir];
      const unsigned o = pack2(lo_f(x1own) + bfround(vA), hi_f(x1own) + bfround(vB));
      st_granule(a.g_x2 + (size_t)ir * (D / 2) + u_out, tag, o);
      if (last) *(unsigned*)(a.xout + (size_t)ir * D + 2 * u_out) = o;
      if (a.trace) *(unsigned*)(a.trace + ((size_t)(8 * li) * R + ir) * D + 2 * u_out) = o;
    }
    __syncthreads();                                       // P(3)
    stamp();                                               // 7
    if (last) {
      // ---- op 4 of the last block: logits = heads(norm_f(x2))
      u32x4 lnw[NCH], lnbb[NCH];
#pragma unroll
      for (int c2 = 0; c2 < NCH; ++c2) { lnw[c2] = ld16g(Lr.lnn_w + (c2 * 64 + lane) * 8); lnbb[c2] = ld16g(Lr.lnn_b + (c2 * 64 + lane) * 8); }
      sweep_granules<NCH>(zn_rsrc(a.g_x2), goff, tag, g, a.tmo, lane, SweepWho{(4u << 8) | (unsigned)li, a.diag});
      chain_layernorm_row<NCH>(g, lnw, lnbb, a.eps);
#pragma unroll
      for (int c2 = 0; c2 < NCH; ++c2) *(u32x4*)&s_act[myr * D + (c2 * 64 + lane) * 8] = g[c2];
      __syncthreads();                                     // B(4)
      __syncthreads();                                     // A(4)
      if (epi && ij < n_hd) {                              // EPI_F32 (gemv_epilogue): bf16-valued fp32 logits
        constexpr int par = 0;                             // op 4
        const int u = s_hd + ij;
        a.heads_out[(size_t)ir * a.heads_rows + 2 * u] = bfround(s_res[par][ij][0][ir]);
        if (2 * u + 1 < a.heads_rows) a.heads_out[(size_t)ir * a.heads_rows + 2 * u + 1] = bfround(s_res[par][ij][1][ir]);
      }
      __syncthreads();                                     // P(4)
    }
    pace.start();                                          // the wait for the next block's x1 starts here
  }
  if (epi && bc == 0 && lane == 0) st_sc1_u32(a.epoch, tag0 + (unsigned)a.n_layer);   // every workgroup read the epoch before its first publish, which this one has seen
}
