// Persistent post-attention chain of one transformer block at batch 1 (R = 2 rows, CFG pair), one launch per layer:
//
//   op 0  y1 = out_proj(a)                                   (_torch.py:419)
//   op 1  x  = x + out_proj(y1)                              (_torch.py:420, 326)
//   op 2  m  = y * silu(gate), (y, gate) = fc1(LayerNorm2(x))  (_torch.py:327, 473-474)
//   op 3  x  = x + fc2(m)
//   op 4  q | k | v = in_proj(LayerNorm1'(x)) of the NEXT block, RoPE, KV append   (_torch.py:399-408, 105-106)
//         (last block: logits = heads(norm_f(x)), model.py:139-141, instead)
//
// As five launches these cost ~3.3 us of fixed time each (boundary + ramp until the first weights arrive) on top of
// their streaming time; here one workgroup per CU stays resident for the whole chain: its 4 compute waves walk a STATIC
// list of weight tiles (a tile = two weight rows x 512*NCH columns = one gemv_kernel work unit, same arithmetic order,
// so the results are bit-identical to the launches path) with up to ZN_CH_NBUF tiles requested ahead in registers — requests run
// across the op boundaries, so the HBM stream no longer drains at every dependency —, and two communication waves, one per
// activation row (they issue no weight loads, so their own waits cover only the hand-off traffic) finish each op: epilogue of the
// workgroup's rows, published as 8-byte {tag, two bf16} granules (one write-through store each: the data is the flag, no
// drain, no counter: cdna_hip_programming.md Guideline 16 R2); the consumer sweeps the op's whole output vector with sc1
// loads until every tag carries this launch's epoch (bounded), applies LayerNorm where the next op wants it and puts the
// vector into LDS for the compute waves; fc2's input (32 KB) is swept by the compute waves themselves, a K quarter each.
// The first version arrived on sharded counters and polled them: 5-6 us per hand-off (publish 0.5-1.7 behind the CU's own
// prefetch queue, 2.6-4.8 until every arrival was seen, gather 0.4-1.5); DESIGN.md section 4 has both timelines.
#pragma once
#include <utility>
#include "zn_decode_kernels.h"

#ifndef ZN_CH_DEFER_MASK
#define ZN_CH_DEFER_MASK 0xF                               // bit k: weight requests for LATER ops raised during op k wait until op k's results are published
// (measured, decode step at 400 tokens: 0x0 1.103 ms, 0x3 1.071, 0xB 1.048, 0xF 1.027: a request queued in front of a publish or a
// sweep delays the hand-off by its whole service time, and the stream it feeds is not the bottleneck at that moment)
#endif
#define ZN_CH_CWAVES 4                                     // compute waves (a multiple of 4); the next two waves = communication waves
#ifndef ZN_CH_NBUF
#define ZN_CH_NBUF 3                                       // weight tiles requested ahead per compute wave (register buffers)
#endif
#define ZN_CH_THREADS ((ZN_CH_CWAVES + 2) * 64)              // + one communication wave per activation row
#ifndef ZN_CH_SWEEP_DELAY
#define ZN_CH_SWEEP_DELAY 40                               // s_sleep units (64 cycles) between an op's publish and the first sweep pass for its output (zn_step_kernel.h: ZN_SK_SWEEP_DELAY)
#endif
#ifndef ZN_SWEEP_BACKOFF
#define ZN_SWEEP_BACKOFF 0                                 // s_sleep units (64 cycles) between a failed sweep pass and the next
#endif
// Timing-only builds (tools/build_variants.py; outputs WRONG, never shipped): -DZN_TIMING_STREAM_ONLY makes every hand-off wait succeed on its
// first pass whatever it read (the weight stream and the arithmetic alone: what a block costs without its dependency chain);
// -DZN_TIMING_HANDOFFS_ONLY replaces the whole-step kernel's weight tiles by zeros without loading them (the dependency chain alone).
#ifdef ZN_TIMING_STREAM_ONLY
#define ZN_TIMING_PASS(bad) (bad) = 0ull
#else
#define ZN_TIMING_PASS(bad) (void)0
#endif
#define ZN_CH_TIMEOUT_TICKS 2000000ull                     // 20 ms of s_memrealtime (100 MHz) ...
#define ZN_CH_TIMEOUT_PASSES 4096u                         // ... AND this many sweep passes (~1 us each when the wave runs): see sweep_granules

struct ChainArgs {
  const bf16_t *W_out, *W_fc1, *W_fc2, *W_in;              // W_in = next block's in_proj (NULL: the chain ends after fc2)
  const bf16_t *ln2_w, *ln2_b, *lnn_w, *lnn_b;             // this block's norm2, next block's norm
  float eps;
  int F, nqkv;                                             // d_ff (= 4 d_model), rows of the next in_proj (or of the heads matrix)
  float* heads_out;                                        // != NULL: op 4 = final LayerNorm + fused heads of the LAST block, fp32 logits [2][nqkv] (W_in = heads, lnn_* = norm_f)
  const bf16_t* a;                                         // [2][d] attention output
  // No handed-off address is written twice inside a launch, nor read before it is written there (the residual stream enters
  // through one buffer and leaves through another): every sweep then either sees this launch's tag or an older launch's.
  const bf16_t* xin;                                       // [2][d] residual stream entering the block (read only)
  bf16_t* xout;                                            // [2][d] residual stream leaving the block (op 3); != xin
  // granule buffers {tag << 32 | two bf16}: [2][len / 2] each, written once and then swept inside a launch
  unsigned long long *g_y1, *g_x1, *g_x2;                  // len = d
  unsigned long long* g_m;                                 // len = F
  bf16_t* q_out; bf16_t* kv; const float* rope; const int* lengths;   // op 4 epilogue (next block's cache)
  int max_len, hd, n_heads, n_heads_kv, rope_positions;
  unsigned* epoch;                                         // this launch's tag (> 0); workgroup 0 leaves epoch + 1 for the next launch
  int* tmo;                                                // sticky timeout word (GenState.pad[0])
  unsigned long long* stamps;                              // optional [32] (diagnostic builds of the timeline: workgroup 0's communication wave)
  // ---- STACK instantiation: every block of the decode step in ONE launch (attention as an op of the chain)
  const struct StackLayer* layers;                         // [n_layer] device table
  int n_layer, stamp_layer;
  unsigned long long *g_qkv, *g_a;                         // granule vectors [2][nqkv / 2] (q | k | v of the next block) and [2][d / 2] (attention output)
  const bf16_t* q0;                                        // block 0's q [2][n_heads * hd] (from the in_proj launch before)
  float scale;                                             // 1 / sqrt(hd)
  int heads_rows;                                          // rows of the fused heads matrix (last block's op 4)
  bf16_t* trace;                                           // diagnostic [n_layer][8][2][d]: slot 0 x after the block, 1 attention output, 2 q
  unsigned* diag;                                          // [8] words describing the first hand-off wait that timed out (sweep_granules)
  unsigned dbg_pause;                                      // test hook (step_kernel): every wave stops for this many 10 ns ticks in block 2, as a paused device would
  // ---- whole-step kernel, key-block attention role (contexts beyond one 512-key block): the first `natt` workgroups of the grid are
  // attention workgroups, one per (row, kv head, 512-key block); their hand-offs among themselves:
  int natt;                                                // attention workgroups at the front of the grid (legacy role: rows * kv heads * hd / 32)
  unsigned long long* g_bmax;                              // granules {tag, fp32 bits} [rows * kv heads][ZN_SK_KB_MAXNB][4]: per-block score maxima of the pair's 4 heads
  unsigned long long* g_part;                              // granules [rows * kv heads][ZN_SK_KB_MAXNB][ZN_SK_KB_PSZ]: per-block unnormalised P.V [4][128] and e sums [4]
  // ---- whole-step kernel, pre-block: LayerNorm + in_proj + RoPE + KV append of block 0 inside the launch (pre_W == NULL: a launch before it
  // has left q in q0 and the new row in block 0's cache)
  const bf16_t *pre_W, *pre_ln_w, *pre_ln_b;               // block 0's in_proj and norm
  bf16_t* pre_kv;                                          // block 0's KV cache
};
struct StackLayer {
  const bf16_t *W_out, *W_fc1, *W_fc2, *W_in;              // W_in = NEXT block's in_proj, or the heads matrix (last block)
  const bf16_t *ln2_w, *ln2_b, *lnn_w, *lnn_b;             // this block's norm2; next block's norm, or norm_f
  const bf16_t* kv;                                        // this block's KV cache (attention)
  bf16_t* kv_next;                                         // next block's KV cache (op 4 appends), NULL for the last block
};

template <int I, int N, class Fn> ZN_DEVINL void zn_static_for(Fn&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); zn_static_for<I + 1, N>(f); }
}
ZN_DEVINL u32x4 ld_sc1_16(__amdgpu_buffer_rsrc_t rs, int byte_off) { return __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16); }   // aux 16 = sc1
ZN_DEVINL u32x2 ld_sc1_8(__amdgpu_buffer_rsrc_t rs, int byte_off) { return __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, 16); }     // one granule
ZN_DEVINL __amdgpu_buffer_rsrc_t zn_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000); }
ZN_DEVINL void st_sc1_u32(void* p, unsigned v) { __hip_atomic_store((unsigned*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
ZN_DEVINL unsigned ld_sc1_u32(const void* p) { return __hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
ZN_DEVINL unsigned dpp_movu(unsigned v, int) { return v; }
template <int CTRL> ZN_DEVINL unsigned dpp_u(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }

ZN_DEVINL void st_granule(unsigned long long* g, unsigned tag, unsigned value) {      // ONE aligned 8-byte write-through store
  __hip_atomic_store(g, ((unsigned long long)tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// One wave sweeps the 4 * N granules it needs (two per 16-byte sc1 load: .x/.z values, .y/.w tags; byte offsets off[]) until every
// tag equals `tag`, re-reading all of them every pass; bounded.  data[i] = the eight bf16 of off[i].
// The bound is wall time AND work: a wait gives up only after 20 ms in which it has also made ZN_CH_TIMEOUT_PASSES passes.  The first
// timeout this diagnostic ever recorded (round 3, profiles/r03_handoff_timeout_record.txt) showed a wave that had made 14 passes in 20 ms
// - a pass takes about a microsecond - i.e. the wave itself had not been running for almost all of that time (the queue preempted or
// the device stalled; s_memrealtime keeps counting), and on resumption every wait on the chip found its 20 ms "expired" although no
// hand-off was lost.  A wall-clock bound alone turns such a pause into a failed generation; counting passes as well does not, and a
// hand-off that really never arrives is still reported after 20 ms of genuine spinning.
// A wait that gives up describes itself: the first wave to time out (tmo 0 -> 1) leaves, in diag[0..7], the caller's code
// (stage << 8 | block), workgroup, wave, the tag it waited for, the byte offset and the tag of its first stale granule, the lane that
// held it and the passes made; zn_all_stopped prints them.  Later waits see tmo != 0 and return at once (the results are void anyway).
struct SweepWho { unsigned code; unsigned* diag; };
// (the byte offsets come from a functor evaluated at every use, not from an array: in the larger whole-step instantiations an offset array
// survived as a dead 32-byte stack object and gave the kernel a private segment, which a persistent kernel must not have)
template <int N, class OffFn>
ZN_DEVINL bool sweep_granules_at(__amdgpu_buffer_rsrc_t rs, OffFn off, unsigned tag, u32x4 (&data)[N], int* tmo, int lane, SweepWho who, unsigned* passes_out = nullptr) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned np = 0;
  for (;;) {
    ++np;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int o = off(i);
      const u32x4 l0 = ld_sc1_16(rs, o), l1 = ld_sc1_16(rs, o + 16);
      ok &= (l0.y == tag) & (l0.w == tag) & (l1.y == tag) & (l1.w == tag);
      data[i] = u32x4{l0.x, l0.z, l1.x, l1.z};
    }
    unsigned long long bad = __builtin_amdgcn_ballot_w64(!ok);
    ZN_TIMING_PASS(bad);
    if (bad == 0ull) { if (passes_out) *passes_out = np; return true; }
    if constexpr (ZN_SWEEP_BACKOFF > 0) __builtin_amdgcn_s_sleep(ZN_SWEEP_BACKOFF);      // a failed pass: let the publishers' stores through before asking again
    if ((np >= ZN_CH_TIMEOUT_PASSES && __builtin_amdgcn_s_memrealtime() - t0 > ZN_CH_TIMEOUT_TICKS) || __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
      if (lane == (int)__builtin_ctzll(bad) && atomicAdd(tmo, 1) == 0 && who.diag) {
        unsigned badoff = 0xffffffffu, badtag = 0;            // the first stale granule of this lane, read once more
#pragma unroll
        for (int i = N - 1; i >= 0; --i) {
          const int o = off(i);
          const u32x4 l0 = ld_sc1_16(rs, o), l1 = ld_sc1_16(rs, o + 16);
          if (l1.w != tag) { badoff = o + 24; badtag = l1.w; }
          if (l1.y != tag) { badoff = o + 16; badtag = l1.y; }
          if (l0.w != tag) { badoff = o + 8; badtag = l0.w; }
          if (l0.y != tag) { badoff = o; badtag = l0.y; }
        }
        who.diag[0] = who.code; who.diag[1] = blockIdx.x; who.diag[2] = threadIdx.x >> 6; who.diag[3] = tag;
        who.diag[4] = badoff; who.diag[5] = badtag; who.diag[6] = (unsigned)lane; who.diag[7] = np;
      }
      return false;
    }
  }
}
template <int N>
ZN_DEVINL bool sweep_granules(__amdgpu_buffer_rsrc_t rs, const int (&off)[N], unsigned tag, u32x4 (&data)[N], int* tmo, int lane, SweepWho who, unsigned* passes_out = nullptr) {
  return sweep_granules_at<N>(rs, [&](int i) { return off[i]; }, tag, data, tmo, lane, who, passes_out);
}

// One 32-byte piece per lane (four granules), DEPTH passes in flight: a new pass goes out every time the oldest returns, so that the wait ends
// within a fraction of a memory round trip of the granules becoming visible instead of within a whole one.  For the attention workgroups of
// the whole-step kernel only: their CUs stream no weights and they are few (polling from the 448 communication waves of the streaming
// workgroups this way loads the fabric their publishers' stores travel on: measured slower).
template <int DEPTH>
ZN_DEVINL bool sweep_granules_piped(__amdgpu_buffer_rsrc_t rs, int off, unsigned tag, u32x4& data, int* tmo, int lane, SweepWho who) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned np = 0;
  u32x4 l0[DEPTH], l1[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) { l0[d] = ld_sc1_16(rs, off); l1[d] = ld_sc1_16(rs, off + 16); }
  for (;;) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      ++np;
      const u32x4 a0 = l0[d], a1 = l1[d];                    // (waits for this pass only: the later ones stay in flight)
      const bool ok = (a0.y == tag) & (a0.w == tag) & (a1.y == tag) & (a1.w == tag);
      unsigned long long bad = __builtin_amdgcn_ballot_w64(!ok);
      ZN_TIMING_PASS(bad);
      if (bad == 0ull) { data = u32x4{a0.x, a0.z, a1.x, a1.z}; return true; }
      if ((np >= ZN_CH_TIMEOUT_PASSES && __builtin_amdgcn_s_memrealtime() - t0 > ZN_CH_TIMEOUT_TICKS) || __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        if (lane == (int)__builtin_ctzll(bad) && atomicAdd(tmo, 1) == 0 && who.diag) {
          const unsigned badoff = a0.y != tag ? off : a0.w != tag ? off + 8 : a1.y != tag ? off + 16 : off + 24;
          const unsigned badtag = a0.y != tag ? a0.y : a0.w != tag ? a0.w : a1.y != tag ? a1.y : a1.w;
          who.diag[0] = who.code; who.diag[1] = blockIdx.x; who.diag[2] = threadIdx.x >> 6; who.diag[3] = tag;
          who.diag[4] = badoff; who.diag[5] = badtag; who.diag[6] = (unsigned)lane; who.diag[7] = np;
        }
        data = u32x4{a0.x, a0.z, a1.x, a1.z};
        return false;
      }
      l0[d] = ld_sc1_16(rs, off); l1[d] = ld_sc1_16(rs, off + 16);
    }
  }
}

// The same bounded, self-describing wait for sweeps whose requests do not have sweep_granules' shape (the key-block attention role's
// block maxima and partials): pass() after every failed pass says whether to give up; report() is sweep_granules' diagnostic record.
struct SpinBound {
  unsigned long long t0; unsigned np;
  ZN_DEVINL void begin() { t0 = __builtin_amdgcn_s_memrealtime(); np = 0; }
  ZN_DEVINL bool give_up(int* tmo) const {
    return (np >= ZN_CH_TIMEOUT_PASSES && __builtin_amdgcn_s_memrealtime() - t0 > ZN_CH_TIMEOUT_TICKS) || __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  }
  ZN_DEVINL void report(int* tmo, SweepWho who, unsigned long long bad, int lane, unsigned tag, unsigned badoff, unsigned badtag) const {
    if (lane == (int)__builtin_ctzll(bad) && atomicAdd(tmo, 1) == 0 && who.diag) {
      who.diag[0] = who.code; who.diag[1] = blockIdx.x; who.diag[2] = threadIdx.x >> 6; who.diag[3] = tag;
      who.diag[4] = badoff; who.diag[5] = badtag; who.diag[6] = (unsigned)lane; who.diag[7] = np;
    }
  }
};

// nn.LayerNorm on ONE row held as gemv_kernel holds it (lane owns elements (c*64 + lane)*8 .. +8), through the helpers
// gemv_kernel's PRO_LN prologue uses (explicit roundings, rows independent of each other): identical statistics and outputs.
template <int NCH>
ZN_DEVINL void chain_layernorm_row(u32x4 (&x)[NCH], const u32x4 (&lng)[NCH], const u32x4 (&lnb)[NCH], float eps) {
  const float invK = 1.0f / (float)(NCH * 512);
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) s += ln_sum8(x[c]);
  const float mean = __fmul_rn(wave_sum(s), invK);
#pragma unroll
  for (int c = 0; c < NCH; ++c) ss = ln_sq8(x[c], mean, ss);
  const float rstd = ln_rstd(wave_sum(ss), invK, eps);
#pragma unroll
  for (int c = 0; c < NCH; ++c) x[c] = ln_norm8(x[c], mean, rstd, lng[c], lnb[c]);
}


// T_* = tiles per compute wave per op (upper bounds; a wave skips the tiles its workgroup does not have).  d_model =
// 512 * NCH, d_ff = 4 * d_model; every op's units divide evenly over the grid (host-checked).
template <int NCH, int T_OUT, int T_FC1, int T_FC2, int T_IN>
__global__ __launch_bounds__(ZN_CH_THREADS) void chain_kernel(ChainArgs a) {
  constexpr int R = 2, D = NCH * 512, CW = ZN_CH_CWAVES;
  constexpr int S1 = T_OUT, S2 = 2 * T_OUT, S3 = S2 + T_FC1, S4 = S3 + T_FC2, NS = S4 + T_IN;     // slot ranges per op
  constexpr int NOPS = T_IN > 0 ? 5 : 4;
  constexpr int MASK = ZN_CH_DEFER_MASK;
  // The second out_proj (op 1) reuses the first one's weight tiles in place: its slots raise no request.  Requests are numbered
  // l = 0 .. NL-1 in slot order without them (buffer l % 3); a buffer takes request l + 3 after its last use.
  constexpr int NB = ZN_CH_NBUF;
  constexpr int NL = NS - T_OUT, INIT = T_OUT + 1 < NB ? T_OUT + 1 : NB;    // requests raised at kernel start: op 0's tiles + one more
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = blockIdx.x, G = gridDim.x;
  const int F = a.F;
  const unsigned tag = *a.epoch;
  // units (weight-row pairs) per workgroup and op
  const int units_in = (a.nqkv + 1) / 2;                  // row pairs of op 4 (the heads matrix has an odd row count: its last pair is half)
  const int ppw_out = (D / 2) / G, ppw_fc1 = F / G, ppw_fc2 = (D / 2) / G, ppw_in = T_IN > 0 ? (units_in + G - 1) / G : 0;
  __shared__ __attribute__((aligned(16))) bf16_t s_act[R * D];              // the current op's input vector (ops 0, 1, 2, 4)
  __shared__ float s_res[2][64][2][R];                                      // per-unit results of even / odd ops (fc2: [unit * 4 + quarter])

  auto op_of = [](int s) constexpr { return s < S1 ? 0 : s < S2 ? 1 : s < S3 ? 2 : s < S4 ? 3 : 4; };
  auto first_of = [](int op) constexpr { return op == 0 ? 0 : op == 1 ? S1 : op == 2 ? S2 : op == 3 ? S3 : S4; };

  if (wave < CW) {
    // ------------------------------------------------------------------------------------ compute waves
    struct WT { u32x4 a[NCH], b[NCH]; };
    WT bufs[NB];                                           // every index below is a compile-time constant: the buffers live in registers
    static_assert(NB >= 2 && CW % 4 == 0, "rotating tile buffers; fc2 splits K over groups of four waves");
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
      ok = j < ppw && (op < 4 || c * ppw + j < units_in);
      const int u = c * ppw + (ok ? j : 0);
      ridx = j;
      if (op <= 1) { pa = a.W_out + (size_t)(2 * u) * D + lane * 8; pb = pa + D; }
      else if (op == 2) { pa = a.W_fc1 + (size_t)u * D + lane * 8; pb = pa + (size_t)F * D; }
      else { pa = a.W_in + (size_t)(2 * u) * D + lane * 8; pb = (2 * u + 1 < a.nqkv) ? pa + D : pa; }   // half pair: row B repeats row A, result dropped
    };
    auto load = [&](int s, WT& w) {
      bool ok; const bf16_t *pa, *pb; int ridx;
      tile(s, ok, pa, pb, ridx);
      if (ok) {                                           // wave-uniform
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
    // request raised by the last use of slot s's buffer (-1: none), and whether it waits for the op's publish
    auto raised_by = [](int s) constexpr {
      const int op = s < S1 ? 0 : s < S2 ? 1 : s < S3 ? 2 : s < S4 ? 3 : 4;
      if (op == 0) return -1;                              // the tile stays for op 1
      const int l = (s - T_OUT) + NB;                      // op 1 slot s shares the buffer of request s - T_OUT; later slots hold request s - T_OUT
      return l < NL ? l : -1;
    };
#ifdef ZN_CH_EARLY_O
    zn_static_for<0, T_OUT>([&](auto LC) { load_req(LC); });      // op 0's own tiles are on the critical path: requested at once
    __syncthreads();                                      // S: the communication wave's own requests are in the CU's queue before the prefetch
    zn_static_for<T_OUT, INIT>([&](auto LC) { load_req(LC); });
#else
    __syncthreads();                                      // S: the communication wave's own requests are in the CU's queue first
    zn_static_for<0, INIT>([&](auto LC) { load_req(LC); });
#endif
    zn_static_for<0, NS>([&](auto SC) {
      constexpr int s = decltype(SC)::value;
      constexpr int op = op_of(s);
      if constexpr (s == first_of(op)) {
        if constexpr (op > 0) {
          __syncthreads();                                // A(op-1): this workgroup's results of the previous op are in LDS
          __syncthreads();                                // P(op-1): ... and published; the requests held back for that go out now
          if constexpr (op == 1) zn_static_for<INIT, (NL < NB ? NL : NB)>([&](auto LC) { load_req(LC); });
          if constexpr (((MASK >> (op - 1)) & 1) != 0) {
            zn_static_for<first_of(op - 1), first_of(op)>([&](auto QC) {
              constexpr int q = decltype(QC)::value, l = raised_by(q);
              if constexpr (l >= 0) { if constexpr (op_of(slot_of_load(l >= 0 ? l : 0)) != op - 1) load_req(std::integral_constant<int, (l >= 0 ? l : 0)>{}); }
            });
          }
        }
        if constexpr (op == 3) {
          // fc2's input m [2][4 d]: this wave's K quarter straight from the granules (no LDS, no barrier)
          const int qt = wave & 3;
          int off[NCH * R];
          u32x4 dat[NCH * R];
#pragma unroll
          for (int c2 = 0; c2 < NCH; ++c2)
#pragma unroll
            for (int r = 0; r < R; ++r) off[c2 * R + r] = (r * (2 * D) + qt * (D / 2) + (c2 * 64 + lane) * 4) * 8;
          sweep_granules<NCH * R>(zn_rsrc(a.g_m), off, tag, dat, a.tmo, lane, SweepWho{(3u << 8) | (unsigned)a.stamp_layer, a.diag});
#pragma unroll
          for (int c2 = 0; c2 < NCH; ++c2)
#pragma unroll
            for (int r = 0; r < R; ++r) xr[c2][r] = dat[c2 * R + r];
        } else {
          __syncthreads();                                // B(op): the op's input vector is in LDS
#pragma unroll
          for (int c2 = 0; c2 < NCH; ++c2)
#pragma unroll
            for (int r = 0; r < R; ++r) xr[c2][r] = *(const u32x4*)&s_act[r * D + (c2 * 64 + lane) * 8];
        }
      }
      constexpr int lb = (op == 0 ? s : s - T_OUT) % NB;   // buffer holding this slot's weights
      process(s, bufs[lb]);
      constexpr int l = raised_by(s);
      if constexpr (l >= 0) {
        if constexpr (((MASK >> op) & 1) == 0 || op_of(slot_of_load(l >= 0 ? l : 0)) == op) load_req(std::integral_constant<int, (l >= 0 ? l : 0)>{});
      }
    });
    __syncthreads();                                      // A(last op)
    return;
  }

  // -------------------------------------------------------------------------------------- communication waves
  // Wave CW + r gathers (sweeps), normalises and stages row r of every hand-off; wave CW also runs the epilogues.  (One wave
  // for both rows spent 1.2 us per LayerNorm, VALU-bound on a single SIMD, on the critical path of ops 2 and 4.)
  const int myr = wave - CW;
  const bool epi = myr == 0;
  // operands that do not depend on this launch's hand-offs are requested up front
  u32x4 g[NCH];                                            // this wave's row in gemv_kernel's lane layout
  u32x4 l2w[NCH], l2b[NCH], lnw[NCH], lnbb[NCH];
#pragma unroll
  for (int c2 = 0; c2 < NCH; ++c2) g[c2] = ld16(a.a + (size_t)myr * D + (c2 * 64 + lane) * 8);
#pragma unroll
  for (int c2 = 0; c2 < NCH; ++c2) { l2w[c2] = ld16(a.ln2_w + (c2 * 64 + lane) * 8); l2b[c2] = ld16(a.ln2_b + (c2 * 64 + lane) * 8); }
  if constexpr (T_IN > 0) {
#pragma unroll
    for (int c2 = 0; c2 < NCH; ++c2) { lnw[c2] = ld16(a.lnn_w + (c2 * 64 + lane) * 8); lnbb[c2] = ld16(a.lnn_b + (c2 * 64 + lane) * 8); }
  }
  // items of the row-pair ops (0, 1, 3, 4): lane = j * R + r
  const int ij = lane >> 1, ir = lane & 1;
  const bool it_out = epi && ij < ppw_out;                 // ops 0, 1, 3 (same units: d/2 pairs)
  const int u_out = c * ppw_out + (it_out ? ij : 0);
  unsigned resid = 0;
  if (it_out) resid = *(const unsigned*)(a.xin + (size_t)ir * D + 2 * u_out);
  const bool it_in = epi && T_IN > 0 && ij < ppw_in && c * ppw_in + ij < units_in;
  const int u_in = c * ppw_in + (it_in ? ij : 0);
  int pos = 0; float cs = 1.f, sn = 0.f;
  if constexpr (T_IN > 0) {
    if (it_in && !a.heads_out) {
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
  __syncthreads();                                         // S
#pragma unroll
  for (int c2 = 0; c2 < NCH; ++c2) *(u32x4*)&s_act[myr * D + (c2 * 64 + lane) * 8] = g[c2];
  __syncthreads();                                         // B(0)
  int nst = 0;
  auto stamp = [&]() { if (a.stamps && epi && c == 0 && lane == 0) a.stamps[nst] = __builtin_amdgcn_s_memrealtime(); ++nst; };
  stamp();

  int goff[NCH];                                           // byte offsets of this lane's granules of row myr in a [2][d / 2] granule vector
#pragma unroll
  for (int c2 = 0; c2 < NCH; ++c2) goff[c2] = (myr * (D / 2) + (c2 * 64 + lane) * 4) * 8;
  unsigned x1own = 0;
  zn_static_for<0, NOPS>([&](auto OC) {
    constexpr int op = decltype(OC)::value;
    constexpr int par = op & 1;
    __syncthreads();                                       // A(op): every compute wave's results are in LDS
    stamp();
    // ---- epilogue of this workgroup's units, published as granules
    if constexpr (op == 0) {                               // EPI_STORE
      if (it_out) st_granule(a.g_y1 + (size_t)ir * (D / 2) + u_out, tag, pack2(s_res[par][ij][0][ir], s_res[par][ij][1][ir]));
    } else if constexpr (op == 1) {                        // EPI_RESID
      if (it_out) {
        x1own = pack2(lo_f(resid) + bfround(s_res[par][ij][0][ir]), hi_f(resid) + bfround(s_res[par][ij][1][ir]));
        st_granule(a.g_x1 + (size_t)ir * (D / 2) + u_out, tag, x1own);
      }
    } else if constexpr (op == 2) {                        // EPI_SILU: lane = r * ppw_fc1 + j, neighbours share a granule
      if (epi) {                                           // wave-uniform
        const int r2 = lane / ppw_fc1, j2 = lane % ppw_fc1;
        const bool on = r2 < R;
        const int jj = on ? j2 : 0, rr = on ? r2 : 0;
        const float y = bfround(s_res[par][jj][0][rr]), gt = bfround(s_res[par][jj][1][rr]);
        const float sg = bfround(gt / (1.0f + expf(-gt)));
        const unsigned mine = (unsigned)f2bf(y * sg);
        const unsigned nb = (unsigned)__shfl_down((int)mine, 1);
        if (on && (j2 & 1) == 0) st_granule(a.g_m + (size_t)r2 * (F / 2) + ((c * ppw_fc1 + j2) >> 1), tag, mine | (nb << 16));
      }
    } else if constexpr (op == 3) {                        // EPI_RESID over the four K quarters, in gemv_kernel's order
      if (it_out) {
        const float vA = ((s_res[par][ij * 4 + 0][0][ir] + s_res[par][ij * 4 + 1][0][ir]) + s_res[par][ij * 4 + 2][0][ir]) + s_res[par][ij * 4 + 3][0][ir];
        const float vB = ((s_res[par][ij * 4 + 0][1][ir] + s_res[par][ij * 4 + 1][1][ir]) + s_res[par][ij * 4 + 2][1][ir]) + s_res[par][ij * 4 + 3][1][ir];
        const unsigned o = pack2(lo_f(x1own) + bfround(vA), hi_f(x1own) + bfround(vB));
        if constexpr (NOPS == 5) st_granule(a.g_x2 + (size_t)ir * (D / 2) + u_out, tag, o);
        *(unsigned*)(a.xout + (size_t)ir * D + 2 * u_out) = o;              // the plain copy later launches read
      }
    } else {                                               // EPI_ROPE_KV of the next block (read by the next launch), or the logits
      if (it_in && a.heads_out) {                          // EPI_F32 (gemv_epilogue): bf16-valued fp32
        a.heads_out[(size_t)ir * a.nqkv + 2 * u_in] = bfround(s_res[par][ij][0][ir]);
        if (2 * u_in + 1 < a.nqkv) a.heads_out[(size_t)ir * a.nqkv + 2 * u_in + 1] = bfround(s_res[par][ij][1][ir]);
      } else if (it_in) {
        GemvArgs ga{};
        ga.hd = a.hd; ga.n_heads = a.n_heads; ga.n_heads_kv = a.n_heads_kv; ga.q_out = a.q_out; ga.kv = a.kv; ga.max_len = a.max_len;
        gemv_epilogue<EPI_ROPE_KV>(ga, ir, 2 * u_in, 2 * u_in + 1, true, u_in, s_res[par][ij][0][ir], s_res[par][ij][1][ir], 0u, cs, sn, pos);
      }
    }
    if constexpr (op + 1 < NOPS) {
      __syncthreads();                                     // P(op): published; the compute waves may queue weight requests again
      stamp();
      if constexpr (op != 2) {                             // (fc2's input is swept by the compute waves)
        // (requesting the first pass ahead of the compute waves' held-back requests was measured slower, 1.089 vs 1.023 ms per step: it
        // comes back before the slowest publishers' stores are visible, and the second pass then queues behind those requests)
        if constexpr (ZN_CH_SWEEP_DELAY > 0) __builtin_amdgcn_s_sleep(ZN_CH_SWEEP_DELAY);
        sweep_granules<NCH>(zn_rsrc(op == 0 ? a.g_y1 : op == 1 ? a.g_x1 : a.g_x2), goff, tag, g, a.tmo, lane,
                            SweepWho{((op == 0 ? 1u : op == 1 ? 2u : 4u) << 8) | (unsigned)a.stamp_layer, a.diag},
                            (a.stamps && epi && c == 0 && lane == 0) ? (unsigned*)(a.stamps + 24 + op) : nullptr);
        stamp();
        if constexpr (op == 1 || op == 3) chain_layernorm_row<NCH>(g, op == 1 ? l2w : lnw, op == 1 ? l2b : lnbb, a.eps);
#pragma unroll
        for (int c2 = 0; c2 < NCH; ++c2) *(u32x4*)&s_act[myr * D + (c2 * 64 + lane) * 8] = g[c2];
        __syncthreads();                                   // B(op + 1)
      } else stamp();
      stamp();
    }
  });
  stamp();
  if (epi && c == 0 && lane == 0) *a.epoch = tag + 1;      // every workgroup read the epoch before its first publish, which this one has seen
}
