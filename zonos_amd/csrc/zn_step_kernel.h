// Whole decode step of the transformer stack at batch 1 (R = 2 rows) in ONE persistent launch (round 3; attention role rebuilt in round 4).
//
// 256 workgroups of 8 waves, one per CU, in two fixed roles:
//
//  * 8 * NBK ATTENTION workgroups, one per (row, kv head, 512-key block) of the launch's context bound (NBK = 1 .. 12: contexts up to 6144
//    keys), carry NO weight tiles.  Their register budget holds K and full-width V of the NEXT block's cache, requested as soon as the
//    block's result is published - a whole block (~20 us) before q arrives -, so the attention of a block is hand-offs plus ~3 us of
//    arithmetic on the matrix cores (scores and P.V), and nothing in it waits for memory.  (step_attention_kb_role; _torch.py:376-420)
//  * the other 248 - 160 STREAMING workgroups split every weight matrix of the block (out_proj, fc1, fc2, next in_proj / heads) with
//    chain_kernel's static tile schedule: 4 compute waves, 2 communication waves (one per activation row), 2 idle waves (the launch's shape
//    is the attention's).  While the attention runs they have nothing to wait for but weights: each compute wave streams the first
//    ZN_SK_PARK tiles of the block through its register buffers into LDS ("parked" tiles, 128 KB per CU) and leaves ZN_SK_NBUF more
//    in flight in registers - 7 of its ~17 tiles (~50 MB chip-wide, ~40 % of the block's weights) are on chip before op 0's input
//    exists.  (_torch.py:307-328)
//
// Hand-offs are chain_kernel's tagged granules (tag = epoch + block).  Reuse of a granule buffer block after block is safe
// because every streaming workgroup publishes in every op 0-3 and every sweep covers the whole vector: a workgroup can publish
// stage s of block b + 1 only after EVERY streaming workgroup has published stage s' > s of block b, i.e. after each of them has
// finished sweeping stage s of block b; the attention workgroups' stages (q|k|v in, attention output out) sit inside that
// chain: an attention workgroup publishes g_a(b + 1) only after its own sweep of g_qkv(b), and no workgroup writes g_qkv(b + 1)
// before it has swept all of g_a(b + 1); the attention workgroups' own exchanges (block maxima, partials) sit between their sweep of
// g_qkv(b) and the publication of g_a(b + 1): a pair's maxima and partials of block b have all been swept before its combiner publishes,
// i.e. before q of the next block exists.  A consumer therefore never sees a tag newer than the one it waits for, and the sweeps
// compare for equality.  Every wait is bounded and describes itself when it gives up (sweep_granules, SpinBound, SweepWho).
//
// Results (codes, logits) are bit-identical to the per-block path and to the launches path at every context length.
#pragma once
#include "zn_chain_kernel.h"
#include "zn_step_sched.h"

#define ZN_SK_THREADS 512
#define ZN_SK_CW 4                                          // compute waves of a streaming workgroup (waves 4, 5: communication; 6, 7: idle)
#ifndef ZN_SK_NBUF
#define ZN_SK_NBUF 3                                        // register tile buffers per compute wave
#endif
#ifndef ZN_SK_PARK
#define ZN_SK_PARK 4                                        // tiles per compute wave parked in LDS during the attention (>= T_OUT); 3: 0.8743 vs 0.8694 ms per step
#endif
#ifndef ZN_SK_HELP
#define ZN_SK_HELP 0                                        // tiles per compute wave that a helper wave (6, 7) holds in ITS registers during the attention and
#endif                                                      // drops into the LDS slots op 1 has left (0: helper waves only run the barriers).  Measured, 400 tokens:
                                                            // 2 at block start 0.8773 ms per step vs 0.8771 without; 2 after P(0) 0.8871 (y1 hand-off 2.3 -> 3.4 us)
#ifndef ZN_SK_HELP_AT
#define ZN_SK_HELP_AT 0                                     // where the helper requests them: 0 = block start, 1 = after B(0) (attention output in), 2 = after P(0)
#endif
#ifndef ZN_SK_EARLY
#define ZN_SK_EARLY 0                                       // requests for a LATER op's tiles that a wave may raise per op before that op's results are published (the
#endif                                                      // rest wait for the publish: ZN_CH_DEFER_MASK).  Measured, 400 tokens: 0 0.8694 ms per step, 1 0.8907, 2 0.9398
#ifndef ZN_SK_MSWEEP_DELAY
#define ZN_SK_MSWEEP_DELAY 0                                // the same for fc2's input, swept by the compute waves behind fc2's held-back tile requests (24 / 48: 0.837 vs 0.8335;
                                                            // that sweep's first pass issued AHEAD of those requests, 1 us after the publish: 0.868 vs 0.829)
#endif
#ifndef ZN_SK_SWEEP_DELAY
#define ZN_SK_SWEEP_DELAY 32                                // s_sleep units (64 cycles) between an op's publish and the first sweep pass for its output
#endif                                                      // (two sweep passes in flight per wave, a new one every half round trip: 0.891 vs 0.869 ms per step - more polling loads the fabric).
                                                            // With the key-block attention role, ms per step at contexts 300 / 600 / 1600 / 3800 (profiles/r04_sweep_delay.txt): 20: .852 .879
                                                            // .902 .976; 24: .844 .866 .886 .973; 28: .834 .853 .868 .975; 32: .835 .852 .867 .980; 40: .834 .856 .875 .990; 48: .834 .862 .881 1.001
#define ZN_SK_DYN_LDS (ZN_SK_CW * ZN_SK_PARK * 8192)        // parked tiles (streaming role) / StepKbLds (attention role)
#ifndef ZN_SK_APOLL
#define ZN_SK_APOLL 1                                       // sweep passes an attention workgroup keeps in flight while it waits for a hand-off.  Its CU streams nothing, and
#endif                                                      // still: 2 / 3 passes in flight cost 0.914 / 0.940 ms per step at 600 keys against 0.861 with one (polling loads the fabric)
#ifndef ZN_SK_XDELAY
#define ZN_SK_XDELAY 16                                     // s_sleep units (64 cycles) between an attention workgroup's publish (block maxima, partials) and its first sweep pass for the
#endif                                                      // others', from ZN_SK_XDELAY_NB blocks on: 0 / 16 / 32 / 48 units: 0.857 / 0.870 / 0.884 / 0.900 ms per step at 2 blocks, 0.878 / 0.896 /
#ifndef ZN_SK_XDELAY_NB                                     // 0.906 / 0.919 at 4, 0.939 / 0.913 / 0.929 / 0.944 at 6, 1.030 / 1.004 / 1.018 / 1.034 at 8, 1.084 / 1.052 / 1.069 / 1.085 at 10
#define ZN_SK_XDELAY_NB 5
#endif
#ifndef ZN_SK_PACE_MARGIN
#define ZN_SK_PACE_MARGIN 0                                 // > 0: sleep until this many 10 ns ticks before the previous wait's length instead of a fraction of it
#endif
#ifndef ZN_SK_PACE_SHIFT
#define ZN_SK_PACE_SHIFT 2                                  // a waiting wave sleeps through the first (1 - 2^-SHIFT) of the wait it measured one block earlier
#endif

// A wave that waits for a hand-off far in the future (the attention workgroups for q|k|v, the streaming ones for the attention
// output) sleeps through most of the wait it measured one block earlier instead of polling through it (polls sit in the CU's
// memory queue in front of its own prefetch and load the fabric; megakernel price list "polling-cost").
#ifndef ZN_SK_PACE_CAP
#define ZN_SK_PACE_CAP 8192u            // 82 us in the 100 MHz ticks of s_memrealtime
#endif
#define ZN_SK_LONG_WAIT 20000u          // 0.2 ms in the 100 MHz ticks of s_memrealtime: six block times
struct StepPacer {
  unsigned long long t_ref; unsigned prev;
  unsigned worst = 0, n_long = 0;         // observation only (zn_get_counters): the longest wait of this wave in the launch, waits longer than ZN_SK_LONG_WAIT
  ZN_DEVINL void start() { t_ref = __builtin_amdgcn_s_memrealtime(); }
  ZN_DEVINL void sleep() const {
#ifdef ZN_TIMING_STREAM_ONLY
    return;
#endif
    const unsigned long long until = t_ref + (ZN_SK_PACE_MARGIN > 0 ? (prev > ZN_SK_PACE_MARGIN ? prev - ZN_SK_PACE_MARGIN : 0u) : (prev - (prev >> ZN_SK_PACE_SHIFT)));
    while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(8);
  }
  // What is learnt is capped at a few block times: a wait that contained a pause of the device (queue preemption; seen twice in some
  // thousand generations, 20 - 30 ms each) must not become the next block's sleep - the waves that slept through 3/4 of such a "wait"
  // kept everybody else polling past the hand-off timeout (profiles/r03_handoff_timeout_record.txt, second record).
  ZN_DEVINL void done() {
    const unsigned long long d = __builtin_amdgcn_s_memrealtime() - t_ref;
    prev = d < ZN_SK_PACE_CAP ? (unsigned)d : ZN_SK_PACE_CAP;
    const unsigned du = d < 0xffffffffull ? (unsigned)d : 0xffffffffu;
    worst = du > worst ? du : worst;
    n_long += du > ZN_SK_LONG_WAIT ? 1u : 0u;
  }
  // A pause of the device (section "Pauses of the device") shows up here as one wait of its length in every waiting wave: the launch leaves the
  // longest wait any wave measured in diag[8] and the number of waits beyond 0.2 ms in diag[9] - observed, never provoked.
  ZN_DEVINL void report(unsigned* diag, int lane) const {
    if (lane == 0 && diag) { if (worst > ZN_SK_LONG_WAIT / 2) atomicMax(diag + 8, worst); if (n_long) atomicAdd(diag + 9, n_long); }
  }
};
// Test hook (ChainArgs::dbg_pause, zn_debug_tune(14, 11)): what a pause of the device looks like from inside - every wave stops at about the
// same wall-clock time, the attention workgroups inside their measured wait for q | k | v, the streaming workgroups outside theirs.
ZN_DEVINL void step_debug_pause(unsigned ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(127);
}

// ------------------------------------------------------------------------------------------------ attention role: one workgroup per KEY BLOCK
// The first natt = 8 * NBK workgroups of the grid are one per (row, kv head, 512-key block) and carry NO weight tiles.  A workgroup holds K of
// its block in the MFMA B-fragment layout (64 VGPRs) and V at full width in AttnV's layout (64 VGPRs), both requested the moment the previous
// block's result is published - a whole block (~20 us) before q arrives, so nothing in the attention waits for memory.  Per block: q (and, in
// the newest block's workgroup, this step's key and value row) from the hand-off granules; the block's 512 x 4 scores on the matrix cores into
// LDS; block maxima exchanged with the other blocks of the (row, kv head) pair as granules (the running max m_j of the reference's recurrence is
// the max over blocks 0 .. j); pass 2 = attn_block_probs / attn_block_pv (zn_decode_kernels.h: the launches' arithmetic, bit for bit); the
// unnormalised P.V and e sums leave as granules; the workgroup of block 0 sweeps the partials of all blocks, replays acc = acc * f_j + pv_j in
// block order (attn_block_kernel's combine), normalises and publishes the attention output.  One block (contexts up to 512 keys): no exchange,
// no partials - the workgroup normalises and publishes itself, and 248 workgroups stream.  Workgroups of blocks past the context (the grid is
// sized for the launch's upper bound) leave at once.  (Rounds 2-3: one workgroup per 32-wide value slice over one or two blocks, P.V on the
// VALU, limited to 1024 keys; the same time at one block, 2.7 us per block slower at two.)  (_torch.py:413-417)
#define ZN_SK_KB_MAXNB 12                                   // key blocks covered: 6144 keys
#define ZN_SK_KB_PSZ 520                                    // granules per (pair, block): P.V [4][128], e sums [4] (as two 16-byte pairs), padding to 64 B
struct StepKbLds {
  float sc[4][512];                                         // scores [head][key of the block]
  float bm[8][4];                                           // per (wave, head) maxima of the block
  float bmall[ZN_SK_KB_MAXNB][4];                           // maxima of every block of the pair
  float accw[8][4][128];                                    // per-wave partial P.V [head][dim]
  float l[8][4];
  __attribute__((aligned(16))) bf16_t p[4][512];            // P = bf16(e) [head][key of the block]
  __attribute__((aligned(16))) bf16_t q[4][128];            // the kv head's 4 query heads
  __attribute__((aligned(16))) bf16_t knew[128], vnew[128]; // newest key / value row of the kv head
  __attribute__((aligned(16))) bf16_t out[4][128];          // normalised result (one-block contexts)
};
static_assert(sizeof(StepKbLds) <= ZN_SK_DYN_LDS, "the key-block attention role's LDS fits the launch's dynamic LDS");

// The combine of a (row, kv head) pair's blocks j0 .. j0 + CH - 1 (attn_block_kernel's recurrence): sweeps the blocks' partials of one output
// pair (P.V of columns 2 dp, 2 dp + 1 of head g as two granules = one 16-byte load; the e sum of head g as one granule), DEPTH passes in flight,
// and folds them into (tot0, tot1, lt, mprev) in block order.  Every pass requests all CH entries, blocks past the context repeating the last
// one (an entry that kept its value from the pass before would be carried around the retry loop in a second set of registers).
template <int CH, int DEPTH>
ZN_DEVINL void kb_combine_chunk(__amdgpu_buffer_rsrc_t rs_part, int voff, int loff, int j0, int nb, unsigned tag, const float (*bmall)[4], int g,
                                float& tot0, float& tot1, float& lt, float& mprev, int* tmo, int lane, SweepWho who) {
  constexpr int PSZ8 = ZN_SK_KB_PSZ * 8;
  u32x4 pv[DEPTH][CH];
  u32x2 pl[DEPTH][CH];                                       // {e sum, tag} of head g: one granule
  auto issue = [&](auto DC) {
    constexpr int d = decltype(DC)::value;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int jo = min(j0 + j, nb - 1) * PSZ8;
      pv[d][j] = ld_sc1_16(rs_part, voff + jo);
      pl[d][j] = ld_sc1_8(rs_part, loff + jo);
    }
  };
  zn_static_for<0, DEPTH>([&](auto DC) { issue(DC); });
  SpinBound sb; sb.begin();
  int got = -1;
  while (got < 0) {
    zn_static_for<0, DEPTH>([&](auto DC) {
      constexpr int d = decltype(DC)::value;
      if (got >= 0) return;
      ++sb.np;
      bool ok = true;
#pragma unroll
      for (int j = 0; j < CH; ++j) ok &= (pv[d][j].y == tag) & (pv[d][j].w == tag) & (pl[d][j].y == tag);
      unsigned long long bad = __builtin_amdgcn_ballot_w64(!ok);
      ZN_TIMING_PASS(bad);
      if (bad == 0ull) { got = d; return; }
      if (sb.give_up(tmo)) {
        unsigned badoff = 0, badtag = 0;
#pragma unroll
        for (int j = CH - 1; j >= 0; --j) {
          const bool okj = (pv[d][j].y == tag) & (pv[d][j].w == tag) & (pl[d][j].y == tag);
          if (!okj) { badoff = (unsigned)(voff + min(j0 + j, nb - 1) * PSZ8); badtag = pv[d][j].y != tag ? pv[d][j].y : pv[d][j].w != tag ? pv[d][j].w : pl[d][j].y; }
        }
        sb.report(tmo, who, bad, lane, tag, badoff, badtag);
        got = d;
        return;
      }
      issue(DC);
    });
  }
  zn_static_for<0, DEPTH>([&](auto DC) {
    constexpr int d = decltype(DC)::value;
    if (got != d) return;                                    // uniform
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      if (j0 + j < nb) {
        const float mj = fmaxf(mprev, bmall[j0 + j][g]);
        const float fj = (j0 + j == 0) ? 0.f : expf(mprev - mj);
        mprev = mj;
        tot0 = __fadd_rn(__fmul_rn(tot0, fj), __uint_as_float(pv[d][j].x));
        tot1 = __fadd_rn(__fmul_rn(tot1, fj), __uint_as_float(pv[d][j].z));
        lt = __fadd_rn(__fmul_rn(lt, fj), __uint_as_float(pl[d][j].x));
      }
    }
  });
}

template <int NBV>          // blocks this instantiation's launches can be asked to cover (<= ZN_SK_KB_MAXNB): the unroll bound of the combine
ZN_DEVINL void step_attention_kb_role(const ChainArgs& a, StepKbLds& S, const unsigned tag0, const int c, const int wave, const int lane_in) {
  static_assert(NBV >= 1 && NBV <= ZN_SK_KB_MAXNB, "blocks covered");
  constexpr int HD = 128, G = 4, NW = 8, KST = HD / 32, TPW = 4, R = 2, MAXNB = ZN_SK_KB_MAXNB, PSZ = ZN_SK_KB_PSZ;
  typedef __attribute__((ext_vector_type(4))) float f32x4_t;
  const int npairs = a.n_heads_kv * R;
  const int pair = c % npairs, jb = c / npairs, kvh = pair % a.n_heads_kv, ar = pair / a.n_heads_kv;
  const int nq = a.n_heads * HD, nk = a.n_heads_kv * HD, D = nq;
  const int L = a.lengths[ar] + 1, nb = (L + 511) >> 9;     // keys including this step's
  if (jb >= nb) return;                                      // a block past the context: nothing to do in this launch
  const int tb = jb * 512, nkeys = min(512, L - tb), tend = tb + nkeys;
  const bool has_newest = jb == nb - 1;                      // row L-1 (this step's key and value) comes from the hand-off, not from the cache
  const int hi = L >= 2 ? L - 2 : 0;                         // rows 0 .. L-2 are in the cache from earlier launches
  const size_t kvrow = (size_t)2 * nk;
  const bool stamped = a.stamps && c == 0 && wave == 0 && lane_in == 0;
  u32x4 kk[TPW][KST];
  AttnV<HD> V;                                               // [32-key step of the wave's 64 keys][key 8 kg + j of the step] x dims 8 kn .. + 8
  const int rowbytes = (int)kvrow * 2;
  // (offsets formed at every issue from a value the optimiser cannot see through, as in the legacy role: hoisted they spill)
  // Everything derived from the lane index is formed anew in every block from a copy of it the optimiser cannot see through: hoisted out
  // of the block loop, the per-lane LDS addresses and granule offsets (~80 VGPRs) stayed live beside the 128 K / V registers and spilled.
  auto issue_kv = [&](const bf16_t* kv, int opq, int ln) {
    const int kn = ln & 15, kg = ln >> 4;
    const __amdgpu_buffer_rsrc_t rs = zn_rsrc(kv);
    const int kbase = ar * a.max_len * rowbytes + (kvh * HD + 8 * kg) * 2, kr0 = opq + tb + wave * 16 + kn;
#pragma unroll
    for (int tl = 0; tl < TPW; ++tl)
#pragma unroll
      for (int st = 0; st < KST; ++st) kk[tl][st] = __builtin_amdgcn_raw_buffer_load_b128(rs, kbase + min(kr0 + tl * NW * 16, hi) * rowbytes + 64 * st, 0, 0);
    // V in the layout the P.V contraction on the matrix cores wants (k = keys): lane (kn, kg) holds keys 8 kg .. + 8 of each 32-key step x dims
    // 8 kn .. + 8, transposed in registers into the B operands; a load instruction covers four whole 256-byte rows
    const int vbase = ar * a.max_len * rowbytes + ((a.n_heads_kv + kvh) * HD + kn * 8) * 2, vr0 = opq + tb + wave * 64 + 8 * kg;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, vbase + min(vr0 + 32 * ks + j, hi) * rowbytes, 0, 0);
        V.v[ks][j][0] = x.x; V.v[ks][j][1] = x.y; V.v[ks][j][2] = x.z; V.v[ks][j][3] = x.w;
      }
  };
  issue_kv(a.layers[0].kv, 0, lane_in);
  const __amdgpu_buffer_rsrc_t rs_bmax = zn_rsrc(a.g_bmax), rs_part = zn_rsrc(a.g_part);
  StepPacer pace{0ull, 0u};
  pace.start();                                              // (with the pre-block the first block's q | k | v is waited for like every other's)
#pragma unroll 1
  for (int li = 0; li < a.n_layer; ++li) {
    const unsigned tag = tag0 + 1u + (unsigned)li;           // (tag0 itself: the pre-block's q | k | v)
    if (a.dbg_pause && li == 3) step_debug_pause(a.dbg_pause);     // (inside the measured wait for block 3's q | k | v)
    const bool st_on = stamped && li == a.stamp_layer;
    auto stamp = [&](int i) { if (st_on) a.stamps[32 + i] = __builtin_amdgcn_s_memrealtime(); };
    stamp(0);
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int kn = lane & 15, kg = lane >> 4, vsub = lane & 3;
    const int tid = wave * 64 + lane;
    // ---- q (4 heads) and, in the newest block's workgroup, this step's key and value row of the kv head -> LDS
    if (wave == ZN_SK_CW || (wave == ZN_SK_CW + 1 && has_newest)) {
      const bool first = wave == ZN_SK_CW;
      if (li == 0 && !a.pre_W) {                             // from the in_proj launch before this one: plain loads
        if (first) *(u32x4*)&S.q[lane >> 4][(lane & 15) * 8] = ld16(a.q0 + ((size_t)ar * a.n_heads + kvh * G + (lane >> 4)) * HD + (lane & 15) * 8);
        else if (lane < 32) {
          const bf16_t* rowp = a.layers[0].kv + ((size_t)ar * a.max_len + (L - 1)) * kvrow + (size_t)(lane < 16 ? 0 : nk) + (size_t)kvh * HD + (lane & 15) * 8;
          const u32x4 v = ld16(rowp);
          if (lane < 16) *(u32x4*)&S.knew[lane * 8] = v; else *(u32x4*)&S.vnew[(lane - 16) * 8] = v;
        }
      } else {                                               // granules of the block before (tag - 1)
        const int qoff = (ar * (a.nqkv / 2) + ((kvh * G) * HD) / 2 + lane * 4) * 8;
        const int kvsel = lane < 16 ? nq + kvh * HD : nq + nk + kvh * HD;
        const int koff = (ar * (a.nqkv / 2) + kvsel / 2 + (lane & 15) * 4) * 8;
        u32x4 d1[1];
        pace.sleep();
        sweep_granules_piped<ZN_SK_APOLL>(zn_rsrc(a.g_qkv), first ? qoff : koff, tag - 1u, d1[0], a.tmo, lane, SweepWho{(5u << 8) | (unsigned)li, a.diag});
        pace.done();
        if (first) *(u32x4*)&S.q[lane >> 4][(lane & 15) * 8] = d1[0];
        else if (lane < 16) *(u32x4*)&S.knew[lane * 8] = d1[0];
        else if (lane < 32) *(u32x4*)&S.vnew[(lane - 16) * 8] = d1[0];
      }
    }
    __syncthreads();                                         // K1: q (and the newest rows) are in LDS
    stamp(1);
    // ---- the block's scores on the matrix cores (the arithmetic of attn_scores_kernel / the fused launch): S[head][key] as 16x16x32 tiles
    {
      zn_bf16x8 qa[KST];
#pragma unroll
      for (int st = 0; st < KST; ++st) {
        u32x4 v = u32x4{0, 0, 0, 0};
        if (kn < G) v = *(const u32x4*)&S.q[kn][32 * st + 8 * kg];
        qa[st] = __builtin_bit_cast(zn_bf16x8, v);
      }
      u32x4 knew_frag[KST];
#pragma unroll
      for (int st = 0; st < KST; ++st) knew_frag[st] = *(const u32x4*)&S.knew[32 * st + 8 * kg];
      float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int tl = 0; tl < TPW; ++tl) {
        const int tt = tb + (tl * NW + wave) * 16;
        if (tt < tend) {                                     // wave-uniform
          const bool newest = tt + kn >= L - 1;
          f32x4_t cc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int st = 0; st < KST; ++st) {
            const u32x4 bfrag = newest ? knew_frag[st] : kk[tl][st];
            cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[st], __builtin_bit_cast(zn_bf16x8, bfrag), cc, 0, 0, 0);
          }
          const int t = tt + kn;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int head = 4 * kg + reg;
            if (head < G && t < tend) {
              const float sv = __fmul_rn(cc[reg], a.scale);
              mx[reg] = fmaxf(mx[reg], sv);
              S.sc[head][t - tb] = sv;
            }
          }
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float m = wave_max(kg == g / 4 ? mx[g % 4] : -INFINITY);
        if (lane == 0) S.bm[wave][g] = m;
      }
    }
    __syncthreads();                                         // K2: every score and per-wave maximum of the block is in LDS
    stamp(2);
    // ---- block maxima of the pair's blocks: publish this block's, sweep all of them (one block: nothing to exchange)
    if (wave == 0) {
      float bmx = -INFINITY;
      if (lane < G) {
#pragma unroll
        for (int w = 0; w < NW; ++w) bmx = fmaxf(bmx, S.bm[w][lane]);
      }
      if (nb > 1) {
        if (lane < G) st_granule(a.g_bmax + ((size_t)pair * MAXNB + jb) * G + lane, tag, __float_as_uint(bmx));
        const bool act = lane < 2 * nb;                      // lane = (block lane >> 1, head pair lane & 1): one 16-byte load = two granules
        const int off = ((pair * MAXNB) * G + 2 * (act ? lane : 0)) * 8;
        if (ZN_SK_XDELAY > 0 && nb >= ZN_SK_XDELAY_NB) __builtin_amdgcn_s_sleep(ZN_SK_XDELAY);
        SpinBound sb; sb.begin();
        u32x4 pv_[ZN_SK_APOLL];
#pragma unroll
        for (int d = 0; d < ZN_SK_APOLL; ++d) pv_[d] = ld_sc1_16(rs_bmax, off);
        for (bool done = false; !done;) {
#pragma unroll
          for (int d = 0; d < ZN_SK_APOLL; ++d) {
            if (done) break;
            ++sb.np;
            const u32x4 v = pv_[d];
            const bool ok = !act || ((v.y == tag) & (v.w == tag));
            unsigned long long bad = __builtin_amdgcn_ballot_w64(!ok);
            ZN_TIMING_PASS(bad);
            if (bad == 0ull) {
              if (act) { S.bmall[lane >> 1][2 * (lane & 1)] = __uint_as_float(v.x); S.bmall[lane >> 1][2 * (lane & 1) + 1] = __uint_as_float(v.z); }
              done = true;
            } else if (sb.give_up(a.tmo)) {
              sb.report(a.tmo, SweepWho{(7u << 8) | (unsigned)li, a.diag}, bad, lane, tag, (unsigned)off + (v.y != tag ? 0u : 8u), v.y != tag ? v.y : v.w);
              done = true;
            } else pv_[d] = ld_sc1_16(rs_bmax, off);
          }
        }
      } else if (lane < G) S.bmall[0][lane] = bmx;
    }
    __syncthreads();                                         // K3: the running maxima are known
    stamp(3);
    // ---- pass 2 on this block's keys: the launches' arithmetic (zn_decode_kernels.h)
    {
      float m_run = -INFINITY;
      for (int j = 0; j < jb; ++j) m_run = fmaxf(m_run, S.bmall[j][vsub]);
      const float mnew[1] = {fmaxf(m_run, S.bmall[jb][vsub])};
      attn_block_probs<G>(S.sc, mnew, nkeys, nkeys & ~15, S.p, S.l, wave, lane);      // decode steps: the reference's block loop spans exactly the context
    }
    __syncthreads();                                         // K3b: P of the whole block is in LDS
    if (has_newest && ((L - 1 - tb) >> 6) == wave) {          // wave-uniform: this step's value row comes from the hand-off, not from the cache
      const u32x4 vnew_piece = *(const u32x4*)&S.vnew[8 * kn];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (tb + wave * 64 + 32 * ks + 8 * kg + j == L - 1) { V.v[ks][j][0] = vnew_piece.x; V.v[ks][j][1] = vnew_piece.y; V.v[ks][j][2] = vnew_piece.z; V.v[ks][j][3] = vnew_piece.w; }
    }
    attn_block_pv<HD, G>(S.p, V, S.accw, wave, lane);
    __syncthreads();                                         // K4: partial sums of all 8 waves
    stamp(4);
    {
      const int g = tid >> 7, d = tid & 127;                 // 512 threads = 4 heads x 128 value columns
      float v = 0.f, l = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += S.accw[w][g][d];
#pragma unroll
      for (int w = 0; w < NW; ++w) l += S.l[w][g];
      if (nb == 1) S.out[g][d] = f2bf(__fmul_rn(v, 1.0f / l));
      else {
        unsigned long long* pp = a.g_part + ((size_t)pair * MAXNB + jb) * PSZ;
        st_granule(pp + g * 128 + d, tag, __float_as_uint(v));
        if (d == 0) st_granule(pp + G * 128 + g, tag, __float_as_uint(l));
      }
    }
    if (nb == 1) __syncthreads();                            // K5 (workgroup-uniform): the result is in LDS
    stamp(5);
    if (wave < 4 && (nb == 1 || jb == 0)) {                  // 256 granules of the pair's attention output: lane = (head, column pair)
      const int p = tid, g = p >> 6, dp = p & 63;
      unsigned o;
      if (nb == 1) o = *(const unsigned*)&S.out[g][2 * dp];
      else {
        // attn_block_kernel's combine: the reference's recurrence over the blocks, in order
        const int voff = ((pair * MAXNB) * PSZ + g * 128 + 2 * dp) * 8, loff = ((pair * MAXNB) * PSZ + G * 128 + g) * 8;
        // in chunks of up to 6 blocks, sized to the context (kb_combine_chunk)
        float tot0 = 0.f, tot1 = 0.f, lt = 0.f, mprev = -INFINITY;
        const SweepWho who{(8u << 8) | (unsigned)li, a.diag};
        if (ZN_SK_XDELAY > 0 && nb >= ZN_SK_XDELAY_NB) __builtin_amdgcn_s_sleep(ZN_SK_XDELAY);
        if (nb <= 2) kb_combine_chunk<2, ZN_SK_APOLL>(rs_part, voff, loff, 0, nb, tag, S.bmall, g, tot0, tot1, lt, mprev, a.tmo, lane, who);
        else if (nb <= 4) kb_combine_chunk<4, 2>(rs_part, voff, loff, 0, nb, tag, S.bmall, g, tot0, tot1, lt, mprev, a.tmo, lane, who);
        else {
#pragma unroll
          for (int j0 = 0; j0 < NBV; j0 += 6)
            if (j0 < nb) kb_combine_chunk<6, 1>(rs_part, voff, loff, j0, nb, tag, S.bmall, g, tot0, tot1, lt, mprev, a.tmo, lane, who);      // uniform
        }
        const float inv = 1.0f / lt;
        o = pack2(__fmul_rn(tot0, inv), __fmul_rn(tot1, inv));
      }
      st_granule(a.g_a + (size_t)ar * (D / 2) + ((kvh * G + g) * HD) / 2 + dp, tag, o);
      if (a.trace) *(unsigned*)(a.trace + ((size_t)(8 * li + 1) * R + ar) * D + (kvh * G + g) * HD + 2 * dp) = o;
    }
    if (a.trace && jb == 0 && wave == ZN_SK_CW)
      *(u32x4*)(a.trace + ((size_t)(8 * li + 2) * R + ar) * D + (kvh * G + (lane >> 4)) * HD + (lane & 15) * 8) = *(const u32x4*)&S.q[lane >> 4][(lane & 15) * 8];
    stamp(6);
    // the key and value registers are free: the next block's rows are requested now, a whole block (~20 us) ahead of their use
    if (li + 1 < a.n_layer) {
      int opq = 0;
      asm volatile("" : "+v"(opq));
      issue_kv(a.layers[li + 1].kv, opq, lane);
    }
    pace.start();                                            // the wait for the next block's q | k | v starts here
  }
  if (wave == ZN_SK_CW) pace.report(a.diag, lane_in);
}

// ------------------------------------------------------------------------------------------------ the launch
// T_* = tiles per compute wave per op (upper bounds: the matrices do not divide evenly over 224 workgroups; a wave skips the tiles
// its workgroup does not have).  d_model = 512 * NCH, d_ff = 4 * d_model (host-checked, as are the bounds).
// NBV: key blocks a launch of this instantiation may be asked to cover (a.natt = 8 * blocks of the launch's context bound <= 8 * NBV).
template <int NCH, int T_OUT, int T_FC1, int T_FC2, int T_IN, int NBV>
__global__ __launch_bounds__(ZN_SK_THREADS) void step_kernel(ChainArgs a_in) {
  // Integer arguments that the whole kernel keeps using are detached from the kernarg segment's wide scalar loads: read as parts of an
  // s_load_dwordx8 they made the register allocator spill the 256-bit tuple, rematerialise it instead, and leave its 32-byte stack slot
  // behind - a private segment without a single scratch instruction, which a persistent kernel must not have (tests/test_abi.py).
  // (Pointers are left alone: an asm operand has no address space, and accesses through a laundered pointer become flat_ instructions.)
  ChainArgs a = a_in;
  asm volatile("" : "+s"(a.max_len), "+s"(a.hd), "+s"(a.n_heads), "+s"(a.n_heads_kv), "+s"(a.rope_positions), "+s"(a.F), "+s"(a.nqkv), "+s"(a.natt), "+s"(a.dbg_pause));
  constexpr int R = 2, D = NCH * 512, CW = ZN_SK_CW;
  constexpr int NS = 2 * T_OUT + T_FC1 + T_FC2 + T_IN;         // slots of a block (StepSched)
  constexpr int NOPS = 5;
  constexpr int MASK = ZN_CH_DEFER_MASK;
  constexpr int NB = ZN_SK_NBUF, P = ZN_SK_PARK;
  constexpr int NL = NS - T_OUT;                              // distinct tiles (requests) of a block: op 1 reuses op 0's
  static_assert(T_IN > 0 && NCH == 4, "step_kernel: d_model 2048, head size 128");
  static_assert(P >= T_OUT && NB >= 2 && P + NB <= NL, "op 0's tiles are parked (op 1 reads them again); the prefetch stays inside the block");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = blockIdx.x;
  const int F = a.F;
  const unsigned tag0 = ld_sc1_u32(a.epoch);
  extern __shared__ __attribute__((aligned(16))) unsigned char zn_dyn_lds[];
  const int natt = a.natt;
  if (c < natt) {
    step_attention_kb_role<NBV>(a, *reinterpret_cast<StepKbLds*>(zn_dyn_lds), tag0, c, wave, lane);
    return;
  }
  // ------------------------------------------------------------------------------------------------ streaming role
  __shared__ __attribute__((aligned(16))) bf16_t s_act[R * D];              // the current op's input vector (ops 0, 1, 2, 4)
  __shared__ float s_res[2][64][2][R];                                      // per-unit results of even / odd ops (fc2: [unit * 4 + quarter])
  __shared__ __attribute__((aligned(16))) bf16_t s_ln[4 * D];               // the block's LayerNorm parameters (norm2 w, b; next norm w, b): staged by the helper waves
  const int sc = c - natt, NSW = (int)gridDim.x - natt;
  auto split = [&](int U, int& n, int& start) { const int base = U / NSW, extra = U % NSW; n = base + (sc < extra ? 1 : 0); start = sc * base + min(sc, extra); };
  int n_out, s_out, ng1, sg1;
  split(D / 2, n_out, s_out);                                 // row pairs of out_proj (ops 0, 1) and fc2 (op 3)
  split(F / 2, ng1, sg1);                                     // fc1: pairs of m elements (one granule each), so that no granule straddles two workgroups
  const int n_fc1 = 2 * ng1, s_fc1 = 2 * sg1;
  int n_qkv, s_qkv, n_hd, s_hd;
  split((a.nqkv + 1) / 2, n_qkv, s_qkv);
  split((a.heads_rows + 1) / 2, n_hd, s_hd);
  constexpr int NBAR = 14;                                    // workgroup barriers per block (every wave of a streaming workgroup runs them all)
  // ---- pre-block (a.pre_W): LayerNorm + in_proj + RoPE + KV append of block 0 from the step's embedding, by the waves that have nothing to
  // do at the start of a launch - the two communication waves normalise a row each, the two helper waves hold the workgroup's <= 10 weight
  // row pairs (requested at kernel entry) and contract them, communication wave 0 runs the epilogue and publishes q | k | v under tag0 - while
  // the compute waves park block 0's tiles as in every block.  The four waves meet through two counters in LDS, not through workgroup
  // barriers (the compute waves would have to join those).  Same arithmetic as gemv_kernel<PRO_LN, EPI_ROPE_KV> / the op-4 path of a block.
  __shared__ int s_pf[2];                                                   // [0] rows normalised (2), [1] helper waves done (2)
  constexpr int NPT = 5;                                                    // pre-block row pairs per helper wave (host-checked: <= 10 per workgroup)
  const bool pre = a.pre_W != nullptr;
  if (tid < 2) s_pf[tid] = 0;
  __syncthreads();

  using SC = StepSched<T_OUT, T_FC1, T_FC2, T_IN, NB, P, ZN_SK_HELP, MASK, ZN_SK_EARLY>;
  auto op_of = [](int s) constexpr { return SC::op_of(s); };
  auto first_of = [](int op) constexpr { return SC::first_of(op); };

  // ---- the block's tiles ("loads" l = 0 .. NL-1 in consumption order: op 0's T_OUT tiles, which op 1 reads again, then fc1, fc2, in_proj / heads)
  // and where each comes from: PARK = streamed through the wave's register buffers into its LDS slots 0 .. P-1 while the attention runs;
  // HELP = held by a helper wave in ITS registers over the attention and dropped into slots 0 .. NH-1 once op 1 has read them for the
  // last time (the last NH fc1 tiles); REG = the rotating register buffers, requested NB tiles ahead.
  constexpr int L_F2 = SC::L_F2;                              // first load of fc2
  constexpr int NH = ZN_SK_HELP;
  static_assert(NH >= 0 && NH <= T_OUT && NH < T_FC1 - (P - T_OUT), "helper tiles take the slots of op 0's tiles");
  constexpr int NREG = SC::NREG;
  auto slot_of_load = [](int l) constexpr { return SC::slot_of_load(l); };
  struct WT { u32x4 a[NCH], b[NCH]; };
  // tile of slot s for compute wave w: exists?, weight pointers of rows A and B (lane's first chunk), result index
  struct LW { const bf16_t *out, *fc1, *fc2, *in; };           // the block's weight matrices, read from the layer table ONCE per block (SGPRs)
  auto tile_of = [&](const LW& Lr, int rows_in, int n_in, int s_in, int s, int w, bool& ok, const bf16_t*& pa, const bf16_t*& pb, int& ridx) {
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
    const int n = op <= 1 ? n_out : op == 2 ? n_fc1 : n_in;
    const int st = op <= 1 ? s_out : op == 2 ? s_fc1 : s_in;
    ok = j < n;
    const int u = st + (ok ? j : 0);
    ridx = j;
    if (op <= 1) { pa = Lr.out + (size_t)(2 * u) * D + lane * 8; pb = pa + D; }
    else if (op == 2) { pa = Lr.fc1 + (size_t)u * D + lane * 8; pb = pa + (size_t)F * D; }
    else { pa = Lr.in + (size_t)(2 * u) * D + lane * 8; pb = (2 * u + 1 < rows_in) ? pa + D : pa; }   // half pair: row B repeats row A, result dropped
  };
  auto park_of = [&](int w) { return reinterpret_cast<u32x4*>(zn_dyn_lds) + (size_t)w * (P * 2 * NCH * 64) + lane; };

  if (wave >= CW + 2) {
    // ------------------------------------------------------------------------------------ helper waves
    const int hw = wave - (CW + 2);                           // serves compute waves 2 hw and 2 hw + 1
    WT hb[2 * (NH > 0 ? NH : 1)];
    if (pre) {
      WT pt[NPT];
#pragma unroll
      for (int t = 0; t < NPT; ++t) {
        const int j = hw + 2 * t;
        if (j < n_qkv) {                                      // wave-uniform
          const int u = s_qkv + j;
          const bf16_t* pa = a.pre_W + (size_t)(2 * u) * D + lane * 8;
          const bf16_t* pb = (2 * u + 1 < a.nqkv) ? pa + D : pa;
#pragma unroll
          for (int c2 = 0; c2 < NCH; ++c2) { pt[t].a[c2] = ld_nt16(pa + c2 * 512); pt[t].b[c2] = ld_nt16(pb + c2 * 512); }
        }
      }
      while (__hip_atomic_load(&s_pf[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < R) __builtin_amdgcn_s_sleep(1);
      u32x4 xr[NCH][R];
#pragma unroll
      for (int c2 = 0; c2 < NCH; ++c2)
#pragma unroll
        for (int r = 0; r < R; ++r) xr[c2][r] = *(const u32x4*)&s_act[r * D + (c2 * 64 + lane) * 8];
#pragma unroll
      for (int t = 0; t < NPT; ++t) {
        const int j = hw + 2 * t;
        if (j < n_qkv) {
          float accA[R] = {0.f, 0.f}, accB[R] = {0.f, 0.f};
#pragma unroll
          for (int c2 = 0; c2 < NCH; ++c2) {
#pragma unroll
            for (int r = 0; r < R; ++r) { accA[r] = dot8(pt[t].a[c2], xr[c2][r], accA[r]); accB[r] = dot8(pt[t].b[c2], xr[c2][r], accB[r]); }
          }
#pragma unroll
          for (int r = 0; r < R; ++r) { accA[r] = wave_sum(accA[r]); accB[r] = wave_sum(accB[r]); }
          if (lane == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) { s_res[0][j][0][r] = accA[r]; s_res[0][j][1][r] = accB[r]; }
          }
        }
      }
      if (lane == 0) __hip_atomic_fetch_add(&s_pf[1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#pragma unroll 1
    for (int li = 0; li < a.n_layer; ++li) {
      const LW Lr{a.layers[li].W_out, a.layers[li].W_fc1, a.layers[li].W_fc2, a.layers[li].W_in};
      auto hload = [&]() {
        if constexpr (NH > 0) {
          zn_static_for<0, 2 * NH>([&](auto IC) {
            constexpr int i = decltype(IC)::value, cwi = i / NH, hh = i % NH;
            bool ok; const bf16_t *pa, *pb; int ridx;
            tile_of(Lr, a.nqkv, 0, 0, slot_of_load(L_F2 - NH + hh), 2 * hw + cwi, ok, pa, pb, ridx);     // an fc1 tile: the op-4 arguments are unused
            if (ok) {
#pragma unroll
              for (int c2 = 0; c2 < NCH; ++c2) { hb[i].a[c2] = ld_nt16g(pa + c2 * 512); hb[i].b[c2] = ld_nt16g(pb + c2 * 512); }
            }
          });
        }
      };
      {   // the block's LayerNorm parameters -> LDS (the communication waves' registers are for sweep passes in flight); the previous
          // block's were last read before its B(4)
        const bf16_t* p0 = hw == 0 ? a.layers[li].ln2_w : a.layers[li].lnn_w;
        const bf16_t* p1 = hw == 0 ? a.layers[li].ln2_b : a.layers[li].lnn_b;
#pragma unroll
        for (int c2 = 0; c2 < NCH; ++c2) {
          *(u32x4*)&s_ln[(2 * hw) * D + (c2 * 64 + lane) * 8] = ld16g(p0 + (c2 * 64 + lane) * 8);
          *(u32x4*)&s_ln[(2 * hw + 1) * D + (c2 * 64 + lane) * 8] = ld16g(p1 + (c2 * 64 + lane) * 8);
        }
      }
      if constexpr (ZN_SK_HELP_AT == 0) hload();
      __syncthreads();                                        // B(0)
      if constexpr (ZN_SK_HELP_AT == 1) hload();
      __syncthreads();                                        // A(0)
      __syncthreads();                                        // P(0)
      if constexpr (ZN_SK_HELP_AT == 2) hload();
      __syncthreads();                                        // B(1)
      __syncthreads();                                        // A(1): op 1 has read slots 0 .. T_OUT-1 for the last time
      if constexpr (NH > 0) {
        zn_static_for<0, 2 * NH>([&](auto IC) {
          constexpr int i = decltype(IC)::value, cwi = i / NH, hh = i % NH;
          bool ok; const bf16_t *pa, *pb; int ridx;
          tile_of(Lr, a.nqkv, 0, 0, slot_of_load(L_F2 - NH + hh), 2 * hw + cwi, ok, pa, pb, ridx);
          if (ok) {
            u32x4* pk = park_of(2 * hw + cwi);
#pragma unroll
            for (int c2 = 0; c2 < NCH; ++c2) { pk[(hh * 2 * NCH + c2) * 64] = hb[i].a[c2]; pk[(hh * 2 * NCH + NCH + c2) * 64] = hb[i].b[c2]; }
          }
        });
      }
#pragma unroll
      for (int b = 5; b < NBAR; ++b) __syncthreads();         // P(1) (the tiles are in LDS before any compute wave passes B(2)) ... P(4)
    }
    return;
  }

  if (wave < CW) {
    // ------------------------------------------------------------------------------------ compute waves
    WT bufs[NB];                                           // every index below is a compile-time constant: the buffers live in registers
    u32x4* park = park_of(wave);
    static_assert(CW == 4, "fc2 splits K over the four compute waves");
    const bool st_on = a.stamps && sc == 0 && wave == 0 && lane == 0;
#pragma unroll 1
    for (int li = 0; li < a.n_layer; ++li) {
      const LW Lr{a.layers[li].W_out, a.layers[li].W_fc1, a.layers[li].W_fc2, a.layers[li].W_in};
      const bool last = li + 1 == a.n_layer;
      const int rows_in = last ? a.heads_rows : a.nqkv;
      const int n_in = last ? n_hd : n_qkv, s_in = last ? s_hd : s_qkv;
      const unsigned tag = tag0 + 1u + (unsigned)li;
      const bool stamped = st_on && li == a.stamp_layer;
      auto cstamp = [&](int i) { if (stamped) a.stamps[40 + i] = __builtin_amdgcn_s_memrealtime(); };
      auto tile = [&](int s, bool& ok, const bf16_t*& pa, const bf16_t*& pb, int& ridx) { tile_of(Lr, rows_in, n_in, s_in, s, wave, ok, pa, pb, ridx); };
      // load l into register buffer b
      auto load_into = [&](auto LC, auto BC) {
        constexpr int l = decltype(LC)::value, b = decltype(BC)::value;
        if constexpr (l >= 0 && l < NL) {
          bool ok; const bf16_t *pa, *pb; int ridx;
          tile(slot_of_load(l), ok, pa, pb, ridx);
          if (ok) {                                         // wave-uniform
            WT& w = bufs[b];
#ifdef ZN_TIMING_HANDOFFS_ONLY
#pragma unroll
            for (int c2 = 0; c2 < NCH; ++c2) { w.a[c2] = u32x4{0, 0, 0, 0}; w.b[c2] = u32x4{0, 0, 0, 0}; }
#else
#pragma unroll
            for (int c2 = 0; c2 < NCH; ++c2) { w.a[c2] = ld_nt16g(pa + c2 * 512); w.b[c2] = ld_nt16g(pb + c2 * 512); }
#endif
          }
        }
      };
      // REG load number k lives in buffer (k + P) % NB: the buffer that the prefetch phase's transit of load k + P - NB ... has just left
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
      // ---- prefetch while the attention runs: loads 0 .. P-1 through the register buffers into LDS, then the first NB REG loads stay in flight
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
        constexpr int i = l + NB;                           // next item of [transit 0 .. P-1, REG 0 .. NB-1] into the buffer just drained
        if constexpr (i < P) load_into(std::integral_constant<int, i>{}, std::integral_constant<int, i % NB>{});
        else reg_req(std::integral_constant<int, i - P>{});
      });
      cstamp(1);
      zn_static_for<0, NS>([&](auto SC) {
        constexpr int s = decltype(SC)::value;
        constexpr int op = op_of(s);
        if constexpr (s == first_of(op)) {
          if constexpr (op > 0) {
            __syncthreads();                                // A(op-1): this workgroup's results of the previous op are in LDS
            __syncthreads();                                // P(op-1): ... and published; the requests held back for that go out now
            zn_static_for<SC::first_of(op - 1), SC::first_of(op)>([&](auto QC) {
              constexpr int q = decltype(QC)::value;
              if constexpr (SC::raise_late(q)) reg_req(std::integral_constant<int, SC::raised_by(q)>{});
            });
          }
          if constexpr (op == 3) {
            // fc2's input m [2][4 d]: this wave's K quarter straight from the granules (no LDS, no barrier)
            u32x4 dat[NCH * R];
            const int mbase = (wave * (D / 2) + lane * 4) * 8;
            if constexpr (ZN_SK_MSWEEP_DELAY > 0) __builtin_amdgcn_s_sleep(ZN_SK_MSWEEP_DELAY);
            sweep_granules_at<NCH * R>(zn_rsrc(a.g_m), [&](int i) { return mbase + ((i % R) * (2 * D) + (i / R) * 256) * 8; }, tag, dat, a.tmo, lane,
                                       SweepWho{(3u << 8) | (unsigned)li, a.diag});
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
          cstamp(2 + 2 * op);
        }
        constexpr int l = SC::load_of_slot(s);
        if constexpr (SC::src_of(l) != 0) {                 // parked tile (own prefetch or a helper wave's): from LDS
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
        if constexpr (s + 1 == NS || op_of(s + 1 < NS ? s + 1 : s) != op) cstamp(3 + 2 * op);
      });
      __syncthreads();                                      // A(4)
      __syncthreads();                                      // P(4): q | k | v are published; the next block's prefetch may enter the CU's queue
    }
    return;
  }

  // -------------------------------------------------------------------------------------- communication waves
  // Wave CW + r gathers (sweeps), normalises and stages row r of every hand-off; wave CW also runs the row-pair epilogues.
  const int myr = wave - CW;
  const bool epi = myr == 0;
  u32x4 g[NCH];
  const int ij = lane >> 1, ir = lane & 1;
  const bool it_out = epi && ij < n_out;
  const int u_out = s_out + (it_out ? ij : 0);
  unsigned resid = 0;
  if (it_out) resid = *(const unsigned*)(a.xin + (size_t)ir * D + 2 * u_out);
  const int nq = a.n_heads * a.hd, nk = a.n_heads_kv * a.hd;
  const bool it_qkv = epi && ij < n_qkv;
  const int u_qkv = s_qkv + (it_qkv ? ij : 0);
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
  int goff[NCH];
#pragma unroll
  for (int c2 = 0; c2 < NCH; ++c2) goff[c2] = (myr * (D / 2) + (c2 * 64 + lane) * 4) * 8;
  unsigned x1own = 0;
  if (pre) {
    // row myr of the step's embedding (written by the previous step's tail: plain loads) -> LayerNorm of block 0 -> s_act
    u32x4 lw[NCH], lb[NCH];
#pragma unroll
    for (int c2 = 0; c2 < NCH; ++c2) {
      g[c2] = ld16(a.xin + (size_t)myr * D + (c2 * 64 + lane) * 8);
      lw[c2] = ld16(a.pre_ln_w + (c2 * 64 + lane) * 8);
      lb[c2] = ld16(a.pre_ln_b + (c2 * 64 + lane) * 8);
    }
    chain_layernorm_row<NCH>(g, lw, lb, a.eps);
#pragma unroll
    for (int c2 = 0; c2 < NCH; ++c2) *(u32x4*)&s_act[myr * D + (c2 * 64 + lane) * 8] = g[c2];
    if (lane == 0) __hip_atomic_fetch_add(&s_pf[0], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    // both communication waves wait for the helper waves: s_act and s_res are theirs again only then
    while (__hip_atomic_load(&s_pf[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 2) __builtin_amdgcn_s_sleep(1);
    if (it_qkv) {                                            // EPI_ROPE_KV (gemv_epilogue) of block 0, q | k | v also as granules under tag0
      const int rowA = 2 * u_qkv;
      const float x0 = bfround(s_res[0][ij][0][ir]), x1 = bfround(s_res[0][ij][1][ir]);
      unsigned packed;
      if (rowA < nq + nk) {
        float re, im;
        zn_rope_pair(x0, x1, cs, sn, re, im);
        packed = pack2(re, im);
      } else packed = pack2(x0, x1);
      st_granule(a.g_qkv + (size_t)ir * (a.nqkv / 2) + u_qkv, tag0, packed);
      if (rowA >= nq && pos < a.max_len) {
        const int which = rowA < nq + nk ? 0 : 1, colk = rowA - nq - which * nk;
        *(unsigned*)(a.pre_kv + (((size_t)ir * a.max_len + pos) * 2 + which) * nk + colk) = packed;
      }
    }
  }
  StepPacer pace{0ull, 0u};
  pace.start();
#pragma unroll 1
  for (int li = 0; li < a.n_layer; ++li) {
    const StackLayer& Lr = a.layers[li];
    const bool last = li + 1 == a.n_layer;
    const unsigned tag = tag0 + 1u + (unsigned)li;
    const bool stamped = a.stamps && li == a.stamp_layer && epi && sc == 0 && lane == 0;
    int nst = 0;
    auto stamp = [&]() { if (stamped) a.stamps[nst] = __builtin_amdgcn_s_memrealtime(); ++nst; };
    stamp();                                               // 0: block starts
    int opq = 0;                                           // 0 the optimiser cannot see through: the granule addresses below are formed where they are used
    asm volatile("" : "+v"(opq));                          // (hoisted out of the block loop they sat beside the sweep passes in flight and spilled)
    const int irq = ir + opq, uoq = u_out + opq, uqq = u_qkv + opq;
    // ---- the attention output of all heads -> s_act
    pace.sleep();
    stamp();                                               // 1: polling starts
    sweep_granules<NCH>(zn_rsrc(a.g_a), goff, tag, g, a.tmo, lane, SweepWho{(6u << 8) | (unsigned)li, a.diag});
    pace.done();
#pragma unroll
    for (int c2 = 0; c2 < NCH; ++c2) *(u32x4*)&s_act[myr * D + (c2 * 64 + lane) * 8] = g[c2];
    __syncthreads();                                       // B(0)
    stamp();                                               // 2: op 0's input ready
    zn_static_for<0, NOPS>([&](auto OC) {
      constexpr int op = decltype(OC)::value;
      constexpr int par = op & 1;
      __syncthreads();                                     // A(op): every compute wave's results are in LDS
      stamp();
      // ---- epilogue of this workgroup's units, published as granules
      if constexpr (op == 0) {                             // EPI_STORE
        if (it_out) st_granule(a.g_y1 + (size_t)irq * (D / 2) + uoq, tag, pack2(s_res[par][ij][0][ir], s_res[par][ij][1][ir]));
      } else if constexpr (op == 1) {                      // EPI_RESID
        if (it_out) {
          x1own = pack2(lo_f(resid) + bfround(s_res[par][ij][0][ir]), hi_f(resid) + bfround(s_res[par][ij][1][ir]));
          st_granule(a.g_x1 + (size_t)irq * (D / 2) + uoq, tag, x1own);
        }
      } else if constexpr (op == 2) {                      // EPI_SILU: each communication wave its own row; lane = m element, neighbours share a granule
        const bool on = lane < n_fc1;
        const int jj = on ? lane : 0;
        const float y = bfround(s_res[par][jj][0][myr]), gt = bfround(s_res[par][jj][1][myr]);
        const float sg = bfround(gt / (1.0f + expf(-gt)));
        const unsigned mine = (unsigned)f2bf(y * sg);
        const unsigned nbv = (unsigned)__shfl_down((int)mine, 1);
        if (on && (lane & 1) == 0) st_granule(a.g_m + (size_t)(myr + opq) * (F / 2) + ((s_fc1 + lane) >> 1), tag, mine | (nbv << 16));
      } else if constexpr (op == 3) {                      // EPI_RESID over the four K quarters, in gemv_kernel's order
        if (it_out) {
          const float vA = ((s_res[par][ij * 4 + 0][0][ir] + s_res[par][ij * 4 + 1][0][ir]) + s_res[par][ij * 4 + 2][0][ir]) + s_res[par][ij * 4 + 3][0][ir];
          const float vB = ((s_res[par][ij * 4 + 0][1][ir] + s_res[par][ij * 4 + 1][1][ir]) + s_res[par][ij * 4 + 2][1][ir]) + s_res[par][ij * 4 + 3][1][ir];
          const unsigned o = pack2(lo_f(x1own) + bfround(vA), hi_f(x1own) + bfround(vB));
          st_granule(a.g_x2 + (size_t)irq * (D / 2) + uoq, tag, o);
          if (last) *(unsigned*)(a.xout + (size_t)irq * D + 2 * uoq) = o;
          if (a.trace) *(unsigned*)(a.trace + ((size_t)(8 * li) * R + irq) * D + 2 * uoq) = o;
          resid = o;                                       // the residual stream entering the next block
        }
      } else {
        if (last) {                                        // EPI_F32 (gemv_epilogue): bf16-valued fp32 logits
          if (epi && ij < n_hd) {
            const int u = s_hd + ij;
            a.heads_out[(size_t)irq * a.heads_rows + 2 * u] = bfround(s_res[par][ij][0][ir]);
            if (2 * u + 1 < a.heads_rows) a.heads_out[(size_t)irq * a.heads_rows + 2 * u + 1] = bfround(s_res[par][ij][1][ir]);
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
          st_granule(a.g_qkv + (size_t)irq * (a.nqkv / 2) + uqq, tag, packed);
          if (rowA >= nq && pos < a.max_len) {
            const int which = rowA < nq + nk ? 0 : 1, colk = rowA - nq - which * nk;
            *(unsigned*)(Lr.kv_next + (((size_t)irq * a.max_len + pos) * 2 + which) * nk + colk) = packed;
          }
        }
      }
      __syncthreads();                                     // P(op): published; the compute waves may queue weight requests again
      stamp();
      if constexpr (op + 1 < NOPS) {
        if constexpr (op != 2) {                           // (fc2's input is swept by the compute waves)
          if constexpr (ZN_SK_SWEEP_DELAY > 0) __builtin_amdgcn_s_sleep(ZN_SK_SWEEP_DELAY);
          sweep_granules<NCH>(zn_rsrc(op == 0 ? a.g_y1 : op == 1 ? a.g_x1 : a.g_x2), goff, tag, g, a.tmo, lane,
                              SweepWho{((op == 0 ? 1u : op == 1 ? 2u : 4u) << 8) | (unsigned)li, a.diag});
          stamp();
          if constexpr (op == 1 || op == 3) {
            u32x4 lw[NCH], lb[NCH];
#pragma unroll
            for (int c2 = 0; c2 < NCH; ++c2) {
              lw[c2] = *(const u32x4*)&s_ln[(op == 1 ? 0 : 2) * D + (c2 * 64 + lane) * 8];
              lb[c2] = *(const u32x4*)&s_ln[(op == 1 ? 1 : 3) * D + (c2 * 64 + lane) * 8];
            }
            chain_layernorm_row<NCH>(g, lw, lb, a.eps);
          }
#pragma unroll
          for (int c2 = 0; c2 < NCH; ++c2) *(u32x4*)&s_act[myr * D + (c2 * 64 + lane) * 8] = g[c2];
          __syncthreads();                                 // B(op + 1)
        } else stamp();
        stamp();
      }
    });
    if (a.dbg_pause && li == 2) step_debug_pause(a.dbg_pause);     // (outside the measured wait)
    pace.start();                                          // the wait for the next block's attention output starts here
  }
  pace.report(a.diag, lane);
  if (epi && sc == 0 && lane == 0) st_sc1_u32(a.epoch, tag0 + 1u + (unsigned)a.n_layer);   // every workgroup read the epoch before its first publish, which this one has seen
}
