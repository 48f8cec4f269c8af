// libzonos_hip.so — C ABI (include/zonos_hip.h) over the gfx950 kernels.  Host logic only: argument checks,
// workspace, launch geometry, hipGraph capture of one decode step.
#include "../../include/zonos_hip.h"
#include "zn_decode_kernels.h"
#include "zn_chain_kernel.h"
#include "zn_step_kernel.h"
#include "zn_prefill_kernels.h"
#include "zn_cond_kernels.h"
#include "zn_mamba_kernels.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

static thread_local std::string g_create_err;
#define ZN_GRAPH_STEPS 8
#define ZN_G16_PART_BYTES ((size_t)8 << 20)
#define ZN_G16_MAX_GROUPS 1024
#define ZN_REARM_AFTER 4

struct zn_handle_s {
  zn_config cfg;
  int max_rows = 0, hd = 0, G = 0;
  std::vector<zn_layer_weights> layers;
  const void* heads = nullptr;
  const void *norm_f_w = nullptr, *norm_f_b = nullptr;
  const float* rope = nullptr;
  const bf16_t** emb_tables_dev = nullptr;
  bool has_io = false;  // embeddings + heads bound (false: backbone-only handle)
  // workspace (device)
  bf16_t *x = nullptr, *q = nullptr, *o1 = nullptr, *mbuf = nullptr, *nbuf = nullptr;
  // hybrid backbone (arch 1): residual stream, normalised activations, Mamba2 intermediates
  bf16_t *res = nullptr, *hn = nullptr, *m_zx = nullptr, *m_xbc = nullptr, *m_y = nullptr, *m_g = nullptr;
  float* m_vg = nullptr;            // [rows][m_d_inner] fp32 y * silu(z) (mamba_ssm_kernel -> out_proj prologue)
  int m_nheads = 0, m_conv_dim = 0, m_d_in_proj = 0;
  // persistent post-attention chain (zn_chain_kernel.h): granule buffers of the four in-launch hand-offs ([rows][len / 2] x 8 B),
  // the launch epoch (tag), the second residual-stream buffer (blocks alternate h->x / ch_x2)
  unsigned long long *ch_gy1 = nullptr, *ch_gx1 = nullptr, *ch_gx2 = nullptr, *ch_gm = nullptr;
  unsigned long long *ch_gqkv = nullptr, *ch_ga = nullptr;   // whole-step kernel: q | k | v of the next block, attention output
  unsigned long long *ch_gbmax = nullptr, *ch_gpart = nullptr;   // key-block attention role: per-block score maxima and P.V partials (zn_step_kernel.h)
  int stack_nbk = 1;                         // key blocks (8 attention workgroups each) of the whole-step launches being enqueued
  bool stackv_ok[4] = {};                    // whole-step kernel instantiations 1 .. 3 that fit this model and device (stack_variant_ok)
  StackLayer* stack_layers = nullptr;        // device table [n_layer], rebuilt by zn_gen_begin (it holds the KV cache pointers)
  bool use_stack = false, stack_ok = false, stack_checked = false;   // use_stack: the steps being enqueued run the whole-step kernel
  unsigned* ch_epoch = nullptr;
  unsigned long long epoch_bound = 1;        // host-side upper bound of the device epoch (tags advance by one per block of every decode step enqueued)
  unsigned* ch_diag = nullptr;               // [16] words: [0..7] the first hand-off wait that timed out describes itself (sweep_granules); [8] the longest hand-off wait a whole-step launch measured (10 ns ticks), [9] waits beyond 0.2 ms (StepPacer::report)
  unsigned diag_host[8] = {};
  bf16_t* ch_x2 = nullptr;
  bf16_t* dbg_trace = nullptr;               // diagnostic: [n_layer][2][rows * d] copies of (x after the block, attention output) per decode step
  bf16_t* x_emb = nullptr;                  // [max_rows][d] embedding of the column the next decode step consumes (written by the step's tail)
  int* tail_ticket = nullptr;               // arrival ticket of the sampler launch whose last workgroup runs the step's tail
  bool emb_valid = false;                   // x_emb holds the embedding of column st->offset
  unsigned long long* at_stamps = nullptr;   // diagnostic: [n_layer][8] timeline of the fused attention launch (second half of the chain stamp buffer)
  unsigned long long* ch_stamps = nullptr;   // diagnostic: [n_layer][32] timeline stamps of workgroup 0 (zn_debug_chain_stamps)
  int device = 0;              // the HIP device the handle was created on
  // hand-off timeouts (zn_get_counters): a reported timeout demotes the handle to the launches path (no in-launch hand-offs) until ZN_REARM_AFTER
  // generations in a row have completed there cleanly, or zn_debug_tune(8, 1) re-arms it at once
  bool demoted = false;
  int clean_since_demotion = 0;
  long long n_timeouts = 0, n_generations = 0, n_fallback_generations = 0, n_rearms = 0;
  bool gen_timed_out = false;         // this generation reported a hand-off timeout
  hipStream_t gen_stream = nullptr;   // the stream the generation's steps were last enqueued on (handoff_timeout drains it)
  bool persist_ok = true;      // this handle may launch the persistent kernels (zn_gen_begin: it holds the device's tenancy, or nobody competes)
  int ch_variant = 0;          // 0 = shapes do not fit (launches path), 1 = <4,1,8,4,2> (Zonos-v0.1 dims), 2 = <1,1,2,1,1> (d_model 512)
  float* g16_part = nullptr;   // gemm16s_kernel: split-K partial tiles
  int* g16_tickets = nullptr;
  size_t g16_part_bytes = 0;
  float* ln_part = nullptr;    // LayerNorm statistics per (row, 16-column tile) handed from out_proj's epilogue to fc1 (rows 5..16): [max_rows][ZN_G16_LNT][2]
  float *logits_raw = nullptr, *last_logits = nullptr;
  int* tok_raw = nullptr;
  float *scores = nullptr, *cmax = nullptr;
  float* pv_part = nullptr;    // split P.V pass: partials per (row, kv head, block)
  int* pv_tickets = nullptr;
  bf16_t *pf_x = nullptr, *pf_n = nullptr, *pf_qkv = nullptr, *pf_a = nullptr, *pf_u = nullptr, *pf_m = nullptr;   // batched-prefill workspace
  // hybrid batched prefill: residual stream (bf16, or fp32 with residual_in_fp32), in_proj output, conv output, scan output, gated-normalised
  bf16_t *pf_res = nullptr, *pf_zx = nullptr, *pf_xbc = nullptr, *pf_y = nullptr, *pf_g = nullptr;
  size_t pf_rows = 0;
  int* fw_lengths = nullptr;   // [max_rows] positions of zn_op_backbone_forward's rows
  bf16_t* qkv_tmp = nullptr;   // [rows][(H + 2 Hkv) hd]: in_proj output of an attention layer whose RoPE / bias form the fused epilogue does not cover
  int prefill_mode = 1;     // 1 = batched (MFMA GEMMs + tiled attention), 0 = position by position through the decode kernels
  int lcap = 0;
  GenState* st = nullptr;
  int *remaining = nullptr, *stopping = nullptr;
  int* done_host = nullptr;  // pinned: [0..3] synchronous stop check, [4..7] asynchronous one
  hipEvent_t stop_event = nullptr;
  bool stop_pending = false;
  // per-generation state (host mirror)
  bool gen_active = false;
  bool gen_ended = false;      // zn_gen_end was called: the state stays readable, no further steps until the next zn_gen_begin
  int batch = 0, rows = 0, max_len = 0, t_total = 0, offset0 = 0, max_new = 0;
  float cfg_scale = 2.f;
  zn_sampling sp{};
  std::vector<const void*> kv_layers;
  int *lengths = nullptr, *codes = nullptr;
  int force_eos_step = -1;
  float eos_bias = 0.f;
  unsigned dbg_pause = 0;           // ChainArgs::dbg_pause of this generation's whole-step launches
  int tune[20] = {256, 512, 512, 1024, 512, 512, 2, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // target workgroups: in_proj, out_proj, fc1, fc2, heads; [5] longest context of the fused attention launch (<= 512); [6] > 1: multi-step graphs; [7] > 1: LDS-staged small-M projections; [8] = 2: no persistent chain kernel (five launches per block instead); [9] = 2: fc1's LayerNorm as a launch of its own at 5..16 rows (default: statistics from out_proj's epilogue); [10] = 2: VALU prefill attention; [11] = 2: no in-workgroup-split small-M kernel
  const int* tok_override = nullptr;
  int tok_override_calls = 0;
  hipStream_t cap_stream = nullptr;
  // captured decode steps per attention launch shape (k & 1: 1 = fused single launch, 0 = two passes) and per length
  // (k >> 1: 0 = one step, 1 = ZN_GRAPH_STEPS consecutive steps: fewer graph launches on the chain)
  // ... and, for the whole-step kernel, per key-block count of the attention role: slots 8 + nb (one step) and 8 + 13 + nb (ZN_GRAPH_STEPS steps).
  // A count's graph is captured once per generation and kept to its end: no graph is destroyed while launches of it may still be queued.
#define ZN_NGRAPHS (8 + 2 * (ZN_SK_KB_MAXNB + 1))
  hipGraphExec_t graph_exec[ZN_NGRAPHS] = {};
  hipGraph_t graph[ZN_NGRAPHS] = {};
  bool graph_tried[ZN_NGRAPHS] = {};
  int len_hi = 0;            // host-side upper bound of the rows' KV lengths (keys already cached)
  int attn_fused = 0;        // launch shape of the next run_attention (attn_fused_for)
  std::string err;
};

#define ZN_FAIL(h, code, ...)                                   \
  do {                                                          \
    char _b[1024]; snprintf(_b, sizeof _b, __VA_ARGS__);         \
    if (h) (h)->err = _b; else g_create_err = _b;               \
    return (code);                                              \
  } while (0)
#define HIPCHK(h, call)                                                                       \
  do {                                                                                        \
    hipError_t _e = (call);                                                                   \
    if (_e != hipSuccess) ZN_FAIL(h, ZN_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(_e)); \
  } while (0)

extern "C" int zn_abi_version(void) { return ZN_ABI_VERSION; }

extern "C" const char* zn_last_error(zn_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }

extern "C" size_t zn_kv_bytes_per_layer(const zn_config* c, int32_t rows, int32_t max_len) {
  if (!c || c->n_heads <= 0) return 0;
  const size_t hd = (size_t)c->d_model / c->n_heads;
  return (size_t)rows * max_len * 2 * c->n_heads_kv * hd * 2;
}

extern "C" size_t zn_mamba_state_bytes_per_layer(const zn_config* c, int32_t rows, size_t* conv_bytes) {
  if (conv_bytes) *conv_bytes = 0;
  if (!c || c->arch != 1 || c->m_headdim <= 0 || rows < 0) return 0;
  const size_t conv_dim = (size_t)c->m_d_inner + 2 * (size_t)c->m_ngroups * c->m_d_state;
  const size_t cb = (size_t)rows * conv_dim * c->m_d_conv * 2;
  if (conv_bytes) *conv_bytes = cb;
  return cb + (size_t)rows * c->m_d_inner * c->m_d_state * 2;   // nheads * headdim = d_inner
}

static void free_graph(zn_handle h) {
  for (int k = 0; k < ZN_NGRAPHS; ++k) {
    if (h->graph_exec[k]) { (void)hipGraphExecDestroy(h->graph_exec[k]); h->graph_exec[k] = nullptr; }
    if (h->graph[k]) { (void)hipGraphDestroy(h->graph[k]); h->graph[k] = nullptr; }
    h->graph_tried[k] = false;
  }
}

// The persistent kernels' hand-offs wait on EVERY workgroup of their grid, one per CU: two such launches on one device at the same time
// (two handles generating concurrently, on any streams) could each hold part of the CUs and wait for the rest until the bounded
// waits give up.  One generation per device owns the persistent kernels; a generation that begins while another handle's is active
// on the same device runs the launches path (same results, no in-launch hand-offs).  Single-tenant per PROCESS: other processes
// sharing the GPU are outside this library's reach.
static std::mutex g_tenant_mu;
static const void* g_tenant[64] = {};
extern "C" int zn_tenant_try_claim(int32_t device, const void* owner) {
  if (device < 0 || device >= 64 || !owner) return 0;
  std::lock_guard<std::mutex> lk(g_tenant_mu);
  if (g_tenant[device] && g_tenant[device] != owner) return 0;
  g_tenant[device] = owner;
  return 1;
}
extern "C" int zn_tenant_release(int32_t device, const void* owner) {
  if (device < 0 || device >= 64) return 0;
  std::lock_guard<std::mutex> lk(g_tenant_mu);
  if (g_tenant[device] != owner) return 0;
  g_tenant[device] = nullptr;
  return 1;
}

extern "C" int zn_destroy(zn_handle h) {
  if (!h) return ZN_OK;
  (void)zn_tenant_release(h->device, h);
  free_graph(h);
  void* ptrs[] = {h->emb_tables_dev, h->x, h->q, h->o1, h->mbuf, h->nbuf, h->logits_raw, h->last_logits, h->tok_raw, h->scores, h->cmax, h->pv_part, h->pv_tickets, h->pf_x, h->pf_n, h->pf_qkv, h->pf_a, h->pf_u, h->pf_m, h->pf_res, h->pf_zx, h->pf_xbc, h->pf_y, h->pf_g, h->qkv_tmp, h->fw_lengths, h->st, h->remaining, h->stopping, h->res, h->hn, h->m_zx, h->m_xbc, h->m_y, h->m_g, h->m_vg, h->g16_part, h->g16_tickets, h->ln_part, h->ch_gy1, h->ch_gx1, h->ch_gx2, h->ch_gm, h->ch_epoch, h->ch_x2, h->x_emb, h->tail_ticket, h->ch_gqkv, h->ch_ga, h->ch_gbmax, h->ch_gpart, h->stack_layers, h->ch_diag};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (h->done_host) (void)hipHostFree(h->done_host);
  if (h->stop_event) (void)hipEventDestroy(h->stop_event);
  if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
  delete h;
  return ZN_OK;
}

// ------------------------------------------------------------------------------------------------ persistent chain
#define ZN_CH_GRID 256
#define ZN_CH_DYN_LDS (64 * 1024)     // unused dynamic LDS on top of the static 33 KB: never two workgroups on one CU
template <int NCH, int T_OUT, int T_FC1, int T_FC2, int T_IN>
static bool chain_resident(int n_cus) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chain_kernel<NCH, T_OUT, T_FC1, T_FC2, T_IN>, ZN_CH_THREADS, ZN_CH_DYN_LDS) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return per_cu >= 1 && n_cus >= ZN_CH_GRID;       // every workgroup of the grid is resident at once (the hand-offs wait on all of them)
}
// Which instantiation of chain_kernel serves this model at batch 1 (0 = none: the launches path).
static int chain_variant_for(const zn_config& c) {
  if (c.arch != 0 || !c.double_out_proj || c.d_ff != 4 * c.d_model) return 0;

  int dev = 0, n_cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  const int G = ZN_CH_GRID, hd = c.d_model / c.n_heads, nqkv = (c.n_heads + 2 * c.n_heads_kv) * hd;
  if ((c.d_model / 2) % G || c.d_ff % G || (nqkv / 2) % G || nqkv % 2) return 0;
  const int p_out = c.d_model / 2 / G, p_fc1 = c.d_ff / G, p_in = nqkv / 2 / G;
  if (p_fc1 * 2 > 64 || p_out * 2 > 64 || p_in * 2 > 64) return 0;                 // one communication-wave lane per (unit, row)
  auto fits = [&](int t_out, int t_fc1, int t_fc2, int t_in) {
    return p_out <= ZN_CH_CWAVES * t_out && p_fc1 <= ZN_CH_CWAVES * t_fc1 && p_out <= (ZN_CH_CWAVES / 4) * t_fc2 && p_in <= ZN_CH_CWAVES * t_in;
  };
  // the last block's launch ends with the fused heads (T_IN = 5 / 3 tiles per wave) when they fit its lanes, else after fc2 (T_IN = 0)
  if (c.d_model == 2048 && fits(1, 8, 4, 2) && chain_resident<4, 1, 8, 4, 2>(n_cus) && chain_resident<4, 1, 8, 4, 0>(n_cus) && chain_resident<4, 1, 8, 4, 5>(n_cus)) return 1;
  if (c.d_model == 512 && fits(1, 2, 1, 1) && chain_resident<1, 1, 2, 1, 1>(n_cus) && chain_resident<1, 1, 2, 1, 0>(n_cus) && chain_resident<1, 1, 2, 1, 5>(n_cus)) return 2;
  return 0;
}

extern "C" int zn_create(const zn_config* cfg, const zn_weights* w, int32_t max_rows, zn_handle* out) {
  if (!cfg || !w || !out) ZN_FAIL((zn_handle) nullptr, ZN_ERR_ARG, "zn_create: null argument");
  *out = nullptr;
  const zn_config& c = *cfg;
  if (c.d_model <= 0 || c.n_layer <= 0 || c.n_heads <= 0 || c.n_heads_kv <= 0 || c.d_ff <= 0 || c.n_codebooks <= 0)
    ZN_FAIL((zn_handle) nullptr, ZN_ERR_ARG, "zn_create: non-positive dimension");
  if (c.d_model % c.n_heads || c.n_heads % c.n_heads_kv) ZN_FAIL((zn_handle) nullptr, ZN_ERR_ARG, "zn_create: heads do not divide");
  const int hd = c.d_model / c.n_heads, G = c.n_heads / c.n_heads_kv;
  if (hd != 32 && hd != 64 && hd != 128) ZN_FAIL((zn_handle) nullptr, ZN_ERR_UNSUPPORTED, "head_dim %d not in {32,64,128}", hd);
  if (G != 1 && G != 2 && G != 4 && G != 8) ZN_FAIL((zn_handle) nullptr, ZN_ERR_UNSUPPORTED, "GQA group %d not in {1,2,4,8}", G);
  if (c.d_model % 8 || c.d_ff % 8 || c.d_model > 4096) ZN_FAIL((zn_handle) nullptr, ZN_ERR_UNSUPPORTED, "d_model/d_ff must be multiples of 8, d_model <= 4096");
  if (c.n_codebooks > ZN_FRAME_MAXQ) ZN_FAIL((zn_handle) nullptr, ZN_ERR_UNSUPPORTED, "n_codebooks > %d", ZN_FRAME_MAXQ);
  if (c.vocab_head > ZN_SAMPLE_MAXV) ZN_FAIL((zn_handle) nullptr, ZN_ERR_UNSUPPORTED, "vocab_head > %d", ZN_SAMPLE_MAXV);
  if (max_rows < 2 || max_rows % 2) ZN_FAIL((zn_handle) nullptr, ZN_ERR_ARG, "max_rows must be even and >= 2");
  if (c.arch != 0 && c.arch != 1) ZN_FAIL((zn_handle) nullptr, ZN_ERR_ARG, "zn_create: arch must be 0 (transformer) or 1 (hybrid)");
  if (c.arch == 1) {
    if (c.m_headdim != 64 || (c.m_d_state != 64 && c.m_d_state != 128) || c.m_d_conv != 4)
      ZN_FAIL((zn_handle) nullptr, ZN_ERR_UNSUPPORTED, "Mamba2: headdim %d / d_state %d / d_conv %d not in {64} x {64,128} x {4}", c.m_headdim, c.m_d_state, c.m_d_conv);
    if (c.m_d_inner <= 0 || c.m_d_inner % 64 || c.m_ngroups < 1 || (c.m_d_inner / 64) % c.m_ngroups || (c.m_d_inner / c.m_ngroups) % 8 ||
        c.m_d_inner / c.m_ngroups > 8192)
      ZN_FAIL((zn_handle) nullptr, ZN_ERR_UNSUPPORTED, "Mamba2: d_inner %d / ngroups %d not supported", c.m_d_inner, c.m_ngroups);
  }
  if (c.rope_mode < 0 || c.rope_mode > 2) ZN_FAIL((zn_handle) nullptr, ZN_ERR_ARG, "zn_create: rope_mode must be 0, 1 or 2");
  zn_handle h = new zn_handle_s();
  h->cfg = c; h->max_rows = max_rows; h->hd = hd; h->G = G;
  if (hipGetDevice(&h->device) != hipSuccess) { (void)hipGetLastError(); h->device = 0; }
  if (c.arch == 1) {
    h->m_nheads = c.m_d_inner / c.m_headdim;
    h->m_conv_dim = c.m_d_inner + 2 * c.m_ngroups * c.m_d_state;
    h->m_d_in_proj = 2 * c.m_d_inner + 2 * c.m_ngroups * c.m_d_state + h->m_nheads;
  }
  if (const char* e = getenv("ZN_PREFILL_MODE")) h->prefill_mode = atoi(e);
  h->layers.assign(w->layers, w->layers + c.n_layer);
  h->heads = w->heads; h->norm_f_w = w->norm_f_w; h->norm_f_b = w->norm_f_b; h->rope = w->rope_table;
#define ZC(call) do { hipError_t _e = (call); if (_e != hipSuccess) { g_create_err = std::string(#call) + ": " + hipGetErrorString(_e); zn_destroy(h); return ZN_ERR_HIP; } } while (0)
  ZC(hipMalloc(&h->emb_tables_dev, sizeof(void*) * c.n_codebooks));
  if (w->embeddings) ZC(hipMemcpy(h->emb_tables_dev, w->embeddings, sizeof(void*) * c.n_codebooks, hipMemcpyHostToDevice));
  h->has_io = (w->embeddings != nullptr && w->heads != nullptr);
  const size_t R = max_rows;
  ZC(hipMalloc(&h->x, R * c.d_model * 2));
  ZC(hipMalloc(&h->q, R * c.d_model * 2));
  ZC(hipMalloc(&h->o1, R * c.d_model * 2));
  ZC(hipMalloc(&h->mbuf, R * c.d_ff * 2));
  ZC(hipMalloc(&h->nbuf, R * c.d_model * 2));
  ZC(hipMalloc(&h->logits_raw, R * c.n_codebooks * c.vocab_head * sizeof(float)));
  ZC(hipMalloc(&h->last_logits, (R / 2) * c.n_codebooks * c.vocab_head * sizeof(float)));
  ZC(hipMalloc(&h->tok_raw, (R / 2) * c.n_codebooks * sizeof(int)));
  ZC(hipMalloc(&h->fw_lengths, R * sizeof(int)));
  ZC(hipMalloc(&h->st, sizeof(GenState)));
  ZC(hipMemset(h->st, 0, sizeof(GenState)));
  ZC(hipMalloc(&h->remaining, (R / 2) * sizeof(int)));
  ZC(hipMalloc(&h->stopping, (R / 2) * sizeof(int)));
  ZC(hipHostMalloc(&h->done_host, sizeof(int) * 8));
  ZC(hipEventCreateWithFlags(&h->stop_event, hipEventDisableTiming));
  for (unsigned long long** g : {&h->ch_gy1, &h->ch_gx1, &h->ch_gx2}) {
    ZC(hipMalloc(g, R * (c.d_model / 2) * 8));
    ZC(hipMemset(*g, 0, R * (c.d_model / 2) * 8));             // tag 0 is never a launch's epoch
  }
  { const size_t nqkv = (size_t)(c.n_heads + 2 * c.n_heads_kv) * hd;
    ZC(hipMalloc(&h->ch_gqkv, R * (nqkv / 2 + 1) * 8)); ZC(hipMemset(h->ch_gqkv, 0, R * (nqkv / 2 + 1) * 8));
    ZC(hipMalloc(&h->ch_ga, R * (c.d_model / 2) * 8)); ZC(hipMemset(h->ch_ga, 0, R * (c.d_model / 2) * 8));
    { const size_t nbm = (size_t)2 * c.n_heads_kv * ZN_SK_KB_MAXNB * 4 * 8, npt = (size_t)2 * c.n_heads_kv * ZN_SK_KB_MAXNB * ZN_SK_KB_PSZ * 8;
      ZC(hipMalloc(&h->ch_gbmax, nbm)); ZC(hipMemset(h->ch_gbmax, 0, nbm));
      ZC(hipMalloc(&h->ch_gpart, npt)); ZC(hipMemset(h->ch_gpart, 0, npt)); }
    ZC(hipMalloc(&h->stack_layers, (size_t)c.n_layer * sizeof(StackLayer))); }
  ZC(hipMalloc(&h->ch_gm, R * (c.d_ff / 2 + 1) * 8));
  ZC(hipMemset(h->ch_gm, 0, R * (c.d_ff / 2 + 1) * 8));
  ZC(hipMalloc(&h->ch_epoch, 64));
  ZC(hipMalloc(&h->ch_diag, 64));
  ZC(hipMemset(h->ch_diag, 0, 64));
  { const unsigned one = 1; ZC(hipMemcpy(h->ch_epoch, &one, sizeof one, hipMemcpyHostToDevice)); }
  ZC(hipMalloc(&h->ch_x2, R * c.d_model * 2));
  h->ch_variant = chain_variant_for(c);
  if (const char* e = getenv("ZN_CHAIN")) if (atoi(e) == 0) h->tune[8] = 2;
  if (const char* e = getenv("ZN_STACK")) if (atoi(e) == 0) h->tune[15] = 2;
  {   // split-K partial tiles + tickets of the small-M projections (batches of 3..8 utterances; short-prompt prefill at any batch)
    ZC(hipMalloc(&h->x_emb, R * c.d_model * 2));
    ZC(hipMalloc(&h->tail_ticket, sizeof(int)));
    ZC(hipMemset(h->tail_ticket, 0, sizeof(int)));
    ZC(hipMalloc(&h->g16_part, ZN_G16_PART_BYTES));
    ZC(hipMalloc(&h->g16_tickets, ZN_G16_MAX_GROUPS * sizeof(int)));
    ZC(hipMemset(h->g16_tickets, 0, ZN_G16_MAX_GROUPS * sizeof(int)));
    h->g16_part_bytes = ZN_G16_PART_BYTES;
    ZC(hipMalloc(&h->ln_part, R * ZN_G16_LNT * 2 * sizeof(float)));
  }
  if (c.arch == 1) {
    ZC(hipMalloc(&h->res, R * c.d_model * 4));     // bf16, or fp32 with residual_in_fp32
    ZC(hipMalloc(&h->qkv_tmp, R * (size_t)(c.n_heads + 2 * c.n_heads_kv) * hd * 2));
    ZC(hipMalloc(&h->hn, R * c.d_model * 2));
    ZC(hipMalloc(&h->m_zx, R * h->m_d_in_proj * 2));
    ZC(hipMalloc(&h->m_xbc, R * h->m_conv_dim * 2));
    ZC(hipMalloc(&h->m_y, R * c.m_d_inner * 2));
    ZC(hipMalloc(&h->m_vg, R * c.m_d_inner * sizeof(float)));
    ZC(hipMalloc(&h->m_g, R * c.m_d_inner * 2));
  }
#undef ZC
  *out = h;
  return ZN_OK;
}

// ------------------------------------------------------------------------------------------------ launches
template <int R, int NCH, int KS, int PRO, int EPI>
static void launch_gemv_t(const GemvArgs& a, int blocks, bool full, hipStream_t s) {
  if (full) hipLaunchKernelGGL((gemv_kernel<R, NCH, KS, PRO, EPI, true>), dim3(blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((gemv_kernel<R, NCH, KS, PRO, EPI, false>), dim3(blocks), dim3(256), 0, s, a);
}
template <int R, int KS, int PRO, int EPI>
static int launch_gemv_nch(const GemvArgs& a, int nch, int blocks, bool full, hipStream_t s) {
  switch (nch) {
    case 1: launch_gemv_t<R, 1, KS, PRO, EPI>(a, blocks, full, s); return 0;
    case 2: launch_gemv_t<R, 2, KS, PRO, EPI>(a, blocks, full, s); return 0;
    case 4: launch_gemv_t<R, 4, KS, PRO, EPI>(a, blocks, full, s); return 0;
    case 8: launch_gemv_t<R, 8, KS, PRO, EPI>(a, blocks, full, s); return 0;
  }
  return -1;
}
template <int PRO, int EPI>
static int launch_gemv_rows(const GemvArgs& a, int rgroup, int ks, int nch, int blocks, bool full, hipStream_t s) {
  if (ks == 1) {
    if (rgroup <= 2) return launch_gemv_nch<2, 1, PRO, EPI>(a, nch, blocks, full && rgroup == 2, s);
    return launch_gemv_nch<4, 1, PRO, EPI>(a, nch, blocks, full && rgroup == 4, s);
  }
  if constexpr (PRO == PRO_NONE || PRO == PRO_GATED) {
    if (rgroup <= 2) return launch_gemv_nch<2, 4, PRO, EPI>(a, nch, blocks, full && rgroup == 2, s);
    return launch_gemv_nch<4, 4, PRO, EPI>(a, nch, blocks, full && rgroup == 4, s);
  }
  return -1;
}

// Runs one fused GEMV over `rows` activation rows (groups of <= 4 rows per launch; weights are re-streamed per
// group, so batches beyond 2 utterances pay extra HBM traffic until the MFMA small-M path lands).
// rows in (4, 16]: one weight pass on the matrix cores (gemm16_kernel); LayerNorm, when fused in the GEMV, is a row-wise
// launch here (amortised over the batch).
// rows in (4, 16], K a multiple of 256: the LDS-staged kernel (coalesced weight stream); K is split over workgroups until
// the grid has >= 512 of them, the partial tiles meet in a scratch buffer (allocated by zn_create when max_rows > 4).
template <int EPI, bool LNP = false>
static bool run_gemm16s(zn_handle h, GemvArgs g, hipStream_t s) {
  const int K = g.K;
  if (K % ZN_G16_KC || !h->g16_part) return false;
  if (LNP && (K != 16 * ZN_G16_LNT || !g.ln_part_in)) return false;
  const int nrows_w = (EPI == EPI_SILU) ? g.N / 2 : g.N;          // weight rows that define the grid
  const int per64 = (EPI == EPI_SILU) ? 32 : 64;
  // 64-row workgroups when that already gives >= 512 of them, else 32-row ones, else split K as well
  int nwv = 4, groups = (nrows_w + per64 - 1) / per64;
  if (groups < 512) { nwv = 2; groups = (nrows_w + per64 / 2 - 1) / (per64 / 2); }
  if (groups > ZN_G16_MAX_GROUPS) return false;
  int ks = 1;
  while (groups * ks < 448 && ks < 16 && K % (2 * ks * ZN_G16_KC) == 0) ks *= 2;
  // (fc2 at 16 rows, K = 8192 over 64 groups: 8 slices; 4: 1.690, 8: 1.681, 16: 1.746 ms per batch-8 step)
  // (the Mamba2 in_proj, N = 8512 over 266 groups: 1 / 2 (default) / 4 slices: 1.933 / 1.937 / 1.945 ms per batch-8 hybrid step)
  if (ks > 1 && K / ks < 512) {
    // short slices: the combine costs more than the direct-fragment kernel's access pattern, unless a shallower split
    // still fills the chip (in_proj, N = 3072: 96 groups x 4 slices of 512)
    ks /= 2;
    if (ks < 2 || K / ks < 512 || groups * ks < 320) return false;
  }
  if ((size_t)ks * 16 * groups * nwv * 16 * sizeof(float) > h->g16_part_bytes) return false;
  g.part = h->g16_part; g.tickets = h->g16_tickets; g.ksplit = ks;
  if (nwv == 4) hipLaunchKernelGGL((gemm16s_kernel<EPI, 4, LNP>), dim3(groups, ks), dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm16s_kernel<EPI, 2, LNP>), dim3(groups, ks), dim3(128), 0, s, g);
  return true;
}

// rows in (4, 16], K = 8 waves x 128 x {2, 4} and few weight rows (the LDS-staged kernel would have to split K over
// workgroups): one 16-row tile per workgroup, K split over its waves, no cross-workgroup combine.  tune[11] = 2 disables.
static bool gemm16k_fits(zn_handle h, int epi, int N, int K) {
  if (epi == EPI_SILU || h->tune[7] <= 1 || h->tune[11] == 2) return false;
  const int per = ZN_G16K_NKW * ZN_G16K_KCH, nch = K / per;
  if (K % per || (nch != 2 && nch != 4)) return false;
  return (N + 15) / 16 < (h->tune[13] > 0 ? h->tune[13] : 1024);     // many rows: the 64-row workgroups fill the chip without a split (tune[13]: the tile count from which they take over)
}
template <int PRO, int EPI>
static void run_gemm16k(const GemvArgs& g, hipStream_t s) {
  if constexpr (EPI != EPI_SILU) {
    const int tiles = (g.N + 15) / 16;
    if (g.K / (ZN_G16K_NKW * ZN_G16K_KCH) == 2) hipLaunchKernelGGL((gemm16k_kernel<EPI, 2, PRO>), dim3(tiles), dim3(ZN_G16K_NKW * 64), 0, s, g);
    else if constexpr (PRO == PRO_NONE) hipLaunchKernelGGL((gemm16k_kernel<EPI, 4, PRO>), dim3(tiles), dim3(ZN_G16K_NKW * 64), 0, s, g);   // (the LayerNorm prologue at K = 4096 would not fit the register file: the caller keeps it a launch)
  }
}

template <int PRO, int EPI>
static int run_gemm16(zn_handle h, GemvArgs a, int rows, hipStream_t s) {
  if constexpr (PRO == PRO_GATED) ZN_FAIL(h, ZN_ERR_UNSUPPORTED, "gated-norm prologue: at most 4 rows (got %d)", rows);
  else {
  const int K = a.K;
  // few weight-row tiles (N = d_model: 128 workgroups) -> more waves per workgroup so that every CU still keeps enough
  // loads in flight
  const int tiles_n = (EPI == EPI_SILU) ? (a.N / 2) / 16 : (a.N + 15) / 16;
  int nw = 4;
  if (tiles_n <= 256 && K % 256 == 0) nw = 8;
  if (tiles_n <= 256 && K >= 8192 && K % 512 == 0) nw = 16;
  if (K % (nw * 32)) ZN_FAIL(h, ZN_ERR_UNSUPPORTED, "gemm16: K=%d not a multiple of %d", K, nw * 32);
  if (EPI == EPI_SILU && (a.N / 2) % 16) ZN_FAIL(h, ZN_ERR_UNSUPPORTED, "gemm16: d_ff must be a multiple of 16");
  const int tiles = (EPI == EPI_SILU) ? (a.N / 2) / 16 : (a.N + 15) / 16;
  for (int r0 = 0; r0 < rows; r0 += 16) {
    GemvArgs g = a;
    const int nr = rows - r0 < 16 ? rows - r0 : 16;
    g.nrows = nr;
    const bool k16 = gemm16k_fits(h, EPI, a.N, K);
    // that kernel normalises its rows itself when at most one workgroup per CU repeats the statistics (in_proj: 8.7 us vs
    // 4.8 + 6.2; the heads' 577 tiles: 21.8 vs 5.0 + 11.9)
    const bool k16_ln = k16 && PRO == PRO_LN && tiles <= 256 && K == 2 * ZN_G16K_NKW * ZN_G16K_KCH;
    if constexpr (PRO == PRO_LN && EPI == EPI_SILU) {
      // the producer of these rows left LayerNorm statistics per 16-column tile (layer_post_attention): fc1 normalises while it stages
      if (a.ln_part_in && !k16 && h->tune[7] > 1) {
        GemvArgs gl = a;
        gl.nrows = nr; gl.x = a.x + (size_t)r0 * K; gl.ln_part_in = a.ln_part_in + (size_t)r0 * ZN_G16_LNT * 2;
        if (gl.out) gl.out += (size_t)r0 * (a.N / 2);
        if (run_gemm16s<EPI, true>(h, gl, s)) continue;
      }
    }
    if (PRO == PRO_LN && !k16_ln) {
      hipLaunchKernelGGL(layernorm_kernel, dim3(nr), dim3(64), 0, s, a.x + (size_t)r0 * K, a.ln_w, a.ln_b, h->nbuf, K, a.eps);
      g.x = h->nbuf;
    } else g.x = a.x + (size_t)r0 * K;
    if (g.out) g.out += (size_t)r0 * (EPI == EPI_SILU ? a.N / 2 : a.N);
    if (g.resid) g.resid += (size_t)r0 * a.N;
    if (g.out_f32) g.out_f32 += (size_t)r0 * a.N;
    if (g.lengths) g.lengths += r0;
    if (g.q_out) g.q_out += (size_t)r0 * a.n_heads * a.hd;
    if (g.kv) g.kv += (size_t)r0 * a.max_len * 2 * a.n_heads_kv * a.hd;
    if (g.conv_state) { g.conv_state += (size_t)r0 * a.conv_dim * 4; g.xbc += (size_t)r0 * a.conv_dim; }
    if (g.ln_part_out) { if (k16 && EPI == EPI_RESID && a.N == 16 * ZN_G16_LNT) g.ln_part_out += (size_t)r0 * ZN_G16_LNT * 2; else g.ln_part_out = nullptr; }
    if (k16) {
      if (k16_ln) run_gemm16k<PRO, EPI>(g, s); else run_gemm16k<PRO_NONE, EPI>(g, s);
      continue;
    }
    if (h->tune[7] > 1 && run_gemm16s<EPI>(h, g, s)) continue;
    if constexpr (EPI != EPI_SILU) {
      if (tiles <= 192 && a.N % 8 == 0) {   // N = d_model: 8-row tiles so that every CU gets a workgroup
        const int t8 = a.N / 8;
        if (nw == 16) hipLaunchKernelGGL((gemm16_kernel<16, EPI, 8>), dim3(t8), dim3(1024), 0, s, g);
        else if (nw == 8) hipLaunchKernelGGL((gemm16_kernel<8, EPI, 8>), dim3(t8), dim3(512), 0, s, g);
        else hipLaunchKernelGGL((gemm16_kernel<4, EPI, 8>), dim3(t8), dim3(256), 0, s, g);
        continue;
      }
    }
    if (nw == 16) hipLaunchKernelGGL((gemm16_kernel<16, EPI>), dim3(tiles), dim3(1024), 0, s, g);
    else if (nw == 8) hipLaunchKernelGGL((gemm16_kernel<8, EPI>), dim3(tiles), dim3(512), 0, s, g);
    else hipLaunchKernelGGL((gemm16_kernel<4, EPI>), dim3(tiles), dim3(256), 0, s, g);
  }
  return ZN_OK;
  }
}

template <int PRO, int EPI>
static int run_gemv(zn_handle h, GemvArgs a, int rows, int target_blocks, hipStream_t s) {
  if (rows > 4) return run_gemm16<PRO, EPI>(h, a, rows, s);
  const int K = a.K;
  int ks = 1;
  if ((PRO == PRO_NONE || PRO == PRO_GATED) && K >= 4096 && K % 2048 == 0) ks = 4;
  const int kw = K / ks;
  int nch = (kw + 511) / 512;
  nch = nch <= 1 ? 1 : nch <= 2 ? 2 : nch <= 4 ? 4 : nch <= 8 ? 8 : 99;
  if (nch > 8) ZN_FAIL(h, ZN_ERR_UNSUPPORTED, "gemv: K=%d too large for the register-resident activation path", K);
  a.units = (EPI == EPI_SILU) ? a.N / 2 : (a.N + 1) / 2;
  const int lanes_units = (ks == 1) ? 4 : 1;  // units in flight per block
  int upw = (a.units + target_blocks * lanes_units - 1) / (target_blocks * lanes_units);
  if (upw < 1) upw = 1;
  a.upw = upw;
  const int blocks = (a.units + upw * lanes_units - 1) / (upw * lanes_units);
  const bool full = (kw == nch * 512) && (a.N % 2 == 0) && (a.units == blocks * lanes_units * upw);
  for (int r0 = 0; r0 < rows; r0 += 4) {
    GemvArgs g = a;
    const int nr = rows - r0 < 4 ? rows - r0 : 4;
    g.nrows = nr;
    if (g.x) g.x += (size_t)r0 * K;
    if (g.out) g.out += (size_t)r0 * (EPI == EPI_SILU ? a.N / 2 : a.N);
    if (g.resid) g.resid += (size_t)r0 * a.N;
    if (g.out_f32) g.out_f32 += (size_t)r0 * a.N;
    if (g.lengths) g.lengths += r0;
    if (g.q_out) g.q_out += (size_t)r0 * a.n_heads * a.hd;
    if (g.kv) g.kv += (size_t)r0 * a.max_len * 2 * a.n_heads_kv * a.hd;
    if (g.conv_state) { g.conv_state += (size_t)r0 * a.conv_dim * 4; g.xbc += (size_t)r0 * a.conv_dim; }
    if (launch_gemv_rows<PRO, EPI>(g, nr, ks, nch, blocks, full, s) != 0) ZN_FAIL(h, ZN_ERR_UNSUPPORTED, "gemv: no kernel for ks=%d nch=%d", ks, nch);
  }
  return ZN_OK;
}

// ONE arithmetic for the decode attention on every path (launches, per-block chain, whole-step kernels): scores on the matrix cores in
// one summation order, P.V per 512-key block (the reference's block size) on the matrix cores in one order, the blocks' unnormalised partials
// combined by the reference's recurrence acc = acc * f_j + pv_j in block order (attn_block_probs / attn_block_pv, shared by the launches and
// by the attention role of zn_step_kernel.h).  The fused launch (scores + pass 2 + normalisation in one launch) serves contexts of one block:
// tune[5] may lower that limit (tests), never raise it.
#define ZN_AFUSED_LIMIT 512
// Launch shape of the decode attention for contexts of at most keys_upper_bound keys: 1 = the one-launch shape (one block), 0 = scores launch +
// block launch.
static bool attn_split_cols(zn_handle h, int rows) { return h->tune[19] == 2 || (h->tune[19] != 1 && rows > 4); }
static int attn_fused_for(zn_handle h, int keys_upper_bound, int rows) {
  (void)rows;
  const int lim = h->tune[5] < ZN_AFUSED_LIMIT ? h->tune[5] : ZN_AFUSED_LIMIT;
  return keys_upper_bound <= lim ? 1 : 0;
}

#ifndef ZN_ATTN_DSF
#define ZN_ATTN_DSF 4
#endif
#ifndef ZN_ATTN_DSS
#define ZN_ATTN_DSS 2
#endif
template <int HD>
static int launch_attn_g(const AttnArgs& a, int G, dim3 grid, int fused, bool split_cols, hipStream_t s) {
  constexpr int DS = HD == 128 ? ZN_ATTN_DSS : 1;       // value-column parts per (row, kv head[, block]) when split_cols (attn_block_kernel)
  constexpr int DSF = HD == 128 ? ZN_ATTN_DSF : 1;      // ... of the one-launch shape
  switch (G) {
#define ZN_ATTN_CASE(GG) case GG: \
    if (fused) { \
      if (split_cols && DSF > 1) hipLaunchKernelGGL((attn_block_kernel<HD, GG, true, DSF>), dim3(grid.y * grid.z * DSF), dim3(512), 0, s, a); \
      else hipLaunchKernelGGL((attn_block_kernel<HD, GG, true>), dim3(grid.y * grid.z), dim3(512), 0, s, a); \
      return 0; \
    } \
    hipLaunchKernelGGL((attn_scores_kernel<HD, GG>), grid, dim3(256), 0, s, a); \
    if (split_cols && DS > 1) hipLaunchKernelGGL((attn_block_kernel<HD, GG, false, DS>), dim3(grid.y * DS, grid.z * a.nbcap), dim3(512), 0, s, a); \
    else hipLaunchKernelGGL((attn_block_kernel<HD, GG, false>), dim3(grid.y, grid.z * a.nbcap), dim3(512), 0, s, a); \
    return 0;
    ZN_ATTN_CASE(1) ZN_ATTN_CASE(2) ZN_ATTN_CASE(4) ZN_ATTN_CASE(8)
#undef ZN_ATTN_CASE
  }
  return -1;
}

static int ensure_attn_ws(zn_handle h, int max_len) {
  const int lcap = ((max_len + 511) / 512) * 512;
  if (lcap <= h->lcap) return ZN_OK;
  free_graph(h);
  for (float** p : {&h->scores, &h->cmax, &h->pv_part}) if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (h->pv_tickets) { (void)hipFree(h->pv_tickets); h->pv_tickets = nullptr; }
  const size_t RH = (size_t)h->max_rows * h->cfg.n_heads;
  const int nc = lcap / ZN_ACHUNK;
  HIPCHK(h, hipMalloc(&h->scores, RH * lcap * sizeof(float)));
  HIPCHK(h, hipMalloc(&h->cmax, RH * nc * sizeof(float)));
  { const size_t groups = (size_t)h->max_rows * h->cfg.n_heads_kv;
    HIPCHK(h, hipMalloc(&h->pv_part, groups * (lcap / 512) * (size_t)(h->G * h->hd + 4 * h->G) * sizeof(float)));     // (up to four column parts, each with its e sums)
    HIPCHK(h, hipMalloc(&h->pv_tickets, 4 * groups * sizeof(int)));
    HIPCHK(h, hipMemset(h->pv_tickets, 0, 4 * groups * sizeof(int))); }
  h->lcap = lcap;
  return ZN_OK;
}

static int run_attention(zn_handle h, const bf16_t* q, const bf16_t* kv, int max_len, const int* lengths, const int* ext, int ext_scalar,
                         bf16_t* out, int rows, hipStream_t s, unsigned long long* stamps = nullptr) {
  const zn_config& c = h->cfg;
  if (max_len > h->lcap || rows > h->max_rows) ZN_FAIL(h, ZN_ERR_STATE, "attention workspace too small (max_len %d rows %d)", max_len, rows);
  AttnArgs a{};
  a.q = q; a.kv = kv; a.lengths = lengths; a.ext = ext; a.ext_scalar = ext_scalar; a.max_len = max_len;
  a.n_heads = c.n_heads; a.n_heads_kv = c.n_heads_kv; a.lcap = h->lcap; a.scale = (float)(1.0 / std::sqrt((double)h->hd));
  a.scores = h->scores; a.cmax = h->cmax; a.out = out; a.rows = rows; a.stamps = stamps;
  const int hd = h->hd;
  dim3 grid((max_len + ZN_ACHUNK - 1) / ZN_ACHUNK, c.n_heads_kv, rows);
  // one fused launch for contexts of one 512-key block (the caller bounds the context: h->attn_fused); beyond: scores, then the P.V pass
  // with one workgroup per (slice, kv head, row, 512-key block) and the ticketed in-order combine
  const int fused = h->attn_fused;
  a.part = h->pv_part; a.tickets = h->pv_tickets; a.nbcap = (max_len + 511) / 512;
  if (a.nbcap > 32) ZN_FAIL(h, ZN_ERR_ARG, "attention: %d keys of capacity exceed the 32 blocks the split pass combines", max_len);
  // batches of 3..8 utterances: two workgroups per (row, kv head[, block]), each with half of the value columns (tune[19] = 1: never, 2: always)
  const bool sc = attn_split_cols(h, rows);
  int r2 = hd == 128 ? launch_attn_g<128>(a, h->G, grid, fused, sc, s) : hd == 64 ? launch_attn_g<64>(a, h->G, grid, fused, sc, s)
                                                                                 : launch_attn_g<32>(a, h->G, grid, fused, sc, s);
  if (r2) ZN_FAIL(h, ZN_ERR_UNSUPPORTED, "attention: unsupported group %d", h->G);
  return ZN_OK;
}

// LayerNorm -> in_proj -> split -> RoPE(q,k) -> KV append of block `li` (q in h->q, k/v in the cache row lengths[r])
static int layer_in_proj(zn_handle h, int li, bf16_t* x, bf16_t* kv, int max_len, const int* lengths, int rows, hipStream_t s) {
  const zn_config& c = h->cfg;
  const zn_layer_weights& lw = h->layers[li];
  const int d = c.d_model, hd = h->hd, nq = c.n_heads * hd, nkv = c.n_heads_kv * hd;
  GemvArgs a{};
  a.W = (const bf16_t*)lw.in_proj; a.N = nq + 2 * nkv; a.K = d; a.x = x;
  a.ln_w = (const bf16_t*)lw.norm_w; a.ln_b = (const bf16_t*)lw.norm_b; a.eps = c.norm_eps;
  a.lengths = lengths; a.hd = hd; a.n_heads = c.n_heads; a.n_heads_kv = c.n_heads_kv;
  a.q_out = h->q; a.kv = kv; a.rope = h->rope; a.max_len = max_len; a.rope_positions = c.rope_positions;
  return run_gemv<PRO_LN, EPI_ROPE_KV>(h, a, rows, h->tune[0], s);
}

// out_proj (-> out_proj again, _torch.py:419-420) -> residual -> LayerNorm -> fc1 -> y * silu(gate) -> fc2 -> residual
static int layer_post_attention(zn_handle h, int li, bf16_t* x, int rows, hipStream_t s, bf16_t* x1_copy = nullptr) {
  const zn_config& c = h->cfg;
  const zn_layer_weights& lw = h->layers[li];
  const int d = c.d_model, nq = c.n_heads * h->hd;
  int rc;
  // rows 5..16: the projection that completes the residual stream leaves LayerNorm statistics per 16-column tile and fc1 normalises its
  // activation chunks from them - no LayerNorm launch in between (GemvArgs::ln_part_out).  tune[9] = 2: the launch.
  const bool lnp = rows > 4 && h->tune[9] != 2 && h->ln_part && d == 16 * ZN_G16_LNT && gemm16k_fits(h, EPI_RESID, d, nq);
  {
    GemvArgs a{};
    a.W = (const bf16_t*)lw.out_proj; a.N = d; a.K = nq; a.x = h->o1;
    if (c.double_out_proj) {
      a.out = h->q;   // q is dead after attention: reuse it for the intermediate projection
      if ((rc = run_gemv<PRO_NONE, EPI_STORE>(h, a, rows, h->tune[1], s))) return rc;
      GemvArgs b{};
      b.W = (const bf16_t*)lw.out_proj; b.N = d; b.K = nq; b.x = h->q; b.resid = x; b.out = x;
      if (lnp) b.ln_part_out = h->ln_part;
      if ((rc = run_gemv<PRO_NONE, EPI_RESID>(h, b, rows, h->tune[1], s))) return rc;
    } else {
      a.resid = x; a.out = x;
      if (lnp) a.ln_part_out = h->ln_part;
      if ((rc = run_gemv<PRO_NONE, EPI_RESID>(h, a, rows, h->tune[1], s))) return rc;
    }
  }
  if (x1_copy) (void)hipMemcpyAsync(x1_copy, x, (size_t)rows * d * 2, hipMemcpyDeviceToDevice, s);      // diagnostic trace
  {  // LayerNorm -> fc1 -> y * silu(gate)
    GemvArgs a{};
    a.W = (const bf16_t*)lw.fc1; a.N = 2 * c.d_ff; a.K = d; a.x = x;
    a.ln_w = (const bf16_t*)lw.norm2_w; a.ln_b = (const bf16_t*)lw.norm2_b; a.eps = c.norm_eps; a.out = h->mbuf;
    if (lnp) a.ln_part_in = h->ln_part;
    if ((rc = run_gemv<PRO_LN, EPI_SILU>(h, a, rows, h->tune[2], s))) return rc;
  }
  {  // fc2 -> residual
    GemvArgs a{};
    a.W = (const bf16_t*)lw.fc2; a.N = d; a.K = c.d_ff; a.x = h->mbuf; a.resid = x; a.out = x;
    if ((rc = run_gemv<PRO_NONE, EPI_RESID>(h, a, rows, h->tune[3], s))) return rc;
  }
  return ZN_OK;
}

// One decode step of block `li` on x [rows][d] in place (_torch.py:307-328), as launches.
static int layer_decode(zn_handle h, int li, bf16_t* x, bf16_t* kv, int max_len, const int* lengths, const int* ext,
                        int ext_scalar, int rows, hipStream_t s) {
  int rc;
  if ((rc = layer_in_proj(h, li, x, kv, max_len, lengths, rows, s))) return rc;
  // KV-cached GQA attention over keys [0, lengths+1) (_torch.py:413-417) -> attention output in h->o1
  if ((rc = run_attention(h, h->q, kv, max_len, lengths, ext, ext_scalar, h->o1, rows, s))) return rc;
  return layer_post_attention(h, li, x, rows, s);
}

// The persistent chain serves the step when the model fits an instantiation, at batch 1 (two rows), unless switched off
// (zn_debug_tune(8, 2), or ZN_CHAIN=0 in the environment at zn_create): 1.07 vs 1.16 ms per decode step at the Zonos-v0.1 dimensions.
static bool chain_active(zn_handle h, int rows) { return h->ch_variant != 0 && rows == 2 && h->tune[8] != 2 && !h->demoted && h->persist_ok; }

// The chain never updates the residual stream in place (zn_chain_kernel.h): block li reads it from one buffer and leaves it
// in the other.
static bf16_t* chain_x(zn_handle h, int li) { return (li & 1) ? h->ch_x2 : h->x; }

// Post-attention chain of block `li` plus the in_proj of block li + 1 in ONE launch (zn_chain_kernel.h); x = h->x, the
// attention output in h->o1, the next block's q in h->q.
// Heads matrix rows per workgroup fit the chain's op-4 schedule (5 tiles per compute wave, one epilogue lane per (pair, row))?
static bool chain_heads_fit(zn_handle h) {
  const int units = (h->cfg.n_codebooks * h->cfg.vocab_head + 1) / 2, ppw = (units + ZN_CH_GRID - 1) / ZN_CH_GRID;
  return ppw <= ZN_CH_CWAVES * 5 && ppw * 2 <= 64;
}

// with_heads: the last block's launch also applies norm_f + the fused heads (fp32 logits into h->logits_raw)
static int launch_chain(zn_handle h, int li, const std::vector<const void*>& kv_layers, int max_len, const int* lengths, hipStream_t s,
                        const bf16_t* xin = nullptr, bool with_heads = false) {
  const zn_config& c = h->cfg;
  const zn_layer_weights& lw = h->layers[li];
  const bool last = li + 1 >= c.n_layer;
  ChainArgs a{};
  a.W_out = (const bf16_t*)lw.out_proj; a.W_fc1 = (const bf16_t*)lw.fc1; a.W_fc2 = (const bf16_t*)lw.fc2;
  a.ln2_w = (const bf16_t*)lw.norm2_w; a.ln2_b = (const bf16_t*)lw.norm2_b; a.eps = c.norm_eps; a.F = c.d_ff;
  a.a = h->o1; a.xin = xin ? xin : chain_x(h, li); a.xout = chain_x(h, li + 1);
  a.g_y1 = h->ch_gy1; a.g_x1 = h->ch_gx1; a.g_x2 = h->ch_gx2; a.g_m = h->ch_gm;
  a.epoch = h->ch_epoch; a.tmo = &h->st->pad[0]; a.diag = h->ch_diag;
  a.dbg_pause = h->dbg_pause; a.stamp_layer = li;
  a.stamps = h->ch_stamps ? h->ch_stamps + (size_t)li * 32 : nullptr;
  if (!last) {
    const zn_layer_weights& nx = h->layers[li + 1];
    a.W_in = (const bf16_t*)nx.in_proj; a.lnn_w = (const bf16_t*)nx.norm_w; a.lnn_b = (const bf16_t*)nx.norm_b;
    a.nqkv = (c.n_heads + 2 * c.n_heads_kv) * h->hd;
    a.q_out = h->q; a.kv = (bf16_t*)kv_layers[li + 1]; a.rope = h->rope; a.lengths = lengths;
    a.max_len = max_len; a.hd = h->hd; a.n_heads = c.n_heads; a.n_heads_kv = c.n_heads_kv; a.rope_positions = c.rope_positions;
  }
  const bool heads = last && with_heads;
  if (heads) {
    a.W_in = (const bf16_t*)h->heads; a.lnn_w = (const bf16_t*)h->norm_f_w; a.lnn_b = (const bf16_t*)h->norm_f_b;
    a.nqkv = c.n_codebooks * c.vocab_head; a.heads_out = h->logits_raw;
  }
  const dim3 grid(ZN_CH_GRID), block(ZN_CH_THREADS);
  if (h->ch_variant == 1) {
    if (heads) hipLaunchKernelGGL((chain_kernel<4, 1, 8, 4, 5>), grid, block, ZN_CH_DYN_LDS, s, a);
    else if (last) hipLaunchKernelGGL((chain_kernel<4, 1, 8, 4, 0>), grid, block, ZN_CH_DYN_LDS, s, a);
    else hipLaunchKernelGGL((chain_kernel<4, 1, 8, 4, 2>), grid, block, ZN_CH_DYN_LDS, s, a);
  } else {
    if (heads) hipLaunchKernelGGL((chain_kernel<1, 1, 2, 1, 5>), grid, block, ZN_CH_DYN_LDS, s, a);
    else if (last) hipLaunchKernelGGL((chain_kernel<1, 1, 2, 1, 0>), grid, block, ZN_CH_DYN_LDS, s, a);
    else hipLaunchKernelGGL((chain_kernel<1, 1, 2, 1, 1>), grid, block, ZN_CH_DYN_LDS, s, a);
  }
  return ZN_OK;
}

// Whole-step kernel (zn_step_kernel.h): batch 1 at the Zonos-v0.1 shapes: in_proj(0) + ONE launch per decode step.  The default at batch 1;
// zn_debug_tune(15, 2) or ZN_STACK=0 in the environment at zn_create selects one chain launch per block.  Instantiations (static tile schedules
// for the streaming workgroups the attention role leaves):
//   variant 1  <4, 2, 10, 5, 6, 6>   1 .. 6 key blocks  (8 .. 48 attention workgroups: contexts up to 3072 keys)
//   variant 2  <4, 2, 11, 6, 7, 8>   7 .. 8 blocks   (up to 4096 keys)
//   variant 3  <4, 2, 13, 7, 8, 12>  9 .. 12 blocks  (up to 6144 keys); longer contexts take the per-block path
#define ZN_SK_T1 4, 2, 10, 5, 6, 6
#define ZN_SK_T2 4, 2, 11, 6, 7, 8
#define ZN_SK_T3 4, 2, 13, 7, 8, 12
static int stack_variant_of(int mode) { return mode <= 6 ? 1 : mode <= 8 ? 2 : 3; }
template <int NCH, int T_OUT, int T_FC1, int T_FC2, int T_IN, int NBV>
static bool stack_variant_ok(zn_handle h, int natt) {
  const zn_config& c = h->cfg;
  const int nsw = ZN_CH_GRID - natt;
  if (nsw < 64) return false;
  auto most = [&](int units) { return (units + nsw - 1) / nsw; };                     // units of the fullest streaming workgroup
  const int nqkv = (c.n_heads + 2 * c.n_heads_kv) * h->hd;
  const int p_out = most(c.d_model / 2), p_fc1 = 2 * most(c.d_ff / 2), p_qkv = most((nqkv + 1) / 2), p_hd = most((c.n_codebooks * c.vocab_head + 1) / 2);
  if (p_out > ZN_SK_CW * T_OUT || p_out > T_FC2 || p_fc1 > ZN_SK_CW * T_FC1 || p_qkv > ZN_SK_CW * T_IN || p_hd > ZN_SK_CW * T_IN) return false;   // the static schedule
  if (p_qkv > 10) return false;                                                                                                   // the pre-block: five row pairs per helper wave
  if (p_out * 2 > 64 || p_out * 4 > 64 || p_fc1 > 64 || p_qkv * 2 > 64 || p_hd * 2 > 64 || nqkv % 2) return false;              // one epilogue lane per (unit, row); s_res slots
  // every workgroup of the grid must be resident at once (the hand-offs wait on all of them): one per CU by its LDS, no scratch
  const void* fn = (const void*)step_kernel<NCH, T_OUT, T_FC1, T_FC2, T_IN, NBV>;
  int dev = 0, n_cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, ZN_SK_DYN_LDS) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, step_kernel<NCH, T_OUT, T_FC1, T_FC2, T_IN, NBV>, ZN_SK_THREADS, ZN_SK_DYN_LDS) != hipSuccess) { (void)hipGetLastError(); return false; }
  hipFuncAttributes fa{};
  if (hipFuncGetAttributes(&fa, fn) != hipSuccess) { (void)hipGetLastError(); return false; }
  return per_cu >= 1 && n_cus >= ZN_CH_GRID && fa.localSizeBytes == 0;
}
static bool stack_shapes_ok(zn_handle h) {
  const zn_config& c = h->cfg;
  if (h->ch_variant != 1 || h->hd != 128 || c.n_heads_kv < 1 || c.n_heads != 4 * c.n_heads_kv) return false;
  const int npairs = 2 * c.n_heads_kv;
  h->stackv_ok[0] = false;
  h->stackv_ok[1] = stack_variant_ok<ZN_SK_T1>(h, npairs * 6);
  h->stackv_ok[2] = stack_variant_ok<ZN_SK_T2>(h, npairs * 8);
  h->stackv_ok[3] = stack_variant_ok<ZN_SK_T3>(h, npairs * ZN_SK_KB_MAXNB);
  return h->stackv_ok[1];
}
// The key blocks a whole-step launch covering `keys_upper_bound` keys needs: -1 = the whole-step kernel does not serve the step (the
// per-block path does), n >= 1 = 8 n attention workgroups.
static int stack_mode_for(zn_handle h, int rows, int keys_upper_bound) {
  if (!chain_active(h, rows) || h->tune[15] == 2 || !h->stack_ok) return -1;
  const int nb = keys_upper_bound <= 512 ? 1 : (keys_upper_bound + 511) / 512;
  if (nb > ZN_SK_KB_MAXNB) return -1;
  return h->stackv_ok[stack_variant_of(nb)] ? nb : -1;
}
static bool stack_pre(zn_handle h) { return h->tune[18] != 2; }
static int launch_stack(zn_handle h, hipStream_t s) {
  const zn_config& c = h->cfg;
  ChainArgs a{};
  a.eps = c.norm_eps; a.F = c.d_ff; a.nqkv = (c.n_heads + 2 * c.n_heads_kv) * h->hd;
  a.xin = h->x_emb; a.xout = h->x;
  a.g_y1 = h->ch_gy1; a.g_x1 = h->ch_gx1; a.g_x2 = h->ch_gx2; a.g_m = h->ch_gm; a.g_qkv = h->ch_gqkv; a.g_a = h->ch_ga;
  a.g_bmax = h->ch_gbmax; a.g_part = h->ch_gpart;
  a.epoch = h->ch_epoch; a.tmo = &h->st->pad[0]; a.diag = h->ch_diag;
  a.dbg_pause = h->dbg_pause;
  a.stamps = h->ch_stamps; a.stamp_layer = c.n_layer / 2;
  a.rope = h->rope; a.lengths = h->lengths; a.max_len = h->max_len; a.hd = h->hd; a.n_heads = c.n_heads; a.n_heads_kv = c.n_heads_kv;
  a.rope_positions = c.rope_positions;
  a.layers = h->stack_layers; a.n_layer = c.n_layer; a.q0 = h->q; a.scale = (float)(1.0 / std::sqrt((double)h->hd));
  a.heads_rows = c.n_codebooks * c.vocab_head; a.heads_out = h->logits_raw; a.trace = h->dbg_trace;
  const int mode = h->stack_nbk, npairs = 2 * c.n_heads_kv;
  a.natt = npairs * (mode < 1 ? 1 : mode);
  if (stack_pre(h)) {       // block 0's LayerNorm + in_proj + RoPE + KV append inside the launch (zn_debug_tune(18, 2): as a launch before it)
    a.pre_W = (const bf16_t*)h->layers[0].in_proj; a.pre_ln_w = (const bf16_t*)h->layers[0].norm_w; a.pre_ln_b = (const bf16_t*)h->layers[0].norm_b;
    a.pre_kv = (bf16_t*)h->kv_layers[0];
  }
  const dim3 grid(ZN_CH_GRID), block(ZN_SK_THREADS);
  switch (stack_variant_of(mode)) {
    case 1: hipLaunchKernelGGL((step_kernel<ZN_SK_T1>), grid, block, ZN_SK_DYN_LDS, s, a); break;
    case 2: hipLaunchKernelGGL((step_kernel<ZN_SK_T2>), grid, block, ZN_SK_DYN_LDS, s, a); break;
    default: hipLaunchKernelGGL((step_kernel<ZN_SK_T3>), grid, block, ZN_SK_DYN_LDS, s, a); break;
  }
  return ZN_OK;
}
// device table of the whole-step kernel (KV cache pointers of this generation)
static int build_stack_table(zn_handle h) {
  const zn_config& c = h->cfg;
  std::vector<StackLayer> t(c.n_layer);
  for (int li = 0; li < c.n_layer; ++li) {
    const zn_layer_weights& lw = h->layers[li];
    const bool last = li + 1 == c.n_layer;
    StackLayer& L = t[li];
    L.W_out = (const bf16_t*)lw.out_proj; L.W_fc1 = (const bf16_t*)lw.fc1; L.W_fc2 = (const bf16_t*)lw.fc2;
    L.ln2_w = (const bf16_t*)lw.norm2_w; L.ln2_b = (const bf16_t*)lw.norm2_b;
    L.kv = (const bf16_t*)h->kv_layers[li];
    if (last) { L.W_in = (const bf16_t*)h->heads; L.lnn_w = (const bf16_t*)h->norm_f_w; L.lnn_b = (const bf16_t*)h->norm_f_b; L.kv_next = nullptr; }
    else {
      const zn_layer_weights& nx = h->layers[li + 1];
      L.W_in = (const bf16_t*)nx.in_proj; L.lnn_w = (const bf16_t*)nx.norm_w; L.lnn_b = (const bf16_t*)nx.norm_b; L.kv_next = (bf16_t*)h->kv_layers[li + 1];
    }
  }
  HIPCHK(h, hipMemcpy(h->stack_layers, t.data(), t.size() * sizeof(StackLayer), hipMemcpyHostToDevice));
  return ZN_OK;
}

// All blocks of one decode step on h->x (transformer): launches per op, or in_proj(0) + (attention, chain) per block.
// x0 != NULL: the residual stream enters the first block from there (the decode step's embedding buffer) instead of h->x.
// heads_done != NULL: the caller wants the logits too; set when the last block's chain launch has produced them.
static int decode_blocks(zn_handle h, const int* ext, int ext_scalar, hipStream_t s, const bf16_t* x0 = nullptr, bool* heads_done = nullptr) {
  const zn_config& c = h->cfg;
  int rc;
  const size_t tb = (size_t)h->rows * c.d_model * 2;
  const bool chain = chain_active(h, h->rows);
  if (x0 && !chain) ZN_FAIL(h, ZN_ERR_STATE, "decode_blocks: a separate input buffer is the chain path's");
  auto trace = [&](int li) {        // slots per block: x after the block, attention output, q, m (first d values per row pair), x after the attention half
    if (!h->dbg_trace) return;
    (void)hipMemcpyAsync((char*)h->dbg_trace + (size_t)(8 * li) * tb, chain ? chain_x(h, li + 1) : h->x, tb, hipMemcpyDeviceToDevice, s);
    (void)hipMemcpyAsync((char*)h->dbg_trace + (size_t)(8 * li + 1) * tb, h->o1, tb, hipMemcpyDeviceToDevice, s);
    if (!chain) (void)hipMemcpyAsync((char*)h->dbg_trace + (size_t)(8 * li + 3) * tb, h->mbuf, (size_t)h->rows * c.d_ff * 2 <= 4 * tb ? (size_t)h->rows * c.d_ff * 2 : 4 * tb, hipMemcpyDeviceToDevice, s);
  };
  auto trace_q = [&](int li) {      // q of block li, as the attention launch reads it
    if (h->dbg_trace) (void)hipMemcpyAsync((char*)h->dbg_trace + (size_t)(8 * li + 2) * tb, h->q, tb, hipMemcpyDeviceToDevice, s);
  };
  auto trace_x1 = [&]() {};
  (void)trace_x1;
  if (!chain) {
    for (int li = 0; li < c.n_layer; ++li) {
      if ((rc = layer_in_proj(h, li, h->x, (bf16_t*)h->kv_layers[li], h->max_len, h->lengths, h->rows, s))) return rc;
      trace_q(li);
      if ((rc = run_attention(h, h->q, (bf16_t*)h->kv_layers[li], h->max_len, h->lengths, ext, ext_scalar, h->o1, h->rows, s))) return rc;
      if ((rc = layer_post_attention(h, li, h->x, h->rows, s, h->dbg_trace ? (bf16_t*)((char*)h->dbg_trace + (size_t)(8 * li + 7) * tb) : nullptr))) return rc;
      trace(li);
    }
    return ZN_OK;
  }
  if ((rc = layer_in_proj(h, 0, x0 ? const_cast<bf16_t*>(x0) : h->x, (bf16_t*)h->kv_layers[0], h->max_len, h->lengths, h->rows, s))) return rc;
  h->epoch_bound += c.n_layer;                             // one tag per chain launch (counted again by zn_decode_steps: the bound stays an upper bound)
  for (int li = 0; li < c.n_layer; ++li) {
    trace_q(li);
    if ((rc = run_attention(h, h->q, (bf16_t*)h->kv_layers[li], h->max_len, h->lengths, ext, ext_scalar, h->o1, h->rows, s,
                            h->ch_stamps ? h->ch_stamps + (size_t)(c.n_layer + li) * 32 : nullptr))) return rc;
    const bool wh = heads_done && li + 1 == c.n_layer && chain_heads_fit(h);
    if ((rc = launch_chain(h, li, h->kv_layers, h->max_len, h->lengths, s, li == 0 ? x0 : nullptr, wh))) return rc;
    if (wh) *heads_done = true;
    trace(li);
  }
  if (c.n_layer & 1) HIPCHK(h, hipMemcpyAsync(h->x, h->ch_x2, tb, hipMemcpyDeviceToDevice, s));   // odd depth: the stream ends in the second buffer
  return ZN_OK;
}

static int heads_logits(zn_handle h, const bf16_t* x, int rows, hipStream_t s) {
  const zn_config& c = h->cfg;
  GemvArgs a{};
  a.W = (const bf16_t*)h->heads; a.N = c.n_codebooks * c.vocab_head; a.K = c.d_model; a.x = x;
  a.ln_w = (const bf16_t*)h->norm_f_w; a.ln_b = (const bf16_t*)h->norm_f_b; a.eps = c.norm_eps; a.out_f32 = h->logits_raw;
  return run_gemv<PRO_LN, EPI_F32>(h, a, rows, h->tune[4], s);
}

// ------------------------------------------------------------------------------------------------ hybrid backbone
// fused residual add + norm of the mamba_ssm Block over `rows` rows; the norm kind and the residual's type follow the config
static void launch_add_ln(zn_handle h, const bf16_t* hid, bf16_t* res, int has_res, int write_res, const void* w, const void* b, bf16_t* out, int rows,
                          hipStream_t s) {
  AddLnArgs a{};
  a.h = hid; a.res = res; a.w = (const bf16_t*)w; a.b = (const bf16_t*)b; a.out = out; a.d = h->cfg.d_model; a.has_res = has_res; a.write_res = write_res;
  a.eps = h->cfg.norm_eps; a.rms = h->cfg.rms_norm; a.res32 = h->cfg.residual_in_fp32;
  hipLaunchKernelGGL(add_ln_kernel, dim3(rows), dim3(256), 0, s, a);
}

// One token through the Mamba2 mixer of layer li (mamba_ssm Mamba2.step): n [rows][d] normalised -> out [rows][d]
static int mamba_mixer(zn_handle h, int li, const bf16_t* n, void* state, bf16_t* out, int rows, hipStream_t s) {
  const zn_config& c = h->cfg;
  const zn_layer_weights& lw = h->layers[li];
  int rc;
  {
    GemvArgs a{};
    // in_proj with the conv window update + SiLU of its xBC rows in the epilogue (causal_conv1d_update; one launch less)
    a.W = (const bf16_t*)lw.m_in_proj; a.N = h->m_d_in_proj; a.K = c.d_model; a.x = n; a.out = h->m_zx;
    a.conv_state = (bf16_t*)state; a.conv_w = (const bf16_t*)lw.m_conv_w; a.conv_b = (const bf16_t*)lw.m_conv_b; a.xbc = h->m_xbc;
    a.d_inner = c.m_d_inner; a.conv_dim = h->m_conv_dim;
    if ((rc = run_gemv<PRO_NONE, EPI_MAMBA>(h, a, rows, (h->m_d_in_proj / 2 + 3) / 4, s))) return rc;
  }
  size_t conv_bytes = 0;   // the state buffer is laid out for exactly `rows` rows (zn_mamba_state_bytes_per_layer)
  (void)zn_mamba_state_bytes_per_layer(&c, rows, &conv_bytes);
  MambaArgs m{};
  m.zx = h->m_zx; m.conv_state = (bf16_t*)state; m.ssm_state = (bf16_t*)((char*)state + conv_bytes);
  m.conv_w = (const bf16_t*)lw.m_conv_w; m.conv_b = (const bf16_t*)lw.m_conv_b;
  m.dt_bias = (const bf16_t*)lw.m_dt_bias; m.A_log = (const bf16_t*)lw.m_A_log; m.D = (const bf16_t*)lw.m_D;
  m.norm_w = (const bf16_t*)lw.m_norm_w; m.xbc = h->m_xbc; m.y = h->m_y; m.g = h->m_g;
  m.d_inner = c.m_d_inner; m.conv_dim = h->m_conv_dim; m.nheads = h->m_nheads; m.d_state = c.m_d_state; m.ngroups = c.m_ngroups;
  m.d_in_proj = h->m_d_in_proj; m.eps = c.norm_eps;
  m.rows = rows;
  // RMSNormGated: up to 4 rows (the GEMV, whose waves hold whole rows) the gate product leaves the state-update kernel
  // in fp32 and the row statistic + weight run as the prologue of out_proj (one launch less; batch-1 step 1.28 -> 1.17 ms);
  // more rows, or several norm groups: a launch of its own (at 16 rows the prologue's fp32 fragment loads in each of the
  // 128 workgroups cost more than the launch: 1.92 -> 2.14 ms, measured)
  const bool gated_pro = c.m_ngroups == 1 && c.m_d_inner % 8 == 0 && rows <= 4 && c.m_d_inner <= 4096;
  m.vg = gated_pro ? h->m_vg : nullptr;
  if (c.m_d_state == 128) hipLaunchKernelGGL((mamba_ssm_kernel<128, 2>), dim3(h->m_nheads, (rows + 1) / 2), dim3(256), 0, s, m);
  else hipLaunchKernelGGL((mamba_ssm_kernel<64, 2>), dim3(h->m_nheads, (rows + 1) / 2), dim3(256), 0, s, m);
  GemvArgs o{};
  o.W = (const bf16_t*)lw.m_out_proj; o.N = c.d_model; o.K = c.m_d_inner; o.out = out;
  if (gated_pro) {
    o.gv = h->m_vg; o.ln_w = (const bf16_t*)lw.m_norm_w; o.eps = c.norm_eps;
    return run_gemv<PRO_GATED, EPI_STORE>(h, o, rows, h->tune[3], s);
  }
  hipLaunchKernelGGL(mamba_gated_norm_kernel, dim3(c.m_ngroups, rows), dim3(256), 0, s, m);
  o.x = h->m_g;
  return run_gemv<PRO_NONE, EPI_STORE>(h, o, rows, h->tune[3], s);
}

// One token through hybrid layer li (mamba_ssm Block, fused_add_norm): hidden h->x / residual h->res in, same out.
static int hybrid_layer(zn_handle h, int li, void* cache, int max_len, const int* lengths, int rows, hipStream_t s) {
  const zn_config& c = h->cfg;
  const zn_layer_weights& lw = h->layers[li];
  const int d = c.d_model, hd = h->hd, nq = c.n_heads * hd, nkv = c.n_heads_kv * hd;
  int rc;
  // (the add + LayerNorm as a prologue of the GEMV that consumes it - every workgroup repeating it, the residual
  // ping-ponging between two buffers - was measured slower than this launch: batch-1 step 1.17 -> 1.24 ms)
  launch_add_ln(h, h->x, h->res, li > 0, 1, lw.norm_w, lw.norm_b, h->hn, rows, s);
  if (lw.kind == 1) return mamba_mixer(h, li, h->hn, cache, h->x, rows, s);
  if (c.rope_mode == 0 && !lw.in_proj_bias) {  // MHA: in_proj -> split -> interleaved RoPE(q,k) -> KV append, one launch
    GemvArgs a{};
    a.W = (const bf16_t*)lw.in_proj; a.N = nq + 2 * nkv; a.K = d; a.x = h->hn;
    a.lengths = lengths; a.hd = hd; a.n_heads = c.n_heads; a.n_heads_kv = c.n_heads_kv;
    a.q_out = h->q; a.kv = (bf16_t*)cache; a.rope = h->rope; a.max_len = max_len; a.rope_positions = c.rope_positions;
    if ((rc = run_gemv<PRO_NONE, EPI_ROPE_KV>(h, a, rows, h->tune[0], s))) return rc;
  } else {   // the other attn_cfg forms (half-split or no rotary, qkv bias): projection (+ bias), then rotation + KV append
    GemvArgs a{};
    a.W = (const bf16_t*)lw.in_proj; a.N = nq + 2 * nkv; a.K = d; a.x = h->hn; a.out = h->qkv_tmp; a.bias = (const bf16_t*)lw.in_proj_bias;
    if ((rc = run_gemv<PRO_NONE, EPI_STORE>(h, a, rows, h->tune[0], s))) return rc;
    hipLaunchKernelGGL(rope_kv_any_kernel, dim3(1, rows), dim3(256), 0, s, h->qkv_tmp, h->q, nq, (bf16_t*)cache, h->rope, 1, 0, lengths, max_len,
                       c.n_heads, c.n_heads_kv, hd, c.rope_positions, c.rope_mode);
  }
  if ((rc = run_attention(h, h->q, (const bf16_t*)cache, max_len, lengths, nullptr, 0, h->o1, rows, s))) return rc;
  {
    GemvArgs a{};
    a.W = (const bf16_t*)lw.out_proj; a.N = d; a.K = nq; a.x = h->o1; a.out = h->x; a.bias = (const bf16_t*)lw.out_proj_bias;
    if ((rc = run_gemv<PRO_NONE, EPI_STORE>(h, a, rows, h->tune[1], s))) return rc;
  }
  launch_add_ln(h, h->x, h->res, 1, 1, lw.norm2_w, lw.norm2_b, h->hn, rows, s);
  {
    GemvArgs a{};
    a.W = (const bf16_t*)lw.fc1; a.N = 2 * c.d_ff; a.K = d; a.x = h->hn; a.out = h->mbuf;
    if ((rc = run_gemv<PRO_NONE, EPI_SILU>(h, a, rows, h->tune[2], s))) return rc;
  }
  {
    GemvArgs a{};
    a.W = (const bf16_t*)lw.fc2; a.N = d; a.K = c.d_ff; a.x = h->mbuf; a.out = h->x;
    if ((rc = run_gemv<PRO_NONE, EPI_STORE>(h, a, rows, h->tune[3], s))) return rc;
  }
  return ZN_OK;
}

// final add + norm (_mamba_ssm.py:111-119) of the token in h->x / h->res, then the heads
static int hybrid_heads(zn_handle h, hipStream_t s);

// All layers for the token in h->x, then the final add + LayerNorm and the heads.
static int hybrid_token(zn_handle h, bool want_logits, hipStream_t s) {
  const zn_config& c = h->cfg;
  int rc;
  for (int li = 0; li < c.n_layer; ++li)
    if ((rc = hybrid_layer(h, li, (void*)h->kv_layers[li], h->max_len, h->lengths, h->rows, s))) return rc;
  if (!want_logits) return ZN_OK;
  return hybrid_heads(h, s);
}

static int hybrid_heads(zn_handle h, hipStream_t s) {
  const zn_config& c = h->cfg;
  launch_add_ln(h, h->x, h->res, 1, 0, h->norm_f_w, h->norm_f_b, h->hn, h->rows, s);
  GemvArgs a{};
  a.W = (const bf16_t*)h->heads; a.N = c.n_codebooks * c.vocab_head; a.K = c.d_model; a.x = h->hn; a.out_f32 = h->logits_raw;
  return run_gemv<PRO_NONE, EPI_F32>(h, a, h->rows, h->tune[4], s);
}

static SampleArgs make_sample_args(zn_handle h, const zn_sampling& sp) {
  SampleArgs a{};
  const zn_config& c = h->cfg;
  a.n_q = c.n_codebooks; a.V = c.vocab_head; a.eos_id = c.eos_id;
  a.temperature = sp.temperature; a.top_p = sp.top_p; a.top_k = sp.top_k; a.min_p = sp.min_p;
  a.linear = sp.linear; a.conf = sp.conf; a.quad = sp.quad;
  a.penalty = sp.repetition_penalty; a.pen_window = sp.repetition_penalty_window;
  a.seed = sp.seed;
  return a;
}

static EmbedArgs make_embed_args(zn_handle h) {
  const zn_config& c = h->cfg;
  EmbedArgs e{};
  e.tables = h->emb_tables_dev; e.codes = h->codes; e.col_dev = &h->st->offset; e.sb = c.n_codebooks * h->t_total; e.si = h->t_total;
  e.col = 0; e.n_q = c.n_codebooks; e.d = c.d_model; e.batch = h->batch; e.vocab_embed = c.vocab_embed; e.out = h->x_emb; e.dup = 1;
  return e;
}

// Batch 1 on the chain path: the step's tail (bookkeeping + next step's embedding) runs in the sampler launch's last workgroup.
// Larger batches keep three launches (embed_kernel spreads the utterances over workgroups; one workgroup embedding 8 utterances
// in turn cost 31 us per step at batch 8), and so does everything off the chain path (its first op reads h->x).
static bool tail_fused(zn_handle h) { return h->cfg.arch == 0 && chain_active(h, h->rows) && h->batch <= 2; }

// One iteration of model.py:467-502: embed -> 26 blocks -> heads -> CFG/bias/penalty/sample -> bookkeeping.  With the fused tail the
// embedding of the current column is already in h->x_emb when the step starts (zn_decode_steps launches embed_kernel before the
// first step of a run; every step's sampler launch leaves the next one's, SampleArgs::ticket).
static int enqueue_step(zn_handle h, hipStream_t s) {
  const zn_config& c = h->cfg;
  int rc;
  const bool fused = tail_fused(h);
  if (!fused) {
    EmbedArgs e = make_embed_args(h);
    e.out = h->x;
    hipLaunchKernelGGL(embed_kernel, dim3(h->batch), dim3(256), 0, s, e);
  }
  if (c.arch == 1) {
    if ((rc = hybrid_token(h, true, s))) return rc;
  } else {
    bool heads_done = false;
    if (h->use_stack) {                                    // every block + the heads in one launch (in_proj of block 0 inside it, or as a launch before it)
      if (!stack_pre(h) && (rc = layer_in_proj(h, 0, h->x_emb, (bf16_t*)h->kv_layers[0], h->max_len, h->lengths, h->rows, s))) return rc;
      if ((rc = launch_stack(h, s))) return rc;
      heads_done = true;
    } else if ((rc = decode_blocks(h, nullptr, 0, s, fused ? h->x_emb : nullptr, &heads_done))) return rc;
    if (!heads_done && (rc = heads_logits(h, h->x, h->rows, s))) return rc;
  }
  SampleArgs a = make_sample_args(h, h->sp);
  a.raw = h->logits_raw; a.mix = 1; a.cfg_scale = h->cfg_scale; a.apply_bias = 1; a.batch = h->batch;
  a.codes = h->codes; a.t_total = h->t_total; a.ctx = h->max_new < 100 ? h->max_new : 100;
  a.use_penalty = (h->sp.repetition_penalty != 1.0f); a.st = h->st; a.logits_out = h->last_logits; a.tokens = h->tok_raw;
  a.draw = 1;
  FrameArgs& f = a.fr;
  f.st = h->st; f.codes = h->codes; f.t_total = h->t_total; f.batch = h->batch; f.n_q = c.n_codebooks; f.eos_id = c.eos_id;
  f.mask_id = c.mask_id; f.tokens = h->tok_raw; f.remaining = h->remaining; f.stopping = h->stopping; f.lengths = h->lengths;
  f.rows = h->rows; f.first = 0; f.override = h->tok_override; f.override_calls = h->tok_override_calls;
  if (fused) { a.ticket = h->tail_ticket; a.em = make_embed_args(h); }
  // batch 1, greedy decoding: sampling, bookkeeping and the next embedding in one workgroup (sample1_kernel: 0.8337 -> 0.8312 ms per step).  With a
  // temperature its one wave per codebook carries 17 exp / log / hash evaluations per lane and the nine ticketed workgroups are ahead again (0.8396 vs
  // 0.8415): they keep those steps.  tune[16] = 2: always the ticketed kernel; 3: the one-workgroup kernel for every parameter set it implements (tests).
  const bool one_wg = fused && h->batch == 1 && h->tune[16] != 2 && (h->tune[16] == 3 || !(h->sp.temperature > 0.f)) && c.vocab_head <= 64 * ZN_S1_IT &&
                      c.n_codebooks <= 16 && !(h->sp.top_p > 0.f) && h->sp.top_k <= 0 && !(h->sp.linear > 0.f) &&
                      (!a.use_penalty || h->sp.repetition_penalty_window <= 16) && h->rows <= 1024;
  if (one_wg) { hipLaunchKernelGGL(sample1_kernel, dim3(1), dim3(1024), 0, s, a); return ZN_OK; }
  hipLaunchKernelGGL(sample_kernel, dim3(c.n_codebooks, h->batch), dim3(256), 0, s, a);
  if (!fused) hipLaunchKernelGGL(frame_update_kernel, dim3(1), dim3(256), 0, s, f);
  return ZN_OK;
}

// ------------------------------------------------------------------------------------------------ generation
extern "C" int zn_gen_begin(zn_handle h, int32_t batch, const void* const* kv_layers_dev, int32_t max_len, int32_t* lengths_dev,
                            int32_t* delayed_codes_dev, int32_t t_total, int32_t offset0, int32_t max_new_tokens, float cfg_scale,
                            const zn_sampling* sp, zn_stream stream) {
  if (!h) return ZN_ERR_ARG;
  if (!kv_layers_dev || !lengths_dev || !delayed_codes_dev || !sp) ZN_FAIL(h, ZN_ERR_ARG, "zn_gen_begin: null argument");
  if (!h->has_io) ZN_FAIL(h, ZN_ERR_STATE, "zn_gen_begin: handle was created without embeddings/heads");
  if (batch < 1 || 2 * batch > h->max_rows) ZN_FAIL(h, ZN_ERR_ARG, "zn_gen_begin: batch %d exceeds max_rows %d / 2", batch, h->max_rows);
  if (max_len < 1 || t_total < 1 || offset0 < 1 || offset0 >= t_total) ZN_FAIL(h, ZN_ERR_ARG, "zn_gen_begin: bad lengths");
  if (cfg_scale == 1.0f) ZN_FAIL(h, ZN_ERR_ARG, "cfg_scale == 1 is not supported (zonos/model.py:399)");
  if (max_len > h->cfg.rope_positions)
    ZN_FAIL(h, ZN_ERR_ARG, "sequence length %d exceeds the %d-position RoPE table (zonos/backbone/_torch.py:206)", max_len, h->cfg.rope_positions);
  if (sp->repetition_penalty_window < 0 || sp->repetition_penalty_window > 64) ZN_FAIL(h, ZN_ERR_ARG, "repetition_penalty_window out of range");
  hipStream_t s = (hipStream_t)stream;
  int rc = ensure_attn_ws(h, max_len);
  if (rc) return rc;
  free_graph(h);
  h->batch = batch; h->rows = 2 * batch; h->max_len = max_len; h->t_total = t_total; h->offset0 = offset0; h->max_new = max_new_tokens;
  h->cfg_scale = cfg_scale; h->sp = *sp;
  h->kv_layers.assign(kv_layers_dev, kv_layers_dev + h->cfg.n_layer);
  h->lengths = lengths_dev; h->codes = delayed_codes_dev;
  // The hand-off tags are 32-bit and advance by n_layer per decode step (~41 hours of continuous batch-1 decoding): long before they
  // can wrap, between two generations, the epoch restarts at 1 over zeroed granule buffers (tag 0 is never a launch's epoch).
  if (h->tune[14] == 7) { h->epoch_bound = 0x70000001ull; h->tune[14] = 0; }   // test hook (zn_debug_tune(14, 7)): behave as if the tags were about to wrap
  if (h->epoch_bound > 0x70000000ull) {
    const size_t R = h->max_rows, nqkv = (size_t)(h->cfg.n_heads + 2 * h->cfg.n_heads_kv) * h->hd;
    for (unsigned long long* g : {h->ch_gy1, h->ch_gx1, h->ch_gx2, h->ch_ga}) HIPCHK(h, hipMemsetAsync(g, 0, R * (h->cfg.d_model / 2) * 8, s));
    HIPCHK(h, hipMemsetAsync(h->ch_gqkv, 0, R * (nqkv / 2 + 1) * 8, s));
    HIPCHK(h, hipMemsetAsync(h->ch_gm, 0, R * (h->cfg.d_ff / 2 + 1) * 8, s));
    HIPCHK(h, hipMemsetAsync(h->ch_gbmax, 0, (size_t)2 * h->cfg.n_heads_kv * ZN_SK_KB_MAXNB * 4 * 8, s));
    HIPCHK(h, hipMemsetAsync(h->ch_gpart, 0, (size_t)2 * h->cfg.n_heads_kv * ZN_SK_KB_MAXNB * ZN_SK_KB_PSZ * 8, s));
    const unsigned one = 1;
    HIPCHK(h, hipMemcpyAsync(h->ch_epoch, &one, sizeof one, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipStreamSynchronize(s));
    h->epoch_bound = 1;
  }
  // A handle demoted by a reported hand-off timeout goes back to the persistent kernels after ZN_REARM_AFTER generations in a row
  // that completed on the launches path (a pause of the device is transient; a demotion for the handle's whole life would cost 25 % of
  // every later utterance on a say-so of one event).
  h->n_generations++;
  if (h->demoted && h->clean_since_demotion >= ZN_REARM_AFTER) { h->demoted = false; h->clean_since_demotion = 0; h->n_rearms++; }
  if (h->demoted && h->cfg.arch == 0 && h->ch_variant != 0 && batch == 1 && h->tune[8] != 2) h->n_fallback_generations++;
  // batch 1 on a model the persistent kernels serve: claim the device for them, or run this generation on the launches path
  h->persist_ok = !(h->cfg.arch == 0 && h->ch_variant != 0 && batch == 1) || zn_tenant_try_claim(h->device, h) != 0;
  if (h->cfg.arch == 0 && h->ch_variant == 1) {
    if (!h->stack_checked) { h->stack_ok = stack_shapes_ok(h); h->stack_checked = true; }
    if (h->stack_ok) { int rc = build_stack_table(h); if (rc) return rc; }
  }
  GenState st{};
  st.offset = offset0; st.step = 0; st.all_done = 0; st.force_eos_step = h->force_eos_step; st.eos_bias = h->eos_bias;
  h->dbg_pause = 0;
  if (h->tune[14] == 11) { h->dbg_pause = 3000000u; h->tune[14] = 0; }   // test hook (zn_debug_tune(14, 11)): every launch of this generation pauses all its waves for 30 ms in block 2
  if (h->tune[14] == 9) { st.pad[0] = 1; h->tune[14] = 0; }   // test hook (zn_debug_tune(14, 9)): this generation's hand-off waits find the timeout word set
  HIPCHK(h, hipMemcpyAsync(h->st, &st, sizeof st, hipMemcpyHostToDevice, s));
  std::vector<int> rem(batch, t_total - offset0), stop(batch, 0);   // model.py:439-441
  HIPCHK(h, hipMemcpyAsync(h->remaining, rem.data(), batch * sizeof(int), hipMemcpyHostToDevice, s));
  HIPCHK(h, hipMemcpyAsync(h->stopping, stop.data(), batch * sizeof(int), hipMemcpyHostToDevice, s));
  std::vector<int> len0(2 * batch, 0);
  HIPCHK(h, hipMemcpyAsync(len0.data(), lengths_dev, 2 * batch * sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));  // host vectors go out of scope
  h->len_hi = 0;
  for (int v : len0) if (v > h->len_hi) h->len_hi = v;
  h->gen_active = true;
  h->gen_ended = false;
  h->gen_timed_out = false;
  h->gen_stream = s;
  h->stop_pending = false;
  h->emb_valid = false;
  return ZN_OK;
}


static int ensure_prefill_ws(zn_handle h, size_t M) {
  if (M <= h->pf_rows) return ZN_OK;
  const zn_config& c = h->cfg;
  for (bf16_t** p : {&h->pf_x, &h->pf_n, &h->pf_qkv, &h->pf_a, &h->pf_u, &h->pf_m, &h->pf_res, &h->pf_zx, &h->pf_xbc, &h->pf_y, &h->pf_g})
    if (*p) { (void)hipFree(*p); *p = nullptr; }
  h->pf_rows = 0;
  const size_t nqkv = (size_t)(c.n_heads + 2 * c.n_heads_kv) * h->hd;
  HIPCHK(h, hipMalloc(&h->pf_x, M * c.d_model * 2));
  HIPCHK(h, hipMalloc(&h->pf_n, M * c.d_model * 2));
  HIPCHK(h, hipMalloc(&h->pf_qkv, M * nqkv * 2));
  HIPCHK(h, hipMalloc(&h->pf_a, M * (size_t)c.n_heads * h->hd * 2));
  HIPCHK(h, hipMalloc(&h->pf_u, M * 2 * (size_t)c.d_ff * 2));
  HIPCHK(h, hipMalloc(&h->pf_m, M * (size_t)c.d_ff * 2));
  if (c.arch == 1) {
    HIPCHK(h, hipMalloc(&h->pf_res, M * c.d_model * 4));
    HIPCHK(h, hipMalloc(&h->pf_zx, M * (size_t)h->m_d_in_proj * 2));
    HIPCHK(h, hipMalloc(&h->pf_xbc, M * (size_t)h->m_conv_dim * 2));
    HIPCHK(h, hipMalloc(&h->pf_y, M * (size_t)c.m_d_inner * 2));
    HIPCHK(h, hipMalloc(&h->pf_g, M * (size_t)c.m_d_inner * 2));
  }
  h->pf_rows = M;
  return ZN_OK;
}

static void launch_gemm(const bf16_t* A, int lda, const bf16_t* W, bf16_t* out, int ldo, const bf16_t* resid, int M, int N, int K, hipStream_t s,
                        const bf16_t* bias = nullptr) {
  GemmArgs g{A, W, out, resid, M, N, K, lda, ldo, bias};
  dim3 grid((N + 127) / 128, (M + 127) / 128);
  if (M >= 256 && K % ZN_PG_KC == 0 && lda % 8 == 0) {   // long prompts: LDS-staged panels (coalesced row pieces)
    if (resid) hipLaunchKernelGGL((gemm_bf16s_kernel<1>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_bf16s_kernel<0>), grid, dim3(256), 0, s, g);
    return;
  }
  if (resid) hipLaunchKernelGGL((gemm_bf16_kernel<1>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_bf16_kernel<0>), grid, dim3(256), 0, s, g);
}

template <int HD>
static int launch_prefill_attn(const PrefillAttnArgs& a, int G, int R, bool mfma, hipStream_t s) {
  if constexpr (HD == 128) {
    if (mfma) {   // both contractions on the matrix cores (same row semantics)
      dim3 grid((a.S + 64 / G - 1) / (64 / G), a.n_heads_kv, R);
      switch (G) {
        case 1: hipLaunchKernelGGL((attn_prefill_mfma_kernel<1>), grid, dim3(256), 0, s, a); return 0;
        case 2: hipLaunchKernelGGL((attn_prefill_mfma_kernel<2>), grid, dim3(256), 0, s, a); return 0;
        case 4: hipLaunchKernelGGL((attn_prefill_mfma_kernel<4>), grid, dim3(256), 0, s, a); return 0;
        case 8: hipLaunchKernelGGL((attn_prefill_mfma_kernel<8>), grid, dim3(256), 0, s, a); return 0;
      }
    }
  }
  switch (G) {
#define ZN_PA(GG) case GG: hipLaunchKernelGGL((attn_prefill_kernel<HD, GG>), dim3((a.S + 64 / GG - 1) / (64 / GG), a.n_heads_kv, R), dim3(256), 0, s, a); return 0;
    ZN_PA(1) ZN_PA(2) ZN_PA(4) ZN_PA(8)
#undef ZN_PA
  }
  return -1;
}

static int qsplit(int S) { return S >= 768 ? 256 : S >= 192 ? 64 : 32; }   // query split of the CPU flash kernel

// All S positions at once: row-wise kernels over M = R*S rows, MFMA GEMMs, tiled exact causal attention.
// causal attention of S prefill positions over the keys already written for them (SDPA is_causal=True, _torch.py:415)
static int prefill_attention(zn_handle h, const bf16_t* q, int ldq, const bf16_t* kv, int max_len, bf16_t* out, int ldo, int S, int R,
                             hipStream_t s, int base = 0) {
  const zn_config& c = h->cfg;
  const int hd = h->hd;
  PrefillAttnArgs pa{};
  pa.q = q; pa.ldq = ldq; pa.kv = kv; pa.out = out; pa.ldo = ldo; pa.S = S; pa.base = base; pa.max_len = max_len;
  pa.n_heads = c.n_heads; pa.n_heads_kv = c.n_heads_kv; pa.qsplit = qsplit(S); pa.scale = (float)(1.0 / std::sqrt((double)hd));
  const bool mfma = h->tune[10] != 2;   // tune[10] = 2: the VALU kernel at every head size
  int r2 = hd == 128 ? launch_prefill_attn<128>(pa, h->G, R, mfma, s) : hd == 64 ? launch_prefill_attn<64>(pa, h->G, R, mfma, s) : launch_prefill_attn<32>(pa, h->G, R, mfma, s);
  if (r2) ZN_FAIL(h, ZN_ERR_UNSUPPORTED, "prefill attention: unsupported group %d", h->G);
  return ZN_OK;
}

// Projection of a short prompt (17..64 rows) through the weight-streaming 64-row kernel; false = shape not served.
template <int EPI>
static bool run_gemm64s(zn_handle h, const bf16_t* x, const void* W, int N, int K, bf16_t* out, const bf16_t* resid, int M, hipStream_t s) {
  if (M > 64 || K % 256 || (EPI == EPI_SILU && (N / 2) % 32) || h->tune[7] <= 1) return false;
  const int groups = (EPI == EPI_SILU) ? (N / 2 + 31) / 32 : (N + 63) / 64;
  if (groups > ZN_G16_MAX_GROUPS) return false;
  int ks = 1;
  while (groups * ks < 256 && ks < 16 && K % (2 * ks * 256) == 0) ks *= 2;
  // (fc2, K = 8192: the split this picks, 8, against 4 / 16 / 2 forced: prefill 2.71 vs 2.90 / 2.77 / 3.13 ms)
  if ((size_t)ks * 64 * groups * 64 * sizeof(float) > h->g16_part_bytes) return false;
  GemvArgs g{};
  g.W = (const bf16_t*)W; g.N = N; g.K = K; g.x = x; g.out = out; g.resid = resid; g.nrows = M;
  g.part = h->g16_part; g.tickets = h->g16_tickets; g.ksplit = ks;
  hipLaunchKernelGGL((gemm64s_kernel<EPI>), dim3(groups, ks), dim3(256), 0, s, g);
  return true;
}

// Short prompts, contractions of 2048 or 4096 with few weight rows (in_proj, both out_proj calls): the decode side's gemm16k_kernel
// (one 16-row weight tile per workgroup over the whole K, K split over its 8 waves: no cross-workgroup combine; LayerNorm as its
// prologue) over ceil(M / 16) row groups in ONE launch, instead of gemm64s_kernel's 8-way split-K with a ticketed combine
// (14.4 us for 8-13 MB) behind a LayerNorm launch.
template <int PRO, int EPI>
static bool run_gemm16k_rows(zn_handle h, const bf16_t* x, const void* ln_w, const void* ln_b, const void* W, int N, int K, bf16_t* out, const bf16_t* resid,
                             int M, hipStream_t s) {
  if (M > 64 || !gemm16k_fits(h, EPI, N, K) || h->tune[12] == 2) return false;
  const int nch = K / (ZN_G16K_NKW * ZN_G16K_KCH);
  if (PRO == PRO_LN && nch != 2) return false;
  GemvArgs g{};
  g.W = (const bf16_t*)W; g.N = N; g.K = K; g.x = x; g.out = out; g.resid = resid; g.nrows = M; g.eps = h->cfg.norm_eps;
  g.ln_w = (const bf16_t*)ln_w; g.ln_b = (const bf16_t*)ln_b;
  const dim3 grid((N + 15) / 16, (M + 15) / 16), block(ZN_G16K_NKW * 64);
  if (nch == 2) hipLaunchKernelGGL((gemm16k_kernel<EPI, 2, PRO>), grid, block, 0, s, g);
  else if constexpr (PRO == PRO_NONE) hipLaunchKernelGGL((gemm16k_kernel<EPI, 4, PRO>), grid, block, 0, s, g);
  return true;
}

// Transformer blocks over all S positions of R rows (hidden [R][S][d]) with `base` keys already cached per row: the
// residual stream of every position ends in h->pf_x.
static int transformer_prefill_core(zn_handle h, const bf16_t* hidden, int S, int R, const void* const* kv_layers, int max_len, int base, hipStream_t s) {
  const zn_config& c = h->cfg;
  const int M = R * S, d = c.d_model, hd = h->hd, nq = c.n_heads * hd, nkv = c.n_heads_kv * hd, nqkv = nq + 2 * nkv, F = c.d_ff;
  int rc = ensure_prefill_ws(h, (size_t)M);
  if (rc) return rc;
  HIPCHK(h, hipMemcpyAsync(h->pf_x, hidden, (size_t)M * d * 2, hipMemcpyDeviceToDevice, s));
  for (int li = 0; li < c.n_layer; ++li) {
    const zn_layer_weights& lw = h->layers[li];
    bf16_t* kv = (bf16_t*)kv_layers[li];
    // short prompts (<= 64 rows): every projection streams its weights through a small-M kernel (10.7 -> ~2.5 ms per prefill)
    // (LayerNorm as gemm16k's prologue over four row groups: 22.0 us against 4.7 + 9 for the launch pair: 768 workgroups repeat the statistics)
    hipLaunchKernelGGL(layernorm_kernel, dim3(M), dim3(64), 0, s, h->pf_x, (const bf16_t*)lw.norm_w, (const bf16_t*)lw.norm_b, h->pf_n, d, c.norm_eps);
    // short prompts: split, RoPE and the KV append in the projection's epilogue (q compact [M][Hq * hd] in pf_qkv), as in a decode step
    int ldq = nqkv;
    if (M <= 64 && gemm16k_fits(h, EPI_ROPE_KV, nqkv, d) && d / (ZN_G16K_NKW * ZN_G16K_KCH) == 2 && h->tune[12] != 2) {
      GemvArgs g{};
      g.W = (const bf16_t*)lw.in_proj; g.N = nqkv; g.K = d; g.x = h->pf_n; g.nrows = M; g.eps = c.norm_eps;
      g.hd = hd; g.n_heads = c.n_heads; g.n_heads_kv = c.n_heads_kv; g.q_out = h->pf_qkv; g.kv = kv; g.rope = h->rope; g.max_len = max_len;
      g.rope_positions = c.rope_positions; g.pf_S = S; g.pf_base = base;
      hipLaunchKernelGGL((gemm16k_kernel<EPI_ROPE_KV, 2, PRO_NONE>), dim3((nqkv + 15) / 16, (M + 15) / 16), dim3(ZN_G16K_NKW * 64), 0, s, g);
      ldq = nq;
    } else {
      if (!run_gemm16k_rows<PRO_NONE, EPI_STORE>(h, h->pf_n, nullptr, nullptr, lw.in_proj, nqkv, d, h->pf_qkv, nullptr, M, s) &&
          !run_gemm64s<EPI_STORE>(h, h->pf_n, lw.in_proj, nqkv, d, h->pf_qkv, nullptr, M, s))
        launch_gemm(h->pf_n, d, (const bf16_t*)lw.in_proj, h->pf_qkv, nqkv, nullptr, M, nqkv, d, s);
      hipLaunchKernelGGL(rope_kv_rows_kernel, dim3(S, R), dim3(256), 0, s, h->pf_qkv, kv, h->rope, S, base, max_len, c.n_heads, c.n_heads_kv, hd, c.rope_positions);
    }
    rc = prefill_attention(h, h->pf_qkv, ldq, kv, max_len, h->pf_a, nq, S, R, s, base);
    if (rc) return rc;
    if (c.double_out_proj) {
      if (!run_gemm16k_rows<PRO_NONE, EPI_STORE>(h, h->pf_a, nullptr, nullptr, lw.out_proj, d, nq, h->pf_n, nullptr, M, s) &&
          !run_gemm64s<EPI_STORE>(h, h->pf_a, lw.out_proj, d, nq, h->pf_n, nullptr, M, s))
        launch_gemm(h->pf_a, nq, (const bf16_t*)lw.out_proj, h->pf_n, d, nullptr, M, d, nq, s);
      if (!run_gemm16k_rows<PRO_NONE, EPI_RESID>(h, h->pf_n, nullptr, nullptr, lw.out_proj, d, nq, h->pf_x, h->pf_x, M, s) &&
          !run_gemm64s<EPI_RESID>(h, h->pf_n, lw.out_proj, d, nq, h->pf_x, h->pf_x, M, s))
        launch_gemm(h->pf_n, d, (const bf16_t*)lw.out_proj, h->pf_x, d, h->pf_x, M, d, nq, s);
    } else if (!run_gemm16k_rows<PRO_NONE, EPI_RESID>(h, h->pf_a, nullptr, nullptr, lw.out_proj, d, nq, h->pf_x, h->pf_x, M, s) &&
               !run_gemm64s<EPI_RESID>(h, h->pf_a, lw.out_proj, d, nq, h->pf_x, h->pf_x, M, s))
      launch_gemm(h->pf_a, nq, (const bf16_t*)lw.out_proj, h->pf_x, d, h->pf_x, M, d, nq, s);
    hipLaunchKernelGGL(layernorm_kernel, dim3(M), dim3(64), 0, s, h->pf_x, (const bf16_t*)lw.norm2_w, (const bf16_t*)lw.norm2_b, h->pf_n, d, c.norm_eps);
    if (!run_gemm64s<EPI_SILU>(h, h->pf_n, lw.fc1, 2 * F, d, h->pf_m, nullptr, M, s)) {
      launch_gemm(h->pf_n, d, (const bf16_t*)lw.fc1, h->pf_u, 2 * F, nullptr, M, 2 * F, d, s);
      hipLaunchKernelGGL(silu_mul_rows_kernel, dim3(M), dim3(256), 0, s, h->pf_u, h->pf_m, F);
    }
    if (!run_gemm64s<EPI_RESID>(h, h->pf_m, lw.fc2, d, F, h->pf_x, h->pf_x, M, s))
      launch_gemm(h->pf_m, F, (const bf16_t*)lw.fc2, h->pf_x, d, h->pf_x, M, d, F, s);
  }
  return ZN_OK;
}

static int prefill_batched(zn_handle h, const bf16_t* hidden, int S, hipStream_t s) {
  const zn_config& c = h->cfg;
  const int R = h->rows, d = c.d_model;
  int rc = transformer_prefill_core(h, hidden, S, R, h->kv_layers.data(), h->max_len, 0, s);
  if (rc) return rc;
  hipLaunchKernelGGL(gather_last_kernel, dim3(R), dim3(256), 0, s, h->pf_x, h->x, S, d);
  hipLaunchKernelGGL(add_lengths_kernel, dim3(1), dim3(64 > R ? 64 : R), 0, s, h->lengths, R, S);
  return ZN_OK;
}

// A projection over M rows of the hybrid prefill: out [M][N] = A [M][K] W^T (+ bias).  exact: row by row through the decode
// step's GEMV (same summation order as a single step: lets tests compare a prefill with single steps bit for bit).
static int proj_rows(zn_handle h, const bf16_t* A, int lda, const void* W, const void* bias, bf16_t* out, int ldo, int M, int N, int K, bool exact,
                     hipStream_t s) {
  if (exact) {
    for (int m = 0; m < M; ++m) {
      GemvArgs g{};
      g.W = (const bf16_t*)W; g.N = N; g.K = K; g.x = A + (size_t)m * lda; g.out = out + (size_t)m * ldo; g.bias = (const bf16_t*)bias;
      int rc = run_gemv<PRO_NONE, EPI_STORE>(h, g, 1, h->tune[1], s);
      if (rc) return rc;
    }
    return ZN_OK;
  }
  launch_gemm(A, lda, (const bf16_t*)W, out, ldo, nullptr, M, N, K, s, (const bf16_t*)bias);
  return ZN_OK;
}

// Hybrid blocks over all S positions of R rows (mamba_ssm Block semantics): Mamba2 layers run the sequence conv and the
// selective scan (zn_mamba_kernels.h), attention layers the batched projections + tiled causal attention.  The mixer
// output of every position ends in h->pf_x, the residual stream in h->pf_res; caches advance by S positions.
static int hybrid_prefill_core(zn_handle h, const bf16_t* hidden, int S, int R, const void* const* caches, int max_len, int base, bool exact,
                               hipStream_t s) {
  const zn_config& c = h->cfg;
  const int M = R * S, d = c.d_model, hd = h->hd, nq = c.n_heads * hd, nkv = c.n_heads_kv * hd, nqkv = nq + 2 * nkv, F = c.d_ff;
  int rc = ensure_prefill_ws(h, (size_t)M);
  if (rc) return rc;
  HIPCHK(h, hipMemcpyAsync(h->pf_x, hidden, (size_t)M * d * 2, hipMemcpyDeviceToDevice, s));
  for (int li = 0; li < c.n_layer; ++li) {
    const zn_layer_weights& lw = h->layers[li];
    launch_add_ln(h, h->pf_x, h->pf_res, li > 0, 1, lw.norm_w, lw.norm_b, h->pf_n, M, s);
    if (lw.kind == 1) {
      if ((rc = proj_rows(h, h->pf_n, d, lw.m_in_proj, nullptr, h->pf_zx, h->m_d_in_proj, M, h->m_d_in_proj, d, exact, s))) return rc;
      size_t conv_bytes = 0;
      (void)zn_mamba_state_bytes_per_layer(&c, R, &conv_bytes);
      MambaArgs m{};
      m.zx = h->pf_zx; m.conv_state = (bf16_t*)caches[li]; m.ssm_state = (bf16_t*)((char*)caches[li] + conv_bytes);
      m.conv_w = (const bf16_t*)lw.m_conv_w; m.conv_b = (const bf16_t*)lw.m_conv_b;
      m.dt_bias = (const bf16_t*)lw.m_dt_bias; m.A_log = (const bf16_t*)lw.m_A_log; m.D = (const bf16_t*)lw.m_D;
      m.norm_w = (const bf16_t*)lw.m_norm_w; m.xbc = h->pf_xbc; m.y = h->pf_y; m.g = h->pf_g; m.vg = nullptr;
      m.d_inner = c.m_d_inner; m.conv_dim = h->m_conv_dim; m.nheads = h->m_nheads; m.d_state = c.m_d_state; m.ngroups = c.m_ngroups;
      m.d_in_proj = h->m_d_in_proj; m.eps = c.norm_eps; m.rows = R;
      hipLaunchKernelGGL(mamba_conv_seq_kernel, dim3((h->m_conv_dim + 255) / 256, R), dim3(256), 0, s, m, S);
      if (c.m_d_state == 128) hipLaunchKernelGGL((mamba_scan_kernel<128>), dim3(h->m_nheads, R), dim3(256), 0, s, m, S);
      else hipLaunchKernelGGL((mamba_scan_kernel<64>), dim3(h->m_nheads, R), dim3(256), 0, s, m, S);
      m.rows = M;
      hipLaunchKernelGGL(mamba_gated_norm_kernel, dim3(c.m_ngroups, M), dim3(256), 0, s, m);
      if ((rc = proj_rows(h, h->pf_g, c.m_d_inner, lw.m_out_proj, nullptr, h->pf_x, d, M, d, c.m_d_inner, exact, s))) return rc;
      continue;
    }
    bf16_t* kv = (bf16_t*)caches[li];
    if ((rc = proj_rows(h, h->pf_n, d, lw.in_proj, lw.in_proj_bias, h->pf_qkv, nqkv, M, nqkv, d, exact, s))) return rc;
    hipLaunchKernelGGL(rope_kv_any_kernel, dim3(S, R), dim3(256), 0, s, h->pf_qkv, h->pf_qkv, nqkv, kv, h->rope, S, base, (const int*)nullptr, max_len,
                       c.n_heads, c.n_heads_kv, hd, c.rope_positions, c.rope_mode);
    if ((rc = prefill_attention(h, h->pf_qkv, nqkv, kv, max_len, h->pf_a, nq, S, R, s, base))) return rc;
    if ((rc = proj_rows(h, h->pf_a, nq, lw.out_proj, lw.out_proj_bias, h->pf_x, d, M, d, nq, exact, s))) return rc;
    launch_add_ln(h, h->pf_x, h->pf_res, 1, 1, lw.norm2_w, lw.norm2_b, h->pf_n, M, s);
    if ((rc = proj_rows(h, h->pf_n, d, lw.fc1, nullptr, h->pf_u, 2 * F, M, 2 * F, d, exact, s))) return rc;
    hipLaunchKernelGGL(silu_mul_rows_kernel, dim3(M), dim3(256), 0, s, h->pf_u, h->pf_m, F);
    if ((rc = proj_rows(h, h->pf_m, F, lw.fc2, nullptr, h->pf_x, d, M, d, F, exact, s))) return rc;
  }
  return ZN_OK;
}

extern "C" int zn_debug_prefill_mode(zn_handle h, int32_t mode) { if (!h) return ZN_ERR_ARG; h->prefill_mode = mode; return ZN_OK; }

extern "C" int zn_prefill(zn_handle h, const void* hidden_dev, int32_t S, zn_stream stream) {
  if (!h) return ZN_ERR_ARG;
  if (!h->gen_active || h->gen_ended) ZN_FAIL(h, ZN_ERR_STATE, "zn_prefill before zn_gen_begin");
  if (!hidden_dev || S < 1 || S > h->max_len) ZN_FAIL(h, ZN_ERR_ARG, "zn_prefill: bad S=%d (max_len %d)", S, h->max_len);
  hipStream_t s = (hipStream_t)stream;
  const zn_config& c = h->cfg;
  int rc;
  if (c.arch == 1 && h->prefill_mode >= 1 && S > 1 && c.d_model % 32 == 0 && c.d_ff % 32 == 0 && c.m_d_inner % 32 == 0) {
    // all positions at once (mode 2: projections row by row through the step's GEMV: bit-comparable with single steps)
    if ((rc = hybrid_prefill_core(h, (const bf16_t*)hidden_dev, S, h->rows, h->kv_layers.data(), h->max_len, 0, h->prefill_mode == 2, s))) return rc;
    hipLaunchKernelGGL(gather_last_kernel, dim3(h->rows), dim3(256), 0, s, h->pf_x, h->x, S, c.d_model);
    hipLaunchKernelGGL(gather_last_bytes_kernel, dim3(h->rows), dim3(256), 0, s, (const void*)h->pf_res, (void*)h->res, S, c.d_model * (c.residual_in_fp32 ? 4 : 2));
    hipLaunchKernelGGL(add_lengths_kernel, dim3(1), dim3(64 > h->rows ? 64 : h->rows), 0, s, h->lengths, h->rows, S);
    h->len_hi += S;
    if ((rc = hybrid_heads(h, s))) return rc;
  } else if (c.arch == 0 && h->prefill_mode == 1 && S > 1 && c.d_model % 32 == 0 && c.d_ff % 32 == 0) {
    if ((rc = prefill_batched(h, (const bf16_t*)hidden_dev, S, s))) return rc;
    h->len_hi += S;
  } else {
    // Position by position through the decode kernels.  Row-wise ops are independent of S; attention reproduces the
    // reference's causal flash-attention blocking through `ext` = keys spanned by the query block of this position.
    const int qb = qsplit(S);
    for (int p = 0; p < S; ++p) {
      hipLaunchKernelGGL(gather_pos_kernel, dim3(h->rows), dim3(256), 0, s, (const bf16_t*)hidden_dev, h->x, S, p, c.d_model);
      int ext = (p / qb) * qb + qb; if (ext > S) ext = S;
      h->attn_fused = attn_fused_for(h, ++h->len_hi, h->rows);
      if (c.arch == 1) {
        if ((rc = hybrid_token(h, p == S - 1, s))) return rc;
      } else {
        if ((rc = decode_blocks(h, nullptr, ext, s))) return rc;
      }
      hipLaunchKernelGGL(add_lengths_kernel, dim3(1), dim3(64 > h->rows ? 64 : h->rows), 0, s, h->lengths, h->rows, 1);
    }
  }
  if (c.arch == 0 && (rc = heads_logits(h, h->x, h->rows, s))) return rc;
  HIPCHK(h, hipGetLastError());
  return ZN_OK;
}

extern "C" int zn_sample_first(zn_handle h, zn_stream stream) {
  if (!h) return ZN_ERR_ARG;
  if (!h->gen_active) ZN_FAIL(h, ZN_ERR_STATE, "zn_sample_first before zn_gen_begin");
  hipStream_t s = (hipStream_t)stream;
  const zn_config& c = h->cfg;
  SampleArgs a = make_sample_args(h, h->sp);
  a.raw = h->logits_raw; a.mix = 1; a.cfg_scale = h->cfg_scale; a.apply_bias = 0; a.batch = h->batch;
  a.use_penalty = 0; a.st = h->st; a.logits_out = h->last_logits; a.tokens = h->tok_raw; a.draw = 0;
  hipLaunchKernelGGL(sample_kernel, dim3(c.n_codebooks, h->batch), dim3(256), 0, s, a);
  FrameArgs f{};
  f.st = h->st; f.codes = h->codes; f.t_total = h->t_total; f.batch = h->batch; f.n_q = c.n_codebooks; f.eos_id = c.eos_id;
  f.mask_id = c.mask_id; f.tokens = h->tok_raw; f.remaining = h->remaining; f.stopping = h->stopping; f.lengths = h->lengths;
  f.rows = h->rows; f.first = 1; f.override = h->tok_override; f.override_calls = h->tok_override_calls;
  hipLaunchKernelGGL(frame_update_kernel, dim3(1), dim3(256), 0, s, f);
  h->emb_valid = false;
  HIPCHK(h, hipGetLastError());
  return ZN_OK;
}

extern "C" int zn_decode_steps(zn_handle h, int32_t n, zn_stream stream) {
  if (!h) return ZN_ERR_ARG;
  if (!h->gen_active || h->gen_ended) ZN_FAIL(h, ZN_ERR_STATE, "zn_decode_steps before zn_gen_begin");
  if (n < 0) ZN_FAIL(h, ZN_ERR_ARG, "n < 0");
  hipStream_t s = (hipStream_t)stream;
  h->gen_stream = s;
  if (n > 0 && tail_fused(h) && !h->emb_valid) {   // first step of the generation: later ones find the embedding their predecessor's tail left
    hipLaunchKernelGGL(embed_kernel, dim3(h->batch), dim3(256), 0, s, make_embed_args(h));
    h->emb_valid = true;
  }
  for (int i = 0; i < n;) {
    // this step appends one key per row; a run of ZN_GRAPH_STEPS steps with one launch shape replays the long graph
    const int fused = attn_fused_for(h, h->len_hi + 1, h->rows);
    const bool want_run = n - i >= ZN_GRAPH_STEPS && h->tune[6] > 1;
    // the whole-step kernel serves every step whose context fits one of its instantiations (stack_mode_for); a run of steps takes the
    // attention role its LAST step needs (workgroups of key blocks past a step's context leave at once)
    const bool st_ok = tail_fused(h);
    const int mode_run = (st_ok && want_run) ? stack_mode_for(h, h->rows, h->len_hi + ZN_GRAPH_STEPS) : -1;
    const int mode_one = st_ok ? stack_mode_for(h, h->rows, h->len_hi + 1) : -1;
    const bool stack_run = mode_run >= 0, stack = stack_run || mode_one >= 0;
    const int mode = stack_run ? mode_run : mode_one;
    const int run = stack ? (stack_run ? ZN_GRAPH_STEPS : 1) : ((want_run && attn_fused_for(h, h->len_hi + ZN_GRAPH_STEPS, h->rows) == fused) ? ZN_GRAPH_STEPS : 1);
    const int k = stack ? (8 + (run > 1 ? ZN_SK_KB_MAXNB + 1 : 0) + mode) : (fused + (run > 1 ? 4 : 0));      // launches path: slots 0..2 and 4..6
    h->attn_fused = fused; h->use_stack = stack;
    if (stack) h->stack_nbk = mode;
    if (!h->graph_exec[k] && !h->graph_tried[k] && n > 1) {
      // capture; every step-varying quantity (column, positions) is read from device memory
      h->graph_tried[k] = true;
      // capture on an internal stream: the caller's stream may be the legacy null stream, which cannot be captured
      if (!h->cap_stream) (void)hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking);
      hipStream_t cs = h->cap_stream;
      if (cs && hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        int rc = ZN_OK;
        for (int j = 0; j < run && rc == ZN_OK; ++j) rc = enqueue_step(h, cs);
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(cs, &g);
        if (rc == ZN_OK && e == hipSuccess && g && hipGraphInstantiate(&h->graph_exec[k], g, nullptr, nullptr, 0) == hipSuccess) h->graph[k] = g;
        else { if (g) (void)hipGraphDestroy(g); h->graph_exec[k] = nullptr; (void)hipGetLastError(); }
      } else (void)hipGetLastError();
    }
    if (h->graph_exec[k]) HIPCHK(h, hipGraphLaunch(h->graph_exec[k], s));
    else for (int j = 0; j < run; ++j) { int rc = enqueue_step(h, s); if (rc) return rc; }
    i += run; h->len_hi += run; h->epoch_bound += (unsigned long long)run * (h->cfg.n_layer + 1);
  }
  HIPCHK(h, hipGetLastError());
  return ZN_OK;
}

extern "C" int zn_gen_end(zn_handle h) {
  if (!h) return ZN_ERR_ARG;
  (void)zn_tenant_release(h->device, h);
  if (h->gen_active && !h->gen_ended && h->demoted && !h->gen_timed_out) h->clean_since_demotion++;
  h->gen_ended = true;
  return ZN_OK;
}

// [0] hand-off timeouts reported on this handle, [1] generations begun, [2] of those, batch-1 generations that ran the launches path because an
// earlier timeout had demoted the handle, [3] 1 while demoted, [4] times the handle was re-armed, [5] clean generations since the demotion,
// [6] the longest in-kernel hand-off wait any whole-step launch of this handle measured, in microseconds (0: none beyond 0.1 ms), [7] waits
// beyond 0.2 ms (a pause of the device shows up as one such wait per waiting wave).  Reading [6], [7] synchronises with the device.
extern "C" int zn_get_counters(zn_handle h, int64_t* out, int32_t n) {
  if (!h || !out || n < 0) return ZN_ERR_ARG;
  unsigned w[2] = {0, 0};
  if (n > 6 && h->ch_diag) HIPCHK(h, hipMemcpy(w, h->ch_diag + 8, sizeof w, hipMemcpyDeviceToHost));
  const long long v[8] = {h->n_timeouts, h->n_generations, h->n_fallback_generations, h->demoted ? 1 : 0, h->n_rearms, h->clean_since_demotion, (long long)(w[0] / 100u), (long long)w[1]};
  for (int i = 0; i < n && i < 8; ++i) out[i] = v[i];
  return ZN_OK;
}

extern "C" int zn_decode_path(zn_handle h) { return (h && h->gen_active && h->cfg.arch == 0 && chain_active(h, h->rows)) ? 1 : 0; }
extern "C" int zn_decode_path_detail(zn_handle h) {
  if (!h || !h->gen_active || h->cfg.arch != 0 || !chain_active(h, h->rows)) return 0;
  return h->use_stack ? 2 : 1;
}
extern "C" int zn_graph_active(zn_handle h) {
  if (!h) return 0;
  for (int k = 0; k < ZN_NGRAPHS; ++k) if (h->graph_exec[k]) return 1;
  return 0;
}

// A bounded hand-off wait of the persistent kernels gave up (sweep_granules): the generation's results are void.  The message names the
// wait (stage: 1 y1, 2 x1, 3 m, 4 x2, 5 q|k|v, 6 attention output, 7 block maxima and 8 partials of the key-block attention role; block; workgroup; wave; the tag waited for and the first stale granule),
// the handle falls back to one launch per op (no in-launch hand-offs) for its later generations, and the sticky word is cleared so
// that a new generation can run.
static int handoff_timeout(zn_handle h, int count) {
  // zn_all_stopped_end only waited for the stop event: in deferred mode up to 16 further steps (graph launches) may still be queued behind
  // it.  Drain the generation's stream before the diagnostic words are read and cleared and before the graphs those steps run from are
  // destroyed (on a non-blocking stream the null-stream copies below would not wait for them).
  if (h->gen_stream) (void)hipStreamSynchronize(h->gen_stream);
  (void)hipMemcpy(h->diag_host, h->ch_diag, sizeof h->diag_host, hipMemcpyDeviceToHost);
  (void)hipMemset(h->ch_diag, 0, 32);                   // (the wait statistics in words 8, 9 stay)
  (void)hipMemset(&h->st->pad[0], 0, sizeof(int));
  h->demoted = true; h->clean_since_demotion = 0; h->n_timeouts++; h->gen_timed_out = true;
  free_graph(h);
  const unsigned* d = h->diag_host;
  ZN_FAIL(h, ZN_ERR_HIP, "decode chain: %d hand-off wait(s) timed out - the results of this generation are invalid. First: stage %u of block %u, workgroup %u wave %u lane %u "
          "waited for tag %u, granule at byte %u carried tag %u after %u passes. This handle runs the launches path for its next %d generations (zn_debug_tune(8, 1) re-arms the persistent kernels at once; zn_get_counters)",
          count, d[0] >> 8, d[0] & 255u, d[1], d[2], d[6], d[3], d[4], d[5], d[7], ZN_REARM_AFTER);
}

extern "C" int zn_all_stopped(zn_handle h, int32_t* out, zn_stream stream) {
  if (!h || !out) return ZN_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  // all_done, force_eos_step, eos_bias, pad[0] = the persistent chain's sticky timeout word: one 16-byte copy
  HIPCHK(h, hipMemcpyAsync(h->done_host, &h->st->all_done, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  *out = h->done_host[0];
  if (h->done_host[3] != 0) return handoff_timeout(h, h->done_host[3]);
  return ZN_OK;
}

// The stop check without a host round trip on the critical path: *_begin queues the copy of the loop state behind the steps
// enqueued so far, the caller enqueues the next steps, *_end (one batch later) waits for the copy - long done by then.
extern "C" int zn_all_stopped_begin(zn_handle h, zn_stream stream) {
  if (!h) return ZN_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(h, hipMemcpyAsync(h->done_host + 4, &h->st->all_done, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipEventRecord(h->stop_event, s));
  h->stop_pending = true;
  return ZN_OK;
}
extern "C" int zn_all_stopped_end(zn_handle h, int32_t* out) {
  if (!h || !out) return ZN_ERR_ARG;
  if (!h->stop_pending) ZN_FAIL(h, ZN_ERR_STATE, "zn_all_stopped_end without zn_all_stopped_begin");
  HIPCHK(h, hipEventSynchronize(h->stop_event));
  h->stop_pending = false;
  *out = h->done_host[4];
  if (h->done_host[7] != 0) return handoff_timeout(h, h->done_host[7]);
  return ZN_OK;
}

extern "C" int zn_get_step_outputs(zn_handle h, float* logits_dev, int32_t* tokens_dev, zn_stream stream) {
  if (!h) return ZN_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const zn_config& c = h->cfg;
  if (logits_dev) HIPCHK(h, hipMemcpyAsync(logits_dev, h->last_logits, (size_t)h->batch * c.n_codebooks * c.vocab_head * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (tokens_dev) HIPCHK(h, hipMemcpyAsync(tokens_dev, h->tok_raw, (size_t)h->batch * c.n_codebooks * sizeof(int), hipMemcpyDeviceToDevice, s));
  return ZN_OK;
}

extern "C" int zn_codes_changed(zn_handle h) { if (!h) return ZN_ERR_ARG; h->emb_valid = false; return ZN_OK; }
extern "C" int zn_debug_force_eos(zn_handle h, int32_t step) { if (!h) return ZN_ERR_ARG; h->force_eos_step = step; return ZN_OK; }
extern "C" int zn_debug_token_override(zn_handle h, const int32_t* tokens_dev, int32_t calls) {
  if (!h) return ZN_ERR_ARG;
  h->tok_override = tokens_dev; h->tok_override_calls = tokens_dev ? calls : 0;
  free_graph(h);
  return ZN_OK;
}
extern "C" int zn_debug_tune(zn_handle h, int32_t key, int32_t value) {
  if (!h || key < 0 || key >= 20 || value < 1) return ZN_ERR_ARG;
  if (key == 14 && value == 13) {                        // observation hook: forget the wait statistics (zn_get_counters [6], [7]) gathered so far
    if (h->ch_diag) { (void)hipDeviceSynchronize(); (void)hipMemset(h->ch_diag + 8, 0, 2 * sizeof(unsigned)); }
    return ZN_OK;
  }
  h->tune[key] = value; free_graph(h); h->emb_valid = false;
  if (key == 8 && value == 1 && h->demoted) { h->demoted = false; h->clean_since_demotion = 0; h->n_rearms++; }
  return ZN_OK;
}
extern "C" int zn_debug_trace(zn_handle h, void* trace_dev) {
  if (!h) return ZN_ERR_ARG;
  h->dbg_trace = (bf16_t*)trace_dev; free_graph(h);
  return ZN_OK;
}
extern "C" int zn_debug_chain_stamps(zn_handle h, uint64_t* stamps_dev) {
  if (!h) return ZN_ERR_ARG;
  h->ch_stamps = (unsigned long long*)stamps_dev; free_graph(h);
  return ZN_OK;
}
extern "C" int zn_debug_eos_bias(zn_handle h, float bias) { if (!h) return ZN_ERR_ARG; h->eos_bias = bias; return ZN_OK; }

// ------------------------------------------------------------------------------------------------ measurement
// Launches one of the step's weight-streaming kernels `iters` times on `stream`, cycling over the layers so that
// every launch streams its weights from HBM (26 x 67 MB >> the 256 MiB Infinity Cache), bracketed by HIP events on
// that stream.  which: 0 = LayerNorm+fc1+SiLU-gate, 1 = fc2+residual, 2 = out_proj+residual, 3 = LayerNorm+heads.
extern "C" int zn_bench_kernel(zn_handle h, int32_t which, int32_t rows, int32_t iters, float* ms_per_launch, double* bytes_per_launch,
                               zn_stream stream) {
  if (!h) return ZN_ERR_ARG;
  if (!ms_per_launch || !bytes_per_launch || iters < 1 || (rows & 0xff) < 1 || (rows & 0xff) > h->max_rows || which < 0 || which > 6)
    ZN_FAIL(h, ZN_ERR_ARG, "zn_bench_kernel: bad argument");
  h->gen_stream = (hipStream_t)stream;
  const bool same_layer = (rows & 0x100) != 0;   // measurement variant: keep hitting layer 0 (weights stay in the Infinity Cache)
  const int ctx_arg = (rows >> 16) & 0x7fff;     // which == 6: keys already in the cache (0 = 450, the mean context of a 10 s utterance)
  rows &= 0xff;
  hipStream_t s = (hipStream_t)stream;
  const zn_config& c = h->cfg;
  const int d = c.d_model;
  HIPCHK(h, hipMemsetAsync(h->x, 0, (size_t)rows * d * 2, s));
  HIPCHK(h, hipMemsetAsync(h->mbuf, 0, (size_t)rows * c.d_ff * 2, s));
  HIPCHK(h, hipMemsetAsync(h->o1, 0, (size_t)rows * d * 2, s));
  const int hd = h->hd, nq = c.n_heads * hd, nkv = c.n_heads_kv * hd;
  bf16_t* tkv = nullptr; int* tlen = nullptr;             // which == 4, 5: a scratch cache of 8 positions per row, all rows at position 0
  if (which == 5 && !chain_active(h, rows)) ZN_FAIL(h, ZN_ERR_UNSUPPORTED, "zn_bench_kernel: the persistent chain serves batch 1 (2 rows) of the transformer only");
  if (which == 4 || which == 5) {
    HIPCHK(h, hipMalloc(&tkv, (size_t)rows * 8 * 2 * nkv * 2));
    HIPCHK(h, hipMalloc(&tlen, (size_t)rows * sizeof(int)));
    HIPCHK(h, hipMemsetAsync(tlen, 0, (size_t)rows * sizeof(int), s));
  }
  // which == 6: the whole-step kernel (every block of a decode step + the heads in ONE launch) on a scratch cache of its own per layer,
  // all rows at position ctx; the handle's generation state is put back afterwards
  std::vector<void*> skv;
  std::vector<const void*> saved_kv = h->kv_layers;
  const int saved_max_len = h->max_len; int* const saved_lengths = h->lengths;
  const int ctx = ctx_arg > 0 ? ctx_arg : 450;
  if (which == 6) {
    if (!chain_active(h, rows)) ZN_FAIL(h, ZN_ERR_UNSUPPORTED, "zn_bench_kernel: the whole-step kernel serves batch 1 (2 rows) of the transformer only");
    if (!h->stack_checked) { h->stack_ok = stack_shapes_ok(h); h->stack_checked = true; }
    h->stack_nbk = stack_mode_for(h, rows, ctx + 1);
    if (h->stack_nbk < 0) ZN_FAIL(h, ZN_ERR_UNSUPPORTED, "zn_bench_kernel: the whole-step kernel does not serve this model / context");
    const int cap = ctx + 8;
    std::vector<int> lens(rows, ctx);
    HIPCHK(h, hipMalloc(&tlen, (size_t)rows * sizeof(int)));
    HIPCHK(h, hipMemcpy(tlen, lens.data(), (size_t)rows * sizeof(int), hipMemcpyHostToDevice));
    skv.assign(c.n_layer, nullptr);
    h->kv_layers.assign(c.n_layer, nullptr);
    for (int li = 0; li < c.n_layer; ++li) {
      HIPCHK(h, hipMalloc(&skv[li], (size_t)rows * cap * 2 * nkv * 2));
      HIPCHK(h, hipMemsetAsync(skv[li], 0, (size_t)rows * cap * 2 * nkv * 2, s));
      h->kv_layers[li] = skv[li];
    }
    h->max_len = cap; h->lengths = tlen;
    HIPCHK(h, hipMemsetAsync(h->x_emb, 0, (size_t)rows * d * 2, s));
    HIPCHK(h, hipMemsetAsync(h->q, 0, (size_t)rows * d * 2, s));
    int rcb = build_stack_table(h);
    if (rcb) return rcb;
  }
  hipEvent_t e0, e1;
  HIPCHK(h, hipEventCreate(&e0));
  HIPCHK(h, hipEventCreate(&e1));
  int rc = ZN_OK;
  // which == 0 at 5..16 rows: the launch a decode step makes there - fc1 normalising from the statistics its producer left (layer_post_attention);
  // the producer runs once, outside the timed loop, on the zeroed rows
  bool fc1_lnp = false;
  if (which == 0 && rows > 4 && h->tune[9] != 2 && h->ln_part && d == 16 * ZN_G16_LNT && gemm16k_fits(h, EPI_RESID, d, nq)) {
    GemvArgs b{};
    b.W = (const bf16_t*)h->layers[0].out_proj; b.N = d; b.K = nq; b.x = h->o1; b.resid = h->x; b.out = h->x; b.eps = c.norm_eps; b.ln_part_out = h->ln_part;
    rc = run_gemv<PRO_NONE, EPI_RESID>(h, b, rows, h->tune[1], s);
    fc1_lnp = rc == ZN_OK;
  }
  for (int pass = 0; pass < 2 && rc == ZN_OK; ++pass) {   // pass 0 = warm-up
    if (pass == 1) HIPCHK(h, hipEventRecord(e0, s));
    for (int i = 0; i < (pass ? iters : (iters < 8 ? iters : 8)) && rc == ZN_OK; ++i) {
      const zn_layer_weights& lw = h->layers[same_layer ? 0 : i % c.n_layer];
      GemvArgs a{};
      a.eps = c.norm_eps;
      if (which == 0) {
        a.W = (const bf16_t*)lw.fc1; a.N = 2 * c.d_ff; a.K = d; a.x = h->x; a.ln_w = (const bf16_t*)lw.norm2_w; a.ln_b = (const bf16_t*)lw.norm2_b; a.out = h->mbuf;
        if (fc1_lnp) a.ln_part_in = h->ln_part;
        rc = run_gemv<PRO_LN, EPI_SILU>(h, a, rows, h->tune[2], s);
      } else if (which == 1) {
        a.W = (const bf16_t*)lw.fc2; a.N = d; a.K = c.d_ff; a.x = h->mbuf; a.resid = h->x; a.out = h->x;
        rc = run_gemv<PRO_NONE, EPI_RESID>(h, a, rows, h->tune[3], s);
      } else if (which == 2) {
        a.W = (const bf16_t*)lw.out_proj; a.N = d; a.K = c.n_heads * h->hd; a.x = h->o1; a.resid = h->x; a.out = h->x;
        rc = run_gemv<PRO_NONE, EPI_RESID>(h, a, rows, h->tune[1], s);
      } else if (which == 6) {
        rc = launch_stack(h, s);
      } else if (which == 5) {
        if (i % c.n_layer == c.n_layer - 1) continue;       // the last block's launch has no in_proj: not the launch being priced
        rc = launch_chain(h, i % c.n_layer, std::vector<const void*>(c.n_layer, tkv), 8, tlen, s);
      } else if (which == 4) {
        a.W = (const bf16_t*)lw.in_proj; a.N = nq + 2 * nkv; a.K = d; a.x = h->x; a.ln_w = (const bf16_t*)lw.norm_w; a.ln_b = (const bf16_t*)lw.norm_b;
        a.lengths = tlen; a.hd = hd; a.n_heads = c.n_heads; a.n_heads_kv = c.n_heads_kv;
        a.q_out = h->q; a.kv = tkv; a.rope = h->rope; a.max_len = 8; a.rope_positions = c.rope_positions;
        rc = run_gemv<PRO_LN, EPI_ROPE_KV>(h, a, rows, h->tune[0], s);
      } else {
        rc = heads_logits(h, h->x, rows, s);
      }
    }
  }
  if (rc) return rc;
  HIPCHK(h, hipEventRecord(e1, s));
  HIPCHK(h, hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (tkv) (void)hipFree(tkv);
  if (tlen) (void)hipFree(tlen);
  if (which == 6) {
    for (void* p : skv) if (p) (void)hipFree(p);
    h->kv_layers = saved_kv; h->max_len = saved_max_len; h->lengths = saved_lengths;
    if (h->gen_active && (int)h->kv_layers.size() == c.n_layer) { int rcb = build_stack_table(h); if (rcb) return rcb; }
    int tmo = 0;
    HIPCHK(h, hipMemcpy(&tmo, &h->st->pad[0], sizeof(int), hipMemcpyDeviceToHost));
    if (tmo != 0) return handoff_timeout(h, tmo);
  }
  if (which == 5 || which == 6) h->epoch_bound += (unsigned long long)(iters + 8) * (c.n_layer + 1);
  int launches = iters;
  if (which == 5) { launches = 0; for (int i = 0; i < iters; ++i) launches += (i % c.n_layer != c.n_layer - 1); }
  *ms_per_launch = ms / launches;
  // chain launch: out_proj (read once, applied twice), fc1, fc2 of the block and in_proj of the next one
  const double wbytes = which == 5 ? ((double)d * nq + 3.0 * c.d_ff * d + (double)(nq + 2 * nkv) * d) * 2 : which == 0 ? 2.0 * c.d_ff * d * 2 : which == 1 ? (double)d * c.d_ff * 2 : which == 2 ? (double)d * c.n_heads * h->hd * 2
                        : which == 4 ? (double)(nq + 2 * nkv) * d * 2 : (double)c.n_codebooks * c.vocab_head * d * 2;
  *bytes_per_launch = wbytes;   // algorithmic bytes = the weight matrix, read once (activations are KBs)
  if (which == 6) {             // every block's out_proj (once), fc1, fc2, in_proj; the heads; K and V of `ctx` keys read, one row written
    const double per_block = ((double)d * nq + 3.0 * c.d_ff * d) * 2, inp = (double)(nq + 2 * nkv) * d * 2, kvpos = 2.0 * nkv * 2;
    const int nin = stack_pre(h) ? c.n_layer : c.n_layer - 1;      // in_proj of block 0 inside the launch (the pre-block) or before it
    *bytes_per_launch = c.n_layer * per_block + nin * inp + (double)c.n_codebooks * c.vocab_head * d * 2 +
                        (double)rows * c.n_layer * kvpos * ctx + (double)rows * nin * kvpos;
  }
  return ZN_OK;
}

// ------------------------------------------------------------------------------------------------ single ops
extern "C" int zn_op_linear(zn_handle h, const void* x, const void* ln_w, const void* ln_b, const void* W, void* out, int32_t rows,
                            int32_t N, int32_t K, zn_stream stream) {
  if (!h) return ZN_ERR_ARG;
  if (!x || !W || !out || rows < 1 || N < 1 || K < 8 || K % 8) ZN_FAIL(h, ZN_ERR_ARG, "zn_op_linear: bad argument");
  GemvArgs a{};
  a.W = (const bf16_t*)W; a.N = N; a.K = K; a.x = (const bf16_t*)x; a.out = (bf16_t*)out; a.eps = h->cfg.norm_eps;
  int rc;
  if (ln_w) {
    if (K > 4096) ZN_FAIL(h, ZN_ERR_UNSUPPORTED, "zn_op_linear: fused LayerNorm needs K <= 4096");
    a.ln_w = (const bf16_t*)ln_w; a.ln_b = (const bf16_t*)ln_b;
    rc = run_gemv<PRO_LN, EPI_STORE>(h, a, rows, 1024, (hipStream_t)stream);
  } else rc = run_gemv<PRO_NONE, EPI_STORE>(h, a, rows, 1024, (hipStream_t)stream);
  if (rc) return rc;
  HIPCHK(h, hipGetLastError());
  return ZN_OK;
}

extern "C" int zn_op_linear_bias(zn_handle h, const void* x, const void* W, const void* bias, void* out, int32_t rows, int32_t N, int32_t K, zn_stream stream) {
  if (!h) return ZN_ERR_ARG;
  if (!x || !W || !out || rows < 1 || N < 1 || K < 8 || K % 8) ZN_FAIL(h, ZN_ERR_ARG, "zn_op_linear_bias: bad argument");
  GemvArgs a{};
  a.W = (const bf16_t*)W; a.N = N; a.K = K; a.x = (const bf16_t*)x; a.out = (bf16_t*)out; a.bias = (const bf16_t*)bias; a.eps = h->cfg.norm_eps;
  hipStream_t s = (hipStream_t)stream;
  for (int r0 = 0; r0 < rows; r0 += 4) {      // always the GEMV (its epilogue carries the bias); rows are few here
    GemvArgs g = a;
    g.x = a.x + (size_t)r0 * K; g.out = a.out + (size_t)r0 * N;
    int rc = run_gemv<PRO_NONE, EPI_STORE>(h, g, rows - r0 < 4 ? rows - r0 : 4, 1024, s);
    if (rc) return rc;
  }
  HIPCHK(h, hipGetLastError());
  return ZN_OK;
}

extern "C" int zn_op_gather_rows(zn_handle h, const void* table, const int32_t* ids, void* out, int32_t n, int32_t d, int32_t table_rows, int32_t id_offset,
                                 zn_stream stream) {
  if (!h) return ZN_ERR_ARG;
  if (!table || !ids || !out || n < 1 || d < 8 || d % 8 || table_rows < 1) ZN_FAIL(h, ZN_ERR_ARG, "zn_op_gather_rows: bad argument");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)table, ids, (bf16_t*)out, d, table_rows, id_offset);
  HIPCHK(h, hipGetLastError());
  return ZN_OK;
}

extern "C" int zn_op_fourier(zn_handle h, const float* x, const void* weight, void* out, int32_t n, int32_t in_dim, int32_t half, float min_val, float max_val,
                             zn_stream stream) {
  if (!h) return ZN_ERR_ARG;
  if (!x || !weight || !out || n < 1 || in_dim < 1 || half < 1 || !(max_val > min_val)) ZN_FAIL(h, ZN_ERR_ARG, "zn_op_fourier: bad argument");
  hipLaunchKernelGGL(fourier_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, x, (const bf16_t*)weight, (bf16_t*)out, in_dim, half, min_val, max_val);
  HIPCHK(h, hipGetLastError());
  return ZN_OK;
}

extern "C" int zn_op_silu(zn_handle h, const void* x, void* out, int64_t n, zn_stream stream) {
  if (!h) return ZN_ERR_ARG;
  if (!x || !out || n < 1) ZN_FAIL(h, ZN_ERR_ARG, "zn_op_silu: bad argument");
  const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(silu_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out, (size_t)n);
  HIPCHK(h, hipGetLastError());
  return ZN_OK;
}

extern "C" int zn_op_layernorm(zn_handle h, const void* x, const void* w, const void* b, void* out, int32_t rows, int32_t d, zn_stream stream) {
  if (!h) return ZN_ERR_ARG;
  if (!x || !w || !b || !out || rows < 1 || d < 8 || d % 8 || d > 4096) ZN_FAIL(h, ZN_ERR_ARG, "zn_op_layernorm: bad argument (d <= 4096, multiple of 8)");
  hipLaunchKernelGGL(layernorm_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b,
                     (bf16_t*)out, d, h->cfg.norm_eps);
  HIPCHK(h, hipGetLastError());
  return ZN_OK;
}

extern "C" int zn_op_layer_decode(zn_handle h, int32_t layer, void* x, void* kv, int32_t max_len, const int32_t* lengths,
                                  const int32_t* ext, int32_t rows, zn_stream stream) {
  if (!h) return ZN_ERR_ARG;
  if (!x || !kv || !lengths || layer < 0 || layer >= h->cfg.n_layer || rows < 1 || rows > h->max_rows || max_len < 1)
    ZN_FAIL(h, ZN_ERR_ARG, "zn_op_layer_decode: bad argument");
  int rc = ensure_attn_ws(h, max_len);
  if (rc) return rc;
  h->attn_fused = attn_fused_for(h, max_len, rows);   // lengths live on the device: bound the context by the capacity
  rc = layer_decode(h, layer, (bf16_t*)x, (bf16_t*)kv, max_len, lengths, ext, 0, rows, (hipStream_t)stream);
  if (rc) return rc;
  HIPCHK(h, hipGetLastError());
  return ZN_OK;
}

extern "C" int zn_op_backbone_forward(zn_handle h, const void* hidden_dev, void* out_dev, const void* const* caches_dev, int32_t max_len,
                                      int32_t base, int32_t S, int32_t rows, zn_stream stream) {
  if (!h) return ZN_ERR_ARG;
  const zn_config& c = h->cfg;
  if (!hidden_dev || !out_dev || !caches_dev || S < 1 || rows < 1 || rows > h->max_rows || base < 0 || max_len < 1 || base + S > max_len)
    ZN_FAIL(h, ZN_ERR_ARG, "zn_op_backbone_forward: bad argument (rows %d of at most %d, positions %d + %d of %d)", rows, h->max_rows, base, S, max_len);
  if (base + S > c.rope_positions)
    ZN_FAIL(h, ZN_ERR_ARG, "sequence length %d exceeds the %d-position RoPE table (zonos/backbone/_torch.py:206)", base + S, c.rope_positions);
  hipStream_t s = (hipStream_t)stream;
  int rc = ensure_attn_ws(h, max_len);
  if (rc) return rc;
  const int d = c.d_model;
  const bf16_t* hid = (const bf16_t*)hidden_dev;
  bf16_t* out = (bf16_t*)out_dev;
  const bool batched = S > 1 && h->prefill_mode >= 1 && d % 32 == 0 && c.d_ff % 32 == 0 && (c.arch == 0 || c.m_d_inner % 32 == 0);
  if (batched) {
    const int M = rows * S;
    if (c.arch == 1) {
      if ((rc = hybrid_prefill_core(h, hid, S, rows, caches_dev, max_len, base, h->prefill_mode == 2, s))) return rc;
      launch_add_ln(h, h->pf_x, h->pf_res, 1, 0, h->norm_f_w, h->norm_f_b, out, M, s);
    } else {
      if ((rc = transformer_prefill_core(h, hid, S, rows, caches_dev, max_len, base, s))) return rc;
      hipLaunchKernelGGL(layernorm_kernel, dim3(M), dim3(64), 0, s, h->pf_x, (const bf16_t*)h->norm_f_w, (const bf16_t*)h->norm_f_b, out, d, c.norm_eps);
    }
    HIPCHK(h, hipGetLastError());
    return ZN_OK;
  }
  // position by position through the decode kernels (S = 1: the loop's step).  Attention reproduces the reference's causal
  // flash-attention blocking through the keys spanned by this position's query block.
  const int qb = qsplit(S);
  for (int p = 0; p < S; ++p) {
    hipLaunchKernelGGL(gather_pos_kernel, dim3(rows), dim3(256), 0, s, hid, h->x, S, p, d);
    hipLaunchKernelGGL(fill_int_kernel, dim3(1), dim3(64 > rows ? 64 : rows), 0, s, h->fw_lengths, rows, base + p);
    int ext = (p / qb) * qb + qb; if (ext > S) ext = S;
    h->attn_fused = attn_fused_for(h, base + p + 1, rows);
    for (int li = 0; li < c.n_layer; ++li) {
      if (c.arch == 1) rc = hybrid_layer(h, li, (void*)caches_dev[li], max_len, h->fw_lengths, rows, s);
      else rc = layer_decode(h, li, h->x, (bf16_t*)caches_dev[li], max_len, h->fw_lengths, nullptr, S > 1 ? base + ext : 0, rows, s);
      if (rc) return rc;
    }
    if (c.arch == 1) launch_add_ln(h, h->x, h->res, 1, 0, h->norm_f_w, h->norm_f_b, h->nbuf, rows, s);
    else hipLaunchKernelGGL(layernorm_kernel, dim3(rows), dim3(64), 0, s, h->x, (const bf16_t*)h->norm_f_w, (const bf16_t*)h->norm_f_b, h->nbuf, d, c.norm_eps);
    hipLaunchKernelGGL(scatter_pos_kernel, dim3(rows), dim3(256), 0, s, h->nbuf, out, S, p, d);
  }
  HIPCHK(h, hipGetLastError());
  return ZN_OK;
}

extern "C" int zn_op_add_layernorm(zn_handle h, const void* hidden, void* res, const void* w, const void* b, void* out, int32_t rows,
                                   int32_t d, float eps, int32_t flags, zn_stream stream) {
  if (!h) return ZN_ERR_ARG;
  if (!hidden || !w || (!b && !(flags & 1)) || !out || rows < 1 || d < 8 || d % 8 || d > 4096) ZN_FAIL(h, ZN_ERR_ARG, "zn_op_add_layernorm: bad argument");
  AddLnArgs a{};
  a.h = (const bf16_t*)hidden; a.res = (bf16_t*)res; a.w = (const bf16_t*)w; a.b = (const bf16_t*)b; a.out = (bf16_t*)out; a.d = d;
  a.has_res = res != nullptr; a.write_res = res != nullptr; a.eps = eps; a.rms = flags & 1; a.res32 = (flags >> 1) & 1;
  hipLaunchKernelGGL(add_ln_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, a);
  HIPCHK(h, hipGetLastError());
  return ZN_OK;
}

extern "C" int zn_op_mamba_step(zn_handle h, int32_t layer, const void* x, void* state, void* out, int32_t rows, zn_stream stream) {
  if (!h) return ZN_ERR_ARG;
  if (h->cfg.arch != 1) ZN_FAIL(h, ZN_ERR_STATE, "zn_op_mamba_step: the handle is not a hybrid backbone");
  if (!x || !state || !out || layer < 0 || layer >= h->cfg.n_layer || rows < 1 || rows > h->max_rows) ZN_FAIL(h, ZN_ERR_ARG, "zn_op_mamba_step: bad argument");
  if (h->layers[layer].kind != 1) ZN_FAIL(h, ZN_ERR_ARG, "zn_op_mamba_step: layer %d is an attention layer", layer);
  int rc = mamba_mixer(h, layer, (const bf16_t*)x, state, (bf16_t*)out, rows, (hipStream_t)stream);
  if (rc) return rc;
  HIPCHK(h, hipGetLastError());
  return ZN_OK;
}

extern "C" int zn_op_attn_decode(zn_handle h, const void* q, const void* kv, int32_t max_len, const int32_t* lengths, const int32_t* ext,
                                 void* out, int32_t rows, zn_stream stream) {
  if (!h) return ZN_ERR_ARG;
  if (!q || !kv || !lengths || !out || rows < 1 || max_len < 1) ZN_FAIL(h, ZN_ERR_ARG, "zn_op_attn_decode: bad argument");
  if (rows > h->max_rows) ZN_FAIL(h, ZN_ERR_ARG, "zn_op_attn_decode: rows > max_rows");
  int rc = ensure_attn_ws(h, max_len);
  if (rc) return rc;
  h->attn_fused = attn_fused_for(h, max_len, rows);
  rc = run_attention(h, (const bf16_t*)q, (const bf16_t*)kv, max_len, lengths, ext, 0, (bf16_t*)out, rows, (hipStream_t)stream);
  if (rc) return rc;
  HIPCHK(h, hipGetLastError());
  return ZN_OK;
}

extern "C" int zn_op_attn_prefill(zn_handle h, const void* q, const void* kv, int32_t max_len, void* out, int32_t positions, int32_t rows,
                                  zn_stream stream) {
  if (!h) return ZN_ERR_ARG;
  if (!q || !kv || !out || rows < 1 || positions < 1 || max_len < positions) ZN_FAIL(h, ZN_ERR_ARG, "zn_op_attn_prefill: bad argument");
  const int nq = h->cfg.n_heads * h->hd;
  int rc = prefill_attention(h, (const bf16_t*)q, nq, (const bf16_t*)kv, max_len, (bf16_t*)out, nq, positions, rows, (hipStream_t)stream);
  if (rc) return rc;
  HIPCHK(h, hipGetLastError());
  return ZN_OK;
}

extern "C" int zn_op_embed(zn_handle h, const int32_t* codes, void* out, int32_t batch, zn_stream stream) {
  if (!h) return ZN_ERR_ARG;
  if (!codes || !out || batch < 1) ZN_FAIL(h, ZN_ERR_ARG, "zn_op_embed: bad argument");
  if (!h->has_io) ZN_FAIL(h, ZN_ERR_STATE, "zn_op_embed: handle was created without embeddings");
  const zn_config& c = h->cfg;
  EmbedArgs e{};
  e.tables = h->emb_tables_dev; e.codes = codes; e.sb = c.n_codebooks; e.si = 1; e.col = 0; e.n_q = c.n_codebooks; e.d = c.d_model;
  e.batch = batch; e.vocab_embed = c.vocab_embed; e.out = (bf16_t*)out; e.dup = 0;
  hipLaunchKernelGGL(embed_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, e);
  HIPCHK(h, hipGetLastError());
  return ZN_OK;
}

extern "C" int zn_op_sample(zn_handle h, const float* logits, const int32_t* recent, int32_t window, const zn_sampling* sp,
                            uint64_t draw_index, int32_t* tokens, float* probs_out, int32_t batch, zn_stream stream) {
  if (!h) return ZN_ERR_ARG;
  if (!logits || !sp || !tokens || batch < 1 || (recent && window < 1)) ZN_FAIL(h, ZN_ERR_ARG, "zn_op_sample: bad argument");
  SampleArgs a = make_sample_args(h, *sp);
  a.raw = logits; a.mix = 0; a.apply_bias = 0; a.batch = batch; a.recent = recent; a.window = window;
  a.use_penalty = (recent != nullptr && sp->repetition_penalty != 1.0f); a.tokens = tokens; a.probs_out = probs_out; a.draw = draw_index;
  hipLaunchKernelGGL(sample_kernel, dim3(h->cfg.n_codebooks, batch), dim3(256), 0, (hipStream_t)stream, a);
  HIPCHK(h, hipGetLastError());
  return ZN_OK;
}
