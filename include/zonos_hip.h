/* zonos_hip.h — C ABI of libzonos_hip.so: the MI355X (gfx950) implementation of the Zonos TTS hot path.
 *
 * The reference (langfod/Zonos) is 100 % Python and has no FFI; its seam for this path is the Python surface
 * zonos/model.py (Zonos.generate :354-548, _compute_logits :225-234, setup_cache :305-338), the backbone plugin
 * contract zonos/backbone/__init__.py:24-36 + _torch.py:157,213, zonos/sampling.py:166-231 and
 * zonos/autoencoder.py:119-170 (DACAutoencoder.decode / decode_to_int16).  Each entry point below names the
 * reference interface it replaces.  INTEGRATION.md shows the ctypes binding a maintainer of the reference adds.
 *
 * Conventions: extern "C", plain pointers and sizes, no torch types.  Every call returns an int status
 * (ZN_OK = 0, < 0 = error) and never throws or aborts; zn_last_error(h) gives the message.  All `*_dev`
 * pointers are device (HBM) pointers owned by the caller (torch tensors' data_ptr()); the library owns only the
 * handle-scoped workspace.  `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream), 0 = default.
 * One handle = one device; calls on one handle must not overlap in time; distinct handles are independent.
 * bf16 = raw uint16 bit pattern, row-major tensors, nn.Linear weights are [out_features][in_features].
 */
#ifndef ZONOS_HIP_H
#define ZONOS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZN_ABI_VERSION 5

enum zn_status {
  ZN_OK = 0,
  ZN_ERR_ARG = -1,          /* bad argument / shape the kernels do not support */
  ZN_ERR_HIP = -2,          /* a HIP runtime call failed */
  ZN_ERR_STATE = -3,        /* call order violated (e.g. decode before zn_gen_begin) */
  ZN_ERR_UNSUPPORTED = -4,  /* valid in the reference, not built yet */
  ZN_ERR_NOMEM = -5
};

typedef struct zn_handle_s* zn_handle;
typedef struct zn_dac_s* zn_dac;
typedef struct zn_spk_s* zn_spk;
typedef void* zn_stream;

/* Model hyper-parameters: zonos/config.py:55-84 BackboneConfig + :105-126 ZonosConfig (read from config.json
 * by the host; never hard-coded). */
typedef struct zn_config {
  int32_t d_model, n_layer, n_heads, n_heads_kv, d_ff; /* attn_mlp_d_intermediate */
  int32_t n_codebooks;     /* 9 */
  int32_t vocab_head;      /* 1025 logits per codebook (zonos/model.py:82) */
  int32_t vocab_embed;     /* 1032 embedding rows (zonos/model.py:80) */
  int32_t eos_id, mask_id; /* 1024, 1025 */
  int32_t rope_positions;  /* rows of the RoPE table, 16384 (_torch.py:206) */
  int32_t double_out_proj; /* 1 = reproduce _torch.py:419-420 (out_proj applied twice), 0 = upstream behaviour */
  float norm_eps;          /* 1e-5 */
  /* ABI 2 — hybrid backbone (zonos/backbone/_mamba_ssm.py:8-119; BackboneConfig.ssm_cfg, zonos/config.py:79).
   * arch 0: every layer is a _torch.py TransformerBlock (fields below ignored).  arch 1: mamba_ssm Block semantics
   * (fused residual-add + LayerNorm with the sum kept in fp32 for the norm, single out_proj, deferred residual),
   * layer i is a Mamba2 mixer unless zn_layer_weights.kind says attention. */
  int32_t arch;
  int32_t m_d_inner;       /* expand * d_model */
  int32_t m_headdim;       /* 64 */
  int32_t m_d_state;       /* 64 or 128 */
  int32_t m_ngroups;       /* 1 */
  int32_t m_d_conv;        /* 4 */
  /* ABI 3 - remaining BackboneConfig switches of the hybrid stack (zonos/config.py:80-84, create_block arguments at
   * _mamba_ssm.py:45-58); arch 0 ignores them (the torch backbone reads none of them). */
  int32_t rms_norm;         /* 1: every block norm and the final norm are RMSNorm (no mean; the final norm keeps its bias) */
  int32_t residual_in_fp32; /* 1: the residual stream between blocks is fp32 */
  int32_t rope_mode;        /* attention layers (mamba_ssm MHA, attn_cfg): 0 = interleaved pairs over the whole head,
                               1 = half-split (rotary_emb_interleaved = False, the library default and the Zonos-v0.1-hybrid
                               checkpoint's), 2 = no rotary (rotary_emb_dim = 0) */
} zn_config;

typedef struct zn_layer_weights { /* bf16; names = _torch.py:278-281,373-374,453-454 */
  const void *norm_w, *norm_b;     /* [d] */
  const void *in_proj;             /* [(H+2Hkv)*hd, d] */
  const void *out_proj;            /* [d, H*hd] */
  const void *norm2_w, *norm2_b;   /* [d] */
  const void *fc1;                 /* [2*d_ff, d]  rows [0,d_ff) = value, [d_ff,2d_ff) = gate */
  const void *fc2;                 /* [d, d_ff] */
  /* ABI 2: kind 0 = attention + gated MLP (fields above), 1 = Mamba2 mixer (fields below + norm_w/norm_b);
   * names = mamba_ssm Mamba2 parameters under layers.{i}.mixer. */
  int32_t kind;
  const void *m_in_proj;           /* [2*d_inner + 2*ngroups*d_state + nheads, d] */
  const void *m_conv_w, *m_conv_b; /* conv1d.weight [conv_dim, 1, d_conv], conv1d.bias [conv_dim]; conv_dim = d_inner + 2*ngroups*d_state */
  const void *m_dt_bias, *m_A_log, *m_D;   /* [nheads], nheads = d_inner / headdim */
  const void *m_norm_w;            /* mixer.norm.weight [d_inner] (RMSNormGated) */
  const void *m_out_proj;          /* [d, d_inner] */
  /* ABI 3: optional nn.Linear biases of a mamba_ssm MHA layer (attn_cfg qkv_proj_bias / out_proj_bias; NULL = none) */
  const void *in_proj_bias;        /* [(H+2Hkv)*hd] */
  const void *out_proj_bias;       /* [d] */
} zn_layer_weights;

typedef struct zn_weights {
  const void* const* embeddings;   /* host array of n_codebooks device pointers, each bf16 [vocab_embed, d] */
  const void* heads;               /* bf16 [n_codebooks*vocab_head, d]  (fused_heads, zonos/model.py:82,208-223) */
  const void *norm_f_w, *norm_f_b; /* bf16 [d] */
  const zn_layer_weights* layers;  /* host array [n_layer] */
  const float* rope_table;         /* fp32 [rope_positions, hd/2, 2] (cos,sin), built by the host exactly as
                                      _torch.py:29-34 does (torch.polar on CPU) */
} zn_weights;

/* sampling_params of Zonos.generate (zonos/sampling.py:166-178 defaults). */
typedef struct zn_sampling {
  float temperature;   /* <= 0: greedy argmax */
  float top_p; int32_t top_k; float min_p;
  float linear, conf, quad;
  float repetition_penalty; int32_t repetition_penalty_window;
  uint64_t seed;       /* device RNG stream for the Gumbel-max draw */
} zn_sampling;

/* ---------------------------------------------------------------- lifecycle */
int zn_abi_version(void);
/* Replaces Zonos.__init__/from_local weight binding (zonos/model.py:68-86,128-176).  Weights stay owned by the
 * caller and must outlive the handle.  max_rows = 2 * max batch (CFG doubles rows, generation_utils.py:192). */
int zn_create(const zn_config* cfg, const zn_weights* w, int32_t max_rows, zn_handle* out);
int zn_destroy(zn_handle h);
const char* zn_last_error(zn_handle h); /* h may be NULL: last creation error */
/* Bytes of one layer's KV cache [rows, max_len, 2, Hkv, hd] bf16 (_torch.py:305). */
size_t zn_kv_bytes_per_layer(const zn_config* cfg, int32_t rows, int32_t max_len);
/* Bytes of one Mamba2 layer's decode state (_mamba_ssm.py:65-86 -> Mamba2.allocate_inference_cache): conv_state
 * bf16 [rows, conv_dim, d_conv] followed by ssm_state bf16 [rows, nheads, headdim, d_state], one contiguous buffer
 * whose device pointer takes the layer's slot in zn_gen_begin's kv_layers_dev.  *conv_bytes (optional) receives the
 * offset of ssm_state. */
size_t zn_mamba_state_bytes_per_layer(const zn_config* cfg, int32_t rows, size_t* conv_bytes);

/* ---------------------------------------------------------------- generation (Zonos.generate, model.py:354-548) */
/* Binds the per-call state that zonos/model.py:410-463 builds: KV caches (one device pointer per layer, layout
 * [2B, max_len, 2, Hkv, hd] bf16 = TorchZonosBackbone.allocate_inference_cache), lengths_per_sample int32[2B]
 * (device, zeroed by the caller), the delay-patterned code buffer int32 [B, n_codebooks, t_total] with -1 for
 * unknown (model.py:414-420), the first column to write `offset0` = prefix_len + 1, cfg_scale and sampling. */
/* (zn_gen_begin discards the hipGraphs captured for the previous generation: call it only when that generation's steps have drained from
 * their stream - `Zonos.generate` synchronises before it returns.) */
int zn_gen_begin(zn_handle h, int32_t batch, const void* const* kv_layers_dev, int32_t max_len,
                 int32_t* lengths_dev, int32_t* delayed_codes_dev, int32_t t_total, int32_t offset0,
                 int32_t max_new_tokens, float cfg_scale, const zn_sampling* sp, zn_stream stream);
/* prefill_static (generation_utils.py:206-244) + _compute_logits: hidden bf16 [2B, S, d] = [cond ‖ uncond]
 * conditioning concatenated with embed(delayed[..., :prefix+1]); fills KV positions [0,S), lengths += S and
 * leaves the CFG-mixed fp32 logits [B, n_codebooks, vocab_head] in the handle. */
int zn_prefill(zn_handle h, const void* hidden_dev, int32_t S, zn_stream stream);
/* model.py:423-431: sample the first frame from the prefill logits (no repetition penalty, no logit bias) and
 * write it into column offset0 where that column is -1. */
int zn_sample_first(zn_handle h, zn_stream stream);
/* n iterations of the hot loop (model.py:467-502): embed column offset-1, 26 blocks, heads, CFG, logit bias,
 * repetition penalty, sample, EOS bookkeeping (tensor_ops.py:155-211), frame write (tensor_ops.py:12-53),
 * offsets += 1.  Asynchronous; replays one hipGraph per step. */
int zn_decode_steps(zn_handle h, int32_t n, zn_stream stream);
/* 1 if the decode step is currently replayed as an instantiated hipGraph, 0 if launched kernel by kernel. */
int zn_graph_active(zn_handle h);
/* 1 if the decode steps of the generation begun by zn_gen_begin run the persistent kernels (batch 1 on a model whose shapes they
 * serve), 0 if every op is a launch of its own.  Both paths give bit-identical results at the Zonos-v0.1 shapes (same tiles, same
 * summation order in every GEMV; ONE arithmetic for the decode attention on every path: scores on the matrix cores, contexts of one
 * 512-key block accumulated in place, longer ones block by block with the partials combined in block order). */
int zn_decode_path(zn_handle h);
/* Which kernels served the decode step enqueued last: 0 = one launch per op, 1 = one attention launch (two beyond 512 keys) + one
 * persistent chain launch per block, 2 = the whole-step persistent kernel (every block of the step in one launch; contexts up to
 * 6144 keys: one attention workgroup per (row, kv head, 512-key block)).  All give bit-identical results. */
int zn_decode_path_detail(zn_handle h);
/* Hand-off timeouts are never silent: out[0] = bounded in-kernel hand-off waits that gave up and were reported on this handle (each voids
 * its generation; zn_all_stopped* returns the error), [1] generations begun, [2] batch-1 generations that ran the launches path because
 * an earlier timeout had demoted the handle, [3] 1 while the handle is demoted, [4] times it was re-armed (automatically after 4 clean
 * generations on the launches path, or by zn_debug_tune(8, 1)), [5] clean generations since the demotion, [6] the longest in-kernel hand-off
 * wait any whole-step launch of this handle measured, in microseconds (0: none beyond 0.1 ms; a pause of the device shows up here with its
 * length), [7] waits beyond 0.2 ms.  n <= 8 values are written; asking for [6], [7] synchronises with the device. */
int zn_get_counters(zn_handle h, int64_t* out, int32_t n);
/* Ends the generation begun by zn_gen_begin: releases the device's persistent-kernel tenancy (below) so that another handle's next
 * generation may take it.  The handle's state stays readable (zn_decode_path, zn_get_step_outputs); further steps need a new
 * zn_gen_begin.  Optional: zn_gen_begin of the same handle and zn_destroy release too. */
int zn_gen_end(zn_handle h);
/* One generation per device and process owns the persistent decode kernels (their in-launch hand-offs need every workgroup of the
 * grid resident: two such grids on one device could starve each other).  zn_gen_begin claims the device for `h`; a generation that
 * begins while another handle holds it runs the launches path (bit-identical results).  The two primitives are exported for the
 * host-side tests: try_claim returns 1 when `owner` holds the device afterwards, release 1 when `owner` held it. */
int zn_tenant_try_claim(int32_t device, const void* owner);
int zn_tenant_release(int32_t device, const void* owner);
/* (remaining_steps <= 0).all() of tensor_ops.py:95,102 — synchronises the stream. */
int zn_all_stopped(zn_handle h, int32_t* out, zn_stream stream);
/* The same check off the critical path: zn_all_stopped_begin queues the read-back of the loop state behind the steps enqueued
 * so far and returns; zn_all_stopped_end (called after the NEXT steps have been enqueued) waits for it and reports the
 * state as of _begin.  Steps that over-run a stop write only columns the caller's cut drops (model.py:511-528). */
int zn_all_stopped_begin(zn_handle h, zn_stream stream);
int zn_all_stopped_end(zn_handle h, int32_t* all_stopped_out);
/* Copies the fp32 logits the sampler last consumed ([B, n_codebooks, vocab_head], after CFG and logit bias) and
 * the raw sampled tokens int32 [B, n_codebooks] to device buffers (either may be NULL).  For parity tests. */
int zn_get_step_outputs(zn_handle h, float* logits_dev, int32_t* tokens_dev, zn_stream stream);
/* The caller rewrote cells of delayed_codes between two zn_decode_steps calls (teacher forcing): every step's sampler launch
 * also leaves the embedding of the column it wrote for the next step, and that embedding is stale now.  The next
 * zn_decode_steps call embeds the current column again. */
int zn_codes_changed(zn_handle h);
/* Test hook: at loop step `step` (0-based) force codebook-0 EOS by setting its logit to 1e4 (-1 = off). */
int zn_debug_force_eos(zn_handle h, int32_t step);
/* Test hook: replace the sampled raw tokens of call k (0 = first frame, k = loop step k-1) by
 * tokens_dev[k] (int32 [calls, B, n_codebooks], device) so that the EOS bookkeeping, frame writes and stop
 * cadence can be checked bit-exactly against the reference's recorded token stream.  NULL = off. */
int zn_debug_token_override(zn_handle h, const int32_t* tokens_dev, int32_t calls);
/* Test hook: 1 = batched prefill (default: MFMA GEMMs + tiled causal attention over all positions), 0 = position by
 * position through the decode kernels (both reproduce the reference's rounding points). */
int zn_debug_prefill_mode(zn_handle h, int32_t mode);
/* Tuning hook: target workgroup count of a GEMV class (0 in_proj, 1 out_proj, 2 fc1, 3 fc2, 4 heads); 5: longest context of the
 * fused attention launch (at most 512 keys, one block: beyond, every path walks the blocks with the split pass); 6: 1 = single-step
 * graphs only; 8: 2 = per-op launches instead of the persistent kernels (also ZN_CHAIN=0 at zn_create), 1 = back to the default and
 * re-arm a handle demoted by a hand-off timeout; 15: 2 = one chain launch per block instead of the whole-step kernel at batch 1; 18: 2 = block 0's in_proj as a launch before the
 * whole-step kernel instead of inside it; 16: 2 = the ticketed sampler launch instead of the one-workgroup step tail at batch 1 (the default for greedy decoding), 3 = the one-workgroup tail also with a temperature.  Every path gives bit-identical
 * results.  14: one-shot test hooks for the next generation (7: hand-off tags about to wrap; 9: the timeout word found set; 11: every
 * whole-step launch stops all its waves for 30 ms in block 2, as a paused device would; 13: forget the wait statistics of zn_get_counters
 * [6], [7] now).  Batches of 3 .. 8 utterances (5 .. 16 rows): 9: 2 = fc1's LayerNorm as a launch of its own (default: fc1 normalises its
 * activation chunks from statistics the preceding out_proj's epilogue left per 16-column tile - the same nn.LayerNorm with its sums taken
 * tile-wise, a bf16 ulp apart in about one value of a hundred); 19: value-column parts of the decode attention launches: 1 = never split,
 * 2 = always, other = the default (5 rows and more: 4 workgroups per (row, kv head) in the one-launch shape, 2 per (row, kv head, block)
 * beyond 512 keys; bit-identical either way).  Keys 0 .. 19. */
int zn_debug_tune(zn_handle h, int32_t key, int32_t value);
/* Diagnostic: workgroup 0 of every persistent chain launch records s_memrealtime (100 MHz) stamps of its phases into
 * stamps_dev [2 * n_layer][32] (NULL = off); rows n_layer.. hold the fused attention launch's (start, length known, scores
 * issued, scores done, P.V done, reduced, stored).  Stamp order per launch: input ready; then per op: results ready, arrived,
 * all arrived, next input ready; last: end. */
int zn_debug_chain_stamps(zn_handle h, uint64_t* stamps_dev);
/* Diagnostic: every decode step copies, per block, the residual stream after the block, the block's attention output, its
 * rotated queries, the gated MLP values m [rows][d_ff <= 4 d_model] (slots 3..6) and the residual stream after the attention
 * half (slot 7) into trace_dev [n_layer][8][rows][d_model] bf16 (NULL = off). */
int zn_debug_trace(zn_handle h, void* trace_dev);
/* Test/benchmark hook: add `bias` to the codebook-0 EOS logit on every step (-inf suppresses EOS so that all
 * max_new_tokens+7 steps run). */
int zn_debug_eos_bias(zn_handle h, float bias);

/* The backbone plugin seam (zonos/backbone/__init__.py:24-36; TorchZonosBackbone.forward _torch.py:213-238,
 * MambaSSMZonosBackbone.forward _mamba_ssm.py:88-119): hidden [rows][S][d] -> out [rows][S][d] after the final norm, the
 * caches advanced by S positions.  caches_dev: host array [n_layer] of device pointers (KV cache of an attention layer,
 * conv+SSM state buffer of a Mamba2 layer); every row holds `base` keys already (InferenceParams.seqlen_offset ==
 * lengths_per_sample in every reference call site).  S = 1 runs the decode kernels, S > 1 the batched prefill kernels
 * (Mamba2 layers: sequence conv + selective scan).  Needs no zn_gen_begin. */
int zn_op_backbone_forward(zn_handle h, const void* hidden_dev, void* out_dev, const void* const* caches_dev, int32_t max_len,
                           int32_t base, int32_t S, int32_t rows, zn_stream stream);

/* ---------------------------------------------------------------- measurement */
/* Average duration (HIP events on `stream`) of one of the decode step's weight-streaming kernels over `iters`
 * launches that cycle through the layers' weights, and its algorithmic bytes per launch (the weight matrix).
 * which: 0 = LayerNorm+fc1+SiLU-gate (5..16 rows: the one launch a decode step makes there, fc1 normalising from the statistics its
 * producer left - that producer runs once outside the timed loop; zn_debug_tune(9, 2): the LayerNorm launch + fc1), 1 = fc2+residual, 2 = out_proj+residual, 3 = LayerNorm+heads,
 * 4 = LayerNorm+in_proj+RoPE+KV-append (into a scratch cache), 5 = the persistent post-attention chain of one block
 * (out_proj twice, LayerNorm+fc1+SiLU-gate, fc2, next block's LayerNorm+in_proj+RoPE+KV-append in ONE launch: batch 1 only;
 * bytes = those four weight matrices, out_proj counted once), 6 = the whole-step kernel (every block of a decode step and the heads in
 * ONE launch, batch 1 only) on scratch KV caches of its own holding `ctx` keys per row; bytes = every weight the step reads once
 * (in_proj of block 0 included: the launch's pre-block; excluded under zn_debug_tune(18, 2), where it is a launch of its own) + K/V of
 * ctx keys read and one row written per layer.
 * rows: bits 0-7 = activation rows; bit 8 = keep streaming layer 0's weights (cache-hot variant); bits 16-30 = ctx for which == 6
 * (0 = 450, the mean context of a 10 s utterance). */
int zn_bench_kernel(zn_handle h, int32_t which, int32_t rows, int32_t iters, float* ms_per_launch,
                    double* bytes_per_launch, zn_stream stream);

/* ---------------------------------------------------------------- single ops (parity tests call these) */
/* nn.LayerNorm + nn.Linear(no bias): out[r,n] = bf16(sum_k LN(x)[r,k] W[n,k]); ln_w == NULL skips the norm. */
int zn_op_linear(zn_handle h, const void* x_dev, const void* ln_w, const void* ln_b, const void* W_dev,
                 void* out_dev, int32_t rows, int32_t N, int32_t K, zn_stream stream);
/* Prefix-conditioner pieces (zonos/conditioning.py): nn.Linear with bias (Conditioner.project, :52-60; bias added in fp32
 * before the one bf16 rounding), nn.Embedding row gather (:364-365,467), FourierConditioner.apply_cond (:436-441), nn.SiLU. */
int zn_op_linear_bias(zn_handle h, const void* x_dev, const void* W_dev, const void* bias_dev, void* out_dev, int32_t rows,
                      int32_t N, int32_t K, zn_stream stream);
int zn_op_gather_rows(zn_handle h, const void* table_dev, const int32_t* ids_dev, void* out_dev, int32_t n, int32_t d,
                      int32_t table_rows, int32_t id_offset, zn_stream stream);
int zn_op_fourier(zn_handle h, const float* x_dev, const void* weight_dev, void* out_dev, int32_t n, int32_t in_dim,
                  int32_t half, float min_val, float max_val, zn_stream stream);
int zn_op_silu(zn_handle h, const void* x_dev, void* out_dev, int64_t n, zn_stream stream);
/* nn.LayerNorm (_torch.py:155 norm_f): bf16 [rows, d] -> bf16 [rows, d], fp32 statistics. */
int zn_op_layernorm(zn_handle h, const void* x_dev, const void* w_dev, const void* b_dev, void* out_dev, int32_t rows,
                    int32_t d, zn_stream stream);
/* One decode step of TransformerBlock `layer` (_torch.py:307-328) on x bf16 [rows, d] in place, reading and
 * appending to kv_dev at position lengths_dev[r]; ext_dev (optional, int32[rows]) = number of keys the
 * reference's CPU flash-attention block sees for that row (prefill emulation), NULL = lengths+1. */
int zn_op_layer_decode(zn_handle h, int32_t layer, void* x_dev, void* kv_dev, int32_t max_len,
                       const int32_t* lengths_dev, const int32_t* ext_dev, int32_t rows, zn_stream stream);
/* Decode attention alone (_torch.py:413-417): q bf16 [rows, Hq*hd] (post-RoPE), kv [rows, max_len, 2, Hkv, hd] holding
 * lengths[r]+1 keys -> out bf16 [rows, Hq*hd]; reproduces the CPU flash-attention rounding points. */
int zn_op_attn_decode(zn_handle h, const void* q_dev, const void* kv_dev, int32_t max_len, const int32_t* lengths_dev,
                      const int32_t* ext_dev, void* out_dev, int32_t rows, zn_stream stream);
/* Causal prefill attention alone (_torch.py:413-417 with is_causal=True, the S > 1 call): q bf16 [rows, positions, Hq*hd]
 * (post-RoPE), kv [rows, max_len, 2, Hkv, hd] holding the keys of positions 0..positions-1 -> out bf16
 * [rows, positions, Hq*hd]; same CPU flash-attention rounding points as zn_op_attn_decode, incl. its query-block split. */
int zn_op_attn_prefill(zn_handle h, const void* q_dev, const void* kv_dev, int32_t max_len, void* out_dev, int32_t positions,
                       int32_t rows, zn_stream stream);
/* mamba_ssm layer_norm_fn(prenorm=True) of the hybrid Block: s = h + res (fp32), res <- bf16(s) in place (res NULL:
 * s = h), out = bf16(LayerNorm(s)).  h/out bf16 [rows, d].  flags bit 0: RMSNorm instead of LayerNorm (rms_norm; b may be
 * NULL), bit 1: res is float [rows, d] and keeps the unrounded sum (residual_in_fp32). */
int zn_op_add_layernorm(zn_handle h, const void* hidden, void* res, const void* w, const void* b, void* out, int32_t rows,
                        int32_t d, float eps, int32_t flags, zn_stream stream);
/* One token through the Mamba2 mixer of hybrid layer `layer` (mamba_ssm Mamba2.step): x bf16 [rows, d] (already
 * normalised), state = the layer's zn_mamba_state_bytes_per_layer buffer (updated), out bf16 [rows, d]. */
int zn_op_mamba_step(zn_handle h, int32_t layer, const void* x, void* state, void* out, int32_t rows, zn_stream stream);
/* embed_codes_static (codec_utils.py:37): codes int32 [B, n_codebooks] -> bf16 [B, d], sequential bf16 adds. */
int zn_op_embed(zn_handle h, const int32_t* codes_dev, void* out_dev, int32_t batch, zn_stream stream);
/* sample_from_logits (sampling.py:166-231) on fp32 logits [B, n_codebooks, vocab_head]; recent int32
 * [B, n_codebooks, window] or NULL; tokens int32 [B, n_codebooks]; probs_out (optional) receives the filtered
 * probabilities the Gumbel-max draw uses. */
int zn_op_sample(zn_handle h, const float* logits_dev, const int32_t* recent_dev, int32_t window,
                 const zn_sampling* sp, uint64_t draw_index, int32_t* tokens_dev, float* probs_out_dev,
                 int32_t batch, zn_stream stream);

/* ---------------------------------------------------------------- DAC decode (autoencoder.py:119-170) */
typedef struct zn_dac_config { /* transformers DacConfig fields used by decode / encode */
  int32_t n_codebooks, codebook_size, codebook_dim, hidden_size, decoder_hidden_size;
  int32_t n_ratios; int32_t ratios[8]; /* upsampling_ratios, e.g. 8,8,4,2 (downsampling_ratios = reversed) */
  int32_t encoder_hidden_size;         /* 64; 0 = no encoder */
} zn_dac_config;
typedef struct zn_dac_tensor { const char* name; const float* data_dev; int64_t numel; } zn_dac_tensor;
/* Weights by their transformers state-dict names (fp32, device).  The library re-lays them out once.  Decode runs on the bf16 matrix
 * cores with three-term fp32 operands (zn_conv3_kernels.h; waveform RMS error vs the reference 1e-6, as on the fp32 matrix cores);
 * ZONOS_DAC_CONV=fp32 in the environment of this call keeps the decoder on the fp32 matrix cores (development: A/B of the two). */
int zn_dac_create(const zn_dac_config* cfg, const zn_dac_tensor* tensors, int32_t n_tensors, zn_dac* out);
int zn_dac_destroy(zn_dac d);
const char* zn_dac_last_error(zn_dac d);
/* DACAutoencoder.decode: codes int32 [B, n_codebooks, T] -> wav fp32 [B, 1, hop*T]. */
int zn_dac_decode(zn_dac d, const int32_t* codes_dev, int32_t batch, int32_t T, float* wav_dev, zn_stream stream);
/* DACAutoencoder.encode (zonos/autoencoder.py:103-117 -> DacModel.encode): wav fp32 [B, T] at the codec rate, T a
 * positive multiple of the hop (preprocess pads) -> codes int32 [B, n_codebooks, T / hop].  Needs the encoder.* and
 * quantizer.quantizers.{i}.in_proj tensors at zn_dac_create. */
int zn_dac_encode(zn_dac d, const float* wav_dev, int32_t batch, int32_t T, int32_t* codes_dev, zn_stream stream);

/* ---------------------------------------------------------------- speaker embedding (zonos/speaker_cloning.py) */
/* ResNet293_based (speaker_cloning.py:419-472: ResNet293 of SimAM blocks -> ASP -> bottleneck Linear) and the LDA Linear
 * of SpeakerEmbeddingLDA (:800-883), fp32.  Tensors by the reference's state-dict names (front.*, pooling.*,
 * bottleneck.*; optionally lda.weight / lda.bias), device pointers that must outlive the handle; BatchNorm is folded
 * when the handle is built.  Runs once per speaker, outside the decode loop. */
int zn_spk_create(const zn_dac_tensor* tensors, int32_t n_tensors, zn_spk* out);
int zn_spk_destroy(zn_spk d);
const char* zn_spk_last_error(zn_spk d);
/* feat fp32 [B, n_mels, T]: mean-normalised log-mel features (what logFbankCal returns, speaker_cloning.py:81-87), T >= 8
 * -> emb fp32 [B, emb_dim] (bottleneck output) and, if lda_out != NULL, lda_out fp32 [B, lda_dim]. */
int zn_spk_embed(zn_spk d, const float* feat_dev, int32_t batch, int32_t T, float* emb_dev, float* lda_out_dev, zn_stream stream);

#ifdef __cplusplus
}
#endif
#endif
