// DAC decode (zonos/autoencoder.py:119-170 -> transformers modeling_dac.py:347-371 from_codes, :407-441 DacDecoder).
//
// fp32 end to end (the reference's CPU path disables autocast; waveform RMS error budget 1e-4, so no bf16).  Every
// convolution is an implicit GEMM on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32): activations are kept
// channels-last [B][T][C] so that the reduction index (input channel) is contiguous, a workgroup owns a 128-time x
// (NT*32)-channel output tile, the input rows it needs (with the dilation halo) are staged once per 16-channel chunk in
// LDS with the Snake activation applied on the way in, and all taps of the chunk's weights sit next to them.
// ConvTranspose1d(k = 2s, stride s) is s independent 2-tap GEMMs (one per output phase).
//
// DAC encode (zonos/autoencoder.py:103-117 -> modeling_dac.py:583-608 DacModel.encode, :444-473 DacEncoder, :212-233
// DacEncoderBlock, :283-345 residual VQ, :123-172 DacVectorQuantize).  The strided Conv1d(k = 2s, stride s, pad
// ceil(s/2)) of an encoder block reads the channels-last input as rows of s time steps ([T/s][s*C], the same memory):
// out[t] touches rows t-1, t, t+1, i.e. it is a 3-tap stride-1 GEMM with K = s*C whose re-laid-out weight is zero where
// the 2s-tap window does not reach (1.5x the MACs of the minimum on layers that hold ~10 % of the encoder's work, no
// gather, and the zero padding is the kernel's ordinary row-range check).  The residual VQ runs one workgroup per frame.
#include "../../include/zonos_hip.h"
#include "zn_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "zn_conv_kernels.h"
#include "zn_conv3_kernels.h"

static thread_local std::string g_dac_err;

// z[b][t][c] = sum_i table[i][codes[b][i][t]][c]   (from_codes: (0.0 + o_0) + o_1 + ... in codebook order)
__global__ __launch_bounds__(256) void dac_codes_kernel(const int* codes, const float* table, float* z, int nq, int T, int hidden, int cbsize) {
  const int t = blockIdx.x, b = blockIdx.y;
  for (int c = threadIdx.x; c < hidden; c += 256) {
    float acc = 0.f;
    for (int i = 0; i < nq; ++i) {
      int code = codes[((size_t)b * nq + i) * T + t];
      code = code < 0 ? 0 : (code >= cbsize ? cbsize - 1 : code);
      acc = acc + table[((size_t)i * cbsize + code) * hidden + c];
    }
    z[((size_t)b * T + t) * hidden + c] = acc;
  }
}
// table[i][code][c] = bias_i[c] + sum_j W_i[c][j] * E_i[code][j]   (Embedding -> 1x1 conv, modeling_dac.py:366-369)
__global__ void dac_table_kernel(const float* emb, const float* w, const float* bias, float* table, int cbsize, int cbdim, int hidden) {
  const int code = blockIdx.x;
  for (int c = threadIdx.x; c < hidden; c += blockDim.x) {
    float acc = 0.f;
    for (int j = 0; j < cbdim; ++j) acc = fmaf(w[(size_t)c * cbdim + j], emb[(size_t)code * cbdim + j], acc);
    table[(size_t)code * hidden + c] = acc + bias[c];
  }
}
// conv weight [Cout][Cin][K] -> [K][Cin][CoutPad]
__global__ void dac_wconv_kernel(const float* w, float* o, int Cout, int Cin, int K, int CoutPad) {
  const size_t n = (size_t)K * Cin * CoutPad;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int co = i % CoutPad, ci = (i / CoutPad) % Cin, k = i / ((size_t)CoutPad * Cin);
    o[i] = co < Cout ? w[((size_t)co * Cin + ci) * K + k] : 0.f;
  }
}
// conv-transpose weight [Cin][Cout][2s] -> [phase p][tap j][Cin][CoutPad], tap 0 <-> k = p (input q), tap 1 <-> k = p + s (input q-1)
__global__ void dac_wconvt_kernel(const float* w, float* o, int Cin, int Cout, int s, int CoutPad) {
  const size_t n = (size_t)s * 2 * Cin * CoutPad;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int co = i % CoutPad, ci = (i / CoutPad) % Cin, j = (i / ((size_t)CoutPad * Cin)) % 2, p = i / ((size_t)CoutPad * Cin * 2);
    o[i] = co < Cout ? w[((size_t)ci * Cout + co) * (2 * s) + p + j * s] : 0.f;
  }
}
// last layer: Snake -> Conv1d(C -> 1, k = 7, pad 3) -> tanh  (modeling_dac.py:438-441) on the VALU.  128 output times per workgroup; the
// 134 input rows are requested in one batch of 16-byte loads, activated on the way into LDS (row stride C + 1: conflict-free column
// walks); two threads per output time take the even and the odd channels (7 x C/2 fused multiply-adds each, in tap-then-channel order)
// and meet through one DPP add.
#define DAC_FIN_T 128
#define DAC_FIN_MAXP 16
__global__ __launch_bounds__(256) void dac_final_kernel(const float* in, const float* alpha, const float* w /*[1][C][7]*/, const float* bias,
                                                        float* out, int T, int C) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_x = smem;                         // [DAC_FIN_T + 6][C + 1]
  float* s_w = smem + (DAC_FIN_T + 6) * (C + 1);   // [7][C]
  const int t0 = blockIdx.x * DAC_FIN_T, b = blockIdx.y, tid = threadIdx.x;
  const float* inb = in + (size_t)b * T * C;
  const int c4n = C / 4, npiece = (DAC_FIN_T + 6) * c4n;
  f32x4 r[DAC_FIN_MAXP];
#pragma unroll
  for (int j = 0; j < DAC_FIN_MAXP; ++j) {
    const int i = tid + j * 256, row = i / c4n, c4 = i - row * c4n;
    int t = t0 - 3 + row;
    t = t < 0 ? 0 : (t >= T ? T - 1 : t);
    if (i < npiece) r[j] = *(const f32x4*)(inb + (size_t)t * C + c4 * 4);
  }
  for (int i = tid; i < 7 * C; i += 256) { const int k = i / C, c = i - k * C; s_w[i] = w[(size_t)c * 7 + k]; }
  float* s_al = s_w + 7 * C;                 // [C] alpha, [C] 1 / (alpha + 1e-9)
  for (int c = tid; c < C; c += 256) { const float al = alpha[c]; s_al[c] = al; s_al[C + c] = 1.0f / (al + 1e-9f); }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < DAC_FIN_MAXP; ++j) {
    const int i = tid + j * 256, row = i / c4n, c4 = i - row * c4n;
    if (i < npiece) {
      const int t = t0 - 3 + row;
      const float* al = s_al + c4 * 4;       // (the LDS offsets of s_w / s_al are not 16-byte aligned: scalar reads)
      const float x[4] = {r[j].x, r[j].y, r[j].z, r[j].w};
      float* d = s_x + row * (C + 1) + c4 * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e)            // Snake (modeling_dac.py:98) with the decoder kernels' sin^2 (zn_conv3_kernels.h); zero padding outside [0, T)
        d[e] = (t >= 0 && t < T) ? x[e] + al[C + e] * c3_sin2(al[e] * x[e]) : 0.f;
    }
  }
  __syncthreads();
  const int tl = tid >> 1, hf = tid & 1, t = t0 + tl;
  float acc = 0.f;
  for (int k = 0; k < 7; ++k) {
    const float* xr = s_x + (tl + k) * (C + 1) + hf;
    const float* wr = s_w + k * C + hf;
    for (int c = 0; c < C; c += 2) acc = fmaf(xr[c], wr[c], acc);
  }
  acc += dpp_mov<ZN_DPP_XOR1>(acc);
  if (hf == 0 && t < T) out[(size_t)b * T + t] = tanhf(acc + bias[0]);
}

// ------------------------------------------------------------------------------------------------ encoder pieces
// strided conv weight [Cout][C][2s] -> [3 row taps][s*C][CoutPad]: row tap j (input row t-1+j), element e of the row,
// channel ci <-> conv tap (j-1)*s + e + pad, zero where that tap does not exist (modeling_dac.py:222-224)
__global__ void dac_wstride_kernel(const float* w, float* o, int Cout, int C, int s, int pad, int CoutPad) {
  const size_t n = (size_t)3 * s * C * CoutPad;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int co = i % CoutPad, ci = (i / CoutPad) % C, e = (i / ((size_t)CoutPad * C)) % s, j = i / ((size_t)CoutPad * C * s);
    const int tap = (j - 1) * s + e + pad;
    o[i] = (co < Cout && tap >= 0 && tap < 2 * s) ? w[((size_t)co * C + ci) * (2 * s) + tap] : 0.f;
  }
}
__global__ void dac_tile_kernel(const float* a, float* o, int C, int reps) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < C * reps; i += gridDim.x * blockDim.x) o[i] = a[i % C];
}
// encoder.conv1: Conv1d(1 -> C, k = 7, pad 3) on the waveform (modeling_dac.py:451,465); out channels-last [B][T][C]
__global__ __launch_bounds__(256) void dac_enc_conv1_kernel(const float* wav, const float* w /*[C][1][7]*/, const float* bias, float* out, int T, int C) {
  const int b = blockIdx.y;
  const size_t idx = blockIdx.x * (size_t)256 + threadIdx.x;       // over T * C
  if (idx >= (size_t)T * C) return;
  const int t = idx / C, co = idx % C;
  const float* x = wav + (size_t)b * T;
  float acc = bias[co];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int tt = t + k - 3;
    acc = fmaf(w[co * 7 + k], (tt >= 0 && tt < T) ? x[tt] : 0.f, acc);
  }
  out[(size_t)b * T * C + idx] = acc;
}
// F.normalize rows of a codebook (x / max(||x||, 1e-12)) and their squared norms (modeling_dac.py:163,167)
__global__ void dac_cbnorm_kernel(const float* cb, float* cbn, float* cbsq, int n, int dim) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float ss = 0.f;
  for (int j = 0; j < dim; ++j) ss += cb[(size_t)i * dim + j] * cb[(size_t)i * dim + j];
  const float nrm = fmaxf(sqrtf(ss), 1e-12f);
  float s2 = 0.f;
  for (int j = 0; j < dim; ++j) { const float v = cb[(size_t)i * dim + j] / nrm; cbn[(size_t)i * dim + j] = v; s2 += v * v; }
  cbsq[i] = s2;
}
// Residual vector quantisation of one latent frame (modeling_dac.py:310-340 over :123-172): for each codebook in order:
// e = in_proj(residual) (Conv1d 1x1 hidden -> dim), nearest code by dist = -(|e_n|^2 - 2 e_n . c_n) + |c_n|^2 on
// L2-normalised vectors (first maximum wins), residual -= out_proj(codebook[idx]).  256 threads per frame.
#define DAC_RVQ_MAXDIM 8
struct RvqArgs {
  const float* z;            // [B][T][hidden] encoder output
  const float* const* w_in;  // per codebook [dim][hidden]
  const float* const* b_in;  // [dim]
  const float* const* cb;    // [size][dim] raw
  const float* const* cbn;   // [size][dim] normalised
  const float* const* cbsq;  // [size]
  const float* const* w_out; // [hidden][dim]
  const float* const* b_out; // [hidden]
  int* codes;                // [B][nq][T]
  int nq, T, hidden, dim, size;
};
__global__ __launch_bounds__(256) void dac_rvq_kernel(RvqArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* res = smem;                       // [hidden]
  __shared__ float s_part[4][DAC_RVQ_MAXDIM];
  __shared__ float s_e[DAC_RVQ_MAXDIM];
  __shared__ float s_bv[4];
  __shared__ int s_bi[4];
  __shared__ int s_idx;
  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* zr = a.z + ((size_t)b * a.T + t) * a.hidden;
  for (int c = tid; c < a.hidden; c += 256) res[c] = zr[c];
  __syncthreads();
  for (int q = 0; q < a.nq; ++q) {
    // projected latent e[j] = b_in[j] + sum_c W_in[j][c] * res[c]
    float pj[DAC_RVQ_MAXDIM];
#pragma unroll
    for (int j = 0; j < DAC_RVQ_MAXDIM; ++j) pj[j] = 0.f;
    for (int c = tid; c < a.hidden; c += 256) {
      const float r = res[c];
#pragma unroll
      for (int j = 0; j < DAC_RVQ_MAXDIM; ++j) if (j < a.dim) pj[j] = fmaf(a.w_in[q][(size_t)j * a.hidden + c], r, pj[j]);
    }
#pragma unroll
    for (int j = 0; j < DAC_RVQ_MAXDIM; ++j) { pj[j] = wave_sum(pj[j]); if (lane == 0) s_part[wave][j] = pj[j]; }
    __syncthreads();
    if (tid == 0) {
      float e[DAC_RVQ_MAXDIM], ss = 0.f;
      for (int j = 0; j < a.dim; ++j) { e[j] = ((s_part[0][j] + s_part[1][j]) + (s_part[2][j] + s_part[3][j])) + a.b_in[q][j]; ss += e[j] * e[j]; }
      const float nrm = fmaxf(sqrtf(ss), 1e-12f);
      for (int j = 0; j < a.dim; ++j) s_e[j] = e[j] / nrm;
    }
    __syncthreads();
    float en[DAC_RVQ_MAXDIM], l2 = 0.f;
#pragma unroll
    for (int j = 0; j < DAC_RVQ_MAXDIM; ++j) { en[j] = j < a.dim ? s_e[j] : 0.f; l2 += en[j] * en[j]; }
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int code = tid; code < a.size; code += 256) {
      float mm = 0.f;
#pragma unroll
      for (int j = 0; j < DAC_RVQ_MAXDIM; ++j) if (j < a.dim) mm = fmaf(en[j], a.cbn[q][(size_t)code * a.dim + j], mm);
      const float dist = -(l2 - 2.0f * mm) + a.cbsq[q][code];
      if (dist > best) { best = dist; bi = code; }           // ascending codes per thread: the first maximum stays
    }
    // block argmax, lowest index among equal maxima
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float ov = __shfl_xor(best, off); const int oi = __shfl_xor(bi, off);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { s_bv[wave] = best; s_bi[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      float bv = s_bv[0]; int ix = s_bi[0];
      for (int w = 1; w < 4; ++w) if (s_bv[w] > bv || (s_bv[w] == bv && s_bi[w] < ix)) { bv = s_bv[w]; ix = s_bi[w]; }
      s_idx = ix;
      a.codes[((size_t)b * a.nq + q) * a.T + t] = ix;
    }
    __syncthreads();
    const int ix = s_idx;
    for (int c = tid; c < a.hidden; c += 256) {
      float o = 0.f;
      for (int j = 0; j < a.dim; ++j) o = fmaf(a.w_out[q][(size_t)c * a.dim + j], a.cb[q][(size_t)ix * a.dim + j], o);
      res[c] = res[c] - (o + a.b_out[q][c]);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ host
struct ConvLayer { float *w = nullptr; bf16_t* w3 = nullptr; const float *bias = nullptr, *alpha = nullptr; int Cin = 0, Cout = 0, CoutPad = 0, K = 0, dil = 1, stride = 0; };
struct zn_dac_s {
  zn_dac_config cfg;
  float* table = nullptr;
  ConvLayer conv1;
  struct Block { ConvLayer convt; ConvLayer c1[3], c2[3]; } blocks[8];
  const float *fin_alpha = nullptr, *fin_w = nullptr, *fin_b = nullptr;
  int fin_C = 0;
  float* buf[3] = {nullptr, nullptr, nullptr};
  size_t buf_elems = 0;
  // decode on the bf16 matrix cores with three-term operands (zn_conv3_kernels.h): the default.  ZONOS_DAC_CONV=fp32 in the environment of
  // zn_dac_create keeps the decoder on the fp32 matrix cores (zn_conv_kernels.h; development: A/B of the two paths)
  bool split3 = true;
  // encoder (optional: present when the state dict carries encoder.* and quantizer in_proj tensors)
  bool has_encoder = false;
  const float *enc_w1 = nullptr, *enc_b1 = nullptr;
  struct EncBlock { ConvLayer c1[3], c2[3]; ConvLayer down; float* alpha_tiled = nullptr; int stride = 0; } eblocks[8];
  ConvLayer enc_conv2;
  const float** rvq_ptrs = nullptr;      // device: 7 arrays of n_codebooks pointers (w_in, b_in, cb, cbn, cbsq, w_out, b_out)
  std::vector<float*> owned;
  std::string err;
};

#define DFAIL(d, code, ...) do { char _b[512]; snprintf(_b, sizeof _b, __VA_ARGS__); if (d) (d)->err = _b; else g_dac_err = _b; return (code); } while (0)
#define DHIP(d, call) do { hipError_t _e = (call); if (_e != hipSuccess) DFAIL(d, ZN_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(_e)); } while (0)

extern "C" const char* zn_dac_last_error(zn_dac d) { return d ? d->err.c_str() : g_dac_err.c_str(); }

extern "C" int zn_dac_destroy(zn_dac d) {
  if (!d) return ZN_OK;
  for (float* p : d->owned) (void)hipFree(p);
  for (float* p : d->buf) if (p) (void)hipFree(p);
  delete d;
  return ZN_OK;
}

static int pad32(int c) { return zn_conv_pad(c); }

static int make_conv(zn_dac d, const std::map<std::string, const zn_dac_tensor*>& t, const std::string& wname, const std::string& bname,
                     const char* aname, int Cin, int Cout, int K, int dil, ConvLayer& L, bool dec3 = false) {
  auto w = t.find(wname), b = t.find(bname);
  if (w == t.end() || b == t.end()) DFAIL(d, ZN_ERR_ARG, "missing tensor %s / %s", wname.c_str(), bname.c_str());
  if (w->second->numel != (int64_t)Cout * Cin * K || b->second->numel != Cout) DFAIL(d, ZN_ERR_ARG, "shape mismatch for %s", wname.c_str());
  L.Cin = Cin; L.Cout = Cout; L.K = K; L.dil = dil; L.CoutPad = pad32(Cout); L.bias = b->second->data_dev;
  if (aname) {
    auto a = t.find(aname);
    if (a == t.end() || a->second->numel != Cin) DFAIL(d, ZN_ERR_ARG, "missing/mis-shaped tensor %s", aname);
    L.alpha = a->second->data_dev;
  }
  if (Cin % DAC_KC) DFAIL(d, ZN_ERR_UNSUPPORTED, "Cin %d not a multiple of %d", Cin, DAC_KC);
  DHIP(d, hipMalloc(&L.w, (size_t)K * Cin * L.CoutPad * sizeof(float)));
  d->owned.push_back(L.w);
  hipLaunchKernelGGL(dac_wconv_kernel, dim3(512), dim3(256), 0, 0, w->second->data_dev, L.w, Cout, Cin, K, L.CoutPad);
  if (dec3) {
    const int cp = zn_conv3_pad(Cout);
    DHIP(d, hipMalloc(&L.w3, (size_t)K * Cin * cp * 3 * sizeof(bf16_t)));
    d->owned.push_back((float*)L.w3);
    hipLaunchKernelGGL(dac_w3conv_kernel, dim3(512), dim3(256), 0, 0, w->second->data_dev, L.w3, Cout, Cin, K, cp);
  }
  return ZN_OK;
}

extern "C" int zn_dac_create(const zn_dac_config* cfg, const zn_dac_tensor* tensors, int32_t n, zn_dac* out) {
  if (!cfg || !tensors || !out || n < 1) DFAIL((zn_dac) nullptr, ZN_ERR_ARG, "zn_dac_create: null argument");
  *out = nullptr;
  if (cfg->n_ratios < 1 || cfg->n_ratios > 8 || cfg->n_codebooks < 1 || cfg->hidden_size % DAC_KC || cfg->decoder_hidden_size % DAC_KC)
    DFAIL((zn_dac) nullptr, ZN_ERR_ARG, "zn_dac_create: unsupported configuration");
  std::map<std::string, const zn_dac_tensor*> t;
  for (int i = 0; i < n; ++i) if (tensors[i].name && tensors[i].data_dev) t[tensors[i].name] = &tensors[i];
  zn_dac d = new zn_dac_s();
  d->cfg = *cfg;
  { const char* e = getenv("ZONOS_DAC_CONV"); d->split3 = !(e && !strcmp(e, "fp32")); }
  int rc = ZN_OK;
  auto fail = [&](int code) { g_dac_err = d->err; zn_dac_destroy(d); return code; };
  // codebook tables
  const size_t tsz = (size_t)cfg->codebook_size * cfg->hidden_size;
  if (hipMalloc(&d->table, (size_t)cfg->n_codebooks * tsz * sizeof(float)) != hipSuccess) { d->err = "hipMalloc(table) failed"; return fail(ZN_ERR_HIP); }
  d->owned.push_back(d->table);
  for (int i = 0; i < cfg->n_codebooks; ++i) {
    const std::string q = "quantizer.quantizers." + std::to_string(i) + ".";
    auto e = t.find(q + "codebook.weight"), w = t.find(q + "out_proj.weight"), b = t.find(q + "out_proj.bias");
    if (e == t.end() || w == t.end() || b == t.end()) { d->err = "missing quantizer tensors for codebook " + std::to_string(i); return fail(ZN_ERR_ARG); }
    if (e->second->numel != (int64_t)cfg->codebook_size * cfg->codebook_dim || w->second->numel != (int64_t)cfg->hidden_size * cfg->codebook_dim) {
      d->err = "quantizer tensor shape mismatch"; return fail(ZN_ERR_ARG);
    }
    hipLaunchKernelGGL(dac_table_kernel, dim3(cfg->codebook_size), dim3(256), 0, 0, e->second->data_dev, w->second->data_dev, b->second->data_dev,
                       d->table + (size_t)i * tsz, cfg->codebook_size, cfg->codebook_dim, cfg->hidden_size);
  }
  int c = cfg->decoder_hidden_size;
  if ((rc = make_conv(d, t, "decoder.conv1.weight", "decoder.conv1.bias", nullptr, cfg->hidden_size, c, 7, 1, d->conv1, d->split3))) return fail(rc);
  for (int bi = 0; bi < cfg->n_ratios; ++bi) {
    const std::string p = "decoder.block." + std::to_string(bi) + ".";
    const int s = cfg->ratios[bi], co = c / 2;
    auto& B = d->blocks[bi];
    auto w = t.find(p + "conv_t1.weight"), b = t.find(p + "conv_t1.bias"), al = t.find(p + "snake1.alpha");
    if (w == t.end() || b == t.end() || al == t.end() || w->second->numel != (int64_t)c * co * 2 * s || s < 2 || s % 2) {
      d->err = "missing/mis-shaped conv_t1 tensors in block " + std::to_string(bi); return fail(ZN_ERR_ARG);
    }
    ConvLayer& T = B.convt;
    T.Cin = c; T.Cout = co; T.CoutPad = pad32(co); T.K = 2; T.stride = s; T.bias = b->second->data_dev; T.alpha = al->second->data_dev;
    if (c % DAC_KC) { d->err = "channel count not a multiple of 16"; return fail(ZN_ERR_UNSUPPORTED); }
    if (hipMalloc(&T.w, (size_t)s * 2 * c * T.CoutPad * sizeof(float)) != hipSuccess) { d->err = "hipMalloc failed"; return fail(ZN_ERR_HIP); }
    d->owned.push_back(T.w);
    hipLaunchKernelGGL(dac_wconvt_kernel, dim3(512), dim3(256), 0, 0, w->second->data_dev, T.w, c, co, s, T.CoutPad);
    if (d->split3) {
      const int cp = zn_conv3_pad(co);
      if (hipMalloc(&T.w3, (size_t)s * 2 * c * cp * 3 * sizeof(bf16_t)) != hipSuccess) { d->err = "hipMalloc failed"; return fail(ZN_ERR_HIP); }
      d->owned.push_back((float*)T.w3);
      hipLaunchKernelGGL(dac_w3convt_kernel, dim3(512), dim3(256), 0, 0, w->second->data_dev, T.w3, c, co, s, cp);
    }
    const int dils[3] = {1, 3, 9};
    for (int u = 0; u < 3; ++u) {
      const std::string r = p + "res_unit" + std::to_string(u + 1) + ".";
      const std::string a1 = r + "snake1.alpha", a2 = r + "snake2.alpha";
      if ((rc = make_conv(d, t, r + "conv1.weight", r + "conv1.bias", a1.c_str(), co, co, 7, dils[u], B.c1[u], d->split3))) return fail(rc);
      if ((rc = make_conv(d, t, r + "conv2.weight", r + "conv2.bias", a2.c_str(), co, co, 1, 1, B.c2[u], d->split3))) return fail(rc);
    }
    c = co;
  }
  auto fa = t.find("decoder.snake1.alpha"), fw = t.find("decoder.conv2.weight"), fb = t.find("decoder.conv2.bias");
  if (fa == t.end() || fw == t.end() || fb == t.end() || fw->second->numel != (int64_t)c * 7) { d->err = "missing/mis-shaped final conv tensors"; return fail(ZN_ERR_ARG); }
  d->fin_alpha = fa->second->data_dev; d->fin_w = fw->second->data_dev; d->fin_b = fb->second->data_dev; d->fin_C = c;
  if (c % 4 || (DAC_FIN_T + 6) * (c / 4) > DAC_FIN_MAXP * 256) { d->err = "final conv: channel count " + std::to_string(c) + " not supported (multiple of 4, <= 120)"; return fail(ZN_ERR_UNSUPPORTED); }
  // ---- encoder + residual VQ (DacEncoder, modeling_dac.py:444-473; DacResidualVectorQuantizer :283-345)
  if (t.count("encoder.conv1.weight") && t.count("quantizer.quantizers.0.in_proj.weight") && cfg->encoder_hidden_size > 0) {
    const int eh = cfg->encoder_hidden_size;
    auto w1 = t.find("encoder.conv1.weight"), b1 = t.find("encoder.conv1.bias");
    if (b1 == t.end() || w1->second->numel != (int64_t)eh * 7 || eh % 64) { d->err = "missing/mis-shaped encoder.conv1"; return fail(ZN_ERR_ARG); }
    d->enc_w1 = w1->second->data_dev; d->enc_b1 = b1->second->data_dev;
    int ch = eh;
    for (int bi = 0; bi < cfg->n_ratios; ++bi) {
      const std::string p = "encoder.block." + std::to_string(bi) + ".";
      const int st = cfg->ratios[cfg->n_ratios - 1 - bi];                 // downsampling_ratios = reversed upsampling ratios
      auto& E = d->eblocks[bi];
      E.stride = st;
      const int dils[3] = {1, 3, 9};
      for (int u = 0; u < 3; ++u) {
        const std::string r = p + "res_unit" + std::to_string(u + 1) + ".";
        const std::string a1 = r + "snake1.alpha", a2 = r + "snake2.alpha";
        if ((rc = make_conv(d, t, r + "conv1.weight", r + "conv1.bias", a1.c_str(), ch, ch, 7, dils[u], E.c1[u]))) return fail(rc);
        if ((rc = make_conv(d, t, r + "conv2.weight", r + "conv2.bias", a2.c_str(), ch, ch, 1, 1, E.c2[u]))) return fail(rc);
      }
      auto w = t.find(p + "conv1.weight"), b = t.find(p + "conv1.bias"), al = t.find(p + "snake1.alpha");
      if (w == t.end() || b == t.end() || al == t.end() || w->second->numel != (int64_t)2 * ch * ch * 2 * st || al->second->numel != ch) {
        d->err = "missing/mis-shaped strided conv tensors in encoder block " + std::to_string(bi); return fail(ZN_ERR_ARG);
      }
      ConvLayer& D = E.down;
      D.Cin = st * ch; D.Cout = 2 * ch; D.CoutPad = pad32(2 * ch); D.K = 3; D.dil = 1; D.bias = b->second->data_dev;
      if (hipMalloc(&D.w, (size_t)3 * st * ch * D.CoutPad * sizeof(float)) != hipSuccess || hipMalloc(&E.alpha_tiled, (size_t)st * ch * sizeof(float)) != hipSuccess) {
        d->err = "hipMalloc failed"; return fail(ZN_ERR_HIP);
      }
      d->owned.push_back(D.w); d->owned.push_back(E.alpha_tiled);
      hipLaunchKernelGGL(dac_wstride_kernel, dim3(512), dim3(256), 0, 0, w->second->data_dev, D.w, 2 * ch, ch, st, (st + 1) / 2, D.CoutPad);
      hipLaunchKernelGGL(dac_tile_kernel, dim3(64), dim3(256), 0, 0, al->second->data_dev, E.alpha_tiled, ch, st);
      D.alpha = E.alpha_tiled;
      ch *= 2;
    }
    if ((rc = make_conv(d, t, "encoder.conv2.weight", "encoder.conv2.bias", "encoder.snake1.alpha", ch, cfg->hidden_size, 3, 1, d->enc_conv2))) return fail(rc);
    if (cfg->codebook_dim > DAC_RVQ_MAXDIM) { d->err = "codebook_dim > 8"; return fail(ZN_ERR_UNSUPPORTED); }
    const int nq = cfg->n_codebooks;
    std::vector<const float*> ptrs(7 * nq);
    for (int i = 0; i < nq; ++i) {
      const std::string q = "quantizer.quantizers." + std::to_string(i) + ".";
      auto wi = t.find(q + "in_proj.weight"), bi2 = t.find(q + "in_proj.bias"), e = t.find(q + "codebook.weight"), wo = t.find(q + "out_proj.weight"),
           bo = t.find(q + "out_proj.bias");
      if (wi == t.end() || bi2 == t.end() || wi->second->numel != (int64_t)cfg->codebook_dim * cfg->hidden_size) { d->err = "missing/mis-shaped in_proj of codebook " + std::to_string(i); return fail(ZN_ERR_ARG); }
      float *cbn = nullptr, *cbsq = nullptr;
      if (hipMalloc(&cbn, (size_t)cfg->codebook_size * cfg->codebook_dim * sizeof(float)) != hipSuccess || hipMalloc(&cbsq, cfg->codebook_size * sizeof(float)) != hipSuccess) {
        d->err = "hipMalloc failed"; return fail(ZN_ERR_HIP);
      }
      d->owned.push_back(cbn); d->owned.push_back(cbsq);
      hipLaunchKernelGGL(dac_cbnorm_kernel, dim3((cfg->codebook_size + 255) / 256), dim3(256), 0, 0, e->second->data_dev, cbn, cbsq, cfg->codebook_size, cfg->codebook_dim);
      ptrs[0 * nq + i] = wi->second->data_dev; ptrs[1 * nq + i] = bi2->second->data_dev; ptrs[2 * nq + i] = e->second->data_dev;
      ptrs[3 * nq + i] = cbn; ptrs[4 * nq + i] = cbsq; ptrs[5 * nq + i] = wo->second->data_dev; ptrs[6 * nq + i] = bo->second->data_dev;
    }
    float* dp = nullptr;
    if (hipMalloc(&dp, ptrs.size() * sizeof(float*)) != hipSuccess || hipMemcpy(dp, ptrs.data(), ptrs.size() * sizeof(float*), hipMemcpyHostToDevice) != hipSuccess) {
      d->err = "hipMalloc/hipMemcpy failed"; return fail(ZN_ERR_HIP);
    }
    d->owned.push_back(dp);
    d->rvq_ptrs = (const float**)dp;
    d->has_encoder = true;
  }
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { d->err = "weight re-layout kernels failed"; return fail(ZN_ERR_HIP); }
  { hipError_t e = zn_conv3_set_attrs(); if (e != hipSuccess) { d->err = std::string("hipFuncSetAttribute: ") + hipGetErrorString(e); return fail(ZN_ERR_HIP); } }
  { hipError_t e = zn_conv_set_attrs(); if (e != hipSuccess) { d->err = std::string("hipFuncSetAttribute: ") + hipGetErrorString(e); return fail(ZN_ERR_HIP); } }
  (void)hipFuncSetAttribute((const void*)dac_final_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  *out = d;
  return ZN_OK;
}

static int launch_conv_args(const ConvArgs& a, int coutpad, int B, hipStream_t s) { zn_conv_launch(a, coutpad, B, s); return ZN_OK; }

static int launch_conv(zn_dac d, const ConvLayer& L, const float* in, int Tin, const float* skip, float* out, int Tout, int B, bool transpose, hipStream_t s) {
  ConvArgs a{};
  a.in = in; a.Tin = Tin; a.Cin = L.Cin; a.w = L.w; a.bias = L.bias; a.alpha = L.alpha; a.skip = skip; a.out = out; a.Tout = Tout; a.Cout = L.Cout; a.CoutPad = L.CoutPad;
  if (!transpose) {
    a.M = Tin; a.taps = L.K; a.off0 = -((L.K - 1) * L.dil) / 2; a.offstep = L.dil; a.ostride = 1; a.ooff = 0; a.phases = 1;
  } else {
    const int st = L.stride, padT = (st + 1) / 2;   // math.ceil(stride / 2), modeling_dac.py:250
    a.M = Tin + 1; a.taps = 2; a.off0 = 0; a.offstep = -1; a.ostride = st; a.ooff = -padT; a.phases = st;
  }
  return launch_conv_args(a, L.CoutPad, B, s);
}

// Three-term path: `in` is the layer's input as the producer left it (activated, fp32); alpha_next = the Snake the consumer of this layer's
// output applies (NULL: none / the consumer reads out32).
static void launch_conv3(const ConvLayer& L, const float* in, int Tin, const float* skip, float* out32, float* out_act, const float* alpha_next,
                         int Tout, int B, bool transpose, hipStream_t s) {
  Conv3Args a{};
  a.in = in; a.Tin = Tin; a.Cin = L.Cin; a.w = L.w3; a.bias = L.bias; a.alpha = alpha_next; a.skip = skip; a.out32 = out32; a.out_act = out_act;
  a.Tout = Tout; a.Cout = L.Cout; a.CoutPad = zn_conv3_pad(L.Cout);
  if (!transpose) {
    a.M = Tin; a.taps = L.K; a.off0 = -((L.K - 1) * L.dil) / 2; a.offstep = L.dil; a.ostride = 1; a.ooff = 0; a.phases = 1;
  } else {
    const int st = L.stride, padT = (st + 1) / 2;
    a.M = Tin + 1; a.taps = 2; a.off0 = 0; a.offstep = -1; a.ostride = st; a.ooff = -padT; a.phases = st;
  }
  zn_conv3_launch(a, B, s);
}

extern "C" int zn_dac_decode(zn_dac d, const int32_t* codes, int32_t B, int32_t T, float* wav, zn_stream stream) {
  if (!d) return ZN_ERR_ARG;
  if (!codes || !wav || B < 1 || T < 1) DFAIL(d, ZN_ERR_ARG, "zn_dac_decode: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const zn_dac_config& c = d->cfg;
  // largest activation: [B][T*prod(ratios[:k])][channels_k]
  size_t need = (size_t)B * T * (c.hidden_size > c.decoder_hidden_size ? c.hidden_size : c.decoder_hidden_size);
  { size_t tt = T; int ch = c.decoder_hidden_size;
    for (int i = 0; i < c.n_ratios; ++i) { tt *= c.ratios[i]; ch /= 2; need = std::max(need, (size_t)B * tt * ch); } }
  if (need > d->buf_elems) {
    DHIP(d, hipStreamSynchronize(s));
    for (auto& p : d->buf) { if (p) (void)hipFree(p); p = nullptr; }
    for (auto& p : d->buf) DHIP(d, hipMalloc(&p, need * sizeof(float)));
    d->buf_elems = need;
  }
  float *x = d->buf[0], *y = d->buf[1], *z = d->buf[2];
  hipLaunchKernelGGL(dac_codes_kernel, dim3(T, B), dim3(256), 0, s, codes, d->table, x, c.n_codebooks, T, c.hidden_size, c.codebook_size);
  // the three-term kernels address a batch element's activations with 32-bit byte offsets (buffer loads): 4 bytes per element, < 2 GiB per
  // layer and batch element (126 s of audio at the last block's 96 channels); longer clips take the fp32 kernels
  if (d->split3 && (need / (size_t)B) * 4 < 0x7fffffffull) {
    // x: the fp32 residual stream (a unit's conv2 adds to it in place); p / q: the activated input of the next convolution
    float *p = y, *q = z;
    launch_conv3(d->conv1, x, T, nullptr, nullptr, q, d->blocks[0].convt.alpha, T, B, false, s);     // decoder.conv1 on the latent -> block 0's snake1
    std::swap(p, q);
    int t = T;
    for (int bi = 0; bi < c.n_ratios; ++bi) {
      auto& Bk = d->blocks[bi];
      const int to = t * c.ratios[bi];
      launch_conv3(Bk.convt, p, t, nullptr, x, q, Bk.c1[0].alpha, to, B, true, s);                   // conv_t1 -> x, unit 1's snake1
      std::swap(p, q);
      t = to;
      for (int u = 0; u < 3; ++u) {                                                                    // x += conv2(snake2(conv1(snake1(x))))
        const bool last_unit = u == 2, last_block = bi + 1 == c.n_ratios;
        const float* an = !last_unit ? Bk.c1[u + 1].alpha : (!last_block ? d->blocks[bi + 1].convt.alpha : nullptr);
        launch_conv3(Bk.c1[u], p, t, nullptr, nullptr, q, Bk.c2[u].alpha, t, B, false, s);
        launch_conv3(Bk.c2[u], q, t, x, x, (last_unit && last_block) ? nullptr : p, an, t, B, false, s);
      }
    }
    const size_t lds = (size_t)((DAC_FIN_T + 6) * (d->fin_C + 1) + 9 * d->fin_C) * sizeof(float);
    hipLaunchKernelGGL(dac_final_kernel, dim3((t + DAC_FIN_T - 1) / DAC_FIN_T, B), dim3(256), lds, s, x, d->fin_alpha, d->fin_w, d->fin_b, wav, t, d->fin_C);
    DHIP(d, hipGetLastError());
    return ZN_OK;
  }
  launch_conv(d, d->conv1, x, T, nullptr, y, T, B, false, s);          // decoder.conv1
  std::swap(x, y);
  int t = T;
  for (int bi = 0; bi < c.n_ratios; ++bi) {
    auto& Bk = d->blocks[bi];
    const int to = t * c.ratios[bi];
    launch_conv(d, Bk.convt, x, t, nullptr, y, to, B, true, s);          // snake1 -> conv_t1
    std::swap(x, y);
    t = to;
    for (int u = 0; u < 3; ++u) {                                        // res units: x + conv2(snake2(conv1(snake1(x))))
      launch_conv(d, Bk.c1[u], x, t, nullptr, y, t, B, false, s);
      launch_conv(d, Bk.c2[u], y, t, x, z, t, B, false, s);
      std::swap(x, z);
    }
  }
  const size_t lds = (size_t)((DAC_FIN_T + 6) * (d->fin_C + 1) + 9 * d->fin_C) * sizeof(float);
  hipLaunchKernelGGL(dac_final_kernel, dim3((t + DAC_FIN_T - 1) / DAC_FIN_T, B), dim3(256), lds, s, x, d->fin_alpha, d->fin_w, d->fin_b, wav, t, d->fin_C);
  DHIP(d, hipGetLastError());
  return ZN_OK;
}

// DACAutoencoder.encode (zonos/autoencoder.py:103-117): wav fp32 [B][T] (T a multiple of the hop: preprocess pads) ->
// codes int32 [B][n_codebooks][T / hop].
extern "C" int zn_dac_encode(zn_dac d, const float* wav, int32_t B, int32_t T, int32_t* codes, zn_stream stream) {
  if (!d) return ZN_ERR_ARG;
  if (!d->has_encoder) DFAIL(d, ZN_ERR_STATE, "zn_dac_encode: the handle was created without encoder / in_proj tensors");
  const zn_dac_config& c = d->cfg;
  int hop = 1;
  for (int i = 0; i < c.n_ratios; ++i) hop *= c.ratios[i];
  if (!wav || !codes || B < 1 || T < hop || T % hop) DFAIL(d, ZN_ERR_ARG, "zn_dac_encode: T must be a positive multiple of %d", hop);
  hipStream_t s = (hipStream_t)stream;
  // largest activation: [B][T][encoder_hidden] (later stages halve T*C or keep it)
  size_t need = (size_t)B * T * c.encoder_hidden_size;
  { size_t tt = T; int ch = c.encoder_hidden_size;
    for (int i = 0; i < c.n_ratios; ++i) { tt /= d->eblocks[i].stride; ch *= 2; need = std::max(need, (size_t)B * tt * ch); }
    need = std::max(need, (size_t)B * tt * c.hidden_size); }
  if (need > d->buf_elems) {
    DHIP(d, hipStreamSynchronize(s));
    for (auto& p : d->buf) { if (p) (void)hipFree(p); p = nullptr; }
    for (auto& p : d->buf) DHIP(d, hipMalloc(&p, need * sizeof(float)));
    d->buf_elems = need;
  }
  float *x = d->buf[0], *y = d->buf[1], *z = d->buf[2];
  int ch = c.encoder_hidden_size, t = T;
  { const size_t n = (size_t)T * ch;
    hipLaunchKernelGGL(dac_enc_conv1_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, s, wav, d->enc_w1, d->enc_b1, x, T, ch); }
  for (int bi = 0; bi < c.n_ratios; ++bi) {
    auto& E = d->eblocks[bi];
    for (int u = 0; u < 3; ++u) {                                        // res units: x + conv2(snake2(conv1(snake1(x))))
      launch_conv(d, E.c1[u], x, t, nullptr, y, t, B, false, s);
      launch_conv(d, E.c2[u], y, t, x, z, t, B, false, s);
      std::swap(x, z);
    }
    // snake1 -> Conv1d(k = 2s, stride s, pad ceil(s/2)) as a 3-tap GEMM over rows of s time steps
    const int st = E.stride, to = t / st;
    ConvArgs a{};
    a.in = x; a.Tin = to; a.Cin = E.down.Cin; a.w = E.down.w; a.bias = E.down.bias; a.alpha = E.down.alpha; a.skip = nullptr;
    a.out = y; a.Tout = to; a.Cout = E.down.Cout; a.CoutPad = E.down.CoutPad;
    a.M = to; a.taps = 3; a.off0 = -1; a.offstep = 1; a.ostride = 1; a.ooff = 0; a.phases = 1;
    launch_conv_args(a, E.down.CoutPad, B, s);
    std::swap(x, y);
    t = to; ch *= 2;
  }
  launch_conv(d, d->enc_conv2, x, t, nullptr, y, t, B, false, s);       // snake1 -> conv2 (k = 3)
  RvqArgs r{};
  const int nq = c.n_codebooks;
  r.z = y; r.w_in = d->rvq_ptrs + 0 * nq; r.b_in = d->rvq_ptrs + 1 * nq; r.cb = d->rvq_ptrs + 2 * nq; r.cbn = d->rvq_ptrs + 3 * nq;
  r.cbsq = d->rvq_ptrs + 4 * nq; r.w_out = d->rvq_ptrs + 5 * nq; r.b_out = d->rvq_ptrs + 6 * nq;
  r.codes = codes; r.nq = nq; r.T = t; r.hidden = c.hidden_size; r.dim = c.codebook_dim; r.size = c.codebook_size;
  hipLaunchKernelGGL(dac_rvq_kernel, dim3(t, B), dim3(256), c.hidden_size * sizeof(float), s, r);
  DHIP(d, hipGetLastError());
  return ZN_OK;
}
