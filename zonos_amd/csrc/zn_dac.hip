// DAC decode (zonos/autoencoder.py:119-170 -> transformers modeling_dac.py:347-371 from_codes, :407-441 DacDecoder).
//
// fp32 end to end (the reference's CPU path disables autocast; waveform RMS error budget 1e-4, so no bf16).  Every
// convolution is an implicit GEMM on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32): activations are kept
// channels-last [B][T][C] so that the reduction index (input channel) is contiguous, a workgroup owns a 128-time x
// (NT*32)-channel output tile, the input rows it needs (with the dilation halo) are staged once per 16-channel chunk in
// LDS with the Snake activation applied on the way in, and all taps of the chunk's weights sit next to them.
// ConvTranspose1d(k = 2s, stride s) is s independent 2-tap GEMMs (one per output phase).
#include "../../include/zonos_hip.h"
#include "zn_common.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;

static thread_local std::string g_dac_err;

#define DAC_KC 16
#define DAC_TM 128
#define DAC_MAXTAPS 7
#define DAC_MAXROWS (DAC_TM + 6 * 9)

struct ConvArgs {
  const float* in; int Tin, Cin;          // [B][Tin][Cin]
  const float* w;                          // [phase][tap][Cin][CoutPad]
  const float* bias;                       // [Cout]
  const float* alpha;                      // Snake alpha of the input channels, or NULL
  const float* skip;                       // residual [B][Tout][Cout], or NULL
  float* out; int Tout, Cout, CoutPad;     // [B][Tout][Cout]
  int M;                                   // GEMM rows per phase
  int taps, off0, offstep;                 // input row of GEMM row m, tap k: m + off0 + k*offstep
  int ostride, ooff, phases;               // output time of row m in phase p: m*ostride + ooff + p
};

__device__ __forceinline__ float snake_f(float x, float alpha) {
  // modeling_dac.py:98: x + (alpha + 1e-9)^-1 * sin(alpha x)^2   (accurate sinf, IEEE reciprocal)
  const float s = sinf(alpha * x);
  return x + (1.0f / (alpha + 1e-9f)) * (s * s);
}

template <int NT>
__global__ __launch_bounds__(256) void dac_conv_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int TN = NT * 32;
  float* s_in = smem;                                   // [rows][KC+1]
  float* s_w = smem + DAC_MAXROWS * (DAC_KC + 1);       // [taps*KC][TN]
  const int m0 = blockIdx.x * DAC_TM, n0 = blockIdx.y * TN;
  const int b = blockIdx.z / a.phases, phase = blockIdx.z % a.phases;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int offlast = a.off0 + (a.taps - 1) * a.offstep;
  const int offmin = a.off0 < offlast ? a.off0 : offlast, offmax = a.off0 < offlast ? offlast : a.off0;
  const int nrows = DAC_TM + offmax - offmin;
  const float* inb = a.in + (size_t)b * a.Tin * a.Cin;
  const float* wp = a.w + (size_t)phase * a.taps * a.Cin * a.CoutPad;
  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

  for (int c0 = 0; c0 < a.Cin; c0 += DAC_KC) {
    __syncthreads();
    // stage input rows [m0+offmin, m0+offmin+nrows) x KC channels, Snake on the way in (zero outside [0,Tin))
    for (int i = tid; i < nrows * (DAC_KC / 4); i += 256) {
      const int row = i / (DAC_KC / 4), c4 = (i % (DAC_KC / 4)) * 4;
      const int t = m0 + offmin + row;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (t >= 0 && t < a.Tin) {
        v = *(const f32x4*)(inb + (size_t)t * a.Cin + c0 + c4);
        if (a.alpha) {
          const f32x4 al = *(const f32x4*)(a.alpha + c0 + c4);
          v.x = snake_f(v.x, al.x); v.y = snake_f(v.y, al.y); v.z = snake_f(v.z, al.z); v.w = snake_f(v.w, al.w);
        }
      }
      float* d = s_in + row * (DAC_KC + 1) + c4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    // stage weights: taps x KC rows of TN output channels
    for (int i = tid; i < a.taps * DAC_KC * (TN / 4); i += 256) {
      const int rowi = i / (TN / 4), c4 = (i % (TN / 4)) * 4;
      const int tap = rowi / DAC_KC, ci = rowi % DAC_KC;
      *(f32x4*)(s_w + (size_t)rowi * TN + c4) = *(const f32x4*)(wp + ((size_t)tap * a.Cin + c0 + ci) * a.CoutPad + n0 + c4);
    }
    __syncthreads();
    const int ai = lane & 31, ak = lane >> 5;
    for (int tap = 0; tap < a.taps; ++tap) {
      const float* arow = s_in + (wave * 32 + ai + a.off0 + tap * a.offstep - offmin) * (DAC_KC + 1) + ak;
      const float* brow = s_w + (size_t)(tap * DAC_KC + ak) * TN + ai;
#pragma unroll
      for (int k = 0; k < DAC_KC; k += 2) {
        const float av = arow[k];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, brow[(size_t)k * TN + nt * 32], acc[nt], 0, 0, 0);
      }
    }
  }
  // epilogue: bias, residual, store.  C layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int col = lane & 31;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = n0 + nt * 32 + col;
    if (co >= a.Cout) continue;
    const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      const int m = m0 + wave * 32 + row;
      if (m >= a.M) continue;
      const int to = m * a.ostride + a.ooff + phase;
      if (to < 0 || to >= a.Tout) continue;
      const size_t o = ((size_t)b * a.Tout + to) * a.Cout + co;
      float v = acc[nt][reg] + bv;
      if (a.skip) v = a.skip[o] + v;
      a.out[o] = v;
    }
  }
}

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
// last layer: Snake -> Conv1d(C -> 1, k = 7, pad 3) -> tanh  (modeling_dac.py:438-441)
#define DAC_FIN_T 128
__global__ __launch_bounds__(DAC_FIN_T) void dac_final_kernel(const float* in, const float* alpha, const float* w /*[1][C][7]*/, const float* bias,
                                                             float* out, int T, int C) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_x = smem;                         // [DAC_FIN_T + 6][C + 1]
  float* s_w = smem + (DAC_FIN_T + 6) * (C + 1);   // [7][C]
  const int t0 = blockIdx.x * DAC_FIN_T, b = blockIdx.y;
  const float* inb = in + (size_t)b * T * C;
  for (int i = threadIdx.x; i < (DAC_FIN_T + 6) * C; i += DAC_FIN_T) {
    const int row = i / C, c = i % C, t = t0 - 3 + row;
    s_x[row * (C + 1) + c] = (t >= 0 && t < T) ? snake_f(inb[(size_t)t * C + c], alpha[c]) : 0.f;
  }
  for (int i = threadIdx.x; i < 7 * C; i += DAC_FIN_T) { const int k = i / C, c = i % C; s_w[i] = w[(size_t)c * 7 + k]; }
  __syncthreads();
  const int t = t0 + threadIdx.x;
  if (t >= T) return;
  float acc = 0.f;
  for (int k = 0; k < 7; ++k) {
    const float* xr = s_x + (threadIdx.x + k) * (C + 1);
    const float* wr = s_w + k * C;
    for (int c = 0; c < C; ++c) acc = fmaf(xr[c], wr[c], acc);
  }
  out[(size_t)b * T + t] = tanhf(acc + bias[0]);
}

// ------------------------------------------------------------------------------------------------ host
struct ConvLayer { float *w = nullptr; const float *bias = nullptr, *alpha = nullptr; int Cin = 0, Cout = 0, CoutPad = 0, K = 0, dil = 1, stride = 0; };
struct zn_dac_s {
  zn_dac_config cfg;
  float* table = nullptr;
  ConvLayer conv1;
  struct Block { ConvLayer convt; ConvLayer c1[3], c2[3]; } blocks[8];
  const float *fin_alpha = nullptr, *fin_w = nullptr, *fin_b = nullptr;
  int fin_C = 0;
  float* buf[3] = {nullptr, nullptr, nullptr};
  size_t buf_elems = 0;
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

// output channels are tiled by 128 (NT = 4) or, when that divides evenly, by 96 (NT = 3: the 192- and 96-channel stages)
static int pad32(int c) { return c % 128 == 0 ? c : (c % 96 == 0 ? c : (c + 127) / 128 * 128); }

static int make_conv(zn_dac d, const std::map<std::string, const zn_dac_tensor*>& t, const std::string& wname, const std::string& bname,
                     const char* aname, int Cin, int Cout, int K, int dil, ConvLayer& L) {
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
  if ((rc = make_conv(d, t, "decoder.conv1.weight", "decoder.conv1.bias", nullptr, cfg->hidden_size, c, 7, 1, d->conv1))) return fail(rc);
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
    const int dils[3] = {1, 3, 9};
    for (int u = 0; u < 3; ++u) {
      const std::string r = p + "res_unit" + std::to_string(u + 1) + ".";
      const std::string a1 = r + "snake1.alpha", a2 = r + "snake2.alpha";
      if ((rc = make_conv(d, t, r + "conv1.weight", r + "conv1.bias", a1.c_str(), co, co, 7, dils[u], B.c1[u]))) return fail(rc);
      if ((rc = make_conv(d, t, r + "conv2.weight", r + "conv2.bias", a2.c_str(), co, co, 1, 1, B.c2[u]))) return fail(rc);
    }
    c = co;
  }
  auto fa = t.find("decoder.snake1.alpha"), fw = t.find("decoder.conv2.weight"), fb = t.find("decoder.conv2.bias");
  if (fa == t.end() || fw == t.end() || fb == t.end() || fw->second->numel != (int64_t)c * 7) { d->err = "missing/mis-shaped final conv tensors"; return fail(ZN_ERR_ARG); }
  d->fin_alpha = fa->second->data_dev; d->fin_w = fw->second->data_dev; d->fin_b = fb->second->data_dev; d->fin_C = c;
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { d->err = "weight re-layout kernels failed"; return fail(ZN_ERR_HIP); }
  for (int nt : {3, 4}) {
    const int bytes = (DAC_MAXROWS * (DAC_KC + 1) + DAC_MAXTAPS * DAC_KC * nt * 32) * (int)sizeof(float);
    hipError_t e = nt == 3 ? hipFuncSetAttribute((const void*)dac_conv_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)
                           : hipFuncSetAttribute((const void*)dac_conv_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { d->err = std::string("hipFuncSetAttribute: ") + hipGetErrorString(e); return fail(ZN_ERR_HIP); }
  }
  (void)hipFuncSetAttribute((const void*)dac_final_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  *out = d;
  return ZN_OK;
}

static int launch_conv(zn_dac d, const ConvLayer& L, const float* in, int Tin, const float* skip, float* out, int Tout, int B, bool transpose, hipStream_t s) {
  ConvArgs a{};
  a.in = in; a.Tin = Tin; a.Cin = L.Cin; a.w = L.w; a.bias = L.bias; a.alpha = L.alpha; a.skip = skip; a.out = out; a.Tout = Tout; a.Cout = L.Cout; a.CoutPad = L.CoutPad;
  if (!transpose) {
    a.M = Tin; a.taps = L.K; a.off0 = -((L.K - 1) * L.dil) / 2; a.offstep = L.dil; a.ostride = 1; a.ooff = 0; a.phases = 1;
  } else {
    const int st = L.stride, padT = (st + 1) / 2;   // math.ceil(stride / 2), modeling_dac.py:250
    a.M = Tin + 1; a.taps = 2; a.off0 = 0; a.offstep = -1; a.ostride = st; a.ooff = -padT; a.phases = st;
  }
  const int nt = (L.CoutPad % 128 == 0) ? 4 : 3;
  const int TN = nt * 32;
  dim3 grid((a.M + DAC_TM - 1) / DAC_TM, L.CoutPad / TN, B * a.phases);
  const size_t lds = (size_t)(DAC_MAXROWS * (DAC_KC + 1) + DAC_MAXTAPS * DAC_KC * TN) * sizeof(float);
  if (nt == 4) hipLaunchKernelGGL((dac_conv_kernel<4>), grid, dim3(256), lds, s, a);
  else hipLaunchKernelGGL((dac_conv_kernel<3>), grid, dim3(256), lds, s, a);
  return ZN_OK;
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
  const size_t lds = (size_t)((DAC_FIN_T + 6) * (d->fin_C + 1) + 7 * d->fin_C) * sizeof(float);
  hipLaunchKernelGGL(dac_final_kernel, dim3((t + DAC_FIN_T - 1) / DAC_FIN_T, B), dim3(DAC_FIN_T), lds, s, x, d->fin_alpha, d->fin_w, d->fin_b, wav, t, d->fin_C);
  DHIP(d, hipGetLastError());
  return ZN_OK;
}
