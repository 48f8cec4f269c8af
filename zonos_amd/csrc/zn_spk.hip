// Speaker embedding network (zonos/speaker_cloning.py:323-472 ResNet293_based = ResNet293 of SimAMBasicBlock :139-215,
// ASP :90-136, bottleneck Linear; :800-883 SpeakerEmbeddingLDA's LDA Linear) on log-mel features, fp32 like the reference.
// runs once per speaker, outside the decode loop.
//
// Feature maps are channels-last [H + 2][Wp][C]: H = mel bins with one zero row above and below (so that every row pass of
// a 3x3 conv covers every output row and the border reads are ordinary zero rows), Wp = frames rounded up to even with
// the pad column kept zero.
// A 3x3 Conv2d is three row-shifted passes of the shared implicit-GEMM 1-D kernel (zn_conv_kernels.h) over image rows
// (a "batch" of rows, batch strides = row pitch): the centre row pass writes bias + conv, the other two accumulate
// through the skip path.  Stride 2 reads every other row (in_bs = 2 rows) and, along the width, rows of 2 steps
// ([Wp/2][2*Cin], a 2-tap GEMM whose re-laid-out weight is zero where the 3-tap window does not reach).  BatchNorm (eval)
// is folded into each conv's weight and bias when the handle is built.  SimAM + residual + ReLU is two launches per block
// (per-channel sums in double over position slices, then the element-wise pass).
#include "../../include/zonos_hip.h"
#include "zn_conv_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

static thread_local std::string g_spk_err;

// ------------------------------------------------------------------------------------------------ weight preparation
struct BnRef { const float *g, *b, *m, *v; };
// 3x3 conv weight [Cout][Cin][3][3] (+ eval BatchNorm) -> 3 x [taps][Cin'][CoutPad] and bias[Cout].
// stride 1: taps = 3 (dw), Cin' = Cin.  stride 2: taps = 2 over rows of 2 steps, Cin' = 2*Cin: tap 0 (row wo-1) element 1
// <-> dw 0; tap 1 (row wo) elements 0, 1 <-> dw 1, 2.
__global__ void spk_fold3x3_kernel(const float* w, BnRef bn, float eps, float* o, float* bias, int Cout, int Cin, int stride, int CoutPad) {
  const int taps = stride == 1 ? 3 : 2, cinp = stride * Cin;
  const size_t n = (size_t)3 * taps * cinp * CoutPad;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int co = i % CoutPad, cc = (i / CoutPad) % cinp, tap = (i / ((size_t)CoutPad * cinp)) % taps, dh = i / ((size_t)CoutPad * cinp * taps);
    int ci = cc, dw = tap;
    if (stride == 2) { const int e = cc / Cin; ci = cc % Cin; dw = tap == 0 ? (e == 1 ? 0 : -1) : (e == 0 ? 1 : 2); }
    float v = 0.f;
    if (co < Cout && dw >= 0) v = w[(((size_t)co * Cin + ci) * 3 + dh) * 3 + dw] * (bn.g[co] / sqrtf(bn.v[co] + eps));
    o[i] = v;
  }
  for (int co = blockIdx.x * blockDim.x + threadIdx.x; co < Cout; co += gridDim.x * blockDim.x)
    bias[co] = bn.b[co] - bn.m[co] * (bn.g[co] / sqrtf(bn.v[co] + eps));
}
// 1x1 stride-2 downsample conv [Cout][Cin] (+ BatchNorm) -> [1 tap][2*Cin][CoutPad] (element 0 of the 2-step row)
__global__ void spk_fold1x1s2_kernel(const float* w, BnRef bn, float eps, float* o, float* bias, int Cout, int Cin, int CoutPad) {
  const size_t n = (size_t)2 * Cin * CoutPad;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int co = i % CoutPad, cc = i / CoutPad;
    o[i] = (co < Cout && cc < Cin) ? w[(size_t)co * Cin + cc] * (bn.g[co] / sqrtf(bn.v[co] + eps)) : 0.f;
  }
  for (int co = blockIdx.x * blockDim.x + threadIdx.x; co < Cout; co += gridDim.x * blockDim.x)
    bias[co] = bn.b[co] - bn.m[co] * (bn.g[co] / sqrtf(bn.v[co] + eps));
}
// ASP attention conv 1 [A][F] with F = c*H + h  ->  [1][F' = h*C + c][APad]
__global__ void spk_asp1_kernel(const float* w, float* o, int A, int C, int H, int APad) {
  const size_t n = (size_t)C * H * APad;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int a = i % APad, fp = i / APad, h = fp / C, c = fp % C;
    o[i] = a < A ? w[(size_t)a * C * H + c * H + h] : 0.f;
  }
}
// ASP attention conv 2 [F][A] after BatchNorm1d(A) on its input: W2' = W2 diag(g/sqrt(v+eps)), b2' = b2 + W2 (beta - mu g/sqrt(v+eps));
// -> [1][A][F' (padded)], rows reordered to F' = h*C + c
__global__ void spk_asp2_kernel(const float* w, const float* b, BnRef bn, float eps, float* o, float* bias, int A, int C, int H, int FPad) {
  const int F = C * H;
  const size_t n = (size_t)A * FPad;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int fp = i % FPad, a = i / FPad;
    float v = 0.f;
    if (fp < F) { const int h = fp / C, c = fp % C; v = w[(size_t)(c * H + h) * A + a] * (bn.g[a] / sqrtf(bn.v[a] + eps)); }
    o[i] = v;
  }
  for (int fp = blockIdx.x * blockDim.x + threadIdx.x; fp < F; fp += gridDim.x * blockDim.x) {
    const int h = fp / C, c = fp % C, f = c * H + h;
    float acc = b[f];
    for (int a = 0; a < A; ++a) acc += w[(size_t)f * A + a] * (bn.b[a] - bn.m[a] * (bn.g[a] / sqrtf(bn.v[a] + eps)));
    bias[fp] = acc;
  }
}

// ------------------------------------------------------------------------------------------------ forward kernels
// front.conv1 (1 -> C, 3x3, pad 1) + BatchNorm + ReLU on the feature map [H][W] -> [H][Wp][C]
__global__ __launch_bounds__(256) void spk_conv1_kernel(const float* feat, const float* w /*[C][1][3][3]*/, BnRef bn, float eps, float* out, int H, int W,
                                                        int Wp, int C) {
  const size_t idx = blockIdx.x * (size_t)256 + threadIdx.x;
  if (idx >= (size_t)H * W * C) return;
  const int co = idx % C, x = (idx / C) % W, y = idx / ((size_t)C * W);
  float acc = 0.f;
#pragma unroll
  for (int dh = 0; dh < 3; ++dh)
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) {
      const int yy = y + dh - 1, xx = x + dw - 1;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) acc = fmaf(w[(co * 3 + dh) * 3 + dw], feat[(size_t)yy * W + xx], acc);
    }
  const float sc = bn.g[co] / sqrtf(bn.v[co] + eps);
  out[((size_t)(y + 1) * Wp + x) * C + co] = fmaxf((acc - bn.m[co]) * sc + bn.b[co], 0.f);
}
// SimAM (speaker_cloning.py:192-215) + residual + ReLU: per channel mu = mean(X), d = (X - mu)^2, v = sum(d) / (H*W - 1),
// out = relu(X * sigmoid(d / (4 (v + 1e-4)) + 0.5) + res).  Two launches over the whole map: per-channel sums of x and
// x^2 in double (slices of positions per workgroup, fp64 atomics: sum(d) = sum x^2 - n mu^2 is exact to fp32 accuracy in
// double), then the element-wise pass.
__global__ __launch_bounds__(256) void spk_simam_stats_kernel(const float* x, double* stats /*[C][2], zeroed*/, int H, int W, int Wp, int C, int per) {
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), pl = threadIdx.x >> 5;
  __shared__ double red[2][8][32];
  const int npos = H * W, p0 = blockIdx.y * per, p1 = min(npos, p0 + per);
  double s = 0.0, q = 0.0;
  for (int p = p0 + pl; p < p1; p += 8) { const double v = (double)x[((size_t)(p / W + 1) * Wp + p % W) * C + c]; s += v; q += v * v; }
  red[0][pl][threadIdx.x & 31] = s; red[1][pl][threadIdx.x & 31] = q;
  __syncthreads();
  if (pl < 2) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += red[pl][j][threadIdx.x & 31];
    atomicAdd(stats + (size_t)c * 2 + pl, t);
  }
}
__global__ __launch_bounds__(256) void spk_simam_apply_kernel(const float* x, const float* res, const double* stats, float* out, int H, int W, int Wp, int C) {
  const size_t n = (size_t)H * W * C;
  const double npos = (double)H * W;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int c = i % C, w = (i / C) % W, h = i / ((size_t)C * W);
    const double sx = stats[(size_t)c * 2], sq = stats[(size_t)c * 2 + 1];
    const float mu = (float)(sx / npos);
    const float v = (float)((sq - sx * sx / npos) / (npos - 1.0));
    const size_t o = ((size_t)(h + 1) * Wp + w) * C + c;
    const float xv = x[o], dd = (xv - mu) * (xv - mu);
    const float e = dd / (4.0f * (v + 1e-4f)) + 0.5f;
    out[o] = fmaxf(xv * (1.0f / (1.0f + expf(-e))) + res[o], 0.f);
  }
}
// [H][Wp][C] -> [W][H*C] (time-major rows for the attention GEMMs)
__global__ void spk_transpose_kernel(const float* x, float* o, int H, int W, int Wp, int C) {
  const size_t n = (size_t)H * W * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = i % C, h = (i / C) % H, t = i / ((size_t)C * H);
    o[i] = x[((size_t)(h + 1) * Wp + t) * C + c];
  }
}
// ASP statistics (speaker_cloning.py:128-136): per feature f', w = softmax_t(logit), mu = sum_t x w, sg = sqrt(clamp(sum_t x^2 w
// - mu^2, 1e-5)); pooled[c*H + h] = mu, pooled[F + c*H + h] = sg (the reference's feature order).  One thread per feature.
__global__ void spk_asp_stats_kernel(const float* xt /*[T][F]*/, const float* logit /*[T][F]*/, float* pooled, int T, int C, int H) {
  const int F = C * H, fp = blockIdx.x * blockDim.x + threadIdx.x;
  if (fp >= F) return;
  float m = -INFINITY;
  for (int t = 0; t < T; ++t) m = fmaxf(m, logit[(size_t)t * F + fp]);
  float den = 0.f;
  for (int t = 0; t < T; ++t) den += expf(logit[(size_t)t * F + fp] - m);
  float mu = 0.f, sq = 0.f;
  for (int t = 0; t < T; ++t) {
    const float w = expf(logit[(size_t)t * F + fp] - m) / den, xv = xt[(size_t)t * F + fp];
    mu += xv * w; sq += xv * xv * w;
  }
  const int h = fp / C, c = fp % C;
  pooled[c * H + h] = mu;
  pooled[F + c * H + h] = sqrtf(fmaxf(sq - mu * mu, 1e-5f));
}
// y[j] = b[j] + sum_i W[j][i] x[i]   (one wave per output)
__global__ __launch_bounds__(64) void spk_linear_kernel(const float* x, const float* w, const float* b, float* y, int N, int K) {
  const int j = blockIdx.x, lane = threadIdx.x;
  float acc = 0.f;
  for (int i = lane; i < K; i += 64) acc = fmaf(w[(size_t)j * K + i], x[i], acc);
  acc = wave_sum(acc);
  if (lane == 0) y[j] = acc + b[j];
}

// ------------------------------------------------------------------------------------------------ host
struct SpkConv { float* w = nullptr; float* bias = nullptr; int Cin = 0, Cout = 0, CoutPad = 0, stride = 1; };
struct SpkBlock { SpkConv c1, c2, down; bool has_down = false; int stride = 1; };
struct zn_spk_s {
  int in_planes = 64, n_mels = 80, att_dim = 128, emb_dim = 256, lda_dim = 0;
  const float* conv1_w = nullptr; BnRef bn1{};
  std::vector<SpkBlock> blocks;
  float *asp_w1 = nullptr, *asp_w2 = nullptr, *asp_b2 = nullptr; const float* asp_b1 = nullptr;
  int FPad = 0;
  const float *bott_w = nullptr, *bott_b = nullptr, *lda_w = nullptr, *lda_b = nullptr;
  float* buf[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t buf_elems = 0;
  float *xt = nullptr, *att = nullptr, *logit = nullptr, *pooled = nullptr;
  double* simam_stats = nullptr;   // [max channels][2]
  size_t asp_T = 0;
  std::vector<float*> owned;
  std::string err;
};

#define SFAIL(d, code, ...) do { char _b[512]; snprintf(_b, sizeof _b, __VA_ARGS__); if (d) (d)->err = _b; else g_spk_err = _b; return (code); } while (0)
#define SHIP(d, call) do { hipError_t _e = (call); if (_e != hipSuccess) SFAIL(d, ZN_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(_e)); } while (0)

extern "C" const char* zn_spk_last_error(zn_spk d) { return d ? d->err.c_str() : g_spk_err.c_str(); }

extern "C" int zn_spk_destroy(zn_spk dd) {
  zn_spk d = dd;
  if (!d) return ZN_OK;
  for (float* p : d->owned) (void)hipFree(p);
  for (float* p : d->buf) if (p) (void)hipFree(p);
  for (float* p : {d->xt, d->att, d->logit, d->pooled}) if (p) (void)hipFree(p);
  if (d->simam_stats) (void)hipFree(d->simam_stats);
  delete d;
  return ZN_OK;
}

typedef std::map<std::string, const zn_dac_tensor*> TMap;
static const zn_dac_tensor* need(zn_spk d, const TMap& t, const std::string& k, int64_t numel) {
  auto it = t.find(k);
  if (it == t.end() || (numel > 0 && it->second->numel != numel)) { d->err = "missing/mis-shaped tensor " + k; return nullptr; }
  return it->second;
}
static bool get_bn(zn_spk d, const TMap& t, const std::string& p, int C, BnRef& bn) {
  auto g = need(d, t, p + "weight", C), b = need(d, t, p + "bias", C), m = need(d, t, p + "running_mean", C), v = need(d, t, p + "running_var", C);
  if (!g || !b || !m || !v) return false;
  bn.g = g->data_dev; bn.b = b->data_dev; bn.m = m->data_dev; bn.v = v->data_dev;
  return true;
}
static int make3x3(zn_spk d, const TMap& t, const std::string& wname, const std::string& bnp, int Cin, int Cout, int stride, SpkConv& L) {
  auto w = need(d, t, wname, (int64_t)Cout * Cin * 9);
  BnRef bn{};
  if (!w || !get_bn(d, t, bnp, Cout, bn)) return ZN_ERR_ARG;
  if ((stride * Cin) % DAC_KC) SFAIL(d, ZN_ERR_UNSUPPORTED, "channel count %d not a multiple of %d", stride * Cin, DAC_KC);
  L.Cin = stride * Cin; L.Cout = Cout; L.CoutPad = zn_conv_pad(Cout); L.stride = stride;
  const int taps = stride == 1 ? 3 : 2;
  SHIP(d, hipMalloc(&L.w, (size_t)3 * taps * L.Cin * L.CoutPad * sizeof(float)));
  d->owned.push_back(L.w);
  SHIP(d, hipMalloc(&L.bias, Cout * sizeof(float)));
  d->owned.push_back(L.bias);
  hipLaunchKernelGGL(spk_fold3x3_kernel, dim3(512), dim3(256), 0, 0, w->data_dev, bn, 1e-5f, L.w, L.bias, Cout, Cin, stride, L.CoutPad);
  return ZN_OK;
}

// Tensors by the reference's state-dict names (ResNet293_based: front.*, pooling.*, bottleneck.*; optional lda.weight /
// lda.bias = the LDA file's keys prefixed with "lda.").  fp32, device.
extern "C" int zn_spk_create(const zn_dac_tensor* tensors, int32_t n, zn_spk* out) {
  if (!tensors || !out || n < 1) SFAIL((zn_spk) nullptr, ZN_ERR_ARG, "zn_spk_create: null argument");
  *out = nullptr;
  TMap t;
  for (int i = 0; i < n; ++i) if (tensors[i].name && tensors[i].data_dev) t[tensors[i].name] = &tensors[i];
  zn_spk d = new zn_spk_s();
  auto fail = [&](int code) { g_spk_err = d->err; zn_spk_destroy(d); return code; };
  auto w1 = t.find("front.conv1.weight");
  if (w1 == t.end() || w1->second->numel % 9) { d->err = "missing front.conv1.weight"; return fail(ZN_ERR_ARG); }
  const int P = (int)(w1->second->numel / 9);
  d->in_planes = P; d->conv1_w = w1->second->data_dev;
  if (P % 64) { d->err = "in_planes must be a multiple of 64"; return fail(ZN_ERR_UNSUPPORTED); }
  if (!get_bn(d, t, "front.bn1.", P, d->bn1)) return fail(ZN_ERR_ARG);
  int in_ch = P, rc;
  for (int li = 1; li <= 4; ++li) {
    const int planes = P << (li - 1);
    for (int bi = 0;; ++bi) {
      const std::string p = "front.layer" + std::to_string(li) + "." + std::to_string(bi) + ".";
      if (!t.count(p + "conv1.weight")) break;
      const int stride = (bi == 0 && li > 1) ? 2 : 1;
      SpkBlock B;
      B.stride = stride;
      if ((rc = make3x3(d, t, p + "conv1.weight", p + "bn1.", in_ch, planes, stride, B.c1))) return fail(rc);
      if ((rc = make3x3(d, t, p + "conv2.weight", p + "bn2.", planes, planes, 1, B.c2))) return fail(rc);
      if (stride != 1 || in_ch != planes) {
        if (stride != 2) { d->err = "stride-1 downsample convs are not supported"; return fail(ZN_ERR_UNSUPPORTED); }
        auto dw = need(d, t, p + "downsample.0.weight", (int64_t)planes * in_ch);
        BnRef bn{};
        if (!dw || !get_bn(d, t, p + "downsample.1.", planes, bn)) return fail(ZN_ERR_ARG);
        SpkConv& D = B.down;
        D.Cin = 2 * in_ch; D.Cout = planes; D.CoutPad = zn_conv_pad(planes); D.stride = 2;
        if (hipMalloc(&D.w, (size_t)D.Cin * D.CoutPad * sizeof(float)) != hipSuccess || hipMalloc(&D.bias, planes * sizeof(float)) != hipSuccess) {
          d->err = "hipMalloc failed"; return fail(ZN_ERR_HIP);
        }
        d->owned.push_back(D.w); d->owned.push_back(D.bias);
        hipLaunchKernelGGL(spk_fold1x1s2_kernel, dim3(256), dim3(256), 0, 0, dw->data_dev, bn, 1e-5f, D.w, D.bias, planes, in_ch, D.CoutPad);
        B.has_down = true;
      }
      d->blocks.push_back(B);
      in_ch = planes;
    }
  }
  if (d->blocks.empty()) { d->err = "no front.layer*.* blocks found"; return fail(ZN_ERR_ARG); }
  // ASP + bottleneck (+ LDA)
  auto a1b = t.find("pooling.attention.0.bias");
  if (a1b == t.end()) { d->err = "missing pooling.attention.0.bias"; return fail(ZN_ERR_ARG); }
  d->att_dim = (int)a1b->second->numel;
  auto a1w = t.find("pooling.attention.0.weight");
  if (a1w == t.end() || a1w->second->numel % d->att_dim || d->att_dim % 64) { d->err = "missing/mis-shaped pooling.attention.0.weight"; return fail(ZN_ERR_ARG); }
  const int F = (int)(a1w->second->numel / d->att_dim), C = in_ch, H = F / C;
  if (H * C != F || F % DAC_KC) { d->err = "ASP feature size does not match the last layer"; return fail(ZN_ERR_ARG); }
  d->n_mels = H * 8;
  d->FPad = zn_conv_pad(F);
  BnRef abn{};
  auto a2w = need(d, t, "pooling.attention.3.weight", (int64_t)F * d->att_dim), a2b = need(d, t, "pooling.attention.3.bias", F);
  if (!a2w || !a2b || !get_bn(d, t, "pooling.attention.2.", d->att_dim, abn)) return fail(ZN_ERR_ARG);
  const int APad = zn_conv_pad(d->att_dim);
  if (hipMalloc(&d->asp_w1, (size_t)F * APad * sizeof(float)) != hipSuccess || hipMalloc(&d->asp_w2, (size_t)d->att_dim * d->FPad * sizeof(float)) != hipSuccess ||
      hipMalloc(&d->asp_b2, F * sizeof(float)) != hipSuccess) { d->err = "hipMalloc failed"; return fail(ZN_ERR_HIP); }
  d->owned.push_back(d->asp_w1); d->owned.push_back(d->asp_w2); d->owned.push_back(d->asp_b2);
  d->asp_b1 = a1b->second->data_dev;
  hipLaunchKernelGGL(spk_asp1_kernel, dim3(512), dim3(256), 0, 0, a1w->second->data_dev, d->asp_w1, d->att_dim, C, H, APad);
  hipLaunchKernelGGL(spk_asp2_kernel, dim3(512), dim3(256), 0, 0, a2w->data_dev, a2b->data_dev, abn, 1e-5f, d->asp_w2, d->asp_b2, d->att_dim, C, H, d->FPad);
  auto bb = t.find("bottleneck.bias");
  if (bb == t.end()) { d->err = "missing bottleneck.bias"; return fail(ZN_ERR_ARG); }
  d->emb_dim = (int)bb->second->numel;
  auto bw = need(d, t, "bottleneck.weight", (int64_t)d->emb_dim * 2 * F);
  if (!bw) return fail(ZN_ERR_ARG);
  d->bott_w = bw->data_dev; d->bott_b = bb->second->data_dev;
  if (t.count("lda.bias")) {
    d->lda_dim = (int)t["lda.bias"]->numel;
    auto lw = need(d, t, "lda.weight", (int64_t)d->lda_dim * d->emb_dim);
    if (!lw) return fail(ZN_ERR_ARG);
    d->lda_w = lw->data_dev; d->lda_b = t["lda.bias"]->data_dev;
  }
  if (hipMalloc(&d->pooled, (size_t)(2 * F + d->emb_dim) * sizeof(float)) != hipSuccess || hipMalloc(&d->simam_stats, (size_t)C * 2 * sizeof(double)) != hipSuccess) {
    d->err = "hipMalloc failed"; return fail(ZN_ERR_HIP);
  }
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { d->err = "weight preparation kernels failed"; return fail(ZN_ERR_HIP); }
  { hipError_t e = zn_conv_set_attrs(); if (e != hipSuccess) { d->err = std::string("hipFuncSetAttribute: ") + hipGetErrorString(e); return fail(ZN_ERR_HIP); } }
  *out = d;
  return ZN_OK;
}

// one 3x3 conv (three row passes, centre row first); in [H+2][Wp][C], out [Ho+2][Wpo][Cout] (data rows 1..H / 1..Ho)
static void conv3x3(const SpkConv& L, const float* in, int Wp, int C, float* out, int Ho, int Wo, int Wpo, bool relu, hipStream_t s) {
  const int st = L.stride;
  const int order[3] = {1, 0, 2};
  for (int pass = 0; pass < 3; ++pass) {
    const int dh = order[pass];
    ConvArgs a{};
    a.in = in + (size_t)dh * Wp * C;                // padded input row of output row ho: st*ho + dh
    a.Cin = L.Cin; a.Tin = Wp / st;                 // rows of `st` steps
    a.w = L.w + (size_t)dh * (st == 1 ? 3 : 2) * L.Cin * L.CoutPad;
    a.bias = pass == 0 ? L.bias : nullptr;
    a.out = out + (size_t)Wpo * L.Cout; a.Tout = Wo; a.Cout = L.Cout; a.CoutPad = L.CoutPad;
    a.skip = pass == 0 ? nullptr : a.out;
    a.M = Wo; a.taps = st == 1 ? 3 : 2; a.off0 = -1; a.offstep = 1; a.ostride = 1; a.ooff = 0; a.phases = 1;
    a.in_bs = (long long)st * Wp * C; a.out_bs = (long long)Wpo * L.Cout;
    a.relu = (relu && pass == 2) ? 1 : 0;
    zn_conv_launch(a, L.CoutPad, Ho, s);
  }
}

// SpeakerEmbedding.model minus the feature front end: feat fp32 [B][n_mels][T] (log-mel, mean-normalised, what
// logFbankCal returns, speaker_cloning.py:81-87) -> emb [B][emb_dim] (bottleneck output) and, when the handle has LDA
// weights and lda_out != NULL, lda_out [B][lda_dim] (SpeakerEmbeddingLDA.forward, :870-883).
extern "C" int zn_spk_embed(zn_spk dd, const float* feat, int32_t B, int32_t T, float* emb, float* lda_out, zn_stream stream) {
  zn_spk d = dd;
  if (!d) return ZN_ERR_ARG;
  if (!feat || !emb || B < 1 || T < 8) SFAIL(d, ZN_ERR_ARG, "zn_spk_embed: bad argument (T >= 8 frames)");
  if (lda_out && !d->lda_w) SFAIL(d, ZN_ERR_STATE, "zn_spk_embed: the handle has no LDA weights");
  hipStream_t s = (hipStream_t)stream;
  const int H0 = d->n_mels, P = d->in_planes;
  const int Wp0 = (T + 1) & ~1;
  const size_t need_elems = (size_t)(H0 + 2) * Wp0 * P;   // layer 1 is the largest map (later stages quarter H*W and double C)
  if (need_elems > d->buf_elems) {
    SHIP(d, hipStreamSynchronize(s));
    for (auto& p : d->buf) { if (p) (void)hipFree(p); p = nullptr; }
    for (auto& p : d->buf) SHIP(d, hipMalloc(&p, need_elems * sizeof(float)));
    d->buf_elems = need_elems;
  }
  // final map width after three stride-2 stages
  int Wl = T;
  for (int i = 0; i < 3; ++i) Wl = (Wl + 1) / 2;
  const int Cl = P * 8, Hl = H0 / 8, F = Cl * Hl;
  if ((size_t)Wl > d->asp_T) {
    SHIP(d, hipStreamSynchronize(s));
    for (float** p : {&d->xt, &d->att, &d->logit}) { if (*p) (void)hipFree(*p); *p = nullptr; }
    SHIP(d, hipMalloc(&d->xt, (size_t)Wl * F * sizeof(float)));
    SHIP(d, hipMalloc(&d->att, (size_t)Wl * d->att_dim * sizeof(float)));
    SHIP(d, hipMalloc(&d->logit, (size_t)Wl * F * sizeof(float)));
    d->asp_T = Wl;
  }
  for (int b = 0; b < B; ++b) {
    for (auto& p : d->buf) SHIP(d, hipMemsetAsync(p, 0, d->buf_elems * sizeof(float), s));   // pad columns must read as zero
    float *x = d->buf[0], *t1 = d->buf[1], *t2 = d->buf[2], *r = d->buf[3];
    int H = H0, W = T, Wp = Wp0, C = P;
    { const size_t n = (size_t)H * W * C;
      hipLaunchKernelGGL(spk_conv1_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, feat + (size_t)b * H0 * T, d->conv1_w, d->bn1, 1e-5f, x, H, W, Wp, C); }
    for (const SpkBlock& Bk : d->blocks) {
      const int st = Bk.stride;
      const int Ho = (H + st - 1) / st, Wo = (W + st - 1) / st, Wpo = (Wo + 1) & ~1, Co = Bk.c1.Cout;
      if (st == 2) {
        // smaller maps reuse the buffers with a new pitch: clear what the new layout treats as pad columns
        SHIP(d, hipMemsetAsync(t1, 0, (size_t)(Ho + 2) * Wpo * Co * sizeof(float), s));
        SHIP(d, hipMemsetAsync(t2, 0, (size_t)(Ho + 2) * Wpo * Co * sizeof(float), s));
        SHIP(d, hipMemsetAsync(r, 0, (size_t)(Ho + 2) * Wpo * Co * sizeof(float), s));
      }
      conv3x3(Bk.c1, x, Wp, C, t1, Ho, Wo, Wpo, true, s);                      // relu(bn1(conv1(x)))
      conv3x3(Bk.c2, t1, Wpo, Co, t2, Ho, Wo, Wpo, false, s);                  // bn2(conv2(.))
      const float* res = x;
      if (Bk.has_down) {                                                        // bn(conv1x1 stride 2 (x))
        ConvArgs a{};
        a.in = x + (size_t)Wp * C; a.Cin = Bk.down.Cin; a.Tin = Wp / 2; a.w = Bk.down.w; a.bias = Bk.down.bias;
        a.out = r + (size_t)Wpo * Co; a.Tout = Wo; a.Cout = Co; a.CoutPad = Bk.down.CoutPad;
        a.M = Wo; a.taps = 1; a.off0 = 0; a.offstep = 1; a.ostride = 1; a.ooff = 0; a.phases = 1;
        a.in_bs = (long long)2 * Wp * C; a.out_bs = (long long)Wpo * Co;
        zn_conv_launch(a, Bk.down.CoutPad, Ho, s);
        res = r;
      }
      // SimAM(out) + residual, ReLU -> new x (written over t1, which is free now)
      SHIP(d, hipMemsetAsync(d->simam_stats, 0, (size_t)Co * 2 * sizeof(double), s));
      { const int npos = Ho * Wo, per = 2048, ns = (npos + per - 1) / per;
        hipLaunchKernelGGL(spk_simam_stats_kernel, dim3(Co / 32, ns), dim3(256), 0, s, t2, d->simam_stats, Ho, Wo, Wpo, Co, per);
        const size_t n = (size_t)npos * Co;
        hipLaunchKernelGGL(spk_simam_apply_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, t2, res, d->simam_stats, t1, Ho, Wo, Wpo, Co); }
      if (st == 2) SHIP(d, hipMemsetAsync(x, 0, (size_t)(Ho + 2) * Wpo * Co * sizeof(float), s));   // x becomes scratch with the new layout
      std::swap(x, t1);
      H = Ho; W = Wo; Wp = Wpo; C = Co;
    }
    // ASP (speaker_cloning.py:128-136) on x [H][Wp][C] -> time-major [W][H*C]
    hipLaunchKernelGGL(spk_transpose_kernel, dim3(512), dim3(256), 0, s, x, d->xt, H, W, Wp, C);
    { ConvArgs a{};                                                            // relu(conv1d 1x1 F -> A)
      a.in = d->xt; a.Cin = F; a.Tin = W; a.w = d->asp_w1; a.bias = d->asp_b1; a.out = d->att; a.Tout = W; a.Cout = d->att_dim; a.CoutPad = zn_conv_pad(d->att_dim);
      a.M = W; a.taps = 1; a.off0 = 0; a.offstep = 1; a.ostride = 1; a.ooff = 0; a.phases = 1; a.relu = 1;
      zn_conv_launch(a, a.CoutPad, 1, s); }
    { ConvArgs a{};                                                            // BatchNorm1d folded, conv1d 1x1 A -> F
      a.in = d->att; a.Cin = d->att_dim; a.Tin = W; a.w = d->asp_w2; a.bias = d->asp_b2; a.out = d->logit; a.Tout = W; a.Cout = F; a.CoutPad = d->FPad;
      a.M = W; a.taps = 1; a.off0 = 0; a.offstep = 1; a.ostride = 1; a.ooff = 0; a.phases = 1;
      zn_conv_launch(a, d->FPad, 1, s); }
    hipLaunchKernelGGL(spk_asp_stats_kernel, dim3((F + 255) / 256), dim3(256), 0, s, d->xt, d->logit, d->pooled, W, C, H);
    hipLaunchKernelGGL(spk_linear_kernel, dim3(d->emb_dim), dim3(64), 0, s, d->pooled, d->bott_w, d->bott_b, emb + (size_t)b * d->emb_dim, d->emb_dim, 2 * F);
    if (lda_out)
      hipLaunchKernelGGL(spk_linear_kernel, dim3(d->lda_dim), dim3(64), 0, s, emb + (size_t)b * d->emb_dim, d->lda_w, d->lda_b, lda_out + (size_t)b * d->lda_dim,
                         d->lda_dim, d->emb_dim);
  }
  SHIP(d, hipGetLastError());
  return ZN_OK;
}
