// Prefix-conditioner numeric path (zonos/conditioning.py:338-522): embedding gathers, Fourier features, bias/SiLU.
// Once per utterance on a few dozen rows — latency only; kept on the device so the conditioning tensor never leaves HBM.
#pragma once
#include "zn_common.h"

// out[i][:] = table[ids[i] - id_offset][:]   (nn.Embedding rows, bf16; IntegerConditioner subtracts min_val)
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* table, const int* ids, bf16_t* out, int d, int n_rows_table, int id_offset) {
  const int i = blockIdx.x;
  int id = ids[i] - id_offset;
  id = id < 0 ? 0 : (id >= n_rows_table ? n_rows_table - 1 : id);
  for (int k = threadIdx.x * 8; k < d; k += 256 * 8) *(u32x4*)(out + (size_t)i * d + k) = ld16(table + (size_t)id * d + k);
}

// FourierConditioner.apply_cond (conditioning.py:436-441): x fp32 [n][in_dim]; xn = (x - min) / (max - min) in fp32;
// t = bf16(2*pi * bf16(xn)); f = bf16(sum_j t_j * W[c][j]) (fp32 accumulate); out = [bf16(cos f) | bf16(sin f)]
__global__ __launch_bounds__(256) void fourier_kernel(const float* x, const bf16_t* w, bf16_t* out, int in_dim, int half, float min_val, float max_val) {
  const int i = blockIdx.x;
  const float two_pi = 6.283185307179586f;
  for (int c = threadIdx.x; c < half; c += 256) {
    float acc = 0.f;
    for (int j = 0; j < in_dim; ++j) {
      const float xn = (x[(size_t)i * in_dim + j] - min_val) / (max_val - min_val);
      const float t = bfround(two_pi * bfround(xn));
      acc = fmaf(t, bf2f(w[(size_t)c * in_dim + j]), acc);
    }
    const float f = bfround(acc);
    out[(size_t)i * 2 * half + c] = f2bf(cosf(f));
    out[(size_t)i * 2 * half + half + c] = f2bf(sinf(f));
  }
}

// nn.SiLU on bf16 (fp32 math, one rounding)
__global__ __launch_bounds__(256) void silu_kernel(const bf16_t* x, bf16_t* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = bf2f(x[i]);
    out[i] = f2bf(v / (1.0f + expf(-v)));
  }
}
