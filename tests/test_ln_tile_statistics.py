"""The LayerNorm statistics that batches of 3..8 utterances hand from out_proj's epilogue to fc1 (zonos_amd/csrc/zn_decode_kernels.h:
gemm16k_kernel<EPI_RESID> -> GemvArgs::ln_part_out -> gemm16s_kernel<EPI_SILU, ., LNP>), restated in numpy fp32 with the kernels' order of
operations (CPU; the GPU test of the kernels themselves is tests/test_gpu_decode.py::test_fc1_layernorm_from_handed_over_statistics):

  producer, per row and 16-column tile: the two 8-column halves {s8, q8 = sum (v - s8/8)^2}, combined by Chan's pairwise update for equal
      counts: {s8a + s8b, q8a + q8b + 4 ((s8b - s8a) / 8)^2};
  consumer, per row: mean = (sum_t s_t) / K, then sum_t M2_t + 16 (s_t / 16 - mean)^2 - the row's centred second moment about `mean`,
      EXACTLY in real arithmetic (the cross terms vanish tile by tile) - and rstd = 1 / sqrt(that / K + eps).

What the test pins: (1) the identity, in float64; (2) in fp32 the statistics agree with the reference's nn.LayerNorm (torch, the op at
_torch.py:325) to a few ulps on rows of very different mean and scale - including rows whose mean dwarfs their spread, where a
sum-of-squares formulation (E[x^2] - mean^2) would lose every digit; (3) the normalised bf16 rows differ from torch's in at most a bf16
ulp, on a small fraction of the values."""
import numpy as np
import pytest
import torch

K, T = 2048, 128
F32 = np.float32


def tile_partials(x):
    """x: [rows, K] float32 (bf16-representable values) -> (s, q) [rows, T] float32, the producer's arithmetic"""
    v = x.reshape(x.shape[0], T, 2, 8).astype(F32)
    s8 = np.zeros(v.shape[:3], F32)
    for i in range(8):                                    # (the kernel adds pair sums across four lanes: another order of the same eight adds)
        s8 = s8 + v[..., i]
    m8 = s8 * F32(0.125)
    q8 = np.zeros_like(s8)
    for i in range(8):
        d = v[..., i] - m8[...]
        q8 = q8 + d * d
    sa, sb, qa, qb = s8[..., 0], s8[..., 1], q8[..., 0], q8[..., 1]
    dm = (sb - sa) * F32(0.125)
    return sa + sb, qa + qb + F32(4.0) * dm * dm


def row_statistics(s, q, eps):
    """the consumer's arithmetic: 8 segments of 16 tiles, segments added in order (NT = 128 threads: LSEG = 8, LTS = 16)"""
    rows = s.shape[0]
    seg = np.zeros((rows, 8), F32)
    for t in range(16):
        seg = seg + s.reshape(rows, 8, 16)[:, :, t]
    tot = np.zeros(rows, F32)
    for g in range(8):
        tot = tot + seg[:, g]
    mean = tot * F32(1.0 / K)
    segq = np.zeros((rows, 8), F32)
    for t in range(16):
        dm = s.reshape(rows, 8, 16)[:, :, t] * F32(0.0625) - mean[:, None]
        segq = segq + (q.reshape(rows, 8, 16)[:, :, t] + F32(16.0) * dm * dm)
    tq = np.zeros(rows, F32)
    for g in range(8):
        tq = tq + segq[:, g]
    rstd = F32(1.0) / np.sqrt(tq * F32(1.0 / K) + F32(eps), dtype=F32)
    return mean, rstd, tq


def rows_for_test(seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(16, K, generator=g)
    scale = torch.tensor([1e-3, 0.05, 0.5, 1.0, 1.0, 2.0, 3.0, 8.0, 30.0, 1.0, 1.0, 1.0, 0.2, 0.2, 5.0, 100.0])[:, None]
    shift = torch.tensor([0.0, 0.0, 0.0, 0.0, 1.0, -2.0, 5.0, 0.0, 10.0, 50.0, -200.0, 1000.0, 3.0, -30.0, 0.3, -7.0])[:, None]
    x = x * scale + shift
    x[3, 100] = 80.0                                      # an outlier channel, as residual streams have
    return x.to(torch.bfloat16)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_tile_statistics_identity_and_fp32_accuracy(seed):
    xb = rows_for_test(seed)
    x = xb.float().numpy()
    s, q = tile_partials(x)
    mean, rstd, tq = row_statistics(s, q, 1e-5)
    x64 = x.astype(np.float64)
    mean64 = x64.mean(1)
    m2_64 = ((x64 - mean64[:, None]) ** 2).sum(1)
    # (1) the identity, evaluated in float64 from float64 partials
    v = x64.reshape(16, T, 16)
    st = v.sum(2)
    mt = st / 16
    qt = ((v - mt[..., None]) ** 2).sum(2)
    assert np.allclose((qt + 16 * (mt - mean64[:, None]) ** 2).sum(1), m2_64, rtol=1e-12, atol=0)
    # (2) fp32: mean within a few ulps of the row's scale, second moment to ~1e-6 relative even where |mean| >> spread
    scale = np.abs(x64).max(1)
    assert (np.abs(mean - mean64) <= 4 * np.finfo(F32).eps * scale).all(), np.abs(mean - mean64) / scale
    rel = np.abs(tq - m2_64) / m2_64
    naive = np.abs((x.astype(F32) ** 2).sum(1, dtype=F32) - F32(K) * mean * mean - m2_64) / m2_64       # E[x^2] - mean^2 in fp32, for contrast
    assert rel.max() < 2e-5, rel                          # (rows with |mean| ~ 1000 x their spread: the bf16 inputs themselves carry the spread in 2-3 bits)
    assert naive.max() > 1e-2                             # the formulation this one avoids
    rstd64 = 1.0 / np.sqrt(m2_64 / K + 1e-5)
    assert (np.abs(rstd - rstd64) / rstd64 < 2e-5).all()


@pytest.mark.parametrize("seed", [0, 3])
def test_rows_normalised_from_tile_statistics_match_torch_layer_norm(seed):
    xb = rows_for_test(seed)
    g = torch.Generator().manual_seed(100 + seed)
    w = (1.0 + 0.1 * torch.randn(K, generator=g)).to(torch.bfloat16)
    b = (0.1 * torch.randn(K, generator=g)).to(torch.bfloat16)
    ref = torch.nn.functional.layer_norm(xb, (K,), w, b, 1e-5)                                          # the reference's op, bf16 in / out, fp32 inside
    x = xb.float().numpy()
    s, q = tile_partials(x)
    mean, rstd, _ = row_statistics(s, q, 1e-5)
    y = ((x - mean[:, None]) * rstd[:, None]).astype(F32) * w.float().numpy() + b.float().numpy()      # ln_one: fma(mul(sub(x, m), q), g, b)
    got = torch.from_numpy(y.astype(F32)).to(torch.bfloat16)
    same = (got.view(torch.int16) == ref.view(torch.int16)).float().mean().item()
    gf, rf = got.float(), ref.float()
    worst = ((gf - rf).abs() / torch.maximum(rf.abs(), torch.tensor(2.0 ** -6))).max().item()           # in units of the value: a bf16 ulp is 2^-8 .. 2^-7 of it
    assert same > 0.98 and worst <= 2.0 ** -7, (same, worst)
