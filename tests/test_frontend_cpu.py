"""CPU property tests of the audio front-end restatements (torch ops only, no GPU): `sinc_resample` (torchaudio's published
sinc_interp_hann, the call at zonos/autoencoder.py:98 and speaker_cloning.py:753-782) and the mel filterbank of
`logFbankCal` (speaker_cloning.py:39-87).  torchaudio is not installed, so these are unpinned against it; the tests check
what the published algorithms guarantee."""
import math

import numpy as np
import torch

from zonos_amd.autoencoder import sinc_resample
from zonos_amd.speaker_cloning import melscale_fbanks


def test_resample_identity_and_length():
    x = torch.randn(2, 1000)
    assert sinc_resample(x, 44100, 44100) is x
    for sr_in, sr_out, n in ((16000, 44100, 1600), (48000, 16000, 4803), (22050, 44100, 777)):
        y = sinc_resample(torch.randn(1, n), sr_in, sr_out)
        assert y.shape == (1, math.ceil(sr_out * n / sr_in))


def test_resample_preserves_a_tone():
    sr_in, sr_out, f = 48000, 16000, 1000.0
    t = torch.arange(48000) / sr_in
    y = sinc_resample(torch.sin(2 * math.pi * f * t)[None], sr_in, sr_out)[0]
    spec = torch.fft.rfft(y * torch.hann_window(y.numel())).abs()
    peak_hz = float(spec.argmax()) * sr_out / y.numel()
    assert abs(peak_hz - f) < 2.0
    mid = y[2000:-2000]
    assert abs(float(mid.abs().max()) - 1.0) < 0.02          # pass-band gain ~ 1
    # a tone above the new Nyquist frequency is removed (anti-aliasing low-pass)
    z = sinc_resample(torch.sin(2 * math.pi * 11000.0 * t)[None], sr_in, sr_out)[0]
    assert float(z[2000:-2000].abs().max()) < 0.02


def test_mel_filterbank_shape_and_coverage():
    fb = melscale_fbanks(257, 0.0, 8000.0, 80, 16000)
    assert fb.shape == (257, 80) and float(fb.min()) >= 0.0
    assert bool((fb.sum(0) > 0).all())                       # every filter has support
    centres = fb.argmax(0).numpy()
    assert np.all(np.diff(centres) >= 0) and centres[0] < 5 and centres[-1] > 240   # ordered, spanning 0 .. Nyquist
    peaks = fb.max(0).values
    assert float(peaks.max()) <= 1.0 + 1e-6                  # triangular filters with unit peak (norm=None)

