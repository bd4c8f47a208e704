"""Self-consistency of the front-end oracle (parity unpinned: no torchaudio, no reference fixture -- see its header)."""
import numpy as np

from oracle import fbank_oracle as F


def test_shapes_floor_and_tone_peak():
    rate, n = 16000, 16000
    t = np.arange(n) / rate
    fb = F.mel_filterbank(rate, 1024)
    assert fb.shape == (513, 80) and fb.min() >= 0.0 and (fb.sum(0) > 0).all()
    assert np.allclose(F.extract_audio_features(np.zeros(n), rate), -100.0)                  # clamp at 1e-10
    feats = F.extract_audio_features(np.sin(2 * np.pi * 1000.0 * t), rate)
    assert feats.shape == (1 + n // 512, 80)
    centre = np.argmax(fb[int(round(1000.0 / (rate / 2) * 512))])                             # mel bin that owns 1 kHz
    assert abs(int(np.argmax(feats[10])) - int(centre)) <= 1
    # a window-energy check on a full-scale tone: |X(f0)|^2 ~ (sum(win)/2)^2 within the owning triangle's weight
    win = 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(1024) / 1024)
    assert abs(feats[10].max() - 10 * np.log10((win.sum() / 2) ** 2)) < 3.0


def test_linearity_in_db():
    rng = np.random.default_rng(0)
    x = rng.standard_normal(8000)
    a, b = F.extract_audio_features(x, 16000), F.extract_audio_features(10.0 * x, 16000)
    assert np.allclose(b - a, 20.0, atol=1e-9)                                                 # power scales with amplitude^2
