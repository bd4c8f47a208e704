"""TEST INFRASTRUCTURE ONLY -- CPU restatement (NumPy float64) of the reference's feature front-end,
data/dataloader.py:15-41: torchaudio.transforms.MelSpectrogram(sample_rate, n_fft=1024, n_mels=80, hop_length=512,
window_fn=torch.hamming_window, power=2.0) followed by AmplitudeToDB().

PARITY UNPINNED: torchaudio is not installed in the build container and the reference holds no fixture for this
function (example1.pt stores features of unknown audio), so the formulas below are restated from torchaudio's
documented defaults -- Spectrogram(win_length=n_fft, center=True, pad_mode="reflect", normalized=False, onesided),
melscale_fbanks(n_freqs, f_min=0, f_max=sample_rate//2, n_mels, sample_rate, norm=None, mel_scale="htk"),
AmplitudeToDB(stype="power", top_db=None): 10*log10(clamp(x, 1e-10)) - 10*log10(max(1e-10, 1.0)) -- and checked only
for self-consistency (tests/test_fbank_cpu.py).  Only tests/ may import this module."""
import numpy as np


def mel_filterbank(sample_rate, n_fft, n_mels=80):
    n_freqs = n_fft // 2 + 1
    all_freqs = np.linspace(0.0, sample_rate // 2, n_freqs)
    m_max = 2595.0 * np.log10(1.0 + float(sample_rate // 2) / 700.0)
    f_pts = 700.0 * (10.0 ** (np.linspace(0.0, m_max, n_mels + 2) / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up))          # [n_freqs, n_mels]


def extract_audio_features(wave, sample_rate, n_fft=1024, hop=512, n_mels=80):
    """wave [n] -> [1 + n // hop, n_mels] (dB)."""
    x = np.asarray(wave, np.float64)
    xp = np.pad(x, n_fft // 2, mode="reflect")
    n_frames = 1 + x.shape[0] // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    win = 0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)      # torch.hamming_window(periodic=True)
    spec = np.fft.rfft(xp[idx] * win[None, :], axis=1)
    power = spec.real ** 2 + spec.imag ** 2
    mel = power @ mel_filterbank(sample_rate, n_fft, n_mels)
    return 10.0 * np.log10(np.maximum(mel, 1e-10))
