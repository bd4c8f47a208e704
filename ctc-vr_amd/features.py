"""Device feature front-end with the semantics of the reference's extract_audio_features (data/dataloader.py:15-41):
80-bin HTK mel power spectrogram (n_fft 1024, hop 512, periodic Hamming window, centred reflect padding) in dB.
The arithmetic runs in librnnt_hip.so (`rnnt_fbank`: windowed DFT and mel projection as exact-f32 MFMA GEMMs).
Reading audio files is host code: `load_wav` covers PCM WAV through the standard library (torchaudio is not a
dependency)."""
import wave as _wave

import numpy as np
import torch


def load_wav(path):
    """-> (waveform float32 [channels, n] in [-1, 1), sample_rate); PCM 8/16/32-bit WAV."""
    with _wave.open(path, "rb") as f:
        n, ch, width, rate = f.getnframes(), f.getnchannels(), f.getsampwidth(), f.getframerate()
        raw = f.readframes(n)
    if width == 2:
        a = np.frombuffer(raw, "<i2").astype(np.float32) / 32768.0
    elif width == 4:
        a = np.frombuffer(raw, "<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        a = (np.frombuffer(raw, np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"unsupported sample width {width}")
    return torch.from_numpy(a.reshape(-1, ch).T.copy()), rate


def extract_audio_features(engine, waveform, sample_rate, n_fft=1024, stream=None):
    """waveform: [n] or [B, n] float tensor (moved to the engine's GPU if needed) -> [n_frames, 80] or [B, n_frames, 80]
    float32 CUDA tensor, n_frames = 1 + n // 512."""
    single = waveform.dim() == 1
    w = waveform.reshape(1, -1) if single else waveform
    w = w.to(device=f"cuda:{engine.cfg.device}", dtype=torch.float32).contiguous()
    B, n = w.shape
    out = torch.empty((B, 1 + n // 512, 80), dtype=torch.float32, device=w.device)
    if stream is None:
        stream = torch.cuda.current_stream(w.device).cuda_stream
    t = engine.fbank(w.data_ptr(), B, n, int(sample_rate), out.data_ptr(), n_fft=n_fft, stream=stream)
    assert t == out.shape[1]
    return out[0] if single else out
