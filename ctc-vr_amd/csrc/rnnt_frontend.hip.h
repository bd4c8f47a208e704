// Feature front-end helpers (rnnt_fbank): reflect padding and power spectrum around the two GEMMs.
// Part of rnnt_kernels.hip.h (include that umbrella, not this file).
#pragma once

// ------------------------------------------------------------------------------------------------
// Feature front-end (data/dataloader.py:15-41, torchaudio MelSpectrogram(center=True, pad_mode="reflect") + AmplitudeToDB):
// reflect_pad makes the n_fft/2-padded signal, the windowed DFT is a GEMM over implicit frames (row stride = hop) against
// interleaved (w cos, -w sin) rows, power_spectrum squares and adds the pairs, the mel projection is a second GEMM with
// the dB conversion as its epilogue.
// ------------------------------------------------------------------------------------------------
__global__ void reflect_pad(const float* __restrict__ x, float* __restrict__ y, int B, int n, int pad, long long ystride) {
    const long long total = (long long)B * ystride;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(id / ystride);
        const int i = (int)(id - (long long)b * ystride);
        float v = 0.f;
        if (i < n + 2 * pad) {
            int j = i - pad;
            if (j < 0) j = -j;                      // reflect without repeating the edge sample
            if (j >= n) j = 2 * (n - 1) - j;
            v = x[(long long)b * n + j];
        }
        y[id] = v;
    }
}
// spec [M][2*nfp] interleaved (re, im) -> pw [M][kp]: re^2 + im^2 for k < nfreq, 0 for the padding columns
__global__ void power_spectrum(const float* __restrict__ spec, float* __restrict__ pw, long long M, int nfreq, int kp, int ldspec) {
    const long long total = M * kp;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const long long m = id / kp;
        const int k = (int)(id - m * kp);
        float v = 0.f;
        if (k < nfreq) {
            const float2 c = *reinterpret_cast<const float2*>(spec + m * ldspec + 2 * k);
            v = c.x * c.x + c.y * c.y;
        }
        pw[id] = v;
    }
}
