// Small utilities: fills and the gathers behind the state read-back entry points.
// Part of rnnt_kernels.hip.h (include that umbrella, not this file).
#pragma once

// small helpers ------------------------------------------------------------------------------------
__global__ void fill_f32(float* p, float v, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill_i32(int* p, int v, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}
// rnnt_streams_reset: the decode state of every stream in ONE launch (was eight memsets and a fill, each a ~5 us stream operation):
// h, c [2][B][256] = 0, sel / fidx / nsym / count [B] = 0, key [B] = 0, tok [B] = blank, n_active [4] = 0
__global__ void decode_state_reset(float* h, float* c, int* sel, unsigned long long* key, int* fidx, int* nsym, int* count, int* n_active,
                                   int* tok, int blank, int B) {
    const long long n = (long long)2 * B * RNNT_D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        h[i] = 0.f; c[i] = 0.f;
        if (i < B) { sel[i] = 0; key[i] = 0ull; fidx[i] = 0; nsym[i] = 0; count[i] = 0; tok[i] = blank; }
        if (i < 4) n_active[i] = 0;
    }
}
// gather the reference's att_cache layout [L][H][len][128] (K|V) of one stream.
__global__ void gather_att_cache(const float* __restrict__ kc, const float* __restrict__ vc, float* __restrict__ dst, int b, int B,
                                 long long kv_stride, int kv_start, int len) {
    const long long n = (long long)RNNT_L * RNNT_H * len * 128;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(id & 127);
        long long r = id >> 7;
        const int j = (int)(r % len);
        r /= len;
        const int h = (int)(r % RNNT_H);
        const int l = (int)(r / RNNT_H);
        const float* src = (e < 64 ? kc : vc) + (((long long)l * B + b) * kv_stride + kv_start + j) * RNNT_D + h * RNNT_DK + (e & 63);
        dst[id] = *src;
    }
}
// reference cnn_cache layout [L][1][256][30] of one stream = LayerNorm(norm_conv) of the last 30
// conv-module input rows (zeros before stream start).  One wave per (l, frame).
__global__ void gather_cnn_cache(const float* __restrict__ xring, const float* __restrict__ lng /*[L][256]*/,
                                 const float* __restrict__ lnb, float* __restrict__ dst, int b, int B, int cap, int pos) {
    const int l = blockIdx.x / RNNT_LORDER, i = blockIdx.x % RNNT_LORDER;
    const int lane = threadIdx.x;
    const int frame = pos - RNNT_LORDER + i;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (frame >= 0) {
        const float* xp = xring + (((long long)l * B + b) * cap + frame % cap) * RNNT_D;
        const float4 v = *reinterpret_cast<const float4*>(xp + lane * 4);
        const float mu = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / 256.0f);
        const float dx = v.x - mu, dy = v.y - mu, dz = v.z - mu, dw = v.w - mu;
        const float rstd = 1.0f / sqrtf(wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.0f / 256.0f) + 1e-5f);
        const float4 gg = *reinterpret_cast<const float4*>(lng + l * RNNT_D + lane * 4);
        const float4 bb = *reinterpret_cast<const float4*>(lnb + l * RNNT_D + lane * 4);
        o.x = dx * rstd * gg.x + bb.x;
        o.y = dy * rstd * gg.y + bb.y;
        o.z = dz * rstd * gg.z + bb.z;
        o.w = dw * rstd * gg.w + bb.w;
    }
    float* d = dst + (long long)l * RNNT_D * RNNT_LORDER;
    d[(lane * 4 + 0) * RNNT_LORDER + i] = o.x;
    d[(lane * 4 + 1) * RNNT_LORDER + i] = o.y;
    d[(lane * 4 + 2) * RNNT_LORDER + i] = o.z;
    d[(lane * 4 + 3) * RNNT_LORDER + i] = o.w;
}

// Diagnostic (tools/decoder_contention.py): a plain streaming copy with selectable cache policy, to see what a streaming
// neighbour does to the resident decoder.  mode 0: default loads/stores, 1: non-temporal loads, 2: non-temporal loads and stores.
__global__ void debug_stream_copy(const float* __restrict__ src, float* __restrict__ dst, long long n4, int mode) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = mode >= 1 ? ldg4_nt(src + 4 * i) : ldg4(src + 4 * i);
        if (mode >= 2) stg4_nt(dst + 4 * i, v);
        else stg4(dst + 4 * i, v);
    }
}

// Diagnostic (RNNT_LM_DEBUG=1): order-independent exact checksum of a buffer (sum of the 32-bit patterns), to find the first
// launch of a schedule whose output differs between two runs.
__global__ void debug_checksum(const float* __restrict__ p, long long n, unsigned long long* __restrict__ out) {
    unsigned long long acc = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        acc += (unsigned long long)__float_as_uint(p[i]);
    atomicAdd(out, acc);
}
