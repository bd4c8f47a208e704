// librnnt_hip.so — host side: context, weight ingest/packing, per-chunk launch sequences, C ABI.
// See include/rnnt_hip.h for the contract and DESIGN.md for the data layout in HBM.
#include "rnnt_kernels.hip.h"
#include "../../include/rnnt_hip.h"

#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <algorithm>
#include <array>
#include <vector>

namespace {

constexpr int D = RNNT_D, FF = RNNT_FF, L = RNNT_L, DK = RNNT_DK;
constexpr int BIG = INT_MAX;

struct LayerW {
    const float *ln_ffm_g, *ln_ffm_b, *w1m, *b1m, *w2m, *b2m;
    const float *ln_mha_g, *ln_mha_b, *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo, *pu, *pv;
    float* ptab;   // [5000][256] = pe * W_pos^T
    const float *wpos;
    const float *ln_conv_g, *ln_conv_b, *pw1, *bpw1, *wdw_t, *bdw, *bn_s, *bn_t, *pw2, *bpw2;
    const float *ln_ff_g, *ln_ff_b, *w1, *b1, *w2, *b2, *ln_fin_g, *ln_fin_b;
};

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> dims;
};

}  // namespace

struct rnnt_ctx {
    rnnt_config cfg;
    std::string err;
    std::map<std::string, HostTensor> host;
    bool finalized = false;
    int numerics = 0;

    // packed weights (one device blob)
    float* blob = nullptr;
    size_t blob_floats = 0;
    LayerW lw[L];
    const float *conv1_wt, *conv1_b, *conv2_w, *conv2_b, *emb_w, *emb_b, *pe, *after_g, *after_b;
    const float *ln_conv_g_all, *ln_conv_b_all, *glu0;
    const float *whh_il, *wih_il, *b_lstm_il, *pred_embed, *wpr, *bpr, *wenc, *benc, *wpf, *bpf, *wout, *bout;
    float* egate = nullptr;   // [vocab][1024] interleaved input-gate table

    // geometry
    int tmax = 0, t1max = 0, cap = 0, tcap = 0, fcap = 0, fstride = 0, vpad = 0;
    // activations
    float *y1 = nullptr, *y2 = nullptr, *x = nullptr, *hbuf = nullptr, *qbuf = nullptr, *abuf = nullptr, *dbuf = nullptr;
    float *kcache = nullptr, *vcache = nullptr, *gring = nullptr, *xring = nullptr;
    float *encbuf = nullptr, *encp = nullptr;
    // decode state
    // LSTM state: two buffers [2][B][256] per h and c; sel[b] says which one is committed, the other receives the candidate
    float *h = nullptr, *c = nullptr, *pred = nullptr, *z = nullptr, *logits = nullptr;
    int *tok = nullptr, *fidx = nullptr, *nsym = nullptr, *count = nullptr, *tokens = nullptr, *n_active = nullptr, *klen = nullptr, *sel = nullptr;
    unsigned long long* key = nullptr;
    int* dec_ctrl = nullptr;   // persistent decoder control block: [0] frames_ready, [1] error, [2] evaluations, [3..6] cooperative decoder
    int use_persistent = 1;
    int attn_stream = 1;       // RNNT_ATTN_STREAM=0: LDS-tiled attention kernel for every chunk
    int fuse_after_norm = 1;   // RNNT_FUSE_AFTER_NORM=0: keep after_norm as its own launch in the pipelined greedy path
    int overlap_ok = -1;       // -1 not probed; 1: kernels of the decode stream run concurrently with the caller's stream
    int use_coop = 0;          // RNNT_COOP=1: cooperative weights-stationary decoder (n_streams <= 64), experiment
    unsigned long long* flow_buf = nullptr;   // greedy_flow exchange words: xh [2][64][256] | xz [2][64][256] | xa [2][4][16][16][4]
    const float *wjc = nullptr, *bjc = nullptr;   // folded joint.pred_ffn o predictor.projection
    const float *wctc = nullptr, *bctc = nullptr; // ctc_head.ctc_lo (optional)
    // beam search: state pools [rows][n_steps+1][512] (ping-pong), per-row buffers
    int max_rows = 0;
    float *pool[2] = {nullptr, nullptr}, *bpred = nullptr, *bz = nullptr, *blogits = nullptr, *b_blank = nullptr, *b_toplp = nullptr;
    int *b_tok = nullptr, *b_frame = nullptr, *b_active = nullptr, *b_steps = nullptr, *b_toptok = nullptr, *b_srcrow = nullptr, *b_srcstep = nullptr;
    int pool_cur = 0;
    int* pinned = nullptr;   // host-pinned scratch (n_active read-back)
    float* scratch = nullptr;  // device scratch for getters / step API
    size_t scratch_floats = 0;

    // stream state (all streams lock step)
    int n_streams = 0;
    int cache_len = 0, kv_start = 0, conv_pos = 0;
    int frames_buffered = 0, frames_decoded = 0;
    int64_t launches = 0, greedy_steps = 0;
    // wavefront (whole-utterance) path: per-chunk x rows, per-layer scratch, subsampling slabs, descriptor tables
    float *wf_x = nullptr, *wf_h = nullptr, *wf_q = nullptr, *wf_a = nullptr, *wf_d = nullptr, *wf_y1 = nullptr, *wf_y2 = nullptr;
    int wf_slab = 0;
    int* wf_starts = nullptr;
    size_t wf_starts_cap = 0;
    GemmP* wf_gtab = nullptr; AttnP* wf_atab = nullptr; DwP* wf_dtab = nullptr; LnP* wf_ltab = nullptr;
    size_t wf_gcap = 0, wf_acap = 0, wf_dcap = 0, wf_lcap = 0;
    hipStream_t dec_stream = nullptr;          // decode runs here while the encoder wavefront runs on the caller's stream
    hipStream_t grp_stream[4] = {nullptr, nullptr, nullptr, nullptr};   // layer groups 1.. of the wavefront (group 0 = caller's stream)
    hipStream_t sub_stream = nullptr;          // subsampling slabs
    int wf_groups = 2, wf_sub_async = 1;       // RNNT_WF_GROUPS (1..4), RNNT_WF_SUB_ASYNC
    int wf_merge = 2;                          // RNNT_WF_MERGE (1..WF_MERGE_MAX): chunks of one layer per wavefront stage
    // descriptor tables of the last rnnt_encoder_chunks call, reused when the next call has the same plan and entry state
    struct WfLaunch { int type, off, n, maxM, maxT2; };   // type 0..7 gemm (ffn1m ffn2m qkv out pw1 pw2 ffn1 ffn2), 10 attn, 11 dw, 12 ln
    std::vector<WfLaunch> wf_seq;
    std::vector<std::array<int, 13>> wf_lstart;
    std::vector<int> wf_sc_first, wf_key;
    std::vector<hipEvent_t> ev_pool;
    // native beam bookkeeping (rnnt_beam_advance): per stream, hypotheses in device-row order
    struct Hyp { std::vector<int> tokens; double log_prob; };
    std::vector<std::vector<Hyp>> beams;
    int use_beam_chain = 1;    // RNNT_BEAM_CHAIN=0: launched extension steps (5 kernels + one host sync per step)
    // feature front-end (rnnt_fbank): DFT / mel matrices for (fb_rate, fb_nfft) and grow-only work buffers
    float *fb_dft = nullptr, *fb_mel = nullptr, *fb_pad = nullptr, *fb_spec = nullptr, *fb_pow = nullptr;
    size_t fb_pad_cap = 0, fb_spec_cap = 0, fb_pow_cap = 0;
    int fb_rate = 0, fb_nfft = 0;
    hipStream_t cap_stream = nullptr;          // stream-capture scratch stream
    struct DecGraph { int n_streams, k; hipGraphExec_t exec; };
    std::vector<DecGraph> dec_graphs;          // K greedy steps captured once per (n_streams, K)
    bool capturing = false;
    int use_graphs = 1;
    std::vector<hipEvent_t> wf_ev;
    // optional per-kernel-site timing with HIP events on the launch stream (bench.py roofline leg)
    int prof_tag = -1;
    std::vector<hipEvent_t> prof_ev;
    size_t prof_used = 0;
};

namespace {

int fail(rnnt_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(ctx, RNNT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define LAUNCHCHK(name)                                                                                \
    do {                                                                                               \
        hipError_t e_ = hipGetLastError();                                                             \
        if (e_ != hipSuccess) return fail(ctx, RNNT_ERR_HIP, "launch %s failed: %s", name, hipGetErrorString(e_)); \
        ctx->launches++;                                                                               \
    } while (0)

// launch-site tags (rnnt_profile_begin)
enum { TAG_NONE = 0, TAG_CONV1 = 1, TAG_CONV2 = 2, TAG_EMBED = 3, TAG_FFN1 = 4, TAG_FFN2 = 5, TAG_QKV = 6, TAG_ATTN = 7, TAG_ATTN_OUT = 8,
       TAG_PW1 = 9, TAG_DWCONV = 10, TAG_PW2 = 11, TAG_LN = 12, TAG_ENC_PROJ = 13, TAG_LSTM = 20, TAG_PRED_PROJ = 21,
       TAG_JOINT_TANH = 22, TAG_JOINT_OUT = 23, TAG_GREEDY_UPDATE = 24 };

struct ProfScope {   // records a start/stop event pair around one launch when its site is selected
    rnnt_ctx* ctx; hipStream_t s; bool on;
    ProfScope(rnnt_ctx* c, hipStream_t st, int tag) : ctx(c), s(st), on(c->prof_tag == tag && tag != TAG_NONE && !c->capturing) {
        if (on) {
            if (ctx->prof_used + 2 > ctx->prof_ev.size()) {
                for (int i = 0; i < 2; ++i) { hipEvent_t e; (void)hipEventCreate(&e); ctx->prof_ev.push_back(e); }
            }
            (void)hipEventRecord(ctx->prof_ev[ctx->prof_used], s);
        }
    }
    ~ProfScope() {
        if (on) { (void)hipEventRecord(ctx->prof_ev[ctx->prof_used + 1], s); ctx->prof_used += 2; }
    }
};

template <typename T>
int dmalloc(rnnt_ctx* ctx, T** p, size_t n) {
    if (hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)) != hipSuccess)
        return fail(ctx, RNNT_ERR_OOM, "hipMalloc of %zu bytes failed", n * sizeof(T));
    return RNNT_OK;
}

constexpr size_t FLOW_XH_WORDS = 2 * 64 * 256, FLOW_WORDS = 2 * FLOW_XH_WORDS + 2 * 4 * 16 * 16 * 4;   // greedy_flow exchange buffers
constexpr int WF_MERGE_MAX = 4;   // chunks of one layer per wavefront stage (rnnt_encoder_chunks), upper bound
inline int sub_len(int T) { return ((T - 3) / 2 + 1 - 3) / 2 + 1; }   // subsampling.py:188-193
inline int sub1_len(int T) { return (T - 3) / 2 + 1; }

GemmP plain_gemm(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M, int N, int K,
                 int epi = EPI_BIAS, float alpha = 1.f) {
    GemmP p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.W = W; p.bias = bias; p.C = C; p.R = nullptr; p.ln_g = nullptr; p.ln_b = nullptr;
    p.M = M; p.N = N; p.K = K;
    p.a_n1 = BIG; p.a_n2 = BIG; p.a_s0 = 0; p.a_s1 = 0; p.a_s2 = lda; p.a_seg = BIG; p.a_seg_stride = 0;
    p.ldw = ldw;
    p.c_n = BIG; p.c_r0 = 0; p.c_mod = BIG; p.c_s0 = 0; p.c_s1 = ldc;
    p.epi = epi; p.alpha = alpha;
    p.x_n = 1;
    return p;
}

// q = umulhi(n, magic) >> shift, exact for 0 <= n < 2^31 (round-up method: magic = ceil(2^(32+shift) / d))
inline void div_magic(int d, unsigned& magic, int& shift) {
    if (d <= 1) { magic = 0; shift = 0; return; }
    int l = 0;
    while ((1ll << l) < d) ++l;          // l = ceil(log2 d)
    shift = l - 1;
    const unsigned long long num = 1ull << (32 + shift);
    magic = (unsigned)((num + (unsigned long long)d - 1) / (unsigned long long)d);
}

// fast-path flags and division magics of one GEMM descriptor (gemm16's a_row_off / c_row_off)
int prepare_gemm(rnnt_ctx* ctx, GemmP& g) {
    g.a_plain = (g.a_n1 == BIG && g.a_n2 == BIG && g.a_seg == BIG) ? 1 : 0;
    g.c_plain = (g.c_n == BIG && g.c_r0 == 0) ? 1 : 0;
    if ((long long)g.M >= (1ll << 31) || (long long)g.K >= (1ll << 31)) return fail(ctx, RNNT_ERR_SHAPE, "gemm index range too large");
    if (!g.a_plain) {
        if (g.a_n1 == BIG) g.a_n1 = g.M > 0 ? g.M + 1 : 1;   // quotient 0, remainder m
        if (g.a_n2 == BIG) g.a_n2 = g.M > 0 ? g.M + 1 : 1;
        if (g.a_seg == BIG) { g.a_seg = g.K + 1; g.a_seg_stride = 0; }
    }
    if (!g.c_plain && g.c_n == BIG) g.c_n = g.M > 0 ? g.M + 1 : 1;
    div_magic(g.a_n1, g.a_n1_magic, g.a_n1_shift); div_magic(g.a_n2, g.a_n2_magic, g.a_n2_shift);
    div_magic(g.a_seg, g.a_seg_magic, g.a_seg_shift); div_magic(g.c_n, g.c_n_magic, g.c_n_shift);
    div_magic(g.x_n, g.x_n_magic, g.x_n_shift);
    return RNNT_OK;
}

template <int WK, int NT>
void launch_gemm16(hipStream_t s, const GemmBatch& gb, int maxM, int maxN, int ng) {
    dim3 grid((maxN + 16 * NT - 1) / (16 * NT), (maxM + 15) / 16, ng);
    hipLaunchKernelGGL((gemm16<WK, 1, NT>), grid, dim3(64 * WK), 0, s, gb);
}

// wk > 0: gemm32 (32x32 tiles, large-M implicit-GEMM conv2); wk == 0: gemm16 with a shape heuristic
// gemm_ns with the XCD-aware 1-D grid (see the kernel): ceil(ntm / 8) * 8 * ntn workgroups per descriptor
static int prefetch_depth() {   // K blocks in flight per workgroup (register ring of gemm_ns_body): 1 or 2
    static const int pd = getenv("RNNT_GEMM_PD") ? atoi(getenv("RNNT_GEMM_PD")) : 2;
    return pd;
}
template <int MT, int NT, bool ATANH = false, bool ANT = false>
void launch_gemm_ns(hipStream_t s, const GemmBatch& gb, int maxM, int maxN, int ng) {
    const int ntn = (maxN + 32 * NT - 1) / (32 * NT), ntm = (maxM + 32 * MT - 1) / (32 * MT);
    dim3 grid(((ntm + 7) / 8) * 8 * ntn, 1, ng);
    switch (prefetch_depth()) {
        case 1: hipLaunchKernelGGL((gemm_ns<MT, NT, 32, 1, ATANH, ANT>), grid, dim3(256), 0, s, gb, ntn, ntm); break;
        default: hipLaunchKernelGGL((gemm_ns<MT, NT, 32, 2, ATANH, ANT>), grid, dim3(256), 0, s, gb, ntn, ntm); break;
    }
}

int launch_gemm(rnnt_ctx* ctx, hipStream_t s, int wk, const GemmP* gs, int ng, int tag = TAG_NONE) {
    ProfScope prof(ctx, s, tag);
    GemmBatch gb;
    memset(&gb, 0, sizeof(gb));
    int maxM = 0, maxN = 0;
    for (int i = 0; i < ng; ++i) {
        gb.g[i] = gs[i];
        int rc = prepare_gemm(ctx, gb.g[i]);
        if (rc) return rc;
        if (gs[i].ln_g && gs[i].K != 256) return fail(ctx, RNNT_ERR_SHAPE, "LayerNorm prologue needs K=256");
        maxM = gs[i].M > maxM ? gs[i].M : maxM;
        maxN = gs[i].N > maxN ? gs[i].N : maxN;
    }
    if (maxM <= 0 || maxN <= 0) return RNNT_OK;
    if (wk == 0) {
        const int K = gs[0].K;
        for (int i = 0; i < ng; ++i)
            if (gs[i].K != K) return fail(ctx, RNNT_ERR_SHAPE, "grouped gemm needs one K");
        const int epi0 = gs[0].epi;
        if (maxM >= 1024 && K % 32 == 0 && epi0 != EPI_LSTM && epi0 != EPI_ARGMAX) {
            // large M (full-context encoder, batched subsampling, joint lattice): LDS-tiled kernel, no split-K
            if (gs[0].a_tanh) launch_gemm_ns<2, 2, true>(s, gb, maxM, maxN, ng);
            else if (maxN >= 512) launch_gemm_ns<2, 2>(s, gb, maxM, maxN, ng);
            else launch_gemm_ns<1, 2>(s, gb, maxM, maxN, ng);
            LAUNCHCHK("gemm_ns");
            return RNNT_OK;
        }
        const bool wide = maxN >= 512 && gs[0].epi != EPI_LSTM ? true : (maxN >= 512);
        const int wkk = K >= 1024 ? 8 : 4;
        if (K % (wkk * 16) != 0) return fail(ctx, RNNT_ERR_SHAPE, "gemm16 K=%d not divisible by %d", K, wkk * 16);
        if (wide) { if (wkk == 8) launch_gemm16<8, 2>(s, gb, maxM, maxN, ng); else launch_gemm16<4, 2>(s, gb, maxM, maxN, ng); }
        else { if (wkk == 8) launch_gemm16<8, 1>(s, gb, maxM, maxN, ng); else launch_gemm16<4, 1>(s, gb, maxM, maxN, ng); }
        LAUNCHCHK("gemm16");
        return RNNT_OK;
    }
    for (int i = 0; i < ng; ++i)
        if (gs[i].K % (wk * 8) != 0) return fail(ctx, RNNT_ERR_SHAPE, "gemm K=%d not divisible by %d", gs[i].K, wk * 8);
    dim3 grid((maxN + 31) / 32, (maxM + 31) / 32, ng);
    size_t lds = (size_t)(wk * 1024 + 64) * sizeof(float);
    switch (wk) {
        case 1: hipLaunchKernelGGL(gemm32<1>, grid, dim3(64), lds, s, gb); break;
        case 2: hipLaunchKernelGGL(gemm32<2>, grid, dim3(128), lds, s, gb); break;
        case 4: hipLaunchKernelGGL(gemm32<4>, grid, dim3(256), lds, s, gb); break;
        case 8: hipLaunchKernelGGL(gemm32<8>, grid, dim3(512), lds, s, gb); break;
        case 16: hipLaunchKernelGGL(gemm32<16>, grid, dim3(1024), lds, s, gb); break;
        default: return fail(ctx, RNNT_ERR_ARG, "bad WK %d", wk);
    }
    LAUNCHCHK("gemm32");
    return RNNT_OK;
}

inline int grid_for(long long n, int block = 256) {
    long long g = (n + block - 1) / block;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

const HostTensor* find(rnnt_ctx* ctx, const std::string& name) {
    auto it = ctx->host.find(name);
    return it == ctx->host.end() ? nullptr : &it->second;
}

// ---- descriptors of one Conformer block over rows [M = B*tq] of `x` ------------------------------------
// (ConformerEncoderLayer.forward, wenet/transformer/encoder_layer.py:188-265).  Shared by the eager per-chunk
// path (descriptors passed by value) and the wavefront path (descriptor tables in device memory).
struct LayerDescs {
    GemmP ffn1m, ffn2m, qkv[3], out, pw1, pw2, ffn1, ffn2;
    AttnP attn;
    DwP dw;
    LnP lnf;
};
struct LayerBufs { float *x, *hbuf, *qbuf, *abuf, *dbuf; };

int build_layer(rnnt_ctx* ctx, int l, int B, int tq, int T2, int kv_row0, int pos_start, int ring_pos, const int* klen_dev,
                const LayerBufs& bf, LayerDescs& d) {
    const LayerW& w = ctx->lw[l];
    const int M = B * tq;
    float* kc = ctx->kcache + (size_t)l * ctx->cfg.max_streams * ctx->tcap * D;
    float* vc = ctx->vcache + (size_t)l * ctx->cfg.max_streams * ctx->tcap * D;
    float* gr = ctx->gring + (size_t)l * ctx->cfg.max_streams * ctx->cap * D;
    float* xr = ctx->xring + (size_t)l * ctx->cfg.max_streams * ctx->cap * D;
    // x += 0.5 * FFN_macaron(LN(x))
    d.ffn1m = plain_gemm(bf.x, D, w.w1m, D, w.b1m, bf.hbuf, FF, M, FF, D, EPI_SILU);
    d.ffn1m.ln_g = w.ln_ffm_g; d.ffn1m.ln_b = w.ln_ffm_b;
    d.ffn2m = plain_gemm(bf.hbuf, FF, w.w2m, FF, w.b2m, bf.x, D, M, D, FF, EPI_RESID, 0.5f);
    d.ffn2m.R = bf.x;
    // x += linear_out(attention(LN(x))): q to a buffer, the new K/V rows appended behind the cached ones
    d.qkv[0] = plain_gemm(bf.x, D, w.wq, D, w.bq, bf.qbuf, D, M, D, D);
    d.qkv[1] = plain_gemm(bf.x, D, w.wk, D, w.bk, kc, D, M, D, D);
    d.qkv[2] = plain_gemm(bf.x, D, w.wv, D, w.bv, vc, D, M, D, D);
    for (int i = 0; i < 3; ++i) { d.qkv[i].ln_g = w.ln_mha_g; d.qkv[i].ln_b = w.ln_mha_b; }
    for (int i = 1; i < 3; ++i) {
        d.qkv[i].c_n = tq; d.qkv[i].c_s0 = (long long)ctx->tcap * D; d.qkv[i].c_r0 = kv_row0 + (T2 - tq); d.qkv[i].c_mod = BIG; d.qkv[i].c_s1 = D;
    }
    d.attn = AttnP{bf.qbuf, kc, vc, w.ptab, w.pu, w.pv, klen_dev, bf.abuf, tq, T2, kv_row0, pos_start, (long long)ctx->tcap};
    d.out = plain_gemm(bf.abuf, D, w.wo, D, w.bo, bf.x, D, M, D, D, EPI_RESID, 1.0f);
    d.out.R = bf.x;
    // x += conv_module(LN(x))
    d.pw1 = plain_gemm(bf.x, D, w.pw1, D, w.bpw1, gr, D, M, 2 * D, D, EPI_GLU);
    d.pw1.ln_g = w.ln_conv_g; d.pw1.ln_b = w.ln_conv_b;
    d.pw1.c_n = tq; d.pw1.c_s0 = (long long)ctx->cap * D; d.pw1.c_r0 = ring_pos % ctx->cap; d.pw1.c_mod = ctx->cap; d.pw1.c_s1 = D;
    d.dw = DwP{gr, w.wdw_t, w.bdw, w.bn_s, w.bn_t, bf.dbuf, bf.x, xr, B, tq, ctx->cap, ring_pos};
    d.pw2 = plain_gemm(bf.dbuf, D, w.pw2, D, w.bpw2, bf.x, D, M, D, D, EPI_RESID, 1.0f);
    d.pw2.R = bf.x;
    // x += 0.5 * FFN(LN(x)); x = LN_final(x)
    d.ffn1 = plain_gemm(bf.x, D, w.w1, D, w.b1, bf.hbuf, FF, M, FF, D, EPI_SILU);
    d.ffn1.ln_g = w.ln_ff_g; d.ffn1.ln_b = w.ln_ff_b;
    d.ffn2 = plain_gemm(bf.hbuf, FF, w.w2, FF, w.b2, bf.x, D, M, D, FF, EPI_RESID, 0.5f);
    d.ffn2.R = bf.x;
    d.lnf = LnP{bf.x, w.ln_fin_g, w.ln_fin_b, bf.x, M, BIG, 0, 0LL, (long long)D};
    return RNNT_OK;
}

// streaming chunks (<= 4 new frames): direct-stream kernel; dynamic LDS = 4 score rows + the PV partial sums
static bool attn_stream_ok(const rnnt_ctx* ctx, int tq, int T2) {
    return ctx->attn_stream && tq <= 4 && T2 >= 1 && T2 <= 4096;
}
static int attn_t2cap(int T2) { return (T2 + 63) / 64 * 64; }
static size_t attn_stream_lds(int t2cap) { return (size_t)(4 * t2cap + 16 * 4 * RNNT_DK) * sizeof(float); }

int launch_attn(rnnt_ctx* ctx, hipStream_t s, const AttnP& a, int B) {
    ProfScope prof(ctx, s, TAG_ATTN);
    if (attn_stream_ok(ctx, a.tq, a.T2)) {
        const int cap = attn_t2cap(a.T2);
        hipLaunchKernelGGL(rel_attention_stream, dim3(B * RNNT_H), dim3(256), attn_stream_lds(cap), s, a, cap);
        LAUNCHCHK("rel_attention_stream");
        return RNNT_OK;
    }
    const int nq = a.tq <= 4 ? 1 : (a.tq <= 8 ? 2 : 4);
    dim3 grid(B * RNNT_H, (a.tq + 4 * nq - 1) / (4 * nq));
    if (nq == 1) hipLaunchKernelGGL(rel_attention<1>, grid, dim3(256), 0, s, a);
    else if (nq == 2) hipLaunchKernelGGL(rel_attention<2>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(rel_attention<4>, grid, dim3(256), 0, s, a);
    LAUNCHCHK("rel_attention");
    return RNNT_OK;
}
int launch_dw(rnnt_ctx* ctx, hipStream_t s, const DwP& d) {
    ProfScope prof(ctx, s, TAG_DWCONV);
    hipLaunchKernelGGL(dwconv_bn_silu, dim3(grid_for((long long)d.B * d.tq * D)), dim3(256), 0, s, d);
    LAUNCHCHK("dwconv_bn_silu");
    return RNNT_OK;
}
int launch_ln(rnnt_ctx* ctx, hipStream_t s, const LnP& p) {
    hipLaunchKernelGGL(layer_norm, dim3((p.M + 3) / 4), dim3(256), 0, s, p);
    LAUNCHCHK("layer_norm");
    return RNNT_OK;
}

int run_layer(rnnt_ctx* ctx, hipStream_t s, int l, int B, int tq, int T2, int kv_row0, int pos_start, int ring_pos,
              const int* klen_dev) {
    LayerDescs d;
    LayerBufs bf{ctx->x, ctx->hbuf, ctx->qbuf, ctx->abuf, ctx->dbuf};
    int rc = build_layer(ctx, l, B, tq, T2, kv_row0, pos_start, ring_pos, klen_dev, bf, d);
    if (rc) return rc;
    if ((rc = launch_gemm(ctx, s, 0, &d.ffn1m, 1, TAG_FFN1))) return rc;
    if ((rc = launch_gemm(ctx, s, 0, &d.ffn2m, 1, TAG_FFN2))) return rc;
    if ((rc = launch_gemm(ctx, s, 0, d.qkv, 3, TAG_QKV))) return rc;
    if ((rc = launch_attn(ctx, s, d.attn, B))) return rc;
    if ((rc = launch_gemm(ctx, s, 0, &d.out, 1, TAG_ATTN_OUT))) return rc;
    if ((rc = launch_gemm(ctx, s, 0, &d.pw1, 1, TAG_PW1))) return rc;
    if ((rc = launch_dw(ctx, s, d.dw))) return rc;
    if ((rc = launch_gemm(ctx, s, 0, &d.pw2, 1, TAG_PW2))) return rc;
    if ((rc = launch_gemm(ctx, s, 0, &d.ffn1, 1, TAG_FFN1))) return rc;
    if ((rc = launch_gemm(ctx, s, 0, &d.ffn2, 1, TAG_FFN2))) return rc;
    return launch_ln(ctx, s, d.lnf);
}

// Conv2dSubsampling4 (+ x16) (subsampling.py:203-228) of `nc` equal-length chunks of every stream at once:
// virtual stream v = c*B + b; output rows (v, r) -> xout[(v*tq + r)][256].  starts_dev == null: one chunk at 0.
int run_subsample(rnnt_ctx* ctx, hipStream_t s, const float* fbank, int B, int Tstride, int T, const int* starts_dev, int nc,
                  float* y1, float* y2, float* xout) {
    const int t1 = sub1_len(T), tq = sub_len(T);
    const int VB = nc * B;
    int rc;
    { ProfScope prof(ctx, s, TAG_CONV1);
    hipLaunchKernelGGL(conv1_relu, dim3(grid_for((long long)VB * t1 * RNNT_F1 * D)), dim3(256), 0, s, fbank, ctx->conv1_wt, ctx->conv1_b,
                       y1, B, Tstride, t1, starts_dev, nc); }
    LAUNCHCHK("conv1_relu");
    // conv2 as implicit GEMM: rows (v,t',f), K = (kh, kw, ci) = 3 segments of 768 contiguous floats of y1
    GemmP g = plain_gemm(y1, 0, ctx->conv2_w, 2304, ctx->conv2_b, y2, D, VB * tq * RNNT_FSUB, D, 2304, EPI_RELU);
    g.a_n1 = tq * RNNT_FSUB; g.a_n2 = RNNT_FSUB;
    g.a_s0 = (long long)t1 * RNNT_F1 * D; g.a_s1 = 2LL * RNNT_F1 * D; g.a_s2 = 2LL * D;
    g.a_seg = 768; g.a_seg_stride = (long long)RNNT_F1 * D;
    static const int conv2_lds = getenv("RNNT_CONV2_LDS") ? atoi(getenv("RNNT_CONV2_LDS")) : 1;
    if (conv2_lds && g.M >= 2048) {   // big M: LDS-tiled 64x64 tiles (full-line operand staging); N = 256 -> 4 column tiles
        ProfScope prof(ctx, s, TAG_CONV2);
        GemmBatch gb;
        memset(&gb, 0, sizeof(gb));
        gb.g[0] = g;
        if ((rc = prepare_gemm(ctx, gb.g[0]))) return rc;
        if (conv2_lds == 2 && nc > 1) launch_gemm_ns<2, 2, false, true>(s, gb, g.M, g.N, 1);   // experiment: non-temporal A
        else launch_gemm_ns<2, 2>(s, gb, g.M, g.N, 1);
        LAUNCHCHK("gemm_ns");
    } else if ((rc = launch_gemm(ctx, s, 8, &g, 1, TAG_CONV2))) return rc;
    // Linear(4864 -> 256) * sqrt(256); y2 is [VB*t', f*256 + c] (weight columns permuted to match)
    GemmP go = plain_gemm(y2, RNNT_FSUB * D, ctx->emb_w, RNNT_FSUB * D, ctx->emb_b, xout, D, VB * tq, D, RNNT_FSUB * D, EPI_SCALE, 16.0f);
    if ((rc = launch_gemm(ctx, s, 0, &go, 1, TAG_EMBED))) return rc;
    return RNNT_OK;
}

template <int WK, int MT, int NT>
void launch_gemm16_tab(hipStream_t s, const GemmP* tab, int n, int maxM, int maxN) {
    dim3 grid((maxN + 16 * NT - 1) / (16 * NT), (maxM + 16 * MT - 1) / (16 * MT), n);
    hipLaunchKernelGGL((gemm16_tab<WK, MT, NT>), grid, dim3(64 * WK), 0, s, tab);
}
// n descriptors of one shape class (same N, K) in device memory.  Tile choice: with n >= 6 groups there are enough
// workgroups to spend registers on operand reuse (64x64 / 32x64 tiles); few groups keep the 16-row tiles.
int launch_gemm_tab(rnnt_ctx* ctx, hipStream_t s, const GemmP* tab_dev, int n, int maxM, int N, int K, int tag) {
    ProfScope prof(ctx, s, tag);
    static const int ns_mode = getenv("RNNT_GEMM_NS") ? atoi(getenv("RNNT_GEMM_NS")) : 1;
    static const int ns_min = getenv("RNNT_NS_MIN_GROUPS") ? atoi(getenv("RNNT_NS_MIN_GROUPS")) : 2;   // pipeline fill/drain stages have few pairs
    if (ns_mode && n >= ns_min && K % 32 == 0) {   // enough groups: no split-K, epilogue from registers
        // tile choice from tools/microbench2.hip (12 groups x 192 rows): the kernel is occupancy/latency-bound, so the
        // narrow shapes want many small workgroups; only K = 1024 profits from 64-deep K blocks (half the barriers)
        // XCDs per descriptor (see gemm_ns_tab): the largest split that keeps the groups balanced
        static const int x_env = getenv("RNNT_XCD_X") ? atoi(getenv("RNNT_XCD_X")) : 0;
        auto pick_x = [&](int ntn) {
            if (x_env == 1 || x_env == 2 || x_env == 4 || x_env == 8) return x_env;
            for (int X = 2; X < 8; X *= 2)
                if (n % (8 / X) == 0 && ntn % X == 0) return X;
            return 8;
        };
        auto grid_for_tab = [&](int ntn, int ntm, int X) {
            const int dpg = (n + 8 / X - 1) / (8 / X), cpx = (ntn + X - 1) / X;
            return dim3(8 * dpg * cpx * ntm);
        };
#define NS_TAB(MT_, NT_, BK_)                                                                                              \
    switch (prefetch_depth()) {                                                                                            \
        case 1: hipLaunchKernelGGL((gemm_ns_tab<MT_, NT_, BK_, 1>), grid_for_tab(ntn, ntm, X), dim3(256), 0, s, tab_dev, n, ntn, ntm, X); break; \
        default: hipLaunchKernelGGL((gemm_ns_tab<MT_, NT_, BK_, 2>), grid_for_tab(ntn, ntm, X), dim3(256), 0, s, tab_dev, n, ntn, ntm, X); break; \
    }
        if (N >= 512) {                       // ffn1 / pointwise_conv1: 32x64 tiles
            const int ntn = (N + 63) / 64, ntm = (maxM + 31) / 32, X = pick_x(ntn);
            NS_TAB(1, 2, 32)
        } else if (K >= 1024) {               // ffn2: 32x32 tiles, BK = 64
            const int ntn = (N + 31) / 32, ntm = (maxM + 31) / 32, X = pick_x(ntn);
            NS_TAB(1, 1, 64)
        } else {                              // q/k/v, linear_out, pointwise_conv2: 32x32 tiles
            const int ntn = (N + 31) / 32, ntm = (maxM + 31) / 32, X = pick_x(ntn);
            NS_TAB(1, 1, 32)
        }
#undef NS_TAB
        LAUNCHCHK("gemm_ns_tab");
        return RNNT_OK;
    }
    const int wk = K >= 1024 ? 8 : 4;
    if (K % (wk * 16) != 0) return fail(ctx, RNNT_ERR_SHAPE, "gemm16 K=%d not divisible by %d", K, wk * 16);
    const long long out = (long long)n * maxM * N;
    if (out >= 256ll * 64 * 64 * 2 && wk == 4) launch_gemm16_tab<4, 4, 4>(s, tab_dev, n, maxM, N);
    else if (out >= 256ll * 32 * 64 && wk == 4) launch_gemm16_tab<4, 2, 4>(s, tab_dev, n, maxM, N);
    else if (out >= 256ll * 32 * 64 && wk == 8) launch_gemm16_tab<8, 2, 4>(s, tab_dev, n, maxM, N);
    else if (N >= 512) { if (wk == 8) launch_gemm16_tab<8, 1, 2>(s, tab_dev, n, maxM, N); else launch_gemm16_tab<4, 1, 2>(s, tab_dev, n, maxM, N); }
    else { if (wk == 8) launch_gemm16_tab<8, 1, 1>(s, tab_dev, n, maxM, N); else launch_gemm16_tab<4, 1, 1>(s, tab_dev, n, maxM, N); }
    LAUNCHCHK("gemm16_tab");
    return RNNT_OK;
}

// `n` lock-step greedy evaluations for all streams over the buffered frames (n_frames in device memory)
// (_decode_chunk_streaming_logic inner loop, online_rnnt_model.py:196-220), 4 launches per evaluation:
//   greedy_decide   apply the previous argmax to every stream's state machine (token / frame / state-buffer select)
//   LSTM cell       gates = E[tok] + h * W_hh^T, candidate (h', c') into the non-committed buffer (predictor.py:200-204)
//   joint tanh      z = tanh(enc_ffn(enc)[t_b] + (pred_ffn o projection)(h')) (joint.py:54-66, folded Linear pair)
//   joint out       logits = z * W_out^T + b, argmax fused into the epilogue (packed atomicMax; online_rnnt_model.py:212)
// Streams without frames idle.
GreedyState greedy_state(rnnt_ctx* ctx) {
    return GreedyState{ctx->tok, ctx->fidx, ctx->nsym, ctx->count, ctx->tokens, ctx->sel, ctx->key, ctx->n_active, ctx->pinned + 8};
}

int greedy_steps_raw(rnnt_ctx* ctx, hipStream_t s, int n) {
    const int B = ctx->n_streams, V = ctx->cfg.vocab_size;
    const long long bs = (long long)ctx->cfg.max_streams * D;   // floats between the two state buffers
    GreedyState st = greedy_state(ctx);
    int rc;
    for (int it = 0; it < n; ++it) {
        hipLaunchKernelGGL(greedy_decide, dim3(1), dim3(64), 0, s, B, ctx->cfg.blank_id, ctx->cfg.n_steps, ctx->cfg.max_tokens, 0, st);
        LAUNCHCHK("greedy_decide");
        GemmP g1 = plain_gemm(ctx->h, D, ctx->whh_il, D, nullptr, ctx->h, D, B, 4 * D, D, EPI_LSTM);
        g1.X = ctx->egate; g1.I = ctx->tok; g1.X2 = ctx->c; g1.Y2 = ctx->c;
        g1.Asel = ctx->sel; g1.asel_stride = bs; g1.asel_invert = 0;
        g1.act_idx = ctx->fidx; g1.act_lim = ctx->n_active + 2;
        if ((rc = launch_gemm(ctx, s, 0, &g1, 1, TAG_LSTM))) return rc;
        GemmP g3 = plain_gemm(ctx->h, D, ctx->wjc, D, ctx->bjc, ctx->z, D, B, D, D, EPI_TANH_ADD);
        g3.Asel = ctx->sel; g3.asel_stride = bs; g3.asel_invert = 1;   // candidate h' lives in the other buffer
        g3.X = ctx->encp; g3.I = ctx->fidx; g3.x_n = 1; g3.x_s0 = (long long)ctx->fstride * D; g3.x_s1 = D;
        g3.act_idx = ctx->fidx; g3.act_lim = ctx->n_active + 2;
        if ((rc = launch_gemm(ctx, s, 0, &g3, 1, TAG_JOINT_TANH))) return rc;
        GemmP g4 = plain_gemm(ctx->z, D, ctx->wout, D, ctx->bout, ctx->logits, ctx->vpad, B, V, D, EPI_ARGMAX);
        g4.key = ctx->key; g4.I = ctx->fidx; g4.nframes = ctx->n_active + 2;
        g4.act_idx = ctx->fidx; g4.act_lim = ctx->n_active + 2;
        if ((rc = launch_gemm(ctx, s, 0, &g4, 1, TAG_JOINT_OUT))) return rc;
    }
    return RNNT_OK;
}

// n greedy steps over the first n_frames buffered frames; the step sequence has static arguments, so it is captured
// once per (n_streams, n) into a hipGraph and replayed with ONE host call (the path is launch-bound: 5 kernels/step).
int greedy_steps(rnnt_ctx* ctx, hipStream_t s, int n, int n_frames) {
    int rc;
    hipLaunchKernelGGL(fill_i32, dim3(1), dim3(64), 0, s, ctx->n_active + 2, n_frames, 1LL);
    LAUNCHCHK("fill_i32");
    ctx->greedy_steps += n;
    if (!ctx->use_graphs || ctx->prof_tag >= 20) return greedy_steps_raw(ctx, s, n);   // decode sites being timed: eager
    for (auto& g : ctx->dec_graphs)
        if (g.n_streams == ctx->n_streams && g.k == n) {
            HIPCHK(hipGraphLaunch(g.exec, s));
            ctx->launches += 4 * n;
            return RNNT_OK;
        }
    if (!ctx->cap_stream) HIPCHK(hipStreamCreateWithFlags(&ctx->cap_stream, hipStreamNonBlocking));
    hipGraph_t graph = nullptr;
    const int64_t l0 = ctx->launches;
    HIPCHK(hipStreamBeginCapture(ctx->cap_stream, hipStreamCaptureModeThreadLocal));
    ctx->capturing = true;
    rc = greedy_steps_raw(ctx, ctx->cap_stream, n);
    ctx->capturing = false;
    hipError_t e = hipStreamEndCapture(ctx->cap_stream, &graph);
    ctx->launches = l0;
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return fail(ctx, RNNT_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(ctx, RNNT_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    ctx->dec_graphs.push_back({ctx->n_streams, n, exec});
    HIPCHK(hipGraphLaunch(exec, s));
    ctx->launches += 4 * n;
    return RNNT_OK;
}

// run step batches until every stream has consumed all n_frames frames (host checks a device counter)
int greedy_drain(rnnt_ctx* ctx, hipStream_t s, int n_frames, int done_steps) {
    hipLaunchKernelGGL(fill_i32, dim3(1), dim3(64), 0, s, ctx->n_active + 2, n_frames, 1LL);
    LAUNCHCHK("fill_i32");
    const int max_steps = (n_frames - ctx->frames_decoded) * (ctx->cfg.n_steps + 1) + 8;
    int rc;
    while (true) {
        // apply the last evaluation and count the streams that still have frames
        hipLaunchKernelGGL(greedy_decide, dim3(1), dim3(64), 0, s, ctx->n_streams, ctx->cfg.blank_id, ctx->cfg.n_steps, ctx->cfg.max_tokens, 1,
                           greedy_state(ctx));
        LAUNCHCHK("greedy_decide");
        HIPCHK(hipMemcpyAsync(ctx->pinned, ctx->n_active, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (ctx->pinned[0] <= 0) break;
        if (done_steps > max_steps) return fail(ctx, RNNT_ERR_STATE, "greedy decode did not terminate");
        static const int dstep = getenv("RNNT_DRAIN_STEPS") ? atoi(getenv("RNNT_DRAIN_STEPS")) : 4;
        if ((rc = greedy_steps(ctx, s, dstep, n_frames))) return rc;
        done_steps += dstep;
    }
    return RNNT_OK;
}

// Persistent greedy decoder: one resident workgroup per 2 streams decodes every frame up to n_total, waiting on
// dec_ctrl[0] (frames_ready).  The control block must have been initialised on a stream this one is ordered after.
int launch_persistent_decoder(rnnt_ctx* ctx, hipStream_t s, int n_total, int n_steps_override = 0, const int* nlim = nullptr) {
    if (ctx->use_coop && ctx->n_streams <= 64 && !nlim && !n_steps_override) {
        // cooperative decoder: 4 x 16 resident workgroups, weights stationary in LDS, tagged-word exchanges
        FlowP c;
        memset(&c, 0, sizeof(c));
        c.whh = ctx->whh_il; c.egate = ctx->egate; c.wjc = ctx->wjc; c.bjc = ctx->bjc; c.wout = ctx->wout; c.bout = ctx->bout;
        c.encp = ctx->encp; c.h = ctx->h; c.c = ctx->c; c.sel = ctx->sel; c.tok = ctx->tok; c.fidx = ctx->fidx;
        c.nsym = ctx->nsym; c.count = ctx->count; c.tokens = ctx->tokens; c.ctrl = ctx->dec_ctrl;
        c.xh = ctx->flow_buf; c.xz = ctx->flow_buf + FLOW_XH_WORDS; c.xa = ctx->flow_buf + 2 * FLOW_XH_WORDS;
        c.fstride_f = (long long)ctx->fstride * D; c.bstride = (long long)ctx->cfg.max_streams * D;
        c.B = ctx->n_streams; c.vocab = ctx->cfg.vocab_size; c.blank = ctx->cfg.blank_id; c.n_steps = ctx->cfg.n_steps;
        c.max_tokens = ctx->cfg.max_tokens; c.n_total = n_total;
        c.timeout_ticks = 300000000ll;   // 3 s per wait (100 MHz counter)
        static const bool fdbg = getenv("RNNT_COOP_DBG") != nullptr;
        c.dbg = fdbg ? reinterpret_cast<long long*>(ctx->flow_buf + FLOW_WORDS) : nullptr;
        hipLaunchKernelGGL(greedy_flow, dim3(FLOW_G), dim3(256), 0, s, c);
        LAUNCHCHK("greedy_flow");
        return RNNT_OK;
    }
    DecP d;
    memset(&d, 0, sizeof(d));
    d.whh = ctx->whh_il; d.egate = ctx->egate; d.wjc = ctx->wjc; d.bjc = ctx->bjc; d.wout = ctx->wout; d.bout = ctx->bout;
    d.encp = ctx->encp; d.h = ctx->h; d.c = ctx->c; d.sel = ctx->sel; d.tok = ctx->tok; d.fidx = ctx->fidx; d.nsym = ctx->nsym;
    d.count = ctx->count; d.tokens = ctx->tokens; d.ctrl = ctx->dec_ctrl;
    d.fstride_f = (long long)ctx->fstride * D; d.bstride = (long long)ctx->cfg.max_streams * D;
    d.B = ctx->n_streams; d.vocab = ctx->cfg.vocab_size; d.blank = ctx->cfg.blank_id; d.n_steps = ctx->cfg.n_steps;
    d.max_tokens = ctx->cfg.max_tokens; d.n_total = n_total;
    if (n_steps_override > 0) d.n_steps = n_steps_override;
    d.nlim = nlim;
    d.timeout_ticks = 500000000ll;   // 5 s of the 100 MHz real-time counter: every wait in the kernel is bounded
    const int B = ctx->n_streams;
    static const int kf = getenv("RNNT_DEC_KF") ? atoi(getenv("RNNT_DEC_KF")) : 4;   // frames per vocabulary pass
    if (kf == 1) hipLaunchKernelGGL(greedy_stream<1>, dim3(B), dim3(512), 0, s, d);
    else if (kf == 2) hipLaunchKernelGGL(greedy_stream<2>, dim3(B), dim3(512), 0, s, d);
    else if (kf == 8) hipLaunchKernelGGL(greedy_stream<8>, dim3(B), dim3(512), 0, s, d);
    else hipLaunchKernelGGL(greedy_stream<4>, dim3(B), dim3(512), 0, s, d);
    LAUNCHCHK("greedy_stream");
    return RNNT_OK;
}

int init_decoder_ctrl(rnnt_ctx* ctx, hipStream_t s, int frames_ready) {
    if (ctx->use_coop && ctx->n_streams <= 64)   // tags restart at 1 every launch: no word of an earlier launch may survive
        HIPCHK(hipMemsetAsync(ctx->flow_buf, 0, (size_t)FLOW_WORDS * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(fill_i32, dim3(1), dim3(64), 0, s, ctx->dec_ctrl, 0, 32LL);
    LAUNCHCHK("fill_i32");
    hipLaunchKernelGGL(publish_frames, dim3(1), dim3(1), 0, s, ctx->dec_ctrl, frames_ready);
    LAUNCHCHK("publish_frames");
    return RNNT_OK;
}

// One-time check that a kernel on `s2` can stay resident while kernels on `s` run (what the pipelined resident decoder
// relies on).  Bounded to 20 ms; on failure the pipelined path falls back to graph-launched evaluation batches.
int probe_overlap(rnnt_ctx* ctx, hipStream_t s, hipStream_t s2) {
    hipLaunchKernelGGL(fill_i32, dim3(1), dim3(64), 0, s, ctx->dec_ctrl, 0, 32LL);
    LAUNCHCHK("fill_i32");
    HIPCHK(hipStreamSynchronize(s));
    hipLaunchKernelGGL(probe_overlap_wait, dim3(1), dim3(1), 0, s2, ctx->dec_ctrl, 2000000LL);
    LAUNCHCHK("probe_overlap_wait");
    hipLaunchKernelGGL(publish_frames, dim3(1), dim3(1), 0, s, ctx->dec_ctrl, 1);
    LAUNCHCHK("publish_frames");
    HIPCHK(hipStreamSynchronize(s2));
    HIPCHK(hipStreamSynchronize(s));
    int r[2] = {0, 0};
    HIPCHK(hipMemcpy(r, ctx->dec_ctrl, sizeof(r), hipMemcpyDeviceToHost));
    ctx->overlap_ok = r[1] ? 1 : 0;
    if (!ctx->overlap_ok)
        fprintf(stderr, "[rnnt] kernels of two HIP streams do not overlap here (serialising profiler or shared hardware queue): "
                        "the resident decoder is replaced by launched evaluation batches\n");
    return RNNT_OK;
}

// wait for the decoder and check its error word; updates the evaluation counter
int finish_persistent_decoder(rnnt_ctx* ctx, hipStream_t s) {
    HIPCHK(hipMemcpyAsync(ctx->pinned + 12, ctx->dec_ctrl, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    ctx->greedy_steps += ctx->pinned[14];
    if (getenv("RNNT_COOP_DBG") && ctx->use_coop && ctx->n_streams <= 64) {
        long long t[16];
        (void)hipMemcpy(t, ctx->flow_buf + FLOW_WORDS, sizeof(t), hipMemcpyDeviceToHost);
        const double ev = t[11] > 0 ? (double)t[11] : 1.0;
        fprintf(stderr, "[flow] workgroup 0: %lld evals; us/eval: wait XA %.2f, decide+L %.2f, wait XH %.2f, J %.2f, wait XZ %.2f, O+send %.2f; polls/eval: XA %.2f XH %.2f XZ %.2f\n",
                t[11], t[0] / ev / 100.0, t[1] / ev / 100.0, t[2] / ev / 100.0, t[3] / ev / 100.0, t[4] / ev / 100.0, t[5] / ev / 100.0, t[8] / ev, t[9] / ev, t[10] / ev);
    }
    if (ctx->pinned[13] != 0) return fail(ctx, RNNT_ERR_STATE, "persistent decoder gave up (code %d: 1 = frame wait, 2 = barrier, 3 = idle bound)", ctx->pinned[13]);
    return RNNT_OK;
}

template <typename T>
int grow(rnnt_ctx* ctx, T** p, size_t* cap, size_t need) {
    if (need <= *cap) return RNNT_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    int rc = dmalloc(ctx, p, need);
    *cap = rc ? 0 : need;
    return rc;
}

}  // namespace

extern "C" {

int rnnt_abi_version(void) { return 2; }

const char* rnnt_last_error(const rnnt_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int rnnt_create(const rnnt_config* cfg, rnnt_ctx** out) {
    if (!cfg || !out) return RNNT_ERR_ARG;
    *out = nullptr;
    rnnt_ctx* ctx = new rnnt_ctx();
    ctx->cfg = *cfg;
    *out = ctx;   // returned even on failure so the caller can read rnnt_last_error, then destroy
    if (cfg->max_streams < 1 || cfg->max_chunk_frames < 7 || cfg->max_cache_frames < 1 || cfg->max_cache_frames > RNNT_PE_LEN ||
        cfg->max_enc_frames < 1 || cfg->max_tokens < 1 || cfg->vocab_size < 2 || cfg->blank_id < 0 || cfg->blank_id >= cfg->vocab_size ||
        cfg->n_steps < 1 || cfg->max_beam < 0 || cfg->max_beam > 64)
        return fail(ctx, RNNT_ERR_ARG, "rnnt_create: bad config");
    HIPCHK(hipSetDevice(cfg->device));
    if (const char* ng = getenv("RNNT_NO_GRAPH")) ctx->use_graphs = (ng[0] == '1') ? 0 : 1;
    if (const char* pe = getenv("RNNT_PERSISTENT")) ctx->use_persistent = (pe[0] == '0') ? 0 : 1;
    if (const char* ce = getenv("RNNT_COOP")) ctx->use_coop = (ce[0] == '0') ? 0 : 1;
    if (const char* ae = getenv("RNNT_ATTN_STREAM")) ctx->attn_stream = (ae[0] == '0') ? 0 : 1;
    if (const char* be = getenv("RNNT_BEAM_CHAIN")) ctx->use_beam_chain = (be[0] == '0') ? 0 : 1;
    if (const char* fe = getenv("RNNT_FUSE_AFTER_NORM")) ctx->fuse_after_norm = (fe[0] == '0') ? 0 : 1;
    if (const char* ge = getenv("RNNT_WF_GROUPS")) { const int g = atoi(ge); ctx->wf_groups = g < 1 ? 1 : (g > 4 ? 4 : g); }
    if (const char* me = getenv("RNNT_WF_MERGE")) { const int m = atoi(me); ctx->wf_merge = m < 1 ? 1 : (m > WF_MERGE_MAX ? WF_MERGE_MAX : m); }
    if (const char* se = getenv("RNNT_WF_SUB_ASYNC")) ctx->wf_sub_async = (se[0] == '0') ? 0 : 1;
    const int B = cfg->max_streams;
    ctx->tmax = sub_len(cfg->max_chunk_frames);
    ctx->t1max = sub1_len(cfg->max_chunk_frames);
    ctx->cap = RNNT_LORDER + WF_MERGE_MAX * ctx->tmax;   // several chunks of one layer may be in flight in one wavefront stage
    ctx->tcap = cfg->max_cache_frames;
    ctx->fcap = cfg->max_enc_frames;
    ctx->fstride = ctx->fcap + 1;   // +1 row: finished streams read one frame past the end
    ctx->vpad = (cfg->vocab_size + 31) / 32 * 32;
    const size_t M = (size_t)B * ctx->tmax;
    int rc;
#define ALLOC(p, n) if ((rc = dmalloc(ctx, &ctx->p, (size_t)(n)))) return rc
    ALLOC(y1, (size_t)B * ctx->t1max * RNNT_F1 * D);
    ALLOC(y2, M * RNNT_FSUB * D);
    ALLOC(x, M * D);
    ALLOC(hbuf, M * FF);
    ALLOC(qbuf, M * D);
    ALLOC(abuf, M * D);
    ALLOC(dbuf, M * D);
    ALLOC(kcache, (size_t)L * B * ctx->tcap * D);
    ALLOC(vcache, (size_t)L * B * ctx->tcap * D);
    ALLOC(gring, (size_t)L * B * ctx->cap * D);
    ALLOC(xring, (size_t)L * B * ctx->cap * D);
    ALLOC(encbuf, (size_t)B * ctx->fstride * D);
    ALLOC(encp, (size_t)B * ctx->fstride * D);
    ALLOC(h, (size_t)2 * B * D); ALLOC(c, (size_t)2 * B * D); ALLOC(sel, B); ALLOC(key, B);
    ALLOC(pred, (size_t)B * D); ALLOC(z, (size_t)B * D); ALLOC(logits, (size_t)B * ctx->vpad);
    ALLOC(tok, B); ALLOC(fidx, B); ALLOC(nsym, B); ALLOC(count, B); ALLOC(tokens, (size_t)B * cfg->max_tokens);
    ALLOC(n_active, 4); ALLOC(klen, B); ALLOC(dec_ctrl, 32);
    ALLOC(flow_buf, FLOW_WORDS + 16);
    if (cfg->max_beam > 0) {
        ctx->max_rows = B * cfg->max_beam;
        const size_t R = ctx->max_rows, NS = cfg->n_steps, KB = cfg->max_beam;
        ALLOC(pool[0], R * (NS + 1) * 512); ALLOC(pool[1], R * (NS + 1) * 512);
        ALLOC(bpred, R * D); ALLOC(bz, R * D); ALLOC(blogits, R * ctx->vpad);
        ALLOC(b_blank, R * NS); ALLOC(b_toplp, R * NS * KB); ALLOC(b_toptok, R * NS * KB);
        ALLOC(b_tok, R); ALLOC(b_frame, R); ALLOC(b_active, R); ALLOC(b_steps, R); ALLOC(b_srcrow, R); ALLOC(b_srcstep, R);
    }
    ctx->scratch_floats = (size_t)L * RNNT_H * ctx->tcap * 128;
    if (ctx->scratch_floats < (size_t)B * ctx->fstride * D) ctx->scratch_floats = (size_t)B * ctx->fstride * D;
    ALLOC(scratch, ctx->scratch_floats);
#undef ALLOC
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ctx->pinned), 64));
    HIPCHK(hipMemset(ctx->encbuf, 0, (size_t)B * ctx->fstride * D * sizeof(float)));
    HIPCHK(hipMemset(ctx->encp, 0, (size_t)B * ctx->fstride * D * sizeof(float)));
    return RNNT_OK;
}

void rnnt_destroy(rnnt_ctx* ctx) {
    if (!ctx) return;
    void* ptrs[] = {ctx->blob, ctx->egate, ctx->y1, ctx->y2, ctx->x, ctx->hbuf, ctx->qbuf, ctx->abuf, ctx->dbuf, ctx->kcache, ctx->vcache,
                    ctx->gring, ctx->xring, ctx->encbuf, ctx->encp, ctx->h, ctx->c, ctx->sel, ctx->key, ctx->dec_ctrl, ctx->flow_buf, ctx->pred, ctx->z, ctx->logits,
                    ctx->tok, ctx->fidx, ctx->nsym, ctx->count, ctx->tokens, ctx->n_active, ctx->klen, ctx->scratch,
                    ctx->pool[0], ctx->pool[1], ctx->bpred, ctx->bz, ctx->blogits, ctx->b_blank, ctx->b_toplp, ctx->b_toptok,
                    ctx->b_tok, ctx->b_frame, ctx->b_active, ctx->b_steps, ctx->b_srcrow, ctx->b_srcstep};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (int l = 0; l < L; ++l)
        if (ctx->lw[l].ptab) (void)hipFree(ctx->lw[l].ptab);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    for (hipEvent_t e : ctx->prof_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->wf_ev) (void)hipEventDestroy(e);
    if (ctx->dec_stream) (void)hipStreamDestroy(ctx->dec_stream);
    for (auto& g : ctx->dec_graphs) (void)hipGraphExecDestroy(g.exec);
    if (ctx->cap_stream) (void)hipStreamDestroy(ctx->cap_stream);
    for (hipStream_t x : ctx->grp_stream) if (x) (void)hipStreamDestroy(x);
    if (ctx->sub_stream) (void)hipStreamDestroy(ctx->sub_stream);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    for (float* q : {ctx->fb_dft, ctx->fb_mel, ctx->fb_pad, ctx->fb_spec, ctx->fb_pow}) if (q) (void)hipFree(q);
    void* wf[] = {ctx->wf_x, ctx->wf_h, ctx->wf_q, ctx->wf_a, ctx->wf_d, ctx->wf_y1, ctx->wf_y2, ctx->wf_starts, ctx->wf_gtab, ctx->wf_atab,
                  ctx->wf_dtab, ctx->wf_ltab};
    for (void* q : wf)
        if (q) (void)hipFree(q);
    delete ctx;
}

int rnnt_load_tensor(rnnt_ctx* ctx, const char* name, const float* host_data, int32_t ndim, const int64_t* dims) {
    if (!ctx || !name || ndim < 0 || ndim > 8) return fail(ctx, RNNT_ERR_ARG, "rnnt_load_tensor: bad argument");
    std::string n(name);
    if (n.find("num_batches_tracked") != std::string::npos) return RNNT_OK;   // int64 counter, unused in eval
    if (!host_data) return fail(ctx, RNNT_ERR_ARG, "rnnt_load_tensor(%s): null data", name);
    HostTensor t;
    size_t cnt = 1;
    for (int i = 0; i < ndim; ++i) {
        if (dims[i] < 0) return fail(ctx, RNNT_ERR_SHAPE, "rnnt_load_tensor(%s): negative dim", name);
        t.dims.push_back(dims[i]);
        cnt *= (size_t)dims[i];
    }
    t.data.assign(host_data, host_data + cnt);
    ctx->host[n] = std::move(t);
    ctx->finalized = false;
    return RNNT_OK;
}

int rnnt_finalize_weights(rnnt_ctx* ctx, int32_t numerics_mode, void* stream) {
    if (!ctx) return RNNT_ERR_ARG;
    if (numerics_mode != RNNT_NUMERICS_FP32) return fail(ctx, RNNT_ERR_ARG, "unsupported numerics mode %d", numerics_mode);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ctx->cfg.device));
    ctx->wf_key.clear();   // cached descriptor tables point into the weight blob
    const int V = ctx->cfg.vocab_size;
    std::vector<float> blob;
    std::vector<std::pair<const float**, size_t>> fix;   // pointer slots to patch with blob offsets
    auto need = [&](const std::string& name, std::initializer_list<int64_t> dims) -> const HostTensor* {
        const HostTensor* t = find(ctx, name);
        if (!t) { fail(ctx, RNNT_ERR_STATE, "missing tensor %s", name.c_str()); return nullptr; }
        if (t->dims != std::vector<int64_t>(dims)) { fail(ctx, RNNT_ERR_SHAPE, "tensor %s has the wrong shape", name.c_str()); return nullptr; }
        return t;
    };
    auto put = [&](const float** slot, const float* data, size_t n) {
        while (blob.size() % 4) blob.push_back(0.f);   // 16-byte alignment of every tensor
        fix.push_back({slot, blob.size()});
        blob.insert(blob.end(), data, data + n);
    };
    auto putv = [&](const float** slot, const std::vector<float>& v) { put(slot, v.data(), v.size()); };
#define NEED(var, name, ...) const HostTensor* var = need(name, {__VA_ARGS__}); if (!var) return ctx->err.find("missing") != std::string::npos ? RNNT_ERR_STATE : RNNT_ERR_SHAPE

    // --- subsampling -------------------------------------------------------------------------
    NEED(c1w, "encoder.embed.conv.0.weight", D, 1, 3, 3);
    NEED(c1b, "encoder.embed.conv.0.bias", D);
    NEED(c2w, "encoder.embed.conv.2.weight", D, D, 3, 3);
    NEED(c2b, "encoder.embed.conv.2.bias", D);
    NEED(eow, "encoder.embed.out.0.weight", D, D * RNNT_FSUB);
    NEED(eob, "encoder.embed.out.0.bias", D);
    NEED(pe, "encoder.embed.pos_enc.pe", 1, RNNT_PE_LEN, D);
    NEED(ang, "encoder.after_norm.weight", D);
    NEED(anb, "encoder.after_norm.bias", D);
    {
        std::vector<float> w1t(9 * D);   // [kh*3+kw][c]
        for (int c = 0; c < D; ++c)
            for (int k = 0; k < 9; ++k) w1t[k * D + c] = c1w->data[c * 9 + k];
        putv(&ctx->conv1_wt, w1t);
        put(&ctx->conv1_b, c1b->data.data(), D);
        std::vector<float> w2((size_t)D * 2304);   // [co][kh][kw][ci]
        for (int co = 0; co < D; ++co)
            for (int ci = 0; ci < D; ++ci)
                for (int k = 0; k < 9; ++k) w2[(size_t)co * 2304 + k * D + ci] = c2w->data[((size_t)co * D + ci) * 9 + k];
        putv(&ctx->conv2_w, w2);
        put(&ctx->conv2_b, c2b->data.data(), D);
        std::vector<float> ew((size_t)D * RNNT_FSUB * D);   // column c*19+f -> f*256+c
        for (int n = 0; n < D; ++n)
            for (int c = 0; c < D; ++c)
                for (int f = 0; f < RNNT_FSUB; ++f) ew[(size_t)n * RNNT_FSUB * D + f * D + c] = eow->data[(size_t)n * RNNT_FSUB * D + c * RNNT_FSUB + f];
        putv(&ctx->emb_w, ew);
        put(&ctx->emb_b, eob->data.data(), D);
        put(&ctx->pe, pe->data.data(), (size_t)RNNT_PE_LEN * D);
        put(&ctx->after_g, ang->data.data(), D);
        put(&ctx->after_b, anb->data.data(), D);
    }
    // --- encoder layers ----------------------------------------------------------------------
    std::vector<float> lncg((size_t)L * D), lncb((size_t)L * D), glu0((size_t)L * D);
    for (int l = 0; l < L; ++l) {
        LayerW& w = ctx->lw[l];
        const std::string p = "encoder.encoders." + std::to_string(l) + ".";
        auto vec = [&](const float** slot, const std::string& nm, std::initializer_list<int64_t> dims) -> int {
            const HostTensor* t = need(p + nm, dims);
            if (!t) return -1;
            put(slot, t->data.data(), t->data.size());
            return 0;
        };
#define VEC(slot, nm, ...) if (vec(&w.slot, nm, {__VA_ARGS__})) return RNNT_ERR_STATE
        VEC(ln_ffm_g, "norm_ff_macaron.weight", D); VEC(ln_ffm_b, "norm_ff_macaron.bias", D);
        VEC(w1m, "feed_forward_macaron.w_1.weight", FF, D); VEC(b1m, "feed_forward_macaron.w_1.bias", FF);
        VEC(w2m, "feed_forward_macaron.w_2.weight", D, FF); VEC(b2m, "feed_forward_macaron.w_2.bias", D);
        VEC(ln_mha_g, "norm_mha.weight", D); VEC(ln_mha_b, "norm_mha.bias", D);
        VEC(wq, "self_attn.linear_q.weight", D, D); VEC(bq, "self_attn.linear_q.bias", D);
        VEC(wk, "self_attn.linear_k.weight", D, D); VEC(bk, "self_attn.linear_k.bias", D);
        VEC(wv, "self_attn.linear_v.weight", D, D); VEC(bv, "self_attn.linear_v.bias", D);
        VEC(wo, "self_attn.linear_out.weight", D, D); VEC(bo, "self_attn.linear_out.bias", D);
        VEC(pu, "self_attn.pos_bias_u", RNNT_H, DK); VEC(pv, "self_attn.pos_bias_v", RNNT_H, DK);
        VEC(wpos, "self_attn.linear_pos.weight", D, D);
        VEC(ln_conv_g, "norm_conv.weight", D); VEC(ln_conv_b, "norm_conv.bias", D);
        VEC(pw2, "conv_module.pointwise_conv2.weight", D, D, 1); VEC(bpw2, "conv_module.pointwise_conv2.bias", D);
        VEC(bdw, "conv_module.depthwise_conv.bias", D);
        VEC(ln_ff_g, "norm_ff.weight", D); VEC(ln_ff_b, "norm_ff.bias", D);
        VEC(w1, "feed_forward.w_1.weight", FF, D); VEC(b1, "feed_forward.w_1.bias", FF);
        VEC(w2, "feed_forward.w_2.weight", D, FF); VEC(b2, "feed_forward.w_2.bias", D);
        VEC(ln_fin_g, "norm_final.weight", D); VEC(ln_fin_b, "norm_final.bias", D);
#undef VEC
        NEED(p1w, p + "conv_module.pointwise_conv1.weight", 2 * D, D, 1);
        NEED(p1b, p + "conv_module.pointwise_conv1.bias", 2 * D);
        NEED(dww, p + "conv_module.depthwise_conv.weight", D, 1, RNNT_KDW);
        NEED(bng, p + "conv_module.norm.weight", D);
        NEED(bnb, p + "conv_module.norm.bias", D);
        NEED(bnm, p + "conv_module.norm.running_mean", D);
        NEED(bnv, p + "conv_module.norm.running_var", D);
        // GLU pairs interleaved: row 2j = value row j, row 2j+1 = gate row 256+j (F.glu dim=1, convolution.py:139)
        std::vector<float> p1((size_t)2 * D * D), p1bi(2 * D);
        for (int j = 0; j < D; ++j) {
            memcpy(&p1[(size_t)(2 * j) * D], &p1w->data[(size_t)j * D], D * sizeof(float));
            memcpy(&p1[(size_t)(2 * j + 1) * D], &p1w->data[(size_t)(D + j) * D], D * sizeof(float));
            p1bi[2 * j] = p1b->data[j];
            p1bi[2 * j + 1] = p1b->data[D + j];
            glu0[(size_t)l * D + j] = p1b->data[j] * (1.0f / (1.0f + expf(-p1b->data[D + j])));
        }
        putv(&w.pw1, p1);
        putv(&w.bpw1, p1bi);
        std::vector<float> dwt((size_t)RNNT_KDW * D), bs(D), bt(D);
        for (int c = 0; c < D; ++c) {
            for (int k = 0; k < RNNT_KDW; ++k) dwt[(size_t)k * D + c] = dww->data[(size_t)c * RNNT_KDW + k];
            // BatchNorm1d eval: y = (x - mean) / sqrt(var + eps) * gamma + beta  ->  x*s + t
            const float inv = 1.0f / sqrtf(bnv->data[c] + 1e-5f);
            bs[c] = bng->data[c] * inv;
            bt[c] = bnb->data[c] - bnm->data[c] * bs[c];
        }
        putv(&w.wdw_t, dwt);
        putv(&w.bn_s, bs);
        putv(&w.bn_t, bt);
        const HostTensor* g = find(ctx, p + "norm_conv.weight");
        const HostTensor* b = find(ctx, p + "norm_conv.bias");
        memcpy(&lncg[(size_t)l * D], g->data.data(), D * sizeof(float));
        memcpy(&lncb[(size_t)l * D], b->data.data(), D * sizeof(float));
    }
    putv(&ctx->ln_conv_g_all, lncg);
    putv(&ctx->ln_conv_b_all, lncb);
    putv(&ctx->glu0, glu0);
    // --- predictor / joint ---------------------------------------------------------------------
    NEED(emb, "predictor.embed.weight", V, D);
    NEED(wih, "predictor.rnn.weight_ih_l0", 4 * D, D);
    NEED(whh, "predictor.rnn.weight_hh_l0", 4 * D, D);
    NEED(bih, "predictor.rnn.bias_ih_l0", 4 * D);
    NEED(bhh, "predictor.rnn.bias_hh_l0", 4 * D);
    NEED(wpr, "predictor.projection.weight", D, D);
    NEED(bpr, "predictor.projection.bias", D);
    NEED(wen, "joint.enc_ffn.weight", D, D);
    NEED(ben, "joint.enc_ffn.bias", D);
    NEED(wpf, "joint.pred_ffn.weight", D, D);
    NEED(bpf, "joint.pred_ffn.bias", D);
    NEED(wou, "joint.ffn_out.weight", V, D);
    NEED(bou, "joint.ffn_out.bias", V);
    {
        // gate rows interleaved: row 4j+g = torch row g*256+j (gate order i,f,g,o)
        std::vector<float> hh((size_t)4 * D * D), ih((size_t)4 * D * D), bb(4 * D);
        for (int j = 0; j < D; ++j)
            for (int g = 0; g < 4; ++g) {
                memcpy(&hh[(size_t)(4 * j + g) * D], &whh->data[(size_t)(g * D + j) * D], D * sizeof(float));
                memcpy(&ih[(size_t)(4 * j + g) * D], &wih->data[(size_t)(g * D + j) * D], D * sizeof(float));
                bb[4 * j + g] = bih->data[g * D + j] + bhh->data[g * D + j];
            }
        putv(&ctx->whh_il, hh);
        putv(&ctx->wih_il, ih);
        putv(&ctx->b_lstm_il, bb);
        put(&ctx->pred_embed, emb->data.data(), emb->data.size());
        put(&ctx->wpr, wpr->data.data(), wpr->data.size()); put(&ctx->bpr, bpr->data.data(), D);
        put(&ctx->wenc, wen->data.data(), wen->data.size()); put(&ctx->benc, ben->data.data(), D);
        put(&ctx->wpf, wpf->data.data(), wpf->data.size()); put(&ctx->bpf, bpf->data.data(), D);
        put(&ctx->wout, wou->data.data(), wou->data.size()); put(&ctx->bout, bou->data.data(), V);
        // greedy decode only needs pred_ffn(projection(h)): fold the two Linears (joint.py:54, predictor.py:205) into
        // W_c = W_pf * W_pr, b_c = W_pf * b_pr + b_pf (accumulated in double, stored in float32)
        std::vector<float> wc((size_t)D * D), bc(D);
        for (int n = 0; n < D; ++n) {
            for (int k = 0; k < D; ++k) {
                double acc = 0.0;
                for (int j = 0; j < D; ++j) acc += (double)wpf->data[(size_t)n * D + j] * (double)wpr->data[(size_t)j * D + k];
                wc[(size_t)n * D + k] = (float)acc;
            }
            double acc = bpf->data[n];
            for (int j = 0; j < D; ++j) acc += (double)wpf->data[(size_t)n * D + j] * (double)bpr->data[j];
            bc[n] = (float)acc;
        }
        putv(&ctx->wjc, wc);
        putv(&ctx->bjc, bc);
        const HostTensor* cw = find(ctx, "ctc_head.ctc_lo.weight");
        const HostTensor* cb = find(ctx, "ctc_head.ctc_lo.bias");
        ctx->wctc = ctx->bctc = nullptr;
        if (cw && cb && cw->dims == std::vector<int64_t>{V, D} && cb->dims == std::vector<int64_t>{V}) {
            put(&ctx->wctc, cw->data.data(), cw->data.size());
            put(&ctx->bctc, cb->data.data(), V);
        }
    }
#undef NEED
    // upload
    if (ctx->blob) { (void)hipFree(ctx->blob); ctx->blob = nullptr; }
    int rc;
    if ((rc = dmalloc(ctx, &ctx->blob, blob.size()))) return rc;
    ctx->blob_floats = blob.size();
    HIPCHK(hipMemcpyAsync(ctx->blob, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice, s));
    for (auto& f : fix) *f.first = ctx->blob + f.second;
    // derived tables on the device: P_l = pe * W_pos^T (attention.py:396, batch-invariant), input-gate
    // table E = embed * W_ih^T + b_ih + b_hh (predictor.py:200,204)
    for (int l = 0; l < L; ++l) {
        if (!ctx->lw[l].ptab && (rc = dmalloc(ctx, &ctx->lw[l].ptab, (size_t)RNNT_PE_LEN * D))) return rc;
        GemmP g = plain_gemm(ctx->pe, D, ctx->lw[l].wpos, D, nullptr, ctx->lw[l].ptab, D, RNNT_PE_LEN, D, D);
        if ((rc = launch_gemm(ctx, s, 0, &g, 1))) return rc;
    }
    if (!ctx->egate && (rc = dmalloc(ctx, &ctx->egate, (size_t)V * 4 * D))) return rc;
    {
        GemmP g = plain_gemm(ctx->pred_embed, D, ctx->wih_il, D, ctx->b_lstm_il, ctx->egate, 4 * D, V, 4 * D, D);
        if ((rc = launch_gemm(ctx, s, 0, &g, 1))) return rc;
    }
    HIPCHK(hipStreamSynchronize(s));
    ctx->numerics = numerics_mode;
    ctx->finalized = true;
    return RNNT_OK;
}

int rnnt_streams_reset(rnnt_ctx* ctx, int32_t n_streams, void* stream) {
    if (!ctx) return RNNT_ERR_ARG;
    if (!ctx->finalized) return fail(ctx, RNNT_ERR_STATE, "weights not finalized");
    if (n_streams < 1 || n_streams > ctx->cfg.max_streams) return fail(ctx, RNNT_ERR_ARG, "n_streams %d out of range", n_streams);
    hipStream_t s = (hipStream_t)stream;
    const int B = ctx->cfg.max_streams;
    ctx->n_streams = n_streams;
    ctx->cache_len = 0; ctx->kv_start = 0; ctx->conv_pos = 0;
    ctx->frames_buffered = 0; ctx->frames_decoded = 0;
    ctx->launches = 0; ctx->greedy_steps = 0;
    hipLaunchKernelGGL(conv_ring_init, dim3(grid_for((long long)L * B * ctx->cap * D)), dim3(256), 0, s, ctx->gring, ctx->xring, ctx->glu0, B, ctx->cap);
    LAUNCHCHK("conv_ring_init");
    HIPCHK(hipMemsetAsync(ctx->h, 0, (size_t)2 * B * D * sizeof(float), s));
    HIPCHK(hipMemsetAsync(ctx->c, 0, (size_t)2 * B * D * sizeof(float), s));
    HIPCHK(hipMemsetAsync(ctx->sel, 0, B * sizeof(int), s));
    HIPCHK(hipMemsetAsync(ctx->key, 0, B * sizeof(unsigned long long), s));
    HIPCHK(hipMemsetAsync(ctx->fidx, 0, B * sizeof(int), s));
    HIPCHK(hipMemsetAsync(ctx->nsym, 0, B * sizeof(int), s));
    HIPCHK(hipMemsetAsync(ctx->count, 0, B * sizeof(int), s));
    HIPCHK(hipMemsetAsync(ctx->n_active, 0, 4 * sizeof(int), s));
    hipLaunchKernelGGL(fill_i32, dim3(1), dim3(256), 0, s, ctx->tok, ctx->cfg.blank_id, (long long)B);
    LAUNCHCHK("fill_i32");
    if (ctx->max_rows > 0) {   // one empty hypothesis per stream with the zero LSTM state (online_rnnt_model.py:407-415)
        ctx->beams.assign(n_streams, std::vector<rnnt_ctx::Hyp>(1, rnnt_ctx::Hyp{{}, 0.0}));
        ctx->pool_cur = 0;
        HIPCHK(hipMemsetAsync(ctx->pool[0], 0, (size_t)ctx->max_rows * (ctx->cfg.n_steps + 1) * 512 * sizeof(float), s));
    }
    return RNNT_OK;
}

int rnnt_encoder_chunk(rnnt_ctx* ctx, const float* fbank_dev, int32_t T, int32_t offset, int32_t required_cache_size,
                       int32_t* frames_out, void* stream) {
    if (!ctx || !fbank_dev) return fail(ctx, RNNT_ERR_ARG, "rnnt_encoder_chunk: null argument");
    if (!ctx->finalized || ctx->n_streams < 1) return fail(ctx, RNNT_ERR_STATE, "rnnt_encoder_chunk: no weights / no streams");
    if (T < 7 || T > ctx->cfg.max_chunk_frames) return fail(ctx, RNNT_ERR_SHAPE, "chunk of %d frames outside [7, %d]", T, ctx->cfg.max_chunk_frames);
    hipStream_t s = (hipStream_t)stream;
    const int B = ctx->n_streams;
    const int tq = sub_len(T);
    const int T2 = ctx->cache_len + tq;                 // attention_key_size (encoder.py:256)
    const int pos_start = offset - ctx->cache_len;      // encoder.py:257
    if (pos_start < 0 || pos_start + T2 > RNNT_PE_LEN)
        return fail(ctx, RNNT_ERR_SHAPE, "positional window [%d, %d) outside the 5000-entry table", pos_start, pos_start + T2);
    if (ctx->kv_start + T2 > ctx->tcap) return fail(ctx, RNNT_ERR_SHAPE, "K/V cache capacity %d exceeded", ctx->tcap);
    if (ctx->frames_buffered + tq > ctx->fcap) return fail(ctx, RNNT_ERR_SHAPE, "encoder-frame buffer capacity %d exceeded", ctx->fcap);
    int rc;
    if ((rc = run_subsample(ctx, s, fbank_dev, B, T, T, nullptr, 1, ctx->y1, ctx->y2, ctx->x))) return rc;
    for (int l = 0; l < L; ++l)
        if ((rc = run_layer(ctx, s, l, B, tq, T2, ctx->kv_start, pos_start, ctx->conv_pos, nullptr))) return rc;
    // after_norm straight into the frame buffer, then the joint's encoder projection for the new frames
    if ((rc = launch_ln(ctx, s, LnP{ctx->x, ctx->after_g, ctx->after_b, ctx->encbuf, B * tq, tq, ctx->frames_buffered,
                                    (long long)ctx->fstride * D, (long long)D}))) return rc;
    {
        GemmP g = plain_gemm(ctx->encbuf + (size_t)ctx->frames_buffered * D, D, ctx->wenc, D, ctx->benc, ctx->encp, D, B * tq, D, D);
        g.a_n1 = tq; g.a_n2 = tq; g.a_s0 = (long long)ctx->fstride * D; g.a_s1 = 0; g.a_s2 = D;
        g.c_n = tq; g.c_s0 = (long long)ctx->fstride * D; g.c_r0 = ctx->frames_buffered; g.c_mod = BIG; g.c_s1 = D;
        if ((rc = launch_gemm(ctx, s, 0, &g, 1, TAG_ENC_PROJ))) return rc;
    }
    // cache bookkeeping (encoder.py:259-264,288)
    int next_start;
    if (required_cache_size < 0) next_start = 0;
    else if (required_cache_size == 0) next_start = T2;
    else next_start = T2 - required_cache_size > 0 ? T2 - required_cache_size : 0;
    ctx->kv_start += next_start;
    ctx->cache_len = T2 - next_start;
    if (ctx->cache_len == 0) ctx->kv_start = 0;
    ctx->conv_pos += tq;
    ctx->frames_buffered += tq;
    if (frames_out) *frames_out = tq;
    return RNNT_OK;
}

// Whole-utterance encoder: every chunk of every stream, same results as n_chunks calls of rnnt_encoder_chunk.
// (a) subsampling batched over runs of equal-length chunks; (b) WAVEFRONT over (chunk c, layer l): stage s runs
// all pairs with c + l = s as ONE grouped launch per kernel type (layer l of chunk c needs only layer l-1 of
// chunk c and layer l's K/V + conv caches after chunk c-1), so the dependent-launch chain is
// (n_chunks + 11) stages instead of 12 * n_chunks; (c) after_norm + joint.enc_ffn for all new frames at once.
int rnnt_encoder_chunks(rnnt_ctx* ctx, const float* fbank_dev, int32_t total_frames, int32_t n_chunks, const int32_t* chunk_start,
                        const int32_t* chunk_len, const int32_t* offsets, const int32_t* required, int32_t greedy, int32_t* frames_out,
                        void* stream) {
    if (!ctx || !fbank_dev || !chunk_start || !chunk_len || !offsets || !required || n_chunks < 1)
        return fail(ctx, RNNT_ERR_ARG, "rnnt_encoder_chunks: bad argument");
    if (!ctx->finalized || ctx->n_streams < 1) return fail(ctx, RNNT_ERR_STATE, "rnnt_encoder_chunks: no weights / no streams");
    hipStream_t s = (hipStream_t)stream;
    const int B = ctx->n_streams, C = n_chunks;
    const int Mmax = ctx->cfg.max_streams * ctx->tmax;
    int rc;
    // ---- static schedule: simulate the reference's per-chunk bookkeeping (encoder.py:254-264) -------------
    struct CI { int len, tq, T2, kv_row0, pos_start, ring_pos, fpos; size_t xoff; };
    std::vector<CI> ci(C);
    int cache_len = ctx->cache_len, kv_start = ctx->kv_start, conv_pos = ctx->conv_pos, fb = ctx->frames_buffered;
    size_t xrows = 0;
    for (int c = 0; c < C; ++c) {
        CI& k = ci[c];
        k.len = chunk_len[c];
        if (k.len < 7 || k.len > ctx->cfg.max_chunk_frames || chunk_start[c] < 0 || chunk_start[c] + k.len > total_frames)
            return fail(ctx, RNNT_ERR_SHAPE, "chunk %d [%d,+%d) invalid for %d frames / max_chunk_frames %d", c, chunk_start[c], k.len,
                        total_frames, ctx->cfg.max_chunk_frames);
        k.tq = sub_len(k.len);
        k.T2 = cache_len + k.tq;
        k.pos_start = offsets[c] - cache_len;
        k.kv_row0 = kv_start;
        k.ring_pos = conv_pos;
        k.fpos = fb;
        k.xoff = xrows;
        if (k.pos_start < 0 || k.pos_start + k.T2 > RNNT_PE_LEN) return fail(ctx, RNNT_ERR_SHAPE, "chunk %d: positional window outside the table", c);
        if (kv_start + k.T2 > ctx->tcap) return fail(ctx, RNNT_ERR_SHAPE, "K/V cache capacity %d exceeded", ctx->tcap);
        if (fb + k.tq > ctx->fcap) return fail(ctx, RNNT_ERR_SHAPE, "encoder-frame buffer capacity %d exceeded", ctx->fcap);
        int next_start;
        if (required[c] < 0) next_start = 0;
        else if (required[c] == 0) next_start = k.T2;
        else next_start = k.T2 - required[c] > 0 ? k.T2 - required[c] : 0;
        kv_start += next_start;
        cache_len = k.T2 - next_start;
        if (cache_len == 0) kv_start = 0;
        conv_pos += k.tq;
        fb += k.tq;
        xrows += (size_t)B * k.tq;
    }
    // ---- buffers ------------------------------------------------------------------------------------------------
    if (!ctx->wf_x) {
        const size_t Bm = ctx->cfg.max_streams;
        size_t per_chunk = Bm * ctx->t1max * RNNT_F1 * D * sizeof(float);
        ctx->wf_slab = (int)((192ull << 20) / per_chunk);
        if (ctx->wf_slab < 1) ctx->wf_slab = 1;
        if (ctx->wf_slab > 16) ctx->wf_slab = 16;
        if ((rc = dmalloc(ctx, &ctx->wf_x, Bm * ctx->fcap * D))) return rc;
        if ((rc = dmalloc(ctx, &ctx->wf_h, (size_t)L * WF_MERGE_MAX * Mmax * FF))) return rc;
        if ((rc = dmalloc(ctx, &ctx->wf_q, (size_t)L * WF_MERGE_MAX * Mmax * D))) return rc;
        if ((rc = dmalloc(ctx, &ctx->wf_a, (size_t)L * WF_MERGE_MAX * Mmax * D))) return rc;
        if ((rc = dmalloc(ctx, &ctx->wf_d, (size_t)L * WF_MERGE_MAX * Mmax * D))) return rc;
        if ((rc = dmalloc(ctx, &ctx->wf_y1, (size_t)ctx->wf_slab * Bm * ctx->t1max * RNNT_F1 * D))) return rc;
        if ((rc = dmalloc(ctx, &ctx->wf_y2, (size_t)ctx->wf_slab * Mmax * RNNT_FSUB * D))) return rc;
    }
    if ((rc = grow(ctx, &ctx->wf_starts, &ctx->wf_starts_cap, (size_t)C))) return rc;
    // ---- streams ---------------------------------------------------------------------------------------------------
    // The layers are split into G groups of consecutive layers, one HIP stream each (group 0 = the caller's stream):
    // within a group the stages are stream-ordered; group g+1's stage st+1 waits for group g's stage st (layer lo(g+1)
    // of chunk c needs layer lo(g+1)-1 of the same chunk, nothing else crosses a group).  Kernels of different groups
    // run concurrently, so one group's MFMA phases fill the other's prologue / epilogue / launch ramps, and the
    // subsampling (big-M conv2, MFMA-bound) runs on its own stream under the latency-bound stages instead of in front
    // of them -- the first frames reach the decoder ~4 ms earlier.
    const int G = ctx->wf_groups;
    hipStream_t gs[4] = {s, s, s, s};
    for (int g = 1; g < G; ++g) {
        if (!ctx->grp_stream[g]) HIPCHK(hipStreamCreateWithFlags(&ctx->grp_stream[g], hipStreamNonBlocking));
        gs[g] = ctx->grp_stream[g];
    }
    if (ctx->wf_sub_async && !ctx->sub_stream) HIPCHK(hipStreamCreateWithFlags(&ctx->sub_stream, hipStreamNonBlocking));
    hipStream_t ss = ctx->wf_sub_async ? ctx->sub_stream : s;
    size_t ev_next = 0;
    auto new_event = [&](hipEvent_t* out) -> int {
        if (ev_next == ctx->ev_pool.size()) {
            hipEvent_t e;
            HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->ev_pool.push_back(e);
        }
        *out = ctx->ev_pool[ev_next++];
        return RNNT_OK;
    };
    hipEvent_t e_in;
    if ((rc = new_event(&e_in))) return rc;
    HIPCHK(hipEventRecord(e_in, s));                       // everything the caller enqueued before this call
    for (int g = 0; g < G; ++g)
        if (gs[g] != s) HIPCHK(hipStreamWaitEvent(gs[g], e_in, 0));
    if (ss != s) HIPCHK(hipStreamWaitEvent(ss, e_in, 0));
    // ---- (a) subsampling, runs of equal-length chunks in slabs ------------------------------------------
    HIPCHK(hipMemcpyAsync(ctx->wf_starts, chunk_start, C * sizeof(int), hipMemcpyHostToDevice, ss));
    std::vector<hipEvent_t> slab_ev(C, nullptr);           // set on the first chunk of every slab
    for (int c0 = 0; c0 < C;) {
        int c1 = c0 + 1;
        const int slab = (c0 == 0 && ss != s) ? (ctx->wf_slab < 4 ? ctx->wf_slab : 4) : ctx->wf_slab;   // a short first slab: stage 0 starts sooner
        while (c1 < C && ci[c1].len == ci[c0].len && c1 - c0 < slab) ++c1;
        if ((rc = run_subsample(ctx, ss, fbank_dev, B, total_frames, ci[c0].len, ctx->wf_starts + c0, c1 - c0, ctx->wf_y1, ctx->wf_y2,
                                ctx->wf_x + ci[c0].xoff * D)))
            return rc;
        if (ss != s) {
            if ((rc = new_event(&slab_ev[c0]))) return rc;
            HIPCHK(hipEventRecord(slab_ev[c0], ss));
        }
        c0 = c1;
    }
    using Launch = rnnt_ctx::WfLaunch;
    const auto t_tab0 = std::chrono::steady_clock::now();
    const int KM = ctx->wf_merge;
    std::vector<int> key = {B, C, KM, total_frames, ctx->cache_len, ctx->kv_start, ctx->conv_pos, ctx->frames_buffered};
    key.insert(key.end(), chunk_start, chunk_start + C); key.insert(key.end(), chunk_len, chunk_len + C);
    key.insert(key.end(), offsets, offsets + C); key.insert(key.end(), required, required + C);
    std::vector<rnnt_ctx::WfLaunch>& seq = ctx->wf_seq;
    std::vector<std::array<int, 13>>& lstart = ctx->wf_lstart;
    std::vector<int>& sc_first = ctx->wf_sc_first;
    if (key != ctx->wf_key) {                                // same plan from the same state: the device tables are still valid
    seq.clear(); sc_first.clear();
    ctx->wf_key.clear();
    // ---- (b) wavefront tables ---------------------------------------------------------------------------------
    std::vector<GemmP> gt; std::vector<AttnP> at; std::vector<DwP> dt; std::vector<LnP> lt;
    gt.reserve((size_t)C * L * 12); at.reserve((size_t)C * L); dt.reserve((size_t)C * L); lt.reserve((size_t)C * (L + 1));
    // Stage of pair (chunk c, layer l) = c / KM + l: KM consecutive chunks of a layer share a stage.  Everything but
    // attention and the depthwise conv is per-frame, and those two only need the SAME layer's K/V rows / ring rows of the
    // earlier chunks, which the stage's QKV / pointwise_conv1 launch has written before the attention / depthwise launch
    // starts.  Fewer, fatter stages: the fixed cost of a launch (ramp, prologue, epilogue) is paid per 2 chunks.
    // A chunk joins its predecessor's stage only if the K/V rows it appends lie behind everything the predecessor reads or
    // writes (the reference re-bases the cache at row 0 after the first chunk, whose K/V are dropped: chunk 1 would overwrite
    // chunk 0's rows inside one launch).
    for (int c = 0, cnt = 0; c < C; ++c) {
        const bool behind = c > 0 && ci[c].kv_row0 + ci[c].T2 - ci[c].tq >= ci[c - 1].kv_row0 + ci[c - 1].T2;
        if (c == 0 || cnt == KM || !behind) { sc_first.push_back(c); cnt = 0; }
        ++cnt;
    }
    sc_first.push_back(C);
    const int NSC = (int)sc_first.size() - 1;                // super-chunks
    const int NS = NSC + L - 1;                              // stages
    std::vector<LayerDescs> cur;
    lstart.assign((size_t)NS, std::array<int, 13>());       // per stage: first pair index of every layer (+ total)
    for (int st = 0; st < NS; ++st) {
        cur.clear();
        int maxM = 0, maxtq = 0, maxT2 = 0;
        for (int l = 0; l < L; ++l) {
            lstart[st][l] = (int)cur.size();
            const int sc = st - l;
            if (sc < 0 || sc >= NSC) continue;
            for (int c = sc_first[sc]; c < sc_first[sc + 1]; ++c) {
                const int j = c - sc_first[sc];
                LayerDescs d;
                const size_t slot = (size_t)l * WF_MERGE_MAX + j;
                LayerBufs bf{ctx->wf_x + ci[c].xoff * D, ctx->wf_h + slot * Mmax * FF, ctx->wf_q + slot * Mmax * D,
                             ctx->wf_a + slot * Mmax * D, ctx->wf_d + slot * Mmax * D};
                if ((rc = build_layer(ctx, l, B, ci[c].tq, ci[c].T2, ci[c].kv_row0, ci[c].pos_start, ci[c].ring_pos, nullptr, bf, d))) return rc;
                cur.push_back(d);
                if (B * ci[c].tq > maxM) maxM = B * ci[c].tq;
                if (ci[c].tq > maxtq) maxtq = ci[c].tq;
                if (ci[c].T2 > maxT2) maxT2 = ci[c].T2;
            }
        }
        lstart[st][L] = (int)cur.size();
        const int n = (int)cur.size();
        auto add_g = [&](int type, GemmP LayerDescs::*f) -> int {
            seq.push_back({type, (int)gt.size(), n, maxM, 0});
            for (auto& d : cur) { GemmP g = d.*f; int r2 = prepare_gemm(ctx, g); if (r2) return r2; gt.push_back(g); }
            return 0;
        };
        if ((rc = add_g(0, &LayerDescs::ffn1m))) return rc;
        if ((rc = add_g(1, &LayerDescs::ffn2m))) return rc;
        seq.push_back({2, (int)gt.size(), 3 * n, maxM, 0});
        for (auto& d : cur)
            for (int i = 0; i < 3; ++i) { GemmP g = d.qkv[i]; if ((rc = prepare_gemm(ctx, g))) return rc; gt.push_back(g); }
        seq.push_back({10, (int)at.size(), n, maxtq, maxT2});
        for (auto& d : cur) at.push_back(d.attn);
        if ((rc = add_g(3, &LayerDescs::out))) return rc;
        if ((rc = add_g(4, &LayerDescs::pw1))) return rc;
        seq.push_back({11, (int)dt.size(), n, maxM, 0});
        for (auto& d : cur) dt.push_back(d.dw);
        if ((rc = add_g(5, &LayerDescs::pw2))) return rc;
        if ((rc = add_g(6, &LayerDescs::ffn1))) return rc;
        if ((rc = add_g(7, &LayerDescs::ffn2))) return rc;
        seq.push_back({12, (int)lt.size(), n, maxM, 0});
        for (auto& d : cur) lt.push_back(d.lnf);
    }
    if ((rc = grow(ctx, &ctx->wf_gtab, &ctx->wf_gcap, gt.size()))) return rc;
    if ((rc = grow(ctx, &ctx->wf_atab, &ctx->wf_acap, at.size()))) return rc;
    if ((rc = grow(ctx, &ctx->wf_dtab, &ctx->wf_dcap, dt.size()))) return rc;
    if ((rc = grow(ctx, &ctx->wf_ltab, &ctx->wf_lcap, lt.size()))) return rc;
    HIPCHK(hipMemcpyAsync(ctx->wf_gtab, gt.data(), gt.size() * sizeof(GemmP), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(ctx->wf_atab, at.data(), at.size() * sizeof(AttnP), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(ctx->wf_dtab, dt.data(), dt.size() * sizeof(DwP), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(ctx->wf_ltab, lt.data(), lt.size() * sizeof(LnP), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));   // the host vectors die at return; tables are small (a few MB)
    if (getenv("RNNT_TIMING")) fprintf(stderr, "[rnnt timing] descriptor tables: %.3f ms on the host (%zu GEMM descriptors)\n",
                                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_tab0).count(), gt.size());
    ctx->wf_key = key;
    }
    const int NSC = (int)sc_first.size() - 1, NS = NSC + L - 1;   // super-chunks, stages
    static const int gN[8] = {FF, D, D, D, 2 * D, D, FF, D};
    static const int gK[8] = {D, FF, D, D, D, D, D, FF};
    static const int gTag[8] = {TAG_FFN1, TAG_FFN2, TAG_QKV, TAG_ATTN_OUT, TAG_PW1, TAG_PW2, TAG_FFN1, TAG_FFN2};
    {                                                       // the other streams read the tables copied on s
        hipEvent_t e_tab;
        if ((rc = new_event(&e_tab))) return rc;
        HIPCHK(hipEventRecord(e_tab, s));
        for (int g = 0; g < G; ++g)
            if (gs[g] != s) HIPCHK(hipStreamWaitEvent(gs[g], e_tab, 0));
    }
    hipStream_t sl = gs[G - 1];                             // the stream the last layer runs on
    // decode stream + events (greedy != 0): chunk c's frames are decodable once its layer-11 stage, after_norm and
    // joint.enc_ffn projection are done; the decoder runs on ctx->dec_stream concurrently with later stages.
    hipStream_t s2 = s;
    bool resident = false;
    if (greedy) {
        if (!ctx->dec_stream) {   // decode = the latency-critical dependent chain: highest stream priority (own hardware queue)
            int lo = 0, hi = 0;
            HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            HIPCHK(hipStreamCreateWithPriority(&ctx->dec_stream, hipStreamNonBlocking, hi));
        }
        s2 = ctx->dec_stream;
        while ((int)ctx->wf_ev.size() < C + 1) {
            hipEvent_t e;
            HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->wf_ev.push_back(e);
        }
        ctx->pinned[8] = 0;
        if (ctx->use_persistent && ctx->overlap_ok < 0) {   // the resident decoder must not block ANY stream the encoder uses
            if ((rc = probe_overlap(ctx, s, s2))) return rc;
            for (int g = 0; g < G && ctx->overlap_ok == 1; ++g)
                if (gs[g] != s && (rc = probe_overlap(ctx, gs[g], s2))) return rc;
            if (ss != s && ctx->overlap_ok == 1 && (rc = probe_overlap(ctx, ss, s2))) return rc;
        }
        resident = ctx->use_persistent && ctx->overlap_ok == 1;
        if (resident && (rc = init_decoder_ctrl(ctx, s, ctx->frames_buffered))) return rc;
        HIPCHK(hipEventRecord(ctx->wf_ev[C], s));          // everything enqueued before this call (reset, earlier decode)
        HIPCHK(hipStreamWaitEvent(s2, ctx->wf_ev[C], 0));
        if (sl != s) HIPCHK(hipStreamWaitEvent(sl, ctx->wf_ev[C], 0));   // publish_frames comes after the control block's init
        if (resident && (rc = launch_persistent_decoder(ctx, s2, fb))) return rc;
    }
    int dec_steps = 0;
    const int fb0 = ctx->frames_buffered;
    static const bool timing = getenv("RNNT_TIMING") != nullptr;
    double t_enc = 0, t_dec = 0;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tl = now();
    std::vector<hipEvent_t> grp_ev((size_t)G * (NS + 1), nullptr);   // [g][st]: group g finished its part of stage st
    for (int st = 0; st < NS; ++st) {
        const std::array<int, 13>& ls = lstart[st];
        for (int g = 0; g < G; ++g) {
            const int lo = g * L / G, hi = (g + 1) * L / G;                                  // layers [lo, hi) of this group
            const int p0 = ls[lo], pn = ls[hi] - ls[lo];                                     // pairs [p0, p0 + pn) of the stage
            if (pn <= 0) continue;
            hipStream_t x = gs[g];
            if (lo == 0 && ls[1] > ls[0])                                                    // layer 0: its chunks' slabs are subsampled
                for (int c = sc_first[st]; c < sc_first[st + 1]; ++c)
                    if (slab_ev[c]) HIPCHK(hipStreamWaitEvent(x, slab_ev[c], 0));
            if (g > 0 && ls[lo + 1] > ls[lo] && st > 0 && grp_ev[(size_t)(g - 1) * (NS + 1) + st - 1])
                HIPCHK(hipStreamWaitEvent(x, grp_ev[(size_t)(g - 1) * (NS + 1) + st - 1], 0));
            for (int j = 0; j < 11; ++j) {
                const Launch& q = seq[(size_t)st * 11 + j];
                if (q.type < 8) {
                    const int mult = q.type == 2 ? 3 : 1;
                    if ((rc = launch_gemm_tab(ctx, x, ctx->wf_gtab + q.off + mult * p0, mult * pn, q.maxM, gN[q.type], gK[q.type], gTag[q.type]))) return rc;
                } else if (q.type == 10) {
                    ProfScope prof(ctx, x, TAG_ATTN);
                    static const int attn_pair_major = getenv("RNNT_ATTN_PAIR_MAJOR") ? atoi(getenv("RNNT_ATTN_PAIR_MAJOR")) : 1;
                    if (attn_stream_ok(ctx, q.maxM, q.maxT2 > 0 ? q.maxT2 : 1)) {
                        const int cap = attn_t2cap(q.maxT2);
                        hipLaunchKernelGGL(rel_attention_stream_tab, dim3((B * RNNT_H + 7) / 8 * 8 * pn), dim3(256), attn_stream_lds(cap), x,
                                           ctx->wf_atab + q.off + p0, cap, pn, B * RNNT_H, attn_pair_major);
                        LAUNCHCHK("rel_attention_stream_tab");
                        continue;
                    }
                    const int nq = q.maxM <= 4 ? 1 : (q.maxM <= 8 ? 2 : 4);
                    dim3 grid(B * RNNT_H, (q.maxM + 4 * nq - 1) / (4 * nq), pn);
                    if (nq == 1) hipLaunchKernelGGL(rel_attention_tab<1>, grid, dim3(256), 0, x, ctx->wf_atab + q.off + p0);
                    else if (nq == 2) hipLaunchKernelGGL(rel_attention_tab<2>, grid, dim3(256), 0, x, ctx->wf_atab + q.off + p0);
                    else hipLaunchKernelGGL(rel_attention_tab<4>, grid, dim3(256), 0, x, ctx->wf_atab + q.off + p0);
                    LAUNCHCHK("rel_attention_tab");
                } else if (q.type == 11) {
                    ProfScope prof(ctx, x, TAG_DWCONV);
                    hipLaunchKernelGGL(dwconv_bn_silu_tab, dim3(grid_for((long long)q.maxM * D), 1, pn), dim3(256), 0, x, ctx->wf_dtab + q.off + p0);
                    LAUNCHCHK("dwconv_bn_silu_tab");
                } else {
                    hipLaunchKernelGGL(layer_norm_tab, dim3((q.maxM + 3) / 4, 1, pn), dim3(256), 0, x, ctx->wf_ltab + q.off + p0);
                    LAUNCHCHK("layer_norm_tab");
                }
            }
            if (g < G - 1 && ls[hi] > ls[hi - 1]) {         // the next group's first layer reads this group's last layer
                hipEvent_t e;
                if ((rc = new_event(&e))) return rc;
                HIPCHK(hipEventRecord(e, x));
                grp_ev[(size_t)g * (NS + 1) + st] = e;
            }
        }
        const int scl = st - (L - 1);   // super-chunk whose last block just ran
        if (scl < 0) continue;
        int stage_frames = 0, c_last = -1;
        for (int c = sc_first[scl]; c < sc_first[scl + 1]; ++c) {
        // (c) after_norm straight into the frame buffer + joint.enc_ffn projection of the chunk's frames
        if (greedy && resident && ctx->fuse_after_norm) {
            // greedy decode reads only enc_proj: after_norm goes into the projection's LayerNorm prologue and the
            // normalised frames are not materialised (rnnt_get_enc_frames is not defined after such a call)
            const int F = ci[c].tq;
            GemmP g = plain_gemm(ctx->wf_x + ci[c].xoff * D, D, ctx->wenc, D, ctx->benc, ctx->encp, D, B * F, D, D);
            g.ln_g = ctx->after_g; g.ln_b = ctx->after_b;
            g.c_n = F; g.c_s0 = (long long)ctx->fstride * D; g.c_r0 = ci[c].fpos; g.c_mod = BIG; g.c_s1 = D;
            if ((rc = launch_gemm(ctx, sl, 0, &g, 1, TAG_ENC_PROJ))) return rc;
        } else {
            if ((rc = launch_ln(ctx, sl, LnP{ctx->wf_x + ci[c].xoff * D, ctx->after_g, ctx->after_b, ctx->encbuf, B * ci[c].tq, ci[c].tq, ci[c].fpos,
                                             (long long)ctx->fstride * D, (long long)D}))) return rc;
            const int F = ci[c].tq;
            GemmP g = plain_gemm(ctx->encbuf + (size_t)ci[c].fpos * D, D, ctx->wenc, D, ctx->benc, ctx->encp, D, B * F, D, D);
            g.a_n1 = F; g.a_n2 = F; g.a_s0 = (long long)ctx->fstride * D; g.a_s1 = 0; g.a_s2 = D;
            g.c_n = F; g.c_s0 = (long long)ctx->fstride * D; g.c_r0 = ci[c].fpos; g.c_mod = BIG; g.c_s1 = D;
            if ((rc = launch_gemm(ctx, sl, 0, &g, 1, TAG_ENC_PROJ))) return rc;
        }
            stage_frames += ci[c].tq;
            c_last = c;
        }
        if (timing) { double t = now(); t_enc += t - tl; tl = t; }
        if (c_last < 0) continue;
        if (greedy && resident) {   // the resident decoder sees the stage's frames as soon as this lands
            hipLaunchKernelGGL(publish_frames, dim3(1), dim3(1), 0, sl, ctx->dec_ctrl, ci[c_last].fpos + ci[c_last].tq);
            LAUNCHCHK("publish_frames");
        } else if (greedy) {
            HIPCHK(hipEventRecord(ctx->wf_ev[c_last], sl));
            HIPCHK(hipStreamWaitEvent(s2, ctx->wf_ev[c_last], 0));
            // launched decode path: this stage's frames + a little slack, in hipGraph-captured batches
            static const int slack = getenv("RNNT_DEC_SLACK") ? atoi(getenv("RNNT_DEC_SLACK")) : 8;
            int budget = stage_frames + slack;
            budget = (budget + 3) / 4 * 4;   // few distinct graph sizes
            if ((rc = greedy_steps(ctx, s2, budget, ci[c_last].fpos + ci[c_last].tq))) return rc;
            dec_steps += budget;
            if (timing) { double t = now(); t_dec += t - tl; tl = t; }
        }
    }
    if (timing) fprintf(stderr, "[rnnt timing] host enqueue: encoder stages %.2f ms, decode batches %.2f ms\n", t_enc, t_dec);
    if (frames_out) *frames_out = fb - fb0;
    ctx->cache_len = cache_len; ctx->kv_start = kv_start; ctx->conv_pos = conv_pos; ctx->frames_buffered = fb;
    // ---- join: the caller's stream continues after every internal stream ---------------------------------------------
    for (int g = 0; g < G; ++g) {
        if (gs[g] == s) continue;
        hipEvent_t e;
        if ((rc = new_event(&e))) return rc;
        HIPCHK(hipEventRecord(e, gs[g]));
        HIPCHK(hipStreamWaitEvent(s, e, 0));
    }
    if (ss != s) {
        hipEvent_t e;
        if ((rc = new_event(&e))) return rc;
        HIPCHK(hipEventRecord(e, ss));
        HIPCHK(hipStreamWaitEvent(s, e, 0));
    }
    if (greedy && resident) {
        double t_e = 0;
        if (timing) { (void)hipStreamSynchronize(s); t_e = now(); }
        if ((rc = finish_persistent_decoder(ctx, s2))) return rc;      // synchronises the decode stream (=> encoder done too)
        if (timing) fprintf(stderr, "[rnnt timing] decoder finished %.3f ms after the encoder streams drained\n", now() - t_e);
        ctx->frames_decoded = fb;
        HIPCHK(hipEventRecord(ctx->wf_ev[C], s2));
        HIPCHK(hipStreamWaitEvent(s, ctx->wf_ev[C], 0));
    } else if (greedy) {
        if ((rc = greedy_drain(ctx, s2, fb, dec_steps))) return rc;   // synchronises the decode stream (=> encoder done too)
        ctx->frames_decoded = fb;
        HIPCHK(hipEventRecord(ctx->wf_ev[C], s2));                    // later work on the caller's stream sees the decode
        HIPCHK(hipStreamWaitEvent(s, ctx->wf_ev[C], 0));
    }
    return RNNT_OK;
}

int rnnt_greedy_decode(rnnt_ctx* ctx, void* stream) {
    if (!ctx) return RNNT_ERR_ARG;
    if (!ctx->finalized || ctx->n_streams < 1) return fail(ctx, RNNT_ERR_STATE, "rnnt_greedy_decode: no weights / no streams");
    hipStream_t s = (hipStream_t)stream;
    const int nf = ctx->frames_buffered;
    if (nf <= ctx->frames_decoded) return RNNT_OK;
    int rc;
    if (ctx->use_persistent) {
        if ((rc = init_decoder_ctrl(ctx, s, nf))) return rc;
        if ((rc = launch_persistent_decoder(ctx, s, nf))) return rc;
        if ((rc = finish_persistent_decoder(ctx, s))) return rc;
        ctx->frames_decoded = nf;
        return RNNT_OK;
    }
    const int first = nf - ctx->frames_decoded + 2;
    if ((rc = greedy_steps(ctx, s, first, nf))) return rc;
    if ((rc = greedy_drain(ctx, s, nf, first))) return rc;
    ctx->frames_decoded = nf;
    return RNNT_OK;
}

int rnnt_get_tokens(rnnt_ctx* ctx, int32_t* counts_host, int32_t* tokens_host, void* stream) {
    if (!ctx) return RNNT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (counts_host) HIPCHK(hipMemcpyAsync(counts_host, ctx->count, ctx->n_streams * sizeof(int), hipMemcpyDeviceToHost, s));
    if (tokens_host)
        HIPCHK(hipMemcpyAsync(tokens_host, ctx->tokens, (size_t)ctx->n_streams * ctx->cfg.max_tokens * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return RNNT_OK;
}

int rnnt_frames_consume(rnnt_ctx* ctx, void* stream) {
    if (!ctx) return RNNT_ERR_ARG;
    if (ctx->frames_decoded != ctx->frames_buffered) return fail(ctx, RNNT_ERR_STATE, "undecoded frames in the buffer");
    hipStream_t s = (hipStream_t)stream;
    ctx->frames_buffered = 0;
    ctx->frames_decoded = 0;
    HIPCHK(hipMemsetAsync(ctx->fidx, 0, ctx->cfg.max_streams * sizeof(int), s));
    return RNNT_OK;
}

int rnnt_beam_frame(rnnt_ctx* ctx, int32_t frame_idx, int32_t n_rows, const int32_t* row_stream_host, const int32_t* row_tok_host,
                    int32_t beam_k, int32_t* steps_host, float* blank_lp_host, float* top_lp_host, int32_t* top_tok_host, void* stream) {
    if (!ctx || !row_stream_host || !row_tok_host || !steps_host || !blank_lp_host || !top_lp_host || !top_tok_host)
        return fail(ctx, RNNT_ERR_ARG, "rnnt_beam_frame: null argument");
    if (!ctx->finalized || ctx->n_streams < 1) return fail(ctx, RNNT_ERR_STATE, "rnnt_beam_frame: no weights / no streams");
    if (ctx->max_rows == 0) return fail(ctx, RNNT_ERR_STATE, "rnnt_beam_frame: context created with max_beam = 0");
    if (n_rows < 1 || n_rows > ctx->max_rows || beam_k < 1 || beam_k > ctx->cfg.max_beam || beam_k > ctx->cfg.vocab_size - 1)
        return fail(ctx, RNNT_ERR_ARG, "rnnt_beam_frame: n_rows=%d beam_k=%d out of range", n_rows, beam_k);
    if (frame_idx < 0 || frame_idx >= ctx->frames_buffered) return fail(ctx, RNNT_ERR_ARG, "rnnt_beam_frame: frame %d not buffered", frame_idx);
    hipStream_t s = (hipStream_t)stream;
    const int V = ctx->cfg.vocab_size, NS = ctx->cfg.n_steps, slots = NS + 1, R = n_rows;
    std::vector<int> fr(R);
    for (int r = 0; r < R; ++r) {
        if (row_stream_host[r] < 0 || row_stream_host[r] >= ctx->n_streams) return fail(ctx, RNNT_ERR_ARG, "rnnt_beam_frame: bad stream index");
        fr[r] = row_stream_host[r] * ctx->fstride + frame_idx;   // row of the projected-encoder-frame table
    }
    HIPCHK(hipMemcpyAsync(ctx->b_frame, fr.data(), R * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(ctx->b_tok, row_tok_host, R * sizeof(int), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(fill_i32, dim3(1), dim3(256), 0, s, ctx->b_active, 1, (long long)R);
    LAUNCHCHK("fill_i32");
    hipLaunchKernelGGL(fill_i32, dim3(1), dim3(64), 0, s, ctx->n_active + 1, R, 1LL);
    LAUNCHCHK("fill_i32");
    HIPCHK(hipMemsetAsync(ctx->b_steps, 0, R * sizeof(int), s));
    float* pool = ctx->pool[ctx->pool_cur];
    BeamOut bo{ctx->b_active, ctx->b_tok, ctx->b_steps, ctx->b_blank, ctx->b_toplp, ctx->b_toptok, ctx->n_active + 1};
    int rc;
    if (ctx->use_beam_chain && V <= 512) {   // one resident workgroup per row runs its whole extension chain
        BeamChainP c;
        memset(&c, 0, sizeof(c));
        c.whh = ctx->whh_il; c.egate = ctx->egate; c.wpr = ctx->wpr; c.bpr = ctx->bpr; c.wpf = ctx->wpf; c.bpf = ctx->bpf;
        c.wout = ctx->wout; c.bout = ctx->bout; c.encp = ctx->encp; c.pool = pool; c.frame = ctx->b_frame; c.tok_in = ctx->b_tok;
        c.steps = ctx->b_steps; c.blank_lp = ctx->b_blank; c.top_lp = ctx->b_toplp; c.top_tok = ctx->b_toptok;
        c.vocab = V; c.blank = ctx->cfg.blank_id; c.k = beam_k; c.n_steps = NS; c.slots = slots;
        hipLaunchKernelGGL(beam_chain, dim3(R), dim3(512), 0, s, c);
        LAUNCHCHK("beam_chain");
        ctx->launches += 1;
    } else
    for (int st = 0; st < NS; ++st) {
        float* sin = pool + (size_t)st * 512;
        float* sout = pool + (size_t)(st + 1) * 512;
        GemmP g1 = plain_gemm(sin, slots * 512, ctx->whh_il, D, nullptr, sout, D, R, 4 * D, D, EPI_LSTM);
        g1.X = ctx->egate; g1.I = ctx->b_tok; g1.X2 = sin + D; g1.Y2 = sout + D; g1.lstm_ld = slots * 512;
        if ((rc = launch_gemm(ctx, s, 0, &g1, 1, TAG_LSTM))) return rc;
        GemmP g2 = plain_gemm(sout, slots * 512, ctx->wpr, D, ctx->bpr, ctx->bpred, D, R, D, D);
        if ((rc = launch_gemm(ctx, s, 0, &g2, 1, TAG_PRED_PROJ))) return rc;
        GemmP g3 = plain_gemm(ctx->bpred, D, ctx->wpf, D, ctx->bpf, ctx->bz, D, R, D, D, EPI_TANH_ADD);
        g3.X = ctx->encp; g3.I = ctx->b_frame; g3.x_n = 1; g3.x_s0 = 0; g3.x_s1 = D;
        if ((rc = launch_gemm(ctx, s, 0, &g3, 1, TAG_JOINT_TANH))) return rc;
        GemmP g4 = plain_gemm(ctx->bz, D, ctx->wout, D, ctx->bout, ctx->blogits, ctx->vpad, R, V, D);
        if ((rc = launch_gemm(ctx, s, 0, &g4, 1, TAG_JOINT_OUT))) return rc;
        hipLaunchKernelGGL(beam_reduce, dim3(R), dim3(64), 0, s, ctx->blogits, ctx->vpad, V, ctx->cfg.blank_id, beam_k, st, NS, bo);
        LAUNCHCHK("beam_reduce");
        HIPCHK(hipMemcpyAsync(ctx->pinned + 1, ctx->n_active + 1, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (ctx->pinned[1] <= 0) break;
    }
    HIPCHK(hipMemcpyAsync(steps_host, ctx->b_steps, R * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(blank_lp_host, ctx->b_blank, (size_t)R * NS * sizeof(float), hipMemcpyDeviceToHost, s));
    // device layout of the top-k arrays is [R][NS][beam_k] with THIS call's beam_k
    HIPCHK(hipMemcpyAsync(top_lp_host, ctx->b_toplp, (size_t)R * NS * beam_k * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(top_tok_host, ctx->b_toptok, (size_t)R * NS * beam_k * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return RNNT_OK;
}

int rnnt_beam_select(rnnt_ctx* ctx, int32_t n_new, const int32_t* src_row_host, const int32_t* src_step_host, void* stream) {
    if (!ctx || !src_row_host || !src_step_host) return fail(ctx, RNNT_ERR_ARG, "rnnt_beam_select: null argument");
    if (ctx->max_rows == 0 || n_new < 1 || n_new > ctx->max_rows) return fail(ctx, RNNT_ERR_ARG, "rnnt_beam_select: n_new out of range");
    hipStream_t s = (hipStream_t)stream;
    const int slots = ctx->cfg.n_steps + 1;
    for (int r = 0; r < n_new; ++r)
        if (src_row_host[r] < 0 || src_row_host[r] >= ctx->max_rows || src_step_host[r] < 0 || src_step_host[r] >= slots)
            return fail(ctx, RNNT_ERR_ARG, "rnnt_beam_select: bad source slot");
    HIPCHK(hipMemcpyAsync(ctx->b_srcrow, src_row_host, n_new * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(ctx->b_srcstep, src_step_host, n_new * sizeof(int), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(beam_gather, dim3(n_new), dim3(128), 0, s, ctx->pool[ctx->pool_cur], ctx->pool[ctx->pool_cur ^ 1], ctx->b_srcrow, ctx->b_srcstep,
                       n_new, slots);
    LAUNCHCHK("beam_gather");
    HIPCHK(hipStreamSynchronize(s));   // the host arrays may be reused by the caller
    ctx->pool_cur ^= 1;
    return RNNT_OK;
}

// Host half of one encoder frame of _decode_chunk_beam_search (model/online_rnnt_model.py:419-518) for ONE stream, pure
// host code (also exported for CPU tests): candidates in the reference's order (per hypothesis, per evaluation: the blank
// candidate keeping the old state, then the top-k non-blank), scores accumulated in double exactly as Python floats
// (float32 log-prob widened, then added), stable descending sort (:506), first-wins de-duplication on the token
// sequence (:508-516), truncation to the beam.  src_row / src_step name the pooled LSTM state each survivor keeps.
namespace {
struct BeamCand { std::vector<int> tokens; double score; int row, step; };
void beam_merge_stream(const std::vector<rnnt_ctx::Hyp>& beam, int row0, const int* steps, const float* blank_lp, const float* top_lp,
                       const int* top_tok, int n_steps, int k, int beam_size, std::vector<rnnt_ctx::Hyp>& out, std::vector<int>& src_row,
                       std::vector<int>& src_step) {
    std::vector<BeamCand> cands;
    int r = row0;
    for (const rnnt_ctx::Hyp& h : beam) {
        std::vector<int> toks = h.tokens;
        double lp = h.log_prob;
        const int n = steps[r];
        for (int st = 0; st < n; ++st) {
            cands.push_back({toks, lp + (double)blank_lp[(size_t)r * n_steps + st], r, st});
            for (int j = 0; j < k; ++j) {
                BeamCand c{toks, lp + (double)top_lp[((size_t)r * n_steps + st) * k + j], r, st + 1};
                c.tokens.push_back(top_tok[((size_t)r * n_steps + st) * k + j]);
                cands.push_back(std::move(c));
            }
            if (st < n - 1) {   // chain continued with the best non-blank (:489-499)
                toks.push_back(top_tok[((size_t)r * n_steps + st) * k]);
                lp += (double)top_lp[((size_t)r * n_steps + st) * k];
            }
        }
        ++r;
    }
    std::stable_sort(cands.begin(), cands.end(), [](const BeamCand& a, const BeamCand& b) { return a.score > b.score; });
    out.clear();
    for (BeamCand& c : cands) {
        bool dup = false;
        for (const rnnt_ctx::Hyp& u : out)
            if (u.tokens == c.tokens) { dup = true; break; }
        if (dup) continue;
        out.push_back(rnnt_ctx::Hyp{std::move(c.tokens), c.score});
        src_row.push_back(c.row);
        src_step.push_back(c.step);
        if ((int)out.size() >= beam_size) break;
    }
}
}  // namespace

// Pure-host export of beam_merge_stream for one stream (CPU tests; no context, no GPU).  Hypotheses are passed flat:
// hyp_len[n_hyp], hyp_tokens (concatenated), hyp_score[n_hyp]; outputs likewise (out_tokens needs room for
// beam_size * (longest input + n_steps) ints).  Returns the number of surviving hypotheses.
int rnnt_beam_merge_host(int32_t n_hyp, const int32_t* hyp_len, const int32_t* hyp_tokens, const double* hyp_score, const int32_t* steps,
                         const float* blank_lp, const float* top_lp, const int32_t* top_tok, int32_t n_steps, int32_t k, int32_t beam_size,
                         int32_t* out_len, int32_t* out_tokens, double* out_score, int32_t* out_src_row, int32_t* out_src_step) {
    if (n_hyp < 1 || !hyp_len || !hyp_score || !steps || !blank_lp || !top_lp || !top_tok || !out_len || !out_tokens || !out_score) return RNNT_ERR_ARG;
    std::vector<rnnt_ctx::Hyp> beam(n_hyp), out;
    size_t off = 0;
    for (int i = 0; i < n_hyp; ++i) {
        beam[i].tokens.assign(hyp_tokens + off, hyp_tokens + off + hyp_len[i]);
        beam[i].log_prob = hyp_score[i];
        off += hyp_len[i];
    }
    std::vector<int> sr, ss;
    beam_merge_stream(beam, 0, steps, blank_lp, top_lp, top_tok, n_steps, k, beam_size, out, sr, ss);
    off = 0;
    for (size_t i = 0; i < out.size(); ++i) {
        out_len[i] = (int)out[i].tokens.size();
        for (int t : out[i].tokens) out_tokens[off++] = t;
        out_score[i] = out[i].log_prob;
        if (out_src_row) out_src_row[i] = sr[i];
        if (out_src_step) out_src_step[i] = ss[i];
    }
    return (int)out.size();
}

// Beam search over the buffered encoder frames [frame_begin, frame_end) of every stream with the bookkeeping inside the
// library (the reference's per-frame loop, online_rnnt_model.py:419-518, for all streams at once): per frame one
// beam_chain launch, one copy of the candidates to the host, the merge above, one state-pool gather.
int rnnt_beam_advance(rnnt_ctx* ctx, int32_t frame_begin, int32_t frame_end, int32_t beam_size, void* stream) {
    if (!ctx) return RNNT_ERR_ARG;
    if (!ctx->finalized || ctx->n_streams < 1) return fail(ctx, RNNT_ERR_STATE, "rnnt_beam_advance: no weights / no streams");
    if (ctx->max_rows == 0) return fail(ctx, RNNT_ERR_STATE, "rnnt_beam_advance: context created with max_beam = 0");
    if (beam_size < 1 || beam_size > ctx->cfg.max_beam) return fail(ctx, RNNT_ERR_ARG, "rnnt_beam_advance: beam_size %d outside [1, max_beam %d]", beam_size, ctx->cfg.max_beam);
    if (frame_begin < 0 || frame_end > ctx->frames_buffered || frame_begin > frame_end) return fail(ctx, RNNT_ERR_ARG, "rnnt_beam_advance: frames [%d, %d) not buffered", frame_begin, frame_end);
    if ((int)ctx->beams.size() != ctx->n_streams) return fail(ctx, RNNT_ERR_STATE, "rnnt_beam_advance: call rnnt_streams_reset first");
    const int NS = ctx->cfg.n_steps, B = ctx->n_streams;
    const int k = beam_size < ctx->cfg.vocab_size - 1 ? beam_size : ctx->cfg.vocab_size - 1;          // :467
    std::vector<int> row_stream, row_tok, steps, top_tok, src_row, src_step;
    std::vector<float> blank_lp, top_lp;
    std::vector<rnnt_ctx::Hyp> next;
    int rc;
    for (int f = frame_begin; f < frame_end; ++f) {
        row_stream.clear(); row_tok.clear();
        for (int b = 0; b < B; ++b)
            for (const rnnt_ctx::Hyp& h : ctx->beams[b]) {
                row_stream.push_back(b);
                row_tok.push_back(h.tokens.empty() ? ctx->cfg.blank_id : h.tokens.back());              // :429
            }
        const int R = (int)row_stream.size();
        steps.resize(R); blank_lp.resize((size_t)R * NS); top_lp.resize((size_t)R * NS * k); top_tok.resize((size_t)R * NS * k);
        if ((rc = rnnt_beam_frame(ctx, f, R, row_stream.data(), row_tok.data(), k, steps.data(), blank_lp.data(), top_lp.data(), top_tok.data(), stream)))
            return rc;
        src_row.clear(); src_step.clear();
        int row0 = 0;
        for (int b = 0; b < B; ++b) {
            const int nh = (int)ctx->beams[b].size();
            beam_merge_stream(ctx->beams[b], row0, steps.data(), blank_lp.data(), top_lp.data(), top_tok.data(), NS, k, beam_size, next, src_row, src_step);
            row0 += nh;
            ctx->beams[b].swap(next);
        }
        if ((rc = rnnt_beam_select(ctx, (int)src_row.size(), src_row.data(), src_step.data(), stream))) return rc;
    }
    return RNNT_OK;
}

// hypotheses of one stream after rnnt_beam_advance: count, then tokens / score of hypothesis i (device row = rows of the
// earlier streams + i, the index rnnt_beam_get_states uses)
int rnnt_beam_hyp_count(rnnt_ctx* ctx, int32_t stream_idx, int32_t* n_out) {
    if (!ctx || !n_out || stream_idx < 0 || stream_idx >= (int)ctx->beams.size()) return fail(ctx, RNNT_ERR_ARG, "rnnt_beam_hyp_count: bad argument");
    *n_out = (int)ctx->beams[stream_idx].size();
    return RNNT_OK;
}
int rnnt_beam_get_hyp(rnnt_ctx* ctx, int32_t stream_idx, int32_t hyp_idx, int32_t cap, int32_t* tokens_host, int32_t* n_tokens, double* log_prob) {
    if (!ctx || stream_idx < 0 || stream_idx >= (int)ctx->beams.size() || hyp_idx < 0 || hyp_idx >= (int)ctx->beams[stream_idx].size())
        return fail(ctx, RNNT_ERR_ARG, "rnnt_beam_get_hyp: bad index");
    const rnnt_ctx::Hyp& h = ctx->beams[stream_idx][hyp_idx];
    if (n_tokens) *n_tokens = (int)h.tokens.size();
    if (log_prob) *log_prob = h.log_prob;
    if (tokens_host) {
        if (cap < (int)h.tokens.size()) return fail(ctx, RNNT_ERR_ARG, "rnnt_beam_get_hyp: %d tokens, room for %d", (int)h.tokens.size(), cap);
        for (size_t i = 0; i < h.tokens.size(); ++i) tokens_host[i] = h.tokens[i];
    }
    return RNNT_OK;
}

int rnnt_beam_get_states(rnnt_ctx* ctx, int32_t n_rows, float* h_host, float* c_host, void* stream) {
    if (!ctx || !h_host || !c_host || n_rows < 1 || n_rows > ctx->max_rows) return fail(ctx, RNNT_ERR_ARG, "rnnt_beam_get_states: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const size_t pitch = (size_t)(ctx->cfg.n_steps + 1) * 512 * sizeof(float);
    const float* pool = ctx->pool[ctx->pool_cur];
    HIPCHK(hipMemcpy2DAsync(h_host, D * sizeof(float), pool, pitch, D * sizeof(float), n_rows, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpy2DAsync(c_host, D * sizeof(float), pool + D, pitch, D * sizeof(float), n_rows, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return RNNT_OK;
}

int rnnt_frames_discard(rnnt_ctx* ctx, void* stream) {
    if (!ctx) return RNNT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    ctx->frames_buffered = 0;
    ctx->frames_decoded = 0;
    HIPCHK(hipMemsetAsync(ctx->fidx, 0, ctx->cfg.max_streams * sizeof(int), s));
    return RNNT_OK;
}

int rnnt_predictor_step(rnnt_ctx* ctx, const int32_t* tokens_dev, const float* h_in, const float* c_in, int32_t rows, float* out_dev,
                        float* h_out, float* c_out, void* stream) {
    if (!ctx || !tokens_dev || !h_in || !c_in || !out_dev || !h_out || !c_out || rows < 1) return fail(ctx, RNNT_ERR_ARG, "rnnt_predictor_step: bad argument");
    if (!ctx->finalized) return fail(ctx, RNNT_ERR_STATE, "weights not finalized");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    GemmP g1 = plain_gemm(h_in, D, ctx->whh_il, D, nullptr, h_out, D, rows, 4 * D, D, EPI_LSTM);
    g1.X = ctx->egate; g1.I = tokens_dev; g1.X2 = c_in; g1.Y2 = c_out;
    if ((rc = launch_gemm(ctx, s, 0, &g1, 1))) return rc;
    GemmP g2 = plain_gemm(h_out, D, ctx->wpr, D, ctx->bpr, out_dev, D, rows, D, D);
    return launch_gemm(ctx, s, 0, &g2, 1);
}

int rnnt_joint(rnnt_ctx* ctx, const float* enc_dev, const float* pred_dev, int32_t B, int32_t T, int32_t U, int32_t mode, float* logits_dev,
               void* stream) {
    if (!ctx || !enc_dev || !pred_dev || !logits_dev || B < 1 || T < 1 || U < 1) return fail(ctx, RNNT_ERR_ARG, "rnnt_joint: bad argument");
    if (!ctx->finalized) return fail(ctx, RNNT_ERR_STATE, "weights not finalized");
    hipStream_t s = (hipStream_t)stream;
    const int V = ctx->cfg.vocab_size;
    const size_t needf = (size_t)B * T * D + (size_t)B * U * D;
    if (needf > ctx->scratch_floats) return fail(ctx, RNNT_ERR_SHAPE, "joint lattice B=%d T=%d U=%d exceeds the context scratch", B, T, U);
    float* e = ctx->scratch;
    float* pp = e + (size_t)B * T * D;
    int rc;
    GemmP ge = plain_gemm(enc_dev, D, ctx->wenc, D, ctx->benc, e, D, B * T, D, D);
    if ((rc = launch_gemm(ctx, s, 0, &ge, 1))) return rc;
    GemmP gp = plain_gemm(pred_dev, D, ctx->wpf, D, ctx->bpf, pp, D, B * U, D, D);
    if ((rc = launch_gemm(ctx, s, 0, &gp, 1))) return rc;
    // logits[b,t,u,:] = tanh(e[b,t,:] + pp[b,u,:]) * W_out^T + b_out in ONE kernel: rows m = (b,t,u); the tanh of the
    // broadcast sum is formed while the A tile is staged into LDS (no [B,T,U,256] intermediate in HBM).
    {
        ProfScope prof(ctx, s, TAG_JOINT_OUT);
        GemmBatch gb;
        memset(&gb, 0, sizeof(gb));
        GemmP& g = gb.g[0];
        g = plain_gemm(pp, D, ctx->wout, D, ctx->bout, logits_dev, V, B * T * U, V, D);
        g.a_n1 = T * U; g.a_n2 = U; g.a_s0 = (long long)U * D; g.a_s1 = 0; g.a_s2 = D;   // A row = pp[b, u]
        g.a_tanh = 1; g.X = e; g.x_n = U; g.x_s0 = D;                                    // X row = e[(b,t)] = e[m / U]
        if ((rc = prepare_gemm(ctx, g))) return rc;
        launch_gemm_ns<2, 2, true>(s, gb, g.M, V, 1);
        LAUNCHCHK("gemm_ns(joint lattice)");
    }
    if (mode == 1) {
        const long long rows = (long long)B * T * U;
        hipLaunchKernelGGL(log_softmax_rows, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits_dev, rows, V);
        LAUNCHCHK("log_softmax_rows");
    }
    return RNNT_OK;
}

int rnnt_encoder_full(rnnt_ctx* ctx, const float* fbank_dev, const int32_t* lens_host, int32_t B, int32_t T, float* out_dev,
                      int32_t* frames_out, void* stream) {
    if (!ctx || !fbank_dev || !lens_host || !out_dev) return fail(ctx, RNNT_ERR_ARG, "rnnt_encoder_full: null argument");
    if (!ctx->finalized) return fail(ctx, RNNT_ERR_STATE, "weights not finalized");
    if (B < 1 || B > ctx->cfg.max_streams || T < 7 || T > ctx->cfg.max_chunk_frames)
        return fail(ctx, RNNT_ERR_SHAPE, "rnnt_encoder_full: B=%d T=%d outside the context limits", B, T);
    const int tq = sub_len(T);
    if (tq > ctx->tcap) return fail(ctx, RNNT_ERR_SHAPE, "rnnt_encoder_full: %d frames exceed the K/V capacity %d", tq, ctx->tcap);
    hipStream_t s = (hipStream_t)stream;
    // padding mask after subsampling: masks[:, :, 2::2][:, :, 2::2] (subsampling.py:228)
    std::vector<int> kl(B);
    for (int b = 0; b < B; ++b) {
        const int len = lens_host[b] < T ? lens_host[b] : T;
        const int n1 = len > 2 ? (len - 1) / 2 : 0;
        kl[b] = n1 > 2 ? (n1 - 1) / 2 : 0;
        if (kl[b] < 1) return fail(ctx, RNNT_ERR_SHAPE, "rnnt_encoder_full: utterance %d too short", b);
    }
    HIPCHK(hipMemcpyAsync(ctx->klen, kl.data(), B * sizeof(int), hipMemcpyHostToDevice, s));
    int rc;
    // fresh left context for the causal conv (zero pad, convolution.py:122-124); streaming state is clobbered
    hipLaunchKernelGGL(conv_ring_init, dim3(grid_for((long long)L * ctx->cfg.max_streams * ctx->cap * D)), dim3(256), 0, s, ctx->gring, ctx->xring,
                       ctx->glu0, ctx->cfg.max_streams, ctx->cap);
    LAUNCHCHK("conv_ring_init");
    const int saved = ctx->n_streams;
    ctx->n_streams = 0;   // streaming state invalid after a full-context pass
    (void)saved;
    if ((rc = run_subsample(ctx, s, fbank_dev, B, T, T, nullptr, 1, ctx->y1, ctx->y2, ctx->x))) return rc;
    for (int l = 0; l < L; ++l)
        if ((rc = run_layer(ctx, s, l, B, tq, tq, 0, 0, 0, ctx->klen))) return rc;
    if ((rc = launch_ln(ctx, s, LnP{ctx->x, ctx->after_g, ctx->after_b, out_dev, B * tq, BIG, 0, 0LL, (long long)D}))) return rc;
    if (frames_out) *frames_out = tq;
    return RNNT_OK;
}

// OnlineCTC.argmax over the full-context encoder (model/online_rnnt_model.py:37-38,655-658): per-frame argmax of
// ctc_lo(encoder(x)) for B utterances; ids_host [B, T'] int32.  The collapse rule (:660-671) stays on the host.
int rnnt_ctc_argmax(rnnt_ctx* ctx, const float* fbank_dev, const int32_t* lens_host, int32_t B, int32_t T, int32_t* ids_host,
                    int32_t* frames_out, void* stream) {
    if (!ctx || !ids_host) return fail(ctx, RNNT_ERR_ARG, "rnnt_ctc_argmax: null argument");
    if (!ctx->finalized) return fail(ctx, RNNT_ERR_STATE, "weights not finalized");
    if (!ctx->wctc) return fail(ctx, RNNT_ERR_STATE, "rnnt_ctc_argmax: ctc_head.ctc_lo.* not loaded");
    hipStream_t s = (hipStream_t)stream;
    const int tq = sub_len(T);
    const size_t rows = (size_t)B * tq;
    if (rows * D + rows * 3 > ctx->scratch_floats) return fail(ctx, RNNT_ERR_SHAPE, "rnnt_ctc_argmax: B=%d T=%d exceeds the context scratch", B, T);
    float* enc = ctx->scratch;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(ctx->scratch + ((rows * D + 1) / 2) * 2);
    int* ids = reinterpret_cast<int*>(keys + rows);
    int rc, fo = 0;
    if ((rc = rnnt_encoder_full(ctx, fbank_dev, lens_host, B, T, enc, &fo, stream))) return rc;
    HIPCHK(hipMemsetAsync(keys, 0, rows * sizeof(unsigned long long), s));
    GemmP g = plain_gemm(enc, D, ctx->wctc, D, ctx->bctc, nullptr, 0, (int)rows, ctx->cfg.vocab_size, D, EPI_ARGMAX);
    g.key = keys;
    if ((rc = launch_gemm(ctx, s, 0, &g, 1))) return rc;   // EPI_ARGMAX always takes the gemm16 (split-K) kernel
    hipLaunchKernelGGL(unpack_keys, dim3(grid_for((long long)rows)), dim3(256), 0, s, keys, ids, (long long)rows);
    LAUNCHCHK("unpack_keys");
    HIPCHK(hipMemcpyAsync(ids_host, ids, rows * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (frames_out) *frames_out = tq;
    return RNNT_OK;
}

// Offline greedy search: model/component/transducer.py:22-70 `basic_greedy_search` as reached from
// OnlineRNNTModel.forward (online_rnnt_model.py:234-235,268) for a non-streaming model: full-context encoder, then per
// utterance a greedy RNN-T loop over its own valid frames with at most n_steps symbols per frame (default 64 there), zero
// predictor state, first token = blank.  (The reference reads `model.blank`, an attribute OnlineRNNTModel does not have;
// the blank id of the context is used.)  Unlike the streaming loop there is no forced frame advance after n_steps symbols
// other than leaving the inner loop -- which is the same thing -- so the resident decoder is reused unchanged with
// per-stream frame counts.  counts_host [B], tokens_host [B][max_tokens].
int rnnt_greedy_search_full(rnnt_ctx* ctx, const float* fbank_dev, const int32_t* lens_host, int32_t B, int32_t T, int32_t n_steps,
                            int32_t* counts_host, int32_t* tokens_host, void* stream) {
    if (!ctx || !counts_host) return fail(ctx, RNNT_ERR_ARG, "rnnt_greedy_search_full: null argument");
    if (n_steps < 1) return fail(ctx, RNNT_ERR_ARG, "rnnt_greedy_search_full: n_steps %d", n_steps);
    hipStream_t s = (hipStream_t)stream;
    const int tq = sub_len(T);
    const size_t rows = (size_t)B * tq;
    if (rows * D > ctx->scratch_floats) return fail(ctx, RNNT_ERR_SHAPE, "rnnt_greedy_search_full: B=%d T=%d exceeds the context scratch", B, T);
    if (tq > ctx->fcap) return fail(ctx, RNNT_ERR_SHAPE, "rnnt_greedy_search_full: %d encoder frames exceed max_enc_frames %d", tq, ctx->fcap);
    int rc, fo = 0;
    if ((rc = rnnt_streams_reset(ctx, B, stream))) return rc;                                   // zero predictor state, token = blank
    if ((rc = rnnt_encoder_full(ctx, fbank_dev, lens_host, B, T, ctx->scratch, &fo, stream))) return rc;   // also fills ctx->klen (valid frames)
    ctx->n_streams = B;
    {   // joint.enc_ffn of every frame, [B*tq, 256] -> enc_proj [B][fstride][256]
        GemmP g = plain_gemm(ctx->scratch, D, ctx->wenc, D, ctx->benc, ctx->encp, D, (int)rows, D, D);
        g.c_n = tq; g.c_s0 = (long long)ctx->fstride * D; g.c_r0 = 0; g.c_mod = BIG; g.c_s1 = D;
        if ((rc = launch_gemm(ctx, s, 0, &g, 1, TAG_ENC_PROJ))) return rc;
    }
    if ((rc = init_decoder_ctrl(ctx, s, tq))) return rc;
    if ((rc = launch_persistent_decoder(ctx, s, tq, n_steps, ctx->klen))) return rc;
    if ((rc = finish_persistent_decoder(ctx, s))) return rc;
    ctx->n_streams = 0;                                                                        // streaming state is not meaningful afterwards
    HIPCHK(hipMemcpyAsync(counts_host, ctx->count, B * sizeof(int), hipMemcpyDeviceToHost, s));
    if (tokens_host) HIPCHK(hipMemcpyAsync(tokens_host, ctx->tokens, (size_t)B * ctx->cfg.max_tokens * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return RNNT_OK;
}

// Feature front-end on the device: the reference's extract_audio_features (data/dataloader.py:15-41) =
// torchaudio MelSpectrogram(sample_rate, n_fft, n_mels=80, hop_length=512, window_fn=hamming_window, power=2.0)
// [defaults: win_length = n_fft, center=True, pad_mode="reflect", onesided, HTK mel scale, norm=None, f_min=0,
// f_max=sample_rate/2] followed by AmplitudeToDB() [power: 10*log10(clamp(x, 1e-10)), ref 1.0, no top_db].
// wave_dev [B][n_samples] float32 mono -> out_dev [B][1 + n_samples/512][80].  No model weights are needed.
int rnnt_fbank(rnnt_ctx* ctx, const float* wave_dev, int32_t B, int32_t n_samples, int32_t sample_rate, int32_t n_fft, float* out_dev,
               int32_t* frames_out, void* stream) {
    if (!ctx || !wave_dev || !out_dev) return fail(ctx, RNNT_ERR_ARG, "rnnt_fbank: null argument");
    const int hop = 512, n_mels = 80;
    if (B < 1 || sample_rate < 2 || n_fft < 64 || n_fft > 4096 || n_fft % 64 != 0)
        return fail(ctx, RNNT_ERR_SHAPE, "rnnt_fbank: B=%d sample_rate=%d n_fft=%d (n_fft must be a multiple of 64 in [64, 4096])", B, sample_rate, n_fft);
    if (n_samples <= n_fft / 2) return fail(ctx, RNNT_ERR_SHAPE, "rnnt_fbank: reflect padding needs more than n_fft/2 = %d samples (got %d)", n_fft / 2, n_samples);
    hipStream_t s = (hipStream_t)stream;
    const int nfreq = n_fft / 2 + 1;
    const int n2p = (2 * nfreq + 63) / 64 * 64;          // DFT output columns (re, im interleaved), padded
    const int kp = (nfreq + 63) / 64 * 64;               // K of the mel projection, padded with zero columns
    const int T = 1 + n_samples / hop;
    const long long pstride = ((long long)n_samples + n_fft + 3) / 4 * 4;   // padded signal per stream, 16-B aligned rows
    const long long M = (long long)B * T;
    if (M > 0x7fffffffLL / 8) return fail(ctx, RNNT_ERR_SHAPE, "rnnt_fbank: %lld frames in one call", M);
    int rc;
    if (ctx->fb_rate != sample_rate || ctx->fb_nfft != n_fft) {   // matrices in double on the host, once per (rate, n_fft)
        HIPCHK(hipStreamSynchronize(s));
        if (ctx->fb_dft) { (void)hipFree(ctx->fb_dft); ctx->fb_dft = nullptr; }
        if (ctx->fb_mel) { (void)hipFree(ctx->fb_mel); ctx->fb_mel = nullptr; }
        if ((rc = dmalloc(ctx, &ctx->fb_dft, (size_t)n2p * n_fft))) return rc;
        if ((rc = dmalloc(ctx, &ctx->fb_mel, (size_t)n_mels * kp))) return rc;
        const double pi = 3.14159265358979323846;
        std::vector<float> dft((size_t)n2p * n_fft, 0.f), mel((size_t)n_mels * kp, 0.f);
        std::vector<double> win(n_fft);
        for (int n = 0; n < n_fft; ++n) win[n] = 0.54 - 0.46 * cos(2.0 * pi * n / n_fft);   // torch.hamming_window (periodic)
        for (int k = 0; k < nfreq; ++k)
            for (int n = 0; n < n_fft; ++n) {
                const long long kn = ((long long)k * n) % n_fft;                              // exact phase reduction
                const double ph = 2.0 * pi * (double)kn / n_fft;
                dft[(size_t)(2 * k) * n_fft + n] = (float)(win[n] * cos(ph));
                dft[(size_t)(2 * k + 1) * n_fft + n] = (float)(-win[n] * sin(ph));
            }
        // torchaudio.functional.melscale_fbanks(n_freqs, 0, rate/2, 80, rate, norm=None, mel_scale="htk")
        const double fmax = (double)(sample_rate / 2);   // all_freqs = linspace(0, sample_rate // 2, n_freqs)
        const double m_max = 2595.0 * log10(1.0 + fmax / 700.0);             // f_max = float(sample_rate // 2)
        std::vector<double> fpts(n_mels + 2);
        for (int i = 0; i < n_mels + 2; ++i) fpts[i] = 700.0 * (pow(10.0, (m_max * i / (n_mels + 1)) / 2595.0) - 1.0);
        for (int k = 0; k < nfreq; ++k) {
            const double f = fmax * k / (nfreq - 1);
            for (int m = 0; m < n_mels; ++m) {
                const double down = (f - fpts[m]) / (fpts[m + 1] - fpts[m]);
                const double up = (fpts[m + 2] - f) / (fpts[m + 2] - fpts[m + 1]);
                const double v = down < up ? down : up;
                mel[(size_t)m * kp + k] = (float)(v > 0.0 ? v : 0.0);
            }
        }
        HIPCHK(hipMemcpy(ctx->fb_dft, dft.data(), dft.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(ctx->fb_mel, mel.data(), mel.size() * sizeof(float), hipMemcpyHostToDevice));
        ctx->fb_rate = sample_rate;
        ctx->fb_nfft = n_fft;
    }
    if ((rc = grow(ctx, &ctx->fb_pad, &ctx->fb_pad_cap, (size_t)B * pstride))) return rc;
    if ((rc = grow(ctx, &ctx->fb_spec, &ctx->fb_spec_cap, (size_t)M * n2p))) return rc;
    if ((rc = grow(ctx, &ctx->fb_pow, &ctx->fb_pow_cap, (size_t)M * kp))) return rc;
    hipLaunchKernelGGL(reflect_pad, dim3(grid_for((long long)B * pstride)), dim3(256), 0, s, wave_dev, ctx->fb_pad, B, n_samples, n_fft / 2, pstride);
    LAUNCHCHK("reflect_pad");
    // windowed DFT: implicit frames (row t of stream b starts at b*pstride + t*hop), K = n_fft
    GemmP g1 = plain_gemm(ctx->fb_pad, hop, ctx->fb_dft, n_fft, nullptr, ctx->fb_spec, n2p, (int)M, n2p, n_fft);
    g1.a_n1 = T; g1.a_n2 = T; g1.a_s0 = pstride; g1.a_s1 = 0; g1.a_s2 = hop;
    if ((rc = launch_gemm(ctx, s, 0, &g1, 1))) return rc;
    hipLaunchKernelGGL(power_spectrum, dim3(grid_for(M * kp)), dim3(256), 0, s, ctx->fb_spec, ctx->fb_pow, M, nfreq, kp, n2p);
    LAUNCHCHK("power_spectrum");
    GemmP g2 = plain_gemm(ctx->fb_pow, kp, ctx->fb_mel, kp, nullptr, out_dev, n_mels, (int)M, n_mels, kp, EPI_DB);
    if ((rc = launch_gemm(ctx, s, 0, &g2, 1))) return rc;
    if (frames_out) *frames_out = T;
    return RNNT_OK;
}

// diagnostic only (not in the header): n_bytes streamed src -> dst `iters` times with the given cache policy
int rnnt_debug_stream_copy(rnnt_ctx* ctx, const float* src_dev, float* dst_dev, int64_t n_bytes, int32_t mode, int32_t iters, void* stream) {
    if (!ctx || !src_dev || !dst_dev) return RNNT_ERR_ARG;
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL(debug_stream_copy, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, src_dev, dst_dev, (long long)(n_bytes / 16), mode);
    LAUNCHCHK("debug_stream_copy");
    return RNNT_OK;
}

int rnnt_get_att_cache(rnnt_ctx* ctx, int32_t b, float* dst_host, int32_t* len_out, void* stream) {
    if (!ctx || b < 0 || b >= ctx->n_streams) return fail(ctx, RNNT_ERR_ARG, "rnnt_get_att_cache: bad stream index");
    hipStream_t s = (hipStream_t)stream;
    if (len_out) *len_out = ctx->cache_len;
    if (!dst_host || ctx->cache_len == 0) return RNNT_OK;
    const long long n = (long long)L * RNNT_H * ctx->cache_len * 128;
    hipLaunchKernelGGL(gather_att_cache, dim3(grid_for(n)), dim3(256), 0, s, ctx->kcache, ctx->vcache, ctx->scratch, b, ctx->cfg.max_streams,
                       (long long)ctx->tcap, ctx->kv_start, ctx->cache_len);
    LAUNCHCHK("gather_att_cache");
    HIPCHK(hipMemcpyAsync(dst_host, ctx->scratch, n * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return RNNT_OK;
}

int rnnt_get_cnn_cache(rnnt_ctx* ctx, int32_t b, float* dst_host, void* stream) {
    if (!ctx || !dst_host || b < 0 || b >= ctx->n_streams) return fail(ctx, RNNT_ERR_ARG, "rnnt_get_cnn_cache: bad argument");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gather_cnn_cache, dim3(L * RNNT_LORDER), dim3(64), 0, s, ctx->xring, ctx->ln_conv_g_all, ctx->ln_conv_b_all, ctx->scratch, b,
                       ctx->cfg.max_streams, ctx->cap, ctx->conv_pos);
    LAUNCHCHK("gather_cnn_cache");
    HIPCHK(hipMemcpyAsync(dst_host, ctx->scratch, (size_t)L * D * RNNT_LORDER * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return RNNT_OK;
}

int rnnt_get_predictor_state(rnnt_ctx* ctx, int32_t b, float* h_host, float* c_host, int32_t* last_token, void* stream) {
    if (!ctx || b < 0 || b >= ctx->n_streams) return fail(ctx, RNNT_ERR_ARG, "rnnt_get_predictor_state: bad stream index");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(ctx->pinned + 4, ctx->sel + b, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const size_t bo = (size_t)(ctx->pinned[4] & 1) * ctx->cfg.max_streams * D;   // committed buffer of this stream
    if (h_host) HIPCHK(hipMemcpyAsync(h_host, ctx->h + bo + (size_t)b * D, D * sizeof(float), hipMemcpyDeviceToHost, s));
    if (c_host) HIPCHK(hipMemcpyAsync(c_host, ctx->c + bo + (size_t)b * D, D * sizeof(float), hipMemcpyDeviceToHost, s));
    if (last_token) HIPCHK(hipMemcpyAsync(last_token, ctx->tok + b, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return RNNT_OK;
}

int rnnt_get_enc_frames(rnnt_ctx* ctx, float* dst_host, int32_t* frames_out, void* stream) {
    if (!ctx) return RNNT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nf = ctx->frames_buffered;
    if (frames_out) *frames_out = nf;
    if (!dst_host || nf == 0) return RNNT_OK;
    HIPCHK(hipMemcpy2DAsync(dst_host, (size_t)nf * D * sizeof(float), ctx->encbuf, (size_t)ctx->fstride * D * sizeof(float),
                            (size_t)nf * D * sizeof(float), ctx->n_streams, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return RNNT_OK;
}

const float* rnnt_enc_frames_dev(rnnt_ctx* ctx, int32_t* frames_out, int32_t* stride_frames) {
    if (!ctx) return nullptr;
    if (frames_out) *frames_out = ctx->frames_buffered;
    if (stride_frames) *stride_frames = ctx->fstride;
    return ctx->encbuf;
}

int rnnt_profile_begin(rnnt_ctx* ctx, int32_t tag) {
    if (!ctx) return RNNT_ERR_ARG;
    ctx->prof_tag = tag;
    ctx->prof_used = 0;
    return RNNT_OK;
}

int rnnt_profile_end(rnnt_ctx* ctx, double* total_ms, int64_t* n_launches) {
    if (!ctx) return RNNT_ERR_ARG;
    double tot = 0;
    for (size_t i = 0; i + 1 < ctx->prof_used; i += 2) {
        HIPCHK(hipEventSynchronize(ctx->prof_ev[i + 1]));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, ctx->prof_ev[i], ctx->prof_ev[i + 1]));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (n_launches) *n_launches = (int64_t)(ctx->prof_used / 2);
    ctx->prof_tag = -1;
    ctx->prof_used = 0;
    return RNNT_OK;
}

int rnnt_get_counters(rnnt_ctx* ctx, int64_t* launches, int64_t* greedy_steps) {
    if (!ctx) return RNNT_ERR_ARG;
    if (launches) *launches = ctx->launches;
    if (greedy_steps) *greedy_steps = ctx->greedy_steps;
    return RNNT_OK;
}

}  // extern "C"
