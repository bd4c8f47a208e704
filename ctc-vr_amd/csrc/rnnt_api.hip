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
    unsigned short *blob_hi = nullptr, *blob_lo = nullptr;   // 16-bit hi / lo planes of the blob (split-operand numerics modes)
    // fused Conformer-block kernels (rnnt_fused.hip.h): fragment-major packed layer weights of the current numerics mode and
    // the per-layer pointer table in device memory
    uint4* fuse_w = nullptr;
    size_t fuse_w_vecs = 0;
    LayerDev* layers_dev = nullptr;
    const uint4* conv2_wp = nullptr;          // fragment-major packed conv2 weights of the current numerics mode (gemm_bw)
    unsigned char* joint_wfrag = nullptr;     // joint.ffn_out as the LDS-DMA ring's stage stream (pack_joint_w), split modes and bf16
    size_t joint_wfrag_bytes = 0;
    int* joint_counter = nullptr;             // joint_lattice_rows' dynamic row-tile queue (zeroed before every launch)
    std::map<const void*, int> dyn_lds;       // kernels whose dynamic-LDS limit was raised for THIS context's device (ensure_dyn_lds)
    std::vector<LayerDev> layers_host;        // host copy (packed-weight pointers for gemm_as / ffn_as launches)
    int use_as = 1;                            // RNNT_AS=0: LDS-tiled gemm_bf for every layer contraction of the layer-major schedule
    int use_fused = 1;                         // RNNT_FUSED=0: the unfused wavefront (11 launches per stage)
    FuseItem* wf_ftab = nullptr;
    size_t wf_fcap = 0;
    int wf_fused_plan = 0;                     // the cached plan (wf_key) was built for the fused schedule
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
    int use_multi = 1;         // RNNT_DEC_MULTI=0: one CU per stream (greedy_stream) instead of greedy_multi (4 CUs per stream)
    int n_cus = 0;
    unsigned long long *gm_x1 = nullptr, *gm_xa = nullptr;   // greedy_multi mailboxes
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
    int wf_groups = 1, wf_sub_async = 1;       // RNNT_WF_GROUPS (1..4; default 1: see api_encoder.hip.inc), RNNT_WF_SUB_ASYNC
    int wf_merge = 2;                          // RNNT_WF_MERGE (1..WF_MERGE_MAX): chunks of one layer per wavefront stage
    // descriptor tables of the last rnnt_encoder_chunks call, reused when the next call has the same plan and entry state
    struct WfLaunch { int type, off, n, maxM, maxT2; };   // type 0..7 gemm (ffn1m ffn2m qkv out pw1 pw2 ffn1 ffn2), 10 attn, 11 dw, 12 ln
    std::vector<WfLaunch> wf_seq;
    std::vector<std::array<int, 13>> wf_lstart;
    std::vector<int> wf_sc_first, wf_key;
    std::vector<hipEvent_t> ev_pool;
    // layer-major schedule (host_lm.hip.inc): activations over all B*F rows of a call, per-layer linear post-GLU rows, one
    // subsampling slab, attention block table (reused while the plan and the entry state stay the same)
    int use_lm = 1;                            // RNNT_LM=0: wavefront schedule for every whole-utterance call
    int lm_pw2_head = 2;                       // RNNT_LM_PW2_HEAD: 2 depthwise conv + pointwise_conv2 inside the FFN launch, 1 only pointwise_conv2, 0 three launches
    int lm_ffn_merge = 0;                      // RNNT_LM_FFN_MERGE=1: one launch per layer boundary (better alone, worse with two batches in flight)
    int lm_qkv_tail = 1, lm_out_chain = 1;     // RNNT_LM_QKV_TAIL=0 / RNNT_LM_OUT_CHAIN=0: q/k/v and pointwise_conv1 as launches of their own
    int lm_side = 1;                           // RNNT_LM_SIDE=0: the tail chunk class's subsampling in line instead of on the side stream
    float *lm_x = nullptr, *lm_h = nullptr, *lm_q = nullptr, *lm_a = nullptr, *lm_d = nullptr, *lm_g = nullptr, *lm_y1 = nullptr, *lm_y2 = nullptr;
    size_t lm_y1_cap = 0, lm_y2_cap = 0, lm_blocks_cap = 0;
    float *lm_y1b = nullptr, *lm_y2b = nullptr;      // slabs of the tail chunk class (subsampled on sub_stream beside the main class)
    size_t lm_y1b_cap = 0, lm_y2b_cap = 0;
    hipEvent_t sub_ev[2] = {nullptr, nullptr};     // fork / join of the tail class on sub_stream
    // ragged batches (rnnt_decode_ragged): gathered tail frames, their subsampled rows, gather / scatter entries
    float *rg_fb = nullptr, *rg_xt = nullptr;
    int2* rg_ent = nullptr;
    size_t rg_fb_cap = 0, rg_xt_cap = 0, rg_ent_cap = 0;
    LmBlock* lm_blocks = nullptr;
    std::vector<LmBlock> lm_blocks_host;
    std::vector<int> lm_key;
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

#include "host_launch.hip.inc"
#include "host_lm.hip.inc"

extern "C" {
#include "api_lifecycle.hip.inc"
#include "api_encoder.hip.inc"
#include "api_decode.hip.inc"
#include "api_beam.hip.inc"
#include "api_ops.hip.inc"
#include "api_state.hip.inc"
}  // extern "C"
