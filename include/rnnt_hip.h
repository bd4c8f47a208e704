/*
 * rnnt_hip.h — C ABI of librnnt_hip.so: the MI355X (gfx950) streaming RNN-Transducer
 * inference path (chunked Conformer encoder + LSTM predictor + joint + greedy/beam step kernels).
 *
 * The reference (CentaureaHO/CTC-VR) has no FFI/plugin layer: its boundary for this path is the
 * Python class OnlineRNNTModel (model/online_rnnt_model.py:58) and, one level below, WeNet's
 * step API forward_encoder_chunk / forward_predictor_step / forward_joint_step
 * (wenet/transducer/transducer.py:444-472).  Each entry point below names the reference
 * function it replaces.  The Python facade in ctc-vr_amd/online_rnnt_model.py binds these
 * symbols with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions: extern "C"; plain pointers and sizes; every function returns 0 on success or a
 * negative rnnt_status; rnnt_last_error(ctx) returns a static/ctx-owned message; no exceptions
 * cross the ABI.  The caller owns every buffer it passes; the library owns what it allocates
 * inside a context.  "dev" pointers are HIP device pointers on the context's device; "host"
 * pointers are ordinary host memory.  A context is confined to one host thread at a time; all
 * work of one call is enqueued on the hipStream_t passed as `stream` (void*; NULL = default
 * stream).  Calls that return data to the host synchronise that stream.
 * All streams of one context advance in lock step (same chunk length per call).
 */
#ifndef RNNT_HIP_H
#define RNNT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rnnt_ctx rnnt_ctx;

typedef enum {
    RNNT_OK = 0,
    RNNT_ERR_ARG = -1,      /* bad argument / unknown tensor name                         */
    RNNT_ERR_SHAPE = -2,    /* tensor shape or capacity mismatch                          */
    RNNT_ERR_OOM = -3,      /* hipMalloc failed                                           */
    RNNT_ERR_HIP = -4,      /* a HIP runtime call or kernel launch failed                 */
    RNNT_ERR_STATE = -5     /* call sequence error (weights not finalized, no streams...) */
} rnnt_status;

/* numerics modes for rnnt_finalize_weights: how the dense contractions of the encoder and of the joint lattice
 * (positionwise_feed_forward.py:50-58, attention.py:109-131, subsampling.py:188-193, convolution.py:138-148,
 * model/component/joint.py:62-69) are evaluated.  Storage is fp32 in every mode; LayerNorm, softmax, depthwise conv,
 * the LSTM predictor and the greedy/beam decode arithmetic are fp32 in every mode. */
#define RNNT_NUMERICS_FP32   0 /* exact-f32 MFMA (v_mfma_f32_16x16x4_f32, a k-ordered fmaf chain): default parity mode     */
#define RNNT_NUMERICS_BF16X3 1 /* split bf16: x = hi + lo, hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16, fp32        */
                               /* accumulate; ~1e-5 relative per product; parity-gated (tokens exact, logits <= 1e-3)      */
#define RNNT_NUMERICS_BF16   2 /* plain bf16 operands, fp32 accumulate: perf mode, token-match rate reported, no parity   */
#define RNNT_NUMERICS_F16X3  3 /* split f16 (11-bit planes): ~5e-7 relative per product, same cost as bf16x3; operands     */
                               /* must stay below 65504 in magnitude (LayerNorm / activation outputs and weights do)       */

typedef struct {
    int32_t max_streams;       /* B: lock-stepped streams held by the context                     */
    int32_t max_chunk_frames;  /* largest fbank chunk (input frames) passed to rnnt_encoder_chunk  */
    int32_t max_cache_frames;  /* K/V cache capacity per stream, in encoder frames (<= 5000)       */
    int32_t max_enc_frames;    /* encoder-output frame buffer per stream (frames awaiting decode)  */
    int32_t max_tokens;        /* token buffer per stream                                          */
    int32_t vocab_size;        /* 412 for the reference tokenizer (tokenizer/tokenizer.py:53-60)   */
    int32_t blank_id;          /* 5                                                                */
    int32_t n_steps;           /* max symbols per encoder frame (online_rnnt_model.py:174) = 10    */
    int32_t device;            /* HIP device ordinal                                               */
    int32_t max_beam;          /* largest beam size for rnnt_beam_frame (0 = beam search unused)   */
} rnnt_config;

/* -- lifetime ------------------------------------------------------------------------------ */
/* replaces OnlineRNNTModel.__init__ (model/online_rnnt_model.py:58-143): fixed architecture
 * (12 Conformer blocks, D=256, H=4, FFN=1024, conv2d/4, rel_pos, causal dw k=31 + BatchNorm,
 * Embedding+LSTM(256) predictor, add/tanh joint). */
int rnnt_create(const rnnt_config* cfg, rnnt_ctx** out);
void rnnt_destroy(rnnt_ctx* ctx);
const char* rnnt_last_error(const rnnt_ctx* ctx);
int rnnt_abi_version(void);

/* -- weights --------------------------------------------------------------------------------- */
/* replaces load_state_dict(checkpoint['model']) (online_rnnt_decode.py:49-50): one call per
 * entry of the reference's 504-key state dict (names as in SURVEY.md §8b), float32 host data
 * (num_batches_tracked entries are accepted and ignored). */
int rnnt_load_tensor(rnnt_ctx* ctx, const char* name, const float* host_data, int32_t ndim,
                     const int64_t* dims);
/* The whole state dict in ONE call from a flat float32 blob (the packed blob of the multi-GPU weight broadcast, SURVEY.md §8e:
 * rank 0 broadcasts it over RCCL, every rank hands its device copy to its context).  blob: n_floats floats, on the context's
 * device (on_device != 0) or on the host; tensor i is `names[i]` with `ndims[i]` dimensions taken in order from dims_flat and
 * starts where tensor i-1 ends.  Equivalent to n_tensors calls of rnnt_load_tensor (one device-to-host copy instead of none:
 * the packing below needs the values on the host).  Fails with RNNT_ERR_SHAPE if the table does not cover exactly n_floats. */
int rnnt_load_packed(rnnt_ctx* ctx, const float* blob, int64_t n_floats, int32_t on_device, int32_t n_tensors,
                     const char* const* names, const int32_t* ndims, const int64_t* dims_flat);
/* packs weights for the kernels (BatchNorm fold, linear_pos table pe*W_pos^T, LSTM input table,
 * GLU/LSTM row interleave, conv2 channels-last) and uploads them.  Fails with RNNT_ERR_STATE if
 * any required tensor is missing. */
int rnnt_finalize_weights(rnnt_ctx* ctx, int32_t numerics_mode, void* stream);

/* -- stream state ------------------------------------------------------------------------------ */
/* replaces OnlineRNNTModel.reset_streaming_cache (model/online_rnnt_model.py:145-164) for
 * n_streams lock-stepped streams: empty K/V cache, zero conv left-context, zero LSTM state,
 * last token = blank, no beam, frame buffer empty. */
int rnnt_streams_reset(rnnt_ctx* ctx, int32_t n_streams, void* stream);

/* replaces encoder.forward_chunk(xs, offset, required_cache_size, att_cache, cnn_cache)
 * (wenet/transformer/encoder.py:203-299) as called from _decode_chunk_streaming_logic
 * (model/online_rnnt_model.py:175-181), for all streams at once.
 *   fbank_dev  [n_streams, chunk_frames, 80] float32, device
 *   offset, required_cache_size: the reference's arguments (estimated encoder offset; see
 *       model/online_rnnt_model.py:364-370).
 * K/V and conv caches live in the context.  The t' = ((T-3)/2+1-3)/2+1 output frames of every
 * stream are appended to the context's encoder-frame buffer; *frames_out receives t'. */
int rnnt_encoder_chunk(rnnt_ctx* ctx, const float* fbank_dev, int32_t chunk_frames, int32_t offset,
                       int32_t required_cache_size, int32_t* frames_out, void* stream);

/* replaces the whole chunk loop around forward_chunk (online_rnnt_decode.py:87-117, or streaming_inference,
 * model/online_rnnt_model.py:311-342) for utterances that are fully available: chunk c of every stream covers fbank
 * frames [chunk_start[c], chunk_start[c]+chunk_len[c]) and is encoded with (offsets[c], required[c]) exactly as
 * n_chunks calls of rnnt_encoder_chunk would (same float32 arithmetic up to summation order).  Schedule: layer-major -- each
 * of the 12 blocks runs over all chunks at once (one GEMM over streams x frames rows per contraction; every query keeps its
 * own chunk's key window and positional window); plans with a cache reset in the middle of the call take the wavefront over
 * (chunk, layer) instead.  fbank_dev [n_streams, total_frames, 80]; host int arrays.
 * greedy != 0: the greedy decode of rnnt_greedy_decode follows on the same stream; the call returns with all frames
 * decoded (synchronises).  Two contexts driven by two host threads overlap one call's decode with the other's encoder. */
int rnnt_encoder_chunks(rnnt_ctx* ctx, const float* fbank_dev, int32_t total_frames, int32_t n_chunks,
                        const int32_t* chunk_start, const int32_t* chunk_len, const int32_t* offsets,
                        const int32_t* required, int32_t greedy, int32_t* frames_out, void* stream);

/* replaces the greedy loops of _decode_chunk_streaming_logic (model/online_rnnt_model.py:183-222)
 * over every buffered encoder frame not yet decoded, all streams in parallel, state carried in
 * the context (LSTM [h,c], last token).  Appends to the per-stream token buffers.  Synchronises. */
int rnnt_greedy_decode(rnnt_ctx* ctx, void* stream);

/* token buffers: counts_host[n_streams] total tokens so far; tokens_host [n_streams, max_tokens]
 * (row-major, int32).  Either pointer may be NULL.  Synchronises. */
int rnnt_get_tokens(rnnt_ctx* ctx, int32_t* counts_host, int32_t* tokens_host, void* stream);

/* drop decoded frames from the encoder-frame buffer (keeps undecoded ones). */
int rnnt_frames_consume(rnnt_ctx* ctx, void* stream);

/* Greedy decode of a padded batch of whole utterances of DIFFERENT lengths in one call (utils/utils.py:29-50 pads a batch,
 * online_rnnt_eval.py:86-94 decodes every utterance with its own audio_lens): stream b runs the decode script's chunk loop
 * (online_rnnt_decode.py:81-117) over its own lens_host[b] frames of fbank_dev [n_streams, total_frames, 80]; tokens / counts are
 * read with the usual getters and equal a B = 1 run of that utterance.  Needs freshly reset streams (rnnt_streams_reset) and leaves
 * them finished (reset before the next call).  Utterances of fewer than 7 frames give no tokens; an utterance that is a single
 * chunk (at least 7 but fewer than chunk_frames + max(16, chunk_frames) frames) is refused with RNNT_ERR_SHAPE -- run those through
 * rnnt_encoder_chunks.  frames_out [n_streams] (optional, host): encoder frames per stream. */
int rnnt_decode_ragged(rnnt_ctx* ctx, const float* fbank_dev, int32_t total_frames, const int32_t* lens_host, int32_t chunk_frames,
                       int32_t* frames_out, void* stream);

/* -- beam search: device half of _decode_chunk_beam_search (model/online_rnnt_model.py:419-503) -- */
/* One encoder frame, all live hypotheses ("rows") of all streams at once.  For every row the library runs the
 * reference's greedy extension chain (<= n_steps evaluations: predictor step, joint, log_softmax, blank
 * log-prob, top-beam_k non-blank, stop when blank >= max - 1e-6, else extend with the best non-blank) and
 * keeps every intermediate LSTM state in a pool.  The caller (host) owns the hypothesis bookkeeping — token
 * lists, Python-double scores, stable sort, first-wins de-dup (:505-518) — and then tells the library which
 * pooled state each surviving hypothesis keeps.
 *   frame_idx            index into the buffered encoder frames
 *   row_stream_host[n]   stream of each row;  row_tok_host[n] predictor input token (last token or blank)
 *   steps_host[n]        evaluations done per row
 *   blank_lp_host [n][n_steps], top_lp_host / top_tok_host [n][n_steps][beam_k]
 * Row r's state before evaluation s is pool slot (r, s); the state after consuming the input token of
 * evaluation s is slot (r, s+1).  Synchronises. */
int rnnt_beam_frame(rnnt_ctx* ctx, int32_t frame_idx, int32_t n_rows, const int32_t* row_stream_host,
                    const int32_t* row_tok_host, int32_t beam_k, int32_t* steps_host, float* blank_lp_host,
                    float* top_lp_host, int32_t* top_tok_host, void* stream);
/* new row r takes pool slot (src_row_host[r], src_step_host[r]); rows are renumbered 0..n_new-1. */
int rnnt_beam_select(rnnt_ctx* ctx, int32_t n_new, const int32_t* src_row_host, const int32_t* src_step_host, void* stream);
/* current [h,c] of rows 0..n_rows-1: h_host, c_host [n_rows,256].  Synchronises. */
int rnnt_beam_get_states(rnnt_ctx* ctx, int32_t n_rows, float* h_host, float* c_host, void* stream);

/* Beam search with the bookkeeping inside the library: the per-frame loop of _decode_chunk_beam_search
 * (model/online_rnnt_model.py:419-518) over the buffered frames [frame_begin, frame_end) of every stream -- extension
 * chains on the device (one resident workgroup per hypothesis), candidate order / double-precision scores / stable
 * sort / first-wins de-duplication on the host in C++, state pool gather.  rnnt_streams_reset starts every stream with
 * one empty hypothesis (:407-415).  Results: rnnt_beam_hyp_count / rnnt_beam_get_hyp (tokens_host may be NULL to query
 * the length); hypothesis i of stream b is device row (hypotheses of streams < b) + i for rnnt_beam_get_states. */
int rnnt_beam_advance(rnnt_ctx* ctx, int32_t frame_begin, int32_t frame_end, int32_t beam_size, void* stream);
int rnnt_beam_hyp_count(rnnt_ctx* ctx, int32_t stream_idx, int32_t* n_out);
int rnnt_beam_get_hyp(rnnt_ctx* ctx, int32_t stream_idx, int32_t hyp_idx, int32_t cap, int32_t* tokens_host, int32_t* n_tokens,
                      double* log_prob);
/* The host half of one frame for one stream as a pure function (no context, no GPU; CPU tests): flat hypotheses in,
 * flat survivors out, returns their number.  Same code rnnt_beam_advance runs. */
int rnnt_beam_merge_host(int32_t n_hyp, const int32_t* hyp_len, const int32_t* hyp_tokens, const double* hyp_score,
                         const int32_t* steps, const float* blank_lp, const float* top_lp, const int32_t* top_tok,
                         int32_t n_steps, int32_t k, int32_t beam_size, int32_t* out_len, int32_t* out_tokens,
                         double* out_score, int32_t* out_src_row, int32_t* out_src_step);
/* drop all buffered encoder frames (beam path; the greedy path uses rnnt_frames_consume). */
int rnnt_frames_discard(rnnt_ctx* ctx, void* stream);

/* -- step API (WeNet export precedent, wenet/transducer/transducer.py:444-472) ---------------- */
/* forward_predictor_step (wenet/transducer/predictor.py:185-210): tokens_dev int32 [rows],
 * h/c in/out float32 [rows,256] device; out_dev [rows,256]. */
int rnnt_predictor_step(rnnt_ctx* ctx, const int32_t* tokens_dev, const float* h_in_dev,
                        const float* c_in_dev, int32_t rows, float* out_dev, float* h_out_dev,
                        float* c_out_dev, void* stream);
/* TransducerJoint.forward (model/component/joint.py:48-69), lattice form:
 * enc_dev [B,T,256], pred_dev [B,U,256] -> logits_dev [B,T,U,vocab]; mode 0 = raw logits,
 * 1 = log_softmax over the vocabulary (online_rnnt_model.py:446-447). */
int rnnt_joint(rnnt_ctx* ctx, const float* enc_dev, const float* pred_dev, int32_t B, int32_t T,
               int32_t U, int32_t mode, float* logits_dev, void* stream);

/* replaces BaseEncoder.forward(xs, lens, decoding_chunk_size=-1) (full context,
 * wenet/transformer/encoder.py:121-180).  fbank_dev [B,T,80], lens_host[B];
 * out_dev [B,T',256]; *frames_out = T'.  Runs as the layer-major schedule with ONE chunk of T frames (all valid keys at
 * positional window 0, per-stream padding mask); invalidates the streaming state. */
int rnnt_encoder_full(rnnt_ctx* ctx, const float* fbank_dev, const int32_t* lens_host, int32_t B,
                      int32_t T, float* out_dev, int32_t* frames_out, void* stream);

/* CTC head on the same encoder: per-frame argmax of OnlineCTC.ctc_lo over the full-context encoder
 * output (model/online_rnnt_model.py:37-38,655-658).  ids_host [B, T'] int32; the repeat/blank collapse (:660-671)
 * is host code.  Needs ctc_head.ctc_lo.{weight,bias} among the loaded tensors. */
int rnnt_ctc_argmax(rnnt_ctx* ctx, const float* fbank_dev, const int32_t* lens_host, int32_t B, int32_t T,
                    int32_t* ids_host, int32_t* frames_out, void* stream);

/* OnlineCTC.log_softmax (model/online_rnnt_model.py:34-35) on encoder frames already on the device: out_dev [rows, vocab] =
 * log_softmax(ctc_lo(enc_dev [rows, 256])).  The CTC term of the WeNet prefix beam search (wenet/transducer/search/
 * prefix_beam_search.py:66,99-101), whose frame loop runs in the facade over rnnt_encoder_full / rnnt_predictor_step / rnnt_joint. */
int rnnt_ctc_logprobs(rnnt_ctx* ctx, const float* enc_dev, int32_t rows, float* out_dev, void* stream);

/* Offline greedy search: basic_greedy_search (model/component/transducer.py:22-70) behind
 * OnlineRNNTModel.forward(audios, audio_lens) of a non-streaming model (model/online_rnnt_model.py:234-235,268):
 * full-context encoder + per-utterance greedy loop over its valid frames, <= n_steps symbols per frame (reference
 * default 64).  counts_host [B]; tokens_host [B, max_tokens] (may be NULL).  Streaming state is clobbered. */
int rnnt_greedy_search_full(rnnt_ctx* ctx, const float* fbank_dev, const int32_t* lens_host, int32_t B, int32_t T,
                            int32_t n_steps, int32_t* counts_host, int32_t* tokens_host, void* stream);

/* Feature front-end on the device: replaces extract_audio_features (data/dataloader.py:15-41) =
 * torchaudio MelSpectrogram(sample_rate, n_fft, n_mels=80, hop_length=512, hamming window, power 2, centred reflect
 * padding, HTK mel scale) + AmplitudeToDB().  wave_dev [B, n_samples] mono float32 on the device ->
 * out_dev [B, 1 + n_samples/512, 80] (the layout rnnt_encoder_chunk takes).  Independent of the model weights. */
int rnnt_fbank(rnnt_ctx* ctx, const float* wave_dev, int32_t B, int32_t n_samples, int32_t sample_rate, int32_t n_fft,
               float* out_dev, int32_t* frames_out, void* stream);

/* -- state read-back in the reference's layouts (parity tests, facade attributes) -------------- */
/* streaming_att_cache of one stream: [12, 4, len, 128] (K = [...,:64], V = [...,64:],
 * wenet/transformer/encoder.py:284); *len_out = cached frames.  dst_host may be NULL to query len. */
int rnnt_get_att_cache(rnnt_ctx* ctx, int32_t stream_idx, float* dst_host, int32_t* len_out, void* stream);
/* streaming_cnn_cache of one stream: [12, 1, 256, 30] (wenet/transformer/convolution.py:130). */
int rnnt_get_cnn_cache(rnnt_ctx* ctx, int32_t stream_idx, float* dst_host, void* stream);
/* predictor state [h,c] each [256] and last token of one stream. */
int rnnt_get_predictor_state(rnnt_ctx* ctx, int32_t stream_idx, float* h_host, float* c_host,
                             int32_t* last_token, void* stream);
/* buffered encoder frames [n_streams, frames, 256] starting at the oldest buffered frame. */
int rnnt_get_enc_frames(rnnt_ctx* ctx, float* dst_host, int32_t* frames_out, void* stream);
/* device pointer of the encoder-frame buffer [n_streams, max_enc_frames, 256] (borrowed). */
const float* rnnt_enc_frames_dev(rnnt_ctx* ctx, int32_t* frames_out, int32_t* stride_frames);

/* per-launch-site timing with HIP events recorded on the launch stream (bench.py roofline leg).
 * tag selects ONE launch site: 1 conv1, 2 conv2 (implicit GEMM), 3 embed linear, 4 FFN w_1, 5 FFN w_2, 6 QKV,
 * 7 attention, 8 attention out-proj, 9 pointwise_conv1+GLU, 10 depthwise conv, 11 pointwise_conv2, 13 joint enc
 * projection, 20 LSTM cell, 21 predictor projection, 22 joint pred_ffn+tanh, 23 joint ffn_out.
 * rnnt_profile_end synchronises the recorded events and returns the summed kernel time and launch count. */
int rnnt_profile_begin(rnnt_ctx* ctx, int32_t tag);
int rnnt_profile_end(rnnt_ctx* ctx, double* total_ms, int64_t* n_launches);

/* counters for bench/roofline: number of kernel launches and greedy steps since the last reset. */
int rnnt_get_counters(rnnt_ctx* ctx, int64_t* launches, int64_t* greedy_steps);

#ifdef __cplusplus
}
#endif
#endif /* RNNT_HIP_H */
