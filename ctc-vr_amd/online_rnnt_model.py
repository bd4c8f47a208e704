"""Host-side mirror of the reference's OnlineRNNTModel (model/online_rnnt_model.py:58-671) over the
HIP library.  Same constructor keywords, method names, argument meaning, return shapes and error
behaviour for the streaming inference path; the arithmetic runs in librnnt_hip.so (gfx950 kernels),
never on the CPU.  PyTorch is used only for device memory and streams.

Additions over the reference (which is batch-1 only, :277-278,348-349): `StreamingBatch`, B lock-stepped
independent streams in one context; each stream's result equals the reference's B=1 result.
"""
from typing import List, Optional, Tuple

import numpy as np
import torch

from .lib import RnntEngine, RnntError


def _stream_ptr():
    return torch.cuda.current_stream().cuda_stream


class BeamHypothesis:
    """model/online_rnnt_model.py:41-55."""

    def __init__(self, tokens: List[int], log_prob: float, predictor_states=None):
        self.tokens = tokens
        self.log_prob = log_prob
        self.predictor_states = predictor_states

    def __lt__(self, other):
        return self.log_prob < other.log_prob

    def copy(self):
        return BeamHypothesis(tokens=self.tokens.copy(), log_prob=self.log_prob, predictor_states=self.predictor_states)


def beam_advance_frame(engine, frame_idx, beams, blank_id, beam_size, stream=None):
    """Host half of one encoder frame of _decode_chunk_beam_search (model/online_rnnt_model.py:419-518) for a
    list of per-stream beams (each a list of BeamHypothesis whose device rows are numbered in list order,
    streams concatenated).  The device evaluates every hypothesis' extension chain (rnnt_beam_frame); here the
    candidates are rebuilt in the reference's order (per hypothesis, per step: blank candidate, then the top-k
    non-blank), scored in Python floats (double), sorted stably in descending order (:506), de-duplicated
    first-wins on the token tuple (:508-516) and truncated to the beam."""
    row_stream, row_tok = [], []
    for b, beam in enumerate(beams):
        for h in beam:
            row_stream.append(b)
            row_tok.append(h.tokens[-1] if h.tokens else blank_id)          # :429
    k = min(beam_size, engine.cfg.vocab_size - 1)                            # :467
    steps, blank_lp, top_lp, top_tok = engine.beam_frame(frame_idx, row_stream, row_tok, k, stream)
    new_beams, src_row, src_step = [], [], []
    r = 0
    for b, beam in enumerate(beams):
        cands = []
        for h in beam:
            toks, lp = list(h.tokens), h.log_prob
            n = int(steps[r])
            for st in range(n):
                cands.append((toks.copy(), lp + float(blank_lp[r, st]), r, st))                 # blank: old state
                for j in range(k):
                    cands.append((toks + [int(top_tok[r, st, j])], lp + float(top_lp[r, st, j]), r, st + 1))
                if st < n - 1:                                                                   # chain continued (:489-499)
                    toks.append(int(top_tok[r, st, 0]))
                    lp += float(top_lp[r, st, 0])
            r += 1
        cands.sort(key=lambda c: c[1], reverse=True)
        uniq, seen = [], set()
        for c in cands:
            t = tuple(c[0])
            if t not in seen:
                uniq.append(c)
                seen.add(t)
                if len(uniq) >= beam_size:
                    break
        uniq = uniq[:beam_size]
        new_beams.append([BeamHypothesis(c[0], c[1]) for c in uniq])
        src_row.extend(c[2] for c in uniq)
        src_step.extend(c[3] for c in uniq)
    engine.beam_select(src_row, src_step, stream)
    return new_beams


class _EncoderView:
    """Attribute surface the reference's callers read: encoder.static_chunk_size and
    encoder.embed.subsampling_rate (model/online_rnnt_model.py:283-287)."""

    class _Embed:
        subsampling_rate = 4
        right_context = 6

    def __init__(self, static_chunk_size):
        self.static_chunk_size = static_chunk_size
        self.embed = self._Embed()


class OnlineRNNTModel:
    """Drop-in for the streaming-inference surface of the reference class of the same name."""

    def __init__(self, input_dim: int = 80, hidden_dim: int = 256, vocab_size: int = 4336, blank_id: int = 0,
                 streaming: bool = True, static_chunk_size: int = 32, use_dynamic_chunk: bool = True,
                 ctc_weight: float = 0.3, predictor_layers: int = 1, predictor_dropout: float = 0.1,
                 ctc_dropout_rate: float = 0.1, rnnt_loss_clamp: float = -1.0, ignore_id: int = -1,
                 # engine sizing (not in the reference)
                 max_streams: int = 1, max_chunk_frames: int = 256, max_cache_frames: int = 1024,
                 max_enc_frames: int = 1024, max_tokens: int = 8192, device: int = 0, max_beam: int = 8, numerics=None):
        if input_dim != 80 or hidden_dim != 256 or predictor_layers != 1:
            raise ValueError("the HIP path implements the reference's configured architecture: input_dim=80, hidden_dim=256, predictor_layers=1")
        self.blank_id = blank_id
        self.vocab_size = vocab_size
        self.streaming = streaming
        self.ctc_weight = ctc_weight
        self.ignore_id = ignore_id
        self.rnnt_loss_clamp = rnnt_loss_clamp
        self.encoder_output_size = hidden_dim
        self.encoder = _EncoderView(static_chunk_size if streaming else 0)
        self.device = torch.device("cuda", device)
        self._engine = RnntEngine(max_streams=max_streams, max_chunk_frames=max_chunk_frames, max_cache_frames=max_cache_frames,
                                  max_enc_frames=max_enc_frames, max_tokens=max_tokens, vocab_size=vocab_size, blank_id=blank_id,
                                  n_steps=10, device=device, max_beam=max_beam)
        self.numerics = numerics          # None -> $RNNT_NUMERICS or "fp32" (lib.numerics_id)
        self._loaded = False
        self._chunks_done = None          # None = reset_streaming_cache not called yet (attributes are None, :138-143)
        self._tok_count = 0
        self.streaming_beam_hypotheses = None
        self._global_encoder_offset = 0

    # ---- nn.Module-like plumbing ------------------------------------------------------------------
    def eval(self):
        return self

    def to(self, device):
        return self

    def load_state_dict(self, state_dict, strict: bool = True):
        self._engine.load_state_dict(state_dict, numerics=self.numerics)
        self._loaded = True

    # ---- streaming state (model/online_rnnt_model.py:138-164) ----------------------------------------
    def extract_audio_features(self, waveform, sample_rate, n_fft=1024):
        """Device version of data/dataloader.py:extract_audio_features: waveform [n] or [B, n] -> [.., 1 + n // 512, 80] dB
        mel features on this model's GPU (ctc_vr_amd.features)."""
        from .features import extract_audio_features
        return extract_audio_features(self._engine, waveform, sample_rate, n_fft=n_fft)

    def reset_streaming_cache(self, device=None):
        self._require_loaded()
        self._engine.reset(1, _stream_ptr())
        self._chunks_done = 0
        self._tok_count = 0
        self.streaming_beam_hypotheses = None
        self._global_encoder_offset = 0

    @property
    def streaming_att_cache(self) -> Optional[torch.Tensor]:
        if self._chunks_done is None:
            return None
        if self._chunks_done == 0:
            return torch.zeros((0, 0, 0, 0), device=self.device)
        return torch.from_numpy(self._engine.att_cache(0, _stream_ptr())).to(self.device)

    @property
    def streaming_cnn_cache(self) -> Optional[torch.Tensor]:
        if self._chunks_done is None:
            return None
        if self._chunks_done == 0:
            return torch.zeros((0, 0, 0, 0), device=self.device)
        return torch.from_numpy(self._engine.cnn_cache(0, _stream_ptr())).to(self.device)

    @property
    def streaming_predictor_states(self) -> Optional[List[torch.Tensor]]:
        if not self._chunks_done:
            return None
        h, c, _ = self._engine.predictor_state(0, _stream_ptr())
        return [torch.from_numpy(h).view(1, 1, 256).to(self.device), torch.from_numpy(c).view(1, 1, 256).to(self.device)]

    @property
    def streaming_last_emitted_token(self) -> int:
        if not self._chunks_done:
            return self.blank_id
        return self._engine.predictor_state(0, _stream_ptr())[2]

    def _require_loaded(self):
        if not self._loaded:
            raise RnntError("load_state_dict() must be called before streaming")

    # ---- one chunk, greedy (model/online_rnnt_model.py:166-222, 346-387) -----------------------------
    def _decode_chunk_streaming_logic(self, chunk_xs: torch.Tensor, offset: int, required_cache_size: int) -> List[int]:
        x = chunk_xs.to(self.device, torch.float32).contiguous()
        s = _stream_ptr()
        self._engine.encoder_chunk(x.data_ptr(), x.size(1), offset, required_cache_size, s)
        self._engine.greedy_decode(s)
        toks = self._engine.tokens(s)[0]
        new = toks[self._tok_count:]
        self._tok_count = len(toks)
        self._engine.frames_consume(s)
        self._chunks_done += 1
        return new

    def process_single_chunk(self, chunk_audio: torch.Tensor, chunk_len: torch.Tensor) -> Tuple[List[int], None, None]:
        assert self.streaming, "Model is not in streaming mode for process_single_chunk."
        assert chunk_audio.size(0) == 1, "Single chunk processing currently supports batch size 1 only."
        if self._chunks_done is None:
            self.reset_streaming_cache()
        if chunk_audio.size(1) < 7:
            print(f"Warning: Chunk too small ({chunk_audio.size(1)} frames), skipping")
            return [], None, None
        subsampling_rate = self.encoder.embed.subsampling_rate
        off = self._global_encoder_offset
        hyp = self._decode_chunk_streaming_logic(chunk_audio, off, off)
        self._global_encoder_offset += chunk_audio.size(1) // subsampling_rate
        return hyp, None, None

    # ---- whole utterance, greedy (model/online_rnnt_model.py:274-344) --------------------------------
    def _utterance_chunks(self, n_frames: int, chunk_size_ms: Optional[int]):
        sr = self.encoder.embed.subsampling_rate
        frames = (self.encoder.static_chunk_size if self.encoder.static_chunk_size > 0 else 16) * sr
        if chunk_size_ms is not None:
            frames = int(chunk_size_ms / 10)
        min_frames = max(16, sr * 4)
        if frames < min_frames:
            if n_frames >= min_frames:
                frames = min_frames
            else:
                if n_frames < 7:
                    return None
                frames = n_frames
        plan = []
        cur = 0
        while cur < n_frames:
            end = min(cur + frames, n_frames)
            if end - cur == 0:
                break
            if end - cur >= 7:
                plan.append((cur, end, cur // sr))
            cur = end
        return plan

    def streaming_inference(self, audios: torch.Tensor, audio_lens: torch.Tensor,
                            chunk_size_ms: Optional[int] = None) -> Tuple[List[List[int]], None, None]:
        assert self.streaming, "Model is not in streaming mode for streaming_inference."
        assert audios.size(0) == 1, "Streaming inference currently supports batch size 1 only."
        self.reset_streaming_cache()
        n = int(audio_lens.item())
        plan = self._utterance_chunks(n, chunk_size_ms)
        if plan is None:
            print(f"Error: Input audio too short ({n} frames) for conv layers. Skipping.")
            return [[] for _ in range(audios.size(0))], None, None
        full = []
        for s, e, off in plan:
            full.extend(self._decode_chunk_streaming_logic(audios[:, s:e, :], off, off))
        return [full], None, None

    # ---- beam search (model/online_rnnt_model.py:389-645) -------------------------------------------
    def _decode_chunk_beam_search(self, chunk_xs: torch.Tensor, offset: int, required_cache_size: int,
                                  beam_hypotheses_in: Optional[List[BeamHypothesis]], beam_size: int = 4) -> List[BeamHypothesis]:
        x = chunk_xs.to(self.device, torch.float32).contiguous()
        s = _stream_ptr()
        tq = self._engine.encoder_chunk(x.data_ptr(), x.size(1), offset, required_cache_size, s)
        # The hypotheses live in the library (one empty hypothesis with the zero LSTM state after a reset, :407-415); the
        # list handed back mirrors them, and `beam_hypotheses_in` is expected to be that list (as in the reference's callers).
        self._engine.beam_advance(0, tq, beam_size, s)
        beam = [BeamHypothesis(tokens=t, log_prob=lp) for t, lp in self._engine.beam_hyps(0)]
        h, c = self._engine.beam_states(len(beam), s)
        for i, hyp in enumerate(beam):
            hyp.predictor_states = [torch.from_numpy(h[i]).view(1, 1, 256).to(self.device),
                                    torch.from_numpy(c[i]).view(1, 1, 256).to(self.device)]
        self._engine.frames_discard(s)
        self._chunks_done += 1
        return beam

    def process_single_chunk_beam_search(self, chunk_audio: torch.Tensor, chunk_len: torch.Tensor,
                                         beam_size: int = 4) -> Tuple[List[BeamHypothesis], None, None]:
        assert self.streaming, "Model is not in streaming mode for process_single_chunk_beam_search."
        assert chunk_audio.size(0) == 1, "Single chunk beam search currently supports batch size 1 only."
        if self._chunks_done is None:
            self.reset_streaming_cache()
        if chunk_audio.size(1) < 7:
            print(f"Warning: Chunk too small ({chunk_audio.size(1)} frames), skipping")
            return self.streaming_beam_hypotheses or [], None, None
        off = self._global_encoder_offset
        self.streaming_beam_hypotheses = self._decode_chunk_beam_search(chunk_audio, off, off, self.streaming_beam_hypotheses, beam_size)
        self._global_encoder_offset += chunk_audio.size(1) // self.encoder.embed.subsampling_rate
        return self.streaming_beam_hypotheses, None, None

    def streaming_beam_search(self, audios: torch.Tensor, audio_lens: torch.Tensor, beam_size: int = 4,
                              chunk_size_ms: Optional[int] = None) -> Tuple[List[List[int]], None, None]:
        assert self.streaming, "Model is not in streaming mode for streaming_beam_search."
        assert audios.size(0) == 1, "Streaming beam search currently supports batch size 1 only."
        self.reset_streaming_cache()
        n = int(audio_lens.item())
        plan = self._utterance_chunks(n, chunk_size_ms)
        if plan is None:
            print(f"Error: Input audio too short ({n} frames) for conv layers. Skipping.")
            return [[] for _ in range(audios.size(0))], None, None
        for s, e, off in plan:
            self.streaming_beam_hypotheses = self._decode_chunk_beam_search(audios[:, s:e, :], off, off, self.streaming_beam_hypotheses, beam_size)
        if self.streaming_beam_hypotheses:
            return [max(self.streaming_beam_hypotheses, key=lambda h: h.log_prob).tokens], None, None     # :598-601
        return [[]], None, None

    # ---- CTC head (model/online_rnnt_model.py:647-671) ---------------------------------------------------
    def ctc_greedy_search(self, audios: torch.Tensor, audio_lens: torch.Tensor) -> List[List[int]]:
        """Greedy CTC decode on the same encoder.  Deviation, on purpose: the reference calls `self.encoder(x, lens)`,
        which in eval mode with use_dynamic_chunk draws a RANDOM chunk mask (wenet/utils/mask.py:170-183; SURVEY.md
        §0.8: two identical calls differ by 0.61); this uses the deterministic full-context encoder
        (decoding_chunk_size=-1).  Collapse rule as in :660-671: drop blanks and repeats over the valid frames.
        Invalidates the streaming state of this object (the full-context pass reuses the conv rings)."""
        if self.ctc_weight <= 0.0:
            return [[] for _ in range(audios.size(0))]
        self._require_loaded()
        B, T = audios.size(0), audios.size(1)
        x = audios.to(self.device, torch.float32).contiguous()
        lens = audio_lens.detach().cpu().numpy().astype(np.int32)
        ids = self._engine.ctc_argmax(x.data_ptr(), lens, B, T, _stream_ptr())
        self._chunks_done = None
        hyps = []
        for b in range(B):
            n1 = max(0, (min(int(lens[b]), T) - 1) // 2)          # valid frames after masks[:, :, 2::2][:, :, 2::2]
            n = max(0, (n1 - 1) // 2)
            hyp, prev = [], -1
            for t in range(n):
                tok = int(ids[b, t])
                if tok != self.blank_id and tok != prev:
                    hyp.append(tok)
                prev = tok
            hyps.append(hyp)
        return hyps

    def forward(self, audios, audio_lens, texts=None, text_lens=None):
        """Inference branches of the reference's forward (model/online_rnnt_model.py:224-272): a streaming model goes to
        streaming_inference (:271-272); a non-streaming model runs the full-context encoder and basic_greedy_search
        (:234-235,268; model/component/transducer.py:22-70, n_steps=64) for the whole batch.  The training branch
        (texts given: joint lattice + rnnt_loss) is outside the accelerated path."""
        if texts is not None:
            raise NotImplementedError("training forward (loss) is outside the accelerated path (SURVEY.md §8a)")
        if self.streaming:
            return self.streaming_inference(audios, audio_lens)
        return self.greedy_search_full(audios, audio_lens), None, None

    def prefix_beam_search(self, audios: torch.Tensor, audio_lens: torch.Tensor, beam_size: int = 5, ctc_weight: float = 0.3,
                           transducer_weight: float = 0.7):
        """WeNet prefix beam search (wenet/transducer/search/prefix_beam_search.py:42-148) on the full-context encoder
        (decoding_chunk_size=-1), B = 1: at most one symbol per frame, CTC shallow fusion, prefix merging with log_add.
        Device: encoder (rnnt_encoder_full), CTC posteriors (rnnt_ctc_logprobs), one predictor step and one joint + log_softmax
        per frame for all hypotheses (rnnt_predictor_step, rnnt_joint).  Host: the fusion / top-k (float32, torch CPU ops as
        in the reference), candidate order, log_add merge (Python double, the LIST form the reference's call site was written
        for: its vendored log_add(*args) raises TypeError on a merge), stable sort, truncation.
        Returns [(tokens incl. the leading blank, score)], best first."""
        import math
        self._require_loaded()
        assert audios.size(0) == 1, "prefix_beam_search is batch-1 in the reference (:58)"
        eng, dev, V = self._engine, self.device, self.vocab_size
        x = audios.to(dev, torch.float32).contiguous()
        T = x.size(1)
        tq = ((T - 3) // 2 + 1 - 3) // 2 + 1
        s = _stream_ptr()
        enc = torch.empty(1, tq, 256, device=dev)
        eng.encoder_full(x.data_ptr(), np.asarray([int(audio_lens[0])], np.int32), 1, T, enc.data_ptr(), s)
        ctc_dev = torch.empty(tq, V, device=dev)
        eng.ctc_logprobs(enc.data_ptr(), tq, ctc_dev.data_ptr(), s)
        ctc = ctc_dev.cpu()
        hyps, scores = [[self.blank_id]], [0.0]
        h = torch.zeros(1, 256, device=dev)
        c = torch.zeros(1, 256, device=dev)

        def log_add(args):
            if all(a == -float("inf") for a in args):
                return -float("inf")
            a_max = max(args)
            return a_max + math.log(sum(math.exp(a - a_max) for a in args))
        for i in range(tq):                                      # every encoder frame, padded ones included (maxlen = encoder_out.size(1), :64,76)
            n = len(hyps)
            tok = torch.tensor([hy[-1] for hy in hyps], dtype=torch.int32, device=dev)
            pred, h2, c2 = torch.empty(n, 256, device=dev), torch.empty(n, 256, device=dev), torch.empty(n, 256, device=dev)
            eng.predictor_step(tok.data_ptr(), h.data_ptr(), c.data_ptr(), n, pred.data_ptr(), h2.data_ptr(), c2.data_ptr(), s)
            lp_dev = torch.empty(1, 1, n, V, device=dev)
            eng.joint(enc[:, i:i + 1].contiguous().data_ptr(), pred.data_ptr(), 1, 1, n, 1, lp_dev.data_ptr(), s)
            logp = lp_dev.view(n, V).cpu()
            logp = torch.log(torch.add(transducer_weight * torch.exp(logp), ctc_weight * torch.exp(ctc[i].unsqueeze(0))))   # :99-101
            top_lp, top_ix = logp.topk(beam_size)                                                                            # :104
            sc = torch.add(torch.tensor(scores).unsqueeze(1), top_lp)                                                        # :105 (float32)
            cand = []                                            # [tokens, score, state row in cat(h, h2)]
            for j in range(n):
                for t in range(beam_size):
                    if int(top_ix[j, t]) == self.blank_id:
                        cand.append([list(hyps[j]), sc[j, t].item(), j])
                    else:
                        cand.append([list(hyps[j]) + [int(top_ix[j, t])], sc[j, t].item(), n + j])
            fused = [cand[0]]
            for cnd in cand[1:]:
                for f in fused:
                    if cnd[0] == f[0]:
                        f[1] = log_add([f[1], cnd[1]])
                        break
                else:
                    fused.append(cnd)
            fused.sort(key=lambda v: v[1], reverse=True)
            fused = fused[:beam_size]
            rows = torch.tensor([f[2] for f in fused], device=dev)
            h = torch.cat([h, h2], 0).index_select(0, rows).contiguous()
            c = torch.cat([c, c2], 0).index_select(0, rows).contiguous()
            hyps, scores = [f[0] for f in fused], [f[1] for f in fused]
        self._prefix_states = (h, c)
        return list(zip(hyps, scores))

    def greedy_search_full(self, audios, audio_lens, n_steps: int = 64):
        """basic_greedy_search over the deterministic full-context encoder; audios [B,T,80] with B <= max_streams,
        T <= max_chunk_frames, ((T-3)//2+1-3)//2+1 <= max_enc_frames.  Invalidates the streaming state."""
        self._require_loaded()
        B, T = audios.size(0), audios.size(1)
        x = audios.to(self.device, torch.float32).contiguous()
        lens = audio_lens.detach().cpu().numpy().astype(np.int32)
        hyps = self._engine.greedy_search_full(x.data_ptr(), lens, B, T, n_steps, _stream_ptr())
        self._chunks_done = None
        return hyps

    __call__ = forward


class StreamingBatch:
    """B independent streams advanced in lock step through one context (not in the reference, which
    is B=1): the chunk loop of online_rnnt_decode.py:81-117 / streaming_inference for a whole batch."""

    def __init__(self, state_dict, n_streams: int, vocab_size: int = 412, blank_id: int = 5, max_chunk_frames: int = 64,
                 max_cache_frames: int = 512, max_enc_frames: int = 512, max_tokens: int = 4096, device: int = 0, max_beam: int = 0,
                 numerics=None, packed=None):
        """state_dict: the reference's 504-key dict (numpy / torch values), or None with packed=(blob, vocab): the flat float32
        blob dist.broadcast_packed left on this rank's device, handed to the context in one call (RnntEngine.load_packed)."""
        self.device = torch.device("cuda", device)
        self.n = n_streams
        self.blank_id = blank_id
        self.engine = RnntEngine(max_streams=n_streams, max_chunk_frames=max_chunk_frames, max_cache_frames=max_cache_frames,
                                 max_enc_frames=max_enc_frames, max_tokens=max_tokens, vocab_size=vocab_size, blank_id=blank_id,
                                 n_steps=10, device=device, max_beam=max_beam)
        self.beams = None
        self.python_beam = False      # True: host half of the beam search in Python (beam_advance_frame), for tests
        if packed is not None:
            assert state_dict is None and int(packed[1]) == vocab_size, "packed=(blob, vocab): vocab must equal vocab_size"
            self.engine.load_packed(packed[0], int(packed[1]), numerics=numerics)
        else:
            self.engine.load_state_dict(state_dict, numerics=numerics)
        self.offset = 0

    def reset(self):
        self.engine.reset(self.n, _stream_ptr())
        self.offset = 0
        self.beams = None

    def process_chunk_beam(self, chunks: torch.Tensor, beam_size: int = 4):
        """process_single_chunk_beam_search semantics for every stream (online_rnnt_model.py:605-645)."""
        assert chunks.size(0) == self.n and chunks.is_cuda and chunks.dtype == torch.float32 and chunks.is_contiguous()
        if chunks.size(1) < 7:
            return self.beams
        s = _stream_ptr()
        tq = self.engine.encoder_chunk(chunks.data_ptr(), chunks.size(1), self.offset, self.offset, s)
        self.offset += chunks.size(1) // 4
        if self.python_beam:                                   # reference implementation of the host half (tests)
            if self.beams is None:
                self.beams = [[BeamHypothesis([], 0.0)] for _ in range(self.n)]
            for t in range(tq):
                self.beams = beam_advance_frame(self.engine, t, self.beams, self.blank_id, beam_size, s)
        else:                                                  # bookkeeping inside the library (rnnt_beam_advance)
            self.engine.beam_advance(0, tq, beam_size, s)
            self.beams = self._native_beams()
        self.engine.frames_discard(s)
        return self.beams

    def _native_beams(self):
        return [[BeamHypothesis(t, lp) for t, lp in self.engine.beam_hyps(b)] for b in range(self.n)]

    def beam_script(self, audios: torch.Tensor, chunk_frames: int, beam_size: int = 4, pipelined: bool = False):
        """Beam loop of online_rnnt_decode.py:123-178 over [B,T,80]; returns the final beams per stream.
        pipelined=True: the whole utterance's encoder in one rnnt_encoder_chunks call, then ONE rnnt_beam_advance over
        all frames (same hypotheses: the beam recursion only consumes encoder frames in order); needs
        max_enc_frames >= the utterance's encoder frames."""
        from .layout import chunk_plan
        self.reset()
        if pipelined and not self.python_beam:
            assert audios.is_cuda and audios.dtype == torch.float32 and audios.is_contiguous()
            plan = [(a, b) for a, b in chunk_plan(audios.size(1), chunk_frames) if b - a >= 7]
            offs, o = [], 0
            for a, b in plan:
                offs.append(o)
                o += (b - a) // 4
            s = _stream_ptr()
            frames = self.engine.encoder_chunks(audios.data_ptr(), audios.size(1), [a for a, _ in plan], [b - a for a, b in plan], offs, offs, s, greedy=False)
            self.offset = o
            self.engine.beam_advance(0, frames, beam_size, s)
            self.beams = self._native_beams()
            self.engine.frames_discard(s)
            return self.beams
        for (a, b) in chunk_plan(audios.size(1), chunk_frames):
            self.process_chunk_beam(audios[:, a:b, :].contiguous(), beam_size)
        return self.beams

    def process_chunk(self, chunks: torch.Tensor, decode: bool = True):
        """chunks [B,T,80] on the device; process_single_chunk semantics for every stream."""
        assert chunks.size(0) == self.n and chunks.is_cuda and chunks.dtype == torch.float32 and chunks.is_contiguous()
        if chunks.size(1) < 7:
            return None
        s = _stream_ptr()
        self.engine.encoder_chunk(chunks.data_ptr(), chunks.size(1), self.offset, self.offset, s)
        self.offset += chunks.size(1) // 4
        if decode:
            self.engine.greedy_decode(s)
            self.engine.frames_consume(s)

    def decode_script_ragged(self, audios: torch.Tensor, audio_lens, chunk_frames: int, pipelined: bool = True) -> List[List[int]]:
        """Greedy loop of online_rnnt_decode.py:81-117 for a PADDED batch of utterances of different lengths (utils/utils.py:29-50
        pads them; online_rnnt_eval.py:86-94 decodes each with its own audio_lens): stream b is decoded over its own
        audio_lens[b] frames with its own chunk plan (tail-merge rule, < 7-frame skip), so its tokens equal its B = 1 result.
        pipelined=True: ONE library call (rnnt_decode_ragged: per-stream chunk plans inside the layer-major launches) for every
        utterance of at least two chunks; utterances that are a single chunk (< chunk_frames + max(16, chunk_frames) frames) and the
        per-chunk form (pipelined=False) run as LENGTH CLASSES: streams of equal length share one whole-utterance call."""
        assert audios.is_cuda and audios.dtype == torch.float32 and audios.size(0) == self.n
        lens = [int(v) for v in (audio_lens.tolist() if hasattr(audio_lens, "tolist") else audio_lens)]
        assert len(lens) == self.n and max(lens) <= audios.size(1) and min(lens) >= 0
        out: List[Optional[List[int]]] = [None] * self.n
        two_chunks = chunk_frames + max(16, chunk_frames)
        done = set()
        if pipelined and audios.is_contiguous():
            ragged = [b for b in range(self.n) if lens[b] >= two_chunks or lens[b] < 7]
            big = [lens[b] for b in ragged if lens[b] >= two_chunks]
            if big and max(big) >= 2 * chunk_frames + max(16, chunk_frames):    # the longest one has at least three chunks
                self.reset()
                call_lens = [lens[b] if b in set(ragged) else 0 for b in range(self.n)]
                s = _stream_ptr()
                self.engine.decode_ragged(audios.data_ptr(), audios.size(1), call_lens, chunk_frames, s)
                toks = self.engine.tokens(s)
                for b in ragged:
                    out[b] = toks[b]
                done = set(ragged)
        n_all = self.n
        try:
            for T_ in sorted({lens[b] for b in range(n_all) if b not in done}, reverse=True):
                idx = [b for b in range(n_all) if lens[b] == T_ and b not in done]
                if T_ < 7:                                   # shorter than the conv front-end's receptive field: skipped (:356-359)
                    for b in idx:
                        out[b] = []
                    continue
                sub = audios[torch.tensor(idx, device=audios.device), :T_, :].contiguous()
                self.n = len(idx)
                toks = self.decode_script(sub, chunk_frames, per_chunk_decode=not pipelined, pipelined=pipelined)
                for b, t in zip(idx, toks):
                    out[b] = t
        finally:
            self.n = n_all
        return out

    def decode_script(self, audios: torch.Tensor, chunk_frames: int, per_chunk_decode: bool = True, pipelined: bool = False) -> List[List[int]]:
        """Greedy loop of online_rnnt_decode.py:81-117 over [B,T,80] equal-length utterances.
        per_chunk_decode=False runs the encoder over all chunks first and decodes once at the end
        (identical tokens: the greedy state machine is causal in the frame index).
        pipelined=True hands the whole chunk plan to rnnt_encoder_chunks (wavefront over chunk x layer,
        bit-identical encoder output) and decodes once."""
        from .layout import chunk_plan
        self.reset()
        T = audios.size(1)
        if pipelined:
            assert audios.is_cuda and audios.dtype == torch.float32 and audios.is_contiguous()
            plan = chunk_plan(T, chunk_frames)
            starts = [a for a, b in plan if b - a >= 7]
            lens = [b - a for a, b in plan if b - a >= 7]
            offs, o = [], 0
            for a, b in plan:                      # process_single_chunk: offset += frames // 4 (online_rnnt_model.py:384-385)
                if b - a >= 7:
                    offs.append(o)
                    o += (b - a) // 4
            s = _stream_ptr()
            self.engine.encoder_chunks(audios.data_ptr(), T, starts, lens, offs, offs, s, greedy=True)
            self.offset = o
            self.engine.frames_consume(s)
            return self.engine.tokens(s)
        for (a, b) in chunk_plan(T, chunk_frames):
            self.process_chunk(audios[:, a:b, :].contiguous(), decode=per_chunk_decode)
        if not per_chunk_decode:
            s = _stream_ptr()
            self.engine.greedy_decode(s)
            self.engine.frames_consume(s)
        return self.engine.tokens(_stream_ptr())
