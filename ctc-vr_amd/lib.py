"""ctypes binding of librnnt_hip.so (include/rnnt_hip.h).  Fails loudly when the HIP library is
missing or a call returns an error: there is no CPU fallback in the product path."""
import ctypes
import os

import numpy as np

from . import build as _build

c_i32, c_i64, c_f32p, c_i32p, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32), ctypes.c_void_p


class RnntConfig(ctypes.Structure):
    _fields_ = [("max_streams", c_i32), ("max_chunk_frames", c_i32), ("max_cache_frames", c_i32),
                ("max_enc_frames", c_i32), ("max_tokens", c_i32), ("vocab_size", c_i32), ("blank_id", c_i32),
                ("n_steps", c_i32), ("device", c_i32), ("max_beam", c_i32)]


# symbol -> (restype, argtypes); exactly the entry points declared in include/rnnt_hip.h
SIGNATURES = {
    "rnnt_create": (c_i32, [ctypes.POINTER(RnntConfig), ctypes.POINTER(c_vp)]),
    "rnnt_destroy": (None, [c_vp]),
    "rnnt_last_error": (ctypes.c_char_p, [c_vp]),
    "rnnt_abi_version": (c_i32, []),
    "rnnt_load_tensor": (c_i32, [c_vp, ctypes.c_char_p, c_vp, c_i32, ctypes.POINTER(c_i64)]),
    "rnnt_decode_ragged": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp]),
    "rnnt_load_packed": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(c_i32), ctypes.POINTER(c_i64)]),
    "rnnt_finalize_weights": (c_i32, [c_vp, c_i32, c_vp]),
    "rnnt_streams_reset": (c_i32, [c_vp, c_i32, c_vp]),
    "rnnt_encoder_chunk": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32p, c_vp]),
    "rnnt_encoder_chunks": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32p, c_vp]),
    "rnnt_greedy_decode": (c_i32, [c_vp, c_vp]),
    "rnnt_get_tokens": (c_i32, [c_vp, c_vp, c_vp, c_vp]),
    "rnnt_frames_consume": (c_i32, [c_vp, c_vp]),
    "rnnt_beam_frame": (c_i32, [c_vp, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rnnt_beam_select": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp]),
    "rnnt_beam_get_states": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp]),
    "rnnt_beam_advance": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp]),
    "rnnt_beam_hyp_count": (c_i32, [c_vp, c_i32, c_i32p]),
    "rnnt_beam_get_hyp": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_i32p, ctypes.POINTER(ctypes.c_double)]),
    "rnnt_beam_merge_host": (c_i32, [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rnnt_frames_discard": (c_i32, [c_vp, c_vp]),
    "rnnt_predictor_step": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "rnnt_joint": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "rnnt_encoder_full": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_i32p, c_vp]),
    "rnnt_ctc_argmax": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_i32p, c_vp]),
    "rnnt_ctc_logprobs": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp]),
    "rnnt_fbank": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32p, c_vp]),
    "rnnt_greedy_search_full": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "rnnt_get_att_cache": (c_i32, [c_vp, c_i32, c_vp, c_i32p, c_vp]),
    "rnnt_get_cnn_cache": (c_i32, [c_vp, c_i32, c_vp, c_vp]),
    "rnnt_get_predictor_state": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_i32p, c_vp]),
    "rnnt_get_enc_frames": (c_i32, [c_vp, c_vp, c_i32p, c_vp]),
    "rnnt_enc_frames_dev": (c_vp, [c_vp, c_i32p, c_i32p]),
    "rnnt_profile_begin": (c_i32, [c_vp, c_i32]),
    "rnnt_profile_end": (c_i32, [c_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64)]),
    "rnnt_get_counters": (c_i32, [c_vp, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
}

_LIB = None

# numerics modes of rnnt_finalize_weights (include/rnnt_hip.h RNNT_NUMERICS_*)
NUMERICS = {"fp32": 0, "bf16x3": 1, "bf16": 2, "f16x3": 3}


def numerics_id(mode=None):
    """None -> $RNNT_NUMERICS or "fp32" (the exact-f32 parity mode); accepts the names above or their integer ids."""
    if mode is None:
        mode = os.environ.get("RNNT_NUMERICS", "fp32")
    if isinstance(mode, str):
        if mode not in NUMERICS:
            raise ValueError(f"unknown numerics mode {mode!r}: one of {sorted(NUMERICS)}")
        return NUMERICS[mode]
    return int(mode)


class RnntError(RuntimeError):
    pass


def load(build_if_needed=True):
    """Load (building in-tree with hipcc when sources are newer) librnnt_hip.so."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (soname libamdhip64.so.7).
    # Importing torch FIRST makes the loader resolve our NEEDED libamdhip64.so.7 to that already-loaded
    # runtime; loading ours first would pull in /opt/rocm's copy and the second runtime finds no device.
    import torch  # noqa: F401
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tlib):
        ctypes.CDLL(tlib, mode=ctypes.RTLD_GLOBAL)
    path = _build.lib_path()
    if build_if_needed and _build.needs_build() and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        _build.build()
    if not os.path.exists(path):
        raise RnntError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the library does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    _LIB = lib
    return lib


def _np_ptr(a):
    return a.ctypes.data_as(c_vp)


class RnntEngine:
    """One context = one GPU = up to `max_streams` lock-stepped streams."""

    def __init__(self, max_streams=1, max_chunk_frames=64, max_cache_frames=1024, max_enc_frames=1024, max_tokens=4096,
                 vocab_size=412, blank_id=5, n_steps=10, device=0, max_beam=0):
        self.lib = load()
        self.cfg = RnntConfig(max_streams, max_chunk_frames, max_cache_frames, max_enc_frames, max_tokens, vocab_size,
                              blank_id, n_steps, device, max_beam)
        self.ctx = c_vp()
        rc = self.lib.rnnt_create(ctypes.byref(self.cfg), ctypes.byref(self.ctx))
        if rc != 0:
            msg = self.lib.rnnt_last_error(self.ctx).decode() if self.ctx else "rnnt_create failed"
            if self.ctx:
                self.lib.rnnt_destroy(self.ctx)
                self.ctx = c_vp()
            raise RnntError(f"rnnt_create: {msg} (status {rc})")
        self.n_streams = 0

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.rnnt_destroy(self.ctx)
            self.ctx = c_vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise RnntError(f"{what}: {self.lib.rnnt_last_error(self.ctx).decode()} (status {rc})")

    # ---- weights ----------------------------------------------------------------------------
    def load_state_dict(self, sd, stream=None, numerics=None):
        """sd: name -> float32 numpy array or torch tensor (the reference's 504-key layout); numerics: see numerics_id."""
        for name, v in sd.items():
            if hasattr(v, "detach"):
                v = v.detach().cpu().numpy()
            if "num_batches_tracked" in name:
                continue
            a = np.ascontiguousarray(v, dtype=np.float32)
            dims = (c_i64 * max(a.ndim, 1))(*a.shape)
            self._chk(self.lib.rnnt_load_tensor(self.ctx, name.encode(), _np_ptr(a), a.ndim, dims), f"rnnt_load_tensor({name})")
        self.numerics = numerics_id(numerics)
        self._chk(self.lib.rnnt_finalize_weights(self.ctx, self.numerics, stream), "rnnt_finalize_weights")

    def load_packed(self, blob, vocab, stream=None, numerics=None):
        """The whole state dict from ONE flat float32 blob in layout.state_dict_spec(vocab) order (num_batches_tracked slots are
        one float each and ignored): a torch tensor on this context's device (the blob dist.broadcast_packed leaves there) or a
        numpy array on the host.  One C-ABI call (rnnt_load_packed), then rnnt_finalize_weights."""
        from .layout import state_dict_spec
        spec = state_dict_spec(vocab)
        names = (ctypes.c_char_p * len(spec))(*[n.encode() for n, _, _ in spec])
        ndims = (c_i32 * len(spec))(*[len(s) for _, s, _ in spec])
        flat = [d for _, s, _ in spec for d in s]
        dims = (c_i64 * max(len(flat), 1))(*flat)
        if hasattr(blob, "data_ptr"):
            assert blob.dtype.is_floating_point and blob.element_size() == 4 and blob.is_contiguous()
            ptr, n, on_dev = blob.data_ptr(), blob.numel(), 1 if blob.is_cuda else 0
        else:
            blob = np.ascontiguousarray(blob, np.float32)
            ptr, n, on_dev = _np_ptr(blob), blob.size, 0
        self._chk(self.lib.rnnt_load_packed(self.ctx, ptr, n, on_dev, len(spec), names, ndims, dims), "rnnt_load_packed")
        self.numerics = numerics_id(numerics)
        self._chk(self.lib.rnnt_finalize_weights(self.ctx, self.numerics, stream), "rnnt_finalize_weights")

    # ---- streaming --------------------------------------------------------------------------
    def reset(self, n_streams, stream=None):
        self._chk(self.lib.rnnt_streams_reset(self.ctx, n_streams, stream), "rnnt_streams_reset")
        self.n_streams = n_streams

    def encoder_chunk(self, fbank_ptr, chunk_frames, offset, required_cache_size, stream=None):
        t = c_i32(0)
        self._chk(self.lib.rnnt_encoder_chunk(self.ctx, fbank_ptr, chunk_frames, offset, required_cache_size, ctypes.byref(t), stream),
                  "rnnt_encoder_chunk")
        return t.value

    def decode_ragged(self, fbank_ptr, total_frames, lens, chunk_frames, stream=None):
        """rnnt_decode_ragged: every stream over its own lens[b] frames (decode-script chunk loop), one call; returns encoder frames per stream."""
        a = np.ascontiguousarray(lens, np.int32)
        assert a.size == self.n_streams
        fo = np.zeros(self.n_streams, np.int32)
        self._chk(self.lib.rnnt_decode_ragged(self.ctx, fbank_ptr, total_frames, _np_ptr(a), chunk_frames, _np_ptr(fo), stream), "rnnt_decode_ragged")
        return fo

    def encoder_chunks(self, fbank_ptr, total_frames, starts, lens, offsets, required, stream=None, greedy=False):
        a, b, c, d = (np.ascontiguousarray(v, np.int32) for v in (starts, lens, offsets, required))
        t = c_i32(0)
        self._chk(self.lib.rnnt_encoder_chunks(self.ctx, fbank_ptr, total_frames, len(a), _np_ptr(a), _np_ptr(b), _np_ptr(c), _np_ptr(d),
                                               1 if greedy else 0, ctypes.byref(t), stream), "rnnt_encoder_chunks")
        return t.value

    def greedy_decode(self, stream=None):
        self._chk(self.lib.rnnt_greedy_decode(self.ctx, stream), "rnnt_greedy_decode")

    def frames_consume(self, stream=None):
        self._chk(self.lib.rnnt_frames_consume(self.ctx, stream), "rnnt_frames_consume")

    def token_counts(self, stream=None):
        counts = np.zeros(self.n_streams, np.int32)
        self._chk(self.lib.rnnt_get_tokens(self.ctx, _np_ptr(counts), None, stream), "rnnt_get_tokens")
        return counts

    def tokens(self, stream=None):
        counts = np.zeros(self.n_streams, np.int32)
        toks = np.zeros((self.n_streams, self.cfg.max_tokens), np.int32)
        self._chk(self.lib.rnnt_get_tokens(self.ctx, _np_ptr(counts), _np_ptr(toks), stream), "rnnt_get_tokens")
        if counts.max(initial=0) > self.cfg.max_tokens:
            raise RnntError("token buffer overflow: raise max_tokens")
        return [toks[b, :counts[b]].tolist() for b in range(self.n_streams)]

    # ---- beam search (device half) ------------------------------------------------------------
    def beam_frame(self, frame_idx, row_stream, row_tok, beam_k, stream=None):
        n, ns = len(row_stream), self.cfg.n_steps
        rs = np.ascontiguousarray(row_stream, np.int32)
        rt = np.ascontiguousarray(row_tok, np.int32)
        steps = np.zeros(n, np.int32)
        blank = np.zeros((n, ns), np.float32)
        top_lp = np.zeros((n, ns, beam_k), np.float32)
        top_tok = np.zeros((n, ns, beam_k), np.int32)
        self._chk(self.lib.rnnt_beam_frame(self.ctx, frame_idx, n, _np_ptr(rs), _np_ptr(rt), beam_k, _np_ptr(steps), _np_ptr(blank),
                                           _np_ptr(top_lp), _np_ptr(top_tok), stream), "rnnt_beam_frame")
        return steps, blank, top_lp, top_tok

    def beam_select(self, src_row, src_step, stream=None):
        a = np.ascontiguousarray(src_row, np.int32)
        b = np.ascontiguousarray(src_step, np.int32)
        self._chk(self.lib.rnnt_beam_select(self.ctx, len(a), _np_ptr(a), _np_ptr(b), stream), "rnnt_beam_select")

    def beam_advance(self, frame_begin, frame_end, beam_size, stream=None):
        """Native beam search over buffered frames [frame_begin, frame_end) of every stream (bookkeeping in the library)."""
        self._chk(self.lib.rnnt_beam_advance(self.ctx, frame_begin, frame_end, beam_size, stream), "rnnt_beam_advance")

    def beam_hyps(self, b):
        """[(tokens, log_prob), ...] of stream b in device-row order."""
        n = c_i32(0)
        self._chk(self.lib.rnnt_beam_hyp_count(self.ctx, b, ctypes.byref(n)), "rnnt_beam_hyp_count")
        out = []
        for i in range(n.value):
            nt, lp = c_i32(0), ctypes.c_double(0.0)
            self._chk(self.lib.rnnt_beam_get_hyp(self.ctx, b, i, 0, None, ctypes.byref(nt), ctypes.byref(lp)), "rnnt_beam_get_hyp")
            toks = np.zeros(max(nt.value, 1), np.int32)
            self._chk(self.lib.rnnt_beam_get_hyp(self.ctx, b, i, toks.size, _np_ptr(toks), ctypes.byref(nt), ctypes.byref(lp)), "rnnt_beam_get_hyp")
            out.append((toks[:nt.value].tolist(), lp.value))
        return out

    def beam_states(self, n_rows, stream=None):
        h, c = np.zeros((n_rows, 256), np.float32), np.zeros((n_rows, 256), np.float32)
        self._chk(self.lib.rnnt_beam_get_states(self.ctx, n_rows, _np_ptr(h), _np_ptr(c), stream), "rnnt_beam_get_states")
        return h, c

    def frames_discard(self, stream=None):
        self._chk(self.lib.rnnt_frames_discard(self.ctx, stream), "rnnt_frames_discard")

    # ---- step API ---------------------------------------------------------------------------
    def predictor_step(self, tok_ptr, h_ptr, c_ptr, rows, out_ptr, h_out_ptr, c_out_ptr, stream=None):
        self._chk(self.lib.rnnt_predictor_step(self.ctx, tok_ptr, h_ptr, c_ptr, rows, out_ptr, h_out_ptr, c_out_ptr, stream), "rnnt_predictor_step")

    def joint(self, enc_ptr, pred_ptr, B, T, U, mode, out_ptr, stream=None):
        self._chk(self.lib.rnnt_joint(self.ctx, enc_ptr, pred_ptr, B, T, U, mode, out_ptr, stream), "rnnt_joint")

    def encoder_full(self, fbank_ptr, lens, B, T, out_ptr, stream=None):
        lens = np.ascontiguousarray(lens, np.int32)
        t = c_i32(0)
        self._chk(self.lib.rnnt_encoder_full(self.ctx, fbank_ptr, _np_ptr(lens), B, T, out_ptr, ctypes.byref(t), stream), "rnnt_encoder_full")
        self.n_streams = 0
        return t.value

    def ctc_argmax(self, fbank_ptr, lens, B, T, stream=None):
        lens = np.ascontiguousarray(lens, np.int32)
        tq = ((T - 3) // 2 + 1 - 3) // 2 + 1
        ids = np.zeros((B, tq), np.int32)
        t = c_i32(0)
        self._chk(self.lib.rnnt_ctc_argmax(self.ctx, fbank_ptr, _np_ptr(lens), B, T, _np_ptr(ids), ctypes.byref(t), stream), "rnnt_ctc_argmax")
        self.n_streams = 0
        return ids

    def ctc_logprobs(self, enc_ptr, rows, out_ptr, stream=None):
        self._chk(self.lib.rnnt_ctc_logprobs(self.ctx, enc_ptr, rows, out_ptr, stream), "rnnt_ctc_logprobs")

    def greedy_search_full(self, fbank_ptr, lens, B, T, n_steps=64, stream=None):
        """Offline greedy search over the full-context encoder (model/component/transducer.py:22-70) -> list of token lists."""
        lens = np.ascontiguousarray(lens, np.int32)
        counts = np.zeros(B, np.int32)
        toks = np.zeros((B, self.cfg.max_tokens), np.int32)
        self._chk(self.lib.rnnt_greedy_search_full(self.ctx, fbank_ptr, _np_ptr(lens), B, T, n_steps, _np_ptr(counts), _np_ptr(toks), stream),
                  "rnnt_greedy_search_full")
        self.n_streams = 0
        if counts.max(initial=0) > self.cfg.max_tokens:
            raise RnntError("token buffer overflow: raise max_tokens")
        return [toks[b, :counts[b]].tolist() for b in range(B)]

    def fbank(self, wave_ptr, B, n_samples, sample_rate, out_ptr, n_fft=1024, stream=None):
        """Device feature front-end (data/dataloader.py:15-41): wave [B, n_samples] -> out [B, 1 + n_samples // 512, 80]."""
        t = c_i32(0)
        self._chk(self.lib.rnnt_fbank(self.ctx, wave_ptr, B, n_samples, sample_rate, n_fft, out_ptr, ctypes.byref(t), stream), "rnnt_fbank")
        return t.value

    # ---- state read-back ----------------------------------------------------------------------
    def att_cache(self, b=0, stream=None):
        n = c_i32(0)
        self._chk(self.lib.rnnt_get_att_cache(self.ctx, b, None, ctypes.byref(n), stream), "rnnt_get_att_cache")
        out = np.zeros((12, 4, n.value, 128), np.float32)
        if n.value:
            self._chk(self.lib.rnnt_get_att_cache(self.ctx, b, _np_ptr(out), ctypes.byref(n), stream), "rnnt_get_att_cache")
        return out

    def cnn_cache(self, b=0, stream=None):
        out = np.zeros((12, 1, 256, 30), np.float32)
        self._chk(self.lib.rnnt_get_cnn_cache(self.ctx, b, _np_ptr(out), stream), "rnnt_get_cnn_cache")
        return out

    def predictor_state(self, b=0, stream=None):
        h, c, tok = np.zeros(256, np.float32), np.zeros(256, np.float32), c_i32(0)
        self._chk(self.lib.rnnt_get_predictor_state(self.ctx, b, _np_ptr(h), _np_ptr(c), ctypes.byref(tok), stream), "rnnt_get_predictor_state")
        return h, c, tok.value

    def enc_frames(self, stream=None):
        n = c_i32(0)
        self._chk(self.lib.rnnt_get_enc_frames(self.ctx, None, ctypes.byref(n), stream), "rnnt_get_enc_frames")
        out = np.zeros((self.n_streams, n.value, 256), np.float32)
        if n.value:
            self._chk(self.lib.rnnt_get_enc_frames(self.ctx, _np_ptr(out), ctypes.byref(n), stream), "rnnt_get_enc_frames")
        return out

    def profile_begin(self, tag):
        self._chk(self.lib.rnnt_profile_begin(self.ctx, tag), "rnnt_profile_begin")

    def profile_end(self):
        ms, n = ctypes.c_double(0), c_i64(0)
        self._chk(self.lib.rnnt_profile_end(self.ctx, ctypes.byref(ms), ctypes.byref(n)), "rnnt_profile_end")
        return ms.value, n.value

    def counters(self):
        a, b = c_i64(0), c_i64(0)
        self._chk(self.lib.rnnt_get_counters(self.ctx, ctypes.byref(a), ctypes.byref(b)), "rnnt_get_counters")
        return a.value, b.value
