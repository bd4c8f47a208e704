"""Real-time-factor harness with the semantics of the reference's online_rnnt_delay.py:14-133: per utterance, per
chunk, wall time around process_single_chunk / process_single_chunk_beam_search divided by the chunk's audio duration
(10 ms frame shift, :22,:55-59), then mean / p50 / p80 / p90 / p95 / max over all chunks (:99-131).  The chunk calls
return token lists to the host, so they are synchronous, exactly as the reference's `.item()` makes them."""
import time

import numpy as np
import torch


def _chunks(n_frames, chunk_frames):
    min_chunk = max(16, chunk_frames)                      # online_rnnt_delay.py:20-21
    off = 0
    while off < n_frames:
        end = min(off + chunk_frames, n_frames)
        if n_frames - end < min_chunk and end < n_frames:  # :41-42
            end = n_frames
        if end - off == 0:
            break
        yield off, end
        off = end
        if end >= n_frames:
            break


def _stats(rtfs):
    a = np.asarray(rtfs, np.float64)
    if a.size == 0:
        return None
    return {"mean": float(a.mean()), "p50": float(np.percentile(a, 50)), "p80": float(np.percentile(a, 80)),
            "p90": float(np.percentile(a, 90)), "p95": float(np.percentile(a, 95)), "max": float(a.max()), "chunks": int(a.size)}


def evaluate_rtf(model, utterances, chunk_frames, beam_size=4, frame_shift_s=0.01, with_beam=True):
    """utterances: iterable of [T,80] float tensors.  Returns {'greedy': stats, 'beam': stats}."""
    g_rtf, b_rtf = [], []
    for audio in utterances:
        x = audio.unsqueeze(0)
        for rtfs, fn in ((g_rtf, lambda c, n: model.process_single_chunk(c, n)),
                         (b_rtf, (lambda c, n: model.process_single_chunk_beam_search(c, n, beam_size=beam_size)) if with_beam else None)):
            if fn is None:
                continue
            model.reset_streaming_cache()
            for a, b in _chunks(x.shape[1], chunk_frames):
                c = x[:, a:b, :]
                t0 = time.time()                           # :50-53
                fn(c, torch.tensor([c.shape[1]]))
                dt = time.time() - t0
                dur = c.shape[1] * frame_shift_s
                rtfs.append(dt / dur if dur > 0 else 0.0)  # :55-59
    return {"greedy": _stats(g_rtf), "beam": _stats(b_rtf)}
