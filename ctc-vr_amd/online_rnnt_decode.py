"""wav -> tokens on the GPU with the flow of the reference's online_rnnt_decode.py:10-190 (`decode_single_audio`): feature
extraction (device front-end instead of torchaudio), model construction with the reference's keywords, checkpoint =
dict with key 'model', then the chunk loop twice -- greedy (process_single_chunk, :81-117) and beam search
(process_single_chunk_beam_search, :123-178, new tokens = best hypothesis beyond what was already reported) -- with the
reference's chunk rule (tail shorter than max(16, chunk) merged into the last chunk, :88-91).

Differences, on purpose: the tokenizer is optional (any object with .decode(ids) -> list of str; the reference's Tokenizer
needs its vocabulary file), the checkpoint is read with weights_only=True, and the results are returned as well as printed."""
import argparse

import torch

from .features import load_wav
from .online_rnnt_model import OnlineRNNTModel


def chunk_bounds(n_frames, chunk_frames):
    """(start, end) pairs of online_rnnt_decode.py:84-116."""
    min_chunk = max(16, chunk_frames)
    off, out = 0, []
    while off < n_frames:
        end = min(off + chunk_frames, n_frames)
        if n_frames - end < min_chunk and end < n_frames:
            end = n_frames
        out.append((off, end))
        off = end
    return out


def decode_features(model, audio_features, chunk_frames, beam_size=4, tokenizer=None, verbose=True):
    """audio_features [T, 80] -> {'greedy_tokens', 'beam_tokens', 'beam_hypotheses', 'chunks'}."""
    x = audio_features.unsqueeze(0)
    show = (lambda ids: " ".join(tokenizer.decode(ids))) if tokenizer is not None else (lambda ids: str(ids))
    bounds = chunk_bounds(x.shape[1], chunk_frames)
    model.eval()
    model.reset_streaming_cache()
    greedy = []
    for a, b in bounds:
        toks, _, _ = model.process_single_chunk(x[:, a:b, :], torch.tensor([b - a]))
        greedy.extend(toks)
        if verbose:
            print(f"chunk {a}:{b} greedy {show(toks) if toks else '[none]'}")
    model.reset_streaming_cache()
    beam, hyps = [], []
    for a, b in bounds:
        hyps, _, _ = model.process_single_chunk_beam_search(x[:, a:b, :], torch.tensor([b - a]), beam_size=beam_size)
        if hyps:
            best = max(hyps, key=lambda h: h.log_prob)
            new = best.tokens[len(beam):]                       # :148
            beam.extend(new)
            if verbose:
                print(f"chunk {a}:{b} beam {show(new) if new else '[nothing new]'} (best log_prob {best.log_prob:.4f})")
    if verbose:
        print(f"greedy: {show(greedy)}\nbeam:   {show(beam)}\nchunks: {len(bounds)}" + ("\ngreedy == beam" if greedy == beam else ""))
    return {"greedy_tokens": greedy, "beam_tokens": beam, "beam_hypotheses": hyps, "chunks": len(bounds)}


def decode_single_audio(audio_file, model_path, vocab_size, blank_id, static_chunk_size=32, beam_size=4, tokenizer=None, device=0,
                        verbose=True, **model_kwargs):
    wav, rate = load_wav(audio_file)
    model = OnlineRNNTModel(input_dim=80, hidden_dim=256, vocab_size=vocab_size, blank_id=blank_id, streaming=True,
                            static_chunk_size=static_chunk_size, device=device, **model_kwargs)
    checkpoint = torch.load(model_path, map_location="cpu", weights_only=True)
    model.load_state_dict(checkpoint["model"])
    feats = model.extract_audio_features(wav[0], rate)         # mono: first channel, as torchaudio.load + squeeze(0) for 1-channel files
    if verbose:
        print(f"{audio_file}: {wav.shape[1]} samples at {rate} Hz -> features {tuple(feats.shape)}; epoch {checkpoint.get('epoch', -1) + 1}")
    return decode_features(model, feats, static_chunk_size, beam_size, tokenizer, verbose)


def main():
    ap = argparse.ArgumentParser(description="streaming RNN-T decode of one PCM wav file on an MI355X")
    ap.add_argument("audio_file")
    ap.add_argument("--model_path", default="./online_model.pt")
    ap.add_argument("--vocab_size", type=int, required=True)
    ap.add_argument("--blank_id", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=32, help="Config.static_chunk_size of the reference (rnnt_common.py:16)")
    ap.add_argument("--beam_size", type=int, default=4)
    a = ap.parse_args()
    decode_single_audio(a.audio_file, a.model_path, a.vocab_size, a.blank_id, a.chunk, a.beam_size)


if __name__ == "__main__":
    main()
