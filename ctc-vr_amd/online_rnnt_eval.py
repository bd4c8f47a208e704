"""Character-error-rate harness with the semantics of the reference's online_rnnt_eval.py: `calculate_cer` (:11-56,
Levenshtein table + one fixed back-trace order, so S/D/I come out exactly as the reference counts them) and
`evaluate_streaming` (:59-151: per utterance reset -> streaming_inference, reset -> streaming_beam_search, pooled
(S+D+I)/N for greedy and beam).  Host code only; the decoding itself is the HIP path behind `OnlineRNNTModel`.

Naming follows the reference, including its convention that D counts surplus HYPOTHESIS tokens and I surplus
reference tokens (the first argument is the hypothesis)."""


def calculate_cer(pre_tokens, gt_tokens):
    """-> (cer, S, D, I, N) for one hypothesis / reference pair (online_rnnt_eval.py:11-56)."""
    hyp, ref = list(pre_tokens), list(gt_tokens)
    m, n = len(hyp), len(ref)
    # cost[i][j] = edit distance between hyp[:i] and ref[:j]
    cost = [list(range(n + 1))] + [[i] + [0] * n for i in range(1, m + 1)]
    for i in range(1, m + 1):
        row, up, h = cost[i], cost[i - 1], hyp[i - 1]
        for j in range(1, n + 1):
            row[j] = min(up[j] + 1, row[j - 1] + 1, up[j - 1] + (0 if h == ref[j - 1] else 1))
    # back-trace, the reference's preference order: match, substitution, hypothesis-side step, reference-side step
    subs = dels = ins = 0
    i, j = m, n
    while i > 0 and j > 0:
        if hyp[i - 1] == ref[j - 1]:
            i, j = i - 1, j - 1
        elif cost[i][j] == cost[i - 1][j - 1] + 1:
            subs, i, j = subs + 1, i - 1, j - 1
        elif cost[i][j] == cost[i - 1][j] + 1:
            dels, i = dels + 1, i - 1
        else:
            ins, j = ins + 1, j - 1
    dels += i
    ins += j
    return ((subs + dels + ins) / n if n else 0.0), subs, dels, ins, n


def pooled_cer(hyps, refs):
    """(S+D+I)/N over all pairs; 1.0 when there is no reference token (online_rnnt_eval.py:116-137)."""
    s = d = i = n = 0
    for hyp, ref in zip(hyps, refs):
        _, ss, dd, ii, nn = calculate_cer(hyp, ref)
        s, d, i, n = s + ss, d + dd, i + ii, n + nn
    return (s + d + i) / n if n > 0 else 1.0


def evaluate_streaming(batches, model, tokenizer=None, output_file=None, beam_size=4, verbose=True):
    """batches: iterable of dicts with 'audios' [B,T,80], 'audio_lens' [B], 'texts' [B,U], 'text_lens' [B] (the reference
    dataloader's batch layout).  Every utterance is decoded alone, greedy and beam, each from a fresh stream state.
    Returns (greedy_cer, beam_cer)."""
    refs, greedy, beam = [], [], []
    out = open(output_file, "w", encoding="utf-8") if output_file else None
    try:
        for batch in batches:
            audios, audio_lens, texts, text_lens = batch["audios"], batch["audio_lens"], batch["texts"], batch["text_lens"]
            for j in range(audios.shape[0]):
                a, al = audios[j:j + 1], audio_lens[j:j + 1]
                model.reset_streaming_cache()
                hg, _, _ = model.streaming_inference(a, al)
                hg = hg[0] if hg else []
                model.reset_streaming_cache()
                hb, _, _ = model.streaming_beam_search(a, al, beam_size=beam_size)
                hb = hb[0] if hb else []
                ref = [int(t) for t in texts[j, :int(text_lens[j])].tolist()]
                refs.append(ref)
                greedy.append(list(hg))
                beam.append(list(hb))
                if out and tokenizer is not None:
                    out.write(f"REF:    {tokenizer.decode(ref)}\nGREEDY: {tokenizer.decode(hg)}\nBEAM:   {tokenizer.decode(hb)}\n\n")
    finally:
        if out:
            out.close()
    g, b = pooled_cer(greedy, refs), pooled_cer(beam, refs)
    if verbose:
        print(f"streaming greedy search CER: {g:.4f}")
        print(f"streaming beam search (beam_size={beam_size}) CER: {b:.4f}")
        print(f"CER change greedy -> beam: {g - b:.4f}")
    return g, b
