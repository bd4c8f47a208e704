"""CPU ORACLE for the streaming RNN-T inference path.  TEST INFRASTRUCTURE ONLY.

This file restates, op by op, the algorithm of the reference's streaming RNN-Transducer
path (CentaureaHO/CTC-VR, Python/PyTorch) on plain torch-CPU float32 tensors held in a flat
state dict.  It contains no nn.Module tree, imports nothing from /root/reference, and is
never imported by the product package: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg use it, and only as the checker / CPU baseline, never as the thing shipped.

PARITY PIN: the reference holds no tests or golden vectors for this path, so
the oracle is pinned against outputs of the reference itself, imported in the build container
by tests/golden/gen_golden.py (with the four ordinary import shims of SURVEY.md §8c) on
seeded weights; the resulting vectors are committed under tests/golden/*.npz and checked by
tests/test_oracle_golden.py.

Every function cites the reference file:line it follows (paths relative to the reference
root).  All arithmetic is float32; tokens and indices are Python ints, beam scores are Python
floats (double) exactly as in the reference.
"""
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

NUM_LAYERS = 12      # model/online_rnnt_model.py:90
HEADS = 4            # model/online_rnnt_model.py:88
LORDER = 30          # causal depthwise k=31 -> lorder 30 (wenet/transformer/convolution.py:58-60)
LN_EPS = 1e-5        # wenet/transformer/encoder.py:55
BN_EPS = 1e-5        # wenet/transformer/convolution.py:36


def to_torch_sd(np_sd) -> SD:
    return {k: torch.from_numpy(v) for k, v in np_sd.items()}


# --------------------------------------------------------------------------------------
# encoder pieces
# --------------------------------------------------------------------------------------
def subsample(sd: SD, xs: Tensor) -> Tensor:
    """Conv2dSubsampling4.forward (wenet/transformer/subsampling.py:203-228) followed by the
    x*sqrt(d_model) of RelPositionalEncoding.forward (wenet/transformer/embedding.py:145).
    xs [B,T,80] -> [B,t',256] with t' = ((T-3)//2+1-3)//2+1; pos_emb is NOT added for rel_pos."""
    x = xs.unsqueeze(1)                                                         # :224
    x = F.relu(F.conv2d(x, sd["encoder.embed.conv.0.weight"], sd["encoder.embed.conv.0.bias"], stride=2))
    x = F.relu(F.conv2d(x, sd["encoder.embed.conv.2.weight"], sd["encoder.embed.conv.2.bias"], stride=2))
    b, c, t, f = x.size()
    x = x.transpose(1, 2).contiguous().view(b, t, c * f)                        # :227
    x = F.linear(x, sd["encoder.embed.out.0.weight"], sd["encoder.embed.out.0.bias"])
    return x * math.sqrt(x.size(-1))                                            # embedding.py:145


def position_encoding(sd: SD, offset: int, size: int) -> Tensor:
    """PositionalEncoding.position_encoding, int-offset branch (embedding.py:101-103)."""
    pe = sd["encoder.embed.pos_enc.pe"]
    assert offset + size <= pe.size(1)
    assert offset >= 0  # the reference would silently wrap a negative python slice; never reached
    return pe[:, offset:offset + size]


def _ln(sd: SD, name: str, x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.size(-1),), sd[name + ".weight"], sd[name + ".bias"], LN_EPS)


def feed_forward(sd: SD, p: str, x: Tensor) -> Tensor:
    """PositionwiseFeedForward.forward with SiLU (positionwise_feed_forward.py:50-58)."""
    h = F.silu(F.linear(x, sd[p + ".w_1.weight"], sd[p + ".w_1.bias"]))
    return F.linear(h, sd[p + ".w_2.weight"], sd[p + ".w_2.bias"])


def rel_attention(sd: SD, p: str, x: Tensor, pos_emb: Tensor,
                  k_cache: Optional[Tensor], v_cache: Optional[Tensor],
                  mask: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """RelPositionMultiHeadedAttention.forward (wenet/transformer/attention.py:364-420), eval
    mode, no rel_shift (:407-409), with forward_qkv (:109-131), _update_kv_and_cache (:180-245)
    and forward_attention (:133-178).
    x [B,t,256]; pos_emb [1,T2,256]; caches [B,4,Tc,64] or None.  mask: None (chunk mode,
    'fake mask' branch :169-171) or bool [B,1,T2] / [B,t,T2] (full-context branch :158-164).
    Returns (out [B,t,256], K [B,4,T2,64], V [B,4,T2,64])."""
    B, t, D = x.shape
    dk = D // HEADS
    q = F.linear(x, sd[p + ".linear_q.weight"], sd[p + ".linear_q.bias"]).view(B, t, HEADS, dk)
    k = F.linear(x, sd[p + ".linear_k.weight"], sd[p + ".linear_k.bias"]).view(B, t, HEADS, dk).transpose(1, 2)
    v = F.linear(x, sd[p + ".linear_v.weight"], sd[p + ".linear_v.bias"]).view(B, t, HEADS, dk).transpose(1, 2)
    if k_cache is not None:                                                      # :207-211
        k = torch.cat([k_cache, k], dim=2)
        v = torch.cat([v_cache, v], dim=2)
    pp = F.linear(pos_emb, sd[p + ".linear_pos.weight"]).view(pos_emb.size(0), -1, HEADS, dk).transpose(1, 2)  # :395-397
    q_u = (q + sd[p + ".pos_bias_u"]).transpose(1, 2)                            # :400
    q_v = (q + sd[p + ".pos_bias_v"]).transpose(1, 2)                            # :402
    matrix_bd = torch.matmul(q_v, pp.transpose(-2, -1))                          # :406
    matrix_ac = torch.matmul(q_u, k.transpose(-2, -1))                           # :415
    scores = (matrix_ac + matrix_bd) / math.sqrt(dk)                             # :417
    if mask is not None and mask.size(-1) > 0:                                   # :158-164
        m = mask.unsqueeze(-3).eq(0)[..., :scores.size(-1)]
        scores = scores.masked_fill(m, -float("inf"))
        attn = torch.softmax(scores.float(), dim=-1).masked_fill(m, 0.0)
    else:
        attn = torch.softmax(scores.float(), dim=-1)                             # :170
    o = torch.matmul(attn, v).transpose(1, 2).contiguous().view(B, t, D)          # :174-177
    return F.linear(o, sd[p + ".linear_out.weight"], sd[p + ".linear_out.bias"]), k, v


def conv_module(sd: SD, p: str, x: Tensor, cache: Optional[Tensor],
                mask_pad: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """ConvolutionModule.forward, causal, BatchNorm1d in eval mode, SiLU
    (wenet/transformer/convolution.py:98-153).  x [B,t,256]; cache [B,256,30] or None (then the
    input is zero-padded on the left BEFORE pointwise_conv1, :122-124).  mask_pad bool [B,1,t]
    or None.  Returns (y [B,t,256], new_cache [B,256,30])."""
    x = x.transpose(1, 2)
    if mask_pad is not None and mask_pad.size(2) > 0:                            # :117-118
        x = x.masked_fill(~mask_pad, 0.0)
    if cache is None or cache.size(2) == 0:                                      # :122-124
        x = F.pad(x, (LORDER, 0), "constant", 0.0)
    else:
        x = torch.cat((cache, x), dim=2)                                         # :128
    new_cache = x[:, :, -LORDER:]                                                # :130
    x = F.conv1d(x, sd[p + ".pointwise_conv1.weight"], sd[p + ".pointwise_conv1.bias"])
    x = F.glu(x, dim=1)                                                          # :139
    x = F.conv1d(x, sd[p + ".depthwise_conv.weight"], sd[p + ".depthwise_conv.bias"], groups=x.size(1))
    x = F.batch_norm(x, sd[p + ".norm.running_mean"], sd[p + ".norm.running_var"],
                     sd[p + ".norm.weight"], sd[p + ".norm.bias"], False, 0.0, BN_EPS)
    x = F.silu(x)                                                                # :145
    x = F.conv1d(x, sd[p + ".pointwise_conv2.weight"], sd[p + ".pointwise_conv2.bias"])
    if mask_pad is not None and mask_pad.size(2) > 0:                            # :150-151
        x = x.masked_fill(~mask_pad, 0.0)
    return x.transpose(1, 2), new_cache


def conformer_layer(sd: SD, i: int, x: Tensor, pos_emb: Tensor,
                    k_cache: Optional[Tensor], v_cache: Optional[Tensor], cnn_cache: Optional[Tensor],
                    mask: Optional[Tensor] = None, mask_pad: Optional[Tensor] = None,
                    trace: Optional[dict] = None):
    """ConformerEncoderLayer.forward, normalize_before=True, macaron, dropout off
    (wenet/transformer/encoder_layer.py:188-265)."""
    p = f"encoder.encoders.{i}"
    x = x + 0.5 * feed_forward(sd, p + ".feed_forward_macaron", _ln(sd, p + ".norm_ff_macaron", x))   # :216-223
    if trace is not None:
        trace["after_ffm"] = x
    att, k, v = rel_attention(sd, p + ".self_attn", _ln(sd, p + ".norm_mha", x), pos_emb, k_cache, v_cache, mask)
    x = x + att                                                                                     # :226-233
    if trace is not None:
        trace["after_mha"] = x
    y, new_cnn = conv_module(sd, p + ".conv_module", _ln(sd, p + ".norm_conv", x), cnn_cache, mask_pad)
    x = x + y                                                                                       # :238-244
    if trace is not None:
        trace["after_conv"] = x
    x = x + 0.5 * feed_forward(sd, p + ".feed_forward", _ln(sd, p + ".norm_ff", x))                   # :250-255
    x = _ln(sd, p + ".norm_final", x)                                                               # :259-260
    if trace is not None:
        trace["out"] = x
    return x, k, v, new_cnn


def forward_chunk(sd: SD, xs: Tensor, offset: int, required_cache_size: int,
                  att_cache: Tensor, cnn_cache: Tensor, trace: Optional[dict] = None):
    """BaseEncoder.forward_chunk (wenet/transformer/encoder.py:203-299).
    xs [B,T,80]; att_cache [12,4,Tc,128] (zeros(0,0,0,0) at stream start), cnn_cache
    [12,B,256,30] (or zeros(0,0,0,0)).  The reference asserts B == 1 (:242); the restatement
    keeps the same tensor layout, so B == 1 is required whenever att_cache is non-empty."""
    x = subsample(sd, xs)                                                        # :252
    elayers, cache_t1 = att_cache.size(0), att_cache.size(2)
    chunk_size = x.size(1)
    attention_key_size = cache_t1 + chunk_size
    pos_emb = position_encoding(sd, offset - cache_t1, attention_key_size)       # :257-258
    if required_cache_size < 0:                                                  # :259-264
        next_cache_start = 0
    elif required_cache_size == 0:
        next_cache_start = attention_key_size
    else:
        next_cache_start = max(attention_key_size - required_cache_size, 0)
    if trace is not None:
        trace["embed"] = x
        trace["pos_start"] = offset - cache_t1
        trace["next_cache_start"] = next_cache_start
    r_att, r_cnn = [], []
    for i in range(NUM_LAYERS):
        if elayers == 0:                                                         # :271-272
            kc = vc = None
        else:
            size = att_cache.size(-1) // 2
            kc, vc = att_cache[i:i + 1, :, :, :size], att_cache[i:i + 1, :, :, size:]
        cc = cnn_cache[i] if cnn_cache.size(0) > 0 else None                     # :283
        lt = {} if (trace is not None and i in trace.get("layers", ())) else None
        x, k, v, new_cnn = conformer_layer(sd, i, x, pos_emb, kc, vc, cc, trace=lt)
        if lt is not None:
            trace[f"layer{i}"] = lt
        r_att.append(torch.cat((k, v), dim=-1)[:, :, next_cache_start:, :])      # :284,288
        r_cnn.append(new_cnn.unsqueeze(0))
    x = _ln(sd, "encoder.after_norm", x)                                         # :290-291
    return x, torch.cat(r_att, dim=0), torch.cat(r_cnn, dim=0)


def encoder_full(sd: SD, xs: Tensor, xs_lens: Tensor) -> Tuple[Tensor, Tensor]:
    """BaseEncoder.forward with decoding_chunk_size=-1 (full context; the only deterministic
    non-streaming call) (wenet/transformer/encoder.py:121-180) with
    make_pad_mask (wenet/utils/mask.py:201-227) and add_optional_chunk_mask's full-context
    branch (mask.py:170-172,194-195 -> chunk_masks = masks).  Returns (ys [B,T',256], masks [B,1,T'])."""
    T = xs.size(1)
    masks = (torch.arange(T).unsqueeze(0) < xs_lens.unsqueeze(1)).unsqueeze(1)    # ~make_pad_mask
    x = subsample(sd, xs)
    masks = masks[:, :, 2::2][:, :, 2::2]                                         # subsampling.py:228
    pos_emb = position_encoding(sd, 0, x.size(1))
    for i in range(NUM_LAYERS):
        x, _, _, _ = conformer_layer(sd, i, x, pos_emb, None, None, None, mask=masks, mask_pad=masks)
    return _ln(sd, "encoder.after_norm", x), masks


def basic_greedy_search_full(sd, x, lens, blank=5, n_steps=64):
    """model/component/transducer.py:22-70 as reached from OnlineRNNTModel.forward (:234-235,268) for a non-streaming
    model: full-context encoder, then per utterance the greedy loop over its valid frames -- fresh zero predictor state,
    previous token = blank, up to n_steps symbols per frame, state/token advance only on non-blank, leave the inner loop
    on blank.  x [B,T,80], lens [B] -> list of token lists."""
    y, mask = encoder_full(sd, x, lens)
    out_lens = mask.squeeze(1).sum(1)
    hyps = []
    for b in range(x.shape[0]):
        hyp = []
        h = torch.zeros(1, 1, 256)
        c = torch.zeros(1, 1, 256)
        tok = blank
        for t in range(int(out_lens[b])):
            enc_t = y[b:b + 1, t:t + 1, :]
            for _ in range(n_steps):
                pred, (h2, c2) = predictor_step(sd, torch.tensor([[tok]]), (h, c))
                k = int(torch.argmax(joint(sd, enc_t, pred).reshape(-1)))
                if k == blank:
                    break
                hyp.append(k)
                tok, h, c = k, h2, c2
        hyps.append(hyp)
    return hyps


def ctc_greedy_search_full(sd: SD, xs: Tensor, xs_lens: Tensor, blank: int) -> List[List[int]]:
    """OnlineRNNTModel.ctc_greedy_search (model/online_rnnt_model.py:647-671) on the deterministic full-context
    encoder (the reference's own call uses a random dynamic chunk mask in eval, SURVEY.md §0.8, so only this variant
    can be pinned): argmax of log_softmax(ctc_lo(enc)) per frame, then drop blanks and repeats over the valid frames.
    Parity note: pinned only through encoder_full (golden) + torch Linear/argmax; the reference has no fixture for it."""
    enc, mask = encoder_full(sd, xs, xs_lens)
    lp = torch.log_softmax(F.linear(enc, sd["ctc_head.ctc_lo.weight"], sd["ctc_head.ctc_lo.bias"]), dim=2)
    am = torch.argmax(lp, dim=2)
    hyps = []
    for b in range(xs.size(0)):
        n = int(mask[b].squeeze().sum().item())
        hyp, prev = [], -1
        for t in range(n):
            tok = int(am[b, t])
            if tok != blank and tok != prev:
                hyp.append(tok)
            prev = tok
        hyps.append(hyp)
    return hyps


def _log_add(args: List[float]) -> float:
    """wenet.utils.common.log_add as its prefix-beam-search call site uses it: stable log-sum-exp of a LIST of Python floats
    (wenet/transducer/search/prefix_beam_search.py:136-138 passes one list; the vendored common.py:302-310 declares *args, so the
    reference raises TypeError whenever two prefixes merge -- the list form is upstream WeNet's and the only one the call fits)."""
    if all(a == -float("inf") for a in args):
        return -float("inf")
    a_max = max(args)
    return a_max + math.log(sum(math.exp(a - a_max) for a in args))


def prefix_beam_search_full(sd: SD, xs: Tensor, xs_lens: Tensor, blank: int, beam_size: int = 5, ctc_weight: float = 0.3,
                            transducer_weight: float = 0.7):
    """PrefixBeamSearch.prefix_beam_search (wenet/transducer/search/prefix_beam_search.py:42-148) on the full-context encoder
    (decoding_chunk_size=-1), B = 1: one symbol at most per frame; per frame every hypothesis runs one predictor step on its last
    token, joint + log_softmax, shallow fusion with the CTC posterior log(tw * exp(logp) + cw * exp(ctc[i])) (:99-101), top-beam
    per hypothesis (:104), blank keeps hypothesis and predictor state, a token extends both (:110-126), equal prefixes are merged
    with log_add keeping the first one's state (:128-141), stable sort by score, truncate (:144-145).
    Returns [(tokens incl. the leading blank, score, [h, c])]."""
    assert xs.size(0) == 1
    enc, _ = encoder_full(sd, xs, xs_lens)
    ctc = torch.log_softmax(F.linear(enc, sd["ctc_head.ctc_lo.weight"], sd["ctc_head.ctc_lo.bias"]), dim=2).squeeze(0)
    beam = [([blank], 0.0, predictor_init_state(1))]
    for i in range(enc.size(1)):
        toks = torch.tensor([b[0][-1] for b in beam], dtype=torch.long)
        state = [torch.cat([b[2][0] for b in beam], 1), torch.cat([b[2][1] for b in beam], 1)]
        scores = torch.tensor([b[1] for b in beam])                                     # float32, as torch.tensor(list of floats)
        pred, new_state = predictor_step(sd, toks.view(-1, 1), state)                    # [N,1,256]
        logp = joint(sd, enc[:, i:i + 1, :], pred).log_softmax(dim=-1).squeeze(1).squeeze(1)   # [N, V]
        logp = torch.log(torch.add(transducer_weight * torch.exp(logp), ctc_weight * torch.exp(ctc[i].unsqueeze(0))))
        top_lp, top_ix = logp.topk(beam_size)
        sc = torch.add(scores.unsqueeze(1), top_lp)
        cand = []
        for j, (hyp, _, st) in enumerate(beam):
            for t in range(beam_size):
                if int(top_ix[j, t]) == blank:
                    cand.append([list(hyp), sc[j, t].item(), st])
                else:
                    cand.append([list(hyp) + [int(top_ix[j, t])], sc[j, t].item(), [new_state[0][:, j:j + 1], new_state[1][:, j:j + 1]]])
        fused = [cand[0]]
        for c in cand[1:]:
            for f in fused:
                if c[0] == f[0]:
                    f[1] = _log_add([f[1], c[1]])
                    break
            else:
                fused.append(c)
        fused.sort(key=lambda v: v[1], reverse=True)
        beam = [(f[0], f[1], f[2]) for f in fused[:beam_size]]
    return beam


# --------------------------------------------------------------------------------------
# predictor / joint
# --------------------------------------------------------------------------------------
def predictor_init_state(batch: int = 1) -> List[Tensor]:
    """RNNPredictor.init_state (wenet/transducer/predictor.py:165-183): [h, c] each [1,B,256]."""
    return [torch.zeros(1, batch, 256), torch.zeros(1, batch, 256)]


def predictor_step(sd: SD, tok: Tensor, state: List[Tensor]) -> Tuple[Tensor, List[Tensor]]:
    """RNNPredictor.forward_step with padding == 0 (predictor.py:185-210): Embedding -> one LSTM
    cell (torch gate order i,f,g,o) -> Linear.  tok int64 [B,1]; returns (out [B,1,256], [h,c])."""
    x = F.embedding(tok, sd["predictor.embed.weight"])[:, 0]                     # [B,256]
    h, c = state[0][0], state[1][0]
    gates = (F.linear(x, sd["predictor.rnn.weight_ih_l0"], sd["predictor.rnn.bias_ih_l0"])
             + F.linear(h, sd["predictor.rnn.weight_hh_l0"], sd["predictor.rnn.bias_hh_l0"]))
    i, f, g, o = gates.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    out = F.linear(h2, sd["predictor.projection.weight"], sd["predictor.projection.bias"])
    return out.unsqueeze(1), [h2.unsqueeze(0), c2.unsqueeze(0)]


def joint(sd: SD, enc_out: Tensor, pred_out: Tensor) -> Tensor:
    """TransducerJoint.forward (model/component/joint.py:48-69): raw LOGITS [B,T,U,V]."""
    e = F.linear(enc_out, sd["joint.enc_ffn.weight"], sd["joint.enc_ffn.bias"]).unsqueeze(2)
    p = F.linear(pred_out, sd["joint.pred_ffn.weight"], sd["joint.pred_ffn.bias"]).unsqueeze(1)
    return F.linear(torch.tanh(e + p), sd["joint.ffn_out.weight"], sd["joint.ffn_out.bias"])


# --------------------------------------------------------------------------------------
# decoding loops
# --------------------------------------------------------------------------------------
class BeamHypothesis:
    """model/online_rnnt_model.py:41-55."""

    def __init__(self, tokens, log_prob, predictor_states=None):
        self.tokens, self.log_prob, self.predictor_states = tokens, log_prob, predictor_states

    def __lt__(self, other):
        return self.log_prob < other.log_prob

    def copy(self):
        return BeamHypothesis(self.tokens.copy(), self.log_prob, self.predictor_states)


def greedy_frames(sd: SD, encoder_out: Tensor, state: Optional[List[Tensor]], prev_token: int,
                  blank: int, n_steps: int = 10, margins: Optional[list] = None):
    """Inner loops of OnlineRNNTModel._decode_chunk_streaming_logic
    (model/online_rnnt_model.py:183-222): argmax on LOGITS (:212), first-max index on ties,
    token/LSTM state advance only on non-blank (:218-220), at most n_steps symbols per frame."""
    if state is None:
        state = predictor_init_state(encoder_out.size(0))
    tok = prev_token
    hyp = []
    for t in range(encoder_out.size(1)):
        enc_t = encoder_out[:, t:t + 1, :]
        for _ in range(n_steps):
            pred, nxt = predictor_step(sd, torch.tensor([[tok]], dtype=torch.long), state)
            logits = joint(sd, enc_t, pred).squeeze(0).squeeze(0).squeeze(0)
            if margins is not None:
                top2 = torch.topk(logits, 2).values
                margins.append(float(top2[0] - top2[1]))
            k = int(torch.argmax(logits).item())
            if k == blank:
                break
            hyp.append(k)
            tok = k
            state = nxt
    return hyp, state, tok


def beam_frames(sd: SD, encoder_out: Tensor, beam_in: Optional[List[BeamHypothesis]], blank: int,
                beam_size: int = 4, n_steps: int = 10, stats: Optional[dict] = None):
    """Per-frame hypothesis expansion of OnlineRNNTModel._decode_chunk_beam_search
    (model/online_rnnt_model.py:407-522): per live hypothesis a greedy extension chain of at most
    n_steps; every step pushes a blank candidate (old state) and top-`beam` non-blank candidates
    (new state); stop when blank is within 1e-6 of the max (:486); Python-double scores; stable
    descending sort (:506), first-wins de-dup on the token tuple (:508-516)."""
    if beam_in is None:
        st = predictor_init_state(encoder_out.size(0))
        beam_in = [BeamHypothesis([], 0.0, [s.clone() for s in st])]
    cur = beam_in
    for t in range(encoder_out.size(1)):
        enc_t = encoder_out[:, t:t + 1, :]
        all_c = []
        for hyp in cur:
            toks = hyp.tokens.copy()
            lp = hyp.log_prob
            st = [s.clone() for s in hyp.predictor_states]
            last = toks[-1] if toks else blank                                   # :429
            for _ in range(n_steps):
                pred, nxt = predictor_step(sd, torch.tensor([[last]], dtype=torch.long), st)
                logp = torch.log_softmax(joint(sd, enc_t, pred).squeeze(), dim=-1)   # :446-447
                if stats is not None:
                    stats["evals"] = stats.get("evals", 0) + 1
                blank_lp = logp[blank].item()
                all_c.append(BeamHypothesis(toks.copy(), lp + blank_lp, [s.clone() for s in st]))
                nb_mask = torch.ones_like(logp, dtype=torch.bool)
                nb_mask[blank] = False
                nb = logp[nb_mask]
                nb_idx = torch.arange(logp.size(0))[nb_mask]
                k = min(beam_size, nb.size(0))
                top_lp, top_i = torch.topk(nb, k)                                # :468
                top_tok = nb_idx[top_i]
                for j in range(k):
                    all_c.append(BeamHypothesis(toks + [top_tok[j].item()], lp + top_lp[j].item(),
                                                [s.clone() for s in nxt]))
                if blank_lp >= logp.max().item() - 1e-6:                          # :486
                    break
                bi = torch.argmax(nb)                                            # :490
                toks.append(nb_idx[bi].item())
                lp += nb[bi].item()
                st = nxt
                last = toks[-1]
        all_c.sort(key=lambda h: h.log_prob, reverse=True)                       # :506
        uniq, seen = [], set()
        for c in all_c:
            tt = tuple(c.tokens)
            if tt not in seen:
                uniq.append(c)
                seen.add(tt)
                if len(uniq) >= beam_size:
                    break
        cur = uniq[:beam_size]
    return cur


class OracleStream:
    """State machine of OnlineRNNTModel's streaming methods for ONE stream
    (model/online_rnnt_model.py:138-164 state, :274-387 greedy, :534-645 beam)."""

    def __init__(self, sd: SD, blank: int = 5, static_chunk_size: int = 32):
        self.sd, self.blank_id, self.static_chunk_size = sd, blank, static_chunk_size
        self.subsampling_rate = 4
        self.reset_streaming_cache()

    def reset_streaming_cache(self):                                             # :145-164
        self.att_cache = torch.zeros((0, 0, 0, 0))
        self.cnn_cache = torch.zeros((0, 0, 0, 0))
        self.predictor_states = None
        self.last_token = self.blank_id
        self.beam = None
        self.global_offset = 0

    def _greedy_chunk(self, chunk, offset, required, trace=None, margins=None):  # :166-222
        enc, self.att_cache, self.cnn_cache = forward_chunk(
            self.sd, chunk, offset, required, self.att_cache, self.cnn_cache, trace)
        if trace is not None:
            trace["enc_out"] = enc
        hyp, self.predictor_states, self.last_token = greedy_frames(
            self.sd, enc, self.predictor_states, self.last_token, self.blank_id, margins=margins)
        return hyp

    def _beam_chunk(self, chunk, offset, required, beam_size, stats=None):       # :389-522
        enc, self.att_cache, self.cnn_cache = forward_chunk(
            self.sd, chunk, offset, required, self.att_cache, self.cnn_cache)
        self.beam = beam_frames(self.sd, enc, self.beam, self.blank_id, beam_size, stats=stats)
        return self.beam

    def process_single_chunk(self, chunk: Tensor, trace=None, margins=None) -> List[int]:   # :346-387
        assert chunk.size(0) == 1
        if chunk.size(1) < 7:
            return []
        off = self.global_offset
        hyp = self._greedy_chunk(chunk, off, off, trace, margins)
        self.global_offset += chunk.size(1) // self.subsampling_rate             # :384-385
        return hyp

    def process_single_chunk_beam_search(self, chunk: Tensor, beam_size: int = 4, stats=None):  # :605-645
        assert chunk.size(0) == 1
        if chunk.size(1) < 7:
            return self.beam or []
        off = self.global_offset
        beam = self._beam_chunk(chunk, off, off, beam_size, stats)
        self.global_offset += chunk.size(1) // self.subsampling_rate
        return beam

    def _utt_chunks(self, n_frames: int, chunk_size_ms: Optional[int]):
        """Chunking of streaming_inference / streaming_beam_search (:283-322, :543-580):
        yields (start, end, encoder_offset) for the chunks that are actually decoded."""
        frames = (self.static_chunk_size if self.static_chunk_size > 0 else 16) * self.subsampling_rate
        if chunk_size_ms is not None:
            frames = int(chunk_size_ms / 10)
        min_frames = max(16, self.subsampling_rate * 4)
        if frames < min_frames:
            if n_frames >= min_frames:
                frames = min_frames
            else:
                if n_frames < 7:
                    return
                frames = n_frames
        cur = 0
        while cur < n_frames:
            end = min(cur + frames, n_frames)
            if end - cur == 0:
                break
            if end - cur >= 7:
                yield cur, end, cur // self.subsampling_rate                      # :324
            cur = end

    def streaming_inference(self, audio: Tensor, n_frames: int, chunk_size_ms=None) -> List[int]:   # :274-344
        assert audio.size(0) == 1
        self.reset_streaming_cache()
        full = []
        for s, e, off in self._utt_chunks(n_frames, chunk_size_ms):
            full.extend(self._greedy_chunk(audio[:, s:e, :], off, off))
        return full

    def streaming_beam_search(self, audio: Tensor, n_frames: int, beam_size=4, chunk_size_ms=None) -> List[int]:  # :534-603
        assert audio.size(0) == 1
        self.reset_streaming_cache()
        for s, e, off in self._utt_chunks(n_frames, chunk_size_ms):
            self._beam_chunk(audio[:, s:e, :], off, off, beam_size)
        if self.beam:
            return max(self.beam, key=lambda h: h.log_prob).tokens                # :598-601
        return []


def decode_script_greedy(sd: SD, audio: Tensor, chunk_frames: int, blank: int = 5):
    """Greedy chunk loop of online_rnnt_decode.py:81-117 (chunk = Config.static_chunk_size INPUT
    frames, tail-merge rule :88-91).  Returns (all tokens, per-chunk token lists)."""
    st = OracleStream(sd, blank, chunk_frames)
    T = audio.size(1)
    toks, per = [], []
    off = 0
    min_chunk = max(16, chunk_frames)
    while off < T:
        end = min(off + chunk_frames, T)
        if T - end < min_chunk and end < T:
            end = T
        r = st.process_single_chunk(audio[:, off:end, :])
        per.append(r)
        toks.extend(r)
        off = end
        if end >= T:
            break
    return toks, per, st
