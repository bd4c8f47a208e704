"""Deterministic synthetic weights and fbank for parity tests, smoke and bench.

No trained checkpoint ships with the reference (reference .gitignore:10,13 excludes
online_model.pt), so every parity artefact is pinned on seeded weights generated here
with NumPy's Philox bit generator (stream-stable across platforms).  The key set and
shapes are exactly the 504-entry state dict of the reference's OnlineRNNTModel
(model/online_rnnt_model.py:58-143).

Nothing here touches /root/reference or oracle/.
"""
import math
import numpy as np

from .layout import D, H, FF, L, KDW, IDIM, FSUB, VOCAB, BLANK, MAX_LEN, state_dict_spec, chunk_plan  # noqa: F401  (re-exported for the tests)


def positional_table(max_len=MAX_LEN, d=D):
    """pe[pos,2i]=sin(pos/10000^(2i/d)), pe[pos,2i+1]=cos(..) in float32
    (wenet/transformer/embedding.py:50-58).  It is a persistent buffer of the
    reference state dict, so both sides load THIS table."""
    pos = np.arange(max_len, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0, d, 2, dtype=np.float32) * np.float32(-(math.log(10000.0) / d)))
    ang = (pos * div).astype(np.float32)
    pe = np.zeros((max_len, d), np.float32)
    pe[:, 0::2] = np.sin(ang)
    pe[:, 1::2] = np.cos(ang)
    return pe[None]


# Per-key gains on top of the 1/sqrt(fan_in) base scale.  Calibrated against the imported
# reference (tests/golden/gen_golden.py prints the achieved rates): residual-branch output
# projections are damped so frame-to-frame variation survives 12 blocks (un-damped random
# blocks collapse every frame onto one vector and greedy emits a single token forever);
# the joint/predictor gains make the logits depend on both encoder frame and label history.
GAINS = (
    (("w_2.weight", "linear_out.weight", "pointwise_conv2.weight"), 0.3),
    (("joint.enc_ffn.weight", "joint.pred_ffn.weight"), 4.0),
    (("predictor.embed.weight",), 2.0),
    (("predictor.rnn.weight_ih_l0", "predictor.rnn.weight_hh_l0"), 4.0),
)


def make_state_dict(seed=0, vocab=VOCAB, blank=BLANK, blank_bias=11.0, out_gain=4.0):
    """504-key state dict as float32 numpy arrays (num_batches_tracked: int64 scalar).

    `out_gain` widens the logit spread so greedy top-2 margins (min ~3e-3 on the fixtures)
    stay far above fp32 rounding; `blank_bias` is added to ffn_out.bias[blank] so greedy
    emits on the order of one symbol per encoder frame (un-biased random weights give the
    10-symbols-per-frame worst case)."""
    sd = {}
    for idx, (name, shape, kind) in enumerate(state_dict_spec(vocab)):
        g = np.random.Generator(np.random.Philox(key=[seed, idx]))
        if kind.startswith("w:"):
            a = g.standard_normal(shape, dtype=np.float32) * np.float32(1.0 / math.sqrt(int(kind[2:])))
        elif kind == "b":
            a = g.standard_normal(shape, dtype=np.float32) * np.float32(0.05)
        elif kind == "g":
            a = np.float32(1.0) + g.standard_normal(shape, dtype=np.float32) * np.float32(0.1)
        elif kind == "pb":
            a = g.standard_normal(shape, dtype=np.float32) * np.float32(0.1)
        elif kind == "bn_mean":
            a = g.standard_normal(shape, dtype=np.float32) * np.float32(0.1)
        elif kind == "bn_var":
            a = g.uniform(0.5, 1.5, shape).astype(np.float32)
        elif kind == "nbt":
            a = np.array(100, dtype=np.int64)
        elif kind == "emb":
            a = g.standard_normal(shape, dtype=np.float32) * np.float32(0.5)
        elif kind == "lstm":
            a = g.uniform(-1.0 / 16, 1.0 / 16, shape).astype(np.float32) * np.float32(2.0)
        elif kind == "out":
            a = g.standard_normal(shape, dtype=np.float32) * np.float32(out_gain / math.sqrt(D))
        elif kind == "outb":
            a = g.standard_normal(shape, dtype=np.float32) * np.float32(0.3)
            a[blank] += np.float32(blank_bias)
        elif kind == "pe":
            a = positional_table()
        else:
            raise ValueError(kind)
        for suffixes, gain in GAINS:
            if name.endswith(suffixes):
                a = a * np.float32(gain)
        sd[name] = np.ascontiguousarray(a)
    return sd


FBANK_MEAN, FBANK_STD, FBANK_MIN, FBANK_MAX = -3.72, 5.01, -15.94, 7.01  # example1.pt stats


def synth_fbank(batch, frames, seed=1234):
    """[batch, frames, 80] float32 with the statistics of the reference's example1.pt."""
    g = np.random.Generator(np.random.Philox(key=[seed, 0xFBA]))
    x = g.standard_normal((batch, frames, IDIM), dtype=np.float32) * np.float32(FBANK_STD) + np.float32(FBANK_MEAN)
    return np.clip(x, FBANK_MIN, FBANK_MAX).astype(np.float32)


def greedy_margins(sd_np, enc_frames, tokens, blank=BLANK, device="cuda", n_steps=10):
    """Top-2 logit margin of every decision of a greedy decode (a token flip must be attributable).

    Teacher-forced replay in float64 torch: `enc_frames` [B, F, 256] are the encoder frames the decode ran on, `tokens` the emitted
    tokens per stream; the predictor states come from the LSTM recurrence over each stream's own tokens
    (wenet/transducer/predictor.py:185-210), then the greedy walk (model/online_rnnt_model.py:193-220) evaluates
    joint(enc[t], pred[u]) at every cell it visits.  Returns (min margin per stream [B] float64 numpy, replay_ok [B] bool numpy:
    the replay emitted exactly `tokens`)."""
    import torch
    B = len(tokens)
    dev = torch.device(device)
    enc = torch.as_tensor(np.asarray(enc_frames)).to(dev, torch.float64)
    W = {k: torch.from_numpy(np.asarray(v, np.float32)).to(dev, torch.float64) for k, v in sd_np.items() if k.startswith(("predictor.", "joint."))}
    F_ = enc.size(1)
    nmax = max([len(t) for t in tokens] + [0])
    tk = torch.full((B, nmax + 1), blank, dtype=torch.long, device=dev)       # input token of predictor step u: blank, then the emitted tokens
    for b, t in enumerate(tokens):
        if t:
            tk[b, 1:len(t) + 1] = torch.tensor(t, device=dev)
    h = torch.zeros(B, D, dtype=torch.float64, device=dev)
    c = torch.zeros_like(h)
    P = torch.empty(B, nmax + 1, D, dtype=torch.float64, device=dev)
    bias = W["predictor.rnn.bias_ih_l0"] + W["predictor.rnn.bias_hh_l0"]
    for u in range(nmax + 1):
        g = W["predictor.embed.weight"][tk[:, u]] @ W["predictor.rnn.weight_ih_l0"].T + h @ W["predictor.rnn.weight_hh_l0"].T + bias
        i_, f_, g_, o_ = g.chunk(4, dim=1)
        c = torch.sigmoid(f_) * c + torch.sigmoid(i_) * torch.tanh(g_)
        h = torch.sigmoid(o_) * torch.tanh(c)
        P[:, u] = (h @ W["predictor.projection.weight"].T + W["predictor.projection.bias"]) @ W["joint.pred_ffn.weight"].T + W["joint.pred_ffn.bias"]
    E = enc @ W["joint.enc_ffn.weight"].T + W["joint.enc_ffn.bias"]
    ar = torch.arange(B, device=dev)
    t_ = torch.zeros(B, dtype=torch.long, device=dev)
    u_ = torch.zeros_like(t_)
    cnt = torch.zeros_like(t_)
    mmin = torch.full((B,), float("inf"), dtype=torch.float64, device=dev)
    ok = torch.ones(B, dtype=torch.bool, device=dev)
    nt = torch.tensor([len(t) for t in tokens], device=dev)
    for _ in range(F_ + nmax + 2):
        act = t_ < F_
        if not bool(act.any()):
            break
        lg = torch.tanh(E[ar, t_.clamp(max=F_ - 1)] + P[ar, u_.clamp(max=nmax)]) @ W["joint.ffn_out.weight"].T + W["joint.ffn_out.bias"]
        top = lg.topk(2, dim=1)
        k = top.indices[:, 0]
        mmin = torch.where(act, torch.minimum(mmin, top.values[:, 0] - top.values[:, 1]), mmin)
        emit = act & (k != blank)
        ok &= ~emit | ((u_ < nt) & (tk[ar, (u_ + 1).clamp(max=nmax)] == k))
        u_ = u_ + emit.long()
        cnt = cnt + emit.long()
        adv = act & ((k == blank) | (cnt >= n_steps))
        t_ = t_ + adv.long()
        cnt = torch.where(adv, torch.zeros_like(cnt), cnt)
    ok &= u_ == nt
    return mmin.cpu().numpy(), ok.cpu().numpy()

