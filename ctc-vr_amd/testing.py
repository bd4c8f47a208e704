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
