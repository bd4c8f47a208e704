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

D = 256          # encoder / predictor / joint width (rnnt_common.py:11)
H = 4            # attention heads            (online_rnnt_model.py:88)
FF = 1024        # linear_units               (online_rnnt_model.py:89)
L = 12           # num_blocks                 (online_rnnt_model.py:90)
KDW = 31         # cnn_module_kernel          (online_rnnt_model.py:104)
IDIM = 80
FSUB = 19        # ((80-1)//2-1)//2           (wenet/transformer/subsampling.py:194)
VOCAB = 412      # tokenizer.size()           (tokenizer/tokenizer.py:53-60)
BLANK = 5        # tokenizer.blk_id()
MAX_LEN = 5000   # positional table           (wenet/transformer/embedding.py:41)


def state_dict_spec(vocab=VOCAB):
    """Ordered (name, shape, kind) list of the reference state dict."""
    s = []
    s += [("encoder.embed.conv.0.weight", (D, 1, 3, 3), "w:9"),
          ("encoder.embed.conv.0.bias", (D,), "b"),
          ("encoder.embed.conv.2.weight", (D, D, 3, 3), "w:2304"),
          ("encoder.embed.conv.2.bias", (D,), "b"),
          ("encoder.embed.out.0.weight", (D, D * FSUB), "w:4864"),
          ("encoder.embed.out.0.bias", (D,), "b"),
          ("encoder.embed.pos_enc.pe", (1, MAX_LEN, D), "pe"),
          ("encoder.after_norm.weight", (D,), "g"),
          ("encoder.after_norm.bias", (D,), "b")]
    for i in range(L):
        p = f"encoder.encoders.{i}."
        s += [(p + "self_attn.pos_bias_u", (H, D // H), "pb"),
              (p + "self_attn.pos_bias_v", (H, D // H), "pb")]
        for n in ("linear_q", "linear_k", "linear_v", "linear_out"):
            s += [(p + f"self_attn.{n}.weight", (D, D), "w:256"),
                  (p + f"self_attn.{n}.bias", (D,), "b")]
        s += [(p + "self_attn.linear_pos.weight", (D, D), "w:256")]
        for n in ("feed_forward", "feed_forward_macaron"):
            s += [(p + f"{n}.w_1.weight", (FF, D), "w:256"),
                  (p + f"{n}.w_1.bias", (FF,), "b"),
                  (p + f"{n}.w_2.weight", (D, FF), "w:1024"),
                  (p + f"{n}.w_2.bias", (D,), "b")]
        s += [(p + "conv_module.pointwise_conv1.weight", (2 * D, D, 1), "w:256"),
              (p + "conv_module.pointwise_conv1.bias", (2 * D,), "b"),
              (p + "conv_module.depthwise_conv.weight", (D, 1, KDW), "w:31"),
              (p + "conv_module.depthwise_conv.bias", (D,), "b"),
              (p + "conv_module.norm.weight", (D,), "g"),
              (p + "conv_module.norm.bias", (D,), "b"),
              (p + "conv_module.norm.running_mean", (D,), "bn_mean"),
              (p + "conv_module.norm.running_var", (D,), "bn_var"),
              (p + "conv_module.norm.num_batches_tracked", (), "nbt"),
              (p + "conv_module.pointwise_conv2.weight", (D, D, 1), "w:256"),
              (p + "conv_module.pointwise_conv2.bias", (D,), "b")]
        for n in ("norm_ff", "norm_mha", "norm_ff_macaron", "norm_conv", "norm_final"):
            s += [(p + f"{n}.weight", (D,), "g"), (p + f"{n}.bias", (D,), "b")]
    s += [("predictor.embed.weight", (vocab, D), "emb"),
          ("predictor.rnn.weight_ih_l0", (4 * D, D), "lstm"),
          ("predictor.rnn.weight_hh_l0", (4 * D, D), "lstm"),
          ("predictor.rnn.bias_ih_l0", (4 * D,), "lstm"),
          ("predictor.rnn.bias_hh_l0", (4 * D,), "lstm"),
          ("predictor.projection.weight", (D, D), "w:256"),
          ("predictor.projection.bias", (D,), "b"),
          ("joint.enc_ffn.weight", (D, D), "w:256"),
          ("joint.enc_ffn.bias", (D,), "b"),
          ("joint.pred_ffn.weight", (D, D), "w:256"),
          ("joint.pred_ffn.bias", (D,), "b"),
          ("joint.ffn_out.weight", (vocab, D), "out"),
          ("joint.ffn_out.bias", (vocab,), "outb"),
          ("ctc_head.ctc_lo.weight", (vocab, D), "w:256"),
          ("ctc_head.ctc_lo.bias", (vocab,), "b")]
    return s


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


def chunk_plan(total_frames, chunk_frames):
    """Chunk slicing rule of online_rnnt_decode.py:87-93,113-116 -> list of (start, end)."""
    out, off = [], 0
    min_chunk = max(16, chunk_frames)
    while off < total_frames:
        end = min(off + chunk_frames, total_frames)
        if total_frames - end < min_chunk and end < total_frames:
            end = total_frames
        out.append((off, end))
        off = end
        if end >= total_frames:
            break
    return out
