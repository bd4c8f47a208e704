"""Layout of the reference's model and driver that the PRODUCT code needs (no test data here): the dimensions of OnlineRNNTModel
(model/online_rnnt_model.py:58-143), its ordered 504-key state dict -- the packed-blob layout of the multi-GPU
weight broadcast (dist.py) and of rnnt_load_packed -- and the chunk slicing rule of online_rnnt_decode.py:87-93."""

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
