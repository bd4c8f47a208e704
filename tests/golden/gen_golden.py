"""Generate golden vectors from the REFERENCE ITSELF (run in the build container only).

    python tests/golden/gen_golden.py

Imports /root/reference through _ref_loader (the four ordinary shims of SURVEY.md §8c), loads the
seeded weights of ctc_vr_amd.testing via load_state_dict(strict=True) and records inputs and
expected outputs as small .npz files next to this script.  Only DATA is written (inputs, outputs,
traces); no reference source or bytecode.  /root/reference is never needed to *check* the vectors.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _ref_loader  # noqa: E402
import ctc_vr_amd.testing as T  # noqa: E402

torch.set_num_threads(8)
ref = _ref_loader.load_reference()


def build(seed, chunk):
    net = ref.OnlineRNNTModel(input_dim=80, hidden_dim=256, vocab_size=T.VOCAB, blank_id=T.BLANK,
                              streaming=True, static_chunk_size=chunk, use_dynamic_chunk=True,
                              ctc_weight=0.3, predictor_layers=1, predictor_dropout=0,
                              ctc_dropout_rate=0.1).eval()
    sd = {k: torch.from_numpy(v) for k, v in T.make_state_dict(seed).items()}
    net.load_state_dict(sd, strict=True)
    return net


def f32(x):
    return x.detach().cpu().numpy().astype(np.float32)


def save(name, **kw):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **kw)
    print(f"  wrote {name}: {os.path.getsize(path) / 1024:.0f} KiB")


@torch.no_grad()
def gen_inputs():
    ex = torch.load("/root/reference/example1.pt", weights_only=True)
    lens = ex["audio_lens"].tolist()
    out = {}
    for u in (0, 6, 12):
        out[f"ex{u}"] = f32(ex["audios"][u, :lens[u]])
    save("inputs_example1.npz", **out)
    return {k: torch.from_numpy(v)[None] for k, v in out.items()}


@torch.no_grad()
def gen_modules(seed, inputs):
    """Component-level vectors: subsampling, one forward_chunk with traces, predictor, joint."""
    net = build(seed, 16)
    out = {}
    x = torch.from_numpy(T.synth_fbank(2, 64, seed=7))
    for tc in (16, 24, 41, 64):
        y, pos, _ = net.encoder.embed(x[:1, :tc], torch.ones(1, 1, tc, dtype=torch.bool), 0)
        out[f"subsample_T{tc}"] = f32(y)
    # three consecutive chunks through forward_chunk with hooks on layers 0 and 11
    caps = {}

    def hook(name):
        def fn(mod, inp, outp):
            caps[name] = outp[0] if isinstance(outp, tuple) else outp
        return fn
    hs = []
    for li in (0, 11):
        lay = net.encoder.encoders[li]
        for nm in ("norm_ff_macaron", "feed_forward_macaron", "norm_mha", "self_attn", "norm_conv",
                   "conv_module", "norm_ff", "feed_forward", "norm_final"):
            hs.append(getattr(lay, nm).register_forward_hook(hook(f"l{li}.{nm}")))
    att = torch.zeros(0, 0, 0, 0)
    cnn = torch.zeros(0, 0, 0, 0)
    xin = inputs["ex0"]
    off = 0
    for ci in range(3):
        chunk = xin[:, ci * 16:(ci + 1) * 16]
        y, att, cnn = net.encoder.forward_chunk(chunk, off, off, att, cnn)
        out[f"fc{ci}_out"] = f32(y)
        out[f"fc{ci}_att_cache"] = f32(att)
        out[f"fc{ci}_cnn_cache"] = f32(cnn)
        for k, v in caps.items():
            out[f"fc{ci}_{k}"] = f32(v)
        off += 16 // 4
    for h in hs:
        h.remove()
    # predictor: fixed token string, state carried
    toks = [5, 17, 353, 17, 0, 411, 200, 5, 5, 99]
    st = net.predictor.init_state(1, torch.device("cpu"))
    po, hh, cc = [], [], []
    for t in toks:
        o, st = net.predictor.forward_step(torch.tensor([[t]]), torch.zeros(1, 1, dtype=torch.long), st)
        po.append(f32(o)[0, 0]); hh.append(f32(st[0])[0, 0]); cc.append(f32(st[1])[0, 0])
    out["pred_tokens"] = np.array(toks, np.int64)
    out["pred_out"] = np.stack(po); out["pred_h"] = np.stack(hh); out["pred_c"] = np.stack(cc)
    # joint lattice [2,7,5,V] on seeded hidden states
    g = np.random.Generator(np.random.Philox(key=[seed, 0x701]))
    enc = torch.from_numpy(g.standard_normal((2, 7, 256), dtype=np.float32))
    prd = torch.from_numpy(g.standard_normal((2, 5, 256), dtype=np.float32) * np.float32(0.5))
    out["joint_enc"] = f32(enc); out["joint_pred"] = f32(prd)
    out["joint_logits"] = f32(net.joint(enc, prd))
    out["joint_logp"] = f32(torch.log_softmax(net.joint(enc, prd), dim=-1))
    save(f"modules_seed{seed}.npz", **out)


@torch.no_grad()
def run_decode_script(net, x, chunk, beam=None):
    """online_rnnt_decode.py:81-117 / :123-178 loop, verbatim semantics, on the reference model."""
    net.reset_streaming_cache()
    T_ = x.shape[1]
    min_chunk = max(16, chunk)
    off = 0
    per, encs, offs, evals = [], [], [], 0
    caps = {}
    h = net.encoder.register_forward_hook(lambda m, i, o: None)
    h.remove()
    orig = net.encoder.forward_chunk

    def spy(xs, offset, required_cache_size, att_cache, cnn_cache):
        r = orig(xs=xs, offset=offset, required_cache_size=required_cache_size, att_cache=att_cache,
                 cnn_cache=cnn_cache)
        encs.append(f32(r[0])[0]); offs.append((offset, required_cache_size, att_cache.size(2) if att_cache.dim() == 4 else 0))
        return r
    net.encoder.forward_chunk = spy
    joint_calls = [0]
    jorig = net.joint.forward

    def jspy(*a, **k):
        joint_calls[0] += 1
        return jorig(*a, **k)
    net.joint.forward = jspy
    try:
        while off < T_:
            end = min(off + chunk, T_)
            if T_ - end < min_chunk and end < T_:
                end = T_
            c = x[:, off:end]
            if beam is None:
                r, _, _ = net.process_single_chunk(c, torch.tensor([c.shape[1]]))
                per.append(list(r))
            else:
                hyps, _, _ = net.process_single_chunk_beam_search(c, torch.tensor([c.shape[1]]), beam_size=beam)
                per.append([(list(hh.tokens), float(hh.log_prob)) for hh in hyps])
            off = end
            if end >= T_:
                break
    finally:
        net.encoder.forward_chunk = orig
        net.joint.forward = jorig
    return per, encs, offs, joint_calls[0]


def pack_tokens(per):
    flat = [t for r in per for t in r]
    return np.array(flat, np.int64), np.array([len(r) for r in per], np.int64)


@torch.no_grad()
def gen_streams(inputs):
    syn = torch.from_numpy(T.synth_fbank(2, 1000))
    cases = [
        # name, seed, input, chunk(decode-script semantics), keep_enc
        ("syn0_c16_s0", 0, syn[0:1], 16, True),
        ("syn1_c16_s0", 0, syn[1:2], 16, False),
        ("syn0_c16_s1", 1, syn[0:1], 16, False),
        ("ex0_c32_s0", 0, inputs["ex0"], 32, True),       # BASELINE config 1
        ("ex0_c32_s1", 1, inputs["ex0"], 32, False),
        ("ex6_c16_s0", 0, inputs["ex6"], 16, False),
        ("ex12_c64_s0", 0, inputs["ex12"], 64, False),
        ("ex12_c16_s1", 1, inputs["ex12"], 16, False),
    ]
    nets = {}
    for name, seed, x, chunk, keep in cases:
        net = nets.setdefault((seed, chunk), build(seed, chunk))
        t0 = time.time()
        per, encs, offs, jc = run_decode_script(net, x, chunk)
        toks, counts = pack_tokens(per)
        out = dict(tokens=toks, counts=counts, offsets=np.array(offs, np.int64), joint_calls=np.int64(jc),
                   chunk=np.int64(chunk), seed=np.int64(seed), frames=np.int64(x.shape[1]),
                   att_cache_shape=np.array(net.streaming_att_cache.shape, np.int64),
                   att_cache_l0_last=f32(net.streaming_att_cache[0, :, -3:, :]),
                   att_cache_l11_first=f32(net.streaming_att_cache[11, :, :3, :]),
                   att_cache_sum=np.float64(net.streaming_att_cache.double().sum().item()),
                   cnn_cache=f32(net.streaming_cnn_cache),
                   pred_h=f32(net.streaming_predictor_states[0]), pred_c=f32(net.streaming_predictor_states[1]),
                   last_token=np.int64(net.streaming_last_emitted_token),
                   global_offset=np.int64(net._global_encoder_offset),
                   enc_frames=np.array([e.shape[0] for e in encs], np.int64))
        enc_all = np.concatenate(encs, 0)
        out["enc_checksum"] = np.float64(enc_all.astype(np.float64).sum())
        if keep:
            out["enc_out"] = enc_all
        else:
            out["enc_out_head"] = enc_all[:8]; out["enc_out_tail"] = enc_all[-8:]
        print(f"{name}: {x.shape[1]} frames, {len(per)} chunks, {enc_all.shape[0]} enc frames, {len(toks)} tokens "
              f"({len(toks) / enc_all.shape[0]:.2f}/frame, {len(set(toks.tolist()))} unique), joint calls {jc}, "
              f"att_cache {tuple(net.streaming_att_cache.shape)}, {time.time() - t0:.1f}s")
        save(f"stream_{name}.npz", **out)

    # streaming_inference semantics (chunk = static_chunk_size*4 input frames, offset = in_off//4)
    for name, seed, x, scs, ms in (("si_ex0_scs16_s0", 0, inputs["ex0"], 16, None),
                                   ("si_syn0_scs16_s0", 0, syn[0:1], 16, None),
                                   ("si_ex6_scs32_ms200_s0", 0, inputs["ex6"], 32, 200)):
        net = nets.setdefault((seed, scs), build(seed, scs))
        r, _, _ = net.streaming_inference(x, torch.tensor([x.shape[1]]), chunk_size_ms=ms)
        print(f"{name}: {len(r[0])} tokens, att_cache {tuple(net.streaming_att_cache.shape)}")
        save(f"stream_{name}.npz", tokens=np.array(r[0], np.int64), static_chunk_size=np.int64(scs),
             chunk_size_ms=np.int64(-1 if ms is None else ms), seed=np.int64(seed), frames=np.int64(x.shape[1]),
             att_cache_shape=np.array(net.streaming_att_cache.shape, np.int64),
             att_cache_sum=np.float64(net.streaming_att_cache.double().sum().item()))

    # beam search, decode-script semantics
    for name, seed, x, chunk, beam in (("beam_ex6_c16_s0", 0, inputs["ex6"], 16, 4),
                                       ("beam_ex0_c32_s0", 0, inputs["ex0"], 32, 4),
                                       ("beam_syn0_c16_s1_f320", 1, syn[0:1, :320], 16, 4)):
        net = nets.setdefault((seed, chunk), build(seed, chunk))
        t0 = time.time()
        per, encs, offs, jc = run_decode_script(net, x, chunk, beam=beam)
        out = dict(joint_calls=np.int64(jc), chunk=np.int64(chunk), seed=np.int64(seed), beam=np.int64(beam),
                   frames=np.int64(x.shape[1]), n_chunks=np.int64(len(per)))
        for ci, hyps in enumerate(per):
            out[f"c{ci}_n"] = np.int64(len(hyps))
            for hi, (tk, lp) in enumerate(hyps):
                out[f"c{ci}_h{hi}_tokens"] = np.array(tk, np.int64)
                out[f"c{ci}_h{hi}_logp"] = np.float64(lp)
        best = max(per[-1], key=lambda h: h[1])
        print(f"{name}: {len(per)} chunks, joint calls {jc}, best {len(best[0])} tokens lp {best[1]:.4f}, {time.time() - t0:.1f}s")
        save(f"{name}.npz", **out)


@torch.no_grad()
def gen_streaming_beam(inputs):
    """OnlineRNNTModel.streaming_beam_search (model/online_rnnt_model.py:534-603): whole-utterance beam search with
    streaming_inference's chunking (static_chunk_size * 4 input frames, or chunk_size_ms); best hypothesis + every final one."""
    syn = torch.from_numpy(T.synth_fbank(2, 1000))
    for name, seed, x, scs, ms, beam in (("beam_si_ex6_scs16_s0", 0, inputs["ex6"], 16, None, 4),
                                         ("beam_si_syn1_scs16_s1_f400", 1, syn[1:2, :400], 16, None, 4),
                                         ("beam_si_ex6_scs32_ms200_s0", 0, inputs["ex6"], 32, 200, 3)):
        net = build(seed, scs)
        t0 = time.time()
        r, _, _ = net.streaming_beam_search(x, torch.tensor([x.shape[1]]), beam_size=beam, chunk_size_ms=ms)
        hyps = net.streaming_beam_hypotheses
        flat, cnt = pack_tokens([[int(t) for t in h.tokens] for h in hyps])
        print(f"{name}: best {len(r[0])} tokens, {len(hyps)} final hypotheses, {time.time() - t0:.1f}s")
        save(f"{name}.npz", tokens=np.array(r[0], np.int64), static_chunk_size=np.int64(scs), chunk_size_ms=np.int64(-1 if ms is None else ms),
             seed=np.int64(seed), beam=np.int64(beam), frames=np.int64(x.shape[1]),
             hyp_tokens=flat, hyp_counts=cnt, hyp_logp=np.array([float(h.log_prob) for h in hyps], np.float64))


@torch.no_grad()
def gen_full(inputs):
    net = build(0, 16)
    syn = torch.from_numpy(T.synth_fbank(2, 300, seed=99))
    lens = torch.tensor([300, 203])
    y, m = net.encoder(syn, lens, decoding_chunk_size=-1)
    save("full_seed0.npz", lens=lens.numpy().astype(np.int64), out=f32(y), mask=m.numpy(),
         fbank_seed=np.int64(99))
    x = inputs["ex6"]
    y, m = net.encoder(x, torch.tensor([x.shape[1]]), decoding_chunk_size=-1)
    save("full_ex6_seed0.npz", out=f32(y), mask=m.numpy())


def gen_cer():
    """Edit-distance cases for the CER harness: the reference's own calculate_cer (online_rnnt_eval.py:11-56) run on
    seeded random token lists.  The module cannot be imported (torchaudio/librosa at import time, py3.12-only
    f-strings), so only that one function is compiled from the parsed source and executed here."""
    import ast
    import re
    src = open("/root/reference/online_rnnt_eval.py", encoding="utf-8").read()
    src = re.sub(r"\{\n\s+", "{", src)
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "calculate_cer"]
    ns = {}
    exec(compile(ast.Module(body=fn, type_ignores=[]), "online_rnnt_eval.py", "exec"), ns)
    rng = np.random.Generator(np.random.Philox(99))
    n_cases, maxlen = 200, 24
    hyp = np.zeros((n_cases, maxlen), np.int32)
    refs = np.zeros((n_cases, maxlen), np.int32)
    lens = np.zeros((n_cases, 2), np.int32)
    res = np.zeros((n_cases, 5), np.float64)
    for i in range(n_cases):
        vocab = int(rng.integers(2, 9))                      # small alphabets: many equal-cost alignments
        a = rng.integers(0, vocab, int(rng.integers(0, maxlen + 1))).tolist()
        if i % 3 == 0:                                        # a noisy copy of the hypothesis
            b = [t for t in a if rng.random() > 0.2]
            b = [int(rng.integers(0, vocab)) if rng.random() < 0.2 else t for t in b]
            b = b[:maxlen]
        else:
            b = rng.integers(0, vocab, int(rng.integers(0, maxlen + 1))).tolist()
        hyp[i, :len(a)] = a
        refs[i, :len(b)] = b
        lens[i] = (len(a), len(b))
        res[i] = ns["calculate_cer"](a, b)
    save("cer_cases.npz", hyp=hyp, ref=refs, lens=lens, result=res)


@torch.no_grad()
def gen_offline():
    """Offline greedy search: the reference's basic_greedy_search (model/component/transducer.py:22-70) on the
    deterministic full-context encoder output (decoding_chunk_size=-1).  The function reads `model.blank`, which
    OnlineRNNTModel does not define (it has blank_id), so it is given a three-attribute view of the model."""
    import types
    from model.component.transducer import basic_greedy_search
    for seed in (0, 1):
        net = build(seed, 16)
        view = types.SimpleNamespace(predictor=net.predictor, joint=net.joint, blank=net.blank_id)
        x = torch.from_numpy(T.synth_fbank(3, 240, seed=77 + seed))
        lens = torch.tensor([240, 171, 96])
        y, m = net.encoder(x, lens, decoding_chunk_size=-1)
        out_lens = m.squeeze(1).sum(1)
        out = {}
        for n_steps in (64, 3):
            flat, cnt = pack_tokens(basic_greedy_search(view, y, out_lens, n_steps=n_steps))
            out[f"tokens_n{n_steps}"] = flat
            out[f"counts_n{n_steps}"] = cnt
        save(f"offline_greedy_seed{seed}.npz", lens=lens.numpy(), out_lens=out_lens.numpy(), **out)


@torch.no_grad()
def gen_ctc():
    """CTC head: OnlineCTC.log_softmax / argmax (model/online_rnnt_model.py:34-38) and the reference's own ctc_greedy_search
    (:647-671) on the DETERMINISTIC full-context encoder.  The method itself calls self.encoder(audios, audio_lens), which draws a
    random dynamic-chunk mask even in eval; here the encoder's forward is given decoding_chunk_size=-1, everything
    else (log_softmax, argmax, collapse loop) is the reference's code."""
    for seed in (0, 1):
        net = build(seed, 16)
        x = torch.from_numpy(T.synth_fbank(2, 300, seed=99 + seed))
        lens = torch.tensor([300, 203])
        enc_fwd = net.encoder.forward
        net.encoder.forward = lambda xs, xl, *a, **k: enc_fwd(xs, xl, decoding_chunk_size=-1)
        hyps = net.ctc_greedy_search(x, lens)
        y, m = net.encoder(x, lens)
        lp = net.ctc_head.log_softmax(y)
        ids = net.ctc_head.argmax(y)
        top2 = lp.topk(2, dim=2).values
        flat, cnt = pack_tokens(hyps)
        save(f"ctc_seed{seed}.npz", lens=lens.numpy(), fbank_seed=np.int64(99 + seed), ids=ids.numpy().astype(np.int64), mask=m.numpy(),
             hyp_tokens=flat, hyp_counts=cnt, logp_first8=f32(lp[:, :8]), logp_rowmax=f32(lp.max(dim=2).values),
             min_margin=np.float32((top2[..., 0] - top2[..., 1]).min().item()))


@torch.no_grad()
def gen_prefix():
    """WeNet prefix beam search (wenet/transducer/search/prefix_beam_search.py:42-148) run as the reference class on the model's
    own encoder / predictor / joint / CTC head, full context (decoding_chunk_size=-1), B = 1."""
    import math
    import wenet.transducer.search.prefix_beam_search as pbs_mod
    from wenet.transducer.search.prefix_beam_search import PrefixBeamSearch

    # The reference's call site is `log_add([a, b])` (prefix_beam_search.py:136-138) but the vendored wenet/utils/common.py:302-310
    # declares `log_add(*args)`: max() then returns the list and `a - a_max` raises TypeError whenever two prefixes merge, i.e. on
    # practically every utterance.  The generator binds the NAME log_add inside that module to the list form the call was written
    # for (upstream WeNet's signature, same arithmetic); no other line of the reference is touched.
    def log_add_list(args):
        if all(a == -float("inf") for a in args):
            return -float("inf")
        a_max = max(args)
        return a_max + math.log(sum(math.exp(a - a_max) for a in args))
    pbs_mod.log_add = log_add_list
    # (seed, frames of the padded input, valid frames, beam, file tag): the third case is a PADDED utterance (audio_lens < T): the
    # reference iterates every encoder frame, the padded ones included (prefix_beam_search.py:64,76)
    for seed, frames, valid, beam, tag in ((0, 240, 240, 4, "seed0"), (1, 171, 171, 5, "seed1"), (0, 240, 187, 4, "seed0_padded")):
        net = build(seed, 16)
        x = torch.from_numpy(T.synth_fbank(1, frames, seed=55 + seed))
        pbs = PrefixBeamSearch(net.encoder, net.predictor, net.joint, net.ctc_head, net.blank_id)
        beam_out, enc = pbs.prefix_beam_search(x, torch.tensor([valid]), decoding_chunk_size=-1, beam_size=beam, ctc_weight=0.3, transducer_weight=0.7)
        flat, cnt = pack_tokens([[int(t) for t in s.hyp] for s in beam_out])
        save(f"prefix_beam_{tag}.npz", frames=np.int64(frames), valid_frames=np.int64(valid), fbank_seed=np.int64(55 + seed), beam=np.int64(beam), hyp_tokens=flat, hyp_counts=cnt,
             scores=np.array([float(s.score) for s in beam_out], np.float64), enc_frames=np.int64(enc.size(1)),
             h=f32(torch.cat([s.cache[0] for s in beam_out], 1)), c=f32(torch.cat([s.cache[1] for s in beam_out], 1)))


if __name__ == "__main__":
    print("torch", torch.__version__, "threads", torch.get_num_threads())
    if len(sys.argv) > 1 and sys.argv[1] == "cer":
        gen_cer()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "offline":
        gen_offline()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "beam_si":
        ex = {k: torch.from_numpy(v)[None] for k, v in np.load(os.path.join(HERE, "inputs_example1.npz")).items()}
        gen_streaming_beam(ex)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] in ("ctc", "prefix"):
        (gen_ctc if sys.argv[1] == "ctc" else gen_prefix)()
        sys.exit(0)
    inp = gen_inputs()
    gen_modules(0, inp)
    gen_modules(1, inp)
    gen_streams(inp)
    gen_streaming_beam(inp)
    gen_full(inp)
    gen_cer()
    gen_offline()
    gen_ctc()
    gen_prefix()
