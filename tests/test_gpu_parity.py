"""Parity of the HIP path (through the C ABI) against the golden vectors produced by the reference and
against the CPU oracle on fresh seeded inputs.  Needs a real MI355X: `pytest -m gpu`.

Bars (BASELINE.json north_star): greedy token sequences bit-exact; encoder outputs / joint logits within
1e-3 (float32)."""
import numpy as np
import pytest
import torch

import ctc_vr_amd.testing as T
from conftest import load_golden

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3
# parity-gated numerics modes (include/rnnt_hip.h RNNT_NUMERICS_*): every golden test below runs in each of them with the
# same bars -- greedy tokens bit-exact, encoder outputs / logits within 1e-3.  Plain bf16 is a perf mode: test_bf16_perf_mode.
PARITY_MODES = ["fp32", "bf16x3", "f16x3"]


@pytest.fixture(params=PARITY_MODES)
def numerics(request, monkeypatch):
    monkeypatch.setenv("RNNT_NUMERICS", request.param)   # contexts created in the test pick it up (lib.numerics_id)
    return request.param


def maxdiff(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0


def ex_inputs():
    g = load_golden("inputs_example1.npz")
    return {k: torch.from_numpy(g[k])[None] for k in g.files}


def stream_input(name):
    src = name.split("_")[0]
    if src.startswith("syn"):
        return torch.from_numpy(T.synth_fbank(2, 1000))[int(src[3:]):int(src[3:]) + 1]
    return ex_inputs()[src]


@pytest.fixture(scope="module")
def models(np_state_dict):
    from ctc_vr_amd.online_rnnt_model import OnlineRNNTModel
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    cache = {}

    def get(seed, chunk):
        import os
        key = (seed, os.environ.get("RNNT_NUMERICS", "fp32"))
        return get2(key, seed, chunk)

    def get2(seed, real_seed, chunk):
        if seed not in cache:
            m = OnlineRNNTModel(input_dim=80, hidden_dim=256, vocab_size=T.VOCAB, blank_id=T.BLANK, streaming=True,
                                static_chunk_size=chunk, predictor_dropout=0)
            m.load_state_dict(np_state_dict(real_seed))
            cache[seed] = m
        cache[seed].encoder.static_chunk_size = chunk
        return cache[seed]
    return get


@pytest.mark.parametrize("seed", [0, 1])
def test_predictor_and_joint_step_api(seed, models, numerics):
    m = models(seed, 16)
    eng = m._engine
    g = load_golden(f"modules_seed{seed}.npz")
    dev = m.device
    h = torch.zeros(1, 256, device=dev)
    c = torch.zeros(1, 256, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    for i, tok in enumerate(g["pred_tokens"].tolist()):
        t = torch.tensor([tok], dtype=torch.int32, device=dev)
        out, h2, c2 = torch.empty(1, 256, device=dev), torch.empty(1, 256, device=dev), torch.empty(1, 256, device=dev)
        eng.predictor_step(t.data_ptr(), h.data_ptr(), c.data_ptr(), 1, out.data_ptr(), h2.data_ptr(), c2.data_ptr(), s)
        torch.cuda.synchronize()
        assert maxdiff(out[0], g["pred_out"][i]) < 1e-4
        assert maxdiff(h2[0], g["pred_h"][i]) < 1e-4
        assert maxdiff(c2[0], g["pred_c"][i]) < 1e-4
        h, c = h2, c2
    enc = torch.from_numpy(g["joint_enc"]).to(dev)
    prd = torch.from_numpy(g["joint_pred"]).to(dev)
    for mode, key in ((0, "joint_logits"), (1, "joint_logp")):
        out = torch.empty(2, 7, 5, T.VOCAB, device=dev)
        eng.joint(enc.data_ptr(), prd.data_ptr(), 2, 7, 5, mode, out.data_ptr(), s)
        torch.cuda.synchronize()
        assert maxdiff(out, g[key]) < LOGIT_TOL
        assert np.array_equal(out.argmax(-1).cpu().numpy(), g[key].argmax(-1))


@pytest.mark.parametrize("shape", [(1, 1, 1), (3, 37, 11), (5, 64, 28), (2, 129, 3)])
def test_joint_lattice_shapes(shape, models, numerics, np_state_dict):
    """rnnt_joint in lattice form (joint.py:48-69, online_rnnt_model.py:243,446-447) on row counts that are not multiples of the
    kernel's 64-row tile, through the persistent row-tile queue: logits and log-softmax within LOGIT_TOL of the oracle, same
    argmax, and two calls bit-identical (the queue hands tiles to workgroups in a different order every launch)."""
    from oracle import rnnt_oracle as O
    B, Tn, U = shape
    m = models(0, 16)
    eng, dev = m._engine, m.device
    sd = O.to_torch_sd(np_state_dict(0))
    g = torch.Generator().manual_seed(100 * B + Tn)
    enc = torch.randn(B, Tn, 256, generator=g)
    prd = torch.randn(B, U, 256, generator=g) * 0.5
    want = O.joint(sd, enc, prd)                                  # [B, T, U, V] logits
    want_lp = torch.log_softmax(want, dim=-1)
    s = torch.cuda.current_stream().cuda_stream
    enc_d, prd_d = enc.to(dev), prd.to(dev)
    for mode, ref in ((0, want), (1, want_lp)):
        out = torch.full((B, Tn, U, T.VOCAB), float("nan"), device=dev)
        eng.joint(enc_d.data_ptr(), prd_d.data_ptr(), B, Tn, U, mode, out.data_ptr(), s)
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        assert maxdiff(out, ref.numpy()) < LOGIT_TOL
        assert np.array_equal(out.argmax(-1).cpu().numpy(), ref.argmax(-1).numpy())
        again = torch.empty_like(out)
        eng.joint(enc_d.data_ptr(), prd_d.data_ptr(), B, Tn, U, mode, again.data_ptr(), s)
        torch.cuda.synchronize()
        assert torch.equal(out, again)


@pytest.mark.parametrize("seed", [0, 1])
def test_forward_chunk_traces(seed, models, numerics):
    """Three consecutive 16-frame chunks: encoder output and both caches in the reference's layouts,
    including the dropped first-chunk K/V and the offset drift (§0.5)."""
    m = models(seed, 16)
    g = load_golden(f"modules_seed{seed}.npz")
    x = ex_inputs()["ex0"]
    m.reset_streaming_cache()
    eng = m._engine
    s = torch.cuda.current_stream().cuda_stream
    off = 0
    for ci in range(3):
        chunk = x[:, ci * 16:(ci + 1) * 16].to(m.device).contiguous()
        tq = eng.encoder_chunk(chunk.data_ptr(), 16, off, off, s)
        assert tq == 3
        enc = eng.enc_frames(s)
        assert maxdiff(enc[0, -3:], g[f"fc{ci}_out"][0]) < LOGIT_TOL
        att = eng.att_cache(0, s)
        assert att.shape == g[f"fc{ci}_att_cache"].shape
        assert maxdiff(att, g[f"fc{ci}_att_cache"]) < LOGIT_TOL
        assert maxdiff(eng.cnn_cache(0, s), g[f"fc{ci}_cnn_cache"]) < LOGIT_TOL
        off += 4
    eng.greedy_decode(s)
    eng.frames_consume(s)


STREAMS = ["syn0_c16_s0", "syn1_c16_s0", "syn0_c16_s1", "ex0_c32_s0", "ex0_c32_s1", "ex6_c16_s0", "ex12_c64_s0", "ex12_c16_s1"]


@pytest.mark.parametrize("name", STREAMS)
def test_decode_script_greedy_matches_reference(name, models, numerics):
    """online_rnnt_decode.py greedy loop through the facade's process_single_chunk: tokens bit-exact per
    chunk, final caches / predictor state / offsets as the reference left them."""
    g = load_golden(f"stream_{name}.npz")
    chunk = int(g["chunk"])
    m = models(int(g["seed"]), chunk)
    x = stream_input(name)
    m.reset_streaming_cache()
    toks, counts = [], []
    for (a, b) in T.chunk_plan(x.shape[1], chunk):
        r, _, _ = m.process_single_chunk(x[:, a:b], torch.tensor([b - a]))
        counts.append(len(r))
        toks.extend(r)
    assert toks == g["tokens"].tolist()
    assert counts == g["counts"].tolist()
    assert m._global_encoder_offset == int(g["global_offset"])
    assert m.streaming_last_emitted_token == int(g["last_token"])
    att = m.streaming_att_cache.cpu().numpy()
    assert att.shape == tuple(g["att_cache_shape"])
    assert maxdiff(att[0, :, -3:, :], g["att_cache_l0_last"]) < LOGIT_TOL
    assert maxdiff(att[11, :, :3, :], g["att_cache_l11_first"]) < LOGIT_TOL
    assert abs(float(att.astype(np.float64).sum()) - float(g["att_cache_sum"])) < 0.5
    assert maxdiff(m.streaming_cnn_cache, g["cnn_cache"]) < LOGIT_TOL
    assert maxdiff(m.streaming_predictor_states[0], g["pred_h"]) < 1e-4


@pytest.mark.parametrize("name", ["si_ex0_scs16_s0", "si_syn0_scs16_s0", "si_ex6_scs32_ms200_s0"])
def test_streaming_inference_matches_reference(name, models, numerics):
    g = load_golden(f"stream_{name}.npz")
    m = models(int(g["seed"]), int(g["static_chunk_size"]))
    x = stream_input(name[3:])
    ms = int(g["chunk_size_ms"])
    r, _, _ = m.streaming_inference(x, torch.tensor([x.shape[1]]), None if ms < 0 else ms)
    assert r[0] == g["tokens"].tolist()
    assert tuple(m.streaming_att_cache.shape) == tuple(g["att_cache_shape"])


@pytest.mark.parametrize("name", ["beam_si_ex6_scs16_s0", "beam_si_syn1_scs16_s1_f400", "beam_si_ex6_scs32_ms200_s0"])
def test_streaming_beam_search_matches_reference(name, models, numerics):
    """streaming_beam_search through the facade (online_rnnt_model.py:534-603): the returned best hypothesis and the final beam
    (token lists in order, Python-double scores) equal the reference's."""
    g = load_golden(f"{name}.npz")
    m = models(int(g["seed"]), int(g["static_chunk_size"]))
    x = stream_input(name[8:])[:, :int(g["frames"])]
    ms = int(g["chunk_size_ms"])
    r, _, _ = m.streaming_beam_search(x, torch.tensor([x.shape[1]]), beam_size=int(g["beam"]), chunk_size_ms=None if ms < 0 else ms)
    assert r[0] == g["tokens"].tolist()
    cnt, flat = g["hyp_counts"].tolist(), g["hyp_tokens"].tolist()
    want = [flat[sum(cnt[:k]):sum(cnt[:k + 1])] for k in range(len(cnt))]
    assert [h.tokens for h in m.streaming_beam_hypotheses] == want
    assert np.allclose([h.log_prob for h in m.streaming_beam_hypotheses], g["hyp_logp"], atol=2e-3)


@pytest.mark.parametrize("name", ["beam_ex6_c16_s0", "beam_syn0_c16_s1_f320", "beam_ex0_c32_s0"])
def test_beam_search_matches_reference(name, models, numerics):
    """process_single_chunk_beam_search through the facade: after every chunk the beam (token lists in order,
    Python-double scores) equals the reference's (online_rnnt_model.py:389-522)."""
    g = load_golden(f"{name}.npz")
    chunk, beam = int(g["chunk"]), int(g["beam"])
    m = models(int(g["seed"]), chunk)
    x = stream_input(name[5:])[:, :int(g["frames"])]
    m.reset_streaming_cache()
    for ci, (a, b) in enumerate(T.chunk_plan(x.shape[1], chunk)):
        hyps, _, _ = m.process_single_chunk_beam_search(x[:, a:b], torch.tensor([b - a]), beam_size=beam)
        assert len(hyps) == int(g[f"c{ci}_n"]), ci
        for hi, h in enumerate(hyps):
            assert h.tokens == g[f"c{ci}_h{hi}_tokens"].tolist(), (ci, hi)
            assert abs(h.log_prob - float(g[f"c{ci}_h{hi}_logp"])) < 2e-3, (ci, hi)
    assert hyps[0].predictor_states[0].shape == (1, 1, 256)


def test_batched_beam_equals_single_stream(np_state_dict):
    from ctc_vr_amd.online_rnnt_model import StreamingBatch
    g = load_golden("beam_ex6_c16_s0.npz")
    x = ex_inputs()["ex6"]
    xb = torch.cat([x, x, x], 0).cuda().contiguous()
    sb = StreamingBatch(np_state_dict(0), 3, max_chunk_frames=32, max_cache_frames=64, max_enc_frames=16, max_beam=4)
    beams = sb.beam_script(xb, 16, 4)
    last = int(g["n_chunks"]) - 1
    for b in range(3):
        assert [h.tokens for h in beams[b]] == [g[f"c{last}_h{i}_tokens"].tolist() for i in range(int(g[f"c{last}_n"]))]


def test_native_beam_paths(np_state_dict, monkeypatch):
    """The beam search with the bookkeeping in the library (rnnt_beam_advance: beam_chain kernel + C++ merge) against
    the Python host logic over the same device calls (hypotheses and double scores identical), the whole-utterance form
    (one encoder call + one rnnt_beam_advance) against the per-chunk loop, and the launched extension steps
    (RNNT_BEAM_CHAIN=0) against the chain kernel (same tokens; scores differ by float32 summation order)."""
    from ctc_vr_amd.online_rnnt_model import StreamingBatch
    syn = torch.from_numpy(T.synth_fbank(3, 320, seed=21)).cuda().contiguous()
    mk = lambda: StreamingBatch(np_state_dict(1), 3, max_chunk_frames=32, max_cache_frames=128, max_enc_frames=128, max_beam=4)
    sb = mk()
    native = sb.beam_script(syn, 16, 4)
    sb.python_beam = True
    python = sb.beam_script(syn, 16, 4)
    sb.python_beam = False
    piped = sb.beam_script(syn, 16, 4, pipelined=True)
    for b in range(3):
        assert [h.tokens for h in native[b]] == [h.tokens for h in python[b]], b
        assert [h.log_prob for h in native[b]] == [h.log_prob for h in python[b]], b
        assert [h.tokens for h in piped[b]] == [h.tokens for h in native[b]], b
        assert max(abs(x.log_prob - y.log_prob) for x, y in zip(piped[b], native[b])) < 1e-3, b
    monkeypatch.setenv("RNNT_BEAM_CHAIN", "0")
    launched = mk().beam_script(syn, 16, 4)
    for b in range(3):
        assert launched[b][0].tokens == native[b][0].tokens, b
        assert abs(launched[b][0].log_prob - native[b][0].log_prob) < 1e-3, b


def test_enc_out_full_trace(models, numerics):
    g = load_golden("stream_syn0_c16_s0.npz")
    from ctc_vr_amd.online_rnnt_model import StreamingBatch  # noqa: F401
    m = models(0, 16)
    x = stream_input("syn0_c16_s0").to(m.device)
    m.reset_streaming_cache()
    eng = m._engine
    s = torch.cuda.current_stream().cuda_stream
    off = 0
    for (a, b) in T.chunk_plan(1000, 16):
        c = x[:, a:b].contiguous()
        eng.encoder_chunk(c.data_ptr(), b - a, off, off, s)
        off += (b - a) // 4
    enc = eng.enc_frames(s)[0]
    assert enc.shape == g["enc_out"].shape == (188, 256)
    assert maxdiff(enc, g["enc_out"]) < LOGIT_TOL
    eng.greedy_decode(s)          # whole-utterance decode after the encoder: same tokens as chunk-by-chunk
    assert eng.tokens(s)[0] == g["tokens"].tolist()


def test_batched_streams_equal_single_stream_reference(np_state_dict, numerics):
    """B=6 lock-stepped streams (two distinct inputs interleaved): every stream equals the B=1 reference."""
    from ctc_vr_amd.online_rnnt_model import StreamingBatch
    g0, g1 = load_golden("stream_syn0_c16_s0.npz"), load_golden("stream_syn1_c16_s0.npz")
    syn = torch.from_numpy(T.synth_fbank(2, 1000))
    x = torch.stack([syn[i % 2] for i in range(6)]).cuda().contiguous()
    sb = StreamingBatch(np_state_dict(0), 6, max_chunk_frames=32, max_cache_frames=256, max_enc_frames=256)
    for per_chunk in (True, False):
        toks = sb.decode_script(x, 16, per_chunk_decode=per_chunk)
        for i in range(6):
            assert toks[i] == (g0, g1)[i % 2]["tokens"].tolist(), (per_chunk, i)


def test_wavefront_encoder_matches_chunk_by_chunk(np_state_dict, numerics):
    """rnnt_encoder_chunks (wavefront over chunk x layer, batched subsampling, LDS-tiled grouped GEMMs) vs
    chunk-by-chunk rnnt_encoder_chunk (split-K GEMMs): same arithmetic with a different K summation order, so
    encoder frames / caches agree to float32 rounding (1e-4 observed ~1e-6) and the tokens equal the reference."""
    from ctc_vr_amd.online_rnnt_model import StreamingBatch
    g0, g1 = load_golden("stream_syn0_c16_s0.npz"), load_golden("stream_syn1_c16_s0.npz")
    syn = torch.from_numpy(T.synth_fbank(2, 1000))
    x = torch.stack([syn[i % 2] for i in range(4)]).cuda().contiguous()
    sb = StreamingBatch(np_state_dict(0), 4, max_chunk_frames=32, max_cache_frames=256, max_enc_frames=256)
    s = torch.cuda.current_stream().cuda_stream
    sb.reset()
    for (a, b) in T.chunk_plan(1000, 16):
        sb.process_chunk(x[:, a:b].contiguous(), decode=False)
    enc_seq = sb.engine.enc_frames(s)
    att_seq = sb.engine.att_cache(1, s)
    cnn_seq = sb.engine.cnn_cache(1, s)
    sb.engine.greedy_decode(s)
    toks = sb.decode_script(x, 16, pipelined=True)
    for i in range(4):
        assert toks[i] == (g0, g1)[i % 2]["tokens"].tolist(), i
    sb.reset()
    plan = T.chunk_plan(1000, 16)
    offs = [4 * i for i in range(len(plan))]
    sb.engine.encoder_chunks(x.data_ptr(), 1000, [a for a, _ in plan], [b - a for a, b in plan], offs, offs, s)
    assert maxdiff(sb.engine.enc_frames(s), enc_seq) < 1e-4
    assert maxdiff(sb.engine.att_cache(1, s), att_seq) < 1e-4
    assert maxdiff(sb.engine.cnn_cache(1, s), cnn_seq) < 1e-4
    assert maxdiff(enc_seq[0], g0["enc_out"]) < LOGIT_TOL


def test_full_size_properties(np_state_dict, numerics):
    """BASELINE configs[1] at full size (64 streams x 1000 frames, chunk 16), size-independent properties:
    whole-utterance call == per-chunk API, run-to-run determinism, independence of a stream from its batch
    position and neighbours, and two of the streams against the CPU oracle (the bench checks eight)."""
    from oracle import rnnt_oracle as O
    from ctc_vr_amd.online_rnnt_model import StreamingBatch
    sd_np = np_state_dict(0)
    B = 64
    x_cpu = torch.from_numpy(T.synth_fbank(B, 1000, seed=1234))
    x = x_cpu.cuda().contiguous()
    sb = StreamingBatch(sd_np, B, max_chunk_frames=24, max_cache_frames=200, max_enc_frames=200, max_tokens=1900)
    pipelined = sb.decode_script(x, 16, pipelined=True)
    assert sb.decode_script(x, 16, pipelined=True) == pipelined                      # deterministic
    per_chunk = sb.decode_script(x, 16, per_chunk_decode=True)                       # per-chunk API
    if numerics == "fp32":
        assert per_chunk == pipelined                                                # exact-f32 mode: the two schedules share every product and its order
    else:
        # Split modes: the whole-utterance schedule runs attention on 16-bit split MFMAs, the per-chunk API on the exact-f32
        # kernels, so the encoder frames differ at the split error (~1e-5) and a decision whose top-2 logits are closer than
        # that may fall the other way.  Attributable, not arbitrary: tokens may differ ONLY on streams whose smallest top-2
        # margin (float64 replay on the whole-utterance call's frames, ctc_vr_amd.testing.greedy_margins) is under MARGIN_TOL,
        # and only on a few of the 64.
        MARGIN_TOL = 1e-3
        plan = T.chunk_plan(1000, 16)
        offs = [4 * i for i in range(len(plan))]
        sb.reset()
        sb.engine.encoder_chunks(x.data_ptr(), 1000, [a for a, _ in plan], [b - a for a, b in plan], offs, offs, torch.cuda.current_stream().cuda_stream)
        margins, replay_ok = T.greedy_margins(sd_np, sb.engine.enc_frames(torch.cuda.current_stream().cuda_stream), pipelined)
        assert (replay_ok | (margins < 1e-4)).all()                                   # the float64 replay follows the f32 decoder except across a near-tie
        differ = [b for b in range(B) if per_chunk[b] != pipelined[b]]
        assert len(differ) <= 3, differ
        for b in differ:
            assert margins[b] < MARGIN_TOL, (b, margins[b])
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(7))
    permuted = sb.decode_script(x[perm.cuda()].contiguous(), 16, pipelined=True)
    for i in range(B):
        assert permuted[i] == pipelined[int(perm[i])], i                              # no cross-stream leakage
    counts = [len(t) for t in pipelined]
    assert min(counts) > 0 and max(counts) < 1880
    sd = O.to_torch_sd(sd_np)
    for b in (int(np.argmax(counts)), int(np.argmin(counts))):                         # the busiest and the quietest stream
        want, _, _ = O.decode_script_greedy(sd, x_cpu[b:b + 1], 16)
        assert pipelined[b] == want, b


@pytest.mark.parametrize("env", [{"RNNT_PERSISTENT": "0"}, {"RNNT_COOP": "1"}, {"RNNT_ATTN_STREAM": "0"}, {"RNNT_LM": "0"},
                                 {"RNNT_LM": "0", "RNNT_FUSED": "0"}])
def test_alternative_decoder_paths(np_state_dict, env, monkeypatch):
    """The launched decode path (what a serialising profiler falls back to), the cooperative decoder, the LDS-tiled attention and
    the wavefront schedules (fused / unfused; what a whole-utterance call with a cache reset in its middle falls back to from
    the layer-major schedule) produce the reference's tokens too."""
    from ctc_vr_amd.online_rnnt_model import StreamingBatch
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    g0, g1 = load_golden("stream_syn0_c16_s0.npz"), load_golden("stream_syn1_c16_s0.npz")
    syn = torch.from_numpy(T.synth_fbank(2, 1000))
    x = torch.stack([syn[i % 2] for i in range(6)]).cuda().contiguous()
    sb = StreamingBatch(np_state_dict(0), 6, max_chunk_frames=32, max_cache_frames=256, max_enc_frames=256)
    for mode in ({"pipelined": True}, {"per_chunk_decode": True}):
        toks = sb.decode_script(x, 16, **mode)
        for i in range(6):
            assert toks[i] == (g0, g1)[i % 2]["tokens"].tolist(), (env, mode, i)


@pytest.mark.parametrize("env", [{}, {"RNNT_LM_PW2_HEAD": "0"}, {"RNNT_LM_PW2_HEAD": "1"}, {"RNNT_LM_FFN_MERGE": "1"}, {"RNNT_LM_SIDE": "0"},
                                 {"RNNT_LM_QKV_TAIL": "0"}, {"RNNT_LM_OUT_CHAIN": "0"}])
@pytest.mark.parametrize("mode", ["bf16x3", "f16x3"])
def test_layer_major_fusion_variants(np_state_dict, env, mode, monkeypatch):
    """The fusions of the layer-major schedule are re-orderings of the same arithmetic: with the depthwise conv / pointwise_conv2 head
    inside the FFN launch or as launches of their own, with one launch per layer boundary, with the tail chunk class's subsampling on
    the side stream or in line, the split modes return the same tokens and the same encoder frames, bit for bit."""
    from ctc_vr_amd.online_rnnt_model import StreamingBatch
    x = torch.from_numpy(T.synth_fbank(6, 1000, seed=77)).cuda().contiguous()
    s = torch.cuda.current_stream().cuda_stream

    def run():
        sb = StreamingBatch(np_state_dict(0), 6, max_chunk_frames=32, max_cache_frames=256, max_enc_frames=256, numerics=mode)
        toks = sb.decode_script(x, 16, pipelined=True)
        return toks, np.array(sb.engine.enc_frames(s), copy=True)

    base_t, base_e = run()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    got_t, got_e = run()                                           # a new context reads the knobs at rnnt_create
    assert got_t == base_t, env
    assert np.array_equal(got_e, base_e), env


def test_long_utterance_pipelined_equals_per_chunk(np_state_dict, numerics):
    """30 s utterances (562 cached keys per layer at the end, 186 chunks: many wavefront stages, streaming attention over
    several 64-key rounds, tail-merged last chunk): whole-utterance call == per-chunk API, for two chunk sizes."""
    from ctc_vr_amd.online_rnnt_model import StreamingBatch
    sb = StreamingBatch(np_state_dict(1), 4, max_chunk_frames=64, max_cache_frames=800, max_enc_frames=800, max_tokens=8000)
    x = torch.from_numpy(T.synth_fbank(4, 3000, seed=9)).cuda().contiguous()
    for chunk in (16, 32):
        a = sb.decode_script(x, chunk, pipelined=True)
        assert a == sb.decode_script(x, chunk, per_chunk_decode=True), chunk
        assert min(len(t) for t in a) > 0


def test_fresh_inputs_against_oracle(np_state_dict, numerics):
    """Seeded inputs no fixture covers: HIP (B=4, chunk 24) vs the CPU oracle run stream by stream."""
    from oracle import rnnt_oracle as O
    from ctc_vr_amd.online_rnnt_model import StreamingBatch
    sd_np = np_state_dict(1)
    sd = O.to_torch_sd(sd_np)
    x = torch.from_numpy(T.synth_fbank(4, 200, seed=4321))
    sb = StreamingBatch(sd_np, 4, max_chunk_frames=48, max_cache_frames=128, max_enc_frames=128)
    got = sb.decode_script(x.cuda().contiguous(), 24)
    for b in range(4):
        want, _, st = O.decode_script_greedy(sd, x[b:b + 1], 24)
        assert got[b] == want, b
    att = sb.engine.att_cache(3)
    assert maxdiff(att, st.att_cache.numpy()) < LOGIT_TOL


def test_encoder_full_context(models, numerics):
    m = models(0, 16)
    eng = m._engine
    g = load_golden("full_seed0.npz")
    x = torch.from_numpy(T.synth_fbank(2, 300, seed=int(g["fbank_seed"])))
    from ctc_vr_amd.lib import RnntEngine
    e2 = RnntEngine(max_streams=2, max_chunk_frames=320, max_cache_frames=128, max_enc_frames=8, vocab_size=T.VOCAB, blank_id=T.BLANK)
    e2.load_state_dict(T.make_state_dict(0))
    assert e2.numerics == {"fp32": 0, "bf16x3": 1, "bf16": 2, "f16x3": 3}[numerics]
    xd = x.cuda().contiguous()
    out = torch.empty(2, 74, 256, device="cuda")
    tq = e2.encoder_full(xd.data_ptr(), g["lens"], 2, 300, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert tq == 74
    valid = g["mask"][:, 0, :, None]
    assert maxdiff(out.cpu().numpy() * valid, g["out"] * valid) < LOGIT_TOL
    del eng


@pytest.mark.parametrize("seed", [0, 1])
def test_ctc_greedy_search_matches_reference(seed, np_state_dict, numerics):
    """CTC head on the full-context encoder against the reference's own OnlineCTC.argmax and ctc_greedy_search
    (model/online_rnnt_model.py:34-38,647-671; golden ctc_seed*.npz from gen_golden.py: per-frame argmax ids on the valid frames
    and the collapsed hypotheses, every numerics mode; the fixture's smallest top-2 log-prob margin is recorded next to them)."""
    from ctc_vr_amd.online_rnnt_model import OnlineRNNTModel
    g = load_golden(f"ctc_seed{seed}.npz")
    m = OnlineRNNTModel(input_dim=80, hidden_dim=256, vocab_size=T.VOCAB, blank_id=T.BLANK, max_streams=2, max_chunk_frames=320,
                        max_cache_frames=128, max_enc_frames=128, max_beam=0)
    m.load_state_dict(np_state_dict(seed))
    x = torch.from_numpy(T.synth_fbank(2, 300, seed=int(g["fbank_seed"])))
    lens = torch.from_numpy(g["lens"])
    got = m.ctc_greedy_search(x, lens)
    want, o = [], 0
    for c in g["hyp_counts"].tolist():
        want.append(g["hyp_tokens"][o:o + c].tolist())
        o += c
    assert got == want and len(got[0]) > 0
    xd = x.cuda().contiguous()                      # keep the tensor alive across the call
    ids = m._engine.ctc_argmax(xd.data_ptr(), lens.numpy().astype(np.int32), 2, 300, torch.cuda.current_stream().cuda_stream)
    valid = g["mask"][:, 0, :]
    assert np.array_equal(ids[valid], g["ids"][valid])
    assert float(g["min_margin"]) > 5e-4     # the fixture's decisions sit well outside the split-operand modes' ~6e-5 encoder error


def test_fbank_frontend_against_oracle(models):
    """rnnt_fbank (windowed DFT + mel projection as f32 MFMA GEMMs, dB epilogue) vs the NumPy float64 restatement of
    torchaudio's MelSpectrogram + AmplitudeToDB.  PARITY UNPINNED against the reference itself (no torchaudio here, no
    fixture in the reference); tolerance 2e-3 dB on bins above -60 dB (f32 DFT by GEMM vs f64 FFT), 0.05 dB below."""
    from oracle import fbank_oracle as F
    from ctc_vr_amd.features import extract_audio_features
    eng = models(0, 16)._engine
    rng = np.random.default_rng(5)
    for rate, n, B in ((16000, 16000, 3), (48000, 30001, 2), (16000, 700, 1)):
        t = np.arange(n) / rate
        w = 0.1 * rng.standard_normal((B, n)) + 0.5 * np.sin(2 * np.pi * 440.0 * t)[None, :]
        w[0, n // 2:] = 0.0                                                        # digital silence: the -100 dB floor
        got = extract_audio_features(eng, torch.from_numpy(w.astype(np.float32)), rate).cpu().numpy()
        assert got.shape == (B, 1 + n // 512, 80)
        for b in range(B):
            want = F.extract_audio_features(w[b].astype(np.float32), rate)
            hi = want > -60.0
            assert np.max(np.abs(got[b][hi] - want[hi])) < 2e-3, (rate, n, b)
            lo = ~hi
            if lo.any():
                assert np.max(np.abs(got[b][lo] - want[lo])) < 0.05 or np.all(got[b][lo] < -59.9), (rate, n, b)
    one = extract_audio_features(eng, torch.from_numpy(w[0].astype(np.float32)), rate).cpu().numpy()   # 1-D input form
    assert np.array_equal(one, got[0])
    assert models(0, 16).extract_audio_features(torch.from_numpy(w[0].astype(np.float32)), rate).shape == got[0].shape
    with pytest.raises(Exception):
        extract_audio_features(eng, torch.zeros(400), 16000)                       # <= n_fft/2 samples: reflect padding impossible


@pytest.mark.parametrize("seed", [0, 1])
def test_offline_greedy_search_matches_reference(seed, np_state_dict):
    """rnnt_greedy_search_full / OnlineRNNTModel.forward of a non-streaming model vs the reference's basic_greedy_search
    on its full-context encoder (golden, ragged batch of 3, n_steps 64 and 3): tokens exact."""
    from ctc_vr_amd.online_rnnt_model import OnlineRNNTModel
    g = load_golden(f"offline_greedy_seed{seed}.npz")
    m = OnlineRNNTModel(input_dim=80, hidden_dim=256, vocab_size=T.VOCAB, blank_id=T.BLANK, streaming=False, predictor_dropout=0,
                        max_streams=3, max_chunk_frames=256, max_enc_frames=64, max_tokens=4096)
    m.load_state_dict(np_state_dict(seed))
    x = torch.from_numpy(T.synth_fbank(3, 240, seed=77 + seed))
    lens = torch.from_numpy(g["lens"])
    for n_steps in (64, 3):
        hyps = m.greedy_search_full(x, lens, n_steps=n_steps)
        assert [len(h) for h in hyps] == g[f"counts_n{n_steps}"].tolist(), n_steps
        assert [t for h in hyps for t in h] == g[f"tokens_n{n_steps}"].tolist(), n_steps
    hyps, a, b = m(x, lens)                                   # forward(): n_steps = 64
    assert a is None and b is None and [len(h) for h in hyps] == g["counts_n64"].tolist()


def test_wav_to_tokens_cli_flow(tmp_path, np_state_dict):
    """online_rnnt_decode.py's flow end to end: PCM wav -> device features -> chunk loop (greedy + beam) from a
    checkpoint file; the greedy tokens equal the oracle run on the same features, the beam's best path is reported
    incrementally as in the reference (:148)."""
    import wave
    from oracle import rnnt_oracle as O
    from ctc_vr_amd.online_rnnt_decode import decode_single_audio, chunk_bounds
    rate, n = 16000, 16000 * 8                                   # 8 s -> 251 feature frames (hop 512)
    rng = np.random.default_rng(3)
    t = np.arange(n) / rate
    sig = 0.3 * np.sin(2 * np.pi * (300 + 200 * np.sin(2 * np.pi * 0.7 * t)) * t) + 0.05 * rng.standard_normal(n)
    pcm = np.clip(sig * 32768, -32768, 32767).astype("<i2")
    wav = tmp_path / "a.wav"
    with wave.open(str(wav), "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(rate); f.writeframes(pcm.tobytes())
    sd_np = np_state_dict(0)
    ckpt = tmp_path / "online_model.pt"
    torch.save({"model": {k: torch.from_numpy(v) for k, v in sd_np.items()}, "epoch": 4}, str(ckpt))
    res = decode_single_audio(str(wav), str(ckpt), vocab_size=T.VOCAB, blank_id=T.BLANK, static_chunk_size=32, beam_size=4, verbose=False,
                              predictor_dropout=0)
    assert chunk_bounds(251, 32)[-1] == (192, 251) and res["chunks"] == 7
    from ctc_vr_amd.online_rnnt_model import OnlineRNNTModel
    m = OnlineRNNTModel(input_dim=80, hidden_dim=256, vocab_size=T.VOCAB, blank_id=T.BLANK, streaming=True, static_chunk_size=32)
    feats = m.extract_audio_features(torch.from_numpy(pcm.astype(np.float32) / 32768.0), rate).cpu()
    want, _, _ = O.decode_script_greedy(O.to_torch_sd(sd_np), feats[None], 32)
    assert res["greedy_tokens"] == want
    best = max(res["beam_hypotheses"], key=lambda h: h.log_prob)
    assert res["beam_tokens"][:len(best.tokens)] == res["beam_tokens"] and len(res["beam_tokens"]) >= 1


def test_rtf_harness(models):
    """per-chunk RTF statistics with online_rnnt_delay.py's definition."""
    from ctc_vr_amd.online_rnnt_delay import evaluate_rtf
    m = models(0, 16)
    ex = ex_inputs()
    r = evaluate_rtf(m, [ex["ex6"][0]], 16, beam_size=4)
    assert r["greedy"]["chunks"] == 10 and r["beam"]["chunks"] == 10
    assert 0 < r["greedy"]["p50"] <= r["greedy"]["max"] and r["greedy"]["mean"] < 1.0   # faster than real time


def test_error_paths(models):
    from ctc_vr_amd.lib import RnntEngine, RnntError
    m = models(0, 16)
    m.reset_streaming_cache()
    r, _, _ = m.process_single_chunk(torch.zeros(1, 5, 80), torch.tensor([5]))   # <7 frames: warn + [] (:356-359)
    assert r == []
    with pytest.raises(AssertionError):
        m.process_single_chunk(torch.zeros(2, 16, 80), torch.tensor([16, 16]))     # B != 1 (:348-349)
    e = RnntEngine(max_streams=1, max_chunk_frames=16, max_cache_frames=8, max_enc_frames=8, vocab_size=T.VOCAB, blank_id=T.BLANK)
    with pytest.raises(RnntError):
        e.reset(1)                                                                # weights not finalized
    with pytest.raises(RnntError):
        e.load_state_dict({"encoder.after_norm.weight": np.ones(256, np.float32)})  # missing tensors


def test_bf16_perf_mode(np_state_dict):
    """RNNT_NUMERICS_BF16 (plain bf16 operands, fp32 accumulate) is a perf mode, not a parity mode: with top-2 logit margins
    down to 3e-3 on these seeded weights bf16 arithmetic flips greedy tokens.  The bars here are only that it
    runs the same path, stays close (encoder output within 0.1 of fp32 on values of magnitude ~3) and is deterministic; the
    bench reports its token-match rate and encoder error next to its throughput instead of claiming parity."""
    from ctc_vr_amd.online_rnnt_model import StreamingBatch
    x = torch.from_numpy(T.synth_fbank(4, 1000)).cuda().contiguous()
    plan = T.chunk_plan(1000, 16)
    offs = [4 * i for i in range(len(plan))]
    s = torch.cuda.current_stream().cuda_stream
    res = {}
    for mode in ("fp32", "bf16"):
        sb = StreamingBatch(np_state_dict(0), 4, max_chunk_frames=32, max_cache_frames=256, max_enc_frames=256, numerics=mode)
        sb.reset()
        sb.engine.encoder_chunks(x.data_ptr(), 1000, [a for a, _ in plan], [b - a for a, b in plan], offs, offs, s)
        enc = sb.engine.enc_frames(s).copy()
        toks = sb.decode_script(x, 16, pipelined=True)
        assert sb.decode_script(x, 16, pipelined=True) == toks
        res[mode] = (enc, toks)
    err = maxdiff(res["bf16"][0], res["fp32"][0])
    assert 1e-4 < err < 0.1, err          # really reduced precision, and still the same function
    n_tok = sum(len(t) for t in res["fp32"][1])
    same = sum(sum(int(a == b) for a, b in zip(p, q)) for p, q in zip(res["bf16"][1], res["fp32"][1]))
    assert same > 0.3 * n_tok             # position-wise agreement before the first flip shifts a stream


def test_large_vocabulary_paths():
    """Vocabulary > 512 (the reference's default constructor vocabulary is 4336): the beam search takes the launched extension
    steps (beam_reduce scans the row in strides: its first version kept 512 entries in registers and silently ignored the rest),
    greedy decode takes the one-CU-per-stream decoder (greedy_multi holds <= 128 vocabulary rows per part).  Checked against the
    CPU oracle on a seeded 600-token model whose best tokens are spread over the whole range."""
    from oracle import rnnt_oracle as O
    from ctc_vr_amd.online_rnnt_model import OnlineRNNTModel
    V, blank = 600, 5
    sd_np = T.make_state_dict(5, vocab=V, blank=blank, blank_bias=11.0, out_gain=6.0)   # 56 tokens on this input, 10 of them >= 512
    sd = O.to_torch_sd(sd_np)
    x = torch.from_numpy(T.synth_fbank(1, 192, seed=31))
    m = OnlineRNNTModel(input_dim=80, hidden_dim=256, vocab_size=V, blank_id=blank, streaming=True, static_chunk_size=16, predictor_dropout=0)
    m.load_state_dict(sd_np)
    want, _, _ = O.decode_script_greedy(sd, x, 16, blank=blank)
    m.reset_streaming_cache()
    got = []
    for (a, b) in T.chunk_plan(192, 16):
        got.extend(m.process_single_chunk(x[:, a:b], torch.tensor([b - a]))[0])
    assert got == want and max(want) >= 512, (max(want), len(want))
    st = O.OracleStream(sd, blank, 16)
    m.reset_streaming_cache()
    big = 0
    for (a, b) in T.chunk_plan(192, 16):
        ob = st.process_single_chunk_beam_search(x[:, a:b], beam_size=4)
        hb, _, _ = m.process_single_chunk_beam_search(x[:, a:b], torch.tensor([b - a]), beam_size=4)
        assert [h.tokens for h in hb] == [h.tokens for h in ob]
        assert max(abs(p.log_prob - q.log_prob) for p, q in zip(hb, ob)) < 2e-3
        big += sum(t >= 512 for h in hb for t in h.tokens)
    assert big > 0        # hypotheses along the way carry tokens >= 512 (9 of the 12 chunks' beams on this input)


def test_ragged_batch_equals_single_stream_references(np_state_dict, numerics):
    """Eight utterances of different lengths in one padded batch (utils/utils.py:29-50; online_rnnt_eval.py:86-94 decodes each with
    its own audio_lens): every stream's tokens equal its own B = 1 golden / oracle result -- three example1.pt utterances with
    reference goldens (521, 160, 648 frames), a duplicate length (two streams share one class) and synthetic lengths down to one
    below the 7-frame minimum."""
    from oracle import rnnt_oracle as O
    from ctc_vr_amd.online_rnnt_model import StreamingBatch
    ex = ex_inputs()
    syn = torch.from_numpy(T.synth_fbank(5, 1000))
    utts = [ex["ex0"][0], ex["ex6"][0], ex["ex12"][0], syn[0, :1000], syn[1, :333], syn[2, :160], syn[3, :47], syn[4, :6]]
    lens = [u.shape[0] for u in utts]
    x = torch.zeros(len(utts), max(lens), 80)
    for b, u in enumerate(utts):
        x[b, :lens[b]] = u
    sb = StreamingBatch(np_state_dict(0), len(utts), max_chunk_frames=48, max_cache_frames=256, max_enc_frames=256)
    got = sb.decode_script_ragged(x.cuda().contiguous(), torch.tensor(lens), 16)
    assert got[1] == load_golden("stream_ex6_c16_s0.npz")["tokens"].tolist()        # reference golden (160 frames, chunk 16, seed 0)
    assert got[3] == load_golden("stream_syn0_c16_s0.npz")["tokens"].tolist()       # reference golden (1000 frames)
    sd = O.to_torch_sd(np_state_dict(0))
    for b, u in enumerate(utts):
        want, _, _ = O.decode_script_greedy(sd, u[None], 16) if lens[b] >= 7 else ([], None, None)
        assert got[b] == want, (b, lens[b])
    assert got[7] == [] and len(got[0]) > 0 and len(got[6]) >= 0
    assert sb.decode_script_ragged(x.cuda().contiguous(), torch.tensor(lens), 16, pipelined=False) == got


def test_ragged_batch_64_lengths_one_call(np_state_dict, numerics):
    """64 utterances of 64 DISTINCT lengths (40 .. 1000 frames) in one padded batch: decode_script_ragged makes exactly ONE library
    call (rnnt_decode_ragged: per-stream chunk plans, tail chunks and key windows inside the layer-major launches) and no
    per-class call, and every stream's tokens equal its own B = 1 whole-utterance run; three streams are also checked against
    the CPU oracle run on the unpadded utterance (utils/utils.py:29-50, online_rnnt_eval.py:86-94)."""
    from oracle import rnnt_oracle as O
    from ctc_vr_amd.online_rnnt_model import StreamingBatch
    n = 64
    rng = np.random.default_rng(5)
    lens = sorted(rng.choice(np.arange(40, 1001), size=n, replace=False).tolist(), reverse=True)
    lens[0], lens[-1] = 1000, 40
    perm = rng.permutation(n)
    lens = [lens[i] for i in perm]                                   # long and short utterances interleaved over the batch positions
    full = torch.from_numpy(T.synth_fbank(n, 1000, seed=4321))
    x = torch.zeros(n, 1000, 80)
    for b in range(n):
        x[b, :lens[b]] = full[b, :lens[b]]
    xd = x.cuda().contiguous()
    sb = StreamingBatch(np_state_dict(0), n, max_chunk_frames=48, max_cache_frames=256, max_enc_frames=256)
    calls = {"ragged": 0, "uniform": 0}
    orig_r, orig_u = sb.engine.decode_ragged, sb.engine.encoder_chunks
    sb.engine.decode_ragged = lambda *a, **k: (calls.__setitem__("ragged", calls["ragged"] + 1), orig_r(*a, **k))[1]
    sb.engine.encoder_chunks = lambda *a, **k: (calls.__setitem__("uniform", calls["uniform"] + 1), orig_u(*a, **k))[1]
    got = sb.decode_script_ragged(xd, torch.tensor(lens), 16)
    assert calls == {"ragged": 1, "uniform": 0}
    assert sb.decode_script_ragged(xd, torch.tensor(lens), 16) == got          # reproducible
    one = StreamingBatch(np_state_dict(0), 1, max_chunk_frames=48, max_cache_frames=256, max_enc_frames=256)
    for b in range(n):
        want = one.decode_script(xd[b:b + 1, :lens[b]].contiguous(), 16, pipelined=True)[0]
        assert got[b] == want, (b, lens[b], len(got[b]), len(want))
    sd = O.to_torch_sd(np_state_dict(0))
    for b in (int(np.argmax(lens)), int(np.argmin(lens)), 7):
        want, _, _ = O.decode_script_greedy(sd, x[b:b + 1, :lens[b]], 16)
        assert got[b] == want, (b, lens[b])


@pytest.mark.parametrize("seed", ["0", "1", "0_padded"])
def test_prefix_beam_search_matches_reference(seed, np_state_dict, numerics):
    """WeNet prefix beam search (CTC-fused, one symbol per frame, log_add prefix merge) on the full-context encoder
    against the reference class itself (golden prefix_beam_seed*.npz): hypotheses exact, scores within 2e-3 (double sums of fp32
    log-probs), final LSTM h of every hypothesis within 1e-3."""
    from ctc_vr_amd.online_rnnt_model import OnlineRNNTModel
    g = load_golden(f"prefix_beam_seed{seed}.npz")
    frames = int(g["frames"])
    m = OnlineRNNTModel(input_dim=80, hidden_dim=256, vocab_size=T.VOCAB, blank_id=T.BLANK, max_streams=8, max_chunk_frames=256,
                        max_cache_frames=128, max_enc_frames=128, max_beam=0)
    m.load_state_dict(np_state_dict(int(seed[0])))
    x = torch.from_numpy(T.synth_fbank(1, frames, seed=int(g["fbank_seed"])))
    beam = m.prefix_beam_search(x, torch.tensor([int(g["valid_frames"])]), beam_size=int(g["beam"]))   # "_padded": audio_lens < T
    want, o = [], 0
    for c in g["hyp_counts"].tolist():
        want.append(g["hyp_tokens"][o:o + c].tolist())
        o += c
    assert [b[0] for b in beam] == want
    assert max(abs(b[1] - s) for b, s in zip(beam, g["scores"])) < 2e-3
    assert maxdiff(m._prefix_states[0].cpu().numpy()[None], g["h"]) < LOGIT_TOL


def test_full_size_beam_properties(np_state_dict):
    """BASELINE configs[2] at full size (64 streams x 1000 frames, chunk 16, beam 4), size-independent properties: the
    whole-utterance form (one encoder call + one rnnt_beam_advance) equals the per-chunk beam API (hypotheses and scores), a
    stream's beam does not depend on its batch position, and two streams equal the CPU oracle's beam search run chunk by chunk
    (online_rnnt_model.py:389-522 restated in oracle/rnnt_oracle.py, pinned by the reference's beam goldens)."""
    from oracle import rnnt_oracle as O
    from ctc_vr_amd.online_rnnt_model import StreamingBatch
    sd_np = np_state_dict(0)
    B = 64
    x_cpu = torch.from_numpy(T.synth_fbank(B, 1000, seed=1234))
    x = x_cpu.cuda().contiguous()
    sb = StreamingBatch(sd_np, B, max_chunk_frames=24, max_cache_frames=200, max_enc_frames=200, max_tokens=64, max_beam=4)
    sig = lambda beams: [[(tuple(h.tokens), round(h.log_prob, 3)) for h in bm] for bm in beams]
    whole = sb.beam_script(x, 16, 4, pipelined=True)
    assert sig(sb.beam_script(x, 16, 4, pipelined=True)) == sig(whole)                    # deterministic
    per_chunk = sb.beam_script(x, 16, 4, pipelined=False)
    for b in range(B):
        assert [h.tokens for h in per_chunk[b]] == [h.tokens for h in whole[b]], b
        assert max(abs(p.log_prob - q.log_prob) for p, q in zip(per_chunk[b], whole[b])) < 2e-3, b
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(11))
    permuted = sb.beam_script(x[perm.cuda()].contiguous(), 16, 4, pipelined=True)
    for i in range(B):
        assert sig([permuted[i]]) == sig([whole[int(perm[i])]]), i                           # no cross-stream leakage
    sd = O.to_torch_sd(sd_np)
    for b in (0, 37):
        st = O.OracleStream(sd, T.BLANK, 16)
        for (a, e) in T.chunk_plan(1000, 16):
            ob = st.process_single_chunk_beam_search(x_cpu[b:b + 1, a:e], beam_size=4)
        assert [h.tokens for h in whole[b]] == [h.tokens for h in ob], b
        assert max(abs(p.log_prob - q.log_prob) for p, q in zip(whole[b], ob)) < 2e-3, b
    assert len(whole[0][0].tokens) > 0


def test_full_context_long_utterance(np_state_dict, numerics):
    """BASELINE configs[4] shape: a 3000-frame utterance (749 encoder frames: 12 key tiles of the full-context attention; the
    reference golden covers 74 frames) against the CPU oracle's full-context encoder (encoder.py:121-180 with
    decoding_chunk_size = -1), and position invariance inside a batch of 32 x 30 s: the same utterance in slots 0, 5 and 31
    between different neighbours gives the same frames."""
    from oracle import rnnt_oracle as O
    from ctc_vr_amd.lib import RnntEngine
    sd_np = np_state_dict(0)
    Bf, Tn, tq = 32, 3000, 749
    eng = RnntEngine(max_streams=Bf, max_chunk_frames=Tn, max_cache_frames=760, max_enc_frames=8, vocab_size=T.VOCAB, blank_id=T.BLANK)
    eng.load_state_dict(sd_np, numerics=numerics)
    xs = torch.from_numpy(T.synth_fbank(4, Tn, seed=77))
    xb = torch.stack([xs[0] if b in (0, 5, 31) else xs[1 + b % 3] for b in range(Bf)]).cuda().contiguous()
    out = torch.empty(Bf, tq, 256, device="cuda")
    lens = np.full(Bf, Tn, np.int32)
    s = torch.cuda.current_stream().cuda_stream
    assert eng.encoder_full(xb.data_ptr(), lens, Bf, Tn, out.data_ptr(), s) == tq
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert maxdiff(o[5], o[0]) < 1e-5 and maxdiff(o[31], o[0]) < 1e-5
    want, _ = O.encoder_full(O.to_torch_sd(sd_np), xs[0:1], torch.tensor([Tn]))
    assert tuple(want.shape) == (1, tq, 256)
    assert maxdiff(o[0], want[0].numpy()) < LOGIT_TOL
    one = torch.empty(1, tq, 256, device="cuda")
    x1 = xs[0:1].cuda().contiguous()
    assert eng.encoder_full(x1.data_ptr(), lens[:1], 1, Tn, one.data_ptr(), s) == tq
    torch.cuda.synchronize()
    assert maxdiff(one.cpu().numpy()[0], o[0]) < 1e-4      # B = 1 call (M = 749 rows: other GEMM kernels, other summation order) ~ its slot in the batch of 32


def test_weight_reload_invalidates_cached_launch_state(np_state_dict, monkeypatch):
    """A second load_state_dict on a live context (same architecture, other weights) must not replay anything that holds
    addresses or values of the first blob: the launched decode path (RNNT_PERSISTENT=0) replays hipGraphs whose kernel
    arguments point into the weight blob, the whole-utterance schedules cache descriptor tables.  Tokens after the reload
    equal a fresh context's, for both APIs."""
    from ctc_vr_amd.online_rnnt_model import StreamingBatch
    monkeypatch.setenv("RNNT_PERSISTENT", "0")
    x = torch.from_numpy(T.synth_fbank(3, 400, seed=5)).cuda().contiguous()
    mk = lambda seed: StreamingBatch(np_state_dict(seed), 3, max_chunk_frames=32, max_cache_frames=128, max_enc_frames=128)
    fresh = mk(1)
    want_pc = fresh.decode_script(x, 16, per_chunk_decode=True)
    want_wu = fresh.decode_script(x, 16, pipelined=True)
    assert want_pc == want_wu
    sb = mk(0)
    first = sb.decode_script(x, 16, per_chunk_decode=True)
    assert sb.decode_script(x, 16, pipelined=True) == first and first != want_pc
    sb.engine.load_state_dict(np_state_dict(1))
    assert sb.decode_script(x, 16, per_chunk_decode=True) == want_pc
    assert sb.decode_script(x, 16, pipelined=True) == want_wu


def test_large_stream_count_whole_utterance(np_state_dict):
    """More streams than the four-CUs-per-stream decoder can hold (n_streams x 4 > CUs): the whole-utterance call takes the
    one-workgroup-per-stream resident decoder after the encoder, with more decoder workgroups than can be resident at once
    (they do not wait on each other).  320 streams x 160 frames built from two golden utterances: every stream's tokens equal
    the reference's for its utterance."""
    from ctc_vr_amd.online_rnnt_model import StreamingBatch
    g0, g1 = load_golden("stream_syn0_c16_s0.npz"), load_golden("stream_syn1_c16_s0.npz")
    syn = torch.from_numpy(T.synth_fbank(2, 1000))
    n = 320
    x = torch.stack([syn[i % 2, :1000] for i in range(n)]).cuda().contiguous()
    sb = StreamingBatch(np_state_dict(0), n, max_chunk_frames=32, max_cache_frames=256, max_enc_frames=256, max_tokens=2048)
    toks = sb.decode_script(x, 16, pipelined=True)
    for i in range(n):
        assert toks[i] == (g0, g1)[i % 2]["tokens"].tolist(), i


def test_packed_blob_ingest(np_state_dict):
    """rnnt_load_packed (the multi-GPU weight path: dist.broadcast_packed leaves the packed blob on the device, the context takes it
    in one call): same tokens as the per-tensor ingest, from a device blob and from a host blob; a wrong table is an error."""
    import ctc_vr_amd.dist as Dm
    from ctc_vr_amd.online_rnnt_model import StreamingBatch
    sd = np_state_dict(0)
    x = torch.from_numpy(T.synth_fbank(2, 160, seed=77)).cuda().contiguous()
    kw = dict(max_chunk_frames=32, max_cache_frames=64, max_enc_frames=64)
    want = StreamingBatch(sd, 2, **kw).decode_script(x, 16)
    flat = Dm.pack_state_dict(sd)
    for blob in (torch.from_numpy(flat).cuda(), flat):
        assert StreamingBatch(None, 2, packed=(blob, T.VOCAB), **kw).decode_script(x, 16) == want
    with pytest.raises(Exception):
        StreamingBatch(None, 2, packed=(torch.from_numpy(flat[:-8]).cuda(), T.VOCAB), **kw)


@pytest.mark.parametrize("n", [16, 64])
def test_two_contexts_in_flight(n, np_state_dict):
    """Two contexts driven by two host threads on two HIP streams at the same time (how bench.py keeps two batches in flight:
    one batch's decode beside the other's encoder; include/rnnt_hip.h: distinct contexts are independent, one host thread
    per context at a time): every pass of either context returns the reference's tokens."""
    import threading
    from ctc_vr_amd.online_rnnt_model import StreamingBatch
    g0, g1 = load_golden("stream_syn0_c16_s0.npz"), load_golden("stream_syn1_c16_s0.npz")
    syn = torch.from_numpy(T.synth_fbank(2, 1000))
    # n = 64: the configuration bench.py and INTEGRATION.md recommend -- two resident 256-workgroup decoder grids wanted by two
    # contexts on 256 CUs (why that is safe: multi_decoder_ok's comment in host_launch.hip.inc)
    x = torch.stack([syn[i % 2] for i in range(n)]).cuda().contiguous()
    want = [(g0, g1)[i % 2]["tokens"].tolist() for i in range(n)]
    sbs = [StreamingBatch(np_state_dict(0), n, max_chunk_frames=32, max_cache_frames=256, max_enc_frames=256, max_tokens=2048) for _ in range(2)]
    for sb in sbs:
        assert sb.decode_script(x, 16, pipelined=True) == want
    torch.cuda.synchronize()
    bad, streams = [], [torch.cuda.Stream() for _ in range(2)]

    def worker(k):
        with torch.cuda.stream(streams[k]):
            for it in range(6 if n == 16 else 4):
                if sbs[k].decode_script(x, 16, pipelined=True) != want:
                    bad.append((k, it))
        streams[k].synchronize()

    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not bad, bad
