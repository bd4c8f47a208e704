"""Pin the CPU oracle (oracle/rnnt_oracle.py) against vectors produced by the reference itself
(tests/golden/gen_golden.py).  CPU only; this is what makes the oracle a trustworthy checker."""
import numpy as np
import pytest
import torch

import ctc_vr_amd.testing as T
from conftest import load_golden
from oracle import rnnt_oracle as O

TOL = 2e-5   # float32, same torch kernels as the reference -> expected ~0; slack for thread-count effects


@pytest.fixture(scope="module")
def sds(np_state_dict):
    return {s: O.to_torch_sd(np_state_dict(s)) for s in (0, 1)}


def ex_inputs():
    g = load_golden("inputs_example1.npz")
    return {k: torch.from_numpy(g[k])[None] for k in g.files}


def maxdiff(a, b):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0


def test_state_dict_spec_matches_reference_layout(np_state_dict):
    sd = np_state_dict(0)
    assert len(sd) == 504                                      # SURVEY.md §0.1 [ran]
    n = sum(v.size for k, v in sd.items() if "num_batches_tracked" not in k and "pos_enc.pe" not in k)
    assert n == 21953592    # 21,947,448 params + 12x(256+256) BatchNorm running stats
    pe = sd["encoder.embed.pos_enc.pe"]
    assert pe.shape == (1, 5000, 256) and pe.dtype == np.float32
    assert abs(pe[0, 1, 0] - np.sin(1.0)) < 1e-6 and pe[0, 0, 1] == 1.0


@pytest.mark.parametrize("seed", [0, 1])
def test_modules(seed, sds):
    sd = sds[seed]
    g = load_golden(f"modules_seed{seed}.npz")
    x = torch.from_numpy(T.synth_fbank(2, 64, seed=7))
    for tc in (16, 24, 41, 64):
        y = O.subsample(sd, x[:1, :tc])
        assert y.shape[1] == ((tc - 3) // 2 + 1 - 3) // 2 + 1
        assert maxdiff(y, g[f"subsample_T{tc}"]) < 5e-4        # values are O(70) after the x16 scale
    # predictor
    st = O.predictor_init_state(1)
    for i, tok in enumerate(g["pred_tokens"].tolist()):
        o, st = O.predictor_step(sd, torch.tensor([[tok]]), st)
        assert maxdiff(o[0, 0], g["pred_out"][i]) < TOL
        assert maxdiff(st[0][0, 0], g["pred_h"][i]) < TOL
        assert maxdiff(st[1][0, 0], g["pred_c"][i]) < TOL
    # joint lattice
    lg = O.joint(sd, torch.from_numpy(g["joint_enc"]), torch.from_numpy(g["joint_pred"]))
    assert lg.shape == (2, 7, 5, T.VOCAB)
    assert maxdiff(lg, g["joint_logits"]) < 1e-4
    assert maxdiff(torch.log_softmax(lg, -1), g["joint_logp"]) < 1e-4


@pytest.mark.parametrize("seed", [0, 1])
def test_forward_chunk_traces(seed, sds):
    """Three consecutive 16-frame chunks: output, caches (incl. the dropped first-chunk K/V and the
    estimated-offset drift) and layer-0 / layer-11 sub-module activations."""
    sd = sds[seed]
    g = load_golden(f"modules_seed{seed}.npz")
    x = ex_inputs()["ex0"]
    att = torch.zeros(0, 0, 0, 0)
    cnn = torch.zeros(0, 0, 0, 0)
    off = 0
    for ci in range(3):
        tr = {"layers": (0, 11)}
        y, att, cnn = O.forward_chunk(sd, x[:, ci * 16:(ci + 1) * 16], off, off, att, cnn, tr)
        assert tuple(att.shape) == tuple(g[f"fc{ci}_att_cache"].shape)
        assert maxdiff(y, g[f"fc{ci}_out"]) < TOL
        assert maxdiff(att, g[f"fc{ci}_att_cache"]) < 1e-4
        assert maxdiff(cnn, g[f"fc{ci}_cnn_cache"]) < TOL
        for li in (0, 11):
            assert maxdiff(tr[f"layer{li}"]["out"], g[f"fc{ci}_l{li}.norm_final"]) < TOL
        off += 4
    assert tuple(g["fc0_att_cache"].shape) == (12, 4, 0, 128)
    assert tuple(g["fc1_att_cache"].shape) == (12, 4, 3, 128)


STREAMS = ["syn0_c16_s0", "syn1_c16_s0", "syn0_c16_s1", "ex0_c32_s0", "ex0_c32_s1", "ex6_c16_s0",
           "ex12_c64_s0", "ex12_c16_s1"]


def stream_input(name):
    src = name.split("_")[0]
    if src.startswith("syn"):
        return torch.from_numpy(T.synth_fbank(2, 1000))[int(src[3:]):int(src[3:]) + 1]
    return ex_inputs()[src]


@pytest.mark.parametrize("name", STREAMS)
def test_decode_script_greedy(name, sds):
    g = load_golden(f"stream_{name}.npz")
    sd = sds[int(g["seed"])]
    x = stream_input(name)
    assert x.shape[1] == int(g["frames"])
    toks, per, st = O.decode_script_greedy(sd, x, int(g["chunk"]))
    assert toks == g["tokens"].tolist()
    assert [len(r) for r in per] == g["counts"].tolist()
    assert tuple(st.att_cache.shape) == tuple(g["att_cache_shape"])
    assert st.global_offset == int(g["global_offset"])
    assert st.last_token == int(g["last_token"])
    assert maxdiff(st.cnn_cache, g["cnn_cache"]) < TOL
    assert maxdiff(st.att_cache[0, :, -3:, :], g["att_cache_l0_last"]) < 1e-4
    assert maxdiff(st.att_cache[11, :, :3, :], g["att_cache_l11_first"]) < 1e-4
    assert maxdiff(st.predictor_states[0], g["pred_h"]) < TOL
    assert abs(float(st.att_cache.double().sum()) - float(g["att_cache_sum"])) < 1e-2


def test_chunk_plan_and_offsets():
    g = load_golden("stream_syn0_c16_s0.npz")
    plan = T.chunk_plan(1000, 16)
    assert len(plan) == 62 and plan[-1] == (976, 1000) and plan[0] == (0, 16)
    offs = g["offsets"]
    assert offs[:, 0].tolist() == [4 * i for i in range(62)]                    # estimated offsets 0,4,...,244
    assert offs[:, 2].tolist() == [0, 0] + [3 * i for i in range(1, 61)]        # cache_t1 per chunk
    assert g["enc_frames"].tolist() == [3] * 61 + [5]
    plan32 = T.chunk_plan(521, 32)
    assert len(plan32) == 16 and plan32[-1] == (480, 521)                       # config 1: 15x32 + 41


@pytest.mark.parametrize("name", ["si_ex0_scs16_s0", "si_syn0_scs16_s0", "si_ex6_scs32_ms200_s0"])
def test_streaming_inference(name, sds):
    g = load_golden(f"stream_{name}.npz")
    sd = sds[int(g["seed"])]
    x = stream_input(name[3:])
    st = O.OracleStream(sd, T.BLANK, int(g["static_chunk_size"]))
    ms = int(g["chunk_size_ms"])
    toks = st.streaming_inference(x, x.shape[1], None if ms < 0 else ms)
    assert toks == g["tokens"].tolist()
    assert tuple(st.att_cache.shape) == tuple(g["att_cache_shape"])


@pytest.mark.parametrize("name", ["beam_si_ex6_scs16_s0", "beam_si_syn1_scs16_s1_f400", "beam_si_ex6_scs32_ms200_s0"])
def test_streaming_beam_search(name, sds):
    """OnlineRNNTModel.streaming_beam_search (model/online_rnnt_model.py:534-603): best hypothesis and every final hypothesis
    (tokens in the reference's order, Python-double scores)."""
    g = load_golden(f"{name}.npz")
    sd = sds[int(g["seed"])]
    x = stream_input(name[8:])[:, :int(g["frames"])]
    st = O.OracleStream(sd, T.BLANK, int(g["static_chunk_size"]))
    ms = int(g["chunk_size_ms"])
    toks = st.streaming_beam_search(x, x.shape[1], int(g["beam"]), None if ms < 0 else ms)
    assert toks == g["tokens"].tolist()
    cnt, flat = g["hyp_counts"].tolist(), g["hyp_tokens"].tolist()
    want = [flat[sum(cnt[:k]):sum(cnt[:k + 1])] for k in range(len(cnt))]
    assert [h.tokens for h in st.beam] == want
    assert np.allclose([h.log_prob for h in st.beam], g["hyp_logp"], atol=2e-3)


@pytest.mark.parametrize("name", ["beam_ex6_c16_s0", "beam_syn0_c16_s1_f320", "beam_ex0_c32_s0"])
def test_beam_search(name, sds):
    g = load_golden(f"{name}.npz")
    sd = sds[int(g["seed"])]
    x = stream_input(name[5:])[:, :int(g["frames"])]
    chunk, beam = int(g["chunk"]), int(g["beam"])
    st = O.OracleStream(sd, T.BLANK, chunk)
    stats = {}
    for ci, (s, e) in enumerate(T.chunk_plan(x.shape[1], chunk)):
        hyps = st.process_single_chunk_beam_search(x[:, s:e], beam, stats)
        assert len(hyps) == int(g[f"c{ci}_n"])
        for hi, h in enumerate(hyps):
            assert h.tokens == g[f"c{ci}_h{hi}_tokens"].tolist(), (ci, hi)
            assert abs(h.log_prob - float(g[f"c{ci}_h{hi}_logp"])) < 1e-3
    assert stats["evals"] == int(g["joint_calls"])


def test_encoder_full_context(sds):
    sd = sds[0]
    g = load_golden("full_seed0.npz")
    x = torch.from_numpy(T.synth_fbank(2, 300, seed=int(g["fbank_seed"])))
    y, m = O.encoder_full(sd, x, torch.from_numpy(g["lens"]))
    assert np.array_equal(m.numpy(), g["mask"])
    valid = m[:, 0, :, None].numpy()
    assert maxdiff(y.numpy() * valid, g["out"] * valid) < 1e-4
    g6 = load_golden("full_ex6_seed0.npz")
    x6 = ex_inputs()["ex6"]
    y6, _ = O.encoder_full(sd, x6, torch.tensor([x6.shape[1]]))
    assert maxdiff(y6, g6["out"]) < 1e-4


@pytest.mark.parametrize("seed", [0, 1])
def test_offline_greedy_search(seed, np_state_dict):
    """basic_greedy_search (model/component/transducer.py:22-70) on the full-context encoder, n_steps 64 and 3."""
    g = load_golden(f"offline_greedy_seed{seed}.npz")
    sd = O.to_torch_sd(np_state_dict(seed))
    x = torch.from_numpy(T.synth_fbank(3, 240, seed=77 + seed))
    lens = torch.from_numpy(g["lens"])
    for n_steps in (64, 3):
        hyps = O.basic_greedy_search_full(sd, x, lens, blank=T.BLANK, n_steps=n_steps)
        cnt = g[f"counts_n{n_steps}"]
        assert [len(h) for h in hyps] == cnt.tolist()
        assert [t for h in hyps for t in h] == g[f"tokens_n{n_steps}"].tolist()


@pytest.mark.parametrize("seed", [0, 1])
def test_ctc_head_matches_reference(seed):
    """oracle ctc_greedy_search_full vs the reference's ctc_greedy_search / OnlineCTC.argmax on the deterministic full-context
    encoder (golden ctc_seed*.npz)."""
    g = load_golden(f"ctc_seed{seed}.npz")
    sd = O.to_torch_sd(T.make_state_dict(seed))
    x = torch.from_numpy(T.synth_fbank(2, 300, seed=int(g["fbank_seed"])))
    lens = torch.from_numpy(g["lens"])
    got = O.ctc_greedy_search_full(sd, x, lens, T.BLANK)
    want, o = [], 0
    for c in g["hyp_counts"].tolist():
        want.append(g["hyp_tokens"][o:o + c].tolist())
        o += c
    assert got == want
    enc, mask = O.encoder_full(sd, x, lens)
    lp = torch.log_softmax(torch.nn.functional.linear(enc, sd["ctc_head.ctc_lo.weight"], sd["ctc_head.ctc_lo.bias"]), dim=2)
    valid = g["mask"][:, 0, :]
    assert np.array_equal(lp.argmax(2).numpy()[valid], g["ids"][valid])
    assert np.abs(lp[:, :8].numpy() - g["logp_first8"]).max() < 1e-4


@pytest.mark.parametrize("seed", ["0", "1", "0_padded"])
def test_prefix_beam_search_matches_reference(seed):
    """oracle prefix_beam_search_full vs the reference's PrefixBeamSearch.prefix_beam_search run on the model's own encoder /
    predictor / joint / CTC head (golden prefix_beam_seed*.npz; the generator rebinds the name log_add in that module to the list
    form its call site uses, see gen_golden.py): hypotheses exact, scores to 1e-4, final predictor states to 1e-4."""
    g = load_golden(f"prefix_beam_seed{seed}.npz")    # "_padded": audio_lens < T, the search still walks every encoder frame
    sd = O.to_torch_sd(T.make_state_dict(int(seed[0])))
    x = torch.from_numpy(T.synth_fbank(1, int(g["frames"]), seed=int(g["fbank_seed"])))
    beam = O.prefix_beam_search_full(sd, x, torch.tensor([int(g["valid_frames"])]), T.BLANK, beam_size=int(g["beam"]))
    want, o = [], 0
    for c in g["hyp_counts"].tolist():
        want.append(g["hyp_tokens"][o:o + c].tolist())
        o += c
    assert [b[0] for b in beam] == want
    assert max(abs(b[1] - s) for b, s in zip(beam, g["scores"])) < 1e-4
    assert np.abs(torch.cat([b[2][0] for b in beam], 1).numpy() - g["h"]).max() < 1e-4
