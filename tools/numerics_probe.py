"""Encoder-output error of every numerics mode: per-chunk API vs whole-utterance call vs the fp32 run of the same path."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_vr_amd.testing as T  # noqa: E402
from ctc_vr_amd.online_rnnt_model import StreamingBatch  # noqa: E402

B = int(os.environ.get("PROBE_B", "4"))
sd = T.make_state_dict(0)
x = torch.from_numpy(T.synth_fbank(B, 1000)).cuda().contiguous()
plan = T.chunk_plan(1000, 16)
offs = [4 * i for i in range(len(plan))]
ref = {}
for mode in ("fp32", "bf16x3", "f16x3", "bf16"):
    sb = StreamingBatch(sd, B, max_chunk_frames=32, max_cache_frames=256, max_enc_frames=256, numerics=mode)
    s = torch.cuda.current_stream().cuda_stream
    sb.reset()
    for (a, b) in plan:
        sb.process_chunk(x[:, a:b].contiguous(), decode=False)
    e_seq = sb.engine.enc_frames(s).copy()
    sb.engine.greedy_decode(s)
    t_seq = sb.engine.tokens(s)
    sb.reset()
    sb.engine.encoder_chunks(x.data_ptr(), 1000, [a for a, _ in plan], [b - a for a, b in plan], offs, offs, s)
    e_wf = sb.engine.enc_frames(s).copy()
    sb.engine.greedy_decode(s)
    t_wf = sb.engine.tokens(s)
    if mode == "fp32":
        ref = {"seq": e_seq, "wf": e_wf, "tok": t_seq}
    d = lambda a, b: float(np.abs(a.astype(np.float64) - b).max())
    r = lambda a, b: float(np.sqrt(np.mean((a.astype(np.float64) - b) ** 2)))
    same = sum(int(t_seq[i] == ref["tok"][i]) for i in range(B))
    same_wf = sum(int(t_wf[i] == ref["tok"][i]) for i in range(B))
    print(f"{mode:7s} seq-vs-wf max {d(e_seq, e_wf):.2e} | seq vs fp32 max {d(e_seq, ref['seq']):.2e} rms {r(e_seq, ref['seq']):.2e} | "
          f"wf vs fp32 max {d(e_wf, ref['wf']):.2e} rms {r(e_wf, ref['wf']):.2e} | streams with fp32 tokens: seq {same}/{B} wf {same_wf}/{B}", flush=True)
    del sb
