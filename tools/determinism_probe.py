"""Run-to-run reproducibility of the whole-utterance call: encoder frames (bitwise) and greedy tokens over repeated calls."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_vr_amd.testing as T  # noqa: E402
from ctc_vr_amd.online_rnnt_model import StreamingBatch  # noqa: E402

B = int(os.environ.get("PROBE_B", "64"))
N = int(os.environ.get("PROBE_N", "6"))
modes = os.environ.get("PROBE_MODES", "bf16x3,fp32").split(",")
sd = T.make_state_dict(0)
x = torch.from_numpy(T.synth_fbank(B, 1000, seed=1234)).cuda().contiguous()
plan = T.chunk_plan(1000, 16)
offs = [4 * i for i in range(len(plan))]
for mode in modes:
    sb = StreamingBatch(sd, B, max_chunk_frames=24, max_cache_frames=200, max_enc_frames=200, max_tokens=1900, numerics=mode)
    s = torch.cuda.current_stream().cuda_stream
    encs, toks = [], []
    for i in range(N):
        sb.reset()
        sb.engine.encoder_chunks(x.data_ptr(), 1000, [a for a, _ in plan], [b - a for a, b in plan], offs, offs, s)
        encs.append(sb.engine.enc_frames(s).copy())
    for i in range(N):
        toks.append(sb.decode_script(x, 16, pipelined=True))
    for i in range(1, N):
        d = np.abs(encs[i].astype(np.float64) - encs[0])
        bad = np.argwhere(d.max(axis=2) > 0)
        tdiff = [b for b in range(B) if toks[i][b] != toks[0][b]]
        print(f"{mode} run {i}: enc max diff {d.max():.3e}, (stream, frame) pairs differing {len(bad)} first {bad[:6].tolist()}; token streams differing {tdiff}", flush=True)
    del sb
