import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import ctc_vr_amd.testing as T
from ctc_vr_amd.online_rnnt_model import StreamingBatch
sd = T.make_state_dict(0)
syn = torch.from_numpy(T.synth_fbank(2, 1000))
x = torch.stack([syn[i % 2] for i in range(4)]).cuda().contiguous()
plan = T.chunk_plan(1000, 16)
offs = [4 * i for i in range(len(plan))]
d = lambda a, b: float(np.abs(a.astype(np.float64) - b).max())
for mode in ("fp32", "bf16x3"):
    for greedy_first in (0, 1):
        sb = StreamingBatch(sd, 4, max_chunk_frames=32, max_cache_frames=256, max_enc_frames=256, numerics=mode)
        s = torch.cuda.current_stream().cuda_stream
        sb.reset()
        for (a, b) in plan:
            sb.process_chunk(x[:, a:b].contiguous(), decode=False)
        e_seq = sb.engine.enc_frames(s).copy()
        sb.engine.greedy_decode(s)
        if greedy_first:
            toks = sb.decode_script(x, 16, pipelined=True)
        sb.reset()
        sb.engine.encoder_chunks(x.data_ptr(), 1000, [a for a, _ in plan], [b - a for a, b in plan], offs, offs, s)
        e_wf = sb.engine.enc_frames(s).copy()
        dd = np.abs(e_seq.astype(np.float64) - e_wf)
        idx = np.unravel_index(dd.argmax(), dd.shape)
        print(mode, "greedy_first", greedy_first, "seq-vs-wf", d(e_seq, e_wf), "argmax at (stream, frame, col)", idx, "frames with diff>1e-4:", np.unique(np.where(dd > 1e-4)[1])[:20], flush=True)
        del sb
