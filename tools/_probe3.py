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
mode = os.environ.get("MODE", "bf16x3")
sb = StreamingBatch(sd, 4, max_chunk_frames=32, max_cache_frames=256, max_enc_frames=256, numerics=mode)
s = torch.cuda.current_stream().cuda_stream
seqs, wfs = [], []
for rep in range(3):
    sb.reset()
    for (a, b) in plan:
        sb.process_chunk(x[:, a:b].contiguous(), decode=False)
    seqs.append(sb.engine.enc_frames(s).copy())
    sb.reset()
    sb.engine.encoder_chunks(x.data_ptr(), 1000, [a for a, _ in plan], [b - a for a, b in plan], offs, offs, s)
    wfs.append(sb.engine.enc_frames(s).copy())
print(mode, "seq run-to-run:", d(seqs[0], seqs[1]), d(seqs[0], seqs[2]), "| wf run-to-run:", d(wfs[0], wfs[1]), d(wfs[0], wfs[2]),
      "| seq vs wf:", d(seqs[0], wfs[0]), "| stream0 vs stream2 (identical inputs): seq", d(seqs[0][0], seqs[0][2]), "wf", d(wfs[0][0], wfs[0][2]))
