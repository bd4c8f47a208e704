import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import ctc_vr_amd.testing as T
from ctc_vr_amd.online_rnnt_model import StreamingBatch
sd = T.make_state_dict(0)
x = torch.from_numpy(T.synth_fbank(4, 1000)).cuda().contiguous()
plan = T.chunk_plan(1000, 16)
offs = [4 * i for i in range(len(plan))]
mode = os.environ.get("MODE", "bf16x3")
sb = StreamingBatch(sd, 4, max_chunk_frames=32, max_cache_frames=256, max_enc_frames=256, numerics=mode)
s = torch.cuda.current_stream().cuda_stream
sb.reset()
for (a, b) in plan:
    sb.process_chunk(x[:, a:b].contiguous(), decode=False)
e0 = sb.engine.enc_frames(s).copy()
att0 = sb.engine.att_cache(0, s).copy()
sb.reset()
sb.engine.encoder_chunks(x.data_ptr(), 1000, [a for a, _ in plan], [b - a for a, b in plan], offs, offs, s)
e1 = sb.engine.enc_frames(s).copy()
att1 = sb.engine.att_cache(0, s).copy()
dd = np.abs(e1.astype(np.float64) - e0).max(-1)      # [4, 188]
print(mode, "first frame with any diff per stream:", [int(np.argmax(dd[b] > 0)) if (dd[b] > 0).any() else -1 for b in range(4)], "max", dd.max())
da = np.abs(att1.astype(np.float64) - att0)          # [12,4,len,128]
for l in range(12):
    pr = da[l].max(axis=(0, 2))
    nz = np.nonzero(pr > 0)[0]
    print(" layer", l, "first cache row with diff:", int(nz[0]) if len(nz) else -1, "max", pr.max())
