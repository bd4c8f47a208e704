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
mode = os.environ.get("MODE", "bf16x3")
sb = StreamingBatch(sd, 4, max_chunk_frames=32, max_cache_frames=256, max_enc_frames=256, numerics=mode)
s = torch.cuda.current_stream().cuda_stream
sb.reset()
for (a, b) in plan:
    sb.process_chunk(x[:, a:b].contiguous(), decode=False)
e0 = sb.engine.enc_frames(s).copy()
att0 = [sb.engine.att_cache(b, s).copy() for b in range(4)]
for rep in range(2):
    sb.reset()
    sb.engine.encoder_chunks(x.data_ptr(), 1000, [a for a, _ in plan], [b - a for a, b in plan], offs, offs, s)
    e1 = sb.engine.enc_frames(s).copy()
    dd = np.abs(e1.astype(np.float64) - e0)
    bad = np.argwhere(dd.max(-1) > 1e-4)
    print("rep", rep, "bad (stream, frame):", bad.tolist()[:30], "cols>1e-4 per bad row:", [(int((dd[b, f] > 1e-4).sum())) for b, f in bad[:30]])
    for b in range(4):
        a1 = sb.engine.att_cache(b, s)
        da = np.abs(a1.astype(np.float64) - att0[b])          # [12,4,len,128]
        per_layer = da.max(axis=(1, 3))                       # [12, len]
        first = [(l, int(np.argmax(per_layer[l] > 1e-4))) for l in range(12) if (per_layer[l] > 1e-4).any()]
        kv = [(l, float(da[l, :, :, :64].max()), float(da[l, :, :, 64:].max())) for l in range(12) if da[l].max() > 1e-4]
        print("  stream", b, "first bad cache row per layer (layer, row):", first[:12], "K/V maxerr:", [(l, round(k, 5), round(v, 5)) for l, k, v in kv][:12])
