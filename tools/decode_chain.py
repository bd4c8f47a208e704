"""Per-stream symbol counts of the bench workload and the stand-alone decoder time (no encoder running)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ctc_vr_amd.testing as T
from ctc_vr_amd.online_rnnt_model import StreamingBatch
B = 64
sb = StreamingBatch(T.make_state_dict(0), B, max_chunk_frames=24, max_cache_frames=256, max_enc_frames=256, max_tokens=4096)
x = torch.from_numpy(T.synth_fbank(B, 1000)).cuda().contiguous()
plan = T.chunk_plan(1000, 16)
starts = [a for a, b in plan]; lens = [b - a for a, b in plan]; offs = [4 * i for i in range(len(plan))]
s = torch.cuda.current_stream().cuda_stream
for it in range(3):
    sb.reset(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    sb.engine.encoder_chunks(x.data_ptr(), 1000, starts, lens, offs, offs, s, greedy=False)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    sb.engine.greedy_decode(s)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"encoder alone {1e3*(t1-t0):.2f} ms, decoder alone {1e3*(t2-t1):.2f} ms, counters {sb.engine.counters()}")
toks = sb.engine.tokens()
n = np.array([len(t) for t in toks])
print("symbols per stream: mean", n.mean(), "max", n.max(), "sorted top", np.sort(n)[-8:], "bottom", np.sort(n)[:4])
