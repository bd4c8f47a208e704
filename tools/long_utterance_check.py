import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ctc_vr_amd.testing as T
from ctc_vr_amd.online_rnnt_model import StreamingBatch
B, N = 8, 4500
sb = StreamingBatch(T.make_state_dict(1, blank_bias=12.0), B, max_chunk_frames=64, max_cache_frames=1200, max_enc_frames=1200, max_tokens=12000)
x = torch.from_numpy(T.synth_fbank(B, N, seed=9)).cuda().contiguous()
for chunk in (16, 32):
    t0 = time.perf_counter(); a = sb.decode_script(x, chunk, pipelined=True); torch.cuda.synchronize(); t1 = time.perf_counter()
    b = sb.decode_script(x, chunk, per_chunk_decode=True); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(chunk, "equal", a == b, "tokens", [len(t) for t in a][:4], f"pipelined {1e3*(t1-t0):.1f} ms, per-chunk {1e3*(t2-t1):.1f} ms")
