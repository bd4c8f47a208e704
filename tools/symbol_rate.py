import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ctc_vr_amd.testing as T
from ctc_vr_amd.online_rnnt_model import StreamingBatch
B = 64
x = torch.from_numpy(T.synth_fbank(B, 1000, seed=1234)).cuda().contiguous()
for bb in (11.0, 12.0, 13.0, 14.0, 16.0):
    sb = StreamingBatch(T.make_state_dict(0, blank_bias=bb), B, max_chunk_frames=24, max_cache_frames=200, max_enc_frames=200, max_tokens=2000)
    toks = sb.decode_script(x, 16, pipelined=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    toks = sb.decode_script(x, 16, pipelined=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n = np.array([len(t) for t in toks])
    print(f"blank_bias {bb}: symbols/frame mean {n.mean()/188:.2f} max-stream {n.max()/188:.2f} min {n.min()/188:.2f}; slowest stream evals {n.max()+188}; "
          f"uniq tokens {len(set(t for r in toks for t in r))}; pass {dt*1e3:.1f} ms; steps {sb.engine.counters()[1]}")
    del sb
