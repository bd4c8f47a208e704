"""Per-chunk streaming call pattern (one call + token read-back per 16-frame chunk, 64 streams): the eager per-chunk path
(rnnt_encoder_chunk) against rnnt_encoder_chunks with a ONE-chunk plan (layer-major schedule, or RNNT_LM=0: fused wavefront)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ctc_vr_amd.testing as T
from ctc_vr_amd.online_rnnt_model import StreamingBatch
B = 64
sb = StreamingBatch(T.make_state_dict(0, blank_bias=12.0), B, max_chunk_frames=24, max_cache_frames=256, max_enc_frames=256, max_tokens=4096, numerics=os.environ.get("NUM", "bf16x3"))
x = torch.from_numpy(T.synth_fbank(B, 1000)).cuda().contiguous()
plan = T.chunk_plan(1000, 16)
s = torch.cuda.current_stream().cuda_stream
ref = sb.decode_script(x, 16, pipelined=True)
def eager():
    sb.reset()
    for (a, b) in plan:
        sb.process_chunk(x[:, a:b, :].contiguous(), decode=True)
        sb.engine.token_counts(s)
    return sb.engine.tokens()
def one_chunk_plans():
    sb.reset()
    off = 0
    for (a, b) in plan:
        c = x[:, a:b, :].contiguous()
        sb.engine.encoder_chunks(c.data_ptr(), b - a, [0], [b - a], [off], [off], s, greedy=True)
        off += (b - a) // 4
        sb.engine.token_counts(s)
    return sb.engine.tokens()
for name, fn in (("eager rnnt_encoder_chunk + greedy_decode", eager), ("rnnt_encoder_chunks(1 chunk, greedy)", one_chunk_plans)):
    try:
        t = fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); t = fn(); torch.cuda.synchronize()
        print(f"{name}: {1e3 * (time.perf_counter() - t0):.1f} ms per 64 x 10 s, tokens equal whole-utterance: {t == ref}")
    except Exception as e:
        print(name, "FAILED:", e)
