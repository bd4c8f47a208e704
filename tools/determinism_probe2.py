"""RNNT_LM_DEBUG=1: two whole-utterance encoder calls; the per-launch checksums go to stderr (diff the two halves)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctc_vr_amd.testing as T  # noqa: E402
from ctc_vr_amd.online_rnnt_model import StreamingBatch  # noqa: E402
B = 64
sd = T.make_state_dict(0)
x = torch.from_numpy(T.synth_fbank(B, 1000, seed=1234)).cuda().contiguous()
plan = T.chunk_plan(1000, 16)
offs = [4 * i for i in range(len(plan))]
sb = StreamingBatch(sd, B, max_chunk_frames=24, max_cache_frames=200, max_enc_frames=200, max_tokens=1900, numerics="bf16x3")
s = torch.cuda.current_stream().cuda_stream
for i in range(3):
    sys.stderr.write(f"=== run {i}\n"); sys.stderr.flush()
    sb.reset()
    sb.engine.encoder_chunks(x.data_ptr(), 1000, [a for a, _ in plan], [b - a for a, b in plan], offs, offs, s)
    torch.cuda.synchronize()
