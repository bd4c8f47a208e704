"""Where does a beam-search pass spend its time? (cProfile of StreamingBatch.beam_script, 64 streams x 10 s, beam 4)"""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ctc_vr_amd.testing as T
from ctc_vr_amd.online_rnnt_model import StreamingBatch
B = 64
sb = StreamingBatch(T.make_state_dict(0, blank_bias=12.0), B, max_chunk_frames=24, max_cache_frames=200, max_enc_frames=16, max_tokens=16, max_beam=4)
x = torch.from_numpy(T.synth_fbank(B, 1000)).cuda().contiguous()
sb.beam_script(x, 16, 4)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
sb.beam_script(x, 16, 4)
torch.cuda.synchronize()
pr.disable()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(18)
print(st.getvalue()[:3500])
