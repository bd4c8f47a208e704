"""Throughput with two batches in flight in ONE process: two contexts, two host threads (ctypes releases the GIL), each
running whole-utterance passes on its own torch stream.  Does the second batch's encoder fill the first batch's decode tail?"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ctc_vr_amd.testing as T
from ctc_vr_amd.online_rnnt_model import StreamingBatch
B, STEPS = 64, 6
sd = T.make_state_dict(0, blank_bias=12.0)
x = torch.from_numpy(T.synth_fbank(B, 1000)).cuda().contiguous()
def make():
    return StreamingBatch(sd, B, max_chunk_frames=24, max_cache_frames=200, max_enc_frames=200, max_tokens=1900, numerics="bf16x3")
def worker(sb, n, out):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(n):
            out.append(sb.decode_script(x, 16, pipelined=True))
    st.synchronize()
for nctx in (1, 2):
    sbs = [make() for _ in range(nctx)]
    for sb in sbs:
        worker(sb, 1, [])
    torch.cuda.synchronize()
    outs = [[] for _ in sbs]
    th = [threading.Thread(target=worker, args=(sb, STEPS, o)) for sb, o in zip(sbs, outs)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = all(o == outs[0][0] for oo in outs for o in oo)
    print(f"{nctx} context(s): {nctx * STEPS} batches in {dt * 1e3:.1f} ms = {dt * 1e3 / (nctx * STEPS):.2f} ms per batch, {nctx * STEPS * B * 1000 / dt / 1e6:.2f} M frames/s, tokens identical: {same}")
    del sbs
