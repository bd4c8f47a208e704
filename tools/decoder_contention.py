"""What slows the resident decoder down?  Decode the same buffered frames (encoder already done) alone and next to three
synthetic background loads on another stream: an HBM stream (large copies), an L2-resident stream (small copies) and
dense f32 matmuls (rocBLAS; MFMA + L2 + HBM)."""
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
main = torch.cuda.Stream(); side = torch.cuda.Stream()
big_a = torch.empty(256 << 20, dtype=torch.uint8, device="cuda"); big_b = torch.empty_like(big_a)          # 256 MB: HBM
small_a = torch.empty(2 << 20, dtype=torch.uint8, device="cuda"); small_b = torch.empty_like(small_a)      # 2 MB: L2
ma = torch.randn(4096, 4096, device="cuda"); mb = torch.randn(4096, 4096, device="cuda")

import ctypes
_lib = sb.engine.lib
_lib.rnnt_debug_stream_copy.restype = ctypes.c_int
_lib.rnnt_debug_stream_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
def own_copy(mode):
    def f():
        _lib.rnnt_debug_stream_copy(sb.engine.ctx, big_a.data_ptr(), big_b.data_ptr(), big_a.numel(), mode, 40, side.cuda_stream)
    return f
def bg_none(): pass
def bg_hbm():
    for _ in range(40): big_b.copy_(big_a)
def bg_l2():
    for _ in range(4000): small_b.copy_(small_a)
def bg_mm():
    for _ in range(12): torch.mm(ma, mb)
for name, bg in (("alone", bg_none), ("HBM stream (256 MB copies)", bg_hbm), ("L2 stream (2 MB copies)", bg_l2), ("f32 matmul 4096^3", bg_mm), ("own copy, default policy", own_copy(0)),
                 ("own copy, nt loads", own_copy(1)), ("own copy, nt loads + nt stores", own_copy(2))):
    res = []
    for it in range(3):
        with torch.cuda.stream(main):
            sb.reset()
            sb.engine.encoder_chunks(x.data_ptr(), 1000, starts, lens, offs, offs, main.cuda_stream, greedy=False)
        torch.cuda.synchronize()
        e0, e1, b0, b1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        with torch.cuda.stream(side):
            b0.record(); bg(); b1.record()
        with torch.cuda.stream(main):
            e0.record()
            sb.engine.greedy_decode(main.cuda_stream)
            e1.record()
        torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1), b0.elapsed_time(b1)))
    print(f"{name:32s} decoder {res[-1][0]:7.2f} ms   (background busy for {res[-1][1]:7.2f} ms)")
