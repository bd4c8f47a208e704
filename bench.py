#!/usr/bin/env python3
"""bench.py — streaming RNN-T greedy decode throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: 64 independent 10 s streams per GPU
(synthetic 80-dim fbank already resident in HBM), 16-frame chunks with online_rnnt_decode.py's
slicing/offset rules (BASELINE.json configs[1]), chunked Conformer encoder + greedy RNN-T decode.  Default mode
"pipelined": the whole chunk plan goes to rnnt_encoder_chunks (wavefront over chunk x layer, decode overlapped on a
second stream), tokens copied back once per utterance batch -- the same tokens as the per-chunk API, whose
throughput (tokens to the host after EVERY chunk, the reference script's loop) is reported beside it.  Streams
shard across ranks with no data-path collective (weak scaling); weights are broadcast once over RCCL.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import ctc_vr_amd.testing as T  # noqa: E402

# launch-site tags of rnnt_profile_begin (include/rnnt_hip.h)
TAGS = {"conv1": 1, "conv2": 2, "embed": 3, "ffn1": 4, "ffn2": 5, "qkv": 6, "attn": 7, "attn_out": 8, "pw1": 9,
        "dwconv": 10, "pw2": 11, "enc_proj": 13, "lstm": 20, "pred_proj": 21, "joint_tanh": 22, "joint_out": 23}
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0


def sub_len(t):
    return ((t - 3) // 2 + 1 - 3) // 2 + 1


# kernel behind each launch site (ctc-vr_amd/csrc/rnnt_kernels.hip.h) and the prefix rocprofv3 prints for it
SITE_KERNEL = {"conv2": "gemm_ns<2,2,32> (conv2 implicit GEMM)", "ffn1": "gemm_ns_tab<1,2,32> (ffn w_1)", "ffn2": "gemm_ns_tab<1,1,64> (ffn w_2)",
               "qkv": "gemm_ns_tab<1,1,32> (linear_q/k/v)", "attn_out": "gemm_ns_tab<1,1,32> (linear_out)", "pw1": "gemm_ns_tab<1,2,32> (pointwise_conv1)",
               "pw2": "gemm_ns_tab<1,1,32> (pointwise_conv2)", "attn": "rel_attention_stream_tab", "dwconv": "dwconv_bn_silu_tab"}
SITE_PMC_PREFIX = {"conv2": "void gemm_ns<2, 2, 32", "ffn2": "void gemm_ns_tab<1, 1, 64", "attn": "rel_attention_stream_tab", "dwconv": "dwconv_bn_silu_tab"}


def pmc_traffic(site):
    """HBM-side bytes per launch of the site's kernel from the committed PMC summary (profiles/, rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate passes of this same command, tools/pmc_summary.py); None if there is no summary for it."""
    pref = SITE_PMC_PREFIX.get(site)
    if pref is None:
        return None
    here = os.path.dirname(os.path.abspath(__file__))
    cands = sorted(f for f in os.listdir(os.path.join(here, "profiles")) if f.endswith("pmc_traffic.json")) if os.path.isdir(os.path.join(here, "profiles")) else []
    if not cands:
        return None
    d = json.load(open(os.path.join(here, "profiles", cands[-1])))
    for k, v in d["kernels"].items():
        if k.startswith(pref):
            return {"traffic_bytes_per_launch": v["traffic_bytes_per_launch"], "source": f"profiles/{cands[-1]} (PMC, separate passes; 2 x FETCH_SIZE + WRITE_SIZE)"}
    return None


def site_flops_bytes(site, B, plan):
    """Algorithmic FLOPs (2*MAC) and bytes of all launches of one site in one step."""
    fl = by = 0.0
    n = 0
    t2 = 0
    for i, (a, b) in enumerate(plan):
        tq = sub_len(b - a)
        t1 = (b - a - 3) // 2 + 1
        M = B * tq
        kv = t2 + tq            # keys seen by this chunk
        per = {"conv2": (2.0 * M * 19 * 256 * 2304, 4.0 * (B * t1 * 39 * 256 + 256 * 2304 + M * 19 * 256), 1),
               "embed": (2.0 * M * 4864 * 256, 4.0 * (M * 4864 + 4864 * 256 + M * 256), 1),
               "ffn1": (2.0 * M * 256 * 1024, 4.0 * (M * 256 + 256 * 1024 + M * 1024), 24),
               "ffn2": (2.0 * M * 256 * 1024, 4.0 * (M * 1024 + 256 * 1024 + 2 * M * 256), 24),
               "qkv": (2.0 * M * 256 * 768, 4.0 * (M * 256 + 3 * 256 * 256 + 3 * M * 256), 12),
               "attn": (2.0 * B * 4 * tq * kv * 64 * 3, 4.0 * (2 * B * kv * 256 + 2 * M * 256 + kv * 256), 12),
               "attn_out": (2.0 * M * 256 * 256, 4.0 * (M * 256 + 256 * 256 + 2 * M * 256), 12),
               "pw1": (2.0 * M * 256 * 512, 4.0 * (M * 256 + 512 * 256 + M * 256), 12),
               "dwconv": (2.0 * M * 256 * 31, 4.0 * (B * (30 + tq) * 256 + 2 * M * 256), 12),
               "pw2": (2.0 * M * 256 * 256, 4.0 * (M * 256 + 256 * 256 + 2 * M * 256), 12)}
        if site in per:
            f, y, cnt = per[site]
            fl += f * cnt
            by += y * cnt
            n += cnt
        t2 = kv if i > 0 else 0    # first chunk's K/V are dropped (required_cache_size = 0)
    return fl, by, n


def side_workload(args, sd_np, dev, world, rank):
    """BASELINE configs[2] (beam 4) and configs[4] (full-context encoder): secondary lines, same JSON shape."""
    from ctc_vr_amd.online_rnnt_model import StreamingBatch
    from ctc_vr_amd.lib import RnntEngine
    assert world == 1
    if args.workload == "beam":
        B = args.batch
        plan = T.chunk_plan(args.frames, args.chunk)
        enc_frames = sum(sub_len(b - a) for a, b in plan)
        sb = StreamingBatch(sd_np, B, max_chunk_frames=max(b - a for a, b in plan), max_cache_frames=enc_frames + 8, max_enc_frames=enc_frames + 8,
                            max_tokens=16, device=0, max_beam=4)
        x = torch.from_numpy(T.synth_fbank(B, args.frames, seed=1234)).to(dev).contiguous()

        def timed(**kw):
            for _ in range(args.warmup):
                sb.beam_script(x, args.chunk, 4, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                beams = sb.beam_script(x, args.chunk, 4, **kw)
            torch.cuda.synchronize()
            return time.perf_counter() - t0, beams
        el, beams = timed(pipelined=True)
        el_pc, beams_pc = timed()
        same = all([h.tokens for h in beams[b]] == [h.tokens for h in beams_pc[b]] for b in range(B))
        out = {"metric": "audio-frames/sec, streaming RNN-T beam search (beam 4)", "value": round(B * args.frames * args.steps / el, 1),
               "unit": "audio-frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 2),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"configs[2]: batch={B} beam_search beam=4, streaming chunk={args.chunk}; whole-utterance encoder call + one "
                                      f"rnnt_beam_advance (resident extension-chain kernel per hypothesis, C++ bookkeeping in the reference's order)",
                          "best_tokens_stream0": len(max(beams[0], key=lambda h: h.log_prob).tokens)},
               "per_chunk_api": {"value": round(B * args.frames * args.steps / el_pc, 1), "ms_per_step": round(el_pc / args.steps * 1e3, 2),
                                 "hypotheses_equal_pipelined": bool(same)}}
        print(json.dumps(out))
        return
    if args.workload == "joint_lattice":
        # SURVEY.md §8d bench shape of the T x U joint: B=64, T=249 (10 s full-context frames), U=28; output-dominated
        B, Tn, U, V = 64, 249, 28, T.VOCAB
        eng = RnntEngine(max_streams=B, max_chunk_frames=16, max_cache_frames=8, max_enc_frames=Tn + 64, vocab_size=V, blank_id=T.BLANK, device=0)
        eng.load_state_dict(sd_np)
        g = torch.Generator(device="cpu").manual_seed(5)
        enc = torch.randn(B, Tn, 256, generator=g).to(dev)
        prd = (torch.randn(B, U, 256, generator=g) * 0.5).to(dev)
        out_t = torch.empty(B, Tn, U, V, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        res = {}
        for mode in (0, 1):
            for _ in range(args.warmup):
                eng.joint(enc.data_ptr(), prd.data_ptr(), B, Tn, U, mode, out_t.data_ptr(), s)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                eng.joint(enc.data_ptr(), prd.data_ptr(), B, Tn, U, mode, out_t.data_ptr(), s)
            torch.cuda.synchronize()
            res[mode] = (time.perf_counter() - t0) / args.steps
        cells = B * Tn * U
        byts = 4.0 * (B * Tn * 256 + B * U * 256) + 4.0 * (2 * 256 * 256 + 256 * V + 2 * 256 + V) + 4.0 * cells * V
        fl = 2.0 * cells * 256 * V
        out = {"metric": "joint lattice cells/sec (add+tanh+projection, logits)", "value": round(cells / res[0], 1), "unit": "lattice-cells/s",
               "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(res[0] * 1e3, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"joint lattice B={B} T={Tn} U={U} V={V} (SURVEY.md §8d), fused tanh-add A-prologue + f32 MFMA projection",
                          "log_softmax_ms": round(res[1] * 1e3, 3)},
               "roofline": {"bound": "hbm", "kernel": "rnnt_joint (3 launches)", "achieved": round(byts / res[0] / 1e9, 1), "peak": PEAK_HBM_GBS,
                            "unit": "GB/s", "frac": round(byts / res[0] / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
                            "mfma_tflops": round(fl / res[0] / 1e12, 1), "mfma_frac": round(fl / res[0] / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                            "note": "exact-f32 MFMA makes this shape compute-bound (128 FLOP per output byte vs 19.7 at the f32 peak)"}}
        print(json.dumps(out))
        return
    # full context: B x 30 s, decoding_chunk_size = -1
    B, Tn = 32, 3000
    eng = RnntEngine(max_streams=B, max_chunk_frames=Tn, max_cache_frames=760, max_enc_frames=8, vocab_size=T.VOCAB, blank_id=T.BLANK, device=0)
    eng.load_state_dict(sd_np)
    x = torch.from_numpy(T.synth_fbank(B, Tn, seed=1234)).to(dev).contiguous()
    tq = sub_len(Tn)
    out_t = torch.empty(B, tq, 256, device=dev)
    lens = np.full(B, Tn, np.int32)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(args.warmup):
        eng.encoder_full(x.data_ptr(), lens, B, Tn, out_t.data_ptr(), s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.encoder_full(x.data_ptr(), lens, B, Tn, out_t.data_ptr(), s)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    flops = B * 2.0 * (tq * (11.2e6 + 1.25e6 + 0.18e6 + 18.183168e6 + 9216.0 * tq))
    out = {"metric": "audio-frames/sec, full-context Conformer encoder", "value": round(B * Tn * args.steps / el, 1), "unit": "audio-frames/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 2), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"configs[4]: full-context encoder, batch={B} x 30 s (decoding_chunk_size=-1), {tq} frames per utterance"},
           "roofline": {"bound": "mfma", "kernel": "whole encoder pass", "achieved": round(flops * args.steps / el / 1e12, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(flops * args.steps / el / 1e12 / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None}}
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="streams per GPU")
    ap.add_argument("--frames", type=int, default=1000, help="fbank frames per stream (10 ms each)")
    ap.add_argument("--chunk", type=int, default=16, help="fbank frames per chunk (online_rnnt_decode.py semantics)")
    ap.add_argument("--mode", default="pipelined", choices=["per_chunk", "deferred", "pipelined"],
                    help="per_chunk: tokens returned to the host after every chunk; deferred: one decode after the last chunk; "
                         "pipelined: whole chunk plan in one rnnt_encoder_chunks call (wavefront over chunk x layer), one decode")
    ap.add_argument("--site", default="conv2", choices=sorted(TAGS), help="launch site timed for the roofline object")
    ap.add_argument("--cpu-streams", type=int, default=8, help="streams of the same workload timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--cpu-threads", type=int, default=8, help="torch threads for the CPU oracle (8 = the reference survey's setting; "
                                                                "B=1 ops are tiny, more threads are slower)")
    ap.add_argument("--also-per-chunk", type=int, default=1, help="also time the per-chunk API mode (reported as extra fields)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--blank-bias", type=float, default=12.0,
                    help="bias on the blank logit of the seeded weights: sets the greedy symbol rate (SURVEY.md §8d asks for 0.3-1 symbols per "
                         "encoder frame; 12.0 gives 0.75, the fixtures' 11.0 gives 1.40, 14.0 gives 0.17)")
    ap.add_argument("--workload", default="greedy", choices=["greedy", "beam", "full_context", "joint_lattice"],
                    help="greedy = BASELINE configs[1] (default); beam = configs[2] (beam 4, per-chunk); full_context = configs[4]")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the multi-rank path on a one-GPU box: BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses gloo
    # (RCCL refuses two ranks on one device); never set by the driver
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm

    # ---- weights: generated on rank 0, ONE RCCL broadcast of the packed blob over xGMI --------------------
    import ctc_vr_amd.dist as D
    sd0 = T.make_state_dict(0, blank_bias=args.blank_bias) if rank == 0 else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sd_np = D.broadcast_state_dict(sd0, src=0, device=dev)
    torch.cuda.synchronize()
    bcast_ms = (time.perf_counter() - t0) * 1e3 if world > 1 else 0.0

    if args.workload != "greedy":
        return side_workload(args, sd_np, dev, world, rank)
    from ctc_vr_amd.online_rnnt_model import StreamingBatch
    B = args.batch
    plan = T.chunk_plan(args.frames, args.chunk)
    enc_frames = sum(sub_len(b - a) for a, b in plan)
    sb = StreamingBatch(sd_np, B, max_chunk_frames=max(b - a for a, b in plan), max_cache_frames=enc_frames + 8,
                        max_enc_frames=enc_frames + 8, max_tokens=enc_frames * 10 + 16, device=local_rank)
    x = torch.from_numpy(T.synth_fbank(B, args.frames, seed=1234 + rank)).to(dev).contiguous()   # inputs resident in HBM
    per_chunk = args.mode == "per_chunk"

    def step():
        return sb.decode_script(x, args.chunk, per_chunk_decode=per_chunk, pipelined=args.mode == "pipelined")

    side = torch.cuda.Stream(device=dev) if os.environ.get("BENCH_SIDE_STREAM", "1") == "1" else None
    if side is not None:   # never launch the path on the legacy null stream (implicit cross-stream synchronisation)
        side.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(side)
    toks = None
    for _ in range(args.warmup):
        toks = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    sb.engine.profile_begin(TAGS[args.site])
    t0 = time.perf_counter()
    for _ in range(args.steps):
        toks = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    site_ms, site_launches = sb.engine.profile_end()
    launches, gsteps = sb.engine.counters()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_frames = world * B * args.frames
    value = total_frames * args.steps / elapsed
    syms = float(np.mean([len(t) for t in toks])) / enc_frames
    fl, by, n_per_step = site_flops_bytes(args.site, B, plan)
    roofline = None
    if site_launches > 0:
        avg_s = site_ms * 1e-3 / site_launches
        site_s = site_ms * 1e-3           # summed duration of every timed launch of the site over the timed region
        hbm = args.site in ("attn", "dwconv")
        if hbm:
            ach = by * args.steps / site_s / 1e9
            roofline = {"bound": "hbm", "kernel": SITE_KERNEL.get(args.site, args.site), "achieved": round(ach, 1), "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None}
        else:
            ach = fl * args.steps / site_s / 1e12
            roofline = {"bound": "mfma", "kernel": SITE_KERNEL.get(args.site, args.site), "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None}
        lps = site_launches / args.steps      # launches of this site per step (the wavefront schedule groups many chunk-layer pairs per launch)
        roofline["algorithmic_flops_per_launch"] = round(fl / lps)
        roofline["algorithmic_bytes_per_launch"] = round(by / lps)
        tr = pmc_traffic(args.site) if (args.batch, args.frames, args.chunk, args.mode) == (64, 1000, 16, "pipelined") else None   # counters were taken on the default workload
        if tr is not None:
            roofline["traffic"] = tr["traffic_bytes_per_launch"]
            roofline["traffic_source"] = tr["source"]
        roofline["avg_launch_us"] = round(avg_s * 1e6, 2)
        roofline["launches_timed"] = int(site_launches)
        roofline["share_of_step"] = round(site_ms * 1e-3 / elapsed, 4)

    # The other launch sites, two extra passes each in the same overlapped pipeline (outside the timed region): the FFN and
    # projection GEMMs together take more of the step than conv2 does, at a lower fraction of the peak.
    other_sites = None
    if roofline is not None and world == 1 and args.mode == "pipelined" and args.site == "conv2":
        other_sites = {}
        for site in ("ffn1", "ffn2", "qkv", "attn_out", "pw1", "pw2", "attn", "dwconv"):
            sb.engine.profile_begin(TAGS[site])
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            ms, nl = sb.engine.profile_end()
            f2, b2, _ = site_flops_bytes(site, B, plan)
            if nl > 0:
                hbm2 = site in ("attn", "dwconv")
                ach2 = (b2 if hbm2 else f2) * 2 / (ms * 1e-3) / (1e9 if hbm2 else 1e12)
                pk = PEAK_HBM_GBS if hbm2 else PEAK_F32_MFMA_TFLOPS
                other_sites[site] = {"kernel": SITE_KERNEL.get(site, site), "bound": "hbm" if hbm2 else "mfma", "achieved": round(ach2, 2),
                                     "unit": "GB/s" if hbm2 else "TFLOP/s", "frac": round(ach2 / pk, 4), "avg_launch_us": round(ms * 1e3 / nl, 2),
                                     "ms_per_step": round(ms / 2, 3)}

    # The same kernel with nothing else on the device: in the timed region the subsampling stream runs UNDER the encoder
    # stages and next to the resident decoder, which is good for the step time and bad for this one kernel's duration.
    # A second context with the overlaps switched off times it alone (same inputs, same launches, outside the timed region).
    roofline_isolated = None
    encoder_only = None
    if roofline is not None and world == 1 and args.mode == "pipelined":
        saved = {k: os.environ.get(k) for k in ("RNNT_WF_SUB_ASYNC", "RNNT_WF_GROUPS")}
        os.environ["RNNT_WF_SUB_ASYNC"] = "0"
        os.environ["RNNT_WF_GROUPS"] = "1"
        sb2 = StreamingBatch(sd_np, B, max_chunk_frames=max(b - a for a, b in plan), max_cache_frames=enc_frames + 8,
                             max_enc_frames=enc_frames + 8, max_tokens=enc_frames * 10 + 16, device=local_rank)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        plan_args = ([a for a, _ in plan], [b - a for a, b in plan], [4 * i for i in range(len(plan))])
        cs = torch.cuda.current_stream().cuda_stream

        def enc_only():
            sb2.reset()
            sb2.engine.encoder_chunks(x.data_ptr(), args.frames, plan_args[0], plan_args[1], plan_args[2], plan_args[2], cs, greedy=False)
        enc_only()
        torch.cuda.synchronize()
        sb2.engine.profile_begin(TAGS[args.site])
        t_iso = time.perf_counter()
        for _ in range(2):
            enc_only()
        torch.cuda.synchronize()
        t_iso = (time.perf_counter() - t_iso) / 2
        iso_ms, iso_n = sb2.engine.profile_end()
        # the whole encoder (no decoder, one stream): all dense contractions + attention, algorithmic FLOPs of SURVEY.md §8d
        enc_flops = sum(site_flops_bytes(k, B, plan)[0] for k in ("conv2", "embed", "ffn1", "ffn2", "qkv", "attn", "attn_out", "pw1", "dwconv", "pw2"))
        encoder_only = {"ms": round(t_iso * 1e3, 3), "algorithmic_tflop": round(enc_flops / 1e12, 4), "achieved": round(enc_flops / t_iso / 1e12, 2),
                        "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(enc_flops / t_iso / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                        "note": "whole chunked encoder of the batch, single stream, no decoder; wall clock around rnnt_encoder_chunks"}
        if iso_n > 0:
            hbm = roofline["bound"] == "hbm"
            ach = (by if hbm else fl) * 2 / (iso_ms * 1e-3) / (1e9 if hbm else 1e12)
            peak = PEAK_HBM_GBS if hbm else PEAK_F32_MFMA_TFLOPS
            roofline_isolated = {"achieved": round(ach, 2), "peak": peak, "unit": roofline["unit"], "frac": round(ach / peak, 4),
                                 "avg_launch_us": round(iso_ms * 1e3 / iso_n, 2), "launches_timed": int(iso_n),
                                 "note": "same kernel and launches, encoder only, subsampling stream and layer-group streams off (nothing else on the device)"}
        del sb2

    per_chunk_extra = None
    if args.also_per_chunk and args.mode != "per_chunk" and world == 1:
        sb.decode_script(x, args.chunk, per_chunk_decode=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        toks_pc = sb.decode_script(x, args.chunk, per_chunk_decode=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        per_chunk_extra = {"value": round(B * args.frames / dt, 1), "ms_per_step": round(dt * 1e3, 3), "tokens_equal_pipelined": toks_pc == toks,
                           "note": "tokens copied to the host after every chunk (online_rnnt_decode.py loop, process_single_chunk API)"}

    cpu = None
    if not args.no_cpu and world == 1:
        from oracle import rnnt_oracle as O   # CPU baseline leg ONLY (checker / baseline, never the product path)
        torch.set_num_threads(args.cpu_threads)
        sd_t = O.to_torch_sd(sd_np)
        xc = x[:args.cpu_streams].cpu()
        t0 = time.perf_counter()
        ok = True
        with torch.no_grad():
            for b in range(xc.size(0)):
                want, _, _ = O.decode_script_greedy(sd_t, xc[b:b + 1], args.chunk)
                ok = ok and (want == toks[b])
        ct = time.perf_counter() - t0
        cpu = {"value": round(xc.size(0) * args.frames / ct, 1), "unit": "audio-frames/s", "cores": torch.get_num_threads(),
               "kind": "port", "sample": f"{xc.size(0)} of the {B} streams x {args.frames} frames, B=1 serial (the reference cannot batch), "
                                         f"oracle/rnnt_oracle.py torch-CPU float32",
               "tokens_match_gpu": bool(ok)}

    out = {
        "metric": "audio-frames/sec, streaming RNN-T greedy decode",
        "value": round(value, 1),
        "unit": "audio-frames/s",
        "rtfx": round(value / 100.0, 1),
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"configs[1]: batch={B}/GPU synthetic {args.frames / 100:.0f} s 80-dim fbank, streaming chunk={args.chunk} "
                               f"(online_rnnt_decode.py semantics), greedy decode, {args.mode} token return",
                   "streams_per_gpu": B, "frames_per_stream": args.frames, "chunk_frames": args.chunk, "chunks": len(plan),
                   "encoder_frames_per_stream": enc_frames, "symbols_per_encoder_frame": round(syms, 3), "weights": f"seeded synthetic (seed 0, blank_bias {args.blank_bias})",
                   "parallelism": f"streams sharded x{world}, no data-path collective"},
        "kernel_launches_per_step": int(launches),
        "greedy_steps_per_step": int(gsteps),
        "weight_broadcast_ms": round(bcast_ms, 3),
        "roofline": roofline,
        "roofline_isolated": roofline_isolated,
        "roofline_other_sites": other_sites,
        "encoder_only": encoder_only,
        "cpu_baseline": cpu,
        "per_chunk_api": per_chunk_extra,
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
