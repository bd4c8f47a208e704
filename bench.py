#!/usr/bin/env python3
"""bench.py — streaming RNN-T greedy decode throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...           (no launcher: the script starts its own N rank processes)

One "step" = one pass of the hot path over one batch: 64 independent 10 s streams per GPU (synthetic 80-dim fbank already
resident in HBM), 16-frame chunks with online_rnnt_decode.py's slicing/offset rules (BASELINE.json configs[1]), chunked
Conformer encoder + greedy RNN-T decode, whole chunk plan in one rnnt_encoder_chunks call, tokens copied back once per batch
(the same tokens as the per-chunk API, whose throughput and per-chunk RTF percentiles are reported beside it).

Timed region: `--in-flight 2` (default) keeps TWO batches in flight per GPU -- two contexts driven by two host threads, K steps in
all -- so one batch's latency-bound greedy decode (the slowest stream's dependent chain, most CUs idle) runs beside the other
batch's encoder; every step is still one whole 64-stream batch and returns the same tokens.  The latency of a single batch with
nothing else on the GPU is reported as `one_batch_in_flight` (and is what `modes` compares).

Numerics: the headline mode is the FASTEST PARITY-GATED mode -- the first of bf16x3, f16x3, fp32 whose greedy tokens equal the
exact-f32 mode's on every stream of the batch (checked live, before the timed region; the f32 tokens are themselves checked
against the CPU oracle on a sample).  Every mode's time, token match and encoder error are in `modes`.  `--numerics X` forces one.

The single JSON line also carries (N = 1): `roofline` of the kernel site with the largest summed launch time in the timed
region (HIP events on the launch stream), the other sites, the decode chain, short legs for BASELINE configs[2] (beam 4),
configs[4] (full-context encoder 32 x 30 s), the joint lattice and the C64 chunking, per-chunk RTF percentiles
(online_rnnt_delay.py:99-131), the first-call time and the CPU baseline.  Streams shard across ranks with no data-path
collective (weak scaling); weights are broadcast once over RCCL.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# launch-site tags of rnnt_profile_begin (include/rnnt_hip.h)
TAGS = {"conv1": 1, "conv2": 2, "embed": 3, "ffn1": 4, "ffn2": 5, "qkv": 6, "attn": 7, "attn_out": 8, "pw1": 9,
        "dwconv": 10, "pw2": 11, "enc_proj": 13, "block_front": 30, "block_back": 31, "ffn": 32, "ffn_qkv": 33, "out_pw1": 34, "ffn_merged": 35,
        "conv1_minor": 36, "conv2_minor": 37, "embed_minor": 38}
TAG_JOINT_OUT = 23   # the lattice kernel's launch site (host_launch.hip.inc)
PEAK_TFLOPS = {"fp32": 157.3, "bf16x3": 2500.0, "f16x3": 2500.0, "bf16": 2500.0}   # MI355X_MICROARCH.md: dense MFMA peak of the operand type
MFMA_PER_ALG = {"fp32": 1, "bf16x3": 3, "f16x3": 3, "bf16": 1}                      # MFMA products issued per algorithmic product
PEAK_HBM_GBS = 8000.0
HBM_SITES = ("attn", "dwconv", "conv1")
PARITY_ORDER = ["bf16x3", "f16x3", "fp32"]


def sub_len(t):
    return ((t - 3) // 2 + 1 - 3) // 2 + 1


def site_kernel(site, mode):
    bf = mode != "fp32"
    gk = "gemm_bf" if bf else "gemm_ns"
    lm = os.environ.get("RNNT_LM", "1") != "0"          # layer-major schedule: one launch per layer over all B*F rows
    if lm:
        if bf and os.environ.get("RNNT_AS", "1") != "0" and site in ("ffn", "ffn_qkv", "ffn_merged", "out_pw1", "qkv", "pw1"):
            return {"out_pw1": "ffn_as chain (linear_out + residual, then LayerNorm + pointwise_conv1 + GLU from the result rows in LDS)",
                    "ffn_qkv": "ffn_as + tail (macaron FFN module, then LayerNorm + linear_q/k/v from its result rows in LDS; K/V rows into the cache)",
                    "ffn_merged": "ffn_as, layer boundary in one launch: depthwise conv + pointwise_conv2 head, FFN + norm_final of layer l, then layer l+1's macaron FFN on the result rows in LDS and LayerNorm + linear_q/k/v from those",
                    "ffn": "ffn_as ([pointwise_conv2 + residual head, x' in LDS,] LayerNorm + w_1 + SiLU + w_2 + half-step residual + norm_final, hidden activation in LDS, M = B*F)",
                    "qkv": "gemm_as x3 (LayerNorm once, linear_q/k/v from one staged operand image, K/V rows into the cache)",
                    "pw1": "gemm_as (LayerNorm + pointwise_conv1 + GLU)"}[site]
        if site.endswith("_minor"):
            return {"conv1_minor": "conv1_relu (the call's tail chunk class)", "conv2_minor": f"{gk} (conv2 implicit GEMM of the tail chunk class)", "embed_minor": f"{gk} (embed Linear of the tail chunk class)"}[site]
        return {"conv1": "conv1_relu_rows", "conv2": ("gemm_bw (conv2 implicit GEMM of all equal-length chunks: 128x256 tiles, weights streamed from L2 in fragment order)" if bf else f"{gk} (conv2 implicit GEMM)"), "embed": f"{gk} (embed Linear)",
                "ffn1": f"{gk} (ffn w_1 + LayerNorm prologue + SiLU, M = B*F)", "ffn2": f"{gk} (ffn w_2 + half-step residual, M = B*F)",
                "qkv": f"{gk} x3 (linear_q/k/v + LayerNorm prologue, K/V rows into the cache)", "attn_out": f"{gk} (linear_out + residual)",
                "pw1": f"{gk} (pointwise_conv1 + LayerNorm prologue + GLU)", "pw2": f"{gk} (pointwise_conv2 + residual)",
                "attn": ("rel_attention_lm_bf (split 16-bit MFMAs for the three products, f32 softmax)" if bf else "rel_attention_lm_mfma (8 chunks per workgroup, v_mfma_f32_16x16x4_f32)"), "dwconv": "dwconv_lm", "enc_proj": "joint.enc_ffn projection (+ after_norm prologue)"}.get(site, site)
    return {"conv1": "conv1_relu", "conv2": "gemm_bf<4,4> (conv2 implicit GEMM)" if bf else "gemm_ns<2,2,32> (conv2 implicit GEMM)",
            "embed": "gemm_bf (embed Linear)" if bf else "gemm_ns / gemm16 (embed Linear)",
            "block_front": "block_front (LN + FFN-macaron + LN + q/k/v, fused)", "block_back": "block_back (out-proj + conv module + FFN + LN, fused)",
            "ffn1": "gemm_ns_tab<1,2,32> (ffn w_1)", "ffn2": "gemm_ns_tab<1,1,64> (ffn w_2)", "qkv": "gemm_ns_tab<1,1,32> (linear_q/k/v)",
            "attn_out": "gemm_ns_tab<1,1,32> (linear_out)", "pw1": "gemm_ns_tab<1,2,32> (pointwise_conv1)", "pw2": "gemm_ns_tab<1,1,32> (pointwise_conv2)",
            "attn": "rel_attention_stream_tab", "dwconv": "dwconv_bn_silu_tab", "enc_proj": "joint.enc_ffn projection (+ after_norm prologue)"}.get(site, site)


def site_flops_bytes(site, B, plan):
    """Algorithmic FLOPs (2*MAC) and bytes of all launches of one site in one step.  Layer-major schedule:
    FLOPs are the same sums; bytes count a layer's weights, K/V rows and conv rows once per layer instead of once per chunk."""
    fl = by = 0.0
    t2 = 0
    if os.environ.get("RNNT_LM", "1") != "0" and site not in ("conv1", "conv2", "embed", "conv1_minor", "conv2_minor", "embed_minor", "block_front", "block_back"):
        F = sum(sub_len(b - a) for a, b in plan)
        M = B * F
        tail = os.environ.get("RNNT_LM_QKV_TAIL", "1") != "0"
        pw2_head = os.environ.get("RNNT_LM_PW2_HEAD", "1") != "0" and os.environ.get("RNNT_AS", "1") != "0"
        merged = pw2_head and tail and os.environ.get("RNNT_LM_FFN_MERGE", "0") != "0"   # (off by default) 11 layer boundaries in one launch each: the plain sites keep one launch
        kv_rows = F                                     # every frame's K/V row is written once and staged by the attention tiles
        att_fl = 0.0
        for i, (a, b) in enumerate(plan):
            tq = sub_len(b - a)
            kv = t2 + tq
            att_fl += 2.0 * B * 4 * tq * kv * 64 * 3
            t2 = kv if i > 0 else 0
        per = {"ffn1": (2.0 * M * 256 * 1024, 4.0 * (M * 256 + 256 * 1024 + M * 1024), 24),
               "ffn2": (2.0 * M * 256 * 1024, 4.0 * (M * 1024 + 256 * 1024 + 2 * M * 256), 24),
               # fused module: x in, x out, both weight matrices; with the pointwise_conv2 head in front (RNNT_LM_PW2_HEAD, default) also its product, dw rows and weights
               "ffn": ((4.0 * M * 256 * 1024 + (2.0 * M * 256 * 256 if pw2_head else 0.0), 4.0 * (2 * M * 256 + 2 * 256 * 1024 + ((M * 256 + 256 * 256) if pw2_head else 0)), 1 if merged else (12 if tail else 24))),
               # layer boundary in one launch (RNNT_LM_FFN_MERGE=1): head + FFN + norm_final of layer l, macaron FFN + q/k/v of layer l + 1
               "ffn_merged": (8.0 * M * 256 * 1024 + 2.0 * M * 256 * 256 + 2.0 * M * 256 * 768, 4.0 * (3 * M * 256 + 4 * 256 * 1024 + 4 * 256 * 256 + 3 * M * 256), 11),
               "ffn_qkv": (4.0 * M * 256 * 1024 + 2.0 * M * 256 * 768, 4.0 * (2 * M * 256 + 2 * 256 * 1024 + 3 * 256 * 256 + 3 * M * 256), 1 if merged else 12),   # macaron FFN + q/k/v from its result rows
               "qkv": (2.0 * M * 256 * 768, 4.0 * (M * 256 + 3 * 256 * 256 + 3 * M * 256), 12),
               "attn": (att_fl, 4.0 * (2 * B * kv_rows * 256 + 2 * M * 256 + (kv_rows + len(plan)) * 256), 12),
               "attn_out": (2.0 * M * 256 * 256, 4.0 * (M * 256 + 256 * 256 + 2 * M * 256), 12),
               "pw1": (2.0 * M * 256 * 512, 4.0 * (M * 256 + 512 * 256 + M * 256), 12),
               "out_pw1": (2.0 * M * 256 * 768, 4.0 * (4 * M * 256 + 3 * 256 * 256), 12),   # linear_out + residual, norm_conv + pointwise_conv1 + GLU: att in, x in/out, GLU rows out
               "dwconv": (2.0 * M * 256 * 31, 4.0 * (B * (30 + F) * 256 + 2 * M * 256), 12),
               "pw2": (2.0 * M * 256 * 256, 4.0 * (M * 256 + 256 * 256 + 2 * M * 256), 12),
               "enc_proj": (2.0 * M * 256 * 256, 4.0 * (2 * M * 256 + 256 * 256), 1)}
        f, y, cnt = per[site]
        return f * cnt, y * cnt
    # layer-major call: the subsampling launches of the call's main chunk class (the length of the first chunk) and of the other
    # classes (the tail chunk) are profiled as different sites (`*_minor`), so a site's average is over launches of one kind
    lm_call = os.environ.get("RNNT_LM", "1") != "0"
    minor_site = site.endswith("_minor")
    base = site[:-6] if minor_site else site
    for i, (a, b) in enumerate(plan):
        if lm_call and base in ("conv1", "conv2", "embed") and ((b - a) != (plan[0][1] - plan[0][0])) != minor_site:
            continue
        tq = sub_len(b - a)
        t1 = (b - a - 3) // 2 + 1
        M = B * tq
        kv = t2 + tq            # keys seen by this chunk
        w_front, w_back = 2 * 256 * 1024 + 3 * 256 * 256, 256 * 256 + 512 * 256 + 31 * 256 + 256 * 256 + 2 * 256 * 1024
        per = {"conv1": (2.0 * B * t1 * 39 * 256 * 9, 4.0 * (B * (b - a) * 80 + B * t1 * 39 * 256), 1),
               "conv2": (2.0 * M * 19 * 256 * 2304, 4.0 * (B * t1 * 39 * 256 + 256 * 2304 + M * 19 * 256), 1),
               "embed": (2.0 * M * 4864 * 256, 4.0 * (M * 4864 + 4864 * 256 + M * 256), 1),
               "ffn1": (2.0 * M * 256 * 1024, 4.0 * (M * 256 + 256 * 1024 + M * 1024), 24),
               "ffn2": (2.0 * M * 256 * 1024, 4.0 * (M * 1024 + 256 * 1024 + 2 * M * 256), 24),
               "qkv": (2.0 * M * 256 * 768, 4.0 * (M * 256 + 3 * 256 * 256 + 3 * M * 256), 12),
               "attn": (2.0 * B * 4 * tq * kv * 64 * 3, 4.0 * (2 * B * kv * 256 + 2 * M * 256 + kv * 256), 12),
               "attn_out": (2.0 * M * 256 * 256, 4.0 * (M * 256 + 256 * 256 + 2 * M * 256), 12),
               "pw1": (2.0 * M * 256 * 512, 4.0 * (M * 256 + 512 * 256 + M * 256), 12),
               "dwconv": (2.0 * M * 256 * 31, 4.0 * (B * (30 + tq) * 256 + 2 * M * 256), 12),
               "pw2": (2.0 * M * 256 * 256, 4.0 * (M * 256 + 256 * 256 + 2 * M * 256), 12),
               "enc_proj": (2.0 * M * 256 * 256, 4.0 * (2 * M * 256 + 256 * 256), 1),
               # fused half blocks: every contraction of the half, weights once per (chunk, layer) pair, x in and out (+ q/k/v or att rows)
               "block_front": (2.0 * M * w_front, 4.0 * (w_front + 2 * M * 256 + 3 * M * 256), 12),
               "block_back": (2.0 * M * w_back, 4.0 * (w_back + 3 * M * 256 + B * 30 * 256 + 2 * M * 256), 12)}
        if base in per:
            f, y, cnt = per[base]
            fl += f * cnt
            by += y * cnt
        t2 = kv if i > 0 else 0    # first chunk's K/V are dropped (required_cache_size = 0)
    return fl, by


def roof(site, mode, fl, by, ms, n_launches, steps):
    """roofline object of one site from its summed launch time `ms` over `steps` steps"""
    if n_launches <= 0 or ms <= 0:
        return None
    s = ms * 1e-3
    lm = os.environ.get("RNNT_LM", "1") != "0"
    if site == "attn" and lm:
        # layer-major attention runs its three contractions on the exact-f32 matrix instruction in every numerics mode
        ach = fl * steps / s / 1e12
        r = {"bound": "mfma", "kernel": site_kernel(site, mode), "achieved": round(ach, 2), "peak": PEAK_TFLOPS["fp32"], "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS["fp32"], 4),
             "traffic": None, "note": "algorithmic FLOPs (3 queries per chunk); the kernel issues 4/3 of them for the padded 4-slot units plus 25 % for the positional rows the units' windows share",
             "achieved_GBs_on_layer_major_bytes": round(by * steps / s / 1e9, 1)}
    elif site in HBM_SITES:
        ach = by * steps / s / 1e9
        r = {"bound": "hbm", "kernel": site_kernel(site, mode), "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None}
    else:
        ach = fl * steps / s / 1e12
        pk = PEAK_TFLOPS[mode]
        r = {"bound": "mfma", "kernel": site_kernel(site, mode), "achieved": round(ach, 2), "peak": pk, "unit": "TFLOP/s", "frac": round(ach / pk, 4), "traffic": None,
             "mfma_products_per_algorithmic_product": MFMA_PER_ALG[mode], "frac_of_mfma_issue_rate": round(ach * MFMA_PER_ALG[mode] / pk, 4)}
    lps = n_launches / steps
    r.update({"algorithmic_flops_per_launch": round(fl / lps), "algorithmic_bytes_per_launch": round(by / lps), "avg_launch_us": round(ms * 1e3 / n_launches, 2),
              "launches_timed": int(n_launches), "ms_per_step": round(ms / steps, 3)})
    return r


def pmc_traffic(kernel_prefixes):
    """HBM-side bytes per launch from the newest committed PMC summary (profiles/*pmc_traffic.json; rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate passes of this command, tools/pmc_summary.py); None if the kernel is not in it."""
    d = os.path.join(ROOT, "profiles")
    cands = sorted(f for f in os.listdir(d) if f.endswith("pmc_traffic.json")) if os.path.isdir(d) else []
    for c in reversed(cands):
        j = json.load(open(os.path.join(d, c)))
        for k, v in j.get("kernels", {}).items():
            if any(k.startswith(p) for p in kernel_prefixes):
                return {"traffic_bytes_per_launch": v["traffic_bytes_per_launch"], "source": f"profiles/{c} (PMC, separate passes; 2 x FETCH_SIZE + WRITE_SIZE)"}
    return None


PMC_PREFIX = {"block_front": ["void block_front"], "block_back": ["void block_back"], "conv2": ["void gemm_bw", "void gemm_bf<2, false, 4, 4", "void gemm_ns<2, 2, 32"],
              "attn": ["void rel_attention_lm_bf", "rel_attention_lm_mfma", "rel_attention_stream_tab"], "ffn2": ["void gemm_ns_tab<1, 1, 64"], "dwconv": ["dwconv_lm", "dwconv_bn_silu_tab"],
              "ffn": ["void ffn_as"], "ffn_qkv": ["void ffn_as"], "ffn_merged": ["void ffn_as"], "out_pw1": ["void ffn_as"], "conv1": ["conv1_relu_rows"]}


def spawn_ranks(args):
    """--gpus N without a launcher: start N rank processes (before anything here touches the GPU) and relay rank 0's line."""
    port = 29500 + os.getpid() % 2000
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        p.wait()
        rc = rc or p.returncode
    sys.stdout.write(out.decode())
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="streams per GPU")
    ap.add_argument("--frames", type=int, default=1000, help="fbank frames per stream (10 ms each)")
    ap.add_argument("--chunk", type=int, default=16, help="fbank frames per chunk (online_rnnt_decode.py semantics)")
    ap.add_argument("--numerics", default="auto", choices=["auto", "fp32", "bf16x3", "f16x3", "bf16"],
                    help="auto = fastest parity-gated mode (tokens equal to the exact-f32 mode's on every stream)")
    ap.add_argument("--site", default="auto", help="launch site timed in the timed region (auto = the one with the largest summed launch time)")
    ap.add_argument("--cpu-streams", type=int, default=8, help="streams of the same workload timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--cpu-threads", type=int, default=8)
    ap.add_argument("--in-flight", type=int, default=2, choices=[1, 2, 3, 4],
                    help="batches in flight per GPU in the timed region: 2 = two contexts driven by two host threads, so one batch's latency-bound decode "
                         "runs beside the other's encoder (every step is still one whole batch; the single-batch latency is reported beside it)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the secondary legs (beam, full context, joint lattice, C64, per-chunk API)")
    ap.add_argument("--legs", default="all", help="secondary legs to run: all, none, or a comma list of chunkapi,c64,beam4,full,joint (tools/profile_bench.sh profiles the joint lattice with --legs joint)")
    ap.add_argument("--blank-bias", type=float, default=12.0,
                    help="bias on the blank logit of the seeded weights: sets the greedy symbol rate (SURVEY.md §8d asks for 0.3-1 symbols per "
                         "encoder frame; 12.0 gives 0.75, the fixtures' 11.0 gives 1.40, 14.0 gives 0.17)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)

    import numpy as np
    import torch
    import ctc_vr_amd.testing as T
    import ctc_vr_amd.dist as D
    from ctc_vr_amd.online_rnnt_model import StreamingBatch
    from ctc_vr_amd.lib import RnntEngine

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"   # every rank on cuda:0 over gloo (one-GPU box); never set by the driver
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
    sd0 = T.make_state_dict(0, blank_bias=args.blank_bias) if rank == 0 else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wblob, wvocab = D.broadcast_packed(sd0, src=0, device=dev)     # the blob stays on this rank's device: contexts take it in one call
    torch.cuda.synchronize()
    bcast_ms = (time.perf_counter() - t0) * 1e3 if world > 1 else 0.0
    sd_np = sd0 if sd0 is not None else D.unpack_state_dict(wblob.cpu().numpy(), wvocab)   # name -> array view for the secondary legs' engines

    B = args.batch
    plan = T.chunk_plan(args.frames, args.chunk)
    enc_frames = sum(sub_len(b - a) for a, b in plan)
    x = torch.from_numpy(T.synth_fbank(B, args.frames, seed=1234 + rank)).to(dev).contiguous()   # inputs resident in HBM
    side = torch.cuda.Stream(device=dev)    # never launch the path on the legacy null stream (implicit cross-stream synchronisation)
    side.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(side)

    def make_sb(mode, chunk=args.chunk, frames=args.frames, max_beam=0, max_tokens=None, batch=B):
        pl = T.chunk_plan(frames, chunk)
        ef = sum(sub_len(b - a) for a, b in pl)
        return StreamingBatch(None, batch, vocab_size=wvocab, max_chunk_frames=max(b - a for a, b in pl), max_cache_frames=ef + 8, max_enc_frames=ef + 8,
                              max_tokens=max_tokens or ef * 10 + 16, device=local_rank, max_beam=max_beam, numerics=mode, packed=(wblob, wvocab))

    # ---- parity gate: one untimed pass per mode -> tokens, first-call time; encoder error vs fp32 ----------------------------
    modes = ["fp32", "bf16x3", "f16x3", "bf16"] if args.numerics == "auto" else sorted({"fp32", args.numerics}, key=["fp32", "bf16x3", "f16x3", "bf16"].index)
    sbs, info = {}, {}
    cs = torch.cuda.current_stream().cuda_stream
    plan_args = ([a for a, _ in plan], [b - a for a, b in plan], [4 * i for i in range(len(plan))])
    for m in modes:
        sb = make_sb(m)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        toks = sb.decode_script(x, args.chunk, pipelined=True)
        torch.cuda.synchronize()
        first_ms = (time.perf_counter() - t0) * 1e3
        sb.reset()
        sb.engine.encoder_chunks(x.data_ptr(), args.frames, plan_args[0], plan_args[1], plan_args[2], plan_args[2], cs, greedy=False)
        enc = sb.engine.enc_frames(cs)
        sbs[m] = sb
        info[m] = {"tokens": toks, "first_call_ms": round(first_ms, 2), "enc": enc}
    ref = info["fp32"]
    for m in modes:
        same = sum(int(a == b) for a, b in zip(info[m]["tokens"], ref["tokens"]))
        info[m]["streams_with_fp32_tokens"] = same
        info[m]["enc_max_abs_err_vs_fp32"] = float(np.abs(info[m]["enc"].astype(np.float64) - ref["enc"]).max())
    for m in modes:
        del info[m]["enc"]
    if args.numerics == "auto":
        choice = next(m for m in PARITY_ORDER if info[m]["streams_with_fp32_tokens"] == B)
    else:
        choice = args.numerics
    if world > 1:   # every rank runs the mode rank 0 chose
        c = torch.tensor([["fp32", "bf16x3", "f16x3", "bf16"].index(choice)], device=dev if not rehearsal else "cpu")
        dist.broadcast(c, src=0)
        choice = ["fp32", "bf16x3", "f16x3", "bf16"][int(c.item())]
        if choice not in sbs:
            sbs[choice] = make_sb(choice)
    sb = sbs[choice]

    def step():
        return sb.decode_script(x, args.chunk, pipelined=True)

    # ---- site survey (untimed): one step per launch site -> which kernel dominates -------------------------------------------
    sites = ["conv1", "conv2", "embed", "conv1_minor", "conv2_minor", "embed_minor", "attn", "enc_proj", "block_front", "block_back", "ffn", "ffn_qkv", "ffn_merged", "out_pw1", "ffn1", "ffn2", "qkv", "attn_out", "pw1", "pw2", "dwconv"]   # sites without launches drop out
    survey = {}
    for _ in range(args.warmup):
        toks = step()
    for sname in sites:
        sb.engine.profile_begin(TAGS[sname])
        step()
        torch.cuda.synchronize()
        ms, nl = sb.engine.profile_end()
        fl, by = site_flops_bytes(sname, B, plan)
        survey[sname] = roof(sname, choice, fl, by, ms, nl, 1)
    live = {k: v for k, v in survey.items() if v}
    site = args.site if args.site != "auto" else max(live, key=lambda k: live[k]["ms_per_step"])

    # ---- one batch in flight (untimed extra): the latency of a batch and the dominant site's launches alone on the device --------
    torch.cuda.synchronize()
    sb.engine.profile_begin(TAGS[site])
    t0 = time.perf_counter()
    n_single = min(args.steps, 5)
    for _ in range(n_single):
        toks = step()
    torch.cuda.synchronize()
    single_s = (time.perf_counter() - t0) / n_single
    iso_ms, iso_launches = sb.engine.profile_end()
    launches, gsteps = sb.engine.counters()

    # ---- timed region: EXACTLY K steps, barrier + synchronize on both sides, max over ranks -------------------------------------
    import threading
    nfl = args.in_flight if args.steps >= 6 else 1              # fewer steps than that cannot fill the two-context pipeline
    ctxs = [sb]
    for _ in range(nfl - 1):
        sbk = make_sb(choice)
        sbk.decode_script(x, args.chunk, pipelined=True)         # first call (allocations, tables) outside the timed region
        ctxs.append(sbk)
    torch.cuda.synchronize()
    share = [args.steps // nfl + (1 if k < args.steps % nfl else 0) for k in range(nfl)]
    results = [None] * nfl

    wstreams = [torch.cuda.Stream(device=dev) for _ in range(nfl)]
    gate = threading.Barrier(sum(1 for k in range(nfl) if share[k] > 0) + 1)

    def worker(idx, sbk, n):
        with torch.cuda.stream(wstreams[idx]):
            gate.wait()                                          # threads and streams exist before the clock starts
            for _ in range(n):
                results[idx] = sbk.decode_script(x, args.chunk, pipelined=True)
        wstreams[idx].synchronize()

    threads = [threading.Thread(target=worker, args=(k, ctxs[k], share[k])) for k in range(nfl) if share[k] > 0]
    for th in threads:
        th.start()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    sb.engine.profile_begin(TAGS[site])
    t0 = time.perf_counter()
    gate.wait()
    for th in threads:
        th.join()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    site_ms, site_launches = sb.engine.profile_end()
    toks = results[0]
    tokens_same = all(r is None or r == toks for r in results)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if not rehearsal else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_frames = world * B * args.frames
    value = total_frames * args.steps / elapsed
    syms = np.array([len(t) for t in toks], dtype=np.float64) / enc_frames
    fl, by = site_flops_bytes(site, B, plan)
    roofline = roof(site, choice, fl, by, site_ms, site_launches, share[0])
    if roofline:
        roofline["site"] = site
        roofline["share_of_step"] = round(site_ms * 1e-3 / share[0] / single_s, 4)
        roofline["timed_under"] = f"{nfl} batch(es) in flight (the launches of one of the contexts, HIP events on its stream)"
        iso = roof(site, choice, fl, by, iso_ms, iso_launches, n_single)
        if iso:
            roofline["alone_on_the_device"] = {k: iso[k] for k in ("achieved", "frac", "avg_launch_us", "ms_per_step") if k in iso}
        roofline["selection"] = "site with the largest summed launch time in an untimed one-step survey of every site; timed here with HIP events on its launch stream"
        tr = pmc_traffic(PMC_PREFIX.get(site, [])) if (B, args.frames, args.chunk) == (64, 1000, 16) else None
        if tr:
            roofline["traffic"] = tr["traffic_bytes_per_launch"]
            roofline["traffic_source"] = tr["source"]
            if site in ("ffn", "ffn_qkv", "ffn_merged", "out_pw1"):   # one kernel name for three launch kinds: the profile cannot tell them apart
                roofline["traffic_note"] = "mean over ALL ffn_as launches of the step (FFN + q/k/v, linear_out + pointwise_conv1, FFN + norm_final share one kernel name), not this site alone"
    other_sites = {k: v for k, v in live.items() if k != site}

    out = {
        "metric": "audio-frames/sec, streaming RNN-T greedy decode",
        "value": round(value, 1), "unit": "audio-frames/s", "rtfx": round(value / 100.0, 1), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": choice, "data": "synthetic",
        "config": {"workload": f"configs[1]: batch={B}/GPU synthetic {args.frames / 100:.0f} s 80-dim fbank, streaming chunk={args.chunk} "
                               f"(online_rnnt_decode.py semantics), greedy decode, whole-utterance call (tokens identical to the per-chunk API)",
                   "streams_per_gpu": B, "frames_per_stream": args.frames, "chunk_frames": args.chunk, "chunks": len(plan), "encoder_frames_per_stream": enc_frames,
                   "symbols_per_encoder_frame": {"mean": round(float(syms.mean()), 3), "slowest_stream": round(float(syms.max()), 3)},
                   "weights": f"seeded synthetic (seed 0, blank_bias {args.blank_bias})", "parallelism": f"streams sharded x{world}, no data-path collective",
                   "numerics": {"headline": choice, "selection": "fastest parity-gated mode: first of bf16x3, f16x3, fp32 whose greedy tokens equal the exact-f32 mode's on every stream" if args.numerics == "auto" else "forced by --numerics",
                                "storage": "fp32", "accumulate": "fp32", "decoder_arithmetic": "fp32 (VALU)"}},
        "batches_in_flight": nfl,
        "one_batch_in_flight": {"ms_per_step": round(single_s * 1e3, 3), "value": round(world * B * args.frames / single_s, 1), "unit": "audio-frames/s",
                                "note": "latency of one 64-stream batch with nothing else on the GPU (this rank); the headline keeps two batches in flight on two contexts"},
        "tokens_identical_across_contexts": bool(tokens_same),
        "kernel_launches_per_step": int(launches), "greedy_evaluations_per_step": int(gsteps), "weight_broadcast_ms": round(bcast_ms, 3),
        "roofline": roofline, "roofline_other_sites": other_sites,
    }
    if world > 1:
        print(json.dumps(out))
        dist.destroy_process_group()
        return

    # ---- every numerics mode: time, token parity, encoder error (short runs outside the timed region) ----------------------------
    mode_objs = {}
    for m in modes:
        o = dict(info[m])
        del o["tokens"]
        if m == choice:
            o["ms_per_step"] = round(single_s * 1e3, 3)
        else:
            s2 = sbs[m]
            for _ in range(2):
                s2.decode_script(x, args.chunk, pipelined=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = min(args.steps, 5)
            for _ in range(n):
                s2.decode_script(x, args.chunk, pipelined=True)
            torch.cuda.synchronize()
            o["ms_per_step"] = round((time.perf_counter() - t0) / n * 1e3, 3)
        o["audio_frames_per_s"] = round(B * args.frames / (o["ms_per_step"] * 1e-3), 1)
        o["parity_gated"] = m in PARITY_ORDER
        o["tokens_equal_fp32_all_streams"] = o["streams_with_fp32_tokens"] == B
        mode_objs[m] = o
    for o in mode_objs.values():
        o["note"] = "one batch in flight"
    out["modes"] = mode_objs

    # ---- encoder alone / decode chain alone (headline mode) ---------------------------------------------------------------------------
    def enc_only():
        sb.reset()
        sb.engine.encoder_chunks(x.data_ptr(), args.frames, plan_args[0], plan_args[1], plan_args[2], plan_args[2], cs, greedy=False)
    enc_only(); sb.engine.greedy_decode(cs)
    torch.cuda.synchronize()
    te = td = 0.0
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        enc_only()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        sb.engine.greedy_decode(cs)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        te += (t1 - t0) / 3; td += (t2 - t1) / 3
    _, ev = sb.engine.counters()
    enc_flops = sum(site_flops_bytes(k, B, plan)[0] for k in ("conv1", "conv2", "embed", "ffn1", "ffn2", "qkv", "attn", "attn_out", "pw1", "dwconv", "pw2"))
    out["encoder_only"] = {"ms": round(te * 1e3, 3), "algorithmic_tflop": round(enc_flops / 1e12, 4), "achieved": round(enc_flops / te / 1e12, 2), "peak": PEAK_TFLOPS[choice],
                           "unit": "TFLOP/s", "frac": round(enc_flops / te / 1e12 / PEAK_TFLOPS[choice], 4), "note": "whole chunked encoder of the batch, no decoder; wall clock around rnnt_encoder_chunks"}
    nsym = np.array([len(t) for t in toks])
    multi = B * 4 <= 256 and os.environ.get("RNNT_DEC_MULTI", "1") != "0"
    out["decode_chain"] = {"kernel": "greedy_multi<4> (4 CUs per stream, W_hh / W_c slices in registers, W_out slice in LDS; 2 tagged-word exchanges per symbol, 1 per run of blank frames)" if multi
                                     else "greedy_stream<4> (1 CU per stream, 1.7 MB of weight rows streamed from L2 per symbol)",
                           "ms_alone": round(td * 1e3, 3), "schedule": "after the encoder stages on the caller's stream" if multi else "resident beside the encoder stages",
                           "evaluations_all_streams": int(ev), "symbols_per_stream": {"mean": round(float(nsym.mean()), 1), "max": int(nsym.max())},
                           "us_per_symbol_slowest_stream": round(td * 1e6 / max(int(nsym.max()), 1), 2),
                           "weight_bytes_streamed_per_symbol": 0 if multi else 4 * (1024 * 256 + 256 * 256 + 412 * 256),
                           "exchange_bytes_per_symbol_per_part": 8 * (320 + 8) if multi else 0,
                           "bound": "dependent chain of the slowest stream (latency), not bandwidth"}
    # ---- top-2 logit margins of every greedy decision of the batch (a token flip must be attributable) ----------------------------
    # Teacher-forced replay in float64 torch on the GPU (ctc_vr_amd.testing.greedy_margins) on the headline mode's encoder frames.
    # The minimum-margin stream joins the CPU-baseline sample below.
    def margin_report():
        sb.reset()
        sb.engine.encoder_chunks(x.data_ptr(), args.frames, plan_args[0], plan_args[1], plan_args[2], plan_args[2], cs, greedy=False)
        mm, replay_ok = T.greedy_margins(sd_np, sb.engine.enc_frames(cs), toks, T.BLANK, dev)
        order = np.argsort(mm)
        return {"definition": "top-1 minus top-2 joint logit at every cell the greedy walk visits, float64 torch replay on the headline mode's encoder frames (teacher-forced predictor)",
                "min": float(mm.min()), "min_stream": int(order[0]), "five_smallest": [{"stream": int(b), "margin": float(mm[b])} for b in order[:5]],
                "median_over_streams": float(np.median(mm)), "replay_reproduces_gpu_tokens_streams": int(replay_ok.sum()), "streams": B}
    margins = margin_report()
    out["top2_margins"] = margins

    ALL_LEGS = ["chunkapi", "c64", "beam4", "full", "joint"]
    want = [] if (args.no_legs or args.legs == "none") else (ALL_LEGS if args.legs == "all" else [w for w in args.legs.split(",") if w])
    for w in want:
        if w not in ALL_LEGS:
            raise SystemExit(f"--legs: unknown leg {w!r} (choose from {ALL_LEGS})")
    if not want:
        print(json.dumps(out))
        return

    # ---- per-chunk API (the reference's actual call pattern): throughput + per-chunk RTF percentiles (online_rnnt_delay.py:55-59,99-131) ----
    if "chunkapi" in want:
        sb.decode_script(x, args.chunk, per_chunk_decode=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        toks_pc = sb.decode_script(x, args.chunk, per_chunk_decode=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rtfs = []
        sb.reset()
        for (a, b) in plan:
            c = x[:, a:b, :].contiguous()
            torch.cuda.synchronize()
            t1 = time.time()
            sb.process_chunk(c, decode=True)
            sb.engine.token_counts(cs)                      # tokens of the chunk reach the host (synchronises, like the reference's .item())
            rtfs.append((time.time() - t1) / ((b - a) * 0.01))
        r = np.asarray(rtfs)
        out["per_chunk_api"] = {"value": round(B * args.frames / dt, 1), "ms_per_step": round(dt * 1e3, 3), "tokens_equal_whole_utterance_call": toks_pc == toks,
                                "rtf_per_chunk": {"mean": float(r.mean()), "p50": float(np.percentile(r, 50)), "p80": float(np.percentile(r, 80)), "p90": float(np.percentile(r, 90)),
                                                  "p95": float(np.percentile(r, 95)), "max": float(r.max()), "chunks": int(r.size),
                                                  "definition": f"wall time of one chunk call for all {B} streams / chunk audio duration (online_rnnt_delay.py:55-59)"},
                                "note": "tokens copied to the host after every chunk (online_rnnt_decode.py loop, process_single_chunk API)"}

    # ---- C64 chunking (SURVEY.md §8d secondary) -------------------------------------------------------------------------------------------------
    legs = {}
    if "c64" in want:
        sb64 = make_sb(choice, chunk=64)
        sb64.decode_script(x, 64, pipelined=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2):
            t64 = sb64.decode_script(x, 64, pipelined=True)
        torch.cuda.synchronize()
        d64 = (time.perf_counter() - t0) / 2
        legs["c64"] = {"workload": f"configs[1] with streaming_inference chunking: 64-frame chunks, batch={B}", "ms_per_step": round(d64 * 1e3, 3), "value": round(B * args.frames / d64, 1),
                       "unit": "audio-frames/s", "dtype": choice, "symbols_per_encoder_frame": round(float(np.mean([len(t) for t in t64])) / sum(sub_len(b - a) for a, b in T.chunk_plan(args.frames, 64)), 3)}
        del sb64

    # ---- configs[2]: beam 4 ------------------------------------------------------------------------------------------------------------------
    if "beam4" in want:
        sbb = make_sb(choice, max_beam=4, max_tokens=16)
        sbb.beam_script(x, args.chunk, 4, pipelined=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2):
            beams = sbb.beam_script(x, args.chunk, 4, pipelined=True)
        torch.cuda.synchronize()
        db = (time.perf_counter() - t0) / 2
        legs["beam4"] = {"workload": f"configs[2]: batch={B} beam_search beam=4, streaming chunk={args.chunk}; one encoder call + one rnnt_beam_advance", "ms_per_step": round(db * 1e3, 2),
                         "value": round(B * args.frames / db, 1), "unit": "audio-frames/s", "dtype": choice, "best_tokens_stream0": len(max(beams[0], key=lambda h: h.log_prob).tokens),
                         "roofline": {"bound": "latency", "note": "per frame one beam_chain launch whose length is the longest extension chain (<= 10 evaluations x ~27 us), "
                                      "a copy of the candidate tables, the C++ merge and a state gather; no bandwidth or MFMA roofline applies"}}
        del sbb

    # ---- configs[4]: full-context encoder 32 x 30 s -----------------------------------------------------------------------------------------------
    if "full" in want:
        Bf, Tn = 32, 3000
        eng = RnntEngine(max_streams=Bf, max_chunk_frames=Tn, max_cache_frames=760, max_enc_frames=8, vocab_size=T.VOCAB, blank_id=T.BLANK, device=local_rank)
        eng.load_state_dict(sd_np, numerics=choice)
        xf = torch.from_numpy(T.synth_fbank(Bf, Tn, seed=1234)).to(dev).contiguous()
        tq = sub_len(Tn)
        of = torch.empty(Bf, tq, 256, device=dev)
        lens = np.full(Bf, Tn, np.int32)
        eng.encoder_full(xf.data_ptr(), lens, Bf, Tn, of.data_ptr(), cs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2):
            eng.encoder_full(xf.data_ptr(), lens, Bf, Tn, of.data_ptr(), cs)
        torch.cuda.synchronize()
        df = (time.perf_counter() - t0) / 2
        flops = Bf * 2.0 * (tq * (11.2e6 + 1.25e6 + 0.18e6 + 18.183168e6 + 9216.0 * tq))
        legs["full_context"] = {"workload": f"configs[4]: full-context encoder, batch={Bf} x 30 s (decoding_chunk_size=-1), {tq} frames per utterance", "ms_per_step": round(df * 1e3, 2),
                                "value": round(Bf * Tn / df, 1), "unit": "audio-frames/s", "dtype": choice,
                                "roofline": {"bound": "mfma", "kernel": "whole encoder pass = the layer-major schedule with one chunk of T frames (gemm_bw conv2, ffn_as, gemm_as, " + ("rel_attention_lm_mfma" if choice == "fp32" else "rel_attention_lm_bf") + " over 749 keys, dwconv_lm)", "achieved": round(flops / df / 1e12, 2),
                                             "peak": PEAK_TFLOPS[choice], "unit": "TFLOP/s", "frac": round(flops / df / 1e12 / PEAK_TFLOPS[choice], 4), "traffic": None}}
        del eng, xf, of

    # ---- joint lattice B64 x T249 x U28: the HBM-roofline kernel of north_star ---------------------------------------------------
    if "joint" in want:
        Bj, Tj, U, V = 64, 249, 28, T.VOCAB
        g = torch.Generator(device="cpu").manual_seed(5)
        enc = torch.randn(Bj, Tj, 256, generator=g).to(dev)
        prd = (torch.randn(Bj, U, 256, generator=g) * 0.5).to(dev)
        lat = torch.empty(Bj, Tj, U, V, device=dev)
        cells = Bj * Tj * U
        byts = 4.0 * (Bj * Tj * 256 + Bj * U * 256) + 4.0 * (2 * 256 * 256 + 256 * V + 2 * 256 + V) + 4.0 * cells * V

        def joint_times(mode_name):
            """(wall s of rnnt_joint, HIP-event s of the lattice kernel alone, its launches) per form: 0 logits, 1 log-softmax."""
            e_ = RnntEngine(max_streams=Bj, max_chunk_frames=16, max_cache_frames=8, max_enc_frames=Tj + 64, vocab_size=V, blank_id=T.BLANK, device=local_rank)
            e_.load_state_dict(sd_np, numerics=mode_name)
            res = {}
            for jm in (0, 1):
                for _ in range(2):
                    e_.joint(enc.data_ptr(), prd.data_ptr(), Bj, Tj, U, jm, lat.data_ptr(), cs)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5):
                    e_.joint(enc.data_ptr(), prd.data_ptr(), Bj, Tj, U, jm, lat.data_ptr(), cs)
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) / 5
                e_.profile_begin(TAG_JOINT_OUT)          # HIP events around the lattice kernel's launch on its stream
                for _ in range(5):
                    e_.joint(enc.data_ptr(), prd.data_ptr(), Bj, Tj, U, jm, lat.data_ptr(), cs)
                kms, kn = e_.profile_end()
                res[jm] = (wall, kms * 1e-3 / max(kn, 1), int(kn))
            return e_, res

        eng, res = joint_times(choice)
        lat_ref = torch.empty(Bj, Tj, U, V, device=dev)
        eng.joint(enc.data_ptr(), prd.data_ptr(), Bj, Tj, U, 0, lat_ref.data_ptr(), cs)
        # the ceiling of any kernel that writes this lattice: a plain fill of the same bytes (torch fill_ on the caller's stream)
        lat.fill_(0.0); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            lat.fill_(1.0)
        torch.cuda.synchronize()
        t_fill = (time.perf_counter() - t0) / 5
        kern = "joint_lattice_rows (rnnt_joint.hip.h)" if choice != "fp32" else "gemm_ns (tanh-add prologue) + log_softmax_rows"
        pt = pmc_traffic(["void joint_lattice_rows"]) if choice != "fp32" else None
        k1 = res[1][1]
        legs["joint_lattice"] = {"workload": f"joint lattice B={Bj} T={Tj} U={U} V={V} (SURVEY.md §8d): enc_ffn, pred_ffn (two small GEMMs), then ONE lattice kernel (tanh-add operand formation, projection, log-softmax on the accumulators)",
                                 "dtype": choice, "logits_ms": round(res[0][0] * 1e3, 3), "log_softmax_ms": round(res[1][0] * 1e3, 3), "value": round(cells / res[1][0], 1), "unit": "lattice-cells/s (log-softmax form, whole rnnt_joint call)",
                                 "roofline": {"bound": "hbm", "kernel": kern, "achieved": round(byts / k1 / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                              "frac": round(byts / k1 / 1e9 / PEAK_HBM_GBS, 4), "traffic": pt["traffic_bytes_per_launch"] if pt else None, "traffic_source": pt["source"] if pt else None,
                                              "algorithmic_bytes": round(byts), "avg_launch_us": round(k1 * 1e6, 1), "launches_timed": res[1][2],
                                              "timing": "HIP events around the lattice kernel on its launch stream (log-softmax form)",
                                              "logits_form_avg_launch_us": round(res[0][1] * 1e6, 1), "logits_form_achieved_GBs": round(byts / res[0][1] / 1e9, 1),
                                              "whole_call_frac": round(byts / res[1][0] / 1e9 / PEAK_HBM_GBS, 4),
                                              "fill_of_the_same_bytes_us": round(t_fill * 1e6, 1), "fill_frac_of_peak": round(byts / t_fill / 1e9 / PEAK_HBM_GBS, 4),
                                              "note": "the lattice is write-bound: a plain fill of its 735 MB is the ceiling of this kernel (fill_frac_of_peak)"}}
        # the same lattice in the single-product bf16 perf mode (what north_star's ">= 60 % of HBM" presumes): time + max logit error vs the headline mode
        engb, resb = joint_times("bf16")
        engb.joint(enc.data_ptr(), prd.data_ptr(), Bj, Tj, U, 0, lat.data_ptr(), cs)
        torch.cuda.synchronize()
        legs["joint_lattice"]["bf16_perf_mode"] = {"logits_ms": round(resb[0][0] * 1e3, 3), "log_softmax_ms": round(resb[1][0] * 1e3, 3),
                                                   "kernel_avg_launch_us": round(resb[1][1] * 1e6, 1), "achieved_GBs": round(byts / resb[1][1] / 1e9, 1),
                                                   "frac_of_hbm": round(byts / resb[1][1] / 1e9 / PEAK_HBM_GBS, 4), "whole_call_frac": round(byts / resb[1][0] / 1e9 / PEAK_HBM_GBS, 4),
                                                   "max_abs_logit_err_vs_headline_mode": float((lat - lat_ref).abs().max().item()), "parity_gated": False}
        del eng, engb, lat, lat_ref
    out["legs"] = legs

    # ---- CPU baseline: the oracle on this host, B=1 streams serially (baseline only) -----------------------------------------------------------------
    if not args.no_cpu:
        from oracle import rnnt_oracle as O   # CPU baseline leg ONLY (checker / baseline, never the product path)
        torch.set_num_threads(args.cpu_threads)
        sd_t = O.to_torch_sd(sd_np)
        # the first streams of the batch plus the one with the smallest top-2 margin (the stream a rounding difference would flip first)
        sample = list(range(min(args.cpu_streams, B)))
        if margins["min_stream"] not in sample:
            sample[-1] = margins["min_stream"]
        xc = x[sample].cpu()
        t0 = time.perf_counter()
        ok, bad = True, []
        with torch.no_grad():
            for j, b in enumerate(sample):
                want, _, _ = O.decode_script_greedy(sd_t, xc[j:j + 1], args.chunk)
                if want != toks[b]:
                    ok = False; bad.append(b)
        ct = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(xc.size(0) * args.frames / ct, 1), "unit": "audio-frames/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{xc.size(0)} of the {B} streams (the first ones + the minimum-margin stream) x {args.frames} frames, B=1 serial (the reference cannot batch), oracle/rnnt_oracle.py torch-CPU float32",
                               "streams_sampled": sample, "tokens_match_gpu": bool(ok), "streams_with_different_tokens": bad,
                               "min_margin_stream_in_sample": margins["min_stream"], "its_min_top2_margin": margins["min"]}
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))


if __name__ == "__main__":
    main()
