"""Summarise a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA (+ optionally SQ_WAVE_CYCLES) run per kernel.

usage: python tools/mfma_summary.py <counter_collection.csv> <out.json>

SQ_VALU_MFMA_BUSY_CYCLES is exact SIMD-cycles of matrix work summed over the 1024 SIMDs (16 per v_mfma_f32_16x16x32_bf16/f16, 32 per
v_mfma_f32_16x16x4_f32: checked on the f32 conv2 GEMM, 24.5 GFLOP / 2048 FLOP x 32 cycles x 77 launches = 2.95e10 = the counter).
SQ_BUSY_CYCLES comes back summed over the 32 shader engines, so the SIMD-cycles a kernel had available are 32 x SQ_BUSY_CYCLES
(1024 SIMDs / 32 engines; the same check gives 270 us per launch for that kernel, its measured duration) and
    mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES).
Under --pmc dispatches are serialised, so the fractions are per kernel ALONE on the device.
"""
import collections
import csv
import json
import sys


def main():
    path, out = sys.argv[1:3]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        n[r["Kernel_Name"]].add(r["Dispatch_Id"])
    res = {}
    for k, c in agg.items():
        busy = c.get("SQ_BUSY_CYCLES", 0.0)
        res[k] = {"launches": len(n[k]), "SQ_VALU_MFMA_BUSY_CYCLES": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), "SQ_BUSY_CYCLES": busy,
                  "SQ_INSTS_MFMA": c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", c.get("SQ_INSTS_MFMA", 0.0)),
                  "mfma_busy_frac": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (32.0 * busy), 4) if busy else None}
        for extra in ("SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY"):
            if extra in c:
                res[k][extra] = c[extra]
    json.dump({"note": "sums over all launches of a kernel; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES), see tools/mfma_summary.py; dispatches serialised by --pmc",
               "kernels": res}, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["SQ_BUSY_CYCLES"])[:14]:
        print(f"{k[:72]:72s} n={v['launches']:5d} mfma_busy {v['mfma_busy_frac']}")


if __name__ == "__main__":
    main()
