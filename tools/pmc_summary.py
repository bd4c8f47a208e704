"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs (two separate passes) per kernel.

usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

rocprofv3 reports both counters in KiB.  On gfx950 FETCH_SIZE counts a 128-B request as 64 B for wide (16 B/lane)
coalesced reads (MI355X_MICROARCH.md, HBM section), which is how every large stream of this library is read, so the
corrected HBM-side read bytes are 2 x FETCH_SIZE (except for kernels in FETCH_CORRECTION below); WRITE_SIZE is exact.  Infinity-Cache hits are included in both.
"""
import csv, collections, json, sys

# Read-side correction per kernel (prefix match).  The x2 holds for wide coalesced streams (64 lanes x 16 B contiguous); it does NOT
# hold for gemm_bw's implicit-GEMM operand (16 B per lane in 32-byte runs of different rows): calibrated in round 3 on conv2, whose
# 1.09 GB operand (4 x the Infinity Cache, so every byte is fetched at least once) gives FETCH_SIZE raw = 1091.7 MB per launch with
# plain and with non-temporal loads alike, i.e. raw = the bytes (gpurun_out/conv2nt, profiles/r03 notes).
FETCH_CORRECTION = {"void gemm_bw": 1.0}


def fetch_factor(kernel):
    for pre, f in FETCH_CORRECTION.items():
        if kernel.startswith(pre):
            return f
    return 2.0


def collect(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = collect(fetch, "FETCH_SIZE"), collect(write, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        fv, wv = f.get(k, []), w.get(k, [])
        fa = sum(fv) / len(fv) * 1024 if fv else 0.0
        wa = sum(wv) / len(wv) * 1024 if wv else 0.0
        res[k] = {"launches": max(len(fv), len(wv)), "fetch_size_bytes_raw_avg": round(fa), "write_size_bytes_avg": round(wa),
                  "fetch_correction": fetch_factor(k), "traffic_bytes_per_launch": round(fetch_factor(k) * fa + wa)}
    json.dump({"note": "per-launch averages; traffic = fetch_correction x FETCH_SIZE + WRITE_SIZE (2 x = gfx950 wide-stream correction; 1 x where calibrated, see tools/pmc_summary.py)", "kernels": res}, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{k[:70]:70s} n={v['launches']:5d} traffic/launch {v['traffic_bytes_per_launch'] / 1e6:9.2f} MB")


if __name__ == "__main__":
    main()
