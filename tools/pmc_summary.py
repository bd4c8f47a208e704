"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs (two separate passes) per kernel.

usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

rocprofv3 reports both counters in KiB.  On gfx950 FETCH_SIZE counts a 128-B request as 64 B for wide (16 B/lane)
coalesced reads (MI355X_MICROARCH.md, HBM section), which is how every large stream of this library is read, so the
corrected HBM-side read bytes are 2 x FETCH_SIZE; WRITE_SIZE is exact.  Infinity-Cache hits are included in both.
"""
import csv, collections, json, sys


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
                  "traffic_bytes_per_launch": round(2 * fa + wa)}
    json.dump({"note": "per-launch averages; traffic = 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE", "kernels": res}, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{k[:70]:70s} n={v['launches']:5d} traffic/launch {v['traffic_bytes_per_launch'] / 1e6:9.2f} MB")


if __name__ == "__main__":
    main()
