"""Summarise a rocprofv3 --pmc SQ_LDS_* / SQ_INSTS_VALU counter_collection CSV per kernel (sums over all launches).

usage: python tools/lds_summary.py <counter_collection.csv> <out.json>
lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (conflict cycles per active LDS cycle).
"""
import collections
import csv
import json
import sys

path, out = sys.argv[1:3]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for r in csv.DictReader(open(path)):
    agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    n[r["Kernel_Name"]].add(r["Dispatch_Id"])
res = {}
for k, c in agg.items():
    d = {"launches": len(n[k])}
    d.update({a: int(b) for a, b in sorted(c.items())})
    act = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    d["lds_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / act, 4) if act else None
    res[k] = d
json.dump({"note": "sums over all launches of a kernel (tools/profile_bench.sh, separate --pmc pass)", "kernels": res}, open(out, "w"), indent=1)
