#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per pass) for one kernel: per-launch means.

FETCH_SIZE is doubled (MI355X_MICROARCH.md §HBM: on gfx950 it reports exactly half of the bytes of a wide
coalesced streaming read); WRITE_SIZE is exact for 16-byte-per-lane streams.  Units: KB as rocprofv3 reports."""
import collections
import csv
import glob
import json
import sys


def main(root, kernel_substr, out_json=None):
    agg = collections.defaultdict(float)
    disp = collections.defaultdict(set)
    dur = {}
    for f in sorted(glob.glob(root + "/**/*_counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if kernel_substr not in r["Kernel_Name"]:
                continue
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            disp[r["Counter_Name"]].add((f, r["Dispatch_Id"]))
            dur[(f, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if not agg:
        print("no rows for", kernel_substr)
        return
    print(f"## PMC counters for `{kernel_substr}` (per-launch means)\n")
    print("| counter | launches | sum | mean per launch |")
    print("|---|---:|---:|---:|")
    res = {"kernel": kernel_substr}
    for k in sorted(agg):
        n = len(disp[k])
        print(f"| {k} | {n} | {agg[k]:.4g} | {agg[k]/n:.4g} |")
        res[k] = {"launches": n, "sum": agg[k]}
    if "FETCH_SIZE" in agg and "WRITE_SIZE" in agg:
        nf, nw = len(disp["FETCH_SIZE"]), len(disp["WRITE_SIZE"])
        fetch_b = 2.0 * agg["FETCH_SIZE"] * 1024 / nf
        write_b = agg["WRITE_SIZE"] * 1024 / nw
        tdur = sum(dur[d] for d in disp["FETCH_SIZE"]) * 1e-9
        print(f"\nHBM traffic per launch: read {fetch_b/1e9:.3f} GB (FETCH_SIZE x 2, gfx950 correction) + "
              f"written {write_b/1e9:.3f} GB = {(fetch_b+write_b)/1e9:.3f} GB; "
              f"over the launches' {tdur*1e3:.1f} ms -> {(fetch_b*nf+write_b*nf)/tdur/1e12:.2f} TB/s")
        res["hbm_bytes_per_launch"] = fetch_b + write_b
        res["hbm_read_bytes_per_launch"] = fetch_b
        res["hbm_write_bytes_per_launch"] = write_b
    if "TCC_HIT_sum" in agg:
        print(f"L2 hit rate {agg['TCC_HIT_sum']/(agg['TCC_HIT_sum']+agg['TCC_MISS_sum']):.3f}")
    if out_json:
        json.dump(res, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
