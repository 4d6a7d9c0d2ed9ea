#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats CSV directory into a small markdown table (profiles/)."""
import csv
import glob
import sys


def main(d, title, samples):
    f = glob.glob(d + "/**/*_kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"## {title}\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % of GPU time |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for r in rows:
        if float(r["TotalDurationNs"]) < 1e-4 * tot:
            continue
        print(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | "
              f"{float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | "
              f"{float(r['Percentage']):.1f} |")
    print(f"\nsum of kernel time {tot/1e6:.1f} ms" + (f" = {tot/1e3/samples:.1f} us per posterior sample" if samples else ""))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0)
