#!/usr/bin/env python3
"""Average the per-dispatch PMC counters of rocprofv3 csv passes: pmc_summary.py <dir prefix> <pass> [<pass> ...]"""
import csv, glob, re, sys
from collections import defaultdict


def main(prefix, passes):
    for ps in passes:
        files = glob.glob(f"{prefix}{ps}/**/*counter_collection.csv", recursive=True)
        print(f"== pass {ps}")
        agg = defaultdict(lambda: defaultdict(float))
        disp = defaultdict(set)
        dur = defaultdict(float)
        order = []
        for f in files:
            for r in csv.DictReader(open(f)):
                k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
                if k not in agg:
                    order.append(k)
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                d = r["Dispatch_Id"]
                if d not in disp[k]:
                    disp[k].add(d)
                    if "Start_Timestamp" in r and "End_Timestamp" in r:
                        dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        for k in order:
            if not k.startswith("sinddm"):
                continue
            n = len(disp[k])
            cs = " ".join(f"{c}={v / n:.0f}" for c, v in sorted(agg[k].items()))
            print(f"{k[:50]:50s} n={n:3d} avg_ns={dur[k] / n:12.0f} {cs}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
