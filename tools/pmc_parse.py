"""Average the per-dispatch PMC values of kernels matching a substring: python tools/pmc_parse.py <counter_collection.csv> <substr>"""
import csv, collections, sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:32s} {sum(v) / len(v):16.0f}   (n={len(v)})")
