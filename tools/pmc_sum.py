"""Sum PMC counters per kernel family over a whole run: python tools/pmc_sum.py <counter_collection.csv>"""
import csv, re, collections, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    fam = ("conv3" if "conv_mfma" in k and re.search(r", 9(, \d)?>", k) else "conv1/tconv" if "conv_mfma" in k else "fft" if "200_" in k or "_pass_" in k
           else "other")
    agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); dur[fam] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
for fam, c in agg.items():
    print(f"== {fam}: {dur[fam]:.0f} us total")
    for k, v in sorted(c.items()):
        print(f"   {k:30s} {v:16.0f}")
