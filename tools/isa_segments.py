"""Static instruction mix of one kernel, split at s_barrier: python tools/isa_segments.py file.s mangled_name_substr"""
import sys, collections
s = open(sys.argv[1]).read()
key = sys.argv[2]
i = s.index(key); i = s.index(":", i); j = s.index(".Lfunc_end", i)
seg, segs = collections.Counter(), []
for l in s[i:j].split("\n"):
    l = l.strip()
    if not l or l.startswith((".", ";")) or l.endswith(":"):
        continue
    op = l.split()[0]
    if op == "s_barrier":
        segs.append(seg); seg = collections.Counter(); continue
    cls = ("mfma" if "mfma" in op else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
           "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other")
    seg[cls] += 1
    if op.startswith("scratch_"): seg["scratch"] += 1
    if "lane" in op: seg["lane"] += 1
segs.append(seg)
for k, c in enumerate(segs):
    print(k, dict(c))
