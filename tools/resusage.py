"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output: python tools/resusage.py <stderr file> [filter]"""
import re, subprocess, sys
t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split("\n")[0].strip(" []")
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except FileNotFoundError:
        pass
    if flt not in name:
        continue
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    name = re.sub(r"\(.*", "", name.replace("void cine::", ""))
    print(name[:64].ljust(64), "V", g("VGPRs").rjust(3), "A", g("AGPRs").rjust(3), "spill", g("VGPR Spill").rjust(3),
          "scratch", g(r"ScratchSize \[bytes/lane\]").rjust(4), "occ", g(r"Occupancy \[waves/SIMD\]"), "lds", g(r"LDS Size \[bytes/block\]"))
