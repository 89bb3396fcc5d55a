"""MFMA utilisation per kernel from one rocprofv3 --pmc pass (tools/collect_profiles.sh).
  python tools/pmc_mfma.py <counter_collection.csv> <forwards in the run> [out.json] [commit]
SQ_VALU_MFMA_BUSY_CYCLES sums the cycles the MFMA pipe of every SIMD is busy (32 per v_mfma_f32_16x16x4_f32).  GRBM_GUI_ACTIVE
comes back summed over the 8 XCDs, so a dispatch has GUI / 8 x 256 CUs x 4 SIMDs pipe-cycles and mfma_util = busy / (GUI / 8 * 1024).
gui_cycles_per_ns = GUI / 8 / the dispatch's duration: for dispatches of >= 20 us it reads as the shader clock (2.3 - 2.5 GHz); for short ones the
counter window is wider than the dispatch (it includes the launch ramp), the ratio exceeds any clock the part has and is NOT a clock -- it is
printed for the long dispatches only."""
import csv, re, collections, json, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
def _forwards(a):
    """an integer, or the log of the bench run (its JSON line carries forwards_run)"""
    try:
        return int(a)
    except ValueError:
        for ln in reversed(open(a).read().splitlines()):
            if ln.startswith("{"):
                return int(json.loads(ln)["forwards_run"])
        raise
fw = _forwards(sys.argv[2])
per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
dur = collections.defaultdict(float)
for r in rows:
    k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "")
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in cnt[k]:
        dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    cnt[k].add(r["Dispatch_Id"])
out = {}
for k, c in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
    busy, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
    if busy == 0 and "conv" not in k and "unet" not in k:
        continue
    n = len(cnt[k])
    out[k] = {"dispatches": n, "mfma_busy_cycles": busy, "gui_active_cycles": gui,
              "mfma_util": round(busy / (gui / 8.0 * 1024.0), 4) if gui else None,
              "duration_us_per_dispatch": round(dur[k] / n / 1e3, 2),
              "gui_cycles_per_ns": round(gui / 8.0 / dur[k], 3) if dur[k] and dur[k] / n >= 20e3 else None,
              "valu_insts_per_mfma_mop_x512": round(c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0) / 512.0, 1), 3),
              "insts_valu": c.get("SQ_INSTS_VALU", 0), "mfma_mops_f32": c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0),
              "active_inst_valu_cycles": c.get("SQ_ACTIVE_INST_VALU", 0), "busy_cu_cycles": c.get("SQ_BUSY_CU_CYCLES", 0), "waves": c.get("SQ_WAVES", 0)}
    print(f"{k[:70]:70s} n={n:4d} mfma_util={out[k]['mfma_util']} {out[k]['duration_us_per_dispatch']:8.1f} us/dispatch  gui_cycles_per_ns={out[k]['gui_cycles_per_ns']}")
def is_conv3(k):
    return ("conv_plane_kernel" in k and "tconv" not in k) or "conv_wide_kernel" in k or "conv_coarse_kernel" in k or ("conv_mfma_kernel" in k and re.search(r", (9|27)(, \d)?>", k))
tot_busy = sum(v["mfma_busy_cycles"] for k, v in out.items() if is_conv3(k))
tot_gui = sum(v["gui_active_cycles"] for k, v in out.items() if is_conv3(k))
summary = {"conv3x3_family_mfma_util": round(tot_busy / (tot_gui / 8.0 * 1024.0), 4) if tot_gui else None}
print(summary)
if len(sys.argv) > 3:
    json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU "
                         "SQ_WAVES GRBM_GUI_ACTIVE (one pass, eager, one slice in flight); mfma_util = busy / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)",
               "commit": sys.argv[4] if len(sys.argv) > 4 else None, "forwards": fw, "summary": summary, "kernels": out},
              open(sys.argv[3], "w"), indent=1)
