"""HBM-side traffic per kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB per dispatch).
  python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <forwards in the run> [out.json] [commit] [config]
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-B requests of wide (16 B/lane) streaming reads at 64 B,
so the read side is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  Calibrated here on conv1x1_stream_kernel, whose
algorithmic bytes are known (85.2 MB read, 10.6 MB written per launch at cfg 2)."""
import csv, re, collections, json, sys

def fam(k):
    if "tconv_plane_kernel" in k or "s2d_gemm_plane_kernel" in k: return "tconv_conv1x1_mfma"      # the lean kernels of csrc/conv_plane.hip
    if "conv_plane_kernel" in k or "conv_wide_kernel" in k or "conv_coarse_kernel" in k: return "conv3x3_mfma"
    if "conv_mfma" in k and re.search(r", (9|27)(, \d)?>", k): return "conv3x3_mfma"
    if "conv_mfma" in k: return "tconv_conv1x1_mfma"
    if "conv1x1_stream" in k: return "conv1x1_stream"
    if "col200" in k or "col_pass" in k or "imgdc" in k: return "fft_col_pass"
    if "unet_bottom" in k: return "conv3x3_mfma"
    if "row200" in k or "row_pass" in k: return "fft_row_pass"
    if "cine::" in k: return "pack_unpack_misc"
    return "other"

def load(path, name):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name: continue
        f = fam(r["Kernel_Name"]); tot[f] += float(r["Counter_Value"]); cnt[f] += 1
    return tot, cnt

fetch, nf = load(sys.argv[1], "FETCH_SIZE")
write, nw = load(sys.argv[2], "WRITE_SIZE")
def _forwards(a):
    """an integer, or the log of the bench run (its JSON line carries forwards_run)"""
    try:
        return int(a)
    except ValueError:
        for ln in reversed(open(a).read().splitlines()):
            if ln.startswith("{"):
                return int(json.loads(ln)["forwards_run"])
        raise
fw = _forwards(sys.argv[3])
out = {}
for f in sorted(set(fetch) | set(write)):
    rd = 2.0 * fetch[f] * 1024 / fw; wr = write[f] * 1024 / fw
    out[f] = {"launches_per_slice": nf[f] / fw, "read_MB_per_slice": round(rd / 1e6, 1), "write_MB_per_slice": round(wr / 1e6, 1),
              "hbm_MB_per_slice": round((rd + wr) / 1e6, 1), "hbm_MB_per_launch": round((rd + wr) / 1e6 / max(nf[f] / fw, 1e-9), 2)}
    print(f"{f:22s} {out[f]}")
if len(sys.argv) > 4:
    cfg = sys.argv[6] if len(sys.argv) > 6 else "2"
    json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), read side x2 (gfx950), per slice of BASELINE config {cfg} (one forward)",
               "config": int(cfg), "commit": sys.argv[5] if len(sys.argv) > 5 else None, "families": out},
              open(sys.argv[4], "w"), indent=1)
