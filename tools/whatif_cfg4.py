"""What-if runs for the cfg-4 timed region (NOT a bench line: the model is changed): how much of the in-flight step time the
conjugate-gradient DC blocks and the U-Net3D cost.  usage: [CINE_FUSED_CG=0] whatif_cfg4.py <cg_iters> [bench flags]
Prints bench.py's JSON line for a CineNet with `cg_iters` CG iterations per cascade (6 = the real configuration)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
sys.path.insert(0, os.path.join(ROOT, "deep-cine-cardiac-mri_amd"))
from cine_hip import ops
ops.FUSED_CG = os.environ.get("CINE_FUSED_CG", "1") != "0"      # A/B: 0 = the four-launch conjugate-gradient iteration
cg = int(sys.argv[1])
base = bench.CONFIGS[4]
def variant():
    import reconstruction.models as M
    d = base()
    d["hip"] = lambda: M.CineNet(6, cg, 16, 3, "3D")
    d["name"] = f"WHAT-IF cfg 4 with {cg} CG iterations"
    return d
bench.CONFIGS[4] = variant
sys.argv = [sys.argv[0], "--config", "4", "--no-cpu-baseline", "--repeats", "0", "--headline-only"] + sys.argv[2:]
bench.main()
