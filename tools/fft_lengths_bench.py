"""Line-FFT engines by length: cine_fft2c on nimg planes of h x w, HIP events on the current stream, with torch.fft.fft2 (rocFFT,
no shifts) beside it for scale.  Algorithmic bytes = one read + one write of the planes (SURVEY 8d counts a 2-D FFT as one pass).
  python tools/fft_lengths_bench.py  ->  one JSON line per shape"""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "deep-cine-cardiac-mri_amd"))
from cine_hip import ops


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for nimg, h, w in ((225, 200, 200), (225, 192, 160), (225, 256, 320), (64, 384, 512), (225, 198, 202), (225, 399, 77)):
    x = torch.randn(nimg, h, w, 2, device="cuda")
    z = torch.view_as_complex(x)
    ms = timed(lambda: ops.fft2c(x))
    ms_t = timed(lambda: torch.fft.fft2(z, norm="ortho"))
    gb = 2 * x.numel() * 4 / 1e9
    print(json.dumps({"planes": nimg, "h": h, "w": w, "cine_fft2c_ms": round(ms, 4), "GB_per_s": round(gb / ms * 1e3, 1),
                      "rocfft_fft2_ms": round(ms_t, 4)}))
