"""CPU-side checks of the product: C-ABI surface, drop-in module tree, loud failure without a GPU."""
import ctypes
import os
import subprocess

import pytest
import torch

from conftest import PKG, ROOT, state_dict_from
from cine_hip import _lib


def test_library_exports_every_declared_symbol():
    declared = _lib.declared_symbols()
    assert len(declared) >= 20
    handle = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared if not hasattr(handle, s)]
    assert not missing, f"declared in include/cine_hip.h but not exported: {missing}"
    # and the Python binding table covers the same set
    assert set(declared) == set(_lib._SIGS)
    assert _lib.lib().cine_version() == 3
    assert _lib.lib().cine_build_arch() == b"gfx950"
    assert _lib.lib().cine_pad16(200) == 208 and _lib.lib().cine_pad16(15) == 16 and _lib.lib().cine_pad16(16) == 16


def test_argument_validation_without_gpu():
    """Bad arguments are rejected on the host before any launch."""
    L = _lib.lib()
    assert L.cine_fft2c(None, None, 1, 8, 8, 0, None) == -1
    assert b"null" in L.cine_last_error()
    assert L.cine_conv3x3_packed_floats(16, 2) == 1 * 9 * 8 * 16
    assert L.cine_conv3x3_packed_floats(10, 12) == 2 * 9 * 8 * 16
    assert L.cine_tconv2x2_packed_floats(32, 16) == 2 * 16 * 64
    # LeakyReLU slopes outside [0, 1] are rejected before anything is launched (act() evaluates max(v, v * slope))
    assert L.cine_tconv2x2_in(None, None, 0, 0, None, None, 0, None, None, 1, 1, 1, 2, 2, 1e-5, 1.5, None) == -1
    assert b"slope" in L.cine_last_error()
    assert L.cine_conv_stat_partials(16, 208, 16, 0) == 4 and L.cine_conv_stat_partials(32, 104, 8, 0) == 2
    assert L.cine_conv_stat_partials(16, 104, 8, 1) == 16
    assert L.cine_unet2d_ws_bytes(4, 16, 16, 2, 2, 4, 2) > 0
    assert L.cine_unet2d_ws_bytes(4, 16, 16, 2, 2, 4, 0) == 0
    assert L.cine_xfyf_ws_bytes(1, 15, 200, 200) == 15 * 200 * 200 * 8


def test_fft_engine_host_check():
    """Compile csrc/fft_core.h for the host and compare the 200-point engine with a direct DFT."""
    exe = "/tmp/cine_fft_host_check"
    src = os.path.join(ROOT, "tests", "host", "fft_host_check.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(PKG, "csrc"), src, "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.strip().endswith("OK"), out


@pytest.mark.parametrize("tag,dyn,ws", [("XF", "XF", False), ("XT", "XT", False), ("2D", "2D", False),
                                        ("3D", "3D", False), ("XFws", "XF", True)])
def test_state_dict_is_reference_compatible(golden, tag, dyn, ws):
    """Reference checkpoints (incl. aliased cascades.N.model.* keys) load strict=True."""
    import reconstruction.models as M
    g = golden("varnet_tiny")
    sd = state_dict_from(g, f"{tag}::sd::")
    net = M.VarNet(2, 4, 2, 4, 2, dyn, ws)
    res = net.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert sorted(net.state_dict().keys()) == sorted(sd.keys())


def test_default_constructor_matches_reference_key_count():
    import reconstruction.models as M
    assert len(M.VarNet(6, 8, 3, 16, 3, "XF").state_dict()) == 291


def test_no_cpu_fallback():
    """CPU tensors must fail loudly, not silently run somewhere else."""
    import reconstruction.models as M
    import reconstruction.utils as U
    from cine_hip._lib import CineHipError
    with pytest.raises(CineHipError):
        U.fft2c(torch.zeros(1, 8, 8, 2))
    net = M.VarNet(1, 4, 2, 4, 2, "XF").eval()
    k = torch.zeros(1, 5, 3, 24, 20, 2)
    m = torch.zeros(1, 5, 1, 24, 1, 1, dtype=torch.uint8); m[:, :, :, 10:14] = 1
    with pytest.raises(CineHipError):
        net(k, m)
    with pytest.raises(ValueError):
        U.fft2c(torch.zeros(4, 4, 3))


def test_product_does_not_import_oracle():
    bad = []
    for d, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(d, f)).read()
                if "import oracle" in txt or "from oracle" in txt:
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_import_after_hip_initialisation_warns_about_hardware_queues():
    """cine_hip sets GPU_MAX_HW_QUEUES=16 on import when the user has not; that only takes effect before the HIP runtime starts, so an import
    AFTER initialisation says so (RuntimeWarning) instead of silently running the side streams on 4 shared queues."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import os, sys, types, warnings
        os.environ.pop("GPU_MAX_HW_QUEUES", None)
        sys.path[:0] = [%r]
        import torch
        torch.cuda.is_initialized = lambda: True          # stand-in for "a script touched the GPU first" (no GPU here)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            import cine_hip
        print("WARNED" if any("GPU_MAX_HW_QUEUES" in str(x.message) for x in w) else "SILENT", os.environ.get("GPU_MAX_HW_QUEUES"))
        """ % os.path.join(ROOT, "deep-cine-cardiac-mri_amd"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "WARNED 16" in r.stdout, r.stdout + r.stderr
    code2 = code.replace('os.environ.pop("GPU_MAX_HW_QUEUES", None)', 'os.environ["GPU_MAX_HW_QUEUES"] = "8"')
    r = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "SILENT 8" in r.stdout, r.stdout + r.stderr
