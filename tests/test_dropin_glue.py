"""Drop-in boundary against the reference's own callers.

Build-container only: needs the reference checkout (CINE_REFERENCE_ROOT or /root/reference), which never travels to the
GPU box -- skipped there.  Each case runs in a fresh interpreter with exactly the path order INTEGRATION.md section 1
prescribes (this build's package first, the reference checkout second) and stubs for the third-party packages the image
lacks (pytorch_lightning, torchmetrics, bart, h5py, skimage): import-time stand-ins only, none of them is on the path
under test.  The reference-made checkpoint is produced by a second interpreter that sees ONLY the reference.
"""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import PKG, ROOT

REF = os.environ.get("CINE_REFERENCE_ROOT", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "reconstruction")),
                                reason="reference checkout not present (build container only)")

STUBS = textwrap.dedent("""
    import sys, types, torch
    for name in ("bart", "h5py", "torchmetrics", "skimage", "skimage.metrics"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["skimage.metrics"].peak_signal_noise_ratio = None
    sys.modules["skimage.metrics"].structural_similarity = None
    pl = types.ModuleType("pytorch_lightning")
    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k): pass
    class LightningDataModule: pass
    class Metric(torch.nn.Module):
        def __init__(self, dist_sync_on_step=True): super().__init__()
        def add_state(self, name, default, dist_reduce_fx=None): self.register_buffer(name, default)
    pl.LightningModule, pl.LightningDataModule = LightningModule, LightningDataModule
    pl.metrics = types.ModuleType("pytorch_lightning.metrics"); pl.metrics.Metric = Metric
    sys.modules["pytorch_lightning"] = pl; sys.modules["pytorch_lightning.metrics"] = pl.metrics
""")


def _run(code, pythonpath, extra_env=None):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join(pythonpath)
    env.pop("CINE_REFERENCE_ROOT", None)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, "-c", STUBS + textwrap.dedent(code)], env=env, capture_output=True, text=True,
                       timeout=600, cwd="/tmp")
    assert r.returncode == 0, r.stdout + "\n" + r.stderr
    return r.stdout


def test_reference_import_lines_resolve_with_integration_path_order():
    """The import lines of pl_modules/varnet_module.py:4-6, data_module.py:15, mri_module.py:18-19 and run_inference.py."""
    out = _run("""
        import os
        import reconstruction
        from reconstruction.data import transforms
        from reconstruction.utils import SSIMLoss
        from reconstruction.models import VarNet, VarNet_RNN, CineNet, CineNet_RNN, XPDNet, XPDNet_RNN
        from reconstruction.data import CombinedSliceDataset, SliceDataset, VolumeSampler
        from reconstruction.data.mri_data import fetch_dir
        from reconstruction.utils import evaluate
        from reconstruction.data.transforms import VarNetDataTransform, center_crop_to_smallest, to_tensor, mask_center
        from reconstruction.data.subsample import create_mask_for_mask_type
        import reconstruction.pl_modules as P
        build = os.environ["BUILD_PKG"]; ref = os.environ["CINE_REFERENCE_ROOT"]
        here = lambda o: os.path.abspath(sys.modules[o.__module__].__file__)
        assert here(VarNet).startswith(build) and here(SSIMLoss).startswith(build) and here(XPDNet_RNN).startswith(build)
        assert os.path.abspath(evaluate.__file__).startswith(build)
        assert here(SliceDataset).startswith(ref) and here(VolumeSampler).startswith(ref) and here(fetch_dir).startswith(ref)
        assert here(VarNetDataTransform).startswith(ref) and here(P.VarNetModule).startswith(ref)
        assert mask_center.__module__ == "reconstruction.data.transforms" and here(mask_center).startswith(build)
        print("ok")
    """, [PKG, REF], {"CINE_REFERENCE_ROOT": REF, "BUILD_PKG": PKG})
    assert out.strip().endswith("ok")


def test_varnet_module_constructs_on_the_build_and_loads_a_reference_checkpoint(tmp_path):
    ckpt = str(tmp_path / "ref_varnet_module_state.pt")
    # (1) reference only: a Lightning-style state dict of the DEFAULT VarNetModule (varnet_module.py:74-90), 'varnet.' prefix
    _run(f"""
        torch.Tensor.cuda = lambda self, *a, **k: self
        import reconstruction.utils
        from reconstruction.models import VarNet
        import reconstruction.models.varnet as V
        assert V.__file__.startswith({REF!r})
        torch.manual_seed(3)
        net = VarNet(num_cascades=12, sens_chans=8, sens_pools=4, chans=18, pools=4, dynamic_type="XF", weight_sharing=False)
        torch.save({{"varnet." + k: v for k, v in net.state_dict().items()}}, {ckpt!r})
    """, [REF])
    # (2) build first, reference second: the reference's VarNetModule, unmodified, wraps the build's VarNet
    out = _run(f"""
        import os, pathlib
        import reconstruction.pl_modules as P
        import reconstruction.pl_modules.mri_module as MM
        MM.fetch_dir = lambda key, cfg=None: pathlib.Path("/tmp")      # MriModule.__init__ writes a yaml next to the scripts otherwise
        m = P.VarNetModule()
        import reconstruction.models.varnet as V
        assert V.__file__.startswith(os.environ["BUILD_PKG"]) and type(m.varnet) is V.VarNet
        assert type(m.loss).__module__ == "reconstruction.utils.losses"
        sd = torch.load({ckpt!r})
        own = {{k: v for k, v in m.state_dict().items() if k.startswith("varnet.")}}
        assert set(own) == set(sd), (sorted(set(own) ^ set(sd))[:5])
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected and all(not k.startswith("varnet.") for k in missing), (missing[:5], unexpected[:5])
        m.varnet.load_state_dict({{k[len("varnet."):]: v for k, v in sd.items()}}, strict=True)
        k0 = "varnet.cascades.3.model.0.unet.conv.layers.0.weight"
        assert torch.equal(m.state_dict()[k0], sd[k0]) and m.state_dict()[k0].data_ptr() == m.state_dict()["varnet.model.0.unet.conv.layers.0.weight"].data_ptr()
        # the module's forward is the reference's code calling the build's model: GPU tensors only, loud on CPU
        from cine_hip._lib import CineHipError
        try:
            m(torch.zeros(1, 2, 2, 16, 16, 2), torch.ones(1, 2, 1, 16, 1, 1, dtype=torch.uint8))
        except (CineHipError, IndexError):
            print("ok")
    """, [PKG, REF], {"CINE_REFERENCE_ROOT": REF, "BUILD_PKG": PKG})
    assert out.strip().endswith("ok")


def test_cinenet_and_xpdnet_modules_construct_on_the_build():
    out = _run("""
        import os, pathlib
        import reconstruction.pl_modules as P
        import reconstruction.pl_modules.mri_module as MM
        MM.fetch_dir = lambda key, cfg=None: pathlib.Path("/tmp")
        build = os.environ["BUILD_PKG"]
        c = P.CineNetModule(); x = P.XPDNetModule()
        for mod in (c.cinenet, x.xpdnet):
            assert os.path.abspath(sys.modules[type(mod).__module__].__file__).startswith(build), type(mod)
        r = P.VarNetModule(dynamic_type="CRNN")
        assert type(r.varnet).__name__ == "VarNet_RNN" and sys.modules[type(r.varnet).__module__].__file__.startswith(build)
        print("ok")
    """, [PKG, REF], {"CINE_REFERENCE_ROOT": REF, "BUILD_PKG": PKG})
    assert out.strip().endswith("ok")


def test_ssim_loss_matches_reference_on_cpu(tmp_path):
    """The build's SSIMLoss (device-agnostic) against the reference's (utils/losses.py:25-58, hard-wired .to('cuda'))."""
    vec = str(tmp_path / "ssim_vec.pt")
    _run(f"""
        torch.Tensor.to = (lambda orig: (lambda self, *a, **k: self if a and a[0] == "cuda" else orig(self, *a, **k)))(torch.Tensor.to)
        from reconstruction.utils.losses import SSIMLoss
        import reconstruction.utils.losses as L
        assert L.__file__.startswith({REF!r})
        g = torch.Generator().manual_seed(0)
        x, y = torch.rand(1, 1, 4, 40, 36, generator=g), torch.rand(1, 1, 4, 40, 36, generator=g)
        torch.save({{"x": x, "y": y, "loss": SSIMLoss()(x, y, torch.tensor([1.0]))}}, {vec!r})
    """, [REF])
    out = _run(f"""
        from reconstruction.utils import SSIMLoss
        v = torch.load({vec!r})
        got = SSIMLoss()(v["x"], v["y"], torch.tensor([1.0]))
        assert abs(float(got) - float(v["loss"])) < 1e-6, (float(got), float(v["loss"]))
        print("ok")
    """, [PKG, REF], {"CINE_REFERENCE_ROOT": REF})
    assert out.strip().endswith("ok")
