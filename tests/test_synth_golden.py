"""Host-side input preparation (masks, apply_mask, seeded weights) vs the reference,
and the oracle on the full-size BASELINE configs (fingerprints)."""
import numpy as np
import torch

from conftest import rel_err, rnd
from cine_hip import synth
from oracle import varnet_ref as V


def test_random_masks_bit_exact(golden):
    g = golden("masks")
    for seed, acc, nx, nt in ((0, 4, 200, 15), (3, 8, 200, 15), (5, 6, 64, 4)):
        np.random.seed(seed)
        mf = synth.create_mask_for_mask_type("random", [10], [acc])
        m = mf((nt, 1, nx, 7, 2), None)
        assert np.array_equal(m.numpy(), g[f"random_s{seed}_a{acc}_n{nx}"])
        assert int(m.sum()) == nt * int(nx / acc)            # lines per frame incl. centre


def test_equispaced_and_apply_mask(golden):
    g = golden("masks")
    mf = synth.create_mask_for_mask_type("equispaced", [0.08], [4])
    assert np.array_equal(mf((15, 1, 200, 7, 2), 9).numpy(), g["equi_s9"])
    np.random.seed(2)
    mf = synth.create_mask_for_mask_type("random", [4], [4])
    md, m = synth.apply_mask(torch.from_numpy(g["am_data"]), mf, None)
    assert np.array_equal(m.numpy(), g["am_mask"]) and np.array_equal(md.numpy(), g["am_masked"])


def test_make_cine_slice_shapes():
    ex = synth.make_cine_slice(5, 3, 24, 20, accel=4, center_lines=4, seed=0)
    assert ex["masked_kspace"].shape == (1, 5, 3, 24, 20, 2)
    assert ex["mask"].shape == (1, 5, 1, 24, 1, 1) and ex["mask"].dtype == torch.uint8
    assert ex["sens_maps"].shape == (1, 1, 3, 24, 20, 2)
    # zero rows where the mask is zero
    m = ex["mask"].float()
    assert torch.all(ex["masked_kspace"] * (1 - m) == 0)


def test_fill_parameters_is_name_keyed():
    a = V.VarNet(1, 4, 2, 4, 2, "XF"); b = V.VarNet(1, 4, 2, 4, 2, "XF")
    synth.fill_parameters_(a, 5); synth.fill_parameters_(b, 5)
    for (n1, p1), (n2, p2) in zip(a.named_parameters(), b.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2)


def test_oracle_cfg1_full(golden):
    """BASELINE configs[0]: 2D VarNet, 2 cascades, 8 coils, one 200x200 frame, R=4."""
    g = golden("varnet_cfg1")
    ex = synth.make_cine_slice(1, 8, 200, 200, accel=4, seed=int(g["data_seed"]))
    assert np.array_equal(ex["mask"].numpy(), g["mask"])
    net = V.VarNet(2, 8, 3, 16, 3, "2D").eval()
    synth.fill_parameters_(net, int(g["weight_seed"]))
    with torch.no_grad():
        out = net(ex["masked_kspace"], ex["mask"])
    assert rel_err(out, g["out"]) < 1e-4


def test_oracle_cfg2_full(golden):
    """BASELINE configs[1]: XF-VarNet, 6 cascades, 15 coils x 15 frames x 200x200, R=4."""
    g = golden("varnet_cfg2")
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=4, seed=int(g["data_seed"]))
    assert np.array_equal(ex["mask"].numpy(), g["mask"])
    net = V.VarNet(6, 8, 3, 16, 3, "XF").eval()
    synth.fill_parameters_(net, int(g["weight_seed"]))
    with torch.no_grad():
        out = net(ex["masked_kspace"], ex["mask"])
    assert rel_err(out[:, :, ::4, ::4], g["out_strided"]) < 1e-4
    assert abs(float(out.double().sum()) - float(g["out_sum"])) / float(g["out_sum"]) < 1e-5


def _check_fingerprint(out, g, tol=1e-4):
    assert rel_err(out[:, :, ::4, ::4], g["out_strided"]) < tol
    assert abs(float(out.double().sum()) - float(g["out_sum"])) / float(g["out_sum"]) < 1e-5
    assert abs(float(out.double().norm()) - float(g["out_l2"])) / float(g["out_l2"]) < 1e-5


def test_oracle_cfg3_full(golden):
    """BASELINE configs[2]: XT-XPDNet (MWCNN, script defaults), 10 cascades, n_primal 5, 15 coils x 15 frames x 200x200, R=8."""
    from oracle import xpdnet_ref as X
    g = golden("xpdnet_cfg3")
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=int(g["accel"]), seed=int(g["data_seed"]), noise_std=float(g["noise_std"]))
    assert np.array_equal(ex["mask"].numpy(), g["mask"])
    net = X.XPDNet(num_cascades=10, sens_chans=8, sens_pools=3, n_primal=5, dynamic_type="XT").eval()
    synth.fill_parameters_(net, int(g["weight_seed"]), keep=())
    with torch.no_grad():
        out = net(ex["masked_kspace"], ex["mask"])
    _check_fingerprint(out, g)


def test_oracle_cfg4_full(golden):
    """BASELINE configs[3]: 3D CineNet, 6 cascades, CG 6, 15 coils x 15 frames x 200x200, R=6."""
    from oracle import cinenet_ref as C
    g = golden("cinenet_cfg4")
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=int(g["accel"]), seed=int(g["data_seed"]))
    assert np.array_equal(ex["mask"].numpy(), g["mask"])
    net = C.CineNet(6, 6, 16, 3, "3D").eval()
    synth.fill_parameters_(net, int(g["weight_seed"]))
    with torch.no_grad():
        out = net(ex["masked_kspace"], ex["mask"], ex["sens_maps"])
    _check_fingerprint(out, g)


def test_oracle_cfg5_full(golden):
    """BASELINE configs[4]: CRNN-VarNet, 5 cascades, 15 coils x 15 frames x 200x200, R=8."""
    from oracle import recurrent_ref as R
    g = golden("rnn_cfg5")
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=int(g["accel"]), seed=int(g["data_seed"]))
    assert np.array_equal(ex["mask"].numpy(), g["mask"])
    net = R.VarNet_RNN(5, 8, 3, 16).eval()
    synth.fill_parameters_(net, int(g["weight_seed"]))
    with torch.no_grad():
        out = net(ex["masked_kspace"], ex["mask"])
    _check_fingerprint(out, g)


def test_host_metrics_vs_reference_ssimloss(golden):
    """The numpy SSIM (utils/evaluate.py restatement of skimage's defaults) with SSIMLoss's per-frame data range reproduces
    the reference's SSIMLoss (utils/losses.py:25-58) on the committed vector."""
    import importlib.util, os
    from conftest import PKG
    spec = importlib.util.spec_from_file_location("_ev", os.path.join(PKG, "reconstruction", "utils", "evaluate.py"))
    ev = importlib.util.module_from_spec(spec); spec.loader.exec_module(ev)
    g = golden("metrics")
    rs = np.random.RandomState(int(g["seed"]))
    tgt = rs.uniform(0, 1.5, size=(15, 180, 180)).astype(np.float32)
    rec = np.maximum(tgt + (0.1 * rs.standard_normal((15, 180, 180))).astype(np.float32), 0)
    frames = np.array([1.0 - ev._ssim2d(tgt[i], rec[i], float(tgt[i].max())) for i in range(15)])
    assert np.abs(frames - g["ssim_loss_frames"]).max() < 2e-5
    assert abs(frames.mean() - float(g["ssim_loss"])) < 2e-5
