"""Pin the CPU oracle (oracle/) against the reference's own outputs.

The golden vectors were produced by importing the reference in the build
container (tests/golden/make_golden.py).  CPU only.
"""
import numpy as np
import pytest
import torch

from conftest import rel_err, rnd, state_dict_from
from oracle import centered_fft as cf, complex_ops as co
from oracle import regularisers as R, varnet_ref as V

OP_TOL = 1e-5      # op-level relative tolerance


@pytest.mark.parametrize("tag", ["odd", "t15", "even", "mixed"])
def test_centered_ffts(golden, tag):
    g = golden("ops")
    x = torch.from_numpy(g[f"{tag}_x"])
    for name, fn in (("fft1c", cf.fft1c), ("ifft1c", cf.ifft1c), ("fft2c", cf.fft2c),
                     ("ifft2c", cf.ifft2c)):
        assert rel_err(fn(x), g[f"{tag}_{name}"]) < OP_TOL, name
    assert torch.equal(cf.fftshift(x, [-3, -2]), torch.from_numpy(g[f"{tag}_fftshift"]))
    assert torch.equal(cf.ifftshift(x, [-3, -2]), torch.from_numpy(g[f"{tag}_ifftshift"]))


def test_fft2c_200(golden):
    g = golden("ops")
    x = rnd(int(g["full200_seed"]), 2, 200, 200, 2)
    assert rel_err(cf.fft2c(x), g["full200_fft2c"]) < OP_TOL
    assert rel_err(cf.ifft2c(x), g["full200_ifft2c"]) < OP_TOL
    assert rel_err(cf.ifft2c(cf.fft2c(x)), x) < OP_TOL


SMOOTH_PLANES = ["a96x120", "a192x160", "a256x320", "a384x512", "a45x250", "a400x405"]
SMOOTH_LINES = [30, 128, 360, 512]


@pytest.mark.parametrize("tag", SMOOTH_PLANES)
def test_fft2c_smooth_lengths(golden, tag):
    """The lengths the HIP mixed-radix engine serves (2^a 3^b 5^c); large planes are pinned on a strided lattice."""
    g = golden("fft_smooth")
    n, h, w = (int(v) for v in g[f"{tag}_shape"])
    sh, sw = (int(v) for v in g[f"{tag}_stride"])
    x = rnd(int(g[f"{tag}_seed"]), n, h, w, 2)
    assert rel_err(cf.fft2c(x)[:, ::sh, ::sw], g[f"{tag}_fft2c"]) < OP_TOL
    assert rel_err(cf.ifft2c(x)[:, ::sh, ::sw], g[f"{tag}_ifft2c"]) < OP_TOL


@pytest.mark.parametrize("n", SMOOTH_LINES)
def test_fft1c_smooth_lengths(golden, n):
    g = golden("fft_smooth")
    x = rnd(int(g[f"l{n}_seed"]), 3, n, 2)
    assert rel_err(cf.fft1c(x), g[f"l{n}_fft1c"]) < OP_TOL
    assert rel_err(cf.ifft1c(x), g[f"l{n}_ifft1c"]) < OP_TOL


def test_fft_rejects_non_pair():
    with pytest.raises(ValueError):
        cf.fft2c(torch.zeros(4, 4, 3))
    with pytest.raises(ValueError):
        co.complex_mul(torch.zeros(4, 2), torch.zeros(4, 3))


def test_complex_ops(golden):
    g = golden("ops")
    x, y = torch.from_numpy(g["cm_x"]), torch.from_numpy(g["cm_y"])
    assert rel_err(co.complex_mul(x, y), g["cm_mul"]) < 1e-6
    assert torch.equal(co.complex_conj(x), torch.from_numpy(g["cm_conj"]))
    assert rel_err(co.complex_abs(x), g["cm_abs"]) < 1e-6
    assert rel_err(co.complex_abs_sq(x), g["cm_abs_sq"]) < 1e-6
    assert rel_err(co.rss(x, 1), g["cm_rss"]) < 1e-6
    assert rel_err(co.rss_complex(x, 1), g["cm_rss_complex"]) < 1e-6
    z = co.real_to_complex_multi_ch(torch.from_numpy(g["mc_r"]), 6)
    assert np.array_equal(z.real.numpy(), g["mc_z_re"]) and np.array_equal(z.imag.numpy(), g["mc_z_im"])
    assert np.array_equal(co.complex_to_real_multi_ch(z).numpy(), g["mc_back"])
    assert np.array_equal(co.mask_center(torch.from_numpy(g["mcen_x"]), 4, 9).numpy(), g["mcen_out"])


def test_unet_blocks(golden):
    g = golden("unet")
    cb = R.ConvBlock(3, 8, 0.0, 2).eval()
    cb.load_state_dict(state_dict_from(g, "cb::"), strict=True)
    tb = R.TransposeConvBlock(8, 4, 2).eval()
    tb.load_state_dict(state_dict_from(g, "tb::"), strict=True)
    un = R.Unet(chans=4, num_pool_layers=2).eval()
    un.load_state_dict(state_dict_from(g, "un::"), strict=True)
    with torch.no_grad():
        assert rel_err(cb(torch.from_numpy(g["cb_x"])), g["cb_y"]) < OP_TOL
        assert rel_err(tb(torch.from_numpy(g["tb_x"])), g["tb_y"]) < OP_TOL
        assert rel_err(un(torch.from_numpy(g["un_x"])), g["un_y"]) < OP_TOL
        assert rel_err(un(torch.from_numpy(g["un_odd_x"])), g["un_odd_y"]) < OP_TOL


def test_norm_unets(golden):
    g = golden("unet")
    nu = R.NormUnet(4, 2).eval()
    nu.load_state_dict(state_dict_from(g, "nu::"), strict=True)
    nu3 = R.NormUnet3D(4, 2).eval()
    nu3.load_state_dict(state_dict_from(g, "nu3::"), strict=True)
    with torch.no_grad():
        x = torch.from_numpy(g["nu_x"])
        b, c, h, w, _ = x.shape
        xn, mean, std = nu.norm(x.permute(0, 4, 1, 2, 3).reshape(b, 2 * c, h, w))
        assert rel_err(xn, g["nu_norm"]) < OP_TOL
        assert rel_err(mean, g["nu_mean"]) < OP_TOL and rel_err(std, g["nu_std"]) < OP_TOL
        assert rel_err(nu(x), g["nu_y"]) < OP_TOL
        assert rel_err(nu3(torch.from_numpy(g["nu3_x"])), g["nu3_y"]) < OP_TOL
    with pytest.raises(ValueError):
        nu(torch.zeros(1, 1, 8, 8, 3))


@pytest.mark.parametrize("dyn", ["XF", "XT", "2D", "3D"])
def test_varnet_block(golden, dyn):
    g = golden("varnet_block")
    net = V.VarNet(1, 4, 2, 4, 2, dyn).eval()
    net.load_state_dict(state_dict_from(g, f"{dyn}::sd::"), strict=True)
    blk = net.cascades[0]
    k, kref, sens = (torch.from_numpy(g[n]) for n in ("k", "kref", "sens"))
    mask = torch.from_numpy(g["mask"])
    with torch.no_grad():
        img = blk.sens_reduce(k, sens)
        assert rel_err(img, g[f"{dyn}_reduce"]) < OP_TOL
        assert rel_err(blk.sens_expand(img, sens), g[f"{dyn}_expand"]) < OP_TOL
        if dyn in ("XF", "XT"):
            assert rel_err(blk.xfyf_transform(img.squeeze(2)), g[f"{dyn}_xfyf"]) < 2e-5
        assert rel_err(blk(k, kref, mask, sens), g[f"{dyn}_block"]) < 2e-5


def test_sensitivity_model(golden):
    g = golden("varnet_block")
    sm = V.SensitivityModel(4, 2).eval()
    sm.load_state_dict(state_dict_from(g, "sens::sd::"), strict=True)
    with torch.no_grad():
        out = sm(torch.from_numpy(g["kref"]), torch.from_numpy(g["mask"]))
    assert rel_err(out, g["sens_out"]) < 2e-5


@pytest.mark.parametrize("tag,dyn,ws", [("XF", "XF", False), ("XT", "XT", False), ("2D", "2D", False),
                                        ("3D", "3D", False), ("XFws", "XF", True)])
def test_varnet_tiny(golden, tag, dyn, ws):
    g = golden("varnet_tiny")
    net = V.VarNet(2, 4, 2, 4, 2, dyn, ws).eval()
    missing = net.load_state_dict(state_dict_from(g, f"{tag}::sd::"), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    with torch.no_grad():
        out = net(torch.from_numpy(g["masked_kspace"]), torch.from_numpy(g["mask"]))
    assert out.shape == g[f"{tag}_out"].shape
    assert rel_err(out, g[f"{tag}_out"]) < 1e-4


# ------------------------------------------------------------------ CineNet
def test_cinenet_block_pieces(golden):
    from oracle import cinenet_ref as C
    g = golden("cinenet")
    net = C.CineNet(2, 3, 4, 2, "XF").eval()
    net.load_state_dict(state_dict_from(g, "XF::sd::"), strict=True)
    blk = net.cascades[0]
    mk, sens, mask = (torch.from_numpy(g[n]) for n in ("masked_kspace", "sens", "mask"))
    with torch.no_grad():
        img = blk.sens_reduce(mk, sens)
        assert rel_err(img, g["img"]) < OP_TOL
        assert rel_err(blk.HOperator(img, mask, sens), g["H_img"]) < OP_TOL
        assert rel_err(blk.xfyf_transform(img.squeeze(2)), g["xfyf"]) < 2e-5
        assert rel_err(blk.ConjGrad(torch.from_numpy(g["xfyf"]), torch.from_numpy(g["cg_rhs"]), mask, sens, 3), g["cg_out"]) < 2e-5
        assert rel_err(blk(img, img, mask, sens), g["block_out"]) < 5e-5


@pytest.mark.parametrize("tag,dyn,ws", [("XF", "XF", False), ("XT", "XT", False), ("2D", "2D", False),
                                        ("3D", "3D", False), ("XFws", "XF", True)])
def test_cinenet_tiny(golden, tag, dyn, ws):
    from oracle import cinenet_ref as C
    g = golden("cinenet")
    net = C.CineNet(2, 3, 4, 2, dyn, ws).eval()
    net.load_state_dict(state_dict_from(g, f"{tag}::sd::"), strict=True)
    with torch.no_grad():
        out = net(torch.from_numpy(g["masked_kspace"]), torch.from_numpy(g["mask"]), torch.from_numpy(g["sens"]))
    assert rel_err(out, g[f"{tag}_out"]) < 1e-4


# ------------------------------------------------------------------ XPDNet / MWCNN
def test_wavelets_and_mwcnn_padding(golden):
    from oracle import xpdnet_ref as X
    g = golden("xpdnet")
    x = torch.from_numpy(g["dwt_x"])
    assert rel_err(X.DWT()(x), g["dwt_y"]) < 1e-6
    assert rel_err(X.IWT()(X.DWT()(x)), g["iwt_y"]) < 1e-6 and rel_err(X.IWT()(X.DWT()(x)), x) < 1e-6
    for tag in ("p1", "p2", "p3"):
        y, pads = X.pad_for_mwcnn(torch.from_numpy(g[f"{tag}_x"]), 3)
        assert list(pads) == list(g[f"{tag}_pads"]) and np.array_equal(y.numpy(), g[f"{tag}_y"])
        assert np.array_equal(X.unpad_from_mwcnn(y, pads).numpy(), g[f"{tag}_back"])


def test_mwcnn(golden):
    from oracle import xpdnet_ref as X
    g = golden("xpdnet")
    mw = X.MWCNN(in_chans=6, out_chans=4, n_scales=2, n_filters_per_scale=[8, 16], n_convs_per_scale=[2, 1],
                 first_conv_n_filters=8).eval()
    mw.load_state_dict(state_dict_from(g, "mw::"), strict=True)
    with torch.no_grad():
        assert rel_err(mw(torch.from_numpy(g["mw_x"])), g["mw_y"]) < 2e-5


@pytest.mark.parametrize("tag,dyn,ws,po", [("XF", "XF", False, True), ("XT", "XT", False, True), ("2D", "2D", False, True),
                                           ("XFws", "XF", True, True), ("XFdual", "XF", False, False)])
def test_xpdnet_tiny(golden, tag, dyn, ws, po):
    from oracle import xpdnet_ref as X
    g = golden("xpdnet")
    net = X.XPDNet(num_cascades=2, sens_chans=4, sens_pools=2, n_scales=2, n_filters_per_scale=[8, 16],
                   n_convs_per_scale=[1, 1], first_conv_n_filters=8, n_primal=2, dynamic_type=dyn,
                   weight_sharing=ws, primal_only=po).eval()
    net.load_state_dict(state_dict_from(g, f"{tag}::sd::"), strict=True)
    with torch.no_grad():
        out = net(torch.from_numpy(g["masked_kspace"]), torch.from_numpy(g["mask"]))
        if tag == "XF":
            assert rel_err(net.sens_net(torch.from_numpy(g["masked_kspace"]), torch.from_numpy(g["mask"])), g["sens_out"]) < 2e-5
    assert rel_err(out, g[f"{tag}_out"]) < 1e-4


# ------------------------------------------------------------------ CRNN hybrids
def test_rnn_models(golden):
    from oracle import recurrent_ref as R
    g = golden("rnn")
    mk, mask, sens = (torch.from_numpy(g[n]) for n in ("masked_kspace", "mask", "sens"))
    with torch.no_grad():
        net = R.VarNet_RNN(3, 4, 2, 6).eval(); net.load_state_dict(state_dict_from(g, "varnet_rnn::sd::"), strict=True)
        assert rel_err(net(mk, mask), g["varnet_rnn_out"]) < 1e-4
        net = R.CineNet_RNN(3, 3, 6).eval(); net.load_state_dict(state_dict_from(g, "cinenet_rnn::sd::"), strict=True)
        assert rel_err(net(mk, mask, sens), g["cinenet_rnn_out"]) < 1e-4
        net = R.XPDNet_RNN(3, 4, 2, 6, True, 2, 1).eval(); net.load_state_dict(state_dict_from(g, "xpdnet_rnn::sd::"), strict=True)
        assert rel_err(net(mk, mask), g["xpdnet_rnn_out"]) < 1e-4


@pytest.mark.parametrize("tag,dyn,share", [("XF", "XF", False), ("XT", "XT", False), ("2D", "2D", False), ("XFws", "XF", True), ("3D", "3D", False)])
def test_oracle_training_step_gradients_vs_reference_golden(golden, tag, dyn, share):
    """The oracle under autograd reproduces the reference's training-step gradients (pl_modules/varnet_module.py:97-113 +
    loss.backward(); varnet_grad.npz) -- it is the checker of the HIP backward kernels (tests/test_hip_grad.py)."""
    from reconstruction.utils.losses import SSIMLoss        # the build's device-agnostic SSIMLoss (pinned to the reference's by metrics.npz)
    g = golden("varnet_grad")
    net = V.VarNet(2, 4, 2, 4, 2, dyn, share)
    net.load_state_dict(state_dict_from(g, f"{tag}::sd::"), strict=True)
    mk, mask, target = (torch.from_numpy(g[k]) for k in ("masked_kspace", "mask", "target"))
    with torch.enable_grad():
        out = net(mk, mask)
        h0, w0 = (out.shape[-2] - target.shape[-2]) // 2, (out.shape[-1] - target.shape[-1]) // 2
        crop = out[..., h0:h0 + target.shape[-2], w0:w0 + target.shape[-1]]
        loss = SSIMLoss()(crop.unsqueeze(1), target.unsqueeze(1), data_range=target.max())
        loss.backward()
    assert abs(float(loss) - float(g[f"{tag}_loss"])) < 1e-6
    for k, p in net.named_parameters():
        floor = float(g[f"{tag}::floor::{k}"])
        assert rel_err(p.grad, g[f"{tag}::grad::{k}"]) < max(1e-5, 20 * floor), k


@pytest.mark.parametrize("tag,dyn,share", [("XF", "XF", False), ("2D", "2D", False), ("3D", "3D", False)])
def test_oracle_cinenet_training_gradients_vs_reference_golden(golden, tag, dyn, share):
    """The CineNet oracle under autograd (conjugate gradients with detached step sizes, cinenet.py:159-169) reproduces the
    reference's training-step gradients (cinenet_grad.npz)."""
    from oracle import cinenet_ref as C
    from reconstruction.utils.losses import SSIMLoss
    g = golden("cinenet_grad")
    net = C.CineNet(2, 3, 4, 2, dyn, share)
    net.load_state_dict(state_dict_from(g, f"{tag}::sd::"), strict=True)
    mk, mask, target, sens = (torch.from_numpy(g[k]) for k in ("masked_kspace", "mask", "target", "sens_maps"))
    with torch.enable_grad():
        out = net(mk, mask, sens)
        h0, w0 = (out.shape[-2] - target.shape[-2]) // 2, (out.shape[-1] - target.shape[-1]) // 2
        crop = out[..., h0:h0 + target.shape[-2], w0:w0 + target.shape[-1]]
        loss = SSIMLoss()(crop.unsqueeze(1), target.unsqueeze(1), data_range=target.max())
        loss.backward()
    assert abs(float(loss) - float(g[f"{tag}_loss"])) < 1e-6
    for k, p in net.named_parameters():
        floor = float(g[f"{tag}::floor::{k}"])
        assert rel_err(p.grad, g[f"{tag}::grad::{k}"]) < max(1e-5, 20 * floor), k


_XPD_GRAD_KW = dict(num_cascades=2, sens_chans=4, sens_pools=2, n_scales=2, n_filters_per_scale=[8, 16], n_convs_per_scale=[2, 1],
                    first_conv_n_filters=8, n_primal=2)


@pytest.mark.parametrize("tag,dyn,share", [("XF", "XF", False), ("XT", "XT", False)])
def test_oracle_xpdnet_training_gradients_vs_reference_golden(golden, tag, dyn, share):
    """The XPDNet oracle under autograd reproduces the reference's training-step gradients (xpdnet_grad.npz)."""
    from oracle import xpdnet_ref as X
    from reconstruction.utils.losses import SSIMLoss
    g = golden("xpdnet_grad")
    net = X.XPDNet(dynamic_type=dyn, weight_sharing=share, primal_only=True, **_XPD_GRAD_KW)
    net.load_state_dict(state_dict_from(g, f"{tag}::sd::"), strict=True)
    mk, mask, target = (torch.from_numpy(g[k]) for k in ("masked_kspace", "mask", "target"))
    with torch.enable_grad():
        out = net(mk, mask)
        h0, w0 = (out.shape[-2] - target.shape[-2]) // 2, (out.shape[-1] - target.shape[-1]) // 2
        crop = out[..., h0:h0 + target.shape[-2], w0:w0 + target.shape[-1]]
        loss = SSIMLoss()(crop.unsqueeze(1), target.unsqueeze(1), data_range=target.max())
        loss.backward()
    assert abs(float(loss) - float(g[f"{tag}_loss"])) < 1e-6
    for k, p in net.named_parameters():
        floor = float(g[f"{tag}::floor::{k}"])
        assert rel_err(p.grad, g[f"{tag}::grad::{k}"]) < max(1e-5, 20 * floor), k


@pytest.mark.parametrize("tag", ["varnet_rnn", "cinenet_rnn", "xpdnet_rnn"])
def test_oracle_rnn_training_gradients_vs_reference_golden(golden, tag):
    """The CRNN oracles under autograd reproduce the reference's training-step gradients (rnn_grad.npz)."""
    from oracle import recurrent_ref as R
    from reconstruction.utils.losses import SSIMLoss
    g = golden("rnn_grad")
    net = {"varnet_rnn": lambda: R.VarNet_RNN(3, 4, 2, 6), "cinenet_rnn": lambda: R.CineNet_RNN(3, 3, 6),
           "xpdnet_rnn": lambda: R.XPDNet_RNN(3, 4, 2, 6, True, 2, 1)}[tag]()
    net.load_state_dict(state_dict_from(g, f"{tag}::sd::"), strict=True)
    mk, mask, target, sens = (torch.from_numpy(g[k]) for k in ("masked_kspace", "mask", "target", "sens_maps"))
    with torch.enable_grad():
        out = net(mk, mask, sens) if tag == "cinenet_rnn" else net(mk, mask)
        h0, w0 = (out.shape[-2] - target.shape[-2]) // 2, (out.shape[-1] - target.shape[-1]) // 2
        crop = out[..., h0:h0 + target.shape[-2], w0:w0 + target.shape[-1]]
        loss = SSIMLoss()(crop.unsqueeze(1), target.unsqueeze(1), data_range=target.max())
        loss.backward()
    assert abs(float(loss) - float(g[f"{tag}_loss"])) < 1e-6
    for k, p in net.named_parameters():
        floor = float(g[f"{tag}::floor::{k}"])
        assert rel_err(p.grad, g[f"{tag}::grad::{k}"]) < max(1e-5, 20 * floor), k
