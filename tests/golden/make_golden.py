#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py [--only NAME]

The reference (f78bono/deep-cine-cardiac-mri) is imported unmodified with the
three shims of SURVEY.md appendix C (stub ``bart``/``h5py`` at import time,
``Tensor.cuda`` -> identity).  Every .npz holds inputs, the reference's
outputs, and (for modules) the full reference state_dict, so tests can replay
them anywhere without the reference.  Only data is written -- no reference
source text.
"""
import argparse
import os
import sys
import types
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("CINE_REFERENCE_ROOT", "/root/reference")

for _m in ("bart", "h5py"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
torch.Tensor.cuda = lambda self, *a, **k: self
sys.path.insert(0, REF)
import reconstruction.utils as RU            # noqa: E402
import reconstruction.models as RM           # noqa: E402
from reconstruction.models.denoisers import unet as r_unet, norm_unet as r_norm_unet  # noqa: E402
from reconstruction.models.denoisers import mwcnn as r_mwcnn                          # noqa: E402
from reconstruction.data import subsample as r_sub, transforms as r_tf                 # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "deep-cine-cardiac-mri_amd"))
sys.path.insert(0, ROOT)
from cine_hip import synth                    # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)


def rnd(seed, *shape):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32))


def sd_np(module, prefix="sd::"):
    """State dict as arrays; tensors registered under several names (the aliased cascades.N.model.* /
    cascades.M.image_net.* entries) are stored once, the other names go into a JSON alias table."""
    import json
    out, first, alias = {}, {}, {}
    for k, v in module.state_dict().items():
        key = (v.data_ptr(), tuple(v.shape))
        if key in first:
            alias[k] = first[key]
        else:
            first[key] = k
            out[prefix + k] = v.detach().cpu().numpy()
    out[prefix + "__alias__"] = np.frombuffer(json.dumps(alias).encode(), dtype=np.uint8)
    return out


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays)")


def tiny_mask(t, h, rows_extra=(2, 5, 20), centre=(10, 14)):
    m = torch.zeros(1, t, 1, h, 1, 1, dtype=torch.uint8)
    m[:, :, :, centre[0]:centre[1]] = 1
    for f in range(t):
        for r in rows_extra:
            m[:, f, :, (r + 3 * f) % h] = 1
    m[:, :, :, centre[0]:centre[1]] = 1
    m[:, 0, :, centre[0] - 1] = 0      # frame 0 needs zeros either side of the centre
    m[:, 0, :, centre[1]] = 0
    return m


# ------------------------------------------------------------------ generators
def g_ops():
    """fftc.py / math.py / coil_combine.py / transforms.mask_center."""
    a = {}
    for tag, shape in (("odd", (2, 5, 7, 2)), ("t15", (3, 4, 15, 2)), ("even", (2, 24, 20, 2)),
                       ("mixed", (1, 3, 9, 16, 2))):
        x = rnd(zlib.crc32(tag.encode()) % 1000, *shape)
        a[f"{tag}_x"] = x
        a[f"{tag}_fft1c"] = RU.fft1c(x)
        a[f"{tag}_ifft1c"] = RU.ifft1c(x)
        a[f"{tag}_fft2c"] = RU.fft2c(x)
        a[f"{tag}_ifft2c"] = RU.ifft2c(x)
        a[f"{tag}_fftshift"] = RU.fftshift(x, dim=[-3, -2])
        a[f"{tag}_ifftshift"] = RU.ifftshift(x, dim=[-3, -2])
    # one full-size 200x200 pair, input regenerated from RandomState(7)
    x = rnd(7, 2, 200, 200, 2)
    a["full200_seed"] = 7
    a["full200_fft2c"] = RU.fft2c(x)
    a["full200_ifft2c"] = RU.ifft2c(x)
    x, y = rnd(11, 2, 3, 6, 5, 2), rnd(12, 2, 1, 6, 5, 2)
    a.update(cm_x=x, cm_y=y, cm_mul=RU.complex_mul(x, y), cm_conj=RU.complex_conj(x),
             cm_abs=RU.complex_abs(x), cm_abs_sq=RU.complex_abs_sq(x),
             cm_rss=RU.rss(x, dim=1), cm_rss_complex=RU.rss_complex(x, dim=1))
    r = rnd(13, 2, 4, 5, 12)
    z = RU.real_to_complex_multi_ch(r, 6)
    a.update(mc_r=r, mc_z_re=z.real, mc_z_im=z.imag, mc_back=RU.complex_to_real_multi_ch(z))
    k = rnd(14, 1, 3, 12, 10, 2)
    a.update(mcen_x=k, mcen_out=r_tf.mask_center(k, 4, 9))
    save("ops", **a)


def g_unet():
    """ConvBlock / TransposeConvBlock / Unet / NormUnet / NormUnet3D (tiny widths)."""
    a = {}
    cb = r_unet.ConvBlock(3, 8, 0.0, 2).eval(); synth.fill_parameters_(cb, 21)
    x = rnd(21, 2, 3, 12, 10)
    a.update(sd_np(cb, "cb::")); a.update(cb_x=x, cb_y=cb(x))
    tb = r_unet.TransposeConvBlock(8, 4, 2).eval(); synth.fill_parameters_(tb, 22)
    x = rnd(22, 2, 8, 6, 5)
    a.update(sd_np(tb, "tb::")); a.update(tb_x=x, tb_y=tb(x))
    un = r_unet.Unet(chans=4, num_pool_layers=2, in_chans=2, out_chans=2).eval()
    synth.fill_parameters_(un, 23)
    x = rnd(23, 3, 2, 16, 16)
    a.update(sd_np(un, "un::")); a.update(un_x=x, un_y=un(x))
    # odd sizes exercise the up-path zero pad (unet.py:106-120)
    x = rnd(24, 2, 2, 13, 10)
    a.update(un_odd_x=x, un_odd_y=un(x))
    nu = r_norm_unet.NormUnet(4, 2).eval(); synth.fill_parameters_(nu, 25)
    x = rnd(25, 3, 1, 20, 5, 2) * 3 + 1.5
    xn, mean, std = nu.norm(nu.complex_to_chan_dim(x))
    a.update(sd_np(nu, "nu::")); a.update(nu_x=x, nu_y=nu(x), nu_norm=xn, nu_mean=mean, nu_std=std)
    nu3 = r_norm_unet.NormUnet3D(4, 2).eval(); synth.fill_parameters_(nu3, 26)
    x = rnd(26, 1, 1, 5, 12, 10, 2)
    a.update(sd_np(nu3, "nu3::")); a.update(nu3_x=x, nu3_y=nu3(x))
    save("unet", **a)


def g_varnet_block():
    """VarNetBlock pieces + SensitivityModel at tiny shape (t5 c3 24x20)."""
    t, c, h, w = 5, 3, 24, 20
    a = {}
    k = rnd(31, 1, t, c, h, w, 2)
    mask = tiny_mask(t, h)
    sens = rnd(32, 1, 1, c, h, w, 2) * 0.5
    kref = k * mask
    a.update(k=k, kref=kref, mask=mask, sens=sens)
    for dyn in ("XF", "XT", "2D", "3D"):
        net = RM.VarNet(1, 4, 2, 4, 2, dyn).eval()
        synth.fill_parameters_(net, 33)
        blk = net.cascades[0]
        blk.lambda_reg.fill_(0.3)
        a.update(sd_np(net, f"{dyn}::sd::"))
        img = blk.sens_reduce(k, sens)
        a[f"{dyn}_reduce"] = img
        a[f"{dyn}_expand"] = blk.sens_expand(img, sens)
        if dyn in ("XF", "XT"):
            a[f"{dyn}_xfyf"] = blk.xfyf_transform(img.squeeze(2))
        a[f"{dyn}_block"] = blk(k, kref, mask, sens)
    net = RM.VarNet(1, 4, 2, 4, 2, "XF").eval()
    synth.fill_parameters_(net, 34)
    a.update(sd_np(net.sens_net, "sens::sd::"))
    a["sens_out"] = net.sens_net(kref, mask)
    save("varnet_block", **a)


def g_varnet_tiny():
    """Whole VarNet forward, all dynamic types (+weight sharing), tiny shape."""
    t, c, h, w = 5, 3, 24, 20
    k = rnd(41, 1, t, c, h, w, 2)
    mask = tiny_mask(t, h)
    mk = k * mask
    a = dict(masked_kspace=mk, mask=mask)
    for tag, dyn, ws in (("XF", "XF", False), ("XT", "XT", False), ("2D", "2D", False),
                         ("3D", "3D", False), ("XFws", "XF", True)):
        net = RM.VarNet(2, 4, 2, 4, 2, dyn, ws).eval()
        synth.fill_parameters_(net, 42)
        for i, cas in enumerate(net.cascades):
            cas.lambda_reg.fill_(0.2 + 0.5 * i)
        a.update(sd_np(net, f"{tag}::sd::"))
        a[f"{tag}_out"] = net(mk, mask)
        a[f"{tag}_sens"] = net.sens_net(mk, mask)
    save("varnet_tiny", **a)


def g_masks():
    """RandomMaskFunc / EquispacedMaskFunc / apply_mask."""
    a = {}
    for seed, acc, nx, nt in ((0, 4, 200, 15), (3, 8, 200, 15), (5, 6, 64, 4)):
        np.random.seed(seed)
        mf = r_sub.create_mask_for_mask_type("random", [10], [acc])
        a[f"random_s{seed}_a{acc}_n{nx}"] = mf((nt, 1, nx, 7, 2), None)
    mf = r_sub.create_mask_for_mask_type("equispaced", [0.08], [4])
    a["equi_s9"] = mf((15, 1, 200, 7, 2), 9)
    np.random.seed(2)
    data = rnd(51, 4, 2, 32, 6, 2)
    mf = r_sub.create_mask_for_mask_type("random", [4], [4])
    md, m = r_tf.apply_mask(data, mf, None)
    a.update(am_data=data, am_masked=md, am_mask=m)
    save("masks", **a)


def g_varnet_full():
    """cfg 2 fingerprint: XF-VarNet 6 cascades, 15 coils, 15 frames, 200x200, R=4.
    Inputs/weights are regenerated from seeds (data 0, weights 1); only the
    strided output fingerprint and scalar summaries are stored."""
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=4, seed=0)
    net = RM.VarNet(6, 8, 3, 16, 3, "XF").eval()
    synth.fill_parameters_(net, 1)
    out = net(ex["masked_kspace"], ex["mask"])
    sens = net.sens_net(ex["masked_kspace"], ex["mask"])
    save("varnet_cfg2", out_strided=out[:, :, ::4, ::4].contiguous(),
         out_sum=out.double().sum(), out_l2=out.double().norm(), out_max=out.max(),
         sens_strided=sens[:, :, :, ::8, ::8].contiguous(),
         mask=ex["mask"], data_seed=0, weight_seed=1)


def g_varnet_cfg1():
    """cfg 1: 2D VarNet, 2 cascades, 8 coils, single 200x200 frame, R=4."""
    ex = synth.make_cine_slice(1, 8, 200, 200, accel=4, seed=0)
    net = RM.VarNet(2, 8, 3, 16, 3, "2D").eval()
    synth.fill_parameters_(net, 1)
    out = net(ex["masked_kspace"], ex["mask"])
    save("varnet_cfg1", out=out, out_sum=out.double().sum(), mask=ex["mask"],
         data_seed=0, weight_seed=1)


def g_cinenet():
    """CineNet blocks + whole models (reference models/cinenet.py), tiny shape."""
    t, c, h, w = 5, 3, 24, 20
    a = {}
    k = rnd(61, 1, t, c, h, w, 2)
    mask = tiny_mask(t, h)
    sens = rnd(62, 1, 1, c, h, w, 2) * 0.5
    mk = k * mask
    a.update(masked_kspace=mk, mask=mask, sens=sens)
    for tag, dyn, ws in (("XF", "XF", False), ("XT", "XT", False), ("2D", "2D", False), ("3D", "3D", False),
                         ("XFws", "XF", True)):
        net = RM.CineNet(2, 3, 4, 2, dyn, ws).eval()
        synth.fill_parameters_(net, 63)
        for i, cas in enumerate(net.cascades):
            cas.lambda_reg.fill_(0.1 + 0.4 * i)
        a.update(sd_np(net, f"{tag}::sd::"))
        a[f"{tag}_out"] = net(mk, mask, sens)
        if tag == "XF":
            blk = net.cascades[0]
            img = blk.sens_reduce(mk, sens)
            a["img"] = img
            a["H_img"] = blk.HOperator(img, mask, sens)
            a["xfyf"] = blk.xfyf_transform(img.squeeze(2))
            rhs = img + 0.7 * a["xfyf"]
            a["cg_rhs"] = rhs
            a["cg_out"] = blk.ConjGrad(a["xfyf"], rhs, mask, sens, 3)
            a["block_out"] = blk(img, img, mask, sens)
    save("cinenet", **a)


def g_xpdnet():
    """DWT / IWT / pad_for_mwcnn / MWCNN / XPDNet (reference denoisers/mwcnn.py, utils/padding.py, models/xpdnet.py)."""
    a = {}
    x = rnd(71, 2, 3, 8, 6)
    a.update(dwt_x=x, dwt_y=r_mwcnn.DWT()(x), iwt_y=r_mwcnn.IWT()(r_mwcnn.DWT()(x)))
    for tag, shape in (("p1", (2, 3, 20, 5)), ("p2", (1, 2, 15, 9)), ("p3", (1, 1, 16, 8))):
        x = rnd(72, *shape)
        y, pads = RU.pad_for_mwcnn(x, 3)
        a[f"{tag}_x"] = x; a[f"{tag}_y"] = y; a[f"{tag}_pads"] = np.array([int(p) for p in pads])
        a[f"{tag}_back"] = RU.unpad_from_mwcnn(y, pads)
    mw = r_mwcnn.MWCNN(in_chans=6, out_chans=4, n_scales=2, n_filters_per_scale=[8, 16], n_convs_per_scale=[2, 1],
                       first_conv_n_filters=8).eval()
    synth.fill_parameters_(mw, 73, keep=())
    x = rnd(73, 3, 6, 16, 8)
    a.update(sd_np(mw, "mw::")); a.update(mw_x=x, mw_y=mw(x))
    t, c, h, w = 5, 3, 24, 20
    k = rnd(74, 1, t, c, h, w, 2)
    mask = tiny_mask(t, h)
    mk = k * mask
    a.update(masked_kspace=mk, mask=mask)
    kw = dict(num_cascades=2, sens_chans=4, sens_pools=2, n_scales=2, n_filters_per_scale=[8, 16],
              n_convs_per_scale=[1, 1], first_conv_n_filters=8, n_primal=2)
    for tag, dyn, ws, po in (("XF", "XF", False, True), ("XT", "XT", False, True), ("2D", "2D", False, True),
                             ("XFws", "XF", True, True), ("XFdual", "XF", False, False)):
        net = RM.XPDNet(dynamic_type=dyn, weight_sharing=ws, primal_only=po, **kw).eval()
        synth.fill_parameters_(net, 75, keep=())
        a.update(sd_np(net, f"{tag}::sd::"))
        a[f"{tag}_out"] = net(mk, mask)
        if tag == "XF":
            a["sens_out"] = net.sens_net(mk, mask)
    save("xpdnet", **a)


def g_rnn():
    """VarNet_RNN / CineNet_RNN / XPDNet_RNN (reference models/recurrent_*.py), tiny shape."""
    t, c, h, w = 5, 3, 24, 20
    a = {}
    k = rnd(81, 1, t, c, h, w, 2)
    mask = tiny_mask(t, h)
    sens = rnd(82, 1, 1, c, h, w, 2) * 0.5
    mk = k * mask
    a.update(masked_kspace=mk, mask=mask, sens=sens)
    net = RM.VarNet_RNN(3, 4, 2, 6).eval(); synth.fill_parameters_(net, 83); net.lambda_reg.fill_(0.4)
    a.update(sd_np(net, "varnet_rnn::sd::")); a["varnet_rnn_out"] = net(mk, mask)
    net = RM.CineNet_RNN(3, 3, 6).eval(); synth.fill_parameters_(net, 84); net.lambda_reg.fill_(0.4)
    a.update(sd_np(net, "cinenet_rnn::sd::")); a["cinenet_rnn_out"] = net(mk, mask, sens)
    net = RM.XPDNet_RNN(3, 4, 2, 6, True, 2, 1).eval(); synth.fill_parameters_(net, 85, keep=())
    a.update(sd_np(net, "xpdnet_rnn::sd::")); a["xpdnet_rnn_out"] = net(mk, mask)
    save("rnn", **a)


def _fingerprint(name, out, ex, **extra):
    """Full-size model fingerprint: strided output + scalar summaries (inputs / weights regenerate from seeds)."""
    save(name, out_strided=out[:, :, ::4, ::4].contiguous(), out_sum=out.double().sum(), out_l2=out.double().norm(),
         out_max=out.max(), mask=ex["mask"], **extra)


def g_xpdnet_cfg3():
    """cfg 3: XT-XPDNet, MWCNN regulariser (script defaults), 10 cascades, n_primal 5, 15 coils x 15 frames x 200x200, R=8
    (reference models/xpdnet.py:301-326; widths traintest_scripts/xpdnet/train_test_xpdnet.py:258-271).

    k-space carries white noise (std 0.01): on the noise-free phantom the y-f planes of empty image columns hold only
    rounding noise, which the MWCNN's InstanceNorm layers (no input normalisation, no residual) amplify to O(1).
    An untrained 10-cascade XPDNet is also expansive (a relative perturbation roughly doubles per cascade), so besides the
    reference's fp32 output the file holds how far that output is from the same network in fp64 (build oracle, pinned to the
    reference): `ref_fp32_vs_fp64`, the reproducibility floor any fp32 implementation has on this configuration, and the
    fp64 fingerprint itself."""
    from oracle import xpdnet_ref as X
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=8, seed=5, noise_std=0.01)
    kw = dict(num_cascades=10, sens_chans=8, sens_pools=3, n_primal=5, dynamic_type="XT")
    net = RM.XPDNet(**kw).eval()
    synth.fill_parameters_(net, 6, keep=())
    out = net(ex["masked_kspace"], ex["mask"])
    o64 = X.XPDNet(**kw).double().eval()
    o64.load_state_dict({k: v.double() for k, v in net.state_dict().items()}, strict=True)
    out64 = o64(ex["masked_kspace"].double(), ex["mask"])
    gap = float((out.double() - out64).abs().max() / out64.abs().max())
    nmse = float(((out.double() - out64) ** 2).sum() / (out64 ** 2).sum())
    print(f"  reference fp32 vs fp64: max|d|/peak {gap:.3e}, NMSE {nmse:.3e}")
    _fingerprint("xpdnet_cfg3", out, ex, data_seed=5, weight_seed=6, accel=8, noise_std=0.01,
                 out64_strided=out64[:, :, ::4, ::4].float().contiguous(), ref_fp32_vs_fp64=gap, ref_fp32_vs_fp64_nmse=nmse)


def g_cinenet_cfg4():
    """cfg 4: 3D CineNet, 6 cascades, CG 6, chans 16, pools 3, analytic sens maps, R=6 (reference models/cinenet.py:61-73)."""
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=6, seed=4)
    net = RM.CineNet(6, 6, 16, 3, "3D").eval()
    synth.fill_parameters_(net, 7)
    out = net(ex["masked_kspace"], ex["mask"], ex["sens_maps"])
    _fingerprint("cinenet_cfg4", out, ex, data_seed=4, weight_seed=7, accel=6)


def g_rnn_cfg5():
    """cfg 5: CRNN-VarNet, 5 cascades, sens 8/3, chans 16, R=8 (reference models/recurrent_varnet.py:93-150)."""
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=8, seed=8)
    net = RM.VarNet_RNN(5, 8, 3, 16).eval()
    synth.fill_parameters_(net, 9)
    out = net(ex["masked_kspace"], ex["mask"])
    _fingerprint("rnn_cfg5", out, ex, data_seed=8, weight_seed=9, accel=8)


def g_metrics():
    """SSIMLoss (reference utils/losses.py:6-58) on a (1, 1, 15, 180, 180) pair; inputs regenerate from seeds.  The
    reference hard-wires `.to('cuda')` (:34): Tensor.to is pinned to the CPU for this call.  Also the zero-filled
    reconstruction of traintest_scripts/run_inference.py:64-67 (ifft2c norm=None, rss_complex) on a small k-space."""
    from reconstruction.utils.losses import SSIMLoss
    orig_to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: self if (a and a[0] == "cuda") else orig_to(self, *a, **k)
    try:
        rs = np.random.RandomState(91)
        tgt = torch.from_numpy(rs.uniform(0, 1.5, size=(15, 180, 180)).astype(np.float32))
        rec = (tgt + torch.from_numpy((0.1 * rs.standard_normal((15, 180, 180))).astype(np.float32))).clamp_min(0)
        loss = SSIMLoss()(rec[None, None], tgt[None, None], torch.tensor([1.0]))
        frames = [float(SSIMLoss()(rec[None, None, t:t + 1], tgt[None, None, t:t + 1], torch.tensor([1.0]))) for t in range(15)]
    finally:
        torch.Tensor.to = orig_to
    k = rnd(92, 1, 3, 4, 24, 20, 2)
    scaling = torch.sqrt(torch.prod(torch.as_tensor(k.shape[-3:-1])))
    zf = RU.rss_complex(RU.ifft2c(k, norm=None) * scaling, dim=2)
    save("metrics", seed=91, ssim_loss=loss, ssim_loss_frames=np.array(frames), zf_k=k, zf_out=zf,
         ifft2c_none=RU.ifft2c(k, norm=None), fft2c_none=RU.fft2c(k, norm=None))


def g_frontend():
    """The reference's data front-end (data/mri_data.py:283-303) on a small seeded raw k-space (t=7, x=40, y=36, c=3):
    the numpy lines of SliceDataset.__getitem__ with the reference's own filtered_crop_center_and_slices / center_crop
    (data/transforms.py:186-220, 136-158; scipy.ndimage.gaussian_filter underneath).  h5py and bart.ecalib are not part of
    the vector: the raw array is generated here and the sensitivity maps are a seeded input."""
    rs = np.random.RandomState(77)
    nt, nx, ny, nc = 7, 40, 36, 3
    raw = (rs.standard_normal((nt, nx, ny, nc)) + 1j * rs.standard_normal((nt, nx, ny, nc))).astype(np.complex64) * 1e-6
    scaling, crop_shape, crop_target, n_slices, filter_size = 1e6, (24, 20), (20, 16), 5, [0.7, 0., 0.3, 0.3]
    ax = (-2, -1)

    def unnormalised(fn, a):                              # the reference's shift / transform / shift order around numpy's norm=None transforms
        return np.fft.fftshift(fn(np.fft.ifftshift(a, axes=ax), axes=ax, norm=None), axes=ax) if fn is np.fft.ifftn else \
            np.fft.ifftshift(fn(np.fft.fftshift(a, axes=ax), axes=ax, norm=None), axes=ax)

    coilfirst = (np.array(raw, dtype="complex64") * scaling).transpose(0, 3, 1, 2)            # (t, c, x, y)
    images = unnormalised(np.fft.ifftn, coilfirst) * np.sqrt(np.prod(coilfirst.shape[-2:]))  # mri_data.py:288-289
    images_cropped, images_filter = r_tf.filtered_crop_center_and_slices(images, crop_shape, n_slices, filter_size)   # :290
    k2 = unnormalised(np.fft.fftn, images_filter) / np.sqrt(np.prod(images_filter.shape[-2:]))                     # :291-292
    k2 = k2.transpose(0, 2, 3, 1).astype("complex64")                                                              # :293
    sens = (rs.standard_normal((nc,) + crop_shape) + 1j * rs.standard_normal((nc,) + crop_shape)).astype(np.complex64)
    target = r_tf.center_crop(np.abs(np.sum(images_filter * np.conjugate(sens[None]), axis=1)).astype("float32"), crop_target)   # :302-303
    time_avg = np.mean(k2, axis=0, keepdims=True)
    save("frontend", raw=raw, sens=sens, crop_shape=np.array(crop_shape), crop_target=np.array(crop_target), n_slices=n_slices,
         filter_size=np.array(filter_size), images_cropped=images_cropped.astype(np.complex64),
         images_filter=images_filter.astype(np.complex64), kspace=k2.transpose(0, 3, 1, 2), target=target,
         time_avg_kspace=time_avg[0].transpose(2, 0, 1).astype(np.complex64))


def _training_step(net, mk, mask, target, lr=0.0003, dtype=None, extra=()):
    """The body of reference pl_modules/varnet_module.py:97-113 (forward, center_crop_to_smallest, SSIMLoss) followed by
    loss.backward() and one step of the optimiser of :151-154 (Adam, lr 0.0003, weight_decay 0).  Returns (loss, {name: grad},
    {name: updated weight}, output).  dtype float64 re-runs the same reference code in double precision (the reproducibility
    floor of any float32 implementation)."""
    from reconstruction.utils.losses import SSIMLoss
    orig_to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: self if (a and a[0] == "cuda") else orig_to(self, *a, **k)     # losses.py:34 hard-wires 'cuda'
    orig_default = torch.get_default_dtype()
    if dtype is not None:
        torch.set_default_dtype(dtype)       # the CRNN models create their zero hidden states with torch.zeros(size) (recurrent_varnet.py:112, 236)
    try:
        with torch.enable_grad():
            if dtype is not None:
                net = net.to(dtype); mk = mk.to(dtype); target = target.to(dtype); extra = tuple(e.to(dtype) for e in extra)
            lossf = SSIMLoss()
            if dtype is not None:
                lossf = lossf.to(dtype)
            opt = torch.optim.Adam(net.parameters(), lr=lr, weight_decay=0.0)
            opt.zero_grad()
            output = net(mk, mask, *extra)
            tgt, out = r_tf.center_crop_to_smallest(target, output)
            if dtype is not None:       # losses.py:34 builds the data range with torch.Tensor(...) (float32): keep the module's dtype
                loss = _ssim_loss_any_dtype(lossf, out.unsqueeze(1), tgt.unsqueeze(1))
            else:
                loss = lossf(out.unsqueeze(1), tgt.unsqueeze(1), data_range=tgt.max())
            loss.backward()
            grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
            opt.step()
            new = {k: p.detach().clone() for k, p in net.named_parameters()}
    finally:
        torch.Tensor.to = orig_to
        torch.set_default_dtype(orig_default)
    return loss.detach(), grads, new, output.detach()


def _ssim_loss_any_dtype(lossf, Xt, Yt):
    """reference utils/losses.py:25-58 line for line, with the frame's data range kept in the inputs' dtype (float64 runs)."""
    import torch.nn.functional as F
    ssims = 0.
    Nt = Xt.shape[2]
    for t in range(Nt):
        X = Xt[:, :, t, :]; Y = Yt[:, :, t, :]
        data_range = Y.max().reshape(1)[:, None, None, None]
        C1 = (lossf.k1 * data_range) ** 2; C2 = (lossf.k2 * data_range) ** 2
        ux = F.conv2d(X, lossf.w); uy = F.conv2d(Y, lossf.w)
        uxx = F.conv2d(X * X, lossf.w); uyy = F.conv2d(Y * Y, lossf.w); uxy = F.conv2d(X * Y, lossf.w)
        vx = lossf.cov_norm * (uxx - ux * ux); vy = lossf.cov_norm * (uyy - uy * uy); vxy = lossf.cov_norm * (uxy - ux * uy)
        A1, A2, B1, B2 = (2 * ux * uy + C1, 2 * vxy + C2, ux ** 2 + uy ** 2 + C1, vx + vy + C2)
        S = (A1 * A2) / (B1 * B2)
        ssims += 1 - S.mean()
    return ssims / Nt


def _kink_stability(net, mk, mask, target, trials=6, extra=()):
    """LeakyReLU has a kink at 0 and InstanceNorm planes whose mean is zero up to rounding (the first conv of every NormUnet on a
    zero-padded, mean-normalised plane) put whole groups of activations within rounding of it: the reference's OWN float32
    gradient then jumps by ~1e-3 when the input changes by 1e-6.  A fixture can pin an implementation only where that does not
    happen: largest relative change of any parameter gradient of the reference's training step under `trials` random relative
    input perturbations of 1e-6."""
    import copy
    base = _training_step(copy.deepcopy(net), mk, mask, target, extra=extra)[1]
    worst = 0.0
    for i in range(trials):
        pert = mk * (1 + 1e-6 * rnd(900 + i, *mk.shape))
        gi = _training_step(copy.deepcopy(net), pert, mask, target, extra=extra)[1]
        worst = max(worst, max(float((gi[k] - base[k]).abs().max() / base[k].abs().max().clamp_min(1e-30)) for k in base))
    return worst


def g_varnet_grad():
    """Gradients of the reference's training step (pl_modules/varnet_module.py:97-113 + loss.backward() + one Adam step,
    :151-154) for the tiny VarNets: XF, XT, 2D, XF with weight sharing.  Stored per variant: state dict, loss, every
    parameter's gradient, the weights after the step, and -- from the same reference code run in float64 -- the gradients'
    float32 reproducibility floor (max |g32 - g64| / max |g64| per parameter)."""
    import copy
    t, c, h, w = 5, 3, 24, 20
    mask = tiny_mask(t, h)
    mk = rnd(44, 1, t, c, h, w, 2) * mask          # white k-space: every x-f / y-f plane carries signal (well-conditioned InstanceNorms)
    target = rnd(45, 1, t, 20, 18).abs() + 0.1     # (1, t, 20, 18): exercises the center crop
    a = dict(masked_kspace=mk, mask=mask, target=target)
    for tag, dyn, ws in (("XF", "XF", False), ("XT", "XT", False), ("2D", "2D", False), ("XFws", "XF", True), ("3D", "3D", False)):
        # the weight seed is the first whose reference gradients are stable under 1e-6 input perturbations (see _kink_stability):
        # about one fixture in three sits on a LeakyReLU kink and cannot pin any float32 implementation, the reference included
        for seed in range(43, 143):
            net = RM.VarNet(2, 4, 2, 4, 2, dyn, ws)
            synth.fill_parameters_(net, seed)
            with torch.no_grad():
                for i, cas in enumerate(net.cascades):
                    cas.lambda_reg.fill_(0.2 + 0.5 * i)
            stab = _kink_stability(net, mk, mask, target)
            print(f"    {tag}: weight seed {seed}: gradient change under 1e-6 input perturbations {stab:.2e}")
            if stab <= 2e-5:
                break
        else:
            raise RuntimeError("no kink-stable seed")
        a[f"{tag}_seed"] = seed; a[f"{tag}_stability"] = stab
        a.update(sd_np(copy.deepcopy(net), f"{tag}::sd::"))      # a copy: the optimiser step below updates the parameters in place
        net64 = copy.deepcopy(net)
        loss, grads, new, out = _training_step(net, mk, mask, target)
        loss64, grads64, _, _ = _training_step(net64, mk, mask, target, dtype=torch.float64)
        a[f"{tag}_loss"] = loss; a[f"{tag}_out"] = out; a[f"{tag}_loss64"] = loss64
        for k, g in grads.items():
            a[f"{tag}::grad::{k}"] = g
            a[f"{tag}::new::{k}"] = new[k]
            g64 = grads64[k]
            a[f"{tag}::floor::{k}"] = float((g.double() - g64).abs().max() / g64.abs().max().clamp_min(1e-300))
    save("varnet_grad", **a)


def g_varnet_grad_cfg2():
    """cfg 2 (XF-VarNet, 6 cascades, 15 coils x 15 frames x 200 x 200): strided fingerprints of the reference's parameter
    gradients of the training step (inputs / weights regenerate from seeds 0 / 1; target = the synthetic phantom's target)."""
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=4, seed=0)
    net = RM.VarNet(6, 8, 3, 16, 3, "XF")
    synth.fill_parameters_(net, 1)
    target = ex["target"].contiguous()
    import copy
    net0 = copy.deepcopy(net)
    loss, grads, new, out = _training_step(net, ex["masked_kspace"], ex["mask"], target)
    a = dict(loss=loss, data_seed=0, weight_seed=1, out_strided=out[:, :, ::4, ::4].contiguous())
    # the reference's own float32 gradients move when the input changes by 1e-6 (LeakyReLU kinks, see _kink_stability): that
    # movement, per parameter, is the resolution at which ANY float32 implementation can be compared with this fixture
    self_max, self_norm = {k: 0.0 for k in grads}, {k: 0.0 for k in grads}
    for i in range(2):
        gi = _training_step(copy.deepcopy(net0), ex["masked_kspace"] * (1 + 1e-6 * rnd(950 + i, *ex["masked_kspace"].shape)), ex["mask"], target)[1]
        for k, g in grads.items():
            self_max[k] = max(self_max[k], float((gi[k] - g).abs().max() / g.abs().max()))
            self_norm[k] = max(self_norm[k], float((gi[k] - g).double().norm() / g.double().norm()))
    for k, g in grads.items():
        flat = g.reshape(-1)
        a[f"grad::{k}"] = flat[::max(1, flat.numel() // 512)].contiguous()
        a[f"gnorm::{k}"] = g.double().norm()
        a[f"gmax::{k}"] = g.abs().max()
        a[f"selfmax::{k}"] = self_max[k]; a[f"selfnorm::{k}"] = self_norm[k]
    save("varnet_grad_cfg2", **a)


def g_lightning():
    """Lightning-style checkpoints of the reference's pl_modules (VarNetModule / CineNetModule / XPDNetModule: the state-dict keys
    carry the `varnet.` / `cinenet.` / `xpdnet.` prefixes plus the loss window and the metric accumulators) for the tiny models,
    and what the lines of traintest_scripts/run_inference.py:53-78 (InferenceTransform.forward) make of one slice: target, output
    and zero-filled reconstruction after the center crops.  pytorch_lightning / torchmetrics / skimage are absent from the image:
    import-time stand-ins (a LightningModule is an nn.Module here; nothing of them is on the path)."""
    import pathlib
    for name in ("torchmetrics", "skimage", "skimage.metrics"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["skimage.metrics"].peak_signal_noise_ratio = None
    sys.modules["skimage.metrics"].structural_similarity = None
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k): pass

    class Metric(torch.nn.Module):
        def __init__(self, dist_sync_on_step=True): super().__init__()
        def add_state(self, name, default, dist_reduce_fx=None): self.register_buffer(name, default)
    pl.LightningModule, pl.LightningDataModule = LightningModule, type("LightningDataModule", (), {})
    pl.metrics = types.ModuleType("pytorch_lightning.metrics"); pl.metrics.Metric = Metric
    sys.modules["pytorch_lightning"] = pl; sys.modules["pytorch_lightning.metrics"] = pl.metrics
    import reconstruction.pl_modules as P
    import reconstruction.pl_modules.mri_module as MM
    MM.fetch_dir = lambda key, cfg=None: pathlib.Path("/tmp")      # MriModule.__init__ writes a yaml next to the scripts otherwise
    t, c, h, w = 5, 3, 24, 20
    ex = synth.make_cine_slice(t, c, h, w, accel=4, center_lines=4, seed=11, noise_std=0.01)
    mk, mask, sens = ex["masked_kspace"], ex["mask"], ex["sens_maps"]
    target = ex["target"][:, :, 2:-2, 1:-1].contiguous()
    a = dict(masked_kspace=mk, mask=mask, sens_maps=sens, target=target)
    mods = dict(varnet=P.VarNetModule(num_cascades=2, pools=2, chans=4, sens_pools=2, sens_chans=4, dynamic_type="XF"),
                cinenet=P.CineNetModule(num_cascades=2, CG_iters=3, pools=2, chans=4, dynamic_type="XF"),
                xpdnet=P.XPDNetModule(num_cascades=2, sens_chans=4, sens_pools=2, n_scales=2, n_filters_per_scale=[8, 16],
                                      n_convs_per_scale=[1, 1], first_conv_n_filters=8, n_primal=2, dynamic_type="XF"))
    for kind, mod in mods.items():
        mod.eval()
        synth.fill_parameters_(getattr(mod, kind), 61, keep=("lambda",) if kind != "xpdnet" else ())
        a.update(sd_np(mod, f"{kind}::ckpt::"))
        # ---- run_inference.py:53-78 (device = cpu here)
        if kind == "cinenet":
            output = mod(mk, mask, sens)
        else:
            output = mod(mk, mask)
        scaling_factor = torch.sqrt(torch.prod(torch.as_tensor(mk.shape[-3:-1])))
        images = RU.ifft2c(mk, norm=None) * scaling_factor
        zero_filled = RU.rss_complex(images, dim=2)
        tgt, output = r_tf.center_crop_to_smallest(target, output)
        tgt, zero_filled = r_tf.center_crop_to_smallest(tgt, zero_filled)
        a[f"{kind}_target"] = tgt.numpy().astype("float32")[0]
        a[f"{kind}_output"] = output.numpy().astype("float32")[0]
        a[f"{kind}_zero_filled"] = zero_filled.numpy().astype("float32")[0]
        print("   ", kind, len(mod.state_dict()), "keys; non-model:", [k for k in mod.state_dict() if not k.startswith(kind + ".")])
    save("lightning_ckpt", **a)


def g_cinenet_grad():
    """Gradients of the reference's training step for the tiny CineNets (pl_modules/cinenet_module.py:98-114: forward(masked_kspace,
    mask, sens_maps) + SSIMLoss; conjugate gradient with detached step sizes, cinenet.py:159-169): XF, XT, 2D, XF with weight sharing.
    Same contents and the same kink-stable seed selection as varnet_grad.npz."""
    import copy
    t, c, h, w = 5, 3, 24, 20
    mask = tiny_mask(t, h)
    mk = rnd(64, 1, t, c, h, w, 2) * mask
    sens = rnd(66, 1, 1, c, h, w, 2)
    sens = sens / RU.rss_complex(sens, dim=2).unsqueeze(-1).unsqueeze(2)
    target = rnd(65, 1, t, 20, 18).abs() + 0.1
    a = dict(masked_kspace=mk, mask=mask, target=target, sens_maps=sens)
    for tag, dyn, ws in (("XF", "XF", False), ("XT", "XT", False), ("2D", "2D", False), ("XFws", "XF", True), ("3D", "3D", False)):
        for seed in range(43, 143):
            net = RM.CineNet(2, 3, 4, 2, dyn, ws)
            synth.fill_parameters_(net, seed)
            with torch.no_grad():
                for i, cas in enumerate(net.cascades):
                    cas.lambda_reg.fill_(0.2 + 0.5 * i)
            stab = _kink_stability(net, mk, mask, target, extra=(sens,))
            print(f"    {tag}: weight seed {seed}: gradient change under 1e-6 input perturbations {stab:.2e}")
            if stab <= 2e-5:
                break
        else:
            raise RuntimeError("no kink-stable seed")
        a[f"{tag}_seed"] = seed; a[f"{tag}_stability"] = stab
        a.update(sd_np(copy.deepcopy(net), f"{tag}::sd::"))
        net64 = copy.deepcopy(net)
        loss, grads, new, out = _training_step(net, mk, mask, target, extra=(sens,))
        loss64, grads64, _, _ = _training_step(net64, mk, mask, target, dtype=torch.float64, extra=(sens,))
        a[f"{tag}_loss"] = loss; a[f"{tag}_out"] = out; a[f"{tag}_loss64"] = loss64
        for k, g in grads.items():
            a[f"{tag}::grad::{k}"] = g
            a[f"{tag}::new::{k}"] = new[k]
            g64 = grads64[k]
            a[f"{tag}::floor::{k}"] = float((g.double() - g64).abs().max() / g64.abs().max().clamp_min(1e-300))
    save("cinenet_grad", **a)


def g_xpdnet_grad():
    """Gradients of the reference's training step for the tiny primal-only XPDNets (pl_modules/xpdnet_module.py: forward(masked_kspace,
    mask) + SSIMLoss; sensitivity network, K step + masked backward operator, I-step MWCNNs on x-f / y-f planes): XF, XT, XF with weight
    sharing.  Same contents and the same kink-stable seed selection as varnet_grad.npz."""
    import copy
    t, c, h, w = 5, 3, 24, 20
    mask = tiny_mask(t, h)
    mk = rnd(84, 1, t, c, h, w, 2) * mask
    target = rnd(85, 1, t, 20, 18).abs() + 0.1
    a = dict(masked_kspace=mk, mask=mask, target=target)
    kw = dict(num_cascades=2, sens_chans=4, sens_pools=2, n_scales=2, n_filters_per_scale=[8, 16],
              n_convs_per_scale=[2, 1], first_conv_n_filters=8, n_primal=2)
    for tag, dyn, ws in (("XF", "XF", False), ("XT", "XT", False), ("XFws", "XF", True), ("2D", "2D", False), ("XFdual", "XF", False)):
        for seed in range(43, 143):
            net = RM.XPDNet(dynamic_type=dyn, weight_sharing=ws, primal_only=tag != "XFdual", **kw)
            synth.fill_parameters_(net, seed, keep=())
            stab = _kink_stability(net, mk, mask, target)
            print(f"    {tag}: weight seed {seed}: gradient change under 1e-6 input perturbations {stab:.2e}")
            if stab <= 2e-5:
                break
        else:
            raise RuntimeError("no kink-stable seed")
        a[f"{tag}_seed"] = seed; a[f"{tag}_stability"] = stab
        a.update(sd_np(copy.deepcopy(net), f"{tag}::sd::"))
        net64 = copy.deepcopy(net)
        loss, grads, new, out = _training_step(net, mk, mask, target)
        loss64, grads64, _, _ = _training_step(net64, mk, mask, target, dtype=torch.float64)
        a[f"{tag}_loss"] = loss; a[f"{tag}_out"] = out; a[f"{tag}_loss64"] = loss64
        for k, g in grads.items():
            a[f"{tag}::grad::{k}"] = g
            a[f"{tag}::new::{k}"] = new[k]
            g64 = grads64[k]
            a[f"{tag}::floor::{k}"] = float((g.double() - g64).abs().max() / g64.abs().max().clamp_min(1e-300))
    save("xpdnet_grad", **a)


def g_rnn_grad():
    """Gradients of the reference's training step for the tiny convolutional-RNN hybrids (models/recurrent_*.py: BCRNN over time, hidden
    states carried across cascades, weights shared by all cascades): VarNet_RNN, CineNet_RNN, XPDNet_RNN.  Same contents and the same
    kink-stable seed selection as varnet_grad.npz (the sensitivity U-Nets have LeakyReLU kinks; the CRNN's ReLU has its own at 0)."""
    import copy
    t, c, h, w = 5, 3, 24, 20
    mask = tiny_mask(t, h)
    mk = rnd(94, 1, t, c, h, w, 2) * mask
    sens = rnd(96, 1, 1, c, h, w, 2)
    sens = sens / RU.rss_complex(sens, dim=2).unsqueeze(-1).unsqueeze(2)
    target = rnd(95, 1, t, 20, 18).abs() + 0.1
    a = dict(masked_kspace=mk, mask=mask, target=target, sens_maps=sens)
    for tag, make, extra, keep in (("varnet_rnn", lambda: RM.VarNet_RNN(3, 4, 2, 6), (), ("lambda",)),
                                   ("cinenet_rnn", lambda: RM.CineNet_RNN(3, 3, 6), (sens,), ("lambda",)),
                                   ("xpdnet_rnn", lambda: RM.XPDNet_RNN(3, 4, 2, 6, True, 2, 1), (), ()),
                                   ("xpdnet_rnn_dual", lambda: RM.XPDNet_RNN(3, 4, 2, 6, False, 2, 1), (), ())):
        for seed in range(43, 143):
            net = make()
            synth.fill_parameters_(net, seed, keep=keep)
            if keep:
                with torch.no_grad():
                    net.lambda_reg.fill_(0.4)
            stab = _kink_stability(net, mk, mask, target, extra=extra)
            print(f"    {tag}: weight seed {seed}: gradient change under 1e-6 input perturbations {stab:.2e}")
            if stab <= 2e-5:
                break
        else:
            raise RuntimeError("no kink-stable seed")
        a[f"{tag}_seed"] = seed; a[f"{tag}_stability"] = stab
        a.update(sd_np(copy.deepcopy(net), f"{tag}::sd::"))
        net64 = copy.deepcopy(net)
        loss, grads, new, out = _training_step(net, mk, mask, target, extra=extra)
        loss64, grads64, _, _ = _training_step(net64, mk, mask, target, dtype=torch.float64, extra=extra)
        a[f"{tag}_loss"] = loss; a[f"{tag}_out"] = out; a[f"{tag}_loss64"] = loss64
        for k, g in grads.items():
            a[f"{tag}::grad::{k}"] = g
            a[f"{tag}::new::{k}"] = new[k]
            g64 = grads64[k]
            a[f"{tag}::floor::{k}"] = float((g.double() - g64).abs().max() / g64.abs().max().clamp_min(1e-300))
    save("rnn_grad", **a)


def _grad_fingerprint(name, net, ex, extra=(), noise_seed=950, nper=2):
    """Strided fingerprints of the reference's parameter gradients at a full BASELINE shape + the reference's own movement under 1e-6
    input perturbations (see g_varnet_grad_cfg2)."""
    import copy
    target = ex["target"].contiguous()
    net0 = copy.deepcopy(net)
    mk = ex["masked_kspace"]
    loss, grads, new, out = _training_step(net, mk, ex["mask"], target, extra=extra)
    a = dict(loss=loss, out_strided=out[:, :, ::4, ::4].contiguous())
    self_max, self_norm = {k: 0.0 for k in grads}, {k: 0.0 for k in grads}
    for i in range(nper):
        gi = _training_step(copy.deepcopy(net0), mk * (1 + 1e-6 * rnd(noise_seed + i, *mk.shape)), ex["mask"], target, extra=extra)[1]
        for k, g in grads.items():
            self_max[k] = max(self_max[k], float((gi[k] - g).abs().max() / g.abs().max().clamp_min(1e-30)))
            self_norm[k] = max(self_norm[k], float((gi[k] - g).double().norm() / g.double().norm().clamp_min(1e-30)))
    for k, g in grads.items():
        flat = g.reshape(-1)
        a[f"grad::{k}"] = flat[::max(1, flat.numel() // 256)].contiguous()
        a[f"gnorm::{k}"] = g.double().norm()
        a[f"gmax::{k}"] = g.abs().max()
        a[f"selfmax::{k}"] = self_max[k]; a[f"selfnorm::{k}"] = self_norm[k]
    save(name, **a)


def g_xpdnet_grad_cfg3():
    """cfg 3 (XT-XPDNet, 10 cascades, MWCNN defaults, n_primal 5, R = 8, noise 0.01): gradient fingerprints of the reference's training step."""
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=8, seed=0, noise_std=0.01)
    net = RM.XPDNet(num_cascades=10, sens_chans=8, sens_pools=3, n_primal=5, dynamic_type="XT")
    synth.fill_parameters_(net, 6, keep=())
    _grad_fingerprint("xpdnet_grad_cfg3", net, ex, nper=1)


def g_cinenet_grad_cfg4():
    """cfg 4 (3D CineNet, 6 cascades, CG 6, R = 6, analytic sensitivity maps): gradient fingerprints of the reference's training step."""
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=6, seed=0)
    net = RM.CineNet(6, 6, 16, 3, "3D")
    synth.fill_parameters_(net, 7)
    _grad_fingerprint("cinenet_grad_cfg4", net, ex, extra=(ex["sens_maps"],), nper=4)


def g_rnn_grad_cfg5():
    """cfg 5 (CRNN-VarNet, 5 cascades, 16 hidden channels, R = 8): gradient fingerprints of the reference's training step."""
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=8, seed=0)
    net = RM.VarNet_RNN(5, 8, 3, 16)
    synth.fill_parameters_(net, 9)
    _grad_fingerprint("rnn_grad_cfg5", net, ex, nper=2)


def _linear_activation():
    """Context: LeakyReLU -> identity inside the reference (nn.LeakyReLU.forward calls F.leaky_relu): the networks keep their InstanceNorms
    but lose their kinks, so float32 gradients are smooth functions of the input and a full-size comparison is sharp."""
    import contextlib
    import torch.nn.functional as F

    @contextlib.contextmanager
    def cm():
        orig, orig_relu = F.leaky_relu, F.relu
        F.leaky_relu = lambda x, *a, **k: x
        F.relu = lambda x, *a, **k: x          # nn.ReLU.forward calls F.relu: the CRNN cells (recurrent_varnet.py:126-134,198)
        try:
            yield
        finally:
            F.leaky_relu, F.relu = orig, orig_relu
    return cm()


def _grad_fingerprint_linear(name, make, ex, extra=()):
    """Full-size gradient fingerprints with the LeakyReLUs replaced by the identity (the HIP path: slope 1 / ReLU off as per-call
    arguments): the reference's float32 gradients, its float64 gradients, and their distance (the float32 floor) per parameter."""
    target = ex["target"].contiguous()
    mk = ex["masked_kspace"]
    with _linear_activation():
        loss, grads, new, out = _training_step(make(), mk, ex["mask"], target, extra=extra)
        loss64, grads64, _, _ = _training_step(make(), mk, ex["mask"], target, dtype=torch.float64, extra=extra)
    a = dict(loss=loss, loss64=loss64, out_strided=out[:, :, ::4, ::4].contiguous())
    for k, g in grads.items():
        g64 = grads64[k]
        st = max(1, g.numel() // 256)
        a[f"grad::{k}"] = g.reshape(-1)[::st].contiguous()
        a[f"grad64::{k}"] = g64.reshape(-1)[::st].contiguous()
        a[f"gmax::{k}"] = g64.abs().max()
        a[f"gnorm::{k}"] = g64.norm()
        a[f"floormax::{k}"] = float((g.double() - g64).abs().max() / g64.abs().max().clamp_min(1e-300))
        a[f"floornorm::{k}"] = float((g.double() - g64).norm() / g64.norm().clamp_min(1e-300))
    save(name, **a)


def g_varnet_grad_cfg2_linear():
    """cfg 2 with identity activations: see _grad_fingerprint_linear."""
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=4, seed=0)

    def make():
        net = RM.VarNet(6, 8, 3, 16, 3, "XF")
        synth.fill_parameters_(net, 1)
        return net
    _grad_fingerprint_linear("varnet_grad_cfg2_linear", make, ex)


def g_cinenet_grad_cfg4_linear():
    """cfg 4 (3-D U-Net: LeakyReLU only) with identity activations."""
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=6, seed=0)

    def make():
        net = RM.CineNet(6, 6, 16, 3, "3D")
        synth.fill_parameters_(net, 7)
        return net
    _grad_fingerprint_linear("cinenet_grad_cfg4_linear", make, ex, extra=(ex["sens_maps"],))



def g_xpdnet_grad_cfg3_linear():
    """cfg 3 (XT-XPDNet, 10 cascades; MWCNN conv blocks = conv + InstanceNorm + LeakyReLU, mwcnn.py:199-208) with identity activations."""
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=8, seed=0, noise_std=0.01)

    def make():
        net = RM.XPDNet(num_cascades=10, sens_chans=8, sens_pools=3, n_primal=5, dynamic_type="XT")
        synth.fill_parameters_(net, 6, keep=())
        return net
    _grad_fingerprint_linear("xpdnet_grad_cfg3_linear", make, ex)


def g_rnn_grad_cfg5_linear():
    """cfg 5 (CRNN-VarNet, 5 cascades, 15-frame bidirectional time sweeps) with identity activations: nn.ReLU of the cells and conv blocks
    (recurrent_varnet.py:126-134,198) and the sensitivity U-Net's LeakyReLU."""
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=8, seed=0)

    def make():
        net = RM.VarNet_RNN(5, 8, 3, 16)
        synth.fill_parameters_(net, 9)
        return net
    _grad_fingerprint_linear("rnn_grad_cfg5_linear", make, ex)


def g_fft_smooth():
    """fftc.py:13-117 on the lengths the mixed-radix line engine serves (2^a 3^b 5^c, not 200): inputs regenerated from the seed,
    outputs stored whole for the small planes and on a strided lattice for the large ones (every sample of a transform depends on
    every input sample, so a lattice pins the whole transform)."""
    a = {}
    shapes = dict(a96x120=(2, 96, 120), a192x160=(1, 192, 160), a256x320=(1, 256, 320), a384x512=(1, 384, 512), a45x250=(2, 45, 250),
                  a400x405=(1, 400, 405))
    for i, (tag, (n, h, w)) in enumerate(shapes.items()):
        x = rnd(900 + i, n, h, w, 2)
        sh, sw = (1, 1) if h * w <= 192 * 160 else (5, 7)
        a[f"{tag}_seed"] = 900 + i
        a[f"{tag}_shape"] = np.array([n, h, w])
        a[f"{tag}_stride"] = np.array([sh, sw])
        a[f"{tag}_fft2c"] = RU.fft2c(x)[:, ::sh, ::sw].contiguous()
        a[f"{tag}_ifft2c"] = RU.ifft2c(x)[:, ::sh, ::sw].contiguous()
    for i, n in enumerate((30, 128, 360, 512)):      # fft1c along the second-to-last axis (the XF transform's axis)
        x = rnd(950 + i, 3, n, 2)
        a[f"l{n}_seed"] = 950 + i
        a[f"l{n}_fft1c"] = RU.fft1c(x)
        a[f"l{n}_ifft1c"] = RU.ifft1c(x)
    save("fft_smooth", **a)


GENERATORS = dict(xpdnet_grad_cfg3_linear=g_xpdnet_grad_cfg3_linear, rnn_grad_cfg5_linear=g_rnn_grad_cfg5_linear, varnet_grad_cfg2_linear=g_varnet_grad_cfg2_linear, cinenet_grad_cfg4_linear=g_cinenet_grad_cfg4_linear, xpdnet_grad_cfg3=g_xpdnet_grad_cfg3, cinenet_grad_cfg4=g_cinenet_grad_cfg4, rnn_grad_cfg5=g_rnn_grad_cfg5, rnn_grad=g_rnn_grad, xpdnet_grad=g_xpdnet_grad, cinenet_grad=g_cinenet_grad, lightning=g_lightning, varnet_grad=g_varnet_grad, varnet_grad_cfg2=g_varnet_grad_cfg2, rnn=g_rnn, xpdnet=g_xpdnet, cinenet=g_cinenet, ops=g_ops, unet=g_unet, varnet_block=g_varnet_block,
                  varnet_tiny=g_varnet_tiny, masks=g_masks, varnet_full=g_varnet_full,
                  varnet_cfg1=g_varnet_cfg1, xpdnet_cfg3=g_xpdnet_cfg3,
                  cinenet_cfg4=g_cinenet_cfg4, rnn_cfg5=g_rnn_cfg5, metrics=g_metrics, frontend=g_frontend, fft_smooth=g_fft_smooth)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    for name, fn in GENERATORS.items():
        if args.only and name not in args.only:
            continue
        print(f"[{name}]")
        fn()
