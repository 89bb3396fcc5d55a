"""GPU parity: the HIP path (through the C ABI) against the oracle and the reference goldens.

Tolerances (fp32 path): op level rel 1e-5 of the output peak, block level 5e-5,
model level max|d|/peak <= 1e-4.
"""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err, rnd, state_dict_from
from cine_hip import ops as cine_ops

pytestmark = pytest.mark.gpu

OP_TOL, BLOCK_TOL, MODEL_TOL = 1e-5, 5e-5, 1e-4


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def cuda(a, dev):
    return torch.as_tensor(a).to(dev)


# ------------------------------------------------------------------ FFT ops
@pytest.mark.parametrize("tag", ["odd", "t15", "even", "mixed"])
def test_ffts_vs_reference_golden(golden, dev, tag):
    import reconstruction.utils as U
    g = golden("ops")
    x = cuda(g[f"{tag}_x"], dev)
    for name, fn in (("fft1c", U.fft1c), ("ifft1c", U.ifft1c), ("fft2c", U.fft2c), ("ifft2c", U.ifft2c)):
        assert rel_err(fn(x).cpu(), g[f"{tag}_{name}"]) < OP_TOL, name


def test_fft2c_200_vs_reference_golden(golden, dev):
    import reconstruction.utils as U
    g = golden("ops")
    x = rnd(int(g["full200_seed"]), 2, 200, 200, 2).to(dev)
    assert rel_err(U.fft2c(x).cpu(), g["full200_fft2c"]) < OP_TOL
    assert rel_err(U.ifft2c(x).cpu(), g["full200_ifft2c"]) < OP_TOL


def test_fft1c_variants_and_200(dev):
    from cine_hip import ops
    from oracle import centered_fft as cf
    for n in (5, 15, 16, 200):
        x = rnd(n, 7, n, 2)
        for inv in (False, True):
            ref = (cf.ifft1c if inv else cf.fft1c)(x)
            assert rel_err(ops.fft1c(x.to(dev), inverse=inv).cpu(), ref) < OP_TOL
            z = torch.view_as_complex(x)
            ref1 = torch.view_as_real((cf.xpd_temporal_ifft if inv else cf.xpd_temporal_fft)(z, dim=1))
            assert rel_err(ops.fft1c(x.to(dev), inverse=inv, variant=1).cpu(), ref1) < OP_TOL


def test_fft2c_full_batch_properties(dev):
    """cfg-2 sized batch (225 images of 200x200): round trip, Parseval, linearity."""
    import reconstruction.utils as U
    x = torch.randn(225, 200, 200, 2, device=dev)
    y = torch.randn(225, 200, 200, 2, device=dev)
    X = U.fft2c(x)
    assert rel_err(U.ifft2c(X).cpu(), x.cpu()) < OP_TOL
    assert abs(float((X.double() ** 2).sum() / (x.double() ** 2).sum()) - 1) < 1e-5
    assert rel_err(U.fft2c(2 * x - 3 * y).cpu(), (2 * X - 3 * U.fft2c(y)).cpu()) < OP_TOL
    # in-place call
    from cine_hip import ops, _lib
    z = x.clone()
    _lib.check(_lib.lib().cine_fft2c(z.data_ptr(), z.data_ptr(), 225, 200, 200, 0, torch.cuda.current_stream().cuda_stream))
    assert torch.equal(z, X)


@pytest.mark.parametrize("tag", ["a96x120", "a192x160", "a256x320", "a384x512", "a45x250", "a400x405"])
def test_fft2c_mixed_radix_vs_reference_golden(golden, dev, tag):
    """2^a 3^b 5^c lengths other than 200 run the mixed-radix Stockham engine (fft_core.h MixedRadix; 512 needs the > 64 KB LDS
    opt-in); the reference's outputs are pinned whole or on a strided lattice (make_golden.py g_fft_smooth)."""
    import reconstruction.utils as U
    g = golden("fft_smooth")
    n, h, w = (int(v) for v in g[f"{tag}_shape"])
    sh, sw = (int(v) for v in g[f"{tag}_stride"])
    x = rnd(int(g[f"{tag}_seed"]), n, h, w, 2).to(dev)
    X = U.fft2c(x)
    assert rel_err(X.cpu()[:, ::sh, ::sw], g[f"{tag}_fft2c"]) < OP_TOL
    assert rel_err(U.ifft2c(x).cpu()[:, ::sh, ::sw], g[f"{tag}_ifft2c"]) < OP_TOL
    assert rel_err(U.ifft2c(X).cpu(), x.cpu()) < OP_TOL
    assert abs(float((X.double() ** 2).sum() / (x.double() ** 2).sum()) - 1) < 1e-5


@pytest.mark.parametrize("n", [30, 128, 360, 512])
def test_fft1c_mixed_radix_vs_reference_golden(golden, dev, n):
    import reconstruction.utils as U
    g = golden("fft_smooth")
    x = rnd(int(g[f"l{n}_seed"]), 3, n, 2).to(dev)
    assert rel_err(U.fft1c(x).cpu(), g[f"l{n}_fft1c"]) < OP_TOL
    assert rel_err(U.ifft1c(x).cpu(), g[f"l{n}_ifft1c"]) < OP_TOL


@pytest.mark.parametrize("n", [2, 3, 4, 6, 8, 9, 10, 12, 18, 20, 25, 27, 32, 48, 50, 64, 75, 100, 125, 243, 250, 300, 320, 375, 480, 486, 500])
def test_fft1c_every_radix_pattern_vs_oracle(dev, n):
    """Each stage pattern of the planner (4s, one 2, 3s, 5s; one stage only; even and odd stage counts -> result in either tile)."""
    from cine_hip import ops
    from oracle import centered_fft as cf
    x = rnd(n, 11, n, 2)
    for inv in (False, True):
        ref = (cf.ifft1c if inv else cf.fft1c)(x)
        assert rel_err(ops.fft1c(x.to(dev), inverse=inv).cpu(), ref) < OP_TOL, (n, inv)


def test_unsupported_length_fails_loudly(dev):
    """Lengths with a prime factor above 5 stop at 400 (direct DFT), smooth ones at 512."""
    from cine_hip._lib import CineHipError
    import reconstruction.utils as U
    with pytest.raises(CineHipError):
        U.fft2c(torch.zeros(1, 401, 8, 2, device=dev))
    with pytest.raises(CineHipError):
        U.fft2c(torch.zeros(1, 8, 540, 2, device=dev))


# ------------------------------------------------------------------ coil operators
def _block_inputs(g, dev):
    return (cuda(g["k"], dev), cuda(g["kref"], dev), cuda(g["mask"], dev), cuda(g["sens"], dev))


def test_sens_reduce_expand_dc_vs_golden(golden, dev):
    from cine_hip import ops
    g = golden("varnet_block")
    k, kref, mask, sens = _block_inputs(g, dev)
    img = ops.sens_reduce(k, sens)
    assert rel_err(img.cpu(), g["XF_reduce"]) < OP_TOL
    assert torch.equal(k.cpu(), torch.from_numpy(g["k"]))            # input untouched
    assert rel_err(ops.sens_expand_dc(img, sens).cpu(), g["XF_expand"]) < OP_TOL
    # magnitude variant == |reduce|
    mag = ops.sens_reduce(k, sens, magnitude=True)
    ref = torch.from_numpy(g["XF_reduce"]).squeeze(2).pow(2).sum(-1).sqrt()
    assert rel_err(mag.cpu(), ref) < OP_TOL
    # destroy_input variant gives the same image
    img2 = ops.sens_reduce(k.clone(), sens, destroy_input=True)
    assert torch.equal(img2, img)


def test_soft_and_hard_dc_vs_oracle(dev):
    from cine_hip import ops
    from oracle import varnet_ref as V
    import torch.nn.functional as F
    t, c, h, w = 5, 3, 24, 20
    img, sens, kref = rnd(1, 1, t, 1, h, w, 2), rnd(2, 1, 1, c, h, w, 2), rnd(3, 1, t, c, h, w, 2)
    mask = (torch.rand(1, t, 1, h, 1, 1, generator=torch.Generator().manual_seed(0)) > 0.6).byte()
    for lam in (-1.3, 0.5413, 25.0):
        lam_t = torch.tensor([lam])
        kth = V.VarNetBlock.sens_expand(img, sens)
        v = F.softplus(lam_t)
        ref = (1 - mask) * kth + mask * (kth + v * kref) / (1 + v)
        out = ops.sens_expand_dc(img.to(dev), sens.to(dev), kref.to(dev), mask.to(dev), lam_t.to(dev))
        assert rel_err(out.cpu(), ref) < OP_TOL
    hard = ops.sens_expand_dc(img.to(dev), sens.to(dev), None, mask.to(dev), None, hard_mask=True)
    assert rel_err(hard.cpu(), V.VarNetBlock.sens_expand(img, sens) * mask + 0.0) < OP_TOL


def test_coil_ops_full_size_vs_oracle(dev):
    """cfg-2 shapes: 15 frames x 15 coils x 200x200."""
    from cine_hip import ops, synth
    from oracle import varnet_ref as V
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=4, seed=0)
    k, sens, mask = ex["masked_kspace"], ex["sens_maps"], ex["mask"]
    ref_img = V.VarNetBlock.sens_reduce(k, sens)
    img = ops.sens_reduce(k.to(dev), sens.to(dev))
    assert rel_err(img.cpu(), ref_img) < OP_TOL
    lam = torch.tensor([0.5413])
    kth = V.VarNetBlock.sens_expand(ref_img, sens)
    v = torch.nn.functional.softplus(lam)
    ref = (1 - mask) * kth + mask * (kth + v * k) / (1 + v)
    out = ops.sens_expand_dc(img, sens.to(dev), k.to(dev), mask.to(dev), lam.to(dev))
    assert rel_err(out.cpu(), ref) < OP_TOL
    # adjointness <A x, y> == <x, A^H y> with A = sens_expand, A^H = sens_reduce
    x = torch.randn_like(ref_img).to(dev); y = torch.randn_like(k).to(dev)
    lhs = (ops.sens_expand_dc(x, sens.to(dev)).double() * y.double()).sum()
    rhs = (x.double() * ops.sens_reduce(y, sens.to(dev)).double()).sum()
    assert abs(float(lhs - rhs)) / abs(float(lhs)) < 1e-5


@pytest.mark.parametrize("shape", [(5, 3, 24, 20), (3, 15, 200, 200), (2, 4, 200, 24), (2, 5, 24, 200)])
def test_hybrid_space_chain_equals_kspace_chain(dev, shape):
    """kspace_to_hybrid / hybrid_reduce / expand_dc_hybrid against the k-space operators they split."""
    from cine_hip import ops
    t, c, h, w = shape
    g = torch.Generator().manual_seed(h * w)
    k = torch.randn(1, t, c, h, w, 2, generator=g).to(dev)
    sens = torch.randn(1, 1, c, h, w, 2, generator=g).to(dev)
    mask = (torch.rand(1, t, 1, h, 1, 1, generator=g) > 0.7).byte().to(dev)
    lam = torch.tensor([0.3]).to(dev)
    hyb = ops.kspace_to_hybrid(k)
    img = ops.sens_reduce(k, sens)
    assert rel_err(ops.hybrid_reduce(hyb, sens).cpu(), img.cpu()) < OP_TOL
    assert rel_err(ops.hybrid_reduce(hyb, sens, magnitude=True).cpu(), ops.sens_reduce(k, sens, magnitude=True).cpu()) < OP_TOL
    knew = ops.sens_expand_dc(img, sens, k, mask, lam)
    hyb2 = ops.expand_dc_hybrid(img, sens, k, mask, lam)
    assert rel_err(hyb2.cpu(), ops.kspace_to_hybrid(knew).cpu()) < OP_TOL
    # in place on an existing hybrid buffer
    ops.expand_dc_hybrid(img, sens, k, mask, lam, out=hyb)
    assert torch.equal(hyb, hyb2)


def test_sens_prologue_and_rss(dev):
    from cine_hip import ops
    from oracle import centered_fft as cf, complex_ops as co
    k = rnd(5, 2, 4, 3, 12, 10, 2)
    ref = cf.ifft2c(co.mask_center(k.mean(dim=1), 4, 9))
    out = ops.sens_prologue(k.to(dev), 4, 9)
    assert rel_err(out.cpu(), ref) < OP_TOL
    x = rnd(6, 2, 3, 12, 10, 2)
    ref = x / co.rss_complex(x, dim=1).unsqueeze(-1).unsqueeze(1)
    assert rel_err(ops.rss_normalise_(x.to(dev).clone()).cpu(), ref) < OP_TOL


# ------------------------------------------------------------------ U-Net pieces
def test_unet_blocks_vs_reference_golden(golden, dev):
    from reconstruction.models.denoisers import unet as HU
    g = golden("unet")
    cb = HU.ConvBlock(3, 8, 0.0, 2); cb.load_state_dict(state_dict_from(g, "cb::"), strict=True); cb.to(dev)
    tb = HU.TransposeConvBlock(8, 4, 2); tb.load_state_dict(state_dict_from(g, "tb::"), strict=True); tb.to(dev)
    un = HU.Unet(chans=4, num_pool_layers=2); un.load_state_dict(state_dict_from(g, "un::"), strict=True); un.to(dev).eval()
    assert rel_err(cb(cuda(g["cb_x"], dev)).cpu(), g["cb_y"]) < BLOCK_TOL
    assert rel_err(tb(cuda(g["tb_x"], dev)).cpu(), g["tb_y"]) < BLOCK_TOL
    assert rel_err(un(cuda(g["un_x"], dev)).cpu(), g["un_y"]) < BLOCK_TOL
    assert rel_err(un(cuda(g["un_odd_x"], dev)).cpu(), g["un_odd_y"]) < BLOCK_TOL


@pytest.mark.parametrize("shape", [(2, 3, 5, 24, 20), (1, 4, 15, 40, 32), (2, 2, 4, 9, 11)])
def test_standalone_3d_blocks_vs_oracle(dev, shape):
    """ConvBlock(dims=3) / TransposeConvBlock(dims=3) called on their own (reference unet.py:149-182, 204-233: Conv3d / ConvTranspose3d + InstanceNorm3d +
    LeakyReLU; the U-Nets use whole launch sequences instead) on the volume kernels, against the oracle's blocks; MWCNN(dims=3) is constructible with
    the reference's state-dict keys and says why its forward has no defined result."""
    from cine_hip import synth
    from oracle import regularisers as R
    from reconstruction.models.denoisers import unet as HU
    from reconstruction.models.denoisers.mwcnn import MWCNN
    n, cin, d, h, w = shape
    x = rnd(5, *shape)
    cb = HU.ConvBlock(cin, 8, 0.0, 3).eval(); synth.fill_parameters_(cb, 3, keep=())
    rb = R.ConvBlock(cin, 8, 0.0, 3).eval(); rb.load_state_dict(cb.state_dict(), strict=True)
    with torch.no_grad():
        assert rel_err(cb.to(dev)(x.to(dev)).cpu(), rb(x)) < BLOCK_TOL
    tb = HU.TransposeConvBlock(cin, 6, 3).eval(); synth.fill_parameters_(tb, 4, keep=())
    rt = R.TransposeConvBlock(cin, 6, 3).eval(); rt.load_state_dict(tb.state_dict(), strict=True)
    with torch.no_grad():
        got = tb.to(dev)(x.to(dev)).cpu()
    assert got.shape == (n, 6, 2 * d, 2 * h, 2 * w) and rel_err(got, rt(x)) < BLOCK_TOL
    net = MWCNN(2, 2, dims=3, n_scales=2, n_filters_per_scale=[8, 16], n_convs_per_scale=[1, 1], first_conv_n_filters=8)
    assert net.first_convs[0].layers[0].weight.dim() == 5
    with pytest.raises(NotImplementedError, match="unpacks four dimensions"):
        net.to(dev)(rnd(6, 1, 2, 4, 8, 8).to(dev))


def test_norm_unet_vs_reference_golden(golden, dev):
    from reconstruction.models.denoisers import NormUnet
    from cine_hip import ops
    g = golden("unet")
    nu = NormUnet(4, 2); nu.load_state_dict(state_dict_from(g, "nu::"), strict=True); nu.to(dev).eval()
    x = cuda(g["nu_x"], dev)
    planes, stats = ops.normunet_pack(x.reshape(3, 20, 5, 2))
    ref_norm = torch.from_numpy(g["nu_norm"])                      # (3, 2, 20, 5) before padding
    assert rel_err(planes[:, :, 6:26, 5:10].cpu(), ref_norm) < OP_TOL   # pads: h 20->32 [6,6], w 5->16 [5,6]
    pad_only = planes.clone(); pad_only[:, :, 6:26, 5:10] = 0
    assert float(pad_only.abs().max()) == 0.0
    assert rel_err(stats[:, :, 0].cpu(), g["nu_mean"].reshape(3, 2)) < OP_TOL
    assert rel_err(stats[:, :, 1].cpu(), g["nu_std"].reshape(3, 2)) < OP_TOL
    assert rel_err(nu(x).cpu(), g["nu_y"]) < BLOCK_TOL
    with pytest.raises(ValueError):
        nu(torch.zeros(1, 1, 8, 8, 3, device=dev))


@pytest.mark.parametrize("cin,cout,h,w", [(2, 16, 208, 16), (16, 16, 208, 16), (16, 32, 104, 8), (32, 32, 104, 8),
                                          (32, 64, 52, 4), (64, 64, 52, 4), (64, 128, 26, 2), (128, 128, 26, 2),
                                          (128, 64, 52, 4), (64, 32, 104, 8), (32, 16, 208, 16),
                                          (8, 8, 208, 208), (5, 10, 25, 25), (18, 36, 13, 7), (2, 8, 1, 1)])
def test_conv3x3_shapes_vs_torch(dev, cin, cout, h, w):
    """Every cfg-2 layer shape plus odd ones, against torch's fp32 conv on the CPU; the fused
    InstanceNorm partial statistics are merged and compared with the exact mean / rstd."""
    from cine_hip import ops
    import torch.nn.functional as F
    n = 3
    x = rnd(cin * 7 + h, n, cin, h, w); wt = rnd(cout, cout, cin, 3, 3) / (3 * cin ** 0.5)
    y, part = ops.conv3x3_in([(x.to(dev), None, 0)], ops.pack_conv3x3(wt.to(dev)), cout, h, w)
    ref = F.conv2d(x, wt, padding=1)
    assert rel_err(y.cpu(), ref) < OP_TOL
    assert float(part[..., 0].sum(dim=2).min()) == h * w and float(part[..., 0].sum(dim=2).max()) == h * w
    st = ops.instnorm_finalize(part)
    assert rel_err(st[..., 0].cpu(), ref.mean(dim=(2, 3))) < 1e-4
    rstd = 1 / torch.sqrt(ref.var(dim=(2, 3), unbiased=False) + 1e-5)
    assert rel_err(st[..., 1].cpu(), rstd) < 1e-4


def test_conv3x3_two_weight_sets_in_one_launch(dev):
    from cine_hip import ops
    import torch.nn.functional as F
    x = rnd(1, 6, 16, 52, 16); wa = rnd(2, 16, 16, 3, 3) / 12; wb = rnd(3, 16, 16, 3, 3) / 12
    y, _ = ops.conv3x3_in([(x.to(dev), None, 0)], ops.pack_conv3x3(wa.to(dev)), 16, 52, 16,
                          wpacked2=ops.pack_conv3x3(wb.to(dev)), set_split=4)
    ref = torch.cat([F.conv2d(x[:4], wa, padding=1), F.conv2d(x[4:], wb, padding=1)])
    assert rel_err(y.cpu(), ref) < OP_TOL


def test_conv3x3_fused_sources_vs_torch(dev):
    """norm+LReLU on load, pooled source, concat of two sources with a short `up` extent."""
    from cine_hip import ops
    import torch.nn.functional as F
    act = lambda t: F.leaky_relu(F.instance_norm(t, eps=1e-5), 0.2)
    for (n, c, h, w, uh) in ((2, 8, 13, 10, 12), (2, 16, 52, 16, 52), (3, 8, 26, 8, 26)):
        skip = rnd(1, n, c, h, w); up = rnd(2, n, c, uh, w)           # uh < h -> zero pad row
        wt = rnd(3, 16, 2 * c, 3, 3) / 12
        p_skip, p_up = ops.instnorm_partials(skip.to(dev)), ops.instnorm_partials(up.to(dev))
        y, _ = ops.conv3x3_in([(up.to(dev), p_up, 1), (skip.to(dev), p_skip, 1)], ops.pack_conv3x3(wt.to(dev)), 16, h, w)
        ref = F.conv2d(torch.cat([F.pad(act(up), [0, 0, 0, h - uh]), act(skip)], 1), wt, padding=1)
        assert rel_err(y.cpu(), ref) < BLOCK_TOL
    # pooled sources incl. odd widths / heights (avg_pool2d floors): 208 x 15 -> 104 x 7 is what the bare CineNet / XPDNet
    # U-Nets see at full size (15 frames), 2h x (2w + 1) and (2h + 1) x 2w the generic odd cases
    for (n, c, H2, W2) in ((2, 8, 26, 20), (2, 16, 104, 16), (2, 8, 27, 21), (2, 64, 8, 4), (2, 16, 208, 15), (3, 16, 26, 17),
                           (2, 8, 27, 16), (1, 32, 104, 7)):
        big = rnd(4, n, c, H2, W2)
        p_big = ops.instnorm_partials(big.to(dev))
        wt2 = rnd(5, 16, c, 3, 3) / 8
        y, _ = ops.conv3x3_in([(big.to(dev), p_big, 2)], ops.pack_conv3x3(wt2.to(dev)), 16, H2 // 2, W2 // 2)
        ref = F.conv2d(F.avg_pool2d(act(big), 2), wt2, padding=1)
        assert rel_err(y.cpu(), ref) < BLOCK_TOL


def test_conv3x3_coarse_planes_vs_torch(dev):
    """2-D planes on the flattened-position K-split kernel (conv_coarse.hip: wider than the plane-wide tiles, small, > 32 output rows --
    the sensitivity network's 26 x 26 level with 64 channels on 15 coil planes): normalised source, pooled source, concat with a short
    `up` extent, a ragged channel count, two weight sets in one launch, and the element-wise staging that serves the sources the fast
    paths do not (Haar DWT on load, an added skip).  One statistics record per 32-position tile of the flattened plane."""
    from cine_hip import ops
    from cine_hip._lib import lib
    import torch.nn.functional as F
    act = lambda t: F.leaky_relu(F.instance_norm(t, eps=1e-5), 0.2)
    n, h, w = 15, 26, 26
    assert lib().cine_conv_stat_partials(64, h, w, 0) == -(-h * (w + 1) // 32)

    def check(y, part, ref, tol=BLOCK_TOL):
        assert rel_err(y.cpu(), ref) < tol
        assert part.shape[2] == lib().cine_conv_stat_partials(ref.shape[1], ref.shape[2], ref.shape[3], 0)
        assert float(part[..., 0].sum(dim=2).min()) == ref.shape[2] * ref.shape[3] == float(part[..., 0].sum(dim=2).max())
        st = ops.instnorm_finalize(part)
        assert rel_err(st[..., 0].cpu(), ref.mean(dim=(2, 3))) < 1e-4
        assert rel_err(st[..., 1].cpu(), 1 / torch.sqrt(ref.var(dim=(2, 3), unbiased=False) + 1e-5)) < 1e-4
    x = rnd(31, n, 64, h, w) * 1.3 + 0.2
    wt = rnd(32, 64, 64, 3, 3) / 24
    px = ops.instnorm_partials(x.to(dev))
    y, part = ops.conv3x3_in([(x.to(dev), px, 1)], ops.pack_conv3x3(wt.to(dev)), 64, h, w)
    check(y, part, F.conv2d(act(x), wt, padding=1))
    y, part = ops.conv3x3_in([(x.to(dev), None, 0)], ops.pack_conv3x3(wt.to(dev)), 64, h, w)       # plain source
    check(y, part, F.conv2d(x, wt, padding=1), OP_TOL)
    big = rnd(33, n, 32, 2 * h, 2 * w + 1)                                                      # pooled (avg_pool2d floors)
    wp = rnd(34, 64, 32, 3, 3) / 17
    y, part = ops.conv3x3_in([(big.to(dev), ops.instnorm_partials(big.to(dev)), 2)], ops.pack_conv3x3(wp.to(dev)), 64, h, w)
    check(y, part, F.conv2d(F.avg_pool2d(act(big), 2), wp, padding=1))
    up, skip = rnd(35, 4, 24, h - 1, w), rnd(36, 4, 20, h, w)                                   # ragged concat, `up` one row short, 72 rows
    wc = rnd(37, 72, 44, 3, 3) / 20
    y, part = ops.conv3x3_in([(up.to(dev), ops.instnorm_partials(up.to(dev)), 1), (skip.to(dev), None, 0)], ops.pack_conv3x3(wc.to(dev)), 72, h, w)
    check(y, part, F.conv2d(torch.cat([F.pad(act(up), [0, 0, 0, 1]), skip], 1), wc, padding=1))
    wb = rnd(38, 64, 64, 3, 3) / 24                                                             # two weight sets
    y, part = ops.conv3x3_in([(x.to(dev), px, 1)], ops.pack_conv3x3(wt.to(dev)), 64, h, w, wpacked2=ops.pack_conv3x3(wb.to(dev)), set_split=6)
    check(y, part, torch.cat([F.conv2d(act(x[:6]), wt, padding=1), F.conv2d(act(x[6:]), wb, padding=1)]))
    # sources outside the fast staging paths, through cine_conv3x3_ex: Haar DWT of a normalised tensor + an added plain skip
    src = rnd(39, 3, 16, 2 * h, 2 * w); skp = rnd(40, 3, 64, h, w)
    wd = rnd(41, 48, 64, 3, 3) / 24
    src_d, skp_d, wd_p = src.to(dev), skp.to(dev), ops.pack_conv3x3(wd.to(dev))      # (kept alive: raw pointers go to the call below)
    psrc = ops.instnorm_partials(src_d)
    a_ = act(src) * 0.5
    x1, x2, x3, x4 = a_[:, :, 0::2, 0::2], a_[:, :, 1::2, 0::2], a_[:, :, 0::2, 1::2], a_[:, :, 1::2, 1::2]      # mwcnn.py:224-236
    dwt = torch.cat([x1 + x2 + x3 + x4, -x1 - x2 + x3 + x4, -x1 + x2 - x3 + x4, x1 - x2 - x3 + x4], 1)
    yd = torch.empty((3, 48, h, w), device=dev)
    pd = torch.empty((3, 48, lib().cine_conv_stat_partials(48, h, w, 0), 3), device=dev)
    P = lambda t_: None if t_ is None else t_.data_ptr()
    from cine_hip import _lib
    _lib.check(lib().cine_conv3x3_ex(src_d.data_ptr(), psrc.data_ptr(), psrc.shape[2], 16, 3 | 8, 2 * h, 2 * w,
                                     skp_d.data_ptr(), None, 0, 64, 0, h, w, 1, wd_p.data_ptr(), None, None, 0,
                                     yd.data_ptr(), pd.data_ptr(), 3, 48, h, w, ops.IN_EPS, ops.lrelu_slope(), ops._stream()), "cine_conv3x3_ex")
    check(yd, pd, F.conv2d(dwt + skp, wd, padding=1))


@pytest.mark.parametrize("cin,cout,h,w", [(128, 64, 26, 2), (64, 32, 52, 4), (32, 16, 104, 8), (8, 4, 6, 5), (16, 8, 26, 26)])
def test_tconv_and_conv1x1_vs_torch(dev, cin, cout, h, w):
    from cine_hip import ops
    import torch.nn.functional as F
    n = 3
    x = rnd(cin + h, n, cin, h, w); wt = rnd(cout, cin, cout, 2, 2) / cin ** 0.5
    act = lambda t: F.leaky_relu(F.instance_norm(t, eps=1e-5), 0.2)
    px = ops.instnorm_partials(x.to(dev))
    y, part = ops.tconv2x2_in(x.to(dev), px, 1, ops.pack_tconv2x2(wt.to(dev)), cout)
    ref = F.conv_transpose2d(act(x), wt, stride=2)
    assert rel_err(y.cpu(), ref) < BLOCK_TOL
    st = ops.instnorm_finalize(part)
    assert rel_err(st[..., 0].cpu(), ref.mean(dim=(2, 3))) < 1e-4
    assert rel_err(st[..., 1].cpu(), 1 / torch.sqrt(ref.var(dim=(2, 3), unbiased=False) + 1e-5)) < 1e-4
    for c1 in (2, 3, 6):       # <= 4 output channels: streaming kernel (unless h*w is odd); more: the MFMA kernel
        w1 = rnd(7, c1, cin, 1, 1) / cin ** 0.5; b1 = rnd(8, c1)
        y1 = ops.conv1x1_bias(x.to(dev), px, 1, ops.pack_conv1x1(w1.to(dev)), b1.to(dev))
        assert rel_err(y1.cpu(), F.conv2d(act(x), w1, b1)) < BLOCK_TOL
        y0 = ops.conv1x1_bias(x.to(dev), None, 0, ops.pack_conv1x1(w1.to(dev)), b1.to(dev))
        assert rel_err(y0.cpu(), F.conv2d(x, w1, b1)) < BLOCK_TOL


def test_conv3x3_random_shapes_vs_torch(dev):
    """Seeded sweep over odd / tiny / wide extents, channel counts off the chunk size, both on-load modes, pooled and concat
    sources, two weight sets: raw output and merged statistics against torch (fp32 CPU)."""
    from cine_hip import ops
    import torch.nn.functional as F
    import random
    rng = random.Random(1234)
    act = lambda t: F.leaky_relu(F.instance_norm(t, eps=1e-5), 0.2)
    for case in range(40):
        n = rng.choice([1, 2, 3, 5])
        h, w = rng.choice([1, 2, 3, 5, 8, 13, 16, 26, 31, 52]), rng.choice([1, 2, 3, 4, 6, 8, 15, 16, 17, 33])
        c0 = rng.choice([1, 2, 3, 8, 9, 16, 20])
        cout = rng.choice([1, 2, 8, 16, 17, 32, 40, 64, 72])
        kind = rng.choice(["plain", "norm", "pool", "concat"])
        g = torch.Generator().manual_seed(case)
        wt = torch.randn(cout, c0 + (c0 if kind == "concat" else 0), 3, 3, generator=g) / 6
        if kind == "pool":
            x = torch.randn(n, c0, 2 * h + rng.choice([0, 1]), 2 * w + rng.choice([0, 1]), generator=g)
            if x.shape[3] % 2:      # the pooled source is addressed pairwise along x
                x = x[..., :-1].contiguous()
            ref_in = F.avg_pool2d(act(x), 2)[..., :h, :w]
            if ref_in.shape[2] < h or ref_in.shape[3] < w:
                continue
            srcs = [(x.to(dev), ops.instnorm_partials(x.to(dev)), 2)]
        elif kind == "concat":
            x = torch.randn(n, c0, h, w, generator=g); x2 = torch.randn(n, c0, h, w, generator=g)
            ref_in = torch.cat([act(x), act(x2)], 1)
            srcs = [(x.to(dev), ops.instnorm_partials(x.to(dev)), 1), (x2.to(dev), ops.instnorm_partials(x2.to(dev)), 1)]
        else:
            x = torch.randn(n, c0, h, w, generator=g)
            ref_in = act(x) if kind == "norm" else x
            srcs = [(x.to(dev), ops.instnorm_partials(x.to(dev)) if kind == "norm" else None, 1 if kind == "norm" else 0)]
        if h * w == 1 and kind != "plain":
            continue                                  # InstanceNorm of a single pixel is degenerate in the reference too
        y, part = ops.conv3x3_in(srcs, ops.pack_conv3x3(wt.to(dev)), cout, h, w)
        ref = F.conv2d(ref_in, wt, padding=1)
        assert rel_err(y.cpu(), ref) < BLOCK_TOL, (case, kind, n, c0, cout, h, w)
        st = ops.instnorm_finalize(part).cpu()
        assert (st[..., 0] - ref.mean(dim=(2, 3))).abs().max() < 1e-4 * max(1.0, float(ref.abs().max())), (case, kind)


def test_conv_plane_bit_identical_to_general_kernel(dev):
    """csrc/conv_plane.hip (the lean kernel of plane-wide tiles) against conv_tile on every cfg-2 U-Net layer kind: raw output AND
    statistics records bit for bit -- same geometry, accumulation order and statistics arithmetic (incl. the v_permlane swaps
    standing in for __shfl_xor); the Haar-DWT-on-load convs (round 5: band-interleaved chunks) to fp32 rounding.  Two weight sets, 7 samples (full and boundary tiles in every launch)."""
    from cine_hip import ops
    from cine_hip._lib import lib
    n = 7
    cases = []            # (kind, c0, cout, h, w)
    for d, (c, h, w) in enumerate(((16, 208, 16), (32, 104, 8), (64, 52, 4), (128, 26, 2))):
        cases += [("norm", c, c, h, w), ("plain", c, c, h, w)]
        if d == 0:
            cases += [("plain", 2, c, h, w)]
        else:
            cases += [("pool", c // 2, c, h, w)]
        if d < 3:
            cases += [("concat", c, c, h, w), ("plain", c, 2 * c, h, w)]          # up path; input gradient of the concat conv
    cases += [("norm", 16, 16, 200, 16), ("pool", 16, 32, 100, 8), ("norm", 128, 128, 25, 2), ("norm", 24, 16, 52, 16)]     # ragged last tiles, 3 chunks
    cases += [("norm", 16, 16, 100, 8), ("norm", 32, 32, 50, 4), ("norm", 64, 64, 25, 2), ("norm", 64, 32, 48, 4),
              ("norm", 16, 64, 100, 8), ("plain", 10, 16, 200, 16), ("norm", 12, 16, 200, 16)]          # the MWCNN's inner conv blocks (cfg 3 planes), 16 -> 64 before an IWT, 8 + 2 input channels
    cases += [("dwt", 16, 16, 100, 8), ("dwt", 16, 32, 50, 4), ("dwt", 32, 64, 25, 2), ("dwt", 8, 16, 96, 8)]              # Haar DWT of the scale above on load (first conv of an MWCNN scale)
    cases += [("iwt", 16, 16, 100, 8), ("iwt", 32, 32, 50, 4), ("iwt", 16, 10, 200, 16), ("iwt", 8, 16, 96, 8)]             # Haar IWT of the scale below + additive skip on load (last: bias, no statistics)
    cases += [("relu", 16, 16, 208, 8), ("relu", 32, 32, 100, 4), ("relu", 64, 64, 50, 2), ("relu", 16, 16, 200, 16)]       # the MWCNN's conv + bias + ReLU blocks
    try:
        for kind, c0, cout, h, w in cases:
            g = torch.Generator().manual_seed(c0 * 131 + cout + h)
            cin = 2 * c0 if kind == "concat" else 4 * c0 if kind == "dwt" else c0
            wa = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(dev)
            wb = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(dev)
            if kind == "pool":
                x = torch.randn(n, c0, 2 * h, 2 * w, generator=g).to(dev)
                srcs = [(x, ops.instnorm_partials(x), 2)]
            elif kind == "dwt":
                x = torch.randn(n, c0, 2 * h, 2 * w, generator=g).to(dev)
                srcs = [(x, ops.instnorm_partials(x), 3 | 8)]                    # mode 3 = DWT on load, bit 3 = the source is raw (normalise + LeakyReLU first)
            elif kind == "concat":
                x = torch.randn(n, c0, h, w, generator=g).to(dev); x2 = torch.randn(n, c0, h, w, generator=g).to(dev)
                # `up` comes out of a transpose conv: several statistics records per plane
                p_up = ops.tconv2x2_in(torch.randn(n, 8, h // 2, w // 2, generator=g).to(dev), None, 0,
                                       ops.pack_tconv2x2((torch.randn(8, c0, 2, 2, generator=g) / 3).to(dev)), c0)
                srcs = [(p_up[0], p_up[1], 1), (x2, ops.instnorm_partials(x2), 1)]
            else:
                x = torch.randn(n, c0, h, w, generator=g).to(dev)
                srcs = [(x, ops.instnorm_partials(x) if kind == "norm" else None, 1 if kind == "norm" else 0)]
            elif_iwt = kind == "iwt"
            if elif_iwt:
                cur = torch.randn(n, 4 * c0, h // 2, w // 2, generator=g).to(dev)
                pc, px = ops.instnorm_partials(cur), ops.instnorm_partials(x)
                bias_i = torch.randn(cout, generator=g).to(dev) if cout == 10 else None
            outs = []
            for on in (7, 0):
                cine_ops.set_conv_plane(on)
                if elif_iwt:
                    y = torch.empty(n, cout, h, w, device=dev)
                    py = None if bias_i is not None else torch.empty((n, cout, lib().cine_conv_stat_partials(cout, h, w, 0), 3), device=dev)
                    pa, pb = ops.pack_conv3x3(wa), ops.pack_conv3x3(wb)
                    ptr = lambda t_: None if t_ is None else t_.data_ptr()
                    rc = lib().cine_conv3x3_ex2(cur.data_ptr(), pc.data_ptr(), pc.shape[2], 4 * c0, 4 | 8, h // 2, w // 2,
                                                x.data_ptr(), px.data_ptr(), px.shape[2], c0, 1, h, w, 1,
                                                pa.data_ptr(), ptr(bias_i), pb.data_ptr(), ptr(bias_i), 4, None, 0,
                                                y.data_ptr(), ptr(py), n, cout, h, w, ops.IN_EPS, ops.lrelu_slope(), torch.cuda.current_stream().cuda_stream)
                    assert rc == 0
                    outs.append((y, py))
                elif kind == "relu":
                    bias = torch.randn(cout, generator=torch.Generator().manual_seed(cout)).to(dev)
                    outs.append((ops.conv3x3_sum([x], ops.pack_conv3x3(wa), bias, cout, relu=True), None))
                else:
                    outs.append(ops.conv3x3_in(srcs, ops.pack_conv3x3(wa), cout, h, w, wpacked2=ops.pack_conv3x3(wb), set_split=4))
            (y1, p1), (y0, p0) = outs
            if kind == "dwt":
                # the lean kernel's chunks hold the four bands of two source channels (every 2 x 2 source block read and activated once),
                # the general kernel's eight channels of one band: the same products summed in another order -- equal to fp32 rounding
                assert rel_err(y1, y0) < 2e-6, (kind, c0, cout, h, w, float((y1 - y0).abs().max()))
                s1, s0 = ops.instnorm_finalize(p1), ops.instnorm_finalize(p0)
                assert rel_err(s1, s0) < 1e-5, (kind, c0, cout, h, w)
                continue
            assert torch.equal(y1, y0), (kind, c0, cout, h, w, float((y1 - y0).abs().max()))
            assert p1 is None or torch.equal(p1, p0), (kind, c0, cout, h, w)
    finally:
        cine_ops.set_conv_plane(7)


def test_tconv_plane_bit_identical_to_general_kernel(dev):
    """The lean transpose conv of csrc/conv_plane.hip (all input channels staged at once, weights streamed from L2) against
    conv_tile's TAPS = 1 path on the cfg-2 U-Net's three transpose convs, plus overhanging last tiles and a plain source."""
    from cine_hip import ops
    from cine_hip._lib import lib
    n = 5
    try:
        for cin, cout, h, w, mode in ((32, 16, 104, 8, 1), (64, 32, 52, 4, 1), (128, 64, 26, 2, 1), (32, 16, 100, 8, 1), (128, 64, 25, 2, 1),
                                      (64, 32, 50, 4, 0)):
            g = torch.Generator().manual_seed(cin + h)
            x = torch.randn(n, cin, h, w, generator=g).to(dev)
            wt = ops.pack_tconv2x2((torch.randn(cin, cout, 2, 2, generator=g) / cin ** 0.5).to(dev))
            if mode:      # several statistics records per plane, as a conv output has them
                px = ops.conv3x3_in([(x, None, 0)], ops.pack_conv3x3(torch.eye(cin, device=dev).view(cin, cin, 1, 1) * torch.tensor([[0., 0, 0], [0, 1, 0], [0, 0, 0]], device=dev)), cin, h, w)
                assert torch.equal(px[0], x)
                part = px[1]
            else:
                part = None
            outs = []
            for on in (7, 0):
                cine_ops.set_conv_plane(on)
                outs.append(ops.tconv2x2_in(x, part, mode, wt, cout))
            (y1, p1), (y0, p0) = outs
            assert torch.equal(y1, y0), (cin, cout, h, w, float((y1 - y0).abs().max()))
            assert torch.equal(p1, p0), (cin, cout, h, w)
    finally:
        cine_ops.set_conv_plane(7)


def test_conv_wide_bit_identical_to_general_kernel(dev):
    """csrc/conv_plane.hip's kernel for 16-wide column tiles of wider planes and volumes (sensitivity network, CRNN cells, 3-D
    U-Net) against conv_tile: 2-D layers of every source kind it takes (ragged last column tile, 52 statistics records per plane,
    bias + addend + ReLU epilogue), then a whole 3-D U-Net and the CRNN hybrids' conv3d on odd depths (three-pass form, `up`
    volumes shorter than the skip, narrow first layer)."""
    from cine_hip import ops, synth
    from cine_hip._lib import lib
    from reconstruction.models.denoisers.unet import Unet
    import torch.nn.functional as F
    n = 4                # (>= 4 planes of 200 x 200: the regular 52-row tiles, not the few-sample small-tile configuration)
    try:
        for kind, c0, cout, h, w in (("norm", 8, 8, 208, 208), ("norm", 16, 16, 104, 104), ("norm", 32, 32, 52, 52), ("norm", 64, 64, 26, 28),
                                     ("plain", 2, 8, 208, 208), ("concat", 8, 8, 120, 200), ("sum", 16, 16, 200, 200), ("sum1", 16, 2, 200, 200),
                                     ("sum_ragged", 16, 16, 200, 200), ("plain", 10, 8, 200, 200)):     # the BCRNN's all-frame conv over cat(hidden 16, image 2): 8 + 8 + 2
            g = torch.Generator().manual_seed(c0 * 7 + cout + h)
            c2 = 2 if kind == "sum_ragged" else c0
            cin = c0 + c2 if kind in ("concat", "sum", "sum_ragged") else c0
            wp = ops.pack_conv3x3((torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(dev))
            x = torch.randn(n, c0, h, w, generator=g).to(dev); x2 = torch.randn(n, c2, h, w, generator=g).to(dev)
            bias = torch.randn(cout, generator=g).to(dev); add = torch.randn(n, cout, h, w, generator=g).to(dev)
            outs = []
            for on in (7, 3):
                cine_ops.set_conv_plane(on)
                if kind in ("sum", "sum1", "sum_ragged"):
                    outs.append((ops.conv3x3_sum([x, x2] if kind != "sum1" else [x], wp, bias, cout, addend=add, relu=kind == "sum"), None))
                elif kind == "concat":
                    outs.append(ops.conv3x3_in([(x, ops.instnorm_partials(x), 1), (x2, ops.instnorm_partials(x2), 1)], wp, cout, h, w))
                else:
                    part = None
                    if kind == "norm":           # one record per tile of the producing layer, as inside the network
                        ident = torch.zeros(c0, c0, 3, 3, device=dev); ident[range(c0), range(c0), 1, 1] = 1.0
                        part = ops.conv3x3_in([(x, None, 0)], ops.pack_conv3x3(ident), c0, h, w)[1]
                    outs.append(ops.conv3x3_in([(x, part, 1 if kind == "norm" else 0)], wp, cout, h, w))
            (y1, p1), (y0, p0) = outs
            if kind == "dwt":
                # the lean kernel's chunks hold the four bands of two source channels (every 2 x 2 source block read and activated once),
                # the general kernel's eight channels of one band: the same products summed in another order -- equal to fp32 rounding
                assert rel_err(y1, y0) < 2e-6, (kind, c0, cout, h, w, float((y1 - y0).abs().max()))
                s1, s0 = ops.instnorm_finalize(p1), ops.instnorm_finalize(p0)
                assert rel_err(s1, s0) < 1e-5, (kind, c0, cout, h, w)
                continue
            assert torch.equal(y1, y0), (kind, c0, cout, h, w, float((y1 - y0).abs().max()))
            assert p1 is None or torch.equal(p1, p0), (kind, c0, cout, h, w)
        # one step of both directions of a BCRNN time sweep (recurrent_varnet.py:241-254: pair launch, addend + ReLU, second output
        # stored by one direction and added to by the other) and a single direction
        gg = torch.Generator().manual_seed(11)
        hx = [torch.randn(1, 16, 200, 200, generator=gg).to(dev) for _ in range(4)]
        w_hh = ops.pack_conv3x3((torch.randn(16, 16, 3, 3, generator=gg) / 12).to(dev))
        res = []
        for on in (7, 3):
            cine_ops.set_conv_plane(on)
            yf, yb, acc_f, acc_b = (torch.zeros_like(hx[0]) for _ in range(4))
            acc_b.fill_(0.5)
            ops.crnn_step2(w_hh, (hx[0], hx[1], yf, acc_f, True), (hx[2], hx[3], yb, acc_b, False))
            y1 = torch.zeros_like(hx[0])
            ops.crnn_step2(w_hh, (hx[2], hx[1], y1, acc_f, False))
            res.append((yf, yb, acc_f, acc_b, y1))
        for a_, b_ in zip(*res):
            assert torch.equal(a_, b_)
        assert torch.equal(res[0][3], 0.5 + res[0][1]) and torch.equal(res[0][2], res[0][0] + res[0][4])      # accum = y (store), then += y
        # volumes (enough tiles for the regular three-pass configurations): a 3-D U-Net (unet.py dims = 3: `up` volumes of depth 14
        # under a skip of depth 15, narrow first layer) and the hybrids' conv3d + bias + ReLU with 32 and 64 output channels
        net = Unet(in_chans=2, out_chans=2, chans=8, num_pool_layers=2, dims=3).eval(); synth.fill_parameters_(net, 5, keep=())
        net = net.to(dev)
        vol = torch.randn(1, 2, 15, 104, 104, generator=torch.Generator().manual_seed(3)).to(dev)
        v16 = torch.randn(1, 16, 7, 104, 104, generator=torch.Generator().manual_seed(6)).to(dev)
        ws_ = [(torch.randn(co, 16, 3, 3, 3, generator=torch.Generator().manual_seed(4 + co)) / 20).to(dev) for co in (32, 64)]
        bs_ = [torch.randn(co, generator=torch.Generator().manual_seed(5)).to(dev) for co in (32, 64)]
        outs = []
        for on in (7, 3):
            cine_ops.set_conv_plane(on)
            outs.append([ops.unet3d_forward(vol, ops.UnetWeights([net]))] + [ops.conv3d_bias_relu(v16, w_, b_, True) for w_, b_ in zip(ws_, bs_)])
        for a_, b_ in zip(*outs):
            assert torch.equal(a_, b_)
        assert rel_err(outs[0][1].cpu(), F.relu(F.conv3d(v16.cpu(), ws_[0].cpu(), bs_[0].cpu(), padding=1))) < OP_TOL
    finally:
        cine_ops.set_conv_plane(7)


def test_tconv_dgrad_plane_bit_identical_and_vs_torch(dev):
    """Input gradient of the k2 s2 transpose conv (cine_tconv2x2_dgrad: a 1x1 GEMM over the space-to-depth view of the output
    gradient) on the lean kernel of csrc/conv_plane.hip against conv_tile's element-wise mode-5 staging (bit for bit) and against
    torch's autograd of conv_transpose2d, on the cfg-2 U-Net's three shapes and an overhanging last tile."""
    from cine_hip import ops
    from cine_hip._lib import lib, check
    import torch.nn.functional as F
    n = 5
    try:
        for cin, cout, h, w in ((32, 16, 104, 8), (64, 32, 52, 4), (128, 64, 26, 2), (32, 16, 98, 8), (128, 64, 25, 2)):
            g = torch.Generator().manual_seed(cin + h)
            wt = (torch.randn(cin, cout, 2, 2, generator=g) / cin ** 0.5)
            gy = torch.randn(n, cout, 2 * h, 2 * w, generator=g)
            wp, wp2 = ops._pack("tcd", wt.to(dev)), ops._pack("tcd", (2 * wt).to(dev))
            outs = []
            for on in (7, 5):
                cine_ops.set_conv_plane(on)
                gx = torch.empty(n, cin, h, w, device=dev)
                check(lib().cine_tconv2x2_dgrad(gy.to(dev).data_ptr(), wp.data_ptr(), wp2.data_ptr(), 3, gx.data_ptr(), n, cin, cout, h, w, None), "cine_tconv2x2_dgrad")
                outs.append(gx)
            assert torch.equal(outs[0], outs[1]), (cin, cout, h, w, float((outs[0] - outs[1]).abs().max()))
            x = torch.zeros(n, cin, h, w, requires_grad=True)
            with torch.enable_grad():
                (F.conv_transpose2d(x, wt, stride=2) * gy).sum().backward()
            want = x.grad.clone(); want[3:] *= 2                                   # samples >= set_split use the second weight set
            assert rel_err(outs[0].cpu(), want) < OP_TOL, (cin, cout, h, w)
    finally:
        cine_ops.set_conv_plane(7)


# ------------------------------------------------------------------ blocks and models
@pytest.mark.parametrize("dyn", ["XF", "XT", "2D", "3D"])
def test_varnet_block_vs_reference_golden(golden, dev, dyn):
    import reconstruction.models as M
    g = golden("varnet_block")
    net = M.VarNet(1, 4, 2, 4, 2, dyn)
    net.load_state_dict(state_dict_from(g, f"{dyn}::sd::"), strict=True)
    net.to(dev).eval()
    blk = net.cascades[0]
    k, kref, mask, sens = _block_inputs(g, dev)
    if dyn in ("XF", "XT"):
        img = cuda(g[f"{dyn}_reduce"], dev)
        assert rel_err(blk.xfyf_transform(img.squeeze(2)).cpu(), g[f"{dyn}_xfyf"]) < BLOCK_TOL
    assert rel_err(blk(k, kref, mask, sens).cpu(), g[f"{dyn}_block"]) < BLOCK_TOL
    assert torch.equal(k.cpu(), torch.from_numpy(g["k"]))


def test_sensitivity_model_vs_reference_golden(golden, dev):
    import reconstruction.models as M
    g = golden("varnet_block")
    sm = M.SensitivityModel(4, 2)
    sm.load_state_dict(state_dict_from(g, "sens::sd::"), strict=True)
    sm.to(dev).eval()
    out = sm(cuda(g["kref"], dev), cuda(g["mask"], dev))
    assert rel_err(out.cpu(), g["sens_out"]) < BLOCK_TOL


@pytest.mark.parametrize("tag,dyn,ws", [("XF", "XF", False), ("XT", "XT", False), ("2D", "2D", False), ("3D", "3D", False),
                                        ("XFws", "XF", True)])
def test_varnet_tiny_vs_reference_golden(golden, dev, tag, dyn, ws):
    import reconstruction.models as M
    g = golden("varnet_tiny")
    net = M.VarNet(2, 4, 2, 4, 2, dyn, ws)
    net.load_state_dict(state_dict_from(g, f"{tag}::sd::"), strict=True)
    net.to(dev).eval()
    mk, mask = cuda(g["masked_kspace"], dev), cuda(g["mask"], dev)
    out = net(mk, mask)
    assert out.shape == g[f"{tag}_out"].shape
    assert rel_err(out.cpu(), g[f"{tag}_out"]) < MODEL_TOL
    assert torch.equal(mk.cpu(), torch.from_numpy(g["masked_kspace"]))     # caller's input not mutated


def test_varnet_cfg1_vs_reference_golden(golden, dev):
    """BASELINE configs[0]: 2D VarNet, 2 cascades, 8 coils, one 200x200 frame, R=4."""
    import reconstruction.models as M
    from cine_hip import synth
    g = golden("varnet_cfg1")
    ex = synth.make_cine_slice(1, 8, 200, 200, accel=4, seed=int(g["data_seed"]))
    net = M.VarNet(2, 8, 3, 16, 3, "2D")
    synth.fill_parameters_(net, int(g["weight_seed"]))
    net.to(dev).eval()
    out = net(ex["masked_kspace"].to(dev), ex["mask"].to(dev))
    assert rel_err(out.cpu(), g["out"]) < MODEL_TOL


def test_varnet_cfg2_vs_reference_golden(golden, dev):
    """BASELINE configs[1]: XF-VarNet, 6 cascades, 15 coils x 15 frames x 200x200, R=4.
    Checked against the reference's own output fingerprint: max|d|/peak, NMSE and SSIM delta."""
    import reconstruction.models as M
    from reconstruction.utils import evaluate
    from cine_hip import synth
    g = golden("varnet_cfg2")
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=4, seed=int(g["data_seed"]))
    net = M.VarNet(6, 8, 3, 16, 3, "XF")
    synth.fill_parameters_(net, int(g["weight_seed"]))
    net.to(dev).eval()
    out = net(ex["masked_kspace"].to(dev), ex["mask"].to(dev)).cpu()
    ref = torch.from_numpy(g["out_strided"])
    got = out[:, :, ::4, ::4]
    assert rel_err(got, ref) < MODEL_TOL
    assert float(((got - ref).double() ** 2).sum() / (ref.double() ** 2).sum()) < 1e-8          # NMSE
    assert abs(float(out.double().sum()) - float(g["out_sum"])) / float(g["out_sum"]) < 1e-5
    tgt = ex["target"][0, :, ::4, ::4].numpy()
    d_ssim = abs(evaluate.ssim(tgt, got[0].numpy()) - evaluate.ssim(tgt, ref[0].numpy()))
    assert d_ssim < 1e-4


# ------------------------------------------------------------------ CineNet
def test_cg_vector_ops(dev):
    from cine_hip import ops
    a, b = rnd(1, 3, 7, 200, 2), rnd(2, 3, 7, 200, 2)
    d = ops.dot(a.to(dev), b.to(dev))
    assert abs(float(d) - float(torch.dot(a.flatten().double(), b.flatten().double()))) < 1e-3
    num, den = torch.tensor([3.0], device=dev), torch.tensor([4.0], device=dev)
    assert rel_err(ops.axpby_dev(a.to(dev), b.to(dev), num=num, den=den, sign=-1.0).cpu(), a - 0.75 * b) < 1e-6
    lam = torch.tensor([0.3])
    assert rel_err(ops.axpby_dev(a.to(dev), b.to(dev), lambda_reg=lam.to(dev)).cpu(),
                   a + torch.nn.functional.softplus(lam) * b) < 1e-6


def test_cinenet_block_vs_reference_golden(golden, dev):
    import reconstruction.models as M
    g = golden("cinenet")
    net = M.CineNet(2, 3, 4, 2, "XF")
    net.load_state_dict(state_dict_from(g, "XF::sd::"), strict=True)
    net.to(dev).eval()
    blk = net.cascades[0]
    mk, sens, mask = cuda(g["masked_kspace"], dev), cuda(g["sens"], dev), cuda(g["mask"], dev)
    img = cuda(g["img"], dev)
    assert rel_err(blk.HOperator(img, mask, sens).cpu(), g["H_img"]) < OP_TOL
    assert rel_err(blk.xfyf_transform(img.squeeze(2)).cpu(), g["xfyf"]) < BLOCK_TOL
    out = blk.ConjGrad(cuda(g["xfyf"], dev), cuda(g["cg_rhs"], dev), mask, sens, 3)
    assert rel_err(out.cpu(), g["cg_out"]) < BLOCK_TOL
    assert rel_err(blk(img, img, mask, sens).cpu(), g["block_out"]) < MODEL_TOL


@pytest.mark.parametrize("tag,dyn,ws", [("XF", "XF", False), ("XT", "XT", False), ("2D", "2D", False), ("3D", "3D", False),
                                        ("XFws", "XF", True)])
def test_cinenet_tiny_vs_reference_golden(golden, dev, tag, dyn, ws):
    import reconstruction.models as M
    g = golden("cinenet")
    net = M.CineNet(2, 3, 4, 2, dyn, ws)
    net.load_state_dict(state_dict_from(g, f"{tag}::sd::"), strict=True)
    net.to(dev).eval()
    out = net(cuda(g["masked_kspace"], dev), cuda(g["mask"], dev), cuda(g["sens"], dev))
    assert rel_err(out.cpu(), g[f"{tag}_out"]) < MODEL_TOL


def test_cinenet_full_size_2d_vs_oracle(dev):
    """CineNet 2D, 2 cascades, CG 6, 15 coils x 15 frames x 200x200 (script widths) against the CPU oracle."""
    import reconstruction.models as M
    from oracle import cinenet_ref as C
    from cine_hip import synth
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=6, seed=2)
    hip = M.CineNet(2, 6, 16, 3, "2D").eval(); synth.fill_parameters_(hip, 3)
    ref = C.CineNet(2, 6, 16, 3, "2D").eval(); ref.load_state_dict(hip.state_dict(), strict=True)
    with torch.no_grad():
        want = ref(ex["masked_kspace"], ex["mask"], ex["sens_maps"])
    got = hip.to(dev)(ex["masked_kspace"].to(dev), ex["mask"].to(dev), ex["sens_maps"].to(dev)).cpu()
    assert rel_err(got, want) < MODEL_TOL


# ------------------------------------------------------------------ MWCNN / XPDNet
def test_mwcnn_vs_reference_golden(golden, dev):
    from reconstruction.models.denoisers import MWCNN
    g = golden("xpdnet")
    mw = MWCNN(in_chans=6, out_chans=4, n_scales=2, n_filters_per_scale=[8, 16], n_convs_per_scale=[2, 1],
               first_conv_n_filters=8)
    mw.load_state_dict(state_dict_from(g, "mw::"), strict=True)
    mw.to(dev).eval()
    assert rel_err(mw(cuda(g["mw_x"], dev)).cpu(), g["mw_y"]) < BLOCK_TOL


def test_mwcnn_default_topology_vs_oracle(dev):
    """The 3-scale default XPDNet builds (12 -> 10 channels) on cfg-3 sized planes 200 x 16."""
    from reconstruction.models.denoisers import MWCNN
    from oracle import xpdnet_ref as X
    from cine_hip import synth
    hip = MWCNN(in_chans=12, out_chans=10).eval(); synth.fill_parameters_(hip, 5, keep=())
    ref = X.MWCNN(in_chans=12, out_chans=10).eval(); ref.load_state_dict(hip.state_dict(), strict=True)
    x = rnd(9, 3, 12, 200, 16)
    with torch.no_grad():
        want = ref(x)
    assert rel_err(hip.to(dev)(x.to(dev)).cpu(), want) < BLOCK_TOL


def test_xpd_buffer_pack_unpack_vs_oracle(dev):
    from cine_hip import ops
    from oracle import xpdnet_ref as X, centered_fft as cf, complex_ops as co
    b, t, h, w, n, ns = 1, 5, 12, 10, 2, 2
    buf = rnd(1, b, t, 1, h, w, 2 * n); extra = rnd(2, b, t, 1, h, w, 2)
    for xf in (True, False):
        pxf, pyf, mean = ops.xpd_pack(buf.to(dev), extra.to(dev), n, ns, xf)
        ib = co.complex_to_real_multi_ch(torch.cat([co.real_to_complex_multi_ch(buf, n), co.real_to_complex_multi_ch(extra, 1)], -1)).squeeze(2)
        m = ib.mean(dim=1, keepdim=True)
        x = ib - m
        if xf:
            x = co.complex_to_real_multi_ch(cf.xpd_temporal_fft(co.real_to_complex_multi_ch(x, n + 1), dim=1))
        rxf, _ = X.pad_for_mwcnn(x.permute(0, 2, 4, 3, 1).reshape(b * h, 2 * (n + 1), w, t), ns)
        ryf, _ = X.pad_for_mwcnn(x.permute(0, 3, 4, 2, 1).reshape(b * w, 2 * (n + 1), h, t), ns)
        assert rel_err(pxf.cpu(), rxf) < OP_TOL and rel_err(pyf.cpu(), ryf) < OP_TOL
        # back half on synthetic "network outputs"
        oxf = rnd(3, b * h, 2 * n, *rxf.shape[2:]); oyf = rnd(4, b * w, 2 * n, *ryf.shape[2:])
        got = ops.xpd_unpack(oxf.to(dev), oyf.to(dev), mean, b, t, h, w, n, ns, xf)
        _, padx = X.pad_for_mwcnn(torch.zeros(1, 1, w, t), ns); _, pady = X.pad_for_mwcnn(torch.zeros(1, 1, h, t), ns)
        a_ = X.unpad_from_mwcnn(oxf, padx).reshape(b, h, 1, 2 * n, w, t).permute(0, 5, 2, 1, 4, 3)
        b_ = X.unpad_from_mwcnn(oyf, pady).reshape(b, w, 1, 2 * n, h, t).permute(0, 5, 2, 4, 1, 3)
        out = 0.5 * (a_ + b_)
        if xf:
            out = co.complex_to_real_multi_ch(cf.xpd_temporal_ifft(co.real_to_complex_multi_ch(out, n), dim=1))
        mm = m.unsqueeze(2)
        want = out + torch.cat([mm[..., :n], mm[..., n + 1:-1]], dim=-1)
        assert rel_err(got.cpu(), want) < OP_TOL


@pytest.mark.parametrize("tag,dyn,ws,po", [("XF", "XF", False, True), ("XT", "XT", False, True), ("2D", "2D", False, True),
                                           ("XFws", "XF", True, True), ("XFdual", "XF", False, False)])
def test_xpdnet_tiny_vs_reference_golden(golden, dev, tag, dyn, ws, po):
    import reconstruction.models as M
    g = golden("xpdnet")
    net = M.XPDNet(num_cascades=2, sens_chans=4, sens_pools=2, n_scales=2, n_filters_per_scale=[8, 16],
                   n_convs_per_scale=[1, 1], first_conv_n_filters=8, n_primal=2, dynamic_type=dyn, weight_sharing=ws,
                   primal_only=po)
    net.load_state_dict(state_dict_from(g, f"{tag}::sd::"), strict=True)
    net.to(dev).eval()
    mk, mask = cuda(g["masked_kspace"], dev), cuda(g["mask"], dev)
    if tag == "XF":
        assert rel_err(net.sens_net(mk, mask).cpu(), g["sens_out"]) < BLOCK_TOL
    out = net(mk, mask)
    assert rel_err(out.cpu(), g[f"{tag}_out"]) < MODEL_TOL
    assert torch.equal(mk.cpu(), torch.from_numpy(g["masked_kspace"]))


def test_xpdnet_rnn_dual_vs_oracle(dev):
    import reconstruction.models as M
    from oracle import recurrent_ref as R
    from cine_hip import synth
    ex = synth.make_cine_slice(5, 3, 24, 20, accel=4, center_lines=4, seed=3)
    hip = M.XPDNet_RNN(2, 4, 2, 6, False, 2, 1).eval(); synth.fill_parameters_(hip, 9, keep=())
    ref = R.XPDNet_RNN(2, 4, 2, 6, False, 2, 1).eval(); ref.load_state_dict(hip.state_dict(), strict=True)
    with torch.no_grad():
        want = ref(ex["masked_kspace"], ex["mask"])
    assert rel_err(hip.to(dev)(ex["masked_kspace"].to(dev), ex["mask"].to(dev)).cpu(), want) < MODEL_TOL


# ------------------------------------------------------------------ convolutional-RNN hybrids
def test_conv3x3_sum_epilogue_vs_torch(dev):
    """Sum of two biased convolutions + addend + ReLU as one launch (the CRNN cell update)."""
    from cine_hip import ops
    import torch.nn.functional as F
    a, b = rnd(1, 2, 16, 24, 20), rnd(2, 2, 2, 24, 20)
    wa, wb = rnd(3, 16, 16, 3, 3) / 12, rnd(4, 16, 2, 3, 3) / 4
    ba, bb, add = rnd(5, 16), rnd(6, 16), rnd(7, 2, 16, 24, 20)
    w = ops.pack_conv3x3(torch.cat([wa, wb], 1).to(dev))
    y = ops.conv3x3_sum([a.to(dev), b.to(dev)], w, (ba + bb).to(dev), 16, addend=add.to(dev), relu=True)
    ref = F.relu(F.conv2d(a, wa, ba, padding=1) + F.conv2d(b, wb, bb, padding=1) + add)
    assert rel_err(y.cpu(), ref) < OP_TOL


@pytest.mark.parametrize("n,h,w", [(1, 16, 20), (2, 40, 36), (3, 53, 200), (7, 120, 36), (9, 52, 20), (15, 200, 200), (16, 17, 68), (5, 104, 104)])
def test_column_tile_grid_decode_covers_every_tile_2d(dev, n, h, w):
    """The column-tile kernel runs on a 1-D grid whose ids are decoded XCD-aware (bands of tiles dealt over the XCDs, the last bands mod 8 as per-XCD runs, idle
    ids behind a run's end): whatever the number of bands -- fewer than 8, a multiple of 8, any remainder -- every tile is computed exactly once.  Output
    pre-filled with NaN; plain-source conv + bias + addend + ReLU against torch, and the InstanceNorm records of the statistics form against the output's own
    moments (a record that is missing or written twice shows in count / mean)."""
    from cine_hip import ops
    import torch.nn.functional as F
    c = 8
    x = rnd(1, n, c, h, w).to(dev)
    wt = (rnd(2, c, c, 3, 3) / 8).to(dev)
    bias, add = rnd(3, c).to(dev), rnd(4, n, c, h, w).to(dev)
    out = torch.full((n, c, h, w), float("nan"), device=dev)
    y = ops.conv3x3_sum([x], ops.pack_conv3x3(wt), bias, c, addend=add, relu=True, out=out)
    ref = F.relu(F.conv2d(x, wt, bias, padding=1) + add)
    assert not torch.isnan(y).any(), "a tile was never written"
    assert rel_err(y.cpu(), ref.cpu()) < OP_TOL
    yr, part = ops.conv3x3_in([(x, None, 0)], ops.pack_conv3x3(wt), c, h, w)
    assert rel_err(yr.cpu(), F.conv2d(x, wt, padding=1).cpu()) < OP_TOL
    cnt = part[..., 0].sum(-1)
    assert torch.equal(cnt, torch.full_like(cnt, float(h * w))), "statistics records do not cover the plane once"
    mean = (part[..., 0] * part[..., 1]).sum(-1) / cnt
    assert torch.allclose(mean, yr.mean((2, 3)), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("n,d,h,w", [(1, 1, 20, 20), (1, 3, 53, 36), (2, 5, 40, 68), (1, 15, 200, 200), (1, 7, 100, 100), (3, 2, 16, 20)])
def test_column_tile_grid_decode_covers_every_tile_3d(dev, n, d, h, w):
    """... and for volumes (whole bands to the end, idle ids behind the last band): 3x3x3 conv against torch, records against the output's own moments."""
    from cine_hip import ops
    import torch.nn.functional as F
    c = 8
    x = rnd(1, n, c, d, h, w).to(dev)
    wt = (rnd(2, c, c, 3, 3, 3) / 14).to(dev)
    y, part = ops.conv3d_in(x, wt)
    assert rel_err(y.cpu(), F.conv3d(x, wt, padding=1).cpu()) < OP_TOL
    cnt = part[..., 0].sum(-1)
    assert torch.equal(cnt, torch.full_like(cnt, float(d * h * w))), "statistics records do not cover the volume once"
    mean = (part[..., 0] * part[..., 1]).sum(-1) / cnt
    assert torch.allclose(mean, y.mean((2, 3, 4)), rtol=1e-4, atol=1e-5)


def test_all_frame_conv_tilings_give_the_same_bits(dev):
    """The CRNN cells' all-frame convolutions write no statistics records, so the dispatcher is free to tile them by resident rounds: 15 frames of
    200 x 200 take 40-row tiles (975 workgroups on 1 024 slots; 52-row tiles would be 780 on 768: a round plus a sliver), 14 frames keep the 52-row
    tiles (728 on 768).  Same products in the same order: frame for frame the same bits; and the torch value.  Covers the one-source form, the
    ragged cat(hidden 16, image 2) form with bias / addend / ReLU, and a 2-row output (the cells' last conv)."""
    from cine_hip import ops
    import torch.nn.functional as F
    h = w = 200
    a, b = rnd(1, 15, 16, h, w).to(dev), rnd(2, 15, 2, h, w).to(dev)
    wa, wb = (rnd(3, 16, 16, 3, 3) / 12).to(dev), (rnd(4, 16, 2, 3, 3) / 4).to(dev)
    bias, add = rnd(5, 16).to(dev), rnd(6, 15, 16, h, w).to(dev)
    w2 = ops.pack_conv3x3(torch.cat([wa, wb], 1))
    w1 = ops.pack_conv3x3(wa)
    wo = (rnd(7, 2, 16, 3, 3) / 12).to(dev)
    for srcs, wp, cout, kw, ref in (
            ([a, b], w2, 16, dict(addend=add, relu=True), lambda n: F.relu(F.conv2d(a[:n], wa, bias, padding=1) + F.conv2d(b[:n], wb, None, padding=1) + add[:n])),
            ([a], w1, 16, dict(), lambda n: F.conv2d(a[:n], wa, bias, padding=1)),
            ([a], ops.pack_conv3x3(wo), 2, dict(), lambda n: F.conv2d(a[:n], wo, bias[:2], padding=1))):
        bsel = bias[:cout].contiguous()
        y15 = ops.conv3x3_sum(srcs, wp, bsel, cout, **kw)
        kw14 = {k: (v[:14].contiguous() if torch.is_tensor(v) else v) for k, v in kw.items()}
        y14 = ops.conv3x3_sum([s_[:14].contiguous() for s_ in srcs], wp, bsel, cout, **kw14)
        assert torch.equal(y15[:14], y14), "40-row and 52-row tiles differ"
        assert rel_err(y15.cpu(), ref(15).cpu()) < OP_TOL


@pytest.mark.parametrize("c,h,w,n", [(6, 24, 20, 1), (16, 200, 200, 1), (16, 52, 16, 2), (16, 208, 208, 15)])
def test_crnn_step2_vs_torch(dev, c, h, w, n):
    """Both directions of the BCRNN time sweep in one launch: y = ReLU(conv(x) + addend) per direction, second output stored by
    the first direction to reach a frame and added to by the second; the single-direction form; the pair launch is
    bit-identical to two separate launches (shapes inside and outside the pair configuration)."""
    from cine_hip import ops
    import torch.nn.functional as F
    wt = (rnd(1, c, c, 3, 3) / (3.0 * c ** 0.5)).to(dev)
    wp = ops.pack_conv3x3(wt)
    xf, xb, af, ab = (rnd(s, n, c, h, w).to(dev) for s in (2, 3, 4, 5))
    ref_f = F.relu(F.conv2d(xf, wt, padding=1) + af)
    ref_b = F.relu(F.conv2d(xb, wt, padding=1) + ab)
    yf, yb = torch.empty_like(xf), torch.empty_like(xb)
    of, ob = torch.full_like(xf, 7.0), torch.full_like(xb, 3.0)
    ops.crnn_step2(wp, (xf, af, yf, of, True), (xb, ab, yb, ob, False))
    assert rel_err(yf.cpu(), ref_f.cpu()) < OP_TOL and rel_err(yb.cpu(), ref_b.cpu()) < OP_TOL
    assert torch.equal(of, yf) and rel_err((ob - 3.0).cpu(), ref_b.cpu()) < OP_TOL
    y1, o1 = torch.empty_like(xf), torch.full_like(xf, 7.0)
    ops.crnn_step2(wp, (xf, af, y1, o1, True))
    y2, o2 = torch.empty_like(xb), torch.full_like(xb, 3.0)
    ops.crnn_step2(wp, (xb, ab, y2, o2, False))
    assert torch.equal(y1, yf) and torch.equal(o1, of) and torch.equal(y2, yb) and torch.equal(o2, ob)
    with pytest.raises(Exception, match="same tensor"):
        ops.crnn_step2(wp, (xf, af, yf, of, True), (xb, ab, yb, of, False))


def test_rnn_models_vs_reference_golden(golden, dev):
    import reconstruction.models as M
    g = golden("rnn")
    mk, mask, sens = cuda(g["masked_kspace"], dev), cuda(g["mask"], dev), cuda(g["sens"], dev)
    net = M.VarNet_RNN(3, 4, 2, 6); net.load_state_dict(state_dict_from(g, "varnet_rnn::sd::"), strict=True)
    assert rel_err(net.to(dev).eval()(mk, mask).cpu(), g["varnet_rnn_out"]) < MODEL_TOL
    net = M.CineNet_RNN(3, 3, 6); net.load_state_dict(state_dict_from(g, "cinenet_rnn::sd::"), strict=True)
    assert rel_err(net.to(dev).eval()(mk, mask, sens).cpu(), g["cinenet_rnn_out"]) < MODEL_TOL
    net = M.XPDNet_RNN(3, 4, 2, 6, True, 2, 1); net.load_state_dict(state_dict_from(g, "xpdnet_rnn::sd::"), strict=True)
    assert rel_err(net.to(dev).eval()(mk, mask).cpu(), g["xpdnet_rnn_out"]) < MODEL_TOL
    assert torch.equal(mk.cpu(), torch.from_numpy(g["masked_kspace"]))


# ------------------------------------------------------------------ 3-D path
@pytest.mark.parametrize("cin,cout,d,h,w", [(2, 16, 5, 24, 20), (16, 16, 15, 40, 32), (32, 64, 3, 13, 9), (64, 128, 1, 6, 5), (5, 7, 4, 9, 11)])
def test_conv3d_vs_torch(dev, cin, cout, d, h, w):
    from cine_hip import ops, _lib
    import torch.nn.functional as F
    n = 2
    x = rnd(cin + d, n, cin, d, h, w); wt = rnd(cout, cout, cin, 3, 3, 3) / (5 * cin ** 0.5)
    L = _lib.lib()
    wp = ops._pack("c27", wt.to(dev))
    y = torch.empty((n, cout, d, h, w), device=dev)
    npart = L.cine_conv_stat_partials3d(cout, d, h, w, 0)
    part = torch.empty((n, cout, npart, 3), device=dev)
    xd = x.to(dev)
    _lib.check(L.cine_conv3d_in(xd.data_ptr(), None, 0, cin, 0, d, h, w, None, None, 0, 0, 0, 0, 0, 0, wp.data_ptr(), None, None, 0,
                                y.data_ptr(), part.data_ptr(), n, cout, d, h, w, 1e-5, 0.2, torch.cuda.current_stream().cuda_stream))
    ref = F.conv3d(x, wt, padding=1)
    assert rel_err(y.cpu(), ref) < OP_TOL
    st = ops.instnorm_finalize(part)
    assert rel_err(st[..., 0].cpu(), ref.mean(dim=(2, 3, 4))) < 1e-4
    assert rel_err(st[..., 1].cpu(), 1 / torch.sqrt(ref.var(dim=(2, 3, 4), unbiased=False) + 1e-5)) < 1e-4


def test_two_threads_two_streams_are_independent(dev):
    """include/cine_hip.h, "re-entrant": what a call computes depends on its arguments only, and the settings that outlive a call (error
    string, side stream, diagnostic kernel-selection mask) belong to the calling thread.  The contract behind it is the reference's
    Lightning `dp` strategy (traintest_scripts/varnet/train_test_varnet.py:148,290: one Python thread per device in one process).
    Two threads, each with its own model and stream, run concurrently: thread A reconstructs with ANOTHER LeakyReLU slope and flips
    cine_set_conv_plane between iterations (general <-> lean kernels, bit-identical by construction); thread B runs forward + backward
    training steps with the defaults.  Every result must equal, bit for bit, what the same work gives alone on the main thread."""
    import threading
    import reconstruction.models as M
    from cine_hip import ops, synth
    from cine_hip._lib import lib
    exa = synth.make_cine_slice(5, 3, 24, 20, accel=4, center_lines=4, seed=0)
    exb = synth.make_cine_slice(6, 4, 32, 24, accel=4, center_lines=4, seed=1)
    exa = {k: v.to(dev) for k, v in exa.items() if torch.is_tensor(v)}
    exb = {k: v.to(dev) for k, v in exb.items() if torch.is_tensor(v)}
    neta = M.VarNet(2, 4, 2, 4, 2, "XF").eval(); synth.fill_parameters_(neta, 1); neta.to(dev)
    netb = M.CineNet(2, 3, 4, 2, "XT").train(); synth.fill_parameters_(netb, 2); netb.to(dev)
    iters = 6

    def work_a(flip):
        outs = []
        with torch.no_grad(), ops.activation(slope=0.5):
            for i in range(iters):
                if flip:
                    cine_ops.set_conv_plane(0 if i % 2 == 0 else 7)
                outs.append(neta(exa["masked_kspace"], exa["mask"]).clone())
        return outs

    def work_b():
        outs = []
        for _ in range(iters):
            netb.zero_grad()
            with torch.enable_grad():
                y = netb(exb["masked_kspace"], exb["mask"], exb["sens_maps"])
                ((y - exb["target"]) ** 2).sum().backward()
            outs.append([y.detach().clone()] + [p.grad.detach().clone() for p in netb.parameters()])
        return outs

    want_a = work_a(False)
    want_b = work_b()
    with torch.no_grad():
        default_a = neta(exa["masked_kspace"], exa["mask"])
    assert not torch.equal(default_a, want_a[0])             # the slope argument does change the result
    torch.cuda.synchronize()
    res, errs, gate = {}, [], threading.Barrier(2)

    def runner(name, fn, *args):
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                gate.wait(timeout=60)
                res[name] = fn(*args)
                st.synchronize()
        except Exception as e:                                # noqa: BLE001 -- reported by the main thread
            errs.append((name, repr(e)))
    ta = threading.Thread(target=runner, args=("a", work_a, True))
    tb = threading.Thread(target=runner, args=("b", work_b))
    ta.start(); tb.start(); ta.join(300); tb.join(300)
    assert not errs and not ta.is_alive() and not tb.is_alive(), errs
    for got, want in zip(res["a"], want_a):
        assert torch.equal(got, want)
    for got, want in zip(res["b"], want_b):
        for g_, w_ in zip(got, want):
            assert torch.equal(g_, w_)
    # thread A's mask and slope died with it: the main thread still reconstructs with its defaults
    with torch.no_grad():
        assert torch.equal(neta(exa["masked_kspace"], exa["mask"]), default_a)


@pytest.mark.parametrize("case", ["plain16", "ragged50", "partial_chunk", "norm_concat", "pooled", "pooled_odd", "concat_unaligned", "bias_relu",
                                  "coarse_l2", "coarse_l2_pooled", "coarse_l2_up", "coarse_l3", "coarse_l3_pooled", "coarse_ragged", "coarse_bias_relu"])
def test_conv3d_v3_paths_vs_torch(dev, case):
    """The V3 form of the 3x3x3 convolution (three 3x3 passes on the 2-D kernel) at volumes large enough for its regular tiles:
    vectorised staging of plain / normalised / 2x2x2-pooled sources, ragged row widths (4-byte aligned loads, shifted last
    piece), a channel chunk that is only partly filled, concatenation, the element-wise fallback inside the V3 kernel
    (first source not a multiple of 8 channels), bias + ReLU epilogue, depth-boundary slices skipped."""
    from cine_hip import ops, _lib
    import torch.nn.functional as F
    L = _lib.lib()
    n = 1
    cfgs = {  # c0, mode0, c1, mode1, cout, d, h, w
        "plain16": (16, 0, 0, 0, 16, 16, 104, 64), "ragged50": (8, 1, 0, 0, 32, 16, 104, 50), "partial_chunk": (12, 1, 0, 0, 16, 16, 104, 64),
        "norm_concat": (16, 1, 16, 1, 16, 16, 104, 64), "pooled": (16, 2, 0, 0, 32, 16, 104, 52), "pooled_odd": (8, 2, 0, 0, 32, 16, 104, 25),
        "concat_unaligned": (12, 1, 4, 0, 16, 16, 104, 64), "bias_relu": (16, 0, 0, 0, 16, 16, 104, 64),
        # the coarse levels of cfg 4's 3-D U-Net (conv_coarse.hip: flattened positions, K split over the waves): 64 -> 64 on 3 x 50 x 50,
        # the pooled 32 -> 64 from 7 x 100 x 100, the up path's cat(2 x 50 x 50 zero-padded, 3 x 50 x 50), 128 -> 128 and the pooled
        # 64 -> 128 on 1 x 25 x 25, a ragged layer (44 input channels over two sources, 72 rows, odd sizes), bias + ReLU
        "coarse_l2": (64, 1, 0, 0, 64, 3, 50, 50), "coarse_l2_pooled": (32, 2, 0, 0, 64, 3, 50, 50), "coarse_l2_up": (64, 1, 64, 1, 64, 3, 50, 50),
        "coarse_l3": (128, 1, 0, 0, 128, 1, 25, 25), "coarse_l3_pooled": (64, 2, 0, 0, 128, 1, 25, 25), "coarse_ragged": (30, 1, 14, 0, 72, 2, 21, 19),
        "coarse_bias_relu": (20, 0, 0, 0, 64, 2, 30, 34)}
    c0, m0, c1, m1, cout, d, h, w = cfgs[case]
    def source(seed, c, mode):
        if c == 0:
            return None, None, 0, (0, 0, 0), None
        dd, hh, ww = (2 * d, 2 * h, 2 * w + (1 if case == "pooled_odd" else 0)) if mode == 2 else (d, h, w)
        if case.startswith("coarse") and mode == 2:
            dd += 1                                     # avg_pool3d floors: 7 -> 3 slices at cfg 4
        short = case == "coarse_l2_up" and seed == 11   # the transpose conv's output ends one slice early (unet.py:106-120: zero pad at the end)
        if short:
            dd -= 1
        x = rnd(seed, n, c, dd, hh, ww) * 1.5 + 0.3
        part = ops.instnorm_partials(x.to(dev)) if mode else None
        xr = x
        if mode:
            xr = F.leaky_relu(F.instance_norm(x), 0.2)
            if mode == 2:
                xr = F.avg_pool3d(xr, 2, 2)
        if short:
            xr = F.pad(xr, [0, 0, 0, 0, 0, 1])
        return x.to(dev), part, 1 if mode else 0, (dd, hh, ww), xr
    x0, p0, np0, e0, r0 = source(11, c0, m0)
    x1, p1, np1, e1, r1 = source(12, c1, m1)
    cin = c0 + c1
    wt = rnd(13, cout, cin, 3, 3, 3) / (5 * cin ** 0.5)
    bias = rnd(14, cout) if case.endswith("bias_relu") else None
    wp = ops._pack("c27", wt.to(dev))
    y = torch.empty((n, cout, d, h, w), device=dev)
    part = torch.empty((n, cout, L.cine_conv_stat_partials3d(cout, d, h, w, 0), 3), device=dev)
    bd = None if bias is None else bias.to(dev)
    P = lambda t_: None if t_ is None else t_.data_ptr()
    _lib.check(L.cine_conv3d_in(P(x0), P(p0), np0, c0, m0, e0[0], e0[1], e0[2], P(x1), P(p1), np1, c1, m1, e1[0], e1[1], e1[2],
                                wp.data_ptr(), P(bd), None, 1 if bias is not None else 0,
                                y.data_ptr(), part.data_ptr(), n, cout, d, h, w, 1e-5, 0.2, torch.cuda.current_stream().cuda_stream))
    xin = r0 if r1 is None else torch.cat([r0, r1], 1)
    ref = F.conv3d(xin, wt, bias, padding=1)
    if bias is not None:
        ref = F.relu(ref)
    assert rel_err(y.cpu(), ref) < 2 * OP_TOL
    st = ops.instnorm_finalize(part)
    assert rel_err(st[..., 0].cpu(), ref.mean(dim=(2, 3, 4))) < 1e-4
    if case.startswith("coarse"):
        assert rel_err(st[..., 1].cpu(), 1 / torch.sqrt(ref.var(dim=(2, 3, 4), unbiased=False) + 1e-5)) < 1e-4
        want_np = {"coarse_l2": 240, "coarse_l3": 41}.get(case)      # one record per position tile of the flattened slices: the coarse kernel ran
        assert want_np is None or part.shape[2] == want_np


@pytest.mark.parametrize("shape", [(3, 15, 200, 200), (2, 7, 10, 13), (5, 4, 6, 16)])
def test_pool3d_act_vs_torch(dev, shape):
    """cine_pool3d_act: avg_pool3d(LeakyReLU(InstanceNorm3d(x)), 2) materialised (the 3-D U-Net's level-1 input at cfg 4: 15 x 200 x 200 ->
    7 x 100 x 100; odd extents floor), vectorised and element-wise forms, against torch."""
    from cine_hip import ops, _lib
    import torch.nn.functional as F
    c, d, h, w = shape
    x = rnd(61, 1, c, d, h, w) * 1.7 + 0.4
    xd = x.to(dev)
    part = ops.instnorm_partials(xd)
    y = torch.empty((1, c, d // 2, h // 2, w // 2), device=dev)
    _lib.check(_lib.lib().cine_pool3d_act(xd.data_ptr(), part.data_ptr(), part.shape[2], y.data_ptr(), c, d, h, w, 1e-5, 0.2, ops._stream()), "cine_pool3d_act")
    ref = F.avg_pool3d(F.leaky_relu(F.instance_norm(x), 0.2), 2, 2)
    assert rel_err(y.cpu(), ref) < OP_TOL
    L = _lib.lib()
    assert L.cine_conv3d_pools_on_load(64, 3, 50, 50) == 1 and L.cine_conv3d_pools_on_load(32, 7, 100, 100) == 0


def test_unet3d_and_normunet3d_vs_reference_golden(golden, dev):
    from reconstruction.models.denoisers import NormUnet3D
    g = golden("unet")
    nu3 = NormUnet3D(4, 2); nu3.load_state_dict(state_dict_from(g, "nu3::"), strict=True); nu3.to(dev).eval()
    assert rel_err(nu3(cuda(g["nu3_x"], dev)).cpu(), g["nu3_y"]) < BLOCK_TOL


def test_cinenet_3d_cfg4_shape_vs_oracle(dev):
    """BASELINE configs[3] shape: 3-D CineNet (1 cascade for test time), CG 6, 15 coils x 15 frames x 200x200, R=6."""
    import reconstruction.models as M
    from oracle import cinenet_ref as C
    from cine_hip import synth
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=6, seed=4)
    hip = M.CineNet(1, 6, 16, 3, "3D").eval(); synth.fill_parameters_(hip, 7)
    ref = C.CineNet(1, 6, 16, 3, "3D").eval(); ref.load_state_dict(hip.state_dict(), strict=True)
    with torch.no_grad():
        want = ref(ex["masked_kspace"], ex["mask"], ex["sens_maps"])
    got = hip.to(dev)(ex["masked_kspace"].to(dev), ex["mask"].to(dev), ex["sens_maps"].to(dev)).cpu()
    assert rel_err(got, want) < MODEL_TOL


# ------------------------------------------------------------------ full-size BASELINE configs 3 / 4 / 5 vs the reference's fingerprints
def _check_full_fingerprint(out, g, ex):
    """max|d|/peak <= 1e-4, NMSE <= 1e-8, |dSSIM| <= 1e-4 against the reference's strided output (tests/golden/make_golden.py)."""
    from reconstruction.utils import evaluate
    ref = torch.from_numpy(g["out_strided"])
    got = out[:, :, ::4, ::4]
    assert rel_err(got, ref) < MODEL_TOL
    assert float(((got - ref).double() ** 2).sum() / (ref.double() ** 2).sum()) < 1e-8
    assert abs(float(out.double().sum()) - float(g["out_sum"])) / float(g["out_sum"]) < 1e-5
    assert abs(float(out.double().norm()) - float(g["out_l2"])) / float(g["out_l2"]) < 1e-5
    tgt = ex["target"][0, :, ::4, ::4].numpy()
    assert abs(evaluate.ssim(tgt, got[0].numpy()) - evaluate.ssim(tgt, ref[0].numpy())) < 1e-4


def _cfg3(g):
    from cine_hip import synth
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=int(g["accel"]), seed=int(g["data_seed"]), noise_std=float(g["noise_std"]))
    kw = dict(num_cascades=10, sens_chans=8, sens_pools=3, n_primal=5, dynamic_type="XT")
    return ex, kw


def test_xpdnet_cfg3_vs_reference_golden(golden, dev):
    """BASELINE configs[2]: XT-XPDNet with the MWCNN regulariser, 10 cascades, n_primal 5, 15 coils x 15 frames x 200x200, R=8.

    An untrained 10-cascade XPDNet is expansive: the reference's own fp32 output sits 7e-4 (max|d|/peak; NMSE 3e-7) away from
    the same network evaluated in fp64 (`ref_fp32_vs_fp64` in the golden file, measured by tests/golden/make_golden.py), so
    1e-4 is below the reproducibility floor of ANY fp32 implementation here.  Whole-model bar: the HIP output is as close to
    the fp64 result as the reference's fp32 output is (factor 2), and as close to the reference's fp32 output as two fp32
    evaluations can be (factor 3); |dSSIM| <= 1e-4 holds as stated.  The per-cascade test below holds the 1e-4 bar."""
    import reconstruction.models as M
    from reconstruction.utils import evaluate
    from cine_hip import synth
    g = golden("xpdnet_cfg3")
    ex, kw = _cfg3(g)
    net = M.XPDNet(**kw)
    synth.fill_parameters_(net, int(g["weight_seed"]), keep=())
    net.to(dev).eval()
    out = net(ex["masked_kspace"].to(dev), ex["mask"].to(dev)).cpu()
    got = out[:, :, ::4, ::4]
    ref32, ref64 = torch.from_numpy(g["out_strided"]), torch.from_numpy(g["out64_strided"])
    floor = float(g["ref_fp32_vs_fp64"])
    assert 1e-4 < floor < 5e-3                                        # the premise of the widened bar
    assert rel_err(got, ref64) < 2 * floor
    assert rel_err(got, ref32) < 3 * floor
    nmse = lambda a, b: float(((a - b).double() ** 2).sum() / (b.double() ** 2).sum())
    assert nmse(got, ref64) < 4 * float(g["ref_fp32_vs_fp64_nmse"])
    assert abs(float(out.double().sum()) - float(g["out_sum"])) / float(g["out_sum"]) < 1e-4
    tgt = ex["target"][0, :, ::4, ::4].numpy()
    assert abs(evaluate.ssim(tgt, got[0].numpy()) - evaluate.ssim(tgt, ref32[0].numpy())) < 1e-4


def test_xpdnet_cfg3_every_cascade_vs_oracle(golden, dev):
    """cfg 3 at the 1e-4 bar, cascade by cascade: the CPU oracle (pinned to the reference on this very configuration by
    tests/test_synth_golden.py::test_oracle_cfg3_full) runs the 10 cascades once; each HIP cascade (K step: masked forward
    operator minus k_ref; I step: backward operator, buffer pack, two full-size MWCNNs, unpack) starts from the oracle's
    buffer and must reproduce the oracle's next buffer, so every one of the 20 MWCNNs is checked at full size."""
    import reconstruction.models as M
    from cine_hip import ops, synth
    from oracle import xpdnet_ref as X
    g = golden("xpdnet_cfg3")
    ex, kw = _cfg3(g)
    hip = M.XPDNet(**kw).eval(); synth.fill_parameters_(hip, int(g["weight_seed"]), keep=())
    ref = X.XPDNet(**kw).eval(); ref.load_state_dict(hip.state_dict(), strict=True)
    hip.to(dev)
    mk, mask = ex["masked_kspace"], ex["mask"]
    mkd, maskd = mk.to(dev), mask.to(dev)
    n = 5
    with torch.no_grad():
        sens = ref.sens_net(mk, mask)
        sens_d = hip.sens_net(mkd, maskd)
        assert rel_err(sens_d.cpu(), sens) < BLOCK_TOL
        image = X.backward_operator(mk, mask, sens, 1, False)
        assert rel_err(ops.sens_reduce(mkd, sens_d).cpu(), image) < OP_TOL
        ib, kb = torch.repeat_interleave(image, n, dim=-1), mk
        hyb = torch.empty_like(mkd)
        worst = 0.0
        for i, dom in enumerate(ref.domain_sequence):
            prev = ib
            ib, kb = ref.cascades[i](dom, i, ib, kb, mk, mask, sens)
            if dom != 'I':
                continue
            prev_d = prev.to(dev)
            ops.expand_resid_hybrid(ops.extract_complex(prev_d, 0, n), sens_d, mkd, maskd, out=hyb)
            got = hip.cascades[i].regularise(i, prev_d, ops.hybrid_reduce(hyb, sens_d)).cpu()
            err = rel_err(got, ib)
            worst = max(worst, err)
            assert err < MODEL_TOL, (i // 2, err)
        out = ops.complex_abs(ops.extract_complex(ib.to(dev), 0, n).squeeze(2)).cpu()
        # the teacher chain is the reference's chain (up to the fp32 reproducibility floor on this host's thread count)
        assert rel_err(out[:, :, ::4, ::4], g["out_strided"]) < 3 * float(g["ref_fp32_vs_fp64"])
    print(f"cfg3 worst per-cascade error {worst:.2e}")


def test_cinenet_cfg4_vs_reference_golden(golden, dev):
    """BASELINE configs[3]: 3D CineNet, all 6 cascades, CG 6, 15 coils x 15 frames x 200x200, R=6."""
    import reconstruction.models as M
    from cine_hip import synth
    g = golden("cinenet_cfg4")
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=int(g["accel"]), seed=int(g["data_seed"]))
    net = M.CineNet(6, 6, 16, 3, "3D")
    synth.fill_parameters_(net, int(g["weight_seed"]))
    net.to(dev).eval()
    out = net(ex["masked_kspace"].to(dev), ex["mask"].to(dev), ex["sens_maps"].to(dev)).cpu()
    _check_full_fingerprint(out, g, ex)


def test_rnn_cfg5_vs_reference_golden(golden, dev):
    """BASELINE configs[4]: CRNN-VarNet, 5 cascades, 15 coils x 15 frames x 200x200, R=8."""
    import reconstruction.models as M
    from cine_hip import synth
    g = golden("rnn_cfg5")
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=int(g["accel"]), seed=int(g["data_seed"]))
    net = M.VarNet_RNN(5, 8, 3, 16)
    synth.fill_parameters_(net, int(g["weight_seed"]))
    net.to(dev).eval()
    out = net(ex["masked_kspace"].to(dev), ex["mask"].to(dev)).cpu()
    _check_full_fingerprint(out, g, ex)


# ------------------------------------------------------------------ hipGraph replay == eager == reference
def test_cfg2_graph_replay_bitexact_and_vs_golden(golden, dev):
    """Capture the cfg-2 forward in a hipGraph on a side stream (as bench.py does), replay it twice: both replays
    equal the eager output bit for bit, and the replayed output matches the reference's cfg-2 fingerprint."""
    import reconstruction.models as M
    from cine_hip import synth
    g = golden("varnet_cfg2")
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=4, seed=int(g["data_seed"]))
    net = M.VarNet(6, 8, 3, 16, 3, "XF")
    synth.fill_parameters_(net, int(g["weight_seed"]))
    net.to(dev).eval()
    mk, mask = ex["masked_kspace"].to(dev), ex["mask"].to(dev)
    acs = net.sens_net.acs_window(mask)
    eager = net(mk, mask, acs=acs).clone()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        net(mk, mask, acs=acs)                       # warm this stream's caches outside capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=st):
        gout = net(mk, mask, acs=acs)
    for _ in range(2):
        gout.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(gout, eager)
    _check_full_fingerprint(gout.cpu(), g, ex)


def test_utils_helpers_backpropagate_on_gpu_tensors(dev):
    """User code written against the reference (a loss built from rss_complex, complex_mul, fft2c ...) differentiates through its utils:
    on a GPU tensor that requires grad every shim must return a tensor with a grad_fn, and the gradient must equal what torch.autograd
    derives from the reference's tensor expressions (utils/math.py:20-62, coil_combine.py, fftc.py:13-163, padding.py:22-47) on the CPU."""
    import torch.nn.functional as F
    import reconstruction.utils as U

    def ref_fft(x, two_d, inverse):
        c = torch.view_as_complex(x.contiguous())
        dims = (-2, -1) if two_d else (-1,)
        c = torch.fft.ifftshift(c, dim=dims)
        c = (torch.fft.ifftn if inverse else torch.fft.fftn)(c, dim=dims, norm="ortho")
        return torch.view_as_real(torch.fft.fftshift(c, dim=dims))
    cmul = lambda a, b: torch.view_as_real(torch.view_as_complex(a.contiguous()) * torch.view_as_complex(b.contiguous()))
    cases = {
        "complex_mul": (lambda x, y: U.complex_mul(x, y), lambda x, y: cmul(x, y.expand(3, 4, 6, 5, 2)), ((3, 4, 6, 5, 2), (1, 4, 1, 5, 2))),
        "complex_conj": (lambda x: U.complex_conj(x), lambda x: x * x.new_tensor([1.0, -1.0]), ((3, 6, 5, 2),)),
        "complex_abs": (lambda x: U.complex_abs(x), lambda x: (x * x).sum(-1).sqrt(), ((3, 6, 5, 2),)),
        "complex_abs_sq": (lambda x: U.complex_abs_sq(x), lambda x: (x * x).sum(-1), ((3, 6, 5, 2),)),
        "rss": (lambda x: U.rss(x, 1), lambda x: (x * x).sum(1).sqrt(), ((2, 4, 6, 5),)),
        "rss_complex": (lambda x: U.rss_complex(x, 1), lambda x: (x * x).sum(-1).sum(1).sqrt(), ((2, 4, 6, 5, 2),)),
        "fftshift": (lambda x: U.fftshift(x, [-3, -2]), lambda x: torch.roll(x, (3, 2), (-3, -2)), ((2, 7, 5, 2),)),
        "ifftshift": (lambda x: U.ifftshift(x, [-3, -2]), lambda x: torch.roll(x, (4, 3), (-3, -2)), ((2, 7, 5, 2),)),
        "pad_for_mwcnn": (lambda x: U.pad_for_mwcnn(x, 3)[0], lambda x: F.pad(x, [2, 1, 3, 3]), ((2, 3, 10, 13),)),
        "fft2c": (lambda x: U.fft2c(x), lambda x: ref_fft(x, True, False), ((3, 12, 10, 2),)),
        "ifft2c": (lambda x: U.ifft2c(x), lambda x: ref_fft(x, True, True), ((3, 9, 10, 2),)),
        "fft1c": (lambda x: U.fft1c(x), lambda x: ref_fft(x, False, False), ((4, 15, 2),)),
        "ifft1c": (lambda x: U.ifft1c(x, norm=None), lambda x: ref_fft(x, False, True) * 15 ** -0.5, ((4, 15, 2),)),
    }
    for name, (hip_fn, ref_fn, shapes) in cases.items():
        xs = [rnd(700 + i, *sh) + 0.1 for i, sh in enumerate(shapes)]
        cpu = [x.clone().requires_grad_(True) for x in xs]
        gpu = [x.to(dev).requires_grad_(True) for x in xs]
        with torch.enable_grad():                                 # (the suite runs with autograd off by default: conftest.py)
            want = ref_fn(*cpu)
            got = hip_fn(*gpu)
            assert got.grad_fn is not None, f"{name}: the result is cut off the autograd graph"
            assert rel_err(got.detach().cpu(), want.detach()) < OP_TOL, name
            w = rnd(720, *want.shape)
            (want * w).sum().backward()
            (got * w.to(dev)).sum().backward()
        for a, b in zip(gpu, cpu):
            assert a.grad is not None and rel_err(a.grad.cpu(), b.grad) < 2 * OP_TOL, name
    # without requires_grad (inference, Lightning's validation) the raw kernels run: no graph
    with torch.no_grad():
        assert U.complex_abs(xs[0].to(dev)).grad_fn is None


# ------------------------------------------------------------------ utils.math / coil_combine vs the reference's cm_* vectors
def test_complex_math_helpers_vs_reference_golden(golden, dev):
    import reconstruction.utils as U
    g = golden("ops")
    x, y = cuda(g["cm_x"], dev), cuda(g["cm_y"], dev)
    assert rel_err(U.complex_mul(x, y).cpu(), g["cm_mul"]) < OP_TOL
    assert torch.equal(U.complex_conj(x).cpu(), torch.from_numpy(g["cm_conj"]))
    assert rel_err(U.complex_abs(x).cpu(), g["cm_abs"]) < OP_TOL
    assert rel_err(U.complex_abs_sq(x).cpu(), g["cm_abs_sq"]) < OP_TOL
    assert rel_err(U.rss(x, dim=1).cpu(), g["cm_rss"]) < OP_TOL
    assert rel_err(U.rss_complex(x, dim=1).cpu(), g["cm_rss_complex"]) < OP_TOL
    from cine_hip import ops
    assert rel_err(ops.complex_abs(x).cpu(), g["cm_abs"]) < OP_TOL
    with pytest.raises(ValueError):
        U.complex_abs(x[..., :1])
    with pytest.raises(ValueError):
        U.complex_mul(x[..., :1], y)
    # the helpers are HIP kernels on GPU tensors (csrc/ew_kernels.hip): bit-exact against the reference's unfused tensor expressions
    # evaluated on the host, with broadcasting (a (b, t, 1, h, w, 2) image times (b, 1, c, h, w, 2) maps), a non-contiguous operand, every
    # rolled / reduced dimension, and pads with the odd element on the left
    gen = torch.Generator().manual_seed(5)
    img, maps = torch.randn(2, 3, 1, 9, 7, 2, generator=gen), torch.randn(2, 1, 4, 9, 7, 2, generator=gen)
    ref_mul = torch.stack([img[..., 0] * maps[..., 0] - img[..., 1] * maps[..., 1], img[..., 0] * maps[..., 1] + img[..., 1] * maps[..., 0]], -1)
    assert torch.equal(U.complex_mul(img.to(dev), maps.to(dev)).cpu(), ref_mul)
    nc = torch.randn(7, 9, 2, 2, generator=gen).permute(2, 1, 0, 3)                  # non-contiguous (2, 9, 7, 2)
    assert torch.equal(U.complex_conj(nc.to(dev)).cpu(), nc * torch.tensor([1.0, -1.0]))
    assert torch.equal(U.complex_abs_sq(maps.to(dev)).cpu(), maps[..., 0] * maps[..., 0] + maps[..., 1] * maps[..., 1])
    for d in range(5):
        want = torch.zeros_like(maps.select(d, 0)[..., 0])
        for k in range(maps.shape[d]):
            v = maps.select(d, k)
            want = want + (v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1])
        assert rel_err(U.rss_complex(maps.to(dev), dim=d).cpu(), want.sqrt()) < 2e-7, d        # (the device square root is within 1 ulp)
    for d in (0, 3, -1):
        want = torch.zeros_like(maps.select(d, 0))
        for k in range(maps.shape[d]):
            want = want + maps.select(d, k) * maps.select(d, k)
        assert rel_err(U.rss(maps.to(dev), dim=d).cpu(), want.sqrt()) < 2e-7, d
    for shift, dims in (([1], [0]), ([4, 3], [-3, -2]), ([-2, 5, 1], [1, 3, 4]), ([9], [3])):
        assert torch.equal(U.roll(maps.to(dev), shift, dims).cpu(), torch.roll(maps, shift, dims))
    from reconstruction.utils.padding import pad_for_mwcnn, unpad_from_mwcnn
    for shape, ns in (((3, 2, 9, 7), 2), ((2, 16, 8), 3), ((5, 13, 15), 3)):
        t_ = torch.randn(*shape, generator=gen)
        got, pads = pad_for_mwcnn(t_.to(dev), ns)
        want, pads_h = pad_for_mwcnn(t_, ns)
        assert pads == pads_h and torch.equal(got.cpu(), want) and torch.equal(unpad_from_mwcnn(got, pads).cpu(), t_)


# ------------------------------------------------------------------ image-space data consistency (cine_image_dc)
def _row_mask(t, h, seed, keep=0.35):
    g = torch.Generator().manual_seed(seed)
    m = (torch.rand(1, t, 1, h, 1, 1, generator=g) < keep).byte()
    m[:, :, :, h // 2] = 1
    return m


@pytest.mark.parametrize("t,c,h,w", [(5, 3, 24, 20), (2, 1, 7, 5), (3, 4, 15, 33), (1, 2, 400, 9),      # mixed-radix engine (odd / even stage counts)
                                     (2, 3, 96, 120), (1, 2, 512, 12), (1, 3, 18, 512), (1, 2, 256, 320),
                                     (2, 2, 7, 11), (1, 2, 77, 26), (1, 2, 398, 14),                         # direct-DFT engine (other prime factors)
                                     (2, 15, 200, 200), (1, 1, 200, 37), (3, 16, 200, 8), (2, 17, 200, 203), (1, 5, 200, 1)])
def test_image_dc_vs_oracle(dev, t, c, h, w):
    """cine_image_dc == sens_reduce(DC(sens_expand(x))) of reference varnet.py:181-194, 281-282 for row masks: soft DC
    for several lambdas, CineNet's normal operator (cinenet.py:121-133), XPDNet's backward image (xpdnet.py:128-167),
    and the magnitude output; ragged widths, coil counts around the kernel's coil-slot grouping, both FFT engines."""
    from cine_hip import ops
    from oracle import varnet_ref as V
    import torch.nn.functional as F
    img, sens, kref = rnd(1, 1, t, 1, h, w, 2), rnd(2, 1, 1, c, h, w, 2), rnd(3, 1, t, c, h, w, 2)
    mask = _row_mask(t, h, 4)
    imgd, sensd, maskd = img.to(dev), sens.to(dev), mask.to(dev)
    zf = V.VarNetBlock.sens_reduce(kref * mask, sens)
    zfd = ops.sens_reduce((kref * mask).to(dev), sensd)
    assert rel_err(zfd.cpu(), zf) < OP_TOL
    # zero-filled term through the masked hybrid pass, on k-space that is NOT pre-masked (unsampled rows must be ignored)
    hyb = ops.kspace_to_hybrid(kref.to(dev), mask=maskd)
    assert rel_err(ops.hybrid_reduce(hyb, sensd).cpu(), zf) < OP_TOL
    kth = V.VarNetBlock.sens_expand(img, sens)
    for lam in (-1.3, 0.5413, 25.0):
        lam_t = torch.tensor([lam])
        v = F.softplus(lam_t)
        ref = V.VarNetBlock.sens_reduce((1 - mask) * kth + mask * (kth + v * kref) / (1 + v), sens)
        got = ops.image_dc(imgd, sensd, zfd, maskd, lam_t.to(dev))
        assert got.shape == ref.shape and rel_err(got.cpu(), ref) < OP_TOL, lam
    ref = V.VarNetBlock.sens_reduce(kth * mask + 0.0, sens)                       # A^H M A x
    assert rel_err(ops.image_dc(imgd, sensd, None, maskd, weights=(1.0, 0.0, 0.0)).cpu(), ref) < OP_TOL
    ref = V.VarNetBlock.sens_reduce((kth - kref) * mask + 0.0, sens)              # A^H M (A x - k_ref)
    assert rel_err(ops.image_dc(imgd, sensd, zfd, maskd, weights=(1.0, 0.0, -1.0)).cpu(), ref) < OP_TOL
    lam_t = torch.tensor([0.2])
    v = F.softplus(lam_t)
    ref = V.VarNetBlock.sens_reduce((1 - mask) * kth + mask * (kth + v * kref) / (1 + v), sens).squeeze(2).pow(2).sum(-1).sqrt()
    got = ops.image_dc(imgd, sensd, zfd, maskd, lam_t.to(dev), magnitude=True)
    assert got.shape == (1, t, h, w) and rel_err(got.cpu(), ref) < OP_TOL
    assert torch.equal(imgd.cpu(), img)                                           # input untouched
    # CineNet's H operator with its regulariser weight in the same chain (cinenet.py:121-133): A^H M A x + softplus(lambda) x
    for lam in (-1.3, 0.5413):
        lam_t = torch.tensor([lam])
        ref = V.VarNetBlock.sens_reduce(kth * mask + 0.0, sens) + F.softplus(lam_t) * img
        got = ops.normal_op(imgd, sensd, maskd, lam_t.to(dev))
        assert got.shape == ref.shape and rel_err(got.cpu(), ref) < OP_TOL, lam
        two = ops.axpby_dev(ops.image_dc(imgd, sensd, None, maskd, weights=(1.0, 0.0, 0.0)), imgd.view_as(got), lambda_reg=lam_t.to(dev))
        assert torch.equal(got, two)                                              # bit-identical to the two-kernel form


def test_cg_step_vs_separate_kernels(dev):
    """cine_cg_step (three launches) == dot + axpby_dev sequence of reference cinenet.py:155-169, bit for bit, at cfg-4 size."""
    from cine_hip import ops
    n = 15 * 200 * 200 * 2
    x, r, p, d = (rnd(s_, n).to(dev) for s_ in (1, 2, 3, 4))
    d = d + 3.0 * p                                                               # keep p.d away from zero
    rr_old = ops.dot(r, r)
    x2, r2, p2 = x.clone(), r.clone(), p.clone()
    pd = ops.dot(p2, d)
    ops.axpby_dev(x2, p2, num=rr_old, den=pd, out=x2)
    ops.axpby_dev(r2, d, num=rr_old, den=pd, sign=-1.0, out=r2)
    rr_new2 = ops.dot(r2, r2)
    ops.axpby_dev(r2, p2, num=rr_new2, den=rr_old, out=p2)
    rr_new = torch.empty_like(rr_old)
    ops.cg_step(x, r, p, d, rr_old, rr_new)
    assert torch.equal(x, x2) and torch.equal(r, r2) and torch.equal(p, p2) and torch.equal(rr_new, rr_new2)
    ref = (r2.double() ** 2).sum()
    assert abs(float(rr_new) - float(ref)) / float(ref) < 1e-5
    with pytest.raises(Exception, match="different scalars"):
        ops.cg_step(x, r, p, d, rr_old, rr_old)


def test_image_dc_batch_and_identities(dev):
    """b > 1 (per-sample sens maps and masks), and two size-independent properties at cfg-2 size: with every row sampled
    and weights (1, 1, 0) the operator is sum_c |S_c|^2 x; it is linear in x."""
    from cine_hip import ops
    from oracle import varnet_ref as V
    b, t, c, h, w = 2, 3, 4, 200, 24
    img, sens = rnd(11, b, t, 1, h, w, 2), rnd(12, b, 1, c, h, w, 2)
    mask = torch.cat([_row_mask(t, h, 5), _row_mask(t, h, 6)])
    ref = V.VarNetBlock.sens_reduce(V.VarNetBlock.sens_expand(img, sens) * mask + 0.0, sens)
    assert rel_err(ops.image_dc(img.to(dev), sens.to(dev), None, mask.to(dev), weights=(1.0, 0.0, 0.0)).cpu(), ref) < OP_TOL
    t, c, h, w = 15, 15, 200, 200
    x, y, sens = torch.randn(1, t, 1, h, w, 2, device=dev), torch.randn(1, t, 1, h, w, 2, device=dev), torch.randn(1, 1, c, h, w, 2, device=dev)
    ones = torch.ones(1, t, 1, h, 1, 1, dtype=torch.uint8, device=dev)
    got = ops.image_dc(x, sens, None, ones, weights=(1.0, 1.0, 0.0))
    want = x * (sens ** 2).sum(dim=(2, 5), keepdim=True)
    assert rel_err(got.cpu(), want.cpu()) < OP_TOL
    mask = _row_mask(t, h, 7).to(dev)
    lam = torch.tensor([0.3], device=dev)
    f = lambda z: ops.image_dc(z, sens, None, mask, lam)
    assert rel_err(f(2 * x - 3 * y).cpu(), (2 * f(x) - 3 * f(y)).cpu()) < OP_TOL


@pytest.mark.parametrize("t,c,w", [(15, 15, 200), (2, 17, 203), (3, 16, 8), (1, 5, 1), (2, 6, 37)])
def test_tiled_sensitivities_give_the_same_bits(dev, t, c, w):
    """cine_sens_tile_pack + the *_t operators (the maps in column-tile-major order, read as contiguous runs) against the plain maps:
    identical outputs for the soft DC, CineNet's normal operator and a whole CG iteration; ragged last column tiles, 1 - 4 coil groups."""
    from cine_hip import ops
    h = 200
    sens = rnd(2, 1, 1, c, h, w, 2).to(dev)
    img, zf = rnd(3, 1, t, 1, h, w, 2).to(dev), rnd(4, 1, t, 1, h, w, 2).to(dev)
    mask = _row_mask(t, h, 5).to(dev)
    lam = torch.tensor([0.3], device=dev)
    tiled = ops.sens_tile_pack(sens)
    assert tiled is not None and ops.sens_tile_pack(rnd(5, 1, 1, 2, 24, 20, 2).to(dev)) is None
    assert torch.equal(ops.image_dc(img, sens, zf, mask, lam, sens_tiled=tiled), ops.image_dc(img, sens, zf, mask, lam))
    assert torch.equal(ops.image_dc(img, sens, zf, mask, lam, magnitude=True, sens_tiled=tiled), ops.image_dc(img, sens, zf, mask, lam, magnitude=True))
    assert torch.equal(ops.normal_op(img, sens, mask, lam, tiled), ops.normal_op(img, sens, mask, lam))
    outs = []
    for st in (tiled, None):
        x, r, p = img.clone(), zf.clone(), zf.clone()
        rr = [ops.dot(r, r), torch.empty(1, device=dev)]
        ops.normal_op_cg_step(x, r, p, sens, mask, lam, rr[0], rr[1], sens_tiled=st)
        outs.append((x, r, p, rr[1]))
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)


def test_acs_window_found_on_the_device_equals_the_host_form(dev):
    """cine_acs_window (the forward pass without ``acs=``: no host read-back of the mask, reference varnet.py:64-68) against SensitivityModel.acs_window
    on the synthetic generator's masks and on hand-made ones (asymmetric centre block, odd height, several batch entries), and the two
    prologues -- host integers / device window -- give the same bits."""
    from reconstruction.models.varnet import SensitivityModel
    from cine_hip import ops, synth
    masks = [synth.make_cine_slice(3, 2, h, 8, accel=a, center_lines=cl, seed=s)["mask"] for h, a, cl, s in [(200, 4, 24, 0), (96, 8, 8, 1), (33, 4, 5, 2)]]
    m = torch.zeros(1, 2, 1, 40, 1, 1); m[:, :, :, 14:29] = 1; m[:, :, :, 3] = 1; m[:, :, :, 37] = 1
    masks.append(m)
    masks.append(torch.cat([masks[1], masks[1]], 0))                 # b = 2: frame 0 of every batch entry, the first centre
    for mk in masks:
        want = SensitivityModel.acs_window(mk)
        win = ops.acs_window_dev(mk.to(dev)).cpu().tolist()
        assert win == [want[0], want[0] + want[1]], (win, want, tuple(mk.shape))
    # no unsampled row on one side of batch 0's centre (fully sampled, or sampled to the edge): the host form raises like the reference's
    # nonzero(...)[-1]; the device form keeps every row -- and never lets a LATER batch entry's rows supply the window's end
    full = torch.ones(1, 2, 1, 40, 1, 1)
    edge = torch.cat([torch.ones(1, 2, 1, 40, 1, 1), m], 0); edge[0, :, :, :10] = 0
    for mk in (full, edge):
        with pytest.raises(IndexError):
            SensitivityModel.acs_window(mk[:1])
        assert ops.acs_window_dev(mk.to(dev)).cpu().tolist() == [0, 40]
    ex = synth.make_cine_slice(4, 3, 48, 40, accel=4, center_lines=6, seed=5)
    k, mk = ex["masked_kspace"].to(dev), ex["mask"].to(dev)
    pad, n_low = SensitivityModel.acs_window(mk)
    assert torch.equal(ops.sens_prologue(k, pad, pad + n_low), ops.sens_prologue(k, ops.acs_window_dev(mk)))


@pytest.mark.parametrize("t,c,w,iters", [(15, 15, 200, 6), (3, 8, 23, 4), (2, 6, 5, 1)])
def test_conj_grad_two_launches_per_iteration_vs_oracle_and_three_launch_form(dev, t, c, w, iters):
    """cine_conj_grad: the whole solve of reference cinenet.py:136-171 (set-up + `iters` iterations, two launches each: the operator forms
    the new direction on load) against the reference's ConjGrad in float64 and against the three-launch iteration of round 4 on the same
    start value and right-hand side."""
    from cine_hip import ops
    from oracle import cinenet_ref as C
    import reconstruction.models as M
    h = 200
    sens = rnd(2, 1, 1, c, h, w, 2)
    sens = sens / sens.pow(2).sum(dim=(2, 5), keepdim=True).sqrt()
    mask = _row_mask(t, h, 5)
    lam = torch.tensor([0.37])
    x0, b0 = rnd(3, 1, t, 1, h, w, 2), rnd(4, 1, t, 1, h, w, 2)
    blk = C.CineNetBlock(torch.nn.Identity(), iters, "XF", True).double()
    with torch.no_grad():
        blk.lambda_reg.copy_(lam.double())
        want = blk.ConjGrad(x0.double(), b0.double(), mask, sens.double(), iters)
    hipb = M.CineNetBlock(torch.nn.Identity(), iters, "XF", True).to(dev)
    with torch.no_grad():
        hipb.lambda_reg.copy_(lam.to(dev))
    xd = x0.to(dev)
    got2 = ops.conj_grad(xd.clone(), b0.to(dev), sens.to(dev), mask.to(dev), hipb.lambda_reg, iters)
    assert got2 is not None and rel_err(got2.cpu(), want.float()) < 2e-5
    try:
        ops.CG_SOLVER = False                                  # the model's own loop: operator + update + direction per iteration
        got3 = hipb.ConjGrad(xd, b0.to(dev), mask.to(dev), sens.to(dev), iters)
    finally:
        ops.CG_SOLVER = True
    assert torch.equal(xd.cpu(), x0)                           # the public method leaves its start value alone
    assert rel_err(got3.cpu(), want.float()) < 2e-5 and rel_err(got2.cpu(), got3.cpu()) < 1e-5
    got = hipb.ConjGrad(xd, b0.to(dev), mask.to(dev), sens.to(dev), iters)         # the default route = cine_conj_grad
    assert torch.equal(got, got2) and torch.equal(xd.cpu(), x0)
    # the DC block's own call: the right-hand side x_ref + softplus(lambda) x0 formed inside the set-up kernel (cinenet.py:106-107)
    v = float(torch.nn.functional.softplus(lam))
    rhs = ops.axpby_dev(b0.to(dev), xd, lambda_reg=hipb.lambda_reg)
    sep = ops.conj_grad(xd.clone(), rhs, sens.to(dev), mask.to(dev), hipb.lambda_reg, iters)
    folded = ops.conj_grad(xd.clone(), b0.to(dev), sens.to(dev), mask.to(dev), hipb.lambda_reg, iters, rhs_is_ref=True)
    assert torch.equal(sep, folded)
    with torch.no_grad():
        want_blk = blk.ConjGrad(x0.double(), b0.double() + v * x0.double(), mask, sens.double(), iters)
    assert rel_err(folded.cpu(), want_blk.float()) < 2e-5


@pytest.mark.parametrize("t,c,w", [(15, 15, 200), (3, 7, 36), (2, 6, 7)])
def test_cg_iteration_three_launch_form_vs_oracle_and_four_launch_form(dev, t, c, w):
    """cine_normal_op_cg_fused (operator with per-workgroup p.Hp partial sums -> update that adds the coil groups itself -> direction)
    against the literal iteration of reference cinenet.py:153-169 in float64 and against the four-launch form (cine_normal_op_pd +
    cine_cg_step_pd): same x, r, p, r.r, p.Hp to rounding; three iterations chained (rr_new feeds the next one)."""
    from cine_hip import ops
    from oracle import cinenet_ref as C
    import torch.nn.functional as F
    h = 200
    sens = rnd(2, 1, 1, c, h, w, 2)
    sens = sens / sens.pow(2).sum(dim=(2, 5), keepdim=True).sqrt()
    mask = _row_mask(t, h, 5)
    lam = torch.tensor([0.37])
    x0, r0 = rnd(3, 1, t, 1, h, w, 2), rnd(4, 1, t, 1, h, w, 2)
    blk = C.CineNetBlock(torch.nn.Identity(), 1, "XF", True).double()
    with torch.no_grad():
        blk.lambda_reg.copy_(lam.double())

    def run(fused):
        ops.FUSED_CG = fused
        x, r, p = x0.to(dev), r0.to(dev), r0.to(dev).clone()
        rr = [ops.dot(r, r), torch.empty(1, device=dev)]
        pd = torch.empty(3, device=dev)
        for k in range(3):
            ops.normal_op_cg_step(x, r, p, sens.to(dev), mask.to(dev), lam.to(dev), rr[k % 2], rr[(k + 1) % 2], pd_out=pd[k:k + 1])
        return x.cpu(), r.cpu(), p.cpu(), rr[1].cpu(), pd.cpu()
    try:
        got3, got4 = run(True), run(False)
    finally:
        ops.FUSED_CG = True
    # float64 reference of the same three iterations
    x, r = x0.double(), r0.double()
    p = r.clone()
    rr = torch.dot(r.flatten(), r.flatten())
    pds = []
    with torch.no_grad():
        for _ in range(3):
            d = blk.HOperator(p, mask, sens.double())
            pdv = torch.dot(p.flatten(), d.flatten()); pds.append(pdv)
            al = rr / pdv
            x = x + al * p; r = r - al * d
            rn = torch.dot(r.flatten(), r.flatten())
            p = r + (rn / rr) * p
            rr = rn
    want = (x, r, p, rr.reshape(1), torch.stack(pds))
    for a3, a4, wv, name in zip(got3, got4, want, ("x", "r", "p", "rr", "pd")):
        assert rel_err(a3, wv.float()) < 2e-5, name
        assert rel_err(a3, a4) < 2e-6, name


# ------------------------------------------------------------------ the steps either side of the path
def _metric_pair(seed):
    rs = np.random.RandomState(seed)
    tgt = rs.uniform(0, 1.5, size=(15, 180, 180)).astype(np.float32)
    rec = np.maximum(tgt + (0.1 * rs.standard_normal((15, 180, 180))).astype(np.float32), 0)
    return tgt, rec


def test_device_metrics_vs_reference_ssimloss_and_host(golden, dev):
    """cine_image_metrics at (15, 180, 180): per-frame-range SSIM against the reference's SSIMLoss vector (fp32 conv2d
    arithmetic there, hence 2e-5), and SSIM / NMSE / PSNR / MSE against the float64 numpy forms of utils/evaluate.py."""
    from cine_hip import ops
    from reconstruction.utils import evaluate
    g = golden("metrics")
    tgt, rec = _metric_pair(int(g["seed"]))
    td, rd = cuda(tgt, dev), cuda(rec, dev)
    m = ops.image_metrics(td, rd, per_frame_range=True)
    assert np.abs((1.0 - m["ssim_frames"].cpu().numpy()) - g["ssim_loss_frames"]).max() < 2e-5
    assert abs(1.0 - float(m["ssim"]) - float(g["ssim_loss"])) < 2e-5
    m = evaluate.metrics_device(td, rd)
    # 1e-8: the host form squares K * data_range with data_range a float32 scalar (the image's max), the kernel in float64
    assert abs(float(m["ssim"]) - evaluate.ssim(tgt, rec)) < 1e-8
    assert abs(float(m["nmse"]) - float(evaluate.nmse(tgt.astype(np.float64), rec.astype(np.float64)))) < 1e-12
    assert abs(float(m["psnr"]) - float(evaluate.psnr(tgt, rec))) < 1e-8
    assert abs(float(m["mse"]) - float(evaluate.mse(tgt.astype(np.float64), rec.astype(np.float64)))) < 1e-12
    assert abs(float(evaluate.ssim_device(td, rd, maxval=2.0)) - evaluate.ssim(tgt, rec, maxval=2.0)) < 1e-8


def test_device_metrics_center_crop_and_edges(dev):
    """Reconstruction 200x200 against a 180x150 target: both are center-cropped to the smaller size first
    (transforms.py:161-183); odd sizes, one frame, a window as large as the frame."""
    from cine_hip import ops
    from reconstruction.utils import evaluate
    rs = np.random.RandomState(5)
    pred = rs.uniform(0, 1, size=(3, 200, 200)).astype(np.float32)
    tgt = rs.uniform(0, 1, size=(3, 180, 150)).astype(np.float32)
    crop = pred[:, 10:190, 25:175]
    m = ops.image_metrics(cuda(tgt, dev), cuda(pred, dev))
    assert abs(float(m["ssim"]) - evaluate.ssim(tgt, crop)) < 1e-8
    assert abs(float(m["nmse"]) - float(evaluate.nmse(tgt.astype(np.float64), crop.astype(np.float64)))) < 1e-12
    a, b = rs.uniform(0, 1, size=(1, 33, 7)).astype(np.float32), rs.uniform(0, 1, size=(1, 33, 7)).astype(np.float32)
    assert abs(float(ops.image_metrics(cuda(a, dev), cuda(b, dev))["ssim"]) - evaluate.ssim(a, b)) < 1e-8
    with pytest.raises(ValueError):
        ops.image_metrics(cuda(a[:, :5], dev), cuda(b[:, :5], dev))
    with pytest.raises(ValueError):
        ops.image_metrics(cuda(a[0], dev), cuda(b[0], dev))


def test_apply_mask_zero_filled_and_fft_norms_vs_reference_golden(golden, dev):
    import reconstruction.utils as U
    from reconstruction.data import transforms as T
    from cine_hip import ops, synth
    g = golden("masks")
    # apply_mask on the device with the reference's mask draw (np.random.seed(2), RandomMaskFunc([4], [4]))
    np.random.seed(2)
    mf = synth.create_mask_for_mask_type("random", [4], [4])
    md, m = T.apply_mask(cuda(g["am_data"], dev), mf, None)
    assert torch.equal(md.cpu(), torch.from_numpy(g["am_masked"])) and torch.equal(m.cpu(), torch.from_numpy(g["am_mask"]))
    k = cuda(g["am_data"], dev)
    out = ops.apply_mask(k, cuda(g["am_mask"], dev).to(torch.uint8), out=k)           # in place
    assert out.data_ptr() == k.data_ptr() and torch.equal(k.cpu(), torch.from_numpy(g["am_masked"]))
    g = golden("metrics")
    k = cuda(g["zf_k"], dev)
    assert rel_err(ops.zero_filled_rss(k).cpu(), g["zf_out"]) < OP_TOL
    assert torch.equal(k.cpu(), torch.from_numpy(g["zf_k"]))
    # the literal lines of run_inference.py:64-67 against this build's utils, GPU and CPU-resident input
    for kk in (k, k.cpu()):
        scaling = torch.sqrt(torch.prod(torch.as_tensor(kk.shape[-3:-1])))
        images = U.ifft2c(kk, norm=None) * scaling.to(kk.device)
        assert images.device == kk.device
        assert rel_err(U.rss_complex(images, dim=2).cpu(), g["zf_out"]) < OP_TOL
    assert rel_err(U.ifft2c(k, norm=None).cpu(), g["ifft2c_none"]) < OP_TOL
    assert rel_err(U.fft2c(k, norm=None).cpu(), g["fft2c_none"]) < OP_TOL
    assert rel_err(U.fft2c(k, norm="forward").cpu(), g["fft2c_none"] / (24 * 20)) < OP_TOL
    with pytest.raises(ValueError):
        U.fft2c(k, norm="bogus")
    # full-size zero-filled reconstruction of the bench slice: every pixel >= the coil-combined magnitude's scale, finite
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=4, seed=0)
    zf = ops.zero_filled_rss(ex["masked_kspace"].to(dev))
    from oracle import centered_fft as cfo, complex_ops as co
    want = co.rss_complex(cfo.ifft2c(ex["masked_kspace"]), dim=2)
    assert zf.shape == (1, 15, 200, 200) and rel_err(zf.cpu(), want) < OP_TOL


# ------------------------------------------------------------------ the cfg-2 U-Net on its own plane shape
@pytest.mark.parametrize("n,sets", [(6, 2), (3, 1), (1, 1)])
def test_unet_cfg2_planes_vs_oracle(dev, n, sets):
    """U-Net(16 ch, 3 pools) on cfg 2's 208 x 16 planes (levels 2 / 3 = 52 x 4 x 64 / 26 x 2 x 128 channels: one workgroup owns
    a plane there), one and two weight sets (x-f / y-f nets in the same launches), against the CPU oracle's Unet."""
    from cine_hip import ops, synth
    from oracle import regularisers as R
    from reconstruction.models.denoisers.unet import Unet
    nets_h, nets_r = [], []
    for k in range(sets):
        hnet = Unet(in_chans=2, out_chans=2, chans=16, num_pool_layers=3).eval(); synth.fill_parameters_(hnet, 31 + k, keep=())
        rnet = R.Unet(in_chans=2, out_chans=2, chans=16, num_pool_layers=3).eval(); rnet.load_state_dict(hnet.state_dict(), strict=True)
        nets_h.append(hnet.to(dev)); nets_r.append(rnet)
    x = rnd(77, n, 2, 208, 16)
    with torch.no_grad():
        per = n // sets
        want = torch.cat([nets_r[k](x[k * per:(k + 1) * per]) for k in range(sets)])
    got = ops.unet2d_forward(x.to(dev), ops.UnetWeights(nets_h)).cpu()
    assert rel_err(got, want) < BLOCK_TOL


@pytest.mark.parametrize("n,sets", [(6, 2), (8, 2), (15, 1), (3, 1), (1, 1)])
def test_unet_branches_bit_identical_to_one_stream(dev, n, sets):
    """cine_unet2d_forward_branches: the planes of a U-Net pass as 2 / 4 concurrent runs on side streams (the x-f / y-f networks of a cascade,
    reference varnet.py:216-232, are independent until their sum; the sens-net's coils until the RSS) against the one-stream launch sequence,
    bit for bit; the counter proves the branched entry point ran."""
    from cine_hip import ops, synth
    from cine_hip._lib import lib
    from reconstruction.models.denoisers.unet import Unet
    nets = []
    for k in range(sets):
        hnet = Unet(in_chans=2, out_chans=2, chans=16, num_pool_layers=3).eval(); synth.fill_parameters_(hnet, 31 + k, keep=())
        nets.append(hnet.to(dev))
    wts = ops.UnetWeights(nets)
    x = rnd(78, n, 2, 208, 16).to(dev)
    with ops.branches(1):
        want = ops.unet2d_forward(x, wts)
    for nb in (2, 4):
        lib().cine_diag_counter(2, 1)
        with ops.branches(nb):
            got = ops.unet2d_forward(x, wts)
        torch.cuda.synchronize()
        with ops.branches(nb):
            eff = ops._branch_count(n, sets)
        assert lib().cine_diag_counter(2, 1) == (1 if eff > 1 else 0), (n, sets, nb, eff)
        assert torch.equal(got, want), (n, sets, nb)


def test_varnet_branches_eager_and_hipgraph_replay_bit_identical(golden, dev):
    """The whole cfg-2 forward with two-branch U-Net passes: eager and replayed from a hipGraph (the side streams join the capture through
    the fork events), equal to the one-stream forward bit for bit."""
    import reconstruction.models as M
    from cine_hip import ops, synth
    g = golden("varnet_cfg2")
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=4, seed=int(g["data_seed"]))
    net = M.VarNet(6, 8, 3, 16, 3, "XF")
    synth.fill_parameters_(net, int(g["weight_seed"]))
    net.to(dev).eval()
    mk, mask = ex["masked_kspace"].to(dev), ex["mask"].to(dev)
    with ops.branches(1):
        want = net(mk, mask).clone()
    for nb in (2, 4):
        with ops.branches(nb):
            assert torch.equal(net(mk, mask), want), nb
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with ops.branches(2):
        with torch.cuda.stream(st):
            net(mk, mask)                            # this stream's caches and side streams, outside capture
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            gout = net(mk, mask)
    for _ in range(2):
        gout.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(gout, want)


def test_side_streams_are_probed_for_a_hardware_queue_of_their_own(dev):
    """ops.side_streams picks streams that really run beside their main stream: chains of cine_spin kernels of known duration on the two streams, forked and joined like the branches, overlap
    (streams_run_concurrently); a stream probed against ITSELF is, of course, serial -- the probe can tell the difference."""
    from cine_hip import ops
    main = torch.cuda.current_stream()
    side = ops.side_streams(dev, 1)[0]
    assert ops.streams_run_concurrently(main, side)
    assert not ops.streams_run_concurrently(main, main)
    # a process that has used many streams: the side stream of a fresh main stream still gets a queue of its own (or a warning says that it could not)
    idle = [torch.cuda.Stream() for _ in range(30)]
    for s_ in idle:
        with torch.cuda.stream(s_):
            torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    m2 = torch.cuda.Stream()
    import warnings
    with torch.cuda.stream(m2), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        s2 = ops.side_streams(dev, 1)[0]
        ok = ops.streams_run_concurrently(m2, s2)
    assert ok or any("hardware queue" in str(x.message) for x in w)
    ops.release_side_streams([m2.cuda_stream])


# ------------------------------------------------------------------ direct tests of the small helpers
@pytest.mark.parametrize("tag", ["odd", "t15", "even", "mixed"])
def test_shift_helpers_vs_reference_golden(golden, dev, tag):
    """utils.fftshift / ifftshift / roll (reference fftc.py:141-213) on the device, bit-exact (index moves only)."""
    import reconstruction.utils as U
    g = golden("ops")
    x = cuda(g[f"{tag}_x"], dev)
    assert torch.equal(U.fftshift(x, dim=[-3, -2]).cpu(), torch.from_numpy(g[f"{tag}_fftshift"]))
    assert torch.equal(U.ifftshift(x, dim=[-3, -2]).cpu(), torch.from_numpy(g[f"{tag}_ifftshift"]))
    assert torch.equal(U.ifftshift(U.fftshift(x, dim=[-3, -2]), dim=[-3, -2]), x)
    sh = [x.shape[-3] // 2, x.shape[-2] // 2]
    assert torch.equal(U.roll(x, sh, [-3, -2]).cpu(), torch.from_numpy(g[f"{tag}_fftshift"]))


@pytest.mark.parametrize("tag", ["p1", "p2", "p3"])
def test_mwcnn_padding_vs_reference_golden(golden, dev, tag):
    """utils.pad_for_mwcnn / unpad_from_mwcnn (reference padding.py:26-66): odd sizes put the extra element on the left."""
    import reconstruction.utils as U
    g = golden("xpdnet")
    x = cuda(g[f"{tag}_x"], dev)
    y, pads = U.pad_for_mwcnn(x, 3)
    assert [int(p) for p in pads] == [int(p) for p in g[f"{tag}_pads"]]
    assert torch.equal(y.cpu(), torch.from_numpy(g[f"{tag}_y"]))
    assert torch.equal(U.unpad_from_mwcnn(y, pads).cpu(), torch.from_numpy(g[f"{tag}_back"]))


def test_dwt_iwt_alone_vs_reference_golden(golden, dev):
    """The Haar DWT / IWT staging of the conv kernel by themselves (reference mwcnn.py:216-263): cine_conv3x3_ex with source
    modes 3 / 4 and a delta kernel (centre tap = identity), against the reference's DWT()(x) and IWT()(DWT()(x))."""
    from cine_hip import ops
    from cine_hip._lib import check, lib
    g = golden("xpdnet")
    x = cuda(g["dwt_x"], dev)                                   # (2, 3, 8, 6)
    n, c, h, w = x.shape

    def delta_conv(src, mode, cin, hs, ws, ho, wo):
        wt = torch.zeros(cin, cin, 3, 3, device=dev)
        wt[torch.arange(cin), torch.arange(cin), 1, 1] = 1.0
        y = torch.empty((n, cin, ho, wo), device=dev)
        check(lib().cine_conv3x3_ex(src.data_ptr(), None, 0, src.shape[1], mode, hs, ws, None, None, 0, 0, 0, 0, 0, 0,
                                    ops.pack_conv3x3(wt).data_ptr(), None, None, 0, y.data_ptr(), None, n, cin, ho, wo,
                                    ops.IN_EPS, ops.lrelu_slope(), ops._stream()), "cine_conv3x3_ex")
        return y
    dwt = delta_conv(x, 3, 4 * c, h, w, h // 2, w // 2)
    assert rel_err(dwt.cpu(), g["dwt_y"]) < 1e-6
    iwt = delta_conv(cuda(g["dwt_y"], dev), 4, c, h // 2, w // 2, h, w)
    assert rel_err(iwt.cpu(), g["iwt_y"]) < 1e-6
    assert rel_err(iwt.cpu(), g["dwt_x"]) < 1e-6               # the Haar pair is orthogonal: IWT(DWT(x)) = x


def test_varnet_batch_of_two_equals_two_single_slices(golden, dev):
    """The reference's batch axis (b = 2) on the drop-in VarNet: every slice of the batch equals its own b = 1 run against the
    CPU oracle (the reference's scripts use b = 1; the C ABI underneath is batch-capable)."""
    import reconstruction.models as M
    from cine_hip import synth
    from oracle import varnet_ref as V
    exs = [synth.make_cine_slice(5, 3, 24, 20, accel=4, center_lines=4, seed=s) for s in (21, 22)]
    hip = M.VarNet(2, 4, 2, 4, 2, "XF")
    synth.fill_parameters_(hip, 23)
    ref = V.VarNet(2, 4, 2, 4, 2, "XF")
    ref.load_state_dict(hip.state_dict())
    hip = hip.to(dev)
    mk = torch.cat([e["masked_kspace"] for e in exs]).to(dev)
    mask = torch.cat([exs[0]["mask"], exs[0]["mask"]]).to(dev)          # one ACS window for the batch (the reference reads sample 0's mask, varnet.py:64-68)
    got = hip(mk, mask).cpu()
    for i, e in enumerate(exs):
        want = ref(e["masked_kspace"], exs[0]["mask"])
        assert rel_err(got[i:i + 1], want) < MODEL_TOL, i


# ------------------------------------------------------------------ Lightning-style checkpoints + the inference lines
@pytest.mark.parametrize("kind", ["varnet", "cinenet", "xpdnet"])
def test_lightning_checkpoint_and_inference_lines_vs_reference_golden(golden, dev, kind):
    """A reference-made Lightning checkpoint state dict (keys prefixed `varnet.` / `cinenet.` / `xpdnet.`, plus the loss window
    and the metric accumulators of pl_modules/mri_module.py:55-62) loads with strict=True into a module that holds the DROP-IN
    model under the reference's attribute names, and the literal lines of traintest_scripts/run_inference.py:53-78 on it
    reproduce the reference's stored target / output / zero-filled arrays (lightning_ckpt.npz)."""
    import reconstruction as rec
    import reconstruction.models as M
    from reconstruction.data.transforms import center_crop_to_smallest
    from reconstruction.utils import SSIMLoss
    g = golden("lightning_ckpt")

    class MetricSum(torch.nn.Module):                       # pl_modules/mri_module.py:22-35: one accumulator buffer
        def __init__(self):
            super().__init__()
            self.register_buffer("quantity", torch.tensor(0.0))

    class Module(torch.nn.Module):                          # the attribute tree of the reference's *Module classes
        def __init__(self):
            super().__init__()
            for name in ("NMSE", "SSIM", "PSNR", "ValLoss", "TrainLoss", "TestLoss", "TotExamples", "TotSliceExamples"):
                setattr(self, name, MetricSum())
            if kind == "varnet":
                self.varnet = M.VarNet(num_cascades=2, sens_chans=4, sens_pools=2, chans=4, pools=2, dynamic_type="XF", weight_sharing=False)
            elif kind == "cinenet":
                self.cinenet = M.CineNet(num_cascades=2, CG_iters=3, chans=4, pools=2, dynamic_type="XF", weight_sharing=False)
            else:
                self.xpdnet = M.XPDNet(num_cascades=2, sens_chans=4, sens_pools=2, n_scales=2, n_filters_per_scale=[8, 16],
                                       n_convs_per_scale=[1, 1], first_conv_n_filters=8, n_primal=2, dynamic_type="XF")
            self.loss = SSIMLoss()

        def forward(self, *a):                              # pl_modules/*_module.py forward
            return getattr(self, kind)(*a)

    model = Module()
    sd = state_dict_from(g, f"{kind}::ckpt::")
    assert any(k.startswith(kind + ".") for k in sd) and "loss.w" in sd and "NMSE.quantity" in sd
    model.load_state_dict(sd, strict=True)
    # ---- run_inference.py:41, 53-78
    device = "cuda"
    model = model.to(device).eval()
    masked_kspace, mask, target, sens_maps = (torch.from_numpy(g[k]) for k in ("masked_kspace", "mask", "target", "sens_maps"))
    masked_k = masked_kspace.to(device)
    mask = mask.to(device)
    if kind == "cinenet":
        output = model(masked_k, mask, sens_maps.to(device))
    else:
        output = model(masked_k, mask)
    output = output.cpu()
    scaling_factor = torch.sqrt(torch.prod(torch.as_tensor(masked_kspace.shape[-3:-1])))
    images = rec.utils.ifft2c(masked_kspace, norm=None) * scaling_factor
    zero_filled = rec.utils.rss_complex(images, dim=2)
    target, output = center_crop_to_smallest(target, output)
    target, zero_filled = center_crop_to_smallest(target, zero_filled)
    assert torch.equal(target[0], torch.from_numpy(g[f"{kind}_target"]))
    assert rel_err(output[0], g[f"{kind}_output"]) < MODEL_TOL
    assert rel_err(zero_filled[0].cpu(), g[f"{kind}_zero_filled"]) < 1e-5


def test_bench_force_dist_initialises_rccl_on_hardware(dev):
    """bench.py --gpus 1 --force-dist under torch.distributed.run (one rank): init_process_group("nccl") = RCCL, the device
    barrier and the all_gather_into_tensor of the volume assembly run on the GPU at world size 1 (SURVEY 8 e readiness; the
    8-GPU scaling runs are the driver's).  A child process, started as such -- never an exec of this GPU process."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("NCCL_DEBUG", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "1", "--force-dist",
           "--headline-only", "--no-cpu-baseline", "--repeats", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    line = json.loads(lines[-1])
    assert line["rccl_initialised"] is True and line["rccl_ranks"] == 1 and line["n_gpus"] == 1
    assert line["value"] > 50 and line["steps"] == 6


@pytest.mark.gpu
def test_training_gradient_all_reduce_runs_on_rccl(dev):
    """Data-parallel training readiness on hardware: tools/train_bench.py under torch.distributed.run (one rank, the collective forced):
    init_process_group("nccl") = RCCL, cine_hip.shard.GradientAllReduce's flat all-reduce of all gradients every step, and the step does
    not get slower than the side-stream overlap allows (the package raises GPU_MAX_HW_QUEUES: with four hardware queues RCCL's streams
    push the weight-gradient stream onto the main stream's queue).  A child process -- never an exec of this GPU process."""
    import re
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("NCCL_DEBUG", None)
    env.pop("GPU_MAX_HW_QUEUES", None)
    env["CINE_FORCE_COLLECTIVE"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29519", os.path.join(ROOT, "tools", "train_bench.py"), "4", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    m = re.search(r"training step: ([0-9.]+) ms\s+loss ([0-9.]+).*\[1 rank\(s\), RCCL gradient all-reduce of ([0-9.]+) MiB", r.stdout)
    assert m, r.stdout[-2000:] + "\n" + r.stderr[-2000:]
    assert 0.0 < float(m.group(2)) < 1.0 and 3.5 < float(m.group(3)) < 5.0
    # the same step in the same situation (a child of this test process, right now) without a process group: the bar is relative --
    # an absolute one depends on what the box and the parent process are doing (60.7 ms measured once at the end of a whole -m gpu run,
    # 36.0 ms alone); four hardware queues instead of sixteen cost 13 % (40.4 -> 45.7 ms)
    env0 = dict(env)
    env0.pop("CINE_FORCE_COLLECTIVE")
    r0 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_bench.py"), "4", "2"], env=env0, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r0.returncode == 0, r0.stdout[-2000:] + "\n" + r0.stderr[-4000:]
    m0 = re.search(r"training step: ([0-9.]+) ms", r0.stdout)
    assert m0, r0.stdout[-2000:]
    # a four-step region is 0.13 s: the first GPU process after an idle spell (RCCL's start-up is one) runs 30 - 70 % slower than the next one on the same
    # box (45.4, then 32.2 - 34.4 ms in six back-to-back runs of the plain step).  What this bar is for is a side lane that SERIALISES behind the collective's
    # streams (a step of ~2x), so a slow first sample is taken again before it counts
    t_rccl, t_plain = float(m.group(1)), float(m0.group(1))
    for _ in range(2):
        if t_rccl < 1.25 * t_plain + 3.0:
            break
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
        m = re.search(r"training step: ([0-9.]+) ms", r.stdout)
        assert m, r.stdout[-2000:]
        t_rccl = min(t_rccl, float(m.group(1)))
    assert t_rccl < 1.25 * t_plain + 3.0, (t_rccl, t_plain)
