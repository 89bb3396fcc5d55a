"""Training through the HIP path (second half): gradients of the drop-in models against

  * the REFERENCE's own gradients of its training step (tests/golden/varnet_grad*.npz, written by make_golden.py from
    pl_modules/varnet_module.py:97-113 + loss.backward() + one Adam step of :151-154), and
  * the CPU oracle's autograd on seeded inputs, piece by piece (U-Net, NormUnet, x-f / y-f regulariser, image-space DC,
    coil reduce, rss normalisation, complex_abs).

Tolerance: 1e-4 of each gradient tensor's largest magnitude (float32 arithmetic on both sides; the fixtures store how far
the reference's float32 gradients are from its own float64 run -- 2e-6 .. 8e-6 -- and the bar is max(1e-4, 20x that floor)).
The tiny fixtures are chosen kink-stable (make_golden.py:_kink_stability): most weight seeds put a whole InstanceNorm plane
within rounding of LeakyReLU's kink, where the reference's own float32 gradient jumps by 1e-3 under a 1e-6 input change and
no implementation can be pinned.
"""
import numpy as np
import pytest
import torch

from conftest import rel_err, rnd, state_dict_from
from cine_hip import ops as cine_ops

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    return torch.device("cuda:0")


def _grads(module, loss):
    module.zero_grad(set_to_none=True)
    loss.backward()
    return {k: p.grad.detach().clone() for k, p in module.named_parameters() if p.grad is not None}


def _cmp(got, want, what, tol=TOL):
    bad = {}
    for k, w in want.items():
        assert k in got and got[k] is not None, f"{what}: no gradient for {k}"
        e = rel_err(got[k].cpu(), w)
        if e > tol:
            bad[k] = e
    assert not bad, f"{what}: {bad}"


# ------------------------------------------------------------------ U-Net / NormUnet against the oracle's autograd
@pytest.mark.parametrize("n,chans,pools,h,w", [(6, 4, 2, 32, 16), (4, 8, 3, 48, 16), (3, 4, 2, 26, 22), (2, 18, 4, 32, 32), (5, 4, 1, 10, 6)])
def test_unet_backward_vs_oracle(dev, n, chans, pools, h, w):
    """cine_unet2d_backward (reference denoisers/unet.py:73-125 under autograd): every weight gradient and the input
    gradient; odd sizes after pooling exercise the zero-pad crop of :106-120, chans 18 the ragged 16-channel chunks."""
    from reconstruction.models.denoisers.unet import Unet
    from oracle import regularisers as R
    from cine_hip import synth
    hip = Unet(chans, pools, 2, 2).to(dev)
    synth.fill_parameters_(hip, 11)
    ref = R.Unet(chans, pools, 2, 2)
    ref.load_state_dict(hip.state_dict())
    x, gy = rnd(1, n, 2, h, w), rnd(2, n, 2, h, w)
    with torch.enable_grad():
        xr = x.clone().requires_grad_(True)
        want = _grads(ref, (ref(xr) * gy).sum())
        xh = x.to(dev).requires_grad_(True)
        yh = hip(xh)
        got = _grads(hip, (yh * gy.to(dev)).sum())
    _cmp(got, want, "unet weight gradients")
    assert rel_err(xh.grad.cpu(), xr.grad) < TOL


def test_unet_two_sets_and_determinism(dev):
    """Two weight sets in one launch sequence (the x-f / y-f U-Nets, varnet.py:224-226) == two single launches; a repeated
    backward pass is bit-identical (fixed-order reductions)."""
    from reconstruction.models.denoisers.unet import Unet
    from cine_hip import ops, synth, autograd as ag
    a, b = Unet(4, 2, 2, 2).to(dev), Unet(4, 2, 2, 2).to(dev)
    synth.fill_parameters_(a, 21); synth.fill_parameters_(b, 22)
    x, gy = rnd(3, 8, 2, 32, 16).to(dev), rnd(4, 8, 2, 32, 16).to(dev)
    both = ops.UnetWeights([a, b])
    with torch.enable_grad():
        y = ag.unet2d(x.clone().requires_grad_(True), both)
        g2 = _grads(torch.nn.ModuleList([a, b]), (y * gy).sum())
        y_again = ag.unet2d(x.clone().requires_grad_(True), both)
        g2b = _grads(torch.nn.ModuleList([a, b]), (y_again * gy).sum())
        ya = a(x[:4].clone().requires_grad_(True)); ga = _grads(a, (ya * gy[:4]).sum())
        yb = b(x[4:].clone().requires_grad_(True)); gb = _grads(b, (yb * gy[4:]).sum())
    for k in g2:
        assert torch.equal(g2[k], g2b[k]), k
    for k, v in ga.items():
        assert rel_err(g2["0." + k].cpu(), v.cpu()) < 1e-5, k
    for k, v in gb.items():
        assert rel_err(g2["1." + k].cpu(), v.cpu()) < 1e-5, k


@pytest.mark.parametrize("h,w", [(24, 20), (33, 17)])
def test_normunet_backward_vs_oracle(dev, h, w):
    """NormUnet.forward under autograd (norm_unet.py:98-114): group norm with the unbiased std, pad, U-Net, unpad, un-norm."""
    from reconstruction.models.denoisers.norm_unet import NormUnet
    from oracle import regularisers as R
    from cine_hip import synth
    hip = NormUnet(4, 2).to(dev)
    synth.fill_parameters_(hip, 31)
    ref = R.NormUnet(4, 2)
    ref.load_state_dict(hip.state_dict())
    x, gy = rnd(5, 5, 1, h, w, 2) * 3 + 0.5, rnd(6, 5, 1, h, w, 2)
    with torch.enable_grad():
        xr = x.clone().requires_grad_(True)
        want = _grads(ref, (ref(xr) * gy).sum())
        xh = x.to(dev).requires_grad_(True)
        got = _grads(hip, (hip(xh) * gy.to(dev)).sum())
    _cmp(got, want, "normunet weight gradients")
    assert rel_err(xh.grad.cpu(), xr.grad) < TOL


# ------------------------------------------------------------------ the pieces around the regulariser
@pytest.mark.parametrize("dyn,share", [("XF", False), ("XT", False), ("XF", True)])
def test_xfyf_backward_vs_oracle(dev, dyn, share):
    """VarNetBlock.xfyf_transform under autograd (varnet.py:196-241), h != w so the two plane sets differ in shape for the
    unshared case too."""
    import reconstruction.models as M
    from oracle import varnet_ref as V
    from cine_hip import synth
    hip = M.VarNet(1, 4, 2, 4, 2, dyn, share).to(dev)
    synth.fill_parameters_(hip, 41)
    ref = V.VarNet(1, 4, 2, 4, 2, dyn, share)
    ref.load_state_dict(hip.state_dict())
    for hh, ww in ((24, 20), (16, 16)):
        img, g = rnd(7, 1, 5, hh, ww, 2), rnd(8, 1, 5, 1, hh, ww, 2)
        with torch.enable_grad():
            ir = img.clone().requires_grad_(True)
            want = _grads(ref, (ref.cascades[0].xfyf_transform(ir) * g).sum())
            ih = img.to(dev).requires_grad_(True)
            got = _grads(hip, (hip.cascades[0].xfyf_transform(ih) * g.to(dev)).sum())
        _cmp(got, want, f"xfyf {dyn} {hh}x{ww}")
        assert rel_err(ih.grad.cpu(), ir.grad) < TOL


@pytest.mark.parametrize("t,c,h,w", [(3, 4, 24, 20), (2, 3, 200, 12), (4, 7, 17, 9), (2, 2, 96, 10), (1, 2, 512, 9)])
def test_image_dc_backward_vs_oracle(dev, t, c, h, w):
    """cine_image_dc under autograd against the literal k-space formula of varnet.py:181-194, 281-282 differentiated by torch:
    gradients for the image, the maps, the zero-filled term and lambda_reg (h = 200 takes the 10 x 20 FFT engine, 24 / 96 / 512 the
    mixed-radix one, 17 the direct DFT)."""
    from cine_hip import autograd as ag
    from oracle import centered_fft as cf, complex_ops as co
    m, sens, kref = rnd(9, 1, t, 1, h, w, 2), rnd(10, 1, 1, c, h, w, 2), rnd(11, 1, t, c, h, w, 2)
    mask = (torch.from_numpy(np.random.RandomState(12).rand(1, t, 1, h, 1, 1)) < 0.4).to(torch.uint8)
    g = rnd(13, 1, t, 1, h, w, 2)
    lam = torch.tensor([0.37])
    with torch.enable_grad():
        mr, sr, lr = m.clone().requires_grad_(True), sens.clone().requires_grad_(True), lam.clone().requires_grad_(True)
        v = torch.nn.functional.softplus(lr)
        kth = cf.fft2c(co.complex_mul(mr, sr))
        mk = mask.to(kth.dtype)
        k = (1 - mk) * kth + mk * (kth + v * (kref * mk)) / (1 + v)
        out_r = co.complex_mul(cf.ifft2c(k), co.complex_conj(sr)).sum(dim=2, keepdim=True)
        (out_r * g).sum().backward()
        md, sd, ld = m.to(dev).requires_grad_(True), sens.to(dev).requires_grad_(True), lam.to(dev).requires_grad_(True)
        zf = ag.CoilReduceFn.apply(kref.to(dev), sd, mask.to(dev))
        out = ag.ImageDcFn.apply(md, sd, zf, mask.to(dev), ld)
        (out * g.to(dev)).sum().backward()
    assert rel_err(out.detach().cpu(), out_r.detach()) < 1e-5
    assert rel_err(md.grad.cpu(), mr.grad) < TOL
    assert rel_err(sd.grad.cpu(), sr.grad) < TOL
    assert rel_err(ld.grad.cpu(), lr.grad) < TOL


def test_rss_abs_backward_vs_torch(dev):
    """divide_root_sum_of_squares (varnet.py:58-59) and complex_abs (math.py:48-62) under autograd."""
    from cine_hip import autograd as ag
    from oracle import complex_ops as co
    x, g = rnd(14, 2, 5, 12, 10, 2), rnd(15, 2, 5, 12, 10, 2)
    with torch.enable_grad():
        xr = x.clone().requires_grad_(True)
        ((xr / co.rss_complex(xr, dim=1).unsqueeze(-1).unsqueeze(1)) * g).sum().backward()
        xd = x.to(dev).requires_grad_(True)
        (ag.RssNormFn.apply(xd) * g.to(dev)).sum().backward()
        ar = x.clone().requires_grad_(True)
        (co.complex_abs(ar) * g[..., 0]).sum().backward()
        ad = x.to(dev).requires_grad_(True)
        (ag.AbsFn.apply(ad) * g[..., 0].to(dev)).sum().backward()
    assert rel_err(xd.grad.cpu(), xr.grad) < 1e-5
    assert rel_err(ad.grad.cpu(), ar.grad) < 1e-5


# ------------------------------------------------------------------ whole model against the reference's gradients
def _training_step(model, mk, mask, target, lr=0.0003, extra=()):
    """Body of reference pl_modules/varnet_module.py:97-113 on the drop-in modules, loss.backward(), one Adam step (:151-154)."""
    from reconstruction.data import transforms
    from reconstruction.utils import SSIMLoss
    lossf = SSIMLoss().to(mk.device)
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=0.0)
    opt.zero_grad()
    output = model(mk, mask, *extra)
    tgt, out = transforms.center_crop_to_smallest(target, output)
    loss = lossf(out.unsqueeze(1), tgt.unsqueeze(1), data_range=tgt.max())
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    opt.step()
    return loss.detach(), grads, output.detach()


@pytest.mark.parametrize("tag,dyn,share", [("XF", "XF", False), ("XT", "XT", False), ("2D", "2D", False), ("XFws", "XF", True), ("3D", "3D", False)])
def test_varnet_training_step_vs_reference_golden(dev, golden, tag, dyn, share):
    """The reference's training step on the drop-in VarNet: loss, every parameter gradient and the weights after one Adam
    step against the reference's own (varnet_grad.npz)."""
    import reconstruction.models as M
    g = golden("varnet_grad")
    net = M.VarNet(2, 4, 2, 4, 2, dyn, share)
    net.load_state_dict(state_dict_from(g, f"{tag}::sd::"), strict=True)
    net = net.to(dev).train()
    mk, mask, target = (torch.from_numpy(g[k]).to(dev) for k in ("masked_kspace", "mask", "target"))
    with torch.enable_grad():
        loss, grads, out = _training_step(net, mk, mask, target)
    assert rel_err(out.cpu(), g[f"{tag}_out"]) < TOL
    assert abs(float(loss) - float(g[f"{tag}_loss"])) < 1e-5
    new = dict(net.named_parameters())
    bad = {}
    for k in (k[len(tag) + 8:] for k in g if k.startswith(f"{tag}::grad::")):
        floor = float(g[f"{tag}::floor::{k}"])
        tol = TOL if floor < 5e-6 else max(TOL, 20 * floor)
        e = rel_err(grads[k].cpu(), g[f"{tag}::grad::{k}"])
        if e > tol:
            bad[k] = (e, floor)
        # Adam's first step moves every weight by lr * g / (|g| + eps): compare where the gradient is well above eps
        want_new, gref = torch.from_numpy(g[f"{tag}::new::{k}"]), torch.from_numpy(g[f"{tag}::grad::{k}"])
        sel = gref.abs() > 1e-5
        assert (new[k].detach().cpu() - want_new)[sel].abs().max() <= 0.02 * 0.0003, k
    assert not bad, bad


@pytest.mark.parametrize("tag,dyn,share", [("XF", "XF", False), ("XT", "XT", False), ("2D", "2D", False), ("XFws", "XF", True), ("3D", "3D", False)])
def test_cinenet_training_step_vs_reference_golden(dev, golden, tag, dyn, share):
    """The training step of pl_modules/cinenet_module.py:98-114 on the drop-in CineNet: bare U-Nets through the HIP backward kernels,
    conjugate gradients through the adjoint recurrence (the reference detaches alpha / beta, cinenet.py:159-169), lambda_reg through
    both; loss, gradients and the Adam-updated weights against the reference's own (cinenet_grad.npz)."""
    import reconstruction.models as M
    from reconstruction.data import transforms
    from reconstruction.utils import SSIMLoss
    g = golden("cinenet_grad")
    net = M.CineNet(2, 3, 4, 2, dyn, share)
    net.load_state_dict(state_dict_from(g, f"{tag}::sd::"), strict=True)
    net = net.to(dev).train()
    mk, mask, target, sens = (torch.from_numpy(g[k]).to(dev) for k in ("masked_kspace", "mask", "target", "sens_maps"))
    with torch.enable_grad():
        opt = torch.optim.Adam(net.parameters(), lr=0.0003, weight_decay=0.0)
        opt.zero_grad()
        output = net(mk, mask, sens)
        tgt, out = transforms.center_crop_to_smallest(target, output)
        loss = SSIMLoss().to(dev)(out.unsqueeze(1), tgt.unsqueeze(1), data_range=tgt.max())
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
        opt.step()
    assert rel_err(output.detach().cpu(), g[f"{tag}_out"]) < TOL
    assert abs(float(loss) - float(g[f"{tag}_loss"])) < 1e-5
    new = dict(net.named_parameters())
    bad = {}
    for k in (k[len(tag) + 8:] for k in g if k.startswith(f"{tag}::grad::")):
        floor = float(g[f"{tag}::floor::{k}"])
        e = rel_err(grads[k].cpu(), g[f"{tag}::grad::{k}"])
        if e > max(TOL, 20 * floor):
            bad[k] = (e, floor)
        want_new, gref = torch.from_numpy(g[f"{tag}::new::{k}"]), torch.from_numpy(g[f"{tag}::grad::{k}"])
        sel = gref.abs() > 1e-5
        if sel.any():
            assert (new[k].detach().cpu() - want_new)[sel].abs().max() <= 0.02 * 0.0003, k
    assert not bad, bad


_XPD_GRAD_KW = dict(num_cascades=2, sens_chans=4, sens_pools=2, n_scales=2, n_filters_per_scale=[8, 16], n_convs_per_scale=[2, 1],
                    first_conv_n_filters=8, n_primal=2)


@pytest.mark.parametrize("tag,dyn,share", [("XF", "XF", False), ("XT", "XT", False), ("XFws", "XF", True), ("2D", "2D", False), ("XFdual", "XF", False)])
def test_xpdnet_training_step_vs_reference_golden(dev, golden, tag, dyn, share):
    """The training step of pl_modules/xpdnet_module.py on the drop-in primal-only XPDNet: sensitivity network (residual U-Net, RSS
    normalisation), K step + masked backward operator with respect to image and maps, the I-step network (buffer pack with XPDNet's own
    temporal transform, both MWCNNs through the HIP backward kernels, unpack); loss, gradients and the Adam-updated weights against the
    reference's own (xpdnet_grad.npz)."""
    import reconstruction.models as M
    g = golden("xpdnet_grad")
    net = M.XPDNet(dynamic_type=dyn, weight_sharing=share, primal_only=tag != "XFdual", **_XPD_GRAD_KW)
    net.load_state_dict(state_dict_from(g, f"{tag}::sd::"), strict=True)
    net = net.to(dev).train()
    mk, mask, target = (torch.from_numpy(g[k]).to(dev) for k in ("masked_kspace", "mask", "target"))
    with torch.enable_grad():
        loss, grads, out = _training_step(net, mk, mask, target)
    assert rel_err(out.cpu(), g[f"{tag}_out"]) < TOL
    assert abs(float(loss) - float(g[f"{tag}_loss"])) < 1e-5
    new = dict(net.named_parameters())
    bad = {}
    for k in (k[len(tag) + 8:] for k in g if k.startswith(f"{tag}::grad::")):
        floor = float(g[f"{tag}::floor::{k}"])
        e = rel_err(grads[k].cpu(), g[f"{tag}::grad::{k}"])
        if e > max(TOL, 20 * floor):
            bad[k] = (e, floor)
        want_new, gref = torch.from_numpy(g[f"{tag}::new::{k}"]), torch.from_numpy(g[f"{tag}::grad::{k}"])
        sel = gref.abs() > 1e-5
        if sel.any():
            assert (new[k].detach().cpu() - want_new)[sel].abs().max() <= 0.02 * 0.0003, k
    assert not bad, bad


@pytest.mark.parametrize("tag", ["varnet_rnn", "cinenet_rnn", "xpdnet_rnn", "xpdnet_rnn_dual"])
def test_rnn_training_step_vs_reference_golden(dev, golden, tag):
    """The reference's training step on the drop-in convolutional-RNN hybrids: back-propagation through the BCRNN time sweeps (both
    directions) and through the hidden states carried across cascades, on the HIP conv / weight-gradient kernels; loss, gradients and
    Adam-updated weights against the reference's own (rnn_grad.npz)."""
    import reconstruction.models as M
    g = golden("rnn_grad")
    net = {"varnet_rnn": lambda: M.VarNet_RNN(3, 4, 2, 6), "cinenet_rnn": lambda: M.CineNet_RNN(3, 3, 6),
           "xpdnet_rnn": lambda: M.XPDNet_RNN(3, 4, 2, 6, True, 2, 1), "xpdnet_rnn_dual": lambda: M.XPDNet_RNN(3, 4, 2, 6, False, 2, 1)}[tag]()
    net.load_state_dict(state_dict_from(g, f"{tag}::sd::"), strict=True)
    net = net.to(dev).train()
    mk, mask, target, sens = (torch.from_numpy(g[k]).to(dev) for k in ("masked_kspace", "mask", "target", "sens_maps"))
    with torch.enable_grad():
        loss, grads, out = _training_step(net, mk, mask, target, extra=(sens,) if tag == "cinenet_rnn" else ())
    assert rel_err(out.cpu(), g[f"{tag}_out"]) < TOL
    assert abs(float(loss) - float(g[f"{tag}_loss"])) < 1e-5
    new = dict(net.named_parameters())
    bad = {}
    for k in (k[len(tag) + 8:] for k in g if k.startswith(f"{tag}::grad::")):
        floor = float(g[f"{tag}::floor::{k}"])
        e = rel_err(grads[k].cpu(), g[f"{tag}::grad::{k}"])
        if e > max(TOL, 20 * floor):
            bad[k] = (e, floor)
        want_new, gref = torch.from_numpy(g[f"{tag}::new::{k}"]), torch.from_numpy(g[f"{tag}::grad::{k}"])
        sel = gref.abs() > 1e-5
        if sel.any():
            assert (new[k].detach().cpu() - want_new)[sel].abs().max() <= 0.02 * 0.0003, k
    assert not bad, bad


def test_in_lrelu_bwd_large_planes_vs_torch_autograd(dev):
    """cine_in_lrelu_bwd on few, large planes (the 3-D U-Net's volumes): planes above 32 768 elements are cut into 8 192-element chunks
    (chunk sums, fixed-order re-add, apply) -- against float64 autograd of LeakyReLU(InstanceNorm(x)); ragged last chunk, odd width."""
    from cine_hip import ops
    from cine_hip._lib import lib, check
    import torch.nn.functional as F
    torch.manual_seed(2)
    n, c, h, w = 2, 3, 211, 199                       # 41 989 elements per plane: six chunks, the last one ragged
    x = torch.randn(n, c, h, w) * 1.7 + 0.3
    g = torch.randn(n, c, h, w)
    x64 = x.double().requires_grad_(True)
    with torch.enable_grad():
        F.leaky_relu(F.instance_norm(x64, eps=1e-5), 0.2).backward(g.double())
    xd, gd = x.to(dev), g.to(dev)
    part = ops.instnorm_partials(xd)
    out = torch.empty_like(xd)
    L = lib()
    nb = L.cine_in_lrelu_bwd_ws_bytes(n, c, h, w)
    assert nb > 0
    ws = torch.empty(nb, device=dev, dtype=torch.uint8)
    st = torch.cuda.current_stream().cuda_stream
    check(L.cine_in_lrelu_bwd(xd.data_ptr(), part.data_ptr(), 1, gd.data_ptr(), out.data_ptr(), n, c, h, w, 1e-5, 0.2, ws.data_ptr(), nb, st), "cine_in_lrelu_bwd")
    assert rel_err(out.cpu(), x64.grad) < 2e-5
    out2 = torch.empty_like(xd)                         # without a workspace: one workgroup per plane, same result up to summation order
    check(L.cine_in_lrelu_bwd(xd.data_ptr(), part.data_ptr(), 1, gd.data_ptr(), out2.data_ptr(), n, c, h, w, 1e-5, 0.2, None, 0, st), "cine_in_lrelu_bwd")
    assert rel_err(out2.cpu(), x64.grad) < 2e-5


@pytest.mark.parametrize("shape,chans,pools", [((1, 2, 5, 12, 10), 4, 2), ((2, 2, 7, 9, 11), 3, 1), ((1, 2, 6, 88, 80), 2, 1),
                                               ((1, 2, 6, 24, 24), 16, 2)])
def test_unet3d_backward_vs_oracle_autograd(dev, shape, chans, pools):
    """Unet3dFn = cine_unet3d_forward_train + cine_unet3d_backward (3x3x3 weight gradients as depth-offset passes of the 2-D weight-gradient kernel over
    depth-major slices, transpose conv through the space-to-depth copy, odd extents with the up-path zero pad -- depth-only crops as shorter planes,
    in-plane crops as volume windows --, the 2x2x2 pooling adjoint gathered on load; two samples; 16-channel levels on the coarse conv kernel)
    against the oracle's float64 autograd."""
    import reconstruction.models as M
    from oracle import regularisers as R
    from cine_hip import synth
    from reconstruction.models.denoisers.unet import Unet
    net = Unet(chans, pools, in_chans=shape[1], out_chans=2, dims=3)
    synth.fill_parameters_(net, 5, keep=())
    ref = R.Unet(chans, pools, in_chans=shape[1], out_chans=2, dims=3).double()
    ref.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    torch.manual_seed(1)
    x = torch.randn(*shape); gy = torch.randn(shape[0], 2, *shape[2:])
    x64 = x.double().requires_grad_(True)
    with torch.enable_grad():
        y64 = ref(x64); y64.backward(gy.double())
    net = net.to(dev).train()
    xg = x.to(dev).requires_grad_(True)
    with torch.enable_grad():
        y = net(xg); y.backward(gy.to(dev))
    assert rel_err(y.detach().cpu(), y64.detach()) < 2e-5
    assert rel_err(xg.grad.cpu(), x64.grad) < 5e-5
    want = dict(ref.named_parameters())
    for k, p in net.named_parameters():
        assert rel_err(p.grad.cpu(), want[k].grad) < 5e-5, k


def test_conv_sum_and_bcrnn_backward_vs_torch_autograd(dev):
    """ConvSumFn / BcrnnFn against torch's own autograd of the same formulas in float64 (odd sizes: unaligned frames, ragged tiles)."""
    from cine_hip import autograd as ag
    import torch.nn.functional as F
    torch.manual_seed(3)
    T, ch, c, h, w = 4, 3, 5, 9, 7
    x = torch.randn(T, ch, h, w); hid = torch.randn(T, c, h, w)
    w_in = torch.randn(c, c + ch, 3, 3) * 0.2; w_hh = torch.randn(c, c, 3, 3) * 0.2; bias = torch.randn(c) * 0.1
    gout = torch.randn(T, c, h, w)

    def ref(x, hid, w_in, w_hh, bias):
        P = F.conv2d(torch.cat([hid, x], 1), w_in, bias, padding=1)
        hf, hb, hcur = [], [None] * T, torch.zeros(1, c, h, w, dtype=x.dtype)
        for t in range(T):
            hcur = F.relu(F.conv2d(hcur, w_hh, padding=1) + P[t:t + 1]); hf.append(hcur)
        hcur = torch.zeros(1, c, h, w, dtype=x.dtype)
        for t in range(T - 1, -1, -1):
            hcur = F.relu(F.conv2d(hcur, w_hh, padding=1) + P[t:t + 1]); hb[t] = hcur
        return torch.cat(hf) + torch.cat(hb)
    args64 = [a.double().requires_grad_(True) for a in (x, hid, w_in, w_hh, bias)]
    with torch.enable_grad():
        y64 = ref(*args64); y64.backward(gout.double())
    args = [a.clone().to(dev).requires_grad_(True) for a in (x, hid, w_in, w_hh, bias)]
    with torch.enable_grad():
        y = ag.BcrnnFn.apply(*args)
        y.backward(gout.to(dev))
    assert rel_err(y.detach().cpu(), y64.detach()) < 1e-5
    for a, b, name in zip(args, args64, ("x", "hid_iter", "w_in", "w_hh", "bias")):
        assert rel_err(a.grad.cpu(), b.grad) < 2e-5, name
    # conv pair + bias + addend + ReLU
    x0 = torch.randn(T, c, h, w); x1 = torch.randn(T, 2, h, w); wt = torch.randn(4, c + 2, 3, 3) * 0.2; b = torch.randn(4) * 0.1
    add = torch.randn(T, 4, h, w); gy = torch.randn(T, 4, h, w)
    a64 = [a.double().requires_grad_(True) for a in (x0, x1, wt, b, add)]
    with torch.enable_grad():
        y64 = F.relu(F.conv2d(torch.cat([a64[0], a64[1]], 1), a64[2], a64[3], padding=1) + a64[4]); y64.backward(gy.double())
    a32 = [a.clone().to(dev).requires_grad_(True) for a in (x0, x1, wt, b, add)]
    with torch.enable_grad():
        y = ag.ConvSumFn.apply(*a32, True)
        y.backward(gy.to(dev))
    assert rel_err(y.detach().cpu(), y64.detach()) < 1e-5
    for a, b_, name in zip(a32, a64, ("x0", "x1", "weight", "bias", "addend")):
        assert rel_err(a.grad.cpu(), b_.grad) < 2e-5, name


def test_training_takes_general_masks_and_rejects_malformed_ones(dev):
    """A mask that varies along w trains on the literal k-space chain (round 6: XPDNet and the CRNN models too; parity in
    test_masks_that_vary_along_w_*); a mask that does not broadcast against the k-space as (b|1, t|1, 1, h, w|1, 1) is refused loudly instead of
    being mis-indexed."""
    import reconstruction.models as M
    t, c, h, w = 3, 2, 16, 16
    mk = torch.randn(1, t, c, h, w, 2, device=dev)
    mask = (torch.rand(1, t, 1, h, w, 1, device=dev) > 0.5).to(torch.uint8)
    mask[:, :, :, h // 2 - 2:h // 2 + 2] = 1
    net = M.XPDNet(num_cascades=1, sens_chans=2, sens_pools=1, n_scales=1, n_filters_per_scale=[4], n_convs_per_scale=[1],
                   first_conv_n_filters=4, n_primal=2, dynamic_type="XF").to(dev).train()
    with torch.enable_grad():
        out = net(mk * mask, mask, acs=(6, 4))
        out.sum().backward()
    assert all(p.grad is None or torch.isfinite(p.grad).all() for p in net.parameters())
    with torch.no_grad():
        assert rel_err(net(mk * mask, mask, acs=(6, 4)).cpu(), out.detach().cpu()) < 1e-5
    for bad in (mask[:, :, :, :h - 1], mask.expand(1, t, 2, h, w, 1), mask[..., 0]):
        with torch.enable_grad(), pytest.raises(ValueError):
            net(mk * mask, bad, acs=(6, 4))


_FULL_SIZE = {
    "xpdnet_grad_cfg3": (lambda M: M.XPDNet(num_cascades=10, sens_chans=8, sens_pools=3, n_primal=5, dynamic_type="XT"), 6, (), 8, 0.01, False),
    "cinenet_grad_cfg4": (lambda M: M.CineNet(6, 6, 16, 3, "3D"), 7, ("lambda",), 6, 0.0, True),
    "rnn_grad_cfg5": (lambda M: M.VarNet_RNN(5, 8, 3, 16), 9, ("lambda",), 8, 0.0, False),
}


@pytest.mark.parametrize("name", sorted(_FULL_SIZE))
def test_other_configs_training_step_vs_reference_fingerprint(dev, golden, name):
    """cfg 3 / 4 / 5 at full size (15 coils x 15 frames x 200 x 200): the training step on the HIP path against strided fingerprints of
    the reference's parameter gradients, with the same bar as cfg 2 -- the larger of 5e-3 (max) / 3e-3 (L2) and 3x the reference's own
    movement under a 1e-6 input change.  The full-size code paths (vectorised staging with halo columns, chunked InstanceNorm backward
    on the 3-D volumes, 15-frame time sweeps) only run here."""
    import reconstruction.models as M
    from cine_hip import synth
    make, wseed, keep, accel, noise, needs_sens = _FULL_SIZE[name]
    g = golden(name)
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=accel, seed=0, noise_std=noise)
    net = make(M)
    synth.fill_parameters_(net, wseed, keep=keep)
    net = net.to(dev).train()
    with torch.enable_grad():
        loss, grads, out = _training_step(net, ex["masked_kspace"].to(dev), ex["mask"].to(dev), ex["target"].to(dev),
                                          extra=(ex["sens_maps"].to(dev),) if needs_sens else ())
    assert rel_err(out[:, :, ::4, ::4].cpu(), g["out_strided"]) < (2e-3 if name == "xpdnet_grad_cfg3" else TOL)    # cfg 3: see bench.py parity_note
    assert abs(float(loss) - float(g["loss"])) < 1e-4
    bad = {}
    for k in (k[6:] for k in g if k.startswith("grad::")):
        flat = grads[k].reshape(-1)
        got = flat[::max(1, flat.numel() // 256)].cpu()
        e = float((got.double() - torch.from_numpy(g[f"grad::{k}"]).double()).abs().max() / max(float(g[f"gmax::{k}"]), 1e-30))
        en = abs(float(grads[k].double().norm()) - float(g[f"gnorm::{k}"])) / max(float(g[f"gnorm::{k}"]), 1e-30)
        if e > max(5e-3, 3 * float(g[f"selfmax::{k}"])) or en > max(3e-3, 3 * float(g[f"selfnorm::{k}"])):
            bad[k] = (e, en, float(g[f"selfmax::{k}"]))
    assert not bad, bad


def test_plane_wgrad_kernel_bit_identical_to_general_kernel(dev):
    """The lean weight-gradient kernel of the U-Nets' plane-wide 3x3 convs (grad_kernels.hip: wgrad_plane_kernel) against
    wgrad_mfma_kernel on a whole XF-VarNet training step at cfg 2's plane shapes (2 cascades): every parameter gradient bit for bit
    -- same tile order, K split and reduction order.  cine_set_conv_plane bit 4 routes the weight gradients through the general kernel."""
    import reconstruction.models as M
    from cine_hip import synth
    from cine_hip._lib import lib
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=4, seed=0)
    res, counts = [], []
    try:
        for mask in (7, 7 | 16):
            cine_ops.set_conv_plane(mask)
            net = M.VarNet(2, 8, 3, 16, 3, "XF"); synth.fill_parameters_(net, 1); net = net.to(dev).train()
            lib().cine_diag_counter(0, 1); lib().cine_diag_counter(1, 1)
            with torch.enable_grad():
                out = net(ex["masked_kspace"].to(dev), ex["mask"].to(dev))
                (out - ex["target"].to(dev)).pow(2).sum().backward()
            torch.cuda.synchronize()
            counts.append((lib().cine_diag_counter(0, 1), lib().cine_diag_counter(1, 1)))
            res.append({k: p.grad.clone() for k, p in net.named_parameters()})
    finally:
        cine_ops.set_conv_plane(7)
    # the mask is a per-thread setting and loss.backward() runs on the autograd engine's thread: the Functions carry it across.
    # Proof that the two runs took DIFFERENT kernels: launch counts of (lean, general) plane-eligible weight gradients
    (lean7, gen7), (lean23, gen23) = counts
    assert lean7 > 0 and lean23 == 0 and gen23 >= gen7 + lean7, counts
    bad = [k for k in res[0] if not torch.equal(res[0][k], res[1][k])]
    assert not bad, bad


def test_training_forward_as_branches_gives_the_same_bits(dev):
    """The training forward of the cascade U-Nets as two concurrent branches (cine_unet2d_forward_branches with train = 1: the x-f / y-f planes write
    their halves of the ONE workspace layout cine_unet2d_backward reads) against the one-stream sequence: output and every gradient bit for bit."""
    import reconstruction.models as M
    from cine_hip import ops, synth
    from cine_hip._lib import lib
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=4, seed=0)
    res = []
    for nb in (1, 2):
        net = M.VarNet(1, 8, 3, 16, 3, "XF"); synth.fill_parameters_(net, 1); net = net.to(dev).train()
        lib().cine_diag_counter(2, 1)
        with ops.branches(nb), torch.enable_grad():
            out = net(ex["masked_kspace"].to(dev), ex["mask"].to(dev))
            (out - ex["target"].to(dev)).pow(2).sum().backward()
        torch.cuda.synchronize()
        assert (lib().cine_diag_counter(2, 1) > 0) == (nb > 1)
        res.append((out.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0])
    bad = [k for k in res[0][1] if not torch.equal(res[0][1][k], res[1][1][k])]
    assert not bad, bad


@pytest.mark.parametrize("T,ch,c,h,w,relu", [(15, 2, 16, 200, 200, True), (4, 2, 16, 40, 36, True), (5, 3, 6, 24, 20, True), (1, 2, 16, 40, 36, True), (6, 2, 16, 52, 16, False)])
def test_bcrnn_sweeps_in_c_bit_identical_to_the_step_loops_and_vs_torch(dev, T, ch, c, h, w, relu):
    """cine_bcrnn_sweep / cine_bcrnn_sweep_bwd (both loops over the frames of BCRNNlayer.forward, reference recurrent_varnet.py:236-254, and their
    back-propagation through time in ONE C call each; the ReLU mask rides in the conv epilogue as a gate) against the step-by-step composition
    (cine_crnn_step2 / conv + cine_relu_mask per frame): output, hidden-state gradients and every parameter gradient bit for bit; and against
    torch's autograd of the literal loops in float64."""
    from cine_hip import autograd as ag, ops
    from cine_hip._lib import lib
    x = rnd(1, T, ch, h, w).to(dev); hid = rnd(2, T, c, h, w).to(dev)
    w_in = (0.1 * rnd(3, c, c + ch, 3, 3)).to(dev); w_hh = (0.1 * rnd(4, c, c, 3, 3)).to(dev); bias = (0.1 * rnd(5, c)).to(dev)
    gout = rnd(6, T, c, h, w).to(dev)
    res = []
    old = ops.BCRNN_SWEEP_IN_C
    try:
        for in_c in (False, True):
            ops.BCRNN_SWEEP_IN_C = in_c
            leaves = [t_.clone().requires_grad_(True) for t_ in (x, hid, w_in, w_hh, bias)]
            lib().cine_diag_counter(3, 1)
            with ops.activation(relu=relu), torch.enable_grad():
                out = ag.BcrnnFn.apply(*leaves)
                out.backward(gout)
            torch.cuda.synchronize()
            assert (lib().cine_diag_counter(3, 1) > 0) == in_c
            res.append([out.detach()] + [t_.grad for t_ in leaves])
    finally:
        ops.BCRNN_SWEEP_IN_C = old
    for a, b in zip(*res):
        assert torch.equal(a, b)
    # the literal loops in float64 (recurrent_varnet.py:241-254 with the input terms batched)
    xd, hd, wi, wh, bd = (t_.double().cpu().requires_grad_(True) for t_ in (x, hid, w_in, w_hh, bias))
    F = torch.nn.functional
    act = torch.relu if relu else (lambda v: v)
    with torch.enable_grad():
        P = F.conv2d(torch.cat([hd, xd], 1), wi, bd, padding=1)
        hf = hb = torch.zeros(1, c, h, w, dtype=torch.float64)
        of, ob = [], []
        for t_ in range(T):
            hf = act(F.conv2d(hf, wh, padding=1) + P[t_:t_ + 1]); of.append(hf)
        for t_ in range(T - 1, -1, -1):
            hb = act(F.conv2d(hb, wh, padding=1) + P[t_:t_ + 1]); ob.append(hb)
        want = torch.cat(of) + torch.cat(ob[::-1])
        want.backward(gout.double().cpu())
    assert rel_err(res[1][0].cpu(), want.detach()) < 2e-5
    for got, ref in zip(res[1][1:], (xd, hd, wi, wh, bd)):
        if relu:        # a hidden value within rounding of 0 flips its ReLU mask between float32 and float64: isolated O(1) entries, so the bar is the L2 norm
            assert float((got.cpu().double() - ref.grad).norm() / ref.grad.norm().clamp_min(1e-30)) < 2e-3
        else:
            assert rel_err(got.cpu(), ref.grad) < 5e-5


@pytest.mark.parametrize("n,sets,h,w,chans,pools,p", [(4, 1, 24, 16, 8, 2, 0.3), (6, 2, 208, 16, 16, 3, 0.1), (3, 1, 40, 36, 4, 1, 0.5)])
def test_unet_dropout_with_a_fixed_mask_vs_oracle_autograd(dev, n, sets, h, w, chans, pools, p):
    """Dropout2d behind every LeakyReLU of the ConvBlocks (reference unet.py:22,40,159-168; training mode, drop_prob > 0) on the HIP path: the
    channel multipliers (0 or 1 / (1 - p)) are folded into the planes' InstanceNorm statistics records in the forward (cine_unet2d_forward_branches'
    `drop`) and enter the normalisation's backward once more (cine_unet2d_backward_drop).  Parity: the SAME multipliers replace the oracle's
    nn.Dropout2d modules; output, input gradient and every weight gradient against its float64 autograd; one and two weight sets; also as branches."""
    from cine_hip import autograd as ag, ops
    from cine_hip._lib import lib
    from cine_hip import synth
    from oracle import regularisers as R
    from reconstruction.models.denoisers.unet import Unet
    total = lib().cine_unet2d_drop_floats(n, chans, pools)
    gen = torch.Generator().manual_seed(5)
    mult = (torch.rand(total, generator=gen) >= p).float() / (1 - p)
    assert 0 < int((mult == 0).sum()) < total
    x = rnd(21, n, 2, h, w); gy = rnd(22, n, 2, h, w)

    class Mult(torch.nn.Module):
        def __init__(self, m):
            super().__init__(); self.m = m
        def forward(self, v):
            return v * self.m[:, :, None, None]

    nets_h, outs, per = [], [], n // sets
    want_gx = torch.empty(n, 2, h, w, dtype=torch.float64)
    ref_grads = []
    for k in range(sets):
        hnet = Unet(in_chans=2, out_chans=2, chans=chans, num_pool_layers=pools, drop_prob=p).train(); synth.fill_parameters_(hnet, 41 + k, keep=())
        rnet = R.Unet(in_chans=2, out_chans=2, chans=chans, num_pool_layers=pools, drop_prob=p).double().train()
        rnet.load_state_dict({kk: v.double() for kk, v in hnet.state_dict().items()}, strict=True)
        blocks = list(rnet.down_sample_layers) + [rnet.conv] + [rnet.up_conv[i] if i < pools - 1 else rnet.up_conv[i][0] for i in range(pools)]
        off = 0
        for bi, blk in enumerate(blocks):
            ch = blk.layers[0].out_channels
            for slot in (3, 7):
                m = mult[off:off + n * ch].view(n, ch)[k * per:(k + 1) * per].double()
                blk.layers[slot] = Mult(m); off += n * ch
        assert off == total
        xk = x[k * per:(k + 1) * per].double().requires_grad_(True)
        with torch.enable_grad():
            o = rnet(xk)
            o.backward(gy[k * per:(k + 1) * per].double())
        outs.append(o.detach()); want_gx[k * per:(k + 1) * per] = xk.grad
        ref_grads.append({kk: v.grad for kk, v in rnet.named_parameters()})
        nets_h.append(hnet.to(dev))
    want = torch.cat(outs)
    wts = ops.UnetWeights(nets_h)
    for nb in (1, 2):
        for net in nets_h: net.zero_grad()
        xd = x.to(dev).requires_grad_(True)
        with ops.fixed_dropout(mult.to(dev)), ops.branches(nb), torch.enable_grad():
            got = ag.unet2d(xd, wts)
            got.backward(gy.to(dev))
        assert rel_err(got.detach().cpu(), want) < 2e-5, nb
        assert rel_err(xd.grad.cpu(), want_gx) < 5e-5, nb
        for k, net in enumerate(nets_h):
            for kk, v in net.named_parameters():
                assert rel_err(v.grad.cpu(), ref_grads[k][kk]) < 5e-5, (nb, k, kk)
    # eval mode: no dropout (and no multipliers drawn)
    for net in nets_h: net.eval()
    assert wts.dropout_multipliers(n, dev) is None


def test_unet3d_dropout_with_a_fixed_mask_vs_oracle_autograd(dev):
    """Dropout3d in the 3-D U-Net (reference unet.py:24,159-168 with dims = 3): the multipliers are folded into the merged statistics record of every
    3x3x3 conv output (cine_unet3d_forward_train_drop) and enter cine_unet3d_backward_drop; same multipliers in the oracle's nn.Dropout3d slots."""
    from cine_hip import autograd as ag, ops, synth
    from cine_hip._lib import lib
    from oracle import regularisers as R
    from reconstruction.models.denoisers.unet import Unet
    n, d, h, w, chans, pools, p = 2, 8, 24, 20, 4, 2, 0.25
    total = lib().cine_unet2d_drop_floats(n, chans, pools)
    mult = (torch.rand(total, generator=torch.Generator().manual_seed(9)) >= p).float() / (1 - p)
    assert 0 < int((mult == 0).sum()) < total
    x = rnd(51, n, 2, d, h, w); gy = rnd(52, n, 2, d, h, w)

    class Mult(torch.nn.Module):
        def __init__(self, m):
            super().__init__(); self.m = m
        def forward(self, v):
            return v * self.m[:, :, None, None, None]
    hnet = Unet(in_chans=2, out_chans=2, chans=chans, num_pool_layers=pools, drop_prob=p, dims=3).train(); synth.fill_parameters_(hnet, 43, keep=())
    rnet = R.Unet(in_chans=2, out_chans=2, chans=chans, num_pool_layers=pools, drop_prob=p, dims=3).double().train()
    rnet.load_state_dict({kk: v.double() for kk, v in hnet.state_dict().items()}, strict=True)
    blocks = list(rnet.down_sample_layers) + [rnet.conv] + [rnet.up_conv[i] if i < pools - 1 else rnet.up_conv[i][0] for i in range(pools)]
    off = 0
    for blk in blocks:
        ch = blk.layers[0].out_channels
        for slot in (3, 7):
            blk.layers[slot] = Mult(mult[off:off + n * ch].view(n, ch).double()); off += n * ch
    assert off == total
    xr = x.double().requires_grad_(True)
    with torch.enable_grad():
        want = rnet(xr); want.backward(gy.double())
    hnet = hnet.to(dev)
    xd = x.to(dev).requires_grad_(True)
    with ops.fixed_dropout(mult.to(dev)), torch.enable_grad():
        got = hnet(xd)
        got.backward(gy.to(dev))
    assert rel_err(got.detach().cpu(), want.detach()) < 2e-5
    assert rel_err(xd.grad.cpu(), xr.grad) < 5e-5
    ref = dict(rnet.named_parameters())
    for kk, v in hnet.named_parameters():
        assert rel_err(v.grad.cpu(), ref[kk].grad) < 5e-5, kk


def test_sensitivity_model_with_dropout_trains_reproducibly(dev):
    """SensitivityModel(drop_prob > 0) in training mode (reference varnet.py:29-36 hands drop_prob to its NormUnet): forward + backward on the HIP path,
    reproducible under torch.manual_seed like nn.Dropout2d, different from the eval-mode output, identical to it with the draw switched off."""
    from reconstruction.models.varnet import SensitivityModel
    from cine_hip import ops, synth
    ex = synth.make_cine_slice(4, 3, 48, 40, accel=4, center_lines=6, seed=5)
    k, mk = ex["masked_kspace"].to(dev), ex["mask"].to(dev)
    net = SensitivityModel(4, 2, drop_prob=0.25); synth.fill_parameters_(net, 3); net = net.to(dev).train()
    outs = []
    for seed in (7, 7, 8):
        torch.manual_seed(seed)
        net.zero_grad()
        with torch.enable_grad():
            o = net(k, mk)
            o.pow(2).sum().backward()
        assert all(p_.grad is not None and torch.isfinite(p_.grad).all() for p_ in net.parameters())
        outs.append(o.detach().clone())
    assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], outs[2])
    with torch.no_grad():
        ev = net.eval()(k, mk)
        net.train()
        with ops.fixed_dropout(False):
            assert torch.equal(net(k, mk), ev)
        assert not torch.equal(net(k, mk), ev)          # training mode without autograd still drops (nn.Dropout2d does)


_LINEAR = {
    "varnet_grad_cfg2_linear": (lambda M: M.VarNet(6, 8, 3, 16, 3, "XF"), 1, 4, False),
    "cinenet_grad_cfg4_linear": (lambda M: M.CineNet(6, 6, 16, 3, "3D"), 7, 6, True),
    # cfg 3: the MWCNN backward (DWT / IWT adjoints, 200 x 16 planes); cfg 5: back-propagation through the 15-frame BCRNN time sweeps
    "xpdnet_grad_cfg3_linear": (lambda M: M.XPDNet(num_cascades=10, sens_chans=8, sens_pools=3, n_primal=5, dynamic_type="XT"), 6, 8, False),
    "rnn_grad_cfg5_linear": (lambda M: M.VarNet_RNN(5, 8, 3, 16), 9, 8, False),
}


@pytest.mark.parametrize("name", sorted(_LINEAR))
def test_full_size_gradients_with_identity_activations_vs_reference(dev, golden, name, monkeypatch):
    """The sharp full-size pin of the backward pass.  With the LeakyReLUs replaced by the identity on BOTH sides (the reference run of
    make_golden.py patches F.leaky_relu; here slope 1 as the per-call argument of every entry point) the networks have no kinks: the
    reference's float32 gradients are a smooth function of the input and their distance from its own float64 gradients (stored per
    parameter) is pure rounding.  An indexing or staging error in a full-size code path (200-wide planes, 15 x 200 x 200 volumes) moves a
    gradient by O(1) of its size; the bar is 1e-4 of each tensor's largest entry, or twice the reference's own float32 floor where that
    is larger -- against the kinked fixtures the same tensors are only held to 5e-3."""
    import reconstruction.models as M
    from cine_hip import ops, synth
    from cine_hip._lib import lib
    make, wseed, accel, needs_sens = _LINEAR[name]
    g = golden(name)
    cfg3 = name.startswith("xpdnet")
    ex = synth.make_cine_slice(15, 15, 200, 200, accel=accel, seed=0, noise_std=0.01 if cfg3 else 0.0)
    net = make(M)
    synth.fill_parameters_(net, wseed, **(dict(keep=()) if cfg3 else {}))
    net = net.to(dev).train()
    # identity activations as per-call arguments (no library state): LeakyReLU slope 1, nn.ReLU of the CRNN cells off -- what
    # make_golden.py's F.leaky_relu / F.relu patches do on the reference side
    with ops.activation(slope=1.0, relu=False), torch.enable_grad():
        loss, grads, out = _training_step(net, ex["masked_kspace"].to(dev), ex["mask"].to(dev), ex["target"].to(dev),
                                          extra=(ex["sens_maps"].to(dev),) if needs_sens else ())
    assert rel_err(out[:, :, ::4, ::4].cpu(), g["out_strided"]) < TOL
    assert abs(float(loss) - float(g["loss64"])) < 1e-4
    bad, worst = {}, 0.0
    for k in (k[8:] for k in g if k.startswith("grad64::")):
        flat = grads[k].reshape(-1)
        got = flat[::max(1, flat.numel() // 256)].cpu().double()
        want = torch.from_numpy(g[f"grad64::{k}"]).double()
        gmax = max(float(g[f"gmax::{k}"]), 1e-300)
        e = float((got - want).abs().max() / gmax)
        en = abs(float(grads[k].double().norm()) - float(g[f"gnorm::{k}"])) / max(float(g[f"gnorm::{k}"]), 1e-300)
        bar = max(1e-4, 2 * float(g[f"floormax::{k}"])); barn = max(1e-4, 2 * float(g[f"floornorm::{k}"]))
        worst = max(worst, e / bar)
        if e > bar or en > barn:
            bad[k] = (e, en, float(g[f"floormax::{k}"]), float(g[f"floornorm::{k}"]))
    assert not bad, bad


def _odd_mask(t, h):
    m = torch.zeros(1, t, 1, h, 1, 1, dtype=torch.uint8)
    c0 = h // 2 - 2
    m[:, :, :, c0:c0 + 4] = 1
    for f in range(t):
        for r in (1, 6, h - 3):
            m[:, f, :, (r + 2 * f) % h] = 1
    m[:, 0, :, c0 - 1] = 0
    m[:, 0, :, c0 + 4] = 0
    return m


@pytest.mark.parametrize("family", ["varnet_XF", "varnet_XT", "cinenet_XF", "xpdnet_XT", "varnet_3D", "varnet_rnn"])
def test_training_gradients_on_odd_shapes_vs_oracle_float64(dev, family):
    """Odd extents everywhere (7 frames, 23 x 19 pixels: ragged tiles, zero pads on both sides, the up-path pad of the 3-D U-Net, unaligned
    frames in the time sweep): the HIP training gradients against the oracle's float64 autograd of the same model, on three different
    k-spaces.  A float32 run can sit on a LeakyReLU / ReLU kink the float64 run does not (make_golden.py:_kink_stability: one of five
    k-spaces moves two XPDNet gradient tensors by 1e-2, the others agree to 1e-5), an indexing mistake is there for every input: the bar
    is 1e-3 of each tensor's largest gradient on the BEST of the three inputs, and 5e-2 on every one of them."""
    import reconstruction.models as M
    from reconstruction.utils import SSIMLoss
    from cine_hip import synth
    from oracle import varnet_ref as V, cinenet_ref as C, xpdnet_ref as X, recurrent_ref as R
    t, c, h, w = 7, 3, 23, 19
    kwx = dict(num_cascades=2, sens_chans=4, sens_pools=2, n_scales=2, n_filters_per_scale=[8, 16], n_convs_per_scale=[2, 1],
               first_conv_n_filters=8, n_primal=2, dynamic_type="XT")
    make = {"varnet_XF": (lambda m: m.VarNet(2, 4, 2, 4, 2, "XF"), V), "varnet_XT": (lambda m: m.VarNet(2, 4, 2, 4, 2, "XT"), V),
            "varnet_3D": (lambda m: m.VarNet(2, 4, 2, 4, 2, "3D"), V), "cinenet_XF": (lambda m: m.CineNet(2, 3, 4, 2, "XF"), C),
            "xpdnet_XT": (lambda m: m.XPDNet(**kwx), X), "varnet_rnn": (lambda m: m.VarNet_RNN(2, 4, 2, 5), R)}[family]
    net = make[0](M)
    synth.fill_parameters_(net, 11, keep=() if family == "xpdnet_XT" else ("lambda",))
    ref = make[0](make[1]).double()
    ref.load_state_dict({k: v.double() for k, v in net.state_dict().items()}, strict=True)
    net = net.to(dev).train()
    mask = _odd_mask(t, h)
    sens = rnd(32, 1, 1, c, h, w, 2)
    sens = sens / sens.pow(2).sum(dim=(2, 5), keepdim=True).sqrt()
    target = rnd(33, 1, t, h - 4, w - 2).abs() + 0.1
    extra = (sens,) if family.startswith("cinenet") else ()

    def loss_of(model, mk, device, dtype):
        out = model(mk.to(device, dtype), mask.to(device), *(e.to(device, dtype) for e in extra))
        h0, w0 = (out.shape[-2] - target.shape[-2]) // 2, (out.shape[-1] - target.shape[-1]) // 2
        crop = out[..., h0:h0 + target.shape[-2], w0:w0 + target.shape[-1]]
        tg = target.to(device, dtype)
        lossf = SSIMLoss().to(device)
        return (lossf.double() if dtype == torch.float64 else lossf)(crop.unsqueeze(1), tg.unsqueeze(1), data_range=tg.max())
    best, worst = {}, {}
    for seed in (31, 41, 51):
        mk = rnd(seed, 1, t, c, h, w, 2) * mask
        ref.zero_grad(); net.zero_grad()
        with torch.enable_grad():
            l64 = loss_of(ref, mk, torch.device("cpu"), torch.float64); l64.backward()
            l32 = loss_of(net, mk, dev, torch.float32); l32.backward()
        assert abs(float(l32) - float(l64)) < 1e-4
        want = dict(ref.named_parameters())
        for k, p in net.named_parameters():
            e = rel_err(p.grad.cpu(), want[k].grad.float())
            best[k] = min(best.get(k, 1e9), e); worst[k] = max(worst.get(k, 0.0), e)
    bad = {k: (best[k], worst[k]) for k in best if best[k] > 1e-3 or worst[k] > 5e-2}
    assert not bad, bad


@pytest.mark.parametrize("family", ["varnet_XF", "cinenet_XF", "cinenet_3D"])
def test_masks_that_vary_along_w_inference_and_training_vs_oracle_float64(dev, family):
    """A sampling mask that varies along w (b|1, t, 1, h, w, 1) -- varnet.py:281-282 and cinenet.py:129 multiply by any mask that
    broadcasts -- is served by the literal k-space chain (coil operators through their kernels and adjoints, the DC line term by
    term): forward and training gradients against the oracle's float64 autograd.  VarNet gets its maps from the caller here: the
    reference's sens-net reads the ACS window off a 1-D mask (varnet.py:64-68) and has no meaning for a 2-D one."""
    import reconstruction.models as M
    from cine_hip import synth
    from oracle import varnet_ref as V, cinenet_ref as C
    t, c, h, w = 5, 3, 20, 18
    make = {"varnet_XF": (lambda m: m.VarNet(2, 4, 2, 4, 2, "XF"), V), "cinenet_XF": (lambda m: m.CineNet(2, 3, 4, 2, "XF"), C),
            "cinenet_3D": (lambda m: m.CineNet(2, 2, 4, 2, "3D"), C)}[family]
    net = make[0](M)
    synth.fill_parameters_(net, 13, keep=("lambda",))
    ref = make[0](make[1]).double()
    ref.load_state_dict({k: v.double() for k, v in net.state_dict().items()}, strict=True)
    net = net.to(dev).train()
    g = torch.Generator().manual_seed(5)
    mask = (torch.rand(1, t, 1, h, w, 1, generator=g) < 0.4).to(torch.uint8)
    mask[:, :, :, h // 2 - 2:h // 2 + 2, w // 2 - 3:w // 2 + 3] = 1
    sens = rnd(32, 1, 1, c, h, w, 2)
    sens = sens / sens.pow(2).sum(dim=(2, 5), keepdim=True).sqrt()
    target = rnd(33, 1, t, h, w).abs() + 0.1

    def run(model, mk, device, dtype):
        mk, sm, m = mk.to(device, dtype), sens.to(device, dtype), mask.to(device)
        out = model(mk, m, sm)            # (the oracle's VarNet takes the caller's maps the same way: varnet.py:145-151 after :144)
        return out, ((out - target.to(device, dtype)) ** 2).mean()
    best, worst = {}, {}
    for seed in (31, 41, 51):
        mk = rnd(seed, 1, t, c, h, w, 2) * mask
        ref.zero_grad(); net.zero_grad()
        with torch.enable_grad():
            o64, l64 = run(ref, mk, torch.device("cpu"), torch.float64); l64.backward()
            o32, l32 = run(net, mk, dev, torch.float32); l32.backward()
        assert rel_err(o32.detach().cpu(), o64.detach().float()) < 2e-5, seed
        with torch.no_grad():
            oi, _ = run(net, mk, dev, torch.float32)                      # the inference path on the same mask
        assert rel_err(oi.cpu(), o64.detach().float()) < 2e-5, seed
        want = {k: p for k, p in ref.named_parameters() if p.grad is not None}
        assert len(want) >= 10
        for k, p in net.named_parameters():
            if k not in want:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
                continue
            e = rel_err(p.grad.cpu(), want[k].grad.float())
            best[k] = min(best.get(k, 1e9), e); worst[k] = max(worst.get(k, 0.0), e)
    bad = {k: (best[k], worst[k]) for k in best if best[k] > 1e-3 or worst[k] > 5e-2}
    assert not bad, bad


@pytest.mark.parametrize("family", ["xpdnet", "xpdnet_dual", "varnet_rnn", "cinenet_rnn", "xpdnet_rnn", "xpdnet_rnn_dual"])
def test_masks_that_vary_along_w_xpdnet_and_crnn_models_vs_oracle_float64(dev, family, monkeypatch):
    """A sampling mask that varies along w in XPDNet and the convolutional-RNN hybrids (reference xpdnet.py:128-131, 161-167 and recurrent_*.py
    multiply by whatever mask broadcasts): the literal k-space chain -- A^H m (m A x0 - k_ref) term by term, the soft-DC line of the CRNN-VarNet,
    the conjugate gradient's operator -- in inference AND training against the oracle's float64 autograd.  The sensitivity networks read their ACS
    window off a 1-D mask (varnet.py:64-68): with a 2-D mask the caller passes ``acs=`` (the oracle's window function is pinned to the same rows)."""
    import reconstruction.models as M
    from cine_hip import synth
    from oracle import recurrent_ref as R, xpdnet_ref as X, varnet_ref as V
    t, c, h, w = 4, 3, 24, 20
    kw = dict(num_cascades=2, sens_chans=4, sens_pools=2, n_scales=2, n_filters_per_scale=[8, 16], n_convs_per_scale=[1, 1], first_conv_n_filters=8,
              n_primal=2, dynamic_type="XF", weight_sharing=False)
    make, needs_sens = {
        "xpdnet": (lambda m: m.XPDNet(primal_only=True, **kw), False), "xpdnet_dual": (lambda m: m.XPDNet(primal_only=False, **kw), False),
        "varnet_rnn": (lambda m: m.VarNet_RNN(2, 4, 2, 6), False), "cinenet_rnn": (lambda m: m.CineNet_RNN(2, 3, 6), True),
        "xpdnet_rnn": (lambda m: m.XPDNet_RNN(2, 4, 2, 6, True, 2, 1), False), "xpdnet_rnn_dual": (lambda m: m.XPDNet_RNN(2, 4, 2, 6, False, 2, 1), False)}[family]
    net = make(M)
    synth.fill_parameters_(net, 17, keep=("lambda",))
    ref = make(X if family.startswith("xpdnet") and "rnn" not in family else R).double()
    ref.load_state_dict({k: v.double() for k, v in net.state_dict().items()}, strict=True)
    net = net.to(dev).train(); ref.train()
    acs = (9, 6)                                     # rows [9, 15): what the sens-nets keep; pinned on both sides
    for mod in (V, X, R):
        for name in dir(mod):
            cls = getattr(mod, name)
            if isinstance(cls, type) and hasattr(cls, "acs_window"):
                monkeypatch.setattr(cls, "acs_window", staticmethod(lambda mask: acs))
    g = torch.Generator().manual_seed(6)
    mask = (torch.rand(1, t, 1, h, w, 1, generator=g) < 0.4).to(torch.uint8)
    mask[:, :, :, 9:15, w // 2 - 4:w // 2 + 4] = 1
    sens = rnd(32, 1, 1, c, h, w, 2)
    sens = sens / sens.pow(2).sum(dim=(2, 5), keepdim=True).sqrt()
    target = rnd(33, 1, t, h, w).abs() + 0.1
    mk = rnd(34, 1, t, c, h, w, 2) * mask

    def run(model, device, dtype, hip):
        a = (mk.to(device, dtype), mask.to(device))
        if needs_sens:
            out = model(*a, sens.to(device, dtype))
        else:
            out = model(*a, acs=acs) if hip else model(*a)
        return out, ((out - target.to(device, dtype)) ** 2).mean()
    with torch.enable_grad():
        o64, l64 = run(ref, torch.device("cpu"), torch.float64, False); l64.backward()
        o32, l32 = run(net, dev, torch.float32, True); l32.backward()
    assert rel_err(o32.detach().cpu(), o64.detach().float()) < 5e-5
    with torch.no_grad():
        oi, _ = run(net, dev, torch.float32, True)                          # the inference path on the same mask
    assert rel_err(oi.cpu(), o64.detach().float()) < 5e-5
    want = {k: p for k, p in ref.named_parameters() if p.grad is not None}
    assert len(want) >= 8
    bad = {}
    for k, p in net.named_parameters():
        if k not in want:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        e = float((p.grad.cpu().double() - want[k].grad).norm() / want[k].grad.norm().clamp_min(1e-30))
        if e > 2e-3:                                 # (L2: a ReLU / LeakyReLU kink that flips between float32 and float64 moves isolated entries)
            bad[k] = e
    assert not bad, bad


def test_masks_that_broadcast_along_batch_and_time_take_the_row_mask_kernels(dev, golden):
    """(1, 1, 1, h, 1, 1) (one pattern for every frame) and float masks are what the reference's ``*`` accepts: they are expanded to the
    (b, t, 1, h, 1, 1) row layout once and give the row-mask path's bits."""
    import reconstruction.models as M
    from cine_hip import ops
    g = golden("varnet_grad")
    net = M.VarNet(2, 4, 2, 4, 2, "XF")
    net.load_state_dict(state_dict_from(g, "XF::sd::"), strict=True)
    net = net.to(dev)
    mk, mask = torch.from_numpy(g["masked_kspace"]).to(dev), torch.from_numpy(g["mask"]).to(dev)
    b, t, c, h, w, _ = mk.shape
    shared = mask[:, :1].contiguous()                                            # frame 0's pattern for every frame
    full = shared.expand(b, t, 1, h, 1, 1).contiguous()
    mk = mk * full
    want = net(mk, full)
    assert torch.equal(net(mk, shared), want) and torch.equal(net(mk, shared.float()), want)
    assert ops.is_row_mask(ops.as_mask_u8(shared, mk), mk)
    with pytest.raises(ValueError):
        net(mk, mask[:, :, :, : h - 1])                                          # does not broadcast
    cn = M.CineNet(2, 2, 4, 2, "XF").to(dev)
    sens = torch.randn(b, 1, c, h, w, 2, device=dev)
    assert torch.equal(cn(mk, shared, sens), cn(mk, full, sens))


def test_inference_path_untouched_by_grad_mode(dev, golden):
    """With autograd off the drop-in model takes the inference path (bit-identical to a no_grad call), and a grad-mode forward
    returns the same values to rounding."""
    import reconstruction.models as M
    g = golden("varnet_grad")
    net = M.VarNet(2, 4, 2, 4, 2, "XF")
    net.load_state_dict(state_dict_from(g, "XF::sd::"), strict=True)
    net = net.to(dev)
    mk, mask = torch.from_numpy(g["masked_kspace"]).to(dev), torch.from_numpy(g["mask"]).to(dev)
    a = net(mk, mask)
    with torch.no_grad():
        b = net(mk, mask)
    with torch.enable_grad():
        c = net(mk, mask)
    assert torch.equal(a, b) and not a.requires_grad and c.requires_grad
    assert rel_err(c.detach().cpu(), a.cpu()) < 1e-5
    assert rel_err(net(mk, mask.float()).cpu(), a.cpu()) == 0.0           # a float 0/1 mask (apply_mask's return) is accepted


@pytest.mark.parametrize("t,h,w", [(15, 180, 180), (3, 20, 18), (2, 7, 9)])
def test_ssim_loss_kernels_vs_torch_formula(dev, t, h, w):
    """cine_ssim_loss / cine_ssim_loss_bwd (reference utils/losses.py:25-58) against the same formula in float64 torch ops on the
    CPU: loss and d loss / d reconstruction; through SSIMLoss.forward's (1, 1, t, h, w) calling convention."""
    from reconstruction.utils import SSIMLoss
    rs = np.random.RandomState(t * 100 + h)
    tgt = torch.from_numpy(rs.uniform(0, 1.5, size=(t, h, w)).astype(np.float32))
    rec = (tgt + torch.from_numpy((0.1 * rs.standard_normal((t, h, w))).astype(np.float32))).clamp_min(0)
    with torch.enable_grad():
        r64 = rec.double().requires_grad_(True)
        want = SSIMLoss().double()(r64[None, None], tgt.double()[None, None], tgt.max())
        want.backward()
        rd = rec.to(dev).requires_grad_(True)
        got = SSIMLoss().to(dev)(rd[None, None], tgt.to(dev)[None, None], tgt.max())
        (3.0 * got).backward()
    assert abs(float(got) - float(want)) < 2e-6
    assert rel_err(rd.grad.cpu() / 3.0, r64.grad) < 1e-5


def test_ssim_loss_kernel_vs_reference_golden(golden, dev):
    """The reference's SSIMLoss value on a (1, 1, 15, 180, 180) pair (metrics.npz)."""
    from reconstruction.utils import SSIMLoss
    g = golden("metrics")
    rs = np.random.RandomState(int(g["seed"]))
    tgt = torch.from_numpy(rs.uniform(0, 1.5, size=(15, 180, 180)).astype(np.float32))
    rec = (tgt + torch.from_numpy((0.1 * rs.standard_normal((15, 180, 180))).astype(np.float32))).clamp_min(0)
    got = SSIMLoss().to(dev)(rec.to(dev)[None, None], tgt.to(dev)[None, None], torch.tensor([1.0]))
    assert abs(float(got) - float(g["ssim_loss"])) < 2e-6


def test_training_loop_tracks_the_oracle_and_learns(dev):
    """Ten Adam steps of the reference's training recipe (pl_modules/varnet_module.py:97-113, 151-154) on the drop-in XF-VarNet and on
    the CPU oracle from the same initial weights: the loss trajectories agree step by step and the loss goes down -- the HIP
    gradients are good enough to train with, not just close at step 0."""
    import reconstruction.models as M
    from reconstruction.data import transforms
    from reconstruction.utils import SSIMLoss
    from oracle import varnet_ref as V
    from cine_hip import synth
    ex = synth.make_cine_slice(5, 3, 24, 20, accel=4, center_lines=4, seed=7, noise_std=0.01)
    hip = M.VarNet(2, 4, 2, 4, 2, "XF")
    synth.fill_parameters_(hip, 8)
    ref = V.VarNet(2, 4, 2, 4, 2, "XF")
    ref.load_state_dict(hip.state_dict())
    hip = hip.to(dev).train()

    def run(model, device):
        mk, mask, target = ex["masked_kspace"].to(device), ex["mask"].to(device), ex["target"].to(device)
        lossf = SSIMLoss().to(device)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        losses = []
        with torch.enable_grad():
            for _ in range(10):
                opt.zero_grad()
                out = model(mk, mask)
                tgt, o = transforms.center_crop_to_smallest(target, out)
                loss = lossf(o.unsqueeze(1), tgt.unsqueeze(1), tgt.max())
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
        return losses
    lh, lr_ = run(hip, dev), run(ref, torch.device("cpu"))
    assert lh[-1] < lh[0] - 1e-3, lh                       # it learns
    for a, b in zip(lh, lr_):
        assert abs(a - b) < 2e-3 * max(abs(b), 1e-3), (lh, lr_)


def test_batched_weight_packs_equal_the_per_tensor_packs_and_follow_the_parameters(dev):
    """cine_pack_desc / cine_pack_batch (training re-packs every weight after every optimiser step in ONE launch into persistent buffers) against the
    per-tensor entry points, bit for bit: forward and input-gradient packings of a 2-D U-Net (3x3, transpose, 1x1 weights, odd channel counts) and of
    an MWCNN; after an in-place parameter update the same buffers hold the new packs."""
    from reconstruction.models.denoisers.unet import Unet
    from reconstruction.models.denoisers.mwcnn import MWCNN
    from cine_hip import ops, synth
    unet = Unet(5, 2, in_chans=3, out_chans=2); synth.fill_parameters_(unet, 3, keep=()); unet = unet.to(dev)
    mw = MWCNN(6, 4, n_scales=2, n_filters_per_scale=[8, 16], n_convs_per_scale=[2, 1], n_first_convs=1, first_conv_n_filters=8, res=False)
    synth.fill_parameters_(mw, 4, keep=()); mw = mw.to(dev)

    def check_all(w, items_fwd):
        for train_ptrs, holder, items in ((w.pointers(train=True), w._tp, items_fwd),
                                          (w.dgrad_pointers(), w._tdp, [(None if k == "raw" else k + "d", p) for k, p in items_fwd])):
            off = 0
            for (k, p), ptr in zip(items, list(train_ptrs)):
                if k is None:
                    assert ptr is None; continue
                if k == "raw":
                    assert ptr == p.data_ptr(); continue
                want = ops._pack(k, p)
                assert ptr == holder.flat.data_ptr() + 4 * off
                assert torch.equal(holder.flat[off:off + want.numel()], want), k
                off += (want.numel() + 63) // 64 * 64
    u = ops.UnetWeights([unet])
    items_u = [(k, p) for seq in u._params() for k, p in seq]
    m = ops.MwcnnWeights(mw)
    items_m = list(m._params())
    check_all(u, items_u); check_all(m, items_m)
    addr = (u._tp.flat.data_ptr(), m._tdp.flat.data_ptr())
    with torch.no_grad():
        for p in list(unet.parameters()) + list(mw.parameters()):
            p.mul_(1.5).add_(0.01)
    check_all(u, items_u); check_all(m, items_m)
    assert addr == (u._tp.flat.data_ptr(), m._tdp.flat.data_ptr())          # re-packed in place


@pytest.mark.parametrize("family", ["varnet_XF", "cinenet_3D"])
def test_training_step_captured_in_one_hipgraph_matches_the_eager_step(dev, family):
    """cine_hip.train.GraphedTrainingStep: forward + SSIMLoss + backward (two streams: the weight gradients' side lane is captured as a branch) + a
    capturable Adam in ONE hipGraph.  Replaying it is the eager step -- the same loss sequence from the same start -- and the parameters move;
    the weight packs the step makes are nodes of the graph (ops.training_capture), and the caches they bypass are invalidated afterwards."""
    import reconstruction.models as M
    from reconstruction.models.varnet import SensitivityModel
    from reconstruction.utils import SSIMLoss
    from cine_hip import synth, train, ops
    ex = synth.make_cine_slice(6, 3, 24, 16, accel=4, center_lines=4, seed=3, noise_std=0.01)
    mk, mask, target = ex["masked_kspace"].to(dev), ex["mask"].to(dev), ex["target"].to(dev)
    if family == "varnet_XF":
        make, extra, kw = (lambda: M.VarNet(2, 4, 2, 4, 2, "XF")), (), {"acs": SensitivityModel.acs_window(mask)}
    else:
        make, extra, kw = (lambda: M.CineNet(2, 2, 4, 2, "3D")), (ex["sens_maps"].to(dev),), {}
    lossf = SSIMLoss().to(dev)
    loss_fn = lambda out, tgt: lossf(out.unsqueeze(1), tgt.unsqueeze(1), tgt.max())

    def fresh(capturable):
        net = make(); synth.fill_parameters_(net, 4); net = net.to(dev).train()
        return net, torch.optim.Adam(net.parameters(), lr=1e-3, capturable=capturable)
    net, opt = fresh(True)
    eager = []
    with torch.enable_grad():
        for _ in range(6):
            opt.zero_grad(set_to_none=True)
            loss = loss_fn(net(mk, mask, *extra, **kw), target)
            loss.backward(); opt.step()
            eager.append(float(loss.detach()))
    net, opt = fresh(True)
    w0 = [p.detach().clone() for p in net.parameters()]
    epoch = ops.cache_epoch()
    gs = train.GraphedTrainingStep(net, loss_fn, opt, (mk, mask) + extra, target, forward_kwargs=kw, warmup=2)
    assert ops.cache_epoch() > epoch                                   # packs made during the capture are not served to later eager calls
    graphed = [float(l) for l in gs.warmup_losses] + [float(gs.step(mk, mask, *extra, target=target).clone()) for _ in range(4)]
    for a, b in zip(graphed, eager):
        assert abs(a - b) < 5e-4 * max(abs(b), 1e-3), (graphed, eager)
    assert graphed[-1] < graphed[0]
    assert any(float((p.detach() - q).abs().max()) > 0 for p, q in zip(net.parameters(), w0))
    with torch.no_grad():                                              # and the model is usable eagerly afterwards (re-packed weights)
        out = net.eval()(mk, mask, *extra)
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("n,in_ch,out_ch,scales,nf,nc,first,h,w", [(3, 6, 4, 2, [8, 16], [2, 1], 8, 16, 8), (4, 12, 10, 3, [16, 32, 64], [2, 2, 2], 16, 32, 16)])
def test_mwcnn_backward_vs_oracle(dev, n, in_ch, out_ch, scales, nf, nc, first, h, w):
    """cine_mwcnn_backward (reference denoisers/mwcnn.py:135-179 under autograd): every weight / bias gradient and the input gradient,
    incl. the Haar DWT / IWT adjoints and the additive skips; the second case is XPDNet's default topology."""
    from reconstruction.models.denoisers import MWCNN
    from oracle import xpdnet_ref as R
    from cine_hip import synth
    kw = dict(in_chans=in_ch, out_chans=out_ch, n_scales=scales, n_filters_per_scale=nf, n_convs_per_scale=nc, first_conv_n_filters=first)
    hip = MWCNN(**kw).to(dev)
    synth.fill_parameters_(hip, 51, keep=())
    ref = R.MWCNN(**kw)
    ref.load_state_dict(hip.state_dict())
    x, gy = rnd(16, n, in_ch, h, w), rnd(17, n, out_ch, h, w)
    with torch.enable_grad():
        xr = x.clone().requires_grad_(True)
        want = _grads(ref, (ref(xr) * gy).sum())
        xh = x.to(dev).requires_grad_(True)
        yh = hip(xh)
        got = _grads(hip, (yh * gy.to(dev)).sum())
    assert rel_err(yh.detach().cpu(), ref(x)) < 1e-5
    _cmp(got, want, "mwcnn weight gradients")
    assert rel_err(xh.grad.cpu(), xr.grad) < TOL
