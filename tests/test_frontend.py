"""the data front-end and the ESPIRiT calibration.

CPU part: the oracle (oracle/frontend_ref.py) against the reference-generated golden (tests/golden/frontend.npz, made by
the reference's own filtered_crop_center_and_slices / center_crop and the numpy lines of mri_data.py:283-303) and, for
ESPIRiT -- parity unpinned, the reference shells out to BART -- against the analytic coil maps of the synthetic phantom.
GPU part: the HIP path (through the C ABI) against the same golden and against the oracle.
"""
import numpy as np
import pytest
import torch

from conftest import rel_err


def _cplx(a):
    a = np.asarray(a)
    return torch.view_as_real(torch.from_numpy(np.ascontiguousarray(a.astype(np.complex64)))).contiguous()


def _phantom(t, c, n, seed=0):
    from cine_hip import synth
    ex = synth.make_cine_slice(t, c, n, n, accel=4, center_lines=10, seed=seed)
    k = torch.view_as_complex(ex["kspace"][0].contiguous()).numpy()                  # (t, c, n, n)
    s = torch.view_as_complex(ex["sens_maps"][0, 0].contiguous()).numpy()            # (c, n, n)
    tgt = ex["target"][0].numpy().mean(0)
    sn = s / np.sqrt((np.abs(s) ** 2).sum(0, keepdims=True))
    sn = sn * np.exp(-1j * np.angle(sn[:1]))                                         # unit norm, coil 0 real: ESPIRiT's gauge
    return k, sn, tgt > 0.1 * tgt.max()


# ------------------------------------------------------------------ CPU: oracle pins
def test_oracle_frontend_vs_reference_golden(golden):
    from oracle import frontend_ref as F
    g = golden("frontend")
    k, filt = F.prepare_slice(g["raw"], tuple(g["crop_shape"]), int(g["n_slices"]), tuple(g["filter_size"]))
    assert rel_err(np.stack([filt.real, filt.imag]), np.stack([g["images_filter"].real, g["images_filter"].imag])) < 2e-6
    assert rel_err(np.stack([k.real, k.imag]), np.stack([g["kspace"].real, g["kspace"].imag])) < 2e-6
    tgt = F.combine_target(g["images_filter"], g["sens"], tuple(g["crop_target"]))
    assert tgt.shape == g["target"].shape and rel_err(tgt, g["target"]) < 2e-6
    # scipy's reflect boundary for an overhang longer than the axis (radius 3 on 2 frames)
    from scipy.ndimage import gaussian_filter
    x = np.random.RandomState(3).standard_normal((2, 3, 5, 4)).astype(np.float32)
    want = gaussian_filter(x, sigma=[0.7, 0.0, 0.3, 0.3])
    got = x
    for ax, s in enumerate([0.7, 0.0, 0.3, 0.3]):
        got = F.gaussian_filter_axis(got, s, ax)
    assert np.abs(got - want).max() < 1e-6


def test_oracle_espirit_recovers_analytic_maps():
    from oracle import frontend_ref as F
    k, sn, sup = _phantom(5, 6, 64)
    maps, lam = F.espirit_maps(k.mean(0), r=24)
    sup = sup & (lam >= 0.8)
    assert sup.sum() > 1000
    assert abs(float(lam[sup].mean()) - 1.0) < 1e-3 and float(lam[sup].min()) > 0.99
    err = np.abs(maps - sn)[:, sup]
    assert float(np.sqrt((err ** 2).mean())) < 5e-3 and float(np.quantile(err, 0.99)) < 2e-2


# ------------------------------------------------------------------ GPU: HIP path
@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_frontend_vs_reference_golden(golden, dev):
    from cine_hip import frontend as FE
    g = golden("frontend")
    raw = torch.from_numpy(g["raw"]).to(dev)
    k, filt = FE.prepare_slice(raw, tuple(int(v) for v in g["crop_shape"]), int(g["n_slices"]), tuple(float(v) for v in g["filter_size"]))
    assert rel_err(filt.cpu(), _cplx(g["images_filter"])) < 1e-5
    assert rel_err(k.cpu(), _cplx(g["kspace"])) < 1e-5
    tgt = FE.combine_target(_cplx(g["images_filter"]).to(dev), _cplx(g["sens"]).to(dev), tuple(int(v) for v in g["crop_target"]))
    assert tuple(tgt.shape) == g["target"].shape and rel_err(tgt.cpu(), g["target"]) < 1e-5
    # crop alone and the error behaviour of transforms.py:206-207
    images = _cplx(np.fft.fftshift(np.fft.ifftn(np.fft.ifftshift(g["raw"].transpose(0, 3, 1, 2) * 1e6, axes=(-2, -1)), axes=(-2, -1), norm="ortho"),
                                   axes=(-2, -1)))
    crop, _ = FE.filtered_crop_center_and_slices(images.to(dev), tuple(int(v) for v in g["crop_shape"]), int(g["n_slices"]), (0.7, 0, 0.3, 0.3))
    assert rel_err(crop.cpu(), _cplx(g["images_cropped"])) < 1e-5
    with pytest.raises(ValueError, match="Invalid shapes"):
        FE.crop_select(images.to(dev), 5, (400, 20))


@pytest.mark.gpu
def test_gaussian_filter_vs_scipy_shapes(dev):
    """radius larger than the axis (reflect wraps more than once), sigma 0 axes, wide sigma."""
    from scipy.ndimage import gaussian_filter
    from cine_hip import frontend as FE
    rs = np.random.RandomState(5)
    for shape, sig in (((2, 3, 5, 4), (0.7, 0.0, 0.3, 0.3)), ((15, 2, 40, 36), (0.7, 0.0, 0.3, 0.3)), ((4, 1, 9, 33), (2.0, 0.0, 1.1, 3.9))):
        x = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
        want = gaussian_filter(x.real, sigma=sig) + 1j * gaussian_filter(x.imag, sigma=sig)
        got = FE.gaussian_filter(_cplx(x).to(dev), sig)
        assert rel_err(got.cpu(), _cplx(want)) < 1e-6, (shape, sig)


@pytest.mark.gpu
@pytest.mark.parametrize("c,n,r", [(6, 64, 24), (3, 48, 16)])
def test_espirit_vs_oracle(dev, c, n, r):
    from cine_hip import frontend as FE
    from oracle import frontend_ref as F
    k, sn, sup = _phantom(5, c, n, seed=1)
    kavg = k.mean(0)
    want, lam_w, lam2 = F.espirit_maps(kavg, r=r, with_second=True)
    got, lam_g = FE.espirit_maps(_cplx(kavg).to(dev), r=r)
    got = torch.view_as_complex(got.cpu()).numpy()
    lam_g = lam_g.cpu().numpy()
    inside = sup & (lam_w >= 0.9) & (lam2 < 0.9 * lam_w)  # where the dominant eigenvalue is separated (the power iteration
                                                           # converges like (lam2 / lam)^iters; the oracle diagonalises exactly)
    assert inside.sum() > 500
    assert np.abs(lam_g - lam_w)[inside].max() < 1e-3
    assert np.abs(got - want)[:, inside].max() < 5e-3
    assert np.sqrt((np.abs(got - want)[:, inside] ** 2).mean()) < 5e-4
    # the crop decision agrees except within rounding of the threshold
    flip = (lam_g >= 0.8) != (lam_w >= 0.8)
    assert np.all(np.abs(lam_w[flip] - 0.8) < 5e-3)


@pytest.mark.gpu
def test_espirit_full_size_recovers_analytic_maps(dev):
    """cfg-2 shape (15 coils, 200 x 200, 15 frames): eigenvalue 1 on the object, maps equal to the analytic ones in
    ESPIRiT's gauge, and the ecalib-convention wrapper."""
    from cine_hip import frontend as FE
    k, sn, sup = _phantom(15, 15, 200, seed=2)
    kavg = k.mean(0)
    maps, lam = FE.espirit_maps(_cplx(kavg).to(dev), r=24)
    maps = torch.view_as_complex(maps.cpu()).numpy()
    lam = lam.cpu().numpy()
    sup = sup & (lam >= 0.8)
    assert sup.sum() > 5000
    assert abs(float(lam[sup].mean()) - 1.0) < 2e-3
    err = np.abs(maps - sn)[:, sup]
    assert float(np.sqrt((err ** 2).mean())) < 1e-2 and float(np.quantile(err, 0.99)) < 5e-2
    calib = FE.ecalib(kavg.transpose(1, 2, 0)[None], r=24)
    assert isinstance(calib, np.ndarray) and calib.shape == (200, 200, 15)
    assert np.abs(calib.transpose(2, 0, 1) - maps).max() < 1e-6


@pytest.mark.gpu
def test_cinenet_on_espirit_maps(dev):
    """The reference's CineNet input preparation (data/transforms.py:395-440: time-averaged masked k-space -> ecalib -r 15 ->
    coils_maps) with the device calibration, then one CineNet forward on those maps: finite, and close to the forward on the
    analytic maps where both sets of maps exist."""
    import reconstruction.models as M
    from cine_hip import frontend as FE, synth
    ex = synth.make_cine_slice(5, 6, 64, 64, accel=4, center_lines=12, seed=4)
    mk = ex["masked_kspace"].to(dev)
    kavg = mk[0].mean(0)                                                     # (c, h, w, 2)
    maps, lam = FE.espirit_maps(kavg, r=12)
    net = M.CineNet(2, 3, 4, 2, "XF").eval(); synth.fill_parameters_(net, 5)
    net = net.to(dev)
    out_e = net(mk, ex["mask"].to(dev), maps[None, None])
    assert torch.isfinite(out_e).all() and float(out_e.abs().max()) > 0
    # ESPIRiT's gauge is unit norm with coil 0 real; the magnitude output does not depend on that per-pixel phase
    s = torch.view_as_complex(ex["sens_maps"][0, 0].contiguous())
    sn = s / s.abs().pow(2).sum(0, keepdim=True).sqrt()
    sn = sn * torch.exp(-1j * torch.angle(sn[:1]))
    out_a = net(mk, ex["mask"].to(dev), torch.view_as_real(sn)[None, None].contiguous().to(dev))
    keep = (lam >= 0.8).cpu() & (ex["target"][0].mean(0) > 0.1 * ex["target"].max())
    d = (out_e - out_a).abs().cpu()[0][:, keep]
    assert keep.sum() > 500 and float(d.mean()) < 0.05 * float(out_a.abs().max())


@pytest.mark.gpu
def test_prepare_example_vs_reference_golden(golden, dev):
    """cine_hip.frontend.prepare_example = SliceDataset.__getitem__ (reference data/mri_data.py:267-311) in one piece, fed through
    an h5py-like mapping: k-space of the filtered crop and the coil-combined target against the reference's arrays, with the
    stored maps standing in for `bart ecalib` (frontend.npz).  With sens=None the maps come from the build's ESPIRiT -- PARITY
    UNPINNED (BART is not in the reference tree nor in this image): only shapes and finiteness are asserted for that leg."""
    from cine_hip import frontend as FE
    g = golden("frontend")
    hf = {"y": g["raw"], "mask": np.arange(5)}
    kw = dict(crop_shape=tuple(int(v) for v in g["crop_shape"]), crop_target=tuple(int(v) for v in g["crop_target"]),
              n_slices=int(g["n_slices"]), filter_size=tuple(float(v) for v in g["filter_size"]))
    k, mask, target, attrs, fname, dataslice = FE.prepare_example(hf, sens=g["sens"], fname="slice_0.h5", **kw)
    assert k.dtype == np.complex64 and k.shape == g["kspace"].shape and rel_err(_cplx(k), _cplx(g["kspace"])) < 1e-5
    assert target.dtype == np.float32 and rel_err(target, g["target"]) < 1e-5
    assert (mask == np.arange(5)).all() and attrs == {} and fname == "slice_0.h5" and dataslice == 0
    k2, _, target2, *_ = FE.prepare_example(g["raw"], ecalib_r=16, **kw)                   # the ESPIRiT leg (unpinned)
    assert rel_err(_cplx(k2), _cplx(g["kspace"])) < 1e-5 and target2.shape == g["target"].shape and np.isfinite(target2).all()


@pytest.mark.gpu
def test_prepare_slice_odd_crop_uses_the_reference_shift_order(dev):
    """mri_data.py:291 returns to k-space with ifftshift(fftn(fftshift(x))), which differs from fft2c for odd sizes."""
    from cine_hip import frontend as FE
    rs = np.random.RandomState(3)
    raw = (rs.standard_normal((4, 30, 28, 2)) + 1j * rs.standard_normal((4, 30, 28, 2))).astype(np.complex64)
    k, filt = FE.prepare_slice(torch.from_numpy(raw).to(dev), (21, 17), 3, (0.7, 0.0, 0.3, 0.3))
    f = torch.view_as_complex(filt.cpu().contiguous()).numpy()
    want = np.fft.ifftshift(np.fft.fftn(np.fft.fftshift(f, axes=(-2, -1)), axes=(-2, -1), norm=None), axes=(-2, -1)) / np.sqrt(21 * 17)
    assert rel_err(k.cpu(), _cplx(want.astype(np.complex64))) < 1e-5
