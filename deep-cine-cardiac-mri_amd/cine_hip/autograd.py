"""Training through the HIP path: ``torch.autograd.Function`` wrappers whose backward passes run hand-written kernels.

What the reference differentiates with ATen autograd in ``pl_modules/varnet_module.py:97-113`` (``training_step``:
forward + ``SSIMLoss``) is differentiated here by the gradient entry points of ``include/cine_hip.h`` ("Training"):

  UnetFn       : denoisers/unet.py:73-125 -- conv3x3 / transpose-conv / 1x1 input gradients (the forward MFMA kernel on
                 re-packed weights), weight gradients (MFMA GEMM over pixels x samples), InstanceNorm + LeakyReLU backward
  NormUnetFn   : denoisers/norm_unet.py:98-114 around it (group norm with unbiased std, pad, un-norm)
  XfyfFn       : models/varnet.py:196-241 (temporal mean, centered temporal DFT, x-f / y-f rotations, both NormUnets)
  ImageDcFn    : models/varnet.py:181-194, 281-282 on the coil-combined image (self-adjoint in the image; gradients for the
                 sensitivity maps, the zero-filled term and lambda_reg)
  CoilReduceFn : models/varnet.py:187-194 with respect to the maps
  RssNormFn    : models/varnet.py:58-59;  AbsFn: utils/math.py:48-62
  MwcnnFn      : denoisers/mwcnn.py:135-179 (Haar DWT / IWT adjoints, additive skips);  XpdRegFn: models/xpdnet.py:424-509 (XPDNet's I-step
                 network: buffer pack with its own temporal transform, both MWCNNs, unpack);  ImageDcFixedFn: xpdnet.py:128-167
  ConjGradFn   : models/cinenet.py:136-171 (the adjoint recurrence of the CG iteration with the recorded step sizes),
                 AxpbyLamFn: cinenet.py:255-257;  SsimLossFn: utils/losses.py:25-58

Gradient convention for complex tensors: the trailing (re, im) pair carries (dL/dre, dL/dim), as torch does for real
views.  Every reduction in the kernels runs in a fixed order: repeated backward passes are bit-identical.
"""
import ctypes
from typing import Optional, Sequence

import torch
from torch.autograd import Function

from . import ops
from ._lib import CineHipError, check, lib

_p = ops._p
_stream = ops._stream

def _use_side_stream(device: torch.device) -> None:
    """Hand a second stream (``ops.side_streams``: one per (device, current stream), created by the binding -- the library creates none) to
    the backward entry points that run weight gradients beside the input-gradient chain (cine_set_side_stream).  CINE_SIDE_STREAM=0 in
    the environment of THIS binding keeps the whole backward pass on the caller's stream."""
    import os
    if os.environ.get("CINE_SIDE_STREAM", "1") == "0":
        check(lib().cine_set_side_stream(None), "cine_set_side_stream")
        return
    check(lib().cine_set_side_stream(ops.side_streams(device, 1)[0].cuda_stream), "cine_set_side_stream")


def _masked(cls):
    """Class decorator of every Function below: the diagnostic kernel-selection mask is a setting of the CALLING THREAD
    (cine_set_conv_plane), and torch runs ``backward`` on the autograd engine's device thread -- so the forward records the caller's mask
    (``ops.conv_plane_mask()``) and the backward applies it to whichever thread it runs on before it launches anything."""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args):
        ctx.cine_plane_mask = ops.conv_plane_mask()
        return fwd(ctx, *args)

    def backward(ctx, *grads):
        ops.apply_conv_plane(getattr(ctx, "cine_plane_mask", 7))
        try:
            return bwd(ctx, *grads)
        finally:
            ops.apply_conv_plane(ops.conv_plane_mask())      # back to this thread's own setting

    forward.__doc__, backward.__doc__ = fwd.__doc__, bwd.__doc__
    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


def _c(x: torch.Tensor) -> torch.Tensor:
    return x if x.is_contiguous() else x.contiguous()


# ------------------------------------------------------------------ U-Net
def _unet_call_shapes(x: torch.Tensor, weights: "ops.UnetWeights"):
    n, cin, h, w = x.shape
    if cin != weights.in_ch:
        raise ValueError(f"unet input has {cin} channels, expected {weights.in_ch}")
    return n, cin, h, w


def unet2d_forward_train(x: torch.Tensor, weights: "ops.UnetWeights"):
    """cine_unet2d_forward_train: returns (y, workspace with every raw layer output)."""
    x = ops._dev(x, "unet input")
    n, cin, h, w = _unet_call_shapes(x, weights)
    key = weights.training_key()                     # raises for Dropout in training mode
    need = lib().cine_unet2d_train_ws_bytes(n, h, w, cin, weights.out_ch, weights.chans, weights.pools)
    if need == 0:
        raise CineHipError("cine_unet2d_train_ws_bytes rejected the shape")
    ws = torch.empty(need, device=x.device, dtype=torch.uint8)
    y = torch.empty((n, weights.out_ch, h, w), device=x.device, dtype=x.dtype)
    nb = ops._branch_count(n, len(weights.unets))
    drop = weights.dropout_multipliers(n, x.device)          # Dropout2d of the ConvBlocks (training mode, drop_prob > 0), else None
    if nb > 1 or drop is not None:          # the same launch sequence, as concurrent branches into the same workspace layout (bit-identical) and / or with dropout
        side = ops.side_streams(x.device, nb - 1) if nb > 1 else []
        sarr = (ctypes.c_void_p * max(len(side), 1))(*[s_.cuda_stream for s_ in side])
        check(lib().cine_unet2d_forward_branches(x.data_ptr(), y.data_ptr(), weights.pointers(train=True), len(weights.unets), n, h, w, cin,
                                                 weights.out_ch, weights.chans, weights.pools, ops.lrelu_slope(), ws.data_ptr(), ws.numel(), _stream(),
                                                 sarr, len(side), 1, _p(drop)), "cine_unet2d_forward_branches")
    else:
        check(lib().cine_unet2d_forward_train(x.data_ptr(), y.data_ptr(), weights.pointers(train=True), len(weights.unets), n, h, w, cin,
                                              weights.out_ch, weights.chans, weights.pools, ops.lrelu_slope(), ws.data_ptr(), ws.numel(), _stream()),
              "cine_unet2d_forward_train")
    ws.cine_drop = drop                              # the backward pass needs the same multipliers
    ws.cine_slope = ops.lrelu_slope()                  # the backward pass differentiates the activation the forward pass applied
    ws.cine_training_key = key                       # checked by unet2d_backward
    return y, ws


def _zero_grads(plists, device):
    """Zeroed gradient buffers shaped like the parameters of every list, carved out of ONE zero fill (the C entry points accumulate)."""
    sizes = [(p.numel() + 3) // 4 * 4 for pl in plists for p in pl]          # 16-byte aligned pieces
    flat = torch.zeros(sum(sizes), device=device, dtype=torch.float32)
    out, off, k = [], 0, 0
    for pl in plists:
        gl = []
        for p in pl:
            gl.append(flat[off:off + p.numel()].view(p.shape)); off += sizes[k]; k += 1
        out.append(gl)
    return out


def unet2d_backward(x: torch.Tensor, gy: torch.Tensor, weights: "ops.UnetWeights", fwd_ws: torch.Tensor, need_gx: bool):
    """cine_unet2d_backward: returns (gx | None, [per-set list of parameter gradients in ``weights.param_list()`` order])."""
    x = ops._dev(x, "unet input"); gy = ops._dev(gy, "unet output gradient")
    n, cin, h, w = _unet_call_shapes(x, weights)
    nsets = len(weights.unets)
    need = lib().cine_unet2d_backward_ws_bytes(n, h, w, cin, weights.out_ch, weights.chans, weights.pools)
    ws = torch.empty(need, device=x.device, dtype=torch.uint8)
    plists = weights.param_lists()
    grads = _zero_grads(plists, x.device)
    gptr = (ctypes.c_void_p * (nsets * len(plists[0])))(*[g.data_ptr() for gl in grads for g in gl])
    gx = torch.empty_like(x) if need_gx else None
    weights.check_training_key(getattr(fwd_ws, "cine_training_key", None), "cine_unet2d_backward")
    _use_side_stream(x.device)
    check(lib().cine_unet2d_backward_drop(x.data_ptr(), gy.data_ptr(), weights.dgrad_pointers(), gptr, nsets, n, h, w, cin,
                                          weights.out_ch, weights.chans, weights.pools, getattr(fwd_ws, "cine_slope", ops.lrelu_slope()), fwd_ws.data_ptr(), fwd_ws.numel(),
                                          ws.data_ptr(), ws.numel(), _p(gx), _p(getattr(fwd_ws, "cine_drop", None)), _stream()), "cine_unet2d_backward")
    return gx, grads


def _param_grads(weights: "ops.UnetWeights", grads, params: Sequence[torch.Tensor]):
    """Map the per-set gradient lists onto the flat ``params`` tuple a Function received (one entry per DISTINCT parameter:
    with weight sharing both sets of a launch belong to the same tensors and are summed)."""
    out = {}
    for pl, gl in zip(weights.param_lists(), grads):
        for p, g in zip(pl, gl):
            out[id(p)] = g if id(p) not in out else out[id(p)] + g
    return tuple(out.get(id(p)) for p in params)


@_masked
class UnetFn(Function):
    """y = Unet(x) on (n, in_ch, h, w) planes; ``params`` = the distinct parameters of ``weights`` (for autograd's bookkeeping)."""

    @staticmethod
    def forward(ctx, x, weights, *params):
        y, ws = unet2d_forward_train(x, weights)
        ctx.weights, ctx.ws, ctx.params = weights, ws, params
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gx, grads = unet2d_backward(x, _c(gy), ctx.weights, ctx.ws, ctx.needs_input_grad[0])
        return (gx, None) + _param_grads(ctx.weights, grads, ctx.params)


def unet2d(x: torch.Tensor, weights: "ops.UnetWeights") -> torch.Tensor:
    return UnetFn.apply(x, weights, *weights.distinct_params())


# ------------------------------------------------------------------ NormUnet (2-D)
@_masked
class NormUnetFn(Function):
    """NormUnet.forward (norm_unet.py:98-114) on x (n, h, w, 2)."""

    @staticmethod
    def forward(ctx, x, weights, norm, *params):
        x = ops._dev(x, "normunet input")
        n, h, w, _ = x.shape
        planes, stats = ops.normunet_pack(x, norm=norm)          # norm False: the plain repack around CineNet's bare Unet (cinenet.py:242-244)
        q, ws = unet2d_forward_train(planes, weights)
        y = ops.normunet_unpack(q, stats, h, w)
        ctx.weights, ctx.ws, ctx.params, ctx.hw, ctx.norm = weights, ws, params, (h, w), norm
        ctx.save_for_backward(planes, q, *([stats] if norm else []))
        return y

    @staticmethod
    def backward(ctx, gy):
        planes, q = ctx.saved_tensors[:2]
        stats = ctx.saved_tensors[2] if ctx.norm else None
        h, w = ctx.hw
        n = planes.shape[0]
        gy = ops._dev(_c(gy), "normunet output gradient")
        gq = torch.empty_like(q)
        dstats = torch.empty_like(stats) if ctx.norm else None
        check(lib().cine_normunet_unpack_bwd(gy.data_ptr(), q.data_ptr(), _p(stats), gq.data_ptr(), _p(dstats), n, h, w, _stream()),
              "cine_normunet_unpack_bwd")
        need_gx = ctx.needs_input_grad[0]
        gp, grads = unet2d_backward(planes, gq, ctx.weights, ctx.ws, need_gx)
        gx = None
        if need_gx:
            gx = torch.empty((n, h, w, 2), device=gy.device, dtype=gy.dtype)
            check(lib().cine_normunet_pack_bwd(gp.data_ptr(), planes.data_ptr(), _p(stats), _p(dstats), gx.data_ptr(), n, h, w, _stream()),
                  "cine_normunet_pack_bwd")
        return (gx, None, None) + _param_grads(ctx.weights, grads, ctx.params)


def norm_unet(x: torch.Tensor, weights: "ops.UnetWeights", norm: bool = True) -> torch.Tensor:
    return NormUnetFn.apply(x, weights, norm, *weights.distinct_params())


# ------------------------------------------------------------------ x-f / y-f regulariser (both NormUnets)
@_masked
class XfyfFn(Function):
    """VarNetBlock.xfyf_transform (varnet.py:196-241) on image (b, t, h, w, 2) -> (b, t, 1, h, w, 2).
    ``wboth`` holds both U-Nets (x-f first); ``wx`` / ``wy`` the single ones for plane sets of different shapes."""

    @staticmethod
    def forward(ctx, image, xf, norm, wboth, wx, wy, *params):
        image = ops._dev(image, "image")
        b, t, h, w, _ = image.shape
        pxf, pyf, sxf, syf, mean = ops.xfyf_pack(image, xf, norm=norm)     # norm False: CineNet's bare U-Nets (cinenet.py:181-219)
        joint = pxf.shape == pyf.shape and pxf.data_ptr() + pxf.numel() * 4 == pyf.data_ptr()
        if joint:
            planes = torch.as_strided(pxf, (2 * pxf.shape[0],) + tuple(pxf.shape[1:]), pxf.stride())
            q, ws = unet2d_forward_train(planes, wboth)
            oxf, oyf = q[:pxf.shape[0]], q[pxf.shape[0]:]
            ctx.ws = (ws,)
        else:
            oxf, ws0 = unet2d_forward_train(pxf, wx)
            oyf, ws1 = unet2d_forward_train(pyf, wy)
            ctx.ws = (ws0, ws1)
        out = ops.xfyf_unpack(oxf, oyf, sxf, syf, mean, b, t, h, w, xf)
        ctx.cfg = (b, t, h, w, bool(xf), joint, bool(norm))
        ctx.weights, ctx.params = (wboth, wx, wy), params
        ctx.save_for_backward(pxf, pyf, oxf, oyf, *([sxf, syf] if norm else []))
        return out

    @staticmethod
    def backward(ctx, gout):
        pxf, pyf, oxf, oyf = ctx.saved_tensors[:4]
        b, t, h, w, xf, joint, norm = ctx.cfg
        sxf, syf = (ctx.saved_tensors[4], ctx.saved_tensors[5]) if norm else (None, None)
        wboth, wx, wy = ctx.weights
        gout = ops._dev(_c(gout), "xfyf output gradient")
        dev, dt = gout.device, gout.dtype
        nbytes = lib().cine_xfyf_bwd_ws_bytes(b, t, h, w)
        wsb = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        if joint:
            gq = torch.empty((pxf.shape[0] + pyf.shape[0],) + tuple(pxf.shape[1:]), device=dev, dtype=dt)
            gqx, gqy = gq[:pxf.shape[0]], gq[pxf.shape[0]:]
        else:
            gqx, gqy = torch.empty_like(oxf), torch.empty_like(oyf)
        dsx, dsy = (torch.empty_like(sxf), torch.empty_like(syf)) if norm else (None, None)
        gmean = torch.empty((b, h, w, 2), device=dev, dtype=dt)
        check(lib().cine_xfyf_unpack_bwd(gout.data_ptr(), oxf.data_ptr(), oyf.data_ptr(), _p(sxf), _p(syf),
                                         gqx.data_ptr(), gqy.data_ptr(), _p(dsx), _p(dsy), gmean.data_ptr(),
                                         b, t, h, w, int(xf), wsb.data_ptr(), nbytes, _stream()), "cine_xfyf_unpack_bwd")
        if joint:
            planes = torch.as_strided(pxf, (2 * pxf.shape[0],) + tuple(pxf.shape[1:]), pxf.stride())
            gp, grads = unet2d_backward(planes, gq, wboth, ctx.ws[0], True)
            gpx, gpy = gp[:pxf.shape[0]], gp[pxf.shape[0]:]
            pg = _param_grads(wboth, grads, ctx.params)
        else:
            gpx, g0 = unet2d_backward(pxf, gqx, wx, ctx.ws[0], True)
            gpy, g1 = unet2d_backward(pyf, gqy, wy, ctx.ws[1], True)
            a, c = _param_grads(wx, g0, ctx.params), _param_grads(wy, g1, ctx.params)
            pg = tuple(u if v is None else (v if u is None else u + v) for u, v in zip(a, c))
        gimg = None
        if ctx.needs_input_grad[0]:
            gimg = torch.empty((b, t, h, w, 2), device=dev, dtype=dt)
            check(lib().cine_xfyf_pack_bwd(gpx.data_ptr(), gpy.data_ptr(), pxf.data_ptr(), pyf.data_ptr(), _p(sxf), _p(syf),
                                           _p(dsx), _p(dsy), gmean.data_ptr(), gimg.data_ptr(), b, t, h, w, int(xf),
                                           wsb.data_ptr(), nbytes, _stream()), "cine_xfyf_pack_bwd")
        return (gimg, None, None, None, None, None) + pg


def xfyf(image: torch.Tensor, xf: bool, wboth, wx, wy, norm: bool = True) -> torch.Tensor:
    return XfyfFn.apply(image, xf, norm, wboth, wx, wy, *wboth.distinct_params())


# ------------------------------------------------------------------ MWCNN
@_masked
class MwcnnFn(Function):
    """MWCNN.forward (denoisers/mwcnn.py:135-179) on (n, in_ch, h, w) planes.  ``w2`` / ``split``: samples [split, n) go through a second
    network of the same topology in the same launches (XPDNet's x-t / y-t networks); ``params`` = the parameters of both, for autograd."""

    @staticmethod
    def forward(ctx, x, w, w2, split, *params):
        x = ops._dev(x, "mwcnn input")
        net = w.net
        n, cin, h, wd = x.shape
        two = w2 is not None and w2 is not w
        L = lib()
        need = L.cine_mwcnn_train_ws_bytes(n, h, wd, cin, net.out_chans, net.n_scales, w.nf, w.nc, net.first_conv_n_filters)
        ws = torch.empty(max(need, 1), device=x.device, dtype=torch.uint8)
        y = torch.empty((n, net.out_chans, h, wd), device=x.device, dtype=x.dtype)
        check(L.cine_mwcnn_forward_train(x.data_ptr(), y.data_ptr(), w.pointers(train=True), w2.pointers(train=True) if two else None, int(split) if two else n,
                                         n, h, wd, cin, net.out_chans, net.n_scales, w.nf, w.nc, net.n_first_convs, net.first_conv_n_filters,
                                         int(net.res), ops.lrelu_slope(), ws.data_ptr(), ws.numel(), _stream()), "cine_mwcnn_forward_train")
        ctx.slope = ops.lrelu_slope()
        ctx.cfg = (w, w2 if two else None, int(split) if two else n)
        ctx.keys = tuple(tuple((p.data_ptr(), p._version) for p in wt.param_list()) for wt in ((w, w2) if two else (w,)))
        ctx.ws, ctx.params = ws, params
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        w, w2, split = ctx.cfg
        net = w.net
        for wt, key in zip((w, w2) if w2 is not None else (w,), ctx.keys):
            if key != tuple((p.data_ptr(), p._version) for p in wt.param_list()):
                raise RuntimeError("cine_mwcnn_backward: an MWCNN parameter was modified between the forward and the backward pass; "
                                   "the saved activations belong to the old weights")
        gy = ops._dev(_c(gy), "mwcnn output gradient")
        n, cin, h, wd = x.shape
        L = lib()
        need = L.cine_mwcnn_backward_ws_bytes(n, h, wd, cin, net.out_chans, net.n_scales, w.nf, w.nc, net.first_conv_n_filters)
        ws = torch.empty(need, device=x.device, dtype=torch.uint8)

        def grads_of(wt):
            pl = wt.param_list()
            gl = _zero_grads([pl], x.device)[0]
            return pl, gl, (ctypes.c_void_p * len(gl))(*[g.data_ptr() for g in gl])
        p1, g1, gp1 = grads_of(w)
        p2, g2, gp2 = grads_of(w2) if w2 is not None else (None, None, None)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        _use_side_stream(x.device)
        check(L.cine_mwcnn_backward(x.data_ptr(), gy.data_ptr(), w.dgrad_pointers(), w2.dgrad_pointers() if w2 is not None else None, gp1, gp2,
                                    split, n, h, wd, cin, net.out_chans, net.n_scales, w.nf, w.nc, net.first_conv_n_filters, ctx.slope,
                                    ctx.ws.data_ptr(), ctx.ws.numel(), ws.data_ptr(), ws.numel(), _p(gx), _stream()), "cine_mwcnn_backward")
        out = {}
        for pl, gl in ((p1, g1),) + (((p2, g2),) if w2 is not None else ()):
            for p, g in zip(pl, gl):
                out[id(p)] = g if id(p) not in out else out[id(p)] + g
        return (gx, None, None, None) + tuple(out.get(id(p)) for p in ctx.params)


def mwcnn(x: torch.Tensor, w, w2=None, split: int = 0) -> torch.Tensor:
    params, seen = [], set()
    for wt in (w,) + ((w2,) if w2 is not None and w2 is not w else ()):
        for p in wt.param_list():
            if id(p) not in seen:
                seen.add(id(p)); params.append(p)
    return MwcnnFn.apply(x, w, w2, split, *params)


# ------------------------------------------------------------------ coil operators
@_masked
class ImageDcFn(Function):
    """cine_image_dc with the soft-DC weights of softplus(lambda) (varnet.py:181-194, 281-282):
    out = sum_c conj(S_c) T(S_c m) + v / (1 + v) zf,  T = IFFT_h [mask ? 1 / (1 + v) : 1] FFT_h."""

    @staticmethod
    def forward(ctx, m, sens, zf, mask, lam):
        out = ops.image_dc(m, sens, zf, mask, lam)
        ctx.save_for_backward(m, sens, zf, mask, lam)
        return out

    @staticmethod
    def backward(ctx, gout):
        m, sens, zf, mask, lam = ctx.saved_tensors
        gout = ops._dev(_c(gout), "image_dc output gradient")
        m = ops._dev(m, "image")
        b, _, c, h, w, _ = sens.shape
        t = m.shape[1]
        need = ctx.needs_input_grad
        gm = ops.image_dc(gout, sens, None, mask, lam).view(m.shape) if need[0] else None      # T is Hermitian
        gs = None
        if need[1]:
            part = torch.empty((b, t, c, h, w, 2), device=m.device, dtype=m.dtype)
            check(lib().cine_image_dc_sens_grad(m.data_ptr(), gout.data_ptr(), ops._dev(sens, "sens_maps").data_ptr(), mask.data_ptr(),
                                                lam.detach().data_ptr(), 0.0, 0.0, part.data_ptr(), b, t, c, h, w, _stream()),
                  "cine_image_dc_sens_grad")
            gs = coil_accum(None, part)
        gzf = None
        if need[2]:
            gzf = torch.empty_like(zf)
            check(lib().cine_axpby_lam(gzf.data_ptr(), None, gout.data_ptr(), gout.numel(), lam.detach().data_ptr(), 1, 1.0, _stream()),
                  "cine_axpby_lam")
        glam = None
        if need[4]:
            # d out / d v = (zf - A^H M A m) / (1 + v)^2;  v = softplus(lambda), dv / dlambda = sigmoid(lambda)
            d = ops.image_dc(m, sens, zf, mask, None, weights=(-1.0, 0.0, 1.0))
            dot = ops.dot(gout, d)
            lv = lam.detach()
            v = torch.nn.functional.softplus(lv)
            glam = (dot * torch.sigmoid(lv) / ((1 + v) * (1 + v))).view(lam.shape)
        return gm, gs, gzf, None, glam


@_masked
class ImageDcFixedFn(Function):
    """cine_image_dc with fixed weights: sum_c conj(S_c) IFFT_h[(mask ? w1 : w0) FFT_h(S_c m)] + beta zf -- XPDNet's K step + masked backward
    operator A^H M (A x - k_ref) is (1, 0, -1) (xpdnet.py:128-131, 161-167, 295-298)."""

    @staticmethod
    def forward(ctx, m, sens, zf, mask, w1, w0, beta):
        out = ops.image_dc(m, sens, zf, mask, None, weights=(w1, w0, beta))
        ctx.save_for_backward(m, sens, mask)
        ctx.wts = (float(w1), float(w0), float(beta))
        return out

    @staticmethod
    def backward(ctx, gout):
        m, sens, mask = ctx.saved_tensors
        w1, w0, beta = ctx.wts
        gout = ops._dev(_c(gout), "image_dc output gradient")
        m = ops._dev(m, "image")
        b, _, c, h, w, _ = sens.shape
        t = m.shape[1]
        need = ctx.needs_input_grad
        gm = ops.image_dc(gout, sens, None, mask, None, weights=(w1, w0, 0.0)).view(m.shape) if need[0] else None
        gs = None
        if need[1]:
            part = torch.empty((b, t, c, h, w, 2), device=m.device, dtype=m.dtype)
            check(lib().cine_image_dc_sens_grad(m.data_ptr(), gout.data_ptr(), ops._dev(sens, "sens_maps").data_ptr(), mask.data_ptr(), None,
                                                w1, w0, part.data_ptr(), b, t, c, h, w, _stream()), "cine_image_dc_sens_grad")
            gs = coil_accum(None, part)
        gzf = (gout * beta) if need[2] else None
        return gm, gs, gzf, None, None, None, None


@_masked
class XpdRegFn(Function):
    """XPDNetBlock's I-step network (xpdnet.py:424-509) for the XT / XF dynamic types: buffer (b, t, 1, h, w, 2n) + backward-operator image
    -> new buffer, through cine_xpd_pack, the two MWCNNs (one launch sequence when the plane sets have one shape) and cine_xpd_unpack."""

    @staticmethod
    def forward(ctx, buf, extra, n, n_scales, xf, wx, wy, *params):
        buf = ops._dev(buf, "image buffer"); extra = ops._dev(extra, "backward-op image")
        b, t, _, h, w, _ = buf.shape
        pxf, pyf, mean = ops.xpd_pack(buf, extra, n, n_scales, xf)
        joint = pxf.shape[1:] == pyf.shape[1:] and pxf.data_ptr() + pxf.numel() * 4 == pyf.data_ptr()
        L = lib()

        def run(planes, w1, w2, split):
            net = w1.net
            nn_, cin, hh, ww = planes.shape
            need = L.cine_mwcnn_train_ws_bytes(nn_, hh, ww, cin, net.out_chans, net.n_scales, w1.nf, w1.nc, net.first_conv_n_filters)
            ws = torch.empty(max(need, 1), device=planes.device, dtype=torch.uint8)
            y = torch.empty((nn_, net.out_chans, hh, ww), device=planes.device, dtype=planes.dtype)
            two = w2 is not None and w2 is not w1
            check(L.cine_mwcnn_forward_train(planes.data_ptr(), y.data_ptr(), w1.pointers(train=True), w2.pointers(train=True) if two else None, split if two else nn_,
                                             nn_, hh, ww, cin, net.out_chans, net.n_scales, w1.nf, w1.nc, net.n_first_convs,
                                             net.first_conv_n_filters, int(net.res), ops.lrelu_slope(), ws.data_ptr(), ws.numel(), _stream()), "cine_mwcnn_forward_train")
            ws.cine_slope = ops.lrelu_slope()
            return y, ws
        if joint:
            planes = torch.as_strided(pxf, (pxf.shape[0] + pyf.shape[0],) + tuple(pxf.shape[1:]), pxf.stride())
            q, ws = run(planes, wx, wy, pxf.shape[0])
            oxf, oyf = q[:pxf.shape[0]], q[pxf.shape[0]:]
            ctx.ws = (ws,)
        else:
            oxf, ws0 = run(pxf, wx, None, 0)
            oyf, ws1 = run(pyf, wy, None, 0)
            ctx.ws = (ws0, ws1)
        out = ops.xpd_unpack(oxf, oyf, mean, b, t, h, w, n, n_scales, xf)
        ctx.cfg = (b, t, h, w, int(n), int(n_scales), bool(xf), joint)
        ctx.weights, ctx.params = (wx, wy), params
        ctx.save_for_backward(pxf, pyf)
        return out

    @staticmethod
    def backward(ctx, gout):
        pxf, pyf = ctx.saved_tensors
        b, t, h, w, n, n_scales, xf, joint = ctx.cfg
        wx, wy = ctx.weights
        gout = ops._dev(_c(gout), "I-step output gradient")
        dev, dt = gout.device, gout.dtype
        L = lib()
        oc = 2 * n
        if joint:
            gq = torch.empty((pxf.shape[0] + pyf.shape[0], oc) + tuple(pxf.shape[2:]), device=dev, dtype=dt)
            gqx, gqy = gq[:pxf.shape[0]], gq[pxf.shape[0]:]
        else:
            gqx = torch.empty((pxf.shape[0], oc) + tuple(pxf.shape[2:]), device=dev, dtype=dt)
            gqy = torch.empty((pyf.shape[0], oc) + tuple(pyf.shape[2:]), device=dev, dtype=dt)
        gmean = torch.empty((b, h, w, n + 1, 2), device=dev, dtype=dt)
        check(L.cine_xpd_unpack_bwd(gout.data_ptr(), gqx.data_ptr(), gqy.data_ptr(), gmean.data_ptr(), b, t, h, w, n, n_scales, int(xf), _stream()),
              "cine_xpd_unpack_bwd")
        out = {}

        def back(planes, gy, w1, w2, split, fws):
            net = w1.net
            nn_, cin, hh, ww = planes.shape
            need = L.cine_mwcnn_backward_ws_bytes(nn_, hh, ww, cin, net.out_chans, net.n_scales, w1.nf, w1.nc, net.first_conv_n_filters)
            ws = torch.empty(need, device=dev, dtype=torch.uint8)
            two = w2 is not None and w2 is not w1
            lists = []
            pls = [wt.param_list() for wt in (w1,) + ((w2,) if two else ())]
            for pl, gl in zip(pls, _zero_grads(pls, dev)):
                lists.append((pl, gl, (ctypes.c_void_p * len(gl))(*[g.data_ptr() for g in gl])))
            gx = torch.empty_like(planes)
            check(L.cine_mwcnn_backward(planes.data_ptr(), gy.data_ptr(), w1.dgrad_pointers(), w2.dgrad_pointers() if two else None, lists[0][2],
                                        lists[1][2] if two else None, split if two else nn_, nn_, hh, ww, cin, net.out_chans, net.n_scales, w1.nf, w1.nc,
                                        net.first_conv_n_filters, getattr(fws, "cine_slope", ops.lrelu_slope()), fws.data_ptr(), fws.numel(), ws.data_ptr(), ws.numel(), gx.data_ptr(), _stream()),
                  "cine_mwcnn_backward")
            for pl, gl, _ in lists:
                for p, g in zip(pl, gl):
                    out[id(p)] = g if id(p) not in out else out[id(p)] + g
            return gx
        if joint:
            planes = torch.as_strided(pxf, (pxf.shape[0] + pyf.shape[0],) + tuple(pxf.shape[1:]), pxf.stride())
            gp = back(planes, gq, wx, wy, pxf.shape[0], ctx.ws[0])
            gpx, gpy = gp[:pxf.shape[0]], gp[pxf.shape[0]:]
        else:
            gpx = back(pxf, gqx, wx, None, 0, ctx.ws[0])
            gpy = back(pyf, gqy, wy, None, 0, ctx.ws[1])
        gbuf = torch.empty((b, t, 1, h, w, 2 * n), device=dev, dtype=dt)
        gextra = torch.empty((b, t, 1, h, w, 2), device=dev, dtype=dt)
        check(L.cine_xpd_pack_bwd(gpx.data_ptr(), gpy.data_ptr(), gmean.data_ptr(), gbuf.data_ptr(), gextra.data_ptr(), b, t, h, w, n, n_scales,
                                  int(xf), _stream()), "cine_xpd_pack_bwd")
        return (gbuf, gextra, None, None, None, None, None) + tuple(out.get(id(p)) for p in ctx.params)


def xpd_regularise(buf, extra, n, n_scales, xf, wx, wy):
    params, seen = [], set()
    for wt in (wx, wy):
        for p in wt.param_list():
            if id(p) not in seen:
                seen.add(id(p)); params.append(p)
    return XpdRegFn.apply(buf, extra, n, n_scales, xf, wx, wy, *params)


# ---- 3-D U-Net (denoisers/unet.py with dims = 3) -----------------------------------------------------------------------------------------
@_masked
class Unet3dFn(Function):
    """y = Unet(x), dims = 3, on (n, in_ch, d, h, w): cine_unet3d_forward_train keeps every raw layer output with its merged InstanceNorm record,
    cine_unet3d_backward walks them in reverse with the kernels of the 2-D backward pass (InstanceNorm + LeakyReLU backward on (d h, w) planes with
    the 2x2x2 pool adjoint / zero-pad crop gathered on load; input gradients on the forward 3x3x3 / 1x1x1 kernels; the 3x3x3 weight gradient as
    three 3x3 weight gradients over depth-shifted slice pairs, on the side stream)."""

    @staticmethod
    def forward(ctx, x, weights, *params):
        x = ops._dev(x, "unet3d input")
        n, cin, d, h, w = x.shape
        if len(weights.unets) != 1 or cin != weights.in_ch:
            raise ValueError("unet3d: one weight set, input channels as built")
        ctx.training_key = weights.training_key()
        L = lib()
        need = L.cine_unet3d_train_ws_bytes(n, d, h, w, cin, weights.out_ch, weights.chans, weights.pools)
        if need == 0 or min(d >> weights.pools, h >> weights.pools, w >> weights.pools) < 1:
            raise CineHipError("unet3d: volume too small for the number of pools")
        ws = torch.empty(need, device=x.device, dtype=torch.uint8)
        y = torch.empty((n, weights.out_ch, d, h, w), device=x.device, dtype=x.dtype)
        slope = ops.lrelu_slope()
        drop = weights.dropout_multipliers(n, x.device)          # Dropout3d of the ConvBlocks (training mode, drop_prob > 0), else None
        check(L.cine_unet3d_forward_train_drop(x.data_ptr(), y.data_ptr(), weights.pointers(), n, d, h, w, cin, weights.out_ch, weights.chans, weights.pools,
                                               slope, ws.data_ptr(), ws.numel(), _p(drop), _stream()), "cine_unet3d_forward_train")
        ctx.weights, ctx.ws, ctx.params, ctx.drop = weights, ws, params, drop
        ctx.slope = slope                      # the backward pass runs on an autograd thread: it differentiates what THIS call applied
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        weights = ctx.weights
        weights.check_training_key(ctx.training_key, "unet3d backward")
        gy = ops._dev(_c(gy), "unet3d output gradient")
        n, cin, d, h, w = x.shape
        L = lib()
        ws = torch.empty(L.cine_unet3d_backward_ws_bytes(n, d, h, w, cin, weights.out_ch, weights.chans, weights.pools), device=x.device, dtype=torch.uint8)
        plists = weights.param_lists()
        grads = _zero_grads(plists, x.device)
        gptr = (ctypes.c_void_p * len(plists[0]))(*[g.data_ptr() for g in grads[0]])
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        _use_side_stream(x.device)
        check(L.cine_unet3d_backward_drop(x.data_ptr(), gy.data_ptr(), weights.dgrad_pointers(), gptr, n, d, h, w, cin, weights.out_ch, weights.chans, weights.pools,
                                          ctx.slope, ctx.ws.data_ptr(), ctx.ws.numel(), ws.data_ptr(), ws.numel(), _p(gx), _p(ctx.drop), _stream()), "cine_unet3d_backward")
        return (gx, None) + _param_grads(weights, grads, ctx.params)


def unet3d(x: torch.Tensor, weights: "ops.UnetWeights") -> torch.Tensor:
    return Unet3dFn.apply(x, weights, *weights.distinct_params())


# ---- k-space networks (XPDNet's dual update, denoisers/kspace_net.py; xpdnet.py:372-403) ----------------------------------------------------
def _conv3d_dgrad(g, weight):
    """(n, cout, d, h, w) -> (n, cin, d, h, w): the forward 3x3x3 kernel on the flipped / transposed packing."""
    wp = ops._pack("c27", weight.detach().flip(2, 3, 4).transpose(0, 1).contiguous())
    n, cout, d, h, w = g.shape
    ci = weight.shape[1]
    gx = torch.empty((n, ci, d, h, w), device=g.device, dtype=g.dtype)
    check(lib().cine_conv3d_in(g.data_ptr(), None, 0, cout, 0, d, h, w, None, None, 0, 0, 0, 0, 0, 0, wp.data_ptr(), None, None, 0,
                               gx.data_ptr(), None, n, ci, d, h, w, ops.IN_EPS, ops.lrelu_slope(), _stream()), "cine_conv3d_in")
    return gx


def _conv3d_wgrad(x, g, weight, want_bias):
    """Weight (and bias) gradient of a 3x3x3 conv: one launch of the 2-D weight-gradient kernel per depth tap, depth slices as samples."""
    gwz = torch.zeros((3,) + tuple(weight.shape[:2]) + (3, 3), device=g.device, dtype=g.dtype)
    gb = torch.zeros(weight.shape[0], device=g.device, dtype=g.dtype) if want_bias else None
    d = g.shape[2]
    for i in range(g.shape[0]):
        xs = x[i].transpose(0, 1).contiguous()
        gs = g[i].transpose(0, 1).contiguous()
        for kz in range(3):
            dz = kz - 1
            z0, z1 = max(0, -dz), d - max(0, dz)
            if z1 > z0:
                _conv_wgrad_(gwz[kz], gb if kz == 1 else None, xs[z0 + dz:z1 + dz], None, gs[z0:z1])
    return gwz.permute(1, 2, 0, 3, 4).contiguous(), gb


@_masked
class Conv3dBiasReluFn(Function):
    """y = [ReLU](Conv3d(x; W, 3x3x3, 'same') + b) on (n, c, d, h, w) (kspace_net.py:33-46)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        x = ops._dev(x, "conv3d input")
        relu = bool(relu) and ops.relu_on()
        y = ops.conv3d_bias_relu(x, weight, bias, relu)
        ctx.relu = relu
        ctx.save_for_backward(x, weight, y if relu else torch.empty(0))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        g = ops._dev(_c(gy), "conv3d output gradient")
        if ctx.relu:
            g = _relu_mask_(g.clone(), y)
        need = ctx.needs_input_grad
        gx = _conv3d_dgrad(g, weight) if need[0] else None
        gw = gb = None
        if need[1] or need[2]:
            gw, gb = _conv3d_wgrad(x, g, weight, need[2])
        return gx, gw if need[1] else None, gb, None


@_masked
class SensExpandFn(Function):
    """k = [M] FFT2(S x) (xpdnet.py:104-131, ForwardOperator): x (b, t, 1, h, w, 2), maps (b, 1, c, h, w, 2) -> (b, t, c, h, w, 2)."""

    @staticmethod
    def forward(ctx, x, sens, mask):
        x = ops._dev(x, "image"); sens = ops._dev(sens, "sens_maps")
        ctx.save_for_backward(x, sens, mask if mask is not None else torch.empty(0))
        ctx.masked = mask is not None
        return ops.sens_expand_dc(x, sens, None, mask, None, hard_mask=mask is not None)

    @staticmethod
    def backward(ctx, gk):
        x, sens, mask = ctx.saved_tensors
        gk = ops._dev(_c(gk), "k-space gradient")
        if ctx.masked:
            gk = ops.apply_mask(gk, mask)
        z = ops.fft2c(gk, inverse=True)                       # coil images of the gradient (the FFT is unitary)
        need = ctx.needs_input_grad
        gx = ops.sens_reduce(gk, sens) if need[0] else None
        gs = coil_accum(x, z) if need[1] else None            # sum_t conj(x_t) z_{t, c}
        return gx, gs, None


@_masked
class SensReduceFn(Function):
    """x = sum_c conj(S_c) IFFT2([M] k) (xpdnet.py:137-167, BackwardOperator) with gradients for the k-space too (the dual buffer is learned)."""

    @staticmethod
    def forward(ctx, k, sens, mask):
        k = ops._dev(k, "k-space"); sens = ops._dev(sens, "sens_maps")
        if mask is not None:
            k = ops.apply_mask(k, mask)
        ctx.save_for_backward(k, sens, mask if mask is not None else torch.empty(0))
        ctx.masked = mask is not None
        return ops.sens_reduce(k, sens)

    @staticmethod
    def backward(ctx, gx):
        k, sens, mask = ctx.saved_tensors
        gx = ops._dev(_c(gx), "image gradient")
        need = ctx.needs_input_grad
        gk = gs = None
        if need[0]:
            gk = ops.sens_expand_dc(gx, sens, None, mask if ctx.masked else None, None, hard_mask=ctx.masked)
        if need[1]:
            gs = coil_accum(gx, ops.fft2c(k, inverse=True))
        return gk, gs, None


# ---- convolutional-RNN cells (models/recurrent_varnet.py:153-259) --------------------------------------------------------------------
def _relu_mask_(g: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    check(lib().cine_relu_mask(g.data_ptr(), y.data_ptr(), g.numel(), _stream()), "cine_relu_mask")
    return g


def _conv_wgrad_(gw, gb, x0, x1, g):
    """gw (cout, c0 + c1, 3, 3) += weight gradient of conv3x3(cat(x0, x1)) from g; gb (cout) += bias gradient when not None."""
    n, c0, h, w = x0.shape
    c1 = 0 if x1 is None else x1.shape[1]
    cout = g.shape[1]
    L = lib()
    ws = torch.empty(L.cine_conv3x3_wgrad_ws_bytes(cout, c0 + c1, n), device=g.device, dtype=torch.uint8)
    check(L.cine_conv3x3_wgrad(x0.data_ptr(), c0, _p(x1), c1, g.data_ptr(), gw.data_ptr(), _p(gb), n, cout, h, w,
                               ws.data_ptr(), ws.numel(), _stream()), "cine_conv3x3_wgrad")


def _conv_dgrad(g, weight):
    """(n, cout, h, w) -> (n, cin, h, w) through the forward conv kernel on the flipped / transposed packing."""
    n, cout, h, w = g.shape
    cin = weight.shape[1]
    gx = torch.empty((n, cin, h, w), device=g.device, dtype=g.dtype)
    check(lib().cine_conv3x3_dgrad(g.data_ptr(), ops._pack("c3d", weight).data_ptr(), None, n, gx.data_ptr(), n, cout, cin, h, w, _stream()),
          "cine_conv3x3_dgrad")
    return gx


_zero_states = {}


def _zero_state(c: int, h: int, w: int, like: torch.Tensor) -> torch.Tensor:
    """A shared READ-ONLY (1, c, h, w) zero tensor per device (hid_init of the time sweeps, recurrent_varnet.py:236): no fill per step."""
    key = (like.device, c, h, w)
    z = _zero_states.get(key)
    if z is None:
        z = torch.zeros((1, c, h, w), device=like.device, dtype=torch.float32)
        if not torch.cuda.is_current_stream_capturing():
            _zero_states[key] = z
    return z


@_masked
class ConvSumFn(Function):
    """y = [ReLU](conv3x3(cat(x0, x1); W) + bias + addend): the "conv_x(a) + conv_h(b)" pairs of the CRNN body (recurrent_varnet.py:122-134)
    as one convolution over the concatenated inputs; W (cout, c0 + c1, 3, 3) in the module's own layout."""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, addend, relu):
        x0 = ops._dev(x0, "conv input")
        x1 = None if x1 is None else ops._dev(x1, "conv input 1")
        cout = weight.shape[0]
        relu = bool(relu) and ops.relu_on()
        y = ops.conv3x3_sum([x0] + ([x1] if x1 is not None else []), ops.pack_conv3x3(weight), None if bias is None else ops._dev(bias.detach(), "bias"),
                            cout, addend=None if addend is None else ops._dev(addend, "addend"), relu=relu)
        ctx.relu = relu
        ctx.has = (x1 is not None, bias is not None, addend is not None)
        ctx.save_for_backward(x0, x1 if x1 is not None else torch.empty(0), weight, y if relu else torch.empty(0))
        return y

    @staticmethod
    def backward(ctx, gy):
        x0, x1, weight, y = ctx.saved_tensors
        has_x1, has_bias, has_add = ctx.has
        x1 = x1 if has_x1 else None
        g = ops._dev(_c(gy), "conv output gradient")
        if ctx.relu:
            g = _relu_mask_(g.clone(), y)
        need = ctx.needs_input_grad
        c0 = x0.shape[1]
        gx0 = gx1 = gw = gb = None
        if need[0] or (has_x1 and need[1]):
            gx = _conv_dgrad(g, weight)
            gx0 = gx[:, :c0] if need[0] else None
            gx1 = gx[:, c0:] if has_x1 and need[1] else None
        if need[2] or (has_bias and need[3]):
            gw = torch.zeros_like(weight, memory_format=torch.contiguous_format)
            gb = torch.zeros(weight.shape[0], device=g.device, dtype=g.dtype) if has_bias and need[3] else None
            _conv_wgrad_(gw, gb, x0, x1, g)
        return gx0, gx1, gw if need[2] else None, gb, g if has_add and need[4] else None, None


@_masked
class BcrnnFn(Function):
    """BCRNNlayer.forward (recurrent_varnet.py:220-259) for batch 1: P_t = conv([hid_iter_t, x_t]; [W_ih2ih | W_i2h]) + the three biases for
    all frames in one launch, then both time sweeps h_t = ReLU(conv(h_prev; W_h2h) + P_t) step by step (both directions in one launch
    per step), output = forward + backward hidden states.  Every hidden state is kept; the backward pass is back-propagation through
    time on the same conv kernel (the flipped / transposed packing of W_h2h, the gradient of the frame riding in as the addend), then
    ONE weight-gradient launch per weight over all frames."""

    @staticmethod
    def forward(ctx, x, hid_iter, w_in, w_hh, bias):
        x = ops._dev(x, "BCRNN input"); hid_iter = ops._dev(hid_iter, "BCRNN iteration state")
        T, ch, h, w = x.shape
        c = w_hh.shape[0]
        wpi, wph = ops.pack_conv3x3(w_in), ops.pack_conv3x3(w_hh)
        P = ops.conv3x3_sum([hid_iter, x], wpi, ops._dev(bias.detach(), "bias"), c)
        zero = _zero_state(c, h, w, x)
        ctx.relu = bool(ops.relu_on())            # what the sweep applied
        if ops.BCRNN_SWEEP_IN_C:
            out, hf, hb = ops.bcrnn_sweep(P, wph, zero, keep=True)
            ctx.save_for_backward(x, hid_iter, w_in, w_hh, hf, hb)
            return out
        hf, hb, out = torch.empty_like(P), torch.empty_like(P), torch.empty_like(P)
        hid_f = hid_b = zero
        for s in range(T):
            i_f, i_b = s, T - 1 - s
            first = i_f < i_b
            if i_f == i_b:
                ops.crnn_step2(wph, (hid_f, P[i_f:i_f + 1], hf[i_f:i_f + 1], out[i_f:i_f + 1], True))
                ops.crnn_step2(wph, (hid_b, P[i_b:i_b + 1], hb[i_b:i_b + 1], out[i_b:i_b + 1], False))
            else:
                ops.crnn_step2(wph, (hid_f, P[i_f:i_f + 1], hf[i_f:i_f + 1], out[i_f:i_f + 1], first),
                               (hid_b, P[i_b:i_b + 1], hb[i_b:i_b + 1], out[i_b:i_b + 1], first))
            hid_f, hid_b = hf[i_f:i_f + 1], hb[i_b:i_b + 1]
        ctx.relu = bool(ops.relu_on())            # what crnn_step2 applied
        ctx.save_for_backward(x, hid_iter, w_in, w_hh, hf, hb)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, hid_iter, w_in, w_hh, hf, hb = ctx.saved_tensors
        gout = ops._dev(_c(gout), "BCRNN output gradient")
        T, ch, h, w = x.shape
        c = w_hh.shape[0]
        wdh = ops._pack("c3d", w_hh)
        if ops.BCRNN_SWEEP_IN_C:
            gf, gb, gP = ops.bcrnn_sweep_bwd(gout, wdh, _zero_state(c, h, w, x), hf, hb, ctx.relu)
            return BcrnnFn._tail(ctx, x, hid_iter, w_in, w_hh, hf, hb, gf, gb, gP)
        gf, gb = torch.empty_like(gout), torch.empty_like(gout)       # d loss / d (pre-activation) of the two chains
        for t in range(T - 1, -1, -1):                                 # forward-in-time chain, walked backwards
            if t == T - 1:
                gf[t:t + 1].copy_(gout[t:t + 1])
            else:
                ops.conv3x3_sum([gf[t + 1:t + 2]], wdh, None, c, addend=gout[t:t + 1], out=gf[t:t + 1])
            if ctx.relu:
                _relu_mask_(gf[t:t + 1], hf[t:t + 1])
        for t in range(T):                                             # backward-in-time chain
            if t == 0:
                gb[t:t + 1].copy_(gout[t:t + 1])
            else:
                ops.conv3x3_sum([gb[t - 1:t]], wdh, None, c, addend=gout[t:t + 1], out=gb[t:t + 1])
            if ctx.relu:
                _relu_mask_(gb[t:t + 1], hb[t:t + 1])
        gP = gf + gb
        return BcrnnFn._tail(ctx, x, hid_iter, w_in, w_hh, hf, hb, gf, gb, gP)

    @staticmethod
    def _tail(ctx, x, hid_iter, w_in, w_hh, hf, hb, gf, gb, gP):
        """From the pre-activation gradients of the two chains: everything upstream of P and the weight gradients."""
        T = x.shape[0]
        c = w_hh.shape[0]
        need = ctx.needs_input_grad
        gx = ghid = gw_in = gw_hh = gbias = None
        if need[0] or need[1]:
            gcat = _conv_dgrad(gP, w_in)
            ghid, gx = (gcat[:, :c] if need[1] else None), (gcat[:, c:] if need[0] else None)
        if need[2] or need[4]:
            gw_in = torch.zeros_like(w_in, memory_format=torch.contiguous_format)
            gbias = torch.zeros(c, device=gP.device, dtype=gP.dtype) if need[4] else None
            _conv_wgrad_(gw_in, gbias, hid_iter, x, gP)
        if need[3]:
            gw_hh = torch.zeros_like(w_hh, memory_format=torch.contiguous_format)
            if T > 1:       # h_{t-1} -> h_t (the first frame of each chain starts from zeros)
                _conv_wgrad_(gw_hh, None, hf[:T - 1], None, gf[1:])
                _conv_wgrad_(gw_hh, None, hb[1:], None, gb[:T - 1])
        return gx, ghid, gw_in if need[2] else None, gw_hh, gbias


def _dgrad_gated(g, wd, cin, addend=None, gate=None):
    """gx (n, cin, h, w) = [gate > 0] (conv(g; wd) + addend): cine_conv3x3_dgrad_gated (wd = the c3d packing of the (cout, cin, 3, 3) weight)."""
    n, cout, h, w = g.shape
    gx = torch.empty((n, cin, h, w), device=g.device, dtype=g.dtype)
    check(lib().cine_conv3x3_dgrad_gated(g.data_ptr(), wd.data_ptr(), _p(addend), _p(gate), gx.data_ptr(), n, cout, cin, h, w, _stream()),
          "cine_conv3x3_dgrad_gated")
    return gx


CRNN_BODY_FN = __import__("os").environ.get("CINE_CRNN_BODY_FN", "1") == "1"       # diagnostics (this binding): False = one autograd node per layer (BcrnnFn, ConvSumFn)
CRNN_SIDE_LANE = __import__("os").environ.get("CINE_CRNN_SIDE_LANE", "1") == "1"   # diagnostics: the body's weight gradients on the side stream


@_masked
class CrnnBodyFn(Function):
    """One cascade of the CRNN body (reference recurrent_varnet.py:116-136 with BCRNNlayer :220-259, batch 1) as ONE autograd node:
        P   = conv([s0, x]; [W_ih2ih | W_i2h]) + b_i2h + b_h2h + b_ih2ih          all frames, one launch
        x0  = both time sweeps of the BCRNN layer (cine_bcrnn_sweep)
        x_k = ReLU(conv([x_{k-1}, s_k]; [W_kx | W_kh]) + b_kx + b_kh),  k = 1, 2, 3
        out = conv(x3; W4) + b4 + residual
    forward(x, s0..s3, residual, packs, *params) -> (out, x0, x1, x2, x3); `packs` = the module's packed weights of this optimiser step
    (CRNNBody._train_packs: forward packings of the concatenated weights, input-gradient packings of every single weight).
    Backward: the input-gradient chain runs on the caller's stream -- every dgrad is cine_conv3x3_dgrad_gated (the gradient x_k receives as the
    next cascade's state rides in as the addend, the ReLU mask as the gate: no add / mask / concat-split kernels), the concatenated convs'
    gradients are two launches (one per input: both outputs contiguous), back-propagation through time is cine_bcrnn_sweep_bwd -- and the seven
    weight gradients of the body run beside it on the side stream (each depends only on a layer's output gradient), joined before the
    node returns.  Gradients go straight to the raw parameters: no cat / add nodes around the node."""

    @staticmethod
    def forward(ctx, x, s0, s1, s2, s3, residual, packs, *params):
        x = ops._dev(x, "CRNN body input"); residual = ops._dev(residual, "CRNN body residual")
        st = [ops._dev(s_, "CRNN iteration state") for s_ in (s0, s1, s2, s3)]
        T, ch, h, w = x.shape
        c = st[0].shape[1]
        relu = bool(ops.relu_on())
        P = ops.conv3x3_sum([st[0], x], packs["in"], packs["b_in"], c)
        zero = _zero_state(c, h, w, x)
        x0, hf, hb = ops.bcrnn_sweep(P, packs["hh"], zero, keep=True)
        feats = [x0]
        for k in (1, 2, 3):
            feats.append(ops.conv3x3_sum([feats[-1], st[k]], packs[f"p{k}"], packs[f"b{k}"], c, relu=True))
        out = ops.conv3x3_sum([feats[3]], packs["w4"], packs["b4"], residual.shape[1], addend=residual)
        ctx.relu, ctx.packs, ctx.dims = relu, packs, (T, ch, c, h, w, residual.shape[1])
        ctx.save_for_backward(x, *st, *feats, hf, hb)
        return (out, *feats)

    @staticmethod
    def backward(ctx, gout, g0, g1, g2, g3):
        x, s0, s1, s2, s3, x0, x1, x2, x3, hf, hb = ctx.saved_tensors
        T, ch, c, h, w, och = ctx.dims
        pk, relu = ctx.packs, ctx.relu
        st, feats, gfe = (s0, s1, s2, s3), (x0, x1, x2, x3), [g0, g1, g2, g3]
        gout = ops._dev(_c(gout), "CRNN body output gradient")
        gfe = [None if g is None else ops._dev(_c(g), "CRNN state gradient") for g in gfe]
        need = ctx.needs_input_grad
        dev = gout.device
        main = torch.cuda.current_stream(dev)
        side = ops.side_streams(dev, 1)[0] if CRNN_SIDE_LANE else None
        # weight gradients: zeroed buffers of the CONCATENATED layouts, one fill
        shapes = [(c, c + ch, 3, 3), (c, c, 3, 3), (c,), (c, 2 * c, 3, 3), (c,), (c, 2 * c, 3, 3), (c,), (c, 2 * c, 3, 3), (c,), (och, c, 3, 3), (och,)]
        sizes = [(int(torch.Size(s_).numel()) + 3) // 4 * 4 for s_ in shapes]
        flat = torch.zeros(sum(sizes), device=dev, dtype=torch.float32)
        bufs, off = [], 0
        for s_, n_ in zip(shapes, sizes):
            bufs.append(flat[off:off + torch.Size(s_).numel()].view(s_)); off += n_
        gw_in, gw_hh, gb_in, gW1, gb1, gW2, gb2, gW3, gb3, gw4, gb4 = bufs

        alive = []      # tensors the side stream reads stay referenced until the join (the allocator would hand their memory back to the main stream)

        def wgrad(gw, gb, a0, a1, g):
            """on the side stream, after everything enqueued on the main stream so far (g's producer)"""
            alive.append(g)
            if side is None:
                _conv_wgrad_(gw, gb, a0, a1, g)
                return
            side.wait_stream(main)
            with torch.cuda.stream(side):
                _conv_wgrad_(gw, gb, a0, a1, g)

        # ---- conv4 (+ residual): out = conv(x3; W4) + b4 + residual
        wgrad(gw4, gb4, x3, None, gout)
        g = _dgrad_gated(gout, pk["d4"], c, addend=gfe[3], gate=x3 if relu else None)          # d loss / d (pre-activation of x3)
        gst = [None, None, None, None]
        gWs, gbs = (None, gW1, gW2, gW3), (None, gb1, gb2, gb3)
        for k in (3, 2, 1):
            wgrad(gWs[k], gbs[k], feats[k - 1], st[k], g)
            if need[1 + k]:
                gst[k] = _dgrad_gated(g, pk[f"d{k}h"], c)
            g = _dgrad_gated(g, pk[f"d{k}x"], c, addend=gfe[k - 1], gate=(feats[k - 1] if relu and k > 1 else None))
        # ---- BCRNN: g = d loss / d x0 = d loss / d (hidden_f + hidden_b)
        gf, gb_, gP = ops.bcrnn_sweep_bwd(g, pk["dhh"], _zero_state(c, h, w, x), hf, hb, relu)
        wgrad(gw_in, gb_in, s0, x, gP)
        if T > 1:       # h_{t-1} -> h_t (the first frame of each chain starts from zeros)
            wgrad(gw_hh, None, hf[:T - 1], None, gf[1:])
            wgrad(gw_hh, None, hb[1:], None, gb_[:T - 1])
        if need[1]:
            gst[0] = _dgrad_gated(gP, pk["d_ih2ih"], c)
        gx = _dgrad_gated(gP, pk["d_i2h"], ch) if need[0] else None
        if side is not None:
            main.wait_stream(side)
        alive.clear()
        # parameter order of CRNNBody._body_params: w_ih2ih, w_i2h, w_h2h, b_i2h, b_h2h, b_ih2ih, then (w_kx, w_kh, b_kx, b_kh) for k = 1..3, w4, b4
        gp = [gw_in[:, :c], gw_in[:, c:], gw_hh, gb_in, gb_in, gb_in]
        for gW, gbk in ((gW1, gb1), (gW2, gb2), (gW3, gb3)):
            gp += [gW[:, :c], gW[:, c:], gbk, gbk]
        gp += [gw4, gb4]
        return (gx, gst[0], gst[1], gst[2], gst[3], gout if need[5] else None, None, *gp)


def coil_accum(g: Optional[torch.Tensor], z: torch.Tensor) -> torch.Tensor:
    """sum_t conj(g[b, t]) z[b, t, c] -> (b, 1, c, h, w, 2) (g None: sum_t z)."""
    b, t, c, h, w, _ = z.shape
    gs = torch.empty((b, 1, c, h, w, 2), device=z.device, dtype=z.dtype)
    check(lib().cine_coil_accum(_p(g), z.data_ptr(), gs.data_ptr(), b, t, c, h, w, 0, _stream()), "cine_coil_accum")
    return gs


@_masked
class CoilReduceFn(Function):
    """sens_reduce(mask * k) (varnet.py:187-194) as a function of the maps; k-space is data (no gradient)."""

    @staticmethod
    def forward(ctx, kspace, sens, mask):
        hyb = ops.kspace_to_hybrid(kspace, mask=mask)
        out = ops.hybrid_reduce(hyb, sens)
        ctx.save_for_backward(kspace, mask if mask is not None else torch.empty(0))
        ctx.has_mask = mask is not None
        return out

    @staticmethod
    def backward(ctx, gout):
        kspace, mask = ctx.saved_tensors
        if not ctx.needs_input_grad[1]:
            return None, None, None
        gout = ops._dev(_c(gout), "sens_reduce output gradient")
        k = ops.apply_mask(kspace, mask) if ctx.has_mask else kspace
        z = ops.fft2c(k, inverse=True)                        # coil images
        return None, coil_accum(gout, z), None


@_masked
class RssNormFn(Function):
    """x / rss_complex(x, coil) (varnet.py:58-59) on (b, c, h, w, 2)."""

    @staticmethod
    def forward(ctx, x):
        x = ops._dev(x, "sens-net output")
        ctx.save_for_backward(x)
        return ops.rss_normalise_(x.clone())

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = ops._dev(_c(gy), "rss_normalise output gradient")
        b, c, h, w, _ = x.shape
        gx = torch.empty_like(x)
        check(lib().cine_rss_normalise_bwd(gy.data_ptr(), x.data_ptr(), gx.data_ptr(), b, c, h, w, _stream()), "cine_rss_normalise_bwd")
        return gx


@_masked
class AbsFn(Function):
    """complex_abs (utils/math.py:48-62)."""

    @staticmethod
    def forward(ctx, x):
        x = ops._dev(x, "complex_abs input")
        ctx.save_for_backward(x)
        return ops.complex_abs(x)

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = ops._dev(_c(gy), "complex_abs output gradient")
        gx = torch.empty_like(x)
        check(lib().cine_complex_abs_bwd(gy.data_ptr(), x.data_ptr(), gx.data_ptr(), gy.numel(), _stream()), "cine_complex_abs_bwd")
        return gx


@_masked
class ConjFn(Function):
    """complex_conj (utils/math.py:36-45): the adjoint of conjugation is conjugation."""

    @staticmethod
    def forward(ctx, x):
        return ops.complex_conj(ops._dev(x, "complex_conj input"))

    @staticmethod
    def backward(ctx, gy):
        return ops.complex_conj(ops._dev(_c(gy), "complex_conj output gradient"))


@_masked
class RollFn(Function):
    """fftc.roll (utils/fftc.py:141-163): the adjoint of a circular shift is the opposite shift."""

    @staticmethod
    def forward(ctx, x, shifts, dims):
        ctx.sd = (tuple(shifts), tuple(dims))
        return ops.roll(ops._dev(x, "roll input"), list(shifts), list(dims))

    @staticmethod
    def backward(ctx, gy):
        shifts, dims = ctx.sd
        return ops.roll(ops._dev(_c(gy), "roll output gradient"), [-s for s in shifts], list(dims)), None, None


@_masked
class Pad2dFn(Function):
    """F.pad(x, [left, right, top, bottom]) of padding.pad_for_mwcnn (utils/padding.py:46): the adjoint crops."""

    @staticmethod
    def forward(ctx, x, left, right, top, bottom):
        ctx.pad = (int(left), int(right), int(top), int(bottom))
        return ops.pad2d(ops._dev(x, "pad input"), *ctx.pad)

    @staticmethod
    def backward(ctx, gy):
        left, right, top, bottom = ctx.pad
        h, w = gy.shape[-2], gy.shape[-1]
        return gy[..., top:h - bottom, left:w - right].contiguous(), None, None, None, None


@_masked
class CenteredFftFn(Function):
    """utils.fftc fft1c / ifft1c / fft2c / ifft2c (fftc.py:13-117) times a scalar: the centered orthonormal transform is unitary, so the
    adjoint of s F is s F^-1 -- the same kernel in the other direction on the output gradient."""

    @staticmethod
    def forward(ctx, x, two_d, inverse, scale):
        ctx.cfg = (bool(two_d), bool(inverse), float(scale))
        x = ops._dev(x, "fft input")
        out = ops.fft2c(x, inverse=inverse) if two_d else ops.fft1c(x, inverse=inverse)
        if scale != 1.0:
            ops.scale_(out, scale)
        return out

    @staticmethod
    def backward(ctx, gy):
        two_d, inverse, scale = ctx.cfg
        g = ops._dev(_c(gy), "fft output gradient")
        gx = ops.fft2c(g, inverse=not inverse) if two_d else ops.fft1c(g, inverse=not inverse)
        if scale != 1.0:
            ops.scale_(gx, scale)
        return gx, None, None, None


def needs_grad(*xs) -> bool:
    """True when autograd has to see this call: the utils shims then take a differentiable form (the raw kernels would cut the graph)."""
    return torch.is_grad_enabled() and any(torch.is_tensor(x) and x.requires_grad for x in xs)


def _axpby_lam(a, b, lam, kind, sign=1.0):
    out = torch.empty_like(b)
    check(lib().cine_axpby_lam(out.data_ptr(), _p(a), b.data_ptr(), b.numel(), lam.detach().data_ptr(), kind, float(sign), _stream()),
          "cine_axpby_lam")
    return out


def _lam_grad(dot: torch.Tensor, lam: torch.Tensor) -> torch.Tensor:
    """d/d lambda of softplus(lambda) times a device scalar."""
    return (dot * torch.sigmoid(lam.detach())).view(lam.shape)


@_masked
class AxpbyLamFn(Function):
    """a + softplus(lambda) * b: CineNet's right-hand side x_ref + v x_reg (cinenet.py:255-257)."""

    @staticmethod
    def forward(ctx, a, b, lam):
        a = ops._dev(a, "axpby a"); b = ops._dev(b, "axpby b")
        ctx.save_for_backward(b, lam)
        return _axpby_lam(a, b, lam, 0)

    @staticmethod
    def backward(ctx, g):
        b, lam = ctx.saved_tensors
        g = ops._dev(_c(g), "axpby gradient")
        ga = g if ctx.needs_input_grad[0] else None
        gb = _axpby_lam(None, g, lam, 0) if ctx.needs_input_grad[1] else None
        gl = _lam_grad(ops.dot(g, b), lam) if ctx.needs_input_grad[2] else None
        return ga, gb, gl


CG_SOLVER_IN_TRAINING = __import__("os").environ.get("CINE_CG_SOLVER_TRAIN", "1") == "1"      # diagnostics (this binding): False = iterate cine_normal_op_cg_fused from Python


@_masked
class ConjGradFn(Function):
    """CineNetBlock.ConjGrad (cinenet.py:136-171): K iterations of conjugate gradients on H x = b, H = A^H M A + softplus(lambda) I,
    from the start value x0.  The reference takes alpha and beta out of the graph (``.item()``, :159-169), so the iteration it
    differentiates is LINEAR in (x0, b) with the recorded step sizes; H is self-adjoint, so the adjoint recurrence is K + 1 more
    applications of the same operator (ops.h_operator: the image-space kernel cine_normal_op for row masks, the literal
    expand -> mask -> reduce chain for masks that vary along w), run backwards:
        gp_k = beta_k gp_{k+1} + alpha_k gx - alpha_k H(gr_{k+1} + gp_{k+1}),   gr_k = gr_{k+1} + gp_{k+1}
        gb = gr_0 + gp_0,  gx0 = gx - H(gb),  d/d v = -sum_k alpha_k <gr_{k+1} + gp_{k+1}, p_k> - <gb, x0>.
    Forward: the inference path's fused iteration (cine_normal_op_pd + cine_cg_step_pd2, four launches) with p_k kept and the scalars
    rr_k, p_k.d_k recorded on the device; backward: one operator application + ONE launch per iteration (cine_cg_adjoint_step)."""

    @staticmethod
    def forward(ctx, x0, b, lam, mask, sens, iters):
        x0 = ops._dev(x0, "CG start value"); b = ops._dev(b, "CG right-hand side")
        dev = x0.device
        if CG_SOLVER_IN_TRAINING and iters >= 1 and ops.is_row_mask(mask, sens.expand(-1, x0.shape[1], -1, -1, -1, -1)):
            # the whole solve in one C call (2 + 2 * iters launches), the directions and step sizes recorded by its own kernels
            x = x0.clone()
            rec = ops.conj_grad_rec(x, b, sens, mask, lam, iters)
            if rec is not None:
                p_rec, rr, pd = rec
                ctx.save_for_backward(x0, lam, mask, sens, rr, pd, *p_rec.unbind(0))
                return x
        one = torch.ones(1, device=dev, dtype=torch.float32)
        r = ops.axpby_dev(b, ops.h_operator(x0, sens, mask, lam), num=one, sign=-1.0)
        p = r.clone()
        x = x0.clone()
        # the step sizes stay on the device: rr[k] = r_k . r_k, pd[k] = p_k . H p_k  (alpha_k = rr[k] / pd[k], beta_k = rr[k + 1] / rr[k])
        rr = torch.empty(iters + 1, device=dev, dtype=torch.float32)
        pd = torch.empty(max(iters, 1), device=dev, dtype=torch.float32)
        ops.dot(r, r, out=rr[0:1])
        bsz, t = x0.shape[0], x0.shape[1]
        fused = ops.is_row_mask(mask, sens.expand(-1, t, -1, -1, -1, -1)) and \
            lib().cine_image_dc_ws_bytes(bsz, t, sens.shape[2], sens.shape[3], sens.shape[4]) > 0
        ps = []
        for k in range(iters):
            ps.append(p.clone() if fused else p)
            if fused:       # the inference path's four launches per iteration (operator with p.d partial sums, update, direction), p.d recorded
                ops.normal_op_cg_step(x, r, p, sens, mask, lam, rr[k:k + 1], rr[k + 1:k + 2], pd_out=pd[k:k + 1])
                continue
            d = ops.h_operator(p, sens, mask, lam)
            ops.dot(p, d, out=pd[k:k + 1])
            x = ops.axpby_dev(x, p, num=rr[k:k + 1], den=pd[k:k + 1])                   # x + alpha p
            r = ops.axpby_dev(r, d, num=rr[k:k + 1], den=pd[k:k + 1], sign=-1.0)        # r - alpha d
            ops.dot(r, r, out=rr[k + 1:k + 2])
            p = ops.axpby_dev(r, p, num=rr[k + 1:k + 2], den=rr[k:k + 1])               # r + beta p
        ctx.save_for_backward(x0, lam, mask, sens, rr, pd, *ps)
        return x

    @staticmethod
    def backward(ctx, gx):
        x0, lam, mask, sens, rr, pd = ctx.saved_tensors[:6]
        ps = ctx.saved_tensors[6:]
        gx = ops._dev(_c(gx), "CG output gradient")
        dev = gx.device
        L = lib()
        K = len(ps)
        nf = L.cine_cg_adjoint_part_floats()
        part = torch.empty(max(K, 1) * nf, device=dev, dtype=torch.float32)
        q = torch.zeros_like(gx)                                      # q_k = gr_{k+1} + gp_{k+1}: the gradient reaching r_{k+1}
        gp = torch.zeros_like(gx)
        for k in reversed(range(K)):
            hg = ops.h_operator(q, sens, mask, lam)
            check(L.cine_cg_adjoint_step(gp.data_ptr(), q.data_ptr(), gx.data_ptr(), hg.data_ptr(), ps[k].data_ptr(), gx.numel(),
                                         rr[k:k + 1].data_ptr(), pd[k:k + 1].data_ptr(), rr[k + 1:k + 2].data_ptr(),
                                         part[k * nf:].data_ptr(), _stream()), "cine_cg_adjoint_step")
        gb = q                                                        # gr_0 + gp_0
        need = ctx.needs_input_grad
        gx0 = None
        if need[0]:
            one = torch.ones(1, device=dev, dtype=torch.float32)
            gx0 = ops.axpby_dev(gx, ops.h_operator(gb, sens, mask, lam), num=one, sign=-1.0)
        glam = None
        if need[2]:
            gv = torch.empty(1, device=dev, dtype=torch.float32)
            check(L.cine_cg_adjoint_finish(part.data_ptr(), rr.data_ptr(), pd.data_ptr(), K, gv.data_ptr(), _stream()), "cine_cg_adjoint_finish")
            glam = _lam_grad(gv - ops.dot(gb, x0), lam)
        return gx0, (gb if need[1] else None), glam, None, None, None


@_masked
class SsimLossFn(Function):
    """SSIMLoss.forward (utils/losses.py:25-58) on GPU tensors: x = reconstruction, y = target, both (t, h, w)."""

    @staticmethod
    def forward(ctx, x, y, win, k1, k2):
        x = ops._dev(x, "SSIMLoss reconstruction"); y = ops._dev(y.detach(), "SSIMLoss target")
        t, h, w = x.shape
        nbytes = lib().cine_ssim_loss_ws_bytes(t, h, w, win)
        if nbytes == 0:
            raise ValueError(f"SSIMLoss: frames {tuple(x.shape)} too small for a {win} x {win} window")
        ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
        loss = torch.empty(1, device=x.device, dtype=torch.float32)
        check(lib().cine_ssim_loss(x.data_ptr(), y.data_ptr(), t, h, w, win, k1, k2, loss.data_ptr(), ws.data_ptr(), nbytes, _stream()),
              "cine_ssim_loss")
        ctx.save_for_backward(x, y, ws)
        ctx.win = win
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gloss):
        x, y, ws = ctx.saved_tensors
        t, h, w = x.shape
        gl = ops._dev(gloss.reshape(1).to(torch.float32), "SSIMLoss upstream gradient")
        gx = torch.empty_like(x)
        check(lib().cine_ssim_loss_bwd(x.data_ptr(), y.data_ptr(), t, h, w, ctx.win, gl.data_ptr(), ws.data_ptr(), ws.numel(), gx.data_ptr(),
                                       _stream()), "cine_ssim_loss_bwd")
        return gx, None, None, None, None


def masked_residual_backward(x0, sens, kref, mask):
    """``ops.masked_residual_backward`` as an autograd graph: the coil operators through their HIP kernels and adjoints (SensExpandFn / SensReduceFn),
    the two mask products and the subtraction in torch elementwise ops (differentiable in x0 and the maps)."""
    m = mask.to(x0.dtype)
    k = SensExpandFn.apply(x0, sens, None)
    k = (k * m - kref) * m
    return SensReduceFn.apply(k.contiguous(), sens, None)


def grad_mode(module: torch.nn.Module) -> bool:
    """True when the call should build an autograd graph (what ``loss.backward()`` in a training_step needs)."""
    return torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters())
