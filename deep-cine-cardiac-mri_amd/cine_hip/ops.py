"""Tensor-level wrappers over the C ABI: validate, allocate outputs, pass raw pointers.

PyTorch is plumbing here (device memory + the current HIP stream); all
arithmetic happens in libcine_hip.so.  Inputs must be CUDA(HIP) float32
tensors; anything else raises -- there is no CPU path.
"""
from typing import Optional, Sequence

import contextlib
import ctypes
import threading
import torch

from ._lib import CineHipError, check, lib

IN_EPS = 1e-5       # nn.InstanceNorm2d default eps (reference unet.py:161)
LRELU_SLOPE = 0.2   # nn.LeakyReLU(0.2)            (reference unet.py:162, mwcnn.py:204)
RELU_ON = True      # nn.ReLU of the CRNN cells / conv blocks (reference recurrent_varnet.py:126,178); False = identity.  Both are host-side
                    # defaults that the binding passes as ARGUMENTS of every call (the library keeps no activation state);
                    # `activation(...)` overrides them for the calling THREAD (the gradient fixtures without activation kinks)
_act_tls = threading.local()


def lrelu_slope() -> float:
    """The LeakyReLU slope this thread's calls pass to the library."""
    return getattr(_act_tls, "slope", LRELU_SLOPE)


def relu_on() -> bool:
    """Whether this thread's calls apply the CRNN / k-space-net ReLUs."""
    return getattr(_act_tls, "relu", RELU_ON)


@contextlib.contextmanager
def activation(slope: Optional[float] = None, relu: Optional[bool] = None):
    """Override the activations of the CALLING THREAD's calls: ``slope`` = LeakyReLU slope of the U-Net / MWCNN conv blocks (1 = identity),
    ``relu`` = False turns the nn.ReLU of the CRNN cells and the k-space net into the identity.  Per-call arguments at the C ABI."""
    old = (getattr(_act_tls, "slope", None), getattr(_act_tls, "relu", None))
    if slope is not None:
        if not 0.0 <= float(slope) <= 1.0:
            raise ValueError("activation: slope outside [0, 1]")
        _act_tls.slope = float(slope)
    if relu is not None:
        _act_tls.relu = bool(relu)
    try:
        yield
    finally:
        for name, v in zip(("slope", "relu"), old):
            if v is None:
                if hasattr(_act_tls, name):
                    delattr(_act_tls, name)
            else:
                setattr(_act_tls, name, v)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# ---- the calling thread's diagnostic kernel-selection mask (cine_set_conv_plane) ------------------------------------------------------
_CONV_PLANE_DEFAULT = 7


def conv_plane_mask() -> int:
    """The mask this thread set through ``set_conv_plane`` (default 7 = every lean kernel, lean weight gradients)."""
    return getattr(_act_tls, "plane_mask", _CONV_PLANE_DEFAULT)


def set_conv_plane(mask: int) -> None:
    """cine_set_conv_plane for the CALLING thread, remembered on the Python side as well: the library's mask is a per-thread setting and
    ``loss.backward()`` runs a Function's backward on the autograd engine's device thread, so ``cine_hip.autograd`` records this value
    in every Function's forward and re-applies it (``apply_conv_plane``) at the top of its backward -- a mask set before the
    forward pass also selects the kernels of the backward pass."""
    _act_tls.plane_mask = int(mask)
    check(lib().cine_set_conv_plane(int(mask)), "cine_set_conv_plane")
    _act_tls.applied_mask = int(mask)


def apply_conv_plane(mask: int) -> None:
    """Make `mask` the library's setting of the thread that calls this, without touching that thread's own record (``conv_plane_mask``):
    the caller restores with ``apply_conv_plane(conv_plane_mask())``."""
    if getattr(_act_tls, "applied_mask", _CONV_PLANE_DEFAULT) != mask:
        check(lib().cine_set_conv_plane(int(mask)), "cine_set_conv_plane")
        _act_tls.applied_mask = int(mask)


# ---- side streams: the caller-owned streams the branch / side-lane entry points fork onto --------------------------------------------
_SIDE_STREAMS = {}
import os as _os
UNET_BRANCHES = int(_os.environ.get("CINE_UNET_BRANCHES", "2"))   # default number of concurrent branches of a 2-D U-Net pass (cine_unet2d_forward_branches), see
                                                                  # `branches`: 2 -- the reference's scripts run ONE slice at a time (run_inference.py:53-61, every
                                                                  # training step), where two branches fill the chip better (cfg 2: 8.3 -> 7.6 ms per slice); a caller
                                                                  # that keeps many slices in flight on streams of its own asks for 1 (bench.py's timed mode).  The
                                                                  # environment variable belongs to THIS binding (A/B runs), the library reads none


BRANCH_SINGLE_SET = _os.environ.get("CINE_BRANCH_SINGLE_SET", "0") == "1"      # also cut a ONE-network pass (the sensitivity network's coil planes) into branches: measured
                                                                              # slower (cfg 5 one slice 4.42 -> 4.59 ms: 15 planes are too few to be worth a fork / join)
BRANCH_INTERLEAVE = _os.environ.get("CINE_BRANCH_INTERLEAVE", "0") == "1"     # diagnostics: enqueue the branches layer by layer (lockstep) instead of sequence by sequence


def side_streams(device: torch.device, count: int = 1):
    """`count` torch streams that belong to (device, the CURRENT stream): the library creates no streams, the binding does, once per
    main stream, outside any capture -- concurrent slices on different main streams never share a side stream."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    have = _SIDE_STREAMS.get(key)
    if have is None:
        have = _SIDE_STREAMS[key] = []
    while len(have) < count:
        _no_capture("a side stream")
        have.append(_pick_concurrent_stream(idx, [torch.cuda.current_stream(idx)] + have))
    return have[:count]


PROBE_SIDE_STREAMS = _os.environ.get("CINE_PROBE_SIDE_STREAMS", "1") == "1"      # (this binding) 0: take whatever stream torch hands out
_PROBE_US = 20


def streams_run_concurrently(a: "torch.cuda.Stream", b: "torch.cuda.Stream") -> bool:
    """Do two streams of one device execute side by side -- in the pattern the branches use them?  Stream b is forked from a with an event, both run a
    CHAIN of eight 20-us one-workgroup kernels (cine_spin), b is joined back into a: ~160 us when the chains overlap, ~320 us when they run one after
    the other -- which happens when the two streams share a hardware queue, and also (measured) for some pairs of queues that do not: a pair of plain
    concurrent kernels passes there, a fork / chain / join does not, so the probe times the real pattern.  Blocks the host for about a millisecond; meant
    for the moment a side stream is chosen, never for a hot path."""
    return _fork_join_probe_us(a, b) < 1.45 * _PROBE_CHAIN * _PROBE_US


_PROBE_CHAIN = 8


def _fork_join_probe_us(a: "torch.cuda.Stream", b: "torch.cuda.Stream") -> float:
    dev = a.device
    L = lib()
    best = float("inf")
    for _ in range(2):                     # best of two: the first launch on a stream pays its one-time set-up
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fork, done = torch.cuda.Event(), torch.cuda.Event()
        e0.record(a)
        fork.record(a)
        b.wait_event(fork)
        for _k in range(_PROBE_CHAIN):
            check(L.cine_spin(_PROBE_US, a.cuda_stream), "cine_spin")
            check(L.cine_spin(_PROBE_US, b.cuda_stream), "cine_spin")
        done.record(b)
        a.wait_event(done)
        e1.record(a)
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


def _pick_concurrent_stream(idx: int, others, tries: int = 12) -> "torch.cuda.Stream":
    """A new stream that really runs beside every stream in `others`.  torch hands out streams from a fixed pool and the runtime maps them onto
    GPU_MAX_HW_QUEUES hardware queues as they are first used: in a process that has touched many streams, the next one may share a queue with the
    very stream it is meant to overlap (measured: a cfg-2 training step 33 -> 46.6 ms with 26 idle streams left behind).  Candidates are probed
    (`streams_run_concurrently`) and the first that passes is kept; if none does, the last one is used and a RuntimeWarning says so."""
    cand = torch.cuda.Stream(device=idx)
    if not PROBE_SIDE_STREAMS:
        return cand
    for _ in range(tries):
        if all(streams_run_concurrently(o, cand) for o in others):
            return cand
        if _os.environ.get("CINE_PROBE_DEBUG"):
            print(f"# cine_hip: side-stream candidate {cand.cuda_stream:#x} rejected ({[round(_fork_join_probe_us(o, cand)) for o in others]} us)", flush=True)
        cand = torch.cuda.Stream(device=idx)
    import warnings
    warnings.warn("cine_hip: no stream on a hardware queue of its own was found for a side stream (GPU_MAX_HW_QUEUES is "
                  f"{_os.environ.get('GPU_MAX_HW_QUEUES', 'unset')}, many streams are alive): concurrent branches / weight-gradient side lanes may serialise",
                  RuntimeWarning, stacklevel=3)
    return cand


def release_side_streams(main_streams=None) -> None:
    """Forget (and thereby destroy) the side streams that belong to the given main streams (raw handles), or all of them.  Streams are a
    finite resource in effect: once more are alive than the runtime has hardware queues (GPU_MAX_HW_QUEUES), a new stream shares a queue with
    an existing one, and a side stream that shares its main stream's queue serialises with it.  Call after the work on those streams is done."""
    if main_streams is None:
        _SIDE_STREAMS.clear()
        return
    want = set(main_streams)
    for key in [k for k in _SIDE_STREAMS if k[1] in want]:
        del _SIDE_STREAMS[key]


def unet_branches() -> int:
    return getattr(_act_tls, "branches", UNET_BRANCHES)


@contextlib.contextmanager
def fixed_dropout(multipliers):
    """Tests: the training forwards of the CALLING THREAD take these Dropout multipliers (cine_unet2d_forward_branches' layout) instead of drawing
    them -- the same mask can then be fed to the oracle.  ``False``: no dropout whatever the modules say."""
    old = getattr(_act_tls, "dropout", None)
    _act_tls.dropout = multipliers
    try:
        yield
    finally:
        _act_tls.dropout = old


def fork_side(device: torch.device) -> "torch.cuda.Stream":
    """A side stream of (device, current stream) that has been made to wait for everything enqueued on the current stream so far; launch
    independent work on it (``with torch.cuda.stream(side): ...``, outputs allocated BEFORE entering it) and ``join_side(side)`` before the
    current stream reads the results.  (The second side stream of this main stream: a two-branch U-Net pass uses the first.)"""
    side = side_streams(device, 2)[1]
    side.wait_stream(torch.cuda.current_stream(device))
    return side


def join_side(side: "torch.cuda.Stream") -> None:
    torch.cuda.current_stream(side.device).wait_stream(side)


@contextlib.contextmanager
def branches(n: int):
    """Run the 2-D U-Net passes of the CALLING THREAD as `n` concurrent branches (1 = one stream, the classic launch sequence; 2 = the x-f and
    y-f networks of a cascade / two halves of the coils beside each other; 4 = each of those halved again).  Bit-identical results."""
    if n not in (1, 2, 4, 8):
        raise ValueError("branches: 1, 2, 4 or 8")
    old = getattr(_act_tls, "branches", None)
    _act_tls.branches = int(n)
    try:
        yield
    finally:
        if old is None:
            del _act_tls.branches
        else:
            _act_tls.branches = old


_pack_capture_streams = set()   # streams inside a training_capture(): the weight-pack kernels BELONG to the captured step (the weights change every replay)
_cache_epoch = 0              # part of every packed-weight cache key: bumped when a training capture ends


def cache_epoch() -> int:
    return _cache_epoch


@contextlib.contextmanager
def training_capture():
    """Around the hipGraph capture of a whole TRAINING step (cine_hip.train.GraphedTrainingStep): the parameters change at every replay, so
    the kernels that re-pack them must be part of the graph -- the capture guard of the packed-weight caches is lifted.  The packs made inside
    belong to the graph's memory pool and are refreshed only by its replays: when the capture ends every cache is invalidated (the epoch in
    their keys moves), so a later eager call (validation) packs the parameters' current values into memory of its own."""
    global _cache_epoch
    # scoped to the capturing STREAM, not the thread: loss.backward() packs on the autograd engine's thread (same stream), while another
    # thread's inference capture (another stream) keeps its guard
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    _pack_capture_streams.add(key)
    try:
        yield
    finally:
        _pack_capture_streams.discard(key)
        _cache_epoch += 1        # process-wide on purpose: EVERY thread's packed-weight caches are rebuilt (they may alias this capture's pool)


def _no_capture(what: str, pack: bool = False) -> None:
    """Caches that outlive a call (packed weights, per-stream scratch) must not be filled during hipGraph capture: the
    tensors would come from the graph's private pool yet stay referenced afterwards.  (pack: a packed-weight cache, allowed inside
    ``training_capture()``.)"""
    if torch.cuda.is_current_stream_capturing() and not (pack and (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream) in _pack_capture_streams):
        raise CineHipError(f"{what} would be created during hipGraph capture; run one eager forward on this stream first "
                           "(and re-capture after changing weights)")


def _dev(x: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if not x.is_cuda:
        raise CineHipError(f"{name}: tensor is on {x.device}; the HIP path needs a GPU tensor (no CPU fallback)")
    if x.dtype != dtype:
        raise CineHipError(f"{name}: dtype {x.dtype}, expected {dtype}")
    return x if x.is_contiguous() else x.contiguous()


def _pair(x: torch.Tensor, msg="Tensor does not have separate complex dim."):
    if x.shape[-1] != 2:
        raise ValueError(msg)


def _p(x: Optional[torch.Tensor]):
    return None if x is None else x.data_ptr()


# ------------------------------------------------------------------ centered FFTs
def fft2c(x: torch.Tensor, inverse: bool = False) -> torch.Tensor:
    """reference utils/fftc.py:59-110."""
    _pair(x)
    x = _dev(x, "fft2c input")
    if x.dim() < 3:
        raise ValueError("fft2c needs (..., h, w, 2)")
    h, w = x.shape[-3], x.shape[-2]
    out = torch.empty_like(x)
    nimg = x.numel() // (h * w * 2)
    check(lib().cine_fft2c(x.data_ptr(), out.data_ptr(), nimg, h, w, int(inverse), _stream()), "cine_fft2c")
    return out


def fft1c(x: torch.Tensor, inverse: bool = False, variant: int = 0) -> torch.Tensor:
    """reference utils/fftc.py:5-56 (variant 0) / xpdnet.py:466,500 (variant 1); acts on dim -2."""
    _pair(x)
    x = _dev(x, "fft1c input")
    n = x.shape[-2]
    out = torch.empty_like(x)
    check(lib().cine_fft1c(x.data_ptr(), out.data_ptr(), x.numel() // (2 * n), n, int(inverse), variant, _stream()),
          "cine_fft1c")
    return out


# ------------------------------------------------------------------ coil operators
def sens_reduce(k: torch.Tensor, sens: torch.Tensor, magnitude: bool = False,
                destroy_input: bool = False) -> torch.Tensor:
    """reference varnet.py:187-194 (and :150-151 with magnitude=True).
    k (b,t,c,h,w,2), sens (b,1,c,h,w,2) -> (b,t,1,h,w,2) or (b,t,h,w)."""
    _pair(k); _pair(sens)
    k = _dev(k, "k-space"); sens = _dev(sens, "sens_maps")
    b, t, c, h, w, _ = k.shape
    if sens.shape != (b, 1, c, h, w, 2):
        raise ValueError(f"sens_maps shape {tuple(sens.shape)} does not match k-space {tuple(k.shape)}")
    tmp = k if destroy_input else torch.empty_like(k)
    out = torch.empty((b, t, h, w) if magnitude else (b, t, 1, h, w, 2), device=k.device, dtype=k.dtype)
    check(lib().cine_sens_reduce(k.data_ptr(), sens.data_ptr(), out.data_ptr(), tmp.data_ptr(),
                                 b, t, c, h, w, int(magnitude), _stream()), "cine_sens_reduce")
    return out


def sens_expand_dc(img: torch.Tensor, sens: torch.Tensor, kref: Optional[torch.Tensor] = None,
                   mask: Optional[torch.Tensor] = None, lambda_reg: Optional[torch.Tensor] = None,
                   hard_mask: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """reference varnet.py:181-185 fused with the DC of :281-282 (or cinenet.py:129 with hard_mask)."""
    _pair(img); _pair(sens)
    img = _dev(img, "image"); sens = _dev(sens, "sens_maps")
    b, _, c, h, w, _ = sens.shape
    t = img.shape[1]
    if img.numel() != b * t * h * w * 2:
        raise ValueError(f"image shape {tuple(img.shape)} does not match sens_maps {tuple(sens.shape)}")
    if kref is not None:
        kref = _dev(kref, "ref_kspace")
        if kref.shape != (b, t, c, h, w, 2):
            raise ValueError("ref_kspace shape mismatch")
    if mask is not None:
        mask = _dev(mask, "mask", torch.uint8)
        if mask.numel() != b * t * h:
            raise ValueError(f"mask shape {tuple(mask.shape)}: expected (b, t, 1, h, 1, 1)")
    if lambda_reg is not None:
        lambda_reg = _dev(lambda_reg.detach(), "lambda_reg")
    if out is None:
        out = torch.empty((b, t, c, h, w, 2), device=img.device, dtype=img.dtype)
    check(lib().cine_sens_expand_dc(img.data_ptr(), sens.data_ptr(), _p(kref), _p(mask), _p(lambda_reg),
                                    out.data_ptr(), b, t, c, h, w, int(hard_mask), _stream()),
          "cine_sens_expand_dc")
    return out


def kspace_to_hybrid(k: torch.Tensor, out: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Centered column IFFT (first half of ifft2c): k-space -> hybrid space (image along h, k along w).
    With ``mask`` ((b,t,1,h,1,1) uint8, k (b,t,c,h,w,2)): the transform of mask * k, reading only the sampled rows."""
    _pair(k)
    k = _dev(k, "k-space")
    h, w = k.shape[-3], k.shape[-2]
    if out is None:
        out = torch.empty_like(k)
    if mask is not None:
        mask = _dev(mask, "mask", torch.uint8)
        b, t, c = k.shape[:3]
        if k.dim() != 6 or mask.numel() != b * t * h:
            raise ValueError(f"mask shape {tuple(mask.shape)}: expected (b, t, 1, h, 1, 1) for k-space {tuple(k.shape)}")
        check(lib().cine_masked_kspace_to_hybrid(k.data_ptr(), mask.data_ptr(), out.data_ptr(), b * t, c, h, w, _stream()),
              "cine_masked_kspace_to_hybrid")
        return out
    check(lib().cine_kspace_to_hybrid(k.data_ptr(), out.data_ptr(), k.numel() // (h * w * 2), h, w, _stream()),
          "cine_kspace_to_hybrid")
    return out


def as_mask_u8(mask: torch.Tensor, kspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The kernels read uint8 masks; the reference's models accept any numeric 0 / 1 mask that broadcasts against the k-space
    (``apply_mask`` returns a float one, data/transforms.py:66-92; varnet.py:281-282 multiplies).  Converted once, outside any kernel
    (not during hipGraph capture).  With ``kspace`` (b, t, c, h, w, 2) the mask is also brought into one of the two layouts the
    models dispatch on: a mask that is constant along w -- (b|1, t|1, 1, h, 1, 1) -- becomes the (b, t, 1, h, 1, 1) ROW mask of
    the fused kernels; one that varies along w becomes a GENERAL mask (b, t, 1, h, w, 1), served by the literal k-space chain."""
    if mask.dtype != torch.uint8:
        _no_capture("a uint8 copy of the sampling mask")
        mask = (mask != 0).to(torch.uint8)
    if kspace is None or kspace.dim() != 6:
        return mask
    b, t, _, h, w, _ = kspace.shape
    if mask.dim() != 6 or mask.shape[2] != 1 or mask.shape[5] != 1 or mask.shape[3] != h or mask.shape[0] not in (1, b) or \
            mask.shape[1] not in (1, t) or mask.shape[4] not in (1, w):
        raise ValueError(f"mask {tuple(mask.shape)} does not broadcast against k-space {tuple(kspace.shape)} as (b|1, t|1, 1, h, w|1, 1)")
    want = (b, t, 1, h, mask.shape[4], 1)
    if tuple(mask.shape) != want:
        _no_capture("an expanded copy of the sampling mask")
        mask = mask.expand(want).contiguous()
    return mask


def is_row_mask(mask: torch.Tensor, kspace: torch.Tensor) -> bool:
    """True for the reference's mask layout (b, t, 1, h, 1, 1) (data/transforms.py:341-343)."""
    b, t, _, h, _, _ = kspace.shape
    return mask.dim() == 6 and tuple(mask.shape) == (b, t, 1, h, 1, 1)


def is_general_mask(mask: torch.Tensor, kspace: torch.Tensor) -> bool:
    """True for a mask that varies along w: (b, t, 1, h, w, 1) (what ``as_mask_u8(mask, kspace)`` returns for one)."""
    b, t, _, h, w, _ = kspace.shape
    return mask.dim() == 6 and w > 1 and tuple(mask.shape) == (b, t, 1, h, w, 1)


def soft_dc_blend(model_term: torch.Tensor, ref_kspace: torch.Tensor, mask: torch.Tensor, lambda_reg: torch.Tensor) -> torch.Tensor:
    """The data-consistency line of reference varnet.py:281-282 for a GENERAL mask, term by term on the coil-wise k-space
    (torch elementwise kernels: the fused DC kernels read row masks).  Differentiable in model_term and lambda_reg."""
    m = mask.to(model_term.dtype)
    v = torch.nn.functional.softplus(lambda_reg)
    return (1 - m) * model_term + m * (model_term + v * ref_kspace) / (1 + v)


def masked_residual_backward(x0: torch.Tensor, sens: torch.Tensor, kref: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """XPDNet's K step + masked backward operator for a GENERAL mask (one that varies along w), literally on the coil-wise k-space:
    A^H [m (m A x0 - k_ref)]  (reference xpdnet.py:128-131: the forward operator's output is multiplied by whatever mask broadcasts and the
    measurements subtracted; :161-167: the backward operator masks again).  Row masks take the one-kernel image-space form (``image_dc`` with
    weights (1, 0, -1)).  Inference; the autograd form is ``cine_hip.autograd.masked_residual_backward``."""
    m = mask.to(x0.dtype)
    k = sens_expand_dc(x0, sens)                 # A x0, (b, t, c, h, w, 2)
    k = (k * m - kref) * m
    return sens_reduce(k, sens, destroy_input=True)


def sens_tile_pack(sens: torch.Tensor) -> Optional[torch.Tensor]:
    """The maps (b, 1, c, h, w, 2) in the column-tile-major order the h == 200 image-space operators read fastest (cine_sens_tile_pack);
    None where no kernel reads it.  The maps are constant over a forward pass: pack once after the sens-net, hand it to every
    ``image_dc`` / ``normal_op`` / ``normal_op_cg_step`` call as ``sens_tiled`` (results are identical with or without it)."""
    _pair(sens)
    sens = _dev(sens, "sens_maps")
    b, _, c, h, w, _ = sens.shape
    nfl = lib().cine_sens_tile_floats(b, c, h, w)
    if nfl == 0:
        return None
    out = torch.empty(nfl, device=sens.device, dtype=sens.dtype)
    check(lib().cine_sens_tile_pack(sens.data_ptr(), out.data_ptr(), b, c, h, w, _stream()), "cine_sens_tile_pack")
    return out


def image_dc(img: torch.Tensor, sens: torch.Tensor, zf: Optional[torch.Tensor], mask: torch.Tensor,
             lambda_reg: Optional[torch.Tensor] = None, weights=(1.0, 0.0, 0.0), magnitude: bool = False,
             out: Optional[torch.Tensor] = None, sens_tiled: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sens_reduce(DC(sens_expand(img))) of reference varnet.py:181-194, 281-282 on the coil-combined image (row masks):
    sum_c conj(S_c) IFFT_h[wgt * FFT_h(S_c img)] + beta * zf.  ``lambda_reg``: soft-DC weights from softplus(lambda);
    else ``weights`` = (w_sampled, w_unsampled, beta).  img (b,t,[1,]h,w,2) -> (b,t,1,h,w,2), or (b,t,h,w) magnitude."""
    _pair(img); _pair(sens)
    img = _dev(img, "image"); sens = _dev(sens, "sens_maps"); mask = _dev(mask, "mask", torch.uint8)
    b, _, c, h, w, _ = sens.shape
    t = img.shape[1]
    if img.numel() != b * t * h * w * 2 or mask.numel() != b * t * h:
        raise ValueError(f"image_dc: image {tuple(img.shape)} / mask {tuple(mask.shape)} do not match sens_maps {tuple(sens.shape)}")
    if zf is not None:
        zf = _dev(zf, "zero-filled image")
        if zf.numel() != img.numel():
            raise ValueError("image_dc: zero-filled image shape mismatch")
    lam = None if lambda_reg is None else _dev(lambda_reg.detach(), "lambda_reg")
    if out is None:
        out = torch.empty((b, t, h, w) if magnitude else (b, t, 1, h, w, 2), device=img.device, dtype=img.dtype)
    w1, w0, beta = (float(v) for v in weights)
    nbytes = lib().cine_image_dc_ws_bytes(b, t, c, h, w)
    ws = torch.empty(nbytes, device=img.device, dtype=torch.uint8) if nbytes else None
    check(lib().cine_image_dc_t(img.data_ptr(), sens.data_ptr(), _p(sens_tiled), _p(zf), mask.data_ptr(), _p(lam), w1, w0, beta,
                                out.data_ptr(), b, t, c, h, w, int(magnitude), _p(ws), nbytes, _stream()), "cine_image_dc")
    return out


def normal_op(img: torch.Tensor, sens: torch.Tensor, mask: torch.Tensor, lambda_reg: torch.Tensor,
              sens_tiled: Optional[torch.Tensor] = None) -> torch.Tensor:
    """A^H M A img + softplus(lambda) img for a row mask (CineNet's H operator, reference cinenet.py:121-133) -> (b,t,1,h,w,2)."""
    _pair(img); _pair(sens)
    img = _dev(img, "image"); sens = _dev(sens, "sens_maps"); mask = _dev(mask, "mask", torch.uint8)
    lam = _dev(lambda_reg.detach(), "lambda_reg")
    b, _, c, h, w, _ = sens.shape
    t = img.shape[1]
    if img.numel() != b * t * h * w * 2 or mask.numel() != b * t * h:
        raise ValueError(f"normal_op: image {tuple(img.shape)} / mask {tuple(mask.shape)} do not match sens_maps {tuple(sens.shape)}")
    out = torch.empty((b, t, 1, h, w, 2), device=img.device, dtype=img.dtype)
    nbytes = lib().cine_image_dc_ws_bytes(b, t, c, h, w)
    ws = torch.empty(nbytes, device=img.device, dtype=torch.uint8) if nbytes else None
    check(lib().cine_normal_op_t(img.data_ptr(), sens.data_ptr(), _p(sens_tiled), mask.data_ptr(), lam.data_ptr(), out.data_ptr(), b, t, c, h, w,
                                 _p(ws), nbytes, _stream()), "cine_normal_op")
    return out


def h_operator(x: torch.Tensor, sens: torch.Tensor, mask: torch.Tensor, lambda_reg: torch.Tensor,
               _hyb: Optional[torch.Tensor] = None, sens_tiled: Optional[torch.Tensor] = None) -> torch.Tensor:
    """CineNet's H = A^H M A + softplus(lambda) I (reference cinenet.py:121-133) for either mask layout: the one-kernel image-space
    operator for a (b, t, 1, h, 1, 1) row mask, the literal expand -> mask -> reduce chain for a mask that varies along w."""
    full = sens.expand(-1, x.shape[1], -1, -1, -1, -1)
    if is_row_mask(mask, full):
        return normal_op(x, sens, mask, lambda_reg, sens_tiled)
    if is_general_mask(mask, full):
        k = sens_expand_dc(x, sens)
        k = k * mask.to(k.dtype) + 0.0                       # cinenet.py:129
        return axpby_dev(sens_reduce(k, sens, destroy_input=True), x, lambda_reg=lambda_reg)
    hyb = expand_mask_hybrid(x, sens, mask, out=_hyb)         # b * t * h mask entries in another shape
    return axpby_dev(hybrid_reduce(hyb, sens), x, lambda_reg=lambda_reg)


def hybrid_reduce(hyb: torch.Tensor, sens: torch.Tensor, magnitude: bool = False) -> torch.Tensor:
    """Second half of sens_reduce (reference varnet.py:187-194) on hybrid-space data."""
    hyb = _dev(hyb, "hybrid k-space"); sens = _dev(sens, "sens_maps")
    b, t, c, h, w, _ = hyb.shape
    if sens.shape != (b, 1, c, h, w, 2):
        raise ValueError(f"sens_maps shape {tuple(sens.shape)} does not match {tuple(hyb.shape)}")
    out = torch.empty((b, t, h, w) if magnitude else (b, t, 1, h, w, 2), device=hyb.device, dtype=hyb.dtype)
    check(lib().cine_hybrid_reduce(hyb.data_ptr(), sens.data_ptr(), out.data_ptr(), b, t, c, h, w, int(magnitude),
                                   _stream()), "cine_hybrid_reduce")
    return out


def expand_dc_hybrid(img: torch.Tensor, sens: torch.Tensor, kref: torch.Tensor, mask: torch.Tensor,
                     lambda_reg: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sens_expand + soft DC (reference varnet.py:181-185, 281-282) followed by the column IFFT that the
    next cascade's sens_reduce starts with, without writing the k-space in between."""
    img = _dev(img, "image"); sens = _dev(sens, "sens_maps"); kref = _dev(kref, "ref_kspace")
    mask = _dev(mask, "mask", torch.uint8); lambda_reg = _dev(lambda_reg.detach(), "lambda_reg")
    b, t, c, h, w, _ = kref.shape
    if img.numel() != b * t * h * w * 2 or sens.shape != (b, 1, c, h, w, 2) or mask.numel() != b * t * h:
        raise ValueError("expand_dc_hybrid: shape mismatch")
    if out is None:
        out = torch.empty_like(kref)
    check(lib().cine_expand_dc_hybrid(img.data_ptr(), sens.data_ptr(), kref.data_ptr(), mask.data_ptr(),
                                      lambda_reg.data_ptr(), out.data_ptr(), b, t, c, h, w, 0, _stream()),
          "cine_expand_dc_hybrid")
    return out


def expand_mask_hybrid(img: torch.Tensor, sens: torch.Tensor, mask: torch.Tensor,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Hybrid-space image of M A x: sens_expand, hard mask (reference cinenet.py:126-129), column IFFT."""
    img = _dev(img, "image"); sens = _dev(sens, "sens_maps"); mask = _dev(mask, "mask", torch.uint8)
    b, _, c, h, w, _ = sens.shape
    t = img.shape[1]
    if img.numel() != b * t * h * w * 2 or mask.numel() != b * t * h:
        raise ValueError("expand_mask_hybrid: shape mismatch")
    if out is None:
        out = torch.empty((b, t, c, h, w, 2), device=img.device, dtype=img.dtype)
    check(lib().cine_expand_dc_hybrid(img.data_ptr(), sens.data_ptr(), None, mask.data_ptr(), None,
                                      out.data_ptr(), b, t, c, h, w, 1, _stream()), "cine_expand_dc_hybrid")
    return out


def acs_window_dev(mask: torch.Tensor) -> torch.Tensor:
    """The ACS window {first kept row, one past the last} of a row mask (b, t, 1, h, 1, 1) as an int32 tensor of two ON THE DEVICE
    (cine_acs_window; reference varnet.py:64-68 reads the mask on the host): nothing here waits for the GPU, so a forward pass can be
    enqueued -- or captured into a hipGraph -- straight from (masked_kspace, mask)."""
    if mask.shape[-2] != 1 or mask.shape[-1] != 1:
        raise ValueError("the ACS window is read from a row mask (reference varnet.py:64-68 indexes the mask's h axis); with a mask that "
                         "varies along w pass acs=(pad, n_low) or sens_maps")
    h = mask.shape[-3]
    rows = mask[:, 0].reshape(-1)
    rows = _dev(rows if rows.dtype == torch.float32 else rows.float(), "mask")
    win = torch.empty(2, device=rows.device, dtype=torch.int32)
    check(lib().cine_acs_window(rows.data_ptr(), rows.numel(), h, win.data_ptr(), _stream()), "cine_acs_window")
    return win


def sens_prologue(masked_kspace: torch.Tensor, row_lo, row_hi: Optional[int] = None) -> torch.Tensor:
    """reference varnet.py:71-74: ifft2c(mask_center(mean_t(k))).  row_lo: an int (with row_hi) or the device window of ``acs_window_dev``."""
    k = _dev(masked_kspace, "masked_kspace")
    b, t, c, h, w, _ = k.shape
    out = torch.empty((b, c, h, w, 2), device=k.device, dtype=k.dtype)
    if isinstance(row_lo, torch.Tensor):
        if row_lo.dtype != torch.int32 or row_lo.numel() != 2 or row_lo.device != k.device:
            raise ValueError("sens_prologue: the device window is an int32 tensor of two on the k-space's device")
        check(lib().cine_sens_prologue_win(k.data_ptr(), out.data_ptr(), b, t, c, h, w, row_lo.data_ptr(), _stream()), "cine_sens_prologue_win")
        return out
    check(lib().cine_sens_prologue(k.data_ptr(), out.data_ptr(), b, t, c, h, w, int(row_lo), int(row_hi), _stream()),
          "cine_sens_prologue")
    return out


def rss_normalise_(x: torch.Tensor) -> torch.Tensor:
    """reference varnet.py:58-59, in place on (b,c,h,w,2)."""
    assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32
    b, c, h, w, _ = x.shape
    check(lib().cine_rss_normalise(x.data_ptr(), b, c, h, w, _stream()), "cine_rss_normalise")
    return x


def complex_abs(x: torch.Tensor) -> torch.Tensor:
    """reference utils/math.py:48-62."""
    _pair(x)
    x = _dev(x, "complex_abs input")
    out = torch.empty(x.shape[:-1], device=x.device, dtype=x.dtype)
    check(lib().cine_complex_abs(x.data_ptr(), out.data_ptr(), out.numel(), _stream()), "cine_complex_abs")
    return out


# ------------------------------------------------------------------ the reference's small helpers (utils/math.py, coil_combine.py, fftc.roll, padding.py)
def complex_mul(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """reference utils/math.py:20-33 with broadcasting: the operands are read through their own strides (0 on broadcast dimensions)."""
    x = _dev(x, "complex_mul x"); y = _dev(y, "complex_mul y")
    shape = torch.broadcast_shapes(x.shape[:-1], y.shape[:-1])
    if len(shape) > 6:
        raise ValueError("complex_mul: at most 6 dimensions besides the complex pair")
    xe, ye = x.expand(*shape, 2), y.expand(*shape, 2)
    out = torch.empty((*shape, 2), device=x.device, dtype=x.dtype)
    if out.numel() == 0:
        return out
    nd = len(shape)
    sh = (ctypes.c_int * max(nd, 1))(*shape)
    xs = (ctypes.c_long * max(nd, 1))(*[s // 2 for s in xe.stride()[:-1]])       # (contiguous pairs: every stride is even)
    ys = (ctypes.c_long * max(nd, 1))(*[s // 2 for s in ye.stride()[:-1]])
    check(lib().cine_complex_mul(x.data_ptr(), y.data_ptr(), out.data_ptr(), nd, sh, xs, ys, _stream()), "cine_complex_mul")
    return out


def complex_conj(x: torch.Tensor) -> torch.Tensor:
    x = _dev(x, "complex_conj input")
    out = torch.empty_like(x)
    if x.numel():
        check(lib().cine_complex_conj(x.data_ptr(), out.data_ptr(), x.numel() // 2, _stream()), "cine_complex_conj")
    return out


def complex_abs_sq(x: torch.Tensor) -> torch.Tensor:
    x = _dev(x, "complex_abs_sq input")
    out = torch.empty(x.shape[:-1], device=x.device, dtype=x.dtype)
    if out.numel():
        check(lib().cine_complex_abs_sq(x.data_ptr(), out.data_ptr(), out.numel(), _stream()), "cine_complex_abs_sq")
    return out


def rss(x: torch.Tensor, dim: int, is_complex: bool) -> torch.Tensor:
    """reference utils/coil_combine.py: sqrt(sum over `dim` of x^2), or of |x|^2 for (..., 2) data (dim counts the tensor's own dims)."""
    x = _dev(x, "rss input")
    nd = x.dim()
    dim = dim % nd
    if is_complex and dim == nd - 1:
        raise ValueError("rss_complex: dim is the complex pair")
    lead = x.shape[:dim]; k = x.shape[dim]; tail = x.shape[dim + 1:nd - 1] if is_complex else x.shape[dim + 1:]
    outer = 1
    for d in lead: outer *= d
    inner = 1
    for d in tail: inner *= d
    out = torch.empty(tuple(lead) + tuple(tail), device=x.device, dtype=x.dtype)
    if out.numel():
        check(lib().cine_rss(x.data_ptr(), out.data_ptr(), outer, k, inner, int(is_complex), _stream()), "cine_rss")
    return out


def roll(x: torch.Tensor, shifts, dims) -> torch.Tensor:
    """reference utils/fftc.py:141-163: one kernel launch per rolled dimension."""
    x = _dev(x, "roll input")
    cur = x
    for s, d in zip(shifts, dims):
        d = d % cur.dim()
        n = cur.shape[d]
        outer = 1
        for v in cur.shape[:d]: outer *= v
        inner = 1
        for v in cur.shape[d + 1:]: inner *= v
        out = torch.empty_like(cur)
        if cur.numel():
            check(lib().cine_roll(cur.data_ptr(), out.data_ptr(), outer, n, inner, int(s) % n if n else 0, _stream()), "cine_roll")
        cur = out
    return cur if cur is not x else x.clone()


def pad2d(x: torch.Tensor, left: int, right: int, top: int, bottom: int) -> torch.Tensor:
    """Zero padding of the last two dimensions (reference utils/padding.py:46 F.pad(x, [left, right, top, bottom]))."""
    x = _dev(x, "pad input")
    h, w = x.shape[-2], x.shape[-1]
    hp, wp = h + top + bottom, w + left + right
    out = torch.empty(x.shape[:-2] + (hp, wp), device=x.device, dtype=x.dtype)
    planes = x.numel() // (h * w) if h * w else 0
    if out.numel():
        check(lib().cine_pad2d(x.data_ptr(), out.data_ptr(), planes, h, w, top, left, hp, wp, _stream()), "cine_pad2d")
    return out


# ------------------------------------------------------------------ the steps either side of the path
def apply_mask(kspace: torch.Tensor, mask: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """kspace * mask + 0.0 (reference data/transforms.py:66-92) for a row mask; kspace (..., c, h, w, 2) with the mask's
    (..., 1, h, 1, 1) leading dims, e.g. (b, t, c, h, w, 2) & (b, t, 1, h, 1, 1) or (t, c, h, w, 2) & (t, 1, h, 1, 1)."""
    _pair(kspace)
    kspace = _dev(kspace, "k-space"); mask = _dev(mask, "mask", torch.uint8)
    c, h, w = kspace.shape[-4], kspace.shape[-3], kspace.shape[-2]
    bt = kspace.numel() // (c * h * w * 2)
    if mask.numel() != bt * h:
        raise ValueError(f"apply_mask: mask {tuple(mask.shape)} does not match k-space {tuple(kspace.shape)}")
    if out is None:
        out = torch.empty_like(kspace)
    check(lib().cine_apply_mask(kspace.data_ptr(), mask.data_ptr(), out.data_ptr(), bt, c, h, w, _stream()), "cine_apply_mask")
    return out


def scale_(x: torch.Tensor, s: float) -> torch.Tensor:
    assert x.is_cuda and x.is_contiguous() and x.dtype == torch.float32
    check(lib().cine_scale(x.data_ptr(), x.numel(), float(s), _stream()), "cine_scale")
    return x


def zero_filled_rss(kspace: torch.Tensor, destroy_input: bool = False) -> torch.Tensor:
    """Zero-filled reconstruction of reference run_inference.py:64-67: (b,t,c,h,w,2) -> (b,t,h,w)."""
    _pair(kspace)
    k = _dev(kspace, "k-space")
    b, t, c, h, w, _ = k.shape
    tmp = k if destroy_input else torch.empty_like(k)
    out = torch.empty((b, t, h, w), device=k.device, dtype=k.dtype)
    check(lib().cine_zero_filled_rss(k.data_ptr(), out.data_ptr(), tmp.data_ptr(), b, t, c, h, w, _stream()), "cine_zero_filled_rss")
    return out


def image_metrics(gt: torch.Tensor, pred: torch.Tensor, maxval: Optional[float] = None, per_frame_range: bool = False,
                  win_size: int = 7, k1: float = 0.01, k2: float = 0.03) -> dict:
    """SSIM / NMSE / PSNR / MSE of pred (t, hp, wp) against gt (t, hg, wg) on the device, after the reference's
    center_crop_to_smallest (data/transforms.py:161-183): reference utils/evaluate.py:6-50, and utils/losses.py:25-58 with
    ``per_frame_range``.  Returns 0-d / (t,) float64 device tensors: ssim, nmse, psnr, mse, ssim_frames."""
    gt = _dev(gt, "target"); pred = _dev(pred, "reconstruction")
    if gt.dim() != 3:
        raise ValueError("Unexpected number of dimensions in ground truth.")
    if pred.dim() != 3 or pred.shape[0] != gt.shape[0]:
        raise ValueError("Ground truth dimensions does not match pred.")
    t, hg, wg = gt.shape
    _, hp, wp = pred.shape
    nbytes = lib().cine_image_metrics_ws_bytes(t, hg, wg, hp, wp, win_size)
    if nbytes == 0:
        raise ValueError(f"image_metrics: frames {tuple(gt.shape)} / {tuple(pred.shape)} too small for a {win_size} window")
    ws = torch.empty(nbytes, device=gt.device, dtype=torch.uint8)
    out = torch.empty(4 + t, device=gt.device, dtype=torch.float64)
    mode = 1 if per_frame_range else (0 if maxval is None else 2)
    check(lib().cine_image_metrics(gt.data_ptr(), pred.data_ptr(), t, hg, wg, hp, wp, win_size, k1, k2, mode,
                                   0.0 if maxval is None else float(maxval), out.data_ptr(), ws.data_ptr(), nbytes, _stream()),
          "cine_image_metrics")
    return {"ssim": out[0], "nmse": out[1], "psnr": out[2], "mse": out[3], "ssim_frames": out[4:]}


# ------------------------------------------------------------------ CG vector ops (device-side scalars)
_dot_ws = {}


def dot(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """torch.dot(a.flatten(), b.flatten()) into a 1-element device tensor (reference cinenet.py:148,155,163)."""
    a = _dev(a, "dot lhs"); b = _dev(b, "dot rhs")
    if a.numel() != b.numel():
        raise ValueError("dot: size mismatch")
    if out is None:
        out = torch.empty(1, device=a.device, dtype=a.dtype)
    key = (a.device, torch.cuda.current_stream().cuda_stream)
    ws = _dot_ws.get(key)
    if ws is None:
        _no_capture("the dot-product workspace of this stream")
        ws = _dot_ws[key] = torch.empty(lib().cine_dot_ws_bytes(), device=a.device, dtype=torch.uint8)
    check(lib().cine_dot(a.data_ptr(), b.data_ptr(), a.numel(), out.data_ptr(), ws.data_ptr(), _stream()), "cine_dot")
    return out


_cg_ws = {}
CG_SOLVER = True     # A/B switch: False = one normal_op_cg_step call per iteration (3 launches each) instead of cine_conj_grad (2 launches each)
FUSED_CG = True      # A/B switch of the diagnostics (tools/whatif_cfg4.py): False = operator + partial-sum pass + update + direction (4 launches)


def cg_step(x: torch.Tensor, r: torch.Tensor, p: torch.Tensor, d: torch.Tensor, rr_old: torch.Tensor, rr_new: torch.Tensor) -> torch.Tensor:
    """One conjugate-gradient iteration after d = H p (reference cinenet.py:155-169), x / r / p updated in place, rr_new written
    (a different 1-element tensor than rr_old): three launches, bit-identical to dot + axpby_dev."""
    for t_, name in ((x, "x"), (r, "r"), (p, "p"), (d, "d")):
        if not (t_.is_cuda and t_.is_contiguous() and t_.dtype == torch.float32 and t_.numel() == x.numel()):
            raise ValueError(f"cg_step: {name} must be a contiguous float32 GPU tensor of x's size")
    key = (x.device, torch.cuda.current_stream().cuda_stream)
    ws = _cg_ws.get(key)
    if ws is None:
        _no_capture("the conjugate-gradient workspace of this stream")
        ws = _cg_ws[key] = torch.empty(lib().cine_cg_ws_bytes(), device=x.device, dtype=torch.uint8)
    check(lib().cine_cg_step(x.data_ptr(), r.data_ptr(), p.data_ptr(), d.data_ptr(), x.numel(), rr_old.data_ptr(), rr_new.data_ptr(),
                             ws.data_ptr(), _stream()), "cine_cg_step")
    return rr_new


def normal_op_cg_step(x, r, p, sens, mask, lambda_reg, rr_old, rr_new, pd_out: Optional[torch.Tensor] = None,
                      sens_tiled: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One conjugate-gradient iteration of reference cinenet.py:153-169 for a row mask: d = H p with the partial sums of p.d produced by
    the operator's last kernel (cine_normal_op_pd), then the alpha / x / r / r.r / beta / p updates (cine_cg_step_pd).  Falls back to
    normal_op + cg_step where the operator has no partial-sum kernel."""
    b, _, c, h, w, _ = sens.shape
    t = p.shape[1]
    nbytes = lib().cine_image_dc_ws_bytes(b, t, c, h, w)
    if nbytes == 0:
        if pd_out is not None:
            raise CineHipError("normal_op_cg_step: this shape's operator has no partial-sum kernel, p.d cannot be recorded")
        return cg_step(x, r, p, normal_op(p, sens, mask, lambda_reg), rr_old, rr_new)
    lam = _dev(lambda_reg.detach(), "lambda_reg")
    fbytes = lib().cine_cg_fused_ws_bytes(b, t, c, h, w) if FUSED_CG else 0
    if fbytes:                    # three launches: the operator's coil-group sums are consumed by the update kernel, H p is never written
        for t_, name in ((x, "x"), (r, "r"), (p, "p")):
            if not (t_.is_cuda and t_.is_contiguous() and t_.dtype == torch.float32 and t_.numel() == b * t * h * w * 2):
                raise ValueError(f"normal_op_cg_step: {name} must be a contiguous float32 GPU tensor of (b, t, 1, h, w, 2)")
        key = (x.device, torch.cuda.current_stream().cuda_stream, "fused", fbytes)
        fws = _cg_ws.get(key)
        if fws is None:
            _no_capture("the conjugate-gradient workspace of this stream")
            fws = _cg_ws[key] = torch.empty(fbytes, device=x.device, dtype=torch.uint8)
        dws = torch.empty(nbytes, device=p.device, dtype=torch.uint8)
        check(lib().cine_normal_op_cg_fused_t(x.data_ptr(), r.data_ptr(), p.data_ptr(), _dev(sens, "sens_maps").data_ptr(), _p(sens_tiled),
                                              _dev(mask, "mask", torch.uint8).data_ptr(), lam.data_ptr(), rr_old.data_ptr(), rr_new.data_ptr(),
                                              _p(pd_out), b, t, c, h, w, dws.data_ptr(), nbytes, fws.data_ptr(), fbytes, _stream()),
              "cine_normal_op_cg_fused")
        return rr_new
    key = (x.device, torch.cuda.current_stream().cuda_stream)
    ws = _cg_ws.get(key)
    if ws is None:
        _no_capture("the conjugate-gradient workspace of this stream")
        ws = _cg_ws[key] = torch.empty(lib().cine_cg_ws_bytes(), device=x.device, dtype=torch.uint8)
    d = torch.empty((b, t, 1, h, w, 2), device=p.device, dtype=p.dtype)
    dws = torch.empty(nbytes, device=p.device, dtype=torch.uint8)
    check(lib().cine_normal_op_pd(p.data_ptr(), sens.data_ptr(), mask.data_ptr(), lam.data_ptr(), d.data_ptr(), ws.data_ptr(), b, t, c, h, w,
                                  dws.data_ptr(), nbytes, _stream()), "cine_normal_op_pd")
    if pd_out is not None:        # training: p.d recorded for the adjoint recurrence
        check(lib().cine_cg_step_pd2(x.data_ptr(), r.data_ptr(), p.data_ptr(), d.data_ptr(), x.numel(), rr_old.data_ptr(), rr_new.data_ptr(),
                                     pd_out.data_ptr(), ws.data_ptr(), _stream()), "cine_cg_step_pd2")
        return rr_new
    check(lib().cine_cg_step_pd(x.data_ptr(), r.data_ptr(), p.data_ptr(), d.data_ptr(), x.numel(), rr_old.data_ptr(), rr_new.data_ptr(),
                                ws.data_ptr(), _stream()), "cine_cg_step_pd")
    return rr_new


def conj_grad(x: torch.Tensor, rhs: torch.Tensor, sens: torch.Tensor, mask: torch.Tensor, lambda_reg: torch.Tensor, iters: int,
              sens_tiled: Optional[torch.Tensor] = None, rhs_is_ref: bool = False) -> Optional[torch.Tensor]:
    """reference cinenet.py:136-171 for a row mask: the whole solve in 2 + 2 * iters launches (cine_conj_grad), x updated IN PLACE and
    returned.  rhs_is_ref: `rhs` is x_ref and the right-hand side x_ref + softplus(lambda) x is formed inside (cinenet.py:106-107).
    None when the shape has no such path (h != 200 or <= 5 coils): the caller iterates normal_op_cg_step."""
    b, _, c, h, w, _ = sens.shape
    t = x.shape[1]
    nbytes = lib().cine_conj_grad_ws_bytes(b, t, c, h, w)
    if nbytes == 0:
        return None
    for t_, name in ((x, "x"), (rhs, "rhs")):
        if not (t_.is_cuda and t_.is_contiguous() and t_.dtype == torch.float32 and t_.numel() == b * t * h * w * 2):
            raise ValueError(f"conj_grad: {name} must be a contiguous float32 GPU tensor of (b, t, 1, h, w, 2)")
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    check(lib().cine_conj_grad(x.data_ptr(), rhs.data_ptr(), int(rhs_is_ref), _dev(sens, "sens_maps").data_ptr(), _p(sens_tiled), _dev(mask, "mask", torch.uint8).data_ptr(),
                               _dev(lambda_reg.detach(), "lambda_reg").data_ptr(), int(iters), b, t, c, h, w, ws.data_ptr(), nbytes, _stream()),
          "cine_conj_grad")
    return x


def conj_grad_rec(x: torch.Tensor, rhs: torch.Tensor, sens: torch.Tensor, mask: torch.Tensor, lambda_reg: torch.Tensor, iters: int,
                  sens_tiled: Optional[torch.Tensor] = None):
    """``conj_grad`` for training (cine_conj_grad_rec): x updated in place; returns (p_rec (iters, *x.shape), rr (iters + 1), pd (iters)) -- every
    direction and the step sizes' numerators / denominators, what ConjGradFn's adjoint recurrence reads -- or None when the shape has no such path."""
    b, _, c, h, w, _ = sens.shape
    t = x.shape[1]
    nbytes = lib().cine_conj_grad_ws_bytes(b, t, c, h, w)
    if nbytes == 0 or iters < 1:
        return None
    for t_, name in ((x, "x"), (rhs, "rhs")):
        if not (t_.is_cuda and t_.is_contiguous() and t_.dtype == torch.float32 and t_.numel() == b * t * h * w * 2):
            raise ValueError(f"conj_grad_rec: {name} must be a contiguous float32 GPU tensor of (b, t, 1, h, w, 2)")
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    p_rec = torch.empty((iters,) + tuple(x.shape), device=x.device, dtype=x.dtype)
    rr = torch.empty(iters + 1, device=x.device, dtype=torch.float32)
    pd = torch.empty(iters, device=x.device, dtype=torch.float32)
    check(lib().cine_conj_grad_rec(x.data_ptr(), rhs.data_ptr(), 0, _dev(sens, "sens_maps").data_ptr(), _p(sens_tiled), _dev(mask, "mask", torch.uint8).data_ptr(),
                                   _dev(lambda_reg.detach(), "lambda_reg").data_ptr(), int(iters), b, t, c, h, w, ws.data_ptr(), nbytes,
                                   p_rec.data_ptr(), rr.data_ptr(), pd.data_ptr(), _stream()), "cine_conj_grad_rec")
    return p_rec, rr, pd


def axpby_dev(a: torch.Tensor, b: torch.Tensor, num: Optional[torch.Tensor] = None, den: Optional[torch.Tensor] = None,
              lambda_reg: Optional[torch.Tensor] = None, sign: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = a + sign * s * b, s = num/den (device scalars) or softplus(lambda_reg)."""
    a = _dev(a, "axpby a"); b = _dev(b, "axpby b")
    if out is None:
        out = torch.empty_like(a)
    lam = None if lambda_reg is None else _dev(lambda_reg.detach(), "lambda_reg")
    check(lib().cine_axpby_dev(out.data_ptr(), a.data_ptr(), b.data_ptr(), a.numel(), _p(num), _p(den), _p(lam),
                               float(sign), _stream()), "cine_axpby_dev")
    return out


# ------------------------------------------------------------------ NormUnet halves / rotations
def pad16(n: int) -> int:
    return ((n - 1) | 15) + 1


def normunet_pack(x: torch.Tensor, norm: bool = True):
    """(n,h,w,2) -> planes (n,2,hp,wp), stats (n,2,2); reference norm_unet.py:48-86.
    norm=False: plain (re, im) -> 2-channel repack, no normalisation / padding (cinenet.py:242), stats None."""
    x = _dev(x, "normunet_pack input")
    n, h, w, _ = x.shape
    hp, wp = (pad16(h), pad16(w)) if norm else (h, w)
    planes = torch.empty((n, 2, hp, wp), device=x.device, dtype=x.dtype)
    stats = torch.empty((n, 2, 2), device=x.device, dtype=x.dtype) if norm else None
    check(lib().cine_normunet_pack(x.data_ptr(), planes.data_ptr(), _p(stats), n, h, w, int(norm), _stream()),
          "cine_normunet_pack")
    return planes, stats


def normunet_unpack(planes: torch.Tensor, stats: Optional[torch.Tensor], h: int, w: int) -> torch.Tensor:
    """reference norm_unet.py:88-96, 71-74, 53-57 (stats None: inverse of the plain repack)."""
    planes = _dev(planes, "planes")
    n = planes.shape[0]
    y = torch.empty((n, h, w, 2), device=planes.device, dtype=planes.dtype)
    check(lib().cine_normunet_unpack(planes.data_ptr(), _p(stats), y.data_ptr(), n, h, w, _stream()),
          "cine_normunet_unpack")
    return y


def xfyf_pack(img: torch.Tensor, xf: bool, norm: bool = True):
    """reference varnet.py:202-217 + NormUnet front halves (norm=True), or cinenet.py:181-195 (norm=False:
    plain unpadded planes, stats None).  img (b,t,h,w,2)."""
    _pair(img)
    img = _dev(img, "image")
    b, t, h, w, _ = img.shape
    dev, dt = img.device, img.dtype
    pd = pad16 if norm else (lambda v: v)
    if pd(w) == pd(h):
        # one allocation so equal-sized x-f / y-f plane sets can go through the U-Net launches together
        joint = torch.empty((b * h + b * w, 2, pd(w), pd(t)), device=dev, dtype=dt)
        pxf, pyf = joint[:b * h], joint[b * h:]
    else:
        pxf = torch.empty((b * h, 2, pd(w), pd(t)), device=dev, dtype=dt)
        pyf = torch.empty((b * w, 2, pd(h), pd(t)), device=dev, dtype=dt)
    sxf = torch.empty((b * h, 2, 2), device=dev, dtype=dt) if norm else None
    syf = torch.empty((b * w, 2, 2), device=dev, dtype=dt) if norm else None
    mean = torch.empty((b, h, w, 2), device=dev, dtype=dt)
    nbytes = lib().cine_xfyf_ws_bytes(b, t, h, w)
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    check(lib().cine_xfyf_pack(img.data_ptr(), pxf.data_ptr(), pyf.data_ptr(), _p(sxf), _p(syf),
                               mean.data_ptr(), b, t, h, w, int(xf), int(norm), ws.data_ptr(), nbytes, _stream()),
          "cine_xfyf_pack")
    return pxf, pyf, sxf, syf, mean


def xfyf_unpack(pxf, pyf, sxf, syf, mean, b: int, t: int, h: int, w: int, xf: bool) -> torch.Tensor:
    """reference varnet.py:229-241 + NormUnet back halves -> (b,t,1,h,w,2)."""
    out = torch.empty((b, t, 1, h, w, 2), device=pxf.device, dtype=pxf.dtype)
    check(lib().cine_xfyf_unpack(_dev(pxf, "pxf").data_ptr(), _dev(pyf, "pyf").data_ptr(), _p(sxf),
                                 _p(syf), mean.data_ptr(), out.data_ptr(), b, t, h, w, int(xf), _stream()),
          "cine_xfyf_unpack")
    return out


# ------------------------------------------------------------------ U-Net pieces
def _pack(kind: str, w: torch.Tensor) -> torch.Tensor:
    L = lib()
    w = _dev(w.detach(), f"{kind} weight")
    if kind == "c3":
        cout, cin, kh, kw = w.shape
        if (kh, kw) != (3, 3):
            raise ValueError("pack_conv3x3 expects a (cout, cin, 3, 3) weight")
        out = torch.empty(L.cine_conv3x3_packed_floats(cout, cin), device=w.device, dtype=w.dtype)
        check(L.cine_pack_conv3x3(w.data_ptr(), out.data_ptr(), cout, cin, _stream()), "cine_pack_conv3x3")
    elif kind == "tc":
        cin, cout, kh, kw = w.shape
        if (kh, kw) != (2, 2):
            raise ValueError("pack_tconv2x2 expects a (cin, cout, 2, 2) weight")
        out = torch.empty(L.cine_tconv2x2_packed_floats(cin, cout), device=w.device, dtype=w.dtype)
        check(L.cine_pack_tconv2x2(w.data_ptr(), out.data_ptr(), cin, cout, _stream()), "cine_pack_tconv2x2")
    elif kind == "c27":
        cout, cin = w.shape[:2]
        if tuple(w.shape[2:]) != (3, 3, 3):
            raise ValueError("pack_conv3d expects a (cout, cin, 3, 3, 3) weight")
        out = torch.empty(L.cine_conv3d_packed_floats(cout, cin), device=w.device, dtype=w.dtype)
        check(L.cine_pack_conv3d(w.data_ptr(), out.data_ptr(), cout, cin, _stream()), "cine_pack_conv3d")
    elif kind == "tc3":
        cin, cout = w.shape[:2]
        if tuple(w.shape[2:]) != (2, 2, 2):
            raise ValueError("pack_tconv3d expects a (cin, cout, 2, 2, 2) weight")
        out = torch.empty(L.cine_tconv3d_packed_floats(cin, cout), device=w.device, dtype=w.dtype)
        check(L.cine_pack_tconv3d(w.data_ptr(), out.data_ptr(), cin, cout, _stream()), "cine_pack_tconv3d")
    elif kind == "c1":
        cout, cin = w.shape[0], w.shape[1]
        out = torch.empty(L.cine_conv1x1_packed_floats(cout, cin), device=w.device, dtype=w.dtype)
        check(L.cine_pack_conv1x1(w.data_ptr(), out.data_ptr(), cout, cin, _stream()), "cine_pack_conv1x1")
    elif kind == "c3d":      # input-gradient packings (training): include/cine_hip.h "Training"
        cout, cin = w.shape[:2]
        out = torch.empty(L.cine_conv3x3_dgrad_packed_floats(cout, cin), device=w.device, dtype=w.dtype)
        check(L.cine_pack_conv3x3_dgrad(w.data_ptr(), out.data_ptr(), cout, cin, _stream()), "cine_pack_conv3x3_dgrad")
    elif kind == "tcd":
        cin, cout = w.shape[:2]
        out = torch.empty(L.cine_tconv2x2_dgrad_packed_floats(cin, cout), device=w.device, dtype=w.dtype)
        check(L.cine_pack_tconv2x2_dgrad(w.data_ptr(), out.data_ptr(), cin, cout, _stream()), "cine_pack_tconv2x2_dgrad")
    elif kind == "c1d":
        cout, cin = w.shape[:2]
        out = torch.empty(L.cine_conv1x1_dgrad_packed_floats(cout, cin), device=w.device, dtype=w.dtype)
        check(L.cine_pack_conv1x1_dgrad(w.data_ptr(), out.data_ptr(), cout, cin, _stream()), "cine_pack_conv1x1_dgrad")
    else:
        raise ValueError(kind)
    return out


class _BatchedPacks:
    """Packed copies of a network's 2-D conv weights in PERSISTENT buffers, re-packed by ONE launch when a parameter's version moves (training:
    every optimiser step; per tensor that was 596 launches of 4 us in a cfg-3 step).  ``pointers(items)``: items = (kind, parameter) in pointer-list
    order, kind in _OPS or "raw" (the parameter itself, e.g. a bias) or None (a NULL slot).  The descriptor table goes to the device once per
    (parameter addresses, cache epoch); inside a hipGraph capture only the re-pack launch is enqueued, so a captured training step packs in place.
    Inference keeps its own per-tensor packs (UnetWeights.pointers()): graphs captured from those never see these buffers change."""
    _OPS = {"c3": 0, "tc": 1, "c1": 2, "c3d": 3, "tcd": 4, "c1d": 5}

    def __init__(self):
        self.idkey = self.vkey = None
        self.keep = []                       # superseded buffers stay alive: a captured graph may still re-pack into them

    @classmethod
    def supports(cls, items) -> bool:
        # contiguous parameters only: the descriptor table holds the parameters' OWN addresses (a .contiguous() copy of a channels_last
        # model would be packed once and then go stale); anything else takes the per-tensor packs
        return all(k is None or ((k == "raw" or (k in cls._OPS and p.dim() == 4)) and p.is_contiguous()) for k, p in items)

    def pointers(self, items):
        L = lib()
        live = [(k, p) for k, p in items if k is not None]
        idkey = (_cache_epoch,) + tuple((k, p.data_ptr(), tuple(p.shape)) for k, p in live)
        vkey = tuple(p._version for _, p in live)
        if idkey != self.idkey:
            _no_capture("packed training weights", pack=False)      # the table is built by an eager step (GraphedTrainingStep's warm-up)
            dev = live[0][1].device
            sizes, dims = [], []
            for k, p in live:
                if k == "raw":
                    sizes.append(0); dims.append(None); continue
                a, b = int(p.shape[0]), int(p.shape[1])
                fl = {"c3": L.cine_conv3x3_packed_floats, "tc": L.cine_tconv2x2_packed_floats, "c1": L.cine_conv1x1_packed_floats,
                      "c3d": L.cine_conv3x3_dgrad_packed_floats, "tcd": L.cine_tconv2x2_dgrad_packed_floats, "c1d": L.cine_conv1x1_dgrad_packed_floats}[k](a, b)
                sizes.append((int(fl) + 63) // 64 * 64); dims.append((a, b))
            flat = torch.empty(max(sum(sizes), 1), device=dev, dtype=torch.float32)
            nb = L.cine_pack_desc_bytes()
            host = ctypes.create_string_buffer(nb * max(1, sum(1 for s_ in sizes if s_)))
            ptrs, off, nd = [], 0, 0
            params = []
            for (k, p), sz, dm in zip(live, sizes, dims):
                src = _dev(p.detach(), "weight")
                params.append(src)
                if k == "raw":
                    ptrs.append(src.data_ptr()); continue
                dst = flat.data_ptr() + 4 * off
                check(L.cine_pack_desc(ctypes.addressof(host) + nb * nd, self._OPS[k], src.data_ptr(), dst, dm[0], dm[1]), "cine_pack_desc")
                ptrs.append(dst); off += sz; nd += 1
            desc = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(dev) if nd else None
            out, it = [], iter(ptrs)
            for k, _ in items:
                out.append(None if k is None else next(it))
            self.keep.append((flat, desc))
            self.flat, self.desc, self.nd, self.max_total = flat, desc, nd, max(sizes) if sizes else 0
            self.ptrs = (ctypes.c_void_p * len(out))(*out)
            self.idkey, self.vkey = idkey, None
        if vkey != self.vkey:
            if self.nd:
                check(L.cine_pack_batch(self.desc.data_ptr(), self.nd, self.max_total, _stream()), "cine_pack_batch")
            self.vkey = vkey
        return self.ptrs


def pack_conv3x3(w): return _pack("c3", w)
def pack_tconv2x2(w): return _pack("tc", w)
def pack_conv1x1(w): return _pack("c1", w)


def _np(part: Optional[torch.Tensor]) -> int:
    return 0 if part is None else part.shape[2]


def conv3x3_in(srcs: Sequence, wpacked: torch.Tensor, cout: int, h: int, w: int, want_stats: bool = True,
               wpacked2: Optional[torch.Tensor] = None, set_split: int = 0):
    """srcs: one or two (x, part|None, mode) with x (n, c, hs, ws), part (n, c, np, 3).
    Returns (y, part_y): raw conv output and its partial InstanceNorm statistics."""
    (x0, p0, m0) = srcs[0]
    x0 = _dev(x0, "conv source 0")
    n = x0.shape[0]
    if len(srcs) > 1:
        (x1, p1, m1) = srcs[1]
        x1 = _dev(x1, "conv source 1")
        c1, h1, w1 = x1.shape[1:]
    else:
        x1 = p1 = None; m1 = c1 = h1 = w1 = 0
    y = torch.empty((n, cout, h, w), device=x0.device, dtype=x0.dtype)
    py = None
    if want_stats:
        py = torch.empty((n, cout, lib().cine_conv_stat_partials(cout, h, w, 0), 3), device=x0.device, dtype=x0.dtype)
    check(lib().cine_conv3x3_in(x0.data_ptr(), _p(p0), _np(p0), x0.shape[1], m0, x0.shape[2], x0.shape[3],
                                _p(x1), _p(p1), _np(p1), c1, m1, h1, w1, wpacked.data_ptr(), _p(wpacked2), set_split,
                                y.data_ptr(), _p(py), n, cout, h, w, IN_EPS, lrelu_slope(), _stream()), "cine_conv3x3_in")
    return y, py


def tconv2x2_in(x, part, mode: int, wpacked: torch.Tensor, cout: int):
    x = _dev(x, "tconv source")
    n, cin, h, w = x.shape
    y = torch.empty((n, cout, 2 * h, 2 * w), device=x.device, dtype=x.dtype)
    py = torch.empty((n, cout, lib().cine_conv_stat_partials(cout, h, w, 1), 3), device=x.device, dtype=x.dtype)
    check(lib().cine_tconv2x2_in(x.data_ptr(), _p(part), _np(part), mode, wpacked.data_ptr(), None, 0,
                                 y.data_ptr(), py.data_ptr(), n, cin, cout, h, w, IN_EPS, lrelu_slope(), _stream()),
          "cine_tconv2x2_in")
    return y, py


def conv1x1_bias(x, part, mode: int, wpacked: torch.Tensor, bias: torch.Tensor):
    x = _dev(x, "conv1x1 source"); bias = _dev(bias.detach(), "conv1x1 bias")
    n, cin, h, w = x.shape
    cout = bias.shape[0]
    y = torch.empty((n, cout, h, w), device=x.device, dtype=x.dtype)
    check(lib().cine_conv1x1_bias(x.data_ptr(), _p(part), _np(part), mode, wpacked.data_ptr(), bias.data_ptr(),
                                  None, None, n, y.data_ptr(), n, cin, cout, h, w, IN_EPS, lrelu_slope(), _stream()),
          "cine_conv1x1_bias")
    return y


def instnorm_partials(x: torch.Tensor) -> torch.Tensor:
    """(n, c, ...) -> partial statistics (n, c, 1, 3) = {count, mean, M2}."""
    x = _dev(x, "instnorm source")
    n, c = x.shape[:2]
    pe = x.numel() // (n * c)
    part = torch.empty((n, c, 1, 3), device=x.device, dtype=x.dtype)
    check(lib().cine_instnorm_partials(x.data_ptr(), part.data_ptr(), n * c, pe, _stream()), "cine_instnorm_partials")
    return part


def instnorm_finalize(part: torch.Tensor) -> torch.Tensor:
    """(n, c, np, 3) -> (n, c, 2) = {mean, rstd} (biased variance, eps 1e-5)."""
    n, c, npart, _ = part.shape
    st = torch.empty((n, c, 2), device=part.device, dtype=part.dtype)
    check(lib().cine_instnorm_finalize(_dev(part, "partials").data_ptr(), st.data_ptr(), n * c, npart, IN_EPS, _stream()),
          "cine_instnorm_finalize")
    return st


def instnorm_lrelu_apply(x: torch.Tensor, part: torch.Tensor) -> torch.Tensor:
    x = _dev(x, "instnorm source")
    n, c = x.shape[:2]
    y = torch.empty_like(x)
    check(lib().cine_instnorm_lrelu_apply(x.data_ptr(), part.data_ptr(), part.shape[2], y.data_ptr(), n * c,
                                          x.numel() // (n * c), IN_EPS, lrelu_slope(), _stream()),
          "cine_instnorm_lrelu_apply")
    return y


class UnetWeights:
    """Device-side weight pointers of one (or several) reference ``Unet`` modules in the
    order ``cine_unet2d_forward`` expects; 3x3 weights are repacked once and re-packed
    automatically when a parameter is modified in place (``Tensor._version``)."""

    def __init__(self, unets: Sequence[torch.nn.Module]):
        self.unets = list(unets)
        u0 = self.unets[0]
        self.chans, self.pools = u0.chans, u0.num_pool_layers
        self.in_ch, self.out_ch = u0.in_chans, u0.out_chans
        self._key = None
        self._keep = []
        self._ptrs = None
        self._captured = False
        self._dkey = None
        self._dkeep = []
        self._dptrs = None

    def __reduce__(self):
        # copies / pickles carry the modules only; the packed device buffers and pointer tables are rebuilt on first use
        return (type(self), (self.unets,))

    def _params(self):
        out = []
        for u in self.unets:
            seq = []
            k3, kt = ("c3", "tc") if getattr(u, "dims", 2) == 2 else ("c27", "tc3")
            for blk in list(u.down_sample_layers) + [u.conv]:
                seq += [(k3, blk.layers[0].weight), (k3, blk.layers[4].weight)]
            for i, (tc, uc) in enumerate(zip(u.up_transpose_conv, u.up_conv)):
                last = i == len(u.up_conv) - 1
                blk = uc[0] if last else uc
                seq += [(kt, tc.layers[0].weight), (k3, blk.layers[0].weight), (k3, blk.layers[4].weight)]
            fin = u.up_conv[-1][1]
            seq += [("c1", fin.weight), ("raw", fin.bias)]
            out.append(seq)
        return out

    def pointers(self, train: bool = False):
        """train=True (the training forward): persistent packs re-packed by one launch per optimiser step (_BatchedPacks)."""
        params = self._params()
        if train:
            items = [(k, p) for seq in params for k, p in seq]
            if _BatchedPacks.supports(items):
                if self.__dict__.get("_tp") is None:
                    self._tp = _BatchedPacks()
                return self._tp.pointers(items)
        key = (_cache_epoch,) + tuple((p.data_ptr(), p._version) for seq in params for _, p in seq)
        if key != self._key:
            _no_capture("packed U-Net weights", pack=True)
            # graphs captured earlier still hold the old pointers: keep the superseded packs alive only when a capture has
            # taken place since they were made (a training loop re-packs every step and must not pile them up)
            if self._captured:
                self._old = getattr(self, "_old", []) + [self._keep]
            self._captured = False
            keep, ptrs = [], []
            for seq in params:
                for kind, p in seq:
                    t = _dev(p.detach(), "unet weight") if kind == "raw" else _pack(kind, p)
                    keep.append(t); ptrs.append(t.data_ptr())
            self._keep, self._key = keep, key
            self._ptrs = (ctypes.c_void_p * len(ptrs))(*ptrs)
        if torch.cuda.is_current_stream_capturing():
            self._captured = True
        return self._ptrs

    def release_old(self) -> None:
        """Drop packed weight sets that only earlier-captured graphs may reference (call after destroying those graphs)."""
        self._old = []

    # ---- training (cine_unet2d_backward)
    def training_key(self):
        """Called where a training ``autograd.Function`` is entered: the parameters' (address, version) key -- the backward pass re-packs the
        input-gradient weights from the parameters' CURRENT values and compares this key first (torch would raise 'modified by an inplace
        operation')."""
        return tuple((p.data_ptr(), p._version) for seq in self._params() for _, p in seq)

    def drops(self) -> bool:
        """True when a network of the set applies Dropout in its current mode."""
        return any(u.training and float(getattr(u, "drop_prob", 0.0)) > 0 for u in self.unets)

    def dropout_multipliers(self, n: int, device) -> Optional[torch.Tensor]:
        """Dropout2d / Dropout3d of the reference's ConvBlocks (unet.py:22,40,159-168: behind every LeakyReLU, active when the module is in training
        mode and drop_prob > 0): the multiplier of every (3x3 conv, sample, channel) plane -- 0 with probability p, else 1 / (1 - p) -- in
        cine_unet2d_forward_branches' layout, drawn from torch's generator of the device (``torch.manual_seed`` makes a step reproducible, as it
        does for nn.Dropout2d; the random stream itself differs from ATen's).  None when no network of the set drops anything.  The 3-D U-Net (Dropout3d: whole
        volumes) uses the same layout."""
        ps = [float(getattr(u, "drop_prob", 0.0)) if u.training else 0.0 for u in self.unets]
        fixed = getattr(_act_tls, "dropout", None)
        if fixed is not None:
            return None if fixed is False else _dev(fixed, "dropout multipliers")
        if not any(p > 0 for p in ps):
            return None
        if any(not 0.0 <= p < 1.0 for p in ps):
            raise ValueError("dropout probability has to be in [0, 1)")
        total = lib().cine_unet2d_drop_floats(n, self.chans, self.pools)
        u = torch.rand(total, device=device, dtype=torch.float32)
        if len(set(ps)) == 1:
            p = ps[0]
            return (u >= p).to(torch.float32).mul_(1.0 / (1.0 - p))
        # two networks with different probabilities: rows [0, n / 2) of every (n, ch) block belong to the first
        out = torch.empty_like(u)
        off = 0
        for conv in range(4 * self.pools + 2):
            d = conv // 2 if conv < 2 * (self.pools + 1) else self.pools - 1 - (conv - 2 * (self.pools + 1)) // 2
            ch = self.chans << d
            blk, ob = u[off:off + n * ch].view(n, ch), out[off:off + n * ch].view(n, ch)
            for k, p in enumerate(ps):
                rows = slice(k * n // len(ps), (k + 1) * n // len(ps))
                ob[rows] = (blk[rows] >= p).to(torch.float32) / (1.0 - p)
            off += n * ch
        return out

    def check_training_key(self, key, what: str) -> None:
        if key is not None and key != tuple((p.data_ptr(), p._version) for seq in self._params() for _, p in seq):
            raise RuntimeError(f"{what}: a U-Net parameter was modified between the forward and the backward pass "
                               "(optimizer.step() or an in-place update before loss.backward()); the saved activations belong to the old weights")

    def param_lists(self):
        """Per weight set, the parameters in the order of the pointer lists (bias last)."""
        return [[p for _, p in seq] for seq in self._params()]

    def distinct_params(self):
        seen, out = set(), []
        for seq in self._params():
            for _, p in seq:
                if id(p) not in seen:
                    seen.add(id(p)); out.append(p)
        return out

    def dgrad_pointers(self):
        """Input-gradient packings in the order of ``pointers()`` (NULL in the bias slot); re-packed when a parameter changes."""
        params = self._params()
        items = [(None if k == "raw" else k + "d", p) for seq in params for k, p in seq]
        if _BatchedPacks.supports(items):
            if self.__dict__.get("_tdp") is None:
                self._tdp = _BatchedPacks()
            return self._tdp.pointers(items)
        key = (_cache_epoch,) + tuple((p.data_ptr(), p._version) for seq in params for _, p in seq)
        if key != self._dkey:
            _no_capture("packed U-Net gradient weights", pack=True)
            keep, ptrs = [], []
            for seq in params:
                for kind, p in seq:
                    if kind == "raw":
                        ptrs.append(None); continue
                    if kind in ("c3", "tc"):
                        t = _pack(kind + "d", p)
                    elif kind == "c1":
                        # 2-D: the dgrad packing; 3-D (cine_unet3d_backward): the forward 1x1x1 kernel on the transposed matrix
                        t = _pack("c1d", p) if p.dim() == 4 else _pack("c1", p.detach().reshape(p.shape[0], -1).t().contiguous())
                    elif kind == "c27":      # the forward 3x3x3 kernel on the flipped taps, (cout, cin) transposed
                        t = _pack("c27", p.detach().flip(2, 3, 4).transpose(0, 1).contiguous())
                    elif kind == "tc3":      # (cin, cout, 2, 2, 2) read as the (cin, 8 cout) matrix of a 1x1x1 conv over the space-to-depth view
                        t = _pack("c1", p.detach().reshape(p.shape[0], -1))
                    else:
                        raise CineHipError(f"no input-gradient packing for weight kind {kind}")
                    keep.append(t); ptrs.append(t.data_ptr())
            self._dkeep, self._dkey = keep, key
            self._dptrs = (ctypes.c_void_p * len(ptrs))(*ptrs)
        return self._dptrs


def _branch_count(n: int, nsets: int) -> int:
    """How many branches this thread's setting gives a pass of n planes in nsets weight sets (the largest admissible count <= the setting)."""
    nb = unet_branches()
    if nsets == 1 and not BRANCH_SINGLE_SET:
        return 1
    while nb > 1 and (nb % nsets or n // nb < 1):
        nb //= 2
    return max(nb, 1)


def unet2d_forward(x: torch.Tensor, weights: UnetWeights, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """reference denoisers/unet.py:73-125 on (n, in_ch, h, w) planes."""
    x = _dev(x, "unet input")
    n, cin, h, w = x.shape
    nsets = len(weights.unets)
    if cin != weights.in_ch:
        raise ValueError(f"unet input has {cin} channels, expected {weights.in_ch}")
    nb = _branch_count(n, nsets)
    if nb > 1:
        side = side_streams(x.device, nb - 1)
        need = lib().cine_unet2d_branch_ws_bytes(n, h, w, cin, weights.out_ch, weights.chans, weights.pools, nsets, nb, 0)
        if need == 0:
            raise CineHipError("cine_unet2d_branch_ws_bytes rejected the shape")
        if workspace is None or workspace.numel() < need:
            workspace = torch.empty(need, device=x.device, dtype=torch.uint8)
        y = torch.empty((n, weights.out_ch, h, w), device=x.device, dtype=x.dtype)
        sarr = (ctypes.c_void_p * len(side))(*[s_.cuda_stream for s_ in side])
        check(lib().cine_unet2d_forward_branches(x.data_ptr(), y.data_ptr(), weights.pointers(), nsets, n, h, w, cin, weights.out_ch, weights.chans,
                                                 weights.pools, lrelu_slope(), workspace.data_ptr(), workspace.numel(), _stream(), sarr, len(side), 2 * int(BRANCH_INTERLEAVE), None),
              "cine_unet2d_forward_branches")
        return y
    need = lib().cine_unet2d_ws_bytes(n, h, w, cin, weights.out_ch, weights.chans, weights.pools)
    if need == 0:
        raise CineHipError("cine_unet2d_ws_bytes rejected the shape")
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=x.device, dtype=torch.uint8)
    y = torch.empty((n, weights.out_ch, h, w), device=x.device, dtype=x.dtype)
    check(lib().cine_unet2d_forward(x.data_ptr(), y.data_ptr(), weights.pointers(), nsets, n, h, w, cin,
                                    weights.out_ch, weights.chans, weights.pools, lrelu_slope(), workspace.data_ptr(),
                                    workspace.numel(), _stream()), "cine_unet2d_forward")
    return y


# ------------------------------------------------------------------ MWCNN / XPDNet plumbing
def mwcnn_pad(size: int, n_scales: int):
    """(padded, left, right) of reference utils/padding.py:26-47 for one dimension."""
    l, r = ctypes.c_int(), ctypes.c_int()
    padded = lib().cine_mwcnn_pad(int(size), int(n_scales), ctypes.byref(l), ctypes.byref(r))
    return padded, l.value, r.value


class MwcnnWeights:
    """Packed 3x3 weights of one reference MWCNN in the order cine_mwcnn_forward expects."""

    def __init__(self, net: torch.nn.Module):
        self.net = net
        self._key = None
        self._keep = []
        self._ptrs = None
        self.nf = (ctypes.c_int * net.n_scales)(*net.n_filters_per_scale)
        self.nc = (ctypes.c_int * net.n_scales)(*net.n_convs_per_scale)

    def __reduce__(self):
        return (type(self), (self.net,))

    def _params(self):
        n = self.net
        seq = [("c3", n.first_convs[0].layers[0].weight)]
        for blocks in n.conv_blocks_per_scale:
            for blk in blocks:
                seq.append(("c3", blk.layers[0].weight))
        seq += [("c3", n.first_convs[-1].weight), ("raw", n.first_convs[-1].bias)]
        return seq

    def pointers(self, train: bool = False):
        params = self._params()
        if train and _BatchedPacks.supports(params):
            if self.__dict__.get("_tp") is None:
                self._tp = _BatchedPacks()
            return self._tp.pointers(params)
        key = (_cache_epoch,) + tuple((p.data_ptr(), p._version) for _, p in params)
        if key != self._key:
            _no_capture("packed MWCNN weights", pack=True)
            if getattr(self, "_captured", False):                    # graphs captured earlier still hold the old pointers
                self._old = getattr(self, "_old", []) + [self._keep]
            self._captured = False
            keep = [(_dev(p.detach(), "mwcnn bias") if kind == "raw" else _pack(kind, p)) for kind, p in params]
            self._keep, self._key = keep, key
            self._ptrs = (ctypes.c_void_p * len(keep))(*[t.data_ptr() for t in keep])
        if torch.cuda.is_current_stream_capturing():
            self._captured = True
        return self._ptrs

    def release_old(self) -> None:
        self._old = []

    # ---- training (cine_mwcnn_backward)
    def param_list(self):
        return [p for _, p in self._params()]

    def dgrad_pointers(self):
        params = self._params()
        items = [(None if k == "raw" else "c3d", p) for k, p in params]
        if _BatchedPacks.supports(items):
            if self.__dict__.get("_tdp") is None:
                self._tdp = _BatchedPacks()
            return self._tdp.pointers(items)
        key = (_cache_epoch,) + tuple((p.data_ptr(), p._version) for _, p in params)
        if key != getattr(self, "_dkey", None):
            _no_capture("packed MWCNN gradient weights", pack=True)
            keep, ptrs = [], []
            for kind, p in params:
                if kind == "raw":
                    ptrs.append(None); continue
                t = _pack("c3d", p)
                keep.append(t); ptrs.append(t.data_ptr())
            self._dkeep, self._dkey = keep, key
            self._dptrs = (ctypes.c_void_p * len(ptrs))(*ptrs)
        return self._dptrs


def mwcnn_forward(x: torch.Tensor, w: MwcnnWeights, w2: Optional[MwcnnWeights] = None, split: int = 0) -> torch.Tensor:
    """reference denoisers/mwcnn.py:135-179 on (n, in_ch, h, w), h and w multiples of 2^n_scales.  ``w2`` / ``split``: samples
    [split, n) go through a second network of the same topology in the same launches (XPDNet's x-t / y-t networks)."""
    x = _dev(x, "mwcnn input")
    net = w.net
    n, cin, h, wd = x.shape
    if cin != net.in_chans:
        raise ValueError(f"mwcnn input has {cin} channels, expected {net.in_chans}")
    y = torch.empty((n, net.out_chans, h, wd), device=x.device, dtype=x.dtype)
    two = w2 is not None and w2 is not w
    if unet_branches() > 1 and n >= 2 and (not two or 0 < split < n):
        # the plane sets as two concurrent runs of the one-set launch sequence (same kernels and tiles per plane: bit-identical), the second
        # on a side stream: XPDNet's x-t / y-t networks (xpdnet.py:424-446) are independent until the sum behind them
        cut = int(split) if two else (n + 1) // 2
        side = side_streams(x.device, 1)[0]
        main = torch.cuda.current_stream(x.device)
        parts = []
        for lo, hi, wt in ((0, cut, w), (cut, n, w2 if two else w)):
            need = lib().cine_mwcnn_ws_bytes(hi - lo, h, wd, cin, net.out_chans, net.n_scales, wt.nf, wt.nc, net.first_conv_n_filters)
            parts.append((lo, hi, wt, wt.pointers(), torch.empty(max(need, 1), device=x.device, dtype=torch.uint8)))
        side.wait_stream(main)
        for k, (lo, hi, wt, ptrs, ws) in enumerate(parts):
            with torch.cuda.stream(side if k else main):
                check(lib().cine_mwcnn_forward(x[lo:hi].data_ptr(), y[lo:hi].data_ptr(), ptrs, hi - lo, h, wd, cin, net.out_chans, net.n_scales,
                                               wt.nf, wt.nc, net.n_first_convs, net.first_conv_n_filters, int(net.res), lrelu_slope(),
                                               ws.data_ptr(), ws.numel(), _stream()), "cine_mwcnn_forward")
        main.wait_stream(side)
        return y
    need = lib().cine_mwcnn_ws_bytes(n, h, wd, cin, net.out_chans, net.n_scales, w.nf, w.nc, net.first_conv_n_filters)
    ws = torch.empty(max(need, 1), device=x.device, dtype=torch.uint8)
    if w2 is not None and w2 is not w:
        n2 = w2.net
        if (n2.in_chans, n2.out_chans, n2.n_scales, list(n2.n_filters_per_scale), list(n2.n_convs_per_scale), n2.first_conv_n_filters) != \
                (net.in_chans, net.out_chans, net.n_scales, list(net.n_filters_per_scale), list(net.n_convs_per_scale), net.first_conv_n_filters):
            raise ValueError("mwcnn_forward: the two networks differ in topology")
        check(lib().cine_mwcnn_forward2(x.data_ptr(), y.data_ptr(), w.pointers(), w2.pointers(), int(split), n, h, wd, cin, net.out_chans,
                                        net.n_scales, w.nf, w.nc, net.n_first_convs, net.first_conv_n_filters, int(net.res), lrelu_slope(),
                                        ws.data_ptr(), ws.numel(), _stream()), "cine_mwcnn_forward2")
        return y
    check(lib().cine_mwcnn_forward(x.data_ptr(), y.data_ptr(), w.pointers(), n, h, wd, cin, net.out_chans, net.n_scales,
                                   w.nf, w.nc, net.n_first_convs, net.first_conv_n_filters, int(net.res), lrelu_slope(),
                                   ws.data_ptr(), ws.numel(), _stream()), "cine_mwcnn_forward")
    return y


def xpd_pack(buf: torch.Tensor, extra: torch.Tensor, n_primal: int, n_scales: int, xf: bool):
    """reference xpdnet.py:424-474: buffer (b,t,1,h,w,2n) + backward-op image -> padded x-f / y-f planes + mean."""
    buf = _dev(buf, "image buffer"); extra = _dev(extra, "backward-op image")
    b, t, _, h, w, ch = buf.shape
    if ch != 2 * n_primal:
        raise ValueError("image buffer channel count")
    nc = n_primal + 1
    tp, wp, hp = mwcnn_pad(t, n_scales)[0], mwcnn_pad(w, n_scales)[0], mwcnn_pad(h, n_scales)[0]
    if wp == hp:        # both plane sets have one shape: one buffer, so that the two networks can run in the same launches
        joint = torch.empty((b * h + b * w, 2 * nc, wp, tp), device=buf.device, dtype=buf.dtype)
        pxf, pyf = joint[:b * h], joint[b * h:]
    else:
        pxf = torch.empty((b * h, 2 * nc, wp, tp), device=buf.device, dtype=buf.dtype)
        pyf = torch.empty((b * w, 2 * nc, hp, tp), device=buf.device, dtype=buf.dtype)
    mean = torch.empty((b, h, w, nc, 2), device=buf.device, dtype=buf.dtype)
    nbytes = lib().cine_xpd_ws_bytes(b, t, h, w, n_primal)
    ws = torch.empty(nbytes, device=buf.device, dtype=torch.uint8)
    check(lib().cine_xpd_pack(buf.data_ptr(), extra.data_ptr(), pxf.data_ptr(), pyf.data_ptr(), mean.data_ptr(),
                              b, t, h, w, n_primal, n_scales, int(xf), ws.data_ptr(), nbytes, _stream()), "cine_xpd_pack")
    return pxf, pyf, mean


def xpd_unpack(oxf, oyf, mean, b, t, h, w, n_primal, n_scales, xf) -> torch.Tensor:
    """reference xpdnet.py:485-509 -> new image buffer (b,t,1,h,w,2n)."""
    out = torch.empty((b, t, 1, h, w, 2 * n_primal), device=oxf.device, dtype=oxf.dtype)
    check(lib().cine_xpd_unpack(_dev(oxf, "oxf").data_ptr(), _dev(oyf, "oyf").data_ptr(), mean.data_ptr(), out.data_ptr(),
                                b, t, h, w, n_primal, n_scales, int(xf), _stream()), "cine_xpd_unpack")
    return out


def chanlast_to_planes(x: torch.Tensor, n_scales: int = 0) -> torch.Tensor:
    """(n, h, w, c) -> (n, c, pad(h), pad(w)) zero padded per utils/padding.py (n_scales = 0: no padding)."""
    x = _dev(x, "channel-last tensor")
    n, h, w, c = x.shape
    out = torch.empty((n, c, mwcnn_pad(h, n_scales)[0], mwcnn_pad(w, n_scales)[0]), device=x.device, dtype=x.dtype)
    check(lib().cine_chanlast_to_planes(x.data_ptr(), out.data_ptr(), n, c, h, w, n_scales, _stream()), "cine_chanlast_to_planes")
    return out


def planes_to_chanlast(p: torch.Tensor, h: int, w: int, n_scales: int = 0) -> torch.Tensor:
    p = _dev(p, "planes")
    n, c = p.shape[:2]
    out = torch.empty((n, h, w, c), device=p.device, dtype=p.dtype)
    check(lib().cine_planes_to_chanlast(p.data_ptr(), out.data_ptr(), n, c, h, w, n_scales, _stream()), "cine_planes_to_chanlast")
    return out


def extract_complex(buf: torch.Tensor, c_re: int, c_im: int) -> torch.Tensor:
    """(..., C) real buffer -> (..., 2) complex image from channels (c_re, c_im) (reference xpdnet.py:128,161,321-326)."""
    buf = _dev(buf, "buffer")
    out = torch.empty(buf.shape[:-1] + (2,), device=buf.device, dtype=buf.dtype)
    check(lib().cine_extract_complex(buf.data_ptr(), out.data_ptr(), buf.numel() // buf.shape[-1], buf.shape[-1], c_re, c_im,
                                     _stream()), "cine_extract_complex")
    return out


def repeat_complex(img: torch.Tensor, n: int) -> torch.Tensor:
    """torch.repeat_interleave(img, n, dim=-1) of a (..., 2) image (reference xpdnet.py:306-307)."""
    img = _dev(img, "image")
    out = torch.empty(img.shape[:-1] + (2 * n,), device=img.device, dtype=img.dtype)
    check(lib().cine_repeat_complex(img.data_ptr(), out.data_ptr(), img.numel() // 2, n, _stream()), "cine_repeat_complex")
    return out


def expand_resid_hybrid(img: torch.Tensor, sens: torch.Tensor, kref: torch.Tensor, mask: torch.Tensor,
                        out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Hybrid-space image of M (A x) - k_ref: XPDNet's K step with the measurement residual
    (reference xpdnet.py:128-131, 295-298), ready for hybrid_reduce (the masked backward operator :161-167)."""
    img = _dev(img, "image"); sens = _dev(sens, "sens_maps"); kref = _dev(kref, "ref_kspace")
    mask = _dev(mask, "mask", torch.uint8)
    b, t, c, h, w, _ = kref.shape
    if out is None:
        out = torch.empty_like(kref)
    check(lib().cine_expand_dc_hybrid(img.data_ptr(), sens.data_ptr(), kref.data_ptr(), mask.data_ptr(), None,
                                      out.data_ptr(), b, t, c, h, w, 2, _stream()), "cine_expand_dc_hybrid")
    return out


# ------------------------------------------------------------------ convolutional-RNN cells
def conv3x3_sum(srcs: Sequence, wpacked: torch.Tensor, bias: Optional[torch.Tensor], cout: int,
                addend: Optional[torch.Tensor] = None, relu: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = [ReLU](conv3x3(cat(srcs)) + bias + addend): a sum of convolutions of different inputs
    (reference recurrent_varnet.py:172-178, 122-134) as ONE convolution over the concatenated inputs;
    `wpacked` = pack_conv3x3(cat(weights, dim=1)).  srcs: one or two (n, c, h, w) tensors, used as is."""
    x0 = _dev(srcs[0], "conv source 0")
    n, _, h, w = x0.shape
    x1 = _dev(srcs[1], "conv source 1") if len(srcs) > 1 else None
    if out is None:
        y = torch.empty((n, cout, h, w), device=x0.device, dtype=x0.dtype)
    else:
        y = _dev(out, "conv output")
        if tuple(y.shape) != (n, cout, h, w):
            raise ValueError("conv3x3_sum: out has the wrong shape")
    check(lib().cine_conv3x3_ex(x0.data_ptr(), None, 0, x0.shape[1], 0, h, w,
                                _p(x1), None, 0, 0 if x1 is None else x1.shape[1], 0, h, w, 0,
                                wpacked.data_ptr(), _p(bias), _p(addend), int(bool(relu) and relu_on()),
                                y.data_ptr(), None, n, cout, h, w, IN_EPS, lrelu_slope(), _stream()), "cine_conv3x3_ex")
    return y


def crnn_step(x: torch.Tensor, w_hh: torch.Tensor, addend: torch.Tensor, out: torch.Tensor,
              accum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = ReLU(conv3x3(x; w_hh) + addend) [and accum += out]: one step of the BCRNN time sweep
    (reference recurrent_varnet.py:241-254); every tensor (n, c, h, w), `out` / `accum` written in place."""
    x = _dev(x, "hidden state"); addend = _dev(addend, "input term")
    n, c, h, w = x.shape
    for t_, name in ((out, "out"), (accum, "accum")):
        if t_ is not None and not (t_.is_cuda and t_.is_contiguous() and t_.dtype == torch.float32 and t_.shape == x.shape):
            raise ValueError(f"crnn_step: {name} must be a contiguous float32 GPU tensor shaped like x")
    check(lib().cine_crnn_step(x.data_ptr(), w_hh.data_ptr(), addend.data_ptr(), out.data_ptr(), _p(accum), n, c, h, w, int(relu_on()), _stream()),
          "cine_crnn_step")
    return out


def crnn_step2(w_hh: torch.Tensor, fwd, bwd=None) -> None:
    """One step of BOTH directions of the BCRNN time sweep in one launch (reference recurrent_varnet.py:241-254: the forward
    and the backward pass over time are independent chains).  fwd / bwd = (x, addend, y, accum, store): y = ReLU(conv3x3(x;
    w_hh) + addend); accum = y when `store` (the first direction to reach that frame) else accum += y.  bwd None: fwd alone."""
    def prep(s):
        x, addend, y, accum, store = s
        x = _dev(x, "hidden state"); addend = _dev(addend, "input term")
        for t_, name in ((y, "y"), (accum, "accum")):
            if t_ is not None and not (t_.is_cuda and t_.is_contiguous() and t_.dtype == torch.float32 and t_.shape == x.shape):
                raise ValueError(f"crnn_step2: {name} must be a contiguous float32 GPU tensor shaped like x")
        return x, addend, y, accum, int(bool(store))
    xf, af, yf, cf, sf = prep(fwd)
    n, c, h, w = xf.shape
    if bwd is None:
        xb = ab = yb = cb = None; sb = 0
    else:
        xb, ab, yb, cb, sb = prep(bwd)
        if xb.shape != xf.shape:
            raise ValueError("crnn_step2: the two directions differ in shape")
    check(lib().cine_crnn_step2(xf.data_ptr(), af.data_ptr(), yf.data_ptr(), _p(cf), sf, _p(xb), _p(ab), _p(yb), _p(cb), sb,
                                w_hh.data_ptr(), n, c, h, w, int(relu_on()), _stream()), "cine_crnn_step2")


BCRNN_SWEEP_IN_C = _os.environ.get("CINE_BCRNN_SWEEP_IN_C", "1") == "1"        # diagnostics (this binding): False = the step-by-step Python loops over cine_crnn_step2


def bcrnn_sweep(P: torch.Tensor, w_hh: torch.Tensor, zero: torch.Tensor, keep: bool = False):
    """Both time sweeps of a BCRNN layer (reference recurrent_varnet.py:236-254) in one C call: P (T, c, h, w) input terms of all frames,
    w_hh = pack_conv3x3(W_h2h), zero (1, c, h, w) zeros.  Returns (out, hf, hb): out = hidden_f + hidden_b; hf / hb every hidden state
    of the two chains with ``keep`` (training), else two-frame scratch."""
    P = _dev(P, "BCRNN input terms"); zero = _dev(zero, "zero state")
    T, c, h, w = P.shape
    out = torch.empty_like(P)
    hshape = (T if keep else 2, c, h, w)
    hf, hb = torch.empty(hshape, device=P.device, dtype=P.dtype), torch.empty(hshape, device=P.device, dtype=P.dtype)
    check(lib().cine_bcrnn_sweep(P.data_ptr(), w_hh.data_ptr(), zero.data_ptr(), hf.data_ptr(), hb.data_ptr(), out.data_ptr(), T, c, h, w,
                                 int(relu_on()), int(keep), _stream()), "cine_bcrnn_sweep")
    return out, hf, hb


def bcrnn_sweep_bwd(gout: torch.Tensor, w_hh_dgrad: torch.Tensor, zero: torch.Tensor, hf: torch.Tensor, hb: torch.Tensor, relu: bool):
    """Back-propagation through time of ``bcrnn_sweep(keep=True)``: (gf, gb, gP), see cine_bcrnn_sweep_bwd."""
    gout = _dev(gout, "BCRNN output gradient")
    T, c, h, w = gout.shape
    gf, gb, gP = torch.empty_like(gout), torch.empty_like(gout), torch.empty_like(gout)
    check(lib().cine_bcrnn_sweep_bwd(gout.data_ptr(), w_hh_dgrad.data_ptr(), zero.data_ptr(), hf.data_ptr(), hb.data_ptr(), gf.data_ptr(), gb.data_ptr(),
                                     gP.data_ptr(), T, c, h, w, int(bool(relu)), _stream()), "cine_bcrnn_sweep_bwd")
    return gf, gb, gP


# ------------------------------------------------------------------ 3-D U-Net path (dynamic_type '3D')
def normunet3d_pack(x: torch.Tensor, norm: bool = True):
    """(n, t, h, w, 2) -> planes (n, 2, Tp, Hp, Wp) [+ stats (n, 2, 2)]; reference norm_unet.py:149-189."""
    x = _dev(x, "normunet3d_pack input")
    n, t, h, w, _ = x.shape
    pd = pad16 if norm else (lambda v: v)
    planes = torch.empty((n, 2, pd(t), pd(h), pd(w)), device=x.device, dtype=x.dtype)
    stats = torch.empty((n, 2, 2), device=x.device, dtype=x.dtype) if norm else None
    check(lib().cine_normunet3d_pack(x.data_ptr(), planes.data_ptr(), _p(stats), n, t, h, w, int(norm), _stream()),
          "cine_normunet3d_pack")
    return planes, stats


def normunet3d_unpack(planes: torch.Tensor, stats: Optional[torch.Tensor], t: int, h: int, w: int) -> torch.Tensor:
    planes = _dev(planes, "planes")
    n = planes.shape[0]
    y = torch.empty((n, t, h, w, 2), device=planes.device, dtype=planes.dtype)
    check(lib().cine_normunet3d_unpack(planes.data_ptr(), _p(stats), y.data_ptr(), n, t, h, w, _stream()),
          "cine_normunet3d_unpack")
    return y


def unet3d_forward(x: torch.Tensor, weights: UnetWeights) -> torch.Tensor:
    """reference denoisers/unet.py:73-125 with dims = 3 on (n, in_ch, d, h, w) volumes."""
    x = _dev(x, "unet3d input")
    n, cin, d, h, w = x.shape
    if len(weights.unets) != 1:
        raise ValueError("unet3d_forward takes one weight set")
    need = lib().cine_unet3d_ws_bytes(n, d, h, w, cin, weights.out_ch, weights.chans, weights.pools)
    if need == 0:
        raise CineHipError("cine_unet3d_ws_bytes rejected the shape")
    ws = torch.empty(need, device=x.device, dtype=torch.uint8)
    y = torch.empty((n, weights.out_ch, d, h, w), device=x.device, dtype=x.dtype)
    check(lib().cine_unet3d_forward(x.data_ptr(), y.data_ptr(), weights.pointers(), n, d, h, w, cin, weights.out_ch,
                                    weights.chans, weights.pools, lrelu_slope(), ws.data_ptr(), ws.numel(), _stream()), "cine_unet3d_forward")
    return y


def conv3d_in(x: torch.Tensor, weight: torch.Tensor, part: Optional[torch.Tensor] = None, mode: int = 0):
    """Conv3d(3x3x3, pad 1, no bias) of a volume (n, c, d, h, w) -- raw (mode 0) or InstanceNorm3d + LeakyReLU'd on load from its statistics
    records (mode 1) -- returning the RAW output and its records (n, cout, np, 3) (reference unet.py:149-157 with dims = 3: one half of a ConvBlock)."""
    x = _dev(x, "conv3d input")
    n, cin, d, h, w = x.shape
    cout = weight.shape[0]
    wp = _pack("c27", weight)
    y = torch.empty((n, cout, d, h, w), device=x.device, dtype=x.dtype)
    npart = lib().cine_conv_stat_partials3d(cout, d, h, w, 0)
    py = torch.empty((n, cout, npart, 3), device=x.device, dtype=x.dtype)
    check(lib().cine_conv3d_in(x.data_ptr(), _p(part), _np(part), cin, mode, d, h, w, None, None, 0, 0, 0, 0, 0, 0, wp.data_ptr(),
                               None, None, 0, y.data_ptr(), py.data_ptr(), n, cout, d, h, w, IN_EPS, lrelu_slope(), _stream()), "cine_conv3d_in")
    return y, py


def tconv3d_in(x: torch.Tensor, weight: torch.Tensor, part: Optional[torch.Tensor] = None, mode: int = 0):
    """ConvTranspose3d(k 2, s 2, no bias) of (n, cin, d, h, w) -> raw (n, cout, 2d, 2h, 2w) + its statistics records (reference unet.py:204-210, dims = 3)."""
    x = _dev(x, "tconv3d input")
    n, cin, d, h, w = x.shape
    cout = weight.shape[1]
    wp = _pack("tc3", weight)
    y = torch.empty((n, cout, 2 * d, 2 * h, 2 * w), device=x.device, dtype=x.dtype)
    npart = lib().cine_conv_stat_partials3d(cout, d, h, w, 1)
    py = torch.empty((n, cout, npart, 3), device=x.device, dtype=x.dtype)
    check(lib().cine_tconv3d_in(x.data_ptr(), _p(part), _np(part), mode, wp.data_ptr(), y.data_ptr(), py.data_ptr(), n, cin, cout, d, h, w,
                                IN_EPS, lrelu_slope(), _stream()), "cine_tconv3d_in")
    return y, py


def conv3d_bias_relu(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, relu: bool) -> torch.Tensor:
    """Conv3d(3x3x3, 'same') + bias (+ ReLU) on (n, c, d, h, w) (reference denoisers/kspace_net.py:33-46)."""
    x = _dev(x, "conv3d input")
    n, cin, d, h, w = x.shape
    cout = weight.shape[0]
    wp = _pack("c27", weight)
    bias = _dev(bias.detach(), "conv3d bias")
    y = torch.empty((n, cout, d, h, w), device=x.device, dtype=x.dtype)
    check(lib().cine_conv3d_in(x.data_ptr(), None, 0, cin, 0, d, h, w, None, None, 0, 0, 0, 0, 0, 0, wp.data_ptr(),
                               bias.data_ptr(), None, int(bool(relu) and relu_on()), y.data_ptr(), None, n, cout, d, h, w,
                               IN_EPS, lrelu_slope(), _stream()), "cine_conv3d_in")
    return y
