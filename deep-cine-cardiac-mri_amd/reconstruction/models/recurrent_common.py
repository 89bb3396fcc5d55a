"""Shared body of the three convolutional-RNN hybrids on the MI355X conv kernel.

The reference repeats ``CRNNcell`` / ``BCRNNlayer`` verbatim in recurrent_varnet.py:153-259,
recurrent_cinenet.py and recurrent_xpdnet.py; here they are stated once (parameter holders with the same
attribute names) and re-exported by the three model files.

A cell update h = ReLU(i2h(x_t) + h2h(h) + ih2ih(h_iter_t)) (:172-178) is evaluated as
  P_t = conv3x3([h_iter_t, x_t]; [W_ih2ih | W_i2h]) + (b_i2h + b_h2h + b_ih2ih)     all frames in one launch
  h_t = ReLU(conv3x3(h_{t-1}; W_h2h) + P_t)                                           the serial chain
so only the genuinely sequential h2h convolution sits on the critical path; every "conv_x(a) + conv_h(b)"
pair of the body (:122-134) is one convolution over the concatenated inputs.  Sums, biases and ReLU ride in
the MFMA kernel's epilogue.
"""
import torch
from torch import nn

from cine_hip import autograd as ag
from cine_hip import ops


_zeros = {}


def zeros_ro(shape, like: torch.Tensor) -> torch.Tensor:
    """A READ-ONLY zero tensor, shared by every caller on the device (initial hidden states, recurrent_varnet.py:236, 118-121): no
    fill kernel per forward / per replayed graph.  Created outside hipGraph capture (the first eager forward); inside a capture
    with nothing cached yet it falls back to a fresh tensor."""
    key = (like.device, like.dtype, tuple(shape))
    z = _zeros.get(key)
    if z is None:
        z = torch.zeros(shape, device=like.device, dtype=like.dtype)
        if not torch.cuda.is_current_stream_capturing():
            _zeros[key] = z
    return z


class CRNNcell(nn.Module):
    def __init__(self, input_size: int, hidden_size: int, kernel_size: int):
        super().__init__()
        self.i2h = nn.Conv2d(input_size, hidden_size, kernel_size, padding=kernel_size // 2)
        self.h2h = nn.Conv2d(hidden_size, hidden_size, kernel_size, padding=kernel_size // 2)
        self.ih2ih = nn.Conv2d(hidden_size, hidden_size, kernel_size, padding=kernel_size // 2)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, input, hidden_iteration, hidden):
        c = self.h2h.out_channels
        w = ops.pack_conv3x3(torch.cat([self.ih2ih.weight, self.i2h.weight], dim=1))
        p = ops.conv3x3_sum([hidden_iteration, input], w, (self.i2h.bias + self.h2h.bias + self.ih2ih.bias).detach(), c)
        return ops.conv3x3_sum([hidden], ops.pack_conv3x3(self.h2h.weight), None, c, addend=p, relu=True)


class BCRNNlayer(nn.Module):
    def __init__(self, input_size: int, hidden_size: int, kernel_size: int):
        super().__init__()
        self.hidden_size = hidden_size
        self.CRNN_model = CRNNcell(input_size, hidden_size, kernel_size)
        self._key = None

    def _packed(self):
        cell = self.CRNN_model
        params = (cell.i2h.weight, cell.h2h.weight, cell.ih2ih.weight, cell.i2h.bias, cell.h2h.bias, cell.ih2ih.bias)
        key = (ops.cache_epoch(),) + tuple((p.data_ptr(), p._version) for p in params)
        if key != self._key:
            ops._no_capture("packed CRNN weights", pack=True)
            self._w_in = ops.pack_conv3x3(torch.cat([cell.ih2ih.weight, cell.i2h.weight], dim=1))
            self._w_hh = ops.pack_conv3x3(cell.h2h.weight)
            self._bias = (cell.i2h.bias + cell.h2h.bias + cell.ih2ih.bias).detach().contiguous()
            self._key = key
        return self._w_in, self._w_hh, self._bias

    def forward(self, input: torch.Tensor, hidden_iteration: torch.Tensor) -> torch.Tensor:
        """input (t, b, ch, h, w), hidden_iteration (t, b, hidden, h, w) -> (t, b, hidden, h, w)."""
        t, b, ch, h, w = input.shape
        c = self.hidden_size
        w_in, w_hh, bias = self._packed()
        p = ops.conv3x3_sum([hidden_iteration.reshape(t * b, c, h, w), input.reshape(t * b, ch, h, w)], w_in, bias, c)
        p = p.view(t, b, c, h, w)
        zero = zeros_ro((b, c, h, w), p)                                       # hid_init (:236)
        if b == 1 and ops.BCRNN_SWEEP_IN_C:                                     # both loops over the frames in one C call (cine_bcrnn_sweep)
            return ops.bcrnn_sweep(p.view(t, c, h, w), w_hh, zero)[0].view(t, b, c, h, w)
        out = torch.empty_like(p)
        # The forward pass over time (:241-245) and the backward pass (:247-252, same cell) are independent chains; only their
        # sum couples them (:254).  Step s advances both in ONE launch: frame s of the forward chain, frame t-1-s of the
        # backward chain.  The first direction to reach a frame stores its hidden state into `out`, the second adds to it.
        hf, hb = (torch.empty_like(zero), torch.empty_like(zero)), (torch.empty_like(zero), torch.empty_like(zero))
        hid_f = hid_b = zero
        for s in range(t):
            i_f, i_b = s, t - 1 - s
            first = i_f < i_b                # before the chains cross, each of them is the first to reach its frame
            fwd = (hid_f, p[i_f], hf[s & 1], out[i_f], first)
            bwd = (hid_b, p[i_b], hb[s & 1], out[i_b], first)
            if i_f == i_b:                                                      # the middle frame of an odd t: one after the other
                ops.crnn_step2(w_hh, (hid_f, p[i_f], hf[s & 1], out[i_f], True))
                ops.crnn_step2(w_hh, (hid_b, p[i_b], hb[s & 1], out[i_b], False))
            else:
                ops.crnn_step2(w_hh, fwd, bwd)
            hid_f, hid_b = hf[s & 1], hb[s & 1]
        return out


_ones = {}


def _one(like: torch.Tensor) -> torch.Tensor:
    t = _ones.get(like.device)
    if t is None:
        t = _ones[like.device] = torch.ones(1, device=like.device, dtype=torch.float32)
    return t


class CRNNBody(nn.Module):
    """BCRNN + three (conv_x + conv_h -> ReLU) layers + conv4 (reference recurrent_varnet.py:48-63, 116-136)."""

    def _make_body(self, in_ch: int, chans: int, out_ch: int):
        self.bcrnn = BCRNNlayer(input_size=in_ch, hidden_size=chans, kernel_size=3)
        for k in (1, 2, 3):
            setattr(self, f"conv{k}_x", nn.Conv2d(chans, chans, 3, padding=3 // 2))
            setattr(self, f"conv{k}_h", nn.Conv2d(chans, chans, 3, padding=3 // 2))
        self.conv4_x = nn.Conv2d(chans, out_ch, 3, padding=3 // 2)
        self.relu = nn.ReLU(inplace=True)
        self._body_key = None

    def _body_packed(self):
        convs = [getattr(self, f"conv{k}_{s}") for k in (1, 2, 3) for s in ("x", "h")] + [self.conv4_x]
        key = (ops.cache_epoch(),) + tuple((p.data_ptr(), p._version) for c in convs for p in (c.weight, c.bias))
        if key != self._body_key:
            ops._no_capture("packed CRNN weights", pack=True)
            self._pairs = []
            for k in (1, 2, 3):
                cx, chh = getattr(self, f"conv{k}_x"), getattr(self, f"conv{k}_h")
                self._pairs.append((ops.pack_conv3x3(torch.cat([cx.weight, chh.weight], dim=1)),
                                    (cx.bias + chh.bias).detach().contiguous()))
            self._w4 = ops.pack_conv3x3(self.conv4_x.weight)
            self._b4 = self.conv4_x.bias.detach().contiguous()
            self._body_key = key
        return self._pairs, self._w4, self._b4

    def zero_state(self, t: int, b: int, h: int, w: int, like: torch.Tensor):
        return [zeros_ro((t * b, self.chans, h, w), like) for _ in range(4)]     # read only: sources of the first cascade

    def _body_params(self):
        cell = self.bcrnn.CRNN_model
        ps = [cell.ih2ih.weight, cell.i2h.weight, cell.h2h.weight, cell.i2h.bias, cell.h2h.bias, cell.ih2ih.bias]
        for k in (1, 2, 3):
            cx, chh = getattr(self, f"conv{k}_x"), getattr(self, f"conv{k}_h")
            ps += [cx.weight, chh.weight, cx.bias, chh.bias]
        return ps + [self.conv4_x.weight, self.conv4_x.bias]

    def _train_packs(self):
        """The body's packed weights for ag.CrnnBodyFn, once per optimiser step (the cascades share them): forward packings of the concatenated
        weights ([W_ih2ih | W_i2h], [W_kx | W_kh]) with their summed biases, and the input-gradient packing of every single weight."""
        ps = self._body_params()
        key = (ops.cache_epoch(),) + tuple((p.data_ptr(), p._version) for p in ps)
        if key != self.__dict__.get("_tp_key"):
            ops._no_capture("packed CRNN training weights", pack=True)
            with torch.no_grad():
                w_ih2ih, w_i2h, w_h2h, b_i2h, b_h2h, b_ih2ih = ps[:6]
                pk = {"in": ops.pack_conv3x3(torch.cat([w_ih2ih, w_i2h], dim=1)), "b_in": (b_i2h + b_h2h + b_ih2ih).contiguous(),
                      "hh": ops.pack_conv3x3(w_h2h), "dhh": ops._pack("c3d", w_h2h), "d_ih2ih": ops._pack("c3d", w_ih2ih), "d_i2h": ops._pack("c3d", w_i2h)}
                for k in (1, 2, 3):
                    wx, wh, bx, bh = ps[6 + 4 * (k - 1): 10 + 4 * (k - 1)]
                    pk[f"p{k}"] = ops.pack_conv3x3(torch.cat([wx, wh], dim=1)); pk[f"b{k}"] = (bx + bh).contiguous()
                    pk[f"d{k}x"] = ops._pack("c3d", wx); pk[f"d{k}h"] = ops._pack("c3d", wh)
                pk["w4"] = ops.pack_conv3x3(ps[18]); pk["b4"] = ps[19].detach().contiguous(); pk["d4"] = ops._pack("c3d", ps[18])
            self.__dict__["_tp"], self.__dict__["_tp_key"] = pk, key
        return self.__dict__["_tp"], ps

    def body_train(self, x: torch.Tensor, state, residual: torch.Tensor):
        """``body`` as an autograd graph (batch 1).  One node per cascade (ag.CrnnBodyFn: the launch sequence of ``body`` forward; backward with
        the weight gradients on a side stream); with ag.CRNN_BODY_FN off, one node per layer: the BCRNN layer is one Function (back-propagation
        through time inside), every "conv_x(a) + conv_h(b) -> ReLU" pair one ConvSumFn on the concatenated weights; torch concatenates / adds the parameters."""
        t, b, ch, h, w = x.shape
        if ag.CRNN_BODY_FN and b == 1:
            packs, ps = self._train_packs()
            out, *feats = ag.CrnnBodyFn.apply(x.reshape(t, ch, h, w), state[0], state[1], state[2], state[3], residual, packs, *ps)
            return out, list(feats)
        cell = self.bcrnn.CRNN_model
        x0 = ag.BcrnnFn.apply(x.reshape(t * b, ch, h, w), state[0], torch.cat([cell.ih2ih.weight, cell.i2h.weight], dim=1), cell.h2h.weight,
                              cell.i2h.bias + cell.h2h.bias + cell.ih2ih.bias)
        feats = [x0]
        cur = x0
        for k in (1, 2, 3):
            cx, chh = getattr(self, f"conv{k}_x"), getattr(self, f"conv{k}_h")
            cur = ag.ConvSumFn.apply(cur, state[k], torch.cat([cx.weight, chh.weight], dim=1), cx.bias + chh.bias, None, True)
            feats.append(cur)
        out = ag.ConvSumFn.apply(cur, None, self.conv4_x.weight, self.conv4_x.bias, residual, False)
        return out, feats

    def body(self, x: torch.Tensor, state, residual: torch.Tensor):
        """x (t, b, ch, h, w); state [x0..x3] of the previous cascade; returns (residual + conv4(x3), new state)."""
        t, b, _, h, w = x.shape
        c = self.chans
        pairs, w4, b4 = self._body_packed()
        x0 = self.bcrnn(x, state[0].view(t, b, c, h, w)).view(t * b, c, h, w)
        feats = [x0]
        cur = x0
        for k in range(3):
            cur = ops.conv3x3_sum([cur, state[k + 1]], pairs[k][0], pairs[k][1], c, relu=True)
            feats.append(cur)
        out = ops.conv3x3_sum([cur], w4, b4, self.conv4_x.out_channels, addend=residual)
        return out, feats
