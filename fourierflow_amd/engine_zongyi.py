"""Host driver of the FNOZongyi2DBlock baseline (BASELINE config 0).

Reference: fourierflow/modules/zongyi_fno/grid_2d.py:16-78 (SpectralConv2d: rfft2 -> two K x K corner blocks ->
irfft2, plus a pointwise linear, ReLU) and :81-129 (FNOZongyi2DBlock: in_proj, n_layers of those, a
Linear-ReLU-Linear head).  The spectral convolution is the same operator as FNOPlus2DBlock's, so it runs on the same
kernels (ffno_dft_fwd -> ffno_cdft_rows -> ffno_mode_mix -> ffno_cdft_rows(inverse) -> ffno_dft_inv); everything
pointwise runs on csrc/plin.hip.  The baseline is 20 channels wide: activations and weights live in buffers padded
to the 32- (or 64-) channel tiles of the spectral kernels with exact zeros in the pad, one ffno_pad_copy launch
moves all parameters into their padded twins before a pass and all gradients back after it.

Unlike the F-FNO engine this one returns the INPUT gradient as well and keeps several forward passes alive at once
(``slot``): Grid2DRolloutExperiment back-propagates through a 10-step autoregressive rollout
(routines/grid_2d_rollout.py:104-134).
"""
from __future__ import annotations

import ctypes
from typing import Dict

import numpy as np
import torch

from . import _capi, _lib
from .engine import _p, _View

HEAD_DIM = 128


class ZongyiEngine:
    def __init__(self, *, modes: int, width: int, input_dim: int, n_layers: int, residual: bool = False,
                 conv_residual: bool = True):
        if width > 32:
            raise NotImplementedError("FNOZongyi2DBlock: width > 32 is outside the compiled tile set of the baseline path")
        self.K, self.W, self.Cin, self.L = modes, width, input_dim, n_layers
        # residual: x = layer(x) + x (grid_2d.py:126);  conv_residual: act(spectral(x) + linear(x)), else
        # act(linear(spectral(x))) (grid_2d.py:74-77) -- torus_li/ablation/zongyi_markov_residual uses (True, False)
        self.residual, self.conv_residual = bool(residual), bool(conv_residual)
        self.C = 32            # channel tile of the spectral kernels; channels width..31 are exact zeros
        self.O = 1
        W, K = width, modes
        self.param_names, self.param_shapes = [], {}
        # (R, Cc, inner) of the plain tensor and (Rp, Cp) of its padded twin, per parameter
        self._pad_geom = {}

        def add(name, shape, geom):
            self.param_names.append(name)
            self.param_shapes[name] = tuple(shape)
            self._pad_geom[name] = geom

        C = self.C
        add("in_proj.weight", (W, input_dim), (W, input_dim, 1, C, input_dim))
        add("in_proj.bias", (W,), (1, W, 1, 1, C))
        for l in range(n_layers):
            pre = f"spectral_layers.{l}."
            add(pre + "linear.weight", (W, W), (W, W, 1, C, C))
            add(pre + "linear.bias", (W,), (1, W, 1, 1, C))
            for j in range(2):
                add(pre + f"fourier_weight.{j}", (W, W, K, K, 2), (W, W, K * K * 2, C, C))
        add("feedforward.0.weight", (HEAD_DIM, W), (HEAD_DIM, W, 1, HEAD_DIM, C))
        add("feedforward.0.bias", (HEAD_DIM,), (1, HEAD_DIM, 1, 1, HEAD_DIM))
        add("feedforward.2.weight", (1, HEAD_DIM), (1, HEAD_DIM, 1, 1, HEAD_DIM))
        add("feedforward.2.bias", (1,), (1, 1, 1, 1, 1))
        self._offsets, off = {}, 0
        self._poffsets, poff = {}, 0
        for n in self.param_names:
            self._offsets[n] = off
            off += int(np.prod(self.param_shapes[n]))
            R, Cc, inner, Rp, Cp = self._pad_geom[n]
            self._poffsets[n] = poff
            poff += Rp * Cp * inner
        self.n_params, self.n_padded = off, poff
        self.params: Dict[str, torch.Tensor] = {}
        self.device = None
        self.timer = None
        self._issue_stream = 0
        self.paired_last = False
        self._ws = {}
        self._tw = {}
        self._ptr_sig = None
        self._packed = False

    # ------------------------------------------------------------------------------------------------
    def _k(self, name, fn, *args):
        t = self.timer
        if t is not None and t.want(name):
            t.start(name, self._issue_stream)
            rc = fn(*args)
            t.stop(name, self._issue_stream)
        else:
            rc = fn(*args)
        if rc != 0:
            _capi.check(rc, name)

    def bind(self, params: Dict[str, torch.Tensor]):
        dev = None
        for n in self.param_names:
            if n not in params:
                raise KeyError(f"missing parameter {n}")
            t = params[n]
            _lib.require_device_tensor(t, n)
            if tuple(t.shape) != self.param_shapes[n]:
                raise ValueError(f"{n}: expected shape {self.param_shapes[n]}, got {tuple(t.shape)}")
            if not t.is_contiguous():
                raise ValueError(f"{n} must be contiguous")
            dev = dev or t.device
            if t.device != dev:
                raise ValueError("all parameters must live on one device")
        sig = tuple(params[n].data_ptr() for n in self.param_names)
        if sig != getattr(self, "_bound_sig", None):
            self._packed = False          # other tensors: the padded twins are stale whatever the caller says
        self._bound_sig = sig
        self.params = {n: params[n] for n in self.param_names}
        if dev != self.device:
            self.device = dev
            f32 = dict(dtype=torch.float32, device=dev)
            self.gflat = torch.zeros(self.n_params, **f32)
            self.ppad = torch.zeros(self.n_padded, **f32)      # padded parameters (pad entries stay 0 forever)
            self.gpad = torch.zeros(self.n_padded, **f32)      # padded gradients
            K2 = 2 * self.K * self.K
            self.planes = [(torch.empty(2 * K2 * self.C * self.C, **f32), torch.empty(2 * K2 * self.C * self.C, **f32))
                           for _ in range(self.L)]
            self._ws, self._tw = {}, {}
        self._ptr_sig = None

    def zero_grad(self):
        """Start a fresh accumulation for backward(..., accumulate=True)."""
        self.gpad.zero_()
        self.gflat.zero_()

    def grad_view(self, name: str) -> torch.Tensor:
        o = self._offsets[name]
        return self.gflat[o:o + int(np.prod(self.param_shapes[name]))].view(self.param_shapes[name])

    def _aliased(self, name) -> bool:
        """The padded twin has the parameter's own shape (width == channel tile): use the tensor itself, no copy."""
        R, Cc, inner, Rp, Cp = self._pad_geom[name]
        return R == Rp and Cc == Cp

    def _pp(self, name, buf=None) -> torch.Tensor:
        R, Cc, inner, Rp, Cp = self._pad_geom[name]
        if self._aliased(name):
            return (self.params[name] if buf is None else self.grad_view(name)).reshape(-1)
        o = self._poffsets[name]
        return (self.ppad if buf is None else buf)[o:o + Rp * Cp * inner]

    def _refresh_pointers(self):
        sig = tuple(self.params[n].data_ptr() for n in self.param_names)
        if sig == self._ptr_sig:
            return
        self._ptr_sig = sig

        def table(plain_of, padded_buf):
            descs = []
            for n in self.param_names:
                if self._aliased(n):
                    continue
                R, Cc, inner, Rp, Cp = self._pad_geom[n]
                descs.append(_capi.PadDesc(plain_of(n).data_ptr(), self._pp(n, padded_buf).data_ptr(), R, Cc, inner, Cp))
            self._n_pad = len(descs)
            if not descs:
                return None
            arr = (_capi.PadDesc * len(descs))(*descs)
            return torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(self.device)

        self._ptab = table(lambda n: self.params[n], self.ppad)
        self._gtab = table(self.grad_view, self.gpad)

    def _twiddle(self, L: int) -> torch.Tensor:
        if L not in self._tw:
            host = np.zeros(2 * L, np.float32)
            _capi.check(_lib.get_lib().ffno_twiddle_fill_host(host.ctypes.data_as(ctypes.c_void_p), L), "twiddle")
            self._tw[L] = torch.from_numpy(host).to(self.device)
        return self._tw[L]

    # ------------------------------------------------------------------------------------------------
    def _workspace(self, B, M, N, n_slots):
        key = (B, M, N)
        ws = self._ws.get(key)
        if ws is not None and len(ws.slots) >= n_slots:
            return ws
        lib = _lib.get_lib()
        C, L = self.C, self.L
        f32 = dict(dtype=torch.float32, device=self.device)
        if ws is None:
            if M != N:
                raise ValueError(f"FNOZongyi2DBlock needs a square grid, got {M} x {N}: the reference's "
                                 "irfft2(s=(N, M)) (grid_2d.py:68) fails on anything else")
            if 2 * self.K > M or self.K > N // 2 + 1:
                raise ValueError(f"modes={self.K} does not fit a {M} x {N} grid (the reference fails in its einsum here)")
            ws = type("WS", (), {})()
            P = B * M * N
            ws.P = P
            v = _View(B, M, N, 0, self.K, C)
            v.R, v.K2 = B, 2 * self.K * self.K
            v.spec_y = v.spec
            v.spec = v.K2 * B * 2 * C
            ws.v = v
            ws.SYa = torch.empty(v.spec_y, **f32)
            ws.SYb = torch.empty(v.spec_y, **f32)
            ws.CW = torch.empty(int(lib.ffno_cdft_rows_ws_floats(B, C, self.K, self.K)), **f32)     # first-axis DFT scratch
            ws.SY = torch.empty(v.spec, **f32)
            ws.SX0 = torch.empty(v.spec, **f32)             # forward spectrum when nothing is saved
            ws.SD = torch.empty(v.spec, **f32)              # adjoint spectrum dY of the current layer
            ws.S = torch.empty(P, C, **f32)                 # spectral branch output of the current layer
            ws.Xa = [torch.empty(P, C, **f32) for _ in range(2)]     # inference ping-pong
            ws.A0 = torch.empty(P, C, **f32) if self.residual else None   # layer output before the block-level residual
            ws.H0 = torch.empty(P, HEAD_DIM, **f32)
            ws.G = [torch.empty(P, C, **f32) for _ in range(2)]
            ws.DP = torch.empty(P, C, **f32)
            ws.DH = torch.empty(P, HEAD_DIM, **f32)
            ws.fwpart = torch.empty(2 * v.K2 * C * C, **f32)
            ws.part = torch.empty(int(lib.ffno_plin_wgrad_partial_floats(P, C, HEAD_DIM)), **f32)
            ws.slots = []
            self._ws[key] = ws
            while len(self._ws) > 3:
                self._ws.pop(next(iter(self._ws)))
        while len(ws.slots) < n_slots:
            s = type("Slot", (), {})()
            s.x = torch.empty(ws.P, self.Cin, **f32)
            s.X = torch.empty(L + 1, ws.P, C, **f32)        # X[0] = in_proj(x), X[l+1] = output of layer l
            s.SX = torch.empty(L, ws.v.spec, **f32)          # forward 2-D spectra (inputs of the weight gradient)
            s.H = torch.empty(ws.P, HEAD_DIM, **f32)
            s.A = torch.empty(L, ws.P, C, **f32) if self.residual else None          # relu outputs (ReLU masks)
            s.S = torch.empty(L, ws.P, C, **f32) if not self.conv_residual else None # spectral outputs (inputs of `linear`)
            s.live = False
            ws.slots.append(s)
        return ws

    def _spectral(self, ws, src, dst, save, planes, fwd: bool, accumulate: int, st, resid=None):
        """dst (+)= irfft2(corner-mix(rfft2(src)))  (or its adjoint), grid_2d.py:48-71."""
        lib = _lib.get_lib()
        v, C = ws.v, self.C
        tw = self._twiddle(v.L)
        ck_f, ck_i, conj = (0, 1, 0) if fwd else (1, 0, 1)
        self._k("dft_fwd", lib.ffno_dft_fwd, _p(src), _p(ws.SYa), _p(tw), v.Bv, v.Mv, v.Nv, C, v.K, 0, ck_f, st)
        twm = self._twiddle(v.Mv)
        self._k("cdft_rows", lib.ffno_cdft_rows_mfma, _p(ws.SYa), _p(save), _p(ws.CW), _p(twm), v.Bv, v.Mv, C, v.K, v.K, 0, st)
        self._k("mode_mix", lib.ffno_mode_mix, _p(save), _p(planes), _p(ws.SY), v.Bv, C, v.K2, conj, st)
        self._k("cdft_rows", lib.ffno_cdft_rows_mfma, _p(ws.SY), _p(ws.SYb), _p(ws.CW), _p(twm), v.Bv, v.Mv, C, v.K, v.K, 1, st)
        self._k("dft_inv", lib.ffno_dft_inv, _p(ws.SYb), _p(dst), _p(resid), _p(tw), v.Bv, v.Mv, v.Nv, C, v.K, 0, ck_i,
                accumulate, st)

    def _prepare_weights(self, st):
        lib = _lib.get_lib()
        self._refresh_pointers()
        self._packed = True
        if self._n_pad:
            self._k("pad_copy", lib.ffno_pad_copy, _p(self._ptab), self._n_pad, 1, st)
        for l in range(self.L):
            pre = f"spectral_layers.{l}."
            self._k("fw2d_pack", lib.ffno_fw2d_pack, _p(self._pp(pre + "fourier_weight.0")),
                    _p(self._pp(pre + "fourier_weight.1")), _p(self.planes[l][0]), _p(self.planes[l][1]), self.C, self.K, st)

    # ------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, save_for_backward: bool, slot: int = 0, n_slots: int = 1,
                weights_ready: bool = False) -> torch.Tensor:
        """x [B, M, N, input_dim] -> forecast [B, M, N, 1].  With ``save_for_backward`` the activations are kept in
        ``slot`` until its backward() ran."""
        _lib.require_device_tensor(x, "x")
        if x.dim() != 4 or x.shape[-1] != self.Cin:
            raise ValueError(f"expected x of shape [B, M, N, {self.Cin}], got {tuple(x.shape)}")
        if not self.params:
            raise RuntimeError("bind() the parameters first")
        lib = _lib.get_lib()
        B, M, N, _ = x.shape
        ws = self._workspace(B, M, N, max(n_slots, slot + 1))
        st = _lib.current_stream(self.device)
        self._issue_stream = st
        C, L, P, Cin = self.C, self.L, ws.P, self.Cin
        if not (weights_ready and self._packed):      # weights_ready: the caller vouches the parameters did not change
            self._prepare_weights(st)
        pp = self._pp
        if save_for_backward:
            sl = ws.slots[slot]
            sl.x.copy_(x.reshape(P, Cin))
            xin, X, H = sl.x, sl.X, sl.H
            sl.live = True
        else:
            xin, X, H = x.contiguous().view(P, Cin), None, ws.H0
        cur = X[0] if X is not None else ws.Xa[0]
        self._k("in_proj", lib.ffno_plin_fwd, _p(xin), Cin, _p(pp("in_proj.weight")), _p(pp("in_proj.bias")), None, _p(cur), C,
                None, None, None, P, Cin, C, 0, st)
        for l in range(L):
            pre = f"spectral_layers.{l}."
            nxt = X[l + 1] if X is not None else ws.Xa[(l + 1) & 1]
            S = sl.S[l] if (save_for_backward and not self.conv_residual) else ws.S
            self._spectral(ws, cur, S, sl.SX[l] if save_for_backward else ws.SX0, self.planes[l][0], True, 0, st)
            # conv_residual: act(spectral(x) + linear(x)), else act(linear(spectral(x)))   (grid_2d.py:74-77);
            # block-level residual: x = layer(x) + x as a second output, the ReLU output itself is kept for the mask (:126)
            A = nxt if not self.residual else (sl.A[l] if save_for_backward else ws.A0)
            src, add = (cur, S) if self.conv_residual else (S, None)
            self._k("layer_linear", lib.ffno_plin_fwd, _p(src), C, _p(pp(pre + "linear.weight")), _p(pp(pre + "linear.bias")),
                    _p(add), _p(A), C, _p(cur) if self.residual else None, _p(nxt) if self.residual else None, None, P, C, C, 1, st)
            cur = nxt
        self._k("head_fc1", lib.ffno_plin_fwd, _p(cur), C, _p(pp("feedforward.0.weight")), _p(pp("feedforward.0.bias")), None,
                _p(H), HEAD_DIM, None, None, None, P, C, HEAD_DIM, 1, st)
        y = torch.empty(P, 1, dtype=torch.float32, device=self.device)
        self._k("head_fc2", lib.ffno_plin_fwd, _p(H), HEAD_DIM, _p(pp("feedforward.2.weight")), _p(pp("feedforward.2.bias")),
                None, _p(y), 1, None, None, None, P, HEAD_DIM, 1, 0, st)
        return y.view(B, M, N, 1)

    # ------------------------------------------------------------------------------------------------
    def backward(self, gy: torch.Tensor, slot: int = 0, need_dx: bool = False, accumulate: bool = False):
        """gy = dL/dforecast [B, M, N, 1] for the pass saved in ``slot``.  Fills (or, with ``accumulate``, adds to) the
        flat gradient buffer and returns ``gflat`` -- or ``(gflat, dx)`` with ``need_dx``."""
        _lib.require_device_tensor(gy, "gy")
        lib = _lib.get_lib()
        B, M, N, _ = gy.shape
        ws = self._ws.get((B, M, N))
        if ws is None or slot >= len(ws.slots) or not ws.slots[slot].live:
            raise RuntimeError("backward() needs a preceding forward(save_for_backward=True) in the same slot")
        sl = ws.slots[slot]
        sl.live = False
        gy = gy.contiguous().view(ws.P, 1)
        st = _lib.current_stream(self.device)
        self._issue_stream = st
        C, L, P, Cin = self.C, self.L, ws.P, self.Cin
        pp = self._pp
        gp = lambda n: self._pp(n, self.gpad)       # noqa: E731
        acc = int(bool(accumulate))
        XL = sl.X[L]
        # head: y = W2 relu(W1 x + b1) + b2
        self._k("head_fc2_bwd_w", lib.ffno_plin_bwd_weights, _p(gy), 1, None, _p(sl.H), HEAD_DIM, _p(ws.part),
                _p(gp("feedforward.2.weight")), _p(gp("feedforward.2.bias")), P, HEAD_DIM, 1, acc, 0, st)
        self._k("head_fc2_bwd", lib.ffno_plin_bwd_data, _p(gy), 1, None, _p(pp("feedforward.2.weight")), _p(ws.DH), HEAD_DIM,
                None, P, HEAD_DIM, 1, 0, 0, st)
        self._k("head_fc1_bwd_w", lib.ffno_plin_bwd_weights, _p(ws.DH), HEAD_DIM, _p(sl.H), _p(XL), C, _p(ws.part),
                _p(gp("feedforward.0.weight")), _p(gp("feedforward.0.bias")), P, C, HEAD_DIM, acc, 1, st)
        g = ws.G[0]
        self._k("head_fc1_bwd", lib.ffno_plin_bwd_data, _p(ws.DH), HEAD_DIM, _p(sl.H), _p(pp("feedforward.0.weight")), _p(g), C,
                None, P, C, HEAD_DIM, 0, 1, st)
        cur = 0
        v = ws.v
        for l in range(L - 1, -1, -1):
            pre = f"spectral_layers.{l}."
            xin = sl.X[l]
            act = sl.A[l] if self.residual else sl.X[l + 1]       # the ReLU output of this layer
            gn = ws.G[1 - cur]
            res = g if self.residual else None                    # x = layer(x) + x: g also flows straight through
            lin_in = xin if self.conv_residual else sl.S[l]
            self._k("layer_linear_bwd_w", lib.ffno_plin_bwd_weights, _p(g), C, _p(act), _p(lin_in), C, _p(ws.part),
                    _p(gp(pre + "linear.weight")), _p(gp(pre + "linear.bias")), P, C, C, acc, 1, st)
            if self.conv_residual:
                # dpre = g * 1[out > 0] feeds both the linear (here) and the spectral adjoint (below)
                self._k("layer_linear_bwd", lib.ffno_plin_bwd_data, _p(g), C, _p(act), _p(pp(pre + "linear.weight")), _p(gn),
                        C, _p(ws.DP), P, C, C, 0, 1, st)
                self._spectral(ws, ws.DP, gn, ws.SD, self.planes[l][1], False, 1, st, resid=res)
            else:
                self._k("layer_linear_bwd", lib.ffno_plin_bwd_data, _p(g), C, _p(act), _p(pp(pre + "linear.weight")),
                        _p(ws.DP), C, None, P, C, C, 0, 1, st)
                self._spectral(ws, ws.DP, gn, ws.SD, self.planes[l][1], False, 0, st, resid=res)
            self._k("fw_grad_partial", lib.ffno_fw_grad_partial, _p(sl.SX[l]), _p(ws.SD), _p(ws.fwpart), v.R, C, v.K2, 1, 0, 1,
                    v.spec, v.spec, st)
            self._k("fw2d_grad_reduce", lib.ffno_fw2d_grad_reduce, _p(ws.fwpart), _p(gp(pre + "fourier_weight.0")),
                    _p(gp(pre + "fourier_weight.1")), C, self.K, 1, acc, st)
            g, cur = gn, 1 - cur
        self._k("in_proj_bwd_w", lib.ffno_plin_bwd_weights, _p(g), C, None, _p(sl.x), Cin, _p(ws.part),
                _p(gp("in_proj.weight")), _p(gp("in_proj.bias")), P, Cin, C, acc, 0, st)
        dx = None
        if need_dx:
            dx = torch.empty(P, Cin, dtype=torch.float32, device=self.device)
            self._k("in_proj_bwd", lib.ffno_plin_bwd_data, _p(g), C, None, _p(pp("in_proj.weight")), _p(dx), Cin, None, P, Cin,
                    C, 0, 0, st)
            dx = dx.view(B, M, N, Cin)
        if self._n_pad:
            self._k("pad_copy(grads)", lib.ffno_pad_copy, _p(self._gtab), self._n_pad, 0, st)
        return (self.gflat, dx) if need_dx else self.gflat
