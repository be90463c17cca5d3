"""Host driver of FNOMesh2D, the geo-FNO baseline of the airfoil / pipe experiments
(reference fourierflow/modules/zongyi_fno/mesh_2d.py:14-106; 18 shipped configs under experiments/{airfoil,pipe}/geo-fno*).

    x [B, X, Y, 2] + grid -> fc0 (4 -> width) -> zero-pad X, Y by 8 at the far end -> n_layers x [ rfft2 -> two corner
    blocks (modes1 rows each, modes2 columns) x complex weights -> irfft2,  + 1x1 conv,  GELU except after the last ]
    -> crop -> fc1 (width -> 128) -> GELU -> fc2 (128 -> 1)

Everything is the kernel set of the FNOZongyi2DBlock path (engine_zongyi.py): the spectral convolution is the
dft_fwd -> cdft_rows2 -> mode_mix -> cdft_rows2^-1 -> dft_inv chain with separate row / column mode counts, the pointwise
parts are plin.hip with the exact GELU (the pre-activation of every layer is kept for its derivative), fc0 writes
straight into the padded buffer through the pad map of the F-FNO lift kernel.  width 32 or 64 = the channel tiles of the
spectral kernels (narrower widths are zero-padded like the Zongyi baseline).
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _capi, _lib
from .engine import _p, _View
from .engine_zongyi import HEAD_DIM, ZongyiEngine

PAD = 8            # mesh_2d.py:61
ACT_NONE, ACT_GELU = 0, 2


class GeoFNO2DEngine(ZongyiEngine):
    def __init__(self, *, modes1: int, modes2: int, width: int, n_layers: int, input_dim: int = 4):
        if width > 64:
            raise NotImplementedError("FNOMesh2D: width > 64 is outside the compiled tile set (32 / 64 channels)")
        self.Kx, self.Ky, self.K = modes1, modes2, modes2     # K: modes of the last-axis transform
        self.W, self.Cin, self.L = width, input_dim, n_layers
        self.residual, self.conv_residual = False, True
        self.C = C = 32 if width <= 32 else 64
        self.O = 1
        W = width
        self.param_names, self.param_shapes, self._pad_geom = [], {}, {}

        def add(name, shape, geom):
            self.param_names.append(name)
            self.param_shapes[name] = tuple(shape)
            self._pad_geom[name] = geom      # (R, Cc, inner) of the plain tensor, (Rp, Cp) of its padded twin

        # reference registration order: fc0, convs (weights1, weights2 per layer), ws (weight, bias per layer), fc1, fc2
        add("fc0.weight", (W, input_dim), (W, input_dim, 1, C, input_dim))
        add("fc0.bias", (W,), (1, W, 1, 1, C))
        for l in range(n_layers):
            for j in (1, 2):     # complex [I, O, modes1, modes2] seen through view_as_real
                add(f"convs.{l}.weights{j}", (W, W, modes1, modes2, 2), (W, W, modes1 * modes2 * 2, C, C))
        for l in range(n_layers):
            add(f"ws.{l}.weight", (W, W, 1, 1), (W, W, 1, C, C))       # nn.Conv2d(width, width, 1)
            add(f"ws.{l}.bias", (W,), (1, W, 1, 1, C))
        add("fc1.weight", (HEAD_DIM, W), (HEAD_DIM, W, 1, HEAD_DIM, C))
        add("fc1.bias", (HEAD_DIM,), (1, HEAD_DIM, 1, 1, HEAD_DIM))
        add("fc2.weight", (1, HEAD_DIM), (1, HEAD_DIM, 1, 1, HEAD_DIM))
        add("fc2.bias", (1,), (1, 1, 1, 1, 1))
        self._offsets, off, self._poffsets, poff = {}, 0, {}, 0
        for n in self.param_names:
            self._offsets[n] = off
            off += int(np.prod(self.param_shapes[n]))
            R, Cc, inner, Rp, Cp = self._pad_geom[n]
            self._poffsets[n] = poff
            poff += Rp * Cp * inner
        self.n_params, self.n_padded = off, poff
        self.params = {}
        self.device = None
        self.timer = None
        self._issue_stream = 0
        self.paired_last = False
        self._ws, self._tw = {}, {}
        self._ptr_sig = None
        self._packed = False

    # the planes hold 2 * modes1 * modes2 (ky, kx') modes
    def bind(self, params):
        dev_before = self.device
        super().bind(params)
        if self.device != dev_before:
            f32 = dict(dtype=torch.float32, device=self.device)
            K2 = 2 * self.Kx * self.Ky
            self.planes = [(torch.empty(2 * K2 * self.C * self.C, **f32), torch.empty(2 * K2 * self.C * self.C, **f32))
                           for _ in range(self.L)]

    def _prepare_weights(self, st):
        lib = _lib.get_lib()
        self._refresh_pointers()
        self._packed = True
        if self._n_pad:
            self._k("pad_copy", lib.ffno_pad_copy, _p(self._ptab), self._n_pad, 1, st)
        for l in range(self.L):
            self._k("fw2d_pack", lib.ffno_fw2d_pack2, _p(self._pp(f"convs.{l}.weights1")), _p(self._pp(f"convs.{l}.weights2")),
                    _p(self.planes[l][0]), _p(self.planes[l][1]), self.C, self.Kx, self.Ky, st)

    def _workspace(self, B, X, Y, n_slots=1):
        key = (B, X, Y)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        lib = _lib.get_lib()
        C, L = self.C, self.L
        M, N = X + PAD, Y + PAD
        if 2 * self.Kx > M or self.Ky > N // 2 + 1:
            raise ValueError(f"modes=({self.Kx}, {self.Ky}) do not fit the padded {M} x {N} grid")
        f32 = dict(dtype=torch.float32, device=self.device)
        ws = type("WS", (), {})()
        ws.P_in, ws.P, ws.M, ws.N = B * X * Y, B * M * N, M, N
        ws.padmap = _capi.PadMap((ctypes.c_int32 * 3)(1, X, Y), (ctypes.c_int32 * 3)(1, M, N))
        v = _View(B, M, N, 0, self.Ky, C)
        v.R, v.K2 = B, 2 * self.Kx * self.Ky
        v.spec_y = v.spec
        v.spec = v.K2 * B * 2 * C
        ws.v = v
        P, P_in = ws.P, ws.P_in
        ws.SYa, ws.SYb = torch.empty(v.spec_y, **f32), torch.empty(v.spec_y, **f32)
        ws.SY, ws.SD = torch.empty(v.spec, **f32), torch.empty(v.spec, **f32)
        ws.CW = torch.empty(int(lib.ffno_cdft_rows_ws_floats(B, C, self.Kx, self.Ky)), **f32)      # first-axis DFT scratch
        ws.S = torch.empty(P, C, **f32)
        ws.x = torch.empty(P_in, self.Cin, **f32)
        ws.X = torch.zeros(L + 1, P, C, **f32)           # X[0]: fc0 output in the padded frame (pad stays 0 for ever)
        ws.PRE = torch.empty(L, P, C, **f32)             # pre-activations (GELU derivative); last layer: unused
        ws.SX = torch.empty(L, v.spec, **f32)
        ws.XC = torch.empty(P_in, C, **f32)              # cropped X[L]
        ws.H, ws.HPRE = torch.empty(P_in, HEAD_DIM, **f32), torch.empty(P_in, HEAD_DIM, **f32)
        ws.DH, ws.GC = torch.empty(P_in, HEAD_DIM, **f32), torch.empty(P_in, C, **f32)
        ws.G = [torch.zeros(P, C, **f32) for _ in range(2)]
        ws.DP = torch.empty(P, C, **f32)
        ws.fwpart = torch.empty(2 * v.K2 * C * C, **f32)
        ws.part = torch.empty(int(lib.ffno_plin_wgrad_partial_floats(P, C, HEAD_DIM)), **f32)
        ws.nsplit_lift = max(1, min(1024, (P_in + 127) // 128))
        ws.liftpart = torch.empty(ws.nsplit_lift * C * (self.Cin + 1), **f32)
        ws.live = False
        self._ws[key] = ws
        while len(self._ws) > 3:
            self._ws.pop(next(iter(self._ws)))
        return ws

    def _spectral(self, ws, src, dst, save, planes, fwd: bool, accumulate: int, st, resid=None):
        """dst (+)= irfft2(corner-mix(rfft2(src))) on the padded grid (mesh_2d.py:38-53), or its adjoint."""
        lib = _lib.get_lib()
        v, C = ws.v, self.C
        tw = self._twiddle(v.L)
        ck_f, ck_i, conj = (0, 1, 0) if fwd else (1, 0, 1)
        self._k("dft_fwd", lib.ffno_dft_fwd, _p(src), _p(ws.SYa), _p(tw), v.Bv, v.Mv, v.Nv, C, self.Ky, 0, ck_f, st)
        twm = self._twiddle(v.Mv)
        self._k("cdft_rows", lib.ffno_cdft_rows_mfma, _p(ws.SYa), _p(save), _p(ws.CW), _p(twm), v.Bv, v.Mv, C, self.Kx, self.Ky,
                0, st)
        self._k("mode_mix", lib.ffno_mode_mix, _p(save), _p(planes), _p(ws.SY), v.Bv, C, v.K2, conj, st)
        self._k("cdft_rows", lib.ffno_cdft_rows_mfma, _p(ws.SY), _p(ws.SYb), _p(ws.CW), _p(twm), v.Bv, v.Mv, C, self.Kx, self.Ky,
                1, st)
        self._k("dft_inv", lib.ffno_dft_inv, _p(ws.SYb), _p(dst), _p(resid), _p(tw), v.Bv, v.Mv, v.Nv, C, self.Ky, 0, ck_i,
                accumulate, st)

    # ------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, save_for_backward: bool) -> torch.Tensor:
        """x [B, X, Y, input_dim] (mesh coordinates + grid channels) -> [B, X, Y, 1]."""
        _lib.require_device_tensor(x, "x")
        if x.dim() != 4 or x.shape[-1] != self.Cin:
            raise ValueError(f"expected x of shape [B, X, Y, {self.Cin}], got {tuple(x.shape)}")
        if not self.params:
            raise RuntimeError("bind() the parameters first")
        lib = _lib.get_lib()
        B, X, Y, _ = x.shape
        ws = self._workspace(B, X, Y)
        st = _lib.current_stream(self.device)
        self._issue_stream = st
        C, L, P, P_in, Cin = self.C, self.L, ws.P, ws.P_in, self.Cin
        self._prepare_weights(st)
        pp = self._pp
        pm = ctypes.byref(ws.padmap)
        ws.x.copy_(x.reshape(P_in, Cin))
        self._k("fc0", lib.ffno_lift_fwd, _p(ws.x), _p(pp("fc0.weight")), _p(pp("fc0.bias")), _p(ws.X[0]), P_in, Cin, C, pm, st)
        for l in range(L):
            last = l == L - 1
            self._spectral(ws, ws.X[l], ws.S, ws.SX[l], self.planes[l][0], True, 0, st)
            # x = conv(x) + w(x), GELU except after the last layer (mesh_2d.py:88-93)
            self._k("layer_conv1x1", lib.ffno_plin_fwd, _p(ws.X[l]), C, _p(pp(f"ws.{l}.weight")), _p(pp(f"ws.{l}.bias")),
                    _p(ws.S), _p(ws.X[l + 1]), C, None, None, None if last else _p(ws.PRE[l]), P, C, C,
                    ACT_NONE if last else ACT_GELU, st)
        ws.XC.copy_(ws.X[L].view(B, ws.M, ws.N, C)[:, :X, :Y].reshape(P_in, C))      # crop (mesh_2d.py:95)
        self._k("fc1", lib.ffno_plin_fwd, _p(ws.XC), C, _p(pp("fc1.weight")), _p(pp("fc1.bias")), None, _p(ws.H), HEAD_DIM, None,
                None, _p(ws.HPRE), P_in, C, HEAD_DIM, ACT_GELU, st)
        y = torch.empty(P_in, 1, dtype=torch.float32, device=self.device)
        self._k("fc2", lib.ffno_plin_fwd, _p(ws.H), HEAD_DIM, _p(pp("fc2.weight")), _p(pp("fc2.bias")), None, _p(y), 1, None, None,
                None, P_in, HEAD_DIM, 1, ACT_NONE, st)
        ws.live = bool(save_for_backward)
        self._saved = (B, X, Y) if save_for_backward else None
        return y.view(B, X, Y, 1)

    def backward(self, gy: torch.Tensor) -> torch.Tensor:
        """gy = dL/dout [B, X, Y, 1] -> the flat gradient buffer (``param_names`` order)."""
        _lib.require_device_tensor(gy, "gy")
        if getattr(self, "_saved", None) is None:
            raise RuntimeError("backward() needs a preceding forward(save_for_backward=True)")
        lib = _lib.get_lib()
        B, X, Y = self._saved
        self._saved = None
        ws = self._workspace(B, X, Y)
        st = _lib.current_stream(self.device)
        self._issue_stream = st
        C, L, P, P_in, Cin = self.C, self.L, ws.P, ws.P_in, self.Cin
        pp = self._pp
        gp = lambda n: self._pp(n, self.gpad)      # noqa: E731
        gy = gy.contiguous().view(P_in, 1)
        self._k("fc2_bwd_w", lib.ffno_plin_bwd_weights, _p(gy), 1, None, _p(ws.H), HEAD_DIM, _p(ws.part), _p(gp("fc2.weight")),
                _p(gp("fc2.bias")), P_in, HEAD_DIM, 1, 0, ACT_NONE, st)
        self._k("fc2_bwd", lib.ffno_plin_bwd_data, _p(gy), 1, None, _p(pp("fc2.weight")), _p(ws.DH), HEAD_DIM, None, P_in,
                HEAD_DIM, 1, 0, ACT_NONE, st)
        self._k("fc1_bwd_w", lib.ffno_plin_bwd_weights, _p(ws.DH), HEAD_DIM, _p(ws.HPRE), _p(ws.XC), C, _p(ws.part),
                _p(gp("fc1.weight")), _p(gp("fc1.bias")), P_in, C, HEAD_DIM, 0, ACT_GELU, st)
        self._k("fc1_bwd", lib.ffno_plin_bwd_data, _p(ws.DH), HEAD_DIM, _p(ws.HPRE), _p(pp("fc1.weight")), _p(ws.GC), C, None,
                P_in, C, HEAD_DIM, 0, ACT_GELU, st)
        g = ws.G[0]
        g.zero_()                                                    # adjoint of the crop: zeros in the pad frame
        g.view(B, ws.M, ws.N, C)[:, :X, :Y].copy_(ws.GC.view(B, X, Y, C))
        cur = 0
        v = ws.v
        for l in range(L - 1, -1, -1):
            last = l == L - 1
            act, mode = (None, ACT_NONE) if last else (ws.PRE[l], ACT_GELU)
            gn = ws.G[1 - cur]
            self._k("layer_conv1x1_bwd_w", lib.ffno_plin_bwd_weights, _p(g), C, _p(act), _p(ws.X[l]), C, _p(ws.part),
                    _p(gp(f"ws.{l}.weight")), _p(gp(f"ws.{l}.bias")), P, C, C, 0, mode, st)
            self._k("layer_conv1x1_bwd", lib.ffno_plin_bwd_data, _p(g), C, _p(act), _p(pp(f"ws.{l}.weight")), _p(gn), C,
                    _p(ws.DP), P, C, C, 0, mode, st)
            self._spectral(ws, ws.DP, gn, ws.SD, self.planes[l][1], False, 1, st)
            self._k("fw_grad_partial", lib.ffno_fw_grad_partial, _p(ws.SX[l]), _p(ws.SD), _p(ws.fwpart), v.R, C, v.K2, 1, 0, 1,
                    v.spec, v.spec, st)
            self._k("fw2d_grad_reduce", lib.ffno_fw2d_grad_reduce2, _p(ws.fwpart), _p(gp(f"convs.{l}.weights1")),
                    _p(gp(f"convs.{l}.weights2")), C, self.Kx, self.Ky, 1, 0, st)
            g, cur = gn, 1 - cur
        self._k("fc0_bwd", lib.ffno_lift_bwd, _p(ws.x), _p(g), _p(ws.liftpart), _p(gp("fc0.weight")), _p(gp("fc0.bias")),
                P_in, Cin, C, ws.nsplit_lift, 0, ctypes.byref(ws.padmap), st)
        if self._n_pad:
            self._k("pad_copy(grads)", lib.ffno_pad_copy, _p(self._gtab), self._n_pad, 0, st)
        return self.gflat
