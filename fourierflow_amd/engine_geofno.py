"""Host driver of FNOMesh2D / FNOMesh3D, the geo-FNO baselines of the airfoil / pipe / plasticity experiments
(reference fourierflow/modules/zongyi_fno/mesh_2d.py:14-106, mesh_3d.py:7-113; 30 shipped configs under
experiments/{airfoil,pipe,plasticity}/geo-fno*).  2-D:

    x [B, X, Y, 2] + grid -> fc0 (4 -> width) -> zero-pad X, Y by 8 at the far end -> n_layers x [ rfft2 -> two corner
    blocks (modes1 rows each, modes2 columns) x complex weights -> irfft2,  + 1x1 conv,  GELU except after the last ]
    -> crop -> fc1 (width -> 128) -> GELU -> fc2 (128 -> 1)

Everything is the kernel set of the FNOZongyi2DBlock path (engine_zongyi.py): the spectral convolution is the
dft_fwd -> cdft_rows2 -> mode_mix -> cdft_rows2^-1 -> dft_inv chain with separate row / column mode counts, the pointwise
parts are plin.hip with the exact GELU (the pre-activation of every layer is kept for its derivative), fc0 writes
straight into the padded buffer through the pad map of the F-FNO lift kernel.  width 32 or 64 = the channel tiles of the
spectral kernels (narrower widths are zero-padded like the Zongyi baseline).

3-D (FNOMesh3D): the same with rfftn over three axes, FOUR corner blocks (+-x, +-y, low z), padding 5, a 1x1x1 convolution and
four output channels: the z transform is ffno_dft_fwd, the y and x transforms two passes of ffno_cdft_rows_mfma (lines
(kz, b, x) then ((kz, ky'), b)), the mix runs over the K3 * 2K2 * 2K1 retained modes with the B samples as rows.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _capi, _lib
from .engine import _p, _View
from .engine_zongyi import HEAD_DIM, ZongyiEngine

ACT_NONE, ACT_GELU = 0, 2


class GeoFNO2DEngine(ZongyiEngine):
    """``modes3 is None``: FNOMesh2D (padding 8, one output channel); else FNOMesh3D (padding 5, four output channels)."""

    def __init__(self, *, modes1: int, modes2: int, width: int, n_layers: int, input_dim: int = 4, modes3=None):
        if width > 64:
            raise NotImplementedError("FNOMesh2D / FNOMesh3D: width > 64 is outside the compiled tile set (32 / 64 channels)")
        self.nd = 2 if modes3 is None else 3
        self.pad = 8 if self.nd == 2 else 5           # mesh_2d.py:61, mesh_3d.py:67
        self.Kx, self.Ky, self.Kz = modes1, modes2, modes3
        self.K = modes2 if self.nd == 2 else modes3   # modes of the last-axis (real) transform
        self.W, self.Cin, self.L = width, input_dim, n_layers
        self.residual, self.conv_residual = False, True
        self.C = C = 32 if width <= 32 else 64
        self.O = 1 if self.nd == 2 else 4             # fc2: 128 -> 1 (mesh_2d.py:76) / 128 -> 4 (mesh_3d.py:84)
        self.nw = 2 if self.nd == 2 else 4            # corner-block weight tensors per layer
        self.Ktot = 2 * modes1 * modes2 * (1 if self.nd == 2 else 2 * modes3)     # retained modes
        W = width
        self.param_names, self.param_shapes, self._pad_geom = [], {}, {}

        def add(name, shape, geom):
            self.param_names.append(name)
            self.param_shapes[name] = tuple(shape)
            self._pad_geom[name] = geom      # (R, Cc, inner) of the plain tensor, (Rp, Cp) of its padded twin

        # reference registration order: fc0, convs (weights1, weights2 per layer), ws (weight, bias per layer), fc1, fc2
        add("fc0.weight", (W, input_dim), (W, input_dim, 1, C, input_dim))
        add("fc0.bias", (W,), (1, W, 1, 1, C))
        mshape = (modes1, modes2) if self.nd == 2 else (modes1, modes2, modes3)
        for l in range(n_layers):
            for j in range(1, self.nw + 1):     # complex [I, O, *modes] seen through view_as_real
                add(f"convs.{l}.weights{j}", (W, W, *mshape, 2), (W, W, int(np.prod(mshape)) * 2, C, C))
        for l in range(n_layers):
            add(f"ws.{l}.weight", (W, W) + (1,) * self.nd, (W, W, 1, C, C))       # nn.Conv2d / nn.Conv3d(width, width, 1)
            add(f"ws.{l}.bias", (W,), (1, W, 1, 1, C))
        add("fc1.weight", (HEAD_DIM, W), (HEAD_DIM, W, 1, HEAD_DIM, C))
        add("fc1.bias", (HEAD_DIM,), (1, HEAD_DIM, 1, 1, HEAD_DIM))
        add("fc2.weight", (self.O, HEAD_DIM), (self.O, HEAD_DIM, 1, self.O, HEAD_DIM))
        add("fc2.bias", (self.O,), (1, self.O, 1, 1, self.O))
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

    # the planes hold one [2][C][C] block per retained mode
    def bind(self, params):
        dev_before = self.device
        super().bind(params)
        if self.device != dev_before:
            f32 = dict(dtype=torch.float32, device=self.device)
            n = 2 * self.Ktot * self.C * self.C
            self.planes = [(torch.empty(n, **f32), torch.empty(n, **f32)) for _ in range(self.L)]

    def _prepare_weights(self, st):
        lib = _lib.get_lib()
        self._refresh_pointers()
        self._packed = True
        if self._n_pad:
            self._k("pad_copy", lib.ffno_pad_copy, _p(self._ptab), self._n_pad, 1, st)
        for l in range(self.L):
            w = [_p(self._pp(f"convs.{l}.weights{j}")) for j in range(1, self.nw + 1)]
            if self.nd == 2:
                self._k("fw2d_pack", lib.ffno_fw2d_pack2, *w, _p(self.planes[l][0]), _p(self.planes[l][1]), self.C, self.Kx,
                        self.Ky, st)
            else:
                self._k("fw3d_pack", lib.ffno_fw3d_pack, *w, _p(self.planes[l][0]), _p(self.planes[l][1]), self.C, self.Kx,
                        self.Ky, self.Kz, st)

    def _workspace(self, B, S):
        """S = (X, Y) or (X, Y, Z): the unpadded mesh."""
        key = (B, tuple(S))
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        lib = _lib.get_lib()
        C, L, nd = self.C, self.L, self.nd
        Sp = tuple(d + self.pad for d in S)
        Ks = (self.Kx, self.Ky) if nd == 2 else (self.Kx, self.Ky, self.Kz)
        if any(2 * k > d for k, d in zip(Ks[:-1], Sp[:-1])) or Ks[-1] > Sp[-1] // 2 + 1:
            raise ValueError(f"modes={Ks} do not fit the padded {Sp} grid")
        f32 = dict(dtype=torch.float32, device=self.device)
        ws = type("WS", (), {})()
        ws.S, ws.Sp = tuple(S), Sp
        ws.P_in, ws.P = B * int(np.prod(S)), B * int(np.prod(Sp))
        s3, p3 = (1,) * (3 - nd) + tuple(S), (1,) * (3 - nd) + Sp
        ws.padmap = _capi.PadMap((ctypes.c_int32 * 3)(*s3), (ctypes.c_int32 * 3)(*p3))
        # last-axis (real) transform: lines = everything but the last axis
        lines = B * int(np.prod(Sp[:-1]))
        ws.v = v = _View(lines // Sp[-2], Sp[-2], Sp[-1], 0, self.K, C)
        ws.spec_z = self.K * lines * 2 * C                        # [kz][line][2][C]
        ws.spec = self.Ktot * B * 2 * C                           # [mode][b][2][C]
        P, P_in = ws.P, ws.P_in
        ws.SYa, ws.SYb = torch.empty(ws.spec_z, **f32), torch.empty(ws.spec_z, **f32)
        ws.SY, ws.SD = torch.empty(ws.spec, **f32), torch.empty(ws.spec, **f32)
        if nd == 2:
            ws.CW = torch.empty(int(lib.ffno_cdft_rows_ws_floats(B, C, self.Kx, self.Ky)), **f32)   # first-axis DFT scratch
        else:
            X1 = Sp[0]
            ws.spec_y = self.Kz * 2 * self.Ky * B * X1 * 2 * C    # after the y transform: [kz][ky'][(b, x)][2][C]
            ws.SYc, ws.SYd = torch.empty(ws.spec_y, **f32), torch.empty(ws.spec_y, **f32)
            ws.CW = torch.empty(max(int(lib.ffno_cdft_rows_ws_floats(B * X1, C, self.Ky, self.Kz)),
                                    int(lib.ffno_cdft_rows_ws_floats(B, C, self.Kx, self.Kz * 2 * self.Ky))), **f32)
        ws.Sb = torch.empty(P, C, **f32)
        ws.x = torch.empty(P_in, self.Cin, **f32)
        ws.X = torch.zeros(L + 1, P, C, **f32)           # X[0]: fc0 output in the padded frame (pad stays 0 for ever)
        ws.PRE = torch.empty(L, P, C, **f32)             # pre-activations (GELU derivative); last layer: unused
        ws.SX = torch.empty(L, ws.spec, **f32)
        ws.XC = torch.empty(P_in, C, **f32)              # cropped X[L]
        ws.H, ws.HPRE = torch.empty(P_in, HEAD_DIM, **f32), torch.empty(P_in, HEAD_DIM, **f32)
        ws.DH, ws.GC = torch.empty(P_in, HEAD_DIM, **f32), torch.empty(P_in, C, **f32)
        ws.G = [torch.zeros(P, C, **f32) for _ in range(2)]
        ws.DP = torch.empty(P, C, **f32)
        ws.fwpart = torch.empty(2 * self.Ktot * C * C, **f32)
        ws.part = torch.empty(int(lib.ffno_plin_wgrad_partial_floats(P, C, HEAD_DIM)), **f32)
        ws.nsplit_lift = max(1, min(1024, (P_in + 127) // 128))
        ws.liftpart = torch.empty(ws.nsplit_lift * C * (self.Cin + 1), **f32)
        self._ws[key] = ws
        while len(self._ws) > 3:
            self._ws.pop(next(iter(self._ws)))
        return ws

    def _spectral(self, ws, src, dst, save, planes, fwd: bool, accumulate: int, st, resid=None):
        """dst (+)= irfftn(corner-mix(rfftn(src))) on the padded grid (mesh_2d.py:38-53, mesh_3d.py:38-61), or its adjoint."""
        lib = _lib.get_lib()
        v, C, B = ws.v, self.C, ws.P // int(np.prod(ws.Sp))
        tw = self._twiddle(v.L)
        ck_f, ck_i, conj = (0, 1, 0) if fwd else (1, 0, 1)
        self._k("dft_fwd", lib.ffno_dft_fwd, _p(src), _p(ws.SYa), _p(tw), v.Bv, v.Mv, v.Nv, C, self.K, 0, ck_f, st)
        if self.nd == 2:
            M = ws.Sp[0]
            twm = self._twiddle(M)
            self._k("cdft_rows", lib.ffno_cdft_rows_mfma, _p(ws.SYa), _p(save), _p(ws.CW), _p(twm), B, M, C, self.Kx, self.Ky, 0,
                    st)
            self._k("mode_mix", lib.ffno_mode_mix, _p(save), _p(planes), _p(ws.SY), B, C, self.Ktot, conj, st)
            self._k("cdft_rows", lib.ffno_cdft_rows_mfma, _p(ws.SY), _p(ws.SYb), _p(ws.CW), _p(twm), B, M, C, self.Kx, self.Ky, 1,
                    st)
        else:
            X1, Y1 = ws.Sp[0], ws.Sp[1]
            twy, twx = self._twiddle(Y1), self._twiddle(X1)
            KyK = self.Kz * 2 * self.Ky             # "columns" of the x transform: the (kz, ky') pairs
            self._k("cdft_rows(y)", lib.ffno_cdft_rows_mfma, _p(ws.SYa), _p(ws.SYc), _p(ws.CW), _p(twy), B * X1, Y1, C, self.Ky,
                    self.Kz, 0, st)
            self._k("cdft_rows(x)", lib.ffno_cdft_rows_mfma, _p(ws.SYc), _p(save), _p(ws.CW), _p(twx), B, X1, C, self.Kx, KyK, 0,
                    st)
            self._k("mode_mix", lib.ffno_mode_mix, _p(save), _p(planes), _p(ws.SY), B, C, self.Ktot, conj, st)
            self._k("cdft_rows(x)", lib.ffno_cdft_rows_mfma, _p(ws.SY), _p(ws.SYd), _p(ws.CW), _p(twx), B, X1, C, self.Kx, KyK, 1,
                    st)
            self._k("cdft_rows(y)", lib.ffno_cdft_rows_mfma, _p(ws.SYd), _p(ws.SYb), _p(ws.CW), _p(twy), B * X1, Y1, C, self.Ky,
                    self.Kz, 1, st)
        self._k("dft_inv", lib.ffno_dft_inv, _p(ws.SYb), _p(dst), _p(resid), _p(tw), v.Bv, v.Mv, v.Nv, C, self.K, 0, ck_i,
                accumulate, st)

    @staticmethod
    def _inner(t, B, Sp, S, C):
        """The unpadded corner of a padded [B, *Sp, C] buffer (mesh_2d.py:95, mesh_3d.py:100)."""
        v = t.view(B, *Sp, C)
        return v[:, :S[0], :S[1]] if len(S) == 2 else v[:, :S[0], :S[1], :S[2]]

    # ------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, save_for_backward: bool) -> torch.Tensor:
        """x [B, *mesh, input_dim] (mesh values + grid channels) -> [B, *mesh, out]."""
        _lib.require_device_tensor(x, "x")
        if x.dim() != self.nd + 2 or x.shape[-1] != self.Cin:
            raise ValueError(f"expected x of shape [B, {'X, Y' if self.nd == 2 else 'X, Y, Z'}, {self.Cin}], got {tuple(x.shape)}")
        if not self.params:
            raise RuntimeError("bind() the parameters first")
        lib = _lib.get_lib()
        B, S = x.shape[0], tuple(x.shape[1:-1])
        ws = self._workspace(B, S)
        st = _lib.current_stream(self.device)
        self._issue_stream = st
        C, L, P, P_in, Cin, O = self.C, self.L, ws.P, ws.P_in, self.Cin, self.O
        self._prepare_weights(st)
        pp = self._pp
        pm = ctypes.byref(ws.padmap)
        ws.x.copy_(x.reshape(P_in, Cin))
        self._k("fc0", lib.ffno_lift_fwd, _p(ws.x), _p(pp("fc0.weight")), _p(pp("fc0.bias")), _p(ws.X[0]), P_in, Cin, C, pm, None, st)
        for l in range(L):
            last = l == L - 1
            self._spectral(ws, ws.X[l], ws.Sb, ws.SX[l], self.planes[l][0], True, 0, st)
            # x = conv(x) + w(x), GELU except after the last layer (mesh_2d.py:88-93, mesh_3d.py:93-98)
            self._k("layer_conv1x1", lib.ffno_plin_fwd, _p(ws.X[l]), C, _p(pp(f"ws.{l}.weight")), _p(pp(f"ws.{l}.bias")),
                    _p(ws.Sb), _p(ws.X[l + 1]), C, None, None, None if last else _p(ws.PRE[l]), P, C, C,
                    ACT_NONE if last else ACT_GELU, st)
        ws.XC.copy_(self._inner(ws.X[L], B, ws.Sp, S, C).reshape(P_in, C))      # crop
        self._k("fc1", lib.ffno_plin_fwd, _p(ws.XC), C, _p(pp("fc1.weight")), _p(pp("fc1.bias")), None, _p(ws.H), HEAD_DIM, None,
                None, _p(ws.HPRE), P_in, C, HEAD_DIM, ACT_GELU, st)
        y = torch.empty(P_in, O, dtype=torch.float32, device=self.device)
        self._k("fc2", lib.ffno_plin_fwd, _p(ws.H), HEAD_DIM, _p(pp("fc2.weight")), _p(pp("fc2.bias")), None, _p(y), O, None, None,
                None, P_in, HEAD_DIM, O, ACT_NONE, st)
        self._saved = (B, S) if save_for_backward else None
        return y.view(B, *S, O)

    def backward(self, gy: torch.Tensor) -> torch.Tensor:
        """gy = dL/dout [B, *mesh, out] -> the flat gradient buffer (``param_names`` order)."""
        _lib.require_device_tensor(gy, "gy")
        if getattr(self, "_saved", None) is None:
            raise RuntimeError("backward() needs a preceding forward(save_for_backward=True)")
        lib = _lib.get_lib()
        B, S = self._saved
        self._saved = None
        ws = self._workspace(B, S)
        st = _lib.current_stream(self.device)
        self._issue_stream = st
        C, L, P, P_in, Cin, O = self.C, self.L, ws.P, ws.P_in, self.Cin, self.O
        pp = self._pp
        gp = lambda n: self._pp(n, self.gpad)      # noqa: E731
        gy = gy.contiguous().view(P_in, O)
        self._k("fc2_bwd_w", lib.ffno_plin_bwd_weights, _p(gy), O, None, _p(ws.H), HEAD_DIM, _p(ws.part), _p(gp("fc2.weight")),
                _p(gp("fc2.bias")), P_in, HEAD_DIM, O, 0, ACT_NONE, st)
        self._k("fc2_bwd", lib.ffno_plin_bwd_data, _p(gy), O, None, _p(pp("fc2.weight")), _p(ws.DH), HEAD_DIM, None, P_in,
                HEAD_DIM, O, 0, ACT_NONE, st)
        self._k("fc1_bwd_w", lib.ffno_plin_bwd_weights, _p(ws.DH), HEAD_DIM, _p(ws.HPRE), _p(ws.XC), C, _p(ws.part),
                _p(gp("fc1.weight")), _p(gp("fc1.bias")), P_in, C, HEAD_DIM, 0, ACT_GELU, st)
        self._k("fc1_bwd", lib.ffno_plin_bwd_data, _p(ws.DH), HEAD_DIM, _p(ws.HPRE), _p(pp("fc1.weight")), _p(ws.GC), C, None,
                P_in, C, HEAD_DIM, 0, ACT_GELU, st)
        g = ws.G[0]
        g.zero_()                                                    # adjoint of the crop: zeros in the pad frame
        self._inner(g, B, ws.Sp, S, C).copy_(ws.GC.view(B, *S, C))
        cur = 0
        for l in range(L - 1, -1, -1):
            last = l == L - 1
            act, mode = (None, ACT_NONE) if last else (ws.PRE[l], ACT_GELU)
            gn = ws.G[1 - cur]
            self._k("layer_conv1x1_bwd_w", lib.ffno_plin_bwd_weights, _p(g), C, _p(act), _p(ws.X[l]), C, _p(ws.part),
                    _p(gp(f"ws.{l}.weight")), _p(gp(f"ws.{l}.bias")), P, C, C, 0, mode, st)
            self._k("layer_conv1x1_bwd", lib.ffno_plin_bwd_data, _p(g), C, _p(act), _p(pp(f"ws.{l}.weight")), _p(gn), C,
                    _p(ws.DP), P, C, C, 0, mode, st)
            self._spectral(ws, ws.DP, gn, ws.SD, self.planes[l][1], False, 1, st)
            self._k("fw_grad_partial", lib.ffno_fw_grad_partial, _p(ws.SX[l]), _p(ws.SD), _p(ws.fwpart), B, C, self.Ktot, 1, 0, 1,
                    ws.spec, ws.spec, st)
            gw = [_p(gp(f"convs.{l}.weights{j}")) for j in range(1, self.nw + 1)]
            if self.nd == 2:
                self._k("fw2d_grad_reduce", lib.ffno_fw2d_grad_reduce2, _p(ws.fwpart), *gw, C, self.Kx, self.Ky, 1, 0, st)
            else:
                self._k("fw3d_grad_reduce", lib.ffno_fw3d_grad_reduce, _p(ws.fwpart), *gw, C, self.Kx, self.Ky, self.Kz, 1, 0,
                        st)
            g, cur = gn, 1 - cur
        self._k("fc0_bwd", lib.ffno_lift_bwd, _p(ws.x), _p(g), _p(ws.liftpart), _p(gp("fc0.weight")), _p(gp("fc0.bias")),
                P_in, Cin, C, ws.nsplit_lift, 0, ctypes.byref(ws.padmap), st)
        if self._n_pad:
            self._k("pad_copy(grads)", lib.ffno_pad_copy, _p(self._gtab), self._n_pad, 0, st)
        return self.gflat


class GeoFNO3DEngine(GeoFNO2DEngine):
    """FNOMesh3D (mesh_3d.py:63-113)."""

    def __init__(self, *, modes1: int, modes2: int, modes3: int, width: int, n_layers: int, input_dim: int = 4):
        super().__init__(modes1=modes1, modes2=modes2, modes3=modes3, width=width, n_layers=n_layers, input_dim=input_dim)
