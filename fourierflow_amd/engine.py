"""Host-side driver of the F-FNO 2-D block: sequences the C-ABI kernels of include/ffno.h.

Mirrors the data flow of the reference's ``FNOFactorized2DBlock.forward``
(fourierflow/modules/factorized_fno/grid_2d.py:154-177) and of its autograd, but over pre-allocated
device workspaces and raw pointers:

  forward :  lift -> L x [ dft_fwd(y,x) -> mode_mix -> dft_inv(+sum) -> fused FF(+residual) ] -> head
  backward:  head_bwd -> L x [ ff_bwd_data, ff_bwd_weights, dft_fwd(adjoint), fw_grad, mode_mix^H,
                                dft_inv(adjoint, accumulating into the running gradient) ] -> lift_bwd
             -> weight-norm backward (one batched launch)

PyTorch is used for device memory and the stream only.  All parameter gradients land in one flat
fp32 buffer (``gflat``) so the trainer can run ONE fused AdamW and ONE RCCL all-reduce per step.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _capi, _lib

MODES = {"full": 0, "low-pass": 1, "no-fourier": 2}
HEAD_DIM = 128  # grid_2d.py:150-152: WNLinear(width, 128) -> WNLinear(128, 1)
_SUPPORTED_CH = {(64, 256), (64, 128), (32, 128), (32, 64)}


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class _Linear:
    __slots__ = ("prefix", "rows", "cols", "wnorm", "weff", "gweff", "wt")

    def __init__(self, prefix, rows, cols, wnorm):
        self.prefix, self.rows, self.cols, self.wnorm = prefix, rows, cols, wnorm
        self.weff = None   # effective weight [rows, cols] (after weight-norm)
        self.gweff = None  # gradient w.r.t. the effective weight
        self.wt = None     # transposed effective weight [cols, rows] (feed-forward linears only; backward)


class FFNO2DEngine:
    """Kernel sequencer for one FNOFactorized2DBlock configuration (fp32)."""

    def __init__(self, *, modes: int, width: int, input_dim: int, n_layers: int, factor: int,
                 share_weight: bool, share_fork: bool, ff_weight_norm: bool, mode: str = "full"):
        if mode not in MODES:
            raise ValueError(f"mode must be one of {list(MODES)}, got {mode!r}")
        C, H = width, factor * width
        if (C, H) not in _SUPPORTED_CH:
            raise ValueError(f"(width, factor*width)=({C},{H}) is outside the compiled HIP kernel set {_SUPPORTED_CH}")
        if not (0 < input_dim < 64):
            raise ValueError("input_dim must be in 1..63")
        self.K, self.C, self.H, self.Cin, self.L = modes, C, H, input_dim, n_layers
        self.mode, self.mode_id = mode, MODES[mode]
        self.share_weight, self.share_fork, self.wnorm = share_weight, share_fork, ff_weight_norm

        # ---- parameter inventory, in the reference's named_parameters() naming --------------------
        self.linears: Dict[str, _Linear] = {}
        self.param_shapes: Dict[str, Tuple[int, ...]] = {}

        def add_linear(prefix, rows, cols):
            if prefix in self.linears:
                return
            self.linears[prefix] = _Linear(prefix, rows, cols, ff_weight_norm)
            if ff_weight_norm:
                self.param_shapes[prefix + "weight_g"] = (rows, 1)
                self.param_shapes[prefix + "weight_v"] = (rows, cols)
            else:
                self.param_shapes[prefix + "weight"] = (rows, cols)
            self.param_shapes[prefix + "bias"] = (rows,)

        add_linear("in_proj.", C, input_dim)
        self.ff_prefix: List[str] = []
        self.fw_names: List[Tuple[str, str]] = []
        for l in range(n_layers):
            fp = "backcast_ff." if share_fork else f"spectral_layers.{l}.backcast_ff."
            self.ff_prefix.append(fp)
            add_linear(fp + "layers.0.0.", H, C)
            add_linear(fp + "layers.1.0.", C, H)
            if mode == "full":
                base = "fourier_weight." if share_weight else f"spectral_layers.{l}.fourier_weight."
                names = (base + "0", base + "1")
                self.fw_names.append(names)
                for n in names:
                    self.param_shapes.setdefault(n, (C, C, modes, 2))
        add_linear("out.0.", HEAD_DIM, C)
        add_linear("out.1.", 1, HEAD_DIM)
        self.param_names = list(self.param_shapes)
        self.n_params = sum(int(np.prod(s)) for s in self.param_shapes.values())
        self._offsets = {}
        off = 0
        for n in self.param_names:
            self._offsets[n] = off
            off += int(np.prod(self.param_shapes[n]))

        self.params: Dict[str, torch.Tensor] = {}
        self.device = None
        self._ptr_sig = None
        self._ws_key = None
        self._tw: Dict[int, torch.Tensor] = {}
        self._saved = None
        self.use_fused = True   # fused A->B->C branch kernel when (C, K, L) fits its LDS tile; else 3 stage kernels
        self._issue_stream = None   # torch stream object the next launches go to (None = current stream)
        # backward: FF weight-gradient kernels on a side stream next to the spectral adjoint.  Measured on MI355X
        # (profiles/r01_overlap_trace.md): co-running slows both kernels 2-4x (net -5 %), so it is OFF by default.
        self.overlap = False
        self.timer = None   # optional KernelTimer (bench.py): HIP-event timing of individual launches

    def _k(self, name, fn, *args):
        """Enqueue one C-ABI call; with a timer attached, bracket it with HIP events on the launch stream."""
        t = self.timer
        if t is not None and t.want(name):
            t.start(name, self._issue_stream)
            rc = fn(*args)
            t.stop(name, self._issue_stream)
        else:
            rc = fn(*args)
        if rc != 0:
            _capi.check(rc, name)

    # ------------------------------------------------------------------------------------------------
    def bind(self, params: Dict[str, torch.Tensor]):
        """Attach the parameter tensors (unique reference names -> fp32 contiguous device tensors)."""
        missing = [n for n in self.param_names if n not in params]
        if missing:
            raise KeyError(f"missing parameters: {missing[:4]}{'...' if len(missing) > 4 else ''}")
        dev = None
        for n in self.param_names:
            t = params[n]
            _lib.require_device_tensor(t, n)
            if tuple(t.shape) != self.param_shapes[n]:
                raise ValueError(f"{n}: expected shape {self.param_shapes[n]}, got {tuple(t.shape)}")
            if not t.is_contiguous():
                raise ValueError(f"{n} must be contiguous")
            dev = dev or t.device
            if t.device != dev:
                raise ValueError("all parameters must live on one device")
        self.params = {n: params[n] for n in self.param_names}
        if dev != self.device:
            self.device = dev
            self._alloc_param_buffers()
        self._ptr_sig = None

    def _alloc_param_buffers(self):
        dev = self.device
        self.gflat = torch.zeros(self.n_params, dtype=torch.float32, device=dev)
        n_eff = sum(l.rows * l.cols for l in self.linears.values()) if self.wnorm else 0
        self.weff_flat = torch.empty(max(n_eff, 1), dtype=torch.float32, device=dev)
        self.gweff_flat = torch.zeros(max(n_eff, 1), dtype=torch.float32, device=dev)
        self.fold = torch.zeros(self.C + 1, dtype=torch.float32, device=dev)
        n_ff = sum(l.rows * l.cols for p, l in self.linears.items() if "_ff." in p)
        self.wt_flat = torch.empty(max(n_ff, 1), dtype=torch.float32, device=dev)
        n_sets = len({n for n in self.fw_names})
        self._fw_sets = []  # unique (name_y, name_x) in first-use order
        for names in self.fw_names:
            if names not in self._fw_sets:
                self._fw_sets.append(names)
        plane = 2 * self.K * self.C * self.C
        self.planes = torch.empty((max(len(self._fw_sets), 1), 2, 2, plane), dtype=torch.float32, device=dev)
        self._ws_key = None
        self._tw = {}
        del n_sets

    def grad_view(self, name: str) -> torch.Tensor:
        o = self._offsets[name]
        return self.gflat[o:o + int(np.prod(self.param_shapes[name]))].view(self.param_shapes[name])

    def _refresh_pointers(self):
        sig = tuple(self.params[n].data_ptr() for n in self.param_names)
        if sig == self._ptr_sig:
            return
        self._ptr_sig = sig
        off = 0
        descs = []
        for lin in self.linears.values():
            n = lin.rows * lin.cols
            if lin.wnorm:
                lin.weff = self.weff_flat[off:off + n].view(lin.rows, lin.cols)
                lin.gweff = self.gweff_flat[off:off + n].view(lin.rows, lin.cols)
                off += n
                g, v = self.params[lin.prefix + "weight_g"], self.params[lin.prefix + "weight_v"]
                descs.append(_capi.WnDesc(g.data_ptr(), v.data_ptr(), lin.weff.data_ptr(), lin.gweff.data_ptr(),
                                          self.grad_view(lin.prefix + "weight_g").data_ptr(),
                                          self.grad_view(lin.prefix + "weight_v").data_ptr(), lin.rows, lin.cols))
            else:
                lin.weff = self.params[lin.prefix + "weight"]
                lin.gweff = self.grad_view(lin.prefix + "weight")
        self._n_desc = len(descs)
        toff, tdescs = 0, []
        for pfx, lin in self.linears.items():
            if "_ff." not in pfx:
                continue
            n = lin.rows * lin.cols
            lin.wt = self.wt_flat[toff:toff + n].view(lin.cols, lin.rows)
            toff += n
            tdescs.append(_capi.TrDesc(lin.weff.data_ptr(), lin.wt.data_ptr(), lin.rows, lin.cols))
        self._n_tr = len(tdescs)
        arr = (_capi.TrDesc * len(tdescs))(*tdescs)
        self._tr_dev = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(self.device)
        self._max_rows = max([l.rows for l in self.linears.values()])
        if descs:
            arr = (_capi.WnDesc * len(descs))(*descs)
            raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
            self._desc_dev = torch.from_numpy(raw).to(self.device)
        else:
            self._desc_dev = None

    def _twiddle(self, L: int) -> torch.Tensor:
        if L not in self._tw:
            host = np.zeros(2 * L, np.float32)
            _capi.check(_lib.get_lib().ffno_twiddle_fill_host(host.ctypes.data_as(ctypes.c_void_p), L), "twiddle")
            self._tw[L] = torch.from_numpy(host).to(self.device)
        return self._tw[L]

    # ------------------------------------------------------------------------------------------------
    def _workspace(self, B, M, N, save: bool):
        key = (B, M, N, bool(save))
        if self._ws_key == key:
            return self._ws
        lib = _lib.get_lib()
        dev, C, H, K, L = self.device, self.C, self.H, self.K, self.L
        P = B * M * N
        f32 = dict(dtype=torch.float32, device=dev)
        ns = L if save else 1
        ws = type("WS", (), {})()
        ws.P, ws.R = P, (B * M, B * N)
        ws.X = torch.empty(P, C, **f32)
        ws.Blast = torch.empty(P, C, **f32)
        ws.Y = torch.empty(P, **f32)
        ws.S = torch.empty(ns, P, C, **f32)
        spec = [K * ws.R[0] * 2 * C, K * ws.R[1] * 2 * C]
        ws.spec = spec
        ws.SXall = [torch.empty(ns, spec[a], **f32) for a in (0, 1)]      # forward spectra, per axis, layer-major
        ws.SX = [[ws.SXall[a][i] for a in (0, 1)] for i in range(ns)]
        ws.SY = torch.empty(max(spec), **f32)
        ws.mask_words = int(lib.ffno_ff_mask_words(P, H))
        if save:
            ws.Hbuf = torch.empty(ns, P, H, **f32)
            ws.MASK = torch.zeros(ns, ws.mask_words, dtype=torch.int32, device=dev)
            ws.DH = [torch.empty(P, H, **f32) for _ in range(2)]   # ping-pong: the side stream reads one while the next layer writes the other
            ws.DS = torch.empty(P, C, **f32)
            ws.G = [torch.empty(P, C, **f32) for _ in range(2)]    # running gradient, ping-pong per layer
            ws.SD = torch.empty(max(spec), **f32)
            ws.SDall = [torch.empty(L, spec[a], **f32) for a in (0, 1)] if self.mode == "full" else None
            ws.nsplit_ff = max(1, min(256, (P + 127) // 128))
            ws.ffpart = torch.empty(int(lib.ffno_ff_wgrad_partial_floats(C, H, ws.nsplit_ff)), **f32)
            ws.nsplit_fw = [max(1, min(max(1, 512 // K), (L * r + 63) // 64)) for r in ws.R]
            plane = 2 * K * C * C
            ws.fwpart = [[torch.empty(ws.nsplit_fw[a] * plane, **f32) for a in (0, 1)]
                         for _ in range(max(len(self._fw_sets), 1))]
            ws.nsplit_lift = max(1, min(256, (P + 255) // 256))
            ws.liftpart = torch.empty(ws.nsplit_lift * C * (self.Cin + 1), **f32)
            ws.nsplit_head = max(1, min(256, (P + 255) // 256))
            ws.headpart = torch.empty(ws.nsplit_head * (C + 1), **f32)
            ws.red = torch.empty(C + 1, **f32)
        self._ws, self._ws_key = ws, key
        return ws

    def _prepare_weights(self, st):
        lib = _lib.get_lib()
        self._refresh_pointers()
        if self._desc_dev is not None:
            self._k("weightnorm_fwd", lib.ffno_weightnorm_fwd, _p(self._desc_dev), self._n_desc, self._max_rows, st)
        for i, (ny, nx) in enumerate(self._fw_sets):
            for a, n in enumerate((ny, nx)):
                self._k("fw_pack", lib.ffno_fw_pack, _p(self.params[n]), _p(self.planes[i, a, 0]), _p(self.planes[i, a, 1]),
                                             self.C, self.K, st)
        o0, o1 = self.linears["out.0."], self.linears["out.1."]
        self._k("head_fold", lib.ffno_head_fold, _p(o0.weff), _p(self.params["out.0.bias"]), _p(o1.weff),
                                       _p(self.params["out.1.bias"]), _p(self.fold), self.C, HEAD_DIM, st)

    def _ff_weights(self, l):
        fp = self.ff_prefix[l]
        l0, l1 = self.linears[fp + "layers.0.0."], self.linears[fp + "layers.1.0."]
        return l0, l1, self.params[fp + "layers.0.0.bias"], self.params[fp + "layers.1.0.bias"]

    # ------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, save_for_backward: bool) -> torch.Tensor:
        """x [B, M, N, input_dim] fp32 on the device -> forecast [B, M, N, 1] (a fresh tensor)."""
        if not self.params:
            raise RuntimeError("bind() parameters first")
        _lib.require_device_tensor(x, "x")
        if x.dim() != 4 or x.shape[-1] != self.Cin:
            raise ValueError(f"x must be [B, M, N, {self.Cin}], got {tuple(x.shape)}")
        x = x.contiguous()
        B, M, N, _ = x.shape
        lib = _lib.get_lib()
        C, H, K, L = self.C, self.H, self.K, self.L
        if self.mode != "no-fourier" and (K > N // 2 + 1 or K > M // 2 + 1):
            raise ValueError(f"modes={K} exceeds grid//2+1 for grid {M}x{N} (the reference raises an einsum size error)")
        ws = self._workspace(B, M, N, save_for_backward)
        st = _lib.current_stream(self.device)
        P = ws.P
        self._prepare_weights(st)
        tw = (self._twiddle(N), self._twiddle(M))
        fused = (self.use_fused and self.mode != "no-fourier" and
                 all(lib.ffno_spectral_fused_supported(C, K, Lx) for Lx in (N, M)))
        lin_in = self.linears["in_proj."]
        self._k("lift_fwd", lib.ffno_lift_fwd, _p(x), _p(lin_in.weff), _p(self.params["in_proj.bias"]), _p(ws.X), P, self.Cin, C, st)
        for l in range(L):
            sv = l if save_for_backward else 0
            last = l == L - 1
            s_l = ws.S[sv]
            if self.mode == "no-fourier":
                s_l.copy_(ws.X)
            else:
                for a in (0, 1):
                    sx = ws.SX[sv][a]
                    si = self._fw_sets.index(self.fw_names[l]) if self.mode == "full" else 0
                    if fused:
                        self._k("spectral_fused", lib.ffno_spectral_fused, _p(ws.X), _p(s_l), None,
                                _p(sx) if (save_for_backward and self.mode == "full") else None,
                                _p(self.planes[si, a, 0]) if self.mode == "full" else None, _p(tw[a]),
                                B, M, N, C, K, a, 0, 1, 0, int(a == 1), st)
                        continue
                    self._k("dft_fwd", lib.ffno_dft_fwd, _p(ws.X), _p(sx), _p(tw[a]), B, M, N, C, K, a, 0, st)
                    y = sx
                    if self.mode == "full":
                        self._k("mode_mix", lib.ffno_mode_mix, _p(sx), _p(self.planes[si, a, 0]), _p(ws.SY), ws.R[a], C, K, 0, st)
                        y = ws.SY
                    self._k("dft_inv", lib.ffno_dft_inv, _p(y), _p(s_l), None, _p(tw[a]), B, M, N, C, K, a, 1, int(a == 1), st)
            l0, l1, b0, b1 = self._ff_weights(l)
            self._k("ff_fwd", lib.ffno_ff_fwd, _p(s_l), None if last else _p(ws.X), _p(l0.weff), _p(b0), _p(l1.weff), _p(b1),
                                        _p(ws.Blast if last else ws.X),
                                        _p(ws.Hbuf[sv]) if save_for_backward else None,
                                        _p(ws.MASK[sv]) if save_for_backward else None, P, C, H, st)
        self._k("head_fwd", lib.ffno_head_fwd, _p(ws.Blast), _p(self.fold), _p(ws.Y), P, C, 0, st)
        self._saved = (x, B, M, N) if save_for_backward else None
        return ws.Y.view(B, M, N, 1).clone()

    # ------------------------------------------------------------------------------------------------
    def backward(self, gy: torch.Tensor) -> torch.Tensor:
        """gy = dL/dforecast [B, M, N, 1].  Fills and returns the flat gradient buffer ``gflat``
        (layout: ``param_names`` order; use ``grad_view(name)``)."""
        if self._saved is None:
            raise RuntimeError("backward() needs a preceding forward(save_for_backward=True)")
        x, B, M, N = self._saved
        _lib.require_device_tensor(gy, "gy")
        gy = gy.contiguous()
        lib = _lib.get_lib()
        C, H, K, L = self.C, self.H, self.K, self.L
        ws = self._workspace(B, M, N, True)
        st = _lib.current_stream(self.device)
        P = ws.P
        tw = (self._twiddle(N), self._twiddle(M))
        fused = (self.use_fused and self.mode != "no-fourier" and
                 all(lib.ffno_spectral_fused_supported(C, K, Lx) for Lx in (N, M)))
        o0, o1 = self.linears["out.0."], self.linears["out.1."]
        gv = self.grad_view
        # Two streams: the main stream carries the dependency chain  ff_bwd_data(l) -> spectral adjoint(l) ->
        # ff_bwd_data(l-1) ...; the FF weight-gradient GEMMs of layer l only need (G_l, dh_l, s_l, h_l), so they run
        # on a side stream concurrently with the adjoint of layer l and the data gradient of layer l-1 (the kernels
        # co-reside on a CU: 67 KiB + 66 KiB of LDS).  G and dh are ping-pong buffers; events order the reuse.
        use_side = self.overlap and self.device.type == "cuda" and self.mode != "no-fourier"
        side = ev_a = ev_b = None
        main_obj = None
        if use_side:
            if getattr(self, "_side", None) is None:
                self._side = torch.cuda.Stream(self.device)
                self._ev = [torch.cuda.Event() for _ in range(3)]
            side, main_obj = self._side, torch.cuda.current_stream(self.device)
            ev_a, ev_b = self._ev[0], self._ev[1:]
            st_side = ctypes.c_void_p(side.cuda_stream)
        else:
            st_side = st
        cur = 0
        self._k("head_bwd", lib.ffno_head_bwd, _p(ws.Blast), _p(gy), _p(self.fold), _p(ws.G[cur]), _p(ws.headpart), _p(ws.red), P, C,
                                      ws.nsplit_head, st)
        self._k("head_param_grads", lib.ffno_head_param_grads, _p(ws.red), _p(o0.weff), _p(self.params["out.0.bias"]), _p(o1.weff),
                                              _p(o0.gweff), _p(gv("out.0.bias")), _p(o1.gweff), _p(gv("out.1.bias")),
                                              C, HEAD_DIM, 0, st)
        self._k("transpose_batched", lib.ffno_transpose_batched, _p(self._tr_dev), self._n_tr, max(C, H), max(C, H), st)
        ff_seen = set()
        for l in reversed(range(L)):
            last = l == L - 1
            l0, l1, _, _ = self._ff_weights(l)
            fp = self.ff_prefix[l]
            g_in, g_out, dh = ws.G[cur], ws.G[1 - cur], ws.DH[l & 1]
            self._k("ff_bwd_data", lib.ffno_ff_bwd_data, _p(g_in), _p(ws.MASK[l]), _p(l0.wt), _p(l1.wt), _p(dh), _p(ws.DS),
                    P, C, H, st)
            if use_side:
                ev_a.record(main_obj)
                side.wait_event(ev_a)
                self._issue_stream = side
            self._k("ff_bwd_weights_partial", lib.ffno_ff_bwd_weights_partial, _p(ws.S[l]), _p(g_in), _p(ws.Hbuf[l]), _p(dh),
                    _p(ws.ffpart), P, C, H, ws.nsplit_ff, st_side)
            self._k("ff_bwd_weights_reduce", lib.ffno_ff_bwd_weights_reduce, _p(ws.ffpart), _p(l0.gweff), _p(l1.gweff),
                    _p(gv(fp + "layers.0.0.bias")), _p(gv(fp + "layers.1.0.bias")), C, H, ws.nsplit_ff, int(fp in ff_seen), st_side)
            ff_seen.add(fp)
            if use_side:
                self._issue_stream = None
                ev_b[l & 1].record(side)
                if not last:
                    # layer l+1's weight-gradient kernels read g_out's buffer (their G) and DH[(l+1)&1]: wait for them
                    main_obj.wait_event(ev_b[(l + 1) & 1])
            if self.mode == "no-fourier":
                if last:
                    g_out.copy_(ws.DS)
                else:
                    torch.add(g_in, ws.DS, out=g_out)
                cur = 1 - cur
                continue
            si = self._fw_sets.index(self.fw_names[l]) if self.mode == "full" else 0
            full = self.mode == "full"
            resid = None if last else _p(g_in)      # G_{l-1} = G_l (residual path) + adjoint terms; last layer: no residual
            for a in (0, 1):
                sd = ws.SDall[a][l] if full else ws.SD   # dY of every layer is kept for the dW launch
                if fused:
                    self._k("spectral_fused(adj)", lib.ffno_spectral_fused, _p(ws.DS), _p(g_out), resid if a == 0 else None,
                            _p(sd) if full else None, _p(self.planes[si, a, 1]) if full else None, _p(tw[a]),
                            B, M, N, C, K, a, 1, 0, 1, int(a == 1), st)
                    continue
                self._k("dft_fwd(adj)", lib.ffno_dft_fwd, _p(ws.DS), _p(sd), _p(tw[a]), B, M, N, C, K, a, 1, st)
                dxs = sd
                if full:
                    self._k("mode_mix(adj)", lib.ffno_mode_mix, _p(sd), _p(self.planes[si, a, 1]), _p(ws.SY), ws.R[a], C, K, 1, st)
                    dxs = ws.SY
                self._k("dft_inv(adj)", lib.ffno_dft_inv, _p(dxs), _p(g_out), resid if a == 0 else None, _p(tw[a]), B, M, N, C, K, a,
                        0, int(a == 1), st)
            cur = 1 - cur
        if use_side:
            main_obj.wait_event(ev_b[0])      # every FF gradient is in place before weight-norm backward / the optimiser
        g_fin = ws.G[cur]
        lin_in = self.linears["in_proj."]
        self._k("lift_bwd", lib.ffno_lift_bwd, _p(x), _p(g_fin), _p(ws.liftpart), _p(lin_in.gweff), _p(gv("in_proj.bias")), P,
                                      self.Cin, C, ws.nsplit_lift, 0, st)
        for si, names in enumerate(self._fw_sets):
            layers = [l for l in range(L) if self.fw_names[l] == names]
            l0_, nl = layers[0], len(layers)
            assert layers == list(range(l0_, l0_ + nl))
            for a, n in enumerate(names):
                # dW = sum over the lines of every layer that uses this weight: ONE launch per axis
                self._k("fw_grad_partial", lib.ffno_fw_grad_partial, _p(ws.SXall[a][l0_]), _p(ws.SDall[a][l0_]),
                        _p(ws.fwpart[si][a]), ws.R[a], C, K, ws.nsplit_fw[a], 0, nl, ws.spec[a], ws.spec[a], st)
                self._k("fw_grad_reduce", lib.ffno_fw_grad_reduce, _p(ws.fwpart[si][a]), _p(gv(n)), C, K, ws.nsplit_fw[a], 0, st)
        if self._desc_dev is not None:
            self._k("weightnorm_bwd", lib.ffno_weightnorm_bwd, _p(self._desc_dev), self._n_desc, self._max_rows, st)
        return self.gflat
