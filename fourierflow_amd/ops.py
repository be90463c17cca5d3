"""Operator-level autograd wrappers over the C ABI (include/ffno.h).

These expose the single operators of the hot path the way the reference exposes them as module
methods (``SpectralConv2d.forward_fourier``, ``FeedForward.forward``); the whole-block fast path is
:class:`fourierflow_amd.engine.FFNO2DEngine`.  HIP only -- CPU tensors raise.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _capi, _lib
from .engine import MODES, _p

_TW = {}


def _twiddle(L: int, device) -> torch.Tensor:
    key = (L, str(device), _lib.is_test_backend())
    if key not in _TW:
        host = np.zeros(2 * L, np.float32)
        _capi.check(_lib.get_lib().ffno_twiddle_fill_host(host.ctypes.data_as(ctypes.c_void_p), L), "twiddle")
        _TW[key] = torch.from_numpy(host).to(device)
    return _TW[key]


_SP2D_WS = {}


def _spectral2d_workspace(x, w_y, w_x, modes):
    """The operator's workspace for a shape, kept per (shape, device): besides the scratch spectra it holds the packed weight sets
    of the fused kernels (ffno_spectral2d_path), and the library reuses them across calls when the caller declares the VERSION of
    the weights (include/ffno.h: ffno_spectral2d_weights_version) -- here (data pointer, torch version counter) of both tensors, so
    a forward / backward pair and repeated calls on unchanged weights pack once, and an optimizer step re-packs."""
    lib = _lib.get_lib()
    B, M, N, C = x.shape
    key = (B, M, N, C, modes, str(x.device), _lib.is_test_backend())
    ws = _SP2D_WS.get(key)
    if ws is None:
        ws = torch.empty(int(lib.ffno_spectral2d_ws_floats(B, M, N, C, modes)), dtype=torch.float32, device=x.device)
        _SP2D_WS[key] = ws
        while len(_SP2D_WS) > 8:
            _SP2D_WS.pop(next(iter(_SP2D_WS)))
    version = 0
    if w_y is not None and w_x is not None:
        try:
            version = (hash((w_y.data_ptr(), w_y._version, w_x.data_ptr(), w_x._version)) & 0x7FFFFFFFFFFFFFFF) or 1
        except RuntimeError:        # inference tensors have no version counter: undeclared, the library packs on every call
            version = 0
    _capi.check(lib.ffno_spectral2d_weights_version(_p(ws), version), "spectral2d_weights_version")
    return ws


class _SpectralConv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_y, w_x, modes, mode_id):
        lib = _lib.get_lib()
        B, M, N, C = x.shape
        st = _lib.current_stream(x.device)
        ws = _spectral2d_workspace(x, w_y, w_x, modes)
        out = torch.empty_like(x)
        twn, twm = _twiddle(N, x.device), _twiddle(M, x.device)
        _capi.check(lib.ffno_spectral2d_fwd(_p(x), _p(w_y), _p(w_x), _p(out), _p(ws), _p(twn), _p(twm), B, M, N, C,
                                            modes, mode_id, st), "spectral2d_fwd")
        ctx.save_for_backward(x, w_y, w_x)
        ctx.cfg = (modes, mode_id)
        return out

    @staticmethod
    def backward(ctx, gy):
        x, w_y, w_x = ctx.saved_tensors
        modes, mode_id = ctx.cfg
        lib = _lib.get_lib()
        B, M, N, C = x.shape
        st = _lib.current_stream(x.device)
        gy = gy.contiguous()
        ws = _spectral2d_workspace(x, w_y, w_x, modes)
        gx = torch.empty_like(x)
        full = mode_id == MODES["full"]
        gwy = torch.empty_like(w_y) if full else None
        gwx = torch.empty_like(w_x) if full else None
        twn, twm = _twiddle(N, x.device), _twiddle(M, x.device)
        _capi.check(lib.ffno_spectral2d_bwd(_p(x), _p(w_y), _p(w_x), _p(gy), _p(gx), _p(gwy), _p(gwx), _p(ws), _p(twn),
                                            _p(twm), B, M, N, C, modes, mode_id, 0, 0, st), "spectral2d_bwd")
        return gx, gwy, gwx, None, None


def spectral_conv2d(x, w_y, w_x, modes: int, mode: str = "full"):
    """``SpectralConv2d.forward_fourier`` (reference grid_2d.py:51-99): x [B,M,N,C] -> [B,M,N,C].
    ``w_y`` = fourier_weight[0] (last spatial axis), ``w_x`` = fourier_weight[1]."""
    _lib.require_device_tensor(x, "spectral_conv2d input")
    if mode not in ("full", "low-pass"):
        raise ValueError("mode must be 'full' or 'low-pass'")
    B, M, N, C = x.shape
    if modes > N // 2 + 1 or modes > M // 2 + 1:
        raise ValueError(f"modes={modes} exceeds grid//2+1 for grid {M}x{N}")
    return _SpectralConv2dFn.apply(x.contiguous(), w_y.contiguous(), w_x.contiguous(), modes, MODES[mode])


class _WeightNormFn(torch.autograd.Function):
    @staticmethod
    def _run(g, v, w, dw, dg, dv, bwd):
        lib = _lib.get_lib()
        desc = (_capi.WnDesc * 1)(_capi.WnDesc(g.data_ptr(), v.data_ptr(), w.data_ptr() if w is not None else 0,
                                               dw.data_ptr() if dw is not None else 0,
                                               dg.data_ptr() if dg is not None else 0,
                                               dv.data_ptr() if dv is not None else 0, v.shape[0], v.shape[1]))
        dev = torch.from_numpy(np.frombuffer(bytes(desc), dtype=np.uint8).copy()).to(v.device)
        fn = lib.ffno_weightnorm_bwd if bwd else lib.ffno_weightnorm_fwd
        _capi.check(fn(_p(dev), 1, v.shape[0], _lib.current_stream(v.device)), "weightnorm")
        return dev  # keep alive until enqueued work is ordered behind later work on the same stream

    @staticmethod
    def forward(ctx, g, v):
        w = torch.empty_like(v)
        ctx.keep = _WeightNormFn._run(g, v, w, None, None, None, False)
        ctx.save_for_backward(g, v)
        return w

    @staticmethod
    def backward(ctx, dw):
        g, v = ctx.saved_tensors
        dw = dw.contiguous()
        dg, dv = torch.empty_like(g), torch.empty_like(v)
        ctx.keep2 = _WeightNormFn._run(g, v, None, dw, dg, dv, True)
        return dg, dv


def weight_norm_weight(lin) -> torch.Tensor:
    """Effective weight of a WNLinear container (W = g v/||v||, linear.py:48-49) through the HIP kernel."""
    if not lin.wnorm:
        return lin.weight
    return _WeightNormFn.apply(lin.weight_g.contiguous(), lin.weight_v.contiguous())


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b on [P, Cin] rows through the pointwise-linear kernels (csrc/plin.hip)."""

    @staticmethod
    def forward(ctx, x, W, b):
        lib = _lib.get_lib()
        P, Cin, Cout = x.shape[0], x.shape[1], W.shape[0]
        out = torch.empty(P, Cout, dtype=torch.float32, device=x.device)
        _capi.check(lib.ffno_plin_fwd(_p(x), Cin, _p(W), _p(b), None, _p(out), Cout, None, None, None, P, Cin, Cout, 0,
                                      _lib.current_stream(x.device)), "plin_fwd")
        ctx.save_for_backward(x, W)
        return out

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        lib = _lib.get_lib()
        P, Cin, Cout = x.shape[0], x.shape[1], W.shape[0]
        st = _lib.current_stream(x.device)
        g = g.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _capi.check(lib.ffno_plin_bwd_data(_p(g), Cout, None, _p(W), _p(dx), Cin, None, P, Cin, Cout, 0, 0, st), "plin_bwd_data")
        part = torch.empty(int(lib.ffno_plin_wgrad_partial_floats(P, Cin, Cout)), dtype=torch.float32, device=x.device)
        dW, db = torch.empty_like(W), torch.empty(Cout, dtype=torch.float32, device=x.device)
        _capi.check(lib.ffno_plin_bwd_weights(_p(g), Cout, None, _p(x), Cin, _p(part), _p(dW), _p(db), P, Cin, Cout, 0, 0, st),
                    "plin_bwd_weights")
        return dx, dW, db


def wn_linear(x, lin):
    """``WNLinear.forward`` (reference linear.py:41-52 = nn.Linear with weight_norm): x [..., in] -> [..., out]."""
    _lib.require_device_tensor(x, "WNLinear input")
    if x.shape[-1] != lin.in_features:
        raise ValueError(f"expected {lin.in_features} input features, got {tuple(x.shape)}")
    if not _lib.get_lib().ffno_plin_supported(lin.in_features, lin.out_features):
        raise NotImplementedError(f"stand-alone WNLinear({lin.in_features}, {lin.out_features}) is outside the pointwise-linear "
                                  "kernel set (both widths <= 128); inside the F-FNO block the linears run in the fused kernels")
    W = weight_norm_weight(lin)
    y = _LinearFn.apply(x.reshape(-1, lin.in_features).contiguous(), W.contiguous(), lin.bias)
    return y.view(*x.shape[:-1], lin.out_features)


class _FeedForwardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s, resid, W1, b1, W2, b2):
        lib = _lib.get_lib()
        P, C, H = s.shape[0], s.shape[1], W1.shape[0]
        st = _lib.current_stream(s.device)
        out = torch.empty_like(s)
        need = any(ctx.needs_input_grad)
        h = torch.empty(P, H, dtype=torch.float32, device=s.device) if need else None
        mask = torch.zeros(int(lib.ffno_ff_mask_words(P, H)), dtype=torch.int32, device=s.device) if need else None
        _capi.check(lib.ffno_ff_fwd(_p(s), _p(resid), _p(W1), _p(b1), _p(W2), _p(b2), _p(out), _p(h), _p(mask), P, C, H, st),
                    "ff_fwd")
        ctx.save_for_backward(s, W1, W2, h, mask)
        ctx.has_resid = resid is not None
        return out

    @staticmethod
    def backward(ctx, gout):
        s, W1, W2, h, mask = ctx.saved_tensors
        lib = _lib.get_lib()
        P, C, H = s.shape[0], s.shape[1], W1.shape[0]
        st = _lib.current_stream(s.device)
        gout = gout.contiguous()
        dh = torch.empty(P, H, dtype=torch.float32, device=s.device)
        ds = torch.empty_like(s)
        W1t, W2t = W1.t().contiguous(), W2.t().contiguous()
        _capi.check(lib.ffno_ff_bwd_data(_p(gout), _p(mask), _p(W1t), _p(W2t), _p(dh), _p(ds), P, C, H, st), "ff_bwd_data")
        nsplit = max(1, min(256, (P + 127) // 128))
        part = torch.empty(int(lib.ffno_ff_wgrad_partial_floats(C, H, nsplit)), dtype=torch.float32, device=s.device)
        _capi.check(lib.ffno_ff_bwd_weights_partial(_p(s), _p(gout), _p(h), _p(dh), _p(part), P, C, H, nsplit, st),
                    "ff_bwd_weights_partial")
        dW1, dW2 = torch.empty_like(W1), torch.empty_like(W2)
        db1 = torch.empty(H, dtype=torch.float32, device=s.device)
        db2 = torch.empty(C, dtype=torch.float32, device=s.device)
        _capi.check(lib.ffno_ff_bwd_weights_reduce(_p(part), _p(dW1), _p(dW2), _p(db1), _p(db2), C, H, nsplit, 0, st),
                    "ff_bwd_weights_reduce")
        return ds, (gout if ctx.has_resid else None), dW1, db1, dW2, db2


def feedforward(x, resid, lin0, lin1):
    """``FeedForward.forward`` (feedforward.py:13-19) [+ residual]: x [..., C] -> [..., C]."""
    _lib.require_device_tensor(x, "feedforward input")
    shp = x.shape
    s = x.reshape(-1, shp[-1]).contiguous()
    r = resid.reshape(-1, shp[-1]).contiguous() if resid is not None else None
    W1, W2 = weight_norm_weight(lin0), weight_norm_weight(lin1)
    out = _FeedForwardFn.apply(s, r, W1.contiguous(), lin0.bias, W2.contiguous(), lin1.bias)
    return out.view(shp)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, gamma, beta, eps):
        lib = _lib.get_lib()
        P, C = t.shape
        out = torch.empty_like(t)
        stats = torch.empty(P, 2, dtype=torch.float32, device=t.device)
        _capi.check(lib.ffno_layernorm_fwd(_p(t), _p(gamma), _p(beta), None, _p(out), _p(stats), P, C, float(eps),
                                           _lib.current_stream(t.device)), "layernorm_fwd")
        ctx.save_for_backward(t, gamma, stats)
        return out

    @staticmethod
    def backward(ctx, g):
        t, gamma, stats = ctx.saved_tensors
        lib = _lib.get_lib()
        P, C = t.shape
        g = g.contiguous()
        dt = torch.empty_like(t)
        part = torch.empty(2 * C * int(lib.ffno_layernorm_nsplit(P)), dtype=torch.float32, device=t.device)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        _capi.check(lib.ffno_layernorm_bwd(_p(t), _p(stats), _p(gamma), _p(g), None, None, _p(dt), _p(part), _p(dgamma),
                                           _p(dbeta), P, C, 0, _lib.current_stream(t.device)), "layernorm_bwd")
        return dt, dgamma, dbeta, None


def layer_norm(x, ln):
    """``nn.LayerNorm(C)`` over the last axis (the final stage of FeedForward(layer_norm=True), feedforward.py:18-19)."""
    _lib.require_device_tensor(x, "layer_norm input")
    C = x.shape[-1]
    y = _LayerNormFn.apply(x.reshape(-1, C).contiguous(), ln.weight.contiguous(), ln.bias.contiguous(), ln.eps)
    return y.view(x.shape)


class _LpRelLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        lib = _lib.get_lib()
        B = pred.shape[0]
        n = pred.numel() // B
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        g = torch.empty_like(pred)
        tmp = torch.empty(int(lib.ffno_lploss_tmp_floats(B, n)), dtype=torch.float32, device=pred.device)
        _capi.check(lib.ffno_lploss_fwd_bwd(_p(pred), _p(target), _p(loss), _p(g), _p(tmp), B, n, 1.0, None,
                                            _lib.current_stream(pred.device)), "lploss")
        ctx.save_for_backward(g)
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        (g,) = ctx.saved_tensors
        return g * gout, None


def lp_rel_loss(pred, target):
    """``LpLoss(size_average=True)(pred.reshape(B, -1), target.reshape(B, -1))`` (reference modules/loss.py:33-46):
    mean_b ||pred_b - target_b||_2 / ||target_b||_2, loss and gradient from one fused HIP pass."""
    _lib.require_device_tensor(pred, "pred")
    _lib.require_device_tensor(target, "target")
    if pred.numel() != target.numel() or pred.shape[0] != target.shape[0]:
        raise ValueError(f"shape mismatch: {tuple(pred.shape)} vs {tuple(target.shape)}")
    return _LpRelLossFn.apply(pred.contiguous(), target.contiguous())
