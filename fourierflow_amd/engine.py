"""Host-side driver of the F-FNO block: sequences the C-ABI kernels of include/ffno.h.

Mirrors the data flow of the reference's ``FNOFactorized2DBlock.forward``
(fourierflow/modules/factorized_fno/grid_2d.py:154-177), of the 3-D mesh variant
``FNOFactorizedMesh3D.forward`` (mesh_3d.py:160-176) and of their autograd, over pre-allocated device
workspaces and raw pointers:

  forward :  lift -> L x [ per axis: DFT -> mode mix -> iDFT (+sum) ; fused FF (+residual) ] -> head
  backward:  head_bwd -> L x [ ff_bwd_data, ff_bwd_weights, per axis: adjoint DFT/mix/iDFT accumulating into
             the running gradient ] -> lift_bwd -> Fourier-weight gradients (one launch per weight over all
             layers) -> weight-norm backward (one batched launch)

Every spatial axis is a *view* of the channels-last activation buffer as [B', M', N', C] plus a flag saying
whether the transform runs along N' (contiguous lines) or M' (strided lines), so the same spectral kernels
serve the 2-D grid (2 axes) and the 3-D mesh (3 axes, any sizes incl. the +8 padding).

PyTorch is used for device memory and the stream only.  All parameter gradients land in one flat
fp32 buffer (``gflat``) so the trainer can run ONE fused AdamW and ONE RCCL all-reduce per step.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _capi, _lib

MODES = {"full": 0, "low-pass": 1, "no-fourier": 2}
HEAD_DIM = 128  # grid_2d.py:150-152 / mesh_3d.py:155-157: WNLinear(width, 128) -> WNLinear(128, out)
_SUPPORTED_CH = {(64, 256), (64, 128), (32, 128), (32, 64)}
# paired spectral launch, workgroup -> (branch, tile) map (ffno_spectral_x3_pair `interleave`): bit 1 = image-local where the shapes
# allow it (the workgroups that read one image share an XCD, so the image crosses HBM once: PMC traffic 178.9 -> 150.7 MB per launch,
# round 2), else bit 0 = even workgroups branch a, odd ones branch b
X3_INTERLEAVE = 3
# all-layers feed-forward weight-gradient launch: rounds of resident workgroups (FF_WGRAD_ROUNDS x CUs / L slices per layer)
FF_WGRAD_ROUNDS = 3


def _fmix32(h: int) -> int:
    h &= 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    return h ^ (h >> 16)


def _site_seed(*words: int) -> int:
    """Seed of one dropout site: a hash CHAIN over (base seed, training call, layer, linear, block kind), not a linear
    combination of them -- linear seeds made the masks of neighbouring sites shifted copies of one sequence."""
    h = 0x9E3779B1
    for w in words:
        h = _fmix32(h ^ _fmix32((int(w) + 0x7F4A7C15) & 0xFFFFFFFF))
    return h


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class _StackRetry(Exception):
    """The first persistent inference launch of a workspace reported an error: forward() starts over on the per-layer launches."""


class _Linear:
    __slots__ = ("prefix", "rows", "cols", "wnorm", "weff", "gweff", "wt", "fx")

    def __init__(self, prefix, rows, cols, wnorm):
        self.prefix, self.rows, self.cols, self.wnorm = prefix, rows, cols, wnorm
        self.weff = None   # effective weight [rows, cols] (after weight-norm)
        self.gweff = None  # gradient w.r.t. the effective weight
        self.wt = None     # transposed effective weight [cols, rows] (feed-forward linears only; backward)


class _View:
    """One spatial axis as a [Bv, Mv, Nv, C] view: a01 = 0 transforms along Nv, 1 along Mv."""
    __slots__ = ("Bv", "Mv", "Nv", "a01", "L", "K", "R", "spec", "K2", "spec_y", "x3fmt")

    def __init__(self, Bv, Mv, Nv, a01, K, C):
        self.Bv, self.Mv, self.Nv, self.a01, self.K = Bv, Mv, Nv, a01, K
        self.L = Nv if a01 == 0 else Mv
        self.R = Bv * Mv if a01 == 0 else Bv * Nv
        self.spec = K * self.R * 2 * C


class FFNOEngine:
    """Kernel sequencer for one factorized-FNO configuration (fp32).

    spatial_dims = 2: FNOFactorized2DBlock -- fourier_weight[0] mixes the LAST axis, [1] the first
                      (grid_2d.py:68,86); no padding; one output channel.
                      (``first_axis_first``: FNOFactorizedMesh2D -- [0] mixes x, [1] mixes y (mesh_2d.py:71-75,92-96), +8
                      zero padding like the 3-D operator (mesh_2d.py:150,158)).
    spatial_dims = 3: FNOFactorizedMesh3D -- fourier_weight[0,1,2] mix x, y, z (mesh_3d.py:71,86,101);
                      activations zero-padded by ``padding`` at the end of every axis after the lift and
                      cropped before the head (mesh_3d.py:165,173); ``output_dim`` outputs.
    """

    can_return_view = True      # forward(..., own_output=False) exists (FFNOTrainer.train_step)

    def __init__(self, *, modes, width: int, input_dim: int, n_layers: int, factor: int, share_weight: bool,
                 share_fork: bool = False, ff_weight_norm: bool = False, mode: str = "full", spatial_dims: int = 2,
                 padding: int = 0, output_dim: int = 1, use_fork: bool = False, first_axis_first: bool = False,
                 spectral: str = "factorized", layer_norm: bool = False, n_ff_layers: int = 2, dropout: float = 0.0,
                 in_dropout: float = 0.0):
        if mode not in MODES:
            raise ValueError(f"mode must be one of {list(MODES)}, got {mode!r}")
        if spatial_dims not in (2, 3):
            raise ValueError("spatial_dims must be 2 or 3")
        C, H = width, factor * width
        if (C, H) not in _SUPPORTED_CH:
            raise ValueError(f"(width, factor*width)=({C},{H}) is outside the compiled HIP kernel set {_SUPPORTED_CH}")
        if not (0 < input_dim < 64):
            raise ValueError("input_dim must be in 1..63")
        if not (1 <= output_dim <= 8):
            raise ValueError("output_dim must be in 1..8")
        if n_ff_layers < 2:
            raise NotImplementedError("n_ff_layers = 1 (a single width -> width linear) is not built")
        if not (0.0 <= dropout < 1.0 and 0.0 <= in_dropout < 1.0):
            raise ValueError("dropout / in_dropout must lie in [0, 1)")
        # FeedForward beyond the fused kernels' shape (n_layers = 2, dropout = 0; feedforward.py:13-23): the GENERAL path, one
        # glin kernel per linear layer with kept hidden activations and regenerated dropout masks (csrc/glin.hip)
        self.n_ff, self.dropout, self.in_dropout = int(n_ff_layers), float(dropout), float(in_dropout)
        self.general_ff = self.n_ff != 2 or self.dropout > 0.0
        self.drop_seed = int(torch.initial_seed()) & 0x7FFFFFFF      # base seed of the dropout masks (tests may set it)
        self._drop_calls = 0                                          # training forwards so far: every one draws new masks
        # FeedForward(layer_norm=True): nn.LayerNorm(width) after the last linear of every feed-forward (feedforward.py:18-19)
        self.layer_norm = bool(layer_norm)
        if spectral not in ("factorized", "plus", "dct"):
            raise ValueError("spectral must be 'factorized', 'plus' or 'dct'")
        if spectral == "dct" and mode != "full":
            raise ValueError("the DCT operators (CNOFactorized*) have no `mode` switch")
        if spectral == "plus" and (spatial_dims != 2 or padding or mode == "low-pass"):
            raise ValueError("the non-factorized (FNOPlus2DBlock) spectral conv is 2-D, unpadded, mode 'full' or 'no-fourier'")
        # "plus": rfft2 + two K x K corner blocks (zongyi_fno/grid_plus_2d.py:52-83);  "dct": orthonormal DCT-II per axis with
        # real per-mode weights [C, C, K] (CNOFactorized*, factorized_cno/grid_2d.py:51-96)
        self.spectral = spectral
        self.nd = spatial_dims
        # 2-D weight order: grid_2d.py has fourier_weight[0] on the LAST axis; mesh_2d.py has [0] on the FIRST (x) axis
        self.first_axis_first = bool(first_axis_first)
        self.Ks: Tuple[int, ...] = tuple(modes) if isinstance(modes, (tuple, list)) else (int(modes),) * spatial_dims
        if len(self.Ks) != spatial_dims:
            raise ValueError("one mode count per spatial axis")
        self.K = self.Ks[0]
        self.C, self.H, self.Cin, self.L, self.O, self.pad = C, H, input_dim, n_layers, output_dim, padding
        self.mode, self.mode_id = mode, MODES[mode]
        self.share_weight, self.share_fork, self.wnorm = share_weight, share_fork, ff_weight_norm
        self.use_fork = use_fork   # per-layer forecast heads: forecast = sum_l out(forecast_ff_l(s_l))  (grid_2d.py:164-167)

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
            if layer_norm and prefix.endswith(f"_ff.layers.{self.n_ff - 1}.0."):   # the Sequential slot of the reference: layers.<last>.3
                self.param_shapes[prefix[:-2] + "3.weight"] = (rows,)
                self.param_shapes[prefix[:-2] + "3.bias"] = (rows,)

        add_linear("in_proj.", C, input_dim)
        self.ff_prefix: List[str] = []
        self.fc_prefix: List[str] = []     # forecast feed-forwards (use_fork)
        self.fw_names: List[Tuple[str, ...]] = []
        def add_ff(prefix):             # FeedForward(dim, factor, ..., n_layers): dims C -> H (-> H)* -> C (feedforward.py:13-16)
            for k in range(self.n_ff):
                add_linear(prefix + f"layers.{k}.0.", C if k == self.n_ff - 1 else H, C if k == 0 else H)

        if share_fork:                      # registration order of the reference: forecast_ff, backcast_ff (grid_2d.py:116-122)
            if use_fork:
                add_ff("forecast_ff.")
            add_ff("backcast_ff.")
        for l in range(n_layers):
            if use_fork:
                fc = "forecast_ff." if share_fork else f"spectral_layers.{l}.forecast_ff."
                self.fc_prefix.append(fc)
                add_ff(fc)
            fp = "backcast_ff." if share_fork else f"spectral_layers.{l}.backcast_ff."
            self.ff_prefix.append(fp)
            add_ff(fp)
            if mode == "full":
                base = "fourier_weight." if share_weight else f"spectral_layers.{l}.fourier_weight."
                names = tuple(base + str(w) for w in range(self.nd))
                self.fw_names.append(names)
                for w, n in enumerate(names):
                    self.param_shapes.setdefault(n, (C, C, self.Ks[w], self.Ks[w], 2) if spectral == "plus"
                                                 else (C, C, self.Ks[w]) if spectral == "dct" else (C, C, self.Ks[w], 2))
        add_linear("out.0.", HEAD_DIM, C)
        add_linear("out.1.", output_dim, HEAD_DIM)
        self.param_names = list(self.param_shapes)
        self.n_params = sum(int(np.prod(s)) for s in self.param_shapes.values())
        self._offsets = {}
        off = 0
        for n in self.param_names:
            self._offsets[n] = off
            off += int(np.prod(self.param_shapes[n]))
        self._fw_sets: List[Tuple[str, ...]] = []   # unique weight tuples in first-use order
        for names in self.fw_names:
            if names not in self._fw_sets:
                self._fw_sets.append(names)

        self.params: Dict[str, torch.Tensor] = {}
        self.device = None
        self._ptr_sig = None
        self._ws_key = None
        self._tw: Dict[int, torch.Tensor] = {}
        self._saved = None
        self._training = False
        self.use_fused = True   # fused A->B->C branch kernel when (C, K, L) fits its LDS tile; else 3 stage kernels
        # operand split of those kernels: "fp16x2" (ffno_ffh_*: two fp16 planes, three MFMAs per product block -- half the matrix
        # work; gradients are range-scaled by a device-resident power of two taken from the loss gradient) or "bf16x3"
        # (ffno_ffx_*: three bf16 planes, six MFMAs, any fp32 range).  Both are fp32-grade (tests/test_kernels_ffh.py).
        self.ff_split = os.environ.get("FFNO_FF_SPLIT", "fp16x2")
        # fused branches on the bf16 matrix cores at fp32 accuracy (spectral_x3.hip: 16 or 8 lines per workgroup -- the library
        # picks 8 while the launch still fits one round of workgroups -- packed pre-split weights) for the axes the library
        # takes (C = 64, K <= 16); 17..32 modes go through the split-bf16 stage kernels
        self.use_x3 = True
        # the per-mode channel mix of the fused x3 kernel on "fp16x2" (three fp16 MFMAs per product block, weight packs two thirds
        # the size; gradient passes hold the spectrum tile range-scaled like the feed-forward) or "bf16x3"
        self.x3_mix_split = os.environ.get("FFNO_X3_MIX_SPLIT", "fp16x2")
        self._x3_fmt = None
        # operand split of the Fourier-weight-gradient contraction: None = follow x3_mix_split (fp16x2: three fp16 MFMAs per product,
        # operands scaled from the range words of the layer inputs / data gradients; the bf16x3 launch runs on the 1400 W power cap)
        self.fw_grad_split = os.environ.get("FFNO_FW_GRAD_SPLIT") or None
        self._dft_tabs = {}
        # feed-forward weight gradients of ALL layers as one launch after the backward loop (ffno_ffh_bwd_weights_partial_multi):
        # every layer keeps its own gradient buffer instead of the ping-pong pair; FF_WGRAD_ROUNDS x resident workgroups / L
        # slices per layer
        self.ff_wgrad_deferred = os.environ.get("FFNO_FF_WGRAD_DEFERRED", "1") != "0"
        # ... and then the chain kernels leave the sums of their two input tensors unwritten (s_sum / db_sum NULL): every layer keeps
        # both branch outputs and both gradient buffers, the weight-gradient launch adds them while it stages its rows
        #   "s": the forward's input sums (the backward-data launch still writes its summed gradient: leaving that one out too was
        #   measured slower, round 4);  "0": none
        self.ff_lazy_sums = os.environ.get("FFNO_FF_LAZY_SUMS", "s")
        self.ff_schedule = int(os.environ.get("FFNO_FF_SCHED", "0"))      # ffno.h FFNO_FF_SCHED_* (0: the library's choice)
        # cap on the persistent workgroups of a feed-forward chain launch (ffno_ff_opts.max_workgroups and the field of the layer
        # descriptors; 0 = one per CU, the library's choice): for callers that share the device with another stream
        self.ff_max_workgroups = 0
        self.x3_min_lines = 1
        # workgroups (= partial slices) of the weight-gradient kernel: one per CU at width 64 (eight waves each), two per CU at
        # width 32 (four waves each: measured 157 -> 165 steps/s at 72^3 x 32 together with three chain workgroups per CU)
        self.ff_wgrad_slices = int(os.environ.get("FFNO_FF_WGRAD_SLICES", "512" if width == 32 else "256"))
        # one C call per layer and direction (ffno_layer_fwd / ffno_layer_bwd: paired branches + feed-forward) instead of two /
        # three; per-kernel timing (a timer attached) needs the individual calls
        self.use_layer_calls = True
        # forward passes that save nothing (predict / validation / rollout) through the two-launch INFERENCE layer (csrc/infer.hip:
        # mixed spectra between the launches, no branch image in memory) where the library takes the shape and the launch is large
        # enough to fill the chip; small launches (a rollout at batch 1) stay on the latency kernels of the training path
        self.use_infer_layer = os.environ.get("FFNO_INFER_LAYER", "1") != "0"
        self.infer_self_range = True      # inference layers of axis length <= 64 scale every line from its own maximum (no range words)
        self.use_infer_stack = os.environ.get("FFNO_INFER_STACK", "1") != "0"      # ... and run as one persistent launch where they can
        self.infer_stack_any_batch = False      # (see _stack_pays)
        # <= 16 modes: the fused split kernels read their DFT-matrix fragments from the precomputed tables too (round 6; the many-mode
        # and latency kernels always did) -- same values, results bit-identical; 0 = every wave rebuilds them from the twiddles
        self.x3_dft_tables = os.environ.get("FFNO_X3_DFT_TABLES", "1") != "0"
        self.x3_tile_lines = int(os.environ.get("FFNO_X3_TILE_LINES", "0"))      # ffno_fused_branch.tile_lines: 0 = the library's choice, 8 / 16
        self.infer_min_lines = None      # lines per axis pair from which the inference layer is used (None: more than 4 per CU)
        self._issue_stream = None   # torch stream object the next launches go to (None = current stream)
        self.timer = None   # optional KernelTimer (bench.py): HIP-event timing of individual launches
        # fp16x2 packs: max |W| of what goes into them is folded at every n-th rebuild of the derived operands (0 = never) and
        # looked at one rebuild later -- with 1 (the default since round 4: three 3-us launches per step) an overflowing weight
        # is reported after at most one more optimiser step (include/ffno.h: |W| < 65504 is a hard precondition of the packs)
        self.weight_range_check_every = 1
        # storage format of the activation tensors in HBM (include/ffno.h "Storage formats"): "fp32" = the parity path (the
        # reference is precision: 32); "bf16" = the bf16 storage twins of the hot kernels -- half the activation bytes, results
        # rounded to bf16 wherever a tensor is stored (a throughput variant with its own tolerance).  Available on the fused split
        # kernels of every BASELINE shape (2-D width 64 up to 64 modes, 3-D width 32), 2-layer feed-forward, no fork heads /
        # LayerNorm / dropout.
        self.storage = os.environ.get("FFNO_STORAGE", "fp32")

    def _bf16(self) -> bool:
        if self.storage not in ("fp32", "bf16"):
            raise ValueError("storage must be 'fp32' or 'bf16', got %r" % (self.storage,))
        return self.storage == "bf16"

    def _st(self) -> int:
        return 1 if self._bf16() else 0      # FFNO_STORE_F32 / FFNO_STORE_BF16

    def _conc(self) -> bool:
        return bool(self._ffx() and not self.use_fork
                    and self.mode != "no-fourier" and self.spectral == "factorized")

    @staticmethod
    def _schedule(fused, views):
        """(views that run on their own first, (a, b) = the two views that share paired launches | None).
        Preferred: the last two FUSED views (one launch); else the last two STAGED views whose stage kernels share a
        template instance (three paired launches: 256 x 256 with 32 / 64 modes)."""
        n = len(fused)
        fz = [w for w in range(n) if fused[w]]
        sg = [w for w in range(n) if not fused[w]]
        pair = None
        if len(fz) >= 2:
            pair = (fz[-2], fz[-1])
            # a many-mode axis (4-line tiles, the whole weight set streamed per tile) does not share a launch with a <= 16-mode
            # axis: the pair kernel would drag the short axis onto 4-line tiles too (airfoil, 32 x 16 modes: 91 vs 99 steps/s)
            if (views[pair[0]].K > 16) != (views[pair[1]].K > 16):
                pair = None
        elif len(sg) >= 2:
            a, b = sg[-2], sg[-1]
            rt = lambda v: (2 * v.K + 31) // 32      # noqa: E731
            if rt(views[a]) == rt(views[b]) <= 4:
                pair = (a, b)
        if pair is None:
            return list(range(n)), None
        return [w for w in range(n) if w not in pair], pair

    def _pair(self, name, ws, v0, v1, src, dst0, dst1, resid0, save0, save1, planes0, planes1, fwd: bool, st, acc0: int = 0,
              fused: bool = True, x3: bool = False, rin=None, rout=None, resid1=None):
        """Branches of views v0 and v1 side by side -- ONE launch when both are fused, three paired stage launches else:
        dst0 (+)= [resid0 +] branch0(src)  (``acc0``: accumulate into dst0),  dst1 = [resid1 +] branch1(src).
        rin / rout: range words of src / of the two outputs (both branches fold into ``rout``)."""
        lib = _lib.get_lib()
        ck_f, ck_i, conj = (0, 1, 0) if fwd else (1, 0, 1)
        if not fused:
            sv0 = save0 if save0 is not None else ws.SD
            sv1 = save1 if save1 is not None else ws.SD2
            ba = _capi.FusedBranch(_p(src), _p(dst0), resid0, _p(sv0), _p(planes0), _p(self._twiddle(v0.L)),
                                   v0.Bv, v0.Mv, v0.Nv, v0.K, v0.a01, acc0, 0, 0, None, rout if x3 else None)
            bb = _capi.FusedBranch(_p(src), _p(dst1), resid1, _p(sv1), _p(planes1), _p(self._twiddle(v1.L)),
                                   v1.Bv, v1.Mv, v1.Nv, v1.K, v1.a01, 0, 0, 0, None, rout if x3 else None)
            fn = lib.ffno_spectral_x3_staged_pair if x3 else lib.ffno_spectral_staged_pair      # x3: planes are the packed sets
            self._k("spectral_staged_pair" + ("" if fwd else "(adj)"), fn, ctypes.byref(ba),
                    ctypes.byref(bb), _p(ws.SY), _p(ws.SY2), self.C, ck_f, ck_i, conj, st)
            if not x3:      # the fp32 stage kernels do not fold their output maxima
                self._fold(dst0, rout, st)
                self._fold(dst1, rout, st)
            return
        ba = self._branch(v0, src, dst0, resid0, save0, planes0, acc0, x3, fwd, rin, rout)
        bb = self._branch(v1, src, dst1, resid1, save1, planes1, 0, x3, fwd, rin, rout)
        if x3:       # planes0 / planes1 are the packed split-bf16 sets
            self._k(name, lib.ffno_spectral_x3_pair, ctypes.byref(ba), ctypes.byref(bb), self.C, ck_f, ck_i, conj,
                    X3_INTERLEAVE, st)
            return
        self._k(name, lib.ffno_spectral_fused_pair, ctypes.byref(ba), ctypes.byref(bb), self.C, ck_f, ck_i, conj, st)

    def _branch(self, v, src, dst, resid, save, planes, acc, x3=False, fwd=True, rin=None, rout=None):
        """Branch descriptor; with fp16x2 packs the x3 kernel scales its spectrum tile from the range word of ``src``."""
        fmt = int(getattr(v, "x3fmt", 1)) if (x3 and planes is not None and self._x3_h2()) else 0      # ffno.h FFNO_PLANES_*
        dft = self._dft_frags(v.L, v.K, fwd) if (fmt and (v.K > 16 or self.x3_dft_tables)) else None
        return _capi.FusedBranch(_p(src), _p(dst), resid, _p(save), _p(planes), _p(self._twiddle(v.L)), v.Bv, v.Mv, v.Nv, v.K,
                                 v.a01, acc, fmt, int(self.x3_tile_lines), rin if fmt else None, rout, self._st(), 0, _p(dft))

    def _dft_frags(self, L: int, K: int, fwd: bool):
        """DFT-matrix fragment table of the many-mode fused kernel for (axis length, modes, direction): built once per engine and
        device on the launch stream (ffno_spectral_x3_dft_frags), then shared by every line, layer and step."""
        key = (L, K, bool(fwd))
        tab = self._dft_tabs.get(key)
        if tab is None:
            lib = _lib.get_lib()
            nb = int(lib.ffno_spectral_x3_dft_frags_bytes(L, K)) if self.C in (32, 64) else 0      # 0: no table for this shape
            if nb:
                tab = torch.empty(nb // 4, dtype=torch.int32, device=self.device)
                ck_f, ck_i = (0, 1) if fwd else (1, 0)
                self._k("dft_frags", lib.ffno_spectral_x3_dft_frags, _p(self._twiddle(L)), L, K, ck_f, ck_i, _p(tab),
                        _lib.current_stream(self.device))
            else:
                tab = False
            self._dft_tabs[key] = tab
        return tab if tab is not False else None

    # ---- range words (include/ffno.h "Range words"): one uint32 per (tensor kind, layer) in ws.RW ---------------------
    def _ranged(self) -> bool:
        """True when some kernel of this configuration cuts operands into fp16 planes (then every producer on the path
        records its output maximum and every fp16x2 consumer scales from it -- on the device)."""
        return bool((self._ffx() and self._h2()) or (self.use_x3 and self.spectral == "factorized" and self._x3_h2()))

    def _rw(self, ws, kind: str, l: int):
        if not self._ranged():
            return None
        return ctypes.c_void_p(ws.RW.data_ptr() + 4 * (ws.rw_kinds[kind] * (self.L + 1) + l))

    def _fw_words(self, ws, l0: int):
        """Range words of the tensors whose spectra the Fourier-weight-gradient launch contracts, from layer l0 on: the layer inputs
        (kind x) and the feed-forward data gradients (kind d) -- (x words, d words) as device pointers, or (None, None): then the
        launch keeps the bf16x3 arithmetic (any range).  fp16x2 (three MFMAs per product instead of six) when the spectral kernels
        run on fp16x2 packs, i.e. when every producer on the path records its maximum (`fw_grad_split`)."""
        split = self.fw_grad_split or self.x3_mix_split
        if split not in ("fp16x2", "bf16x3"):
            raise ValueError("fw_grad_split must be 'fp16x2', 'bf16x3' or None (follow x3_mix_split), got %r" % (split,))
        if not (split == "fp16x2" and self._ranged() and self.spectral == "factorized" and self.use_x3 and self._x3_h2()):
            return None, None
        base = ws.RW.data_ptr()
        return (ctypes.c_void_p(base + 4 * (ws.rw_kinds["x"] * (self.L + 1) + l0)),
                ctypes.c_void_p(base + 4 * (ws.rw_kinds["d"] * (self.L + 1) + l0)))

    def _fold(self, t, word, st):
        """Producers that do not record their output maximum themselves: one more pass over the tensor."""
        if word is None or t is None:
            return
        self._k("amax", _lib.get_lib().ffno_amax, _p(t), t.numel(), word, st)

    def _x3_h2(self) -> bool:
        if self.x3_mix_split not in ("fp16x2", "bf16x3"):
            raise ValueError("x3_mix_split must be 'fp16x2' or 'bf16x3', got %r" % (self.x3_mix_split,))
        return self.x3_mix_split == "fp16x2"

    def _ffx(self) -> bool:
        return bool(not self.general_ff and _lib.get_lib().ffno_ffx_supported(self.C, self.H))

    def _h2(self) -> bool:
        if self.ff_split not in ("fp16x2", "bf16x3"):
            raise ValueError("ff_split must be 'fp16x2' or 'bf16x3', got %r" % (self.ff_split,))
        return self.ff_split == "fp16x2"

    # ---- the split feed-forward kernels of either family (same operators; own weight packs) ----
    def _ffs_fwd2(self, s, s2, s_sum, resid, l0, b0, b1, out, mask, P, st, rin=None, rout=None):
        lib = _lib.get_lib()
        fn = lib.ffno_ffh_fwd2 if self._h2() else lib.ffno_ffx_fwd2
        o = _capi.FfOpts(rin, rout, int(self.ff_max_workgroups), int(self.ff_schedule), self._st())
        self._k("ff_fwd", fn, _p(s), _p(s2), _p(s_sum), _p(resid), _p(l0.fx[0]), _p(b0), _p(l0.fx[1]), _p(b1), _p(out), _p(mask),
                P, self.C, self.H, ctypes.byref(o), st)

    def _ffs_bwd2(self, g, g2, g_sum, mask, l0, ds, P, st, rin=None, rout=None):
        lib = _lib.get_lib()
        fn = lib.ffno_ffh_bwd_data2 if self._h2() else lib.ffno_ffx_bwd_data2
        o = _capi.FfOpts(rin, rout, int(self.ff_max_workgroups), int(self.ff_schedule), self._st())
        self._k("ff_bwd_data", fn, _p(g), _p(g2), _p(g_sum), _p(mask), _p(l0.fx[2]), _p(l0.fx[3]), _p(ds),
                P, self.C, self.H, ctypes.byref(o), st)

    def _ffs_wgrad(self, s, g, l0, b0, part, P, nsplit, st, rs=None, rg=None):
        lib = _lib.get_lib()
        if self._h2():
            self._k("ff_bwd_weights_partial", lib.ffno_ffh_bwd_weights_partial, _p(s), _p(g), _p(l0.fx[0]), _p(b0), _p(l0.fx[2]),
                    _p(part), P, self.C, self.H, nsplit, rs, rg, self._st(), st)
        else:
            self._k("ff_bwd_weights_partial", lib.ffno_ffx_bwd_weights_partial, _p(s), _p(g), _p(l0.fx[0]), _p(b0), _p(l0.fx[2]),
                    _p(part), P, self.C, self.H, nsplit, st)

    def _k(self, name, fn, *args):
        """Enqueue one C-ABI call; with a timer attached, bracket it with HIP events on the launch stream."""
        t = self.timer
        if t is not None and hasattr(t, "seen"):
            t.seen(name, fn, args)          # bench.py: capture (entry point, arguments) for replay timing
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
        # the same storage as last time (a module re-binds its detached parameters on every forward): validated already
        sig = tuple(params[n].data_ptr() for n in self.param_names) if all(n in params for n in self.param_names) else None
        if sig is not None and self.params and sig == getattr(self, "_bind_sig", None):
            self.params = {n: params[n] for n in self.param_names}
            return
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
        self._bind_sig = sig
        if dev != self.device:
            self.device = dev
            self._alloc_param_buffers()
        self._ptr_sig = None

    def _alloc_param_buffers(self):
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        self.gflat = torch.zeros(self.n_params, **f32)
        n_eff = sum(l.rows * l.cols for l in self.linears.values()) if self.wnorm else 0
        self.weff_flat = torch.empty(max(n_eff, 1), **f32)
        self.gweff_flat = torch.zeros(max(n_eff, 1), **f32)
        self.fold = torch.zeros(self.O * (self.C + 1), **f32)
        n_ff = sum(l.rows * l.cols for p, l in self.linears.items() if "_ff." in p)
        self.wt_flat = torch.empty(max(n_ff, 1), **f32)
        # mode-major weight planes: planes[set][w] = (wp, wpt), each 2*K_w*C*C floats
        plane_modes = [2 * self.K * self.K] if self.spectral == "plus" else list(self.Ks)   # plus: one joint (ky, kx') set
        self.planes = [[(torch.empty(2 * K * self.C * self.C, **f32), torch.empty(2 * K * self.C * self.C, **f32))
                        for K in plane_modes] for _ in range(max(len(self._fw_sets), 1))]
        # packed split-bf16 twins of the planes for the x3 branch kernel: xplanes[set][w] = (forward, adjoint) or None
        lib = _lib.get_lib()
        self.xplanes = [[None] * len(plane_modes) for _ in self.planes]
        if self.spectral == "factorized" and self.mode == "full":
            for si in range(len(self.planes)):
                for w, K in enumerate(plane_modes):
                    nb = int(lib.ffno_spectral_x3_pack_bytes(self.C, K))      # 0 = no split kernel for (C, K)
                    if nb:
                        self.xplanes[si][w] = tuple(torch.empty(nb // 4, dtype=torch.int32, device=dev) for _ in range(2))
        self._x3pack_sig = None
        self._prep_sig = None
        self._ws_key = None
        self._ws_cache = {}
        self._tw = {}
        self._dft_tabs = {}

    def grad_view(self, name: str) -> torch.Tensor:
        o = self._offsets[name]
        return self.gflat[o:o + int(np.prod(self.param_shapes[name]))].view(self.param_shapes[name])

    def _refresh_pointers(self):
        sig = tuple(self.params[n].data_ptr() for n in self.param_names) + (self.ff_split,)
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
            if "_ff." not in pfx or self.general_ff:      # (the fp32 fused feed-forward's backward reads transposed weights)
                continue
            n = lin.rows * lin.cols
            lin.wt = self.wt_flat[toff:toff + n].view(lin.cols, lin.rows)
            toff += n
            tdescs.append(_capi.TrDesc(lin.weff.data_ptr(), lin.wt.data_ptr(), lin.rows, lin.cols))
        # split-bf16 operand packs of every feed-forward block (include/ffno.h "packed"): forward A1/A2, backward A1/A2
        lib = _lib.get_lib()
        blocks = list(dict.fromkeys(self.ff_prefix + self.fc_prefix))
        self._n_fx = 0
        if lib.ffno_ffx_supported(self.C, self.H) and not self.general_ff:
            words = int((lib.ffno_ffh_pack_bytes if self._h2() else lib.ffno_ffx_pack_bytes)(self.C, self.H)) // 4
            if getattr(self, "_fx_flat", None) is None or self._fx_flat.numel() != 4 * words * len(blocks) \
                    or self._fx_flat.device != self.device:
                self._fx_flat = torch.empty(4 * words * len(blocks), dtype=torch.int32, device=self.device)
            fdescs = []
            for bi, pfx in enumerate(blocks):
                l0, l1 = self.linears[pfx + "layers.0.0."], self.linears[pfx + "layers.1.0."]
                pk = [self._fx_flat[(4 * bi + q) * words:(4 * bi + q + 1) * words] for q in range(4)]
                l0.fx = pk      # both linears of the block see the four packs
                l1.fx = pk
                C_, H_ = self.C, self.H
                for (src, sh, sc, ty), dst in zip(((l0.weff, C_, 1, 1), (l1.weff, 1, H_, 2), (l1.weff, 1, H_, 1),
                                                   (l0.weff, C_, 1, 2)), pk):
                    fdescs.append(_capi.FxPackDesc(src.data_ptr(), dst.data_ptr(), sh, sc, ty, 0))
            self._n_fx = len(fdescs)
            arr = (_capi.FxPackDesc * len(fdescs))(*fdescs)
            self._fx_dev = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(self.device)
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
    def _views(self, B: int, Sp: Sequence[int]) -> List[_View]:
        """views[w] = the axis mixed by fourier_weight[w], as a [Bv, Mv, Nv, C] view of the (padded) buffer."""
        C = self.C
        if self.spectral == "plus":
            M, N = Sp
            v = _View(B, M, N, 0, self.K, C)          # geometry of the last-axis transforms
            v.R, v.K2 = B, 2 * self.K * self.K         # rows per mode = samples; retained (ky, kx') modes
            v.spec_y = v.spec                          # y-transformed spectrum [K][B*M][2][C]
            v.spec = v.K2 * B * 2 * C                  # 2-D spectrum [K2][B][2][C] (what is saved for the weight gradient)
            return [v]
        if self.nd == 2:
            M, N = Sp
            if self.first_axis_first:      # mesh_2d.py:71-75,92-96: weight 0 <-> x (first axis), weight 1 <-> y (last axis)
                return [_View(B, M, N, 1, self.Ks[0], C), _View(B, M, N, 0, self.Ks[1], C)]
            return [_View(B, M, N, 0, self.Ks[0], C), _View(B, M, N, 1, self.Ks[1], C)]
        X, Y, Z = Sp
        return [_View(B, X, Y * Z, 1, self.Ks[0], C),       # x: lines (b, y, z), stride Y*Z*C
                _View(B * X, Y, Z, 1, self.Ks[1], C),       # y: lines (b, x, z), stride Z*C
                _View(B, X * Y, Z, 0, self.Ks[2], C)]       # z: contiguous lines (b, x, y)

    def _sched_sig(self):
        """Everything a workspace's buffers and schedule decisions (deferred weight-gradient launch, lazy input sums, slice counts)
        are derived from besides the geometry: part of the workspace key, so an attribute switched on a live engine gets a
        workspace laid out for it instead of decisions frozen for the old value."""
        return (self._h2(), self._ranged(), bool(self.ff_wgrad_deferred), str(self.ff_lazy_sums), int(self.ff_wgrad_slices),
                bool(self.use_x3), bool(self.use_fused))

    def _workspace(self, B: int, S: Tuple[int, ...], save: bool):
        key = (B, tuple(S), bool(save), self._ffx(), self._conc(), self.general_ff, self._bf16(), self._sched_sig())
        if self._ws_key == key:
            return self._ws
        cache = self.__dict__.setdefault("_ws_cache", {})   # a few recent geometries (train batch / validation batch /
        if key in cache:                                     # rollout batch alternate in a real run): no re-allocation
            self._ws, self._ws_key = cache[key], key
            return self._ws
        lib = _lib.get_lib()
        dev, C, H, L, O = self.device, self.C, self.H, self.L, self.O
        Sp = tuple(s + self.pad for s in S)
        P_in = B * int(np.prod(S))            # unpadded pixels (inputs / outputs)
        P = B * int(np.prod(Sp))              # pixels of the activation buffers
        f32 = dict(dtype=torch.float32, device=dev)
        act = dict(dtype=torch.bfloat16 if self._bf16() else torch.float32, device=dev)     # activation tensors of the layer stack
        ns = L if save else 1
        ws = type("WS", (), {})()
        ws.P, ws.P_in, ws.Sp = P, P_in, Sp
        ws.views = self._views(B, Sp)
        ws.padmap = None
        if self.pad:
            s3 = (1,) * (3 - self.nd) + tuple(S)
            p3 = (1,) * (3 - self.nd) + tuple(Sp)
            ws.padmap = _capi.PadMap((ctypes.c_int32 * 3)(*s3), (ctypes.c_int32 * 3)(*p3))
        ws.X = torch.empty(P, C, **act)
        ws.Blast = torch.empty(P, C, **act)
        ws.Y = torch.empty(P_in * O, **f32)
        ws.S = torch.empty(ns, P, C, **act)
        nv = len(ws.views)
        ws.SXall = [torch.empty(ns, v.spec, **f32) for v in ws.views]      # forward spectra, per axis, layer-major
        ws.SY = torch.empty(max(v.spec for v in ws.views), **f32)
        ws.SD = torch.empty(max(v.spec for v in ws.views), **f32)    # scratch spectrum of the staged path
        if self._conc():
            ws.SD2 = torch.empty_like(ws.SD)                          # second branch of a paired STAGE launch
            ws.SY2 = torch.empty_like(ws.SY)
            ws.T = torch.empty(P, C, **act)                           # output of the second branch of a paired launch
            if save:
                ws.G1 = torch.empty(P, C, **act)                      # ... and of the second adjoint branch
        if self.spectral == "plus":
            ws.SYa = torch.empty(ws.views[0].spec_y, **f32)           # last-axis spectra on either side of the x transform
            ws.SYb = torch.empty(ws.views[0].spec_y, **f32)
            ws.CW = torch.empty(int(lib.ffno_cdft_rows_ws_floats(B, C, self.K, self.K)), **f32)   # first-axis DFT scratch
        ws.mask_words = int(lib.ffno_ff_mask_words(P, H))
        # range words, one per (tensor kind, layer): x = layer inputs, s = spectral-branch outputs (feed-forward inputs),
        # g = running gradient (both buffers), d = feed-forward data gradients, t / f = LayerNorm / fork-head gradients
        ws.rw_kinds = {"x": 0, "s": 1, "g": 2, "d": 3, "t": 4, "f": 5}
        ws.RW = torch.zeros(len(ws.rw_kinds) * (L + 1), dtype=torch.int32, device=dev)
        if self.layer_norm:
            ws.TL = torch.empty(ns, P, C, **f32)                     # feed-forward outputs before the LayerNorm
            ws.LNS = torch.empty(ns, P, 2, **f32)                    # {mean, rstd} per pixel
            if self.use_fork:            # the forecast feed-forwards end in their own LayerNorm (feedforward.py:18-19)
                ws.TLF = torch.empty(ns, P, C, **f32)
                ws.LNSF = torch.empty(ns, P, 2, **f32)
            if save:
                ws.DT = torch.empty(P, C, **f32)
                ws.lnpart = torch.empty(2 * C * int(lib.ffno_layernorm_nsplit(P)), **f32)
                if self.use_fork:
                    ws.DTF = torch.empty(P, C, **f32)
        if self.use_fork:
            ws.F = torch.empty(ns, P, C, **f32)                 # forecast_ff outputs f_l (inputs of the shared head)
            ws.YL = torch.empty(L, P_in * O, **f32)             # per-layer head outputs (forecast_list)
        if self.general_ff:
            # kept hidden activations of the general feed-forward path, per block kind / saved layer / hidden layer
            kinds = ["backcast"] + (["forecast"] if self.use_fork else [])
            ws.HG = {kind: [[torch.empty(P, H, **f32) for _ in range(self.n_ff - 1)] for _ in range(ns)] for kind in kinds}
            if save:
                ws.DA = [torch.empty(P, H, **f32) for _ in range(2)]
                ws.glpart = torch.empty(int(lib.ffno_glin_wgrad_partial_floats(P, H, H)), **f32)
        fp32_ff = not self._ffx() and not self.general_ff      # the fp32 fused feed-forward keeps h / dh in HBM
        if save and self.use_fork:
            ws.HF = torch.empty(ns, P, H, **f32) if fp32_ff else [None] * ns
            ws.MASKF = torch.zeros(ns, ws.mask_words, dtype=torch.int32, device=dev)
            ws.DSF = torch.empty(P, C, **f32)
            ws.GF = torch.empty(P, C, **f32)
            ws.redl = torch.empty(O * (C + 1), **f32)
        if save:
            ws.Hbuf = torch.empty(ns, P, H, **f32) if fp32_ff else [None] * ns
            ws.MASK = torch.zeros(ns, ws.mask_words, dtype=torch.int32, device=dev)
            ws.DH = [torch.empty(P, H, **f32) if fp32_ff else None for _ in range(2)]   # ping-pong pair (fp32 feed-forward: stored dh)
            ws.DS = torch.empty(P, C, **act)
            # running gradient: a ping-pong pair -- or one buffer per layer when the weight-gradient launches are deferred to
            # the end of the pass (they read every layer's summed gradient then)
            ws.defer_wgrad = bool(self.ff_wgrad_deferred and self._ffx() and self._h2() and self._ranged() and not self.share_fork
                                  and not self.use_fork and not self.layer_norm and not self.general_ff
                                  and (C, H) in ((64, 256), (32, 128)) and self.mode != "no-fourier")
            # (what the deferred launch costs in memory: L - 1 more gradient buffers, and with lazy sums L more branch images --
            #  2 L P C words together, 1.7 GB at the headline shape.  Beyond FFNO_FF_DEFER_MAX_BYTES (default 16 GiB of the 288)
            #  the engine keeps the ping-pong pair and per-layer launches: ADVICE r04.)
            extra = 2 * L * P * C * (2 if self._bf16() else 4)
            if extra > int(os.environ.get("FFNO_FF_DEFER_MAX_BYTES", str(16 << 30))):
                ws.defer_wgrad = False
            ws.G = [torch.empty(P, C, **act) for _ in range(L + 1 if ws.defer_wgrad else 2)]
            if self.ff_lazy_sums not in ("s", "0", ""):
                raise ValueError("ff_lazy_sums must be 's' or '0', got %r" % (self.ff_lazy_sums,))
            ws.lazy_sums = "s" if (ws.defer_wgrad and self.ff_lazy_sums == "s" and self._conc() and (C, H) == (64, 256)) else ""
            if ws.lazy_sums:
                ws.TS = [torch.empty(P, C, **act) for _ in range(L)]         # second branch output of every layer
            ws.SDall = [torch.empty(L, v.spec, **f32) for v in ws.views] if self.mode == "full" else None
            ws.nsplit_ff = max(1, min(int(self.ff_wgrad_slices), (P + 127) // 128))
            cus = torch.cuda.get_device_properties(dev).multi_processor_count if dev.type == "cuda" else 256
            ws.nsplit_ffm = max(1, min(ws.nsplit_ff, (FF_WGRAD_ROUNDS * (3 if H <= 128 else 1) * cus) // L))
            ws.wg_sig, ws.wg_table = None, None
            ws.ffpart = torch.empty(int(lib.ffno_ff_wgrad_partial_floats(C, H, ws.nsplit_ff)), **f32)
            # split-bf16 path with per-layer feed-forwards: every layer keeps its own slices and ONE batched launch
            # reduces them all at the end of the backward pass (24 kernel boundaries less per step)
            ws.defer_reduce = bool(self._ffx() and not self.share_fork)
            if ws.defer_reduce:
                njobs = L * (2 if self.use_fork else 1)
                ws.ffparts = torch.empty(njobs, ws.ffpart.numel(), **f32)
                ws.red_sig, ws.red_table = None, None
            fwg = int(os.environ.get("FFNO_FW_GRAD_WGS", "512"))      # workgroups of a Fourier-weight-gradient launch (slices x modes)
            ws.nsplit_fw = [max(1, min(max(1, fwg // v.K), (L * v.R + 63) // 64)) for v in ws.views]
            if self.spectral == "plus":
                ws.nsplit_fw = [1]
            ws.fwpart = [[torch.empty(ws.nsplit_fw[w] * 2 * (ws.views[w].K2 if self.spectral == "plus" else ws.views[w].K)
                                      * C * C, **f32) for w in range(nv)]
                         for _ in range(max(len(self._fw_sets), 1))]
            ws.nsplit_lift = max(1, min(1024, (P_in + 127) // 128))     # all slices co-resident (16 KB LDS per workgroup)
            ws.liftpart = torch.empty(ws.nsplit_lift * C * (self.Cin + 1), **f32)
            ws.nsplit_head = max(1, min(256, (P_in + 255) // 256))
            ws.headpart = torch.empty(ws.nsplit_head * O * (C + 1), **f32)
            ws.red = torch.empty(O * (C + 1), **f32)
        self._ws, self._ws_key = ws, key
        cache[key] = ws
        while len(cache) > 4:
            cache.pop(next(iter(cache)))
        return ws

    def _forward_workspace(self, B: int, S: Tuple[int, ...]):
        """The workspace the last ``forward(save_for_backward=True)`` filled.  A schedule / arithmetic attribute switched since then
        (``ff_split``, ``x3_mix_split``, ``ff_wgrad_deferred``, ``ff_lazy_sums``, ``use_x3``, ``use_fused``, ``storage``, slice
        counts) would select ANOTHER workspace -- whose saved tensors were never written: refuse instead of differentiating
        uninitialised memory."""
        ws = getattr(self, "_saved_ws", None)
        if ws is None:
            raise RuntimeError("backward() needs a preceding forward(save_for_backward=True)")
        key = (B, tuple(S), True, self._ffx(), self._conc(), self.general_ff, self._bf16(), self._sched_sig())
        if key != self._saved_ws_key:
            raise RuntimeError("an engine attribute that shapes the workspace (ff_split / x3_mix_split / ff_wgrad_deferred / "
                               "ff_lazy_sums / ff_wgrad_slices / use_x3 / use_fused / storage) changed between forward() and "
                               "backward(): run the forward pass again")
        return ws

    def weights_changed(self):
        """Tell the engine that parameter memory was written behind PyTorch's back (a raw-pointer kernel such as the fused
        AdamW of FFNOTrainer): the derived operands (weight-norm products, packed fragments, folded head) are rebuilt by the
        next forward.  In-place torch ops on the BOUND tensors or on anything that shares their version counter (the module
        mirrors and FFNOTrainer bind ``parameter.detach()``: optimizer.step(), load_state_dict, p.copy_() are seen) need no
        call; writes through another alias of the same memory (a view of a flat buffer, ``p.data``'s source) do."""
        self._prep_sig = None

    def _prepare_weights(self, st):
        lib = _lib.get_lib()
        self._refresh_pointers()
        # derived operands only depend on the parameters and the arithmetic choices: an inference loop (rollout: 10-100 forwards
        # on the same weights) prepares them once -- five launches less per forward
        try:
            sig = (tuple((t.data_ptr(), t._version) for t in self.params.values()), self.ff_split, self.x3_mix_split,
                   tuple(self._x3_fmt or ()), self.use_x3, getattr(st, "value", st))
        except RuntimeError:      # inference tensors (created under torch.inference_mode()) have no version counter:
            sig = None            # nothing to compare -- the operands are rebuilt by every forward, as before round 3
        if sig is not None and sig == getattr(self, "_prep_sig", None):
            return
        self._prep_sig = sig
        if self._desc_dev is not None:
            self._k("weightnorm_fwd", lib.ffno_weightnorm_fwd, _p(self._desc_dev), self._n_desc, self._max_rows, st)
        if self._ffx() and self._n_fx:
            self._k("ffx_pack", lib.ffno_ffh_pack if self._h2() else lib.ffno_ffx_pack, _p(self._fx_dev), self._n_fx, self.C,
                    self.H, st)
        for i, names in enumerate(self._fw_sets if self.spectral == "plus" else []):
            self._k("fw2d_pack", lib.ffno_fw2d_pack, _p(self.params[names[0]]), _p(self.params[names[1]]),
                    _p(self.planes[i][0][0]), _p(self.planes[i][0][1]), self.C, self.K, st)
        if self.spectral != "plus" and self._fw_sets:      # every (weight set, axis) tensor in ONE launch
            sig = tuple((self.params[n].data_ptr(), self.planes[i][w][0].data_ptr())
                        for i, names in enumerate(self._fw_sets) for w, n in enumerate(names))
            if sig != getattr(self, "_fwpack_sig", None):
                descs = [_capi.FwPackDesc(self.params[n].data_ptr(), self.planes[i][w][0].data_ptr(),
                                          self.planes[i][w][1].data_ptr(), self.Ks[w], int(self.spectral == "dct"))
                         for i, names in enumerate(self._fw_sets) for w, n in enumerate(names)]
                arr = (_capi.FwPackDesc * len(descs))(*descs)
                self._fwpack_dev = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(self.device)
                self._fwpack_sig, self._fwpack_n = sig, len(descs)
            self._k("fw_pack", lib.ffno_fw_pack_batched, _p(self._fwpack_dev), self._fwpack_n, self.C, max(self.Ks), st)
            if self.use_x3 and any(x is not None for row in self.xplanes for x in row):
                fmt = self._x3_fmt or [0] * len(self.Ks)
                sig = tuple((self.planes[i][w][d].data_ptr(), xp[d].data_ptr(), fmt[w]) for i, row in enumerate(self.xplanes)
                            for w, xp in enumerate(row) if xp is not None for d in range(2))
                if sig != self._x3pack_sig:
                    descs = [_capi.X3PackDesc(self.planes[i][w][d].data_ptr(), xp[d].data_ptr(), self.Ks[w], fmt[w])
                             for i, row in enumerate(self.xplanes) for w, xp in enumerate(row) if xp is not None
                             for d in range(2)]
                    arr = (_capi.X3PackDesc * len(descs))(*descs)
                    self._x3pack_dev = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(self.device)
                    self._x3pack_sig, self._x3pack_n = sig, len(descs)
                    self._x3pack_maxk = max(d.K for d in descs)
                self._k("fw_pack_x3", lib.ffno_spectral_x3_pack, _p(self._x3pack_dev), self._x3pack_n, self.C, self._x3pack_maxk, st)
        o0, o1 = self.linears["out.0."], self.linears["out.1."]
        self._k("head_fold", lib.ffno_head_fold, _p(o0.weff), _p(self.params["out.0.bias"]), _p(o1.weff),
                _p(self.params["out.1.bias"]), _p(self.fold), self.C, HEAD_DIM, self.O, st)
        self._check_weight_range(st)

    # ---- the one operand the range words do not cover: the WEIGHTS inside the fp16x2 packs ---------------------------------
    def _check_weight_range(self, st):
        """Activations, spectra and gradients are brought into the half format's range per launch (range words).  Weights are
        packed as they are: one at or above 65504 would overflow its fp16 planes.  Every ``weight_range_check_every``-th time
        the derived operands are rebuilt, max |W| of everything that goes into an fp16x2 pack is folded on the device
        (ffno_amax) and copied to pinned host memory asynchronously; the value is looked at the next time round -- no
        synchronisation, the error arrives a few steps late but it arrives (the loss is NaN by then, and says why)."""
        every = int(self.weight_range_check_every)
        fp16_ff, fp16_mix = self._ffx() and self._h2(), bool(self._x3_fmt and any(self._x3_fmt))
        if every <= 0 or not (fp16_ff or fp16_mix):
            return
        pend = self.__dict__.get("_wr_pending")
        if pend is not None:
            host, ev = pend
            if ev is None or ev.query():
                self._wr_pending = None
                wmax = float(host.view(torch.float32)[0])
                if 0.0 < wmax < 2.0 ** -9 and not self.__dict__.get("_wr_warned"):
                    import warnings
                    self._wr_warned = True
                    warnings.warn(f"every weight that goes into the split-fp16 packs is below {wmax:.3g}: the packs keep a fixed "
                                  f"absolute error of 2^-36 there, i.e. relative accuracy degrades as the weights shrink "
                                  f"(1e-6 at 2e-5); ff_split / x3_mix_split = 'bf16x3' is exact at any magnitude", RuntimeWarning)
                if not wmax < 65504.0:
                    raise FloatingPointError(
                        f"a weight of magnitude {wmax:.3g} does not fit the split-fp16 packs (|W| < 65504): set "
                        f"engine.ff_split = engine.x3_mix_split = 'bf16x3' (FFNO_FF_SPLIT / FFNO_X3_MIX_SPLIT) -- any fp32 range")
        self._wr_count = self.__dict__.get("_wr_count", 0) + 1
        if (self._wr_count - 1) % every or self.__dict__.get("_wr_pending") is not None:
            return
        lib = _lib.get_lib()
        if self.__dict__.get("_wr_word") is None or self._wr_word.device != self.device:
            self._wr_word = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._wr_host = torch.zeros(1, dtype=torch.int32)
            if self.device.type == "cuda":
                self._wr_host = self._wr_host.pin_memory()
        self._wr_word.zero_()
        srcs = []
        if fp16_ff:
            if self.wnorm:
                srcs.append(self.weff_flat)       # every effective linear weight, one buffer
            else:
                srcs += [self.params[p + "weight"] for p in self.linears if "_ff." in p and p + "weight" in self.params]
        if fp16_mix:
            srcs += [self.params[n] for names in self._fw_sets for n in names]
        # ONE launch for all of them (per-layer Fourier weights are 2-3 tensors per layer: the airfoil model issued 49 of these
        # 4-us launches per step until round 5); the descriptor table is rebuilt when a tensor moved
        sig = tuple((t.data_ptr(), t.numel()) for t in srcs)
        if sig != self.__dict__.get("_wr_sig"):
            arr = (_capi.AmaxDesc * len(sig))(*[_capi.AmaxDesc(ptr, n) for ptr, n in sig])
            self._wr_table = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(self.device)
            self._wr_sig = sig
        self._k("amax", lib.ffno_amax_batched, _p(self._wr_table), len(sig), max(n for _, n in sig), _p(self._wr_word), st)
        ev = None
        if self.device.type == "cuda":
            self._wr_host.copy_(self._wr_word, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            self._wr_host.copy_(self._wr_word)
        self._wr_pending = (self._wr_host, ev)

    def _ff_weights(self, l):
        fp = self.ff_prefix[l]
        l0, l1 = self.linears[fp + "layers.0.0."], self.linears[fp + "layers.1.0."]
        return l0, l1, self.params[fp + "layers.0.0.bias"], self.params[fp + "layers.1.0.bias"]

    def _fc_weights(self, l):
        fc = self.fc_prefix[l]
        l0, l1 = self.linears[fc + "layers.0.0."], self.linears[fc + "layers.1.0."]
        return l0, l1, self.params[fc + "layers.0.0.bias"], self.params[fc + "layers.1.0.bias"]

    # ---- feed-forward dispatch: split-bf16 kernels (ffx.hip) or the fp32-MFMA kernels (ff.hip) ---------------------
    def _ff_fwd(self, s, resid, l0, l1, b0, b1, out, hbuf, mask, P, st, rin=None, rout=None):
        lib = _lib.get_lib()
        if self._ffx():
            self._ffs_fwd2(s, None, None, resid, l0, b0, b1, out, mask, P, st, rin, rout)
        else:
            self._k("ff_fwd", lib.ffno_ff_fwd, _p(s), _p(resid), _p(l0.weff), _p(b0), _p(l1.weff), _p(b1), _p(out),
                    _p(hbuf), _p(mask), P, self.C, self.H, st)
            self._fold(out, rout, st)

    def _ff_bwd_data(self, g, mask, l0, l1, dh, ds, P, st, rin=None, rout=None):
        lib = _lib.get_lib()
        if self._ffx():
            self._ffs_bwd2(g, None, None, mask, l0, ds, P, st, rin, rout)
        else:
            self._k("ff_bwd_data", lib.ffno_ff_bwd_data, _p(g), _p(mask), _p(l0.wt), _p(l1.wt), _p(dh), _p(ds), P,
                    self.C, self.H, st)
            self._fold(ds, rout, st)

    def _ff_bwd_weights(self, ws, s, g, hbuf, dh, l0, l1, b0, gb0, gb1, accumulate, P, st, rs=None, rg=None):
        lib = _lib.get_lib()
        C, H = self.C, self.H
        if self._ffx() and ws.defer_reduce:
            assert not accumulate
            part = ws.ffparts[len(ws.red_jobs)]
            if getattr(ws, "wg_jobs", None) is not None:
                s2 = getattr(self, "_wg_second", None)      # second addend of s (lazy input sums)
                ws.wg_jobs.append((s.data_ptr(), g.data_ptr(), l0.fx[0].data_ptr(), b0.data_ptr(), l0.fx[2].data_ptr(),
                                   part.data_ptr(), rs.value, rg.value, s2.data_ptr() if s2 is not None else 0, 0))
            else:
                self._ffs_wgrad(s, g, l0, b0, part, P, ws.nsplit_ff, st, rs, rg)
            ws.red_jobs.append((part.data_ptr(), l0.gweff.data_ptr(), l1.gweff.data_ptr(), gb0.data_ptr(), gb1.data_ptr()))
        elif self._ffx():
            self._ffs_wgrad(s, g, l0, b0, ws.ffpart, P, ws.nsplit_ff, st, rs, rg)
            self._k("ff_bwd_weights_reduce", lib.ffno_ffx_bwd_weights_reduce, _p(ws.ffpart), _p(l0.gweff), _p(l1.gweff),
                    _p(gb0), _p(gb1), C, H, ws.nsplit_ff, accumulate, st)
        else:
            self._k("ff_bwd_weights_partial", lib.ffno_ff_bwd_weights_partial, _p(s), _p(g), _p(hbuf), _p(dh),
                    _p(ws.ffpart), P, C, H, ws.nsplit_ff, st)
            self._k("ff_bwd_weights_reduce", lib.ffno_ff_bwd_weights_reduce, _p(ws.ffpart), _p(l0.gweff), _p(l1.gweff),
                    _p(gb0), _p(gb1), C, H, ws.nsplit_ff, accumulate, st)

    # ---- the GENERAL feed-forward path (n_ff_layers != 2 or dropout > 0): one glin kernel per linear layer ------------------
    def _drop_args(self, layer: int, kind: str, k: int):
        """(p, seed) of the dropout site behind linear k of a feed-forward block in this pass: a new mask per training forward,
        layer, block kind and site; the backward pass regenerates it from the same pair."""
        if not (self.dropout > 0.0 and self._training):
            return 0.0, 0
        return self.dropout, _site_seed(self.drop_seed, self._drop_calls, layer, k + 1, int(kind == "forecast"))

    def _ffg_fwd(self, ws, prefix, kind, layer, sv, s, resid, out, P, st, rout=None):
        lib = _lib.get_lib()
        a = s
        for k in range(self.n_ff):
            last = k == self.n_ff - 1
            lin = self.linears[prefix + f"layers.{k}.0."]
            dst = out if last else ws.HG[kind][sv][k]
            p, seed = self._drop_args(layer, kind, k)
            self._k("ff_fwd", lib.ffno_glin_fwd, _p(a), _p(lin.weff), _p(self.params[prefix + f"layers.{k}.0.bias"]),
                    _p(resid) if last else None, _p(dst), P, lin.cols, lin.rows, int(not last), p, seed, st)
            a = dst
        self._fold(out, rout, st)

    def _ffg_bwd(self, ws, prefix, kind, layer, s, g, ds, accumulate, P, st, rout=None):
        """g = dL/d(feed-forward output) -> ds = dL/ds, and the parameter gradients of every linear of the block."""
        lib = _lib.get_lib()
        gv = self.grad_view
        g_cur = g
        for k in reversed(range(self.n_ff)):
            last = k == self.n_ff - 1
            lin = self.linears[prefix + f"layers.{k}.0."]
            x_in = s if k == 0 else ws.HG[kind][layer][k - 1]
            y = None if last else ws.HG[kind][layer][k]
            p, seed = self._drop_args(layer, kind, k)
            self._k("ff_bwd_weights_partial", lib.ffno_glin_bwd_weights, _p(g_cur), _p(y), _p(x_in), _p(ws.glpart), _p(lin.gweff),
                    _p(gv(prefix + f"layers.{k}.0.bias")), P, lin.cols, lin.rows, p, seed, int(accumulate), st)
            dx = ds if k == 0 else ws.DA[k & 1]
            self._k("ff_bwd_data", lib.ffno_glin_bwd_data, _p(g_cur), _p(y), _p(lin.weff), _p(dx), P, lin.cols, lin.rows, p, seed,
                    0, st)
            g_cur = dx
        self._fold(ds, rout, st)

    def _planes_for(self, si, w, adj: int, x3: bool):
        if self.mode != "full":
            return None
        return self.xplanes[si][w][adj] if x3 else self.planes[si][w][adj]

    def _can_fuse(self, views):
        """Per axis: the fused branch kernel when its LDS tile holds (C, K_axis, L_axis), else the three stage kernels
        (e.g. plasticity: x with 32 modes is staged, y / z with 12 / 8 modes are fused)."""
        lib = _lib.get_lib()
        if self.spectral == "plus":
            return [False]
        if self.spectral == "dct":
            return [False] * len(views)
        # fused = the fp32-MFMA fused kernel takes the shape, or the split (x3) fused kernels do (K <= 16 on the 16-line tile,
        # 17..64 modes on the 4-line tile of spectral_x3k: 256 x 256 grids)
        return [bool(self.use_fused and self.mode != "no-fourier"
                     and (lib.ffno_spectral_fused_supported(self.C, v.K, v.L) or self._x3_ok(w, v, views)))
                for w, v in enumerate(views)]

    def _x3_ok(self, w, v, views) -> bool:
        lib = _lib.get_lib()
        if not (self.use_x3 and self.spectral == "factorized" and self.mode != "no-fourier"):
            return False
        P4 = 4 * self.C * max(u.Bv * u.Mv * u.Nv for u in views)       # 32-bit byte offsets inside the kernel
        if not (v.R >= self.x3_min_lines and P4 < 2 ** 32 and lib.ffno_spectral_x3_supported(self.C, v.K, v.L)
                and (self.mode != "full" or self.xplanes[0][w] is not None)):
            return False
        return True

    def _use_x3(self, views, fused):
        """Per axis: the split-bf16 fused branch instead of the fp32-MFMA one (same operator, same flags)."""
        return [bool(fused[w] and self._x3_ok(w, v, views)) for w, v in enumerate(views)]

    def _infer_ok(self, ws, B, conc, pair, singles, x3pair, fused) -> bool:
        """The two-launch inference layer (ffno_layer_infer) serves this forward: both axes of a 2-D layer in one paired launch of
        the fused split kernels with fp16x2 packs and <= 16 modes, width 64 / hidden 256 on the split-fp16 feed-forward, fp32
        activations, no fork heads / LayerNorm / dropout -- and enough lines to fill the chip."""
        if not (self.use_infer_layer and conc and pair is not None and not singles and x3pair and fused[pair[0]]):
            return False
        if not (self.nd == 2 and self.mode == "full" and self.spectral == "factorized" and self._ffx() and self._h2() and self._x3_h2()
                and not self._bf16() and not self.use_fork and not self.layer_norm and not self.general_ff
                and (self.C, self.H) == (64, 256)):
            return False
        va, vb = ws.views[pair[0]], ws.views[pair[1]]
        if {va.a01, vb.a01} != {0, 1} or (va.Bv, va.Mv, va.Nv) != (vb.Bv, vb.Mv, vb.Nv):
            return False
        if any(getattr(v, "x3fmt", 0) != 1 for v in (va, vb)):
            return False
        row, col = (va, vb) if va.a01 == 0 else (vb, va)
        lib = _lib.get_lib()
        if not lib.ffno_layer_infer_supported(va.Bv, va.Mv, va.Nv, self.C, self.H, row.K, col.K):
            return False
        if self._dft_frags(row.L, row.K, True) is None or self._dft_frags(col.L, col.K, True) is None:
            return False
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count if self.device.type == "cuda" else 256
        need = self.infer_min_lines if self.infer_min_lines is not None else 4 * cus + 1
        return va.R + vb.R >= need

    def _stack_pays(self, B: int) -> bool:
        """The persistent launch runs CUs / 8 = 32 groups of 8 workgroups (16 groups of 16 for a batch <= 16), a group per image: a
        batch below the group count leaves groups idle, a larger one makes groups walk images in rounds of 32.  Measured against the
        per-layer launches (`tools/ab_stack_batch.py`, `profiles/r06_stack_batch.log`; their own row-tile rule was fixed at the end of the
        round, which is what these figures are against): -3 ... -6 % at 9 / 12 / 16 / 24 images, -9 % at 19, -2 ... -6 % at 32 / 48 / 64 /
        96, but +5 % at 40 (a last round with a quarter of the groups busy): a last round of 1..16 images keeps the per-layer launches.
        `engine.infer_stack_any_batch = True` takes it regardless."""
        if self.infer_stack_any_batch:
            return True
        if B <= 16:      # (16 members per image: 8-line / 4-row tiles, 0.98-1.00 ms per forward at 9 / 12 / 16 images against 1.01-1.06)
            return True
        tail = B % 32
        return tail == 0 or tail > 16

    def _run_infer_stack(self, ws, pair, full, st) -> bool:
        """Enqueue ffno_infer_stack over ws.X (in place; the last layer's feed-forward output lands in ws.Blast).  False: not this
        shape / device, or the persistent launch reported a placement or barrier error (then it is switched off for this engine and
        the caller runs the per-layer launches)."""
        lib = _lib.get_lib()
        C, H, L = self.C, self.H, self.L
        a, b = pair
        va, vb = ws.views[a], ws.views[b]
        row, col = (va, vb) if va.a01 == 0 else (vb, va)
        if int(lib.ffno_infer_stack_supported(va.Bv, va.Mv, va.Nv, C, H, row.K, col.K, L)) != 2:
            return False
        if not self._stack_pays(va.Bv):
            return False
        pend = getattr(ws, "stack_pending", None)
        if pend is not None:      # the error word of the PREVIOUS persistent launch on this workspace (copied without synchronising)
            host, ev = pend
            if ev is None or ev.query():
                ws.stack_pending = None
                if int(host[0]) != 0:
                    self.use_infer_stack = False
                    raise RuntimeError("ffno_infer_stack: the previous forward pass reported a workgroup without a group or a barrier "
                                       "time-out (error word %d): its result was invalid; the persistent launch is now off for this "
                                       "engine (engine.use_infer_stack)" % int(host[0]))
        if getattr(ws, "stack_sync", None) is None:
            n = int(lib.ffno_infer_stack_sync_words(va.Bv))
            ws.stack_sync = torch.zeros(n, dtype=torch.int32, device=self.device)
            ws.stack_host = torch.zeros(1, dtype=torch.int32)
            if self.device.type == "cuda":
                ws.stack_host = ws.stack_host.pin_memory()
            ws.stack_checked = False
        ba = self._branch(va, ws.X, ws.MIX[0], None, None, None, 0, True, True, None, None)
        bb = self._branch(vb, ws.X, ws.MIX[1], None, None, None, 0, True, True, None, None)
        for br, v in ((ba, va), (bb, vb)):
            br.planes_format, br.flags, br.in_amax = 1, _capi.BRANCH_SELF_RANGE, None
            br.dft_frags = _p(self._dft_frags(v.L, v.K, True))
        rows = []
        for l in range(L):
            si = self._fw_sets.index(self.fw_names[l]) if full else 0
            l0, l1, b0, b1 = self._ff_weights(l)
            rows.append(_capi.InferStackLayer(_p(self._planes_for(si, a, 0, True)), _p(self._planes_for(si, b, 0, True)), _p(l0.fx[0]),
                                              _p(b0), _p(l0.fx[1]), _p(b1)))
        arr = (_capi.InferStackLayer * L)(*rows)
        d = _capi.InferStackDesc(ba, bb, ctypes.cast(arr, ctypes.c_void_p), L, C, H, 0, _p(ws.Blast), _p(ws.stack_sync))
        t = self.timer
        if t is not None and t.want("infer_stack"):      # (bench.py: HIP events around the one launch)
            t.start("infer_stack", self._issue_stream)
            rc = lib.ffno_infer_stack(ctypes.byref(d), st)
            t.stop("infer_stack", self._issue_stream)
        else:
            rc = lib.ffno_infer_stack(ctypes.byref(d), st)
        if rc != 0:      # (e.g. the persistent launch was refused: the kernel does not fit once per CU on this device)
            self.use_infer_stack = False
            return False
        err_word = ws.stack_sync[-1:]
        ws.stack_calls = getattr(ws, "stack_calls", 0) + 1
        if not ws.stack_checked:
            # first persistent launch on this workspace: look at its error word before trusting the path
            ws.stack_checked = True
            if int(err_word.cpu()[0]) != 0:
                self.use_infer_stack = False
                import warnings
                warnings.warn("ffno_infer_stack reported a placement / barrier error on this device: using the per-layer launches",
                              RuntimeWarning)
                raise _StackRetry()
        elif self.device.type == "cuda" and getattr(ws, "stack_pending", None) is None and ws.stack_calls % 16 == 0:
            # (every 16th launch: the 4-byte copy is a launch of its own; an error after the first checked launch would be a
            #  hardware / scheduling anomaly -- it is reported a few passes late, but it is reported)
            ws.stack_host.copy_(err_word, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            ws.stack_pending = (ws.stack_host, ev)
        return True

    def _spectral(self, name, ws, v: _View, src, dst, resid, save, planes, fwd: bool, accumulate: int, fused: bool, st,
                  x3: bool = False, rin=None, rout=None):
        """One spectral branch  dst (+)= [resid +] iDFT(mix(DFT(src)))  along view v (forward or adjoint);
        rin / rout = range words of src / dst."""
        if not (fused and x3):
            folded = self._spectral_plain(name, ws, v, src, dst, resid, save, planes, fwd, accumulate, fused, st, rout)
            if not folded:
                self._fold(dst, rout, st)       # the stage / DCT / 2-D kernels do not record their output maximum themselves
            return
        br = self._branch(v, src, dst, resid, save, planes, accumulate, True, fwd, rin, rout)
        ck_f, ck_i, conj = (0, 1, 0) if fwd else (1, 0, 1)
        self._k(name, _lib.get_lib().ffno_spectral_x3, ctypes.byref(br), self.C, ck_f, ck_i, conj, st)

    def _spectral_plain(self, name, ws, v: _View, src, dst, resid, save, planes, fwd: bool, accumulate: int, fused: bool, st,
                        rout=None) -> bool:
        """-> True when the kernel that ran folded max |dst| into ``rout`` itself."""
        lib = _lib.get_lib()
        C = self.C
        tw = self._twiddle(v.L)
        ck_f, ck_i, conj = (0, 1, 0) if fwd else (1, 0, 1)
        if self.spectral == "plus":
            # rfft2 = last-axis DFT (K bins) then complex DFT along the first axis (2K retained rows); corner mix with the
            # (ky, kx') pairs as modes and the samples as rows; zero-padded inverse in the opposite order.  The adjoint
            # is the same chain with the c_k / conjugate flags swapped (cdft_rows(inverse) is the adjoint of the forward).
            z = save if save is not None else ws.SD
            self._k("dft_fwd", lib.ffno_dft_fwd, _p(src), _p(ws.SYa), _p(tw), v.Bv, v.Mv, v.Nv, C, v.K, 0, ck_f, st)
            twm = self._twiddle(v.Mv)
            self._k("cdft_rows", lib.ffno_cdft_rows_mfma, _p(ws.SYa), _p(z), _p(ws.CW), _p(twm), v.Bv, v.Mv, C, v.K, v.K, 0, st)
            self._k("mode_mix", lib.ffno_mode_mix, _p(z), _p(planes), _p(ws.SY), v.Bv, C, v.K2, conj, st)
            self._k("cdft_rows", lib.ffno_cdft_rows_mfma, _p(ws.SY), _p(ws.SYb), _p(ws.CW), _p(twm), v.Bv, v.Mv, C, v.K, v.K, 1,
                    st)
            self._k("dft_inv", lib.ffno_dft_inv, _p(ws.SYb), _p(dst), resid, _p(tw), v.Bv, v.Mv, v.Nv, C, v.K, 0, ck_i,
                    accumulate, st)
            return
        if self.spectral == "dct":       # DCT branch: the same transform pair serves the forward and the adjoint (orthonormal)
            spec = save if save is not None else ws.SD
            self._k("dct_branch" + ("" if fwd else "(adj)"), lib.ffno_dct_branch, _p(src), _p(dst), resid, _p(spec), _p(ws.SY),
                    _p(planes), _p(self._twiddle(2 * v.L)), v.Bv, v.Mv, v.Nv, C, v.K, v.a01, conj, accumulate, st)
            return
        if fused:
            self._k(name, lib.ffno_spectral_fused, _p(src), _p(dst), resid, _p(save), _p(planes), _p(tw),
                    v.Bv, v.Mv, v.Nv, C, v.K, v.a01, ck_f, ck_i, conj, accumulate, rout, st)
            return True
        SD, SY = ws.SD, ws.SY
        spec = save if save is not None else SD
        self._k("dft_fwd", lib.ffno_dft_fwd, _p(src), _p(spec), _p(tw), v.Bv, v.Mv, v.Nv, C, v.K, v.a01, ck_f, st)
        y = spec
        if planes is not None:
            self._k("mode_mix", lib.ffno_mode_mix, _p(spec), _p(planes), _p(SY), v.R, C, v.K, conj, st)
            y = SY
        self._k("dft_inv", lib.ffno_dft_inv, _p(y), _p(dst), resid, _p(tw), v.Bv, v.Mv, v.Nv, C, v.K, v.a01, ck_i,
                accumulate, st)

    # ------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, save_for_backward: bool, training: Optional[bool] = None, own_output: bool = True) -> torch.Tensor:
        try:
            return self._forward(x, save_for_backward, training, own_output)
        except _StackRetry:      # (the persistent inference launch is off now: the same pass on the per-layer launches)
            return self._forward(x, save_for_backward, training, own_output)

    def _forward(self, x: torch.Tensor, save_for_backward: bool, training: Optional[bool] = None, own_output: bool = True) -> torch.Tensor:
        """x [B, *spatial, input_dim] fp32 on the device -> [B, *spatial, output_dim] (a fresh tensor).
        ``training`` (default: save_for_backward) switches the dropout masks on (nn.Module.training of the reference).
        ``own_output=False`` returns a view of the workspace's output buffer instead of a copy -- valid until the next forward of
        this shape (FFNOTrainer.train_step consumes it at once: one copy launch less per step)."""
        if not self.params:
            raise RuntimeError("bind() parameters first")
        _lib.require_device_tensor(x, "x")
        if x.dim() != self.nd + 2 or x.shape[-1] != self.Cin:
            raise ValueError(f"x must be [B, {'M, N' if self.nd == 2 else 'X, Y, Z'}, {self.Cin}], got {tuple(x.shape)}")
        x = x.contiguous()
        B, S = x.shape[0], tuple(x.shape[1:-1])
        lib = _lib.get_lib()
        C, H, L = self.C, self.H, self.L
        ws = self._workspace(B, S, save_for_backward)
        if self.mode != "no-fourier":
            for v in ws.views:
                if v.K > v.L // 2 + 1:
                    raise ValueError(f"modes={v.K} exceeds size//2+1 for an axis of length {v.L} "
                                     f"(the reference raises an einsum size error)")
        st = _lib.current_stream(self.device)
        P = ws.P
        self._training = bool(save_for_backward if training is None else training)
        if self._training and (self.dropout > 0.0 or self.in_dropout > 0.0):
            self._drop_calls += 1
        fused = self._can_fuse(ws.views)
        x3 = self._use_x3(ws.views, fused)
        # format of the packed x3 weight sets per axis: fp16x2 where the FUSED x3 kernel mixes with them (the stage kernels of
        # the 17..32-mode axes read bf16x3 packs)
        # ... and FFNO_PLANES_FP16X2_M16 (the 16-row mix) for the many-mode axes (width 64, 17..64 modes: 4-line tiles with 8 live mix
        # rows) when their DFT-fragment tables are in use and ALL axes are such and share a tile height (a paired launch carries one
        # format and one tile height)
        many = [v.K > 16 for v in ws.views]
        m16 = bool(C == 64 and all(many) and len({v.K <= 32 for v in ws.views}) == 1)
        self._x3_fmt = [(2 if (m16 and many[w]) else 1) if (self._x3_h2() and fused[w] and x3[w]) else 0 for w in range(len(ws.views))]
        for w, v in enumerate(ws.views):
            v.x3fmt = self._x3_fmt[w]
        self._prepare_weights(st)
        full = self.mode == "full"
        singles, pair = self._schedule(fused, ws.views) if self._conc() else (list(range(len(ws.views))), None)
        if pair is not None and fused[pair[0]] and not (x3[pair[0]] and x3[pair[1]]):
            # one fused launch for two axes needs ONE kernel family to take both shapes: either both on the split (x3) kernels
            # or both on the fp32-MFMA fused kernel; an axis that is "fused" through the x3 kernels only next to one that the
            # x3 kernels refuse (x3_min_lines, a missing pack) would be handed to a kernel that does not support it
            if not all(lib.ffno_spectral_fused_supported(C, ws.views[w].K, ws.views[w].L) for w in pair):
                singles, pair = list(range(len(ws.views))), None
        conc = pair is not None
        x3pair = bool(conc and x3[pair[0]] and x3[pair[1]])
        if conc and not fused[pair[0]]:      # both axes staged: the split-bf16 stage kernels when the library takes the shape
            x3pair = bool(self.use_x3 and self.spectral == "factorized" and all(
                lib.ffno_spectral_x3_staged_supported(C, ws.views[w].K, ws.views[w].L)
                and (not full or self.xplanes[0][w] is not None) for w in pair)
                and 4 * C * max(v.Bv * v.Mv * v.Nv for v in ws.views) < 2 ** 32)
        layer_calls = bool(self.use_layer_calls and self.timer is None and conc and fused[pair[0]] and not self.use_fork
                           and not self.layer_norm)
        infer = bool(not save_for_backward and self._infer_ok(ws, B, conc, pair, singles, x3pair, fused))
        if infer and getattr(ws, "MIX", None) is None:
            ws.MIX = [torch.empty(int(lib.ffno_infer_mix_bytes(C, ws.views[w].K, ws.views[w].R)) // 4, dtype=torch.int32, device=self.device)
                      for w in pair]
        self.infer_last = infer      # (tests / bench.py: which layer kernels the last forward ran)
        # self-ranged inference layers (axis lengths <= 64: ffno.h FFNO_BRANCH_SELF_RANGE): the first kernel scales every line from its
        # own maximum, so no launch of this forward has to fold a range word for it (lift, second kernel: no atomics, no fold barrier)
        infer_sr = bool(infer and self.infer_self_range and all(ws.views[w].L <= 64 for w in pair))
        self.infer_self_ranged_last = infer_sr
        # both branch outputs of every layer are kept and the feed-forward does not write their sum (the deferred weight-gradient
        # launch forms it): decided here, the backward pass follows (self._saved_lazy)
        lazy = bool(save_for_backward and conc and getattr(ws, "lazy_sums", ""))
        bf16 = self._bf16()
        if bf16 and not (all(fused) and all(x3) and self._h2() and self._x3_h2() and self._ffx() and self.spectral == "factorized"
                         and (C, H) in ((64, 256), (32, 128)) and full and not self.use_fork and not self.layer_norm
                         and self.dropout == 0.0 and self.in_dropout == 0.0):
            raise NotImplementedError(
                "storage='bf16' covers the fused split-kernel path of the factorized operators (2-D grids / meshes and the 3-D mesh: "
                "width 64 with up to 64 modes or width 32 with up to 16, factor 4, mode='full', fp16x2 splits, 2-layer feed-forward "
                "without fork heads / LayerNorm / dropout); this configuration runs with storage='fp32'")
        lin_in = self.linears["in_proj."]
        pm = ctypes.byref(ws.padmap) if ws.padmap is not None else None
        if pm is not None:
            ws.X.zero_()     # F.pad(..., 0) of the lifted features (mesh_3d.py:165)
        rw = self._rw
        if self._ranged() and not infer_sr:
            # the forward's words: layer inputs (x) and branch outputs (s) -- and, in the same launch, the words of the backward
            # pass that will follow this forward
            ws.RW.zero_()
            ws.rw_bwd_clean = True
        in_drop = self._training and self.in_dropout > 0.0
        self._k("lift_fwd", lib.ffno_lift_fwd_bf16 if bf16 else lib.ffno_lift_fwd, _p(x), _p(lin_in.weff),
                _p(self.params["in_proj.bias"]), _p(ws.X), ws.P_in,
                self.Cin, C, pm, None if (in_drop or infer_sr) else rw(ws, "x", 0), st)
        if in_drop:      # x = self.drop(x) after in_proj (grid_2d.py:158): a regenerated mask over the lifted features
            self._in_drop_seed = _site_seed(self.drop_seed, self._drop_calls, 0xFFFF, 0, 2)
            self._k("in_dropout", lib.ffno_dropout, _p(ws.X), ws.X.numel(), self.in_dropout, self._in_drop_seed, st)
            self._fold(ws.X, rw(ws, "x", 0), st)
        # the whole stack of self-ranged inference layers as ONE persistent launch (ffno_infer_stack: 64 x 64 images, batch x 8 = the CU
        # count; the 8 workgroups of an image run both kernels of every layer as phases): no launch boundary, no chip-wide lock step
        self.infer_stack_last = False
        if (infer_sr and self.use_infer_stack and (self.timer is None or getattr(self.timer, "allow_stack", False))
                and self._run_infer_stack(ws, pair, full, st)):
            self.infer_stack_last = True
            n_loop = 0
        else:
            n_loop = L
        for l in range(n_loop):
            sv = l if save_for_backward else 0
            last = l == L - 1
            s_l = ws.S[sv]
            rx, rs_, rxn = rw(ws, "x", l), rw(ws, "s", l), rw(ws, "x", l + 1)
            if self.mode == "no-fourier":
                s_l.copy_(ws.X)
                self._fold(s_l, rs_, st)
            else:
                si = self._fw_sets.index(self.fw_names[l]) if full else 0
                nwrit = 0
                for w in singles:
                    v = ws.views[w]
                    keep = ws.SXall[w][sv] if full else None     # stage-A spectrum (kept per layer when training)
                    if fused[w] and not save_for_backward:
                        keep = None
                    self._spectral("spectral_fused", ws, v, ws.X, s_l, None, keep,
                                   self._planes_for(si, w, 0, x3[w]), True, int(nwrit > 0), fused[w], st, x3=x3[w],
                                   rin=rx, rout=rs_)
                    nwrit += 1
                if infer:
                    a, b = pair
                    l0, l1, b0, b1 = self._ff_weights(l)
                    ba = self._branch(ws.views[a], ws.X, ws.MIX[0], None, None, self._planes_for(si, a, 0, True), 0, True, True, rx, None)
                    bb = self._branch(ws.views[b], ws.X, ws.MIX[1], None, None, self._planes_for(si, b, 0, True), 0, True, True, rx, None)
                    if infer_sr:
                        ba.flags = bb.flags = _capi.BRANCH_SELF_RANGE
                        ba.in_amax = bb.in_amax = None
                        rxn = None
                    out_l = ws.Blast if last else ws.X
                    if self.timer is None:
                        d = _capi.LayerInferDesc(ba, bb, X3_INTERLEAVE, 0, _p(l0.fx[0]), _p(b0), _p(l0.fx[1]), _p(b1),
                                                 None if last else _p(ws.X), _p(out_l), C, H, rxn)
                        self._k("layer_infer", lib.ffno_layer_infer, ctypes.byref(d), st)
                    else:       # (per-kernel timing: the two launches as two calls)
                        self._k("spectral_mix", lib.ffno_spectral_x3_mix_pair, ctypes.byref(ba), ctypes.byref(bb), C, X3_INTERLEAVE, st)
                        self._k("infer_ff", lib.ffno_infer_ff, ctypes.byref(ba), ctypes.byref(bb), _p(l0.fx[0]), _p(b0), _p(l0.fx[1]),
                                _p(b1), None if last else _p(ws.X), _p(out_l), C, H, rxn, st)
                    continue
                if conc:
                    a, b = pair
                    keep = [ws.SXall[w][sv] if (full and save_for_backward) else None for w in pair]
                    t_l = ws.TS[sv] if lazy else ws.T
                    if layer_calls:
                        l0, l1, b0, b1 = self._ff_weights(l)
                        d = _capi.LayerFwdDesc(
                            self._branch(ws.views[a], ws.X, s_l, None, keep[0], self._planes_for(si, a, 0, x3pair), int(nwrit > 0),
                                         x3pair, True, rx, rs_),
                            self._branch(ws.views[b], ws.X, t_l, None, keep[1], self._planes_for(si, b, 0, x3pair), 0, x3pair, True,
                                         rx, rs_),
                            int(x3pair), X3_INTERLEAVE, _p(l0.fx[0]), _p(b0), _p(l0.fx[1]), _p(b1),
                            _p(s_l) if (save_for_backward and not lazy) else None, None if last else _p(ws.X), _p(ws.Blast if last else ws.X),
                            _p(ws.MASK[sv]) if save_for_backward else None, P, C, H, int(self._h2()),
                            int(self.ff_schedule), int(self.ff_max_workgroups), 0, 0, rxn)
                        self._k("layer_fwd", lib.ffno_layer_fwd, ctypes.byref(d), st)
                        continue
                    self._pair("spectral_fused", ws, ws.views[a], ws.views[b], ws.X, s_l, t_l, None, keep[0], keep[1],
                               self._planes_for(si, a, 0, x3pair), self._planes_for(si, b, 0, x3pair), True, st,
                               acc0=int(nwrit > 0), fused=fused[a], x3=x3pair, rin=rx, rout=rs_)
            l0, l1, b0, b1 = self._ff_weights(l)
            # FeedForward(layer_norm=True): the feed-forward writes its raw output, the LayerNorm kernel adds the residual
            ff_out = ws.TL[sv] if self.layer_norm else (ws.Blast if last else ws.X)
            ff_res = None if (self.layer_norm or last) else ws.X
            ff_rout = None if self.layer_norm else rxn     # (LayerNorm: the next layer's input is the normalised tensor)
            if conc:
                self._ffs_fwd2(s_l, ws.TS[sv] if lazy else ws.T, s_l if (save_for_backward and not lazy) else None, ff_res, l0, b0, b1, ff_out,
                               ws.MASK[sv] if save_for_backward else None, P, st, rs_, ff_rout)
            elif self.use_fork and last:          # with fork heads the last layer's backcast only feeds the dead x_L
                pass
            elif self.general_ff:
                self._ffg_fwd(ws, self.ff_prefix[l], "backcast", l, sv, s_l, ff_res, ff_out, P, st, ff_rout)
            else:
                self._ff_fwd(s_l, ff_res, l0, l1, b0, b1, ff_out,
                             ws.Hbuf[sv] if save_for_backward else None, ws.MASK[sv] if save_for_backward else None, P, st,
                             rs_, ff_rout)
            if self.layer_norm and not (self.use_fork and last):
                ln = self.ff_prefix[l] + f"layers.{self.n_ff - 1}.3."
                self._k("layernorm_fwd", lib.ffno_layernorm_fwd, _p(ws.TL[sv]), _p(self.params[ln + "weight"]),
                        _p(self.params[ln + "bias"]), None if last else _p(ws.X), _p(ws.Blast if last else ws.X), _p(ws.LNS[sv]),
                        P, C, 1e-5, st)
                if not last:
                    self._fold(ws.X, rxn, st)
            if self.use_fork:
                c0, c1, cb0, cb1 = self._fc_weights(l)
                f_raw = ws.TLF[sv] if self.layer_norm else ws.F[sv]
                if self.general_ff:
                    self._ffg_fwd(ws, self.fc_prefix[l], "forecast", l, sv, s_l, None, f_raw, P, st)
                else:
                    self._ff_fwd(s_l, None, c0, c1, cb0, cb1, f_raw, ws.HF[sv] if save_for_backward else None,
                                 ws.MASKF[sv] if save_for_backward else None, P, st, rs_, None)
                if self.layer_norm:
                    ln = self.fc_prefix[l] + f"layers.{self.n_ff - 1}.3."
                    self._k("layernorm_fwd", lib.ffno_layernorm_fwd, _p(f_raw), _p(self.params[ln + "weight"]),
                            _p(self.params[ln + "bias"]), None, _p(ws.F[sv]), _p(ws.LNSF[sv]), P, C, 1e-5, st)
                self._k("head_fwd", lib.ffno_head_fwd, _p(ws.F[sv]), _p(self.fold), _p(ws.YL[l]), ws.P_in, C, self.O, 0, pm, st)
        if self.use_fork:
            torch.sum(ws.YL, dim=0, out=ws.Y)     # forecast = sum of the per-layer head outputs
            self.forecast_list = [ws.YL[l].view(B, *S, self.O).clone() for l in range(L)]
        else:
            self._k("head_fwd", lib.ffno_head_fwd_bf16 if bf16 else lib.ffno_head_fwd, _p(ws.Blast), _p(self.fold), _p(ws.Y),
                    ws.P_in, C, self.O, 0, pm, st)
        self._saved = (x, B, S, fused, conc) if save_for_backward else None
        # the workspace this pass filled (saved tensors, sign bits, range words): the backward pass and the diagnostic readers use
        # THIS object -- an attribute switched between forward() and backward() changes the workspace key, and a freshly
        # allocated workspace would hand them uninitialised memory
        self._saved_ws = ws if save_for_backward else None
        self._saved_ws_key = self._ws_key if save_for_backward else None
        self._saved_x3 = (x3, x3pair)
        self._saved_sched = (singles, pair)
        self._saved_lazy = lazy
        self.paired_last = conc      # (bench.py: which algorithmic-work table applies)
        y = ws.Y.view(B, *S, self.O)
        return y.clone() if own_output else y

    # ------------------------------------------------------------------------------------------------
    def backward(self, gy: torch.Tensor, need_dx: bool = False) -> torch.Tensor:
        """gy = dL/dout [B, *spatial, output_dim].  Fills and returns the flat gradient buffer ``gflat``
        (layout: ``param_names`` order; use ``grad_view(name)``).  ``need_dx``: also compute the gradient with respect to
        the input tensor (``self.dx``, [B, *spatial, input_dim]) -- what autograd of the reference module would hand upstream."""
        if self._saved is None:
            raise RuntimeError("backward() needs a preceding forward(save_for_backward=True)")
        x, B, S, fused, conc = self._saved
        x3, x3pair = self._saved_x3
        singles, pair = self._saved_sched
        _lib.require_device_tensor(gy, "gy")
        gy = gy.contiguous()
        lib = _lib.get_lib()
        C, H, L = self.C, self.H, self.L
        ws = self._forward_workspace(B, S)
        lazy_s = bool(getattr(self, "_saved_lazy", False))            # forward input sums left to the weight-gradient launch
        st = _lib.current_stream(self.device)
        rw = self._rw
        if self._ranged():
            # the backward's range words (g: running gradient, d: feed-forward data gradients, t / f: LayerNorm / fork heads):
            # every fp16x2 kernel of the pass scales its operands from the maximum its producer recorded -- per layer, on the
            # device, so no growth or decay of the gradient through the layers can leave the half format's range
            if not getattr(ws, "rw_bwd_clean", False):      # (a second backward pass over the same forward)
                ws.RW[2 * (L + 1):].zero_()
            ws.rw_bwd_clean = False
        P = ws.P
        full = self.mode == "full"
        pm = ctypes.byref(ws.padmap) if ws.padmap is not None else None
        o0, o1 = self.linears["out.0."], self.linears["out.1."]
        gv = self.grad_view
        have_g1 = False      # G1 holds the second adjoint branch's part of the running gradient (to be added to g_in)
        cur = 0
        if pm is not None:
            (ws.GF if self.use_fork else ws.G[cur]).zero_()    # adjoint of the crop (mesh_3d.py:173)
        if self.use_fork:
            # every layer's f_l goes through the same head: d/df_l = gy . weff for all l (computed once), and the
            # head-parameter reduction sums over layers
            for l in range(L):
                self._k("head_bwd", lib.ffno_head_bwd, _p(ws.F[l]), _p(gy), _p(self.fold), _p(ws.GF) if l == 0 else None,
                        _p(ws.headpart), _p(ws.red if l == 0 else ws.redl), ws.P_in, C, self.O, ws.nsplit_head, pm,
                        rw(ws, "f", 0) if l == 0 else None, st)
                if l > 0:
                    self._k("axpy", lib.ffno_axpy, _p(ws.red), _p(ws.redl), 1.0, self.O * (C + 1), st)
        else:
            self._k("head_bwd", lib.ffno_head_bwd_bf16 if self._bf16() else lib.ffno_head_bwd, _p(ws.Blast), _p(gy), _p(self.fold),
                    _p(ws.G[cur]), _p(ws.headpart), _p(ws.red), ws.P_in, C, self.O, ws.nsplit_head, pm, rw(ws, "g", L - 1), st)
        self._k("head_param_grads", lib.ffno_head_param_grads, _p(ws.red), _p(o0.weff), _p(self.params["out.0.bias"]),
                _p(o1.weff), _p(o0.gweff), _p(gv("out.0.bias")), _p(o1.gweff), _p(gv("out.1.bias")), C, HEAD_DIM, self.O, 0, st)
        if not self._ffx() and self._n_tr:
            self._k("transpose_batched", lib.ffno_transpose_batched, _p(self._tr_dev), self._n_tr, max(C, H), max(C, H), st)
        ff_seen = set()
        ws.red_jobs = []
        nG = len(ws.G)
        # deferred weight-gradient launch: (s, summed gradient, packs, words, slices) of every layer, one launch after the loop
        # (paired launches or not: the airfoil mesh -- a 32-mode axis beside a 16-mode one, two launches per layer -- ran 24 per-layer
        #  weight-gradient launches of 56 us until round 5)
        ws.wg_jobs = [] if getattr(ws, "defer_wgrad", False) else None
        if lazy_s and ws.wg_jobs is None:
            raise RuntimeError("the forward pass left the feed-forward input sums to a deferred weight-gradient launch that this "
                               "backward pass cannot run")
        layer_calls = bool(self.use_layer_calls and self.timer is None and conc and pair is not None and fused[pair[0]]
                           and not singles and not self.use_fork and getattr(ws, "defer_reduce", False)
                           and self.mode != "no-fourier" and not self.layer_norm)
        for l in reversed(range(L)):
            last = l == L - 1
            l0, l1, _, _ = self._ff_weights(l)
            fp = self.ff_prefix[l]
            g_in, g_out, dh = ws.G[cur], ws.G[(cur + 1) % nG], ws.DH[l & 1]
            g1_in = g1_out = getattr(ws, "G1", None)
            # words: gradient entering this layer, its feed-forward input, the data gradient, the gradient it hands on
            rg, rs_, rd, rgo = rw(ws, "g", l), rw(ws, "s", l), rw(ws, "d", l), (rw(ws, "g", l - 1) if l > 0 else rw(ws, "g", L))
            if self.use_fork:
                c0, c1, _, _ = self._fc_weights(l)
                fc = self.fc_prefix[l]
                ds_f = ws.DS if last else ws.DSF     # last layer: the forecast path is the only contribution to ds
                rf = rw(ws, "f", 0)
                g_fc = ws.GF
                if self.layer_norm:      # through the forecast block's LayerNorm first
                    ln = fc + f"layers.{self.n_ff - 1}.3."
                    self._k("layernorm_bwd", lib.ffno_layernorm_bwd, _p(ws.TLF[l]), _p(ws.LNSF[l]), _p(self.params[ln + "weight"]),
                            _p(ws.GF), None, None, _p(ws.DTF), _p(ws.lnpart), _p(gv(ln + "weight")), _p(gv(ln + "bias")), P, C,
                            int(fc in ff_seen), st)
                    g_fc, rf = ws.DTF, rw(ws, "f", 1)
                    self._fold(ws.DTF, rf, st)
                if self.general_ff:
                    self._ffg_bwd(ws, fc, "forecast", l, ws.S[l], g_fc, ds_f, int(fc in ff_seen), P, st, rd)
                else:
                    self._ff_bwd_data(g_fc, ws.MASKF[l], c0, c1, dh, ds_f, P, st, rf, rd)
                    self._ff_bwd_weights(ws, ws.S[l], g_fc, ws.HF[l], dh, c0, c1, self.params[fc + "layers.0.0.bias"],
                                         gv(fc + "layers.0.0.bias"), gv(fc + "layers.1.0.bias"), int(fc in ff_seen), P, st, rs_, rf)
                ff_seen.add(fc)
            if self.use_fork and last:
                # x_L is never used with fork heads: the last backcast_ff gets a zero gradient
                if fp not in ff_seen:
                    for k in range(self.n_ff):
                        gv(fp + f"layers.{k}.0.bias").zero_()
                        self.linears[fp + f"layers.{k}.0."].gweff.zero_()
                    if self.layer_norm:
                        gv(fp + f"layers.{self.n_ff - 1}.3.weight").zero_()
                        gv(fp + f"layers.{self.n_ff - 1}.3.bias").zero_()
                    ff_seen.add(fp)
                if self.mode == "no-fourier":
                    g_out.copy_(ws.DS)
                    self._fold(g_out, rgo, st)
                else:
                    si = self._fw_sets.index(self.fw_names[l]) if full else 0
                    for w, v in enumerate(ws.views):
                        keep = ws.SDall[w][l] if full else None
                        self._spectral("spectral_fused(adj)", ws, v, ws.DS, g_out, None, keep,
                                       self._planes_for(si, w, 1, x3[w]), False, int(w > 0), fused[w], st, x3=x3[w],
                                       rin=rd, rout=rgo)
                cur = (cur + 1) % nG
                continue
            if layer_calls:
                # the whole layer backward in one call: FF data gradient (g_in (+)= G1), weight-gradient slices, adjoint pair
                si = self._fw_sets.index(self.fw_names[l]) if full else 0
                a, b = pair
                part = ws.ffparts[len(ws.red_jobs)]
                if ws.wg_jobs is not None:
                    ws.wg_jobs.append((ws.S[l].data_ptr(), g_in.data_ptr(), l0.fx[0].data_ptr(),
                                       self.params[fp + "layers.0.0.bias"].data_ptr(), l0.fx[2].data_ptr(), part.data_ptr(),
                                       rs_.value, rg.value, ws.TS[l].data_ptr() if lazy_s else 0, 0))
                d = _capi.LayerBwdDesc(
                    self._branch(ws.views[a], ws.DS, g_out, None if last else _p(g_in), ws.SDall[a][l] if full else None,
                                 self._planes_for(si, a, 1, x3pair), 0, x3pair, False, rd, rgo),
                    self._branch(ws.views[b], ws.DS, g1_out, None, ws.SDall[b][l] if full else None,
                                 self._planes_for(si, b, 1, x3pair), 0, x3pair, False, rd, rgo),
                    int(x3pair), X3_INTERLEAVE, _p(g_in), _p(g1_in) if have_g1 else None, _p(g_in),
                    _p(ws.MASK[l]),
                    _p(l0.fx[2]), _p(l0.fx[3]), _p(ws.DS), _p(ws.S[l]), _p(l0.fx[0]), _p(self.params[fp + "layers.0.0.bias"]),
                    None if ws.wg_jobs is not None else _p(part), ws.nsplit_ff, P, C, H, int(self._h2()),
                    int(self.ff_schedule), int(self.ff_max_workgroups), 0, rg, rs_, rd)
                self._k("layer_bwd", lib.ffno_layer_bwd, ctypes.byref(d), st)
                ws.red_jobs.append((part.data_ptr(), l0.gweff.data_ptr(), l1.gweff.data_ptr(), gv(fp + "layers.0.0.bias").data_ptr(),
                                    gv(fp + "layers.1.0.bias").data_ptr()))
                ff_seen.add(fp)
                have_g1 = True
                cur = (cur + 1) % nG
                continue
            g_ff = g_in       # gradient w.r.t. the feed-forward output
            if self.layer_norm:
                # through the LayerNorm first (it also folds in the second gradient buffer of a paired adjoint launch)
                ln = fp + f"layers.{self.n_ff - 1}.3."
                two = bool(conc and have_g1)
                self._k("layernorm_bwd", lib.ffno_layernorm_bwd, _p(ws.TL[l]), _p(ws.LNS[l]), _p(self.params[ln + "weight"]),
                        _p(g_in), _p(ws.G1) if two else None, _p(g_in) if two else None, _p(ws.DT), _p(ws.lnpart),
                        _p(gv(ln + "weight")), _p(gv(ln + "bias")), P, C, int(fp in ff_seen), st)
                g_ff = ws.DT
                rg = rw(ws, "t", l)
                self._fold(ws.DT, rg, st)
            if conc:
                # g_in (+)= G1 while it is staged; the sum is stored back for the weight gradient and the residual path
                two = bool(have_g1 and not self.layer_norm)
                self._ffs_bwd2(g_ff, g1_in if two else None, g_ff if two else None, ws.MASK[l], l0, ws.DS, P, st, rg, rd)
                self._wg_second = ws.TS[l] if lazy_s else None
            elif self.general_ff:
                self._ffg_bwd(ws, fp, "backcast", l, ws.S[l], g_ff, ws.DS, int(fp in ff_seen), P, st, rd)
            else:
                self._ff_bwd_data(g_ff, ws.MASK[l], l0, l1, dh, ws.DS, P, st, rg, rd)
                self._wg_second = None
            if not self.general_ff:
                self._ff_bwd_weights(ws, ws.S[l], g_ff, ws.Hbuf[l], dh, l0, l1, self.params[fp + "layers.0.0.bias"],
                                     gv(fp + "layers.0.0.bias"), gv(fp + "layers.1.0.bias"), int(fp in ff_seen), P, st, rs_, rg)
            ff_seen.add(fp)
            if self.use_fork:
                self._k("axpy", lib.ffno_axpy, _p(ws.DS), _p(ws.DSF), 1.0, P * C, st)     # ds = ds(backcast) + ds(forecast)
                self._fold(ws.DS, rd, st)     # (the word already holds both addends' maxima; the sum may exceed either)
            if self.mode == "no-fourier":
                if last:
                    g_out.copy_(ws.DS)
                else:
                    torch.add(g_in, ws.DS, out=g_out)
                self._fold(g_out, rgo, st)
                cur = (cur + 1) % nG
                continue
            si = self._fw_sets.index(self.fw_names[l]) if full else 0
            resid = None if last else _p(g_in)      # G_{l-1} = G_l (residual path) + adjoint terms; last layer: none
            nwrit = 0
            for w in singles:
                v = ws.views[w]
                keep = ws.SDall[w][l] if full else None   # dY of every layer is kept for the dW launch
                self._spectral("spectral_fused(adj)", ws, v, ws.DS, g_out, resid if nwrit == 0 else None, keep,
                               self._planes_for(si, w, 1, x3[w]), False, int(nwrit > 0), fused[w], st, x3=x3[w], rin=rd, rout=rgo)
                nwrit += 1
            if conc:
                a, b = pair
                self._pair("spectral_fused(adj)", ws, ws.views[a], ws.views[b], ws.DS, g_out, g1_out,
                           resid if nwrit == 0 else None, ws.SDall[a][l] if full else None, ws.SDall[b][l] if full else None,
                           self._planes_for(si, a, 1, x3pair), self._planes_for(si, b, 1, x3pair), False, st,
                           acc0=int(nwrit > 0), fused=fused[a], x3=x3pair, rin=rd, rout=rgo,
                           resid1=None)
            have_g1 = conc
            cur = (cur + 1) % nG
        g_second = None           # second addend of the gradient that enters the lift (fp32 storage, no input gradient wanted:
        if conc and have_g1:      # ffno_lift_bwd2 adds it while it stages its rows; else the sum is formed first)
            g1_fin = ws.G1
            if self._bf16():
                ws.G[cur].add_(g1_fin)      # (a torch kernel on the same stream; rounds the sum to bf16 like every stored tensor)
            elif need_dx or (self._training and self.in_dropout > 0.0):
                self._k("axpy", lib.ffno_axpy, _p(ws.G[cur]), _p(g1_fin), 1.0, P * C, st)
            else:
                g_second = g1_fin
        nsl = ws.nsplit_ff
        if ws.wg_jobs:
            nsl = ws.nsplit_ffm
            sig = tuple(ws.wg_jobs)
            if sig != ws.wg_sig:      # (a HOST table: the library copies it into the kernel arguments at enqueue)
                ws.wg_table = (_capi.FfWgDesc * len(sig))(*[_capi.FfWgDesc(*j) for j in sig])
                ws.wg_sig = sig
            self._k("ff_bwd_weights_partial", lib.ffno_ffh_bwd_weights_partial_multi, ws.wg_table, len(sig), P, C, H, nsl,
                    self._st(), int(lazy_s), st)
        if getattr(ws, "defer_reduce", False) and ws.red_jobs:
            sig = tuple(ws.red_jobs)
            if sig != ws.red_sig:       # pointers only change when parameters are re-bound or the workspace is rebuilt
                arr = (_capi.FxRedDesc * len(sig))(*[_capi.FxRedDesc(*j) for j in sig])
                ws.red_table = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(self.device)
                ws.red_sig = sig
            self._k("ff_bwd_weights_reduce", lib.ffno_ffx_bwd_weights_reduce_batched, _p(ws.red_table), len(sig), C, H,
                    nsl, st)
        g_fin = ws.G[cur]
        lin_in = self.linears["in_proj."]
        if self._training and self.in_dropout > 0.0:      # backward of x = self.drop(in_proj(x)): the same mask, regenerated
            self._k("in_dropout", lib.ffno_dropout, _p(g_fin), g_fin.numel(), self.in_dropout, self._in_drop_seed, st)
        if g_second is not None:
            self._k("lift_bwd", lib.ffno_lift_bwd2, _p(x), _p(g_fin), _p(g_second), _p(ws.liftpart), _p(lin_in.gweff),
                    _p(gv("in_proj.bias")), ws.P_in, self.Cin, C, ws.nsplit_lift, 0, pm, st)
        else:
            self._k("lift_bwd", lib.ffno_lift_bwd_bf16 if self._bf16() else lib.ffno_lift_bwd, _p(x), _p(g_fin), _p(ws.liftpart),
                    _p(lin_in.gweff), _p(gv("in_proj.bias")), ws.P_in, self.Cin, C, ws.nsplit_lift, 0, pm, st)
        self.dx = None
        if need_dx:
            self.dx = torch.empty(B, *S, self.Cin, dtype=torch.float32, device=self.device)
            g_dx = g_fin.float() if self._bf16() else g_fin      # (the input gradient is an fp32 tensor of the caller)
            self._k("lift_bwd_data", lib.ffno_lift_bwd_data, _p(g_dx), _p(lin_in.weff), _p(self.dx), ws.P_in, self.Cin, C, pm, st)
        multi = self.spectral != "plus" and len(self._fw_sets) == L and L > 1      # per-layer weights: one launch per axis
        if multi:
            real = int(self.spectral == "dct")
            for w in range(len(ws.views)):
                v = ws.views[w]
                # slices per (layer, mode): the grid is slices x modes x LAYERS here, so the per-axis count of the shared-weight
                # launch (512 / K) would make 12 288 workgroups of 37 lines each at the airfoil shape and 384 MB of partial
                # slices for the reduce launch to read back; ~512 workgroups in all are enough
                fwg = int(os.environ.get("FFNO_FW_GRAD_WGS", "512"))      # the knob that sizes ws.nsplit_fw, here per (layer, mode)
                nsf = max(1, min(-(-fwg // (v.K * L)), (v.R + 63) // 64))
                pstride = nsf * 2 * v.K * C * C
                if getattr(ws, "fwpart_multi", None) is None:
                    ws.fwpart_multi, ws.fwgrad_tab, ws.fwgrad_sig = {}, {}, {}
                if w not in ws.fwpart_multi or ws.fwpart_multi[w].numel() != L * pstride:
                    ws.fwpart_multi[w] = torch.empty(L * pstride, dtype=torch.float32, device=self.device)
                ptrs = tuple(gv(self._fw_sets[l][w]).data_ptr() for l in range(L))
                if ws.fwgrad_sig.get(w) != ptrs:
                    ws.fwgrad_tab[w] = torch.tensor(ptrs, dtype=torch.int64).to(self.device)
                    ws.fwgrad_sig[w] = ptrs
                xw, dw = self._fw_words(ws, 0)
                self._k("fw_grad_partial", lib.ffno_fw_grad_partial_multi_h2, _p(ws.SXall[w]), _p(ws.SDall[w]),
                        _p(ws.fwpart_multi[w]), v.R, C, v.K, nsf, L, v.spec, v.spec, pstride, xw, dw, v.L, st)
                self._k("fw_grad_reduce", lib.ffno_fw_grad_reduce_multi, _p(ws.fwpart_multi[w]), _p(ws.fwgrad_tab[w]), L, C, v.K,
                        nsf, pstride, 0, real, st)
        for si, names in enumerate(self._fw_sets if not multi else []):
            layers = [l for l in range(L) if self.fw_names[l] == names]
            l0_, nl = layers[0], len(layers)
            assert layers == list(range(l0_, l0_ + nl))
            if self.spectral == "plus":
                v = ws.views[0]
                self._k("fw_grad_partial", lib.ffno_fw_grad_partial, _p(ws.SXall[0][l0_]), _p(ws.SDall[0][l0_]),
                        _p(ws.fwpart[si][0]), v.R, C, v.K2, ws.nsplit_fw[0], 0, nl, v.spec, v.spec, st)
                self._k("fw2d_grad_reduce", lib.ffno_fw2d_grad_reduce, _p(ws.fwpart[si][0]), _p(gv(names[0])),
                        _p(gv(names[1])), C, self.K, ws.nsplit_fw[0], 0, st)
                continue
            for w, n in enumerate(names):
                v = ws.views[w]
                # dW = sum over the lines of every layer that uses this weight: ONE launch per axis
                xw, dw = self._fw_words(ws, l0_) if self.spectral == "factorized" else (None, None)
                self._k("fw_grad_partial", lib.ffno_fw_grad_partial_h2, _p(ws.SXall[w][l0_]), _p(ws.SDall[w][l0_]),
                        _p(ws.fwpart[si][w]), v.R, C, v.K, ws.nsplit_fw[w], 0, nl, v.spec, v.spec, xw, dw, v.L, st)
                reduce = lib.ffno_fw_grad_reduce_real if self.spectral == "dct" else lib.ffno_fw_grad_reduce
                self._k("fw_grad_reduce", reduce, _p(ws.fwpart[si][w]), _p(gv(n)), C, v.K, ws.nsplit_fw[w], 0, st)
        if self._desc_dev is not None:
            self._k("weightnorm_bwd", lib.ffno_weightnorm_bwd, _p(self._desc_dev), self._n_desc, self._max_rows, st)
        return self.gflat


    def relu_active_sets(self) -> Dict[Tuple[str, int], torch.Tensor]:
        """Diagnostics / parity tests: the ReLU active sets of the last ``forward(save_for_backward=True)``,
        {("backcast" | "forecast", layer): uint8 [pixels of the (padded) activation buffer, hidden]} -- what
        ``threshold_backward`` of feedforward.py:17 would see.  Split-bf16 feed-forward only."""
        if self._saved is not None and self.general_ff:
            _, B, S, _, _ = self._saved
            ws = self._forward_workspace(B, S)
            # a LIST per block: one active set per hidden activation; with dropout a dropped unit counts as inactive
            return {(kind, l): [(h > 0).to(torch.uint8) for h in ws.HG[kind][l]] for kind in ws.HG for l in range(self.L)
                    if not (kind == "backcast" and self.use_fork and l == self.L - 1)}
        if self._saved is None or not self._ffx():
            raise RuntimeError("relu_active_sets() needs a preceding forward(save_for_backward=True) on the ffx path")
        _, B, S, _, _ = self._saved
        ws = self._forward_workspace(B, S)
        lib = _lib.get_lib()
        st = _lib.current_stream(self.device)
        out = {}
        for kind, masks in (("backcast", ws.MASK), ("forecast", getattr(ws, "MASKF", None))):
            if masks is None:
                continue
            for l in range(self.L):
                a = torch.zeros(ws.P, self.H, dtype=torch.uint8, device=self.device)
                self._k("mask_unpack", lib.ffno_ffx_mask_unpack, _p(masks[l]), _p(a), ws.P, self.C, self.H, st)
                out[(kind, l)] = a
        return out


    def dropout_keep_sets(self):
        """Diagnostics / parity tests: the dropout masks of the last training forward, regenerated from their seeds:
        {("backcast" | "forecast", layer): [keep mask (uint8 [pixels, features]) behind every linear of the block]} and
        {"in": keep mask of the lifted features} -- what nn.Dropout drew in the reference (feedforward.py:16, grid_2d.py:158)."""
        if self._saved is None:
            raise RuntimeError("dropout_keep_sets() needs a preceding forward(save_for_backward=True)")
        _, B, S, _, _ = self._saved
        ws = self._forward_workspace(B, S)
        lib = _lib.get_lib()
        st = _lib.current_stream(self.device)
        out = {}

        def mask(n_rows, n_cols, p, seed):
            m = torch.zeros(n_rows, n_cols, dtype=torch.uint8, device=self.device)
            self._k("dropout_mask", lib.ffno_dropout_mask, _p(m), m.numel(), p, seed, st)
            return m

        if self._training and self.in_dropout > 0.0:
            out["in"] = mask(ws.P, self.C, self.in_dropout, self._in_drop_seed)
        if self._training and self.dropout > 0.0:
            for kind in (["backcast"] + (["forecast"] if self.use_fork else [])):
                for l in range(self.L):
                    out[(kind, l)] = [mask(ws.P, self.C if k == self.n_ff - 1 else self.H, *self._drop_args(l, kind, k))
                                      for k in range(self.n_ff)]
        return out


class FFNO2DEngine(FFNOEngine):
    """FNOFactorized2DBlock geometry (the 2-D entry point used by the module mirror and the tests)."""

    def __init__(self, *, modes: int, width: int, input_dim: int, n_layers: int, factor: int, share_weight: bool,
                 share_fork: bool, ff_weight_norm: bool, mode: str = "full", use_fork: bool = False, layer_norm: bool = False,
                 n_ff_layers: int = 2, dropout: float = 0.0, in_dropout: float = 0.0):
        super().__init__(modes=modes, width=width, input_dim=input_dim, n_layers=n_layers, factor=factor,
                         share_weight=share_weight, share_fork=share_fork, ff_weight_norm=ff_weight_norm, mode=mode,
                         spatial_dims=2, padding=0, output_dim=1, use_fork=use_fork, layer_norm=layer_norm,
                         n_ff_layers=n_ff_layers, dropout=dropout, in_dropout=in_dropout)
