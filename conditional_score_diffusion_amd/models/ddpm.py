"""DDPM-family score networks as thin adapters over the HIP graph executor.

Registers the reference's names ``ddpm``, ``ddpm_paired_SR3`` and ``ddpm_paired``
(models/ddpm.py:80,275,287) with the reference's constructor / call signatures:

    model = create_model(config)              # cls(config)
    out = model(x | {'x': x, 'y': y}, labels) # Tensor | {'x': .., 'y': ..}

The nn.Module holds ONLY the parameters, named exactly like the reference ``state_dict``
(``all_modules.{i}.Conv_0.weight`` ...; conv OIHW, NIN ``W`` [in,out], Linear [out,in]) so that
reference checkpoints load with ``load_state_dict``; the parameter table itself comes from the
library (csd_unet_param_info), which rebuilds the module list from the config the way
``DDPM.__init__`` does.  ``forward`` hands raw device pointers to ``csd_unet_forward``; all
arithmetic runs in libcsd_hip.so.  There is no PyTorch fallback.
"""
import ctypes
import math
import os

import torch
import torch.nn as nn

from .. import _lib
from .._lib import check, current_stream, lib, ptr, require_gpu_tensor
from . import utils


class _Node(nn.Module):
    """Parameter container: gives nested state_dict keys (``3.Conv_0.weight``)."""


def _fan_avg_uniform(shape, scale, is_nin):
    """``default_init(scale)`` = variance_scaling(scale, 'fan_avg', 'uniform') with the reference's
    default axes in_axis=1, out_axis=0 (models/layers.py:54-91)."""
    rf = 1
    for s in shape[2:]:
        rf *= s
    fan_in, fan_out = shape[1] * rf, shape[0] * rf
    var = (1e-10 if scale == 0 else scale) / ((fan_in + fan_out) / 2.0)
    return (torch.rand(*shape) * 2. - 1.) * math.sqrt(3 * var)


class HipUNet(nn.Module):
    """Base adapter. Subclasses fix (x_channels, y_channels) conventions."""

    arch = 0

    def __init__(self, config, precision=None):
        super().__init__()
        m, d = config.model, config.data
        # arithmetic of the 3x3 contractions: 'fp32' (exact fp32 MFMA), 'fp16x3' (split-fp16 operands, three fp16 MFMAs per product:
        # fp32-class accuracy - the DEFAULT: a model built from an unchanged reference config computes in the reference's precision
        # class), 'fp16f8' (split operands, correction products with e4m3 operands on the fp8 matrix cores: 1e-5-class accuracy on the
        # tested weights, opt-in), 'fp16' (opt-in, not certified).  Not a reference key: config.model.csd_precision or $CSD_PRECISION.
        if precision is None:
            precision = m.get('csd_precision', None) if hasattr(m, 'get') else getattr(m, 'csd_precision', None)
        if precision is None:
            precision = os.environ.get('CSD_PRECISION', 'fp16x3')
        if precision not in _lib.PREC_IDS:
            raise ValueError('unknown csd precision %r (choose from %s)' % (precision, sorted(_lib.PREC_IDS)))
        self.precision = precision
        self.nf = m.nf
        self.num_res_blocks = m.num_res_blocks
        self.attn_resolutions = tuple(m.attn_resolutions)
        self.num_resolutions = len(m.ch_mult)
        self.conditional = bool(m.conditional)
        self.centered = bool(d.centered)
        self.image_size = int(d.effective_image_size)
        self.out_channels = int(self._out_channels(config))
        self.x_channels, self.y_channels = self._channels(config)
        act = m.nonlinearity.lower()
        if act not in _lib.ACT_IDS or act == 'none':
            raise NotImplementedError('activation function does not exist!')   # models/layers.py:29-41
        self._act_name = act
        self._train_calls = 0
        self.train_layout = os.environ.get('CSD_TRAIN_LAYOUT', 'nhwc')     # 'nhwc' | 'nchw' (see _train_forward)
        self.train_executor = os.environ.get('CSD_TRAIN_EXECUTOR', 'planned')   # 'planned' (csd_unet_backward) | 'operators' (autograd)
        self._train_ws = None
        self._train_ws_busy = False        # a forward whose backward has not run yet holds the shared training workspace
        self._train_ws_owner = None        # ... identified by its call index
        self.dropout_seed = int(getattr(config, 'seed', 0) or 0)   # Philox key of the dropout masks
        cfg = _lib.UNetConfig()
        cfg.arch = self.arch
        cfg.nf = m.nf
        cfg.n_levels = len(m.ch_mult)
        for i, v in enumerate(m.ch_mult):
            cfg.ch_mult[i] = int(v)
        cfg.num_res_blocks = m.num_res_blocks
        cfg.n_attn = len(m.attn_resolutions)
        for i, v in enumerate(m.attn_resolutions):
            cfg.attn_resolutions[i] = int(v)
        cfg.image_size = self.image_size
        cfg.x_channels, cfg.y_channels, cfg.out_channels = self.x_channels, self.y_channels, self.out_channels
        cfg.resamp_with_conv = int(bool(m.resamp_with_conv))
        cfg.conditional = int(self.conditional)
        cfg.centered = int(self.centered)
        cfg.act = _lib.ACT_IDS[act]
        cfg.precision = _lib.PREC_IDS[precision]
        self._extra_config(cfg, config)
        self._cfg = cfg
        self._h = ctypes.c_void_p()
        check(lib().csd_unet_create(ctypes.byref(cfg), ctypes.byref(self._h)), 'unet_create')
        self._dropout = float(m.get('dropout', 0.0)) if hasattr(m, 'get') else float(getattr(m, 'dropout', 0.0))
        self._build_params()
        self._packed = None
        self._packed_key = None
        self._ws = None

    # -- parameters ------------------------------------------------------------------------------
    def _channels(self, config):
        raise NotImplementedError

    def _out_channels(self, config):
        return config.model.output_channels

    def _extra_config(self, cfg, config):
        """hook: architecture-specific fields of csd_unet_config"""

    def _param_table(self):
        n = lib().csd_unet_num_params(self._h)
        out = []
        name, ndim, shape = ctypes.c_char_p(), ctypes.c_int(), (ctypes.c_int64 * 4)()
        for i in range(n):
            check(lib().csd_unet_param_info(self._h, i, ctypes.byref(name), ctypes.byref(ndim), shape), 'param_info')
            out.append((name.value.decode(), tuple(shape[j] for j in range(ndim.value))))
        return out

    def _build_params(self):
        table = self._param_table()
        last_idx = max(int(k.split('.')[1]) for k, _ in table)
        nodes = {}
        for key, shape in table:
            parts = key.split('.')           # all_modules, idx, [sub], leaf
            idx = int(parts[1])
            node = nodes.setdefault(idx, _Node())
            for sub in parts[2:-1]:
                if not hasattr(node, sub):
                    node.add_module(sub, _Node())
                node = getattr(node, sub)
            leaf = parts[-1]
            if len(shape) >= 2:
                is_nin = leaf == 'W'
                if 'Conv_1' in key or 'NIN_3' in key or (idx == last_idx and len(shape) == 4):
                    scale = 0.          # init_scale=0 -> 1e-10 (models/layers.py:648,575; ddpm.py:146)
                elif is_nin:
                    scale = 0.1         # NIN default init_scale (models/layers.py:556)
                else:
                    scale = 1.
                val = _fan_avg_uniform(shape, scale, is_nin)
            elif leaf == 'weight':      # GroupNorm gamma
                val = torch.ones(shape)
            else:
                val = torch.zeros(shape)
            node.register_parameter(leaf, nn.Parameter(val))
        self.all_modules = nn.ModuleList([nodes[i] for i in sorted(nodes)])
        assert sorted(nodes) == list(range(len(nodes)))
        self._param_names = [k for k, _ in table]

    @property
    def device(self):
        return next(self.parameters()).device

    def __del__(self):
        try:
            if getattr(self, '_h', None):
                lib().csd_unet_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- packing -----------------------------------------------------------------------------------
    def _ensure_packed(self):
        params = dict(self.named_parameters())
        key = (_lib.WEIGHT_EPOCH[0],) + tuple((p.data_ptr(), p._version) for p in params.values())
        if self._packed is not None and key == self._packed_key:
            return
        dev = self.device
        if dev.type != 'cuda':
            raise RuntimeError('the HIP score network runs on the MI355X only: move the model with .to("cuda") '
                               '(parameters are on %s); there is no CPU fallback' % dev)
        for name in self._param_names:
            p = params[name]
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError('parameter %s must be contiguous float32' % name)
            check(lib().csd_unet_set_param(self._h, name.encode(), ctypes.c_void_p(p.data_ptr()), p.numel()),
                  'set_param')
        nbytes = lib().csd_unet_packed_bytes(self._h)
        if self._packed is None or self._packed.numel() < nbytes or self._packed.device != dev:
            self._packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        check(lib().csd_unet_pack(self._h, ptr(self._packed), current_stream(dev)), 'unet_pack')
        self._packed_key = key

    def _workspace(self, B):
        need = lib().csd_unet_workspace_bytes(self._h, B)
        if need == 0:
            raise RuntimeError('libcsd_hip: cannot plan batch %d: %s' % (B, lib().csd_last_error().decode()))
        if self._ws is None or self._ws.numel() < need or self._ws.device != self.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def stats(self, B):
        """(kernel launches, algorithmic FLOPs, algorithmic bytes) of one evaluation at batch B."""
        n, fl, by = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        check(lib().csd_unet_stats(self._h, B, ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by)), 'unet_stats')
        return n.value, fl.value, by.value

    # -- evaluation --------------------------------------------------------------------------------
    def _run(self, x, y, labels, y_noise=None, y_sigma=0.0):
        require_gpu_tensor(x, 'x')
        if self.training and torch.is_grad_enabled():
            if y_noise is not None:
                raise RuntimeError('the fused y-perturbation is a sampling feature; perturb y before a training step')
            return self._train_forward(x, y, labels)
        B = x.shape[0]
        S = self.image_size
        if tuple(x.shape) != (B, self.x_channels, S, S):
            raise RuntimeError('x has shape %s, expected %s' % (tuple(x.shape), (B, self.x_channels, S, S)))
        if self.y_channels:
            require_gpu_tensor(y, 'y')
            if tuple(y.shape) != (B, self.y_channels, S, S):
                raise RuntimeError('y has shape %s, expected %s' % (tuple(y.shape), (B, self.y_channels, S, S)))
            y = y.contiguous()
        labels = labels.to(device=x.device, dtype=torch.float32).contiguous()
        if labels.shape != (B,):
            raise RuntimeError('labels must have shape [%d]' % B)
        self._ensure_packed()
        ws = self._workspace(B)
        out = torch.empty(B, self.out_channels, S, S, dtype=torch.float32, device=x.device)
        check(lib().csd_unet_forward(self._h, ptr(self._packed), ptr(ws), ws.numel(), ptr(x.contiguous()),
                                     ptr(y) if self.y_channels else None, ptr(labels), ptr(out), B,
                                     ptr(y_noise) if y_noise is not None else None, float(y_sigma),
                                     current_stream(x.device)), 'unet_forward')
        return out


    # -- training-mode evaluation: ONE planned graph behind the C ABI (csd_unet_train_forward / csd_unet_backward) -------------------
    def _train_workspace(self, B):
        need = lib().csd_unet_train_workspace_bytes(self._h, B, self._dropout)
        if need == 0:
            raise RuntimeError('libcsd_hip: cannot plan the training graph at batch %d: %s' % (B, lib().csd_last_error().decode()))
        if self._train_ws is None or self._train_ws.numel() < need or self._train_ws.device != self.device:
            self._train_ws = None
            self._train_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._train_ws

    def _train_params(self):
        params = dict(self.named_parameters())
        return [params[name] for name in self._param_names]

    def _train_forward_planned(self, x, y, labels):
        """``model.train()`` + autograd as ONE node: the forward is csd_unet_train_forward (the reference's layer sequence,
        models/ddpm.py:149-213, dropout on, activations kept in a library workspace), the backward csd_unet_backward (every
        parameter gradient from one call).  ``self.grad_sink``: when a Trainer owns the flat gradient buffer the gradients are
        written straight into ``p.grad`` (which it zeroed) and autograd is handed nothing to accumulate."""
        B, S = x.shape[0], self.image_size
        if tuple(x.shape) != (B, self.x_channels, S, S):
            raise RuntimeError('x has shape %s, expected %s' % (tuple(x.shape), (B, self.x_channels, S, S)))
        if self.y_channels:
            require_gpu_tensor(y, 'y')
            if tuple(y.shape) != (B, self.y_channels, S, S):
                raise RuntimeError('y has shape %s, expected %s' % (tuple(y.shape), (B, self.y_channels, S, S)))
        labels = labels.to(device=x.device, dtype=torch.float32).contiguous()
        if labels.shape != (B,):
            raise RuntimeError('labels must have shape [%d]' % B)
        self._train_calls += 1
        return _PlannedNet.apply(self, x.contiguous(), y.contiguous() if self.y_channels else None, labels, *self._train_params())

    # -- training-mode evaluation: differentiable, operator-granular ------------------------------------------
    def _train_forward(self, x, y, labels):
        """``model.train()`` + autograd: the reference forward (models/ddpm.py:149-213) layer by layer on differentiable HIP
        operators (forward AND backward kernels behind include/csd.h), with ``nn.Dropout`` active (models/layers.py:647,662).
        The planned graph executor (csd_unet_forward) stays the inference path; this one keeps every activation that a
        gradient needs.  ``self.train_layout``: 'nhwc' (default) keeps activations in the library's layout between layers
        (grad_ops_nhwc: no layout change anywhere but the network's NCHW input and output), 'nchw' runs every layer through
        the NCHW per-operator ABI (grad_ops)."""
        from .. import ops
        if not self._cfg.resamp_with_conv:
            raise NotImplementedError('training with resamp_with_conv=False is not provided')
        if self.train_executor == 'planned' and self.arch == 0 and self.train_layout == 'nhwc':
            return self._train_forward_planned(x, y, labels)
        nhwc = self.train_layout == 'nhwc'
        if nhwc:
            from .. import grad_ops_nhwc as G
        else:
            from .. import grad_ops as G
        m, prec, act = self.all_modules, self.precision, self._act_name
        B, S = x.shape[0], self.image_size
        if tuple(x.shape) != (B, self.x_channels, S, S):
            raise RuntimeError('x has shape %s, expected %s' % (tuple(x.shape), (B, self.x_channels, S, S)))
        labels = labels.to(device=x.device, dtype=torch.float32).contiguous()
        self._train_calls += 1
        drop = [0]
        cdim = 3 if nhwc else 1            # channel axis of an activation
        side = (lambda t: t.shape[1]) if nhwc else (lambda t: t.shape[-1])
        bias_add = G.bias_add if nhwc else G.bias_add_nchw

        def dropout(h):
            drop[0] += 1
            return G.dropout(h, self._dropout, self.dropout_seed, (self._train_calls << 16) + drop[0])

        def res(node, h, temb):
            cin, cout = h.shape[cdim], node.Conv_0.weight.shape[0]
            t = G.groupnorm_act(h, node.GroupNorm_0.weight, node.GroupNorm_0.bias, 32, 1e-6, act)
            t = G.conv2d(t, node.Conv_0.weight, node.Conv_0.bias, precision=prec)
            if temb is not None:
                t = bias_add(t, G.linear(temb, node.Dense_0.weight, node.Dense_0.bias, act_in=act))
            t = G.groupnorm_act(t, node.GroupNorm_1.weight, node.GroupNorm_1.bias, 32, 1e-6, act)
            t = G.conv2d(dropout(t), node.Conv_1.weight, node.Conv_1.bias, precision=prec)
            if cin != cout:
                h = G.nin(h, node.NIN_0.W, node.NIN_0.b, prec)
            return G.axpby(h, t)

        def attn(node, h):
            t = G.groupnorm_act(h, node.GroupNorm_0.weight, node.GroupNorm_0.bias, 32, 1e-6, 'none')
            if nhwc:      # q | k | v from ONE 1x1 contraction (weights concatenated: data movement), packed per pixel
                W = torch.cat([node.NIN_0.W, node.NIN_1.W, node.NIN_2.W], dim=1)
                b = torch.cat([node.NIN_0.b, node.NIN_1.b, node.NIN_2.b])
                a = G.attention(G.nin(t, W, b, prec))
            else:
                a = G.attention(G.nin(t, node.NIN_0.W, node.NIN_0.b, prec), G.nin(t, node.NIN_1.W, node.NIN_1.b, prec),
                                G.nin(t, node.NIN_2.W, node.NIN_2.b, prec))
            return G.axpby(h, G.nin(a, node.NIN_3.W, node.NIN_3.b, prec))

        h = torch.cat([x, y], dim=1) if self.y_channels else x
        i = 0
        temb = None
        if self.conditional:
            temb = ops.timestep_embedding(labels, self.nf)
            temb = G.linear(temb, m[0].weight, m[0].bias)
            temb = G.linear(temb, m[1].weight, m[1].bias, act_in=act)
            i = 2
        if not self.centered:
            h = G.axpby(h, None, 2.0, 0.0, -1.0, 1.0)
        if nhwc:          # network input is the reference's NCHW; from here on activations are [B, H, W, C]
            hs = [G.conv2d(h, m[i].weight, m[i].bias, precision=prec, layout=G.OUT_NHWC)]
        else:
            hs = [G.conv2d(h, m[i].weight, m[i].bias, precision=prec)]
        i += 1
        for lvl in range(self.num_resolutions):
            for _ in range(self.num_res_blocks):
                h = res(m[i], hs[-1], temb)
                i += 1
                if side(h) in self.attn_resolutions:
                    h = attn(m[i], h)
                    i += 1
                hs.append(h)
            if lvl != self.num_resolutions - 1:
                hs.append(G.conv2d(hs[-1], m[i].Conv_0.weight, m[i].Conv_0.bias, stride=2, downsample_pad=True, precision=prec))
                i += 1
        h = res(m[i], hs[-1], temb)
        h = attn(m[i + 1], h)
        h = res(m[i + 2], h, temb)
        i += 3
        for lvl in reversed(range(self.num_resolutions)):
            for _ in range(self.num_res_blocks + 1):
                h = res(m[i], torch.cat([h, hs.pop()], dim=cdim), temb)
                i += 1
            if side(h) in self.attn_resolutions:
                h = attn(m[i], h)
                i += 1
            if lvl != 0:
                h = G.conv2d(h, m[i].Conv_0.weight, m[i].Conv_0.bias, up2=True, precision=prec)
                i += 1
        assert not hs and i == len(m) - 2
        h = G.groupnorm_act(h, m[i].weight, m[i].bias, 32, 1e-6, act)
        if nhwc:
            return G.conv2d(h, m[i + 1].weight, m[i + 1].bias, precision=prec, layout=G.IN_NHWC)
        return G.conv2d(h, m[i + 1].weight, m[i + 1].bias, precision=prec)


def _release_shared_ws(model_ref, owner):
    """the forward that held the model's shared training workspace is done with it (its backward ran, or its graph was dropped).
    Ownership is keyed by the forward's call index, not by the buffer: the graph of iteration N is usually freed only after
    iteration N + 1's forward has taken the (same) workspace, and a release keyed by the pointer would then free it under the new owner."""
    model = model_ref()
    if model is not None and model._train_ws_busy and getattr(model, '_train_ws_owner', None) == owner:
        model._train_ws_busy = False
        model._train_ws_owner = None


def _release_private_ws(model_ref, handle, ws_ptr, call):
    """a monitoring forward's private workspace is about to be freed: drop the library's record of THAT forward (by call index: a late
    finalizer must not erase the record of a newer forward that was handed the same address - csd_unet_train_release_call)"""
    model = model_ref()
    if model is not None and getattr(model, '_h', None) is handle and handle is not None:
        lib().csd_unet_train_release_call(handle, ctypes.c_void_p(ws_ptr), ctypes.c_uint64(call))


class _PlannedNet(torch.autograd.Function):
    """The whole training-mode network as one autograd node over csd_unet_train_forward / csd_unet_backward."""

    @staticmethod
    def forward(ctx, model, x, y, labels, *params):
        B = x.shape[0]
        for name, p in zip(model._param_names, params):
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != x.device:
                raise RuntimeError('parameter %s must be contiguous float32 on %s' % (name, x.device))
        # the shared training workspace belongs to ONE live forward; a second one (e.g. a monitoring forward between a training
        # forward and its backward) gets a workspace of its own, released with its autograd context
        shared = not model._train_ws_busy
        ws = model._train_workspace(B) if shared else torch.empty(
            lib().csd_unet_train_workspace_bytes(model._h, B, model._dropout), dtype=torch.uint8, device=x.device)
        table = (ctypes.c_void_p * len(params))(*[p.data_ptr() for p in params])
        out = torch.empty(B, model.out_channels, model.image_size, model.image_size, dtype=torch.float32, device=x.device)
        check(lib().csd_unet_train_forward(model._h, table, ptr(ws), ws.numel(), ptr(x), ptr(y) if y is not None else None,
                                           ptr(labels), ptr(out), B, model._dropout, model.dropout_seed, model._train_calls,
                                           current_stream(x.device)), 'unet_train_forward')
        ctx.model, ctx.table, ctx.ws, ctx.B, ctx.call = model, table, ws, B, model._train_calls
        ctx.fin = None
        if shared:
            model._train_ws_busy = True
            model._train_ws_owner = ctx.call
            import weakref
            ctx.fin = weakref.finalize(ctx, _release_shared_ws, weakref.ref(model), ctx.call)
        else:
            import weakref
            weakref.finalize(ctx, _release_private_ws, weakref.ref(model), model._h, ws.data_ptr(), ctx.call)
        ctx.shapes = [(p.shape, p.numel()) for p in params]
        # direct mode: the backward writes straight into the parameters' .grad views (the flat gradient buffer).  A parameter that
        # takes no gradient (the fixed Gaussian-Fourier W of NCSN++) has none: the library still writes one - into a throw-away buffer
        ctx.sink = [p.grad if p.requires_grad else False for p in params] if getattr(model, 'grad_sink', False) else None
        return out

    @staticmethod
    def backward(ctx, dout):
        model, B = ctx.model, ctx.B
        dout = dout.contiguous()
        direct = ctx.sink is not None and all(g is False or (g is not None and g.is_contiguous() and g.data_ptr() % 16 == 0)
                                              for g in ctx.sink)
        # (distributed.GradSync reads this: the gradient-ready events of the planned backward describe the FINAL gradients only
        # when the kernels wrote .grad themselves; otherwise autograd's accumulation runs after them)
        model._last_backward_direct = direct
        if direct:
            grads = [torch.empty(shape, dtype=torch.float32, device=dout.device) if g is False else g
                     for g, (shape, _) in zip(ctx.sink, ctx.shapes)]
        else:
            offs = [0]
            for _, n in ctx.shapes:
                offs.append(offs[-1] + (n + 3) // 4 * 4)
            flat = torch.empty(offs[-1], dtype=torch.float32, device=dout.device)
            grads = [flat[o:o + n].view(shape) for o, (shape, n) in zip(offs, ctx.shapes)]
        gtable = (ctypes.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
        check(lib().csd_unet_backward(model._h, ctx.table, gtable, ptr(ctx.ws), ctx.ws.numel(), ptr(dout), B, ctx.call,
                                      current_stream(dout.device)), 'unet_backward')
        if ctx.fin is not None:          # (the backward releases the workspace itself; the finalizer of this context must not fire later)
            ctx.fin.detach()
            _release_shared_ws(lambda: model, ctx.call)
        if direct:
            return (None, None, None, None) + (None,) * len(grads)
        return (None, None, None, None) + tuple(g if need else None for g, need in zip(grads, ctx.needs_input_grad[4:]))


@utils.register_model(name='ddpm')
class DDPM(HipUNet):
    """Unconditional DDPM U-Net: ``model(x, labels) -> Tensor`` (models/ddpm.py:80-213)."""

    def _channels(self, config):
        return int(config.model.input_channels), 0

    def forward(self, x, labels):
        return self._run(x, None, labels)


class _Paired(HipUNet):
    def _channels(self, config):
        cx = int(config.data.shape_x[0])
        return cx, int(config.model.input_channels) - cx


@utils.register_model(name='ddpm_paired_SR3')
class DDPM_paired_SR3(_Paired):
    """CDE / SR3 estimator: net(cat(x, y)) -> score_x (models/ddpm.py:275-285)."""

    def forward(self, input_dict, labels):
        return self._run(input_dict['x'], input_dict['y'], labels)


@utils.register_model(name='ddpm_paired')
class DDPM_paired(_Paired):
    """CMDE / VS-CMDE estimator: net(cat(x, y)) -> {'x': .., 'y': ..} (models/ddpm.py:287-298)."""

    def forward(self, input_dict, labels):
        out = self._run(input_dict['x'], input_dict['y'], labels)
        c = self.x_channels
        return {'x': out[:, :c], 'y': out[:, c:]}
