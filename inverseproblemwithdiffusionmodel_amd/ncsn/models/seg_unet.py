"""Segmentation network of the ACDC sampler and its likelihood gradient.

The reference builds ``monai.networks.nets.UNet(spatial_dims=2, in_channels=1, out_channels=2,
channels=[64, 128, 256, 512, 1024], strides=[2, 2, 2, 2])`` (``ncsn/configs/general_config.yml:1-6``,
``helpers/load_model.py:30,140-141``) and differentiates ``sum log softmax(seg(X))[label]`` with respect to X by autograd
(``ncsn/models/__init__.py:197-215``).  MONAI is not vendored (version unpinned): this is a restatement of
the published architecture for ``num_res_units=0`` (the reference's arguments) --
    level:  Convolution(stride 2) = Conv2d(3x3, s2, p1) -> InstanceNorm2d (no affine) -> PReLU
            SkipConnection(cat) around the next level / the bottom Convolution(stride 1)
            transposed Convolution = ConvTranspose2d(3x3, s2, p1, output_padding 1) -> InstanceNorm -> PReLU  (top: conv only)
-- with MONAI's parameter names (``model.0.conv.weight``, ``model.0.adn.A.weight``, ``model.1.submodule...``,
``model.2.conv.weight``), so that a Lightning ``TrainSeg`` checkpoint loads by key.  PARITY UNPINNED against MONAI
itself; pinned against the torch-CPU restatement + autograd in ``oracle/seg_unet.py``.

Execution: no autograd.  ``loglh_grad`` runs the forward and the input-gradient as one chain of libipdm.so launches: a
stride-2 convolution is the stride-1 MFMA convolution followed by even-position subsampling, a transposed convolution is
zero insertion followed by the stride-1 convolution with flipped weights, and each one's input-gradient is the other
form with transposed weights; InstanceNorm + PReLU have a fused forward and a fused backward kernel."""
import torch
import torch.nn as nn

from ... import ops


class _ADN(nn.Module):
    """norm (InstanceNorm2d, no parameters) -> dropout (none) -> act: MONAI's 'NDA' block; only the PReLU has a parameter"""

    def __init__(self):
        super().__init__()
        self.A = nn.PReLU()                                    # one slope, init 0.25 -> key 'adn.A.weight'


class _Convolution(ops.PackedWeightMixin, nn.Module):
    def __init__(self, cin, cout, stride, transposed=False, conv_only=False):
        super().__init__()
        self.cin, self.cout, self.stride, self.transposed, self.conv_only = cin, cout, stride, transposed, conv_only
        if transposed:
            self.conv = nn.ConvTranspose2d(cin, cout, 3, stride=stride, padding=1, output_padding=stride - 1)
        else:
            self.conv = nn.Conv2d(cin, cout, 3, stride=stride, padding=1)
        if not conv_only:
            self.adn = _ADN()
        self._cache = ops.PackedWeightCache()

    # stride-1 'same' convolution weights [Cout'][Cin'][3][3] for the forward and for the input-gradient
    def _w_fwd(self, w):
        return w.flip(2, 3).permute(1, 0, 2, 3).contiguous() if self.transposed else w

    def _w_bwd(self, w):
        return w.contiguous() if self.transposed else w.flip(2, 3).permute(1, 0, 2, 3).contiguous()

    def packed(self, which):
        build = (lambda w: ops.conv_weight(self._w_fwd(w))) if which == "fwd" else (lambda w: ops.conv_weight(self._w_bwd(w)))
        return self._cache.get(self.conv.weight, f"{which}_{ops.CONV_IMPL}", build)

    def forward_saved(self, x):
        """-> (y, saved) with saved = (xhat, rstd) or None"""
        if self.transposed and self.stride == 2:
            x = ops.zero_insert2(x)
        # (f16x2 family: every convolution here measures its input -- the state, InstanceNorm'ed activations and, on the way back,
        #  gradients many orders of magnitude below 1 -- and runs with the dynamic range; ~2 % of a guided iteration)
        c = ops.conv2d(x, self.packed("fwd"), self.conv.bias.data, in_amax=ops.in_amax_for(x))
        if not self.transposed and self.stride == 2:
            c = ops.subsample2(c)
        if self.conv_only:
            return c, None
        xhat, y, rstd = ops.in_prelu_fwd(c, self.adn.A.weight.data)
        return y, (xhat, rstd)

    def backward_input(self, g, saved):
        if not self.conv_only:
            g = ops.in_prelu_bwd(g, saved[0], saved[1], self.adn.A.weight.data)
        if not self.transposed and self.stride == 2:
            g = ops.zero_insert2(g)
        g = ops.conv2d(g, self.packed("bwd"), in_amax=ops.in_amax_for(g))
        if self.transposed and self.stride == 2:
            g = ops.subsample2(g)
        return g


class _Skip(nn.Module):
    def __init__(self, submodule):
        super().__init__()
        self.submodule = submodule


class UNet(nn.Module):
    def __init__(self, spatial_dims=2, in_channels=1, out_channels=2, channels=(64, 128, 256, 512, 1024),
                 strides=(2, 2, 2, 2), kernel_size=3, up_kernel_size=3, num_res_units=0, **kwargs):
        super().__init__()
        if spatial_dims != 2 or kernel_size != 3 or up_kernel_size != 3 or num_res_units != 0 or any(s != 2 for s in strides):
            raise NotImplementedError("UNet: only the reference's configuration family (2-D, 3x3, stride 2, no residual units)")
        if len(channels) != len(strides) + 1 or len(channels) < 2:
            raise ValueError("UNet: len(channels) must be len(strides) + 1")
        self.in_channels, self.out_channels, self.channels, self.strides = in_channels, out_channels, tuple(channels), tuple(strides)

        def block(inc, outc, chans, strs, is_top):
            c, s = chans[0], strs[0]
            if len(chans) > 2:
                sub, upc = block(c, c, chans[1:], strs[1:], False), c * 2
            else:
                sub, upc = _Convolution(c, chans[1], 1), c + chans[1]          # bottom layer
            down = _Convolution(inc, c, s)
            up = _Convolution(upc, outc, s, transposed=True, conv_only=is_top)
            return nn.Sequential(down, _Skip(sub), up)

        self.model = block(in_channels, out_channels, list(channels), list(strides), True)

    # ---- forward (logits) and forward + input-gradient, both as launch chains ------------------------------------
    def _fwd(self, seq, x):
        """-> (y, ctx): ctx is the nested record the backward walk needs (modules + saved (xhat, rstd) pairs)"""
        if isinstance(seq, _Convolution):                                        # bottom layer
            y, saved = seq.forward_saved(x)
            return y, ("bottom", seq, saved)
        down, skip, up = seq[0], seq[1], seq[2]
        d, sd = down.forward_saved(x)
        inner, ci = self._fwd(skip.submodule, d)
        y, su = up.forward_saved(torch.cat([d, inner], dim=1))                   # SkipConnection: cat([x, submodule(x)])
        return y, ("level", down, sd, ci, d.shape[1], up, su)

    def _bwd(self, ctx, g):
        if ctx[0] == "bottom":
            return ctx[1].backward_input(g, ctx[2])
        _, down, sd, ci, c_skip, up, su = ctx
        gcat = up.backward_input(g, su)
        g_d = self._bwd(ci, gcat[:, c_skip:].contiguous())                       # through the inner branch ...
        g_d = ops.add(g_d, gcat[:, :c_skip].contiguous())                        # ... plus the skip half of the concat
        return down.backward_input(g_d, sd)

    def _check_size(self, x):
        """every stride-2 level halves the image and the transposed convolutions double it back: H and W must be multiples
        of 2^len(strides), otherwise the skip concatenations / the backward walk meet mismatched shapes mid-chain"""
        m = 2 ** len(self.strides)
        if x.dim() != 4 or x.shape[-2] % m or x.shape[-1] % m:
            raise ValueError(f"UNet: input {tuple(x.shape)}: H and W must be multiples of {m} (2^len(strides))")

    @torch.no_grad()
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("UNet: expected GPU tensors (no CPU fallback in this build)")
        self._check_size(x)
        return self._fwd(self.model, x.contiguous().float())[0]

    @torch.no_grad()
    def loglh_grad(self, x, label, mode="full"):
        """compute_seg_grad (ncsn/models/__init__.py:197-215): d/dx sum log softmax(self(x), dim=1)[label], x (B, C, H, W),
        label (B, 1, H, W) int64 (or (1, 1, H, W), shared by the batch); mode 'FG' multiplies by the label"""
        assert mode in ["full", "FG"]
        if not x.is_cuda:
            raise RuntimeError("UNet.loglh_grad: expected GPU tensors (no CPU fallback in this build)")
        self._check_size(x)
        label = label.to(x.device, torch.int64)
        if label.shape[0] != x.shape[0]:
            label = label.expand(x.shape[0], *label.shape[1:])
        label = label.contiguous()
        logits, ctx = self._fwd(self.model, x.contiguous().float())
        g = self._bwd(ctx, ops.seg_loglh_grad(logits, label))
        if mode == "FG":
            g = g * label.to(g.dtype)
        return g
