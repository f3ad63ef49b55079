"""Drop-in U-Net family for the MI355X path: UNet2d / UNet3d / AnisotropicUNet.

Mirrors the public interface of the reference (torch_em/model/unet.py): constructor
signatures (`:504-519`, `:610-624`, `:701-714`), `init_kwargs`, properties `in_channels /
out_channels / depth` (`:150-160`), `load_{encoder,decoder,base}_state` (`:185-192`), the
`ValueError` messages of `_check_shape` (`:229-235`, `:671-680`), and -- exactly -- the
`state_dict()` keys and shapes (`encoder.blocks.{l}.block.{1,4}.*`, `decoder.samplers.{l}.conv.*`,
`out_conv.*`, ...), so checkpoints are interchangeable in both directions.

The torch.nn sub-modules below are PARAMETER CONTAINERS only (same construction order as the
reference, hence the same random init under the same seed).  `forward` never calls them:
the whole network runs as one autograd node (`engine.UNetFunction`) whose forward and
hand-written backward sequence the HIP kernels of libtem_hip.so over channels-last (NDHWC)
activations: fused pre-norm -> 3x3x3 MFMA conv -> bias -> ReLU blocks, max-pool, 1x1x1 conv +
trilinear upsampling written straight into the skip-concat buffer, fused ReLU/norm backward.
"""
from typing import List, Optional, Union

import numpy as np
import torch
import torch.nn as nn

from . import engine


def get_norm_layer(norm, dim, channels, n_groups=32):
    """Parameter container for the pre-norm of a conv (reference model/unet.py:391-406)."""
    if norm is None:
        return None
    if norm == "InstanceNorm":
        return nn.InstanceNorm2d(channels) if dim == 2 else nn.InstanceNorm3d(channels)
    if norm == "GroupNorm":
        return nn.GroupNorm(min(n_groups, channels), channels)
    if norm == "InstanceNormTrackStats":
        kwargs = {"affine": True, "track_running_stats": True, "momentum": 0.01}
        return nn.InstanceNorm2d(channels, **kwargs) if dim == 2 else nn.InstanceNorm3d(channels, **kwargs)
    if norm == "BatchNorm":
        return nn.BatchNorm2d(channels) if dim == 2 else nn.BatchNorm3d(channels)
    raise ValueError(f"Invalid norm: expect one of 'InstanceNorm', 'BatchNorm' or 'GroupNorm', got {norm}")


class AccumulateChannels(nn.Module):
    """cat([x[:, i0:i1], reduce(x[:, c0:c1], dim=1)]) -- export-time post-processing of affinity predictions
    (reference model/unet.py:15-44); one HIP pass over the (channels-last) prediction, inference only."""

    def __init__(self, invariant_channels, accumulate_channels, accumulator):
        super().__init__()
        self.invariant_channels = invariant_channels
        self.accumulate_channels = accumulate_channels
        assert accumulator in ("mean", "min", "max")
        self.accumulator = accumulator

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("torch_em_amd.model.AccumulateChannels runs on MI355X only; there is no CPU fallback")
        if x.requires_grad:
            raise NotImplementedError("AccumulateChannels is export-time post-processing (no backward)")
        from .. import _lib, ops
        x = x.float()
        N, C = x.shape[:2]
        V = x[0, 0].numel()
        nst = ops._ncv_strides(x)
        if nst is None:
            x = x.contiguous()
            nst = ops._ncv_strides(x)
        sv = nst[2]
        i0, i1 = (0, 0) if self.invariant_channels is None else self.invariant_channels
        c0, c1 = self.accumulate_channels
        out = torch.empty((N, i1 - i0 + 1) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().tem_accumulate_channels(
            ops._p(x), x.stride(0), x.stride(1), sv, ops._p(out), N, C, V, i0, i1, c0, c1,
            {"mean": 0, "min": 1, "max": 2}[self.accumulator], ops._stream(x)), "tem_accumulate_channels")
        return out


POSTPROCESSING = {
    "affinities_to_boundaries_anisotropic": lambda: AccumulateChannels(None, (1, 3), "max"),
    "affinities_to_boundaries2d": lambda: AccumulateChannels(None, (0, 2), "max"),
    "affinities_with_foreground_to_boundaries2d": lambda: AccumulateChannels((0, 1), (1, 3), "max"),
    "affinities_to_boundaries3d": lambda: AccumulateChannels(None, (0, 3), "max"),
    "affinities_with_foreground_to_boundaries3d": lambda: AccumulateChannels((0, 1), (1, 4), "max"),
}


class ConvBlock(nn.Module):
    """[norm -> conv -> ReLU] x 2; `block` indices match the reference (0/3 norms, 1/4 convs)."""

    def __init__(self, in_channels, out_channels, dim, kernel_size=3, padding=1, norm="InstanceNorm"):
        super().__init__()
        self.in_channels, self.out_channels, self.dim, self.norm = in_channels, out_channels, dim, norm
        ks = (kernel_size,) * dim if isinstance(kernel_size, int) else tuple(kernel_size)
        pd = (padding,) * dim if isinstance(padding, int) else tuple(padding)
        if any(k not in (1, 3) for k in ks) or any(p != k // 2 for k, p in zip(ks, pd)):
            raise NotImplementedError(
                f"kernel_size={kernel_size}, padding={padding}: the MI355X path implements 'same' convolutions "
                "with kernel size 1 or 3 per axis (the sizes the reference U-Nets use)"
            )
        conv = nn.Conv2d if dim == 2 else nn.Conv3d
        layers = []
        for cin in (in_channels, out_channels):
            if norm is not None:
                layers.append(get_norm_layer(norm, dim, cin))
            layers.append(conv(cin, out_channels, kernel_size=kernel_size, padding=padding))
            layers.append(nn.ReLU(inplace=True))
        self.block = nn.Sequential(*layers)

    def conv_specs(self):
        mods = list(self.block)
        specs, pending_norm = [], None
        for m in mods:
            if isinstance(m, (nn.GroupNorm, nn.InstanceNorm2d, nn.InstanceNorm3d, nn.BatchNorm2d, nn.BatchNorm3d)):
                pending_norm = m
            elif isinstance(m, (nn.Conv2d, nn.Conv3d)):
                specs.append(engine.ConvSpec(m, pending_norm))
                pending_norm = None
        return specs

    def forward(self, x):  # stand-alone use of a block (e.g. as `base`): run it through the engine
        return engine.run_single_block(self, x)


class ConvBlock2d(ConvBlock):
    def __init__(self, in_channels, out_channels, **kwargs):
        super().__init__(in_channels, out_channels, dim=2, **kwargs)


class ConvBlock3d(ConvBlock):
    def __init__(self, in_channels, out_channels, **kwargs):
        super().__init__(in_channels, out_channels, dim=3, **kwargs)


class Upsampler(nn.Module):
    """interpolate(scale_factor, linear, align_corners=False) then 1x1 conv (reference :444-458).
    The engine applies the (commuting, 8x cheaper) 1x1 conv first and interpolates its output."""

    def __init__(self, scale_factor, in_channels, out_channels, dim, mode):
        super().__init__()
        expected = "bilinear" if dim == 2 else "trilinear"
        if mode != expected:
            raise NotImplementedError(f"Upsampler mode '{mode}': the MI355X path implements '{expected}'")
        self.mode, self.scale_factor, self.dim = mode, scale_factor, dim
        conv = nn.Conv2d if dim == 2 else nn.Conv3d
        self.conv = conv(in_channels, out_channels, 1)


class Upsampler2d(Upsampler):
    def __init__(self, scale_factor, in_channels, out_channels, mode="bilinear"):
        super().__init__(scale_factor, in_channels, out_channels, dim=2, mode=mode)


class Upsampler3d(Upsampler):
    def __init__(self, scale_factor, in_channels, out_channels, mode="trilinear"):
        super().__init__(scale_factor, in_channels, out_channels, dim=3, mode=mode)


def _update_conv_kwargs(kwargs, scale_factor):
    """Anisotropic kernels: size 1 along axes that are not pooled (reference :256-272).

    Reference quirk reproduced on purpose (state_dict shapes depend on it): the reference builds
    `[conv_block_kwargs] * depth` -- ONE dict shared by all levels -- and updates it IN PLACE here
    before any block is constructed, so as soon as one level is anisotropic EVERY block of that
    encoder/decoder gets the anisotropic kernel (the `base` block keeps the isotropic one).
    Verified against the imported reference: tests/golden/g2_aniso_1.npz."""
    if isinstance(scale_factor, int) or len(set(scale_factor)) == 1:
        return kwargs
    ks, pd = kwargs.get("kernel_size", 3), kwargs.get("padding", 1)
    if not (isinstance(ks, int) and isinstance(pd, int)):
        return kwargs
    kwargs.update({"kernel_size": tuple(1 if f == 1 else ks for f in scale_factor),
                   "padding": tuple(0 if f == 1 else pd for f in scale_factor)})
    return kwargs


class Encoder(nn.Module):
    def __init__(self, features, scale_factors, conv_block_impl, pooler_impl, anisotropic_kernel=False,
                 **conv_block_kwargs):
        super().__init__()
        if len(features) != len(scale_factors) + 1:
            raise ValueError("Incompatible number of features {len(features)} and scale_factors {len(scale_factors)}")
        kws = [conv_block_kwargs] * len(scale_factors)
        if anisotropic_kernel:
            kws = [_update_conv_kwargs(kw, sf) for kw, sf in zip(kws, scale_factors)]
        self.blocks = nn.ModuleList([conv_block_impl(i, o, **kw) for i, o, kw in zip(features[:-1], features[1:], kws)])
        self.poolers = nn.ModuleList([pooler_impl(f) for f in scale_factors])
        self.scale_factors = [f for f in scale_factors]
        self.return_outputs = True
        self.in_channels, self.out_channels = features[0], features[-1]

    def __len__(self):
        return len(self.blocks)


class Decoder(nn.Module):
    def __init__(self, features, scale_factors, conv_block_impl, sampler_impl, anisotropic_kernel=False,
                 **conv_block_kwargs):
        super().__init__()
        if len(features) != len(scale_factors) + 1:
            raise ValueError("Incompatible number of features {len(features)} and scale_factors {len(scale_factors)}")
        kws = [conv_block_kwargs] * len(scale_factors)
        if anisotropic_kernel:
            kws = [_update_conv_kwargs(kw, sf) for kw, sf in zip(kws, scale_factors)]
        self.blocks = nn.ModuleList([conv_block_impl(i, o, **kw) for i, o, kw in zip(features[:-1], features[1:], kws)])
        self.samplers = nn.ModuleList(
            [sampler_impl(f, i, o) for f, i, o in zip(scale_factors, features[:-1], features[1:])]
        )
        self.scale_factors = [f for f in scale_factors]
        self.return_outputs = False
        self.in_channels, self.out_channels = features[0], features[-1]

    def __len__(self):
        return len(self.blocks)


class UNetBase(nn.Module):
    """Holds encoder / base / decoder / out_conv and runs them through the HIP engine."""

    def __init__(self, encoder, base, decoder, out_conv=None, final_activation=None, postprocessing=None,
                 check_shape=True):
        super().__init__()
        if len(encoder) != len(decoder):
            raise ValueError(f"Incompatible depth of encoder (depth={len(encoder)}) and decoder (depth={len(decoder)})")
        # one output conv, one per decoder level (side outputs), or none: `out_channels` reports what forward() returns
        per_level = isinstance(out_conv, nn.ModuleList)
        if per_level and len(out_conv) != len(decoder):
            raise ValueError(f"Invalid length of out_conv, expected {len(decoder)}, got {len(out_conv)}")
        self.encoder, self.base, self.decoder, self.out_conv = encoder, base, decoder, out_conv
        self.return_decoder_outputs, self.check_shape = per_level, check_shape
        if per_level:
            self._out_channels = [getattr(conv, "out_channels", None) for conv in out_conv]   # a level without a conv: None
        else:
            self._out_channels = (decoder if out_conv is None else out_conv).out_channels
        self.final_activation = self._get_activation(final_activation)
        self.postprocessing = self._get_postprocessing(postprocessing)

    in_channels = property(lambda self: self.encoder.in_channels)
    out_channels = property(lambda self: self._out_channels)
    depth = property(lambda self: len(self.encoder))

    def _get_activation(self, activation):
        if activation is None:
            return None
        if isinstance(activation, nn.Module):
            return activation
        act = getattr(nn, activation, None) if isinstance(activation, str) else None
        if act is None:
            raise ValueError(f"Invalid activation: {activation}")
        return act()

    def _get_postprocessing(self, postprocessing):
        if postprocessing is None:
            return None
        if isinstance(postprocessing, nn.Module):
            return postprocessing
        if postprocessing in POSTPROCESSING:
            return POSTPROCESSING[postprocessing]()
        raise ValueError(f"Invalid postprocessing: {postprocessing}")

    def load_encoder_state(self, state):
        self.encoder.load_state_dict(state)

    def load_decoder_state(self, state):
        self.decoder.load_state_dict(state)

    def load_base_state(self, state):
        self.base.load_state_dict(state)

    def _check_shape(self, x):
        spatial_shape = tuple(x.shape)[2:]
        factor = [2 ** len(self.encoder)] * len(spatial_shape)
        if any(sh % fac != 0 for sh, fac in zip(spatial_shape, factor)):
            raise ValueError(f"Invalid shape for U-Net: {spatial_shape} is not divisible by {factor}")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if getattr(self, "check_shape", True):
            self._check_shape(x)
        y = engine.unet_forward(self, x)
        if isinstance(y, list):  # return_side_outputs: per-level outputs, full resolution first (reference :211-228)
            if self.final_activation is not None and not engine.fused_activation(self.final_activation):
                y = [self.final_activation(yy) for yy in y]
            if self.postprocessing is not None:
                y = [self.postprocessing(yy) for yy in y]
            return y
        if self.final_activation is not None and not engine.fused_activation(self.final_activation):
            y = self.final_activation(y)
        if self.postprocessing is not None:
            y = self.postprocessing(y)
        return y


def _widths_and_head(conv_cls, in_channels, out_channels, depth, initial_features, gain, return_side_outputs):
    """Channel widths down / up the U and the output conv(s) of the three U-Net constructors -> (down, up, out_conv,
    out_channels as stored in init_kwargs).  The output convs are CREATED HERE, i.e. before encoder / decoder / base: the
    reference draws its random initial weights in that order (tests/golden/g1b_init_unet3d.npz)."""
    level = [initial_features * gain ** i for i in range(depth + 1)]
    down, up = [in_channels] + level[:depth], level[::-1]
    if not return_side_outputs:
        return down, up, (None if out_channels is None else conv_cls(up[-1], out_channels, 1)), out_channels
    if out_channels is None or isinstance(out_channels, int):
        out_channels = [out_channels] * depth
    if len(out_channels) != depth:
        raise ValueError()
    return down, up, nn.ModuleList([conv_cls(f, o, 1) for f, o in zip(up[1:], out_channels)]), out_channels


class UNet2d(UNetBase):
    """2-D U-Net (reference model/unet.py:481-563); runs on the 3-D kernels with D == 1."""

    def __init__(self, in_channels: int, out_channels: int, depth: int = 4, initial_features: int = 32, gain: int = 2,
                 final_activation=None, return_side_outputs: bool = False, conv_block_impl: nn.Module = ConvBlock2d,
                 pooler_impl: nn.Module = nn.MaxPool2d, sampler_impl: nn.Module = Upsampler2d,
                 postprocessing: Optional[Union[nn.Module, str]] = None, check_shape: bool = True,
                 **conv_block_kwargs):
        if pooler_impl is not nn.MaxPool2d:
            raise NotImplementedError("the MI355X path implements nn.MaxPool2d pooling")
        down, up, out_conv, out_channels = _widths_and_head(nn.Conv2d, in_channels, out_channels, depth, initial_features, gain,
                                                            return_side_outputs)
        factors = [2] * depth
        encoder = Encoder(down, factors, conv_block_impl, pooler_impl, **conv_block_kwargs)
        decoder = Decoder(up, factors, conv_block_impl, sampler_impl, **conv_block_kwargs)
        base = conv_block_impl(down[-1], down[-1] * gain, **conv_block_kwargs)
        super().__init__(encoder, base, decoder, out_conv, final_activation, postprocessing, check_shape)
        # what `DefaultTrainer.save_checkpoint` / `get_constructor_arguments` read back (reference util/util.py)
        self.init_kwargs = dict(in_channels=in_channels, out_channels=out_channels, depth=depth, initial_features=initial_features,
                                gain=gain, final_activation=final_activation, return_side_outputs=return_side_outputs,
                                conv_block_impl=conv_block_impl, pooler_impl=pooler_impl, sampler_impl=sampler_impl,
                                postprocessing=postprocessing, **conv_block_kwargs)


class AnisotropicUNet(UNetBase):
    """3-D U-Net with per-level pooling factors (reference model/unet.py:584-680)."""

    def __init__(self, in_channels: int, out_channels: int, scale_factors: List[List[int]], initial_features: int = 32,
                 gain: int = 2, final_activation: Optional[Union[str, nn.Module]] = None,
                 return_side_outputs: bool = False, conv_block_impl: nn.Module = ConvBlock3d,
                 anisotropic_kernel: bool = False, postprocessing: Optional[Union[str, nn.Module]] = None,
                 check_shape: bool = True, **conv_block_kwargs):
        down, up, out_conv, out_channels = _widths_and_head(nn.Conv3d, in_channels, out_channels, len(scale_factors),
                                                            initial_features, gain, return_side_outputs)
        encoder = Encoder(down, scale_factors, conv_block_impl, nn.MaxPool3d, anisotropic_kernel=anisotropic_kernel,
                          **conv_block_kwargs)
        decoder = Decoder(up, scale_factors[::-1], conv_block_impl, Upsampler3d, anisotropic_kernel=anisotropic_kernel,
                          **conv_block_kwargs)
        base = conv_block_impl(down[-1], down[-1] * gain, **conv_block_kwargs)
        super().__init__(encoder, base, decoder, out_conv, final_activation, postprocessing, check_shape)
        self.init_kwargs = dict(in_channels=in_channels, out_channels=out_channels, scale_factors=scale_factors,
                                initial_features=initial_features, gain=gain, final_activation=final_activation,
                                return_side_outputs=return_side_outputs, conv_block_impl=conv_block_impl,
                                anisotropic_kernel=anisotropic_kernel, postprocessing=postprocessing, **conv_block_kwargs)

    def _check_shape(self, x):
        spatial_shape = tuple(x.shape)[2:]
        scale_factors = self.init_kwargs.get("scale_factors", [[2, 2, 2]] * len(self.encoder))
        factor = [int(np.prod([sf[i] for sf in scale_factors])) for i in range(3)]
        if len(spatial_shape) != len(factor):
            raise ValueError(f"Invalid shape for U-Net: dimensions don't agree {len(spatial_shape)} != {len(factor)}")
        if any(sh % fac != 0 for sh, fac in zip(spatial_shape, factor)):
            raise ValueError(f"Invalid shape for U-Net: {spatial_shape} is not divisible by {factor}")


class UNet3d(AnisotropicUNet):
    """Isotropic 3-D U-Net (reference model/unet.py:683-728)."""

    def __init__(self, in_channels: int, out_channels: int, depth: int = 4, initial_features: int = 32, gain: int = 2,
                 final_activation: Optional[Union[str, nn.Module]] = None, return_side_outputs: bool = False,
                 conv_block_impl: nn.Module = ConvBlock3d, postprocessing: Optional[Union[str, nn.Module]] = None,
                 check_shape: bool = True, **conv_block_kwargs):
        super().__init__(in_channels, out_channels, depth * [2], initial_features=initial_features, gain=gain,
                         final_activation=final_activation, return_side_outputs=return_side_outputs,
                         anisotropic_kernel=False, postprocessing=postprocessing, conv_block_impl=conv_block_impl,
                         check_shape=check_shape, **conv_block_kwargs)
        self.init_kwargs = dict(in_channels=in_channels, out_channels=out_channels, depth=depth, initial_features=initial_features,
                                gain=gain, final_activation=final_activation, return_side_outputs=return_side_outputs,
                                conv_block_impl=conv_block_impl, postprocessing=postprocessing, **conv_block_kwargs)
