"""CPU restatement of the reference U-Net forward (TEST INFRASTRUCTURE, see oracle/__init__.py).

Functional, driven by a reference-format state_dict, so the same weights go through the
reference (when generating fixtures), this oracle, and the HIP engine.  Follows:
  UNetBase._apply_default   /root/reference/torch_em/model/unet.py:194-209
  Encoder.forward           :311-321   (block -> keep skip -> MaxPool(factor))
  ConvBlock                 :429-438   (norm -> conv(k, pad=k//2) -> ReLU) x 2, norm=None drops norms
  get_norm_layer            :391-406   (InstanceNorm: no affine, eps 1e-5; GroupNorm(min(32,C), C))
  Upsampler.forward         :455-458   (interpolate(scale, linear, align_corners=False) -> 1x1 conv)
  Decoder.forward/_concat   :363-388   (cat([upsampled, skip], dim=1) -> block)
  out_conv / activation     :202-205
Gradients come from torch autograd over these ops (which is what the reference's
loss.backward() does as well).
"""
import torch
import torch.nn.functional as F


def _conv(x, w, b):
    pad = tuple(k // 2 for k in w.shape[2:])
    return (F.conv2d if w.dim() == 4 else F.conv3d)(x, w, b, padding=pad)


def _norm(x, norm, gamma, beta, n_groups=32, buffers=None, training=True):
    """buffers = (running_mean, running_var) for the stateful norms (updated in place in training mode, like torch)."""
    if norm is None:
        return x
    if norm == "BatchNorm":  # nn.BatchNorm3d defaults: eps 1e-5, momentum 0.1, affine
        return F.batch_norm(x, buffers[0], buffers[1], gamma, beta, training=training, momentum=0.1, eps=1e-5)
    if norm == "InstanceNormTrackStats":  # reference model/unet.py:398-400: affine, running stats, momentum 0.01
        return F.instance_norm(x, buffers[0], buffers[1], gamma, beta, use_input_stats=training, momentum=0.01, eps=1e-5)
    if norm == "InstanceNorm":
        return F.instance_norm(x, eps=1e-5)
    if norm == "GroupNorm":
        c = x.shape[1]
        return F.group_norm(x, min(n_groups, c), gamma, beta, eps=1e-5)
    raise ValueError(norm)


def _block(sd, prefix, x, norm, training=True):
    idx = (1, 4) if norm is not None else (0, 2)
    nidx = (0, 3)
    affine = norm in ("GroupNorm", "BatchNorm", "InstanceNormTrackStats")
    for j in range(2):
        gamma = sd.get(f"{prefix}.block.{nidx[j]}.weight") if affine else None
        beta = sd.get(f"{prefix}.block.{nidx[j]}.bias") if affine else None
        buffers = None
        if norm in ("BatchNorm", "InstanceNormTrackStats"):
            buffers = (sd[f"{prefix}.block.{nidx[j]}.running_mean"], sd[f"{prefix}.block.{nidx[j]}.running_var"])
        x = _norm(x, norm, gamma, beta, buffers=buffers, training=training)
        x = F.relu(_conv(x, sd[f"{prefix}.block.{idx[j]}.weight"], sd[f"{prefix}.block.{idx[j]}.bias"]))
    return x


def unet_forward(sd, x, scale_factors, norm="InstanceNorm", final_activation=None, training=True):
    """sd: reference-layout state_dict (tensors, may require grad); x: [N,C,*spatial];
    scale_factors: per-level pooling factor (int or list), encoder order."""
    dim = x.dim() - 2
    depth = len(scale_factors)
    pool = F.max_pool2d if dim == 2 else F.max_pool3d
    mode = "bilinear" if dim == 2 else "trilinear"
    skips = []
    for l in range(depth):
        x = _block(sd, f"encoder.blocks.{l}", x, norm, training)
        skips.append(x)
        f = scale_factors[l]
        x = pool(x, f if isinstance(f, int) else tuple(f))
    x = _block(sd, "base", x, norm, training)
    dec_out = []
    for i in range(depth):
        f = scale_factors[depth - 1 - i]
        x = F.interpolate(x, scale_factor=f if isinstance(f, int) else tuple(float(v) for v in f), mode=mode,
                          align_corners=False)
        x = _conv(x, sd[f"decoder.samplers.{i}.conv.weight"], sd[f"decoder.samplers.{i}.conv.bias"])
        # Decoder._concat / _crop (reference model/unet.py:363-373): the skip tensor is centre-cropped by (difference // 2)
        # per side in EVERY dimension before the concat (an odd difference leaves it one too large: torch.cat raises)
        skip = skips[depth - 1 - i]
        off = [(a - b) // 2 for a, b in zip(skip.shape, x.shape)]
        skip = skip[tuple(slice(o, a - o) for o, a in zip(off, skip.shape))]
        x = torch.cat([x, skip], dim=1)
        x = _block(sd, f"decoder.blocks.{i}", x, norm, training)
        dec_out.append(x)
    if "out_conv.0.weight" in sd:
        # return_side_outputs (UNetBase._apply_with_side_outputs, :211-228): one 1x1 conv per decoder level,
        # activation on each, list reversed so that the full-resolution output comes first
        outs = [_conv(d, sd[f"out_conv.{i}.weight"], sd[f"out_conv.{i}.bias"]) for i, d in enumerate(dec_out)]
        if final_activation == "Sigmoid":
            outs = [torch.sigmoid(o) for o in outs]
        elif final_activation is not None:
            raise ValueError(final_activation)
        return outs[::-1]
    if "out_conv.weight" in sd:
        x = _conv(x, sd["out_conv.weight"], sd["out_conv.bias"])
    if final_activation == "Sigmoid":
        x = torch.sigmoid(x)
    elif final_activation is not None:
        raise ValueError(final_activation)
    return x


def unet_loss_and_grads(sd, x, y, scale_factors, norm="InstanceNorm", final_activation=None, loss_fn=None):
    """Returns (prediction, loss, {param: grad}) with fp32 CPU autograd."""
    from .loss_ref import dice_loss
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    pred = unet_forward(sd, x, scale_factors, norm, final_activation)
    loss = (loss_fn or dice_loss)(pred, y)
    loss.backward()
    return pred.detach(), loss.detach(), {k: v.grad for k, v in sd.items()}


class DecisionTap:
    """Context manager around unet_forward: records the input of every ReLU (`pre`, forward order) and of every max-pool
    (`pool_in`), and -- given `force = {"relu": [bool masks], "pool": [flat arg-max indices as max_pool*_with_indices
    returns them]}` -- replaces the network's own decisions by those: ReLU becomes `x * mask`, the pooling a gather.
    With the decisions of another arithmetic forced, what is left of the distance between the two gradients is rounding,
    not near-ties resolved differently (scripts/flip_census.py, tests/test_gpu_unet.py)."""

    def __init__(self, force=None):
        self.pre, self.pool_in, self.force, self._i, self._j = [], [], force, 0, 0

    def _relu(self, t):
        self.pre.append(t.detach())
        if self.force is None:
            return torch.relu(t)
        m = self.force["relu"][self._i]
        self._i += 1
        return t * m.to(t.dtype)

    def _pool(self, t, f):
        self.pool_in.append(t.detach())
        if self.force is None:
            return torch.max_pool3d(t, f) if t.dim() == 5 else torch.max_pool2d(t, f)
        idx = self.force["pool"][self._j]
        self._j += 1
        return torch.gather(t.flatten(2), 2, idx.flatten(2)).view(idx.shape)

    def __enter__(self):
        self._orig = (F.relu, F.max_pool2d, F.max_pool3d)
        F.relu, F.max_pool2d, F.max_pool3d = self._relu, self._pool, self._pool
        return self

    def __exit__(self, *exc):
        F.relu, F.max_pool2d, F.max_pool3d = self._orig
        return False
