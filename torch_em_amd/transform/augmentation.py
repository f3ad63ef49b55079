"""On-device augmentations for the MI355X path.

Mirrors `torch_em.transform.augmentation` (reference transform/augmentation.py): `RandomElasticDeformationStacked`
(:11-88), `RandomElasticDeformation` (:91-151), `KorniaAugmentationPipeline` (:156-228), `AUGMENTATIONS` /
`DEFAULT_*_AUGMENTATIONS` (:233-265), `create_augmentation` / `get_augmentations` (:270-302).  The reference runs
kornia on the CPU inside the data-loader workers; here the pipeline runs on the batch that is already in HBM, as
hand-written HIP kernels (csrc/augment.hip): consecutive flips are fused into ONE pass per tensor, the elastic
deformation is one small blur of the noise field plus one warp pass.  kornia itself is not a dependency; the flip
classes carry kornia's names and (p, same_on_batch) arguments, random decisions come from the torch CPU generator
(flips) and the global numpy generator (elastic noise, exactly as the reference :52-55).

Parameter replay: like the reference (:212-220) the first tensor draws the parameters and every further tensor
(labels) replays them, with NEAREST interpolation for non-float inputs; everything is returned as `dtype`.
Besides the reference's default pipelines and its two elastic classes, the 3-D affine / rotation augmentations it lists
(`RandomAffine3D`, `RandomRotation3D`, :235,240) run as one trilinear / nearest warp pass per tensor (`tem_affine_warp3d`);
the 2-D `RandomAffine` / `RandomRotation` (:234,239; incl. shear) use the same kernel with a depth of one.
"""
from typing import List, Sequence, Tuple, Union

import numpy as np
import torch

from .. import _lib, ops


def _as_batch(t: torch.Tensor, spatial: int) -> torch.Tensor:
    """kornia's shape promotion: ([C,]*S) -> (1,C,*S); batched tensors pass through."""
    while t.dim() < spatial + 2:
        t = t[None]
    if t.dim() != spatial + 2:
        raise ValueError(f"expected a tensor with {spatial}..{spatial + 2} dimensions, got {t.dim()}")
    return t


class _RandomFlip(torch.nn.Module):
    """Flip of one spatial axis with probability p per sample (kornia RandomHorizontalFlip[3D] and friends)."""
    axis = -1
    spatial = 3

    def __init__(self, p: float = 0.5, same_on_batch: bool = False, keepdim: bool = False):
        super().__init__()
        self.p, self.same_on_batch, self.keepdim = p, same_on_batch, keepdim
        self._params = None
        self.flags = {}

    def generate_parameters(self, batch_shape):
        n = batch_shape[0]
        if self.same_on_batch:
            draw = (torch.rand(1) < self.p).expand(n)
        else:
            draw = torch.rand(n) < self.p
        return {"batch_prob": draw.clone()}

    def forward(self, input: torch.Tensor, params=None) -> torch.Tensor:
        return KorniaAugmentationPipeline(self, dtype=input.dtype)._run([input], [params])[0]


class RandomHorizontalFlip3D(_RandomFlip):
    axis, spatial = -1, 3


class RandomVerticalFlip3D(_RandomFlip):
    axis, spatial = -2, 3


class RandomDepthicalFlip3D(_RandomFlip):
    axis, spatial = -3, 3


class RandomHorizontalFlip(_RandomFlip):
    axis, spatial = -1, 2


class RandomVerticalFlip(_RandomFlip):
    axis, spatial = -2, 2


def _pair(v, name):
    if isinstance(v, (int, float)):
        return (-float(v), float(v))
    v = tuple(float(a) for a in v)
    if len(v) != 2:
        raise ValueError(f"{name}: expected a number or a (min, max) pair, got {v}")
    return v


class RandomAffine3D(torch.nn.Module):
    """Random 3-D rotation about the volume centre + isotropic scaling, one pass on the device (kornia's
    `RandomAffine3D(degrees, translate=None, scale=None, ...)`; the reference configures degrees=(90,90,90),
    scale=(0.0,1.1), transform/augmentation.py:235).  `degrees`: one number d (each angle in [-d, d]), a triple of such
    numbers, or a triple of (min, max) pairs, for yaw (about z = the D axis), pitch (about y) and roll (about x);
    `scale`: (min, max) of the isotropic zoom or None; `translate`: fractions of (W, H, D) or None.  Parameters are drawn
    per sample from the torch CPU generator (`same_on_batch=False`) and replayed on the label tensors with nearest
    interpolation.  kornia is not in this image: the composition order (R = Rz(yaw) Ry(pitch) Rx(roll), centre (size-1)/2,
    zeros outside) is this build's definition -- parity unpinned."""
    spatial = 3
    min_scale = 1e-3  # a drawn scale of 0 (allowed by the reference's (0.0, 1.1)) has no inverse map

    def __init__(self, degrees, translate=None, scale=None, shears=None, resample="bilinear", same_on_batch: bool = False,
                 align_corners: bool = False, p: float = 0.5, keepdim: bool = False):
        super().__init__()
        if shears is not None:
            raise NotImplementedError("RandomAffine3D: shears have no MI355X kernel parameters")
        if isinstance(degrees, (int, float)):
            degrees = (degrees,) * 3
        self.degrees = tuple(_pair(d, "degrees") for d in degrees)
        if len(self.degrees) != 3:
            raise ValueError("degrees: expected 3 entries (yaw, pitch, roll)")
        self.scale = None if scale is None else _pair(scale, "scale")
        self.translate = None if translate is None else tuple(float(t) for t in translate)
        self.p, self.same_on_batch, self.keepdim = p, same_on_batch, keepdim
        self.flags = dict(interpolation=resample, align_corners=align_corners)
        self._params = None

    def generate_parameters(self, batch_shape):
        n = batch_shape[0]
        m = 1 if self.same_on_batch else n

        def draw(lo, hi):
            return (torch.rand(m, dtype=torch.float64) * (hi - lo) + lo).expand(n).clone()
        apply = (torch.rand(m) < self.p).expand(n).clone()
        angles = torch.stack([draw(*d) for d in self.degrees], 1)
        scale = draw(*self.scale) if self.scale is not None else torch.ones(n, dtype=torch.float64)
        if self.translate is not None:
            dims = (batch_shape[-1], batch_shape[-2], batch_shape[-3])
            trans = torch.stack([draw(-t * s, t * s) for t, s in zip(self.translate, dims)], 1)
        else:
            trans = torch.zeros(n, 3, dtype=torch.float64)
        return {"batch_prob": apply, "angles": angles, "scale": scale, "translations": trans}

    @classmethod
    def inverse_matrices(cls, params, shape) -> torch.Tensor:
        """[N, 3, 4] float64: output voxel (x, y, z, 1) -> source position, for the drawn parameters."""
        D, H, W = shape[-3:]
        n = params["angles"].shape[0]
        out = torch.zeros(n, 3, 4, dtype=torch.float64)
        c = torch.tensor([(W - 1) / 2.0, (H - 1) / 2.0, (D - 1) / 2.0], dtype=torch.float64)
        for i in range(n):
            if not bool(params["batch_prob"][i]):
                out[i, :, :3] = torch.eye(3, dtype=torch.float64)
                continue
            yaw, pitch, roll = [float(a) * np.pi / 180.0 for a in params["angles"][i]]
            cz, sz, cy, sy, cx, sx = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
            rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=torch.float64)
            ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=torch.float64)
            rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=torch.float64)
            fwd = (rz @ ry @ rx) * max(float(params["scale"][i]), cls.min_scale)   # dst = fwd (src - c) + c + t
            inv = torch.linalg.inv(fwd)
            out[i, :, :3] = inv
            out[i, :, 3] = c - inv @ (c + params["translations"][i])
        return out

    def forward(self, input: torch.Tensor, params=None) -> torch.Tensor:
        return KorniaAugmentationPipeline(self, dtype=input.dtype)._run([input], [params])[0]


class RandomRotation3D(RandomAffine3D):
    """kornia `RandomRotation3D(degrees)` (reference :240: degrees=(90,90,90)): RandomAffine3D without scale / shift."""

    def __init__(self, degrees, resample="bilinear", same_on_batch: bool = False, align_corners: bool = False,
                 p: float = 0.5, keepdim: bool = False):
        super().__init__(degrees, None, None, None, resample, same_on_batch, align_corners, p, keepdim)


class RandomAffine(RandomAffine3D):
    """kornia's 2-D `RandomAffine(degrees, translate=None, scale=None, shear=None, ...)` (reference AUGMENTATIONS :234:
    degrees=90, scale=(0.9, 1.1)) on the 3-D warp kernel with a depth of one: rotation about the image centre by an
    angle in [-d, d] (or (min, max)), isotropic zoom, translation fractions of (W, H), x / y shear angles in degrees
    (`shear`: one number s -> x shear in [-s, s]; (a, b) -> x shear in [a, b]; (a, b, c, d) -> x in [a, b], y in [c, d]).
    The forward map is dst = R * Shear * scale * (src - c) + c + t; kornia is not in this image -- composition order and
    centre ((size - 1) / 2) are this build's definition: parity unpinned."""
    spatial = 2

    def __init__(self, degrees, translate=None, scale=None, shear=None, resample="bilinear", same_on_batch: bool = False,
                 align_corners: bool = False, padding_mode="zeros", p: float = 0.5, keepdim: bool = False):
        if padding_mode not in ("zeros", 0):
            raise NotImplementedError("RandomAffine: the MI355X warp kernel pads with zeros")
        torch.nn.Module.__init__(self)
        self.degrees = (_pair(degrees, "degrees"), (0.0, 0.0), (0.0, 0.0))     # yaw only
        self.scale = None if scale is None else _pair(scale[:2] if len(scale) > 2 else scale, "scale")
        self.translate = None if translate is None else (float(translate[0]), float(translate[1]), 0.0)
        if shear is None:
            self.shear = None
        elif isinstance(shear, (int, float)):
            self.shear = ((-float(shear), float(shear)), (0.0, 0.0))
        elif len(shear) == 2:
            self.shear = ((float(shear[0]), float(shear[1])), (0.0, 0.0))
        elif len(shear) == 4:
            self.shear = ((float(shear[0]), float(shear[1])), (float(shear[2]), float(shear[3])))
        else:
            raise ValueError("shear: a number, (min, max) or (xmin, xmax, ymin, ymax)")
        self.p, self.same_on_batch, self.keepdim = p, same_on_batch, keepdim
        self.flags = dict(interpolation=resample, align_corners=align_corners)
        self._params = None

    def generate_parameters(self, batch_shape):
        shape3 = tuple(batch_shape[:-2]) + (1,) + tuple(batch_shape[-2:])
        params = super().generate_parameters(shape3)
        n = batch_shape[0]
        m = 1 if self.same_on_batch else n
        if self.shear is not None:
            draw = lambda lo, hi: (torch.rand(m, dtype=torch.float64) * (hi - lo) + lo).expand(n).clone()  # noqa: E731
            params["shear"] = torch.stack([draw(*self.shear[0]), draw(*self.shear[1])], 1)
        return params

    @classmethod
    def inverse_matrices(cls, params, shape) -> torch.Tensor:
        shape3 = tuple(shape[:-2]) + (1,) + tuple(shape[-2:])
        out = RandomAffine3D.inverse_matrices(params, shape3)
        out[:, 2] = torch.tensor([0.0, 0.0, 1.0, 0.0], dtype=torch.float64)     # the one z-plane maps onto itself
        sh = params.get("shear")
        if sh is None:
            return out
        H, W = shape[-2:]
        c = torch.tensor([(W - 1) / 2.0, (H - 1) / 2.0, 0.0], dtype=torch.float64)
        for i in range(out.shape[0]):
            if not bool(params["batch_prob"][i]):
                continue
            sx, sy = [np.tan(float(a) * np.pi / 180.0) for a in sh[i]]
            shear = torch.tensor([[1.0, sx, 0.0], [sy, 1.0, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float64)
            # out = inverse of (dst = F (src - c) + c + t) with F = R * scale; the sheared map has F' = F * Shear
            inv = torch.linalg.inv(shear) @ out[i, :, :3]
            out[i, :, :3] = inv
            out[i, :, 3] = c - inv @ (c + params["translations"][i])
        return out


class RandomRotation(RandomAffine):
    """kornia's 2-D `RandomRotation(degrees)` (reference AUGMENTATIONS :239: degrees=90): RandomAffine without zoom / shift."""

    def __init__(self, degrees, resample="bilinear", same_on_batch: bool = False, align_corners: bool = True,
                 p: float = 0.5, keepdim: bool = False):
        super().__init__(degrees, None, None, None, resample, same_on_batch, align_corners, "zeros", p, keepdim)


def _gauss1d(ksize: int, sigma: float) -> torch.Tensor:
    x = torch.arange(ksize, dtype=torch.float32) - ksize // 2
    g = torch.exp(-x.pow(2.0) / (2 * float(sigma) ** 2))
    return g / g.sum()


def _resize_order3(field: np.ndarray, shape) -> np.ndarray:
    """skimage.transform.resize(order=3) of the control grid (reference :56-58): identity for spacing 1, else the
    cubic-spline zoom skimage delegates to (scipy.ndimage.zoom, mode='reflect', grid_mode=True)."""
    if tuple(field.shape) == tuple(shape):
        return field
    from scipy import ndimage
    zoom = [s / f for s, f in zip(shape, field.shape)]
    return ndimage.zoom(field, zoom, order=3, mode="reflect", grid_mode=True)


class _ElasticBase(torch.nn.Module):
    spatial = 2
    kernel_size = 63  # kornia elastic_transform2d default

    def __init__(self, control_point_spacing: Union[int, Sequence[int]] = 1, sigma: Tuple[float, float] = (32.0, 32.0),
                 alpha: Tuple[float, float] = (4.0, 4.0), interpolation="bilinear", p: float = 0.5,
                 keepdim: bool = False, same_on_batch: bool = True):
        super().__init__()
        if isinstance(control_point_spacing, int):
            self.control_point_spacing = [control_point_spacing] * 2
        else:
            self.control_point_spacing = control_point_spacing
        assert len(self.control_point_spacing) == 2
        # the reference overrides __call__, so kornia's `p` gate never runs: the deformation is always applied
        self.p, self.same_on_batch, self.keepdim = p, same_on_batch, keepdim
        self.flags = dict(interpolation=interpolation, sigma=sigma, alpha=alpha)
        self._params = None

    def generate_parameters(self, batch_shape):
        shape = tuple(batch_shape[-2:])
        control_shape = tuple(sh // sp for sh, sp in zip(shape, self.control_point_spacing))
        fields = [np.random.uniform(-1, 1, control_shape), np.random.uniform(-1, 1, control_shape)]
        fields = [_resize_order3(df, shape)[None] for df in fields]
        noise = np.concatenate(fields, axis=0)[None].astype("float32")
        return {"noise": torch.from_numpy(noise)}

    def displacement(self, noise: torch.Tensor, device) -> torch.Tensor:
        """[1,2,H,W] noise -> [2,H,W] displacement field on the device (normalised grid units)."""
        lib = _lib.load()
        noise = noise.to(device=device, dtype=torch.float32).reshape(2, *noise.shape[-2:]).contiguous()
        H, W = noise.shape[-2:]
        sigma, alpha = self.flags["sigma"], self.flags["alpha"]
        g = torch.stack([_gauss1d(self.kernel_size, sigma[0]), _gauss1d(self.kernel_size, sigma[1])]).to(device)
        disp = torch.empty_like(noise)
        _lib.check(lib.tem_elastic_field(ops._p(noise), ops._p(g), self.kernel_size, H, W, float(alpha[0]),
                                         float(alpha[1]), ops._p(disp), ops._stream(noise)), "tem_elastic_field")
        return disp

    def forward(self, input: torch.Tensor, params=None) -> torch.Tensor:
        return KorniaAugmentationPipeline(self, dtype=input.dtype)._run([input], [params])[0]


class RandomElasticDeformationStacked(_ElasticBase):
    """The same random 2-D elastic deformation applied to every plane of a 3-D volume (reference :11-88)."""
    spatial = 3


class RandomElasticDeformation(_ElasticBase):
    """Random 2-D elastic deformation (reference :91-151; the reference calls the interpolation flag `resample`)."""
    spatial = 2

    def __init__(self, control_point_spacing=1, sigma=(32.0, 32.0), alpha=(4.0, 4.0), resample="bilinear", p=0.5,
                 keepdim=False, same_on_batch=True):
        super().__init__(control_point_spacing, sigma, alpha, resample, p, keepdim, same_on_batch)


class KorniaAugmentationPipeline(torch.nn.Module):
    """Applies a list of augmentations to several tensors with shared random parameters (reference :156-228)."""
    interpolatable_torch_types = [torch.float16, torch.float32, torch.float64]
    interpolatable_numpy_types = [np.dtype("float32"), np.dtype("float64")]

    def __init__(self, *kornia_augmentations, dtype: Union[str, torch.dtype] = torch.float32):
        super().__init__()
        for aug in kornia_augmentations:
            if not isinstance(aug, (_RandomFlip, _ElasticBase, RandomAffine3D)):
                raise NotImplementedError(f"{type(aug).__name__} has no MI355X kernel (flips, elastic deformations and "
                                          "2-D / 3-D affine / rotation warps do)")
        self.augmentations = torch.nn.ModuleList(kornia_augmentations)
        self.dtype = dtype
        self.halo = self.compute_halo()

    def compute_halo(self):
        """The fixed 32-pixel hint the reference gives for the pure rotations only (:174-183; unused by its datasets)."""
        halo = None
        for aug in self.augmentations:
            if isinstance(aug, RandomRotation):
                halo = [32, 32]
            if isinstance(aug, RandomRotation3D):
                halo = [32, 32, 32]
        return halo

    def is_interpolatable(self, tensor):
        if torch.is_tensor(tensor):
            return tensor.dtype in self.interpolatable_torch_types
        return tensor.dtype in self.interpolatable_numpy_types

    def _run(self, tensors, params_in=None):
        interp = [self.is_interpolatable(t) for t in tensors]
        dtype = getattr(torch, self.dtype) if isinstance(self.dtype, str) else self.dtype
        if dtype != torch.float32:
            raise NotImplementedError("the MI355X augmentation kernels move 4-byte float32 elements")
        work = []
        for t in tensors:
            t = torch.as_tensor(t)
            if not t.is_cuda:
                raise RuntimeError("torch_em_amd augmentations run on MI355X only (got a CPU tensor); no CPU fallback")
            work.append(t.to(dtype))
        lib = _lib.load()
        augs = list(self.augmentations)
        i = 0
        while i < len(augs):
            aug = augs[i]
            spatial = aug.spatial
            work = [_as_batch(t, spatial).contiguous() for t in work]
            shape = work[0].shape
            N = shape[0]
            if isinstance(aug, _RandomFlip):
                # fuse the run of consecutive flips into one pass per tensor
                flags = torch.zeros(N, 3, dtype=torch.int32)
                while i < len(augs) and isinstance(augs[i], _RandomFlip) and augs[i].spatial == spatial:
                    a = augs[i]
                    given = params_in[i] if params_in is not None and i < len(params_in) else None
                    params = given if given is not None else a.generate_parameters(shape)
                    a._params = params
                    flags[:, 3 + a.axis] ^= params["batch_prob"].to(torch.int32)
                    i += 1
                fl = flags.to(work[0].device)
                out = []
                for t in work:
                    assert t.shape[0] == N and t.shape[2:] == shape[2:], "all tensors must share batch and spatial shape"
                    D = t.shape[2] if spatial == 3 else 1
                    dst = torch.empty_like(t)
                    _lib.check(lib.tem_flip3d(ops._p(t), ops._p(dst), ops._p(fl), N, t.shape[1], D, t.shape[-2],
                                              t.shape[-1], ops._stream(t)), "tem_flip3d")
                    out.append(dst)
                work = out
            elif isinstance(aug, RandomAffine3D):
                given = params_in[i] if params_in is not None and i < len(params_in) else None
                params = given if given is not None else aug.generate_parameters(shape)
                aug._params = params
                mats = aug.inverse_matrices(params, shape).to(torch.float32).reshape(N, 12).to(work[0].device)
                out = []
                for t, ip in zip(work, interp):
                    assert t.shape[0] == N and t.shape[2:] == shape[2:], "all tensors must share batch and spatial shape"
                    nearest = not (ip and aug.flags["interpolation"] in ("bilinear", 1))
                    dst = torch.empty_like(t)
                    depth = t.shape[2] if spatial == 3 else 1   # 2-D images: one z-plane
                    _lib.check(lib.tem_affine_warp3d(ops._p(t), ops._p(mats), ops._p(dst), N, t.shape[1], depth,
                                                     t.shape[-2], t.shape[-1], int(nearest), ops._stream(t)),
                               "tem_affine_warp3d")
                    out.append(dst)
                work = out
                i += 1
            else:
                given = params_in[i] if params_in is not None and i < len(params_in) else None
                params = given if given is not None else aug.generate_parameters(shape)
                aug._params = params
                disp = aug.displacement(params["noise"], work[0].device)
                out = []
                for t, ip in zip(work, interp):
                    H, W = t.shape[-2:]
                    assert (H, W) == tuple(disp.shape[-2:]), "all tensors must share the spatial shape"
                    nearest = not (ip and aug.flags["interpolation"] in ("bilinear", 1))
                    dst = torch.empty_like(t)
                    _lib.check(lib.tem_elastic_warp2d(ops._p(t), ops._p(disp), ops._p(dst), t.numel() // (H * W), H, W,
                                                      int(nearest), ops._stream(t)), "tem_elastic_warp2d")
                    out.append(dst)
                work = out
                i += 1
        return work

    def forward(self, *tensors) -> List[torch.Tensor]:
        return self._run(list(tensors))


AUGMENTATIONS = {
    "RandomAffine": {"degrees": 90, "scale": (0.9, 1.1)},
    "RandomRotation": {"degrees": 90},
    "RandomAffine3D": {"degrees": (90, 90, 90), "scale": (0.0, 1.1)},
    "RandomRotation3D": {"degrees": (90, 90, 90)},
    "RandomDepthicalFlip3D": {},
    "RandomHorizontalFlip": {},
    "RandomHorizontalFlip3D": {},
    "RandomVerticalFlip": {},
    "RandomVerticalFlip3D": {},
    "RandomElasticDeformation": {"alpha": [5, 5], "sigma": [30, 30]},
    "RandomElasticDeformationStacked": {"alpha": [5, 5], "sigma": [30, 30]},
}
DEFAULT_2D_AUGMENTATIONS = ["RandomHorizontalFlip", "RandomVerticalFlip"]
DEFAULT_3D_AUGMENTATIONS = ["RandomHorizontalFlip3D", "RandomVerticalFlip3D", "RandomDepthicalFlip3D"]
DEFAULT_ANISOTROPIC_AUGMENTATIONS = ["RandomHorizontalFlip3D", "RandomVerticalFlip3D", "RandomDepthicalFlip3D"]


def create_augmentation(trafo):
    assert trafo in AUGMENTATIONS and trafo in globals(), f"Transformation {trafo} not defined"
    return globals()[trafo](**AUGMENTATIONS[trafo])


def get_augmentations(ndim: Union[int, str] = 2, transforms=None, dtype: Union[str, torch.dtype] = torch.float32):
    """Augmentation pipeline for 2-D, 3-D or anisotropic data (reference :279-302)."""
    if transforms is None:
        assert ndim in (2, 3, "anisotropic"), f"Expect ndim to be one of (2, 3, 'anisotropic'), got {ndim}"
        transforms = {2: DEFAULT_2D_AUGMENTATIONS, 3: DEFAULT_3D_AUGMENTATIONS}.get(ndim, DEFAULT_ANISOTROPIC_AUGMENTATIONS)
    transforms = [create_augmentation(t) if isinstance(t, str) else t for t in transforms]
    return KorniaAugmentationPipeline(*transforms, dtype=dtype)
