"""CPU restatement of the augmentation arithmetic (TEST INFRASTRUCTURE, see oracle/__init__.py).

PARITY UNPINNED for the elastic deformation: the reference delegates it to kornia (setup.py:12, unpinned), which is not
installed in this image, so no golden vectors could be generated.  This file restates kornia's published
`elastic_transform2d` / `get_gaussian_kernel2d` / `filter2d(border_type="constant")` / `create_meshgrid` with torch-CPU
library ops (F.conv2d, F.grid_sample), following the reference's call sites:
  RandomElasticDeformationStacked.__call__   /root/reference/torch_em/transform/augmentation.py:63-88
  RandomElasticDeformation.__call__          /root/reference/torch_em/transform/augmentation.py:134-151
  KorniaAugmentationPipeline.forward         /root/reference/torch_em/transform/augmentation.py:203-223
Flips (kornia RandomHorizontalFlip3D / RandomVerticalFlip3D / RandomDepthicalFlip3D = flip of W / H / D) are exact.
"""
import torch
import torch.nn.functional as F


def gaussian_kernel1d(ksize, sigma):
    x = torch.arange(ksize, dtype=torch.float32) - ksize // 2
    if ksize % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2.0) / (2 * sigma ** 2))
    return g / g.sum()


def elastic_field(noise, sigma=(32.0, 32.0), alpha=(4.0, 4.0), ksize=63):
    """noise [2,H,W] -> displacement [2,H,W] in normalised grid units."""
    gx, gy = gaussian_kernel1d(ksize, sigma[0]), gaussian_kernel1d(ksize, sigma[1])
    kern_x = torch.outer(gx, gx)[None, None]
    kern_y = torch.outer(gy, gy)[None, None]
    pad = ksize // 2
    dx = F.conv2d(noise[None, :1], kern_y, padding=pad)[0, 0] * alpha[0]
    dy = F.conv2d(noise[None, 1:], kern_x, padding=pad)[0, 0] * alpha[1]
    return torch.stack([dx, dy])


def elastic_warp(planes, disp, nearest=False):
    """planes [P,H,W] float, disp [2,H,W] -> warped [P,H,W]."""
    p, h, w = planes.shape
    xs = torch.linspace(-1, 1, w) if w > 1 else torch.tensor([-1.0])
    ys = torch.linspace(-1, 1, h) if h > 1 else torch.tensor([-1.0])
    grid = torch.stack(torch.meshgrid(xs, ys, indexing="xy"), dim=-1)  # [H,W,2] (x, y)
    grid = (grid + disp.permute(1, 2, 0)).clamp(-1, 1)
    return F.grid_sample(planes[None], grid[None], mode="nearest" if nearest else "bilinear",
                         padding_mode="reflection", align_corners=False)[0]


def flip(x, fz, fy, fx):
    dims = [d for d, f in zip((-3, -2, -1), (fz, fy, fx)) if f]
    return torch.flip(x, dims) if dims else x.clone()


def affine_warp3d(x, inv_mats, nearest=False):
    """x [N,C,D,H,W], inv_mats [N,3,4] (output voxel (x,y,z,1) -> source position, voxel units) -> warped x.
    PARITY UNPINNED (kornia's warp_affine3d is absent).  Independent of the HIP kernel's arithmetic: the voxel-space map
    is re-expressed in normalised coordinates and resampled by F.affine_grid + F.grid_sample(align_corners=True,
    padding_mode="zeros"), for which x_norm = 2 x / (size - 1) - 1."""
    n, c, d, h, w = x.shape
    size = torch.tensor([w, h, d], dtype=torch.float64)
    to_norm = torch.diag(torch.cat([2.0 / (size - 1).clamp(min=1), torch.ones(1, dtype=torch.float64)]))
    to_norm[:3, 3] = -1.0
    from_norm = torch.linalg.inv(to_norm)
    theta = []
    for i in range(n):
        a = torch.eye(4, dtype=torch.float64)
        a[:3] = inv_mats[i].double()
        theta.append((to_norm @ a @ from_norm)[:3])
    grid = F.affine_grid(torch.stack(theta).float(), [n, c, d, h, w], align_corners=True)
    return F.grid_sample(x, grid, mode="nearest" if nearest else "bilinear", padding_mode="zeros", align_corners=True)
