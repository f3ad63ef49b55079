"""GPU parity of the on-device augmentations (csrc/augment.hip) against oracle/augment_ref.py.
Flips are bit-exact.  Elastic deformation (floating point): displacement field rtol 1e-5 of its max, bilinear warp
atol 1e-4 on a unit-variance noise image, nearest warp identical except at rounding ties (< 0.1 % of the pixels may differ)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_flips_fused_and_replayed_on_labels():
    from oracle import augment_ref
    from torch_em_amd.transform import get_augmentations
    torch.manual_seed(0)
    x = torch.randn(5, 2, 6, 10, 12)
    lbl = torch.randint(0, 9, (5, 1, 6, 10, 12))
    pipe = get_augmentations(3)
    xt, lt = pipe(x.cuda(), lbl.cuda())
    assert xt.dtype == lt.dtype == torch.float32 and xt.shape == x.shape
    fx, fy, fz = [a._params["batch_prob"] for a in pipe.augmentations]  # H, V, D flips in this order
    assert 0 < int(fx.sum() + fy.sum() + fz.sum()) < 15
    for n in range(5):
        assert torch.equal(xt[n].cpu(), augment_ref.flip(x[n], fz[n], fy[n], fx[n]))
        assert torch.equal(lt[n].cpu(), augment_ref.flip(lbl[n].float(), fz[n], fy[n], fx[n]))
    # replaying the recorded parameters undoes the flips (involution)
    back = pipe._run([xt], [a._params for a in pipe.augmentations])[0]
    assert torch.equal(back.cpu(), x)


def test_default_2d_pipeline_shape_like_reference_test():
    # reference test/transform/test_augmentations.py:6-11
    from torch_em_amd.transform import get_augmentations
    x = torch.rand(12, 64, 64)
    xt = get_augmentations()(x.cuda())[0]
    assert xt.shape == (1,) + x.shape
    assert torch.equal(torch.sort(xt.flatten())[0].cpu(), torch.sort(x.flatten())[0])


@pytest.mark.parametrize("shape,spacing", [((2, 1, 5, 48, 40), 1), ((1, 2, 3, 64, 64), 4)])
def test_elastic_stacked_vs_oracle(shape, spacing):
    from oracle import augment_ref
    from torch_em_amd.transform import KorniaAugmentationPipeline, RandomElasticDeformationStacked
    torch.manual_seed(1)
    np.random.seed(1)
    x = torch.randn(shape)
    lbl = torch.randint(0, 7, shape[:1] + (1,) + shape[2:])
    aug = RandomElasticDeformationStacked(control_point_spacing=spacing, sigma=(8.0, 6.0), alpha=(9.0, 7.0))
    xt, lt = KorniaAugmentationPipeline(aug)(x.cuda(), lbl.cuda())
    noise = aug._params["noise"]
    assert noise.shape == (1, 2) + shape[-2:]
    want_disp = augment_ref.elastic_field(noise[0], (8.0, 6.0), (9.0, 7.0))
    got_disp = aug.displacement(noise, "cuda").cpu()
    assert float((got_disp - want_disp).abs().max()) < 1e-5 * float(want_disp.abs().max())
    assert float(want_disp.abs().max()) * shape[-1] / 2 > 1.0  # the field moves pixels by more than one pixel
    # the warp is checked on the SAME field (the field itself was compared above): sampling a random image turns a
    # 1e-6 coordinate difference into a 1e-5 value difference, so the two stages are pinned separately
    want_x = augment_ref.elastic_warp(x.reshape(-1, *shape[-2:]), got_disp).reshape(shape)
    assert float((xt.cpu() - want_x).abs().max()) < 1e-4  # fp32 grid coordinates (1 ulp * W/2 px) x image gradient (~4/px)
    want_l = augment_ref.elastic_warp(lbl.float().reshape(-1, *shape[-2:]), got_disp, nearest=True).reshape(lbl.shape)
    assert float((lt.cpu() != want_l).float().mean()) < 1e-3
    assert set(torch.unique(lt).tolist()) <= set(range(7))  # nearest: labels stay labels


def test_elastic_2d_runs_like_reference_test():
    # reference test/transform/test_augmentations.py:14-20
    from torch_em_amd.transform import KorniaAugmentationPipeline, RandomElasticDeformation
    deform = RandomElasticDeformation(alpha=(1.0, 1.0), p=1)
    x = torch.rand(1, 1, 64, 64)
    xt = KorniaAugmentationPipeline(deform)(x.cuda())[0]
    assert xt.shape == x.shape and torch.isfinite(xt).all()


def test_affine3d_warp_vs_oracle_and_replay_on_labels():
    """RandomAffine3D / RandomRotation3D (reference transform/augmentation.py:235,240): one trilinear pass for the image,
    the same drawn parameters replayed with nearest interpolation on the labels; against the torch-CPU restatement."""
    from oracle import augment_ref
    from torch_em_amd.transform import KorniaAugmentationPipeline, RandomAffine3D, RandomRotation3D, get_augmentations
    torch.manual_seed(3)
    x = torch.randn(3, 2, 12, 20, 16)
    lbl = torch.randint(0, 7, (3, 1, 12, 20, 16))
    aug = RandomAffine3D((90, 90, 90), scale=(0.6, 1.1), p=1.0)
    xt, lt = KorniaAugmentationPipeline(aug)(x.cuda(), lbl.cuda())
    inv = RandomAffine3D.inverse_matrices(aug._params, x.shape)
    assert float((inv[:, :, :3] - torch.eye(3, dtype=torch.float64)).abs().max()) > 0.1   # not the identity
    want = augment_ref.affine_warp3d(x, inv)
    assert float((xt.cpu() - want).abs().max()) < 2e-4      # fp32 coordinates x image gradient, like the elastic warp
    want_l = augment_ref.affine_warp3d(lbl.float(), inv, nearest=True)
    assert float((lt.cpu() != want_l).float().mean()) < 5e-3      # rounding ties of the nearest neighbour
    assert set(torch.unique(lt).tolist()) <= set(range(7))
    # p = 0: identity, bit for bit; rotation by 90 degrees about z maps the volume onto itself exactly (W == H)
    same = KorniaAugmentationPipeline(RandomRotation3D((90, 90, 90), p=0.0))(x.cuda())[0]
    assert torch.equal(same.cpu(), x)
    sq = torch.randn(1, 1, 4, 9, 9)
    rot = RandomRotation3D(((90, 90), (0, 0), (0, 0)), p=1.0)
    got = KorniaAugmentationPipeline(rot)(sq.cuda())[0].cpu()
    assert float((got - torch.rot90(sq, 1, (-2, -1))).abs().max()) < 1e-5 or \
        float((got - torch.rot90(sq, -1, (-2, -1))).abs().max()) < 1e-5
    # by name, as the reference's get_augmentations does
    pipe = get_augmentations(3, transforms=["RandomRotation3D", "RandomHorizontalFlip3D"])
    assert pipe(x.cuda())[0].shape == x.shape and pipe.halo == [32, 32, 32]


def test_affine2d_warp_with_shear_vs_oracle():
    """2-D RandomAffine / RandomRotation (reference transform/augmentation.py:234,239) on the 3-D warp kernel with one
    z-plane: rotation + zoom + shift + x / y shear for the image, nearest replay for the labels."""
    from oracle import augment_ref
    from torch_em_amd.transform import KorniaAugmentationPipeline, RandomAffine, RandomRotation, get_augmentations
    torch.manual_seed(5)
    x = torch.randn(4, 3, 40, 28)
    lbl = torch.randint(0, 5, (4, 1, 40, 28))
    aug = RandomAffine(70, translate=(0.1, 0.2), scale=(0.8, 1.2), shear=(-20, 20, -10, 10), p=1.0)
    xt, lt = KorniaAugmentationPipeline(aug)(x.cuda(), lbl.cuda())
    assert xt.shape == x.shape and lt.shape == lbl.shape
    inv = RandomAffine.inverse_matrices(aug._params, x.shape)
    assert float(inv[:, 2, :].abs().sub(torch.tensor([0, 0, 1.0, 0], dtype=torch.float64)).abs().max()) < 1e-12  # z stays
    assert float(aug._params["shear"].abs().max()) > 1.0
    want = augment_ref.affine_warp3d(x[:, :, None], inv)[:, :, 0]
    assert float((xt.cpu() - want).abs().max()) < 2e-4
    want_l = augment_ref.affine_warp3d(lbl.float()[:, :, None], inv, nearest=True)[:, :, 0]
    assert float((lt.cpu() != want_l).float().mean()) < 5e-3
    # shear only: the centre row / column stays, x moves with y
    sh = RandomAffine(0, shear=(45, 45), p=1.0)
    img = torch.zeros(1, 1, 9, 9)
    img[0, 0, :, 4] = 1.0                              # a vertical line through the centre
    got = KorniaAugmentationPipeline(sh)(img.cuda())[0].cpu()[0, 0]
    assert abs(float(got[4, 4]) - 1.0) < 1e-5 and abs(float(got[4].sum()) - 1.0) < 1e-5
    cols = got.argmax(1)
    assert (cols[1:] - cols[:-1]).abs().eq(1).all()   # the line leans by one column per row
    # the plain rotation: 90 degrees on a square image, identity for p = 0, the reference's halo hint
    sq = torch.randn(2, 1, 11, 11)
    rot = KorniaAugmentationPipeline(RandomRotation((90, 90), p=1.0))
    got = rot(sq.cuda())[0].cpu()
    assert float((got - torch.rot90(sq, 1, (-2, -1))).abs().max()) < 1e-5 or \
        float((got - torch.rot90(sq, -1, (-2, -1))).abs().max()) < 1e-5
    assert rot.halo == [32, 32]
    assert torch.equal(KorniaAugmentationPipeline(RandomRotation(90, p=0.0))(sq.cuda())[0].cpu(), sq)
    pipe = get_augmentations(2, transforms=["RandomAffine", "RandomVerticalFlip"])
    assert pipe.halo is None and pipe(x.cuda())[0].shape == x.shape
