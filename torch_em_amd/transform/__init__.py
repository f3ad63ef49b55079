"""Transforms of the MI355X path that belong to the hot path (reference torch_em/transform/)."""
from .label import AffinityTransform, BoundaryTransform, labels_to_binary
from .raw import standardize
from .augmentation import (KorniaAugmentationPipeline, RandomAffine, RandomAffine3D, RandomElasticDeformation, RandomRotation,
                           RandomElasticDeformationStacked, RandomRotation3D, get_augmentations)
