"""Trainer runtime of the MI355X path (reference torch_em/trainer/__init__.py)."""
from .default_trainer import DefaultTrainer
from .spoco_trainer import SPOCOTrainer
