"""In-memory datasets for the MI355X path (harness; reference torch_em/data/tensor_dataset.py)."""
from .tensor_dataset import TensorDataset
