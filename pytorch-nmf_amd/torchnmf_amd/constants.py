"""Numerical constants shared with the reference (torchnmf/constants.py:3)."""
import torch

eps = torch.finfo(torch.float32).eps  # 2**-23
