"""Coordinate MSE loss on the HIP path (reference model/loss.py:6-66)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .easydict import EasyDict as edict


class MSELossComputer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        tr = config["training"] if isinstance(config, dict) and "training" in config else getattr(config, "training", None)
        if tr is None or ("coord_mse_loss_weight" not in tr):
            raise ValueError("Configuration must have 'config.training.coord_mse_loss_weight' defined.")
        self._weight = float(tr["coord_mse_loss_weight"])

    def forward(self, coords_pred: torch.Tensor, coords_target: torch.Tensor):
        if not (coords_pred.ndim == 4 and coords_target.ndim == 4 and coords_pred.shape == coords_target.shape):
            raise ValueError(
                f"Shape mismatch or invalid shape for coordinate MSE. Expected both tensors of shape (B, T, N, C). "
                f"Got pred: {coords_pred.shape}, target: {coords_target.shape}")
        metrics = edict()
        if self._weight > 0.0:
            mse = ops.mse(coords_pred, coords_target, 1.0)
            metrics.coord_mse_loss = mse
            metrics.loss = mse * self._weight
        else:
            z = torch.zeros((), device=coords_pred.device)
            metrics.coord_mse_loss = z
            metrics.loss = z.clone()
        return metrics
