"""AverageDistortionLoss (reference sympa/losses.py:4-19): sum |(d_manifold / d_graph)^2 - 1|."""
import torch


class AverageDistortionLoss:
    def calculate_loss(self, graph_distances, manifold_distances):
        return torch.abs(torch.pow(manifold_distances / graph_distances, 2) - 1).sum()
