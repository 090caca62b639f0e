"""Restated subset of torch_geometric.nn.models (PyG 2.0.1) -- TEST INFRASTRUCTURE ONLY."""
import torch


class InnerProductDecoder(torch.nn.Module):
    """sigma(z[i] . z[j]) -- only referenced by `MyGAE` (/root/reference/src/layers.py:258)."""

    def forward(self, z, edge_index, sigmoid=True):
        value = (z[edge_index[0]] * z[edge_index[1]]).sum(dim=1)
        return torch.sigmoid(value) if sigmoid else value
