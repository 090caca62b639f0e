"""Restated subset of torch_geometric.data (PyG 2.0.1) -- TEST INFRASTRUCTURE ONLY."""
import torch


class Data(object):
    """Attribute bag: `Data.from_dict(d)` sets every key as an attribute; `.to(device)` moves
    tensor attributes (recursing into lists/tuples/dicts) and leaves everything else alone."""

    @classmethod
    def from_dict(cls, dictionary):
        data = cls()
        for key, item in dictionary.items():
            setattr(data, key, item)
        return data

    def to(self, device):
        def move(v):
            if isinstance(v, torch.Tensor):
                return v.to(device)
            if isinstance(v, (list, tuple)):
                return type(v)(move(u) for u in v)
            if isinstance(v, dict):
                return {k: move(u) for k, u in v.items()}
            return v
        for key, item in list(self.__dict__.items()):
            setattr(self, key, move(item))
        return self
