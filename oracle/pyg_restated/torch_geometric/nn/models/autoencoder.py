"""Restated subset of torch_geometric.nn.models.autoencoder (PyG 2.0.1) -- TEST INFRASTRUCTURE ONLY.

`/root/reference/data/utils.py:6` imports `negative_sampling` from here at module level; none of the
functions the hot path's data contract uses (`load_data_torch` :34-169, `process_prot_edge` :212-229)
calls it.  The restatement follows the published semantics (uniform random node pairs that are not
edges of `edge_index`) so that the import resolves to something meaningful."""
import torch

from . import InnerProductDecoder  # noqa: F401


def negative_sampling(edge_index, num_nodes=None, num_neg_samples=None):
    n = int(edge_index.max()) + 1 if num_nodes is None else int(num_nodes)
    m = edge_index.shape[1] if num_neg_samples is None else int(num_neg_samples)
    pos = set((edge_index[0] * n + edge_index[1]).tolist())
    out = []
    while len(out) < m:
        cand = torch.randint(0, n * n, (m,)).tolist()
        out.extend(c for c in cand if c not in pos)
    out = torch.tensor(out[:m], dtype=torch.long)
    return torch.stack([out // n, out % n])
