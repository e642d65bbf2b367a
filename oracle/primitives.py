"""Third-party primitives the reference calls, restated on plain torch CPU ops.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference delegates all sparse arithmetic to torch_geometric==2.6.1 and
torch_scatter==2.1.2 (requirements.txt:14-15), neither of which is vendored
under /root/reference nor installed here.  The functions below restate the
published behaviour of exactly the entry points the hot path reaches; each
cites the reference call site that depends on it.  Accumulation is in index
order (what the CPU kernels of torch_scatter do), which is also the order the
HIP kernels use (CSR sorted by original edge id).
"""
from __future__ import annotations

import torch
from torch import Tensor


def scatter_sum(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """torch_scatter.scatter_add / scatter(reduce='sum') along dim 0.

    Call sites: att_pooling.py:73, PyG aggregate behind mgat_v2_conv.py:215.
    """
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
    out.index_add_(0, index, src)
    return out


def scatter_count(index: Tensor, dim_size: int, dtype=torch.float32) -> Tensor:
    cnt = torch.zeros(dim_size, dtype=dtype)
    cnt.index_add_(0, index, torch.ones(index.numel(), dtype=dtype))
    return cnt


def scatter_mean(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """sum / clamp(count, min=1)   (scene_graph_encoder.py:141; GraphNorm)."""
    s = scatter_sum(src, index, dim_size)
    cnt = scatter_count(index, dim_size, src.dtype).clamp(min=1)
    return s / cnt.view((-1,) + (1,) * (src.dim() - 1))


def scatter_max(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """Per-segment maximum along dim 0.  Empty segments hold 0 (never read)."""
    out = torch.full((dim_size,) + tuple(src.shape[1:]), float("-inf"), dtype=src.dtype)
    idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    out.scatter_reduce_(0, idx, src, reduce="amax", include_self=True)
    return torch.where(torch.isinf(out) & (out < 0), torch.zeros_like(out), out)


def pyg_softmax(src: Tensor, index: Tensor, num_nodes: int) -> Tensor:
    """torch_geometric.utils.softmax(src, index, num_nodes=N, dim=0).

    m = scatter_max; e = exp(src - m[index]); e / (scatter_sum(e)[index] + 1e-16)
    Call sites: mgat_v2_conv.py:272, att_pooling.py:71.
    """
    m = scatter_max(src, index, num_nodes)
    e = (src - m.index_select(0, index)).exp()
    s = scatter_sum(e, index, num_nodes) + 1e-16
    return e / s.index_select(0, index)


def scatter_softmax_1d(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """torch_scatter.scatter_softmax(src, index, dim=-1) on a 1-D src: no epsilon.

    Call site: utils/scatter_scaled_dot_product.py:7-14.
    """
    m = scatter_max(src, index, dim_size)
    e = (src - m.index_select(0, index)).exp()
    s = scatter_sum(e, index, dim_size)
    return e / s.index_select(0, index)


def graph_norm(x: Tensor, batch: Tensor, weight: Tensor, bias: Tensor,
               mean_scale: Tensor, eps: float = 1e-5, num_graphs: int | None = None) -> Tensor:
    """torch_geometric.nn.norm.GraphNorm.forward (PyG 2.6.1).

    mean = scatter_mean(x); out = x - mean[batch]*mean_scale;
    var = scatter_mean(out^2); weight*out/sqrt(var+eps)[batch] + bias.
    Call sites: mgat.py:93-95,171; scene_graph_encoder.py:33,101 (fp64 there).
    """
    B = int(batch.max()) + 1 if num_graphs is None else num_graphs
    mean = scatter_mean(x, batch, B)
    out = x - mean.index_select(0, batch) * mean_scale
    var = scatter_mean(out.pow(2), batch, B)
    std = (var + eps).sqrt().index_select(0, batch)
    return weight * out / std + bias


def to_dense_batch(x: Tensor, batch: Tensor, num_graphs: int | None = None):
    """torch_geometric.utils.to_dense_batch(x, batch): zero fill, row order kept.

    Returns (dense[B, Nmax, ...], mask[B, Nmax] bool).  Call site: masking.py:162.
    """
    B = int(batch.max()) + 1 if num_graphs is None else num_graphs
    counts = torch.zeros(B, dtype=torch.long)
    counts.index_add_(0, batch, torch.ones_like(batch))
    ptr = torch.zeros(B + 1, dtype=torch.long)
    ptr[1:] = counts.cumsum(0)
    nmax = int(counts.max())
    pos = torch.arange(batch.numel()) - ptr[batch]
    dense = torch.zeros((B, nmax) + tuple(x.shape[1:]), dtype=x.dtype)
    dense[batch, pos] = x
    mask = torch.zeros(B, nmax, dtype=torch.bool)
    mask[batch, pos] = True
    return dense, mask


def gelu(x: Tensor) -> Tensor:
    """torch.nn.functional.gelu (exact erf form), the default everywhere in the reference."""
    return torch.nn.functional.gelu(x)
