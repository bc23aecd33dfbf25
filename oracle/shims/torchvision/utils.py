import torch


def make_grid(tensors, nrow=8, **kw):
    if isinstance(tensors, (list, tuple)):
        tensors = torch.stack([t if t.dim() == 3 else t.unsqueeze(0) for t in tensors])
    return torch.cat(list(tensors), dim=-1)
