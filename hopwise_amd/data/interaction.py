"""Batch container with the part of the reference ``Interaction`` API the models and the driver use
(/root/reference/hopwise/data/interaction.py:44-118 container, :190-216 ``to``, :241-288
``repeat`` / ``repeat_interleave`` / ``update``)."""

from __future__ import annotations

import numpy as np
import torch


class Interaction:
    def __init__(self, interaction):
        self.interaction = {}
        for k, v in dict(interaction).items():
            if isinstance(v, np.ndarray):
                v = torch.from_numpy(v)
            elif isinstance(v, (list, tuple)):
                v = torch.as_tensor(v)
            if not isinstance(v, torch.Tensor):
                raise ValueError(f"The type of Interaction [{k}] must be a tensor-like, got {type(v)}")
            self.interaction[k] = v
        self.length = -1
        for v in self.interaction.values():
            self.length = max(self.length, v.shape[0])

    def __iter__(self):
        return iter(self.interaction)

    def __contains__(self, item):
        return item in self.interaction

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        if isinstance(index, str):
            return self.interaction[index]
        return Interaction({k: v[index] for k, v in self.interaction.items()})

    def __setitem__(self, key, value):
        if not isinstance(key, str):
            raise KeyError(f"{type(key)} object does not support item assignment")
        self.interaction[key] = value

    @property
    def columns(self):
        return list(self.interaction.keys())

    def to(self, device, selected_field=None):
        sel = set(selected_field) if selected_field is not None else None
        out = {}
        for k, v in self.interaction.items():
            out[k] = v.to(device) if sel is None or k in sel else v
        return Interaction(out)

    def cpu(self):
        return self.to("cpu")

    def numpy(self):
        return {k: v.cpu().numpy() for k, v in self.interaction.items()}

    def repeat(self, sizes):
        return Interaction({k: v.repeat([sizes] + [1] * (v.dim() - 1)) for k, v in self.interaction.items()})

    def repeat_interleave(self, repeats, dim=0):
        return Interaction({k: v.repeat_interleave(repeats, dim=dim) for k, v in self.interaction.items()})

    def update(self, new_inter):
        for k in new_inter.interaction:
            self.interaction[k] = new_inter.interaction[k]

    def __repr__(self):
        info = [f"The batch_size of interaction: {self.length}"]
        for k, v in self.interaction.items():
            info.append(f"    {k}, {tuple(v.shape)}, {v.device.type}, {v.dtype}")
        return "\n".join(info) + "\n"
