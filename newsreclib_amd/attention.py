"""Parameter container mirroring the reference ``AdditiveAttention`` (layers/attention.py:6-42).

Same constructor checks, same parameter names (``linear.weight``, ``linear.bias``, ``query``) and
the same initialisation; the arithmetic (tanh-linear, softmax, weighted sum) runs inside the fused
encoder kernels, so this module has no standalone ``forward``.
"""
import torch
import torch.nn as nn


class AdditiveAttention(nn.Module):
    def __init__(self, input_dim: int, query_dim: int) -> None:
        super().__init__()
        if not isinstance(input_dim, int):
            raise ValueError(f"Expected keyword argument `input_dim` to be an `int` but got {input_dim}")
        if not isinstance(query_dim, int):
            raise ValueError(f"Expected keyword argument `query_dim` to be an `int` but got {query_dim}")
        self.linear = nn.Linear(in_features=input_dim, out_features=query_dim)
        self.query = nn.Parameter(torch.empty(query_dim).uniform_(-0.1, 0.1))

    def forward(self, input_vector):  # pragma: no cover - fused into the encoders
        raise RuntimeError("newsreclib_amd.AdditiveAttention is fused into MHSAAddAtt / UserEncoder; "
                           "call the encoder instead")
