"""``AdditiveAttention`` with the reference's interface (layers/attention.py:6-42).

Same constructor checks, parameter names (``linear.weight``, ``linear.bias``, ``query``) and
initialisation.  Inside ``MHSAAddAtt`` / ``CNNAddAtt`` / the NRMS user encoder the arithmetic is fused into
those encoders' pipelines (this module is then only the parameter container); called on its own -- the NAML
view combination and user encoder -- it runs ``nrl_additive_attention_fwd/_bwd``.
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

    def forward(self, input_vector: torch.Tensor) -> torch.Tensor:
        """(batch, length, dim) -> (batch, dim): tanh-linear scores, softmax over length (no mask), weighted sum."""
        from . import ops_blocks
        params = (self.linear.weight, self.linear.bias, self.query)
        bufs = tuple(getattr(p, "main_grad", None) for p in params)
        return ops_blocks.AdditiveAttentionFn.apply(input_vector, *params, bufs if any(b is not None for b in bufs) else None)
