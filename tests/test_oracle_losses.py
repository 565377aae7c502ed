"""Known-answer cases for the sup-con loss restatement (oracle/losses_oracle.py).  The library half of the
reference loss (pytorch-metric-learning 2.2.0) is absent here, so these hand-computed values are what pins it."""
import math

import torch

from oracle import losses_oracle as LO


def test_one_positive_one_negative_is_softplus_of_the_scaled_gap():
    s = torch.tensor([[0.3, -0.2], [0.1, 0.4]])
    y = torch.tensor([[1.0, 0.0], [0.0, 1.0]])
    m = torch.ones(2, 2, dtype=torch.bool)
    want = (math.log1p(math.exp((-0.2 - 0.3) / 0.1)) + math.log1p(math.exp((0.1 - 0.4) / 0.1))) / 2
    assert abs(float(LO.sup_con_loss(s, y, m)) - want) < 1e-6


def test_padding_slots_are_neither_positive_nor_negative():
    s = torch.tensor([[0.5, 0.1, 0.0, 0.0], [0.2, 0.3, -0.1, 0.6]])     # row 0 has 2 real candidates, 2 pads (score 0)
    y = torch.tensor([[1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 1.0]])
    m = torch.tensor([[True, True, False, False], [True, True, True, True]])
    r0 = -(0.5 / 0.1 - math.log(math.exp(5.0) + math.exp(1.0)))
    x = [2.0, 3.0, -1.0, 6.0]
    lse = math.log(sum(math.exp(v) for v in x))
    r1 = -((3.0 - lse) + (6.0 - lse)) / 2                               # two positives: mean of their log-probs
    assert abs(float(LO.sup_con_loss(s, y, m)) - (r0 + r1) / 2) < 1e-5
    idx = LO.indices_tuple(y, m)
    assert idx[2].tolist() == [0, 1, 1] and idx[3].tolist() == [1, 0, 2]


def test_rows_without_positives_drop_out_of_the_mean_and_degenerate_batches_give_zero():
    s = torch.tensor([[0.3, -0.2, 0.1], [0.1, 0.4, 0.2]])
    y = torch.tensor([[1.0, 0.0, 0.0], [0.0, 0.0, 0.0]])
    m = torch.ones(2, 3, dtype=torch.bool)
    only = -(3.0 - math.log(math.exp(3.0) + math.exp(-2.0) + math.exp(1.0)))
    assert abs(float(LO.sup_con_loss(s, y, m)) - only) < 1e-6          # AvgNonZeroReducer: row 1 (loss 0) not counted
    assert float(LO.sup_con_loss(s[:1, :2], y[:1, :2], m[:1, :2])) == 0.0   # one positive pair, one negative pair
    assert float(LO.sup_con_loss(s, torch.zeros(2, 3), m)) == 0.0       # no positive at all


def test_dual_loss_mixes_the_two():
    g = torch.Generator().manual_seed(0)
    s = torch.randn(4, 5, generator=g)
    y = torch.zeros(4, 5)
    y[torch.arange(4), torch.tensor([0, 2, 4, 1])] = 1
    m = torch.ones(4, 5, dtype=torch.bool)
    ce = -(y * torch.log_softmax(s, 1)).sum(1).mean()
    assert abs(float(LO.dual_loss(s, y, m, 0.3)) - float(0.7 * ce + 0.3 * LO.sup_con_loss(s, y, m))) < 1e-6
