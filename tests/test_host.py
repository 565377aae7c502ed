"""CPU-tier tests (`-m "not gpu"`): the C-ABI library loads and exports what the header declares,
the host-side mirror keeps the reference's interface, the product never touches the oracle, and
the data-parallel plumbing works across two processes (gloo)."""
import ast
import os
import re
import subprocess
import sys
from functools import partial

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _module(**over):
    from newsreclib_amd.nrms_module import NRMSModule
    kw = dict(dataset_attributes=["title", "abstract"], attributes2encode=["title"],
              outputs={"train": ["preds", "targets", "cand_news_size"], "val": [], "test": []},
              dual_loss_training=False, dual_loss_coef=None, loss="cross_entropy_loss", late_fusion=False,
              temperature=None, use_plm=False, pretrained_embeddings_path=None, plm_model=None,
              frozen_layers=None, embed_dim=300, num_heads=15, query_dim=200, dropout_probability=0.2,
              top_k_list=[5, 10], num_categ_classes=18, num_sent_classes=3, save_recs=False, recs_fpath=None,
              optimizer=partial(torch.optim.Adam, lr=1e-4), scheduler=None,
              pretrained_embeddings=torch.randn(64, 300))
    kw.update(over)
    return NRMSModule(**kw)


def test_library_builds_loads_and_exports_every_declared_symbol():
    from newsreclib_amd import _build, _lib
    _build.build(verbose=False)
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "newsreclib_amd.h")).read()
    declared = set(re.findall(r"\b(nrl_[a-z0-9_]+)\s*\(", header))
    assert declared, "header declares no entry points?"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported by the .so"
    assert lib.nrl_abi_version() == _lib.ABI_VERSION
    # host-side part of the dropout spec needs no GPU
    from oracle import nrms_oracle as O
    for seed, stream in [(0, 0), (1234, 1), (2 ** 63 + 5, 7)]:
        assert lib.nrl_dropout_key(seed, stream) == O.dropout_key(seed, stream)


def test_product_fails_loudly_without_gpu_and_never_imports_the_oracle():
    from newsreclib_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.linear(torch.zeros(4, 4), torch.zeros(4, 4))
    mod = _module()
    from newsreclib_amd.synthetic import make_batch
    with pytest.raises(RuntimeError, match="GPU"):
        mod(make_batch(2, 64, "fixed", seed=0))
    pkg = os.path.join(ROOT, "newsreclib_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            tree = ast.parse(open(os.path.join(pkg, fn)).read())
            for node in ast.walk(tree):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    names = [node.module or ""]
                assert not any(n.split(".")[0] == "oracle" for n in names), f"{fn} imports the oracle"


def test_module_interface_matches_reference_contract():
    from oracle import nrms_oracle as O
    mod = _module()
    # state_dict keys == the reference checkpoint keys
    assert set(mod.state_dict().keys()) == set(O.param_shapes(64).keys())
    for k, shape in O.param_shapes(64).items():
        assert tuple(mod.state_dict()[k].shape) == shape
    for attr in ("news_encoder", "user_encoder", "click_predictor"):
        assert isinstance(getattr(mod, attr), torch.nn.Module)
    assert mod.hparams.embed_dim == 300 and mod.hparams.num_heads == 15
    opt = mod.configure_optimizers()["optimizer"]
    assert isinstance(opt, torch.optim.Adam) and opt.defaults["lr"] == 1e-4
    # embedding row 0 is kept (padding_idx only masks its gradient), text.py:215-217
    emb = mod.news_encoder.text_encoders["title"].embedding_layer
    assert emb.padding_idx == 0 and float(emb.weight[0].abs().sum()) > 0
    # reference error behaviour
    from newsreclib_amd.attention import AdditiveAttention
    from newsreclib_amd.news_encoder import MHSAAddAtt
    with pytest.raises(ValueError, match="dropout_probability"):
        MHSAAddAtt(torch.randn(8, 300), 300, 15, 200, 1)
    with pytest.raises(ValueError, match="input_dim"):
        AdditiveAttention(300.0, 200)
    # loss selection as abstract_recommender.py:113-124
    from newsreclib_amd.click_predictor import CrossEntropyLoss, SupConLoss
    assert isinstance(_module(loss="sup_con_loss").criterion, SupConLoss)
    dual = _module(loss="dual_loss", dual_loss_training=True, dual_loss_coef=0.3)
    assert isinstance(dual.ce_criterion, CrossEntropyLoss) and isinstance(dual.scl_criterion, SupConLoss)
    with pytest.raises(ValueError, match="Loss not defined"):
        _module(loss="hinge")
    with pytest.raises(AssertionError):
        _module(loss="dual_loss", dual_loss_training=True, dual_loss_coef=None)


def test_synthetic_batches_are_mind_shaped():
    from newsreclib_amd.synthetic import make_batch
    b = make_batch(32, 70_000, "fixed", seed=1234)
    assert b["x_hist"]["title"].shape == (32 * 50, 30) and b["x_cand"]["title"].shape == (32 * 5, 30)
    assert b["x_hist"]["title"].dtype == torch.int64 and b["labels"].dtype == torch.float32
    assert torch.equal(b["batch_hist"], torch.arange(32).repeat_interleave(50))
    assert float(b["labels"].sum()) == 32 and int(b["x_hist"]["title"].max()) < 70_000
    lens = (b["x_hist"]["title"] != 0).sum(1)
    assert 3 <= int(lens.min()) and int(lens.max()) <= 30 and 10 < float(lens.float().mean()) < 13
    assert torch.equal(make_batch(32, 70_000, "fixed", seed=1234)["x_cand"]["title"], b["x_cand"]["title"])
    r = make_batch(64, 5000, "ragged", seed=3)
    sizes = torch.bincount(r["batch_cand"])
    assert (sizes % 5 == 0).all() and int(torch.bincount(r["batch_hist"]).max()) <= 50
    assert float(r["labels"].sum()) == float(sizes.sum()) / 5


def test_ranking_metrics_known_answers():
    from newsreclib_amd.metrics import ranking_metrics
    preds = torch.tensor([0.9, 0.1, 0.5, 0.2, 0.8, 0.3])
    targets = torch.tensor([1.0, 0.0, 0.0, 0.0, 0.0, 1.0])
    m = ranking_metrics(preds, targets, torch.tensor([3, 3]), [5])
    assert abs(m["mrr"] - (1.0 + 0.5) / 2) < 1e-6
    assert abs(m["ndcg@5"] - (1.0 + 1.0 / np.log2(3)) / 2) < 1e-6
    assert abs(m["auc"] - 6.0 / 8.0) < 1e-6


def test_ranking_metrics_agree_with_scikit_learn_on_ragged_impressions():
    """torchmetrics (the reference's metric library, nrms_module.py:182-195) is not installed here; scikit-learn is, and
    implements the same published definitions independently: binary AUROC with tie-averaged ranks (`roc_auc_score`), nDCG@k
    with the log2 discount (`ndcg_score`; it ranks ties by AVERAGING, so the scores below are tie-free per impression), and
    the reciprocal rank of the first positive.  Impressions without a positive count 0 (empty_target_action="neg")."""
    from sklearn.metrics import ndcg_score, roc_auc_score
    from newsreclib_amd.metrics import ranking_metrics
    rng = np.random.default_rng(7)
    sizes = rng.integers(2, 40, size=200)
    preds, targets = [], []
    for i, n in enumerate(sizes):
        preds.append(rng.permutation(n).astype(np.float32) / 64.0 + rng.integers(0, 3))     # tie-free inside, ties across impressions
        t = (rng.random(n) < 0.15).astype(np.float32)
        if i % 17 == 0:
            t[:] = 0.0                                                                       # an impression without a click
        targets.append(t)
    p, t = np.concatenate(preds), np.concatenate(targets)
    m = ranking_metrics(torch.from_numpy(p), torch.from_numpy(t), torch.from_numpy(sizes), [5, 10])
    assert abs(m["auc"] - roc_auc_score(t, p)) < 1e-6
    for k in (5, 10):
        want = np.mean([ndcg_score(tt[None], pp[None], k=k) if tt.sum() > 0 else 0.0 for pp, tt in zip(preds, targets)])
        assert abs(m[f"ndcg@{k}"] - want) < 1e-6, (k, m[f"ndcg@{k}"], want)
    rr = []
    for pp, tt in zip(preds, targets):
        order = np.argsort(-pp, kind="stable")
        hit = np.nonzero(tt[order] > 0)[0]
        rr.append(1.0 / (hit[0] + 1) if hit.size else 0.0)
    assert abs(m["mrr"] - np.mean(rr)) < 1e-6
    # an epoch's worth of pairs with heavy ties: the rank sums exceed 2^24 many times over
    n = 3_000_000
    p = (rng.integers(0, 4096, size=n) / 4096.0).astype(np.float32)
    t = (rng.random(n) < 0.2 + 0.1 * p).astype(np.float32)
    m = ranking_metrics(torch.from_numpy(p), torch.from_numpy(t), torch.full((n // 5,), 5), [5])
    assert abs(m["auc"] - roc_auc_score(t, p)) < 1e-9


_OVERLAP_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO"])
from newsreclib_amd.trainer import FlatParams, OverlappedGradReduce
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["PORT"],
                        rank=int(os.environ["RANK"]), world_size=2)
rank = dist.get_rank()
torch.manual_seed(0)
emb = torch.nn.Embedding(50, 8)
lin = torch.nn.Linear(8, 4)
params = list(emb.parameters()) + list(lin.parameters())
flat = FlatParams(params)
red = OverlappedGradReduce(flat, flat.offsets[1])


class TwoPhase(torch.autograd.Function):          # mimics ops.NewsEncoderFn: table grad, hook, then weights
    @staticmethod
    def forward(ctx, w, lw, lb):
        return (w.sum() + lw.sum() + lb.sum()).reshape(())

    @staticmethod
    def backward(ctx, g):
        params[0].main_grad.add_(float(rank + 1))          # "phase 1": table gradient complete
        red.start_head(params[0].main_grad)
        params[1].main_grad.add_(10.0 * (rank + 1))        # "phase 2": weight gradients
        params[2].main_grad.add_(100.0 * (rank + 1))
        return None, None, None


TwoPhase.apply(*params).backward()
scale = red.finish()
assert scale == 0.5
assert torch.allclose(params[0].main_grad * scale, torch.full_like(params[0], 1.5))
assert torch.allclose(params[1].main_grad * scale, torch.full_like(params[1], 15.0))
assert torch.allclose(params[2].main_grad * scale, torch.full_like(params[2], 150.0))
# hook never fired -> finish() must still reduce everything
flat.grad.zero_()
flat.grad.add_(float(rank + 1))
assert red.finish() == 0.5 and torch.allclose(flat.grad, torch.full_like(flat.grad, 3.0))
# chunked head + pipelined finish: the slices tile the flat buffer exactly once, each holds the summed gradient when it
# is handed out (the trainer runs Adam on it while the next slice is still on the wire)
red3 = OverlappedGradReduce(flat, red.head, chunks=3)
assert red3.bounds[0] == 0 and red3.bounds[-1] == red.head and all(b % 64 == 0 for b in red3.bounds[:-1])
flat.grad.zero_()
flat.grad.add_(torch.arange(flat.numel, dtype=torch.float32) * (rank + 1))
red3.start_head(params[0].main_grad)
seen = torch.zeros(flat.numel)
for lo, hi, scale in red3.finish_pipelined():
    assert scale == 0.5
    assert torch.equal(flat.grad[lo:hi], torch.arange(lo, hi, dtype=torch.float32) * 3.0)
    seen[lo:hi] += 1
assert torch.equal(seen, torch.ones(flat.numel))
assert red3.info()["head_chunks"] == 3
dist.destroy_process_group()
print("OK", rank)
"""

_ROWS_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO"])
from newsreclib_amd.trainer import FlatParams, OverlappedGradReduce, TouchedRowsExchange
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["PORT"],
                        rank=int(os.environ["RANK"]), world_size=2)
rank = dist.get_rank()
V, D = 97, 12
torch.manual_seed(0)
def fresh():
    params = [torch.nn.Parameter(torch.zeros(V, D)), torch.nn.Parameter(torch.zeros(5, 7)), torch.nn.Parameter(torch.zeros(3))]
    return params, FlatParams(params)
def fill(params, r):
    # what a backward leaves in main_grad: rows of the rank's own ids (duplicates, the pad id 0, ids shared with the
    # other rank), dense gradients for the rest
    g = torch.Generator().manual_seed(100 + r)
    ids = torch.randint(0, V, (6, 9), generator=g)
    ids[0, :3] = 0
    ids[1, :] = torch.arange(40, 49)            # shared between the ranks
    rows = torch.randn(ids.numel(), D, generator=g)
    params[0].main_grad.index_add_(0, ids.reshape(-1), rows)
    params[1].main_grad.add_(torch.randn(5, 7, generator=g))
    params[2].main_grad.add_(torch.randn(3, generator=g))
    return ids
# dense reference
pd, fd = fresh()
fill(pd, rank)
red = OverlappedGradReduce(fd, V * D)
red.start_head(pd[0].main_grad)
assert red.finish() == 0.5
# touched-row exchange
pr, fr = fresh()
ids = fill(pr, rank)
ex = TouchedRowsExchange(fr, V * D, pr[0])
ex.start_head(pr[0].main_grad, ids)
assert ex.finish() == 0.5
assert torch.equal(fr.grad, fd.grad), "touched-row exchange != dense all-reduce (must be bit-identical at 2 ranks)"
# == the sum of the two ranks' local gradients, and identical on both replicas
tot = None
for r in range(2):
    p2, f2 = fresh()
    fill(p2, r)
    tot = f2.grad.clone() if tot is None else tot + f2.grad
assert torch.allclose(fr.grad, tot, atol=1e-6)
both = [torch.empty_like(fr.grad) for _ in range(2)]
dist.all_gather(both, fr.grad)
assert torch.equal(both[0], both[1]), "replicas diverged"
info = ex.info()
assert info["mode"] == "rows" and 0 < info["payload_bytes_per_rank"] < info["dense_payload_bytes_per_rank"], info
assert info["last_step_choice"] == "rows" and info["predicted_wire_ms"]["dense"]["ring"] > 0, info
# the early form: unique ids + the async count exchange prepared BEFORE the backward (what NRMSTrainer.step does), here
# with the id-sorted visiting order the library's sort would hand over (n + 1 entries)
pp, fp = fresh()
ids = fill(pp, rank)
exp = TouchedRowsExchange(fp, V * D, pp[0])
flat_ids = ids.reshape(-1)
order = torch.cat([torch.argsort(flat_ids, stable=True), (flat_ids == 0).sum().reshape(1)])
exp.prepare(ids, order)
exp.start_head(pp[0].main_grad)                 # (no ids here: the prepared state carries them)
assert exp.finish() == 0.5 and torch.equal(fp.grad, fd.grad), "prepared touched-row exchange != dense all-reduce"
assert exp.info()["max_unique_rows_per_rank"] == max(int(torch.unique(i).numel()) for i in
                                                     [fill(fresh()[0], 0), fill(fresh()[0], 1)])
# auto (legacy two-way form): whichever of rows / dense the wire model prices lower.  ~50 unique rows of 97 per rank at two
# ranks: the all-gather moves 1 x 50 x (8 + 48) B per rank, the ring all-reduce 2 x 1/2 x 4656 B -> rows; with EVERY row of the
# table touched the all-gather is the larger one -> dense.  Either way the result is the dense all-reduce's.
pa, fa = fresh()
ids = fill(pa, rank)
exa = TouchedRowsExchange(fa, V * D, pa[0], auto=True)
exa.start_head(pa[0].main_grad, ids)
assert exa.finish() == 0.5 and torch.equal(fa.grad, fd.grad)
assert exa.info()["mode"] == "auto" and exa.info()["last_step_choice"] == "rows", exa.info()
pall, fall = fresh()
ids = fill(pall, rank)
every = torch.arange(V).reshape(1, V)
pall[0].main_grad.add_(1.0)
dall = fall.grad.clone()
dist.all_reduce(dall)
exall = TouchedRowsExchange(fall, V * D, pall[0], auto=True)
exall.start_head(pall[0].main_grad, every)
assert exall.finish() == 0.5 and torch.equal(fall.grad, dall)
assert exall.info()["last_step_choice"] == "dense", exall.info()
# ragged batches: this rank's TOKEN count (one slot per token in its unique buffer) is smaller than the other rank's UNIQUE
# count -- the padded buffers of the two all-gathers must still have one size on both ranks (round-4 advisor finding)
prg, frg = fresh()
g = torch.Generator().manual_seed(7 + rank)
ids_r = torch.arange(1, 4).reshape(1, 3) if rank == 0 else torch.randperm(V, generator=g)[:60].reshape(6, 10)
prg[0].main_grad.index_add_(0, ids_r.reshape(-1), torch.randn(ids_r.numel(), D, generator=g))
drg = frg.grad.clone()
dist.all_reduce(drg)
exr = TouchedRowsExchange(frg, V * D, prg[0])
exr.prepare(ids_r)
exr.start_head(prg[0].main_grad)
assert exr.finish() == 0.5 and torch.equal(frg.grad, drg), "ragged touched-row exchange != dense all-reduce"
# the hook never fired (or carried no ids): dense fallback
pf, ff = fresh()
fill(pf, rank)
ex2 = TouchedRowsExchange(ff, V * D, pf[0])
assert ex2.finish() == 0.5 and torch.equal(ff.grad, fd.grad)
dist.destroy_process_group()
print("OK", rank)
"""


_OWNERS_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO"])
from newsreclib_amd.trainer import FlatParams, OverlappedGradReduce, OwnerRowsExchange, predicted_wire_ms
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["PORT"],
                        rank=int(os.environ["RANK"]), world_size=2)
rank = dist.get_rank()
V, D = 97, 12
def fresh(v=V):
    params = [torch.nn.Parameter(torch.zeros(v, D)), torch.nn.Parameter(torch.zeros(5, 7)), torch.nn.Parameter(torch.zeros(3))]
    return params, FlatParams(params)
def fill(params, r, shape=(6, 9), v=V):
    g = torch.Generator().manual_seed(100 + r)
    ids = torch.randint(0, v, shape, generator=g)
    ids[0, :3] = 0
    ids[1, :] = torch.arange(40, 40 + shape[1])  # shared between the ranks
    rows = torch.randn(ids.numel(), D, generator=g)
    params[0].main_grad.index_add_(0, ids.reshape(-1), rows)
    params[1].main_grad.add_(torch.randn(5, 7, generator=g))
    params[2].main_grad.add_(torch.randn(3, generator=g))
    return ids
def dense(shape=(6, 9), v=V, shapes=None):
    pd, fd = fresh(v)
    fill(pd, rank, shapes[rank] if shapes else shape, v)
    red = OverlappedGradReduce(fd, v * D)
    red.start_head(pd[0].main_grad)
    assert red.finish() == 0.5
    return fd.grad
fd = dense()
# owner-partitioned exchange: prepared before the "backward", all-to-all to the owners, sum in rank order, all-gather of
# the reduced rows: bit-identical to the dense all-reduce at two ranks
po, fo = fresh()
ids = fill(po, rank)
ex = OwnerRowsExchange(fo, V * D, po[0], mode="owners", id_capacity=64)
ex.prepare(ids)
ex.start_head(po[0].main_grad)
assert ex.finish() == 0.5
assert torch.equal(fo.grad, fd), "owner-partitioned exchange != dense all-reduce (must be bit-identical at 2 ranks)"
both = [torch.empty_like(fo.grad) for _ in range(2)]
dist.all_gather(both, fo.grad)
assert torch.equal(both[0], both[1]), "replicas diverged"
info = ex.info()
assert info["last_step_choice"] == "owners" and 0 < info["payload_bytes_per_rank"] < info["dense_payload_bytes_per_rank"], info
# the union it reports (the lazy optimizer's marks) = the ids either rank touched
uni = torch.unique(torch.cat([fill(fresh()[0], 0).reshape(-1), fill(fresh()[0], 1).reshape(-1)]))
got = torch.sort(torch.cat(ex.last_gathered)).values
assert torch.equal(got, uni) and info["union_rows"] == uni.numel(), (got, uni)
# unprepared (the hook carries the ids) and ragged (rank 0 has 3 tokens, rank 1 sixty): same result as dense
shapes = [(2, 9), (8, 9)]
fdr = dense(shapes=shapes)
pr, fr = fresh()
ids = fill(pr, rank, shapes[rank])
ex2 = OwnerRowsExchange(fr, V * D, pr[0], mode="owners", id_capacity=64)
ex2.start_head(pr[0].main_grad, ids)
assert ex2.finish() == 0.5 and torch.equal(fr.grad, fdr), "ragged owner exchange != dense"
# a rank with more unique ids than the capacity: every rank takes the dense all-reduce that step
pc, fc = fresh()
ids = fill(pc, rank)
ex3 = OwnerRowsExchange(fc, V * D, pc[0], mode="owners", id_capacity=8)
ex3.prepare(ids)
ex3.start_head(pc[0].main_grad)
assert ex3.finish() == 0.5 and torch.equal(fc.grad, fd) and ex3.info()["last_step_choice"].startswith("dense"), ex3.info()
# auto: min over the wire model's prices -- identical decision on both ranks, result always the dense all-reduce's
for v in (V, 40 * V):
    pa, fa = fresh(v)
    ids = fill(pa, rank, v=V)
    ref = fa.grad.clone()
    dist.all_reduce(ref)
    exa = OwnerRowsExchange(fa, v * D, pa[0], mode="auto", id_capacity=64)
    exa.prepare(ids)
    exa.start_head(pa[0].main_grad)
    assert exa.finish() == 0.5 and torch.equal(fa.grad, ref)
    i = exa.info()
    priced = exa._last["priced_ms"]
    assert i["last_step_choice"] == min(priced, key=priced.get), i
    ch = [None, None]
    dist.all_gather_object(ch, i["last_step_choice"])
    assert ch[0] == ch[1]
assert i["last_step_choice"] in ("owners", "rows")          # a table 40x larger than what a step touches never goes dense
# hook never fired: dense fallback
pf, ff = fresh()
fill(pf, rank)
ex4 = OwnerRowsExchange(ff, V * D, pf[0], mode="owners")
assert ex4.finish() == 0.5 and torch.equal(ff.grad, fd)
# the predictor at the configs[2] rank shape (V = 150k, D = 300, 8 ranks, ~9.4k unique rows per rank): rows and owners both
# beat the dense ring all-reduce by > 3x under either model (VERDICT round 4, weak item 14)
S = 4.0 * 150_000 * 300
pw = lambda m, *a: predicted_wire_ms(m, *a)
for model in ("ring", "direct"):
    assert pw("rows", 9400 * 1208, 8)[model] * 3 < pw("dense", S, 8)[model]
    assert pw("owners", 9400 * 1200, 8, 5000 * 1200)[model] * 3 < pw("dense", S, 8)[model]
dist.destroy_process_group()
print("OK", rank)
"""


_DP_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO"])
from newsreclib_amd.trainer import FlatParams, allreduce_gradients
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["PORT"],
                        rank=int(os.environ["RANK"]), world_size=2)
rank = dist.get_rank()
torch.manual_seed(0)
lin = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
flat = FlatParams(lin.parameters())
assert all(p.data_ptr() >= flat.flat.data_ptr() for p in lin.parameters())
x = torch.arange(12.).reshape(2, 6) * (rank + 1)
lin(x).sum().backward()
for p in lin.parameters():          # emulate kernels that accumulate into main_grad
    p.main_grad.add_(p.grad)
scale = allreduce_gradients(flat.grad)
assert scale == 0.5
# reference: mean of the two ranks' gradients computed locally
ref = []
for r in range(2):
    torch.manual_seed(0)
    l2 = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
    l2(torch.arange(12.).reshape(2, 6) * (r + 1)).sum().backward()
    ref.append([p.grad.clone() for p in l2.parameters()])
for i, p in enumerate(lin.parameters()):
    want = (ref[0][i] + ref[1][i]) / 2
    assert torch.allclose(p.main_grad * scale, want, atol=1e-5), i
dist.destroy_process_group()
print("OK", rank)
"""


def _run_two_ranks(tmp_path, script_text, name):
    script = tmp_path / name
    script.write_text(script_text)
    port = str(29500 + (os.getpid() * 7 + len(name)) % 2000)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), PORT=port, REPO=ROOT, MASTER_ADDR="127.0.0.1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0 and b"OK" in out, out.decode()


def test_overlapped_gradient_reduce_two_processes_gloo(tmp_path):
    """The two-piece all-reduce (async head launched from inside backward, tail afterwards) equals one
    sum all-reduce of the whole flat gradient; also when the hook never fires."""
    _run_two_ranks(tmp_path, _OVERLAP_SCRIPT, "overlap.py")


def test_touched_row_exchange_two_processes_gloo(tmp_path):
    """The optional touched-row gradient exchange (all-gather of unique ids + their table-gradient rows, dense rest
    all-reduced) is bit-identical to the dense all-reduce at two ranks, equals the sum of the per-rank gradients,
    leaves both replicas identical, ships fewer bytes, and falls back to the dense all-reduce when the hook did not fire."""
    _run_two_ranks(tmp_path, _ROWS_SCRIPT, "rows.py")


def test_owner_partitioned_row_exchange_two_processes_gloo(tmp_path):
    """The owner-partitioned exchange (rank r owns rows id % world == r: all-to-all of the touched rows to their owners, sum in
    rank order, all-gather of the reduced rows) is bit-identical to the dense all-reduce at two ranks -- prepared, unprepared,
    ragged, over capacity (dense that step), under `auto` (min over the wire model), and when the hook never fired."""
    _run_two_ranks(tmp_path, _OWNERS_SCRIPT, "owners.py")


def test_data_parallel_allreduce_two_processes_gloo(tmp_path):
    """N>1 path on CPU: flat gradient buffer + sum all-reduce + 1/world scale == mean of per-rank
    gradients (what reference DDP computes), world_size 2 over gloo."""
    script = tmp_path / "dp.py"
    script.write_text(_DP_SCRIPT)
    port = str(29500 + os.getpid() % 2000)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), PORT=port, REPO=ROOT, MASTER_ADDR="127.0.0.1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0 and b"OK" in out, out.decode()


def test_aspect_metrics_match_per_impression_restatement():
    """Vectorised diversity / personalization == the per-impression loops of the reference's functional
    definitions (oracle/metrics_oracle.py), incl. ragged sizes, k larger than an impression and the
    all-zero-aspect rule."""
    from newsreclib_amd.metrics import aspect_metrics
    from oracle import metrics_oracle as MO
    g = torch.Generator().manual_seed(3)
    cand_sizes = torch.tensor([5, 12, 3, 40, 7, 9])
    hist_sizes = torch.tensor([1, 50, 4, 9, 2, 6])
    nc = 19
    preds = torch.randn(int(cand_sizes.sum()), generator=g)
    cand_a = torch.randint(0, nc, (int(cand_sizes.sum()),), generator=g)
    hist_a = torch.randint(0, nc, (int(hist_sizes.sum()),), generator=g)
    cand_a[5:17] = 0                                   # impression 1: aspects sum to 0 -> scores 0
    got = aspect_metrics(preds, cand_a, hist_a, cand_sizes, hist_sizes, nc, [5, 10])
    ref = MO.aspect_metrics(preds, cand_a, hist_a, cand_sizes, hist_sizes, nc, [5, 10])
    assert got.keys() == ref.keys()
    for k in ref:
        assert abs(got[k] - ref[k]) <= 1e-6, (k, got[k], ref[k])
    # known answer: one impression, top-2 of aspects (1, 2) -> entropy ln2 / ln4 = 0.5; history (1, 1) -> 1/3
    m = aspect_metrics(torch.tensor([0.9, 0.8, 0.1]), torch.tensor([1, 2, 3]), torch.tensor([1, 1]),
                       torch.tensor([3]), torch.tensor([2]), 4, [2])
    assert abs(m["categ_div@2"] - 0.5) <= 1e-6 and abs(m["categ_pers@2"] - 1.0 / 3.0) <= 1e-6


def test_ranking_metrics_count_impressions_without_a_positive_as_zero():
    """torchmetrics' RetrievalMRR / RetrievalNormalizedDCG default to empty_target_action="neg" (the reference
    builds them with defaults, nrms_module.py:186,190): an all-negative impression scores 0 and STAYS in the mean."""
    from newsreclib_amd.metrics import ranking_metrics
    preds = torch.tensor([0.9, 0.1, 0.5, 0.2, 0.8, 0.3, 0.7, 0.6])
    targets = torch.tensor([1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0])
    m = ranking_metrics(preds, targets, torch.tensor([3, 3, 2]), [5])
    assert abs(m["mrr"] - (1.0 + 0.5 + 0.0) / 3) < 1e-6
    assert abs(m["ndcg@5"] - (1.0 + 1.0 / np.log2(3) + 0.0) / 3) < 1e-6


def test_save_recs_writes_the_recommendations_file(tmp_path):
    """abstract_recommender.py:159-193 / nrms_module.py:520-531: {"U<user>": {"N<news>": score}}; a later score of
    the same (user, news) overwrites the earlier one."""
    import json
    fpath = str(tmp_path / "recs.json")
    mod = _module(save_recs=True, recs_fpath=fpath,
                  outputs={"train": [], "val": [],
                           "test": ["preds", "targets", "cand_news_size", "user_ids", "cand_news_ids"]})
    out = mod.test_step_outputs
    out["preds"] += [torch.tensor([0.5, 0.25, 0.75]), torch.tensor([1.5, -1.0])]
    out["targets"] += [torch.tensor([1.0, 0.0, 0.0]), torch.tensor([0.0, 1.0])]
    out["cand_news_size"] += [torch.tensor([2, 1]), torch.tensor([2])]
    out["user_ids"] += [torch.tensor([7, 9]), torch.tensor([7])]
    out["cand_news_ids"] += [torch.tensor([11, 12, 13]), torch.tensor([11, 14])]
    mod.on_test_epoch_end()
    got = json.load(open(fpath))
    assert got == {"U7": {"N11": 1.5, "N12": 0.25, "N14": -1.0}, "U9": {"N13": 0.75}}
    assert all(len(v) == 0 for v in mod.test_step_outputs.values())
    # a configuration that cannot produce the file fails loudly instead of silently writing nothing
    bad = _module(save_recs=True, recs_fpath=fpath, outputs={"train": [], "val": [], "test": ["preds"]})
    bad.test_step_outputs["preds"].append(torch.tensor([0.1]))
    with pytest.raises(RuntimeError, match="save_recs"):
        bad.on_test_epoch_end()


_EPOCH_SYNC_SCRIPT = r"""
import os, sys, json, torch, torch.distributed as dist
from functools import partial
sys.path.insert(0, os.environ["REPO"])
from newsreclib_amd.nrms_module import NRMSModule
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["PORT"],
                        rank=int(os.environ["RANK"]), world_size=2)
rank = dist.get_rank()
def module():
    return NRMSModule(dataset_attributes=["title"], attributes2encode=["title"],
        outputs={"train": [], "val": ["preds", "targets", "cand_news_size"], "test": []},
        dual_loss_training=False, dual_loss_coef=None, loss="cross_entropy_loss", late_fusion=False, temperature=None,
        use_plm=False, pretrained_embeddings_path=None, plm_model=None, frozen_layers=None, embed_dim=300, num_heads=15,
        query_dim=200, dropout_probability=0.2, top_k_list=[5, 10], num_categ_classes=18, num_sent_classes=3,
        save_recs=False, recs_fpath=None, optimizer=partial(torch.optim.Adam, lr=1e-4), scheduler=None,
        pretrained_embeddings=torch.randn(16, 300))
g = torch.Generator().manual_seed(5)
sizes = [torch.tensor([3, 5, 2]), torch.tensor([4, 6]), torch.tensor([7]), torch.tensor([2, 2, 9])]   # 4 "steps"
preds = [torch.randn(int(s.sum()), generator=g) for s in sizes]
targets = [(torch.rand(int(s.sum()), generator=g) < 0.3).float() for s in sizes]
losses = [0.7, 0.5, 0.9, 0.3]
def feed(mod, steps):
    for i in steps:
        mod.val_step_outputs["preds"].append(preds[i]); mod.val_step_outputs["targets"].append(targets[i])
        mod.val_step_outputs["cand_news_size"].append(sizes[i]); mod._track("val", torch.tensor(losses[i]))
# rank r owns steps r, r + 2 (what a DistributedSampler shard looks like); every rank must log the metrics of ALL
# impressions and the mean loss of ALL steps
shard = module(); feed(shard, [rank, rank + 2])
got = shard._epoch_end("val", shard.val_step_outputs)
dist.destroy_process_group()                      # the single-process reference below must not gather
whole = module(); feed(whole, [0, 1, 2, 3])
ref = whole._epoch_end("val", whole.val_step_outputs)
assert got.keys() == ref.keys(), (got, ref)
for k in ref:
    # the global AUC and the per-impression means do not depend on the impression ORDER (rank-major here)
    assert abs(got[k] - ref[k]) < 1e-6, (k, got[k], ref[k])
assert all(len(v) == 0 for v in shard.val_step_outputs.values())
print("OK", rank)
"""


def test_epoch_end_metrics_cover_every_rank_two_processes_gloo(tmp_path):
    """Under data parallelism every rank logs the loss / AUC / MRR / nDCG of ALL ranks' impressions (what the
    reference's torchmetrics objects do by synchronising their states), not of its own shard."""
    _run_two_ranks(tmp_path, _EPOCH_SYNC_SCRIPT, "epoch_sync.py")


def test_per_call_engine_field_and_backward_guard():
    """The GEMM engine a forward ran under travels with the call (NrlBlockParams.gemm_engine / autograd ctx); entry
    points without the field refuse a backward under another engine."""
    from newsreclib_amd import _lib
    lib = _lib.load()
    before = _lib.get_gemm_engine()
    try:
        _lib.set_gemm_engine("f32")
        code = _lib.engine_code()
        assert code == 1
        _lib.set_gemm_engine("bf16x3")
        assert _lib.engine_code() == 2
        with pytest.raises(RuntimeError, match="engine changed"):
            _lib.require_engine(code, "a test call")
        _lib.require_engine(2, "a test call")
        bp = _lib.NrlBlockParams(1, 1, 1, 1, 1, 1, 1, 300, 15, 200, 3)       # bad per-call engine value
        rc = lib.nrl_news_encoder_fwd(bp, None, 1, None, 0, 30, 0.0, 0, 0, 0, None, None, 0, None)
        assert rc == -1 and b"gemm_engine" in lib.nrl_last_error()
    finally:
        _lib.set_gemm_engine(before)


def test_library_build_id_is_the_content_hash_of_its_sources():
    """The loaded library says which sources it was built from (nrl_build_id(), 32 hex digits); `_build.source_hash()` over
    the translation units and the files they include must reproduce it -- `_lib.load()` refuses a library for which it does
    not (file times play no part), and `_build.library_build_id()` reads the same id without loading the library."""
    from newsreclib_amd import _build, _lib
    lib = _lib.load()
    have = lib.nrl_build_id().decode()
    assert len(have) == 32 and all(c in "0123456789abcdef" for c in have)
    assert have == _build.source_hash() == _build.library_build_id()
    assert not _build._stale()
    # every translation unit's hash covers the files it includes: the ABI header is in each closure
    seen = {}
    _build._closure(os.path.join(_build.CSRC, "nrl_api.hip"), seen)
    names = {os.path.basename(p) for p in seen}
    assert {"nrl_api.hip", "nrl_api_internal.h", "nrl_common.h", "newsreclib_amd.h", "nrl_news_tail_api.h"} <= names
    assert "nrl_news_tail.h" not in names         # the tail kernels live in their own unit



def test_option_names_agree_between_header_library_source_and_python():
    """The kernel-selection switches are addressed by bit: the order in include/newsreclib_amd.h (the ABI's statement),
    in the library (kOptName, nrl_api.hip) and in _lib.OPTION_NAMES must be one list."""
    import re
    from newsreclib_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "newsreclib_amd.h")).read()
    doc = re.findall(r'^ \*\s+(\d+) "(\w+)"', hdr, re.M)
    assert [int(i) for i, _ in doc] == list(range(len(doc)))
    src = open(os.path.join(root, "newsreclib_amd", "csrc", "nrl_api.hip")).read()
    body = src[src.index("kOptName[O_COUNT] = {"):]
    names = re.findall(r'"(\w+)"', body[:body.index("};")])
    assert tuple(n for _, n in doc) == tuple(names) == tuple(_lib.OPTION_NAMES)


def test_dense_rows_of_a_full_batch_is_the_reshape_and_its_adjoint():
    """``dense_rows(..., max_is_exact=True)``: when every group is full (N == B * max with the TRUE maximum) the dense tensor of
    ``to_dense_batch`` is the row-major reshape -- checked against the oracle's to_dense_batch, forward and backward; a
    batch that merely has B * cap rows under a truncating cap is NOT taken for full unless the caller vouches for the maximum."""
    from newsreclib_amd.dense_batch import dense_rows
    from oracle import nrms_oracle as O
    gen = torch.Generator().manual_seed(5)
    B, H, D = 4, 3, 8
    x = torch.randn(B * H, D, generator=gen, requires_grad=True)
    batch = torch.arange(B).repeat_interleave(H)
    offsets = torch.arange(B + 1) * H
    dense = dense_rows(x, batch, B, H, offsets, max_is_exact=True)
    ref, mask = O.to_dense_batch(x.detach(), batch)
    assert torch.equal(dense.detach(), ref) and bool(mask.all())
    g = torch.randn(B, H, D, generator=gen)
    dense.backward(g)
    assert torch.equal(x.grad, g.reshape(B * H, D))
    labels = torch.arange(B * H, dtype=torch.float32)
    y = dense_rows(labels, batch, B, H, offsets, max_is_exact=True)
    assert torch.equal(y, O.to_dense_batch(labels, batch)[0])
    # without the caller's word the product path is taken (and, with no GPU here, refuses loudly rather than guessing)
    with pytest.raises((RuntimeError, ValueError, OSError)):
        dense_rows(x.detach(), batch, B, H, offsets)


def test_product_matches_the_reference_contract_fixture():
    """The drop-in contract as the REFERENCE's files state it (tests/golden/reference_contract.json, generated by
    tests/golden/make_contract.py with `ast` / `yaml` over the reference sources and by instantiating its importable
    components): constructor keywords in the reference's order (the product may append defaulted extras), forward
    arguments, every scalar key of configs/model/<name>.yaml accepted by the mirrored module, and the components' state_dict
    keys + shapes at the configs' sizes."""
    import importlib
    import inspect
    import json
    c = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_contract.json")))

    def product(name):
        mod, cls = name.rsplit(".", 1)
        return getattr(importlib.import_module("newsreclib_amd." + mod), cls)

    for group in ("modules", "components"):
        for name, ref in c[group].items():
            cls = product(name)
            for fn in ("__init__", "forward"):
                if fn not in ref:
                    continue
                sig = inspect.signature(getattr(cls, fn))
                mine = [p for p in sig.parameters.values() if p.name != "self" and p.kind == p.POSITIONAL_OR_KEYWORD]
                names = [p.name for p in mine]
                want = ref[fn]["args"]
                assert names[: len(want)] == want, (name, fn, names, want)
                for p in mine[len(want):]:            # extras must be optional, or the reference's call sites break
                    assert p.default is not inspect.Parameter.empty, (name, fn, p.name)
    # Hydra: overriding `model._target_` alone must work, so every key of the reference's model config is a ctor keyword
    for cfg_name, cfg in c["configs"].items():
        target = cfg["values"]["_target_"]
        assert target.startswith("newsreclib.models.general_rec."), target
        mod_name, cls_name = target[len("newsreclib.models.general_rec."):].rsplit(".", 1)
        cls = product(mod_name + "." + cls_name)
        params = inspect.signature(cls.__init__).parameters
        for key in cfg["values"]:
            if not key.startswith("_"):
                assert key in params, (cfg_name, key)
    # state_dict keys + shapes of the components, built with the same constructor calls as make_contract.py
    from newsreclib_amd.attention import AdditiveAttention
    from newsreclib_amd.news_encoder import CNNAddAtt, CNNMHSAAddAtt, LinearEncoder, MHSAAddAtt
    from newsreclib_amd.user_encoder import UserEncoder as NrmsUser
    from newsreclib_amd.user_encoder_cen_news_rec import UserEncoder as CenUser
    from newsreclib_amd.user_encoder_lstur import UserEncoder as LsturUser
    from newsreclib_amd.user_encoder_mins import UserEncoder as MinsUser
    from newsreclib_amd.user_encoder_naml import UserEncoder as NamlUser
    V = c["state_dicts"]["vocab_rows_used"]
    emb = torch.zeros(V, 300)
    built = {
        "news_encoder.MHSAAddAtt": MHSAAddAtt(pretrained_embeddings=emb, embed_dim=300, num_heads=15, query_dim=200,
                                              dropout_probability=0.2),
        "news_encoder.CNNAddAtt": CNNAddAtt(pretrained_embeddings=emb, embed_dim=300, num_filters=300, window_size=3,
                                            query_dim=200, dropout_probability=0.2),
        "news_encoder.CNNMHSAAddAtt": CNNMHSAAddAtt(pretrained_embeddings=emb, embed_dim=300, num_filters=400, window_size=3,
                                                    num_heads=20, query_dim=200, dropout_probability=0.2),
        "news_encoder.LinearEncoder": LinearEncoder(pretrained_embeddings=None, from_pretrained=False, freeze_pretrained_emb=False,
                                                    num_categories=19, embed_dim=100, use_dropout=False,
                                                    dropout_probability=None, linear_transform=False, output_dim=None),
        "attention.AdditiveAttention": AdditiveAttention(input_dim=300, query_dim=200),
        "user_encoder.UserEncoder": NrmsUser(news_embed_dim=300, num_heads=15, query_dim=200),
        "user_encoder_lstur.UserEncoder": LsturUser(num_users=100, input_dim=700, user_masking_probability=0.5,
                                                    long_short_term_method="ini"),
        "user_encoder_naml.UserEncoder": NamlUser(news_embed_dim=400, query_dim=200),
        "user_encoder_mins.UserEncoder": MinsUser(news_embed_dim=300, query_dim=200, num_filters=300, num_gru_channels=6),
        "user_encoder_cen_news_rec.UserEncoder": CenUser(num_filters=400, num_heads=20, query_dim=200, gru_hidden_dim=400,
                                                         num_recent_news=20, dropout_probability=0.2),
    }
    for name, m in built.items():
        got = {k: list(v.shape) for k, v in m.state_dict().items()}
        assert got == c["state_dicts"][name], (name, set(got) ^ set(c["state_dicts"][name]))


def test_trainer_losses_are_bounded_and_optimizer_state_round_trips():
    """NRMSTrainer keeps at most LOSS_FOLD loss scalars whatever the caller does, `epoch_end` still reports the exact mean,
    and state_dict / load_state_dict carry Adam's step count and moments (CPU: the bookkeeping only, no kernels)."""
    from newsreclib_amd.trainer import NRMSTrainer
    lin = torch.nn.Linear(4, 3)
    tr = NRMSTrainer.__new__(NRMSTrainer)
    tr._losses, tr._loss_sum, tr._loss_count, tr.group = [], None, 0, None
    vals = torch.arange(1.0, 1001.0)
    for v in vals:
        tr._losses.append(v.clone())
        if len(tr._losses) >= tr.LOSS_FOLD:
            tr._fold_losses()
        assert len(tr._losses) < tr.LOSS_FOLD
    assert abs(tr.epoch_end()["train/loss"] - float(vals.mean())) < 1e-9
    assert tr.epoch_end() == {}
    from newsreclib_amd.trainer import FlatParams, FusedAdam
    flat = FlatParams(lin.parameters())
    tr.module, tr.flat, tr.opt = lin, flat, FusedAdam(flat, lr=3e-4)
    tr.opt.step_count = 7
    tr.opt.exp_avg.normal_()
    tr.opt.exp_avg_sq.uniform_()
    state = tr.state_dict()
    assert [n for n, _, _ in state["layout"]] == ["weight", "bias"]
    lin2 = torch.nn.Linear(4, 3)
    tr2 = NRMSTrainer.__new__(NRMSTrainer)
    flat2 = FlatParams(lin2.parameters())
    tr2.module, tr2.flat, tr2.opt = lin2, flat2, FusedAdam(flat2, lr=1e-4)
    tr2.load_state_dict(state)
    assert tr2.opt.step_count == 7 and tr2.opt.lr == 3e-4
    assert torch.equal(tr2.opt.exp_avg, tr.opt.exp_avg) and torch.equal(tr2.opt.exp_avg_sq, tr.opt.exp_avg_sq)
    other = NRMSTrainer.__new__(NRMSTrainer)
    lin3 = torch.nn.Linear(5, 3)
    flat3 = FlatParams(lin3.parameters())
    other.module, other.flat, other.opt = lin3, flat3, FusedAdam(flat3)
    with pytest.raises(ValueError, match="different parameter layout"):
        other.load_state_dict(state)


def test_frozen_image_cache_keys_and_invalidation():
    """FrozenImages: a buffer is ready only after `commit`, never for a trainable weight, and `invalidate`,
    the module-wide generation, an in-place modification and a re-freeze all force a rebuild."""
    from newsreclib_amd import ops_blocks

    class FakeLib:
        @staticmethod
        def nrl_linear_workspace_bytes(n, k):
            return 64

    w = torch.zeros(8, 4)
    im = ops_blocks.FrozenImages()
    dev = torch.device("cpu")
    buf, ready, key = im.buffer("fwd", w, FakeLib, dev)
    assert ready == 0 and key is not None
    assert im.buffer("fwd", w, FakeLib, dev)[1] == 0, "an un-committed build must not count as ready (failed call)"
    im.commit("fwd", key)
    assert im.buffer("fwd", w, FakeLib, dev)[1] == 1
    w.add_(1.0)                                   # version counter moves
    _, ready, key = im.buffer("fwd", w, FakeLib, dev)
    assert ready == 0
    im.commit("fwd", key)
    im.invalidate()
    _, ready, key = im.buffer("fwd", w, FakeLib, dev)
    assert ready == 0
    im.commit("fwd", key)
    ops_blocks.invalidate_frozen_images()
    _, ready, key = im.buffer("fwd", w, FakeLib, dev)
    assert ready == 0
    im.commit("fwd", key)
    w.requires_grad_(True)                        # trained ...
    assert im.buffer("fwd", w, FakeLib, dev) == (None, 0, None)
    w.requires_grad_(False)                       # ... and frozen again: the old image must not be trusted
    assert im.buffer("fwd", w, FakeLib, dev)[1] == 0


def test_plm_body_shared_across_calls_of_different_lengths(tmp_path):
    """the reference pads each encoder call to its own longest text (rec_dataset.py:181), so history and
    candidate calls normally differ in sequence length.  ``PLM.share_body`` pads the shorter call (padding id, mask 0), runs ONE
    body pass and hands each call its own columns back: on the host body the hidden states are EXACTLY those of separate calls;
    hits and refusals are counted (a padding overhead above a quarter of the token rows is refused)."""
    from newsreclib_amd import news_encoder as ne
    from tests.helpers import PLM_HEADS, PLM_Q, make_tiny_roberta
    enc = ne.PLM(plm_model=make_tiny_roberta(str(tmp_path)), frozen_layers=[0], embed_dim=96, use_mhsa=True, apply_reduce_dim=False,
                 reduced_embed_dim=None, num_heads=PLM_HEADS, query_dim=PLM_Q, dropout_probability=0.2).eval()
    rng = np.random.default_rng(0)

    def toks(n, L, mask=True):
        ids = rng.integers(3, 200, (n, L))
        lens = rng.integers(3, L + 1, n)
        lens[0] = L
        m = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
        out = {"input_ids": torch.from_numpy(np.where(m == 1, ids, 1))}
        if mask:
            out["attention_mask"] = torch.from_numpy(m)
        return out

    ne.reset_fallback_calls()
    with torch.no_grad():
        a, b, c = toks(9, 20), toks(4, 17), toks(3, 20)
        ref = [enc.plm_model(**t)[0] for t in (a, b, c)]
        assert enc.share_body([a, b, c])
        assert ne.SHARE_BODY_CALLS == {"hit_same_length": 0, "hit_padded": 1, "miss": 0}
        for (ids, h), t, r in zip(enc._shared, (a, b, c), ref):
            assert ids is t["input_ids"] and h.shape == r.shape and torch.equal(h, r)
        # (forward() picks the rows up by the identity of input_ids -- on the GPU: tests/test_gpu_plm_full.py; the tail has no host path)
        # equal lengths: plain concatenation
        assert enc.share_body([a, c]) and ne.SHARE_BODY_CALLS["hit_same_length"] == 1
        # too much padding (4 x 5 x 15 extra rows > 180 + 25 real ones): refused, counted, nothing shared
        d = toks(5, 5)
        assert not enc.share_body([a, d]) and ne.SHARE_BODY_CALLS["miss"] == 1 and not enc._shared
        # texts without a mask attend to all of THEIR columns: the padded call must not attend to the extra ones
        e, f = toks(5, 12, mask=False), toks(2, 9, mask=False)
        ref = [enc.plm_model(**t)[0] for t in (e, f)]
        assert enc.share_body([e, f])
        for (ids, h), r in zip(enc._shared, ref):
            assert float((h - r).abs().max()) <= 1e-5
    os.environ["NRL_PLM_SHARE_BODY"] = "0"
    try:
        assert not enc.share_body([a, b])
    finally:
        del os.environ["NRL_PLM_SHARE_BODY"]
