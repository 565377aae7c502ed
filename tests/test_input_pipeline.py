"""Host input pipeline (newsreclib_amd/input_pipeline.py) against the loop/pandas restatement of the reference's
dataset + collate (oracle/input_oracle.py) on hand-made known-answer data.  CPU-only (the pipeline is index
plumbing on torch tensors; the GPU tier runs it on the device in tests/test_gpu_eval.py)."""
import numpy as np
import pandas as pd
import pytest
import torch

from newsreclib_amd import input_pipeline as IP
from oracle import input_oracle as IO

ATTRS = ["title", "abstract", "category", "sentiment_class"]


def make_frames(n_news=40, n_imp=23, seed=0):
    rng = np.random.default_rng(seed)
    nids = [f"N{int(i)}" for i in rng.permutation(np.arange(1000, 1000 + n_news))]
    news = pd.DataFrame({
        "tokenized_title": [list(rng.integers(1, 500, rng.integers(0, 9))) for _ in nids],     # some empty, some > max
        "tokenized_abstract": [list(rng.integers(1, 500, rng.integers(1, 14))) for _ in nids],
        "category_class": rng.integers(0, 18, n_news), "subcategory_class": rng.integers(0, 200, n_news),
        "sentiment_class": rng.integers(0, 3, n_news), "sentiment_score": rng.random(n_news).astype(np.float32),
    }, index=nids)
    rows = []
    for i in range(n_imp):
        nc = int(rng.integers(2, 12))
        labels = np.zeros(nc, dtype=np.int64)
        labels[rng.choice(nc, int(rng.integers(1, max(2, nc // 3 + 1))), replace=False)] = 1
        if labels.all():
            labels[0] = 0
        rows.append({"uid": f"U{int(rng.integers(1, 99999))}", "user": int(rng.integers(0, 50)),
                     "history": list(rng.choice(nids, int(rng.integers(1, 9)))),
                     "candidates": list(rng.choice(nids, nc, replace=False)), "labels": list(labels)})
    return news, pd.DataFrame(rows)


def assert_batches_equal(a, b):
    assert set(a) >= {"batch_hist", "batch_cand", "x_hist", "x_cand", "labels", "user_ids", "user_idx"}
    for k in ("batch_hist", "batch_cand", "labels", "user_ids", "user_idx"):
        assert a[k].dtype == b[k].dtype and torch.equal(a[k].cpu(), b[k]), k
    for part in ("x_hist", "x_cand"):
        assert set(a[part]) == set(b[part]), (part, set(a[part]), set(b[part]))
        for k in b[part]:
            assert a[part][k].dtype == b[part][k].dtype and torch.equal(a[part][k].cpu(), b[part][k]), (part, k)


def test_pad_token_lists_known_answers():
    text = [[5, 6, 7], [], [1, 2, 3, 4, 5, 6], [9]]
    want = np.array([[5, 6, 7, 0], [0, 0, 0, 0], [1, 2, 3, 4], [9, 0, 0, 0]])
    assert np.array_equal(IP.pad_token_lists(text, 4), want)                     # pads right, truncates at max_len
    assert np.array_equal(IP.pad_token_lists(text, None), IO.pad_tokens(text, None).numpy())   # longest row = 6
    assert IP.pad_token_lists(text, None).shape == (4, 6)
    assert IP.pad_token_lists([], 3).shape == (0, 3)
    rng = np.random.default_rng(1)
    text = [list(rng.integers(1, 99, rng.integers(0, 40))) for _ in range(200)]
    assert np.array_equal(IP.pad_token_lists(text, 30), IO.pad_tokens(text, 30).numpy())


@pytest.mark.parametrize("concat", [False, True])
def test_test_batches_match_collate(concat):
    news, bhv = make_frames()
    table, nid2row = IP.news_table_from_frame(news, ATTRS, 6, 10, concat, device="cpu")
    bt = IP.BehaviorTable.from_frame(bhv, nid2row, max_history_len=5, device="cpu")
    imps = np.array([3, 0, 22, 7, 7])
    got = IP.build_batch(table, bt, imps)
    want = IO.collate([IO.get_item(news, bhv, int(i), 5) for i in imps], ATTRS, 6, 10, concat)
    assert_batches_equal(got, want)
    assert int(got["batch_hist"].bincount().max()) <= 5                          # history truncated to the first 5


def test_train_batches_match_collate_for_the_same_picks():
    """Whatever the sampler drew, the batch must be what the reference's collate builds from those picks."""
    news, bhv = make_frames(seed=3)
    table, nid2row = IP.news_table_from_frame(news, ATTRS, 6, 10, False, device="cpu")
    bt = IP.BehaviorTable.from_frame(bhv, nid2row, max_history_len=50, device="cpu")
    imps = np.arange(len(bhv))
    gen = torch.Generator().manual_seed(5)
    got = IP.build_batch(table, bt, imps, neg_sampling_ratio=4, gen=gen)
    # recover the candidate POSITIONS the sampler picked from the news ids it returned (candidates are unique
    # within an impression in this data) and run the reference-shaped collate on them
    items, start = [], 0
    sizes = got["batch_cand"].bincount(minlength=len(bhv)).tolist()
    for i, sz in zip(imps, sizes):
        ids = got["x_cand"]["news_ids"][start:start + sz].tolist()
        cands = [int(c[1:]) for c in bhv.iloc[int(i)]["candidates"]]
        items.append(IO.get_item(news, bhv, int(i), 50, np.array([cands.index(x) for x in ids], dtype=np.int64)))
        start += sz
    assert_batches_equal(got, IO.collate(items, ATTRS, 6, 10, False))


def test_sampler_contract():
    news, bhv = make_frames(n_imp=60, seed=9)
    table, nid2row = IP.news_table_from_frame(news, ATTRS, 6, 10, False, device="cpu")
    bt = IP.BehaviorTable.from_frame(bhv, nid2row, 50, device="cpu")
    gen = torch.Generator().manual_seed(1)
    imps = np.arange(len(bhv))
    rows, labels, sizes = IP.sample_train_candidates(bt, imps, 4, gen)
    start = 0
    saw_replacement = False
    for i, sz in zip(imps, sizes):
        b = bhv.iloc[int(i)]
        lab = np.array(b["labels"])
        cand_rows = np.array([nid2row[c] for c in b["candidates"]])
        npos, nneg = int(lab.sum()), int((lab == 0).sum())
        got_rows, got_lab = rows[start:start + sz].numpy(), labels[start:start + sz].numpy()
        assert sz == 5 * npos and int(got_lab.sum()) == npos                    # every positive once + 4 negatives each
        assert sorted(got_rows[got_lab == 1]) == sorted(cand_rows[lab == 1])
        negs = got_rows[got_lab == 0]
        assert set(negs) <= set(cand_rows[lab == 0])
        if 4 * npos <= nneg:
            assert len(set(negs)) == len(negs)                                   # a k-subset: no repeats
        else:
            saw_replacement = True
        start += sz
    assert saw_replacement and start == len(rows)


def test_sampler_distribution_matches_reference_rule():
    """k-subsets and shuffle positions are uniform, as with np.random.choice(permutation(neg), k) + permutation."""
    labels = [0, 1, 0, 0, 0, 0]                                                   # 1 positive, 5 negatives, k = 2
    bt = IP.BehaviorTable(np.array([0, 1]), np.array([0]), np.array([0, 6]), np.arange(6), np.array(labels),
                          np.array([1]), np.array([0]), device="cpu")
    gen = torch.Generator().manual_seed(0)
    rng = np.random.default_rng(0)
    n = 6000
    big = IP.BehaviorTable(np.arange(n + 1), np.zeros(n, dtype=np.int64), np.arange(0, 6 * n + 1, 6),
                           np.tile(np.arange(6), n), np.tile(labels, n), np.ones(n, dtype=np.int64),
                           np.zeros(n, dtype=np.int64), device="cpu")
    rows, lab, sizes = IP.sample_train_candidates(big, np.arange(n), 2, gen)
    rows, lab = rows.reshape(n, 3).numpy(), lab.reshape(n, 3).numpy()
    assert (sizes == 3).all() and (lab.sum(1) == 1).all()
    pos_slot = np.bincount(lab.argmax(1), minlength=3) / n
    assert np.abs(pos_slot - 1 / 3).max() < 0.03                                  # shuffle: positive uniform over slots
    subsets = {}
    for r, l in zip(rows, lab):
        key = tuple(sorted(r[l == 0]))
        subsets[key] = subsets.get(key, 0) + 1
    assert len(subsets) == 10 and all(len(set(k)) == 2 for k in subsets)          # C(5, 2) subsets, no repeats
    freq = np.array(list(subsets.values())) / n
    assert np.abs(freq - 0.1).max() < 0.02
    ref = {}
    for _ in range(n):                                                            # the reference rule, same statistic
        idx = IO.sample_candidates(np.array(labels), 2, rng)
        key = tuple(sorted(i for i in idx if labels[i] == 0))
        ref[key] = ref.get(key, 0) + 1
    assert set(ref) == set(subsets)
    # with replacement (k = 4 > 3 negatives): i.i.d. uniform draws
    labels2 = [1, 0, 0, 0]
    big2 = IP.BehaviorTable(np.arange(n + 1), np.zeros(n, dtype=np.int64), np.arange(0, 4 * n + 1, 4),
                            np.tile(np.arange(4), n), np.tile(labels2, n), np.ones(n, dtype=np.int64),
                            np.zeros(n, dtype=np.int64), device="cpu")
    rows, lab, sizes = IP.sample_train_candidates(big2, np.arange(n), 4, gen)
    negs = rows[lab == 0].numpy()
    assert (sizes == 5).all() and np.abs(np.bincount(negs, minlength=4)[1:] / negs.size - 1 / 3).max() < 0.02
    assert (bt.npos == 1).all() and (bt.nneg == 5).all()


def test_sampler_raises_without_negatives():
    bt = IP.BehaviorTable(np.array([0, 1]), np.array([0]), np.array([0, 2]), np.arange(2), np.array([1, 1]),
                          np.array([1]), np.array([0]), device="cpu")
    with pytest.raises(ValueError):
        IP.sample_train_candidates(bt, np.array([0]), 4, torch.Generator().manual_seed(0))


@pytest.mark.parametrize("n,world,drop_last", [(23, 1, False), (23, 2, False), (23, 4, False), (23, 4, True), (3, 8, False)])
def test_epoch_indices_match_distributed_sampler(n, world, drop_last):
    from torch.utils.data import DistributedSampler
    ds = list(range(n))
    for epoch in (0, 3):
        for rank in range(world):
            if world == 1:
                g = torch.Generator()
                g.manual_seed(7 + epoch)
                want = torch.randperm(n, generator=g).tolist()
            else:
                smp = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=7, drop_last=drop_last)
                smp.set_epoch(epoch)
                want = list(iter(smp))
            got = IP.epoch_indices(n, True, 7, epoch, rank, world, drop_last).tolist()
            assert got == want, (epoch, rank)


def test_train_loader_epoch_is_reproducible_and_covers_every_impression():
    news, bhv = make_frames(n_imp=23, seed=4)
    table, nid2row = IP.news_table_from_frame(news, ATTRS, 6, 10, False, device="cpu")
    bt = IP.BehaviorTable.from_frame(bhv, nid2row, 50, device="cpu")
    seen = []
    for rank in range(2):
        ld = IP.TrainBatchLoader(table, bt, batch_size=4, neg_sampling_ratio=4, seed=3, rank=rank, world_size=2)
        ld.set_epoch(1)
        a = list(ld)
        b = list(ld)
        assert len(a) == len(ld) == 3
        for x, y in zip(a, b):
            assert_batches_equal(x, {k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items()})
                                     for k, v in y.items() if k != "batch_size"})
        seen += [int(u) for x in a for u in x["user_ids"]]
    uid = sorted(int(u[1:]) for u in bhv["uid"])
    assert sorted(set(seen)) == sorted(set(uid))                                  # 24 slots for 23 impressions: one repeats
    assert len(seen) == 24
