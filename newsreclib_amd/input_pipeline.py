"""Host input pipeline: the ``RecommendationBatch`` the reference builds per
step with pandas ``.loc`` + ``pd.concat`` + one ``F.pad`` per news on the host
(``data/components/rec_dataset.py:39-95,148-293``; ``num_workers: 0``, ``configs/data/mind_rec.yaml:66``) built
instead by INDEXING device-resident tables:

* ``news_table_from_frame`` pre-tokenises the news dataframe ONCE into a ``DeviceNewsTable`` (one padded row per
  unique news; the padding/truncation rule of ``_tokenize_embeddings``, ``rec_dataset.py:170-178``);
* ``BehaviorTable`` holds the click logs as CSR index arrays (history truncated to ``max_history_len``,
  ``rec_dataset.py:45``), row lengths also on the host so batch sizes never need a device->host sync;
* ``sample_train_candidates`` is the negative sampling of ``_sample_candidates`` (``rec_dataset.py:60-95``) for a
  whole batch at once on the device: same distribution (all positives; ``ratio * npos`` negatives, a uniform
  k-subset when the impression has enough of them, i.i.d. with replacement otherwise; one uniform shuffle), drawn
  from a seeded ``torch.Generator`` instead of numpy's global state;
* ``TrainBatchLoader`` / ``TestBatchLoader`` iterate an epoch the way ``DataLoader(shuffle=True)`` +
  Lightning's ``DistributedSampler`` do (same permutation and padding for a given seed/epoch/rank) and yield
  batches in the exact ``RecommendationBatch`` layout, ready for ``module.training_step``.

At ~10^4 impressions/s per GPU the reference loader (a few hundred impressions/s on one core) would starve the
kernels; this one is a handful of index kernels per batch.  Plumbing on torch tensors (works on "cpu" for the CPU
tests); nothing here touches the model arithmetic.
"""
from __future__ import annotations

from typing import Dict, Iterator, Optional, Sequence, Tuple

import numpy as np
import torch

from .evaluation import DeviceNewsTable


def pad_token_lists(text: Sequence[Sequence[int]], max_len: Optional[int]) -> np.ndarray:
    """(n, max_len) int64: right-padded with 0, rows longer than ``max_len`` truncated; ``max_len=None`` = the
    longest row (``_tokenize_embeddings``, rec_dataset.py:170-178)."""
    n = len(text)
    lens = np.fromiter((len(t) for t in text), dtype=np.int64, count=n)
    if max_len is None:
        max_len = int(lens.max()) if n else 0
    out = np.zeros((n, max_len), dtype=np.int64)
    if n == 0 or max_len == 0:
        return out
    keep = np.minimum(lens, max_len)
    flat = np.fromiter((tok for t, k in zip(text, keep) for tok in t[:k]), dtype=np.int64, count=int(keep.sum()))
    rows = np.repeat(np.arange(n), keep)
    cols = np.arange(int(keep.sum())) - np.repeat(np.cumsum(keep) - keep, keep)
    out[rows, cols] = flat
    return out


def _numeric_suffix(ids, letter: str) -> np.ndarray:
    return np.fromiter((int(str(s).split(letter)[-1]) for s in ids), dtype=np.int64, count=len(ids))


def news_table_from_frame(news, dataset_attributes: Sequence[str], max_title_len: int,
                          max_abstract_len: Optional[int] = None, concatenate_inputs: bool = False,
                          device="cuda") -> Tuple[DeviceNewsTable, Dict[str, int]]:
    """``news``: the reference's news dataframe (index = NID strings; columns ``tokenized_title``,
    ``tokenized_abstract``, ``category_class`` ... as ``_tokenize_df`` reads them, rec_dataset.py:184-287).
    Returns the device table (attribute names = the keys of ``x_hist`` / ``x_cand``) and the NID -> row map."""
    attrs = {"news_ids": torch.from_numpy(_numeric_suffix(news.index.values, "N"))}
    has_abstract = "abstract" in dataset_attributes
    if has_abstract and not (isinstance(max_abstract_len, int) and max_abstract_len > 0):
        raise AssertionError("max_abstract_len must be a positive int when abstracts are used")   # :137-139
    col = lambda name: news[name].values.tolist()  # noqa: E731
    if not concatenate_inputs:
        attrs["title"] = torch.from_numpy(pad_token_lists(col("tokenized_title"), max_title_len))
        if has_abstract:
            attrs["abstract"] = torch.from_numpy(pad_token_lists(col("tokenized_abstract"), max_abstract_len))
        if "title_entities" in dataset_attributes:
            attrs["title_entities"] = torch.from_numpy(pad_token_lists(col("title_entities"), max_title_len))
        if "abstract_entities" in dataset_attributes:
            attrs["abstract_entities"] = torch.from_numpy(pad_token_lists(col("abstract_entities"), max_abstract_len))
    else:
        if "title_entities" in dataset_attributes or "abstract_entities" in dataset_attributes:
            raise NotImplementedError("concatenated entity inputs are padded to the longest row OF EACH BATCH in the "
                                      "reference (rec_dataset.py:262); not built")
        if has_abstract:
            text = [[*a, *b] for a, b in zip(col("tokenized_title"), col("tokenized_abstract"))]
            attrs["text"] = torch.from_numpy(pad_token_lists(text, max_title_len + max_abstract_len))
        else:
            attrs["text"] = torch.from_numpy(pad_token_lists(col("tokenized_title"), max_title_len))
    attrs["category"] = torch.from_numpy(news["category_class"].values.astype(np.int64))
    attrs["subcategory"] = torch.from_numpy(news["subcategory_class"].values.astype(np.int64))
    if "sentiment_class" in dataset_attributes or "sentiment_score" in dataset_attributes:
        attrs["sentiment"] = torch.from_numpy(news["sentiment_class"].values.astype(np.int64))
        attrs["sentiment_score"] = torch.from_numpy(news["sentiment_score"].values.astype(np.float32))
    nid2row = {nid: i for i, nid in enumerate(news.index.values)}
    return DeviceNewsTable(attrs, device=device), nid2row


class BehaviorTable:
    """Click logs as CSR arrays of news-table rows.  ``hist_len`` / ``npos`` / ``nneg`` / ``ncand`` stay on the host
    (numpy) so that the sizes of a batch are known without reading anything back from the device."""

    def __init__(self, hist_ptr: np.ndarray, hist_rows: np.ndarray, cand_ptr: np.ndarray, cand_rows: np.ndarray,
                 cand_labels: np.ndarray, user_ids: np.ndarray, user_idx: np.ndarray, device="cuda"):
        self.n = len(user_ids)
        self.hist_len = np.diff(hist_ptr).astype(np.int64)
        self.ncand = np.diff(cand_ptr).astype(np.int64)
        lab = np.asarray(cand_labels)
        seg = np.repeat(np.arange(self.n), self.ncand)
        self.npos = np.bincount(seg, weights=(lab == 1), minlength=self.n).astype(np.int64)
        self.nneg = np.bincount(seg, weights=(lab == 0), minlength=self.n).astype(np.int64)
        self.device = torch.device(device)
        t = lambda a, dt=torch.int64: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(self.device)  # noqa: E731
        self.hist_ptr, self.hist_rows = t(hist_ptr), t(hist_rows)
        self.cand_ptr, self.cand_rows, self.cand_labels = t(cand_ptr), t(cand_rows), t(cand_labels)
        self.user_ids, self.user_idx = t(user_ids), t(user_idx)

    @classmethod
    def from_frame(cls, behaviors, nid2row: Dict[str, int], max_history_len: int, device="cuda") -> "BehaviorTable":
        """``behaviors``: the reference's dataframe (columns ``uid``, ``user``, ``history``, ``candidates``,
        ``labels``; rec_dataset.py:42-49)."""
        hists = [list(h)[:max_history_len] for h in behaviors["history"].values]
        cands = [list(c) for c in behaviors["candidates"].values]
        labs = [list(x) for x in behaviors["labels"].values]
        for c, x in zip(cands, labs):
            if len(c) != len(x):
                raise ValueError("every candidate needs a label")
        ptr = lambda ls: np.concatenate(([0], np.cumsum([len(x) for x in ls]))).astype(np.int64)  # noqa: E731
        rows = lambda ls: np.fromiter((nid2row[n] for x in ls for n in x), dtype=np.int64)  # noqa: E731  (KeyError = .loc's)
        return cls(ptr(hists), rows(hists), ptr(cands), rows(cands),
                   np.fromiter((v for x in labs for v in x), dtype=np.int64),
                   _numeric_suffix(behaviors["uid"].values, "U"), behaviors["user"].values.astype(np.int64), device)


def _h2d(arrays: Sequence[np.ndarray], device: torch.device):
    """int64 host arrays -> device tensors through ONE pinned staging buffer and ONE asynchronous copy.  (A plain
    ``.to(device)`` from pageable memory synchronises the stream, which would stop the host from building the next
    batch while the previous train step is still running.)"""
    arrays = [np.ascontiguousarray(a, dtype=np.int64).reshape(-1) for a in arrays]
    if device.type != "cuda":
        return [torch.from_numpy(a) for a in arrays]
    sizes = [a.size for a in arrays]
    host = torch.empty(sum(sizes), dtype=torch.int64, pin_memory=True)
    np.concatenate(arrays, out=host.numpy())
    return list(host.to(device, non_blocking=True).split(sizes))


def _segments(ptr_lo: torch.Tensor, sizes: torch.Tensor, total: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """flat positions ``ptr_lo[s] + j`` for j < sizes[s], and the segment number of each; ``total`` = sum(sizes)
    known on the host (no read-back)."""
    dev = sizes.device
    seg = torch.repeat_interleave(torch.arange(sizes.shape[0], device=dev), sizes, output_size=total)
    start = torch.cumsum(sizes, 0) - sizes
    within = torch.arange(total, device=dev) - start[seg]
    return ptr_lo[seg] + within, seg


def _sample(bt: BehaviorTable, cand_lo: torch.Tensor, ncand: torch.Tensor, npos: torch.Tensor, nneg: torch.Tensor,
            k: torch.Tensor, replace: torch.Tensor, tot_cand: int, tot_out: int, gen: torch.Generator) -> torch.Tensor:
    """positions (into ``bt.cand_rows`` / ``cand_labels``) of the sampled, shuffled candidates; no host sync."""
    dev = bt.device
    rnd = lambda m: torch.rand(m, generator=gen, device=dev, dtype=torch.float64)  # noqa: E731
    src, seg = _segments(cand_lo, ncand, tot_cand)
    is_neg = (bt.cand_labels[src] != 1).double()
    # one sort puts, inside every impression, the positives first and then its negatives in uniform random order
    src = src[torch.argsort(seg.double() * 2.0 + is_neg + 0.999 * rnd(tot_cand))]
    seg_start = torch.cumsum(ncand, 0) - ncand
    # output slot j of impression s: positive j (j < npos), else pool position j - npos (k-subset: the first k of
    # the permuted negatives) or floor(u * nneg) (with replacement: i.i.d. uniform)
    slot, out_seg = _segments(torch.zeros_like(npos), npos + k, tot_out)
    drawn = torch.minimum((rnd(tot_out) * nneg[out_seg].double()).long(), torch.clamp(nneg[out_seg] - 1, min=0))
    j_neg = slot - npos[out_seg]
    pool = torch.where(replace[out_seg] != 0, drawn, j_neg)
    pos_in_seg = torch.where(j_neg < 0, slot, npos[out_seg] + pool)
    picked = src[seg_start[out_seg] + pos_in_seg]
    return picked[torch.argsort(out_seg.double() + 0.999 * rnd(tot_out))]        # rec_dataset.py:89-90


def _sampling_sizes(bt: BehaviorTable, imps: np.ndarray, ratio: int):
    npos, nneg = bt.npos[imps], bt.nneg[imps]
    k = ratio * npos
    if np.any((nneg == 0) & (k > 0)):
        raise ValueError("an impression with clicks has no negative candidate to sample from")   # np.random.choice's
    return npos, nneg, k, (k > nneg).astype(np.int64)                            # rec_dataset.py:76-80


def sample_train_candidates(bt: BehaviorTable, imps: np.ndarray, neg_sampling_ratio: int,
                            gen: torch.Generator) -> Tuple[torch.Tensor, torch.Tensor, np.ndarray]:
    """-> (news rows, labels, per-impression sizes) of the sampled candidates of impressions ``imps``."""
    imps = np.asarray(imps, dtype=np.int64)
    npos, nneg, k, replace = _sampling_sizes(bt, imps, int(neg_sampling_ratio))
    ncand = bt.ncand[imps]
    d = _h2d([imps, ncand, npos, nneg, k, replace], bt.device)
    picked = _sample(bt, bt.cand_ptr[d[0]], d[1], d[2], d[3], d[4], d[5], int(ncand.sum()), int((npos + k).sum()), gen)
    return bt.cand_rows[picked], bt.cand_labels[picked].float(), npos + k


def build_batch(table: DeviceNewsTable, bt: BehaviorTable, imps: np.ndarray, neg_sampling_ratio: Optional[int] = None,
                gen: Optional[torch.Generator] = None) -> Dict:
    """The ``RecommendationBatch`` of impressions ``imps`` (``DatasetCollate.__call__``, rec_dataset.py:148-168):
    train (``neg_sampling_ratio`` given) or validation/test (all candidates, in order).  Every size comes from the
    host-side row lengths, so nothing is read back from the device; the ragged-layout metadata
    ``nrms_module.prepare_batch`` would otherwise derive (offsets, max sizes) is attached for free."""
    dev = bt.device
    imps = np.asarray(imps, dtype=np.int64)
    B = len(imps)
    hist_len, ncand = bt.hist_len[imps], bt.ncand[imps]
    train = neg_sampling_ratio is not None
    if train:
        if gen is None:
            raise ValueError("training batches need a torch.Generator")
        npos, nneg, k, replace = _sampling_sizes(bt, imps, int(neg_sampling_ratio))
        cand_sizes = npos + k
    else:
        npos = nneg = k = replace = np.zeros(0, dtype=np.int64)
        cand_sizes = ncand
    offs = lambda sz: np.concatenate(([0], np.cumsum(sz)))  # noqa: E731
    d = _h2d([imps, hist_len, offs(hist_len), ncand, cand_sizes, offs(cand_sizes), npos, nneg, k, replace], dev)
    imps_d, hist_len_d, hist_off, ncand_d, cand_sizes_d, cand_off = d[:6]
    n_hist, n_cand = int(hist_len.sum()), int(cand_sizes.sum())
    hist_src, batch_hist = _segments(bt.hist_ptr[imps_d], hist_len_d, n_hist)
    if train:
        cand_src = _sample(bt, bt.cand_ptr[imps_d], ncand_d, d[6], d[7], d[8], d[9], int(ncand.sum()), n_cand, gen)
        batch_cand = torch.repeat_interleave(torch.arange(B, device=dev), cand_sizes_d, output_size=n_cand)
    else:
        cand_src, batch_cand = _segments(bt.cand_ptr[imps_d], ncand_d, n_cand)
    x_hist, x_cand = table.gather(bt.hist_rows[hist_src]), table.gather(bt.cand_rows[cand_src])
    if "news_ids" not in table.attrs:
        x_cand["news_ids"] = bt.cand_rows[cand_src]
    return {
        "x_hist": x_hist, "x_cand": x_cand, "batch_hist": batch_hist, "batch_cand": batch_cand,
        "labels": bt.cand_labels[cand_src].float(), "user_idx": bt.user_idx[imps_d], "user_ids": bt.user_ids[imps_d],
        "batch_size": B,
        # prepare_batch metadata, known here without touching the device
        "hist_offsets": hist_off, "cand_offsets": cand_off, "hist_sizes": hist_len_d, "cand_sizes": cand_sizes_d,
        "max_hist": int(hist_len.max()) if B else 0, "max_cand": int(cand_sizes.max()) if B else 0,
        "min_hist": int(hist_len.min()) if B else 0,
    }


def epoch_indices(n: int, shuffle: bool, seed: int, epoch: int, rank: int = 0, world_size: int = 1,
                  drop_last: bool = False) -> np.ndarray:
    """The impressions rank ``rank`` visits in epoch ``epoch``, in order.  world_size > 1 follows
    ``torch.utils.data.DistributedSampler`` (what Lightning wraps the reference's loaders in): permutation from
    ``Generator().manual_seed(seed + epoch)``, padded by wrapping around (or truncated, ``drop_last``) to a multiple
    of the world size, then strided ``rank::world_size``."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).numpy()
    else:
        idx = np.arange(n)
    if world_size == 1:
        return idx
    if drop_last and n % world_size != 0:
        total = (n - world_size + world_size - 1) // world_size * world_size   # ceil((n - w) / w) * w
        idx = idx[:total]
    else:
        total = (n + world_size - 1) // world_size * world_size
        pad = total - n
        if pad:
            reps = (pad + n - 1) // n
            idx = np.concatenate([idx, np.tile(idx, reps)[:pad]])
    return idx[rank:total:world_size]


def _tensors(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors(v)


class TrainBatchLoader:
    """``DataLoader(RecommendationDatasetTrain, batch_size, shuffle=True, collate_fn=DatasetCollate)``
    (mind_rec_datamodule.py:313-326) on the device tables.

    On a GPU the next batch is assembled (and run through ``nrms_module.prepare_batch``) on a SIDE HIP stream while
    the consumer's train step for the current one is still executing: the few dozen index kernels of a batch are
    latency-, not throughput-bound, so they slot in beside the step's kernels instead of queueing behind them.
    The consumer's stream waits on the batch's event before it is handed out."""

    def __init__(self, table: DeviceNewsTable, behaviors: BehaviorTable, batch_size: int, neg_sampling_ratio: int,
                 shuffle: bool = True, seed: int = 0, rank: int = 0, world_size: int = 1, drop_last: bool = False,
                 prefetch: bool = True):
        self.table, self.bt, self.batch_size = table, behaviors, int(batch_size)
        self.ratio, self.shuffle, self.seed = int(neg_sampling_ratio), shuffle, int(seed)
        self.rank, self.world, self.drop_last, self.epoch = int(rank), int(world_size), drop_last, 0
        self.gen = torch.Generator(device=behaviors.device)
        self.prefetch = bool(prefetch) and behaviors.device.type == "cuda"
        self._side = torch.cuda.Stream(device=behaviors.device) if self.prefetch else None

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def _indices(self) -> np.ndarray:
        return epoch_indices(self.bt.n, self.shuffle, self.seed, self.epoch, self.rank, self.world)

    def __len__(self) -> int:
        n = len(self._indices())
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _build(self, imps: np.ndarray) -> Dict:
        return build_batch(self.table, self.bt, imps, self.ratio, self.gen)

    def _batches(self, idx: np.ndarray) -> Iterator[Dict]:
        slices = (idx[b * self.batch_size:(b + 1) * self.batch_size] for b in range(len(self)))
        if not self.prefetch:
            for sl in slices:
                yield self._build(sl)
            return
        from .nrms_module import prepare_batch
        main = torch.cuda.current_stream(self.bt.device)

        def release(item):
            batch, event = item
            main.wait_event(event)
            for t in _tensors(batch):            # allocated on the side stream, consumed (and freed) on the main one
                t.record_stream(main)
            return batch

        pending = None
        for sl in slices:
            with torch.cuda.stream(self._side):
                item = (prepare_batch(self._build(sl)), self._side.record_event())
            if pending is not None:
                yield release(pending)
            pending = item
        if pending is not None:
            yield release(pending)

    def __iter__(self) -> Iterator[Dict]:
        # the sampling stream is a function of (seed, epoch, rank): reruns reproduce the same batches
        self.gen.manual_seed((self.seed * 1000003 + self.epoch) * 65537 + self.rank)
        return self._batches(self._indices())


class TestBatchLoader(TrainBatchLoader):
    """Validation / test loader (mind_rec_datamodule.py:331-362): every candidate, log order, no shuffle."""

    def __init__(self, table: DeviceNewsTable, behaviors: BehaviorTable, batch_size: int, rank: int = 0,
                 world_size: int = 1, prefetch: bool = True):
        super().__init__(table, behaviors, batch_size, 0, shuffle=False, rank=rank, world_size=world_size,
                         prefetch=prefetch)

    def _build(self, imps: np.ndarray) -> Dict:
        return build_batch(self.table, self.bt, imps)
