#!/usr/bin/env python3
"""bench.py -- NRMS train-step throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full train step of the hot path over one synthetic MIND-shaped batch that is
already resident in HBM: forward (history + candidate news encode, user encode, score), CE loss,
backward, (N > 1: gradient exchange over RCCL), Adam with the reference's dense semantics -- dropout active, fp32.
The optimizer is `trainer.LazyTableAdam` + the dense kernel for the 0.84 M non-table parameters: rows of the embedding
table with a zero gradient are advanced lazily (bit-identical to dense Adam after a flush; a rolling flush bounds every
row's lag by 64 steps), so a timed region ends with rows whose update is pending -- `config.optimizer` says so.
Workload at every N (weak scaling): BASELINE.json configs[1], B = 128 impressions per GPU,
H = 50 clicks, C = 5 candidates, L = 30 tokens, V = 70,000, D = 300, 15 heads, Q = 200.
Rank 0 prints ONE JSON line.

`--gpus N` with N > 1 and no launcher environment (WORLD_SIZE unset) re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU
over RCCL), so the same command line works with and without an external launcher.
"""
import argparse
import ctypes
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU, VOCAB, L, H, C, D, HEADS, Q, P_DROP, LR = 128, 70_000, 30, 50, 5, 300, 15, 200, 0.2, 1e-4
FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: dense fp32 MFMA (= vector) peak
BF16_MFMA_PEAK_TFLOPS = 2500.0         # dense bf16 MFMA peak (no sparsity)
HBM_PEAK_GBPS = 8000.0                 # HBM3E spec (6.3 TB/s achievable)
M_ROWS = B_PER_GPU * (H + C) * L
N_BATCHES = 4                          # distinct pre-generated batches cycled through


def build_module(device):
    from functools import partial

    from newsreclib_amd.nrms_module import NRMSModule
    torch.manual_seed(42)              # configs/experiment/nrms_mindsmall_pretrainedemb_*.yaml:18
    emb = torch.randn(VOCAB, D)        # N(0,1) table mirrors data_utils.py:56
    mod = NRMSModule(
        dataset_attributes=["title", "abstract", "category"], attributes2encode=["title"],
        outputs={"train": [], "val": [], "test": []},
        dual_loss_training=False, dual_loss_coef=None, loss="cross_entropy_loss", late_fusion=False,
        temperature=None, use_plm=False, pretrained_embeddings_path=None, plm_model=None, frozen_layers=None,
        embed_dim=D, num_heads=HEADS, query_dim=Q, dropout_probability=P_DROP, top_k_list=[5, 10],
        num_categ_classes=18, num_sent_classes=3, save_recs=False, recs_fpath=None,
        optimizer=partial(torch.optim.Adam, lr=LR), scheduler=None, pretrained_embeddings=emb)
    return mod.to(device)


def cpu_baseline(time_budget_s=20.0):
    """The reference's CPU path, restated (oracle.TorchGraphNRMS: same nn graph as the reference,
    proven equal to it in tests/test_oracle_golden.py), timed on this box's host cores on a bounded
    sample of the SAME workload: B=128 train steps (fwd + bwd + torch.optim.Adam, dropout on).
    SURVEY.md section 8(d) asks for two thread counts: one socket's physical cores (the headline leg: `value`, `cores`)
    and n = 8 (`legs`)."""
    from newsreclib_amd.synthetic import make_batch
    from oracle.nrms_oracle import TorchGraphNRMS          # checker/baseline leg only
    socket_cores = max(1, min(64, (os.cpu_count() or 2) // 2))    # one socket's physical cores, capped
    batch = make_batch(B_PER_GPU, VOCAB, "fixed", seed=1234)

    def leg(cores, budget):
        torch.set_num_threads(cores)
        torch.manual_seed(42)
        model = TorchGraphNRMS(torch.randn(VOCAB, D), D, HEADS, Q, P_DROP).train()
        opt = torch.optim.Adam(model.parameters(), lr=LR)

        def step():
            opt.zero_grad()
            loss = model.loss(batch)
            loss.backward()
            opt.step()

        step()                                                 # warm-up
        t0, n = time.perf_counter(), 0
        while n < 2 or (time.perf_counter() - t0 < budget and n < 50):
            step()
            n += 1
        dt = time.perf_counter() - t0
        return {"value": round(B_PER_GPU * n / dt, 2), "unit": "impressions/s", "cores": cores,
                "sample": f"{n} train steps of the same B=128 workload ({dt:.1f} s), torch {torch.__version__} "
                          f"CPU, {cores} threads"}

    head = leg(socket_cores, time_budget_s)
    legs = [dict(head)]
    if socket_cores != 8 and (os.cpu_count() or 1) >= 8:
        legs.append(leg(8, 0.6 * time_budget_s))
    cpu = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {**head, "kind": "port", "cpu_model": cpu, "logical_cpus": os.cpu_count(), "legs": legs}


def _timed_steps(trainer, batch, steps, warmup):
    """-> (median, min, max) seconds per step over `steps` steps after `warmup` untimed ones; every step between its own pair of
    events on the launch stream (one ~1 us host call each), one synchronize at the end."""
    for _ in range(warmup):
        trainer.step(batch)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    torch.cuda.synchronize()
    marks[0].record()
    for i in range(steps):
        trainer.step(batch)
        marks[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    return statistics.median(ms) * 1e-3, ms[0] * 1e-3, ms[-1] * 1e-3


def extra_lstur(device, batch_size=128, steps=15):   # (median of 15 timed steps after 3 untimed)
    """BASELINE.json configs[4]: LSTUR (CNN news encoder, title 30 + abstract 50 tokens, 300 filters, window 3, category
    embedding 100, GRU 700 user encoder, 45,215 users) train step under the same click_predictor API, B = 128."""
    from functools import partial

    from newsreclib_amd.lstur_module import LSTURModule
    from newsreclib_amd.nrms_module import attach_layout
    from newsreclib_amd.synthetic import add_lstur_fields, make_batch
    from newsreclib_amd.trainer import NRMSTrainer
    torch.manual_seed(0)
    mod = LSTURModule(
        dataset_attributes=["title", "abstract", "category"], attributes2encode=["title", "abstract", "category"],
        outputs={"train": [], "val": [], "test": []}, dual_loss_training=False, dual_loss_coef=None,
        loss="cross_entropy_loss", late_fusion=False, temperature=None, use_plm=False,
        pretrained_embeddings_path=None, plm_model=None, frozen_layers=None, text_embed_dim=300, num_heads=15,
        num_filters=300, window_size=3, query_dim=200, categ_embed_dim=100, dropout_probability=0.2,
        num_users=45214, user_masking_probability=0.5, long_short_term_method="ini", top_k_list=[5, 10],
        num_categ_classes=18, num_sent_classes=3, save_recs=False, recs_fpath=None,
        optimizer=partial(torch.optim.Adam, lr=LR), scheduler=None,
        pretrained_embeddings=torch.randn(VOCAB, 300) * 0.3).to(device)
    trainer = NRMSTrainer(mod, lr=LR)
    batch = attach_layout(add_lstur_fields(make_batch(batch_size, VOCAB, "fixed", seed=1234, device=device), VOCAB))
    dt, dt_min, dt_max = _timed_steps(trainer, batch, steps, 3)
    # algorithmic FLOPs (SURVEY.md Appendix A, config-5 extras): 52.8 MFLOP per news forward (title 16.2 + 3.6, abstract 27 + 6),
    # 55 news + <= 294 MFLOP of GRU per impression, train step = 3 x forward
    flops = 3.0 * (55 * 52.8e6 + 294e6) * batch_size
    return {"value": round(batch_size / dt, 1), "unit": "impressions/s", "ms_per_step": round(dt * 1e3, 3),
            "basis": "median of %d steps" % steps, "min_max_ms": [round(dt_min * 1e3, 3), round(dt_max * 1e3, 3)],
            "config": "LSTUR MINDsmall-shaped train step, B=128, title 30 + abstract 50 tokens, GRU 700 (BASELINE.json configs[4])",
            "roofline": {"bound": "mfma", "scope": "whole step (not one kernel)", "algorithmic_flops_per_step": flops,
                         "achieved": round(3 * flops / dt / 1e12, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(3 * flops / dt / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
                         "note": "issued bf16 FLOPs (3 bf16 products per fp32 product) of the whole step against the dense bf16 "
                                 "MFMA peak; timed steps: %d" % steps}}


def extra_plm(device, batch_size=8, steps=7):
    """BASELINE.json configs[3]: NRMS with the PLM news encoder (roberta-base SHAPE, random init -- no network for the
    checkpoint; d = 768, 16 heads, L = 96, layers 0-7 frozen), B = 8 as in the reference's experiment file.  The
    transformer body is HF's module graph on PyTorch-ROCm (attention, layer norms, GELU: third-party) with its nn.Linear
    projections swapped for this library's GEMM engine (news_encoder.swap_linears; NRL_PLM_LINEAR=0 keeps hipBLASLt:
    205 -> 117 ms per step); the encoder tail, user encoder, scorer, loss and Adam are this library's."""
    import tempfile
    from functools import partial

    from transformers import RobertaConfig, RobertaModel

    from newsreclib_amd.nrms_module import NRMSModule, prepare_batch
    from newsreclib_amd.synthetic import make_batch
    from newsreclib_amd.trainer import NRMSTrainer
    torch.manual_seed(0)
    cfg = RobertaConfig(vocab_size=50265, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                        intermediate_size=3072, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1,
                        bos_token_id=0, eos_token_id=2)
    tmp = tempfile.mkdtemp()
    RobertaModel(cfg, add_pooling_layer=False).save_pretrained(tmp)
    mod = NRMSModule(
        dataset_attributes=["title", "abstract", "category"], attributes2encode=["title"],
        outputs={"train": [], "val": [], "test": []}, dual_loss_training=False, dual_loss_coef=None,
        loss="cross_entropy_loss", late_fusion=False, temperature=None, use_plm=True, pretrained_embeddings_path=None,
        plm_model=tmp, frozen_layers=list(range(8)), embed_dim=768, num_heads=16, query_dim=200,
        dropout_probability=0.2, top_k_list=[5, 10], num_categ_classes=18, num_sent_classes=3, save_recs=False,
        recs_fpath=None, optimizer=partial(torch.optim.Adam, lr=1e-5), scheduler=None).to(device)
    trainer = NRMSTrainer(mod, lr=1e-5)
    b = make_batch(batch_size, vocab=50000, mode="fixed", seed=1, L=96, device=device)
    for part in ("x_hist", "x_cand"):          # tokenizer-style inputs (rec_dataset.py:180-190)
        ids = b[part]["title"].clamp_min(3)
        b[part]["title"] = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
    from newsreclib_amd import news_encoder as _ne
    pb = prepare_batch(b)
    _timed_steps(trainer, pb, 1, 1)            # warm-up (first-call mask check of the body attention, image builds)
    _ne.reset_fallback_calls()
    dt, dt_min, dt_max = _timed_steps(trainer, pb, steps, 0)
    share_calls = dict(_ne.SHARE_BODY_CALLS)
    fallbacks = dict(_ne.FALLBACK_CALLS)
    # the same step with each encoder call running its own body pass (NRL_PLM_SHARE_BODY=0), and on a batch whose two calls
    # differ in sequence length, as a collate that pads each call to its own longest text produces them (rec_dataset.py:181;
    # ADVICE round 5): history 96 tokens, candidates 80 -- one body pass over the padded union vs two passes
    variants = {}
    b2 = make_batch(batch_size, vocab=50000, mode="fixed", seed=1, L=96, device=device)
    for part, Lp_ in (("x_hist", 96), ("x_cand", 80)):
        ids = b2[part]["title"].clamp_min(3)[:, :Lp_].contiguous()
        b2[part]["title"] = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
    pb2 = prepare_batch(b2)
    prev = os.environ.get("NRL_PLM_SHARE_BODY")
    try:
        for name, batch_v, share in (("two_body_passes", pb, "0"), ("cand_80_tokens_one_padded_pass", pb2, "1"),
                                     ("cand_80_tokens_two_passes", pb2, "0")):
            os.environ["NRL_PLM_SHARE_BODY"] = share
            _ne.reset_fallback_calls()
            v, _, _ = _timed_steps(trainer, batch_v, 3, 1)
            variants[name] = {"ms_per_step": round(v * 1e3, 1), "share_body_calls": dict(_ne.SHARE_BODY_CALLS)}
    finally:
        if prev is None:
            os.environ.pop("NRL_PLM_SHARE_BODY", None)
        else:
            os.environ["NRL_PLM_SHARE_BODY"] = prev
    # algorithmic FLOPs: body forward 2 x (4 d^2 + 2 d f) per token and layer + 4 L d per token and layer of attention;
    # dgrad through all 12 layers (the embeddings train, text.py:70-73), weight gradients for the 4 unfrozen layers;
    # tail (seq-first MHA across the news of a call + additive attention, d = 768) forward x 3
    tokens, d, f, layers, Lp = batch_size * 55 * 96, 768, 3072, 12, 96
    body_fwd = tokens * layers * (2 * (4 * d * d + 2 * d * f) + 4 * Lp * d)
    n_hist, n_cand = batch_size * 50, batch_size * 5
    tail_fwd = tokens * 2 * d * (3 * d + d + 200) + 4 * 48 * 16 * Lp * (n_hist ** 2 + n_cand ** 2)
    flops = body_fwd * (2.0 + 4.0 / 12.0) + 3.0 * tail_fwd
    return {"value": round(batch_size / dt, 2), "unit": "impressions/s", "ms_per_step": round(dt * 1e3, 1),
            "basis": "median of %d steps; one body pass over both encoder calls (equal sequence lengths)" % steps,
            "min_max_ms": [round(dt_min * 1e3, 1), round(dt_max * 1e3, 1)], "share_body_calls": share_calls, "variants": variants,
            "roofline": {"bound": "mfma", "scope": "whole step (not one kernel)", "algorithmic_flops_per_step": flops,
                         "achieved": round(3 * flops / dt / 1e12, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(3 * flops / dt / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
                         "note": "issued bf16 FLOPs (3 per fp32 product) against the dense bf16 MFMA peak, i.e. fp32-equivalent "
                                 "FLOPs against 833 TF; timed steps: %d" % steps},
            "framework_fallback_calls": fallbacks,
            "config": "NRMS-PLM train step, roberta-base-shaped random body (HF module graph; its 72 projections on this library's "
                      "GEMM engine via news_encoder.NrlLinear, its self-attention on nrl_sdpa_fwd / _bwd via the HF attention "
                      "registry), d=768, 16 heads, L=96, B=8 (BASELINE.json configs[3])",
            "body_linears_on_this_library": int(mod.news_encoder.text_encoders["title"].nrl_linears),
            "body_attention_on_this_library": bool(mod.news_encoder.text_encoders["title"].nrl_attention),
            "body_output_blocks_on_this_library": int(mod.news_encoder.text_encoders["title"].nrl_output_blocks),
            "body_embedding_tables_on_this_library": int(getattr(mod.news_encoder.text_encoders["title"], "nrl_embeddings", 0)),
            "body_ffn_blocks_as_one_function": int(getattr(mod.news_encoder.text_encoders["title"], "nrl_ffn_blocks", 0)),
            "body_attention_blocks_as_one_function": int(getattr(mod.news_encoder.text_encoders["title"], "nrl_attention_blocks", 0))}


def predict_multi_gpu(step_ms: float, world: int = 8):
    """UNMEASURED (no multi-GPU node has been available in any round): what `world` ranks of this workload would take per step
    under trainer.predicted_wire_ms -- compute = this run's one-GPU step; the exchange overlaps the weight-gradient phase of the
    news-encoder backward (22 % of the step in profiles/r04_x3_kernel_stats.csv) and its exposed rest adds to the step.  The
    unique / union row counts are those of the `world` synthetic rank batches (seeds 1234 + rank)."""
    from newsreclib_amd.synthetic import make_batch
    from newsreclib_amd.trainer import predicted_wire_ms
    ids = [torch.cat([b["x_hist"]["title"].reshape(-1), b["x_cand"]["title"].reshape(-1)])
           for b in (make_batch(B_PER_GPU, VOCAB, "fixed", seed=1234 + r) for r in range(world))]
    uniq = [int(torch.unique(i).numel()) for i in ids]
    union = torch.unique(torch.cat(ids))
    owners = [int((union % world == r).sum()) for r in range(world)]
    row_b, head = 4 * D, 4.0 * VOCAB * D
    wire = {"dense": predicted_wire_ms("dense", head + 4 * 843_200, world),
            "rows": predicted_wire_ms("rows", max(uniq) * (row_b + 8), world),
            "owners": predicted_wire_ms("owners", max(uniq) * row_b, world, max(owners) * row_b)}
    overlap = 0.22 * step_ms
    out = {"world": world, "unmeasured": True, "compute_ms": round(step_ms, 4), "overlap_window_ms": round(overlap, 4),
           "unique_rows_per_rank_max": max(uniq), "union_rows": int(union.numel()), "predicted_wire_ms": wire, "modes": {}}
    for mode, w in wire.items():
        for model in ("ring", "direct"):
            step = step_ms + max(0.0, w[model] - overlap) + (0.03 if mode != "dense" else 0.0)
            out["modes"][f"{mode}/{model}"] = {"step_ms": round(step, 4), "speedup_over_1gpu": round(world * step_ms / step, 2)}
    return out


def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` without a launcher: spawn the N ranks ourselves (reference multi-GPU leg:
    configs/trainer/ddp.yaml:4 `strategy: ddp`, one process per device)."""
    have = torch.cuda.device_count()
    if have < n_gpus:
        raise SystemExit(f"bench.py: --gpus {n_gpus} needs {n_gpus} visible GPUs, this node has {have}")
    with socket.socket() as sock:                       # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def git_head():
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return None


WORKLOADS = {
    # BASELINE.json configs[1] (the configuration `metric` is quoted on) and configs[2]'s per-rank shape
    "mindsmall": {"batch": 128, "vocab": 70_000,
                  "name": "NRMS pretrained-emb (d=300, 15 heads, Q=200) MINDsmall-shaped train step: B=128/GPU, H=50, C=5, "
                          "L=30, V=70000, dropout 0.2, Adam lr 1e-4 (BASELINE.json configs[1])"},
    "mind32": {"batch": 32, "vocab": 70_000,
               "name": "NRMS pretrained-emb (d=300, 15 heads, Q=200) train step at the reference's own CPU-runnable size: B=32, "
                       "H=50, C=5, L=30, V=70000, dropout 0.2, Adam lr 1e-4 (BASELINE.json configs[0])"},
    "mindlarge": {"batch": 64, "vocab": 150_000,
                  "name": "NRMS pretrained-emb (d=300, 15 heads, Q=200) MINDlarge-shaped train step: B=64/GPU (512 global on "
                          "8 GPUs), H=50, C=5, L=30, V=150000, dropout 0.2, Adam lr 1e-4 (BASELINE.json configs[2])"},
}


def main():
    global B_PER_GPU, VOCAB, M_ROWS
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed region (profiling runs): skip forward_only / f32_engine / cpu_baseline")
    ap.add_argument("--engine", choices=["f32", "bf16x3"], default="bf16x3",
                    help="projection-GEMM engine: exact fp32 MFMA, or fp32 via 3 bf16 MFMAs per product")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="mindsmall",
                    help="mindsmall = BASELINE.json configs[1] (headline); mind32 = configs[0] (B=32); mindlarge = configs[2]'s per-rank shape (V=150k, B=64/GPU)")
    ap.add_argument("--grad-exchange", choices=["dense", "rows", "owners", "auto"], default=None,
                    help="N > 1: dense = all-reduce of the whole flat gradient; rows = all-gather of the touched table rows + "
                         "dense rest; owners = touched rows reduced by their owner rank (id %% world), then all-gathered; "
                         "auto = per step whichever of the three trainer.predicted_wire_ms prices lowest (dense stays the default: it is the "
                         "one exchange whose RCCL calls have run on hardware, at world size 1)")
    args = ap.parse_args()
    if args.grad_exchange is None:
        # dense all-reduce for every workload: it is the one exchange whose RCCL calls have run on hardware (at world size 1; no
        # multi-GPU node was available to any round).  `--grad-exchange auto` prices the three exchanges per step
        # (trainer.predicted_wire_ms) and would pick the owner-partitioned row exchange for configs[2] (183 MB of dense gradient at
        # B = 64 per GPU) -- whose all-to-all on a private stream has never run at world > 1 (round-5 advisor): opt-in until it has
        args.grad_exchange = "dense"
    B_PER_GPU, VOCAB = WORKLOADS[args.workload]["batch"], WORKLOADS[args.workload]["vocab"]
    M_ROWS = B_PER_GPU * (H + C) * L

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    distributed = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ   # launched by torch.distributed.run
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", device_id=device)   # RCCL over xGMI

    from newsreclib_amd import _lib
    from newsreclib_amd.nrms_module import attach_layout, prepare_batch
    from newsreclib_amd.synthetic import make_batch
    from newsreclib_amd.trainer import NRMSTrainer
    lib = _lib.load()
    _lib.set_gemm_engine(args.engine)

    mod = build_module(device)
    trainer = NRMSTrainer(mod, lr=LR, grad_exchange=args.grad_exchange)
    # impressions shard on the user axis: rank r owns its own B_PER_GPU impressions (seed by rank)
    # what a collate function hands over: the RecommendationBatch tensors in HBM plus the row-length metadata it
    # has on the host anyway (offsets, max sizes).  The per-step device work on the ids -- concatenating history
    # and candidate ids and the argsort the embedding gradient needs -- happens INSIDE every timed step.
    batches = [attach_layout(make_batch(B_PER_GPU, VOCAB, "fixed", seed=1234 + 1000 * i + rank, device=device))
               for i in range(N_BATCHES)]
    assert all("x_all" not in b for b in batches)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the timed region: EXACTLY `steps` steps.  No profiling hook inside (nrl_prof stays off); the only additions are the
    # per-step event records the median is computed from (one ~1 us host call each, nothing on the device) ----
    # the loop knows its next batch (a prefetching loader does): the trainer runs that step's id bookkeeping -- id concatenation,
    # counting sort, lazy-optimizer marks + catch-up -- on its side stream beside the current step (trainer._prefetch)
    for i in range(args.warmup):
        trainer.step(batches[i % N_BATCHES], batches[(i + 1) % N_BATCHES])
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        trainer.step(batches[(args.warmup + i) % N_BATCHES], batches[(args.warmup + i + 1) % N_BATCHES])
        marks[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]

    # ---- second, instrumented pass (NOT the headline, feeds `roofline` only): HIP events around the dominant kernel
    # (nrl_prof, recorded on the launch stream) ----
    lib.nrl_prof_enable(1)
    if hasattr(trainer.reduce, "measure"):
        # (N > 1: event pairs around every wait for a collective -- how much of the exchange the backward did not hide; zeros at
        #  world 1.  In this pass only: the timed region above carries no such instrumentation)
        trainer.reduce.reset_exposed_wait()
        trainer.reduce.measure = True
    for i in range(args.steps):
        trainer.step(batches[i % N_BATCHES], batches[(i + 1) % N_BATCHES])
    barrier()
    if hasattr(trainer.reduce, "measure"):
        trainer.reduce.measure = False
    tot_ms, launches, flops = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
    lib.nrl_prof_read(ctypes.byref(tot_ms), ctypes.byref(launches), ctypes.byref(flops))
    lib.nrl_prof_enable(0)
    # what the id bookkeeping inside the step costs (measured after the timed region)
    t1 = time.perf_counter()
    for i in range(20):
        prepare_batch(batches[i % N_BATCHES], VOCAB)
    torch.cuda.synchronize()
    prepare_ms = (time.perf_counter() - t1) / 20 * 1e3

    median_ms = statistics.median(step_ms)
    if distributed:
        tmax = torch.tensor([dt, median_ms], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt, median_ms = float(tmax[0]), float(tmax[1])

    if rank == 0:
        # the metric is B / median step time over >= 50 steps; with fewer steps the mean of the
        # timed region stands in (both are printed either way)
        mean_value = world * B_PER_GPU * args.steps / dt
        median_value = world * B_PER_GPU / (median_ms * 1e-3)
        value = median_value if args.steps >= 50 else mean_value
        avg_s = tot_ms.value * 1e-3 / max(1, launches.value)
        tflops = (flops.value / max(1, launches.value)) / avg_s / 1e12 if avg_s > 0 else None
        # The dominant kernel of the step and its ALGORITHMIC work (no materialised
        # intermediates): per launch it needs the ids (8 B per token) and the gathered embedding rows (1200 B per
        # token) -- 0.255 GB at B = 128 -- and 2*M*3D*D (+ the per-head L x L attention it now contains) FLOPs.
        algo_bytes = M_ROWS * 8 + M_ROWS * D * 4
        fused = args.engine == "bf16x3" and os.environ.get("NRL_NEWS_FUSED", "1") != "0"
        pmc_name = "pmc_news_fused_fwd_bf16x3.json" if fused else f"pmc_in_proj_fwd_{args.engine}.json"
        traffic, step_bytes, prof_meta = None, None, None
        pmc = os.path.join(ROOT, "profiles", pmc_name)
        build_id = lib.nrl_build_id().decode()
        if os.path.exists(pmc) and args.workload == "mindsmall":
            j = json.load(open(pmc))
            traffic, step_bytes = j.get("hbm_bytes_per_launch"), j.get("hbm_bytes_per_step")
            # the counters are a committed profile (rocprofv3 PMC passes cannot run inside the timed process): say which
            # build of the kernels they were taken on, and whether that is the build running now
            prof_meta = {"file": "profiles/" + pmc_name, "git_head": j.get("git_head"), "build_id": j.get("build_id"),
                         "matches_running_build": j.get("build_id") == build_id}
            if not prof_meta["matches_running_build"]:
                print(f"bench.py: WARNING: {pmc_name} was taken on build {j.get('build_id')}, the library running now is "
                      f"{build_id}: `traffic` / `counter_bytes` may be stale (tools/profile_round.sh regenerates them)", file=sys.stderr)
        if args.engine == "f32":
            # exact fp32 MFMA: intensity (114 GFLOP / 0.255 GB) >> ridge 25 FLOP/B -> MFMA-bound
            roof = {"bound": "mfma", "kernel": "gemm_f32_kernel<4,2,2,5,16,KCGather,KCPlain,EpiLinear> (in-projection "
                                              "forward with fused embedding gather + dropout)",
                    "achieved": round(tflops, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tflops / FP32_MFMA_PEAK_TFLOPS, 4)}
        else:
            # bf16x3: every fp32 product is three bf16 MFMA products; against section 8(d)'s bytes the binding
            # roofline is the matrix pipe (3 x 122 GFLOP on 2.5 PF = 0.146 ms vs 0.255 GB over 8 TB/s = 0.032 ms)
            name = ("news_fused_fwd_kernel<20, true> (embedding gather + dropout + in-projection + per-head token "
                    "attention in one launch)") if fused else \
                   "gemm_bf16x3_kernel<4,2,4,5,KCGather,KCSplit,EpiLinear> (in-projection forward with fused gather)"
            issued = 3.0 * tflops
            roof = {"bound": "mfma", "kernel": name, "achieved": round(issued, 1), "peak": BF16_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(issued / BF16_MFMA_PEAK_TFLOPS, 4),
                    "algorithmic_fp32_TFLOPs": round(tflops, 1),
                    "hbm_view": {"algorithmic_GBps": round(algo_bytes / avg_s / 1e9, 1), "peak_GBps": HBM_PEAK_GBPS,
                                 "frac": round(algo_bytes / avg_s / 1e9 / HBM_PEAK_GBPS, 4)}}
        roof["algorithmic_bytes_per_launch"] = algo_bytes
        # whole step against the survey's algorithmic model: 4.49 GFLOP and 11.7 MB per impression (fp32-equivalent)
        step_s = median_ms * 1e-3
        step_flops = 4.49e9 * B_PER_GPU
        roof["step"] = {"algorithmic_flops": step_flops, "algorithmic_bytes": 11.7e6 * B_PER_GPU,
                        "counter_bytes": step_bytes,
                        "frac_of_bf16x3_peak": round(3 * step_flops / step_s / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
                        "frac_of_hbm_peak_algorithmic": round(11.7e6 * B_PER_GPU / step_s / 1e9 / HBM_PEAK_GBPS, 4)}
        roof.update({"traffic": traffic, "traffic_profile": prof_meta, "launches": launches.value,
                     "avg_launch_ms": round(avg_s * 1e3, 4), "measured": "HIP events on the launch stream, instrumented second pass (value / median come from the clean timed region)"})
        out = {
            "metric": "impressions/sec (train step) NRMS MINDlarge-shape" if args.workload == "mindlarge"
                      else "impressions/sec (train step) NRMS MINDsmall-shape", "value": round(value, 1),
            "unit": "impressions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "median_ms_per_step": round(median_ms, 4), "value_basis": "median step" if args.steps >= 50 else "mean of the timed region",
            "mean_value": round(mean_value, 1), "prepare_ms": round(prepare_ms, 4),
            "rccl_ranks": dist.get_world_size() if distributed else 1,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.engine == "f32" else "f32 (projections: 3xbf16 split MFMA)",
            "data": "synthetic",
            "config": {"workload": WORKLOADS[args.workload]["name"],
                       "global_batch": world * B_PER_GPU, "parallelism": f"dp{world}", "gemm_engine": args.engine,
                       "optimizer": ("lazy-dense table Adam (bit-identical to dense Adam after flush), rolling flush period 64; "
                                     "dense Adam kernel for the non-table parameters") if trainer.lazy_tables else "dense Adam"},
            "build_id": build_id, "git_head": git_head() or "none on this box (no .git in the snapshot): build_id is the content hash of the kernel sources",
            "roofline": roof,
        }
        if distributed:
            # self-diagnosing multi-GPU line: the exchange that ran, its bytes on the wire per rank, the
            # wire time the xGMI model predicts for it and the MEASURED time the launch stream spent blocked on it, next to the
            # overlap window it had; what the one-GPU step of this same build takes is `multi_gpu_prediction.compute_ms` of a
            # `--gpus 1` run.  scaling_check: the 8-GPU speed-up the model predicts for this exchange, and the step time above which
            # that prediction is falsified
            gx = trainer.exchange_info()
            pw = gx.get("predicted_wire_ms", {})
            choice = gx.get("last_step_choice", gx.get("mode"))
            key = "dense" if str(choice).startswith("dense") else ("owners" if "owner" in str(choice) else ("rows" if "row" in str(choice) else str(choice)))
            pred = pw.get(key, {}) if isinstance(pw.get(key, {}), dict) else {}
            overlap = 0.22 * median_ms
            gx["per_step"] = {"exchange_chosen": choice, "bytes_on_the_wire_per_rank": gx.get("payload_bytes_per_rank"),
                              "predicted_wire_ms": pred, "overlap_window_ms": round(overlap, 4),
                              "predicted_exposed_ms": {m: round(max(0.0, v - overlap), 4) for m, v in pred.items()},
                              "measured_exposed_wait_ms": gx["measured"]["exposed_wait_ms_per_step"],
                              "steps_measured": gx["measured"]["steps_measured"]}
            out["grad_exchange"] = gx
        if world == 1 and not args.no_extras:
            # the forward-only (evaluation-mode) rate of the same workload, outside the timed region
            mod.eval()
            with torch.no_grad():
                # 20 untimed forwards: after a host-only pause the clocks take ~20 forwards to settle (0.61 -> 0.52 ms per forward
                # behind a 1.5 s pause, tools/eval_idle_probe.py); then four groups of ten, every group reported IN ORDER
                for i in range(20):
                    mod.forward(batches[i % N_BATCHES])
                torch.cuda.synchronize()
                groups = []
                for gi in range(4):
                    t1 = time.perf_counter()
                    for i in range(10):
                        mod.forward(batches[i % N_BATCHES])
                    torch.cuda.synchronize()
                    groups.append((time.perf_counter() - t1) / 10)
            groups_in_order = list(groups)
            groups.sort()
            fdt = 0.5 * (groups[1] + groups[2])
            # the same forward with the per-token q|k|v table switched off (every position projected again: rounds 1-5's path)
            from newsreclib_amd.news_encoder import MHSAAddAtt
            uses = dict(MHSAAddAtt.TOKEN_TABLE_USES)
            prev_tt = os.environ.get("NRL_TOKEN_TABLE")
            os.environ["NRL_TOKEN_TABLE"] = "0"
            try:
                with torch.no_grad():
                    for i in range(10):
                        mod.forward(batches[i % N_BATCHES])
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for i in range(20):
                        mod.forward(batches[i % N_BATCHES])
                    torch.cuda.synchronize()
                    plain_ms = (time.perf_counter() - t1) / 20 * 1e3
            finally:
                if prev_tt is None:
                    del os.environ["NRL_TOKEN_TABLE"]
                else:
                    os.environ["NRL_TOKEN_TABLE"] = prev_tt
            mod.train()
            out["forward_only"] = {"value": round(B_PER_GPU / fdt, 1), "unit": "impressions/s", "ms": round(fdt * 1e3, 4),
                                   "group_ms": [round(g * 1e3, 4) for g in groups],
                                   "group_ms_in_order": [round(g * 1e3, 4) for g in groups_in_order],
                                   "token_table": {"tables_built": uses["built"], "forwards_from_it": uses["forwards"],
                                                   "note": "q|k|v of the 70,000 vocabulary ids projected once per weight version "
                                                           "(outside the timed groups: the first untimed forward builds it), "
                                                           "gathered per position; news vectors torch.equal to the table-less forward"},
                                   "without_token_table_ms": round(plain_ms, 4)}
        unforked_ms = None
        if world == 1 and not args.no_extras:
            # the one-GPU step as a rank of an N > 1 job runs it: `news_fork` off (trainer.NRMSTrainer switches it off on more than one
            # rank, so that all three weight gradients stay in the window the gradient exchange hides in) -- the compute term of
            # `multi_gpu_prediction`
            was = bool((lib.nrl_get_options() >> _lib.OPTION_NAMES.index("news_fork")) & 1)
            _lib.set_option("news_fork", False)
            try:
                for i in range(5):
                    trainer.step(batches[i % N_BATCHES], batches[(i + 1) % N_BATCHES])
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
                torch.cuda.synchronize()
                ev[0].record()
                for i in range(30):
                    trainer.step(batches[i % N_BATCHES], batches[(i + 1) % N_BATCHES])
                    ev[i + 1].record()
                torch.cuda.synchronize()
                unforked_ms = statistics.median(ev[i].elapsed_time(ev[i + 1]) for i in range(30))
            finally:
                _lib.set_option("news_fork", was)
        if world == 1 and args.engine != "f32" and not args.no_extras:
            # the exact-fp32 projection engine on the same workload (extra key, outside the timed region), with its own
            # roofline: the in-projection GEMM with the fused gather, against the fp32 MFMA peak
            _lib.set_gemm_engine("f32")
            for i in range(3):
                trainer.step(batches[i % N_BATCHES])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n32 = max(10, min(args.steps, 30))
            for i in range(n32):
                trainer.step(batches[i % N_BATCHES])
            torch.cuda.synchronize()
            d32 = (time.perf_counter() - t1) / n32
            lib.nrl_prof_enable(1)
            for i in range(5):
                trainer.step(batches[i % N_BATCHES])
            torch.cuda.synchronize()
            t32, l32, f32 = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
            lib.nrl_prof_read(ctypes.byref(t32), ctypes.byref(l32), ctypes.byref(f32))
            lib.nrl_prof_enable(0)
            _lib.set_gemm_engine(args.engine)
            out["f32_engine"] = {"value": round(B_PER_GPU / d32, 1), "unit": "impressions/s",
                                 "ms_per_step": round(d32 * 1e3, 4), "dtype": "f32 (v_mfma_f32_16x16x4_f32 projections)"}
            dl, dms, dfl = l32.value, t32.value, f32.value          # (nrl_prof_enable(1) restarted the counters)
            if dl > 0 and dms > 0:
                tf32 = dfl / dl / (dms * 1e-3 / dl) / 1e12
                out["f32_engine"]["roofline"] = {
                    "bound": "mfma", "kernel": "gemm_f32_kernel<4,2,2,5,16,KCGather,KCPlain,EpiLinear> (in-projection forward "
                                               "with fused embedding gather + dropout)",
                    "achieved": round(tf32, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tf32 / FP32_MFMA_PEAK_TFLOPS, 4), "avg_launch_ms": round(dms / dl, 4), "launches": dl,
                    "step_frac_of_fp32_mfma_peak": round(step_flops / d32 / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)}
        if world == 1 and not args.no_extras:
            # the other single-GPU configurations of BASELINE.json, driver-timed (outside the timed region)
            del trainer, mod, batches
            torch.cuda.empty_cache()
            for key, fn in (("lstur", extra_lstur), ("plm", extra_plm)):
                try:
                    out[key] = fn(device)
                except Exception as e:                     # an extra must never cost the headline line
                    out[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
                torch.cuda.empty_cache()
        if not distributed and not args.no_extras:
            # host-only model of the 8-GPU step.  LAST of the device-side extras on purpose (round 6): its host work (sort-based
            # torch.unique over 1.7 M ids on the framework's thread pool, nine times) was what set off the ~85 ms stall that
            # rounds 4-5 saw once inside the evaluation loop that used to follow it -- tools/eval_idle_probe.py:
            # 9 stalls in 33 processes with this host work in front of the loop, 0 in 21 (+ 24 rounds inside one) without it; DESIGN section 5
            try:
                out["multi_gpu_prediction"] = predict_multi_gpu(unforked_ms or median_ms)
                out["multi_gpu_prediction"]["compute_basis"] = (
                    "median of 30 one-GPU steps with news_fork OFF, as ranks of an N > 1 job run (%.4f ms; this run's own step with the "
                    "fork: %.4f ms)" % (unforked_ms, median_ms)) if unforked_ms else "this run's one-GPU step"
            except Exception as e:                         # an extra must never cost the headline line
                out["multi_gpu_prediction"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        if world == 1 and not args.no_cpu_baseline and not args.no_extras:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
