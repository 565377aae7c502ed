"""300 fused-Adam steps on one fixed ragged batch: the loss must fall and every parameter stay finite."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functools import partial
from newsreclib_amd.nrms_module import NRMSModule, prepare_batch
from newsreclib_amd.synthetic import make_batch
from newsreclib_amd.trainer import NRMSTrainer
torch.manual_seed(0)
V=20000
mod = NRMSModule(dataset_attributes=["title","abstract","category"], attributes2encode=["title"], outputs={"train":[],"val":[],"test":[]},
    dual_loss_training=False, dual_loss_coef=None, loss="cross_entropy_loss", late_fusion=False, temperature=None, use_plm=False,
    pretrained_embeddings_path=None, plm_model=None, frozen_layers=None, embed_dim=300, num_heads=15, query_dim=200, dropout_probability=0.2,
    top_k_list=[5,10], num_categ_classes=18, num_sent_classes=3, save_recs=False, recs_fpath=None,
    optimizer=partial(torch.optim.Adam, lr=1e-4), scheduler=None, pretrained_embeddings=torch.randn(V,300)*0.3).cuda()
tr = NRMSTrainer(mod, lr=1e-4)
fixed = prepare_batch(make_batch(128, vocab=V, mode="ragged", seed=7, device="cuda"))
losses=[]
for i in range(300):
    l = tr.step(fixed)
    if i % 50 == 0 or i == 299: losses.append(float(l))
print("same batch, 300 steps:", [round(x,4) for x in losses])
assert all(torch.isfinite(p).all() for p in mod.parameters())
assert losses[-1] < losses[0] * 0.5
