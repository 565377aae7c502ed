#!/usr/bin/env python3
"""Generates tests/golden/reference_contract.json: the drop-in contract of the reference as the reference's own files state it.

Run in the BUILD container only (needs /root/reference):   python tests/golden/make_contract.py

What goes in (data only -- names, orders, shapes, defaults; no source text):
  * constructor keyword names, IN ORDER, of the recommender modules and of the component classes the hot path instantiates,
    taken from the reference sources with `ast` (the module classes cannot be imported here: lightning / torch_geometric /
    torchmetrics are not installed);
  * the `configs/model/*.yaml` defaults of the recommenders this repo mirrors (`_target_`, scalar hyper-parameters, optimizer);
  * state_dict keys and shapes of the reference COMPONENT classes that do import here (text encoders, user encoders,
    additive attention, category encoder), instantiated at the configs' sizes with a small vocabulary.
tests/test_host.py::test_product_matches_the_reference_contract_fixture compares the product classes with this file, so the
names the builder typed by hand in tests/helpers.py are no longer the only statement of the contract.
"""
import ast
import json
import os
import sys

import yaml

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_contract.json")

MODULES = {            # product module file -> (reference file, class)
    "nrms_module.NRMSModule": ("newsreclib/models/general_rec/nrms_module.py", "NRMSModule"),
    "lstur_module.LSTURModule": ("newsreclib/models/general_rec/lstur_module.py", "LSTURModule"),
    "naml_module.NAMLModule": ("newsreclib/models/general_rec/naml_module.py", "NAMLModule"),
    "tanr_module.TANRModule": ("newsreclib/models/general_rec/tanr_module.py", "TANRModule"),
    "mins_module.MINSModule": ("newsreclib/models/general_rec/mins_module.py", "MINSModule"),
    "cen_news_rec_module.CenNewsRecModule": ("newsreclib/models/general_rec/cen_news_rec_module.py", "CenNewsRecModule"),
}
COMPONENTS = {
    "news_encoder.MHSAAddAtt": ("newsreclib/models/components/encoders/news/text.py", "MHSAAddAtt"),
    "news_encoder.CNNAddAtt": ("newsreclib/models/components/encoders/news/text.py", "CNNAddAtt"),
    "news_encoder.CNNMHSAAddAtt": ("newsreclib/models/components/encoders/news/text.py", "CNNMHSAAddAtt"),
    "news_encoder.PLM": ("newsreclib/models/components/encoders/news/text.py", "PLM"),
    "news_encoder.NewsEncoder": ("newsreclib/models/components/encoders/news/news.py", "NewsEncoder"),
    "news_encoder.LinearEncoder": ("newsreclib/models/components/encoders/news/category.py", "LinearEncoder"),
    "attention.AdditiveAttention": ("newsreclib/models/components/layers/attention.py", "AdditiveAttention"),
    "user_encoder.UserEncoder": ("newsreclib/models/components/encoders/user/nrms.py", "UserEncoder"),
    "user_encoder_lstur.UserEncoder": ("newsreclib/models/components/encoders/user/lstur.py", "UserEncoder"),
    "user_encoder_naml.UserEncoder": ("newsreclib/models/components/encoders/user/naml.py", "UserEncoder"),
    "user_encoder_mins.UserEncoder": ("newsreclib/models/components/encoders/user/mins.py", "UserEncoder"),
    "user_encoder_cen_news_rec.UserEncoder": ("newsreclib/models/components/encoders/user/cen_news_rec.py", "UserEncoder"),
    "click_predictor.DotProduct": ("newsreclib/models/components/layers/click_predictor.py", "DotProduct"),
}
CONFIGS = ["nrms", "lstur", "naml", "tanr", "mins", "cen_news_rec"]


def ctor_and_forward(path, cls):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == cls:
            out = {"file": path, "line": node.lineno}
            for fn in node.body:
                if isinstance(fn, ast.FunctionDef) and fn.name in ("__init__", "forward"):
                    args = [a.arg for a in fn.args.args if a.arg != "self"]
                    n_def = len(fn.args.defaults)
                    out[fn.name] = {"args": args, "line": fn.lineno,
                                    "required": args[: len(args) - n_def] if n_def else args}
            return out
    raise KeyError((path, cls))


def scalars(d):
    """yaml mapping -> the scalar / list-of-scalar entries (interpolations such as ${paths.data_dir} kept as strings)."""
    out = {}
    for k, v in d.items():
        if isinstance(v, (int, float, str, bool)) or v is None:
            out[k] = v
        elif isinstance(v, list) and all(isinstance(x, (int, float, str, bool)) for x in v):
            out[k] = v
        elif isinstance(v, dict) and k in ("optimizer", "scheduler", "outputs"):
            out[k] = v
    return out


def component_state_dicts():
    sys.path.insert(0, REF)
    import torch
    from newsreclib.models.components.encoders.news.category import LinearEncoder
    from newsreclib.models.components.encoders.news.text import CNNAddAtt, CNNMHSAAddAtt, MHSAAddAtt
    from newsreclib.models.components.encoders.user.cen_news_rec import UserEncoder as CenUser
    from newsreclib.models.components.encoders.user.lstur import UserEncoder as LsturUser
    from newsreclib.models.components.encoders.user.mins import UserEncoder as MinsUser
    from newsreclib.models.components.encoders.user.naml import UserEncoder as NamlUser
    from newsreclib.models.components.encoders.user.nrms import UserEncoder as NrmsUser
    from newsreclib.models.components.layers.attention import AdditiveAttention
    V = 64
    emb = torch.zeros(V, 300)

    def sd(m):
        return {k: list(v.shape) for k, v in m.state_dict().items()}

    return {
        "vocab_rows_used": V,
        "news_encoder.MHSAAddAtt": sd(MHSAAddAtt(pretrained_embeddings=emb, embed_dim=300, num_heads=15, query_dim=200,
                                                 dropout_probability=0.2)),
        "news_encoder.CNNAddAtt": sd(CNNAddAtt(pretrained_embeddings=emb, embed_dim=300, num_filters=300, window_size=3,
                                               query_dim=200, dropout_probability=0.2)),
        "news_encoder.CNNMHSAAddAtt": sd(CNNMHSAAddAtt(pretrained_embeddings=emb, embed_dim=300, num_filters=400, window_size=3,
                                                       num_heads=20, query_dim=200, dropout_probability=0.2)),
        "news_encoder.LinearEncoder": sd(LinearEncoder(pretrained_embeddings=None, from_pretrained=False, freeze_pretrained_emb=False,
                                                       num_categories=19, embed_dim=100, use_dropout=False,
                                                       dropout_probability=None, linear_transform=False, output_dim=None)),
        "attention.AdditiveAttention": sd(AdditiveAttention(input_dim=300, query_dim=200)),
        "user_encoder.UserEncoder": sd(NrmsUser(news_embed_dim=300, num_heads=15, query_dim=200)),
        "user_encoder_lstur.UserEncoder": sd(LsturUser(num_users=100, input_dim=700, user_masking_probability=0.5,
                                                       long_short_term_method="ini")),
        "user_encoder_naml.UserEncoder": sd(NamlUser(news_embed_dim=400, query_dim=200)),
        "user_encoder_mins.UserEncoder": sd(MinsUser(news_embed_dim=300, query_dim=200, num_filters=300, num_gru_channels=6)),
        "user_encoder_cen_news_rec.UserEncoder": sd(CenUser(num_filters=400, num_heads=20, query_dim=200, gru_hidden_dim=400,
                                                            num_recent_news=20, dropout_probability=0.2)),
    }


def main():
    contract = {
        "generated_by": "tests/golden/make_contract.py (ast + yaml over /root/reference; component state dicts by instantiation)",
        "modules": {k: ctor_and_forward(*v) for k, v in MODULES.items()},
        "components": {k: ctor_and_forward(*v) for k, v in COMPONENTS.items()},
        "configs": {}, "state_dicts": component_state_dicts(),
    }
    for name in CONFIGS:
        path = os.path.join("configs", "model", name + ".yaml")
        contract["configs"][name] = {"file": path, "values": scalars(yaml.safe_load(open(os.path.join(REF, path))))}
    json.dump(contract, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
