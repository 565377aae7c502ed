"""Base class for the drop-in module: ``lightning.LightningModule`` when Lightning is installed
(the reference's runtime, ``abstract_recommender.py:17``), else a minimal stand-in with the members
the module itself uses (``save_hyperparameters``/``hparams``, ``log``/``log_dict``, ``device``).
Lightning is not installed in the build image, so the stand-in is what the tests exercise."""
import inspect
from types import SimpleNamespace

import torch
import torch.nn as nn

try:  # pragma: no cover - depends on the environment
    from lightning import LightningModule as _Base
    HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    try:  # pragma: no cover
        from pytorch_lightning import LightningModule as _Base
        HAVE_LIGHTNING = True
    except Exception:  # noqa: BLE001
        _Base = None
        HAVE_LIGHTNING = False


class _HParams(SimpleNamespace):
    def __getitem__(self, k):
        return getattr(self, k)

    def keys(self):
        return self.__dict__.keys()


class _MiniLightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self._hparams = _HParams()
        self.logged = {}

    def save_hyperparameters(self, *args, logger=True, ignore=(), **kwargs):
        frame = inspect.currentframe().f_back
        # walk up to the outermost __init__ of this object so subclass kwargs are captured
        init_args = {}
        while frame is not None:
            local = frame.f_locals
            if local.get("self") is self and frame.f_code.co_name == "__init__":
                for k, v in local.items():
                    if k not in ("self", "__class__", "args", "kwargs") and not k.startswith("_") \
                            and k not in ignore:
                        init_args.setdefault(k, v)
                for k, v in local.get("kwargs", {}).items():
                    init_args.setdefault(k, v)
            frame = frame.f_back
        self._hparams = _HParams(**init_args)

    @property
    def hparams(self):
        return self._hparams

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")

    def log(self, name, value, **kwargs):
        self.logged[name] = value

    def log_dict(self, d, **kwargs):
        self.logged.update(dict(d))


LightningModuleBase = _Base if HAVE_LIGHTNING else _MiniLightningModule
