"""Reading real checkpoints without the packages that wrote them.

``hubert_base.pt`` is a fairseq checkpoint: a pickle whose ``"model"`` entry is the tensor dict we need
and whose ``"cfg"`` / ``"args"`` entries reference fairseq / omegaconf classes that are not installed.
A restricted unpickler resolves torch / numpy / builtins normally and replaces every other global by an
inert stub, so only tensors are ever materialised.
"""
from __future__ import annotations

import pickle
import types

_SAFE_ROOTS = ("torch", "collections", "numpy", "builtins", "_codecs", "copyreg")


class _Stub:
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        pass

    def __call__(self, *a, **k):
        return _Stub()


class _StubUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split(".")[0] in _SAFE_ROOTS:
            return super().find_class(module, name)
        return type(name, (_Stub,), {})


_stub_pickle = types.ModuleType("rvcx_stub_pickle")
_stub_pickle.Unpickler = _StubUnpickler
_stub_pickle.load = lambda f, **kw: _StubUnpickler(f, **kw).load()
_stub_pickle.__name__ = "pickle"


def load_fairseq_hubert(path: str) -> dict:
    """-> {name: torch.Tensor} of the HubertModel (fairseq key names)."""
    import torch
    ck = torch.load(path, map_location="cpu", pickle_module=_stub_pickle, weights_only=False)
    if isinstance(ck, dict) and "model" in ck:
        ck = ck["model"]
    state = {k: v for k, v in ck.items() if hasattr(v, "shape")}
    if "encoder.pos_conv.0.weight_g" not in state:
        raise ValueError(f"{path}: not a fairseq HuBERT checkpoint (no encoder.pos_conv.0.weight_g)")
    return state
