"""Reading real checkpoints without the packages that wrote them.

``hubert_base.pt`` is a fairseq checkpoint: a pickle whose ``"model"`` entry is the tensor dict we need
and whose ``"cfg"`` / ``"args"`` entries reference fairseq / omegaconf classes that are not installed.
A restricted unpickler resolves an exact allow-list of (module, name) pairs -- tensor rebuild helpers, storages,
dtypes, OrderedDict, harmless builtin containers -- and replaces every other global (including builtins.eval /
exec / getattr / __import__) by an inert stub, so only tensors are ever materialised.
"""
from __future__ import annotations

import pickle
import types

import re

# Exact (module, name) pairs the unpickler may resolve: what torch.save emits for tensors and plain containers.
# Everything else -- fairseq / omegaconf / argparse classes AND dangerous builtins such as eval, exec, getattr or
# __import__ -- becomes an inert stub, so a crafted file cannot run code through find_class.
_ALLOWED = {
    ("collections", "OrderedDict"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"),
    ("torch._utils", "_rebuild_parameter"), ("torch._utils", "_rebuild_parameter_with_state"),
    ("torch._tensor", "_rebuild_from_type_v2"), ("torch", "Tensor"), ("torch", "Size"), ("torch", "device"),
    ("torch.nn.parameter", "Parameter"), ("torch.serialization", "_get_layout"),
    ("torch.storage", "UntypedStorage"), ("torch.storage", "TypedStorage"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "dtype"), ("numpy", "ndarray"), ("_codecs", "encode"),
}
_SAFE_BUILTINS = {"set", "frozenset", "slice", "complex", "bytearray", "range", "int", "float", "bool", "list",
                  "dict", "tuple", "str", "bytes", "object"}
_TORCH_DATA = re.compile(r"^(\w+Storage|float\d+|bfloat16|half|double|int\d+|uint8|long|short|bool|complex\d+)$")


class _Stub:
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        pass

    def __call__(self, *a, **k):
        return _Stub()


class _StubUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED or (module == "builtins" and name in _SAFE_BUILTINS) or \
                (module == "torch" and _TORCH_DATA.match(name)):
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
