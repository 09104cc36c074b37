"""Stand-in for the absent ``librosa`` package, used ONLY by tools/gen_golden.py to import the
reference's modules in the build container.  Our own restatement of the four functions the
rmvpe+ path reaches; "parity unpinned" against real librosa."""
from . import filters, util, feature  # noqa: F401


def resample(y, orig_sr, target_sr, **kw):
    if orig_sr == target_sr:
        return y
    raise NotImplementedError("librosa.resample stand-in: only the identity case is reached")


def to_mono(y):
    import numpy as np
    return np.mean(y, axis=0) if y.ndim > 1 else y
