import numpy as np


def pad_center(data, size, axis=-1, **kw):
    n = data.shape[axis]
    lpad = int((size - n) // 2)
    lengths = [(0, 0)] * data.ndim
    lengths[axis] = (lpad, int(size - n - lpad))
    return np.pad(data, lengths, **kw)


def tiny(x):
    x = np.asarray(x)
    dtype = x.dtype if np.issubdtype(x.dtype, np.floating) else np.float32
    return np.finfo(dtype).tiny


def normalize(S, norm=np.inf, **kw):
    if norm is None:
        return S
    raise NotImplementedError
