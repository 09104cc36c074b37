"""Drop-in mirror of the reference's ``rvc.infer`` package (infer.py + pipeline.py)."""
from .infer import Config, load_hubert, get_vc, rvc_infer  # noqa: F401
from .pipeline import VC  # noqa: F401
