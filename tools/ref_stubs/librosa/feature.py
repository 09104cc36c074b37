import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", ".."))
from oracle.pipeline import frame_rms


def rms(y=None, frame_length=2048, hop_length=512, **kw):
    return frame_rms(y, frame_length, hop_length)
