"""CPU: host-side logic of the package (config structs, checkpoint/index readers, mirror signatures)."""
import inspect
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch

import polgen_rvc_amd  # noqa: F401
from polgen_rvc_amd import ckpt_io, dist as D, index_io, synthetic as S, weights as W
from polgen_rvc_amd.infer import infer as I, pipeline as P


def test_synth_cfg_struct_roundtrip():
    s = W.synth_cfg_struct(S.SYNTH_CFG_48K, 768)
    assert (s.inter_channels, s.hidden_channels, s.filter_channels, s.n_heads, s.n_layers) == (192, 192, 768, 2, 6)
    assert list(s.up_rates)[:4] == [12, 10, 2, 2] and list(s.up_kernels)[:4] == [24, 20, 4, 4]
    assert [list(r) for r in s.res_dilations][:3] == [[1, 3, 5]] * 3 and s.sr == 48000 and s.input_dim == 768
    with pytest.raises(ValueError):
        bad = list(S.SYNTH_CFG_48K)
        bad[9] = "2"
        W.synth_cfg_struct(bad)


def test_get_vc_rejects_bad_checkpoints():
    with pytest.raises(ValueError):
        I.get_vc("cuda:0", False, I.Config(), "x.pth", cpt={"weight": {}})      # infer.py:80-84
    cpt = S.synth_checkpoint(S.SYNTH_CFG_TINY, 0, version="v1")
    with pytest.raises(ValueError):
        I.get_vc("cuda:0", False, I.Config(), "x.pth", cpt=cpt)


def test_mirror_signatures_match_reference():
    """Parameter names/order of the reference's entry points (rvc/infer/infer.py:109-128,
    rvc/infer/pipeline.py:289-311,132-144) -- extra trailing keyword-only-style extras allowed."""
    ri = ["index_path", "index_rate", "input_path", "output_path", "pitch", "f0_method", "cpt", "version", "net_g",
          "filter_radius", "tgt_sr", "volume_envelope", "protect", "hop_length", "vc", "hubert_model", "f0_min",
          "f0_max"]
    assert list(inspect.signature(I.rvc_infer).parameters) == ri
    pp = ["self", "model", "net_g", "sid", "audio", "input_audio_path", "pitch", "f0_method", "file_index",
          "index_rate", "pitch_guidance", "filter_radius", "tgt_sr", "resample_sr", "volume_envelope", "version",
          "protect", "hop_length", "f0_file", "f0_min", "f0_max"]
    assert list(inspect.signature(P.VC.pipeline).parameters)[:len(pp)] == pp
    assert list(inspect.signature(I.get_vc).parameters)[:4] == ["device", "is_half", "config", "model_path"]
    assert list(inspect.signature(I.load_hubert).parameters)[:3] == ["device", "is_half", "model_path"]
    cfg = I.Config()
    assert (cfg.x_pad, cfg.x_query, cfg.x_center, cfg.x_max) == (1, 6, 38, 41) and cfg.is_half is False
    vc = P.VC(48000, cfg)
    assert (vc.t_pad, vc.t_pad_tgt, vc.t_query, vc.t_center, vc.t_max, vc.window) == (16000, 48000, 96000, 608000,
                                                                                    656000, 160)


def test_fairseq_checkpoint_without_fairseq(tmp_path):
    """hubert_base.pt pickles reference fairseq/omegaconf classes; the restricted unpickler must ignore
    them and return the tensor dict."""
    mod = types.ModuleType("fairseq_fake_cfg")
    exec("class HubertConfig:\n    def __init__(self):\n        self.label_rate = 50\n", mod.__dict__)
    HubertConfig = mod.HubertConfig
    HubertConfig.__module__ = "fairseq_fake_cfg"
    HubertConfig.__qualname__ = "HubertConfig"
    sys.modules["fairseq_fake_cfg"] = mod
    state = S.to_torch(S.hubert_state(S.HUBERT_CFG_TINY, 0))
    path = os.path.join(tmp_path, "hubert.pt")
    torch.save({"cfg": HubertConfig(), "args": None, "model": state}, path)
    del sys.modules["fairseq_fake_cfg"]
    got = ckpt_io.load_fairseq_hubert(path)
    assert set(got) == set(state)
    assert torch.equal(got["encoder.pos_conv.0.weight_g"], state["encoder.pos_conv.0.weight_g"])


def test_index_io_npy_and_flat(tmp_path):
    big = S.make_index(100, 16, 1)
    p = os.path.join(tmp_path, "big.npy")
    np.save(p, big)
    assert np.array_equal(index_io.read_index_vectors(p), big)
    # hand-built IndexFlatL2 file in the published faiss io layout
    import struct
    blob = b"IxF2" + struct.pack("<iqqqBi", 16, 100, 1 << 20, 1 << 20, 1, 1) + struct.pack("<Q", 1600) + big.tobytes()
    q = os.path.join(tmp_path, "flat.index")
    open(q, "wb").write(blob)
    assert np.array_equal(index_io.read_index_vectors(q), big)


def test_shard_is_a_balanced_partition():
    lens = [5, 30, 12, 7, 30, 9, 15, 3, 11]
    parts = [D.shard(len(lens), r, 4, lens) for r in range(4)]
    assert sorted(sum(parts, [])) == list(range(len(lens)))
    assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_make_clip_is_deterministic():
    a, b = S.make_clip(3, 1.0), S.make_clip(3, 1.0)
    assert a.dtype == np.float32 and a.shape == (16000,) and np.array_equal(a, b) and np.abs(a).max() < 1.0
