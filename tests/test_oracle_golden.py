"""CPU: the oracle (oracle/*.py) against the vectors captured from the reference's own modules
(tests/golden/*.npz, produced by tools/gen_golden.py in the build container).  This is what pins the
oracle; HuBERT is pinned against the transformers twin only (fairseq absent -> parity unpinned)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import rms

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _S():
    import polgen_rvc_amd  # noqa: F401
    from polgen_rvc_amd import synthetic
    return synthetic


@pytest.mark.parametrize("tag", ["tiny", "48k_T24", "48k_T24_outliers"])
def test_synth_oracle_vs_reference(tag):
    from oracle import synth as O
    S = _S()
    d = np.load(os.path.join(GOLD, f"synth_{tag}.npz"))
    cfg = json.loads(str(d["cfg"]))
    sd = S.to_torch(S.synth_state(cfg, int(d["seed"]), outliers=tag.endswith("outliers")))
    T = d["phone"].shape[1]
    out, parts = O.synthesizer_infer(sd, cfg, torch.from_numpy(d["phone"]), torch.tensor([T]),
                                     torch.from_numpy(d["pitch"]), torch.from_numpy(d["f0"]), torch.tensor([0]),
                                     torch.from_numpy(d["z_noise"]), torch.from_numpy(d["src_noise"]),
                                     return_parts=True)
    assert rms(parts["m_p"].numpy() - d["m_p"]) < 1e-6
    assert rms(parts["z"].numpy() - d["z"]) < 1e-5
    assert rms(out.numpy() - d["audio"]) < 1e-5


@pytest.mark.parametrize("tag", ["tiny", "full_1s", "full_1s_outliers"])
def test_rmvpe_oracle_vs_reference(tag):
    from oracle import rmvpe as O
    S = _S()
    d = np.load(os.path.join(GOLD, f"rmvpe_{tag}.npz"))
    cfg = json.loads(str(d["cfg"]))
    sd = S.to_torch(S.rmvpe_state(cfg, int(d["seed"]), outliers=tag.endswith("outliers")))
    f0, hid, mel = O.infer_f0(sd, cfg, d["audio"].astype(np.float64), return_hidden=True)
    st = int(d["stride"])
    assert rms(hid[::st] - d["hidden"]) / rms(d["hidden"]) < 1e-4
    assert np.abs(f0 - d["f0"]).max() < 1e-2
    assert len(O.unstable_frames(hid)) == 0          # the golden instance is well-conditioned


@pytest.mark.parametrize("tag", ["tiny", "base_1s"])
def test_hubert_oracle_vs_hf_twin(tag):
    from oracle import hubert as O
    S = _S()
    d = np.load(os.path.join(GOLD, f"hubert_{tag}.npz"))
    cfg = json.loads(str(d["cfg"]))
    sd = S.to_torch(S.hubert_state(cfg, int(d["seed"])))
    out = O.extract_features(sd, cfg, torch.from_numpy(d["wav"]), cfg["layers"]).numpy()
    assert rms(out - d["out"]) / rms(d["out"]) < 1e-4


@pytest.mark.parametrize("tag", ["tiny_single", "tiny_ciargs", "tiny_chunked", "tiny_short", "tiny_v1", "tiny_sid3_noprotect"])
def test_pipeline_oracle_vs_reference(tag):
    from oracle import pipeline as OP
    S = _S()
    d = np.load(os.path.join(GOLD, f"pipeline_{tag}.npz"))
    hcfg, rcfg, scfg = json.loads(str(d["cfgs"]))
    seed = int(d["seed"])
    version = str(d["version"]) if "version" in d.files else "v2"         # "v1": HuBERT layer 9 + final_proj, emb_phone on final_dim
    in_dim = hcfg["final_dim"] if version == "v1" else hcfg["embed_dim"]
    models = OP.Models(S.to_torch(S.hubert_state(hcfg, seed)), hcfg, S.to_torch(S.rmvpe_state(rcfg, seed)), rcfg,
                       S.to_torch(S.synth_state(scfg, seed, input_dim=in_dim)), scfg, version=version)
    noises = [(torch.from_numpy(d[f"z_noise_{i}"]), torch.from_numpy(d[f"src_noise_{i}"]))
              for i in range(int(d["n_chunks"]))]
    audio = S.make_clip(int(d["clip"]), float(d["seconds"]))
    sid = int(d["sid"]) if "sid" in d.files else 0
    pcm, parts = OP.pipeline(models, OP.Geometry(scfg[-1], *[int(v) for v in d["geo"]]), audio, float(d["pitch"]),
                             sid, None, 0.0, float(d["volume_envelope"]), float(d["protect"]), float(d["f0_min"]),
                             float(d["f0_max"]), noises=noises, return_parts=True)
    assert len(parts["plan"]) == int(d["n_chunks"])
    assert pcm.shape == d["pcm"].shape
    assert np.abs(pcm.astype(np.int32) - d["pcm"].astype(np.int32)).max() <= 2
    assert (parts["coarse"] == d["coarse"]).all()
    assert rms(np.concatenate(parts["raw"]) - d["raw"]) < 1e-4


def test_layout_contracts():
    """Synthetic checkpoints carry exactly the key/shape set of the reference modules (layouts.json was
    dumped from Synthesizer / E2E state_dicts)."""
    S = _S()
    with open(os.path.join(GOLD, "layouts.json")) as f:
        L = json.load(f)
    for name, state in (("synth_48k", S.synth_state(S.SYNTH_CFG_48K)), ("rmvpe", S.rmvpe_state(S.RMVPE_CFG_FULL))):
        assert set(state) == set(L[name])
        for k, v in state.items():
            assert list(v.shape) == L[name][k], k


def test_index_blend_oracle_properties():
    from oracle import pipeline as OP
    S = _S()
    big = S.make_index(512, 32, 0)
    q = big[[3, 77, 200]] + 1e-3
    out, ids, dist = OP.index_blend(q.astype(np.float32), big, 1.0)
    assert (ids[:, 0] == [3, 77, 200]).all() and (np.diff(dist, axis=1) >= 0).all()
    assert rms(out - big[[3, 77, 200]]) < 1e-2          # (1/d)^2 weights are dominated by the exact neighbour


@pytest.mark.parametrize("tag", ["tiny", "full_2s"])
def test_fcpe_oracle_vs_reference(tag):
    """oracle/fcpe.py against the reference's FCPE module and VC.get_f0(f0_method="fcpe") call site."""
    from oracle import fcpe as O, pipeline as OP
    S = _S()
    d = np.load(os.path.join(GOLD, f"fcpe_{tag}.npz"))
    cfg = json.loads(str(d["cfg"]))
    sd = S.to_torch(S.fcpe_state(cfg, int(d["seed"])))
    st = int(d["stride"])
    mel = O.mel_spectrogram(torch.from_numpy(d["x"])[None])
    assert np.abs(mel[0].numpy().T[:, ::st] - d["mel"]).max() < 1e-5
    sal = O.salience(sd, mel)[0].numpy()
    assert rms(sal[::st] - d["salience"]) / rms(d["salience"]) < 1e-5
    raw = O.infer_hz(sd, d["x"], 0.03)
    assert ((raw > 0) == (d["raw_f0"] > 0)).all() and np.abs(raw - d["raw_f0"]).max() < 1e-2
    coarse, f0 = OP.f0_to_coarse(O.compute_f0(sd, d["x"], len(d["x"]) // 160), float(d["pitch"]), 50, 1100)
    assert np.abs(f0 - d["f0"]).max() < 1e-2 and (coarse == d["coarse"]).all()


def test_fcpe_pipeline_oracle_vs_reference():
    from oracle import pipeline as OP
    S = _S()
    d = np.load(os.path.join(GOLD, "pipeline_tiny_fcpe.npz"))
    hcfg, fcfg, scfg = json.loads(str(d["cfgs"]))
    seed = int(d["seed"])
    models = OP.Models(S.to_torch(S.hubert_state(hcfg, seed)), hcfg, None, None,
                       S.to_torch(S.synth_state(scfg, seed, input_dim=hcfg["embed_dim"])), scfg,
                       fcpe_sd=S.to_torch(S.fcpe_state(fcfg, seed)))
    noises = [(torch.from_numpy(d["z_noise_0"]), torch.from_numpy(d["src_noise_0"]))]
    pcm = OP.pipeline(models, OP.Geometry(scfg[-1], *[int(v) for v in d["geo"]]), S.make_clip(int(d["clip"]), float(d["seconds"])),
                      float(d["pitch"]), 0, None, 0.0, float(d["volume_envelope"]), float(d["protect"]), 50, 1100,
                      noises=noises, f0_method="fcpe")
    diff = np.abs(pcm.astype(np.int32) - d["pcm"].astype(np.int32))
    assert pcm.shape == d["pcm"].shape and diff.max() <= 8 and np.mean(diff > 1) < 0.02
