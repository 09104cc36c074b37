"""The drop-in boundary on the GPU: ``rvc_infer`` with file paths end to end (WAV in, checkpoints
from disk in their real container formats, lazy RMVPE, WAV out) against the reference's own output, and the
stage methods ``VC.get_f0`` / ``VC.get_f0_rmvpe`` / ``VC.vc`` with the reference's argument meaning."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import rms

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _pack_noise(d):
    parts = []
    for i in range(int(d["n_chunks"])):
        parts += [d[f"z_noise_{i}"].ravel(), d[f"src_noise_{i}"].ravel()]
    return np.concatenate(parts).astype(np.float32)


@pytest.fixture()
def fresh_ctx():
    """A context of its own registered as device 0's, so that residency / lazy loading start from nothing."""
    from polgen_rvc_amd import _lib
    from polgen_rvc_amd.infer import infer as I, _state
    saved = (dict(_state._CTX), dict(_state._RESIDENT), dict(_state._SYNTHS), dict(_state._INDEX_RESIDENT))
    c = _lib.Context(0)
    _state._CTX.clear()
    _state._CTX[0] = c
    I.clear_cache()
    yield c
    I.clear_cache()
    for dst, src in zip((_state._CTX, _state._RESIDENT, _state._SYNTHS, _state._INDEX_RESIDENT), saved):
        dst.clear()
        dst.update(src)
    c.close()


def _write_assets(tmp_path, d, fp16=False, legacy_wn=False):
    """Synthetic checkpoints of the golden's models in the real container formats: a fairseq-shaped hubert_base.pt,
    rvc/models/predictors/rmvpe.pt (bare state dict), a voice-model .pth (optionally fp16 tensors and legacy
    weight_g / weight_v names), and the 16 kHz clip as a float32 WAV."""
    from scipy.io import wavfile
    from polgen_rvc_amd import synthetic as S
    hcfg, rcfg, scfg = json.loads(str(d["cfgs"]))
    seed = int(d["seed"])
    hub = tmp_path / "hubert_base.pt"
    torch.save({"cfg": None, "args": None, "model": S.to_torch(S.hubert_state(hcfg, seed))}, hub)
    rdir = tmp_path / "rvc" / "models" / "predictors"
    rdir.mkdir(parents=True)
    torch.save(S.to_torch(S.rmvpe_state(rcfg, seed)), rdir / "rmvpe.pt")
    cpt = S.synth_checkpoint(scfg, seed)
    w = S.synth_state(scfg, seed, input_dim=hcfg["embed_dim"])
    if legacy_wn:
        w = {k.replace(".parametrizations.weight.original0", ".weight_g")
              .replace(".parametrizations.weight.original1", ".weight_v"): v for k, v in w.items()}
    w = S.to_torch(w)
    if fp16:
        w = {k: v.half() for k, v in w.items()}
    cpt["weight"] = w
    pth = tmp_path / "voice.pth"
    torch.save(cpt, pth)
    wav = tmp_path / "in.wav"
    wavfile.write(wav, 16000, S.make_clip(int(d["clip"]), float(d["seconds"])))
    return str(hub), str(rdir / "rmvpe.pt"), str(pth), str(wav), (hcfg, rcfg, scfg)


def test_rvc_infer_files_end_to_end_vs_reference_golden(fresh_ctx, tmp_path, monkeypatch):
    """The call sequence of rvc/scripts/voice_conversion.py:71-96 -- Config, load_hubert(path), get_vc(path),
    rvc_infer(paths...) -- with CI's canonical arguments (test_cli.yml:43); RMVPE is NOT loaded by the caller
    (lazy load from rvc/models/predictors/rmvpe.pt, pipeline.py:123-126).  The WAV written equals the
    reference's VC.pipeline output (fixture pipeline_tiny_ciargs) within the PCM tolerance."""
    from scipy.io import wavfile
    from polgen_rvc_amd.infer import infer as I, pipeline as P
    d = np.load(os.path.join(GOLD, "pipeline_tiny_ciargs.npz"))
    hub_path, rmvpe_path, pth, wav, cfgs = _write_assets(tmp_path, d)
    monkeypatch.setattr(P, "RMVPE_DIR", rmvpe_path)
    config = I.Config()
    config.x_pad, config.x_query, config.x_center, config.x_max = [int(v) for v in d["geo"]]
    hubert_model = I.load_hubert(config.device, config.is_half, hub_path)
    cpt, version, net_g, tgt_sr, vc = I.get_vc(config.device, config.is_half, config, pth)
    assert not getattr(fresh_ctx, "rmvpe_loaded", False)
    vc.parity_noise = _pack_noise(d)          # test hook: the two randn_like draws of the reference run
    out = str(tmp_path / "out.mp3")           # WAV bytes whatever the extension, like infer.py:153
    I.rvc_infer(None, 0, wav, out, float(d["pitch"]), "rmvpe+", cpt, version, net_g, 3, tgt_sr,
                float(d["volume_envelope"]), float(d["protect"]), 128, vc, hubert_model, float(d["f0_min"]),
                float(d["f0_max"]))
    assert fresh_ctx.rmvpe_loaded
    sr, pcm = wavfile.read(out)
    assert sr == tgt_sr and pcm.dtype == np.int16 and pcm.shape == d["pcm"].shape
    diff = np.abs(pcm.astype(np.int32) - d["pcm"].astype(np.int32))
    print(f"rvc_infer e2e: pcm max diff {diff.max()} LSB, frac>1 {np.mean(diff > 1):.2e}")
    assert diff.max() <= 8 and np.mean(diff > 1) < 0.02
    # second request for the same files: everything is resident, nothing is parsed again
    h2 = I.load_hubert(config.device, config.is_half, hub_path)
    again = I.get_vc(config.device, config.is_half, config, pth)
    assert h2 is hubert_model and again[2] is net_g


@pytest.mark.parametrize("fp16,legacy", [(True, False), (False, True), (True, True)])
def test_real_checkpoint_containers_fp16_and_legacy_weight_norm(fresh_ctx, tmp_path, monkeypatch, fp16, legacy):
    """voice models are usually saved as fp16 tensors (infer.py:86-100 widens them with
    net_g.float()), older ones carry weight_g / weight_v instead of parametrizations.*.  Both load through get_vc
    and convert to (almost) the fp32 checkpoint's waveform: fp16 rounding of the weights only."""
    from polgen_rvc_amd import synthetic as S
    from polgen_rvc_amd.infer import infer as I, pipeline as P
    d = np.load(os.path.join(GOLD, "pipeline_tiny_single.npz"))
    hub_path, rmvpe_path, pth, wav, cfgs = _write_assets(tmp_path, d, fp16=fp16, legacy_wn=legacy)
    monkeypatch.setattr(P, "RMVPE_DIR", rmvpe_path)
    config = I.Config()
    hub = I.load_hubert(config.device, config.is_half, hub_path)
    cpt, version, net_g, tgt_sr, vc = I.get_vc(config.device, config.is_half, config, pth)
    audio = S.make_clip(int(d["clip"]), float(d["seconds"]))
    pcm, f32 = vc.pipeline(hub, net_g, 0, audio.astype(np.float64), "x.wav", 0.0, "rmvpe+", None, 0, 1, 3, tgt_sr, 0,
                           1.0, "v2", 0.33, 128, None, 50, 1100, noise=_pack_noise(d), return_f32=True)
    tp = int(tgt_sr)
    ref = d["raw"][tp:-tp]
    e = rms(f32 - ref)
    print(f"fp16={fp16} legacy={legacy}: float rms err {e:.3e} (rms {rms(ref):.3f})")
    assert pcm.shape == d["pcm"].shape
    assert e < (5e-3 if fp16 else 1e-4)          # fp16 weights: 2^-11 relative rounding per weight
    assert e > 0 or not fp16


def test_stage_methods_have_the_reference_meaning(ctx):
    """VC.get_f0 takes the padded + filtered signal and returns un-truncated (coarse, f0); VC.get_f0_rmvpe returns
    Hz; VC.vc turns one chunk of audio_pad into the un-trimmed float waveform -- all against the fixture the
    reference's own VC produced."""
    from polgen_rvc_amd import synthetic as S
    from polgen_rvc_amd.infer import infer as I
    d = np.load(os.path.join(GOLD, "pipeline_tiny_single.npz"))
    hcfg, rcfg, scfg = json.loads(str(d["cfgs"]))
    seed = int(d["seed"])
    I._CTX[0] = ctx
    hub = I.load_hubert("cuda:0", False, None, state=S.hubert_state(hcfg, seed), cfg=hcfg)
    I.load_rmvpe("cuda:0", state=S.rmvpe_state(rcfg, seed), cfg=rcfg)
    cpt = S.synth_checkpoint(scfg, seed)
    cpt["weight"] = S.synth_state(scfg, seed, input_dim=hcfg["embed_dim"])
    cpt, version, net_g, tgt_sr, vc = I.get_vc("cuda:0", False, I.Config(), None, cpt=cpt)
    audio = S.make_clip(int(d["clip"]), float(d["seconds"])).astype(np.float64)
    audio_pad = np.pad(ctx.highpass(audio), (vc.t_pad, vc.t_pad), mode="reflect")
    p_len = audio_pad.shape[0] // vc.window
    coarse, f0 = vc.get_f0("x.wav", audio_pad, p_len, 0.0, "rmvpe+", 3, 128, None, 50, 1100)
    assert coarse.shape == f0.shape == (p_len + 1,) and coarse.dtype.kind == "i"
    assert np.mean(coarse[:p_len] != d["coarse"]) < 1e-3
    raw_hz = vc.get_f0_rmvpe(audio_pad, f0_min=50, f0_max=1100)
    assert np.allclose(raw_hz, f0, rtol=1e-6)                 # pitch shift 0: get_f0's f0 is the estimate itself
    out = vc.vc(hub, net_g, 0, audio_pad, torch.from_numpy(coarse[:p_len]).long().unsqueeze(0),
                torch.from_numpy(f0[:p_len]).float().unsqueeze(0), None, None, 0, "v2", 0.33,
                z_noise=d["z_noise_0"], src_noise=d["src_noise_0"])
    assert out.dtype == np.float32 and out.shape == d["raw"].shape
    e = rms(out - d["raw"])
    print(f"VC.vc float rms err {e:.3e} (rms {rms(d['raw']):.3f})")
    assert e < 1e-4
    with pytest.raises(ValueError):
        vc.vc(hub, net_g, 0, audio_pad, None, None, None, None, 0, "v2", 0.33)
    with pytest.raises(ValueError):
        vc.get_f0("x.wav", audio_pad, p_len, 0.0, "pm", 3, 128)
    if not getattr(ctx, "crepe_loaded", False):    # torchcrepe's weights are not under rvc/models/predictors here
        with pytest.raises(FileNotFoundError):
            vc.get_f0("x.wav", audio_pad, p_len, 0.0, "mangio-crepe", 3, 128)
    if not getattr(ctx, "fcpe_loaded", False):     # no fcpe.pt under rvc/models/predictors here: as in the reference
        with pytest.raises(FileNotFoundError):
            vc.get_f0("x.wav", audio_pad, p_len, 0.0, "fcpe", 3, 128)


def test_f0_file_branch_of_get_f0(ctx):
    """pipeline.py:185-191: an f0 curve given as (time s, Hz) rows replaces the estimate from x_pad seconds on."""
    from polgen_rvc_amd import synthetic as S
    from polgen_rvc_amd.infer import infer as I
    I._CTX[0] = ctx
    I.load_rmvpe("cuda:0", state=S.rmvpe_state(S.RMVPE_CFG_TINY, 1), cfg=S.RMVPE_CFG_TINY)
    vc = I.VC(4800, I.Config())
    x = np.pad(S.make_clip(5, 1.5).astype(np.float64), (16000, 16000), mode="reflect")
    inp = np.stack([np.linspace(0.0, 1.0, 11), np.full(11, 200.0)], axis=1)
    coarse, f0 = vc.get_f0("x", x, len(x) // 160, 0.0, "rmvpe+", 3, 128, inp)
    assert np.allclose(f0[100:201], 200.0) and (coarse[100:201] == coarse[100]).all() and coarse[100] > 1
