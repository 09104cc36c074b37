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
    # every public method of the reference's VC (rvc/infer/pipeline.py:65-467), positional part identical; extras
    # must be keyword-only so that positional calls keep their meaning
    ref_vc = {
        "__init__": ["self", "tgt_sr", "config"],
        "get_f0_crepe": ["self", "x", "f0_min", "f0_max", "p_len", "hop_length", "model"],
        "get_f0_rmvpe": ["self", "x", "f0_min", "f0_max", "args", "kwargs"],
        "get_f0": ["self", "input_audio_path", "x", "p_len", "pitch", "f0_method", "filter_radius", "hop_length",
                   "inp_f0", "f0_min", "f0_max"],
        "vc": ["self", "model", "net_g", "sid", "audio0", "pitch", "pitchf", "index", "big_npy", "index_rate",
               "version", "protect"],
        "pipeline": pp,
    }
    for name, want in ref_vc.items():
        params = inspect.signature(getattr(P.VC, name)).parameters
        got = list(params)
        assert got[:len(want)] == want, (name, got)
        for extra in got[len(want):]:
            assert params[extra].kind is inspect.Parameter.KEYWORD_ONLY, (name, extra)
    sig = inspect.signature(P.VC.get_f0).parameters
    assert (sig["inp_f0"].default, sig["f0_min"].default, sig["f0_max"].default) == (None, 50, 1100)
    sig = inspect.signature(P.VC.get_f0_rmvpe).parameters
    assert (sig["f0_min"].default, sig["f0_max"].default) == (1, 40000)
    assert P.RMVPE_DIR == os.path.join(os.getcwd(), "rvc", "models", "predictors", "rmvpe.pt")   # pipeline.py:14-16
    assert P.FCPE_DIR == os.path.join(os.getcwd(), "rvc", "models", "predictors", "fcpe.pt")
    assert set(P.F0_METHODS) == {"rmvpe+", "rmvpe", "fcpe", "mangio-crepe"}      # pipeline.py:142-181 (+ the "rmvpe" alias)
    assert inspect.signature(P.VC.get_f0_crepe).parameters["model"].default == "full"
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


def test_restricted_unpickler_stubs_dangerous_globals(tmp_path):
    """a crafted hubert_base.pt must not reach builtins.eval / exec / getattr / __import__ or arbitrary
    torch / numpy callables through find_class -- only the exact tensor-rebuild allow-list resolves."""
    import io

    class Evil:
        def __reduce__(self):
            return (eval, ("__import__('os').environ.__setitem__('RVCX_PWNED', '1')",))
    buf = io.BytesIO()
    pickle.dump({"model": {"encoder.pos_conv.0.weight_g": Evil()}}, buf)
    buf.seek(0)
    os.environ.pop("RVCX_PWNED", None)
    out = ckpt_io._StubUnpickler(buf).load()
    assert "RVCX_PWNED" not in os.environ
    assert isinstance(out["model"]["encoder.pos_conv.0.weight_g"], ckpt_io._Stub)
    for mod, name in [("builtins", "eval"), ("builtins", "exec"), ("builtins", "getattr"), ("builtins", "__import__"),
                      ("os", "system"), ("torch", "load"), ("numpy", "load"), ("torch.hub", "load")]:
        assert issubclass(ckpt_io._StubUnpickler(io.BytesIO(b"")).find_class(mod, name), ckpt_io._Stub)
    assert ckpt_io._StubUnpickler(io.BytesIO(b"")).find_class("collections", "OrderedDict").__name__ == "OrderedDict"


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


def test_index_io_ivf_flat_roundtrip(tmp_path):
    """A faiss "IVF{nlist},Flat" byte stream (tests/faiss_writer.py, the layout RVC's training writes): vectors come
    back in id order as reconstruct_n returns them, with the coarse centroids, the list of every vector and nprobe;
    both encodings of the list sizes ("full" / "sprs"); an empty list is fine."""
    import faiss_writer as FW
    g = np.random.Generator(np.random.PCG64(4))
    big = S.make_index(300, 24, 2)
    cent = g.standard_normal((7, 24)).astype(np.float32)
    d2 = ((big[:, None, :].astype(np.float64) - cent[None].astype(np.float64)) ** 2).sum(-1)
    d2[:, 5] = np.inf                                     # list 5 stays empty
    assign = np.argmin(d2, axis=1).astype(np.int32)
    for sparse in (False, True):
        q = os.path.join(tmp_path, f"ivf{int(sparse)}.index")
        open(q, "wb").write(FW.ivf_flat_bytes(big, cent, assign, nprobe=1, sparse_sizes=sparse))
        ix = index_io.read_index(q)
        assert ix.is_ivf and ix.nprobe == 1
        assert np.array_equal(ix.vectors, big) and np.array_equal(ix.centroids, cent) and np.array_equal(ix.assign, assign)
        assert np.array_equal(index_io.read_index_vectors(q), big)
    q = os.path.join(tmp_path, "flat.index")
    open(q, "wb").write(FW.flat_bytes(big))
    ix = index_io.read_index(q)
    assert not ix.is_ivf and np.array_equal(ix.vectors, big)


def test_oracle_ivf_search_differs_from_flat_where_it_should():
    """nprobe = 1 is not brute force: a query whose true nearest neighbours sit in a neighbouring list gets other
    ids.  The oracle restates both rules; the padding case (a list shorter than 8) yields id -1 with weight 0."""
    from oracle import pipeline as OP
    g = np.random.Generator(np.random.PCG64(7))
    big = g.standard_normal((500, 16)).astype(np.float32)
    cent = g.standard_normal((12, 16)).astype(np.float32)
    d2 = ((big[:, None, :].astype(np.float64) - cent[None].astype(np.float64)) ** 2).sum(-1)
    assign = np.argmin(d2, axis=1).astype(np.int32)
    q = g.standard_normal((64, 16)).astype(np.float32)
    _, flat_ids, _ = OP.index_blend(q, big, 0.5)
    out, ivf_ids, score = OP.index_blend_ivf(q, big, cent, assign, 0.5)
    assert (np.sort(flat_ids, 1) != np.sort(ivf_ids, 1)).any()
    qlist = np.argmin(((q[:, None, :].astype(np.float64) - cent[None].astype(np.float64)) ** 2).sum(-1), axis=1)
    for t in range(len(q)):
        real = ivf_ids[t][ivf_ids[t] >= 0]
        assert (assign[real] == qlist[t]).all()
        assert len(real) == min(8, int((assign == qlist[t]).sum()))
    assert np.isfinite(out[(ivf_ids >= 0).any(1)]).all()


def test_shard_is_a_balanced_partition():
    lens = [5, 30, 12, 7, 30, 9, 15, 3, 11]
    parts = [D.shard(len(lens), r, 4, lens) for r in range(4)]
    assert sorted(sum(parts, [])) == list(range(len(lens)))
    assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_make_clip_is_deterministic():
    a, b = S.make_clip(3, 1.0), S.make_clip(3, 1.0)
    assert a.dtype == np.float32 and a.shape == (16000,) and np.array_equal(a, b) and np.abs(a).max() < 1.0


class _FakeCtx:
    """Counts loads/unloads so the residency cache can be tested without a GPU."""

    def __init__(self):
        import threading
        self.loads = []
        self._h = None
        self.lock = threading.RLock()      # what _lib.Context carries (the mirror holds it around its call sequences)

    def load_hubert(self, cfg, state):
        self.loads.append("hubert")

    def load_rmvpe(self, cfg, state):
        self.loads.append("rmvpe")

    def load_synth(self, cfg, state):
        self.loads.append("synth")
        return len(self.loads)


def test_resident_asset_cache_is_keyed_by_path_and_mtime(tmp_path, monkeypatch):
    """a checkpoint path loaded before (same realpath/mtime/size) is not reloaded; a touched
    file is; voice models are kept least-recently-used up to MAX_RESIDENT_SYNTHS."""
    import os
    from polgen_rvc_amd import synthetic as S, weights as W, ckpt_io
    from polgen_rvc_amd.infer import infer as I
    fake = _FakeCtx()
    monkeypatch.setitem(I._CTX, 0, fake)
    I.clear_cache()
    monkeypatch.setattr(W, "hubert_cfg_struct", lambda cfg: None)
    monkeypatch.setattr(W, "rmvpe_cfg_struct", lambda cfg: None)
    monkeypatch.setattr(W, "synth_cfg_struct", lambda cfg, d: None)
    monkeypatch.setattr(W, "hubert_cfg_from_state", lambda st: {})
    monkeypatch.setattr(W, "rmvpe_cfg_from_state", lambda st: {})
    monkeypatch.setattr(ckpt_io, "load_fairseq_hubert", lambda p: {})
    cpt = S.synth_checkpoint(S.SYNTH_CFG_TINY, 0)
    cpt["weight"] = {"emb_g.weight": np.zeros((5, 4), np.float32),
                     "enc_p.emb_phone.weight": np.zeros((48, 32), np.float32)}
    monkeypatch.setattr(I, "_torch_load", lambda p: {k: (dict(v) if isinstance(v, dict) else
                                                         (list(v) if isinstance(v, list) else v)) for k, v in cpt.items()})
    hub, rm = tmp_path / "hubert_base.pt", tmp_path / "rmvpe.pt"
    hub.write_bytes(b"x" * 10)
    rm.write_bytes(b"y" * 10)
    h1 = I.load_hubert("cuda:0", False, str(hub))
    h2 = I.load_hubert("cuda:0", False, str(hub))
    I.load_rmvpe("cuda:0", str(rm))
    I.load_rmvpe("cuda:0", str(rm))
    assert h1 is h2 and fake.loads == ["hubert", "rmvpe"]
    os.utime(hub, ns=(1, 1))                                   # "new" file behind the same path
    h3 = I.load_hubert("cuda:0", False, str(hub))
    assert h3 is not h1 and fake.loads.count("hubert") == 2
    # voice models
    paths = []
    for i in range(I.MAX_RESIDENT_SYNTHS + 2):
        p = tmp_path / f"voice{i}.pth"
        p.write_bytes(bytes([i]) * 8)
        paths.append(str(p))
    cfg = I.Config()
    a = I.get_vc("cuda:0", False, cfg, paths[0])
    b = I.get_vc("cuda:0", False, cfg, paths[0])
    assert a[2] is b[2] and fake.loads.count("synth") == 1      # same resident model
    assert b[0]["config"][-1] == a[3] == 4800 and "weight" not in b[0] and b[1] == "v2"
    for p in paths[1:]:
        I.get_vc("cuda:0", False, cfg, p)
    assert len(I._SYNTHS) == I.MAX_RESIDENT_SYNTHS
    n = fake.loads.count("synth")
    I.get_vc("cuda:0", False, cfg, paths[0])                    # evicted meanwhile -> loaded again
    assert fake.loads.count("synth") == n + 1
    I.clear_cache()


def test_mirror_holds_the_context_lock_around_a_request_sequence(tmp_path, monkeypatch):
    """VERDICT r4 item 3, the host half (no GPU needed): VC.pipeline is a SEQUENCE of calls on the shared context -- make the
    request's index resident, then convert with it.  Two threads with two different index files on one (fake) context: the
    per-context lock the mirror holds (infer/pipeline.py: _with_ctx_lock) must keep every convert paired with its own
    thread's index, however the scheduler interleaves them; the loaders take context lock then table lock in that order."""
    import threading
    import time
    from polgen_rvc_amd.infer import infer as I, pipeline as P, _state

    class Ctx(_FakeCtx):
        def __init__(self):
            super().__init__()
            self.resident, self.pairs, self.rmvpe_loaded = None, [], True
            self.busy = 0
            self.overlap = False

        def load_index(self, big):
            self.resident = None if big is None else float(big[0, 0])
            time.sleep(0.002)                       # a window for the other thread to slip in

        def convert_batch(self, model_id, clips, p, noise, want_f32=False, inp_f0=None, crepe_dither=None):
            self.busy += 1
            self.overlap |= self.busy > 1
            time.sleep(0.001)
            self.pairs.append((float(clips[0][0]), self.resident, p.index_rate))
            self.busy -= 1
            return [np.zeros(4, np.int16)]

    ctx = Ctx()
    monkeypatch.setitem(_state._CTX, 0, ctx)
    monkeypatch.setattr(_state, "_INDEX_RESIDENT", {})
    monkeypatch.setattr(P, "_INDEX_RESIDENT", _state._INDEX_RESIDENT)
    net_g, hub = I.SynthHandle(ctx, 0, []), I.HubertHandle(ctx, {})
    net_g.__class__.__del__ = lambda self: None
    paths = []
    for k in range(2):
        path = tmp_path / f"idx{k}.npy"
        np.save(path, np.full((16, 8), float(k + 1), np.float32))
        paths.append(str(path))
    errs = []

    def work(k):
        try:
            for _ in range(25):
                vc = P.VC(4800, I.Config())
                vc.pipeline(hub, net_g, 0, np.full(100, float(k + 1)), "x.wav", 0.0, "rmvpe+", paths[k], 0.5, 1, 3, 4800, 0,
                            1.0, "v2", 0.33, 128, None)
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert len(ctx.pairs) == 50 and not ctx.overlap
    assert all(clip == resident and rate == 0.5 for clip, resident, rate in ctx.pairs), ctx.pairs[:6]


def test_load_audio_and_convert_to_stereo_host_side(tmp_path):
    """my_utils.py:5-16 / voice_conversion.py:45-51 without a rate change (the resampler itself is a GPU kernel:
    tests/test_gpu_audio.py): PCM scaling, mono mean, mono -> stereo doubling at the ORIGINAL rate, PCM_16 output, and a
    clear error for containers this image cannot decode."""
    from scipy.io import wavfile
    from oracle import audio as OA
    from polgen_rvc_amd.infer import audio as A, infer as I
    g = np.random.Generator(np.random.PCG64(2))
    left, right = g.uniform(-0.9, 0.9, 16000), g.uniform(-0.9, 0.9, 16000)
    pcm = (np.stack([left, right], 1) * 32767).astype(np.int16)
    wavfile.write(tmp_path / "st.wav", 16000, pcm)
    y = I.load_audio(str(tmp_path / "st.wav"), 16000)
    want = OA.load_audio_from_array(pcm.astype(np.float64) / 32768.0, 16000, 16000)
    assert y.dtype == np.float64 and np.array_equal(y, want)
    wavfile.write(tmp_path / "f32.wav", 16000, left.astype(np.float32))
    assert np.array_equal(I.load_audio(' "' + str(tmp_path / "f32.wav") + '" ', 16000), left.astype(np.float32).astype(np.float64))
    # convert_to_stereo: mono 44.1 kHz in -> two identical channels, still 44.1 kHz, 16-bit
    mono = (left[:4410] * 32767).astype(np.int16)
    wavfile.write(tmp_path / "mono.wav", 44100, mono)
    A.convert_to_stereo(str(tmp_path / "mono.wav"), str(tmp_path / "stereo.wav"))
    sr, out = wavfile.read(tmp_path / "stereo.wav")
    ref = OA.convert_to_stereo_array(mono.astype(np.float64) / 32768.0)
    assert sr == 44100 and out.dtype == np.int16 and out.shape == (4410, 2) == ref.shape
    assert np.array_equal(out[:, 0], out[:, 1]) and np.abs(out[:, 0].astype(np.int32) - mono).max() <= 1
    A.convert_to_stereo(str(tmp_path / "st.wav"), str(tmp_path / "st2.wav"))       # already stereo: unchanged
    assert np.abs(wavfile.read(tmp_path / "st2.wav")[1].astype(np.int32) - pcm).max() <= 1
    # compressed containers without soundfile: FLAC is decoded by the library itself (round 5, csrc/flac.hip) -- a damaged
    # stream is an error that names the reason; mp3 has no decoder in this image -> RuntimeError saying so (my_utils.py:14 wraps both)
    for name, magic, why in (("a.flac", b"fLaC\x00\x00\x00\x22", "flac"), ("a.mp3", b"ID3\x04\x00\x00\x00\x00\x00\x00", "no decoder")):
        open(tmp_path / name, "wb").write(magic + bytes(64))
        try:
            import soundfile  # noqa: F401
        except ImportError:
            with pytest.raises(RuntimeError, match=why):
                I.load_audio(str(tmp_path / name), 16000)


def test_bench_refuses_more_gpus_than_visible():
    """bench.py --gpus N must stop before touching anything when the node does not have N GPUs (this container: 0),
    as a torch.distributed.run worker and as its own launcher (tests/test_dist_gloo.py drives the launcher itself)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)
    # invoked plainly (no WORLD_SIZE) it launches its own ranks -- but never past the visible GPU count
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)


def test_fcpe_cfg_is_read_off_the_checkpoint():
    """FCPEInfer builds the module from fcpe.pt's config block (FCPE.py:715-733); the shapes must say the same,
    and give the module defaults the block does not hold (heads x dim_head, random features, depth-wise kernel)."""
    from polgen_rvc_amd import synthetic as S, weights as W
    for cfg in (S.FCPE_CFG_TINY, S.FCPE_CFG_FULL):
        ck = S.fcpe_checkpoint(cfg, 0) if cfg is S.FCPE_CFG_TINY else None
        sd = ck["model"] if ck else {k: np.zeros(v, np.float32) for k, v in _fcpe_shapes(cfg).items()}
        got = W.fcpe_cfg_from_state(sd, ck["config"] if ck else None)
        assert (got["n_layers"], got["n_chans"]) == (cfg["n_layers"], cfg["n_chans"])
        assert (got["heads"], got["dim_head"], got["nb_features"], got["dw_kernel"], got["out_dims"]) == (8, 64, 266, 31, 360)
        st = W.fcpe_cfg_struct(got)
        assert (st.n_layers, st.n_chans, st.mel_fmax) == (cfg["n_layers"], cfg["n_chans"], 8000.0)
    bad = S.fcpe_checkpoint(S.FCPE_CFG_TINY, 0)
    bad["config"]["mel"]["hop_size"] = 256
    with pytest.raises(ValueError, match="mel front end"):
        W.fcpe_cfg_from_state(bad["model"], bad["config"])


def _fcpe_shapes(cfg):
    C, L = cfg["n_chans"], cfg["n_layers"]
    s = {"stack.0.weight": (C, 128, 3), "dense_out.weight_g": (360, 1), "dense_out.weight_v": (360, C)}
    for i in range(L):
        p = f"decoder._layers.{i}"
        s[p + ".norm.weight"] = (C,)
        s[p + ".attn.to_q.weight"] = (512, C)
        s[p + ".attn.fast_attention.projection_matrix"] = (266, 64)
        s[p + ".conformer.net.4.conv.weight"] = (2 * C, 1, 31)
    return s


def test_f0_file_track_equals_numpy_interp():
    """VC.get_f0's f0-file branch (pipeline.py:186-189) behind the C ABI: delta_t in float32 arithmetic, np.interp in
    float64 -- bit for bit what numpy gives on the same table (host code of librvcx.so, no GPU call)."""
    from oracle import pipeline as OP
    from polgen_rvc_amd import _lib
    g = np.random.Generator(np.random.PCG64(17))
    for rows in (1, 2, 7, 300):
        t = np.sort(g.uniform(0.0, 4.0, rows)).astype(np.float32)
        if rows > 3:
            t[3] = t[2]                                            # a repeated time stamp
        tab = np.stack([t, g.uniform(80.0, 400.0, rows).astype(np.float32)], 1)
        delta_t = np.round((tab[:, 0].max() - tab[:, 0].min()) * 100 + 1).astype("int16")
        want = np.interp(list(range(delta_t)), tab[:, 0] * 100, tab[:, 1])
        got = _lib.f0_file_track(tab)
        assert got.shape == want.shape and np.array_equal(got, want), rows
    # and the oracle's get_f0 tail uses exactly that track
    f0 = np.full(500, 100.0)
    tab = np.array([[0.0, 200.0], [1.0, 300.0]], np.float32)
    coarse, bak = OP.f0_to_coarse(f0, 0.0, 50, 1100, tab, 1)
    assert bak[99] == 100.0 and bak[100] == 200.0 and bak[200] == 300.0 and bak[201] == 100.0
    assert coarse[150] == OP.f0_to_coarse(np.array([250.0]), 0.0)[0][0]


def test_unsupported_index_files_raise_instead_of_converting_without_index(tmp_path):
    """an index faiss would read and search but rvcx does not implement (other index classes, IVF with
    nprobe != 1) must not degrade silently to "no index"; a corrupt file does, like the reference (pipeline.py:321-326)."""
    import struct
    import faiss_writer as FW
    from polgen_rvc_amd.infer.pipeline import VC

    class FakeCtx:
        def __init__(self):
            self.loaded = []

        def load_index(self, a):
            self.loaded.append(None if a is None else "flat")

        def load_index_ivf(self, *a):
            self.loaded.append("ivf")
    vc = VC(48000, I.Config())
    big = S.make_index(64, 8, 0)
    cent = big[:4].copy()
    assign = (np.arange(64) % 4).astype(np.int32)
    p2 = os.path.join(tmp_path, "nprobe2.index")
    open(p2, "wb").write(FW.ivf_flat_bytes(big, cent, assign, nprobe=2))
    with pytest.raises(index_io.UnsupportedIndex):
        vc._load_index(FakeCtx(), p2, 0.5)
    pq = os.path.join(tmp_path, "pq.index")
    open(pq, "wb").write(b"IwPQ" + struct.pack("<iqqqBi", 8, 64, 1 << 20, 1 << 20, 1, 1))
    with pytest.raises(index_io.UnsupportedIndex):
        vc._load_index(FakeCtx(), pq, 0.5)
    bad = os.path.join(tmp_path, "garbage.index")
    open(bad, "wb").write(b"\x00\x01\x02")
    c = FakeCtx()
    assert vc._load_index(c, bad, 0.5) == (None, None) and c.loaded[-1] is None
    ok = os.path.join(tmp_path, "ok.index")
    open(ok, "wb").write(FW.ivf_flat_bytes(big, cent, assign, nprobe=1))
    c = FakeCtx()
    h, flag = vc._load_index(c, ok, 0.5)
    assert h is not None and flag is True and c.loaded == ["ivf"]


def test_broadcast_refuses_device_regions_on_a_cpu_backend():
    """dist._view must never wrap hipMalloc'd addresses as host memory."""
    import torch.distributed as dist

    class Ctx:
        regions_on_device = True

        def weights_regions(self):
            return [(0x7F0000000000, 4096)], 1
    if dist.is_initialized():
        pytest.skip("a process group is already up")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        with pytest.raises(RuntimeError, match="device-resident"):
            D.broadcast_weights(Ctx(), 0, 0, force=True)
    finally:
        dist.destroy_process_group()


def test_calls_that_name_no_device_use_the_gpu_this_process_serves(monkeypatch):
    """load_audio resampled on a hard-wired "cuda:0" -- a rank serving cuda:N would have built a second full
    context on GPU 0.  The default device is the resident context's, else LOCAL_RANK, else 0."""
    import polgen_rvc_amd  # noqa: F401
    from polgen_rvc_amd.infer import _state
    monkeypatch.setattr(_state, "_CTX", {})
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    assert _state.default_device() == 0
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert _state.default_device() == 5
    monkeypatch.setattr(_state, "_CTX", {3: object()})
    assert _state.default_device() == 3
    import inspect
    from polgen_rvc_amd.infer import infer
    sig = inspect.signature(infer.load_audio)
    assert list(sig.parameters)[:2] == ["file", "sample_rate"]
    assert sig.parameters["device"].kind is inspect.Parameter.KEYWORD_ONLY


def test_shape_only_placeholders_match_the_real_layouts():
    """bench.py's ranks != 0 load zeros of the right shapes without drawing random numbers (synthetic.shapes_only)."""
    import numpy as np
    from polgen_rvc_amd import synthetic as S
    real = S.rmvpe_state(S.RMVPE_CFG_TINY, 3)
    with S.shapes_only():
        ph = S.rmvpe_state(S.RMVPE_CFG_TINY, 3)
        ph_s = S.synth_state(S.SYNTH_CFG_TINY, 3, input_dim=128)
    again = S.rmvpe_state(S.RMVPE_CFG_TINY, 3)
    assert set(ph) == set(real) and all(ph[k].shape == real[k].shape and ph[k].dtype == real[k].dtype for k in real)
    assert all(not np.any(v) for k, v in ph.items() if "running_var" not in k and k.endswith(".weight") and v.ndim > 1)
    assert all(np.array_equal(real[k], again[k]) for k in real)          # the mode does not leak out of the with-block
    real_s = S.synth_state(S.SYNTH_CFG_TINY, 3, input_dim=128)
    assert set(ph_s) == set(real_s) and all(ph_s[k].shape == real_s[k].shape for k in real_s)


def test_get_vc_accepts_v1_and_refuses_a_version_that_contradicts_the_weights():
    """rvc/infer/infer.py:91-97: input_dim = 768 if version == "v2" else 256.  A checkpoint whose ``version`` contradicts the
    width of enc_p.emb_phone is refused before anything touches the GPU (the reference dies later, in load_state_dict)."""
    from polgen_rvc_amd import synthetic as S
    from polgen_rvc_amd.infer import infer as I
    cfg = S.SYNTH_CFG_TINY
    for version, dim in (("v2", 256), ("v1", 768)):
        cpt = S.synth_checkpoint(cfg, 1, version=version)
        cpt["weight"] = S.synth_state(cfg, 1, input_dim=dim)
        with pytest.raises(ValueError, match="emb_phone"):
            I.get_vc("cuda:0", False, I.Config(), None, cpt=cpt)
    cpt = S.synth_checkpoint(cfg, 1, version="v3")
    with pytest.raises(ValueError, match="version"):
        I.get_vc("cuda:0", False, I.Config(), None, cpt=cpt)
