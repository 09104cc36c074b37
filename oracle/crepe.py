"""CPU oracle of the reference's `mangio-crepe` F0 path -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(),
bench.py's cpu_baseline); the product path never imports it.

**Parity unpinned.**  The reference calls `torchcrepe.predict` (rvc/infer/pipeline.py:86-117; requirements.txt pins
torchcrepe==0.0.23, librosa for the Viterbi decoder) -- neither package is vendored in /root/reference nor installed
here, and torchcrepe's weights (assets/full.pth) are not available offline.  This file restates the published algorithm
of torchcrepe 0.0.23 (`core.predict / preprocess / infer / postprocess`, `model.Crepe`, `decode.viterbi`,
`convert.*`) and of `librosa.sequence.viterbi` (0.9.x); it is anchored on the reference's call site only:
    torchcrepe.predict(audio, 16000, hop_length, f0_min, f0_max, "full", batch_size=hop_length * 2, device, pad=True)
i.e. decoder = Viterbi (the default of 0.0.23), one Viterbi pass PER BATCH of 2 * hop_length frames, and the random
+-20 cent triangular dither of `convert.bins_to_cents` (scipy's global RNG there; an explicit array here).

State dict keys are torchcrepe's: conv{1..6}.weight (Cout, Cin, K, 1) / .bias, conv{1..6}_BN.{weight,bias,running_mean,
running_var}, classifier.{weight,bias}.  Capacities: "full" (1024,128,128,128,256,512 filters) and "tiny" (128,16,16,16,32,64).
"""
import numpy as np
import torch
import torch.nn.functional as F

SAMPLE_RATE = 16000
WINDOW_SIZE = 1024
PITCH_BINS = 360
CENTS_PER_BIN = 20
CENTS_OFFSET = 1997.3794084376191
BN_EPS = 0.0010000000474974513


def frequency_to_bins(freq: float, ceil: bool = False) -> int:
    """convert.frequency_to_bins on a float32 scalar tensor (floor by default, torch.ceil for fmax)."""
    cents = np.float32(1200.0) * np.log2(np.float32(freq) / np.float32(10.0), dtype=np.float32)
    bins = (cents - np.float32(CENTS_OFFSET)) / np.float32(CENTS_PER_BIN)
    return int(np.ceil(bins) if ceil else np.floor(bins))


def model_forward(sd, frames: torch.Tensor) -> torch.Tensor:
    """model.Crepe.forward: six {pad, conv, relu, batch-norm (eval), max-pool 2} layers, Linear, sigmoid."""
    x = frames[:, None, :, None]
    for i in range(1, 7):
        pad = (0, 0, 254, 254) if i == 1 else (0, 0, 31, 32)
        x = F.pad(x, pad)
        x = F.conv2d(x, sd[f"conv{i}.weight"], sd[f"conv{i}.bias"], stride=(4, 1) if i == 1 else (1, 1))
        x = F.relu(x)
        x = F.batch_norm(x, sd[f"conv{i}_BN.running_mean"], sd[f"conv{i}_BN.running_var"], sd[f"conv{i}_BN.weight"],
                         sd[f"conv{i}_BN.bias"], False, 0.0, BN_EPS)
        x = F.max_pool2d(x, (2, 1), (2, 1))
    in_features = sd["classifier.weight"].shape[1]
    x = x.permute(0, 2, 1, 3).reshape(-1, in_features)
    return torch.sigmoid(F.linear(x, sd["classifier.weight"], sd["classifier.bias"]))


def frames_of(audio: torch.Tensor, hop: int, start_frame: int, n_frames: int) -> torch.Tensor:
    """core.preprocess for one batch: `audio` (1, n) is ALREADY padded by 512 on both sides; zero-mean / unit-std frames."""
    start = start_frame * hop
    end = min(audio.shape[1], (start_frame + n_frames - 1) * hop + WINDOW_SIZE)
    fr = F.unfold(audio[:, None, None, start:end], kernel_size=(1, WINDOW_SIZE), stride=(1, hop))
    fr = fr.transpose(1, 2).reshape(-1, WINDOW_SIZE).clone()
    fr -= fr.mean(dim=1, keepdim=True)
    fr /= torch.max(torch.tensor(1e-10), fr.std(dim=1, keepdim=True))
    return fr


_TRANSITION = None


def transition_matrix() -> np.ndarray:
    global _TRANSITION
    if _TRANSITION is None:
        xx, yy = np.meshgrid(range(PITCH_BINS), range(PITCH_BINS))
        t = np.maximum(12 - abs(xx - yy), 0)
        _TRANSITION = t / t.sum(axis=1, keepdims=True)
    return _TRANSITION


def viterbi_path(prob: np.ndarray, transition: np.ndarray) -> np.ndarray:
    """librosa.sequence.viterbi(prob (n_states, n_steps) float32, transition float64), uniform p_init."""
    n_states, n_steps = prob.shape
    eps = np.finfo(prob.dtype).tiny
    log_trans = np.log(transition + eps)
    log_prob = np.log(prob.T + eps)
    log_p_init = np.log(np.full(n_states, 1.0 / n_states) + eps)
    values = np.zeros((n_steps, n_states), dtype=float)
    ptr = np.zeros((n_steps, n_states), dtype=np.int64)
    values[0] = log_prob[0] + log_p_init
    lt = log_trans.T
    for t in range(1, n_steps):
        trans_out = values[t - 1][None, :] + lt            # [to, from]
        ptr[t] = np.argmax(trans_out, axis=1)
        values[t] = log_prob[t] + trans_out[np.arange(n_states), ptr[t]]
    states = np.zeros(n_steps, dtype=np.int64)
    states[-1] = np.argmax(values[-1])
    for t in range(n_steps - 2, -1, -1):
        states[t] = ptr[t + 1, states[t + 1]]
    return states


def decode_batch(prob: torch.Tensor, fmin: float, fmax: float, noise: np.ndarray):
    """core.postprocess + decode.viterbi + convert.bins_to_frequency for one batch: prob (n_frames, 360) sigmoid outputs,
    noise (n_frames,) the triangular dither in cents.  Returns (bins, pitch float32)."""
    p = prob.T.clone()                                    # (360, n_frames)
    p[:frequency_to_bins(fmin)] = -float("inf")
    p[frequency_to_bins(fmax, ceil=True):] = -float("inf")
    sm = torch.softmax(p, dim=0).numpy()
    bins = viterbi_path(sm, transition_matrix())
    cents = torch.tensor(bins) * CENTS_PER_BIN + CENTS_OFFSET          # int64 tensor -> float32 (torch promotion)
    cents = cents + torch.tensor(np.asarray(noise), dtype=cents.dtype)
    pitch = 10 * 2 ** (cents / 1200)
    return bins, pitch.numpy()


@torch.no_grad()
def predict(sd, audio: np.ndarray, hop: int, fmin: float, fmax: float, noise: np.ndarray, batch_size: int = None,
            return_parts: bool = False):
    """torchcrepe.predict(audio[None], 16000, hop, fmin, fmax, model, batch_size, pad=True) with the Viterbi decoder."""
    a = torch.from_numpy(np.asarray(audio, np.float32))[None]
    total = 1 + a.shape[1] // hop
    a = F.pad(a, (WINDOW_SIZE // 2, WINDOW_SIZE // 2))
    batch_size = total if batch_size is None else batch_size
    pitch, bins, probs = [], [], []
    for i in range(0, total, batch_size):
        nb = min(batch_size, total - i)
        prob = model_forward(sd, frames_of(a, hop, i, nb))
        b, p = decode_batch(prob, fmin, fmax, noise[i:i + nb])
        pitch.append(p)
        bins.append(b)
        probs.append(prob.numpy())
    pitch = np.concatenate(pitch).astype(np.float32)
    if return_parts:
        return pitch, dict(bins=np.concatenate(bins), probs=np.concatenate(probs))
    return pitch


def n_frames(n_samples: int, hop: int) -> int:
    return 1 + n_samples // hop


def get_f0_crepe(sd, x: np.ndarray, f0_min: float, f0_max: float, p_len: int, hop: int, noise: np.ndarray,
                 return_parts: bool = False):
    """VC.get_f0_crepe (pipeline.py:86-117)."""
    x = np.asarray(x).astype(np.float32)
    x = x / np.quantile(np.abs(x), 0.999)
    out = predict(sd, x, hop, f0_min, f0_max, noise, batch_size=hop * 2, return_parts=return_parts)
    pitch, parts = out if return_parts else (out, None)
    p_len = p_len or x.shape[0] // hop
    source = np.array(pitch)
    source[source < 0.001] = np.nan
    target = np.interp(np.arange(0, len(source) * p_len, len(source)) / p_len, np.arange(0, len(source)), source)
    f0 = np.nan_to_num(target)
    if return_parts:
        parts["pitch"] = pitch
        return f0, parts
    return f0
