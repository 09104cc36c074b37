#!/usr/bin/env python3
"""bench.py -- real-time factor of the rvc/infer hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: invoked plainly, this process -- before it touches any GPU -- starts N worker processes (one
per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, RCCL rendezvous on 127.0.0.1), relays rank 0's JSON line and
exits non-zero if any worker did; invoked by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
it is a worker already (WORLD_SIZE in the environment).

Workload c2 (default, BASELINE.json configs[1]): one 30 s 16 kHz mono clip per step per GPU, RVC v2 48 kHz voice
model, rmvpe+ F0, contentvec-shaped HuBERT-base, index_rate 0, fp32, chunk geometry (1,6,38,41);
synthetic clip + synthetic weights in the real checkpoint layouts (no real weights exist offline).
Workload c3 (--workload c3, BASELINE.json configs[2]): a batch of 64 x 30 s clips per step, index_rate 0.75 with
a 65 536 x 768 retrieval matrix resident in HBM.  The default (c2, N = 1) run also carries a "c3" object measured by
a child process of the same run, like "exact_fp32".  With --gpus 8 this is BASELINE.json configs[3] (C4: 512 clips, 64 per rank).
Workload c5 (--workload c5, BASELINE.json configs[4]): 256 utterances of U(3, 15) s per step for the whole job, a
40 k and a 48 k voice model resident beside one HuBERT and one RMVPE; the utterances are sharded over the ranks by
length (dist.shard), odd ones go to the 48 k model, even ones to the 40 k model; mixed lengths are converted as ragged
micro-batches (length classes).  The default run carries a "c5" object too (child process).
A step = VC.pipeline on the step's clip(s) as SURVEY.md 8(d) defines the metric: H2D of the float PCM (pinned
host memory), every kernel, D2H of the int16 PCM -- all inside the timed region.
Weak scaling (c2 / c3): every rank converts its own clip(s) per step; value = all ranks' audio seconds / max-rank wall
(whole-job aggregate, as the driver contract asks; value_per_gpu = value / n_gpus).  c5 is strong scaling.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import polgen_rvc_amd  # noqa: E402
from polgen_rvc_amd import _lib, dist as D, synthetic as S, weights as W  # noqa: E402

PEAK_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 MFMA (v_mfma_f32_32x32x2_f32) = fp32 vector peak
PEAK_F16_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense fp16/bf16 MFMA (v_mfma_f32_32x32x16_f16)
# conv_h3 kernels form every fp32 product block from three fp16 MFMAs (hi/lo split, fp32 accumulate): the ceiling
# for ALGORITHMIC conv FLOPs on them is a third of the fp16 MFMA peak.
PEAK_H3_TFLOPS = PEAK_F16_TFLOPS / 3.0
CLIP_SECONDS = 30.0
C3_BATCH = 64
C3_INDEX_ROWS = 65536
C5_UTTERANCES = 256
CPU_SAMPLE_SECONDS = 30.0    # the clip the metric is quoted on (attention is O(T^2): a shorter sample would flatter the CPU)
PMC_FILES = ("pmc_traffic_r06.json", "pmc_traffic_r05.json", "pmc_traffic_r04.json", "pmc_traffic_r03.json")     # newest first
WORKER_DEADLINE_S = 3600.0   # launch_workers: the whole multi-rank run
# algorithmic HBM bytes of one launch of the dominant kernel (fused ResBlock step, NSF stage 2 of a 32 s chunk at 48 k:
# C = 128 channels x 383 760 positions x 4 B, read x once + write y once); DESIGN.md "Algorithmic work per unit"
DOM_ALGO_BYTES_PER_LAUNCH = 2 * 128 * 383760 * 4


def dom_algo_bytes_per_launch():
    """The same under the decoder window (round 6, default): a decoder call covers frames [skip, T - skip) of its chunk,
    skip = (t_pad frames - the decoder's receptive field) & ~3 = 84 of 3198 at 48 k -> 363 600 positions at stage 2."""
    if os.environ.get("RVCX_DEC_WINDOW", "1") == "0":
        return DOM_ALGO_BYTES_PER_LAUNCH
    from polgen_rvc_amd import synthetic as S, weights as W
    skip = (100 - W.synth_dec_rf(S.SYNTH_CFG_48K)) & ~3
    return 2 * 128 * (383760 - 2 * skip * 120) * 4
PEAK_FILE = "mfma_peak_r04.json"   # profiles/: measured split-fp16 ceiling on random operands (tools/mfma_peak.hip)


def load_models(ctx, zero=False, fcpe=False, also_40k=False, crepe=False, outliers=False, dec_outliers=False):
    """zero: this rank receives rank 0's folded weights by broadcast -- it loads shape-only placeholders (zeros, no
    random numbers drawn: the region layout depends on shapes only)."""
    if zero:
        with S.shapes_only():
            return load_models(ctx, False, fcpe, also_40k, crepe, outliers, dec_outliers)

    def z(state):
        return state
    # outliers: the HuBERT with planted massive-activation units (synthetic.py: _plant_outliers) -- what ContentVec-shaped
    # weights do to the fp16-range guard of the split kernels
    ctx.load_hubert(W.hubert_cfg_struct(S.HUBERT_CFG_BASE), z(S.hubert_state(S.HUBERT_CFG_BASE, 0, outliers=outliers)))
    if crepe:
        ctx.load_crepe(z(S.crepe_state("full", 0)))
    elif fcpe:
        sd = S.fcpe_state(S.FCPE_CFG_FULL, 0)
        ctx.load_fcpe(W.fcpe_cfg_struct(W.fcpe_cfg_from_state(sd)), z(sd))
    else:
        ctx.load_rmvpe(W.rmvpe_cfg_struct(S.RMVPE_CFG_FULL), z(S.rmvpe_state(S.RMVPE_CFG_FULL, 0, outliers=dec_outliers)))
    # dec_outliers: planted outlier channels in the NSF decoder (weight-norm g spread 1 : 150, c1 -> c2 hand-offs of a few
    # hundred) and in the F0 U-Net (BatchNorm scale x 150), function-preserving (synthetic.py)
    mid48 = ctx.load_synth(W.synth_cfg_struct(S.SYNTH_CFG_48K, 768), z(S.synth_state(S.SYNTH_CFG_48K, 0, outliers=dec_outliers)))
    if not also_40k:
        return mid48
    mid40 = ctx.load_synth(W.synth_cfg_struct(S.SYNTH_CFG_40K, 768), z(S.synth_state(S.SYNTH_CFG_40K, 1)))
    return mid48, mid40


def ctx_bucket_frames():
    v = os.environ.get("RVCX_BUCKET_FRAMES")
    return int(v) if v else 128


def make_params(seed=0, fcpe=False, crepe=False):
    p = _lib.Params()
    p.f0_method = _lib.F0_CREPE if crepe else (_lib.F0_FCPE if fcpe else _lib.F0_RMVPE)
    p.hop_length = 128
    p.pitch, p.f0_min, p.f0_max = 0.0, 50.0, 1100.0
    p.index_rate, p.protect, p.volume_envelope = 0.0, 0.33, 1.0
    p.sid = 0
    p.x_pad, p.x_query, p.x_center, p.x_max = 1, 6, 38, 41
    p.seed = seed
    return p


def c5_lengths():
    """BASELINE configs[4]'s stand-in for TTS utterances: 256 lengths U(3, 15) s, whole 10 ms frames."""
    g = np.random.Generator(np.random.PCG64(5))
    return [int(round(s * 100)) * 160 for s in g.uniform(3.0, 15.0, C5_UTTERANCES)]


def cpu_baseline():
    """The CPU oracle (oracle/pipeline.py, the pinned restatement of the reference path) timed on the host
    cores on a bounded sample of the same workload.  Reported beside the GPU number; never the product."""
    from oracle import pipeline as OP
    models = OP.Models(S.to_torch(S.hubert_state(S.HUBERT_CFG_BASE, 0)), S.HUBERT_CFG_BASE,
                       S.to_torch(S.rmvpe_state(S.RMVPE_CFG_FULL, 0)), S.RMVPE_CFG_FULL,
                       S.to_torch(S.synth_state(S.SYNTH_CFG_48K, 0)), S.SYNTH_CFG_48K)
    audio = S.make_clip(0, CPU_SAMPLE_SECONDS)
    geo = OP.Geometry(48000, 1, 6, 38, 41)
    t0 = time.perf_counter()
    OP.pipeline(models, geo, audio, 0.0, 0, None, 0.0, 1.0, 0.33, 50, 1100, seed=0)
    dt = time.perf_counter() - t0
    return {"value": CPU_SAMPLE_SECONDS / dt, "unit": "x real-time", "cores": torch.get_num_threads(),
            "kind": "port", "seconds": dt,
            "sample": f"one {CPU_SAMPLE_SECONDS:.0f} s clip of the same workload (oracle/pipeline.py, torch-CPU fp32)"}


def practical_peak():
    """Measured ceiling of the split-fp16 arithmetic: v_mfma_f32_32x32x16_f16 on RANDOM operands sustains less than the
    dense peak (the chip lowers its clock under the matrix pipes' power draw); profiles/mfma_peak_r04.json is the
    output of tools/mfma_peak.hip on this pool (register-fed and LDS-fed loops, 1-3 waves per SIMD).  Returns
    (TFLOP/s of three-MFMA products with register-fed operands, the same with one ds_read_b128 per MFMA, the same for the
    kernels' own k-step -- two accumulators x three dependent MFMAs, six fragment reads, the wh rebuild -- , source)."""
    path = os.path.join(ROOT, "profiles", PEAK_FILE)
    if not os.path.exists(path):
        return None, None, None, None
    res = json.load(open(path))["results"]
    reg = max(r["tflops"] for r in res if r["kernel"].startswith("f16 R"))
    lds = max(r["tflops"] for r in res if r["kernel"].startswith("f16 L ("))
    kstep = max([r["tflops"] for r in res if r["kernel"].startswith("h3 step")] or [lds])
    return reg / 3.0, lds / 3.0, kstep / 3.0, "profiles/" + PEAK_FILE


def pmc_traffic(tile_name):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE are separate profiler runs of this same command; bench.py cannot collect them itself)."""
    import re
    path = next((os.path.join(ROOT, "profiles", f) for f in PMC_FILES
                 if os.path.exists(os.path.join(ROOT, "profiles", f))), None)
    if path is None:
        return None, None
    kernels = json.load(open(path))["kernels"]
    rel = "profiles/" + os.path.basename(path)
    src = f"{rel} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, RVCX_SERIAL=1)"
    mp = re.match(r"resblock_pair<C=(\d+),N1=(\d+)>", tile_name)
    if mp:
        want = (f"resblock_pair_kernel<{mp.group(1)},", f"resblock_pair_persist_kernel<{mp.group(1)},")
        tot_b, tot_n = 0.0, 0
        for k, v in kernels.items():          # the (C, k) instantiations of one channel count share the profile slot
            if want[0] in k or want[1] in k:
                tot_b += v["hbm_bytes_per_launch"] * v["launches"]
                tot_n += v["launches"]
        return (tot_b / tot_n, src) if tot_n else (None, None)
    mh = re.match(r"conv_h3<(\d+),(\d+),(halo(\d+)|linear|stride2)>", tile_name)
    if mh:
        want = f"conv_h3_kernel<{mh.group(1)}, {mh.group(2)},"
        # template args: <BM, BN, WR, WC, KKT, HALO, STRIDE, LIN, XS>; XS = false is the fp32-input instantiation
        tail = {"linear": ", 0, 1, true, false>", "stride2": ", 64, 2, false, false>"}.get(
            mh.group(3), f", {mh.group(4)}, 1, false, false>")
        for k, v in kernels.items():
            if want in k and tail in k:
                return v["hbm_bytes_per_launch"], src
        return None, None
    mg = re.match(r"gemm_h3<(\d+),(\d+)>", tile_name)
    if mg:
        want = f"gemm_h3_kernel<{mg.group(1)}, {mg.group(2)},"
        tot_b, tot_n = 0.0, 0
        for k, v in kernels.items():
            if want in k:
                tot_b += v["hbm_bytes_per_launch"] * v["launches"]
                tot_n += v["launches"]
        return (tot_b / tot_n, src) if tot_n else (None, None)
    m = re.match(r"conv_fast_(sb|db)<(\d+),(\d+),(halo(\d+)|linear|stride2)>", tile_name)
    if not m:
        return None, None
    fn = "conv_fast_sb_kernel" if m.group(1) == "sb" else "conv_fast_kernel"
    want = f"{fn}<{m.group(2)}, {m.group(3)},"
    tail = {"linear": ", 32, 0", "stride2": ", 16, 64, 2>"}.get(m.group(4), f", 16, {m.group(5)}")
    if m.group(1) == "sb" and m.group(4) != "stride2":
        tail += ", 1>"   # trailing STRIDE template argument of conv_fast_sb_kernel
    for k, v in kernels.items():
        if want in k and tail in k:
            return v["hbm_bytes_per_launch"], src
    return None, None


def child_bench(extra_args, env_extra, label, roofline=True):
    """One more bench of this run in a FRESH child process, started before this process touches the GPU and run to
    completion (never an exec of a GPU-initialised process, never two benches sharing the device)."""
    env = dict(os.environ, **env_extra)
    cmd = [sys.executable, os.path.abspath(__file__)] + extra_args + ["--no-cpu-baseline", "--no-children"]
    if not roofline:
        cmd.append("--no-roofline")
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        out = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
               "warmup": d["warmup"], "how": "child process of this bench run (started before the parent touched the GPU)"}
        out.update(label)
        # did the child stay on the fast path?  (split-fp16 range guard / BiGRU cluster fallback counters of its context)
        out["fast_path"] = {k: d["config"].get(k) for k in ("fp32_layers", "fp32_reruns", "gru_fallbacks")}
        if d.get("roofline"):
            out["roofline"] = d["roofline"]
        out["stage_ms"] = d.get("stage_ms")
        return out, d
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}, None


def exact_fp32_child(steps, warmup):
    """The same workload with every product on the exact-fp32 MFMA (RVCX_H3=0 RVCX_ATT_H3=0)."""
    out, _ = child_bench(["--steps", str(steps), "--warmup", str(warmup)], {"RVCX_H3": "0", "RVCX_ATT_H3": "0"},
                         {"dtype": "f32 (v_mfma_f32_32x32x2_f32 products, exact fp32)", "env": "RVCX_H3=0 RVCX_ATT_H3=0",
                          "workload": "c2"})
    return out


def outliers_child(steps, warmup):
    """C2 again with a HuBERT that carries planted outlier units: FFN activations of 400-900 (inside the split kernels'
    range: must stay on the fast path) and attention V heads of ~1200 in two layers (beyond fp16 range for K / V: the
    guard pins exactly those two attention calls to the exact-fp32 kernel during warm-up; `fast_path` shows the counts)."""
    out, _ = child_bench(["--steps", str(steps), "--warmup", str(max(3, warmup)), "--hubert-outliers"], {},
                         {"workload": "c2 with outlier units planted in HuBERT (synthetic.hubert_state(outliers=True): FFN units "
                                      "x200 in layers 2/6/10, V head 3 x400 in layers 4/8)"}, roofline=False)
    return out


def decoder_outliers_child(steps, warmup):
    """C2 again with outlier channels planted in the NSF decoder and the F0 U-Net: the split kernels'
    range guard must leave (almost) every layer where it is -- `fast_path` shows what it pinned during warm-up."""
    out, _ = child_bench(["--steps", str(steps), "--warmup", str(max(3, warmup)), "--decoder-outliers"], {},
                         {"workload": "c2 with outlier channels planted in the NSF decoder (conv_pre / ups.0 and one step of every "
                                      "ResBlock1: x150 / /150) and in the F0 U-Net (BatchNorm scale x150 in eight blocks)"},
                         roofline=False)
    return out


def full_decoder_child(steps, warmup):
    """C2 with the NSF decoder evaluated over every frame of every call (RVCX_DEC_WINDOW=0), as the reference does before it
    throws t_pad_tgt samples away at both ends: what the decoder window (round 6) is worth, in the open."""
    out, _ = child_bench(["--steps", str(steps), "--warmup", str(warmup)], {"RVCX_DEC_WINDOW": "0"},
                         {"env": "RVCX_DEC_WINDOW=0", "workload": "c2, NSF decoder over all 3198 frames of the padded chunk "
                          "(default: frames 84 .. 3114, the kept 30 s + the decoder's receptive field)"}, roofline=False)
    return out


def long_clip_child():
    """The reference's real workload in small: one 95 s clip per step -- F0 once over 9 700 frames, three silence-aligned
    chunks through HuBERT / TextEncoder / flow / decoder (the branch tests/test_gpu_round6.py pins to the reference)."""
    out, _ = child_bench(["--steps", "5", "--warmup", "2", "--clip-seconds", "95"], {},
                         {"workload": "one 95 s 16 kHz clip per step (3 chunks), RVC v2 48k, rmvpe+, geometry (1,6,38,41)"},
                         roofline=False)
    return out


def c3_child():
    """BASELINE configs[2] at its stated size: 64 x 30 s per step, index_rate 0.75 over 65 536 x 768."""
    out, d = child_bench(["--workload", "c3", "--steps", "3", "--warmup", "1"], {},
                         {"workload": "c3: batch of 64 x 30 s clips per step, index_rate 0.75, 65 536 x 768 index resident"})
    if d is not None:
        out["clips_per_step"] = d["config"]["clips_per_step"]
        out["micro_batch"] = d["config"]["micro_batch"]
        out["ms_per_clip"] = d["ms_per_step"] / d["config"]["clips_per_step"]
        out["stage_ms_note"] = d.get("stage_ms_note")
    return out


def c5_child():
    """BASELINE configs[4] on one GPU: 256 utterances of U(3, 15) s per step, a 40 k and a 48 k voice model resident
    beside one HuBERT and one RMVPE; mixed lengths run as ragged micro-batches (length classes)."""
    out, d = child_bench(["--workload", "c5", "--steps", "3", "--warmup", "1"], {},
                         {"workload": "c5: 256 utterances of U(3,15) s per step, 40 k + 48 k voice models resident, "
                                      "ragged micro-batches by length class"})
    if d is not None:
        out["clips_per_step"] = d["config"]["clips_per_step"]
        out["micro_batches"] = d["config"].get("micro_batches")
        out["ms_per_clip"] = d["ms_per_step"] / d["config"]["clips_per_step"]
        out["stage_ms_note"] = d.get("stage_ms_note")
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_workers(n, argv):
    """`python bench.py --gpus N` invoked plainly: start N worker processes of this same command (one per GPU) BEFORE
    this process makes any GPU call, wait for all of them, relay rank 0's JSON line.  Returns the exit code."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   RVCX_BENCH_WORKER="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    t_start, t_term = time.monotonic(), None
    while any(p.poll() is None for p in procs):
        late = time.monotonic() - t_start > WORKER_DEADLINE_S
        if late or any(p.poll() not in (None, 0) for p in procs):   # a rank died: the others would wait in a collective forever
            if t_term is None:
                time.sleep(2.0)
                t_term = time.monotonic()
                for p in procs:
                    if p.poll() is None:
                        p.terminate()                       # exactly the processes started above, by handle
            elif time.monotonic() - t_term > 20.0:          # a rank stuck in a GPU wait ignores SIGTERM
                for p in procs:
                    if p.poll() is None:
                        p.kill()
        time.sleep(0.05)
    reader.join(10)
    codes = [p.returncode for p in procs]
    out0 = buf[0] if buf else ""
    lines = [l for l in (out0 or "").splitlines() if l.startswith("{")]
    if lines:
        print(lines[-1])
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad or not lines:
        print(f"bench.py: workers failed (rank, exit code): {bad}" if bad else "bench.py: rank 0 printed no result",
              file=sys.stderr)
        return next((c for _, c in bad), 1) or 1
    return 0


def dry_run(a, rank, world):
    """The multi-process plumbing without a GPU (tests/test_dist_gloo.py drives it through the launcher): gloo
    rendezvous, the c5 shard, barrier + max-over-ranks timing, rank 0's JSON line, a failing rank's exit code."""
    D.init("gloo")
    lengths = c5_lengths()
    mine = D.shard(len(lengths), rank, world, lengths)
    if a.dry_run_fail_rank == rank:
        raise SystemExit(3)
    D.barrier()
    t0 = time.perf_counter()
    time.sleep(0.001 * len(mine))
    D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0)
    counts = torch.zeros(world, dtype=torch.int64)
    counts[rank] = len(mine)
    secs = torch.zeros(world, dtype=torch.float64)
    secs[rank] = sum(lengths[i] for i in mine) / 16000.0
    if world > 1:
        torch.distributed.all_reduce(counts)
        torch.distributed.all_reduce(secs)
    digests = D.gather_int64((1 << 63) + 7 * rank) if a.verify_ranks else None      # the --verify-ranks exchange, CPU side
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "rank_digests": digests, "steps": a.steps or 1, "utterances_per_rank": counts.tolist(),
                          "audio_seconds_per_rank": secs.tolist(), "value": float(secs.sum()) / dt, "workload": "c5"}))
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["c2", "c3", "c5"], default="c2")
    ap.add_argument("--batch", type=int, default=None, help="clips per step (c3 default 64, c2 default 1)")
    ap.add_argument("--f0-method", choices=["rmvpe+", "fcpe", "mangio-crepe"], default="rmvpe+",
                    help="F0 back-end of VC.get_f0; BASELINE's metric is quoted on rmvpe+ (the others: secondary lines)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-children", "--no-exact-fp32", dest="no_children", action="store_true",
                    help="skip the exact_fp32 and c3 child benches of the default run")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--profile-out", default="")
    ap.add_argument("--hubert-outliers", action="store_true",
                    help="HuBERT with planted massive-activation units (what real ContentVec-shaped weights do to the range guard)")
    ap.add_argument("--decoder-outliers", action="store_true",
                    help="NSF decoder and F0 U-Net with planted outlier channels (synthetic.synth_state / rmvpe_state(outliers=True))")
    ap.add_argument("--clip-seconds", type=float, default=CLIP_SECONDS, help="clip length of the c2 / c3 workloads (default 30)")
    ap.add_argument("--verify-ranks", action="store_true",
                    help="every rank converts one probe clip after the weight broadcast; rank 0 reports whether all PCM digests agree")
    ap.add_argument("--dry-run", action="store_true", help="CPU-only plumbing run (gloo): launcher, shard, timing")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1)
    a = ap.parse_args()
    c3, c5 = a.workload == "c3", a.workload == "c5"
    crepe = a.f0_method == "mangio-crepe"
    fcpe = a.f0_method == "fcpe" or crepe            # "a secondary line": no children, no roofline object, no CPU baseline
    B = a.batch or (C3_BATCH if c3 else 1)
    if a.steps is None:
        a.steps = 3 if (c3 or c5) else 10

    rank, local, world = D.env_rank()
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    # torch.cuda.device_count() does not initialise the GPU on this image
    if not a.dry_run and torch.cuda.device_count() < max(a.gpus, local + 1):
        raise SystemExit(f"--gpus {a.gpus}: only {torch.cuda.device_count()} GPU(s) visible")
    if a.gpus > 1 and world == 1:
        # plain invocation: become the launcher -- no GPU call has been made by this process
        raise SystemExit(launch_workers(a.gpus, sys.argv[1:]))
    if a.dry_run:
        return dry_run(a, rank, world)
    fp32 = c3_obj = c5_obj = out_obj = dec_out_obj = long_obj = full_dec_obj = None
    clip_seconds = float(a.clip_seconds)
    if (world == 1 and not a.no_children and a.workload == "c2" and B == 1 and not fcpe and not a.hubert_outliers
            and not a.decoder_outliers and clip_seconds == CLIP_SECONDS):
        fp32 = exact_fp32_child(a.steps, a.warmup)       # all before the first GPU call of this process
        out_obj = outliers_child(a.steps, a.warmup)
        dec_out_obj = decoder_outliers_child(a.steps, a.warmup)
        long_obj = long_clip_child()
        full_dec_obj = full_decoder_child(a.steps, a.warmup)
        c3_obj = c3_child()
        c5_obj = c5_child()
        time.sleep(5.0)                                  # the children left the chip warm: let it idle before the headline loop
    rank, local, world = D.init("nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ctx = _lib.Context(local)

    # rank 0 parses/folds/packs the checkpoints; the folded weight regions go to the other GPUs over RCCL/xGMI
    t0 = time.perf_counter()
    mids = load_models(ctx, zero=(rank != 0), fcpe=fcpe, also_40k=c5, crepe=crepe, outliers=a.hubert_outliers,
                       dec_outliers=a.decoder_outliers)
    mid = mids[0] if c5 else mids
    if c3:
        big = S.make_index(C3_INDEX_ROWS, 768, 0)
        ctx.load_index(np.zeros_like(big) if rank != 0 else big)
    t_load = time.perf_counter() - t0
    t0 = time.perf_counter()
    nbytes = D.broadcast_weights(ctx, local, 0)
    t_bcast = time.perf_counter() - t0

    params = make_params(fcpe=fcpe, crepe=crepe)
    if c3:
        params.index_rate = 0.75
    ranks_agree = rank_digests = None
    if a.verify_ranks:
        # ranks != 0 loaded shape-only placeholders: the same PCM on every rank means rank 0's folded weights (and the
        # index) arrived through the RCCL broadcast and were adopted.  Same clip, same Philox seed, outside the timed region.
        import hashlib
        probe = ctx.convert_batch(mid, [S.make_clip(9999, 2.0)], params)[0][0]
        digest = int.from_bytes(hashlib.sha256(probe.tobytes()).digest()[:8], "little")
        rank_digests = D.gather_int64(digest, dev)
        ranks_agree = len(set(rank_digests)) == 1 and bool(np.any(probe != 0))
    # pinned host buffers: the step's H2D / D2H copies are asynchronous DMA inside the timed region
    if c5:
        lengths = c5_lengths()
        mine = D.shard(len(lengths), rank, world, lengths)
        clips = [S.make_clip(5000 + i, lengths[i] / 16000.0) for i in mine]
        model_of = [mids[1] if i % 2 == 0 else mids[0] for i in mine]       # even -> 40 k, odd -> 48 k
        audio_seconds_per_step = sum(lengths) / 16000.0                     # the whole job's, all ranks
    else:
        clips = [S.make_clip(rank * B + i, clip_seconds) for i in range(B)]
        model_of = [mid] * B
        audio_seconds_per_step = world * B * clip_seconds
    wavs = [torch.from_numpy(c).pin_memory() for c in clips]
    outs = [torch.empty(ctx.out_capacity(m, c.shape[0], params), dtype=torch.int16).pin_memory()
            for c, m in zip(clips, model_of)]
    calls = []                                            # one convert_batch call per resident voice model
    for m in sorted(set(model_of)):
        sel = [i for i, mm in enumerate(model_of) if mm == m]
        calls.append((m, [wavs[i].data_ptr() for i in sel], [clips[i].shape[0] for i in sel],
                      [outs[i].data_ptr() for i in sel]))
    n = clips[0].shape[0]
    torch.cuda.synchronize()

    def step():
        got = []
        for m, wp, ns, op in calls:
            got += ctx.convert_batch_raw(m, wp, ns, params, op)
        return got

    for _ in range(a.warmup):
        step()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        got = step()
    torch.cuda.synchronize()
    D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0, dev)
    ms_per_step = dt / a.steps * 1e3
    rtf = a.steps * audio_seconds_per_step / dt
    stage = ctx.last_timing()
    # The headline's timed region (steps x ~26 ms) ends inside the few hundred milliseconds the boxes of this pool hold their
    # boost clock (DESIGN.md "What sustained load does").  For the record, NOT the headline: the same step 80 more times
    # (~2 s of uninterrupted load), reported as `sustained`.
    sustained = None
    if world == 1 and not (c3 or c5) and not fcpe and B == 1 and not a.no_roofline and clip_seconds == CLIP_SECONDS:
        n_sus = 80
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_sus):
            step()
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t1
        sustained = {"steps": n_sus, "ms_per_step": dt_s / n_sus * 1e3, "value": n_sus * audio_seconds_per_step / dt_s,
                     "note": "the same step, 80 more times right behind the timed loop (~2 s of continuous load); not the headline"}
    # retrieval: queries whose neighbours the split-fp16 pre-filter could not certify and that were searched exhaustively
    # (the slow path: one workgroup reads the whole matrix per query) during warm-up + timed steps; 0 on this workload
    idx_exhaustive = ctx.index_exhaustive() if c3 else None
    mbs_per_step = None
    if c5 or len(clips) > 1:
        # micro-batches the LAST call of a step formed (c5: one call per voice model)
        mbs_per_step = ctx.last_micro_batches()

    # ---- roofline of the dominant kernel family (MFMA implicit-GEMM conv): one extra, untimed step in SERIAL mode
    # (every launch on the library's one stream, so a launch's duration is its own) with a HIP event pair around
    # every conv launch.  `rocprofv3 --kernel-trace --stats` of `RVCX_SERIAL=1 python bench.py ...` gives the same
    # per-kernel averages (profiles/rocprof_r03_*).
    roofline, prof = None, None
    if not a.no_roofline:
        ctx.flop_counter(reset=True)
        ctx.conv_profile_begin()
        step()
        prof = ctx.conv_profile_end()
        if a.profile_out and rank == 0:
            with open(a.profile_out.replace('.json', '') + '_conv_launches.csv', 'w') as f:
                f.write(ctx.conv_profile_csv())
        total_flops = ctx.flop_counter()
        # "dominant" = the profile slot that does most of the path's arithmetic.  Until round 4 that was also the slot with the
        # most time; since every 1-D conv of the path runs on ONE tile (conv_h3<64,64,halo64>: ~110 launches of a dozen
        # different shapes, 20 us ... 400 us each) that mixed slot has slightly more time than the fused ResBlock step of
        # C = 128 (9 launches, 35 % of all FLOPs).  The roofline stays on the single-purpose kernel; the mixed slot is
        # reported beside it (`most_time`).
        prof.sort(key=lambda r: -r["ms"])
        by_time = prof[0]
        dom = max(prof, key=lambda r: r["flops"])
        conv_ms = sum(r["ms"] for r in prof)
        conv_flops = sum(r["flops"] for r in prof)
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic(dom["tile"])
        h3 = dom["tile"].startswith(("conv_h3", "resblock_pair", "gemm_h3"))
        peak = PEAK_H3_TFLOPS if h3 else PEAK_F32_TFLOPS
        pp_reg, pp_lds, pp_kstep, pp_src = practical_peak() if h3 else (None, None, None, None)
        roofline = {"bound": "mfma", "kernel": dom["tile"], "achieved": achieved,
                    "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                    "practical_peak": pp_reg, "frac_practical": (achieved / pp_reg) if pp_reg else None,
                    "practical_peak_lds_fed": pp_lds, "practical_peak_kstep": pp_kstep,
                    "frac_kstep": (achieved / pp_kstep) if pp_kstep else None,
                    "practical_peak_note": (None if not pp_reg else
                                            f"{pp_src}: v_mfma_f32_32x32x16_f16 on random fp16 operands sustains "
                                            f"{3 * pp_reg:.0f} TFLOP/s register-fed ({3 * pp_lds:.0f} with one ds_read_b128 per "
                                            "MFMA) at ~1.6 GHz -- the chip lowers its clock under the matrix pipes' power "
                                            "draw; a third of that is the ceiling of three-MFMA products on real data; "
                                            f"the kernels' own k-step pattern (six fragment reads + six MFMAs, three of "
                                            f"them dependent per accumulator) sustains {3 * pp_kstep:.0f} at ~1.43 GHz "
                                            "(practical_peak_kstep)"),
                    "peak_note": ("dense fp16 MFMA peak 2500 TFLOP/s / 3 (each fp32 product block = 3 fp16 MFMAs of a "
                                  "hi/lo split); 'achieved' counts algorithmic 2*M*N*K conv FLOPs" if h3 else
                                  "fp32 MFMA peak (v_mfma_f32_32x32x2_f32)"),
                    "traffic_source": traffic_src,
                    "launches": dom["launches"], "avg_launch_ms": dom["ms"] / dom["launches"],
                    "flops_per_launch": dom["flops"] / dom["launches"],
                    "dominant_by": "algorithmic FLOPs",
                    "definition": ("rounds 1-3: the profile slot with the most TIME; since round 4: the slot with the most "
                                   "algorithmic FLOPs (a single-purpose kernel).  The time-dominant slot is reported as a "
                                   "peer object, `roofline_by_time`, so that trends can follow either definition"),
                    "most_time": (None if by_time is dom else
                                  {"kernel": by_time["tile"], "launches": by_time["launches"], "ms": by_time["ms"],
                                   "achieved": by_time["flops"] / (by_time["ms"] * 1e-3) / 1e12,
                                   "frac": by_time["flops"] / (by_time["ms"] * 1e-3) / 1e12 / peak,
                                   "note": "launches of different shapes share this tile; ms = their sum in the serial step"}),
                    "family": {"ms": conv_ms, "tflops": conv_flops / (conv_ms * 1e-3) / 1e12,
                               "frac_of_fp32_mfma_peak": conv_flops / (conv_ms * 1e-3) / 1e12 / PEAK_F32_TFLOPS,
                               "share_of_step": conv_ms / ms_per_step},
                    "whole_path_tflops": total_flops / (ms_per_step * 1e-3) / 1e12}

    if rank == 0:
        if c5:
            wl = (f"{C5_UTTERANCES} utterances of U(3,15) s per step for the whole job ({len(clips)} on this rank, sharded by "
                  "length), a 40 k and a 48 k RVC v2 voice model resident beside one HuBERT-base and one RMVPE, "
                  "f0_method=rmvpe+, index_rate=0, geometry (1,6,38,41); mixed lengths run as ragged micro-batches "
                  f"(length classes of {ctx_bucket_frames()} frames)")
        elif c3:
            wl = (f"batch of {B} x 30 s 16 kHz clips per GPU per step, RVC v2 48k, f0_method=rmvpe+, HuBERT-base, "
                  f"index_rate=0.75 over a resident {C3_INDEX_ROWS} x 768 index, geometry (1,6,38,41)")
            if world > 1:
                wl = (f"C4 (BASELINE configs[3]): {world * B} x 30 s clips sharded {world} ways ({B} per rank), weights "
                      "broadcast from rank 0 over RCCL/xGMI, no collective in the hot loop; per rank: ") + wl
        else:
            cs = f"{clip_seconds:g}"
            wl = ((f"single {cs} s 16 kHz clip per GPU per step" if B == 1 else f"{B} x {cs} s 16 kHz clips per GPU per step") +
                  ", RVC v2 48k, f0_method=rmvpe+, HuBERT-base, index_rate=0, geometry (1,6,38,41)")
        if fcpe:
            wl = wl.replace("f0_method=rmvpe+", f"f0_method={a.f0_method} (secondary line: BASELINE's metric is quoted on rmvpe+)")
        multi = len(clips) > 1
        res = {"metric": "real-time-factor (audio-sec/wall-sec) per GPU, 30s@16kHz RMVPE->48kHz",
               "value": rtf, "unit": "x real-time", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if c5 else "weak",
               "vs_baseline": None,
               "dtype": ("f32 (conv products as 3 fp16 MFMAs of a hi/lo split, fp32 accumulate; fp32 elsewhere)"
                         if os.environ.get("RVCX_H3", "1") != "0" else "f32 (exact fp32 MFMA products)"),
               "data": "synthetic",
               "value_per_gpu": rtf / world,
               "value_is": "whole-job aggregate over n_gpus (driver contract); value_per_gpu = value / n_gpus",
               "config": {"workload": wl + "; timed region = H2D of float PCM (pinned host) + all kernels + D2H of int16"
                          + ("; NSF decoder over every frame of every call (RVCX_DEC_WINDOW=0)"
                             if os.environ.get("RVCX_DEC_WINDOW", "1") == "0" else
                             "; the NSF decoder evaluates the samples VC.pipeline keeps + its receptive field, not the t_pad ends "
                             "it throws away (same output; `full_decoder_value` = without)"),
                          "clips_per_step": len(clips), "micro_batch": ctx.micro_batch(mid, n, params),
                          "micro_batches": mbs_per_step, "index_exhaustive_queries": idx_exhaustive,
                          "out_samples": got[0] if len(got) == 1 else sum(got), "weights_bcast_bytes": nbytes,
                          "weights_bcast_s": t_bcast, "load_s": t_load,
                          "ranks_agree": ranks_agree, "rank_pcm_digests": rank_digests,
                          # fast-path evidence: layers the fp16-range guard pinned to the exact-fp32 kernels, calls it
                          # repeated, calls that fell back to the single-workgroup BiGRU (all 0 = every launch of the
                          # timed region ran on the kernels the roofline describes)
                          "fp32_layers": ctx.fp32_layers(), "fp32_reruns": ctx.fp32_reruns(),
                          "gru_fallbacks": ctx.gru_fallbacks()},
               "stage_ms": stage, "sustained": sustained,
               "stage_ms_note": ("sums over the call's micro-batches of each stage's own span on its own stream; the "
                                 "streams overlap, so the stages do not add up to `total`" if multi else
                                 "single clip: rmvpe and hubert run side by side, the rest in sequence"),
               "roofline": roofline,
               "roofline_by_time": (None if roofline is None else
                                    ({"same_as": "roofline"} if roofline["most_time"] is None else
                                     dict(roofline["most_time"], bound="mfma", peak=roofline["peak"], unit="TFLOP/s",
                                          dominant_by="time in the serial profile step"))),
               "conv_tiles": prof}
        # ---- driver-visible peers: the driver keeps `config` whole and reduces the rich child objects
        # below to their names, so the scalar every reader needs from each of them is repeated here
        cfgd = res["config"]

        def _val(o, *path):
            for k in path:
                o = o.get(k) if isinstance(o, dict) else None
            return o
        cfgd["exact_fp32_value"] = _val(fp32, "value")
        cfgd["exact_fp32_roofline_frac"] = _val(fp32, "roofline", "frac")
        cfgd["sustained_value"] = _val(sustained, "value")
        cfgd["outliers_value"] = _val(out_obj, "value")
        cfgd["decoder_outliers_value"] = _val(dec_out_obj, "value")
        cfgd["decoder_outliers_fp32_layers"] = _val(dec_out_obj, "fast_path", "fp32_layers")
        cfgd["c3_value"] = _val(c3_obj, "value")
        cfgd["c3_roofline_frac"] = _val(c3_obj, "roofline", "frac")
        cfgd["c5_value"] = _val(c5_obj, "value")
        cfgd["long_clip_value"] = _val(long_obj, "value")
        cfgd["full_decoder_value"] = _val(full_dec_obj, "value")
        # `value` is PCIe-inclusive (host PCM in, host int16 out: SURVEY 8d).  The same step from its first kernel to its last
        # on the device (HIP events of the last timed step: input already in HBM, output left there), for a reader who wants
        # the device-resident rate beside it -- never the headline
        if stage and stage.get("total") and world == 1 and not (c3 or c5) and B == 1:
            cfgd["device_resident_value"] = clip_seconds / (stage["total"] * 1e-3)
            cfgd["device_resident_note"] = "audio seconds / the device-side span of one step (first kernel -> last kernel, H2D / D2H outside)"
        cfgd["roofline_traffic_ratio"] = (None if roofline is None or not roofline.get("traffic") else
                                          roofline["traffic"] / dom_algo_bytes_per_launch())
        cfgd["roofline_traffic_ratio_note"] = ("PMC bytes per launch of the dominant kernel / its algorithmic bytes "
                                               "(read x + write y of a C=128 stage-2 step over the decoder window: 2 x 186.2 MB; 2 x 196.5 MB "
                                               "with RVCX_DEC_WINDOW=0)")
        if fp32 is not None:
            res["exact_fp32"] = fp32
        if dec_out_obj is not None:
            res["decoder_outliers"] = dec_out_obj
        if long_obj is not None:
            res["long_clip"] = long_obj
        if full_dec_obj is not None:
            res["full_decoder"] = full_dec_obj
        if out_obj is not None:
            res["outliers"] = out_obj
        if c3_obj is not None:
            res["c3"] = c3_obj
        if c5_obj is not None:
            res["c5"] = c5_obj
        if fp32 is not None or c3_obj is not None:
            res["order"] = ("children (exact_fp32, outliers, decoder_outliers, long_clip, full_decoder, c3, c5) ran first, each to completion; 5 s idle; "
                            "then this process's warm-up and timed loop")
        if not a.no_cpu_baseline and world == 1 and not fcpe:
            res["cpu_baseline"] = cpu_baseline()
        else:
            res["cpu_baseline"] = None
        if a.profile_out:
            with open(a.profile_out, "w") as f:
                json.dump(res, f, indent=1)
        print(json.dumps(res))
    ctx.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
