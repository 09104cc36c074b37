#!/usr/bin/env python3
"""bench.py -- real-time factor of the rvc/infer hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...)

Workload c2 (default, BASELINE.json configs[1]): one 30 s 16 kHz mono clip per step per GPU, RVC v2 48 kHz voice
model, rmvpe+ F0, contentvec-shaped HuBERT-base, index_rate 0, fp32, chunk geometry (1,6,38,41);
synthetic clip + synthetic weights in the real checkpoint layouts (no real weights exist offline).
Workload c3 (--workload c3, BASELINE.json configs[2]): a batch of 64 x 30 s clips per step, index_rate 0.75 with
a 65 536 x 768 retrieval matrix resident in HBM.
A step = VC.pipeline on the step's clip(s) as SURVEY.md 8(d) defines the metric: H2D of the float PCM (pinned
host memory), every kernel, D2H of the int16 PCM -- all inside the timed region.
Weak scaling: every rank converts its own clip(s) per step; value = all ranks' audio seconds / max-rank wall
(whole-job aggregate, as the driver contract asks; value_per_gpu = value / n_gpus).
"""
import argparse
import json
import os
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
CPU_SAMPLE_SECONDS = 8.0


def load_models(ctx, zero=False, fcpe=False):
    def z(state):
        return {k: np.zeros_like(v) for k, v in state.items()} if zero else state
    ctx.load_hubert(W.hubert_cfg_struct(S.HUBERT_CFG_BASE), z(S.hubert_state(S.HUBERT_CFG_BASE, 0)))
    if fcpe:
        sd = S.fcpe_state(S.FCPE_CFG_FULL, 0)
        ctx.load_fcpe(W.fcpe_cfg_struct(W.fcpe_cfg_from_state(sd)), z(sd))
    else:
        ctx.load_rmvpe(W.rmvpe_cfg_struct(S.RMVPE_CFG_FULL), z(S.rmvpe_state(S.RMVPE_CFG_FULL, 0)))
    return ctx.load_synth(W.synth_cfg_struct(S.SYNTH_CFG_48K, 768), z(S.synth_state(S.SYNTH_CFG_48K, 0)))


def make_params(seed=0, fcpe=False):
    p = _lib.Params()
    p.f0_method = _lib.F0_FCPE if fcpe else _lib.F0_RMVPE
    p.pitch, p.f0_min, p.f0_max = 0.0, 50.0, 1100.0
    p.index_rate, p.protect, p.volume_envelope = 0.0, 0.33, 1.0
    p.sid = 0
    p.x_pad, p.x_query, p.x_center, p.x_max = 1, 6, 38, 41
    p.seed = seed
    return p


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


def pmc_traffic(tile_name):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE are separate profiler runs of this same command; bench.py cannot collect them itself)."""
    import re
    path = os.path.join(ROOT, "profiles", "pmc_traffic_r02.json")
    if not os.path.exists(path):
        return None, None
    kernels = json.load(open(path))["kernels"]
    src = "profiles/pmc_traffic_r02.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, RVCX_SERIAL=1)"
    mp = re.match(r"resblock_pair<C=(\d+),N1=(\d+)>", tile_name)
    if mp:
        want = f"resblock_pair_kernel<{mp.group(1)},"
        tot_b, tot_n = 0.0, 0
        for k, v in kernels.items():          # the (C, k) instantiations of one channel count share the profile slot
            if want in k:
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
                return v["hbm_bytes_per_launch"], "profiles/pmc_traffic_r02.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)"
        return None, None
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
            return v["hbm_bytes_per_launch"], "profiles/pmc_traffic_r02.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)"
    return None, None


def exact_fp32_child(steps, warmup):
    """The same workload with every product on the exact-fp32 MFMA (RVCX_H3=0 RVCX_ATT_H3=0) in a FRESH child
    process, started before this process touches the GPU and run to completion (never an exec of a GPU-initialised
    process, never two benches sharing the device)."""
    import subprocess
    env = dict(os.environ, RVCX_H3="0", RVCX_ATT_H3="0")
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", str(warmup),
           "--no-cpu-baseline", "--no-exact-fp32", "--no-roofline"]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                "warmup": d["warmup"], "dtype": "f32 (v_mfma_f32_32x32x2_f32 products, exact fp32)",
                "env": "RVCX_H3=0 RVCX_ATT_H3=0", "how": "child process of this bench run, same workload"}
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["c2", "c3"], default="c2")
    ap.add_argument("--batch", type=int, default=None, help="clips per step (c3 default 64, c2 default 1)")
    ap.add_argument("--f0-method", choices=["rmvpe+", "fcpe"], default="rmvpe+",
                    help="F0 back-end of VC.get_f0; BASELINE's metric is quoted on rmvpe+ (fcpe: secondary line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-fp32", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--profile-out", default="")
    a = ap.parse_args()
    c3 = a.workload == "c3"
    fcpe = a.f0_method == "fcpe"
    B = a.batch or (C3_BATCH if c3 else 1)
    if a.steps is None:
        a.steps = 3 if c3 else 10

    rank, local, world = D.env_rank()
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit(f"--gpus {a.gpus} needs one process per GPU: python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {a.gpus} --master-addr 127.0.0.1 bench.py --gpus {a.gpus} ...")
    # torch.cuda.device_count() does not initialise the GPU on this image
    if torch.cuda.device_count() < max(a.gpus, local + 1):
        raise SystemExit(f"--gpus {a.gpus}: only {torch.cuda.device_count()} GPU(s) visible")
    fp32 = None
    if world == 1 and not a.no_exact_fp32 and not c3 and not fcpe:
        fp32 = exact_fp32_child(a.steps, a.warmup)       # before the first GPU call of this process
    rank, local, world = D.init("nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ctx = _lib.Context(local)

    # rank 0 parses/folds/packs the checkpoints; the folded weight regions go to the other GPUs over RCCL/xGMI
    t0 = time.perf_counter()
    mid = load_models(ctx, zero=(rank != 0), fcpe=fcpe)
    if c3:
        big = S.make_index(C3_INDEX_ROWS, 768, 0)
        ctx.load_index(np.zeros_like(big) if rank != 0 else big)
    t_load = time.perf_counter() - t0
    t0 = time.perf_counter()
    nbytes = D.broadcast_weights(ctx, local, 0)
    t_bcast = time.perf_counter() - t0

    params = make_params(fcpe=fcpe)
    if c3:
        params.index_rate = 0.75
    # pinned host buffers: the step's H2D / D2H copies are asynchronous DMA inside the timed region
    clips = [S.make_clip(rank * B + i, CLIP_SECONDS) for i in range(B)]
    n = clips[0].shape[0]
    wavs = [torch.from_numpy(c).pin_memory() for c in clips]
    cap = ctx.out_capacity(mid, n, params)
    outs = [torch.empty(cap, dtype=torch.int16).pin_memory() for _ in range(B)]
    wp, op, ns = [w.data_ptr() for w in wavs], [o.data_ptr() for o in outs], [n] * B
    torch.cuda.synchronize()

    def step():
        return ctx.convert_batch_raw(mid, wp, ns, params, op)[0]

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
    rtf = world * a.steps * B * CLIP_SECONDS / dt
    stage = ctx.last_timing()

    # ---- roofline of the dominant kernel family (MFMA implicit-GEMM conv): one extra, untimed step in SERIAL mode
    # (every launch on the library's one stream, so a launch's duration is its own) with a HIP event pair around
    # every conv launch.  `rocprofv3 --kernel-trace --stats` of `RVCX_SERIAL=1 python bench.py ...` gives the same
    # per-kernel averages (profiles/rocprof_r02_*).
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
        prof.sort(key=lambda r: -r["ms"])
        dom = prof[0]
        conv_ms = sum(r["ms"] for r in prof)
        conv_flops = sum(r["flops"] for r in prof)
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic(dom["tile"])
        h3 = dom["tile"].startswith("conv_h3") or dom["tile"].startswith("resblock_pair")
        peak = PEAK_H3_TFLOPS if h3 else PEAK_F32_TFLOPS
        roofline = {"bound": "mfma", "kernel": dom["tile"], "achieved": achieved,
                    "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                    "peak_note": ("dense fp16 MFMA peak 2500 TFLOP/s / 3 (each fp32 product block = 3 fp16 MFMAs of a "
                                  "hi/lo split); 'achieved' counts algorithmic 2*M*N*K conv FLOPs" if h3 else
                                  "fp32 MFMA peak (v_mfma_f32_32x32x2_f32)"),
                    "traffic_source": traffic_src,
                    "launches": dom["launches"], "avg_launch_ms": dom["ms"] / dom["launches"],
                    "flops_per_launch": dom["flops"] / dom["launches"],
                    "family": {"ms": conv_ms, "tflops": conv_flops / (conv_ms * 1e-3) / 1e12,
                               "frac_of_fp32_mfma_peak": conv_flops / (conv_ms * 1e-3) / 1e12 / PEAK_F32_TFLOPS,
                               "share_of_step": conv_ms / ms_per_step},
                    "whole_path_tflops": total_flops / (ms_per_step * 1e-3) / 1e12}

    if rank == 0:
        wl = (f"batch of {B} x 30 s 16 kHz clips per GPU per step, RVC v2 48k, f0_method=rmvpe+, HuBERT-base, "
              f"index_rate=0.75 over a resident {C3_INDEX_ROWS} x 768 index, geometry (1,6,38,41)" if c3 else
              ("single 30 s 16 kHz clip per GPU per step" if B == 1 else f"{B} x 30 s 16 kHz clips per GPU per step") +
              ", RVC v2 48k, f0_method=rmvpe+, HuBERT-base, index_rate=0, geometry (1,6,38,41)")
        if fcpe:
            wl = wl.replace("f0_method=rmvpe+", "f0_method=fcpe (secondary line: BASELINE's metric is quoted on rmvpe+)")
        res = {"metric": "real-time-factor (audio-sec/wall-sec) per GPU, 30s@16kHz RMVPE->48kHz",
               "value": rtf, "unit": "x real-time", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": ("f32 (conv products as 3 fp16 MFMAs of a hi/lo split, fp32 accumulate; fp32 elsewhere)"
                         if os.environ.get("RVCX_H3", "1") != "0" else "f32 (exact fp32 MFMA products)"),
               "data": "synthetic",
               "value_per_gpu": rtf / world,
               "value_is": "whole-job aggregate over n_gpus (driver contract); value_per_gpu = value / n_gpus",
               "config": {"workload": wl + "; timed region = H2D of float PCM (pinned host) + all kernels + D2H of int16",
                          "clips_per_step": B, "micro_batch": ctx.micro_batch(mid, n, params),
                          "out_samples": got, "weights_bcast_bytes": nbytes, "weights_bcast_s": t_bcast,
                          "load_s": t_load},
               "stage_ms": stage, "roofline": roofline, "conv_tiles": prof}
        if fp32 is not None:
            res["exact_fp32"] = fp32
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
