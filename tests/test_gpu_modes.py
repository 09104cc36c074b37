"""The non-default arithmetic modes, end to end against the REFERENCE goldens.

The library chooses its kernels from environment variables read once per process, so each mode runs the golden tests of
tests/test_gpu_pipeline.py in a fresh child process (started as a child, never an exec of this GPU-initialised one):

  RVCX_H3=0 RVCX_ATT_H3=0   every product on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): the mode bench.py reports
                            as `exact_fp32`, and the mode an fp16-range overflow falls back to
  RVCX_FUSE=0               the NSF ResBlock steps as two conv launches instead of the fused kernel
  RVCX_GEMM=0               Linear layers on the channel-first conv tiles instead of the time-major GEMM kernel

Each child runs: the multi-chunk tiny golden, the CI-argument tiny golden, C2 at full size (30 s, 48 k) and the
float-waveform-vs-oracle test -- the same assertions as the default mode (float <= 1e-4 RMS, PCM <= 8 LSB)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
SELECT = "tiny_chunked or tiny_ciargs or c2_30s_48k or float_waveform_vs_oracle"


def _run_mode(env_extra, file="test_gpu_pipeline.py", select=SELECT, passed="4 passed"):
    env = dict(os.environ, **env_extra)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(HERE, file), "-k", select, "-s"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=os.path.dirname(HERE))
    tail = (r.stdout[-3000:] + "\n" + r.stderr[-2000:])
    print(tail)
    assert r.returncode == 0, tail
    assert passed in r.stdout, tail
    return r.stdout


def test_exact_fp32_mode_passes_the_reference_goldens():
    out = _run_mode({"RVCX_H3": "0", "RVCX_ATT_H3": "0"})
    assert "float rms err" in out


def test_unfused_resblock_mode_passes_the_reference_goldens():
    _run_mode({"RVCX_FUSE": "0"})


def test_channel_first_linear_mode_passes_the_reference_goldens():
    _run_mode({"RVCX_GEMM": "0"})


@pytest.mark.parametrize("variant", ["1", "2", "5", "6", "7", "8", "11", "13"])
def test_alternative_fused_resblock_tiles_are_bit_identical_too(variant):
    """RVCX_PAIR_VARIANT selects other forms of the fused ResBlock step (2: small tiles, two workgroups per CU; 5: 8 waves
    of 64 x 64 on 256-position tiles with the input tile overlaid on Y1; 6 / 7: weights straight from L2 into registers,
    barrier-free c2 loop; 8: split-phase staging on LDS counters, no s_barrier in the k-loops -- round-4 experiments; 11 / 13: round 5's persistent form with EARLY requests / the 512-thread N1 = 192 tile at k = 3): the k-order is the same, so each must equal the two conv
    launches bit for bit and torch within fp32 rounding."""
    _run_mode({"RVCX_PAIR_VARIANT": variant}, "test_gpu_conv.py", "fused_resblock", "10 passed")


def test_per_tile_fused_step_of_round_4_is_bit_identical_too():
    """RVCX_PAIR_PERSIST=0: one workgroup per tile with the wide epilogue (round 4's default, now the fallback of the persistent
    form and the form of C = 256)."""
    _run_mode({"RVCX_PAIR_PERSIST": "0"}, "test_gpu_conv.py", "fused_resblock", "10 passed")


@pytest.mark.parametrize("env", [{"RVCX_GRU_FORM": "0"}, {"RVCX_GRU_FORM": "1"}, {"RVCX_GRU_COLOCATE": "0"},
                                 {"RVCX_GRU_FORM": "1", "RVCX_GRU_COLOCATE": "0"}])
def test_bigru_forms_and_the_cross_xcd_hand_off_pass_the_f0_goldens(env):
    """Round 4's BiGRU kernel publishes h_t with a plain store when its cluster sits on one XCD (read from HW_REG_XCC_ID) and
    with the write-through sc1 store otherwise.  RVCX_GRU_COLOCATE=0 spreads every cluster over XCDs, so the second path --
    the one a different dispatch pattern would take -- runs the same goldens (torch GRU, rmvpe_tiny / full_1s / illcond);
    RVCX_GRU_FORM selects the 256-thread form (1) and the round-3 kernel (0)."""
    _run_mode(env, "test_gpu_rmvpe_hubert.py", "gru or rmvpe", "passed")


def test_a_failed_publish_probe_selects_the_write_through_store_and_passes_the_f0_goldens():
    """Round 6: the plain-store publish is probed once per device (gru_publish_probe_kernel).  RVCX_GRU_PROBE_FAIL=1 (debug
    builds of the environment only) makes the probe report failure: the co-located clusters must then run the write-through
    publish -- same goldens, no time-out fallback -- and the probe state must read 0."""
    _run_mode({"RVCX_DEBUG": "1", "RVCX_GRU_PROBE_FAIL": "1", "RVCX_EXPECT_PROBE": "0"}, "test_gpu_rmvpe_hubert.py",
              "gru or rmvpe", "passed")
    _run_mode({"RVCX_DEBUG": "1", "RVCX_GRU_PROBE_FAIL": "1", "RVCX_EXPECT_PROBE": "0"}, "test_gpu_round6.py",
              "publish_probe", "1 passed")


def test_fp32_hand_off_between_layers_passes_the_hubert_and_f0_goldens():
    """RVCX_NO_SPLIT=1: no pre-split fp16 hand-off (HuBERT extractor, U-Net blocks): consumers convert fp32 themselves."""
    _run_mode({"RVCX_NO_SPLIT": "1"}, "test_gpu_rmvpe_hubert.py", "hubert or rmvpe", "passed")


@pytest.mark.parametrize("mode", ["0", "2"])
def test_weight_stationary_conv_tile_off_and_everywhere_pass_the_f0_goldens(mode):
    """RVCX_CONV_WS (csrc/conv_deep.hip, round 6): 0 takes the weight-stationary 64 x 320 tile out (every U-Net level on the
    conv_h3 tiles, round 5's dispatch), 2 puts every eligible layer on it (3 x 3 convs from 64 channels up, the 2 x 2 polyphase
    ConvTranspose2d, the 1-D layers with >= 32 k-steps: measured slower there, which is why the default is the >= 256-channel
    3 x 3 convs only).  Same goldens; in mode 2 also the full C2 conversion (its decoder convs run on the tile then)."""
    _run_mode({"RVCX_CONV_WS": mode}, "test_gpu_rmvpe_hubert.py", "rmvpe", "passed")
    _run_mode({"RVCX_CONV_WS": mode}, "test_gpu_conv.py", "conv2d3x3 or convtranspose2d or convblockres", "passed")
    if mode == "2":
        _run_mode({"RVCX_CONV_WS": mode}, "test_gpu_pipeline.py", "c2_30s_48k or tiny_chunked", "2 passed")


def test_b_direct_gemm_tile_everywhere_passes_the_hubert_goldens():
    """RVCX_GEMM_BD=2 (csrc/gemm.hip, round 6): every Linear with Cout % 128 == 0 on gemm_bd_kernel (activations straight
    from global memory; off by default: measured slower).  HuBERT goldens, the kernel-level GEMM tests, C2 end to end."""
    _run_mode({"RVCX_GEMM_BD": "2"}, "test_gpu_rmvpe_hubert.py", "hubert", "passed")
    _run_mode({"RVCX_GEMM_BD": "2"}, "test_gpu_gemm.py", "gemm", "passed")
    _run_mode({"RVCX_GEMM_BD": "2"}, "test_gpu_pipeline.py", "c2_30s_48k", "1 passed")


@pytest.mark.parametrize("db", ["1", "3"])
def test_decoder_grouping_does_not_change_a_bit(db):
    """RVCX_DEC_BATCH (round 6; default 8): how many utterances of equal length go through one decoder launch sequence.
    1 = round 5's one at a time, 3 = groups that do not divide the micro-batch.  The batch == single tests (8 x 30 s with the
    index, a ragged full-size batch, 64 x 30 s against their single runs) must hold whatever the value."""
    _run_mode({"RVCX_DEC_BATCH": db}, "test_gpu_fullsize_batch.py", "c3_batch_of_8 or ragged_full_size", "2 passed")
    _run_mode({"RVCX_DEC_BATCH": db}, "test_gpu_c3_full.py", "c3", "passed")


@pytest.mark.parametrize("mode", ["0", "1"])
def test_branch_stream_modes_pass_the_reference_goldens(mode):
    """RVCX_RESBLOCK_STREAMS (default 2 since round 6: the ResBlock branches of an NSF stage on the main stream + aux[0]): 1 =
    three streams (rounds 2 - 5), 0 = all on the main stream.  The running mean over the branches is ordered by events, so the
    PCM must not change: the multi-chunk tiny golden, the CI-argument golden, C2 at full size, the float waveform vs the oracle."""
    _run_mode({"RVCX_RESBLOCK_STREAMS": mode})


def test_full_decoder_evaluation_passes_the_reference_goldens_too():
    """RVCX_DEC_WINDOW=0 (round 6): the NSF decoder over every frame of every call, as before the decoder window -- the
    reference goldens (tiny multi-chunk, CI arguments, C2 full size, float waveform) and the full-size batch == single tests."""
    _run_mode({"RVCX_DEC_WINDOW": "0"})
    _run_mode({"RVCX_DEC_WINDOW": "0"}, "test_gpu_fullsize_batch.py", "c3_batch_of_8 or ragged_full_size", "2 passed")


def test_unfused_hubert_first_layer_passes_the_hubert_goldens():
    """RVCX_HUBERT_FUSE0=0 (round 6): the extractor's first layer as conv -> GroupNorm statistics -> normalise + GELU + split
    over a stored fp32 map, instead of the FIR recomputed inside the two GroupNorm passes (ops.hip: hubert_conv0_*)."""
    _run_mode({"RVCX_HUBERT_FUSE0": "0"}, "test_gpu_rmvpe_hubert.py", "hubert", "passed")
    _run_mode({"RVCX_HUBERT_FUSE0": "0"}, "test_gpu_ragged.py", "ragged", "passed")
