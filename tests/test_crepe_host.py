"""CPU checks of the `mangio-crepe` restatement (oracle/crepe.py) and of the mirror's host side.  torchcrepe and librosa
are not installed: the Viterbi restatement is checked against an independent brute-force path search, the constants
against torchcrepe's published values."""
import itertools

import numpy as np

from oracle import crepe as OC


def test_bin_range_of_the_reference_defaults():
    # convert.frequency_to_bins: f0_min 50 Hz -> bin 39 (floor), f0_max 1100 Hz -> bin 308 (ceil): 1200 log2(f/10) - 1997.379 over 20
    assert OC.frequency_to_bins(50.0) == 39 and OC.frequency_to_bins(1100.0, ceil=True) == 308
    assert OC.frequency_to_bins(32.70) == 2 and OC.frequency_to_bins(1975.5, ceil=True) == 358
    t = OC.transition_matrix()
    assert t.shape == (360, 360) and np.allclose(t.sum(1), 1.0)
    assert t[100, 100] == 12 / 144 and t[100, 111] == 1 / 144 and t[100, 112] == 0 and t[0, 0] == 12 / 78


def test_viterbi_restatement_equals_brute_force_path_search():
    """librosa.sequence.viterbi maximises sum log p(t, s_t) + sum log T[s_t-1, s_t] + log p_init: enumerate every path of
    small problems and compare (ties are avoided by the random inputs)."""
    rng = np.random.default_rng(0)
    for n_states, n_steps in ((3, 5), (4, 6), (5, 5)):
        for _ in range(5):
            prob = rng.random((n_states, n_steps)).astype(np.float32)
            trans = rng.random((n_states, n_states))
            trans /= trans.sum(1, keepdims=True)
            got = OC.viterbi_path(prob, trans)
            eps = np.finfo(np.float32).tiny
            lp, lt = np.log(prob.astype(np.float64) + eps), np.log(trans + eps)
            best, arg = -np.inf, None
            for path in itertools.product(range(n_states), repeat=n_steps):
                v = sum(lp[s, t] for t, s in enumerate(path)) + sum(lt[a, b] for a, b in zip(path[:-1], path[1:]))
                if v > best:
                    best, arg = v, path
            assert tuple(got.tolist()) == arg


def test_get_f0_crepe_tail_nan_and_resize_rules():
    """pipeline.py:108-116 on a hand-made pitch track: values < 0.001 become NaN, np.interp spreads a NaN over the
    interval it borders (except exactly at a sample point), nan_to_num turns what is left into 0."""
    src = np.array([100.0, 0.0, 120.0, 130.0, 0.0005, 150.0], np.float32)
    source = src.copy()
    source[source < 0.001] = np.nan
    p_len = 12
    target = np.interp(np.arange(0, len(source) * p_len, len(source)) / p_len, np.arange(0, len(source)), source)
    f0 = np.nan_to_num(target)
    assert f0[0] == 100.0 and f0[1] == 0.0 and f0[4] == 120.0 and f0[5] == 125.0 and f0[6] == 130.0 and f0[7] == 0.0
    assert f0[10] == 150.0 and f0[11] == 150.0            # beyond the last sample point: the last value


def test_mirror_dither_follows_numpys_global_generator():
    from polgen_rvc_amd.infer import pipeline as P
    np.random.seed(5)
    a = P._crepe_dither(1000)
    np.random.seed(5)
    b = P._crepe_dither(1000)
    assert a.dtype == np.float32 and np.array_equal(a, b) and np.abs(a).max() <= 20.0 and abs(float(a.mean())) < 1.5
    assert 7.0 < float(a.std()) < 9.5                     # triangular(-20, 0, 20): sigma = 20 / sqrt(6) = 8.16
