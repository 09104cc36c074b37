"""CPU: the host-side FLAC codec of librvcx.so (csrc/flac.hip) against an independent reader / writer of
the format (tests/flac_codec.py).  Lossless: every round trip is bit-exact."""
import os

import numpy as np
import pytest

import polgen_rvc_amd  # noqa: F401
from polgen_rvc_amd import _lib

import flac_codec as FC


def _speechlike(n, seed, ch=1):
    g = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(n) / 48000.0
    cols = []
    for c in range(ch):
        x = 0.4 * np.sin(2 * np.pi * (180 + 7 * c) * t + 0.3 * np.sin(2 * np.pi * 3 * t)) * (0.5 + 0.5 * np.sin(2 * np.pi * 1.3 * t))
        x += 0.1 * np.sin(2 * np.pi * 1900 * t) + 0.003 * g.standard_normal(n)
        cols.append(np.clip(np.rint(x * 32767), -32768, 32767).astype(np.int16))
    return cols[0] if ch == 1 else np.stack(cols, axis=1)


@pytest.mark.parametrize("n,ch", [(0, 1), (1, 1), (3, 2), (15, 1), (4095, 1), (4096, 2), (4097, 1), (10000, 2), (20001, 1)])
def test_encoder_output_is_decoded_bit_exactly_by_the_independent_reader(n, ch):
    pcm = _speechlike(n, 7 + n, ch)
    blob = _lib.flac_encode(pcm, 48000)
    got, sr, bps = FC.read(blob)                  # checks CRC-8, CRC-16, MD5, STREAMINFO frame sizes on the way
    assert (sr, bps) == (48000, 16) and got.shape == (n, ch)
    assert np.array_equal(got.reshape(pcm.shape), pcm)
    back, sr2, bps2 = _lib.flac_decode(blob)      # and by the product's own decoder
    assert (sr2, bps2) == (48000, 16) and np.array_equal(back.reshape(pcm.shape), pcm)
    if n >= 4096:
        assert len(blob) < 0.75 * pcm.nbytes      # a smooth signal really is compressed (FIXED + Rice chosen)


def test_encoder_edge_blocks_constant_verbatim_and_full_scale():
    g = np.random.Generator(np.random.PCG64(3))
    silence = np.zeros(9000, np.int16)                                    # CONSTANT subframes
    noise = g.integers(-32768, 32768, 9000, dtype=np.int64).astype(np.int16)   # incompressible: VERBATIM fallback
    square = np.where(np.arange(9000) % 50 < 25, 32767, -32768).astype(np.int16)  # largest order-4 residuals (2^19)
    dc = np.full(5000, -12345, np.int16)
    for name, pcm in [("silence", silence), ("noise", noise), ("square", square), ("dc", dc)]:
        blob = _lib.flac_encode(pcm, 40000)
        got, sr, _ = FC.read(blob)
        assert sr == 40000 and np.array_equal(got[:, 0], pcm), name
        assert np.array_equal(_lib.flac_decode(blob)[0], pcm), name
    assert len(_lib.flac_encode(silence, 40000)) < 200
    assert len(_lib.flac_encode(noise, 40000)) <= _lib.lib().rvcx_flac_encode_bound(9000, 1)
    assert len(_lib.flac_encode(noise, 40000)) < noise.nbytes + 100       # verbatim costs a few bytes per frame, no more


@pytest.mark.parametrize("kw", [
    dict(kinds=("lpc",), stereo_modes=("mid",)),
    dict(kinds=("lpc", "fixed2", "verbatim"), stereo_modes=("left", "right", "mid", "indep")),
    dict(kinds=("fixed4", "fixed1", "fixed0", "fixed3"), stereo_modes=("indep",), method=1),
    dict(kinds=("lpc",), stereo_modes=("mid",), escape_first=True, block=576),
    dict(kinds=("fixed2",), stereo_modes=("left",), variable_sizes=True, block=1000),
])
def test_decoder_reads_what_the_encoder_never_writes(kw):
    """LPC subframes, the three stereo decorrelations, Rice2, escape partitions, 8 / 16-bit block-size fields, a PADDING
    block: written by the test-side writer, decoded by the product (and by the test-side reader, as a check of the writer)."""
    pcm = _speechlike(5000, 11, 2).astype(np.int64)
    blob = FC.write(pcm, 44100, **kw)
    ref, sr, bps = FC.read(blob)
    assert np.array_equal(ref, pcm) and (sr, bps) == (44100, 16)
    got, sr, bps = _lib.flac_decode(blob)
    assert (sr, bps) == (44100, 16) and np.array_equal(got, pcm)


def test_decoder_24_bit_and_wasted_bits():
    g = np.random.Generator(np.random.PCG64(5))
    t = np.arange(4000)
    x = (0.5 * np.sin(t * 0.01) * (1 << 23)).astype(np.int64) + g.integers(-2000, 2000, 4000)
    x = (x >> 4) << 4                                      # 4 wasted bits
    blob = FC.write(x.reshape(-1, 1), 96000, bps=24, kinds=("lpc", "fixed3"), wasted=4, method=1)
    got, sr, bps = _lib.flac_decode(blob)
    assert (sr, bps) == (96000, 24) and np.array_equal(got, x)


def test_decoder_rejects_damage():
    pcm = _speechlike(6000, 2)
    blob = bytearray(_lib.flac_encode(pcm, 48000))
    with pytest.raises(_lib.RvcxError, match="fLaC"):
        _lib.flac_decode(b"RIFF" + bytes(blob[4:]))
    bad = bytearray(blob)
    bad[len(bad) // 2] ^= 0x10                             # a flipped bit inside a frame: CRC-16 (or the Rice stream) notices
    with pytest.raises(_lib.RvcxError):
        _lib.flac_decode(bytes(bad))
    bad = bytearray(blob)
    bad[4 + 4 + 18] ^= 0xFF                                # the MD5 in STREAMINFO
    with pytest.raises(_lib.RvcxError, match="MD5"):
        _lib.flac_decode(bytes(bad))
    with pytest.raises(_lib.RvcxError):
        _lib.flac_decode(bytes(blob[:len(blob) - 100]))    # truncated


def test_mirror_writes_flac_only_for_the_flac_extension_and_reads_it_back(tmp_path):
    """rvc/infer/infer.py:153 writes WAV bytes whatever the extension; the mirror keeps that for every extension but
    ".flac" (SURVEY 8 f3).  read_audio decodes FLAC without soundfile: float64 in [-1, 1) like soundfile returns it."""
    from polgen_rvc_amd.infer import audio as A
    pcm = _speechlike(12345, 4)
    for ext, magic in [(".flac", b"fLaC"), (".FLAC", b"fLaC"), (".mp3", b"RIFF"), (".wav", b"RIFF"), (".m4a", b"RIFF")]:
        path = str(tmp_path / ("out" + ext))
        A.write_output(path, pcm, 48000)
        assert open(path, "rb").read(4) == magic, ext
        a, sr = A.read_audio(path)
        assert sr == 48000 and a.dtype == np.float64 and np.array_equal(np.rint(a * 32768.0).astype(np.int16), pcm), ext
    st = _speechlike(5000, 9, 2)
    p2 = str(tmp_path / "st.flac")
    open(p2, "wb").write(_lib.flac_encode(st, 44100))
    a, sr = A.read_audio(p2)
    assert sr == 44100 and a.shape == (5000, 2) and np.array_equal(np.rint(a * 32768.0).astype(np.int16), st)


def _patch_streaminfo(blob, total=None, sample_rate=None):
    """rewrite fields of the STREAMINFO block (bytes 8 .. 41 of a stream whose first metadata block is STREAMINFO)"""
    b = bytearray(blob)
    packed = int.from_bytes(b[18:26], "big")
    if total is not None:
        packed = (packed & ~0xFFFFFFFFF) | (total & 0xFFFFFFFFF)
    if sample_rate is not None:
        packed = (packed & ~(0xFFFFF << 44)) | ((sample_rate & 0xFFFFF) << 44)
    b[18:26] = packed.to_bytes(8, "big")
    return bytes(b)


def test_streams_of_unknown_length_and_crafted_headers():
    """a STREAMINFO total of 0 (streamed encodes) is legal -- the frames are counted, a stream of silence (a
    CONSTANT subframe holds 4096 samples in a dozen bytes) decodes; totals that are implausible for the byte count and a
    sample rate of 0 are refused before anything is allocated."""
    silence = np.zeros(50000, np.int16)
    speech = _speechlike(12345, 5, 2)
    for pcm in (silence, speech):
        blob = _patch_streaminfo(_lib.flac_encode(pcm, 48000), total=0)
        got, sr, bps = _lib.flac_decode(blob)
        assert sr == 48000 and bps == 16 and np.array_equal(got.reshape(pcm.shape), pcm)
    blob = _lib.flac_encode(speech, 48000)
    with pytest.raises(_lib.RvcxError, match="implausible"):
        _lib.flac_decode(_patch_streaminfo(blob, total=(1 << 36) - 1))
    with pytest.raises(_lib.RvcxError, match="sample rate 0"):
        _lib.flac_decode(_patch_streaminfo(blob, sample_rate=0))
    with pytest.raises(_lib.RvcxError):                       # a total that is plausible but wrong still fails, after decoding
        _lib.flac_decode(_patch_streaminfo(blob, total=12346))


# ---- streams the product did not write: the three example files of RFC 9639 (FLAC), Appendix D -------------------------------
# Written by the reference encoder (example 2 carries its vendor string, "reference libFLAC 1.3.3 20190804") and published byte
# for byte with their decoded samples.  Each carries the encoder's own MD5 of the PCM in STREAMINFO and a CRC-16 per frame, so
# the fixtures authenticate themselves (the test recomputes the MD5 from what the product's decoder returns).  Between them:
# VERBATIM subframes with wasted bits (1), left/side decorrelation, FIXED predictors, Rice partitions with an escape code,
# SEEKTABLE / VORBIS_COMMENT / PADDING blocks and a short last frame (2), an LPC subframe of order 2 at 8 bits per sample (3).
RFC9639 = {
    "1": (44100, 16, [[25588, 10416]]),
    "2": (44100, 16, [[10372, 6070], [18041, 10545], [14942, 8743], [17876, 10449], [15627, 9143], [17899, 10463],
                      [16242, 9502], [18077, 10569], [16824, 9840], [18263, 10680], [17295, 10113], [-14418, -8428],
                      [-15201, -8895], [-14508, -8476], [-15195, -8896], [-14818, -8653], [-15486, -9072], [-15349, -8958],
                      [-16054, -9410]]),
    "3": (32000, 8, [[v] for v in (0, 79, 111, 78, 8, -61, -90, -68, -13, 42, 67, 53, 13, -27, -46, -38, -12, 14, 24, 19, 6,
                                   -4, -5, 0)]),
}


@pytest.mark.parametrize("ex", ["1", "2", "3"])
def test_decoder_reads_the_reference_encoders_streams_of_rfc9639(ex):
    import hashlib
    sr, bps, want = RFC9639[ex]
    blob = open(os.path.join(os.path.dirname(__file__), "golden", f"flac_rfc9639_example{ex}.flac"), "rb").read()
    got, sr2, bps2 = _lib.flac_decode(blob)
    want = np.asarray(want)
    assert (sr2, bps2) == (sr, bps) and got.size == want.size      # (mono comes back as a vector)
    got = got.reshape(want.shape)
    assert np.array_equal(got, want)
    # STREAMINFO's MD5 (bytes 26..41 of the file: 4 'fLaC' + 4 block header + 18) is libFLAC's signature over the interleaved
    # little-endian PCM at the stream's sample width
    pcm = got.astype({8: np.int8, 16: "<i2"}[bps]).tobytes()
    assert hashlib.md5(pcm).digest() == blob[26:42]
    ref, sr3, bps3 = FC.read(blob)                # the independent reader agrees (and checks every CRC on the way)
    assert (sr3, bps3) == (sr, bps) and np.array_equal(ref.reshape(want.shape), want)
