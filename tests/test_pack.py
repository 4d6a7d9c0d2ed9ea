"""Posterior pack round trip (src/io.jl:14-34 / test/io.jl:1-23 analogue)."""
import numpy as np
import pytest

import cases
import causalgpslc_jl_amd as gp


@pytest.mark.parametrize("shape", ["UX", "U", "X", "T"])
def test_round_trip(tmp_path, shape):
    c = cases.make_case(17, shape, False, S=4, seed=3)
    g = cases.gpslc_object(gp, c, hyperparams=gp.HyperParameters(nU=2, nOuter=30, nBurnIn=27, predictionCovarianceNoise=1e-9))
    p = str(tmp_path / "g.gpslcpk")
    gp.saveGPSLCObject(g, p)
    h = gp.loadGPSLCObject(p)
    for name in ("X", "T", "Y", "U", "uyLS", "xyLS", "tyLS", "yNoise", "yScale"):
        a, b = getattr(g, name), getattr(h, name)
        assert (a is None and b is None) or np.array_equal(a, b), name
    assert h.hyperparams == g.hyperparams
    assert gp.getNumPosteriorSamples(h) == 4 and gp.getN(h) == 17


def test_rejects_garbage(tmp_path):
    p = tmp_path / "x.bin"
    p.write_bytes(b"not a pack at all")
    with pytest.raises(ValueError):
        gp.loadGPSLCObject(str(p))
    c = cases.make_case(5, "U", False, S=2, seed=1)
    g = cases.gpslc_object(gp, c)
    q = str(tmp_path / "g.pk")
    gp.saveGPSLCObject(g, q)
    data = open(q, "rb").read()
    (tmp_path / "t.pk").write_bytes(data[:-8])
    with pytest.raises(ValueError):
        gp.loadGPSLCObject(str(tmp_path / "t.pk"))


def test_sample_block_loading_and_c_header(tmp_path):
    """A rank of a sharded prediction loads only its own block of posterior samples (gpslc_pack_load s0, s1)."""
    from causalgpslc_jl_amd.pack import readPackHeader
    c = cases.make_case(23, "UX", True, S=7, seed=5)
    g = cases.gpslc_object(gp, c)
    p = str(tmp_path / "g.pk")
    gp.saveGPSLCObject(g, p, binary_t=True)
    hd = readPackHeader(p)
    assert (hd["n"], hd["nX"], hd["nU"], hd["S"], hd["binary_t"]) == (23, 3, 2, 7, True)
    # the Python writer of round 1 and the C writer produce the same bytes (format is frozen)
    import struct
    hp = g.hyperparams
    raw = b"GPSLCPK1" + struct.pack("<6q", 23, 3, 2, 7, 1, 0) + struct.pack(
        "<7d", float(hp.nU), hp.nOuter, hp.nMHInner, hp.nESInner, hp.nBurnIn, hp.stepSize, hp.predictionCovarianceNoise)
    for a in (g.X, g.T, g.Y, g.U, g.uyLS, g.xyLS, g.tyLS, g.yNoise, g.yScale):
        raw += np.asfortranarray(a, dtype="<f8").tobytes(order="F")
    assert open(p, "rb").read() == raw
    for s0, s1 in ((0, 7), (2, 5), (6, 7), (3, 3)):
        h = gp.loadGPSLCObject(p, samples=(s0, s1))
        assert gp.getNumPosteriorSamples(h) == s1 - s0
        assert np.array_equal(h.X, g.X) and np.array_equal(h.Y, g.Y)
        if s1 > s0:
            assert np.array_equal(h.U, g.U[:, :, s0:s1]) and np.array_equal(h.xyLS, g.xyLS[:, s0:s1])
            assert np.array_equal(h.tyLS, g.tyLS[s0:s1]) and np.array_equal(h.yScale, g.yScale[s0:s1])
    with pytest.raises(IndexError):
        gp.loadGPSLCObject(p, samples=(5, 9))
    with pytest.raises(OSError):
        gp.loadGPSLCObject(str(tmp_path / "missing.pk"))
    (tmp_path / "long.pk").write_bytes(raw + b"\0" * 8)     # trailing bytes
    with pytest.raises(ValueError):
        gp.loadGPSLCObject(str(tmp_path / "long.pk"))


@pytest.mark.gpu
def test_prediction_from_reloaded_pack_is_identical(tmp_path):
    c = cases.make_case(140, "UX", False, S=3, seed=9)
    g = cases.gpslc_object(gp, c)
    p = str(tmp_path / "g.pk")
    gp.saveGPSLCObject(g, p)
    h = gp.loadGPSLCObject(p)
    a = gp.predict(g, c["doTs"], want_mean_ite=True)
    b = gp.predict(h, c["doTs"], want_mean_ite=True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
