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
