"""Literal restatement <-> structured restatement <-> 80-bit evaluation, and the committed goldens."""
import os

import numpy as np
import pytest

import cases
import gpslc_oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gpslc_golden.npz")


@pytest.mark.parametrize("n,shape,bt", [(3, "UX", False), (24, "UX", True), (24, "U", False), (60, "X", False),
                                         (60, "T", True), (150, "UX", False)])
def test_literal_vs_structured_vs_longdouble(n, shape, bt):
    c = cases.make_case(n, shape, bt, seed=7)
    e = cases.oracle_expected(c)
    smp = cases.samples_of(c)
    for s, p in enumerate(smp):
        ms, vs, logdet, quad = orc.structured_sate(p, c["X"], c["T"], c["Y"], c["doTs"])
        lp = -0.5 * (n * np.log(2 * np.pi) + logdet + quad)
        assert np.isclose(lp, e["logpdf"][s], rtol=1e-11, atol=1e-9)
        for l, doT in enumerate(c["doTs"]):
            # SURVEY.md §8d tolerance: 1e-6 rel (+ mixed abs for the variance); observed ~1e-12
            assert abs(ms[l] - e["meanSATE"][s, l]) <= 1e-9 * abs(e["meanSATE"][s, l]) + 1e-13
            assert abs(vs[l] - e["varSATE"][s, l]) <= 1e-9 * abs(e["varSATE"][s, l]) + 1e-12 * p.yScale
            m, cv = orc.structured_ite(p, c["X"], c["T"], c["Y"], doT)
            assert np.max(np.abs(m - e["meanITE"][:, s, l])) <= 1e-9 * np.max(np.abs(e["meanITE"][:, s, l])) + 1e-13
            if n <= 60:
                mld, cld, msld, vsld = orc.literal_sate_longdouble(p, c["X"], c["T"], c["Y"], doT)
                assert abs(float(msld) - e["meanSATE"][s, l]) <= 1e-9 * abs(float(msld)) + 1e-13
                assert abs(float(vsld) - e["varSATE"][s, l]) <= 1e-8 * abs(float(vsld)) + 1e-12 * p.yScale
                assert abs(float(vsld) - vs[l]) <= 1e-8 * abs(float(vsld)) + 1e-12 * p.yScale
                assert np.max(np.abs(cld.astype(float) - e["covITE"][s, l])) <= 1e-9 * p.yScale


def test_oracle_reproduces_committed_goldens():
    g = np.load(GOLD)
    for i, (n, shape, bt) in enumerate(cases.GOLDEN_GRID):
        key = cases.golden_name(n, shape, bt)
        c = cases.make_case(n, shape, bt, seed=i)
        for k in ("T", "Y", "tyLS", "yNoise", "yScale", "doTs"):
            assert np.array_equal(g[f"{key}/in/{k}"], c[k]), (key, k)   # generator is deterministic
        if n > 24:
            continue   # covered by the n <= 24 cases; keeps the CPU suite short
        e = cases.oracle_expected(c)
        for k in ("meanITE", "meanSATE", "varSATE", "logpdf", "covITE"):
            assert np.allclose(g[f"{key}/out/{k}"], e[k], rtol=1e-10, atol=1e-13), (key, k)


def test_philox_known_answer():
    # Random123 known-answer vectors for Philox4x32-10
    out = orc.philox4x32_10(np.zeros((1, 4), dtype=np.uint32), (0, 0))
    assert [hex(int(v)) for v in out[0]] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    out = orc.philox4x32_10(np.full((1, 4), 0xFFFFFFFF, dtype=np.uint32), (0xFFFFFFFF, 0xFFFFFFFF))
    assert [hex(int(v)) for v in out[0]] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    out = orc.philox4x32_10(np.array([[0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344]], dtype=np.uint32),
                            (0xA4093822, 0x299F31D0))
    assert [hex(int(v)) for v in out[0]] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_philox_normals_moments():
    z = orc.philox_normals(1234, 7, 200000)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01


@pytest.mark.gpu      # no GPU call in it: it rides with the GPU-box run because that host has the cores (about 15 s there)
@pytest.mark.skipif((os.cpu_count() or 1) < 32, reason="one 16384 x 16384 Cholesky + 2 GB matrices: 6 minutes on 8 cores")
def test_config5_literal_golden_agrees_with_the_structured_restatement_at_full_size():
    """tests/golden/config5_literal.npz holds the LITERAL restatement's outputs at BASELINE configs[4]'s size (N = 16384,
    D 16, nU 4, binary treatment; 11 minutes and ~30 GB to regenerate, tests/golden/make_golden_config5.py).  Here, on the
    CPU: its inputs are what the generator yields today, and the STRUCTURED restatement — the algebra the HIP path uses —
    reproduces its MeanSATE / VarSATE at the same size (one Cholesky of a 16384 x 16384 matrix: about a minute)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "synth", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "causalgpslc.jl_amd", "synth.py"))
    synth = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(synth)
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config5_literal.npz"))
    n, D, K, S = (int(v) for v in gold["shape"])
    X, T, Y, objid = synth.make_dataset(n, D, binary_t=True)
    post = synth.make_posterior(n, D, K, S, objid)
    chk = np.array([X.sum(), T.sum(), Y.sum(), post["U"].sum(), post["uyLS"].sum(), post["xyLS"].sum(),
                    post["tyLS"][0], post["yNoise"][0], post["yScale"][0]])
    assert np.allclose(chk, gold["in_checksums"], rtol=1e-13, atol=0)
    assert abs(float(gold["meanITE"].mean()) - float(gold["meanSATE"])) <= 1e-12 * abs(float(gold["meanSATE"])) + 1e-16
    p = orc.PosteriorSample(post["uyLS"][:, 0], post["xyLS"][:, 0], float(post["tyLS"][0]), float(post["yNoise"][0]),
                            float(post["yScale"][0]), post["U"][:, :, 0])
    ms, vs, _, _ = orc.structured_sate(p, X, T, Y, np.array([float(gold["doT"])]))
    assert abs(ms[0] - float(gold["meanSATE"])) <= 1e-9 * abs(float(gold["meanSATE"]))
    assert abs(vs[0] - float(gold["varSATE"])) <= 1e-9 * abs(float(gold["varSATE"]))
