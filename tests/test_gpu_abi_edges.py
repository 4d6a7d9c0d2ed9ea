"""ABI edges added in round 2: the Y override of the :Y score, device-pointer forms of
rbfKernelLog / processCov, node scores on a ctx without data, several ctxs driven from several threads, and that
the production library ignores the measurement switches of the environment."""
import ctypes as C
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import cases
import gpslc_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_y_logpdf_scores_the_value_it_is_given(gp):
    """HipYNormal.logpdf(y, ...): Gen hands the distribution the value to score; it is the ctx's
    Y only while :Y is constrained to the data (src/model_likelihood.jl:89)."""
    c = cases.make_case(150, "UX", False, S=3, seed=12)
    g = cases.gpslc_object(gp, c)
    y2 = np.random.default_rng(1).standard_normal(150)
    lp_data = gp.yLogpdf(g)
    lp_other = gp.yLogpdf(g, Y_override=y2)
    lp_same = gp.yLogpdf(g, Y_override=c["Y"])
    assert np.array_equal(lp_same, lp_data)
    for s, p in enumerate(cases.samples_of(c)):
        ref = orc.y_logpdf(p.uyLS, p.xyLS, p.tyLS, p.yScale, p.yNoise, p.U, c["X"], c["T"], y2)
        assert abs(lp_other[s] - ref) <= 1e-11 * abs(ref)
    assert not np.allclose(lp_other, lp_data)
    # and the override does not stick
    assert np.array_equal(gp.yLogpdf(g), lp_data)


class _Hip:
    """hipMalloc / hipMemcpy through ctypes on the HIP runtime the library itself uses (no torch needed)."""

    def __init__(self):
        self.rt = C.CDLL("libamdhip64.so.7")      # already loaded by libgpslc_hip.so: same instance
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.bufs = []

    def up(self, x):
        x = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1, order="F"))
        p = self.empty(x.size)
        assert self.rt.hipMemcpy(p, x.ctypes.data_as(C.c_void_p), x.nbytes, 1) == 0
        return p

    def empty(self, count):
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), 8 * count) == 0
        self.bufs.append(p)
        return p

    def down(self, p, count):
        out = np.empty(count)
        assert self.rt.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, 8 * count, 2) == 0
        return out

    def free(self):
        for p in self.bufs:
            self.rt.hipFree(p)


def test_rbf_log_and_process_cov_device_pointer_forms(gp):
    n, d = 130, 3
    rng = np.random.default_rng(3)
    A, B, ls = rng.standard_normal((n, d)), rng.standard_normal((n, d)), np.array([0.7, 1.3, 2.1])
    ref = gp.rbfKernelLog(A, B, ls)                       # host-pointer form
    hip = _Hip()
    dA, dB, dl, out = hip.up(A), hip.up(B), hip.up(ls), hip.empty(n * n)
    ctx = gp.Context(1, 0, 0)
    ctx.check(ctx.lib.gpslc_rbf_log_dev(ctx.h, dA, dB, n, d, dl, 3, out))
    assert np.array_equal(hip.down(out, n * n).reshape(n, n, order="F"), ref)
    ctx.check(ctx.lib.gpslc_process_cov_dev(ctx.h, out, n, 1.7, 0.3, out))     # in place
    assert np.array_equal(hip.down(out, n * n).reshape(n, n, order="F"), gp.processCov(ref, 1.7, 0.3))
    assert ctx.lib.gpslc_rbf_log_dev(ctx.h, None, dB, n, d, dl, 3, out) == -2
    assert ctx.lib.gpslc_process_cov_dev(ctx.h, out, 0, 1.0, 0.0, out) == -3
    hip.free()


def test_scalar_kernel_logit_expit(gp):
    # test/kernel.jl: the scalar form is the matrix form's entry; src/kernel.jl:46-49
    x, y, ls = np.array([1.0, 2.0, 3.0]), np.array([0.5, 2.5, 1.0]), np.array([1.0, 2.0, 0.5])
    assert gp.rbfKernelLogScalar(x, y, ls) == orc.rbf_kernel_log_scalar(x, y, ls)
    assert gp.rbfKernelLogScalar(2.0, 5.0, 3.0) == -1.0
    with pytest.raises(AssertionError):
        gp.rbfKernelLogScalar(x, y, np.ones(2))
    assert gp.logit(0.5) == 0.0 and gp.expit(0.0) == 0.5
    assert abs(gp.expit(gp.logit(0.3)) - 0.3) < 1e-15


def test_node_scores_work_on_a_ctx_without_data(gp):
    n = 140
    rng = np.random.default_rng(8)
    F, tgt = rng.standard_normal((n, 2)), rng.standard_normal(n)
    ctx = gp.Context(n, 0, 0)                             # gpslc_set_data never called
    out = gp.gpLogpdf(F, [0.9, 1.4], [1.2], [0.5], tgt, ctx=ctx)
    cov = orc.process_cov(orc.rbf_kernel_log(F, F, np.array([0.9, 1.4])), 1.2, 0.5)
    ref = orc.mvnormal_logpdf(tgt, cov)
    assert abs(out[0] - ref) <= 1e-11 * abs(ref)


def test_two_ctxs_from_two_threads_are_bit_identical(gp):
    """One ctx per thread on the same device: no shared mutable state in the library."""
    c = cases.make_case(260, "UX", False, S=6, seed=77)
    base = gp.predict(cases.gpslc_object(gp, c), c["doTs"], want_mean_ite=True)
    res = [None, None]

    def work(i):
        g = cases.gpslc_object(gp, c)
        for _ in range(3):
            res[i] = gp.predict(g, c["doTs"], want_mean_ite=True)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    for r in res:
        for a, b in zip(r, base):
            assert np.array_equal(a, b)


def test_production_library_ignores_measurement_switches():
    """GPSLC_GEMM_DIAG & co. exist only in the -DGPSLC_DIAG build (tools/): with them set, the production library
    returns the same bits."""
    code = ("import sys; sys.path[:0] = [%r, %r]; import numpy as np, cases, causalgpslc_jl_amd as gp;"
            "c = cases.make_case(300, 'UX', False, S=3, seed=5);"
            "ms, vs, mi = gp.predict(cases.gpslc_object(gp, c), c['doTs'], want_mean_ite=True);"
            "sys.stdout.write(ms.tobytes().hex() + vs.tobytes().hex())") % (ROOT, os.path.join(ROOT, "tests"))
    env0 = {k: v for k, v in os.environ.items() if not k.startswith("GPSLC_")}
    env0["PYTHONPATH"] = os.path.join(ROOT, "oracle")
    env1 = dict(env0, GPSLC_GEMM_DIAG="2", GPSLC_GEMM_QUEUE="0", GPSLC_SYRK_DIAG="0", GPSLC_FUSE_PANEL="0",
                GPSLC_GEMM_SLOTS="7", GPSLC_ORDER_BLOCK="1", GPSLC_GEMM_DBG="9")
    a = subprocess.run([sys.executable, "-c", code], env=env0, capture_output=True, text=True, timeout=600)
    b = subprocess.run([sys.executable, "-c", code], env=env1, capture_output=True, text=True, timeout=600)
    assert a.returncode == 0 and b.returncode == 0, (a.stderr[-500:], b.stderr[-500:])
    assert a.stdout == b.stdout and len(a.stdout) > 0
    syms = subprocess.run(["nm", "-C", os.path.join(ROOT, "causalgpslc.jl_amd", "csrc", "libgpslc_hip.so")],
                          capture_output=True, text=True).stdout
    inst = sorted(set(l.split("tile_gemm_nt_kernel")[1].split("(")[0] for l in syms.splitlines()
                      if "__device_stub__tile_gemm_nt_kernel" in l))
    # tile_gemm_nt_kernel<ACC, DIAG>: no timing-only (DIAG != 0) instantiation in the production library
    assert inst and all(i.strip("<> ").split(",")[1].strip() == "0" for i in inst), inst


def test_plain_c_consumer(tmp_path):
    """The ABI consumed from plain C (tests/c/abi_consumer.c, compiled with gcc against include/gpslc_hip.h and linked
    to libgpslc_hip.so): closed-form node scores at n = 150 (LDS-resident kernel), 272 (left-looking kernel) and 700
    (tiled path), the fused node call, an argument error and a not-positive-definite matrix; then gpslc_set_data /
    gpslc_predict / gpslc_set_ensemble: exact zeros when every treatment equals the intervention level (n = 300) and a
    sample predicted alone at its place in the ensemble drawing the ensemble's normals."""
    inc = os.path.join(ROOT, "include")
    libdir = os.path.join(ROOT, "causalgpslc.jl_amd", "csrc")
    exe = str(tmp_path / "abi_consumer")
    subprocess.run(["gcc", "-std=c99", "-O1", "-I", inc, os.path.join(ROOT, "tests", "c", "abi_consumer.c"), "-o", exe,
                    "-L", libdir, "-lgpslc_hip", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-500:]
    assert "gfx950" in r.stdout and r.stdout.strip().endswith("ok")


def test_nodes_draw_edges(gp):
    """gpslc_nodes_draw argument / failure behaviour: count = 0 is a no-op, NULL output is argument #4, a covariance that
    is not positive definite comes back as LAPACK-style info (never a crash), and the optional log-density by-product is
    the score gpslc_nodes_logpdf gives."""
    from causalgpslc_jl_amd import _lib, api
    n = 150
    ctx = gp.Context(n, 0, 0)
    lib = ctx.lib
    assert lib.gpslc_nodes_draw(ctx.h, 0, None, None, None) == 0
    rng = np.random.default_rng(3)
    F, ls, z = rng.standard_normal((n, 3)), np.array([1.0, 1.5, 0.8]), rng.standard_normal(n)
    arr, keep = api._marshal_nodes([(F, ls, 1.2, 0.5, z)], ctx)
    assert lib.gpslc_nodes_draw(ctx.h, 1, C.cast(arr, C.c_void_p), None, None) == -4
    out, lp = np.empty((n, 1), order="F"), np.empty(1)
    assert lib.gpslc_nodes_draw(ctx.h, 1, C.cast(arr, C.c_void_p), out.ctypes.data_as(C.c_void_p), lp.ctypes.data_as(C.c_void_p)) == 0
    assert abs(lp[0] - gp.nodesLogpdf([(F, ls, 1.2, 0.5, z)], ctx)[0]) <= 1e-12 * abs(lp[0])
    K = orc.process_cov(orc.rbf_kernel_log(F, F, ls), 1.2, 0.5)
    assert np.allclose(out[:, 0], np.linalg.cholesky(K) @ z, rtol=1e-10, atol=1e-12)
    # negative noise: scale * exp(...) - 2 I is indefinite -> positive info, PosDefException through the mirror
    with pytest.raises(gp.PosDefException):
        gp.nodesDraw([(F, ls, 1.2, -2.0, z)], ctx)
    assert ctx.last_info(1)[0] > 0


def test_set_ensemble_argument_checks_and_default(gp):
    """gpslc_set_ensemble(ctx, sample_offset, S_total): negative offsets / totals and an offset beyond the ensemble are
    argument errors (-2 / -3); (0, 0) restores the default; the placement changes the library's own normals only."""
    ctx = gp.Context(8, 0, 0)
    lib = ctx.lib
    assert lib.gpslc_set_ensemble(ctx.h, -1, 10) == -2
    assert lib.gpslc_set_ensemble(ctx.h, 0, -1) == -3
    assert lib.gpslc_set_ensemble(ctx.h, 10, 10) == -3
    assert lib.gpslc_set_ensemble(ctx.h, 3, 10) == 0
    assert lib.gpslc_set_ensemble(ctx.h, 0, 0) == 0
    assert lib.gpslc_set_ensemble(None, 0, 0) == -1
    ctx.close()


def test_ensemble_placement_reproduces_the_single_call_streams(gp):
    """Samples [2, 5) of a 6-sample ensemble predicted on their own with set_ensemble(2, 6) draw exactly the normals the
    6-sample call draws for them (stream id = (offset + s) + S_total * l), for every level."""
    import cases
    c = cases.make_case(40, "UX", False, S=6, seed=77)
    g = cases.gpslc_object(gp, c)
    doTs, spp = np.array([0.1, 0.9]), 3
    full = gp.predict(g, doTs, spp=spp, seed=5, want_draws=True)[3]               # (L, n, S * spp)
    from causalgpslc_jl_amd.sharded import slice_object
    part = slice_object(g, 2, 5)
    part.ctx().set_ensemble(2, 6)
    got = gp.predict(part, doTs, spp=spp, seed=5, want_draws=True)[3]
    assert np.array_equal(got, full[:, :, 2 * spp:5 * spp])
    part.ctx().set_ensemble(0, 0)
    again = gp.predict(part, doTs, spp=spp, seed=5, want_draws=True)[3]
    assert not np.array_equal(again, got)                                        # default numbering: other streams
