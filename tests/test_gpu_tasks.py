"""The persistent factorisation launch (potrf_tasks_kernel, gpslc_set_task_schedule) against the one-launch-per-column
schedule it replaces at small tile counts.  Both run the same device functions on every tile in the same order, so every
output must be equal BIT FOR BIT — a stale read between two workgroups (a missing release / acquire, a task that starts
before its producer has published) shows up as a difference, not as a tolerance question.  The work replaced:
src/likelihood.jl:42-43, src/estimation.jl:46 (the reference factorises CovWWp three times per unit, one matrix at a time).
Parity with the oracle at these sizes is covered by the other GPU suites, which now run through this launch by default."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def _same(a, b):
    assert a.shape == b.shape
    assert np.array_equal(a, b)


def _both(gp, c, doT, max_batch=0, group=0, **kw):
    out = []
    for tiles in (32, 0):                      # persistent launch (from 2 tiles per side and ONE matrix on), one launch per column
        g = cases.gpslc_object(gp, c)
        g.ctx().set_task_schedule(2, tiles, 1, group)
        if max_batch:
            g.ctx().set_tuning(max_batch, 0, 0)
        out.append(gp.predict(g, doT, want_mean_ite=True, **kw))
    return out


@pytest.mark.parametrize("n,S,L", [(300, 37, 1), (1024, 70, 1), (1000, 9, 20), (640, 3, 31), (257, 1, 2), (896, 19, 3), (200, 5, 1)])
def test_task_launch_equals_the_per_column_schedule_bit_for_bit(gp, n, S, L):
    """nt = 3 ... 8 tiles per side (odd and even numbers of strips per column: tasks of one and of two tile rows), 1 / 2 blocks
    of augmented rows (L = 1, 20, 31), fewer matrices than queues (S = 1, 3: workgroups on the other XCDs take tickets of the
    queues that hold work); MeanITE compares the back-substitution task with launch_backsolve's kernels."""
    c = cases.make_case(n, "UX", False, S=S, seed=n + S)
    doT = np.linspace(-0.5, 0.7, L)
    a, b = _both(gp, c, doT)
    for x, y in zip(a, b):
        _same(x, y)


@pytest.mark.parametrize("n,S,L", [(1100, 5, 2), (1536, 3, 1), (2048, 9, 1), (2500, 2, 20), (3072, 3, 1), (3500, 2, 1), (4096, 9, 3)])
def test_task_launch_as_one_panel_beyond_eight_tiles(gp, n, S, L):
    """nt = 9, 12, 16, 20, 24, 28, 32 (the descriptor's limit): with the default panel width the persistent launch factorises the whole
    matrix as ONE left-looking panel; the per-column schedule it is compared with runs panels of 8 + trailing updates — different
    launches, the same MFMA chain per tile (ascending k): bit-identical.  A panel width given through gpslc_set_tuning keeps the
    panel schedule beyond it."""
    c = cases.make_case(n, "UX", False, S=S, seed=n + S)
    doT = np.linspace(-0.5, 0.7, L)
    out = []
    for tiles, panel in ((32, 0), (0, 0), (32, 8)):
        g = cases.gpslc_object(gp, c)
        g._ctx = gp.Context(g.getN(), g.getNX(), g.getNU(), profile=True)      # HIP-event records: which schedule really ran
        g._ctx.set_data(g.X, g.T, g.Y)
        g.ctx().set_task_schedule(2, tiles, 1, 0)
        if panel:
            g.ctx().set_tuning(0, panel, 0)
        g.ctx().profile_reset()
        out.append(gp.predict(g, doT, want_mean_ite=True))
        launches = g.ctx().profile_get(4)[0]
        assert (launches > 0) == (tiles > 0 and not panel)
    for other in out[1:]:
        for x, y in zip(out[0], other):
            _same(x, y)


@pytest.mark.parametrize("n,S,L", [(640, 5, 33), (1024, 9, 64), (1000, 3, 101), (300, 11, 126), (2048, 3, 40), (4096, 2, 64)])
def test_task_launch_with_a_level_sweep_of_more_than_32_right_hand_sides(gp, n, S, L):
    """More than 32 right-hand sides (the sweep of src/prediction.jl:24-33 runs ~100 levels per posterior sample): the augmented
    row no longer rides with the diagonal tasks — it is an ordinary tile row of the task list, strip(nt, k) with its own column
    update over the live rows, exactly the work item the per-column launches give it: bit-identical, MeanITE of every level
    included (the back-substitution task reads right-hand side 0 of that row)."""
    c = cases.make_case(n, "UX", False, S=S, seed=n + L)
    doT = np.linspace(-0.8, 0.9, L)
    a, b = _both(gp, c, doT)
    for x, y in zip(a, b):
        _same(x, y)


@pytest.mark.parametrize("group", [1, 3, 64])
def test_task_order_group_size_and_chunking_do_not_change_results(gp, group):
    """Group size 1 puts a task right behind its producer in the queue (consumers really wait on the progress words);
    chunks of 5 matrices leave queues with 0 or 1 matrix."""
    c = cases.make_case(700, "UX", True, S=23, seed=4)
    a, b = _both(gp, c, [0.0, 1.0], group=group)
    for x, y in zip(a, b):
        _same(x, y)
    a, b = _both(gp, c, [0.0, 1.0], max_batch=5, group=group)
    for x, y in zip(a, b):
        _same(x, y)


def test_task_launch_draws_and_logpdf_paths(gp):
    """Unit B (full ITE covariance + draws) keeps its own schedule after the task launch has factorised A; the :Y score
    (logdet and quadratic form from the same factorisation) goes through it too."""
    c = cases.make_case(400, "UX", False, S=6, seed=9)
    outs = []
    for tiles in (8, 0):
        g = cases.gpslc_object(gp, c)
        g.ctx().set_task_schedule(2, tiles, 1, 0)
        ms, vs, mi, dr = gp.predict(g, [0.1, 0.6], want_mean_ite=True, spp=3, seed=5, want_draws=True)
        outs.append((ms, vs, mi, dr, gp.yLogpdf(g)))
    for x, y in zip(*outs):
        _same(x, y)


@pytest.mark.parametrize("bad", [0, 5])
def test_failing_pivot_is_reported_by_the_task_launch(gp, bad):
    """A negative yNoise makes A = K - 0.5 I indefinite for ONE posterior sample: the same 1-based pivot comes back from both
    schedules (first and later queue, first and later tile column), and the other samples report 0."""
    n, S = 520, 11
    c = cases.make_case(n, "UX", False, S=S, seed=31)
    c["yNoise"][bad] = -0.5
    infos, per_sample = [], []
    for tiles in (8, 0):
        g = cases.gpslc_object(gp, c)
        g.ctx().set_task_schedule(2, tiles, 1, 0)
        with pytest.raises(gp.PosDefException) as ei:
            gp.predict(g, [0.3])
        infos.append(ei.value.info)
        per_sample.append(g.ctx().last_info(S))
    assert infos[0] == infos[1] and 0 < infos[0] <= n
    _same(per_sample[0], per_sample[1])
    assert per_sample[0][bad] == infos[0] and not np.delete(per_sample[0], bad).any()


def test_set_task_schedule_arguments(gp):
    c = cases.make_case(24, "UX", False, S=2, seed=2)
    g = cases.gpslc_object(gp, c)
    lib, h = g.ctx().lib, g.ctx().h
    assert lib.gpslc_set_task_schedule(h, 33, 8, 0, 0) == -2
    assert lib.gpslc_set_task_schedule(h, 2, 33, 0, 0) == -3
    assert lib.gpslc_set_task_schedule(h, 2, 8, 1, 5000) == -5
    assert lib.gpslc_set_task_schedule(h, 0, -1, 0, 0) == 0
    assert lib.gpslc_set_task_schedule(None, 2, 8, 1, 8) == -1


def test_a_broken_hand_off_times_out_instead_of_hanging(tmp_path):
    """The launch's safety net: a consumer's poll is bounded, the first one to exceed its bound sets the time-out word, every
    workgroup stops waiting, the launch drains and the call returns GPSLC_ERR_INTERNAL — never a hung GPU.  Exercised with the
    measurement build (GPSLC_TASK_FENCE bit 6: diag(1) of matrix 0 never publishes its progress) in a fresh process, because that
    build reads its switches once; the production library has no such switch (it never reads the environment)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    diag = os.path.join(root, "causalgpslc.jl_amd", "csrc", "libgpslc_hip_diag.so")
    if not os.path.exists(diag):
        pytest.skip("measurement build not present (make -C causalgpslc.jl_amd/csrc diag)")
    code = (
        "import sys, time, numpy as np\n"
        f"sys.path[:0] = [{root!r}, {os.path.join(root, 'tests')!r}, {os.path.join(root, 'oracle')!r}]\n"
        "import causalgpslc_jl_amd as gp, cases\n"
        "gp._lib.LIB_PATH = gp._lib.LIB_PATH.replace('libgpslc_hip.so', 'libgpslc_hip_diag.so')\n"
        "c = cases.make_case(520, 'UX', False, S=9, seed=3)\n"
        "g = cases.gpslc_object(gp, c)\n"
        "g.ctx().set_task_schedule(2, 32, 1, 0)\n"
        "t0 = time.time()\n"
        "try:\n"
        "    gp.predict(g, [0.2], want_mean_ite=True)\n"
        "    print('NO ERROR')\n"
        "except gp.GPSLCError as e:\n"
        "    print('ERROR', e.status, str(e))\n"
        "print('seconds', round(time.time() - t0, 2))\n")
    env = dict(os.environ, GPSLC_TASK_FENCE=str(0x30 | 0x40))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
    out = r.stdout
    assert r.returncode == 0, (out, r.stderr[-2000:])
    assert "ERROR" in out and "timed out" in out, out
    secs = float(out.strip().splitlines()[-1].split()[1])
    assert secs < 60, out
