"""The host-built task order of the persistent factorisation launch (causalgpslc.jl_amd/csrc/task_list.h) replayed on the CPU against
the wait / publish rules of potrf_tasks_kernel: every task sits behind its producers in its own queue (tickets go out in list order
to running workgroups, so this is the launch's no-deadlock argument), every tile is produced exactly once and column by column, a
matrix never leaves its queue — for 2 .. 32 tiles per side, 1 .. 1,000 matrices, group sizes 1 .. 4,096, 1 .. 4 tile rows per strip
task, with and without the back-substitution task, both forms of the augmented row, merged and separate strip(k + 1, k).
The work the launch replaces: src/likelihood.jl:42-43, src/estimation.jl:46."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_task_order_follows_the_kernels_wait_rules(tmp_path):
    exe = str(tmp_path / "task_list_test")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "c", "task_list_test.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    verdict, n = r.stdout.strip().splitlines()[-1].split()
    assert verdict == "OK" and int(n) > 10000
