"""Chunk sizing of the ensemble driver (causalgpslc.jl_amd/csrc/batch_plan.h): host arithmetic, compiled with g++ and run on the
CPU (the unit-B sub-batch was sized from its target of 128 pairs even for an S = 1, L = 1 call, and the automatic
chunk's 30 % rule subtracted it — draws calls at N >= 4096 failed with GPSLC_ERR_NOMEM on four streams with memory free, and the
unit-A chunk silently halved on two)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_batch_plan_rules(tmp_path):
    exe = str(tmp_path / "batch_plan_test")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "c", "batch_plan_test.cpp")])
    rows = {}
    for line in subprocess.check_output([exe], text=True).splitlines():
        name, bt, bb, ok, fixed = line.split()
        rows[name] = (int(bt), int(bb), int(ok), int(fixed))
    per = (561 + 32) * 131072 + 3 * 32768 + 2 * 32 * 32768
    unit = (1024 + 528) * 131072 + 600000
    assert rows["unitA_1stream"][:3] == (1024, 0, 1)
    assert rows["unitA_2streams"][:3] == (1024, 0, 1)            # round 5's rule halved it to ~540
    bt4 = rows["unitA_4streams"]
    assert bt4[2] == 1 and 500 < bt4[0] < 1024 and bt4[0] * per <= 0.70 * 280 * 2**30 / 4      # the 70 % rule binds
    for k in ("draws_S1_1stream", "draws_S1_4streams"):
        assert rows[k][:3] == (1, 1, 1) and rows[k][3] < 2 * unit   # ONE unit of workspace, not 128
    bt, bb, ok, fixed = rows["draws_S64_4streams"]
    assert ok == 1 and bt == 64 and bb == 64 and bt * per + fixed <= 0.70 * 280 * 2**30 / 4
    bt, bb, ok, fixed = rows["sweep_S8_L200"]
    assert ok == 1 and bt == 8 and bb == 128 and fixed >= 128 * unit + (200 - 128) * 327680
    bt, bb, ok, fixed = rows["small_device"]
    assert ok == 1 and bt >= 1 and 1 <= bb < 128 and bt * per + fixed <= 0.70 * 20 * 2**30
    assert rows["too_small"][2] == 0
    bt, bb, ok, fixed = rows["explicit_2000"]
    assert ok == 1 and bt == 2000
