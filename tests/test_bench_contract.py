"""Host-side checks of bench.py's contract (no GPU): --gpus N never yields a silent 1-GPU number, the committed PMC
summary is tied to the kernel source it was taken from, and the JSON keys the driver reads are spelled as promised."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    if env:
        e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True,
                          timeout=300)


def test_gpus_flag_refuses_a_box_with_fewer_gpus():
    r = _run(["--gpus", "8", "--steps", "1"])          # this container has no GPU at all
    assert r.returncode == 2 and "refusing" in r.stderr and r.stdout.strip() == ""


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--steps", "1"], env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode == 3 and "WORLD_SIZE=1" in r.stderr and r.stdout.strip() == ""


def test_pmc_summary_is_tied_to_the_kernel_source():
    sys.path.insert(0, ROOT)
    import bench
    data = open(bench.KERNEL_SRC, "rb").read()
    sha = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
    assert bench.git_blob_sha(bench.KERNEL_SRC) == sha
    # one summary per dominant kernel: the trailing-update kernel (rounds 1-5, and the panel schedule) and the persistent task
    # launch (round 6: the default schedule up to N = 4096)
    for stem in ("pmc_tile_gemm", "pmc_potrf_tasks"):
        path = bench.pmc_summary_path(stem)
        traffic, note = bench.pmc_traffic(True, stem)
        if not os.path.exists(path):
            assert traffic is None
            continue
        pm = json.load(open(path))
        assert set(("hbm_bytes_per_launch", "kernel_src_sha")) <= set(pm)
        # the line carries the number only while the summary describes the kernel in this tree; otherwise it is withheld
        if pm["kernel_src_sha"] == sha:
            assert traffic == pm["hbm_bytes_per_launch"]
        else:
            assert traffic is None and "STALE" in note
        assert bench.pmc_traffic(False, stem)[0] is None


def test_measured_constants_are_withheld_when_their_sources_change(tmp_path, monkeypatch):
    """VERDICT r04 weak #12 / ADVICE r04: the HBM bytes per sample of config 2 and the fp64-VALU issue times of the Gram build and
    the MeanITE pass are measurements; bench.py reads them from a committed file stamped with the git blob hashes of the
    sources they were taken on and withholds them on a mismatch, like roofline.traffic."""
    sys.path.insert(0, ROOT)
    import bench
    src = os.path.join("causalgpslc.jl_amd", "csrc", "k_gram.hip")
    good = bench.git_blob_sha(os.path.join(ROOT, src))
    f = tmp_path / "constants.json"
    f.write_text(json.dumps({"gram_valu_us_n4096": {"value": 17.0, "source_shas": {src: good}, "from": "profiles/x.md"},
                             "stale": {"value": 1.0, "source_shas": {src: "0" * 40}, "from": "profiles/y.md"}}))
    monkeypatch.setattr(bench, "BENCH_CONSTANTS", str(f))
    v, note = bench.measured_constant("gram_valu_us_n4096")
    assert v == 17.0 and "match" in note
    v, note = bench.measured_constant("stale")
    assert v is None and "STALE" in note and "k_gram.hip" in note
    assert bench.measured_constant("absent")[0] is None
    monkeypatch.setattr(bench, "BENCH_CONSTANTS", str(tmp_path / "nope.json"))
    assert bench.measured_constant("gram_valu_us_n4096")[0] is None
    # the committed file, when present, must name only sources that exist
    real = os.path.join(ROOT, "profiles", "r05_bench_constants.json")
    if os.path.exists(real):
        for name, rec in json.load(open(real)).items():
            assert rec["value"] > 0 and rec["source_shas"], name
            for rel in rec["source_shas"]:
                assert os.path.exists(os.path.join(ROOT, rel)), (name, rel)


def test_launcher_counts_gpus_without_touching_them():
    sys.path.insert(0, ROOT)
    import bench
    n = bench.count_gpus_sysfs()          # KFD topology; None where there is no amdgpu driver (this container)
    assert n is None or n >= 0


def test_committed_bench_line_carries_the_contract_keys():
    d = json.loads(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "sate_rel_err", "units"):
        assert k in d, k
    assert d["metric"].endswith("SATE rel-err vs CPU") and "workload" in d["config"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"])
    assert d["sate_rel_err"]["ok"] and d["sate_rel_err"]["mean"] < 1e-6 and d["sate_rel_err"]["var"] < 1e-6
    assert {"A", "B", "C"} <= set(d["units"])
    # round 3: every number of the line is oracle-backed and no roofline fraction exceeds 1
    assert d["units"]["B"]["parity"]["ok"] and d["units"]["B"]["parity"]["draw_err"] <= d["units"]["B"]["parity"]["draw_bound"]
    assert 0 < d["units"]["C"]["frac"] <= 1 and 0 < d["units"]["B"]["frac"] <= 1 and 0 < d["roofline"]["frac"] <= 1
    assert d["config4"]["levels"] == 64 and d["config4"]["parity"]["ok"]
    assert {"c2", "c5"} <= set(d["configs"]) and all(c["parity"]["ok"] for c in d["configs"].values())
    assert d["units"]["A"]["ceiling_shared_datapath_units_per_s"] < d["units"]["A"]["ceiling_units_per_s"]
    # round 4: config 2 as BASELINE states it (one call, S = 1000) within 10 % of the 8,192-per-step figure; config 2's
    # binding roof (HBM) stated beside the MFMA fraction; the tight draw guard applied where the conditioning permits it
    c2, c2l = d["configs"]["c2"], d["configs"]["c2_literal"]
    assert c2l["samples_per_step"] == 1000 and c2l["parity"]["ok"] and abs(c2l["value"] / c2["value"] - 1.0) <= 0.10
    assert 0 < c2["hbm"]["frac"] <= 1 and c2["hbm"]["bytes_per_unit"] > 8 * 1024 ** 2 / 2
    pb = d["units"]["B"]["parity"]
    assert pb["cond"] < 1e8 and pb["draw_tight_bound"] is not None and pb["draw_err"] <= pb["draw_tight_bound"]
    assert d["roofline"]["traffic"] is not None and d["roofline"]["traffic"] > 0
    # round 5: the spread of the value; config 3 as BASELINE states it (ONE call, S = 5000) within 2 % of the headline; the LITERAL
    # restatement inside the line for configs 4 and 5; unit C on the streaming draw kernel; measured constants present (not withheld)
    assert len(d["value_runs"]) == 3 and d["value_runs"][0] == d["value"]
    assert max(d["value_runs"]) / min(d["value_runs"]) - 1.0 <= 0.02
    c3l = d["configs"]["c3_literal"]
    assert c3l["samples_per_step"] == 5000 and c3l["parity"]["ok"] and abs(c3l["value"] / d["value"] - 1.0) <= 0.02
    lg = d["configs"]["c5"]["parity"]["literal_golden"]
    assert lg["ok"] and max(lg["mean_rel_err"], lg["var_rel_err"], lg["mean_ite_rel_err"]) < 1e-6 and "LITERAL" in lg["reference"]
    lu = d["config4"]["parity"]["literal_unit"]
    assert lu["ok"] and max(lu["mean_rel_err"], lu["var_rel_err"], lu["mean_ite_rel_err"]) < 1e-9
    assert d["units"]["C"]["frac"] >= 0.65 and d["units"]["B"]["frac"] >= 0.75
    assert c2["hbm"]["bytes_per_unit"] is not None and "source hashes match" in c2["hbm"]["note"]
    assert d["units"]["A"]["ceiling_shared_datapath_units_per_s"] is not None


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_two_rank_rehearsal_line_carries_the_whole_contract():
    """The N > 1 path of bench.py, end to end, on a one-GPU box: two ranks under torch.distributed.run share device 0 over
    gloo (GPSLC_BENCH_REHEARSAL=1).  Rank 0 writes the node's posterior pack, every rank loads its own block, both timed
    regions are sharded and gathered, rank 0 runs the CPU leg — and the ONE JSON line must carry what the N = 1 line
    carries: roofline, cpu_baseline, sate_rel_err, config4.parity, with n_gpus = 2 and the rehearsal label.  (No scaling
    figure is read off this: it checks the contract, not the speed.)"""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env["GPSLC_BENCH_REHEARSAL"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--", "--gpus", "2", "--steps", "1",
           "--warmup", "1", "--n", "640", "--d", "4", "--nu", "1", "--samples-per-step", "24", "--config4-levels", "8",
           "--config4-steps", "1", "--cpu-units", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "REHEARSAL" in d["data"] and d["scaling"] == "weak"
    for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype",
              "config", "roofline", "cpu_baseline", "sate_rel_err", "config4", "units", "configs"):
        assert k in d, k
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"]) and 0 < d["roofline"]["frac"] <= 1
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"]) and d["cpu_baseline"]["value"] > 0
    # torch.distributed.run exports OMP_NUM_THREADS=1; the CPU leg restores the BLAS pool of an N = 1 run for its duration
    assert d["cpu_baseline"]["cores"] > 1 or (os.cpu_count() or 1) == 1
    assert d["sate_rel_err"]["ok"] and d["sate_rel_err"]["units"] == 2
    assert d["config4"]["parity"]["ok"] and d["config4"]["levels"] == 8
    assert "2 rank(s)" in d["config4"]["workload"] and "2 rank(s)" in d["config"]["sharding"]
    assert d["units"].startswith("N=1 only") and d["configs"].startswith("N=1 only")
    # whole-job value: both ranks' samples over the slower rank's time
    assert abs(d["value"] - 2 * 24 * 1 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]


@pytest.mark.gpu
def test_one_rank_over_rccl_runs_the_collective_path():
    """the real `nccl` (= RCCL) backend on the box the driver uses, every round.  One rank under
    torch.distributed.run (a fresh child process, started before anything here touches the GPU) with GPSLC_BENCH_FORCE_DIST=1:
    RCCL initialisation on the rank's device, the all_gather of the SATE arrays of both timed regions, the barriers and the
    all_reduce(MAX) of the step time run on device tensors exactly as in an N > 1 job; the line must still be the N = 1 line."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env["GPSLC_BENCH_FORCE_DIST"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--", "--gpus", "1", "--steps", "2",
           "--warmup", "1", "--n", "640", "--d", "4", "--nu", "1", "--samples-per-step", "48", "--config4-levels", "8",
           "--config4-steps", "1", "--no-cpu-baseline", "--no-units", "--no-configs", "--repeats", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and "REHEARSAL" not in d["data"]
    assert "1 rank(s)" in d["config"]["sharding"] and d["config4"]["parity"]["ok"]
    assert abs(d["value"] - 48 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
