"""-m "not gpu": bench.py's own launcher.  `python bench.py --gpus 2` with no launcher environment must start two rank
processes itself (fresh children, started before the parent touches any GPU), run the sharded steps, gather every
rank's per-image records with ONE all-gather and print ONE JSON line that says how many ranks the process group saw.
Driven here on CPU ranks over gloo with --stub-cpu (no model, no kernels: plumbing only)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.timeout(300)
def test_bench_gpus_2_launches_its_own_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--stub-cpu"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                  # rank 0 alone prints, one line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["launcher"] == "bench.py spawned the ranks itself"
    assert j["config"]["global_batch"] == 8 and j["steps"] == 3 and j["scaling"] == "weak"
    assert j["metrics_gathered"]["images"] == 2 * 3 * 4               # every record of every rank arrived
    assert j["per_rank_images_per_s"]["min"] > 0 and j["data"] == "stub"
    assert abs(j["value"] - 2 * 3 * 4 / (j["ms_per_step"] * 3 / 1e3)) / j["value"] < 1e-3


@pytest.mark.timeout(300)
def test_bench_gpus_2_with_three_slots_in_flight_per_rank():
    """the pipelined mode's slot bookkeeping under a process group -- two ranks, three slots each, the first
    ROOFLINE_STEPS steps of the timed region on slot 0 alone, the rest round-robin; every step's records reach the ONE all-gather
    exactly once (unique image ids over ranks x steps x batch) and the JSON says which slot served how many steps."""
    r = _run(["--gpus", "2", "--steps", "20", "--warmup", "1", "--batch", "2", "--inflight", "3", "--stub-cpu"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["inflight"] == 3 and j["config"]["inflight"] == 3
    assert j["metrics_gathered"]["images"] == 2 * 20 * 2               # bench.py itself asserts the ids are 0 .. N-1, each once
    assert sum(j["slot_steps"]) == 20 and j["slot_steps"] == [7, 7, 6]  # step 0 alone on slot 0, then s % 3
    assert j["host_threads"]["issuing"] == 1
    assert j["config"]["value_sequential"] is None or j["config"]["value_sequential"] > 0


@pytest.mark.timeout(300)
def test_bench_gpus_8_on_a_sixteen_cpu_share_starts_and_finishes_quickly():
    """what an 8-GPU driver run meets first -- eight ranks starting at once on the job's CPU share (16 CPUs on
    the driver's box; at most that many here).  Each rank caps torch's intra-op pool at its share (not 8 x all logical CPUs), the
    eight ranks rendezvous, three slots each, every record arrives, start-up seconds are reported per rank, the whole job < 60 s."""
    import time
    cpus = sorted(os.sched_getaffinity(0))[:16]

    def restrict():
        os.sched_setaffinity(0, cpus)

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "12", "--warmup", "1", "--batch", "2",
                        "--inflight", "3", "--stub-cpu"], env=env, capture_output=True, text=True, timeout=240, preexec_fn=restrict)
    wall = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert j["n_gpus"] == 8 and j["ranks_seen"] == 8 and j["inflight"] == 3
    assert j["metrics_gathered"]["images"] == 8 * 12 * 2
    assert j["host_threads"]["torch_intra_op"] <= max(1, len(cpus) // 8), j["host_threads"]
    assert 0 < j["per_rank_startup_s"]["min"] <= j["per_rank_startup_s"]["max"] < 60
    assert j["per_rank_images_per_s"]["min"] > 0
    assert wall < 60, wall


@pytest.mark.timeout(300)
def test_bench_rank_failure_is_a_nonzero_exit():
    # --gpus 2 inside a 3-rank launcher environment: every rank refuses, the job fails loudly
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "2", "--stub-cpu"],
             {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "1"}, timeout=120)
    assert r.returncode != 0
