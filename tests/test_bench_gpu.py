"""bench.py's contract on a small dataset: the one-rank line carries every field the driver reads, and the N > 1 path
(strong scaling, LPT shards, one result all-gather per step) runs as two gloo ranks sharing GPU 0
(JTK_BENCH_BACKEND=gloo; the real launch uses RCCL with one rank per GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def last_json(stdout):
    return json.loads([l for l in stdout.splitlines() if l.startswith("{")][-1])


def run_bounded(cmd, env, timeout):
    """subprocess.run(capture_output=True, text=True) in a process group of its own: when the limit passes, the launcher AND its
    ranks are killed (ranks left behind would keep their share of the GPU for the rest of the suite) and the test fails with what
    the ranks printed."""
    import signal
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, err = p.communicate()
        pytest.fail("%s ... did not finish in %d s:\n%s" % (" ".join(cmd[:6]), timeout, err[-4000:]))
    return subprocess.CompletedProcess(cmd, p.returncode, out, err)


def test_one_rank_line_has_the_contract_fields(jtk_lib):
    assert jtk_lib.jtk_lc_device_ok(0) == 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--chunks", "32", "--steps", "2", "--warmup", "1",
                        "--streams", "2"], capture_output=True, text=True, cwd=ROOT, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    line = last_json(r.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "e2e"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["unit"] == "chunks/s" and line["vs_baseline"] is None
    assert line["config"]["workload"].startswith("cfg3") and line["config"]["chunks_total"] == 32
    assert abs(line["value"] - 32 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    # achieved = chunks/s x algorithmic bytes per chunk (B(60, 2000, 2) = 126,216 B nominal)
    assert abs(rf["achieved"] - line["value"] * rf["algorithmic_bytes_per_chunk"] / 1e9) < 1e-9
    assert abs(rf["algorithmic_bytes_per_chunk"] - 126216) < 0.03 * 126216
    sp = rf["serial_pass"]
    assert sp["kernel_ms_sum"] <= sp["wall_ms"]                       # the numbers used for attribution add up
    assert rf["secondary"]["fp64_tflops"] > 0 and rf["secondary"]["chain_cycles_per_proposal"] > 0
    assert line["serial_step_agrees"] and line["chunks_ok"] == 32
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["one_thread"]["cores"] == 1
    # the core count is what the process may really use (affinity mask / cgroup quota), and the ladder is reported
    assert cb["cores"] == cb["cpus"]["affinity"] or cb["cpus"]["cgroup_quota"] is not None
    assert cb["threads_used"] in [r["threads"] for r in cb["scaling"]] and cb["scaling"][0]["threads"] == 1
    assert line["parity_on_cpu_sample"]["labels_equal"] and line["parity_on_cpu_sample"]["max_abs_dlogpost"] < 1e-4
    assert line["e2e"]["matches_resident"] and line["e2e"]["chunks_per_s"] > 0
    # the stage as JTK enters it (refitted model) is checked against the oracle on the same parameters
    rp = line["parity_on_refit_sample"]
    assert rp["labels_equal"] and rp["cluster_num_equal"] and rp["score_bits_equal"] and rp["max_abs_dlogpost"] < 1e-4
    assert line["per_rank"][0]["rank"] == 0 and line["per_rank"][0]["chunks"] == 32


def test_two_rank_strong_scaling_path(jtk_lib):
    """`python3 bench.py --gpus 2` AS TYPED -- no launcher around it (the driver's N > 1 command has none either): bench.py
    starts its own ranks as child processes and relays rank 0's line and the exit code."""
    env = dict(os.environ, JTK_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--chunks", "24", "--steps", "2", "--warmup", "1", "--streams", "2", "--no-e2e",
                        "--weak-probe"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    line = last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["chunks_total"] == 24
    assert line["config"]["chunks_this_rank"] == 12
    assert line["gather_ok"] and line["gathered_reads"] == 24 * 60
    assert line["roofline"]["peak"] == 16000.0
    assert line["cpu_baseline"] is None
    # N > 1: the line explains itself -- every rank's serial-pass breakdown with its slowest chain, and the weak-scaling figure
    pr = line["per_rank"]
    assert [x["rank"] for x in pr] == [0, 1] and all(x["chunks"] == 12 for x in pr)
    for x in pr:
        assert x["pair_hmm_ms"] > 0 and x["chain_ms_summed_over_slices"] >= 0
        if x["chunks_with_a_chain"]:
            assert x["slowest_chunk_id"] is not None and 0 <= x["slowest_chunk_id"] < 24
            assert x["slowest_chunk_chain_ms"] > 0 and x["slowest_chunk_events"] >= 0
    wp = line["weak_probe"]
    assert wp["chunks_per_gpu"] == 24 and wp["chunks_per_s"] > 0 and wp["ms_per_step"] > 0


def test_eight_rank_run_of_the_full_dataset_matches_the_one_rank_run(jtk_lib, tmp_path):
    """north_star's 8-GPU layout on the ONE GPU of this box: bench.py --gpus 8 over gloo (every rank on device 0) on the full
    2,500-chunk partition; the labels / k / scores every rank's shard contributes to the one all-gather must be those of the
    1-rank run, chunk by chunk.  (The collective is gloo here: RCCL has never seen more than one rank of this code.)"""
    import numpy as np
    from jtk_amd import api
    api.trim_cache(0)  # this (pytest) process may still hold pooled workspaces of earlier tests: nine more processes follow
    env = dict(os.environ, JTK_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", JTK_LC_POOL_GB="16")
    common = ["--steps", "1", "--warmup", "0", "--no-e2e", "--no-cpu-baseline", "--no-shard8", "--no-weak-probe"]
    # (the two runs take under a minute together when nothing else holds device memory: eight ranks map 8 x ~30 GB of the 288)
    one = run_bounded([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dump-labels", str(tmp_path / "one.npz")]
                      + common, env, 400)
    assert one.returncode == 0, one.stderr[-3000:]
    r = run_bounded([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                     "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
                     "--gpus", "8", "--streams", "2", "--dump-labels", str(tmp_path / "eight.npz")] + common, env, 400)
    assert r.returncode == 0, "8-rank run failed:\n" + r.stderr[-6000:]
    line = last_json(r.stdout)
    assert line["n_gpus"] == 8 and line["config"]["chunks_total"] == 2500 and line["gather_ok"]
    assert line["gathered_reads"] == 2500 * 60 and 300 <= line["config"]["chunks_this_rank"] <= 325
    a, b = np.load(tmp_path / "one.npz"), np.load(tmp_path / "eight.npz")
    assert np.array_equal(a["label"], b["label"]) and np.array_equal(a["cluster_num"], b["cluster_num"])
    assert np.array_equal(a["score"].view(np.uint64), b["score"].view(np.uint64))


def test_gpus_flag_must_match_the_launch():
    """under a launcher (WORLD_SIZE set) a --gpus that disagrees with it is an error, not a second launch"""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True,
                       cwd=ROOT, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
