"""The three launchers of the multi-GPU path -- `bench.py --gpus N`, `dlrm -ll:gpu N` (host/launcher.cc) and
`run_dlrm.py -ll:gpu N` -- start their own ranks before anything touches a GPU
[ref: one command with -ll:gpu N, examples/cpp/DLRM/run_random.sh:3, src/runtime/cpp_driver.cc:22-44].

CPU tests: process management, rendezvous and the world_size-2 path over gloo with the oracle as kernel library.
GPU tests (-m gpu): the same entry points with a 1-rank RCCL group on the box's one GPU.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from dlrm_flexflow_amd import build

EXE = os.path.join(ROOT, "dlrm_flexflow_amd", "host", "dlrm_testing")      # the -DFFM_TESTING build of the launcher: the hooks below are compiled out of `dlrm`
RUN_DLRM = os.path.join(ROOT, "dlrm_flexflow_amd", "run_dlrm.py")
BENCH = os.path.join(ROOT, "bench.py")
SMALL = ["-b", "64", "--arch-sparse-feature-size", "8", "--arch-embedding-size", "100-200-50", "--arch-mlp-bot", "13-16-8",
         "--arch-mlp-top", "32-16-1", "--data-size", "128", "--epochs", "2"]


@pytest.fixture(scope="module", autouse=True)
def _built():
    build.build_host()


def test_bench_starts_its_own_ranks():
    """`bench.py --gpus 2` with no WORLD_SIZE: the parent spawns two ranks (env RANK / WORLD_SIZE / MASTER_*), they meet
    over 127.0.0.1 and rank 0's single JSON line is relayed on the parent's stdout."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["ranks_observed"] == 2 and d["sum"] == 2.0 and d["n_gpus"] == 2


def test_bench_parent_fails_when_a_rank_fails():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU: there the ranks refuse to run")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == "" and "failed" in r.stderr


def test_bench_rejects_mismatched_world_size():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def test_dlrm_binary_starts_its_own_ranks_dry():
    """`dlrm -ll:gpu 3`: three child ranks of the binary, rendezvous through the private directory (rank 0 publishes the
    id file atomically, the others wait for it); FFM_LAUNCH_DRYRUN stops before the first GPU call."""
    env = dict(os.environ, FFM_LAUNCH_DRYRUN="1")
    r = subprocess.run([EXE, "-ll:gpu", "3", *SMALL], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = sorted(l for l in r.stdout.splitlines() if l.startswith("[launcher]"))
    assert len(got) == 3
    sums = {l.split("checksum")[1] for l in got}
    assert len(sums) == 1                      # every rank read the id rank 0 wrote
    for k in range(3):
        assert f"rank {k} of 3" in got[k]


def test_dlrm_binary_forwards_sigterm_to_its_ranks():
    """Round-2 advisor finding: SIGTERM / SIGINT to the launcher alone used to orphan the ranks.  Three ranks that never end by
    themselves (dry run + FFM_LAUNCH_TEST_HANG); SIGTERM to the parent ends all of them, the parent returns non-zero and removes
    its rendezvous directory."""
    import signal, time, psutil, glob
    before = set(glob.glob("/tmp/ffm_launch_*"))
    env = dict(os.environ, FFM_LAUNCH_DRYRUN="1", FFM_LAUNCH_TEST_HANG="1")
    p = subprocess.Popen([EXE, "-ll:gpu", "3", *SMALL], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    deadline = time.time() + 30
    kids = []
    while time.time() < deadline:
        kids = psutil.Process(p.pid).children()
        if len(kids) == 3:
            break
        time.sleep(0.05)
    assert len(kids) == 3
    time.sleep(0.5)
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=30)
    assert p.returncode != 0
    gone, alive = psutil.wait_procs(kids, timeout=10)
    assert not alive, alive
    assert set(glob.glob("/tmp/ffm_launch_*")) <= before


def test_dlrm_binary_signal_before_the_first_fork_is_forwarded():
    """Round-3 advisor finding (3): the handlers are installed, with SIGTERM / SIGINT blocked, BEFORE the first fork.  The parent
    raises SIGTERM on itself there (test hook); it stays pending until the ranks exist, is then forwarded to them, and the parent
    returns non-zero and cleans up instead of dying and leaving three hanging ranks behind."""
    import glob
    before = set(glob.glob("/tmp/ffm_launch_*"))
    env = dict(os.environ, FFM_LAUNCH_DRYRUN="1", FFM_LAUNCH_TEST_HANG="1", FFM_LAUNCH_TEST_SIGNAL_SELF_EARLY="1", FFM_LAUNCH_GRACE_S="2.0")
    r = subprocess.run([EXE, "-ll:gpu", "3", *SMALL], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode not in (0, -15), (r.returncode, r.stderr)      # the parent survived its own SIGTERM and reported failure
    assert r.stderr.count("ended with 143") + r.stderr.count("ended with 137") == 3, r.stderr
    assert set(glob.glob("/tmp/ffm_launch_*")) <= before


def test_dlrm_binary_kills_a_rank_that_ignores_sigterm():
    """Round-3 advisor findings (1), (2): the grace timer starts wherever the signal lands (the flag is read on every turn of the
    wait loop), only ranks not yet reaped are signalled, and ranks that ignore SIGTERM are SIGKILLed after the grace period."""
    import signal, time, psutil
    env = dict(os.environ, FFM_LAUNCH_DRYRUN="1", FFM_LAUNCH_TEST_HANG="1", FFM_LAUNCH_TEST_IGNORE_TERM="1", FFM_LAUNCH_GRACE_S="1.0")
    p = subprocess.Popen([EXE, "-ll:gpu", "3", *SMALL], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    deadline = time.time() + 30
    kids = []
    while time.time() < deadline:
        kids = psutil.Process(p.pid).children()
        if len(kids) == 3:
            break
        time.sleep(0.05)
    assert len(kids) == 3
    time.sleep(1.0)                            # the ranks are past their exec and ignore SIGTERM by now
    t0 = time.time()
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=30)
    assert p.returncode != 0
    assert 0.9 < time.time() - t0 < 20
    assert err.count("ended with 137") == 3, err      # 128 + SIGKILL
    gone, alive = psutil.wait_procs(kids, timeout=10)
    assert not alive, alive


def test_dlrm_binary_reports_a_failing_rank():
    """A rank that cannot come up (no GPU here / no such device) makes the parent end the others and return non-zero."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    r = subprocess.run([EXE, "-ll:gpu", "2", *SMALL], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


def test_run_dlrm_two_ranks_gloo(oracle):
    """run_dlrm.py -ll:gpu 2 with the oracle as kernel library: two ranks over gloo run the driver's warm-up + epochs and
    rank 0 prints the reference's THROUGHPUT line once."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, RUN_DLRM, "-ll:gpu", "2", *SMALL, "--backend", oracle.ORACLE_LIB], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("THROUGHPUT = ") == 1
    assert "[DLRM] batchSize(64) workersPerNodes(2)" in r.stdout


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_dlrm_binary_launcher_one_rccl_rank_on_gpu(hip):
    """The C++ launcher end to end with one rank: child process, id file, ncclCommInitRank from C++, the exchange path
    (all-to-all each way + all-reduce) on the HIP streams, the launcher's all-reduce barrier, THROUGHPUT line."""
    env = dict(os.environ, FFM_FORCE_LAUNCHER="1")
    r = subprocess.run([EXE, "-ll:gpu", "1", "--force-exchange", *SMALL], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "THROUGHPUT = " in r.stdout
    plain = subprocess.run([EXE, *SMALL], capture_output=True, text=True, timeout=600)
    assert plain.returncode == 0, plain.stderr
    mse = lambda txt: [l for l in txt.splitlines() if "mean_squared_error" in l][-1].split("mean_squared_error:")[1].split()[0]
    assert abs(float(mse(r.stderr)) - float(mse(plain.stderr))) < 1e-5        # same model, same data: same loss through the exchange path


@pytest.mark.gpu
def test_bench_one_rccl_rank_direct_allreduce_on_gpu(hip):
    """--direct-allreduce through RCCL on one rank (round 6): the gradient sum as all-to-all of slices + ffh_sum_slices_f32 + all-gather on
    the buckets' own communicator -- functional (the calls are enqueued on the HIP streams and the step completes), the line says which
    algorithm ran, and with one rank the sum is the slice itself: the same throughput-independent facts as the ring run (collective counts)."""
    lines = {}
    for tag, extra in (("ring", []), ("direct", ["--shim-flags=--direct-allreduce"])):
        r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--force-exchange", "--per-gpu-batch", "4096", "--steps", "4", "--warmup", "2",
                            "--no-cpu-baseline", "--no-secondary", *extra], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        lines[tag] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    par = {k: v["config"]["gradient_allreduce"] for k, v in lines.items()}
    assert "DIRECT" in par["direct"] and "DIRECT" not in par["ring"], par
    assert "a communicator of their own" in par["direct"], par          # the launchers' default where ncclCommSplit exists
    assert lines["direct"]["config"]["collective_calls_rank0"]["alltoall"] == lines["ring"]["config"]["collective_calls_rank0"]["alltoall"]


@pytest.mark.gpu
def test_run_dlrm_one_rank_on_gpu(hip):
    r = subprocess.run([sys.executable, RUN_DLRM, "-ll:gpu", "1", *SMALL], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("THROUGHPUT = ") == 1


@pytest.mark.gpu
def test_bench_one_rank_exchange_line_on_gpu(hip):
    """bench.py through the exchange path on one RCCL rank: the line carries roofline, the collective counts and the rank count."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--force-exchange", "--workload", "kaggle", "--steps", "5", "--warmup", "2",
                        "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["ranks_observed"] == 1
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["achieved"] > 0
    c = d["config"]["collective_calls_rank0"]
    assert c["alltoall"] >= 2 * 7 and c["allreduce"] + c.get("allreduce_buckets", 0) >= 7       # (round 5: the MLP gradients travel in buckets issued from inside backward())
    assert "allreduce_exposed_us" in d["collectives_in_step_us"] and "allreduce_buckets_us" in d["collectives_in_step_us"]


# ---------------------------------------------------------------------------------------------------------------------
# bench.py's rank path end to end on the CPU: gloo for the collectives, the oracle as kernel library.  The line it prints is
# marked as not a measurement; what is checked is that the N > 1 code of bench.py (batch split, table ownership, the
# roofline probes on rank 0, barriers, MAX-reduced time, the single JSON line) runs and reports what it saw.
# ---------------------------------------------------------------------------------------------------------------------
def _bench_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FFM_TESTING"] = "1"          # bench.py honours --functional-test-backend only with it
    return env


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_bench_two_ranks_functional_on_cpu(oracle, scaling):
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--workload", "tiny", "--steps", "3", "--warmup", "1", "--scaling", scaling,
                        "--functional-test-backend", oracle.ORACLE_LIB], env=_bench_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["FUNCTIONAL_TEST_NOT_A_MEASUREMENT"] is True
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["steps"] == 3 and d["metric"] == "dlrm_training_samples_per_sec"
    assert d["config"]["ranks_observed"] == 2
    assert d["config"]["global_batch"] == (128 if scaling == "strong" else 256) and d["config"]["per_gpu_batch"] == d["config"]["global_batch"] // 2
    c = d["config"]["collective_calls_rank0"]
    assert c["alltoall"] == 2 * (1 + 1 + 3) and c["allreduce"] == 1 + 1 + 3        # the driver's warm-up + W + K steps
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["tables_in_launch"] == 4      # rank 0 owns tables 0, 2, 4, 6 of 8
    assert d["value"] > 0 and "cpu_baseline" not in d
    # round-3 advisor (high): the in-step probes run real steps, i.e. collectives -- every rank must walk them (rank 0 alone hung a
    # real N > 1 run).  The functional run takes that branch now and reports the collectives' in-step intervals.
    cis = d["collectives_in_step_us"]
    for who in ("rank0", "max_over_ranks"):
        assert set(cis[who]) == {"alltoall_fwd_us", "alltoall_bwd_us", "allreduce_us", "embedding_branch_wait_us", "gather_plus_alltoall_fwd_us", "alltoall_bwd_plus_table_update_us"}
        assert cis[who]["alltoall_fwd_us"] >= 0 and cis[who]["allreduce_us"] >= 0
    assert all(cis["max_over_ranks"][k] >= cis["rank0"][k] for k in cis["rank0"])
    pay = d["config"]["collective_payload_bytes_per_step_rank0"]
    B, D = d["config"]["global_batch"], 16
    assert pay["alltoall_forward_sent"] == pay["alltoall_backward_sent"] == 4 * B * D * 4 // 2 and pay["allreduce_buffer"] > 0


def test_bench_failing_rank_is_named_in_the_parents_output(oracle):
    """First-contact readiness for N > 1: when a rank other than 0 dies, the launcher's own stderr carries that rank's output with a
    [rank r] tag, names the rank that failed first, ends the others and exits non-zero without a JSON line."""
    env = dict(_bench_env(), FFM_TEST_FAIL_RANK="1", FFM_SPAWN_TIMEOUT="120")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--workload", "tiny", "--steps", "2", "--warmup", "1",
                        "--functional-test-backend", oracle.ORACLE_LIB], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "[rank 1] rank 1: failing on purpose" in r.stderr and "rank(s) [1] failed first" in r.stderr


def test_bench_as_a_rank_under_torchrun_functional_on_cpu(oracle):
    """the driver's launch line: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N"""
    with __import__("socket").socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--workload", "tiny", "--steps", "2", "--warmup", "1",
                        "--functional-test-backend", oracle.ORACLE_LIB], env=_bench_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_observed"] == 2 and d["steps"] == 2


def test_bench_two_ranks_with_data_parallel_tables_functional_on_cpu(oracle):
    """--replicate-embedding-rows: all eight 1000-row tables of the tiny workload become data-parallel -- no all-to-all is
    issued at all, the table gradients ride in the one all-reduce per step, and the line says which layout ran."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--workload", "tiny", "--steps", "3", "--warmup", "1", "--replicate-embedding-rows", "1000",
                        "--functional-test-backend", oracle.ORACLE_LIB], env=_bench_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["ranks_observed"] == 2
    assert "data-parallel" in d["config"]["parallelism"]
    c = d["config"]["collective_calls_rank0"]
    assert c["alltoall"] == 0 and c["allreduce"] == 1 + 1 + 3
    assert d["value"] > 0
