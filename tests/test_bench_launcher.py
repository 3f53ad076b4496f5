"""`python bench.py --gpus N` launches its own ranks (no torchrun needed): the parent's env plumbing and exit-code handling, on CPU.

The reference trains on one device (`net.cuda()`, /root/reference/Train_SMT.py:160-161); the launcher is the entry of the
data-parallel path this build adds.  The children here are tiny stand-in scripts, not bench.py itself (which needs a GPU).
"""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_rank_env_has_what_torchrun_would_set():
    env = bench.rank_env({"PATH": "/bin"}, 3, 8, 29555)
    assert env["RANK"] == "3" and env["LOCAL_RANK"] == "3" and env["WORLD_SIZE"] == "8" and env["LOCAL_WORLD_SIZE"] == "8"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29555"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/bin"
    assert bench.rank_env({"HSA_ENABLE_IPC_MODE_LEGACY": "1"}, 0, 2, 1)["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"   # never overridden


def _run_launcher(tmp_path, child_src, world, argv=("--x", "1")):
    child = tmp_path / "child.py"
    child.write_text(textwrap.dedent(child_src))
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        raise SystemExit(bench.launch_ranks({world}, {list(argv)!r}, script={str(child)!r}, poll_s=0.05))
    """))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    return subprocess.run([sys.executable, str(driver)], capture_output=True, text=True, timeout=120, env=env)


def test_launcher_passes_rank0_line_through(tmp_path):
    r = _run_launcher(tmp_path, """
        import json, os, sys
        rank = int(os.environ["RANK"])
        rec = {k: os.environ[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        rec["argv"] = sys.argv[1:]
        if rank == 0:
            print(json.dumps(rec), flush=True)
        else:
            print(json.dumps(rec), file=sys.stderr, flush=True)
    """, world=3)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1                                   # exactly ONE line on stdout: rank 0's
    rec = json.loads(lines[0])
    assert rec["RANK"] == "0" and rec["WORLD_SIZE"] == "3" and rec["MASTER_ADDR"] == "127.0.0.1" and rec["argv"] == ["--x", "1"]
    others = [json.loads(l) for l in r.stderr.splitlines() if l.startswith("{")]
    assert sorted(o["RANK"] for o in others) == ["1", "2"]
    assert {o["MASTER_PORT"] for o in others} == {rec["MASTER_PORT"]}


def test_launcher_fails_when_a_rank_fails(tmp_path):
    r = _run_launcher(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(60)          # the surviving rank would wait in a collective: the launcher must end it
    """, world=2)
    assert r.returncode == 7
    assert "rank 1 exited with code 7" in r.stderr


def test_bench_refuses_mismatched_world():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr


def test_bench_fails_fast_without_enough_gpus_for_rccl():
    """One rank per GPU over RCCL: a node with fewer devices than ranks is refused with a clear message before any process group
    exists (no device at all in this container)."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "needs 2 GPUs on this node, found 0" in r.stderr


def test_sigterm_to_the_launcher_ends_every_rank(tmp_path):
    """a `timeout` / scheduler SIGTERM on `python bench.py --gpus N` must not orphan ranks that sit in a collective
    holding GPUs: the launcher ends each rank's process group (grandchildren included) and exits 128 + SIGTERM."""
    import signal
    import time
    child = tmp_path / "child.py"
    child.write_text(textwrap.dedent(f"""
        import os, subprocess, sys, time
        helper = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(300)"])      # a rank's own helper process
        open(os.path.join({str(tmp_path)!r}, "pids.%s" % os.environ["RANK"]), "w").write("%d %d" % (os.getpid(), helper.pid))
        time.sleep(300)
    """))
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        raise SystemExit(bench.launch_ranks(2, [], script={str(child)!r}, poll_s=0.05))
    """))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.Popen([sys.executable, str(driver)], env=env, stderr=subprocess.PIPE, text=True)
    deadline = time.time() + 60
    while time.time() < deadline and not all((tmp_path / f"pids.{r}").exists() and (tmp_path / f"pids.{r}").read_text().count(" ") for r in (0, 1)):
        time.sleep(0.05)
    pids = [int(x) for r in (0, 1) for x in (tmp_path / f"pids.{r}").read_text().split()]
    assert len(pids) == 4
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=60) == 128 + signal.SIGTERM
    assert "ending all ranks" in p.stderr.read()

    def alive(pid):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return False
        try:                                   # a zombie reparented to init still answers kill(0)
            return open(f"/proc/{pid}/stat").read().split(")")[-1].split()[0] != "Z"
        except FileNotFoundError:
            return False
    deadline = time.time() + 10
    while time.time() < deadline and any(alive(x) for x in pids):
        time.sleep(0.1)
    assert not any(alive(x) for x in pids)


def test_a_rank_that_exits_early_does_not_leave_its_helper_behind(tmp_path):
    """the failing rank has ALREADY exited when the launcher tears the job down -- its helper processes (same session)
    must be ended too, or they keep holding the GPU."""
    import time
    child = tmp_path / "child.py"
    child.write_text(textwrap.dedent(f"""
        import os, subprocess, sys, time
        helper = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(300)"])
        open(os.path.join({str(tmp_path)!r}, "pids.%s" % os.environ["RANK"]), "w").write("%d %d" % (os.getpid(), helper.pid))
        if os.environ["RANK"] == "1":
            sys.exit(9)              # leaves its helper running
        time.sleep(300)
    """))
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        raise SystemExit(bench.launch_ranks(2, [], script={str(child)!r}, poll_s=0.05))
    """))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(driver)], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 9
    pids = [int(x) for rk in (0, 1) for x in (tmp_path / f"pids.{rk}").read_text().split()]
    assert len(pids) == 4

    def alive(pid):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return False
        try:
            return open(f"/proc/{pid}/stat").read().split(")")[-1].split()[0] != "Z"
        except FileNotFoundError:
            return False
    deadline = time.time() + 10
    while time.time() < deadline and any(alive(x) for x in pids):
        time.sleep(0.1)
    assert not any(alive(x) for x in pids)
