"""`python bench.py --gpus N` without a launcher around it starts its own ranks (bench.py::launch_ranks): N fresh child
processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rank 0's stdout passed through, worst exit code returned,
the whole group stopped when one rank fails or the wall-clock limit passes.  CPU tests with stand-in rank programs (the
real ranks need a GPU; the one-GPU rehearsal of the real thing is tools/bench_rehearsal_stub.sh)."""
import io
import json
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bench():
    import importlib
    import bench
    return importlib.reload(bench)


def _script(tmp_path, body):
    p = tmp_path / "rank.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    # a zombie still answers kill(0): look at its state
    try:
        with open(f"/proc/{pid}/stat") as f:
            return f.read().rsplit(")", 1)[1].split()[0] != "Z"
    except FileNotFoundError:
        return False


def test_child_command_is_this_script_with_the_same_arguments():
    b = _bench()
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5", "--quick", "--no-train"]
    cmd = b.child_argv(argv)
    assert cmd[0] == sys.executable and os.path.samefile(cmd[1], os.path.join(ROOT, "bench.py")) and cmd[2:] == argv


def test_ranks_get_their_environment_and_rank0_owns_stdout(tmp_path):
    b = _bench()
    prog = _script(tmp_path, """
        import json, os, sys
        keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")
        print(json.dumps({"argv": sys.argv[1:], **{k: os.environ.get(k) for k in keys}}), flush=True)
        print("note from rank " + os.environ["RANK"], file=sys.stderr, flush=True)
    """)
    out, err = io.StringIO(), io.StringIO()
    rc = b.launch_ranks([sys.executable, prog, "--gpus", "3", "--steps", "7"], 3, timeout_s=60, out=out, err=err)
    assert rc == 0
    lines = [json.loads(x) for x in out.getvalue().splitlines()]
    assert len(lines) == 1                                    # rank 0's line alone reaches stdout
    assert lines[0]["RANK"] == "0" and lines[0]["WORLD_SIZE"] == "3" and lines[0]["MASTER_ADDR"] == "127.0.0.1"
    assert lines[0]["argv"] == ["--gpus", "3", "--steps", "7"] and int(lines[0]["MASTER_PORT"]) > 0
    others = [json.loads(x.split("] ", 1)[1]) for x in err.getvalue().splitlines() if x.startswith("[rank") and "{" in x]
    assert sorted((o["RANK"], o["LOCAL_RANK"]) for o in others) == [("1", "1"), ("2", "2")]
    assert {o["MASTER_PORT"] for o in others} == {lines[0]["MASTER_PORT"]}
    assert "note from rank 0" in err.getvalue() and "note from rank 2" in err.getvalue()


def test_a_failing_rank_stops_the_job_and_leaves_no_orphan(tmp_path):
    b = _bench()
    pids = tmp_path / "pids"
    pids.mkdir()
    prog = _script(tmp_path, f"""
        import os, subprocess, sys, time
        r = os.environ["RANK"]
        open(os.path.join({str(pids)!r}, r), "w").write(str(os.getpid()))
        if r == "1":
            time.sleep(0.5)
            sys.exit(3)
        # a rank that would hang in a collective for ever, with a grandchild of its own
        g = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"])
        open(os.path.join({str(pids)!r}, r + "g"), "w").write(str(g.pid))
        time.sleep(600)
    """)
    err = io.StringIO()
    t0 = time.monotonic()
    rc = b.launch_ranks([sys.executable, prog], 3, timeout_s=120, out=io.StringIO(), err=err)
    assert rc == 3 and time.monotonic() - t0 < 60
    assert "rank 1 exited with 3" in err.getvalue()
    time.sleep(0.2)
    for name in os.listdir(pids):
        assert not _alive(int((pids / name).read_text())), name       # ranks AND their own children are gone


def test_wall_clock_limit(tmp_path):
    b = _bench()
    prog = _script(tmp_path, "import time; time.sleep(600)")
    t0 = time.monotonic()
    rc = b.launch_ranks([sys.executable, prog], 2, timeout_s=1.0, out=io.StringIO(), err=io.StringIO())
    assert rc == 124 and time.monotonic() - t0 < 30


def test_bench_main_launches_itself_without_a_launcher(tmp_path):
    """`python bench.py --gpus 2 ...` with no WORLD_SIZE: the parent never imports torch; its ranks are the same command.
    Here the ranks fail at once (no GPU in this container) -- what is checked is that the parent started two of them,
    handed each its rank, and returned their failure as its own exit code instead of the old usage message."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["BENCH_LAUNCH_TIMEOUT"] = "300"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--quick", "--no-train", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                         timeout=600)
    assert res.returncode != 0
    assert "launch N>1 with" not in res.stderr + res.stdout
    assert "exited with" in res.stderr and "stopping the other ranks" in res.stderr
