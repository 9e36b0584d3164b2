"""`python bench.py --gpus N` started plainly (no WORLD_SIZE) launches its ranks itself: torch.distributed.run as a CHILD process,
before torch or the GPU is touched, rank 0's JSON line relayed, the child's exit code returned (VERDICT r2 missing 2)."""
import importlib.util
import json
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_parent_spawns_the_ranks_and_relays_the_line(monkeypatch, capsys):
    b = _bench()
    seen = {}
    line = json.dumps({"metric": "Mcells/sec (FluidSolver3D step)", "value": 1.0, "n_gpus": 4})

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7, stdout="W0101 some launcher chatter\n" + line + "\n")

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    rc = b.self_launch(4)
    out = capsys.readouterr()
    assert rc == 7                                                   # the child's exit code is the bench's
    assert out.out.strip() == line                                   # exactly one JSON line on stdout
    assert "launcher chatter" in out.err
    c = seen["cmd"]
    assert c[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node" in c and c[c.index("--nproc-per-node") + 1] == "4"
    assert c[c.index("--master-addr") + 1] == "127.0.0.1" and os.path.basename(c[c.index("--master-port") + 2]) == "bench.py"
    assert c[-6:] == ["--gpus", "4", "--steps", "5", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_main_takes_the_launch_path_before_importing_torch(monkeypatch):
    b = _bench()
    called = []
    monkeypatch.setattr(b, "self_launch", lambda n: called.append(n) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    had_torch = "torch" in sys.modules
    try:
        b.main()
    except SystemExit as e:
        assert e.code == 0
    assert called == [2]
    assert ("torch" in sys.modules) == had_torch                     # nothing imported torch on the way


def test_under_the_launcher_main_does_not_relaunch(monkeypatch):
    b = _bench()
    monkeypatch.setattr(b, "self_launch", lambda n: (_ for _ in ()).throw(AssertionError("relaunched")))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("LOCAL_RANK", "0")
    try:
        b.main()
    except SystemExit as e:                                          # no GPU here: the rank path stops at "needs a GPU"
        assert "GPU" in str(e.code)
