"""bench.py starts its own ranks: `python bench.py --gpus N` outside a launcher re-executes itself under
torch.distributed.run, one process per GPU, on a 127.0.0.1 rendezvous.  CPU test: the command
line it builds and the condition under which it is taken -- the ranks themselves need a GPU."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_gpus_flag_starts_the_ranks(monkeypatch):
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    import subprocess
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--workload", "cfg5", "--batch", "8", "--steps", "3"])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "4", "--workload", "cfg5", "--batch", "8", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_inside_a_launcher_nothing_is_started(monkeypatch):
    """WORLD_SIZE in the environment (the driver's own torch.distributed.run): no second level of ranks."""
    import subprocess
    monkeypatch.setattr(subprocess, "call", lambda *a, **k: pytest.fail("relaunched under a launcher"))
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    import torch
    if torch.cuda.is_available():
        pytest.skip("would run the benchmark")
    with pytest.raises(SystemExit) as ex:      # no GPU here: main() gets as far as the device check
        bench.main()
    assert "needs a GPU" in str(ex.value.code)
