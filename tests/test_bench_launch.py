"""bench.py --gpus N without a launcher around it: the command it builds to start one rank per GPU, and the launcher
path itself on CPU with a stand-in for the ranks (two gloo ranks that agree on a sum and print one JSON line)."""
import json
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launch_command():
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "4", "--warmup", "1"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    at = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[at + 1:] == ["--gpus", "8", "--steps", "4", "--warmup", "1"]


def test_launcher_starts_ranks_and_relays_one_line(tmp_path):
    stub = tmp_path / "ranks.py"
    stub.write_text(textwrap.dedent("""
        import json, os, sys
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.tensor([float(dist.get_rank() + 1)])
        dist.all_reduce(t)
        print("noise from rank %d" % dist.get_rank())
        if dist.get_rank() == 0:
            print(json.dumps({"n_gpus": dist.get_world_size(), "sum": float(t[0]), "argv": sys.argv[1:],
                              "ipc": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}))
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(3 if "--fail" in sys.argv else 0)
    """))
    driver = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import bench
        sys.exit(bench.launch_ranks(2, sys.argv[2:], script=sys.argv[1]))
    """ % ROOT)
    res = subprocess.run([sys.executable, "-c", driver, str(stub), "--gpus", "2"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    lines = res.stdout.splitlines()
    assert len(lines) == 1, res.stdout                       # rank 0's line and nothing else
    out = json.loads(lines[0])
    assert out == {"n_gpus": 2, "sum": 3.0, "argv": ["--gpus", "2"], "ipc": "0"}
    # a failing rank is the launcher's exit status
    res = subprocess.run([sys.executable, "-c", driver, str(stub), "--fail"], capture_output=True, text=True, timeout=300)
    assert res.returncode != 0


def test_bench_without_launcher_would_launch(monkeypatch):
    # `python bench.py --gpus 2` with no RANK in the environment takes the launcher path before torch is imported
    called = {}
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench, "launch_ranks", lambda n, argv, script=None: called.update(n=n, argv=list(argv)) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert called == {"n": 2, "argv": ["--gpus", "2", "--steps", "3"]}
