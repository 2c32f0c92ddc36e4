"""CPU, world_size 2, gloo: the clip sharding + max-over-ranks timing logic of the multi-GPU path."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fcvsr_amd.harness.sharding import shard, shard_sequences
    from fcvsr_amd.harness.windows import window_indices
    seqs = [100, 100, 41, 34]                     # REDS4-like and Vid4-like lengths
    mine = shard_sequences(seqs, rank, world)
    # every rank processes its frames (here: records which LR frames each window needs) with no communication
    frames = [(s, i, tuple(window_indices(i, 7, seqs[s], "replicate"))) for s, a, b in mine for i in range(a, b)]
    n = torch.tensor([len(frames)], dtype=torch.int64)
    t = torch.tensor([0.5 + 0.25 * rank], dtype=torch.float64)      # pretend timings
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                        # bench.py: max over ranks
    dist.barrier()
    q.put((rank, frames, [int(c) for c in counts], float(t)))
    dist.destroy_process_group()


def test_two_rank_sharding_covers_every_frame_once():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = sorted((s, i) for _, frames, _, _ in res for s, i, _ in frames)
    assert seen == [(s, i) for s, n in enumerate([100, 100, 41, 34]) for i in range(n)]
    for _, _, counts, tmax in res:
        assert sum(counts) == 275 and abs(counts[0] - counts[1]) <= 1
        assert tmax == 0.75
    from fcvsr_amd.harness.sharding import shard, throughput
    assert [shard(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert throughput([138, 137], [0.5, 0.75]) == 275 / 0.75


def test_bench_self_launches_its_ranks_with_gloo():
    """`python bench.py --gpus 2` outside torch.distributed.run must start its own rank processes (before any GPU call),
    relay rank 0's JSON line and exit 0; --stub replaces the GPU step by a no-op so the plumbing runs without a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--stub",
                        "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                        # exactly one JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["data"] == "stub"
    # the same entry point under the driver's launcher form
    port = _free_port()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--backend",
                        "gloo", "--stub", "--steps", "2", "--warmup", "0"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert sum(l.startswith("{") for l in p.stdout.splitlines()) == 1


def test_bench_rejects_world_size_mismatch():
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0


def test_bench_self_launch_stops_all_ranks_when_one_dies():
    """A rank that exits early (bad device ordinal, missing build, ...) must not leave the others blocked in the rendezvous:
    the launcher stops them and returns the failing rank's code promptly."""
    import subprocess
    import sys
    import time
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FCVSR_BENCH_FAIL_RANK"] = "1"
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--stub",
                        "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode != 0
    assert time.time() - t0 < 120, "the surviving rank was left waiting"
    assert "rank 1 exited with code 3" in p.stderr


def _bench_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def test_bench_launcher_parent_never_imports_torch(tmp_path):
    """The `--gpus N` launcher parent (no RANK in its environment) must not import torch at all - torch.cuda.device_count() can
    fall back to a HIP runtime call, and a process that has initialised the GPU must not fork + exec on this pool.  A
    sitecustomize on PYTHONPATH records any torch import made by a process WITHOUT `RANK` and makes device_count() raise there."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    marker = tmp_path / "parent_imported_torch"
    (tmp_path / "sitecustomize.py").write_text(
        "import os, sys\n"
        "if 'RANK' not in os.environ:\n"
        "    class _F:\n"
        "        def find_spec(self, name, path=None, target=None):\n"
        "            if name == 'torch' or name.startswith('torch.'):\n"
        f"                open({str(marker)!r}, 'a').write(name + '\\n')\n"
        "                raise ImportError('launcher parent imported ' + name)\n"
        "            return None\n"
        "    sys.meta_path.insert(0, _F())\n")
    env = dict(_bench_env(), PYTHONPATH=str(tmp_path) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    # a fake KFD topology with 2 GPU nodes and one CPU node: the parent counts devices from it (no *_VISIBLE_DEVICES set)
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        env.pop(k, None)
    topo = tmp_path / "nodes"
    for i, simd in enumerate((0, 1024, 1024)):
        (topo / str(i)).mkdir(parents=True)
        (topo / str(i) / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\n")
    env["FCVSR_KFD_TOPOLOGY"] = str(topo)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--stub",
                        "--steps", "2", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert not marker.exists(), marker.read_text()
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
    # 3 ranks on a 2-GPU topology: refused by the parent, still without torch
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--backend", "gloo", "--stub"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "only 2 HIP device" in p.stderr
    assert not marker.exists()


def test_bench_visible_gpu_count_sources(tmp_path, monkeypatch):
    import importlib.util
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("FCVSR_KFD_TOPOLOGY", str(tmp_path / "absent"))
    assert bench.visible_gpu_count() is None                 # no driver: the children validate their ordinals
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,3,5")
    assert bench.visible_gpu_count() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_count() == 0


def test_bench_train_block_failure_is_collective():
    """One rank whose LOCAL half of a training step raises must not leave the others inside the gradient all-reduce: the ranks
    agree on a MIN-reduced ok flag before every collective, all of them skip the rest, rank 0 still prints the headline line
    (with the error in `train`) and every rank exits 0 promptly (bench.py train block)."""
    import json
    import subprocess
    import sys
    import time
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    for fail in ("1", "0"):
        env = dict(_bench_env(), FCVSR_BENCH_TRAIN_FAIL_RANK=fail)
        t0 = time.time()
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--stub",
                            "--steps", "2", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        assert time.time() - t0 < 120
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        rec = json.loads(lines[0])
        assert "error" in rec["train"] and rec["train"]["steps_done"] == 0
    # and without the hook the protocol completes its 3 timed steps
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--stub",
                        "--steps", "2", "--warmup", "0"], env=_bench_env(), capture_output=True, text=True, timeout=300)
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert rec["train"] == {"steps_done": 3, "world": 2}
