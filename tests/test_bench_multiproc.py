"""CPU, world_size 2, gloo: the only N > 1 logic of this path.  The GEMM path does not shard (replicas
only, no data-path collective); bench.py --gpus N runs N independent replicas and uses torch.distributed just for the
barriers and the max-over-ranks of the timed region.  This test runs that reduction over gloo in two processes."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_max_over_ranks_and_aggregate_with_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        import torch.distributed as dist
        import bench
        dist.init_process_group("gloo")
        rank = dist.get_rank()
        dist.barrier()
        wall = bench.max_over_ranks(0.010 * (rank + 1))          # rank 1 is the slow replica
        val = bench.aggregate_value(2.0 * 4096 ** 3, 10, dist.get_world_size(), wall)
        print(f"RESULT rank={{rank}} wall={{wall:.6f}} value={{val:.3f}}", flush=True)
        dist.barrier()
        dist.destroy_process_group()
    """))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    want_wall = 0.020
    want_val = 2.0 * 4096 ** 3 * 10 * 2 / want_wall / 1e12
    for out in outs:
        line = [l for l in out.splitlines() if l.startswith("RESULT")][0]
        kv = dict(t.split("=") for t in line.split()[1:])
        assert abs(float(kv["wall"]) - want_wall) < 1e-9 and abs(float(kv["value"]) - want_val) < 1e-3, line


def test_single_process_is_identity():
    sys.path.insert(0, ROOT)
    import bench

    assert bench.max_over_ranks(0.5) == 0.5
    assert abs(bench.aggregate_value(2.0 * 4096 ** 3, 200, 1, 200 * 40e-6) - 2.0 * 4096 ** 3 / 40e-6 / 1e12) < 1e-6
