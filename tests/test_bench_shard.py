"""bench.py's multi-GPU launcher and the one global pair list, on CPU ranks (gloo): `bench.py --gpus N` must start N ranks
itself, refuse a WORLD_SIZE that differs from --gpus, and register the very same pairs whatever N is.  The per-rank compute
is the CPU oracle here (tests/bench_cpu_shim.py); on the GPU box it is libmulls_hip.so and nothing else."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "bench_cpu_shim.py")


def _run(cmd, env=None, timeout=280):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    return p


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_flag_spawns_the_ranks_and_results_do_not_depend_on_n(tmp_path):
    t1, t2 = str(tmp_path / "t1.npy"), str(tmp_path / "t2.npy")
    common = ["--tiny", "--total-pairs", "9", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-end-to-end"]
    p1 = _run([sys.executable, SHIM, "--gpus", "1", "--dump-table", t1] + common)
    assert p1.returncode == 0, p1.stderr[-2000:]
    p2 = _run([sys.executable, SHIM, "--gpus", "2", "--dump-table", t2] + common)  # no WORLD_SIZE in the environment: bench.py launches 2 ranks
    assert p2.returncode == 0, p2.stderr[-2000:]
    j1, j2 = _json_line(p1.stdout), _json_line(p2.stdout)
    assert j1["n_gpus"] == 1 and j2["n_gpus"] == 2
    assert j1["scaling"] == j2["scaling"] == "strong"
    assert j2["config"]["pairs_per_step"] == 9 and j2["config"]["pairs_per_gpu_per_step"] == 4  # block_partition(9, 2, 0)
    a, b = np.load(t1), np.load(t2)
    assert a.shape == (9, 56) and np.array_equal(a, b)  # bit-identical T, information matrix, code, iterations, sigma
    assert j1["config"]["result_table_sha256"] == j2["config"]["result_table_sha256"]


def test_driver_invocation_shape_and_world_size_check():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(port), SHIM]
    common = ["--tiny", "--pairs", "3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    ok = _run(base + ["--gpus", "2"] + common)
    assert ok.returncode == 0, ok.stderr[-2000:]
    j = _json_line(ok.stdout)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["registrations_timed"] == 2 * 3 * 2 and j["steps"] == 2 and j["warmup"] == 1
    bad = _run(base + ["--gpus", "4"] + common)  # 2 ranks started, 4 claimed
    assert bad.returncode != 0
    assert "WORLD_SIZE" in bad.stderr


def test_global_pair_list_is_rank_independent():
    sys.path.insert(0, ROOT)
    import bench

    scenes = bench.build_scenes(4, True, 1)
    a = bench.global_pair(scenes, 13)
    b = bench.global_pair(scenes, 13)
    assert np.array_equal(a.init_guess, b.init_guess) and a.tgt[0] is scenes[13 % 4][0].tgt[0]
    assert not np.array_equal(bench.global_pair(scenes, 17).init_guess, a.init_guess)  # same scene, another guess
    args = bench.parse_args(["--total-pairs", "1024", "--gpus", "8"])
    spans = [bench.rank_span(args, 8, r) for r in range(8)]
    assert spans[0] == (0, 128) and spans[7] == (896, 1024)
    args = bench.parse_args(["--pairs", "4096", "--gpus", "2"])
    assert bench.rank_span(args, 2, 1) == (4096, 8192)


def test_config_flag_selects_every_baseline_configuration():
    """bench.py --config k: 0 / 3 are aliases of --data demo / --total-pairs 1024, 2 / 4 the large-cloud lines with the headline line's JSON shape
    (tiny plumbing sizes here, the CPU oracle as the engine; two gloo ranks for configs[2])."""
    sys.path.insert(0, ROOT)
    import bench

    assert bench.parse_args([]).config == 1 and bench.parse_args([]).total_pairs == 0  # the driver's default line is configs[1]
    assert bench.parse_args(["--config", "0"]).data == "demo"
    assert bench.parse_args(["--config", "3"]).total_pairs == 1024 and bench.parse_args(["--config", "3", "--total-pairs", "64"]).total_pairs == 64
    common = ["--tiny", "--steps", "1", "--warmup", "0"]
    p4 = _run([sys.executable, SHIM, "--config", "4"] + common)
    assert p4.returncode == 0, p4.stderr[-2000:]
    j4 = _json_line(p4.stdout)
    assert j4["config"]["workload"].startswith("configs[4]") and j4["unit"] == "registrations/s" and j4["scaling"] == "weak" and j4["dtype"] == "f32"
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(j4["roofline"]) and j4["roofline"]["whole_path"]["B_reg_bytes_per_registration"] > 0
    assert j4["cpu_baseline"]["kind"] == "port" and j4["delta_T_vs_ref"]["integer_outputs_equal"]
    p2 = _run([sys.executable, SHIM, "--config", "2", "--gpus", "2", "--no-cpu-baseline"] + common)
    assert p2.returncode == 0, p2.stderr[-2000:]
    j2 = _json_line(p2.stdout)
    assert j2["n_gpus"] == 2 and j2["config"]["workload"].startswith("configs[2]") and j2["config"]["pairs_per_step"] == 2 * j2["config"]["pairs_per_gpu_per_step"]
