#!/usr/bin/env python
"""bench.py — MULLS-ICP hot path throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs B | --total-pairs T]

With --gpus N > 1 and no torch.distributed environment, bench.py starts its N ranks itself (python -m
torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...); started under torch.distributed.run
by somebody else it checks WORLD_SIZE == N.  One rank per GPU, rank r on device LOCAL_RANK.

Metric (BASELINE.json): scan-pair registrations / second on synthetic 64-beam ~120k-point scans, 20 ICP iterations;
dT vs ref.  Workload = configs[1] (KITTI-like scan-to-scan: classes ground+pillar+facade, source 800/400/1200
fixed-number down-sampled, target = the previous frame's un-down-sampled features, 5-15 k points, sizes drawn per
scene).  There is ONE seeded global pair list: pair g = scene (g mod 64) with the g-th initial guess.
  default            weak scaling: rank r registers pairs [r*B, (r+1)*B) of the list, B = --pairs per GPU and step
  --total-pairs T    strong scaling = configs[3]: the T first pairs of the list, block-partitioned over the ranks
                     (mulls_amd/shard.py::block_partition), so N = 1 and N = 8 register the very same pairs; rank 0
                     gathers every rank's result table and reports its SHA-256 — equal for every N.
One "step" = one lock-step registration of the rank's pairs, clouds already staged in HBM (mulls_batch_create), each
step re-cloning them like cloudblock_t::clone_feature does.  value = registrations of all ranks / T, T = max over
ranks of the barrier-bracketed time of the K steps.

The JSON line also carries
  roofline      — the dominant kernel: algorithmic HBM bytes per launch / average launch duration (hipEvents on the
                  library's own stream, live in the timed region) against the 8 TB/s HBM peak; traffic and the VALU
                  issue view from the committed rocprofv3 --pmc passes of this very configuration.
  cpu_baseline  — the CPU oracle (restatement of the reference, kd-tree NN, the reference's 3-wide OpenMP sections)
                  timed on this box's host cores on a bounded sample of the same workload; cpu_baseline_manycore: 8
                  oracle processes side by side.
  delta_T_vs_ref — pairs of the timed workload against the oracle (checker): max |dt|, max rotation geodesic, integer
                  outputs equal.
  value_end_to_end — the same registrations from host buffers (mulls_icp_batch: staging upload included).
  value_converging — the same pairs with the call site's convergence thresholds (test/mulls_slam.cpp:642-648: converge_tran 0.0005 m,
                  converge_rot_d 0.001 deg) instead of the metric's forced 20 iterations: registrations/s and the mean iteration count.
  other_configs  — (default one-GPU line only) compact legs of every other BASELINE configuration on the same box, same process: configs[0] (the reference's demo
                  scans), configs[2] (scans against a ~1 M-point map), the 128-pair shard configs[3] gives each of 8 GPUs, configs[4] (128-beam dense pairs, six
                  classes, 40 iterations) and the reference's own scan-to-map regime (64 scans against 20 000-point local maps) — value, ms per step, the search's
                  roofline fraction, delta T against the checker.  `python bench.py --config k` is the full-length line of each.
  value_sustained — the timed configuration again over a fixed wall budget (>= 2 s of back-to-back steps), so that a sampler outside
                  the process sees the device busy; value stays the K-step figure the contract defines.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mulls_amd import abi, shard, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
VALU_PEAK_TLOPS = 78.6     # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz, non-FMA lane-ops/s
OPS_PER_EVAL = 9.3         # 3 sub + 3 mul + 2 add + ~1.3 min/compare/select per source-target distance evaluation
N_SCENES = 64              # distinct synthetic scenes of the global pair list
SEED0 = 20260924


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=4096, help="weak scaling: scan pairs per GPU per step (one lock-step batch: one launch set per ICP iteration)")
    ap.add_argument("--total-pairs", type=int, default=0, help="strong scaling (configs[3]): this many pairs of the global list, block-partitioned over the GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--nn-mode", type=int, default=0, help="0 auto (grid staged in LDS), 1 brute force, 2 grid in global memory, 3 grid in LDS")
    ap.add_argument("--tiny", action="store_true", help="plumbing-test sizes (12-beam scans, 4 scenes, 3 iterations): never a bench line")
    ap.add_argument("--dump-table", default="", help="rank 0 writes the gathered result table (npy) here")
    ap.add_argument("--data", default="synthetic", choices=["synthetic", "demo"],
                    help="demo: BASELINE configs[0] on real data — the reference's demo pair and its consecutive frames from tests/golden/demo_pair.npz (class clouds by "
                         "the reference's own extract_semantic_pts lines), test/mulls_reg.cpp:194-195's mm_lls_icp arguments; delta_T against the reference lines' results")
    ap.add_argument("--sustain-s", type=float, default=2.0, help="wall budget of the value_sustained leg (0: skip)")
    ap.add_argument("--no-converging", action="store_true", help="skip the value_converging leg")
    ap.add_argument("--config", type=int, default=1, choices=[0, 1, 2, 3, 4],
                    help="BASELINE.json configs[k]: 1 (default, the headline line) KITTI-like scan-to-scan; 0 the reference's demo pair (= --data demo); 3 the 1024-pair list "
                         "(= --total-pairs 1024); 2 scan-to-local-map against a ~1 M-point map; 4 128-beam ~240 k-point scans, six classes, 40 iterations")
    ap.add_argument("--large-pairs", type=int, default=0, help="configs 2 / 4: pairs per GPU and step (default 32 / 16)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the compact legs of the other BASELINE configurations (configs[0], [2], the 128-pair shard of [3], [4] and the 20 000-point scan-to-map batch) "
                         "that the default one-GPU line carries as `other_configs`")
    a = ap.parse_args(argv)
    if a.config == 0:
        a.data = "demo"
    if a.config == 3 and not a.total_pairs:
        a.total_pairs = 1024
    return a


# ------------------------------------------------------------------------------------------------------------------
# the global pair list
def converging_params(tiny=False):
    """The call site's own arguments (test/mulls_slam.cpp:642-648 with lo_gflag_list_kitti_urban.txt:59-60): the loop stops when a step is
    below 0.0005 m and 0.001 deg (never before the fourth iteration, cregistration.hpp:1357)."""
    if tiny:
        return abi.kitti_params(dis_thre_unit=2.4, max_iter_num=6)
    return abi.kitti_params()


def bench_params(tiny=False):
    # test/mulls_slam.cpp:642-648 with script/config/lo_gflag_list_kitti_urban.txt values; convergence thresholds at 0 so
    # that every registration executes exactly the 20 iterations the metric is quoted on
    if tiny:
        return abi.kitti_params(dis_thre_unit=2.4, max_iter_num=3, converge_translation=0.0, converge_rotation_d=0.0)
    return abi.kitti_params(converge_translation=0.0, converge_rotation_d=0.0)


def _scene_sizes(k, tiny):
    """Target class-cloud sizes of scene k: KITTI scan-to-scan targets hold 5-15 k feature points."""
    rng = np.random.default_rng(SEED0 + 31 * k)
    if tiny:
        return ({abi.GROUND: 200, abi.PILLAR: 100, abi.FACADE: 250, abi.BEAM: 0, abi.ROOF: 0},
                {abi.GROUND: int(rng.integers(400, 800)), abi.PILLAR: int(rng.integers(150, 350)), abi.FACADE: int(rng.integers(500, 900)), abi.BEAM: 0, abi.ROOF: 0})
    # searched classes: 5-15 k target points; beam / roof ride along unused (cloned and cropped like the reference does), at the sizes of
    # script/config/lo_gflag_list_kitti_urban.txt's fixed-number down-sampling (source) and a few hundred un-down-sampled points (target)
    tgt = {abi.GROUND: int(rng.integers(2000, 6001)), abi.PILLAR: int(rng.integers(500, 2001)), abi.FACADE: int(rng.integers(2500, 7001)),
           abi.BEAM: int(rng.integers(300, 901)), abi.ROOF: int(rng.integers(200, 601))}
    return ({abi.GROUND: 800, abi.PILLAR: 400, abi.FACADE: 1200, abi.BEAM: 200, abi.ROOF: 100}, tgt)


def _make_scene(job):
    k, tiny = job
    src_counts, tgt_counts = _scene_sizes(k, tiny)
    if tiny:
        pair, T_gt = synth.make_pair(SEED0 + k, n_beams=12, n_az=300, src_counts=src_counts, tgt_counts=tgt_counts, vertex_count=0)
    else:
        pair, T_gt = synth.make_pair(SEED0 + k, src_counts=src_counts, tgt_counts=tgt_counts, vertex_count=0)
    return pair.tgt, pair.src, pair.init_guess, pair.tgt_bound, T_gt, pair.n_raw


def build_scenes(n_scenes, tiny, workers):
    """The distinct scenes of the pair list (ray-cast in a process pool: 1.7 s each)."""
    jobs = [(k, tiny) for k in range(n_scenes)]
    if workers > 1 and n_scenes > 1:
        import multiprocessing as mp

        with mp.get_context("fork").Pool(min(workers, n_scenes)) as pool:
            raw = pool.map(_make_scene, jobs)
    else:
        raw = [_make_scene(j) for j in jobs]
    scenes = []
    for tgt, src, guess, bound, T_gt, n_raw in raw:
        p = abi.PairData(tgt, src, init_guess=guess, tgt_bound=bound)
        p.n_raw = n_raw
        scenes.append((p, T_gt))
    return scenes


def demo_scenes():
    """configs[0]: the three registrations of tests/golden/demo_pair.npz (000000 <-> 000001, 000000 <-> 000015 with the identity and with an
    odometry's guess) as (PairData, the reference lines' Trans1_2) — real scans, the reference's own feature extraction."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_demo_pair as dp

    gold = np.load(dp.GOLD)
    scenes = []
    for name in dp.CASES:
        p = dp.pair_of(gold, name)
        p.n_raw = (len(gold["scan_0"]), len(gold["scan_15"]))
        p.ref_row = gold[name + "_result"]
        scenes.append((p, dp.unpack(gold[name + "_result"])[0]))
    return scenes, dp.reg_params()


def global_pair(scenes, g):
    """Pair g of the global list: scene g mod S; beyond the first S pairs the initial guess is the ground truth perturbed
    by (0.3 m, 0.5 deg) with a seed that depends on g only — the list does not depend on the number of ranks."""
    base, T_gt = scenes[g % len(scenes)]
    if g < len(scenes):
        return base
    rng = np.random.default_rng([SEED0, g])
    pert = synth.se3(*(rng.normal(0, 0.3 / np.sqrt(3), 3)), *(np.deg2rad(rng.normal(0, 0.5 / np.sqrt(3), 3))))
    p = abi.PairData(base.tgt, base.src, init_guess=pert @ T_gt, tgt_bound=base.tgt_bound)
    p.n_raw = base.n_raw
    return p


def rank_span(args, world, rank):
    if args.total_pairs:
        return shard.block_partition(args.total_pairs, world, rank)
    return rank * args.pairs, (rank + 1) * args.pairs


# ------------------------------------------------------------------------------------------------------------------
# other_configs: compact legs of the other BASELINE configurations inside the default line (the driver runs only `python bench.py --gpus 1`)
OTHER_LEGS = {
    # name: (pairs per step, timed steps, pairs checked against the oracle)
    "configs[2]": (32, 4, 8),
    "configs[4]": (16, 3, 4),
    "scan_to_map_20k": (64, 8, 8),
    "configs[3]_shard_128": (128, 10, 8),
    "configs[0]": (512, 4, 8),  # (+ the fixture's three registrations against the reference's own lines' results)
}


def _gen_other(name):
    """(forked worker, before the parent touches OpenMP or the device) the pairs of one leg's workload"""
    from mulls_amd import workloads as W

    n = OTHER_LEGS[name][0]
    if name == "configs[2]":
        return W.submap_batch(n)
    if name == "configs[4]":
        return W.dense_batch(n)
    return W.s2m20k_batch(n)


def other_params(name):
    from mulls_amd import workloads as W

    return {"configs[2]": W.submap_params, "configs[4]": W.dense_params, "scan_to_map_20k": W.s2m_params}[name]()


def other_leg(ctx, name, pairs, P, checks, workload, demo_refs=None):
    """One compact leg on the context of the headline run: its own resident batch, untimed priming, K timed steps with the search launches bracketed by
    hipEvents (the roofline fraction of SURVEY 8d's algorithmic bytes), the checker's pairs compared."""
    from mulls_amd import workloads as W

    n, steps, _ = OTHER_LEGS[name]
    n = len(pairs)
    b = ctx.batch(pairs)
    res = abi.make_result_array(n)
    # (the legs are timed as the library runs them by default — profiling keeps a batch from iterating as two sub-batches on two streams, 4 - 10 % at 512 pairs —
    # and the search launches are bracketed by hipEvents in as many further steps right after the timed ones; only a leg larger than any of today's is timed with them)
    events_live = n > 512
    ctx.set_profiling(2 if events_live else 0)
    t_prime, k = time.perf_counter(), 0
    while k < 3 or time.perf_counter() - t_prime < 0.25:
        b.run(P, results=res)
        k += 1
    acc = {"ms_nn": 0.0, "launches_nn": 0.0, "nn_src_pts": 0.0, "nn_tgt_unique": 0.0}
    t0 = time.perf_counter()
    for _ in range(steps):
        b.run(P, results=res)
        if events_live:
            pf = ctx.profile()
            for kk in acc:
                acc[kk] += getattr(pf, kk)
    el = time.perf_counter() - t0
    if not events_live:
        ctx.set_profiling(2)
        b.run(P, results=res)
        for _ in range(steps):
            b.run(P, results=res)
            pf = ctx.profile()
            for kk in acc:
                acc[kk] += getattr(pf, kk)
    ctx.set_profiling(0)
    launches = max(acc["launches_nn"], 1)
    avg_ms = acc["ms_nn"] / launches
    alg = (72.0 * acc["nn_src_pts"] + 16.0 * acc["nn_tgt_unique"]) / launches
    achieved = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    used = bytes(P.used_feature_type).decode()[:6]
    value = n * steps / el
    b_reg = W.algorithmic_bytes([res[i] for i in range(n)], used) / n
    out = {"workload": workload, "value": value, "unit": "registrations/s", "pairs_per_step": n, "steps": steps, "untimed_priming_steps": k, "ms_per_step": el / steps * 1e3,
           "mean_iterations": float(np.mean([res[i].iters for i in range(n)])), "all_code_1": bool(all(res[i].code == 1 for i in range(n))),
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "avg_launch_ms": avg_ms,
                        "launches": int(acc["launches_nn"]), "algorithmic_bytes_per_launch": alg, "whole_path_frac": b_reg * value / 1e9 / HBM_PEAK_GBS,
                        "kernel": "the correspondence search launches of one ICP iteration (the tier each class cloud runs on), hipEvents " +
                                  ("live in the timed steps" if events_live else "in %d steps right after the timed ones" % steps)}}
    if checks:
        out["delta_T_vs_ref"] = oracle_check_compare(checks, res)
    if demo_refs:  # the fixture's registrations themselves: against the REFERENCE'S OWN LINES' results
        dt_max = dr_max = 0.0
        codes_equal = True
        for i, (pd, T_ref) in enumerate(demo_refs):
            dt, dr = synth.pose_error(abi_T(res[i]), T_ref)
            dt_max, dr_max = max(dt_max, dt), max(dr_max, dr)
            codes_equal &= res[i].code == int(pd.ref_row[54])
        out["delta_T_vs_reference_lines"] = {"pairs_checked": len(demo_refs), "max_abs_dt_m": dt_max, "max_drot_rad": dr_max, "codes_equal": bool(codes_equal),
                                             "within_tolerance": bool(dt_max <= 1e-4 and dr_max <= 1e-4)}
    b.close()
    return out


# ------------------------------------------------------------------------------------------------------------------
# the engine: libmulls_hip.so through its C ABI.  No fallback: without the library or a gfx950 device this raises.
class HipEngine:
    name = "libmulls_hip.so"

    def __init__(self, device_index, nn_mode):
        from mulls_amd import lib

        self.ctx = lib.Context(device_index)
        self.ctx.set_nn_mode(nn_mode)
        self.batch = None

    def stage(self, pairs):
        self.batch = self.ctx.batch(pairs)  # H2D staging happens here, outside the timed region

    def run(self, P, results):
        self.batch.run(P, results=results)

    def run_from_host(self, pairs, P):
        """(results, seconds inside mulls_icp_batch): the mulls_pair array is built beforehand — a C++ caller holds its clouds already"""
        m = (abi.make_pair_array(pairs), abi.make_result_array(len(pairs)))
        r = self.ctx.icp_batch(pairs, P, marshalled=m)
        return r, self.ctx.last_call_s

    def set_profiling(self, on):
        self.ctx.set_profiling(on)

    def profile(self):
        return self.ctx.profile()

    def close(self):
        if self.batch is not None:
            self.batch.close()
        self.ctx.close()


# ------------------------------------------------------------------------------------------------------------------
# CPU legs (rank 0): the oracle as the reported baseline and as the checker of the timed workload
def _oracle_many(job):
    from oracle import pyoracle

    idx, budget_s, tiny = job  # the scenes are inherited through fork (cpu_baseline_manycore)
    P = bench_params(tiny)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        pyoracle.icp(_MC_SCENES[(idx + n) % len(_MC_SCENES)][0], P, nn_mode=0, use_omp=1)
        n += 1
    return n, time.perf_counter() - t0


_MC_SCENES = None


def cpu_baseline(scenes, P, budget_s=10.0):
    """Oracle timed on the host cores: bounded sample of the same workload, back to back like the reference's caller."""
    from oracle import pyoracle

    pyoracle.icp(scenes[0][0], P)  # warm-up (thread pool, page-in)
    n, t0 = 0, time.perf_counter()
    while True:
        pyoracle.icp(scenes[n % len(scenes)][0], P, nn_mode=0, use_omp=1)
        n += 1
        el = time.perf_counter() - t0
        if (el > budget_s and n >= 16) or n >= 4096:
            break
    return {
        "value": n / el, "unit": "registrations/s", "cores": 3, "kind": "port",
        "sample": "%d registrations of the same workload back-to-back in %.1f s; oracle/mulls_oracle.cpp, kd-tree NN, "
                  "the reference's 3 OpenMP sections (effective width 3 of %d host cores)" % (n, el, os.cpu_count()),
    }


def cpu_baseline_manycore(scenes, tiny, procs=None, budget_s=6.0):
    """'N processes in parallel for a fair many-core number' — the whole host: one oracle process per three cores (the reference's loop is three
    OpenMP sections wide), side by side."""
    import multiprocessing as mp

    global _MC_SCENES
    _MC_SCENES = scenes
    host = os.cpu_count() or 1
    procs = max(1, host // 3) if procs is None else procs
    os.environ.setdefault("OMP_NUM_THREADS", "3")
    with mp.get_context("fork").Pool(procs) as pool:
        out = pool.map(_oracle_many, [(7 * i, budget_s, tiny) for i in range(procs)])
    total = sum(n for n, _ in out)
    wall = max(t for _, t in out)
    return {"value": total / wall, "unit": "registrations/s", "cores": 3 * procs, "host_cores": host, "kind": "port",
            "sample": "the whole host: %d oracle processes x 3 OpenMP sections side by side (%d of %d cores), %d registrations in %.1f s" % (procs, 3 * procs, host, total, wall)}


def cpu_baseline_reference_lines(scenes, P, budget_s=6.0):
    """The reference's OWN function bodies (oracle/_ref/libmulls_ref.so: cregistration.hpp's lines cut out at build time, compiled against the stand-in
    for PCL / Eigen in oracle/ref_shim) on the same workload, beside the port.  None where the library was not built."""
    try:
        from oracle import pyref

        if not pyref.available():
            return None
        pyref.icp(scenes[0][0], P)
        n, t0 = 0, time.perf_counter()
        while True:
            pyref.icp(scenes[n % len(scenes)][0], P)
            n += 1
            el = time.perf_counter() - t0
            if (el > budget_s and n >= 8) or n >= 4096:
                break
        return {"value": n / el, "unit": "registrations/s", "cores": 3, "kind": "reference lines + stand-in PCL",
                "sample": "%d registrations back-to-back in %.1f s through the reference's own mm_lls_icp lines (its kd-tree and Eigen calls answered by oracle/ref_shim)" % (n, el)}
    except Exception as e:  # the checker's optional strengthening: never a reason to lose the bench line
        return {"error": repr(e)}


def oracle_check_prepare(pairs, P, n_check):
    """Oracle results of n_check pairs spread over this rank's workload (computed before the device is touched)."""
    from oracle import pyoracle

    idx = sorted(set(int(i) for i in np.linspace(0, len(pairs) - 1, min(n_check, len(pairs)))))
    return [(i, pyoracle.icp(pairs[i], P)[0]) for i in idx]


def oracle_check_compare(checks, results):
    dt_max = dr_max = 0.0
    ints_equal = True
    for i, ro in checks:
        rg = results[i]
        dt, dr = synth.pose_error(abi_T(rg), abi_T(ro))
        dt_max, dr_max = max(dt_max, dt), max(dr_max, dr)
        ints_equal &= rg.code == ro.code and rg.iters == ro.iters and list(rg.ncorr) == list(ro.ncorr)
    return {"pairs_checked": len(checks), "max_abs_dt_m": dt_max, "max_drot_rad": dr_max, "integer_outputs_equal": bool(ints_equal),
            "tolerance": "1e-4 m / 1e-4 rad (north_star)", "within_tolerance": bool(dt_max <= 1e-4 and dr_max <= 1e-4),
            "checker": "oracle/mulls_oracle.cpp on the same pairs of the timed workload, outside the timed region"}


def abi_T(r):
    return np.array(r.T[:]).reshape(4, 4).T


# ------------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def respawn(args, argv):
    """--gpus N without a torch.distributed environment: start the N ranks ourselves."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(sys.argv[0])] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs[2] and configs[4]: the large-cloud configurations as resident batches (one process per GPU, weak scaling: every rank its own batch)
LARGE = {
    2: dict(pairs=32, name="configs[2]", metric="scan-to-local-map registrations/sec (~1 M-point map, 20 ICP iters); dT vs ref",
            kernel="correspondence search of one ICP iteration, global-memory tier: k_cert_big (rigid step, certificates, exact fixed-radius 1-NN of the uncertified queries on an "
                   "occupancy-bitmap grid) + k_filter (duplicate rule, rejection chain)"),
    4: dict(pairs=16, name="configs[4]", metric="scan-pair registrations/sec (128-beam ~240k pts, six classes, 40 ICP iters); dT vs ref",
            kernel="correspondence search of one ICP iteration, global-memory tier: k_cert_big (rigid step, certificates, exact fixed-radius 1-NN of the uncertified queries on an "
                   "occupancy-bitmap grid) + k_filter (duplicate rule, rejection chain)"),
}


def large_workload(cfg, n_pairs, rank):
    from mulls_amd import workloads as W

    if cfg == 2:
        return W.submap_batch(n_pairs, seed=7 + 100 * rank), W.submap_params(), W.submap_params(converge=True)
    return W.dense_batch(n_pairs, seed=301 + 100 * rank), W.dense_params(), W.dense_params(converge=True)


def large_main(args, rank, local_rank, world, engine_factory=None):
    """python bench.py --config 2|4: the same JSON shape as the headline line — value, roofline of the dominant kernel (SURVEY 8d's algorithmic bytes of the
    search against its live hipEvent duration), cpu_baseline (the oracle on a bounded sample), delta_T_vs_ref."""
    from mulls_amd import workloads as W

    cfg = args.config
    L = LARGE[cfg]
    n_pairs = args.large_pairs or (4 if args.tiny else L["pairs"])
    if args.tiny:  # plumbing sizes: never a bench line
        pair, T_gt = synth.make_pair(SEED0 + rank, n_beams=16, n_az=400, src_counts={c: None for c in range(abi.NCLASS)} if cfg == 4 else synth.R_SOURCE,
                                     tgt_counts={c: None for c in range(abi.NCLASS)}, vertex_count=50)
        pairs = W._guesses(pair, T_gt, n_pairs, 5)
        P = abi.default_params(used_feature_type="111111" if cfg == 4 else "111110", max_iter_num=4, converge_translation=0.0, converge_rotation_d=0.0)
        Pc = None
    else:
        pairs, P, Pc = large_workload(cfg, n_pairs, rank)
    used = bytes(P.used_feature_type).decode()[:6]
    n_total = n_pairs * world
    cpu, checks = None, None
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import pyoracle

        if world == 1:
            pyoracle.icp(pairs[0], P)
            n, t0 = 0, time.perf_counter()
            while True:
                pyoracle.icp(pairs[n % len(pairs)], P, nn_mode=0, use_omp=1)
                n += 1
                el = time.perf_counter() - t0
                if (el > (1.0 if args.tiny else 10.0) and n >= 3) or n >= 512:
                    break
            cpu = {"value": n / el, "unit": "registrations/s", "cores": 3, "kind": "port",
                   "sample": "%d registrations of the same workload back-to-back in %.1f s; oracle/mulls_oracle.cpp, kd-tree NN, the reference's 3 OpenMP sections "
                             "(effective width 3 of %d host cores)" % (n, el, os.cpu_count())}
        checks = oracle_check_prepare(pairs, P, 2 if cfg == 4 else 4)

    import torch
    import torch.distributed as dist

    use_cuda = torch.cuda.is_available()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if use_cuda:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))  # RCCL
        else:
            dist.init_process_group(backend="gloo")
    elif use_cuda:
        torch.cuda.set_device(0)
    device = torch.device("cuda", local_rank if world > 1 else 0) if use_cuda else None
    engine = (engine_factory or HipEngine)(local_rank if world > 1 else 0, args.nn_mode)
    engine.stage(pairs)
    results = abi.make_result_array(n_pairs)
    counts = [n_pairs] * world

    def step(gather=True):
        engine.run(P, results)
        return shard.gather_results(shard.pack_results(results, n_pairs), device=device, counts=counts) if gather else None

    def barrier():
        if world > 1:
            dist.barrier()
        if use_cuda:
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=device if use_cuda else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    PRIME = 0
    engine.set_profiling(2)
    t_prime = time.perf_counter()
    while not args.tiny and (PRIME < 4 or time.perf_counter() - t_prime < 0.6):  # process start-up stalls (see main)
        step()
        PRIME += 1
    for _ in range(args.warmup):
        step()
    prof_keys = ("ms_nn", "launches_nn", "nn_src_pts", "nn_tgt_unique", "ms_setup", "ms_filter", "ms_accum", "ms_residual")
    acc = {k: 0.0 for k in prof_keys}
    barrier()
    t0 = time.perf_counter()
    gathered = None
    for _ in range(args.steps):
        gathered = step()
        pf = engine.profile()
        for k in ("ms_nn", "launches_nn", "nn_src_pts", "nn_tgt_unique"):
            acc[k] += getattr(pf, k)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    engine.set_profiling(1)
    for _ in range(args.steps):
        step()
        pf = engine.profile()
        for k in ("ms_setup", "ms_filter", "ms_accum", "ms_residual"):
            acc[k] += getattr(pf, k)
    engine.set_profiling(0)
    conv = None
    if Pc is not None and not args.no_converging:
        res_c = abi.make_result_array(n_pairs)
        engine.run(Pc, res_c)
        barrier()
        t3 = time.perf_counter()
        for _ in range(args.steps):
            engine.run(Pc, res_c)
        barrier()
        el_c = max_over_ranks(time.perf_counter() - t3)
        it_c = [res_c[i].iters for i in range(n_pairs)]
        conv = {"value": n_total * args.steps / el_c, "unit": "registrations/s", "ms_per_step": el_c / args.steps * 1e3, "mean_iterations": float(np.mean(it_c)),
                "all_converged_code_1": bool(all(res_c[i].code == 1 for i in range(n_pairs))),
                "params": "the same pairs with the call site's convergence thresholds (converge_tran 0.0005 m, converge_rot_d 0.001 deg) instead of the forced iteration count"}
    if rank == 0:
        n_reg = n_total * args.steps
        launches = max(acc["launches_nn"], 1)
        avg_ms = acc["ms_nn"] / launches
        alg_bytes = (72.0 * acc["nn_src_pts"] + 16.0 * acc["nn_tgt_unique"]) / launches  # 64 + 8 B per live source point, 16 B per target point of a searched cloud
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        b_reg = W.algorithmic_bytes([results[i] for i in range(n_pairs)], used) / n_pairs  # the whole path's B_reg per registration
        value = n_reg / elapsed
        p0 = pairs[0]
        pmc = None
        try:  # the committed PMC passes, only if they describe this very configuration
            with open(os.path.join(ROOT, "profiles", "pmc_traffic_cfg%d.json" % cfg)) as f:
                pmc = json.load(f)
            if pmc["config"]["pairs_per_gpu_per_step"] != n_pairs or args.tiny or args.nn_mode != 0:
                pmc = None
        except Exception:
            pmc = None
        out = {
            "metric": L["metric"], "value": value, "unit": "registrations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "untimed_priming_steps": PRIME,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": ("%s: " % L["name"]) + (
                    "scan-to-local-map, source %s points (ground / pillar / facade / beam / roof, fixed-number down-sampled) against ONE ~%.2f M-point map (%s per class: one "
                    "128 x 7500-ray revolution, nothing down-sampled), used_feature_type %s, 20 ICP iterations, weights 1111; every pair of the batch registers its own scan "
                    "guess against the map (a localisation server's load); clouds resident in HBM" % (
                        [len(c) for c in p0.src[:5]], sum(len(c) for c in p0.tgt) / 1e6, [len(c) for c in p0.tgt[:5]], used) if cfg == 2 else
                    "synthetic 128-beam scan pairs, ~%dk returns each, regime D (every return in a class cloud: source %s, target %s), all six classes, 40 ICP iterations, "
                    "weights 1111; clouds resident in HBM" % (sum(len(c) for c in p0.src[:5]) // 1000, [len(c) for c in p0.src], [len(c) for c in p0.tgt])),
                "pairs_per_gpu_per_step": n_pairs, "pairs_per_step": n_total, "registrations_timed": n_reg,
                "parallelism": "%d lock-step batch(es), one process per GPU, no data-path collective; result gather (%s) on rank 0" % (world, "RCCL" if use_cuda else "gloo"),
                "engine": engine.name, "all_code_1": bool((gathered[:, 52] == 1).all()), "mean_iterations": float(np.mean(gathered[:, 53])),
            },
            "roofline": {
                "kernel": L["kernel"], "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc["traffic_bytes_per_launch"] if pmc else None,
                "traffic_note": "bytes per launch set (search + rejection chain) from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this configuration (%s; 2 x FETCH + "
                                "WRITE per the gfx950 guide); null when no pass of this very configuration is committed" % (pmc.get("source") if pmc else "none committed for configs[%d] at this batch size" % cfg),
                "avg_launch_ms": avg_ms, "launches": int(acc["launches_nn"]), "algorithmic_bytes_per_launch": alg_bytes,
                "whole_path": {"B_reg_bytes_per_registration": b_reg, "achieved_GBs": b_reg * value / 1e9, "frac": b_reg * value / 1e9 / HBM_PEAK_GBS,
                               "note": "SURVEY 8d's B_reg (setup + every iteration's search and accumulation + residual pass) x registrations/s against the HBM peak"},
                "kernel_ms_per_step": {k: acc[k] / args.steps for k in ("ms_setup", "ms_nn", "ms_filter", "ms_accum", "ms_residual")},
                "note": "live hipEvent duration of the search launches of the timed steps (the library's stream)",
            },
        }
        if checks is not None:
            out["delta_T_vs_ref"] = oracle_check_compare(checks, results)
        if conv:
            out["value_converging"] = conv
        if cpu:
            out["cpu_baseline"] = cpu
            out["cpu_baseline"]["gpu_over_cpu"] = value / cpu["value"]
        print(json.dumps(out))
    engine.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main(argv=None, engine_factory=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn(args, argv))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.config in (2, 4):
        return large_main(args, rank, local_rank, world, engine_factory)

    # --- workload (CPU, before the device is touched: the scene pool and the oracle legs fork) --------------------------------
    P = bench_params(args.tiny)
    n_scenes = 4 if args.tiny else N_SCENES
    lo, hi = rank_span(args, world, rank)
    n_total = args.total_pairs if args.total_pairs else world * args.pairs
    workers = max(1, min(16, (os.cpu_count() or 1) // max(world, 1)))
    demo = args.data == "demo"
    # the default one-GPU line also carries compact legs of the other BASELINE configurations: their clouds are ray-cast by three forked workers while the
    # headline's scenes are (before this process runs an OpenMP region or touches the device)
    others_on = (world == 1 and not args.tiny and not demo and not args.total_pairs and not args.no_other_configs and args.nn_mode == 0 and engine_factory is None
                 and args.config == 1)
    other_pool, other_jobs = None, {}
    if others_on:
        import multiprocessing as mp

        other_pool = mp.get_context("fork").Pool(3)
        other_jobs = {name: other_pool.apply_async(_gen_other, (name,)) for name in ("configs[2]", "configs[4]", "scan_to_map_20k")}
    if demo:
        scenes, P = demo_scenes()
        args.no_converging = True  # the call site's own thresholds are this configuration's
    else:
        scenes = build_scenes(min(n_scenes, max(n_total, 1)), args.tiny, workers)
    pairs = [global_pair(scenes, g) for g in range(lo, hi)]
    other_work = {}
    if others_on:
        for name, job in other_jobs.items():
            other_work[name] = (job.get(), other_params(name))
        other_pool.close()
        other_pool.join()
        other_work["configs[3]_shard_128"] = ([global_pair(scenes, g) for g in range(OTHER_LEGS["configs[3]_shard_128"][0])], P)
        d_scenes, d_P = demo_scenes()
        n_demo = OTHER_LEGS["configs[0]"][0]
        other_work["configs[0]"] = ([global_pair(d_scenes, g) for g in range(n_demo)], d_P)
    checks, cpu, cpu_mc, cpu_ref, checks_conv = None, None, None, None, None
    other_checks = {}
    if rank == 0 and not args.no_cpu_baseline:
        if world == 1:  # first: its children fork, and libgomp does not survive a fork once this process has run a parallel region
            if not demo:
                cpu_mc = cpu_baseline_manycore(scenes, args.tiny, procs=2 if args.tiny else None, budget_s=1.0 if args.tiny else 6.0)
            cpu = cpu_baseline(scenes, P, budget_s=2.0 if args.tiny else 10.0)
            cpu_ref = cpu_baseline_reference_lines(scenes, P, budget_s=1.0 if args.tiny else 6.0)
        checks = oracle_check_prepare(pairs, P, 16) if pairs else []
        checks_conv = oracle_check_prepare(pairs, converging_params(args.tiny), 8) if pairs and not args.no_converging else None
    if others_on:  # the checker's results of a few pairs of every leg (also with --no-cpu-baseline: a leg without its delta T is half a leg)
        for name, (opairs, oP) in other_work.items():  # (configs[0] is held to the reference's own lines' results too, which travel in the fixture)
            other_checks[name] = oracle_check_prepare(opairs, oP, OTHER_LEGS[name][2])

    import torch
    import torch.distributed as dist

    use_cuda = torch.cuda.is_available()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if use_cuda:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))  # RCCL
        else:
            dist.init_process_group(backend="gloo")  # CPU plumbing tests only: the engine below still needs a GPU unless a test injects one
        assert dist.get_world_size() == args.gpus
    elif use_cuda:
        torch.cuda.set_device(0)
    device = torch.device("cuda", local_rank if world > 1 else 0) if use_cuda else None

    engine = (engine_factory or HipEngine)(local_rank if world > 1 else 0, args.nn_mode)
    engine.stage(pairs)
    results = abi.make_result_array(max(len(pairs), 1))

    # the ranks' row counts follow from the partition: no size exchange, no host sync before the gather (shard.gather_post)
    counts = [rank_span(args, world, r)[1] - rank_span(args, world, r)[0] for r in range(world)]

    def step_post(gather=True):
        """one step; its result gather is only POSTED — completed by shard.gather_wait() while the next step's kernels run"""
        if pairs:
            engine.run(P, results)
        return shard.gather_post(shard.pack_results(results, len(pairs)), device=device, counts=counts) if gather else None

    def step():
        return shard.gather_wait(step_post())

    def barrier():
        if world > 1:
            dist.barrier()
        if use_cuda:
            torch.cuda.synchronize()

    # Process start-up, before the W warm-up steps: the first ~5 runs of a batch in a fresh process hold two or three 40-ms stalls on the host side
    # of the launch path (the HIP runtime growing its kernel-argument and signal pools: kernel times are unchanged, the GPU sits idle; tools/gpu_levels.py,
    # profiles/r03_process_startup.txt) — they would land in the timed steps whenever W is small.  Untimed runs with the timed region's event
    # bracketing take them — at least 8 and at least 0.6 s of them (a 1024-pair step is 5 ms: eight of those end before the stalls do); reported as
    # `untimed_priming_steps`.
    PRIME = 0
    engine.set_profiling(2)
    t_prime = time.perf_counter()
    while not args.tiny and (PRIME < 8 or time.perf_counter() - t_prime < 0.6):  # the stalls come with the process's first ~0.3 s of launches, whatever a step takes
        step()
        PRIME += 1
    engine.set_profiling(0)
    for _ in range(args.warmup):
        step()

    # Timed region: hipEvent pairs around the dominant kernel — the correspondence search of every iteration — on the library's stream
    # (profiling level 2: two events per iteration, the iteration hand-over untouched), so that roofline.achieved is that kernel's launch
    # duration measured live in the very steps that are timed.  The other kernels' times (kernel_ms_per_step) come from the same number of
    # untimed steps with events around every launch (level 1), which also gives value_all_kernel_events_on.
    prof_keys = ("ms_nn", "launches_nn", "nn_pair_evals", "nn_src_pts", "nn_tgt_unique", "nn_tgt_pts", "ms_setup", "ms_filter", "ms_accum", "ms_residual", "nn_corr_pts", "icp_loop_ms")
    acc = {k: 0.0 for k in prof_keys}
    # A shard of a few hundred pairs (configs[3] over 8 GPUs: 128 pairs, a 1.2 ms step) is timed WITHOUT the event pairs — forty event records and their read-back
    # are a tenth of such a step — and the search launches are bracketed in K further steps right after the timed ones (roofline.measured_in says which)
    events_live = len(pairs) >= 512 or args.tiny
    engine.set_profiling(2 if events_live else 0)
    step()  # one more untimed step with the timed region's event bracketing on (the first event records of a process are slow)
    barrier()
    t0 = time.perf_counter()
    gathered, pending = None, None
    step_s = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        h = step_post()  # the gather of step k travels while step k + 1 iterates; every one of the K gathers completes inside the timed region
        gathered = shard.gather_wait(pending) if pending is not None else gathered
        pending = h
        step_s.append(time.perf_counter() - ts)
        if events_live:
            pf = engine.profile()
            for k in prof_keys:
                acc[k] += pf.icp_phase_ms[5] if k == "icp_loop_ms" else getattr(pf, k)
        if os.environ.get("MULLS_BENCH_TRACE"):
            tp = engine.profile()  # (batches below 512 pairs run the timed steps without the event pairs: the host-side fields are filled either way)
            print("step %.2f ms: host launch %.2f wait %.2f search %.2f" % (step_s[-1] * 1e3, tp.ms_host_launch, tp.ms_host_wait, tp.ms_nn), file=sys.stderr)
    gathered = shard.gather_wait(pending)
    barrier()
    elapsed = time.perf_counter() - t0
    if not events_live:
        engine.set_profiling(2)
        step()
        for _ in range(args.steps):
            step()
            pf = engine.profile()
            for k in prof_keys:
                acc[k] += pf.icp_phase_ms[5] if k == "icp_loop_ms" else getattr(pf, k)
    # the same K steps without the result gather (N > 1: what the exchange step costs)
    elapsed_nogather = None
    if world > 1:
        barrier()
        t5 = time.perf_counter()
        for _ in range(args.steps):
            step_post(gather=False)
        barrier()
        elapsed_nogather = time.perf_counter() - t5
    engine.set_profiling(1)
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
        pf = engine.profile()
        for k in ("ms_setup", "ms_filter", "ms_accum", "ms_residual"):
            acc[k] += getattr(pf, k)
    barrier()
    elapsed_all = time.perf_counter() - t1
    engine.set_profiling(0)

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=device if use_cuda else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed, elapsed_all = max_over_ranks(elapsed), max_over_ranks(elapsed_all)
    if elapsed_nogather is not None:
        elapsed_nogather = max_over_ranks(elapsed_nogather)

    # --- the realistic case next to the metric's case: same pairs, the call site's convergence thresholds ---------------------------
    conv = None
    if not args.no_converging:
        Pc = converging_params(args.tiny)
        res_c = abi.make_result_array(max(len(pairs), 1))

        def step_c():
            if pairs:
                engine.run(Pc, res_c)
            return shard.gather_results(shard.pack_results(res_c, len(pairs)), device=device)

        step_c()
        barrier()
        t3 = time.perf_counter()
        gathered_c = None
        for _ in range(args.steps):
            gathered_c = step_c()
        barrier()
        el_c = max_over_ranks(time.perf_counter() - t3)
        if rank == 0:
            it_c = gathered_c[:, 53]
            conv = {"value": n_total * args.steps / el_c, "unit": "registrations/s", "ms_per_step": el_c / args.steps * 1e3,
                    "mean_iterations": float(np.mean(it_c)), "min_iterations": int(it_c.min()), "max_iterations": int(it_c.max()),
                    "all_converged_code_1": bool((gathered_c[:, 52] == 1).all()),
                    "params": "test/mulls_slam.cpp:642-648 call-site values: converge_tran 0.0005 m, converge_rot_d 0.001 deg, max 20 iterations; same pairs, same batch"}
            if checks_conv is not None:
                conv["delta_T_vs_ref"] = oracle_check_compare(checks_conv, res_c)

    # --- the timed configuration again over a fixed wall budget (a sampler outside the process then sees the device busy) -----------------
    sustained = None
    if args.sustain_s > 0 and pairs:
        barrier()
        t4 = time.perf_counter()
        k_s = 0
        while True:
            step()
            k_s += 1
            stop = time.perf_counter() - t4 >= args.sustain_s
            if world > 1:
                tstop = torch.tensor([1.0 if stop else 0.0], dtype=torch.float64, device=device if use_cuda else "cpu")
                dist.all_reduce(tstop, op=dist.ReduceOp.MAX)
                stop = bool(tstop.item() > 0)
            if stop:
                break
        barrier()
        el_s = max_over_ranks(time.perf_counter() - t4)
        sustained = {"value": n_total * k_s / el_s, "unit": "registrations/s", "steps": k_s, "seconds": el_s}

    e2e = None
    if rank == 0 and world == 1 and pairs and not args.no_end_to_end:
        sub = pairs[: min(1024, len(pairs))]
        lean = hasattr(engine, "ctx")
        if lean:  # what the C++ bridge switches on: only the clouds the registration reads are staged (28 live bytes of each 48-byte record)
            lean_before = engine.ctx.get_option(abi.OPT_LEAN_STAGING)
            engine.ctx.set_option(abi.OPT_LEAN_STAGING, 1)
        engine.run_from_host(sub, P)
        calls = []  # the median of seven calls (the context's cached batch is reused: no allocation per call); the slowest is reported beside it
        for _ in range(7):
            t2 = time.perf_counter()
            out2 = engine.run_from_host(sub, P)
            wall = time.perf_counter() - t2
            r2, dt_call = out2 if isinstance(out2, tuple) else (out2, None)
            calls.append((dt_call if dt_call is not None else wall, engine.profile()))  # (an engine without the split: the whole call)
        calls.sort(key=lambda c: c[0])
        e2e_dt, pf2 = calls[len(calls) // 2]
        if lean:
            engine.ctx.set_option(abi.OPT_LEAN_STAGING, lean_before)
        same = all(list(r2[i].T[:]) == list(results[i].T[:]) and r2[i].code == results[i].code and list(r2[i].info[:]) == list(results[i].info[:]) for i in range(len(sub)))
        caller_mb = sum(len(c) for p in sub for c in p.tgt + p.src) * abi.POINT_BYTES / 1e6
        # ... and the same calls through mulls_pipe (include/mulls_hip.h): begun back to back on alternating contexts, so that call k + 1's host gather and
        # upload run under call k's kernels — what a C++ caller with a stream of requests gets (value_pipelined: calls / wall of the whole sequence)
        piped = None
        if lean:
            try:
                from mulls_amd import lib as mlib

                piped = {}
                for depth in (2, 3):
                    pipe = mlib.Pipe(0, depth)
                    pipe.set_option(abi.OPT_LEAN_STAGING, 1)
                    marsh = [(abi.make_pair_array(sub), abi.make_result_array(len(sub))) for _ in range(depth + 1)]
                    for k in range(depth):  # every lane's first call allocates its staging arenas
                        pipe.end(pipe.begin(sub, P, marshalled=marsh[k]))
                    n_calls, pend, last_res = 12, [], None
                    tq = time.perf_counter()
                    for k in range(n_calls):
                        pend.append(pipe.begin(sub, P, marshalled=marsh[k % len(marsh)]))
                        if len(pend) > depth - 1:
                            last_res = pipe.end(pend.pop(0))
                    while pend:
                        last_res = pipe.end(pend.pop(0))
                    wall = time.perf_counter() - tq
                    same_p = all(list(last_res[i].T[:]) == list(results[i].T[:]) and last_res[i].code == results[i].code and list(last_res[i].info[:]) == list(results[i].info[:])
                                 for i in range(len(sub)))
                    piped["depth_%d" % depth] = {"value": n_calls * len(sub) / wall, "ms_per_call": wall / n_calls * 1e3, "calls": n_calls, "equals_resident_results": bool(same_p)}
                    pipe.close()
            except Exception as e:  # never a reason to lose the bench line
                piped = {"error": repr(e)}
        e2e = {"value": len(sub) / e2e_dt, "unit": "registrations/s", "pairs": len(sub), "ms": e2e_dt * 1e3,
               "staged_MB": getattr(pf2, "stage_bytes", 0) / 1e6, "caller_clouds_MB": caller_mb, "ms_staging": getattr(pf2, "ms_stage", 0.0),
               "ms_host_gather": getattr(pf2, "ms_stage_pack", 0.0), "equals_resident_results": bool(same), "calls_ms": [round(c[0] * 1e3, 2) for c in calls], "ms_max": calls[-1][0] * 1e3,
               "value_slowest_call": len(sub) / calls[-1][0],
               "value_pipelined": piped,
               "note": "mulls_icp_batch: class clouds in host memory (48-byte PCL records) -> results; host gather of the live fields of the classes the "
                       "registration reads into pinned memory, upload (PCIe), clone, crop, index build, iterations, residual"}

    others = None
    if others_on and rank == 0:
        t_o = time.perf_counter()
        descr = {
            "configs[2]": "BASELINE configs[2]: %d scans (800 / 400 / 1200 / 300 / 200 points) per step against ONE ~1 M-point local map, classes 111110, 20 iterations; resident batch",
            "configs[4]": "BASELINE configs[4]: %d synthetic 128-beam scan pairs (~236 k returns each, every return in a class cloud) per step, six classes, 40 iterations; resident batch",
            "scan_to_map_20k": "the reference's own scan-to-map regime (src/map_manager.cpp:73-86): %d scans per step against 20 000-point local maps with an 11.5 k-point ground class, "
                               "classes 111000, 20 iterations; resident batch",
            "configs[3]_shard_128": "BASELINE configs[3]'s shard: the %d first pairs of the global list = what each of 8 GPUs registers of the 1024 (same pairs as configs[1]); resident batch",
            "configs[0]": "BASELINE configs[0]: the reference's demo scans (000000 <-> 000001 / 000015), its own class clouds, test/mulls_reg.cpp:194-195's arguments (<= 10 iterations), "
                          "%d pairs per step (the three registrations, then the reference result perturbed by (0.3 m, 0.5 deg) as the guess); resident batch",
        }
        others = {}
        for name in ("configs[0]", "configs[2]", "configs[3]_shard_128", "configs[4]", "scan_to_map_20k"):
            opairs, oP = other_work[name]
            try:
                others[name] = other_leg(engine.ctx, name, opairs, oP, other_checks.get(name), descr[name] % len(opairs), demo_refs=d_scenes if name == "configs[0]" else None)
            except Exception as e:  # a leg must never cost the headline line
                others[name] = {"error": repr(e)}
        others["seconds"] = time.perf_counter() - t_o
        others["note"] = ("compact legs in the headline run's process and context, each a resident batch of its own: value = pairs x steps / wall of the timed steps; "
                          "roofline.frac = SURVEY 8d's algorithmic bytes of the search launches / their live hipEvent duration / 8 TB/s; full-length lines: python bench.py --config k")

    if rank == 0:
        n_reg = n_total * args.steps
        codes, iters = gathered[:, 52], gathered[:, 53]
        tsha = hashlib.sha256(np.ascontiguousarray(gathered).tobytes()).hexdigest()
        if args.dump_table:
            np.save(args.dump_table, gathered)
        launches = max(acc["launches_nn"], 1)
        avg_ms = acc["ms_nn"] / launches
        # Algorithmic bytes per launch of the search: per live source point 64 B (pos+nrm read and written back by
        # the fused transform) + 8 B (index, d2 out); per target point of a searched class cloud 16 B (pos).
        resident = acc["icp_loop_ms"] > 0  # the device-resident loop ran (k_icp: search AND normal equations of every iteration in one launch)
        alg_bytes = (72.0 * acc["nn_src_pts"] + 16.0 * acc["nn_tgt_unique"] + (72.0 * acc["nn_corr_pts"] if resident else 0.0)) / launches
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        brute = acc["nn_pair_evals"] > 0
        valu = acc["nn_pair_evals"] * OPS_PER_EVAL / (acc["ms_nn"] * 1e-3) / 1e12 if brute and acc["ms_nn"] > 0 else 0.0
        pmc = None
        try:  # the committed PMC passes, only if they describe this very configuration
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f)
            if not (pmc["config"]["pairs_per_gpu_per_step"] == len(pairs) and not brute and args.nn_mode in (0, 3) and not args.tiny
                    and pmc["config"].get("workload_seed") == SEED0):
                pmc = None
        except Exception:
            pmc = None
        # pairs one launch of the search covers (the lock-step loop queues one launch per iteration for the whole batch, or one per sub-batch)
        pairs_per_launch = len(pairs) * float(np.mean(iters)) * args.steps / launches
        pmc_scale = pairs_per_launch / pmc["config"]["pairs_per_launch"] if pmc and pmc["config"].get("pairs_per_launch") else 1.0
        valu_view = None
        if brute:
            valu_view = {"achieved": valu, "peak": VALU_PEAK_TLOPS, "unit": "Tlane-op/s", "frac": valu / VALU_PEAK_TLOPS,
                         "distance_evals_per_launch": acc["nn_pair_evals"] / launches}
        elif pmc and pmc.get("valu_wave_insts_per_launch") and avg_ms > 0:
            # a wave64 VALU instruction holds its SIMD for 4 cycles: issue time = instructions x 4 / (SIMDs x clock)
            issue_ms = pmc["valu_wave_insts_per_launch"] * pmc_scale * 4.0 / (1024 * 2.4e9) * 1e3
            valu_view = {"valu_wave_insts_per_launch": pmc["valu_wave_insts_per_launch"] * pmc_scale, "issue_ms": issue_ms, "frac_of_launch": issue_ms / avg_ms,
                         "source": pmc.get("source_sq"), "note": "SQ_INSTS_VALU of the committed pass x 4 cycles / (1024 SIMDs x 2.4 GHz) against the live launch duration"}
        sizes = sorted(sum(len(s[0].tgt[c]) for c in (abi.GROUND, abi.PILLAR, abi.FACADE)) for s in scenes)
        out = {
            "metric": "scan-pair registrations/sec (64-beam ~120k pts, 20 ICP iters); dT vs ref",
            "value": n_reg / elapsed,
            "unit": "registrations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "untimed_priming_steps": PRIME,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.total_pairs else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "real" if demo else "synthetic",
            "profiling_events_on": "around the dominant kernel (correspondence search) in the timed steps" if events_live else "none in the timed steps (see roofline.measured_in)",
            "value_all_kernel_events_on": n_reg / elapsed_all,
            "value_without_gather": n_reg / elapsed_nogather if elapsed_nogather else None,
            "config": {
                "workload": ("configs[0]: the reference's demo data — 000000 <-> 000001, 000000 <-> 000015 (identity guess), 000000 <-> 000015 (odometry's guess), real 64-beam scans of "
                             "%d / %d points, class clouds by the reference's own extract_semantic_pts lines (ground normal method 3, script/run_mulls_reg.sh's flags), "
                             "mm_lls_icp as test/mulls_reg.cpp:194-195 calls it (10 iterations at most, corr_dis_thre 3.0, classes 111110, weights 1101); pair g = registration "
                             "g mod 3, beyond the first three with the reference result perturbed by (0.3 m, 0.5 deg) as its guess" % scenes[0][0].n_raw if demo else
                             "tiny plumbing test (not a bench line)" if args.tiny else
                             ("configs[3]: %d independent KITTI-like scan pairs block-partitioned over %d GPU(s); pairs as in configs[1]: " % (n_total, world)
                              if args.total_pairs else "configs[1]: ") +
                             "KITTI-like scan-to-scan, synthetic 64-beam scans (~%dk returns each), classes ground+pillar+facade (used_feature_type 111000), "
                             "source 800/400/1200, searched target classes drawn per scene (%d scenes, %d-%d points, median %d; beam / roof clouds carried unused), 20 ICP iterations, weights 1111, "
                             "clouds resident in HBM" % (scenes[0][0].n_raw[0] // 1000, len(scenes), sizes[0], sizes[-1], sizes[len(sizes) // 2])),
                "pairs_per_gpu_per_step": len(pairs),
                "pairs_per_step": n_total,
                "registrations_timed": n_reg,
                "pair_list": "one global list, seed %d: pair g = scene g mod %d with the g-th initial guess; rank r takes %s" % (
                    SEED0, len(scenes), "block_partition(total, N, r)" if args.total_pairs else "[r*B, (r+1)*B)"),
                "parallelism": "%d lock-step batch(es), one process per GPU, no data-path collective; result gather (%s) on rank 0" % (
                    world, "RCCL" if use_cuda else "gloo"),
                "engine": engine.name,
                "all_converged_code_1": bool((codes == 1).all()),
                "mean_iterations": float(np.mean(iters)),
                "result_table_sha256": tsha,
            },
            "roofline": {
                "kernel": "k_nn (fused source transform + exact LDS-tiled brute-force 1-NN search)" if brute else
                          "k_icp (device-resident loop: every ICP iteration of every pair in one launch — rigid step, certificates / exact grid search, "
                          "rejection chain, normal equations, 6x6 solve; algorithmic bytes = search + 72 B per correspondence and iteration)" if resident else
                          "correspondence search of one ICP iteration, LDS tier: k_cert (rigid step, certificates, rejection chain) + k_nn_lds (exact "
                          "fixed-radius 1-NN of the uncertified queries on a uniform grid staged in LDS)",
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc["traffic_bytes_per_launch"] * pmc_scale if pmc else None,
                "traffic_note": "bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this configuration (%s; 2 x FETCH + WRITE "
                                "per the gfx950 guide)%s; null when the run differs from it" % (
                                    pmc.get("source") if pmc else "profiles/pmc_traffic.json",
                                    "" if abs(pmc_scale - 1.0) < 0.01 else ", measured on launches of %d pairs and scaled by %.2f to this run's %.0f pairs per launch" % (
                                        pmc["config"]["pairs_per_launch"], pmc_scale, pairs_per_launch)),
                "avg_launch_ms": avg_ms, "launches": int(acc["launches_nn"]), "algorithmic_bytes_per_launch": alg_bytes,
                "measured_in": "the timed steps (hipEvents around the search launches of every iteration, live)" if events_live else
                               "%d steps run right after the timed ones with hipEvents around the search launches (batches below 512 pairs are timed without the event pairs)" % args.steps,
                "note": "the search is an irregular exact query: its light pass (k_cert) is bound by the latency of its memory round trips at the occupancy its "
                        "registers and LDS allow in the early iterations and sits at the HBM roof (4.5-5.2 TB/s of counted traffic) once every point certifies; "
                        "the heavy pass (k_nn_lds, first iterations) by LDS-latency and VALU issue with one workgroup per CU (DESIGN.md sections 4 and 12.3); "
                        "the HBM fraction on the algorithmic bytes is reported as the contract asks",
                "valu_view": valu_view,
                "kernel_ms_per_step": {k: acc[k] / args.steps for k in ("ms_setup", "ms_nn", "ms_filter", "ms_accum", "ms_residual")},
            },
        }
        if checks is not None:
            out["delta_T_vs_ref"] = oracle_check_compare(checks, results)
        if demo and lo == 0 and len(pairs) >= len(scenes):
            # the first three pairs are the fixture's registrations themselves: against the REFERENCE'S OWN LINES' results
            dt_max = dr_max = 0.0
            codes_equal = True
            for i, (p, T_ref) in enumerate(scenes):
                dt, dr = synth.pose_error(abi_T(results[i]), T_ref)
                dt_max, dr_max = max(dt_max, dt), max(dr_max, dr)
                codes_equal &= results[i].code == int(p.ref_row[54])
            out["delta_T_vs_reference_lines"] = {"pairs_checked": len(scenes), "max_abs_dt_m": dt_max, "max_drot_rad": dr_max, "codes_equal": bool(codes_equal),
                                                 "within_tolerance": bool(dt_max <= 1e-4 and dr_max <= 1e-4),
                                                 "checker": "tests/golden/demo_pair.npz: Trans1_2 of the reference's own mm_lls_icp lines (oracle/_ref) on the same clouds"}
        if conv:
            out["value_converging"] = conv
        if sustained:
            out["value_sustained"] = sustained
        if e2e:
            out["value_end_to_end"] = e2e
        if others:
            out["other_configs"] = others
        if cpu:
            out["cpu_baseline"] = cpu
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / cpu["value"]
        if cpu_mc:
            out["cpu_baseline_manycore"] = cpu_mc
            out["cpu_baseline_manycore"]["gpu_over_whole_host"] = out["value"] / cpu_mc["value"]
        if cpu_ref:
            out["cpu_baseline_reference_lines"] = cpu_ref
        print(json.dumps(out))
    engine.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
