"""bench.py on a CPU box, for the sharding / launcher tests only: the same main() with the per-rank compute replaced by the
CPU oracle (tests/ may use the oracle; bench.py itself never does outside its cpu_baseline / checker legs).  Started as a
script so that bench.py's own `--gpus N` launcher re-executes THIS file under torch.distributed.run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from mulls_amd import abi  # noqa: E402


class OracleEngine:
    name = "oracle (CPU plumbing test)"

    def __init__(self, device_index, nn_mode):
        self.pairs = []

    def stage(self, pairs):
        self.pairs = pairs

    def run(self, P, results):
        from oracle import pyoracle

        for i, p in enumerate(self.pairs):
            results[i] = pyoracle.icp(p, P)[0]

    def run_from_host(self, pairs, P):
        res = abi.make_result_array(len(pairs))
        self.stage(pairs)
        self.run(P, res)
        return res

    def set_profiling(self, on):
        pass

    def profile(self):
        return abi.Profile()

    def close(self):
        pass


if __name__ == "__main__":
    bench.main(engine_factory=OracleEngine)
