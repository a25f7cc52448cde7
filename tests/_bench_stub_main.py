"""Entry point for tests/test_dist_cpu.py::test_bench_under_torchrun_with_stub_context: bench.main with the host-memory stub
context, launched exactly as the driver launches bench.py (python -m torch.distributed.run ... <this file> --gpus N ...)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench                                      # noqa: E402
from test_dist_cpu import StubContext             # noqa: E402
from zerovox_amd import config as zcfg            # noqa: E402


def factory(args, local_rank):
    stub = StubContext()
    stub._first = int(os.environ["RANK"]) * (args.batch or 32)
    return stub, (zcfg.medium_modelcfg("styletts"), None, None, None)


if __name__ == "__main__":
    sys.exit(bench.main(sys.argv[1:], ctx_factory=factory))
