"""Development aid: bench.py's N > 1 path with every rank on GPU 0 (a one-GPU box): exercises the torchrun rendezvous, the RCCL
communicator and the grouped send/recv gather across PROCESSES when RCCL accepts several ranks on one device."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

def factory(args, local_rank):
    return bench.default_ctx_factory(args, 0)

if __name__ == "__main__":
    sys.exit(bench.main(sys.argv[1:], ctx_factory=factory))
