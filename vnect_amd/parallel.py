"""Stream-replica sharding across the GPUs of one node (BASELINE.json configs[4]).

The path shards by independent video streams: one process per GPU, one handle per process, no collective on
the data path.  torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" in CPU tests) is used only for the
rendezvous, the barrier around the timed region and the max-reduce of the elapsed time.
"""
import os


def stream_seed(rank, frame, base=1234):
    """Seed of synthetic frame `frame` of the stream owned by `rank`."""
    return base + 1000 * rank + frame


class Group:
    """Thin wrapper: works with world size 1 without touching torch.distributed."""

    def __init__(self, backend="nccl"):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.backend = backend
        self._dist = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            kw = {}
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", self.local_rank)
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
            self._dist = dist

    def barrier(self):
        if self._dist:
            self._dist.barrier()

    def max_over_ranks(self, value):
        """MAX of a Python float over all ranks (every rank gets the result)."""
        if not self._dist:
            return float(value)
        import torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self._dist:
            self._dist.barrier()
            self._dist.destroy_process_group()
            self._dist = None


def aggregate_rate(world, steps_per_rank, elapsed_max):
    """Whole-job frames/s: every rank processed `steps_per_rank` frames within the slowest rank's time."""
    return world * steps_per_rank / elapsed_max
