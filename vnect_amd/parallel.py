"""Sharding across the GPUs of one node: host-side orchestration only, no tensors.

* Stream replicas (BASELINE.json configs[4]): independent video streams, one process per GPU, one handle per process, no
  collective on the data path.  torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" in CPU tests) carries only the
  rendezvous, the barrier around the timed region and the max-reduce of the elapsed time.
* Pyramid sharding (configs[3]): ONE stream, rank r runs scale r (the S images of the batch are independent through the
  net, /root/reference/src/estimator.py:75-80,100-104) and the ranks exchange their (46,46,84) maps once per frame --
  ncclAllGather or direct peer writes over xGMI, inside the library (include/vnect_abi.h).  `PyramidJob` is the host side of
  it: who makes the communicator id / exports the peer-memory handles, who gets which scale, and the rule that EVERY rank
  takes part in every inference (profiling loops included), because each inference contains the exchange.
"""
import os


def stream_seed(rank, frame, base=1234):
    """Seed of synthetic frame `frame` of the stream owned by `rank`."""
    return base + 1000 * rank + frame


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


class Group:
    """Thin wrapper: works with world size 1 without touching torch.distributed -- unless `always_init` asks for the process group
    anyway (a one-rank pyramid rehearsal: the same torch "nccl" == RCCL process group, all-reduce and broadcast an N-rank job runs,
    in the same process as the library's own RCCL communicator).  `device` overrides the rank's device ordinal (one-GPU rehearsals)."""

    def __init__(self, backend="nccl", always_init=False, device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.device = self.local_rank if device is None else int(device)
        self.backend = backend
        self._dist = None
        if self.world > 1 or always_init:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.world == 1:
                os.environ.setdefault("MASTER_PORT", str(_free_port()))
            kw = {}
            if backend == "nccl":
                have = torch.cuda.device_count()
                if self.device >= have:   # a clean failure that names the device, instead of a HIP error out of the rendezvous
                    raise RuntimeError("rank %d needs HIP device %d, but this machine exposes %d device(s): one rank per GPU"
                                       % (self.rank, self.device, have))
                kw["device_id"] = torch.device("cuda", self.device)
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
            self._dist = dist

    def barrier(self):
        if self._dist:
            self._dist.barrier()

    def count_ranks(self):
        """The world size as the backend itself sees it: SUM of 1 over all ranks through a real all-reduce (on the GPUs with
        "nccl" == RCCL).  bench.py reports it, so an N > 1 line says what its collective actually spanned."""
        if not self._dist:
            return 1
        import torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.ones(1, dtype=torch.int32, device=dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return int(t.item())

    def max_over_ranks(self, value):
        """MAX of a Python float over all ranks (every rank gets the result)."""
        if not self._dist:
            return float(value)
        import torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def broadcast_object(self, obj, src=0):
        """A small picklable object from rank `src` to every rank (the 128-byte ncclUniqueId of the pyramid path)."""
        if not self._dist:
            return obj
        box = [obj if self.rank == src else None]
        kw = {}
        if self.backend == "nccl":
            import torch
            kw["device"] = torch.device("cuda", self.device)
        self._dist.broadcast_object_list(box, src=src, **kw)
        return box[0]

    def all_gather_object(self, obj):
        """[rank 0's obj, rank 1's obj, ...] on every rank (the peer-memory handles of the p2p exchange)."""
        if not self._dist:
            return [obj]
        out = [None] * self.world
        self._dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self._dist:
            self._dist.barrier()
            self._dist.destroy_process_group()
            self._dist = None


def aggregate_rate(world, steps_per_rank, elapsed_max):
    """Whole-job frames/s: every rank processed `steps_per_rank` frames within the slowest rank's time."""
    return world * steps_per_rank / elapsed_max


class PyramidJob:
    """Host logic of `bench.py --pyramid` (and of any caller that shards one stream's pyramid over S GPUs).

    `make_handle(rank, world, exchange)` returns an object with the pyramid part of `_native.Handle`'s surface:
    `comm_unique_id()` (static / class level), `comm_init(rank, world, uid)`, `p2p_export() -> bytes`,
    `p2p_init(rank, world, [bytes] * world)`, `upload_frame(slot, frame)`, `infer_resident(slot, t2d, t3d)`.
    The GPU tests pass `_native.Handle`; the gloo CPU test passes a stub that records the calls.
    """

    def __init__(self, group, scales, make_handle, exchange="rccl"):
        if group.world != len(scales):
            raise ValueError("pyramid sharding runs one scale per rank: %d scales need %d ranks, got %d"
                             % (len(scales), len(scales), group.world))
        if exchange not in ("rccl", "p2p"):
            raise ValueError("exchange must be 'rccl' or 'p2p'")
        self.group, self.scales, self.exchange = group, list(scales), exchange
        self.rank, self.world = group.rank, group.world
        self.scale = self.scales[self.rank]     # rank r owns scale r: the order of the batch in gen_input_batch
        self.handle = make_handle(self.rank, self.world, exchange)
        self._connect()

    def _connect(self):
        """rank 0 makes the ncclUniqueId and torch.distributed carries it to the others (rccl); every rank exports the IPC
        handles of its gather buffer + flags and receives everybody's (p2p)."""
        g = self.group
        if self.exchange == "rccl":
            uid = g.broadcast_object(self.handle.comm_unique_id() if self.rank == 0 else None, src=0)
            self.handle.comm_init(self.rank, self.world, uid)
        else:
            handles = g.all_gather_object(self.handle.p2p_export())
            self.handle.p2p_init(self.rank, self.world, handles)

    def upload(self, frames):
        """Every rank sees the SAME stream (stream 0): all ranks upload the same frames into the same slots."""
        for k, f in enumerate(frames):
            self.handle.upload_frame(k, f)

    def run(self, n, nslots, t0, dt=1 / 30):
        """n synchronous frames on every rank (each contains the exchange); returns the last joints and the end time."""
        t, out = t0, None
        for i in range(n):
            t += dt
            out = self.handle.infer_resident(i % nslots, t, t + 1e-3)
        return out, t
