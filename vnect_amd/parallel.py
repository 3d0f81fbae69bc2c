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

    def host_handoff(self, key, release, timeout_s=900.0):
        """A HOST-side rendezvous: the releasing rank sets `key` in the process group's store, every other rank blocks on it in a
        socket wait -- no collective is enqueued, so the waiting ranks' GPUs stay idle (an "nccl" barrier would park a spinning
        kernel on each of them: bench.py's same-job N = 1 reference wants the other seven GPUs drawing idle power).  Returns how it
        waited ("store", or "none" for a world of one / a backend without a store: the caller's barrier behind it still orders things)."""
        if not self._dist:
            return "none"
        try:
            import datetime
            store = self._dist.distributed_c10d._get_default_store()
            if release:
                store.set("vnect/" + key, b"1")
            else:
                store.wait(["vnect/" + key], datetime.timedelta(seconds=timeout_s))
            return "store"
        except Exception:   # noqa: an older torch without the accessor: fall through to the caller's barrier
            return "none"

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


# ----------------------------------------------------------------------------------------------------------------------------------
# Host placement of a rank (round 6): the rank's host thread -- and, through first touch, the pinned buffers the library allocates --
# on the cores LOCAL to its GPU (/sys/bus/pci/devices/<bdf>/local_cpulist), decided BEFORE the process makes its first GPU call.  One
# estimator per process is the reference's own model (/root/reference/run_estimator_ps.py:120-129); eight such processes on a
# two-socket node otherwise sit wherever the scheduler puts them, half of them a socket away from their device.  Everything here
# reads sysfs only (no HIP call), so it is testable on a CPU-only machine against a fake sysfs tree (tests/test_parallel_gloo.py).

def parse_cpulist(text):
    """'0-15,128-143' -> [0..15, 128..143] (the kernel's cpulist format; empty / whitespace -> [])."""
    cpus = set()
    for part in (text or "").strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            if ":" in hi:   # the kernel's "lo-hi:used/group" stride form does not occur for cpulists of a device; refuse it loudly
                raise ValueError("strided cpulist %r" % part)
            cpus.update(range(int(lo), int(hi) + 1))
        else:
            cpus.add(int(part))
    return sorted(cpus)


def format_cpulist(cpus):
    """[0,1,2,3,8,9] -> '0-3,8-9'."""
    cpus = sorted(set(cpus))
    out, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else "%d-%d" % (cpus[i], cpus[j]))
        i = j + 1
    return ",".join(out)


def gpu_bdfs(sysfs="/sys"):
    """PCI addresses of the GPUs in the order HIP enumerates them by default: the KFD topology's nodes with SIMDs, ascending node id
    (`location_id` = bus << 8 | devfn, `domain`).  [] when the machine has no KFD (this container)."""
    base = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    try:
        nodes = sorted((int(n) for n in os.listdir(base) if n.isdigit()))
    except OSError:
        return []
    out = []
    for n in nodes:
        props = {}
        try:
            for ln in open(os.path.join(base, str(n), "properties")):
                kv = ln.split()
                if len(kv) == 2:
                    props[kv[0]] = kv[1]
        except OSError:
            continue   # a node this user may not read (another container's device)
        if int(props.get("simd_count", "0")) == 0:
            continue   # a CPU node
        loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
        out.append("%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7))
    return out


def visible_order(env=None):
    """The physical indices behind HIP device ordinals 0, 1, ... when ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES
    restrict or reorder them (plain integer lists only; None = no restriction; a list by UUID is reported as unknown -> no binding)."""
    env = os.environ if env is None else env
    order = None
    for key in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):   # ROCr filters first, HIP indexes into what is left
        v = env.get(key)
        if v is None or not v.strip():
            continue
        try:
            idx = [int(x) for x in v.split(",") if x.strip()]
        except ValueError:
            return "unknown"
        order = idx if order is None else [order[i] for i in idx if i < len(order)]
    return order


def rank_binding(device, sysfs="/sys", env=None, allowed=None):
    """Where rank's host thread should sit: {"bdf", "numa_node", "local_cpulist", "cpus": [...]} for HIP device ordinal `device`, or
    {"cpus": None, "reason": ...}.  `allowed` = the process's present affinity (the cgroup / taskset limit): the result is the
    intersection, and an empty intersection means "leave the thread where it is"."""
    bdfs = gpu_bdfs(sysfs)
    if not bdfs:
        return {"cpus": None, "reason": "no KFD topology under %s (no GPU visible to sysfs)" % sysfs}
    order = visible_order(env)
    if order == "unknown":
        return {"cpus": None, "reason": "*_VISIBLE_DEVICES is not a list of integers: device order unknown"}
    phys = device if order is None else (order[device] if device < len(order) else None)
    if phys is None or phys >= len(bdfs):
        return {"cpus": None, "reason": "device %d is not among the %d GPU(s) sysfs shows" % (device, len(bdfs))}
    bdf = bdfs[phys]
    d = os.path.join(sysfs, "bus", "pci", "devices", bdf)
    try:
        text = open(os.path.join(d, "local_cpulist")).read().strip()
    except OSError:
        return {"cpus": None, "bdf": bdf, "reason": "no local_cpulist for %s" % bdf}
    try:
        numa = int(open(os.path.join(d, "numa_node")).read().strip())
    except (OSError, ValueError):
        numa = None
    local = parse_cpulist(text)
    cpus = local if allowed is None else [c for c in local if c in set(allowed)]
    if not cpus:
        return {"cpus": None, "bdf": bdf, "numa_node": numa, "local_cpulist": text,
                "reason": "none of the device's local cores is in this process's affinity mask"}
    return {"cpus": cpus, "bdf": bdf, "numa_node": numa, "local_cpulist": text}


def bind_rank(device, enable=True, sysfs="/sys"):
    """Bind THIS process to the cores local to HIP device `device`; call it before the first GPU call (before the HIP library is
    loaded: its helper threads and pinned allocations inherit the mask).  Returns the report bench.py prints in `ranks[]`."""
    try:
        before = sorted(os.sched_getaffinity(0))
    except AttributeError:   # not Linux
        return {"bound": False, "reason": "no sched_getaffinity on this platform"}
    if not enable:
        return {"bound": False, "reason": "--no-bind", "affinity": format_cpulist(before), "n_cpus": len(before)}
    try:
        b = rank_binding(device, sysfs=sysfs, allowed=before)
    except Exception as e:   # noqa: placement is an optimisation -- an unreadable sysfs entry never stops a job
        return {"bound": False, "reason": "sysfs: %s" % e, "affinity": format_cpulist(before), "n_cpus": len(before)}
    rep = {k: b[k] for k in ("bdf", "numa_node", "local_cpulist") if b.get(k) is not None}
    if not b["cpus"]:
        rep.update(bound=False, reason=b["reason"], affinity=format_cpulist(before), n_cpus=len(before))
        return rep
    try:
        os.sched_setaffinity(0, b["cpus"])
    except OSError as e:
        rep.update(bound=False, reason="sched_setaffinity: %s" % e, affinity=format_cpulist(before), n_cpus=len(before))
        return rep
    now = sorted(os.sched_getaffinity(0))
    rep.update(bound=True, affinity=format_cpulist(now), n_cpus=len(now), narrowed_from=len(before))
    return rep


def rebind_by_bus(report, pci_bus, sysfs="/sys"):
    """After HIP is up: `pci_bus` is the PCI bus number HIP reports for this rank's device.  If the sysfs chain bound the process to ANOTHER
    device's cores (HIP's device order differed from the KFD node order), bind the calling thread to the right ones now -- late for the
    library's helper threads and first-touch allocations, in time for the thread that launches every kernel -- and say so in the report.
    Returns the (updated) report; never raises."""
    try:
        if report.get("bdf") is None or pci_bus is None:
            return report
        report["bdf_is_the_hip_device"] = int(report["bdf"].split(":")[1], 16) == int(pci_bus)
        if report["bdf_is_the_hip_device"] or not report.get("bound"):
            return report
        match = [b for b in gpu_bdfs(sysfs) if int(b.split(":")[1], 16) == int(pci_bus)]
        if len(match) != 1:
            report["rebind"] = "HIP reports bus 0x%02x: %d matching device(s) in sysfs, binding left as it is" % (int(pci_bus), len(match))
            return report
        d = os.path.join(sysfs, "bus", "pci", "devices", match[0])
        local = parse_cpulist(open(os.path.join(d, "local_cpulist")).read())
        # the mask was narrowed to the WRONG device's cores: the process's original mask is gone, so take the right device's local list as it is
        os.sched_setaffinity(0, local)
        now = sorted(os.sched_getaffinity(0))
        report.update(bdf=match[0], affinity=format_cpulist(now), n_cpus=len(now), rebound_after_hip_init=True, bdf_is_the_hip_device=True)
        try:
            report["numa_node"] = int(open(os.path.join(d, "numa_node")).read().strip())
        except (OSError, ValueError):
            pass
    except Exception as e:   # noqa: placement is an optimisation, never a reason to fail a job
        report["rebind"] = "failed: %s" % e
    return report
