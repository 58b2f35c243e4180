"""Image-parallel execution across the GPUs of one node: one process per GPU, frames sharded
round-robin, NO collective on the data path.  The only exchange is the final metric reduction
(the reference accumulates per-image PSNR/SSIM into AverageMeters, YOND_SIDD.py:203-206, 653-656,
671-672): one all-reduce of a (2*(max_iter+2)+1)-element float64 vector -- RCCL over xGMI when the
backend is "nccl" (that IS RCCL on ROCm), gloo in the CPU tests.  56 bytes: latency bound, so ring
vs tree and per-link bandwidth are irrelevant here.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


# collectives this process has issued through this module (tests and bench.py report it: a one-rank torchrun job
# exercises the same RCCL calls as an eight-rank one)
STATS = {"all_reduce": 0, "barrier": 0, "backend": None}


def launched_by_torchrun():
    """True when the torchrun environment is present (RANK and WORLD_SIZE set) -- world size 1 included."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment: a process group is created whenever RANK / WORLD_SIZE
    are set -- also for WORLD_SIZE = 1, so that a one-GPU box runs the very collectives (RCCL all-reduce, barrier) an
    eight-GPU node will; a plain `python bench.py` without that environment stays a single process without a group.
    Returns (rank, local_rank, world)."""
    rank, local, world = env_world()
    if (world > 1 or launched_by_torchrun()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        STATS["backend"] = backend
    return rank, local, world


def _active():
    return dist.is_available() and dist.is_initialized()


def shard_indices(n_items, rank, world, sizes=None):
    """Frames -> ranks.  Equal-sized frames: round-robin (item k -> rank k mod world).  With `sizes`
    (pixel counts, e.g. the five SIDD phone models): longest-first greedy onto the least-loaded rank."""
    if sizes is None:
        return list(range(rank, n_items, world))
    order = sorted(range(n_items), key=lambda i: (-sizes[i], i))
    load = [0] * world
    mine = []
    for i in order:
        r = min(range(world), key=lambda j: (load[j], j))
        load[r] += sizes[i]
        if r == rank:
            mine.append(i)
    return sorted(mine)


class MetricSums:
    """Per-rank sums of the per-image metrics, reduced once at the end.
    Layout mirrors the reference's meters: [psnr_it0, ssim_it0, ..., psnr_last, ssim_last, count]."""

    def __init__(self, n_iters):
        self.n_iters = n_iters
        self.vec = torch.zeros(2 * (n_iters + 1) + 1, dtype=torch.float64)

    def update(self, psnrs, ssims):
        """psnrs / ssims: the values of the iterations that RAN for one image (round 2 may have been ended by the
        reference's guard, YOND_SIDD.py:445-447).  Mirrors multiprocess_plot (:643-672): an iteration without output
        feeds -1 into ITS meter (:644-647, `continue`); the 'last' meter gets the last value that was computed
        (:671-672 reuse the loop variables, which a skipped iteration leaves untouched)."""
        if not len(psnrs):
            raise ValueError("MetricSums.update needs the metrics of at least the first iteration")
        for it in range(self.n_iters):
            self.vec[2 * it] += psnrs[it] if it < len(psnrs) else -1.0
            self.vec[2 * it + 1] += ssims[it] if it < len(ssims) else -1.0
        self.vec[2 * self.n_iters] += psnrs[-1]
        self.vec[2 * self.n_iters + 1] += ssims[-1]
        self.vec[-1] += 1

    def reduce(self, device=None):
        """One all-reduce (sum); returns the dataset means as a dict.  Every rank gets the same result."""
        v = self.vec.clone()
        if _active():
            if dist.get_backend() == "nccl":
                v = v.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
            STATS["all_reduce"] += 1
            v = v.cpu()
        cnt = float(v[-1])
        out = {"count": int(cnt)}
        for it in range(self.n_iters):
            out[f"psnr_iter{it}"] = float(v[2 * it]) / max(cnt, 1)
            out[f"ssim_iter{it}"] = float(v[2 * it + 1]) / max(cnt, 1)
        out["psnr_last"] = float(v[2 * self.n_iters]) / max(cnt, 1)
        out["ssim_last"] = float(v[2 * self.n_iters + 1]) / max(cnt, 1)
        return out


def barrier():
    if _active():
        dist.barrier()
        STATS["barrier"] += 1


def max_over_ranks(x, device=None):
    if _active():
        t = torch.tensor([x], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        STATS["all_reduce"] += 1
        return float(t.item())
    return float(x)


def finalize():
    """Tear the process group down (end of a torchrun job)."""
    if _active():
        dist.destroy_process_group()
