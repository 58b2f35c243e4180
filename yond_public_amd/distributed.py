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
STATS = {"all_reduce": 0, "all_gather": 0, "barrier": 0, "broadcast": 0, "grad_all_reduce": 0, "grad_bytes": 0, "backend": None}


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


def shard_dataset(ds, rank, world):
    """The indices of `ds` this rank evaluates: longest-first greedy on the items' full-frame pixel counts where they are known before
    loading (data.item_sizes: the SIDD validation set is 8 scenes from each of five phones with different frame sizes) and
    differ; round-robin otherwise."""
    from .data import item_sizes
    sizes = item_sizes(ds) if world > 1 else None
    if sizes is not None and len(set(sizes)) > 1:
        return shard_indices(len(ds), rank, world, sizes=sizes)
    return shard_indices(len(ds), rank, world)


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


def gather_over_ranks(x, device=None):
    """Every rank's value of the scalar x, in rank order (all_gather; [x] without a process group)."""
    if _active():
        t = torch.tensor([x], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
        out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(out, t)
        STATS["all_gather"] = STATS.get("all_gather", 0) + 1
        return [float(o.item()) for o in out]
    return [float(x)]


def group_world_size():
    """World size the process group itself reports (None without one): what bench.py checks against --gpus."""
    return dist.get_world_size() if _active() else None


class GradReducer:
    """The gradient exchange of data-parallel training: the reference wraps the network in
    torch's DistributedDataParallel (trainer_AWGN.py:59-61), i.e. every rank computes the gradient of the mean loss of ITS
    batch and the ranks' gradients are AVERAGED before the optimiser step.  Here explicitly: the parameters are laid out, in
    reverse registration order (the order backward produces their gradients in), in flat float32 buckets of `bucket_bytes`
    (DDP's default 25 MB: GuidedResUnet's 44.7 MB of gradients are two ring all-reduces -- on xGMI a ring is bound by one
    link's ~153 GB/s, so few large messages, not many small ones); a bucket's all-reduce (RCCL with backend "nccl", gloo in
    the CPU tests) is launched asynchronously from the hook of its last gradient, while backward is still producing the
    earlier layers' gradients; `finish()` waits, divides by the world size and points every .grad at its slice.
    Without a process group it leaves the gradients where they are."""

    def __init__(self, params, bucket_bytes=25 * 2 ** 20):
        self.params = [p for p in params if p.requires_grad]
        self.world = dist.get_world_size() if _active() else 1
        self.buckets = []                                   # [flat tensor, [(param, offset, numel)], pending count, work handle]
        cur, cur_n = [], 0
        for p in reversed(self.params):
            if cur and (cur_n + p.numel()) * 4 > bucket_bytes:
                self.buckets.append(self._make(cur, cur_n))
                cur, cur_n = [], 0
            cur.append((p, cur_n, p.numel()))
            cur_n += p.numel()
        if cur:
            self.buckets.append(self._make(cur, cur_n))
        self.where = {}
        for bi, b in enumerate(self.buckets):
            for p, off, n in b["items"]:
                self.where[id(p)] = (bi, off, n)
        self.hooks = [p.register_post_accumulate_grad_hook(self._ready) for p in self.params] if _active() else []
        self._copied = set()

    @staticmethod
    def _make(items, n):
        dev = items[0][0].device
        return {"flat": torch.zeros(n, dtype=torch.float32, device=dev), "items": list(items), "pending": len(items), "work": None}

    def begin(self):
        """Start of a step: nothing reduced yet."""
        self._copied = set()
        for b in self.buckets:
            b["flat"].zero_()
            b["pending"], b["work"] = len(b["items"]), None

    def _launch(self, b):
        b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, async_op=True)
        STATS["all_reduce"] += 1
        STATS["grad_all_reduce"] += 1
        STATS["grad_bytes"] += b["flat"].numel() * 4

    def _ready(self, p):
        bi, off, n = self.where[id(p)]
        b = self.buckets[bi]
        b["flat"][off:off + n].copy_(p.grad.reshape(-1))
        self._copied.add(id(p))
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def finish(self):
        """After backward: launch what is still waiting (a parameter without a gradient contributes zeros), wait for the
        reductions, average, and hand every parameter its slice as .grad."""
        if not _active():
            return
        for b in self.buckets:
            if b["work"] is None:
                # (a reducer built before the process group existed has no hooks: its gradients are gathered here)
                for p, off, n in b["items"]:
                    if id(p) not in self._copied and p.grad is not None:
                        b["flat"][off:off + n].copy_(p.grad.reshape(-1))
                self._launch(b)
        for b in self.buckets:
            b["work"].wait()
            if self.world > 1:
                b["flat"].mul_(1.0 / self.world)
            for p, off, n in b["items"]:
                p.grad = b["flat"][off:off + n].view_as(p)

    def remove(self):
        for h in self.hooks:
            h.remove()
        self.hooks = []


def broadcast_params(params, src=0):
    """Every rank starts from rank `src`'s weights (what DistributedDataParallel does when it wraps a module)."""
    if not _active():
        return
    ps = [p for p in params]
    if not ps:
        return
    flat = torch.cat([p.detach().reshape(-1).to(torch.float32) for p in ps])
    dist.broadcast(flat, src=src)
    STATS["broadcast"] += 1
    off = 0
    with torch.no_grad():
        for p in ps:
            p.copy_(flat[off:off + p.numel()].view_as(p))
            off += p.numel()


def finalize():
    """Tear the process group down (end of a torchrun job)."""
    if _active():
        dist.destroy_process_group()
