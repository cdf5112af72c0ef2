"""Data-parallel runtime: one process per GPU, RCCL over xGMI.

The reference's multi-GPU path is single-process `nn.DataParallel` (models/models.py:10-59): every
forward call re-broadcasts all parameters (1.46-1.66 GB for G) to the replicas and reduces the
gradients back to device 0; `n_gpus_gen` additionally pipelines the frames of ONE sequence over GPUs
to fit memory (models/vid2vid_model_G.py:126-133,152-153).  On MI355X a whole 2048x1024 chunk fits one
288 GB GPU, so the native design is plain data parallelism over sequences:

  * persistent per-rank replicas; ONE broadcast of rank 0's flat parameter buffer at start-up;
  * per optimizer (G, D, each D_T) ONE flat fp32 gradient buffer (optim.FlatBuffers), all-reduced in
    place (sum, then 1/world) in buckets sized for xGMI: the links are point-to-point
    (7 x ~153 GB/s per GPU), a ring all-reduce is per-link bound, so buckets are large (64 MB default)
    to stay bandwidth- rather than latency-bound, and they are issued back to back (async) so RCCL
    pipelines them; the D / D_T buffers (<= 45 MB) go as a single bucket;
  * per-rank BatchNorm statistics (no SyncBN) -- what the reference's DataParallel replicas do.

`torch.distributed` backend "nccl" IS RCCL on ROCm; the same code runs on "gloo" for the CPU tests
(tests/test_cpu_parallel.py, world size 2).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Join the process group described by torchrun's environment (RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, world, local_rank); a no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend, rank=rank, world_size=world)
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    return rank, world, local_rank


class GradSync:
    """Bucketed in-place all-reduce (mean) of a flat gradient buffer."""

    def __init__(self, group=None, bucket_bytes=64 << 20):
        self.group = group
        self.bucket_elems = max(1, bucket_bytes // 4)

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def buckets(self, flat):
        n = flat.numel()
        return [flat[o:min(o + self.bucket_elems, n)] for o in range(0, n, self.bucket_elems)]

    def all_reduce(self, flat):
        """Sum `flat` over the ranks in place; returns the factor (1/world) that turns the sum into the mean
        the reference's DataParallel implies (losses are averaged over replicas, train.py:65).  The factor is
        applied inside the fused optimizer kernel (v2v_adam_step grad_scale): no extra pass over the buffer."""
        world = self.world
        if world == 1:
            return 1.0
        works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for b in self.buckets(flat)]
        for w in works:
            w.wait()
        return 1.0 / world

    def broadcast(self, flat, src=0):
        if self.world > 1:
            dist.broadcast(flat, src=src, group=self.group)


def sync_optimizers(optimizers, group=None, bucket_bytes=64 << 20):
    """Attach a GradSync to each FusedAdam and make every rank start from rank 0's parameters."""
    gs = GradSync(group, bucket_bytes)
    for opt in optimizers:
        opt.grad_sync = gs
        gs.broadcast(opt.flat.flat_param)
    return gs


def frame_ranks(n_gpus_gen, world):
    """Optional role split kept from the reference's `n_gpus_gen` flag (README.md:175-177): ranks
    [0, n_gpus_gen) generate, the rest discriminate.  With pure data parallelism (the default, and the
    right choice with 288 GB per GPU) every rank does both."""
    if n_gpus_gen <= 0 or n_gpus_gen >= world:
        return list(range(world)), list(range(world))
    return list(range(n_gpus_gen)), list(range(n_gpus_gen, world))


def shared_tuning_cache(rank, world):
    """N > 1 inference replicas: rank 0 runs the tile searches (per-shape + whole-frame), the other ranks replay its
    selections through the tuning cache (`V2V_TUNE_CACHE`) -- the same kernels on every GPU instead of N independent, noisy
    searches.  Call BEFORE the first engine exists.  Ranks > 0 return only after rank 0 has called the returned `release()`
    (rank 0 calls it once its plan is built, i.e. once the cache file is complete); on ranks > 0 and in a single process
    `release` is a no-op.  Does nothing when the user already set V2V_TUNE_CACHE."""
    if world <= 1 or os.environ.get("V2V_TUNE_CACHE"):
        return lambda: None
    import tempfile
    cache = os.path.join(tempfile.gettempdir(), "v2v_tune_%s.json" % os.environ.get("MASTER_PORT", "0"))
    os.environ["V2V_TUNE_CACHE"] = cache
    if rank == 0 and os.path.exists(cache):
        os.remove(cache)
    dist.barrier()                                   # no stale file from an earlier job
    if rank == 0:
        return dist.barrier                          # released by the caller once rank 0's selections are on disk
    dist.barrier()                                   # wait for rank 0's selections
    return lambda: None

