"""Data-parallel retraining: one process per GPU, gradients averaged with a sum all-reduce
(RCCL over xGMI when the backend is "nccl"; "gloo" on CPU for tests).

The reference has no live multi-GPU code; this is the north_star's
"retraining shards by batch across the 8 GPUs of one node".  Each rank holds a full replica
(203 MB of weights), runs fwd+bwd on its B/N images with local BatchNorm statistics (same
semantics as the reference at the per-GPU batch size), and the engine hands the whole
gradient over as ONE flat fp32 buffer (50 655 389 elements = 202.6 MB) laid out in parameter
order.  Backward finishes the layers last-to-first, so the buffer becomes final from its tail:
the reducer all-reduces contiguous tail slices ("buckets" of >= 32 MB, no copies, no
per-parameter bookkeeping) asynchronously as soon as they are final, and the collectives run on
RCCL's stream under the rest of the backward pass (conv19-23 hold 61 % of the parameters and are
done after the first ~20 % of it).  Averaging (not summing) keeps the single-GPU meaning of
`loss / nB` (nets.py:600).  Masks are computed on rank 0 and broadcast.
"""
import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).
    The rank's GPU is selected BEFORE the process group is created and handed to it as
    `device_id`, so RCCL communicators and barriers are bound to the right device."""
    import os
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (see the environment notes)
    kwargs = {}
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
        torch.cuda.set_device(local)
        kwargs["device_id"] = torch.device("cuda", local)
    dist.init_process_group(backend=backend, **kwargs)
    return dist.get_rank(), dist.get_world_size()


class GradReducer:
    """Averages gradients across ranks.  `reduce_flat` is the engine path (one buffer);
    `reduce_params` covers modules whose gradients are separate tensors (flattened in chunks)."""

    def __init__(self, world_size=None, chunk_elems=64 << 20, bucket_elems=8 << 20):
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.chunk = int(chunk_elems)
        self.bucket = int(bucket_elems)
        self.bytes_reduced = 0
        self.collectives = 0
        self._pending = []       # async work handles of the current step
        self._lo = self._hi = None   # final-but-not-yet-sent tail slice [lo, hi)
        self._sent_lo = None     # everything in [sent_lo, end) has been handed to a collective

    # ---- overlapped path: the engine reports gradient slices as they become final (tail first)
    def ready(self, flat, lo, hi):
        """`flat[lo:hi]` is final (all kernels that write it are enqueued on the current stream).
        Slices arrive in descending, contiguous order; a collective is launched whenever the
        accumulated slice reaches `bucket_elems`."""
        if self.world == 1:
            return
        if self._hi is None:
            self._lo, self._hi = lo, hi
        else:
            if hi != self._lo:
                raise RuntimeError("gradient slices must arrive contiguously from the tail: got [%d, %d) after [%d, ...)"
                                   % (lo, hi, self._lo))
            self._lo = lo
        if self._hi - self._lo >= self.bucket:
            self._launch(flat)

    def _launch(self, flat):
        if self._hi is None or self._hi == self._lo:
            return
        for off in range(self._lo, self._hi, self.chunk):
            end = min(off + self.chunk, self._hi)
            self._pending.append(dist.all_reduce(flat[off:end], op=dist.ReduceOp.SUM, async_op=True))
            self.collectives += 1
        self._sent_lo = self._lo
        self._hi = self._lo          # empty slice; the next ready() must end here

    def finish(self, flat):
        """End of backward: send what is left, wait for every collective, average."""
        if self.world == 1:
            return flat
        if self._hi is None:                      # nothing was reported piecewise: whole buffer at once
            self._lo, self._hi = 0, flat.numel()
        elif self._lo != 0:
            raise RuntimeError("gradient slices [0, %d) were never reported" % self._lo)
        self._launch(flat)
        for w in self._pending:
            w.wait()
        self._pending = []
        self._lo = self._hi = self._sent_lo = None
        flat.div_(self.world)
        self.bytes_reduced += flat.numel() * flat.element_size()
        return flat

    def reduce_flat(self, flat):
        if self.world == 1:
            return flat
        # chunks of <= 256 MB keep ring steps pipelined over the 7 xGMI links without a giant staging buffer
        for off in range(0, flat.numel(), self.chunk):
            dist.all_reduce(flat[off:off + self.chunk], op=dist.ReduceOp.SUM)
        flat.div_(self.world)
        self.bytes_reduced += flat.numel() * flat.element_size()
        return flat

    def reduce_params(self, params):
        grads = [p.grad for p in params if p.grad is not None]
        if self.world == 1 or not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        self.reduce_flat(flat)
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()


def attach(model, reducer=None):
    """Hook the reducer behind the engine's backward: p.grad views see the averaged values."""
    reducer = reducer or GradReducer()
    model._grad_ready_hook = reducer.ready      # called per final tail slice during backward
    model._grad_hook = reducer.finish           # called once at the end of backward
    return reducer


def broadcast_masks(masks, src=0):
    """Rank `src` computed the masks (ranking stays single-GPU); everyone else receives them."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for m in masks:
            dist.broadcast(m, src=src)
    return masks


def broadcast_parameters(model, src=0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=src)
        if hasattr(model, "invalidate_packed"):
            model.invalidate_packed()
