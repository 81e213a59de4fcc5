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

Transports (one ring hop moves 2*(7/8)*202.6 MB per GPU over a ~153 GB/s
xGMI link, the same order as the B=32 compute step):
  "fp32"   the flat buffer itself, in place (default; bit-identical ranks);
  "fp16"   each bucket is scaled, rounded to fp16, summed in fp16 and widened again: half the bytes,
           2^-11 relative rounding per element and rank (inside the 1e-3 DP parity bar, outside
           bit-exactness: opt-in, MCAMD_DP_TRANSPORT=fp16).  A scaled entry beyond 65504 / world (so that the
           fp16 SUM stays finite) is clamped and raises the reducer's device flag `overflow`: train.py's
           StepGuard skips that step on every rank and halves `fp16_scale` -- the engine's own policy for its
           fp16-stored gradients;
  sparse   with STATIC weight masks (`weight_prune`, methods.py:9-26) every masked gradient entry is an
           exact zero on every rank (`grad * mask`, layers.py:59), so only the kept entries travel:
           `set_static_masks` builds the kept-index list once, a bucket is gathered into a packed
           buffer, all-reduced and scattered back (80 % pruning: 10.1 M of 50.6 M elements, 5x less
           traffic).  The values the kept entries receive are the ones the dense all-reduce gives them.
"""
import os

import torch
import torch.distributed as dist


def rehearsal():
    """MCAMD_DP_REHEARSE=1: run the data-parallel machinery (process group, bucketed collectives, barriers) with
    however many ranks there are -- including one."""
    return os.environ.get("MCAMD_DP_REHEARSE", "0") == "1"


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).
    The rank's GPU is selected BEFORE the process group is created and handed to it as
    `device_id`, so RCCL communicators and barriers are bound to the right device.

    HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC, the only mode this pool's driver supports) is read by the
    HIP runtime when it initialises, i.e. possibly before this function runs: entry points export it at
    their very top (bench.py, train.py) and the launcher environment carries it; here it is only checked."""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and not rehearsal():
        return 0, 1
    if world == 1:
        # MCAMD_DP_REHEARSE=1: a ONE-rank process group, so that a 1-GPU box still executes every RCCL call of the
        # N > 1 path (communicator creation bound to the device, asynchronous all-reduces on the side stream, barriers)
        import socket
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if "MASTER_PORT" not in os.environ:
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
            s.close()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl" and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
        import warnings
        warnings.warn("HSA_ENABLE_IPC_MODE_LEGACY is not '0' in this process's environment: RCCL's IPC handles may "
                      "fail with `hipIpcGetMemHandle: invalid argument`; export it before the process starts")
    kwargs = {}
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
        torch.cuda.set_device(local)
        kwargs["device_id"] = torch.device("cuda", local)
        # RCCL's kernels on a HIGH-priority stream: HIP gives priority streams hardware queues of their own, so the
        # collectives can never land on the queue of the launch stream or of the weight-gradient stream and be
        # serialised with the backward pass they are supposed to run under (engine.py _get_side_stream found
        # exactly that collision between two normal-priority streams)
        try:
            opts = dist.ProcessGroupNCCL.Options()
            opts.is_high_priority_stream = True
            kwargs["pg_options"] = opts
        except AttributeError:
            pass
    dist.init_process_group(backend=backend, **kwargs)
    return dist.get_rank(), dist.get_world_size()


class GradReducer:
    """Averages gradients across ranks.  `ready` / `finish` is the engine path (one flat buffer, buckets
    overlapped with backward); `reduce_flat` the unoverlapped form; `reduce_params` covers modules whose
    gradients are separate tensors (flattened in chunks)."""

    def __init__(self, world_size=None, chunk_elems=64 << 20, bucket_elems=8 << 20, transport=None, fp16_scale=256.0):
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.active = self.world > 1 or (rehearsal() and dist.is_initialized())
        self.chunk = int(chunk_elems)
        self.bucket = int(bucket_elems)
        self.transport = transport or os.environ.get("MCAMD_DP_TRANSPORT", "fp32")
        if self.transport not in ("fp32", "fp16"):
            raise ValueError("transport must be 'fp32' or 'fp16' (got %r)" % (self.transport,))
        self.fp16_scale = float(fp16_scale)
        self.bytes_reduced = 0       # bytes handed to collectives (per rank)
        self.collectives = 0
        self._pending = []           # (work handle, completion callback or None) of the current step
        self._lo = self._hi = None   # final-but-not-yet-sent tail slice [lo, hi)
        self._sent_lo = None         # everything in [sent_lo, end) has been handed to a collective
        self._kept = None            # static-mask transport: sorted int64 indices of the kept entries of the flat buffer
        self._kept_pos = None        # {flat offset: number of kept entries below it} at the parameter boundaries
        self._kept_total = 0
        self.prescaled = False       # attach(): the engine already divides every gradient by the world size
        self.overflow = None         # fp16 transport: device int32[1], set when a scaled entry had to be clamped

    # ---- static masks: only the kept entries travel
    def set_static_masks(self, params, masks):
        """`params`: model.parameters() in order (the flat buffer's layout); `masks`: one {0,1} tensor per
        parameter with dim != 1 in that order (the list `weight_prune` returns, methods.py:16-25).  Entries whose
        mask is 0 have an exactly-zero gradient on every rank and are left out of the collectives.  One host
        sync here (index construction); none per step.  `masks=None` returns to the dense transport."""
        if masks is None:
            self._kept = self._kept_pos = None
            return
        params = list(params)
        it = iter(masks)
        pieces, off, bounds = [], 0, [0]
        for p in params:
            n = p.numel()
            if p.dim() != 1:
                m = next(it)
                if tuple(m.shape) != tuple(p.shape):
                    raise ValueError("mask shape %s does not match parameter shape %s" % (tuple(m.shape), tuple(p.shape)))
                pieces.append(torch.nonzero(m.reshape(-1) != 0).flatten().to(p.device) + off)
            else:
                pieces.append(torch.arange(off, off + n, device=p.device))
            off += n
            bounds.append(off)
        kept = torch.cat(pieces)
        pos = torch.searchsorted(kept, torch.tensor(bounds, device=kept.device, dtype=kept.dtype)).tolist()
        self._kept, self._kept_pos, self._kept_total = kept, dict(zip(bounds, pos)), off

    @property
    def kept_fraction(self):
        return 1.0 if self._kept is None else self._kept.numel() / max(self._kept_total, 1)

    # ---- one bucket [lo, hi) of the flat buffer -> collective(s)
    def _send(self, flat, lo, hi):
        sparse = self._kept is not None
        if sparse:
            if lo not in self._kept_pos or hi not in self._kept_pos or flat.numel() != self._kept_total:
                raise RuntimeError("static-mask transport: bucket [%d, %d) does not lie on parameter boundaries" % (lo, hi))
            idx = self._kept[self._kept_pos[lo]:self._kept_pos[hi]]
            if idx.numel() == 0:
                return
            src = flat.index_select(0, idx)          # packed kept entries
        else:
            idx, src = None, flat[lo:hi]
        if self.transport == "fp16":
            # the fp16 SUM over the ranks must stay finite: |entry| <= 65504 / world.  Beyond that the value is clamped
            # and flagged (no host synchronisation): the step's result is wrong by construction and is skipped by whoever
            # owns the training loop (train.StepGuard reads `overflow`, all ranks together, and halves fp16_scale)
            lim = 65504.0 / max(self.world, 1)
            src = src * self.fp16_scale
            hit = (src.abs().amax() > lim).to(torch.int32).reshape(1)
            if self.overflow is None or self.overflow.device != hit.device:
                self.overflow = torch.zeros(1, dtype=torch.int32, device=hit.device)
            self.overflow.bitwise_or_(hit)
            src = src.clamp_(-lim, lim).to(torch.float16)
        elif not sparse:
            src = None                               # fp32 dense: all-reduce the slice in place
        if src is None:
            for off in range(lo, hi, self.chunk):
                end = min(off + self.chunk, hi)
                self._pending.append((dist.all_reduce(flat[off:end], op=dist.ReduceOp.SUM, async_op=True), None))
                self.collectives += 1
            self.bytes_reduced += (hi - lo) * flat.element_size()
            return
        works = []
        for off in range(0, src.numel(), self.chunk):
            works.append(dist.all_reduce(src[off:off + self.chunk], op=dist.ReduceOp.SUM, async_op=True))
            self.collectives += 1
        self.bytes_reduced += src.numel() * src.element_size()
        inv = 1.0 / self.fp16_scale if self.transport == "fp16" else 1.0

        def done(src=src, idx=idx, lo=lo, hi=hi, inv=inv):
            if src.is_cuda:
                # `src` was allocated in the stream context the bucket was launched from (the engine's second stream) and is
                # read here on the stream that waited for the collective: tell the caching allocator, or the block could be
                # handed out again on the second stream while this copy is still queued
                src.record_stream(torch.cuda.current_stream(src.device))
            vals = src.to(torch.float32) * inv if self.transport == "fp16" else src
            if idx is not None:
                flat.index_copy_(0, idx, vals)
            else:
                flat[lo:hi].copy_(vals)
        for w in works[:-1]:
            self._pending.append((w, None))
        self._pending.append((works[-1], done))

    # ---- overlapped path: the engine reports gradient slices as they become final (tail first)
    def ready(self, flat, lo, hi, fence=None):
        """`flat[lo:hi]` is final (all kernels that write it are enqueued on the current stream -- or, with `fence`,
        will be visible inside the stream context `fence()` returns: the engine's two-stream backward passes one, and it
        is entered only when a bucket is really launched).
        Slices arrive in descending, contiguous order; a collective is launched whenever the
        accumulated slice reaches `bucket_elems` (of travelling elements)."""
        if not self.active:
            return
        if self._hi is None:
            self._lo, self._hi = lo, hi
        else:
            if hi != self._lo:
                raise RuntimeError("gradient slices must arrive contiguously from the tail: got [%d, %d) after [%d, ...)"
                                   % (lo, hi, self._lo))
            self._lo = lo
        n = self._hi - self._lo
        if self._kept is not None and self._lo in self._kept_pos and self._hi in self._kept_pos:
            n = self._kept_pos[self._hi] - self._kept_pos[self._lo]
        if n >= self.bucket:
            if fence is not None:
                with fence():
                    self._launch(flat)
            else:
                self._launch(flat)

    def _launch(self, flat):
        if self._hi is None or self._hi == self._lo:
            return
        self._send(flat, self._lo, self._hi)
        self._sent_lo = self._lo
        self._hi = self._lo          # empty slice; the next ready() must end here

    def finish(self, flat):
        """End of backward: send what is left, wait for every collective, average."""
        if not self.active:
            return flat
        if self._hi is None:                      # nothing was reported piecewise: whole buffer at once
            self._lo, self._hi = 0, flat.numel()
        elif self._lo != 0:
            raise RuntimeError("gradient slices [0, %d) were never reported" % self._lo)
        self._launch(flat)
        for w, cb in self._pending:
            w.wait()
            if cb is not None:
                cb()
        self._pending = []
        self._lo = self._hi = self._sent_lo = None
        if not self.prescaled:
            flat.div_(self.world)
        return flat

    def transport_overflowed(self, reset=True):
        """True when a bucket of the fp16 transport had to clamp since the last call (one host sync; train.StepGuard
        reads the device flag without one)."""
        if self.overflow is None:
            return False
        hit = bool(int(self.overflow.item()))
        if hit and reset:
            self.overflow.zero_()
        return hit

    def reduce_flat(self, flat):
        if not self.active:
            return flat
        self._lo = self._hi = None
        return self.finish(flat)

    def reduce_params(self, params):
        grads = [p.grad for p in params if p.grad is not None]
        if not self.active or not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        kept, self._kept = self._kept, None      # separate gradient tensors: dense transport ...
        pre, self.prescaled = self.prescaled, False   # ... and nobody divided them by the world size yet
        try:
            self.reduce_flat(flat)
        finally:
            self._kept, self.prescaled = kept, pre
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()


def attach(model, reducer=None, masks=None):
    """Hook the reducer behind the engine's backward: p.grad views see the averaged values.
    `masks`: static weight masks (the list handed to `model.set_masks`) -> packed sparse transport."""
    reducer = reducer or GradReducer()
    if masks is not None:
        reducer.set_static_masks(model.parameters(), masks)
    def ready(flat, lo, hi, fence=None):        # called per final tail slice during backward
        reducer.ready(flat, lo, hi, fence)
    ready.takes_fence = True
    model._grad_ready_hook = ready
    model._grad_hook = reducer.finish           # called once at the end of backward
    # averaging rides on the kernels' 1 / grad_scale factor (engine.py backward): no division pass after the all-reduce
    model._grad_div = float(reducer.world) if reducer.active else 1.0
    reducer.prescaled = True
    model._grad_reducer = reducer               # train.StepGuard folds the transport's overflow flag into its decision
    return reducer


def broadcast_masks(masks, src=0):
    """Rank `src` computed the masks (ranking stays single-GPU); everyone else receives them."""
    if dist.is_initialized() and (dist.get_world_size() > 1 or rehearsal()):
        for m in masks:
            dist.broadcast(m, src=src)
    return masks


def broadcast_parameters(model, src=0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=src)
        if hasattr(model, "invalidate_packed"):
            model.invalidate_packed()


def sync_buffers(model):
    """Average the floating-point buffers (BatchNorm running_mean / running_var) over the ranks: they are
    updated from rank-local batch statistics, and rank 0's copy is what `save_weights` / eval would otherwise use.
    One flat all-reduce; call before a checkpoint or an evaluation (train.py does, once per epoch)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    bufs = [b for n, b in model.named_buffers() if b.is_floating_point() and not n.endswith(".mask")]
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1).float() for b in bufs])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    off = 0
    for b in bufs:
        b.copy_(flat[off:off + b.numel()].view_as(b).to(b.dtype))
        off += b.numel()


def all_ranks_ok(ok, device=None):
    """Collective AND of a per-rank condition (e.g. "my loss and gradients are finite"), so that every rank takes
    the same branch -- a rank that raised alone would leave its peers blocked in the next all-reduce."""
    if not (dist.is_initialized() and (dist.get_world_size() > 1 or rehearsal())):
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else None)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))
