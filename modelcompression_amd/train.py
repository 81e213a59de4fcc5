"""Retrain entry point: mirror of the reference's src/train.py (YOLOv2Train.train, 67-290).

Same signature and the same sequence -- build Darknet from the cfg, load .weights, SGD with
lr 1e-5 / momentum / weight_decay = decay*BATCH_SIZE (train.py:144-147), optional pruning +
set_masks (167-174), epochs of forward / RegionLoss / zero_grad / backward / step (214-235),
per-epoch prune_rate + are_masks_consistent and a checkpoint every 5th epoch when pruning
(247-252) -- with the model running on the HIP engine.  Differences, all additive:
  * launched under torch.distributed (one process per GPU) it shards each batch across the
    ranks and averages gradients with one flat all-reduce per step (RCCL over xGMI);
  * masks are computed on rank 0 and broadcast;
  * `torch.autograd.detect_anomaly()` / pdb of the reference are not entered (a non-finite loss
    raises instead);
  * when the image list is missing, a seeded synthetic detection set of the same shapes is
    used so the entry point can be exercised without VOC on disk;
  * MAX_EPOCHS can be given to stop early (the reference hard-codes 135).
"""
import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL: read when HIP initialises (dp.init_from_env)

import time  # noqa: E402

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.optim as optim  # noqa: E402

from . import dp  # noqa: E402
from .data import VOCList, SyntheticDetection  # noqa: E402
from .nets import Darknet, parse_cfg  # noqa: E402
from .pruning.weightPruning.methods import quick_filter_prune, weight_prune  # noqa: E402
from .pruning.weightPruning.utils import prune_rate, are_masks_consistent  # noqa: E402


def logging(message):
    print('%s %s' % (time.strftime("%Y-%m-%d %H:%M:%S", time.localtime()), message))


def file_lines(thefilepath):
    with open(thefilepath, 'rb') as f:
        return sum(buf.count(b'\n') for buf in iter(lambda: f.read(1 << 20), b''))


class StepGuard:
    """The overflow / non-finite policy of a training step WITHOUT a host synchronisation in the step.

    The engine keeps gradients as fp16 x grad_scale and saturates at +-65504 (a device flag per engine records it;
    RegionLoss's exp terms can push |dL/dlogit| x 256 beyond that), and a non-finite loss must not reach the weights.
    Round 2 read both on the host every step (`isfinite(loss)`, `grad_overflowed().item()`, an `isfinite` scan of the
    202 MB gradient buffer and two collectives).  Here:
      * `decide(loss)` (after backward) folds the engines' flags and `~isfinite(loss)` into ONE device scalar, takes the
        maximum over the ranks (one 4-byte all-reduce on the stream, so every rank takes the same branch) and hands it
        to the fused SGD kernel as `found_inf`: a flagged step's update is skipped ON THE DEVICE;
      * the host reads that scalar one step LATE (pinned copy + event recorded here, consumed by the next call, when
        the following step is already enqueued): engine overflow -> grad_scale halved for the steps after it, overflow
        of the fp16 all-reduce transport (dp.GradReducer.overflow) -> its fp16_scale halved, non-finite loss ->
        FloatingPointError (the reference enters pdb on a NaN loss, train.py:226-231).
    (The gradient buffer needs no scan of its own: every fp16-writing backward kernel clamps and flags, the fp32
    finish passes only scale finite values, so a finite loss gives finite gradients.)"""

    def __init__(self, model, optimizer, dev, log=None):
        self.model, self.opt, self.dev, self.log = model, optimizer, dev, log
        self.flags = torch.zeros(3, dtype=torch.float32, device=dev)   # [engine overflow, non-finite loss, transport overflow]
        self.found = torch.zeros(1, dtype=torch.float32, device=dev)
        optimizer.found_inf = self.found          # torch's fused SGD skips the whole update when this is EXACTLY 1.0
        # momentum buffers exist (zero) from the start: a skipped FIRST step would otherwise leave them uninitialised
        # (torch creates them with empty_like on the first call); zero buffers give the same first update, buf = grad
        for group in optimizer.param_groups:
            if group.get("momentum", 0):
                for p in group["params"]:
                    optimizer.state[p].setdefault("momentum_buffer", torch.zeros_like(p))
        self._host = torch.zeros(3, dtype=torch.float32)
        if dev.type == "cuda":
            self._host = self._host.pin_memory()
        self._event, self._pending = None, False
        self.skipped = 0
        # scales the pending step RAN with: its flag is read one step late, when the next step -- enqueued with the same
        # scales -- has usually overflowed too; a flag only halves a scale that is still the one its step used, so one
        # overflow event costs one halving, not two.  `clean` counts steps since the last skip: after
        # GROWTH_INTERVAL of them a halved scale doubles again (up to where it started), as torch.amp.GradScaler does --
        # otherwise a long run only ever loses gradient bits.
        self._pending_scales = (None, None)
        self._scale0 = float(getattr(model, "grad_scale", 1.0))
        self._tscale0 = None
        self.clean = 0
        self._collective = dist.is_initialized() and (dist.get_world_size() > 1 or dp.rehearsal())

    def consume(self):
        """Act on the flag of the step BEFORE the one just enqueued (no-op when there is none)."""
        if not self._pending:
            return 0
        if self._event is not None:
            self._event.synchronize()             # that step is long done unless the host runs a whole step ahead
        over, bad, tover = (bool(v) for v in self._host.tolist())
        self._pending = False
        if bad:
            raise FloatingPointError("non-finite training loss")
        ran_gs, ran_ts = self._pending_scales
        red = getattr(self.model, "_grad_reducer", None)
        if over or tover:
            self.skipped += 1
            self.clean = 0
        else:
            self.clean += 1
        if over and (ran_gs is None or ran_gs == self.model.grad_scale):
            self.model.grad_scale = max(1.0, self.model.grad_scale / 2.0)
            if self.log:
                self.log('gradient overflow in fp16 storage: step skipped, grad_scale -> %g' % self.model.grad_scale)
        if tover and (ran_ts is None or ran_ts == red.fp16_scale):
            red.fp16_scale = max(1.0 / 65536.0, red.fp16_scale / 2.0)
            if self.log:
                self.log('gradient overflow in the fp16 all-reduce transport: step skipped, fp16_scale -> %g' % red.fp16_scale)
        if self.clean >= self.GROWTH_INTERVAL:
            self.clean = 0
            if self.model.grad_scale < self._scale0:
                self.model.grad_scale = min(self._scale0, self.model.grad_scale * 2.0)
                if self.log:
                    self.log('%d clean steps: grad_scale -> %g' % (self.GROWTH_INTERVAL, self.model.grad_scale))
            if red is not None and self._tscale0 is not None and red.fp16_scale < self._tscale0:
                red.fp16_scale = min(self._tscale0, red.fp16_scale * 2.0)
        return int(over) + 2 * int(bad) + 4 * int(tover)

    GROWTH_INTERVAL = 2000

    def decide(self, loss):
        """Call between backward() and optimizer.step()."""
        red = getattr(self.model, "_grad_reducer", None)
        # the scales the step that just ran its backward USED (consume() below may change them for the next one)
        ran = (float(getattr(self.model, "grad_scale", 1.0)), red.fp16_scale if red is not None else None)
        if red is not None and self._tscale0 is None:
            self._tscale0 = red.fp16_scale
        self.consume()
        self._pending_scales = ran
        flags = [eng.overflow for eng in getattr(self.model, "_engines", {}).values()]
        tflag = red.overflow if red is not None else None
        if (self.dev.type == "cuda" and len(flags) <= 8 and loss.dtype == torch.float32 and loss.numel() == 1
                and all(f.is_cuda and f.dtype == torch.int32 for f in flags + ([tflag] if tflag is not None else []))):
            # one launch (mcamd_step_flags) instead of the dozen torch micro-kernels below: 9.42 -> 9.3x ms per real step
            from . import ops
            ops.step_flags(flags, loss.detach().reshape(1), tflag, self.flags, None if self._collective else self.found)
            if self._collective:
                dist.all_reduce(self.flags, op=dist.ReduceOp.MAX)
                ops.step_flags([], None, None, self.flags, self.found)
            self._host.copy_(self.flags, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()
            self._pending = True
            return
        zero = torch.zeros((), dtype=torch.float32, device=self.dev)
        over = torch.stack([f.reshape(()) for f in flags]).sum().clamp(max=1).to(torch.float32) if flags else zero
        bad = (~torch.isfinite(loss.detach())).to(torch.float32).reshape(())
        tover = tflag.reshape(()).clamp(max=1).to(torch.float32) if tflag is not None else zero
        self.flags.copy_(torch.stack((over, bad, tover)))
        for f in flags + ([tflag] if tflag is not None else []):
            f.zero_()
        if self._collective:
            dist.all_reduce(self.flags, op=dist.ReduceOp.MAX)
        self.found.copy_(self.flags.amax().clamp(max=1).reshape(1))      # exactly 1.0: the fused SGD tests `== 1`
        self._host.copy_(self.flags, non_blocking=True)
        if self.dev.type == "cuda":
            self._event = torch.cuda.Event()
            self._event.record()
        self._pending = True

    def finish(self):
        """End of an epoch / run: the last step's flag."""
        return self.consume()


class YOLOv2Train():

    def __init__(self):
        self.model = ''
        self.optimizer = ''
        self.trainlist = ''
        self.testlist = ''
        self.init_width = ''
        self.init_height = ''
        self.batch_size = ''

    def train(self, PASCAL_DIR, PASCAL_TRAIN, PASCAL_VALID, TRAIN_LOGDIR, VAL_LOGDIR, VAL_OUTPUTDIR_PKL, VAL_PREFIX,
              MODEL_CFG, MODEL_WEIGHT,
              BATCH_SIZE, SAVE_INTERNAL,
              LOGGER='', DEBUG_EPOCHS=-1, verbose=0, pruning_perc=0., pruning_method="weight",
              MAX_EPOCHS=135, SYNTHETIC_SAMPLES=256, EVAL=False):
        rank, world = dp.init_from_env()
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)

        # Step 1 - model
        net_options = parse_cfg(MODEL_CFG)[0]
        self.model = Darknet(MODEL_CFG)
        if MODEL_WEIGHT and os.path.exists(MODEL_WEIGHT):
            self.model.load_weights(MODEL_WEIGHT)
        else:
            from .synthetic import init_synthetic
            init_synthetic(self.model, seed=0)
            logging('weights file %r not found: seeded synthetic initialisation' % (MODEL_WEIGHT,))
        self.model = self.model.to(dev)
        dp.broadcast_parameters(self.model, src=0)

        # Step 2 - dataset
        self.trainlist, self.testlist = PASCAL_TRAIN, PASCAL_VALID
        self.init_width, self.init_height = self.model.width, self.model.height
        if PASCAL_TRAIN and os.path.exists(PASCAL_TRAIN):
            nsamples = file_lines(self.trainlist)
            dataset = VOCList(self.trainlist, shape=(self.init_width, self.init_height), train=True)
        else:
            nsamples = SYNTHETIC_SAMPLES
            dataset = SyntheticDetection(nsamples, shape=(self.init_width, self.init_height))
            logging('train list %r not found: synthetic detection set of %d samples' % (PASCAL_TRAIN, nsamples))
        if TRAIN_LOGDIR and rank == 0 and not os.path.exists(TRAIN_LOGDIR):
            os.makedirs(TRAIN_LOGDIR, exist_ok=True)

        # Step 3 - training parameters (train.py:113-147)
        self.batch_size = BATCH_SIZE
        momentum = float(net_options['momentum'])
        decay = float(net_options['decay'])
        region_loss = self.model.loss
        region_loss.seen = self.model.seen
        init_epoch = int(self.model.seen / nsamples)
        LR = 0.00001
        optimizer = optim.SGD(self.model.parameters(), lr=LR, momentum=momentum, dampening=0,
                              weight_decay=decay * self.batch_size, fused=True)   # same rule (train.py:144-147), one kernel
        self.optimizer = optimizer
        reducer = None

        masks = None
        if pruning_perc > 0:
            if pruning_method == "filter":
                masks = quick_filter_prune(self.model, pruning_perc)
            else:
                masks = weight_prune(self.model, pruning_perc)
            dp.broadcast_masks(masks, src=0)
            self.model.set_masks(masks)
        if world > 1:
            # static weight masks: only the kept gradient entries travel (dp.py); filter masks keep the dense transport
            reducer = dp.attach(self.model, masks=masks if (masks is not None and pruning_method != "filter") else None)
        if masks is not None:
            p_rate = prune_rate(self.model, rank == 0)
            if rank == 0:
                print(' %s=pruned: %s' % (pruning_method, p_rate))

        per_rank = max(1, self.batch_size // world)
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, world, rank, shuffle=True) if world > 1 else None
        loader = torch.utils.data.DataLoader(dataset, batch_size=per_rank, shuffle=(sampler is None), sampler=sampler,
                                             num_workers=4 if isinstance(dataset, VOCList) else 0, pin_memory=True,
                                             drop_last=True)
        guard = StepGuard(self.model, optimizer, dev, log=logging if rank == 0 else None)
        epoch = init_epoch
        for epoch in range(init_epoch, min(MAX_EPOCHS, 135)):
            if sampler is not None:
                sampler.set_epoch(epoch)
            if rank == 0:
                print(' ---------------------------- EPOCH : ', epoch, ' (LR : ', LR, ') ---------------------------------- ')
            self.model.train()
            # the loss is accumulated on the device and read once per epoch; the skip policy lives in StepGuard
            train_loss_total, t0, steps_here = torch.zeros((), dtype=torch.float32, device=dev), time.time(), 0
            skipped_before = guard.skipped
            for batch_idx, (data, target) in enumerate(loader):
                if DEBUG_EPOCHS > -1 and batch_idx > DEBUG_EPOCHS:
                    break
                data = data.to(dev, non_blocking=True)
                target = target.float().to(dev, non_blocking=True)
                output = self.model(data)
                region_loss.seen = region_loss.seen + data.size(0) * world
                train_loss = region_loss(output, target)
                train_loss_total += train_loss.detach()
                optimizer.zero_grad()
                train_loss.backward()
                # no host synchronisation here: a saturated / non-finite step is skipped on the device, by all ranks
                # together, and the host learns of it one step late (StepGuard)
                guard.decide(train_loss)
                optimizer.step()
                steps_here += 1
                if verbose and rank == 0:
                    print(' - loss : ', float(train_loss.detach()))
            guard.finish()
            torch.cuda.synchronize()
            seen_here = (steps_here - (guard.skipped - skipped_before)) * per_rank * world
            train_loss_total = float(train_loss_total)
            if rank == 0:
                logging('training with %f samples/s, mean loss %.4f' % (seen_here / max(time.time() - t0, 1e-9),
                                                                         train_loss_total / max(len(loader), 1)))
            dp.sync_buffers(self.model)     # BatchNorm running statistics are rank-local: average before checkpoint / eval
            if pruning_perc > 0 and rank == 0:
                print(' pruned: %s' % prune_rate(self.model, False))
                print(' pruned weights consistent after retraining: %s ' % are_masks_consistent(self.model, masks))
                if (epoch + 1) % 5 == 0 and TRAIN_LOGDIR:
                    name = '%s/%s-pruned-%s-retrained_%06d.weights' % (TRAIN_LOGDIR, pruning_method, pruning_perc, epoch + 1)
                    logging('save weights to %s' % name)
                    self.model.save_weights(name)
            if LOGGER != '' and rank == 0:
                LOGGER.save_value('Total Loss', 'Train Loss', epoch + 1, train_loss_total / max(len(loader), 1))
            self.model.seen = (epoch + 1) * nsamples
            if EVAL and rank == 0:
                from .predict import PASCALVOCEval
                PASCALVOCEval(self.model, MODEL_CFG, MODEL_WEIGHT, region_loss, PASCAL_DIR, PASCAL_VALID, VAL_LOGDIR,
                              VAL_PREFIX, VAL_OUTPUTDIR_PKL, LOGGER, epoch).predict(BATCH_SIZE)
        if TRAIN_LOGDIR and rank == 0:
            name = '%s/%s-pruned-%s-retrained-final_%06d.weights' % (TRAIN_LOGDIR, pruning_method, pruning_perc, epoch + 1)
            logging('save weights to %s' % name)
            self.model.save_weights(name)
        return self.model

    def adjust_learning_rate(self, optimizer, batch, learning_rate, steps, scales, batch_size):
        """train.py:277-290 (never called by the reference's loop either)."""
        lr = learning_rate
        for i in range(len(steps)):
            scale = scales[i] if i < len(scales) else 1
            if batch >= steps[i]:
                lr = lr * scale
                if batch == steps[i]:
                    break
            else:
                break
        for param_group in optimizer.param_groups:
            param_group['lr'] = lr / batch_size
        return lr
