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
import time

import torch
import torch.optim as optim

from . import dp
from .data import VOCList, SyntheticDetection
from .nets import Darknet, parse_cfg
from .pruning.weightPruning.methods import quick_filter_prune, weight_prune
from .pruning.weightPruning.utils import prune_rate, are_masks_consistent


def logging(message):
    print('%s %s' % (time.strftime("%Y-%m-%d %H:%M:%S", time.localtime()), message))


def file_lines(thefilepath):
    with open(thefilepath, 'rb') as f:
        return sum(buf.count(b'\n') for buf in iter(lambda: f.read(1 << 20), b''))


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
        epoch = init_epoch
        for epoch in range(init_epoch, min(MAX_EPOCHS, 135)):
            if sampler is not None:
                sampler.set_epoch(epoch)
            if rank == 0:
                print(' ---------------------------- EPOCH : ', epoch, ' (LR : ', LR, ') ---------------------------------- ')
            self.model.train()
            train_loss_total, t0, seen_here, skipped_steps = 0.0, time.time(), 0, 0
            for batch_idx, (data, target) in enumerate(loader):
                if DEBUG_EPOCHS > -1 and batch_idx > DEBUG_EPOCHS:
                    break
                data = data.to(dev, non_blocking=True)
                target = target.float().to(dev, non_blocking=True)
                output = self.model(data)
                region_loss.seen = region_loss.seen + data.size(0) * world
                train_loss = region_loss(output, target)
                # every decision below is taken by all ranks together: a rank that raised or skipped alone would
                # leave its peers blocked in the next gradient all-reduce
                if not dp.all_ranks_ok(bool(torch.isfinite(train_loss)), dev):
                    raise FloatingPointError("non-finite training loss at epoch %d batch %d" % (epoch, batch_idx))
                train_loss_total += float(train_loss.detach())
                optimizer.zero_grad()
                train_loss.backward()
                # the engine keeps gradients as fp16 x grad_scale and saturates at +-65504 (RegionLoss's exp terms can
                # produce |dL/dlogit| x 256 beyond that): such a step is skipped and the scale halved
                flat = self.model._last_flat_grad
                good = not self.model.grad_overflowed() and (flat is None or bool(torch.isfinite(flat).all()))
                if not dp.all_ranks_ok(good, dev):
                    self.model.grad_scale = max(1.0, self.model.grad_scale / 2.0)
                    skipped_steps += 1
                    if rank == 0:
                        logging('gradient overflow in fp16 storage: step skipped, grad_scale -> %g' % self.model.grad_scale)
                    continue
                optimizer.step()
                seen_here += data.size(0) * world
                if verbose and rank == 0:
                    print(' - loss : ', float(train_loss.detach()))
            torch.cuda.synchronize()
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
