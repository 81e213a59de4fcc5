"""YOLOv2 region loss of the reference (src/nets.py:282-635), kept on the device.

The reference builds its targets with nested Python loops on the CPU (a D2H copy of every
predicted box, ~10 host/device hops per step).  This version computes the same quantities
with batched tensor ops on whatever device `output` lives on -- no host round trip and no host
synchronisation at all (tests run it under torch.cuda.set_sync_debug_mode("error")) -- and
keeps the reference's arithmetic, including its quirks:
  * w, h = exp(output) AND the boxes used for IoU use exp(w) * anchor (double exp,
    nets.py:511-512, 546-547);
  * tw, th = gw / anchor (no log, nets.py:429-430);
  * conf_mask starts at noobject_scale, is zeroed where the best IoU with any ground-truth
    box exceeds `thresh`, set to object_scale at assigned cells, and enters the loss as its
    square root (nets.py:296, 347, 419, 582);
  * tconf = IoU(ground truth, predicted box) at the assigned cell (nets.py:434-435);
  * ground-truth rows end at the first box whose x is 0 (nets.py:324, 374); when two boxes
    land in one cell/anchor the later one wins;
  * loss = (sum of halved squared errors + summed cross-entropy) / nB (nets.py:594-600).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

MAX_BBOX = 50

_ANCHORS = {}


def _anchor_tensors(anchors_list, anchors_cell, dev):
    """(anchor widths, anchor heights) as fp32 vectors on `dev`, uploaded once per device (a host->device copy of a
    Python list synchronises; the loss runs every step)."""
    key = (tuple(float(a) for a in anchors_list), int(anchors_cell), str(dev))
    hit = _ANCHORS.get(key)
    if hit is None:
        step = int(len(anchors_list) / anchors_cell)
        hit = (torch.tensor([anchors_list[step * n] for n in range(anchors_cell)], dtype=torch.float32).to(dev),
               torch.tensor([anchors_list[step * n + 1] for n in range(anchors_cell)], dtype=torch.float32).to(dev))
        _ANCHORS[key] = hit
    return hit


def _iou_cwh(x1, y1, w1, h1, x2, y2, w2, h2):
    """IoU of centre/size boxes, the reference's bbox_ious(x1y1x2y2=False) (nets2_utils.py:100-131)."""
    mx = torch.min(x1 - w1 / 2.0, x2 - w2 / 2.0)
    Mx = torch.max(x1 + w1 / 2.0, x2 + w2 / 2.0)
    my = torch.min(y1 - h1 / 2.0, y2 - h2 / 2.0)
    My = torch.max(y1 + h1 / 2.0, y2 + h2 / 2.0)
    cw = w1 + w2 - (Mx - mx)
    ch = h1 + h2 - (My - my)
    carea = cw * ch
    carea = torch.where((cw <= 0) | (ch <= 0), torch.zeros_like(carea), carea)
    uarea = w1 * h1 + w2 * h2 - carea
    return carea / uarea


def build_targets(pred_boxes, target, anchors_list, anchors_cell, num_classes, nH, nW, noobject_scale, object_scale,
                  sil_thresh, seen):
    """Same return tuple as nets.py:282-440.  pred_boxes: [nB*nA*nH*nW, 4] (x, y, w, h in grid
    units), target: [nB, 250] rows of (cls, x, y, w, h) * 50, normalised to [0, 1].

    Fully batched over images AND ground-truth boxes, on pred_boxes' device, with no host synchronisation (the
    reference copies the predictions to the host and runs nested Python loops, nets.py:322-347, 368-438): nGT and
    nCorrect come back as 0-dim tensors instead of Python ints (the reference only prints them)."""
    dev = pred_boxes.device
    nB, nA = target.size(0), anchors_cell
    anchor_step = int(len(anchors_list) / anchors_cell)
    target = target.to(dev).float().view(nB, MAX_BBOX, 5)
    pb = pred_boxes.view(nB, nA * nH * nW, 4)
    valid = (target[:, :, 1] != 0).long().cumprod(1).bool()          # [nB, T]: rows end at the first x == 0 (nets.py:324)

    gx, gy = target[:, :, 1] * nW, target[:, :, 2] * nH               # [nB, T]
    gw, gh = target[:, :, 3] * nW, target[:, :, 4] * nH

    # STEP 1 (nets.py:322-347): silence the no-object penalty where a prediction already overlaps some ground truth
    iou_all = _iou_cwh(pb[:, :, None, 0], pb[:, :, None, 1], pb[:, :, None, 2], pb[:, :, None, 3],
                       gx[:, None, :], gy[:, None, :], gw[:, None, :], gh[:, None, :])                # [nB, N, T]
    best = torch.where(valid[:, None, :], iou_all, torch.zeros_like(iou_all)).amax(2).clamp_min(0.0)
    conf_mask = torch.where(best > sil_thresh, torch.zeros_like(best), torch.full_like(best, noobject_scale))

    # STEP 3 (nets.py:368-438): every ground-truth box goes to its best anchor (shape IoU) in its cell
    aw, ah = _anchor_tensors(anchors_list, nA, dev)
    z = torch.zeros(nB, MAX_BBOX, 1, device=dev)
    a_iou = _iou_cwh(z, z, aw.view(1, 1, nA), ah.view(1, 1, nA), z, z, gw[:, :, None], gh[:, :, None])   # [nB, T, nA]
    a_iou = torch.where(torch.isnan(a_iou), torch.zeros_like(a_iou), a_iou)
    best_n = a_iou.argmax(2)                                        # first maximum, as the strict '>' loop picks
    best_n = torch.where(a_iou.amax(2) > 0, best_n, torch.full_like(best_n, nA - 1))                   # best_n = -1 case
    gi, gj = gx.long(), gy.long()
    cell = (best_n * nH + gj.clamp(0, nH - 1)) * nW + gi.clamp(0, nW - 1)                              # [nB, T] in [0, N)
    # two boxes in one cell / anchor: the later one overwrites the earlier (sequential loop) -> only the last
    # valid box of every cell writes
    later = torch.arange(MAX_BBOX, device=dev)
    same = (cell[:, :, None] == cell[:, None, :]) & valid[:, None, :] & (later[None, None, :] > later[None, :, None])
    writer = valid & ~same.any(2)
    p = torch.gather(pb, 1, cell[:, :, None].expand(-1, -1, 4))                                        # predicted box there
    iou_gt = _iou_cwh(gx, gy, gw, gh, p[..., 0], p[..., 1], p[..., 2], p[..., 3])                        # [nB, T]
    nGT = valid.sum()
    nCorrect = (valid & (iou_gt > 0.5)).sum()

    N = nA * nH * nW
    sink = torch.full_like(cell, N)                                  # non-writers scatter into a spare column
    idx = torch.where(writer, cell, sink)

    def scatter(base, values):
        ext = torch.cat((base, base.new_zeros(nB, 1)), 1)
        ext.scatter_(1, idx, values)
        return ext[:, :N].contiguous()
    zeros = torch.zeros(nB, N, device=dev)
    ones_t = torch.ones(nB, MAX_BBOX, device=dev)
    coord_mask = scatter(zeros, ones_t)
    cls_mask = coord_mask.clone()
    conf_mask = scatter(conf_mask, ones_t * object_scale)
    tx = scatter(zeros, gx - gi.float())
    ty = scatter(zeros, gy - gj.float())
    tw = scatter(zeros, gw / aw[best_n])
    th = scatter(zeros, gh / ah[best_n])
    tconf = scatter(zeros, iou_gt)
    tcls = scatter(zeros, target[:, :, 0])
    shape = (nB, nA, nH, nW)
    return (nGT, nCorrect, coord_mask.view(shape), conf_mask.view(shape), cls_mask.view(shape), tx.view(shape),
            ty.view(shape), tw.view(shape), th.view(shape), tconf.view(shape), tcls.view(shape))


class _RegionLossFn(torch.autograd.Function):
    """loss = mcamd_region_loss(output); the kernel returns d(loss)/d(output) with it (csrc/region_loss.hip)."""

    @staticmethod
    def forward(ctx, output, target, mod):
        from . import ops
        loss, grad, _ = ops.region_loss(output.detach().contiguous().float(), target, mod.anchors, mod.num_anchors,
                                        mod.num_classes, mod.coord_scale, mod.noobject_scale, mod.object_scale, mod.class_scale,
                                        mod.thresh)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, go):
        (grad,) = ctx.saved_tensors
        return grad * go, None, None


class RegionLoss(nn.Module):
    def __init__(self, num_classes=20,
                 anchor_list=[1.3221, 1.73145, 3.19275, 4.00944, 5.05587, 8.09892, 9.47112, 4.84053, 11.2364, 10.0071],
                 anchors_cell=5):
        super(RegionLoss, self).__init__()
        self.num_classes = num_classes
        self.anchors = anchor_list
        self.num_anchors = anchors_cell
        self.anchor_step = int(len(anchor_list) / anchors_cell)
        self.coord_scale = 1
        self.noobject_scale = 1
        self.object_scale = 1
        self.class_scale = 1
        self.thresh = 0.6
        self.seen = 0
        self.fused = True          # CUDA tensors: csrc/region_loss.hip (False: the batched torch restatement on the device)

    def forward(self, output, target, verbose=0):
        if output.is_cuda and self.fused and self.num_anchors <= 8 and target.numel() == output.size(0) * MAX_BBOX * 5:
            # one HIP pass: loss and its gradient (the batched torch restatement below is ~60 small launches forward and as
            # many backward: 1.7 ms of a B=64 training step; it stays as the path for CPU tensors and as the kernel's check)
            return _RegionLossFn.apply(output, target, self)
        nB, nA, nC = output.size(0), self.num_anchors, self.num_classes
        nH, nW = output.size(2), output.size(3)
        dev = output.device
        out = output.view(nB, nA, 5 + nC, nH, nW)
        x = torch.sigmoid(out[:, :, 0])
        y = torch.sigmoid(out[:, :, 1])
        w = torch.exp(out[:, :, 2])
        h = torch.exp(out[:, :, 3])
        conf = torch.sigmoid(out[:, :, 4])
        cls = out[:, :, 5:].reshape(nB * nA, nC, nH * nW).transpose(1, 2).contiguous().view(nB * nA * nH * nW, nC)

        with torch.no_grad():
            grid_x = torch.arange(nW, device=dev, dtype=torch.float32).view(1, 1, 1, nW)
            grid_y = torch.arange(nH, device=dev, dtype=torch.float32).view(1, 1, nH, 1)
            step = int(self.anchor_step)
            anchor_w, anchor_h = (a.view(1, nA, 1, 1) for a in _anchor_tensors(self.anchors, nA, dev))
            pred_boxes = torch.stack((x.detach() + grid_x, y.detach() + grid_y,
                                      torch.exp(w.detach()) * anchor_w, torch.exp(h.detach()) * anchor_h), -1)
            nGT, nCorrect, coord_mask, conf_mask, cls_mask, tx, ty, tw, th, tconf, tcls = build_targets(
                pred_boxes.view(-1, 4), target, self.anchors, nA, nC, nH, nW, self.noobject_scale, self.object_scale,
                self.thresh, self.seen)
            cls_w = (cls_mask == 1).view(-1).float()
            conf_mask = conf_mask.sqrt()

        def sse(a, b):
            return F.mse_loss(a, b, reduction='sum')

        loss_x = self.coord_scale * sse(x * coord_mask, tx * coord_mask) / 2.0
        loss_y = self.coord_scale * sse(y * coord_mask, ty * coord_mask) / 2.0
        loss_w = self.coord_scale * sse(w * coord_mask, tw * coord_mask) / 2.0
        loss_h = self.coord_scale * sse(h * coord_mask, th * coord_mask) / 2.0
        loss_conf = sse(conf * conf_mask, tconf * conf_mask) / 2.0
        # cross-entropy summed over the assigned cells (nets.py:590-599 indexes them out with a boolean mask, a
        # dynamic shape = a host sync; weighting every cell by its 0/1 mask gives the same sum)
        loss_cls = self.class_scale * (F.cross_entropy(cls, tcls.view(-1).long(), reduction='none') * cls_w).sum()
        return (loss_x + loss_y + loss_w + loss_h + loss_conf + loss_cls) / nB
