"""YOLOv2 region loss of the reference (src/nets.py:282-635), kept on the device.

The reference builds its targets with nested Python loops on the CPU (a D2H copy of every
predicted box, ~10 host/device hops per step).  This version computes the same quantities
with batched tensor ops on whatever device `output` lives on -- no host round trip -- and
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
    units), target: [nB, 250] rows of (cls, x, y, w, h) * 50, normalised to [0, 1]."""
    dev = pred_boxes.device
    nB, nA = target.size(0), anchors_cell
    anchor_step = int(len(anchors_list) / anchors_cell)
    target = target.to(dev).float().view(nB, MAX_BBOX, 5)
    pb = pred_boxes.view(nB, nA, nH, nW, 4)
    valid = (target[:, :, 1] != 0).long().cumprod(1).bool()          # stop at the first empty row

    coord_mask = torch.zeros(nB, nA, nH, nW, device=dev)
    conf_mask = torch.ones(nB, nA, nH, nW, device=dev) * noobject_scale
    cls_mask = torch.zeros(nB, nA, nH, nW, device=dev)
    tx, ty, tw, th, tconf, tcls = (torch.zeros(nB, nA, nH, nW, device=dev) for _ in range(6))

    gx, gy = target[:, :, 1] * nW, target[:, :, 2] * nH
    gw, gh = target[:, :, 3] * nW, target[:, :, 4] * nH

    # STEP 1: silence the no-object penalty where a prediction already overlaps some ground truth
    best = torch.zeros(nB, nA, nH, nW, device=dev)
    for t in range(MAX_BBOX):
        if not bool(valid[:, t].any()):
            break
        v = valid[:, t].view(nB, 1, 1, 1)
        iou = _iou_cwh(pb[..., 0], pb[..., 1], pb[..., 2], pb[..., 3],
                       gx[:, t].view(nB, 1, 1, 1), gy[:, t].view(nB, 1, 1, 1),
                       gw[:, t].view(nB, 1, 1, 1), gh[:, t].view(nB, 1, 1, 1))
        best = torch.where(v, torch.max(best, iou), best)
    conf_mask = torch.where(best > sil_thresh, torch.zeros_like(conf_mask), conf_mask)

    # STEP 3: assign every ground-truth box to its best anchor (shape IoU) in its cell
    aw = torch.tensor([anchors_list[anchor_step * n] for n in range(nA)], device=dev)
    ah = torch.tensor([anchors_list[anchor_step * n + 1] for n in range(nA)], device=dev)
    bidx = torch.arange(nB, device=dev)
    nGT = int(valid.sum().item())
    nCorrect = 0
    for t in range(MAX_BBOX):
        v = valid[:, t]
        if not bool(v.any()):
            break
        z = torch.zeros(nB, 1, device=dev)
        a_iou = _iou_cwh(z, z, aw.view(1, nA), ah.view(1, nA), z, z, gw[:, t].view(nB, 1), gh[:, t].view(nB, 1))
        a_iou = torch.where(torch.isnan(a_iou), torch.zeros_like(a_iou), a_iou)
        best_n = a_iou.argmax(1)                                   # first maximum, as the strict '>' loop picks
        best_n = torch.where(a_iou.max(1).values > 0, best_n, torch.full_like(best_n, nA - 1))  # best_n = -1 case
        gi, gj = gx[:, t].long(), gy[:, t].long()
        sel = bidx[v]
        b_, n_, j_, i_ = sel, best_n[v], gj[v].clamp(0, nH - 1), gi[v].clamp(0, nW - 1)
        coord_mask[b_, n_, j_, i_] = 1
        conf_mask[b_, n_, j_, i_] = object_scale
        cls_mask[b_, n_, j_, i_] = 1
        tx[b_, n_, j_, i_] = gx[v, t] - gi[v].float()
        ty[b_, n_, j_, i_] = gy[v, t] - gj[v].float()
        tw[b_, n_, j_, i_] = gw[v, t] / aw[n_]
        th[b_, n_, j_, i_] = gh[v, t] / ah[n_]
        p = pb[b_, n_, j_, i_]
        iou = _iou_cwh(gx[v, t], gy[v, t], gw[v, t], gh[v, t], p[:, 0], p[:, 1], p[:, 2], p[:, 3])
        tconf[b_, n_, j_, i_] = iou
        tcls[b_, n_, j_, i_] = target[v, t, 0]
        nCorrect += int((iou > 0.5).sum().item())
    return nGT, nCorrect, coord_mask, conf_mask, cls_mask, tx, ty, tw, th, tconf, tcls


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

    def forward(self, output, target, verbose=0):
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
            anchor_w = torch.tensor(self.anchors[0::step], device=dev).view(1, nA, 1, 1)
            anchor_h = torch.tensor(self.anchors[1::step], device=dev).view(1, nA, 1, 1)
            pred_boxes = torch.stack((x.detach() + grid_x, y.detach() + grid_y,
                                      torch.exp(w.detach()) * anchor_w, torch.exp(h.detach()) * anchor_h), -1)
            nGT, nCorrect, coord_mask, conf_mask, cls_mask, tx, ty, tw, th, tconf, tcls = build_targets(
                pred_boxes.view(-1, 4), target, self.anchors, nA, nC, nH, nW, self.noobject_scale, self.object_scale,
                self.thresh, self.seen)
            cls_sel = (cls_mask == 1).view(-1)
            tcls_sel = tcls.view(-1)[cls_sel].long()
            conf_mask = conf_mask.sqrt()

        def sse(a, b):
            return F.mse_loss(a, b, reduction='sum')

        loss_x = self.coord_scale * sse(x * coord_mask, tx * coord_mask) / 2.0
        loss_y = self.coord_scale * sse(y * coord_mask, ty * coord_mask) / 2.0
        loss_w = self.coord_scale * sse(w * coord_mask, tw * coord_mask) / 2.0
        loss_h = self.coord_scale * sse(h * coord_mask, th * coord_mask) / 2.0
        loss_conf = sse(conf * conf_mask, tconf * conf_mask) / 2.0
        loss_cls = self.class_scale * F.cross_entropy(cls[cls_sel], tcls_sel, reduction='sum') if tcls_sel.numel() \
            else cls.sum() * 0.0
        return (loss_x + loss_y + loss_w + loss_h + loss_conf + loss_cls) / nB
