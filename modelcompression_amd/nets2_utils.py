"""Post-processing of the reference's src/nets2_utils.py that the eval entry point needs:
bbox_iou / bbox_ious (63-131), get_region_boxes (141-234), nms (236-259), file helpers.

get_region_boxes / nms keep the reference's list formats and results; underneath, decoding, thresholding and the
greedy suppression are batched tensor computations (`region_boxes_tensors`, `nms_tensors`) that predict.py uses
directly, so that only the surviving detections cross to the host.
"""
import os
import struct
import time

import torch


def logging(message):
    print('%s %s' % (time.strftime("%Y-%m-%d %H:%M:%S", time.localtime()), message))


def file_lines(thefilepath):
    with open(thefilepath, 'rb') as f:
        return sum(buf.count(b'\n') for buf in iter(lambda: f.read(1 << 20), b''))


def bbox_iou(box1, box2, x1y1x2y2=True):
    """nets2_utils.py:63-98 (scalar boxes)."""
    if x1y1x2y2:
        mx, Mx = min(box1[0], box2[0]), max(box1[2], box2[2])
        my, My = min(box1[1], box2[1]), max(box1[3], box2[3])
        w1, h1, w2, h2 = box1[2] - box1[0], box1[3] - box1[1], box2[2] - box2[0], box2[3] - box2[1]
    else:
        mx = min(box1[0] - box1[2] / 2.0, box2[0] - box2[2] / 2.0)
        Mx = max(box1[0] + box1[2] / 2.0, box2[0] + box2[2] / 2.0)
        my = min(box1[1] - box1[3] / 2.0, box2[1] - box2[3] / 2.0)
        My = max(box1[1] + box1[3] / 2.0, box2[1] + box2[3] / 2.0)
        w1, h1, w2, h2 = box1[2], box1[3], box2[2], box2[3]
    cw, ch = w1 + w2 - (Mx - mx), h1 + h2 - (My - my)
    if cw <= 0 or ch <= 0:
        return 0.0
    carea = cw * ch
    return carea / (w1 * h1 + w2 * h2 - carea)


def bbox_ious(boxes1, boxes2, x1y1x2y2=True):
    """nets2_utils.py:100-131: boxes as [4, N] tensors."""
    if x1y1x2y2:
        mx, Mx = torch.min(boxes1[0], boxes2[0]), torch.max(boxes1[2], boxes2[2])
        my, My = torch.min(boxes1[1], boxes2[1]), torch.max(boxes1[3], boxes2[3])
        w1, h1 = boxes1[2] - boxes1[0], boxes1[3] - boxes1[1]
        w2, h2 = boxes2[2] - boxes2[0], boxes2[3] - boxes2[1]
    else:
        mx = torch.min(boxes1[0] - boxes1[2] / 2.0, boxes2[0] - boxes2[2] / 2.0)
        Mx = torch.max(boxes1[0] + boxes1[2] / 2.0, boxes2[0] + boxes2[2] / 2.0)
        my = torch.min(boxes1[1] - boxes1[3] / 2.0, boxes2[1] - boxes2[3] / 2.0)
        My = torch.max(boxes1[1] + boxes1[3] / 2.0, boxes2[1] + boxes2[3] / 2.0)
        w1, h1, w2, h2 = boxes1[2], boxes1[3], boxes2[2], boxes2[3]
    cw, ch = w1 + w2 - (Mx - mx), h1 + h2 - (My - my)
    carea = cw * ch
    carea = torch.where((cw <= 0) | (ch <= 0), torch.zeros_like(carea), carea)
    return carea / (w1 * h1 + w2 * h2 - carea)


def region_boxes_tensors(output, num_classes, anchors_list, anchors_cell):
    """The decoding half of get_region_boxes (nets2_utils.py:141-190) as device tensors, one row per cell in the
    order the reference's loop visits them, (cy, cx, anchor):
        head     [B, N, 7]  x, y, w, h (network-relative), box_conf, cls_max_conf, cls_max_id
        cls_conf [B, N, C]  softmax class confidences
    N = h * w * anchors_cell.  Nothing leaves the device."""
    if output.dim() == 3:
        output = output.unsqueeze(0)
    assert output.size(1) == (5 + num_classes) * anchors_cell
    step = int(len(anchors_list) / anchors_cell)
    B, h, w = output.size(0), output.size(2), output.size(3)
    dev = output.device
    o = output.detach().float().view(B, anchors_cell, 5 + num_classes, h, w)
    gx = torch.arange(w, device=dev, dtype=torch.float32).view(1, 1, 1, w)
    gy = torch.arange(h, device=dev, dtype=torch.float32).view(1, 1, h, 1)
    aw = torch.tensor(anchors_list[0::step], dtype=torch.float32).to(dev).view(1, anchors_cell, 1, 1)
    ah = torch.tensor(anchors_list[1::step], dtype=torch.float32).to(dev).view(1, anchors_cell, 1, 1)
    xs = (torch.sigmoid(o[:, :, 0]) + gx) / w
    ys = (torch.sigmoid(o[:, :, 1]) + gy) / h
    ws = torch.exp(o[:, :, 2]) * aw / w
    hs = torch.exp(o[:, :, 3]) * ah / h
    box_conf = torch.sigmoid(o[:, :, 4])
    cls_conf = torch.softmax(o[:, :, 5:], dim=2)                       # [B, A, C, h, w]
    cmax, cid = cls_conf.max(2)
    head = torch.stack((xs, ys, ws, hs, box_conf, cmax, cid.float()), -1)           # [B, A, h, w, 7]
    head = head.permute(0, 2, 3, 1, 4).reshape(B, h * w * anchors_cell, 7)          # (cy, cx, a) order
    cls_conf = cls_conf.permute(0, 3, 4, 1, 2).reshape(B, h * w * anchors_cell, num_classes)
    return head, cls_conf


def get_region_boxes(output, CONF_THRESH, num_classes, anchors_list, anchors_cell, only_objectness=1, validation=False):
    """nets2_utils.py:141-234, same return value: per image a list of
    [x, y, w, h, box_conf, cls_max_conf, cls_max_id (, conf_c, c)*], cells in (cy, cx, anchor) order.  Decoding and
    thresholding run on the device (the reference copies seven [B*845] vectors to the host and loops over every
    cell in Python, 200-232); one device->host copy of the surviving rows builds the lists."""
    head, cls_conf = region_boxes_tensors(output, num_classes, anchors_list, anchors_cell)
    conf = head[..., 4] if only_objectness else head[..., 4] * head[..., 5]
    keep = conf > CONF_THRESH
    want_extra = (not only_objectness) and validation
    keep_c, head_c = keep.cpu(), head.cpu()
    cls_c = cls_conf.cpu() if want_extra else None
    all_boxes = []
    for b in range(head_c.size(0)):
        rows = torch.nonzero(keep_c[b]).flatten().tolist()
        boxes = []
        for i in rows:
            r = head_c[b, i]
            box = [r[0], r[1], r[2], r[3], r[4], r[5], int(r[6])]
            if want_extra:
                cc = cls_c[b, i]
                for c in torch.nonzero(r[4] * cc > CONF_THRESH).flatten().tolist():
                    if c != box[6]:
                        box.append(cc[c])
                        box.append(c)
            boxes.append(box)
        all_boxes.append(boxes)
    return all_boxes


def _iou_matrix_cwh(bx):
    """[B, n, 4] centre/size boxes -> [B, n, n] IoU with bbox_iou's arithmetic (x1y1x2y2=False, nets2_utils.py:63-98)."""
    x, y, w, h = bx[..., 0], bx[..., 1], bx[..., 2], bx[..., 3]
    lo_x, hi_x = x - w / 2.0, x + w / 2.0
    lo_y, hi_y = y - h / 2.0, y + h / 2.0
    mx = torch.min(lo_x[:, :, None], lo_x[:, None, :])
    Mx = torch.max(hi_x[:, :, None], hi_x[:, None, :])
    my = torch.min(lo_y[:, :, None], lo_y[:, None, :])
    My = torch.max(hi_y[:, :, None], hi_y[:, None, :])
    cw = w[:, :, None] + w[:, None, :] - (Mx - mx)
    ch = h[:, :, None] + h[:, None, :] - (My - my)
    carea = cw * ch
    uarea = w[:, :, None] * h[:, :, None] + w[:, None, :] * h[:, None, :] - carea
    return torch.where((cw <= 0) | (ch <= 0), torch.zeros_like(carea), carea / uarea)


def nms_tensors(boxes, conf, NMS_THRESH, chunk=16):
    """Greedy NMS of nets2_utils.py:236-259 for a whole batch on the device.
    boxes [B, n, 4] (cx, cy, w, h), conf [B, n] (<= 0: not a candidate).  Returns (order [B, n], kept [B, n]):
    `order` sorts each image by descending confidence (the reference sorts 1 - conf ascending), kept[b, r] tells
    whether the r-th box of that order survives.

    The reference's double loop keeps box j iff no KEPT box i before it (in confidence order) overlaps it by more
    than the threshold.  That recurrence has one solution; it is reached by iterating
    kept <- candidate & ~any_i(kept_i & over[i, j]) from kept = candidate, which fixes at least one more position
    of the order per sweep and in practice converges in a handful of sweeps (the convergence test is the only
    host synchronisation: one per four sweeps and chunk of images)."""
    B, n = conf.shape
    order = torch.sort(1 - conf, dim=1, stable=True).indices
    kept_all = torch.zeros(B, n, dtype=torch.bool, device=conf.device)
    for b0 in range(0, B, chunk):
        o = order[b0:b0 + chunk]
        bx = torch.gather(boxes[b0:b0 + chunk], 1, o[..., None].expand(-1, -1, 4))
        cf = torch.gather(conf[b0:b0 + chunk], 1, o)
        cand = cf > 0
        over = (_iou_matrix_cwh(bx) > NMS_THRESH).triu(1) & cand[:, :, None] & cand[:, None, :]
        kept = cand
        while True:
            prev = kept
            for _ in range(4):
                kept = cand & ~(over & kept[:, :, None]).any(1)
            if bool((kept == prev).all()):
                break
        kept_all[b0:b0 + chunk] = kept
    return order, kept_all


def nms(boxes, NMS_THRESH):
    """nets2_utils.py:236-259: greedy by descending box confidence; suppressed boxes get conf 0 (in the caller's
    list, as the reference does) and are left out of the result.  The O(n^2) Python loop of the reference is one
    batched tensor computation here (`nms_tensors`)."""
    if len(boxes) == 0:
        return boxes
    t = torch.tensor([[float(v) for v in b[:5]] for b in boxes], dtype=torch.float32)
    order, kept = nms_tensors(t[None, :, :4], t[None, :, 4], NMS_THRESH)
    order, kept = order[0].tolist(), kept[0].tolist()
    out = []
    for r, i in enumerate(order):
        if kept[r]:
            out.append(boxes[i])
        elif float(boxes[i][4]) > 0:
            boxes[i][4] = 0
    return out


def detections(output, CONF_THRESH, NMS_THRESH, num_classes, anchors_list, anchors_cell):
    """What predict.py:148-173 derives from a batch of logits -- get_region_boxes(..., only_objectness=0,
    validation=True), nms per image, then one (class, probability) pair per class whose box_conf * cls_conf
    clears CONF_THRESH (the arg-max class always) -- computed for the whole batch on the device with ONE
    device->host copy.  Returns per image a list of ((x, y, w, h) fp32 tensor row, [(cls_id, prob), ...]) in the
    reference's order: boxes by descending objectness, the arg-max class first, then the others ascending."""
    head, cls_conf = region_boxes_tensors(output, num_classes, anchors_list, anchors_cell)
    cand = head[..., 4] * head[..., 5] > CONF_THRESH
    order, kept = nms_tensors(head[..., :4], torch.where(cand, head[..., 4], torch.zeros_like(head[..., 4])), NMS_THRESH)
    head_s = torch.gather(head, 1, order[..., None].expand(-1, -1, 7))
    cls_s = torch.gather(cls_conf, 1, order[..., None].expand(-1, -1, num_classes))
    probs = head_s[..., 4:5] * cls_s
    emit = probs > CONF_THRESH
    emit.scatter_(2, head_s[..., 6:7].long(), True)
    emit &= kept[..., None]
    kept_c, head_c, probs_c, emit_c = kept.cpu(), head_s.cpu(), probs.cpu(), emit.cpu()
    res = []
    for b in range(head_c.size(0)):
        dets = []
        for r in torch.nonzero(kept_c[b]).flatten().tolist():
            top = int(head_c[b, r, 6])
            cls = [top] + [c for c in torch.nonzero(emit_c[b, r]).flatten().tolist() if c != top]
            dets.append((head_c[b, r, :4], [(c, probs_c[b, r, c]) for c in cls]))
        res.append(dets)
    return res


def get_image_size(fname):
    """Width, height of a PNG / GIF / JPEG without decoding it (nets2_utils.py:502-535)."""
    with open(fname, 'rb') as fh:
        head = fh.read(24)
        if len(head) != 24:
            return
        if head[:8] == b'\211PNG\r\n\032\n':
            return struct.unpack('>ii', head[16:24])
        if head[:6] in (b'GIF87a', b'GIF89a'):
            return struct.unpack('<HH', head[6:10])
        if head[:2] == b'\xff\xd8':
            fh.seek(0)
            size, ftype = 2, 0
            while not 0xc0 <= ftype <= 0xcf:
                fh.seek(size, 1)
                byte = fh.read(1)
                while ord(byte) == 0xff:
                    byte = fh.read(1)
                ftype = ord(byte)
                size = struct.unpack('>H', fh.read(2))[0] - 2
            fh.seek(1, 1)
            height, width = struct.unpack('>HH', fh.read(4))
            return width, height
    return
