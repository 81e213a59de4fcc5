"""Post-processing of the reference's src/nets2_utils.py that the eval entry point needs:
bbox_iou / bbox_ious (63-131), get_region_boxes (141-234), nms (236-259), file helpers.

get_region_boxes keeps the reference's output format -- per image a list of
[x, y, w, h, box_conf, cls_max_conf, cls_max_id (, conf_c, c)*] in network-relative units,
cells visited in (cy, cx, anchor) order -- but selects the surviving cells with tensor ops on
the device instead of a Python loop over all B*13*13*5 cells (only survivors cross to the host).
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


def get_region_boxes(output, CONF_THRESH, num_classes, anchors_list, anchors_cell, only_objectness=1, validation=False):
    if output.dim() == 3:
        output = output.unsqueeze(0)
    assert output.size(1) == (5 + num_classes) * anchors_cell
    step = int(len(anchors_list) / anchors_cell)
    B, h, w = output.size(0), output.size(2), output.size(3)
    dev = output.device
    o = output.detach().float().view(B, anchors_cell, 5 + num_classes, h, w)
    gx = torch.arange(w, device=dev, dtype=torch.float32).view(1, 1, 1, w)
    gy = torch.arange(h, device=dev, dtype=torch.float32).view(1, 1, h, 1)
    aw = torch.tensor(anchors_list[0::step], device=dev).view(1, anchors_cell, 1, 1)
    ah = torch.tensor(anchors_list[1::step], device=dev).view(1, anchors_cell, 1, 1)
    xs = (torch.sigmoid(o[:, :, 0]) + gx) / w
    ys = (torch.sigmoid(o[:, :, 1]) + gy) / h
    ws = torch.exp(o[:, :, 2]) * aw / w
    hs = torch.exp(o[:, :, 3]) * ah / h
    box_conf = torch.sigmoid(o[:, :, 4])
    cls_conf = torch.softmax(o[:, :, 5:], dim=2)                       # [B, A, C, h, w]
    cmax, cid = cls_conf.max(2)
    conf = box_conf if only_objectness else box_conf * cmax
    keep = conf > CONF_THRESH
    all_boxes = []
    for b in range(B):
        # the reference visits (cy, cx, anchor): order the survivors the same way
        idx = torch.nonzero(keep[b].permute(1, 2, 0), as_tuple=False)   # [n, 3] = (cy, cx, a) ascending
        if idx.numel() == 0:
            all_boxes.append([])
            continue
        cy, cx, a = idx[:, 0], idx[:, 1], idx[:, 2]
        cols = torch.stack((xs[b, a, cy, cx], ys[b, a, cy, cx], ws[b, a, cy, cx], hs[b, a, cy, cx],
                            box_conf[b, a, cy, cx], cmax[b, a, cy, cx], cid[b, a, cy, cx].float()), 1).cpu()
        extra = None
        if (not only_objectness) and validation:
            cc = cls_conf[b, a, :, cy, cx]                               # [n, C]
            extra = ((box_conf[b, a, cy, cx].unsqueeze(1) * cc) > CONF_THRESH).cpu()
            cc = cc.cpu()
        boxes = []
        for i in range(cols.size(0)):
            r = cols[i]
            box = [r[0], r[1], r[2], r[3], r[4], r[5], int(r[6])]
            if extra is not None:
                for c in torch.nonzero(extra[i]).flatten().tolist():
                    if c != box[6]:
                        box.append(cc[i, c])
                        box.append(c)
            boxes.append(box)
        all_boxes.append(boxes)
    return all_boxes


def nms(boxes, NMS_THRESH):
    """nets2_utils.py:236-259: greedy by descending box confidence; suppressed boxes get conf 0."""
    if len(boxes) == 0:
        return boxes
    confs = torch.tensor([1 - float(b[4]) for b in boxes])
    _, order = torch.sort(confs)
    out = []
    for i in range(len(boxes)):
        bi = boxes[order[i]]
        if bi[4] > 0:
            out.append(bi)
            for j in range(i + 1, len(boxes)):
                bj = boxes[order[j]]
                if bbox_iou(bi, bj, x1y1x2y2=False) > NMS_THRESH:
                    bj[4] = 0
    return out


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
