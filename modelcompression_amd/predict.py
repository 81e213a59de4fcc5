"""Eval entry point: mirror of the reference's src/predict.py (PASCALVOCEval, 37-451).

predict(): batched forward on the HIP engine -> get_region_boxes -> nms per image -> one
detection file per class -> VOC07 11-point AP (voc_ap / voc_eval / _do_python_eval), with the
same constructor and method signatures.  Image lists that are missing fall back to the
synthetic detection set so the plumbing can be exercised without VOC on disk (mAP is then
meaningless and not computed).
"""
import os
import pickle
import xml.etree.ElementTree as ET

import numpy as np
import torch

from .data import VOCList, SyntheticDetection
from .nets import getYOLOv2
from .nets2_utils import get_region_boxes, nms, detections, get_image_size  # noqa: F401


class PASCALVOCEval():

    def __init__(self, MODEL, MODEL_CFGFILE, MODEL_WEIGHTFILE, MODEL_LOSS,
                 PASCAL_DIR, EVAL_IMAGELIST, EVAL_OUTPUTDIR, EVAL_PREFIX, EVAL_OUTPUTDIR_PKL,
                 LOGGER='', LOGGER_EPOCH=-1, verbose=0):
        self.MODEL = MODEL
        self.MODEL_CFGFILE = MODEL_CFGFILE
        self.MODEL_WEIGHTFILE = MODEL_WEIGHTFILE
        self.MODEL_LOSS = MODEL_LOSS
        self.PASCAL_DIR = PASCAL_DIR
        self.EVAL_IMAGELIST = EVAL_IMAGELIST
        self.EVAL_OUTPUTDIR = EVAL_OUTPUTDIR
        self.EVAL_PREFIX = EVAL_PREFIX
        self.EVAL_OUTPUTDIR_PKL = EVAL_OUTPUTDIR_PKL
        self.LOGGER, self.LOGGER_EPOCH, self.verbose = LOGGER, LOGGER_EPOCH, verbose
        self.USE_GPU = torch.cuda.is_available()
        self.VOC_CLASSES = ("aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow",
                            "diningtable", "dog", "horse", "motorbike", "person", "pottedplant", "sheep", "sofa",
                            "train", "tvmonitor")
        self.VOC_YEAR = '2007'
        self.mAP = None

    def predict(self, BATCH_SIZE=2, CONF_THRESH=0.005, NMS_THRESH=0.45):
        if self.MODEL == '' or self.MODEL is None:
            self.MODEL = getYOLOv2(self.MODEL_CFGFILE, self.MODEL_WEIGHTFILE)
        self.MODEL.eval()
        dev = next(self.MODEL.parameters()).device
        have_list = bool(self.EVAL_IMAGELIST) and os.path.exists(self.EVAL_IMAGELIST)
        if have_list:
            with open(self.EVAL_IMAGELIST) as fp:
                valid_files = [item.rstrip() for item in fp.readlines()]
            ds = VOCList(self.EVAL_IMAGELIST, shape=(self.MODEL.width, self.MODEL.height), train=False)
        else:
            ds = SyntheticDetection(4 * BATCH_SIZE, shape=(self.MODEL.width, self.MODEL.height), seed=1)
            valid_files = ['synthetic_%06d.jpg' % i for i in range(len(ds))]
        loader = torch.utils.data.DataLoader(ds, batch_size=BATCH_SIZE, shuffle=False, num_workers=1 if have_list else 0,
                                             pin_memory=True)
        os.makedirs(self.EVAL_OUTPUTDIR, exist_ok=True)
        fps = [open('%s/%s%s.txt' % (self.EVAL_OUTPUTDIR, self.EVAL_PREFIX, c), 'w') for c in self.VOC_CLASSES[:self.MODEL.num_classes]]
        lineId, val_loss_total, ndet = -1, 0.0, 0
        with torch.no_grad():
            for data, target in loader:
                output = self.MODEL(data.to(dev))
                if self.LOGGER != '' and self.MODEL_LOSS is not None and bool((target != 0).any()):
                    val_loss_total += float(self.MODEL_LOSS(output, target.float().to(dev)))
                # predict.py:148-173 (get_region_boxes(.., 0, 1) -> nms -> one line per class above the threshold) as one
                # batched device computation; only the surviving detections come to the host
                batch_dets = detections(output, CONF_THRESH, NMS_THRESH, self.MODEL.num_classes, self.MODEL.anchors,
                                        self.MODEL.num_anchors)
                for i in range(output.size(0)):
                    lineId += 1
                    fileId = os.path.basename(valid_files[lineId]).split('.')[0]
                    size = get_image_size(valid_files[lineId]) if have_list else None
                    width, height = size if size else (self.MODEL.width, self.MODEL.height)
                    for box, classes in batch_dets[i]:
                        x1, y1 = (box[0] - box[2] / 2.0) * width, (box[1] - box[3] / 2.0) * height
                        x2, y2 = (box[0] + box[2] / 2.0) * width, (box[1] + box[3] / 2.0) * height
                        for cls_id, prob in classes:
                            fps[cls_id].write('%s %f %f %f %f %f\n' % (fileId, prob, x1, y1, x2, y2))
                            ndet += 1
        if self.LOGGER != '' and self.MODEL_LOSS is not None:
            self.LOGGER.save_value('Total Loss', 'Val Loss', self.LOGGER_EPOCH + 1, val_loss_total / max(len(loader), 1))
        for f in fps:
            f.close()
        self.num_detections = ndet
        if have_list and os.path.isdir(os.path.join(self.PASCAL_DIR or '', 'VOC' + self.VOC_YEAR, 'Annotations')):
            self._do_python_eval()
        return self.mAP

    def parse_rec(self, filename):
        """ Parse a PASCAL VOC xml file """
        objects = []
        for obj in ET.parse(filename).findall('object'):
            bbox = obj.find('bndbox')
            objects.append({'name': obj.find('name').text, 'pose': obj.find('pose').text,
                            'truncated': int(obj.find('truncated').text), 'difficult': int(obj.find('difficult').text),
                            'bbox': [int(bbox.find(k).text) for k in ('xmin', 'ymin', 'xmax', 'ymax')]})
        return objects

    def voc_ap(self, rec, prec, use_07_metric=False):
        """predict.py:216-247: VOC07 11-point AP, or the area under the monotone envelope."""
        if use_07_metric:
            ap = 0.
            for t in np.arange(0., 1.1, 0.1):
                p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
                ap = ap + p / 11.
            return ap
        mrec = np.concatenate(([0.], rec, [1.]))
        mpre = np.concatenate(([0.], prec, [0.]))
        for i in range(mpre.size - 1, 0, -1):
            mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
        i = np.where(mrec[1:] != mrec[:-1])[0]
        return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])

    def voc_eval(self, detpath, annopath, imagesetfile, classname, cachedir, ovthresh=0.5, use_07_metric=False):
        """predict.py:250-395 (the py-faster-rcnn protocol): greedy matching of detections, sorted by
        confidence, to not-yet-matched ground truth of the class with IoU > ovthresh (+1 pixel
        convention); difficult boxes are ignored."""
        os.makedirs(cachedir, exist_ok=True)
        cachefile = os.path.join(cachedir, 'annots.pkl')
        with open(imagesetfile) as f:
            imagenames = [x.strip() for x in f.readlines()]
        if not os.path.isfile(cachefile):
            recs = {n: self.parse_rec(annopath.format(n)) for n in imagenames}
            with open(cachefile, 'wb') as f:
                pickle.dump(recs, f)
        else:
            with open(cachefile, 'rb') as f:
                recs = pickle.load(f)
        class_recs, npos = {}, 0
        for n in imagenames:
            R = [obj for obj in recs[n] if obj['name'] == classname]
            difficult = np.array([x['difficult'] for x in R]).astype(bool)
            npos += int(np.sum(~difficult))
            class_recs[n] = {'bbox': np.array([x['bbox'] for x in R]), 'difficult': difficult, 'det': [False] * len(R)}
        with open(detpath.format(classname)) as f:
            splitlines = [x.strip().split(' ') for x in f.readlines()]
        if not splitlines:
            return np.zeros(0), np.zeros(0), 0.0
        image_ids = [x[0] for x in splitlines]
        confidence = np.array([float(x[1]) for x in splitlines])
        BB = np.array([[float(z) for z in x[2:]] for x in splitlines])
        order = np.argsort(-confidence)
        BB, image_ids = BB[order, :], [image_ids[x] for x in order]
        nd = len(image_ids)
        tp, fp = np.zeros(nd), np.zeros(nd)
        for d in range(nd):
            R = class_recs[image_ids[d]]
            bb, ovmax, BBGT = BB[d, :].astype(float), -np.inf, R['bbox'].astype(float)
            if BBGT.size > 0:
                iw = np.maximum(np.minimum(BBGT[:, 2], bb[2]) - np.maximum(BBGT[:, 0], bb[0]) + 1., 0.)
                ih = np.maximum(np.minimum(BBGT[:, 3], bb[3]) - np.maximum(BBGT[:, 1], bb[1]) + 1., 0.)
                inters = iw * ih
                uni = ((bb[2] - bb[0] + 1.) * (bb[3] - bb[1] + 1.) +
                       (BBGT[:, 2] - BBGT[:, 0] + 1.) * (BBGT[:, 3] - BBGT[:, 1] + 1.) - inters)
                overlaps = inters / uni
                ovmax, jmax = np.max(overlaps), int(np.argmax(overlaps))
            if ovmax > ovthresh:
                if not R['difficult'][jmax]:
                    if not R['det'][jmax]:
                        tp[d] = 1.
                        R['det'][jmax] = 1
                    else:
                        fp[d] = 1.
            else:
                fp[d] = 1.
        fp, tp = np.cumsum(fp), np.cumsum(tp)
        rec = tp / float(max(npos, 1))
        prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
        return rec, prec, self.voc_ap(rec, prec, use_07_metric)

    def _do_python_eval(self):
        """predict.py:398-437: per-class AP with the VOC07 metric, mean over classes."""
        base = os.path.join(self.PASCAL_DIR, 'VOC' + self.VOC_YEAR)
        annopath = os.path.join(base, 'Annotations', '{:s}.xml')
        imagesetfile = os.path.join(base, 'ImageSets', 'Main', 'test.txt')
        detpath = os.path.join(self.EVAL_OUTPUTDIR, self.EVAL_PREFIX + '{:s}.txt')
        aps = []
        for cls in self.VOC_CLASSES:
            rec, prec, ap = self.voc_eval(detpath, annopath, imagesetfile, cls, self.EVAL_OUTPUTDIR_PKL, 0.5, True)
            aps.append(ap)
            print('AP for {} = {:.4f}'.format(cls, ap))
        self.mAP = float(np.mean(aps))
        print('Mean AP = {:.4f}'.format(self.mAP))
        if self.LOGGER != '':
            self.LOGGER.save_value('mAP', 'Val mAP', self.LOGGER_EPOCH + 1, self.mAP)
        return self.mAP
