"""modelcompression_amd -- MI355X-native hot path of AnishDelft/ModelCompression.

Mirrors the reference's Python surface for the YOLOv2 prune/retrain path:

    from modelcompression_amd import nets                      # src/nets.py
    from modelcompression_amd.pruning.weightPruning import layers, methods, utils
    from modelcompression_amd.train import YOLOv2Train          # src/train.py
    from modelcompression_amd.predict import PASCALVOCEval      # src/predict.py

Compute runs in libmcamd.so (hand-written HIP for gfx950, include/mcamd.h).
"""
import os

import tempfile

from .yolov2_voc import write_cfg as _write_cfg

CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg")


def _materialise_cfg():
    """The YOLOv2-VOC .cfg is generated from yolov2_voc.py (no network text file is shipped)."""
    for d in (CFG_DIR, os.path.join(tempfile.gettempdir(), "modelcompression_amd_cfg_%d" % os.getuid())):
        try:
            os.makedirs(d, exist_ok=True)
            return _write_cfg(os.path.join(d, "yolov2_voc_416.cfg"))
        except OSError:
            continue
    raise RuntimeError("cannot write the generated YOLOv2-VOC cfg")


YOLOV2_VOC_CFG = _materialise_cfg()

__version__ = "0.1.0"
