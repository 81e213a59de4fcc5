"""modelcompression_amd -- MI355X-native hot path of AnishDelft/ModelCompression.

Mirrors the reference's Python surface for the YOLOv2 prune/retrain path:

    from modelcompression_amd import nets                      # src/nets.py
    from modelcompression_amd.pruning.weightPruning import layers, methods, utils
    from modelcompression_amd.train import YOLOv2Train          # src/train.py
    from modelcompression_amd.predict import PASCALVOCEval      # src/predict.py

Compute runs in libmcamd.so (hand-written HIP for gfx950, include/mcamd.h).
"""
import os

CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg")
YOLOV2_VOC_CFG = os.path.join(CFG_DIR, "yolov2-voc.cfg")

__version__ = "0.1.0"
