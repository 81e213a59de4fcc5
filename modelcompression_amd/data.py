"""Minimal data sources for the train / predict entry points.

The reference's data pipeline (src/dataloader.py: PIL/cv2/torchvision augmentation) is outside
the hot path; these are just enough to drive it:
  * VOCList     -- images listed in a darknet-style list file, labels from the sibling
                   `labels/*.txt` files (cls x y w h, normalised), resized to the network input
                   with PIL only, target = 50 x 5 floats as dataloader.py:83-96 builds it;
  * SyntheticDetection -- seeded random images/boxes of the same shapes (benchmarks, smoke runs,
                   and whenever VOC is not on disk).
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

MAX_BOXES = 50


def label_path_for(imgpath):
    p = imgpath.replace('images', 'labels').replace('JPEGImages', 'labels')
    return os.path.splitext(p)[0] + '.txt'


class VOCList(Dataset):
    def __init__(self, listfile, shape=(416, 416), train=True):
        with open(listfile) as f:
            self.lines = [l.strip() for l in f if l.strip()]
        self.shape, self.train = shape, train

    def __len__(self):
        return len(self.lines)

    def __getitem__(self, i):
        from PIL import Image
        path = self.lines[i]
        img = Image.open(path).convert('RGB').resize(self.shape)
        x = torch.from_numpy(np.asarray(img, dtype=np.float32).transpose(2, 0, 1) / 255.0)
        target = torch.zeros(MAX_BOXES * 5)
        lp = label_path_for(path)
        if os.path.exists(lp) and os.path.getsize(lp):
            lab = np.loadtxt(lp).reshape(-1, 5)[:MAX_BOXES]
            target[:lab.size] = torch.from_numpy(lab.astype(np.float32).reshape(-1))
        return x, target


class SyntheticDetection(Dataset):
    def __init__(self, n, shape=(416, 416), seed=0, num_classes=20):
        self.n, self.shape, self.seed, self.nc = n, shape, seed, num_classes

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        x = torch.rand(3, self.shape[1], self.shape[0], generator=g)
        nb = int(torch.randint(1, 6, (1,), generator=g))
        target = torch.zeros(MAX_BOXES * 5)
        for b in range(nb):
            wh = torch.rand(2, generator=g) * 0.4 + 0.05
            xy = torch.rand(2, generator=g) * (1 - wh) + wh / 2
            target[b * 5:(b + 1) * 5] = torch.tensor([float(torch.randint(0, self.nc, (1,), generator=g)), xy[0], xy[1], wh[0], wh[1]])
        return x, target
