"""Per-layer error budget of the logits: which fp16 roundings cost what -- in eval mode (running
statistics) and, with --train, in training mode (batch statistics), where a random-init BatchNorm / LeakyReLU network
amplifies a perturbation by 1.05 ... 1.6 per block on its way to the logits (conv1's operand rounding alone: 1.9e-2).

Emulates the engine's eval forward on the CPU (fp32 oracle arithmetic) with three rounding sources that can be
switched per conv block:
    w  : the conv weights are fp16 MFMA operands
    x  : the conv input (the stored activation) is an fp16 MFMA operand
    y  : the raw conv output is stored as fp16 before BN/LeakyReLU (the non-fused layers)
and prints the relative L2 error of the region-layer logits against the plain fp32 run.

    python tools/error_budget.py [--seed 0] [--batch 1] [--train]

Test/analysis infrastructure: imports oracle/ (never imported by the product).
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import darknet_ref as O  # noqa: E402
from modelcompression_amd import YOLOV2_VOC_CFG  # noqa: E402


def q(t):
    return t.half().float()


TRAIN = False


def run(blocks, state, x, rw=(), rx=(), ry=(), split_w=(), split_x=()):
    """rw / rx / ry: sets of conv ids (1-based) whose weights / inputs / raw outputs are rounded to fp16.
    split_*: conv ids whose operand is carried as hi + lo fp16 pair (error ~2^-22)."""
    outputs = {}
    for ind, op in enumerate(O.plan(blocks)):
        t = op["type"]
        if t == "conv":
            p, i = op["prefix"], op["id"]
            w = state[p + "conv%d.weight" % i]
            if i in rw:
                w = q(w) + (q(w - q(w)) if i in split_w else 0)
            if i in rx:
                x = q(x) + (q(x - q(x)) if i in split_x else 0)
            bias = None if op["bn"] else state[p + "conv%d.bias" % i]
            x = F.conv2d(x, w, bias, 1, op["pad"])
            if op["bn"]:
                if i in ry:
                    x = q(x)
                if TRAIN:
                    x = F.batch_norm(x, None, None, state[p + "bn%d.weight" % i], state[p + "bn%d.bias" % i], True, 0.1, 1e-5)
                else:
                    x = F.batch_norm(x, state[p + "bn%d.running_mean" % i], state[p + "bn%d.running_var" % i],
                                     state[p + "bn%d.weight" % i], state[p + "bn%d.bias" % i], False, 0.1, 1e-5)
            if op["act"] == "leaky":
                x = F.leaky_relu(x, 0.1)
        elif t == "maxpool":
            x = F.max_pool2d(x, op["size"], op["stride"])
        elif t == "reorg":
            x = O.reorg(x, op["stride"])
        elif t == "route":
            ls = op["layers"]
            x = outputs[ls[0]] if len(ls) == 1 else torch.cat((outputs[ls[0]], outputs[ls[1]]), 1)
        elif t == "region":
            continue
        outputs[ind] = x
    return x


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--train", action="store_true", help="training-mode forward (batch statistics)")
    a = ap.parse_args()
    global TRAIN
    TRAIN = a.train
    torch.set_grad_enabled(False)
    blocks = O.parse_cfg(YOLOV2_VOC_CFG)
    state = O.init_state(blocks, seed=a.seed)
    x = torch.rand(a.batch, 3, 416, 416, generator=torch.Generator().manual_seed(a.seed + 7))
    convs = [op for op in O.plan(blocks) if op["type"] == "conv"]
    ids = [op["id"] for op in convs]
    allc = set(ids)
    ref = run(blocks, state, x)
    print("all three sources on every layer (r01 non-fused): %.3e" % rel(run(blocks, state, x, allc, allc, allc), ref))
    print("w + x on every layer (fused eval epilogue)      : %.3e" % rel(run(blocks, state, x, allc, allc), ref))
    print("w only                                           : %.3e" % rel(run(blocks, state, x, allc), ref))
    print("x only                                           : %.3e" % rel(run(blocks, state, x, (), allc), ref))
    print("hi+lo split of both operands everywhere          : %.3e" % rel(run(blocks, state, x, allc, allc, (), allc, allc), ref))
    print("per layer, w + x rounding of that layer alone:")
    flops = {}
    hw = 416
    for op in convs:
        flops[op["id"]] = 2.0 * op["cin"] * op["cout"] * op["k"] ** 2
    per = {}
    for i in ids:
        per[i] = rel(run(blocks, state, x, {i}, {i}), ref)
        print("  conv%-2d  %.3e" % (i, per[i]))
    tot = sum(v * v for v in per.values()) ** 0.5
    print("root-sum-square of the per-layer terms: %.3e" % tot)
    # policy: plain fp16 for the costliest layers, split operands elsewhere
    for plain in ((), (22,), (19, 20, 22), (18, 19, 20, 22), (14, 16, 18, 19, 20, 22), (1, 19, 20, 22)):
        sp = allc - set(plain)
        e = rel(run(blocks, state, x, allc, allc, (), sp, sp), ref)
        print("split everywhere except %-24s -> %.3e" % (list(plain), e))


if __name__ == "__main__":
    main()
