"""What would the correction products of the split-operand forward cost in e2m3 (MX-FP6, block-scaled) instead of e4m3?

The "mixed" training forward multiplies x_hi w_hi on the fp16 MFMAs and x_lo w_hi + x_hi w_lo from e4m3 copies on the
block-scaled fp8 MFMAs.  `v_mfma_scale_f32_32x32x64_f8f6f4` runs e2m3 operands at TWICE the e4m3 rate
(MI355X_MICROARCH.md, matrix-core table), so the two correction products would cost half a fp16 product instead of one.
e2m3 has e4m3's three mantissa bits but two binades of normals: it needs a real e8m0 scale per 32 K-elements (32
channels of one pixel / of one filter tap).  This tool emulates both forms on the CPU (fp32 oracle arithmetic, training
mode = batch statistics) and prints the train-mode logit error of each against the plain fp32 run.

    python tools/f6_budget.py [--batch 8] [--seed 0]

Analysis infrastructure: imports oracle/ (never imported by the product)."""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import darknet_ref as O  # noqa: E402
from modelcompression_amd import YOLOV2_VOC_CFG  # noqa: E402

PLAIN = (19, 20, 22)                                   # x_hi w_hi only (engine.py, MIXED_BUDGET_TRAIN)
CORR = (3, 5, 6, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18)     # the fp8-correction blocks of the default precision


def q16(t):
    return t.half().float()


def e4m3(t, exp):
    """e4m3(t * 2^exp) / 2^exp, saturating (what bn_act.hip / the packer store)."""
    s = float(2.0 ** exp)
    return (t * s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() / s


def e2m3_block(t, axis=1, block=32):
    """OCP MX e2m3: blocks of 32 along `axis` share a power-of-two scale 2^(floor(log2 max) - 2); elements are rounded to
    the e2m3 grid (steps 0.125 below 2, 0.25 below 4, 0.5 below 8) and saturate at 7.5."""
    t = t.movedim(axis, -1)
    shp = t.shape
    C = shp[-1]
    pad = (-C) % block
    if pad:
        t = F.pad(t, (0, pad))
    b = t.reshape(-1, block)
    amax = b.abs().amax(1, keepdim=True).clamp_min(1e-38)
    scale = torch.exp2(torch.floor(torch.log2(amax)) - 2.0)
    v = b / scale
    a = v.abs()
    step = torch.where(a < 2.0, 0.125, torch.where(a < 4.0, 0.25, 0.5))
    r = (torch.round(a / step) * step).clamp_max(7.5) * torch.sign(v)
    out = (r * scale).reshape(*shp[:-1], C + pad)[..., :C]
    return out.movedim(-1, axis)


def run(blocks, state, x, form):
    """form: None = plain fp32; "lo16" = fp16 lo planes in every split block; "e4m3" / "e2m3" = that byte form in CORR."""
    outputs = {}
    for ind, op in enumerate(O.plan(blocks)):
        t = op["type"]
        if t == "conv":
            p, i = op["prefix"], op["id"]
            w = state[p + "conv%d.weight" % i]
            bias = None if op["bn"] else state[p + "conv%d.bias" % i]
            if form is None:
                x = F.conv2d(x, w, bias, 1, op["pad"])
            else:
                xh, wh = q16(x), q16(w)
                xl, wl = q16(x - xh), q16(w - wh)
                if i in PLAIN:
                    x = F.conv2d(xh, wh, bias, 1, op["pad"])
                elif i in CORR and form in ("e4m3", "e2m3"):
                    if form == "e4m3":
                        wexp = int(torch.floor(torch.log2(224.0 / w.abs().max())))
                        xl8, xh8 = e4m3(xl, 12), e4m3(xh, 1)
                        wh8, wl8 = e4m3(wh, wexp), e4m3(wl, wexp + 11)
                    else:
                        xl8, xh8 = e2m3_block(xl), e2m3_block(xh)
                        wh8, wl8 = e2m3_block(wh), e2m3_block(wl)
                    x = F.conv2d(xh, wh, bias, 1, op["pad"]) + F.conv2d(xl8, wh8, None, 1, op["pad"]) \
                        + F.conv2d(xh8, wl8, None, 1, op["pad"])
                else:
                    x = F.conv2d(xh, wh, bias, 1, op["pad"]) + F.conv2d(xl, wh, None, 1, op["pad"]) \
                        + F.conv2d(xh, wl, None, 1, op["pad"])
            if op["bn"]:
                x = F.batch_norm(x, None, None, state[p + "bn%d.weight" % i], state[p + "bn%d.bias" % i], True, 0.1, 1e-5)
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
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    blocks = O.parse_cfg(YOLOV2_VOC_CFG)
    state = O.init_state(blocks, seed=a.seed)
    x = torch.rand(a.batch, 3, 416, 416, generator=torch.Generator().manual_seed(a.seed + 7))
    ref = run(blocks, state, x, None)
    for form in ("lo16", "e4m3", "e2m3"):
        print("B=%d seed %d  %-5s train logits rel-L2 %.3e" % (a.batch, a.seed, form, rel(run(blocks, state, x, form), ref)), flush=True)


if __name__ == "__main__":
    main()
