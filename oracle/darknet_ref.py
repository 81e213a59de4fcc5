"""CPU restatement of the reference's Darknet/YOLOv2 model path (src/nets.py).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Functional PyTorch-CPU fp32:
the network is a list of cfg blocks plus a flat `state` dict whose keys are the
reference's `state_dict()` keys (`models.{i}.conv{id}.weight`,
`models.{i}.bn{id}.running_mean`, ...), so a reference state_dict can be fed to
it unchanged and vice versa.  The convolution itself is `F.conv2d` (PyTorch /
oneDNN, a third-party dependency of the reference that is not under
/root/reference; the reference pins no version -- the pin is torch 2.10.0).

Follows: parse_cfg nets.py:39-73; create_network nets.py:779-895; forward
nets.py:720-774; MaskedConv2d.forward layers.py:53-64; Reorg nets.py:648-667;
load/save weights nets.py:256-279, 897-948, 1007-1051.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------
# cfg grammar -- nets.py:39-73
# ---------------------------------------------------------------------------
def parse_cfg(cfgfile):
    """.cfg text -> list of dicts.  Only rstrip() is applied to a line before the
    blank/comment test (nets.py:45-46), `[name]` opens a block (lstrip('[') /
    rstrip(']'), nets.py:56), convolutional blocks get the *int* default
    batch_normalize=0 (nets.py:58-60), `key=value` must split into exactly two
    parts (nets.py:62) and both are stripped; key `type` is stored as `_type`."""
    blocks, block = [], None
    with open(cfgfile, "r") as fp:
        for raw in fp:
            line = raw.rstrip()
            if line == "" or line[0] == "#":
                continue
            if line[0] == "[":
                if block:
                    blocks.append(block)
                block = dict()
                block["type"] = line.lstrip("[").rstrip("]")
                if block["type"] == "convolutional":
                    block["batch_normalize"] = 0
            else:
                key, value = line.split("=")
                key = key.strip()
                if key == "type":
                    key = "_type"
                block[key] = value.strip()
    if block:
        blocks.append(block)
    return blocks


# ---------------------------------------------------------------------------
# network plan -- nets.py:779-895
# ---------------------------------------------------------------------------
def plan(blocks):
    """One entry per non-[net] block, index == index into the reference's
    `self.models`.  Conv entries carry the state-dict key prefix."""
    ops, out_filters, prev, conv_id = [], [], 3, 0
    for block in blocks:
        t = block["type"]
        if t == "net":
            prev = int(block["channels"])
            continue
        ind = len(ops)
        if t == "convolutional":
            conv_id += 1
            k = int(block["size"])
            pad = int((k - 1) / 2) if int(block["pad"]) else 0
            op = dict(type="conv", id=conv_id, cin=prev, cout=int(block["filters"]), k=k,
                      stride=int(block["stride"]), pad=pad, bn=int(block["batch_normalize"]),
                      act=block["activation"], prefix="models.%d." % ind)
            prev = op["cout"]
        elif t == "maxpool":
            op = dict(type="maxpool", size=int(block["size"]), stride=int(block["stride"]))
        elif t == "reorg":
            op = dict(type="reorg", stride=int(block["stride"]))
            prev = prev * op["stride"] * op["stride"]
        elif t == "route":
            layers = [int(i) if int(i) > 0 else int(i) + ind for i in block["layers"].split(",")]
            if len(layers) == 1:
                prev = out_filters[layers[0]]
            else:
                assert len(layers) == 2 and layers[0] == ind - 1        # nets.py:855
                prev = out_filters[layers[0]] + out_filters[layers[1]]
            op = dict(type="route", layers=layers)
        elif t == "region":
            op = dict(type="region")
        else:
            raise NotImplementedError("block type %r is outside the YOLOv2 hot path" % t)
        out_filters.append(prev)
        ops.append(op)
    return ops


def init_state(blocks, seed=0, w_gain=2.0):
    """Seeded synthetic parameters with realistic scale:
    conv ~ N(0, w_gain/(k*k*cin)), gamma ~ U(.5,1.5), beta ~ N(0,.1),
    running_mean ~ N(0,.1), running_var ~ U(.5,1.5), bias ~ N(0,.1).
    Keys/order follow the reference state_dict."""
    g = torch.Generator().manual_seed(seed)
    st = OrderedDict()
    for op in plan(blocks):
        if op["type"] != "conv":
            continue
        p, i = op["prefix"], op["id"]
        std = (w_gain / (op["k"] * op["k"] * op["cin"])) ** 0.5
        st[p + "conv%d.weight" % i] = torch.randn(op["cout"], op["cin"], op["k"], op["k"], generator=g) * std
        if op["bn"]:
            st[p + "bn%d.weight" % i] = torch.rand(op["cout"], generator=g) + 0.5
            st[p + "bn%d.bias" % i] = torch.randn(op["cout"], generator=g) * 0.1
            st[p + "bn%d.running_mean" % i] = torch.randn(op["cout"], generator=g) * 0.1
            st[p + "bn%d.running_var" % i] = torch.rand(op["cout"], generator=g) + 0.5
            st[p + "bn%d.num_batches_tracked" % i] = torch.zeros((), dtype=torch.long)
        else:
            st[p + "conv%d.bias" % i] = torch.randn(op["cout"], generator=g) * 0.1
    return st


def param_keys(blocks):
    """Keys of the learnable parameters in `model.parameters()` order."""
    keys = []
    for op in plan(blocks):
        if op["type"] != "conv":
            continue
        p, i = op["prefix"], op["id"]
        keys.append(p + "conv%d.weight" % i)
        if op["bn"]:
            keys += [p + "bn%d.weight" % i, p + "bn%d.bias" % i]
        else:
            keys.append(p + "conv%d.bias" % i)
    return keys


# ---------------------------------------------------------------------------
# forward -- nets.py:720-774
# ---------------------------------------------------------------------------
def reorg(x, stride=2):
    """marvis ordering, nets.py:663-666: out channel = (hs*stride + ws)*C + c."""
    B, C, H, W = x.shape
    hs = ws = stride
    x = x.view(B, C, H // hs, hs, W // ws, ws).transpose(3, 4).contiguous()
    x = x.view(B, C, (H // hs) * (W // ws), hs * ws).transpose(2, 3).contiguous()
    x = x.view(B, C, hs * ws, H // hs, W // ws).transpose(1, 2).contiguous()
    return x.view(B, hs * ws * C, H // hs, W // ws)


class _StoreFp16(torch.autograd.Function):
    """A tensor that the HIP engine keeps in HBM as fp16: rounded on the way forward, and its
    gradient (also stored as fp16 by the engine) rounded on the way back."""

    @staticmethod
    def forward(ctx, t):
        return t.half().float()

    @staticmethod
    def backward(ctx, g):
        return g.half().float()


class _GradFp16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return g.half().float()


def border_map(table, H, W):
    """[16, C] border-class table -> [1, C, H, W] (class bits: 1 top row, 2 bottom row, 4 left column,
    8 right column; include/mcamd.h `mcamd_act_desc.border`)."""
    hh, ww = torch.arange(H), torch.arange(W)
    cls = ((hh == 0).long() + 2 * (hh == H - 1).long())[:, None] + (4 * (ww == 0).long() + 8 * (ww == W - 1).long())[None, :]
    return table[cls.reshape(-1)].t().reshape(1, table.shape[1], H, W)


def forward(blocks, state, x, training=False, masks=None, record=None, conv_out=None, storage=None, border=None):
    """Returns the last conv block's output (the region block is skipped,
    nets.py:761-762).  `masks`: list indexed by conv order, applied as
    `weight * mask` (layers.py:59).  `record`: dict filled with every block
    output by models index.  `conv_out`: dict filled with raw conv outputs
    (before BN) by conv id, kept in the autograd graph.

    `storage="fp16"`: same fp32 arithmetic, but every tensor the HIP engine stores in
    HBM as fp16 (input image, masked weights, raw conv outputs, block outputs and the
    gradients flowing back through them) is rounded to fp16 at that point.  Comparing the
    engine with THIS run isolates kernel errors from the (ill-conditioned) response of a
    BatchNorm/LeakyReLU network to fp16 rounding; comparing this run with the plain fp32
    run measures that response.

    `border`: {conv id: [16, Cout] table} added to the raw conv output by border class -- the slim
    (physically filter-pruned) models of modelcompression_amd/slim.py; no reference counterpart."""
    q = (lambda t: _StoreFp16.apply(t)) if storage == "fp16" else (lambda t: t)
    outputs, ci = {}, 0
    if storage == "fp16":
        x = x.half().float()
    for ind, op in enumerate(plan(blocks)):
        t = op["type"]
        if t == "conv":
            p, i = op["prefix"], op["id"]
            w = state[p + "conv%d.weight" % i]
            if masks is not None:
                w = w * masks[ci]
            ci += 1
            if storage == "fp16":
                w = w + (w.half().float() - w).detach()      # fp16 operand, fp32 master gradient
            bias = None if op["bn"] else state[p + "conv%d.bias" % i]
            x = F.conv2d(x, w, bias, op["stride"], op["pad"], 1, 1)
            if border is not None and i in border:       # slim models: folded constant input channels
                x = x + border_map(border[i].to(x.dtype), x.shape[2], x.shape[3])
            if storage == "fp16":
                x = q(x) if op["bn"] else _GradFp16.apply(x)  # logits leave in fp32; their gradient arrives as fp16
            if conv_out is not None:
                conv_out[i] = x
            if op["bn"]:
                x = F.batch_norm(x, state[p + "bn%d.running_mean" % i], state[p + "bn%d.running_var" % i],
                                 state[p + "bn%d.weight" % i], state[p + "bn%d.bias" % i],
                                 training, 0.1, 1e-5)
            if op["act"] == "leaky":
                x = F.leaky_relu(x, 0.1)
            elif op["act"] == "relu":
                x = F.relu(x)
            if op["bn"]:
                x = q(x)
        elif t == "maxpool":
            if op["stride"] > 1:
                x = F.max_pool2d(x, op["size"], op["stride"])
            else:
                x = F.max_pool2d(F.pad(x, (0, 1, 0, 1), mode="replicate"), 2, stride=1)
            x = q(x)
        elif t == "reorg":
            x = q(reorg(x, op["stride"]))
        elif t == "route":
            ls = op["layers"]
            x = outputs[ls[0]] if len(ls) == 1 else torch.cat((outputs[ls[0]], outputs[ls[1]]), 1)
        elif t == "region":
            continue
        outputs[ind] = x
        if record is not None:
            record[ind] = x
    return x


# ---------------------------------------------------------------------------
# Darknet .weights files -- nets.py:256-279, 897-948, 1007-1051
# ---------------------------------------------------------------------------
def save_weights(blocks, state, path, seen=0):
    """Header = int32 [0,0,0,seen] (nets.py:717,1012-1014), then per conv block
    bn.bias, bn.weight, running_mean, running_var, conv.weight (BN) or
    conv.bias, conv.weight (no BN), raw float32."""
    with open(path, "wb") as fp:
        np.array([0, 0, 0, seen], np.int32).tofile(fp)
        for op in plan(blocks):
            if op["type"] != "conv":
                continue
            p, i = op["prefix"], op["id"]
            if op["bn"]:
                names = ["bn%d.bias", "bn%d.weight", "bn%d.running_mean", "bn%d.running_var", "conv%d.weight"]
            else:
                names = ["conv%d.bias", "conv%d.weight"]
            for nme in names:
                state[p + nme % i].detach().numpy().astype(np.float32).tofile(fp)


def load_weights(blocks, state, path):
    """3 x int32 major/minor/revision, then `seen` as int64 when
    major*10+minor >= 2 (and both < 1000) else int32 (nets.py:899-905)."""
    with open(path, "rb") as fp:
        major, minor, _rev = np.fromfile(fp, np.int32, 3)
        wide = (major * 10 + minor >= 2) and major < 1000 and minor < 1000
        seen = int(np.fromfile(fp, np.int64 if wide else np.int32, 1)[0])
        for op in plan(blocks):
            if op["type"] != "conv":
                continue
            p, i = op["prefix"], op["id"]
            if op["bn"]:
                names = ["bn%d.bias", "bn%d.weight", "bn%d.running_mean", "bn%d.running_var", "conv%d.weight"]
            else:
                names = ["conv%d.bias", "conv%d.weight"]
            for nme in names:
                t = state[p + nme % i]
                buf = np.fromfile(fp, np.float32, t.numel())
                t.data.copy_(torch.from_numpy(buf).view_as(t))
    return seen


# ---------------------------------------------------------------------------
# one SGD step as torch.optim.SGD performs it -- train.py:144-147, 233-235
# ---------------------------------------------------------------------------
def sgd_step(params, grads, bufs, lr, momentum, weight_decay):
    """d = g + wd*p; buf = d (first step) or momentum*buf + d; p -= lr*buf
    (dampening 0, no nesterov).  In place; `bufs` entries may be None."""
    for j, (p, g) in enumerate(zip(params, grads)):
        d = g + weight_decay * p if weight_decay != 0 else g.clone()
        if bufs[j] is None:
            bufs[j] = d.clone()
        else:
            bufs[j].mul_(momentum).add_(d)
        p.sub_(lr * bufs[j])
