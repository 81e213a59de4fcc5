"""Physically slim export of a filter-pruned Darknet (BASELINE.json configs[4]: "filter pruning 60 %,
physically-removed channels (slim conv shapes), inference").

The reference never builds the slim network -- README.md:19 only states it as a conclusion, and
`quick_filter_prune` (src/pruning/weightPruning/methods.py:28-78) leaves the pruned filters in
place as all-zero rows.  This module removes them for real while keeping the logits of the
masked-dense model:

* A filter whose effective weights (`weight * mask`, layers.py:59) are all zero produces a raw conv
  output of exactly 0, so after eval-mode BatchNorm + LeakyReLU its channel is the CONSTANT
  `v = leaky(beta - gamma * running_mean / sqrt(running_var + eps))` at every pixel, through max
  pooling, reorg and route alike.
* The consumer convolution therefore sees `sum_c v_c * sum_taps W[n, c, tap]` from those channels --
  except where zero padding clips taps.  There are 16 border classes (top/bottom/left/right row or
  column clipped); the export folds the constant channels into a `[16, Cout]` table per conv
  (`border_bias`), which the engine adds to the raw conv output by pixel class
  (`mcamd_act_desc.border`, include/mcamd.h).  1x1 consumers get the same value in all 16 rows.
* What remains is a smaller Darknet: `filters=` reduced in the cfg, weights gathered to
  [kept_out, live_in, k, k], BN vectors gathered.  Kept counts are rounded up to a multiple of 8 by
  retaining a few zero filters (the fp16 kernels move 8 channels per lane; a retained zero filter is
  still exact).  Blocks without batch_normalize (the linear detection head) keep every filter.

The slim model is inference-only (its border tables are constants of the folded BN statistics).
Everything here is host logic on torch tensors (no kernels): it runs on whatever device the model's
parameters live on.
"""
import copy

import numpy as np
import torch

from . import nets

__all__ = ["write_cfg", "filter_keep_lists", "slim_export", "save_slim", "load_slim", "slim_summary"]


def write_cfg(blocks, path):
    """Inverse of `parse_cfg` (nets.py:39-73): one `[type]` header per block, `key=value` lines."""
    with open(path, "w") as fp:
        for b in blocks:
            fp.write("[%s]\n" % b["type"])
            for k, v in b.items():
                if k == "type":
                    continue
                fp.write("%s=%s\n" % ("type" if k == "_type" else k, v))
            fp.write("\n")


def _effective_weight(conv):
    w = conv.weight.data
    if getattr(conv, "mask_flag", False):
        w = w * conv.mask.to(w.device)
    return w


def filter_keep_lists(model, multiple=8):
    """Per conv block (cfg order): sorted LongTensor of the filter indices a slim export keeps --
    every filter with a non-zero effective weight, padded with the lowest-numbered zero filters up
    to a multiple of `multiple`; blocks without batch_normalize keep all their filters."""
    keeps = []
    for ind, block in enumerate(model.blocks[1:]):
        if block["type"] != "convolutional":
            continue
        conv = model.models[ind][0]
        w = _effective_weight(conv)
        n = w.shape[0]
        if not int(block["batch_normalize"]):
            keeps.append(torch.arange(n))
            continue
        alive = (w.reshape(n, -1) != 0).any(1).cpu()
        if not bool(alive.any()):
            raise ValueError("conv block %d: every filter is pruned; the network computes a constant" % ind)
        kept = torch.nonzero(alive).flatten()
        want = min(n, -(-kept.numel() // multiple) * multiple)
        if want > kept.numel():
            dead = torch.nonzero(~alive).flatten()[: want - kept.numel()]
            kept = torch.sort(torch.cat((kept, dead))).values
        keeps.append(kept)
    return keeps


class _Status:
    """Channel bookkeeping of one block output: which ORIGINAL channels are still computed (`live`,
    sorted) and the constant every other channel holds (`const`, fp32 [C])."""

    def __init__(self, C, live, const):
        self.C, self.live, self.const = C, live, const


def _class_tables(tap_sum, k):
    """tap_sum [N, k, k] (fp64) -> [16, N] sums over the taps a pixel of each border class can see."""
    N = tap_sum.shape[0]
    out = torch.empty(16, N, dtype=torch.float64)
    for cls in range(16):
        t = tap_sum
        if k == 3:
            rows = [r for r in range(3) if not ((cls & 1 and r == 0) or (cls & 2 and r == 2))]
            cols = [c for c in range(3) if not ((cls & 4 and c == 0) or (cls & 8 and c == 2))]
            t = tap_sum[:, rows][:, :, cols]
        out[cls] = t.reshape(N, -1).sum(1)
    return out


def slim_export(model, cfg_out, fp16_constants=True):
    """Build the physically slim counterpart of `model` (a Darknet whose pruned filters are zero rows,
    i.e. after `set_masks(quick_filter_prune(...))` or any equivalent zeroing).

    Writes the slim cfg to `cfg_out`, returns a new `Darknet` on the same device in eval mode, with
    gathered weights and a `border_bias` buffer on every conv that lost input channels.
    `fp16_constants`: round the constants to fp16 first, as the engine stores activations (this is
    what the masked-dense engine feeds its convolutions, so the two engines agree more tightly)."""
    blocks = model.blocks
    keeps = filter_keep_lists(model)
    dev = next(model.parameters()).device
    status = {}
    slim_blocks = copy.deepcopy(blocks)
    gathered = {}            # models index -> dict of tensors for the slim conv block
    ci = 0
    C0 = int(blocks[0]["channels"])
    prev = _Status(C0, torch.arange(C0), torch.zeros(C0))
    for ind, block in enumerate(blocks[1:]):
        t = block["type"]
        if t == "convolutional":
            seq = model.models[ind]
            conv = seq[0]
            w = _effective_weight(conv).detach().double().cpu()
            n, cin, k = w.shape[0], w.shape[1], w.shape[2]
            if cin != prev.C:
                raise ValueError("conv block %d expects %d input channels, its source has %d" % (ind, cin, prev.C))
            kept = keeps[ci]
            ci += 1
            is_const = torch.ones(cin, dtype=torch.bool)
            is_const[prev.live] = False
            g = {"weight": w[kept][:, prev.live].float()}
            if bool(is_const.any()):
                v = prev.const.double()[is_const]
                tap_sum = torch.einsum("nckl,c->nkl", w[kept][:, is_const], v)
                g["border"] = _class_tables(tap_sum, k).float()
            has_bn = int(block["batch_normalize"])
            slope = {"leaky": 0.1, "linear": 1.0}.get(block["activation"])
            if slope is None:
                raise NotImplementedError("activation %r" % block["activation"])
            const = torch.zeros(n)
            if has_bn:
                bn = seq[1]
                for name in ("weight", "bias", "running_mean", "running_var"):
                    g["bn_" + name] = getattr(bn, name).detach().cpu()[kept].clone()
                g["bn_nbt"] = bn.num_batches_tracked.detach().cpu().clone()
                # the engine's arithmetic (bn_coeffs_kernel): double, rounded to fp32 scale / shift
                invstd = 1.0 / torch.sqrt(bn.running_var.detach().double().cpu() + bn.eps)
                sc = (bn.weight.detach().double().cpu() * invstd).float()
                sh = (bn.bias.detach().double().cpu() - bn.running_mean.detach().double().cpu() * sc.double()).float()
                const = torch.where(sh > 0, sh, sh * torch.tensor(slope, dtype=torch.float32))
                if fp16_constants:
                    const = const.half().float()
            elif conv.bias is not None:
                g["bias"] = conv.bias.detach().cpu()[kept].clone()
                const = g["bias"].clone() if kept.numel() == n else torch.zeros(n)
            gathered[ind] = g
            slim_blocks[ind + 1]["filters"] = str(int(kept.numel()))
            prev = _Status(n, kept, const)
        elif t == "maxpool":
            pass                                   # max over equal constants is that constant
        elif t == "reorg":
            s = int(block["stride"])
            q = s * s
            live = torch.cat([prev.live + j * prev.C for j in range(q)])
            prev = _Status(q * prev.C, live, prev.const.repeat(q))
        elif t == "route":
            layers = [int(i) for i in block["layers"].split(",")]
            layers = [i if i > 0 else i + ind for i in layers]
            if len(layers) == 1:
                prev = status[layers[0]]
            else:
                a, b = status[layers[0]], status[layers[1]]
                prev = _Status(a.C + b.C, torch.cat((a.live, b.live + a.C)), torch.cat((a.const, b.const)))
        elif t in ("region", "cost", "softmax"):
            pass
        else:
            raise NotImplementedError("slim export: block type %r" % t)
        status[ind] = prev

    write_cfg(slim_blocks, cfg_out)
    slim = nets.Darknet(cfg_out)
    for ind, g in gathered.items():
        seq = slim.models[ind]
        conv = seq[0]
        if tuple(conv.weight.shape) != tuple(g["weight"].shape):
            raise RuntimeError("slim conv block %d: built %s, gathered %s" % (ind, tuple(conv.weight.shape), tuple(g["weight"].shape)))
        conv.weight.data.copy_(g["weight"])
        if "bias" in g:
            conv.bias.data.copy_(g["bias"])
        if "bn_weight" in g:
            bn = seq[1]
            bn.weight.data.copy_(g["bn_weight"])
            bn.bias.data.copy_(g["bn_bias"])
            bn.running_mean.copy_(g["bn_running_mean"])
            bn.running_var.copy_(g["bn_running_var"])
            bn.num_batches_tracked.copy_(g["bn_nbt"])
        if "border" in g:
            conv.register_buffer("border_bias", g["border"].contiguous())
    slim.seen = model.seen
    slim.header = model.header.clone()
    slim.to(dev)
    slim.eval()
    return slim


def save_slim(slim, prefix):
    """`prefix`.weights (Darknet format, nets.py:1007-1051) + `prefix`.border.npz (tables by models index).
    The cfg was written by `slim_export`."""
    slim.save_weights(prefix + ".weights")
    tables = {}
    for ind, m in enumerate(slim.models):
        if isinstance(m, torch.nn.Sequential) and getattr(m[0], "border_bias", None) is not None:
            tables["b%d" % ind] = m[0].border_bias.detach().cpu().numpy()
    np.savez(prefix + ".border.npz", **tables)


def load_slim(cfg, prefix, device=None):
    slim = nets.Darknet(cfg)
    slim.load_weights(prefix + ".weights")
    with np.load(prefix + ".border.npz") as z:
        for key in z.files:
            slim.models[int(key[1:])][0].register_buffer("border_bias", torch.from_numpy(z[key]).contiguous())
    if device is not None:
        slim.to(device)
    slim.eval()
    return slim


def slim_summary(model, slim):
    """[(conv id, filters before, after, input channels before, after)], dense and slim MACs per pixel-free unit."""
    rows = []
    cid = 0
    for ind, block in enumerate(model.blocks[1:]):
        if block["type"] != "convolutional":
            continue
        cid += 1
        a, b = model.models[ind][0].weight, slim.models[ind][0].weight
        rows.append((cid, a.shape[0], b.shape[0], a.shape[1], b.shape[1]))
    return rows
