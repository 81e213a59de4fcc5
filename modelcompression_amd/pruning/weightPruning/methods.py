"""Mirror of the reference's src/pruning/weightPruning/methods.py: weight_prune,
quick_filter_prune, prune_one_filter, filter_prune -- same signatures, masks bit-identical.

The 50.6 M-element scans run in HIP (mcamd_kth_magnitude, mcamd_magnitude_mask,
mcamd_filter_scores); the host only does the scalar percentile bookkeeping of
`np.percentile` (float32 virtual index for weight_prune, float64 interpolation over the
10 461 filter scores for quick_filter_prune).  Parameters must live on the GPU.
"""
import math

import numpy as np
import torch

from ... import ops
from ..._lib import McamdError
from .utils import prune_rate, arg_nonzero_min


def _virtual_index(n, perc, ftype):
    """np.percentile's index arithmetic for a `ftype` array (numpy 2.x): q = perc/100 in
    ftype, v = ftype(n-1) * q, k = floor(v), gamma = v - k (reference call sites
    methods.py:18 and :55)."""
    q = ftype(perc) / ftype(100)
    v = ftype(n - 1) * q
    k = int(math.floor(float(v)))
    return k, ftype(v - ftype(k)), bool(v >= ftype(n - 1))


def _lerp(a, b, t):
    ft = type(a)
    d = ft(b - a)
    if t >= ft(0.5):
        return ft(b - ft(d * ft(ft(1) - t)))
    return ft(a + ft(d * t))


def _prunable(model):
    ps = [p for p in model.parameters() if p.dim() != 1]
    for p in ps:
        if not p.is_cuda:
            raise McamdError("pruning needs the model on the GPU (model.cuda()); there is no CPU path")
    return ps


def weight_prune(model, pruning_perc):
    '''
    Prune pruning_perc% weights globally (not layer-wise)
    arXiv: 1606.09274                                     (reference methods.py:9-26)
    '''
    ps = _prunable(model)
    ws = [p.data.contiguous() for p in ps]
    n = sum(w.numel() for w in ws)
    k, gamma, above = _virtual_index(n, pruning_perc, np.float32)
    if above:
        k = n - 1
    pair = ops.kth_magnitude(ws, k).cpu().numpy()          # s[k], s[k+1]
    threshold = np.float32(pair[1]) if above else _lerp(np.float32(pair[0]), np.float32(pair[1]), gamma)
    thr = torch.tensor([float(threshold)], dtype=torch.float32, device=ws[0].device)
    return [ops.magnitude_mask(w, thr) for w in ws]


def _layer_scores(p):
    return ops.filter_scores(p.data.contiguous())


def _percentile_f64(values, perc):
    """np.percentile(values, perc) for a float64 vector (method 'linear')."""
    n = values.shape[0]
    k, gamma, above = _virtual_index(n, perc, np.float64)
    s = np.sort(values)
    if above:
        return np.float64(s[-1])
    return _lerp(np.float64(s[k]), np.float64(s[k + 1]), gamma)


def quick_filter_prune(model, pruning_perc):
    '''
    Prune pruning_perc% filters globally                   (reference methods.py:28-78)
    '''
    convs = [p for p in model.parameters() if p.dim() == 4]
    for p in convs:
        if not p.is_cuda:
            raise McamdError("pruning needs the model on the GPU (model.cuda()); there is no CPU path")
    scores = [_layer_scores(p) for p in convs]
    # one device->host read of all 10 461 scores, one host->device write of the keep flags (the reference moves every
    # weight tensor to the host twice, methods.py:37-72)
    values32 = torch.cat(scores).cpu().numpy()
    values = np.concatenate([np.zeros(0, np.float64), values32.astype(np.float64)])
    threshold = _percentile_f64(values, pruning_perc)
    keep_all = torch.from_numpy((~(values < threshold)).astype(np.int32)).to(convs[0].device)
    masks, off = [], 0
    for p in convs:
        masks.append(ops.filter_mask(keep_all[off:off + p.shape[0]], tuple(p.shape)))
        off += p.shape[0]
    return masks


def prune_one_filter(model, masks):
    '''
    Pruning one least ``important'' feature map by the scaled l2norm of
    kernel weights.  arXiv:1611.06440                      (reference methods.py:81-125)
    '''
    NO_MASKS = False
    if not masks:
        masks = []
        NO_MASKS = True
    values = []
    for p in model.parameters():
        if p.dim() == 4:
            if not p.is_cuda:
                raise McamdError("pruning needs the model on the GPU (model.cuda()); there is no CPU path")
            if NO_MASKS:
                masks.append(torch.ones_like(p.data))
            # scaled mean-square normalised by the layer's L2 norm; no /max step in this variant
            v = _scores_without_max(p)
            min_value, min_ind = arg_nonzero_min(list(v))
            values.append([min_value, min_ind])
    assert len(masks) == len(values), "something wrong here"
    values = np.array(values)
    to_prune_layer_ind = np.argmin(values[:, 0])
    to_prune_filter_ind = int(values[to_prune_layer_ind, 1])
    masks[to_prune_layer_ind][to_prune_filter_ind] = 0.
    print('Prune filter #{} in layer #{}'.format(to_prune_filter_ind, to_prune_layer_ind))
    return masks


def _scores_without_max(p):
    """mean square / L2 norm of the layer (methods.py:104-109), fp32 in numpy's order.
    The per-filter mean squares come from the HIP kernel; the O(cout) normalisation is
    finished on the host with the same pairwise sum numpy uses."""
    ms = ops.filter_mean_square(p.data.contiguous()).cpu().numpy()
    norm = np.sqrt(np.square(ms).sum())
    return ms / norm


def filter_prune(model, pruning_perc):
    '''
    Prune filters one by one until reach pruning_perc      (reference methods.py:128-142)
    (not iterative pruning)

    Note: on a Darknet the reference's loop never terminates -- its prune_one_filter builds
    numpy masks, MaskedConv2d.set_mask's register_buffer rejects them and Darknet.set_masks
    swallows the error (nets.py:1059-1060), so prune_rate never moves.  Here the masks are
    tensors, set_masks applies them, and the loop ends as the docstring intends.
    '''
    masks = []
    current_pruning_perc = 0.
    while current_pruning_perc < pruning_perc:
        masks = prune_one_filter(model, masks)
        model.set_masks(masks)
        current_pruning_perc = prune_rate(model, verbose=False)
        print('{:.2f} pruned'.format(current_pruning_perc))
    return masks
