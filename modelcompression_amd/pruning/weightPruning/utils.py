"""Mirror of the reference's src/pruning/weightPruning/utils.py for the hot path.

`prune_rate`, `are_masks_consistent`, `arg_nonzero_min`, `to_var` keep the reference's
names, arguments and return values (utils.py:8-14, 59-133).  Counting/reduction over the
50 M weights runs in HIP (mcamd_count_zeros / mcamd_masked_residual); there is no CPU path.
The generic classifier `train`/`test` loops of the reference (utils.py:17-56) are not part of
the YOLOv2 path and are not provided.
"""
import numpy as np
import torch

from ... import ops


def to_var(x, requires_grad=False, volatile=False):
    """utils.py:8-14: move to the GPU when there is one.  `Variable` is a no-op wrapper in
    current PyTorch, so the tensor itself is returned."""
    if torch.cuda.is_available():
        x = x.cuda()
    return x.requires_grad_(requires_grad) if requires_grad else x


def prune_rate(model, verbose=True):
    """utils.py:59-93: 100 * (#zeros in params with dim != 1) / (#elements of ALL params)."""
    total_nb_param = 0
    nb_zero_param = 0
    layer_id = 0
    for parameter in model.parameters():
        param_this_layer = parameter.numel()
        total_nb_param += param_this_layer
        if parameter.dim() != 1:
            layer_id += 1
            zero_param_this_layer = ops.count_zeros([parameter.data.contiguous()])
            nb_zero_param += zero_param_this_layer
            if verbose:
                print("Layer {} | {} layer | {:.2f}% parameters pruned".format(
                    layer_id, 'Conv' if parameter.dim() == 4 else 'Linear',
                    100. * zero_param_this_layer / param_this_layer))
    pruning_perc = 100. * nb_zero_param / total_nb_param
    if verbose:
        print("Final pruning rate: {:.2f}%".format(pruning_perc))
    return pruning_perc


def arg_nonzero_min(a):
    """utils.py:96-120, quirks included: the seeding loop has no `break` (it ends on the LAST
    non-zero) and `if not min_ix` also fires when that index is 0 -> (inf, inf) + warning."""
    if not a:
        return
    min_ix, min_v = None, None
    for i, e in enumerate(a):
        if e != 0:
            min_ix = i
            min_v = e
    if not min_ix:
        print('Warning: all zero')
        return np.inf, np.inf
    for i, e in enumerate(a):
        if e < min_v and e != 0:
            min_v = e
            min_ix = i
    return min_v, min_ix


def are_masks_consistent(model, masks):
    """utils.py:122-133: every weight under a zero mask entry is still exactly zero."""
    conv_params = [p for p in model.parameters() if p.dim() == 4]
    assert len(conv_params) == len(masks)
    ws = [p.data.contiguous() for p in conv_params]
    ms = [m.to(w.device).contiguous() for m, w in zip(masks, ws)]
    return ops.masked_residual(ws, ms) == 0
