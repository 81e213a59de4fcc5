"""CPU restatement of /root/reference/src/pruning/weightPruning/{methods,utils}.py.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  numpy only; every function
takes plain numpy arrays (the list of model parameters in `model.parameters()`
order) instead of a torch module, so it can run without the reference.

The reference calls `np.percentile`, `np.sum`, `np.square`, `np.sqrt`,
`np.max` (numpy is a third-party dependency that is not under /root/reference;
the project pins no version -- the parity pin is numpy 2.2.6, the version the
golden fixtures were generated with).  The arithmetic those calls perform is
restated here explicitly (float32 virtual index, numpy's pairwise summation
order) so that the HIP kernels have an unambiguous target, and
tests/test_oracle_pruning.py checks each restatement against numpy itself.
"""
import math

import numpy as np


# ---------------------------------------------------------------------------
# np.percentile(arr, q), method="linear" -- reference call sites
# methods.py:18 (float32 data) and methods.py:55 (float64 data).
# ---------------------------------------------------------------------------
def virtual_index(n, perc, dtype):
    """(k, gamma, above) exactly as numpy 2.2 computes them for a `dtype` array.

    np.percentile divides q by `dtype.type(100)` when the data is floating
    (so q is float32 for float32 data), then computes `(n - 1) * q` with the
    Python int `n - 1` converted to that dtype (NEP 50 weak scalar), floors it,
    and takes gamma = v - floor(v) in the same dtype.
    """
    ft = np.dtype(dtype).type
    q = ft(perc) / ft(100)
    v = ft(n - 1) * q
    above = bool(v >= ft(n - 1))          # numpy clamps prev=next=last element
    k = int(math.floor(float(v)))
    gamma = ft(v - ft(k))
    return k, gamma, above


def lerp(a, b, t):
    """numpy's `_lerp` (used by method='linear'): a + (b-a)*t, and for t >= 0.5
    b - (b-a)*(1-t); every operation in the dtype of a/b/t."""
    ft = type(a)
    diff = ft(b - a)
    if t >= ft(0.5):
        return ft(b - ft(diff * ft(ft(1) - t)))
    return ft(a + ft(diff * t))


def percentile_linear(arr, perc):
    """Restatement of `np.percentile(arr, perc)` for a 1-D float32/float64 array."""
    arr = np.asarray(arr)
    assert arr.ndim == 1 and arr.dtype.kind == "f"
    n = arr.shape[0]
    k, gamma, above = virtual_index(n, perc, arr.dtype)
    if above:
        return arr.dtype.type(arr.max())
    part = np.partition(arr, [k, k + 1])
    return lerp(part[k], part[k + 1], gamma)


# ---------------------------------------------------------------------------
# weight_prune -- methods.py:9-26
# ---------------------------------------------------------------------------
def weight_prune_threshold(params, pruning_perc):
    """Global magnitude threshold (np.float32).  `params`: arrays in
    model.parameters() order; those with ndim == 1 are skipped (methods.py:16)."""
    flat = [np.abs(p).ravel() for p in params if p.ndim != 1]
    allw = np.concatenate(flat).astype(np.float32, copy=False)
    return percentile_linear(allw, pruning_perc)


def weight_prune(params, pruning_perc):
    """-> (masks, threshold).  mask = (|w| > threshold) as float32, strict '>'
    (methods.py:24-25); one mask per non-1-D parameter, in order."""
    thr = weight_prune_threshold(params, pruning_perc)
    masks = [(np.abs(p) > thr).astype(np.float32) for p in params if p.ndim != 1]
    return masks, thr


# ---------------------------------------------------------------------------
# numpy's float32 summation orders used by quick_filter_prune (methods.py:43-51)
# ---------------------------------------------------------------------------
def pairwise_sum_rows(a):
    """numpy `pairwise_sum_FLOAT` applied to every row of a [R, n] float32 array.

    n < 8: sequential; n <= 128: eight running lanes r[j] += a[i+j], combined as
    ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail sequentially; n > 128:
    split at n//2 rounded down to a multiple of 8 and recurse.
    """
    a = np.asarray(a, dtype=np.float32)
    n = a.shape[1]
    if n < 8:
        res = np.zeros(a.shape[0], np.float32)
        for i in range(n):
            res = res + a[:, i]
        return res
    if n <= 128:
        r = [a[:, j].copy() for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] = r[j] + a[:, i + j]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res = res + a[:, i]
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return pairwise_sum_rows(a[:, :n2]) + pairwise_sum_rows(a[:, n2:])


def sum_axis1_contiguous(a):
    """`a.sum(axis=1)` of a C-contiguous [R, n] float32 array as numpy's add.reduce
    performs it: one `pairwise_sum_FLOAT` call over all n elements of each row
    (checked against numpy for n = 2 ... 1280 in tests/test_oracle_pruning.py)."""
    return pairwise_sum_rows(np.asarray(a, dtype=np.float32))


def filter_mean_square(w):
    """`np.square(w).sum(axis=1).sum(axis=1).sum(axis=1) / (I*kh*kw)` for a
    float32 [O, I, kh, kw] array, in numpy's summation order (methods.py:43-44).

    kh*kw > 1: the Cin axis is strided, so numpy accumulates plane by plane,
    sequentially over Cin for each (o, kh, kw); then sequentially over kh, then
    over kw.  kh*kw == 1: the Cin axis is contiguous -> sum_axis1_contiguous.
    """
    w = np.asarray(w, dtype=np.float32)
    O, I, kh, kw = w.shape
    sq = w * w                                     # rounded to fp32 before any add
    if kh * kw == 1:
        s = sum_axis1_contiguous(sq.reshape(O, I))
    else:
        acc = sq[:, 0].copy()                      # [O, kh, kw]
        for i in range(1, I):
            acc = acc + sq[:, i]
        acc2 = acc[:, 0].copy()                    # [O, kw]
        for y in range(1, kh):
            acc2 = acc2 + acc[:, y]
        s = acc2[:, 0].copy()
        for x in range(1, kw):
            s = s + acc2[:, x]
    return (s / np.float32(I * kh * kw)).astype(np.float32)


def filter_scores(w):
    """Per-filter score of one conv layer (methods.py:43-51): mean square,
    divided by its own L2 norm over the layer, divided by the layer max."""
    v = filter_mean_square(w)
    norm = np.sqrt(sum_axis1_contiguous((v * v)[None, :])[0])
    v = (v / norm).astype(np.float32)
    return (v / np.max(v)).astype(np.float32)


def quick_filter_prune(params, pruning_perc):
    """-> (masks, info).  Restatement of methods.py:28-78.

    `values` is float64 (np.concatenate with the empty float64 seed list
    promotes, methods.py:34,53); the threshold is a float64 percentile and the
    comparison `score < threshold` promotes the float32 score to float64.
    info = {"scores": [...], "threshold": float64, "pruned": [index arrays]}.
    """
    convs = [p for p in params if p.ndim == 4]
    scores = [filter_scores(p) for p in convs]
    values = np.concatenate([np.zeros(0, np.float64)] + [s.astype(np.float64) for s in scores])
    thr = percentile_linear(values, pruning_perc)
    masks, pruned = [], []
    for p, s in zip(convs, scores):
        m = np.ones(p.shape, np.float32)
        drop = s.astype(np.float64) < thr
        m[drop] = 0.0
        masks.append(m)
        pruned.append(np.nonzero(drop)[0].astype(np.int64))
    return masks, {"scores": scores, "threshold": thr, "pruned": pruned}


# ---------------------------------------------------------------------------
# utils.py:59-133
# ---------------------------------------------------------------------------
def prune_rate(params):
    """utils.py:59-93: 100 * zeros in non-1-D params / elements of ALL params."""
    total = sum(int(p.size) for p in params)
    zeros = sum(int(np.count_nonzero(p == 0)) for p in params if p.ndim != 1)
    return 100.0 * zeros / total


def layer_prune_rates(params):
    """Per-layer percentages printed by prune_rate(verbose=True), utils.py:83-90."""
    return [100.0 * int(np.count_nonzero(p == 0)) / int(p.size) for p in params if p.ndim != 1]


def arg_nonzero_min(a):
    """utils.py:96-120, quirks kept: empty -> None; the seeding loop has no break
    so it ends on the LAST non-zero; `if not min_ix` also fires when that index
    is 0 -> (inf, inf)."""
    if not a:
        return None
    min_ix, min_v = None, None
    for i, e in enumerate(a):
        if e != 0:
            min_ix, min_v = i, e
    if not min_ix:
        return np.inf, np.inf
    for i, e in enumerate(a):
        if e < min_v and e != 0:
            min_v, min_ix = e, i
    return min_v, min_ix


def are_masks_consistent(params, masks):
    """utils.py:122-133: sum over conv params of sum(p * |m - 1|) == 0."""
    convs = [p for p in params if p.ndim == 4]
    assert len(convs) == len(masks)
    total = 0.0
    for p, m in zip(convs, masks):
        total += float((p * np.abs(m - 1)).sum(dtype=np.float32))
    return total == 0


def prune_one_filter(params, masks):
    """methods.py:81-125 (greedy variant; no /max step).  Returns (masks, layer, filt)."""
    no_masks = not masks
    if no_masks:
        masks = []
    values = []
    for p in params:
        if p.ndim == 4:
            if no_masks:
                masks.append(np.ones(p.shape, np.float32))
            v = filter_mean_square(p)
            norm = np.sqrt(sum_axis1_contiguous((v * v)[None, :])[0])
            v = (v / norm).astype(np.float32)
            mv, mi = arg_nonzero_min(list(v))
            values.append([mv, mi])
    assert len(masks) == len(values), "something wrong here"
    values = np.array(values)
    layer = int(np.argmin(values[:, 0]))
    filt = int(values[layer, 1])
    masks[layer][filt] = 0.0
    return masks, layer, filt


def filter_prune(params, pruning_perc):
    """methods.py:128-142: prune one filter at a time until prune_rate >= perc.
    `params` are modified in place the way Darknet.set_masks does (w *= mask)."""
    masks, cur, order = [], 0.0, []
    while cur < pruning_perc:
        masks, layer, filt = prune_one_filter(params, masks)
        order.append((layer, filt))
        convs = [p for p in params if p.ndim == 4]
        for p, m in zip(convs, masks):
            p *= m
        cur = prune_rate(params)
    return masks, order
