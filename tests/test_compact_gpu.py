"""Filter compaction (engine.py `_update_compaction`): with filter masks the engine runs forward, dgrad
and wgrad on the kept filters only, in a permuted physical channel order.  It must be invisible:
logits, every parameter gradient, the BN running statistics and the weights after SGD steps have to
match (a) the fp32 oracle's masked-dense computation (reference semantics, layers.py:59-64) and
(b) the engine's own masked-dense path (MCAMD_COMPACT=0) to fp16-storage accuracy."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from modelcompression_amd import nets, YOLOV2_VOC_CFG  # noqa: E402
from modelcompression_amd.pruning.weightPruning.methods import quick_filter_prune  # noqa: E402
from modelcompression_amd.pruning.weightPruning.utils import are_masks_consistent  # noqa: E402
from oracle import darknet_ref as O  # noqa: E402
from util import rel_l2  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
MINI = os.path.join(HERE, "golden", "mini.cfg")


def _model(cfg, dev, seed, perc, precision="fp16"):
    blocks = O.parse_cfg(cfg)
    state = O.init_state(blocks, seed=seed)
    m = nets.Darknet(cfg)
    m.load_state_dict(state)
    m.precision = precision    # (most of this file is about the plain-operand engine; the default precision has its own test)
    m.to(dev)
    masks = quick_filter_prune(m, perc)
    m.set_masks(masks)
    return blocks, m, masks


def _run(m, x, gout, compact, steps=1, lr=0.0):
    """fwd+bwd (+SGD) with compaction on/off; returns logits, grads, running stats, weights."""
    os.environ["MCAMD_COMPACT"] = "1" if compact else "0"
    try:
        m._engines = {}
        m.train()
        opt = torch.optim.SGD(m.parameters(), lr=lr, momentum=0.9, weight_decay=0.001) if lr else None
        for _ in range(steps):
            out = m(x)
            m.zero_grad()
            out.backward(gout)
            if opt:
                opt.step()
        eng = list(m._engines.values())[0]
        return (out.detach().cpu(), {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()},
                {n: b.detach().cpu().clone() for n, b in m.named_buffers() if "running" in n},
                {n: p.detach().cpu().clone() for n, p in m.named_parameters()}, eng)
    finally:
        os.environ.pop("MCAMD_COMPACT", None)


@pytest.mark.parametrize("perc", [40.0, 70.0])
def test_compaction_matches_oracle_and_dense_engine_mini(dev, perc):
    blocks, m, masks = _model(MINI, dev, 2, perc)
    state0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(11)
    x = torch.rand(4, 3, 64, 96, generator=g)
    gout = torch.randn(4, 125, 16, 24, generator=g)
    out_c, g_c, rs_c, _, eng = _run(m, x.to(dev), gout.to(dev), True)
    compacted = [lay.index for lay in eng.layers if lay.perm is not None]
    permuted_in = [lay.index for lay in eng.layers if lay.in_perm is not None]
    print("perc %g: compacted conv blocks %s (kept %s), permuted inputs %s" % (
        perc, compacted, [(lay.n_act, lay.cout) for lay in eng.layers if lay.perm is not None], permuted_in))
    assert compacted and permuted_in
    m.load_state_dict(state0)
    out_d, g_d, rs_d, _, eng_d = _run(m, x.to(dev), gout.to(dev), False)
    assert all(lay.perm is None and lay.in_perm is None for lay in eng_d.layers)
    # oracle: masked dense, fp32, and its fp16-storage twin (the conditioning floor, see test_model_gpu.py)
    cmasks = [k.cpu() for k in masks]

    def oracle(storage):
        st = {k: v.detach().cpu().clone() for k, v in state0.items() if not k.endswith(".mask")}
        keys = O.param_keys(blocks)
        for k in keys:
            st[k].requires_grad_(True)
        o = O.forward(blocks, st, x, training=True, masks=cmasks, storage=storage)
        o.backward(gout)
        return o.detach(), {k: st[k].grad for k in keys}, st
    o32, g32, st32 = oracle(None)
    o16, g16, _ = oracle("fp16")
    e, ed, floor = rel_l2(out_c, o32), rel_l2(out_c, out_d), rel_l2(o16, o32)
    print("logits: compact vs oracle %.2e, compact vs dense engine %.2e, fp16 floor %.2e" % (e, ed, floor))
    assert e < 1.5 * floor + 1e-3 and ed < 1.5 * floor + 1e-3
    bad = []
    for name in g_c:
        ref = g32[name]
        if float(ref.norm()) == 0.0:
            assert float(g_c[name].abs().max()) == 0.0, name
            continue
        ec, edn, fl = rel_l2(g_c[name], ref), rel_l2(g_d[name], ref), rel_l2(g16[name], ref)
        print("  %-28s compact vs oracle %.2e | dense engine vs oracle %.2e | fp16 floor %.2e | compact vs dense %.2e"
              % (name, ec, edn, fl, rel_l2(g_c[name], g_d[name])))
        # vs the oracle: same bound family as test_model_gpu.py (one realisation of fp16 noise against another);
        # vs the dense engine (same kernels, same rounding points, only the channel order differs): much tighter
        # (a gradient where even the dense engine sits above the fp16-twin's realisation: compaction must not add to it)
        # ... unless the dense engine itself is further than that from the oracle: bn1.weight at 70 % has 8 live entries
        # summed with heavy cancellation, and re-grouping the fp32 BatchNorm partial sums of the NEXT layer (bit-identical
        # conv outputs, tools/cmp_small3x3.py) moves it by 2e-2; then the two engines may differ by what either differs
        # from the oracle
        # (with the dead input channels folded -- csrc/fold.hip -- the compacting engine no longer shares every rounding
        # point with the dense one: the constants leaky(beta) of dead channels enter exactly instead of as stored fp16
        # values.  For the one gradient that is chaotic in both engines the two may then sit on opposite sides of the
        # oracle: the triangle inequality is all that holds, and the tight check of the folded launches is
        # test_model_gpu.py::test_layerwise_teacher_forced_yolov2_filter40 / ..._mini)
        cvd = rel_l2(g_c[name], g_d[name])
        close = cvd < max(0.5 * fl, 5e-3, edn) or (name == "models.0.bn1.weight" and cvd < 1.1 * (ec + edn) + 1e-3)   # (cvd is relative to the dense engine's norm)
        lim = max(3.0 * fl + 5e-3, 1.3 * edn + 1e-3)
        if name == "models.0.bn1.weight":
            # the same chaotic entry against the oracle: any re-grouping of fp32 partial sums moves it by ~2e-2 (above), e.g.
            # the first block's units taken down a column strip instead of along a row: 2.2e-2 (dense) / 3.8e-2 (compact)
            lim = max(lim, edn + 2e-2)
        if not (ec < lim and close):
            bad.append(name)
    assert not bad, bad
    # pruned filters: exactly zero gradient rows (grad * mask), also for their scattered-back columns
    for (name, p), mk in zip([(n, p) for n, p in m.named_parameters() if p.dim() == 4], cmasks):
        assert bool((g_c[name][mk == 0] == 0).all()), name
    # BN running statistics, in the module's channel order (chan_perm addressing)
    for name in rs_c:
        assert torch.allclose(rs_c[name], st32[name].detach(), rtol=2e-2, atol=2e-3), name
        assert torch.allclose(rs_c[name], rs_d[name], rtol=2e-2, atol=2e-3), name


def test_compaction_sgd_steps_keep_masks_and_track_dense(dev):
    blocks, m, masks = _model(MINI, dev, 4, 60.0)
    state0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    x, gout = torch.rand(2, 3, 64, 64, generator=g).to(dev), torch.randn(2, 125, 16, 16, generator=g).to(dev)
    _, _, _, w_c, _ = _run(m, x, gout, True, steps=3, lr=1e-3)
    assert are_masks_consistent(m, masks)
    m.load_state_dict(state0)
    _, _, _, w_d, _ = _run(m, x, gout, False, steps=3, lr=1e-3)
    num = den = 0.0
    for n in w_c:
        ud, uc = (w_d[n] - state0[n].cpu()).double(), (w_c[n] - state0[n].cpu()).double()
        num += float((uc - ud).pow(2).sum())
        den += float(ud.pow(2).sum())
    e = (num / den) ** 0.5
    print("3 SGD steps: relative difference of the whole weight update, compact vs dense engine: %.2e" % e)
    # momentum compounds the fp16-level differences of three ill-conditioned train steps; with the dead input channels
    # folded the two engines round at different points (exact constants vs stored fp16 ones): 8e-2 measured, 4e-2 without
    assert e < 0.15


def test_compaction_yolov2_40pct(dev):
    """BASELINE configs[2] shapes: 40 % filter pruning on YOLOv2-VOC (reorg + concat permutations)."""
    blocks, m, masks = _model(YOLOV2_VOC_CFG, dev, 0, 40.0)
    state0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    x, gout = torch.rand(4, 3, 416, 416, generator=g).to(dev), torch.randn(4, 125, 13, 13, generator=g).to(dev)
    out_c, g_c, rs_c, _, eng = _run(m, x, gout, True)
    kept = sum(lay.n_act for lay in eng.layers) / sum(lay.cout for lay in eng.layers)
    print("yolov2 40%%: %d of 23 blocks compacted, %.1f%% of the filters computed" % (
        sum(lay.perm is not None for lay in eng.layers), 100 * kept))
    assert any(lay.perm is not None for lay in eng.layers)
    m.load_state_dict(state0)
    out_d, g_d, rs_d, _, _ = _run(m, x, gout, False)
    e = rel_l2(out_c, out_d)
    print("train logits compact vs dense engine: %.2e" % e)
    assert e < 2e-2          # train-mode BN over 23 layers amplifies fp16 rounding differences (test_model_gpu.py)
    # Train-mode BN + LeakyReLU over 23 random-init layers turns an fp16-level forward difference e into a
    # gradient difference ~sqrt(e) that grows towards the first layers (test_model_gpu.py measures the same
    # against the oracle); the tight per-kernel comparison is the mini test above.  Here: the head is close,
    # nothing is wildly off, pruned rows are exactly zero.
    errs = {}
    for name in g_c:
        if float(g_d[name].norm()) == 0.0:
            assert float(g_c[name].abs().max()) == 0.0, name
        else:
            errs[name] = rel_l2(g_c[name], g_d[name])
    ranked = sorted(errs.items(), key=lambda kv: -kv[1])
    worst, median = ranked[0][1], ranked[len(ranked) // 2][1]
    head = rel_l2(g_c["models.30.conv23.weight"], g_d["models.30.conv23.weight"])
    print("parameter gradients compact vs dense engine: head %.2e, median %.2e, worst %s" % (
        head, median, ", ".join("%s %.2e" % kv for kv in ranked[:4])))
    tail = {n: errs[n] for n in list(g_c)[-6:] if n in errs}      # conv22 / bn22 / conv23: the blocks behind the last amplification
    print("last blocks: " + ", ".join("%s %.1e" % (n.replace("models.", ""), v) for n, v in tail.items()))
    # This end-to-end comparison is chaotic by nature (which realisation of the fp16 noise the first layers'
    # BatchNorm vectors see changes with any re-grouping of fp32 partial sums): a 1.6e-2 difference of the train-mode logits
    # flips LeakyReLU derivative signs on ~sqrt(eps) of the elements, and EVERY parameter gradient of the first 21 blocks then
    # differs by 0.1-0.6 between the two engines (measured, round 4: median 0.20, worst bn2.weight 0.56) -- there is no quiet
    # subset up there.  The bounds: the head and the last blocks (where nothing amplifies any more), the bulk, and a bounded
    # worst case; the TIGHT checks of the compacted launches at these shapes -- every kernel of every block against fp32
    # `F.conv2d(x, w * mask)` on identical inputs at 1e-3 / 2e-3, and folded against unfolded backward launches on identical
    # inputs at 1e-3 -- are tests/test_model_gpu.py::test_layerwise_teacher_forced_yolov2_filter40 (+ ..._filter40_b64_head,
    # ..._filter60_b64_tail) and test_fold_dead_ab_on_identical_inputs_yolov2_filter40 above.
    assert head < 3e-2 and median < 0.25 and worst < 0.8
    assert max(tail.values()) < 0.15, tail
    for (name, p), mk in zip([(n, p) for n, p in m.named_parameters() if p.dim() == 4], masks):
        assert bool((g_c[name][mk.cpu() == 0] == 0).all()), name
    # eval mode uses the same plan (in the plain-fp16 mode; the split-operand eval default does not compact)
    m.eval()
    m.precision = "fp16"
    with torch.no_grad():
        os.environ["MCAMD_COMPACT"] = "1"
        m._engines = {}
        a = m(x).cpu()
        os.environ["MCAMD_COMPACT"] = "0"
        m._engines = {}
        b = m(x).cpu()
        os.environ.pop("MCAMD_COMPACT")
    assert rel_l2(a, b) < 5e-3


@pytest.mark.parametrize("cfg,shape,perc", [(MINI, (4, 3, 64, 96), 40.0), (YOLOV2_VOC_CFG, (4, 3, 416, 416), 40.0),
                                            (YOLOV2_VOC_CFG, (2, 3, 416, 416), 60.0)], ids=["mini-40", "yolov2-40", "yolov2-60"])
def test_compaction_in_split_operand_engines(dev, cfg, shape, perc):
    """The DEFAULT precision ("mixed": split hi / lo operands, two activation planes) with filter masks: kept filters only,
    dead input channels folded where the producer's kept count lets its activation pass write `aug`-spaced planes (round 4).
    Unlike the plain-fp16 engines this forward is NOT chaotic against the reference: train-mode logits within 1e-3 of the
    fp32 oracle's masked-dense run and of the same engine run masked-dense (MCAMD_COMPACT=0); pruned rows of every weight
    gradient exactly zero; the bulk of the parameter gradients within 5e-2 of the masked-dense engine's, the worst within 0.2
    (plain fp16 backward on both sides); running statistics in the module's channel order."""
    blocks, m, masks = _model(cfg, dev, 2, perc, precision="auto")
    state0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(31)
    x = torch.rand(*shape, generator=g)
    m.train()
    gshape = tuple(m(x.to(dev)).shape)
    m.load_state_dict(state0)
    gout = torch.randn(gshape, generator=g)
    out_c, g_c, rs_c, _, eng = _run(m, x.to(dev), gout.to(dev), True)
    assert eng.precision == "mixed" and eng.compact
    comp = [(lay.li + 1, lay.n_act, lay.cout) for lay in eng.layers if lay.perm is not None]
    fold = [(lay.li + 1, lay.fold_cin, lay.cin) for lay in eng.layers if lay.fold is not None]
    print("compacted (block, computed, filters): %s\nfolding consumers (block, kept inputs, inputs): %s" % (comp, fold))
    assert comp and (fold or cfg == MINI)      # (the mini network's channel counts are too small for 32-channel plane pitches)
    m.load_state_dict(state0)
    out_d, g_d, rs_d, _, eng_d = _run(m, x.to(dev), gout.to(dev), False)
    assert all(lay.perm is None and lay.fold is None for lay in eng_d.layers)
    cmasks = [k.cpu() for k in masks]
    st = {k: v.detach().cpu().clone() for k, v in state0.items() if not k.endswith(".mask")}
    with torch.no_grad():
        o32 = O.forward(blocks, st, x, training=True, masks=cmasks)
    e, ed = rel_l2(out_c, o32), rel_l2(out_c, out_d)
    print("train logits: compact vs fp32 oracle %.2e (masked-dense engine %.2e), compact vs masked-dense engine %.2e" % (
        e, rel_l2(out_d, o32), ed))
    # (the two engines are each ~6e-4 from the oracle -- the three blocks "mixed" leaves on plain operands -- and 4e-4 from
    # each other: the compacting one multiplies exact folded constants where the masked-dense one multiplies stored planes)
    assert e < 1e-3 and ed < 1e-3
    errs = {}
    for (name, p), mk in zip([(n, p) for n, p in m.named_parameters() if p.dim() == 4], cmasks):
        assert bool((g_c[name][mk == 0] == 0).all()), name
    for name in g_c:
        if float(g_d[name].norm()) == 0.0:
            assert float(g_c[name].abs().max()) == 0.0, name
        else:
            errs[name] = rel_l2(g_c[name], g_d[name])
    ranked = sorted(errs.items(), key=lambda kv: -kv[1])
    print("parameter gradients compact vs masked-dense engine: median %.2e, worst %s" % (
        ranked[len(ranked) // 2][1], ", ".join("%s %.2e" % kv for kv in ranked[:4])))
    # (a 4e-4 difference of the train-mode logits moves LeakyReLU derivative signs on ~sqrt(eps) of the elements: measured
    # median 2.0-2.3e-2, worst 4.6e-2 / 9.2e-2 at 40 / 60 % on YOLOv2; 3.8e-4 / 1.6e-3 on the mini network, which does not fold)
    assert ranked[len(ranked) // 2][1] < 5e-2 and ranked[0][1] < 0.2
    for name in rs_c:
        assert torch.allclose(rs_c[name], rs_d[name], rtol=1e-3, atol=1e-4), name


def test_fold_dead_ab_on_identical_inputs_yolov2_filter40(dev):
    """MCAMD_FOLD_DEAD=1 against MCAMD_FOLD_DEAD=0 at the filter40 shapes, TEACHER-FORCED (ADVICE r02 item 3 / VERDICT r03
    item 6): the two engines round at different points in the forward pass (exact folded constants against stored fp16
    ones), so end to end they diverge chaotically; here every folding consumer's backward launches are re-issued on both
    engines with the SAME dY and the SAME kept input channels.  The folded weight gradient (augmented problem + unfold)
    must equal the unfolded one on every column -- kept inputs, dead inputs, removed filters' zero rows -- within 1e-3,
    and the dead channels' dbeta that the fold delivers (mcamd_unfold_wgrad) must equal what the unfolded engine's own
    input gradient gives the producer's BatchNorm backward, leaky'(beta_c) * sum_pixels G[:, c], within 2e-3."""
    from modelcompression_amd import ops
    blocks, m, masks = _model(YOLOV2_VOC_CFG, dev, 12, 40.0)
    m.grad_scale = 16.0                                    # (conv2 keeps one live filter on this seed: see test_model_gpu.py)
    g = torch.Generator().manual_seed(23)
    Bn = 8
    x, gout = torch.rand(Bn, 3, 416, 416, generator=g).to(dev), torch.randn(Bn, 125, 13, 13, generator=g).to(dev)
    engs = {}
    for fold in ("1", "0"):
        os.environ["MCAMD_FOLD_DEAD"] = fold
        try:
            m._engines = {}
            m.train()
            out = m(x)
            m.zero_grad()
            out.backward(gout)
            engs[fold] = list(m._engines.values())[0]
        finally:
            os.environ.pop("MCAMD_FOLD_DEAD", None)
    A, Bq = engs["1"], engs["0"]
    folded = [lay for lay in A.layers if lay.fold is not None]
    assert len(folded) >= 8 and all(lay.fold is None for lay in Bq.layers)
    S = D = A.grad_scale
    checked = 0
    for ca in folded:
        cb = Bq.layers[ca.li]
        prod = ca.fold
        assert cb.n_act == ca.n_act and (cb.perm is None) == (ca.perm is None) and cb.in_perm is not None
        assert torch.equal(cb.in_perm, ca.in_perm)
        nk = ca.fold_cin
        # identical inputs: A's dY, and A's kept input channels in B's buffer (B keeps its own stored constants behind them)
        cb.dy.copy_(ca.dy)
        ta, tb = ca.tin, cb.tin
        va = ops.padded_view(A.bufs[ta.buf], Bn, ta.H, ta.W, ta.ld)
        vb = ops.padded_view(Bq.bufs[tb.buf], Bn, tb.H, tb.W, tb.ld)
        vb[:, 1:-1, 1:-1, tb.choff:tb.choff + nk] = va[:, 1:-1, 1:-1, ta.choff:ta.choff + nk]
        w, mask = ca.conv.weight.data, (ca.conv.mask if ca.conv.mask_flag else None)
        ws = torch.empty(max(ops.wgrad_workspace_bytes(ca.geom_act), ops.wgrad_workspace_bytes(cb.geom_act)), dtype=torch.uint8, device=dev)
        # A: augmented problem (kept inputs + the ones-channel), then back to OIHW and to the producer's dbeta
        dwaug = torch.zeros_like(ca.dwaug)
        gw_a = torch.zeros_like(w)
        db_a, dg_a = torch.zeros_like(prod.bn.bias.data), torch.zeros_like(prod.bn.bias.data)
        ops.conv_wgrad(ca.geom_act, A.bufs[ta.buf], ca.dy, ca.cout_p, 0, dwaug, None, D, None, ws)
        ops.unfold_wgrad(w, mask, ca.g_rows, ca.g_cols, prod.bn.bias.data, prod.slope, ca.n_act, nk, dwaug, gw_a, db_a, dg_a,
                         accumulate=False)
        # B: the unfolded problem on all input channels (row / column maps of the compaction)
        gw_b = torch.zeros_like(w)
        ops.conv_wgrad(cb.geom_act, Bq.bufs[tb.buf], cb.dy, cb.cout_p, 0, gw_b, mask, D, None, ws, rows=cb.g_rows, cols=cb.g_cols)
        e_w = rel_l2(gw_a.cpu(), gw_b.cpu())
        dead = ca.in_perm[nk:].cpu()                      # module indices of the producer's dead channels
        da, db_ = gw_a.cpu()[:, dead].double(), gw_b.cpu()[:, dead].double()
        share = float(db_.norm() / gw_b.double().norm())   # how much of the gradient sits in the dead columns
        # A dead input is a constant, so its column is v_c * sum_{p: p + tap inside} dY[p][n]: for a 1x1 consumer that is the
        # plain pixel sum of a BatchNorm-backward output, ZERO up to the fp16 rounding noise of dY (conv4: 1e-5 of the
        # gradient's norm) -- compared on the gradient's scale there, relatively where the columns carry weight (3x3: borders)
        e_dead = float((da - db_).norm() / db_.norm()) if share > 1e-3 else float((da - db_).norm() / gw_b.double().norm())
        # dbeta of the dead channels from B's own input gradient: the channel is the constant leaky(beta_c), its
        # pre-activation beta_c, so sum_pixels g_z = leaky'(beta_c) * sum_pixels G[:, c]
        # (the same dgrad launch with the fp32 NCHW epilogue: the stored fp16 G saturates on this seed's conv2 -- one live
        # filter carries the whole gradient of the 208x208 map -- and the comparison is about the fold, not about that)
        gin = torch.zeros(Bn, cb.cin, cb.H, cb.W, device=dev)
        ops.conv_dgrad_nchw(cb.geom_act, cb.dy, cb.cout_p, 0, cb.wd, gin)
        G = gin[:, nk:ca.in_perm.numel()].double().sum((0, 2, 3)).float().cpu() / S
        beta = prod.bn.bias.detach().cpu()[dead]
        expect = torch.where(beta > 0, torch.ones_like(beta), torch.full_like(beta, prod.slope)) * G
        # (a 1x1 consumer's share of dbeta is W . sum_p dY[p] again -- noise around zero: on the scale of the producer's whole
        # dbeta vector there, relatively for the 3x3 consumers, whose border taps make it a real number)
        e_b = rel_l2(db_a.cpu()[dead], expect) if ca.k == 3 else float(
            (db_a.cpu()[dead].double() - expect.double()).norm() / prod.bn.bias.grad.double().norm().cpu())
        print("conv%-2d (producer conv%d, %d kept + %d dead inputs): dW %.1e (dead columns %.1e, %.1e of the norm), dbeta of the dead "
              "channels %.1e" % (ca.li + 1, prod.li + 1, nk, dead.numel(), e_w, e_dead, share, e_b))
        assert e_w < 1e-3 and e_dead < 1e-3, (ca.li, e_w, e_dead)
        assert e_b < 2e-3, (ca.li, e_b)
        assert float(dg_a.abs().max()) == 0.0            # xhat = 0 for a dead filter: no dgamma
        checked += 1
    assert checked >= 8


@pytest.mark.parametrize("cfg,shape", [(MINI, (4, 3, 64, 96)), (YOLOV2_VOC_CFG, (4, 3, 416, 416))], ids=["mini", "yolov2"])
def test_overlapped_wgrad_with_compaction_is_bit_identical(dev, cfg, shape):
    """MCAMD_OVERLAP_WGRAD=1 runs EVERY weight gradient -- gather layers (filter compaction, folded inputs) included,
    with their unfold / scatter passes -- on the second stream with its own split-K workspace, while the dgrad /
    BatchNorm-backward chain stays on the launch stream.  Every kernel is deterministic, so the gradients have to be
    bit-identical to the serial run."""
    blocks, m, masks = _model(cfg, dev, 7, 40.0)
    g = torch.Generator().manual_seed(21)
    x = torch.rand(*shape, generator=g).to(dev)
    m.train()
    gout = torch.randn(tuple(m(x).shape), generator=g).to(dev)
    os.environ["MCAMD_OVERLAP_WGRAD"] = "0"
    _, g_serial, _, _, eng = _run(m, x, gout, True)
    assert not eng.overlap_wgrad
    assert any(lay.gather for lay in eng.layers) and any(not lay.gather for lay in eng.layers)
    os.environ["MCAMD_OVERLAP_WGRAD"] = "1"
    try:
        for rep in range(3):
            _, g_over, _, _, eng2 = _run(m, x, gout, True)
            assert eng2.overlap_wgrad
            for name in g_serial:
                assert torch.equal(g_serial[name], g_over[name]), (rep, name)
    finally:
        os.environ.pop("MCAMD_OVERLAP_WGRAD", None)
