"""Small end-to-end check of the HIP path against the CPU oracle (used by __graft_entry__.smoke)."""
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MINI_CFG = os.path.join(ROOT, "tests", "golden", "mini.cfg")


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def smoke_mini(verbose=True):
    """One forward+backward of the mini YOLOv2-grammar network on cuda:0, compared with the
    oracle (fp32 PyTorch-CPU restatement of the reference).  Only smoke()/tests import oracle."""
    from . import nets
    from oracle import darknet_ref as O
    blocks = O.parse_cfg(MINI_CFG)
    state = O.init_state(blocks, seed=0)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(2, 3, 64, 64, generator=g)
    gout = torch.randn(2, 125, 16, 16, generator=g)

    st = {k: v.clone() for k, v in state.items()}
    keys = O.param_keys(blocks)
    for k in keys:
        st[k].requires_grad_(True)
    ref = O.forward(blocks, st, x, training=True)
    ref.backward(gout)

    res = {}
    for prec in ("auto", "fp16x3", "fp16"):        # auto = the training default ("mixed")
        model = nets.Darknet(MINI_CFG)
        model.load_state_dict(state)
        model.precision = prec
        model.cuda().train()
        out = model(x.cuda())
        out.backward(gout.cuda())
        e_out = rel_l2(out.detach(), ref.detach())
        worst = 0.0
        for (name, p) in model.named_parameters():
            worst = max(worst, rel_l2(p.grad, st[name].grad))
        res[prec] = (e_out, worst)
        if verbose:
            print("smoke [%s]: train logits rel-L2 %.2e, worst param-grad rel-L2 %.2e" % (prec, e_out, worst))
    # north_star: region-layer logits within 1e-3 of the reference -- met by the split-operand forward.  The plain
    # fp16 throughput mode is held to its storage floor (3 roundings per block, amplified by train-mode BN).
    assert res["auto"][0] < 1e-3 and res["fp16x3"][0] < 1e-3, res
    assert res["fp16"][0] < 3e-3, res
    # gradients through LeakyReLU kinks respond to fp16 storage as ~sqrt(eps): see tests/test_model_gpu.py
    assert res["auto"][1] < 0.2 and res["fp16x3"][1] < 0.2 and res["fp16"][1] < 0.2, res
    model.eval()          # default eval precision ("auto" -> "mixed")
    model.precision = "auto"
    with torch.no_grad():
        now = {k: v.detach().cpu() for k, v in model.state_dict().items()}    # running statistics moved by the step above
        e_eval = rel_l2(model(x.cuda()), O.forward(blocks, now, x, training=False))
    if verbose:
        print("smoke [eval, default precision]: logits rel-L2 %.2e" % e_eval)
    assert e_eval < 1e-3, e_eval
    return res["auto"]
