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

    model = nets.Darknet(MINI_CFG)
    model.load_state_dict(state)
    model.cuda().train()
    out = model(x.cuda())
    out.backward(gout.cuda())

    st = {k: v.clone() for k, v in state.items()}
    keys = O.param_keys(blocks)
    for k in keys:
        st[k].requires_grad_(True)
    ref = O.forward(blocks, st, x, training=True)
    ref.backward(gout)
    e_out = rel_l2(out.detach(), ref.detach())
    worst = 0.0
    for (name, p) in model.named_parameters():
        worst = max(worst, rel_l2(p.grad, st[name].grad))
    if verbose:
        print("smoke: logits rel-L2 %.2e, worst param-grad rel-L2 %.2e" % (e_out, worst))
    assert e_out < 3e-3, e_out
    # gradients through LeakyReLU kinks respond to fp16 storage as ~sqrt(eps): see tests/test_model_gpu.py
    assert worst < 0.2, worst
    return e_out, worst
