"""Layer-by-layer comparison of the engine's buffers with the oracle (debug aid, GPU box)."""
import os, sys
os.environ.setdefault("MCAMD_STEM_FUSED", "0")   # this tool inspects the per-layer y / dY buffers of every block
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from modelcompression_amd import nets, ops, YOLOV2_VOC_CFG
from oracle import darknet_ref as O
from util import rel_l2, raw_to_nchw, padded_to_nchw

cfg = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests/golden/mini.cfg")
if cfg == "yolo": cfg = YOLOV2_VOC_CFG
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
blocks = O.parse_cfg(cfg)
state = O.init_state(blocks, seed=1)
H = int(blocks[0]["height"]); W = int(blocks[0]["width"])
g = torch.Generator().manual_seed(7)
x = torch.rand(B, 3, H, W, generator=g)
m = nets.Darknet(cfg); m.load_state_dict(state); m.cuda().train()
out = m(x.cuda())
gout = torch.randn(out.shape, generator=g)
out.backward(gout.cuda())
eng = list(m._engines.values())[0]
S = eng.grad_scale
st = {k: v.clone() for k, v in state.items()}
for k in O.param_keys(blocks): st[k].requires_grad_(True)
rec, cout = {}, {}
ref = O.forward(blocks, st, x, training=True, record=rec, conv_out=cout, storage=os.environ.get('ORACLE_STORAGE') or None)
for t in list(rec.values()) + list(cout.values()): t.retain_grad()
ref.backward(gout)
print("logits", rel_l2(out.detach().cpu(), ref.detach()))
for lay in eng.layers:
    cid = lay.li + 1
    line = "conv%-2d idx %2d cin %4d cout %4d k%d %3dx%-3d mode %d |" % (cid, lay.index, lay.cin, lay.cout, lay.k, lay.H, lay.W, lay.mode)
    if not lay.is_last:
        y = raw_to_nchw(lay.y, B, lay.H, lay.W, lay.cout, lay.cout)
        line += " y %.1e" % rel_l2(y, cout[cid].detach())
        mean_ref = cout[cid].detach().mean((0, 2, 3)); var_ref = cout[cid].detach().var((0, 2, 3), unbiased=False)
        line += " mean %.1e istd %.1e" % (rel_l2(lay.mean.cpu(), mean_ref), rel_l2(lay.invstd.cpu(), 1 / torch.sqrt(var_ref + 1e-5)))
        t = lay.out_t
        act = padded_to_nchw(eng.bufs[t.buf], B, t.H, t.W, t.ld, t.C, t.choff)
        line += " out %.1e" % rel_l2(act, rec[lay.out_id].detach())
    dy = padded_to_nchw(lay.dy, B, lay.H, lay.W, lay.cout_p, lay.cout) / S
    line += " | dy %.1e" % rel_l2(dy, cout[cid].grad)
    if lay.gin is not None:
        gin = raw_to_nchw(lay.gin, B, lay.H, lay.W, lay.tin.ld, lay.tin.C, lay.tin.choff) / S
        src = lay.src
        # gradient wrt the conv's input tensor
        gref = None
        # find input tensor in oracle: re-run conv input? use autograd: rec[src] if materialised
        if src in rec and rec[src].grad is not None and lay.mode is not None:
            gref = rec[src].grad
        if gref is not None and gref.shape == gin.shape:
            line += " gin %.1e" % rel_l2(gin, gref)
    wg = dict(m.named_parameters())
    line += " | dW %.1e" % rel_l2(lay.conv.weight.grad.cpu(), st["models.%d.conv%d.weight" % (lay.index, cid)].grad)
    if lay.bn is not None:
        line += " dgam %.1e dbet %.1e" % (rel_l2(lay.bn.weight.grad.cpu(), st["models.%d.bn%d.weight" % (lay.index, cid)].grad),
                                           rel_l2(lay.bn.bias.grad.cpu(), st["models.%d.bn%d.bias" % (lay.index, cid)].grad))
    print(line)
