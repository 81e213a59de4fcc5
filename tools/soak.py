"""Soak run: N training steps with the real RegionLoss on a fixed synthetic batch (overfit): the loss must stay
finite and go down, the weights must stay finite, pruned weights must stay zero.
usage: python tools/soak.py [steps] [batch] [prune: none|weight80|filter40]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompression_amd import nets, YOLOV2_VOC_CFG
from modelcompression_amd.synthetic import init_synthetic, synthetic_batch
from modelcompression_amd.pruning.weightPruning.methods import weight_prune, quick_filter_prune
from modelcompression_amd.pruning.weightPruning.utils import are_masks_consistent

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
prune = sys.argv[3] if len(sys.argv) > 3 else "none"
dev = torch.device("cuda", 0)
model = init_synthetic(nets.Darknet(YOLOV2_VOC_CFG), 0).to(dev).train()
masks = None
if prune == "weight80":
    masks = weight_prune(model, 80.0)
elif prune == "filter40":
    masks = quick_filter_prune(model, 40.0)
if masks is not None:
    model.set_masks(masks)
x = synthetic_batch(B, 416, 416, seed=1, device=dev)
g = torch.Generator().manual_seed(3)
target = torch.zeros(B, 250)
for b in range(B):                      # 3 boxes per image: [cls, x, y, w, h]
    for k in range(3):
        target[b, 5 * k:5 * k + 5] = torch.tensor([float(torch.randint(0, 20, (1,), generator=g)), *(0.2 + 0.6 * torch.rand(2, generator=g)).tolist(),
                                                   *(0.1 + 0.3 * torch.rand(2, generator=g)).tolist()])
target = target.to(dev)
opt = torch.optim.SGD(model.parameters(), lr=1e-4 / B, momentum=0.9, weight_decay=0.0005 * B, fused=True)
losses = []
t0 = time.time()
for it in range(steps):
    out = model(x)
    loss = model.loss(out, target)
    opt.zero_grad()
    loss.backward()
    opt.step()
    if it % max(steps // 10, 1) == 0 or it == steps - 1:
        losses.append(float(loss.detach()))
        print("step %4d loss %.4f" % (it, losses[-1]), flush=True)
assert all(l == l and abs(l) < 1e9 for l in losses), "loss not finite"
assert all(bool(torch.isfinite(p).all()) for p in model.parameters()), "non-finite weights"
if masks is not None:
    assert are_masks_consistent(model, masks), "pruned weights moved"
print("ok: %d steps, %.1f s, loss %.4f -> %.4f%s" % (steps, time.time() - t0, losses[0], losses[-1],
                                                    "" if losses[-1] < losses[0] else "  (did NOT decrease)"))
