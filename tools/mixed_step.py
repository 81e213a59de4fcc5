"""A few training steps (fwd + bwd + fused SGD, mean-of-logits loss) at one operand precision, for kernel traces.
usage: python tools/mixed_step.py [precision=mixed] [B=64] [steps=10]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
from modelcompression_amd import nets, YOLOV2_VOC_CFG
from modelcompression_amd.synthetic import init_synthetic, synthetic_batch

prec = sys.argv[1] if len(sys.argv) > 1 else "mixed"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda", 0)
model = nets.Darknet(YOLOV2_VOC_CFG)
init_synthetic(model, seed=0)
model.to(dev).train()
model.precision = prec
x = synthetic_batch(B, 416, 416, seed=1, device=dev)
opt = torch.optim.SGD(model.parameters(), lr=1e-5, momentum=0.9, weight_decay=0.0005 * B, fused=True)


def step():
    o = model(x)
    loss = o.float().mean()
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    step()
torch.cuda.synchronize()
print("%s: %.3f ms/step" % (prec, (time.perf_counter() - t0) / K * 1e3))
