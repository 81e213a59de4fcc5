"""Seeded synthetic weights / inputs for benchmarks and smoke runs (no datasets or checkpoints
are reachable: SURVEY.md section 8(d)).  Realistic scale: conv ~ N(0, 2/(k*k*cin)),
gamma ~ U(.5,1.5), beta ~ N(0,.1), running_mean ~ N(0,.1), running_var ~ U(.5,1.5)."""
import torch


def init_synthetic(model, seed=0):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 4:
                std = (2.0 / (p.shape[1] * p.shape[2] * p.shape[3])) ** 0.5
                p.copy_(torch.randn(p.shape, generator=g) * std)
            elif name.endswith("weight"):
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        for name, b in model.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
            elif name.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) + 0.5)
    if hasattr(model, "invalidate_packed"):
        model.invalidate_packed()
    return model


def synthetic_batch(batch, height=416, width=416, seed=0, device="cpu"):
    """uniform [0,1) image batch, like ToTensor() output (train.py:187)."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(batch, 3, height, width, generator=g).to(device)
