"""Helpers for the -m gpu tests: device, dump directory, fp32 references."""
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out')


def dev():
    return torch.device('cuda:0')


def dump(name: str, text: str) -> None:
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, name), 'w') as f:
        f.write(text)


def randn_bf16(*shape, scale=1.0, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(dev())


def assert_close(out, ref, rtol, atol, what=''):
    out = out.float(); ref = ref.float()
    err = (out - ref).abs()
    lim = atol + rtol * ref.abs()
    bad = err > lim
    if bad.any():
        i = int(torch.argmax((err - lim).flatten()))
        raise AssertionError(f'{what}: {int(bad.sum())}/{bad.numel()} out of tolerance; worst idx {i}: '
                             f'got {out.flatten()[i].item()} want {ref.flatten()[i].item()} '
                             f'(max abs err {err.max().item():.4g}, ref rms {ref.pow(2).mean().sqrt().item():.4g})')
