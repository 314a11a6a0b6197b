"""Tiny fixed CNN used by the attack parity tests and the golden generator."""
import torch
import torch.nn as nn


class TinyNet(nn.Module):
    """3x32x32 -> 10 logits; takes NORMALISED input (like the reference's `model`)."""

    def __init__(self):
        super().__init__()
        self.c1 = nn.Conv2d(3, 8, 3, padding=1)
        self.c2 = nn.Conv2d(8, 16, 3, padding=1, stride=2)
        self.fc = nn.Linear(16, 10)

    def forward(self, x):
        x = torch.relu(self.c1(x))
        x = torch.relu(self.c2(x))
        x = x.mean((2, 3))
        return self.fc(x) * 4.0


def make_tinynet(state=None):
    torch.manual_seed(1234)
    net = TinyNet().eval()
    if state is not None:
        net.load_state_dict({k: torch.as_tensor(v) for k, v in state.items()})
    for p in net.parameters():
        p.requires_grad_(False)
    return net


def make_batch(n=4, hw=32, seed=7):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, 3, hw, hw, generator=g)
    return x
