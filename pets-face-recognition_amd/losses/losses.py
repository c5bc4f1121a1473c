"""FocalLoss with the reference's signature (/root/reference/losses/losses.py:7-28):
`FocalLoss(num_class, gamma=0, eps=1e-7, alpha=None)`; loss = mean((1 − p_t)^γ · CE).  With the defaults every FE
config uses (γ = 0, α = None) it is the mean cross-entropy."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class FocalLoss(nn.Module):
    def __init__(self, num_class: int, gamma=0, eps=1e-7, alpha=None):
        super().__init__()
        self.gamma, self.eps = gamma, eps
        self.adaptive_flag = bool(alpha)
        if self.adaptive_flag:
            self.alpha = nn.Parameter(torch.ones(num_class))

    def reset_parameters(self):
        if self.adaptive_flag:
            nn.init.ones_(self.alpha)

    def forward(self, input, target):
        if self.adaptive_flag:
            input = self.alpha * input
        if input.is_cuda:
            from ._head_hip import FocalCEFunction
            return FocalCEFunction.apply(input, target, float(self.gamma))
        ce = F.cross_entropy(input, target, reduction="none")
        pt = torch.exp(-ce)
        return ((1 - pt) ** self.gamma * ce).mean()
