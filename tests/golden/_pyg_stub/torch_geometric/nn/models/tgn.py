import torch
class TimeEncoder(torch.nn.Module):
    def __init__(self, *a, **k): super().__init__()
