#!/usr/bin/env python3
"""Launch the self-attention forward/backward a few times on the 4096-token layer shape (for rocprofv3 --pmc)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from guided_attention_amd import ops  # noqa: E402

B, H, N, D = int(sys.argv[1]) if len(sys.argv) > 1 else 1, 8, 4096, 40
q, k, v, do = (torch.randn(B, N, H * D, device="cuda", dtype=torch.half) for _ in range(4))
for _ in range(5):
    o, lse = ops.self_attn_fwd(q, k, v, H, D ** -0.5)
    ops.self_attn_bwd(q, k, v, o, do, lse, H, D ** -0.5)
torch.cuda.synchronize()
