"""AutoencoderKL decoder (SD-1.x / 2.x layout) in plain PyTorch: latents -> image.  Caller of the hot
path (`decode_latents`, pipeline_guided_attention.py:1060); reproduced so `.images[0]` exists, not
accelerated, and outside the images/sec metric.  Parameter names follow the
diffusers checkpoint layout (`post_quant_conv`, `decoder.*`)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Resnet(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(32, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class _Attn(nn.Module):  # single-head spatial self-attention of the VAE mid block
    def __init__(self, c):
        super().__init__()
        self.group_norm = nn.GroupNorm(32, c, eps=1e-6)
        self.query, self.key, self.value, self.proj_attn = (nn.Linear(c, c) for _ in range(4))

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x).view(b, c, h * w).transpose(1, 2)
        q, k, v = self.query(t)[:, None], self.key(t)[:, None], self.value(t)[:, None]
        o = F.scaled_dot_product_attention(q, k, v)[:, 0]
        return x + self.proj_attn(o).transpose(1, 2).reshape(b, c, h, w)


class _Up(nn.Module):
    def __init__(self, cin, cout, upsample):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if i == 0 else cout, cout) for i in range(3)])
        self.upsamplers = nn.ModuleList([nn.Module()]) if upsample else None
        if upsample:
            self.upsamplers[0].conv = nn.Conv2d(cout, cout, 3, padding=1)

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0].conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))
        return x


class _Mid(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(c, c), _Resnet(c, c)])
        self.attentions = nn.ModuleList([_Attn(c)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _Decoder(nn.Module):
    def __init__(self, ch=(128, 256, 512, 512), latent=4):
        super().__init__()
        rev = list(reversed(ch))
        self.conv_in = nn.Conv2d(latent, rev[0], 3, padding=1)
        self.mid_block = _Mid(rev[0])
        ups, prev = [], rev[0]
        for i, c in enumerate(rev):
            ups.append(_Up(prev, c, i != len(rev) - 1))
            prev = c
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(32, ch[0], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[0], 3, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for u in self.up_blocks:
            x = u(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKLDecoder(nn.Module):
    scaling_factor = 0.18215

    def __init__(self, ch=(128, 256, 512, 512)):
        super().__init__()
        self.post_quant_conv = nn.Conv2d(4, 4, 1)
        self.decoder = _Decoder(ch)

    @classmethod
    def tiny(cls):
        return cls(ch=(32, 32, 64, 64))

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))

    def init_weights_(self, seed=1):
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for n, p in self.named_parameters():
                if n.endswith("bias"):
                    p.zero_()
                elif p.dim() == 1:
                    p.fill_(1.0)
                else:
                    p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * (3.0 / p[0].numel()) ** 0.5)
        return self
