"""ViT-B/16 as a plain torch module (parameter container + fp32 reference for the HIP engine).

Reference: RobustART/model/__init__.py:1 -> absent submodule; config type `vit_base` / `vit_b16_224` loads timm's
jx_vit_base_p16_224 (exprs/nips_benchmark/new_adv_train/vit_base/config.yaml:79), i.e. the public
ViT-B/16: patch 16, width 768, depth 12, 12 heads, qkv bias, MLP ratio 4, LayerNorm eps 1e-6, exact GELU, class
token + learned position embedding, head on the class token.  timm is not imported; the architecture is restated.

The module tree carries timm's parameter NAMES (`patch_embed.proj.*`, `blocks.N.mlp.fc{1,2}.*`, 152 keys for ViT-B/16), so the
checkpoint the reference's configs point `saver.pretrain.path` at loads with strict=True and a checkpoint saved here loads in timm.
`legacy_vit_keys` maps the names rounds 1-4 of this repository saved (`patch_embed.weight`, `blocks.N.fc1.*`)."""
import re
import torch
import torch.nn as nn
import torch.nn.functional as F


class Attention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.heads = heads
        self.scale = (dim // heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = (q @ k.transpose(-2, -1)) * self.scale
        attn = attn.softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj(x)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class PatchEmbed(nn.Module):
    def __init__(self, in_chans, embed_dim, patch_size):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, patch_size, stride=patch_size)

    weight = property(lambda self: self.proj.weight)      # (the engines fold from `patch_embed.weight / .bias`)
    bias = property(lambda self: self.proj.bias)

    def forward(self, x):
        return self.proj(x)


class Block(nn.Module):
    def __init__(self, dim, heads, mlp_ratio=4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    fc1 = property(lambda self: self.mlp.fc1)
    fc2 = property(lambda self: self.mlp.fc2)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, num_classes=1000, embed_dim=768, depth=12, num_heads=12, **_):
        super().__init__()
        self.patch_size, self.embed_dim, self.num_heads = patch_size, embed_dim, num_heads
        self.patch_embed = PatchEmbed(3, embed_dim, patch_size)
        n = (img_size // patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, embed_dim))
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Linear(embed_dim, num_classes)
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.cls_token, std=.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        B = x.shape[0]
        x = self.patch_embed(x).flatten(2).transpose(1, 2)
        x = torch.cat([self.cls_token.expand(B, -1, -1), x], 1) + self.pos_embed
        for blk in self.blocks:
            x = blk(x)
        return self.head(self.norm(x)[:, 0])


def vit_base(num_classes=1000, **kw):
    kw.pop('drop_path_rate', None)
    kw.pop('drop_path', None)
    return VisionTransformer(num_classes=num_classes, **kw)


_LEGACY = [(re.compile(r'^patch_embed\.(weight|bias)$'), r'patch_embed.proj.\1'),
           (re.compile(r'^blocks\.(\d+)\.(fc[12])\.(weight|bias)$'), r'blocks.\1.mlp.\2.\3')]


def legacy_vit_keys(state_dict):
    """state dict with the parameter names rounds 1-4 of this repository wrote -> timm's names (a dict already in timm's layout
    passes through unchanged)."""
    out = {}
    for k, v in state_dict.items():
        for pat, rep in _LEGACY:
            if pat.match(k):
                k = pat.sub(rep, k)
                break
        out[k] = v
    return out


def timm_to_legacy_keys(state_dict):
    """the inverse (save-side flag `saver.legacy_vit_keys`): timm's names -> the names rounds 1-4 wrote"""
    out = {}
    for k, v in state_dict.items():
        k = re.sub(r'^patch_embed\.proj\.', 'patch_embed.', k)
        k = re.sub(r'^blocks\.(\d+)\.mlp\.(fc[12])\.', r'blocks.\1.\2.', k)
        out[k] = v
    return out
