"""Train-mode ViT-B/16 on the HIP kernels: forward, backward to every parameter.

The training step the reference's solver runs for `vit_base` (exprs/nips_benchmark/new_adv_train/vit_base/config.yaml:
AdamW, label_smooth 0.1, drop_path_rate 0.0 -- no stochastic layers) with every FLOP on HIP kernels:

  forward / backward-to-input       ViTEngine (igemm GEMMs, fused attention forward / backward, row kernels)
  Linear weight gradients           split-K GEMM on rart_conv_igemm_bf16 over transposed operands
                                    (rart_transpose_gather_bf16, rart_wgrad_reduce_f32) -- the ResNet train engine's path
  Linear biases, position embedding rart_colsum_bf16
  LayerNorm gamma / beta            rart_layernorm_bwd_full_bf16 (fused with the backward to the input)

Gradients are written into the parameters' `.grad` tensors (views of the flat gradient arena, train/arena.py);
`on_grad_ready(param)` lets the arena launch a bucket's all-reduce as soon as its last gradient exists.
"""
import ctypes

from .. import _lib
from .vit_engine import F_GELU_BWD, F_OUT_F32, ViTEngine


class ViTTrainEngine(ViTEngine):
    def __init__(self, model, device='cuda', on_grad_ready=None):
        super().__init__(model, device)
        self.model = model
        self.on_grad_ready = on_grad_ready or (lambda p: None)

    def repack(self):
        """fp32 master weights -> bf16 tables; call after every optimizer step."""
        self.refold(self.model)

    def forward(self, src, src_is_u8, mean, std):
        if not src_is_u8:
            src = src.detach().float().contiguous()
        return self._forward(src, src_is_u8, mean, std, keep=True)

    # ------------------------------------------------------------------ helpers
    def _scratch(self, name, nbytes):
        torch = _lib.require_gpu()
        t = self._buf.get(name)
        if t is None or t.numel() < nbytes:
            t = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            self._buf[name] = t
        return t

    def _wgrad(self, dz, n_out, n_pad, x, c_in, grad, rows, dz_images=None):
        """grad[n_out][c_in] = dz^T . x  (dz: bf16 [rows][n_pad] dense, columns >= n_out zero; x: bf16 [rows][c_in] dense).
        dz_images = (B, rows_per_image_in_memory, rows_used): dz rows are the first `rows_used` of every image block."""
        lib, sp = self.lib, _lib.stream_ptr()
        bn_tile = 128 if n_pad > 64 else 64
        tiles = ((c_in + 127) // 128) * ((n_pad + bn_tile - 1) // bn_tile)
        splits = max(1, min(1024 // max(tiles, 1), rows // 512 if rows >= 1024 else 1, 256))
        chunk = ((rows + splits - 1) // splits + 63) // 64 * 64
        m_pad = chunk * splits
        n_rows = (n_pad + bn_tile - 1) // bn_tile * bn_tile
        dzt = self._scratch('wg_dzT', n_rows * m_pad * 2)
        colt = self._scratch('wg_colT', c_in * m_pad * 2)
        zero = (ctypes.c_int * 1)(0)
        if n_rows > n_pad:
            dzt[:n_rows * m_pad * 2].zero_()
        if dz_images is None:
            b, sh, gh = 1, rows, rows
        else:
            b, sh, gh = dz_images
        _lib.check(lib.rart_transpose_gather_bf16(_lib.ptr(dz), _lib.ptr(dzt), b, sh, 1, n_pad, gh, 1, 1, 1, 1, zero, zero, m_pad,
                                                  chunk, n_rows, sp))
        _lib.check(lib.rart_transpose_gather_bf16(_lib.ptr(x), _lib.ptr(colt), 1, rows, 1, c_in, rows, 1, 1, 1, 1, zero, zero,
                                                  m_pad, chunk, c_in, sp))
        ld_n = (n_pad + 7) // 8 * 8
        part = self._scratch('wg_part', splits * c_in * ld_n * 4)
        self._gemm(colt, dzt, part, c_in, chunk, ld_n, chunk, ld_n, flags=F_OUT_F32,
                   batched=dict(n=splits, inner=splits, src=(0, c_in * chunk), wgt=(0, n_rows * chunk), dst=(0, c_in * ld_n),
                                wgt_row_stride=chunk))
        _lib.check(lib.rart_wgrad_reduce_f32(_lib.ptr(part), splits, 1, c_in, c_in, n_out, ld_n, _lib.ptr(grad), 0, sp))

    def _colsum(self, x, ld, rows, cols, out):
        lib = self.lib
        need = lib.rart_colsum_workspace_bytes(rows, cols)
        ws = self._scratch('cs_ws', need)
        _lib.check(lib.rart_colsum_bf16(_lib.ptr(x), ld, rows, cols, _lib.ptr(out), 0, _lib.ptr(ws), need, _lib.stream_ptr()))

    def _ln_bwd(self, dy, x, gamma, res, dx, rows, strides, norm):
        lib, D = self.lib, self.D
        need = lib.rart_layernorm_bwd_workspace_bytes(D)
        ws = self._scratch('ln_ws', need)
        _lib.check(lib.rart_layernorm_bwd_full_bf16(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(gamma), _lib.ptr(res), _lib.ptr(dx), rows,
                                                    D, strides[0], strides[1], strides[2], strides[3], 1e-6,
                                                    _lib.ptr(norm.weight.grad), _lib.ptr(norm.bias.grad), 0, _lib.ptr(ws), need,
                                                    _lib.stream_ptr()))
        self.on_grad_ready(norm.weight)
        self.on_grad_ready(norm.bias)

    def _linear_grads(self, lin, dz, n_pad, x, rows, dz_images=None):
        n_out, c_in = lin.weight.shape[0], lin.weight[0].numel()
        self._wgrad(dz, n_out, n_pad, x, c_in, lin.weight.grad, rows, dz_images)
        self.on_grad_ready(lin.weight)

    # ------------------------------------------------------------------ backward to every parameter
    def backward(self, dlogits):
        """dlogits: fp32 [B][classes] = d(loss)/dlogits of the last forward().  Fills .grad of every parameter."""
        torch = _lib.require_gpu()
        lib, sp, m = self.lib, _lib.stream_ptr(), self.model
        saved, x_last, (B, Himg, Wimg, P, T) = self._saved
        D, rows = self.D, B * T
        dl = dlogits.detach().float().contiguous()
        m.head.bias.grad.copy_(dl.sum(0))
        self.on_grad_ready(m.head.bias)
        dlb = self._get('dl_bf16', (B, self.head_kpad))
        _lib.check(lib.rart_f32_to_bf16_rows(_lib.ptr(dl), _lib.ptr(dlb), B, self.n_classes, self.head_kpad, sp))
        self._linear_grads(m.head, dlb, self.head_kpad, self._buf['cls'], B)
        dcls = self._get('dcls', (B, D))
        self._gemm(dlb, self.head_wd, dcls, B, self.head_kpad, D, self.head_kpad, D)
        dx = self._get('g_x_a', (B, T, D))
        dx.zero_()
        self._ln_bwd(dcls, x_last, self.ng, None, dx, B, (D, T * D, 0, T * D), m.norm)
        dqkv = self._get('g_qkv', (rows, 3 * D))
        ln = self._get('ln', (B, T, D))
        for li in range(len(self.layers) - 1, -1, -1):
            L, blk = self.layers[li], m.blocks[li]
            x_in, xm, qkv, u, att = saved[li]
            hidden = L['hidden']
            # ---- MLP: x_out = xm + fc2(gelu(fc1(LN2(xm))))
            hid = self._get('hid', (B, T, hidden))
            _lib.check(lib.rart_gelu_bf16(_lib.ptr(u), _lib.ptr(hid), u.numel(), sp))
            self._linear_grads(blk.fc2, dx, D, hid, rows)
            self._colsum(dx, D, rows, D, blk.fc2.bias.grad)
            self.on_grad_ready(blk.fc2.bias)
            dh = self._get('g_hid', (rows, hidden))
            self._gemm(dx, L['fc2_wd'], dh, rows, D, hidden, D, hidden, mask=u, flags=F_GELU_BWD)     # du = (dx W2) * gelu'(u)
            _lib.check(lib.rart_layernorm_bf16(_lib.ptr(xm), _lib.ptr(L['n2g']), _lib.ptr(L['n2b']), _lib.ptr(ln), rows, D, D, D,
                                               1e-6, sp))
            self._linear_grads(blk.fc1, dh, hidden, ln, rows)
            self._colsum(dh, hidden, rows, hidden, blk.fc1.bias.grad)
            self.on_grad_ready(blk.fc1.bias)
            dln = self._get('g_ln', (rows, D))
            self._gemm(dh, L['fc1_wd'], dln, rows, hidden, D, hidden, D)
            dxm = self._get('g_xm', (B, T, D))
            self._ln_bwd(dln, xm, L['n2g'], dx, dxm, rows, (D, D, D, D), blk.norm2)
            # ---- attention: xm = x_in + proj(attn(LN1(x_in)))
            self._linear_grads(blk.attn.proj, dxm, D, att, rows)
            self._colsum(dxm, D, rows, D, blk.attn.proj.bias.grad)
            self.on_grad_ready(blk.attn.proj.bias)
            datt = self._get('g_att', (rows, D))
            self._gemm(dxm, L['proj_wd'], datt, rows, D, D, D, D)
            _lib.check(lib.rart_vit_attention_bwd(_lib.ptr(qkv), _lib.ptr(att), _lib.ptr(datt), _lib.ptr(dqkv), B, T, self.H,
                                                  self.hd, sp))
            _lib.check(lib.rart_layernorm_bf16(_lib.ptr(x_in), _lib.ptr(L['n1g']), _lib.ptr(L['n1b']), _lib.ptr(ln), rows, D, D, D,
                                               1e-6, sp))
            self._linear_grads(blk.attn.qkv, dqkv, 3 * D, ln, rows)
            self._colsum(dqkv, 3 * D, rows, 3 * D, blk.attn.qkv.bias.grad)
            self.on_grad_ready(blk.attn.qkv.bias)
            self._gemm(dqkv, L['qkv_wd'], dln, rows, 3 * D, D, 3 * D, D)
            self._ln_bwd(dln, x_in, L['n1g'], dxm, dx, rows, (D, D, D, D), blk.norm1)
        # ---- embeddings: x0[b][0] = cls + pos[0]; x0[b][1+p] = patch_embed(patch p) + pos[1+p]
        # derive every gradient that READS pos_embed.grad before the first on_grad_ready: with a small dist.bucket_mb the
        # {cls_token, pos_embed} bucket would otherwise start its asynchronous all-reduce (in place, on the RCCL stream)
        # while patch_embed.bias.grad is still being computed from it, and the bias would be summed across ranks twice
        pos_g = m.pos_embed.grad.view(T * D)
        self._colsum(dx, T * D, B, T * D, pos_g)
        m.cls_token.grad.view(D).copy_(pos_g[:D])
        m.patch_embed.bias.grad.copy_(m.pos_embed.grad.view(T, D)[1:].sum(0))
        self.on_grad_ready(m.pos_embed)
        self.on_grad_ready(m.cls_token)
        self.on_grad_ready(m.patch_embed.bias)
        kk = 3 * self.ps * self.ps
        patches_hi = self._buf['patches'][0]
        self._wgrad(dx.view(rows, D)[1:], D, D, patches_hi.view(B * P, kk), kk, m.patch_embed.weight.grad, B * P,
                    dz_images=(B, T, P))
        self.on_grad_ready(m.patch_embed.weight)
