"""ViT-B/16 forward and backward-to-input on the hand-written HIP kernels (evaluation of clean / corrupted images,
BASELINE config 3; the gradient step of every attack, adv/attack.py:21-22, autopgd_base.py:271-289).

Every matmul -- patch embedding, qkv, Q.K^T and P.V per (image, head) as batched problems, proj, MLP, head -- runs on
rart_conv_igemm_bf16; LayerNorm, soft-max rows, the V transpose, patch extraction and the class-token / position add
are the small kernels of csrc/vit_aux.hip.  bf16 activations, fp32 accumulation and statistics; the image enters as
a hi+lo bf16 pair like the ResNet stem.  Backward (forward_backward): every dgrad GEMM and the five per-head products
of the attention backward (S = QK^T recomputed, dP = dO V^T, dQ = dS K, dK = dS^T Q, dV = P^T dO) are igemm launches
(batched over (image, head)); GELU', LayerNorm and soft-max backward, the transposes and the un-patchify are
csrc/vit_bwd.hip / train_convbn.hip kernels.  Reference module: robustart_amd/model/vit_torch.py."""
import ctypes

from .. import _lib

F_RELU, F_OUT_F32, F_GELU, F_GELU_BWD = 1, 2, 4, 8
F_GELU_KEEP = 64          # dst = gelu(u), `mask` receives the pre-activation u (256 x 256 GEMM only)
PRECISIONS = {'bf16': 'bf16', 'bf16x3': 'bf16x3', 'fp32x': 'bf16x3'}


def _pair(t):
    """fp32 tensor -> [2][...] bf16 planes (hi = bf16(v), lo = bf16(v - hi)): the split-bf16 representation"""
    import torch
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo]).contiguous()


class ViTEngine:
    def __init__(self, model, device='cuda', precision='bf16'):
        """precision: 'bf16' -- bf16 storage, fp32 accumulation (the fast path; logits within ~3e-3 of the fp32 network);
        'bf16x3' (alias 'fp32x') -- the REFERENCE-PRECISION mode (`_forward_x3` / `_backward_x3`): the reference evaluates and
        attacks ViT in fp32 (exprs/exp/imagenet_c_loop_mini/config_vit_base.yaml:1-9 has no precision key; adv/attack.py:20-23;
        autopgd_base.py:271-289) and the north star asks for logits within 1e-4 of it, so every activation, gradient and weight is
        a hi + lo pair of bf16 planes, every contraction the three MFMA products of rart_gemm_pair_bf16, and LayerNorm / soft-max /
        GELU are evaluated in fp32 on hi + lo (csrc/vit_pair.hip)."""
        torch = _lib.require_gpu()
        self.lib = _lib.load()
        self.device = torch.device(device)
        if precision not in PRECISIONS:
            raise ValueError('precision must be one of %s' % sorted(PRECISIONS))
        self.precision = PRECISIONS[precision]
        self.profile = None              # a list collects (flops issued, event0, event1, kind) per GEMM launch (bench.py)
        m = model
        self.D, self.H, self.ps = m.embed_dim, m.num_heads, m.patch_size
        self.hd = self.D // self.H
        import os as _os
        self.pair_w_interleaved = _os.environ.get('RART_PAIR_WIL', '0') == '1'     # see engine.py: weight tables interleaved per K step
        self._w_il = {}
        self.refold(model)
        self._buf = {}
        self.fused_attention = True
        self.fused_attention_bwd = True      # False: the decomposition into batched igemm products (cross-check)

    def refold(self, model):
        """(Re)build every weight table from `model`'s current parameters.  Parameters already on the GPU are packed
        there (a dozen small torch ops per layer), so the adversarial-training loop can refresh the attack engine every
        iteration, like ResNet50Engine.refold."""
        torch = _lib.require_gpu()
        m, dev = model, self.device

        def bf(w2d):                                         # fp32 [rows][k] (any device) -> bf16 on the engine's device
            return w2d.detach().to(dev, torch.float32).to(torch.bfloat16)

        def pad_rows(w, mult):
            r = (w.shape[0] + mult - 1) // mult * mult
            if r == w.shape[0]:
                return w.contiguous()
            return torch.cat([w, torch.zeros(r - w.shape[0], w.shape[1], dtype=w.dtype, device=w.device)], 0).contiguous()

        def wt(linear_w, n_cols):
            return pad_rows(bf(linear_w), 128 if n_cols > 64 else 64)

        def wd(linear_w, k_pad=None):
            """backward-to-input table: dx[rows][in] = dy[rows][out] . W  ->  rows = in features, K = out features"""
            w = bf(linear_w).t()
            if k_pad is not None and k_pad > w.shape[1]:
                w = torch.cat([w, torch.zeros(w.shape[0], k_pad - w.shape[1], dtype=w.dtype, device=dev)], 1)
            return pad_rows(w, 128 if w.shape[0] > 64 else 64)

        def f32(t):
            return t.detach().to(dev, torch.float32).contiguous()
        pe = m.patch_embed.weight.detach().reshape(self.D, -1)                         # [D][c*ps*ps + r*ps + s]
        peb = bf(pe)
        self.pe_w = pad_rows(torch.cat([peb, peb], 1), 128)                            # hi | lo taps
        self.pe_b = f32(m.patch_embed.bias)
        pos = f32(m.pos_embed)[0]
        self.pos = pos.contiguous()
        self.cls_pos0 = (f32(m.cls_token)[0, 0] + pos[0]).contiguous()
        self.tokens = pos.shape[0]
        self.layers = []
        for blk in m.blocks:
            self.layers.append(dict(
                n1g=f32(blk.norm1.weight), n1b=f32(blk.norm1.bias), n2g=f32(blk.norm2.weight), n2b=f32(blk.norm2.bias),
                qkv_w=wt(blk.attn.qkv.weight, 3 * self.D), qkv_b=f32(blk.attn.qkv.bias),
                proj_w=wt(blk.attn.proj.weight, self.D), proj_b=f32(blk.attn.proj.bias),
                fc1_w=wt(blk.fc1.weight, blk.fc1.out_features), fc1_b=f32(blk.fc1.bias),
                fc2_w=wt(blk.fc2.weight, self.D), fc2_b=f32(blk.fc2.bias), hidden=blk.fc1.out_features,
                qkv_wd=wd(blk.attn.qkv.weight), proj_wd=wd(blk.attn.proj.weight),
                fc1_wd=wd(blk.fc1.weight), fc2_wd=wd(blk.fc2.weight)))
        self.ng, self.nb = f32(m.norm.weight), f32(m.norm.bias)
        self.n_classes = m.head.out_features
        self.head_w = wt(m.head.weight, self.n_classes)
        self.head_b = f32(m.head.bias)
        self.head_kpad = (self.n_classes + 31) // 32 * 32
        self.head_wd = wd(m.head.weight, self.head_kpad)
        self.pe_wd = wd(pe)
        if self.precision == 'bf16x3':
            self._refold_pair(m)

    def _refold_pair(self, m):
        """Pair ([2][rows][K] bf16: hi, lo) forward tables W and backward-to-input tables W^T of every Linear; rows padded to the
        256-row tile of rart_gemm_pair_bf16 (rows past the matrix are zero and never stored)."""
        torch = _lib.require_gpu()
        dev = self.device

        def tab(w2d, k_pad=None):
            w = w2d.detach().to(dev, torch.float32)
            if k_pad is not None and k_pad > w.shape[1]:
                w = torch.cat([w, torch.zeros(w.shape[0], k_pad - w.shape[1], device=dev)], 1)
            r = (w.shape[0] + 255) // 256 * 256
            if r != w.shape[0]:
                w = torch.cat([w, torch.zeros(r - w.shape[0], w.shape[1], device=dev)], 0)
            t = _pair(w.contiguous())
            if self.pair_w_interleaved and t.shape[2] % 32 == 0:
                # round 5: per row and 32-deep K step the hi slice then the lo slice (rart_gemm_pair_bf16 flag 16: one 128-byte line per
                # row and step); kept beside the planes, keyed by their address
                rows, k = t.shape[1], t.shape[2]
                self._w_il[t.data_ptr()] = torch.stack([t[0].reshape(rows, k // 32, 32), t[1].reshape(rows, k // 32, 32)], 2).reshape(rows, 2 * k).contiguous()
            return t
        self._w_il = {}
        pe = m.patch_embed.weight.detach().reshape(self.D, -1)
        self.x3 = dict(pe_w=tab(pe), pe_wd=tab(pe.t()), head_w=tab(m.head.weight), head_wd=tab(m.head.weight.t(), self.head_kpad),
                       layers=[dict(qkv_w=tab(b.attn.qkv.weight), proj_w=tab(b.attn.proj.weight), fc1_w=tab(b.fc1.weight),
                                    fc2_w=tab(b.fc2.weight), qkv_wd=tab(b.attn.qkv.weight.t()), proj_wd=tab(b.attn.proj.weight.t()),
                                    fc1_wd=tab(b.fc1.weight.t()), fc2_wd=tab(b.fc2.weight.t())) for b in m.blocks])

    def _get(self, name, shape, dtype=None, zero=False):
        torch = _lib.require_gpu()
        dtype = dtype or torch.bfloat16
        t = self._buf.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            self._buf[name] = t
        return t

    def _gemm(self, src, wgt, dst, rows, k, n_cols, src_ld, dst_ld, bias=None, res=None, flags=0, n_taps=1,
              tap_src_off=None, rows_per_image=None, dst_rows_per_image=None, dst_row_off=0, batched=None,
              src_rows_per_image=None, mask=None):
        """rows x k (x n_taps) times wgt^T -> dst.  rows_per_image/dst_rows_per_image/dst_row_off place the output
        rows of image b at b*dst_rows_per_image + dst_row_off (class-token slot).  batched = dict(n, inner,
        src=(outer, inner), wgt=(outer, inner), dst=(outer, inner), wgt_row_stride)."""
        d = _lib.ConvDesc()
        d.src, d.wgt, d.dst = src.data_ptr(), wgt.data_ptr(), dst.data_ptr()
        d.bias = bias.data_ptr() if bias is not None else None
        d.res = res.data_ptr() if res is not None else None
        d.mask = mask.data_ptr() if mask is not None else None
        rpi = rows_per_image or rows
        d.batch, d.grid_h, d.grid_w = rows // rpi, rpi, 1
        d.src_h, d.src_w, d.src_pix_stride = (src_rows_per_image or rpi), 1, src_ld
        d.k_per_tap, d.n_taps = k, n_taps
        d.sy, d.sx = 1, 1
        for i in range(n_taps):
            d.tap_dy[i], d.tap_dx[i] = 0, 0
            d.tap_src_off[i] = tap_src_off[i] if tap_src_off else 0
        d.n_cols = n_cols
        d.dst_h, d.dst_w = (dst_rows_per_image or rpi), 1
        d.dst_sy, d.dst_sx, d.dst_oy, d.dst_ox = 1, 1, dst_row_off, 0
        d.dst_pix_stride = dst_ld
        d.flags = flags
        if batched:
            d.n_batched, d.z_inner = batched['n'], batched['inner']
            d.src_z_outer, d.src_z_inner = batched['src']
            d.wgt_z_outer, d.wgt_z_inner = batched['wgt']
            d.dst_z_outer, d.dst_z_inner = batched['dst']
            d.wgt_row_stride = batched.get('wgt_row_stride', 0)
        _lib.check(self.lib.rart_conv_igemm_bf16(ctypes.byref(d), _lib.stream_ptr()))

    def _forward(self, src, src_is_u8, mean, std, keep=False):
        if self.precision == 'bf16x3':
            return self._forward_x3(src, src_is_u8, mean, std, keep)
        torch = _lib.require_gpu()
        lib, sp = self.lib, _lib.stream_ptr()
        if src_is_u8:
            B, Himg, Wimg = src.shape[0], src.shape[1], src.shape[2]
        else:
            B, Himg, Wimg = src.shape[0], src.shape[2], src.shape[3]
        D, H, hd, ps = self.D, self.H, self.hd, self.ps
        P = (Himg // ps) * (Wimg // ps)
        T = P + 1
        assert T == self.tokens, 'image size does not match the position embedding'
        kk = 3 * ps * ps
        patches = self._get('patches', (2, B, P, kk))
        meanf, stdf = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
        _lib.check(lib.rart_vit_patchify(_lib.ptr(src), 1 if src_is_u8 else 0, _lib.ptr(patches[0]), _lib.ptr(patches[1]),
                                         B, Himg, Wimg, ps, meanf, stdf, sp))
        x = self._get('x0' if keep else 'x', (B, T, D))
        lo_off = (patches[1].data_ptr() - patches[0].data_ptr()) // 2
        self._gemm(patches[0], self.pe_w, x, B * P, kk, D, kk, D, bias=self.pe_b, n_taps=2, tap_src_off=[0, lo_off],
                   rows_per_image=P, dst_rows_per_image=T, dst_row_off=1)
        _lib.check(lib.rart_vit_add_pos_cls(_lib.ptr(x), _lib.ptr(self.cls_pos0), _lib.ptr(self.pos), B, T, D, sp))
        rows = B * T
        ln = self._get('ln', (B, T, D))
        t_pad = (T + 31) // 32 * 32                                   # 224: K extent of P.V
        s_ld = (T + 7) // 8 * 8                                       # 200
        if not self.fused_attention:
            scores = self._get('scores', (B * H, T, s_ld))
            probs = self._get('probs', (B * H, T, t_pad))
            vt = self._get('vt', (B * H * hd + 128, t_pad), zero=True)
        saved = []
        for li, L in enumerate(self.layers):
            # keep mode stores what the backward needs: block input, post-attention stream, qkv, fc1 pre-activation
            qkv = self._get('qkv%d' % li if keep else 'qkv', (B * T + 256, 3 * D), zero=True)   # slack rows: K tiles read in place
            xm = self._get('xm%d' % li, (B, T, D)) if keep else x
            att = self._get('att%d' % li if keep else 'att', (B, T, D))     # kept: delta = rowsum(dO * O) in the backward
            xo = self._get('x%d' % (li + 1), (B, T, D)) if keep else x
            _lib.check(lib.rart_layernorm_bf16(_lib.ptr(x), _lib.ptr(L['n1g']), _lib.ptr(L['n1b']), _lib.ptr(ln), rows, D,
                                               D, D, 1e-6, sp))
            self._gemm(ln, L['qkv_w'], qkv, rows, D, 3 * D, D, 3 * D, bias=L['qkv_b'])
            if self.fused_attention:
                _lib.check(lib.rart_vit_attention(_lib.ptr(qkv), _lib.ptr(att), B, T, H, hd, sp))
            else:
                self._attention_unfused(qkv, scores, probs, vt, att, B, T, s_ld, t_pad)
            self._gemm(att, L['proj_w'], xm, rows, D, D, D, D, bias=L['proj_b'], res=x)
            _lib.check(lib.rart_layernorm_bf16(_lib.ptr(xm), _lib.ptr(L['n2g']), _lib.ptr(L['n2b']), _lib.ptr(ln), rows, D,
                                               D, D, 1e-6, sp))
            hid = self._get('hid', (B, T, L['hidden']))
            if keep:
                u = self._get('u%d' % li, (B, T, L['hidden']))
                if lib.rart_gemm256_supported(rows, D, L['hidden'], D, L['hidden']):
                    # one launch writes both the pre-activation (for GELU' in the backward) and gelu of it
                    self._gemm(ln, L['fc1_w'], hid, rows, D, L['hidden'], D, L['hidden'], bias=L['fc1_b'], mask=u, flags=F_GELU_KEEP)
                else:
                    self._gemm(ln, L['fc1_w'], u, rows, D, L['hidden'], D, L['hidden'], bias=L['fc1_b'])
                    _lib.check(lib.rart_gelu_bf16(_lib.ptr(u), _lib.ptr(hid), u.numel(), sp))
                saved.append((x, xm, qkv, u, att))
            else:
                self._gemm(ln, L['fc1_w'], hid, rows, D, L['hidden'], D, L['hidden'], bias=L['fc1_b'], flags=F_GELU)
            self._gemm(hid, L['fc2_w'], xo, rows, L['hidden'], D, L['hidden'], D, bias=L['fc2_b'], res=xm)
            x = xo
        if keep:
            self._saved = (saved, x, (B, Himg, Wimg, P, T))
        cls = self._get('cls', (B, D))
        _lib.check(lib.rart_layernorm_bf16(_lib.ptr(x), _lib.ptr(self.ng), _lib.ptr(self.nb), _lib.ptr(cls), B, D, T * D, D,
                                           1e-6, sp))
        logits = torch.empty(B, self.n_classes, dtype=torch.float32, device=self.device)
        self._gemm(cls, self.head_w, logits, B, D, self.n_classes, D, self.n_classes, bias=self.head_b, flags=F_OUT_F32)
        return logits

    def _attention_unfused(self, qkv, scores, probs, vt, att, B, T, s_ld, t_pad):
        """Reference decomposition (batched igemm Q.K^T -> soft-max rows -> V transpose -> batched igemm P.V); kept to
        cross-check the fused kernel and to exercise the batched-GEMM path of rart_conv_igemm_bf16."""
        lib, sp = self.lib, _lib.stream_ptr()
        D, H, hd = self.D, self.H, self.hd
        self._gemm(qkv, qkv[:, D:], scores, T, hd, s_ld, 3 * D, s_ld, rows_per_image=T,
                   batched=dict(n=B * H, inner=H, src=(T * 3 * D, hd), wgt=(T * 3 * D, hd), dst=(H * T * s_ld, T * s_ld),
                                wgt_row_stride=3 * D))
        _lib.check(lib.rart_softmax_rows_bf16(_lib.ptr(scores), _lib.ptr(probs), B * H * T, T, s_ld, t_pad,
                                              float(hd) ** -0.5, sp))
        _lib.check(lib.rart_vit_transpose_v(_lib.ptr(qkv), _lib.ptr(vt), B, T, H, hd, 3 * D, 2 * D, t_pad, sp))
        self._gemm(probs, vt, att, T, t_pad, hd, t_pad, D, rows_per_image=T,
                   batched=dict(n=B * H, inner=H, src=(H * T * t_pad, T * t_pad), wgt=(H * hd * t_pad, hd * t_pad),
                                dst=(T * D, hd)))

    # ------------------------------------------------------------------ backward to the input
    def _attention_bwd(self, qkv, datt, dqkv, B, T):
        """dqkv[rows][3D] from datt[rows][D] for one layer; P is recomputed (S = QK^T, soft-max)."""
        lib, sp = self.lib, _lib.stream_ptr()
        D, H, hd = self.D, self.H, self.hd
        t_pad, s_ld = (T + 31) // 32 * 32, (T + 7) // 8 * 8
        BH = B * H
        scale = float(hd) ** -0.5
        scores = self._get('scores', (BH, T, s_ld))
        probs = self._get('probs', (BH, T, t_pad))
        self._gemm(qkv, qkv[:, D:], scores, T, hd, s_ld, 3 * D, s_ld, rows_per_image=T,
                   batched=dict(n=BH, inner=H, src=(T * 3 * D, hd), wgt=(T * 3 * D, hd), dst=(H * T * s_ld, T * s_ld),
                                wgt_row_stride=3 * D))
        _lib.check(lib.rart_softmax_rows_bf16(_lib.ptr(scores), _lib.ptr(probs), BH * T, T, s_ld, t_pad, scale, sp))
        # dP = dO . V^T   (rows: queries, K: head_dim, columns: keys = rows of the V slice of qkv)
        dprobs = self._get('dprobs', (BH, T, s_ld))
        self._gemm(datt, qkv[:, 2 * D:], dprobs, T, hd, s_ld, D, s_ld, rows_per_image=T,
                   batched=dict(n=BH, inner=H, src=(T * D, hd), wgt=(T * 3 * D, hd), dst=(H * T * s_ld, T * s_ld),
                                wgt_row_stride=3 * D))
        ds = self._get('dscores', (BH, T, t_pad))
        _lib.check(lib.rart_softmax_bwd_rows_bf16(_lib.ptr(probs), _lib.ptr(dprobs), _lib.ptr(ds), BH * T, T, t_pad, s_ld,
                                                  t_pad, scale, sp))
        # transposed (token-contiguous) copies of K, Q, dO: [B][H][hd][t_pad]
        kt = self._get('kt', (BH * hd + 128, t_pad), zero=True)
        qt = self._get('qt', (BH * hd + 128, t_pad), zero=True)
        dot = self._get('dot', (BH * hd + 128, t_pad), zero=True)
        _lib.check(lib.rart_vit_transpose_v(_lib.ptr(qkv), _lib.ptr(kt), B, T, H, hd, 3 * D, D, t_pad, sp))
        _lib.check(lib.rart_vit_transpose_v(_lib.ptr(qkv), _lib.ptr(qt), B, T, H, hd, 3 * D, 0, t_pad, sp))
        _lib.check(lib.rart_vit_transpose_v(_lib.ptr(datt), _lib.ptr(dot), B, T, H, hd, D, 0, t_pad, sp))
        # dQ = dS . K
        self._gemm(ds, kt, dqkv, T, t_pad, hd, t_pad, 3 * D, rows_per_image=T,
                   batched=dict(n=BH, inner=H, src=(H * T * t_pad, T * t_pad), wgt=(H * hd * t_pad, hd * t_pad),
                                dst=(T * 3 * D, hd)))
        # query-contiguous copies of dS and P: [t_pad (key)][BH * t_pad (image-head, query)]
        m_all = BH * t_pad
        dst_t = self._get('ds_t', (t_pad, m_all))
        p_t = self._get('p_t', (t_pad, m_all))
        zero = (ctypes.c_int * 1)(0)
        for src_m, dst_m in ((ds, dst_t), (probs, p_t)):
            _lib.check(lib.rart_transpose_gather_bf16(_lib.ptr(src_m), _lib.ptr(dst_m), BH, T, 1, t_pad, t_pad, 1, 1, 1, 1,
                                                      zero, zero, m_all, 0, 0, sp))
        # dK = dS^T . Q,  dV = P^T . dO   (rows: keys, K: queries)
        for a_t, w_t, col in ((dst_t, qt, D), (p_t, dot, 2 * D)):
            self._gemm(a_t, w_t, dqkv[:, col:], T, t_pad, hd, m_all, 3 * D, rows_per_image=T,
                       batched=dict(n=BH, inner=H, src=(H * t_pad, t_pad), wgt=(H * hd * t_pad, hd * t_pad),
                                    dst=(T * 3 * D, hd)))

    def forward_backward(self, x01, mean, std, y, kind, y_target=None, scale=1.0):
        """-> (logits fp32, loss_indiv, d(sum_i scale*loss_i)/dx01 fp32 NCHW, pred int32); same contract as
        ResNet50Engine.forward_backward."""
        from ..noise.adv import logit_loss
        torch = _lib.require_gpu()
        lib, sp = self.lib, _lib.stream_ptr()
        x01 = x01.detach().float().contiguous()
        logits = self._forward(x01, False, mean, std, keep=True)
        saved, x_last, (B, Himg, Wimg, P, T) = self._saved
        loss, dl, pred = logit_loss(logits, y, kind, y_target, scale)
        self.last_dlogits = dl           # exposed for the parity tests (same upstream gradient for the reference)
        if self.precision == 'bf16x3':
            return logits, loss, self._backward_x3(dl, std), pred
        D, rows = self.D, B * T
        dlb = self._get('dl_bf16', (B, self.head_kpad))
        _lib.check(lib.rart_f32_to_bf16_rows(_lib.ptr(dl), _lib.ptr(dlb), B, self.n_classes, self.head_kpad, sp))
        dcls = self._get('dcls', (B, D))
        self._gemm(dlb, self.head_wd, dcls, B, self.head_kpad, D, self.head_kpad, D)
        dx = self._get('g_x_a', (B, T, D))
        dx.zero_()                                      # only the class token receives gradient from the head
        _lib.check(lib.rart_layernorm_bwd_bf16(_lib.ptr(dcls), _lib.ptr(x_last), _lib.ptr(self.ng), None, _lib.ptr(dx), B, D,
                                               D, T * D, 0, T * D, 1e-6, sp))
        dqkv = self._get('g_qkv', (rows, 3 * D))
        for li in range(len(self.layers) - 1, -1, -1):
            L = self.layers[li]
            x_in, xm, qkv, u, att = saved[li]
            dh = self._get('g_hid', (rows, L['hidden']))
            self._gemm(dx, L['fc2_wd'], dh, rows, D, L['hidden'], D, L['hidden'], mask=u, flags=F_GELU_BWD)   # du = (dx W2) * gelu'(u)
            dln = self._get('g_ln', (rows, D))
            self._gemm(dh, L['fc1_wd'], dln, rows, L['hidden'], D, L['hidden'], D)
            dxm = self._get('g_xm', (B, T, D))
            _lib.check(lib.rart_layernorm_bwd_bf16(_lib.ptr(dln), _lib.ptr(xm), _lib.ptr(L['n2g']), _lib.ptr(dx), _lib.ptr(dxm),
                                                   rows, D, D, D, D, D, 1e-6, sp))
            datt = self._get('g_att', (rows, D))
            self._gemm(dxm, L['proj_wd'], datt, rows, D, D, D, D)
            if self.fused_attention_bwd:
                _lib.check(lib.rart_vit_attention_bwd(_lib.ptr(qkv), _lib.ptr(att), _lib.ptr(datt), _lib.ptr(dqkv), B, T,
                                                      self.H, self.hd, sp))
            else:
                self._attention_bwd(qkv, datt, dqkv, B, T)
            self._gemm(dqkv, L['qkv_wd'], dln, rows, 3 * D, D, 3 * D, D)
            _lib.check(lib.rart_layernorm_bwd_bf16(_lib.ptr(dln), _lib.ptr(x_in), _lib.ptr(L['n1g']), _lib.ptr(dxm), _lib.ptr(dx),
                                                   rows, D, D, D, D, D, 1e-6, sp))
        # patch embedding: d(patches)[b][p][c*ps*ps + r*ps + s] = dx[b][1 + p][:] . Wpe ; class token / position rows drop out
        kk = 3 * self.ps * self.ps
        dpatch = self._get('g_patch', (B * P, kk))
        self._gemm(dx.view(rows, D)[1:], self.pe_wd, dpatch, B * P, D, kk, D, kk, rows_per_image=P, src_rows_per_image=T)
        grad = torch.empty(B, 3, Himg, Wimg, dtype=torch.float32, device=self.device)
        _lib.check(lib.rart_vit_unpatchify_f32(_lib.ptr(dpatch), _lib.ptr(grad), B, Himg, Wimg, self.ps, kk,
                                               (ctypes.c_float * 3)(*std), sp))
        return logits, loss, grad, pred


    # ------------------------------------------------------------------ reference-precision ("bf16x3" / "fp32x") mode
    def _gemm_pair(self, a, w, dst, M, N, K, lda, ldc, ldw=None, bias=None, res=None, flags=0, aux=None, w_rows=None,
                   rows_per_image=0, src_rows_per_image=0, src_row_off=0, dst_rows_per_image=0, dst_row_off=0, batched=None,
                   a_off=0, w_off=0, dst_off=0):
        """(a_hi + a_lo)[M][K] . (w_hi + w_lo)[N][K]^T -> dst (pair, or fp32 with F_OUT_F32) on rart_gemm_pair_bf16.  a / w / dst /
        res / aux are pair tensors [2][...] (dst fp32: a plain tensor); *_off are element offsets inside a plane (head / column
        slices); batched = dict(n, inner, a=(outer, inner), w=(outer, inner), c=(outer, inner))."""
        d = _lib.GemmPairDesc()
        es = 2                                                         # bytes per bf16 element
        d.a_hi, d.a_lo = a[0].data_ptr() + a_off * es, a[1].data_ptr() + a_off * es
        d.w_hi, d.w_lo = w[0].data_ptr() + w_off * es, w[1].data_ptr() + w_off * es
        d.bias = bias.data_ptr() if bias is not None else None
        if res is not None:
            d.res_hi, d.res_lo = res[0].data_ptr() + dst_off * es, res[1].data_ptr() + dst_off * es
        if flags & F_OUT_F32:
            d.dst_hi, d.dst_lo = dst.data_ptr() + dst_off * 4, None
        else:
            d.dst_hi, d.dst_lo = dst[0].data_ptr() + dst_off * es, dst[1].data_ptr() + dst_off * es
        if aux is not None:
            d.aux_hi, d.aux_lo = aux[0].data_ptr() + dst_off * es, aux[1].data_ptr() + dst_off * es
        d.M, d.N, d.K, d.lda, d.ldw, d.ldc = M, N, K, lda, (ldw or K), ldc
        il = self._w_il.get(w.data_ptr()) if (w_off == 0 and ldw is None and not batched) else None
        if il is not None and il.shape[1] == 2 * K:
            d.w_hi, d.w_lo, d.ldw = il.data_ptr(), il.data_ptr() + 64, 2 * K
            flags |= 16
        d.w_rows = w_rows if w_rows is not None else w.shape[-2]
        d.rows_per_image, d.src_rows_per_image, d.src_row_off = rows_per_image, src_rows_per_image, src_row_off
        d.dst_rows_per_image, d.dst_row_off = dst_rows_per_image, dst_row_off
        d.flags = flags
        nz = 1
        if batched:
            nz = d.n_batched = batched['n']
            d.z_inner = batched['inner']
            d.a_z_outer, d.a_z_inner = batched['a']
            d.w_z_outer, d.w_z_inner = batched['w']
            d.c_z_outer, d.c_z_inner = batched['c']
        if self.profile is not None:
            torch = _lib.require_gpu()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(self.lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr()))
            e1.record()
            self.profile.append((3 * 2.0 * nz * M * N * K, e0, e1, 'gemm_pair'))       # MFMA FLOPs issued: three products
            return
        _lib.check(self.lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr()))

    def _scores_probs_x3(self, qkv, B, T):
        """S = Q K^T (fp32) and P = softmax(S / sqrt(d)) (pair) of one layer, batched over (image, head)"""
        lib, sp = self.lib, _lib.stream_ptr()
        D, H, hd = self.D, self.H, self.hd
        t_pad, s_ld, BH = (T + 31) // 32 * 32, (T + 7) // 8 * 8, B * H
        torch = _lib.require_gpu()
        scores = self._get('x3_scores', (BH, T, s_ld), torch.float32)
        probs = self._get('x3_probs', (2, BH, T, t_pad))
        self._gemm_pair(qkv, qkv, scores, T, s_ld, hd, 3 * D, s_ld, ldw=3 * D, flags=F_OUT_F32, w_rows=T, w_off=D,
                        batched=dict(n=BH, inner=H, a=(T * 3 * D, hd), w=(T * 3 * D, hd), c=(H * T * s_ld, T * s_ld)))
        _lib.check(lib.rart_softmax_rows_pair(_lib.ptr(scores), _lib.ptr(probs[0]), _lib.ptr(probs[1]), BH * T, T, s_ld, t_pad,
                                              float(hd) ** -0.5, sp))
        return probs

    def _transpose_heads_x3(self, src, name, B, T, ld, off):
        """[2][B*H*hd][t_pad] token-contiguous copy of a per-head slice of a pair tensor [2][B*T][ld] (columns off + h*hd + d)"""
        lib, sp = self.lib, _lib.stream_ptr()
        H, hd = self.H, self.hd
        t_pad = (T + 31) // 32 * 32
        out = self._get(name, (2, B * H * hd, t_pad))
        for p in range(2):
            _lib.check(lib.rart_vit_transpose_v(_lib.ptr(src[p]), _lib.ptr(out[p]), B, T, H, hd, ld, off, t_pad, sp))
        return out

    def _forward_x3(self, src, src_is_u8, mean, std, keep=False):
        """The forward of `_forward` on pairs: rart_gemm_pair_bf16 for every contraction (patch embedding, qkv, Q K^T and P V per
        (image, head) as batched problems, proj, MLP with the exact GELU in the epilogue, head), csrc/vit_pair.hip for the rest."""
        torch = _lib.require_gpu()
        lib, sp = self.lib, _lib.stream_ptr()
        if src_is_u8:
            B, Himg, Wimg = src.shape[0], src.shape[1], src.shape[2]
        else:
            B, Himg, Wimg = src.shape[0], src.shape[2], src.shape[3]
        D, H, hd, ps = self.D, self.H, self.hd, self.ps
        P = (Himg // ps) * (Wimg // ps)
        T = P + 1
        assert T == self.tokens, 'image size does not match the position embedding'
        assert T <= 256, 'the pair soft-max rows hold at most 256 keys'
        kk = 3 * ps * ps
        X = self.x3
        patches = self._get('patches', (2, B, P, kk))
        meanf, stdf = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
        _lib.check(lib.rart_vit_patchify(_lib.ptr(src), 1 if src_is_u8 else 0, _lib.ptr(patches[0]), _lib.ptr(patches[1]),
                                         B, Himg, Wimg, ps, meanf, stdf, sp))
        x = self._get('x3_x0' if keep else 'x3_x', (2, B, T, D))
        self._gemm_pair(patches, X['pe_w'], x, B * P, D, kk, kk, D, bias=self.pe_b, rows_per_image=P, dst_rows_per_image=T,
                        dst_row_off=1)
        _lib.check(lib.rart_vit_add_pos_cls_pair(_lib.ptr(x[0]), _lib.ptr(x[1]), _lib.ptr(self.cls_pos0), _lib.ptr(self.pos), B, T, D, sp))
        rows = B * T
        t_pad = (T + 31) // 32 * 32
        ln = self._get('x3_ln', (2, B, T, D))
        saved = []
        for li, (L, XL) in enumerate(zip(self.layers, X['layers'])):
            qkv = self._get('x3_qkv%d' % li if keep else 'x3_qkv', (2, rows, 3 * D))
            xm = self._get('x3_xm%d' % li, (2, B, T, D)) if keep else x
            att = self._get('x3_att%d' % li if keep else 'x3_att', (2, B, T, D))     # the fused backward reads the output (delta)
            xo = self._get('x3_x%d' % (li + 1), (2, B, T, D)) if keep else x
            _lib.check(lib.rart_layernorm_pair(_lib.ptr(x[0]), _lib.ptr(x[1]), _lib.ptr(L['n1g']), _lib.ptr(L['n1b']), _lib.ptr(ln[0]),
                                               _lib.ptr(ln[1]), rows, D, D, D, 1e-6, sp))
            self._gemm_pair(ln, XL['qkv_w'], qkv, rows, 3 * D, D, D, 3 * D, bias=L['qkv_b'])
            if self.fused_attention and hd == 64 and T <= 224:
                _lib.check(lib.rart_vit_attention_pair(_lib.ptr(qkv[0]), _lib.ptr(qkv[1]), _lib.ptr(att[0]), _lib.ptr(att[1]), B, T, H, hd, sp))
            else:           # the decomposition into batched products with fp32 score-sized temporaries (cross-check)
                probs = self._scores_probs_x3(qkv, B, T)
                vt = self._transpose_heads_x3(qkv, 'x3_vt', B, T, 3 * D, 2 * D)
                self._gemm_pair(probs, vt, att, T, hd, t_pad, t_pad, D, ldw=t_pad, w_rows=hd,
                                batched=dict(n=B * H, inner=H, a=(H * T * t_pad, T * t_pad), w=(H * hd * t_pad, hd * t_pad), c=(T * D, hd)))
            self._gemm_pair(att, XL['proj_w'], xm, rows, D, D, D, D, bias=L['proj_b'], res=x)
            _lib.check(lib.rart_layernorm_pair(_lib.ptr(xm[0]), _lib.ptr(xm[1]), _lib.ptr(L['n2g']), _lib.ptr(L['n2b']), _lib.ptr(ln[0]),
                                               _lib.ptr(ln[1]), rows, D, D, D, 1e-6, sp))
            hid = self._get('x3_hid', (2, B, T, L['hidden']))
            if keep:
                u = self._get('x3_u%d' % li, (2, B, T, L['hidden']))
                self._gemm_pair(ln, XL['fc1_w'], hid, rows, L['hidden'], D, D, L['hidden'], bias=L['fc1_b'], flags=F_GELU_KEEP, aux=u)
                saved.append((x, xm, qkv, u, att))
            else:
                self._gemm_pair(ln, XL['fc1_w'], hid, rows, L['hidden'], D, D, L['hidden'], bias=L['fc1_b'], flags=F_GELU)
            self._gemm_pair(hid, XL['fc2_w'], xo, rows, D, L['hidden'], L['hidden'], D, bias=L['fc2_b'], res=xm)
            x = xo
        if keep:
            self._saved = (saved, x, (B, Himg, Wimg, P, T))
        cls = self._get('x3_cls', (2, B, D))
        _lib.check(lib.rart_layernorm_pair(_lib.ptr(x[0]), _lib.ptr(x[1]), _lib.ptr(self.ng), _lib.ptr(self.nb), _lib.ptr(cls[0]),
                                           _lib.ptr(cls[1]), B, D, T * D, D, 1e-6, sp))
        logits = torch.empty(B, self.n_classes, dtype=torch.float32, device=self.device)
        self._gemm_pair(cls, X['head_w'], logits, B, self.n_classes, D, D, self.n_classes, bias=self.head_b, flags=F_OUT_F32)
        return logits

    def _backward_x3(self, dl, std):
        """d(loss)/d(x01) from the fp32 loss gradient dl [B][classes]: the backward-to-input chain of `forward_backward` on pairs.
        The attention backward is the two-launch fused pair kernel (csrc/vit_pair.hip) or, with `fused_attention_bwd` off (the
        cross-check), the decomposition into batched products (S = Q K^T recomputed, dP = dO V^T, dQ = dS K, dK = dS^T Q,
        dV = P^T dO) with fp32 score-sized temporaries; GELU' runs in the fc2 dgrad epilogue."""
        torch = _lib.require_gpu()
        lib, sp = self.lib, _lib.stream_ptr()
        saved, x_last, (B, Himg, Wimg, P, T) = self._saved
        D, H, hd, rows = self.D, self.H, self.hd, B * T
        X = self.x3
        t_pad, s_ld, BH = (T + 31) // 32 * 32, (T + 7) // 8 * 8, B * H
        scale = float(hd) ** -0.5
        dlp = self._get('x3_dl', (2, B, self.head_kpad))
        _lib.check(lib.rart_f32_to_pair_rows(_lib.ptr(dl), _lib.ptr(dlp[0]), (dlp[1].data_ptr() - dlp[0].data_ptr()) // 2, B,
                                             self.n_classes, self.head_kpad, sp))
        dcls = self._get('x3_dcls', (2, B, D))
        self._gemm_pair(dlp, X['head_wd'], dcls, B, D, self.head_kpad, self.head_kpad, D)
        dx = self._get('x3_g_x_a', (2, B, T, D))
        dx.zero_()                                      # only the class token receives gradient from the head
        _lib.check(lib.rart_layernorm_bwd_pair(_lib.ptr(dcls[0]), _lib.ptr(dcls[1]), _lib.ptr(x_last[0]), _lib.ptr(x_last[1]),
                                               _lib.ptr(self.ng), None, None, _lib.ptr(dx[0]), _lib.ptr(dx[1]), B, D, D, T * D, 0, T * D,
                                               1e-6, sp))
        dqkv = self._get('x3_g_qkv', (2, rows, 3 * D))
        zero = (ctypes.c_int * 1)(0)
        m_all = BH * t_pad
        for li in range(len(self.layers) - 1, -1, -1):
            L, XL = self.layers[li], X['layers'][li]
            x_in, xm, qkv, u, att = saved[li]
            dh = self._get('x3_g_hid', (2, rows, L['hidden']))
            self._gemm_pair(dx, XL['fc2_wd'], dh, rows, L['hidden'], D, D, L['hidden'], flags=F_GELU_BWD, aux=u)    # du = (dx W2) gelu'(u)
            dln = self._get('x3_g_ln', (2, rows, D))
            self._gemm_pair(dh, XL['fc1_wd'], dln, rows, D, L['hidden'], L['hidden'], D)
            dxm = self._get('x3_g_xm', (2, B, T, D))
            _lib.check(lib.rart_layernorm_bwd_pair(_lib.ptr(dln[0]), _lib.ptr(dln[1]), _lib.ptr(xm[0]), _lib.ptr(xm[1]), _lib.ptr(L['n2g']),
                                                   _lib.ptr(dx[0]), _lib.ptr(dx[1]), _lib.ptr(dxm[0]), _lib.ptr(dxm[1]), rows, D, D, D, D, D,
                                                   1e-6, sp))
            datt = self._get('x3_g_att', (2, rows, D))
            self._gemm_pair(dxm, XL['proj_wd'], datt, rows, D, D, D, D)
            # ---- attention backward, per (image, head)
            if self.fused_attention_bwd and hd == 64 and T <= 224:
                stats = self._get('x3_att_stats', (BH, t_pad, 4), torch.float32)
                _lib.check(lib.rart_vit_attention_bwd_pair(_lib.ptr(qkv[0]), _lib.ptr(qkv[1]), _lib.ptr(att[0]), _lib.ptr(att[1]),
                                                           _lib.ptr(datt[0]), _lib.ptr(datt[1]), _lib.ptr(dqkv[0]), _lib.ptr(dqkv[1]),
                                                           _lib.ptr(stats), B, T, H, hd, sp))
            else:
                probs = self._scores_probs_x3(qkv, B, T)
                dprobs = self._get('x3_dprobs', (BH, T, s_ld), torch.float32)
                self._gemm_pair(datt, qkv, dprobs, T, s_ld, hd, D, s_ld, ldw=3 * D, flags=F_OUT_F32, w_rows=T, w_off=2 * D,
                                batched=dict(n=BH, inner=H, a=(T * D, hd), w=(T * 3 * D, hd), c=(H * T * s_ld, T * s_ld)))       # dP = dO V^T
                ds = self._get('x3_dscores', (2, BH, T, t_pad))
                _lib.check(lib.rart_softmax_bwd_rows_pair(_lib.ptr(probs[0]), _lib.ptr(probs[1]), _lib.ptr(dprobs), _lib.ptr(ds[0]),
                                                          _lib.ptr(ds[1]), BH * T, T, t_pad, s_ld, t_pad, scale, sp))
                kt = self._transpose_heads_x3(qkv, 'x3_kt', B, T, 3 * D, D)
                qt = self._transpose_heads_x3(qkv, 'x3_qt', B, T, 3 * D, 0)
                dot = self._transpose_heads_x3(datt, 'x3_dot', B, T, D, 0)
                hb = dict(n=BH, inner=H, a=(H * T * t_pad, T * t_pad), w=(H * hd * t_pad, hd * t_pad), c=(T * 3 * D, hd))
                self._gemm_pair(ds, kt, dqkv, T, hd, t_pad, t_pad, 3 * D, ldw=t_pad, w_rows=hd, batched=hb)                   # dQ = dS K
                # query-contiguous copies of dS and P: [t_pad (key)][BH * t_pad (image-head, query)]; queries past T are zero columns
                ds_t = self._get('x3_ds_t', (2, t_pad, m_all))
                p_t = self._get('x3_p_t', (2, t_pad, m_all))
                for src_m, dst_m in ((ds, ds_t), (probs, p_t)):
                    for p in range(2):
                        _lib.check(lib.rart_transpose_gather_bf16(_lib.ptr(src_m[p]), _lib.ptr(dst_m[p]), BH, T, 1, t_pad, t_pad, 1, 1, 1, 1,
                                                                  zero, zero, m_all, 0, 0, sp))
                tb = dict(n=BH, inner=H, a=(H * t_pad, t_pad), w=(H * hd * t_pad, hd * t_pad), c=(T * 3 * D, hd))
                self._gemm_pair(ds_t, qt, dqkv, T, hd, t_pad, m_all, 3 * D, ldw=t_pad, w_rows=hd, batched=tb, dst_off=D)      # dK = dS^T Q
                self._gemm_pair(p_t, dot, dqkv, T, hd, t_pad, m_all, 3 * D, ldw=t_pad, w_rows=hd, batched=tb, dst_off=2 * D)  # dV = P^T dO
            self._gemm_pair(dqkv, XL['qkv_wd'], dln, rows, D, 3 * D, 3 * D, D)
            _lib.check(lib.rart_layernorm_bwd_pair(_lib.ptr(dln[0]), _lib.ptr(dln[1]), _lib.ptr(x_in[0]), _lib.ptr(x_in[1]),
                                                   _lib.ptr(L['n1g']), _lib.ptr(dxm[0]), _lib.ptr(dxm[1]), _lib.ptr(dx[0]), _lib.ptr(dx[1]),
                                                   rows, D, D, D, D, D, 1e-6, sp))
        # patch embedding: d(patches)[b][p][:] = dx[b][1 + p][:] . Wpe (fp32) ; class token / position rows drop out
        kk = 3 * self.ps * self.ps
        dpatch = self._get('x3_g_patch', (B * P, kk), torch.float32)
        self._gemm_pair(dx, X['pe_wd'], dpatch, B * P, kk, D, D, kk, flags=F_OUT_F32, rows_per_image=P, src_rows_per_image=T,
                        src_row_off=1)
        grad = torch.empty(B, 3, Himg, Wimg, dtype=torch.float32, device=self.device)
        _lib.check(lib.rart_vit_unpatchify_from_f32(_lib.ptr(dpatch), _lib.ptr(grad), B, Himg, Wimg, self.ps, kk,
                                                    (ctypes.c_float * 3)(*std), sp))
        return grad

    def logits(self, x01, mean, std):
        return self._forward(x01.detach().float().contiguous(), False, mean, std)

    def logits_from_u8(self, batch_u8, mean, std):
        return self._forward(batch_u8, True, mean, std)
