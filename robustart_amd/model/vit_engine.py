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


class ViTEngine:
    def __init__(self, model, device='cuda'):
        torch = _lib.require_gpu()
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.precision = 'bf16'          # the reference-precision mode exists for ResNet-50 only (model/engine.py)
        m = model
        self.D, self.H, self.ps = m.embed_dim, m.num_heads, m.patch_size
        self.hd = self.D // self.H
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

    def logits(self, x01, mean, std):
        return self._forward(x01.detach().float().contiguous(), False, mean, std)

    def logits_from_u8(self, batch_u8, mean, std):
        return self._forward(batch_u8, True, mean, std)
