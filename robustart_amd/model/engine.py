"""Eval-mode ResNet-50 on the hand-written HIP kernels: forward and backward-to-input.

This is the `f_model(x)` + autograd step that every iteration of the reference's attacks
performs (RobustART/noise/utils/adv/attack.py:21-22; Attacks/autoattack/autopgd_base.py:271-289,
367-384; Attacks/imfgsm_attack.py:82-84) and the clean / corrupted evaluation forward
(SURVEY.md 3.5).  Model definition: RobustART/model/__init__.py:1 -> absent submodule; the public
ResNet-50 v1.5 (robustart_amd/model/resnet_torch.py is the plain-PyTorch statement of it).

Host side (this file, PyTorch as plumbing only): fold BatchNorm (eval mode) into conv weight/bias,
lay the weights out for the implicit-GEMM kernel (forward: [Cout][tap][Cin]; backward-to-input:
[Cin][tap][Cout], one table per input-parity class for stride-2 convs), own the activation / gradient
buffers, and sequence the C-ABI calls.  Device side: rart_conv_igemm_bf16 (every conv, fc, and their
backward), rart_engine_* (stem input prep, pools, col2im), rart_logit_loss.  bf16 storage, fp32
accumulation; the fp32 pixels enter the stem as a hi+lo bf16 pair so eps-sized perturbations are kept.
"""
import ctypes
import os as _os

from .. import _lib

F_RELU, F_OUT_F32, F_MASK_BITS, F_PAIR = 1, 2, 16, 32
PRECISIONS = {'bf16': 'bf16', 'bf16x3': 'bf16x3', 'fp32x': 'bf16x3'}


def _bf16(t):
    import torch
    return t.to(torch.bfloat16).contiguous()


def _pad_rows(w2d, mult):
    import torch
    rows = w2d.shape[0]
    pr = (rows + mult - 1) // mult * mult
    if pr == rows:
        return w2d
    return torch.cat([w2d, torch.zeros(pr - rows, w2d.shape[1], dtype=w2d.dtype, device=w2d.device)], 0)


def _rows_mult(n_cols):
    return 128 if n_cols > 64 else 64


class _Conv:
    """One folded conv layer: forward table + backward-to-input tables."""

    def __init__(self, conv, bn, device, split=False):
        import torch
        w = conv.weight.detach().float()
        if bn is not None:
            inv = (bn.running_var.detach().float() + bn.eps).rsqrt() * bn.weight.detach().float()
            w = w * inv.view(-1, 1, 1, 1)
            b = bn.bias.detach().float() - bn.running_mean.detach().float() * inv
        else:
            b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(w.shape[0])
        self.cout, self.cin, self.r, self.s = w.shape
        self.stride = conv.stride[0]
        self.pad = conv.padding[0]
        self.w_folded = w                    # fp32, for the reference emulation in tests
        self.b_folded = b
        self.bias = b.contiguous().to(device)
        self.fwd_taps = [(r - self.pad, s - self.pad) for r in range(self.r) for s in range(self.s)]
        wb = w.to(torch.bfloat16).float()    # the values the bf16 kernels see
        if not split:
            self.w_fwd, self.bwd = self._tables(wb, device)
        else:
            # reference-precision mode: w = hi + lo (two bf16 pieces, 16 significand bits); a weight row is the
            # concatenation [hi | lo | hi] matching the products x_hi.w_hi + x_hi.w_lo + x_lo.w_hi (rart_conv_desc flag 32)
            wl = (w - wb).to(torch.bfloat16).float()
            fh, bh = self._tables(wb, device)
            fl, bl = self._tables(wl, device)
            self.w_fwd = torch.cat([fh, fl, fh], 1).contiguous()
            self.bwd = [(par, taps, None if th is None else torch.cat([th, tl, th], 1).contiguous())
                        for (par, taps, th), (_, _, tl) in zip(bh, bl)]

    def _tables(self, wb, device):
        """-> (forward table [cout][tap][cin], backward-to-input tables [(parity, taps, [cin][tap][cout])]) of the bf16-exact
        fp32 weights wb [cout][cin][r][s]."""
        import torch
        # forward: rows = cout, k = (r*S + s)*Cin + c
        w_fwd = _bf16(_pad_rows(wb.permute(0, 2, 3, 1).reshape(self.cout, -1), _rows_mult(self.cout))).to(device)
        # backward to input: rows = cin, k = tap*Cout + cout
        bwd = []    # list of (parity (ph,pw) or None, taps [(dy,dx)], weight)
        if self.stride == 1:
            taps, cols = [], []
            for r in range(self.r):
                for s in range(self.s):
                    taps.append((self.pad - r, self.pad - s))
                    cols.append(wb[:, :, r, s].t())                     # [cin][cout]
            wd = torch.cat(cols, 1)
            bwd.append((None, taps, _bf16(_pad_rows(wd, _rows_mult(self.cin))).to(device)))
        else:
            assert self.stride == 2
            for ph in range(2):
                for pw in range(2):
                    taps, cols = [], []
                    for r in range(self.r):
                        if (ph + self.pad - r) % 2:
                            continue
                        for s in range(self.s):
                            if (pw + self.pad - s) % 2:
                                continue
                            taps.append(((ph + self.pad - r) // 2, (pw + self.pad - s) // 2))
                            cols.append(wb[:, :, r, s].t())
                    if not taps:
                        bwd.append(((ph, pw), [], None))
                        continue
                    wd = torch.cat(cols, 1)
                    bwd.append(((ph, pw), taps, _bf16(_pad_rows(wd, _rows_mult(self.cin))).to(device)))
        return w_fwd, bwd


def _cints(vals):
    return (ctypes.c_int * max(len(vals), 1))(*vals)


class ResNet50Engine:
    """Hand-written HIP eval engine for robustart_amd.model.resnet_torch.ResNet."""

    def __init__(self, model, device='cuda', precision='bf16'):
        """precision: 'bf16' -- bf16 storage, fp32 accumulation (the fast path; logits within ~3e-3 of the fp32 network);
        'bf16x3' (alias 'fp32x') -- the REFERENCE-PRECISION mode: the reference runs fp32 (adv/attack.py:20-23,
        autopgd_base.py:271-289) and the north star asks for logits within 1e-4 of it, so every activation, gradient and
        weight is a hi + lo pair of bf16 values (16 significand bits) and every contraction the three MFMA products
        hi.hi + hi.lo + lo.hi with fp32 accumulation (`_forward_x3`): logits within ~1e-5 of the fp32 network's scale."""
        torch = _lib.require_gpu()
        self.lib = _lib.load()
        self.device = torch.device(device)
        if precision not in PRECISIONS:
            raise ValueError('precision must be one of %s' % sorted(PRECISIONS))
        self.precision = PRECISIONS[precision]
        split = self.precision == 'bf16x3'
        m = model
        assert not m.training, 'the attack / eval engine folds BatchNorm: call model.eval() first'
        dev = self.device
        # ---- stem: 7x7/2 conv on the padded 4-channel hi/lo image; a "tap" = one filter row, 8 px x 4 ch
        st = _Conv(m.conv1, m.bn1, dev, split)
        self.stem = st
        wb = st.w_folded.to(torch.bfloat16).float()                       # [64][3][7][7]
        wrow = torch.zeros(64, 7, 8, 4)
        wrow[:, :, :7, :3] = wb.permute(0, 2, 3, 1)                      # [cout][r][s][c]
        wrow = wrow.reshape(64, 7 * 32)
        self.stem_w = _bf16(torch.cat([wrow, wrow], 1)).to(dev)          # hi taps then lo taps
        # stem backward: patches[(r*7+s)*3+c] = sum_k dz[k] * W[k][c][r][s]; 147 rows zero-padded to the tile
        wp = wb.permute(2, 3, 1, 0).reshape(147, 64)
        self.stem_patch_cols = 152                                        # 147 rounded up to 8
        self.stem_wd = _bf16(_pad_rows(wp, _rows_mult(self.stem_patch_cols))).to(dev)
        self.stem_wt = self._stem_bwd_table(wb).to(dev)                  # fused stem backward (stem_fused.hip)
        if split:
            wl = (st.w_folded - wb).to(torch.bfloat16).float()
            wrow_l = torch.zeros(64, 7, 8, 4)
            wrow_l[:, :, :7, :3] = wl.permute(0, 2, 3, 1)
            wrow_l = wrow_l.reshape(64, 7 * 32)
            self.stem_w = _bf16(torch.cat([wrow, wrow_l, wrow], 1)).to(dev)      # x_hi.w_hi, x_hi.w_lo, x_lo.w_hi row taps
            wpl = wl.permute(2, 3, 1, 0).reshape(147, 64)
            self.stem_wd = _bf16(_pad_rows(torch.cat([wp, wpl, wp], 1), _rows_mult(self.stem_patch_cols))).to(dev)
            self.stem_w_pair = _bf16(torch.stack([wrow, wrow_l])).contiguous().to(dev)   # fused pair stem forward (stem_pair.hip): [2][64][224]
            wt32 = self._stem_bwd_table(st.w_folded, dtype=torch.float32)           # fused pair stem backward (stem_pair.hip)
            wt_hi = wt32.to(torch.bfloat16)
            self.stem_wt_pair = torch.stack([wt_hi, (wt32 - wt_hi.float()).to(torch.bfloat16)]).contiguous().to(dev)
        self.fused_stem_bwd = True       # False: max-pool bwd -> patches GEMM -> col2im (kept as the cross-check)
        self.sign_bit_masks = True       # False: the backward reads the bf16 activations for their ReLU sign (cross-check)
        self.halo_conv3x3 = True         # False: layer1 / layer2 3x3 convs on the generic implicit GEMM (cross-check)
        self.fused_stem_fwd = True       # False: prep_input -> row-tap GEMM -> max pool (cross-check; keeps acts['y1'])
        self.fused_bottleneck = True     # False: layer1's identity blocks as three conv launches each (cross-check)
        # identity blocks of layer2 / layer3 / layer4 on the image-resident fused kernels (bottleneck{28,14,7}_fused.hip):
        self.fused_bottleneck14 = True   # False: all of them as three conv launches each (cross-check)
        self.fused_bottleneck28 = True   # False: only layer2's
        self.fused_bottleneck7 = True    # False: only layer4's
        self.fused_bottleneck_s2 = True  # False: the stride-2 first blocks of layer2 / layer3 as four conv launches in the forward (cross-check)
        self.fused_bottleneck_s2_bwd = True   # False: their backward-to-input as seven conv launches (cross-check)
        self.small_m_fc = True           # False: the classifier head and its backward on the implicit GEMM (cross-check)
        self.pair_tile = (0, 0)
        self.fused_next_pair = True      # reference-precision mode: ... and the neighbouring block's 1x1 reduction in the same launch; False: its own launch (cross-check)
        self.fused_next_channels = (64,)  # ... for these mid-channel counts (measured at B = 256: layer1 -0.35 ms per gradient evaluation; layer2's instance sits at 256 VGPRs and loses 0.6 ms)
        # ... for 3x3 layers of these widths (layer1 = 64, layer2 = 128).  Round 6: layer2 left the list -- with the ping-pong kernels and the
        # 224-row tiles its two launches are faster than its tail kernel at B = 256 (196 workgroups per XCD on 64 resident: 3.06 passes):
        # 20.12 -> 19.98 ms per gradient evaluation, same bits (scratch/r6/tail_channels.py)
        self.fused_tail_channels = (64,)
        self.fused_tail_pair = True      # reference-precision mode: 3x3 + 1x1 expansion of a Bottleneck as one launch (conv_tail_pair.hip); False: two launches (cross-check)
        self.pair_gemm_kernel = True     # reference-precision mode: False = the three products as 3 x the taps of the implicit GEMM (round 3; cross-check)
        self.blocks = []
        for layer in (m.layer1, m.layer2, m.layer3, m.layer4):
            for blk in layer:
                ds = _Conv(blk.downsample[0], blk.downsample[1], dev, split) if blk.downsample is not None else None
                self.blocks.append((_Conv(blk.conv1, blk.bn1, dev, split), _Conv(blk.conv2, blk.bn2, dev, split),
                                    _Conv(blk.conv3, blk.bn3, dev, split), ds))
        # ---- classifier
        wfc = m.fc.weight.detach().float()
        self.n_classes, self.fc_in = wfc.shape
        wfb = wfc.to(torch.bfloat16).float()
        self.fc_w = _bf16(_pad_rows(wfb, 128)).to(dev)                    # [1024][2048]
        self.fc_b = m.fc.bias.detach().float().contiguous().to(dev)
        self.fc_kpad = (self.n_classes + 31) // 32 * 32                   # 1024
        wt = torch.zeros(self.fc_in, self.fc_kpad)
        wt[:, :self.n_classes] = wfb.t()
        self.fc_wd = _bf16(wt).to(dev)                                    # [2048][1024]
        if split:
            wfl = (wfc - wfb).to(torch.bfloat16).float()
            fh, fl = _pad_rows(wfb, 128), _pad_rows(wfl, 128)
            self.fc_w = _bf16(torch.cat([fh, fl, fh], 1)).to(dev)         # [1024][3 * 2048]
            wtl = torch.zeros(self.fc_in, self.fc_kpad)
            wtl[:, :self.n_classes] = wfl.t()
            self.fc_wd = _bf16(torch.cat([wt, wtl, wt], 1)).to(dev)       # [2048][3 * 1024]
        self._buf = {}
        self.profile = None      # set to a list to record (flops, start_event, end_event) per GEMM launch
        # round 5 experiment, OFF by default: the pair GEMM's weight tables with the hi and the lo slice of a 32-deep K step side by side (one
        # 128-byte line per row and step instead of two half lines K apart).  2-5 % per K-deep launch when a shape is replayed back to back
        # (profiles/r05_pair_knockouts.txt), nothing on the whole gradient evaluation (20.61 vs 20.63 ms) and +0.9 % on the forward
        # (scratch/r5/ab_engine_x3.py): the isolated replay keeps the tables hot in L2, the network does not.  RART_PAIR_WIL=1 turns it on.
        self.pair_w_interleaved = _os.environ.get('RART_PAIR_WIL', '0') == '1'
        self._w_il = {}
        if not split:
            self._pack_frag_tables()
        else:
            self._pack_tail_tables()

    def _pack_tail_tables(self):
        """Reference-precision mode: the 1x1 expansion tables of rart_conv3x3_tail_pair in MFMA fragment order, as [2 (hi, lo)][rows * k]:
        element ((blk * (k / 16) + s) * 64 + h * 32 + r) * 8 + e = T[blk * 32 + r][16 s + 8 h + e].  `tail_fwd` on conv3 (rows = its
        output channels), `tail_bwd` on conv1 of an identity block (rows = its INPUT channels: the transposed table of the backward)."""
        torch = _lib.require_gpu()

        def frag(tab, rows, k):          # tab: the [rows_padded][hi | lo | hi] table of _Conv(split=True)
            out = []
            for pl in range(2):
                w = tab[:rows, pl * k:(pl + 1) * k]
                out.append(w.reshape(rows // 32, 32, k // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1))
            return torch.stack(out).contiguous()
        def frag_next(tab, rows, k):     # the neighbour's reduction [rows = C][k = 4C], fragment order by 64-wide K chunk
            out = []
            for pl in range(2):
                w = tab[:rows, pl * k:(pl + 1) * k]
                out.append(w.reshape(rows // 32, 32, k // 64, 4, 2, 8).permute(2, 0, 3, 4, 1, 5).contiguous().view(-1))
            return torch.stack(out).contiguous()
        for bi, (ca, cb, cc, ds) in enumerate(self.blocks):
            if not (cb.r == 3 and cb.stride == 1 and cb.pad == 1 and cb.cin == cb.cout and self.lib.rart_conv3x3_tail_pair_supported(cb.cin)):
                continue
            if cc.cin == cb.cout and cc.cout == 4 * cb.cout and cc.r == 1 and cc.stride == 1:
                cc.tail_fwd = frag(cc.w_fwd, cc.cout, cc.cin)
                if bi + 1 < len(self.blocks):        # the next block's conv1 on this block's output tile
                    na = self.blocks[bi + 1][0]
                    if na.r == 1 and na.stride == 1 and na.cin == cc.cout and na.cout == cb.cout:
                        na.next_fwd = frag_next(na.w_fwd, na.cout, na.cin)
            if ds is None and ca.cout == cb.cin and ca.cin == 4 * cb.cin and ca.r == 1 and ca.stride == 1:
                ca.tail_bwd = frag(ca.bwd[0][2], ca.cin, ca.cout)
                pc = self.blocks[bi - 1][2]          # the previous block's conv3^T on this block's input-gradient tile
                if bi > 0 and pc.r == 1 and pc.stride == 1 and pc.cout == ca.cin and pc.cin == cb.cin:
                    pc.next_bwd = frag_next(pc.bwd[0][2], pc.cin, pc.cout)

    def _pack_frag_tables(self, record=None, sums=None):
        """MFMA-fragment-ordered copies of the 3x3 tables the LDS-resident kernels (conv3x3_halo.hip, bottleneck_fused.hip)
        stream from L2: a fragment load then reads 1 KiB contiguous instead of touching 32 cache lines.
        record / sums (refold): instead of launching, append every (source table, fragment table, rows, k) job to `record` and
        every (conv3 bias, shortcut bias, summed bias) triple to `sums` -- the allocations happen either way."""
        torch = _lib.require_gpu()
        sp = _lib.stream_ptr()
        lib_pack = self.lib.rart_pack_frag_bf16

        def pack(tab, dst, rows, k):
            if record is not None:
                record.append((tab, dst, rows, k))
            else:
                _lib.check(lib_pack(_lib.ptr(tab), _lib.ptr(dst), rows, k, sp))

        def bias_sum(a, b, out):
            if sums is not None:
                sums.append((a, b, out))
            else:
                torch.add(a, b, out=out)
        for ca, cb, cc, ds in self.blocks:
            if cb.r == 3 and cb.stride == 1 and cb.cin == cb.cout and cb.cin in (64, 128, 256, 512):
                for name, tab in (('w_fwd_frag', cb.w_fwd), ('w_bwd_frag', cb.bwd[0][2])):
                    if getattr(cb, name, None) is None:
                        setattr(cb, name, torch.empty(9 * cb.cin * cb.cin, dtype=torch.bfloat16, device=self.device))
                    pack(tab, getattr(cb, name), cb.cin, 9 * cb.cin)
            if ds is None and cb.stride == 1 and (ca.cin, ca.cout, cc.cout) in ((1024, 256, 1024), (512, 128, 512), (2048, 512, 2048)):
                # identity blocks of layer2 / layer3 / layer4 for the image-resident fused kernels: both 1x1 tables in fragment order
                for c_, rows, k in ((ca, ca.cout, ca.cin), (cc, cc.cout, cc.cin)):
                    for name, tab in (('w_fwd_frag', c_.w_fwd), ('w_bwd_frag', c_.bwd[0][2])):
                        r_, k_ = (rows, k) if name == 'w_fwd_frag' else (k, rows)
                        if getattr(c_, name, None) is None:
                            setattr(c_, name, torch.empty(rows * k, dtype=torch.bfloat16, device=self.device))
                        pack(tab, getattr(c_, name), r_, k_)
            if (ds is not None and cb.stride == 2 and cb.r == 3 and ds.stride == 2 and ds.r == 1
                    and (ca.cin, ca.cout, cc.cout) in ((256, 128, 512), (512, 256, 1024))):
                # stride-2 first block of layer2 / layer3 for the fused forward kernel (bottleneck_s2_fused.hip): all four tables in
                # fragment order, conv3 + shortcut bias
                for c_, name, rows, k in ((ca, 's2_w1', ca.cout, ca.cin), (cb, 's2_w2', cb.cout, 9 * cb.cin), (cc, 's2_w3', cc.cout, cc.cin),
                                          (ds, 's2_wd', ds.cout, ds.cin)):
                    if getattr(c_, name, None) is None:
                        setattr(c_, name, torch.empty(rows * k, dtype=torch.bfloat16, device=self.device))
                    pack(c_.w_fwd, getattr(c_, name), rows, k)
                if getattr(ds, 'bias_sum', None) is None:
                    ds.bias_sum = torch.empty_like(cc.bias)
                bias_sum(cc.bias, ds.bias, ds.bias_sum)
                # ... and the backward kernel's: transposed tables of conv3 / conv1 / the shortcut, and the four input-parity-class
                # tables of the 3x3 / 2 (1 / 2 / 2 / 4 taps) packed one by one into a single buffer
                cm = ca.cout
                for c_, name, tab, rows, k in ((cc, 's2_w3t', cc.bwd[0][2], cm, cc.cout), (ca, 's2_w1t', ca.bwd[0][2], ca.cin, cm),
                                               (ds, 's2_wdt', ds.bwd[0][2], ds.cin, ds.cout)):
                    if getattr(c_, name, None) is None:
                        setattr(c_, name, torch.empty(rows * k, dtype=torch.bfloat16, device=self.device))
                    pack(tab, getattr(c_, name), rows, k)
                if getattr(cb, 's2_w2t', None) is None:
                    cb.s2_w2t = torch.empty(9 * cm * cm, dtype=torch.bfloat16, device=self.device)
                off = 0
                for (_, taps, tab) in cb.bwd:                  # parity classes (0,0) (0,1) (1,0) (1,1)
                    pack(tab, cb.s2_w2t[off:], cm, len(taps) * cm)
                    off += len(taps) * cm * cm
                assert off == 9 * cm * cm
            if (ds is not None and ds.stride == 1 and ds.r == 1 and cb.stride == 1 and ca.cin == 64 and ca.cout == 64
                    and cc.cout == 256):
                # first block of layer1 for the fused kernel: the shortcut table in fragment order, conv3 + shortcut bias
                if getattr(ds, 'w_fwd_frag', None) is None:
                    ds.w_fwd_frag = torch.empty(ds.cout * ds.cin, dtype=torch.bfloat16, device=self.device)
                    ds.bias_sum = torch.empty_like(cc.bias)
                pack(ds.w_fwd, ds.w_fwd_frag, ds.cout, ds.cin)
                bias_sum(cc.bias, ds.bias, ds.bias_sum)

    _stem_bwd_index = {}

    @staticmethod
    def _stem_bwd_table(wb, dtype=None):
        """bf16 (or `dtype`) [16][1024] table of rart_engine_stem_bwd_fused from the folded stem weights wb [64][3][7][7]:
        row (py*2+px)*3+c, column ((dp+1)*4+(dq+1))*64+k = W[k][c][py+3-2dp][px+3-2dq] (0 outside 0..6).  One gather + one scatter
        through index tensors built once per device (the adversarial-training loop calls this every step: 147 slice assignments
        were 147 tiny launches)."""
        import torch
        key = str(wb.device)
        idx = ResNet50Engine._stem_bwd_index.get(key)
        if idx is None:
            src, dst = [], []
            for py in range(2):
                for px in range(2):
                    for dp in range(-1, 3):
                        r = py + 3 - 2 * dp
                        if not 0 <= r <= 6:
                            continue
                        for dq in range(-1, 3):
                            s_ = px + 3 - 2 * dq
                            if not 0 <= s_ <= 6:
                                continue
                            for c in range(3):
                                row, blk = (py * 2 + px) * 3 + c, (dp + 1) * 4 + (dq + 1)
                                for k in range(64):
                                    dst.append(row * 1024 + blk * 64 + k)
                                    src.append(((k * 3 + c) * 7 + r) * 7 + s_)
            idx = (torch.tensor(src, dtype=torch.long, device=wb.device), torch.tensor(dst, dtype=torch.long, device=wb.device))
            ResNet50Engine._stem_bwd_index[key] = idx
        t = torch.zeros(16 * 1024, dtype=wb.dtype, device=wb.device)
        t[idx[1]] = wb.reshape(-1)[idx[0]]
        return t.reshape(16, 1024).to(dtype or torch.bfloat16).contiguous()

    # ------------------------------------------------------------------ re-fold from the live parameters
    def refold(self, model):
        """Re-derive every table from `model`'s CURRENT parameters and running statistics, on the GPU: the adversarial-training
        loop attacks the model it is training (cifar10/code/train.py:105-111), so the attack engine is refreshed every iteration.
        Round 4: the ~480 launches this took (five torch elementwise kernels + two to five rart_pack_conv_weight_bf16 calls per
        convolution, one rart_pack_frag_bf16 per fragment-ordered copy) are a handful now -- the BatchNorm fold runs on flat
        concatenated vectors, every conv table is one job of ONE rart_pack_jobs_bf16 launch, every fragment re-order one job of a
        second; the tables are bit-identical to the constructor's (tests/test_engine_gpu.py::test_refold_...)."""
        torch = _lib.require_gpu()
        if self.precision != 'bf16':
            raise NotImplementedError('refold() serves the adversarial-training loop, which attacks on the bf16 engine; build a '
                                      'new ResNet50Engine(model, precision=%r) from the updated weights instead' % self.precision)
        lib, sp = self.lib, _lib.stream_ptr()
        m = model
        pairs = [(self.stem, m.conv1, m.bn1)]
        bi = 0
        for layer in (m.layer1, m.layer2, m.layer3, m.layer4):
            for blk in layer:
                ca, cb, cc, ds = self.blocks[bi]
                pairs += [(ca, blk.conv1, blk.bn1), (cb, blk.conv2, blk.bn2), (cc, blk.conv3, blk.bn3)]
                if ds is not None:
                    pairs.append((ds, blk.downsample[0], blk.downsample[1]))
                bi += 1
        st = getattr(self, '_refold_state', None)
        key = tuple(conv.weight.data_ptr() for _, conv, _ in pairs)
        if st is None or st['key'] != key:
            st = self._build_refold_state(pairs, key)
        # ---- BatchNorm fold on flat vectors: scale = gamma * rsqrt(var + eps), bias = beta - mean * scale
        torch.cat([bn.running_var.detach() for _, _, bn in pairs], out=st['var'])
        torch.cat([bn.weight.detach() for _, _, bn in pairs], out=st['gamma'])
        torch.cat([bn.bias.detach() for _, _, bn in pairs], out=st['beta'])
        torch.cat([bn.running_mean.detach() for _, _, bn in pairs], out=st['mean'])
        torch.add(st['var'], st['eps'], out=st['scale'])
        st['scale'].rsqrt_().mul_(st['gamma'])
        torch.mul(st['mean'], st['scale'], out=st['tmp'])
        torch.sub(st['beta'], st['tmp'], out=st['bias'])        # c.bias of every layer is a view into st['bias']
        # ---- every conv table (forward + backward-to-input classes) in one launch, every fragment-ordered copy in a second
        _lib.check(lib.rart_pack_jobs_bf16(_lib.ptr(st['pack_jobs']), st['n_pack'], 48, sp))
        for w, scale, tab, c, trs, tr in st['big']:
            _lib.check(lib.rart_pack_conv_weight_bf16(w.data_ptr(), scale.data_ptr(), tab.data_ptr(), c.cout, c.cin, c.r, c.s, len(trs),
                                                      _cints([a for a, _ in trs]), _cints([b for _, b in trs]), tr, tab.shape[0], sp))
        # stem extras (row-tap table, patches table, fused-backward table) and the classifier
        sc = st['scale'][:64]
        wb = (m.conv1.weight.detach() * sc.view(-1, 1, 1, 1)).to(torch.bfloat16)          # [64][3][7][7]
        wrow = torch.zeros(64, 7, 8, 4, dtype=torch.bfloat16, device=self.device)
        wrow[:, :, :7, :3] = wb.permute(0, 2, 3, 1)
        wrow = wrow.reshape(64, 224)
        self.stem_w[:, :224] = wrow
        self.stem_w[:, 224:] = wrow
        self.stem_wd.zero_()
        self.stem_wd[:147] = wb.permute(2, 3, 1, 0).reshape(147, 64)
        self.stem_wt.copy_(self._stem_bwd_table(wb.float()))
        wfb = m.fc.weight.detach().to(torch.bfloat16)
        self.fc_w[:self.n_classes] = wfb
        self.fc_wd[:, :self.n_classes] = wfb.t()
        self.fc_b.copy_(m.fc.bias.detach())
        if st['n_frag']:
            _lib.check(lib.rart_pack_jobs_bf16(_lib.ptr(st['frag_jobs']), st['n_frag'], 32, sp))
        for cc_bias, ds_bias, out in st['bias_sums']:
            torch.add(cc_bias, ds_bias, out=out)

    def _build_refold_state(self, pairs, key):
        """Persistent buffers and the two device job lists of `refold` (built once per model: every pointer in them is stable --
        parameters live in the optimizer's arena, tables and flat vectors are allocated here or in the constructor)."""
        torch = _lib.require_gpu()
        dev = self.device
        n_tot = sum(c.cout for c, _, _ in pairs)
        eps = {float(bn.eps) for _, _, bn in pairs}
        assert len(eps) == 1, 'refold folds every BatchNorm with one eps'
        st = {'key': key, 'eps': eps.pop()}
        for nm in ('var', 'gamma', 'beta', 'mean', 'scale', 'tmp', 'bias'):
            st[nm] = torch.empty(n_tot, dtype=torch.float32, device=dev)
        jobs, big, off = [], [], 0
        for c, conv, bn in pairs:
            w = conv.weight
            assert w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()
            old = c.bias
            c.bias = st['bias'][off:off + c.cout]           # a view: the flat fold writes every layer's bias at once
            c.bias.copy_(old)
            scale_ptr = st['scale'][off:off + c.cout].data_ptr()
            rs = [(r, s_) for r in range(c.r) for s_ in range(c.s)]
            todo = [(c.w_fwd, rs, 0)]
            for parity, taps, tab in c.bwd:
                if tab is not None:
                    prs = rs if parity is None else [(r, s_) for r, s_ in rs if (parity[0] + c.pad - r) % 2 == 0 and (parity[1] + c.pad - s_) % 2 == 0]
                    todo.append((tab, prs, 1))
            for tab, trs, tr in todo:
                if len(trs) <= 16:
                    jobs.append(self._pack_job(w, scale_ptr, tab, c, trs, tr))
                else:                                       # the 7 x 7 stem's generic tables (49 taps): single launches, as before
                    big.append((w, st['scale'][off:off + c.cout], tab, c, trs, tr))
            off += c.cout
        frag, sums = [], []
        self._pack_frag_tables(record=frag, sums=sums)
        fj = []
        for src, dst, rows, k in frag:
            j = _lib.PackJob()
            j.kind, j.rows, j.k, j.src16, j.out = 1, rows, k, src.data_ptr(), dst.data_ptr()
            fj.append(j)
        st['pack_jobs'], st['n_pack'] = self._upload_jobs(jobs), len(jobs)
        st['frag_jobs'], st['n_frag'] = (self._upload_jobs(fj) if fj else None), len(fj)
        st['bias_sums'], st['big'] = sums, big
        st['_keep'] = (jobs, fj)
        self._refold_state = st
        return st

    @staticmethod
    def _pack_job(w, scale_ptr, tab, c, rs, transpose):
        j = _lib.PackJob()
        j.kind, j.n_out, j.channels, j.r, j.s, j.n_taps, j.transpose, j.rows_padded = 0, c.cout, c.cin, c.r, c.s, len(rs), transpose, tab.shape[0]
        for i, (a, b) in enumerate(rs):
            j.tap_r[i], j.tap_s[i] = a, b
        j.weight, j.out_channel_scale, j.out = w.data_ptr(), scale_ptr, tab.data_ptr()
        return j

    def _upload_jobs(self, jobs):
        torch = _lib.require_gpu()
        arr = (_lib.PackJob * len(jobs))(*jobs)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        return host.to(self.device)

    # ------------------------------------------------------------------ buffers / launches
    def _get(self, name, shape, dtype=None):
        torch = _lib.require_gpu()
        dtype = dtype or torch.bfloat16
        t = self._buf.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._buf[name] = t
        return t

    def _gemm(self, src, wgt, dst, batch, grid, src_hw, src_pix, k_per_tap, taps, n_cols, dst_hw, dst_pix,
              bias=None, res=None, mask=None, flags=0, stride=(1, 1), dst_stride=(1, 1), dst_off=(0, 0),
              tap_src_off=None, sign_out=None, pair=False):
        if pair and self.pair_gemm_kernel and len(taps) <= 16 and k_per_tap >= 32 and (k_per_tap & (k_per_tap - 1)) == 0:
            return self._gemm_pair(src, wgt, dst, batch, grid, src_hw, src_pix, k_per_tap, taps, n_cols, dst_hw, dst_pix, bias, res,
                                   mask, flags, stride, dst_stride, dst_off, sign_out)
        d = _lib.ConvDesc()
        if pair:
            # split-bf16 tensors [2][...] (hi plane, lo plane): the three products as 3x the taps, the lo planes of dst / res
            # by their element offsets (rart_conv_desc flag 32); an fp32 destination (F_OUT_F32) is a plain tensor
            assert tap_src_off is None and src.shape[0] == 2
            lo = (src[1].data_ptr() - src[0].data_ptr()) // 2
            tap_src_off = [0] * (2 * len(taps)) + [lo] * len(taps)
            taps = list(taps) * 3
            flags |= F_PAIR
            if not (flags & F_OUT_F32):
                d.dst_pair_off = (dst[1].data_ptr() - dst[0].data_ptr()) // 2
            if res is not None:
                d.res_pair_off = (res[1].data_ptr() - res[0].data_ptr()) // 2
        d.src, d.wgt, d.dst = src.data_ptr(), wgt.data_ptr(), dst.data_ptr()
        d.bias = bias.data_ptr() if bias is not None else None
        d.res = res.data_ptr() if res is not None else None
        d.mask = mask.data_ptr() if mask is not None else None
        d.sign_out = sign_out.data_ptr() if sign_out is not None else None
        d.batch, d.grid_h, d.grid_w = batch, grid[0], grid[1]
        d.src_h, d.src_w, d.src_pix_stride = src_hw[0], src_hw[1], src_pix
        d.k_per_tap, d.n_taps = k_per_tap, len(taps)
        d.sy, d.sx = stride
        for i, (dy, dx) in enumerate(taps):
            d.tap_dy[i], d.tap_dx[i] = dy, dx
            d.tap_src_off[i] = tap_src_off[i] if tap_src_off is not None else 0
        d.n_cols = n_cols
        d.dst_h, d.dst_w = dst_hw
        d.dst_sy, d.dst_sx = dst_stride
        d.dst_oy, d.dst_ox = dst_off
        d.dst_pix_stride = dst_pix
        d.flags = flags
        if self.profile is not None:
            torch = _lib.require_gpu()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()          # torch's current stream == the stream the kernel is enqueued on (stream_ptr())
            _lib.check(self.lib.rart_conv_igemm_bf16(ctypes.byref(d), _lib.stream_ptr()))
            e1.record()
            flops = 2.0 * batch * grid[0] * grid[1] * k_per_tap * len(taps) * n_cols
            self.profile.append((flops, e0, e1, 'igemm'))
            return
        _lib.check(self.lib.rart_conv_igemm_bf16(ctypes.byref(d), _lib.stream_ptr()))

    def _gemm_pair(self, src, wgt, dst, batch, grid, src_hw, src_pix, k_per_tap, taps, n_cols, dst_hw, dst_pix, bias, res, mask, flags,
                   stride, dst_stride, dst_off, sign_out):
        """One convolution / product of the reference-precision mode on rart_gemm_pair_bf16 (csrc/gemm_pair.hip, conv mode): the four
        operand planes of a K step staged once, three MFMAs per fragment pair -- round 3 listed the three products as 3 x the taps of
        rart_conv_igemm_bf16, which fetched x_hi twice and ran at a quarter of this kernel's MFMA rate.  src / dst / res: pair tensors
        [2][...]; wgt: the [rows][hi | lo | hi] table of `_Conv(split=True)` (w_hi = columns 0..K, w_lo = columns K..2K, row stride 3K);
        mask: 1-bit ReLU mask of the destination; an fp32 destination (F_OUT_F32) is a plain tensor."""
        assert src.shape[0] == 2
        k_tot = k_per_tap * len(taps)
        assert wgt.shape[1] == 3 * k_tot
        d = _lib.GemmPairDesc()
        d.a_hi, d.a_lo = src[0].data_ptr(), src[1].data_ptr()
        d.w_hi, d.w_lo = wgt.data_ptr(), wgt.data_ptr() + 2 * k_tot
        ldw, wflag = 3 * k_tot, 0
        if self.pair_w_interleaved:
            # per row and 32-deep K step: the hi slice then the lo slice (one 128-byte line per row and step instead of two half lines)
            il = self._w_il.get(wgt.data_ptr())
            if il is None:
                import torch
                rows = wgt.shape[0]
                il = torch.stack([wgt[:, :k_tot].reshape(rows, k_tot // 32, 32), wgt[:, k_tot:2 * k_tot].reshape(rows, k_tot // 32, 32)], 2)
                il = self._w_il[wgt.data_ptr()] = il.reshape(rows, 2 * k_tot).contiguous()
            d.w_hi, d.w_lo, ldw, wflag = il.data_ptr(), il.data_ptr() + 64, 2 * k_tot, 16
        d.bias = bias.data_ptr() if bias is not None else None
        if res is not None:
            d.res_hi, d.res_lo = res[0].data_ptr(), res[1].data_ptr()
        if flags & F_OUT_F32:
            d.dst_hi = dst.data_ptr()
        else:
            d.dst_hi, d.dst_lo = dst[0].data_ptr(), dst[1].data_ptr()
        d.N, d.lda, d.ldw, d.ldc, d.w_rows = n_cols, src_pix, ldw, dst_pix, wgt.shape[0]
        d.flags = (flags & (F_RELU | F_OUT_F32)) | wflag
        d.conv, d.batch, d.grid_h, d.grid_w = 1, batch, grid[0], grid[1]
        d.src_h, d.src_w, d.sy, d.sx = src_hw[0], src_hw[1], stride[0], stride[1]
        d.k_per_tap, d.n_taps = k_per_tap, len(taps)
        for i, (dy, dx) in enumerate(taps):
            d.tap_dy[i], d.tap_dx[i] = dy, dx
        d.dst_h, d.dst_w = dst_hw
        d.dst_sy, d.dst_sx = dst_stride
        d.dst_oy, d.dst_ox = dst_off
        d.mask_bits = mask.data_ptr() if mask is not None else None
        d.sign_out = sign_out.data_ptr() if sign_out is not None else None
        d.tile_m, d.tile_n = self.pair_tile          # (0, 0): the library's choice; profiling sweeps force a tile
        if self.profile is not None:
            torch = _lib.require_gpu()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(self.lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr()))
            e1.record()
            # MFMA FLOPs issued; algorithmic bytes = every operand once: source pair pixels (4 B per element), destination pair (or fp32),
            # residual pair, both weight planes, 1 bit per destination element for a mask / sign tensor
            rows = batch * grid[0] * grid[1]
            nbytes = 4.0 * batch * src_hw[0] * src_hw[1] * k_per_tap + 4.0 * rows * n_cols * (2 if res is not None else 1) + 4.0 * k_tot * n_cols \
                + rows * n_cols / 8.0 * ((mask is not None) + (sign_out is not None))
            self.profile.append((3 * 2.0 * rows * k_tot * n_cols, e0, e1, 'gemm_pair', nbytes))
            return
        _lib.check(self.lib.rart_gemm_pair_bf16(ctypes.byref(d), _lib.stream_ptr()))

    def _tail(self, src, wgt, taps, tail, dst, batch, hw, c_mid, bias_mid=None, bias_out=None, res=None, mask_mid=None, mask_out=None,
              sign_mid=None, sign_out=None, relu=False, nxt=None):
        """3x3 (c_mid -> c_mid) + point-wise step + 1x1 expansion (c_mid -> 4 c_mid) + skip pair + point-wise step of the
        reference-precision mode as ONE launch (csrc/conv_tail_pair.hip).  src / dst / res: pair tensors; wgt: the 3x3's
        [rows][hi | lo | hi] table; tail: `_pack_tail_tables`' [2][...] fragment-ordered 1x1 table.  nxt: the neighbouring block's 1x1
        reduction of the tile in the same launch -- dict(tab, dst, bias, mask, sign, relu)."""
        k_tot = 9 * c_mid
        assert src.shape[0] == 2 and wgt.shape[1] == 3 * k_tot and len(taps) == 9
        d = _lib.ConvTailDesc()
        d.a_hi, d.a_lo = src[0].data_ptr(), src[1].data_ptr()
        d.w_hi, d.w_lo = wgt.data_ptr(), wgt.data_ptr() + 2 * k_tot
        d.t_hi, d.t_lo = tail[0].data_ptr(), tail[1].data_ptr()
        d.bias_mid, d.bias_out = _lib.ptr(bias_mid), _lib.ptr(bias_out)
        d.mask_mid, d.mask_out, d.sign_mid, d.sign_out = _lib.ptr(mask_mid), _lib.ptr(mask_out), _lib.ptr(sign_mid), _lib.ptr(sign_out)
        if res is not None:
            d.res_hi, d.res_lo = res[0].data_ptr(), res[1].data_ptr()
        d.dst_hi, d.dst_lo = dst[0].data_ptr(), dst[1].data_ptr()
        d.batch, d.h, d.w, d.c_mid, d.ldw = batch, hw[0], hw[1], c_mid, 3 * k_tot
        d.relu_mid = d.relu_out = 1 if relu else 0
        for i, (dy, dx) in enumerate(taps):
            d.tap_dy[i], d.tap_dx[i] = dy, dx
        if nxt is not None:
            d.n_hi, d.n_lo = nxt['tab'][0].data_ptr(), nxt['tab'][1].data_ptr()
            d.dstn_hi, d.dstn_lo = nxt['dst'][0].data_ptr(), nxt['dst'][1].data_ptr()
            d.bias_next, d.mask_next, d.sign_next = _lib.ptr(nxt.get('bias')), _lib.ptr(nxt.get('mask')), _lib.ptr(nxt.get('sign'))
            d.relu_next = 1 if nxt.get('relu') else 0
        if self.profile is not None:
            torch = _lib.require_gpu()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(self.lib.rart_conv3x3_tail_pair(ctypes.byref(d), _lib.stream_ptr()))
            e1.record()
            px = batch * hw[0] * hw[1]
            nbytes = 4.0 * px * c_mid * (1 + 4 + (4 if res is not None else 0) + (1 if nxt is not None else 0)) \
                + 4.0 * c_mid * c_mid * (9 + 4 + (4 if nxt is not None else 0))
            self.profile.append((3 * 2.0 * px * (k_tot * c_mid + (8 if nxt is not None else 4) * c_mid * c_mid), e0, e1,
                                 'conv_tail_pair', nbytes))   # MFMA FLOPs issued, algorithmic bytes (every operand once)
            return
        _lib.check(self.lib.rart_conv3x3_tail_pair(ctypes.byref(d), _lib.stream_ptr()))

    def _prof_begin(self):
        if self.profile is None:
            return None
        torch = _lib.require_gpu()
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return e0

    def _prof_end(self, e0, flops, kind, nbytes=None):
        if e0 is None:
            return
        e1 = _lib.require_gpu().cuda.Event(enable_timing=True)
        e1.record()
        self.profile.append((flops, e0, e1, kind, nbytes))

    def _fc(self, a, w, out, m, n, k, bias=None):
        """The classifier head / its backward: rart_gemm_small_m_bf16 (one workgroup per 32 x 32 tile, waves split K) instead of the
        16-workgroup implicit-GEMM launch; `out` fp32 or bf16."""
        torch = _lib.require_gpu()
        args = (_lib.ptr(a), a.shape[-1], _lib.ptr(w), w.shape[-1], _lib.ptr(bias) if bias is not None else None, _lib.ptr(out),
                out.shape[-1], 1 if out.dtype == torch.float32 else 0, m, n, k, _lib.stream_ptr())
        if self.profile is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(self.lib.rart_gemm_small_m_bf16(*args))
            e1.record()
            self.profile.append((2.0 * m * n * k, e0, e1, 'fc_small_m'))
            return
        _lib.check(self.lib.rart_gemm_small_m_bf16(*args))

    @staticmethod
    def _fits32(n, hw, channels):
        """the fused / LDS-resident kernels address with 32-bit element offsets (their C entry points check n*H*W*C < 2^31):
        beyond that the engine falls back to the generic implicit GEMM instead of raising"""
        return n * hw[0] * hw[1] * channels < (1 << 31)

    def _halo_ok(self, c, hw, n=1):
        return (self._fits32(n, hw, c.cin) and self.halo_conv3x3 and c.r == 3 and c.s == 3 and c.stride == 1 and c.pad == 1
                and c.cin == c.cout and getattr(c, 'w_fwd_frag', None) is not None and self.lib.rart_conv3x3_halo_supported(c.cin, hw[0], hw[1]))

    def _halo(self, src, w, dst, B, hw, ch, taps, bias=None, mask=None, sign=None, relu=False):
        if self.profile is not None:
            torch = _lib.require_gpu()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            prof, self.profile = self.profile, None
            try:
                self._halo(src, w, dst, B, hw, ch, taps, bias=bias, mask=mask, sign=sign, relu=relu)
            finally:
                self.profile = prof
            e1.record()
            self.profile.append((2.0 * B * hw[0] * hw[1] * 9 * ch * ch, e0, e1, 'halo3x3'))
            return
        _lib.check(self.lib.rart_conv3x3_halo_bf16(_lib.ptr(src), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(mask), _lib.ptr(sign),
                                                   _lib.ptr(dst), B, hw[0], hw[1], ch, _cints([t[0] for t in taps]),
                                                   _cints([t[1] for t in taps]), 1 if relu else 0, _lib.stream_ptr()))

    def _bneck_ok(self, ca, cb, cc, ds, xhw, n=1):
        return (self._fits32(n, xhw, cc.cout) and self.fused_bottleneck and ds is None and cb.stride == 1 and cb.r == 3 and ca.r == 1 and cc.r == 1
                and ca.cin == cc.cout and getattr(cb, 'w_fwd_frag', None) is not None
                and self.lib.rart_bottleneck_fused_supported(cc.cout, ca.cout, xhw[0], xhw[1]))

    def _image_block_fn(self, ca, cb, cc, ds, xhw, n=1):
        """-> the C entry point of the image-resident fused kernel for this block (layer3: 14 x 14, layer2: 28 x 28) or None."""
        if not (self._fits32(n, xhw, cc.cout) and self.fused_bottleneck14 and ds is None and getattr(ca, 'w_fwd_frag', None) is not None
                and getattr(cb, 'w_fwd_frag', None) is not None and getattr(cc, 'w_fwd_frag', None) is not None):
            return None
        if self.lib.rart_bottleneck14_fused_supported(cc.cout, ca.cout, xhw[0], xhw[1]):
            return self.lib.rart_bottleneck14_fused_bf16
        if self.fused_bottleneck28 and self.lib.rart_bottleneck28_fused_supported(cc.cout, ca.cout, xhw[0], xhw[1]):
            return self.lib.rart_bottleneck28_fused_bf16
        if self.fused_bottleneck7 and self.lib.rart_bottleneck7_fused_supported(cc.cout, ca.cout, xhw[0], xhw[1]):
            return self.lib.rart_bottleneck7_fused_bf16
        return None

    def _bneck14(self, x, w1, w2, w3, b1, b2, b3, m1, m2, m3, out, B, hw, c_io, c_mid, taps, backward, fn=None):
        """One layer3 / layer2 identity Bottleneck as a single launch (csrc/bottleneck14_fused.hip, bottleneck28_fused.hip)."""
        fn = fn or self.lib.rart_bottleneck14_fused_bf16
        if self.profile is not None:
            torch = _lib.require_gpu()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            prof, self.profile = self.profile, None
            try:
                self._bneck14(x, w1, w2, w3, b1, b2, b3, m1, m2, m3, out, B, hw, c_io, c_mid, taps, backward, fn)
            finally:
                self.profile = prof
            e1.record()
            # 5th element: ALGORITHMIC HBM bytes of the launch -- x in, out, the weight tables once, the 1-bit sign tensors
            m_ = B * hw[0] * hw[1]
            by = 2 * m_ * c_io * 2 + (2 * c_io * c_mid + 9 * c_mid * c_mid) * 2 + sum(m_ * c // 8 for c, t in ((c_mid, m1), (c_mid, m2), (c_io, m3)) if t is not None)
            self.profile.append((2.0 * B * hw[0] * hw[1] * c_mid * (2 * c_io + 9 * c_mid), e0, e1,
                                 {14: 'bottleneck14', 28: 'bottleneck28', 7: 'bottleneck7'}[hw[0]], by))
            return
        _lib.check(fn(
            _lib.ptr(x), _lib.ptr(w1), _lib.ptr(w2), _lib.ptr(w3), _lib.ptr(b1), _lib.ptr(b2), _lib.ptr(b3), _lib.ptr(m1),
            _lib.ptr(m2), _lib.ptr(m3), _lib.ptr(out), B, hw[0], hw[1], c_io, c_mid, _cints([t[0] for t in taps]),
            _cints([t[1] for t in taps]), 1 if backward else 0, _lib.stream_ptr()))

    def _s2_ok(self, ca, cb, cc, ds, xhw, n=1):
        return (self.fused_bottleneck_s2 and ds is not None and getattr(ds, 's2_wd', None) is not None
                and self._fits32(n, xhw, ca.cin) and self.lib.rart_bottleneck_s2_fwd_supported(ca.cin, ca.cout, cc.cout, xhw[0], xhw[1]))

    def _bneck_s2(self, x, ca, cb, cc, ds, m1, m2, m3, out, B, xhw):
        """The stride-2 first block of layer2 / layer3, forward, as one launch (csrc/bottleneck_s2_fused.hip)."""
        if self.profile is not None:
            torch = _lib.require_gpu()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            prof, self.profile = self.profile, None
            try:
                self._bneck_s2(x, ca, cb, cc, ds, m1, m2, m3, out, B, xhw)
            finally:
                self.profile = prof
            e1.record()
            pin, pout = xhw[0] * xhw[1], xhw[0] * xhw[1] // 4
            self.profile.append((2.0 * B * (pin * ca.cin * ca.cout + pout * (9 * cb.cin * cb.cout + cc.cin * cc.cout + ds.cin * ds.cout)),
                                 e0, e1, 'bottleneck_s2'))
            return
        _lib.check(self.lib.rart_bottleneck_s2_fwd_bf16(
            _lib.ptr(x), _lib.ptr(ca.s2_w1), _lib.ptr(cb.s2_w2), _lib.ptr(cc.s2_w3), _lib.ptr(ds.s2_wd), _lib.ptr(ca.bias),
            _lib.ptr(cb.bias), _lib.ptr(ds.bias_sum), _lib.ptr(m1), _lib.ptr(m2), _lib.ptr(m3), _lib.ptr(out), B, xhw[0], xhw[1],
            ca.cin, ca.cout, cc.cout, _lib.stream_ptr()))

    def _bneck_s2_bwd(self, g, ca, cb, cc, ds, m2, m1, m0, dx, B, xhw):
        """Backward-to-input of the stride-2 first block of layer2 / layer3 as one launch (csrc/bottleneck_s2_fused.hip)."""
        if self.profile is not None:
            torch = _lib.require_gpu()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            prof, self.profile = self.profile, None
            try:
                self._bneck_s2_bwd(g, ca, cb, cc, ds, m2, m1, m0, dx, B, xhw)
            finally:
                self.profile = prof
            e1.record()
            pin, pout = xhw[0] * xhw[1], xhw[0] * xhw[1] // 4
            self.profile.append((2.0 * B * (pin * ca.cin * ca.cout + pout * (9 * cb.cin * cb.cout + cc.cin * cc.cout + ds.cin * ds.cout)),
                                 e0, e1, 'bottleneck_s2_bwd'))
            return
        _lib.check(self.lib.rart_bottleneck_s2_bwd_bf16(
            _lib.ptr(g), _lib.ptr(cc.s2_w3t), _lib.ptr(cb.s2_w2t), _lib.ptr(ca.s2_w1t), _lib.ptr(ds.s2_wdt), _lib.ptr(m2), _lib.ptr(m1),
            _lib.ptr(m0), _lib.ptr(dx), B, xhw[0], xhw[1], ca.cin, ca.cout, cc.cout, _lib.stream_ptr()))

    def _first_ok(self, ca, cb, cc, ds, xhw, n=1):
        return (self._fits32(n, xhw, cc.cout) and self.fused_bottleneck and ds is not None and getattr(ds, 'w_fwd_frag', None) is not None
                and getattr(cb, 'w_fwd_frag', None) is not None
                and self.lib.rart_bottleneck_first_supported(ca.cin, ca.cout, cc.cout, xhw[0], xhw[1]))

    def _bneck(self, x, w1, w2, w3, b1, b2, b3, m1, m2, m3, out, B, hw, c_io, c_mid, taps, backward, w4=None, c_in=None):
        """One Bottleneck as a single launch (csrc/bottleneck_fused.hip), forward or backward-to-input; w4: the projection
        shortcut's table for the layer's first block (c_in input channels)."""
        if self.profile is not None:
            torch = _lib.require_gpu()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            prof, self.profile = self.profile, None
            try:
                self._bneck(x, w1, w2, w3, b1, b2, b3, m1, m2, m3, out, B, hw, c_io, c_mid, taps, backward, w4, c_in)
            finally:
                self.profile = prof
            e1.record()
            k_all = (2 * c_io + 9 * c_mid) if w4 is None else (c_in + 9 * c_mid + c_io + c_in * c_io // c_mid)
            # 5th element: ALGORITHMIC HBM bytes of the launch -- x in (read ONCE: the residual re-read and the halo rows are
            # implementation traffic), out, the weight tables once, the 1-bit sign tensors
            m_, cin_ = B * hw[0] * hw[1], (c_io if w4 is None else c_in)
            by = m_ * (c_io + cin_) * 2 + ((c_io + cin_) * c_mid + 9 * c_mid * c_mid + (0 if w4 is None else c_in * c_io)) * 2 + \
                sum(m_ * c // 8 for c, t in ((c_mid, m1), (c_mid, m2), (c_io if not backward else cin_, m3)) if t is not None)
            self.profile.append((2.0 * B * hw[0] * hw[1] * c_mid * k_all, e0, e1, 'bottleneck', by))
            return
        if w4 is not None:
            _lib.check(self.lib.rart_bottleneck_first_bf16(
                _lib.ptr(x), _lib.ptr(w1), _lib.ptr(w2), _lib.ptr(w3), _lib.ptr(w4), _lib.ptr(b1), _lib.ptr(b2), _lib.ptr(b3),
                _lib.ptr(m1), _lib.ptr(m2), _lib.ptr(m3), _lib.ptr(out), B, hw[0], hw[1], c_in, c_mid, c_io,
                _cints([t[0] for t in taps]), _cints([t[1] for t in taps]), 1 if backward else 0, _lib.stream_ptr()))
            return
        _lib.check(self.lib.rart_bottleneck_fused_bf16(
            _lib.ptr(x), _lib.ptr(w1), _lib.ptr(w2), _lib.ptr(w3), _lib.ptr(b1), _lib.ptr(b2), _lib.ptr(b3), _lib.ptr(m1),
            _lib.ptr(m2), _lib.ptr(m3), _lib.ptr(out), B, hw[0], hw[1], c_io, c_mid, _cints([t[0] for t in taps]),
            _cints([t[1] for t in taps]), 1 if backward else 0, _lib.stream_ptr()))

    def _conv_fwd(self, c, x, xhw, out, relu, res=None, sign=None, pair=False):
        B = x.shape[1] if pair else x.shape[0]
        oh, ow = xhw[0] // c.stride, xhw[1] // c.stride
        if pair:
            return self._gemm(x, c.w_fwd, out, B, (oh, ow), xhw, c.cin, c.cin, c.fwd_taps, c.cout, (oh, ow), c.cout, bias=c.bias,
                              res=res, flags=F_RELU if relu else 0, stride=(c.stride, c.stride), sign_out=sign, pair=True)
        if res is None and self._halo_ok(c, xhw, B):
            return self._halo(x, c.w_fwd_frag, out, B, xhw, c.cin, c.fwd_taps, bias=c.bias, sign=sign, relu=relu)
        self._gemm(x, c.w_fwd, out, B, (oh, ow), xhw, c.cin, c.cin, c.fwd_taps, c.cout, (oh, ow), c.cout,
                   bias=c.bias, res=res, flags=F_RELU if relu else 0, stride=(c.stride, c.stride), sign_out=sign)

    def _conv_bwd(self, c, dz, dz_hw, dx, dx_hw, res=None, mask=None, pair=False):
        """dx = backward-to-input of conv c applied to dz (then + res, masked).  mask: the forward activation at dx's
        place (bf16) or its 1-bit sign tensor (uint8, written by the forward GEMM's sign_out)."""
        torch = _lib.require_gpu()
        B = dz.shape[1] if pair else dz.shape[0]
        fl = F_MASK_BITS if (mask is not None and mask.dtype == torch.uint8) else 0
        assert not pair or mask is None or fl, 'the split-bf16 mode reads ReLU masks as 1-bit tensors only'
        if not pair and res is None and (mask is None or fl) and self._halo_ok(c, dx_hw, B):
            return self._halo(dz, c.w_bwd_frag, dx, B, dx_hw, c.cin, c.bwd[0][1], mask=mask)
        for parity, taps, w in c.bwd:
            if parity is None:
                self._gemm(dz, w, dx, B, dx_hw, dz_hw, c.cout, c.cout, taps, c.cin, dx_hw, c.cin, res=res, mask=mask,
                           flags=fl, pair=pair)
            else:
                ph, pw = parity
                if not taps:
                    continue        # this input-parity class receives no gradient from a 1x1/2 conv
                self._gemm(dz, w, dx, B, (dx_hw[0] // 2, dx_hw[1] // 2), dz_hw, c.cout, c.cout, taps, c.cin, dx_hw,
                           c.cin, res=res, mask=mask, flags=fl, dst_stride=(2, 2), dst_off=(ph, pw), pair=pair)

    # ------------------------------------------------------------------ forward
    def _forward(self, src, src_is_u8, mean, std, keep):
        if self.precision == 'bf16x3':
            return self._forward_x3(src, src_is_u8, mean, std, keep)
        torch = _lib.require_gpu()
        lib = self.lib
        if src_is_u8:
            B, H, W = src.shape[0], src.shape[1], src.shape[2]
        else:
            B, H, W = src.shape[0], src.shape[2], src.shape[3]
        assert H % 32 == 0 and W % 32 == 0, 'input height/width must be multiples of 32'
        sp = _lib.stream_ptr()
        meanf = (ctypes.c_float * 3)(*mean)
        stdf = (ctypes.c_float * 3)(*std)
        acts = {}
        h1, w1 = H // 2, W // 2
        h2, w2 = h1 // 2, w1 // 2
        p1 = self._get('p1', (B, h2, w2, 64))
        parg = self._get('p1_argmax', (B, h2, w2, 64), torch.uint8) if keep else None
        bits = keep and self.sign_bit_masks
        xs = self._get('p1_sign', (B, h2, w2, 8), torch.uint8) if bits else None
        if self.fused_stem_fwd:
            # normalise + hi/lo split + 7x7/2 conv + bias + ReLU + 3x3/2 max pool in one persistent kernel
            y1 = None
            _lib.check(lib.rart_engine_stem_fwd_fused(_lib.ptr(src), 1 if src_is_u8 else 0, _lib.ptr(self.stem_w),
                                                      self.stem_w.shape[1], _lib.ptr(self.stem.bias), _lib.ptr(p1), _lib.ptr(parg),
                                                      _lib.ptr(xs), B, H, W, meanf, stdf, sp))
        else:
            hi = self._get('in_hi', (2, B, H + 8, W + 8, 4))
            _lib.check(lib.rart_engine_prep_input(_lib.ptr(src), 1 if src_is_u8 else 0, _lib.ptr(hi[0]), _lib.ptr(hi[1]),
                                                  B, H, W, meanf, stdf, sp))
            y1 = self._get('y1', (B, h1, w1, 64))
            lo_off = hi[1].data_ptr() - hi[0].data_ptr()
            assert lo_off % 2 == 0
            taps = [(r, 0) for r in range(7)] * 2
            offs = [0] * 7 + [lo_off // 2] * 7
            self._gemm(hi[0], self.stem_w, y1, B, (h1, w1), (H + 8, W + 8), 4, 32, taps, 64, (h1, w1), 64,
                       bias=self.stem.bias, flags=F_RELU, stride=(2, 2), tap_src_off=offs)
            _lib.check(lib.rart_engine_maxpool_keep(_lib.ptr(y1), _lib.ptr(p1), _lib.ptr(parg), _lib.ptr(xs), B, h1, w1, 64, sp))
        acts['y1'], acts['p1'], acts['p1_argmax'] = y1, p1, parg
        x, xhw = p1, (h2, w2)
        for bi, (ca, cb, cc, ds) in enumerate(self.blocks):
            ohw = (xhw[0] // cb.stride, xhw[1] // cb.stride)
            ya = yb = None       # the fused kernels keep both intermediates on chip: allocated on the unfused branch only
            yc = self._get('b%d_c' % bi, (B, ohw[0], ohw[1], cc.cout))
            sa = self._get('b%d_a_sign' % bi, (B, xhw[0], xhw[1], ca.cout // 8), torch.uint8) if bits else None
            sb = self._get('b%d_b_sign' % bi, (B, ohw[0], ohw[1], cb.cout // 8), torch.uint8) if bits else None
            sc = self._get('b%d_c_sign' % bi, (B, ohw[0], ohw[1], cc.cout // 8), torch.uint8) if bits else None
            if (bits or not keep) and self._bneck_ok(ca, cb, cc, ds, xhw, B):
                # the two 64-channel intermediates stay on chip; the backward pass gets their sign bits
                self._bneck(x, ca.w_fwd, cb.w_fwd_frag, cc.w_fwd, ca.bias, cb.bias, cc.bias, sa, sb, sc, yc, B, xhw, cc.cout,
                            ca.cout, cb.fwd_taps, False)
                acts['b%d' % bi] = (x, xhw, ya, yb, yc, ohw)
                acts['b%d_masks' % bi] = (xs, sa, sb)
                x, xhw, xs = yc, ohw, sc
                continue
            fn14 = self._image_block_fn(ca, cb, cc, ds, xhw, B) if (bits or not keep) else None
            if fn14 is not None:
                self._bneck14(x, ca.w_fwd_frag, cb.w_fwd_frag, cc.w_fwd_frag, ca.bias, cb.bias, cc.bias, sa, sb, sc, yc, B, xhw,
                              cc.cout, ca.cout, cb.fwd_taps, False, fn14)
                acts['b%d' % bi] = (x, xhw, ya, yb, yc, ohw)
                acts['b%d_masks' % bi] = (xs, sa, sb)
                x, xhw, xs = yc, ohw, sc
                continue
            if (bits or not keep) and self._first_ok(ca, cb, cc, ds, xhw, B):
                self._bneck(x, ca.w_fwd, cb.w_fwd_frag, cc.w_fwd, ca.bias, cb.bias, ds.bias_sum, sa, sb, sc, yc, B, xhw, cc.cout,
                            ca.cout, cb.fwd_taps, False, w4=ds.w_fwd_frag, c_in=ca.cin)
                acts['b%d' % bi] = (x, xhw, ya, yb, yc, ohw)
                acts['b%d_masks' % bi] = (xs, sa, sb)
                x, xhw, xs = yc, ohw, sc
                continue
            if (bits or not keep) and self._s2_ok(ca, cb, cc, ds, xhw, B):
                self._bneck_s2(x, ca, cb, cc, ds, sa, sb, sc, yc, B, xhw)
                acts['b%d' % bi] = (x, xhw, ya, yb, yc, ohw)
                acts['b%d_masks' % bi] = (xs, sa, sb)
                x, xhw, xs = yc, ohw, sc
                continue
            ya = self._get('b%d_a' % bi, (B, xhw[0], xhw[1], ca.cout))
            yb = self._get('b%d_b' % bi, (B, ohw[0], ohw[1], cb.cout))
            self._conv_fwd(ca, x, xhw, ya, True, sign=sa)
            self._conv_fwd(cb, ya, xhw, yb, True, sign=sb)
            if ds is not None:
                sk = self._get('b%d_ds' % bi, (B, ohw[0], ohw[1], cc.cout))
                self._conv_fwd(ds, x, xhw, sk, False)
            else:
                sk = x
            self._conv_fwd(cc, yb, ohw, yc, True, res=sk, sign=sc)
            # the backward pass needs the activations only for their ReLU sign: the 1-bit tensors when enabled
            acts['b%d' % bi] = (x, xhw, ya, yb, yc, ohw)
            acts['b%d_masks' % bi] = (xs, sa, sb) if bits else (x, ya, yb)
            x, xhw, xs = yc, ohw, sc
        pooled = self._get('pooled', (B, self.fc_in))
        _lib.check(lib.rart_engine_avgpool(_lib.ptr(x), _lib.ptr(pooled), B, xhw[0] * xhw[1], self.fc_in, sp))
        logits = torch.empty(B, self.n_classes, dtype=torch.float32, device=self.device)
        if self.small_m_fc:
            self._fc(pooled, self.fc_w, logits, B, self.n_classes, self.fc_in, bias=self.fc_b)
        else:
            self._gemm(pooled, self.fc_w, logits, B, (1, 1), (1, 1), self.fc_in, self.fc_in, [(0, 0)], self.n_classes,
                       (1, 1), self.n_classes, bias=self.fc_b, flags=F_OUT_F32)
        acts['last'] = (x, xhw)
        acts['in_shape'] = (B, H, W)
        return logits, acts

    # ------------------------------------------------------------------ reference-precision ("bf16x3") mode
    @staticmethod
    def _lo(t):
        """element offset of the lo plane of a pair tensor [2][...]"""
        return (t[1].data_ptr() - t[0].data_ptr()) // 2

    def _forward_x3(self, src, src_is_u8, mean, std, keep):
        """The forward of `_forward` with every tensor a hi + lo pair of bf16 planes and every contraction the three
        products hi.hi + hi.lo + lo.hi (fp32 accumulate) on the implicit-GEMM kernel; one launch per layer."""
        torch = _lib.require_gpu()
        lib = self.lib
        if src_is_u8:
            B, H, W = src.shape[0], src.shape[1], src.shape[2]
        else:
            B, H, W = src.shape[0], src.shape[2], src.shape[3]
        assert H % 32 == 0 and W % 32 == 0, 'input height/width must be multiples of 32'
        sp = _lib.stream_ptr()
        meanf = (ctypes.c_float * 3)(*mean)
        stdf = (ctypes.c_float * 3)(*std)
        acts = {}
        h1, w1 = H // 2, W // 2
        h2, w2 = h1 // 2, w1 // 2
        p1 = self._get('x3_p1', (2, B, h2, w2, 64))
        parg = self._get('p1_argmax', (B, h2, w2, 64), torch.uint8) if keep else None
        xs = self._get('p1_sign', (B, h2, w2, 8), torch.uint8) if keep else None
        if self.fused_stem_fwd:
            # stem: normalise + split + 7x7/2 conv + bias + ReLU + max pool on pairs, one persistent kernel (stem_pair.hip)
            ev = self._prof_begin()
            _lib.check(lib.rart_engine_stem_fwd_fused_pair(_lib.ptr(src), 1 if src_is_u8 else 0, _lib.ptr(self.stem_w_pair[0]),
                                                           _lib.ptr(self.stem_w_pair[1]), _lib.ptr(self.stem.bias), _lib.ptr(p1[0]),
                                                           _lib.ptr(p1[1]), _lib.ptr(parg), _lib.ptr(xs), B, H, W, meanf, stdf, sp))
            # issued: 3 products x (7 row taps x 32 = 224-deep K) x 64 channels per stem output; bytes: the image once + the pooled pair
            self._prof_end(ev, 3 * 2.0 * B * h1 * w1 * 224 * 64, 'stem_pair', B * H * W * 3 * (1 if src_is_u8 else 4) + 4.0 * B * h2 * w2 * 64)
        else:
            hi = self._get('in_hi', (2, B, H + 8, W + 8, 4))
            _lib.check(lib.rart_engine_prep_input(_lib.ptr(src), 1 if src_is_u8 else 0, _lib.ptr(hi[0]), _lib.ptr(hi[1]),
                                                  B, H, W, meanf, stdf, sp))
            y1 = self._get('x3_y1', (2, B, h1, w1, 64))
            self._gemm(hi, self.stem_w, y1, B, (h1, w1), (H + 8, W + 8), 4, 32, [(r, 0) for r in range(7)], 64, (h1, w1), 64,
                       bias=self.stem.bias, flags=F_RELU, stride=(2, 2), pair=True)
            _lib.check(lib.rart_engine_maxpool_pair(_lib.ptr(y1), self._lo(y1), _lib.ptr(p1), self._lo(p1), _lib.ptr(parg),
                                                    _lib.ptr(xs), B, h1, w1, 64, sp))
        acts['p1_argmax'] = parg
        x, xhw = p1, (h2, w2)
        pre_a = False            # this block's conv1 was already computed by the previous block's launch
        for bi, (ca, cb, cc, ds) in enumerate(self.blocks):
            ohw = (xhw[0] // cb.stride, xhw[1] // cb.stride)
            tail = (self.fused_tail_pair and cb.cout in self.fused_tail_channels and getattr(cc, 'tail_fwd', None) is not None
                    and self._fits32(B, ohw, cc.cout))
            ya = self._get('x3_b%d_a' % bi, (2, B, xhw[0], xhw[1], ca.cout))
            yb = None if tail else self._get('x3_b%d_b' % bi, (2, B, ohw[0], ohw[1], cb.cout))
            yc = self._get('x3_b%d_c' % bi, (2, B, ohw[0], ohw[1], cc.cout))
            sa = self._get('b%d_a_sign' % bi, (B, xhw[0], xhw[1], ca.cout // 8), torch.uint8) if keep else None
            sb = self._get('b%d_b_sign' % bi, (B, ohw[0], ohw[1], cb.cout // 8), torch.uint8) if keep else None
            sc = self._get('b%d_c_sign' % bi, (B, ohw[0], ohw[1], cc.cout // 8), torch.uint8) if keep else None
            if not pre_a:
                self._conv_fwd(ca, x, xhw, ya, True, sign=sa, pair=True)
            pre_a = False
            if ds is not None:
                sk = self._get('x3_b%d_ds' % bi, (2, B, ohw[0], ohw[1], cc.cout))
                self._conv_fwd(ds, x, xhw, sk, False, pair=True)
            else:
                sk = x
            if tail:         # conv2 + relu + conv3 + skip + relu in one launch: the 3x3's output never exists in HBM
                nxt = None
                na = self.blocks[bi + 1][0] if bi + 1 < len(self.blocks) else None
                if self.fused_next_pair and na is not None and getattr(na, 'next_fwd', None) is not None and cb.cout in self.fused_next_channels:
                    # ... and the next block's conv1 + relu on the output tile: that block's own read of this output disappears
                    nxt = dict(tab=na.next_fwd, bias=na.bias, relu=True, dst=self._get('x3_b%d_a' % (bi + 1), (2, B, ohw[0], ohw[1], na.cout)),
                               sign=self._get('b%d_a_sign' % (bi + 1), (B, ohw[0], ohw[1], na.cout // 8), torch.uint8) if keep else None)
                    pre_a = True
                self._tail(ya, cb.w_fwd, cb.fwd_taps, cc.tail_fwd, yc, B, ohw, cb.cout, bias_mid=cb.bias, bias_out=cc.bias, res=sk,
                           sign_mid=sb, sign_out=sc, relu=True, nxt=nxt)
            else:
                self._conv_fwd(cb, ya, xhw, yb, True, sign=sb, pair=True)
                self._conv_fwd(cc, yb, ohw, yc, True, res=sk, sign=sc, pair=True)
            acts['b%d' % bi] = (x, xhw, ya, yb, yc, ohw)
            acts['b%d_masks' % bi] = (xs, sa, sb)
            x, xhw, xs = yc, ohw, sc
        pooled = self._get('x3_pooled', (2, B, self.fc_in))
        _lib.check(lib.rart_engine_avgpool_pair(_lib.ptr(x), self._lo(x), _lib.ptr(pooled), self._lo(pooled), B,
                                                xhw[0] * xhw[1], self.fc_in, sp))
        logits = torch.empty(B, self.n_classes, dtype=torch.float32, device=self.device)
        self._gemm(pooled, self.fc_w, logits, B, (1, 1), (1, 1), self.fc_in, self.fc_in, [(0, 0)], self.n_classes,
                   (1, 1), self.n_classes, bias=self.fc_b, flags=F_OUT_F32, pair=True)
        acts['last'] = (x, xhw)
        acts['last_sign'] = xs
        acts['in_shape'] = (B, H, W)
        return logits, acts

    def _backward_x3(self, acts, dl, std):
        """d(loss)/d(x01) from the fp32 loss gradient dl [B][classes]: the backward-to-input chain on pairs."""
        torch = _lib.require_gpu()
        lib, sp = self.lib, _lib.stream_ptr()
        B, H, W = acts['in_shape']
        dlp = self._get('x3_dl', (2, B, self.fc_kpad))
        _lib.check(lib.rart_f32_to_pair_rows(_lib.ptr(dl), _lib.ptr(dlp), self._lo(dlp), B, self.n_classes, self.fc_kpad, sp))
        dpool = self._get('x3_dpool', (2, B, self.fc_in))
        self._gemm(dlp, self.fc_wd, dpool, B, (1, 1), (1, 1), self.fc_kpad, self.fc_kpad, [(0, 0)], self.fc_in, (1, 1),
                   self.fc_in, pair=True)
        xl, xlhw = acts['last']
        dz = self._get('x3_g_out_%d' % (len(self.blocks) - 1), tuple(xl.shape))
        _lib.check(lib.rart_engine_avgpool_bwd_pair(_lib.ptr(acts['last_sign']), _lib.ptr(dpool), self._lo(dpool), _lib.ptr(dz),
                                                    self._lo(dz), B, xlhw[0] * xlhw[1], self.fc_in, sp))
        pre_b = False            # this block's conv3^T was already computed by the launch of the block above
        for bi in range(len(self.blocks) - 1, -1, -1):
            ca, cb, cc, ds = self.blocks[bi]
            x, xhw, ya, yb, yc, ohw = acts['b%d' % bi]
            mx, ma, mb = acts['b%d_masks' % bi]
            dzb = self._get('x3_g_b%d' % (bi & 1), (2, B, ohw[0], ohw[1], cb.cout))
            if not pre_b:
                self._conv_bwd(cc, dz, ohw, dzb, ohw, mask=mb, pair=True)
            pre_b = False
            dx = self._get('x3_g_out_%d' % (bi - 1), tuple(x.shape))
            if (self.fused_tail_pair and cb.cin in self.fused_tail_channels and getattr(ca, 'tail_bwd', None) is not None and ma is not None
                    and mx is not None and self._fits32(B, xhw, ca.cin)):
                # conv2^T + mask + conv1^T + identity-skip gradient + mask in one launch
                nxt = None
                pc = self.blocks[bi - 1][2] if bi > 0 else None
                if self.fused_next_pair and pc is not None and getattr(pc, 'next_bwd', None) is not None and cb.cin in self.fused_next_channels:
                    # ... and the previous block's conv3^T + mask on the input-gradient tile
                    nxt = dict(tab=pc.next_bwd, mask=acts['b%d_masks' % (bi - 1)][2],
                               dst=self._get('x3_g_b%d' % ((bi - 1) & 1), (2, B, xhw[0], xhw[1], pc.cin)))
                    pre_b = nxt['mask'] is not None
                    if not pre_b:
                        nxt = None
                self._tail(dzb, cb.bwd[0][2], cb.bwd[0][1], ca.tail_bwd, dx, B, xhw, cb.cin, res=dz, mask_mid=ma, mask_out=mx, nxt=nxt)
                dz = dx
                continue
            dza = self._get('x3_g_a', tuple(ya.shape))
            self._conv_bwd(cb, dzb, ohw, dza, xhw, mask=ma, pair=True)
            if ds is None:
                self._conv_bwd(ca, dza, xhw, dx, xhw, res=dz, mask=mx, pair=True)            # identity skip
            else:
                self._conv_bwd(ca, dza, xhw, dx, xhw, mask=mx, pair=True)
                self._conv_bwd(ds, dz, ohw, dx, xhw, res=dx, mask=mx, pair=True)             # accumulate the projection skip
            dz = dx
        grad = torch.empty(B, 3, H, W, dtype=torch.float32, device=self.device)
        stdf = (ctypes.c_float * 3)(*std)
        if self.fused_stem_bwd:
            # stem: max-pool backward + ReLU mask + transposed 7x7/2 conv to the fp32 image on pairs, one kernel (stem_pair.hip)
            ev = self._prof_begin()
            _lib.check(lib.rart_engine_stem_bwd_fused_pair(_lib.ptr(dz[0]), _lib.ptr(dz[1]), _lib.ptr(acts['p1_argmax']),
                                                           _lib.ptr(self.stem_wt_pair[0]), _lib.ptr(self.stem_wt_pair[1]),
                                                           _lib.ptr(grad), B, H, W, stdf, sp))
            # issued: M = stem-output positions, K = 16 taps x 64 channels, N = 16 (4 pixel parities x 3 colours, padded); bytes: pooled
            # gradient pair + argmax codes + the fp32 image gradient
            self._prof_end(ev, 3 * 2.0 * B * (H // 2) * (W // 2) * 16 * 64 * 16, 'stem_pair', 4.0 * B * (H // 4) * (W // 4) * 64 * 1.25 + 4.0 * B * H * W * 3)
            return grad
        h1, w1 = H // 2, W // 2
        dz1 = self._get('x3_g_y1', (2, B, h1, w1, 64))
        _lib.check(lib.rart_engine_maxpool_bwd_pair(_lib.ptr(acts['p1_argmax']), _lib.ptr(dz), self._lo(dz), _lib.ptr(dz1),
                                                    self._lo(dz1), B, h1, w1, 64, sp))
        pc = self.stem_patch_cols
        patches = self._get('x3_patches', (B, h1, w1, pc), torch.float32)
        self._gemm(dz1, self.stem_wd, patches, B, (h1, w1), (h1, w1), 64, 64, [(0, 0)], pc, (h1, w1), pc, flags=F_OUT_F32,
                   pair=True)
        _lib.check(lib.rart_engine_stem_col2im_f32(_lib.ptr(patches), _lib.ptr(grad), B, H, W, pc, stdf, sp))
        return grad

    def logits(self, x01, mean, std):
        """x01: fp32 NCHW in [0,1]; mean/std: 3-tuples applied inside the stem's input kernel."""
        return self._forward(x01.detach().float().contiguous(), False, mean, std, keep=False)[0]

    def logits_from_u8(self, batch_u8, mean, std):
        """batch_u8: uint8 NHWC (the corruption kernels' output) -> logits, normalisation fused."""
        return self._forward(batch_u8, True, mean, std, keep=False)[0]

    # ------------------------------------------------------------------ forward + backward to the input
    def forward_backward(self, x01, mean, std, y, kind, y_target=None, scale=1.0):
        """-> (logits fp32, loss_indiv, d(sum_i scale*loss_i)/dx01 fp32 NCHW, pred int32)."""
        from ..noise.adv import logit_loss
        torch = _lib.require_gpu()
        lib = self.lib
        sp = _lib.stream_ptr()
        x01 = x01.detach().float().contiguous()
        logits, acts = self._forward(x01, False, mean, std, keep=True)
        self.last_acts = acts            # exposed for the parity tests (ReLU masks of this forward)
        loss, dl, pred = logit_loss(logits, y, kind, y_target, scale)
        self.last_dlogits = dl
        if self.precision == 'bf16x3':
            return logits, loss, self._backward_x3(acts, dl, std), pred
        B, H, W = acts['in_shape']
        # fc backward: dpool[B][2048] = dlogits[B][1024 padded] . Wfc
        dlb = self._get('dl_bf16', (B, self.fc_kpad))
        _lib.check(lib.rart_f32_to_bf16_rows(_lib.ptr(dl), _lib.ptr(dlb), B, self.n_classes, self.fc_kpad, sp))
        dpool = self._get('dpool', (B, self.fc_in))
        if self.small_m_fc:
            self._fc(dlb, self.fc_wd, dpool, B, self.fc_in, self.fc_kpad)
        else:
            self._gemm(dlb, self.fc_wd, dpool, B, (1, 1), (1, 1), self.fc_kpad, self.fc_kpad, [(0, 0)], self.fc_in, (1, 1),
                       self.fc_in)
        xl, xlhw = acts['last']
        dz = self._get('g_out_%d' % (len(self.blocks) - 1), tuple(xl.shape))
        _lib.check(lib.rart_engine_avgpool_bwd(_lib.ptr(xl), _lib.ptr(dpool), _lib.ptr(dz), B, xlhw[0] * xlhw[1],
                                               self.fc_in, sp))
        # blocks in reverse; dz = masked gradient at the block output (pre-ReLU)
        for bi in range(len(self.blocks) - 1, -1, -1):
            ca, cb, cc, ds = self.blocks[bi]
            x, xhw, ya, yb, yc, ohw = acts['b%d' % bi]
            mx, ma, mb = acts['b%d_masks' % bi]
            if (ma is not None and ma.dtype == torch.uint8 and mb is not None and self._bneck_ok(ca, cb, cc, ds, xhw, B)):
                dx = self._get('g_out_%d' % (bi - 1), tuple(x.shape))
                self._bneck(dz, cc.bwd[0][2], cb.w_bwd_frag, ca.bwd[0][2], None, None, None, mb, ma, mx, dx, B, xhw, cc.cout,
                            ca.cout, cb.bwd[0][1], True)
                dz = dx
                continue
            fn14 = self._image_block_fn(ca, cb, cc, ds, xhw, B) if (ma is not None and ma.dtype == torch.uint8 and mb is not None) else None
            if fn14 is not None:
                dx = self._get('g_out_%d' % (bi - 1), tuple(x.shape))
                self._bneck14(dz, cc.w_bwd_frag, cb.w_bwd_frag, ca.w_bwd_frag, None, None, None, mb, ma, mx, dx, B, xhw,
                              cc.cout, ca.cout, cb.bwd[0][1], True, fn14)
                dz = dx
                continue
            if (ma is not None and ma.dtype == torch.uint8 and mb is not None and self._first_ok(ca, cb, cc, ds, xhw, B)):
                dx = self._get('g_out_%d' % (bi - 1), tuple(x.shape))
                self._bneck(dz, cc.bwd[0][2], cb.w_bwd_frag, ds.bwd[0][2], None, None, None, mb, ma, mx, dx, B, xhw, cc.cout,
                            ca.cout, cb.bwd[0][1], True, w4=ca.bwd[0][2], c_in=ca.cin)
                dz = dx
                continue
            if (self.fused_bottleneck_s2_bwd and ma is not None and ma.dtype == torch.uint8 and mb is not None
                    and getattr(cb, 's2_w2t', None) is not None and self._s2_ok(ca, cb, cc, ds, xhw, B)):
                dx = self._get('g_out_%d' % (bi - 1), tuple(x.shape))
                self._bneck_s2_bwd(dz, ca, cb, cc, ds, mb, ma, mx, dx, B, xhw)
                dz = dx
                continue
            # (shapes from the layer geometry: a fused forward never materialises ya / yb)
            dzb = self._get('g_b', (B, ohw[0], ohw[1], cb.cout))
            self._conv_bwd(cc, dz, ohw, dzb, ohw, mask=mb)
            dza = self._get('g_a', (B, xhw[0], xhw[1], ca.cout))
            self._conv_bwd(cb, dzb, ohw, dza, xhw, mask=ma)
            dx = self._get('g_out_%d' % (bi - 1), tuple(x.shape))
            if ds is None:
                self._conv_bwd(ca, dza, xhw, dx, xhw, res=dz, mask=mx)            # identity skip
            else:
                self._conv_bwd(ca, dza, xhw, dx, xhw, mask=mx)
                self._conv_bwd(ds, dz, ohw, dx, xhw, res=dx, mask=mx)             # accumulate the projection skip
            dz = dx
        grad = torch.empty(B, 3, H, W, dtype=torch.float32, device=self.device)
        stdf = (ctypes.c_float * 3)(*std)
        if self.fused_stem_bwd:
            # stem: max-pool backward + ReLU mask + transposed 7x7/2 conv to the fp32 image, one kernel
            _lib.check(lib.rart_engine_stem_bwd_fused(_lib.ptr(dz), _lib.ptr(acts['p1_argmax']), _lib.ptr(self.stem_wt),
                                                      _lib.ptr(grad), B, H, W, stdf, sp))
            return logits, loss, grad, pred
        # cross-check path: max-pool backward (+ReLU mask of y1), patches GEMM, col2im to the fp32 image
        y1 = acts['y1']
        assert y1 is not None, 'the unfused stem backward needs the stem output: set fused_stem_fwd = False as well'
        h1, w1 = H // 2, W // 2
        dz1 = self._get('g_y1', tuple(y1.shape))
        _lib.check(lib.rart_engine_maxpool_bwd(_lib.ptr(y1), _lib.ptr(acts['p1_argmax']), _lib.ptr(dz), _lib.ptr(dz1),
                                               B, h1, w1, 64, sp))
        pc = self.stem_patch_cols
        patches = self._get('patches', (B, h1, w1, pc))
        self._gemm(dz1, self.stem_wd, patches, B, (h1, w1), (h1, w1), 64, 64, [(0, 0)], pc, (h1, w1), pc)
        _lib.check(lib.rart_engine_stem_col2im(_lib.ptr(patches), _lib.ptr(grad), B, H, W, pc, stdf, sp))
        return logits, loss, grad, pred


def make_engine(torch_model, device='cuda', precision='bf16'):
    """HIP engine for a model from robustart_amd.model.get_model: ResNet-50 or ViT-B/16, each with forward and
    backward-to-input (`forward_backward`) on the hand-written kernels.  precision 'bf16x3' / 'fp32x': the
    reference-precision mode (both architectures: split-bf16 pairs, three MFMA products per contraction)."""
    from .resnet_torch import ResNet
    from .vit_torch import VisionTransformer
    if isinstance(torch_model, ResNet):
        return ResNet50Engine(torch_model, device, precision)
    if isinstance(torch_model, VisionTransformer):
        from .vit_engine import ViTEngine
        return ViTEngine(torch_model, device, precision)
    raise NotImplementedError('no HIP engine for %s (ResNet-50 / ViT-B/16 only)' % type(torch_model).__name__)


class EngineModel:
    """Callable wrapper that carries a ResNet50Engine through the reference's `model` / `f_model` keys.

    takes_normalized=True  -> drop-in for the reference's `model` (input already normalised; used by
                               mim_linf / autoattack_linf, which apply ImageNet mean/std themselves);
    takes_normalized=False -> drop-in for a foolbox PyTorchModel(model, bounds=(0,1), preprocessing=
                               dict(mean, std, axis=-3)) -- the `f_model` of pgd_linf / pgd_l2 / fgsm."""

    def __init__(self, torch_model, takes_normalized, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225),
                 engine=None, device='cuda', precision='bf16'):
        self.rart_engine = engine or make_engine(torch_model, device, precision)
        self.takes_normalized = takes_normalized
        self.rart_mean_std = (tuple(mean), tuple(std))
        self.rart_torch_model = torch_model          # kept so a reference-precision engine can be folded from the same weights
        self._rart_ref_engine = None

    def rart_reference_engine(self):
        """The reference-precision ('fp32x') engine of the same module, built once and cached; the engine itself when it
        already runs at that precision; None when this wrapper was given a bare engine (no module to fold from).
        FAB needs it: its boundary projections work on the logit DIFFERENCE near zero, where bf16 storage dominates
        (fab_pt.py:102-117; 34 % vs 5 % robust on the fitted network)."""
        if self.rart_engine.precision != 'bf16':
            return self.rart_engine
        if self._rart_ref_engine is None and self.rart_torch_model is not None:
            self._rart_ref_engine = make_engine(self.rart_torch_model, self.rart_engine.device, 'fp32x')
        return self._rart_ref_engine

    def rart_refold(self, torch_model=None):
        """adversarial training: refresh the bf16 engine from the live weights and drop the cached reference engine"""
        self.rart_engine.refold(torch_model or self.rart_torch_model)
        self._rart_ref_engine = None

    def __call__(self, x):
        if self.takes_normalized:
            return self.rart_engine.logits(x, (0., 0., 0.), (1., 1., 1.))
        return self.rart_engine.logits(x, *self.rart_mean_std)
