"""Train-mode ResNet-50 on the HIP kernels: forward with batch statistics, backward to every parameter.

The training step of the reference's solver (RobustART/train/__init__.py:1 -> absent `prototype` cls_solver;
loop shape: cifar10/code/train.py:96-127; config exprs/nips_benchmark/pgd_adv_train/resnet50/config.yaml) for
`resnet50_official`: conv -> BatchNorm(batch statistics) -> ReLU bottlenecks, label-smoothed CE, gradients of all
25.6 M parameters.  PyTorch here is plumbing only (buffers, parameter storage); every FLOP and every pass over
an activation is a HIP kernel:

  contraction (forward conv, backward-to-input, weight gradient, fc)   rart_conv_igemm_bf16
  batch statistics / normalise+ReLU+residual / BatchNorm backward        rart_bn_train_forward_bf16 / _backward_bf16
  K-contiguous operands of the weight-gradient GEMM                      rart_transpose_gather_bf16
  split-K partial sums -> torch weight layout                            rart_wgrad_reduce_f32
  fp32 master weights -> bf16 igemm tables (after every optimizer step)  rart_pack_conv_weight_bf16

Gradients are written straight into the parameters' `.grad` tensors (views of the flat gradient arena,
train/arena.py); `on_grad_ready(param)` lets the arena launch a bucket's all-reduce as soon as its last gradient
exists, so the exchange overlaps the rest of the backward pass.
"""
import ctypes
import os as _os

from .. import _lib
from .engine import F_OUT_F32, _rows_mult


def _ints(vals):
    return (ctypes.c_int * max(len(vals), 1))(*vals)


class _TConv:
    """One conv (+ its BatchNorm) in training mode: packed tables, saved activations, geometry."""

    def __init__(self, conv, bn, device, torch):
        self.conv, self.bn = conv, bn
        self.cout, self.cin, self.r, self.s = conv.weight.shape
        self.stride, self.pad = conv.stride[0], conv.padding[0]
        self.fwd_taps = [(r - self.pad, s - self.pad) for r in range(self.r) for s in range(self.s)]
        self.all_rs = [(r, s) for r in range(self.r) for s in range(self.s)]
        bf = torch.bfloat16
        self.w_fwd = torch.zeros((self.cout + _rows_mult(self.cout) - 1) // _rows_mult(self.cout) * _rows_mult(self.cout),
                                 self.r * self.s * self.cin, dtype=bf, device=device)
        rm = _rows_mult(self.cin)
        rows_b = (self.cin + rm - 1) // rm * rm
        self.bwd = []        # (parity or None, taps [(dy,dx)], rs list, table)
        if self.stride == 1:
            taps = [(self.pad - r, self.pad - s) for r, s in self.all_rs]
            self.bwd.append((None, taps, self.all_rs, torch.zeros(rows_b, len(taps) * self.cout, dtype=bf, device=device)))
        else:
            for ph in range(2):
                for pw in range(2):
                    rs = [(r, s) for r, s in self.all_rs if (ph + self.pad - r) % 2 == 0 and (pw + self.pad - s) % 2 == 0]
                    taps = [((ph + self.pad - r) // 2, (pw + self.pad - s) // 2) for r, s in rs]
                    tab = torch.zeros(rows_b, len(taps) * self.cout, dtype=bf, device=device) if rs else None
                    self.bwd.append(((ph, pw), taps, rs, tab))
        if bn is not None:
            self.mean = torch.empty(self.cout, dtype=torch.float32, device=device)
            self.invstd = torch.empty(self.cout, dtype=torch.float32, device=device)
            self.scale_shift = torch.empty(2, self.cout, dtype=torch.float32, device=device)

    def repack(self, lib, sp):
        w = self.conv.weight
        assert w.dtype.is_floating_point and w.is_contiguous()
        wf = w.detach().float() if str(w.dtype) != 'torch.float32' else w.detach()
        rs = self.all_rs
        _lib.check(lib.rart_pack_conv_weight_bf16(wf.data_ptr(), None, self.w_fwd.data_ptr(), self.cout, self.cin, self.r, self.s,
                                                  len(rs), _ints([a for a, _ in rs]), _ints([b for _, b in rs]), 0,
                                                  self.w_fwd.shape[0], sp))
        for parity, taps, prs, tab in self.bwd:
            if tab is None:
                continue
            _lib.check(lib.rart_pack_conv_weight_bf16(wf.data_ptr(), None, tab.data_ptr(), self.cout, self.cin, self.r, self.s,
                                                      len(prs), _ints([a for a, _ in prs]), _ints([b for _, b in prs]), 1,
                                                      tab.shape[0], sp))


class ResNet50TrainEngine:
    def __init__(self, model, device='cuda', on_grad_ready=None, bn_momentum=None):
        torch = _lib.require_gpu()
        self.torch = torch
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.model = model
        self.on_grad_ready = on_grad_ready or (lambda p: None)
        m, dev = model, self.device
        self.stem = _TConv(m.conv1, m.bn1, dev, torch)
        self.stem_w = torch.zeros(64, 2 * 7 * 32, dtype=torch.bfloat16, device=dev)
        self.blocks = []
        for layer in (m.layer1, m.layer2, m.layer3, m.layer4):
            for blk in layer:
                ds = _TConv(blk.downsample[0], blk.downsample[1], dev, torch) if blk.downsample is not None else None
                self.blocks.append((_TConv(blk.conv1, blk.bn1, dev, torch), _TConv(blk.conv2, blk.bn2, dev, torch),
                                    _TConv(blk.conv3, blk.bn3, dev, torch), ds))
        self.n_classes, self.fc_in = m.fc.weight.shape
        self.fc_kpad = (self.n_classes + 127) // 128 * 128
        self.fc_w = torch.zeros(self.fc_kpad, self.fc_in, dtype=torch.bfloat16, device=dev)      # [1024][2048]
        self.fc_wd = torch.zeros(self.fc_in, self.fc_kpad, dtype=torch.bfloat16, device=dev)     # [2048][1024]
        self._buf = {}
        _fl = dict(kv.split('=', 1) for kv in _os.environ.get('RART_TRAIN_FLAGS', '').split(',') if '=' in kv)   # A/B switches for profiling
        # K splits of a weight-gradient launch: ~1 024 workgroups in all (measured at B = 256: 512 -> 58.1, 1 024 -> 55.1-55.8, 2 048 -> 56.3,
        # 4 096 -> 57.5 ms per adv_train step: more splits fill the CUs, every split writes and re-reads an fp32 copy of the weight tensor);
        # up to 1 024 splits of >= 256 positions each (a cap of 256 left layer1's one- and two-tile 1x1 layers at 256-512 workgroups: +0.6 ms)
        self.wgrad_target_wgs, self.wgrad_min_chunk = 1024, 256
        self._nbt_pending = None
        self.masked_skip = _fl.get('skip', '1') == '1'        # False: the BatchNorm backward writes the masked skip gradient as a tensor (cross-check)
        self.conv_bn_stats = _fl.get('stats', '1') == '1'      # False: every BatchNorm takes its own statistics pass over the conv output (rounds 1-3; cross-check)
        self.bit_masks = _fl.get('bits', '1') == '1'          # False: the BatchNorm backward reads the bf16 activation for its ReLU mask (rounds 1-3; cross-check)
        self._ysign = {}
        self.direct_wgrad = _fl.get('direct', '1') == '1'       # False: transpose_gather (dz^T, im2col^T) + implicit GEMM on the copies (rounds 1-3; cross-check)
        self.repack()

    # ------------------------------------------------------------------ tables
    def repack(self):
        """fp32 master weights -> bf16 igemm tables; call after every optimizer step."""
        torch, sp = self.torch, _lib.stream_ptr()
        convs = [c for blk in self.blocks for c in blk if c is not None]
        key = tuple(c.conv.weight.data_ptr() for c in convs)
        if all(str(c.conv.weight.dtype) == 'torch.float32' and c.conv.weight.is_contiguous() and len(c.all_rs) <= 16 for c in convs):
            # (k_pack_jobs indexes a master weight as dense [N][C][R][S]: a non-contiguous / channels_last one takes the per-conv path)
            # every table of every convolution as one job of ONE launch (rart_pack_jobs_bf16; ~110 launches of 3-10 us kernels before):
            # the job list is built once -- master weights (views into the optimizer's arena) and tables are persistent
            if getattr(self, '_pack_key', None) != key:
                jobs = []
                for c in convs:
                    for tab, rs, tr in [(c.w_fwd, c.all_rs, 0)] + [(tab, prs, 1) for _, _, prs, tab in c.bwd if tab is not None]:
                        j = _lib.PackJob()
                        j.kind, j.n_out, j.channels, j.r, j.s, j.n_taps, j.transpose, j.rows_padded = 0, c.cout, c.cin, c.r, c.s, len(rs), tr, tab.shape[0]
                        for i, (a, b) in enumerate(rs):
                            j.tap_r[i], j.tap_s[i] = a, b
                        j.weight, j.out_channel_scale, j.out = c.conv.weight.data_ptr(), None, tab.data_ptr()
                        jobs.append(j)
                arr = (_lib.PackJob * len(jobs))(*jobs)
                self._pack_jobs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
                self._pack_n, self._pack_key = len(jobs), key
            _lib.check(self.lib.rart_pack_jobs_bf16(_lib.ptr(self._pack_jobs), self._pack_n, 48, sp))
        else:
            for c in convs:
                c.repack(self.lib, sp)
        # stem forward table: a "tap" = one filter row of 8 px x 4 ch on the padded hi/lo planes (engine.py)
        wb = self.model.conv1.weight.detach().float()
        wrow = torch.zeros(64, 7, 8, 4, device=self.device)
        wrow[:, :, :7, :3] = wb.permute(0, 2, 3, 1)
        wrow = wrow.reshape(64, 224).to(torch.bfloat16)
        self.stem_w[:, :224] = wrow
        self.stem_w[:, 224:] = wrow
        one = _ints([0])
        w = self.model.fc.weight.detach()
        _lib.check(self.lib.rart_pack_conv_weight_bf16(w.data_ptr(), None, self.fc_w.data_ptr(), self.n_classes, self.fc_in, 1, 1,
                                                       1, one, one, 0, self.fc_kpad, sp))
        # backward-to-input of the classifier: rows = features, K = classes padded to fc_kpad (row stride fc_kpad)
        self.fc_wd[:, :self.n_classes] = w.t().to(torch.bfloat16)

    # ------------------------------------------------------------------ helpers
    def _get(self, name, shape, dtype=None, zero=False):
        torch = self.torch
        dtype = dtype or torch.bfloat16
        key = (name, tuple(shape), dtype)      # gradient buffers are reused across stages with different shapes
        t = self._buf.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            self._buf[key] = t
        return t

    def _scratch(self, name, nbytes):
        t = self._buf.get(name)
        if t is None or t.numel() < nbytes:
            t = self.torch.empty(int(nbytes), dtype=self.torch.uint8, device=self.device)
            self._buf[name] = t
        return t

    def _gemm(self, src, wgt, dst, batch, grid, src_hw, src_pix, k_per_tap, taps, n_cols, dst_hw, dst_pix, res=None,
              flags=0, stride=(1, 1), dst_stride=(1, 1), dst_off=(0, 0), tap_src_off=None, bias=None, batched=None, stats_out=None,
              mask_bits=None):
        d = _lib.ConvDesc()
        d.bn_stats_out = stats_out.data_ptr() if stats_out is not None else None
        d.src, d.wgt, d.dst = src.data_ptr(), wgt.data_ptr(), dst.data_ptr()
        d.bias = bias.data_ptr() if bias is not None else None
        d.res = res.data_ptr() if res is not None else None
        d.mask = mask_bits.data_ptr() if mask_bits is not None else None
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
        if batched is not None:
            d.n_batched, d.z_inner = batched['n'], batched['n']
            d.src_z_outer, d.src_z_inner = 0, batched['src']
            d.wgt_z_outer, d.wgt_z_inner = 0, batched['wgt']
            d.dst_z_outer, d.dst_z_inner = 0, batched['dst']
            d.wgt_row_stride = batched['wgt_row_stride']
        _lib.check(self.lib.rart_conv_igemm_bf16(ctypes.byref(d), _lib.stream_ptr()))

    def _conv_fwd(self, c, x, xhw, out, stats_name='bn_stats'):
        """-> (partial statistics buffer, row tiles) of the output for the BatchNorm that follows (the igemm's epilogue sums the
        columns of its 128-row tiles), or None with `conv_bn_stats` off (the BatchNorm then takes its own pass over the output)."""
        B = x.shape[0]
        oh, ow = xhw[0] // c.stride, xhw[1] // c.stride
        stats = None
        if self.conv_bn_stats:
            tiles = (B * oh * ow + 127) // 128
            stats = (self._scratch(stats_name, tiles * 2 * c.cout * 4), tiles)
        self._gemm(x, c.w_fwd, out, B, (oh, ow), xhw, c.cin, c.cin, c.fwd_taps, c.cout, (oh, ow), c.cout,
                   stride=(c.stride, c.stride), stats_out=stats[0] if stats else None)
        return stats

    def _conv_dgrad(self, c, dz, dz_hw, dx, dx_hw, res=None, res_mask_bits=None):
        """dx = conv^T(dz) [+ res]; res_mask_bits: the residual is masked by that 1-bit tensor in the epilogue (dx = conv^T(dz) + res . [bit]):
        the skip gradient d_out . [out > 0] without a tensor of its own."""
        B = dz.shape[0]
        for parity, taps, rs, w in c.bwd:
            if parity is None:
                self._gemm(dz, w, dx, B, dx_hw, dz_hw, c.cout, c.cout, taps, c.cin, dx_hw, c.cin, res=res, mask_bits=res_mask_bits,
                           flags=(16 | 128) if res_mask_bits is not None else 0)
            else:
                ph, pw = parity
                if not taps:
                    continue
                self._gemm(dz, w, dx, B, (dx_hw[0] // 2, dx_hw[1] // 2), dz_hw, c.cout, c.cout, taps, c.cin, dx_hw,
                           c.cin, res=res, dst_stride=(2, 2), dst_off=(ph, pw))

    def _bn_fwd(self, c, z, y, rows, relu, res=None, stats=None):
        lib, bn = self.lib, c.bn
        need = lib.rart_bn_workspace_bytes(rows, c.cout)
        ws = self._scratch('bn_ws', need)
        mom = bn.momentum if bn.momentum is not None else 0.1
        sign = None
        if relu and self.bit_masks:      # 1 bit per element of (y > 0): what the backward reads instead of y (1/16 of its bytes)
            sign = self._get('sgn_%x' % id(c), (rows, c.cout // 8), self.torch.uint8)      # one per conv + BatchNorm (stable name: no growth when batch sizes alternate)
            self._ysign[y.data_ptr()] = sign
        _lib.check(lib.rart_bn_train_forward_bf16(
            z.data_ptr(), res.data_ptr() if res is not None else None, y.data_ptr(), _lib.ptr(sign), rows, c.cout, bn.weight.data_ptr(),
            bn.bias.data_ptr(), bn.running_mean.data_ptr() if bn.track_running_stats else None,
            bn.running_var.data_ptr() if bn.track_running_stats else None, mom, bn.eps, 1 if relu else 0,
            c.mean.data_ptr(), c.invstd.data_ptr(), c.scale_shift.data_ptr(), stats[0].data_ptr() if stats else None,
            stats[1] if stats else 0, ws.data_ptr(), need, _lib.stream_ptr()))
        if bn.track_running_stats and bn.num_batches_tracked is not None:
            if self._nbt_pending is not None:
                self._nbt_pending.append(bn.num_batches_tracked)      # one multi-tensor add at the end of forward() instead of 53 launches
            else:
                bn.num_batches_tracked += 1

    def _bn_bwd(self, c, dy, ymask, z, dz, rows, g_out=None):
        lib, bn = self.lib, c.bn
        need = lib.rart_bn_workspace_bytes(rows, c.cout)
        ws = self._scratch('bn_ws', need)
        coef = self._get('bn_coef', (3, 2048), self.torch.float32)
        bits = self._ysign.get(ymask.data_ptr()) if (ymask is not None and self.bit_masks) else None
        _lib.check(lib.rart_bn_train_backward_bf16(
            dy.data_ptr(),
            bits.data_ptr() if bits is not None else (ymask.data_ptr() if ymask is not None else None), 1 if bits is not None else 0,
            z.data_ptr(), dz.data_ptr(),
            g_out.data_ptr() if g_out is not None else None, rows, c.cout, bn.weight.data_ptr(), c.mean.data_ptr(),
            c.invstd.data_ptr(), bn.weight.grad.data_ptr(), bn.bias.grad.data_ptr(), 0, coef.data_ptr(), ws.data_ptr(),
            need, _lib.stream_ptr()))
        self.on_grad_ready(bn.weight)
        self.on_grad_ready(bn.bias)

    def _wgrad(self, dz, n_out, n_pad_cols, x, x_hw, x_c, grid_hw, taps, stride, grad, c_valid=None):
        """grad[n_out][c][taps] = sum_m dz[m][n] * x[pixel(m) + tap][c] as a split-K GEMM on the igemm kernel.
        dz: bf16 [B, gh, gw, n_pad_cols] (columns >= n_out are zero); x: bf16 [B, ih, iw, x_c]."""
        torch, lib, sp = self.torch, self.lib, _lib.stream_ptr()
        B = dz.shape[0]
        gh, gw = grid_hw
        M = B * gh * gw
        kp = len(taps) * x_c                                   # rows of the transposed im2col matrix
        if self.direct_wgrad and (c_valid is None or x_c == 4) and lib.rart_wgrad_direct_supported(x_c, n_pad_cols, len(taps)):
            # straight from the NHWC activations (csrc/wgrad_direct.hip): no transposed copies, no materialised im2col
            tmr = 128                # the library's tile height (csrc/wgrad_direct.hip: 256-row tiles measured slower)
            if x_c == 4:             # the stem's padded hi plane: 32 taps x 4 channels per tile
                row_tiles = (len(taps) + 31) // 32
            else:
                row_tiles = len(taps) * (x_c // tmr) if x_c >= tmr else (len(taps) + tmr // x_c - 1) // (tmr // x_c)
            tiles = row_tiles * (n_pad_cols // (128 if n_pad_cols % 128 == 0 else 64))
            splits = max(1, min(self.wgrad_target_wgs // max(tiles, 1), M // self.wgrad_min_chunk if M >= 2 * self.wgrad_min_chunk else 1,
                                1024))
            chunk = ((M + splits - 1) // splits + 31) // 32 * 32
            splits = (M + chunk - 1) // chunk
            part = self._scratch('wg_part', splits * kp * n_pad_cols * 4)
            _lib.check(lib.rart_wgrad_direct_bf16(x.data_ptr(), dz.data_ptr(), part.data_ptr(), B, x_hw[0], x_hw[1], x_c, gh, gw, n_pad_cols,
                                                  stride, stride, len(taps), _ints([t[0] for t in taps]), _ints([t[1] for t in taps]),
                                                  splits, chunk, n_pad_cols, sp))
            _lib.check(lib.rart_wgrad_reduce_f32(part.data_ptr(), splits, len(taps), c_valid if c_valid is not None else x_c, x_c, n_out,
                                                 n_pad_cols, grad.data_ptr(), 0, sp))
            return
        bn_tile = 128 if n_pad_cols > 64 else 64
        tiles = ((kp + 127) // 128) * ((n_pad_cols + bn_tile - 1) // bn_tile)
        splits = max(1, min(1024 // max(tiles, 1), M // 512 if M >= 1024 else 1, 256))
        chunk = ((M + splits - 1) // splits + 63) // 64 * 64
        m_pad = chunk * splits
        n_rows = (n_pad_cols + bn_tile - 1) // bn_tile * bn_tile
        # both operands are stored as one compact slab per K split: [splits][rows][chunk]
        dzt = self._scratch('wg_dzT', n_rows * m_pad * 2)
        colt = self._scratch('wg_colT', kp * m_pad * 2)
        zero = _ints([0])
        if n_rows > n_pad_cols:
            dzt[:n_rows * m_pad * 2].zero_()                     # tile-padding rows of every slab stay zero
        _lib.check(lib.rart_transpose_gather_bf16(dz.data_ptr(), dzt.data_ptr(), B, gh, gw, n_pad_cols, gh, gw, 1, 1, 1,
                                                  zero, zero, m_pad, chunk, n_rows, sp))
        _lib.check(lib.rart_transpose_gather_bf16(x.data_ptr(), colt.data_ptr(), B, x_hw[0], x_hw[1], x_c, gh, gw,
                                                  stride, stride, len(taps), _ints([t[0] for t in taps]),
                                                  _ints([t[1] for t in taps]), m_pad, chunk, kp, sp))
        ld_n = (n_pad_cols + 7) // 8 * 8
        part = self._scratch('wg_part', splits * kp * ld_n * 4)
        self._gemm(colt, dzt, part, 1, (1, kp), (1, kp), chunk, chunk, [(0, 0)], ld_n, (1, kp), ld_n, flags=F_OUT_F32,
                   batched={'n': splits, 'src': kp * chunk, 'wgt': n_rows * chunk, 'dst': kp * ld_n,
                            'wgt_row_stride': chunk})
        cv = c_valid if c_valid is not None else x_c
        _lib.check(lib.rart_wgrad_reduce_f32(part.data_ptr(), splits, len(taps), cv, x_c, n_out, ld_n, grad.data_ptr(), 0,
                                             sp))

    def _conv_wgrad(self, c, dz, dz_hw, x, x_hw):
        self._wgrad(dz, c.cout, c.cout, x, x_hw, c.cin, dz_hw, c.fwd_taps, c.stride, c.conv.weight.grad)
        self.on_grad_ready(c.conv.weight)

    # ------------------------------------------------------------------ forward
    def forward(self, src, src_is_u8, mean, std):
        torch, lib, sp = self.torch, self.lib, _lib.stream_ptr()
        if src_is_u8:
            B, H, W = src.shape[0], src.shape[1], src.shape[2]
        else:
            src = src.detach().float().contiguous()
            B, H, W = src.shape[0], src.shape[2], src.shape[3]
        assert H % 32 == 0 and W % 32 == 0
        self._nbt_pending = []
        self._ysign = {}               # activation pointer -> its sign tensor, for this forward's backward only
        hi = self._get('in_hi', (2, B, H + 8, W + 8, 4))
        _lib.check(lib.rart_engine_prep_input(_lib.ptr(src), 1 if src_is_u8 else 0, _lib.ptr(hi[0]), _lib.ptr(hi[1]), B, H, W,
                                              (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std), sp))
        h1, w1 = H // 2, W // 2
        z1 = self._get('z1', (B, h1, w1, 64))
        y1 = self._get('y1', (B, h1, w1, 64))
        lo_off = (hi[1].data_ptr() - hi[0].data_ptr()) // 2
        st = None
        if self.conv_bn_stats:
            tiles = (B * h1 * w1 + 127) // 128
            st = (self._scratch('bn_stats', tiles * 2 * 64 * 4), tiles)
        self._gemm(hi[0], self.stem_w, z1, B, (h1, w1), (H + 8, W + 8), 4, 32, [(r, 0) for r in range(7)] * 2, 64, (h1, w1),
                   64, stride=(2, 2), tap_src_off=[0] * 7 + [lo_off] * 7, stats_out=st[0] if st else None)
        self._bn_fwd(self.stem, z1, y1, B * h1 * w1, True, stats=st)
        h2, w2 = h1 // 2, w1 // 2
        p1 = self._get('p1', (B, h2, w2, 64))
        parg = self._get('p1_argmax', (B, h2, w2, 64), torch.uint8)
        _lib.check(lib.rart_engine_maxpool(_lib.ptr(y1), _lib.ptr(p1), _lib.ptr(parg), B, h1, w1, 64, sp))
        acts = {'hi': hi, 'z1': z1, 'y1': y1, 'p1': p1, 'parg': parg, 'in_shape': (B, H, W)}
        x, xhw = p1, (h2, w2)
        for bi, (ca, cb, cc, ds) in enumerate(self.blocks):
            ohw = (xhw[0] // cb.stride, xhw[1] // cb.stride)
            za = self._get('b%d_za' % bi, (B, xhw[0], xhw[1], ca.cout))
            ya = self._get('b%d_ya' % bi, (B, xhw[0], xhw[1], ca.cout))
            zb = self._get('b%d_zb' % bi, (B, ohw[0], ohw[1], cb.cout))
            yb = self._get('b%d_yb' % bi, (B, ohw[0], ohw[1], cb.cout))
            zc = self._get('b%d_zc' % bi, (B, ohw[0], ohw[1], cc.cout))
            out = self._get('b%d_out' % bi, (B, ohw[0], ohw[1], cc.cout))
            st = self._conv_fwd(ca, x, xhw, za)
            self._bn_fwd(ca, za, ya, B * xhw[0] * xhw[1], True, stats=st)
            st = self._conv_fwd(cb, ya, xhw, zb)
            rows_o = B * ohw[0] * ohw[1]
            self._bn_fwd(cb, zb, yb, rows_o, True, stats=st)
            st_c = self._conv_fwd(cc, yb, ohw, zc)
            zd = None
            if ds is not None:
                zd = self._get('b%d_zd' % bi, (B, ohw[0], ohw[1], cc.cout))
                sk = self._get('b%d_sk' % bi, (B, ohw[0], ohw[1], cc.cout))
                st = self._conv_fwd(ds, x, xhw, zd, stats_name='bn_stats_ds')
                self._bn_fwd(ds, zd, sk, rows_o, False, stats=st)
            else:
                sk = x
            self._bn_fwd(cc, zc, out, rows_o, True, res=sk, stats=st_c)
            acts['b%d' % bi] = (x, xhw, za, ya, zb, yb, zc, zd, out, ohw)
            x, xhw = out, ohw
        pooled = self._get('pooled', (B, self.fc_in))
        _lib.check(lib.rart_engine_avgpool(_lib.ptr(x), _lib.ptr(pooled), B, xhw[0] * xhw[1], self.fc_in, sp))
        logits = torch.empty(B, self.n_classes, dtype=torch.float32, device=self.device)
        self._gemm(pooled, self.fc_w, logits, B, (1, 1), (1, 1), self.fc_in, self.fc_in, [(0, 0)], self.n_classes, (1, 1),
                   self.n_classes, bias=self.model.fc.bias.detach(), flags=F_OUT_F32)
        acts['last'], acts['pooled'] = (x, xhw), pooled
        self.acts = acts
        pending, self._nbt_pending = self._nbt_pending, None
        if pending:
            torch._foreach_add_(pending, 1)
        return logits

    # ------------------------------------------------------------------ backward to every parameter
    def backward(self, dlogits):
        """dlogits: fp32 [B][classes] = d(loss)/dlogits.  Fills .grad of every parameter (overwrites)."""
        torch, lib, sp, acts = self.torch, self.lib, _lib.stream_ptr(), self.acts
        B, H, W = acts['in_shape']
        fc = self.model.fc
        dl = dlogits.detach().float().contiguous()
        fc.bias.grad.copy_(dl.sum(0))
        self.on_grad_ready(fc.bias)
        dlb = self._get('dl_bf16', (B, self.fc_kpad))
        _lib.check(lib.rart_f32_to_bf16_rows(_lib.ptr(dl), _lib.ptr(dlb), B, self.n_classes, self.fc_kpad, sp))
        pooled = acts['pooled']
        # classifier weight gradient: [classes][features] = dl^T . pooled  (1x1 "conv" over B pixels)
        self._wgrad(dlb.view(B, 1, 1, self.fc_kpad), self.n_classes, self.fc_kpad, pooled.view(B, 1, 1, self.fc_in), (1, 1),
                    self.fc_in, (1, 1), [(0, 0)], 1, fc.weight.grad)
        self.on_grad_ready(fc.weight)
        dpool = self._get('dpool', (B, self.fc_in))
        self._gemm(dlb, self.fc_wd, dpool, B, (1, 1), (1, 1), self.fc_kpad, self.fc_kpad, [(0, 0)], self.fc_in, (1, 1),
                   self.fc_in)
        xl, xlhw = acts['last']
        d_out = self._get('g_out_a', tuple(xl.shape))
        _lib.check(lib.rart_engine_avgpool_bwd(_lib.ptr(xl), _lib.ptr(dpool), _lib.ptr(d_out), B, xlhw[0] * xlhw[1],
                                               self.fc_in, sp))
        for bi in range(len(self.blocks) - 1, -1, -1):
            ca, cb, cc, ds = self.blocks[bi]
            x, xhw, za, ya, zb, yb, zc, zd, out, ohw = acts['b%d' % bi]
            rows_o, rows_i = B * ohw[0] * ohw[1], B * xhw[0] * xhw[1]
            dzc = self._get('g_zc', tuple(zc.shape))
            skip_bits = self._ysign.get(out.data_ptr()) if (self.bit_masks and self.masked_skip and ds is None) else None
            g = None if skip_bits is not None else self._get('g_skip', tuple(zc.shape))
            self._bn_bwd(cc, d_out, out, zc, dzc, rows_o, g_out=g)
            self._conv_wgrad(cc, dzc, ohw, yb, ohw)
            dyb = self._get('g_yb', tuple(yb.shape))
            self._conv_dgrad(cc, dzc, ohw, dyb, ohw)
            dzb = self._get('g_zb', tuple(zb.shape))
            self._bn_bwd(cb, dyb, yb, zb, dzb, rows_o)
            self._conv_wgrad(cb, dzb, ohw, ya, xhw)
            dya = self._get('g_ya', tuple(ya.shape))
            if cb.stride == 2:
                dya.zero_()                      # parity classes without taps receive no gradient
            self._conv_dgrad(cb, dzb, ohw, dya, xhw)
            dza = self._get('g_za', tuple(za.shape))
            self._bn_bwd(ca, dya, ya, za, dza, rows_i)
            self._conv_wgrad(ca, dza, xhw, x, xhw)
            dx = self._get('g_x_%d' % (bi % 2), tuple(x.shape))
            if ds is None and skip_bits is not None:
                # the identity skip's gradient d_out . [out > 0] is formed in the dgrad epilogue from d_out and the sign bits of `out`
                self._conv_dgrad(ca, dza, xhw, dx, xhw, res=d_out, res_mask_bits=skip_bits)
            elif ds is None:
                self._conv_dgrad(ca, dza, xhw, dx, xhw, res=g)
            else:
                self._conv_dgrad(ca, dza, xhw, dx, xhw)
                dzd = self._get('g_zd', tuple(zd.shape))
                self._bn_bwd(ds, g, None, zd, dzd, rows_o)
                self._conv_wgrad(ds, dzd, ohw, x, xhw)
                # accumulate the projection skip (a 1x1 stride-2 conv reaches only the even/even pixels)
                self._conv_dgrad(ds, dzd, ohw, dx, xhw, res=dx)
            d_out = dx
        # stem: max-pool backward (applies y1's ReLU mask), BatchNorm backward, weight gradient on the padded hi plane
        y1, z1 = acts['y1'], acts['z1']
        h1, w1 = H // 2, W // 2
        dy1 = self._get('g_y1', tuple(y1.shape))
        _lib.check(lib.rart_engine_maxpool_bwd(_lib.ptr(y1), _lib.ptr(acts['parg']), _lib.ptr(d_out), _lib.ptr(dy1), B, h1, w1,
                                               64, sp))
        dz1 = self._get('g_z1', tuple(z1.shape))
        self._bn_bwd(self.stem, dy1, y1, z1, dz1, B * h1 * w1)
        taps = [(r, s) for r in range(7) for s in range(7)]    # hi plane holds the image at offset (3, 3)
        self._wgrad(dz1, 64, 64, acts['hi'][0], (H + 8, W + 8), 4, (h1, w1), taps, 2, self.model.conv1.weight.grad, c_valid=3)
        self.on_grad_ready(self.model.conv1.weight)
