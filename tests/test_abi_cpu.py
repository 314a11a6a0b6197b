"""CPU-only checks of the boundary: the C-ABI library loads and exports every symbol the header
declares, and the Python drop-in reproduces the reference's API behaviour."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, 'include', 'robustart_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(rart_[a-z0-9_]+)\s*\(', src)))


def test_library_loads_and_exports_every_declared_symbol():
    from robustart_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), 'header declares %s but the library does not export it' % s
    # and the ctypes table covers the header exactly (no drift in either direction)
    assert sorted(_lib.SIGNATURES) == syms
    assert _lib.load().rart_version() == _lib.ABI_VERSION


def test_argument_validation_without_gpu():
    """Validation happens before any HIP call, so these are safe on a GPU-less box."""
    from robustart_amd import _lib
    lib = _lib.load()
    assert lib.rart_corruption_name(0) == b'gaussian_noise'
    assert lib.rart_corruption_name(18) == b'saturate'
    assert lib.rart_corruption_name(19) is None
    st = lib.rart_corrupt_u8(None, None, 1, 224, 224, 99, 3, 0, 0, None, 0, None, 0, None)
    assert st == 1 and b'unknown corruption' in lib.rart_last_error_string()
    st = lib.rart_corrupt_u8(None, None, 1, 224, 224, 0, 6, 0, 0, None, 0, None, 0, None)
    assert st == 1 and b'severity' in lib.rart_last_error_string()
    # empty batch: the reference's loop body never runs -> OK, nothing touched
    assert lib.rart_corrupt_u8(None, None, 0, 224, 224, 0, 3, 0, 0, None, 0, None, 0, None) == 0
    assert lib.rart_corrupt_workspace_bytes(11, 3, 4, 224, 224) > 0     # contrast needs channel sums
    assert lib.rart_attack_workspace_bytes(256) >= 256 * 32 * 4
    # gaussian_blur stages its input when called in place (fused separable kernel): the workspace covers the staging copy
    assert lib.rart_corrupt_workspace_bytes(16, 3, 4, 224, 224) >= 4 * 224 * 224 * 3 * 9        # + the fp64 two-pass scratch
    # geometry predicates of the fused Bottleneck kernels (host-only functions)
    assert lib.rart_bottleneck_fused_supported(256, 64, 56, 56) == 1 and lib.rart_bottleneck_fused_supported(256, 64, 28, 28) == 0
    assert lib.rart_bottleneck_first_supported(64, 64, 256, 56, 56) == 1 and lib.rart_bottleneck_first_supported(64, 64, 256, 24, 24) == 0
    assert lib.rart_bottleneck28_fused_supported(512, 128, 28, 28) == 1 and lib.rart_bottleneck28_fused_supported(512, 128, 14, 14) == 0
    assert lib.rart_bottleneck14_fused_supported(1024, 256, 14, 14) == 1 and lib.rart_bottleneck14_fused_supported(1024, 256, 6, 6) == 0
    assert lib.rart_bottleneck7_fused_supported(2048, 512, 7, 7) == 1 and lib.rart_bottleneck7_fused_supported(2048, 512, 3, 3) == 0
    assert lib.rart_conv3x3_halo_supported(64, 56, 56) == 1 and lib.rart_conv3x3_halo_supported(512, 7, 7) == 0
    # aliasing / argument checks happen before any launch
    st = lib.rart_bottleneck14_fused_bf16(None, None, None, None, None, None, None, None, None, None, None, 1, 14, 14, 1024, 256,
                                          None, None, 0, None)
    assert st == 1 and b'bad arguments' in lib.rart_last_error_string()


def test_addnoise_api_matches_reference_behaviour():
    from robustart_amd.noise import AddNoise
    from robustart_amd.noise.registry import noise_list, default_config, function_dict
    from robustart_amd.noise.adv import attack_list
    from robustart_amd.noise import imagenet_c
    assert noise_list == ['imagenet-s', 'imagenet-c', 'pgd_linf', 'pgd_l2', 'fgsm', 'autoattack_linf',
                          'mim_linf', 'pgd_l1']
    assert set(function_dict) == set(noise_list) and set(attack_list) == set(noise_list[2:])
    assert default_config['pgd_linf'] == {'f_model': None, 'eps': 8 / 255, 'rel_stepsize': 3 / 40, 'steps': 20}
    assert default_config['mim_linf']['step_size'] == 0.002 and 'model' in default_config['mim_linf']
    assert [f.__name__ for f in imagenet_c.corruption_tuple][:4] == ['gaussian_noise', 'shot_noise',
                                                                      'impulse_noise', 'defocus_blur']
    assert len(imagenet_c.corruption_tuple) == 19 and imagenet_c.corruption_tuple[15].__name__ == 'speckle_noise'
    with pytest.raises(KeyError):          # add_noise.py:13 raises KeyError before its assert
        AddNoise('nope')
    a = AddNoise('imagenet-c')
    with pytest.raises(AssertionError):    # add_noise.py:21-22
        a.set_config(bogus=1)
    a.set_config(corruption_name='gaussian_noise', severity=3)
    assert a.config == {'severity': 3, 'corruption_name': 'gaussian_noise', 'corruption_number': -1}
    # documented deviation: instances do not share config (the reference aliases the module dict)
    assert AddNoise('imagenet-c').config['severity'] == 1
    with pytest.raises(AssertionError):    # path input only for imagenet-c / -s
        AddNoise('pgd_linf').add_noise('some/file.png')
    with pytest.raises(ValueError):        # imagenet_c/__init__.py:32-33, raised before any device work
        imagenet_c.corrupt(np.zeros((1, 8, 8, 3), np.uint8), severity=1)


def test_product_path_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from robustart_amd.noise import AddNoise
    a = AddNoise('imagenet-c')
    a.config.update(corruption_name='gaussian_noise', severity=3)
    with pytest.raises(RuntimeError, match='no GPU'):
        a.add_noise(np.zeros((1, 224, 224, 3), np.uint8))


def test_product_never_imports_oracle():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import robustart_amd.noise, robustart_amd.noise.registry; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'" % ROOT)
    subprocess.check_call([sys.executable, '-c', code])
    for dp, _, fs in os.walk(os.path.join(ROOT, 'robustart_amd')):
        for f in fs:
            if f.endswith('.py'):
                assert not re.search(r'^\s*(from|import)\s+oracle', open(os.path.join(dp, f)).read(), re.M), f


def test_threefry_known_answers():
    """Random123 kat_vectors for threefry2x32 (13 and 20 rounds) pin the host mirror; the GPU test
    then pins the device generator against this mirror."""
    from robustart_amd.noise.rng import threefry2x32
    kat = [(13, (0, 0), (0, 0), (0x9d1c5ec6, 0x8bd50731)),
           (13, (0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0xfd36d048, 0x2d17272c)),
           (13, (0x243f6a88, 0x85a308d3), (0x13198a2e, 0x03707344), (0xba3e4725, 0xf27d669e)),
           (20, (0, 0), (0, 0), (0x6b200159, 0x99ba4efe)),
           (20, (0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x1cb996fc, 0xbb002be7)),
           (20, (0x243f6a88, 0x85a308d3), (0x13198a2e, 0x03707344), (0xc4923a9c, 0x483df7a0))]
    for rounds, ctr, key, want in kat:
        assert threefry2x32(key[0], key[1], ctr[0], ctr[1], rounds) == want


def test_host_uniform_vector_form_equals_the_scalar_form():
    """rng.host_uniform_many (numpy lanes; the per-image frost draws of a batch) == rng.host_uniform element by element."""
    import numpy as np
    from robustart_amd.noise import rng
    for seed, stream, index in ((0, 8, 0), (0x1234567890ABCDEF, 9, 3), (2 ** 64 - 1, 10, 0x0FFFFFFF)):
        samples = np.array([0, 1, 2, 255, 2 ** 31, 2 ** 32 - 1, 2 ** 32 + 5], dtype=np.int64)
        want = np.array([rng.host_uniform(seed, int(s), stream, index) for s in samples])
        np.testing.assert_array_equal(rng.host_uniform_many(seed, samples, stream, index), want)


def test_argument_checks_of_the_round3_attack_entries_without_gpu():
    """rart_fab_project / rart_row_norm_diff / rart_square_init_lp / rart_square_propose_lp validate before any launch."""
    import ctypes
    from robustart_amd import _lib
    lib = _lib.load()
    one = ctypes.c_void_p(16)           # never dereferenced: the checks run first
    assert lib.rart_fab_project(one, one, one, one, None, 2, 100, 3, None) == 1 and b'rart_fab_project' in lib.rart_last_error_string()
    assert lib.rart_fab_project(None, one, one, one, None, 2, 100, 2, None) == 1
    assert lib.rart_row_norm_diff(one, one, one, 2, 100, 5, None) == 1 and b'rart_row_norm_diff' in lib.rart_last_error_string()
    # Square L2 / L1: norm in {1, 2}, at most 4 channels, the tile grid / windows inside the image
    assert lib.rart_square_init_lp(one, one, 2, 3, 32, 32, 0.5, 0, 6, 1, 5, 5, one, one, one, None) == 1
    assert lib.rart_square_init_lp(one, one, 2, 5, 32, 32, 0.5, 2, 6, 1, 5, 5, one, one, one, None) == 1
    assert lib.rart_square_init_lp(one, one, 2, 3, 32, 32, 0.5, 2, 6, 4, 5, 5, one, one, one, None) == 1
    assert b'tile grid' in lib.rart_last_error_string()
    assert lib.rart_square_propose_lp(one, one, one, 2, 3, 32, 32, 0.5, 2, 4, 0, 0, 0, 29, one, one, None) == 1
    assert b'window outside' in lib.rart_last_error_string()
    assert lib.rart_square_propose_lp(one, one, one, 2, 3, 32, 32, 0.5, 1, 0, 0, 3, 4, 29, one, one, None) == 1


def test_square_eta_pattern_host_side():
    """adv._square_eta (the host-side eta of SquareAttack, square.py:146-186, without the random transposition): unit norm in the
    attack's norm, top half positive / bottom half negative, symmetric left-right for odd sides -- for the window sizes the schedule produces."""
    import torch
    from robustart_amd.noise.adv import _square_eta
    for norm in ('L2', 'L1'):
        for s in (3, 6, 7, 29, 44, 201):
            e = _square_eta(s, norm)
            assert e.shape == (s, s) and e.dtype == torch.float32
            n = e.pow(2).sum().sqrt() if norm == 'L2' else e.abs().sum()
            assert abs(float(n) - 1.0) < 1e-5
            assert (e[:s // 2] > 0).all() and (e[s // 2:] < 0).all()
            if s % 2 == 1:
                assert torch.allclose(e, e.flip(1), atol=1e-7)       # (odd sides: the rectangles are centred)


def test_round4_host_side_predicates_and_tables():
    """Host-only pieces of the round-4 entries (no GPU): which shapes the direct weight-gradient and the fused pair tail kernels accept, the
    descriptor validation of rart_conv3x3_tail_pair / rart_wgrad_direct_bf16, and the vectorised stem-backward table of the eval engine
    against its definition."""
    import torch
    from robustart_amd import _lib
    from robustart_amd.model.engine import ResNet50Engine
    lib = _lib.load()
    assert lib.rart_wgrad_direct_supported(64, 64, 9) == 1 and lib.rart_wgrad_direct_supported(1024, 256, 1) == 1
    assert lib.rart_wgrad_direct_supported(4, 64, 49) == 1            # the stem: 4-channel pixels, 49 taps
    assert lib.rart_wgrad_direct_supported(4, 128, 49) == 0 and lib.rart_wgrad_direct_supported(64, 64, 49) == 0
    assert lib.rart_wgrad_direct_supported(96, 64, 1) == 0 and lib.rart_wgrad_direct_supported(64, 1000, 1) == 0
    assert lib.rart_conv3x3_tail_pair_supported(64) == 1 and lib.rart_conv3x3_tail_pair_supported(128) == 1
    assert lib.rart_conv3x3_tail_pair_supported(256) == 0
    d = _lib.ConvTailDesc()
    assert lib.rart_conv3x3_tail_pair(ctypes.byref(d), None) == 1 and b'null operand plane' in lib.rart_last_error_string()
    assert lib.rart_wgrad_direct_bf16(None, None, None, 1, 8, 8, 64, 8, 8, 64, 1, 1, 1, None, None, 1, 64, 64, None) == 1
    # the stem-backward table: row (py*2+px)*3+c, column ((dp+1)*4+(dq+1))*64+k = W[k][c][py+3-2dp][px+3-2dq], zero where a tap leaves 0..6
    wb = torch.randn(64, 3, 7, 7)
    t = torch.zeros(16, 16, 64)
    for py in range(2):
        for px in range(2):
            for dp in range(-1, 3):
                for dq in range(-1, 3):
                    r, s_ = py + 3 - 2 * dp, px + 3 - 2 * dq
                    if 0 <= r <= 6 and 0 <= s_ <= 6:
                        for c in range(3):
                            t[(py * 2 + px) * 3 + c, (dp + 1) * 4 + (dq + 1)] = wb[:, c, r, s_]
    got = ResNet50Engine._stem_bwd_table(wb, dtype=torch.float32)
    assert torch.equal(got, t.reshape(16, 1024))
    assert ResNet50Engine._stem_bwd_table(wb).dtype == torch.bfloat16


def _signed_digits_to_int(frag_bytes, n_steps):
    """fragment table [step][digit][lane][16] of int8 digits -> integer weights per (step, lane, byte)"""
    import numpy as np
    d = np.frombuffer(frag_bytes, dtype=np.int8).reshape(n_steps, 4, 64, 16).astype(np.int64)
    return d[:, 0] + d[:, 1] * 256 + d[:, 2] * 65536 + d[:, 3] * 16777216


def test_fixed_point_tables_of_the_matrix_core_stencils_match_the_oracle_weights():
    """Round 5 host logic, no GPU: defocus_blur / gaussian_blur / glass_blur run as exact fixed-point filters on the i8 matrix cores.
    rart_stencil_fixed_point_info hands out the tables the kernels consume; here they are decoded with the documented lane mapping
    (include/robustart_hip.h) and compared with the oracle's weights: W / 2^frac_bits reproduces every tap to the stated error, taps
    outside the kernel are zero and the ambiguity band covers the accumulated quantisation error of a whole output."""
    import ctypes
    import numpy as np
    from robustart_amd import _lib
    from oracle import corruptions_np as O
    lib = _lib.load()
    names = O.CORRUPTION_NAMES
    for name in ('defocus_blur', 'gaussian_blur', 'glass_blur'):
        for sev in range(1, 6):
            info = _lib.FixedPointInfo()
            buf = (ctypes.c_ubyte * (11 * 4 * 64 * 16))()
            rc = lib.rart_stencil_fixed_point_info(names.index(name), sev, ctypes.byref(info), buf, len(buf))
            assert rc == 0, (name, sev)
            W = _signed_digits_to_int(bytes(buf)[:info.n_steps * 4 * 64 * 16], info.n_steps)        # [step][lane][byte]
            lane = np.arange(64)
            m, g = lane & 15, lane >> 4
            if name == 'defocus_blur':
                r, alias = O.PARAMS['defocus_blur'][sev - 1]
                k = O.disk_kernel(r, alias).astype(np.float64)
                ksz = 21 if sev == 5 else 17                       # radius 10 -> 21 x 21 (12 outputs per row block), else 17 x 17 (16)
                mout, steps = 33 - ksz, (ksz + 1) // 2
                assert info.kind == 1 and info.ksize == ksz == k.shape[0] and info.n_steps == steps and info.out_frac_bits == info.frac_bits
                F = info.frac_bits
                want = np.zeros((steps, 64, 16), np.int64)
                for j in range(steps):
                    for i in range(16):
                        a = 2 * j + (g >> 1)
                        b = 16 * (g & 1) + i - m
                        ok = (m < mout) & (a < ksz) & (b >= 0) & (b < ksz)
                        want[j, ok, i] = np.rint(k[a[ok], b[ok]] * 2.0 ** F).astype(np.int64)
                np.testing.assert_array_equal(W, want)
                err = np.abs(np.rint(k * 2.0 ** F) / 2.0 ** F - k)
                assert err.max() <= 2.0 ** -(F + 1) and info.max_abs_weight_error == pytest.approx(err.max(), abs=1e-30)
                assert info.corr == 128 * int(np.rint(k * 2.0 ** F).sum())
                assert info.band >= 255.0 * err.sum() * 2.0 ** F                     # the band covers the worst-case quantisation error
                assert info.band < 2.0 ** F * 1e-4                                   # ... and stays a 1e-4 sliver of an output step
            else:
                sigma = O.PARAMS['gaussian_blur'][sev - 1] if name == 'gaussian_blur' else O.PARAMS['glass_blur'][sev - 1][0]
                radius = int(4.0 * sigma + 0.5)
                w = O.gaussian_kernel1d(sigma, radius)
                assert info.kind == 2 and info.ksize == 2 * radius + 1 and info.n_steps == 1 and (info.frac_bits, info.out_frac_bits) == (31, 38)
                want = np.zeros((64, 16), np.int64)
                for i in range(16):
                    t = 16 * g + i - m
                    ok = (t >= 0) & (t <= 2 * radius)
                    want[ok, i] = np.rint(w[t[ok]] * 2.0 ** 31).astype(np.int64)
                # the library's weights come from libm's exp and a sequential sum, numpy's from its own exp and a pairwise sum: equal to
                # a few ulp, i.e. to one unit of 2^-31 at most
                assert np.abs(W[0] - want).max() <= 1
                assert (W[0][want == 0] == 0).all()
                assert info.max_abs_weight_error <= 2.0 ** -32 and info.band * 2.0 ** -38 < 1e-5
                assert info.band * 2.0 ** -38 >= 255.0 * 2 * info.sum_abs_weight_error
    bad = _lib.FixedPointInfo()
    assert lib.rart_stencil_fixed_point_info(names.index('zoom_blur'), 3, ctypes.byref(bad), None, 0) == 2
    assert lib.rart_stencil_fixed_point_info(names.index('gaussian_blur'), 9, ctypes.byref(bad), None, 0) == 1      # RART_ERR_INVALID
