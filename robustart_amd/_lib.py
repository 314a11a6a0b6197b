"""ctypes binding of librobustart_hip.so (the C-ABI in include/robustart_hip.h).

PyTorch is used only as plumbing: device memory (tensor.data_ptr()), the current HIP stream
and torch.distributed.  Fails loudly when the shared library is absent -- never falls back.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'librobustart_hip.so')

c_void_p, c_int, c_size_t, c_u64, c_float = (ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t,
                                              ctypes.c_uint64, ctypes.c_float)
c_double = ctypes.c_double

ABI_VERSION = 109        # == RART_ABI_VERSION of include/robustart_hip.h; load() refuses a library built from another header

# name -> (restype, argtypes); every symbol include/robustart_hip.h declares
SIGNATURES = {
    'rart_version': (c_int, []),
    'rart_last_error_string': (ctypes.c_char_p, []),
    'rart_corruption_name': (ctypes.c_char_p, [c_int]),
    'rart_corrupt_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'rart_corrupt_u8': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_u64, c_u64,
                                ctypes.POINTER(c_void_p), c_int, c_void_p, c_size_t, c_void_p]),
    'rart_stencil_fixed_point_info': (c_int, [c_int, c_int, c_void_p, c_void_p, c_size_t]),
    'rart_noise_multi_u8': (c_int, [c_void_p, ctypes.POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_int),
                                    ctypes.POINTER(c_u64), c_u64, c_void_p]),
    'rart_frost_textures_u8': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, ctypes.POINTER(c_int),
                                       c_u64, c_u64, c_void_p]),
    'rart_pil_resize_workspace_bytes': (c_size_t, [c_int] * 10),
    'rart_pil_resize_u8': (c_int, [c_void_p, c_void_p] + [c_int] * 10 + [c_void_p, c_size_t, c_void_p]),
    'rart_cv_resize_workspace_bytes': (c_size_t, [c_int] * 10),
    'rart_cv_resize_u8': (c_int, [c_void_p, c_void_p] + [c_int] * 10 + [c_void_p, c_size_t, c_void_p]),
    'rart_u8_to_normalized': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'rart_u8_to_unit_f32_nchw': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'rart_rng_uniform_u32': (c_int, [c_void_p, c_int, c_size_t, c_u64, c_u64, c_int, c_void_p]),
    'rart_rng_normal_f32': (c_int, [c_void_p, c_int, c_size_t, c_u64, c_u64, c_int, c_void_p]),
    'rart_set_normal_generator': (c_int, [c_int]),
    'rart_get_normal_generator': (c_int, []),
    'rart_rng_noise_field_f32': (c_int, [c_void_p, c_int, c_size_t, c_u64, c_u64, c_void_p]),
    'rart_attack_workspace_bytes': (c_size_t, [c_int]),
    'rart_attack_init_linf': (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_float, c_float, c_float,
                                      c_u64, c_u64, c_void_p, c_void_p, c_void_p]),
    'rart_pgd_step_linf': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float, c_void_p]),
    'rart_pgd_step_l2': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_float, c_float,
                                 c_void_p, c_size_t, c_void_p]),
    'rart_pgd_step_l1': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_float, c_float,
                                 c_void_p, c_size_t, c_void_p]),
    'rart_random_start_l1': (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_float, c_u64, c_u64, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_size_t, c_void_p]),
    'rart_mim_step': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_float, c_float,
                              c_float, c_void_p, c_size_t, c_void_p]),
    'rart_apgd_init': (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_int, c_float, c_u64, c_u64, c_void_p,
                               c_void_p, c_void_p, c_size_t, c_void_p]),
    'rart_apgd_step': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_int,
                               c_float, c_float, c_void_p, c_size_t, c_void_p]),
    'rart_square_init_linf': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_u64, c_u64, c_void_p, c_void_p,
                                      c_void_p]),
    'rart_rng_signs_f32': (c_int, [c_void_p, c_int, c_int, c_u64, c_u64, c_void_p, c_int, ctypes.c_uint32, c_void_p]),
    'rart_rng_normal_rows_f32': (c_int, [c_void_p, c_int, c_size_t, c_u64, c_void_p, c_int, c_void_p]),
    'rart_square_propose_linf': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_int,
                                         c_int, c_void_p, c_void_p]),
    'rart_fab_project_linf': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_void_p]),
    'rart_row_dot': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_void_p]),
    'rart_row_absmax_diff': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_void_p]),
    'rart_eot_accumulate': (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_float, c_void_p]),
    'rart_square_init_lp': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                    c_void_p, c_void_p, c_void_p]),
    'rart_square_propose_lp': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_int, c_int, c_int,
                                       c_int, c_void_p, c_void_p, c_void_p]),
    'rart_fab_project': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_int, c_void_p]),
    'rart_row_norm_diff': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_int, c_void_p]),
    'rart_fab_update': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_float, c_void_p]),
    'rart_fab_backoff': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_float, c_void_p]),
    'rart_l1_project': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_float, c_int, c_int, c_void_p]),
    'rart_row_kth_abs': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_void_p]),
    'rart_apgd_l1_move': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_void_p]),
    'rart_row_count_diff': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_void_p]),
    'rart_select_rows': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_void_p]),
    'rart_logit_loss': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p,
                                c_void_p, c_void_p, c_void_p]),
    'rart_conv_igemm_bf16': (c_int, [c_void_p, c_void_p]),
    'rart_conv3x3_halo_supported': (c_int, [c_int, c_int, c_int]),
    'rart_conv3x3_pack_frag_bf16': (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    'rart_pack_frag_bf16': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'rart_conv3x3_halo_bf16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                       ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_int, c_void_p]),
    'rart_bottleneck28_fused_supported': (c_int, [c_int, c_int, c_int, c_int]),
    'rart_bottleneck28_fused_bf16': (c_int, [c_void_p] * 11 + [c_int] * 5 + [ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_int,
                                             c_void_p]),
    'rart_bottleneck7_fused_supported': (c_int, [c_int, c_int, c_int, c_int]),
    'rart_bottleneck7_fused_bf16': (c_int, [c_void_p] * 11 + [c_int] * 5 + [ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_int,
                                            c_void_p]),
    'rart_bottleneck_s2_fwd_supported': (c_int, [c_int] * 5),
    'rart_bottleneck_s2_fwd_bf16': (c_int, [c_void_p] * 12 + [c_int] * 6 + [c_void_p]),
    'rart_bottleneck_s2_bwd_bf16': (c_int, [c_void_p] * 9 + [c_int] * 6 + [c_void_p]),
    'rart_bottleneck14_fused_supported': (c_int, [c_int, c_int, c_int, c_int]),
    'rart_bottleneck14_fused_bf16': (c_int, [c_void_p] * 11 + [c_int] * 5 + [ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_int,
                                             c_void_p]),
    'rart_bottleneck_fused_supported': (c_int, [c_int, c_int, c_int, c_int]),
    'rart_bottleneck_first_supported': (c_int, [c_int] * 5),
    'rart_bottleneck_first_bf16': (c_int, [c_void_p] * 12 + [c_int] * 6 + [ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_int,
                                           c_void_p]),
    'rart_bottleneck_fused_bf16': (c_int, [c_void_p] * 11 + [c_int] * 5 + [ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_int,
                                           c_void_p]),
    'rart_igemm_set_bk64_min_k': (c_int, [ctypes.c_longlong]),
    'rart_gemm_small_m_bf16': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'rart_igemm_set_gemm256': (c_int, [c_int]),
    'rart_gemm256_supported': (c_int, [ctypes.c_longlong, c_int, c_int, c_int, c_int]),
    'rart_gemm_pair_bf16': (c_int, [c_void_p, c_void_p]),
    'rart_copy_calibration': (c_int, [c_void_p, c_void_p, ctypes.c_size_t, c_int, c_void_p]),
    'rart_gemm_pair_set_schedule': (c_int, [c_int]),
    'rart_gemm_pair_get_schedule': (c_int, []),
    'rart_pack_jobs_bf16': (c_int, [c_void_p, c_int, c_int, c_void_p]),
    'rart_vit_add_pos_cls_pair': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'rart_layernorm_pair': (c_int, [c_void_p] * 6 + [c_int, c_int, ctypes.c_int64, ctypes.c_int64, c_float, c_void_p]),
    'rart_layernorm_bwd_pair': (c_int, [c_void_p] * 9 + [c_int, c_int] + [ctypes.c_int64] * 4 + [c_float, c_void_p]),
    'rart_softmax_rows_pair': (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_int, c_int, c_int, c_float, c_void_p]),
    'rart_softmax_bwd_rows_pair': (c_int, [c_void_p] * 5 + [ctypes.c_int64, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    'rart_vit_attention_pair': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'rart_vit_attention_bwd_pair': (c_int, [c_void_p] * 9 + [c_int, c_int, c_int, c_int, c_void_p]),
    'rart_vit_unpatchify_from_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, ctypes.c_int64, ctypes.POINTER(c_float), c_void_p]),
    'rart_engine_prep_input': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                       c_void_p]),
    'rart_engine_maxpool': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'rart_engine_maxpool_keep': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'rart_engine_maxpool_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'rart_engine_avgpool': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'rart_engine_avgpool_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'rart_engine_stem_col2im': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'rart_engine_stem_fwd_fused': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                           c_int, c_void_p, c_void_p, c_void_p]),
    'rart_wgrad_direct_supported': (c_int, [c_int, c_int, c_int]),
    'rart_wgrad_direct_bf16': (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 10 + [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'rart_conv3x3_tail_pair_supported': (c_int, [c_int]),
    'rart_conv3x3_tail_pair': (c_int, [c_void_p, c_void_p]),
    'rart_engine_stem_fwd_fused_pair': (c_int, [c_void_p, c_int] + [c_void_p] * 7 + [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'rart_engine_stem_bwd_fused_pair': (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_void_p, c_void_p]),
    'rart_engine_stem_bwd_fused': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'rart_engine_maxpool_pair': (c_int, [c_void_p, ctypes.c_longlong, c_void_p, ctypes.c_longlong, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'rart_engine_maxpool_bwd_pair': (c_int, [c_void_p, c_void_p, ctypes.c_longlong, c_void_p, ctypes.c_longlong, c_int, c_int, c_int, c_int, c_void_p]),
    'rart_engine_avgpool_pair': (c_int, [c_void_p, ctypes.c_longlong, c_void_p, ctypes.c_longlong, c_int, c_int, c_int, c_void_p]),
    'rart_engine_avgpool_bwd_pair': (c_int, [c_void_p, c_void_p, ctypes.c_longlong, c_void_p, ctypes.c_longlong, c_int, c_int, c_int, c_void_p]),
    'rart_f32_to_pair_rows': (c_int, [c_void_p, c_void_p, ctypes.c_longlong, c_int, c_int, c_int, c_void_p]),
    'rart_engine_stem_col2im_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'rart_f32_to_bf16_rows': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'rart_vit_patchify': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                  c_void_p]),
    'rart_vit_add_pos_cls': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'rart_layernorm_bf16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_int64, ctypes.c_int64,
                                    c_float, c_void_p]),
    'rart_softmax_rows_bf16': (c_int, [c_void_p, c_void_p, ctypes.c_int64, c_int, c_int, c_int, c_float, c_void_p]),
    'rart_vit_attention': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'rart_vit_attention_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'rart_vit_transpose_v': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'rart_gelu_bf16': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'rart_gelu_bwd_bf16': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'rart_layernorm_bwd_bf16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_int64,
                                        ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_float, c_void_p]),
    'rart_softmax_bwd_rows_bf16': (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_int, c_int, c_int, c_int, c_float,
                                           c_void_p]),
    'rart_vit_unpatchify_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, ctypes.c_int64,
                                        ctypes.POINTER(c_float), c_void_p]),
    'rart_layernorm_bwd_workspace_bytes': (c_size_t, [c_int]),
    'rart_layernorm_bwd_full_bf16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_int64,
                                             ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_float, c_void_p, c_void_p, c_int,
                                             c_void_p, c_size_t, c_void_p]),
    'rart_colsum_workspace_bytes': (c_size_t, [c_int, c_int]),
    'rart_colsum_bf16': (c_int, [c_void_p, ctypes.c_int64, c_int, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'rart_sgd_step_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_double, c_double, c_double, c_int,
                                  c_double, c_double, c_int, c_void_p]),
    'rart_adamw_step_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_double, c_double,
                                    c_double, c_double, c_double, c_int, c_double, c_double, c_int, c_void_p]),
    'rart_ema_update_f32': (c_int, [c_void_p, c_void_p, c_size_t, c_double, c_void_p]),
    'rart_label_smooth_ce_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_double, c_double, c_void_p, c_void_p,
                                         c_void_p]),
    'rart_bn_workspace_bytes': (c_size_t, [c_size_t, c_int]),
    'rart_bn_train_forward_bf16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_double, c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                           c_size_t, c_void_p]),
    'rart_bn_train_backward_bf16': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t,
                                            c_void_p]),
    'rart_transpose_gather_bf16': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                           c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.c_longlong,
                                           c_int, c_int, c_void_p]),
    'rart_wgrad_reduce_f32': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    'rart_pack_conv_weight_bf16': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                           ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_int, c_int, c_void_p]),
}


class ConvDesc(ctypes.Structure):
    """rart_conv_desc (include/robustart_hip.h)."""
    _fields_ = [('src', c_void_p), ('wgt', c_void_p), ('bias', c_void_p), ('res', c_void_p), ('mask', c_void_p),
                ('dst', c_void_p),
                ('batch', ctypes.c_int32), ('grid_h', ctypes.c_int32), ('grid_w', ctypes.c_int32),
                ('src_h', ctypes.c_int32), ('src_w', ctypes.c_int32), ('src_pix_stride', ctypes.c_int32),
                ('k_per_tap', ctypes.c_int32), ('n_taps', ctypes.c_int32), ('sy', ctypes.c_int32), ('sx', ctypes.c_int32),
                ('tap_dy', ctypes.c_int32 * 32), ('tap_dx', ctypes.c_int32 * 32), ('tap_src_off', ctypes.c_int64 * 32),
                ('n_cols', ctypes.c_int32),
                ('dst_h', ctypes.c_int32), ('dst_w', ctypes.c_int32), ('dst_sy', ctypes.c_int32), ('dst_sx', ctypes.c_int32),
                ('dst_oy', ctypes.c_int32), ('dst_ox', ctypes.c_int32), ('dst_pix_stride', ctypes.c_int32),
                ('flags', ctypes.c_int32),
                ('n_batched', ctypes.c_int32), ('z_inner', ctypes.c_int32), ('wgt_row_stride', ctypes.c_int32),
                ('reserved_', ctypes.c_int32),
                ('src_z_outer', ctypes.c_int64), ('src_z_inner', ctypes.c_int64), ('wgt_z_outer', ctypes.c_int64),
                ('wgt_z_inner', ctypes.c_int64), ('dst_z_outer', ctypes.c_int64), ('dst_z_inner', ctypes.c_int64),
                ('sign_out', c_void_p), ('dst_pair_off', ctypes.c_int64), ('res_pair_off', ctypes.c_int64), ('bn_stats_out', c_void_p)]


class FixedPointInfo(ctypes.Structure):
    """rart_fixed_point_info (include/robustart_hip.h): the fixed-point table of a matrix-core stencil path."""
    _fields_ = [(n, ctypes.c_int32) for n in ('kind', 'ksize', 'n_steps', 'frac_bits', 'out_frac_bits')] + \
               [('corr', ctypes.c_longlong), ('band', ctypes.c_longlong), ('max_abs_weight_error', ctypes.c_double),
                ('sum_abs_weight_error', ctypes.c_double)]


class PackJob(ctypes.Structure):
    """rart_pack_job (include/robustart_hip.h): one table job of rart_pack_jobs_bf16."""
    _fields_ = [(n, ctypes.c_int32) for n in ('kind', 'n_out', 'channels', 'r', 's', 'n_taps', 'transpose', 'rows_padded', 'rows', 'k')] + \
               [('tap_r', ctypes.c_int32 * 16), ('tap_s', ctypes.c_int32 * 16), ('weight', c_void_p), ('out_channel_scale', c_void_p),
                ('src16', c_void_p), ('out', c_void_p)]


class GemmPairDesc(ctypes.Structure):
    """rart_gemm_pair_desc (include/robustart_hip.h): the split-bf16 GEMM of the reference-precision engines."""
    _fields_ = [(n, c_void_p) for n in ('a_hi', 'a_lo', 'w_hi', 'w_lo', 'bias', 'res_hi', 'res_lo', 'dst_hi', 'dst_lo', 'aux_hi', 'aux_lo')] + \
               [(n, ctypes.c_int32) for n in ('M', 'N', 'K', 'lda', 'ldw', 'ldc', 'w_rows', 'rows_per_image', 'src_rows_per_image',
                                              'src_row_off', 'dst_rows_per_image', 'dst_row_off', 'flags', 'n_batched', 'z_inner')] + \
               [(n, ctypes.c_int64) for n in ('a_z_outer', 'a_z_inner', 'w_z_outer', 'w_z_inner', 'c_z_outer', 'c_z_inner')] + \
               [(n, ctypes.c_int32) for n in ('conv', 'batch', 'grid_h', 'grid_w', 'src_h', 'src_w', 'sy', 'sx', 'k_per_tap', 'n_taps')] + \
               [('tap_dy', ctypes.c_int32 * 16), ('tap_dx', ctypes.c_int32 * 16)] + \
               [(n, ctypes.c_int32) for n in ('dst_h', 'dst_w', 'dst_sy', 'dst_sx', 'dst_oy', 'dst_ox')] + \
               [('mask_bits', c_void_p), ('sign_out', c_void_p), ('tile_n', ctypes.c_int32), ('tile_m', ctypes.c_int32)]


class ConvTailDesc(ctypes.Structure):
    """rart_conv_tail_desc (include/robustart_hip.h): 3x3 + 1x1 expansion of a Bottleneck on split-bf16 pairs, one launch."""
    _fields_ = [(n, c_void_p) for n in ('a_hi', 'a_lo', 'w_hi', 'w_lo', 't_hi', 't_lo', 'bias_mid', 'bias_out', 'mask_mid', 'mask_out',
                                        'sign_mid', 'sign_out', 'res_hi', 'res_lo', 'dst_hi', 'dst_lo')] + \
               [(n, ctypes.c_int32) for n in ('batch', 'h', 'w', 'c_mid', 'ldw', 'relu_mid', 'relu_out')] + \
               [('tap_dy', ctypes.c_int32 * 9), ('tap_dx', ctypes.c_int32 * 9)] + \
               [(n, c_void_p) for n in ('n_hi', 'n_lo', 'bias_next', 'mask_next', 'sign_next', 'dstn_hi', 'dstn_lo')] + [('relu_next', ctypes.c_int32)]


_lib = None


class RartError(RuntimeError):
    """A C-ABI call returned a non-zero rart_status."""

    def __init__(self, status, message):
        super().__init__('robustart_hip status %d: %s' % (status, message))
        self.status = status


def load():
    """Load the shared library and attach prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'robustart_amd: %s is missing -- build it with `python robustart_amd/csrc/build.py` '
            '(or __graft_entry__.build()).  There is no CPU fallback for the product path.' % LIB_PATH)
    # torch first: its bundled libamdhip64 must be THE HIP runtime of the process; loading ours
    # first would pull /opt/rocm's copy in and the two runtimes do not share devices/streams.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = header/library drift: fail loudly
        fn.restype = res
        fn.argtypes = args
    got = lib.rart_version()
    if got != ABI_VERSION:
        raise RuntimeError('robustart_amd: %s reports ABI version %d, this binding expects %d (struct layouts / signatures '
                           'differ) -- rebuild it with `python robustart_amd/csrc/build.py --force`' % (LIB_PATH, got, ABI_VERSION))
    _lib = lib
    return lib


def check(status):
    if status != 0:
        raise RartError(status, load().rart_last_error_string().decode('utf-8', 'replace'))


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError('robustart_amd: no GPU visible (torch.cuda.is_available() is False); '
                           'the HIP hot path has no CPU fallback')
    return torch


def stream_ptr():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else c_void_p(t.data_ptr())


_ws_cache = {}


def workspace(nbytes, device):
    """A cached, grow-only scratch buffer per (device, current stream): kernels enqueued on different streams (a loader
    stream corrupting the next batch while an attack runs) never share scratch memory; within one stream the launches
    are ordered, so one buffer is safe.  torch's allocator owns the memory."""
    import torch
    if nbytes <= 0:
        return None
    dev = torch.device(device)
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf
