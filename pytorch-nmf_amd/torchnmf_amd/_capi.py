"""ctypes binding of libnmfmu.so (the C ABI declared in include/nmfmu.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C
pytorch-nmf_amd/csrc`` and sits next to this file.  There is NO fallback: if
the shared object is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

LIB_NAME = 'libnmfmu.so'
LIB_PATH = os.environ.get('NMFMU_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)
# (NMFMU_LIB: experiment hook for kernel-variant builds, e.g. `make VARIANT=_x EXTRA=-DNMFMU_ORDER=0`)

OK = 0
ERR_UNSUPPORTED = -2
ERR_ARG = -3
ERR_ALLOC = -4
PREC_BF16, PREC_BF16X3, PREC_F16, PREC_F16X, PREC_F16R = 0, 1, 2, 3, 4
STAGE_REG, STAGE_DMA, STAGE_DMA_SPLIT = 0, 1, 2
STAGE_DMA_NOP2 = 3          # as STAGE_DMA; nothing reads the factors' transposed images (include/nmfmu.h)
BETA_KL, BETA_EUC, BETA_IS, BETA_GEN = 0, 1, 2, 3
KERNEL_FUSED, KERNEL_PP, KERNEL_SP = 0, 1, 2

PRECISIONS = {'bf16': PREC_BF16, 'bf16x3': PREC_BF16X3, 'f16': PREC_F16, 'f16x': PREC_F16X, 'f16r': PREC_F16R}


class NmfmuError(RuntimeError):
    pass


class Factor(C.Structure):
    """struct nmfmu_factor"""
    _fields_ = [('f', C.c_void_p), ('p1_hi', C.c_void_p), ('p1_lo', C.c_void_p), ('p2_hi', C.c_void_p),
                ('p2_lo', C.c_void_p), ('colsum', C.c_void_p), ('colsum_part', C.c_void_p), ('rows', C.c_int32),
                ('rows_pad', C.c_int32)]


class Step(C.Structure):
    """struct nmfmu_step"""
    _fields_ = [('xp', C.c_void_p), ('owner', Factor), ('panel', Factor), ('slab_num', C.c_void_p),
                ('slab_den', C.c_void_p), ('rank', C.c_int32), ('r_pad', C.c_int32), ('nsplit', C.c_int32),
                ('precision', C.c_int32), ('stage', C.c_int32), ('block_rows', C.c_int32), ('beta', C.c_float), ('gamma', C.c_float),
                ('l1', C.c_float), ('l2', C.c_float), ('status', C.c_void_p), ('stamps', C.c_void_p)]


class GemmDesc(C.Structure):
    """struct nmfmu_gemm_desc"""
    _fields_ = [('a_hi', C.c_void_p), ('a_lo', C.c_void_p), ('b_hi', C.c_void_p), ('b_lo', C.c_void_p),
                ('m_pad', C.c_int32), ('n_pad', C.c_int32), ('k_pad', C.c_int32), ('precision', C.c_int32),
                ('beta', C.c_float), ('x', C.c_void_p), ('gn_hi', C.c_void_p), ('gn_lo', C.c_void_p),
                ('gp_hi', C.c_void_p), ('gp_lo', C.c_void_p), ('out', C.c_void_p), ('m_valid', C.c_int32),
                ('n_valid', C.c_int32), ('ops', C.c_int32), ('t_batch', C.c_int32), ('t_rank', C.c_int32),
                ('t_taps', C.c_int32), ('t_lh', C.c_int32), ('tile_rows', C.c_int32), ('n_ld', C.c_int32), ('k_len', C.c_int32), ('k_split', C.c_int32),
                ('tail_rows', C.c_int32), ('rag_c0', C.c_int32), ('rag_channels', C.c_int32),
                ('win_nd', C.c_int32), ('win_lh', C.c_int32 * 3), ('win_taps', C.c_int32 * 3), ('win_channels', C.c_int32),
                ('win_pitch', C.c_int32), ('win_fold', C.c_int32), ('t_koff', C.c_void_p), ('stage_mode', C.c_int32)]


ABI_VERSION = 9   # include/nmfmu.h: NMFMU_ABI_VERSION
EPI_RATIO, EPI_F32, EPI_LOSS, EPI_FOLD = 0, 1, 2, 3
OPS_PLANES, OPS_B_HU, OPS_B_HUT, OPS_A_HU, OPS_A_WIN = 0, 1, 2, 3, 4

# name -> (restype, argtypes); every symbol include/nmfmu.h declares
SIGNATURES = {
    'nmfmu_abi_version': (C.c_int, []),
    'nmfmu_abi_check': (C.c_int, [C.c_int]),
    'nmfmu_pad_rows': (C.c_int, [C.c_int]),
    'nmfmu_pad_rank': (C.c_int, [C.c_int]),
    'nmfmu_beta_kind': (C.c_int, [C.c_float]),
    'nmfmu_supported': (C.c_int, [C.c_int, C.c_int]),
    'nmfmu_block_rows': (C.c_int, [C.c_int, C.c_int, C.c_float]),
    'nmfmu_step_block_rows': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]),
    'nmfmu_choose_nsplit': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'nmfmu_kernel_family': (C.c_int, [C.c_int, C.c_int, C.c_float]),
    'nmfmu_choose_nsplit_for': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int]),
    'nmfmu_xp_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'nmfmu_image_bytes': (C.c_size_t, [C.c_int, C.c_int]),
    'nmfmu_slab_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'nmfmu_colsum_part_bytes': (C.c_size_t, [C.c_int, C.c_int]),
    'nmfmu_pack_x': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                               C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'nmfmu_pack_factor': (C.c_int, [C.POINTER(Factor), C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'nmfmu_mu_partial': (C.c_int, [C.POINTER(Step), C.c_void_p]),
    'nmfmu_den_partial': (C.c_int, [C.POINTER(Step), C.c_void_p]),
    'nmfmu_mu_step': (C.c_int, [C.POINTER(Step), C.c_void_p, C.c_int, C.c_void_p]),
    'nmfmu_xb_supported': (C.c_int, [C.c_int, C.c_int, C.c_float]),
    'nmfmu_gram_ws_bytes': (C.c_size_t, [C.c_int]),
    'nmfmu_gram_panel': (C.c_int, [C.POINTER(Factor), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    'nmfmu_xb_partial': (C.c_int, [C.POINTER(Step), C.c_void_p]),
    'nmfmu_xb_step': (C.c_int, [C.POINTER(Step), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'nmfmu_colsum_nparts': (C.c_int, [C.POINTER(Step)]),
    'nmfmu_pack_nparts': (C.c_int, [C.c_int]),
    'nmfmu_colsum_finalize': (C.c_int, [C.POINTER(Factor), C.c_int, C.c_int, C.c_void_p]),
    'nmfmu_slab_reduce': (C.c_int, [C.POINTER(Step), C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_mu_apply': (C.c_int, [C.POINTER(Step), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'nmfmu_trainer_apply': (C.c_int, [C.POINTER(Step), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float,
                                      C.c_void_p, C.c_void_p]),
    'nmfmu_loss_part_count': (C.c_int, [C.c_int, C.c_int, C.c_int]),
    'nmfmu_loss': (C.c_int, [C.POINTER(Step), C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_riding_loss_supported': (C.c_int, [C.POINTER(Step)]),
    'nmfmu_riding_loss_part_count': (C.c_int, [C.POINTER(Step)]),
    'nmfmu_target_sums_nparts': (C.c_int, []),
    'nmfmu_target_sums': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_mu_step_with_loss': (C.c_int, [C.POINTER(Step), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_loss_checkpoint': (C.c_int, [C.POINTER(Step), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                        C.c_void_p, C.c_int64, C.c_void_p]),
    'nmfmu_beta_div': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_mu_terms': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_trainer_update': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                       C.c_float, C.c_void_p, C.c_void_p]),
    'nmfmu_norms': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_reconstruct': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                    C.c_void_p]),
    'nmfmu_gemm': (C.c_int, [C.POINTER(GemmDesc), C.c_int, C.c_void_p]),
    'nmfmu_gemm_window_staged': (C.c_int, [C.POINTER(GemmDesc), C.c_int]),
    'nmfmu_pack2d': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_int64,
                               C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_convnd_unfold': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'nmfmu_convnd_fold_apply_h': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float,
                                            C.c_void_p]),
    'nmfmu_conv_apply_pack_w': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_sp_partial': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                                   C.c_void_p, C.c_int, C.c_void_p]),
    'nmfmu_sp_loss_neg': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_gram_part_bytes': (C.c_size_t, [C.c_int]),
    'nmfmu_gram': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_rowmat': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'nmfmu_pack_factor_scaled': (C.c_int, [C.POINTER(Factor), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'nmfmu_plca_part_bytes': (C.c_size_t, [C.c_int, C.c_int]),
    'nmfmu_plca_em': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_plca_normalize': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    'nmfmu_plca_scale': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'nmfmu_plca_z': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    'nmfmu_conv_pack_w_scaled': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_conv_pack_wk': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_conv_apply_h_rows': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    'nmfmu_conv_apply_pack_w_wk': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                             C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_conv_h_rows_parts': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'nmfmu_conv_apply_h_rows_sums': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                               C.c_void_p]),
    'nmfmu_convnd_table_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'nmfmu_convnd_tables': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_convnd_koff': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'nmfmu_conv_rows_fold': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                       C.c_void_p]),
    'nmfmu_slab_sum': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    'nmfmu_convnd_fold': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_void_p]),
    'nmfmu_plca3_part_bytes': (C.c_size_t, [C.c_int]),
    'nmfmu_plca3': (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_float,
                              C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_conv_table_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'nmfmu_conv_tables': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    'nmfmu_conv_unfold': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'nmfmu_rank_sums': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_conv_apply_w': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_float, C.c_float, C.c_float, C.c_void_p]),
    'nmfmu_conv_fold_apply_h': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    'nmfmu_gemm_f16_supported': (C.c_int, [C.c_float, C.c_int, C.c_int]),
    'nmfmu_conv_tables_f16': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_conv_ragged_supported': (C.c_int, [C.c_int, C.c_int]),
    'nmfmu_conv_ragged_blocks': (C.c_int, [C.c_int, C.c_int, C.c_int]),
    'nmfmu_gemm_ragged_supported': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'nmfmu_conv_ragged_rows': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_float, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_conv_apply_pack_w_sums': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                               C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p]),
    'nmfmu_fold_hsum_parts': (C.c_int, [C.c_int, C.c_int]),
    'nmfmu_conv_fold_parts_apply_h_sums': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                     C.c_float, C.c_float, C.c_float, C.c_void_p]),
    'nmfmu_conv_fold_parts_apply_h_tail': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                     C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'nmfmu_fold_hsum_parts_tables': (C.c_int, [C.c_int, C.c_int]),
    'nmfmu_conv_fold_parts_apply_h_tables': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                       C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int,
                                                       C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                       C.c_void_p, C.c_void_p]),
    'nmfmu_fold_part_bytes': (C.c_size_t, [C.c_int, C.c_int]),
    'nmfmu_fold_parts_supported': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'nmfmu_conv_fold_parts_apply_h': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    'nmfmu_comm_available': (C.c_int, []),
    'nmfmu_comm_unique_id': (C.c_int, [C.c_void_p]),
    'nmfmu_comm_init_rank': (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int]),
    'nmfmu_comm_init_all': (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]),
    'nmfmu_comm_nranks': (C.c_int, [C.c_void_p]),
    'nmfmu_comm_allreduce_sum_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'nmfmu_comm_allreduce_sum_f32_multi': (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_size_t,
                                                     C.POINTER(C.c_void_p), C.c_int]),
    'nmfmu_comm_destroy': (C.c_int, [C.c_void_p]),
    'nmfmu_mu_step_allreduce': (C.c_int, [C.POINTER(Step), C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_timer_create': (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    'nmfmu_timer_record': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    'nmfmu_timer_elapsed_ms': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    'nmfmu_timer_destroy': (C.c_int, [C.c_void_p]),
    'nmfmu_probe_mfma': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'nmfmu_probe_lds_dma': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'nmfmu_debug_set_buffer': (C.c_int, [C.c_void_p]),
    'nmfmu_ubench_mfma_hbm': (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p]),
    'nmfmu_ubench_mfma_hbm2': (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libnmfmu.so once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NmfmuError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; '
                             f'g.build()"` or `make -C pytorch-nmf_amd/csrc`. torchnmf_amd has no CPU fallback.')
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the header and the library diverge
            fn.restype = res
            fn.argtypes = args
        if lib.nmfmu_abi_version() != ABI_VERSION:
            raise NmfmuError('libnmfmu.so ABI version mismatch')
        _lib = lib
    return _lib


def check(code: int, what: str) -> None:
    if code == OK:
        return
    if code == ERR_UNSUPPORTED:
        raise NotImplementedError(f'{what}: combination not supported by libnmfmu (rank > 256, or bf16x3 with rank > 128)')
    if code == ERR_ARG:
        raise ValueError(f'{what}: libnmfmu rejected the arguments')
    if code == ERR_ALLOC:
        raise MemoryError(f'{what}: a host allocation inside libnmfmu failed')
    raise NmfmuError(f'{what}: HIP error {code}')
