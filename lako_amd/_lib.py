"""ctypes loader for liblako_hip.so (the C-ABI declared in include/lako_hip.h).

The library is the product: there is NO CPU or PyTorch fallback.  `load()` raises if the shared object
is missing, and every op raises `LakoError` with the library's own message on a non-zero return.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LAKO_LIB") or os.path.join(_HERE, "liblako_hip.so")   # LAKO_LIB: A/B measurements of two builds

ABI_VERSION = 4          # include/lako_hip.h LAKO_ABI_VERSION this binding was written for
LAKO_F32, LAKO_BF16, LAKO_FP8_E4M3 = 0, 1, 2
EPI_RELU, EPI_RESID, EPI_AUXMASK, EPI_ATOMIC, EPI_NORM_A = 1, 2, 4, 8, 16

i64, i32, u32, f32, vp = C.c_int64, C.c_int, C.c_uint32, C.c_float, C.c_void_p


class LakoError(RuntimeError):
    pass


class Dropout(C.Structure):
    _fields_ = [("p", f32), ("seed", u32), ("site", u32)]


NO_DROP = Dropout(0.0, 0, 0)


class Tuning(C.Structure):
    """lako_tuning_t: kernel-selection knobs, owned by the caller (one per HipOps — the library keeps no tuning state)"""
    _fields_ = [(n, i32) for n in ("nt_variant", "nt_tail_split", "nt_ring", "nt_skinny", "nt_side_lds", "nt_wide_epi", "nt_group_m",
                                   "nt_persistent", "nt_stagger", "nt_dephase", "nt_dephase_n", "tn_big", "tn_split", "nt_debug",
                                   "nt_store_aux", "nt_tile192", "nt_queue", "nt_pp", "nt_glds", "nt_tile288", "nt_four", "tn_four")] + [("reserved", i32 * 10)]


class GemmNT(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("M", i64), ("N", i64), ("K", i64), ("lda", i64), ("ldb", i64),
                ("ldc", i64), ("in_dtype", i32), ("out_dtype", i32), ("alpha", f32), ("flags", i32), ("resid", vp),
                ("ldr", i64), ("aux", vp), ("ldaux", i64), ("aux_scale", f32), ("drop", Dropout), ("tuning", vp),
                ("norm_w", vp), ("norm_eps", f32), ("norm_out", vp), ("norm_ld", i64), ("norm_rstd", vp)]


class GemmTNItem(C.Structure):
    _fields_ = [("a", vp), ("b", vp), ("c", vp), ("M", i64), ("N", i64), ("lda", i64), ("ldb", i64), ("ldc", i64),
                ("alpha", f32), ("rows_out", i32)]


TN_GROUP_MAX = 48
HB_MULTI_MAX = 24          # problems per lako_headbatch_tn_multi launch


class HeadBatch(C.Structure):
    _fields_ = [("a", vp), ("b", vp), ("c", vp), ("a_sb", i64), ("a_st", i64), ("a_sh", i64),
                ("b_sb", i64), ("b_st", i64), ("b_sh", i64), ("ldb", i64), ("c_sb", i64), ("c_st", i64), ("c_sh", i64),
                ("M", i32), ("T", i32), ("H", i32), ("N", i32), ("K", i32), ("a_dtype", i32), ("b_dtype", i32),
                ("n_slabs", i32), ("slab_stride", i64)]


class AttnFwd(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("v", vp), ("out", vp), ("lse", vp),
                ("q_stride_b", i64), ("q_stride_t", i64), ("k_stride_b", i64), ("k_stride_t", i64),
                ("v_stride_b", i64), ("v_stride_t", i64), ("o_stride_b", i64), ("o_stride_t", i64),
                ("rel_bias", vp), ("R", i32), ("rel_off", i32), ("key_mask", vp), ("causal", i32),
                ("causal_off", i32), ("Bn", i32), ("H", i32), ("Lq", i32), ("Lk", i32), ("d_head", i32),
                ("dtype", i32), ("drop", Dropout), ("scores_out", vp), ("q_off", vp), ("k_off", vp), ("order", vp)]


class AttnBwd(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("v", vp), ("out", vp), ("dout", vp), ("lse", vp),
                ("dq_out", vp), ("dk_out", vp), ("dv_out", vp),
                ("q_stride_b", i64), ("q_stride_t", i64), ("k_stride_b", i64), ("k_stride_t", i64),
                ("v_stride_b", i64), ("v_stride_t", i64), ("o_stride_b", i64), ("o_stride_t", i64),
                ("rel_bias", vp), ("drel", vp), ("R", i32), ("rel_off", i32), ("key_mask", vp), ("causal", i32),
                ("causal_off", i32), ("Bn", i32), ("H", i32), ("Lq", i32), ("Lk", i32), ("d_head", i32),
                ("dtype", i32), ("drop", Dropout), ("q_off", vp), ("k_off", vp), ("order", vp)]


# name -> argtypes (restype is always int).  Must list EVERY function include/lako_hip.h declares:
# tests/test_abi.py cross-checks this table against the header and the .so's dynamic symbols.
SIGNATURES = {
    "lako_version": [],
    "lako_last_error": [C.c_char_p, C.c_size_t],
    "lako_gemm_nt": [C.POINTER(GemmNT), vp],
    "lako_mx_quantize": [vp, i64, i64, i64, vp, vp, vp],
    "lako_rmsnorm_fwd_mx": [vp, vp, vp, vp, vp, vp, i64, i32, f32, vp],
    "lako_gemm_nt_mx": [C.POINTER(GemmNT), vp, vp, vp],
    "lako_gemm_tn": [vp, vp, vp, i64, i64, i64, i64, i64, i64, i32, f32, i32, vp, vp],
    "lako_gemm_tn_grouped": [C.POINTER(GemmTNItem), i32, i64, i32, i32, vp, vp, i64, vp],
    "lako_gemm_tn_grouped_workspace": [C.POINTER(GemmTNItem), i32, i64, i32, i32, vp],
    # data-parallel communication (RCCL behind the C-ABI; the Python host itself uses torch.distributed — lako_amd/dist.py) and the
    # workspace query of SURVEY.md §8 b2
    "lako_comm_unique_id": [C.POINTER(C.c_uint8)],
    "lako_comm_init": [C.POINTER(vp), i32, i32, C.POINTER(C.c_uint8)],
    "lako_comm_world_size": [vp],
    "lako_allreduce": [vp, vp, i64, i32, vp],
    "lako_comm_destroy": [vp],
    "lako_workspace_bytes": [i32, vp],
    "lako_rmsnorm_fwd": [vp, vp, vp, vp, i64, i32, f32, i32, Dropout, vp],
    "lako_rmsnorm_bwd": [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, Dropout, vp, Dropout, vp],
    "lako_embed_fwd": [vp, vp, vp, i64, i32, i64, i32, Dropout, vp],
    "lako_embed_bwd": [vp, vp, vp, i64, i32, i64, i32, Dropout, vp],
    "lako_embed_bwd_ordered": [vp, vp, vp, vp, i64, i32, i64, i32, Dropout, vp],
    "lako_deterministic": [],
    "lako_relpos_expand": [vp, vp, vp, i32, i32, i32, vp],
    "lako_relpos_reduce": [vp, vp, vp, i32, i32, i32, vp],
    "lako_attn_fwd": [C.POINTER(AttnFwd), vp],
    "lako_attn_bwd": [C.POINTER(AttnBwd), vp],
    "lako_xattn_scores": [vp, i64, i64, vp, i64, vp, vp, i64, vp, i64, i32, i32, i32, vp],
    "lako_xattn_context": [vp, i64, vp, i64, vp, vp, vp, i64, i64, i64, i32, i32, i32, i32, vp],
    "lako_xattn_softmax_fwd": [vp, i64, vp, vp, i64, vp, vp, i32, i32, i32, i32, Dropout, vp],
    "lako_xattn_softmax_bwd": [vp, vp, i64, vp, vp, i64, vp, vp, i32, i32, i32, i32, Dropout, vp],
    "lako_xattn_decode": [vp, i64, i64, vp, i64, vp, vp, vp, i32, i32, i32, i32, vp],
    "lako_xattn_decode_combine": [vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, vp],
    "lako_headbatch_nt": [C.POINTER(HeadBatch), vp],
    "lako_headbatch_tn": [C.POINTER(HeadBatch), vp],
    "lako_headbatch_tn_multi": [C.POINTER(HeadBatch), i32, vp],
    "lako_ce_fwd_bwd": [vp, vp, vp, vp, vp, i64, i64, i32, vp],
    "lako_sumsq": [vp, i64, vp, vp],
    "lako_adamw_step": [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, f32, f32, i32, vp],
    "lako_transpose_cast": [vp, vp, i64, i64, i32, vp],
    "lako_transpose_cast_batched": [vp, i32, vp, vp, vp, i32, i32, i32, vp],
    "lako_cast": [vp, vp, i64, i32, vp],
    "lako_dropout_apply": [vp, vp, i64, i32, Dropout, vp],
    "lako_shift_right": [vp, vp, i32, i32, vp],
    "lako_pack_ids": [vp, vp, vp, i32, i32, vp],
    "lako_greedy_step": [vp, i64, i32, vp, i64, i32, vp, vp, vp, i64, i64, vp],
    "lako_fact_scores": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "lako_topk": [vp, i64, i64, i64, i32, vp, vp, vp],
    "lako_pq_assign": [vp, i64, i64, vp, i32, i32, i32, vp, vp, vp, vp, vp],
    "lako_pq_lut": [vp, i64, i64, vp, i32, i32, i32, vp, vp],
    "lako_pq_scan": [vp, vp, i64, i64, i32, i32, vp, i64, vp],
    "lako_layernorm_fwd": [vp, vp, vp, vp, vp, vp, i64, i32, f32, i32, vp],
    "lako_bert_embed": [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i64, f32, i32, vp],
    "lako_bias_act": [vp, vp, vp, i64, i32, i32, i32, vp],
    "lako_seq_mean": [vp, vp, vp, i32, i32, i32, i32, vp],
    "lako_bi_score": [vp, vp, vp, i32, i32, i32, f32, vp],
    "lako_kldiv_fwd": [vp, vp, vp, i32, i32, vp],
    "lako_layernorm_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, i32, vp],
    "lako_bias_act_bwd": [vp, vp, vp, vp, vp, i64, i32, i64, i32, f32, i32, vp],
    "lako_seq_mean_bwd": [vp, vp, vp, i32, i32, i32, i32, vp],
    "lako_bi_score_bwd": [vp, vp, vp, vp, vp, i32, i32, i32, f32, vp],
    "lako_kldiv_bwd": [vp, vp, vp, vp, i32, i32, vp],
    "lako_bert_embed_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i64, f32, i32, vp],
    "lako_tuning_init": [C.POINTER(Tuning)],
    "lako_tuning_set": [C.POINTER(Tuning), C.c_char_p, i32],
}

_lib = None


def load(path: str | None = None):
    """dlopen the kernel library.  `import torch` must already have happened in the process so that the
    library binds to the HIP runtime torch loaded (it is linked without an rpath to /opt/rocm)."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise LakoError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        f"(lako_amd/csrc/build.sh).  There is no CPU fallback.")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    # the version FIRST: a stale library lacks symbols this binding names, and a bare AttributeError would hide the reason
    try:
        lib.lako_version.restype = C.c_int
        have = lib.lako_version()
    except AttributeError:
        have = None
    if have != ABI_VERSION:     # structs and argument lists differ between versions: a mismatch reads garbage pointers
        raise LakoError(f"{path} implements C-ABI version {have}, this package binds version {ABI_VERSION} "
                        f"(include/lako_hip.h LAKO_ABI_VERSION): rebuild with lako_amd/csrc/build.sh")
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int64 if name in ("lako_gemm_tn_grouped_workspace", "lako_workspace_bytes") else C.c_int
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        buf = C.create_string_buffer(512)
        _lib.lako_last_error(buf, 512)
        raise LakoError(f"{what} failed (rc={rc}): {buf.value.decode(errors='replace')}")
