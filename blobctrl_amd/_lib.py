"""ctypes binding of libblobctrl_hip.so (the C ABI declared in include/blobctrl_hip.h).

The product path has NO fallback: if the shared library is missing or fails to load, importing any compute entry
point raises.  Build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BLOBCTRL_HIP_LIB points at another build of the SAME C ABI (A/B measurements of kernel variants); default = the in-tree build
LIB_PATH = os.environ.get("BLOBCTRL_HIP_LIB") or os.path.join(_HERE, "libblobctrl_hip.so")

A_DENSE, A_CONV3X3 = 0, 1
ACT_NONE, ACT_GELU, ACT_GEGLU, ACT_SILU, ACT_QUICK_GELU = 0, 1, 2, 3, 4
OUT_F16, OUT_F16_T, OUT_F32 = 0, 1, 2
TILE_NAMES = ["auto", "256x128", "128x128_s3", "128x128_s2", "256x64_s2", "256x64_s3", "128x64", "64x64", "halo128x160", "wreg128x160",
              "gw64x128", "gw64x256", "gw64x320", "g256x256"]
TILE_HALO = 8
TILE_WREG = 9
TILE_GW64x128, TILE_GW64x256, TILE_GW64x320 = 10, 11, 12
TILE_G256 = 13                       # gemm256.hip: 256 x 256 tiles, 8 waves, 8-phase LDS-DMA pipeline, persistent workgroups
GW_TILES = {TILE_GW64x128: 2, TILE_GW64x256: 4, TILE_GW64x320: 5}      # configuration -> 16-column tiles per wave (BN = 64 NT)


class BcGemm(C.Structure):
    """Mirror of `struct BcGemm` (include/blobctrl_hip.h) - keep field order identical."""
    _fields_ = [
        ("A", C.c_void_p), ("A2", C.c_void_p), ("a_mode", C.c_int),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("lda", C.c_int), ("lda2", C.c_int), ("C1", C.c_int), ("Cin", C.c_int),
        ("Hin", C.c_int), ("Win", C.c_int), ("Hv", C.c_int), ("Wv", C.c_int),
        ("Hout", C.c_int), ("Wout", C.c_int), ("stride", C.c_int), ("conv_nopad_lo", C.c_int),
        ("W", C.c_void_p), ("ldw", C.c_int),
        ("bias", C.c_void_p), ("rowvec", C.c_void_p), ("ld_rowvec", C.c_int), ("rows_per_batch", C.c_int),
        ("rowvec_idx", C.c_void_p), ("rowvec_step", C.c_int),
        ("act", C.c_int), ("colscale", C.c_void_p), ("alpha", C.c_float),
        ("alpha_dev", C.c_void_p), ("alpha_idx", C.c_void_p), ("alpha_bstride", C.c_int),
        ("R", C.c_void_p), ("ldr", C.c_int),
        ("R2", C.c_void_p), ("ldr2", C.c_int), ("r2_xmin", C.c_int), ("r2_bmod", C.c_int), ("out_w", C.c_int),
        ("out_mode", C.c_int), ("C", C.c_void_p), ("ldc", C.c_int),
        ("splitk", C.c_int), ("slab", C.c_void_p), ("gn_tot", C.c_void_p), ("tile_cfg", C.c_int),
        ("a_affine", C.c_void_p), ("a_act", C.c_int),
        ("a_tot1", C.c_void_p), ("a_tot2", C.c_void_p),
        ("a_gamma", C.c_void_p), ("a_beta", C.c_void_p), ("a_groups", C.c_int), ("a_eps", C.c_float),
        ("ln_colsum", C.c_void_p), ("ln_eps", C.c_float),
        ("C_t", C.c_void_p), ("ldc_t", C.c_int), ("n_t0", C.c_int),
        ("w_bstride", C.c_longlong), ("vec_bstride", C.c_int), ("sm_group", C.c_int), ("sm_valid", C.c_int), ("sm_keep", C.c_int),
    ]


_SIGNATURES = {
    "bc_last_error": (C.c_char_p, []),
    "bc_version": (C.c_int, []),
    "bc_device_info": (C.c_int, [C.POINTER(C.c_int)]),
    "bc_gemm": (C.c_int, [C.POINTER(BcGemm), C.c_void_p]),
    "bc_sizeof_gemm": (C.c_int, []),
    "bc_gemm_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                               C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bc_conv_halo_eligible": (C.c_int, [C.c_int] * 8),
    "bc_conv_halo_max_chunks": (C.c_int, []),
    "bc_conv_wreg_pack": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "bc_gemm_wreg_pack": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "bc_gemm_wreg_stream_elems": (C.c_longlong, [C.c_int, C.c_int]),
    "bc_gemm_wreg_eligible": (C.c_int, [C.c_int] * 5),
    "bc_gemm256_eligible": (C.c_int, [C.c_int] * 7),
    "bc_gn_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "bc_gn_finalize": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bc_gn_apply_fused": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_void_p]),
    "bc_memset_zero": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p]),
    "bc_dup_halves": (C.c_int, [C.c_void_p, C.c_longlong] * 6 + [C.c_void_p]),
    "bc_gn_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                              C.c_void_p, C.c_void_p]),
    "bc_softmax_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "bc_gaussian_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "bc_layernorm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                               C.c_int, C.c_void_p]),
    "bc_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.c_longlong,
                               C.c_longlong, C.c_float, C.c_void_p]),
    "bc_attention_causal": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.c_longlong,
                                      C.c_longlong, C.c_float, C.c_void_p]),
    "bc_embed_tokens": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p]),
    "bc_splat_scores": (C.c_int, [C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "bc_assemble_input": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "bc_assemble_input_im2col": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p]),
    "bc_timestep_embedding": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "bc_timestep_embedding_table": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "bc_silu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "bc_cfg_scheduler_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int,
                                        C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "bc_nchw_to_nhwc_f16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "bc_nhwc_to_nchw": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "bc_add_cls_pos": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "bc_patchify": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "bc_rowchain_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "bc_rowchain_stream_frags": (C.c_longlong, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "bc_rowchain": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                              C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float,
                              C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "bc_rowchain_sum": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bc_rowchain_kv_frags": (C.c_longlong, [C.c_int]),
    "bc_rowchain_pack_kv": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "bc_ctx_fold": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bc_rowchain_midx": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                                   C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    # ---- plan runtime (plan.hip)
    "bc_plan_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "bc_plan_destroy": (C.c_int, [C.c_void_p]),
    "bc_plan_segment": (C.c_int, [C.c_void_p, C.c_char_p]),
    "bc_plan_find_segment": (C.c_int, [C.c_void_p, C.c_char_p]),
    "bc_plan_new_event": (C.c_int, [C.c_void_p]),
    "bc_plan_add_gemm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(BcGemm)]),
    "bc_plan_add_op": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.c_int]),
    "bc_plan_set_slab": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "bc_plan_enable": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "bc_plan_num_launches": (C.c_int, [C.c_void_p, C.c_int]),
    "bc_step": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int]),
    "bc_plan_capture": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int]),
    "bc_plan_release": (C.c_int, [C.c_void_p, C.c_int]),
    "bc_plan_capture_loop": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]),
    "bc_plan_run_timed": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_float)]),
    "bc_plan_run_timed_kernels": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "bc_plan_run_marked": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_float)]),
    "bc_plan_save": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]),
    "bc_plan_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "bc_plan_buffer": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_longlong)]),
    "bc_graph_begin": (C.c_int, [C.c_void_p]),
    "bc_graph_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "bc_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bc_graph_destroy": (C.c_int, [C.c_void_p]),
    "bc_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "bc_event_create_sync": (C.c_int, [C.POINTER(C.c_void_p)]),
    "bc_stream_wait_event": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bc_event_record": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bc_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "bc_event_destroy": (C.c_int, [C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES.keys())

# Recordable entry points of the plan runtime: name -> BC_OP_* code (include/blobctrl_hip.h).  The argument kinds of an op are derived
# from its ctypes signature above (stream excluded): p pointer, i int, f float, l long long - the same strings plan.hip checks.
OPS = {"bc_gemm": 0, "bc_gn_stats": 1, "bc_gn_finalize": 2, "bc_gn_apply_fused": 3, "bc_gn_apply": 4, "bc_layernorm": 5,
       "bc_attention": 6, "bc_attention_causal": 7, "bc_assemble_input": 8, "bc_timestep_embedding": 9,
       "bc_timestep_embedding_table": 10, "bc_cfg_scheduler_step": 11, "bc_embed_tokens": 12, "bc_softmax_rows": 13,
       "bc_patchify": 14, "bc_add_cls_pos": 15, "bc_silu": 16, "bc_nchw_to_nhwc_f16": 17, "bc_nhwc_to_nchw": 18,
       "bc_gaussian_sample": 19, "bc_rowchain": 22, "bc_assemble_input_im2col": 23, "bc_memset_zero": 24,
       "bc_rowchain_midx": 25, "bc_rowchain_pack_kv": 26, "bc_rowchain_sum": 27, "bc_ctx_fold": 28, "bc_dup_halves": 29}
OP_SIGNAL, OP_WAIT = 20, 21
CHAIN_IN, CHAIN_MID, CHAIN_OUT, CHAIN_OUT_FF, CHAIN_OUT_TAIL, CHAIN_MIDX, CHAIN_OUT_FFP = 0, 1, 2, 3, 4, 5, 6
GN_TOT_WORDS = 6                      # 64-bit words per (image, channel) of a GroupNorm statistics table (include/blobctrl_hip.h)
_KIND = {C.c_void_p: "p", C.c_int: "i", C.c_float: "f", C.c_longlong: "l", C.c_char_p: "p"}


def op_signature(name):
    """Argument kinds of a recordable entry point without its trailing stream argument."""
    args = _SIGNATURES[name][1][:-1]
    return "".join("p" if (a not in _KIND) else _KIND[a] for a in args)


class BcPlanBuffer(C.Structure):
    _fields_ = [("name", C.c_char_p), ("address", C.c_void_p), ("bytes", C.c_longlong), ("host_data", C.c_void_p)]

_lib = None


class BlobCtrlHipError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises BlobCtrlHipError if it is missing - there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BlobCtrlHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  blobctrl_amd has no CPU fallback.")
    # PyTorch (the device-memory / stream provider of this package) ships its own libamdhip64; if this library were loaded first
    # it would pull in the system copy and the process would end up with TWO HIP runtimes - the second one then reports "no
    # ROCm-capable device".  Importing torch first makes the loader bind our HIP calls to the runtime torch already uses.
    try:
        import torch  # noqa: F401
    except ImportError:   # pragma: no cover - plain C / ctypes use without PyTorch: the system HIP runtime is the only one
        pass
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:   # pragma: no cover
        raise BlobCtrlHipError(f"failed to load {LIB_PATH}: {e}") from e
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError => symbol missing from the build
        fn.restype = res
        fn.argtypes = args
    if lib.bc_sizeof_gemm() != C.sizeof(BcGemm):
        raise BlobCtrlHipError(f"BcGemm layout mismatch: C {lib.bc_sizeof_gemm()} vs ctypes {C.sizeof(BcGemm)}")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().bc_last_error()
        raise BlobCtrlHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
