"""ctypes binding of libcomposer_hip.so (C ABI declared in include/composer_hip.h).

There is NO CPU fallback: if the shared library is missing, or no MI355X is visible when a device
context is requested, this module raises.  (The numpy oracle under oracle/ is test infrastructure and is
never imported from here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# COMPOSER_HIP_LIB: kernel A/B runs load another build of the same library (tools/ab_build.py); never a CPU path
LIB_PATH = os.environ.get("COMPOSER_HIP_LIB") or os.path.join(_HERE, "lib", "libcomposer_hip.so")

CMP_FP32, CMP_BF16 = 0, 1
DECODE_LITERAL, DECODE_KV = 0, 1
KIND_VALUE, KIND_ADAM_M, KIND_ADAM_V, KIND_GRAD = 0, 1, 2, 3


class HipLibraryError(RuntimeError):
    pass


class ModelCfg(C.Structure):
    _fields_ = [
        ("vocab_size", C.c_int32), ("embedding_size", C.c_int32), ("window_size", C.c_int32),
        ("layers", C.c_int32), ("heads", C.c_int32), ("ln_eps", C.c_float),
        ("scale_attention", C.c_int32), ("use_layer_norm", C.c_int32),
        ("attn_dropout", C.c_float), ("resid_dropout", C.c_float),
        ("dtype", C.c_int32), ("max_batch", C.c_int32), ("max_seq", C.c_int32), ("seed", C.c_uint64),
    ]


_P = C.c_void_p
_i, _f, _i64, _u64, _u32 = C.c_int, C.c_float, C.c_int64, C.c_uint64, C.c_uint32

# name -> (restype, argtypes); every symbol declared in include/composer_hip.h
SIGNATURES = {
    "cmp_last_error": (C.c_char_p, []),
    "cmp_version": (_i, []),
    "cmp_build_key": (C.c_char_p, []),
    "cmp_device_count": (_i, []),
    "cmp_ctx_create": (_i, [_i, C.POINTER(_P)]),
    "cmp_ctx_destroy": (_i, [_P]),
    "cmp_sync": (_i, [_P]),
    "cmp_ctx_stream": (_P, [_P]),
    "cmp_dp_unique_id": (_i, [_P]),
    "cmp_dp_init": (_i, [_P, _i, _i, _P]),
    "cmp_dp_allreduce_test": (_i, [_P, _P, _i]),
    "cmp_dp_init_exchange": (_i, [_P, _i, _i, _P, _P]),
    "cmp_dp_test_hog": (_i, [_P, _i, _i]),
    "cmp_dp_set_gemm_cus": (_i, [_P, _i]),
    "cmp_dp_set_mask_rank": (_i, [_P, _i]),
    "cmp_dp_stats": (_i, [_P, _i, C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(_i)]),
    "cmp_dp_rccl_version": (_i, [C.POINTER(_i)]),
    "cmp_dp_allreduce_pattern": (_i, [_P, _i, C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(_i)]),
    "cmp_model_create": (_i, [_P, C.POINTER(ModelCfg), C.POINTER(_P)]),
    "cmp_model_destroy": (_i, [_P]),
    "cmp_param_count": (_i, [_P, C.POINTER(_i)]),
    "cmp_param_info": (_i, [_P, _i, C.POINTER(C.c_char_p), C.POINTER(_i), C.POINTER(_i64 * 4), C.POINTER(_i64)]),
    "cmp_param_get": (_i, [_P, C.c_char_p, _i, _P, _i64]),
    "cmp_param_set": (_i, [_P, C.c_char_p, _i, _P, _i64]),
    "cmp_adam_iter_get": (_i, [_P, C.POINTER(_i64)]),
    "cmp_adam_iter_set": (_i, [_P, _i64]),
    "cmp_train_step": (_i, [_P, _P, _P, _i, _i, _f, C.POINTER(_f), C.POINTER(_f)]),
    "cmp_train_step_dev": (_i, [_P, _P, _P, _i, _i, _f]),
    "cmp_train_metrics": (_i, [_P, C.POINTER(_f), C.POINTER(_f)]),
    "cmp_train_step_launches": (_i, [_P, _P, _P, _i, _i, C.POINTER(_i), C.POINTER(_i)]),
    "cmp_train_step_graph_probe": (_i, [_P, _P, _P, _i, _i, C.POINTER(_i), C.POINTER(_i), _i, C.POINTER(_f)]),
    "cmp_train_step_async": (_i, [_P, _P, _P, _i, _i, _f, C.POINTER(_i64)]),
    "cmp_train_metrics_wait": (_i, [_P, _i64, C.POINTER(_f), C.POINTER(_f)]),
    "cmp_loss_and_grads": (_i, [_P, _P, _P, _i, _i, C.POINTER(_f), C.POINTER(_f)]),
    "cmp_eval_step": (_i, [_P, _P, _P, _i, _i, C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(_i64)]),
    "cmp_present_get": (_i, [_P, _i, _i, _i, _P]),
    "cmp_forward_generation": (_i, [_P, C.POINTER(_i64)]),
    "cmp_present_get_at": (_i, [_P, _i, _i, _i, _i64, _P]),
    "cmp_hidden_get_at": (_i, [_P, _i, _i, _i, _i64, _P]),
    "cmp_forward_logits": (_i, [_P, _P, _i, _i, _P]),
    "cmp_forward": (_i, [_P, _P, _i, _i, _i, _P, _i, _P]),
    "cmp_forward_ex": (_i, [_P, _P, _i, _i, _i, _P, _i, _P, _P, _P, _P, _P]),
    "cmp_decode_begin": (_i, [_P, _P, _i, _i, _f, _u64]),
    "cmp_decode_steps": (_i, [_P, _i, _P]),
    "cmp_k_sample": (_i, [_P, _P, _i, _f, _u64, _u32, _i, _P]),
    "cmp_prof_begin": (_i, [_i]),
    "cmp_prof_end": (_i, [C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(C.c_double)]),
    "cmp_prof_pause": (_i, []),
    "cmp_prof_resume": (_i, []),
    "cmp_prof_end2": (_i, [C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "cmp_k_embed_fwd": (_i, [_P, _P, _P, _P, _P, _i, _i, _i, _i, _i, _f, _u64, _u32]),
    "cmp_k_embed_bwd": (_i, [_P, _P, _P, _P, _P, _i, _i, _i, _i, _i, _f, _u64, _u32]),
    "cmp_k_embed_bwd_v": (_i, [_P, _P, _P, _P, _P, _i, _i, _i, _i, _i, _f, _u64, _u32, _i]),
    "cmp_k_layernorm_fwd": (_i, [_P, _P, _P, _P, _P, _P, _P, _i, _i, _f, _i]),
    "cmp_k_layernorm_bwd": (_i, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _i, _i, _i]),
    "cmp_k_layernorm_bwd_ws": (_i64, [_i, _i]),
    "cmp_k_layernorm_bwd_fused": (_i, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _i, _i, _i, _P, _P, _f, _u64, _u32]),
    "cmp_k_gemm": (_i, [_P, _i, _i, _i, _i, _i, _i, _P, _i, _P, _i, _P, _i, _P, _i, _P, _i, _P, _i, _i, _i, _f, _u64, _u32, _i]),
    "cmp_k_embed_fwd_stats": (_i, [_P, _P, _P, _P, _P, _P, _i, _i, _i, _i, _f, _u64, _u32]),
    "cmp_k_ln_fold_prep": (_i, [_P, _P, _P, _P, _P, _P, _P, _P, _i, _i]),
    "cmp_gemm_ln_next": (_i, [_P, _i, _f, _P, _P, _P, _P]),
    "cmp_gemm_ln_scale_next": (_i, [_P, _i, _f]),
    "cmp_attn_bwd_ln_next": (_i, [_P]),
    "cmp_k_ln_stats_merge": (_i, [_P, _P, _i, _f, _P, _P, _i]),
    "cmp_k_layernorm_bwd_prescaled": (_i, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _i, _i, _P, _P, _f, _u64, _u32]),
    "cmp_k_wgrad_ln_fix": (_i, [_P, _P, _i, _i, _P, _P, _P]),
    "cmp_k_layernorm_bwd_parts": (_i, [_P, _P, _P, _P, _P, _P, _f, _P, _P, _P, _P, _P, _P, _i, _i, _P, _P, _f, _u64, _u32]),
    "cmp_model_path_info": (_i, [_P, C.POINTER(_i), C.POINTER(_i64)]),
    "cmp_k_wgrad_group": (_i, [_P, _i, _P, _P, _P, _P, _P, _P, _P, _P, _i]),
    "cmp_gemm_set_workspace": (_i, [_P, _i64]),
    "cmp_gemm_colsum_next": (_i, [_P]),
    "cmp_gemm_set_stamps": (_i, [_P]),
    "cmp_k_colsum": (_i, [_P, _P, _i, _P, _i, _i, _i]),
    "cmp_k_attn_fwd": (_i, [_P, _P, _P, _P, _i, _i, _i, _i, _i, _i, _f, _u64, _u32]),
    "cmp_attn_bwd_bias_next": (_i, [_P]),
    "cmp_k_attn_bwd": (_i, [_P, _P, _P, _P, _P, _P, _P, _i, _i, _i, _i, _i, _i, _f, _u64, _u32]),
    "cmp_k_softmax_xent": (_i, [_P, _P, _i, _P, _P, _P, _P, _i, _i, _f, _i]),
    "cmp_k_adam": (_i, [_P, _P, _P, _P, _P, _P, _i64, _f, _f, _f, _f, _i64, _f]),
}

# entry points added after round 3: an OLDER build of the library loaded through COMPOSER_HIP_LIB as the other arm of an A/B
# timing (tools/ab_step.py) may lack them; the package's own library must export every symbol
_ADDED_LATER = {"cmp_gemm_ln_scale_next", "cmp_attn_bwd_ln_next", "cmp_k_layernorm_bwd_prescaled", "cmp_k_wgrad_ln_fix", "cmp_k_ln_stats_merge", "cmp_dp_rccl_version", "cmp_dp_allreduce_pattern", "cmp_dp_init_exchange", "cmp_train_step_graph_probe", "cmp_train_step_launches", "cmp_k_embed_fwd_stats", "cmp_k_ln_fold_prep", "cmp_gemm_ln_next", "cmp_k_layernorm_bwd_parts", "cmp_model_path_info", "cmp_forward_ex", "cmp_hidden_get_at", "cmp_dp_stats", "cmp_prof_end2", "cmp_prof_pause", "cmp_prof_resume", "cmp_k_wgrad_group", "cmp_k_embed_bwd_v"}

_lib = None


def load():
    """Loads the shared library (no GPU needed for this) and sets every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            "libcomposer_hip.so not found at %s -- build it with `python -m composer_amd.build` "
            "(hipcc, gfx950).  composer_amd has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    check_single_runtime()
    for name, (res, args) in SIGNATURES.items():
        if name in _ADDED_LATER and os.environ.get("COMPOSER_HIP_LIB") and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)     # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_RUNTIME_STEMS = ("libamdhip64", "librccl", "libhsa-runtime64")


def mapped_runtime_libraries():
    """{stem: sorted real paths} of the HIP runtime / RCCL / HSA shared objects mapped into THIS process right now
    (/proc/self/maps).  One path per stem is the healthy state."""
    found = {k: set() for k in _RUNTIME_STEMS}
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rstrip("\n").split(None, 5)[-1] if line.count("/") else ""
                base = os.path.basename(path)
                for stem in _RUNTIME_STEMS:
                    if base.startswith(stem + ".so"):
                        found[stem].add(os.path.realpath(path))
    except OSError:
        pass
    return {k: sorted(v) for k, v in found.items()}


def check_single_runtime():
    """Raises when two DIFFERENT copies of the HIP runtime or of RCCL are mapped into this process.

    libcomposer_hip.so asks the dynamic linker for `libamdhip64.so.7` / `librccl.so.1` (rpath /opt/rocm/lib); torch's wheel
    bundles its own copies and asks for them as `libamdhip64.so` / `librccl.so`.  With torch imported FIRST the linker hands the
    library torch's copies (same SONAME, already loaded): one runtime, one RCCL -- the state bench.py and the test suite run in.
    With the library loaded first, a later `import torch` maps a second runtime beside /opt/rocm's: a process that then
    initialises both sees "No HIP GPUs are available" from torch.cuda at best, and its ranks may sit on another RCCL than the
    peers of the job.  The rule ("Library load order"): a process that uses torch at all imports it before
    composer_amd loads the library -- `composer_amd.cli` does so itself when WORLD_SIZE > 1."""
    # (libhsa-runtime64 is reported by runtime_info() but not judged here: rocprofv3's preloaded tool maps /opt/rocm's copy before
    #  python starts and torch then brings its own -- profiled runs have always worked that way; what the library itself binds, and
    #  what breaks when doubled, are the HIP runtime and RCCL)
    dup = {k: v for k, v in mapped_runtime_libraries().items() if len(v) > 1 and k != "libhsa-runtime64"}
    if dup:
        raise HipLibraryError(
            "two copies of the GPU runtime are mapped into this process: %s.  libcomposer_hip.so was loaded before torch was "
            "imported, so it is bound to /opt/rocm's runtime while torch brought its own.  Fix: `import torch` BEFORE the first "
            "composer_amd call that loads the library (composer_amd._lib.load(), Transformer(...)), or do not import torch in "
            "this process, or put torch's lib directory first in LD_LIBRARY_PATH so both resolve to the same files."
            % "; ".join("%s -> %s" % (k, " AND ".join(v)) for k, v in sorted(dup.items())))


def runtime_info():
    """What this process is bound to: the mapped runtime libraries and RCCL's version (for bench.py's multi-GPU line)."""
    v = C.c_int(0)
    rc = load().cmp_dp_rccl_version(C.byref(v))
    libs = mapped_runtime_libraries()
    return {"rccl_version": v.value if rc == 0 else None, "rccl_path": (libs.get("librccl") or [None])[0],
            "hip_runtime_path": (libs.get("libamdhip64") or [None])[0]}


def last_error():
    return load().cmp_last_error().decode("utf-8", "replace")


def check(rc, what=""):
    if rc != 0:
        raise HipLibraryError("%s failed (status %d): %s" % (what or "libcomposer_hip call", rc, last_error()))


def require_gpu():
    check_single_runtime()          # (again: torch may have been imported since load())
    n = load().cmp_device_count()
    if n <= 0:
        raise HipLibraryError("no HIP device visible: composer_amd needs an MI355X (gfx950); there is no CPU fallback")
    return n
