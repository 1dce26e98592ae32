"""ctypes binding of libpgasr_hip.so (the C ABI declared in include/pgasr_hip.h).

There is NO fallback: if the shared library is missing or a call returns a non-zero
status this module raises.  Nothing here imports ``oracle``.
"""
import ctypes as C
import os

# Import order matters: PyTorch bundles its own HIP runtime (torch/lib/libamdhip64.so).  Loading this
# library first would pull in the system ROCm runtime as well, and kernels launched through one
# runtime on memory/streams owned by the other fail.  With torch loaded first the library's
# libamdhip64.so.7 dependency resolves to the runtime torch already mapped: ONE runtime per process.
import torch  # noqa: F401  (must precede CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PGASR_HIP_LIB", os.path.join(_HERE, "libpgasr_hip.so"))   # override: diagnostic builds

c_f32p = C.c_void_p   # device pointers travel as integers (tensor.data_ptr())
c_i32p = C.c_void_p
c_ptr = C.c_void_p

# name -> (restype, argtypes).  Mirrors include/pgasr_hip.h one to one.
SIGNATURES = {
    "pgasr_abi_version": (C.c_int, []),
    "pgasr_status_string": (C.c_char_p, [C.c_int]),
    "pgasr_ctc_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "pgasr_ctc_loss_grad": (C.c_int, [c_f32p, c_i32p, c_i32p, c_i32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, c_f32p, c_f32p, c_i32p, c_f32p, c_f32p, c_ptr, C.c_size_t, c_ptr]),
    "pgasr_ctc_grad_from_lattice": (C.c_int, [c_f32p, c_i32p, c_i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              c_f32p, c_f32p, c_i32p, C.c_int, c_f32p, c_ptr, C.c_size_t, c_ptr]),
    "pgasr_pg_rewards": (C.c_int, [c_i32p, c_i32p, C.c_int, C.c_float, C.c_float, c_f32p, c_f32p, c_f32p, c_f32p, c_ptr]),
    "pgasr_pg_loss_value": (C.c_int, [c_f32p, c_i32p, c_i32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      c_f32p, c_ptr]),
    "pgasr_head_logsoftmax": (C.c_int, [c_f32p, C.c_longlong, C.c_int, C.c_int, c_f32p, c_f32p, C.c_int, c_f32p, c_f32p, c_ptr]),
    "pgasr_pg_step_coefs": (C.c_int, [c_i32p, c_i32p, c_i32p, C.c_int, c_i32p, c_i32p, C.c_int, C.c_int, C.c_int,
                                      C.c_float, C.c_float, c_f32p, c_ptr]),
    "pgasr_frame_argmax_sample": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_int, C.c_int,
                                            c_i32p, c_i32p, c_ptr]),
    "pgasr_batch_prep": (C.c_int, [c_f32p, C.c_int, C.c_int, c_ptr, c_ptr, C.c_int, c_i32p, c_i32p, c_i32p, c_ptr]),
    "pgasr_ctc_collapse": (C.c_int, [c_i32p, c_i32p, C.c_int, C.c_int, C.c_int, C.c_int, c_i32p, c_i32p, c_ptr]),
    "pgasr_edit_distance": (C.c_int, [c_i32p, c_i32p, C.c_int, c_i32p, c_i32p, C.c_int, C.c_int,
                                      c_i32p, c_i32p, c_ptr]),
    "pgasr_reinforce_grad": (C.c_int, [c_f32p, c_i32p, c_f32p, c_i32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       c_f32p, c_ptr]),
    "pgasr_beam_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "pgasr_ctc_beam_search": (C.c_int, [c_ptr, C.c_int, C.c_longlong, C.c_longlong, c_i32p, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, c_i32p, c_i32p, c_ptr, c_ptr, C.c_size_t, c_ptr]),
    "pgasr_dropout": (C.c_int, [c_f32p, c_f32p, C.c_ulonglong, C.c_float, C.c_uint64, C.c_uint32, c_f32p, C.c_float, c_ptr]),
    "pgasr_stream_copy": (C.c_int, [c_ptr, c_ptr, C.c_ulonglong, C.c_int, c_ptr]),
    "pgasr_adam_step": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_ulonglong, C.c_int, C.c_float, C.c_float,
                                  C.c_float, C.c_float, C.c_float, c_i32p, c_i32p, c_i32p, c_ptr]),
    "pgasr_gemm_workspace_bytes": (C.c_size_t, [C.c_int] * 5),
    "pgasr_gemm_f32": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                 c_f32p, C.c_int, C.c_longlong, c_f32p, C.c_int, C.c_longlong,
                                 c_f32p, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int,
                                 c_f32p, c_f32p, C.c_int, C.c_float, C.c_int,
                                 c_f32p, C.c_int, c_f32p, c_f32p, C.c_int, c_ptr, c_ptr, C.c_size_t, c_ptr]),
    "pgasr_split_bf16_planes": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr, c_ptr, c_ptr]),
    "pgasr_gemm_x3w_f32": (C.c_int, [C.c_int, C.c_int, C.c_int, c_f32p, C.c_int, c_ptr, c_ptr, c_f32p, C.c_int,
                                     c_f32p, c_f32p, C.c_float, c_ptr]),
    "pgasr_gemm_x3w_feed_workspace_bytes": (C.c_size_t, []),
    "pgasr_gemm_x3w_feed_col_tiles": (C.c_int, [C.c_int]),
    "pgasr_gemm_x3w_feed_head_items": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "pgasr_gemm_x3w_feed_f32": (C.c_int, [C.c_int, C.c_int, C.c_int, c_f32p, C.c_int, c_ptr, c_ptr, c_f32p, C.c_int,
                                          c_f32p, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, c_ptr, C.c_size_t, c_ptr]),
    "pgasr_feat_frames": (C.c_int, [c_f32p, c_i32p, c_i32p, C.c_int, C.c_longlong, C.c_int, c_f32p, c_ptr]),
    "pgasr_feat_power": (C.c_int, [c_f32p, C.c_longlong, c_f32p, c_ptr]),
    "pgasr_feat_db": (C.c_int, [c_f32p, c_i32p, C.c_int, C.c_int, C.c_int, C.c_float, c_ptr]),
    "pgasr_feat_deltas_stack": (C.c_int, [c_f32p, c_i32p, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, c_ptr]),
    "pgasr_feat_stack": (C.c_int, [c_f32p, c_i32p, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, c_ptr]),
    "pgasr_colsum_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "pgasr_colsum_f32": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, C.c_int,
                                   c_ptr, C.c_size_t, c_ptr]),
    "pgasr_instnorm_stats": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_float, c_f32p, c_f32p, c_ptr]),
    "pgasr_log_softmax_rows": (C.c_int, [c_f32p, C.c_longlong, C.c_int, c_f32p, c_ptr]),
    "pgasr_lstm_pack_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "pgasr_lstm_pack_weights": (C.c_int, [c_f32p] * 8 + [C.c_int, c_f32p, c_f32p, c_ptr, c_ptr, C.c_int, c_ptr]),
    "pgasr_lstm_unpack_grads": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int] + [c_f32p] * 8 + [C.c_int, c_ptr]),
    "pgasr_lstm_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "pgasr_lstm_error_offset": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "pgasr_lstm_busy_offset": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "pgasr_lstm_status": (C.c_int, [c_ptr, C.c_size_t, C.c_int, C.c_int, c_ptr]),
    "pgasr_stream_gate": (C.c_int, [c_ptr, C.c_int, C.c_int, c_ptr]),
    "pgasr_stream_gate_sum": (C.c_int, [c_ptr, C.c_int, C.c_int, c_ptr, C.c_int, c_ptr]),
    "pgasr_stream_gate_report": (C.c_int, [c_ptr, C.c_int, C.c_int, c_ptr, C.c_int, c_ptr, c_ptr]),
    "pgasr_stream_probe": (C.c_int, [c_ptr, C.c_int, c_ptr]),
    "pgasr_lstm_layer_fwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_ptr, c_i32p, C.c_int, C.c_int, C.c_int,
                                       c_f32p, C.c_float, C.c_uint64, C.c_uint32, c_ptr, C.c_size_t, c_ptr]),
    "pgasr_lstm_layer_fwd_fed": (C.c_int, [c_f32p, c_f32p, c_f32p, c_ptr, c_i32p, C.c_int, C.c_int, C.c_int, c_ptr, C.c_int,
                                           c_f32p, C.c_float, C.c_uint64, C.c_uint32, c_ptr, C.c_size_t, c_ptr]),
    "pgasr_lstm_fed_ok": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "pgasr_lstm_layer_bwd_fed": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_ptr, c_i32p, C.c_int, C.c_int, C.c_int,
                                           c_f32p, c_ptr, C.c_int, C.c_float, C.c_uint64, C.c_uint32, c_ptr, C.c_size_t, c_ptr]),
    "pgasr_lstm_layer_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_ptr, c_i32p, C.c_int, C.c_int, C.c_int,
                                       c_f32p, c_ptr, C.c_size_t, c_ptr]),
    "pgasr_lstm_wgrad_slabs": (C.c_int, [C.c_int, c_i32p, C.c_int]),
    "pgasr_lstm_layer_bwd_streamed": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_ptr, c_i32p, C.c_int, C.c_int, C.c_int,
                                                c_f32p, c_ptr, C.c_int, C.c_float, C.c_uint64, C.c_uint32, c_ptr,
                                                c_ptr, C.c_size_t, c_ptr]),
    "pgasr_lstm_wgrads_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "pgasr_lstm_wgrads_streamed": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, c_ptr,
                                             c_ptr, c_ptr, C.c_int, c_ptr, C.c_size_t, c_ptr]),
    "pgasr_error_flag": (C.c_int, [c_i32p, c_i32p, c_f32p, c_ptr]),
    "pgasr_attention_ctx": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_ptr]),
    "pgasr_lstm_cell_f32": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, c_ptr]),
    "pgasr_pack_x6w_planes": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr, c_ptr]),
    "pgasr_split_bf16_planes3": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr, c_ptr, c_ptr, c_ptr]),
    "pgasr_gemm_x6w_f32": (C.c_int, [C.c_int, C.c_int, C.c_int, c_f32p, C.c_int, c_ptr, c_ptr, c_ptr, c_f32p, C.c_int,
                                     c_f32p, c_f32p, C.c_float, c_ptr]),
    "pgasr_gemm_x6w_feed_workspace_bytes": (C.c_size_t, []),
    "pgasr_gemm_x6w_feed_col_tiles": (C.c_int, [C.c_int]),
    "pgasr_gemm_x6w_feed_f32": (C.c_int, [C.c_int, C.c_int, C.c_int, c_f32p, C.c_int, c_ptr, c_ptr, c_ptr, c_f32p, C.c_int,
                                          c_f32p, c_ptr, c_ptr, C.c_int, c_ptr, C.c_size_t, c_ptr]),
    "pgasr_gemm_x6w_feed_head_items": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "pgasr_gemm_x6w_feed_phase_f32": (C.c_int, [C.c_int, C.c_int, C.c_int, c_f32p, C.c_int, c_ptr, c_ptr, c_ptr, c_f32p, C.c_int,
                                                c_f32p, c_ptr, c_ptr, C.c_int, C.c_int, c_ptr, c_ptr, C.c_size_t, c_ptr]),
}


class PgasrError(RuntimeError):
    pass


TIMEOUT = 5     # PGASR_ERR_TIMEOUT


_lib = None


def load():
    """Load the library once; raise (never fall back) if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PgasrError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C policy_gradient_asr_amd/csrc`.  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.pgasr_abi_version() != 7:
        raise PgasrError("libpgasr_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().pgasr_status_string(status).decode()
        raise PgasrError(f"{what} failed: status {status} ({msg})")
