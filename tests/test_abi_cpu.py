"""The C-ABI library loads on a GPU-less host and exports every symbol that
include/pgasr_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pgasr_hip.h")
LIB = os.path.join(ROOT, "policy_gradient_asr_amd", "libpgasr_hip.so")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pgasr_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(LIB)


def test_header_declares_something():
    syms = declared_symbols()
    assert "pgasr_ctc_loss_grad" in syms and len(syms) >= 8


def test_every_declared_symbol_is_exported(lib):
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_python_signature_table_matches_header(lib):
    from policy_gradient_asr_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    # argument counts agree with the header prototypes
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^)]*)\)" % name, src)
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(args), (name, n, len(args))


def test_abi_version_and_status_strings(lib):
    lib.pgasr_abi_version.restype = ctypes.c_int
    assert lib.pgasr_abi_version() == 7
    lib.pgasr_status_string.restype = ctypes.c_char_p
    assert lib.pgasr_status_string(0) == b"ok"
    assert b"workspace" in lib.pgasr_status_string(3)
    assert b"timed out" in lib.pgasr_status_string(5)


def test_workspace_query_needs_no_gpu(lib):
    lib.pgasr_ctc_workspace_bytes.restype = ctypes.c_size_t
    n = lib.pgasr_ctc_workspace_bytes(1000, 32, 29, 100)
    assert n >= 2 * 1000 * 32 * 201 * 4      # alpha, beta as fp32 offsets (+ fp64 row maxima)
    assert lib.pgasr_ctc_workspace_bytes(0, 32, 29, 100) == 0


def test_weight_gradient_time_slabs_need_no_gpu(lib):
    """pgasr_lstm_wgrad_slabs: the frame boundaries that DEFINE the summation order of a layer's weight-gradient products (and
    the publication points of a streamed backward sweep) are a host-side function of T alone: 0 = h_0 < .. < h_n = T, sizes
    16, 24, 32, 40, 48, 64, 80, 104, 128, 128, .. growing away from frame 0 (x 5/4, cap 128: round 5, sized for the six-product kernel),
    the last slab at most 1.5 sizes."""
    lib.pgasr_lstm_wgrad_slabs.restype = ctypes.c_int
    lib.pgasr_lstm_wgrads_workspace_bytes.restype = ctypes.c_size_t
    buf = (ctypes.c_int * 128)()

    def edges(T):
        n = lib.pgasr_lstm_wgrad_slabs(T, buf, 128)
        assert lib.pgasr_lstm_wgrad_slabs(T, None, 0) == n       # the count alone
        return [buf[i] for i in range(n + 1)]
    assert edges(1000) == [0, 16, 40, 72, 112, 160, 224, 304, 408, 536, 664, 792, 920, 1000]
    assert edges(24) == [0, 24] and edges(25) == [0, 16, 25] and edges(1) == [0, 1]
    assert lib.pgasr_lstm_wgrad_slabs(0, buf, 128) == 0
    sizes = [16, 24, 32, 40, 48, 64, 80, 104, 128]
    for T in list(range(1, 400)) + [777, 1000, 1500, 4096, 8191]:
        e = edges(T)
        assert e[0] == 0 and e[-1] == T and all(a < b for a, b in zip(e, e[1:])) and len(e) - 1 <= 72
        d = [b - a for a, b in zip(e, e[1:])]
        want = [sizes[min(i, 8)] for i in range(len(d) - 1)]
        assert d[:-1] == want
        nxt = sizes[min(len(d) - 1, 8)]
        assert 0 < d[-1] <= nxt + nxt // 2
    # workspace of the two products: one partial slab set per time slab
    assert lib.pgasr_lstm_wgrads_workspace_bytes(1000, 512) == 256 + 13 * (2048 * 512 + 2 * 1024 * 256) * 4


def test_product_path_refuses_cpu_tensors():
    import torch
    from policy_gradient_asr_amd import hipops, _lib
    with pytest.raises(_lib.PgasrError):
        hipops.frame_argmax_sample(torch.zeros(2, 2, 4))


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "policy_gradient_asr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f


def test_host_side_shape_rules():
    """Pure-Python shape predicates of the host layer (no GPU call): the six-product kernels address whole 256-row tiles with 32-bit
    offsets, so the 4 GiB bound is on the PADDED row count; streamed weight gradients take B % 16 in the six-product
    arithmetic and B % 32 in bf16x3; the library's default precision mode is the reference's arithmetic."""
    from policy_gradient_asr_amd import hipops
    assert hipops.get_precision() == "f32" and hipops.PRECISION_MODES[hipops.get_precision()] == (hipops.GEMM_PRECISION, hipops.LSTM_PLANES)
    assert hipops.gemm_x3w_ok(2047 * 256, 2048, 512, planes=3)                  # 2047 tiles x 2048 x 4 B < 4 GiB
    assert not hipops.gemm_x3w_ok(2047 * 256 + 1, 2048, 512, planes=3)          # M itself is below the bound, its last tile's rows are not
    assert not hipops.gemm_x3w_ok(524287, 2048, 512, planes=3) and hipops.gemm_x3w_ok(524287, 256, 512, planes=3)
    assert hipops.lstm_wgrads_ok(200, 16, 512, 3) and not hipops.lstm_wgrads_ok(200, 16, 512, 2)
    assert hipops.lstm_wgrads_ok(200, 32, 512, 2) and not hipops.lstm_wgrads_ok(200, 24, 512, 3) and not hipops.lstm_wgrads_ok(1, 32, 512, 3)
