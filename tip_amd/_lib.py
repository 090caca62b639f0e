"""ctypes binding of libtipk.so -- every symbol `include/tipk.h` declares, nothing else.

The library is built in-tree (`tip_amd/libtipk.so`, see `__graft_entry__.build()` /
`tip_amd/csrc/Makefile`).  There is NO fallback: if the shared object is missing or a call returns
a non-zero status, an exception is raised -- the product path never computes on the CPU.
"""
import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libtipk.so')
CSRC = os.path.join(_HERE, 'csrc')

ABI_VERSION = 3


class TipkError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    """struct tipk_gemm_desc (include/tipk.h)."""
    _fields_ = [('m', C.c_int64), ('n', C.c_int64), ('k', C.c_int64),
                ('batch', C.c_int64), ('kbatch', C.c_int64), ('ksplit', C.c_int64),
                ('a', C.c_void_p), ('a_sm', C.c_int64), ('a_sk', C.c_int64), ('a_sq', C.c_int64), ('a_sz', C.c_int64),
                ('b', C.c_void_p), ('b_sk', C.c_int64), ('b_sn', C.c_int64), ('b_sq', C.c_int64), ('b_sz', C.c_int64),
                ('c', C.c_void_p), ('c_sm', C.c_int64), ('c_sz', C.c_int64), ('c_ss', C.c_int64),
                ('c_in', C.c_void_p), ('cin_sm', C.c_int64), ('cin_sz', C.c_int64),
                ('alpha', C.c_float), ('relu', C.c_int)]


class SlabSumDesc(C.Structure):
    """struct tipk_slab_sum_desc (include/tipk.h)."""
    _fields_ = [('in_', C.c_void_p), ('n_slabs', C.c_int64), ('slab_stride', C.c_int64), ('count', C.c_int64),
                ('alpha', C.c_float), ('accumulate', C.c_int),
                ('row_scale', C.c_void_p), ('cols', C.c_int64),
                ('addend', C.c_void_p), ('relu', C.c_int),
                ('gate', C.c_void_p),
                ('out', C.c_void_p)]


GROUP_MAX = 6                                  # TIPK_GROUP_MAX

_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); must list exactly the functions of include/tipk.h
SIGNATURES = {
    'tipk_abi_version': (_I, []),
    'tipk_strerror': (C.c_char_p, [_I]),
    'tipk_device_info': (_I, [_I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.c_char_p, _I]),
    'tipk_gather_sum': (_I, [_P, _L, _P, _P, _P, _L, _P, _L, _P, _P, _P, _I, _I, _I, _P]),
    'tipk_gather_sum_finalize': (_I, [_P, _P, _L, _P, _L, _P, _P, _I, _I, _I, _P]),
    'tipk_rel_gather_supported': (_I, [_L, _I, _I]),
    'tipk_rel_gather': (_I, [_I, _P, _L, _L, _I, _L, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    'tipk_gemm_f32': (_I, [C.POINTER(GemmDesc), _P]),
    'tipk_gemm_f32_group': (_I, [C.POINTER(GemmDesc), C.c_int32, _P]),
    'tipk_sum_slabs_group': (_I, [C.POINTER(SlabSumDesc), C.c_int32, _P]),
    'tipk_rgcn_dy_products_plan': (_I, [_L, _L, _I, C.POINTER(_I), C.POINTER(_I)]),
    'tipk_rgcn_dy_products': (_I, [_P, _L, _P, _L, _P, _L, _L, _L, _I, _P, _P, _P]),
    'tipk_sum_slabs': (_I, [_P, _L, _L, _L, _F, _I, _P, _P]),
    'tipk_sum_slabs_ex': (_I, [_P, _L, _L, _L, _F, _I, _P, _L, _P, _I, _P, _P]),
    'tipk_transpose': (_I, [_P, _L, _L, _P, _P]),
    'tipk_rows_affine': (_I, [_P, _L, _P, _P, _P, _L, _P, _L, _L, _L, _I, _P]),
    'tipk_gate_colsum_groups': (_I, [_L, _L]),
    'tipk_gate_colsum': (_I, [_P, _L, _P, _L, _P, _L, _L, _L, _P, _P]),
    'tipk_col_sum': (_I, [_P, _L, _L, _L, _P, _P, _P]),
    'tipk_distmult_fwd': (_I, [_P, _L, _I, _P, _L, _P, _P, _I, _P, _I, _L, _I, _P, _P]),
    'tipk_distmult_bwd': (_I, [_P, _P, _P, _L, _I, _P, _L, _P, _P, _I, _P, _I, _L, _I, _P, _L, _P, _P, _P]),
    'tipk_distmult_loss': (_I, [_P, _L, _I, _P, _L, _P, _P, _P, _P, _I, _P, _I, _L, _P, _L, _P, _P, _P, _P]),
    'tipk_pair_table_fwd': (_I, [_P, _P, _L, _P, _P, _I, _P, _I, _L, _I, _P, _P]),
    'tipk_pair_table_bwd': (_I, [_P, _P, _L, _P, _P, _I, _P, _I, _L, _I, _P, _P, _P]),
    'tipk_typed_negative_sampling': (_I, [_P, _P, _L, _L, C.c_uint64, _P, _P, _P, _L, _P, _P, _I, _L, _P]),
    'tipk_counter_advance': (_I, [_P, _P]),
    'tipk_rank_metrics': (_I, [_P, _P, _P, _L, _L, _P, _P]),
}

_lib = None


def build(verbose=False):
    """Compile libtipk.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    out = subprocess.run(['make', '-C', CSRC, '-j4'], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
        print(out.stderr)
    if out.returncode != 0:
        raise TipkError('building libtipk.so failed (see output above)')
    return LIB_PATH


def lib():
    """The loaded library (cached).  Raises if it has not been built -- no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TipkError('%s is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                        '(or `make -C tip_amd/csrc`); tip_amd has no CPU fallback' % LIB_PATH)
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if handle.tipk_abi_version() != ABI_VERSION:
        raise TipkError('libtipk.so ABI %d != binding ABI %d: rebuild' % (handle.tipk_abi_version(), ABI_VERSION))
    _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().tipk_strerror(status).decode()
        raise TipkError('%s failed: %s (status %d)' % (what, msg, status))


def stream_ptr(device=None):
    """hipStream_t of torch's CURRENT stream on `device` (so torch ops and tipk kernels order
    correctly and `torch.cuda.graph` capture sees our launches)."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise TipkError('tip_amd kernels need device tensors (got %s); there is no CPU path' % t.device)
