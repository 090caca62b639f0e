"""ctypes binding of libtipk.so -- every symbol `include/tipk.h` declares, nothing else.

The library is built in-tree (`tip_amd/libtipk.so`, see `__graft_entry__.build()` /
`tip_amd/csrc/Makefile`).  There is NO fallback: if the shared object is missing or a call returns
a non-zero status, an exception is raised -- the product path never computes on the CPU.
"""
import ctypes as C
import glob
import hashlib
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libtipk.so')
CSRC = os.path.join(_HERE, 'csrc')

ABI_VERSION = 23


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


class WgGemmDesc(C.Structure):
    """struct tipk_wg_gemm_desc (include/tipk.h)."""
    _fields_ = [('p', GemmDesc),
                ('a2', C.c_void_p), ('a2_sm', C.c_int64), ('a2_sk', C.c_int64),
                ('b2', C.c_void_p), ('b2_sk', C.c_int64), ('b2_sn', C.c_int64),
                ('k2', C.c_int64),
                ('gate', C.c_void_p), ('gate_sm', C.c_int64), ('gate_sz', C.c_int64)]


GROUP_MAX = 6                                  # TIPK_GROUP_MAX
WG_GEMM_MAX, WG_SUMS_MAX = 4, 3                # TIPK_WG_GEMM_MAX, TIPK_WG_SUMS_MAX

_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); must list exactly the functions of include/tipk.h
SIGNATURES = {
    'tipk_abi_version': (_I, []),
    'tipk_strerror': (C.c_char_p, [_I]),
    'tipk_build_id': (C.c_char_p, []),
    'tipk_set_option': (_I, [C.c_char_p, _I]),
    'tipk_get_option': (_I, [C.c_char_p, C.POINTER(_I)]),
    'tipk_device_info': (_I, [_I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.c_char_p, _I]),
    'tipk_gather_sum': (_I, [_P, _L, _L, _P, _P, _P, _L, _P, _L, _P, _P, _P, _I, _I, _I, _P]),
    'tipk_gather_sum_finalize': (_I, [_P, _P, _L, _P, _L, _P, _P, _I, _I, _I, _P]),
    'tipk_gather_sum_riders_supported': (_I, [_I, _I]),
    'tipk_gather_sum_riders': (_I, [_P, _L, _L, _P, _P, _P, _L, _P, _L, _P, _P, _I, _I, _I, _P, _L, _P, C.POINTER(SlabSumDesc), C.c_int32, _P]),
    'tipk_gather_sum_lin_supported': (_I, [_I, _I, _I]),
    'tipk_gather_sum_lin': (_I, [_P, _L, _L, _P, _P, _P, _L, _P, _L, _P, _P, _L, _L, _P, _I, _P, _L, _I, _I, _I, _P]),
    'tipk_gather_rows_csr': (_I, [_P, _L, _L, _P, _P, _L, _P, _L, _I, _P]),
    'tipk_rel_gather_supported': (_I, [_L, _I, _I]),
    'tipk_rel_gather_occupancy': (_I, [_L, _I, _I]),
    'tipk_rel_gather_chunk': (_I, [_L, _I, _I]),
    'tipk_pair_product_supported': (_I, [_I, _I]),
    'tipk_pair_product': (_I, [_P, _P, _L, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    'tipk_rgcn_pair_grads_supported': (_I, [_I, _I]),
    'tipk_rgcn_pair_grads': (_I, [_P, _L, _P, _P, _L, _L, _I, _I, _P, _P, _P, _L, _P, _L, _L, _P, _L, _P]),
    'tipk_stream_gather_parts': (_I, [_P, _L, _I, _L, _P, _L, _P, _L, _P, _P, _P, _I, _P, _P, _P, _L, _P]),
    'tipk_stream_gather_parts_two': (_I, [_P, _P, _L, _I, _L, _P, _L, _P, _L, _P, _P, _P, _I, _P, _P, _P, _P, _L, _P]),
    'tipk_rgcn_dest_products_supported': (_I, [_L, _L, _I, _I]),
    'tipk_rgcn_dest_products': (_I, [_P, _L, _I, _P, _L, _I, _L, _L, _P, _P, _P, _L, _L, _P]),
    'tipk_rgcn_row_products_supported': (_I, [_L, _L, _I, _I]),
    'tipk_rgcn_row_products_slabs': (_L, [_L, _I]),
    'tipk_rgcn_row_products': (_I, [_P, _L, _L, _I, _P, _L, _L, _I, _P, _P, _P, _L, _P, _P, _P]),
    'tipk_rgcn_row_products_s_supported': (_I, [_L, _L, _I, _I]),
    'tipk_rgcn_row_products_s': (_I, [_P, _L, _L, _I, _P, _L, _L, _I, _P, _P, _P, _L, _P, _P, _P]),
    'tipk_sum_slabs_xb': (_I, [_P, _L, _L, _L, _I, _P, _P, _I, _P, _P, _P, _I, _I, _P, _P, _P]),
    'tipk_stream_gather_supported': (_I, [_L, _I, _I]),
    'tipk_stream_gather_piece': (_I, []),
    'tipk_stream_gather_two': (_I, [_P, _P, _L, _L, _I, _L, _P, _P, _P, _I, _P, _P, _L, _P]),
    'tipk_stream_gather': (_I, [_P, _L, _L, _I, _L, _P, _P, _P, _I, _P, _P, _P, _P, _L, _I, _I, _P, _P, _I, _P]),
    'tipk_rel_gather': (_I, [_I, _P, _L, _L, _I, _L, _P, _P, _P, _I, _P, _P, _P, _P, _L, _P]),
    'tipk_gemm_f32': (_I, [C.POINTER(GemmDesc), _P]),
    'tipk_gemm_f32_group': (_I, [C.POINTER(GemmDesc), C.c_int32, _P]),
    'tipk_sum_slabs_group': (_I, [C.POINTER(SlabSumDesc), C.c_int32, _P]),
    'tipk_gemm_wg_group_supported': (_I, [C.POINTER(WgGemmDesc)]),
    'tipk_gemm_wg_group': (_I, [C.POINTER(WgGemmDesc), C.c_int32, C.POINTER(SlabSumDesc), C.c_int32, _P]),
    'tipk_rgcn_dy_products_plan': (_I, [_L, _L, _I, C.POINTER(_I), C.POINTER(_I)]),
    'tipk_rgcn_dy_products': (_I, [_P, _L, _P, _L, _P, _L, _L, _L, _I, _P, _L, _P, _P, _P]),
    'tipk_rgcn_node_products_plan': (_I, [_L, _I, _L, _I, C.POINTER(_I)]),
    'tipk_rgcn_node_products': (_I, [_P, _L, _I, _P, _P, _P, _L, _L, _P, _L, _I, _P, _L, _L, _P, _P, _L, _L, _P, _P]),
    'tipk_sum_slabs': (_I, [_P, _L, _L, _L, _F, _I, _P, _P]),
    'tipk_sum_slabs_ex': (_I, [_P, _L, _L, _L, _F, _I, _P, _L, _P, _I, _P, _P]),
    'tipk_transpose': (_I, [_P, _L, _L, _P, _P]),
    'tipk_rows_affine': (_I, [_P, _L, _P, _P, _P, _L, _P, _L, _L, _L, _I, _P]),
    'tipk_gate_colsum_groups': (_I, [_L, _L]),
    'tipk_gate_colsum': (_I, [_P, _L, _P, _L, _P, _L, _L, _L, _P, _P]),
    'tipk_drug_mix_fwd': (_I, [_P, _L, _P, _P, _L, _P, _I, _I, _L, _I, _I, _P, _L, _P]),
    'tipk_drug_mix_gather_supported': (_I, [_I, _I]),
    'tipk_drug_mix_gather_fwd': (_I, [_P, _L, _P, _P, _L, _P, _P, _P, _P, _P, _L, _P, _I, _I, _L, _I, _I, _P, _L, _P, _P]),
    'tipk_drug_mix_bwd': (_I, [_P, _L, _P, _P, _P, _I, _I, _L, _I, _I, _P, _L, _P, _P, _P]),
    'tipk_drug_mix_gather_xb_supported': (_I, [_I, _I, _I, _I, _I, _I]),
    'tipk_drug_mix_gather_xb_fwd': (_I, [_P, _L, _P, _P, _L, _P, _P, _P, _P, _P, _L, _P, _I, _I, _L, _I, _I, _P, _L, _P, _P, _P, _I, _I, _P, _P, _P]),
    'tipk_pd_stage_bwd_supported': (_I, [_I, _I, _L, _I]),
    'tipk_pd_stage_bwd_limits': (_I, [C.POINTER(_I), C.POINTER(_I)]),
    'tipk_pd_stage_bwd_wh_slabs': (_I, []),
    'tipk_pd_stage_bwd': (_I, [_P, _L, _P, _P, _P, _I, _I, _L, _I, _I, _P, _L, _P, _P, _P, _P, _L, _P, _L, _P, _L, _I, _P, _L, _L, _P, _P, _L, _P, _P, _P]),
    'tipk_col_sum': (_I, [_P, _L, _L, _L, _P, _P, _P]),
    'tipk_distmult_fwd': (_I, [_P, _L, _I, _P, _L, _P, _P, _I, _P, _I, _L, _I, _P, _P]),
    'tipk_distmult_bwd': (_I, [_P, _P, _P, _L, _I, _P, _L, _P, _P, _I, _P, _I, _L, _I, _P, _L, _P, _P, _P]),
    'tipk_distmult_workspace_bytes': (_L, [_L, _I, _L]),
    'tipk_distmult_loss': (_I, [_P, _L, _I, _P, _L, _P, _P, _P, _P, _I, _P, _I, _L, _P, _L, _P, _P, _P, _P, _P]),
    'tipk_distmult_loss_store': (_I, [_P, _L, _I, _P, _L, _P, _P, _P, _P, _I, _P, _I, _L, _P, _L, _P, _P, _P, _P, _P]),
    'tipk_pair_table_fwd': (_I, [_P, _P, _L, _P, _P, _I, _P, _I, _L, _I, _P, _P]),
    'tipk_pair_table_bwd': (_I, [_P, _P, _L, _P, _P, _I, _P, _I, _L, _I, _P, _P, _P]),
    'tipk_pair_table_loss': (_I, [_P, _P, _L, _L, _L, _P, _P, _P, _P, _L, _F, _P, _P, _P, _P]),
    'tipk_negsample_wgs_per_cu': (_I, [_L]),
    'tipk_typed_negative_sampling': (_I, [_P, _P, _L, _L, C.c_uint64, _P, _I, _P, _P, _L, _P, _P, _P, _P, _I, _L, _P]),
    'tipk_counter_advance': (_I, [_P, _P]),
    'tipk_rank_metrics': (_I, [_P, _P, _P, _L, _L, _P, _P]),
    'tipk_peer_mailbox_bytes': (_L, [_I, _L]),
    'tipk_peer_alloc': (_I, [_L, C.POINTER(_P)]),
    'tipk_peer_free': (_I, [_P]),
    'tipk_ipc_get_handle': (_I, [_P, _P]),
    'tipk_ipc_open': (_I, [_P, C.POINTER(_P)]),
    'tipk_ipc_close': (_I, [_P]),
    'tipk_peer_allreduce': (_I, [_P, _L, C.POINTER(_P), _I, _I, _L, _P]),
    'tipk_peer_set_timeout_ms': (_I, [_L]),
    'tipk_peer_status': (_I, [_P, _I, _L, C.POINTER(C.c_uint64)]),
    'tipk_adam_step': (_I, [_I, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_L), C.POINTER(_P), _P,
                       C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _P]),
    'tipk_graph_build': (_I, [_P, _P, _P, _I, _L, _L, _L, _P, C.POINTER(C.c_void_p)]),
    'tipk_graph_destroy': (_I, [_P]),
    'tipk_graph_info': (_I, [_P, C.POINTER(_L), C.POINTER(_L), C.POINTER(_L), C.POINTER(C.c_void_p)]),
    'tipk_rgcn_workspace_bytes': (_L, [_P, _I, _I, _I]),
    'tipk_rgcn_fwd': (_I, [_P, _P, _L, _I, _P, _P, _P, _I, _I, _I, _P, _L, _P, _L, _P]),
    'tipk_rgcn_bwd': (_I, [_P, _P, _L, _I, _P, _P, _P, _I, _I, _P, _L, _P, _L, _P, _L, _P, _P, _P, _P, _L, _P]),
    'tipk_gcn_graph_build': (_I, [_P, _I, _L, _L, C.POINTER(C.c_void_p)]),
    'tipk_gcn_workspace_bytes': (_L, [_P, _I, _I]),
    'tipk_gcn_fwd': (_I, [_P, _P, _L, _I, _P, _L, _L, _P, _I, _I, _P, _L, _P, _L, _P]),
    'tipk_gcn_bwd': (_I, [_P, _P, _L, _I, _P, _L, _L, _I, _P, _L, _P, _L, _P, _L, _P, _L, _L, _P, _P, _L, _P]),
    'tipk_hier_graph_build': (_I, [_P, _I, _L, _L, _L, C.POINTER(C.c_void_p)]),
    'tipk_hier_workspace_bytes': (_L, [_P, _I, _I]),
    'tipk_hier_fwd': (_I, [_P, _P, _L, _I, _P, _I, _P, _L, _P, _L, _P]),
    'tipk_hier_bwd': (_I, [_P, _P, _L, _I, _P, _I, _P, _L, _P, _L, _P, _P, _L, _P]),
    'tipk_rgcn_bwd_ex': (_I, [_P, _P, _L, _I, _P, _P, _P, _I, _I, _P, _L, _P, _L, _P, _L, _P, _P, _P, _P, _L, _I, _P]),
    'tipk_graph_prepare_rgcn': (_I, [_P, _I, _I]),
    'tipk_graph_rgcn_route': (_I, [_P, _I, _I]),
    'tipk_graph_release_host': (_I, [_P]),
    'tipk_plan_stream_rows': (_I, [_P, _P, _L, _L, _L, _I, _I, _I, _I, _I, _P]),
    'tipk_plan_pair_bwd': (_I, [_P, _P, _P, _L, _L, _L, _P, _I, _I, _I, _I, _P]),
    'tipk_plan_link_words': (_I, [_P, _P, _L, _L, _P]),
    'tipk_plan_gather': (_I, [_P, _P, _P, _L, _L, _L, _I, _I, _P]),
    'tipk_host_plan_array': (_I, [_P, C.c_char_p, _P, _P, _P]),
    'tipk_host_plan_scalar': (_L, [_P, C.c_char_p]),
    'tipk_host_plan_free': (None, [_P]),
    'tipk_split_flags': (_I, [_P, _L, _L, C.c_double, C.c_uint64, _P, _P, _P]),
    'tipk_split_scatter': (_I, [_P, _P, _I, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
}

_lib = None

# environment switch -> library option (translated ONCE, when the library is loaded; the library
# itself never reads the environment).  Tests and tools flip options with `set_option`.


def source_digest():
    """sha1 (16 hex digits) over csrc/*.hip, csrc/*.cpp, csrc/*.h (sorted by name) and
    include/tipk.h -- the same bytes, in the same order, as csrc/Makefile's BUILD_ID."""
    names = sorted(os.path.basename(f) for f in glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.cpp'))
                   + glob.glob(os.path.join(CSRC, '*.h')))
    h = hashlib.sha1()
    for n in names:
        with open(os.path.join(CSRC, n), 'rb') as f:
            h.update(f.read())
    with open(os.path.join(_HERE, '..', 'include', 'tipk.h'), 'rb') as f:
        h.update(f.read())
    return h.hexdigest()[:16]


def build(verbose=False, debug=False):
    """Compile libtipk.so for gfx950 with hipcc (cross-compiles without a GPU).  Always a child
    `make` process: a process that has touched the GPU must never exec another program."""
    cmd = ['make', '-C', CSRC, '-j4'] + (['debug'] if debug else [])
    out = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
        print(out.stderr)
    if out.returncode != 0:
        raise TipkError('building libtipk.so failed (see output above)')
    return LIB_PATH if not debug else os.path.join(_HERE, 'libtipk_debug.so')


def _stale_reason(path):
    """None if `path` was built from the sources next to it, else a description."""
    if not os.path.exists(path):
        return '%s is missing' % path
    import re
    with open(path, 'rb') as f:                      # read the marker from the file: no dlopen (see tipk_api.cpp)
        m = re.search(rb'TIPK_BUILD_ID=([0-9a-f]{16})', f.read())
    if m is None:
        return '%s carries no build id' % path
    have = m.group(1).decode()
    want = source_digest()
    return None if have == want else '%s was built from other sources (build id %s, sources %s)' % (path, have, want)


def ensure_built(verbose=False):
    """Build the library if it is missing or stale (child `make`; safe before or after GPU use).
    Called by tests/conftest.py, bench.py and __graft_entry__ before the first kernel launch."""
    global _lib
    why = _stale_reason(LIB_PATH)
    if why is not None:
        if _lib is not None:
            raise TipkError('%s, but a library is already loaded in this process: restart' % why)
        build(verbose=verbose)
        why = _stale_reason(LIB_PATH)
        if why is not None:
            raise TipkError('rebuilt library is still stale: %s' % why)
    return LIB_PATH


def lib():
    """The loaded library (cached).  Raises if it is missing or was built from other sources --
    no fallback, and no silent use of stale kernels.  TIPK_LIB=<path> selects another build of the
    SAME sources (the -DTIPK_DEBUG library of `make debug`)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get('TIPK_LIB') or LIB_PATH
    why = _stale_reason(path)
    if why is not None:
        raise TipkError('%s: run `python -c "import __graft_entry__ as g; g.build()"` (or `make -C tip_amd/csrc`); '
                        'tip_amd has no CPU fallback' % why)
    handle = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if handle.tipk_abi_version() != ABI_VERSION:
        raise TipkError('libtipk.so ABI %d != binding ABI %d: rebuild' % (handle.tipk_abi_version(), ABI_VERSION))
    _lib = handle
    return _lib


def build_id():
    return lib().tipk_build_id().decode()


def set_option(name, value):
    """Process-wide library option (include/tipk.h section 0)."""
    check(lib().tipk_set_option(name.encode(), int(value)), 'tipk_set_option(%s)' % name)


def get_option(name):
    v = C.c_int(0)
    check(lib().tipk_get_option(name.encode(), C.byref(v)), 'tipk_get_option(%s)' % name)
    return v.value


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
