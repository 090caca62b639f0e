"""-m gpu: the op-level C ABI (include/tipk.h section 10) driven by a host that is NOT the tip_amd package.

`examples/c_abi_host.py` is a stand-alone program: ctypes on libtipk.so + torch for device memory, importing neither
`tip_amd.ops` nor `tip_amd.plan` (it asserts that no module of the package was loaded).  It builds graph handles
(`tipk_graph_build`, range-list and edge-type form; `tipk_gcn_graph_build`, `tipk_hier_graph_build`), runs `tipk_rgcn_fwd` /
`tipk_rgcn_bwd_ex` (generic route, and -- after `tipk_graph_prepare_rgcn` -- the LDS-resident pair form on plans the library builds in C++), `tipk_gcn_fwd/_bwd` (PPEncoder on identity and on dense features), `tipk_hier_fwd/_bwd` and compares with the outputs and
autograd gradients of the reference's own MyRGCNConv2 / MyRGCNConv recorded in tests/golden (rgcn_sym, rgcn_directed, and
the two-layer 64 -> 32 -> 16 fixtures with the ReLU between the layers), and finally the WHOLE training step of TIP (`tip_add_small`:
P-P GCN x 2, P -> D, mix, R-GCN x 2, DistMult objective, all 13 parameter gradients) against the reference's loss and autograd.  It runs in a child process: this pytest process
has the package imported (tests/conftest.py builds the library through it)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rgcn_layer_through_the_graph_handle_from_a_foreign_host():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'examples', 'c_abi_host.py')], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'C-ABI host ok' in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])
    assert out.stdout.count('max error') == 8
    # the fixtures at the reference's dims (32 bases, 64 -> 32 -> 16) ran on the generic route AND in pair form behind the handle
    assert out.stdout.count('R-GCN routes taken: [0, 2]') == 2, out.stdout[-2000:]
