#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: D-D edges aggregated / s over encoder forward + backward.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload biosnap|biosnap963|synthetic|synthetic-small]

A "step" is one pass of the hot path over the full graph: `z = FMEncoder.forward(...)` (P-P GCN x2,
P->D mean, two R-GCN layers) followed by its backward with a fixed upstream gradient g ~ N(0,1)
(SURVEY.md section 8(d)).  Inputs (graph plans, weights, g) are resident in HBM before the timed
region; graph preprocessing is timed separately (`preprocess_s`).  value = E * K / t with E the
number of directed D-D train edges, t the max over ranks of the barrier-bracketed wall time.

N > 1: one rank per GPU over RCCL; the D-D relations are sharded over the ranks (tip_amd/dist.py),
partial aggregates and the replicated-parameter gradients are all-reduced; total work is fixed, so
"scaling" is "strong".  The driver may start the ranks itself (torch.distributed.run: RANK /
LOCAL_RANK / WORLD_SIZE in the environment) or run plain `python bench.py --gpus N`: then this
process -- BEFORE it touches the GPU -- starts `python -m torch.distributed.run --nproc-per-node N
bench.py ...` as a child, relays rank 0's JSON line and exits with the child's return code.
`--oversubscribe` lets N ranks share fewer GPUs (gloo collectives; a functional check of the N-rank
path on a 1-GPU box, not a measurement).

Objects on the JSON line besides the contract's fields:
  roofline      the dominant kernel: algorithmic bytes per launch ((4 + 4 d) B per edge)
                / its launch duration, measured live with HIP events on the launch stream around a
                hipGraph of 20 back-to-back launches of that kernel on the step's own plans (an event
                pair around ONE eager launch adds ~8 us of host/event overhead -- VERDICT r1 #3).
                bound = "lds" for the relation-local kernels (rows are read from LDS: the HBM figure
                of 8(d) is not their bound; peak = ds_read_b128 256 B/clk/CU x 256 CUs x 2.4 GHz),
                "hbm" otherwise.  traffic = HBM bytes per launch from the committed PMC summary of
                this command (profiles/*_pmc_traffic.json, keyed by kernel and full grid x*y*z);
                hbm_traffic_frac = traffic / duration / 8 TB/s.
                The object describes the LONGEST kernel of the step (today the two products of dY,
                bound "mfma"); `roofline_aggregation` keeps the slowest D-D aggregation launch.
  step_floor    sum over the step's launches of what bounds each one (LDS time of the gathers, MFMA time
                of the products, HBM time of the streaming kernels, 2 us for everything else = the measured
                hand-over of a launch inside a replayed graph) and floor / ms_per_step.
  parity_in_bench  the replayed graph's z and two gradients against the oracle pass of the cpu_baseline
                leg, which is handed the benched encoder's own weights (non-zero exit code on failure).
  other_configs / train_step  the other BASELINE configurations (TIP-add, the paper's 963 relations,
                config 5 on one GPU) and the whole graphed training epoch (tip.py:24-30), each timed by
                this process after the headline measurement (fewer steps; --no-extras skips them).
  cpu_baseline  the oracle's CPU port on the host cores (rank 0, N = 1 only), bounded sample.
  preprocess_s / init_s  the first step on the real graph (all gather plans are built there) / before it, one step
                on a 6 000-edge toy graph: what a fresh process spends loading torch's and libtipk's device code.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
LDS_PEAK_GBS = 256 * 256 * 2.4   # ds_read_b128: 256 B/clk/CU x 256 CUs x 2.4 GHz = 157 286 GB/s
L2_GATHER_GBS = 18800.0          # MI355X_MICROARCH.md gather table: rows of a table every workgroup shares out of the XCD's L2: 16.8-18.8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='biosnap', choices=['biosnap', 'biosnap963', 'synthetic', 'synthetic-small'])
    ap.add_argument('--mod', default='cat', choices=['cat', 'add'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='budget of the CPU baseline leg')
    ap.add_argument('--launch', default=None, choices=['graph', 'eager'],
                    help='graph: the step is captured once into a hipGraph and replayed (default); '
                         'eager: one ctypes launch per kernel')
    ap.add_argument('--oversubscribe', action='store_true',
                    help='allow more ranks than GPUs (ranks share devices, gloo collectives): functional check only')
    ap.add_argument('--no-kernel-table', action='store_true', help='skip the eager per-kernel event pass')
    ap.add_argument('--no-extras', action='store_true', help='skip other_configs / train_step')
    ap.add_argument('--settle', type=int, default=100,
                    help='untimed replays of the captured step BEFORE the W warm-up steps (a fresh process / a fresh box launches its '
                         'first graphs from cold host caches; reported as settle_replays)')
    ap.add_argument('--no-pmc', action='store_true', help='skip the rocprofv3 --pmc child passes (HBM traffic of this run)')
    ap.add_argument('--step-only', action='store_true',
                    help='run nothing but the warm-up and timed steps (PMC passes: bytes / (steps + warmup) = bytes per step)')
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------
# self-launch (plain `python bench.py --gpus N`): children are started before any GPU call here
# ---------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args):
    import torch                                   # device_count() does not initialise the GPU on this image
    from tip_amd import _lib
    _lib.ensure_built()                            # ONE build, here: N children racing `make` would corrupt the tree
    have = torch.cuda.device_count()
    if have < args.gpus and not args.oversubscribe:
        sys.stderr.write('bench.py: --gpus %d but this box has %d GPU(s); pass --oversubscribe to run %d ranks on '
                         'them (functional check of the N-rank path, gloo collectives)\n' % (args.gpus, have, args.gpus))
        return 3
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)        # stderr passes through
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
    if proc.returncode != 0 or line is None:
        sys.stderr.write('bench.py: the %d-rank run failed (rc %d)\n%s\n' % (args.gpus, proc.returncode, proc.stdout[-2000:]))
        return proc.returncode or 1
    print(line, flush=True)
    return 0


def make_workload(args):
    """-> (data dict on CPU, dims dict, description)."""
    from tip_amd.data import build_data_dict, synthetic_data_dict
    if args.workload.startswith('biosnap'):
        dd = build_data_dict(min_pairs=500 if args.workload == 'biosnap963' else None)
        dims = dict(prot_drug_dim=16, n_embed=48) if args.mod == 'cat' else dict(prot_drug_dim=64, n_embed=64)
        dims.update(n_hid1=32, n_hid2=16, num_base=32)
        name = 'TIP-%s full encoder, BioSNAP (645 drugs, 19081 proteins, R=%d), 1xMI355X config' % (
            args.mod, dd['n_dd_et'])
        return dd, dims, name
    if args.workload == 'synthetic':
        dd = synthetic_data_dict(with_protein_graph=True)
        dims = dict(prot_drug_dim=64, n_embed=64, n_hid1=128, n_hid2=128, num_base=32)
    else:
        dd = synthetic_data_dict(n_drug=2000, n_rel=200, n_edges=2_000_000, with_protein_graph=True)
        dims = dict(prot_drug_dim=64, n_embed=64, n_hid1=128, n_hid2=128, num_base=32)
    name = 'synthetic scaled graph (%d drugs, R=%d, E=%d, dim=128)' % (dd['n_drug'], dd['n_dd_et'],
                                                                    dd['dd_train_idx'].shape[1])
    return dd, dims, name


def cpu_baseline(dd, dims, mod, budget_s, weights, up):
    """The oracle (CPU port of the same algorithm: transform-then-gather with explicit backward)
    on this host's cores, plus the reference-shaped op sequence (PyG-CPU path: lift E x in,
    per-relation slice+mm, cat, scatter-mean, autograd backward) on a bounded relation sample.
    weights: the benched encoder's own parameters (state_dict names, CPU) -- the oracle computes with them, so
    its first pass is also the checker of `parity_in_bench`.  -> (record, z, gradients) of that pass."""
    import torch
    from oracle import tip_oracle as O
    E = dd['dd_train_idx'].shape[1]
    R = dd['n_dd_et']
    p = weights
    # torch's default (= all hardware threads) oversubscribes these small ops on big hosts
    threads = max(1, min(16, os.cpu_count() or 1))
    torch.set_num_threads(threads)

    def step():
        z, saved = O.fm_encoder_fwd(p, dd, mod)
        return z, O.fm_encoder_bwd(up, p, dd, saved, mod)
    zo, go = step()                                           # warm-up (page-in, thread pool) = the parity reference
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        if time.perf_counter() - t0 > budget_s * 0.6 or n >= 20:
            break
    dt = (time.perf_counter() - t0) / n
    out = {'value': E / dt, 'unit': 'edges/s', 'cores': threads, 'kind': 'port',
           'sample': 'full workload: %d oracle encoder fwd+bwd passes over all %d edges (%.2f s each)' % (n, E, dt)}

    # reference-shaped flavour on the first relations (bounded: its cost grows ~ R * E because every
    # `x_j[start:end]` slice backward zero-fills an E x in gradient)
    try:
        d_in = dims['n_embed'] + dims['prot_drug_dim'] if mod == 'cat' else dims['n_embed']

        def shaped(r_s):
            e_s = int(dd['dd_train_range'][r_s - 1, 1])
            ei = dd['dd_train_idx'][:, :e_s]
            rg = dd['dd_train_range'][:r_s]
            x = torch.randn(dd['n_drug'], d_in).requires_grad_(True)
            prm = [p['rgcn1.basis'][:, :d_in].clone().requires_grad_(True), p['rgcn1.att'][:r_s].clone().requires_grad_(True),
                   p['rgcn1.root'][:d_in].clone().requires_grad_(True)]
            g = torch.randn(dd['n_drug'], prm[0].shape[2])
            t0 = time.perf_counter()
            y = O.rgcn_fwd_reference_shaped(x, ei, rg, *prm)
            y.backward(g)
            return e_s, time.perf_counter() - t0
        # TWO samples: the second, 2.7 x as many relations, checks the cost model the full-size figure is extrapolated with
        r_a, r_b = min(R, 24), min(R, 64)
        e_a, dt_a = shaped(r_a)
        e_b, dt_b = shaped(r_b) if r_b > r_a else (e_a, dt_a)
        r_s, e_s, dt_ref = r_b, e_b, dt_b
        out['reference_shaped'] = {
            'value': e_s / dt_ref, 'unit': 'edges/s (one R-GCN layer fwd+bwd, on the SAMPLE)', 'cores': threads,
            'sample': 'first %d relations, %d edges, PyG op sequence under autograd, %.2f s' % (r_s, e_s, dt_ref),
            # time ~ c * R_s * E_s  ->  full graph: E / (c R E) = sample rate * R_s / R
            'full_size_estimate': e_s / dt_ref * r_s / R,
            'cost_model_check': {'samples': [[r_a, e_a, round(dt_a, 3)], [r_b, e_b, round(dt_b, 3)]],
                                 'time_ratio_measured': dt_b / dt_a,
                                 'time_ratio_of_c_R_E': (r_b * e_b) / float(r_a * e_a)},
            'full_size_estimate_note': 'sample rate x R_sample / R (cost model c*R*E of the slice backward, checked on two sample '
                                       'sizes: cost_model_check; the survey probe of the literal reference measured 0.020 M '
                                       'edges/s on 8 cores)'}
    except Exception as exc:                                   # never let the side leg kill the bench
        out['reference_shaped'] = {'error': repr(exc)}
    return out, zo, go


def parity_check(z_dev, grads, zo, go):
    """max |got - want| / max |want| of the replayed graph's z and two gradients against the oracle pass."""
    import torch
    rec, ok = {}, True
    for name, got, want, tol in (('z', z_dev, zo, 1e-5), ('grad rgcn1.att', grads['rgcn1.att'], go['rgcn1.att'], 2e-5),
                                 ('grad pp_encoder.conv1.lin.weight', grads['pp_encoder.conv1.lin.weight'],
                                  go['pp_encoder.conv1.lin.weight'], 2e-5)):
        got, want = got.detach().cpu().double(), want.double()
        err = float((got - want).abs().max()) / max(1e-30, float(want.abs().max()))
        rec[name] = {'max_rel_err': err, 'tol': tol}
        ok = ok and bool(torch.isfinite(got).all()) and err <= tol
    rec['ok'] = ok
    rec['checker'] = 'oracle/tip_oracle.py (CPU, fp32) on the benched encoder\'s own weights and upstream gradient'
    return rec


# ---------------------------------------------------------------------------------------------
# the large launches of the step, timed alone (roofline objects)
# ---------------------------------------------------------------------------------------------
def dd_launches(enc, dev):
    """[dict(label, key, grid, bound, work, unit_work, fn, ...)] for the D-D launches of one step -- the four
    aggregations, the dense halves of the pair form and the products of dY -- on the step's own plans with random
    tables of the step's shapes.  work = ALGORITHMIC bytes (gathers: (4 + 4 d) B per edge walked) or
    flops (products: 2 flops per multiply-add of the sums the math needs -- rows of dY without an edge are not
    counted, although the d att half of the kernel multiplies their zeros)."""
    import torch
    from tip_amd import ops
    out = []

    def add(label, key, grid, bound, work, fn, **extra):
        rec = {'label': label, 'key': key, 'grid': grid, 'bound': bound, 'work': float(work), 'fn': fn}
        rec.update(extra)
        out.append(rec)
    # both layers' pair cells in ONE launch (FMEncoder.forward: rgcn2's ride in rgcn1's cell launch)
    cells_two = False
    att_two = bool(getattr(enc, 'last_route', None) == 'encoder_step')     # ... and both layers' d att gathers (tip_amd/encoder.py)
    att_tables = []
    g1, g2 = enc.rgcn1._cache.value, enc.rgcn2._cache.value
    if g1 is not None and g2 is not None and enc.rgcn1.shard is None and enc.rgcn2.shard is None:
        n1 = g1.scale.numel()
        a1, a2 = enc.rgcn1.att.detach(), enc.rgcn2.att.detach()
        if ops.pair_cells_partner_ok(g1, a1, g2, a2, n1) and 'y' not in [v[0] for v in g1.fwd_route.values()]:
            cells_two = True
            nb1 = enc.rgcn1.num_bases
            c1 = g1.pair_buffers(n1, nb1, enc.rgcn1.out_channels, dev)[0]
            c2 = g2.pair_buffers(n1, nb1, enc.rgcn2.out_channels, dev)[0]
            t1, t2 = torch.randn_like(a1), torch.randn_like(a2)
            pf = g1.pair_fwd
            add('pair_cells[dd.fwd,both layers]', 'stream_gather_kernel<%d, %s, 1' % (nb1 // 4, 'true' if pf.idx_unit == nb1 * 4 else 'false'),
                '%dx2x1' % (pf.n_wg * 1024), 'lds', 2 * pf.n_edges * (4 + 4 * nb1),
                lambda pf=pf, t1=t1, t2=t2, c1=c1, c2=c2, nb1=nb1, n1=n1: ops.stream_gather_two(
                    pf, t1, t2, c1.view(-1, nb1)[:n1 * n1], c2.view(-1, nb1)[:n1 * n1]),
                edges=2 * pf.n_edges, row_floats=nb1, aggregation=True)
    for layer in (enc.rgcn1, enc.rgcn2):
        graph = layer._cache.value
        if graph is None:
            continue
        d = layer.out_channels
        n = graph.scale.numel()
        shard = layer.shard
        r = layer.num_relations if shard is None else int(shard.rel_ids.numel())
        if r == 0:
            continue
        n_edges_all = None
        y = None
        g = torch.randn(n, d, device=dev)
        nb = layer.num_bases
        pair_ok = graph.pair_fwd is not None and graph.pair_fwd.n_table == r and bool(ops.stream_gather_split(r, nb))
        # the backward pass the step takes behind a pair-form forward pass (ops._RGCN): pair form as well
        pair_bwd = graph.pair_bwd if (pair_ok and ops.pair_grads_supported(nb, d) and 'y' not in
                                      [v[0] for v in graph.fwd_route.values()]) else None
        d_in = layer.in_channels
        wave_uniform = getattr(graph.row_fwd, 'ROW_BYTES', None) is not None if not pair_ok else False
        rows_ok = ops.row_products_s_supported if wave_uniform else ops.row_products_supported
        if not pair_ok and graph.row_fwd is not None and rows_ok(n, r, nb, d_in) and rows_ok(n, r, nb, d):
            # LARGE node sets (config 5): both passes sum the (relation, node) rows in LDS and multiply them there
            # (tipk_rgcn_row_products) -- algorithmic flops = the products of the row sums, 2 x R N ch bases each
            x_in, att = torch.randn(n, d_in, device=dev), torch.randn(r, nb, device=dev)
            xb2 = torch.randn(nb, n * d, device=dev)
            rf, rb = graph.row_fwd, graph.row_bwd
            kname = 'row_products_s_kernel' if getattr(rf, 'ROW_BYTES', None) is not None else 'row_products_kernel'
            add('row_products[dd.fwd,ch=%d]' % d_in, kname + '<false', None, 'mfma', 2.0 * r * n * d_in * nb,
                lambda rf=rf, x_in=x_in, att=att: ops.row_products(rf, x_in, att), edges=rf.n_edges, row_floats=d_in,
                gathered_bytes=rf.n_edges * (4.0 + 4 * d_in), batches=int(rf.entries.shape[0]),
                note='T = att^T . S over the (relation, node) row sums S assembled in LDS from %d gathered rows' % rf.n_edges)
            add('row_products[dd.bwd,ch=%d]' % d, kname + '<true', None, 'mfma', 2 * 2.0 * r * n * d * nb,
                lambda rb=rb, g=g, att=att, xb2=xb2: ops.row_products(rb, g, att, xb2)[1], edges=rb.n_edges, row_floats=d,
                gathered_bytes=rb.n_edges * (4.0 + 4 * d), batches=int(rb.entries.shape[0]),
                note='d XB = att^T . S and d att = S . XB^T on the row sums of the transposed pass')
            continue
        for bwd in (False, True):
            rs = graph.rs_bwd if (bwd and pair_bwd is None) else None
            pair = None if bwd else graph.pair_fwd
            if bwd and pair_bwd is not None:
                pb = pair_bwd
                cells, xb_nb, _zeros = graph.pair_buffers(n, nb, d, dev)
                n_dp = int(pb.slot_of_pair.shape[0])
                pg_box = []

                tbl = 0 if layer is enc.rgcn1 else 1

                def grads(pb=pb, cells=cells, xb_nb=xb_nb, g=g, pg_box=pg_box, tbl=tbl):
                    pg_box[:] = [ops.pair_grads(pb, cells, xb_nb, g, table=tbl)[0]]
                grads()
                att_tables.append((pb, pg_box, nb))
                add('pair_grads[dd.bwd,d=%d]' % d, 'pair_grads_kernel<%d>' % d, None, 'mfma', 2 * 2.0 * n_dp * nb * d, grads,
                    pairs=n_dp, hbm_bytes=4.0 * nb * (n_dp + pb.n_slots),
                    note='algorithmic flops = the two products of the pair form over the %d linked (source, neighbour) pairs: '
                         'd XB += cells^T g\' and d C = XB g\', 2 x 2 x pairs x bases x d' % n_dp)
                if not att_two:
                    add('pair_att_gather[dd.bwd,d=%d]' % d, 'stream_gather_kernel<8, %s, 2' % ('true' if pb.gather.idx_unit == nb * 4 else 'false'),
                        '%dx1x1' % (pb.gather.n_wg * 1024), 'lds', pb.gather.n_edges * (4 + 4 * nb),
                        lambda pb=pb, pg_box=pg_box: ops.pair_att_gather(pb, pg_box[0]), edges=pb.gather.n_edges, row_floats=nb,
                        aggregation=True)
                elif len(att_tables) == 2:
                    # the encoder-level schedule (tip_amd/encoder.py) gathers d att of both layers in ONE launch
                    from tip_amd import encoder as _enc
                    (pb0, box0, nb0), (_, box1, _) = att_tables
                    add('pair_att_gather[dd.bwd,both layers]', 'stream_gather_kernel<8, %s, 2' % ('true' if pb0.gather.idx_unit == nb0 * 4 else 'false'),
                        '%dx2x1' % (pb0.gather.n_wg * 1024), 'lds', 2 * pb0.gather.n_edges * (4 + 4 * nb0),
                        lambda pb0=pb0, box0=box0, box1=box1: _enc.pair_att_gather_two(pb0, box0[0], box1[0]),
                        edges=2 * pb0.gather.n_edges, row_floats=nb0, aggregation=True)
            elif pair is not None and pair_ok:
                # forward in pair form: per edge one id + one att row (nb floats) from LDS, then the dense product
                split = ops.stream_gather_split(r, nb)
                key = 'stream_gather_kernel<%d, %s, 1' % (nb // split // 4, 'true' if pair.idx_unit == nb // split * 4 else 'false')
                att = torch.randn(r, nb, device=dev)
                cells, xb_nb, _zeros = graph.pair_buffers(n, nb, d, dev)
                if not cells_two:
                    add('pair_cells[dd.fwd,d=%d]' % d, key, '%dx%dx1' % (pair.n_wg * 1024, split), 'lds', pair.n_edges * (4 + 4 * nb),
                        lambda pair=pair, att=att, cells=cells, nb=nb, n=n: ops.stream_gather(
                            pair, att, write_zeros=False, out=cells.view(-1, nb)[:n * n], kind=1),
                        edges=pair.n_edges, row_floats=nb, aggregation=True)
                add('pair_product[dd.fwd,d=%d]' % d, 'pair_product_kernel', None, 'mfma', 2.0 * n * n * nb * d,
                    lambda cells=cells, xb_nb=xb_nb, pair=pair, z=_zeros: ops.pair_product(cells, xb_nb, symmetric=pair.symmetric,
                                                                                           links=pair.links, zeros=z),
                    hbm_bytes=cells.numel() * 4.0 / (2 if pair.symmetric else 1))
            elif rs is not None and ops.rel_stream_split(n, d):
                split = ops.rel_stream_split(n, d)
                key = 'stream_gather_kernel<%d, %s, 0' % (d // split // 4, 'true' if rs.idx_unit == d // split * 4 else 'false')
                add('rel_stream[dd.bwd,d=%d]' % d, key, '%dx%dx1' % (rs.n_wg * 1024, split), 'lds', rs.n_edges * (4 + 4 * d),
                    lambda rs=rs, g=g: ops.rel_stream_bwd(rs, g, row_scale=graph.scale), edges=rs.n_edges, row_floats=d,
                    aggregation=True)
                if rs.compact is not None:
                    att = torch.randn(r, nb, device=dev)
                    # the operands as the step hands them over: XB node-major (rows padded to 32 columns: the pair product's
                    # buffer, through strides) and a second time as [N, d, bases] for the d att product
                    xb = torch.randn(n, nb, 32 if d <= 32 else d, device=dev).permute(1, 0, 2)[:, :, :d]
                    xbt = xb.permute(1, 2, 0).contiguous()
                    dyc = ops.rel_stream_bwd(rs, g, row_scale=graph.scale)
                    rows = rs.compact.n_rows
                    add('node_products[dd.bwd,d=%d]' % d, 'node_products_kernel<%d, true>' % d, None, 'mfma', 2 * 2.0 * rows * d * nb,
                        lambda dyc=dyc, cr=rs.compact, att=att, xb=xb, xbt=xbt: ops.node_products(dyc, cr, att, xb, xbt),
                        rows=rows, flops_dense_form=2 * 2.0 * r * n * d * nb)
            elif ops.rel_gather_usable(graph.rl_bwd if bwd else graph.rl_fwd, n, d, bwd):
                rp = graph.rl_bwd if bwd else graph.rl_fwd
                split = ops.rel_gather_split(n, d, bwd)
                key = 'rel_gather_kernel<%d, %s' % (d // split // 4, 'true' if bwd else 'false')
                if y is None:
                    y = torch.randn(r * n, d, device=dev)
                fn = (lambda rp=rp, g=g: ops.rel_gather(rp, g, True, row_scale=graph.scale)) if bwd else \
                     (lambda rp=rp, y=y: ops.rel_gather(rp, y, False, reduce=False))
                add('rel_gather[dd.%s,d=%d]' % ('bwd' if bwd else 'fwd', d), key, '%dx%dx1' % (rp.n_wg * 1024, split), 'lds',
                    None, fn, row_floats=d, aggregation=True)
            else:
                lanes = 1
                while lanes < d // 4:
                    lanes *= 2
                if y is None and not bwd:
                    y = torch.randn(r * n, d, device=dev)
                if bwd and d % 4 == 0 and 8 <= d <= 256:
                    csr = graph.csr_bwd                        # the path _RGCN.backward takes on large graphs
                    lanes = max(lanes, 2)
                    rp = lanes - 1 if lanes <= 16 else 16
                    tasks = -(-csr.n_out // rp)
                    waves = -(-tasks // (64 // lanes))
                    add('gather_rows_csr[dd.bwd,d=%d]' % d, 'gather_rows_csr_kernel<%d' % lanes, '%dx1x1' % (-(-waves // 4) * 256),
                        'hbm', csr.n_edges * (4 + 4 * d), lambda csr=csr, g=g: ops.gather_rows_csr(csr, g),
                        edges=csr.n_edges, row_floats=d, aggregation=True)
                    continue
                plan = graph.bwd if bwd else graph.fwd
                waves = -(-plan.items.shape[0] // (64 // lanes))
                fn = (lambda plan=plan, g=g: ops.gather_sum(plan, g)) if bwd else (lambda plan=plan, y=y: ops.gather_sum(plan, y))
                add('gather_sum[dd.%s,d=%d]' % ('bwd' if bwd else 'fwd', d), 'gather_sum_kernel<4, %d' % lanes,
                    '%dx1x1' % (-(-waves // 4) * 256), 'hbm', plan.n_edges * (4 + 4 * d), fn, edges=plan.n_edges, row_floats=d,
                    aggregation=True)
        rs = graph.rs_bwd if pair_bwd is None else None
        if pair_bwd is None and (rs is None or rs.compact is None) and ops.dy_products_fused(r, n * d, nb):
            g_y = torch.randn(r, n * d, device=dev)
            att = torch.randn(r, nb, device=dev)
            xb2 = torch.randn(nb, n * d, device=dev)
            add('dy_products[dd.bwd,d=%d]' % d, 'dy_products_kernel', None, 'mfma', 2 * 2.0 * r * n * d * nb,
                lambda g_y=g_y, att=att, xb2=xb2: ops.dy_products(g_y, att, xb2))
        del y
    return out


def time_launch_us(fn, reps=20, replays=5):
    """Device time of one launch of `fn`: HIP events on the launch stream around `replays` replays of
    a hipGraph holding `reps` back-to-back launches."""
    import torch
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(reps):
                fn()
        g.replay()                                         # warm
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(side)
        for _ in range(replays):
            g.replay()
        b.record(side)
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps * replays)


_LIVE_PMC = None          # {'kernels': {name + grid: hbm bytes per launch}, 'step_hbm_bytes': ..} measured by THIS run (below)


def measure_pmc_in_run(args):
    """HBM traffic of the step's kernels measured BY THIS RUN: two child processes under `rocprofv3 --kernel-trace --pmc`
    (FETCH_SIZE and WRITE_SIZE need separate passes: MI355X_MICROARCH.md, rocprofv3 PMC slots) run `bench.py --step-only
    --launch eager` for a few steps before this process touches the GPU; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 per
    launch (gfx950: FETCH_SIZE counts half of a wide read), keyed by kernel name and full grid.  Any failure -> None (the
    committed summaries of the same build id are looked up instead)."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    from collections import defaultdict
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return None
    csv.field_size_limit(1 << 30)

    def short(name):
        name = re.sub(r'\(anonymous namespace\)::', '', name)
        name = re.sub(r'^void ', '', name)
        m = re.match(r'([A-Za-z0-9_:]+(<[^(]*>)?)\(', name)
        return (m.group(1) if m else name)[:110]
    tmp = tempfile.mkdtemp(prefix='tipk_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp', TIPK_BENCH_CHILD='1')
    steps, warm = 3, 1
    # full-size steps the child runs: prepare() (the plan-building first step) + warm-up + timed; a --step-only child runs
    # no toy-graph step (main: `init_s`), so every libtipk launch it records belongs to one of them
    steps_run = steps + warm + 1
    try:
        per = {}
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(tmp, counter)
            cmd = [exe, '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', out, '--', 'python3',
                   os.path.join(ROOT, 'bench.py'), '--steps', str(steps), '--warmup', str(warm), '--launch', 'eager', '--step-only',
                   '--no-cpu-baseline', '--workload', args.workload, '--mod', args.mod]
            r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                return None
            fn = sorted(glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True))
            tr = sorted(glob.glob(os.path.join(out, '**', '*kernel_trace.csv'), recursive=True))
            if not fn or not tr:
                return None
            shape = {row['Dispatch_Id']: '%sx%sx%s' % (row['Grid_Size_X'], row['Grid_Size_Y'], row['Grid_Size_Z'])
                     for row in csv.DictReader(open(tr[0]))}
            acc = defaultdict(list)
            for row in csv.DictReader(open(fn[0])):
                if row['Counter_Name'] == counter:
                    acc['%s grid=%s' % (short(row['Kernel_Name']), shape.get(row['Dispatch_Id'], '?'))].append(float(row['Counter_Value']))
            for k, v in acc.items():
                per.setdefault(k, {})[counter] = (sum(v) / len(v), len(v))
        return pmc_summary(per, steps_run)
    except Exception:                                                      # noqa: BLE001
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_summary(per, steps_run):
    """per: {kernel + grid: {'FETCH_SIZE': (mean KB per launch, launches), 'WRITE_SIZE': (..)}} of a run of `steps_run` full-size
    steps -> per-launch HBM bytes of every kernel, corrected as MI355X_MICROARCH.md's HBM section prescribes for gfx950
    (2 x FETCH_SIZE + WRITE_SIZE: FETCH_SIZE counts half of a wide streaming read) AND uncorrected (FETCH_SIZE + WRITE_SIZE:
    the factor is stated for wide coalesced reads, a row-per-lane gather may not need it -- the truth lies between), and
    the bytes of ONE step = sum over libtipk's kernels of bytes per launch x launches / steps_run."""
    kernels, raw, total, total_raw = {}, {}, 0.0, 0.0
    for k, d in per.items():
        f_kb, n_f = d.get('FETCH_SIZE', (0.0, 0))
        w_kb, n_w = d.get('WRITE_SIZE', (0.0, 0))
        kernels[k] = (2.0 * f_kb + w_kb) * 1024.0
        raw[k] = (f_kb + w_kb) * 1024.0
        if not (k.startswith('at::') or k.startswith('rocprim') or 'elementwise' in k or k.startswith('__amd_rocclr')):
            total += kernels[k] * max(n_f, n_w)
            total_raw += raw[k] * max(n_f, n_w)
    return {'kernels': kernels, 'kernels_uncorrected': raw, 'step_hbm_bytes': total / steps_run,
            'step_hbm_bytes_uncorrected': total_raw / steps_run, 'steps_run': steps_run}


def pmc_traffic(key_prefix, grid, build_id):
    """(HBM bytes per launch, trace us, source file) of a kernel from the newest committed summaries
    (profiles/*_pmc_traffic.json / *_kernel_by_grid.csv: separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes and the --kernel-trace of this command; bytes corrected as
    MI355X_MICROARCH.md prescribes).  Entries are keyed by kernel name and the FULL grid XxYxZ (grid None: any).
    A summary is only used when it was profiled on THIS build of the library (its `build_id` stamp): numbers of
    other kernels would go stale silently."""
    import csv
    import glob
    traffic = trace_us = src = None

    def match(name):
        return name.startswith(key_prefix) and (grid is None or name.endswith('grid=' + grid))
    if _LIVE_PMC is not None:
        hits = [(b, name) for name, b in _LIVE_PMC['kernels'].items() if match(name)]
        if hits:
            b, name = max(hits)
            return b, None, 'this run', _LIVE_PMC.get('kernels_uncorrected', {}).get(name)
    try:
        for fn in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')), reverse=True):
            doc = json.load(open(fn))
            if doc.get('build_id') != build_id:
                continue
            for name, k in doc['kernels'].items():
                if match(name):
                    traffic, src = k['hbm_bytes_per_launch'], os.path.basename(fn)
                    break
            tr = fn.replace('_pmc_traffic.json', '_kernel_by_grid.csv')
            if traffic is not None and os.path.exists(tr):
                rows = [r for r in csv.reader(l for l in open(tr) if not l.startswith('#'))][1:]
                hit = [r for r in rows if match(r[0])]
                if hit:
                    trace_us = float(hit[0][2]) / 1e3
            if traffic is not None:
                break
    except Exception:
        pass
    return traffic, trace_us, src, None


def step_hbm_bytes(build_id):
    """(HBM bytes of ONE step, file) from the newest committed PMC summary of THIS build that has the total
    (tools/summarize_prof.py: all libtipk launches of `bench.py --step-only` / (steps + warmup))."""
    import glob
    if _LIVE_PMC is not None and _LIVE_PMC.get('step_hbm_bytes'):
        return float(_LIVE_PMC['step_hbm_bytes']), 'this run', _LIVE_PMC.get('step_hbm_bytes_uncorrected')
    try:
        for fn in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')), reverse=True):
            doc = json.load(open(fn))
            if doc.get('build_id') == build_id and doc.get('step_hbm_bytes'):
                return float(doc['step_hbm_bytes']), os.path.basename(fn), doc.get('step_hbm_bytes_uncorrected')
    except Exception:
        pass
    return None


def synthetic_hbm_counter(ms):
    """Measured HBM bytes of one config-5 step (the newest committed profiles/r*_synthetic_pmc_traffic.json) against
    8 TB/s at THIS run's step time; the build the counters were taken on is named, since the file may be older than the library."""
    import glob
    try:
        fn = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_synthetic_pmc_traffic.json')))[-1]
        doc = json.load(open(fn))
        b = float(doc['step_hbm_bytes'])
        return {'bytes_per_step': b, 'bytes_per_step_uncorrected': doc.get('step_hbm_bytes_uncorrected'),
                'frac_of_8TBps': b / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9),
                'source': 'profiles/%s (build %s; the config-5 kernels have not changed since)' % (os.path.basename(fn), doc.get('build_id'))}
    except Exception:
        return None


MFMA_F32_PEAK = 157.3e12          # MI355X_MICROARCH.md: dense fp32 matrix rate
LAUNCH_FLOOR_US = 2.0             # hand-over of a small launch inside a replayed graph (tools/microbench/launch_floor.hip: 1.6 empty, 2.5 at the margin)


def roofline_of(rec, us, build_id):
    """The contract's roofline object of one launch record of `dd_launches` timed at `us`."""
    if rec['bound'] == 'mfma':
        achieved = rec['work'] / (us * 1e-6) / 1e12
        roof = {'bound': 'mfma', 'kernel': rec['label'], 'achieved': achieved, 'peak': MFMA_F32_PEAK / 1e12, 'unit': 'TFLOP/s',
                'frac': achieved * 1e12 / MFMA_F32_PEAK, 'launch_us': us, 'algorithmic_flops_per_launch': rec['work']}
        if rec.get('note'):
            roof['note'] = rec['note']
        if rec.get('flops_dense_form'):
            roof['frac_dense_form'] = rec['flops_dense_form'] / (us * 1e-6) / MFMA_F32_PEAK
            roof['note'] = ('algorithmic flops = both products of dY over the %d (relation, source) rows that have an edge; '
                            'frac_dense_form prices all R x N rows (the figure quoted for the dense kernel of round 2)' % rec['rows'])
    else:
        peak = LDS_PEAK_GBS if rec['bound'] == 'lds' else HBM_PEAK_GBS
        achieved = rec['work'] / (us * 1e-6) / 1e9
        roof = {'bound': rec['bound'], 'kernel': rec['label'], 'achieved': achieved, 'peak': peak, 'unit': 'GB/s',
                'frac': achieved / peak, 'launch_us': us, 'algorithmic_bytes_per_launch': rec['work'],
                'edges_per_launch': rec.get('edges'), 'row_floats': rec.get('row_floats')}
        if rec['bound'] == 'lds':
            roof['note'] = ('rows are gathered from LDS (wave-stream / relation-local kernel): the bound is the ds_read_b128 '
                            'rate; HBM only carries the ids and the output rows')
    roof['grid'] = rec['grid']
    roof['timing'] = 'HIP events on the launch stream around a hipGraph of 20 back-to-back launches'
    traffic, trace_us, src, traffic_raw = pmc_traffic(rec['key'], rec['grid'], build_id)
    roof['traffic'] = traffic
    if traffic_raw is not None:
        roof['traffic_uncorrected'] = traffic_raw          # FETCH_SIZE + WRITE_SIZE without the gfx950 factor 2 on FETCH_SIZE
    if traffic is not None:
        roof['hbm_traffic_frac'] = traffic / (us * 1e-6) / 1e9 / HBM_PEAK_GBS
        roof['traffic_source'] = ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run (child processes, eager launches)'
                                  if src == 'this run' else 'profiles/' + src + ' (same build id)')
    if trace_us is not None:
        roof['rocprof_trace_us'] = trace_us
    return roof


def step_floor(launch_us, launches, kern, pp_edges, dims, pp_rows_edges=None):
    """What the step's own kernels' bounds allow: sum over its launches of the larger of the kernel's roofline time
    and the launch floor.  launches: records of `dd_launches` (modelled); kern: the eager per-kernel table
    (label -> (launches per pass ..)) -- every launch that is not modelled counts LAUNCH_FLOOR_US, the four P-P
    gathers their L2 gather time."""
    parts = {}
    for rec in launches:
        if rec['work'] is None:
            continue
        if rec['bound'] == 'mfma':
            t = rec['work'] / MFMA_F32_PEAK * 1e6
            if rec.get('hbm_bytes'):
                t = max(t, rec['hbm_bytes'] / (HBM_PEAK_GBS * 1e9) * 1e6)
        else:
            t = rec['work'] / ((LDS_PEAK_GBS if rec['bound'] == 'lds' else HBM_PEAK_GBS) * 1e9) * 1e6
        parts[rec['label']] = max(t, LAUNCH_FLOOR_US)
    modelled = len(parts)
    n_launches = None
    if kern:
        passes = max(v[0] for v in kern.values())
        passes = min(v[0] for v in kern.values()) if passes else 1
        n_launches = int(round(sum(v[0] for v in kern.values()) / max(1, passes)))
        pp = 0
        for label in kern:
            if label.startswith('gather_sum[pp.') or label.startswith('gather_sum_lin[pp.'):
                dcol = int(label.split('d=')[1].split(']')[0].split('->')[0])       # ('...,d=32]+2 sums': slab sums riding in the launch)
                n_e = pp_rows_edges if ('.rows' in label and pp_rows_edges) else pp_edges    # conv2: only the rows that are read
                t = n_e * (4 + 4 * dcol) / (L2_GATHER_GBS * 1e9) * 1e6              # the 2.4 MB table is L2-resident
                parts[label] = max(t, LAUNCH_FLOOR_US) * (kern[label][0] // passes)
                pp += kern[label][0] // passes
        rest = max(0, n_launches - modelled - pp)
        parts['%d other launches x %.1f us' % (rest, LAUNCH_FLOOR_US)] = rest * LAUNCH_FLOOR_US
    return {'us': sum(parts.values()), 'launches_per_step': n_launches, 'parts_us': {k: round(v, 2) for k, v in parts.items()},
            'note': 'sum over the launches of one step of max(kernel roofline time, %.1f us launch floor): D-D gathers at the LDS '
                    'ds_read_b128 peak, products at the fp32 MFMA peak (pair product: or its one pass over the cell matrix at 8 TB/s), '
                    'P-P gathers (`gather_sum`: rows of a 2.4 MB table out of the L2 of the XCD) at the L2 gather rate, 18.8 TB/s, '
                    'every other launch at the floor' % LAUNCH_FLOOR_US}


IC_GATHER_GBS = 8600.0            # MI355X_MICROARCH.md gather table: uniformly random rows of a table held by the Infinity Cache


def dense_route_floor(r, n, e, dims, other_us=200.0):
    """What bounds a step on the DENSE route (Y = att . XB and dY materialised: node sets whose pair cells do not fit, config
    5): per layer the Y product (fp32 MFMA peak, or its one write of Y at 8 TB/s), the forward gather of E rows of Y at the
    Infinity-Cache gather rate, the transposed pass (E rows of g' gathered out of L2, dY written once at 8 TB/s) and both
    products of dY (fp32 MFMA peak, or one read of dY at 8 TB/s); everything else (P-P / P->D stage, parameter-gradient
    products, slab sums: `other_us`, the launch floors of ~30 launches + their 10 GB-scale slab traffic) as measured."""
    nb = dims['num_base']
    parts = {}
    for li, name in enumerate(('n_hid1', 'n_hid2')):
        d = dims[name]
        y_bytes = 4.0 * r * n * d
        parts['Y = att . XB [layer %d, d=%d]' % (li + 1, d)] = max(2.0 * r * nb * n * d / MFMA_F32_PEAK, y_bytes / (HBM_PEAK_GBS * 1e9)) * 1e6
        parts['gather of Y [layer %d, d=%d]' % (li + 1, d)] = e * (4 + 4.0 * d) / (IC_GATHER_GBS * 1e9) * 1e6
        parts['transposed pass, dY written [layer %d, d=%d]' % (li + 1, d)] = max(e * (4 + 4.0 * d) / (L2_GATHER_GBS * 1e9), y_bytes / (HBM_PEAK_GBS * 1e9)) * 1e6
        parts['both products of dY [layer %d, d=%d]' % (li + 1, d)] = max(2 * 2.0 * r * nb * n * d / MFMA_F32_PEAK, y_bytes / (HBM_PEAK_GBS * 1e9)) * 1e6
    parts['everything else (measured)'] = other_us
    return {'us': sum(parts.values()), 'parts_us': {k: round(v, 1) for k, v in parts.items()},
            'note': 'dense route: Y / dY round trips at 8 TB/s or the fp32 MFMA peak of their products, the gather of Y at the '
                    'Infinity-Cache gather rate (8.6 TB/s), the gather of g\' rows out of L2 (18.8 TB/s)'}


def row_route_floor(r, n, e, widths, nb, other_us=400.0):
    """What bounds a step on the ROW-SUM route (round 5, config 5: `tipk_rgcn_row_products` -- neither Y nor dY exists): per
    layer the forward launch (the product of the row sums with att at the fp32 MFMA peak, or E rows of X gathered out of L2)
    and the transposed launch (both products, or E rows of g' out of L2); everything else (P-P / P->D stage, XB, T . basis, the
    parameter-gradient products, the d att slab sums: `other_us`) as measured.  widths = [(d_in, d_out)] per layer."""
    parts = {}
    for li, (d_in, d_out) in enumerate(widths):
        parts['forward row products [layer %d, ch=%d]' % (li + 1, d_in)] = max(2.0 * r * nb * n * d_in / MFMA_F32_PEAK,
                                                                              e * (4 + 4.0 * d_in) / (L2_GATHER_GBS * 1e9)) * 1e6
        parts['transposed row products [layer %d, ch=%d]' % (li + 1, d_out)] = max(2 * 2.0 * r * nb * n * d_out / MFMA_F32_PEAK,
                                                                                  e * (4 + 4.0 * d_out) / (L2_GATHER_GBS * 1e9)) * 1e6
    parts['everything else (measured)'] = other_us
    return {'us': sum(parts.values()), 'parts_us': {k: round(v, 1) for k, v in parts.items()},
            'note': 'row-sum route: the products of the (relation, node) row sums at the fp32 MFMA peak, or the gather of the '
                    'E table rows they are summed from out of L2 (18.8 TB/s)'}


# ---------------------------------------------------------------------------------------------
# one configuration: build, capture, time
# ---------------------------------------------------------------------------------------------
class Bench(object):
    """Encoder forward + backward of one workload on this rank: build -> capture -> timed replays."""

    def __init__(self, dd, dims, mod, dev, shard=None):
        import torch
        from tip_amd.data import Data
        from tip_amd.layers import FMEncoder
        self.dd, self.dims, self.mod, self.dev = dd, dims, mod, dev
        torch.manual_seed(1111)
        n_rel = dd['n_dd_et']
        self.enc = FMEncoder(dev, dd['n_drug_feat'], n_rel, dd['n_prot'], dd['n_prot'], dd['n_drug'], mod=mod, **dims).to(dev)
        self.d = Data.from_dict({k: v for k, v in dd.items() if k != 'dd_edge_index'}).to(dev)
        self.g_up_cpu = torch.randn(dd['n_drug'], dims['n_hid2'], generator=torch.Generator().manual_seed(0))
        self.g_up = self.g_up_cpu.to(dev)
        self.z = None
        self.graph = None
        self.static_z, self.static_grads = None, None

    def step(self):
        enc, d = self.enc, self.d
        for prm in enc.parameters():
            prm.grad = None
        z = enc(d.d_feat, d.dd_train_idx, d.dd_train_et, d.dd_train_range, d.d_norm, d.p_feat,
                d.pp_train_indices, d.dp_edge_index, d.dp_range_list)
        z.backward(self.g_up)
        self.z = z.detach()            # the buffer, not the autograd graph: a graph kept alive across steps breaks capture

    def prepare(self):
        """first step: builds + caches all gather plans.  -> seconds"""
        import torch
        t0 = time.perf_counter()
        self.step()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def capture(self, thread_local=False):
        """The whole step (about 30 kernels, plus the RCCL all-reduces when sharded) becomes one hipGraph: replay
        removes the per-launch host cost, which is larger than the kernels themselves at BioSNAP scale."""
        import torch
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        # with a process group alive the RCCL watchdog thread makes HIP calls of its own: only this
        # thread's calls may invalidate the capture
        mode = {'capture_error_mode': 'thread_local'} if thread_local else {}
        with torch.cuda.graph(graph, **mode):
            self.step()
        self.graph = graph
        # the graph's own output buffers (later eager steps re-bind .grad and self.z to fresh tensors)
        self.static_z = self.z
        self.static_grads = {k: prm.grad for k, prm in self.enc.named_parameters()}
        return graph.replay

    def capture_many(self, k):
        """k consecutive steps in ONE hipGraph (a training loop may replay several epochs at once): the hand-over between two
        replays (~9 us) is paid once per k steps.  Reported NEXT TO the headline, never as it."""
        import torch
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(k):
                self.step()
        return graph

    def replicated_stage_us(self, reps=20):
        """us per replay of the part of the step that a relation-sharded run computes on EVERY rank: P-P GCN x2, P -> D mean,
        drug mix -- forward and backward with a fixed upstream gradient, as its own hipGraph."""
        import torch
        enc, d = self.enc, self.d
        params = [enc.embed, enc.hgcn.weight] + list(enc.pp_encoder.parameters())

        def stage():
            for prm in params:
                prm.grad = None
            x0 = enc.mixed_drug_features(d.d_feat, d.d_norm, d.p_feat, d.pp_train_indices, d.dp_edge_index, d.dp_range_list)
            x0.backward(self._g_x0)
        with torch.no_grad():
            probe = enc.mixed_drug_features(d.d_feat, d.d_norm, d.p_feat, d.pp_train_indices, d.dp_edge_index, d.dp_range_list)
        self._g_x0 = torch.randn(probe.shape, generator=torch.Generator().manual_seed(1)).to(self.dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            stage()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode='thread_local'):
            stage()
        for _ in range(3):
            graph.replay()
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            graph.replay()
        b_.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b_) / reps * 1e3

    def outputs(self):
        """(z, {name: gradient}) of the most recent step: the replayed graph's buffers, or the eager tensors."""
        if self.graph is not None:
            return self.static_z, self.static_grads
        return self.z, {k: prm.grad for k, prm in self.enc.named_parameters()}

    def weights_cpu(self):
        return {k: v.detach().cpu().contiguous() for k, v in self.enc.state_dict().items()}


def timed(run, steps, warmup, fence):
    for _ in range(warmup):
        run()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    fence()
    return time.perf_counter() - t0


def replay_event_stats(run, n, fence):
    """Per-replay GPU durations of the timed object (hipEvent timing, median of >= 20): one HIP event pair around
    EACH of n replays on the launch stream -> {n, median_us, p10_us, p90_us, min_us, max_us}.  Reported next to the contract's
    wall-clock figure (K steps between two fences, host perf_counter), which stays the headline `value`."""
    import torch
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        run()
        b.record()
    fence()
    us = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    q = lambda f: us[min(n - 1, int(f * n))]
    return {'n': n, 'median_us': q(0.5), 'p10_us': q(0.1), 'p90_us': q(0.9), 'min_us': us[0], 'max_us': us[-1],
            'timing': 'one HIP event pair per replay on the launch stream (the pair itself adds ~1 us inside the bracket)'}


def release(*objs):
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()


def measure_config(workload, mod, dev, steps=20, warmup=5):
    """ms/step, edges/s and the longest D-D kernel of another configuration (single GPU, hipGraph replay)."""
    import torch
    from tip_amd import _lib
    a = argparse.Namespace(workload=workload, mod=mod)
    t0 = time.perf_counter()
    dd, dims, name = make_workload(a)
    gen_s = time.perf_counter() - t0
    b = Bench(dd, dims, mod, dev)
    pre = b.prepare()
    run = b.capture()
    launches = dd_launches(b.enc, dev)                     # (before the timed region, as in main)
    us = {l['label']: time_launch_us(l['fn'], reps=10 if workload == 'synthetic' else 20, replays=3) for l in launches
          if l['work'] is not None}
    el = timed(run, steps, warmup, torch.cuda.synchronize)
    E = int(dd['dd_train_idx'].shape[1])
    ms = el / steps * 1e3
    rec = {'workload': name, 'mod': mod, 'directed_dd_edges': E, 'relations': dd['n_dd_et'], 'steps': steps, 'warmup': warmup,
           'ms_per_step': ms, 'value': E * steps / el, 'unit': 'edges/s', 'preprocess_s': pre, 'generate_s': gen_s}
    if us:
        dom = max(us, key=us.get)
        roof = roofline_of([l for l in launches if l['label'] == dom][0], us[dom], _lib.build_id())
        rec['dominant_kernel'] = {k: roof[k] for k in ('kernel', 'bound', 'launch_us', 'achieved', 'peak', 'unit', 'frac')}
        rec['dd_launches_us'] = {k: round(v, 2) for k, v in us.items()}
    per_edge = sum(2 * (8 + 4 * dims[k]) for k in ('n_hid1', 'n_hid2'))
    if workload.startswith('synthetic'):
        # config 5: rows of Y / dY are gathered from HBM-resident tables -- SURVEY 8(d)'s HBM yardstick applies;
        # next to it the dense side (Y = att . XB, both products of dY, XB / dX / d basis products)
        rec['hbm_roofline_frac'] = E * per_edge / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9)
        rec['algorithmic_hbm_frac'] = rec['hbm_roofline_frac']
        rec['binding_roofline'] = 'mfma (fp32): the row sums re-read Y / dY out of L2, so the algorithmic-HBM figure is a ' \
                                  'yardstick, not the bound -- see hbm_counter, mfma_side and step_floor'
        if workload == 'synthetic':
            rec['hbm_counter'] = synthetic_hbm_counter(ms)
        n, r, nb = dd['n_drug'], dd['n_dd_et'], dims['num_base']
        flops = sum(3 * 2.0 * r * n * dims[k] * nb for k in ('n_hid1', 'n_hid2'))
        rec['mfma_side'] = {'flops_per_step': flops, 'ms_at_fp32_mfma_peak': flops / MFMA_F32_PEAK * 1e3,
                            'note': 'Y = att . XB forward and the two products of dY backward, per layer; the HBM side prices '
                                    '%d B per edge' % per_edge}
        rows_route = any(l['label'].startswith('row_products') for l in launches)
        if rows_route:
            widths = [(b.enc.rgcn1.in_channels, b.enc.rgcn1.out_channels), (b.enc.rgcn2.in_channels, b.enc.rgcn2.out_channels)]
            flops = sum(2.0 * r * n * nb * (d_in + 2 * d_out) for d_in, d_out in widths)
            rec['mfma_side'] = {'flops_per_step': flops, 'ms_at_fp32_mfma_peak': flops / MFMA_F32_PEAK * 1e3,
                                'note': 'the products of the row sums: T = att^T . S forward, d XB = att^T . S\' and d att = '
                                        'S\' . XB^T backward, per layer; the HBM side prices %d B per edge' % per_edge}
            fl = row_route_floor(r, n, E, widths, nb)
        else:
            fl = dense_route_floor(r, n, E, dims)                        # what the route's own passes allow (composite)
        fl['frac'] = fl['us'] / (ms * 1e3)
        rec['step_floor'] = fl
    del launches, run, b
    release()
    return rec


def op_level_record():
    """What a host that is not this package gets (include/tipk.h section 10): the two D-D layers through tipk_graph_build /
    tipk_graph_prepare_rgcn / tipk_rgcn_fwd / tipk_rgcn_bwd_ex alone, timed by tools/bench_c_abi.py in a CHILD process (ctypes +
    torch for device memory; neither tip_amd.ops nor tip_amd.plan) -- generic route and pair form."""
    import subprocess
    if any(k.startswith('ROCPROF') for k in os.environ) or 'rocprof' in os.environ.get('LD_PRELOAD', ''):
        return {'skipped': 'under rocprofv3 (a traced child would add its kernels to the profile of the step)'}
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'bench_c_abi.py'), '--json', '--steps', '50'], capture_output=True,
                           text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith('{')]
        if p.returncode != 0 or not line:
            return {'error': (p.stderr or p.stdout)[-400:]}
        return json.loads(line[-1])
    except Exception as exc:                                       # noqa: BLE001
        return {'error': repr(exc)}


def train_step_record(dev, epochs=20, decoder='distmult'):
    """The whole graphed training epoch of tip.py:24-30 (sampler + encoder + fused objective + backward + fused Adam)
    on BASELINE config 2: ms per epoch.  decoder = 'nn': the same epoch with the NNDecoder (model/ddm-nn.py:65-102)."""
    import torch
    from tip_amd.layers import TIP, Setting
    from tip_amd.train import GraphedTrainStep
    torch.manual_seed(1111)
    model = TIP(Setting(), dev, decoder=decoder)
    from tip_amd.optim import Adam
    opt = Adam(model.parameters(), lr=0.01)                        # tipk_adam_step: one launch, device-side step count
    step = GraphedTrainStep(model, opt)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(epochs):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / epochs * 1e3
    E = int(model.data.dd_train_idx.shape[1])
    loss = float(step())
    rec = {'ms_per_epoch': ms, 'edges_per_s': E / (ms * 1e-3), 'epochs_timed': epochs, 'loss_after': loss,
           'what': 'hipGraph replay of zero_grad + typed negative sampling + encoder + fused %s objective + backward + '
                   'Adam in one launch (tip_amd/optim.py, tip_amd/train.py), TIP-cat BioSNAP R=%d'
                   % ('DistMult' if decoder == 'distmult' else 'NNDecoder (score tables)', model.data.n_dd_et)}
    if decoder == 'distmult':
        rec['roofline'] = train_step_rooflines(model, dev)
    del step, opt, model
    release()
    return rec


def train_step_rooflines(model, dev):
    """The two launches of the training epoch that the encoder's headline does not see -- the fused objective and the typed
    negative sampler -- timed alone (HIP events around a hipGraph of back-to-back launches) on the epoch's own triple lists."""
    import torch
    from tip_amd import ops
    from tip_amd.neg_sampling import typed_negative_sampling
    d = model.data
    pos, et, rg = d.dd_train_idx, d.dd_train_et, d.dd_train_range
    n, k = d.n_drug, model.embeddings.shape[1] if torch.is_tensor(getattr(model, 'embeddings', None)) else 16
    E = int(pos.shape[1])
    z = torch.randn(n, k, device=dev) * 0.5
    w = torch.randn(d.n_dd_et, k, device=dev) * 0.3
    negp = typed_negative_sampling(pos, n, rg, packed=True)
    obj_us = time_launch_us(lambda: ops.distmult_loss(z, w, pos, negp, et), reps=10, replays=3)
    smp_us = time_launch_us(lambda: typed_negative_sampling(pos, n, rg, packed=True), reps=10, replays=3)
    # objective: every term of d z is one LDS row-add (16 lanes x ds_add_u64 = 128 bytes of read-modify-write) and one 64-byte row
    # read for the scatter; the evaluation reads both rows of a triple once more.  A symmetric relation's positives are
    # evaluated once per undirected pair (weight 2): E / 2 positive + E negative triples.
    triples = E // 2 + E
    lds_bytes = triples * 2 * (128.0 + 64.0) + triples * 2 * 4.0 * k
    atomics = triples * 2 / 4.0                                              # wave-level ds_add_u64: 4 rows per instruction
    obj = {'kernel': 'distmult_objective_kernel<PackedPair,%d> (+ the finalize launch)' % k, 'bound': 'lds', 'launch_us': obj_us,
           'achieved': lds_bytes / (obj_us * 1e-6) / 1e9, 'peak': LDS_PEAK_GBS, 'unit': 'GB/s',
           'frac': lds_bytes / (obj_us * 1e-6) / 1e9 / LDS_PEAK_GBS, 'algorithmic_lds_bytes': lds_bytes, 'row_adds': triples * 2,
           'atomic_floor_us': atomics * 8 / (256 * 2.4e9) * 1e6,
           'note': 'LDS bytes the sums need (row-adds as 128-byte read-modify-writes, 64-byte row reads) against the ds_read_b128 '
                   'rate; atomic_floor_us = the wave-level ds_add_u64 at their measured 8 cycles each (64 B/clk per CU) on 256 CUs'}
    # sampler: one Philox4x32-10 call per 4 positions = 10 rounds x (2 mul_hi + 2 mul_lo + 4 xor / add) on quarter-rate
    # multipliers; the floor below counts the 40 32-bit multiplies of a call at 16 lanes per cycle and SIMD
    calls = E / 4.0
    mul_cycles = calls * 40 / 64.0 * 16                                        # wave-instructions x 16 cycles (quarter rate)
    smp = {'kernel': 'neg_sample_bitmap_kernel<PackedOut>', 'bound': 'valu', 'launch_us': smp_us,
           'positions_per_s': E / (smp_us * 1e-6), 'philox_calls': calls,
           'multiply_floor_us': mul_cycles / (1024 * 2.4e9) * 1e6,
           'frac': mul_cycles / (1024 * 2.4e9) * 1e6 / smp_us,
           'note': 'VALU issue: the 40 quarter-rate 32-bit multiplies of a Philox4x32-10 call alone, on 1 024 SIMDs at 2.4 GHz; '
                   'the bitmap of a relation\'s positives (LDS) and the rejection loop come on top'}
    return {'objective': obj, 'sampler': smp}


def main():
    args = parse()
    if 'RANK' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))               # children first: this process never touches the GPU

    global _LIVE_PMC
    if (int(os.environ.get('WORLD_SIZE', '1')) == 1 and not args.step_only and not args.no_pmc and not os.environ.get('TIPK_BENCH_CHILD')
            and not os.environ.get('TIPK_FORCE_SHARD') and args.workload.startswith('biosnap')):
        _LIVE_PMC = measure_pmc_in_run(args)       # child processes: before this one touches the GPU
    import torch
    from tip_amd import _lib
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    import fcntl
    with open(os.path.join(ROOT, 'tip_amd', '.build.lock'), 'w') as lk:    # ranks started by a launcher: ONE `make` at a time
        fcntl.flock(lk, fcntl.LOCK_EX)                                     # (the first builds, the others find it fresh)
        _lib.ensure_built()                        # child `make` if the .so is missing / stale (before GPU use)
        fcntl.flock(lk, fcntl.LOCK_UN)
    args.gpus = world
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the product path has no CPU fallback)'
    n_dev = torch.cuda.device_count()
    shared = world > n_dev                         # --oversubscribe: ranks share devices
    if shared and not args.oversubscribe:
        raise SystemExit('%d ranks on %d GPU(s): pass --oversubscribe' % (world, n_dev))
    torch.cuda.set_device(local_rank % n_dev)
    dev = torch.device('cuda', local_rank % n_dev)

    import torch.distributed as dist
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to the C-level stdout (seen
    # after the JSON line, when its stdio buffer is flushed): while collectives may run, fd 1 points at
    # stderr; it is restored -- after flushing C stdio -- just before the result is printed.
    stdout_fd = None
    if world > 1 or os.environ.get('TIPK_FORCE_SHARD'):
        sys.stdout.flush()
        stdout_fd = os.dup(1)
        os.dup2(2, 1)
    backend = 'gloo' if shared else 'nccl'         # RCCL refuses two ranks on one device
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)

    from tip_amd import ops
    launch = args.launch or ('eager' if shared else 'graph')

    dd, dims, wl_name = make_workload(args)
    E = int(dd['dd_train_idx'].shape[1])
    R = dd['n_dd_et']
    sharded = world > 1 or bool(os.environ.get('TIPK_FORCE_SHARD'))       # (the env switch exercises the
    shard = None                                                           # collective path on one rank)
    if sharded:
        from tip_amd.dist import make_shard, shard_data_dict, attach_shard
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', str(free_port()))
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        shard = make_shard(dd['dd_train_range'], rank, world)
        dd_rank = shard_data_dict(dd, shard)                   # this rank's relations' edges only
    else:
        dd_rank = dd
    b = Bench(dd_rank, dims, args.mod, dev, shard)
    enc = b.enc
    if sharded:
        for name, prm in enc.named_parameters():               # identical replicas (att rows are shard-local)
            if not name.endswith('.att'):
                buf = prm.data.contiguous()                    # (conv weights are stored transposed: c10d wants contiguous)
                dist.broadcast(buf, 0)
                prm.data.copy_(buf)
        attach_shard(enc, shard)
        if world > 1 and 'TIPK_FWD_ROUTE' not in os.environ:
            ops.FWD_ROUTE_MODE = 'timed'                       # one timed, job-wide decision per layer (ops._fwd_route)
        coll_mode = os.environ.get('TIPK_COLLECTIVE', 'auto')              # auto | direct | group (rccl / gloo)
        if world > 1 and coll_mode != 'group':
            # The step's five all-reduces are 41 ... 430 KB: pure latency.  The one-shot exchange over peer-mapped mailboxes
            # (tip_amd/csrc/tipk_peer.hip) is set up if every rank can map every mailbox and a timed self-test sums right;
            # 'auto' then times it against the process group for the step's message sizes and keeps the faster per size
            # (before any capture; the same decision on every rank).  Anything that fails -> the process group.
            n_d, nb_ = dd['n_drug'], dims['num_base']
            d0 = dims['n_embed'] + dims['prot_drug_dim'] if args.mod == 'cat' else dims['n_embed']
            sizes = [n_d * dims['n_hid1'], n_d * dims['n_hid2'], n_d * d0 + nb_ * d0 * dims['n_hid1'],
                     n_d * dims['n_hid1'] + nb_ * dims['n_hid1'] * dims['n_hid2']]
            ex = shard.try_direct_exchange(dev, max_floats=max(1 << 16, max(sizes)))
            if ex is not None and coll_mode == 'auto':
                shard.choose_collective(sizes, dev)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # one-off initialisation that is not graph preprocessing (first use of torch's device kernels -- sort, bincount,
    # cumsum ... -- and of libtipk's code objects: ~0.4 s of lazy loading on a fresh process), timed on a toy graph and
    # reported as `init_s`; `preprocess_s` is then the plan build of the REAL graph (the first step)
    init_s = None
    if not sharded and not args.step_only:                     # (a --step-only child is a PMC pass: full-size steps only)
        from tip_amd.data import synthetic_data_dict
        t0 = time.perf_counter()
        toy = synthetic_data_dict(n_drug=96, n_rel=6, n_edges=6000, seed=1, with_protein_graph=True, n_prot=128, pp_edges=2048, dp_edges=256)
        tb = Bench(toy, dict(dims), args.mod, dev)
        tb.prepare()
        del tb, toy
        release()
        init_s = time.perf_counter() - t0
    preprocess_s = b.prepare()
    run = b.step
    if launch == 'graph':
        # If capture fails (e.g. a collective that cannot be captured on this RCCL build) every rank falls back to
        # eager launches together.
        ok = torch.ones(1, device=dev)
        try:
            run = b.capture(thread_local=dist.is_initialized())
        except Exception as exc:                               # noqa: BLE001
            sys.stderr.write('graph capture failed on rank %d (%r): eager launches\n' % (rank, exc))
            ok.zero_()
            torch.cuda.synchronize()
        if dist.is_initialized() and world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) <= 0:
            run, launch = b.step, 'eager'
    # the large D-D launches alone (roofline objects: HIP events around graphs of 20 back-to-back launches).  They run BEFORE
    # the headline region: a fresh process times its first milliseconds of GPU work at ramping clocks (20 steps = 7 ms;
    # measured 0.377 ms/step with the order reversed against 0.36 here and for --steps 100 either way)
    launches = [] if args.step_only else dd_launches(enc, dev)
    launch_us = {l['label']: time_launch_us(l['fn']) for l in launches if l['work'] is not None}
    settle = 0
    if launch == 'graph' and args.settle > 0 and world == 1:   # (untimed; the contract's W warm-up steps and K timed steps follow.
        t0 = time.perf_counter()                               # One rank only: an N-rank step holds collectives, and nothing that
        for _ in range(3):                                     # could make the ranks disagree on a count belongs in front of them)
            run()
        fence()
        per = (time.perf_counter() - t0) / 3
        settle = 3 + max(0, min(args.settle - 3, int(0.05 / max(per, 1e-6))))     # at most ~50 ms of them (config 5)
        for _ in range(settle - 3):
            run()
        fence()
    elapsed = timed(run, args.steps, args.warmup, fence)
    replay_stats = None
    if world == 1 and not args.step_only:
        replay_stats = replay_event_stats(run, max(100, args.steps), fence)
    multi = None
    if launch == 'graph' and world == 1 and not args.step_only and not sharded:
        try:                                                   # (after the headline region; its own graph)
            k_multi = 4
            gm = b.capture_many(k_multi)
            reps = max(2, args.steps // k_multi)
            el = timed(gm.replay, reps, 2, fence)
            multi = {'steps_per_graph': k_multi, 'ms_per_step': el / (reps * k_multi) * 1e3,
                     'note': 'the same step, %d per replayed hipGraph: the ~9 us hand-over between two replays is paid once per %d steps '
                             '(a training loop can do this: GraphedTrainStep(steps_per_replay=k)); NOT the headline' % (k_multi, k_multi)}
            del gm
        except Exception as exc:                               # noqa: BLE001
            multi = {'error': repr(exc)}

    # an eager per-kernel table of the whole step
    kern = {}
    if not args.no_kernel_table and not args.step_only:
        ops.timing_start()
        for _ in range(max(3, min(args.steps, 10))):
            b.step()
        fence()
        kern = ops.timing_stop()
    per_rank_ms, replicated = None, None
    if world > 1:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [float(t.item()) / args.steps * 1e3 for t in every]
        elapsed = max(float(t.item()) for t in every)
        if shard is not None and shard.direct is not None:
            shard.direct.check()                                           # a wait that ran out -> PeerTimeout: non-zero exit
    if sharded and args.workload.startswith('biosnap'):
        replicated = b.replicated_stage_us()                               # what every rank computes redundantly

    rc = 0
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        bid = _lib.build_id()
        out = {
            'metric': 'D-D edges aggregated/sec (encoder fwd+bwd)',
            'value': E * args.steps / elapsed, 'unit': 'edges/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'settle_replays': settle,
            'dtype': 'f32', 'data': 'BioSNAP graph (bundled), random-init weights, fixed N(0,1) upstream gradient'
            if args.workload.startswith('biosnap') else 'synthetic',
            'config': {'workload': wl_name, 'mod': args.mod, 'directed_dd_edges': E, 'relations': R,
                       'parallelism': 'relation-sharded x%d%s' % (world, ' (ranks share GPUs, gloo: functional check)'
                                                                  if shared else '') if world > 1 else 'single GPU',
                       'launch': 'hipGraph replay of the captured step' if launch == 'graph'
                       else 'eager (one ctypes call per kernel)',
                       'collective': shard.collective if shard is not None else None,
                       'collective_timing': shard.collective_report if shard is not None else None,
                       'rccl_ranks': dist.get_world_size() if (shard is not None and dist.is_initialized() and dist.get_backend() == 'nccl') else 0,
                       'forward_routes': [list(l._cache.value.fwd_route.values()) if l._cache.value is not None else None
                                          for l in (enc.rgcn1, enc.rgcn2)] if shard is not None else None},
            'preprocess_s': preprocess_s, 'init_s': init_s,
            'build_id': bid,
        }
        if replay_stats is not None:
            out['replay_events'] = replay_stats
        if multi is not None:
            out['multi_step_graph'] = multi
        if per_rank_ms is not None:
            out['per_rank_ms_per_step'] = [round(v, 4) for v in per_rank_ms]
        if replicated is not None:
            # P-P GCN x2 + P -> D + drug mix (forward and backward) are computed on EVERY rank: with N ranks the step cannot
            # get shorter than that.  est_single_rank = replicated + N x (this step - replicated): what one rank would need
            sh_us = max(0.0, ms * 1e3 - replicated)
            out['replicated'] = {'replicated_us': round(replicated, 1), 'sharded_and_collectives_us': round(sh_us, 1),
                                 'amdahl_ceiling': round((replicated + world * sh_us) / replicated, 2),
                                 'what': 'P-P / P->D / drug-mix stage of the encoder, graph-timed on rank 0; ceiling = speed-up '
                                         'over one rank that no number of ranks can exceed while this stage is replicated'}
        if launch_us:
            by = {l['label']: l for l in launches}
            dom = max(launch_us, key=launch_us.get)                               # the longest kernel of the step
            out['roofline'] = roofline_of(by[dom], launch_us[dom], bid)
            aggs = {k: v for k, v in launch_us.items() if by[k].get('aggregation')}
            if aggs:
                dom_a = max(aggs, key=aggs.get)
                out['roofline_aggregation'] = roofline_of(by[dom_a], aggs[dom_a], bid)
            out['dd_launches_us'] = {k: round(v, 2) for k, v in launch_us.items()}
        per_edge = sum(2 * (8 + 4 * dims[k]) for k in ('n_hid1', 'n_hid2'))     # 416 / 2080 B per edge
        if args.workload.startswith('synthetic'):
            # the HBM yardstick of SURVEY 8(d) against the roofline of ALL the GPUs of the job (N x 8 TB/s)
            out['whole_step'] = {'alg_bytes': E * per_edge, 'bytes_per_edge': per_edge, 'GBps': E * per_edge / (ms * 1e-3) / 1e9,
                                 'n_gpus': world, 'peak_GBps': world * HBM_PEAK_GBS,
                                 'frac_of_roofline': E * per_edge / (ms * 1e-3) / 1e9 / (world * HBM_PEAK_GBS)}
            if world == 1:
                if any(l['label'].startswith('row_products') for l in launches):
                    fl = row_route_floor(dd['n_dd_et'], dd['n_drug'], E,
                                         [(enc.rgcn1.in_channels, enc.rgcn1.out_channels), (enc.rgcn2.in_channels, enc.rgcn2.out_channels)],
                                         dims['num_base'])
                else:
                    fl = dense_route_floor(dd['n_dd_et'], dd['n_drug'], E, dims)
                fl['frac'] = fl['us'] / (ms * 1e3)
                out['step_floor'] = fl
        elif launch_us:
            # (N > 1: rank 0's own launches on ITS shard of the relations -- the floor of one rank's step, collectives at the
            # launch floor like every launch that is not modelled)
            pp_edges = int(dd['pp_train_indices'].shape[1]) + dd['n_prot']
            rows_graph = getattr(getattr(getattr(enc, 'pp_encoder', None), 'conv2', None), '_cache_rows', None)
            rows_graph = rows_graph.value if rows_graph is not None else None
            fl = step_floor(launch_us, launches, kern, pp_edges, dims,
                            pp_rows_edges=int(rows_graph.fwd.n_edges) if rows_graph is not None else None)
            fl['frac'] = fl['us'] / (ms * 1e3)
            if world > 1:
                fl['of'] = 'rank 0 (its shard of the relations; the step time is the max over the ranks)'
            out['step_floor'] = fl
            hb = step_hbm_bytes(bid) if world == 1 else None
            if hb is not None:
                out['step_hbm'] = {'bytes_measured': hb[0], 'bytes_measured_uncorrected': hb[2],
                                   'frac_of_8TBps': hb[0] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   'source': 'rocprofv3 --pmc passes of this run' if hb[1] == 'this run' else 'profiles/' + hb[1] + ' (same build id)'}
            # SURVEY 8(d)'s yardstick (416 B per directed edge: one d_out-wide row gathered per edge and pass) next to what the
            # step really moves: the pair form reads half the edges, rows out of LDS -- the yardstick saturates (> 1) and does
            # not describe this step; `roofline` (LDS / MFMA bounds per kernel) and `step_floor` do
            out['whole_step'] = {'algorithmic_bytes': E * per_edge, 'bytes_per_edge': per_edge,
                                 'algorithmic_GBps': E * per_edge / (ms * 1e-3) / 1e9,
                                 'algorithmic_hbm_frac': E * per_edge / (ms * 1e-3) / 1e9 / (world * HBM_PEAK_GBS),
                                 'measured_hbm_frac': (hb[0] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if hb is not None else None,
                                 'note': 'algorithmic_hbm_frac prices SURVEY 8(d)\'s per-edge bytes against 8 TB/s; values above 1 mean '
                                         'the rows never came from HBM (LDS-resident tables, pair form): measured_hbm_frac is the '
                                         'counter figure of the same step'}
            if 'roofline' in out:
                out['roofline']['algorithmic_hbm_frac_whole_step'] = out['whole_step']['algorithmic_hbm_frac']
                out['roofline']['measured_hbm_frac_whole_step'] = out['whole_step']['measured_hbm_frac']
        if kern:
            out['kernels_eager_ms'] = {
                'note': 'one HIP event pair per EAGER launch: includes ~5-8 us of event/launch overhead each; '
                        'kernel-only times: profiles/*_kernel_by_grid.csv',
                'table': {k: {'launches': v[0], 'mean_ms': round(v[1], 5)}
                          for k, v in sorted(kern.items(), key=lambda kv: -kv[1][0] * kv[1][1])}}
        if world == 1 and not args.no_cpu_baseline and not args.step_only:
            if launch == 'graph':
                b.graph.replay()                                   # the timed object itself: one more replay of the captured step
                torch.cuda.synchronize()
            z_dev, grads = b.outputs()
            out['cpu_baseline'], zo, go = cpu_baseline(dd, dims, args.mod, args.cpu_seconds, b.weights_cpu(), b.g_up_cpu)
            out['parity_in_bench'] = parity_check(z_dev, grads, zo, go)
            out['parity_in_bench']['what'] = 'outputs of %s' % ('a replay of the timed hipGraph' if launch == 'graph' else 'an eager step')
            if not out['parity_in_bench']['ok']:
                rc = 1
        extras = (world == 1 and not args.no_extras and not args.step_only and not args.no_cpu_baseline
                  and args.workload == 'biosnap' and args.mod == 'cat')
        if extras:
            del launches, run
            b.graph = None
            del b, enc
            release()
            out['other_configs'] = {}
            for key, (wl, mod) in (('tip_add', ('biosnap', 'add')), ('biosnap963', ('biosnap963', 'cat')), ('synthetic', ('synthetic', 'cat'))):
                try:
                    out['other_configs'][key] = measure_config(wl, mod, dev)
                except Exception as exc:                           # noqa: BLE001 -- never lose the headline line
                    out['other_configs'][key] = {'error': repr(exc)}
                    release()
            for key, dec in (('train_step', 'distmult'), ('train_step_nn_decoder', 'nn')):
                try:
                    out[key] = train_step_record(dev, decoder=dec)
                except Exception as exc:                           # noqa: BLE001
                    out[key] = {'error': repr(exc)}
                    release()
            out['op_level_c_abi'] = op_level_record()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if stdout_fd is not None:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)                         # the banner (if any) goes to stderr
        os.dup2(stdout_fd, 1)
        os.close(stdout_fd)
    if rank == 0:
        print(json.dumps(out), flush=True)
    sys.exit(rc)


if __name__ == '__main__':
    main()
