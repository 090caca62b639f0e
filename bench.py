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
  whole_step    algorithmic bytes of the whole step (416 B/edge BioSNAP, 2080 B/edge synthetic) over
                the measured step time, as a fraction of the 8 TB/s HBM spec.
  cpu_baseline  the oracle's CPU port on the host cores (rank 0, N = 1 only), bounded sample.
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='biosnap', choices=['biosnap', 'biosnap963', 'synthetic', 'synthetic-small'])
    ap.add_argument('--mod', default='cat', choices=['cat', 'add'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='budget of the CPU baseline leg')
    ap.add_argument('--chunk', type=int, default=None, help='gather plan chunk (edges per work item)')
    ap.add_argument('--launch', default=None, choices=['graph', 'eager'],
                    help='graph: the step is captured once into a hipGraph and replayed (default); '
                         'eager: one ctypes launch per kernel')
    ap.add_argument('--oversubscribe', action='store_true',
                    help='allow more ranks than GPUs (ranks share devices, gloo collectives): functional check only')
    ap.add_argument('--no-kernel-table', action='store_true', help='skip the eager per-kernel event pass')
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


def cpu_baseline(dd, dims, mod, budget_s):
    """The oracle (CPU port of the same algorithm: transform-then-gather with explicit backward)
    on this host's cores, plus the reference-shaped op sequence (PyG-CPU path: lift E x in,
    per-relation slice+mm, cat, scatter-mean, autograd backward) on a bounded relation sample."""
    import torch
    from oracle import tip_oracle as O
    E = dd['dd_train_idx'].shape[1]
    R = dd['n_dd_et']
    p = O.init_params(dd['n_drug'], dd['n_prot'], R, mod=mod, seed=1111, prot_drug_dim=dims['prot_drug_dim'],
                      n_embed=dims['n_embed'], n_hid1=dims['n_hid1'], n_hid2=dims['n_hid2'],
                      num_base=dims['num_base'])
    up = torch.randn(dd['n_drug'], dims['n_hid2'], generator=torch.Generator().manual_seed(0))
    # torch's default (= all hardware threads) oversubscribes these small ops on big hosts
    threads = max(1, min(16, os.cpu_count() or 1))
    torch.set_num_threads(threads)

    def step():
        z, saved = O.fm_encoder_fwd(p, dd, mod)
        O.fm_encoder_bwd(up, p, dd, saved, mod)
    step()                                                    # warm-up (page-in, thread pool)
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
        r_s = min(R, 24)
        e_s = int(dd['dd_train_range'][r_s - 1, 1])
        ei = dd['dd_train_idx'][:, :e_s]
        rg = dd['dd_train_range'][:r_s]
        d_in = dims['n_embed'] + dims['prot_drug_dim'] if mod == 'cat' else dims['n_embed']
        x = torch.randn(dd['n_drug'], d_in).requires_grad_(True)
        prm = [p['rgcn1.basis'][:, :d_in].clone().requires_grad_(True), p['rgcn1.att'][:r_s].clone().requires_grad_(True),
               p['rgcn1.root'][:d_in].clone().requires_grad_(True)]
        g = torch.randn(dd['n_drug'], prm[0].shape[2])
        t0 = time.perf_counter()
        y = O.rgcn_fwd_reference_shaped(x, ei, rg, *prm)
        y.backward(g)
        dt_ref = time.perf_counter() - t0
        out['reference_shaped'] = {
            'value': e_s / dt_ref, 'unit': 'edges/s (one R-GCN layer fwd+bwd, on the SAMPLE)', 'cores': threads,
            'sample': 'first %d relations, %d edges, PyG op sequence under autograd, %.2f s' % (r_s, e_s, dt_ref),
            # time ~ c * R_s * E_s  ->  full graph: E / (c R E) = sample rate * R_s / R
            'full_size_estimate': e_s / dt_ref * r_s / R,
            'full_size_estimate_note': 'sample rate x R_sample / R (cost model c*R*E of the slice backward; the survey '
                                       'probe of the literal reference measured 0.020 M edges/s on 8 cores)'}
    except Exception as exc:                                   # never let the side leg kill the bench
        out['reference_shaped'] = {'error': repr(exc)}
    return out


# ---------------------------------------------------------------------------------------------
# the D-D aggregation launches of the step, timed alone (roofline object)
# ---------------------------------------------------------------------------------------------
def dd_aggregation_launches(enc, dev):
    """[(label, kernel key for profiles/, grid 'XxYxZ', d, callable)] for the four D-D aggregations
    of one step, on the step's own plans with random tables of the step's shapes."""
    import torch
    from tip_amd import ops
    out = []
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
        y = torch.randn(r * n, d, device=dev)
        g = torch.randn(n, d, device=dev)
        nb = layer.num_bases
        for bwd in (False, True):
            rs = graph.rs_bwd if bwd else None
            pair = None if bwd or os.environ.get('TIPK_NO_PAIR_FWD') else graph.pair_fwd
            if pair is not None and pair.n_table == r and ops.stream_gather_split(r, nb):
                # forward in pair form: per edge one id + one att row (nb floats) from LDS; the dense product that
                # follows is a separate launch (gemm[...] in kernels_eager_ms)
                split = ops.stream_gather_split(r, nb)
                key = 'stream_gather_kernel<%d, %s, 1' % (nb // split // 4, 'true' if pair.idx_unit == nb // split * 4 else 'false')
                grid = '%dx%dx1' % (pair.n_wg * 1024, split)
                att = torch.randn(r, nb, device=dev)
                cells = graph.pair_buffers(n, nb, d, dev)[0]
                out.append(('pair_cells[dd.fwd,d=%d]' % d, key, grid, nb, 'lds',
                            lambda pair=pair, att=att, cells=cells, nb=nb: ops.stream_gather(
                                pair, att, write_zeros=False, out=cells.view(-1, nb)[:n * n], kind=1), pair.n_edges))
            elif rs is not None and ops.rel_stream_split(n, d):
                split = ops.rel_stream_split(n, d)
                key = 'stream_gather_kernel<%d, %s, 0' % (d // split // 4, 'true' if rs.idx_unit == d // split * 4 else 'false')
                grid = '%dx%dx1' % (rs.n_wg * 1024, split)
                out.append(('rel_stream[dd.bwd,d=%d]' % d, key, grid, d, 'lds',
                            lambda rs=rs, g=g: ops.rel_stream_bwd(rs, g, row_scale=graph.scale), rs.n_edges))
            elif ops.rel_gather_usable(graph.rl_bwd if bwd else graph.rl_fwd, n, d, bwd):
                rp = graph.rl_bwd if bwd else graph.rl_fwd
                split = ops.rel_gather_split(n, d, bwd)
                key = 'rel_gather_kernel<%d, %s' % (d // split // 4, 'true' if bwd else 'false')
                grid = '%dx%dx1' % (rp.n_wg * 1024, split)
                fn = (lambda rp=rp, g=g: ops.rel_gather(rp, g, True, row_scale=graph.scale)) if bwd else \
                     (lambda rp=rp, y=y: ops.rel_gather(rp, y, False, reduce=False))
                out.append(('rel_gather[dd.%s,d=%d]' % ('bwd' if bwd else 'fwd', d), key, grid, d, 'lds', fn, None))
            else:
                lanes = 1
                while lanes < d // 4:
                    lanes *= 2
                if bwd and d % 4 == 0 and 8 <= d <= 256 and not os.environ.get('TIPK_NO_CSR'):
                    csr = graph.csr_bwd                        # the path _RGCN.backward takes on large graphs
                    lanes = max(lanes, 2)
                    rp = lanes - 1 if lanes <= 16 else 16
                    tasks = -(-csr.n_out // rp)
                    waves = -(-tasks // (64 // lanes))
                    key = 'gather_rows_csr_kernel<%d' % lanes
                    grid = '%dx1x1' % (-(-waves // 4) * 256)
                    out.append(('gather_rows_csr[dd.bwd,d=%d]' % d, key, grid, d, 'hbm',
                                lambda csr=csr, g=g: ops.gather_rows_csr(csr, g), None))
                    continue
                plan = graph.bwd if bwd else graph.fwd
                waves = -(-plan.items.shape[0] // (64 // lanes))
                key = 'gather_sum_kernel<4, %d' % lanes
                grid = '%dx1x1' % (-(-waves // 4) * 256)
                fn = (lambda plan=plan, g=g: ops.gather_sum(plan, g)) if bwd else (lambda plan=plan, y=y: ops.gather_sum(plan, y))
                out.append(('gather_sum[dd.%s,d=%d]' % ('bwd' if bwd else 'fwd', d), key, grid, d, 'hbm', fn, None))
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


def pmc_traffic(key_prefix, grid):
    """(HBM bytes per launch, trace us, source file) of a kernel from the newest committed summaries
    (profiles/*_pmc_traffic.json / *_kernel_by_grid.csv: separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes and the --kernel-trace of this command; bytes corrected as
    MI355X_MICROARCH.md prescribes).  Entries are keyed by kernel name and the FULL grid XxYxZ."""
    import csv
    import glob
    traffic = trace_us = src = None
    try:
        for fn in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')), reverse=True):
            for name, k in json.load(open(fn))['kernels'].items():
                if name.startswith(key_prefix) and name.endswith('grid=' + grid):
                    traffic, src = k['hbm_bytes_per_launch'], os.path.basename(fn)
                    break
            if traffic is not None:
                break
        for fn in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_kernel_by_grid.csv')), reverse=True):
            rows = [r for r in csv.reader(l for l in open(fn) if not l.startswith('#'))][1:]
            hit = [r for r in rows if r[0].startswith(key_prefix) and r[0].endswith('grid=' + grid)]
            if hit:
                trace_us = float(hit[0][2]) / 1e3
                break
    except Exception:
        pass
    return traffic, trace_us, src


def step_hbm_bytes():
    """(HBM bytes of ONE step, file) from the newest committed PMC summary that has the total
    (tools/summarize_prof.py: all libtipk launches of `bench.py --step-only` / (steps + warmup))."""
    import glob
    try:
        for fn in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')), reverse=True):
            v = json.load(open(fn)).get('step_hbm_bytes')
            if v:
                return float(v), os.path.basename(fn)
    except Exception:
        pass
    return None


def main():
    args = parse()
    if 'RANK' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))               # children first: this process never touches the GPU

    import torch
    from tip_amd import _lib
    _lib.ensure_built()                            # child `make` if the .so is missing / stale (before GPU use)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
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
    from tip_amd.data import Data
    from tip_amd.layers import FMEncoder
    if args.chunk:
        os.environ['TIPK_CHUNK'] = str(args.chunk)
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
    torch.manual_seed(1111)
    enc = FMEncoder(dev, dd['n_drug_feat'], dd_rank['n_dd_et'], dd['n_prot'], dd['n_prot'], dd['n_drug'],
                    mod=args.mod, **dims).to(dev)
    if sharded:
        for name, prm in enc.named_parameters():               # identical replicas (att rows are shard-local)
            if not name.endswith('.att'):
                buf = prm.data.contiguous()                    # (conv weights are stored transposed: c10d wants contiguous)
                dist.broadcast(buf, 0)
                prm.data.copy_(buf)
        attach_shard(enc, shard)
    d = Data.from_dict({k: v for k, v in dd_rank.items() if k != 'dd_edge_index'}).to(dev)
    g_up = torch.randn(dd['n_drug'], dims['n_hid2'], generator=torch.Generator().manual_seed(0)).to(dev)

    def step():
        for prm in enc.parameters():
            prm.grad = None
        z = enc(d.d_feat, d.dd_train_idx, d.dd_train_et, d.dd_train_range, d.d_norm, d.p_feat,
                d.pp_train_indices, d.dp_edge_index, d.dp_range_list)
        z.backward(g_up)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    t0 = time.perf_counter()
    step()                                                     # builds + caches all gather plans
    torch.cuda.synchronize()
    preprocess_s = time.perf_counter() - t0
    run = step
    if launch == 'graph':
        # the whole step (about 36 kernels, plus the RCCL all-reduces when sharded) becomes one
        # hipGraph: replay removes the per-launch host cost, which is larger than the kernels
        # themselves at BioSNAP scale.  If capture fails (e.g. a collective that cannot be captured on
        # this RCCL build) every rank falls back to eager launches together.
        ok = torch.ones(1, device=dev)
        graph = None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            # with a process group alive the RCCL watchdog thread makes HIP calls of its own: only this
            # thread's calls may invalidate the capture
            mode = {'capture_error_mode': 'thread_local'} if dist.is_initialized() else {}
            with torch.cuda.graph(graph, **mode):
                step()
        except Exception as exc:                               # noqa: BLE001
            sys.stderr.write('graph capture failed on rank %d (%r): eager launches\n' % (rank, exc))
            ok.zero_()
            torch.cuda.synchronize()
        if dist.is_initialized() and world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) > 0:
            run = graph.replay
        else:
            launch = 'eager'
    for _ in range(args.warmup):
        run()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    fence()
    elapsed = time.perf_counter() - t0

    # the D-D aggregation launches alone (roofline), then an eager per-kernel table of the whole step
    agg_us = {}
    launches = [] if args.step_only else dd_aggregation_launches(enc, dev)
    for label, key, grid, d_row, bound, fn, n_e in launches:
        agg_us[label] = time_launch_us(fn)
    kern = {}
    if not args.no_kernel_table and not args.step_only:
        ops.timing_start()
        for _ in range(max(3, min(args.steps, 10))):
            step()
        fence()
        kern = ops.timing_stop()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            'metric': 'D-D edges aggregated/sec (encoder fwd+bwd)',
            'value': E * args.steps / elapsed, 'unit': 'edges/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'BioSNAP graph (bundled), random-init weights, fixed N(0,1) upstream gradient'
            if args.workload.startswith('biosnap') else 'synthetic',
            'config': {'workload': wl_name, 'mod': args.mod, 'directed_dd_edges': E, 'relations': R,
                       'parallelism': 'relation-sharded x%d%s' % (world, ' (ranks share GPUs, gloo: functional check)'
                                                                  if shared else '') if world > 1 else 'single GPU',
                       'launch': 'hipGraph replay of the captured step' if launch == 'graph'
                       else 'eager (one ctypes call per kernel)'},
            'preprocess_s': preprocess_s,
            'build_id': _lib.build_id(),
        }
        # dominant kernel = the D-D aggregation launch with the longest duration
        if agg_us:
            dom = max(agg_us, key=agg_us.get)
            label, key, grid, d_row, bound, _, n_launch = [l for l in launches if l[0] == dom][0]
            # edges the launch walks: rank 0's share when sharded; a pair-form launch of a symmetric graph half of them
            n_edges = int(n_launch) if n_launch is not None else int(dd_rank['dd_train_idx'].shape[1])
            # one id + one d-wide fp32 row per edge and pass (ids are 4 B in the generic
            # plans, 2 B in the relation-local ones; the figure keeps 4 B so runs stay comparable)
            alg_bytes = n_edges * (4 + 4 * d_row)
            us = agg_us[dom]
            achieved = alg_bytes / (us * 1e-6) / 1e9
            peak = LDS_PEAK_GBS if bound == 'lds' else HBM_PEAK_GBS
            traffic, trace_us, src = pmc_traffic(key, grid)
            roof = {'bound': bound, 'kernel': dom, 'grid': grid, 'achieved': achieved, 'peak': peak, 'unit': 'GB/s',
                    'frac': achieved / peak, 'traffic': traffic, 'launch_us': us,
                    'timing': 'HIP events on the launch stream around a hipGraph of 20 back-to-back launches',
                    'algorithmic_bytes_per_launch': alg_bytes, 'edges_per_launch': n_edges, 'row_floats': d_row}
            if traffic is not None:
                roof['hbm_traffic_frac'] = traffic / (us * 1e-6) / 1e9 / HBM_PEAK_GBS
                roof['traffic_source'] = 'profiles/' + src
            if trace_us is not None:
                roof['rocprof_trace_us'] = trace_us
            if bound == 'lds':
                roof['note'] = ('rows are gathered from LDS (wave-stream / relation-local kernel): the bound is the ds_read_b128 '
                                'rate; HBM only carries the ids and the output rows')
            out['roofline'] = roof
            out['dd_aggregations_us'] = {k: round(v, 2) for k, v in agg_us.items()}
        per_edge = sum(2 * (8 + 4 * dims[k]) for k in ('n_hid1', 'n_hid2'))     # 416 / 2080 B per edge
        out['whole_step'] = {'alg_bytes': E * per_edge, 'bytes_per_edge': per_edge,
                             'GBps': E * per_edge / (ms * 1e-3) / 1e9,
                             'frac_of_8TBps': E * per_edge / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             'note': 'SURVEY 8(d) prices one id + one d-wide row per edge and pass from HBM; this build serves the '
                                     'rows from LDS and, in the pair-form forward, adds one att row per edge of HALF the '
                                     '(symmetric) graph, so the figure can exceed 1: it compares against the survey '
                                     'yardstick, it is not an HBM utilisation'}
        hb = step_hbm_bytes()
        if hb is not None and args.workload.startswith('biosnap') and world == 1:
            out['whole_step']['hbm_bytes_measured'] = hb[0]
            out['whole_step']['hbm_frac_measured'] = hb[0] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            out['whole_step']['hbm_source'] = 'profiles/' + hb[1]
        if kern:
            out['kernels_eager_ms'] = {
                'note': 'one HIP event pair per EAGER launch: includes ~5-8 us of event/launch overhead each; '
                        'kernel-only times: profiles/*_kernel_by_grid.csv',
                'table': {k: {'launches': v[0], 'mean_ms': round(v[1], 5)}
                          for k, v in sorted(kern.items(), key=lambda kv: -kv[1][0] * kv[1][1])}}
        if world == 1 and not args.no_cpu_baseline and not args.step_only:
            out['cpu_baseline'] = cpu_baseline(dd, dims, args.mod, args.cpu_seconds)
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


if __name__ == '__main__':
    main()
