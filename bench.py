"""bench.py — render + Chamfer forward+backward images/s on MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch of synthetic input already resident in HBM:
  sampler fwd (Philox in-kernel) -> Chamfer(pred, gt) fwd -> raster fwd (silhouette+depth)
  -> L1(sil) + L1(depth) -> backward of all of it to d/d(v,q,t) [-> RCCL all-gather of the per-rank gradient slices if N>1].
Workload (config.workload): BASELINE configs[2] = C3: B=64 per GPU, K=32 sphere primitives,
256x256, n=256 points per primitive (N=8192) vs M=2048 GT points.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3     # fp32 vector peak (packed FMA)


def synth_inputs(B, K, M, seed, device):
    """SURVEY.md 8d synthetic inputs (seed 1234 = reference config.py:20)."""
    g = torch.Generator().manual_seed(seed)
    v = (torch.rand(B, K, 3, generator=g) + 0.1) / torch.tensor([8.0, 10.0, 10.0])   # vpnet_one_resnet.py:71,84
    q = torch.rand(B, K, 4, generator=g)                                               # sigmoid range (:72)
    t = 0.35 * (torch.rand(B, K, 3, generator=g) * 2 - 1)                              # tanh range (:73), inside the frustum
    gt_points = torch.rand(B, M, 3, generator=g) - 0.5                                 # dataset.py:165
    return torch.cat([v, q, t], 2).to(device), gt_points.to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=64, help='samples per GPU')
    ap.add_argument('--prims', type=int, default=32)
    ap.add_argument('--points', type=int, default=256, help='sampled points per primitive')
    ap.add_argument('--gt-points', type=int, default=2048)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='launch every step eagerly instead of replaying a HIP graph')
    ap.add_argument('--dist-selftest', action='store_true',
                    help='run the multi-rank code path (RCCL group, gradient all-gather) even with one rank')
    ap.add_argument('--cpu-sample', type=int, default=32, help='images in the CPU baseline sample')
    args = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d'
                     % (args.gpus, args.gpus))
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    multi = world > 1 or args.dist_selftest
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    import vpn_amd
    from vpn_amd import _lib
    from vpn_amd.dist import GradAllGather
    _lib.lib()

    B, K, n, M, H = args.batch, args.prims, args.points, args.gt_points, args.size
    W = H
    kinds = vpn_amd.kinds_tensor([vpn_amd.SPHERE] * K, dev)      # reference default: all spheres (config.py:33-34)
    # each rank owns global samples [rank*B, (rank+1)*B): weak scaling, per-GPU work fixed
    params_all, gt_all = synth_inputs(B * world, K, M, 1234, dev)
    params = params_all[rank * B:(rank + 1) * B].clone().requires_grad_(True)
    gt_points = gt_all[rank * B:(rank + 1) * B].contiguous()
    cam = torch.tensor([[1.0, 0.0, 0.0]], device=dev).expand(B, 3).contiguous()    # train.py:172-174
    # GT silhouette / depth: render of a second primitive set (seed 4321), silhouette thresholded at 0.5
    p2, _ = synth_inputs(B * world, K, M, 4321, dev)
    with torch.no_grad():
        a2, d2 = vpn_amd.RasterFunction.apply(p2[rank * B:(rank + 1) * B].contiguous(), kinds, cam, H, W,
                                              vpn_amd.config.RASTER_SIGMA, vpn_amd.config.RASTER_GAMMA,
                                              vpn_amd.config.RASTER_Z_FAR)
    gt_sil = (a2 > 0.5).float()
    gt_depth = d2.clone()
    reducer = GradAllGather(B * world, K, dev, rank, world) if multi else None
    cd_fn = vpn_amd.ChamferDistanceLoss()
    sigma, gamma, z_far = vpn_amd.config.RASTER_SIGMA, vpn_amd.config.RASTER_GAMMA, vpn_amd.config.RASTER_Z_FAR

    unit_total = torch.tensor([0.0, 0.0, 1.0], device=dev)

    def compute(i):
        # total = ChamferDistanceLoss(sample(params), gt) + SilhouetteLoss(L1) + L1 depth loss  (train.py:243-262),
        # one autograd node: sampler -> Chamfer scans -> raster with fused image losses, and the matching backward
        params.grad = None
        out = vpn_amd.HotPathLossFunction.apply(params, kinds, cam, gt_points, gt_sil, gt_depth, n, 1234 + i,
                                                rank * B, H, W, sigma, gamma, z_far, 1.0, 1.0, 1.0)
        # d(total)/d(params): the unit vector selects the total (index 2) of the three losses.  `out[2].backward()`
        # is the same thing through three more ATen kernels (ones_like, and SelectBackward's zeros + copy).
        out.backward(unit_total)
        return out[2]

    def step(i):
        loss = compute(i)
        if reducer is not None:
            return reducer.reduce(params.grad, loss)
        return params.grad, loss

    def sync():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # The step is launch-bound on the host when issued eagerly (about 20 launches of 5-200 us), so the
    # compute part is captured once into a HIP graph and replayed; the RCCL all-gather stays outside.
    use_graph = not args.no_graph
    run_step = step
    if use_graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(3):
                compute(i)                       # warm the allocator / autograd on the capture stream
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        params.grad = None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            g_loss = compute(0)
            if reducer is not None:
                reducer.pack(params.grad, g_loss)          # scaled copy into the send buffer: part of the graph
        g_grad = params.grad

        def run_step(i):
            graph.replay()
            if reducer is not None:
                reducer.gather()                           # the one collective of the step, outside the graph
                return reducer.views()                     # (overlapping it with the next step through a second
            return g_grad, g_loss                          #  graph + communication stream measured slower at 1 rank)

    for i in range(args.warmup):
        run_step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        run_step(args.warmup + i)
    sync()
    dt = time.perf_counter() - t0
    if multi:
        tmax = torch.tensor([dt], device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    ms_per_step = dt / args.steps * 1e3
    value = B * world * args.steps / dt

    # ---- device time per C-ABI entry point and per KERNEL (HIP events on the launch stream, recorded by the
    # binding / by the library around every launch), same steps again, eagerly
    ksteps = max(5, min(args.steps, 20))
    with _lib.KernelTimer() as kt, _lib.KernelProfile() as kp:
        for i in range(ksteps):
            step(i)
        entry = kt.summary()            # entry point -> (calls, mean ms)
    kern = kp.summary()                 # kernel      -> (calls, mean ms)
    entry_us = {k: round(v[1] * 1e3, 2) for k, v in entry.items()}
    kernel_us = {k: {'calls_per_step': round(v[0] / ksteps, 2), 'avg_us': round(v[1] * 1e3, 2)} for k, v in kern.items()}

    # algorithmic bytes / flops per launch (SURVEY.md 8d per-image figures x B)
    N = K * n
    nn_bytes = B * 16 * (N + M)                       # mean of the two directions: both clouds in, dist + idx out
    alg_bytes = {
        'chamfer_nn_mfma_kernel<1>': 2 * nn_bytes, 'chamfer_nn_mfma_kernel<0>': 2 * nn_bytes,     # one launch = both directions
        'chamfer_nn_kernel<R>': nn_bytes,
        'chamfer_nn_pruned_kernel<1>': nn_bytes,
        'raster_fwd_kernel<0>': B * (40 * K + 8 * H * W), 'raster_fwd_kernel<1>': B * (40 * K + 8 * H * W),
        'raster_bwd_kernel<0>': B * (8 * H * W + 80 * K), 'raster_bwd_kernel<1>': B * (8 * H * W + 80 * K),
        'sample_fwd_kernel': B * (40 * K + 12 * N), 'sample_bwd_kernel': B * (12 * N + 80 * K),
        'chamfer_bwd_lds_kernel': B * (12 * (N + M) + 8 * (N + M) + 12 * N),
        'sample_chamfer_bwd_kernel': B * (12 * (N + M) + 8 * (N + M) + 80 * K),
    }
    pair_flops = 8.0 * B * N * M                      # 8 flop per point pair (3 sub, 3 mul, 2 add  ==  K=4 MAC on the matrix pipe)
    dom = max((k for k in kern if k in alg_bytes), key=lambda k: kern[k][0] * kern[k][1])
    dom_s = kern[dom][1] * 1e-3
    hbm_gbs = alg_bytes[dom] / dom_s / 1e9
    if dom.startswith('chamfer_nn_mfma_kernel'):      # the exact scan with the matrix-pipe filter, both directions per launch
        pair_flops *= 2.0
        tf = pair_flops / dom_s / 1e12
        bf16 = dom.endswith('<1>')
        roofline = {'bound': 'mfma', 'kernel': dom, 'achieved': round(tf, 2), 'peak': VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(tf / VALU_PEAK_TFLOPS, 4), 'traffic': None,
                    'algorithmic_flops_per_launch': pair_flops, 'avg_launch_us': round(dom_s * 1e6, 2),
                    'peak_note': 'algorithmic work = 8 fp32 flop per point pair, priced against the fp32 rate of MI355X '
                                 '(157.3 TFLOP/s, vector = fp32-input MFMA; MI355X_MICROARCH.md)',
                    'hbm_view': {'algorithmic_bytes_per_launch': alg_bytes[dom], 'achieved_GBps': round(hbm_gbs, 2),
                                 'frac_of_8TBps': round(hbm_gbs / HBM_PEAK_GBS, 5)}}
        if bf16:    # executed on the bf16 matrix pipe: v_mfma_f32_32x32x16_bf16 + v_mfma_f32_32x32x8_bf16 per 32x32 pairs = 48 flop per pair
            ex = 2.0 * 48.0 * B * N * M / dom_s / 1e12
            roofline['matrix_pipe'] = {'instruction': 'v_mfma_f32_32x32x16_bf16 + v_mfma_f32_32x32x8_bf16 (fp32 coordinates split exactly into 3 bf16 pieces, 21 of 24 K slots used)',
                                       'executed_TFLOPs': round(ex, 1), 'dense_bf16_peak_TFLOPs': 2500.0,
                                       'frac': round(ex / 2500.0, 4)}
    else:
        roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': round(hbm_gbs, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(hbm_gbs / HBM_PEAK_GBS, 5), 'traffic': None,
                    'algorithmic_bytes_per_launch': alg_bytes[dom], 'avg_launch_us': round(dom_s * 1e6, 2)}
        if dom.startswith('chamfer_nn'):
            roofline['valu_tflops'] = round(pair_flops / dom_s / 1e12, 2)
            roofline['valu_frac_of_fp32_peak'] = round(pair_flops / dom_s / 1e12 / VALU_PEAK_TFLOPS, 4)

    # HBM traffic of the dominant kernel from the committed PMC passes (rocprofv3 cannot run inside bench.py)
    try:
        tr = json.load(open(os.path.join(ROOT, 'profiles', 'r01j_traffic.json')))['kernels'].get(dom)
        if tr and (B, K, n, M, H) == (64, 32, 256, 2048, 256):
            roofline['traffic'] = tr['traffic_bytes']
            roofline['traffic_source'] = 'profiles/r01j_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, same workload)'
    except (OSError, ValueError, KeyError):
        pass

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(params_all[:args.cpu_sample].detach().cpu(), gt_all[:args.cpu_sample].cpu(),
                           gt_sil[:args.cpu_sample].cpu(), gt_depth[:args.cpu_sample].cpu(), K, n, H, W,
                           sigma, gamma, z_far, params, kinds, cam, gt_points, gt_sil, gt_depth, vpn_amd)

    if rank == 0:
        out = {
            'metric': 'render+Chamfer fwd+bwd images/sec', 'value': round(value, 1), 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'launch': 'hip-graph replay' if use_graph else 'eager',
            'config': {'workload': 'C3: B=%d/GPU, K=%d sphere primitives, %dx%d silhouette+depth, n=%d pts/prim '
                                   '(N=%d) vs M=%d GT points, Chamfer+L1 fwd+bwd' % (B, K, H, W, n, N, M),
                       'global_batch': B * world, 'parallelism': 'dp%d' % world,
                       'collective': 'rccl all-gather %d B/rank/step' % ((B * K * 10 + 4) * 4) if multi else 'none'},
            'roofline': roofline, 'kernel_us': kernel_us, 'entry_us': entry_us,
        }
        if cpu is not None:
            out['cpu_baseline'] = cpu
        if world == 1:      # row f1, outside the metric: the auction EMD loss the reference adds to the same step
            out['emd'] = emd_extra(B, M, dev, vpn_amd, cpu is not None)
        print(json.dumps(out), flush=True)
    if multi:
        dist.destroy_process_group()


def emd_extra(B, n, dev, vpn_amd, with_cpu):
    """EarthMoverDistanceLoss fwd+bwd as train.py:193 calls it (eps=0.005, 50 iterations) on B clouds of n points;
    not part of `value`.  CPU leg: the oracle's auction on one cloud."""
    gen = torch.Generator().manual_seed(3)
    x1 = torch.rand(B, n, 3, generator=gen).to(dev).requires_grad_(True)
    x2 = torch.rand(B, n, 3, generator=gen).to(dev)
    emd = vpn_amd.modules.loss.EarthMoverDistanceLoss()

    def once():
        d, _ = emd(x1, x2, 0.005, 50)
        x1.grad = None
        d.mean().backward()
    once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        once()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    out = {'workload': 'EMD auction fwd+bwd, B=%d, n=m=%d, eps=0.005, iters=50' % (B, n), 'ms': round(ms, 3),
           'clouds_per_s': round(B / ms * 1e3, 1)}
    if with_cpu:
        from oracle import vpn_oracle as O
        a, b = x1.detach()[:1].cpu(), x2[:1].cpu()
        t0 = time.perf_counter()
        rd, ra = O.emd_auction(a, b, 0.005, 50)
        cpu_s = time.perf_counter() - t0
        d, idx = emd(x1.detach()[:1], x2[:1], 0.005, 50)
        out['cpu_port_clouds_per_s'] = round(1.0 / cpu_s, 2)
        out['parity_vs_oracle'] = {'assignment_equal': bool(torch.equal(idx.cpu(), ra)),
                                   'dist_bit_equal': bool(torch.equal(d.cpu(), rd))}
    return out


def cpu_baseline(params, gt_points, gt_sil, gt_depth, K, n, H, W, sigma, gamma, z_far,
                 gpu_params, kinds, cam, gpu_gt_points, gpu_gt_sil, gpu_gt_depth, vpn_amd):
    """The oracle (CPU PyTorch restatement of the reference path, dense B*N*M Chamfer) timed on
    the host cores on a bounded sample of the same workload, and compared with the HIP path on
    exactly those samples."""
    from oracle import vpn_oracle as O
    cores = min(16, os.cpu_count() or 1)      # the 1-GPU box's CPU share is 16 cores
    torch.set_num_threads(cores)
    S = params.shape[0]
    u = O.philox_uniforms(1234, 0, S, K, n)
    kl = [0] * K
    camc = torch.tensor([[1.0, 0.0, 0.0]]).expand(S, 3).contiguous()

    def run(S=S):
        p = params.clone().requires_grad_(True)
        total = 0.0
        for b in range(S):                               # one image per chunk: dense tensors stay < 2 GB
            pb = p[b:b + 1]
            pts = O.sample_primitives(pb, kl, u[b:b + 1])
            cd = O.chamfer_loss(pts, gt_points[b:b + 1], each_batch=True).sum() / S
            a, d = O.raster(pb, kl, camc[b:b + 1], H, W, sigma, gamma, z_far)
            loss = cd + (a - gt_sil[b:b + 1]).abs().sum() / (S * H * W) + (d - gt_depth[b:b + 1]).abs().sum() / (S * H * W)
            loss.backward()
            total += float(loss.detach())
        return total, p.grad

    run(1)                                               # warm-up on one image
    t0 = time.perf_counter()
    loss_c, grad_c = run()
    dt = time.perf_counter() - t0
    # same samples on the GPU (explicit uniforms = the Philox draws the kernel makes itself)
    pg = gpu_params[:S].detach().clone().requires_grad_(True)
    loss_g = vpn_amd.HotPathLossFunction.apply(pg, kinds, cam[:S].contiguous(), gpu_gt_points[:S].contiguous(),
                                               gpu_gt_sil[:S].contiguous(), gpu_gt_depth[:S].contiguous(), n, 1234, 0,
                                               H, W, sigma, gamma, z_far, 1.0, 1.0, 1.0)[2]
    loss_g.backward()
    gerr = float((pg.grad.cpu() - grad_c).abs().max() / grad_c.abs().max())
    lerr = abs(float(loss_g.detach()) - loss_c) / abs(loss_c)
    return {'value': round(S / dt, 3), 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': '%d images of the same workload (1-image warm-up + 1 timed pass of %.1f s), torch CPU fp32, '
                      'dense B*N*M Chamfer as chamfer_distance.py:14-23' % (S, dt),
            'parity_vs_gpu': {'loss_rel': float('%.3g' % lerr), 'grad_rel': float('%.3g' % gerr)}}


if __name__ == '__main__':
    main()
