"""Benchmark of the e-OSVOS fine-tuning hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
(for N > 1 the driver launches it under torch.distributed.run, one rank per GPU).

Workload = BASELINE.json configs[1] "e-OSVOS-50 DAVIS-2017 val, 50 fine-tune iters,
batch=3": one *step* is one fine-tune iteration (DeepLabV3+-ResNet50 forward, fused BCE
loss+grad, backward, fused per-neuron-lr SGD update) on a batch of 3 synthetic 480x854
frames that are already resident in HBM.  Each rank fine-tunes its own (sequence, object)
replica -- evaluation shards by (sequence, object) with no collective -- so
scaling is weak and `value` = fine-tune iterations/s summed over ranks.
`extra.meta_tasks_per_sec` reports the meta-train metric (K=5 inner steps + meta frame,
B=1, RCCL all-reduce of the 161 MB meta-gradient + fused RAdam at N > 1);
`--metric meta` makes that the headline of the JSON line instead (BASELINE configs[3..4]:
one task per rank per meta-iteration, a step = one meta-iteration).

Prints ONE JSON line on rank 0.
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

H, W, BATCH = 480, 854, 3
FLOPS_PER_FRAME_ITER = 647.8e9      # fwd + dgrad + wgrad, no stem dgrad
FP32_MATRIX_PEAK = 157.3            # TFLOP/s, MI355X_MICROARCH.md "Peak FP32 (matrix)"
HOT_KERNEL_TRAFFIC_BYTES = (2 * 126.9e3 + 220.4e3) * 1024   # PMC, see profiles/r01_pmc_hot_kernel.txt


def cpu_baseline(sd, lrs, x, y, seconds_budget=25.0):
    """The CPU oracle (oracle/meta.py, torch fp32, all host cores) timed on a bounded sample
    of the same workload: whole B=3 fine-tune iterations until ~seconds_budget is used."""
    from oracle import meta
    # oneDNN stops scaling (and thrashes) far below the 256 hardware threads of the GPU
    # box's host; 32 threads is what the timed sample actually uses.
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    P = sd
    meta.finetune_step(P, lrs, x[:1], y[:1])       # warm-up (oneDNN primitives, page-in)
    # bounded sample: single-frame (batch 1) iterations of the same network/size; one
    # batch-3 iteration costs 3 of them, so the batch-3 rate is frames/s / 3
    n, t0 = 0, time.time()
    while True:
        _, _, P = meta.finetune_step(P, lrs, x[:1], y[:1])
        n += 1
        dt = time.time() - t0
        if dt > seconds_budget or n >= 12:
            break
    return {'value': n / dt / x.shape[0], 'unit': 'finetune_iters/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} single-frame fine-tune iterations ({H}x{W}) of the torch-CPU oracle (oracle/meta.py) '
                      f'on {cores} threads in {dt:.1f}s; value = frames/s / {x.shape[0]} (batch-{x.shape[0]} '
                      f'iterations/s)'}


def bench_meta(a, eng, dist, rank, world, sd, lrs, xg, yg, barrier, dev):
    """meta-tasks/s: every rank runs one task (5 inner steps at batch 1 + the meta frame) per meta-iteration, then
    ONE all-reduce of the 161 MB meta-gradient and the fused RAdam step + lr clamp on every rank."""
    from eosvos_amd.meta_run import MetaTrainer
    mt = MetaTrainer(eng, dist=dist, meta_batch_size=world)
    mt.load_state(sd, lrs)
    x1, y1 = xg[:1].contiguous(), yg[:1].contiguous()
    xm, ym = torch.flip(x1, dims=[3]).contiguous(), torch.flip(y1, dims=[3]).contiguous()
    for _ in range(a.warmup):
        mt.meta_iteration([(x1, y1, xm, ym)], inner_steps=5)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = mt.meta_iteration([(x1, y1, xm, ym)], inner_steps=5)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    k_ms, k_flops = eng.time_hot_kernel(1, reps=20)
    achieved = k_flops / (k_ms * 1e-3) / 1e12
    if rank == 0:
        out = {
            'metric': 'meta_tasks_per_sec', 'value': world * a.steps / dt, 'unit': 'meta_tasks/s', 'n_gpus': world,
            'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * dt / a.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'meta-train iteration (BASELINE configs[3..4]): meta_batch_size = number of ranks, one '
                                   f'task per rank = 5 inner fine-tune steps + 1 meta frame at batch 1, {H}x{W}, BCE; one '
                                   'all-reduce(sum) of the 40.3 M-float meta-gradient, RAdam + lr clamp on every rank',
                       'meta_batch_size': world, 'inner_steps': 5, 'height': H, 'width': W,
                       'parallelism': f'tasks sharded x{world}'},
            'roofline': {'bound': 'mfma', 'kernel': 'conv_igemm_kernel<128,false,2>: Winograd-domain batched GEMM of decoder.last_conv.0 forward, batch 1 '
                                   '(own FLOPs)',
                         'achieved': achieved, 'peak': FP32_MATRIX_PEAK, 'unit': 'TFLOP/s',
                         'frac': achieved / FP32_MATRIX_PEAK, 'traffic': None, 'kernel_ms': k_ms,
                         'flops_per_launch': k_flops,
                         'whole_step_tflops': 6 * FLOPS_PER_FRAME_ITER * a.steps / dt / 1e12},   # 9-tap-equivalent FLOPs
            'cpu_baseline': None, 'extra': {'last_meta_loss': losses[-1]},
        }
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-meta', action='store_true')
    ap.add_argument('--metric', choices=['finetune', 'meta'], default='finetune',
                    help="'meta': the JSON line reports meta-tasks/s (BASELINE configs[3..4]: one task per rank per "
                         "meta-iteration, all-reduce + RAdam included); a step is then one meta-iteration")
    a = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device(f'cuda:{local_rank}'))
    dev = f'cuda:{local_rank}'

    from eosvos_amd import synthetic
    from eosvos_amd.engine import Engine
    from eosvos_amd.meta_run import MetaTrainer

    sd = synthetic.synthetic_state('resnet50')
    lrs = synthetic.synthetic_lrs('resnet50')
    x, y = synthetic.synthetic_frames(BATCH, H, W, seed=7 + rank)
    eng = Engine('resnet50', H, W, max_batch=BATCH, device=dev)
    eng.load_model_state(sd, lrs)
    xg, yg = x.to(dev), y.to(dev)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if a.metric == 'meta':
        return bench_meta(a, eng, dist, rank, world, sd, lrs, xg, yg, barrier, dev)

    for _ in range(a.warmup):
        eng.finetune_step(xg, yg, sync_loss=False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.finetune_step(xg, yg, sync_loss=False)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    last_loss = eng.finetune_step(xg, yg)           # sanity: still finite after K steps
    value = world * a.steps / dt

    # dominant kernel, timed live with HIP events on the engine's stream
    k_ms, k_flops = eng.time_hot_kernel(BATCH, reps=20)
    achieved = k_flops / (k_ms * 1e-3) / 1e12
    roofline = {'bound': 'mfma', 'kernel': 'conv_igemm_kernel<128,false,2>: batched GEMM of decoder.last_conv.0 forward in the Winograd '
                'F(4x4,3x3) domain, 36 x [4860 tiles x 304] x [304 x 256] (+ its fix-up launch), the heaviest layer of the '
                'network; achieved counts this GEMM\'s own FLOPs (4x fewer than the 9-tap form of the layer)', 'achieved': achieved, 'peak': FP32_MATRIX_PEAK, 'unit': 'TFLOP/s',
                'frac': achieved / FP32_MATRIX_PEAK,
                # bytes per launch from the rocprofv3 --pmc passes of this kernel (separate runs,
                # profiles/r01_pmc_hot_kernel.txt): 2*FETCH_SIZE (gfx950 half-count correction) +
                # WRITE_SIZE; algorithmic bytes are 213 MB V + 11 MB U in, 179 MB M out
                'traffic': HOT_KERNEL_TRAFFIC_BYTES if BATCH == 3 else None, 'kernel_ms': k_ms,
                'flops_per_launch': k_flops,
                'whole_step_tflops': BATCH * FLOPS_PER_FRAME_ITER * a.steps / dt / 1e12}

    extra = {'last_loss': last_loss, 'mfma_probe_tflops': eng.mfma_probe(),
             'conv_algorithms': 'fp32 throughout; 6 of the 63 convs (decoder 3x3 x2, layer4 conv2 x3: Winograd F(4x4,3x3); '
                                'ASPP d=6: F(2x2,3x3)) run forward, data and weight gradient in the Winograd domain, the rest '
                                'as implicit GEMM; whole_step_tflops counts 9-tap-equivalent FLOPs; full-size parity margins '
                                'in profiles/r01_parity_margins_full_size.txt'}
    if not a.no_meta:
        # meta-train metric: tasks/s with K=5 inner steps + meta frame at B=1 (configs[3..4])
        mt = MetaTrainer(eng, dist=dist, meta_batch_size=world)
        mt.load_state(sd, lrs)
        xm, ym = torch.flip(xg[:1], dims=[3]).contiguous(), torch.flip(yg[:1], dims=[3]).contiguous()
        mt.meta_iteration([(xg[:1].contiguous(), yg[:1].contiguous(), xm, ym)], inner_steps=5)   # warm-up
        barrier()
        t1 = time.perf_counter()
        n_it = 2
        for _ in range(n_it):
            mt.meta_iteration([(xg[:1].contiguous(), yg[:1].contiguous(), xm, ym)], inner_steps=5)
        barrier()
        dtm = time.perf_counter() - t1
        if dist is not None:
            t = torch.tensor([dtm], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtm = float(t.item())
        extra['meta_tasks_per_sec'] = world * n_it / dtm
        extra['meta_config'] = f'meta_batch_size={world}, 5 inner steps + 1 meta frame, batch 1, {H}x{W}'

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(sd, lrs, x, y)
    if rank == 0:
        out = {
            'metric': 'finetune_iters_per_sec', 'value': value, 'unit': 'finetune_iters/s', 'n_gpus': world,
            'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * dt / a.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'e-OSVOS-50 fine-tune iteration (BASELINE configs[1]): DeepLabV3+-ResNet50, '
                                   f'batch {BATCH}, {H}x{W}, BCE, per-neuron-lr SGD; one (sequence, object) per rank',
                       'batch': BATCH, 'height': H, 'width': W, 'parallelism': f'replicas x{world}'},
            'roofline': roofline, 'cpu_baseline': cpu, 'extra': extra,
        }
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
