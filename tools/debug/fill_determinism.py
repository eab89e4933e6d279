"""Does any result depend on memory nothing wrote?  One process per allocation fill (EOSVOS_DEBUG_FILL unset / a huge finite word /
NaN): forward logits, every named gradient tensor, the weight gradients and the parameters after 3 steps, per matrix mode, saved and
compared BITWISE across the fills.

    python tools/debug/fill_determinism.py [H W B]         (parent: spawns the children and compares)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
FILLS = {'zero_pages': None, 'huge': '7f000000', 'nan': '7fc00000', 'one': '3f800000'}


def child(out, H, W, B, norm):
    import torch
    from eosvos_amd import synthetic
    from eosvos_amd.engine import Engine
    sd, lrs = synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50')
    x, y = synthetic.synthetic_frames(B, H, W, seed=21)
    res = {}
    for mode in ('f16x3', 'bf16x6', 'f32'):
        MB = int(os.environ.get('FILL_MAX_BATCH', B))
        e = Engine('resnet50', H, W, max_batch=MB, device='cuda:0', norm=norm)
        e.load_model_state(sd, lrs)
        e._verify_pending = False
        e.set_engine_matrix_mode(mode)
        xg, yg = x.cuda(), y.cuda()
        d = {}
        if MB > B:                       # a bigger batch first (inference), as the evaluation loop does: stale rows beyond B afterwards
            xb = synthetic.synthetic_frames(MB, H, W, seed=33)[0].cuda()
            d['infer_big'] = e.infer(xb).cpu()
            e.snapshot()
            e.finetune_step(xg, yg)
            e.restore()
        d['logits'] = e.forward(xg).cpu()
        e.keep_grads(True)
        d['loss'] = e.finetune_step(xg, yg)
        d['grads'] = e.get_grads().cpu()
        for n in ('g_dcat', 'g_cat', 'g_p1', 'g_c1', 'g_d1', 'g_d2', 'g_proj') + tuple(f'blk{i}.g_out' for i in range(16)) + tuple(f'blk{i}.g_t1' for i in range(16)) + tuple(f'blk{i}.g_t2' for i in range(16)):
            try:
                d[n] = e.debug_tensor(n).cpu()[:B]
            except Exception:      # noqa: BLE001
                pass
        for _ in range(2):
            e.finetune_step(xg, yg)
        d['params3'] = e.get_params().cpu()
        res[mode] = d
        e.close()
    torch.save(res, out)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6])
        sys.exit(0)
    import torch
    H, W, B = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (480, 854, 3)
    norm = sys.argv[4] if len(sys.argv) > 4 else 'bn'
    outs = {}
    runs = dict(FILLS)
    # AB_LIBS="before=e-osvos_amd/variants/libeosvos_before.so": instead of the fills, the default library ('zero_pages') against
    # other builds -- is a kernel change bit-identical?
    libs = {}
    if os.environ.get('AB_LIBS'):
        runs = {'zero_pages': None}
        for item in os.environ['AB_LIBS'].split(','):
            k, v = item.split('=')
            runs[k] = None
            libs[k] = os.path.abspath(v)
    for name, fill in runs.items():
        env = dict(os.environ, EOSVOS_MODE_GUARD='0')
        env.pop('EOSVOS_DEBUG_FILL', None)
        env.pop('EOSVOS_LIB', None)
        if name in libs:
            env['EOSVOS_LIB'] = libs[name]
        if fill:
            env['EOSVOS_DEBUG_FILL'] = fill
        path = f'/tmp/fill_{name}.pt'
        subprocess.run([sys.executable, os.path.abspath(__file__), '--child', path, str(H), str(W), str(B), norm], env=env, check=True)
        outs[name] = torch.load(path, weights_only=False)
    ref = outs['zero_pages']
    bad = 0
    for name, o in outs.items():
        if name == 'zero_pages':
            continue
        for mode in ref:
            for k, v in ref[mode].items():
                w = o[mode][k]
                same = (v == w) if isinstance(v, float) else bool(torch.equal(v, w))
                if not same:
                    bad += 1
                    if isinstance(v, float):
                        print(f'{name:6s} {mode:7s} {k:12s} differs: {v!r} vs {w!r}')
                    else:
                        dd = (v - w).abs()
                        print(f'{name:6s} {mode:7s} {k:12s} differs: {int((dd > 0).sum())} elements, max |diff| {float(dd.max()):.3e} (scale {float(v.abs().max()):.3e})')
    print(f'{H}x{W} batch {B} norm {norm}: {bad} (fill, mode, tensor) combinations differ from the zero-page run')
