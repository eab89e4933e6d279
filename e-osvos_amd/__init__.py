"""e-osvos_amd: MI355X-native e-OSVOS inner-loop engine (host side).

Hot path of dvl-tum/e-osvos -- per-video one-shot fine-tuning / online adaptation of
DeepLabV3+-ResNet with the meta-learned per-neuron-lr SGD step and BCE loss, plus the
meta-train outer step -- executed by hand-written gfx950 HIP kernels in
`csrc/` behind the C-ABI of `include/eosvos.h`.  The Python modules here mirror the
reference's operator interface for that path (`networks.DeepLabV3Plus`,
`meta_optim.MetaOptimizer`, `helper_func.compute_loss/init_parent_model`, `radam.RAdam`).
There is no CPU fallback: without the built library every op raises.
"""
import os as _os_env

# ROCm deals HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4: the first streams get a queue each, later ones
# the least-referenced queue) and the queues onto 4 pipes.  Engines that run side by side -- meta tasks / objects in
# flight, one stream each -- lose a quarter of their rate when two of their streams share a queue or a pipe, and with 4
# queues which ones do depends on every stream the process created before (tools/stream_queue_probe*.py,
# profiles/r03_stream_queue_probe.txt, r03_hw_queue_sweep.txt: 41 vs 29-35 meta-tasks/s, 107 vs 85 iterations/s).  With
# more queues than streams every stream owns one, and engines built back to back WITHOUT a side stream sit on consecutive
# queues = different pipes.  A lone engine is unaffected (90.1 / 90.3 / 90.2 it/s with 4 / 8 / 16 queues).
# The HIP runtime reads this when it initialises (first GPU call), so it is set as early as this package can.
# It is a LAUNCHER setting: it changes the queue allocation of every GPU user of the process, and it has no effect when
# the host application initialised HIP before importing this package -- that case gets a warning instead of silence.
if 'GPU_MAX_HW_QUEUES' not in _os_env.environ:
    import sys as _sys
    _t = _sys.modules.get('torch')
    if _t is not None and getattr(_t, 'cuda', None) is not None and _t.cuda.is_initialized():
        import warnings as _w
        _w.warn('e-osvos_amd: the GPU runtime was initialised before this package was imported, so GPU_MAX_HW_QUEUES=24 cannot '
                'take effect; engines that run side by side (meta tasks / objects in flight) may share hardware queues and lose '
                'about a quarter of their rate.  Export GPU_MAX_HW_QUEUES=24 in the launcher.', RuntimeWarning)
        del _w
    else:
        _os_env.environ['GPU_MAX_HW_QUEUES'] = '24'
    del _sys, _t
del _os_env

__version__ = '0.7.0'
