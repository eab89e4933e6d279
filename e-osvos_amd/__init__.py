"""e-osvos_amd: MI355X-native e-OSVOS inner-loop engine (host side).

Hot path of dvl-tum/e-osvos -- per-video one-shot fine-tuning / online adaptation of
DeepLabV3+-ResNet with the meta-learned per-neuron-lr SGD step and BCE loss, plus the
meta-train outer step -- executed by hand-written gfx950 HIP kernels in
`csrc/` behind the C-ABI of `include/eosvos.h`.  The Python modules here mirror the
reference's operator interface for that path (`networks.DeepLabV3Plus`,
`meta_optim.MetaOptimizer`, `helper_func.compute_loss/init_parent_model`, `radam.RAdam`).
There is no CPU fallback: without the built library every op raises.
"""
__version__ = '0.1.0'
