"""One rank of `bench.py --metric meta` / the fine-tune metric on CPU (gloo) with the stand-in engine:
usage bench_worker.py OUT_PREFIX metric"""
import contextlib
import io
import os
import sys

import common
import bench

out, metric = sys.argv[1], sys.argv[2]
bench.H, bench.W = common.H, common.W
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main(['--gpus', os.environ['WORLD_SIZE'], '--steps', '2', '--warmup', '1', '--metric', metric, '--no-cpu-baseline',
                '--no-ab'], engine_factory=common.FakeEngine, device='cpu', backend='gloo')
open(f'{out}.{os.environ["RANK"]}', 'w').write(buf.getvalue())
