"""A rank of a multi-rank bench launch in which ONE rank dies: rank 1 exits with an error before the collective, the others
wait in a gloo barrier (they would hang for ever without torch.distributed.run's tear-down)."""
import os
import sys

import torch.distributed as dist

rank = int(os.environ['RANK'])
if rank == 1:
    sys.stderr.write('rank 1: simulated failure\n')
    sys.exit(3)
dist.init_process_group('gloo')
dist.barrier()
print('{"metric": "should never be printed"}')
