"""The validation child process of train_meta (tests): eosvos_amd.eval_worker.main with the stand-in model."""
import sys

import common
from eosvos_amd import eval_worker

if __name__ == '__main__':
    eval_worker.main(sys.argv[1:], init_parent_model=common.fake_init_parent_model)
