import torch, os, sys, subprocess
os.environ['GPU_MAX_HW_QUEUES'] = '24'      # after `import torch`, before the first GPU call
sys.argv = ['x', '2', '4', '0']
exec(open('tools/stream_queue_probe2.py').read())
