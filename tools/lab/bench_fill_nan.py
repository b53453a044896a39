"""Lab: the driver command (short) with every torch.empty starting as NaN -- the losses, the packed section and the batch sweep must stay finite."""
import sys, runpy, torch
torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
sys.argv = ['bench.py', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--traffic', 'committed']
runpy.run_path('bench.py', run_name='__main__')
