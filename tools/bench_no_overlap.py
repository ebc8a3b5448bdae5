"""GPU box: bench.py with the weight gradients on the main stream (no second stream): A/B for the two-stream schedule."""
import os, sys, runpy
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops
ops.OVERLAP_WGRAD = False
sys.argv = ['bench.py', '--no-cpu-baseline', '--eager'] + sys.argv[1:]
runpy.run_path(os.path.join(REPO, 'bench.py'), run_name='__main__')
