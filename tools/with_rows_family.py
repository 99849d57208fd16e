"""Run a script with the rows kernel family set first (0 = k_linear_rows only, 1 = k_rows_frag where it applies):
python tools/with_rows_family.py 0 bench.py --workload layout ...   -- same-box A/B of the two families."""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoscene_amd import hip
hip.check(hip.lib().es_rows_set_kernel_family(int(sys.argv[1])), 'es_rows_set_kernel_family')
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
