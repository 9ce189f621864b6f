"""Run the one-rank child of tests/test_dist_gpu.py with extra environment (KEY=VALUE arguments) and show its output."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_dist_gpu as T
env = dict(os.environ, RSCOTR_DIST_SINGLE='1', **dict(a.split('=', 1) for a in sys.argv[1:]))
r = subprocess.run([sys.executable, '-c', T._CHILD, ROOT, '29650'], capture_output=True, text=True, timeout=900, env=env)
print(r.stdout[-1500:])
print('\n'.join(r.stderr.splitlines()[-25:]))
