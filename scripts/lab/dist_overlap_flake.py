"""Repeat the one-rank overlapped-exchange child of tests/test_dist_gpu.py and print the stderr of any run that dies."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_dist_gpu as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
inline = sys.argv[2] if len(sys.argv) > 2 else '0'
bad = 0
for i in range(n):
    env = dict(os.environ, RSCOTR_DIST_SINGLE='1', RSCOTR_DIST_INLINE=inline)
    r = subprocess.run([sys.executable, '-c', T._CHILD, ROOT, str(29600 + i)], capture_output=True, text=True, timeout=900, env=env)
    ok = any(l.startswith('RESULT ') for l in r.stdout.splitlines())
    print(f'run {i}: {"ok" if ok else "FAILED rc=%d" % r.returncode}', flush=True)
    if not ok:
        bad += 1
        print('\n'.join(l for l in r.stderr.splitlines() if 'rror' in l or 'HIP' in l)[:3000], flush=True)
print(f'{bad} of {n} failed')
