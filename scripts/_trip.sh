# scratch trip script of the session (what `gpurun -- 'bash scripts/_trip.sh'` runs): the full GPU suite + the default bench
bash scripts/gpu_suite.sh suite
