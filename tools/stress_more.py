import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import test_stress_gpu as t
bad = 0
import os
lo, hi = int(os.environ.get("STRESS_LO", 4)), int(os.environ.get("STRESS_HI", 28))
for seed in range(lo, hi):
    try:
        t.test_random_operation_sequences(None, seed)
        print("seed", seed, "ok", flush=True)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED", e, flush=True)
print("failures:", bad)
