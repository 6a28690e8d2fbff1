import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import test_stress_gpu as t
bad = 0
for seed in range(4, 28):
    try:
        t.test_random_operation_sequences(None, seed)
        print("seed", seed, "ok", flush=True)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED", e, flush=True)
print("failures:", bad)
