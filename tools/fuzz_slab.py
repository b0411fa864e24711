"""Fuzz of the slab decomposition (TEST INFRASTRUCTURE): random long boxes, every liquid particle with its own velocity (ownership
changes hands all the time), 2-4 HIP ranks sharing the one card over gloo, overlapped or plain exchange, 1-3 predict-correct
iterations; the union of the owned sets must equal the single-domain oracle run bit for bit.
  tools/fuzz_slab.py [first_seed=1] [count=20]"""
import os, sys, tempfile, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import numpy as np
import test_slab as T
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
failed = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(900 + seed)
    world = int(rng.integers(2, 5))
    overlap = str(int(rng.integers(0, 2)))
    maxiter = int(rng.integers(1, 4))
    steps = int(rng.integers(5, 11))
    env = {"SPHMI_TEST_FUZZ_SEED": str(seed), "SPHMI_SLAB_OVERLAP": overlap, "SPHMI_TEST_MAXITER": str(maxiter)}
    with tempfile.TemporaryDirectory() as out:
        try:
            results = T.run_ranks("hip", world, out, steps=steps, env=env)
            sc, pos_ref, vel_ref = T.single_domain_reference(steps=steps, max_iteration=maxiter, env=env)
            T.check_union(results, sc, pos_ref, vel_ref)
            adopted = sum(int(r["adopted"]) for r in results)
            print("seed %3d: %d ranks, overlap %s, %d iterations, %d steps, %d particles, cuts %s, %d changed owner: ok"
                  % (seed, world, overlap, maxiter, steps, sc["cfg"].particleCount, [int(c) for c in results[0]["cuts"]], adopted), flush=True)
        except AssertionError as e:
            failed += 1
            print("seed %3d: %d ranks, overlap %s, %d iterations, %d steps: FAILED %s" % (seed, world, overlap, maxiter, steps, str(e)[-600:]), flush=True)
print("%d runs, %d failed, %.0f s" % (count, failed, time.time() - t0))
sys.exit(1 if failed else 0)
