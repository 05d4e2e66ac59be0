import time, numpy as np, sys
sys.path.insert(0, '.')
import embeddinghub_amd as ehx
d = 1536
m = 131072
rows = np.random.default_rng(1).standard_normal((m, d)).astype(np.float32)
keys = ["s%d" % i for i in range(m)]
for dt, name in ((ehx.DTYPE_F32, "f32"), (ehx.DTYPE_F16, "f16")):
    w = ehx.Space("probe-" + name, d, metric=ehx.METRIC_COSINE, initial_capacity=2 * m, dtype=dt)
    t0 = time.perf_counter(); w.set_batch(keys[:65536], rows[:65536]); t1 = time.perf_counter()
    w.set_batch(keys[65536:], rows[65536:]); t2 = time.perf_counter()
    print(name, "first %.3f s, second %.3f s -> %.0f rows/s" % (t1 - t0, t2 - t1, 65536 / (t2 - t1)))
    w.drop()
