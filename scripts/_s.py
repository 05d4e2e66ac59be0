import time, numpy as np, sys, threading
sys.path.insert(0, '.')
import embeddinghub_amd as ehx
d = 1536
m = 131072
rows = np.random.default_rng(1).standard_normal((m, d)).astype(np.float32)
keys = ["s%d" % i for i in range(m)]
for dt, name in ((ehx.DTYPE_F32, "f32"), (ehx.DTYPE_F16, "f16")):
    w = ehx.Space("probe-" + name, d, metric=ehx.METRIC_COSINE, initial_capacity=2 * m, dtype=dt)
    w.set_batch(keys[:8192], rows[:8192])
    t0 = time.perf_counter()
    for i0 in range(8192, m, 8192):
        w.set_batch(keys[i0:i0 + 8192], rows[i0:i0 + 8192])
    dt_ = time.perf_counter() - t0
    print(name, "chunks of 8192: %.0f rows/s (%.1f ms per chunk)" % ((m - 8192) / dt_, dt_ / ((m - 8192) / 8192) * 1e3), flush=True)
    # prepared chunks (what bench.py's writers call)
    w2 = ehx.Space("probe2-" + name, d, metric=ehx.METRIC_COSINE, initial_capacity=2 * m, dtype=dt)
    preps = [w2.prepare_batch(keys[i0:i0 + 8192], rows[i0:i0 + 8192]) for i0 in range(0, m, 8192)]
    w2.set_prepared(preps[0])
    t0 = time.perf_counter()
    for p in preps[1:]:
        w2.set_prepared(p)
    dt_ = time.perf_counter() - t0
    print(name, "prepared chunks: %.0f rows/s (%.1f ms per chunk)" % ((m - 8192) / dt_, dt_ / (len(preps) - 1) * 1e3), flush=True)
    w.drop(); w2.drop()
