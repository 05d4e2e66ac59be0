"""Loaded by every python process that has tests/bench_stubs on PYTHONPATH.  With EHX_BENCH_STANDINS=1 it replaces the
engine and torch.cuda by the stand-ins of standins.py and turns the ranks' process group into gloo, so that
`python bench.py --gpus 2` — the launcher and the two ranks it starts — runs on a CPU-only box
(tests/test_bench_flow.py::test_gpus_2_launches_two_real_ranks)."""
import os

if os.environ.get("EHX_BENCH_STANDINS") == "1":
    import sys
    _here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(os.path.dirname(_here)))  # the repo root: `oracle`
    sys.path.insert(0, _here)
    import standins
    standins.install(real_sharded=True)
    import torch.distributed as dist
    _real_init = dist.init_process_group
    dist.init_process_group = lambda backend=None, **kw: _real_init("gloo", **{k: v for k, v in kw.items() if k != "device_id"})
