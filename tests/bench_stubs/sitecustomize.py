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

    _real_barrier = dist.barrier

    def _chatter():
        # what RCCL does on a GPU box: a line through C stdio, which is block-buffered on a pipe and would surface at
        # process exit — after the JSON line — if bench.py did not flush / re-point stdout itself
        import ctypes
        ctypes.CDLL(None).printf(b"Librccl path : stand-in (rank %s)\n" % os.environ.get("RANK", "?").encode())

    def _init(backend=None, **kw):
        return _real_init("gloo", **{k: v for k, v in kw.items() if k != "device_id"})

    def _barrier(*a, **kw):
        _chatter()   # (after gloo's own connection message, whose std::endl would flush an earlier line with it)
        return _real_barrier(*a, **kw)
    dist.init_process_group = _init
    dist.barrier = _barrier
