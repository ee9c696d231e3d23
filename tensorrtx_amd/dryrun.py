"""bench.py --dry-run: stand-ins for the GPU-side objects so that bench.py's CONTROL FLOW - rank / world handling, the image partition,
rank 0's build + plan broadcast, warm-up / K-step legs / barriers / max-over-ranks, the host-fed ring, the assembly of the JSON line -
runs on CPU under gloo with world_size > 1 (tests/test_bench_dry_run.py).  The 8-GPU run is the driver's alone to launch; this is how
its code path is exercised beforehand.  Nothing here computes anything: a dry-run line is marked {"dry_run": true} and its numbers are
meaningless by construction.  Never imported unless --dry-run is given."""
import contextlib
import time


class Event:
    def __init__(self, enable_timing=False):
        self.t = None

    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return ((other.t or 0.0) - (self.t or 0.0)) * 1e3

    def synchronize(self):
        pass


class Stream:
    cuda_stream = 0

    def __init__(self, priority=0):
        self.priority = priority

    def wait_event(self, ev):
        pass

    def synchronize(self):
        pass


class FakeCuda:
    """the slice of torch.cuda that bench.py touches"""
    Event = Event
    Stream = Stream

    def __init__(self, n_devices):
        self._n = n_devices
        self._cur = 0

    def is_available(self):
        return True

    def device_count(self):
        return self._n

    def set_device(self, i):
        self._cur = i

    def current_device(self):
        return self._cur

    def synchronize(self):
        pass

    def current_stream(self):
        return Stream()

    @contextlib.contextmanager
    def stream(self, s):
        yield


class DryEngine:
    """engine.Engine's interface over a plan that was really built and lowered (CPU work), executing nothing."""

    def __init__(self, plan, describe, step_seconds=2e-4):
        d = describe(plan)
        self._low = describe(plan, lowered=True)
        ins = {t["id"] for t in d["tensors"] if t["is_input"]}
        outs = [t for t in d["tensors"] if t["is_input"] or t["is_output"]]
        self.names = [t["name"] for t in outs]
        self.dims = [tuple(t["dims"]) for t in outs]
        self.is_input = [t["id"] in ins for t in outs]
        self.nb_bindings = len(outs)
        self._dt = step_seconds

    def enqueue(self, batch, bindings, stream=None):
        time.sleep(self._dt)

    def create_context(self):
        return self

    def profile(self, batch, bindings):
        return [{"name": o.get("name", ""), "kind": o["kind"], "ms": 0.01, "kernel_ms": 0.009} for o in self._low["ops"]]

    def tactics(self):
        return []

    def close(self):
        pass


class DeviceEngine(DryEngine):
    """DryEngine with the [input, output] binding order bench.py's in-process path uses"""


class Replicas:
    """replicas.DeviceReplicas without devices: one stand-in engine per index"""

    def __init__(self, devices, make_engine):
        self.devices = list(devices)
        self.engines = [make_engine(d) for d in self.devices]
        self.streams = [Stream() for _ in self.devices]

    def close(self):
        self.engines = []
