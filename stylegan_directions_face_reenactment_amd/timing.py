"""Per-launch timing of the path, for bench.py and the timing scripts: HIP events around every MFMA conv launch and every
HBM-bound launch (blur, ToRGB, ToRGB finish), recorded on the stream the launch goes to.

    with timing.collect() as t:
        G([w], input_is_latent=True, verify_range=False)
    torch.cuda.synchronize()
    t.conv   # [(start_event, end_event, algorithmic_flops, description), ...] in launch order
    t.hbm    # [(start_event, end_event, algorithmic_bytes, description), ...]

The collector is a context object held per host thread (like functional.Config), not a module global: two threads, or a
timed and an untimed generator in one process, do not see each other.  hipGraph capture / replay is off while one is active
(graph_runner checks `timing.active()`): events cannot be recorded inside a replayed graph.
"""
import threading

import torch

_ambient = threading.local()


class LaunchTimer:
    __slots__ = ('conv', 'hbm')

    def __init__(self):
        self.conv, self.hbm = [], []

    def _record(self, where, amount, desc, launch):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = launch()
        e1.record()
        where.append((e0, e1, amount, desc))
        return r


def active():
    """The LaunchTimer collecting on this thread, or None."""
    return getattr(_ambient, 'timer', None)


class collect:
    """`with timing.collect() as t:` -- every conv / HBM-bound launch this thread issues inside the block is bracketed by events."""

    def __init__(self, timer=None):
        self.timer, self.prev = timer or LaunchTimer(), None

    def __enter__(self):
        self.prev = active()
        _ambient.timer = self.timer
        return self.timer

    def __exit__(self, *exc):
        _ambient.timer = self.prev


def timed_conv(desc, flops, launch):
    """Run `launch()`; under a collector, bracketed by events with its ALGORITHMIC flops."""
    t = active()
    return launch() if t is None else t._record(t.conv, flops, desc, launch)


def timed_hbm(desc, nbytes, launch):
    """Run `launch()`; under a collector, bracketed by events with its ALGORITHMIC bytes (inputs read once, outputs written once)."""
    t = active()
    return launch() if t is None else t._record(t.hbm, nbytes, desc, launch)
