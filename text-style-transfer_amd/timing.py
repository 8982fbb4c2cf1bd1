"""Measurement aid: HIP-event timing of single kernel launches inside a running decode loop.

The events are created AND recorded once before any measurement: ``torch.cuda.Event`` makes its HIP event at the first
``record()``, and creating hundreds of events (each takes a completion signal from the runtime) inside a timed region was
seen to stall the launching thread for tens of milliseconds at a time, which shows up both in the wall time of the
batch and, when the stall falls between a launch and the record after it, in the measured launch duration.
"""
import torch


class LaunchTimer:
    """A fixed pool of (start, end) event pairs.  ``begin()`` / ``end(work)`` bracket one launch on the current
    stream; launches beyond ``capacity`` are counted but not timed.  ``totals()`` synchronises and returns
    (milliseconds, work, launches timed)."""

    def __init__(self, capacity):
        self.pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(capacity)]
        for e0, e1 in self.pairs:
            e0.record()
            e1.record()
        torch.cuda.synchronize()
        self.work = [0.0] * capacity
        self.n = 0
        self.dropped = 0

    def reset(self):
        self.n = 0
        self.dropped = 0

    def begin(self):
        if self.n < len(self.pairs):
            self.pairs[self.n][0].record()

    def end(self, work):
        if self.n < len(self.pairs):
            self.pairs[self.n][1].record()
            self.work[self.n] = work
            self.n += 1
        else:
            self.dropped += 1

    def durations_ms(self):
        torch.cuda.synchronize()
        return [e0.elapsed_time(e1) for e0, e1 in self.pairs[:self.n]]

    def totals(self):
        ms = self.durations_ms()
        return sum(ms), sum(self.work[:self.n]), self.n
