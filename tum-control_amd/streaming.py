"""
streaming.py -- a ring of capsules for callers that stream independent batches through the solver.

One capsule = one batch of OCP instances with its own iterate, inputs and workspaces, and one HIP stream. A batch of a few
thousand instances is only a few rounds of resident wavefronts (1024 interior point wavefronts at a time on an MI355X), so a
capsule that runs alone leaves a good part of the chip idle while the last instances of a batch finish -- and while the short
kernels around the interior point method run. A caller whose batches do not depend on each other (scenario fan-outs, parameter
sweeps, Monte-Carlo runs: every batch configuration of BASELINE.json) can keep several batches in flight: `SolverRing` owns S
capsules of the same OCP and hands them out in turn; everything enqueued on a capsule (device-to-device upload of x0 / yref,
cold start, solve, result packing) runs on that capsule's stream, so consecutive batches overlap on the GPU while every batch
still gets a complete, independent solve. Measured on config 2 (4096 x N = 40, fresh batch every step): 2.49 M solves/s on one
capsule, 2.91 / 3.04 / 3.1 M on two / three / four (scripts/dev/overlap_probe.py).

This is host logic above the C-ABI (which already allows any number of capsules per process); no new entry point is needed.
A closed loop -- where solve k + 1 needs the result of solve k -- has nothing to overlap and keeps using one capsule.
"""


class SolverRing:
    def __init__(self, n_slots, factory, streams=None):
        """factory(slot) -> a configured BatchedOcpSolver (all slots must describe the same OCP); streams: optional list of
        raw hipStream_t handles (ints), one per slot (default: every capsule keeps the non-blocking stream it created)."""
        if n_slots < 1:
            raise ValueError("n_slots < 1")
        self.solvers = [factory(i) for i in range(n_slots)]
        if streams is not None:
            if len(streams) != n_slots:
                raise ValueError("one stream per slot")
            for s, st in zip(self.solvers, streams):
                s.set_stream(st)
        self._next = 0

    def __len__(self):
        return len(self.solvers)

    def __getitem__(self, i):
        return self.solvers[i]

    def __iter__(self):
        return iter(self.solvers)

    def acquire(self):
        """(slot index, solver) of the capsule whose turn it is. Work enqueued on it is ordered behind that capsule's previous
        batch (same stream), and runs beside the other capsules' batches."""
        i = self._next
        self._next = (i + 1) % len(self.solvers)
        return i, self.solvers[i]

    def synchronize(self):
        for s in self.solvers:
            s.synchronize()
