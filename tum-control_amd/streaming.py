"""
streaming.py -- a ring of capsules for callers that stream independent batches through the solver.

One capsule = one batch of OCP instances with its own iterate, inputs and workspaces, and one HIP stream. A batch of a few
thousand instances is only a few rounds of resident wavefronts (1024 interior point wavefronts at a time on an MI355X), so a
capsule that runs alone leaves a good part of the chip idle while the last instances of a batch finish -- and while the short
kernels around the interior point method run. A caller whose batches do not depend on each other (scenario fan-outs, parameter
sweeps, Monte-Carlo runs: every batch configuration of BASELINE.json) can keep several batches in flight: `SolverRing` owns S
capsules of the same OCP and hands them out in turn; everything enqueued on a capsule (device-to-device upload of x0 / yref,
cold start, solve, result packing) runs on that capsule's stream, so consecutive batches overlap on the GPU while every batch
still gets a complete, independent solve. Measured on config 2 (4096 x N = 40, fresh batch every step; profiles/r04_streams.txt,
round 6: profiles/r06_stream_counts.txt, six fresh processes per count on two boxes): 3.4 M solves/s on one capsule, 4.05 / 4.26 / 4.35 M on
two / three / four, 4.15 / 4.23 / 4.22 M on five / six / eight -- FOUR is the optimum of a long stream of batches (the runtime multiplexes
streams onto four hardware queues; GPU_MAX_HW_QUEUES = 8 / 16 changes nothing; a run of a few batches per capsule is mostly ramp and drain: three are
as good at 20 batches and better at 8, which is why bench.py's short legs keep three). Rounds 4-5 measured four and more as UNSTABLE from run to run (2.8-4.0 M at four,
1.8-3.7 M at six) when every capsule still owned a second stream for its result copies; with one stream per capsule the spread at
every count is below 2 %. `SolverRing` refuses more than MAX_STABLE_SLOTS = 4 capsules -- beyond the hardware queues capsules share a
queue and their batches serialise in an order the caller does not control -- unless the caller insists (`allow_unstable=True`).

Results on the host ride the same ring: `request_results(slot)` enqueues, behind the solve on that capsule's stream, the copy
of the batch's results into one of the capsule's two pinned host slabs and an event (C-ABI: tum_ocp_results_async);
`take_results(slot)` waits for the event of the capsule's OLDEST request only and returns views of its slab
(tum_ocp_results_wait). The pattern that keeps the GPU busy: when capsule k mod S comes round again, enqueue batch k and its
request FIRST, then take the results of batch k - S -- the capsule's stream already holds the next batch while the host reads
the previous one, and S batches of GPU work lie between a request and its wait (measured on config 2, three capsules:
host-visible rate = device rate, bench.py `value_host_visible`; reading BEFORE enqueuing leaves the stream dry for the host's
reaction time once per batch, 3.06 against 3.8 M solves/s; a synchronous read after every solve: 2.9 M).

This is host logic above the C-ABI (which already allows any number of capsules per process).
A closed loop -- where solve k + 1 needs the result of solve k -- has nothing to overlap and keeps using one capsule.
"""


MAX_STABLE_SLOTS = 4          # capsules (= HIP streams) a ring runs with a reproducible rate on an MI355X (see above)


class SolverRing:
    def __init__(self, n_slots, factory, streams=None, allow_unstable=False):
        """factory(slot) -> a configured BatchedOcpSolver (all slots must describe the same OCP); streams: optional list of
        raw hipStream_t handles (ints), one per slot (default: every capsule keeps the non-blocking stream it created).
        More than MAX_STABLE_SLOTS capsules are refused (ValueError) unless allow_unstable."""
        if n_slots < 1:
            raise ValueError("n_slots < 1")
        if n_slots > MAX_STABLE_SLOTS and not allow_unstable:
            # (rounds 4-5 cut the ring back with a warning: a caller that sized its own per-slot structures by ITS number then indexed
            #  past the ring. Refused instead; len(ring) / ring.n_slots is the number of capsules a ring has.)
            raise ValueError(f"SolverRing: {n_slots} capsules asked for, at most {MAX_STABLE_SLOTS} are useful -- with more streams than hardware queues "
                             f"capsules share a queue and the measured rate falls (profiles/r06_stream_counts.txt); "
                             f"allow_unstable=True overrides")
        self.solvers = [factory(i) for i in range(n_slots)]
        if streams is not None:
            if len(streams) != n_slots:
                raise ValueError("one stream per slot")
            for s, st in zip(self.solvers, streams):
                s.set_stream(st)
        self._next = 0
        self._pending = [0] * n_slots          # outstanding result requests per capsule (the C-ABI allows two)

    def __len__(self):
        return len(self.solvers)

    @property
    def n_slots(self):
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

    def request_results(self, slot, with_iterate=False):
        """Behind everything enqueued on capsule `slot` so far: copy its batch's results (summary; with_iterate: X and U too)
        into the capsule's pinned host slab, asynchronously."""
        self.solvers[slot].results_async(with_iterate)
        self._pending[slot] += 1

    def outstanding(self, slot):
        return self._pending[slot]

    def take_results(self, slot):
        """(summary, X, U) views of the pinned slabs of capsule `slot`'s OLDEST outstanding request once its copies have landed --
        None when no request is outstanding. A capsule has two sets of slabs: the views stay valid until the second-next
        request_results on this slot. To keep a capsule's stream busy, enqueue the next batch and its request first and take
        the older results afterwards (`outstanding(slot) == 2`)."""
        if not self._pending[slot]:
            return None
        self._pending[slot] -= 1
        return self.solvers[slot].results_wait()

    def drain(self):
        """(slot, results) of every outstanding request, in the order the batches were enqueued"""
        n = len(self.solvers)
        for want in (2, 1):          # capsules with two requests hold the older batches
            for d in range(n):
                i = (self._next + d) % n
                if self._pending[i] == want:
                    yield i, self.take_results(i)

    def synchronize(self):
        for s in self.solvers:
            s.synchronize()
