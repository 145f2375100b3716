"""CPU oracle: a restatement of the reference's algorithms for the hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``stamp_amd/`` imports this package; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may.  Floating-point paths are restated
with plain torch fp32 CPU ops (the reference itself is torch code dispatched to ATen CPU kernels), integer /
bit-level paths with numpy.  Each function cites the reference file:line it follows; how each piece is pinned
(golden vectors captured from the imported reference, or "parity unpinned") is stated per module and in
DESIGN.md.
"""
