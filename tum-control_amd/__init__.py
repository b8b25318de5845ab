"""
tum-control_amd -- MI355X-native batched SQP-RTI solver behind the AcadosOcpSolver surface used by
bzarr/TUM-CONTROL's NMPC controllers (drop-in for `acados_solver.solve()` and nothing else).

  csrc/            hand-written HIP (gfx950) kernels + the C-ABI (include/tum_nmpc.h) -> libtumnmpc.so
  solver.py        ctypes binding with acados method names (BatchedOcpSolver)
  config.py        EDGAR vehicle / tyre / MPC constants
  (more host-side mirrors of the reference's controller interface are added as the path widens)
"""
__version__ = "0.1.0"
