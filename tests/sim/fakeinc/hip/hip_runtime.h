// empty: the simulator's shim (tests/sim/hip_sim.h) is force-included instead
