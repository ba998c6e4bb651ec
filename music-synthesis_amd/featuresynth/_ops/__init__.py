"""MI355X (gfx950) operators of the featuresynth hot path: ctypes binding of the C ABI
(lib), raw tensor-level primitives (prims), whole-network forward/backward schedules (graph)
and the torch.autograd.Function wrappers (functional)."""
