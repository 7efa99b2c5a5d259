"""CPU oracle for the hash-grid / SDF-MLP / ray-tracing hot path.

TEST INFRASTRUCTURE ONLY - parity status: PINNED by tests/golden/*.npz (reference outputs
generated in the build container by tests/golden/make_goldens.py).  The product package
(hashmodnffbanks_idr_amd) never imports this; only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg do.

  c_oracle   - ctypes bindings of oracle/hm_oracle.c (integer/byte work + fp32 forward)
  raytrace   - numpy restatement of model/ray_tracing.py on top of an `sdf` callable
  torch_ref  - torch-CPU fp32 restatement with autograd (floating-point grads only)
"""
