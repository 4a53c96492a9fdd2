"""Drop-in mirror of the reference's `sgm` plugin surface for the V3D dense-multi-view hot path.

Every class here keeps the reference's dotted path (with `sgm.` -> `v3d_amd.sgm.`), constructor params,
forward signature and state-dict keys, so a YAML is switched over by rewriting its `target:` strings only
(reference mechanism: sgm/util.py:170-187).  Compute runs on the hand-written gfx950 kernels through
v3d_amd.ops; none of the reference's code is imported.
"""
