"""CPU oracle for the UniRes ADMM y-update hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``unires_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the CPU number reported
beside the GPU number.

PARITY UNPINNED at the nitorch boundary: the arithmetic of the hot path lives in
the third-party package ``nitorch`` (pinned by the reference at
``setup.py:11`` to commit 8067d60542642a39ab6c6eb5e1157373a9d3dcc3), which is
neither vendored under /root/reference nor installed in the build container,
and the reference has no tests/golden vectors for this path (SURVEY.md §4,
§8(c)).  ``nitorch_restated`` therefore restates nitorch's *published*
algorithm from its documented behaviour; it is pinned only by
  * torch-native partial oracles that ARE importable here
    (``F.grid_sample(align_corners=True)``, ``F.conv3d``/``F.conv_transpose3d``,
    autograd adjoints),
  * algebraic properties (adjointness, SPD, CG-vs-dense solve),
  * the few known answers the reference's demo notebooks print
    (``get_gain`` trace, ``_proj_info`` dimension arithmetic).
``unires_restated`` restates the reference's own files line by line and cites
them - and IS pinned: ``tests/golden/make_golden_from_reference.py`` imports the
reference's ``unires/_project.py`` and ``_update.py`` as they lie (with ``nitorch``
bound to ``nitorch_restated``), runs them on five small seeded problems and writes
their inputs and outputs to ``tests/golden/ref_*.npz``;
``tests/test_reference_pin.py`` requires ``unires_restated`` to reproduce every one
of them to 1e-6.  What remains unpinned is ``nitorch_restated`` alone.
"""
