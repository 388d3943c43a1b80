"""CPU oracle for the per-image correction chain.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(``scanspaceimageprocessor_amd``) may import this package; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do,
and there only as the checker / the timed CPU restatement.

Parity status
-------------
* The reference's own numpy code around the third-party leaves
  (``ImageProcessor/editingTools.py``: negative inversion, exposure gain,
  shadow/highlight curve, unsharp-mask mask logic, ordering and pre-scaling in
  ``apply_all_adjustments``) is PINNED: ``oracle/make_golden.py`` executes that
  file verbatim (loaded by path, leaves stubbed with ``oracle.leaves``) and the
  resulting vectors are committed under ``tests/golden/``.
* The leaves themselves (colour-science 0.4.x ``colour_correction``, sRGB
  CCTFs, sRGB<->XYZ<->Lab; OpenCV 4.x ``GaussianBlur``) are restated from
  their published definitions because neither package is installed here and
  the reference ships no tests or fixtures:  PARITY UNPINNED at the leaves.
  They are guarded by known-answer tests: analytic identities, published
  tables, and the worked examples of colour-science's own docstrings
  (tests/test_oracle_leaves.py).
"""

from . import leaves, chain  # noqa: F401
