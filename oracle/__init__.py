"""CPU oracle for the SSD300 hot path -- TEST INFRASTRUCTURE ONLY.

Nothing in the product package (``object-detection-tensorflow_amd/``) may import
this package.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker / the timed
CPU baseline, never as the thing shipped.

Parity status: the reference (``/root/reference``) is TF-1.13 graph code and
cannot run in this environment (no ``tensorflow`` module; ``SSD300.py:41-43``
does not parse).  Two things pin this restatement instead:

* ``oracle/tf_shim`` executes the reference's OWN ``_get_abbox`` /
  ``_compute_one_image_loss`` / inference-branch Python (loaded from
  ``/root/reference/SSD300.py`` at fixture-generation time, never copied) on a
  small eager TF-1.x API shim; ``tests/golden/make_golden.py`` stores the results
  as fixtures and ``tests/test_oracle_golden.py`` checks this restatement
  against them.
* TF *kernel* semantics inside that shim (SAME padding, NonMaxSuppressionV3,
  fused batch-norm, argmax tie rules ...) are restated from the TF 1.13 sources
  from memory: that layer is "parity unpinned" (SURVEY.md Appendix B).
"""
